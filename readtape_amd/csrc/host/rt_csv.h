/* rt_csv.h — CSV ingest (replaces the reference's CSV path, src/readtape.c:1426-1448 and its converter src/csvtbin.c:619-716). */
#ifndef RT_CSV_H
#define RT_CSV_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define RT_CSV_MAXTRKS 19
struct rt_csv_info {
   int      columns;        /* data columns the second title line announces */
   int64_t  rows;           /* sample rows behind the title lines (after subsampling) */
   uint64_t tstart_ns;      /* time of the first used sample */
   uint32_t tdelta_ns;      /* sample period */
   float    maxvolts;       /* full scale for the int16 codes */
};
/* first pass: period, start time, full scale (maxvolts_given = 0: derive it), row count */
int rt_csv_survey(const char *path, int ntrks, float scale, int subsample, float maxvolts_given, struct rt_csv_info *out);
/* second pass: rows[n][ntrks] int16 codes, column k of the file going to column perm[k] (NULL = identity); returns the rows written */
int64_t rt_csv_load(const char *path, int ntrks, const int *perm, int invert, float scale, int subsample, float maxvolts,
                    int16_t *rows, int64_t capacity, int64_t *clipped);
#ifdef __cplusplus
}
#endif
#endif
