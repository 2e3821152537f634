/* rt_csv.c — CSV ingest: a logic-analyser export ("time, v0, v1, ..." behind two title lines) becomes the
 * int16 rows + TBIN header fields the device front end takes.  The numbers are the ones the reference's converter writes
 * (src/csvtbin.c:619-716): the sample period from the first and last timestamps of the pre-read, the full-scale voltage from
 * the largest magnitude seen (plus 0.5 V, rounded to 0.1 V) unless a larger one is given, and round-half-away quantisation
 * clamped to +-32767.  Text -> number conversion accumulates digit by digit in the precision of the result, as the reference's
 * scanner does (src/csvtbin.c scanfast_*): a correctly rounded strtof() would differ in the last bit now and then.
 */
#include "rt_csv.h"

#include <stdio.h>
#include <string.h>

enum { LINE_MAX_CHARS = 400,            /* MAXLINE,       src/csvtbin.c:123 */
       PREREAD_ROWS   = 1000000 };      /* PREREAD_COUNT, src/csvtbin.c:125 */

/* one decimal number at *p (blanks and commas in front of it skipped), accumulated in `double` */
static double scan_f64(const char **p) {
   const char *s = *p;
   while (*s == ' ' || *s == ',') ++s;
   const int neg = *s == '-';
   if (neg) ++s;
   double v = 0;
   for (; *s >= '0' && *s <= '9'; ++s) v = v * 10 + (*s - '0');
   if (*s == '.') {
      double scale = 10;
      for (++s; *s >= '0' && *s <= '9'; ++s, scale *= 10) v += (*s - '0') / scale; }
   *p = s;
   return neg ? -v : v; }

/* ... accumulated in `float` */
static float scan_f32(const char **p) {
   const char *s = *p;
   while (*s == ' ' || *s == ',') ++s;
   const int neg = *s == '-';
   if (neg) ++s;
   float v = 0;
   for (; *s >= '0' && *s <= '9'; ++s) v = v * 10 + (*s - '0');
   if (*s == '.') {
      float scale = 10;
      for (++s; *s >= '0' && *s <= '9'; ++s, scale *= 10) v += (*s - '0') / scale; }
   *p = s;
   return neg ? -v : v; }

static int next_line(FILE *f, char *line) {
   if (!fgets(line, LINE_MAX_CHARS, f)) return 0;
   line[LINE_MAX_CHARS - 1] = 0;
   return 1; }

int rt_csv_survey(const char *path, int ntrks, float scale, int subsample, float maxvolts_given, struct rt_csv_info *out) {
   char line[LINE_MAX_CHARS + 1];
   FILE *f = fopen(path, "r");
   if (!f) return -1;
   memset(out, 0, sizeof *out);
   if (!next_line(f, line) || !next_line(f, line)) { fclose(f); return -2; }        /* the two title lines */
   for (const char *c = line; *c; ++c) out->columns += *c == ',';
   double t_first = -1;
   float peak = 0;
   int64_t n = 0;
   uint32_t tdelta = 0;
   while (next_line(f, line) && ++n < PREREAD_ROWS) {
      const char *p = line;
      const double t = scan_f64(&p);
      if (t_first < 0) { t_first = t; out->tstart_ns = (uint64_t)((t_first + 0.5e-9) * 1e9); }
      else tdelta = (uint32_t)(((t - t_first) / (double)(n - 1) + 0.5e-9) * 1e9);
      for (int k = 0; k < ntrks; ++k) {
         float v = scan_f32(&p) * scale;
         if (v < 0) v = -v;
         if (peak < v) peak = v; } }
   /* the rows of the whole file (the pre-read stops at a million) */
   int64_t rows = n;                                   /* (n counted the line on which the pre-read stopped, too) */
   if (n >= PREREAD_ROWS) while (next_line(f, line)) ++rows;
   fclose(f);
   peak = ((float)(int)((peak + 0.55f) * 10.0f)) / 10.0f;
   if (subsample > 1) { out->tstart_ns += (uint64_t)(subsample - 1) * tdelta; tdelta *= (uint32_t)subsample; }
   out->tdelta_ns = tdelta;
   out->maxvolts = maxvolts_given > peak ? maxvolts_given : peak;
   out->rows = rows / (subsample > 1 ? subsample : 1);
   return 0; }

int64_t rt_csv_load(const char *path, int ntrks, const int *perm, int invert, float scale, int subsample, float maxvolts,
                    int16_t *rows, int64_t capacity, int64_t *clipped) {
   char line[LINE_MAX_CHARS + 1];
   if (ntrks < 1 || ntrks > RT_CSV_MAXTRKS) return -3;                          /* (v[] below holds RT_CSV_MAXTRKS columns) */
   if (perm) for (int k = 0; k < ntrks; ++k) if (perm[k] < 0 || perm[k] >= ntrks) return -4;
   FILE *f = fopen(path, "r");
   if (!f) return -1;
   if (!next_line(f, line) || !next_line(f, line)) { fclose(f); return -2; }
   if (subsample < 1) subsample = 1;
   int64_t n = 0, clips = 0;
   float v[RT_CSV_MAXTRKS];
   for (;;) {
      int got = 1;
      for (int s = 0; s < subsample && got; ++s) got = next_line(f, line);      /* of every `subsample` lines the last one counts */
      if (!got || n >= capacity) break;
      const char *p = line;
      (void)scan_f64(&p);                                                      /* the timestamp: the period is fixed by now */
      for (int k = 0; k < ntrks; ++k) v[perm ? perm[k] : k] = scan_f32(&p) * scale;
      int16_t *o = rows + n * ntrks;
      for (int k = 0; k < ntrks; ++k) {
         const float x = invert ? -v[k] : v[k];
         int q = (int)((x / maxvolts * 32767) + (x < 0 ? -0.5f : 0.5f));        /* (all float: (int) truncates towards zero) */
         if (q <= -32767) { q = -32767; ++clips; }
         if (q >= 32767) { q = 32767; ++clips; }
         o[k] = (int16_t)q; }
      ++n; }
   fclose(f);
   if (clipped) *clipped = clips;
   return n; }
