/* rt_replay.h — event replay: device front-end output -> host block decoders -> SIMH .tap (see rt_replay.c) */
#ifndef RT_REPLAY_H
#define RT_REPLAY_H
#include "rt_decode.h"
#include "rt_frontend.h"

#ifdef __cplusplus
extern "C" {
#endif

/* asks the caller for an exact device scan (rtfe_scan_exact) of one attempt: rows [reset_row, end_row) of the
 * scanned slice, one parameter set.  On success fills *burst, counts[ntrks], *events (regions of *cap events per
 * track, track-major) and returns 0. */
typedef int  (*rt_exact_fn)(void *user, int64_t reset_row, int64_t end_row, int parmset,
                            rtfe_burst *burst, uint32_t *counts, rtfe_event **events, uint32_t *cap);
typedef void (*rt_exact_free_fn)(void *user, rtfe_event *events);

/* Whirlwind: the device's peak detector keeps its state across attempts, so the replay fetches its events in chunks of rows, handing the
 * state in and getting it back (include/rt_frontend.h: rtfe_ww_scan).  first_row / nscan / seed_row0 as there; state blobs are
 * ntrks rtfe_ww_track each (host memory); events: [ntrks][cap] (sample relative to first_row), counts[ntrks].  Returns 0, or the
 * RTFE_F_* flags of a scan that could not be delivered. */
typedef int (*rt_ww_scan_fn)(void *user, int64_t first_row, int64_t nscan, int64_t seed_row0, const void *state_in, void *state_out,
                             uint32_t *counts, rtfe_event *events, uint32_t cap);

struct rt_replay {
   struct rt_dec *d;
   int     ntrks, nparm;
   int     W[RT_MAXPARMSETS];          /* rtfe_pkww_width() per parameter set */
   int64_t nrows, row_base, tstart_ns, tdelta_ns;
   const rtfe_burst *bursts; int64_t nbursts;
   const uint32_t   *counts;
   const rtfe_event *events;
   rt_exact_fn exact; rt_exact_free_fn exact_free; void *exact_user;
   int64_t pos, saved_pos, cur_row; double saved_time;
   int64_t stop_row;                   /* fragment decode: no attempt starts at or behind this row */
   /* Whirlwind */
   rt_ww_scan_fn ww_scan; void *ww_user;
   unsigned char *ww_state, *ww_chunk_state, *ww_chunk_end;   /* ntrks rtfe_ww_track each: at the attempt's start / at the chunk's first row / behind the chunk */
   rtfe_event *ww_events; uint32_t *ww_counts; uint32_t ww_cap; int64_t ww_chunk_rows;
   int     find_zeros;                 /* events are confirmed zero crossings: the slope gate is applied here */
   FILE   *evtf;                       /* optional dump of every delivered transition (oracle/ref_event_shim.c record format) */
   /* statistics */
   int64_t attempts, exact_scans, chained, events_delivered, agc_mismatches;
   int64_t device_failures;            /* attempts the device could not deliver (exact scan refused / overflowed): the decode stops there */
   int64_t reference_fatal, fatal_row, fatal_trk;   /* an RTFE_EV_FATAL marker was reached: the reference exits there (src/decoder.c:782) */
};

int  rt_replay_readblock(void *ctx, int retry);
void rt_replay_save_pos(void *ctx);
void rt_replay_restore_pos(void *ctx);

struct rt_replay_stats {
   int64_t attempts, exact_scans, chained, events_delivered, agc_mismatches;
   int32_t blocks, tapemarks, blocks_with_errors, blocks_with_warnings, blocks_unusable, all_ok;
   int64_t data_bytes;
   int64_t device_failures;            /* > 0: the decode stopped early because the device could not deliver an attempt */
   int64_t reference_fatal, fatal_row, fatal_trk;   /* != 0: the decode stopped where the reference's "AGC gain bad" assert ends its run */
};

/* Decodes a whole scanned tape: builds a decoder for `opt` (+ `parmsets`, NULL = built-in sets), replays the
 * device output through the retry/selection driver and writes tap_path / log_path (NULL = none). */
int rt_replay_run(const struct rt_options *opt, const struct rt_parms *parmsets, int nparm,
                  int64_t tdelta_ns, int64_t tstart_ns, int64_t nrows, int64_t row_base, const int *W,
                  const rtfe_burst *bursts, int64_t nbursts, const uint32_t *counts, const rtfe_event *events,
                  rt_exact_fn exact, rt_exact_free_fn exact_free, void *user,
                  const char *tap_path, const char *log_path, const char *evt_path, struct rt_replay_stats *stats);


/* -deskew (src/readtape.c:1675-1717): the pre-pass over a scan made WITHOUT deskew delays -> delays[ntrks] in samples,
 * *nblks = blocks used (-1: a track without transitions), *hit_end = the scan ended before the stopping rule was met.
 * The decode of the re-scanned tape then continues the same log / event-dump files (rt_replay_run_after_deskew);
 * append != 0: the pre-pass itself continues files a density detection started. */
int rt_replay_deskew(const struct rt_options *opt, const struct rt_parms *parmsets, int nparm,
                  int64_t tdelta_ns, int64_t tstart_ns, int64_t nrows, int64_t row_base, const int *W,
                  const rtfe_burst *bursts, int64_t nbursts, const uint32_t *counts, const rtfe_event *events,
                  rt_exact_fn exact, rt_exact_free_fn exact_free, void *user,
                  const char *log_path, const char *evt_path, int append, int *delays, int *nblks, int *hit_end);
/* density detection (src/readtape.c:1656-1672) over a scan made with bpi = 0 (window of 8, no AGC feedback):
 * *bpi = the standard density chosen (0: the implied density *implied is not close to one), *nblks, *hit_end as above. */
int rt_replay_density(const struct rt_options *opt, const struct rt_parms *parmsets, int nparm,
                  int64_t tdelta_ns, int64_t tstart_ns, int64_t nrows, int64_t row_base, const int *W,
                  const rtfe_burst *bursts, int64_t nbursts, const uint32_t *counts, const rtfe_event *events,
                  rt_exact_fn exact, rt_exact_free_fn exact_free, void *user,
                  const char *log_path, const char *evt_path, float *bpi, float *implied, int *nblks, int *hit_end);
int rt_replay_run_fragment(const struct rt_options *opt, const struct rt_parms *parmsets, int nparm,
                  int64_t tdelta_ns, int64_t tstart_ns, int64_t nrows, int64_t row_base, const int *W,
                  const rtfe_burst *bursts, int64_t nbursts, const uint32_t *counts, const rtfe_event *events,
                  rt_exact_fn exact, rt_exact_free_fn exact_free, void *user,
                  const char *tap_path, const char *log_path, const char *evt_path, struct rt_replay_stats *stats,
                  int64_t start_row, int64_t stop_row);
/* The same for nfrag fragments of one scan side by side, a thread each (the streaming reader's sub-fragments of a window: cut at burst
 * boundaries, each with its own decoder context and its own piece of the .tap).  stats[nfrag], seconds[nfrag] (each fragment's own time).
 * Returns 0, or the first failing fragment's code. */
int rt_replay_run_fragments(const struct rt_options *opt, const struct rt_parms *parmsets, int nparm,
                  int64_t tdelta_ns, int64_t tstart_ns, int64_t nrows, int64_t row_base, const int *W,
                  const rtfe_burst *bursts, int64_t nbursts, const uint32_t *counts, const rtfe_event *events,
                  rt_exact_fn exact, rt_exact_free_fn exact_free, void *user,
                  int nfrag, const char *const *tap_paths, const int64_t *start_rows, const int64_t *stop_rows, struct rt_replay_stats *stats, double *seconds);
/* The reader's positional read of nbytes at file offset `off` into dst, as nthreads pieces side by side (from the page cache one thread copies
 * ~13 GB/s).  Returns 0, -1 on a short read or an error. */
int rt_read_mt(int fd, void *dst, int64_t off, int64_t nbytes, int nthreads);
int rt_replay_run_ww(const struct rt_options *opt, const struct rt_parms *parmsets,
                  int64_t tdelta_ns, int64_t tstart_ns, int64_t nrows, int W0, rt_ww_scan_fn scan, void *user, const void *initial_state, int64_t chunk_rows,
                  const char *tap_path, const char *out_base, const char *in_name, const char *log_path, const char *evt_path, struct rt_replay_stats *stats,
                  int deskew, int *delays_out);
int rt_replay_run_named(const struct rt_options *opt, const struct rt_parms *parmsets, int nparm,
                  int64_t tdelta_ns, int64_t tstart_ns, int64_t nrows, int64_t row_base, const int *W,
                  const rtfe_burst *bursts, int64_t nbursts, const uint32_t *counts, const rtfe_event *events,
                  rt_exact_fn exact, rt_exact_free_fn exact_free, void *user,
                  const char *out_base, const char *in_name, const char *log_path, const char *evt_path, int append, struct rt_replay_stats *stats);
int rt_replay_run_after_deskew(const struct rt_options *opt, const struct rt_parms *parmsets, int nparm,
                  int64_t tdelta_ns, int64_t tstart_ns, int64_t nrows, int64_t row_base, const int *W,
                  const rtfe_burst *bursts, int64_t nbursts, const uint32_t *counts, const rtfe_event *events,
                  rt_exact_fn exact, rt_exact_free_fn exact_free, void *user,
                  const char *tap_path, const char *log_path, const char *evt_path, struct rt_replay_stats *stats);

#ifdef __cplusplus
}
#endif
#endif
