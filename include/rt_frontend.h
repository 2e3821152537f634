/* rt_frontend.h — C ABI of the MI355X analog front end (librtfe.so).
 *
 * The drop-in boundary.  The reference (LenShustek/readtape V3.18) has no plugin or FFI surface;
 * its de-facto seam is   bool readblock(bool retry)            src/readtape.c:1396
 * which loops            enum bstate_t process_sample(...)     src/decoder.c:817 (decl. src/decoder.h:384)
 * and, beneath it, the callback set the front end drives:
 *        void {nrzi,pe,gcr,ww}_{top,bot}(struct trkstate_t*)   src/decoder.h:385-399
 * after  init_trackstate()                                     src/decoder.c:425.
 * This library replaces everything between those two seams: given the TBIN payload resident in HBM
 * it produces, per block attempt and per parameter set, exactly the sequence of top/bot calls the
 * reference's process_sample() would make (track, polarity, peak time, voltage, detection sample,
 * AGC gain), as 16-byte event records.  INTEGRATION.md shows the reference-side stub.
 *
 * Plain pointers and sizes only; every pointer whose name starts with d_ is a DEVICE pointer
 * (hipMalloc / torch storage).  All calls are asynchronous on `stream` (a hipStream_t passed as
 * void*, NULL = default stream) unless stated; nothing here synchronises the device.  (rtfe_scan may run one of
 * its kernels - the burst heads - on a stream of its own beside the dense pass; it forks from and joins back into
 * `stream` with events, so everything rtfe_scan wrote is complete when `stream` reaches the point behind the call.)
 *
 * Exactness contract.  The reference restarts all detector state at every block attempt
 * (src/decoder.c:425-455), at a sample only its sequential bit decoders know.  rtfe_scan()
 * speculates: it restarts inside every inter-block quiet zone ("burst" boundary) and reports, per
 * burst, the interval [zone_first, safe_last] of restart samples for which its output is PROVABLY
 * identical to the reference's (dead-quiet signal, full windows, common min/max rescan — see
 * DESIGN.md §3).  A caller whose true restart sample falls outside the interval, or whose block
 * does not end inside a zone, calls rtfe_scan_exact() for that one attempt: same kernels, caller-
 * given restart sample, no speculation.  Neither path computes anything on the CPU.
 */
#ifndef RT_FRONTEND_H
#define RT_FRONTEND_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTFE_MAXTRKS     19   /* src/csvtbin.h:29  MAXTRKS      */
#define RTFE_MAXPARMSETS 15   /* src/decoder.h:92  MAXPARMSETS  */
#define RTFE_ABI_VERSION 6   /* 6: rtfe_reset_floor (a handle's first scan estimates the candidate screen's floor from the samples); 5: rtfe_set_graphs; 4: rtfe_pack_events; 3: rtfe_scan_stats out[21] = the smallest learned peak height (screen-floor calibration); 2: rtfe_kernel_ms returns the number of scans summed; twelve timed spans (k_dseg, k_dchain) */

enum { RTFE_PE = 1, RTFE_NRZI = 2, RTFE_GCR = 4, RTFE_WW = 8 };      /* enum mode_t, src/csvtbin.h:47-49 */

/* the front-end half of struct parms_t (src/decoder.h:290-310) */
typedef struct rtfe_parmset {
   float   pkww_bitfrac;     /* window width as a fraction of a bit cell              */
   float   pkww_rise;        /* required rise, volts at 4 V p-p                       */
   float   min_peak;         /* absolute minimum peak, volts (0 = none)               */
   float   agc_alpha;        /* exponential AGC weight (0 = off)                      */
   int32_t agc_window;       /* min-of-last-n AGC window (0 = off)                    */
   float   clk_factor;       /* PE only: clock window factor (decides preamble end)   */
} rtfe_parmset;

typedef struct rtfe_config {
   int32_t mode;                          /* RTFE_NRZI ...                                        */
   int32_t ntrks;                         /* tracks == heads in a TBIN row                        */
   int32_t head_to_trk[RTFE_MAXTRKS];     /* column -> track permutation (src/readtape.c:1419)    */
   int32_t invert;                        /* -invert        (src/readtape.c:1421)                 */
   int32_t differentiate;                 /* -differentiate (src/readtape.c:1383-1388)            */
   int32_t find_zeros;                    /* -zeros         (src/decoder.c:863-865)               */
   int32_t skew_delaycnt[RTFE_MAXTRKS];   /* -skew=n,n,..   in samples (src/decoder.c:232)        */
   float   maxvolts;                      /* TBIN header                                          */
   float   bpi, ips;                      /* bpi = 0: density unknown -> the front end of the reference's density pre-pass
                                           * (window of 8 samples, no AGC feedback; src/readtape.c:1457,1656-1672; NRZI peaks only) */
   int64_t tdelta_ns;                     /* sample period                                        */
   int64_t tstart_ns;                     /* time of row 0                                        */
   int32_t nparmsets;
   rtfe_parmset parmset[RTFE_MAXPARMSETS];
   /* speculation / screening knobs (performance only; results are exact or flagged) */
   int32_t gap_min_samples;               /* quiet run treated as an inter-block gap; 0 = 32 bit cells */
   float   quiet_volts;                   /* |v| band of the dead-quiet test; 0 = default         */
   float   screen_floor_height;           /* assumed lower bound of the AGC baseline (v_avg_height) for
                                             the candidate screen.  A burst whose measured baseline is lower is
                                             flagged RTFE_F_SCREEN_UNDERFLOW.  0 = the handle follows the tape: its first
                                             scan of the peak path estimates the floor from the samples before it screens
                                             (0.45 x the smallest peak-to-peak range inside a block; 1.0 V where it finds
                                             no block), and behind each scan the floor moves to half the smallest peak
                                             height the scan's chains learned (rtfe_scan_stats out[22]; rtfe_reset_floor) */
   float   events_per_sample_cap;         /* per-track event capacity as a fraction of the burst length;
                                             0 = default 1/8 */
} rtfe_config;

/* One flux-transition event = one call of {mode}_top/_bot in the reference (src/decoder.c:574-609). */
typedef struct rtfe_event {
   uint32_t sample;          /* detection sample (the reference's `timenow` row), relative to the burst's reset_sample */
   float    v_peak;          /* t->v_top / t->v_bot, volts                                          */
   float    agc_gain;        /* t->agc_gain when the callback is entered                            */
   uint8_t  trk;
   uint8_t  flags;           /* bit0: 0 = top (up transition), 1 = bottom; bits 1-2: time adjustment
                                code 0 = none, 1 = -0.5 sample, 2 = +0.5 sample (src/decoder.c:718-730).
                                bit 7 (RTFE_EV_FATAL): not a transition - at this row the reference's
                                "AGC gain bad in lookfor_peak" assert (src/decoder.c:782) fires on this track,
                                which ends the whole run there (exit 99); nothing follows it on the track */
   uint8_t  left_distance;   /* 1-based position of the peak in the window (src/decoder.c:704-744)  */
   uint8_t  parmset;
} rtfe_event;                /* 16 bytes */

/* peak time exactly as refine_peak forms it (src/decoder.c:732); W = pkww_width of the parmset:
 *   timenow = (double)(tstart_ns + (reset_sample + sample) * tdelta_ns) / 1e9
 *   t_peak  = timenow - ((float)(W - left_distance) - adj) * sample_deltat          */

enum {
   RTFE_F_EXACT_START      = 1,    /* reset_sample was given by the caller (or is row 0)             */
   RTFE_F_UNSAFE           = 2,    /* no provably-equivalent restart interval exists for this burst  */
   RTFE_F_EVENT_OVERFLOW   = 4,    /* a track's event region filled up; events were dropped          */
   RTFE_F_SCREEN_UNDERFLOW = 8,    /* AGC threshold fell below the candidate screen: rescan exactly with screen off */
   RTFE_F_TRUNCATED        = 32,   /* time shard: the halo ended before this burst's successor zone did          */
   RTFE_F_DETECTOR_FATAL   = 16,   /* the reference would have hit its fatal "peak at window edge" assert (src/decoder.c:709-710,748) */
   RTFE_F_AGC_FATAL        = 128,  /* informational: a track's event list ends in an RTFE_EV_FATAL marker                     */
   RTFE_F_STATE_AT_END     = 64    /* -zeros: at end_sample a track still holds an excursion beyond the 0.2 V threshold or a pending
                                    * crossing - history a restart would not have, even if no event was emitted: an attempt that
                                    * reaches the end of this burst must not continue into the next one */
};

#define RTFE_EV_FATAL 0x80

typedef struct rtfe_burst {
   int64_t  zone_first;      /* first row of the dead-quiet zone that precedes this burst            */
   int64_t  zone_end;        /* one past the last row of that zone                                    */
   int64_t  reset_sample;    /* row at which this burst's detector state was restarted                */
   int64_t  safe_last;       /* any true restart row in [zone_first, safe_last] gives identical events */
   int64_t  end_sample;      /* rows [reset_sample, end_sample) were scanned with this burst's state  */
   uint64_t event_base;      /* index into the event buffer of this burst's first region              */
   uint32_t event_cap;       /* capacity of each (parmset, track) region                              */
   uint32_t flags;           /* RTFE_F_*                                                              */
} rtfe_burst;                /* 56 bytes */
/* region of (burst b, parmset p, track t):  events[ b.event_base + (p * ntrks + t) * b.event_cap ... ],
 * count in counts[(burst_index * nparmsets + p) * ntrks + t]; events of one region are in detection order. */

typedef struct rtfe_handle rtfe_handle;

int         rtfe_abi_version(void);
const char *rtfe_last_error(void);

/* Validates the configuration and precomputes window widths, thresholds and the candidate screen.
 * Returns 0 or a negative error (message via rtfe_last_error). Synchronous, no device work. */
int  rtfe_create(const rtfe_config *cfg, rtfe_handle **out);
void rtfe_destroy(rtfe_handle *h);
int  rtfe_pkww_width(const rtfe_handle *h, int parmset);      /* src/readtape.c:1455-1457 */

/* Sizes the caller must provide for a scan of nrows rows. */
size_t  rtfe_workspace_bytes(const rtfe_handle *h, int64_t nrows);
int64_t rtfe_max_bursts(const rtfe_handle *h, int64_t nrows);
int64_t rtfe_event_capacity(const rtfe_handle *h, int64_t nrows);

/* Speculative scan of d_rows[0 .. nrows) (interleaved int16, ntrks per row, the TBIN payload).
 * Time shards: row_base is the absolute index of d_rows[0] on the tape (used for times only; all rows in
 * the outputs are relative to d_rows[0]); first_is_tape_start says row 0 is a true restart point (the
 * start of the tape); own_rows <= nrows is the part of the slice this scan OWNS — rows
 * [own_rows, nrows) are the halo received from the right neighbour.  A burst is decoded here iff its
 * zone ends at or before own_rows; the last owned burst runs on into the halo up to the next zone's
 * restart row.  If the halo holds no further zone the last owned burst is flagged RTFE_F_TRUNCATED
 * (pass a longer halo).  own_rows == nrows: no halo (single GPU, or the last shard).
 * Outputs (device): bursts[*nbursts] (owned bursts only), counts, events. */
int rtfe_scan(rtfe_handle *h, const int16_t *d_rows, int64_t nrows, int64_t own_rows, int64_t row_base, int first_is_tape_start,
              void *d_workspace, size_t workspace_bytes,
              rtfe_burst *d_bursts, int64_t max_bursts, int32_t *d_nbursts,
              uint32_t *d_counts, rtfe_event *d_events, int64_t event_capacity,
              void *stream);

/* The scans of ONE handle read and move its device-resident screen (above): queue them on one stream, or order the streams yourself.
 * rtfe_reset_floor: the screen back to what rtfe_create made (queued on `stream`) - the next scan is a tape's first scan again (a new tape on an old
 * handle; bench.py's "...f" lines time exactly that). */
int rtfe_reset_floor(rtfe_handle *h, void *stream);

/* Exact scan of one block attempt: detector state restarts at `reset_row` (index into d_rows) and
 * rows [reset_row, end_row) are scanned; parmset_mask selects parameter sets.  With screen_off != 0
 * every sample is examined (use after RTFE_F_SCREEN_UNDERFLOW).  Writes one rtfe_burst. */
int rtfe_scan_exact(rtfe_handle *h, const int16_t *d_rows, int64_t nrows, int64_t row_base,
                    int64_t reset_row, int64_t end_row, uint32_t parmset_mask, int screen_off,
                    void *d_workspace, size_t workspace_bytes,
                    rtfe_burst *d_burst, uint32_t *d_counts, rtfe_event *d_events, int64_t event_capacity,
                    void *stream);

/* The event lists of a scan, packed on the device (what crosses PCIe to the host replay: the arena is laid out for the worst case, event_cap
 * records per list; the lists hold a fraction of it).  Queued on `stream` behind the rtfe_scan whose outputs it reads:
 *   d_plan[b], b < *d_nbursts: burst b's lists in d_packed under the burst table's own addressing rule -
 *     d_packed[plan.event_base + (p * ntrks + t) * plan.event_cap + i], plan.event_cap = the burst's longest list (>= 1);
 *   d_plan[*d_nbursts].event_base = records of d_packed in use (.reserved = the burst count the plan was made for).  If that exceeds
 *     packed_capacity nothing was copied: fetch the arena as it is, or pack again into a larger buffer.
 * d_plan holds max_bursts + 1 entries.  The scan's own outputs are not changed. */
typedef struct rtfe_pack_entry { uint64_t event_base; uint32_t event_cap; uint32_t reserved; } rtfe_pack_entry;      /* 16 bytes */
int rtfe_pack_events(rtfe_handle *h, const rtfe_burst *d_bursts, const int32_t *d_nbursts, int64_t max_bursts, const uint32_t *d_counts,
                     const rtfe_event *d_events, rtfe_event *d_packed, uint64_t packed_capacity, rtfe_pack_entry *d_plan, void *stream);

/* The end of the data inside a window of the payload: *d_first = the first row of d_rows[0 .. nrows) whose head-0 sample is 0x8000 (the reader of
 * src/readtape.c:1410 stops there), INT64_MAX if there is none.  Queued on `stream`; a streaming reader runs it behind a window's upload instead of
 * passing over the window on the host. */
int rtfe_find_end_mark(rtfe_handle *h, const int16_t *d_rows, int64_t nrows, int64_t *d_first, void *stream);

/* Per-kernel timing: with enable != 0, rtfe_scan records HIP events on `stream` around each of its timed spans (rtfe_kernel_name), one
 * set of events per scan in a ring of 64 sets - nothing waits, so scans queued back to back stay back to back.  rtfe_kernel_ms synchronises
 * the sets recorded since its last call and returns, per span, the elapsed milliseconds SUMMED over those scans (out[rtfe_kernel_count()]);
 * its return value is the number of scans summed (>= 0), or a negative error.  Used by bench.py. */
int rtfe_set_timing(rtfe_handle *h, int enable);

/* HIP graphs (ABI 5).  enable != 0: rtfe_scan captures its launches - about twenty kernels and memsets on the caller's stream and a stream of the handle's
 * own - into a hipGraph the first time it meets a set of arguments (pointers, sizes, flags) and replays that graph for every later scan with the same
 * arguments: one launch on the host, no gaps between the kernels on the device.  The handle keeps the eight most recently used graphs (a streaming reader's
 * ring of windows).  Results are the same either way.  Scans on the legacy default stream (stream == NULL), with per-kernel timing on or with debug
 * counters are launched directly; so is everything if the runtime refuses the capture.  The buffers a graph names must stay allocated while scans with
 * those arguments may still come; rtfe_destroy frees the graphs.  Environment default: RTFE_GRAPHS. */
int rtfe_set_graphs(rtfe_handle *h, int enable);
int rtfe_kernel_ms(rtfe_handle *h, float *out);

/* Statistics of the most recent rtfe_scan that used d_workspace (synchronous, call after the stream has finished):
 * out[0] = bursts decoded, out[1] = of which the record chains handed to the exact sample path, out[2] = bytes of peak
 * records written, out[3] / out[4] = detections the chains decided 64 runs at a time / one at a time.  Diagnostics for bench.py
 * and the tests; no effect on results.  out[5..12] = chains that gave up, by reason; out[13..20] = cycle counters of k_sift / k_gain / k_bursts / k_gain_seg (RTFE_DEBUG=3 / 4 / 5 / 6).
 * out[21] = 0x7fffffff - the IEEE bits of the smallest v_avg_height (src/decode_nrzi.c:224-229) a chain of the scan learned, 0 if none did (peak path):
 * what a caller may raise rtfe_config::screen_floor_height towards for the next scans of the same tape - a floor above a later chain's learned height is
 * flagged RTFE_F_SCREEN_UNDERFLOW and costs an exact rescan, never a wrong event.  out[22] = the IEEE bits of the floor the handle's next scan screens against:
 * with rtfe_config::screen_floor_height == 0 the handle's first scan estimates it from the samples and every scan of the peak path moves it, behind itself, to half
 * the smallest height its chains learned (on the device, nothing waits; a floor the caller gives stands).  out[23] = the IEEE bits of the floor THIS scan's screen was built for (peak path).
 * out must hold 24 values. */
int rtfe_scan_stats(rtfe_handle *h, const void *d_workspace, int64_t *out);

/* Names and launch-order of the kernels of one scan, for profilers (static strings). */
int         rtfe_kernel_count(void);
const char *rtfe_kernel_name(int i);

/* ---- Whirlwind (mode RTFE_WW): detector state that survives block attempts ----
 * The reference never restarts its detector between Whirlwind blocks (src/readtape.c:1674, src/decode_ww.c:31-49); a new attempt
 * only re-seeds every track's window (ring slot 0 and both extremes, src/decoder.c:855-861), one track per sample.  Where an
 * attempt starts is the host decoder's decision, so the host hands the state in and gets it back:
 *   rtfe_ww_scan: rows [first_row, first_row + nscan) of d_rows from the state d_state_in (one rtfe_ww_track per track; a tape
 *   starts from rtfe_ww_initial_state), the attempt they belong to having started at seed_row0 (<= first_row: track t sits out
 *   rows < seed_row0 + t and is re-seeded at seed_row0 + t).  Out: per-track event lists (event.sample relative to first_row;
 *   d_events[t * event_capacity ...], d_counts[t]), the state after the last row (d_state_out, may alias d_state_in), *d_flags |=
 *   RTFE_F_* conditions.  One parameter set (the reference forbids -m for Whirlwind), peak detection only.  -deskew: the reference
 *   learns the delays AND the pulse heights in a pre-pass over the first blocks, then clears the windows and starts over
 *   (src/readtape.c:1676-1716): the host replay does the same with this call - pre-pass from the initial state, then left/right/
 *   maxv/minv/countdown = 0, v_avg_height and delay set per track in the blob, and the decode from row 0.
 * Replaces: the per-sample lookfor_peak of src/decoder.c:751-810 for mode WW, at the seam of src/decoder.c:586,604. */
typedef struct rtfe_ww_track {
   int16_t ring[64];                /* pkww_v as int16 codes (after -invert) */
   int32_t left, right, maxv, minv, countdown, peakcount, heightndx;
   int32_t delay;                   /* -deskew: this track's delay in samples (0..50; the FIFO of src/decoder.c:820-830 starts at row 0 of the tape) */
   float   agc_gain, v_avg_height, v_lasttop, v_lastbot, v_top, v_bot;
   float   heights[10];
} rtfe_ww_track;                     /* 224 bytes */
void rtfe_ww_initial_state(rtfe_ww_track *tracks, int ntrks);        /* init_trackstate, src/decoder.c:425-455 (host memory) */
int rtfe_ww_scan(rtfe_handle *h, const int16_t *d_rows, int64_t nrows, int64_t row_base, int64_t first_row, int64_t nscan, int64_t seed_row0,
                 const rtfe_ww_track *d_state_in, rtfe_ww_track *d_state_out, uint32_t *d_counts, rtfe_event *d_events, int64_t event_capacity,
                 uint32_t *d_flags, void *stream);

#ifdef __cplusplus
}
#endif
#endif
