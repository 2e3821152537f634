/* rt_decode.h — host-side block decoders that consume the front end's flux-transition events.
 *
 * This is the part of the pipeline that stays sequential on the host:
 * the per-format bit/clock recovery machines, the AGC / clock-average helpers they own, the
 * parameter sets, the retry/selection driver and the SIMH .tap writer.  It is a from-scratch
 * restatement organised around one explicit decoder context (`rt_dec`) instead of the reference's
 * process-wide globals, so that any number of decoders (one per parmset of a batched sweep, one per
 * time shard) can run side by side.
 *
 * Every function cites the reference lines whose behaviour it reproduces; float/double promotion
 * points are kept exactly as written there (parity depends on them).
 *
 * Two front ends drive it through the same calls:
 *   - the HIP front end (product): events replayed by rt_replay.c
 *   - the scalar CPU restatement in oracle/ (test infrastructure only)
 */
#ifndef RT_DECODE_H
#define RT_DECODE_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_MAXTRKS        19      /* src/csvtbin.h:29 */
#define RT_MAXBLOCK       131072  /* src/decoder.h:91 */
#define RT_MAXPARMSETS    15      /* src/decoder.h:92 */
#define RT_CLKRATE_WINDOW 50      /* src/decoder.h:142 */
#define RT_AGC_MAX_WINDOW 10      /* src/decoder.h:152 */
#define RT_PKWW_MAX_WIDTH 50      /* src/decoder.h:132 */
#define RT_MAXSKEWSAMP    50      /* src/decoder.h:97 */

enum rt_mode { RT_UNKNOWN = 0, RT_PE = 1, RT_NRZI = 2, RT_GCR = 4, RT_WW = 8 };   /* src/csvtbin.h:47-49 */

enum rt_bstate {            /* src/decoder.h:318-325 */
   RT_BS_NONE, RT_BS_TAPEMARK, RT_BS_NOISE, RT_BS_BADBLOCK, RT_BS_BLOCK, RT_BS_ABORTED };

struct rt_parms {           /* src/decoder.h:290-310 (the fields that matter) */
   int   active;
   int   clk_window;
   float clk_alpha;
   int   agc_window;
   float agc_alpha;
   float min_peak;
   float clk_factor;
   float pulse_adj;
   float pkww_bitfrac;
   float pkww_rise;
   float midbit;
   float z1pt;
   float z2pt;
   int   tried, chosen;
};

struct rt_clkavg {          /* src/decoder.h:185-192 */
   float t_bitspacing[RT_CLKRATE_WINDOW];
   int   bitndx;
   float t_bitspaceavg;
};

/* Per-track state shared between the front end and the block decoders
 * (the subset of src/decoder.h:194-255 that crosses the seam, plus decoder-private fields).
 * Detector-private state (window, countdown, deskew FIFO, zero-cross FSM) is NOT here: it lives in
 * the front end (device, or oracle/). */
struct rt_trk {
   int    trknum;
   float  v_now;            /* written by the front end; PE/GCR idle code copies it to v_lastpeak */
   float  v_top;   double t_top;   float v_lasttop;
   float  v_bot;   double t_bot;   float v_lastbot;  double t_lastbot;
   float  v_lastpeak;
   double t_lastpeak, t_prevlastpeak;
   float  t_peakdelta, t_peakdeltaprev;
   double t_lastpulsestart, t_lastpulseend;
   float  v_avg_height, v_avg_height_sum;
   int    v_avg_height_count;
   float  agc_gain, max_agc_gain, min_agc_gain;
   float  v_heights[RT_AGC_MAX_WINDOW];
   int    heightndx;
   double t_lastbit, t_firstbit, t_lastclock;
   int    consecutive_zeroes;
   float  t_clkwindow, t_pulse_adj;
   uint8_t bit1_up;
   struct rt_clkavg clkavg;
   int    datacount, peakcount;
   uint8_t lastdatabit, idle, clknext, datablock;
   uint8_t lastbits;
   int    resync_bitcount;
};

struct rt_results {         /* src/decoder.h:333-358 */
   enum rt_bstate blktype;
   int   minbits, maxbits;
   float avg_bit_spacing;
   int   warncount, missed_midbits, corrected_bits, gcr_bad_dgroups;
   int   ww_leading_clock, ww_missing_onebit, ww_missing_clock;
   uint16_t faked_tracks;
   int   errcount, track_mismatch, vparity_errs, ecc_errs, crc_errs, lrc_errs;
   int   gcr_bad_sequence, ww_bad_length, ww_speed_err;
   int   first_error, crc, lrc;
   float alltrk_max_agc_gain, alltrk_min_agc_gain;
};

struct rt_nrzi {            /* src/decoder.h:266-273 */
   double t_lastclock, t_last_midbit;
   struct rt_clkavg clkavg;
   uint8_t datablock, reset_speed;
   int    post_counter;
};

struct rt_ww {              /* src/decoder.h:275-287: what survives from block to block */
   struct rt_clkavg clkavg;
   uint8_t datablock, blockmark_queued;
   int    datacount;        /* 2-bit characters so far */
   double t_lastpeak;       /* last peak on any track */
   double t_lastclkpulsestart, t_lastclkpulseend, t_lastpriclkpulsestart, t_lastaltclkpulsestart, t_lastpriclkpulseend, t_lastblockmark;
};
enum { RT_FLUX_NEG = 0, RT_FLUX_POS = 1, RT_FLUX_AUTO = 2 };      /* -fluxdir= (src/readtape.c:966-968) */

struct rt_options {         /* the command-line switches that reach the decoders (src/readtape.c:936-1022) */
   enum rt_mode mode;
   int   ntrks;
   float bpi, ips;
   int   specified_parity;  /* 1 = odd (default) */
   int   revparity;
   int   do_correction;     /* -correct */
   int   find_zeros;        /* -zeros */
   int   do_differentiate;  /* -differentiate */
   int   multiple_tries;    /* -m */
   int   tap_format;        /* -tap */
   int   add_parity;
   int   verbose;           /* -v: log every block */
   int   ww_fluxdir;        /* Whirlwind: RT_FLUX_NEG (default) / POS / AUTO */
   int   ww_reverse;        /* Whirlwind -reverse: the tape was read backwards */
   char  ww_order[24];      /* Whirlwind -order=: a role letter per head (CLMclm, x = unused) */
};

#define RT_ESTDEN_NUMBINS 150      /* src/decoder.c:333-338 */
#define RT_ESTDEN_COUNTNEEDED 9999
#define RT_PEAKSTAT_BUCKETS 50     /* src/decoder.c:121 */
#define RT_MAXSKEWSAMP 50          /* src/decoder.h:97-99 */
#define RT_MAXSKEWBLKS 100
#define RT_MINSKEWTRANS 1000

struct rt_dec {
   struct rt_options opt;
   struct rt_parms   parmsets[RT_MAXPARMSETS];
   float   sample_deltat;           /* seconds, float (src/readtape.c:1345) */
   int64_t sample_deltat_ns;
   /* run state */
   double  timenow;                 /* time of the sample being processed (src/decoder.c:92) */
   int     interblock_counter;      /* src/decoder.c:97 */
   int     num_trks_idle;
   int     expected_parity;
   struct rt_trk trk[RT_MAXTRKS];
   struct rt_nrzi nrzi;
   struct rt_ww ww;
   int     ww_type_to_trk[6], ww_trk_to_type[RT_MAXTRKS];      /* role <-> track (src/readtape.c:521-522) */
   int     fatal;                 /* the reader met a condition on which the reference exits (assert in lookfor_peak / refine_peak): nothing more is attempted */
   int     ww_prepassed;          /* a -deskew pre-pass has set up the track state (src/readtape.c:1674 happens once per tape) */
   int     flux_current, num_flux_polarity_changes;
   struct { int bitnum, bytenum; uint8_t sgroup[9]; int bad_parity_in_dgroup; } gcr;   /* src/decode_gcr.c:37,444-445 */
   /* block state (src/decoder.h:327-359) */
   int     tries, parmset;
   uint8_t window_set, endblock_done;
   double  t_blockstart;
   struct rt_results results[RT_MAXPARMSETS];
   uint16_t *data, *data_faked;     /* [RT_MAXBLOCK+1] */
   double   *data_time;
   /* output / bookkeeping (src/readtape.c:497-520) */
   FILE   *tapf;
   long long numoutbytes, numdatabytes;
   int no_tap_end;                    /* fragment decode: the caller writes the end-of-medium marker behind the last fragment */
   /* output files by name (src/readtape.c:1084-1111): with outbase set, <outbase>.tap (-tap) or the numbered <outbase>.NNN.bin
    * files (one per tape file, i.e. a new one behind every tapemark) are created when the first block wants them, and logged */
   char    outbase[1024], outname[1100];
   int     numfiles, numfileblks;
   long long numfilebytes;
   long long lines_in;                /* sample rows read by first attempts (src/readtape.c:1404): what the summary calls "samples" */
   double  data_start_time, last_block_time;
   int     numblks, numtapemarks, numblks_err, numblks_warn, numblks_unusable;
   int     numblks_goodmultiple, numblks_trksmismatched, numblks_midbiterrs, numblks_corrected;
   FILE   *logf;                    /* block log lines (NULL = quiet) */
   /* flux-transition position statistics of the -deskew pre-pass (src/decoder.c:121-173): only gathered while
    * doing_deskew (in the reference they also feed a .csv report, which is out of scope) */
   int     doing_deskew;
   /* density detection (src/decoder.c:329-394, src/readtape.c:1656-1672): while bpi is unknown every transition goes
    * to a histogram of transition distances instead of a block decoder */
   int     doing_density_detection;
   struct { int deltas[RT_ESTDEN_NUMBINS], counts[RT_ESTDEN_NUMBINS], binsused, totalcount, fatal; } estden;
   struct { int initialized; float leftbin, binwidth; int counts[RT_MAXTRKS][RT_PEAKSTAT_BUCKETS]; int trksums[RT_MAXTRKS]; } peakstat;
   /* optional observer: called at the top of every up/down transition, before the format callback
    * (the same seam oracle/ref_event_shim.c wraps in the reference) */
   void  (*on_transition)(struct rt_dec *d, struct rt_trk *t, int is_top, void *user);
   void  (*on_attempt)(struct rt_dec *d, void *user);
   void   *user;
};
#define RT_PARM(d) ((d)->parmsets[(d)->parmset])

/* ---- lifetime ---- */
struct rt_dec *rt_dec_new(const struct rt_options *opt, float sample_deltat, int64_t sample_deltat_ns);
void rt_dec_free(struct rt_dec *d);
void rt_default_parmsets(enum rt_mode mode, struct rt_parms out[RT_MAXPARMSETS]);   /* src/parmsets.c:77-118 */
int  rt_parse_parms_text(enum rt_mode mode, const char *text, struct rt_parms out[RT_MAXPARMSETS]); /* src/parmsets.c:236-327 */
int  rt_pkww_width(const struct rt_dec *d, int parmset);            /* src/readtape.c:1455-1457 */
int  rt_samples_per_bit(const struct rt_dec *d);                    /* src/readtape.c:1402 */

/* ---- shared helpers (src/decoder.c:401-609) ---- */
void rt_init_blockstate(struct rt_dec *d);
void rt_init_trackstate(struct rt_dec *d);
void rt_init_clkavg(struct rt_clkavg *c, float init_avg);
void rt_adjust_clock(struct rt_dec *d, struct rt_clkavg *c, float delta, int trk);
void rt_force_clock(struct rt_clkavg *c, float delta);
void rt_adjust_agc(struct rt_dec *d, struct rt_trk *t);
void rt_up_transition(struct rt_dec *d, struct rt_trk *t);     /* t->v_top / t->t_top already set */
void rt_down_transition(struct rt_dec *d, struct rt_trk *t);   /* t->v_bot / t->t_bot already set */
void rt_set_expected_parity(struct rt_dec *d, int blklength);
int  rt_parity9(uint16_t w);

/* ---- per-sample control: the timers of process_sample (src/decoder.c:841-894), split into
 *      predicates (pure) and actions so an event-driven replay can schedule them ---- */
int  rt_nrzi_zerocheck_due(const struct rt_dec *d);            /* src/decoder.c:844 */
void rt_nrzi_zerocheck(struct rt_dec *d);                      /* src/decode_nrzi.c:232-314 */
void rt_nrzi_end_of_block(struct rt_dec *d);                   /* src/decode_nrzi.c:77-113 */
int  rt_pe_idle_due(const struct rt_dec *d, const struct rt_trk *t);   /* src/decoder.c:868 */
void rt_pe_go_idle(struct rt_dec *d, struct rt_trk *t);        /* src/decoder.c:871-877 */
void rt_pe_end_of_block(struct rt_dec *d);                     /* src/decode_pe.c:33-102 */
int  rt_gcr_idle_due(const struct rt_dec *d, const struct rt_trk *t);  /* src/decoder.c:879-880 */
int  rt_gcr_go_idle(struct rt_dec *d, struct rt_trk *t);       /* src/decoder.c:881-888; returns 1 if block ended */
void rt_gcr_end_of_block(struct rt_dec *d);                    /* src/decode_gcr.c:682-729 */
void rt_force_end_of_block(struct rt_dec *d);                  /* src/readtape.c:1378-1381 */
void rt_finish_attempt(struct rt_dec *d);                      /* src/readtape.c:1508-1515 */

/* Whirlwind (src/decode_ww.c) */
int  rt_ww_assign_roles(struct rt_dec *d, const char *order, int *head_to_trk);   /* src/readtape.c:869-902; returns ntrks or -1 */
void rt_ww_init_blockstate(struct rt_dec *d);                 /* src/decode_ww.c:33-49 */
int  rt_ww_end_due(const struct rt_dec *d);                   /* src/decoder.c:892-894 */
void rt_ww_end_of_block(struct rt_dec *d);                    /* src/decode_ww.c:141-161 */
void rt_ww_blockmark(struct rt_dec *d);                       /* src/decode_ww.c:163-167 */
void rt_ww_top(struct rt_dec *d, struct rt_trk *t);
void rt_ww_bot(struct rt_dec *d, struct rt_trk *t);

/* format callbacks (src/decode_*.c) */
void rt_nrzi_top(struct rt_dec *d, struct rt_trk *t);
void rt_nrzi_bot(struct rt_dec *d, struct rt_trk *t);
void rt_pe_top(struct rt_dec *d, struct rt_trk *t);
void rt_pe_bot(struct rt_dec *d, struct rt_trk *t);
void rt_pe_generate_fake_bits(struct rt_dec *d, struct rt_trk *t);
void rt_gcr_top(struct rt_dec *d, struct rt_trk *t);
void rt_gcr_bot(struct rt_dec *d, struct rt_trk *t);
void rt_gcr_preprocess(struct rt_dec *d);

/* ---- output (src/readtape.c:1076-1111, 1160-1313, 1885) ---- */
void rt_got_tapemark(struct rt_dec *d);
void rt_got_datablock(struct rt_dec *d, int badblock);
void rt_tap_end(struct rt_dec *d);
void rt_close_output(struct rt_dec *d);                       /* src/readtape.c:1084-1089 */
void rt_write_summary(struct rt_dec *d, const char *infilename, double elapsed_seconds);   /* src/readtape.c:2021-2044 */

/* ---- the block retry / selection driver (src/readtape.c:1720-1882) over an abstract block reader.
 * `readblock(ctx, retry)` must run one attempt with d->parmset from the saved position and return
 * 0 at end of data (like !readblock()); `save_pos` / `restore_pos` mirror save/restore_file_position. */
struct rt_reader {
   int  (*readblock)(void *ctx, int retry);
   void (*save_pos)(void *ctx);
   void (*restore_pos)(void *ctx);
   void *ctx;
};
int rt_process_blocks(struct rt_dec *d, struct rt_reader *r, int blklimit);   /* returns 1 if all blocks clean */

/* ---- the -deskew pre-pass (src/readtape.c:1675-1717, NRZI and GCR): decodes the first blocks with no deskew while the
 * decoders record where each track's transitions fall, then turns the per-track averages into delays (in samples).
 * Returns the number of blocks used (>= 0), or -1 if some track saw no transition; *hit_end = 1 if the reader ran out
 * of data before the reference's stopping rule was met (the caller may then retry on a longer prefix of the tape). */
void rt_record_peakstat(struct rt_dec *d, float bitspacing, float peaktime, int trknum);   /* src/decoder.c:136-173 */

/* ---- density detection (src/readtape.c:1656-1672): reads attempts with bpi = 0 until RT_ESTDEN_COUNTNEEDED transition
 * distances are in the histogram (or the data ends), then picks the standard density.  Returns the density (d->opt.bpi
 * is set to it), 0 if the implied density is not close to a standard one, -1 if a transition distance was not positive or the
 * histogram ran out of bins (all three are fatal in the reference, src/decoder.c:355,362,397); *implied = the raw
 * estimate, *nblks = non-noise attempts read, *hit_end = the reader ran out of data first. */
float rt_density_prepass(struct rt_dec *d, struct rt_reader *r, float *implied, int *nblks, int *hit_end);
int  rt_deskew_prepass(struct rt_dec *d, struct rt_reader *r, int delays[RT_MAXTRKS], int *hit_end);

#ifdef __cplusplus
}
#endif
#endif
