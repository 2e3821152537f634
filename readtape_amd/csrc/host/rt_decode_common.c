/* rt_decode_common.c — decoder context, AGC / clock helpers, transition bookkeeping.
 * Restates src/decoder.c:401-609 and the window/parameter setup of src/readtape.c:1402,1453-1457
 * around an explicit context.  No front-end (per-sample detector) code lives here. */
#include "rt_decode.h"

#include <float.h>
#include <stdlib.h>
#include <string.h>

#define PKWW_PEAKHEIGHT 4.0f      /* src/decoder.h:133 */
#define AGC_MAX_VALUE   2.0f      /* src/decoder.h:153 */
#define RT_DATA_GUARD   64

struct rt_dec *rt_dec_new(const struct rt_options *opt, float sample_deltat, int64_t sample_deltat_ns) {
   struct rt_dec *d = (struct rt_dec *)calloc(1, sizeof *d);
   if (!d) return NULL;
   d->opt = *opt;
   d->sample_deltat = sample_deltat;
   d->sample_deltat_ns = sample_deltat_ns;
   /* The reference's data[] / data_time[] are static arrays, and its NRZI mid-bit check can take a track's datacount to
    * -1 (src/decode_nrzi.c:265 with datacount == 0: garbage in, e.g. noise on a differentiated signal), after which one
    * bit is written in front of the array - harmless there, a smashed heap header here.  RT_DATA_GUARD elements in front
    * of each array make it harmless here too (what lands there is never part of a block). */
   d->data = (uint16_t *)calloc(RT_MAXBLOCK + 1 + RT_DATA_GUARD, sizeof(uint16_t)) + RT_DATA_GUARD;
   d->data_faked = (uint16_t *)calloc(RT_MAXBLOCK + 1 + RT_DATA_GUARD, sizeof(uint16_t)) + RT_DATA_GUARD;
   d->data_time = (double *)calloc(RT_MAXBLOCK + 1 + RT_DATA_GUARD, sizeof(double)) + RT_DATA_GUARD;
   d->expected_parity = opt->specified_parity;
   rt_default_parmsets(opt->mode, d->parmsets);
   d->flux_current = RT_FLUX_AUTO;                            /* src/readtape.c:515 */
   if (opt->mode == RT_WW && rt_ww_assign_roles(d, d->opt.ww_order[0] ? d->opt.ww_order : "CMLcml", NULL) != opt->ntrks) { rt_dec_free(d); return NULL; }
   return d; }

void rt_dec_free(struct rt_dec *d) {
   if (!d) return;
   free(d->data - RT_DATA_GUARD); free(d->data_faked - RT_DATA_GUARD); free(d->data_time - RT_DATA_GUARD);
   free(d); }

int rt_samples_per_bit(const struct rt_dec *d) {   /* src/readtape.c:1402 */
   return d->opt.bpi > 0 ? (int)(1 / (d->opt.bpi * d->opt.ips * d->sample_deltat)) : 20; }

int rt_pkww_width(const struct rt_dec *d, int parmset) {   /* src/readtape.c:1455-1457 */
   if (d->opt.bpi) {
      int w = (int)(d->parmsets[parmset].pkww_bitfrac / (d->opt.bpi * d->opt.ips * d->sample_deltat));
      return w < RT_PKWW_MAX_WIDTH ? w : RT_PKWW_MAX_WIDTH; }
   return 8; }

int rt_parity9(uint16_t val) {      /* src/readtape.c:1038-1041 */
   uint16_t p = val & 1;
   while (val >>= 1) p ^= val & 1;
   return p; }

void rt_init_blockstate(struct rt_dec *d) {       /* src/decoder.c:401-405 */
   for (int i = 0; i < RT_MAXPARMSETS; ++i) {
      memset(&d->results[i], 0, sizeof(struct rt_results));
      d->results[i].blktype = RT_BS_NONE; } }

void rt_init_clkavg(struct rt_clkavg *c, float init_avg) {   /* src/decoder.c:407-411 */
   c->t_bitspaceavg = init_avg;
   c->bitndx = 0;
   for (int i = 0; i < RT_CLKRATE_WINDOW; ++i) c->t_bitspacing[i] = init_avg; }

void rt_init_trackstate(struct rt_dec *d) {       /* src/decoder.c:425-455 */
   int ntrks = d->opt.ntrks;
   float bpi = d->opt.bpi, ips = d->opt.ips;
   if (d->on_attempt) d->on_attempt(d, d->user);
   d->num_trks_idle = ntrks;
   d->window_set = 0;
   d->endblock_done = 0;
   d->expected_parity = d->opt.specified_parity;
   if (d->opt.mode == RT_GCR) rt_gcr_preprocess(d);
   memset(&d->results[d->parmset], 0, sizeof(struct rt_results));
   d->results[d->parmset].blktype = RT_BS_NONE;
   d->results[d->parmset].alltrk_max_agc_gain = 0.0;
   d->results[d->parmset].alltrk_min_agc_gain = FLT_MAX;
   memset(d->trk, 0, sizeof d->trk);
   for (int trknum = 0; trknum < ntrks; ++trknum) {
      struct rt_trk *t = &d->trk[trknum];
      t->trknum = trknum;
      t->idle = 1;
      t->agc_gain = 1.0;
      t->max_agc_gain = 0.0;
      t->min_agc_gain = FLT_MAX;
      t->v_avg_height = PKWW_PEAKHEIGHT;
      if (!d->doing_density_detection) rt_init_clkavg(&t->clkavg, 1 / (bpi * ips));
      t->t_clkwindow = t->clkavg.t_bitspaceavg / 2 * RT_PARM(d).clk_factor; }
   if (d->opt.mode == RT_NRZI) {
      memset(&d->nrzi, 0, sizeof d->nrzi);
      if (!d->doing_density_detection) rt_init_clkavg(&d->nrzi.clkavg, 1 / (bpi * ips)); }
   if (d->opt.mode == RT_WW) {                                /* (Whirlwind: once per tape, src/readtape.c:1674) */
      memset(&d->ww, 0, sizeof d->ww);
      if (!d->doing_density_detection) rt_init_clkavg(&d->ww.clkavg, 1 / (bpi * ips)); } }

void rt_set_expected_parity(struct rt_dec *d, int blklength) {   /* src/decoder.c:457-460 */
   d->expected_parity = blklength > 0 && blklength == d->opt.revparity
                        ? 1 - d->opt.specified_parity : d->opt.specified_parity; }

void rt_adjust_agc(struct rt_dec *d, struct rt_trk *t) {   /* src/decoder.c:500-531 */
   if (d->opt.find_zeros) return;
   const struct rt_parms *P = &RT_PARM(d);
   float gain, lastheight;
   if (P->agc_alpha) {
      lastheight = t->v_lasttop - t->v_lastbot;
      if (lastheight > 0) {
         gain = t->v_avg_height / lastheight;
         gain = P->agc_alpha * gain + (1 - P->agc_alpha) * t->agc_gain;
         if (gain > AGC_MAX_VALUE) gain = AGC_MAX_VALUE;
         t->agc_gain = gain;
         if (gain > t->max_agc_gain) t->max_agc_gain = gain;
         if (gain < t->min_agc_gain) t->min_agc_gain = gain; } }
   if (P->agc_window) {
      lastheight = t->v_lasttop - t->v_lastbot;
      if (lastheight > 0) {
         t->v_heights[t->heightndx] = lastheight;
         if (++t->heightndx >= P->agc_window) t->heightndx = 0;
         float minheight = 99;
         for (int i = 0; i < P->agc_window; ++i) if (t->v_heights[i] < minheight) minheight = t->v_heights[i];
         gain = t->v_avg_height / minheight;
         if (gain > AGC_MAX_VALUE) gain = AGC_MAX_VALUE;
         t->agc_gain = gain;
         if (gain > t->max_agc_gain) t->max_agc_gain = gain;
         if (gain < t->min_agc_gain) t->min_agc_gain = gain; } } }

void rt_adjust_clock(struct rt_dec *d, struct rt_clkavg *c, float delta, int trk) {   /* src/decoder.c:533-555 */
   int clk_window = RT_PARM(d).clk_window;
   float clk_alpha = RT_PARM(d).clk_alpha;
   (void)trk;
   if (clk_window > 0) {
      float olddelta = c->t_bitspacing[c->bitndx];
      c->t_bitspacing[c->bitndx] = delta;
      if (++c->bitndx >= clk_window) c->bitndx = 0;
      c->t_bitspaceavg += (delta - olddelta) / clk_window; }
   else if (clk_alpha > 0) {
      c->t_bitspaceavg = clk_alpha * delta + (1 - clk_alpha) * c->t_bitspaceavg; }
   else {
      c->t_bitspaceavg = (d->opt.mode & (RT_PE + RT_WW)) ? 1 / (d->opt.bpi * d->opt.ips)
                                                          : d->nrzi.clkavg.t_bitspaceavg; } }

void rt_force_clock(struct rt_clkavg *c, float delta) {   /* src/decoder.c:556-558 */
   for (int i = 0; i < RT_CLKRATE_WINDOW; ++i) c->t_bitspacing[i] = delta;
   c->t_bitspaceavg = delta; }

static void process_transition(struct rt_dec *d, struct rt_trk *t) {   /* src/decoder.c:560-572 */
   ++t->peakcount;
   if (t->idle) {
      --d->num_trks_idle;
      t->idle = 0;
      if (d->opt.mode == RT_PE && t->datablock && t->datacount > 1)
         rt_pe_generate_fake_bits(d, t); } }

/* one transition distance into the histogram (src/decoder.c:351-367); returns 1 when enough have been seen */
static int estden_transition(struct rt_dec *d, float deltasecs) {
   const int delta = (int)(deltasecs / 0.5e-6);                /* ESTDEN_BINWIDTH: float / double, truncated */
   if (!(deltasecs > 0)) { d->estden.fatal = 1; return 1; }    /* "negative delta ... in estden_transition": fatal in the reference (:355) */
   if (deltasecs > 0 && deltasecs <= 120e-6) {                 /* ESTDEN_MAXDELTA */
      int ndx = 0;
      while (ndx < d->estden.binsused && d->estden.deltas[ndx] != delta) ++ndx;
      if (ndx >= d->estden.binsused) {
         if (d->estden.binsused >= RT_ESTDEN_NUMBINS) { d->estden.fatal = 1; return 1; }   /* "too many transition delta values": fatal (:362) */
         d->estden.deltas[d->estden.binsused++] = delta; }
      ++d->estden.counts[ndx];
      ++d->estden.totalcount; }
   return d->estden.totalcount >= RT_ESTDEN_COUNTNEEDED; }

void rt_up_transition(struct rt_dec *d, struct rt_trk *t) {   /* src/decoder.c:574-590 */
   process_transition(d, t);
   if (d->on_transition && !d->doing_density_detection) d->on_transition(d, t, 1, d->user);   /* (the observer sits at the format callbacks) */
   if (d->doing_density_detection) {                        /* src/decoder.c:578-581, 596-599 */
      if (estden_transition(d, (float)(t->t_top - t->t_lastpeak))) d->results[d->parmset].blktype = RT_BS_ABORTED; }
   else switch (d->opt.mode) {
   case RT_PE:   rt_pe_top(d, t); break;
   case RT_NRZI: rt_nrzi_top(d, t); break;
   case RT_GCR:  rt_gcr_top(d, t); break;
   case RT_WW:   rt_ww_top(d, t); break;
   default: break; }
   t->v_lasttop = t->v_top;
   t->v_lastpeak = t->v_top;
   t->t_prevlastpeak = t->t_lastpeak;
   t->t_lastpeak = t->t_top; }

void rt_down_transition(struct rt_dec *d, struct rt_trk *t) {   /* src/decoder.c:592-609 */
   process_transition(d, t);
   if (d->on_transition && !d->doing_density_detection) d->on_transition(d, t, 0, d->user);   /* (the observer sits at the format callbacks) */
   if (d->doing_density_detection) {                        /* src/decoder.c:578-581, 596-599 */
      if (estden_transition(d, (float)(t->t_bot - t->t_lastpeak))) d->results[d->parmset].blktype = RT_BS_ABORTED; }
   else switch (d->opt.mode) {
   case RT_PE:   rt_pe_bot(d, t); break;
   case RT_NRZI: rt_nrzi_bot(d, t); break;
   case RT_GCR:  rt_gcr_bot(d, t); break;
   case RT_WW:   rt_ww_bot(d, t); break;
   default: break; }
   t->v_lastbot = t->v_bot;
   t->t_lastbot = t->t_bot;
   t->v_lastpeak = t->v_bot;
   t->t_prevlastpeak = t->t_lastpeak;
   t->t_lastpeak = t->t_bot; }

void rt_force_end_of_block(struct rt_dec *d) {   /* src/readtape.c:1378-1381 */
   if (d->opt.mode == RT_PE) rt_pe_end_of_block(d);
   else if (d->opt.mode == RT_NRZI && d->nrzi.datablock) rt_nrzi_end_of_block(d);
   else if (d->opt.mode == RT_GCR) rt_gcr_end_of_block(d); }

void rt_finish_attempt(struct rt_dec *d) {   /* src/readtape.c:1508-1515 */
   struct rt_results *r = &d->results[d->parmset];
   r->errcount = r->track_mismatch + r->vparity_errs + r->ecc_errs + r->crc_errs + r->lrc_errs
                 + r->gcr_bad_sequence + r->ww_bad_length + r->ww_speed_err;
   r->warncount = r->missed_midbits + r->corrected_bits + r->gcr_bad_dgroups
                  + r->ww_leading_clock + r->ww_missing_onebit + r->ww_missing_clock; }

/* ---- PE / GCR idle timers of process_sample (src/decoder.c:868-888) ---- */
#define PE_IDLE_FACTOR  2.5f    /* src/decoder.h:115 */
#define GCR_IDLE_THRESH 6.00    /* src/decoder.h:111 (a double literal) */

int rt_pe_idle_due(const struct rt_dec *d, const struct rt_trk *t) {
   return !t->idle && t->t_lastpeak != 0 && d->timenow - t->t_lastpeak > t->clkavg.t_bitspaceavg * PE_IDLE_FACTOR; }

void rt_pe_go_idle(struct rt_dec *d, struct rt_trk *t) {
   t->v_lastpeak = t->v_now;
   t->idle = 1;
   if (++d->num_trks_idle >= d->opt.ntrks) rt_pe_end_of_block(d); }

int rt_gcr_idle_due(const struct rt_dec *d, const struct rt_trk *t) {
   return t->datablock && d->timenow > t->t_lastpeak + GCR_IDLE_THRESH * t->clkavg.t_bitspaceavg; }

int rt_gcr_go_idle(struct rt_dec *d, struct rt_trk *t) {
   t->datablock = 0;
   t->idle = 1;
   if (++d->num_trks_idle >= d->opt.ntrks) {
      rt_gcr_end_of_block(d);
      return 1; }
   return 0; }

/* ---- flux-transition position statistics (src/decoder.c:136-173; the adjdeskew averages are experimental there
 *      and not restated) ---- */
void rt_record_peakstat(struct rt_dec *d, float bitspacing, float peaktime, int trknum) {
   if (!d->peakstat.initialized) {                            /* the first transition fixes the bins */
      memset(&d->peakstat, 0, sizeof(d->peakstat));
      const enum rt_mode m = d->opt.mode;
      float range = bitspacing * (m == RT_NRZI ? 1.0f : m == RT_PE ? 1.2f : m == RT_GCR ? 3.0f : m == RT_WW ? 0.75f : 1.0f);
      float bw = range / RT_PEAKSTAT_BUCKETS;
      bw = (float)((int)(bw * 10e6 + 0.5) * 1e-6) / 10.0f;   /* to the nearest 0.1 usec (double product, int, double, float) */
      float left = bitspacing - range / 2;
      left = (float)(int)(left / bw) * bw;                   /* next lower multiple of the bin width */
      d->peakstat.binwidth = bw; d->peakstat.leftbin = left;
      d->peakstat.initialized = 1; }
   const int bucket = (int)((peaktime - d->peakstat.leftbin) / d->peakstat.binwidth);
   if (bucket < 0) ++d->peakstat.counts[trknum][0];
   else if (bucket >= RT_PEAKSTAT_BUCKETS) ++d->peakstat.counts[trknum][RT_PEAKSTAT_BUCKETS - 1];
   else { ++d->peakstat.counts[trknum][bucket]; ++d->peakstat.trksums[trknum]; } }
