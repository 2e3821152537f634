/* TEST INFRASTRUCTURE — the parity oracle.  Not product code; nothing under readtape_amd/ may
 * include, link or call this file (only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do).
 *
 * oracle_fe.c — scalar CPU restatement of the reference's analog front end, one sample instant
 * at a time, exactly as LenShustek/readtape V3.18 does it:
 *
 *   readblock()                       src/readtape.c:1396-1517   (TBIN branch only)
 *   differentiate()                   src/readtape.c:1383-1388
 *   process_sample()                  src/decoder.c:817-905
 *   lookfor_peak() / refine_peak()    src/decoder.c:751-810, 700-749
 *   lookfor_zerocrossing()            src/decoder.c:617-649
 *   lookfor_differentiated_zerocrossing()   src/decoder.c:654-683
 *   init_trackpeak_state()            src/decoder.c:413-423
 *
 * The per-format bit decoders it calls into (rt_up_transition() etc.) are the host library of the
 * product (readtape_amd/csrc/host), which is itself pinned against the reference's .tap output.
 *
 * Pinning: oracle/Makefile builds the unmodified reference into oracle/_ref/ and tests/ compare
 * this restatement with it event for event (tests/golden/ holds the committed vectors), see
 * tests/test_oracle_vs_reference.py.
 */
#include "oracle_fe.h"

#include <stdlib.h>
#include <string.h>

#define DIFFERENTIATE_THRESHOLD 0.05f   /* src/decoder.h:135 */
#define DIFFERENTIATE_SCALE     0.4f    /* src/decoder.h:136 */
#define ZEROCROSS_PEAK          0.2f    /* src/decoder.h:138 */
#define ZEROCROSS_SLOPE         1.5f    /* src/decoder.h:139 */
#define PEAK_THRESHOLD          0.005f  /* src/decoder.h:141 */
#define PKWW_PEAKHEIGHT         4.0f    /* src/decoder.h:133 */

struct ofe *ofe_new(struct rt_dec *dec, const int16_t *rows, int64_t nrows, int nheads,
                    float maxvolts, int64_t tstart_ns) {
   struct ofe *fe = (struct ofe *)calloc(1, sizeof *fe);
   fe->dec = dec;
   fe->rows = rows;
   fe->nrows = nrows;
   fe->nheads = nheads;
   fe->maxvolts = maxvolts;
   fe->tstart_ns = tstart_ns;
   fe->timenow_ns = tstart_ns;
   for (int i = 0; i < RT_MAXTRKS; ++i) fe->head_to_trk[i] = i;
   return fe; }

void ofe_free(struct ofe *fe) { free(fe); }

/* ---- init_trackpeak_state (src/decoder.c:413-423) + the detector-private part of the memset in
 *      init_trackstate (src/decoder.c:437): everything the front end owns restarts per attempt ---- */
static void ofe_reset_detectors(struct ofe *fe) {
   memset(fe->skew, 0, sizeof fe->skew);
   memset(fe->det, 0, sizeof fe->det); }

void ofe_save_pos(void *ctx) {        /* src/readtape.c:1127-1133 */
   struct ofe *fe = (struct ofe *)ctx;
   fe->saved_pos = fe->pos;
   fe->saved_time_ns = fe->timenow_ns;
   fe->saved_time = fe->dec->timenow; }

void ofe_restore_pos(void *ctx) {     /* src/readtape.c:1135-1140 */
   struct ofe *fe = (struct ofe *)ctx;
   fe->pos = fe->saved_pos;
   fe->timenow_ns = fe->saved_time_ns;
   fe->dec->timenow = fe->saved_time; }

/* ---- refine_peak (src/decoder.c:700-749) ---- */
static double refine_peak(struct ofe *fe, struct rt_trk *t, struct ofe_det *w, float val, int top) {
   int pkww_width = fe->pkww_width;
   int left_distance = 1;
   int ndx, nextndx, prevndx = -1;
   float time_adjustment = 0;
   for (ndx = w->pkww_left; ;) {
      if (w->pkww_v[ndx] == val) {
         if (!(left_distance < pkww_width) || prevndx == -1) { fe->fatal = 1; return 0; }   /* the reference asserts and exits */
         nextndx = ndx + 1; if (nextndx >= pkww_width) nextndx = 0;
         if (top) {
            float val_minus = val - PEAK_THRESHOLD / t->agc_gain;
            if (w->pkww_v[prevndx] > val_minus && w->pkww_v[nextndx] < val_minus) time_adjustment = -0.5;
            else if (w->pkww_v[nextndx] > val_minus && w->pkww_v[prevndx] < val_minus) time_adjustment = +0.5; }
         else {
            float val_plus = val + PEAK_THRESHOLD / t->agc_gain;
            if (w->pkww_v[prevndx] < val_plus && w->pkww_v[nextndx] > val_plus) time_adjustment = -0.5;
            else if (w->pkww_v[nextndx] < val_plus && w->pkww_v[prevndx] > val_plus) time_adjustment = +0.5; }
         double time = fe->dec->timenow - ((float)(pkww_width - left_distance) - time_adjustment) * fe->dec->sample_deltat;
         w->pkww_countdown = left_distance;
         w->last_left_distance = left_distance;
         w->last_adj2 = (int)(time_adjustment * 2);
         return time; }
      ++left_distance;
      if (ndx == w->pkww_right) break;
      prevndx = ndx;
      if (++ndx >= pkww_width) ndx = 0; }
   fe->fatal = 1;
   return 0; }

/* ---- lookfor_peak (src/decoder.c:751-810), including the never-true min update at :765 ---- */
static void lookfor_peak(struct ofe *fe, struct rt_trk *t, struct ofe_det *w) {
   struct rt_dec *d = fe->dec;
   int pkww_width = fe->pkww_width;
   float old_left = 0;
   if (++w->pkww_right >= pkww_width) w->pkww_right = 0;
   if (w->pkww_right == w->pkww_left) {
      old_left = w->pkww_v[w->pkww_left];
      if (++w->pkww_left >= pkww_width) w->pkww_left = 0; }
   w->pkww_v[w->pkww_right] = t->v_now;
   if (t->v_now > w->pkww_maxv) w->pkww_maxv = t->v_now;
   /* else if (minv < minv) minv = v_now;   -- dead code in the reference: the min only changes on a rescan */
   if (old_left == w->pkww_maxv || old_left == w->pkww_minv) {
      float maxv = -100, minv = +100;
      for (int ndx = w->pkww_left; ; ) {
         maxv = maxv > w->pkww_v[ndx] ? maxv : w->pkww_v[ndx];
         minv = minv < w->pkww_v[ndx] ? minv : w->pkww_v[ndx];
         if (ndx == w->pkww_right) break;
         if (++ndx >= pkww_width) ndx = 0; }
      w->pkww_maxv = maxv;
      w->pkww_minv = minv; }
   if (w->pkww_countdown) {
      --w->pkww_countdown; }
   else {
      if (!(t->agc_gain > 0)) { fe->fatal = 1; return; }              /* "AGC gain bad in lookfor_peak" (src/decoder.c:782): the reference asserts and exits */
      float required_rise = RT_PARM(d).pkww_rise * (t->v_avg_height / (float)PKWW_PEAKHEIGHT) / t->agc_gain;
      float required_min = RT_PARM(d).min_peak * (t->v_avg_height / (float)PKWW_PEAKHEIGHT) / t->agc_gain;
      if (w->pkww_maxv > w->pkww_v[w->pkww_left] + required_rise
            && w->pkww_maxv > w->pkww_v[w->pkww_right] + required_rise
            && (required_min == 0 || w->pkww_maxv > required_min)) {
         t->v_top = w->pkww_maxv;
         t->t_top = refine_peak(fe, t, w, w->pkww_maxv, 1);
         if (fe->fatal) return;
         rt_up_transition(d, t); }
      else if (w->pkww_minv < w->pkww_v[w->pkww_left] - required_rise
               && w->pkww_minv < w->pkww_v[w->pkww_right] - required_rise
               && (required_min == 0 || w->pkww_minv < -required_min)) {
         t->v_bot = w->pkww_minv;
         t->t_bot = refine_peak(fe, t, w, w->pkww_minv, 0);
         if (fe->fatal) return;
         rt_down_transition(d, t); } } }

/* ---- lookfor_zerocrossing (src/decoder.c:617-649) ---- */
static void lookfor_zerocrossing(struct ofe *fe, struct rt_trk *t, struct ofe_det *w) {
   struct rt_dec *d = fe->dec;
   double timenow = d->timenow;
   if (t->v_now > 0) {
      w->zerocross_dn_pending = 0;
      if (t->v_top < t->v_now) {
         t->v_top = t->v_now;
         if (w->zerocross_up_pending && t->v_top > ZEROCROSS_PEAK) {
            if (t->t_top == 0) t->t_top = timenow;
            w->zerocross_up_pending = 0;
            t->v_bot = 0;
            if (timenow - t->t_top <= t->clkavg.t_bitspaceavg * ZEROCROSS_SLOPE)
               rt_up_transition(d, t); } }
      if (w->v_prev < 0 && t->v_bot < -ZEROCROSS_PEAK) {
         t->t_top = timenow;
         w->zerocross_up_pending = 1; } }
   else if (t->v_now < 0) {
      w->zerocross_up_pending = 0;
      if (t->v_bot > t->v_now) {
         t->v_bot = t->v_now;
         if (w->zerocross_dn_pending && t->v_bot < -ZEROCROSS_PEAK) {
            if (t->t_bot == 0) t->t_bot = timenow;
            w->zerocross_dn_pending = 0;
            t->v_top = 0;
            if (timenow - t->t_bot <= t->clkavg.t_bitspaceavg * ZEROCROSS_SLOPE)
               rt_down_transition(d, t); } }
      if (w->v_prev > 0 && t->v_top > ZEROCROSS_PEAK) {
         t->t_bot = timenow;
         w->zerocross_dn_pending = 1; } }
   w->v_prev = t->v_now; }

/* ---- lookfor_differentiated_zerocrossing (src/decoder.c:654-683) ---- */
static void lookfor_differentiated_zerocrossing(struct ofe *fe, struct rt_trk *t, struct ofe_det *w) {
   struct rt_dec *d = fe->dec;
   double timenow = d->timenow;
   if (t->v_now > 0) {
      if (t->v_top < t->v_now) t->v_top = t->v_now;
      if (w->zerocross_up_pending) {
         t->t_top = w->t_firstzero > 0 ? (w->t_firstzero + w->t_lastzero) / 2 : timenow - d->sample_deltat / 2;
         w->zerocross_up_pending = 0;
         w->t_firstzero = 0;
         rt_up_transition(d, t); }
      if (t->v_now > ZEROCROSS_PEAK) {
         w->zerocross_dn_pending = 1;
         w->t_firstzero = 0;
         t->v_bot = 0; } }
   else if (t->v_now < 0) {
      if (t->v_bot > t->v_now) t->v_bot = t->v_now;
      if (w->zerocross_dn_pending) {
         t->t_bot = w->t_firstzero > 0 ? (w->t_firstzero + w->t_lastzero) / 2 : timenow - d->sample_deltat / 2;
         w->zerocross_dn_pending = 0;
         w->t_firstzero = 0;
         rt_down_transition(d, t); }
      if (t->v_now < -ZEROCROSS_PEAK) {
         w->zerocross_up_pending = 1;
         w->t_firstzero = 0;
         t->v_top = 0; } }
   else {
      w->t_lastzero = timenow;
      if (w->t_firstzero == 0) w->t_firstzero = timenow; } }

/* ---- process_sample (src/decoder.c:817-905) ---- */
static enum rt_bstate process_sample(struct ofe *fe, const float *voltage) {
   struct rt_dec *d = fe->dec;
   int ntrks = d->opt.ntrks;
   for (int trknum = 0; trknum < ntrks; ++trknum) {          /* deskew FIFO, :820-830 */
      struct rt_trk *t = &d->trk[trknum];
      int delay = fe->skew_delaycnt[trknum];
      if (delay == 0) t->v_now = voltage[trknum];
      else {
         struct ofe_skew *s = &fe->skew[trknum];
         if (s->slots_filled < delay) {
            t->v_now = voltage[trknum];
            ++s->slots_filled; }
         else t->v_now = s->vdelayed[s->ndx_next];
         s->vdelayed[s->ndx_next] = voltage[trknum];
         if (++s->ndx_next >= delay) s->ndx_next = 0; } }

   if (d->interblock_counter) goto exit;

   if (d->opt.mode == RT_NRZI && rt_nrzi_zerocheck_due(d)) rt_nrzi_zerocheck(d);

   for (int trknum = 0; trknum < ntrks; ++trknum) {
      struct rt_trk *t = &d->trk[trknum];
      struct ofe_det *w = &fe->det[trknum];
      if (t->t_lastpeak == 0) {                              /* first sample of this track, :855-861 */
         w->pkww_v[0] = t->v_now;
         w->pkww_maxv = w->pkww_minv = t->v_now;
         t->v_lastpeak = t->v_now;
         t->t_lastpeak = d->timenow;
         break; }
      if (d->opt.find_zeros) {
         if (d->opt.do_differentiate) lookfor_differentiated_zerocrossing(fe, t, w);
         else lookfor_zerocrossing(fe, t, w); }
      else lookfor_peak(fe, t, w);
      if (fe->fatal || d->fatal) { d->fatal = 1; d->results[d->parmset].blktype = RT_BS_ABORTED; return RT_BS_ABORTED; }      /* (d->fatal: an assert inside the block decoder's callback, e.g. src/decode_nrzi.c:227) */

      if (d->opt.mode == RT_PE && rt_pe_idle_due(d, t)) rt_pe_go_idle(d, t);
      if (d->opt.mode == RT_GCR && rt_gcr_idle_due(d, t))
         if (rt_gcr_go_idle(d, t)) goto exit; }

   if (d->opt.mode == RT_WW && rt_ww_end_due(d)) rt_ww_end_of_block(d);      /* the clock has stopped (src/decoder.c:892-894) */

exit:
   if (fe->eob_row < 0 && d->results[d->parmset].blktype != RT_BS_NONE) fe->eob_row = fe->pos - 1;   /* row at which the block ended */
   if (d->interblock_counter) {
      if (--d->interblock_counter) return RT_BS_NONE; }
   return d->results[d->parmset].blktype; }

/* ---- readblock, TBIN branch (src/readtape.c:1396-1517) ---- */
int ofe_readblock(void *ctx, int retry) {
   struct ofe *fe = (struct ofe *)ctx;
   struct rt_dec *d = fe->dec;
   float voltage[RT_MAXTRKS];
   int did_processing = 0, endfile = 0;
   enum rt_bstate blockkind = RT_BS_NONE;
   int samples_per_bit = rt_samples_per_bit(d);
   fe->eob_row = -1;
   if (d->opt.mode != RT_WW) ofe_reset_detectors(fe);     /* what init_trackstate does to detector state, src/decoder.c:432,437 (Whirlwind: once per tape, src/readtape.c:1674) */
   if (fe->on_attempt_start) fe->on_attempt_start(fe, fe->pos);
   do {
      if (!retry) ++fe->lines_in;
      if (fe->pos >= fe->nrows) {                               /* 0x8000 end marker, :1410-1413 */
         if (did_processing) rt_force_end_of_block(d);
         endfile = 1;
         goto done; }
      const int16_t *row = fe->rows + fe->pos * fe->nheads;
      ++fe->pos;
      for (int head = 0; head < fe->nheads; ++head) {
         int trk = fe->head_to_trk[head];
         voltage[trk] = (float)row[head] / 32767 * fe->maxvolts;      /* :1420 */
         if (fe->invert) voltage[trk] = -voltage[trk];
         if (d->opt.do_differentiate) {                             /* differentiate(), :1383-1388 */
            float v = voltage[trk];
            float delta = v - fe->det[trk].v_last_raw;
            if (delta < DIFFERENTIATE_THRESHOLD && delta > -DIFFERENTIATE_THRESHOLD) delta = 0;
            fe->det[trk].v_last_raw = v;
            voltage[trk] = delta * DIFFERENTIATE_SCALE * samples_per_bit; } }
      double sample_time = (double)fe->timenow_ns / 1e9;           /* :1423 */
      fe->timenow_ns += d->sample_deltat_ns;
      ++fe->numsamples;
      d->timenow = sample_time;
      if (!d->window_set) {                                        /* :1453-1457 */
         fe->pkww_width = rt_pkww_width(d, d->parmset);
         d->window_set = 1; }
      did_processing = 1;
      blockkind = process_sample(fe, voltage); }
   while (blockkind == RT_BS_NONE);
done:
   rt_finish_attempt(d);
   if (fe->on_attempt_end) fe->on_attempt_end(fe);
   return !endfile; }
