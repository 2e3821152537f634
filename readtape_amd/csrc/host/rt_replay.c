/* rt_replay.c — drives the host block decoders from the device front end's event lists.
 *
 * This is the counterpart of the reference's readblock()/process_sample() control flow
 * (src/readtape.c:1396-1517, src/decoder.c:817-905) with the per-sample detectors removed: the
 * detectors ran on the GPU (include/rt_frontend.h) and left, per block attempt and parameter set, the
 * exact sequence of top/bottom detections.  What remains sequential — and is reproduced here row for
 * row, but only at rows where something happens — is:
 *   - the staggered track start (t_lastpeak = time of row s0+trk, src/decoder.c:855-861),
 *   - the NRZI mid-bit timer (src/decoder.c:844-845), the PE and GCR idle timers (:868-888),
 *   - the interblock skip (src/decoder.c:841-842, 901-904),
 *   - end of data (src/readtape.c:1410-1413),
 * in exactly the order process_sample() interleaves them with the detections of a row.
 *
 * Speculation check: an attempt that starts at row s0 may use burst b only if
 * zone_first <= s0 <= safe_last and the burst is not flagged; it may run past the burst's end only
 * while it has not seen a single event (fresh state == the next burst's fresh state).  Otherwise the
 * `exact` callback is asked for an exact device scan of this one attempt.  No detector code runs here.
 */
#include "rt_replay.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static double time_of_row(const struct rt_replay *rp, int64_t row) {        /* src/readtape.c:1423 */
   return (double)(rp->tstart_ns + row * rp->tdelta_ns) / 1e9; }

/* smallest row >= from with pred(time_of_row(row)) true, for a predicate that is monotone in time;
 * `guess_time` is where it is expected to flip.  Returns limit if not before limit. */
typedef int (*time_pred)(const struct rt_dec *d, const struct rt_trk *t, double timenow);
static int64_t first_row_where(const struct rt_replay *rp, time_pred pred, const struct rt_trk *t,
                               double guess_time, int64_t from, int64_t limit) {
   double g = (guess_time * 1e9 - (double)rp->tstart_ns) / (double)rp->tdelta_ns;
   int64_t r = (g < (double)from) ? from : (g > (double)limit ? limit : (int64_t)g);
   if (r < from) r = from;
   if (r > limit) r = limit;
   struct rt_dec *d = rp->d;
   /* walk down while the predicate still holds just below, then up until it holds */
   while (r > from && pred(d, t, time_of_row(rp, r - 1))) --r;
   while (r < limit && !pred(d, t, time_of_row(rp, r))) ++r;
   return r; }

static int pred_nrzi(const struct rt_dec *d, const struct rt_trk *t, double timenow) {
   (void)t;
   return timenow > d->nrzi.t_lastclock + 2 * d->nrzi.clkavg.t_bitspaceavg; }     /* src/decoder.c:844 */
static int pred_pe(const struct rt_dec *d, const struct rt_trk *t, double timenow) {
   (void)d;
   return timenow - t->t_lastpeak > t->clkavg.t_bitspaceavg * 2.5f; }            /* src/decoder.c:868 */
static int pred_gcr(const struct rt_dec *d, const struct rt_trk *t, double timenow) {
   (void)d;
   return timenow > t->t_lastpeak + 6.00 * t->clkavg.t_bitspaceavg; }            /* src/decoder.c:880 */

/* ---- the event source of one attempt: per-track lists of one (burst, parmset) ---- */
struct evsrc {
   const rtfe_event *list[RT_MAXTRKS];
   uint32_t n[RT_MAXTRKS], at[RT_MAXTRKS];
   int64_t nr[RT_MAXTRKS];  /* absolute row of each track's next event, INT64_MAX when its list is used up (evsrc_sync keeps it) */
   int64_t reset;           /* absolute row of sample 0 */
   int64_t end;             /* rows >= end are not covered by this source */
};

static inline void evsrc_sync(struct evsrc *s, int t) {
   s->nr[t] = s->at[t] < s->n[t] ? s->reset + (int64_t)s->list[t][s->at[t]].sample : INT64_MAX; }

static void evsrc_from_burst(struct evsrc *s, const struct rt_replay *rp, int64_t b, int parmset) {
   const rtfe_burst *B = &rp->bursts[b];
   if (rp->find_zeros) parmset = 0;           /* the zero-crossing front end does not depend on the parameter set */
   for (int t = 0; t < rp->ntrks; ++t) {
      s->list[t] = rp->events + B->event_base + (uint64_t)(parmset * rp->ntrks + t) * B->event_cap;
      s->n[t] = rp->counts[((size_t)b * rp->nparm + parmset) * rp->ntrks + t];
      s->at[t] = 0; }
   s->reset = B->reset_sample - rp->row_base;
   s->end = B->end_sample - rp->row_base;
   for (int t = 0; t < rp->ntrks; ++t) evsrc_sync(s, t); }

static inline int64_t evsrc_next_row(const struct evsrc *s, int ntrks) {
   int64_t best = INT64_MAX;
   for (int t = 0; t < ntrks; ++t) if (s->nr[t] < best) best = s->nr[t];
   return best; }

static void evsrc_skip_before(struct evsrc *s, int ntrks, int64_t row) {
   for (int t = 0; t < ntrks; ++t)
      if (s->nr[t] < row) {
         while (s->at[t] < s->n[t] && s->reset + s->list[t][s->at[t]].sample < row) ++s->at[t];
         evsrc_sync(s, t); } }

static int64_t find_burst(const struct rt_replay *rp, int64_t s0) {
   /* last burst whose zone starts at or before s0 */
   int64_t lo = 0, hi = rp->nbursts - 1, ans = -1;
   const int64_t a = s0 + rp->row_base;
   while (lo <= hi) {
      int64_t mid = (lo + hi) / 2;
      if (rp->bursts[mid].zone_first <= a) { ans = mid; lo = mid + 1; } else hi = mid - 1; }
   if (ans < 0) return -1;
   const rtfe_burst *B = &rp->bursts[ans];
   if ((B->flags & (RTFE_F_UNSAFE | RTFE_F_EVENT_OVERFLOW | RTFE_F_SCREEN_UNDERFLOW | RTFE_F_DETECTOR_FATAL)) || a > B->safe_last) return -1;
   return ans; }

static int burst_usable(const struct rt_replay *rp, int64_t b) {
   if (b < 0 || b >= rp->nbursts) return 0;
   return !(rp->bursts[b].flags & (RTFE_F_UNSAFE | RTFE_F_EVENT_OVERFLOW | RTFE_F_SCREEN_UNDERFLOW | RTFE_F_DETECTOR_FATAL)); }

void rt_replay_save_pos(void *ctx) {
   struct rt_replay *rp = (struct rt_replay *)ctx;
   rp->saved_pos = rp->pos; rp->saved_time = rp->d->timenow; }

void rt_replay_restore_pos(void *ctx) {
   struct rt_replay *rp = (struct rt_replay *)ctx;
   rp->pos = rp->saved_pos; rp->d->timenow = rp->saved_time; }

/* one transition, exactly as lookfor_peak/refine_peak hand it to process_*_transition */
static void deliver(struct rt_replay *rp, const struct evsrc *s, int trk, const rtfe_event *e, int W) {
   struct rt_dec *d = rp->d;
   struct rt_trk *t = &d->trk[trk];
   if (rp->find_zeros && d->opt.do_differentiate) {
      /* differentiated signal (src/decoder.c:657-663, 670-676): the crossing is the centre of the run of exact
       * zeros if there was one, else half a sample before the confirming row; no slope gate */
      uint32_t dd; memcpy(&dd, &e->agc_gain, 4);
      const int64_t row = s->reset + (int64_t)e->sample;
      const uint32_t d1 = dd >> 16, d2 = dd & 0xffff;
      const double tz = d1 ? ((double)(rp->tstart_ns + (row - d1) * rp->tdelta_ns) / 1e9 + (double)(rp->tstart_ns + (row - d2) * rp->tdelta_ns) / 1e9) / 2
                           : d->timenow - d->sample_deltat / 2;
      if (e->flags & 1) { t->v_bot = e->v_peak; t->t_bot = tz; rt_down_transition(d, t); }
      else { t->v_top = e->v_peak; t->t_top = tz; rt_up_transition(d, t); }
      ++rp->events_delivered;
      return; }
   if (rp->find_zeros) {
      /* a confirmed zero crossing (src/decoder.c:625-630, 639-644): the extreme is new, the opposite excursion
       * restarts, and the transition counts only if the excursion was reached soon enough after the crossing */
      uint32_t delay; memcpy(&delay, &e->agc_gain, 4);
      const int64_t cross_row = s->reset + (int64_t)e->sample - (int64_t)delay;
      const double t_cross = (double)(rp->tstart_ns + cross_row * rp->tdelta_ns) / 1e9;
      if (e->flags & 1) {
         t->v_bot = e->v_peak; t->t_bot = t_cross; t->v_top = 0;
         if (d->timenow - t->t_bot <= t->clkavg.t_bitspaceavg * 1.5f) { rt_down_transition(d, t); ++rp->events_delivered; } }
      else {
         t->v_top = e->v_peak; t->t_top = t_cross; t->v_bot = 0;
         if (d->timenow - t->t_top <= t->clkavg.t_bitspaceavg * 1.5f) { rt_up_transition(d, t); ++rp->events_delivered; } }
      return; }
   const int adjc = (e->flags >> 1) & 3;
   const float adj = adjc == 1 ? -0.5f : (adjc == 2 ? 0.5f : 0.0f);
   const double tp = d->timenow - ((float)(W - e->left_distance) - adj) * d->sample_deltat;   /* src/decoder.c:732 */
   uint32_t ga, gb;
   memcpy(&ga, &t->agc_gain, 4); memcpy(&gb, &e->agc_gain, 4);
   if (ga != gb) ++rp->agc_mismatches;        /* the device's AGC mirror and the decoder's own AGC must agree bit for bit */
   if (e->flags & 1) { t->v_bot = e->v_peak; t->t_bot = tp; rt_down_transition(d, t); }
   else { t->v_top = e->v_peak; t->t_top = tp; rt_up_transition(d, t); }
   ++rp->events_delivered; }

/* optional dump of what the decoders are handed, in the record format of oracle/ref_event_shim.c */
struct dump_rec { uint32_t kind, trk; int32_t peakcount, parmset; double t_peak; int64_t timenow_ns; float v_peak, agc_gain, v_avg_height; uint32_t pad; };
static void dump_transition(struct rt_dec *d, struct rt_trk *t, int is_top, void *user) {
   struct rt_replay *rp = (struct rt_replay *)user;
   struct dump_rec r; memset(&r, 0, sizeof r);
   r.kind = is_top ? 0 : 1; r.trk = (uint32_t)t->trknum; r.peakcount = t->peakcount; r.parmset = d->parmset;
   r.t_peak = is_top ? t->t_top : t->t_bot;
   r.timenow_ns = rp->tstart_ns + (rp->cur_row + 1) * rp->tdelta_ns;          /* the reference has already advanced it, src/readtape.c:1424 */
   r.v_peak = is_top ? t->v_top : t->v_bot; r.agc_gain = t->agc_gain; r.v_avg_height = t->v_avg_height;
   fwrite(&r, sizeof r, 1, rp->evtf); }
static void dump_attempt(struct rt_dec *d, void *user) {
   struct rt_replay *rp = (struct rt_replay *)user;
   struct dump_rec r; memset(&r, 0, sizeof r);
   r.kind = 2; r.parmset = d->parmset; r.timenow_ns = rp->tstart_ns + rp->pos * rp->tdelta_ns;
   fwrite(&r, sizeof r, 1, rp->evtf); }

static int readblock_once(void *ctx, int retry) {
   struct rt_replay *rp = (struct rt_replay *)ctx;
   struct rt_dec *d = rp->d;
   const int ntrks = rp->ntrks, parmset = d->parmset;
   const int W = rp->W[parmset];
   const int64_t s0 = rp->pos;
   const int64_t nrows = rp->nrows;
   int endfile = 0;
   (void)retry;
   ++rp->attempts;
   if (s0 >= nrows) { rt_finish_attempt(d); return 0; }     /* no row was processed: no forced end of block */
   if (s0 >= rp->stop_row) { rt_finish_attempt(d); return 0; }   /* fragment decode: the next attempt starts in a zone the next fragment owns */

   struct evsrc src;
   const long dump_pos = rp->evtf ? ftell(rp->evtf) : 0;      /* a restarted attempt rewinds the optional dump */
   /* ... and what a pre-pass accumulates ACROSS attempts (the -deskew peak statistics, the density histogram): the reference reads
    * the block once, a restarted attempt must not count its transitions twice (found by tests/stress_gpu.py, seed 810 tape 100:
    * the pre-pass had its 1000 transitions per track one block early) */
   const int prepass_stats = d->doing_deskew || d->doing_density_detection;
   __typeof__(d->peakstat) saved_peakstat; __typeof__(d->estden) saved_estden;
   if (prepass_stats) { saved_peakstat = d->peakstat; saved_estden = d->estden; }
   int64_t b = find_burst(rp, s0);
   rtfe_event *exact_events = NULL;
   int using_exact = 0;
   int restarted = 0;
   /* an exact scan covers this attempt only: up to the end of the device burst after the one s0 lies in,
    * extended (x4) in the rare case the attempt runs longer */
   int64_t exact_len = 1 << 16;
   {  /* first burst whose zone starts behind s0 (the table is ordered: binary search) */
      int64_t lo = 0, hi = rp->nbursts;
      while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (rp->bursts[mid].zone_first - rp->row_base > s0) hi = mid; else lo = mid + 1; }
      if (lo < rp->nbursts) exact_len = rp->bursts[lo].end_sample - rp->row_base - s0; }
   if (exact_len < (1 << 12)) exact_len = 1 << 12;
restart:
   if (rp->evtf && (restarted || using_exact)) fseek(rp->evtf, dump_pos, SEEK_SET);
   if (prepass_stats && (restarted || using_exact)) { d->peakstat = saved_peakstat; d->estden = saved_estden; }
   if (b < 0) {                                               /* outside every proven-safe zone: exact device scan */
      if (!rp->exact) { ++rp->device_failures; d->results[parmset].blktype = RT_BS_ABORTED; rt_finish_attempt(d); return 0; }
      uint32_t cnt[RT_MAXTRKS]; rtfe_burst eb; uint32_t cap = 0;
      if (exact_events && rp->exact_free) { rp->exact_free(rp->exact_user, exact_events); exact_events = NULL; }
      const int64_t ex_end = s0 + exact_len < nrows ? s0 + exact_len : nrows;
      if (rp->exact(rp->exact_user, s0, ex_end, rp->find_zeros ? 0 : parmset, &eb, cnt, &exact_events, &cap) != 0) {
         ++rp->device_failures; d->results[parmset].blktype = RT_BS_ABORTED; rt_finish_attempt(d); return 0; }
      for (int t = 0; t < ntrks; ++t) { src.list[t] = exact_events + (uint64_t)t * cap; src.n[t] = cnt[t]; src.at[t] = 0; }
      src.reset = s0; src.end = ex_end;
      for (int t = 0; t < ntrks; ++t) evsrc_sync(&src, t);
      using_exact = 1; ++rp->exact_scans; }
   else { evsrc_from_burst(&src, rp, b, parmset); evsrc_skip_before(&src, ntrks, s0); }

   int64_t events_seen = 0;
   int64_t row = s0;                 /* next row to "process" */
   int ntrk_started = 0;             /* tracks whose staggered start row has passed */
   for (;;) {
      /* ---- the next row at which anything can happen ---- */
      int64_t next = nrows;
      if (d->interblock_counter) {                          /* rows are skipped; the attempt returns when it reaches 0 */
         next = row + d->interblock_counter - 1; if (next > nrows) next = nrows; }
      else {
         if (ntrk_started < ntrks) { int64_t r = s0 + ntrk_started; if (r < next) next = r; }
         int64_t r = evsrc_next_row(&src, ntrks); if (r < next) next = r;
         if (d->opt.mode == RT_NRZI && d->nrzi.datablock && next > row && pred_nrzi(d, NULL, time_of_row(rp, next - 1))) {
            /* (the mid-bit timer is monotone in time: it can only run out before `next` if it has at row next - 1) */
            r = first_row_where(rp, pred_nrzi, NULL, d->nrzi.t_lastclock + 2 * d->nrzi.clkavg.t_bitspaceavg, row, next);
            if (r < next) next = r; }
         if (d->opt.mode == RT_PE)
            for (int t = 0; t < ntrks; ++t) {
               struct rt_trk *tk = &d->trk[t];
               if (!tk->idle && tk->t_lastpeak != 0) {
                  r = first_row_where(rp, pred_pe, tk, tk->t_lastpeak + tk->clkavg.t_bitspaceavg * 2.5f, row, next);
                  if (r < next) next = r; } }
         if (d->opt.mode == RT_GCR)
            for (int t = 0; t < ntrks; ++t) {
               struct rt_trk *tk = &d->trk[t];
               if (tk->datablock) {
                  r = first_row_where(rp, pred_gcr, tk, tk->t_lastpeak + 6.00 * tk->clkavg.t_bitspaceavg, row, next);
                  if (r < next) next = r; } } }
      /* the event source ends before anything else happens? */
      if (using_exact && !d->interblock_counter && next >= src.end && src.end < nrows) {
         exact_len *= 4;                                       /* the attempt is longer than the exact scan: rescan further */
         { void (*oa)(struct rt_dec *, void *) = d->on_attempt; d->on_attempt = NULL; rt_init_trackstate(d); d->on_attempt = oa; }
         d->interblock_counter = 0;
         goto restart; }
      if (!using_exact && !d->interblock_counter && next >= src.end && src.end < nrows) {
         /* still fresh: the next burst's fresh state is the same state.  During density detection the detector never
          * leaves its fresh AGC / baseline state (no decoder runs), so every proven restart is the same state */
         const int fresh = (events_seen == 0 || d->doing_density_detection)
                           && !(rp->find_zeros && (rp->bursts[b].flags & RTFE_F_STATE_AT_END));     /* (-zeros: an excursion without an event is history too) */
         if (fresh && rp->stop_row != INT64_MAX
             && (b + 1 >= rp->nbursts || rp->bursts[b + 1].zone_first - rp->row_base >= rp->stop_row)) {
            /* fragment decode: the burst behind this one belongs to the next fragment (it is the bounding entry of the table, or not in
             * it at all), whose replay starts from the same fresh state at the start of that burst's zone - this attempt saw nothing
             * and ends here without a block.  (chaining on, or redoing the attempt from an exact scan, decoded the
             * neighbour's block a second time.) */
            rp->pos = rp->stop_row;
            rt_finish_attempt(d);
            return 0; }
         if (fresh && burst_usable(rp, b + 1)) {
            ++b; evsrc_from_burst(&src, rp, b, parmset); evsrc_skip_before(&src, ntrks, row); ++rp->chained;
            continue; }
         if (restarted) { ++rp->device_failures; d->results[parmset].blktype = RT_BS_ABORTED; break; }
         /* this attempt crosses a device restart with state: redo it from s0 with an exact scan */
         restarted = 1; b = -1;
         { void (*oa)(struct rt_dec *, void *) = d->on_attempt; d->on_attempt = NULL; rt_init_trackstate(d); d->on_attempt = oa; }
         d->interblock_counter = 0;
         goto restart; }
      if (next >= nrows) {                                   /* end of data, src/readtape.c:1410-1413 */
         d->timenow = time_of_row(rp, nrows - 1);
         rt_force_end_of_block(d);
         rp->pos = nrows;
         endfile = 1;
         break; }
      row = next;
      rp->cur_row = row;
      d->timenow = time_of_row(rp, row);
      /* ---- process_sample for this row (src/decoder.c:841-904) ---- */
      if (d->interblock_counter) {                          /* we jumped to the row on which the countdown ends */
         d->interblock_counter = 0;
         rp->pos = row + 1;
         break; }
      if (d->opt.mode == RT_NRZI && rt_nrzi_zerocheck_due(d)) rt_nrzi_zerocheck(d);
      int stop_row = 0;
      for (int t = 0; t < ntrks && !stop_row; ++t) {
         struct rt_trk *tk = &d->trk[t];
         if (tk->t_lastpeak == 0) {                          /* first row of this track, src/decoder.c:855-861 */
            tk->v_lastpeak = 0;
            tk->t_lastpeak = d->timenow;
            if (t >= ntrk_started) ntrk_started = t + 1;
            break; }
         while (src.nr[t] == row) {
            if (src.list[t][src.at[t]].flags & RTFE_EV_FATAL) {    /* "AGC gain bad in lookfor_peak" (src/decoder.c:782): the reference exits here */
               rp->reference_fatal = 1; d->fatal = 1; rp->fatal_row = row; rp->fatal_trk = t;
               d->results[parmset].blktype = RT_BS_ABORTED;
               if (exact_events && rp->exact_free) rp->exact_free(rp->exact_user, exact_events);
               rt_finish_attempt(d);
               return 0; }
            deliver(rp, &src, t, &src.list[t][src.at[t]], W);
            if (d->fatal) {                                      /* the decoder's callback met one of the reference's asserts (a learned peak height that is not positive): the run ends inside it */
               rp->reference_fatal = 1; rp->fatal_row = row; rp->fatal_trk = t;
               d->results[parmset].blktype = RT_BS_ABORTED;
               if (exact_events && rp->exact_free) rp->exact_free(rp->exact_user, exact_events);
               rt_finish_attempt(d);
               return 0; }
            ++src.at[t]; ++events_seen; evsrc_sync(&src, t); }
         if (d->opt.mode == RT_PE && rt_pe_idle_due(d, tk)) rt_pe_go_idle(d, tk);
         if (d->opt.mode == RT_GCR && rt_gcr_idle_due(d, tk)) if (rt_gcr_go_idle(d, tk)) stop_row = 1; }
      /* events of tracks the reference did not reach on this row cannot exist; drop anything stale */
      evsrc_skip_before(&src, ntrks, row + 1);
      /* exit: (src/decoder.c:900-904) */
      if (d->interblock_counter) {
         if (--d->interblock_counter) { ++row; continue; }
         rp->pos = row + 1; break; }
      if (d->results[parmset].blktype != RT_BS_NONE) { rp->pos = row + 1; break; }
      ++row; }
   if (exact_events && rp->exact_free) rp->exact_free(rp->exact_user, exact_events);
   rt_finish_attempt(d);
   return !endfile; }

int rt_replay_readblock(void *ctx, int retry) {
   struct rt_replay *rp = (struct rt_replay *)ctx;
   const int64_t from = rp->pos;
   const int more = readblock_once(ctx, retry);
   /* src/readtape.c:1404 counts every pass of the read loop of a first attempt, the one that finds the end marker included */
   if (!retry) rp->d->lines_in += (rp->pos - from) + (more ? 0 : 1);
   return more; }

/* prepass != NULL: run the -deskew pre-pass instead of the decode (no .tap); append: keep what is already in the log
 * and event-dump files (the decode that follows a pre-pass continues both, as the reference's single run does) */
struct deskew_out { int *delays; int *nblks; int *hit_end; float *bpi, *implied; };
static int replay_any(const struct rt_options *opt, const struct rt_parms *parmsets, int nparm,
                  int64_t tdelta_ns, int64_t tstart_ns, int64_t nrows, int64_t row_base, const int *W,
                  const rtfe_burst *bursts, int64_t nbursts, const uint32_t *counts, const rtfe_event *events,
                  rt_exact_fn exact, rt_exact_free_fn exact_free, void *user,
                  const char *tap_path, const char *log_path, const char *evt_path, struct rt_replay_stats *stats,
                  int append, struct deskew_out *prepass, int64_t start_row, int64_t stop_row, int fragment,
                  const char *out_base, const char *in_name) {
   const float sample_deltat = (float)tdelta_ns / 1e9f;              /* src/readtape.c:1345 */
   struct rt_dec *d = rt_dec_new(opt, sample_deltat, tdelta_ns);
   if (!d) return -1;
   if (parmsets) {
      memset(d->parmsets, 0, sizeof d->parmsets);
      memcpy(d->parmsets, parmsets, sizeof(struct rt_parms) * (size_t)nparm); }
   else for (int i = nparm; i < RT_MAXPARMSETS; ++i) d->parmsets[i].active = 0;   /* only the scanned sets are usable */
   if (out_base) snprintf(d->outbase, sizeof d->outbase, "%s", out_base);      /* files by the reference's names, made when first needed */
   else if (tap_path) d->tapf = fopen(tap_path, "wb");
   if (log_path) d->logf = fopen(log_path, append ? "a" : "w");
   const double wall0 = (double)time(NULL);
   struct rt_replay rp; memset(&rp, 0, sizeof rp);
   rp.d = d; rp.ntrks = opt->ntrks; rp.nparm = nparm;
   for (int i = 0; i < nparm; ++i) rp.W[i] = W[i];
   rp.nrows = nrows; rp.row_base = row_base; rp.tstart_ns = tstart_ns; rp.tdelta_ns = tdelta_ns;
   rp.bursts = bursts; rp.nbursts = nbursts; rp.counts = counts; rp.events = events;
   rp.exact = exact; rp.exact_free = exact_free; rp.exact_user = user;
   rp.find_zeros = opt->find_zeros;
   rp.pos = start_row; rp.stop_row = stop_row;
   d->no_tap_end = fragment;                      /* a fragment's .tap is concatenated with its neighbours': the caller ends the file */
   if (evt_path) {
      /* (not "ab": a restarted attempt rewinds the dump, and O_APPEND ignores seeks) */
      rp.evtf = append ? fopen(evt_path, "r+b") : fopen(evt_path, "wb");
      if (rp.evtf && append) fseek(rp.evtf, 0, SEEK_END);
      if (rp.evtf) { d->on_transition = dump_transition; d->on_attempt = dump_attempt; d->user = &rp; } }
   struct rt_reader rd = { rt_replay_readblock, rt_replay_save_pos, rt_replay_restore_pos, &rp };
   int ok = 1;
   if (prepass && prepass->bpi) *prepass->bpi = rt_density_prepass(d, &rd, prepass->implied, prepass->nblks, prepass->hit_end);
   else if (prepass) *prepass->nblks = rt_deskew_prepass(d, &rd, prepass->delays, prepass->hit_end);
   else ok = rt_process_blocks(d, &rd, 0x7fffffff);
   if (in_name && !prepass) rt_write_summary(d, in_name, difftime(time(NULL), (time_t)wall0));
   if (stats) {
      stats->attempts = rp.attempts; stats->exact_scans = rp.exact_scans; stats->chained = rp.chained;
      stats->events_delivered = rp.events_delivered; stats->agc_mismatches = rp.agc_mismatches;
      stats->blocks = d->numblks; stats->tapemarks = d->numtapemarks; stats->blocks_with_errors = d->numblks_err;
      stats->blocks_with_warnings = d->numblks_warn; stats->blocks_unusable = d->numblks_unusable; stats->all_ok = ok;
      stats->data_bytes = d->numdatabytes; stats->device_failures = rp.device_failures;
      stats->reference_fatal = rp.reference_fatal; stats->fatal_row = rp.fatal_row; stats->fatal_trk = rp.fatal_trk; }
   if (d->tapf) fclose(d->tapf);
   if (d->logf) fclose(d->logf);
   if (rp.evtf) { fflush(rp.evtf); if (ftruncate(fileno(rp.evtf), ftell(rp.evtf))) {} fclose(rp.evtf); }
   rt_dec_free(d);
   return (prepass && rp.reference_fatal) ? -3 : 0; }      /* (a pre-pass has no statistics to carry it: the reference died inside the pre-pass) */

int rt_replay_run(const struct rt_options *opt, const struct rt_parms *parmsets, int nparm,
                  int64_t tdelta_ns, int64_t tstart_ns, int64_t nrows, int64_t row_base, const int *W,
                  const rtfe_burst *bursts, int64_t nbursts, const uint32_t *counts, const rtfe_event *events,
                  rt_exact_fn exact, rt_exact_free_fn exact_free, void *user,
                  const char *tap_path, const char *log_path, const char *evt_path, struct rt_replay_stats *stats) {
   return replay_any(opt, parmsets, nparm, tdelta_ns, tstart_ns, nrows, row_base, W, bursts, nbursts, counts, events,
                     exact, exact_free, user, tap_path, log_path, evt_path, stats, 0, NULL, 0, INT64_MAX, 0, NULL, NULL); }

/* One fragment of a tape (a time shard, or one window of a streamed file): rows are relative to the fragment's first row
 * (row_base = its absolute index; the bursts carry absolute rows).  The decode starts at start_row (0, or the start of the zone
 * in front of the fragment's first own burst: any start inside a zone is the same detector state) and ends when an
 * attempt would start at or behind stop_row (the start of the first zone the next fragment owns; INT64_MAX: run to the end of
 * the data).  No end-of-medium marker is written: the fragments' .tap files are concatenated by the caller. */
int rt_replay_run_fragment(const struct rt_options *opt, const struct rt_parms *parmsets, int nparm,
                  int64_t tdelta_ns, int64_t tstart_ns, int64_t nrows, int64_t row_base, const int *W,
                  const rtfe_burst *bursts, int64_t nbursts, const uint32_t *counts, const rtfe_event *events,
                  rt_exact_fn exact, rt_exact_free_fn exact_free, void *user,
                  const char *tap_path, const char *log_path, const char *evt_path, struct rt_replay_stats *stats,
                  int64_t start_row, int64_t stop_row) {
   return replay_any(opt, parmsets, nparm, tdelta_ns, tstart_ns, nrows, row_base, W, bursts, nbursts, counts, events,
                     exact, exact_free, user, tap_path, log_path, evt_path, stats, 0, NULL, start_row, stop_row, 1, NULL, NULL); }

/* ---- fragments side by side: native threads (the caller is one Python thread per window, not one per fragment) ---- */
#include <pthread.h>
struct frag_job {
   const struct rt_options *opt; const struct rt_parms *parmsets; int nparm; int64_t tdelta_ns, tstart_ns, nrows, row_base; const int *W;
   const rtfe_burst *bursts; int64_t nbursts; const uint32_t *counts; const rtfe_event *events;
   rt_exact_fn exact; rt_exact_free_fn exact_free; void *user;
   const char *tap_path; int64_t start_row, stop_row; struct rt_replay_stats *stats; double *seconds; int rc; };
static void *frag_thread(void *p) {
   struct frag_job *j = (struct frag_job *)p;
   struct timespec a, b;
   clock_gettime(CLOCK_MONOTONIC, &a);
   j->rc = rt_replay_run_fragment(j->opt, j->parmsets, j->nparm, j->tdelta_ns, j->tstart_ns, j->nrows, j->row_base, j->W, j->bursts, j->nbursts, j->counts, j->events,
                                  j->exact, j->exact_free, j->user, j->tap_path, NULL, NULL, j->stats, j->start_row, j->stop_row);
   clock_gettime(CLOCK_MONOTONIC, &b);
   *j->seconds = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
   return NULL; }
int rt_replay_run_fragments(const struct rt_options *opt, const struct rt_parms *parmsets, int nparm,
                  int64_t tdelta_ns, int64_t tstart_ns, int64_t nrows, int64_t row_base, const int *W,
                  const rtfe_burst *bursts, int64_t nbursts, const uint32_t *counts, const rtfe_event *events,
                  rt_exact_fn exact, rt_exact_free_fn exact_free, void *user,
                  int nfrag, const char *const *tap_paths, const int64_t *start_rows, const int64_t *stop_rows, struct rt_replay_stats *stats, double *seconds) {
   if (nfrag < 1 || nfrag > 256) return -1;
   struct frag_job *jobs = (struct frag_job *)calloc((size_t)nfrag, sizeof *jobs);
   pthread_t *th = (pthread_t *)calloc((size_t)nfrag, sizeof *th);
   if (!jobs || !th) { free(jobs); free(th); return -1; }
   for (int i = 0; i < nfrag; ++i) {
      struct frag_job j = { opt, parmsets, nparm, tdelta_ns, tstart_ns, nrows, row_base, W, bursts, nbursts, counts, events, exact, exact_free, user,
                            tap_paths[i], start_rows[i], stop_rows[i], &stats[i], &seconds[i], 0 };
      jobs[i] = j; }
   int started = 0, rc = 0;
   for (int i = 1; i < nfrag; ++i) { if (pthread_create(&th[i], NULL, frag_thread, &jobs[i]) != 0) break; started = i; }
   frag_thread(&jobs[0]);                                       /* (the caller's thread takes the first) */
   for (int i = started + 1; i < nfrag; ++i) frag_thread(&jobs[i]);      /* (threads that could not be made: in line) */
   for (int i = 1; i <= started; ++i) pthread_join(th[i], NULL);
   for (int i = 0; i < nfrag; ++i) if (jobs[i].rc != 0 && rc == 0) rc = jobs[i].rc;
   free(jobs); free(th);
   return rc; }

struct read_job { int fd; char *dst; int64_t off, n; int rc; };
static void *read_thread(void *p) {
   struct read_job *j = (struct read_job *)p;
   int64_t done = 0;
   while (done < j->n) {
      const int64_t want = j->n - done > (1ll << 30) ? (1ll << 30) : j->n - done;
      const ssize_t got = pread(j->fd, j->dst + done, (size_t)want, (off_t)(j->off + done));
      if (got <= 0) { j->rc = -1; return NULL; }
      done += got; }
   return NULL; }
int rt_read_mt(int fd, void *dst, int64_t off, int64_t nbytes, int nthreads) {
   if (nbytes <= 0) return 0;
   if (nthreads < 1) nthreads = 1;
   if (nthreads > 64) nthreads = 64;
   if (nbytes < (8ll << 20)) nthreads = 1;
   struct read_job jobs[64];
   pthread_t th[64];
   const int64_t step = ((nbytes + nthreads - 1) / nthreads + 4095) & ~4095ll;
   int n = 0;
   for (int64_t a = 0; a < nbytes; a += step, ++n) { struct read_job j = { fd, (char *)dst + a, off + a, nbytes - a < step ? nbytes - a : step, 0 }; jobs[n] = j; }
   int started = 0;
   for (int i = 1; i < n; ++i) { if (pthread_create(&th[i], NULL, read_thread, &jobs[i]) != 0) break; started = i; }
   read_thread(&jobs[0]);
   for (int i = started + 1; i < n; ++i) read_thread(&jobs[i]);
   for (int i = 1; i <= started; ++i) pthread_join(th[i], NULL);
   for (int i = 0; i < n; ++i) if (jobs[i].rc != 0) return -1;
   return 0; }

int rt_replay_run_after_deskew(const struct rt_options *opt, const struct rt_parms *parmsets, int nparm,
                  int64_t tdelta_ns, int64_t tstart_ns, int64_t nrows, int64_t row_base, const int *W,
                  const rtfe_burst *bursts, int64_t nbursts, const uint32_t *counts, const rtfe_event *events,
                  rt_exact_fn exact, rt_exact_free_fn exact_free, void *user,
                  const char *tap_path, const char *log_path, const char *evt_path, struct rt_replay_stats *stats) {
   return replay_any(opt, parmsets, nparm, tdelta_ns, tstart_ns, nrows, row_base, W, bursts, nbursts, counts, events,
                     exact, exact_free, user, tap_path, log_path, evt_path, stats, 1, NULL, 0, INT64_MAX, 0, NULL, NULL); }

int rt_replay_deskew(const struct rt_options *opt, const struct rt_parms *parmsets, int nparm,
                  int64_t tdelta_ns, int64_t tstart_ns, int64_t nrows, int64_t row_base, const int *W,
                  const rtfe_burst *bursts, int64_t nbursts, const uint32_t *counts, const rtfe_event *events,
                  rt_exact_fn exact, rt_exact_free_fn exact_free, void *user,
                  const char *log_path, const char *evt_path, int append, int *delays, int *nblks, int *hit_end) {
   struct deskew_out o = { delays, nblks, hit_end, NULL, NULL };
   return replay_any(opt, parmsets, nparm, tdelta_ns, tstart_ns, nrows, row_base, W, bursts, nbursts, counts, events,
                     exact, exact_free, user, NULL, log_path, evt_path, NULL, append, &o, 0, INT64_MAX, 0, NULL, NULL); }


int rt_replay_density(const struct rt_options *opt, const struct rt_parms *parmsets, int nparm,
                  int64_t tdelta_ns, int64_t tstart_ns, int64_t nrows, int64_t row_base, const int *W,
                  const rtfe_burst *bursts, int64_t nbursts, const uint32_t *counts, const rtfe_event *events,
                  rt_exact_fn exact, rt_exact_free_fn exact_free, void *user,
                  const char *log_path, const char *evt_path, float *bpi, float *implied, int *nblks, int *hit_end) {
   struct deskew_out o = { NULL, nblks, hit_end, bpi, implied };
   return replay_any(opt, parmsets, nparm, tdelta_ns, tstart_ns, nrows, row_base, W, bursts, nbursts, counts, events,
                     exact, exact_free, user, NULL, log_path, evt_path, NULL, 0, &o, 0, INT64_MAX, 0, NULL, NULL); }

/* ... with the output files made by name as the reference makes them (<out_base>.tap with opt->tap_format, else the numbered
 * <out_base>.NNN.bin files, src/readtape.c:1091-1111) and the end-of-run summary (src/readtape.c:2021-2044) in the log */
int rt_replay_run_named(const struct rt_options *opt, const struct rt_parms *parmsets, int nparm,
                  int64_t tdelta_ns, int64_t tstart_ns, int64_t nrows, int64_t row_base, const int *W,
                  const rtfe_burst *bursts, int64_t nbursts, const uint32_t *counts, const rtfe_event *events,
                  rt_exact_fn exact, rt_exact_free_fn exact_free, void *user,
                  const char *out_base, const char *in_name, const char *log_path, const char *evt_path, int append, struct rt_replay_stats *stats) {
   return replay_any(opt, parmsets, nparm, tdelta_ns, tstart_ns, nrows, row_base, W, bursts, nbursts, counts, events,
                     exact, exact_free, user, NULL, log_path, evt_path, stats, append, NULL, 0, INT64_MAX, 0, out_base, in_name); }

/* ---------------------------------------------------------------------------------------------------------------------
 * Whirlwind: one chain per tape.  The device detector's state is handed from attempt to attempt (and from chunk to chunk of
 * an attempt); an attempt starts by re-seeding the tracks one per row (src/decoder.c:855-861, after src/decode_ww.c:45 zeroed
 * t_lastpeak) and ends where the host decoder says the clock has stopped (src/decoder.c:892-894) or a block mark was seen.
 * --------------------------------------------------------------------------------------------------------------------- */
static int pred_ww(const struct rt_dec *d, const struct rt_trk *t, double timenow) {
   (void)t;
   return timenow - d->ww.t_lastclkpulseend > d->ww.clkavg.t_bitspaceavg * 1.5f; }             /* WW_CLKSTOP_BITS */

static int ww_readblock(void *ctx, int retry) {
   struct rt_replay *rp = (struct rt_replay *)ctx;
   struct rt_dec *d = rp->d;
   const int ntrks = rp->ntrks, W = rp->W[0];
   const size_t sbytes = (size_t)ntrks * sizeof(rtfe_ww_track);
   const int64_t s0 = rp->pos, nrows = rp->nrows, L = rp->ww_chunk_rows;
   int endfile = 0;
   (void)retry;
   ++rp->attempts;
   if (s0 >= nrows) { rt_finish_attempt(d); return 0; }
   int64_t chunk_first = s0;
   memcpy(rp->ww_chunk_state, rp->ww_state, sbytes);
   struct evsrc src; memset(&src, 0, sizeof src);
   int have_chunk = 0;
   int64_t row = s0;
   for (;;) {
      if (!have_chunk) {                                      /* the events of rows [chunk_first, chunk_first + L) */
         if (rp->ww_scan(rp->ww_user, chunk_first, L, s0, rp->ww_chunk_state, rp->ww_chunk_end, rp->ww_counts, rp->ww_events, rp->ww_cap) != 0) {
            ++rp->device_failures; d->results[d->parmset].blktype = RT_BS_ABORTED; rt_finish_attempt(d); return 0; }
         for (int t = 0; t < ntrks; ++t) { src.list[t] = rp->ww_events + (size_t)t * rp->ww_cap; src.n[t] = rp->ww_counts[t]; src.at[t] = 0; }
         src.reset = chunk_first; src.end = chunk_first + L;
         for (int t = 0; t < ntrks; ++t) evsrc_sync(&src, t);
         ++rp->exact_scans;
         have_chunk = 1; }
      /* ---- the next row at which anything can happen ---- */
      int64_t next = nrows;
      if (row < s0 + ntrks) next = row;                       /* the rows on which the tracks are re-seeded, one by one */
      { const int64_t r = evsrc_next_row(&src, ntrks); if (r < next) next = r; }
      if (d->ww.datablock && d->ww.t_lastclkpulseend > 0) {
         const int64_t r = first_row_where(rp, pred_ww, NULL, d->ww.t_lastclkpulseend + d->ww.clkavg.t_bitspaceavg * 1.5f, row, next);
         if (r < next) next = r; }
      if (next >= src.end && src.end < nrows) {               /* nothing more in this chunk: the next one starts from the state behind it */
         chunk_first = src.end;
         memcpy(rp->ww_chunk_state, rp->ww_chunk_end, sbytes);
         have_chunk = 0;
         if (row < chunk_first) row = chunk_first;
         continue; }
      if (next >= nrows) {                                    /* end of data (src/readtape.c:1410-1413; force_end_of_block has no Whirlwind branch) */
         d->timenow = time_of_row(rp, nrows - 1);
         rp->pos = nrows;
         endfile = 1;
         break; }
      row = next;
      rp->cur_row = row;
      d->timenow = time_of_row(rp, row);
      for (int t = 0; t < ntrks; ++t) {                       /* process_sample, src/decoder.c:846-866 */
         struct rt_trk *tk = &d->trk[t];
         if (tk->t_lastpeak == 0) {                            /* first sample of this track in this attempt */
            tk->v_lastpeak = 0;
            tk->t_lastpeak = d->timenow;
            break; }
         while (src.nr[t] == row) {
            if (src.list[t][src.at[t]].flags & RTFE_EV_FATAL) {
               rp->reference_fatal = 1; d->fatal = 1; rp->fatal_row = row; rp->fatal_trk = t;
               d->results[d->parmset].blktype = RT_BS_ABORTED;
               rt_finish_attempt(d);
               return 0; }
            deliver(rp, &src, t, &src.list[t][src.at[t]], W);
            ++src.at[t]; evsrc_sync(&src, t); } }
      evsrc_skip_before(&src, ntrks, row + 1);
      if (rt_ww_end_due(d)) rt_ww_end_of_block(d);             /* src/decoder.c:892-894 */
      if (d->results[d->parmset].blktype != RT_BS_NONE) { rp->pos = row + 1; break; }
      ++row; }
   if (!endfile) {
      /* where the next attempt starts: the detector's state behind the block's last row */
      const int64_t last = rp->pos - 1;
      if (rp->ww_scan(rp->ww_user, chunk_first, last - chunk_first + 1, s0, rp->ww_chunk_state, rp->ww_state, rp->ww_counts, rp->ww_events, rp->ww_cap) != 0) {
         ++rp->device_failures; d->results[d->parmset].blktype = RT_BS_ABORTED; rt_finish_attempt(d); return 0; } }
   rt_finish_attempt(d);
   return !endfile; }

int rt_replay_run_ww(const struct rt_options *opt, const struct rt_parms *parmsets,
                  int64_t tdelta_ns, int64_t tstart_ns, int64_t nrows, int W0, rt_ww_scan_fn scan, void *user, const void *initial_state, int64_t chunk_rows,
                  const char *tap_path, const char *out_base, const char *in_name, const char *log_path, const char *evt_path, struct rt_replay_stats *stats,
                  int deskew, int *delays_out) {
   const float sample_deltat = (float)tdelta_ns / 1e9f;
   struct rt_options o = *opt;
   o.multiple_tries = 0;                                       /* "not implemented yet for Whirlwind" (src/readtape.c:1987) */
   struct rt_dec *d = rt_dec_new(&o, sample_deltat, tdelta_ns);
   if (!d) return -1;
   if (parmsets) { memset(d->parmsets, 0, sizeof d->parmsets); memcpy(d->parmsets, parmsets, sizeof(struct rt_parms)); }
   else for (int i = 1; i < RT_MAXPARMSETS; ++i) d->parmsets[i].active = 0;
   if (out_base) snprintf(d->outbase, sizeof d->outbase, "%s", out_base);
   else if (tap_path) d->tapf = fopen(tap_path, "wb");
   if (log_path) d->logf = fopen(log_path, "w");
   struct rt_replay rp; memset(&rp, 0, sizeof rp);
   rp.d = d; rp.ntrks = o.ntrks; rp.nparm = 1; rp.W[0] = W0;
   rp.nrows = nrows; rp.tstart_ns = tstart_ns; rp.tdelta_ns = tdelta_ns; rp.stop_row = INT64_MAX;
   rp.ww_scan = scan; rp.ww_user = user;
   rp.ww_chunk_rows = chunk_rows > 64 ? chunk_rows : 64;
   rp.ww_cap = (uint32_t)rp.ww_chunk_rows;                     /* (no more events than rows) */
   const size_t sbytes = (size_t)o.ntrks * sizeof(rtfe_ww_track);
   rp.ww_state = (unsigned char *)malloc(3 * sbytes); rp.ww_chunk_state = rp.ww_state + sbytes; rp.ww_chunk_end = rp.ww_state + 2 * sbytes;
   memcpy(rp.ww_state, initial_state, sbytes);
   rp.ww_events = (rtfe_event *)malloc((size_t)o.ntrks * rp.ww_cap * sizeof(rtfe_event));
   rp.ww_counts = (uint32_t *)calloc((size_t)o.ntrks, sizeof(uint32_t));
   if (evt_path) {
      rp.evtf = fopen(evt_path, "wb");
      if (rp.evtf) { d->on_transition = dump_transition; d->on_attempt = dump_attempt; d->user = &rp; } }
   const double wall0 = (double)time(NULL);
   struct rt_reader rd = { ww_readblock, rt_replay_save_pos, rt_replay_restore_pos, &rp };
   int prepass_failed = 0;
   if (deskew) {
      /* -deskew (src/readtape.c:1676-1716): the first blocks are read once to learn the heads' skew and the pulse heights; then the
       * windows and the delay lines are cleared (init_trackpeak_state) - NOT the rings, the AGC or the decoder's track state -
       * and the tape is read again from its first row, every track behind its delay */
      int delays[RT_MAXTRKS] = { 0 }, hit_end = 0;
      rt_replay_save_pos(&rp);
      const int nblks = rt_deskew_prepass(d, &rd, delays, &hit_end);
      rt_replay_restore_pos(&rp);
      if (nblks < 0 || rp.device_failures || rp.reference_fatal) prepass_failed = 1;
      else {
         rtfe_ww_track *st = (rtfe_ww_track *)rp.ww_state;
         for (int t = 0; t < o.ntrks; ++t) {
            st[t].left = st[t].right = st[t].maxv = st[t].minv = st[t].countdown = 0;
            st[t].v_avg_height = d->trk[t].v_avg_height;
            st[t].delay = delays[t]; }
         if (delays_out) memcpy(delays_out, delays, sizeof(int) * (size_t)o.ntrks); } }
   const int ok = prepass_failed ? 0 : rt_process_blocks(d, &rd, 0x7fffffff);
   if (in_name) rt_write_summary(d, in_name, difftime(time(NULL), (time_t)wall0));
   if (stats) {
      memset(stats, 0, sizeof *stats);
      stats->attempts = rp.attempts; stats->exact_scans = rp.exact_scans; stats->events_delivered = rp.events_delivered; stats->agc_mismatches = rp.agc_mismatches;
      stats->blocks = d->numblks; stats->tapemarks = d->numtapemarks; stats->blocks_with_errors = d->numblks_err;
      stats->blocks_with_warnings = d->numblks_warn; stats->blocks_unusable = d->numblks_unusable; stats->all_ok = ok;
      stats->data_bytes = d->numdatabytes; stats->device_failures = rp.device_failures;
      stats->reference_fatal = rp.reference_fatal; stats->fatal_row = rp.fatal_row; stats->fatal_trk = rp.fatal_trk; }
   if (d->tapf) fclose(d->tapf);
   if (d->logf) fclose(d->logf);
   if (rp.evtf) fclose(rp.evtf);
   free(rp.ww_state); free(rp.ww_events); free(rp.ww_counts);
   rt_dec_free(d);
   return (prepass_failed && !rp.device_failures && !rp.reference_fatal) ? -2 : 0; }
