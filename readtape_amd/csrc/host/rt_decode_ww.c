/* rt_decode_ww.c — Whirlwind I (6 tracks, 100 BPI): characters from flux-transition events.
 *
 * What the reference's src/decode_ww.c (V3.18) computes, organised around three notions:
 *
 *   pulse    A recorded bit is a pulse: a flux change and its return.  The polarity of the tape says which peak comes first
 *            (-fluxdir; with "auto" the first peak after two silent bit times decides), so a top or bottom event is first
 *            turned into a LEADING or a TRAILING edge of a pulse on a track that has a role.
 *   roles    The -order string gives each track its role: primary / alternate clock, MSB, LSB.  Clock pulses run the cell clock;
 *            the trailing edge of a clock pulse closes a CELL, whose 2-bit character is read off the data tracks: a one wherever
 *            a data pulse began within the last bit time.  Primary and alternate tracks must agree, else a warning is counted.
 *   block    The clock stopping (1.5 bit times, tested per sample by the caller) ends a block: the characters are packed four
 *            to a byte (optionally back to front for a tape read backwards).  A lone pulse on an LSB track while there is no
 *            clock is a block mark; one that arrives while the clock's stopping is being noticed is queued for the next call.
 *
 * Unlike the other formats the per-track state survives from block to block (rt_ww_init_blockstate clears only what the
 * reference clears, src/decode_ww.c:33-49).  Arithmetic (promotion points, comparison forms) is the reference's.
 */
#include "rt_decode.h"

#include <float.h>
#include <math.h>
#include <string.h>

static const float STOPPED_AFTER_BITS = 1.5f;     /* WW_CLKSTOP_BITS,     src/decoder.h:127 */
static const float SAME_BIT_WITHIN    = 0.5f;     /* WW_PEAKSCLOSE_BITS,  src/decoder.h:128 */
static const float UNRELATED_BEYOND   = 2.0f;     /* WW_PEAKSFAR_BITS,    src/decoder.h:129 */
static const float SPEED_TOLERANCE    = 0.10f;    /* WW_MAX_CLK_VARIATION, src/decoder.h:130 */

enum role { PRI_CLK, PRI_LSB, PRI_MSB, ALT_CLK, ALT_LSB, ALT_MSB, NROLES, NO_ROLE = -1 };     /* enum wwtrk_t, src/decoder.h:121-124 */
static const char ROLE_LETTERS[] = "CLMclm";

/* -order=CMLcml...: heads in file order; a letter gives the head's track its role, 'x' leaves the head out
 * (src/readtape.c:869-902).  Returns the number of tracks, or -1 for a bad string (a role twice, a primary role missing). */
int rt_ww_assign_roles(struct rt_dec *d, const char *order, int *head_to_trk) {
   int ntrks = 0;
   for (int r = 0; r < NROLES; ++r) d->ww_type_to_trk[r] = -1;
   for (int t = 0; t < RT_MAXTRKS; ++t) d->ww_trk_to_type[t] = NO_ROLE;
   for (int head = 0; order[head]; ++head) {
      if (head >= RT_MAXTRKS) return -1;
      if (order[head] == 'x') { if (head_to_trk) head_to_trk[head] = -1; continue; }
      const char *at = strchr(ROLE_LETTERS, order[head]);
      if (!at || !order[head]) return -1;
      const int r = (int)(at - ROLE_LETTERS);
      if (d->ww_type_to_trk[r] != -1) return -1;
      d->ww_type_to_trk[r] = ntrks;
      d->ww_trk_to_type[ntrks] = r;
      if (head_to_trk) head_to_trk[head] = ntrks;
      ++ntrks; }
   if (d->ww_type_to_trk[PRI_CLK] < 0 || d->ww_type_to_trk[PRI_MSB] < 0 || d->ww_type_to_trk[PRI_LSB] < 0) return -1;
   return ntrks; }

void rt_ww_init_blockstate(struct rt_dec *d) {   /* src/decode_ww.c:33-49 */
   if (d->on_attempt) d->on_attempt(d, d->user);
   struct rt_results *res = &d->results[d->parmset];
   memset(res, 0, sizeof *res);
   res->blktype = RT_BS_NONE;
   res->alltrk_max_agc_gain = 0.0;
   res->alltrk_min_agc_gain = FLT_MAX;
   for (int k = 0; k < d->opt.ntrks; ++k) {
      struct rt_trk *t = &d->trk[k];
      t->max_agc_gain = 0.0;
      t->min_agc_gain = FLT_MAX;
      t->t_lastpeak = t->t_prevlastpeak = 0; }        /* (this is also what makes the front end re-seed the track's window) */
   struct rt_ww *w = &d->ww;
   rt_init_clkavg(&w->clkavg, 1 / (d->opt.bpi * d->opt.ips));
   w->t_lastclkpulsestart = w->t_lastclkpulseend = w->t_lastpriclkpulseend = 0;
   w->datablock = 0;
   w->datacount = 0;
   d->data[0] = 0; }                                  /* (only one bits are ever written) */

/* ---- cells ---- */

/* what one data track says about the cell that ends at t_end: 0 no such track, 1 a pulse began in the last bit time, 2 none did */
static int track_says(struct rt_dec *d, int role, double t_end, uint16_t bit) {
   const int k = d->ww_type_to_trk[role];
   if (k < 0) return 0;
   const struct rt_trk *t = &d->trk[k];
   if (t->t_lastpulsestart > t_end - d->ww.clkavg.t_bitspaceavg && t->t_lastpulsestart < t_end) {
      d->data[d->ww.datacount] |= bit;
      return 1; }
   return 2; }

static void close_cell(struct rt_dec *d, double t_end) {   /* src/decode_ww.c:72-100 */
   struct rt_results *res = &d->results[d->parmset];
   /* (both tracks are always asked: each one that saw a pulse sets the bit) */
   const int msb = track_says(d, PRI_MSB, t_end, 0x02) | track_says(d, ALT_MSB, t_end, 0x02);
   if (msb == 3) ++res->ww_missing_onebit;
   const int lsb = track_says(d, PRI_LSB, t_end, 0x01) | track_says(d, ALT_LSB, t_end, 0x01);
   if (lsb == 3) ++res->ww_missing_onebit;
   d->data[++d->ww.datacount] = 0; }

/* ---- pulses ---- */

static int is_clock(int role) { return role == PRI_CLK || role == ALT_CLK; }

static void leading_edge(struct rt_dec *d, struct rt_trk *t, double when) {   /* src/decode_ww.c:171-188 */
   struct rt_ww *w = &d->ww;
   const int role = d->ww_trk_to_type[t->trknum];
   rt_adjust_agc(d, t);
   t->t_lastpulsestart = when;
   if (!is_clock(role)) return;
   if (!w->datablock) { w->datablock = 1; d->t_blockstart = when; }
   w->t_lastclkpulsestart = when;
   if (role == PRI_CLK) w->t_lastpriclkpulsestart = when; else w->t_lastaltclkpulsestart = when;
   /* the cell length follows consecutive pulse starts of the SAME clock track (skew between the two would spoil it) */
   if (when - t->t_prevlastpeak < w->clkavg.t_bitspaceavg * UNRELATED_BEYOND)
      rt_adjust_clock(d, &w->clkavg, (float)(when - t->t_prevlastpeak), t->trknum); }

static void trailing_edge(struct rt_dec *d, struct rt_trk *t, double when) {   /* src/decode_ww.c:190-239 */
   struct rt_ww *w = &d->ww;
   struct rt_results *res = &d->results[d->parmset];
   const int role = d->ww_trk_to_type[t->trknum];
   if (d->doing_deskew && t->v_top > t->v_bot) {        /* the -deskew pre-pass is also where the pulse heights are learned (src/decoder.c:482-487) */
      t->v_avg_height_sum += t->v_top - t->v_bot;
      ++t->v_avg_height_count;
      t->v_heights[t->heightndx] = t->v_top - t->v_bot;
      if (++t->heightndx >= RT_PARM(d).agc_window) t->heightndx = 0; }
   rt_adjust_agc(d, t);
   t->t_lastpulseend = when;
   if (w->t_lastpriclkpulseend > 0) {                   /* skew statistics: this pulse end against the primary clock's last one (src/decode_ww.c:197-208) */
      float delta = (float)(when - w->t_lastpriclkpulseend);
      const float cell = w->clkavg.t_bitspaceavg;
      if (delta > -cell * 1.5 && delta < cell * 1.5) {
         if (delta <= 0 || delta < cell * 0.5) delta += cell;       /* fold onto "one cell later" */
         rt_record_peakstat(d, cell, delta, t->trknum); } }
   if (is_clock(role)) {
      if (when - w->t_lastclkpulseend > w->clkavg.t_bitspaceavg * SAME_BIT_WITHIN) close_cell(d, when);     /* (else: the other clock track of the same cell) */
      w->t_lastclkpulseend = when;
      /* a clock pulse whose twin on the other clock track is more than a cell old */
      const double twin = role == PRI_CLK ? w->t_lastaltclkpulsestart : w->t_lastpriclkpulsestart;
      if (role == PRI_CLK) w->t_lastpriclkpulseend = when;
      if (twin > 0 && twin < when - w->clkavg.t_bitspaceavg) ++res->ww_missing_clock;
      return; }
   if (role == PRI_LSB || role == ALT_LSB) {            /* a lone LSB pulse with the clock silent: a block mark (once per bit time) */
      if (w->t_lastclkpulsestart == 0 && when - w->t_lastblockmark > w->clkavg.t_bitspaceavg) {
         w->t_lastblockmark = when;
         d->t_blockstart = when - w->clkavg.t_bitspaceavg / 2;
         rt_ww_blockmark(d); } } }

void rt_ww_blockmark(struct rt_dec *d) {   /* src/decode_ww.c:163-167 */
   d->results[d->parmset].blktype = RT_BS_TAPEMARK;
   d->ww.blockmark_queued = 0; }

/* a peak of either sign: settle the tape's polarity, then it is one edge or the other (src/decode_ww.c:250-273) */
static void peak(struct rt_dec *d, struct rt_trk *t, int is_top) {
   struct rt_ww *w = &d->ww;
   const double when = is_top ? t->t_top : t->t_bot;
   const int says = is_top ? RT_FLUX_POS : RT_FLUX_NEG;        /* the polarity under which THIS peak would be a leading edge */
   if (d->opt.ww_fluxdir == RT_FLUX_AUTO) {
      if (when - w->t_lastpeak > w->clkavg.t_bitspaceavg * UNRELATED_BEYOND && d->flux_current != says) {
         if (d->flux_current != RT_FLUX_AUTO) ++d->num_flux_polarity_changes;
         d->flux_current = says;
         if (d->logf) fprintf(d->logf, "  the flux direction was set to %s based on a peak on track %d at time %.8lf\n\n",
                              says == RT_FLUX_NEG ? "negative" : "positive", t->trknum, d->timenow); } }
   else d->flux_current = d->opt.ww_fluxdir;
   w->t_lastpeak = when;
   if (d->flux_current == says) leading_edge(d, t, when);
   else if (d->flux_current != RT_FLUX_AUTO) trailing_edge(d, t, when); }
   /* (polarity still undetermined under "auto": the reference asserts; unreachable, the first peak always decides) */

void rt_ww_top(struct rt_dec *d, struct rt_trk *t) { peak(d, t, 1); }
void rt_ww_bot(struct rt_dec *d, struct rt_trk *t) { peak(d, t, 0); }

/* ---- blocks ---- */

int rt_ww_end_due(const struct rt_dec *d) {   /* src/decoder.c:892-894: the clock has stopped */
   const struct rt_ww *w = &d->ww;
   return w->datablock && w->t_lastclkpulseend > 0 && d->timenow - w->t_lastclkpulseend > w->clkavg.t_bitspaceavg * STOPPED_AFTER_BITS; }

/* the 2-bit characters -> bytes, four to a byte, most significant first (src/decode_ww.c:102-139) */
static void pack_characters(struct rt_dec *d, struct rt_results *res) {
   struct rt_ww *w = &d->ww;
   uint16_t *c = d->data;
   if (w->datacount % 8 == 1 && w->datacount >= 9) {       /* one clock too many: the first one was noise */
      memmove(c, c + 1, (size_t)(w->datacount - 1) * sizeof *c);
      --w->datacount;
      res->ww_leading_clock = 1; }
   const int n = w->datacount, nbytes = n / 4;
   if (d->opt.ww_reverse) {                               /* the tape was read backwards: last character first */
      uint16_t *out = d->data_faked;                       /* scratch: the faked-bit map is not used by this format */
      for (int i = 0; i < nbytes; ++i) {
         unsigned a = 0;
         for (int k = 0; k < 4; ++k) a = (a << 2) | (c[n - 1 - 4 * i - k] & 3u);
         out[i] = (uint16_t)((a & 0xff) << 1); }
      memcpy(c, out, (size_t)nbytes * sizeof *c);
      memset(out, 0, (size_t)nbytes * sizeof *out); }
   else for (int i = 0; i < nbytes; ++i) {                 /* in place: byte i only reads characters >= 4i */
      unsigned a = 0;
      for (int k = 0; k < 4; ++k) a = (a << 2) | (c[4 * i + k] & 3u);
      c[i] = (uint16_t)((a & 0xff) << 1); }                /* (a dummy parity bit on the right, as the writers expect) */
   res->minbits = res->maxbits = nbytes;
   if (n % 8 != 0) {                                       /* whole 16-bit words only */
      ++res->ww_bad_length;
      if (!d->doing_deskew && n > 8 && d->logf)
         fprintf(d->logf, "  *** the datacount for the next block is %d 2-bit characters, which is %d more than a multiple of 8\n", n, n % 8); }
   const float nominal = 1 / (d->opt.bpi * d->opt.ips);
   if (fabs(w->clkavg.t_bitspaceavg - nominal) / nominal > SPEED_TOLERANCE) ++res->ww_speed_err; }

void rt_ww_end_of_block(struct rt_dec *d) {   /* src/decode_ww.c:141-161 */
   struct rt_ww *w = &d->ww;
   struct rt_results *res = &d->results[d->parmset];
   rt_set_expected_parity(d, 0);
   pack_characters(d, res);
   res->blktype = RT_BS_BLOCK;
   res->avg_bit_spacing = w->clkavg.t_bitspaceavg;
   for (int k = 0; k < d->opt.ntrks; ++k) {
      const struct rt_trk *t = &d->trk[k];
      if (res->alltrk_max_agc_gain < t->max_agc_gain) res->alltrk_max_agc_gain = t->max_agc_gain;
      if (res->alltrk_min_agc_gain > t->min_agc_gain) res->alltrk_min_agc_gain = t->min_agc_gain; }
   /* an LSB pulse that ended while the clock's stopping was being noticed is a block mark: it is handed out by the next call */
   static const int lsb_roles[2] = { PRI_LSB, ALT_LSB };
   for (int i = 0; i < 2; ++i) {
      const int k = d->ww_type_to_trk[lsb_roles[i]];
      if (k < 0) continue;                               /* (the reference indexes its track array with -1 here) */
      const struct rt_trk *t = &d->trk[k];
      if (t->t_lastpulseend - w->t_lastclkpulseend > w->clkavg.t_bitspaceavg * SAME_BIT_WITHIN) {
         w->blockmark_queued = 1;
         w->t_lastblockmark = t->t_lastpulseend; } } }
