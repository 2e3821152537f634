/* rt_decode_nrzi.c — 7/9-track NRZI: bits from flux-transition events.
 *
 * What the reference's src/decode_nrzi.c (V3.18) computes, organised as three small machines:
 *
 *   cell     one bit time of the tape-wide clock.  Every track VOTES for the cell that just closed (a transition inside
 *            it, a transition that belongs to an earlier look at the same cell, one that already lies in the next cell, or
 *            nothing); the votes move the clock and put one bit per track into the bit matrix.  (src/decode_nrzi.c:232-314)
 *   frame    the check characters behind the data: CRC (9 track) and LRC, verified with a 512-entry step table built from
 *            the shift/xor recurrence of src/decode_nrzi.c:55-66.
 *   verdict  an ordered rule list over the per-track bit counts: tapemark, noise, ragged block, data (src/decode_nrzi.c:77-113)
 *
 * The arithmetic (float/double promotion, accumulation order over the tracks) is the reference's: the .tap has to be byte-identical.
 */
#include "rt_decode.h"

#include <string.h>

static const double IBG_SECS = 200e-6;           /* NRZI_IBG_SECS, src/decoder.h:105 */
enum { SHORTEST_BLOCK = 10,         /* NRZI_MIN_BLOCK,        src/decoder.h:106 */
       WORST_RAGGEDNESS = 10,       /* NRZI_MAX_MISMATCH,     src/decoder.h:107 */
       HEIGHT_FROM = 5, HEIGHT_TO = 15,   /* AGC_STARTBASE / AGC_ENDBASE, src/decoder.h:154-155 */
       POST_BITS = 8 };             /* empty cells that end a block (src/decode_nrzi.c:313) */
static const float WEAKEST_TRACK_RATIO = 2.0f;   /* NRZI_BADTRK_FACTOR, src/decoder.h:109 */

#define TRKBIT(d, trk) ((uint16_t)(1u << ((d)->opt.ntrks - 1 - (trk))))

/* ---------------------------------------------------------------- frame ---- */

static uint16_t crc_step[512];      /* crc_step[c ^ w] = the CRC register after character w (src/decode_nrzi.c:59-64) */
static int crc_step_ready;

static void crc_step_build(void) {
   for (int v = 0; v < 512; ++v) {
      int c = v;
      if (c & 2) c ^= 0xf0;
      c = (c >> 1) | ((c & 1) << 8);
      crc_step[v] = (uint16_t)c; }
   crc_step_ready = 1; }

/* the last 8 bit times of a block hold the CRC (9 track only) and the LRC, each inside a run of blank cells: OR-ing
 * three neighbours finds it wherever the clock put it (src/decode_nrzi.c:40-47) */
static void frame_check(struct rt_dec *d, struct rt_results *res) {
   const uint16_t *w = d->data;
   const int nine = d->opt.ntrks == 9;
   if (!crc_step_ready) crc_step_build();
   res->blktype = RT_BS_BLOCK;
   res->vparity_errs = 0;
   if (res->minbits <= 8) return;
   const int n = res->minbits;
   const int trio = w[n - 6] | w[n - 5] | w[n - 4];
   if (nine) { res->crc = trio; res->lrc = w[n - 1]; }
   else if (d->opt.ntrks == 7) res->lrc = trio;
   res->maxbits -= 8;
   res->minbits -= 8;
   rt_set_expected_parity(d, res->maxbits);
   int lrc = 0, crc = 0;
   for (int i = 0; i < res->minbits; ++i) {
      res->vparity_errs += rt_parity9(w[i]) != d->expected_parity;
      lrc ^= w[i];
      crc = crc_step[(crc ^ w[i]) & 0x1ff]; }
   crc ^= 0x1af;
   if (nine) {
      lrc ^= crc;
      res->crc_errs += crc != res->crc; }
   res->lrc_errs += lrc != res->lrc; }

/* -------------------------------------------------------------- verdict ---- */

/* what a tapemark looks like in the bit matrix: 9 cells, the mark character in the first cell and again where the
 * LRC lands (src/decode_nrzi.c:97-100) */
static int is_tapemark(const struct rt_dec *d, int minbits) {
   const uint16_t *w = d->data;
   if (minbits != 9) return 0;
   switch (d->opt.ntrks) {
   case 9: return w[0] == 0x26 && w[8] == 0x26;
   case 7: return w[0] == 0x1e && (w[3] == 0x1e || w[4] == 0x1e);
   default: return 0; } }

void rt_nrzi_end_of_block(struct rt_dec *d) {   /* src/decode_nrzi.c:77-113 */
   if (d->endblock_done) return;
   d->endblock_done = 1;
   d->nrzi.datablock = 0;
   struct rt_results *res = &d->results[d->parmset];
   const int ntrks = d->opt.ntrks;
   int shortest = RT_MAXBLOCK, longest = 0;
   float spacing_sum = 0;
   for (int k = 0; k < ntrks; ++k) {
      const struct rt_trk *t = &d->trk[k];
      spacing_sum += (float)(t->t_lastbit - t->t_firstbit) / t->datacount;
      if (longest < t->datacount) longest = t->datacount;
      if (shortest > t->datacount) shortest = t->datacount;
      if (res->alltrk_max_agc_gain < t->max_agc_gain) res->alltrk_max_agc_gain = t->max_agc_gain;
      if (res->alltrk_min_agc_gain > t->min_agc_gain) res->alltrk_min_agc_gain = t->min_agc_gain; }
   res->minbits = shortest;
   res->maxbits = longest;
   res->avg_bit_spacing = spacing_sum / ntrks;
   if (is_tapemark(d, shortest)) res->blktype = RT_BS_TAPEMARK;
   else if (longest <= SHORTEST_BLOCK) res->blktype = RT_BS_NOISE;
   else if (longest - shortest > WORST_RAGGEDNESS) { res->blktype = RT_BS_BADBLOCK; res->track_mismatch = longest - shortest; }
   else frame_check(d, res);
   d->num_trks_idle = ntrks;
   d->interblock_counter = (int)(IBG_SECS / d->sample_deltat); }

/* ----------------------------------------------------------------- cell ---- */

/* one bit of one track into the bit matrix (src/decode_nrzi.c:143-175).  The first bit of a block starts the tape-wide clock
 * one bit time before itself. */
static void put_bit(struct rt_dec *d, struct rt_trk *t, int bit, double when) {
   struct rt_nrzi *z = &d->nrzi;
   const float cell = z->clkavg.t_bitspaceavg;
   if (t->datacount == 0) { t->t_firstbit = when; t->max_agc_gain = t->agc_gain; }
   t->t_lastbit = when;
   if (!z->datablock) {
      z->datablock = 1;
      z->t_lastclock = when - cell;
      z->t_last_midbit = z->t_lastclock + RT_PARM(d).midbit * cell;
      d->t_blockstart = d->timenow; }
   const uint16_t m = TRKBIT(d, t->trknum);
   uint16_t *slot = &d->data[t->datacount];
   *slot = bit ? (uint16_t)(*slot | m) : (uint16_t)(*slot & ~m);
   d->data_time[t->datacount] = when;
   if (t->datacount < RT_MAXBLOCK) ++t->datacount;
   /* in the postamble the CRC/LRC characters restart the clock when they come more than a cell late (src/decode_nrzi.c:170-174) */
   if (bit && z->post_counter > 0 && z->t_lastclock < when - (2 - RT_PARM(d).midbit) * cell)
      z->t_lastclock = when - 2 * cell; }

/* a transition on one track: always a one bit; the AGC follows the schedule of src/decode_nrzi.c:184-230
 * (peaks 5..15 are averaged into the nominal height on the tops, afterwards every peak adjusts the gain) */
static void transition(struct rt_dec *d, struct rt_trk *t, int is_top) {
   struct rt_nrzi *z = &d->nrzi;
   const double when = is_top ? t->t_top : t->t_bot;
   const int in_data = z->post_counter == 0;
   if (d->doing_deskew && in_data && z->datablock && z->t_lastclock != 0)
      rt_record_peakstat(d, z->clkavg.t_bitspaceavg, (float)(when - z->t_lastclock), t->trknum);
   if (in_data && when < z->t_last_midbit) ++d->results[d->parmset].missed_midbits;
   put_bit(d, t, 1, when);
   const int learning = t->peakcount >= HEIGHT_FROM && t->peakcount <= HEIGHT_TO, learned = t->peakcount > HEIGHT_TO;
   if (!is_top) { if (learned && t->v_avg_height_count == 0) rt_adjust_agc(d, t); return; }
   if (learning) {
      const float h = t->v_top - t->v_bot;
      t->v_avg_height_sum += h;
      ++t->v_avg_height_count;
      t->v_heights[t->heightndx] = h;
      if (++t->heightndx >= RT_PARM(d).agc_window) t->heightndx = 0; }
   else if (learned) {
      if (t->v_avg_height_count == 0) rt_adjust_agc(d, t);
      else {
         t->v_avg_height = t->v_avg_height_sum / t->v_avg_height_count; t->v_avg_height_count = 0;
         if (!(t->v_avg_height > 0)) d->fatal = 1; } } }      /* "avg peak-to-peak voltage isn't positive" (src/decode_nrzi.c:227): the reference exits inside this callback */

void rt_nrzi_top(struct rt_dec *d, struct rt_trk *t) { transition(d, t, 1); }
void rt_nrzi_bot(struct rt_dec *d, struct rt_trk *t) { transition(d, t, 0); }

int rt_nrzi_zerocheck_due(const struct rt_dec *d) {   /* src/decoder.c:844 */
   return d->nrzi.datablock && d->timenow > d->nrzi.t_lastclock + 2 * d->nrzi.clkavg.t_bitspaceavg; }

/* -correct: when the character just completed has bad parity and one track's gain stands out (>= 2x the runner-up), that
 * track dropped a transition: flip its bit (src/decode_nrzi.c:116-140) */
static void repair_weakest_track(struct rt_dec *d, int at) {
   float top1 = 0, top2 = 0;
   int who = -1;
   for (int k = 0; k < d->opt.ntrks; ++k) {
      const float g = d->trk[k].agc_gain;
      if (g > top1) { top2 = top1; top1 = g; who = k; }
      else if (g > top2) top2 = g; }
   if (who < 0 || top1 < WEAKEST_TRACK_RATIO * top2) return;
   const uint16_t m = TRKBIT(d, who);
   d->data[at] ^= m;
   d->data_faked[at] |= m;
   d->results[d->parmset].faked_tracks |= m;
   ++d->results[d->parmset].corrected_bits; }

/* a track's vote for the cell (open, close) */
enum vote { HIT, HIT_TWICE, HIT_EARLIER, NEXT_CELL, SILENT };
static enum vote vote_of(const struct rt_trk *t, double open, double close) {
   const int last_in = t->t_lastpeak > open && t->t_lastpeak < close;
   const int prev_in = t->t_prevlastpeak > open && t->t_prevlastpeak < close;
   if (last_in) return prev_in ? HIT_TWICE : HIT;
   if (prev_in) return HIT_EARLIER;
   return t->t_lastpeak > close ? NEXT_CELL : SILENT; }

static void unput_bit(struct rt_trk *t) { if (t->datacount > -32) --t->datacount; }   /* (may reach -1: see rt_dec_new) */

/* two bit times after the clock: close the cell (src/decode_nrzi.c:232-314) */
void rt_nrzi_zerocheck(struct rt_dec *d) {
   struct rt_nrzi *z = &d->nrzi;
   const float cell = z->clkavg.t_bitspaceavg;
   const double open = z->t_last_midbit;
   const double close = z->t_lastclock + (1 + RT_PARM(d).midbit) * cell;
   z->t_last_midbit = close;
#define ON_THE_CLOCK (z->t_lastclock + cell)     /* (a late check character may restart the clock while the votes are counted) */
   double where = 0;              /* sum of the transition times inside the cell */
   int ones = 0, early_ones = 0;  /* tracks with a transition in the cell / already in the next one */
   for (int k = 0; k < d->opt.ntrks; ++k) {
      struct rt_trk *t = &d->trk[k];
      switch (vote_of(t, open, close)) {
      case HIT_TWICE:   unput_bit(t);           /* the cell was looked at before: the older one bit is one too many */
                        /* fall through */
      case HIT:         where += t->t_lastpeak; ++ones; break;
      case HIT_EARLIER: where += t->t_prevlastpeak; ++ones; break;
      case NEXT_CELL:   unput_bit(t);           /* the one bit is re-entered behind the zero of this cell */
                        put_bit(d, t, 0, ON_THE_CLOCK);
                        put_bit(d, t, 1, t->t_lastpeak);
                        ++early_ones; break;
      case SILENT:      put_bit(d, t, 0, ON_THE_CLOCK); break; } }
   if (ones == 0) {                              /* nobody: free-run the clock; the first empty cell starts the postamble count */
      if (z->post_counter) ++z->post_counter;
      else if (early_ones == 0) z->post_counter = 1;
      z->t_lastclock += cell; }
   else {
      if (z->post_counter == 1) z->post_counter = 0;          /* a single empty cell was a dropout, not the postamble */
      where /= ones;
      const int steer = z->datablock && z->post_counter == 0;
      const double due = ON_THE_CLOCK;
      const double clock = steer ? due + RT_PARM(d).pulse_adj * (where - due) : where;
      const float moved = (float)(clock - z->t_lastclock);
      if (z->post_counter == 0) rt_adjust_clock(d, &z->clkavg, moved, 0);
      z->t_lastclock = clock;
      /* (the reference indexes the character to repair with a variable only its DEBUG build advances: it is always 0) */
      if (d->opt.do_correction && rt_parity9(d->data[0]) != d->expected_parity) repair_weakest_track(d, 0);
      if (z->post_counter) ++z->post_counter; }
   if (z->post_counter >= POST_BITS) rt_nrzi_end_of_block(d);
#undef ON_THE_CLOCK
}
