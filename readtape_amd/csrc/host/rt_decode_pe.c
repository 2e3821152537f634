/* rt_decode_pe.c — 1600 BPI phase encoding: bits from flux-transition events.
 *
 * What the reference's src/decode_pe.c (V3.18) computes.  Every track clocks itself, so the decoder is one small machine
 * per track with two phases:
 *
 *   preamble   count transitions, learn the nominal pulse height (peaks 5..15) and the polarity of a one; the first
 *              transition of that polarity that comes a full bit after its predecessor (behind >= 70 peaks) is the
 *              all-ones marker that ends the preamble                                     (src/decode_pe.c:127-155)
 *   data       a transition either sits on the cell boundary (phase transition: no information) or mid-cell (a bit whose
 *              value is its direction); which one is decided by the clock window          (src/decode_pe.c:157-201)
 *
 * and a block-level verdict (tapemark pattern, noise, data with the postamble stripped; src/decode_pe.c:33-102).
 * Float/double promotion and the order of accumulation are the reference's: the .tap has to be byte-identical.
 */
#include "rt_decode.h"

#include <limits.h>

static const double GAP_SECS = 200e-6;          /* PE_IBG_SECS, src/decoder.h:116 */
enum { POSTAMBLE_ZEROS_KEPT_OUT = 5,            /* PE_IGNORE_POSTBITS: postamble bits that may be garbage, src/decoder.h:117 */
       PREAMBLE_PEAKS = 70,                     /* PE_MIN_PREBITS,  src/decoder.h:118 */
       POSTAMBLE_LIMIT = 40,                    /* PE_MAX_POSTBITS, src/decoder.h:119 */
       LEARN_FROM = 5, LEARN_TO = 15 };         /* AGC_STARTBASE / AGC_ENDBASE */

#define TRKBIT(d, trk) ((uint16_t)(1u << ((d)->opt.ntrks - 1 - (trk))))

/* ---- verdict ---- */

/* A PE tapemark: flux reversals at the preamble rate on tracks 0 2 5 6 7 8 (P 0 2 5 6 7 in ANSI numbering), tracks 1 3 4
 * erased (src/decode_pe.c:40-50).  "Recorded" = many peaks and practically no data bits; "erased" = practically no peaks. */
static int looks_like_tapemark(const struct rt_trk *T) {
   static const unsigned recorded = 0x1e5, erased = 0x01a;      /* bit k = track k */
   for (int k = 0; k < 9; ++k) {
      if ((recorded >> k & 1) && !(T[k].datacount <= 2 && T[k].peakcount > 75)) return 0;
      if ((erased >> k & 1) && T[k].peakcount > 2) return 0; }
   return 1; }

/* take the postamble (zeros, then the all-ones marker) off one track: walk back over at most 41 bits until a one that is
 * not among the first 6 (those may be garbage); faked bits that go away are no longer corrections (src/decode_pe.c:62-70) */
static void strip_postamble(struct rt_dec *d, struct rt_trk *t) {
   const uint16_t m = TRKBIT(d, t->trknum);
   for (int gone = 0; gone <= POSTAMBLE_LIMIT; ++gone) {
      const int at = --t->datacount;
      if (d->data_faked[at] & m) --d->results[d->parmset].corrected_bits;
      if (gone > POSTAMBLE_ZEROS_KEPT_OUT && (d->data[at] & m)) return; } }

void rt_pe_end_of_block(struct rt_dec *d) {   /* src/decode_pe.c:33-102 */
   if (d->endblock_done) return;
   d->endblock_done = 1;
   struct rt_results *res = &d->results[d->parmset];
   if (looks_like_tapemark(d->trk)) { res->blktype = RT_BS_TAPEMARK; return; }
   const int ntrks = d->opt.ntrks;
   int shortest = RT_MAXBLOCK, longest = 0;
   float spacing_sum = 0;
   for (int k = 0; k < ntrks; ++k) {
      struct rt_trk *t = &d->trk[k];
      spacing_sum += (float)(t->t_lastbit - t->t_firstbit) / t->datacount;          /* (before the postamble goes) */
      if (t->datacount > 0) {
         strip_postamble(d, t);
         if (res->alltrk_max_agc_gain < t->max_agc_gain) res->alltrk_max_agc_gain = t->max_agc_gain;
         if (res->alltrk_min_agc_gain > t->min_agc_gain) res->alltrk_min_agc_gain = t->min_agc_gain; }
      if (longest < t->datacount) longest = t->datacount;
      if (shortest > t->datacount) shortest = t->datacount; }
   res->minbits = shortest;
   res->maxbits = longest;
   res->avg_bit_spacing = spacing_sum / ntrks;
   rt_set_expected_parity(d, longest);
   if (longest == 0) { res->blktype = RT_BS_NOISE; return; }
   res->blktype = RT_BS_BLOCK;
   d->interblock_counter = (int)(GAP_SECS / d->sample_deltat);
   if (shortest != longest) res->track_mismatch = longest - shortest;
   int bad = 0;
   for (int i = 0; i < shortest; ++i) bad += rt_parity9(d->data[i]) != d->expected_parity;
   res->vparity_errs = bad; }

/* ---- data phase ---- */

/* one data bit of one track (src/decode_pe.c:104-125).  A real bit also feeds the track's clock average, which in turn
 * sizes the window that tells phase transitions from data transitions. */
static void put_bit(struct rt_dec *d, struct rt_trk *t, int bit, int faked, double when) {
   if (t->t_lastbit == 0) t->t_lastbit = when - 1 / (d->opt.bpi * d->opt.ips);
   if (!t->datablock) return;
   t->lastdatabit = (uint8_t)bit;
   if (!faked && !t->idle) {
      rt_adjust_clock(d, &t->clkavg, (float)(when - t->t_lastbit), t->trknum);
      t->t_clkwindow = t->clkavg.t_bitspaceavg / 2 * RT_PARM(d).clk_factor; }
   t->t_lastbit = when;
   const int at = t->datacount;
   if (at == 0) t->t_firstbit = when;
   const uint16_t m = TRKBIT(d, t->trknum);
   d->data[at] = bit ? (uint16_t)(d->data[at] | m) : (uint16_t)(d->data[at] & ~m);
   d->data_faked[at] = faked ? (uint16_t)(d->data_faked[at] | m) : (uint16_t)(d->data_faked[at] & ~m);
   d->results[d->parmset].corrected_bits += faked != 0;
   d->data_time[at] = when;
   if (at < RT_MAXBLOCK) t->datacount = at + 1; }

/* ---- preamble phase (src/decode_pe.c:127-155) ---- */
static void preamble_transition(struct rt_dec *d, struct rt_trk *t, int is_top, double when) {
   if (t->peakcount == 1) {                      /* the very first transition of the block: a one goes the other way */
      t->bit1_up = !is_top;
      d->t_blockstart = d->timenow; }
   const int a_one = t->bit1_up == is_top;
   if (a_one && t->peakcount > PREAMBLE_PEAKS && when - t->t_lastpeak > t->t_clkwindow) {
      t->datablock = 1;                          /* the marker: data follow */
      t->v_avg_height = t->v_avg_height_sum / t->v_avg_height_count;
      if (!(t->v_avg_height > 0)) d->fatal = 1;      /* "avg peak-to-peak voltage isn't positive" (src/decode_pe.c:144): the reference exits here (0 / 0 where no height was learned) */
      return; }
   t->clknext = !a_one;
   if (t->peakcount < LEARN_FROM || t->peakcount > LEARN_TO || !(t->v_top > t->v_bot)) return;
   const float h = t->v_top - t->v_bot;
   t->v_avg_height_sum += h;
   ++t->v_avg_height_count;
   t->v_heights[t->heightndx] = h;
   if (++t->heightndx >= RT_PARM(d).agc_window) t->heightndx = 0; }

static void transition(struct rt_dec *d, struct rt_trk *t, int is_top) {   /* src/decode_pe.c:157-201 */
   const double when = is_top ? t->t_top : t->t_bot;
   if (!t->datablock) { preamble_transition(d, t, is_top, when); return; }
   /* more than a clock window after the previous transition (the last pulse's shift taken out): the cell-boundary
    * transition is missing, so this one carries data whatever was expected */
   const int boundary_missing = (when + t->t_pulse_adj) - t->t_lastpeak > t->t_clkwindow;
   if (boundary_missing || !t->clknext) {
      put_bit(d, t, is_top ? t->bit1_up : !t->bit1_up, 0, when);
      t->clknext = 1; }
   else t->clknext = 0;
   t->t_pulse_adj = ((float)(when - t->t_lastpeak) - t->clkavg.t_bitspaceavg / (boundary_missing ? 1 : 2)) * RT_PARM(d).pulse_adj;
   rt_adjust_agc(d, t); }

void rt_pe_top(struct rt_dec *d, struct rt_trk *t) { transition(d, t, 1); }
void rt_pe_bot(struct rt_dec *d, struct rt_trk *t) { transition(d, t, 0); }

/* a track went silent inside a block: fill in copies of its last bit for the time that passed, marked as faked
 * (src/decode_pe.c:204-258, the strategy the reference compiles in) */
void rt_pe_generate_fake_bits(struct rt_dec *d, struct rt_trk *t) {
   int missing = (int)((float)(d->timenow - t->t_lastbit) / t->clkavg.t_bitspaceavg);
   if (missing <= 0) return;
   for (; missing > 0; --missing) put_bit(d, t, t->lastdatabit, 1, d->timenow);
   t->t_lastbit = 0;
   t->clknext = t->lastdatabit != 0; }
