/* rt_decode_gcr.c — 6250 BPI group-coded recording: bit recovery from flux-transition events, then
 * 5-to-4 group recoding, ECC and parity checks.  Restates src/decode_gcr.c (V3.18) on an explicit
 * context.  Every track is self-clocked: a transition is a 1, and the time since the previous
 * transition says whether one or two 0 bits lie in between (src/decode_gcr.c:789-834).
 *
 * -correct: the single-track repair the reference's decoder invokes (src/decode_gcr.c:588-611 always passes the
 * pointer 0x01, i.e. the one-track branch of src/decode_gcr.c:262-295); its two-track branch is unreachable from the
 * decoder and is not restated.
 */
#include "rt_decode.h"

#include <string.h>

#define GCR_IBG_SECS  200e-6       /* src/decoder.h:113 */
#define AGC_STARTBASE 5
#define AGC_ENDBASE   15

/* 5-bit storage-group codes with a meaning of their own (src/decode_gcr.c:419-425) */
enum { SG_MARK1 = 0x07, SG_MARK2 = 0x1c, SG_SYNC = 0x1f };

/* storage group -> data nibble; values >= 16 mark an invalid code and carry the nearest valid nibble
 * in the low 4 bits (src/decode_gcr.c:427-436) */
static const unsigned char SG_TO_NIBBLE[32] = {
   26, 25, 18, 19, 21, 21, 22, 23, 26, 9, 10, 11, 29, 13, 14, 15,
   18, 21, 2, 3, 21, 5, 6, 7, 16, 0, 8, 1, 28, 4, 12, 31 };

void rt_gcr_preprocess(struct rt_dec *d) {           /* src/decode_gcr.c:404-408 */
   d->gcr.bitnum = d->gcr.bytenum = 0;
   d->results[d->parmset].first_error = -1; }

/* ---- bit recovery (src/decode_gcr.c:731-865) ----
 * Every track clocks itself.  A transition is a one; the time since the previous transition, less the shift the last pulse
 * is believed to have suffered, says whether no, one or two zeros lie in between (at most two: that is what the group code
 * guarantees).  The cell length follows the spacing of adjacent ones, and is re-seeded in the middle of every resync burst. */

#define TRKBIT(d, trk) ((uint16_t)(1u << ((d)->opt.ntrks - 1 - (trk))))

/* resync bursts: MARK2, a run of SYNC (all ones), MARK1 - on a 5-cell grid.  Five cells into the burst the spacing of the
 * all-ones pattern is as good a cell length as the tape offers (src/decode_gcr.c:768-786) */
static void watch_resync(struct rt_trk *t) {
   if (t->datacount % 5 == 0) {
      const unsigned group = t->lastbits & 0x1f;
      if (group == SG_MARK2) t->resync_bitcount = 1;
      if (group == SG_MARK1 && t->resync_bitcount > 0) t->resync_bitcount = 0; }
   if (t->resync_bitcount <= 0) return;
   if (t->resync_bitcount == 5) rt_force_clock(&t->clkavg, t->t_peakdelta);
   ++t->resync_bitcount; }

static void put_bit(struct rt_dec *d, struct rt_trk *t, int bit, double when) {      /* src/decode_gcr.c:731-787 */
   const int at = t->datacount;
   if (at == 0) { d->t_blockstart = when; t->t_firstbit = when; t->max_agc_gain = t->agc_gain; }
   t->t_lastbit = when;
   if (!t->datablock) { t->datablock = 1; t->t_lastclock = when - t->clkavg.t_bitspaceavg; }
   const uint16_t m = TRKBIT(d, t->trknum);
   d->data[at] = bit ? (uint16_t)(d->data[at] | m) : (uint16_t)(d->data[at] & ~m);
   d->data_time[at] = when;
   if (at < RT_MAXBLOCK) t->datacount = at + 1;
   t->lastbits = (uint8_t)(t->lastbits << 1 | bit);
   watch_resync(t); }

/* the zeros in front of a transition that comes `gap` after the previous one (src/decode_gcr.c:789-834) */
static void zeros_before(struct rt_dec *d, struct rt_trk *t, float gap) {
   if (!t->datablock) return;
   const struct rt_parms *P = &RT_PARM(d);
   const float limit[2] = { P->z1pt, P->z2pt };              /* in cells: more than z1pt = one zero, more than z2pt = two */
   const float seen = gap - t->t_pulse_adj;
   t->t_peakdeltaprev = t->t_peakdelta;
   t->t_peakdelta = gap;
   int cells = 1;
   double at = t->t_lastpeak;
   while (cells <= 2 && seen > limit[cells - 1] * t->clkavg.t_bitspaceavg) {     /* (the cell length may be re-seeded by put_bit) */
      at += t->clkavg.t_bitspaceavg;
      put_bit(d, t, 0, at);
      ++cells; }
   /* two ones in adjacent cells, twice in a row: the spacing before this one is a clean sample of the cell length */
   if (cells == 1 && t->datacount > 3 && (d->data[t->datacount - 2] & TRKBIT(d, t->trknum)))
      rt_adjust_clock(d, &t->clkavg, t->t_peakdeltaprev, t->trknum);
   t->t_pulse_adj = P->pulse_adj * (cells * t->clkavg.t_bitspaceavg - gap); }

/* AGC schedule: the tops of peaks 5..15 are averaged into the nominal height, then every peak adjusts the gain
 * (src/decode_gcr.c:843,853-864; the same schedule as NRZI) */
static void transition(struct rt_dec *d, struct rt_trk *t, int is_top) {
   const double when = is_top ? t->t_top : t->t_bot;
   const float gap = (float)(when - t->t_lastpeak);
   if (d->doing_deskew && t->t_lastclock != 0) rt_record_peakstat(d, t->clkavg.t_bitspaceavg, gap, t->trknum);
   zeros_before(d, t, gap);
   put_bit(d, t, 1, when);
   const int settled = t->peakcount > AGC_ENDBASE;
   if (!is_top) { if (settled && t->v_avg_height_count == 0) rt_adjust_agc(d, t); return; }
   if (settled) {
      if (t->v_avg_height_count == 0) rt_adjust_agc(d, t);
      else {
         t->v_avg_height = t->v_avg_height_sum / t->v_avg_height_count; t->v_avg_height_count = 0;
         if (!(t->v_avg_height > 0)) d->fatal = 1; }      /* "avg peak-to-peak voltage isn't positive" (src/decode_gcr.c:862): the reference exits here */
      return; }
   if (t->peakcount < AGC_STARTBASE) return;
   const float h = t->v_top - t->v_bot;
   t->v_avg_height_sum += h;
   ++t->v_avg_height_count;
   t->v_heights[t->heightndx] = h;
   if (++t->heightndx >= RT_PARM(d).agc_window) t->heightndx = 0; }

void rt_gcr_bot(struct rt_dec *d, struct rt_trk *t) { transition(d, t, 0); }
void rt_gcr_top(struct rt_dec *d, struct rt_trk *t) { transition(d, t, 1); }

/* ---- group recoding (src/decode_gcr.c:448-674) ---- */

/* expected ECC character of the seven data characters that precede position `end-1`
 * (src/decode_gcr.c:118-144): bit i = parity of (56-bit data word AND row i) */
static unsigned gcr_ecc_of(const uint16_t *chars7) {
   static const uint64_t ROW[8] = {
      0x0f6a71994c5230ULL, 0x70110840108004ULL, 0x5a701108401080ULL, 0x372be95d5a7011ULL,
      0xe95d5a70110840ULL, 0x4c523001884412ULL, 0x2be95d5a701108ULL, 0x5d5a7011084010ULL };
   uint64_t word = 0;
   for (int i = 0; i < 7; ++i) word = (word << 8) | (uint64_t)(chars7[i] >> 1);
   unsigned ecc = 0;
   for (int i = 0; i < 8; ++i) ecc |= (unsigned)(__builtin_parityll(word & ROW[i] & 0x00ffffffffffffffULL)) << i;
   return ecc; }

/* -correct: repair ONE bad track of a data group from its syndrome (what src/decode_gcr.c:205-341 does when it is
 * called with a single track pointer, the only way the decoder calls it, :593).  The eight characters (seven data +
 * check) of a group form a code word over GF(2^8), generator x^8+x^5+x^4+x^3+1: with the tracks taken in "ECC order"
 *     E(c) = sum over the 8 data tracks of c's bit on that track * x^(ecc position of the track)
 * a good group has  sum_i E(c_i) * x^(7-i) = 0  and odd parity in every character.  If all errors lie on one track at
 * ECC position t, then with e_i = 1 where character i has even parity,
 *     S2 = sum_i E(c_i) x^(7-i) = x^t * P,   P = sum_i e_i x^(7-i)
 * so t is found by stepping P through the powers of x; S2 == 0 with P != 0 means the parity track itself.
 * Faithful details: parity is taken as odd whatever -even says; P == 0 "succeeds" without changing anything; the
 * smallest t in 0..7 wins; no t -> failure (group left as it is).  g[i] = (msb)...(lsb)(p).  Returns 1 on success. */
static int gcr_repair_track(uint16_t *g) {
   static const unsigned char ECC_POS[8] = {4, 2, 1, 5, 7, 3, 6, 0};    /* data bit k (lsb = 0) sits at x^ECC_POS[k] */
   unsigned S2 = 0, P = 0;
   for (int i = 0; i < 8; ++i) {
      unsigned e = 0;
      for (int k = 0; k < 8; ++k) e |= ((g[i] >> (k + 1)) & 1u) << ECC_POS[k];
      S2 = ((S2 << 1) ^ ((S2 & 0x80) ? 0x139u : 0u)) ^ e;                /* Horner: S2 = S2 * x + E(c_i)  (mod the generator) */
      P = (P << 1) | (rt_parity9(g[i]) ^ 1u); }                          /* character i -> coefficient of x^(7-i) */
   if (P == 0) return 1;
   unsigned flip;                                                       /* the bad track, as a bit of g[] */
   if (S2 == 0) flip = 1u;                                              /* the parity track */
   else {
      int t = 0;
      unsigned q = P;
      while (t < 8 && q != S2) { q = ((q << 1) ^ ((q & 0x80) ? 0x139u : 0u)); ++t; }
      if (t == 8) return 0;
      int k = 0;
      while (ECC_POS[k] != t) ++k;
      flip = 2u << k; }
   for (int i = 0; i < 8; ++i)
      if ((P >> (7 - i)) & 1u) g[i] ^= (uint16_t)flip;
   return 1; }

/* gather the next 5 cells of every track into per-track storage groups (src/decode_gcr.c:448-459) */
static void gather_sgroups(struct rt_dec *d) {
   for (int cell = 0; cell < 5; ++cell) {
      uint16_t w = d->data[d->gcr.bitnum + cell];
      for (int trk = 8; trk >= 0; --trk) {
         d->gcr.sgroup[trk] = (uint8_t)(((d->gcr.sgroup[trk] << 1) & 0x1f) | (w & 1));
         w >>= 1; } } }

/* one storage group per track -> 4 data cells (one bit per track each) written at bytenum
 * (src/decode_gcr.c:464-493) */
static void store_dgroup(struct rt_dec *d) {
   struct rt_results *result = &d->results[d->parmset];
   uint16_t mask = 1;
   for (int trk = 8; trk >= 0; --trk, mask <<= 1) {
      unsigned nib = SG_TO_NIBBLE[d->gcr.sgroup[trk]];
      if (nib >= 16) { ++result->gcr_bad_dgroups; nib -= 16; }
      for (int bit = 3; bit >= 0; --bit, nib >>= 1) {
         if (nib & 1) d->data[d->gcr.bytenum + bit] |= mask;
         else d->data[d->gcr.bytenum + bit] &= ~mask; } }
   for (int k = 0; k <= 3; ++k)
      if (rt_parity9(d->data[d->gcr.bytenum + k]) != d->expected_parity) {
         ++d->gcr.bad_parity_in_dgroup;
         if (result->first_error < 0) result->first_error = d->gcr.bytenum + k; }
   d->gcr.bytenum += 4; }

static void gcr_postprocess(struct rt_dec *d) {             /* src/decode_gcr.c:503-674 */
   struct rt_results *result = &d->results[d->parmset];
   enum { S_PREAMBLE, S_DATA_A, S_DATA_B, S_RESYNC, S_RESID_A, S_RESID_B, S_CRC_A, S_CRC_B, S_POSTAMBLE } state = S_PREAMBLE;
   result->blktype = RT_BS_BLOCK;
   result->first_error = -1;
   d->gcr.bitnum = 0;
   while (d->gcr.bitnum <= result->maxbits - 5) {
      gather_sgroups(d);
      d->gcr.bitnum += 5;
      const unsigned master = d->gcr.sgroup[0];            /* track 0 tells what kind of subgroup this is */
      switch (state) {
      case S_PREAMBLE:
         if (master == SG_MARK1) { state = S_DATA_A; d->gcr.bytenum = 0; }
         break;
      case S_DATA_A:
         if (master == SG_MARK2) state = S_RESYNC;
         else if (master == SG_SYNC) state = S_RESID_A;
         else { d->gcr.bad_parity_in_dgroup = 0; store_dgroup(d); state = S_DATA_B; }
         break;
      case S_DATA_B:
         store_dgroup(d);
         if (gcr_ecc_of(&d->data[d->gcr.bytenum - 8]) != (unsigned)(d->data[d->gcr.bytenum - 1] >> 1)) {
            ++result->ecc_errs;
            if (result->first_error < 0) result->first_error = d->gcr.bytenum - 1; }
         if (d->gcr.bad_parity_in_dgroup) {
            if (d->opt.do_correction && gcr_repair_track(&d->data[d->gcr.bytenum - 8])) {      /* src/decode_gcr.c:589-606 */
               d->gcr.bad_parity_in_dgroup = 0;
               for (int k = 0; k < 8; ++k)
                  if (rt_parity9(d->data[d->gcr.bytenum - 8 + k]) != d->expected_parity) ++d->gcr.bad_parity_in_dgroup;
               ++result->corrected_bits;
               if (gcr_ecc_of(&d->data[d->gcr.bytenum - 8]) != (unsigned)(d->data[d->gcr.bytenum - 1] >> 1)) ++result->ecc_errs; }
            result->vparity_errs += d->gcr.bad_parity_in_dgroup; }
         d->gcr.bytenum -= 1;                               /* drop the ECC character */
         state = S_DATA_A;
         break;
      case S_RESYNC:
         if (master == SG_MARK1) state = S_DATA_A;
         else if (master != SG_SYNC) ++result->gcr_bad_dgroups;
         break;
      case S_RESID_A: store_dgroup(d); state = S_RESID_B; break;
      case S_RESID_B: store_dgroup(d); state = S_CRC_A; break;
      case S_CRC_A:   store_dgroup(d); state = S_CRC_B; break;
      case S_CRC_B: {
         store_dgroup(d);
         const int residual_count = d->data[d->gcr.bytenum - 2] >> (5 + 1);     /* top 3 bits of the residual character */
         d->gcr.bytenum -= (16 - residual_count);
         state = S_POSTAMBLE;
         break; }
      case S_POSTAMBLE:
         break; } }
   result->minbits = result->maxbits = d->gcr.bytenum;
   d->interblock_counter = (int)(GCR_IBG_SECS / d->sample_deltat); }

/* A GCR tapemark is the PE one at GCR density: 250..400 cells of reversals on tracks 0 2 5 6 7 8, tracks 1 3 4 erased
 * (src/decode_gcr.c:706-716) */
static int looks_like_tapemark(const struct rt_trk *T) {
   static const unsigned recorded = 0x1e5, erased = 0x01a;      /* bit k = track k */
   for (int k = 0; k < 9; ++k) {
      if ((erased >> k & 1) && T[k].peakcount > 2) return 0;
      if ((recorded >> k & 1) && (T[k].datacount < 250 || T[k].datacount > 400)) return 0; }
   return 1; }

void rt_gcr_end_of_block(struct rt_dec *d) {                /* src/decode_gcr.c:682-729 */
   if (d->endblock_done) return;
   d->endblock_done = 1;
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
   rt_set_expected_parity(d, longest);
   if (longest <= 10) res->blktype = RT_BS_NOISE;
   else if (looks_like_tapemark(d->trk)) res->blktype = RT_BS_TAPEMARK;
   else if (longest - shortest > 2) { res->blktype = RT_BS_BADBLOCK; res->track_mismatch = longest - shortest; }
   else gcr_postprocess(d); }
