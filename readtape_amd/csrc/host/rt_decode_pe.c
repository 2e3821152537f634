/* rt_decode_pe.c — 1600 BPI phase-encoded bit recovery from flux-transition events.
 * Restates src/decode_pe.c (V3.18) on an explicit context.  Each track is self-clocking: a
 * transition inside the clock window after the previous one is a phase (clock) transition,
 * otherwise it carries a data bit. */
#include "rt_decode.h"

#include <limits.h>

#define PE_IBG_SECS        200e-6   /* src/decoder.h:116 */
#define PE_IGNORE_POSTBITS 5        /* src/decoder.h:117 */
#define PE_MIN_PREBITS     70       /* src/decoder.h:118 */
#define PE_MAX_POSTBITS    40       /* src/decoder.h:119 */
#define AGC_STARTBASE      5
#define AGC_ENDBASE        15

void rt_pe_end_of_block(struct rt_dec *d) {   /* src/decode_pe.c:33-102 */
   struct rt_results *result = &d->results[d->parmset];
   struct rt_trk *T = d->trk;
   int ntrks = d->opt.ntrks;
   if (d->endblock_done) return;
   d->endblock_done = 1;
   if (T[0].datacount <= 2 && T[0].peakcount > 75 &&
         T[2].datacount <= 2 && T[2].peakcount > 75 &&
         T[5].datacount <= 2 && T[5].peakcount > 75 &&
         T[6].datacount <= 2 && T[6].peakcount > 75 &&
         T[7].datacount <= 2 && T[7].peakcount > 75 &&
         T[8].datacount <= 2 && T[8].peakcount > 75 &&
         T[1].peakcount <= 2 && T[3].peakcount <= 2 && T[4].peakcount <= 2) {
      result->blktype = RT_BS_TAPEMARK;
      return; }
   float avg_bit_spacing = 0;
   result->minbits = RT_MAXBLOCK;
   result->maxbits = 0;
   for (int trk = 0; trk < ntrks; ++trk) {
      struct rt_trk *t = &T[trk];
      avg_bit_spacing += (float)(t->t_lastbit - t->t_firstbit) / t->datacount;
      int postamble_bits;
      if (t->datacount > 0) {
         for (postamble_bits = 0; postamble_bits <= PE_MAX_POSTBITS; ++postamble_bits) {
            --t->datacount;
            if ((d->data_faked[t->datacount] & (1 << (ntrks - 1 - trk))) != 0)
               --d->results[d->parmset].corrected_bits;
            if (postamble_bits > PE_IGNORE_POSTBITS && (d->data[t->datacount] & (1 << (ntrks - 1 - trk))) != 0)
               break; }
         if (result->alltrk_max_agc_gain < t->max_agc_gain) result->alltrk_max_agc_gain = t->max_agc_gain;
         if (result->alltrk_min_agc_gain > t->min_agc_gain) result->alltrk_min_agc_gain = t->min_agc_gain; }
      if (t->datacount > result->maxbits) result->maxbits = t->datacount;
      if (t->datacount < result->minbits) result->minbits = t->datacount; }
   result->avg_bit_spacing = avg_bit_spacing / ntrks;
   rt_set_expected_parity(d, result->maxbits);
   if (result->maxbits == 0) {
      result->blktype = RT_BS_NOISE; }
   else {
      result->blktype = RT_BS_BLOCK;
      d->interblock_counter = (int)(PE_IBG_SECS / d->sample_deltat);
      if (result->minbits != result->maxbits)
         result->track_mismatch = result->maxbits - result->minbits;
      result->vparity_errs = 0;
      for (int i = 0; i < result->minbits; ++i)
         if (rt_parity9(d->data[i]) != d->expected_parity) ++result->vparity_errs; } }

static void pe_addbit(struct rt_dec *d, struct rt_trk *t, int bit, int faked, double t_bit) {   /* src/decode_pe.c:104-125 */
   if (t->t_lastbit == 0) t->t_lastbit = t_bit - 1 / (d->opt.bpi * d->opt.ips);
   if (t->datablock) {
      t->lastdatabit = (uint8_t)bit;
      if (!t->idle && !faked) {
         float delta = (float)(t_bit - t->t_lastbit);
         rt_adjust_clock(d, &t->clkavg, delta, t->trknum);
         t->t_clkwindow = t->clkavg.t_bitspaceavg / 2 * RT_PARM(d).clk_factor; }
      t->t_lastbit = t_bit;
      if (t->datacount == 0) t->t_firstbit = t_bit;
      uint16_t mask = 1 << (d->opt.ntrks - 1 - t->trknum);
      d->data[t->datacount] = bit ? d->data[t->datacount] | mask : d->data[t->datacount] & ~mask;
      d->data_faked[t->datacount] = faked ? d->data_faked[t->datacount] | mask : d->data_faked[t->datacount] & ~mask;
      if (faked) ++d->results[d->parmset].corrected_bits;
      d->data_time[t->datacount] = t_bit;
      if (t->datacount < RT_MAXBLOCK) ++t->datacount; } }

static void pe_preamble_peak(struct rt_dec *d, struct rt_trk *t, int is_top) {   /* src/decode_pe.c:127-155 */
   if (t->peakcount == 1) {
      t->bit1_up = !is_top;
      d->t_blockstart = d->timenow; }
   if (t->peakcount > PE_MIN_PREBITS
         && t->bit1_up == is_top
         && (is_top ? t->t_top : t->t_bot) - t->t_lastpeak > t->t_clkwindow) {
      t->datablock = 1;
      t->v_avg_height = t->v_avg_height_sum / t->v_avg_height_count; }
   else {
      t->clknext = is_top != t->bit1_up;
      if (t->peakcount >= AGC_STARTBASE && t->peakcount <= AGC_ENDBASE) {
         if (t->v_top > t->v_bot) {
            t->v_avg_height_sum += t->v_top - t->v_bot;
            ++t->v_avg_height_count;
            t->v_heights[t->heightndx] = t->v_top - t->v_bot;
            if (++t->heightndx >= RT_PARM(d).agc_window) t->heightndx = 0; } } } }

void rt_pe_top(struct rt_dec *d, struct rt_trk *t) {   /* src/decode_pe.c:157-178 */
   if (t->datablock) {
      int missed_transition = (t->t_top + t->t_pulse_adj) - t->t_lastpeak > t->t_clkwindow;
      if (!t->clknext || missed_transition) {
         pe_addbit(d, t, t->bit1_up, 0, t->t_top);
         t->clknext = 1; }
      else t->clknext = 0;
      t->t_pulse_adj = ((float)(t->t_top - t->t_lastpeak) - t->clkavg.t_bitspaceavg / (missed_transition ? 1 : 2)) * RT_PARM(d).pulse_adj;
      rt_adjust_agc(d, t); }
   else pe_preamble_peak(d, t, 1); }

void rt_pe_bot(struct rt_dec *d, struct rt_trk *t) {   /* src/decode_pe.c:180-201 */
   if (t->datablock) {
      int missed_transition = (t->t_bot + t->t_pulse_adj) - t->t_lastpeak > t->t_clkwindow;
      if (!t->clknext || missed_transition) {
         pe_addbit(d, t, !t->bit1_up, 0, t->t_bot);
         t->clknext = 1; }
      else t->clknext = 0;
      t->t_pulse_adj = ((float)(t->t_bot - t->t_lastpeak) - t->clkavg.t_bitspaceavg / (missed_transition ? 1 : 2)) * RT_PARM(d).pulse_adj;
      rt_adjust_agc(d, t); }
   else pe_preamble_peak(d, t, 0); }

void rt_pe_generate_fake_bits(struct rt_dec *d, struct rt_trk *t) {   /* src/decode_pe.c:204-258, strategy 1 */
   int numbits = (int)((float)(d->timenow - t->t_lastbit) / t->clkavg.t_bitspaceavg);
   if (numbits > 0) {
      while (numbits--) pe_addbit(d, t, t->lastdatabit, 1, d->timenow);
      t->t_lastbit = 0;
      if (t->lastdatabit == 0) t->clknext = 0;
      else t->clknext = 1; } }
