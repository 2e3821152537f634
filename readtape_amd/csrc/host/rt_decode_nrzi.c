/* rt_decode_nrzi.c — 7/9-track NRZI bit recovery from flux-transition events.
 * Restates src/decode_nrzi.c (V3.18) on an explicit context.  One global bit clock follows the
 * transitions of all tracks; zeros are imputed at the mid-bit check two bit times after each clock. */
#include "rt_decode.h"

#include <string.h>

#define NRZI_IBG_SECS      200e-6   /* src/decoder.h:105 */
#define NRZI_MIN_BLOCK     10       /* src/decoder.h:106 */
#define NRZI_MAX_MISMATCH  10       /* src/decoder.h:107 */
#define NRZI_BADTRK_FACTOR 2.0      /* src/decoder.h:109 */
#define AGC_STARTBASE      5        /* src/decoder.h:154 */
#define AGC_ENDBASE        15       /* src/decoder.h:155 */

static void nrzi_postprocess(struct rt_dec *d) {   /* src/decode_nrzi.c:35-75 */
   struct rt_results *result = &d->results[d->parmset];
   uint16_t *data = d->data;
   int ntrks = d->opt.ntrks;
   result->blktype = RT_BS_BLOCK;
   result->vparity_errs = 0;
   if (result->minbits > 8) {
      if (ntrks == 9) {
         result->crc = data[result->minbits - 6] | data[result->minbits - 5] | data[result->minbits - 4];
         result->lrc = data[result->minbits - 1]; }
      else if (ntrks == 7) {
         result->lrc = data[result->minbits - 6] | data[result->minbits - 5] | data[result->minbits - 4]; }
      result->maxbits -= 8;
      result->minbits -= 8;
      rt_set_expected_parity(d, result->maxbits);
      int crc = 0, lrc = 0;
      for (int i = 0; i < result->minbits; ++i) {
         if (rt_parity9(data[i]) != d->expected_parity) ++result->vparity_errs;
         lrc ^= data[i];
         crc ^= data[i];
         if (crc & 2) crc ^= 0xf0;
         int lsb = crc & 1;
         crc >>= 1;
         if (lsb) crc |= 0x100; }
      crc ^= 0x1af;
      if (ntrks == 9) {
         lrc ^= crc;
         if (crc != result->crc) ++result->crc_errs; }
      if (lrc != result->lrc) ++result->lrc_errs; } }

void rt_nrzi_end_of_block(struct rt_dec *d) {   /* src/decode_nrzi.c:77-113 */
   struct rt_results *result = &d->results[d->parmset];
   int ntrks = d->opt.ntrks;
   if (d->endblock_done) return;
   d->endblock_done = 1;
   float avg_bit_spacing = 0;
   d->nrzi.datablock = 0;
   result->minbits = RT_MAXBLOCK;
   result->maxbits = 0;
   for (int trk = 0; trk < ntrks; ++trk) {
      struct rt_trk *t = &d->trk[trk];
      avg_bit_spacing += (float)(t->t_lastbit - t->t_firstbit) / t->datacount;
      if (t->datacount > result->maxbits) result->maxbits = t->datacount;
      if (t->datacount < result->minbits) result->minbits = t->datacount;
      if (result->alltrk_max_agc_gain < t->max_agc_gain) result->alltrk_max_agc_gain = t->max_agc_gain;
      if (result->alltrk_min_agc_gain > t->min_agc_gain) result->alltrk_min_agc_gain = t->min_agc_gain; }
   result->avg_bit_spacing = avg_bit_spacing / ntrks;
   if (result->minbits == 9
         && ((ntrks == 9 && d->data[0] == 0x26 && d->data[8] == 0x26)
             || (ntrks == 7 && d->data[0] == 0x1e && (d->data[3] == 0x1e || d->data[4] == 0x1e)))) {
      result->blktype = RT_BS_TAPEMARK; }
   else if (result->maxbits <= NRZI_MIN_BLOCK) {
      result->blktype = RT_BS_NOISE; }
   else if (result->maxbits - result->minbits > NRZI_MAX_MISMATCH) {
      result->blktype = RT_BS_BADBLOCK;
      result->track_mismatch = result->maxbits - result->minbits; }
   else nrzi_postprocess(d);
   d->num_trks_idle = ntrks;
   d->interblock_counter = (int)(NRZI_IBG_SECS / d->sample_deltat); }

static void nrzi_correct_error(struct rt_dec *d, int last_complete_byte) {   /* src/decode_nrzi.c:116-140 */
   float highest = 0, next_highest = 0;
   int badtrk = -1;
   int ntrks = d->opt.ntrks;
   for (int trknum = 0; trknum < ntrks; ++trknum) {
      float gain = d->trk[trknum].agc_gain;
      if (gain > highest) {
         next_highest = highest;
         highest = gain; badtrk = trknum; }
      else if (gain > next_highest) next_highest = gain; }
   if (badtrk >= 0 && highest >= NRZI_BADTRK_FACTOR * next_highest) {
      uint16_t mask = 1 << (ntrks - 1 - badtrk);
      d->data[last_complete_byte] ^= mask;
      d->data_faked[last_complete_byte] |= mask;
      ++d->results[d->parmset].corrected_bits;
      d->results[d->parmset].faked_tracks |= mask; } }

static void nrzi_addbit(struct rt_dec *d, struct rt_trk *t, int bit, double t_bit) {   /* src/decode_nrzi.c:143-175 */
   struct rt_nrzi *nrzi = &d->nrzi;
   t->t_lastbit = t_bit;
   if (t->datacount == 0) {
      t->t_firstbit = t_bit;
      t->max_agc_gain = t->agc_gain; }
   if (!nrzi->datablock) {
      nrzi->t_lastclock = t_bit - nrzi->clkavg.t_bitspaceavg;
      nrzi->t_last_midbit = nrzi->t_lastclock + RT_PARM(d).midbit * nrzi->clkavg.t_bitspaceavg;
      d->t_blockstart = d->timenow;
      nrzi->datablock = 1; }
   uint16_t mask = 1 << (d->opt.ntrks - 1 - t->trknum);
   d->data[t->datacount] = bit ? d->data[t->datacount] | mask : d->data[t->datacount] & ~mask;
   d->data_time[t->datacount] = t_bit;
   if (t->datacount < RT_MAXBLOCK) ++t->datacount;
   if (nrzi->post_counter > 0 && bit) {
      if (nrzi->t_lastclock < t_bit - (2 - RT_PARM(d).midbit) * nrzi->clkavg.t_bitspaceavg)
         nrzi->t_lastclock = t_bit - 2 * nrzi->clkavg.t_bitspaceavg; } }

void rt_nrzi_bot(struct rt_dec *d, struct rt_trk *t) {   /* src/decode_nrzi.c:184-197 */
   if (d->doing_deskew && d->nrzi.t_lastclock != 0 && d->nrzi.datablock && d->nrzi.post_counter == 0)
      rt_record_peakstat(d, d->nrzi.clkavg.t_bitspaceavg, (float)(t->t_bot - d->nrzi.t_lastclock), t->trknum);
   if (t->t_bot < d->nrzi.t_last_midbit && d->nrzi.post_counter == 0)
      ++d->results[d->parmset].missed_midbits;
   nrzi_addbit(d, t, 1, t->t_bot);
   if (t->peakcount > AGC_ENDBASE && t->v_avg_height_count == 0)
      rt_adjust_agc(d, t); }

void rt_nrzi_top(struct rt_dec *d, struct rt_trk *t) {   /* src/decode_nrzi.c:199-230 */
   if (d->doing_deskew && d->nrzi.t_lastclock != 0 && d->nrzi.datablock && d->nrzi.post_counter == 0)
      rt_record_peakstat(d, d->nrzi.clkavg.t_bitspaceavg, (float)(t->t_top - d->nrzi.t_lastclock), t->trknum);
   if (t->t_top < d->nrzi.t_last_midbit && d->nrzi.post_counter == 0)
      ++d->results[d->parmset].missed_midbits;
   nrzi_addbit(d, t, 1, t->t_top);
   if (t->peakcount >= AGC_STARTBASE && t->peakcount <= AGC_ENDBASE) {
      t->v_avg_height_sum += t->v_top - t->v_bot;
      ++t->v_avg_height_count;
      t->v_heights[t->heightndx] = t->v_top - t->v_bot;
      if (++t->heightndx >= RT_PARM(d).agc_window) t->heightndx = 0; }
   else if (t->peakcount > AGC_ENDBASE) {
      if (t->v_avg_height_count) {
         t->v_avg_height = t->v_avg_height_sum / t->v_avg_height_count;
         t->v_avg_height_count = 0; }
      else rt_adjust_agc(d, t); } }

int rt_nrzi_zerocheck_due(const struct rt_dec *d) {   /* src/decoder.c:844 */
   return d->nrzi.datablock && d->timenow > d->nrzi.t_lastclock + 2 * d->nrzi.clkavg.t_bitspaceavg; }

void rt_nrzi_zerocheck(struct rt_dec *d) {   /* src/decode_nrzi.c:232-314 */
   struct rt_nrzi *nrzi = &d->nrzi;
   int ntrks = d->opt.ntrks;
   int numbits = 0, numlaterbits = 0;
   double left_edge = nrzi->t_last_midbit;
   double right_edge = nrzi->t_lastclock + (1 + RT_PARM(d).midbit) * nrzi->clkavg.t_bitspaceavg;
   nrzi->t_last_midbit = right_edge;
   double avg_pos = 0;
   int last_complete_byte = 0;   /* stays 0 when DEBUG is off in the reference */
   for (int trknum = 0; trknum < ntrks; ++trknum) {
      struct rt_trk *t = &d->trk[trknum];
      int lastpeak_in_window = t->t_lastpeak > left_edge && t->t_lastpeak < right_edge;
      int prevlastpeak_in_window = t->t_prevlastpeak > left_edge && t->t_prevlastpeak < right_edge;
      if (lastpeak_in_window) {
         avg_pos += t->t_lastpeak;
         ++numbits;
         if (prevlastpeak_in_window && t->datacount > -32) --t->datacount; }      /* (may reach -1: see rt_dec_new) */
      else if (prevlastpeak_in_window) {
         avg_pos += t->t_prevlastpeak;
         ++numbits; }
      else {
         if (t->t_lastpeak > right_edge) {
            if (t->datacount > -32) --t->datacount;
            nrzi_addbit(d, t, 0, nrzi->t_lastclock + nrzi->clkavg.t_bitspaceavg);
            nrzi_addbit(d, t, 1, t->t_lastpeak);
            ++numlaterbits; }
         else nrzi_addbit(d, t, 0, nrzi->t_lastclock + nrzi->clkavg.t_bitspaceavg); } }
   if (numbits > 0) {
      if (nrzi->post_counter == 1) nrzi->post_counter = 0;
      avg_pos /= numbits;
      double expected_pos, adjusted_pos;
      expected_pos = nrzi->t_lastclock + nrzi->clkavg.t_bitspaceavg;
      if (!nrzi->datablock || nrzi->post_counter > 0) adjusted_pos = avg_pos;
      else adjusted_pos = expected_pos + RT_PARM(d).pulse_adj * (avg_pos - expected_pos);
      float delta = (float)(adjusted_pos - nrzi->t_lastclock);
      if (nrzi->post_counter == 0) rt_adjust_clock(d, &nrzi->clkavg, delta, 0);
      nrzi->t_lastclock = adjusted_pos;
      if (d->opt.do_correction && rt_parity9(d->data[last_complete_byte]) != d->expected_parity)
         nrzi_correct_error(d, last_complete_byte);
      if (nrzi->post_counter) ++nrzi->post_counter; }
   else {
      if (numlaterbits == 0 && nrzi->post_counter == 0) nrzi->post_counter = 1;
      else if (nrzi->post_counter) ++nrzi->post_counter;
      nrzi->t_lastclock += nrzi->clkavg.t_bitspaceavg; }
   if (nrzi->post_counter >= 8) rt_nrzi_end_of_block(d); }
