/* rt_driver.c — block retry / best-decoding selection and the SIMH .tap writer.
 * Restates the block loop of src/readtape.c:1720-1882 and the writers of src/readtape.c:1076-1082,
 * 1160-1176, 1212-1313, 1885 over an abstract block reader, so the same decisions are taken whether
 * the attempts are re-read one at a time (scalar path) or looked up in a batched parmset sweep. */
#include "rt_decode.h"

#include <float.h>
#include <limits.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static void rlog(struct rt_dec *d, const char *fmt, ...) {
   if (!d->logf) return;
   va_list ap; va_start(ap, fmt); vfprintf(d->logf, fmt, ap); va_end(ap); }

static void output_tap_marker(struct rt_dec *d, uint32_t num) {   /* src/readtape.c:1076-1082 */
   for (int i = 0; i < 4; ++i) {
      unsigned char lsb = num & 0xff;
      if (d->tapf) fwrite(&lsb, 1, 1, d->tapf);
      num >>= 8; }
   d->numoutbytes += 4; }

/* 1234567 -> "1,234,567": how the reference prints byte and sample counts (src/readtape.c:707-718) */
static const char *commas(long long n, char buf[32]) {
   char raw[24];
   const int len = snprintf(raw, sizeof raw, "%lld", n);
   int o = 0;
   for (int i = 0; i < len; ++i) { buf[o++] = raw[i]; if ((len - 1 - i) % 3 == 0 && i != len - 1) buf[o++] = ','; }
   buf[o] = 0;
   return buf; }

/* ---- output files (src/readtape.c:1084-1111).  Two ways to use the writers: the caller opens d->tapf itself (one .tap, nothing
 * logged about it), or it sets d->outbase and the files are made here, by the reference's names and with its log lines ---- */
void rt_close_output(struct rt_dec *d) {
   if (!d->tapf) return;
   fclose(d->tapf);
   d->tapf = NULL;
   if (!d->outbase[0]) return;
   char cb[32];
   rlog(d, "%s was closed at time %.8lf after %s data bytes were extracted from %d blocks\n", d->outname, d->timenow, commas(d->numfilebytes, cb), d->numfileblks); }

static void open_output(struct rt_dec *d) {
   if (!d->outbase[0]) return;
   if (d->tapf) rt_close_output(d);
   if (d->opt.tap_format) snprintf(d->outname, sizeof d->outname, "%s.tap", d->outbase);
   else snprintf(d->outname, sizeof d->outname, "%s.%03d.bin", d->outbase, d->numfiles + 1);
   rlog(d, "creating file \"%s\"\n", d->outname);
   d->tapf = fopen(d->outname, "wb");
   ++d->numfiles;
   d->numfilebytes = 0;
   d->numfileblks = 0;
   if (d->data_start_time == 0) d->data_start_time = d->timenow; }

void rt_tap_end(struct rt_dec *d) {   /* src/readtape.c:1885-1887: the end-of-medium marker only if the file was ever created */
   const int created = d->outbase[0] ? d->tapf != NULL : (d->tapf && d->numoutbytes > 0);
   if (d->opt.tap_format && created && !d->no_tap_end) output_tap_marker(d, 0xffffffffu);
   if (d->outbase[0]) rt_close_output(d); }

void rt_got_tapemark(struct rt_dec *d) {   /* src/readtape.c:1160-1176 */
   ++d->numtapemarks;
   rlog(d, "  tapemark at time %.8lf, tap offset %lld, %d blocks written so far\n", d->timenow, d->numoutbytes, d->numblks);
   if (d->opt.tap_format) {
      if (!d->tapf) open_output(d);
      output_tap_marker(d, 0x00000000); }
   else rt_close_output(d); }                 /* plain data files: a tapemark ends the file (no label processing here) */

static const char *format_block_errors(struct rt_dec *d, struct rt_results *result, char *buf) {   /* src/readtape.c:1179-1209 */
   char *p = buf;
   if (result->errcount > 0) {
      p += sprintf(p, "%d err%s", result->errcount, result->errcount > 1 ? "s" : "");
      if (result->track_mismatch) p += sprintf(p, ", %d bit track mismatch", result->track_mismatch);
      if (result->vparity_errs) p += sprintf(p, ", %d parity", result->vparity_errs);
      if (result->crc_errs) p += sprintf(p, ", %d CRC", result->crc_errs);
      if (result->lrc_errs) p += sprintf(p, ", 1 LRC");
      if (result->ecc_errs) p += sprintf(p, ", %d ECC", result->ecc_errs);
      if (result->ww_bad_length) p += sprintf(p, ", bad length");
      if (result->ww_speed_err) p += sprintf(p, ", bad speed"); }
   else p += sprintf(p, "ok");
   if (result->warncount > 0) {
      p += sprintf(p, ", %d warning%s", result->warncount, result->warncount > 1 ? "s" : "");
      if (d->opt.mode == RT_NRZI && result->corrected_bits > 0) {
         int trkcount = 0; uint16_t tracks = result->faked_tracks;
         for (; tracks; ++trkcount) tracks &= tracks - 1;
         p += sprintf(p, ", %d bits corrected on %d trks", result->corrected_bits, trkcount); }
      if (result->gcr_bad_dgroups) p += sprintf(p, ", %d bad dgroups", result->gcr_bad_dgroups);
      if (result->corrected_bits > 0) p += sprintf(p, ", %d corrected bits", result->corrected_bits);
      if (d->opt.mode == RT_PE) {
         int length = result->minbits, nbits = 0, ntrk = 0; uint16_t faked = 0;
         for (int i = 0; i < length; ++i) { uint16_t v = d->data_faked[i]; faked |= v; for (; v; ++nbits) v &= v - 1; }
         for (; faked; ++ntrk) faked &= faked - 1;
         if (nbits > 0) p += sprintf(p, ", %d faked bits on %d trks", nbits, ntrk); }
      if (result->ww_leading_clock) p += sprintf(p, ", leading clk");
      if (result->ww_missing_onebit) p += sprintf(p, ", missing 1-bit");
      if (result->ww_missing_clock) p += sprintf(p, ", missing clk"); }
   return buf; }

void rt_got_datablock(struct rt_dec *d, int badblock) {   /* src/readtape.c:1212-1313 (no labels, no text file) */
   struct rt_results *result = &d->results[d->parmset];
   int length = result->minbits;
   if (length > 0) {
      if (badblock) {
         ++d->numblks_unusable;
         rlog(d, "ERROR: unusable block, ");
         if (result->track_mismatch) rlog(d, "tracks mismatched with lengths %d to %d", result->minbits, result->maxbits);
         else rlog(d, "unknown reason");
         rlog(d, ", %d tries, parmset %d, at time %.8lf\n", d->tries, d->parmset, d->timenow); }
      else {
         uint32_t errflag = result->errcount ? 0x80000000u : 0;
         d->last_block_time = d->timenow;
         if (!d->tapf) open_output(d);
         if (d->opt.tap_format) output_tap_marker(d, (uint32_t)length | errflag);
         for (int i = 0; i < length; ++i) {
            unsigned char b = (unsigned char)(d->data[i] >> 1);
            if (d->opt.add_parity) b |= (d->data[i] & 1) << (d->opt.ntrks - 1);
            if (d->tapf) fwrite(&b, 1, 1, d->tapf); }
         if (d->opt.tap_format) {
            unsigned char zero = 0;
            if (length & 1) {
               if (d->tapf) fwrite(&zero, 1, 1, d->tapf);
               d->numoutbytes += 1; }
            output_tap_marker(d, (uint32_t)length | errflag); }
         if (result->errcount != 0) ++d->numblks_err;
         if (result->warncount != 0) ++d->numblks_warn;
         if (d->opt.verbose || d->numblks == 0 || result->errcount > 0 || result->warncount > 0) {
            char buf[400];
            rlog(d, "wrote block %3d, %4d bytes, %d %s, parmset %d, ", d->numblks + 1, length, d->tries, d->tries > 1 ? "tries" : "try", d->parmset);
            if (result->alltrk_min_agc_gain == FLT_MAX) rlog(d, "max AGC %.2f, ", result->alltrk_max_agc_gain);
            else rlog(d, "AGC %.2f-%.2f, ", result->alltrk_min_agc_gain, result->alltrk_max_agc_gain);
            rlog(d, "%s", format_block_errors(d, result, buf));
            rlog(d, ", avg speed %.2f IPS at time %.8lf", 1 / (result->avg_bit_spacing * d->opt.bpi), d->timenow);
            rlog(d, ", tap offset %lld\n", d->numoutbytes); }
         if (result->track_mismatch) ++d->numblks_trksmismatched;
         if (result->missed_midbits > 0) {
            ++d->numblks_midbiterrs;
            rlog(d, "   WARNING: %d bits were before the midbit using parmset %d for block %d at %.8lf\n",
                 result->missed_midbits, d->parmset, d->numblks + 1, d->timenow); }
         if (result->corrected_bits > 0) ++d->numblks_corrected;
         d->numfilebytes += length;
         d->numoutbytes += length;
         d->numdatabytes += length;
         ++d->numfileblks;
         ++d->numblks; } } }

/* the end-of-run report (src/readtape.c:2021-2044) */
void rt_write_summary(struct rt_dec *d, const char *infilename, double elapsed) {
   char cb[32];
   rlog(d, "\n");
   rlog(d, "summary for file \"%s\":\n", infilename);
   rlog(d, "  %s samples were processed in %.0lf seconds (%.3lf seconds/block)\n", commas(d->lines_in, cb), elapsed, d->numblks == 0 ? 0 : elapsed / d->numblks);
   rlog(d, "  created %d output file%s with a total of %s bytes\n", d->numfiles, d->numfiles != 1 ? "s" : "", commas(d->numoutbytes, cb));
   rlog(d, "  decoded %d tape marks and %d blocks with %s bytes from %.2lf seconds of tape data\n",
        d->numtapemarks, d->numblks, commas(d->numdatabytes, cb), d->timenow - d->data_start_time);
   if (d->last_block_time) rlog(d, "  the last block written was %.8lf seconds into the tape\n", d->last_block_time);
   rlog(d, "  %d block%s had errors, %d had warnings", d->numblks_err, d->numblks_err != 1 ? "s" : "", d->numblks_warn);
   if (d->opt.mode != RT_WW) rlog(d, ", %d had mismatched tracks, %d had bits corrected", d->numblks_trksmismatched, d->numblks_corrected);
   if (d->opt.mode == RT_NRZI) rlog(d, ", %d had midbit timing errors", d->numblks_midbiterrs);
   rlog(d, "\n");
   if (d->opt.mode == RT_WW && d->num_flux_polarity_changes > 0)
      rlog(d, "  the flux polarity changed %d time%s during decoding\n", d->num_flux_polarity_changes, d->num_flux_polarity_changes > 1 ? "s" : "");
   if (d->numblks_unusable > 0) rlog(d, "  %d blocks were unusable and were not written\n", d->numblks_unusable);
   if (!d->opt.multiple_tries) return;
   rlog(d, "  %d good blocks had to try more than one parmset\n", d->numblks_goodmultiple);
   for (int i = 0; i < RT_MAXPARMSETS; ++i)
      if (d->parmsets[i].tried > 0)
         rlog(d, "  parmset %d was tried %4d times and used %4d times, or %5.1f%%\n", i, d->parmsets[i].tried, d->parmsets[i].chosen,
              100. * d->parmsets[i].chosen / d->parmsets[i].tried); }

/* ---- one block, possibly many attempts (src/readtape.c:1720-1882) ----
 * An attempt is FINAL when nothing could improve on it: a tapemark, noise (the reference skips noise at once, SKIP_NOISE), or
 * a block without errors or warnings.  Otherwise, with -m, the next parameter set that has not been tried on this block gets
 * its turn; when none is left the attempts are RANKED and the best one is (if it was not the last one run) decoded again so
 * that the bit matrix holds its data. */
static int attempt_is_final(const struct rt_results *a) {
   return a->blktype == RT_BS_TAPEMARK || a->blktype == RT_BS_NOISE
          || (a->blktype == RT_BS_BLOCK && a->errcount == 0 && a->warncount == 0); }

/* the parameter set to try next: the first active, untried one behind the current one (cyclically), or -1 */
static int untried_parmset(const struct rt_dec *d) {
   for (int step = 1; step < RT_MAXPARMSETS; ++step) {
      const int p = (d->parmset + step) % RT_MAXPARMSETS;
      if (d->parmsets[p].active != 0 && d->results[p].blktype == RT_BS_NONE) return p; }
   return -1; }

/* rank of an attempt among the imperfect ones, smaller is better (src/readtape.c:1805-1845): error-free blocks by their
 * warnings, then blocks by their errors, then ragged blocks by their raggedness, then noise; the earlier parameter set wins
 * a tie.  Returns 0 for an attempt that cannot be chosen. */
static int attempt_rank(const struct rt_results *a, int *tier, int *badness) {
   switch (a->blktype) {
   case RT_BS_BLOCK:    *tier = a->errcount == 0 ? 0 : 1; *badness = a->errcount == 0 ? a->warncount : a->errcount; return 1;
   case RT_BS_BADBLOCK: *tier = 2; *badness = a->track_mismatch; return 1;
   case RT_BS_NOISE:    *tier = 3; *badness = 0; return 1;
   default: return 0; } }

static int best_attempt(const struct rt_dec *d, int *tier_out) {
   int best = -1, best_tier = INT_MAX, best_badness = INT_MAX;
   for (int p = 0; p < RT_MAXPARMSETS; ++p) {
      int tier, badness;
      if (!attempt_rank(&d->results[p], &tier, &badness)) continue;
      if (badness == INT_MAX) continue;                         /* (the reference's "< INT_MAX" searches cannot pick such a value either) */
      if (tier < best_tier || (tier == best_tier && badness < best_badness)) { best = p; best_tier = tier; best_badness = badness; } }
   *tier_out = best_tier;
   return best; }

int rt_process_blocks(struct rt_dec *d, struct rt_reader *r, int blklimit) {
   int all_clean = 1, out_of_data = 0;
   const int ww = d->opt.mode == RT_WW;
   d->interblock_counter = 0;
   if (ww && !d->ww_prepassed) rt_init_trackstate(d);             /* Whirlwind: once per tape - blocks may be one bit apart (src/readtape.c:1674); a -deskew pre-pass has done it */
   while (!out_of_data && d->numblks < blklimit) {
      rt_init_blockstate(d);
      d->parmset = 0;
      d->tries = 0;
      r->save_pos(r->ctx);
      int decoded_last, final = 0;
      for (;;) {                                                  /* attempts */
         decoded_last = d->parmset;
         if (ww) rt_ww_init_blockstate(d); else rt_init_trackstate(d);
         if (ww && d->ww.blockmark_queued) {                      /* the block mark seen while the last block's end was being noticed */
            rt_ww_blockmark(d);
            d->t_blockstart = d->timenow - d->ww.clkavg.t_bitspaceavg; }
         else out_of_data = !r->readblock(r->ctx, d->tries > 0);
         if (d->fatal) return 0;                                  /* the reference has exited (src/decoder.c:709-710,748,782): no other parameter set is tried */
         const struct rt_results *a = &d->results[d->parmset];
         if (a->blktype == RT_BS_NONE) { rt_tap_end(d); return all_clean; }      /* what was left of the data was no block */
         ++d->tries;
         ++RT_PARM(d).tried;
         if (attempt_is_final(a)) {
            final = 1;
            if (a->blktype == RT_BS_BLOCK && d->tries > 1) ++d->numblks_goodmultiple;
            break; }
         /* (a PE attempt with a dead track was most likely noise: no second opinion) */
         const int next = d->opt.multiple_tries && (d->opt.mode != RT_PE || a->minbits != 0) ? untried_parmset(d) : -1;
         if (next < 0) break;
         d->parmset = next;
         r->restore_pos(r->ctx);
         d->interblock_counter = 0; }
      if (!final) {
         if (d->tries == 1) { if (d->results[d->parmset].errcount > 0) all_clean = 0; }
         else {
            int tier;
            const int pick = best_attempt(d, &tier);
            if (tier > 0) all_clean = 0;
            if (pick < 0) return 0;                                /* ("block state error in process_file()") */
            d->parmset = pick; } }
      if (d->results[d->parmset].blktype == RT_BS_NOISE) continue;
      ++RT_PARM(d).chosen;
      if (d->tries > 1 && decoded_last != d->parmset) {           /* the bit matrix holds another attempt's data */
         r->restore_pos(r->ctx);
         d->interblock_counter = 0;
         rt_init_trackstate(d);
         out_of_data = !r->readblock(r->ctx, 1);
         if (d->fatal) return 0; }
      switch (d->results[d->parmset].blktype) {
      case RT_BS_TAPEMARK: rt_got_tapemark(d); break;
      case RT_BS_BLOCK:    rt_got_datablock(d, 0); break;
      case RT_BS_BADBLOCK: rt_got_datablock(d, 1); break;
      default: return 0; } }
   rt_tap_end(d);
   return all_clean; }

/* ---- -deskew pre-pass (src/readtape.c:1675-1717 + skew_compute_deskew / skew_set_delay, src/decoder.c:235-281) ---- */
int rt_deskew_prepass(struct rt_dec *d, struct rt_reader *r, int delays[RT_MAXTRKS], int *hit_end) {
   const int ntrks = d->opt.ntrks;
   int nblks = 0, min_transitions = 0;
   *hit_end = 0;
   d->doing_deskew = 1;
   d->peakstat.initialized = 0;
   d->interblock_counter = 0;
   const int ww = d->opt.mode == RT_WW;
   if (ww) { d->parmset = 0; rt_init_trackstate(d); d->ww_prepassed = 1; }   /* src/readtape.c:1674 */
   do {                                                      /* one block at a time, first parameter set only */
      rt_init_blockstate(d);
      d->parmset = 0;
      if (ww) rt_ww_init_blockstate(d); else rt_init_trackstate(d);
      if (!r->readblock(r->ctx, 1)) { *hit_end = 1; break; }
      if (d->results[d->parmset].blktype != RT_BS_NOISE) {
         min_transitions = INT_MAX;
         for (int t = 0; t < ntrks; ++t) if (d->peakstat.trksums[t] < min_transitions) min_transitions = d->peakstat.trksums[t];
         ++nblks; } }
   while (nblks < RT_MAXSKEWBLKS && min_transitions < RT_MINSKEWTRANS);
   d->doing_deskew = 0;
   d->interblock_counter = 0;
   if (!d->peakstat.initialized) { d->peakstat.initialized = 0; return -1; }
   if (min_transitions <= 0) { d->peakstat.initialized = 0; return -1; }   /* "Some tracks have no transitions" (fatal there) */
   /* the average transition position of each track, over the buckets between the two catch-alls */
   float avg[RT_MAXTRKS], maxavg = 0;
   const float bw = d->peakstat.binwidth, left = d->peakstat.leftbin;
   for (int t = 0; t < ntrks; ++t) {
      long long sum = 0;
      for (int b = 1; b < RT_PEAKSTAT_BUCKETS - 1; ++b)        /* (each term is truncated to an integer number of usec) */
         sum += (long long)(d->peakstat.counts[t][b] * (bw * 1e6 * b + left * 1e6));
      avg[t] = (float)sum / (float)d->peakstat.trksums[t];
      if (avg[t] > maxavg) maxavg = avg[t]; }
   /* delay every track up to the latest one, rounded to whole samples */
   for (int t = 0; t < ntrks; ++t) {
      const float time = d->peakstat.trksums[t] > 0 ? (maxavg - avg[t]) / 1e6f : 0;
      int delay = (int)((time + d->sample_deltat / 2) / d->sample_deltat);
      delays[t] = delay < RT_MAXSKEWSAMP ? delay : RT_MAXSKEWSAMP;
      rlog(d, "  track %d delayed by %d clocks (%.2f usec) based on %d observed flux transitions\n",
           t, delays[t], delays[t] * d->sample_deltat * 1e6, d->peakstat.trksums[t]); }
   d->peakstat.initialized = 0;
   if (ww) {                                                 /* the pulse heights learned on the way (src/readtape.c:1706-1716); the caller resets the detector */
      rlog(d, "\n");
      d->ww.t_lastblockmark = 0;
      d->ww.blockmark_queued = 0;
      for (int t = 0; t < ntrks; ++t) {
         struct rt_trk *k = &d->trk[t];
         const int count = k->v_avg_height_count;
         if (count) { k->v_avg_height = k->v_avg_height_sum / count; k->v_avg_height_count = 0; k->v_avg_height_sum = 0; }
         rlog(d, "  trk %d average peak height is %.2fV and AGC is %.2f, based on %d measurements\n", t, k->v_avg_height / 2, k->agc_gain, count); }
      rlog(d, "\n"); }
   return nblks; }

/* ---- density detection (src/readtape.c:1656-1672 + estden_setdensity, src/decoder.c:374-399) ---- */
float rt_density_prepass(struct rt_dec *d, struct rt_reader *r, float *implied, int *nblks, int *hit_end) {
   *nblks = 0; *hit_end = 0; *implied = 0;
   d->doing_density_detection = 1;
   d->opt.bpi = 0;
   memset(&d->estden, 0, sizeof d->estden);
   d->interblock_counter = 0;
   do {
      rt_init_blockstate(d);
      d->parmset = 0;
      rt_init_trackstate(d);
      if (!r->readblock(r->ctx, 1)) { *hit_end = 1; break; }
      if (d->estden.fatal) break;
      if (d->results[d->parmset].blktype != RT_BS_NOISE) ++*nblks; }
   while (d->estden.totalcount < RT_ESTDEN_COUNTNEEDED);
   d->doing_density_detection = 0;
   d->interblock_counter = 0;
   if (d->estden.fatal) return -1;
   /* the smallest transition distance seen at least 5 % of the time (ESTDEN_MINPERCENT) */
   int mindist = INT_MAX;
   for (int i = 0; i < d->estden.binsused; ++i)
      if (d->estden.counts[i] > d->estden.totalcount * 5 / 100 && d->estden.deltas[i] < mindist) mindist = d->estden.deltas[i];
   float density = 1.0f / (d->opt.ips * (float)(mindist + 0.5f) * (float)0.5e-6);
   if (d->opt.mode == RT_PE) density /= 2;                     /* twice the transitions */
   *implied = density;
   static const float standard[] = { 200, 556, 800, 1600, 9042 };
   for (int i = 0; i < 5; ++i) {
      float diff = density - standard[i];
      if (diff < 0) diff = -diff;
      if (diff < standard[i] * 20 / 100) {                     /* ESTDEN_CLOSEPERCENT */
         d->opt.bpi = standard[i];
         char cb[32];
         rlog(d, "  density was set to %.0f BPI (%.2f usec/bit) after reading the first %d blocks and seeing %s transitions in %d bins that imply %.0f BPI\n",
              d->opt.bpi, 1e6 / (d->opt.bpi * d->opt.ips), *nblks, commas(d->estden.totalcount, cb), d->estden.binsused, density);
         return d->opt.bpi; } }
   return 0; }
