// rtfe_ww.hip — k_ww: the peak detector for Whirlwind tapes, with detector state that is handed in and out.
//
// Whirlwind blocks can be one bit apart, so the reference never restarts its detector between them (src/readtape.c:1674,
// src/decode_ww.c:31-49): the peak window's ring, its extremes, the blind countdown and the AGC state all carry over.  What a
// new block attempt does do is zero t_lastpeak, and process_sample (src/decoder.c:855-861) then RE-SEEDS each track - one track
// per sample, in track order, the later tracks sitting out until their turn - by overwriting ring slot 0 with the current sample
// and setting the window's extremes to it, while the ring's indices and its other slots keep what they held.  From then on the
// window is not the last W samples any more, and WHERE an attempt starts is the host decoder's decision (its clock average
// decides when the clock has stopped).  So this kernel is the literal detector - ring, stale extremes and all - run by one lane
// per track over the rows it is given, from a state blob the host hands in, to a state blob it hands back:
//      rtfe_ww_scan(first_row, nrows, seed_row0, state_in) -> events of [first_row, first_row + nrows), state after the last row.
// The host replay asks for rows in chunks, and - when its decoder ends a block at row r - for the state after r (a second, shorter
// scan of the same chunk), which is where the next attempt starts.  The tapes are short (100 BPI) and there is one chain per tape:
// this path is exact, not fast.  Included behind rtfe_kernels.hip (volt, refine_code, agc_after_peak, update_thresholds).

namespace rtfe {

constexpr int kWwRing = 64;           // >= RT_PKWW_MAX_WIDTH (50)

__global__ void __launch_bounds__(64) k_ww(const DevCfg *__restrict__ cfgp, const int16_t *__restrict__ rows, long long nrows_total, long long row_base,
                                          long long first_row, long long nscan, long long seed_row0,
                                          const rtfe_ww_track *__restrict__ state_in, rtfe_ww_track *__restrict__ state_out,
                                          uint32_t *__restrict__ counts, rtfe_event *__restrict__ events, long long cap, unsigned int *__restrict__ flags_out) {
   __shared__ short s_ring[RTFE_MAXTRKS][kWwRing];
   __shared__ float s_heights[RTFE_MAXTRKS][10];
   const DevCfg &cfg = *cfgp;
   const int t = threadIdx.x;
   if (t >= cfg.ntrks) return;
   const DevParm &P = cfg.parm[0];
   const int W = P.W;
   const float mv = cfg.maxvolts;
   const int col = cfg.trk_to_head[t];
   const int sgn = cfg.invert ? -1 : 1;
   rtfe_ww_track S = state_in[t];
   short *ring = s_ring[t];
   for (int i = 0; i < kWwRing; ++i) ring[i] = S.ring[i];
   float *heights = s_heights[t];
   for (int i = 0; i < 10; ++i) heights[i] = S.heights[i];
   Walker w = {};
   w.agc_gain = S.agc_gain; w.v_avg_height = S.v_avg_height; w.v_lasttop = S.v_lasttop; w.v_lastbot = S.v_lastbot;
   w.v_top = S.v_top; w.v_bot = S.v_bot; w.peakcount = S.peakcount; w.heightndx = S.heightndx;
   int left = S.left, right = S.right, maxv = S.maxv, minv = S.minv, countdown = S.countdown;
   const int delay = S.delay < 0 ? 0 : (S.delay > 50 ? 50 : S.delay);      // (MAXSKEWSAMP)
   unsigned int nev = 0, fl = 0;
   rtfe_event *out = events + (size_t)t * cap;
   const long long end = first_row + nscan < nrows_total ? first_row + nscan : nrows_total;
   for (long long n = first_row; n < end && !(fl & (RTFE_F_DETECTOR_FATAL | RTFE_F_AGC_FATAL)); ++n) {
      if (n < seed_row0 + t) continue;                               // the tracks in front of this one are being re-seeded: it sits the row out
      const long long src = (row_base + n < delay || n < delay) ? n : n - delay;    // the deskew FIFO runs from the tape's first row: undelayed until it has filled (src/decoder.c:820-830)
      const int v = sgn * (int)rows[src * cfg.ntrks + col];
      if (n == seed_row0 + t) {                                      // src/decoder.c:855-861: slot 0, both extremes; indices and the other slots stay
         ring[0] = (short)v; maxv = minv = v;
         continue; }
      // ---- lookfor_peak, src/decoder.c:751-810 ----
      int old_left = 0;
      if (++right >= W) right = 0;
      if (right == left) { old_left = ring[left]; if (++left >= W) left = 0; }
      ring[right] = (short)v;
      if (v > maxv) maxv = v;
      if (old_left == maxv || old_left == minv) {                    // (the reference compares floats: 0.0f == volt(0), and == on volts is == on codes)
         int mx = -0x7fffffff, mn = 0x7fffffff;
         for (int ndx = left;;) {
            const int u = ring[ndx];
            mx = max(mx, u); mn = min(mn, u);
            if (ndx == right) break;
            if (++ndx >= W) ndx = 0; }
         maxv = mx; minv = mn; }
      if (countdown) { --countdown; continue; }
      if (!(w.agc_gain > 0)) {                                       // src/decoder.c:782: fatal in the reference; the marker tells the replay where
         fl |= RTFE_F_AGC_FATAL;
         if (nev < cap) { rtfe_event e = {}; e.sample = (uint32_t)(n - first_row); e.trk = (uint8_t)t; e.flags = RTFE_EV_FATAL; out[nev] = e; }
         ++nev;
         break; }
      const float rise = P.rise * (w.v_avg_height / 4.0f) / w.agc_gain;
      const float reqmin = P.min_peak * (w.v_avg_height / 4.0f) / w.agc_gain;
      const float vl = volt(ring[left], mv), vr = volt(ring[right], mv);
      const float vmax = volt(maxv, mv), vmin = volt(minv, mv);
      const bool top = vmax > vl + rise && vmax > vr + rise && (reqmin == 0 || vmax > reqmin);
      const bool bot = !top && vmin < vl - rise && vmin < vr - rise && (reqmin == 0 || vmin < -reqmin);
      if (!top && !bot) continue;
      // ---- refine_peak, src/decoder.c:700-749: the first window element equal to the extreme, its neighbours ----
      const int val = top ? maxv : minv;
      int ld = 1, ndx = left, prev = -1;
      bool found = false;
      for (;;) {
         if (ring[ndx] == val) { found = true; break; }
         if (ndx == right) break;
         prev = ndx;
         ++ld;
         if (++ndx >= W) ndx = 0; }
      if (!found || prev < 0 || !(ld < W)) { fl |= RTFE_F_DETECTOR_FATAL; break; }         // src/decoder.c:709-710, 748
      int nxt = ndx + 1; if (nxt >= W) nxt = 0;
      const int adjcode = refine_code(&cfg, val, ring[prev], ring[nxt], w.agc_gain, top);
      if (nev < cap) {
         rtfe_event e;
         e.sample = (uint32_t)(n - first_row);
         const float vp = volt(val, mv);
         e.v_peak = (cfg.invert && vp == 0.0f) ? -0.0f : vp;
         e.agc_gain = w.agc_gain;
         e.trk = (uint8_t)t;
         e.flags = (uint8_t)((top ? 0 : 1) | (adjcode << 1));
         e.left_distance = (uint8_t)ld;
         e.parmset = 0;
         out[nev] = e; }
      else fl |= RTFE_F_EVENT_OVERFLOW;
      ++nev;
      if (top) w.v_top = volt(val, mv); else w.v_bot = volt(val, mv);
      agc_after_peak(w, &cfg, P, heights, top, 0.0);                 // Whirlwind: every pulse edge adjusts the gain (src/decode_ww.c:175,194)
      countdown = ld; }                                              // src/decoder.c:741
   // ---- the state after the last row ----
   for (int i = 0; i < kWwRing; ++i) S.ring[i] = ring[i];
   for (int i = 0; i < 10; ++i) S.heights[i] = heights[i];
   S.left = left; S.right = right; S.maxv = maxv; S.minv = minv; S.countdown = countdown;
   S.agc_gain = w.agc_gain; S.v_avg_height = w.v_avg_height; S.v_lasttop = w.v_lasttop; S.v_lastbot = w.v_lastbot;
   S.v_top = w.v_top; S.v_bot = w.v_bot; S.peakcount = w.peakcount; S.heightndx = w.heightndx;
   state_out[t] = S;
   counts[t] = nev < cap ? nev : (unsigned int)cap;
   if (fl) atomicOr(flags_out, fl); }

}  // namespace rtfe
