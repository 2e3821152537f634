// rtfe_chain.hip — the sparse half of the peak-record path: k_zones, k_chain, k_publish (see rtfe_peaks.hip).
// Included behind rtfe_kernels.hip (it reuses the walker state, the AGC mirror and the threshold code of the sample path).

namespace rtfe {

// ------------------------------------------------------------------------------------------------
// k_zones: where every burst restarts.  For a zone-started burst: per (window width, track) the last forced rescan
// (the sample leaving the window is its maximum and the entering one does not exceed it, src/decoder.c:763-767) inside the zone's last kMarginRows rows; a restart
// at or before (that row - W - max(trk, skew) - 2) has a full, regular window when the rescan happens.
// One wave per burst, a lane per (screen, track); the samples come straight from HBM (a zone's tail is 4.6 KB).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_zones(const DevCfg *__restrict__ cfgp, const int16_t *__restrict__ rows, long long nrows,
                                              const rtfe_burst *__restrict__ bursts, const BurstScratch *__restrict__ scratch, BurstCtl *__restrict__ ctl) {
   const DevCfg &cfg = *cfgp;
   const int ntrks = cfg.ntrks;
   const int nb = scratch->nbursts_total;
   for (int b = blockIdx.x; b < nb; b += gridDim.x) {
      const rtfe_burst B = bursts[b];
      long long reset;
      unsigned int bflags = B.flags;
      int status = kBurstReady;
      if (B.flags & RTFE_F_EXACT_START) { reset = B.reset_sample; status = kBurstNeedsFull; }      // the window fills on live signal: sample path
      else if (B.zone_end - B.zone_first < kMarginRows + 64) { reset = B.zone_end - kMarginRows; bflags |= RTFE_F_UNSAFE; status = kBurstNeedsFull; }
      else {
         const long long z0 = B.zone_end - kMarginRows;
         long long lo = 0x7fffffffffffffffll;
         for (int i = threadIdx.x; i < cfg.nscreens * ntrks; i += 64) {
            const int sc = i / ntrks, t = i - sc * ntrks;
            const int W = cfg.screen[sc].W, d = cfg.skew[t], col = cfg.trk_to_head[t];
            const int sgn = cfg.invert ? -1 : 1;
            long long a = -1;
            for (long long n = B.zone_end - 1; n >= z0; --n) {                 // detector row n reads sample n - d of the column
               const long long s = n - d - W;
               if (s < 0) break;
               const int v = sgn * (int)rows[s * ntrks + col];
               bool dom = true;
               for (int k = 1; k <= W; ++k) if (sgn * (int)rows[(s + k) * ntrks + col] > v) { dom = false; break; }      // (incl. the entering sample: src/decoder.c:763-767)
               if (dom) { a = n; break; } }
            long long hi = a < 0 ? -1 : a - W - max(t, d) - 2;
            if (hi < z0) hi = -1;
            lo = min(lo, hi); }
         #pragma unroll
         for (int o = 32; o >= 1; o >>= 1) {
            const int l2 = __shfl((int)(lo & 0xffffffffll), (threadIdx.x + o) & 63), h2 = __shfl((int)(lo >> 32), (threadIdx.x + o) & 63);
            const long long other = ((long long)h2 << 32) | (unsigned int)l2;
            lo = min(lo, other); }
         if (lo < 0 || lo < B.zone_first) { reset = B.zone_end - kMarginRows; bflags |= RTFE_F_UNSAFE; status = kBurstNeedsFull; }
         else reset = lo; }
      if (threadIdx.x == 0) {
         BurstCtl c; c.reset = reset; c.stop = 0; c.next_tile = 0; c.status = status; c.bflags = bflags; c.pad = 0;
         ctl[b] = c; } } }

// where burst b stops: the next burst's restart row, but no further than tail_rows into its quiet zone
__device__ __forceinline__ long long chain_stop(const DevCfg &cfg, const rtfe_burst *bursts, const BurstCtl *ctl, int b, int nb_total, long long nrows) {
   if (b + 1 >= nb_total) return nrows;
   long long stop = ctl[b + 1].reset;
   const long long zf = bursts[b + 1].zone_first;
   if (cfg.tail_rows > 0 && zf + cfg.tail_rows < stop) stop = zf + cfg.tail_rows;
   return stop; }

// ------------------------------------------------------------------------------------------------
// k_chain
// ------------------------------------------------------------------------------------------------
constexpr int kChRecCap = 384;        // records of one tile's list (own + spilled) a wave keeps in LDS
constexpr int kChEntCap = 2048;

struct Run {
   long long pos, f;                 // column rows
   int nlead, nsure, ntail, val, dprev, dnext, e0;
   bool top, unknown; };

__device__ __forceinline__ Run run_decode(const PeakRec r, long long tile0, int e0) {
   Run u;
   u.pos = tile0 - 64 + (long long)(r.w0 & 0x7ffu);
   u.top = !((r.w0 >> 11) & 1u);
   u.f = u.pos + (long long)((r.w0 >> 12) & 63u);
   u.nlead = (int)((r.w0 >> 18) & 15u);
   u.nsure = (int)((r.w0 >> 22) & 63u);
   u.ntail = (int)((r.w0 >> 28) & 15u);
   u.unknown = r.w1 == 0xffff8000u;
   if (u.nsure == 63) { u.nlead = u.nlead << 4 | u.ntail; u.nsure = 0; u.ntail = 0; }      // every row explicit
   u.val = (int)(int16_t)(r.w1 & 0xffffu);
   u.dprev = (int)((r.w1 >> 16) & 0xffu) - 1;
   u.dnext = (int)((r.w1 >> 24) & 0xffu) - 1;
   u.e0 = e0;
   return u; }

// the rise test of src/decoder.c:790-791 / 800-801 for a row whose margin is m (int16 code difference to the nearer edge)
__device__ __forceinline__ bool rise_pass(const Walker &w, const Run &u, int m, float mv) {
   if (m >= w.rise_hi) return true;
   if (m <= w.rise_lo) return false;
   return u.top ? volt(u.val, mv) > volt(u.val - m, mv) + w.rise : volt(u.val, mv) < volt(u.val + m, mv) - w.rise; }
// ... and its min_peak half (src/decoder.c:792, 802)
__device__ __forceinline__ bool amp_pass(const Walker &w, const Run &u, float mv) {
   if (w.reqmin == 0) return true;
   const int a = u.top ? u.val : -u.val;
   if (a >= w.min_hi) return true;
   if (a <= w.min_lo) return false;
   return u.top ? volt(u.val, mv) > w.reqmin : volt(u.val, mv) < -w.reqmin; }

constexpr long long kNoRow = 0x7fffffffffffffffll;
// first row >= c (and < limit) at which this run makes the detector fire, or kNoRow; doubt = first row >= c that the record cannot decide
__device__ __forceinline__ long long run_fire(const Walker &w, const Run &u, const uint16_t *ents, long long c, long long limit, int W, int sure_i, float mv, long long &doubt) {
   doubt = kNoRow;
   const long long last_row = u.pos + W - 2;                         // the owner is strictly inside the window up to here
   if (last_row < c || u.f >= limit) return kNoRow;
   if (!amp_pass(w, u, mv)) return kNoRow;
   if (u.unknown) { const long long n = max(c, u.f); if (n < u.f + u.nsure && n < limit) doubt = n; return kNoRow; }
   for (int i = 0; i < u.nlead; ++i) {
      const long long n = u.f + i;
      if (n < c) continue;
      if (n >= limit) return kNoRow;
      if (rise_pass(w, u, (int)ents[u.e0 + i], mv)) return n; }
   const long long s0 = u.f + u.nlead;
   if (u.nsure) {
      const long long n = max(c, s0);
      if (n < s0 + u.nsure) {
         if (n >= limit) return kNoRow;
         if (w.rise_hi > sure_i) { doubt = n; return kNoRow; }
         return n; } }
   for (int i = 0; i < u.ntail; ++i) {
      const long long n = s0 + u.nsure + i;
      if (n < c) continue;
      if (n >= limit) return kNoRow;
      if (rise_pass(w, u, (int)ents[u.e0 + u.nlead + i], mv)) return n; }
   return kNoRow; }

// ---- the parallel path of k_chain: up to 64 consecutive records of the window, one per lane ----------------------------------
// In steady state (baseline fixed, alpha-filter AGC) WHICH runs fire does not depend on the exact gain as long as every
// threshold of the stretch stays inside a band around the current one: a run with a row whose margin clears the band fires,
// one without a row that reaches the band does not, and the blind countdown ends when the fired run's owner leaves the window,
// whatever row it fired at (pos + W + 1).  So: (1) every lane classifies its run against the band, the blind rows follow from
// the fired lanes in front of it (iterated to a fixed point), (2) the gains follow from the fired runs' heights - three flops
// per detection, the only sequential part, (3) every lane re-evaluates its run with ITS exact thresholds: the exact firing row,
// and that every threshold stayed in the band.  Anything that does not fit (a margin or an amplitude inside the band with nothing
// clearer behind it, runs that could fire out of list order, a threshold leaving the band) returns -1 and the caller takes
// one sequential step instead.  Returns the number of detections (>= 0) after advancing c and w.
struct LaneRun { int pos, f, nlead, nsure, ntail, e0, val, dprev, dnext; bool top, unknown; };
constexpr int kNone = 0x3fffffff;

__device__ __forceinline__ int wave_excl_max(int v, int lane) {           // max over lanes in front of this one (kNone-free: -kNone = none)
   int x = __shfl_up(v, 1); if (lane == 0) x = -kNone;
   #pragma unroll
   for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o && y > x) x = y; }
   return x; }

// first row >= cmin and < L whose margin is >= hi.  maybe: a row in front of it (>= cmin) has a margin in (lo, hi), or lies in the sure
// stretch while the band reaches above the sure level.
__device__ __forceinline__ int lane_first(const LaneRun &u, const uint16_t *ents, int cmin, int L, int lo, int hi, bool sure_ok, bool &maybe) {
   maybe = false;
   for (int i = 0; i < u.nlead; ++i) {
      const int n = u.f + i;
      if (n < cmin) continue;
      if (n >= L) return kNone;
      const int m = ents[u.e0 + i];
      if (m >= hi) return n;
      if (m > lo) maybe = true; }
   const int s0 = u.f + u.nlead;
   if (u.nsure) {
      const int n = cmin > s0 ? cmin : s0;
      if (n < s0 + u.nsure) { if (n >= L) return kNone; if (sure_ok) return n; maybe = true; } }
   for (int i = 0; i < u.ntail; ++i) {
      const int n = s0 + u.nsure + i;
      if (n < cmin) continue;
      if (n >= L) return kNone;
      const int m = ents[u.e0 + u.nlead + i];
      if (m >= hi) return n;
      if (m > lo) maybe = true; }
   return kNone; }

__device__ __forceinline__ int chain_parallel(const DevCfg &cfg, const DevParm &P, const DevScreen &S, Walker &w, long long &c, long long tile0, long long wlimit,
                                              const PeakRec *recs, const uint16_t *eoff, const uint16_t *ents, int k0, int nrec, int lane,
                                              rtfe_event *ev, unsigned int cap, long long reset, int d, int trk, int pidx, float *ring, float *hv) {
   const int W = P.W;
   const float mv = cfg.maxvolts;
   const int n = nrec - k0 < 64 ? nrec - k0 : 64;
   const bool have = lane < n;
   LaneRun u = {};
   if (have) {
      const PeakRec r = recs[k0 + lane];
      u.pos = (int)(r.w0 & 0x7ffu) - 64; u.top = !((r.w0 >> 11) & 1u); u.f = u.pos + (int)((r.w0 >> 12) & 63u);
      u.nlead = (int)((r.w0 >> 18) & 15u); u.nsure = (int)((r.w0 >> 22) & 63u); u.ntail = (int)((r.w0 >> 28) & 15u);
      u.unknown = r.w1 == 0xffff8000u;
      if (u.nsure == 63) { u.nlead = u.nlead << 4 | u.ntail; u.nsure = 0; u.ntail = 0; }
      u.val = (int)(int16_t)(r.w1 & 0xffffu); u.dprev = (int)((r.w1 >> 16) & 0xffu) - 1; u.dnext = (int)((r.w1 >> 24) & 0xffu) - 1;
      u.e0 = eoff[k0 + lane]; }
   // rows relative to tile0.  L: rows below it are decided by this stretch (a record behind it cannot start earlier)
   long long Ll = wlimit - tile0;
   if (k0 + n < nrec) { const PeakRec rn = recs[k0 + n]; const long long fb = (long long)(rn.w0 & 0x7ffu) - 64 + (long long)((rn.w0 >> 12) & 63u) - W + 2; if (fb < Ll) Ll = fb; }
   const long long crel_l = c - tile0;
   if (Ll <= crel_l) return -1;
   const int L = Ll > kNone ? kNone : (int)Ll, c0 = crel_l < -4096 ? -4096 : (int)crel_l;
   // ---- the band ----
   const float rs = w.rise * cfg.lsb_per_volt, ms = w.reqmin * cfg.lsb_per_volt;
   const int b_lo = (int)floorf(rs * 0.8f) - 2, b_hi = (int)floorf(rs * 1.25f) + 3;
   const int a_lo = (int)floorf(ms * 0.8f) - 2, a_hi = (int)floorf(ms * 1.25f) + 3;
   const bool sure_ok = b_hi <= S.sure_i;
   const int amp = w.reqmin == 0 ? 2 : ((u.top ? u.val : -u.val) >= a_hi ? 2 : ((u.top ? u.val : -u.val) <= a_lo ? 0 : 1));
   // ---- (1) which runs fire ----
   bool fired = false, maybe = false;
   int ck = c0, nk = kNone;
   bool stable = false;
   for (int it = 0; it < 5 && !stable; ++it) {
      const int pf = wave_excl_max(fired ? u.pos : -kNone, lane);
      ck = pf > -kNone && pf + W + 1 > c0 ? pf + W + 1 : c0;
      bool mb = false;
      nk = have && !u.unknown && amp != 0 && u.pos + W - 2 >= ck ? lane_first(u, ents, ck, L, b_lo, b_hi, sure_ok, mb) : kNone;
      maybe = mb;
      const bool nf = nk != kNone;
      stable = __ballot(nf != fired) == 0;
      fired = nf; }
   if (!stable) return -1;
   // undecidable inside the band: a margin or the amplitude between the band's edges with nothing clearer behind it; an unknown minimum in range
   const bool in_range = have && u.pos + W - 2 >= ck && u.f < L;
   bool bad = false;
   if (in_range) {
      if (u.unknown) bad = (ck > u.f ? ck : u.f) < u.f + u.nsure;
      else if (amp == 1) { bool mb; const int t = lane_first(u, ents, ck, L, b_lo, b_lo + 1, true, mb); bad = t != kNone; }      // any row above the band's lower edge
      else if (amp == 2 && !fired) bad = maybe; }
   // list order = firing order: owners and firing rows increase along the fired lanes, and no run starts at or before a detection in front of it
   const int pfx = wave_excl_max(fired ? u.pos : -kNone, lane);
   if (fired && pfx > -kNone && u.pos <= pfx) bad = true;
   if (__ballot(bad)) return -1;
   const u64 fm = __ballot(fired);
   const int nf = __popcll(fm);
   if (nf == 0) { c = tile0 + L > c ? tile0 + L : c; return 0; }
   if (w.nevents + (unsigned)nf > cap) return -1;
   // ---- (2) gains: v_lasttop / v_lastbot as each callback finds them (one event late, src/decoder.c:587-590), then the recurrence ----
   const float val = volt(u.val, mv);
   float lt = 0, lb = 0; bool ht = false, hb = false;                      // last fired top / bottom value at or before this lane
   if (fired) { if (u.top) { lt = val; ht = true; } else { lb = val; hb = true; } }
   #pragma unroll
   for (int o = 1; o < 64; o <<= 1) {
      const float yt = __uint_as_float((unsigned)__shfl_up((int)__float_as_uint(lt), o)), yb = __uint_as_float((unsigned)__shfl_up((int)__float_as_uint(lb), o));
      const int yh = __shfl_up((ht ? 1 : 0) | (hb ? 2 : 0), o);
      if (lane >= o) { if (!ht && (yh & 1)) { lt = yt; ht = true; } if (!hb && (yh & 2)) { lb = yb; hb = true; } } }
   // exclusive: what the lane in front holds
   float plt = __uint_as_float((unsigned)__shfl_up((int)__float_as_uint(lt), 1)), plb = __uint_as_float((unsigned)__shfl_up((int)__float_as_uint(lb), 1));
   int ph = __shfl_up((ht ? 1 : 0) | (hb ? 2 : 0), 1);
   if (lane == 0) ph = 0;
   const float lasttop = (ph & 1) ? plt : w.v_lasttop, lastbot = (ph & 2) ? plb : w.v_lastbot;
   const float lastheight = lasttop - lastbot;
   const bool adj = fired && !cfg.agc_off && lastheight > 0;
   float g = w.agc_gain, gbefore = g;
   const u64 am = __ballot(adj);
   const int NW = P.agc_window;
   if (NW == 0) {                                                           // exponential AGC: g = alpha (h / lastheight) + (1 - alpha) g, clamped (src/decoder.c:505-512)
      const float a = adj ? P.agc_alpha * (w.v_avg_height / lastheight) : 0.0f;
      const float beta = 1 - P.agc_alpha;
      for (u64 m = fm; m; m &= m - 1) {
         const int i = __ffsll((long long)m) - 1;
         if (lane == i) gbefore = g;
         if ((am >> i) & 1ull) {
            const float ai = __uint_as_float((unsigned)__shfl((int)__float_as_uint(a), i));
            float g2 = ai + beta * g;
            if (g2 > 2.0f) g2 = 2.0f;
            g = g2; } } }
   else {                                                                    // window AGC: g = h / min(last NW heights), clamped (src/decoder.c:516-529): no recurrence at all
      const int nv = __popcll(am), jv = __popcll(am & ((1ull << lane) - 1));
      __syncthreads();
      if (adj) hv[jv] = lastheight;
      __syncthreads();
      float ga = 0; bool hg = false;
      if (adj) {
         float mn = lastheight;
         for (int i = 1; i < NW; ++i) {
            const int idx = jv - i;                                          // the i-th height before this one: of this stretch, or from the ring as it stood
            const float v = idx >= 0 ? hv[idx] : ring[(w.heightndx + NW + idx) % NW];
            if (v < mn) mn = v; }
         ga = w.v_avg_height / mn;
         if (ga > 2.0f) ga = 2.0f;
         hg = true; }
      // gain in force when each detection's callback starts = the one the last adjustment in front of it left
      float xg = ga; bool xh = hg;
      #pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
         const float yg = __uint_as_float((unsigned)__shfl_up((int)__float_as_uint(xg), o));
         const int yh = __shfl_up(xh ? 1 : 0, o);
         if (lane >= o && !xh && yh) { xg = yg; xh = true; } }
      const float pg = __uint_as_float((unsigned)__shfl_up((int)__float_as_uint(xg), 1));
      int ph2 = __shfl_up(xh ? 1 : 0, 1);
      if (lane == 0) ph2 = 0;
      gbefore = ph2 ? pg : w.agc_gain;
      const float fg = __uint_as_float((unsigned)__shfl((int)__float_as_uint(xg), 63));
      g = __shfl(xh ? 1 : 0, 63) ? fg : w.agc_gain;
      // (the ring itself is brought up to date at the commit)
      (void)nv; }
   if (!(g > 0)) return -1;
   // ---- (3) exact thresholds per detection: the firing row, and that the band held ----
   Walker wk = w;
   wk.agc_gain = fired ? gbefore : g; wk.flags = 0;
   update_thresholds(wk, P, cfg.lsb_per_volt);
   // (lanes that did not fire evaluate the thresholds behind the last detection: the rows up to L were decided against the band, too)
   bool viol = wk.rise_lo < b_lo || wk.rise_hi > b_hi || (w.reqmin != 0 && (wk.min_lo < a_lo || wk.min_hi > a_hi)) || (wk.flags & RTFE_F_SCREEN_UNDERFLOW);
   int nx = kNone;
   if (fired) {
      // rows in front of nk have margins at or below the band (fail for every threshold in it) or inside it (decided here, exactly)
      Run r2; r2.top = u.top; r2.val = u.val;
      for (int i = 0; i < u.nlead && nx == kNone; ++i) { const int row = u.f + i; if (row < ck) continue; if (row > nk) break; if (rise_pass(wk, r2, (int)ents[u.e0 + i], mv)) nx = row; }
      const int s0 = u.f + u.nlead;
      if (nx == kNone && u.nsure) { const int row = ck > s0 ? ck : s0; if (row < s0 + u.nsure && row <= nk) { if (wk.rise_hi > S.sure_i) viol = true; nx = row; } }
      for (int i = 0; i < u.ntail && nx == kNone; ++i) { const int row = s0 + u.nsure + i; if (row < ck) continue; if (row > nk) break; if (rise_pass(wk, r2, (int)ents[u.e0 + u.nlead + i], mv)) nx = row; }
      if (nx == kNone) viol = true;
      const int ti = (int)floorf(0.005f / gbefore * cfg.lsb_per_volt);
      if (ti + 2 > 254) viol = true; }
   // order: a run with a row that reaches the band, at or before a detection in front of it in the list, might have fired first
   const int pn = wave_excl_max(fired ? nx : -kNone, lane);
   if (have && amp != 0 && !u.unknown && pn > -kNone && u.f <= pn) {
      bool mb; const int t = lane_first(u, ents, c0, L, b_lo, b_lo + 1, true, mb);
      if (t <= pn) viol = true; }
   if (__ballot(viol)) return -1;
   // ---- commit ----
   if (fired) {
      const int eidx = __popcll(fm & ((1ull << lane) - 1));
      const long long nrow = tile0 + nx, ndet = nrow + d;
      const int ld = u.pos - nx + W;
      const int iprev = u.top ? u.val - u.dprev : u.val + u.dprev, inext = u.top ? u.val - u.dnext : u.val + u.dnext;
      const int adjcode = refine_code(&cfg, u.val, iprev, inext, gbefore, u.top);
      rtfe_event e;
      e.sample = (uint32_t)(ndet - reset);
      e.v_peak = (cfg.invert && val == 0.0f) ? -0.0f : val;
      e.agc_gain = gbefore;
      e.trk = (uint8_t)trk;
      e.flags = (uint8_t)((u.top ? 0 : 1) | (adjcode << 1));
      e.left_distance = (uint8_t)ld;
      e.parmset = (uint8_t)pidx;
      ev[w.nevents + eidx] = e; }
   // state behind the last detection (lane 63 holds the inclusive scans; the last fired lane its row)
   const int lastl = 63 - __clzll((long long)fm);
   const int lastpos = __shfl(u.pos, lastl);
   const float flt = __uint_as_float((unsigned)__shfl((int)__float_as_uint(lt), 63)), flb = __uint_as_float((unsigned)__shfl((int)__float_as_uint(lb), 63));
   const int fh = __shfl((ht ? 1 : 0) | (hb ? 2 : 0), 63);
   if (fh & 1) { w.v_lasttop = flt; w.v_top = flt; }
   if (fh & 2) { w.v_lastbot = flb; w.v_bot = flb; }
   if (NW > 0) {                                                             // the ring: the last heights of the stretch, where the one-at-a-time walk would have left them
      const int nv = __popcll(am);
      for (int t = 0; t < nv && t < NW; ++t) ring[(w.heightndx + nv - 1 - t) % NW] = hv[nv - 1 - t];
      w.heightndx = (w.heightndx + nv) % NW; }
   w.agc_gain = g;
   w.peakcount += nf;
   w.nevents += (unsigned)nf;
   update_thresholds(w, P, cfg.lsb_per_volt);
   long long cn = tile0 + lastpos + W + 1;
   if (tile0 + L > cn) cn = tile0 + L;
   c = cn;
   return nf; }

__global__ void __launch_bounds__(64) k_chain(const DevCfg *__restrict__ cfgp, long long nrows, long long row_base,
                                              const rtfe_burst *__restrict__ bursts, BurstScratch *__restrict__ scratch, BurstCtl *__restrict__ ctl,
                                              uint32_t *__restrict__ counts, rtfe_event *__restrict__ events,
                                              const PeakDir *__restrict__ dir_main, const PeakDir *__restrict__ dir_spill,
                                              const unsigned char *__restrict__ pool_own, const unsigned char *__restrict__ pool_spill, long long ntiles) {
   __shared__ PeakRec s_recs[kChRecCap];
   __shared__ uint16_t s_eoff[kChRecCap];
   __shared__ uint16_t s_ents[kChEntCap];
   __shared__ float s_heights[64 * 10];
   __shared__ float s_hv[64];
   const DevCfg &cfg = *cfgp;
   const int ntrks = cfg.ntrks, nwalk = cfg.nparm * ntrks;
   const int lane = threadIdx.x;
   const float mv = cfg.maxvolts;
   unsigned int g_outer = 0;
   for (;;) {
      if (++g_outer > 100000u) { break; }
      int idx = 0;
      if (lane == 0) idx = atomicAdd(&scratch->queue_walk, 1);
      idx = __shfl(idx, 0);
      const int b = idx / nwalk;
      if (b >= scratch->nbursts) break;
      const int wi = idx - b * nwalk, pidx = wi / ntrks, trk = wi - pidx * ntrks;
      if (ctl[b].status != kBurstReady) continue;
      const rtfe_burst B = bursts[b];
      const DevParm &P = cfg.parm[pidx];
      const DevScreen &S = cfg.screen[P.screen];
      const int W = P.W, d = cfg.skew[trk], head = cfg.trk_to_head[trk];
      const long long reset = ctl[b].reset;
      const long long stop = chain_stop(cfg, bursts, ctl, b, scratch->nbursts_total, nrows);
      Walker w = {};
      w.agc_gain = 1.0f; w.v_avg_height = 4.0f;
      update_thresholds(w, P, cfg.lsb_per_volt);
      for (int i = 0; i < 10; ++i) s_heights[lane * 10 + i] = 0;
      // column rows: the detector's row n reads sample n - d.  Before fast_from the window is filling on zone samples only.
      long long c = reset + W + max(trk, d) + 1 - d;
      const long long limit = stop - d;
      rtfe_event *ev = events + B.event_base + (size_t)(pidx * ntrks + trk) * B.event_cap;
      const unsigned int cap = B.event_cap;
      bool failed = false;
      int why = 0;
      unsigned int n_par = 0, n_seq = 0, guard = 0;                                   // detections decided by all lanes at once / one at a time (statistics)
      for (long long g = c / kPkTile; g * kPkTile < limit && g < ntiles && !failed; ++g) {
         const long long tile0 = g * kPkTile;
         // ---- the tile's lists: what the previous tile spilled, then its own, as one list ordered by the row of the sample that
         // made each record (every record's first row lies behind that sample: a record further down the list cannot start
         // more than W - 2 rows before this one) ----
         const PeakDir ds = dir_spill[(g * cfg.nscreens + P.screen) * ntrks + head], dm = dir_main[(g * cfg.nscreens + P.screen) * ntrks + head];
         if (ds.nrec >= 0xfffe || dm.nrec >= 0xfffe) { failed = true; why = 1; break; }                  // (capacity, or a quiet tile nobody was expected to need)
         const int nrec_all = (int)ds.nrec + dm.nrec;
         if (nrec_all == 0) continue;
         // (a slot holds its records at the front and its margin entries from the back: entry e at end[-(e + 1)])
         const size_t li = (size_t)(g * cfg.nscreens + P.screen) * ntrks + head;
         const PeakRec *r0 = reinterpret_cast<const PeakRec *>(pool_spill + li * cfg.pk_sslot), *r1 = reinterpret_cast<const PeakRec *>(pool_own + li * cfg.pk_slot);
         const uint16_t *e0 = reinterpret_cast<const uint16_t *>(pool_spill + (li + 1) * cfg.pk_sslot), *e1 = reinterpret_cast<const uint16_t *>(pool_own + (li + 1) * cfg.pk_slot);
         int base = 0, ent_base = 0;                                          // the window of the list held in LDS starts here
         unsigned int g_win = 0;
         while (base < nrec_all && !failed) {
         if (++g_win > 5000u) { failed = true; why = 7; break; }
         int nrec = min(kChRecCap, nrec_all - base);
         __syncthreads();
         for (int i = lane; i < nrec; i += 64) { const int j = base + i; s_recs[i] = j < (int)ds.nrec ? r0[j] : r1[j - ds.nrec]; }
         __syncthreads();
         int carry = 0;
         for (int b0 = 0; b0 < nrec; b0 += 64) {                             // first entry of every record of the window
            const int i = b0 + lane;
            const int v = i < nrec ? pk_nent(s_recs[i].w0, s_recs[i].w1) : 0;
            int incl = v;
            #pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
            if (i < nrec) s_eoff[i] = (uint16_t)(carry + incl - v);
            carry += __shfl(incl, 63); }
         __syncthreads();
         if (carry > kChEntCap) {                                              // keep the records whose margins fit
            int keep = 0;
            while (keep < nrec && (int)s_eoff[keep] + pk_nent(s_recs[keep].w0, s_recs[keep].w1) <= kChEntCap) ++keep;
            nrec = keep; carry = nrec ? (int)s_eoff[nrec - 1] + pk_nent(s_recs[nrec - 1].w0, s_recs[nrec - 1].w1) : 0; }
         for (int i = lane; i < carry; i += 64) { const int j = ent_base + i; s_ents[i] = j < (int)ds.nent ? e0[-(j + 1)] : e1[-(j - (int)ds.nent + 1)]; }
         __syncthreads();
         // rows this window decides: up to the earliest row a record behind it could start at
         long long wlimit = limit < tile0 + kPkTile ? limit : tile0 + kPkTile;      // (a list holds rows of its tile only)
         if (base + nrec < nrec_all) {
            const int j = base + nrec;
            const PeakRec rn = j < (int)ds.nrec ? r0[j] : r1[j - ds.nrec];
            const long long fb = run_decode(rn, tile0, 0).f - W + 2;
            if (fb < wlimit) wlimit = fb; }
         // ---- the sequential walk (every lane runs it; lane 0 stores): earliest firing run among the tops and the bottoms ----
         int alive = 0;
         for (;;) {
            if (++guard > 200000u) { failed = true; why = 7; break; }           // (cannot happen: every round moves c forward; a safety net against spinning on a GPU)
            while (alive < nrec && run_decode(s_recs[alive], tile0, 0).pos + W - 2 < c) ++alive;
            if (alive >= nrec) break;
            // clean stretches: all lanes at once (steady state of the block decoder's AGC schedule, alpha filter)
            const bool steady = cfg.agc_off || (cfg.mode == RTFE_PE ? w.datablock : (w.peakcount > 15 && w.v_avg_height_count == 0));
            if (cfg.pk_parallel && steady) {
               const int got = chain_parallel(cfg, P, S, w, c, tile0, wlimit, s_recs, s_eoff, s_ents, alive, nrec, lane, ev, cap, reset, d, trk, pidx, s_heights + lane * 10, s_hv);
               if (got >= 0) { n_par += (unsigned)got; if (w.flags & RTFE_F_SCREEN_UNDERFLOW) { failed = true; why = 5; break; } continue; } }
            long long best = kNoRow, best_doubt = kNoRow;
            int best_k = -1;
            bool best_top = false, top_done = false, bot_done = false;
            for (int k = alive; k < nrec && !(top_done && bot_done); ++k) {
               const Run u = run_decode(s_recs[k], tile0, s_eoff[k]);
               if (u.top ? top_done : bot_done) continue;
               if (u.f > best || u.f >= wlimit) { if (u.top) top_done = true; else bot_done = true; continue; }
               long long dr;
               const long long n = run_fire(w, u, s_ents, c, wlimit, W, S.sure_i, mv, dr);
               if (dr < best_doubt) { best_doubt = dr;
#ifdef RTFE_CPU_EMUL
                  if (lane == 0 && getenv("RTFE_PK_DEBUG")) fprintf(stderr, "  doubt row %lld rec k %d pos %lld f %lld top %d nlead %d nsure %d ntail %d unknown %d val %d c %lld\n", dr, k, u.pos, u.f, (int)u.top, u.nlead, u.nsure, u.ntail, (int)u.unknown, u.val, c);
#endif
               }
               if (n != kNoRow) {
                  if (u.top) top_done = true; else bot_done = true;         // (runs of one kind are ordered by row)
                  if (n < best || (n == best && u.top && !best_top)) { best = n; best_k = k; best_top = u.top; } } }
            if (best_doubt != kNoRow && best_doubt <= best) { failed = true; why = 2; break; }
            if (best_k < 0) break;
            // ---- detection: refine_peak + the callback's effect on AGC state (src/decoder.c:700-749, 574-609) ----
            const Run u = run_decode(s_recs[best_k], tile0, s_eoff[best_k]);
            const long long ndet = best + d;                                 // the detector's row
            const int ld = (int)(u.pos - best) + W;                           // left_distance
            const float g = w.agc_gain;
            const float thr = 0.005f / g;
            const int ti = (int)floorf(thr * cfg.lsb_per_volt);
            if (ti + 2 > 254) { failed = true; why = 3; break; }                       // (neighbour distances are stored up to 254)
            const int val_i = u.val;
            const int iprev = u.top ? val_i - u.dprev : val_i + u.dprev, inext = u.top ? val_i - u.dnext : val_i + u.dnext;
            const int adjcode = refine_code(&cfg, val_i, iprev, inext, g, u.top);
            const float val = volt(val_i, mv);
            double t_peak = 0;
            if (cfg.mode == RTFE_PE && !w.datablock && w.peakcount >= 68) {
               const float adj = adjcode == 1 ? -0.5f : (adjcode == 2 ? 0.5f : 0.0f);
               t_peak = time_of(&cfg, row_base + ndet) - ((float)(W - ld) - adj) * cfg.sample_deltat; }
            if (w.nevents >= cap) w.flags |= RTFE_F_EVENT_OVERFLOW;
            else if (lane == 0) {
               rtfe_event e;
               e.sample = (uint32_t)(ndet - reset);
               e.v_peak = (cfg.invert && val == 0.0f) ? -0.0f : val;
               e.agc_gain = g;
               e.trk = (uint8_t)trk;
               e.flags = (uint8_t)((u.top ? 0 : 1) | (adjcode << 1));
               e.left_distance = (uint8_t)ld;
               e.parmset = (uint8_t)pidx;
               ev[w.nevents] = e; }
            if (u.top) w.v_top = val; else w.v_bot = val;
            ++w.nevents; ++n_seq;
            agc_after_peak(w, &cfg, P, s_heights + lane * 10, u.top, t_peak);      // (every lane keeps the same state; the window AGC's ring is per lane)
            if (!(w.agc_gain > 0)) { w.flags |= RTFE_F_DETECTOR_FATAL; failed = true; why = 4; break; }      // src/decoder.c:782
            update_thresholds(w, P, cfg.lsb_per_volt);
            if (w.flags & RTFE_F_SCREEN_UNDERFLOW) { failed = true; why = 5; break; }
            c = u.pos + W + 1; }
         if (failed) break;
         // ---- next window: everything in front of wlimit is decided ----
         if (base + nrec >= nrec_all) break;
         if (c < wlimit) c = wlimit;
         int adv = 0;
         while (adv < nrec && run_decode(s_recs[adv], tile0, 0).pos + W - 2 < c) ++adv;
         if (adv == 0) { failed = true; why = 6; break; }                      // (hundreds of live records inside two window lengths: not a tape)
         ent_base += adv < nrec ? (int)s_eoff[adv] : carry;
         base += adv; } }
      // ---- publish ----
#ifdef RTFE_CPU_EMUL
      if (lane == 0 && failed && getenv("RTFE_PK_DEBUG")) fprintf(stderr, "chain b %d p %d t %d failed why %d at c %lld (rise %f hi %d sure %d)\n", b, pidx, trk, why, c, w.rise, w.rise_hi, S.sure_i);
#endif
      if (lane == 0) {
         if (n_par) atomicAdd(&scratch->dbg[0], (unsigned long long)n_par);
         if (n_seq) atomicAdd(&scratch->dbg[1], (unsigned long long)n_seq);
         if (failed) { atomicExch(&ctl[b].status, (int)kBurstNeedsFull); atomicAdd(&scratch->why[why & 7], 1ull); }
         counts[((size_t)b * cfg.nparm + pidx) * ntrks + trk] = w.nevents < cap ? w.nevents : cap;
         if (w.flags & ~(unsigned)RTFE_F_SCREEN_UNDERFLOW) atomicOr(&ctl[b].bflags, w.flags & ~(unsigned)RTFE_F_SCREEN_UNDERFLOW); } } }

// ------------------------------------------------------------------------------------------------
// k_publish: burst table entries of the bursts the chains finished; stop rows for the ones the sample path redoes
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_publish(const DevCfg *__restrict__ cfgp, long long nrows, rtfe_burst *__restrict__ bursts,
                                                 BurstScratch *__restrict__ scratch, BurstCtl *__restrict__ ctl) {
   const DevCfg &cfg = *cfgp;
   const int nb = scratch->nbursts;
   for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
      const long long stop = chain_stop(cfg, bursts, ctl, b, scratch->nbursts_total, nrows);
      ctl[b].stop = stop;
      if (ctl[b].status == kBurstReady) {
         bursts[b].reset_sample = ctl[b].reset;
         bursts[b].safe_last = ctl[b].reset;
         bursts[b].end_sample = stop < nrows ? stop : nrows;
         bursts[b].flags = ctl[b].bflags;
         ctl[b].status = kBurstDone; }
      else atomicAdd(&scratch->seg_failed, 1); }                    // (statistics: bursts the sample path redoes)
   if (blockIdx.x == 0 && threadIdx.x == 0) scratch->queue_resume = 0; }

}  // namespace rtfe
