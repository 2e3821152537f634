// rtfe_gain.hip — the sparse half of the peak path: k_zones, k_gain, k_emit, k_publish (the dense half: rtfe_sift.hip).
// Included behind rtfe_kernels.hip (it reuses the walker state, the AGC mirror and the threshold code of the sample path).
//
//   k_zones   per burst: the restart row inside its quiet zone from the forced rescans of the zone's last 256 rows.
//   k_gain    one LANE per (burst, parameter set, track): the blind countdown, the AGC schedule of the block decoders
//             (src/decode_nrzi.c:196-229, src/decode_gcr.c:843-864, src/decode_pe.c:127-198) and the thresholds they feed
//             (src/decoder.c:785-786), over k_sift's records, read as one sequential stream per lane.  What is sequential is
//             small: which record fires, and g = a + (1 - alpha) g.  In steady state a record with a sure stretch that nothing
//             else can interfere with takes the fast path: the lane notes the gain and moves on; WHICH row fired, the
//             half-sample refinement and the volt conversion are left to
//   k_emit    one lane per event: finishes the events the fast path only noted (exact thresholds from the noted gain, the
//             record's margins, refine_peak's neighbours), 16 bytes in, 16 bytes out.
//   k_publish burst table entries of the bursts the chains finished.

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
            // detector row n reads sample n - d of the column.  Walk back keeping the running maximum of the W samples behind the
            // candidate: the sample leaving at row n (s = n - d - W) forces a rescan iff it is >= all of them (incl. the entering one)
            for (long long n = B.zone_end - 1; n >= z0; --n) {
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
// k_gain
// ------------------------------------------------------------------------------------------------
struct Run {
   long long pos, f;                 // column rows
   int nlead, nsure, ntail, val, dprev, dnext, e0;
   bool top, unknown; };

__device__ __forceinline__ Run run_decode(uint32_t w0, uint32_t w1, long long tile0, int e0) {
   Run u;
   u.pos = tile0 - kSfPosBias + (long long)(w0 & 0x7ffu);
   u.top = !((w0 >> 11) & 1u);
   u.f = u.pos + (long long)((w0 >> 12) & 63u);
   u.nlead = (int)((w0 >> 18) & 15u);
   u.nsure = (int)((w0 >> 22) & 63u);
   u.ntail = (int)((w0 >> 28) & 15u);
   u.unknown = w1 == 0xffff8000u;
   if (u.nsure == 63) { u.nlead = u.nlead << 4 | u.ntail; u.nsure = 0; u.ntail = 0; }      // every row explicit
   u.val = (int)(int16_t)(w1 & 0xffffu);
   u.dprev = (int)((w1 >> 16) & 0xffu) - 1;
   u.dnext = (int)((w1 >> 24) & 0xffu) - 1;
   u.e0 = e0;
   return u; }

// the rise test of src/decoder.c:790-791 / 800-801 for a row whose margin is m (int16 code difference to the nearer edge)
__device__ __forceinline__ bool rise_pass(const Walker &w, bool top, int val, int m, float mv) {
   if (m >= w.rise_hi) return true;
   if (m <= w.rise_lo) return false;
   return top ? volt(val, mv) > volt(val - m, mv) + w.rise : volt(val, mv) < volt(val + m, mv) - w.rise; }
// ... and its min_peak half (src/decoder.c:792, 802)
__device__ __forceinline__ bool amp_pass(const Walker &w, const Run &u, float mv) {
   if (w.reqmin == 0) return true;
   const int a = u.top ? u.val : -u.val;
   if (a >= w.min_hi) return true;
   if (a <= w.min_lo) return false;
   return u.top ? volt(u.val, mv) > w.reqmin : volt(u.val, mv) < -w.reqmin; }

constexpr long long kNoRow = 0x7fffffffffffffffll;
// first row >= c (and < limit) at which this run makes the detector fire, or kNoRow; doubt = first row >= c that the record cannot decide.
// eend: the end of the list's slot (entry e lives at eend[-(e + 1)])
__device__ __forceinline__ long long run_fire(const Walker &w, const Run &u, const uint16_t *eend, long long c, long long limit, int W, int sure_i, float mv, long long &doubt) {
   doubt = kNoRow;
   const long long last_row = u.pos + W - 2;                         // the owner is strictly inside the window up to here
   if (last_row < c || u.f >= limit) return kNoRow;
   if (!amp_pass(w, u, mv)) return kNoRow;
   if (u.unknown) { const long long n = max(c, u.f); if (n < u.f + u.nsure && n < limit) doubt = n; return kNoRow; }
   for (int i = 0; i < u.nlead; ++i) {
      const long long n = u.f + i;
      if (n < c) continue;
      if (n >= limit) return kNoRow;
      if (rise_pass(w, u.top, u.val, (int)eend[-(u.e0 + i + 1)], mv)) return n; }
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
      if (rise_pass(w, u.top, u.val, (int)eend[-(u.e0 + u.nlead + i + 1)], mv)) return n; }
   return kNoRow; }

// a lane's place in its record stream: record k of tile g's list (of one screen and head); e0 = margin entries in front of it.
// A deferred candidate's placeholder (w1 == 0xffff8001) opens its overflow slot: sub = the record of the slot the lane stands on.
struct RecIt {
   long long g;
   int k, nrec, e0;
   const unsigned char *slot;
   uint32_t w0, w1;
   bool end, bad;
   int sub, nsub, se0;
   const unsigned char *ovf; };
struct RecSrc {                      // the stream: this chain's head and screen
   const PeakDir *dir; const unsigned char *pool, *ovf;
   int nscreens, ntrks, screen, head, hcap;
   long long g_end; };               // tiles at or behind g_end hold no row of the chain
__device__ __forceinline__ const uint16_t *it_eend(const RecIt &it, const RecSrc &S) {      // entry e of the record's list lives at eend[-(e + 1)]
   return reinterpret_cast<const uint16_t *>(it.sub >= 0 ? it.ovf + kSfOvfBytes : it.slot + S.hcap); }
__device__ __forceinline__ int it_e0(const RecIt &it) { return it.sub >= 0 ? it.se0 : it.e0; }
__device__ __forceinline__ void it_open(RecIt &it, const RecSrc &S, long long g);
// the lane has just moved onto record k of its list: read it; a placeholder opens its overflow slot
__device__ __forceinline__ void it_land(RecIt &it, const RecSrc &S) {
   for (;;) {
      const uint2 r = *reinterpret_cast<const uint2 *>(it.slot + 8 * it.k);
      it.w0 = r.x; it.w1 = r.y; it.sub = -1;
      if (r.y != 0xffff8001u) return;
      it.ovf = S.ovf + (size_t)r.x * kSfOvfBytes;
      const int n = *reinterpret_cast<const int *>(it.ovf);
      if (n < 0) { it.end = true; it.bad = true; return; }                        // (not representable: whoever needs it takes the sample path)
      if (n > 0) { it.sub = 0; it.nsub = n; it.se0 = 0; const uint2 r2 = *reinterpret_cast<const uint2 *>(it.ovf + 8); it.w0 = r2.x; it.w1 = r2.y; return; }
      if (++it.k >= it.nrec) { it_open(it, S, it.g + 1); return; } } }         // (a candidate that turned out to have no row above the screen)
__device__ __forceinline__ void it_open(RecIt &it, const RecSrc &S, long long g) {
   it.end = false; it.bad = false; it.k = 0; it.e0 = 0; it.nrec = 0; it.w0 = 0; it.w1 = 0; it.slot = nullptr; it.sub = -1; it.nsub = 0; it.se0 = 0; it.ovf = nullptr;
   for (;; ++g) {
      it.g = g;
      if (g >= S.g_end) { it.end = true; return; }
      const size_t li = (size_t)(g * S.nscreens + S.screen) * S.ntrks + S.head;
      const PeakDir d = S.dir[li];
      if (d.nrec == 0xffffu) { it.end = true; it.bad = true; return; }             // (capacity: the burst takes the sample path)
      if (d.nrec == 0) continue;
      it.nrec = d.nrec; it.slot = S.pool + li * (size_t)S.hcap;
      it_land(it, S);
      return; } }
__device__ __forceinline__ void it_next(RecIt &it, const RecSrc &S) {
   if (it.sub >= 0) {
      it.se0 += pk_nent(it.w0, it.w1);
      if (++it.sub < it.nsub) { const uint2 r = *reinterpret_cast<const uint2 *>(it.ovf + 8 + 8 * it.sub); it.w0 = r.x; it.w1 = r.y; return; }
      it.sub = -1; }                                                             // (the placeholder itself carries no entries)
   else it.e0 += pk_nent(it.w0, it.w1);
   if (++it.k < it.nrec) it_land(it, S);
   else it_open(it, S, it.g + 1); }

// an event the fast path only noted (k_emit finishes it): where its record is, and the gain in force
__device__ __forceinline__ rtfe_event note_event(const RecIt &it, float g) {
   union { rtfe_event e; uint32_t w[4]; } u;
   u.w[0] = (uint32_t)it.g; u.w[1] = __float_as_uint(g); u.w[2] = (uint32_t)it.k | ((uint32_t)it.e0 << 16); u.w[3] = 0xffffffffu;
   return u.e; }

__global__ void __launch_bounds__(64) k_gain(const DevCfg *__restrict__ cfgp, long long nrows, long long row_base,
                                             const rtfe_burst *__restrict__ bursts, BurstScratch *__restrict__ scratch, BurstCtl *__restrict__ ctl,
                                             uint32_t *__restrict__ counts, rtfe_event *__restrict__ events, float *__restrict__ chain_h,
                                             const PeakDir *__restrict__ dir, const unsigned char *__restrict__ pool, const unsigned char *__restrict__ ovf, long long ntiles) {
   __shared__ float s_heights[64 * 10];
   const DevCfg &cfg = *cfgp;
   const int ntrks = cfg.ntrks, nwalk = cfg.nparm * ntrks;
   const int lane = threadIdx.x;
   const float mv = cfg.maxvolts;
   const int nchains = scratch->nbursts * nwalk;
   float *heights = s_heights + lane * 10;
   for (int ci = blockIdx.x * 64 + lane; ci < nchains; ci += gridDim.x * 64) {
      const int b = ci / nwalk;
      const int wi = ci - b * nwalk, pidx = wi / ntrks, trk = wi - pidx * ntrks;
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
      for (int i = 0; i < 10; ++i) heights[i] = 0;
      // column rows: the detector's row n reads sample n - d.  Before c the window is filling on zone samples only.
      long long c = reset + W + max(trk, d) + 1 - d;
      const long long limit = stop - d;
      rtfe_event *ev = events + B.event_base + (size_t)(pidx * ntrks + trk) * B.event_cap;
      const unsigned int cap = B.event_cap;
      bool failed = false;
      int why = 0;
      unsigned int n_fast = 0, n_slow = 0, guard = 0;
      RecSrc src;
      src.dir = dir; src.pool = pool; src.ovf = ovf; src.nscreens = cfg.nscreens; src.ntrks = ntrks; src.screen = P.screen; src.head = head; src.hcap = cfg.pk_slot;
      {  long long ge = limit <= 0 ? 0 : (limit + kSfTile - 1) / kSfTile;       // first tile whose candidates all lie at or behind the limit
         src.g_end = ge < ntiles ? ge : ntiles; }
      RecIt alive;
      {  long long g0 = (c - W) / kSfTile; if (c - W < 0) g0 = 0;                // (a candidate up to W - 2 rows in front of c still has rows at or behind it)
         it_open(alive, src, g0); }
      while (!failed) {
         if (++guard > 4000000u) { failed = true; why = 7; break; }              // (cannot happen: every round moves the stream or c forward)
         if (alive.end) { if (alive.bad) { failed = true; why = 1; } break; }
         const long long tile0 = alive.g * kSfTile;
         const Run ua = run_decode(alive.w0, alive.w1, tile0, it_e0(alive));
         if (ua.pos + W - 2 < c) { it_next(alive, src); continue; }             // its rows are behind the countdown for good
         const bool steady = cfg.agc_off || (cfg.mode == RTFE_PE ? w.datablock : (w.peakcount > 15 && w.v_avg_height_count == 0));
         // ---- the fast path: steady state, a record with a sure stretch, the countdown over before its first row, the thresholds inside
         // the band the sure level stands for, a clear amplitude, and nothing else that could fire before this record's owner has left
         // the window.  Then it fires - at one of its lead rows or at its first sure row, k_emit will say which - and all that
         // feeds back is the extreme's value. ----
         if (cfg.pk_fast && steady && alive.sub < 0 && !ua.unknown && ua.nsure > 0 && c <= ua.f && ua.f + ua.nlead < limit && w.rise_hi <= S.sure_i && w.nevents < cap) {
            const int a = ua.top ? ua.val : -ua.val;
            if (w.reqmin == 0 || a >= w.min_hi) {
               RecIt nx = alive;
               it_next(nx, src);
               bool clear = nx.end && !nx.bad;
               if (!nx.end) { const long long fn = nx.g * kSfTile - kSfPosBias + (long long)(nx.w0 & 0x7ffu) + (long long)((nx.w0 >> 12) & 63u); clear = fn > ua.pos + W; }
               const float g = w.agc_gain;
               const int ti = (int)(0.005f * fast_rcp(g) * cfg.lsb_per_volt);
               if (clear && ti + 4 <= 254) {
                  ev[w.nevents] = note_event(alive, g);
                  const float val = volt(ua.val, mv);
                  if (ua.top) w.v_top = val; else w.v_bot = val;
                  ++w.nevents; ++n_fast;
                  agc_after_peak(w, &cfg, P, heights, ua.top, 0.0);
                  if (!(w.agc_gain > 0)) { w.flags |= RTFE_F_DETECTOR_FATAL; failed = true; why = 4; break; }      // src/decoder.c:782
                  if (!approx_thresholds(w, P, cfg.lsb_per_volt)) {
                     update_thresholds(w, P, cfg.lsb_per_volt);
                     if (w.flags & RTFE_F_SCREEN_UNDERFLOW) { failed = true; why = 5; break; } }
                  c = ua.pos + W + 1;
                  alive = nx;
                  continue; } } }
         // ---- the general step: earliest firing run among the tops and the bottoms from the first live record on ----
         if (w.thr_dirty) {
            update_thresholds(w, P, cfg.lsb_per_volt);
            if (w.flags & RTFE_F_SCREEN_UNDERFLOW) { failed = true; why = 5; break; } }
         long long best = kNoRow, best_doubt = kNoRow;
         bool have = false, best_top = false, top_done = false, bot_done = false;
         long long bad_row0 = kNoRow;                                       // first row of a tile whose list is not there
         Run bu = ua;
         for (RecIt j = alive; !(top_done && bot_done); it_next(j, src)) {
            if (j.end) { if (j.bad) bad_row0 = j.g * kSfTile; break; }
            const Run u = run_decode(j.w0, j.w1, j.g * kSfTile, it_e0(j));
            if (u.top ? top_done : bot_done) continue;
            if (u.f > best || u.f >= limit) { if (u.top) top_done = true; else bot_done = true; continue; }
            long long dr;
            const long long n = run_fire(w, u, it_eend(j, src), c, limit, W, S.sure_i, mv, dr);
            if (dr < best_doubt) best_doubt = dr;
            if (n != kNoRow) {
               if (u.top) top_done = true; else bot_done = true;         // (runs of one kind are ordered by row)
               if (n < best || (n == best && u.top && !best_top)) { best = n; bu = u; best_top = u.top; have = true; } } }
         if (bad_row0 != kNoRow && !(have && best < bad_row0)) { failed = true; why = 1; break; }      // (a list that is not there, and something in it might fire first)
         if (best_doubt != kNoRow && best_doubt <= best) { failed = true; why = 2; break; }
         if (!have) break;                                                  // nothing fires any more: the chain is done
         // ---- detection: refine_peak + the callback's effect on AGC state (src/decoder.c:700-749, 574-609) ----
         const Run &u = bu;
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
         else {
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
         ++w.nevents; ++n_slow;
         agc_after_peak(w, &cfg, P, heights, u.top, t_peak);
         if (!(w.agc_gain > 0)) { w.flags |= RTFE_F_DETECTOR_FATAL; failed = true; why = 4; break; }      // src/decoder.c:782
         update_thresholds(w, P, cfg.lsb_per_volt);
         if (w.flags & RTFE_F_SCREEN_UNDERFLOW) { failed = true; why = 5; break; }
         c = u.pos + W + 1; }
      // ---- publish ----
      if (n_fast) atomicAdd(&scratch->dbg[0], (unsigned long long)n_fast);
      if (n_slow) atomicAdd(&scratch->dbg[1], (unsigned long long)n_slow);
      if (failed) { atomicExch(&ctl[b].status, (int)kBurstNeedsFull); atomicAdd(&scratch->why[why & 7], 1ull); }
      counts[((size_t)b * cfg.nparm + pidx) * ntrks + trk] = w.nevents < cap ? w.nevents : cap;
      chain_h[(size_t)b * nwalk + wi] = w.v_avg_height;
      if (w.flags & ~(unsigned)RTFE_F_SCREEN_UNDERFLOW) atomicOr(&ctl[b].bflags, w.flags & ~(unsigned)RTFE_F_SCREEN_UNDERFLOW); } }

// ------------------------------------------------------------------------------------------------
// k_emit: the events the fast path noted -> the events the reference's callbacks see.  One workgroup per chain at a time,
// a lane per event (16 bytes in, 16 bytes out, consecutive lanes consecutive events).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_emit(const DevCfg *__restrict__ cfgp, const rtfe_burst *__restrict__ bursts, const BurstScratch *__restrict__ scratch,
                                              const BurstCtl *__restrict__ ctl, const uint32_t *__restrict__ counts, rtfe_event *__restrict__ events,
                                              const float *__restrict__ chain_h, const unsigned char *__restrict__ pool) {
   const DevCfg &cfg = *cfgp;
   const int ntrks = cfg.ntrks, nwalk = cfg.nparm * ntrks;
   const int nchains = scratch->nbursts * nwalk;
   const float mv = cfg.maxvolts;
   for (int ci = blockIdx.x; ci < nchains; ci += gridDim.x) {
      const int b = ci / nwalk;
      const int wi = ci - b * nwalk, pidx = wi / ntrks, trk = wi - pidx * ntrks;
      if (ctl[b].status != kBurstReady) continue;                        // (a burst the sample path redoes)
      const rtfe_burst B = bursts[b];
      const DevParm &P = cfg.parm[pidx];
      const int W = P.W, d = cfg.skew[trk], head = cfg.trk_to_head[trk];
      const long long reset = ctl[b].reset;
      const unsigned int nev = counts[((size_t)b * cfg.nparm + pidx) * ntrks + trk];
      rtfe_event *ev = events + B.event_base + (size_t)(pidx * ntrks + trk) * B.event_cap;
      Walker wk = {};
      wk.v_avg_height = chain_h[(size_t)b * nwalk + wi];
      for (unsigned int i = threadIdx.x; i < nev; i += blockDim.x) {
         union { rtfe_event e; uint32_t w[4]; } in;
         in.e = ev[i];
         if (in.w[3] != 0xffffffffu) continue;                            // the general step wrote it out in full
         const long long g = (long long)in.w[0];
         const float gain = __uint_as_float(in.w[1]);
         const int k = (int)(in.w[2] & 0xffffu), e0 = (int)(in.w[2] >> 16);
         const unsigned char *slot = pool + ((size_t)(g * cfg.nscreens + P.screen) * ntrks + head) * (size_t)cfg.pk_slot;
         const uint2 r = *reinterpret_cast<const uint2 *>(slot + 8 * k);
         const uint16_t *eend = reinterpret_cast<const uint16_t *>(slot + cfg.pk_slot);
         const Run u = run_decode(r.x, r.y, g * kSfTile, e0);
         wk.agc_gain = gain; wk.flags = 0;
         update_thresholds(wk, P, cfg.lsb_per_volt);                      // the exact thresholds of src/decoder.c:785-786 at that gain
         long long n = u.f + u.nlead;                                     // the first sure row, unless a lead row passes
         for (int j = u.nlead - 1; j >= 0; --j) if (rise_pass(wk, u.top, u.val, (int)eend[-(u.e0 + j + 1)], mv)) n = u.f + j;
         const int ld = (int)(u.pos - n) + W;
         const int iprev = u.top ? u.val - u.dprev : u.val + u.dprev, inext = u.top ? u.val - u.dnext : u.val + u.dnext;
         const int adjcode = refine_code(&cfg, u.val, iprev, inext, gain, u.top);
         const float val = volt(u.val, mv);
         rtfe_event e;
         e.sample = (uint32_t)(n + d - reset);
         e.v_peak = (cfg.invert && val == 0.0f) ? -0.0f : val;
         e.agc_gain = gain;
         e.trk = (uint8_t)trk;
         e.flags = (uint8_t)((u.top ? 0 : 1) | (adjcode << 1));
         e.left_distance = (uint8_t)ld;
         e.parmset = (uint8_t)pidx;
         ev[i] = e; } } }

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
