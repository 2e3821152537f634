// rtfe_dense.hip — the sample path in two passes, for the formats whose window always holds a top AND a bottom (PE, GCR):
//
//   k_dseg    dense and stateless.  The tape is cut into SUB-SEGMENTS of kDsSub rows on a fixed grid; a lane per (distinct
//             parameter set, track, sub-segment) walks the detector of src/decoder.c:751-810 over the rows of its sub-segment on
//             a tile in LDS (the candidate screen of rtfe_kernels.hip in front of it), started a warm-up early with no countdown
//             pending.  What makes that possible: everything slow in the detector's state - the AGC gain, the baseline - only
//             enters through two thresholds (src/decoder.c:785-786), and WHICH row of an extreme fires changes nothing
//             downstream (the countdown ends when the extreme leaves the window, src/decoder.c:741; the AGC sees the extreme's
//             value).  So the lane decides against a BAND of thresholds [s_lo, s_hi] x (rise, min_peak), taken from the signal's
//             local amplitude: a margin above the band fires for sure, one below it cannot, one inside it is a "maybe" that the
//             next rows usually settle (the same extreme, now for sure: the record says from which row on it may have fired).
//             Whatever the band cannot settle is a DOUBT: the list ends there.  The state that is left - the countdown, and the
//             reference's stale window minimum, a function of the samples (rtfe_kernels.hip: stale_ld) - is noted at the
//             sub-segment's first own row.  Output per lane: a fixed slot {count, countdown at the start, doubt row, records}.
//   k_dchain  a lane per chain (burst, distinct parameter set, track), sequential: the AGC schedule of the block decoders
//             (src/decode_gcr.c:843-864, src/decode_pe.c:127-198, src/decode_nrzi.c:196-229), the thresholds, the events.  It
//             runs the LITERAL detector on the samples in HBM from the burst's restart row (window filling, staggered start,
//             deskew FIFO start-up: src/decoder.c:820-861) until, at a sub-segment boundary, its own state equals what the
//             sub-segment's lane noted (countdown, a forced rescan seen) and its thresholds lie inside the sub-segment's band;
//             from there it consumes records - per record: is the band still right, which of the maybe rows fires (the
//             reference's float comparison on the two samples), refine_peak's half-sample code from the neighbour distances,
//             the event, the AGC mirror, the new thresholds - and drops back to the literal detector wherever a list ends in a
//             doubt, a join fails, or the thresholds leave the band (exact either way; the data decides only the speed).
//
// Parameter sets that differ only in what the HOST decoders read (clk_window, clk_alpha, pulse_adj, z1pt ...; src/parmsets.c:77-110)
// are one chain: its events are stored into every such set's region (DevCfg::uset_*).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtfe {

constexpr int kDsSub = 128;                    // rows of a sub-segment
constexpr int kDsJ = 8;                        // sub-segments per tile (8 x 128 own rows behind one warm-up and one halo: 7.8 % of the rows are screened and classified twice; 4 x 128: 15.6 %)
constexpr int kDsTile = kDsSub * kDsJ;         // own rows of a tile
constexpr int kDsRight = 16;                   // rows behind a tile's own rows: a maybe that begins in the last sub-segment is settled there (kDsMaxMaybe + 1, a multiple of 8)
constexpr int kDsThreads = 512;
constexpr int kDsMaxMaybe = 15;
constexpr int kDsNoJoin = 0xff, kDsNoDoubt = 0xff;
struct DsHdr { uint8_t count, start_blind, doubt, flags; float s_lo, s_hi; uint32_t pad; };      // 16 bytes in front of a slot's records; [s_lo, s_hi]: the band the lane decided against
// a record (8 bytes): w0 = nf (first row the extreme may have fired at, relative to the sub-segment's first row: 0..127) | nmaybe << 8
//                          (rows nf .. nf + nmaybe - 1 are "maybe", row nf + nmaybe fires for sure) | kind << 12 (1 = bottom)
//                          | ld << 13 (left_distance at row nf: 2 .. W - 1) | dprev << 19 >> ... (see ds_pack)
//                     w1 = val (int16 code of the extreme) | dprev << 16 | dnext << 24 (|val - neighbour| towards "beyond the extreme", 0..255)
struct DsRec { uint32_t w0, w1; };
__device__ __forceinline__ DsRec ds_pack(int nf, int nm, int kind, int ld, int val, int dp, int dn) {
   DsRec r; r.w0 = (uint32_t)nf | ((uint32_t)nm << 8) | ((uint32_t)kind << 12) | ((uint32_t)ld << 13);
   r.w1 = (uint32_t)(uint16_t)val | ((uint32_t)dp << 16) | ((uint32_t)dn << 24); return r; }

// (DevCfg::ds_up distinct sets of one window width are classified together - more take further passes over the tile: 2 leaves room for four
//  workgroups per CU, 3 for three; rtfe_create picks by the configuration)
constexpr int kDsPlanes = 6;                   // per (set, track) over the tile's rows: own rows F(ire) Y(maybe) D(oubt) K(ind); warm-up rows F K
struct DsThr { int r_lo, r_hi, q_lo, q_hi; };  // margins / extremes in int16 codes: >= hi passes for every threshold of the band, <= lo for none
struct DsLds { unsigned bits, ldpos, cls, thr, bdl, band, mmax, total; };
__host__ __device__ inline unsigned ds_bstride(int tile_rows) { return (unsigned)((tile_rows + kScreenHalo) / 8 + 8 + 7) & ~7u; }      // bytes of a screen bitmap's row (the halo word in front, one spare)
__host__ __device__ inline unsigned ds_ldstride(int tile_rows) { return (unsigned)(tile_rows + kScreenHalo + 8 + 7) & ~7u; }      // bytes of a left-distance map's row
__host__ __device__ inline unsigned ds_cstride(int tile_rows) { return (unsigned)(tile_rows / 8 + 8 + 7) & ~7u; }      // bytes of a plane's row of bits (64-bit words, one spare)
__host__ __device__ inline DsLds ds_lds_layout(int ntrks, int halo_rows, int tile_rows, int up) {
   DsLds L;
   unsigned off = lds_align16((unsigned)ntrks * (unsigned)(halo_rows + tile_rows + 8) * 2u + 16u);
   L.bits = off;  off = lds_align16(off + (unsigned)ntrks * 3u * ds_bstride(tile_rows));       // (one screen at a time: top / bottom candidates, forced rescans)
   L.ldpos = off; off = lds_align16(off + (unsigned)ntrks * 2u * ds_ldstride(tile_rows));       // left_distance of the window's first maximum | of the reference's (stale) minimum
   L.cls = off;   off = lds_align16(off + (unsigned)up * kDsPlanes * (unsigned)ntrks * ds_cstride(tile_rows));
   L.thr = off;   off = lds_align16(off + (unsigned)up * kDsJ * (unsigned)ntrks * (unsigned)sizeof(DsThr));
   L.bdl = off;   off = lds_align16(off + (unsigned)up * kDsJ * (unsigned)ntrks * 8u);
   L.band = off;  off = lds_align16(off + (unsigned)kDsJ * (unsigned)ntrks * 8u);
   L.mmax = off;  off = lds_align16(off + (unsigned)kDsJ * (unsigned)ntrks * 4u);
   L.total = off;
   return L; }

// the margins of a record's maybe rows - the extreme above the higher edge (tops) / the lower edge above it (bottoms), int16 codes: what the
// reference's two edge comparisons come down to (the nearer edge decides) - four to an 8-byte word behind the record
template <class ColT> __device__ __forceinline__ int ds_put_margins(DsRec *recs, int count, const ColT &yb, int W, int nf, int nm, int kind, int val) {
   for (int m0 = 0; m0 < nm; m0 += 4) {
      uint32_t wd[2] = {0, 0};
      for (int m = m0; m < nm && m < m0 + 4; ++m) {
         const int q = nf + m, vl = yb[q - W + 1], vr = yb[q];
         int mg = kind == 0 ? val - max(vl, vr) : min(vl, vr) - val;
         mg = mg < 0 ? 0 : (mg > 65535 ? 65535 : mg);
         wd[(m - m0) >> 1] |= (uint32_t)mg << (16 * ((m - m0) & 1)); }
      DsRec r; r.w0 = wd[0]; r.w1 = wd[1]; recs[count++] = r; }
   return count; }

enum { kDsMiss = 0, kDsMaybe = 1, kDsSure = 2 };
__device__ __forceinline__ int ds_cls(const DsThr &h, bool amp_on, int m, int v) {
   if (m <= h.r_lo || (amp_on && v <= h.q_lo)) return kDsMiss;
   return (m >= h.r_hi && (!amp_on || v >= h.q_hi)) ? kDsSure : kDsMaybe; }

// ------------------------------------------------------------------------------------------------
// k_dseg.  Per tile and window width: (1) the candidate screen of rtfe_kernels.hip, all lanes; (2) per row, all lanes: the margins of
// the window's maximum and of the reference's (possibly stale) minimum, and - per distinct set of that width - what the detector would do
// at the row if it looked: fire / maybe / doubt / nothing, as bits over the rows; (3) a lane per (set, sub-segment, track) resolves what
// is sequential - the countdown - on those bits: find the next set bit, one load for the extreme's place, jump behind the countdown.
// ------------------------------------------------------------------------------------------------
// NT: the track count at compile time (0: whatever the configuration says) - the rows' stride in the tile is then a constant, and every sample of a strip an
// immediate offset from one address instead of an addition each
template <int NT>
__global__ void __launch_bounds__(kDsThreads, 4) k_dseg(const DevCfg *__restrict__ cfgp, const int16_t *__restrict__ rows, long long nrows, long long ntiles,
                                                     unsigned char *__restrict__ dead, unsigned char *__restrict__ qbytes, unsigned char *__restrict__ slots,
                                                     unsigned long long *__restrict__ dbg) {
#ifdef RTFE_CPU_EMUL
   unsigned char *smem = g_dyn_smem;
#else
   extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
   __shared__ DevCfg cfg;
   __shared__ int s_any, s_amp[RTFE_MAXTRKS], s_bmx[kDsJ * RTFE_MAXTRKS], s_bmn[kDsJ * RTFE_MAXTRKS];
   __shared__ unsigned int s_noisy;
   for (int i = threadIdx.x; i < (int)(sizeof(DevCfg) / 4); i += blockDim.x) reinterpret_cast<int *>(&cfg)[i] = reinterpret_cast<const int *>(cfgp)[i];
   __syncthreads();
   const int ntrks = NT ? NT : cfg.ntrks, pad = cfg.ds_pad, T = pad + kDsTile + kDsRight, nu = cfg.nuset;
   const DsLds L = ds_lds_layout(ntrks, cfg.halo_rows, T, cfg.ds_up);
   Tile tl;
   tl.x = reinterpret_cast<int16_t *>(smem); tl.halo = cfg.halo_rows; tl.ldw = 0; tl.colof = cfg.trk_to_head; tl.ntrks = ntrks; tl.skew = cfg.skew;
   tl.bits = smem + L.bits; tl.bstride = (int)ds_bstride(T); tl.ldpos = smem + L.ldpos; tl.ldstride = (int)ds_ldstride(T); tl.fd = nullptr;
   tl.reset = -(1ll << 40);                                           // (the regular deskew regime everywhere: k_dchain joins only behind the start-up rows)
   float2 *s_band = reinterpret_cast<float2 *>(smem + L.band);         // [j][t]: the band from the amplitude
   int *s_mmax = reinterpret_cast<int *>(smem + L.mmax);               // [j][t]: the largest margin of any candidate the sub-segment's lanes can meet
   DsThr *s_thr = reinterpret_cast<DsThr *>(smem + L.thr);             // [ul][j][t]
   float2 *s_bdl = reinterpret_cast<float2 *>(smem + L.bdl);           // [ul][j][t]: the band the lane decides against (its slot's header)
   unsigned char *s_cls = smem + L.cls;                                // [ul][plane][t][cstride]
   const int cstride = (int)ds_cstride(T);
   const float lsb = cfg.lsb_per_volt;
   const int slot_bytes = cfg.ds_slot, cap = cfg.ds_cap;
   const FastDiv fdn(ntrks);
   long long t_load = 0, t_scr = 0, t_walk = 0, t_sld = 0, tq = 0;
   const bool prof = cfg.debug == 7;
   const int wmax = cfg.halo_rows - kScreenHalo;                       // (>= widest window + 1 + max skew)
   for (long long g = blockIdx.x; g < ntiles; g += gridDim.x) {
      tl.row0 = g * kDsTile - pad; tl.nrows = T;
      __syncthreads();
      if (prof) tq = clock64();
      if (threadIdx.x < RTFE_MAXTRKS) s_amp[threadIdx.x] = 0;
      if (threadIdx.x == 0) s_noisy = 0;
      load_tile(&cfg, tl, rows, nrows);
      __syncthreads();
      // ---- the quiet map's bits of the tile's own rows (k_quiet folded in: bit c = every sample of rows [64 c, 64 c + 64) inside the quiet band;
      // a tile is kDsTile / 64 = 16 groups: two bytes of the map) ----
      {  unsigned noisy = 0;
         const uint32_t qpk = pk_dup(cfg.quiet_i), q2 = 2u * (uint32_t)cfg.quiet_i;
         const int4 *own = reinterpret_cast<const int4 *>(tl.x + (tl.halo + pad) * ntrks);      // (16-byte vectors: halo and pad are multiples of 8 rows)
         const int vpg = 8 * ntrks;                                       // vectors per group of 64 rows
         const FastDiv fdg(vpg);
         for (int v = threadIdx.x; v < (kDsTile / kChunkRows) * vpg; v += blockDim.x) {
            const int4 x4 = own[v];
            const uint32_t m = pk_maxu(pk_maxu(pk_addu((uint32_t)x4.x, qpk), pk_addu((uint32_t)x4.y, qpk)), pk_maxu(pk_addu((uint32_t)x4.z, qpk), pk_addu((uint32_t)x4.w, qpk)));
            if ((m & 0xffffu) > q2 || (m >> 16) > q2) noisy |= 1u << fdg.div(v); }
         if (noisy) atomicOr(&s_noisy, noisy); }
      // ---- the band of every (sub-segment, track): from the amplitude of the rows the sub-segment's lanes can see, [j kDsSub - wmax,
      // pad + (j + 1) kDsSub).  All lanes: the extremes of 8 rows of a track each, into the one or two sub-segments that see them (round 4:
      // a lane per (sub-segment, track) walked its ~200 rows alone - 72 lanes of 512 busy for a tenth of the tile's time) ----
      for (int i = threadIdx.x; i < kDsJ * ntrks; i += blockDim.x) { s_bmx[i] = -40000; s_bmn[i] = 40000; }
      __syncthreads();
      {  const int nch = (wmax + pad + kDsTile) / 8;                     // (wmax, pad: multiples of 8)
         for (int i = threadIdx.x; i < nch * ntrks; i += blockDim.x) {
            const int ch = fdn.div(i), t = i - ch * ntrks;
            const Col yb = tile_col(tl, t, cfg.skew[t]);
            const int a = ch * 8 - wmax;
            int mx = -40000, mn = 40000;
            #pragma unroll
            for (int k = 0; k < 8; ++k) { const int v = yb[a + k]; mx = max(mx, v); mn = min(mn, v); }
            // sub-segment j sees rows [j kDsSub - wmax, pad + (j + 1) kDsSub)
            int j_hi = (a + wmax) / kDsSub; if (j_hi > kDsJ - 1) j_hi = kDsJ - 1;
            int j_lo = a + 8 - pad - kDsSub <= 0 ? 0 : (a + 8 - pad - kDsSub + kDsSub - 1) / kDsSub;
            for (int j = j_lo; j <= j_hi; ++j) { atomicMax(&s_bmx[j * ntrks + t], mx); atomicMin(&s_bmn[j * ntrks + t], mn); } } }
      __syncthreads();
      for (int i = threadIdx.x; i < kDsJ * ntrks; i += blockDim.x) {
         const int j = fdn.div(i), t = i - j * ntrks;
         const int amp = s_bmx[i] - s_bmn[i];
         atomicMax(&s_amp[t], amp);
         const float av = (float)amp / lsb;                             // volts, peak to peak
         // the AGC makes the thresholds follow the signal: (v_avg_height / 4) / agc_gain ~ (recent peak-to-peak height) / 4 <= amplitude / 4
         float hi = av * cfg.ds_band_hi * 0.25f;
         float lo = hi * cfg.ds_band_lo; if (lo < cfg.ds_sfloor) lo = cfg.ds_sfloor;
         if (hi >= 0.6f && hi < 1.02f) hi = 1.02f;                      // (a block's first peaks meet the fresh detector - baseline 4 V, gain 1 -: its scale is in the band of every signal of usual height)
         if (hi < lo) hi = lo;
         s_band[i] = make_float2(lo, hi); }
      __syncthreads();
      if (threadIdx.x == 0) {                                             // (groups that are not complete - the tape ends inside them - are not quiet, as k_quiet has it)
         unsigned complete = 0;
         for (int k = 0; k < kDsTile / kChunkRows; ++k) if (g * kDsTile + (long long)(k + 1) * kChunkRows <= nrows) complete |= 1u << k;
         if (kDsTile / kChunkRows == 8) qbytes[g] = (unsigned char)(~s_noisy & complete);
         else reinterpret_cast<uint16_t *>(qbytes)[g] = (uint16_t)(~s_noisy & complete); }      // (16 groups: two bytes of the map)
      if (prof) { const long long t2 = clock64(); t_load += t2 - tq; tq = t2; }
      const int tile_lim = (nrows - tl.row0 < (long long)T) ? (int)(nrows - tl.row0) : T;
      for (int s = 0; s < cfg.nscreens; ++s) {
         const DevScreen S = cfg.screen[s];
         // nothing in this tile can rise above the screen: no list (k_dchain reads the flag, not the slots)
         bool flat = true;
         for (int t = 0; t < ntrks; ++t) if (s_amp[t] > S.rise_i) flat = false;
         if (flat) { if (threadIdx.x == 0) dead[g * cfg.nscreens + s] = 1; continue; }
         if (threadIdx.x == 0) s_any = 0;
         for (int i = threadIdx.x; i < kDsJ * ntrks; i += blockDim.x) s_mmax[i] = 0;
         __syncthreads();
         const int W = S.W, warm = cfg.ds_warm[s];
         {  const int hs = kScreenHalo / kStrip, nstrips = T / kStrip + hs, per = nstrips * ntrks;
            int any = 0;
            for (int i = (int)threadIdx.x; i < per; i += blockDim.x) {
               const int q = fdn.div(i), t = i - q * ntrks, strip = q - hs;
               int mm = 0;
               any |= screen_strip(tl, S, 0, t, strip, &mm);
               if (mm > 0 && strip >= 0) {                                // whose lanes look at these rows: the sub-segment that owns them, the next one in its warm-up
                  const int r0 = strip * kStrip;
                  int jo = r0 < pad ? 0 : (r0 - pad) / kDsSub;
                  if (jo > kDsJ - 1) jo = kDsJ - 1;
                  atomicMax(&s_mmax[jo * ntrks + t], mm);
                  if (r0 >= pad && jo + 1 < kDsJ && r0 + kStrip > pad + (jo + 1) * kDsSub - warm) atomicMax(&s_mmax[(jo + 1) * ntrks + t], mm); } }
            if (any) s_any = 1; }
         __syncthreads();
         if (prof) { const long long t2 = clock64(); t_scr += t2 - tq; tq = t2; }
         if (!s_any) { if (threadIdx.x == 0) dead[g * cfg.nscreens + s] = 1; __syncthreads(); continue; }
         if (threadIdx.x == 0) dead[g * cfg.nscreens + s] = 0;
         // ---- the distinct sets of this width, ds_up at a time ----
         int us_all[RTFE_MAXPARMSETS], nus_all = 0;
         for (int u = 0; u < nu; ++u) if (cfg.parm[cfg.uset_rep[u]].screen == s) us_all[nus_all++] = u;
         for (int u0 = 0; u0 < nus_all; u0 += cfg.ds_up) {
            const int nus = nus_all - u0 < cfg.ds_up ? nus_all - u0 : cfg.ds_up;
            // (a) every lane's band and thresholds.  A sub-segment of small signal (a gap, a block's first or last rows) is passed by chains
            // whose thresholds come from elsewhere: if nothing in it rises above a level such thresholds clear, its band is [that level, infinity)
            for (int i = threadIdx.x; i < nus * kDsJ * ntrks; i += blockDim.x) {
               const int q = fdn.div(i), t = i - q * ntrks, ul = q / kDsJ, j = q - ul * kDsJ;
               const DevParm &P = cfg.parm[cfg.uset_rep[us_all[u0 + ul]]];
               float2 bd = s_band[j * ntrks + t];
               const float lo_s = (float)(s_mmax[j * ntrks + t] + 4) / (P.rise * lsb);
               if (lo_s <= cfg.ds_quiet_s) bd = make_float2(lo_s > cfg.ds_sfloor ? lo_s : cfg.ds_sfloor, 3.0e38f);
               const float fr = P.rise * bd.y * lsb, fq = P.min_peak * bd.y * lsb;
               DsThr h;
               h.r_hi = fr > 1.0e9f ? 0x3fffffff : (int)floorf(fr) + 3; h.r_lo = (int)floorf(P.rise * bd.x * lsb) - 2;
               h.q_hi = fq > 1.0e9f ? 0x3fffffff : (int)floorf(fq) + 3; h.q_lo = (int)floorf(P.min_peak * bd.x * lsb) - 2;
               s_thr[i] = h; s_bdl[i] = bd; }
            __syncthreads();
            // (b) all lanes, a strip of 8 rows of a track each, straight-line: the window's edges and extremes, the reference's stale minimum
            // carried along the strip (a forced rescan, or the stale sample leaving the window, makes it the true minimum again: the chain of
            // stale_ld, one step per row), the margins - then per set the row's outcome as bits.  The stale minimum's left_distance replaces
            // the true one's in the map, in place: where another lane's stale_ld reads the map - at rescan rows - the two are the same.
            {  const int nst = T / kStrip, per = nst * ntrks;
               const int ppl = ntrks * cstride;                           // bytes of a plane
               const bool first_pass = u0 == 0;
               for (int i = (int)threadIdx.x; i < per; i += blockDim.x) {
                  const int q = fdn.div(i), t = i - q * ntrks;
                  const u64 *tm = tl.map(0, 0, t), *bm = tl.map(0, 1, t), *am = tl.map(0, 2, t);
                  const unsigned char *ldt = tl.ldmap(0, 0, t);
                  unsigned char *ldb = tl.ldmap(0, 1, t);
                  const Col yb = tile_col(tl, t, cfg.skew[t]);
                  const int sh = (q & 7) * 8;
                  const unsigned tb = (unsigned)((tm[q >> 3] >> sh) & 0xff), bb = (unsigned)((bm[q >> 3] >> sh) & 0xff), ab = (unsigned)((am[q >> 3] >> sh) & 0xff);
                  const int r0 = q * kStrip;
                  unsigned char *cp = s_cls + t * cstride + q;
                  if (!(tb | bb)) {                                       // (no candidate in the strip: nobody asks for its stale minima)
                     for (int ul = 0; ul < nus; ++ul) for (int pl = 0; pl < kDsPlanes; ++pl) cp[(ul * kDsPlanes + pl) * ppl] = 0;
                     continue; }
                  const u64 lt8 = *reinterpret_cast<const u64 *>(ldt + r0), lb8 = *reinterpret_cast<const u64 *>(ldb + r0);
                  int vl[8], vr[8];
                  #pragma unroll
                  for (int k = 0; k < 8; ++k) { vl[k] = yb[r0 + k - W + 1]; vr[k] = yb[r0 + k]; }
                  // the stale minimum's row along the strip (kNoSp: unknown; tile rows may be negative - the halo): a later pass over the same strip finds its own earlier result
                  constexpr int kNoSp = -0x40000000;
                  int sp[8];
                  if (first_pass) {
                     const int l0 = stale_ld(am, ldb, r0);
                     sp[0] = l0 ? r0 - W + l0 : kNoSp;
                     #pragma unroll
                     for (int k = 1; k < 8; ++k) {
                        const int lo = r0 + k - W + 1;
                        const bool rescan = ((ab >> k) & 1) || (sp[k - 1] != kNoSp && sp[k - 1] < lo);
                        sp[k] = rescan ? lo + (int)((lb8 >> (8 * k)) & 0xff) - 1 : sp[k - 1]; }
                     u64 out = 0;
                     #pragma unroll
                     for (int k = 0; k < 8; ++k) out |= (u64)(unsigned)(sp[k] != kNoSp ? sp[k] - (r0 + k - W + 1) + 1 : 0) << (8 * k);
                     *reinterpret_cast<u64 *>(ldb + r0) = out; }
                  else {
                     #pragma unroll
                     for (int k = 0; k < 8; ++k) { const int l = (int)((lb8 >> (8 * k)) & 0xff); sp[k] = l ? r0 + k - W + l : kNoSp; } }
                  int mt[8], tv[8], mb[8], bv[8];
                  unsigned unk = 0;                                       // rows whose bottom cannot be decided from the samples here (minimum out of reach, or at the window's edge)
                  #pragma unroll
                  for (int k = 0; k < 8; ++k) {
                     const int n = r0 + k, lo = n - W + 1;
                     const bool ct = (tb >> k) & 1, cb = (bb >> k) & 1;
                     const int tpos = ct ? lo + (int)((lt8 >> (8 * k)) & 0xff) - 1 : n, bpos = (cb && sp[k] != kNoSp) ? sp[k] : n;
                     const int tval = yb[tpos], bval = yb[bpos];
                     const int mv2 = max(vl[k], vr[k]), mn2 = min(vl[k], vr[k]);
                     mt[k] = ct ? tval - mv2 : -1; tv[k] = tval;
                     const bool edge = cb && sp[k] != kNoSp && (bpos <= lo || bpos >= n);
                     mb[k] = (cb && sp[k] != kNoSp && !edge) ? mn2 - bval : -1; bv[k] = -bval;
                     if (cb && (sp[k] == kNoSp || (edge && mn2 - bval > 0))) unk |= 1u << k; }
                  // whose band: the sub-segment that owns the rows; the next one where they are among its warm-up rows
                  const int jo = r0 < pad ? -1 : ((r0 - pad) / kDsSub > kDsJ - 1 ? kDsJ - 1 : (r0 - pad) / kDsSub);
                  int jw = jo + 1;
                  if (jw >= kDsJ || r0 + kStrip <= pad + jw * kDsSub - warm) jw = -1;
                  unsigned wmask = 0;                                     // rows of the strip inside that warm-up
                  if (jw >= 0) { const int w0r = pad + jw * kDsSub - warm; wmask = r0 >= w0r ? 0xffu : (0xffu << (w0r - r0)) & 0xffu; }
                  for (int ul = 0; ul < nus; ++ul) {
                     const DevParm &P = cfg.parm[cfg.uset_rep[us_all[u0 + ul]]];
                     const bool amp_on = P.min_peak != 0;
                     unsigned F = 0, Y = 0, D = 0, K = 0, Fw = 0, Kw = 0;
                     if (jo >= 0) {
                        const DsThr h = s_thr[(ul * kDsJ + jo) * ntrks + t];
                        const int qlo = amp_on ? h.q_lo : -0x40000000, qhi = amp_on ? h.q_hi : -0x40000000;
                        unsigned tS = 0, tN = 0, bS = 0, bN = 0;          // sure / not-miss, tops and bottoms
                        #pragma unroll
                        for (int k = 0; k < 8; ++k) {
                           tN |= (unsigned)(mt[k] > h.r_lo && tv[k] > qlo) << k; tS |= (unsigned)(mt[k] >= h.r_hi && tv[k] >= qhi) << k;
                           bN |= (unsigned)(mb[k] > h.r_lo && bv[k] > qlo) << k; bS |= (unsigned)(mb[k] >= h.r_hi && bv[k] >= qhi) << k; }
                        const unsigned tM = tN & ~tS, bM = bN & ~bS;
                        F = tS | (~tN & ~unk & bS); K = ~tN & ~unk & bN;      // (a bottom only where the top misses for sure; K: the bottom's business)
                        D = ~tS & (unk | (tM & bN));                          // an undecidable bottom behind a top that is not sure; a top maybe with a bottom in play
                        Y = ~D & ~F & (tM | (~tN & bM));
                        F &= ~D; K &= (F | Y); }
                     if (jw >= 0) {                                       // warm-up rows only bring the countdown into step (the join checks that they did): a maybe counts as a hit
                        // (counting it as a miss was tried for noisy tapes: on clean ones fifty times more joins fail - an extreme that stays
                        //  "maybe" for all its rows usually does fire)
                        const DsThr h = s_thr[(ul * kDsJ + jw) * ntrks + t];
                        const int qlo = amp_on ? h.q_lo : -0x40000000;
                        unsigned tN = 0, bN = 0;
                        #pragma unroll
                        for (int k = 0; k < 8; ++k) { tN |= (unsigned)(mt[k] > h.r_lo && tv[k] > qlo) << k; bN |= (unsigned)(mb[k] > h.r_lo && bv[k] > qlo) << k; }
                        Fw = (tN | bN) & wmask; Kw = ~tN & bN & wmask; }
                     unsigned char *o = cp + ul * kDsPlanes * ppl;
                     o[0] = (unsigned char)F; o[ppl] = (unsigned char)Y; o[2 * ppl] = (unsigned char)D; o[3 * ppl] = (unsigned char)K; o[4 * ppl] = (unsigned char)Fw; o[5 * ppl] = (unsigned char)Kw; } } }
            __syncthreads();
            if (prof) { const long long t2 = clock64(); t_sld += t2 - tq; tq = t2; }
            // (c) the lanes: (distinct set, sub-segment, track) - the countdown on the bits
            // (the lanes of this phase are few - 72 for one set on nine tracks - and what they do is sequential: their waves are the workgroup's critical path
            //  while its other waves wait at the barrier.  At a raised priority they issue ahead of the other workgroups' all-lane phases: G1 83.8 -> 81.9 ms,
            //  C4 240.5 -> 234.7.  Measured with it: the waves taking turns at carrying the walk - wave w always runs on SIMD w mod 4 - changes nothing.)
#ifndef RTFE_CPU_EMUL
            if ((int)threadIdx.x < nus * kDsJ * ntrks) __builtin_amdgcn_s_setprio(3);
#endif
            for (int task = threadIdx.x; task < nus * kDsJ * ntrks; task += blockDim.x) {
               const int q = fdn.div(task), t = task - q * ntrks, ul = q / kDsJ, j = q - ul * kDsJ;
               const int u = us_all[u0 + ul];
               const DevParm P = cfg.parm[cfg.uset_rep[u]];
               const bool amp_on = P.min_peak != 0;
               const DsThr h = s_thr[task];
               const float2 bd = s_bdl[task];
               const int ppl = ntrks * cstride;
               const u64 *pF = reinterpret_cast<const u64 *>(s_cls + (ul * kDsPlanes) * ppl + t * cstride);
               const u64 *pY = reinterpret_cast<const u64 *>(s_cls + (ul * kDsPlanes + 1) * ppl + t * cstride), *pD = reinterpret_cast<const u64 *>(s_cls + (ul * kDsPlanes + 2) * ppl + t * cstride);
               const u64 *pK = reinterpret_cast<const u64 *>(s_cls + (ul * kDsPlanes + 3) * ppl + t * cstride);
               const u64 *pFw = reinterpret_cast<const u64 *>(s_cls + (ul * kDsPlanes + 4) * ppl + t * cstride), *pKw = reinterpret_cast<const u64 *>(s_cls + (ul * kDsPlanes + 5) * ppl + t * cstride);
               const unsigned char *ldt = tl.ldmap(0, 0, t), *sld = tl.ldmap(0, 1, t);
               const Col yb = tile_col(tl, t, cfg.skew[t]);
               const int o0 = pad + j * kDsSub, o1 = o0 + kDsSub;
               const int lim = o1 < tile_lim ? o1 : tile_lim;
               unsigned char *slot = slots + (((size_t)(g * kDsJ + j) * nu + u) * ntrks + t) * (size_t)slot_bytes;
               DsRec *recs = reinterpret_cast<DsRec *>(slot + sizeof(DsHdr));
               int n_iter = 0, n_fire = 0;
               const long long tw0 = prof ? clock64() : 0;
               // warm-up: from o0 - warm, no countdown pending
               int n = o0 - warm, blind_until = n - 1;
               {  const int plim = o0 < lim ? o0 : lim;
                  #pragma nounroll
                  while (n < plim) {
                     const int wd = n >> 6;
                     const u64 c = pFw[wd] >> (n & 63);
                     if (!c) { n = (wd + 1) << 6; continue; }
                     n += __ffsll((long long)c) - 1;
                     if (n >= plim) break;
                     const int ld = ((pKw[n >> 6] >> (n & 63)) & 1) ? sld[n] : ldt[n];
                     blind_until = n + ld;                               // = the extreme's row + W (src/decoder.c:741)
                     n = blind_until + 1; ++n_iter; } }
               int start_blind = blind_until + 1 - o0; if (start_blind < 0) start_blind = 0;
               int count = 0, doubt = -1;
               int pk = -1, ppos = 0, pfirst = 0, plast = 0;              // a pending run of "maybe" rows: kind, the extreme's row, the run's first and last row
               // a record: the extreme fires for sure at row nf + nm, and may have fired at the nm maybe rows in front (their margins behind it).
               // A CONDITIONAL record (bit 19) has no sure row: a run of maybe rows that ended - the extreme fires at one of them or not at all,
               // which only the chain's exact thresholds can tell; the walk goes on as if it had not (k_dchain knows what to do if it has)
               auto put_rec = [&](int nf, int nm, int kind, int pos, bool cond) -> bool {
                  if (count + 1 + (nm + 3) / 4 > cap) return false;
                  const int fval = yb[pos], pv = yb[pos - 1], nx = yb[pos + 1];
                  int dp = kind == 0 ? fval - pv : pv - fval, dn = kind == 0 ? fval - nx : nx - fval;
                  dp = dp < 0 ? 0 : (dp > 255 ? 255 : dp); dn = dn < 0 ? 0 : (dn > 255 ? 255 : dn);
                  DsRec r = ds_pack(nf - o0, nm, kind, pos - nf + W, fval, dp, dn);
                  if (cond) r.w0 |= 1u << 19;
                  recs[count++] = r;
                  if (nm) count = ds_put_margins(recs, count, yb, W, nf, nm, kind, fval);
                  return true; };
               if (start_blind >= kDsNoJoin) start_blind = kDsNoJoin;
               else {
                  n = blind_until + 1 > o0 ? blind_until + 1 : o0;
                  #pragma nounroll
                  while (n < lim) {
                     const int wd = n >> 6;
                     const u64 wF = pF[wd], wY = pY[wd], wD = pD[wd];
                     const u64 c = (wF | wY | wD) >> (n & 63);
                     if (!c) { n = (wd + 1) << 6; continue; }
                     n += __ffsll((long long)c) - 1;
                     if (n >= lim) break;
                     ++n_iter;
                     const int bit = n & 63;
                     const bool isD = (wD >> bit) & 1, isY = (wY >> bit) & 1;
                     const int kind = (int)((pK[wd] >> bit) & 1);
                     const int pos = n - W + (kind ? (int)sld[n] : (int)ldt[n]);
                     const bool same = pk >= 0 && !isD && pk == kind && ppos == pos && n == plast + 1;      // the pending run's extreme, the very next row
                     if (pk >= 0 && !(same && (!isY || n - pfirst < kDsMaxMaybe))) {      // the run ends here without a sure row: a conditional record
                        if (!put_rec(pfirst, plast - pfirst + 1, pk, ppos, true)) { doubt = pfirst; break; }
                        pk = -1; }
                     if (isD) { doubt = n; break; }
                     if (isY) {
                        if (pk < 0) { pk = kind; ppos = pos; pfirst = n; }
                        plast = n; ++n; continue; }
                     const int nf = pk >= 0 ? pfirst : n;                 // it fires
                     if (!put_rec(nf, n - nf, kind, pos, false)) { doubt = nf; break; }
                     pk = -1; ++n_fire;
                     n = pos + W + 1; }
                  // a run of maybe rows that reaches the end of the own rows is followed on the rows behind them (a record belongs to the sub-segment of
                  // its FIRST maybe row; the next sub-segment's warm-up passes over the extreme as fired): those rows one at a time, against this
                  // lane's thresholds - a sure row of the same extreme completes the record; anything else leaves a conditional one
                  if (doubt < 0 && pk >= 0) {
                     bool fired = false;
                     if (plast == lim - 1 && lim == o1) {
                        const u64 *tm = tl.map(0, 0, t), *bm = tl.map(0, 1, t);
                        int plim = pfirst + kDsMaxMaybe + 1; if (plim > tile_lim) plim = tile_lim;
                        #pragma nounroll
                        for (n = lim; n < plim; ++n) {                    // (what the bits say there is the NEXT sub-segment's view: not used)
                           const int lo = n - W + 1, vl = yb[lo], vr = yb[n];
                           const bool ct = (tm[n >> 6] >> (n & 63)) & 1, cb = (bm[n >> 6] >> (n & 63)) & 1;
                           const int ls = cb ? (int)sld[n] : 0;
                           const int tpos = ct ? lo + ldt[n] - 1 : n, bpos = ls ? lo + ls - 1 : n;
                           const int tval = yb[tpos], bval = yb[bpos];
                           const int tc = ct ? ds_cls(h, amp_on, tval - max(vl, vr), tval) : kDsMiss;
                           const bool edge = ls && (bpos <= lo || bpos >= n);
                           const int bc = (ls && !edge) ? ds_cls(h, amp_on, min(vl, vr) - bval, -bval) : kDsMiss;
                           const bool uk = cb && (!ls || (edge && min(vl, vr) - bval > 0));
                           // the same extreme and nothing else in play: sure -> the record is complete; maybe -> on; anything else -> the run has ended
                           const bool mine_top = pk == 0 && ppos == tpos && !uk && bc == kDsMiss, mine_bot = pk == 1 && ppos == bpos && !uk && tc == kDsMiss;
                           const int cl = mine_top ? tc : (mine_bot ? bc : kDsMiss);
                           if (cl == kDsSure) { fired = put_rec(pfirst, n - pfirst, pk, ppos, false); if (!fired) doubt = pfirst; break; }
                           if (cl != kDsMaybe) break; } }
                     if (!fired && doubt < 0 && !put_rec(pfirst, plast - pfirst + 1, pk, ppos, true)) doubt = pfirst;
                     pk = -1; } }
               DsHdr hd; hd.count = (uint8_t)count; hd.start_blind = (uint8_t)start_blind;
               hd.doubt = (uint8_t)((doubt >= o0 && start_blind != kDsNoJoin) ? doubt - o0 : kDsNoDoubt); hd.flags = 0; hd.pad = 0; hd.s_lo = bd.x; hd.s_hi = bd.y;
               *reinterpret_cast<DsHdr *>(slot) = hd;
               if (prof) { atomicAdd(&dbg[3], (unsigned long long)n_iter); atomicAdd(&dbg[4], (unsigned long long)n_fire); atomicAdd(&dbg[5], 1ull); atomicAdd(&dbg[6], (unsigned long long)(clock64() - tw0)); } }
#ifndef RTFE_CPU_EMUL
            __builtin_amdgcn_s_setprio(0);
#endif
            __syncthreads();
            if (prof) { const long long t2 = clock64(); t_walk += t2 - tq; tq = t2; } } } }
   if (prof && threadIdx.x == 0) { atomicAdd(&dbg[0], (unsigned long long)t_load); atomicAdd(&dbg[1], (unsigned long long)t_scr); atomicAdd(&dbg[2], (unsigned long long)t_walk); atomicAdd(&dbg[7], (unsigned long long)t_sld); } }

// ------------------------------------------------------------------------------------------------
// k_dchain
// ------------------------------------------------------------------------------------------------
constexpr int kDcChunk = 32;                   // literal rows are worked through this many at a time ...
constexpr int kDcCache = kDcChunk + 56;        // ... from a lane's LDS cache: the chunk's rows and the window + 1 in front of them (W <= 50; LDS per wave decides how many chains run side by side)
struct DcRows {                // the detector's input straight from HBM: row n of track t after -invert and the deskew FIFO (src/decoder.c:820-830)
   const int16_t *col; int P, sgn, d; long long reset;
   __device__ __forceinline__ int operator()(long long n) const {
      const long long m = (n - reset < d) ? n : n - d;
      return sgn * (int)col[m * P]; } };

// k_dorder: the bursts by falling length (a counting sort over classes of 256 rows; ctl[i].pad = the burst at place i).  A chain is a
// dependent walk - what k_dchain takes is its longest chain's latency, and a wave is done when the longest of its lanes' chains is: its
// waves take 64 chains at a time from a queue in this order, the long ones first and like with like.
constexpr int kDoBins = 4096;
__global__ void __launch_bounds__(1024) k_dorder(const rtfe_burst *__restrict__ bursts, const BurstScratch *__restrict__ scratch, BurstCtl *__restrict__ ctl, long long nrows) {
   __shared__ int s_hist[kDoBins];
   __shared__ int lds[32];
   const int nb = scratch->nbursts, nt = scratch->nbursts_total;
   for (int i = threadIdx.x; i < kDoBins; i += 1024) s_hist[i] = 0;
   __syncthreads();
   auto bin = [&](int b) -> int {                                      // bin 0: the longest
      const long long end = b + 1 < nt ? bursts[b + 1].zone_end : nrows;
      long long k = (end - bursts[b].zone_end) >> 8;
      k = k < 0 ? 0 : (k > kDoBins - 1 ? kDoBins - 1 : k);
      return kDoBins - 1 - (int)k; };
   for (int b = threadIdx.x; b < nb; b += 1024) atomicAdd(&s_hist[bin(b)], 1);
   __syncthreads();
   int c[kDoBins / 1024], sum = 0;
   for (int j = 0; j < kDoBins / 1024; ++j) { c[j] = s_hist[threadIdx.x * (kDoBins / 1024) + j]; sum += c[j]; }
   int total;
   int off = block_excl_scan_1024(sum, lds, &total);
   for (int j = 0; j < kDoBins / 1024; ++j) { s_hist[threadIdx.x * (kDoBins / 1024) + j] = off; off += c[j]; }
   __syncthreads();
   for (int b = threadIdx.x; b < nb; b += 1024) ctl[atomicAdd(&s_hist[bin(b)], 1)].pad = b; }

#ifndef RTFE_DC_WAVES
#define RTFE_DC_WAVES 2
#endif
__global__ void __launch_bounds__(64, RTFE_DC_WAVES) k_dchain(const DevCfg *__restrict__ cfgp, const int16_t *__restrict__ rows, long long nrows, long long row_base,
                                               const rtfe_burst *__restrict__ bursts, BurstScratch *__restrict__ scratch, BurstCtl *__restrict__ ctl,
                                               uint32_t *__restrict__ counts, rtfe_event *__restrict__ events,
                                               const unsigned char *__restrict__ dead, const unsigned char *__restrict__ slots, long long ntiles, int ordered) {
   __shared__ float s_heights[64 * 10];
#ifdef RTFE_CPU_EMUL
   unsigned char *smem = g_dyn_smem;
#else
   extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
   // a lane's slot goes through LDS (unit j of lane l at [j][l]: 16 bytes - the header, then two records each): all of its loads from HBM are
   // in flight together, one round trip per sub-segment instead of one per record
   uint4 *s_slot = reinterpret_cast<uint4 *>(smem);
   // ... and so do the rows of a literal stretch: the detector's input of rows [cbase, cbase + kDcCache) of lane l at [k][l]
   int16_t *s_y = reinterpret_cast<int16_t *>(smem + (size_t)(cfgp->ds_slot < 144 ? 144 : cfgp->ds_slot) * 64);      // (nine units of a slot are always staged)
   const DevCfg &cfg = *cfgp;
   const int ntrks = cfg.ntrks, nu = cfg.nuset, nwalk = nu * ntrks;
   const int lane = threadIdx.x;
   const float mv = cfg.maxvolts, lsb = cfg.lsb_per_volt;
   const int nchains = scratch->nbursts * nwalk;
   float *heights = s_heights + lane * 10;
   const int slot_bytes = cfg.ds_slot;
   for (;;) {
      int cbase = 0;                                                   // the next 64 chains of the queue (k_dorder's order: the long bursts first)
      if (lane == 0) cbase = atomicAdd(&scratch->queue_walk, 64);
      cbase = __shfl(cbase, 0);
      if (cbase >= nchains) break;
      const int ci = cbase + lane < nchains ? cbase + lane : nchains - 1;
      const int place = ci / nwalk;
      const int b = ordered ? ctl[place].pad : place;
      const int wi = ci - place * nwalk, u = wi / ntrks, trk = wi - u * ntrks;
      const bool active = cbase + lane < nchains && ctl[b].status == kBurstReady;
      const rtfe_burst B = bursts[b];
      const int pidx = cfg.uset_rep[u];
      const unsigned pmask = cfg.uset_mask[u];
      const DevParm P = cfg.parm[pidx];
      const int cmode = cfg.mode, agc_off = cfg.agc_off;
      const int W = P.W, sc = P.screen;
      const bool amp_on = P.min_peak != 0;
      const long long reset = ctl[b].reset;
      const long long stop = chain_stop(cfg, bursts, ctl, b, scratch->nbursts_total, nrows);
      DcRows y; y.col = rows + cfg.trk_to_head[trk]; y.P = ntrks; y.sgn = cfg.invert ? -1 : 1; y.d = cfg.skew[trk]; y.reset = reset;
      const long long start = reset + trk;
      const long long fast_from = reset + W + max(trk, y.d) + 1;
      Walker w = {};
      w.agc_gain = 1.0f; w.v_avg_height = 4.0f;
      update_thresholds(w, P, lsb);
      for (int i = 0; i < 10; ++i) heights[i] = 0;
      const unsigned int cap = B.event_cap;
      rtfe_event *evb = events + B.event_base;
      // ---- one detection: refine_peak's code, the event (into the region of every parameter set this chain stands for), the AGC mirror ----
      bool done = !active, failed = false, dead_chain = false;
      auto fire = [&](long long n, int ld, int val, bool is_top, int iprev, int inext) {
         const int adjcode = refine_code(&cfg, val, iprev, inext, w.agc_gain, is_top);
         double t_peak = 0;
         if (cmode == RTFE_PE && !w.datablock && w.peakcount >= 68) {      // PE decides the end of its preamble from peak times (src/decode_pe.c:136-138): only there is the time read
            const float adj = adjcode == 1 ? -0.5f : (adjcode == 2 ? 0.5f : 0.0f);
            t_peak = time_of(&cfg, row_base + n) - ((float)(W - ld) - adj) * cfg.sample_deltat; }
         const float vp = volt(val, mv);
         if (w.nevents >= cap) w.flags |= RTFE_F_EVENT_OVERFLOW;
         else {
            rtfe_event e;
            e.sample = (uint32_t)(n - reset); e.v_peak = (cfg.invert && vp == 0.0f) ? -0.0f : vp; e.agc_gain = w.agc_gain;
            e.trk = (uint8_t)trk; e.flags = (uint8_t)((is_top ? 0 : 1) | (adjcode << 1)); e.left_distance = (uint8_t)ld;
            for (unsigned m = pmask; m; m &= m - 1) { const int p = __ffs((int)m) - 1; e.parmset = (uint8_t)p; evb[(size_t)(p * ntrks + trk) * cap + w.nevents] = e; } }
         if (is_top) w.v_top = vp; else w.v_bot = vp;
         ++w.nevents;
         agc_after_peak_m(w, cmode, agc_off, P, heights, is_top, t_peak);
         if (!(w.agc_gain > 0)) {                                        // src/decoder.c:782, at the first row behind the countdown (agc_fatal)
            w.flags |= RTFE_F_AGC_FATAL;
            if (w.nevents < cap) {
               rtfe_event e = {};
               e.sample = (uint32_t)(n + ld + 1 - reset); e.trk = (uint8_t)trk; e.flags = RTFE_EV_FATAL;
               for (unsigned m = pmask; m; m &= m - 1) { const int p = __ffs((int)m) - 1; e.parmset = (uint8_t)p; evb[(size_t)(p * ntrks + trk) * cap + w.nevents] = e; } }
            else w.flags |= RTFE_F_EVENT_OVERFLOW;
            ++w.nevents; dead_chain = true; return; }
         update_thresholds(w, P, lsb); };
      // the steady state the lean step is for: every callback adjusts the AGC (NRZI / GCR: the baseline fixed, src/decode_gcr.c:850,864;
      // PE: inside the data block, src/decode_pe.c:175,198), by the alpha filter or the minimum of the last agc_window heights
      const bool lean_ok = !agc_off && (P.agc_window != 0 || P.agc_alpha != 0) && cmode != RTFE_WW && cfg.ds_lean != 0;
      auto fatal_marker = [&](long long n, int ld) {                      // src/decoder.c:782, as in fire()
         w.flags |= RTFE_F_AGC_FATAL;
         if (w.nevents < cap) {
            rtfe_event e = {};
            e.sample = (uint32_t)(n + ld + 1 - reset); e.trk = (uint8_t)trk; e.flags = RTFE_EV_FATAL;
            for (unsigned m = pmask; m; m &= m - 1) { const int p = __ffs((int)m) - 1; e.parmset = (uint8_t)p; evb[(size_t)(p * ntrks + trk) * cap + w.nevents] = e; } }
         else w.flags |= RTFE_F_EVENT_OVERFLOW;
         ++w.nevents; dead_chain = true; };
      // ---- the literal detector (src/decoder.c:751-810 on the samples in HBM): state after the last row it processed ----
      int lmx = 0, lmn = 0, lcd = 0;
      bool synced = false;                                               // a forced rescan of a full, regular window was seen: from there the stale minimum is what k_dseg derives
      long long cache0 = 0;                                              // first row of the lane's LDS cache
      auto yc = [&](long long n) -> int { return (int)s_y[(int)(n - cache0) * 64 + lane]; };
      auto fill_cache = [&](long long first, long long last) {           // rows [first, last): every load on its way before the first is waited for
         cache0 = first;
         #pragma nounroll
         for (long long r = first; r < last; r += 16) {
            int t16[16];
            #pragma unroll
            for (int k = 0; k < 16; ++k) t16[k] = r + k < last ? y(r + k) : 0;
            #pragma unroll
            for (int k = 0; k < 16; ++k) if (r + k < last) s_y[(int)(r + k - first) * 64 + lane] = (int16_t)t16[k]; } };
      auto rescan_c = [&](long long lo, long long hi) { int mx = -0x7fffffff, mn = 0x7fffffff; for (long long j = lo; j <= hi; ++j) { const int v = yc(j); mx = max(mx, v); mn = min(mn, v); } lmx = mx; lmn = mn; };
      auto rescan = [&](long long lo, long long hi) { int mx = -0x7fffffff, mn = 0x7fffffff; for (long long j = lo; j <= hi; ++j) { const int v = y(j); mx = max(mx, v); mn = min(mn, v); } lmx = mx; lmn = mn; };
      auto lit_step = [&](long long n) {                                  // (rows n - W - 1 .. n are in the cache)
         if (n < start) return;
         if (n == start) { const int v = yc(n); lmx = v; lmn = v; lcd = 0; w.t_lastpeak = time_of(&cfg, row_base + n); return; }      // src/decoder.c:855-861
         const bool popped = n - start + 1 > W;
         const long long lo = popped ? n - W + 1 : start;
         const int vnow = yc(n);
         const int old_left = popped ? yc(n - W) : 0;
         if (vnow > lmx) lmx = vnow;
         if (old_left == lmx || old_left == lmn) { if (popped && old_left == lmx && n >= fast_from) synced = true; rescan_c(lo, n); }
         if (lcd) { --lcd; return; }
         const int vli = yc(lo);
         // the reference's comparisons (src/decoder.c:788-805): on the codes where clear, in floats inside the guard band
         bool top = above_by(lmx, vli, w.rise, w.rise_lo, w.rise_hi, mv) && above_by(lmx, vnow, w.rise, w.rise_lo, w.rise_hi, mv)
                    && (w.reqmin == 0 || lmx >= w.min_hi || (lmx > w.min_lo && volt(lmx, mv) > w.reqmin));
         bool bot = !top && below_by(lmn, vli, w.rise, w.rise_lo, w.rise_hi, mv) && below_by(lmn, vnow, w.rise, w.rise_lo, w.rise_hi, mv)
                    && (w.reqmin == 0 || -lmn >= w.min_hi || (-lmn > w.min_lo && volt(lmn, mv) < -w.reqmin));
         if (top || bot) {
            const int val = top ? lmx : lmn;
            long long p = lo;
            while (p <= n && yc(p) != val) ++p;
            if (p > n || p == lo || p == n) { w.flags |= RTFE_F_DETECTOR_FATAL; return; }      // src/decoder.c:709-710,748
            const int ld = (int)(p - lo) + 1;
            fire(n, ld, val, top, yc(p - 1), yc(p + 1));
            lcd = ld; } };
      // the literal state behind row n0 - 1 from the samples alone: the last forced rescan in front of it (the sample leaving the window
      // is its maximum), the bookkeeping of src/decoder.c:757-775 from there
      auto resync = [&](long long n0, long long blind_until) -> bool {
         long long lowlim = n0 - 1 - (4 * W + 64);
         if (lowlim < fast_from) lowlim = fast_from;
         long long h = n0 - 1;
         for (; h >= lowlim; --h) {
            const int v = y(h - W);
            int mxw = -0x7fffffff;
            for (int k = 1; k <= W; ++k) mxw = max(mxw, y(h - W + k));    // (no early way out: W independent loads, one round trip)
            if (mxw <= v) break; }
         if (h < lowlim) return false;
         rescan(h - W + 1, h);
         for (long long r = h + 1; r <= n0 - 1; ++r) {
            const int vnow = y(r), old_left = y(r - W);
            if (vnow > lmx) lmx = vnow;
            if (old_left == lmx || old_left == lmn) rescan(r - W + 1, r); }
         const long long c = blind_until - (n0 - 1);
         lcd = c > 0 ? (c > 0x3fffffff ? 0x3fffffff : (int)c) : 0;
         synced = true;
         return true; };
      auto in_band = [&](const float2 bd) -> bool {
         return w.rise >= P.rise * bd.x && w.rise <= P.rise * bd.y && (!amp_on || (w.reqmin >= P.min_peak * bd.x && w.reqmin <= P.min_peak * bd.y)); };
      long long cur = reset;                                              // next row to process
      long long blind_until = -1;                                         // (record mode)
      bool lit = true;
      uint4 pf[9]; long long pf_seg = -1; unsigned char pf_dead = 0;
      #pragma unroll
      for (int j = 0; j < 9; ++j) pf[j] = make_uint4(0, 0, 0, 0);
      long long no_join_before = 0;                                       // (behind a doubt the rest of its sub-segment is the literal detector's)
      long long rounds_left = (stop > reset ? (stop - reset) : 0) / 2 + 64;     // (every round moves `cur` on; belt and braces against a loop that does not)
      unsigned n_lit_rows = 0, n_rec_ev = 0, n_doubt = 0, n_nojoin = 0;
      int a_rlo = 0, a_rhi = 0, a_mlo = 0, a_mhi = 0;                      // the lean step's approximate integer guards (valid while w.thr_dirty)
      unsigned pc_rec = 0, pc_maybe = 0, pc_unclear = 0, pc_general = 0, pc_notinb = 0;      // (RTFE_DEBUG=8: records by the path they took)
      // (every lane of the wave goes through the same rounds: the emulator's ballot needs all of them)
      const bool prof = cfg.debug == 8;
      long long pt_join = 0, pt_rec = 0, pt_lit = 0, pn_rounds = 0, pn_litrounds = 0, ptq = 0;
      #pragma nounroll
      while (__ballot(!done) != 0ull) {
         if (prof) { ptq = clock64(); ++pn_rounds; }
         if (done) continue;
         if (dead_chain || cur >= stop) { done = true; continue; }
         if (--rounds_left < 0) { failed = true; done = true; continue; }
         const bool at_bnd = (cur % kDsSub) == 0;
         const long long seg = cur / kDsSub, tile = seg / kDsJ;
         if (w.thr_dirty) update_thresholds(w, P, lsb);                   // (the join check and the literal detector read the exact thresholds)
         bool join = false;
         int h_count = 0, h_sb = kDsNoJoin, h_doubt = kDsNoDoubt;
         float2 bd = make_float2(cfg.ds_sfloor, 3.0e38f);
         bool tile_dead = false;
         if (at_bnd && cur >= fast_from && cur >= no_join_before && (synced || !lit) && tile < ntiles) {
            unsigned char dflag;
            {  // (the slot's loads beside the flag's: a dead tile's slot holds nothing, and nothing of it is used).  The first nine units - a
               // whole slot of 15 records - usually arrived a round ago (pf: asked for when the sub-segment in front was joined)
               const uint4 *gp = reinterpret_cast<const uint4 *>(slots + (((size_t)seg * nu + u) * ntrks + trk) * (size_t)slot_bytes);
               const int nq = slot_bytes >> 4;
               uint4 tq[9];
               if (pf_seg == seg) {
                  dflag = pf_dead;
                  #pragma unroll
                  for (int j = 0; j < 9; ++j) tq[j] = pf[j]; }
               else {
                  dflag = dead[tile * cfg.nscreens + sc];
                  #pragma unroll
                  for (int j = 0; j < 9; ++j) tq[j] = gp[j]; }              // (nine units always: a slot is at least that long, and the pool is padded)
               #pragma unroll
               for (int j = 0; j < 9; ++j) s_slot[j * 64 + lane] = tq[j];
               #pragma nounroll
               for (int j0 = 9; j0 < nq; j0 += 4) {                        // (slots of 31 / 63 records)
                  uint4 t4[4];
                  #pragma unroll
                  for (int j = 0; j < 4; ++j) t4[j] = gp[min(j0 + j, nq - 1)];
                  #pragma unroll
                  for (int j = 0; j < 4; ++j) if (j0 + j < nq) s_slot[(j0 + j) * 64 + lane] = t4[j]; } }
            tile_dead = dflag != 0;
            if (tile_dead) join = in_band(bd);                             // nothing can rise above the screen: no countdown to agree on
            else {
               {  const uint4 h4 = s_slot[lane]; h_count = (int)(h4.x & 0xff); h_sb = (int)((h4.x >> 8) & 0xff); h_doubt = (int)((h4.x >> 16) & 0xff);
                  bd = make_float2(__uint_as_float(h4.y), __uint_as_float(h4.z)); }
               long long mb = lit ? (long long)lcd : blind_until - cur + 1;
               if (mb < 0) mb = 0;
               join = h_sb != kDsNoJoin && mb == (long long)h_sb && in_band(bd);
               if (!join) ++n_nojoin;
#ifdef RTFE_CPU_EMUL
               if (!join && getenv("RTFE_DS_TRACE")) fprintf(stderr, "nojoin b %d u %d trk %d row %lld lit %d: start_blind %d mine %lld band %.3f..%.3f rise %.4f (%.4f..%.4f)\n", b, u, trk, cur, (int)lit, h_sb, mb, bd.x, bd.y, w.rise, P.rise * bd.x, P.rise * bd.y);
#endif
               } }
         if (prof) { const long long t2 = clock64(); pt_join += t2 - ptq; ptq = t2; }
         if (join) {
            if (seg + 1 < ntiles * kDsJ) {                                  // the next sub-segment's slot: on its way while this one's records are worked through
               const uint4 *gn = reinterpret_cast<const uint4 *>(slots + (((size_t)(seg + 1) * nu + u) * ntrks + trk) * (size_t)slot_bytes);
               #pragma unroll
               for (int j = 0; j < 9; ++j) pf[j] = gn[j];
               pf_dead = dead[((seg + 1) / kDsJ) * cfg.nscreens + sc]; pf_seg = seg + 1; }
            if (lit) { blind_until = cur - 1 + lcd; lit = false; }
            const long long r0 = cur;
            if (blind_until < r0 - 1) blind_until = r0 - 1;                // ("not blind" has one value from here on)
            // (rows inside the record loop are ints relative to r0, the sub-segment's first row: a chain is one lane's dependent instructions, and
            //  64-bit row arithmetic doubles many of them; "_r" = relative)
            int blind_r = (int)(blind_until - r0);                         // >= -1
            int lane_blind_r = -1 + (tile_dead ? 0 : h_sb);                // the countdown of the lane that made this list (the join: in step with the chain's, or both over)
            const int stop_r = stop - r0 < (long long)(4 * kDsSub) ? (int)(stop - r0) : 4 * kDsSub;
            const uint32_t sample0 = (uint32_t)(r0 - reset);
            int next_r = kDsSub;                                          // where the chain goes on
            bool to_lit = false;
            // the event regions this chain writes: one set's in the usual case (no loop over the mask on the chain's path)
            const bool one_set = (pmask & (pmask - 1)) == 0;
            rtfe_event *ev1 = evb + (size_t)((__ffs((int)pmask) - 1) * ntrks + trk) * cap;
            if (!tile_dead) {
               // (the band's edges a little inside: an approximate threshold between them is an exact one inside the band)
               const float band_rlo = P.rise * bd.x * 1.00001f, band_rhi = P.rise * bd.y * 0.99999f, band_qlo = P.min_peak * bd.x * 1.00001f, band_qhi = P.min_peak * bd.y * 0.99999f;
               #pragma nounroll
               for (int k = 0; k < h_count; ++k) {
                  const uint4 r4 = s_slot[(1 + (k >> 1)) * 64 + lane];
                  DsRec rc; rc.w0 = (k & 1) ? r4.z : r4.x; rc.w1 = (k & 1) ? r4.w : r4.y;
                  const int nf = (int)(rc.w0 & 0xff), nm = (int)((rc.w0 >> 8) & 0xf), kind = (int)((rc.w0 >> 12) & 1), ld0 = (int)((rc.w0 >> 13) & 0x3f);
                  const int val = (int)(int16_t)(rc.w1 & 0xffff), dp = (int)((rc.w1 >> 16) & 0xff), dn = (int)(rc.w1 >> 24);
                  const bool cond = (rc.w0 >> 19) & 1;                      // a run of maybe rows without a sure one: it fires at one of them, or not at all
                  const int next_k = k + ((nm + 3) >> 2);                   // (the words with the maybe rows' margins lie behind the record)
                  if (nf >= stop_r) { next_r = stop_r; break; }
                  // Where the chain stands against the lane that made the list: in step as long as both countdowns end at the same row.  A
                  // conditional record that fires here (the lane went on as if it had not) puts the chain's countdown ahead: records whose rows
                  // it covers are passed over, and the list is taken up again at the first record the lane found with ITS countdown over no
                  // later than the chain's - it then looked at every row the chain is not blind for.  A lane that was blind where the chain
                  // is not proves nothing: the literal detector takes over.
                  const int last_row = cond ? nf + nm - 1 : nf + nm;        // the last row at which this record can fire
                  if (last_row <= blind_r) { if (!cond) lane_blind_r = nf + ld0; k = next_k; continue; }
                  if (lane_blind_r > blind_r) { next_r = blind_r + 1 > 0 ? blind_r + 1 : 0; to_lit = true; break; }
                  if (!cond) lane_blind_r = nf + ld0;                       // (= the extreme's row + W, whichever of the record's rows fires)
                  // which of the maybe rows fires: the reference's comparison against the nearer edge, from the margin the record carries;
                  // exact thresholds for that
                  int n = nf;
                  if (prof) ++pc_rec;
                  if (nm) {
                     if (prof) ++pc_maybe;
                     // First on integer guards alone - behind a lean step the approximate ones it left (a code wider on either side than the exact
                     // thresholds' own): a margin at or above the upper guard passes the reference's float comparison, one at or below the
                     // lower guard fails it.  Only a margin (or an extreme, for min_peak) between the guards needs the exact thresholds - two
                     // IEEE divisions that a wave of 64 chains would otherwise go through for nearly every record one of its lanes holds.
                     const int rlo = w.thr_dirty ? a_rlo : w.rise_lo, rhi = w.thr_dirty ? a_rhi : w.rise_hi, mlo = w.thr_dirty ? a_mlo : w.min_lo, mhi = w.thr_dirty ? a_mhi : w.min_hi;
                     const int av = kind == 0 ? val : -val;
                     const int m0 = blind_r >= nf ? blind_r + 1 - nf : 0;
                     bool need_exact = amp_on && av < mhi && av > mlo;
                     const bool amp_sure = !amp_on || av >= mhi;
                     n = cond ? -1 : nf + nm;
                     if (!need_exact && amp_sure) {
                        for (int m = m0; m < nm; ++m) {
                           const int kk = k + 1 + (m >> 2);
                           const uint4 e4 = s_slot[(1 + (kk >> 1)) * 64 + lane];
                           const uint32_t ew = (m & 2) ? ((kk & 1) ? e4.w : e4.y) : ((kk & 1) ? e4.z : e4.x);
                           const int mg = (int)((ew >> (16 * (m & 1))) & 0xffff);
                           if (mg >= rhi) { n = nf + m; break; }
                           if (mg > rlo) { need_exact = true; break; } } }
                     if (need_exact) {
                        if (prof) ++pc_notinb;
                        if (w.thr_dirty) update_thresholds(w, P, lsb);
                        n = cond ? -1 : nf + nm;
                        const bool amp_ok = w.reqmin == 0 || (kind == 0 ? (val >= w.min_hi || (val > w.min_lo && volt(val, mv) > w.reqmin))
                                                                       : (-val >= w.min_hi || (-val > w.min_lo && volt(val, mv) < -w.reqmin)));
                        for (int m = m0; m < nm; ++m) {
                           const int kk = k + 1 + (m >> 2);
                           const uint4 e4 = s_slot[(1 + (kk >> 1)) * 64 + lane];
                           const uint32_t ew = (m & 2) ? ((kk & 1) ? e4.w : e4.y) : ((kk & 1) ? e4.z : e4.x);
                           const int mg = (int)((ew >> (16 * (m & 1))) & 0xffff);
                           const bool hit = amp_ok && (kind == 0 ? above_by(val, val - mg, w.rise, w.rise_lo, w.rise_hi, mv) : below_by(val, val + mg, w.rise, w.rise_lo, w.rise_hi, mv));
                           if (hit) { n = nf + m; break; } } }
                     k = next_k;
                     if (n < 0) continue; }                                 // (a conditional record that does not fire: nothing happens)
                  if (n >= stop_r) { next_r = stop_r; break; }
                  const int ld = ld0 - (n - nf);
                  // ---- the record in steady state (NRZI / GCR: the baseline fixed, the alpha filter): straight-line code.  refine_peak's threshold
                  // from a 1-ulp reciprocal with a guard code more on either side (a neighbour inside the guard: the exact code); the thresholds only
                  // as far as the band check needs them (exact ones are made when something reads them) ----
                  if (lean_ok && (cmode == RTFE_PE ? w.datablock : (w.peakcount > 15 && w.v_avg_height_count == 0)) && w.nevents < cap) {
                     const int ti = (int)(0.005f * fast_rcp(w.agc_gain) * lsb);
                     const bool clear = (dp <= ti - 2 || dp >= ti + 3) && (dn <= ti - 2 || dn >= ti + 3) && ti >= 3 && ti + 4 < 255;
                     int adjcode;
                     if (clear) { const bool pclose = dp <= ti - 2, nclose = dn <= ti - 2; adjcode = (pclose && !nclose) ? 1 : ((nclose && !pclose) ? 2 : 0); }
                     else {
                        if (prof) ++pc_unclear;
                        int iprev = kind == 0 ? val - dp : val + dp, inext = kind == 0 ? val - dn : val + dn;
                        if (ti + 4 >= 255 || ti < 3) { const long long p = r0 + (n - W + ld); iprev = y(p - 1); inext = y(p + 1); }
                        adjcode = refine_code(&cfg, val, iprev, inext, w.agc_gain, kind == 0); }
                     const float vp = (float)((double)val * (1.0 / 32767.0)) * mv;      // == volt(val, mv) for every int16 code (the quotient's rounding: checked for all 65 536)
                     rtfe_event e;
                     e.sample = sample0 + (uint32_t)n; e.v_peak = (cfg.invert && vp == 0.0f) ? -0.0f : vp; e.agc_gain = w.agc_gain;
                     e.trk = (uint8_t)trk; e.flags = (uint8_t)(kind | (adjcode << 1)); e.left_distance = (uint8_t)ld;
                     if (one_set) { e.parmset = (uint8_t)(__ffs((int)pmask) - 1); ev1[w.nevents] = e; }
                     else for (unsigned m = pmask; m; m &= m - 1) { const int p = __ffs((int)m) - 1; e.parmset = (uint8_t)p; evb[(size_t)(p * ntrks + trk) * cap + w.nevents] = e; }
                     if (kind == 0) w.v_top = vp; else w.v_bot = vp;
                     ++w.nevents; ++w.peakcount;
                     const float lastheight = w.v_lasttop - w.v_lastbot;            // src/decoder.c:505-512 (both callbacks adjust in steady state: src/decode_gcr.c:850,864)
                     if (lastheight > 0) {
                        float gain;
                        if (P.agc_alpha) { gain = w.v_avg_height / lastheight; gain = P.agc_alpha * gain + (1 - P.agc_alpha) * w.agc_gain; }
                        else {                                              // src/decoder.c:514-531
                           heights[w.heightndx] = lastheight;
                           if (++w.heightndx >= P.agc_window) w.heightndx = 0;
                           float minheight = 99;
                           for (int i2 = 0; i2 < P.agc_window; ++i2) if (heights[i2] < minheight) minheight = heights[i2];
                           gain = w.v_avg_height / minheight; }
                        if (gain > 2.0f) gain = 2.0f;
                        w.agc_gain = gain; }
                     if (kind == 0) w.v_lasttop = vp; else w.v_lastbot = vp;
                     ++n_rec_ev;
                     blind_r = n + ld;
                     if (!(w.agc_gain > 0)) { w.thr_dirty = true; fatal_marker(r0 + n, ld); break; }
                     const float sa = w.v_avg_height * 0.25f * fast_rcp(w.agc_gain), ra = P.rise * sa, qa = P.min_peak * sa;
                     w.thr_dirty = true;
                     {  const int ar = (int)(ra * lsb), aq = (int)(qa * lsb);      // (as approx_thresholds: rtfe_kernels.hip)
                        a_rlo = ar - 2; a_rhi = ar + 3; a_mlo = aq - 2; a_mhi = aq + 3; }
                     const bool inb = ra >= band_rlo && ra <= band_rhi && (!amp_on || (qa >= band_qlo && qa <= band_qhi));
                     if (!inb) { update_thresholds(w, P, lsb); if (!in_band(bd)) { next_r = n + 1; to_lit = true; break; } }
                     continue; }
                  // ---- every other record (a block's first peaks, PE, the window AGC): the general step ----
                  if (prof) ++pc_general;
                  if (w.thr_dirty) update_thresholds(w, P, lsb);
                  const long long p = r0 + (n - W + ld);
                  // refine_peak's neighbours from their distances; a clamped distance only matters if the threshold reaches it
                  int iprev = kind == 0 ? val - dp : val + dp, inext = kind == 0 ? val - dn : val + dn;
                  const int ti = (int)floorf(0.005f / w.agc_gain * lsb);
                  if (ti + 2 >= 255 || ti < 2) { iprev = y(p - 1); inext = y(p + 1); }
                  fire(r0 + n, ld, val, kind == 0, iprev, inext);
                  ++n_rec_ev;
                  blind_r = n + ld;
                  if (dead_chain) break;
                  if (!in_band(bd)) { next_r = n + 1; to_lit = true; break; } }   // the thresholds left the band: what this list says about the rows behind n is not proven
               if (!dead_chain && !to_lit && next_r == kDsSub) {
                  // the list is through: a doubt ends it early; and a lane that was blind (a fire of its own the chain passed over) where the chain
                  // is not has not looked at those rows
                  int lit_from = kDsSub;
                  if (h_doubt != kDsNoDoubt) { lit_from = h_doubt; ++n_doubt; }
                  if (lane_blind_r > blind_r) { const int f = blind_r + 1 > 0 ? blind_r + 1 : 0; if (f < lit_from) lit_from = f; }
                  if (lit_from < kDsSub) { next_r = lit_from; to_lit = true; } } }
            blind_until = r0 + blind_r;
            const long long next = r0 + next_r;
            cur = next;
            if (prof) { const long long t2 = clock64(); pt_rec += t2 - ptq; ptq = t2; }
            if (to_lit && cur < stop) {
               if (!resync(cur, blind_until)) { failed = true; done = true; }
               no_join_before = (cur / kDsSub + 1) * kDsSub;
               lit = true; }
            continue; }
         // ---- literal rows up to the next sub-segment boundary ----
         if (!lit) { if (!resync(cur, blind_until)) { failed = true; done = true; continue; } lit = true; }
         long long end = (seg + 1) * kDsSub;
         if (end > stop) end = stop;
         #pragma nounroll
         for (long long c0 = cur; c0 < end && !dead_chain; c0 += kDcChunk) {
            const long long c1 = c0 + kDcChunk < end ? c0 + kDcChunk : end;
            long long first = c0 - W - 1; if (first < reset) first = reset;      // (nothing in front of the restart row is ever read)
            fill_cache(first, c1);
            #pragma nounroll
            for (long long n = c0; n < c1 && !dead_chain; ++n) lit_step(n); }
         n_lit_rows += (unsigned)(end - cur);
         cur = end;
         if (prof) { const long long t2 = clock64(); pt_lit += t2 - ptq; ptq = t2; ++pn_litrounds; } }
      if (prof && lane == 0) { atomicAdd(&scratch->dbg2[0], (unsigned long long)pt_join); atomicAdd(&scratch->dbg2[1], (unsigned long long)pt_rec); atomicAdd(&scratch->dbg2[2], (unsigned long long)pt_lit);
                               atomicAdd(&scratch->dbg2[3], (unsigned long long)pn_rounds); atomicAdd(&scratch->dbg2[4], (unsigned long long)pn_litrounds); atomicAdd(&scratch->dbg2[5], 1ull); }
      // ---- publish ----
      if (!active) continue;
      if (failed) atomicExch(&ctl[b].status, (int)kBurstNeedsFull);
      const unsigned int ne = w.nevents < cap ? w.nevents : cap;
      for (unsigned m = pmask; m; m &= m - 1) { const int p = __ffs((int)m) - 1; counts[((size_t)b * cfg.nparm + p) * ntrks + trk] = ne; }
      // (RTFE_F_SCREEN_UNDERFLOW - update_thresholds sets it - means nothing here: the lists are only used where the thresholds lie inside
      //  their band, which starts at the screen's level, and the literal detector has no screen)
      if (w.flags & ~(unsigned)RTFE_F_SCREEN_UNDERFLOW) atomicOr(&ctl[b].bflags, w.flags & ~(unsigned)RTFE_F_SCREEN_UNDERFLOW);
      if (n_lit_rows) atomicAdd(&scratch->dbg[0], (unsigned long long)n_lit_rows);
      if (n_rec_ev) atomicAdd(&scratch->dbg[1], (unsigned long long)n_rec_ev);
      if (n_doubt) atomicAdd(&scratch->why[0], (unsigned long long)n_doubt);
      if (n_nojoin) atomicAdd(&scratch->why[1], (unsigned long long)n_nojoin);
      if (failed) atomicAdd(&scratch->why[2], 1ull);
      if (prof) { atomicAdd(&scratch->why[3], (unsigned long long)pc_rec); atomicAdd(&scratch->why[4], (unsigned long long)pc_maybe); atomicAdd(&scratch->why[5], (unsigned long long)pc_unclear);
                  atomicAdd(&scratch->why[6], (unsigned long long)pc_general); atomicAdd(&scratch->why[7], (unsigned long long)pc_notinb); } } }

}  // namespace rtfe
