// rtfe_peaks.hip — the peak-record path of the MI355X analog front end (gfx950 / CDNA4).
//
//   k_peaks   dense and stateless, one workgroup per 1024-row tile.  The tape's bytes go into LDS as they are
//             (16-byte vectors, one 8-byte pad per 16-row strip against bank conflicts); the quiet map falls out of
//             the copy.  One lane = one 16-row strip of TWO neighbouring heads, a head per int16 half of a register
//             (v_pk_* arithmetic): local extremum + prominence against block minima/maxima + amplitude give a
//             conservative set of candidate SAMPLES (not rows).  The candidates are compacted by wave prefix sums and
//             each gets one lane that works out the rows at which lookfor_peak (src/decoder.c:751-810) would test it,
//             the margin of every such row against the window edges, and - for bottoms - whether the reference's
//             stale window minimum is this sample, from the forced rescans alone.  Out: 8-byte records +
//             2-byte margins, ~2.5 bytes per row on a 9-track NRZI tape.
//   k_zones   per burst: the restart row inside its quiet zone from the forced rescans of the zone's
//             last 256 rows.
//   k_chain   one wave per (burst, parameter set, track): the blind countdown, the AGC schedule of the block
//             decoders and the thresholds they feed, over the records; clean stretches of up to 64 runs are decided
//             by all lanes at once (only the three-flop gain recurrence is sequential) and verified, anything else
//             is walked run by run.  Events come out exactly as the reference's callbacks see them.
//
// Everything here is integer / fp32 streaming work: no MFMA.  Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtfe_device.h"
#include "rtfe_pk.h"

namespace rtfe {

// ------------------------------------------------------------------------------------------------
// quiet map: bit c of word c>>6 = every sample of rows [64c, 64c+64) lies inside the quiet band
// (k_quiet: for the scans that do not run k_peaks.  One wave per group of 64 rows.)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_quiet(const int16_t *__restrict__ rows, long long nrows, int ntrks, int quiet_i,
                                               u64 *__restrict__ qwords, long long nwords) {
   __shared__ unsigned int part[4];
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const long long ngroups = nrows / 64;                  // complete groups
   const int vpg = 8 * ntrks;                             // 16-byte vectors per group
   const uint32_t qpk = pk_dup(quiet_i);
   const uint32_t q2 = 2u * (uint32_t)quiet_i;
   for (long long w = blockIdx.x; w < nwords; w += gridDim.x) {
      unsigned int bits = 0;
      for (int k = 0; k < 16; ++k) {
         const long long c = w * 64 + wave * 16 + k;
         bool noisy = false;
         if (c < ngroups) {
            const int4 *src = reinterpret_cast<const int4 *>(rows + c * 64 * ntrks);
            uint32_t m = 0;
            for (int v = lane; v < vpg; v += 64) {
               const int4 q = src[v];
               m = pk_maxu(m, pk_maxu(pk_maxu(pk_addu((uint32_t)q.x, qpk), pk_addu((uint32_t)q.y, qpk)),
                                      pk_maxu(pk_addu((uint32_t)q.z, qpk), pk_addu((uint32_t)q.w, qpk)))); }
            noisy = (m & 0xffffu) > q2 || (m >> 16) > q2; }
         const u64 b = __ballot(noisy);
         if (c < ngroups && b == 0) bits |= 1u << k; }
      if (lane == 0) part[wave] = bits;
      __syncthreads();
      if (threadIdx.x == 0)
         qwords[w] = (u64)part[0] | ((u64)part[1] << 16) | ((u64)part[2] << 32) | ((u64)part[3] << 48);
      __syncthreads(); } }

// ------------------------------------------------------------------------------------------------
// k_peaks
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lds_pair(const unsigned char *p) {      // heads 2j and 2j+1 of one row (2-byte aligned: gfx950 reads it as one ds_read_b32)
   return (uint32_t)*reinterpret_cast<const uint16_t *>(p) | ((uint32_t)*reinterpret_cast<const uint16_t *>(p + 2) << 16); }

struct PkTile {
   const unsigned char *xs;     // LDS: strips of 16 rows, strip_bytes apart
   int strip_bytes, row_bytes;
   int hl;                      // rows in front of the tile
   __device__ __forceinline__ int at(int r, int head) const {      // r relative to the tile's first row (>= -hl)
      const int q = r + hl;
      return *reinterpret_cast<const int16_t *>(xs + (q >> 4) * strip_bytes + (q & 15) * row_bytes + head * 2); } };

// LDS carve of k_peaks.  ONE definition for the kernel and for the host's sizing.
struct PkLds { unsigned xs, blk, cand, slot, recs, rent, ents, total; };
__host__ __device__ inline PkLds pk_lds_layout(int ntrks, int hl, int hr, int nbmax, int cand_cap, int rec_cap, int ent_cap) {
   PkLds L;
   const int nstrips = (hl + kPkTile + hr) / 16;
   const int npairs = (ntrks + 1) / 2;
   const int xb = (nbmax + 3) / 4;
   unsigned o = 0;
   L.xs = o;   o += (unsigned)nstrips * (16 * ntrks * 2 + 8) + 32;  o = (o + 15) & ~15u;
   L.blk = o;                                                    // [pair][kind][4 * (64 + 2 xb)] dwords; dead once the masks exist
   const unsigned blk_bytes = (unsigned)npairs * 2 * 4 * (64 + 2 * xb) * 4;
   L.cand = o;                                                   // staging shares the block area
   unsigned s = o + (unsigned)cand_cap * 4;
   L.slot = s;  s += (unsigned)cand_cap * 8;                     // [cand][4] record indices (uint16); 0xFFFF = none; bit 15 of a valid index = spill
   s = (s + 7) & ~7u;
   L.recs = s;  s += (unsigned)rec_cap * 8;
   L.rent = s;  s += (unsigned)rec_cap * 2;                      // first entry of each record
   L.ents = s;  s += (unsigned)ent_cap * 2;
   const unsigned stage_end = s;
   o = (o + blk_bytes > stage_end ? o + blk_bytes : stage_end);
   L.total = (o + 15) & ~15u;
   return L; }

struct PkCtx {
   PkTile t;
   int W, lo_i, hi_i;            // window, screen threshold (margin > lo_i), sure threshold (margin >= hi_i)
   int tile_rows;                // rows of this tile that exist (<= kPkTile)
   // staging
   PeakRec *recs; uint16_t *rent, *ents;
   int *nrec, *nent;             // LDS counters
   int rec_cap, ent_cap;
   int *overflow;
};

// margin of owner value `val` at row n: tops val - max(edges), bottoms min(edges) - val
__device__ __forceinline__ int pk_margin(const PkCtx &c, int head, int n, int val, bool top) {
   const int xl = c.t.at(n - c.W + 1, head), xr = c.t.at(n, head);
   return top ? val - max(xl, xr) : min(xl, xr) - val; }

// number of margin entries a record carries (k_chain reads the same encoding)
__host__ __device__ __forceinline__ int pk_nent(uint32_t w0) {
   const int nlead = (int)((w0 >> 18) & 15u), nsure = (int)((w0 >> 22) & 63u), ntail = (int)((w0 >> 28) & 15u);
   return nsure == 63 ? (nlead << 4 | ntail) : nlead + ntail; }

// One record for owner `pos` over rows [ra, rb] (all of them rows at which the owner is what the detector tests): from the
// first row above the screen, explicit margins up to the first row at the sure level, the sure stretch, explicit margins
// for what is left up to the last row above the screen; if either explicit part exceeds 15 rows, every row is explicit.
// Returns the staging index of the record or -1 (no candidate row / no room -> *overflow).
__device__ __forceinline__ int pk_build(const PkCtx &c, int head, int pos, int val, bool top, int ra, int rb, bool unknown) {
   int n = ra;
   while (n <= rb && pk_margin(c, head, n, val, top) <= c.lo_i) ++n;
   if (n > rb) return -1;
   const int f = n;
   int l = rb;                                                      // last row above the screen
   while (l > f && pk_margin(c, head, l, val, top) <= c.lo_i) --l;
   int nlead = 0, nsure = 0, ntail = 0;
   if (unknown) { nsure = l - f + 1; }                              // (runs are shorter than 63 rows)
   else {
      while (n <= l && pk_margin(c, head, n, val, top) < c.hi_i) { ++n; ++nlead; }
      while (n <= l && pk_margin(c, head, n, val, top) >= c.hi_i) { ++n; ++nsure; }
      ntail = l - n + 1;
      if (nlead > 15 || ntail > 15 || nsure > 62) { const int all = l - f + 1; nlead = all >> 4; ntail = all & 15; nsure = 63; } }
   const int nent = unknown ? 0 : (nsure == 63 ? (nlead << 4 | ntail) : nlead + ntail);
   const int ri = atomicAdd(c.nrec, 1);
   const int e0 = nent ? atomicAdd(c.nent, nent) : 0;
   if (ri >= c.rec_cap || e0 + nent > c.ent_cap) { *c.overflow = 1; return -1; }
   if (nsure == 63) for (int i = 0; i < nent; ++i) { const int m = pk_margin(c, head, f + i, val, top); c.ents[e0 + i] = (uint16_t)(m < 0 ? 0 : m); }
   else if (!unknown) {
      for (int i = 0; i < nlead; ++i) { const int m = pk_margin(c, head, f + i, val, top); c.ents[e0 + i] = (uint16_t)(m < 0 ? 0 : m); }
      for (int i = 0; i < ntail; ++i) { const int m = pk_margin(c, head, f + nlead + nsure + i, val, top); c.ents[e0 + nlead + i] = (uint16_t)(m < 0 ? 0 : m); } }
   const int prev = c.t.at(pos - 1, head), nxt = c.t.at(pos + 1, head);
   int dp = top ? val - prev : prev - val, dn = top ? val - nxt : nxt - val;
   dp = dp < -1 ? -1 : (dp > 254 ? 254 : dp); dn = dn < -1 ? -1 : (dn > 254 ? 254 : dn);
   PeakRec r;
   r.w0 = (uint32_t)(pos + 64) | ((top ? 0u : 1u) << 11) | ((uint32_t)(f - pos) << 12) | ((uint32_t)nlead << 18) | ((uint32_t)nsure << 22) | ((uint32_t)ntail << 28);
   r.w1 = unknown ? 0xffff8000u : ((uint32_t)(uint16_t)val | ((uint32_t)(dp + 1) << 16) | ((uint32_t)(dn + 1) << 24));
   c.recs[ri] = r;
   c.rent[ri] = (uint16_t)e0;
   return ri; }

// the part of [ra, rb] inside this tile -> a record of the tile's own list, the rest -> a record of the next tile's
// spill list (its rows are relative to THAT tile: pos shifts by kPkTile).  slot[] collects up to four of them.
__device__ __forceinline__ void pk_emit(const PkCtx &c, uint16_t *slot, int &nslot, int head, int pos, int val, bool top, int ra, int rb, bool unknown) {
   if (ra > rb) return;
   const int last = c.tile_rows - 1;
   if (ra <= last) {
      const int ri = pk_build(c, head, pos, val, top, ra, rb < last ? rb : last, unknown);
      if (ri >= 0) { if (nslot < 4) slot[nslot++] = (uint16_t)ri; else *c.overflow = 1; } }
   if (rb > last && c.tile_rows == kPkTile) {
      const int ri = pk_build(c, head, pos, val, top, ra > last + 1 ? ra : last + 1, rb, unknown);
      if (ri >= 0) {
         c.recs[ri].w0 -= (uint32_t)kPkTile;                     // (bits 0-10 hold pos + 64 >= kPkTile - 48 + 64 here: no borrow)
         if (nslot < 4) slot[nslot++] = (uint16_t)(ri | 0x8000); else *c.overflow = 1; } } }

// "a rescan is forced at row r whatever happened before": the sample that leaves the window is
//   (a) the maximum of the old window AND not exceeded by the sample that enters (src/decoder.c:763-767: the new sample is
//       folded into pkww_maxv before old_left is compared with it; the maximum is always exact), or
//   (b) its minimum: the reference's own minimum is the value of a sample still inside the window, hence >= the true
//       minimum, and it was the minimum of an earlier window that already held the leaving sample (a sample that entered
//       later would have to be to the right of it), hence <= it: equal, and old_left == pkww_minv fires.
// x[r-W] >= all of x[r-W+1 .. r], or <= all of x[r-W+1 .. r-1].
__device__ __forceinline__ bool pk_async(const PkCtx &c, int head, int r) {
   const int s = r - c.W;
   const int v = c.t.at(s, head);
   bool dom = true, sub = true;
   for (int i = 1; i < c.W; ++i) {
      const int y = c.t.at(s + i, head);
      dom = dom && y <= v; sub = sub && y >= v;
      if (!dom && !sub) return false; }
   return sub || c.t.at(r, head) <= v; }
// leftmost minimum of the window that ends at row r (the rescan of src/decoder.c:768-775)
__device__ __forceinline__ int pk_argmin(const PkCtx &c, int head, int r) {
   int best = r - c.W + 1, bv = c.t.at(best, head);
   for (int j = best + 1; j <= r; ++j) { const int v = c.t.at(j, head); if (v < bv) { bv = v; best = j; } }
   return best; }

// a top candidate: sample p is a strict maximum towards the left, non-strict towards the right
__device__ __forceinline__ void pk_top(const PkCtx &c, uint16_t *slot, int &nslot, int head, int p) {
   const int W = c.W;
   const int val = c.t.at(p, head);
   int J = 0;                                                       // x[p-1..p-J] < val
   while (J < W - 2 && c.t.at(p - J - 1, head) < val) ++J;
   int D = 0;                                                       // x[p+1..p+D] <= val
   while (D < W - 2 && c.t.at(p + D + 1, head) <= val) ++D;
   // rows n = p+k at which p is the FIRST maximum of the window [n-W+1, n] and lies strictly inside it
   const int ra = p + max(1, W - 1 - J), rb = p + D;
   pk_emit(c, slot, nslot, head, p, val, true, ra, rb, false); }

// a bottom candidate: sample q is the true window minimum (first from the left) at rows [ra, rb]; what the reference
// tests there is its own minimum, refreshed only by rescans (src/decoder.c:765-775).
__device__ __forceinline__ void pk_bot(const PkCtx &c, uint16_t *slot, int &nslot, int head, int q) {
   const int W = c.W;
   const int val = c.t.at(q, head);
   int J = 0;                                                       // x[q-1..q-J] > val
   while (J < W - 1 && c.t.at(q - J - 1, head) > val) ++J;
   int D = 0;                                                       // x[q+1..q+D] >= val
   while (D < W - 2 && c.t.at(q + D + 1, head) >= val) ++D;
   const int aq = q + max(0, W - 1 - J);                             // first row at which q is the first window minimum
   const int ra = max(aq, q + 1), rb = q + D;
   if (ra > rb) return;
   int n0 = ra;                                                     // first row the true minimum would pass the screen at
   while (n0 <= rb && pk_margin(c, head, n0, val, false) <= c.lo_i) ++n0;
   if (n0 > rb) return;
   // a rescan at any row of [aq, n0] makes q the reference's minimum from then on (until q leaves the window)
   for (int r = n0; r >= aq; --r) if (pk_async(c, head, r)) { pk_emit(c, slot, nslot, head, q, val, false, n0, rb, false); return; }
   // none: the minimum the reference holds at n0 comes from further back.  Last forced rescan in front of aq, then
   // the chain of rescans the stale minimum itself forces when it leaves the window.
   int r0 = aq - 1;
   const int stop = aq - 1 - kPkBack;
   while (r0 > stop && !pk_async(c, head, r0)) --r0;
   if (r0 <= stop) { pk_emit(c, slot, nslot, head, q, val, false, n0, rb, true); return; }
   int r1 = n0 + 1;                                                 // first forced rescan behind n0 (within the run)
   while (r1 <= rb && !pk_async(c, head, r1)) ++r1;
   int start = r0, o = pk_argmin(c, head, r0);
   for (int hop = 0; hop < kPkBack + 64; ++hop) {                  // (on a rising slope the minimum is the sample about to leave: a rescan per row)
      if (o == q) { pk_emit(c, slot, nslot, head, q, val, false, max(n0, start), rb, false); return; }
      const int next = min(o + W, r1);                              // the epoch of owner o covers rows [start, next - 1]
      if (next - 1 >= n0) pk_emit(c, slot, nslot, head, o, c.t.at(o, head), false, max(n0, start), min(rb, next - 1), false);
      if (next > rb) return;
      start = next; o = pk_argmin(c, head, next); }
   pk_emit(c, slot, nslot, head, q, val, false, max(n0, start), rb, true); }

constexpr int kPkBatch = 4;

template <int NB>
__device__ __forceinline__ void pk_dense(const uint32_t *x, const uint32_t *bmn, const uint32_t *bmx, uint32_t lo_pk, uint32_t minpk_t, uint32_t minpk_b,
                                         uint32_t &tmask, uint32_t &bmask) {
   // windows of NB+1 blocks: wmn[i] = min(b[i .. i+NB]); block k of the strip is b[NB + k]
   uint32_t wmn[4 + NB], wmx[4 + NB];
   #pragma unroll
   for (int i = 0; i < 4 + NB; ++i) {
      uint32_t a = bmn[i], b = bmx[i];
      #pragma unroll
      for (int j = 1; j <= NB; ++j) { a = pk_min(a, bmn[i + j]); b = pk_max(b, bmx[i + j]); }
      wmn[i] = a; wmx[i] = b; }
   uint32_t tm = 0, bm = 0;
   #pragma unroll
   for (int k = 0; k < 4; ++k) {
      // a top must stand above the lowest sample on either side by more than the screen: x > max(minL, minR) + lo
      const uint32_t thr_t = pk_adds(pk_max(wmn[k], wmn[k + NB]), lo_pk);
      const uint32_t thr_b = pk_subs(pk_min(wmx[k], wmx[k + NB]), lo_pk);
      #pragma unroll
      for (int i = 1 + 4 * k; i <= 4 + 4 * k; ++i) {
         const uint32_t e0 = pk_subs(x[i - 1], x[i]), e1 = pk_subs(x[i], x[i + 1]);       // sign: x[i] > x[i-1] ; x[i+1] > x[i]
         const uint32_t f0 = pk_subs(x[i], x[i - 1]), f1 = pk_subs(x[i + 1], x[i]);       // sign: x[i] < x[i-1] ; x[i+1] < x[i]
         const uint32_t t = e0 & ~e1 & pk_subs(thr_t, x[i]) & pk_subs(minpk_t, x[i]);
         const uint32_t b = f0 & ~f1 & pk_subs(x[i], thr_b) & pk_subs(x[i], minpk_b);
         tm = (tm >> 1) | (t & kPkSigns);
         bm = (bm >> 1) | (b & kPkSigns); } }
   tmask = tm; bmask = bm; }

template <int NB, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT <= 448 ? 4 : 3)) k_peaks(const DevCfg *__restrict__ cfgp, const int16_t *__restrict__ rows, long long nrows, long long ntiles,
                                               uint16_t *__restrict__ qmap, PeakDir *__restrict__ dir_main, PeakDir *__restrict__ dir_spill,
                                               unsigned char *__restrict__ pool, unsigned long long pool_units, unsigned long long *__restrict__ pool_cursor,
                                               int pass, const unsigned int *__restrict__ dead) {
#ifdef RTFE_CPU_EMUL
   unsigned char *smem = g_dyn_smem;
#else
   extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
   __shared__ unsigned int s_noisy;
   __shared__ int s_headcnt[RTFE_MAXTRKS + 1], s_headbase[RTFE_MAXTRKS + 2];
   __shared__ int s_nrec, s_nent, s_overflow, s_ncand;
   __shared__ int s_wsum[16][4];
   __shared__ int s_tot[4];
   __shared__ int s_hoff[RTFE_MAXTRKS + 2][4];
   __shared__ unsigned long long s_blob;
   const DevCfg &cfg = *cfgp;
   const int ntrks = cfg.ntrks, npairs = (ntrks + 1) >> 1;
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
   const int HL = cfg.pk_hl, HR = cfg.pk_hr;
   const int row_bytes = ntrks * 2, strip_bytes = 16 * row_bytes + 8;
   const int nstrips = (HL + kPkTile + HR) >> 4, xls = HL >> 4;
   const int nbmax = NB;
   const int xb = (nbmax + 3) >> 2;
   const int nblk = 4 * (64 + 2 * xb);
   const PkLds L = pk_lds_layout(ntrks, HL, HR, nbmax, cfg.pk_cand_cap, cfg.pk_rec_cap, cfg.pk_ent_cap);
   unsigned char *xs = smem + L.xs;
   uint32_t *blk = reinterpret_cast<uint32_t *>(smem + L.blk);
   uint32_t *cand = reinterpret_cast<uint32_t *>(smem + L.cand);
   uint16_t *slots = reinterpret_cast<uint16_t *>(smem + L.slot);
   PkCtx cx;
   cx.t.xs = xs; cx.t.strip_bytes = strip_bytes; cx.t.row_bytes = row_bytes; cx.t.hl = HL;
   cx.recs = reinterpret_cast<PeakRec *>(smem + L.recs); cx.rent = reinterpret_cast<uint16_t *>(smem + L.rent); cx.ents = reinterpret_cast<uint16_t *>(smem + L.ents);
   cx.nrec = &s_nrec; cx.nent = &s_nent; cx.rec_cap = cfg.pk_rec_cap; cx.ent_cap = cfg.pk_ent_cap; cx.overflow = &s_overflow;
   const int vps = 2 * ntrks;                                        // 16-byte vectors per strip
   const int nvec = nstrips * vps;
   const int vpg = 8 * ntrks;                                        // ... per quiet group of 64 rows
   const int v_own0 = xls * vps, v_own1 = v_own0 + 16 * vpg;
   const uint32_t qpk = pk_dup(cfg.quiet_i), q2 = 2u * (uint32_t)cfg.quiet_i;
   const bool inv = cfg.invert != 0;
   const long long total_elem = nrows * ntrks;
   const long long per_xcd = (ntiles + 7) >> 3;

   for (long long bi = blockIdx.x; bi < per_xcd * 8; bi += gridDim.x) {
      // consecutive workgroups go to different XCDs (bi % 8); give every XCD a contiguous run of tiles so that the halo rows a
      // tile shares with its neighbour are still in that XCD's L2
      const long long tile = (bi & 7) * per_xcd + (bi >> 3);
      if (tile >= ntiles) continue;                                  // (uniform per workgroup)
      // pass 0 leaves tiles that are quiet from end to end alone (no block, mostly gap nothing ever walks); pass 1 - behind
      // k_bursts - does the few of them a burst's head or tail (or its neighbour's spill list) reaches into
      if (pass == 1) {
         const bool live = !((dead[tile >> 5] >> (tile & 31)) & 1u), live_next = tile + 1 < ntiles && !((dead[(tile + 1) >> 5] >> ((tile + 1) & 31)) & 1u);
         if (qmap[tile] != 0xffffu || !(live || live_next)) continue; }
      const long long t0 = tile * kPkTile;
      const int tile_rows = (int)(nrows - t0 < kPkTile ? nrows - t0 : kPkTile);
      cx.tile_rows = tile_rows;
      if (tid == 0) { s_noisy = 0; }
      __syncthreads();
      // ---- 1. the tape's bytes -> LDS strips; quiet groups on the way ----
      {
         const long long e_first = (t0 - HL) * ntrks;
         unsigned int noisy_bits = 0;
         for (int vbase = 0; vbase < nvec; vbase += kPkBatch * nthreads) {
            int4 q[kPkBatch];
            #pragma unroll
            for (int k = 0; k < kPkBatch; ++k) {
               const int vi = vbase + k * nthreads + tid;
               const long long ge = e_first + (long long)vi * 8;
               q[k] = make_int4(0, 0, 0, 0);
               if (vi < nvec) {
                  if (ge >= 0 && ge + 8 <= total_elem) q[k] = *reinterpret_cast<const int4 *>(rows + ge);
                  else if (ge + 8 > 0 && ge < total_elem) {                 // the tape's ends: sample by sample, zeros outside
                     int e[8];
                     #pragma unroll
                     for (int j = 0; j < 8; ++j) { const long long g = ge + j; e[j] = (g >= 0 && g < total_elem) ? (int)(unsigned short)rows[g] : 0; }
                     q[k] = make_int4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)); } } }
            #pragma unroll
            for (int k = 0; k < kPkBatch; ++k) {
               const int vi = vbase + k * nthreads + tid;
               const bool valid = vi < nvec;
               int4 v = q[k];
               const uint32_t m = pk_maxu(pk_maxu(pk_addu((uint32_t)v.x, qpk), pk_addu((uint32_t)v.y, qpk)),
                                          pk_maxu(pk_addu((uint32_t)v.z, qpk), pk_addu((uint32_t)v.w, qpk)));
               const bool own = valid && vi >= v_own0 && vi < v_own1;
               const bool noisy = own && ((m & 0xffffu) > q2 || (m >> 16) > q2);
               const int grp = own ? (vi - v_own0) / vpg : -1;
               // one ballot per quiet group present in the wave (consecutive vectors: at most three)
               int gcur = -2;
               for (;;) {
                  const u64 rest = __ballot(own && grp > gcur);
                  if (!rest) break;
                  const int src = __ffsll((long long)rest) - 1;
                  gcur = __shfl(grp, src);
                  const u64 nb = __ballot(noisy && grp == gcur);
                  if (nb) noisy_bits |= 1u << gcur; }
               if (valid) {
                  if (inv) {                                                  // -invert: 0 - x on every half (src/readtape.c:1421)
                     v.x = (int)pk_addu(~(uint32_t)v.x, 0x00010001u); v.y = (int)pk_addu(~(uint32_t)v.y, 0x00010001u);
                     v.z = (int)pk_addu(~(uint32_t)v.z, 0x00010001u); v.w = (int)pk_addu(~(uint32_t)v.w, 0x00010001u); }
                  const int st = vi / vps, wi = vi - st * vps;
                  unsigned char *dst = xs + st * strip_bytes + wi * 16;
                  reinterpret_cast<int2 *>(dst)[0] = int2{v.x, v.y};
                  reinterpret_cast<int2 *>(dst)[1] = int2{v.z, v.w}; } } }
         if (lane == 0 && noisy_bits) atomicOr(&s_noisy, noisy_bits); }
      __syncthreads();
      if (pass == 0) {
         unsigned int quiet = ~s_noisy & 0xffffu;
         for (int g = 0; g < 16; ++g) if (t0 + 64 * (g + 1) > nrows) quiet &= ~(1u << g);       // only complete groups can be quiet
         if (tid == 0) qmap[tile] = (uint16_t)quiet;
         if (quiet == 0xffffu) {                                                                // deferred (see above)
            for (int i = tid; i < cfg.nscreens * ntrks; i += nthreads) {
               PeakDir z = {}; z.nrec = 0xfffe;
               dir_main[tile * cfg.nscreens * ntrks + i] = z;
               if (tile + 1 < ntiles) dir_spill[(tile + 1) * cfg.nscreens * ntrks + i] = z;
               if (tile == 0) { PeakDir e = {}; dir_spill[i] = e; } }
            __syncthreads();
            continue; } }

      for (int sc = 0; sc < cfg.nscreens; ++sc) {
         const DevScreen S = cfg.screen[sc];
         cx.W = S.W; cx.lo_i = S.rise_i; cx.hi_i = S.sure_i;
         // ---- 2. block minima / maxima (4 rows x 2 heads per dword) ----
         uint32_t x[18];
         int pair = -1, strip = 0;
         bool dense = false;
         if (wave < npairs) { pair = wave; strip = lane; dense = true; }
         else if (wave == npairs && lane < 2 * xb * npairs) {               // the strips on either side whose blocks the windows reach into
            pair = lane / (2 * xb);
            const int h = lane - pair * (2 * xb);
            strip = h < xb ? h - xb : 64 + (h - xb); }
         if (pair >= 0) {
            const unsigned char *base = xs + (strip + xls) * strip_bytes + 4 * pair;
            x[0] = lds_pair(base - 8 - row_bytes);
            #pragma unroll
            for (int i = 0; i < 16; ++i) x[1 + i] = lds_pair(base + i * row_bytes);
            x[17] = lds_pair(base + strip_bytes);
            uint32_t *bo = blk + (pair * 2) * nblk + 4 * (strip + xb);
            #pragma unroll
            for (int k = 0; k < 4; ++k) {
               bo[k] = pk_min(pk_min(x[1 + 4 * k], x[2 + 4 * k]), pk_min(x[3 + 4 * k], x[4 + 4 * k]));
               bo[nblk + k] = pk_max(pk_max(x[1 + 4 * k], x[2 + 4 * k]), pk_max(x[3 + 4 * k], x[4 + 4 * k])); } }
         __syncthreads();
         // ---- 3. candidate samples: local extremum, prominence against the block windows, amplitude ----
         uint32_t tm = 0, bm = 0;
         if (dense) {
            uint32_t bmn[4 + 2 * NB], bmx[4 + 2 * NB];                      // (NB is the widest screen's: a longer window only loosens the pre-filter)
            const uint32_t *bi0 = blk + (pair * 2) * nblk + 4 * (strip + xb) - NB;
            const uint32_t lo_pk = pk_dup(S.rise_i);
            const uint32_t mt = S.minpk_i < 0 ? pk_dup(-32768) : pk_dup(S.minpk_i), mb = S.minpk_i < 0 ? pk_dup(32767) : pk_dup(-S.minpk_i);
            #pragma unroll
            for (int i = 0; i < 4 + 2 * NB; ++i) { bmn[i] = bi0[i]; bmx[i] = bi0[nblk + i]; }
            pk_dense<NB>(x, bmn, bmx, lo_pk, mt, mb, tm, bm);
            if (2 * pair + 1 >= ntrks) { tm &= 0xffffu; bm &= 0xffffu; }       // odd track count: the last pair's upper half is the next row
            // rows that do not exist, and the tape's first / last sample, cannot own a run
            const long long r0 = t0 + 16 * strip;
            if (r0 + 16 > nrows) { const int keep = (int)(nrows - r0 > 0 ? nrows - r0 : 0); const uint32_t mk = keep >= 16 ? 0xffffu : ((1u << keep) - 1u); tm &= mk | (mk << 16); bm &= mk | (mk << 16); } }
         __syncthreads();                                                  // the block area becomes the staging area
         // ---- 4. candidates per head; heads are worked off in groups that fit the staging area (one group almost always) ----
         const uint32_t mlo = (tm | bm) & 0xffffu, mhi = (tm | bm) >> 16;
         {
            int cnt = __popc(mlo) | (__popc(mhi) << 16);
            #pragma unroll
            for (int o = 32; o >= 1; o >>= 1) cnt += __shfl(cnt, (lane + o) & 63);          // (rotating all-reduce: every lane ends with the wave's sum)
            if (dense && lane == 0) { s_headcnt[2 * pair] = cnt & 0xffff; if (2 * pair + 1 < ntrks) s_headcnt[2 * pair + 1] = cnt >> 16; } }
         __syncthreads();
         for (int gh0 = 0; gh0 < ntrks;) {
            int gh1 = gh0 + 1, gsum = s_headcnt[gh0];                       // (every thread computes the same group)
            while (gh1 < ntrks && gsum + s_headcnt[gh1] <= cfg.pk_cand_cap) gsum += s_headcnt[gh1++];
            const bool cand_fit = gsum <= cfg.pk_cand_cap;
            const int ncand = cand_fit ? gsum : 0;
            if (tid == 0) { s_nrec = 0; s_nent = 0; s_overflow = 0; int o = 0; for (int h = gh0; h <= gh1; ++h) { s_headbase[h] = o; if (h < gh1) o += s_headcnt[h]; } }
            __syncthreads();
            // compaction: candidates of the group's heads ordered by (head, row)
            if (cand_fit) {
               const bool in_lo = dense && 2 * pair >= gh0 && 2 * pair < gh1, in_hi = dense && 2 * pair + 1 >= gh0 && 2 * pair + 1 < gh1 && 2 * pair + 1 < ntrks;
               const int cnt = (in_lo ? __popc(mlo) : 0) | ((in_hi ? __popc(mhi) : 0) << 16);
               int incl = cnt;
               #pragma unroll
               for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
               const int excl = incl - cnt;
               if (in_lo) {
                  int o = s_headbase[2 * pair] + (excl & 0xffff);
                  for (uint32_t m = mlo; m; m &= m - 1) { const int b2 = __ffs((int)m) - 1; cand[o++] = ((uint32_t)(2 * pair) << 16) | ((uint32_t)(16 * strip + b2) << 1) | ((bm >> b2) & 1u); } }
               if (in_hi) {
                  int o = s_headbase[2 * pair + 1] + (excl >> 16);
                  for (uint32_t m = mhi; m; m &= m - 1) { const int b2 = __ffs((int)m) - 1; cand[o++] = ((uint32_t)(2 * pair + 1) << 16) | ((uint32_t)(16 * strip + b2) << 1) | ((bm >> (16 + b2)) & 1u); } } }
            __syncthreads();
            // ---- 5. one lane per candidate: its rows, margins, and (bottoms) whose minimum the reference holds ----
            for (int ci = tid; ci < ncand; ci += nthreads) {
               const uint32_t cd = cand[ci];
               const int head = (int)(cd >> 16), p = (int)((cd >> 1) & 0x7fff);
               uint16_t sl[4] = {0xffff, 0xffff, 0xffff, 0xffff};
               int ns = 0;
               if (cd & 1u) pk_bot(cx, sl, ns, head, p); else pk_top(cx, sl, ns, head, p);
               uint16_t *so = slots + ci * 4;
               so[0] = sl[0]; so[1] = sl[1]; so[2] = sl[2]; so[3] = sl[3]; }
            __syncthreads();
            const bool ok = cand_fit && !s_overflow;
#ifdef RTFE_CPU_EMUL
            if (tid == 0 && getenv("RTFE_PK_DEBUG")) fprintf(stderr, "tile %lld sc %d heads %d-%d ncand %d fit %d ovf %d nrec %d nent %d\n", tile, sc, gh0, gh1, gsum, (int)cand_fit, s_overflow, s_nrec, s_nent);
#endif
            // ---- 6. order: records of the tile's own lists and of the lists spilled into the next tile, by (head, candidate) ----
            // counts per thread over a contiguous range of candidates, block exclusive scan, then the copy
            const int per = (ncand + nthreads - 1) / nthreads;
            const int c_lo = min(ncand, tid * per), c_hi = min(ncand, c_lo + per);
            int nm = 0, nme = 0, nsp = 0, nspe = 0;                             // main records / entries, spill records / entries
            if (ok)
               for (int ci = c_lo; ci < c_hi; ++ci)
                  for (int k = 0; k < 4; ++k) {
                     const uint16_t v = slots[ci * 4 + k];
                     if (v == 0xffff) break;
                     const int nl = pk_nent(cx.recs[v & 0x7fff].w0);
                     if (v & 0x8000) { ++nsp; nspe += nl; } else { ++nm; nme += nl; } }
            int sc4[4] = {nm, nme, nsp, nspe}, in4[4];
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
               int v = sc4[j];
               #pragma unroll
               for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(v, o); if (lane >= o) v += y; }
               in4[j] = v;
               if (lane == 63) s_wsum[wave][j] = v; }
            __syncthreads();
            if (tid < 4) { int o = 0; for (int w2 = 0; w2 * 64 < nthreads; ++w2) { const int t = s_wsum[w2][tid]; s_wsum[w2][tid] = o; o += t; } s_tot[tid] = o; }
            __syncthreads();
            const int tot_m = s_tot[0], tot_me = s_tot[1], tot_s = s_tot[2], tot_se = s_tot[3];
            // blob: [main records][spill records][main entries][spill entries], 16-byte units
            const unsigned rec_bytes = (unsigned)(tot_m + tot_s) * 8;
            const unsigned blob_bytes = ((rec_bytes + (unsigned)(tot_me + tot_se) * 2) + 15u) & ~15u;
            if (tid == 0) {
               unsigned long long b2 = ok ? atomicAdd(pool_cursor, (unsigned long long)(blob_bytes >> 4)) : 0;
               if (ok && b2 + (blob_bytes >> 4) > pool_units) b2 = ~0ull;
               s_blob = ok ? b2 : ~0ull; }
            __syncthreads();
            const unsigned long long blob = s_blob;
            const bool avail = blob != ~0ull && tot_m + tot_s < 0xff00 && tot_me + tot_se < 0xff00 && (rec_bytes >> 3) < 0xff00;
            if (avail) {
               unsigned char *bp = pool + blob * 16;
               PeakRec *orec = reinterpret_cast<PeakRec *>(bp);
               uint16_t *oent = reinterpret_cast<uint16_t *>(bp + rec_bytes);
               int om = s_wsum[wave][0] + in4[0] - nm, ome = s_wsum[wave][1] + in4[1] - nme;
               int os = tot_m + s_wsum[wave][2] + in4[2] - nsp, ose = tot_me + s_wsum[wave][3] + in4[3] - nspe;
               int hnext = gh0;
               while (hnext <= gh1 && s_headbase[hnext] < c_lo) ++hnext;
               for (int ci = c_lo; ci < c_hi; ++ci) {
                  while (hnext <= gh1 && s_headbase[hnext] == ci) { s_hoff[hnext][0] = om; s_hoff[hnext][1] = ome; s_hoff[hnext][2] = os - tot_m; s_hoff[hnext][3] = ose - tot_me; ++hnext; }
                  for (int k = 0; k < 4; ++k) {
                     const uint16_t v = slots[ci * 4 + k];
                     if (v == 0xffff) break;
                     const int ri = v & 0x7fff;
                     const PeakRec r = cx.recs[ri];
                     const int nl = pk_nent(r.w0), e0 = cx.rent[ri];
                     if (v & 0x8000) { orec[os++] = r; for (int i = 0; i < nl; ++i) oent[ose + i] = cx.ents[e0 + i]; ose += nl; }
                     else { orec[om++] = r; for (int i = 0; i < nl; ++i) oent[ome + i] = cx.ents[e0 + i]; ome += nl; } } } }
            // ---- 7. directory: per head, where its list starts (s_hoff: noted by the thread that copied the head's first candidate) ----
            __syncthreads();
            if (tid >= gh0 && tid < gh1) {
               PeakDir dm = {}, ds = {};
               if (avail) {
                  const int h0 = tid, h1 = tid + 1;
                  const int a0 = s_headbase[h0] >= ncand ? tot_m : s_hoff[h0][0], a1 = s_headbase[h1] >= ncand ? tot_m : s_hoff[h1][0];
                  const int b0 = s_headbase[h0] >= ncand ? tot_me : s_hoff[h0][1], b1 = s_headbase[h1] >= ncand ? tot_me : s_hoff[h1][1];
                  const int c0 = s_headbase[h0] >= ncand ? tot_s : s_hoff[h0][2], c1 = s_headbase[h1] >= ncand ? tot_s : s_hoff[h1][2];
                  const int d0 = s_headbase[h0] >= ncand ? tot_se : s_hoff[h0][3], d1 = s_headbase[h1] >= ncand ? tot_se : s_hoff[h1][3];
                  dm.blob = (uint32_t)blob; dm.rec_rel = (uint16_t)a0; dm.nrec = (uint16_t)(a1 - a0); dm.ent_rel = (uint16_t)b0; dm.nent = (uint16_t)(b1 - b0); dm.ents8 = (uint16_t)(rec_bytes >> 3);
                  ds.blob = (uint32_t)blob; ds.rec_rel = (uint16_t)(tot_m + c0); ds.nrec = (uint16_t)(c1 - c0); ds.ent_rel = (uint16_t)(tot_me + d0); ds.nent = (uint16_t)(d1 - d0); ds.ents8 = (uint16_t)(rec_bytes >> 3); }
               else { dm.nrec = 0xffff; ds.nrec = 0xffff; }
               dir_main[(tile * cfg.nscreens + sc) * ntrks + tid] = dm;
               if (tile + 1 < ntiles) dir_spill[((tile + 1) * cfg.nscreens + sc) * ntrks + tid] = ds;
               if (tile == 0) { PeakDir z = {}; dir_spill[(size_t)sc * ntrks + tid] = z; } }
            __syncthreads();
            gh0 = gh1; }
         __syncthreads(); } } }

}  // namespace rtfe
