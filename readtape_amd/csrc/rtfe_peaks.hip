// rtfe_peaks.hip — the peak-record path of the MI355X analog front end (gfx950 / CDNA4).
//
//   k_peaks   dense and stateless, one workgroup per 1024-row tile.  The tape's bytes go into LDS as they are
//             (a flat copy in 16-byte vectors); the quiet map falls out of
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

// LDS written by some lanes of a wave, read by others of the SAME wave: the hardware executes a wave's LDS operations in order, the
// compiler must not move them across this point (tests/cpu_emul: the emulated lanes are threads and meet at a wave barrier)
#ifdef RTFE_CPU_EMUL
static inline void rtfe_wave_sync() { (void)__ballot(1); }
#else
__device__ __forceinline__ void rtfe_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
#endif

struct PkTile {
   const unsigned char *xs;     // LDS: the tape's rows as they are, row_bytes apart
   int row_bytes;
   int hl;                      // rows in front of the tile
   __device__ __forceinline__ int at(int r, int head) const {      // r relative to the tile's first row (>= -hl)
      const int q = r + hl;
      return *reinterpret_cast<const int16_t *>(xs + q * row_bytes + head * 2); } };

// LDS carve of k_peaks.  ONE definition for the kernel and for the host's sizing.
struct PkLds { unsigned xs, blk, wl, total; };
__host__ __device__ inline PkLds pk_lds_layout(int ntrks, int hl, int hr, int nbmax, int wave_cap) {
   PkLds L;
   const int nstrips = (hl + kPkTile + hr) / 16;
   const int npairs = (ntrks + 1) / 2;
   const int xb = (nbmax + 3) / 4;
   unsigned o = 0;
   L.xs = o;   o += (unsigned)nstrips * (16 * ntrks * 2) + 32;  o = (o + 15) & ~15u;
   L.blk = o;  o += (unsigned)npairs * 2 * 4 * (64 + 2 * xb) * 4;     // [pair][kind][4 * (64 + 2 xb)] dwords
   L.wl = o;   o += (unsigned)npairs * wave_cap * 2;                  // [pair][wave_cap] candidates of a wave, ordered by (head, row)
   L.total = (o + 15) & ~15u;
   return L; }

// four-entry tables that must stay in registers: no dynamic indexing
__device__ __forceinline__ int pick4(const int (&a)[4], int k) { return k == 0 ? a[0] : (k == 1 ? a[1] : (k == 2 ? a[2] : a[3])); }
__device__ __forceinline__ void add4(int (&a)[4], int k, int v) {
   #pragma unroll
   for (int i = 0; i < 4; ++i) a[i] += i == k ? v : 0; }

struct PkCtx {
   PkTile t;
   int W, lo_i, hi_i;            // window, screen threshold (margin > lo_i), sure threshold (margin >= hi_i)
   int tile_rows;                // rows of this tile that exist (<= kPkTile)
};

// What one candidate turns into: up to four records (without their margin entries), kept in registers until the wave knows where
// they go.  spill bit i: record i belongs to the next tile's spill list.  n > 4: more than fit.
struct PkSink { uint32_t w0[4], w1[4]; int n; unsigned spill; };
__device__ __forceinline__ void sink_add(PkSink &s, uint32_t w0, uint32_t w1, bool spill) {
   #pragma unroll
   for (int i = 0; i < 4; ++i) if (s.n == i) { s.w0[i] = w0; s.w1[i] = w1; }
   if (spill && s.n < 4) s.spill |= 1u << s.n;
   ++s.n; }

// margin of owner value `val` at row n: tops val - max(edges), bottoms min(edges) - val
__device__ __forceinline__ int pk_margin(const PkCtx &c, int head, int n, int val, bool top) {
   const int xl = c.t.at(n - c.W + 1, head), xr = c.t.at(n, head);
   return top ? val - max(xl, xr) : min(xl, xr) - val; }

// number of margin entries a record carries (k_chain reads the same encoding)
__host__ __device__ __forceinline__ int pk_nent(uint32_t w0, uint32_t w1) {
   if (w1 == 0xffff8000u) return 0;
   const int nlead = (int)((w0 >> 18) & 15u), nsure = (int)((w0 >> 22) & 63u), ntail = (int)((w0 >> 28) & 15u);
   return nsure == 63 ? (nlead << 4 | ntail) : nlead + ntail; }

__device__ __forceinline__ uint32_t pk_w0(int pos, bool top, int f, int nlead, int nsure, int ntail, bool spill) {
   return (uint32_t)(pos + 64 - (spill ? kPkTile : 0)) | ((top ? 0u : 1u) << 11) | ((uint32_t)(f - pos) << 12) | ((uint32_t)nlead << 18) | ((uint32_t)nsure << 22) | ((uint32_t)ntail << 28); }
__device__ __forceinline__ uint32_t pk_w1(int val, int prev, int nxt, bool top) {
   int dp = top ? val - prev : prev - val, dn = top ? val - nxt : nxt - val;
   dp = dp < -1 ? -1 : (dp > 254 ? 254 : dp); dn = dn < -1 ? -1 : (dn > 254 ? 254 : dn);
   return (uint32_t)(uint16_t)val | ((uint32_t)(dp + 1) << 16) | ((uint32_t)(dn + 1) << 24); }

// One record for owner `pos` over rows [ra, rb] (all of them rows at which the owner is what the detector tests): from the
// first row above the screen, explicit margins up to the first row at the sure level, the sure stretch, explicit margins
// for what is left up to the last row above the screen; if either explicit part exceeds 15 rows, every row is explicit.
// spill: the rows belong to the next tile (the record goes to its spill list, rows relative to THAT tile).
__device__ __forceinline__ void pk_describe(const PkCtx &c, PkSink &o, int head, int pos, int val, bool top, int ra, int rb, bool unknown, bool spill) {
   int n = ra;
   while (n <= rb && pk_margin(c, head, n, val, top) <= c.lo_i) ++n;
   if (n > rb) return;
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
   sink_add(o, pk_w0(pos, top, f, nlead, nsure, ntail, spill), unknown ? 0xffff8000u : pk_w1(val, c.t.at(pos - 1, head), c.t.at(pos + 1, head), top), spill); }

// the part of [ra, rb] inside this tile -> a record of the tile's own list, the rest -> a record of the next tile's spill list
// (an owner in the last W - 2 rows of this tile: pos + 64 - kPkTile >= 16 there)
__device__ __forceinline__ void pk_emit(const PkCtx &c, PkSink &o, int head, int pos, int val, bool top, int ra, int rb, bool unknown) {
   if (ra > rb) return;
   const int last = c.tile_rows - 1;
   if (ra <= last) pk_describe(c, o, head, pos, val, top, ra, rb < last ? rb : last, unknown, false);
   if (rb > last && c.tile_rows == kPkTile) pk_describe(c, o, head, pos, val, top, ra > last + 1 ? ra : last + 1, rb, unknown, true); }

// the margin entries of a record (lead rows, then tail rows; or every row), recomputed from the samples where the record goes:
// entry i of the record lives at end[-(i + 1)] (the entries of a slot grow from its back)
__device__ __forceinline__ void pk_entries(const PkCtx &c, int head, uint32_t w0, uint32_t w1, bool spill, uint16_t *end) {
   if (w1 == 0xffff8000u) return;
   const int pos = (int)(w0 & 0x7ffu) - 64 + (spill ? kPkTile : 0), f = pos + (int)((w0 >> 12) & 63u);
   const bool top = !((w0 >> 11) & 1u);
   const int val = (int)(int16_t)(w1 & 0xffffu);
   int nlead = (int)((w0 >> 18) & 15u), nsure = (int)((w0 >> 22) & 63u), ntail = (int)((w0 >> 28) & 15u);
   if (nsure == 63) { nlead = nlead << 4 | ntail; nsure = 0; ntail = 0; }
   for (int i = 0; i < nlead; ++i) { const int m = pk_margin(c, head, f + i, val, top); end[-(i + 1)] = (uint16_t)(m < 0 ? 0 : m); }
   for (int i = 0; i < ntail; ++i) { const int m = pk_margin(c, head, f + nlead + nsure + i, val, top); end[-(nlead + i + 1)] = (uint16_t)(m < 0 ? 0 : m); } }

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
      dom = dom && y <= v; sub = sub && y >= v; }
   return sub || (dom && c.t.at(r, head) <= v); }
// leftmost minimum of the window that ends at row r (the rescan of src/decoder.c:768-775)
__device__ __forceinline__ int pk_argmin(const PkCtx &c, int head, int r) {
   int best = r - c.W + 1, bv = c.t.at(best, head);
   for (int j = best + 1; j <= r; ++j) { const int v = c.t.at(j, head); if (v < bv) { bv = v; best = j; } }
   return best; }

// a top candidate, general walk: sample p is a strict maximum towards the left, non-strict towards the right
__device__ __forceinline__ void pk_top(const PkCtx &c, PkSink &o, int head, int p) {
   const int W = c.W;
   const int val = c.t.at(p, head);
   int J = 0;                                                       // x[p-1..p-J] < val
   while (J < W - 2 && c.t.at(p - J - 1, head) < val) ++J;
   int D = 0;                                                       // x[p+1..p+D] <= val
   while (D < W - 2 && c.t.at(p + D + 1, head) <= val) ++D;
   // rows n = p+k at which p is the FIRST maximum of the window [n-W+1, n] and lies strictly inside it
   pk_emit(c, o, head, p, val, true, p + max(1, W - 1 - J), p + D, false); }

// a bottom candidate, general walk: sample q is the true window minimum (first from the left) at rows [ra, rb]; what the
// reference tests there is its own minimum, refreshed only by rescans (src/decoder.c:765-775).
__device__ __attribute__((noinline)) void pk_bot(const PkCtx &c, PkSink &out, int head, int q) {
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
   for (int r = n0; r >= aq; --r) if (pk_async(c, head, r)) { pk_emit(c, out, head, q, val, false, n0, rb, false); return; }
   // none: the minimum the reference holds at n0 comes from further back.  Last forced rescan in front of aq, then
   // the chain of rescans the stale minimum itself forces when it leaves the window.
   int r0 = aq - 1;
   const int stop = aq - 1 - kPkBack;
   while (r0 > stop && !pk_async(c, head, r0)) --r0;
   if (r0 <= stop) { pk_emit(c, out, head, q, val, false, n0, rb, true); return; }
   int r1 = n0 + 1;                                                 // first forced rescan behind n0 (within the run)
   while (r1 <= rb && !pk_async(c, head, r1)) ++r1;
   int start = r0, o = pk_argmin(c, head, r0);
   for (int hop = 0; hop < kPkBack + 64; ++hop) {                  // (on a rising slope the minimum is the sample about to leave: a rescan per row)
      if (o == q) { pk_emit(c, out, head, q, val, false, max(n0, start), rb, false); return; }
      const int next = min(o + W, r1);                              // the epoch of owner o covers rows [start, next - 1]
      if (next - 1 >= n0) pk_emit(c, out, head, o, c.t.at(o, head), false, max(n0, start), min(rb, next - 1), false);
      if (next > rb) return;
      start = next; o = pk_argmin(c, head, next); }
   pk_emit(c, out, head, q, val, false, max(n0, start), rb, true); }

template <bool WIDE> struct PkMask { typedef uint32_t type; };
template <> struct PkMask<true> { typedef uint64_t type; };
__device__ __forceinline__ int pk_ctz(uint32_t m) { return m ? __ffs((int)m) - 1 : 32; }        // (count of trailing zeros; the width when none is set)
__device__ __forceinline__ int pk_ctz(uint64_t m) { return m ? __ffsll((long long)m) - 1 : 64; }
__device__ __forceinline__ int pk_clz(uint32_t m) { return m ? __clz((int)m) : 32; }
__device__ __forceinline__ int pk_clz(uint64_t m) { return m ? __clzll((long long)m) : 64; }

// ---- the common case in registers: every sample a candidate's rows can see is loaded with independent LDS reads (row order, so
// that every register index is static), and the run and its record follow from bit masks over the rows.
// WM >= W.  Returns false when the candidate needs the general walk above (a bottom whose first rows precede every forced rescan).
template <int WM>
__device__ __forceinline__ bool pk_fast(const PkCtx &c, PkSink &o, int head, int p, bool bot) {
   const int W = c.W;
   const unsigned char *base = c.t.xs + head * 2;
   const int rb_ = c.t.row_bytes;
   const int sg = bot ? -1 : 1;                                       // bottoms: the same on the negated signal
   // x[p + k] (right edge of row p + k) and x[p + k - W + 1] (its left edge), k = 0 .. W-2: only bits are kept
   // (left sample strictly below / right sample not above the extreme; margin above the screen / at the sure level), as sign
   // bits of differences: int16 operands cannot overflow
   typedef typename PkMask<(WM > 32)>::type mask_t;
   const unsigned char *pr = base + (p + c.t.hl) * rb_, *pl = pr - (W - 1) * rb_;
   const int v2 = sg * (int)*reinterpret_cast<const int16_t *>(pr);    // the extreme (negated for bottoms)
   mask_t lm = 0, rm = 0, lom = 0, him = 0;
   #pragma unroll 4
   for (int k = 0; k < W - 1; ++k) {
      const int r2 = sg * (int)*reinterpret_cast<const int16_t *>(pr + k * rb_);
      const int l2 = sg * (int)*reinterpret_cast<const int16_t *>(pl + k * rb_);
      const int mk = v2 - max(l2, r2);
      lm |= (mask_t)((uint32_t)(l2 - v2) >> 31) << k;
      rm |= (mask_t)(((uint32_t)(v2 - r2) >> 31) ^ 1u) << k;
      lom |= (mask_t)((uint32_t)(c.lo_i - mk) >> 31) << k;
      him |= (mask_t)(((uint32_t)(mk - c.hi_i) >> 31) ^ 1u) << k; }
   rm &= ~(mask_t)1;
   const mask_t one = 1;
   // J: consecutive left samples below the extreme, from distance 1 (k = W-2) outwards (tops: up to W-2, bottoms: W-1, i.e. k = 0 too)
   constexpr int MB = 8 * (int)sizeof(mask_t);
   const mask_t lsh = lm << (MB - 1 - (W - 2));                        // top bit = k = W-2
   int J = pk_clz((mask_t)~lsh);
   const int jmax = bot ? W - 1 : W - 2;
   if (J > jmax) J = jmax;
   int D = pk_ctz((mask_t)~(rm >> 1));                                  // consecutive right samples not above it, from k = 1
   if (D > W - 2) D = W - 2;
   const int ra = W - 1 - J > 1 ? W - 1 - J : 1;
   if (ra > D) return true;
   mask_t V = (((one << D) << 1) - 1) & ~((one << ra) - 1);
   mask_t C = V & lom;
   if (!C) return true;
   if (bot) {
      // the reference's minimum is this sample from the first forced rescan at or behind aq = q + max(0, W-1-J) on: the common case is
      // a rescan at the very first candidate row
      const int n0 = pk_ctz(C);
      if (!pk_async(c, head, p + n0)) return false;
      C &= ~((one << n0) - 1); }
   // ---- records: the rows of this tile, then the rows that belong to the next one ----
   const int klast = c.tile_rows - 1 - p;                             // last row offset inside this tile
   const int val = sg * v2;
   const uint32_t w1 = pk_w1(val, (int)*reinterpret_cast<const int16_t *>(pr - rb_), (int)*reinterpret_cast<const int16_t *>(pr + rb_), !bot);
   #pragma unroll
   for (int part = 0; part < 2; ++part) {
      mask_t Pm = C;
      if (part == 0) { if (klast < MB - 1) Pm &= klast < 0 ? (mask_t)0 : (mask_t)(((one << klast) << 1) - 1); }
      else { if (c.tile_rows != kPkTile || klast >= MB - 1) break; Pm &= klast < 0 ? (mask_t)~(mask_t)0 : (mask_t)~(((one << klast) << 1) - 1); }
      if (!Pm) continue;
      const int f = pk_ctz(Pm), l = MB - 1 - pk_clz(Pm);
      const int span = l - f + 1;
      int nlead = pk_ctz((mask_t)(him >> f));
      if (nlead > span) nlead = span;
      int nsure = nlead >= span ? 0 : pk_ctz((mask_t)~(him >> (f + nlead)));
      if (nsure > span - nlead) nsure = span - nlead;
      int ntail = span - nlead - nsure;
      if (nlead > 15 || ntail > 15 || nsure > 62) { nlead = span >> 4; ntail = span & 15; nsure = 63; }
      sink_add(o, pk_w0(p, !bot, p + f, nlead, nsure, ntail, part == 1), w1, part == 1); }
   return true; }

template <int WM>
__device__ __forceinline__ bool pk_eval(const PkCtx &c, PkSink &o, int head, int p, bool bot) {
   o.n = 0; o.spill = 0;
   if (pk_fast<WM>(c, o, head, p, bot)) return true;
   o.n = 0; o.spill = 0; pk_bot(c, o, head, p);
   return false; }

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
                                               unsigned char *__restrict__ pool_own, unsigned char *__restrict__ pool_spill, unsigned long long *__restrict__ nbytes,
                                               int pass, const unsigned int *__restrict__ dead, unsigned long long *__restrict__ dbg) {
#ifdef RTFE_CPU_EMUL
   unsigned char *smem = g_dyn_smem;
#else
   extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
   __shared__ unsigned int s_noisy;
   const DevCfg &cfg = *cfgp;
   const int ntrks = cfg.ntrks, npairs = (ntrks + 1) >> 1;
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
   const int HL = cfg.pk_hl, HR = cfg.pk_hr;
   const int row_bytes = ntrks * 2, strip_bytes = 16 * row_bytes;
   const int nstrips = (HL + kPkTile + HR) >> 4, xls = HL >> 4;
   const int nbmax = NB;
   const int xb = (nbmax + 3) >> 2;
   const int nblk = 4 * (64 + 2 * xb);
   const PkLds L = pk_lds_layout(ntrks, HL, HR, nbmax, cfg.pk_wave_cap);
   unsigned char *xs = smem + L.xs;
   uint32_t *blk = reinterpret_cast<uint32_t *>(smem + L.blk);
   PkCtx cx;
   cx.t.xs = xs; cx.t.row_bytes = row_bytes; cx.t.hl = HL;
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
      const bool prof = cfg.debug == 3 && tid == 0;
      long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, tk4 = 0;
      if (prof) tk0 = clock64();
      // ---- 1. the tape's bytes -> LDS strips; quiet groups on the way ----
      {
         const long long e_first = (t0 - HL) * ntrks;
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
               if (noisy) atomicOr(&s_noisy, 1u << ((vi - v_own0) / vpg));       // (quiet tiles do none; a tile inside a block pays a few LDS cycles)
               if (valid) {
                  if (inv) {                                                  // -invert: 0 - x on every half (src/readtape.c:1421)
                     v.x = (int)pk_addu(~(uint32_t)v.x, 0x00010001u); v.y = (int)pk_addu(~(uint32_t)v.y, 0x00010001u);
                     v.z = (int)pk_addu(~(uint32_t)v.z, 0x00010001u); v.w = (int)pk_addu(~(uint32_t)v.w, 0x00010001u); }
                  reinterpret_cast<int4 *>(xs)[vi] = v; } } }
         }
      __syncthreads();
      if (pass == 0) {
         unsigned int quiet = ~s_noisy & 0xffffu;
         for (int g = 0; g < 16; ++g) if (t0 + 64 * (g + 1) > nrows) quiet &= ~(1u << g);       // only complete groups can be quiet
         if (tid == 0) qmap[tile] = (uint16_t)quiet;
         if (quiet == 0xffffu) {                                                                // deferred (see above)
            for (int i = tid; i < cfg.nscreens * ntrks; i += nthreads) {
               PeakDir z; z.nrec = 0xfffe; z.nent = 0;
               dir_main[tile * cfg.nscreens * ntrks + i] = z;
               if (tile + 1 < ntiles) dir_spill[(tile + 1) * cfg.nscreens * ntrks + i] = z;
               if (tile == 0) { PeakDir e = {}; dir_spill[i] = e; } }
            __syncthreads();
            continue; } }

      if (prof) { tk1 = clock64(); atomicAdd(&dbg[0], (unsigned long long)(tk1 - tk0)); atomicAdd(&dbg[7], 1ull); }
      if (cfg.cut == 1) { __syncthreads(); continue; }                  // (RTFE_CUT, timing experiments: the copy alone)
      for (int sc = 0; sc < cfg.nscreens; ++sc) {
         const DevScreen S = cfg.screen[sc];
         if (prof) tk1 = clock64();
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
            x[0] = lds_pair(base - row_bytes);
            #pragma unroll
            for (int i = 0; i < 16; ++i) x[1 + i] = lds_pair(base + i * row_bytes);
            x[17] = lds_pair(base + strip_bytes);
            uint32_t *bo = blk + (pair * 2) * nblk + 4 * (strip + xb);
            #pragma unroll
            for (int k = 0; k < 4; ++k) {
               bo[k] = pk_min(pk_min(x[1 + 4 * k], x[2 + 4 * k]), pk_min(x[3 + 4 * k], x[4 + 4 * k]));
               bo[nblk + k] = pk_max(pk_max(x[1 + 4 * k], x[2 + 4 * k]), pk_max(x[3 + 4 * k], x[4 + 4 * k])); } }
         __syncthreads();
         if (prof) { tk2 = clock64(); atomicAdd(&dbg[1], (unsigned long long)(tk2 - tk1)); }
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
         if (prof) { tk3 = clock64(); atomicAdd(&dbg[2], (unsigned long long)(tk3 - tk2)); }
         // ---- 4. every wave on its own (one pair of heads).  The candidates of its 64 strips are compacted (prefix sums of the per-lane
         // counts) into a list ordered by (head, row); in rounds of 64 lane i evaluates candidate i, prefix sums number the records within
         // their lists, and records and margin entries go straight to the lists' slots in HBM.  More than pk_wave_cap candidates (noise
         // below the screen), or a list that outgrows its slot: the list is marked unavailable and the bursts that need it take the sample path. ----
         if (wave < npairs && cfg.cut != 2) {                              // (RTFE_CUT=2: stop behind the dense pre-filter)
            constexpr int WM = 4 * NB + 2;
            const int h_lo = 2 * pair, h_hi = 2 * pair + 1;
            const bool has_hi = h_hi < ntrks;
            const bool has_next = tile + 1 < ntiles;
            const uint32_t mlo = (tm | bm) & 0xffffu, mhi = has_hi ? (tm | bm) >> 16 : 0u;
            int tot[4] = {0, 0, 0, 0}, tote[4] = {0, 0, 0, 0};                 // records / entries per list: own lo, own hi, spill lo, spill hi
            const int cnt = __popc(mlo) | (__popc(mhi) << 16);
            int incl = cnt;
            #pragma unroll
            for (int s2 = 1; s2 < 64; s2 <<= 1) { const int y = __shfl_up(incl, s2); if (lane >= s2) incl += y; }
            const int totc = __shfl(incl, 63);
            const int n_lo = totc & 0xffff, ncw = n_lo + (totc >> 16);
            bool bad = ncw > cfg.pk_wave_cap;
            uint16_t *wlist = reinterpret_cast<uint16_t *>(smem + L.wl) + wave * cfg.pk_wave_cap;
            const int cap_own = cfg.pk_slot, cap_sp = cfg.pk_sslot;
            unsigned char *slot_lo = pool_own + ((tile * cfg.nscreens + sc) * ntrks + h_lo) * (size_t)cap_own, *slot_hi = slot_lo + cap_own;
            unsigned char *sp_lo = pool_spill + (((tile + 1) * cfg.nscreens + sc) * ntrks + h_lo) * (size_t)cap_sp, *sp_hi = sp_lo + cap_sp;
            if (!bad) {
               const int excl = incl - cnt;
               int o2 = excl & 0xffff;
               for (uint32_t m = mlo; m; m &= m - 1) { const int b2 = __ffs((int)m) - 1; wlist[o2++] = (uint16_t)((16 * strip + b2) | (((bm >> b2) & 1u) << 14)); }
               o2 = n_lo + (excl >> 16);
               for (uint32_t m = mhi; m; m &= m - 1) { const int b2 = __ffs((int)m) - 1; wlist[o2++] = (uint16_t)((16 * strip + b2) | (((bm >> (16 + b2)) & 1u) << 14) | 0x8000u); }
               rtfe_wave_sync();
               #pragma nounroll
               for (int r0 = 0; r0 < ncw; r0 += 64) {
                  const int i = r0 + lane;
                  PkSink sk; sk.n = 0; sk.spill = 0;
                  int half = 0;
                  bool easy = true;
                  if (i < ncw) {
                     const uint32_t cd = wlist[i];
                     half = (int)(cd >> 15);
                     easy = pk_eval<WM>(cx, sk, half ? h_hi : h_lo, (int)(cd & 0x3ffu), (cd >> 14) & 1u); }
                  if (cfg.debug == 3) { const u64 hb = __ballot(!easy); if (lane == 0) { atomicAdd(&dbg[4], (unsigned long long)__popcll(hb)); atomicAdd(&dbg[5], 1ull); atomicAdd(&dbg[6], (unsigned long long)(hb != 0)); } }
                  if (sk.n > 4) { bad = true; sk.n = 4; }
                  // records / entries this lane adds to each list: 4 x 16 bits in two words each
                  int vr0 = 0, vr1 = 0, ve0 = 0, ve1 = 0;
                  #pragma unroll
                  for (int j = 0; j < 4; ++j)
                     if (j < sk.n) {
                        const bool sp = (sk.spill >> j) & 1u;
                        const int nen = pk_nent(sk.w0[j], sk.w1[j]), sh = 16 * half;
                        if (sp) { vr1 += 1 << sh; ve1 += nen << sh; } else { vr0 += 1 << sh; ve0 += nen << sh; } }
                  int ir0 = vr0, ir1 = vr1, ie0 = ve0, ie1 = ve1;
                  #pragma unroll
                  for (int s2 = 1; s2 < 64; s2 <<= 1) {
                     const int y0 = __shfl_up(ir0, s2), y1 = __shfl_up(ir1, s2), y2 = __shfl_up(ie0, s2), y3 = __shfl_up(ie1, s2);
                     if (lane >= s2) { ir0 += y0; ir1 += y1; ie0 += y2; ie1 += y3; } }
                  // this lane's records: index within the list = the list's total so far + exclusive prefix + records of this lane in front
                  int myr[2] = {((ir0 - vr0) >> (16 * half)) & 0xffff, ((ir1 - vr1) >> (16 * half)) & 0xffff};      // [own, spill] of this lane's head
                  int mye[2] = {((ie0 - ve0) >> (16 * half)) & 0xffff, ((ie1 - ve1) >> (16 * half)) & 0xffff};
                  #pragma unroll
                  for (int j = 0; j < 4; ++j)
                     if (j < sk.n) {
                        const bool sp = (sk.spill >> j) & 1u;
                        const int k = (sp ? 2 : 0) + half, nen = pk_nent(sk.w0[j], sk.w1[j]);
                        const int ri = pick4(tot, k) + myr[sp], ei = pick4(tote, k) + mye[sp];
                        unsigned char *slot = sp ? (half ? sp_hi : sp_lo) : (half ? slot_hi : slot_lo);
                        const int cap = sp ? cap_sp : cap_own;
                        if ((!sp || has_next) && 8 * (ri + 1) + 2 * (ei + nen) <= cap) {
                           PeakRec rr; rr.w0 = sk.w0[j]; rr.w1 = sk.w1[j];
                           reinterpret_cast<PeakRec *>(slot)[ri] = rr;
                           pk_entries(cx, half ? h_hi : h_lo, rr.w0, rr.w1, sp, reinterpret_cast<uint16_t *>(slot + cap) - ei); }
                        ++myr[sp]; mye[sp] += nen; }
                  const int tr0 = __shfl(ir0, 63), tr1 = __shfl(ir1, 63), te0 = __shfl(ie0, 63), te1 = __shfl(ie1, 63);
                  tot[0] += tr0 & 0xffff; tot[1] += (tr0 >> 16) & 0xffff; tot[2] += tr1 & 0xffff; tot[3] += (tr1 >> 16) & 0xffff;
                  tote[0] += te0 & 0xffff; tote[1] += (te0 >> 16) & 0xffff; tote[2] += te1 & 0xffff; tote[3] += (te1 >> 16) & 0xffff; } }
            bad = __ballot(bad) != 0;
            // ---- 5. directory ----
            if (lane < 2 && (lane == 0 || has_hi)) {
               const int half = lane, head = half ? h_hi : h_lo;
               PeakDir dm, ds;
               dm.nrec = (uint16_t)tot[half]; dm.nent = (uint16_t)tote[half]; ds.nrec = (uint16_t)tot[2 + half]; ds.nent = (uint16_t)tote[2 + half];
               if (bad || 8 * tot[half] + 2 * tote[half] > cap_own || tot[half] >= 0xff00 || tote[half] >= 0xff00) dm.nrec = 0xffff;
               if (bad || 8 * tot[2 + half] + 2 * tote[2 + half] > cap_sp) ds.nrec = 0xffff;
               dir_main[(tile * cfg.nscreens + sc) * ntrks + head] = dm;
               if (has_next) dir_spill[((tile + 1) * cfg.nscreens + sc) * ntrks + head] = ds;
               if (tile == 0) { PeakDir z = {}; dir_spill[(size_t)sc * ntrks + head] = z; }
               if (cfg.debug && lane == 0 && !bad) atomicAdd(nbytes, (unsigned long long)(8 * (tot[0] + tot[1] + tot[2] + tot[3]) + 2 * (tote[0] + tote[1] + tote[2] + tote[3]))); } }
         if (prof) { tk4 = clock64(); atomicAdd(&dbg[3], (unsigned long long)(tk4 - tk3)); }
         __syncthreads(); } } }

}  // namespace rtfe
