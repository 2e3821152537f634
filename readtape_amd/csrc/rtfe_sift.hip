// rtfe_sift.hip — the dense half of the peak path of the MI355X analog front end (gfx950 / CDNA4).
//
//   k_sift   dense and stateless; PERSISTENT workgroups (a few per CU), each walking a contiguous run of 896-row tiles.  The
//            tape's bytes are read ONCE: 16-byte global loads of tile i+1 are in flight (registers) while tile i is worked on,
//            and go into LDS as they are (a flat copy: a track is a column of the tile).  Per tile:
//              - the quiet map of its 14 groups of 64 rows (k_quiet folded into the pass),
//              - one wave per PAIR of heads, one lane per 14-row strip (row stride 14 x 18 B = 63 dwords: odd, so the 64 lanes
//                of a column read hit 64 different LDS banks), both heads of a pair in the int16 halves of one register
//                (v_pk_max_i16 / v_pk_min_i16 / v_pk_sub_i16 clamp): local extremum + amplitude -> candidate SAMPLES,
//              - candidates compacted by wave prefix sums; one lane per candidate OWNER (the sample that is the window's
//                maximum, or the reference's possibly stale window minimum, while the detector's test rows pass over it)
//                derives everything lookfor_peak (src/decoder.c:751-810) can ask about that sample as ONE 8-byte record
//                (+ 2-byte margins for the rows whose verdict depends on the AGC state),
//              - the tile's lists are staged in LDS and leave in 16-byte coalesced stores.
//   k_quiet  the quiet map alone, for the scans that do not run k_sift (-zeros, PE / GCR sample path).
//
// Everything here is integer streaming work: no MFMA.  Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtfe_device.h"
#include "rtfe_pk.h"

namespace rtfe {

// ------------------------------------------------------------------------------------------------
// quiet map: bit c of word c>>6 = every sample of rows [64c, 64c+64) lies inside the quiet band
// (k_quiet: for the scans that do not run k_sift.  One wave per group of 64 rows.)
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
// LDS access with LDS-typed pointers: a generic pointer makes every access a FLAT instruction that counts against both
// memory counters; the emulator (tests/cpu_emul) has one address space.
// ------------------------------------------------------------------------------------------------
#ifdef RTFE_CPU_EMUL
typedef const unsigned char *lds_cp;
typedef unsigned char *lds_p;
static inline lds_p to_lds(unsigned char *p) { return p; }
static inline int lds_i16(lds_cp p) { int16_t v; memcpy(&v, p, 2); return v; }
static inline uint32_t lds_u32u(lds_cp p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline void rtfe_wave_sync() { (void)__ballot(1); }
#else
typedef const __attribute__((address_space(3))) unsigned char *lds_cp;
typedef __attribute__((address_space(3))) unsigned char *lds_p;
struct __attribute__((packed)) SfU32 { uint32_t v; };
__device__ __forceinline__ lds_p to_lds(unsigned char *p) { return (lds_p)p; }
__device__ __forceinline__ int lds_i16(lds_cp p) { return *reinterpret_cast<const __attribute__((address_space(3))) int16_t *>(p); }
// heads 2j and 2j+1 of one row: rows are 2 ntrks bytes apart, so the dword is only 2-byte aligned (one ds_read_b32 on gfx950)
__device__ __forceinline__ uint32_t lds_u32u(lds_cp p) { return reinterpret_cast<const __attribute__((address_space(3))) SfU32 *>(p)->v; }
// LDS written by some lanes of a wave, read by others of the SAME wave: the hardware executes a wave's LDS operations in order, the
// compiler must not move them across this point
__device__ __forceinline__ void rtfe_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
#endif

struct PkTile {
   lds_cp xs;                   // LDS: the tape's rows as they are, row_bytes apart
   int row_bytes;
   int hl;                      // rows in front of the tile
   __device__ __forceinline__ int at(int r, int head) const {      // r relative to the tile's first row (>= -hl); after -invert
      return sg * lds_i16(xs + (r + hl) * row_bytes + head * 2); }
   int sg;                      // -invert: -1 (the detector sees 0 - x, src/readtape.c:1421)
};

// LDS carve of k_sift.  ONE definition for the kernel and for the host's sizing.
struct SfLds { unsigned xs, wl, stage, tot, total; };
__host__ __device__ inline SfLds sf_lds_layout(int ntrks, int hl, int hr, int wave_cap, int hcap) {
   SfLds L;
   const int npairs = (ntrks + 1) / 2;
   unsigned o = 0;
   L.xs = o;    o += (unsigned)(hl + kSfTile + hr) * (unsigned)(ntrks * 2) + 32;  o = (o + 15) & ~15u;
   L.wl = o;    o += (unsigned)npairs * wave_cap * 2;  o = (o + 15) & ~15u;      // [pair][wave_cap] candidates of a wave, ordered by (head, row)
   L.stage = o; o += (unsigned)ntrks * hcap;                                      // [head][hcap] the tile's lists as they go to HBM
   L.tot = o;   o += (unsigned)ntrks * 8;                                         // [head] records, entries
   L.total = (o + 15) & ~15u;
   return L; }

struct PkCtx {
   PkTile t;
   int W, lo_i, hi_i;            // window, screen threshold (margin > lo_i), sure threshold (margin >= hi_i)
   int last;                     // last row that exists, relative to the tile's first row (the tape's end; else far away)
};

// What one candidate turns into: up to four records (without their margin entries), kept in registers until the wave knows where
// they go.  n > 4: more than fit.
struct PkSink { uint32_t w0[4], w1[4]; int n; };
__device__ __forceinline__ void sink_add(PkSink &s, uint32_t w0, uint32_t w1) {
   #pragma unroll
   for (int i = 0; i < 4; ++i) if (s.n == i) { s.w0[i] = w0; s.w1[i] = w1; }
   ++s.n; }

// margin of owner value `val` at row n: tops val - max(edges), bottoms min(edges) - val
__device__ __forceinline__ int pk_margin(const PkCtx &c, int head, int n, int val, bool top) {
   const int xl = c.t.at(n - c.W + 1, head), xr = c.t.at(n, head);
   return top ? val - max(xl, xr) : min(xl, xr) - val; }

// number of margin entries a record carries (the walkers read the same encoding)
__host__ __device__ __forceinline__ int pk_nent(uint32_t w0, uint32_t w1) {
   if (w1 == 0xffff8000u) return 0;
   const int nlead = (int)((w0 >> 18) & 15u), nsure = (int)((w0 >> 22) & 63u), ntail = (int)((w0 >> 28) & 15u);
   return nsure == 63 ? (nlead << 4 | ntail) : nlead + ntail; }

__device__ __forceinline__ uint32_t pk_w0(int pos, bool top, int f, int nlead, int nsure, int ntail) {
   return (uint32_t)(pos + kSfPosBias) | ((top ? 0u : 1u) << 11) | ((uint32_t)(f - pos) << 12) | ((uint32_t)nlead << 18) | ((uint32_t)nsure << 22) | ((uint32_t)ntail << 28); }
__device__ __forceinline__ uint32_t pk_w1(int val, int prev, int nxt, bool top) {
   int dp = top ? val - prev : prev - val, dn = top ? val - nxt : nxt - val;
   dp = dp < -1 ? -1 : (dp > 254 ? 254 : dp); dn = dn < -1 ? -1 : (dn > 254 ? 254 : dn);
   return (uint32_t)(uint16_t)val | ((uint32_t)(dp + 1) << 16) | ((uint32_t)(dn + 1) << 24); }

// One record for owner `pos` over rows [ra, rb] (all of them rows at which the owner is what the detector tests): from the
// first row above the screen, explicit margins up to the first row at the sure level, the sure stretch, explicit margins
// for what is left up to the last row above the screen; if either explicit part exceeds 15 rows, every row is explicit.
__device__ __forceinline__ void pk_describe(const PkCtx &c, PkSink &o, int head, int pos, int val, bool top, int ra, int rb, bool unknown) {
   if (rb > c.last) rb = c.last;
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
   sink_add(o, pk_w0(pos, top, f, nlead, nsure, ntail), unknown ? 0xffff8000u : pk_w1(val, c.t.at(pos - 1, head), c.t.at(pos + 1, head), top)); }

__device__ __forceinline__ void pk_emit(const PkCtx &c, PkSink &o, int head, int pos, int val, bool top, int ra, int rb, bool unknown) {
   if (ra > rb) return;
   pk_describe(c, o, head, pos, val, top, ra, rb, unknown); }

// the margin entries of a record (lead rows, then tail rows; or every row), recomputed from the samples where the record goes:
// entry i of the record lives at end[-(i + 1)] (the entries of a slot grow from its back)
__device__ __forceinline__ void pk_entries(const PkCtx &c, int head, uint32_t w0, uint32_t w1, lds_p end) {
   if (w1 == 0xffff8000u) return;
   const int pos = (int)(w0 & 0x7ffu) - kSfPosBias, f = pos + (int)((w0 >> 12) & 63u);
   const bool top = !((w0 >> 11) & 1u);
   const int val = (int)(int16_t)(w1 & 0xffffu);
   int nlead = (int)((w0 >> 18) & 15u), nsure = (int)((w0 >> 22) & 63u), ntail = (int)((w0 >> 28) & 15u);
   if (nsure == 63) { nlead = nlead << 4 | ntail; nsure = 0; ntail = 0; }
#ifdef RTFE_CPU_EMUL
   uint16_t *e16 = reinterpret_cast<uint16_t *>(end);
#else
   __attribute__((address_space(3))) uint16_t *e16 = reinterpret_cast<__attribute__((address_space(3))) uint16_t *>(end);
#endif
   for (int i = 0; i < nlead; ++i) { const int m = pk_margin(c, head, f + i, val, top); e16[-(i + 1)] = (uint16_t)(m < 0 ? 0 : m); }
   for (int i = 0; i < ntail; ++i) { const int m = pk_margin(c, head, f + nlead + nsure + i, val, top); e16[-(nlead + i + 1)] = (uint16_t)(m < 0 ? 0 : m); } }

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

// ---- the common case in registers: every sample a candidate's rows can see is loaded with independent LDS reads, and the run
// and its record follow from bit masks over the rows.
// WM >= W.  Returns false when the candidate needs the general walk above (a bottom whose first rows precede every forced rescan).
template <int WM>
__device__ __forceinline__ bool pk_fast(const PkCtx &c, PkSink &o, int head, int p, bool bot) {
   const int W = c.W;
   const int rb_ = c.t.row_bytes;
   const int sg = bot ? -c.t.sg : c.t.sg;                             // bottoms: the same on the negated signal
   // x[p + k] (right edge of row p + k) and x[p + k - W + 1] (its left edge), k = 0 .. W-2: only bits are kept
   // (left sample strictly below / right sample not above the extreme; margin above the screen / at the sure level), as sign
   // bits of differences: int16 operands cannot overflow
   typedef typename PkMask<(WM > 32)>::type mask_t;
   lds_cp pr = c.t.xs + head * 2 + (p + c.t.hl) * rb_, pl = pr - (W - 1) * rb_;
   const int v2 = sg * lds_i16(pr);                                   // the extreme (negated for bottoms)
   mask_t lm = 0, rm = 0, lom = 0, him = 0;
   #pragma unroll 4
   for (int k = 0; k < W - 1; ++k) {
      const int r2 = sg * lds_i16(pr + k * rb_);
      const int l2 = sg * lds_i16(pl + k * rb_);
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
   // rows behind the tape's end do not exist
   const int klast = c.last - p;
   if (klast < MB - 1) C &= klast < 0 ? (mask_t)0 : (mask_t)(((one << klast) << 1) - 1);
   if (!C) return true;
   if (bot) {
      // the reference's minimum is this sample from the first forced rescan at or behind aq = q + max(0, W-1-J) on: the common case is
      // a rescan at the very first candidate row
      const int n0 = pk_ctz(C);
      if (!pk_async(c, head, p + n0)) return false;
      C &= ~((one << n0) - 1); }
   const int val = bot ? -v2 : v2;                                      // (the sample as the detector sees it)
   const uint32_t w1 = pk_w1(val, c.t.sg * lds_i16(pr - rb_), c.t.sg * lds_i16(pr + rb_), !bot);
   const int f = pk_ctz(C), l = MB - 1 - pk_clz(C);
   const int span = l - f + 1;
   int nlead = pk_ctz((mask_t)(him >> f));
   if (nlead > span) nlead = span;
   int nsure = nlead >= span ? 0 : pk_ctz((mask_t)~(him >> (f + nlead)));
   if (nsure > span - nlead) nsure = span - nlead;
   int ntail = span - nlead - nsure;
   if (nlead > 15 || ntail > 15 || nsure > 62) { nlead = span >> 4; ntail = span & 15; nsure = 63; }
   sink_add(o, pk_w0(p, !bot, p + f, nlead, nsure, ntail), w1);
   return true; }

template <int WM>
__device__ __forceinline__ bool pk_eval(const PkCtx &c, PkSink &o, int head, int p, bool bot) {
   o.n = 0;
   if (pk_fast<WM>(c, o, head, p, bot)) return true;
   o.n = 0; pk_bot(c, o, head, p);
   return false; }

// ------------------------------------------------------------------------------------------------
// k_sift
// ------------------------------------------------------------------------------------------------

// the tape's bytes of tile `tile` (with its halo) into registers; rows outside the tape read as zeros
template <int kSfVec>
__device__ __forceinline__ void sf_fetch(int4 (&q)[kSfVec], const int16_t *__restrict__ rows, long long e_first, long long total_elem, int nvec, int tid, int nthreads) {
   #pragma unroll
   for (int k = 0; k < kSfVec; ++k) {
      const int vi = k * nthreads + tid;
      const long long ge = e_first + (long long)vi * 8;
      q[k] = make_int4(0, 0, 0, 0);
      if (vi < nvec) {
         if (ge >= 0 && ge + 8 <= total_elem) q[k] = *reinterpret_cast<const int4 *>(rows + ge);
         else if (ge + 8 > 0 && ge < total_elem) {                 // the tape's ends: sample by sample, zeros outside
            int e[8];
            #pragma unroll
            for (int j = 0; j < 8; ++j) { const long long g = ge + j; e[j] = (g >= 0 && g < total_elem) ? (int)(unsigned short)rows[g] : 0; }
            q[k] = make_int4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)); } } } }

// WM >= the widest window; NV = 16-byte vectors of the next tile a thread holds in registers (>= tile vectors / threads)
template <int WM, int MAXT, int NV>
__global__ void __launch_bounds__(MAXT) k_sift(const DevCfg *__restrict__ cfgp, const int16_t *__restrict__ rows, long long nrows, long long ntiles,
                                               unsigned int *__restrict__ qbits, PeakDir *__restrict__ dir, unsigned char *__restrict__ pool,
                                               unsigned long long *__restrict__ dbg) {
#ifdef RTFE_CPU_EMUL
   unsigned char *smem = g_dyn_smem;
#else
   extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
   __shared__ unsigned int s_noisy;
   __shared__ unsigned int s_bad[kMaxScreens];
   const DevCfg &cfg = *cfgp;
   const int ntrks = cfg.ntrks, npairs = (ntrks + 1) >> 1;
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
   const int HL = cfg.pk_hl, HR = cfg.pk_hr;
   const int row_bytes = ntrks * 2;
   const int hcap = cfg.pk_slot;
   const SfLds L = sf_lds_layout(ntrks, HL, HR, cfg.pk_wave_cap, hcap);
   unsigned char *xs = smem + L.xs;
   const lds_p xsl = to_lds(xs);
   const lds_p stage = to_lds(smem + L.stage);
   int *s_tot = reinterpret_cast<int *>(smem + L.tot);                 // [head][2]
   PkCtx cx;
   cx.t.xs = xsl; cx.t.row_bytes = row_bytes; cx.t.hl = HL; cx.t.sg = cfg.invert ? -1 : 1;
   const int nvec = (HL + kSfTile + HR) * ntrks / 8;                    // 16-byte vectors of a tile with its halo (HL, HR: multiples of 8)
   const int vpg = 8 * ntrks;                                           // ... per quiet group of 64 rows
   const int v_own0 = HL * ntrks / 8;
   const uint32_t qpk = pk_dup(cfg.quiet_i), q2 = 2u * (uint32_t)cfg.quiet_i;
   const long long total_elem = nrows * ntrks;
   // every workgroup walks a contiguous run of tiles (the halo rows a tile shares with its neighbour are then still in its CU's L1 / its XCD's L2)
   const long long per = (ntiles + gridDim.x - 1) / gridDim.x;
   const long long tile_lo = (long long)blockIdx.x * per, tile_hi = tile_lo + per < ntiles ? tile_lo + per : ntiles;
   const bool prof = cfg.debug == 3 && tid == 0;
   constexpr int kSfVec = NV;
   int4 q[kSfVec];
   if (tile_lo < tile_hi) sf_fetch(q, rows, (tile_lo * kSfTile - HL) * ntrks, total_elem, nvec, tid, nthreads);
   for (long long tile = tile_lo; tile < tile_hi; ++tile) {
      const long long t0 = tile * kSfTile;
      const long long lastl = nrows - 1 - t0;
      cx.last = lastl > 0x3fffffff ? 0x3fffffff : (int)lastl;
      long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0;
      if (prof) tk0 = clock64();
      // ---- 1. the prefetched bytes -> LDS; the next tile's loads go out at once ----
      #pragma unroll
      for (int k = 0; k < kSfVec; ++k) { const int vi = k * nthreads + tid; if (vi < nvec) reinterpret_cast<int4 *>(xs)[vi] = q[k]; }
      if (tid == 0) s_noisy = 0;
      if (tid < kMaxScreens) s_bad[tid] = 0;
      __syncthreads();
      if (tile + 1 < tile_hi) sf_fetch(q, rows, ((tile + 1) * kSfTile - HL) * ntrks, total_elem, nvec, tid, nthreads);
      if (prof) { tk1 = clock64(); atomicAdd(&dbg[0], (unsigned long long)(tk1 - tk0)); atomicAdd(&dbg[7], 1ull); }
      // ---- 2. quiet groups: 14 x 64 rows, flat 16-byte reads of the tile proper ----
      for (int vb = 0; vb < kSfGroups * vpg; vb += nthreads) {
         const int vi = vb + tid;
         bool noisy = false;
         if (vi < kSfGroups * vpg) {
            const int4 v = reinterpret_cast<const int4 *>(xs)[v_own0 + vi];
            const uint32_t m = pk_maxu(pk_maxu(pk_addu((uint32_t)v.x, qpk), pk_addu((uint32_t)v.y, qpk)),
                                       pk_maxu(pk_addu((uint32_t)v.z, qpk), pk_addu((uint32_t)v.w, qpk)));
            noisy = (m & 0xffffu) > q2 || (m >> 16) > q2; }
         const u64 nb = __ballot(noisy);
         if (lane == 0 && nb) {                                           // the 64 vectors of this ballot lie in at most three groups
            const int vfirst = vb + wave * 64;
            unsigned int bits = 0;
            for (int g = vfirst / vpg; g <= (vfirst + 63) / vpg && g < kSfGroups; ++g) {
               const int a = g * vpg - vfirst, b = a + vpg;
               const u64 ma = a <= 0 ? ~0ull : (a >= 64 ? 0ull : ~0ull << a), mb = b >= 64 ? ~0ull : (b <= 0 ? 0ull : ~(~0ull << b));
               if (nb & ma & mb) bits |= 1u << g; }
            if (bits) atomicOr(&s_noisy, bits); } }
      if (cfg.cut == 1) { __syncthreads(); continue; }                  // (RTFE_CUT, timing experiments: the copy + quiet map alone)
      for (int sc = 0; sc < cfg.nscreens; ++sc) {
         const DevScreen S = cfg.screen[sc];
         cx.W = S.W; cx.lo_i = S.rise_i; cx.hi_i = S.sure_i;
         if (prof) tk1 = clock64();
         // ---- 3. candidate samples: local extremum + amplitude, one lane per 14-row strip of a pair of heads ----
         uint32_t tm = 0, bm = 0;
         const int pair = wave, strip = lane;
         {
            const lds_cp base = xsl + (HL + kSfStrip * strip) * row_bytes + 4 * pair;
            // (-invert: tops and bottoms swap on the raw codes; "x > A" becomes "x < -A")
            const uint32_t at = S.minpk_i < 0 ? pk_dup(-32768) : pk_dup(S.minpk_i), ab = S.minpk_i < 0 ? pk_dup(32767) : pk_dup(-S.minpk_i);
            uint32_t yp = pk_max(lds_u32u(base - row_bytes), at), zp = pk_min(lds_u32u(base - row_bytes), ab);
            const uint32_t x0 = lds_u32u(base);
            uint32_t yc = pk_max(x0, at), zc = pk_min(x0, ab);
            uint32_t uy = pk_subs(yp, yc), dz = pk_subs(zc, zp);           // sign: rising into this row above the floor / falling into it below the ceiling
            #pragma unroll
            for (int i = 0; i < kSfStrip; ++i) {
               const uint32_t xn = lds_u32u(base + (i + 1) * row_bytes);
               const uint32_t yn = pk_max(xn, at), zn = pk_min(xn, ab);
               const uint32_t uyn = pk_subs(yc, yn), dzn = pk_subs(zn, zc);
               const uint32_t t = uy & ~uyn, b = dz & ~dzn;
               tm = (tm >> 1) | (t & kPkSigns);
               bm = (bm >> 1) | (b & kPkSigns);
               yc = yn; zc = zn; uy = uyn; dz = dzn; }
            tm = (tm >> (16 - kSfStrip)) & 0x3fff3fffu; bm = (bm >> (16 - kSfStrip)) & 0x3fff3fffu;
            if (cfg.invert) { const uint32_t s2 = tm; tm = bm; bm = s2; }
            if (2 * pair + 1 >= ntrks) { tm &= 0xffffu; bm &= 0xffffu; }       // odd track count: the last pair's upper half is the next row
            // rows that do not exist cannot own a run
            const long long r0 = t0 + kSfStrip * strip;
            if (r0 + kSfStrip > nrows) { const int keep = (int)(nrows - r0 > 0 ? nrows - r0 : 0); const uint32_t mk = (1u << keep) - 1u; tm &= mk | (mk << 16); bm &= mk | (mk << 16); } }
         if (prof) { tk2 = clock64(); atomicAdd(&dbg[1], (unsigned long long)(tk2 - tk1)); }
         // ---- 4. every wave on its own (one pair of heads).  The candidates of its 64 strips are compacted (prefix sums of the per-lane
         // counts) into a list ordered by (head, row); in rounds of 64 lane i evaluates candidate i, prefix sums number the records within
         // their lists, and records and margin entries go to the lists' staging slots in LDS.  More than pk_wave_cap candidates (noise
         // above the screen), or a list that outgrows its slot: the list is marked unavailable and the bursts that need it take the sample path. ----
         if (cfg.cut != 2) {                                               // (RTFE_CUT=2: stop behind the dense pre-filter)
            const int h_lo = 2 * pair, h_hi = 2 * pair + 1;
            const bool has_hi = h_hi < ntrks;
            const uint32_t mlo = (tm | bm) & 0xffffu, mhi = has_hi ? (tm | bm) >> 16 : 0u;
            int tot_lo = 0, tot_hi = 0, tote_lo = 0, tote_hi = 0;                 // records / entries per list
            const int cnt = __popc(mlo) | (__popc(mhi) << 16);
            int incl = cnt;
            #pragma unroll
            for (int s2 = 1; s2 < 64; s2 <<= 1) { const int y = __shfl_up(incl, s2); if (lane >= s2) incl += y; }
            const int totc = __shfl(incl, 63);
            const int n_lo = totc & 0xffff, ncw = n_lo + (totc >> 16);
            bool bad = ncw > cfg.pk_wave_cap;
#ifdef RTFE_CPU_EMUL
            uint16_t *wlist = reinterpret_cast<uint16_t *>(smem + L.wl) + wave * cfg.pk_wave_cap;
#else
            __attribute__((address_space(3))) uint16_t *wlist = reinterpret_cast<__attribute__((address_space(3))) uint16_t *>(to_lds(smem + L.wl)) + wave * cfg.pk_wave_cap;
#endif
            const lds_p slot_lo = stage + h_lo * hcap, slot_hi = slot_lo + hcap;
            if (!bad && ncw > 0) {
               const int excl = incl - cnt;
               int o2 = excl & 0xffff;
               for (uint32_t m = mlo; m; m &= m - 1) { const int b2 = __ffs((int)m) - 1; wlist[o2++] = (uint16_t)((kSfStrip * strip + b2) | (((bm >> b2) & 1u) << 14)); }
               o2 = n_lo + (excl >> 16);
               for (uint32_t m = mhi; m; m &= m - 1) { const int b2 = __ffs((int)m) - 1; wlist[o2++] = (uint16_t)((kSfStrip * strip + b2) | (((bm >> (16 + b2)) & 1u) << 14) | 0x8000u); }
               rtfe_wave_sync();
               #pragma nounroll
               for (int r0 = 0; r0 < ncw; r0 += 64) {
                  const int i = r0 + lane;
                  PkSink sk; sk.n = 0;
                  int half = 0;
                  bool easy = true;
                  if (i < ncw) {
                     const uint32_t cd = wlist[i];
                     half = (int)(cd >> 15);
                     easy = pk_eval<WM>(cx, sk, half ? h_hi : h_lo, (int)(cd & 0x3ffu), (cd >> 14) & 1u); }
                  if (cfg.debug == 3) { const u64 hb = __ballot(!easy); if (lane == 0) { atomicAdd(&dbg[4], (unsigned long long)__popcll(hb)); atomicAdd(&dbg[5], 1ull); atomicAdd(&dbg[6], (unsigned long long)(hb != 0)); } }
                  if (sk.n > 4) { bad = true; sk.n = 4; }
                  // records / entries this lane adds to its head's list: the two heads in the two halves of a word
                  int vr = 0, ve = 0;
                  #pragma unroll
                  for (int j = 0; j < 4; ++j) if (j < sk.n) { vr += 1; ve += pk_nent(sk.w0[j], sk.w1[j]); }
                  const int sh = 16 * half;
                  int ir = vr << sh, ie = ve << sh;
                  #pragma unroll
                  for (int s2 = 1; s2 < 64; s2 <<= 1) {
                     const int y0 = __shfl_up(ir, s2), y1 = __shfl_up(ie, s2);
                     if (lane >= s2) { ir += y0; ie += y1; } }
                  int myr = (((ir >> sh) & 0xffff) - vr) + (half ? tot_hi : tot_lo), mye = (((ie >> sh) & 0xffff) - ve) + (half ? tote_hi : tote_lo);
                  const lds_p slot = half ? slot_hi : slot_lo;
                  #pragma unroll
                  for (int j = 0; j < 4; ++j)
                     if (j < sk.n) {
                        const int nen = pk_nent(sk.w0[j], sk.w1[j]);
                        if (8 * (myr + 1) + 2 * (mye + nen) <= hcap) {
#ifdef RTFE_CPU_EMUL
                           uint32_t *rp = reinterpret_cast<uint32_t *>(slot) + 2 * myr;
#else
                           __attribute__((address_space(3))) uint32_t *rp = reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(slot) + 2 * myr;
#endif
                           rp[0] = sk.w0[j]; rp[1] = sk.w1[j];
                           pk_entries(cx, half ? h_hi : h_lo, sk.w0[j], sk.w1[j], slot + hcap - 2 * mye); }
                        ++myr; mye += nen; }
                  const int tr = __shfl(ir, 63), te = __shfl(ie, 63);
                  tot_lo += tr & 0xffff; tot_hi += (tr >> 16) & 0xffff; tote_lo += te & 0xffff; tote_hi += (te >> 16) & 0xffff; } }
            bad = __ballot(bad) != 0;
            if (lane < 2 && (lane == 0 || has_hi)) {
               const int nr = lane ? tot_hi : tot_lo, ne = lane ? tote_hi : tote_lo;
               const bool over = bad || 8 * nr + 2 * ne > hcap || nr >= 0xff00 || ne >= 0xff00;
               s_tot[2 * (h_lo + lane)] = over ? -1 : nr; s_tot[2 * (h_lo + lane) + 1] = over ? 0 : ne; } }
         if (prof) { tk3 = clock64(); atomicAdd(&dbg[2], (unsigned long long)(tk3 - tk2)); }
         __syncthreads();
         // ---- 5. the lists leave: records from the front of each head's slot, margin entries from its back, 16 bytes per lane;
         // the directory ----
         {
            unsigned char *gslot0 = pool + ((size_t)(tile * cfg.nscreens + sc) * ntrks) * (size_t)hcap;
            const int vps = hcap >> 4;                                         // 16-byte vectors per slot
            for (int vi = tid; vi < ntrks * vps; vi += nthreads) {
               const int h = vi / vps, v = vi - h * vps;
               const int nr = s_tot[2 * h], ne = s_tot[2 * h + 1];
               if (nr <= 0) continue;
               const int fv = (8 * nr + 15) >> 4, bv = (2 * ne + 15) >> 4;        // vectors in use at the front / at the back
               if (v < fv || v >= vps - bv)
                  reinterpret_cast<int4 *>(gslot0 + (size_t)h * hcap)[v] = reinterpret_cast<const int4 *>(smem + L.stage + h * hcap)[v]; }
            if (tid < ntrks) {
               PeakDir d; const int nr = s_tot[2 * tid];
               d.nrec = nr < 0 ? (uint16_t)0xffff : (uint16_t)nr; d.nent = (uint16_t)s_tot[2 * tid + 1];
               dir[(size_t)(tile * cfg.nscreens + sc) * ntrks + tid] = d;
               if (cfg.debug && nr > 0) atomicAdd(&dbg[3], (unsigned long long)(8 * nr + 2 * s_tot[2 * tid + 1])); } }
         __syncthreads(); }
      // ---- 6. the quiet map: bit (14 tile + g) of the tape's bit string (only complete groups can be quiet) ----
      if (tid == 0) {
         unsigned int quiet = ~s_noisy & ((1u << kSfGroups) - 1u);
         for (int g = 0; g < kSfGroups; ++g) if (t0 + 64 * (g + 1) > nrows) quiet &= ~(1u << g);
         if (quiet) {
            const long long bit0 = tile * kSfGroups;
            const u64 v = (u64)quiet << (bit0 & 31);
            atomicOr(&qbits[bit0 >> 5], (unsigned int)v);
            if (v >> 32) atomicOr(&qbits[(bit0 >> 5) + 1], (unsigned int)(v >> 32)); } }
      if (prof) { const long long tk4 = clock64(); atomicAdd(&dbg[8], (unsigned long long)(tk4 - tk0)); } } }

}  // namespace rtfe
