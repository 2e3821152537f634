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
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const long long ngroups = nrows / 64;                  // complete groups
   const int vpg = 8 * ntrks;                             // 16-byte vectors per group (>= 8)
   const uint32_t qpk = pk_dup(quiet_i);
   const uint32_t q2 = 2u * (uint32_t)quiet_i;
   auto noisy4 = [&](const int4 q) -> bool {
      const uint32_t m = pk_maxu(pk_maxu(pk_addu((uint32_t)q.x, qpk), pk_addu((uint32_t)q.y, qpk)), pk_maxu(pk_addu((uint32_t)q.z, qpk), pk_addu((uint32_t)q.w, qpk)));
      return (m & 0xffffu) > q2 || (m >> 16) > q2; };
   // A group is quiet only if EVERY sample of its 64 rows is - so one that is not shows it in any part of it, and inside a block every part of a group carries
   // signal: a group's first 128 bytes (eight 16-byte vectors: seven rows of nine tracks) are looked at first, and the group is read in full only where they are
   // quiet - the gaps, a few per cent of a tape.  The map is bit for bit the same.  A WAVE makes a word of the map: eight lanes a group, eight load instructions
   // for the word's 64 first lines (all in flight together), a vote per instruction; the pass is bound by its dependent rounds (loads, votes), not by the lines -
   // so a round does as much as a wave's registers hold (round 5's first cut: a workgroup a word, sixteen groups a wave and two barriers a round - 1.35 ms per
   // 9.98e8 rows where this takes a third of the rounds).
   const int sub = lane >> 3, v8 = lane & 7;
   for (long long wv = (long long)blockIdx.x * 4 + wave; wv < nwords; wv += (long long)gridDim.x * 4) {
      const long long base = wv * 64;
      int4 q1[8];
      #pragma unroll
      for (int k = 0; k < 8; ++k) {
         const long long c = base + k * 8 + sub;
         q1[k] = make_int4(0, 0, 0, 0);
         if (c < ngroups) q1[k] = reinterpret_cast<const int4 *>(rows + c * 64 * ntrks)[v8]; }
      u64 cand = 0;                                                    // groups whose first line is quiet
      #pragma unroll
      for (int k = 0; k < 8; ++k) {
         const u64 b = __ballot(noisy4(q1[k]));                        // byte g: the eight lanes of group base + 8 k + g
         u64 z = ~b;                                                   // a byte of ones = a quiet first line
         z &= z >> 4; z &= z >> 2; z &= z >> 1;                       // bit 8 g = all eight bits of byte g
         z &= 0x0101010101010101ull;
         // gather bits 0, 8, 16 ... 56 into bits 0 .. 7
         const u64 g8 = (z * 0x0102040810204080ull) >> 56;
         cand |= g8 << (8 * k); }
      if (base + 64 > ngroups) cand &= ngroups > base ? ((ngroups - base >= 64) ? ~0ull : ((1ull << (ngroups - base)) - 1ull)) : 0ull;      // (groups behind the tape's end are not quiet)
      u64 bits = 0;
      for (u64 cm = cand; cm; cm &= cm - 1) {                          // the rest of such a group, the whole wave on it
         const int idx = __ffsll((long long)cm) - 1;
         const int4 *src = reinterpret_cast<const int4 *>(rows + (base + idx) * 64 * ntrks);
         bool noisy = false;
         for (int v = 8 + lane; v < vpg; v += 64) noisy = noisy || noisy4(src[v]);
         if (__ballot(noisy) == 0) bits |= 1ull << idx; }
      if (lane == 0) qwords[wv] = bits; } }

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

// the samples a candidate's derivation reads: a tile in LDS (k_sift) ...
struct PkTile {
   lds_cp xs;                   // LDS: the tape's rows as they are, row_bytes apart
   int row_bytes;
   int hl;                      // rows in front of the tile
   int sg;                      // -invert: -1 (the detector sees 0 - x, src/readtape.c:1421)
   __device__ __forceinline__ int at(int r, int head) const {      // r relative to the tile's first row (>= -hl); after -invert
      const int m = sg >> 1;                                        // 0 / -1: (x ^ m) - m = x / -x without a multiplication
      return (lds_i16(xs + (r + hl) * row_bytes + head * 2) ^ m) - m; } };
// ... or the tape in HBM (k_sift_hard: the few candidates that need the general walk); rows outside the tape read as zeros
struct PkTape {
   const int16_t *rows; long long t0, nrows; int ntrks, sg;
   __device__ __forceinline__ int at(int r, int head) const {
      const long long n = t0 + r;
      return (n >= 0 && n < nrows) ? sg * (int)rows[n * ntrks + head] : 0; } };

// ... or ONE head's samples around a candidate, copied to LDS by the candidate's wave (k_sift_hard), the tape behind them
struct PkCol {
   const int16_t *col; int r0, n, head;       // col[i] = the sample of `head` at row r0 + i (relative to the tile's first row), after -invert
   PkTape tape;
   __device__ __forceinline__ int at(int r, int hd) const {
      const int i = r - r0;
      return (hd == head && (unsigned)i < (unsigned)n) ? (int)col[i] : tape.at(r, hd); } };

// ... the same without the tape behind it: every row the general walk can ask for is in the copy (k_sift_hard, W <= 50: kPkBack + 3 W + 16 <= kHardCol samples from
// kPkBack + 2 W + 8 rows in front of the candidate) - no test, no branch to a global load between two LDS reads
struct PkColFast {
   const int16_t *col; int r0;
   __device__ __forceinline__ int at(int r, int) const { return (int)col[r - r0]; } };

// LDS carve of k_sift.  ONE definition for the kernel and for the host's sizing.
struct SfLds { unsigned xs, wl, stage, total; };
__host__ __device__ inline SfLds sf_lds_layout(int ntrks, int hl, int hr, int wave_cap, int hcap) {
   SfLds L;
   const int npairs = (ntrks + 1) / 2;
   unsigned o = 0;
   L.xs = o;    o += (unsigned)(hl + kSfTile + hr) * (unsigned)(ntrks * 2) + 32;  o = (o + 15) & ~15u;
   L.wl = o;    o += (unsigned)npairs * wave_cap * 2;  o = (o + 15) & ~15u;      // [pair][wave_cap] candidates of a wave, ordered by (head, row)
   L.stage = o; o += (unsigned)ntrks * hcap;                                      // [head][hcap] the tile's lists as they go to HBM
   L.total = (o + 15) & ~15u;
   return L; }

// k_sift_s on an odd track count (nine- and seven-track tapes): a wave per PAIR of heads left the last head a wave of its own whose packed
// arithmetic ran half empty and whose round of owner derivations had a third of its lanes at work.  The last head's tile is SPLIT into one
// part per pair wave instead (NT / 2 waves, each: its pair over the whole tile + 1 / (NT / 2) of the last head's rows); the part's candidates
// join the wave's list and fill lanes of its rounds that idled anyway.
#ifndef RTFE_SFS_SPLIT
#define RTFE_SFS_SPLIT 1
#endif
__host__ __device__ constexpr bool sfs_split(int nt) { return RTFE_SFS_SPLIT && (nt & 1) && nt >= 5; }
__host__ __device__ constexpr int sfs_waves(int nt) { return sfs_split(nt) ? nt / 2 : (nt + 1) / 2; }
__host__ __device__ constexpr int sfs_part_strip(int nt) { return ((kSfTile + sfs_waves(nt) - 1) / sfs_waves(nt) + 63) / 64; }                 // rows of the last head a lane screens
__host__ __device__ constexpr int sfs_part_rows(int nt) { return ((kSfTile + sfs_waves(nt) - 1) / sfs_waves(nt) + sfs_part_strip(nt) - 1) / sfs_part_strip(nt) * sfs_part_strip(nt); }      // ... a wave
__host__ __device__ constexpr int sfs_hl(int w) { return (w + 7) & ~7; }            // rows in front of the tile k_sift_s reads: the window (the general walk's look-back is k_sift_hard's business)
__host__ __device__ constexpr int sfs_hr(int w) { return (w + 2 + 7) & ~7; }
__host__ __device__ inline int sfs_part_cap(int nt, int hcap) { return (hcap / sfs_waves(nt) + 64 + 15) & ~15; }      // bytes of a wave's part of the last head's staging slot
// ONE definition for the kernel and for the host's sizing: [pair waves][wave_cap] candidates; the pairs' slots, then - split - the parts of the last head's
struct SfsLds { unsigned xs, wl, stage, part, total; };
__host__ __device__ inline SfsLds sfs_lds_layout(int ntrks, int w, int wave_cap, int hcap) {
   SfsLds L;
   const int nw = sfs_waves(ntrks);
   unsigned o = 0;
   L.xs = o;    o += (unsigned)(sfs_hl(w) + kSfTile + sfs_hr(w)) * (unsigned)(ntrks * 2) + 32;  o = (o + 15) & ~15u;
   L.wl = o;    o += (unsigned)nw * (wave_cap + 8) * 2;  o = (o + 15) & ~15u;      // (+ 8: entry wave_cap of a wave's list takes the stores of lanes that have nothing to list)
   L.stage = o;      // (round 6: a pair's records leave straight from the round that makes them - no staging slots: 10 KB less, seven workgroups a CU)
   L.part = o;  if (sfs_split(ntrks)) o += (unsigned)nw * sfs_part_cap(ntrks, hcap);
   L.total = (o + 15) & ~15u;
   return L; }

template <class TileT> struct PkCtxT {
   TileT t;
   int W, lo_i, hi_i;            // window, screen threshold (margin > lo_i), sure threshold (margin >= hi_i)
   int last;                     // last row that exists, relative to the tile's first row (the tape's end; else far away)
};
typedef PkCtxT<PkTile> PkCtx;

// What one candidate of the general walk turns into: up to four records, each with its margin entries.  n > 4: more than fit.
struct PkSink { uint32_t w0[4], w1[4]; int n; };
__device__ __forceinline__ void sink_add(PkSink &s, uint32_t w0, uint32_t w1) {
   #pragma unroll
   for (int i = 0; i < 4; ++i) if (s.n == i) { s.w0[i] = w0; s.w1[i] = w1; }
   ++s.n; }

// margin of owner value `val` at row n: tops val - max(edges), bottoms min(edges) - val
template <class C> __device__ __forceinline__ int pk_margin(const C &c, int head, int n, int val, bool top) {
   const int xl = c.t.at(n - c.W + 1, head), xr = c.t.at(n, head);
   return top ? val - max(xl, xr) : min(xl, xr) - val; }

// Every record carries the margins of its first kPkMar rows from f on (uint16 entries, clamped at 0: entry j - row f + j - lives at
// end[-(j + 1)] of the record's margin block; the blocks grow from a slot's back, record r's block ends 8 r bytes in front of the slot's
// end).  A clean NRZI peak has two lead rows (99.5 % of C2's records: nlead <= 2); what a walker asks beyond the block - lead rows
// from the fifth on, tail rows - it makes from the samples (run_margin, rtfe_gain.hip).  Round 4 stored one entry per lead and tail row,
// variable per record: a second prefix sum, a predicated store per window row and a 4-byte reference per record downstream.
constexpr int kPkMar = 4;
__device__ __forceinline__ uint32_t pk_w0(int pos, bool top, int f, int nlead, int nsure, int ntail) {
   return (uint32_t)(pos + kSfPosBias) | ((top ? 0u : 1u) << 11) | ((uint32_t)(f - pos) << 12) | ((uint32_t)nlead << 18) | ((uint32_t)nsure << 22) | ((uint32_t)ntail << 28); }
__device__ __forceinline__ uint32_t pk_w1(int val, int prev, int nxt, bool top) {
   int dp = top ? val - prev : prev - val, dn = top ? val - nxt : nxt - val;
   dp = dp < -1 ? -1 : (dp > 253 ? 253 : dp); dn = dn < -1 ? -1 : (dn > 253 ? 253 : dn);      // (253: a bottom at -32768 between far neighbours must not read as 0xffff8000, "minimum unknown")
   return (uint32_t)(uint16_t)val | ((uint32_t)(dp + 1) << 16) | ((uint32_t)(dn + 1) << 24); }

// One record for owner `pos` over rows [ra, rb] (all of them rows at which the owner is what the detector tests): from the
// first row above the screen, explicit margins up to the first row at the sure level, the sure stretch, explicit margins
// for what is left up to the last row above the screen; if either explicit part exceeds 15 rows, every row is explicit.
template <class C> __device__ __forceinline__ void pk_emit(const C &c, PkSink &o, int head, int pos, int val, bool top, int ra, int rb, bool unknown) {
   if (rb > c.last) rb = c.last;
   if (ra > rb) return;
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

// the margin block of a record (rows f .. f + kPkMar - 1), from the samples where the record goes: entry j lives at e16[-(j + 1)]
template <class C, class P16> __device__ __forceinline__ void pk_margins(const C &c, int head, uint32_t w0, uint32_t w1, P16 e16) {
   if ((w1 & 0xfffffffeu) == 0xffff8000u) { for (int j = 0; j < kPkMar; ++j) e16[-(j + 1)] = 0; return; }
   const int pos = (int)(w0 & 0x7ffu) - kSfPosBias, f = pos + (int)((w0 >> 12) & 63u);
   const bool top = !((w0 >> 11) & 1u);
   const int val = (int)(int16_t)(w1 & 0xffffu);
   for (int j = 0; j < kPkMar; ++j) { const int m = pk_margin(c, head, f + j, val, top); e16[-(j + 1)] = (uint16_t)(m < 0 ? 0 : m); } }

// ---- what k_prep adds to a record on its way into a stream (CRec::w0 bits 0-10: the tile-relative row of PeakRec::w0 is replaced by the absolute CRec::pos) ----
enum { kCrBad = 1,                        // the tile's list is not there (capacity)
       kCrClear = 2,                      // nothing behind it in the stream has a row at or before the last row it can fire at: if it fires it fires first (k_prep)
       kCrWeak = 4 };                     // a record without a sure stretch, at most kPkMar rows: every row's margin is in its block, the largest of them M quantised in
                                          // bits 3-10: mq = min(ceil(M / 16), 255), so 16 (mq - 1) < M, and M <= 16 mq unless mq = 255.  While a chain's rise threshold is above
                                          // 16 mq for sure (rise_lo) no row of the record can pass the rise test (src/decoder.c:790-791 / 800-801) - it is passed over like a
                                          // record below the amplitude test (what noise wiggles between a screen and the thresholds turn into); while the threshold is at most
                                          // 16 (mq - 1) for sure (rise_hi) its largest row passes - it FIRES, at the first of its rows that passes (k_emit: the margins, exactly)
__device__ __forceinline__ bool crec_weak_dead(uint32_t w0, int rise_lo) { return (w0 & kCrWeak) && ((w0 >> 3) & 255u) != 255u && (int)(16u * ((w0 >> 3) & 255u)) <= rise_lo; }
// the margin a record's rows reach for sure: the sure level of its screen, or - kCrWeak - what its largest row is known to exceed
__device__ __forceinline__ int crec_level(uint32_t w0, int sure_i) { return (w0 & kCrWeak) ? 16 * (int)((w0 >> 3) & 255u) - 16 : sure_i; }
// kCrWeak and its bits from a record as k_sift left it and its margin block (entry j - row f + j - at the block's end - 2 (j + 1))
__device__ __forceinline__ uint32_t crec_weak_bits(uint32_t w0, uint32_t w1, uint2 mar) {
   const uint32_t ns = (w0 >> 22) & 63u, nl = (w0 >> 18) & 15u;          // (no sure stretch: nl rows, no tail; "every row explicit", ns = 63, means sixteen rows and more)
   const uint32_t my = nl >= 2u ? mar.y : (mar.y & 0xffff0000u), mx = nl >= 4u ? mar.x : (nl == 3u ? (mar.x & 0xffff0000u) : 0u);
   const uint32_t a = max(my >> 16, my & 0xffffu), b = max(mx >> 16, mx & 0xffffu);
   uint32_t mq = (max(a, b) + 15u) >> 4;
   if (mq > 255u) mq = 255u;
   return (ns == 0u && w1 != 0xffff8000u && nl - 1u < (uint32_t)kPkMar) ? (kCrWeak | (mq << 3)) : 0u; }
// a record that can fire on the lean step at all: a sure stretch (1..62 rows) with its minimum known, or kCrWeak
__device__ __forceinline__ bool crec_can_clear(uint32_t w0, uint32_t w1, uint32_t weak) { return w1 != 0xffff8000u && ((unsigned)((int)((w0 >> 22) & 63u) - 1) < 62u || weak != 0u); }

// "a rescan is forced at row r whatever happened before": the sample that leaves the window is
//   (a) the maximum of the old window AND not exceeded by the sample that enters (src/decoder.c:763-767: the new sample is
//       folded into pkww_maxv before old_left is compared with it; the maximum is always exact), or
//   (b) its minimum: the reference's own minimum is the value of a sample still inside the window, hence >= the true
//       minimum, and it was the minimum of an earlier window that already held the leaving sample (a sample that entered
//       later would have to be to the right of it), hence <= it: equal, and old_left == pkww_minv fires.
// x[r-W] >= all of x[r-W+1 .. r], or <= all of x[r-W+1 .. r-1].
template <class C> __device__ __forceinline__ bool pk_async_g(const C &c, int head, int r) {
   const int s = r - c.W;
   const int v = c.t.at(s, head);
   bool dom = true, sub = true;
   for (int i = 1; i < c.W; ++i) {
      const int y = c.t.at(s + i, head);
      dom = dom && y <= v; sub = sub && y >= v; }
   return sub || (dom && c.t.at(r, head) <= v); }
// leftmost minimum of the window that ends at row r (the rescan of src/decoder.c:768-775)
template <class C> __device__ __forceinline__ int pk_argmin_g(const C &c, int head, int r) {
   int best = r - c.W + 1, bv = c.t.at(best, head);
   for (int j = best + 1; j <= r; ++j) { const int v = c.t.at(j, head); if (v < bv) { bv = v; best = j; } }
   return best; }
template <class C> __device__ __forceinline__ bool pk_async(const C &c, int head, int r) { return pk_async_g(c, head, r); }
template <class C> __device__ __forceinline__ int pk_argmin(const C &c, int head, int r) { return pk_argmin_g(c, head, r); }
// a bottom candidate, general walk: sample q is the true window minimum (first from the left) at rows [ra, rb]; what the
// reference tests there is its own minimum, refreshed only by rescans (src/decoder.c:765-775).
template <class C> __device__ __forceinline__ void pk_bot(const C &c, PkSink &out, int head, int q) {
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

// ---- the general walk with the SIXTEEN LANES of a candidate's group at work (k_sift_hard; round 6).  pk_bot / pk_emit above are one lane's loops over rows - up to W - 2 margin
// tests a pass, several passes a record, 64 table look-ups back to the last forced rescan: a few hundred dependent LDS reads a candidate.  Here a lane takes a ROW: a pass over a
// run's rows is one step (three for the widest windows) whose verdicts a ballot gathers into a mask, and the counts are bit scans of the masks (as pk_fast_w does for the common
// candidate).  Control flow is the WAVE's (its four groups go through the same loops; a group that has nothing to do in a step votes with zeros), the results are each group's own,
// the same in all its lanes.  gsh: the group's first lane in its wave.
template <class C> __device__ __forceinline__ uint64_t pk_grp_ballot(bool p, int gsh) { return (uint64_t)((__ballot(p) >> gsh) & 0xffffull); }
template <class C> __device__ __forceinline__ void pk_emit_grp(const C &c, PkSink &o, int head, bool ev, int pos, int val, int ra, int rb, bool unknown, int sl, int gsh) {
   if (rb > c.last) rb = c.last;
   ev = ev && ra <= rb;
   uint64_t lo = 0, hi = 0;                                          // row ra + k: its margin above the screen / at the sure level
   for (int k0 = 0; __ballot(ev && ra + k0 <= rb) != 0ull; k0 += 16) {
      const int n = ra + k0 + sl;
      const bool in = ev && n <= rb;
      const int m = in ? pk_margin(c, head, n, val, false) : 0;
      lo |= pk_grp_ballot<C>(in && m > c.lo_i, gsh) << k0;
      hi |= pk_grp_ballot<C>(in && m >= c.hi_i, gsh) << k0; }
   if (!ev || !lo) return;
   const int f = pk_ctz(lo), l = 63 - pk_clz(lo), span = l - f + 1;
   int nlead = 0, nsure = 0, ntail = 0;
   if (unknown) nsure = span;                                        // (runs are shorter than 63 rows)
   else {
      const uint64_t h = hi >> f;
      nlead = pk_ctz(h);
      if (nlead > span) nlead = span;
      nsure = nlead >= span ? 0 : pk_ctz((uint64_t)~(h >> nlead));
      if (nsure > span - nlead) nsure = span - nlead;
      ntail = span - nlead - nsure;
      if (nlead > 15 || ntail > 15 || nsure > 62) { nlead = span >> 4; ntail = span & 15; nsure = 63; } }
   sink_add(o, pk_w0(pos, false, ra + f, nlead, nsure, ntail), unknown ? 0xffff8000u : pk_w1(val, c.t.at(pos - 1, head), c.t.at(pos + 1, head), false)); }

template <class C> __device__ __forceinline__ void pk_bot_grp(const C &c, PkSink &out, int head, int q, bool live, int sl, int gsh) {
   const int W = c.W;
   const int val = live ? c.t.at(q, head) : 0;
   uint64_t mj = 0, md = 0;                                          // x[q - 1 - j] > val / x[q + 1 + j] >= val
   for (int j0 = 0; __ballot(live && j0 < W - 1) != 0ull; j0 += 16) {
      const int j = j0 + sl;
      const bool in = live && j < W - 1;
      mj |= pk_grp_ballot<C>(in && c.t.at(q - j - 1, head) > val, gsh) << j0;
      md |= pk_grp_ballot<C>(in && j < W - 2 && c.t.at(q + j + 1, head) >= val, gsh) << j0; }
   const int J = pk_ctz((uint64_t)~mj), D = pk_ctz((uint64_t)~md);    // (bits from W - 1 / W - 2 on are clear: the counts stop there)
   const int aq = q + (W - 1 - J > 0 ? W - 1 - J : 0);                // first row at which q is the first window minimum
   const int ra = aq > q + 1 ? aq : q + 1, rb = q + D;
   bool act = live && ra <= rb;
   uint64_t mlo = 0;                                                // rows ra + k at which the true minimum would pass the screen
   for (int k0 = 0; __ballot(act && ra + k0 <= rb) != 0ull; k0 += 16) {
      const int n = ra + k0 + sl;
      const bool in = act && n <= rb;
      mlo |= pk_grp_ballot<C>(in && pk_margin(c, head, n, val, false) > c.lo_i, gsh) << k0; }
   act = act && mlo != 0;
   const int n0 = ra + pk_ctz(mlo);
   uint64_t ma = 0;                                                 // a rescan at any row of [aq, n0] makes q the reference's minimum from then on
   for (int k0 = 0; __ballot(act && aq + k0 <= n0) != 0ull; k0 += 16) {
      const int r = aq + k0 + sl;
      const bool in = act && r <= n0;
      ma |= pk_grp_ballot<C>(in && pk_async(c, head, r), gsh) << k0; }
   // 0: the minimum the reference holds at n0 comes from further back; 1: one last record for q; 2: on the chain of rescans; 3: done
   int mode = !act ? 3 : (ma ? 1 : 0);
   bool fin_unknown = false;
   int r0 = aq - 1;
   {  bool found = false;                                            // last forced rescan in front of aq, kPkBack rows back at the most
      for (int b0 = 0; __ballot(mode == 0 && !found && b0 < kPkBack) != 0ull; b0 += 16) {
         const int r = aq - 1 - b0 - sl;
         const bool in = mode == 0 && !found && b0 + sl < kPkBack;
         const uint64_t m = pk_grp_ballot<C>(in && pk_async(c, head, r), gsh);
         if (mode == 0 && !found && m) { r0 = aq - 1 - b0 - pk_ctz(m); found = true; } }
      if (mode == 0 && !found) { mode = 1; fin_unknown = true; } }
   uint64_t m1 = 0;                                                 // first forced rescan behind n0 (within the run)
   for (int k0 = 0; __ballot(mode == 0 && n0 + 1 + k0 <= rb) != 0ull; k0 += 16) {
      const int r = n0 + 1 + k0 + sl;
      const bool in = mode == 0 && r <= rb;
      m1 |= pk_grp_ballot<C>(in && pk_async(c, head, r), gsh) << k0; }
   const int r1 = m1 ? n0 + 1 + pk_ctz(m1) : rb + 1;
   int start = r0, o = 0, hop = 0;
   if (mode == 0) { o = pk_argmin(c, head, r0); mode = 2; }
   while (__ballot(mode == 1 || mode == 2) != 0ull) {
      bool ev = false, eunk = false;
      int epos = q, eval = val, era = 0, erb = rb;
      if (mode == 1) { ev = true; eunk = fin_unknown; era = n0; mode = 3; }
      else if (mode == 2) {
         if (hop >= kPkBack + 64) { ev = true; eunk = true; era = n0 > start ? n0 : start; mode = 3; }      // (on a rising slope the minimum is the sample about to leave: a rescan per row)
         else if (o == q) { ev = true; era = n0 > start ? n0 : start; mode = 3; }
         else {
            const int next = o + W < r1 ? o + W : r1;                // the epoch of owner o covers rows [start, next - 1]
            if (next - 1 >= n0) { ev = true; epos = o; eval = c.t.at(o, head); era = n0 > start ? n0 : start; erb = rb < next - 1 ? rb : next - 1; }
            if (next > rb) mode = 3;
            else { start = next; o = pk_argmin(c, head, next); } }
         ++hop; }
      pk_emit_grp(c, out, head, ev, epos, eval, era, erb, eunk, sl, gsh); } }

// ---- the common case in registers: every sample a candidate's rows can see is loaded with independent LDS reads, and the run
// and its record follow from bit masks over the rows.  WM >= W.
// Returns 0: no record; 1: the record (w0, w1); 2: the candidate needs the general walk (a bottom whose first rows precede every
// forced rescan) - k_sift_hard.
template <int WM>
__device__ __forceinline__ int pk_fast(const PkCtx &c, int head, int p, bool bot, uint32_t &w0, uint32_t &w1) {
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
   if (ra > D) return 0;
   mask_t V = (((one << D) << 1) - 1) & ~((one << ra) - 1);
   mask_t C = V & lom;
   // rows behind the tape's end do not exist
   const int klast = c.last - p;
   if (klast < MB - 1) C &= klast < 0 ? (mask_t)0 : (mask_t)(((one << klast) << 1) - 1);
   if (!C) return 0;
   if (bot) {
      // the reference's minimum is this sample from the first forced rescan at or behind aq = q + max(0, W-1-J) on: the common case is
      // a rescan at the very first candidate row
      const int n0 = pk_ctz(C), aq = W - 1 - J > 0 ? W - 1 - J : 0;
      bool ok = pk_async(c, head, p + n0);
      for (int r = n0 - 1; r >= aq && !ok; --r) ok = pk_async(c, head, p + r);      // (any rescan in [aq, n0] will do: pk_bot's first case)
      if (!ok) return 2;
      C &= ~((one << n0) - 1); }
   const int val = bot ? -v2 : v2;                                      // (the sample as the detector sees it)
   w1 = pk_w1(val, c.t.sg * lds_i16(pr - rb_), c.t.sg * lds_i16(pr + rb_), !bot);
   const int f = pk_ctz(C), l = MB - 1 - pk_clz(C);
   const int span = l - f + 1;
   int nlead = pk_ctz((mask_t)(him >> f));
   if (nlead > span) nlead = span;
   int nsure = nlead >= span ? 0 : pk_ctz((mask_t)~(him >> (f + nlead)));
   if (nsure > span - nlead) nsure = span - nlead;
   int ntail = span - nlead - nsure;
   if (nlead > 15 || ntail > 15 || nsure > 62) { nlead = span >> 4; ntail = span & 15; nsure = 63; }
   w0 = pk_w0(p, !bot, p + f, nlead, nsure, ntail);
   return 1; }

// ---- the same for a window width known at compile time, two samples per operation: row k's left and right window edge share a
// register (v_perm_b32), a bottom is a top of y = ~x (order-reversing, no overflow), and every test is the sign bit of a packed
// saturating difference (v_pk_sub_i16 clamp) shifted into a mask.  Needs sure_i <= 32767 (margins below the sure level are then exact
// in 16 bits).
template <int W>
__device__ __forceinline__ int pk_fast_w(const PkCtx &c, int head, int p, bool bot, uint32_t &w0, uint32_t &w1) {
   static_assert(W >= 4 && W <= 17, "the row masks live in the 16-bit halves of a register");
   const int rb_ = c.t.row_bytes;
   const uint32_t m2 = (bot != (c.t.sg < 0)) ? 0xffffffffu : 0u;
   lds_cp pr = c.t.xs + head * 2 + (p + c.t.hl) * rb_, pl = pr - (W - 1) * rb_;
   // (round 6: rows k and k + H share a register - the left edges of the two in one, the right edges in another - where rounds 3 - 5 paired a row's own two edges:
   //  the margin is then a plain packed minimum of the two differences (no half swap), the four tests gather each its own mask, and a mask's two halves
   //  are the rows 0 .. H - 1 and H .. 2 H - 1 one behind the other: 7 instead of 9 vector instructions a row)
   constexpr int H = W / 2;                                            // = ceil((W - 1) / 2) rows a half
   uint32_t L2[H], R2[H];                                              // x[p + k - W + 1] / x[p + k] for k = j (low half) and k = j + H (high half)   (as y)
   #pragma unroll
   for (int j = 0; j < H; ++j) {
      const int k2 = j + H < W - 1 ? j + H : W - 2;                    // (an even W has one row less than its halves hold: the last is there twice, its bit W - 1 is masked off)
      L2[j] = ((uint32_t)(uint16_t)lds_i16(pl + j * rb_) | ((uint32_t)(uint16_t)lds_i16(pl + k2 * rb_) << 16)) ^ m2;
      R2[j] = ((uint32_t)(uint16_t)lds_i16(pr + j * rb_) | ((uint32_t)(uint16_t)lds_i16(pr + k2 * rb_) << 16)) ^ m2; }
   const uint32_t vv = (R2[0] & 0xffffu) | (R2[0] << 16);              // the extreme in both halves
   const uint32_t ones = 0x00010001u, thrlo = pk_dup(c.lo_i + 1), thrhi = pk_dup(c.hi_i);
   uint32_t aL = 0, aR = 0, aLo = 0, aHi = 0, dl_prev = 0, dr_next = 0;
   #pragma unroll
   for (int j = 0; j < H; ++j) {
      const uint32_t dl = pk_subs(vv, L2[j]), dr = pk_subs(vv, R2[j]);  // extreme - left edge, extreme - right edge
      aL = (aL >> 1) | (pk_subs(dl, ones) & kPkSigns);                 // sign: left edge not strictly below the extreme
      aR = (aR >> 1) | (dr & kPkSigns);                                // sign: right edge above it
      const uint32_t mn = pk_min(dl, dr);                              // the rows' margins
      aLo = (aLo >> 1) | (pk_subs(mn, thrlo) & kPkSigns);              // sign: margin not above the screen
      aHi = (aHi >> 1) | (pk_subs(mn, thrhi) & kPkSigns);              // sign: margin below the sure level
      if (j == W - 2 - H) dl_prev = dl;                                // (row W - 2's left edge is x[p - 1], row 1's right edge x[p + 1]: the extreme's neighbours)
      if (j == 1) dr_next = dr; }
   constexpr uint32_t ALL = (1u << (W - 1)) - 1u;
   auto gather = [](const uint32_t a2) -> uint32_t { return ((a2 & 0xffffu) >> (16 - H)) | ((a2 >> (32 - H)) << H); };      // bit j of a half sits at 16 - H + j
   const uint32_t lm = ~gather(aL) & ALL, rm = ~gather(aR) & ALL & ~1u;
   const uint32_t lom = ~gather(aLo) & ALL, him = ~gather(aHi) & ALL;
   // J: consecutive left samples below the extreme, from distance 1 (k = W-2) outwards (tops: up to W-2, bottoms: W-1, i.e. k = 0 too)
   const uint32_t lsh = lm << (31 - (W - 2));                          // top bit = k = W-2
   int J = pk_clz((uint32_t)~lsh);
   const int jmax = bot ? W - 1 : W - 2;
   if (J > jmax) J = jmax;
   int D = pk_ctz((uint32_t)~(rm >> 1));                               // consecutive right samples not above it, from k = 1
   if (D > W - 2) D = W - 2;
   const int ra = W - 1 - J > 1 ? W - 1 - J : 1;
   if (ra > D) return 0;
   const uint32_t V = ((2u << D) - 1u) & ~((1u << ra) - 1u);
   uint32_t C = V & lom;
   const int klast = c.last - p;                                       // rows behind the tape's end do not exist
   if (klast < 31) C &= klast < 0 ? 0u : ((2u << klast) - 1u);
   if (!C) return 0;
   if (bot) {
      // the reference's minimum is this sample from the first forced rescan at or behind aq = p + max(0, W-1-J) on: a rescan at any row of
      // [aq, n0] will do (pk_bot's first case) - the first candidate row itself nearly always, else the few rows in front of it
      const int n0 = pk_ctz(C), aq = W - 1 - J > 0 ? W - 1 - J : 0;
      bool ok = pk_async(c, head, p + n0);
      for (int r = n0 - 1; r >= aq && !ok; --r) ok = pk_async(c, head, p + r);
      if (!ok) return 2;
      C &= ~((1u << n0) - 1u); }
   const int raw = lds_i16(pr);
   const int val = c.t.sg < 0 ? -raw : raw;                            // the sample as the detector sees it
   int dp, dn;                                                         // the extreme's distance to its two neighbours
   dp = (int)(int16_t)(dl_prev >> 16); dn = (int)(int16_t)(dr_next & 0xffffu);
   dp = dp < -1 ? -1 : (dp > 253 ? 253 : dp); dn = dn < -1 ? -1 : (dn > 253 ? 253 : dn);      // (253: a bottom at -32768 between far neighbours must not read as 0xffff8000, "minimum unknown")
   w1 = (uint32_t)(uint16_t)val | ((uint32_t)(dp + 1) << 16) | ((uint32_t)(dn + 1) << 24);
   const int f = pk_ctz(C), l = 31 - pk_clz(C);
   const int span = l - f + 1;
   int nlead = pk_ctz((uint32_t)(him >> f));
   if (nlead > span) nlead = span;
   int nsure = nlead >= span ? 0 : pk_ctz((uint32_t)~(him >> (f + nlead)));
   if (nsure > span - nlead) nsure = span - nlead;
   int ntail = span - nlead - nsure;
   if (nlead > 15 || ntail > 15 || nsure > 62) { nlead = span >> 4; ntail = span & 15; nsure = 63; }
   w0 = pk_w0(p, !bot, p + f, nlead, nsure, ntail);
   return 1; }

// the margin block of a record built by pk_fast_w: rows f .. f + kPkMar - 1, two aligned 16-bit LDS reads a row (y = x or ~x as in
// pk_fast_w: differences of y are the detector's margins).  Returns the block as it lies in memory (8 bytes at block end - 8).
template <int W>
__device__ __forceinline__ uint2 pk_margins_w(const PkCtx &c, int head, int p, bool bot, uint32_t w0) {
   const int rb_ = c.t.row_bytes;
   const int m2 = (bot != (c.t.sg < 0)) ? -1 : 0;
   const int fo = (int)((w0 >> 12) & 63u);
   lds_cp pv = c.t.xs + head * 2 + (p + c.t.hl) * rb_, pr = pv + fo * rb_, pl = pr - (W - 1) * rb_;
   const int vy = lds_i16(pv) ^ m2;
   uint32_t e[kPkMar];
   #pragma unroll
   for (int j = 0; j < kPkMar; ++j) {
      const int ly = lds_i16(pl + j * rb_) ^ m2, ry = lds_i16(pr + j * rb_) ^ m2;
      const int m = vy - max(ly, ry);
      e[j] = (uint32_t)(m < 0 ? 0 : m); }
   return make_uint2(e[3] | (e[2] << 16), e[1] | (e[0] << 16)); }

// ---- wave-wide inclusive prefix sum: DPP row shifts and broadcasts (seven dependent VALU operations; __shfl_up would be six
// ds_bpermute round trips through the LDS pipeline) ----
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#ifdef RTFE_CPU_EMUL
   for (int s2 = 1; s2 < 64; s2 <<= 1) { const int y = __shfl_up(v, s2); if (lane >= s2) v += y; }
   return v;
#else
   (void)lane;
   v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
   v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
   v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
   v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
   v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
   v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
   return v;
#endif
}
__device__ __forceinline__ int wave_last(int v) {
#ifdef RTFE_CPU_EMUL
   return __shfl(v, 63);
#else
   return __builtin_amdgcn_readlane(v, 63);
#endif
}

// a pointer that is the same in every lane, kept in scalar registers as it is: an address made of it and a lane's 32-bit offset is the
// global_load ... v_off, s[base] form - the compiler would otherwise fold a constant part of the base into a 64-bit vector addition per access
// (the pointer keeps its address space - global, 1 - through the empty asm: a generic one would make every access a FLAT instruction)
#ifdef RTFE_CPU_EMUL
typedef const char *glb_cp;
typedef char *glb_p;
__device__ __forceinline__ glb_cp sgpr_ptr(glb_cp p) { return p; }
__device__ __forceinline__ glb_p sgpr_ptr(glb_p p) { return p; }
__device__ __forceinline__ int4 glb_ld16(glb_cp p) { int4 v; memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void glb_st16(glb_p p, const int4 v) { memcpy(p, &v, 16); }
__device__ __forceinline__ void glb_st4(glb_p p, const uint32_t v) { memcpy(p, &v, 4); }
#else
typedef const __attribute__((address_space(1))) char *glb_cp;
typedef __attribute__((address_space(1))) char *glb_p;
__device__ __forceinline__ glb_cp sgpr_ptr(glb_cp p) { asm volatile("" : "+s"(p)); return p; }
__device__ __forceinline__ glb_p sgpr_ptr(glb_p p) { asm volatile("" : "+s"(p)); return p; }
typedef int glb_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 glb_ld16(glb_cp p) { const glb_v4 v = *reinterpret_cast<const __attribute__((address_space(1))) glb_v4 *>(p); return make_int4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void glb_st4(glb_p p, const uint32_t v) { *reinterpret_cast<__attribute__((address_space(1))) uint32_t *>(p) = v; }
__device__ __forceinline__ void glb_st16(glb_p p, const int4 v) { glb_v4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; *reinterpret_cast<__attribute__((address_space(1))) glb_v4 *>(p) = w; }
#endif

#ifdef RTFE_CPU_EMUL
typedef uint16_t *lds_u16p;
typedef uint32_t *lds_u32p;
#else
typedef __attribute__((address_space(3))) uint16_t *lds_u16p;
typedef __attribute__((address_space(3))) uint32_t *lds_u32p;
#endif

// ------------------------------------------------------------------------------------------------
// k_sift
// ------------------------------------------------------------------------------------------------
// the tape's bytes of a tile (with its halo) into registers: 16-byte loads, consecutive lanes consecutive vectors.  Only for tiles
// whose rows all exist (e_first >= 0, the last vector inside the tape); the tape's two ends go through sf_fill_edge.
template <int kSfVec>
__device__ __forceinline__ void sf_fetch(int4 (&q)[kSfVec], const int16_t *__restrict__ rows, long long e_first, int nvec, int tid, int nthreads) {
   const int4 *src = reinterpret_cast<const int4 *>(rows + e_first);
   #pragma unroll
   for (int k = 0; k < kSfVec; ++k) {
      const int vi = k * nthreads + tid;
      if (vi < nvec) q[k] = src[vi]; } }
// a tile at one of the tape's ends: sample by sample into LDS, zeros for the rows that do not exist (twice per scan)
__device__ __attribute__((noinline)) void sf_fill_edge(lds_p xs, const int16_t *__restrict__ rows, long long e_first, long long total_elem, int nelem, int tid, int nthreads) {
#ifdef RTFE_CPU_EMUL
   int16_t *d = reinterpret_cast<int16_t *>(xs);
#else
   __attribute__((address_space(3))) int16_t *d = reinterpret_cast<__attribute__((address_space(3))) int16_t *>(xs);
#endif
   for (int i = tid; i < nelem; i += nthreads) { const long long g = e_first + i; d[i] = (g >= 0 && g < total_elem) ? rows[g] : (int16_t)0; } }

// the tile's quiet bits: one 16-bit word per tile (only complete groups can be quiet); k_qpack strings them together
__device__ __forceinline__ void sf_publish_quiet(unsigned int noisy, long long tile, long long nrows, uint16_t *qtile) {
   unsigned int quiet = ~noisy & ((1u << kSfGroups) - 1u);
   const long long left = nrows - tile * kSfTile;
   if (left < kSfTile) quiet &= left < 64 ? 0u : ((1u << (int)(left / 64)) - 1u);
   qtile[tile] = (uint16_t)quiet; }
// the quiet map k_bursts reads: bit c of word c >> 6 = group c of 64 rows is quiet; group c is bit c % 14 of tile c / 14
__global__ void __launch_bounds__(256) k_qpack(const uint16_t *__restrict__ qtile, long long ntiles, u64 *__restrict__ qwords, long long nwords) {
   for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (long long)gridDim.x * blockDim.x) {
      u64 word = 0;
      const long long c0 = w * 64;
      for (long long t = c0 / kSfGroups; t <= (c0 + 63) / kSfGroups && t < ntiles; ++t) {
         const u64 bits = qtile[t];
         const long long rel = t * kSfGroups - c0;
         word |= rel >= 0 ? bits << rel : bits >> -rel; }
      qwords[w] = word; } }

// WM >= the widest window; NV = 16-byte vectors of the next tile a thread holds in registers (>= tile vectors / threads);
// WPS = waves per SIMD the register allocation is held to (workgroups per CU x waves per workgroup / 4)
template <int WM, int MAXT, int NV, int WPS>
__global__ void __launch_bounds__(MAXT, WPS) k_sift(const DevCfg *__restrict__ cfgp, const int16_t *__restrict__ rows, long long nrows, long long ntiles,
                                                    uint16_t *__restrict__ qtile, PeakDir *__restrict__ dir, unsigned char *__restrict__ pool,
                                                    SfHard *__restrict__ hard, int hard_cap, int *__restrict__ hard_count, unsigned long long *__restrict__ dbg) {
#ifdef RTFE_CPU_EMUL
   unsigned char *smem = g_dyn_smem;
#else
   extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
   __shared__ unsigned int s_noisy[2];
   const int ntrks = cfgp->ntrks, nscreens = cfgp->nscreens;
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
   const int HL = cfgp->pk_hl, HR = cfgp->pk_hr;
   const int row_bytes = ntrks * 2;
   const int hcap = cfgp->pk_slot, wave_cap = cfgp->pk_wave_cap;
   const int cut = cfgp->cut;
   const bool inv = cfgp->invert != 0;
   const SfLds L = sf_lds_layout(ntrks, HL, HR, wave_cap, hcap);
   unsigned char *xs = smem + L.xs;
   const lds_p xsl = to_lds(xs);
   PkCtx cx;
   cx.t.xs = xsl; cx.t.row_bytes = row_bytes; cx.t.hl = HL; cx.t.sg = inv ? -1 : 1;
   const int nvec = (HL + kSfTile + HR) * ntrks / 8;                    // 16-byte vectors of a tile with its halo (HL, HR: multiples of 8)
   const int vpg = 8 * ntrks;                                           // ... per quiet group of 64 rows
   const int v_own0 = HL * ntrks / 8;
   const int quiet_i = cfgp->quiet_i;
   // Tile order: round `it` of the grid covers tiles [it G, (it + 1) G) - at any moment the chip works on ONE compact window of the
   // tape (G tiles = 16 MB), which is what the address translation and the HBM channels like (a contiguous run of tiles per
   // workgroup - 1024 streams 1.8 MB apart - streams at 1.4 TB/s, this at several times that).  Inside a round the workgroups of
   // an XCD (block b runs on XCD b % 8) take a contiguous eighth, so the halo rows neighbours share are in that XCD's L2.
   const long long G = gridDim.x;
   const long long tile_first = (G & 7) ? (long long)blockIdx.x : (long long)(blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
   const long long tile_lo = tile_first, tile_hi = ntiles;
   // a tile is "inside" when every row of it and of its halo exists (its bytes then come as 16-byte vectors, one tile ahead)
   const long long inside_lo = (HL + kSfTile - 1) / kSfTile, inside_hi = (nrows - HR - 15) / kSfTile - 1;      // tiles inside_lo .. inside_hi
   const bool prof = cfgp->debug == 3;
   long long pc_copy = 0, pc_dense = 0, pc_own = 0;                     // cycles per phase (RTFE_DEBUG=3; lane 0 of every wave adds its own up)
   unsigned int pn_hard = 0, pn_rounds = 0, pn_bytes = 0;
   // this wave's pair of heads, its candidate list and its two staging slots
   const int pair = wave, h_lo = 2 * pair, h_hi = 2 * pair + 1;
   const bool has_hi = h_hi < ntrks;
   const lds_u16p wlist = reinterpret_cast<lds_u16p>(to_lds(smem + L.wl)) + wave * (wave_cap + 8);
   const lds_p slot_lo = to_lds(smem + L.stage) + h_lo * hcap, slot_hi = slot_lo + hcap;
   int4 q[NV];
   #pragma unroll
   for (int k = 0; k < NV; ++k) q[k] = make_int4(0, 0, 0, 0);
   if (tile_lo < tile_hi && tile_lo >= inside_lo && tile_lo <= inside_hi) sf_fetch(q, rows, (tile_lo * kSfTile - HL) * ntrks, nvec, tid, nthreads);
   if (tid < 2) s_noisy[tid] = 0;
   int par = 0;
   long long last_tile = -1;
   for (long long tile = tile_lo; tile < tile_hi; tile += G, par ^= 1) {
      last_tile = tile;
      const long long t0 = tile * kSfTile;
      const long long lastl = nrows - 1 - t0;
      cx.last = lastl > 0x3fffffff ? 0x3fffffff : (int)lastl;
      long long tk0 = 0, tk1 = 0;
      if (prof) tk0 = clock64();
      // ---- 1. the prefetched bytes -> LDS (every wave is done with the tile in front: the barrier at the end of the round before);
      // the next tile's loads go out at once and travel while this tile is worked on ----
      if (tile >= inside_lo && tile <= inside_hi) {
         #pragma unroll
         for (int k = 0; k < NV; ++k) { const int vi = k * nthreads + tid; if (vi < nvec) reinterpret_cast<int4 *>(xs)[vi] = q[k]; } }
      else sf_fill_edge(xsl, rows, (t0 - HL) * ntrks, nrows * ntrks, nvec * 8, tid, nthreads);
      __syncthreads();
      if (tile + G < tile_hi && tile + G >= inside_lo && tile + G <= inside_hi) sf_fetch(q, rows, ((tile + G) * kSfTile - HL) * ntrks, nvec, tid, nthreads);
      // the quiet map of the tile in front (its bits were complete at the barrier)
      if (tid == 0 && tile > tile_lo) { sf_publish_quiet(s_noisy[par ^ 1], tile - G, nrows, qtile); s_noisy[par ^ 1] = 0; }
      // ---- 2. quiet groups: 14 x 64 rows, flat 16-byte reads of the tile proper ----
      {
         const uint32_t qpk = pk_dup(quiet_i), q2 = 2u * (uint32_t)quiet_i;
         for (int vb = 0; vb < kSfGroups * vpg; vb += nthreads) {
            const int vi = vb + tid;
            bool noisy = false;
            if (vi < kSfGroups * vpg) {
               const int4 v = reinterpret_cast<const int4 *>(xs)[v_own0 + vi];
               const uint32_t m = pk_maxu(pk_maxu(pk_addu((uint32_t)v.x, qpk), pk_addu((uint32_t)v.y, qpk)),
                                          pk_maxu(pk_addu((uint32_t)v.z, qpk), pk_addu((uint32_t)v.w, qpk)));
               noisy = (m & 0xffffu) > q2 || (m >> 16) > q2; }
            const u64 nb = __ballot(noisy);
            if (lane == 0 && nb) {                                           // the 64 vectors of this ballot lie in a few groups
               const int vfirst = vb + wave * 64;
               unsigned int bits = 0;
               for (int g = vfirst / vpg; g <= (vfirst + 63) / vpg && g < kSfGroups; ++g) {
                  const int a = g * vpg - vfirst, b = a + vpg;
                  const u64 ma = a <= 0 ? ~0ull : (a >= 64 ? 0ull : ~0ull << a), mb = b >= 64 ? ~0ull : (b <= 0 ? 0ull : ~(~0ull << b));
                  if (nb & ma & mb) bits |= 1u << g; }
               if (bits) atomicOr(&s_noisy[par], bits); } } }
      if (prof) { tk1 = clock64(); pc_copy += tk1 - tk0; }
      if (cut != 1)
      for (int sc = 0; sc < nscreens; ++sc) {
         cx.W = cfgp->screen[sc].W; cx.lo_i = cfgp->screen[sc].rise_i; cx.hi_i = cfgp->screen[sc].sure_i;
         const int minpk_i = cfgp->screen[sc].minpk_i;
         if (prof) tk0 = clock64();
         // ---- 3. candidate samples: local extremum + amplitude, one lane per 14-row strip of a pair of heads ----
         uint32_t tm = 0, bm = 0;
         {
            const lds_cp base = xsl + (HL + kSfStrip * lane) * row_bytes + 4 * pair;
            // (-invert: tops and bottoms swap on the raw codes; "x > A" becomes "x < -A")
            const uint32_t at = minpk_i < 0 ? pk_dup(-32768) : pk_dup(minpk_i), ab = minpk_i < 0 ? pk_dup(32767) : pk_dup(-minpk_i);
            uint32_t x[kSfStrip + 2];
            #pragma unroll
            for (int i = 0; i < kSfStrip + 2; ++i) x[i] = lds_u32u(base + (i - 1) * row_bytes);
            uint32_t yc = pk_max(x[1], at), zc = pk_min(x[1], ab);
            uint32_t uy = pk_subs(pk_max(x[0], at), yc), dz = pk_subs(zc, pk_min(x[0], ab));      // sign: rising into this row above the floor / falling into it below the ceiling
            #pragma unroll
            for (int i = 0; i < kSfStrip; ++i) {
               const uint32_t yn = pk_max(x[i + 2], at), zn = pk_min(x[i + 2], ab);
               const uint32_t uyn = pk_subs(yc, yn), dzn = pk_subs(zn, zc);
               tm = (tm >> 1) | (uy & ~uyn & kPkSigns);
               bm = (bm >> 1) | (dz & ~dzn & kPkSigns);
               yc = yn; zc = zn; uy = uyn; dz = dzn; }
            tm = (tm >> (16 - kSfStrip)) & 0x3fff3fffu; bm = (bm >> (16 - kSfStrip)) & 0x3fff3fffu;
            if (inv) { const uint32_t s2 = tm; tm = bm; bm = s2; }
            if (!has_hi) { tm &= 0xffffu; bm &= 0xffffu; }                  // odd track count: the last pair's upper half is the next row
            // rows that do not exist cannot own a run
            const long long r0 = t0 + kSfStrip * lane;
            if (r0 + kSfStrip > nrows) { const int keep = (int)(nrows - r0 > 0 ? nrows - r0 : 0); const uint32_t mk = (1u << keep) - 1u; tm &= mk | (mk << 16); bm &= mk | (mk << 16); } }
         if (prof) { tk1 = clock64(); pc_dense += tk1 - tk0; }
         // ---- 4. every wave on its own (one pair of heads).  The candidates of its 64 strips are compacted (a prefix sum of the per-lane
         // counts) into a list ordered by (head, row); in rounds of 64 lane i evaluates candidate i, prefix sums number the records within
         // their lists, and records and margin entries go to the lists' staging slots in LDS.  More than pk_wave_cap candidates (noise
         // above the screen), or a list that outgrows its slot: the list is marked unavailable and the bursts that need it take the sample path.
         // A candidate that needs the general walk leaves a placeholder that k_sift_hard resolves. ----
         int rec_lo = 0, rec_hi = 0;                                           // records in this wave's two lists
         bool bad = false;
         if (cut != 2) {                                                   // (RTFE_CUT=2: stop behind the dense pre-filter)
            uint32_t mlo = (tm | bm) & 0xffffu, mhi = (tm | bm) >> 16;
            const int cnt = __popc(mlo) | (__popc(mhi) << 16);
            const int incl = wave_incl_scan(cnt, lane);
            const int totc = wave_last(incl);
            const int n_lo = totc & 0xffff, ncw = n_lo + (totc >> 16);
            bad = ncw > wave_cap;
            if (!bad && ncw > 0) {
               const int excl = incl - cnt;
               int o2 = excl & 0xffff;
               for (; mlo; mlo &= mlo - 1) { const int b2 = __ffs((int)mlo) - 1; wlist[o2++] = (uint16_t)((kSfStrip * lane + b2) | (((bm >> b2) & 1u) << 14)); }
               o2 = n_lo + (excl >> 16);
               for (; mhi; mhi &= mhi - 1) { const int b2 = __ffs((int)mhi) - 1; wlist[o2++] = (uint16_t)((kSfStrip * lane + b2) | (((bm >> (16 + b2)) & 1u) << 14) | 0x8000u); }
               rtfe_wave_sync();
               #pragma nounroll
               for (int r0 = 0; r0 < (cut == 3 ? 0 : ncw); r0 += 64) {
                  const int i = r0 + lane;
                  uint32_t w0 = 0, w1 = 0;
                  int half = 0, st = 0, cpos = 0, ckind = 0;
                  if (i < ncw) {
                     const uint32_t cd = wlist[i];
                     half = (int)(cd >> 15); cpos = (int)(cd & 0x3ffu); ckind = (int)((cd >> 14) & 1u);
                     st = pk_fast<WM>(cx, half ? h_hi : h_lo, cpos, ckind != 0, w0, w1); }
                  if (st == 2) {                                             // (0.06 % of the candidates of a clean NRZI tape)
                     const int hidx = atomicAdd(hard_count, 1);
                     if (hidx < hard_cap) {
                        SfHard hd; hd.tile = (uint32_t)tile; hd.pos = (uint16_t)cpos; hd.head = (uint8_t)(half ? h_hi : h_lo); hd.screen = (uint8_t)sc;
                        hard[hidx] = hd;
                        w0 = (uint32_t)hidx; w1 = 0xffff8001u; }
                     else { w0 = pk_w0(cpos, false, cpos + 1, 0, cx.W - 2, 0); w1 = 0xffff8000u; }      // (no room: "minimum unknown" at every row the sample could be tested at - the chain that gets there gives up)
                     st = 1; }
                  if (prof) { pn_hard += (unsigned)__popcll(__ballot(w1 == 0xffff8001u)); ++pn_rounds; }
                  if (cut == 4) { rec_lo += (int)(w0 & 1u); continue; }        // (RTFE_CUT=4: the evaluation without the placement)
                  const int vr = st;
                  const int sh = 16 * half;
                  const int ir = wave_incl_scan(vr << sh, lane);
                  const int myr = (((ir >> sh) & 0xffff) - vr) + (half ? rec_hi : rec_lo);
                  if (vr && 16 * (myr + 1) <= hcap) {                         // 16 bytes: the record, its margin block behind it
                     const lds_p slot = half ? slot_hi : slot_lo;
                     lds_u32p rp = reinterpret_cast<lds_u32p>(slot) + 4 * myr;
                     rp[0] = w0; rp[1] = w1;
                     if (cut != 5) pk_margins(cx, half ? h_hi : h_lo, w0, w1, reinterpret_cast<lds_u16p>(slot + 16 * (myr + 1))); }
                  const int tr = wave_last(ir);
                  rec_lo += tr & 0xffff; rec_hi += (tr >> 16) & 0xffff; } } }
         if (prof) { tk0 = clock64(); pc_own += tk0 - tk1; }
         // ---- 5. this wave's two lists leave: records from the front of each head's slot, margin entries from its back, 16 bytes per
         // lane; the directory ----
         rtfe_wave_sync();
         {
            #pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
               if (hh && !has_hi) break;
               const int h = h_lo + hh;
               const int nr = hh ? rec_hi : rec_lo;
               const bool over = bad || 16 * nr > hcap;
               unsigned char *gslot = pool + ((size_t)(tile * nscreens + sc) * ntrks + h) * (size_t)hcap;
               if (!over && nr > 0 && cut != 6) {
                  const int4 *src = reinterpret_cast<const int4 *>(smem + L.stage + h * hcap);
                  for (int v = lane; v < nr; v += 64) reinterpret_cast<int4 *>(gslot)[v] = src[v];      // (a 16-byte vector a record)
                  if (prof) pn_bytes += (unsigned)(16 * nr); }
               if (lane == 0) {
                  PeakDir d; d.nrec = over ? (uint16_t)0xffff : (uint16_t)nr; d.nent = 0;
                  dir[(size_t)(tile * nscreens + sc) * ntrks + h] = d; } } }
         rtfe_wave_sync(); }
      __syncthreads(); }
   if (tid == 0 && last_tile >= 0) sf_publish_quiet(s_noisy[par ^ 1], last_tile, nrows, qtile);
   if (prof && lane == 0) {
      atomicAdd(&dbg[0], (unsigned long long)pc_copy); atomicAdd(&dbg[1], (unsigned long long)pc_dense); atomicAdd(&dbg[2], (unsigned long long)pc_own);
      atomicAdd(&dbg[3], (unsigned long long)pn_bytes); atomicAdd(&dbg[4], (unsigned long long)pn_hard); atomicAdd(&dbg[5], (unsigned long long)pn_rounds);
      if (wave == 0) atomicAdd(&dbg[7], (unsigned long long)(last_tile >= 0 ? (last_tile - tile_lo) / G + 1 : 0)); } }

// ------------------------------------------------------------------------------------------------
// k_sift_s: k_sift for ONE window width W and a track count NT known at compile time (every LDS address an immediate offset, no
// loop over screens, no division), straight-line predicated code in the per-candidate part.  Same tile order, same lists.
// ------------------------------------------------------------------------------------------------
constexpr int kSfHardChunk = 32;      // places of the hard list a wave of k_sift_s takes at a time (<= 64: a wave marks what it leaves unused with one store)
__device__ __forceinline__ SfHard sf_hard_none() { SfHard h; h.tile = 0; h.pos = 0; h.head = 0xff; h.screen = 0; return h; }      // a place nobody took (k_sift_hard passes over it)
struct SfArgs {
   const int16_t *rows; long long nrows; int ntiles;
   uint16_t *qtile; PeakDir *dir; unsigned char *pool; SfHard *hard; int hard_cap; int *hard_count; unsigned long long *dbg;
   const DevCfg *cfg;      // the screen's thresholds are read where they live on the device (k_adapt_floor moves them between scans); hi_i does not depend on the floor
   int hcap, wave_cap, invert, quiet_i, hi_i, cut, debug;
   int nscreens, sc;      // several window widths: a launch per screen `sc` of `nscreens` (lists and directory are [tile][screen][head]); the quiet map comes from the launch that is given qtile
   int defer;      // 1: a tile's lists leave LDS at the start of the NEXT tile step (the stores' acknowledgements are then old when the step's first s_waitcnt vmcnt(0) - the prefetched rows - asks)
};

// PL ("plain"): no -invert, no experiment cut-offs, no cycle counters, the lists leave deferred - what every scan but a test's or a tool's is; the knobs are
// compile-time constants then (as run-time values they are a dozen wave-uniform masks the compiler keeps in - and reloads from - spilled scalar registers)
// RTFE_SIFT_PROF (a build of its own, tools/gpu_r6_prof.sh): where a wave's cycles of a tile step go - lane 0 of every wave reads the shader clock at the
// phase boundaries and adds the intervals up into SfArgs::dbg[0..7] = { tile -> LDS (with the wait for the prefetched rows), barrier 1, next tile's loads +
// the lists' copy-out, quiet groups, strips, compaction, owner rounds + tail, barrier 2 }
// RTFE_SIFT_PROF=2: the two largest of those again, in parts - { the next tile's loads, the quiet word, the pairs' lists out, the split head's out, a round's
// derivation, its placement, the step's tail, everything else }
#ifdef RTFE_SIFT_PROF
#define SF_PROF_DECL long long pf_t = clock64(), pf_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SF_PROF_AT(i) { const long long pf_n = clock64(); pf_acc[i] += pf_n - pf_t; pf_t = pf_n; }
#define SF_PROF_END if (lane == 0) { for (int pf_i = 0; pf_i < 8; ++pf_i) atomicAdd(&a.dbg[pf_i], (unsigned long long)pf_acc[pf_i]); }
#if RTFE_SIFT_PROF == 2
#define SF_PROF(i) SF_PROF_AT(7)
#define SF_PROF2(i) SF_PROF_AT(i)
#else
#define SF_PROF(i) SF_PROF_AT(i)
#define SF_PROF2(i)
#endif
#else
#define SF_PROF_DECL
#define SF_PROF(i)
#define SF_PROF2(i)
#define SF_PROF_END
#endif
template <int W, int NT, int WPS, bool PL>
__global__ void __launch_bounds__(64 * sfs_waves(NT), WPS) k_sift_s(const SfArgs a) {
#ifdef RTFE_CPU_EMUL
   unsigned char *smem = g_dyn_smem;
#else
   extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
   __shared__ unsigned int s_noisy[2];
   __shared__ unsigned short s_part[2][8];                               // split: records in each wave's part of the last head's list (0xffff: the list is unavailable), by tile parity
   constexpr bool SPL = sfs_split(NT);
   constexpr int NP = sfs_waves(NT), NTH = 64 * NP, RB = 2 * NT;
   constexpr int HL = sfs_hl(W), HR = sfs_hr(W);
   constexpr int NVEC = (HL + kSfTile + HR) * NT / 8, VPG = 8 * NT, VOWN0 = HL * NT / 8;
   constexpr int NV = (NVEC + NTH - 1) / NTH;
   constexpr int H3 = NT - 1, R3 = sfs_part_strip(NT), PR3 = sfs_part_rows(NT), LA3 = PR3 / R3;      // split: the last head, rows a lane / a wave screens of it, lanes at work
   const int tid = threadIdx.x, lane = tid & 63;
#ifdef RTFE_CPU_EMUL
   const int wave = tid >> 6;
#else
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
   const int hcap = a.hcap, wave_cap = a.wave_cap, cut = PL ? 0 : a.cut;
   const bool dbg3 = PL ? false : a.debug == 3, inv = PL ? false : a.invert != 0;
   const int cap3 = sfs_part_cap(NT, hcap);
   const SfsLds L = sfs_lds_layout(NT, W, wave_cap, hcap);
   unsigned char *xs = smem + L.xs;
   const lds_p xsl = to_lds(xs);
   PkCtx cx;
   cx.t.xs = xsl; cx.t.row_bytes = RB; cx.t.hl = HL; cx.t.sg = inv ? -1 : 1;
   const int lo_i = a.cfg->screen[a.sc].rise_i, minpk_i = a.cfg->screen[a.sc].minpk_i;
   cx.W = W; cx.lo_i = lo_i; cx.hi_i = a.hi_i;
   const int G = (int)gridDim.x, ntiles = a.ntiles;
   const int tile_lo = (G & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
   // a tile is "inside" when every row of it and of its halo exists (its bytes then come as 16-byte vectors, one tile ahead)
   const int inside_lo = (HL + kSfTile - 1) / kSfTile;
   const long long ih = (a.nrows - HR - 15) / kSfTile - 1;
   const int inside_hi = ih > 0x7ffffff0 ? 0x7ffffff0 : (int)ih;
   const int pair = wave, h_lo = 2 * pair, h_hi = 2 * pair + 1;
   const bool has_hi = h_hi < NT;
   const lds_u16p wlist = reinterpret_cast<lds_u16p>(to_lds(smem + L.wl)) + wave * (wave_cap + 8);
   const uint32_t at = minpk_i < 0 ? pk_dup(-32768) : pk_dup(minpk_i), ab = minpk_i < 0 ? pk_dup(32767) : pk_dup(-minpk_i);
   const uint32_t qpk = pk_dup(a.quiet_i), q2 = 2u * (uint32_t)a.quiet_i;
   int4 q[NV];
   #pragma unroll
   for (int k = 0; k < NV; ++k) q[k] = make_int4(0, 0, 0, 0);
   // (round 6: every round of the copy but the last is whole - no test, no branch; a round's address is a scalar base - the tile's, moved by the round's 4 KB
   //  on the scalar unit - and the lane's 32-bit offset, made once: no vector instruction per load)
   const unsigned voff = (unsigned)tid * 16u;
   auto fetch = [&](int tile, int tidx, unsigned voff) {
      const glb_cp src = (glb_cp)(a.rows + ((long long)tile * kSfTile - HL) * NT);
      #pragma unroll
      for (int k = 0; k < NV; ++k) {
         const glb_cp sk = sgpr_ptr(src + (size_t)k * NTH * 16);
         if ((k + 1) * NTH <= NVEC || k * NTH + tidx < NVEC) q[k] = glb_ld16(sk + voff); } };
   if (tile_lo < ntiles && tile_lo >= inside_lo && tile_lo <= inside_hi) fetch(tile_lo, tid, voff);
   if (tid < 2) s_noisy[tid] = 0;
   int par = 0, last_tile = -1;
   unsigned int pn_hard = 0, pn_rounds = 0, pn_bytes = 0;
   // Round 6: where a tile's lists go.  A wave's tile step is a CHAIN of dependent instructions and round trips, not a queue of work (tools/gpu_sift_prof.py:
   // every instruction of a wave, scalar ones and waits included, costs it 10 - 12 cycles beside the four other waves of its SIMD; the vector unit idles 28 %
   // of the time).  Rounds 3 - 5 staged every record in LDS and copied the three lists out at the start of the next step - a round trip to LDS, a store and a
   // directory store per list, the other waves' counts through LDS and v_readfirstlane: a quarter of the step.  Now a PAIR's records are stored where they
   // belong by the round that makes them (one store instruction a round, scalar base + the lane's offset); only the wave's part of the split head's list -
   // whose place depends on the other waves' counts - waits in LDS, and leaves with the tile's three directory entries behind the next step's barrier
   // (stores issued in front of a step's first s_waitcnt vmcnt(0) would be waited for in full: gfx9 counts loads and stores in one counter).
   int p_tile = -1, p_rec_lo = 0, p_rec_hi = 0;
   bool p_bad = false;
   int h_next = 0, h_end = 0;      // the wave's chunk of the hard list: [h_next, h_end) are its free places
   // (measured and not kept: the pair's records of a step's last round held in registers until that point too - their acknowledgements are on the way when the
   //  next step asks for its prefetched rows, ~1 000 cycles of its s_waitcnt vmcnt(0) -: the wait shrank, the store behind the barrier cost more: 0.643 / 0.632 ms)
   auto tile_pool = [&](const int tile) -> glb_p {      // the pool slot of the tile's head 0: [tile][screen][head]
      const unsigned li0 = ((unsigned)tile * (unsigned)a.nscreens + (unsigned)a.sc) * (unsigned)NT;
      return sgpr_ptr((glb_p)a.pool + (size_t)li0 * (size_t)(unsigned)hcap); };
   auto copy_part = [&](const int tile, const int rec_lo, const int rec_hi, const bool bad, const int pp, const int lanev) {
      const unsigned li0 = ((unsigned)tile * (unsigned)a.nscreens + (unsigned)a.sc) * (unsigned)NT;
      int off3 = 0, tot3 = 0, mine3 = 0;
      bool over3 = !SPL;
      if (SPL) {
         #pragma unroll
         for (int w2 = 0; w2 < NP; ++w2) {
#ifdef RTFE_CPU_EMUL
            const int c = s_part[pp][w2];
#else
            const int c = __builtin_amdgcn_readfirstlane((int)s_part[pp][w2]);      // (the same for every lane: the sums below stay in scalar registers)
#endif
            over3 = over3 || c == 0xffff;
            tot3 += c; if (w2 < wave) off3 += c; if (w2 == wave) mine3 = c; }
         over3 = over3 || 16 * tot3 > hcap; }
      if (!over3 && mine3 > 0 && cut != 6) {
         const glb_p g3 = sgpr_ptr((glb_p)a.pool + ((size_t)li0 + (size_t)H3) * (size_t)(unsigned)hcap + (size_t)off3 * 16);
         const int4 *src = reinterpret_cast<const int4 *>(smem + L.part + wave * cap3);
         #pragma nounroll
         for (int v0 = 0; v0 < mine3; v0 += 64) if (v0 + lanev < mine3) glb_st16(g3 + (unsigned)(v0 + lanev) * 16u, src[v0 + lanev]); }
      // the directory: lane 0 the pair's lower head, lane 1 its upper one, the next lane of wave 0 the split head
      const int ndir = (has_hi ? 2 : 1) + ((SPL && wave == 0) ? 1 : 0);
      if (lanev < 3) {
         const uint32_t d_lo = (bad || 16 * rec_lo > hcap) ? 0xffffu : (uint32_t)rec_lo, d_hi = (bad || 16 * rec_hi > hcap) ? 0xffffu : (uint32_t)rec_hi, d_3 = over3 ? 0xffffu : (uint32_t)tot3;
         const bool third = lanev == (has_hi ? 2 : 1);
         const uint32_t dv = third ? d_3 : (lanev == 0 ? d_lo : d_hi);      // (PeakDir: nrec in the low half, nent = 0)
         const int hd = third ? H3 : h_lo + lanev;
         const glb_p gdir = sgpr_ptr((glb_p)a.dir + (size_t)li0 * sizeof(PeakDir));
         if (lanev < ndir) glb_st4(gdir + (unsigned)hd * (unsigned)sizeof(PeakDir), dv); } };
   SF_PROF_DECL
   for (int tile = tile_lo; tile < ntiles; tile += G, par ^= 1) {
      // (conditions on the thread index alone are the same in every tile step: the compiler computes them once, as wave masks in scalar registers - more than
      //  it has, so it parks them in a vector register's lanes and fetches each back with two v_readlane where one v_cmp would do.  An index it cannot see
      //  through keeps the comparisons where they are used.)
      int tidl = tid, lanel = lane;
      unsigned voffl = voff;      // (... and the 32-bit offset stays one: hoisted as a 64-bit value it costs a 64-bit vector addition per load)
#ifndef RTFE_CPU_EMUL
      asm volatile("" : "+v"(tidl), "+v"(lanel), "+v"(voffl));
#endif
      last_tile = tile;
      const long long lastl = a.nrows - 1 - (long long)tile * kSfTile;
      cx.last = lastl > 0x3fffffff ? 0x3fffffff : (int)lastl;
      // ---- 1. the prefetched bytes -> LDS; the next tile's loads go out at once and travel while this tile is worked on ----
      if (tile >= inside_lo && tile <= inside_hi) {
         #pragma unroll
         for (int k = 0; k < NV; ++k) if ((k + 1) * NTH <= NVEC || k * NTH + tidl < NVEC) reinterpret_cast<int4 *>(xs)[k * NTH + tid] = q[k]; }
      else sf_fill_edge(xsl, a.rows, ((long long)tile * kSfTile - HL) * NT, a.nrows * NT, NVEC * 8, tid, NTH);
      SF_PROF(0)
      __syncthreads();
      SF_PROF(1)
      // (round 6: the stores of the lists in front FIRST, the next tile's loads behind them.  The memory pipeline of a CU is in order: behind the loads - twenty
      //  waves' 5 KB each, every line of them a miss - a store waited until the misses had drained, 1 500 cycles a list; a quarter of a tile step's time)
      if (tid == 0 && tile > tile_lo && a.qtile) {      // (every tile in front of the tape's last one is whole: its word is the complement of what the waves saw)
         a.qtile[tile - G] = (uint16_t)(~s_noisy[par ^ 1] & ((1u << kSfGroups) - 1u)); s_noisy[par ^ 1] = 0; }
      SF_PROF2(1)
      if (p_tile >= 0) { copy_part(p_tile, p_rec_lo, p_rec_hi, p_bad, par ^ 1, lanel); p_tile = -1; rtfe_wave_sync(); }      // (of the tile in front)
      SF_PROF2(2)
      if (tile + G < ntiles && tile + G >= inside_lo && tile + G <= inside_hi) fetch(tile + G, tidl, voffl);
      SF_PROF2(0)
      // ---- 2. quiet groups: flat 16-byte reads of the tile proper; a ballot of 64 vectors lies in one or two groups (several screens: the first launch's business) ----
      SF_PROF(2)
      // (round 6: a group is quiet only if EVERY sample of it is, and inside a block every part of a group carries signal - as k_quiet does, a group's first
      //  128 bytes are looked at first, eight lanes a group, and the rest of it - one more round of the wave - only where those are quiet: the gaps.  A wave
      //  takes three or four of the tile's fourteen groups.  The same bits; a tile inside a block costs one round of the old four.)
      // The step's LDS reads that depend on nothing but the tile are issued TOGETHER, ahead of the arithmetic on any of them - the groups' first lines, the
      // strips' rows, the split head's rows: one round trip for the three passes instead of three (a wave's step is a chain of latencies).
      static_assert(VPG - 8 <= 64 && (kSfGroups + NP - 1) / NP + 1 <= 8, "a group's rest is one round of a wave; a wave's groups' first lines, eight lanes each, too");
      const int g0 = wave * kSfGroups / NP, ng = (wave + 1) * kSfGroups / NP - g0;      // (3 4 3 4 of the fourteen groups for four waves, 4 5 5 for three)
      const int4 *tv = reinterpret_cast<const int4 *>(xs) + VOWN0;
      int4 qv = make_int4(0, 0, 0, 0);
      if (a.qtile && lanel < 8 * ng) qv = tv[(g0 + (lanel >> 3)) * VPG + (lanel & 7)];
      uint32_t x[kSfStrip + 2];
      const int r3 = wave * PR3 + R3 * (lane < LA3 ? lane : LA3 - 1);
      {  const lds_cp base = xsl + (HL + kSfStrip * lane) * RB + 4 * pair;
         // (the rows of a pair are only 2-byte aligned in LDS - rows are 2 NT bytes apart -, and a misaligned ds_read_b32 costs 33 cycles a wave: two aligned 16-bit reads per row)
         #pragma unroll
         for (int i = 0; i < kSfStrip + 2; ++i) x[i] = (uint32_t)(uint16_t)lds_i16(base + (i - 1) * RB) | ((uint32_t)(uint16_t)lds_i16(base + (i - 1) * RB + 2) << 16);
      }
      if (a.qtile) {
         const uint32_t q2pk = pk_dup((int)q2);
         auto noisy4 = [&](const int4 v) -> bool {
            const uint32_t m = pk_maxu(pk_maxu(pk_addu((uint32_t)v.x, qpk), pk_addu((uint32_t)v.y, qpk)),
                                       pk_maxu(pk_addu((uint32_t)v.z, qpk), pk_addu((uint32_t)v.w, qpk)));
            return pk_maxu(m, q2pk) != q2pk; };
         const u64 nb = __ballot(noisy4(qv));                                       // (lanes that read nothing hold zeros: quiet)
         // byte b of nb = the eight lanes of group g0 + b: "any bit of the byte" folded into its lowest bit, the eight of them gathered (scalar, no branch)
         u64 z = nb | (nb >> 4); z |= z >> 2; z |= z >> 1; z &= 0x0101010101010101ull;
         unsigned int bits = (unsigned int)((z * 0x0102040810204080ull) >> 56);      // (relative to g0) groups known to be noisy
         unsigned int redo = ~bits & ((1u << ng) - 1u);                             // ... whose first line is quiet
         for (; redo; redo &= redo - 1) {                                            // (a gap): the rest of the group
            const int b = __ffs((int)redo) - 1;
            bool n2 = false;
            if (lanel < VPG - 8) n2 = noisy4(tv[(g0 + b) * VPG + 8 + lanel]);
            if (__ballot(n2)) bits |= 1u << b; }
         if (lane == 0 && bits) atomicOr(&s_noisy[par], bits << g0); }
      SF_PROF(3)
      if (cut != 1) {
         // ---- 3. candidate samples: local extremum + amplitude, one lane per 14-row strip of a pair of heads.  The rows of a pair are
         // only 2-byte aligned in LDS (rows are 2 NT bytes apart), and a misaligned ds_read_b32 costs 33 cycles a wave: two aligned
         // 16-bit reads per row instead ----
         uint32_t tm = 0, bm = 0;
         {
            // (round 6: the rows' "rising into" / "falling into" bits are gathered first - bit j of a half: x[j] lies above x[j - 1] on the signal clamped at the
            //  floor / below it on the signal clamped at the ceiling - and "rises into the row and not out of it" is made once per strip from the two masks,
            //  not per row: 8 instead of 11 vector instructions a row)
            uint32_t yc = pk_max(x[0], at), zc = pk_min(x[0], ab), ru = 0, fd = 0;
            #pragma unroll
            for (int i = 1; i < kSfStrip + 2; ++i) {
               const uint32_t yn = pk_max(x[i], at), zn = pk_min(x[i], ab);
               ru = (ru >> 1) | (pk_subs(yc, yn) & kPkSigns);
               fd = (fd >> 1) | (pk_subs(zn, zc) & kPkSigns);
               yc = yn; zc = zn; }
            // x[1 .. kSfStrip + 1] sit at bits 1 .. 15 of each half (16 - (kSfStrip + 1) = 1 from the top: kSfStrip = 14), bit 0 is clear: row r = x[r + 1]
            static_assert(kSfStrip == 14, "the strip's masks fill the halves of a register");
            tm = ((ru & ~(ru >> 1)) >> 1) & 0x3fff3fffu; bm = ((fd & ~(fd >> 1)) >> 1) & 0x3fff3fffu;
            if (inv) { const uint32_t s2 = tm; tm = bm; bm = s2; }
            if (!has_hi) { tm &= 0xffffu; bm &= 0xffffu; }                  // odd track count: the last pair's upper half is the next row
            if (cx.last < kSfTile - 1) {                                      // rows that do not exist cannot own a run
               const int keep = cx.last + 1 - kSfStrip * lane;
               const uint32_t mk = keep >= kSfStrip ? 0x3fffu : (keep <= 0 ? 0u : ((1u << keep) - 1u));
               tm &= mk | (mk << 16); bm &= mk | (mk << 16); } }
         // split: the same for this wave's part of the last head - R3 rows a lane, the sample in the low half of the packed operations
         uint32_t m3 = 0, b3 = 0;                                             // candidates / the bottoms among them, bit i = row r3 + i
         if (SPL) {
            uint32_t x3[R3 + 2];      // (these six reads stay here: hoisted with the others they cost the kernel its seventh wave per SIMD - a prefetched vector spilled)
            const lds_cp base3 = xsl + (HL + r3) * RB + 2 * H3;
            #pragma unroll
            for (int i = 0; i < R3 + 2; ++i) x3[i] = (uint32_t)(uint16_t)lds_i16(base3 + (i - 1) * RB);
            uint32_t yc = pk_max(x3[0], at), zc = pk_min(x3[0], ab), ru = 0, fd = 0;
            #pragma unroll
            for (int i = 1; i < R3 + 2; ++i) {
               const uint32_t yn = pk_max(x3[i], at), zn = pk_min(x3[i], ab);
               ru = (ru >> 1) | (pk_subs(yc, yn) & 0x8000u);
               fd = (fd >> 1) | (pk_subs(zn, zc) & 0x8000u);
               yc = yn; zc = zn; }
            static_assert(R3 <= 14, "the part's masks fill the low half of a register");
            uint32_t t3 = ((ru & ~(ru >> 1)) >> (15 - R3)) & ((1u << R3) - 1u);      // (x[1 .. R3 + 1] sit at bits 15 - R3 .. 15: row i = x[i + 1])
            b3 = ((fd & ~(fd >> 1)) >> (15 - R3)) & ((1u << R3) - 1u);
            if (inv) { const uint32_t s2 = t3; t3 = b3; b3 = s2; }
            int keep = kSfTile - r3;                                          // the part's last lanes reach into the next tile (seven tracks) or behind the tape's end
            if (cx.last + 1 - r3 < keep) keep = cx.last + 1 - r3;
            if (lane >= LA3) keep = 0;
            const uint32_t mk = keep >= R3 ? (1u << R3) - 1u : (keep <= 0 ? 0u : ((1u << keep) - 1u));
            m3 = (t3 | b3) & mk; b3 &= mk; }
         SF_PROF(4)
         // ---- 4. the wave's candidates, compacted into a list ordered by (head, row); rounds of 64 ----
         int rec_lo = 0, rec_hi = 0, rec_3 = 0;                                // records in this wave's lists
         const glb_p gtile = tile_pool(tile);
         bool bad = false;
         if (cut != 2) {
            uint32_t mlo = (tm | bm) & 0xffffu, mhi = (tm | bm) >> 16;
            const int cnt = __popc(mlo) | (__popc(mhi) << 10) | (__popc(m3) << 20);      // (a head's 896 rows hold fewer than 1 024 extrema)
            const int incl = wave_incl_scan(cnt, lane);
            const int totc = wave_last(incl);
            const int n_lo = totc & 0x3ff, n_hi = (totc >> 10) & 0x3ff, ncw = n_lo + n_hi + (totc >> 20);
            bad = ncw > wave_cap;
            if (!bad && ncw > 0) {
               const int excl = incl - cnt;
               // (round 6: a lane's first two candidates of each list are listed by straight-line code - a lane without one stores to the spare entry behind the
               //  list -, what is left by the loops: a strip of 14 rows seldom holds more than two flux transitions, and the loops' branches were a tenth of
               //  a step's instructions)
               auto put2 = [&](uint32_t &m, int &o2, const int rowbase, const uint32_t bots, const uint32_t tag) {
                  #pragma unroll
                  for (int u = 0; u < 2; ++u) {
                     const bool has = m != 0u;
                     const int b2 = has ? __ffs((int)m) - 1 : 0;
                     wlist[has ? o2 : wave_cap] = (uint16_t)((uint32_t)(rowbase + b2) | (((bots >> b2) & 1u) << 14) | tag);
                     o2 += has ? 1 : 0; m &= m - 1u; } };
               int o_lo = excl & 0x3ff, o_hi = n_lo + ((excl >> 10) & 0x3ff), o_3 = n_lo + n_hi + (excl >> 20);
#ifndef RTFE_SF_PUT2
#define RTFE_SF_PUT2 0      /* measured: 0.645 ms with it, 0.632 without - the straight-line code costs more vector instructions than the loops' branches it saves */
#endif
               if (RTFE_SF_PUT2) {
                  put2(mlo, o_lo, kSfStrip * lane, bm, 0u);
                  put2(mhi, o_hi, kSfStrip * lane, bm >> 16, 0x8000u);
                  if (SPL) put2(m3, o_3, r3, b3, 0x2000u); }
               if (__ballot((mlo | mhi | m3) != 0u)) {
                  for (; mlo; mlo &= mlo - 1) { const int b2 = __ffs((int)mlo) - 1; wlist[o_lo++] = (uint16_t)((kSfStrip * lane + b2) | (((bm >> b2) & 1u) << 14)); }
                  for (; mhi; mhi &= mhi - 1) { const int b2 = __ffs((int)mhi) - 1; wlist[o_hi++] = (uint16_t)((kSfStrip * lane + b2) | (((bm >> (16 + b2)) & 1u) << 14) | 0x8000u); }
                  if (SPL) for (; m3; m3 &= m3 - 1) { const int b2 = __ffs((int)m3) - 1; wlist[o_3++] = (uint16_t)((r3 + b2) | (((b3 >> b2) & 1u) << 14) | 0x2000u); } }
               rtfe_wave_sync();
               SF_PROF(5)
               #pragma nounroll
               for (int r0 = 0; r0 < (cut == 3 ? 0 : ncw); r0 += 64) {
                  const int i = r0 + lane;
                  const bool live = i < ncw;
                  const uint32_t cd = live ? wlist[i] : 0u;
                  const int sel = SPL && (cd & 0x2000u) ? 2 : (int)(cd >> 15);      // the pair's lower / upper head, the last head's part
                  const int cpos = (int)(cd & 0x3ffu);
                  const int head = sel == 2 ? H3 : h_lo + sel;
                  const bool cbot = (cd >> 14) & 1u;
                  uint32_t w0 = 0, w1 = 0;
                  int st = pk_fast_w<W>(cx, head, cpos, cbot, w0, w1);
                  if (!live) st = 0;
                  SF_PROF2(4)
                  // Deferred candidates (0.004 % of a clean NRZI tape's, 5 % at 60 mV rms of noise: 2.5 a wave and tile step) take their places in the hard list
                  // from a CHUNK the wave owns: one atomic on the list's counter per kSfHardChunk of them.  (Round 6: an atomic a round - 800 000 of them on one
                  // address, each waited for with the prefetched rows in flight - was what a noisy tape's k_sift_s spent its time on: 7.4 ms against 0.63.)
                  const u64 hm = __ballot(st == 2);
                  if (hm) {
                     const int nh = __popcll(hm);
                     if (h_next + nh > h_end) {
                        if (h_next + lane < h_end && h_next + lane < a.hard_cap) a.hard[h_next + lane] = sf_hard_none();      // (what is left of the old chunk: nobody's)
                        const int want = nh > kSfHardChunk ? nh : kSfHardChunk;
                        int base = 0;
                        if (lane == 0) base = atomicAdd(a.hard_count, want);
#ifdef RTFE_CPU_EMUL
                        base = __shfl(base, 0);
#else
                        base = __builtin_amdgcn_readfirstlane(base);
#endif
                        h_next = base; h_end = base + want; }
                     if (st == 2) {
                        const int hidx = h_next + __popcll(hm & ((1ull << lane) - 1ull));
                        if (hidx < a.hard_cap) {
                           SfHard hd; hd.tile = (uint32_t)tile; hd.pos = (uint16_t)cpos; hd.head = (uint8_t)head; hd.screen = (uint8_t)a.sc;
                           a.hard[hidx] = hd;
                           w0 = (uint32_t)hidx; w1 = 0xffff8001u; }
                        else { w0 = pk_w0(cpos, false, cpos + 1, 0, cx.W - 2, 0); w1 = 0xffff8000u; }      // (no room: "minimum unknown" at every row the sample could be tested at - the chain that gets there gives up)
                        st = 1; }
                     h_next += nh; }
                  if (dbg3) { pn_hard += (unsigned)__popcll(__ballot(w1 == 0xffff8001u)); ++pn_rounds; }
                  if (cut == 4) { rec_lo += (int)(w0 & 1u); continue; }        // (RTFE_CUT=4: the evaluation without the placement)
                  const int vr = st;
                  const int sh = 8 * sel;                                      // (a round adds at most 64 to a list)
                  const int ir = wave_incl_scan(vr << sh, lane);
                  const int myr = (((ir >> sh) & 0xff) - vr) + (sel == 2 ? rec_3 : (sel ? rec_hi : rec_lo));
                  if (vr && 16 * (myr + 1) <= (sel == 2 ? cap3 : hcap)) {      // 16 bytes: the record, its margin block behind it
                     const uint2 mb = (cut == 5 || (w1 & 0xfffffffeu) == 0xffff8000u) ? make_uint2(0, 0) : pk_margins_w<W>(cx, head, cpos, cbot, w0);
                     const int4 rec = make_int4((int)w0, (int)w1, (int)mb.x, (int)mb.y);
                     if (sel == 2) *reinterpret_cast<int4 *>(smem + L.part + wave * cap3 + 16 * myr) = rec;      // (the split head's part: staged)
                     else if (cut != 6) glb_st16(gtile + (unsigned)((h_lo + sel) * hcap + 16 * myr), rec); }         // (the pair's lists: where they belong)
                  const int tr = wave_last(ir);
                  rec_lo += tr & 0xff; rec_hi += (tr >> 8) & 0xff; rec_3 += (tr >> 16) & 0xff;
                  SF_PROF2(5) } } }
         // ---- 5. this wave's two lists leave (now, or - a.defer - at the start of the next tile step); its part of the last head's waits for the others' counts ----
         rtfe_wave_sync();
         if (SPL && lane == 0) s_part[par][wave] = (unsigned short)((bad || 16 * rec_3 > cap3) ? 0xffff : rec_3);
         p_tile = tile; p_rec_lo = rec_lo; p_rec_hi = rec_hi; p_bad = bad;
         rtfe_wave_sync(); }
      SF_PROF2(6)
      SF_PROF(6)
      __syncthreads();
      SF_PROF(7) }
   SF_PROF_END
   if (p_tile >= 0) copy_part(p_tile, p_rec_lo, p_rec_hi, p_bad, par ^ 1, lane);
   if (h_next + lane < h_end && h_next + lane < a.hard_cap) a.hard[h_next + lane] = sf_hard_none();      // (the unused places of the wave's last chunk)
   if (tid == 0 && last_tile >= 0 && a.qtile) sf_publish_quiet(s_noisy[par ^ 1], last_tile, a.nrows, a.qtile);
   if (dbg3 && lane == 0) {
      atomicAdd(&a.dbg[3], (unsigned long long)pn_bytes); atomicAdd(&a.dbg[4], (unsigned long long)pn_hard); atomicAdd(&a.dbg[5], (unsigned long long)pn_rounds);
      if (wave == 0) atomicAdd(&a.dbg[7], (unsigned long long)(last_tile >= 0 ? (last_tile - tile_lo) / G + 1 : 0)); } }

// ------------------------------------------------------------------------------------------------
// k_sift_hard: the candidates k_sift deferred (bottoms whose first candidate rows precede every forced rescan: the reference's
// stale minimum has to be followed through its chain of rescans).  One lane per candidate, the samples straight from
// HBM (a few hundred 2-byte reads each; 0.06 % of the candidates of a clean tape).  Out: the candidate's overflow slot - up to
// four records and their margin entries in the layout of a list slot.
// ------------------------------------------------------------------------------------------------
#ifndef RTFE_HARD_PROF
#define RTFE_HARD_PROF 0
#endif
constexpr int kHardCol = 320;      // samples of a candidate's head its wave keeps in LDS: kPkBack + 3 W + 16 <= 230 for W <= 50, what is outside comes from HBM
#ifdef RTFE_CPU_EMUL
#define RTFE_HARD_ATTR
#else
#define RTFE_HARD_ATTR __attribute__((amdgpu_waves_per_eu(6)))      // 79 registers, nothing spilled (82 without the hint): six waves a SIMD - the walk waits on LDS, waves hide it
#endif
__global__ void __launch_bounds__(256) RTFE_HARD_ATTR k_sift_hard(const DevCfg *__restrict__ cfgp, const int16_t *__restrict__ rows, long long nrows,
                                                   const SfHard *__restrict__ hard, int hard_cap, const int *__restrict__ hard_count, unsigned char *__restrict__ ovf, int *__restrict__ extra, unsigned long long *__restrict__ dbg) {
   // SIXTEEN LANES per candidate (round 6; a wave per candidate before): they fetch the head's samples the walk can read - one round trip -, make "rescan
   // forced" and "leftmost window minimum" for every row around the candidate (a few rows a lane), and the first of them walks on those tables.  What a
   // candidate costs is that walk - one lane's loops over table entries, tens of microseconds -, so four of them share a wave: a tape with 60 mV rms of noise
   // defers 1.2 million candidates per 1e8 rows (k_sift_hard 3.9 ms of a 9 ms scan), a clean one a few dozen.
   // (Round 4: the walking lane made the tables as it went - pk_async over a window at every row it stepped back: ~1 500 dependent LDS reads, 60 - 100 us a candidate.)
   constexpr int kSub = 16, kGroups = 256 / kSub;
   __shared__ int16_t s_col[kGroups][kHardCol];
   const DevCfg &cfg = *cfgp;
   const int sl = threadIdx.x & (kSub - 1), wv = threadIdx.x / kSub;
   int n = *hard_count;
   if (n > hard_cap) n = hard_cap;
#if RTFE_HARD_PROF
   const bool prof = cfg.debug == 9 && threadIdx.x == 0;                  // (a build with -DRTFE_HARD_PROF=1, RTFE_DEBUG=9: cycles of the first wave's phases - fetch, walk - and its trips)
#else
   constexpr bool prof = false;                                          // (the counters' registers cost the kernel a wave a SIMD)
#endif
   long long pt[4] = {0, 0, 0, 0};
   for (int i0 = blockIdx.x * kGroups; i0 < n; i0 += gridDim.x * kGroups) {      // (the same trips for every lane of the workgroup: the wave-level fences below)
      const int i = i0 + wv;
      long long tq = prof ? clock64() : 0;
      SfHard hd = sf_hard_none();                                         // (loaded a trip ahead it changed nothing: 12.0 k cycles of fetch either way)
      if (i < n) hd = hard[i];
      const bool live = hd.head != 0xff;                                  // (0xff: a place of a wave's chunk that no candidate took, or none at all)
      if (!live) { hd.head = 0; hd.screen = 0; }
      const DevScreen &S = cfg.screen[hd.screen];
      PkCtxT<PkCol> cx;
      cx.t.tape.rows = rows; cx.t.tape.t0 = (long long)hd.tile * kSfTile; cx.t.tape.nrows = nrows; cx.t.tape.ntrks = cfg.ntrks; cx.t.tape.sg = cfg.invert ? -1 : 1;
      cx.W = S.W; cx.lo_i = S.rise_i; cx.hi_i = S.sure_i;
      const long long lastl = nrows - 1 - cx.t.tape.t0;
      cx.last = lastl > 0x3fffffff ? 0x3fffffff : (int)lastl;
      const int r0 = (int)hd.pos - kPkBack - 2 * S.W - 8;
      int ncol = kPkBack + 3 * S.W + 16;
      if (ncol > kHardCol) ncol = kHardCol;
      rtfe_wave_sync();
      // (a 2-byte load a row - 119 requests a candidate, 12 k cycles a trip; the rows' bytes read as ONE range in aligned 16-byte pieces, each holding the head's sample of
      //  one row, took 15.9 k: finding the sample in its piece costs more than the requests it saves - measured, round 6)
      if (live) for (int k = sl; k < ncol; k += kSub) s_col[wv][k] = (int16_t)cx.t.tape.at(r0 + k, (int)hd.head);
      rtfe_wave_sync();
      if (prof) { const long long t = clock64(); pt[0] += t - tq; tq = t; }
      cx.t.col = s_col[wv]; cx.t.r0 = r0; cx.t.n = ncol; cx.t.head = (int)hd.head;
      // (no tables of "rescan forced" / "leftmost window minimum" for every row around the candidate any more: sixteen lanes made 106 rows' worth - 7.6 us of a trip's 17.7 -
      //  and the group's walk asks for a pass or two of them: each lane makes its row's answer when it is asked, W LDS reads)
      if (prof) { const long long t = clock64(); pt[1] += t - tq; tq = t; ++pt[3]; }
      PkSink sk; sk.n = 0;
      {  // (every row the walk can ask for is in the copy - kPkBack + 3 W + 16 <= kHardCol for the windows the peak path takes, W <= 50 - rows outside the tape as zeros)
         PkCtxT<PkColFast> cf;
         cf.t.col = s_col[wv]; cf.t.r0 = r0; cf.W = cx.W; cf.lo_i = cx.lo_i; cf.hi_i = cx.hi_i; cf.last = cx.last;
         pk_bot_grp(cf, sk, (int)hd.head, (int)hd.pos, live && kPkBack + 3 * cx.W + 16 <= kHardCol, sl, (int)(threadIdx.x & 63 & ~(kSub - 1))); }
      if (live && kPkBack + 3 * cx.W + 16 > kHardCol) sk.n = 5;             // (a window wider than the front end takes: "minimum unknown" below, never silence)
#ifdef RTFE_CPU_EMUL
      if (live && sl == 0 && getenv("RTFE_HARD_CHECK")) {             // (emulator: the group's walk against one lane's)
         PkSink s2; s2.n = 0;
         pk_bot(cx, s2, (int)hd.head, (int)hd.pos);
         bool same = s2.n == sk.n;
         for (int j = 0; same && j < s2.n && j < 4; ++j) same = s2.w0[j] == sk.w0[j] && s2.w1[j] == sk.w1[j];
         if (!same) fprintf(stderr, "hard_check: candidate %d (tile %u pos %d head %d): the group's walk made %d records (%08x %08x ..), one lane's %d (%08x %08x ..)\n", i, hd.tile, (int)hd.pos, (int)hd.head, sk.n, sk.w0[0], sk.w1[0], s2.n, s2.w0[0], s2.w1[0]); }
#endif
      if (!live || sl != 0) continue;
      unsigned char *slot = ovf + (size_t)i * kSfOvfBytes;
      int nrec = sk.n;
      if (nrec > 4) nrec = -1;
      if (nrec < 0) { nrec = 1; sk.w0[0] = pk_w0((int)hd.pos, false, (int)hd.pos + 1, 0, cx.W - 2, 0); sk.w1[0] = 0xffff8000u; }      // (more epochs than a slot holds: "minimum unknown" at every row the sample could be tested at - the chain that gets there gives up)
      *reinterpret_cast<int *>(slot) = nrec;
      *reinterpret_cast<int *>(slot + 4) = (int)hd.pos;                   // (the candidate's row in its tile: k_prep - a stale minimum's record is owned by a sample in front of it)
      if (nrec != 1) atomicAdd(&extra[((size_t)hd.tile * cfg.nscreens + hd.screen) * cfg.ntrks + hd.head], nrec - 1);      // the list's length in its stream (k_pscan)
      for (int j = 0; j < nrec; ++j) {                                      // (8 + 4 x 16 bytes fit the slot; behind them, from byte 72, kCrWeak and its bits per record)
         reinterpret_cast<uint32_t *>(slot + 8)[4 * j] = sk.w0[j]; reinterpret_cast<uint32_t *>(slot + 8)[4 * j + 1] = sk.w1[j];
         uint16_t mb[kPkMar];
         pk_margins(cx, (int)hd.head, sk.w0[j], sk.w1[j], mb + kPkMar);
         const uint2 blk = make_uint2((uint32_t)mb[0] | ((uint32_t)mb[1] << 16), (uint32_t)mb[2] | ((uint32_t)mb[3] << 16));      // (the block as it lies in memory: entry j at its end - 2 (j + 1))
         *reinterpret_cast<uint2 *>(slot + 8 + 16 * j + 8) = blk;
         reinterpret_cast<uint32_t *>(slot + 72)[j] = crec_weak_bits(sk.w0[j], sk.w1[j], blk); }
      if (prof) pt[2] += clock64() - tq; }
   if (prof) { atomicAdd(&dbg[0], (unsigned long long)pt[0]); atomicAdd(&dbg[1], (unsigned long long)pt[1]); atomicAdd(&dbg[2], (unsigned long long)pt[2]); atomicAdd(&dbg[3], (unsigned long long)pt[3]); atomicAdd(&dbg[4], 1ull); } }

}  // namespace rtfe
