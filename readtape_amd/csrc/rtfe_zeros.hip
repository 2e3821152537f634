// rtfe_zeros.hip — k_zeros: the -zeros front end (lookfor_zerocrossing, src/decoder.c:617-649) as a kernel of its own.
//
// The zero-crossing detector has no AGC feedback and no window: its state per track is two extremes, two "crossing pending" flags
// and the previous sample, all of which fit 16 bits.  k_zeros therefore runs TWO tracks per lane on the packed 16-bit vector
// instructions (v_pk_max_i16 / v_pk_min_i16 / v_pk_sub_i16 with saturation, the three-input v_bitop3_b32 for the flag logic): the two
// tracks are neighbouring columns of a row, so one (unaligned) 32-bit load where the row lies in HBM brings both samples, already
// packed.  27 vector instructions per row and PAIR of samples, where the 32-bit step of k_decode's mode (zc_step32) takes 35 per
// sample.  Events are not written inside the step: a lane keeps one bit per row and track for "crossing confirmed" and one for
// "crossing armed" (two words per 16 rows), and the events are made from the bitmaps afterwards - the sample is read again, the
// row of the sign change is the last armed row in front of the event.
//
// A burst is one workgroup's (persistent workgroups take bursts from a queue, the long ones first).  Its rows [restart, stop) are
//   head   32 rows, sequentially (a walker per track: the tracks start staggered, src/decoder.c:855-861),
//   chunks of up to 128 / (pairs of columns) sub-segments of 128 rows: a lane per (sub-segment, column pair).  Sub-segment 0 continues
//          from the walkers' true state; the others start zc_warm rows early - from a fresh state inside the block (what a restart would
//          be), from the walkers' state where the next dead-quiet zone has begun.  The detector forgets: after a confirmed crossing in
//          each direction its state is a function of the samples since.  A sub-segment's result stands if the state it reached at its
//          first own row is EQUIVALENT to its predecessor's final state - the same pending flags, and the same extremes where they
//          matter: an extreme that has not reached the threshold (top < P, bot > -P) takes part in nothing but max / min with later
//          samples, so all such values are one state (quiet stretches, where the extremes never reset, join like the rest).  Where
//          joins fail, a track's first failing sub-segment is run again from its predecessor's end state - the true state, by
//          induction from sub-segment 0; the check repeats until every join holds,
//   tail   the < 128 rows left, sequentially.
// Exact for every input: the parallel part changes who computes, never what.  -invert, -deskew, one track and thresholds beyond
// int16 take k_decode's zero-crossing mode instead (rtfe_api.hip).  Included behind rtfe_kernels.hip.

namespace rtfe {

struct ZWalker {           // what lookfor_zerocrossing keeps per track (+ where the walk stands)
   long long start, next;
   int   z_prev, z_top, z_bot;
   bool  z_up_pending, z_dn_pending;
   long long z_ttop_row, z_tbot_row;
   unsigned int nevents, flags;
};

typedef unsigned int u32;
constexpr int kZpSub = 128;           // rows of a sub-segment (a multiple of 64)
constexpr int kZpLong = 32768;        // bursts of that many rows and more are taken first, those under a quarter of it last
constexpr int kZpHead = 32;           // rows walked sequentially at a burst's start (> RTFE_MAXTRKS: every track has started)
constexpr int kZpThreads = 128;         // (measured on C3: 64 / 128 / 256 / 512 threads -> 15.9 / 14.5 / 14.8 / 16.8 ms; sub-segments of 256 rows: 15.7)
constexpr int kZpStage = 16;           // event samples a half stages in LDS (the 9 state fields of the records: 18 rows of 16 bits)
constexpr int kZpAhead = 2;           // batches of eight rows on their way while a batch is stepped through
enum { kZfTopS, kZfBotS, kZfPuS, kZfPdS, kZfTopE, kZfBotE, kZfPuE, kZfPdE, kZfPvE, kZfCnt, kZfEvm, kZfArm = kZfEvm + kZpSub / 16, kZfN = kZfArm + kZpSub / 16 };

__device__ __forceinline__ u32 zp_mk16(int lo, int hi) { return ((u32)lo & 0xffffu) | ((u32)hi << 16); }
#ifdef RTFE_CPU_EMUL
typedef const char *gptr8;
__device__ __forceinline__ int zp_lo(u32 a) { return (int)(short)(a & 0xffff); }
__device__ __forceinline__ int zp_hi(u32 a) { return (int)(short)(a >> 16); }
__device__ __forceinline__ u32 zp_mk(int lo, int hi) { return zp_mk16(lo, hi); }
__device__ __forceinline__ int zp_sat(int x) { return x > 32767 ? 32767 : (x < -32768 ? -32768 : x); }
__device__ __forceinline__ u32 zq_max(u32 a, u32 b) { return zp_mk(max(zp_lo(a), zp_lo(b)), max(zp_hi(a), zp_hi(b))); }
__device__ __forceinline__ u32 zq_min(u32 a, u32 b) { return zp_mk(min(zp_lo(a), zp_lo(b)), min(zp_hi(a), zp_hi(b))); }
__device__ __forceinline__ u32 zq_subs(u32 a, u32 b) { return zp_mk(zp_sat(zp_lo(a) - zp_lo(b)), zp_sat(zp_hi(a) - zp_hi(b))); }
__device__ __forceinline__ u32 zq_add(u32 a, u32 b) { return zp_mk(zp_lo(a) + zp_lo(b), zp_hi(a) + zp_hi(b)); }
__device__ __forceinline__ u32 zq_sub(u32 a, u32 b) { return zp_mk(zp_lo(a) - zp_lo(b), zp_hi(a) - zp_hi(b)); }
__device__ __forceinline__ u32 zq_sar15(u32 a) { return zp_mk(zp_lo(a) >> 15, zp_hi(a) >> 15); }
#else
typedef const __attribute__((address_space(1))) char *gptr8;
typedef short pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk16 zq_v(u32 a) { return __builtin_bit_cast(pk16, a); }
__device__ __forceinline__ u32 zq_u(pk16 a) { return __builtin_bit_cast(u32, a); }
__device__ __forceinline__ u32 zq_max(u32 a, u32 b) { return zq_u(__builtin_elementwise_max(zq_v(a), zq_v(b))); }
__device__ __forceinline__ u32 zq_min(u32 a, u32 b) { return zq_u(__builtin_elementwise_min(zq_v(a), zq_v(b))); }
__device__ __forceinline__ u32 zq_subs(u32 a, u32 b) { return zq_u(__builtin_elementwise_sub_sat(zq_v(a), zq_v(b))); }       // (saturating)
__device__ __forceinline__ u32 zq_add(u32 a, u32 b) { return zq_u(zq_v(a) + zq_v(b)); }
__device__ __forceinline__ u32 zq_sub(u32 a, u32 b) { return zq_u(zq_v(a) - zq_v(b)); }
__device__ __forceinline__ u32 zq_sar15(u32 a) { return zq_u(zq_v(a) >> 15); }                                                 // 0xffff where the half is negative
#endif
// the two samples of a column pair at byte offset `off` from the (uniform) base: bytes that lie 2-aligned, not 4-aligned
__device__ __forceinline__ u32 zp_load(gptr8 base, u32 off) { u32 x; __builtin_memcpy(&x, (const char *)(base + off), 4); return x; }

// The detector's step (zc_row / zc_step32: the reference's statements in their order) for two tracks, one per 16-bit half.  The flags
// live in the SIGN bits: pu / pd = a crossing upward / downward is pending; the other bits of a flag word are not defined.  pv / pnv =
// the previous sample and its (saturated) negation: "previous sample negative / positive" are their sign bits.
//   pos clears pd, neg clears pu; v > max(top, P - 1) with pu pending confirms upward (top >= 0 always, so "a new maximum that
//   reaches the threshold" is that one compare), v < min(bot, 1 - P) with pd pending downward; the extremes follow, a confirmation
//   takes its flag and zeroes the OTHER extreme; then the arming: pos after neg with bot <= -P, neg after pos with top >= P.
// evm / arm collect `bit` (the row's bit in both halves) where a crossing was confirmed / armed.
struct ZpState { u32 top, bot, pu, pd, pv, pnv; };
__device__ __forceinline__ void zp_step(ZpState &z, u32 v, u32 Pm1, u32 mP1, u32 bit, u32 &evm, u32 &arm) {
   const u32 nv = zq_subs(0u, v);                                    // sign: v > 0 (-32768 saturates to 32767)
   const u32 du = zq_subs(zq_max(z.top, Pm1), v);                    // sign: v > max(top, P - 1)
   const u32 dd = zq_subs(v, zq_min(z.bot, mP1));                    // sign: v < min(bot, 1 - P)
   const u32 eu = du & z.pu, ed = dd & z.pd;                         // confirmed (v > max(..) >= 0 is a positive sample: the clearing of pu by a negative one cannot matter here)
   z.top = zq_max(z.top, v); z.bot = zq_min(z.bot, v);
   const u32 meu = zq_sar15(eu), med = zq_sar15(ed);
   z.bot &= ~meu; z.top &= ~med;
   const u32 au = nv & z.pv & zq_add(z.bot, Pm1);                    // sign: pos, previous neg, bot <= -P
   const u32 ad = v & z.pnv & zq_sub(Pm1, z.top);                    // sign: neg, previous pos, top >= P
   z.pu = (z.pu & ~v & ~du) | au;
   z.pd = (z.pd & ~nv & ~dd) | ad;
   evm |= (meu | med) & bit;
   arm |= zq_sar15(au | ad) & bit;
   z.pv = v; z.pnv = nv; }

// One lane's run: nbw batches of eight warm-up rows (no bitmaps), then - the state at that point noted in the record - the kZpSub own
// rows, whose bitmaps and end state go to the record.  off = byte offset of the first row; the loads run kZpAhead batches ahead and
// stop at the last own batch (that batch again: no row behind the sub-segment is read).
// keep: the halves (0xffff each) whose record stands and is not to be touched (a lane runs again for its other half).  z: the end state on return.
template <int NT> __device__ __forceinline__ void zp_lane(ZpState &z, gptr8 base, u32 off, int nbw, int ntrks, u32 Pm1, u32 mP1, u32 (*rec)[kZpThreads], int lane, u32 keep = 0) {
   auto put = [&](int f, u32 val) { rec[f][lane] = keep ? ((val & ~keep) | (rec[f][lane] & keep)) : val; };
   const u32 stride = 2u * (u32)(NT ? NT : ntrks);
   const u32 off_last = off + (u32)(nbw * 8 + kZpSub - 8) * stride;
   u32 nx[kZpAhead][8];
   #pragma unroll
   for (int a = 0; a < kZpAhead; ++a) {
      #pragma unroll
      for (int k = 0; k < 8; ++k) nx[a][k] = zp_load(base + off, (u32)k * stride);
      off = min(off + 8u * stride, off_last); }
   u32 dummy_e = 0, dummy_a = 0;
   #pragma nounroll
   for (int b = 0; b < nbw; ++b) {
      u32 v8[8];
      #pragma unroll
      for (int k = 0; k < 8; ++k) { v8[k] = nx[0][k]; for (int a = 0; a + 1 < kZpAhead; ++a) nx[a][k] = nx[a + 1][k]; }
      #pragma unroll
      for (int k = 0; k < 8; ++k) nx[kZpAhead - 1][k] = zp_load(base + off, (u32)k * stride);
      off = min(off + 8u * stride, off_last);
      #pragma unroll
      for (int k = 0; k < 8; ++k) zp_step(z, v8[k], Pm1, mP1, 0u, dummy_e, dummy_a); }
   put(kZfTopS, z.top); put(kZfBotS, z.bot); put(kZfPuS, z.pu); put(kZfPdS, z.pd);
   u32 c0 = 0, c1 = 0;
   #pragma nounroll
   for (int blk = 0; blk < kZpSub / 16; ++blk) {
      u32 evm = 0, arm = 0;
      #pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
         u32 v8[8];
         #pragma unroll
         for (int k = 0; k < 8; ++k) { v8[k] = nx[0][k]; for (int a = 0; a + 1 < kZpAhead; ++a) nx[a][k] = nx[a + 1][k]; }
         #pragma unroll
         for (int k = 0; k < 8; ++k) nx[kZpAhead - 1][k] = zp_load(base + off, (u32)k * stride);
         off = min(off + 8u * stride, off_last);
         #pragma unroll
         for (int k = 0; k < 8; ++k) zp_step(z, v8[k], Pm1, mP1, 0x10001u << (hb * 8 + k), evm, arm); }
      put(kZfEvm + blk, evm); put(kZfArm + blk, arm);
      c0 += (u32)__popc(evm & 0xffffu); c1 += (u32)__popc(evm >> 16); }
   put(kZfTopE, z.top); put(kZfBotE, z.bot); put(kZfPuE, z.pu); put(kZfPdE, z.pd); rec[kZfPvE][lane] = z.pv;
   put(kZfCnt, c0 | (c1 << 16)); }

// 64 row bits - rows 64 wd .. 64 wd + 63 - of one half (0 / 1) of a lane's bitmap `f` (kZfEvm / kZfArm)
__device__ __forceinline__ unsigned long long zp_bits(u32 (*rec)[kZpThreads], int f, int lane, int half, int wd) {
   unsigned long long m = 0;
   #pragma unroll
   for (int b = 0; b < 4; ++b) m |= (unsigned long long)((rec[f + 4 * wd + b][lane] >> (16 * half)) & 0xffffu) << (16 * b);
   return m; }
// the last row in front of row `pos` (0 .. kZpSub) whose bit is set in that bitmap; -1: none
__device__ __forceinline__ int zp_last_below(u32 (*rec)[kZpThreads], int f, int lane, int half, int pos) {
   for (int wd = pos >> 6; wd >= 0; --wd) {
      if (wd == kZpSub / 64) continue;
      unsigned long long m = zp_bits(rec, f, lane, half, wd);
      if (wd == (pos >> 6)) m &= (1ull << (pos & 63)) - 1;
      if (m) return wd * 64 + 63 - __clzll((long long)m); }
   return -1; }
// the first row whose bit is set in that bitmap; kZpSub: none
__device__ __forceinline__ int zp_first(u32 (*rec)[kZpThreads], int f, int lane, int half) {
   for (int wd = 0; wd < kZpSub / 64; ++wd) {
      const unsigned long long m = zp_bits(rec, f, lane, half, wd);
      if (m) return wd * 64 + __ffsll((long long)m) - 1; }
   return kZpSub; }
// the extremes as far as they matter (see the head of the file): below the threshold they are all one state
__device__ __forceinline__ u32 zp_canon_top(u32 top, u32 Pm1) { return top & zq_sar15(zq_sub(Pm1, top)); }
__device__ __forceinline__ u32 zp_canon_bot(u32 bot, u32 Pm1) { return bot & zq_sar15(zq_add(bot, Pm1)); }

#ifndef RTFE_ZP_WPS
#define RTFE_ZP_WPS 4      /* waves per SIMD the register allocation is held to (experiments: tools/build_variant.py zpN -DRTFE_ZP_WPS=N) */
#endif
template <int NT> __global__ void __launch_bounds__(kZpThreads, RTFE_ZP_WPS) k_zeros(const DevCfg *__restrict__ cfgp, const int16_t *__restrict__ rows, long long nrows, long long row_base,
                                                                           rtfe_burst *__restrict__ bursts, BurstScratch *__restrict__ scratch,
                                                                           uint32_t *__restrict__ counts, rtfe_event *__restrict__ events) {
   __shared__ DevCfg cfg;
   __shared__ int s_burst;
   __shared__ unsigned int s_flags;
   __shared__ int s_first[RTFE_MAXTRKS + 1];               // per track: the first sub-segment whose join failed (nsub: none)
   __shared__ ZWalker walkers[RTFE_MAXTRKS], wnext[RTFE_MAXTRKS];
   __shared__ u32 rec[kZfN][kZpThreads];                   // the lanes' records, field-major (no bank conflicts)
   for (int i = threadIdx.x; i < (int)(sizeof(DevCfg) / 4); i += blockDim.x) reinterpret_cast<int *>(&cfg)[i] = reinterpret_cast<const int *>(cfgp)[i];
   __syncthreads();
   const int ntrks = NT ? NT : cfg.ntrks;
   const int npair = (ntrks + 1) / 2, nsub_max = kZpThreads / npair;
   const int P = cfg.zc_peak_i;
   const u32 Pm1 = (u32)(P - 1) * 0x10001u, mP1 = ((u32)(1 - P) & 0xffffu) * 0x10001u;
   Ctx cx;
   cx.cfg = &cfg;
   cx.row_base = row_base;
   cx.tile.ldw = cfg.ldw; cx.tile.halo = 0; cx.tile.colof = cfg.trk_to_head; cx.tile.ntrks = ntrks; cx.tile.skew = cfg.skew;
   cx.tile.bits = nullptr; cx.tile.bstride = 0; cx.tile.ldpos = nullptr; cx.tile.ldstride = 0; cx.tile.fd = nullptr;
   cx.heights = nullptr; cx.recs = nullptr; cx.rec_cap = 0; cx.nrec = 0;
   const int T = threadIdx.x;
   const bool is_walker = T < ntrks;
   // thread -> (sub-segment j, column pair p), the pairs of a sub-segment side by side: neighbouring lanes read one row's bytes.  An odd
   // track count's last pair is the last two columns again, its low half idle (no byte outside the row is read)
   const int j = T / npair, p = T - j * npair;
   const bool odd_last = (ntrks & 1) && p == npair - 1;
   const int col1 = odd_last ? ntrks - 1 : 2 * p + 1, col0 = col1 - 1;
   const bool live0 = !odd_last;
   const int trk0 = cfg.head_to_trk[col0], trk1 = cfg.head_to_trk[col1];
   const u32 stride = 2u * (u32)ntrks;
   unsigned long long *dbgp = cfg.debug == 1 ? scratch->dbg2 : (unsigned long long *)nullptr;
   for (;;) {
      if (T == 0) { s_burst = atomicAdd(&scratch->queue, 1); s_flags = 0; }
      __syncthreads();
      // the queue is gone through three times: the long bursts first, the short ones last (a burst is one workgroup's from start to end: the
      // launch ends when its last burst does, and that one should not be a long burst started late)
      const int nq = scratch->nbursts;
      if (s_burst >= 3 * nq) break;
      const int pass = s_burst / (nq > 0 ? nq : 1), b = s_burst - pass * nq;
      const int nb = scratch->nbursts_total;
      const rtfe_burst B = bursts[b];
      const bool exact = B.flags & RTFE_F_EXACT_START;
      cx.events = events + B.event_base;
      cx.cap = B.event_cap;
      // any restart inside the zone is equivalent for this detector: the zone's last kMarginRows rows
      long long reset = B.reset_sample;
      unsigned int bflags = B.flags;
      if (!exact) {
         reset = B.zone_end - kMarginRows;
         if (B.zone_end - B.zone_first < kMarginRows + 64) bflags |= RTFE_F_UNSAFE; }
      long long stop = nrows, zq = nrows;                            // zq: where the next dead-quiet zone begins
      if (b + 1 < nb) {
         const rtfe_burst NB = bursts[b + 1];
         zq = NB.zone_first;
         stop = NB.zone_end - kMarginRows;
         if (cfg.tail_rows > 0 && NB.zone_first + cfg.tail_rows < stop) stop = NB.zone_first + cfg.tail_rows; }
      if (stop > nrows) stop = nrows;
      const long long len = stop - reset;
      if ((len >= kZpLong ? 0 : (len >= kZpLong / 4 ? 1 : 2)) != pass) { __syncthreads(); continue; }
      cx.tile.reset = reset;
      // ---- head: the staggered start of the tracks, row by row ----
      long long c0 = reset + kZpHead < stop ? reset + kZpHead : stop;
      if (is_walker) {
         ZWalker w = {};
         w.start = reset + T; w.next = reset; w.z_ttop_row = 0; w.z_tbot_row = 0;
         cx.tile.x = const_cast<int16_t *>(rows) + reset * ntrks; cx.tile.row0 = reset; cx.tile.nrows = (int)(c0 - reset);
         walk_zeros(w, cx, T, c0);
         w.next = c0;
         walkers[T] = w; }
      __syncthreads();
      // ---- chunks of ns sub-segments ----
      while (c0 + kZpSub <= stop) {
         long long k0 = 0, k1 = 0, k2 = 0;
         if (dbgp) k0 = clock64();
         const long long left = (stop - c0) / kZpSub;
         const int ns = left < nsub_max ? (int)left : nsub_max;
         const bool mine = j < ns;
         const gptr8 base = (gptr8)(rows + (c0 - 1) * ntrks) + 2 * col0;             // row c0 - 1, this lane's column pair
         if (T <= ntrks) s_first[T] = ns;
         if (mine) {
            // sub-segment 0 continues the walkers.  The others start zc_warm rows early: from a fresh state inside the block (a live signal
            // replaces every bit of it within the warm-up rows), from the walkers' state where the next dead-quiet zone has begun (nothing
            // is confirmed any more there: the last extremes and a pending crossing linger through the gap, and that is what the walkers hold
            // once the block has ended)
            const ZWalker &w0 = walkers[trk0], &w1 = walkers[trk1];
            const bool lingering = j == 0 || c0 + (long long)j * kZpSub - cfg.zc_warm >= zq;
            ZpState z;
            z.top = lingering ? zp_mk16(w0.z_top, w1.z_top) : 0u; z.bot = lingering ? zp_mk16(w0.z_bot, w1.z_bot) : 0u; z.pv = zp_mk16(w0.z_prev, w1.z_prev);
            z.pu = lingering ? zp_mk16(w0.z_up_pending ? -32768 : 0, w1.z_up_pending ? -32768 : 0) : 0u;
            z.pd = lingering ? zp_mk16(w0.z_dn_pending ? -32768 : 0, w1.z_dn_pending ? -32768 : 0) : 0u;
            const int warm = j ? cfg.zc_warm : 0;
            const u32 off = (u32)(j * kZpSub - warm) * stride;                         // row c0 + 64 j - warm - 1: it seeds "previous sample"
            if (j) z.pv = zp_load(base, off);
            z.pnv = zq_subs(0u, z.pv);
            zp_lane<NT>(z, base, off + stride, warm / 8, ntrks, Pm1, mP1, rec, T); }
         __syncthreads();
         if (dbgp) k1 = clock64();
         // ---- joins: every lane checks its own.  Where joins fail, the FIRST failing sub-segment of a track runs again from its predecessor's
         // end - the true state, by induction from sub-segment 0 - and the check moves on; the failing ones behind it run again as well,
         // assuming that same state zc_warm rows early (a block's end is where joins fail: what lingers there lingers for all of them).
         for (int round = 0; round <= ns; ++round) {
            bool bad0 = false, bad1 = false;
            if (mine && j > 0) {
               const int q = T - npair;
               const u32 dt = zp_canon_top(rec[kZfTopS][T], Pm1) ^ zp_canon_top(rec[kZfTopE][q], Pm1);
               const u32 db = zp_canon_bot(rec[kZfBotS][T], Pm1) ^ zp_canon_bot(rec[kZfBotE][q], Pm1);
               const u32 df = ((rec[kZfPuS][T] ^ rec[kZfPuE][q]) | (rec[kZfPdS][T] ^ rec[kZfPdE][q])) & 0x80008000u;
               const u32 d = dt | db | df;
               bad0 = live0 && (d & 0xffffu); bad1 = (d >> 16) != 0;
               if (bad0) atomicMin(&s_first[trk0], j);
               if (bad1) atomicMin(&s_first[trk1], j); }
            __syncthreads();
            bool any = false;
            for (int t = 0; t < ntrks; ++t) any = any || s_first[t] < ns;
            if (!any) break;
            if (dbgp && T == 0) atomicAdd(&dbgp[4], 1ull);
            const int fb0 = s_first[trk0], fb1 = s_first[trk1];
            // who runs again: the failing halves, and - behind a track's first failure - every half whose rows lie in the next dead-quiet zone
            // (those assumed one and the same lingering state and so agree with each other, right or wrong)
            const bool in_gap = mine && j > 0 && c0 + (long long)j * kZpSub - cfg.zc_warm >= zq;
            const bool run0 = bad0 || (live0 && in_gap && fb0 < j), run1 = bad1 || (in_gap && fb1 < j);
            __syncthreads();
            if (T <= ntrks) s_first[T] = ns;
            if (run0 || run1) {
               if (dbgp) atomicAdd(&dbgp[3], 1ull);
               // a half that is its track's first failure must start exactly where its predecessor ended: no warm-up rows for the lane then
               const bool exact = (bad0 && fb0 == j) || (bad1 && fb1 == j);
               const int g0 = (exact || !run0 ? j : fb0) - 1, g1 = (exact || !run1 ? j : fb1) - 1;          // whose end state each half assumes
               const int l0 = g0 * npair + p, l1 = g1 * npair + p;
               ZpState z;
               z.top = (rec[kZfTopE][l0] & 0xffffu) | (rec[kZfTopE][l1] & 0xffff0000u); z.bot = (rec[kZfBotE][l0] & 0xffffu) | (rec[kZfBotE][l1] & 0xffff0000u);
               z.pu = (rec[kZfPuE][l0] & 0xffffu) | (rec[kZfPuE][l1] & 0xffff0000u);    z.pd = (rec[kZfPdE][l0] & 0xffffu) | (rec[kZfPdE][l1] & 0xffff0000u);
               const int warm = exact ? 0 : cfg.zc_warm;
               const u32 off = (u32)(j * kZpSub - warm) * stride;
               const u32 keep = (run0 ? 0u : 0xffffu) | (run1 ? 0u : 0xffff0000u);
               z.pv = zp_load(base, off); z.pnv = zq_subs(0u, z.pv);
               zp_lane<NT>(z, base, off + stride, warm / 8, ntrks, Pm1, mP1, rec, T, keep); }
            __syncthreads(); }
         if (dbgp) k2 = clock64();
         // ---- events from the bitmaps, in row order; the last sub-segment's lanes move the walkers to the chunk's end ----
         // The events' samples are read again, eight loads in flight at a time, and wait in LDS for the loop that makes the events - in the
         // space of the record fields that are no longer needed (start and end states: the own ones are in registers by now).
         const u32 e_top = rec[kZfTopE][T], e_bot = rec[kZfBotE][T], e_pu = rec[kZfPuE][T], e_pd = rec[kZfPdE][T], e_pv = rec[kZfPvE][T];
         __syncthreads();
         unsigned short *const stg = reinterpret_cast<unsigned short *>(&rec[0][0]);     // [kZpStage][kZpThreads]
         if (mine) {
            u32 before = 0;                                              // events of the sub-segments in front, both halves (a chunk holds fewer than 2^16 per track)
            for (int k = 0; k < j; ++k) before += rec[kZfCnt][k * npair + p];
            #pragma nounroll
            for (int h = 0; h < 2; ++h) {
               if (h == 0 && !live0) continue;
               const int trk = h ? trk1 : trk0, col = h ? col1 : col0;
               const ZWalker &w = walkers[trk];
               const bool last = j == ns - 1;
               const u32 cnt = (rec[kZfCnt][T] >> (16 * h)) & 0xffffu;
               const bool pend_e = ((e_pu | e_pd) >> (16 * h)) & 0x8000u;
               if (!cnt && !last) continue;
               const long long r0 = c0 + (long long)j * kZpSub;
               const gptr16 colp = (gptr16)rows + r0 * ntrks + col;
               unsigned int idx = w.nevents + ((before >> (16 * h)) & 0xffffu);
               // the crossing that was pending when this sub-segment began - the last armed row of the sub-segments in front, or the walker's -
               // is asked for where an event lies in front of the sub-segment's first armed row, or where the pending crossing leaves it
               const int arm_first = zp_first(rec, kZfArm, T, h), ev_first = zp_first(rec, kZfEvm, T, h), arm_end = zp_last_below(rec, kZfArm, T, h, kZpSub);
               unsigned int cin = 0;                                      // (its row relative to r0, modulo 2^32: only differences of rows are used)
               long long cross_in = -1;
               if ((cnt && ev_first <= arm_first) || (last && pend_e && arm_end < 0)) {
                  int k = j - 1, a = -1;
                  for (; k >= 0 && a < 0; --k) a = zp_last_below(rec, kZfArm, k * npair + p, h, kZpSub);
                  cross_in = a >= 0 ? c0 + (long long)(k + 1) * kZpSub + a : (w.z_up_pending ? w.z_ttop_row : w.z_tbot_row);
                  cin = (unsigned int)(cross_in - r0); }
               rtfe_event *const evp = cx.events + (size_t)trk * cx.cap;
               const unsigned int s_base = (unsigned int)(r0 - reset);
               const float mv = cfg.maxvolts;
               // two walks over the bitmap, one staging the samples (up to kZpStage at a time), one making the events from them
               int swd = 0, ewd = 0, arm_last = -1;
               unsigned long long sev = zp_bits(rec, kZfEvm, T, h, 0), eev = sev, ear = zp_bits(rec, kZfArm, T, h, 0);
               bool ovf = false;
               for (unsigned int left = cnt; left;) {
                  int nst = 0;
                  while (nst <= kZpStage - 8 && (sev || swd + 1 < kZpSub / 64)) {
                     if (!sev) { ++swd; sev = zp_bits(rec, kZfEvm, T, h, swd); continue; }
                     int s8[8], v8[8];
                     #pragma unroll
                     for (int u = 0; u < 8; ++u) { s8[u] = sev ? __ffsll((long long)sev) - 1 : -1; sev &= sev - 1; }
                     #pragma unroll
                     for (int u = 0; u < 8; ++u) v8[u] = colp[(swd * 64 + (s8[u] < 0 ? s8[0] : s8[u])) * ntrks];
                     #pragma unroll
                     for (int u = 0; u < 8; ++u) if (s8[u] >= 0) { stg[nst * kZpThreads + T] = (unsigned short)v8[u]; ++nst; } }
                  #pragma nounroll
                  for (int i = 0; i < nst; ++i) {
                     while (!eev) { if (ear) arm_last = ewd * 64 + 63 - __clzll((long long)ear); ++ewd; eev = zp_bits(rec, kZfEvm, T, h, ewd); ear = zp_bits(rec, kZfArm, T, h, ewd); }
                     const int s = __ffsll((long long)eev) - 1;
                     eev &= eev - 1;
                     const int sr = ewd * 64 + s;
                     const int v = (int)(short)stg[i * kZpThreads + T];
                     const unsigned long long below = ear & ((1ull << s) - 1);
                     const int a = below ? ewd * 64 + 63 - __clzll((long long)below) : arm_last;
                     const unsigned int delay = (unsigned int)sr - (a >= 0 ? (unsigned int)a : cin);
                     rtfe_event e;
                     e.sample = s_base + (unsigned int)sr;
                     e.v_peak = volt(v, mv);
                     e.agc_gain = __uint_as_float(delay);
                     e.trk = (uint8_t)trk; e.flags = (uint8_t)(v > 0 ? 0 : 1); e.left_distance = (uint8_t)(delay < 255 ? delay : 255); e.parmset = 0;
                     if (idx < cx.cap) evp[idx] = e;
                     else ovf = true;
                     ++idx; }
                  left -= (unsigned int)nst; }
               if (ovf) atomicOr(&s_flags, (unsigned int)RTFE_F_EVENT_OVERFLOW);
               if (last) {
                  ZWalker x = w;
                  const u32 sh = 16 * h;
                  x.z_prev = (int)(short)(e_pv >> sh); x.z_top = (int)(short)(e_top >> sh); x.z_bot = (int)(short)(e_bot >> sh);
                  x.z_up_pending = ((e_pu >> sh) & 0x8000u) != 0; x.z_dn_pending = ((e_pd >> sh) & 0x8000u) != 0;
                  if (pend_e) {
                     const long long cr = arm_end >= 0 ? r0 + arm_end : cross_in;
                     if (x.z_up_pending) x.z_ttop_row = cr;
                     if (x.z_dn_pending) x.z_tbot_row = cr; }
                  x.nevents = idx; x.next = c0 + (long long)ns * kZpSub;
                  wnext[trk] = x; } } }
         __syncthreads();
         if (is_walker) walkers[T] = wnext[T];
         c0 += (long long)ns * kZpSub;
         __syncthreads();
         if (dbgp && T == 0) { const long long k3 = clock64(); atomicAdd(&dbgp[0], (unsigned long long)(k1 - k0)); atomicAdd(&dbgp[1], (unsigned long long)(k2 - k1)); atomicAdd(&dbgp[2], (unsigned long long)(k3 - k2)); atomicAdd(&dbgp[6], 1ull); } }
      // ---- tail: the rows left, row by row ----
      if (is_walker && c0 < stop) {
         ZWalker w = walkers[T];
         cx.tile.x = const_cast<int16_t *>(rows) + c0 * ntrks; cx.tile.row0 = c0; cx.tile.nrows = (int)(stop - c0);
         walk_zeros(w, cx, T, stop);
         walkers[T] = w; }
      __syncthreads();
      // ---- publish (as k_decode does for this detector) ----
      if (is_walker) {
         const ZWalker &w = walkers[T];
         unsigned int wf = w.flags;
         // history a restart would not have
         if (w.z_up_pending || w.z_dn_pending || w.z_top >= cfg.zc_peak_i || w.z_bot <= -cfg.zc_peak_i) wf |= RTFE_F_STATE_AT_END;
         counts[((size_t)b * cfg.nparm + 0) * ntrks + T] = w.nevents < cx.cap ? w.nevents : cx.cap;
         if (wf) atomicOr(&s_flags, wf); }
      for (int i = T; i < (cfg.nparm - 1) * ntrks; i += blockDim.x) counts[((size_t)b * cfg.nparm + 1) * ntrks + i] = 0;      // (the detector does not depend on the parameter set: set 0 only)
      __syncthreads();
      if (T == 0) {
         bursts[b].reset_sample = reset;
         bursts[b].safe_last = (bflags & RTFE_F_UNSAFE) ? -1 : (!exact ? B.zone_end - ntrks - 2 : reset);
         bursts[b].end_sample = stop;
         bursts[b].flags = bflags | s_flags; }
      __syncthreads(); } }

}  // namespace rtfe
