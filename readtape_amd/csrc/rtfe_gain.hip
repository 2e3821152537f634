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
   __shared__ __attribute__((aligned(16))) int16_t s_z[(kMarginRows + 2 * 50 + 2) * RTFE_MAXTRKS + 8];
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
         // the rows the walk below can read - the zone's last kMarginRows and up to 50 + 50 (window, skew) in front - through LDS: the
         // walk is a chain of dependent reads, and there is nothing else in the wave to hide HBM's latency behind
         const long long r0 = z0 - 2 * 50 - 2 > 0 ? z0 - 2 * 50 - 2 : 0;
         const int nel = (int)(B.zone_end - r0) * ntrks;
         // (8 bytes a load, from the 8-byte boundary in front of the first row - the tape is 16-byte aligned -: sixteen in flight per lane
         //  are a 9-track zone tail in one round trip; s_z mirrors HBM from that boundary on, `sh` elements in front of the first row)
         const long long base_e = r0 * ntrks, a0 = base_e & ~3ll, tot_e = nrows * ntrks;
         const int sh = (int)(base_e - a0), nq = (nel + sh + 3) >> 2;
         __syncthreads();
         for (int q0 = 0; q0 < nq; q0 += 64 * 16) {
            uint2 t[16];
            #pragma unroll
            for (int k = 0; k < 16; ++k) {
               const int q = q0 + k * 64 + (int)threadIdx.x;
               t[k] = make_uint2(0, 0);
               if (q < nq) {
                  const long long e = a0 + 4ll * q;
                  if (e + 4 <= tot_e) t[k] = *reinterpret_cast<const uint2 *>(rows + e);
                  else {                                                 // (the tape's last elements: no byte behind it is read)
                     uint16_t x[4] = {0, 0, 0, 0};
                     for (int j = 0; j < 4; ++j) if (e + j < tot_e) x[j] = (uint16_t)rows[e + j];
                     t[k] = make_uint2((uint32_t)x[0] | ((uint32_t)x[1] << 16), (uint32_t)x[2] | ((uint32_t)x[3] << 16)); } } }
            #pragma unroll
            for (int k = 0; k < 16; ++k) { const int q = q0 + k * 64 + (int)threadIdx.x; if (q < nq) reinterpret_cast<uint2 *>(s_z)[q] = t[k]; } }
         __syncthreads();
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
               if (s < r0) break;                                                 // (cannot happen: W, skew <= 50)
               const int v = sgn * (int)s_z[(s - r0) * ntrks + col + sh];
               bool dom = true;
               for (int k = 1; k <= W; ++k) if (sgn * (int)s_z[(s + k - r0) * ntrks + col + sh] > v) { dom = false; break; }      // (incl. the entering sample: src/decoder.c:763-767)
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
// k_pscan / k_prep: k_sift's lists (one fixed slot per tile and head) -> ONE contiguous stream of 16-byte records per (screen, head),
// in row order, with everything a chain's lane would otherwise recompute per record on its critical path: the owner's absolute
// row, volt() of its value, where its margin entries are.  k_pscan: the streams' tile offsets (a prefix sum per stream over the
// tile directory); k_prep: a wave per list copies its records over.
// ------------------------------------------------------------------------------------------------
// flags in CRec::w0 bits 0-10 (the tile-relative row of PeakRec::w0 is replaced by the absolute CRec::pos)
// (kCrBad / kCrClear / kCrWeak and the helpers that read them: rtfe_sift.hip, beside the records' layout - k_sift_hard makes kCrWeak for its records)
struct CRec { uint32_t pos, w0, w1; float volt; };

// k_pscan1: a workgroup per chunk of 1024 tiles, a thread per tile: per stream the prefix within the chunk and the chunk's total;
// k_pscan2: the chunks' offsets (one workgroup).  A stream's position of tile t = tstart[t][stream] + coff[t >> 10][stream].
__global__ void __launch_bounds__(1024) k_pscan1(const PeakDir *__restrict__ dir, const int *__restrict__ extra, int ntiles, int nlists, uint32_t *__restrict__ tstart, uint32_t *__restrict__ ctotc) {
   __shared__ int lds[32];
   const int t = blockIdx.x * 1024 + threadIdx.x;
   for (int li = 0; li < nlists; ++li) {
      int n = 0;
      if (t < ntiles) { n = dir[(size_t)t * nlists + li].nrec; n = n == 0xffff ? 1 : n + extra[(size_t)t * nlists + li]; }      // (a deferred candidate stands for 0..4 records)
      int total;
      const int off = block_excl_scan_1024(n, lds, &total);
      if (t < ntiles) tstart[(size_t)t * nlists + li] = (uint32_t)off;
      if (threadIdx.x == 0) ctotc[(size_t)blockIdx.x * nlists + li] = (uint32_t)total; } }
__global__ void __launch_bounds__(1024) k_pscan2(int nchunks, int nlists, const uint32_t *__restrict__ ctotc, uint32_t *__restrict__ coff, uint32_t *__restrict__ ctot) {
   __shared__ int lds[32];
   for (int li = 0; li < nlists; ++li) {
      int run = 0;
      for (int c0 = 0; c0 < nchunks; c0 += 1024) {
         const int c = c0 + threadIdx.x;
         const int n = c < nchunks ? (int)ctotc[(size_t)c * nlists + li] : 0;
         int total;
         const int off = block_excl_scan_1024(n, lds, &total);
         if (c < nchunks) coff[(size_t)c * nlists + li] = (uint32_t)(run + off);
         run += total; }
      if (threadIdx.x == 0) ctot[li] = (uint32_t)run; } }
__device__ __forceinline__ long long stream_pos(const uint32_t *tstart, const uint32_t *coff, int nlists, long long tile, int li) {
   return (long long)tstart[(size_t)tile * nlists + li] + (long long)coff[(size_t)(tile >> 10) * nlists + li]; }

// inclusive prefix sum over each half of the wave (lanes 0-31, 32-63) on its own: the wave scan without its last step
__device__ __forceinline__ int half_incl_scan(int v, int hl) {
#ifdef RTFE_CPU_EMUL
   for (int s2 = 1; s2 < 32; s2 <<= 1) { const int y = __shfl_up(v, s2); if (hl >= s2) v += y; }
   return v;
#else
   (void)hl;
   v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
   v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
   v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
   v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
   v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
   return v;
#endif
}

// k_prep: the tiles' lists -> the streams.  Half a wave per list (a clean NRZI tile holds ~21 records per head), the next list's
// directory entry and records in flight while this one is written (a list is two dependent HBM round trips otherwise, and there is
// little else to hide them).  Per record: its absolute row, its volts, its margin block, kCrWeak, and kCrClear where the record's successor in the stream (the
// list's next entry; the first of the next tile's list) settles it - the rest on a work list for k_clear (below).
#ifdef RTFE_CPU_EMUL
__device__ __forceinline__ int rtfe_uniform(int v) { return v; }
#else
__device__ __forceinline__ int rtfe_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }      // a value every lane of the wave holds: into a scalar register
#endif
constexpr int kPrepWorkChunk = 256;      // places of the work list a wave of k_prep takes at a time (a round puts up to 64 x 4 records on it)
struct PrepArgs { int nlists, ntrks, hcap; float mv; };      // (by value: a wave's first loads do not wait for a read of the configuration block)
__global__ void __launch_bounds__(256) k_prep(const PrepArgs pa, const PeakDir *__restrict__ dir, const unsigned char *__restrict__ pool,
                                              const unsigned char *__restrict__ ovf, const uint32_t *__restrict__ tstart, const uint32_t *__restrict__ coff,
                                              const uint32_t *__restrict__ ctot, long long ntiles, long long ccap, CRec *__restrict__ crec, uint2 *__restrict__ cmar,
                                              unsigned long long *__restrict__ work, long long work_cap, int *__restrict__ work_count) {
   const int nlists = pa.nlists, hcap = pa.hcap;
   const float mv = pa.mv;
   const int lane = threadIdx.x & 63, hl = lane & 31, hbase = lane & 32;
   const long long nall = ntiles * nlists;
   const long long stride = (long long)gridDim.x * 8;                     // lists per sweep: four waves, two lists each
   struct Pre { PeakDir d, dn; uint2 r, r1, rn, m; uint32_t ts, co, ct; };
   auto fetch = [&](long long l, long long tl, int s2) -> Pre {      // (everything the list's step reads, bar a deferred candidate's records: no load inside the step to wait for)
      Pre p; p.d.nrec = 0; p.d.nent = 0; p.dn = p.d; p.r = make_uint2(0, 0); p.r1 = p.r; p.rn = p.r; p.m = p.r; p.ts = 0; p.co = 0; p.ct = 0;
      if (l < nall) {
         const unsigned char *slot = pool + (size_t)l * hcap;
         p.d = dir[l];
         p.ts = tstart[(size_t)tl * nlists + s2]; p.co = coff[(size_t)(tl >> 10) * nlists + s2]; p.ct = ctot[s2];
         {  const uint4 r4 = *reinterpret_cast<const uint4 *>(slot + min(16 * hl, hcap - 16));      // (16 bytes: the record, its margin block behind it)
            p.r = make_uint2(r4.x, r4.y); p.m = make_uint2(r4.z, r4.w); }
         p.r1 = *reinterpret_cast<const uint2 *>(slot + min(16 * (hl + 1), hcap - 16));
         if (l + nlists < nall) { p.dn = dir[l + nlists]; p.rn = *reinterpret_cast<const uint2 *>(slot + (size_t)nlists * hcap); } }
      return p; };
   const unsigned wcap = (unsigned)work_cap;                             // (< 2^31; a count that ran over reads as a place behind it)
   int wk_next = 0, wk_end = 0;                                          // the wave's places on the work list (k_clear); wave-uniform (work_cap < 2^31: the host)
   long long li = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
   // (tile, stream) of the list, stepped along with li: no division in the loop
   const long long dq = stride / nlists;
   const int dr = (int)(stride - dq * nlists);
   long long tile = li / nlists;
   int sl = (int)(li - tile * nlists);
   // (no software prefetch of the next list: gfx9 counts loads and stores in ONE counter, the stores of a list's records are conditional, and
   //  hipcc then waits for everything in flight before the first use of a prefetched value - measured: the loads of list i + 1 were waited for
   //  right behind their issue.  Eight waves per SIMD hide the round trip instead: the registers the second set of values took are free.)
   for (; __ballot(li < nall) != 0ull; li += stride, tile += dq, sl += dr) {
      if (sl >= nlists) { sl -= nlists; ++tile; }
      const Pre cu = fetch(li, tile, sl);
      const PeakDir d = cu.d;
      const bool on = li < nall && d.nrec != 0;
      const bool built = on && (long long)cu.ct <= ccap;                  // a stream that outgrew its capacity is not built: its chains give up (k_gain)
      long long base = built ? (long long)sl * ccap + (long long)cu.ts + (long long)cu.co : 0;
      if (built && d.nrec == 0xffffu) {                                  // a list that did not fit: one marker at the tile's first row
         if (hl == 0) { CRec m; m.pos = (uint32_t)(tile * kSfTile); m.w0 = kCrBad | (1u << 12); m.w1 = 0xffff8000u; m.volt = 0; crec[base] = m; cmar[base] = make_uint2(0, 0); } }
      const int nrec = (built && d.nrec != 0xffffu) ? (int)d.nrec : 0;
      const unsigned char *slot = pool + (size_t)(on ? li : 0) * hcap;
      const long long pos0 = tile * kSfTile - kSfPosBias;
      // kCrClear: nothing behind the record in its stream has a row at or before X, the last row the record can fire at.  A stream is in the order of its
      // CANDIDATES and a candidate's records have their rows behind the candidate, so the look ahead from the record ends - yes - at the first record whose candidate (its owner at
      // the least) lies at or behind X; every record it passes on the way must begin behind X.  The successor settles it nine times in ten: a plain entry of this list - or the first
      // of the next tile's - straight from the registers.  What it does not settle, and the records of deferred candidates, go on a work list: k_clear looks ahead in the finished
      // stream (the loop in here, beside the records' registers, cost k_prep two waves a SIMD: 0.33 -> 0.43 ms on C2).
      // (Rounds 3 - 6a asked this of the record's successor alone: a bottom whose successor - the next bottom - began behind its window was marked although the top between those
      //  two, the record after next, fired at the very row the bottom did, and tops go first: tools/fuzz_shapes.py.)
      const int nrec_l = (on && d.nrec != 0xffffu) ? (int)d.nrec : 0;
      const int p32 = (int)pos0;                                           // (rows in 32 bits: rtfe_scan hands this path fragments of less than 2^31 rows)
      int rounds = (nrec + 31) >> 5;
      {  const int other = __shfl(rounds, lane ^ 32); if (other > rounds) rounds = other; }      // (both halves run the scans of every round)
      for (int rd = 0; rd < rounds; ++rd) {
         const int k = rd * 32 + hl;
         const bool have = k < nrec;
         uint2 q = cu.r, q1 = cu.r1;
         if (rd > 0 && have) { q = *reinterpret_cast<const uint2 *>(slot + 16 * k); if (k + 1 < nrec) q1 = *reinterpret_cast<const uint2 *>(slot + 16 * (k + 1)); }
         const uint32_t w0 = have ? q.x : 0u, w1 = have ? q.y : 0u;
         const bool deferred = have && w1 == 0xffff8001u;                  // its records are in overflow slot w0 (k_sift_hard): they take its place
         const unsigned char *os = ovf + (size_t)(deferred ? w0 : 0u) * kSfOvfBytes;
         const int cnt = deferred ? *reinterpret_cast<const int *>(os) : (have ? 1 : 0);
         const int ic = half_incl_scan(cnt, hl);
         const long long o = base + ic - cnt;
         uint32_t todo = 0;                                                // bit j: record j of this entry goes on the work list
         if (deferred) {
            // (a stale minimum's record is owned by a sample in front of its candidate: further than two rows in front, the countdown it leaves when it fires ends before
            //  the rows of a record the chain has passed over do - such a record is never marked: it is the general step's, which looks back - k_gain)
            const int qrel = kSfPosBias + *reinterpret_cast<const int *>(os + 4);
            #pragma nounroll
            for (int j = 0; j < cnt; ++j) {
               const uint2 e = *reinterpret_cast<const uint2 *>(os + 8 + 16 * j);
               const uint2 em = *reinterpret_cast<const uint2 *>(os + 8 + 16 * j + 8);
               const uint32_t ew = *reinterpret_cast<const uint32_t *>(os + 72 + 4 * j);      // (kCrWeak and its bits: k_sift_hard made them beside the margins - made here, they cost k_prep 22 registers)
               CRec c; c.pos = (uint32_t)(pos0 + (long long)(e.x & 0x7ffu)); c.w0 = (e.x & ~0x7ffu) | ew; c.w1 = e.y; c.volt = volt((int)(int16_t)(e.y & 0xffffu), mv);
               if (crec_can_clear(e.x, e.y, ew) && qrel <= (int)(e.x & 0x7ffu) + 2) todo |= 1u << j;
               crec[o + j] = c;
               cmar[o + j] = em; } }
         else if (have) {
            uint2 mk2 = cu.m;                                              // (its margin block: with the record in the first round)
            if (rd > 0) mk2 = *reinterpret_cast<const uint2 *>(slot + 16 * k + 8);
            const uint32_t wk = crec_weak_bits(w0, w1, mk2);
            CRec c; c.pos = (uint32_t)(pos0 + (long long)(w0 & 0x7ffu)); c.w0 = (w0 & ~0x7ffu) | wk; c.w1 = w1; c.volt = volt((int)(int16_t)(w1 & 0xffffu), mv);
            if (crec_can_clear(w0, w1, wk)) {
               const int X = p32 + (int)(w0 & 0x7ffu) + (int)((w0 >> 12) & 63u) + (int)((w0 >> 18) & 15u) - (wk ? 1 : 0);      // the first sure row; kCrWeak: its last row
               // its successor: the list's next entry, the next tile's first, none
               uint2 sq = q1; int sp0 = p32; int kind = 0;                 // 0: an entry, 1: nothing behind it that could matter, 2: cannot tell from here
               if (k + 1 >= nrec_l) {
                  const int nn = (int)cu.dn.nrec;
                  if (tile + 1 >= ntiles || nn == 0) kind = 1;              // (the stream ends; an empty list: two lists on, and a tile is longer than a window)
                  else if (nn == 0xffff) kind = p32 + kSfPosBias + kSfTile >= X ? 1 : 3;      // a list that is not there: whatever its tile holds has its rows behind the tile's first
                  else { sq = cu.rn; sp0 = p32 + kSfTile; } }
               if (kind == 0) {
                  if (sq.y == 0xffff8001u) kind = 2;
                  else {
                     const int pj = sp0 + (int)(sq.x & 0x7ffu);
                     if (pj + (int)((sq.x >> 12) & 63u) <= X) kind = 3;     // (3: not clear)
                     else kind = pj >= X ? 1 : 2; } }
               if (kind == 1) c.w0 |= kCrClear;
               if (kind == 2) todo = 1u; }
            crec[o] = c;
            cmar[o] = mk2; }
         // the work list: a wave takes its places a chunk at a time (one atomic per record and wave round on ONE address: a 60 mV-noise tape's 1.2 M deferred candidates made
         // k_prep 5 ms longer - as k_sift_s's list of deferred candidates had, before it took chunks); what it leaves of a chunk it marks empty
         if (__ballot(todo != 0u) != 0ull) {
            const int nt = __popc(todo);
            const int incl = wave_incl_scan(nt, lane);
            const int total = wave_last(incl);
            if (wk_next + total > wk_end) {
               for (int x = wk_next + lane; x < wk_end; x += 64) if ((unsigned)x < wcap) work[x] = ~0ull;
               int nb = 0;
               if (lane == 0) nb = atomicAdd(work_count, kPrepWorkChunk);
               wk_next = rtfe_uniform(__shfl(nb, 0)); wk_end = wk_next + kPrepWorkChunk; }
            int wbase = wk_next + incl - nt;
            wk_next = rtfe_uniform(wk_next + total);
            for (int j = 0; j < 4; ++j) if (todo & (1u << j)) { if ((unsigned)wbase < wcap) work[wbase] = (unsigned long long)(o + j); ++wbase; } }
         base += __shfl(ic, hbase + 31); } }
   for (int x = wk_next + lane; x < wk_end; x += 64) if ((unsigned)x < wcap) work[x] = ~0ull; }

// k_clear: kCrClear for the records k_prep could not settle from a record's successor - a thread per work list entry looks ahead in the finished stream.  Every record behind
// it must begin behind X until one's owner - its candidate lies at or behind its owner, the later candidates behind that - is at or behind X; a marker stands for a tile's
// unknown candidates (all behind the tile's first row).  (The flag bit of a word is written by the one thread that owns the record; the others read the word's other bits.)
__global__ void __launch_bounds__(256) k_clear(const unsigned long long *__restrict__ work, long long work_cap, const int *__restrict__ work_count, const uint32_t *__restrict__ ctot,
                                               long long ccap, CRec *__restrict__ crec) {
   long long n = *work_count;
   if (n > work_cap) n = work_cap;                                        // (what did not fit stays unmarked: the general step's)
   for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
      const unsigned long long gw = work[t];
      if (gw == ~0ull) continue;                                          // (a place nobody took)
      const long long g = (long long)gw;
      const long long sl = g / ccap, i = g - sl * ccap;
      const long long ne = (long long)ctot[sl];
      const CRec *r = crec + (size_t)sl * ccap;
      const uint32_t w0 = r[i].w0;
      const int X = (int)r[i].pos + (int)((w0 >> 12) & 63u) + (int)((w0 >> 18) & 15u) - ((w0 & kCrWeak) ? 1 : 0);      // the last row it can fire at: its first sure row; kCrWeak: its last row
      bool ok = false;
      #pragma nounroll
      for (long long j = i + 1; ; ++j) {
         if (j >= ne) { ok = true; break; }
         if (j > i + 16) break;
         const int pj = (int)r[j].pos;
         const uint32_t wj = r[j].w0;
         if (wj & kCrBad) { ok = pj >= X; break; }
         if (pj + (int)((wj >> 12) & 63u) <= X) break;
         if (pj >= X) { ok = true; break; } }
      if (ok) crec[(size_t)g].w0 = w0 | kCrClear; } }

#ifdef RTFE_CPU_EMUL
// (emulator only, RTFE_PREP_CHECK: what kCrClear promises, checked by a pass over the finished streams - no record behind a marked one has a row at or before the
//  last row the marked one can fire at; RTFE_DUMP_SL / _LO / _HI: the records of a stream between two rows)
__global__ void __launch_bounds__(64) k_prep_check(const DevCfg *__restrict__ cfgp, const uint32_t *__restrict__ ctot, long long ccap, const CRec *__restrict__ crec) {
   const DevCfg &cfg = *cfgp;
   const int nlists = cfg.nscreens * cfg.ntrks;
   if (blockIdx.x != 0 || threadIdx.x != 0) return;
   long long nclear = 0, nall = 0;
   for (int sl = 0; sl < nlists; ++sl) {
      const long long n = (long long)ctot[sl] > ccap ? 0 : (long long)ctot[sl];
      const CRec *r = crec + (size_t)sl * ccap;
      for (long long i = 0; i < n; ++i) {
         const uint32_t w0 = r[i].w0, w1 = r[i].w1;
         const int nsure = (int)((w0 >> 22) & 63u);
         if (getenv("RTFE_DUMP_LO") && sl == atoi(getenv("RTFE_DUMP_SL")) && (long long)r[i].pos >= atoll(getenv("RTFE_DUMP_LO")) && (long long)r[i].pos <= atoll(getenv("RTFE_DUMP_HI")))
            fprintf(stderr, "DUMP sl %d i %lld pos %u %s f +%u nlead %u nsure %u ntail %u val %d w1 %08x flags %x\n", sl, i, r[i].pos, (w0 & 0x800u) ? "bot" : "top", (w0 >> 12) & 63u, (w0 >> 18) & 15u, nsure, (w0 >> 28) & 15u, (int)(int16_t)(w1 & 0xffff), w1, w0 & 0x7ffu);
         ++nall;
         if (!(w0 & kCrClear)) continue;
         ++nclear;
         if ((w0 & kCrBad) || w1 == 0xffff8000u || ((unsigned)(nsure - 1) >= 62u && !(w0 & kCrWeak))) fprintf(stderr, "prep_check: stream %d record %lld pos %u is marked clear and has no sure stretch (w0 %08x w1 %08x)\n", sl, i, r[i].pos, w0, w1);
         const long long X = (long long)r[i].pos + (long long)((w0 >> 12) & 63u) + (long long)((w0 >> 18) & 15u) - ((w0 & kCrWeak) ? 1 : 0);
         for (long long j = i + 1; j < n && j < i + 600; ++j) {
            if ((long long)r[j].pos - 2 * kSfPosBias > X) break;            // (owners lie less than kSfPosBias rows in front of their candidates, rows behind them)
            const long long fj = (long long)r[j].pos + (long long)((r[j].w0 >> 12) & 63u);
            if (fj <= X) fprintf(stderr, "prep_check: stream %d record %lld pos %u marked clear (fires by row %lld), record %lld pos %u begins at row %lld\n", sl, i, r[i].pos, X, j, r[j].pos, fj); } } }
   if (getenv("RTFE_PREP_CHECK") && atoi(getenv("RTFE_PREP_CHECK")) > 1) fprintf(stderr, "prep_check: %lld of %lld records marked clear\n", nclear, nall); }
#endif

// ------------------------------------------------------------------------------------------------
// k_gain
// ------------------------------------------------------------------------------------------------
struct Run {
   long long pos, f;                 // column rows
   int nlead, nsure, ntail, val, dprev, dnext;
   bool top, unknown; };

__device__ __forceinline__ Run run_decode(uint32_t w0, uint32_t w1, long long pos) {
   Run u;
   u.pos = pos;
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
constexpr int kGainChunk = 32;      // records a lane steps through between two general steps (multiple of 4)
// The margin of row f + j of a record: the first kPkMar rows' margins came with the record (eend[-(j + 1)]); for any other row the walker
// makes it from the samples, as k_sift does (pk_margin: the extreme against the nearer window edge, clamped at 0; rows outside the tape
// read as zeros).  Column rows: the detector's row n reads sample n - d of the head's column, and records count in column rows.
struct MarSrc { const int16_t *rows; long long nrows; int ntrks, head, sg, W, nmar; };
__device__ __forceinline__ int mar_sample(const MarSrc &m, long long r) { return (r >= 0 && r < m.nrows) ? m.sg * (int)m.rows[r * m.ntrks + m.head] : 0; }
__device__ __forceinline__ int run_margin(const MarSrc &m, const Run &u, const uint16_t *eend, int j) {
   if (j < m.nmar) return (int)eend[-(j + 1)];
   const long long n = u.f + j;
   const int xl = mar_sample(m, n - m.W + 1), xr = mar_sample(m, n);
   const int mg = u.top ? u.val - max(xl, xr) : min(xl, xr) - u.val;
   return mg < 0 ? 0 : (mg > 65535 ? 65535 : mg); }

// first row >= c (and < limit) at which this run makes the detector fire, or kNoRow; doubt = first row >= c that the record cannot decide.
__device__ __forceinline__ long long run_fire(const Walker &w, const Run &u, const uint16_t *eend, const MarSrc &ms, long long c, long long limit, int W, int sure_i, float mv, long long &doubt) {
   doubt = kNoRow;
   const long long last_row = u.pos + W - 2;                         // the owner is strictly inside the window up to here
   if (last_row < c || u.f >= limit) return kNoRow;
   if (!amp_pass(w, u, mv)) return kNoRow;
   if (u.unknown) { const long long n = max(c, u.f); if (n < u.f + u.nsure && n < limit) doubt = n; return kNoRow; }
   for (int i = 0; i < u.nlead; ++i) {
      const long long n = u.f + i;
      if (n < c) continue;
      if (n >= limit) return kNoRow;
      if (rise_pass(w, u.top, u.val, run_margin(ms, u, eend, i), mv)) return n; }
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
      if (rise_pass(w, u.top, u.val, run_margin(ms, u, eend, u.nlead + u.nsure + i), mv)) return n; }
   return kNoRow; }

// a lane's place in its stream for the general step
struct RecIt {
   long long i;
   uint32_t w0, w1; long long pos;
   const uint16_t *eend;
   bool end, bad; };
struct RecSrc { const CRec *rec; const uint2 *cmar; long long iend; };      // a stream's records, their margin blocks (entry j of record i at ((uint16 *)(cmar + i + 1))[-(j + 1)])
__device__ __forceinline__ void it_land(RecIt &it, const RecSrc &S) {
   if (it.i >= S.iend) { it.end = true; return; }
   const CRec r = S.rec[it.i];
   it.pos = r.pos; it.w0 = r.w0; it.w1 = r.w1;
   if (r.w0 & kCrBad) { it.end = true; it.bad = true; return; }
   it.eend = reinterpret_cast<const uint16_t *>(S.cmar + it.i + 1); }
__device__ __forceinline__ void it_open(RecIt &it, const RecSrc &S, long long i) {
   it.end = false; it.bad = false; it.i = i; it.eend = nullptr; it.w0 = 0; it.w1 = 0; it.pos = 0;
   it_land(it, S); }
__device__ __forceinline__ void it_next(RecIt &it, const RecSrc &S) { ++it.i; it_land(it, S); }

// an event the fast path only noted (k_emit finishes it): its record's place in the stream, and the gain in force
__device__ __forceinline__ rtfe_event note_event(long long i, float g, float h) {
   union { rtfe_event e; uint32_t w[4]; } u;
   u.w[0] = (uint32_t)i; u.w[1] = __float_as_uint(g); u.w[2] = __float_as_uint(h); u.w[3] = 0xffffffffu;
   return u.e; }

constexpr int kGsChunk = 16;
struct GsState { float g, vlt, vlb; int c, rise_hi, min_lo, min_hi; };
struct GsSeg {
   int chain, sidx;                  // k_segplan
   long long first, end;             // its own records [first, end) of the chain's stream
   GsState at_first, at_end;         // k_gain_seg<0>: the state in front of record `first`; behind the last record it got through
   long long stop;                   // the record it stopped at (the general step's business), or `end`
   int cnt;                          // events of its own records
   unsigned int evoff;               // k_gain_join: where its first note goes in the chain's event list
   int stands, pad;
   // what k_emit_seg needs of the chain, with the entry (k_gain writes it at the hand-over): its loads of records and gains start behind ONE read, not behind a chain of three
   unsigned long long ev_index; long long reset; float h; int sl; };
__device__ __forceinline__ bool gs_same(const GsState &a, const GsState &b) {
   return __float_as_uint(a.g) == __float_as_uint(b.g) && __float_as_uint(a.vlt) == __float_as_uint(b.vlt) && __float_as_uint(a.vlb) == __float_as_uint(b.vlb)
          && a.c == b.c && a.rise_hi == b.rise_hi && a.min_lo == b.min_lo && a.min_hi == b.min_hi; }

// A chain between the kernels that walk it: k_gain (mode 0: from the burst's restart row until the baseline is fixed) -> k_gain_s (the
// steady stretch: nothing but the common record) -> k_gain (mode 1: whatever k_gain_s stopped at, to the chain's end).
enum { kChNone = 0, kChSteady = 1, kChGeneral = 2, kChDone = 3 };
// The chains of a scan: (burst, DISTINCT parameter set, track).  Parameter sets that differ only in what the host decoders read are one chain (round 6 on this path; the
// dense path since round 4: DevCfg::uset_*) - the reference's eight NRZI defaults are six (src/parmsets.c:77-88: sets 1 / 3 and 2 / 4 differ in clk_window alone); k_dup_sets
// copies the chain's events into the other sets' regions behind k_publish.  Everything addresses a chain as before, ci = (b * nparm + pidx) * ntrks + trk, pidx = the chain's first set.
struct ChainIx { int b, wi, pidx, trk, ci; };
__device__ __forceinline__ int chain_count(const DevCfg &cfg, int nbursts) { return nbursts * cfg.nuset * cfg.ntrks; }
__device__ __forceinline__ ChainIx chain_ix(const DevCfg &cfg, int cu) {
   const int ntrks = cfg.ntrks, nwu = cfg.nuset * ntrks;
   ChainIx x;
   x.b = cu / nwu;
   const int wu = cu - x.b * nwu, u = wu / ntrks;
   x.trk = wu - u * ntrks; x.pidx = cfg.uset_rep[u]; x.wi = x.pidx * ntrks + x.trk; x.ci = x.b * cfg.nparm * ntrks + x.wi;
   return x; }
struct GsConst {                       // what a segment's lane needs to know about its chain (k_gain, mode 0, at the hand-over)
   unsigned long long ev_index;        // the chain's event list in the event arena
   float h, alpha, kr, km, rg_min, g_min;
   int W, sure_i, limit32, amp_on, sl, pad; };
struct ChainSt { Walker w; float heights[10]; long long i, c, iend; int status, seg0, nseg, pad; unsigned int seg_ev0, seg_ev1; GsConst k; };      // (iend, seg0, nseg: the steady stretch and its segments)

// (Chains a wave: 64, also for the tails.  Their time is the slowest wave's - every general step a lane takes, every record it walks alone is paid by its whole wave - but the
//  kernel's 480 registers leave the chip 1 024 waves at once and a 60 mV-noise tape's 18 k chains are 290: sixteen or eight chains a wave took 2.9 / 3.7 ms instead of 2.2,
//  measured - more waves than slots, and the slowest of sixteen lanes walks 36 chunks where the slowest of 64 walks 48.)
__global__ void __launch_bounds__(64) k_gain(const DevCfg *__restrict__ cfgp, int mode, ChainSt *__restrict__ cst, long long nrows, long long row_base,
                                             const rtfe_burst *__restrict__ bursts, BurstScratch *__restrict__ scratch, BurstCtl *__restrict__ ctl,
                                             uint32_t *__restrict__ counts, rtfe_event *__restrict__ events, float *__restrict__ chain_h,
                                             const CRec *__restrict__ crec, const uint2 *__restrict__ cmar, const uint32_t *__restrict__ tstart, const uint32_t *__restrict__ coff, const uint32_t *__restrict__ ctot,
                                             long long ccap, const unsigned char *__restrict__ pool, long long ntiles, GsSeg *__restrict__ segs, long long seg_cap, const int16_t *__restrict__ rows) {
   __shared__ float s_heights[64 * 10];
   __shared__ uint4 s_notes[kGainChunk][64];                           // the events the fast path notes, until the chunk's end
   __shared__ uint4 s_rec[kGainChunk + 1][64];                         // the lanes' records of the current chunk
   const DevCfg &cfg = *cfgp;
   const int ntrks = cfg.ntrks, nwalk = cfg.nparm * ntrks, nlists = cfg.nscreens * ntrks;
   const int lane = threadIdx.x;
   const float mv = cfg.maxvolts, lsb = cfg.lsb_per_volt;
   const int nchains = chain_count(cfg, scratch->nbursts);
   float *heights = s_heights + lane * 10;
   // (every lane of a wave goes through the same rounds - the wave votes on them - so a lane without a chain walks a finished one)
   for (int cbase = blockIdx.x * 64; cbase < nchains; cbase += gridDim.x * 64) {
      const ChainIx cx_ = chain_ix(cfg, cbase + lane < nchains ? cbase + lane : nchains - 1);
      const int ci = cx_.ci, b = cx_.b, wi = cx_.wi, pidx = cx_.pidx, trk = cx_.trk;
      const bool active = cbase + lane < nchains && ctl[b].status == kBurstReady && (mode == 0 || cst[ci].status == kChGeneral);
      const rtfe_burst B = bursts[b];
      // (the chain's constants by value: a reference into the configuration block would be loaded again - a vector load, the lanes'
      //  parameter sets differ - at every use, and each such load waits for everything in flight)
      const DevParm P = cfg.parm[pidx];
      const DevScreen S = cfg.screen[P.screen];
      const int cmode = cfg.mode, agc_off = cfg.agc_off;
      const int W = P.W, d = cfg.skew[trk], head = cfg.trk_to_head[trk];
      const long long reset = ctl[b].reset;
      const long long stop = chain_stop(cfg, bursts, ctl, b, scratch->nbursts_total, nrows);
      Walker w = {};
      w.agc_gain = 1.0f; w.v_avg_height = 4.0f;
      update_thresholds(w, P, lsb);
      for (int i = 0; i < 10; ++i) heights[i] = 0;
      // column rows: the detector's row n reads sample n - d.  Before c the window is filling on zone samples only.
      long long c = reset + W + max(trk, d) + 1 - d;
      if (mode == 1 && active) {                                          // where k_gain_s stopped
         w = cst[ci].w; c = cst[ci].c;
         for (int i = 0; i < 10; ++i) heights[i] = cst[ci].heights[i];
         update_thresholds(w, P, lsb); }
      const long long limit = stop - d;
      rtfe_event *ev = events + B.event_base + (size_t)(pidx * ntrks + trk) * B.event_cap;
      const unsigned int cap = B.event_cap;
      bool failed = false;
      int why = 0;
      unsigned int n_fast = 0, n_slow = 0;
      // this chain's piece of its head's stream: from the tile that holds row c - W to the tile behind the limit
      const int sl = P.screen * ntrks + head;
      RecSrc src;
      src.rec = crec + (size_t)sl * ccap; src.cmar = cmar + (size_t)sl * ccap;
      MarSrc msrc; msrc.rows = rows; msrc.nrows = nrows; msrc.ntrks = ntrks; msrc.head = head; msrc.sg = cfg.invert ? -1 : 1; msrc.W = W; msrc.nmar = cfg.pk_mar;
      long long i;
      {  long long g0 = (c - W) / kSfTile; if (c - W < 0) g0 = 0; if (g0 >= ntiles) g0 = ntiles - 1;
         long long ge = limit <= 0 ? 0 : (limit + kSfTile - 1) / kSfTile;       // first tile whose candidates all lie at or behind the limit
         i = stream_pos(tstart, coff, nlists, g0, sl);
         if (mode == 1 && active) i = cst[ci].i;
         src.iend = ge < ntiles ? stream_pos(tstart, coff, nlists, ge, sl) : (long long)ctot[sl];
         if ((long long)ctot[sl] > ccap) { failed = true; why = 6; src.iend = i; } }      // (the stream was not built: k_prep)
      const bool lean = cfg.pk_fast && cmode != RTFE_PE;                // (PE decides the end of its preamble from peak TIMES: the general step)
      // steady state = the three-flop alpha filter.  A window AGC over ONE height (src/decoder.c:513-527 with agc_window = 1: the reference's NRZI
      // sets 5 - 8, src/parmsets.c:83-86) is the same arithmetic with alpha = 1: gain = h / lastheight, and 1 x q + 0 x g = q exactly; the ring's one
      // slot is overwritten before it is read, so nothing of it survives a detection
      const bool alpha_agc = !agc_off && (P.agc_window == 0 || P.agc_window == 1);
      const float alpha = P.agc_window == 1 ? 1.0f : P.agc_alpha, beta = 1 - alpha;
      // The stream goes through LDS a chunk of kGainChunk records a lane at a time (+ one: the record behind the chunk's last is what
      // its fast path looks at).  While the lanes step through a chunk - LDS reads only, and the notes they take go to LDS too - the
      // next chunk's 16-byte loads travel into registers; at the chunk's end they go to LDS.  (Loads issued inside the steps would
      // each be waited for in full: hipcc does not keep count across the round's branches.)
      // step(): 0 = one record on, in lock step; 1 = this record needs the general step (the lane waits for the chunk's end);
      // 2 = the chain is done.
      int nbuf = 0;
      auto flush_notes = [&]() {
         #pragma unroll
         for (int j = 0; j < kGainChunk; ++j) if (j < nbuf) reinterpret_cast<uint4 *>(ev)[w.nevents - (unsigned)nbuf + (unsigned)j] = s_notes[j][lane];
         nbuf = 0; };
      // (rows fit 32 bits here: rtfe_scan takes fragments of less than 2^31 rows through this path)
      const int limit32 = limit > 0x7ffffff0ll ? 0x7ffffff0 : (int)limit;
      const bool amp_on = P.min_peak != 0;
      // steady state (NRZI / GCR: peakcount > 15, the baseline fixed; the alpha filter): thresholds straight from 1 / g
      bool steady = false;
      float kr = 0, km = 0, rg_min = 0;                                    // rise / min_peak thresholds in int16 units per unit of 1 / g; below rg_min = 1 / g they come near the screen's
      const float g_min = 0.005f * lsb / 249.0f;                         // below it the half-sample refinement's threshold outgrows the records' neighbour distances (253)
      // from here on the baseline is fixed: the steady path's constants
      auto enter_steady = [&]() {
         steady = true;
         const float hs = w.v_avg_height * 0.25f;
         kr = P.rise * hs * lsb; km = P.min_peak * hs * lsb;
         // thresholds stay clear of the screen's while rise * hs / g >= 1.002 screen_rise_v (and the same for min_peak): a bound on 1 / g
         rg_min = 0;
         if (P.rise * hs > 0) rg_min = P.screen_rise_v * 1.002f / (P.rise * hs);
         if (amp_on && P.min_peak * hs > 0) { const float b2 = P.screen_minpk_v * 1.002f / (P.min_peak * hs); if (b2 > rg_min) rg_min = b2; } };
      if (lean && alpha_agc && w.peakcount > 15 && w.v_avg_height_count == 0) enter_steady();      // (mode 1: a chain that comes back from k_gain_s)
      // mode 1, a chain that did not get through its segments (one of them stopped at a record that is the general step's, or did not start in the state it had
      // assumed): this lane walks on from there - but only to the END OF THAT SEGMENT.  The segments behind it were walked from guessed states that their joins
      // could not check once the chain in front of them had broken; if the walker arrives at the next segment's first record in exactly the state that segment
      // assumed there (gs_same: every field the step reads), the segment's result is the true one (it ran from the true state) and the induction of k_gain_join goes
      // on from here: the walker takes the segment's end state and event count and looks at the next one.  A chain with a handful of such records costs a handful
      // of segment remainders (<= pk_seg_recs records each) instead of everything behind the first (a 4 KB block: 4 100 records, 2.7 ms of dependent steps).
      int rj_seg0 = 0, rj_nseg = 0, kseg = 0;
      long long i_lim = 0x7fffffffffffffffll;
      bool rejoined = false;
      if (mode == 1 && active && steady && cst[ci].nseg > 0 && cfg.pk_rejoin) {
         rj_seg0 = cst[ci].seg0; rj_nseg = cst[ci].nseg;
         const long long f0 = segs[rj_seg0].first;
         kseg = (int)((cst[ci].i - f0) / cfg.pk_seg_recs);
         if (kseg < 0) kseg = 0;
         if (kseg >= rj_nseg) { rj_nseg = 0; } else i_lim = segs[rj_seg0 + kseg].end; }
      auto step = [&](const int j, const long long idx) -> int {
         if (idx >= src.iend) return 2;
         if (idx >= i_lim) return 4;                                        // (mode 1: at the next segment's first record - see rejoin)
         const uint4 cur4 = s_rec[j][lane];
         if (steady) {
            // ---- steady state, the common record: everything static about it is in its kCrClear flag ----
            const int pos = (int)cur4.x, c32 = (int)c;
            const uint32_t w0 = cur4.y;
            if ((w0 & kCrBad) ? pos + kSfTile + W < c32 : pos + W - 2 < c32) return 0;      // its rows are behind the countdown for good
            const int val = (int)(int16_t)(cur4.z & 0xffffu);
            const bool top = !(w0 & 0x800u);
            const int a = top ? val : -val;
            if (!(w0 & kCrBad) && cur4.z != 0xffff8000u && amp_on && a <= w.min_lo) return 0;      // below the amplitude test for sure: passed over (see below)
            if (crec_weak_dead(w0, w.rise_lo)) return 0;                                          // no row can pass the rise test: passed over likewise
            const int f = pos + (int)((w0 >> 12) & 63u), fl = f + (int)((w0 >> 18) & 15u);
            const float g = w.agc_gain;
            if ((w0 & kCrClear) && c32 <= f && fl < limit32 && w.rise_hi <= crec_level(w0, S.sure_i) && (!amp_on || a >= w.min_hi) && w.nevents < cap && g >= g_min) {
               s_notes[nbuf][lane] = make_uint4((uint32_t)idx, __float_as_uint(g), __float_as_uint(w.v_avg_height), 0xffffffffu);
               ++nbuf; ++w.nevents; ++w.peakcount;
               c = pos + W + 1;
               // g = alpha h / lastheight + (1 - alpha) g, clamped (src/decoder.c:505-512); the callback sees the heights of the peaks BEFORE this one (:587-590)
               const float lastheight = w.v_lasttop - w.v_lastbot, v = __uint_as_float(cur4.w);
               float g2 = g;
               if (lastheight > 0) { g2 = alpha * (w.v_avg_height / lastheight) + beta * g; if (g2 > 2.0f) g2 = 2.0f; }
               if (top) { w.v_top = v; w.v_lasttop = v; } else { w.v_bot = v; w.v_lastbot = v; }
               w.agc_gain = g2;
               // the integer bands around the thresholds of src/decoder.c:785-786 from a 1-ulp reciprocal (approx_thresholds: two more lsb of guard)
               const float rg = fast_rcp(g2);
               if (g2 > 0 && rg >= rg_min) {
                  const int r = (int)(kr * rg), m = (int)(km * rg);
                  w.rise_lo = r - 2; w.rise_hi = r + 3; w.min_lo = m - 2; w.min_hi = m + 3;
                  w.thr_dirty = true;
                  return 0; }
               if (!(g2 > 0)) { w.flags |= RTFE_F_DETECTOR_FATAL; failed = true; why = 4; return 2; }      // src/decoder.c:782
               update_thresholds(w, P, lsb);                                 // thresholds near the screen's: exactly
               if (w.flags & RTFE_F_SCREEN_UNDERFLOW) { failed = true; why = 5; return 2; }
               return 0; }
            return 1; }
         const int pos = (int)cur4.x;
         const uint32_t w0 = cur4.y, w1 = cur4.z;
         const int c32 = (int)c;
         const bool plain = !(w0 & kCrBad) && w1 != 0xffff8000u;
         if ((w0 & kCrBad) ? pos + kSfTile + W < c32 : pos + W - 2 < c32) return 0;      // its rows are behind the countdown for good
         const int f = pos + (int)((w0 >> 12) & 63u), nlead = (int)((w0 >> 18) & 15u);
         const bool top = !(w0 & 0x800u);
         const int val = (int)(int16_t)(w1 & 0xffffu);
         const int a = top ? val : -val;
         // (a record whose extreme is below the amplitude test for sure cannot fire while the thresholds stand, and they stand until something
         //  fires - behind which all of this record's rows are blind: it is passed over)
         if (lean && plain && amp_on && a <= w.min_lo) return 0;
         if (lean && plain && crec_weak_dead(w0, w.rise_lo)) return 0;
         // ---- the fast path: a record with a sure stretch, the countdown over before its first row, the thresholds inside the band the sure
         // level stands for, a clear amplitude, and nothing else that could fire before this record's owner has left the window.  Then it
         // fires - at one of its lead rows or at its first sure row, k_emit will say which - and all that feeds back is the extreme's value. ----
         const float g = w.agc_gain;
         const bool fire = lean && plain && (w0 & kCrClear) && c32 <= f && f + nlead < limit32
                           && w.rise_hi <= crec_level(w0, S.sure_i) && (!amp_on || a >= w.min_hi) && w.nevents < cap && g >= g_min;
         if (!fire) {
            return 1; }
         const float v = __uint_as_float(cur4.w);
         s_notes[nbuf][lane] = make_uint4((uint32_t)idx, __float_as_uint(g), __float_as_uint(w.v_avg_height), 0xffffffffu);
         ++nbuf; ++w.nevents;
         c = pos + W + 1;
         // the block decoder's whole AGC schedule (start-up, window AGC, density detection)
         if (top) w.v_top = v; else w.v_bot = v;
         agc_after_peak_m(w, cmode, agc_off, P, heights, top, 0.0);
         if (!(w.agc_gain > 0)) { w.flags |= RTFE_F_DETECTOR_FATAL; failed = true; why = 4; return 2; }      // src/decoder.c:782
         if (!approx_thresholds(w, P, lsb)) {
            update_thresholds(w, P, lsb);
            if (w.flags & RTFE_F_SCREEN_UNDERFLOW) { failed = true; why = 5; return 2; } }
         if (lean && alpha_agc && w.peakcount > 15 && w.v_avg_height_count == 0) { enter_steady(); if (mode == 0) return 3; }      // (the steady stretch is k_gain_s's)
         return 0; };
      // ---- the general step (a lane that cannot take the fast path waits for the round's end: the wave pays for it once per round, not once per step) ----
      auto general = [&]() -> int {
         // earliest firing run among the tops and the bottoms from the first live record on
         if (w.thr_dirty) {
            update_thresholds(w, P, lsb);
            if (w.flags & RTFE_F_SCREEN_UNDERFLOW) { failed = true; why = 5; return 2; } }
         RecIt alive;
         it_open(alive, src, i);
         for (;;) {                                                        // records whose rows are behind the countdown (a list that is not there: its whole tile)
            if (alive.end) { if (alive.bad && alive.pos + kSfTile + W < c) { alive.end = false; alive.bad = false; ++alive.i; it_land(alive, src); continue; } break; }
            if (alive.pos + W - 2 >= c) break;
            it_next(alive, src); }
         if (alive.end) { if (alive.bad) { failed = true; why = 1; } return 2; }
         long long best = kNoRow, best_doubt = kNoRow;
         bool have = false, best_top = false, top_done = false, bot_done = false;
         long long bad_row0 = kNoRow;                                       // first row of a tile whose list is not there
         Run bu = {};
         for (RecIt j = alive; !(top_done && bot_done); it_next(j, src)) {
            if (j.end) { if (j.bad) bad_row0 = j.pos; break; }
            const Run u = run_decode(j.w0, j.w1, j.pos);
            if (u.top ? top_done : bot_done) continue;
            if (u.f > best || u.f >= limit) { if (u.top) top_done = true; else bot_done = true; continue; }
            long long dr;
            const long long n = run_fire(w, u, j.eend, msrc, c, limit, W, S.sure_i, mv, dr);
            if (dr < best_doubt) best_doubt = dr;
            if (n != kNoRow) {
               if (u.top) top_done = true; else bot_done = true;         // (runs of one kind are ordered by row)
               if (n < best || (n == best && u.top && !best_top)) { best = n; bu = u; best_top = u.top; have = true; } } }
         if (bad_row0 != kNoRow && !(have && best < bad_row0)) { failed = true; why = 1; return 2; }      // (a list that is not there, and something in it might fire first)
         if (best_doubt != kNoRow && best_doubt <= best) { failed = true; why = 2; return 2; }
         if (!have) return 2;                                               // nothing fires any more: the chain is done
         // ---- detection: refine_peak + the callback's effect on AGC state (src/decoder.c:700-749, 574-609) ----
         const Run &u = bu;
         const long long ndet = best + d;                                 // the detector's row
         const int ld = (int)(u.pos - best) + W;                           // left_distance
         const float g = w.agc_gain;
         const float thr = 0.005f / g;
         const int ti = (int)floorf(thr * lsb);
         if (ti + 2 > 253) { failed = true; why = 3; return 2; }                    // (neighbour distances are stored up to 253)
         const int val_i = u.val;
         const int iprev = u.top ? val_i - u.dprev : val_i + u.dprev, inext = u.top ? val_i - u.dnext : val_i + u.dnext;
         const int adjcode = refine_code(&cfg, val_i, iprev, inext, g, u.top);
         const float val = volt(val_i, mv);
         double t_peak = 0;
         if (cmode == RTFE_PE && !w.datablock && w.peakcount >= 68) {
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
         agc_after_peak_m(w, cmode, agc_off, P, heights, u.top, t_peak);
         if (!(w.agc_gain > 0)) { w.flags |= RTFE_F_DETECTOR_FATAL; failed = true; why = 4; return 2; }      // src/decoder.c:782
         update_thresholds(w, P, lsb);
         if (w.flags & RTFE_F_SCREEN_UNDERFLOW) { failed = true; why = 5; return 2; }
         c = u.pos + W + 1;
         i = alive.i;
         // A record the lean steps passed over (below the amplitude or the rise test for sure: it could not fire while the thresholds stood) is done with if its rows end
         // before the new countdown does - its owner at most two rows behind the fired record's, and a stream being in the order of its candidates that holds whenever
         // the fired record is its own candidate.  A stale minimum's record is owned by a sample in front of its candidate: then the walk goes back to the
         // first record whose first row says that every candidate in front of it ends before the countdown (candidates lie in front of their rows).
         for (int back = 0; i > 0; ++back) {
            const CRec pr = src.rec[i - 1];
            const long long bound = (pr.w0 & kCrBad) ? (long long)pr.pos : (long long)pr.pos + (long long)((pr.w0 >> 12) & 63u) - 1;      // its candidate's row at most
            if (bound + W - 2 < c) break;
            if (back >= 64) { failed = true; why = 7; return 2; }
            --i; }
         if (!steady && lean && alpha_agc && w.peakcount > 15 && w.v_avg_height_count == 0) enter_steady();
         return 1; };
      // ---- mode 1 at a segment boundary: i = the next segment's first record ----
      auto rejoin = [&]() {
         int k2 = kseg + 1;
         GsState cur; cur.g = w.agc_gain; cur.vlt = w.v_lasttop; cur.vlb = w.v_lastbot; cur.c = (int)c; cur.rise_hi = w.rise_hi; cur.min_lo = w.min_lo; cur.min_hi = w.min_hi;
         bool adopted = false;
         while (k2 < rj_nseg) {
            GsSeg &sg = segs[rj_seg0 + k2];
            if (sg.first != i || !gs_same(sg.at_first, cur) || (unsigned long long)w.nevents + (unsigned)sg.cnt > cap) break;
            sg.stands = 1; sg.evoff = w.nevents;
            w.nevents += (unsigned)sg.cnt; w.peakcount += sg.cnt; cur = sg.at_end; i = sg.stop; adopted = true;
            if (sg.stop < sg.end) break;                                   // (it stopped at a record of its own: the walk goes on from there, to its end)
            ++k2; }
         if (adopted) {
            w.agc_gain = cur.g; w.v_lasttop = cur.vlt; w.v_lastbot = cur.vlb; w.v_top = cur.vlt; w.v_bot = cur.vlb; c = cur.c;
            w.rise_hi = cur.rise_hi; w.rise_lo = cur.rise_hi - 5; w.min_lo = cur.min_lo; w.min_hi = cur.min_hi; w.thr_dirty = true;
            rejoined = true; }
         kseg = k2;
         i_lim = k2 < rj_nseg ? segs[rj_seg0 + k2].end : 0x7fffffffffffffffll; };
      uint4 q[kGainChunk + 1];
      #pragma unroll
      for (int j = 0; j <= kGainChunk; ++j) q[j] = make_uint4(0, 0, 0, 0);
      const uint4 *rec4 = reinterpret_cast<const uint4 *>(src.rec);
      auto fetch = [&](long long i0) {
         #pragma unroll
         for (int j = 0; j <= kGainChunk; ++j) if (i0 + j < src.iend) q[j] = rec4[i0 + j]; };
      auto put = [&]() {
         #pragma unroll
         for (int j = 0; j <= kGainChunk; ++j) s_rec[j][lane] = q[j]; };
      int st2 = active ? 0 : 2;                                            // 0: in lock step, 1: waiting for the general step, 2: done, 3: (mode 0) steady from here
      bool handed = false;
      if (st2 == 0) fetch(i);
      put();
      if (st2 == 0) fetch(i + kGainChunk);
      const bool prof = cfg.debug == 4 && lane == 0;
      long long pc_steps = 0, pc_gen = 0, pc_bound = 0, pn_chunks = 0, pn_gen = 0;
      for (;;) {
         long long tk0 = 0, tk1 = 0, tk2 = 0;
         if (prof) tk0 = clock64();
         int jp = 0;                                                       // records of the chunk this lane is through with
         #pragma nounroll
         for (int j0 = 0; j0 < kGainChunk; j0 += 4) {
            #pragma unroll
            for (int jj = 0; jj < 4; ++jj)
               if (st2 == 0) { st2 = step(j0 + jj, i + j0 + jj); if (st2 == 0 || st2 == 3) jp = j0 + jj + 1; }
            if (__ballot(st2 == 0) == 0) break; }
         bool resync = false;
         if (prof) { tk1 = clock64(); pc_steps += tk1 - tk0; ++pn_chunks; if (__ballot(st2 == 1)) ++pn_gen; }
         if (st2 == 4) { i += jp; jp = 0; flush_notes(); rejoin(); st2 = 0; resync = true; }      // (mode 1: the next segment's first record)
         else if (st2 == 1) { i += jp; jp = 0; flush_notes(); st2 = general(); if (st2 != 2) st2 = 0; resync = true; if (st2 == 0 && mode == 0 && steady) st2 = 3; }      // (the general step leaves i at the record to go on with)
         else if (st2 == 0) i += kGainChunk;
         if (st2 == 3) { i += jp; flush_notes(); handed = true; st2 = 2; }      // the baseline is fixed: the chain waits for k_gain_s
         if (prof) { tk2 = clock64(); pc_gen += tk2 - tk1; }
         if (__ballot(st2 != 2) == 0) { flush_notes(); break; }
         if (resync && st2 == 0) fetch(i);                                   // (a lane the general step moved: its chunk afresh - the wave waits for it)
         put();
         if (st2 == 0) fetch(i + kGainChunk);
         flush_notes();                                                     // (behind the loads: the wait for them at the next chunk's end finds the stores long done)
         if (prof) pc_bound += clock64() - tk2; }
      if (prof) { atomicAdd(&scratch->dbg2[0], (unsigned long long)pc_steps); atomicAdd(&scratch->dbg2[1], (unsigned long long)pc_gen); atomicAdd(&scratch->dbg2[2], (unsigned long long)pc_bound);
                  atomicAdd(&scratch->dbg2[3], (unsigned long long)pn_chunks); atomicAdd(&scratch->dbg2[4], (unsigned long long)pn_gen); atomicAdd(&scratch->dbg2[5], 1ull); }
      // ---- publish ----
      if (!active) continue;
      if (handed && !failed) {                                             // (steady: everything the walker is, for k_gain_s)
         ChainSt &cs = cst[ci];
         cs.w = w; cs.i = i; cs.c = c; cs.iend = src.iend; cs.seg0 = 0; cs.nseg = 0; cs.status = kChSteady; cs.seg_ev0 = w.nevents; cs.seg_ev1 = w.nevents; cs.pad = 0;
         cs.k.ev_index = (unsigned long long)(ev - events); cs.k.h = w.v_avg_height; cs.k.alpha = alpha; cs.k.kr = kr; cs.k.km = km; cs.k.rg_min = rg_min; cs.k.g_min = g_min;
         cs.k.W = W; cs.k.sure_i = S.sure_i; cs.k.limit32 = limit32; cs.k.amp_on = amp_on ? 1 : 0; cs.k.sl = sl; cs.k.pad = 0;
         // the steady stretch's segments: their places in the table (any order: one atomic per chain), their entries
         {  const long long len = src.iend > i ? src.iend - i : 0;
            const int SR = cfg.pk_seg_recs;
            int nseg = len <= 0 ? 0 : (int)((len + SR - 1) / SR);
            int seg0 = nseg > 0 ? atomicAdd(&scratch->nsegs, nseg) : 0;
            if ((long long)seg0 + nseg > seg_cap) nseg = 0;                 // (no room in the table: the chain goes to k_gain, mode 1, as a whole; the count stays an upper bound)
            cs.seg0 = seg0; cs.nseg = nseg;
            for (int sg = 0; sg < nseg; ++sg) {
               GsSeg &o = segs[seg0 + sg];
               o.chain = ci; o.sidx = sg; o.first = i + (long long)sg * SR; o.end = sg + 1 == nseg ? src.iend : i + (long long)(sg + 1) * SR; o.stands = 0;
               o.ev_index = (unsigned long long)(ev - events); o.reset = reset; o.h = w.v_avg_height; o.sl = sl; } }
         for (int k = 0; k < 10; ++k) cs.heights[k] = heights[k];
         if (w.v_avg_height > 0) atomicMax(&scratch->min_height_key, 0x7fffffff - (int)__float_as_uint(w.v_avg_height));      // (a chain that is handed over has learned its height)
         if (n_slow) atomicAdd(&scratch->dbg[1], (unsigned long long)n_slow);
         if (w.nevents > n_slow) atomicAdd(&scratch->dbg[0], (unsigned long long)(w.nevents - n_slow));      // (statistics: the head's events on the fast path)
         continue; }
      cst[ci].status = kChDone;
      if (mode == 0) { cst[ci].seg_ev0 = 0; cst[ci].seg_ev1 = 0; cst[ci].pad = 0; }        // (no steady stretch: every event of the chain is k_emit's)
      if (rejoined) cst[ci].pad = 1;                                      // (k_emit: the chain's notes lie between several runs of k_emit_seg's events)
      n_fast = w.nevents - n_slow - (mode == 1 ? cst[ci].w.nevents : 0u);
      if (n_fast) atomicAdd(&scratch->dbg[0], (unsigned long long)n_fast);
      if (n_slow) atomicAdd(&scratch->dbg[1], (unsigned long long)n_slow);
      if (failed) { atomicExch(&ctl[b].status, (int)kBurstNeedsFull); atomicAdd(&scratch->why[why & 7], 1ull); }
      counts[((size_t)b * cfg.nparm + pidx) * ntrks + trk] = w.nevents < cap ? w.nevents : cap;
      chain_h[(size_t)b * nwalk + wi] = w.v_avg_height;
      if (mode == 0 && !agc_off && w.peakcount > 15 && w.v_avg_height_count == 0 && w.v_avg_height > 0) atomicMax(&scratch->min_height_key, 0x7fffffff - (int)__float_as_uint(w.v_avg_height));
      if (w.flags & ~(unsigned)RTFE_F_SCREEN_UNDERFLOW) atomicOr(&ctl[b].bflags, w.flags & ~(unsigned)RTFE_F_SCREEN_UNDERFLOW); } }

// ------------------------------------------------------------------------------------------------
// The steady stretch of every chain (NRZI / GCR, alpha-filter AGC: peakcount > 15, the baseline fixed), and nothing but the common
// record - straight-line code, every decision a select.  Per record: its rows are behind the countdown -> on; its extreme is below the
// amplitude test for sure -> on (it cannot fire while the thresholds stand, and they stand until something fires - behind which all its
// rows are blind); kCrClear and the countdown over before its first row, the thresholds inside the band its sure stretch stands for,
// amplitude clear -> it fires: note (k_emit finds the row), g = alpha h / lastheight + (1 - alpha) g (src/decoder.c:505-512), the
// integer bands around the thresholds of src/decoder.c:785-786 from 1 / g.  Anything else: the walk stops there and k_gain (mode 1)
// takes the chain from that record on.
//
// A chain is sequential, a 4 KB block's is 4 100 records long, and 18 k chains are 290 waves on 1 024 SIMDs - so the stretch is cut
// into SEGMENTS of pk_seg_recs records, a lane each:
//   (k_gain, mode 0, gives every chain it hands over its place in the segment table - one atomic per chain - and writes the entries)
//   k_gain_seg<0>  every segment on its own: segment 0 from the chain's true state, the others from a GUESS (the chain's gain and
//              peaks at the hand-over, no countdown) a warm-up of pk_seg_warm records early - the countdown re-joins the true sequence
//              at the first fired record, the peak memory after a top and a bottom, the alpha filter forgets its start value
//              geometrically - noting the state at its first own record, at its end, and its number of events;
//   k_gain_join  per chain: segment s stands if segment s - 1 stands, ran through, and ENDED in exactly the state segment s assumed
//              at its start - every field the step reads, bit for bit; by induction from segment 0 every standing segment ran
//              from the true state.  Event offsets (a running sum), the chain's state behind the last standing segment;
//   k_gain_seg<1>  the standing segments again, now writing their notes where they belong.
// From the first segment that does not stand the chain goes on in k_gain (mode 1) from the last proven state.  Nothing rests on
// the convergence argument; a warm-up too short only costs time (RTFE_SEG_RECS / RTFE_SEG_WARM: tests force that).
// ------------------------------------------------------------------------------------------------
// Every segment from its (true or guessed) start state.  What it leaves behind: the states at its first record and at its end, its event
// count, and per own record the gain in force if the record fired, else 0 (gfire, pk_seg_recs floats per segment) - k_emit_seg makes the
// events of the segments that stand from those.
// A lane per segment, eight records at a time.  The wave loads TOGETHER: a lane reading its own records (16 bytes here, 16 bytes 2 KB
// further on for the next lane) makes every load instruction 64 requests for a quarter of a line each; so load instruction k fetches the
// 128-byte pieces of eight lanes' segments (lanes 8k .. 8k+7: eight lanes per piece, 16 bytes each - whole lines), through registers
// (a chunk ahead) into LDS, where every lane then finds its own eight records.
constexpr int kGsRegs = 8, kGsDepth = 1;
constexpr int kGsPitch = kGsRegs + 1;                                 // a lane's slot in LDS, in 16-byte units (odd: conflict-free 128-bit reads)
__global__ void __launch_bounds__(64) k_gain_seg(const DevCfg *__restrict__ cfgp, const ChainSt *__restrict__ cst, BurstScratch *__restrict__ scratch,
                                                 const CRec *__restrict__ crec, long long ccap, GsSeg *__restrict__ segs, const int *__restrict__ nsegs_p, long long seg_cap, float *__restrict__ gfire) {
   __shared__ uint4 s_rec[64 * kGsPitch];
   const int lane = threadIdx.x;
   const int nsegs = (int)min((long long)*nsegs_p, seg_cap);      // (the count includes chains that found the table full: their entries were never written)
   const int nchains = scratch->nbursts * cfgp->nparm * cfgp->ntrks;
   const int S = cfgp->pk_seg_recs;
   const bool prof = cfgp->debug == 6 && threadIdx.x == 0;
   long long pt0 = 0, pt_setup = 0, pt_steps = 0, pn_chunks = 0, pn_items = 0;
   for (int sbase = blockIdx.x * 64; sbase < nsegs; sbase += gridDim.x * 64) {
      const int si = sbase + lane < nsegs ? sbase + lane : nsegs - 1;
      if (prof) { pt0 = clock64(); ++pn_items; }
      GsSeg sg = segs[si];
      bool active = sbase + lane < nsegs && (unsigned)sg.chain < (unsigned)nchains;
      if (!active) sg.chain = 0;
      const ChainSt &cs = cst[sg.chain];
      active = active && cs.status == kChSteady && si >= cs.seg0 && si < cs.seg0 + cs.nseg && sg.sidx == si - cs.seg0;      // (an entry its chain wrote)
      if (!active) { sg.sidx = 0; sg.first = 0; sg.end = 0; }
      const GsConst K = cs.k;
      const int W = K.W, sure_i = K.sure_i, limit32 = K.limit32;
      const bool amp_on = K.amp_on != 0;
      const float h = K.h, alpha = K.alpha, beta = 1 - K.alpha, kr = K.kr, km = K.km, rg_min = K.rg_min, g_min = K.g_min;
      const uint4 *rec4 = reinterpret_cast<const uint4 *>(crec + (size_t)K.sl * ccap);
      float *gout = gfire + (size_t)si * S;
      // where the walk begins and in which state: the chain's as k_gain (mode 0) left it - segment 0's true state, the others' guess
      float g = cs.w.agc_gain, vlt = cs.w.v_lasttop, vlb = cs.w.v_lastbot;
      int rise_hi = cs.w.rise_hi, min_lo = cs.w.min_lo, min_hi = cs.w.min_hi;
      int c = sg.sidx == 0 ? (int)cs.c : 0;
      long long i = sg.first;
      if (sg.sidx > 0) { i = sg.first - cfgp->parm[(sg.chain % (cfgp->nparm * cfgp->ntrks)) / cfgp->ntrks].seg_warm; if (i < cs.i) i = cs.i; }
      unsigned int nev = 0;
      if (prof) pt_setup += clock64() - pt0;
      // One phase of the walk: records [i, to) - `own`: the lane stops at a record that is not the step's business and leaves the gains
      // behind (else it passes over such a record: a warm-up only has to arrive in the right state, and the join says whether it did).
      // Wave-uniform loops.
      auto walk = [&](const long long to, const bool own, const bool act) -> long long {
         bool run = act && i < to;
         const int piece = lane & 7, sub = lane >> 3;
         float hq = h / (vlt - vlb);                                           // h / lastheight as of the last fired record (src/decoder.c:505-512)
         bool lhpos = vlt - vlb > 0;
         // the pieces this lane fetches for the others: of lane 8k + sub's records [at, at + 8) the one numbered `piece`.  kGsDepth
         // chunks are in flight (measured: 3 instead of 1 change nothing - the walk is bound by the step's dependent instructions).
         uint4 q[kGsDepth][kGsRegs];
         auto fetch = [&](uint4 (&dst)[kGsRegs], const long long at) {
            const unsigned long long mine = reinterpret_cast<unsigned long long>(rec4 + at);
            const int left = run ? (int)((to - at) < kGsRegs ? (to - at > 0 ? to - at : 0) : kGsRegs) : 0;
            #pragma unroll
            for (int k = 0; k < kGsRegs; ++k) {
               const int src = 8 * k + sub;
               const unsigned lo = (unsigned)__shfl((int)(unsigned)mine, src), hi = (unsigned)__shfl((int)(unsigned)(mine >> 32), src);
               const int n = __shfl(left, src);
               dst[k] = make_uint4(0, kCrBad, 0, 0);
               if (piece < n) dst[k] = reinterpret_cast<const uint4 *>(((unsigned long long)hi << 32) | lo)[piece]; } };
         #pragma unroll
         for (int dd = 0; dd < kGsDepth; ++dd) fetch(q[dd], i + dd * kGsRegs);
         bool more = __ballot(run) != 0ull;
         while (more) {
            #pragma unroll
            for (int dd = 0; dd < kGsDepth; ++dd) {
               if (!more) break;
               long long tc0 = 0;
               if (prof) { tc0 = clock64(); ++pn_chunks; }
               #pragma unroll
               for (int k = 0; k < kGsRegs; ++k) s_rec[(8 * k + sub) * kGsPitch + piece] = q[dd][k];
               rtfe_wave_sync();
               fetch(q[dd], i + kGsDepth * kGsRegs);                            // this set again, kGsDepth chunks on
               int adv = 0;
               const bool ran = run;
               float gb[kGsRegs];
               #pragma unroll
               for (int j = 0; j < kGsRegs; ++j) {
                  const uint4 r = s_rec[lane * kGsPitch + j];
                  const bool inr = run && i + j < to;
                  const int pos = (int)r.x;
                  const uint32_t w0 = r.y;
                  const bool bad = w0 & kCrBad;
                  const bool dead = bad ? pos + kSfTile + W < c : pos + W - 2 < c;
                  const int val = (int)(int16_t)(r.z & 0xffffu);
                  const bool top = !(w0 & 0x800u);
                  const int a = top ? val : -val;
                  const bool ampdead = (!bad && r.z != 0xffff8000u && amp_on && a <= min_lo) || crec_weak_dead(w0, rise_hi - 5);      // (rise_lo = rise_hi - 5: the band around the threshold)
                  const int f = pos + (int)((w0 >> 12) & 63u), fl = f + (int)((w0 >> 18) & 15u);
                  const bool fire = (w0 & kCrClear) && c <= f && fl < limit32 && rise_hi <= crec_level(w0, sure_i) && (!amp_on || a >= min_hi) && g >= g_min;
                  // (h / lastheight for the NEXT fired record, should this one fire: its operands are this record's and the state's - the
                  //  division runs beside the gain's chain instead of inside it)
                  const float v = __uint_as_float(r.w);
                  const float lh_f = top ? v - vlb : vlt - v;
                  const float q_f = h / lh_f;
                  float g2 = alpha * hq + beta * g;
                  g2 = g2 > 2.0f ? 2.0f : g2;
                  g2 = lhpos ? g2 : g;
                  const float rg = fast_rcp(g2);
                  const bool ok = inr && !dead && !ampdead && fire && g2 > 0 && rg >= rg_min;
                  gb[j] = ok ? g : 0.0f;                                       // (the gain in force when it fired: what its event carries)
                  nev += ok ? 1u : 0u;
                  c = ok ? pos + W + 1 : c;
                  vlt = ok && top ? v : vlt; vlb = ok && !top ? v : vlb;
                  hq = ok ? q_f : hq; lhpos = ok ? lh_f > 0 : lhpos;
                  g = ok ? g2 : g;
                  const int rr = (int)(kr * rg), mm = (int)(km * rg);
                  rise_hi = ok ? rr + 3 : rise_hi; min_lo = ok ? mm - 2 : min_lo; min_hi = ok ? mm + 3 : min_hi;
                  const bool on = inr && (dead || ampdead || ok || !own);
                  if (run && !on) { run = false; adv = j; } }
               if (run) adv = kGsRegs;
               if (own && ran) {                                               // (i - first is a multiple of eight here, and so is the segment's room)
                  float4 *o4 = reinterpret_cast<float4 *>(gout + (i - sg.first));
                  o4[0] = make_float4(gb[0], gb[1], gb[2], gb[3]); o4[1] = make_float4(gb[4], gb[5], gb[6], gb[7]); }
               rtfe_wave_sync();
               i += adv;
               if (i >= to) run = false;
               more = __ballot(run) != 0ull;
               if (prof) pt_steps += clock64() - tc0; } }
         return i; };
      walk(sg.first, false, active && sg.sidx > 0);                         // the warm-up: arrive at the first own record
      i = sg.first; nev = 0;
      GsState s0; s0.g = g; s0.vlt = vlt; s0.vlb = vlb; s0.c = c; s0.rise_hi = rise_hi; s0.min_lo = min_lo; s0.min_hi = min_hi;
      const long long at = walk(sg.end, true, active);
      if (active) {
         GsSeg &o = segs[si];
         o.at_first = s0;
         o.at_end.g = g; o.at_end.vlt = vlt; o.at_end.vlb = vlb; o.at_end.c = c; o.at_end.rise_hi = rise_hi; o.at_end.min_lo = min_lo; o.at_end.min_hi = min_hi;
         o.stop = at > sg.end ? sg.end : at; o.cnt = (int)nev; } }
   if (prof) { atomicAdd(&scratch->dbg2[0], (unsigned long long)pt_setup); atomicAdd(&scratch->dbg2[1], (unsigned long long)pt_steps);
               atomicAdd(&scratch->dbg2[3], (unsigned long long)pn_chunks); atomicAdd(&scratch->dbg2[4], (unsigned long long)pn_items); } }

// per chain: which segments stand, where their notes go, and where the chain stands behind them
__global__ void __launch_bounds__(64) k_gain_join(const DevCfg *__restrict__ cfgp, ChainSt *__restrict__ cst, const rtfe_burst *__restrict__ bursts, BurstScratch *__restrict__ scratch,
                                                  const BurstCtl *__restrict__ ctl, uint32_t *__restrict__ counts, float *__restrict__ chain_h, GsSeg *__restrict__ segs) {
   const DevCfg &cfg = *cfgp;
   const int ntrks = cfg.ntrks, nwalk = cfg.nparm * ntrks;
   const int nchains = chain_count(cfg, scratch->nbursts);
   for (int cu = blockIdx.x * 64 + threadIdx.x; cu < nchains; cu += gridDim.x * 64) {
      const ChainIx cx_ = chain_ix(cfg, cu);
      const int ci = cx_.ci, b = cx_.b, wi = cx_.wi, pidx = cx_.pidx, trk = cx_.trk;
      if (ctl[b].status != kBurstReady || cst[ci].status != kChSteady) continue;
      ChainSt &cs = cst[ci];
      const unsigned int cap = bursts[b].event_cap;
      const unsigned int nev0 = cs.w.nevents;
      unsigned int nev = nev0;
      long long resume = cs.i;                                            // where k_gain (mode 1) goes on, if it has to
      GsState st; st.g = cs.w.agc_gain; st.vlt = cs.w.v_lasttop; st.vlb = cs.w.v_lastbot; st.c = (int)cs.c; st.rise_hi = cs.w.rise_hi; st.min_lo = cs.w.min_lo; st.min_hi = cs.w.min_hi;
      bool through = cs.nseg > 0;
      for (int k = 0; k < cs.nseg; ++k) {
         GsSeg &sg = segs[cs.seg0 + k];
         if (!gs_same(sg.at_first, st) || (unsigned long long)nev + (unsigned)sg.cnt > cap) { through = false; break; }      // (segment 0's start state IS the chain's; an event list that would overflow is the general step's to flag)
         sg.stands = 1; sg.evoff = nev;
         nev += (unsigned)sg.cnt; st = sg.at_end; resume = sg.stop;
         if (sg.stop < sg.end) { through = false; break; } }
      if (nev > nev0) atomicAdd(&scratch->dbg[0], (unsigned long long)(nev - nev0));
      cs.w.agc_gain = st.g; cs.w.v_lasttop = st.vlt; cs.w.v_lastbot = st.vlb; cs.w.v_top = st.vlt; cs.w.v_bot = st.vlb;
      cs.w.peakcount += (int)(nev - nev0); cs.w.nevents = nev; cs.i = resume; cs.c = st.c;
      cs.seg_ev0 = nev0; cs.seg_ev1 = nev;                                // (events [nev0, nev): k_emit_seg's; the rest k_emit's)
      if (through) {
         cs.status = kChDone;
         counts[((size_t)b * cfg.nparm + pidx) * ntrks + trk] = nev < cap ? nev : cap;
         chain_h[(size_t)b * nwalk + wi] = cs.w.v_avg_height; }
      else cs.status = kChGeneral; } }
// ------------------------------------------------------------------------------------------------
// k_emit: the events the fast path noted -> the events the reference's callbacks see.  One workgroup per chain at a time,
// a lane per event (16 bytes in, 16 bytes out, consecutive lanes consecutive events).
// ------------------------------------------------------------------------------------------------
// the event of a record that fired at gain `gain` (exact thresholds at that gain, the first lead row that passes, refine_peak)
__device__ __forceinline__ rtfe_event emit_event(const DevCfg &cfg, const DevParm &P, Walker &wk, const CRec &r, const uint16_t *eend, const MarSrc &ms, float gain, int W, int d, long long reset,
                                                 int trk, int pidx, float mv) {
   const Run u = run_decode(r.w0, r.w1, (long long)r.pos);
   wk.agc_gain = gain; wk.flags = 0;
   update_thresholds(wk, P, cfg.lsb_per_volt);                      // the exact thresholds of src/decoder.c:785-786 at that gain
   long long n = u.f + u.nlead;                                     // the first sure row, unless a lead row passes
   for (int j = u.nlead - 1; j >= 0; --j) if (rise_pass(wk, u.top, u.val, run_margin(ms, u, eend, j), mv)) n = u.f + j;
   const int ld = (int)(u.pos - n) + W;
   const int iprev = u.top ? u.val - u.dprev : u.val + u.dprev, inext = u.top ? u.val - u.dnext : u.val + u.dnext;
   const int adjcode = refine_code(&cfg, u.val, iprev, inext, gain, u.top);
   const float val = r.volt;
   rtfe_event e;
   e.sample = (uint32_t)(n + d - reset);
   e.v_peak = (cfg.invert && val == 0.0f) ? -0.0f : val;
   e.agc_gain = gain;
   e.trk = (uint8_t)trk;
   e.flags = (uint8_t)((u.top ? 0 : 1) | (adjcode << 1));
   e.left_distance = (uint8_t)ld;
   e.parmset = (uint8_t)pidx;
   return e; }

// the events of the steady stretches: a wave per standing segment, a lane per record - its gain says whether it fired, a prefix sum where
// its event goes.  Records, entry references and gains are read in stream order.
// (rounds of 64 records a wave loads together against waves a SIMD: 4 rounds at 84 registers = five waves: k_emit_seg 0.253 - 0.267 ms on C2, 1.64 - 1.66 on M8; 2 rounds, seven waves:
//  0.238, 1.50 - 1.54; 1 round, 64 registers, eight waves: 0.236, 1.60, and C2's scan 1.34 against 1.35 ms - more waves hide the round trips better than a wave's own batch)
#ifndef RTFE_ES_AHEAD
#define RTFE_ES_AHEAD 1
#endif
#ifndef RTFE_ES_WAVES
#define RTFE_ES_WAVES 8
#endif
#if RTFE_ES_WAVES > 0 && !defined(RTFE_CPU_EMUL)
#define RTFE_ES_ATTR __attribute__((amdgpu_waves_per_eu(RTFE_ES_WAVES)))
#else
#define RTFE_ES_ATTR
#endif
__global__ void __launch_bounds__(256) RTFE_ES_ATTR k_emit_seg(const DevCfg *__restrict__ cfgp, const ChainSt *__restrict__ cst, const BurstCtl *__restrict__ ctl, rtfe_event *__restrict__ events,
                                                  const CRec *__restrict__ crec, const uint2 *__restrict__ cmar, long long ccap, const unsigned char *__restrict__ pool,
                                                  const GsSeg *__restrict__ segs, const int *__restrict__ nsegs_p, long long seg_cap, const float *__restrict__ gfire, int nchains_max,
                                                  const int16_t *__restrict__ rows, long long nrows) {
   const DevCfg &cfg = *cfgp;
   const int ntrks = cfg.ntrks, nwalk = cfg.nparm * ntrks;
   const int lane = threadIdx.x & 63;
   const int nsegs = (int)min((long long)*nsegs_p, seg_cap), S = cfg.pk_seg_recs;
   const float mv = cfg.maxvolts;
   for (int si = blockIdx.x * 4 + (threadIdx.x >> 6); si < nsegs; si += gridDim.x * 4) {
      const GsSeg sg = segs[si];
      if ((unsigned)sg.chain >= (unsigned)nchains_max || sg.stands != 1) continue;
      const ChainSt &cs = cst[sg.chain];
      const int cs_seg0 = cs.seg0, cs_nseg = cs.nseg;                    // (read beside the records: the entry itself says where they are)
      const int b = sg.chain / nwalk;
      const int wi = sg.chain - b * nwalk, pidx = wi / ntrks, trk = wi - pidx * ntrks;
      const DevParm &P = cfg.parm[pidx];
      const int W = P.W, d = cfg.skew[trk];
      MarSrc msrc; msrc.rows = rows; msrc.nrows = nrows; msrc.ntrks = ntrks; msrc.head = cfg.trk_to_head[trk]; msrc.sg = cfg.invert ? -1 : 1; msrc.W = W; msrc.nmar = cfg.pk_mar;
      const long long reset = sg.reset;
      const size_t sb = (size_t)((unsigned)sg.sl < (unsigned)(cfg.nscreens * ntrks) ? sg.sl : 0) * ccap;      // (a stale entry may hold anything: its loads stay inside the streams, its events are not stored)
      rtfe_event *ev = events + sg.ev_index;
      Walker wk = {};
      wk.v_avg_height = sg.h;
      // (whatever a stale entry holds, the speculative loads stay inside the stream and the gains' table)
      const long long seg_first = sg.first < 0 ? 0 : (sg.first > ccap - S ? (ccap > S ? ccap - S : 0) : sg.first);
      const long long n_own_l = sg.stop - sg.first;
      const int n_own = n_own_l < 0 ? 0 : (n_own_l > S ? S : (int)n_own_l);
      unsigned int at = sg.evoff;
      // (a lane's gain, record and margin block of a round of 64 records are loaded TOGETHER and whether the record fired or not - 93 % do: one round trip
      //  instead of two dependent ones; the kernel waits for memory 85 % of its time - RTFE_ES_AHEAD rounds a batch, see above)
      constexpr int kEsAhead = RTFE_ES_AHEAD;
      for (int k0 = 0; k0 < n_own; k0 += 64 * kEsAhead) {
         float gain[kEsAhead]; CRec rec[kEsAhead]; uint2 mar[kEsAhead];
         #pragma unroll
         for (int u = 0; u < kEsAhead; ++u) {
            const int k = k0 + 64 * u + lane;
            gain[u] = 0.0f; rec[u].pos = 0; rec[u].w0 = 0; rec[u].w1 = 0; rec[u].volt = 0; mar[u] = make_uint2(0, 0);
            if (k < n_own) {
               const size_t ri = sb + (size_t)(seg_first + k);
               gain[u] = gfire[(size_t)si * S + k]; rec[u] = crec[ri]; mar[u] = cmar[ri]; } }
         if (si < cs_seg0 || si >= cs_seg0 + cs_nseg) break;              // (not an entry its chain wrote: a stale one behind a full table)
         #pragma unroll
         for (int u = 0; u < kEsAhead; ++u) {
            if (k0 + 64 * u >= n_own) break;
            const int fired = gain[u] != 0.0f ? 1 : 0;
            const int incl = wave_incl_scan(fired, lane);
            if (fired) {
               uint16_t mb[4] = {(uint16_t)(mar[u].x & 0xffffu), (uint16_t)(mar[u].x >> 16), (uint16_t)(mar[u].y & 0xffffu), (uint16_t)(mar[u].y >> 16)};      // (entry j at end[-(j + 1)]: the block as it lies in memory)
               ev[at + (unsigned)(incl - 1)] = emit_event(cfg, P, wk, rec[u], mb + 4, msrc, gain[u], W, d, reset, trk, pidx, mv); }
            at += (unsigned)wave_last(incl); } } } }

// k_emit: the events k_gain's fast path only noted (the chains' heads and tails).  One workgroup per chain at a time, a lane per event.
__global__ void __launch_bounds__(256) k_emit(const DevCfg *__restrict__ cfgp, const rtfe_burst *__restrict__ bursts, const BurstScratch *__restrict__ scratch,
                                              const BurstCtl *__restrict__ ctl, const uint32_t *__restrict__ counts, rtfe_event *__restrict__ events,
                                              const float *__restrict__ chain_h, const CRec *__restrict__ crec, const uint2 *__restrict__ cmar, long long ccap,
                                              const unsigned char *__restrict__ pool, const ChainSt *__restrict__ cst, const int16_t *__restrict__ rows, long long nrows,
                                              const GsSeg *__restrict__ segs) {
   const DevCfg &cfg = *cfgp;
   const int ntrks = cfg.ntrks, nwalk = cfg.nparm * ntrks;
   const int nchains = chain_count(cfg, scratch->nbursts);
   const float mv = cfg.maxvolts;
   for (int cu = blockIdx.x; cu < nchains; cu += gridDim.x) {
      const ChainIx cx_ = chain_ix(cfg, cu);
      const int ci = cx_.ci, b = cx_.b, wi = cx_.wi, pidx = cx_.pidx, trk = cx_.trk;
      if (ctl[b].status != kBurstReady) continue;                        // (a burst the sample path redoes)
      const rtfe_burst B = bursts[b];
      const DevParm &P = cfg.parm[pidx];
      const int W = P.W, d = cfg.skew[trk], head = cfg.trk_to_head[trk];
      MarSrc msrc; msrc.rows = rows; msrc.nrows = nrows; msrc.ntrks = ntrks; msrc.head = head; msrc.sg = cfg.invert ? -1 : 1; msrc.W = W; msrc.nmar = cfg.pk_mar;
      const long long reset = ctl[b].reset;
      const unsigned int nev = counts[((size_t)b * cfg.nparm + pidx) * ntrks + trk];
      rtfe_event *ev = events + B.event_base + (size_t)(pidx * ntrks + trk) * B.event_cap;
      const size_t sbase = (size_t)(P.screen * ntrks + head) * ccap;
      const unsigned int skip0 = cst[ci].seg_ev0, skip1 = cst[ci].seg_ev1;      // (k_emit_seg's events)
      Walker wk = {};
      wk.v_avg_height = chain_h[(size_t)b * nwalk + wi];
      auto one = [&](const unsigned int i) {
         union { rtfe_event e; uint32_t w[4]; } in;
         in.e = ev[i];
         if (in.w[3] != 0xffffffffu) return;                              // the general step wrote it out in full
         const float gain = __uint_as_float(in.w[1]);
         wk.v_avg_height = __uint_as_float(in.w[2]);
         const CRec r = crec[sbase + in.w[0]];
         ev[i] = emit_event(cfg, P, wk, r, reinterpret_cast<const uint16_t *>(cmar + sbase + in.w[0] + 1), msrc, gain, W, d, reset, trk, pidx, mv); };
      if (cst[ci].pad == 1) {
         // a chain k_gain (mode 1) re-joined to its segments: its notes lie in front of, between and behind the standing segments' events (k_emit_seg's,
         // written beside this kernel: not to be looked at) - the gaps between them, in segment order
         const int seg0 = cst[ci].seg0, nseg = cst[ci].nseg;
         unsigned int pos = 0;
         for (int k = 0; k <= nseg; ++k) {
            unsigned int ge = nev;
            unsigned int next = nev;
            if (k < nseg) { const GsSeg &sg = segs[seg0 + k]; if (sg.stands != 1) continue; ge = sg.evoff < nev ? sg.evoff : nev; next = sg.evoff + (unsigned)sg.cnt; }
            for (unsigned int i = pos + threadIdx.x; i < ge; i += blockDim.x) one(i);
            pos = next < nev ? next : nev; }
         continue; }
      const unsigned int nmine = nev - (skip1 - skip0);
      for (unsigned int t = threadIdx.x; t < nmine; t += blockDim.x) one(t < skip0 ? t : t + (skip1 - skip0)); } }

// ------------------------------------------------------------------------------------------------
// k_dup_sets: a chain's events into the regions of the other parameter sets it stands for (chain_ix) - a wave per (burst, such set, track), 16 bytes a lane,
// the event's parameter-set byte its own.  Bursts the sample path redoes are left alone: k_decode walks every set of them.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_dup_sets(const DevCfg *__restrict__ cfgp, const rtfe_burst *__restrict__ bursts, const BurstScratch *__restrict__ scratch,
                                                  const BurstCtl *__restrict__ ctl, uint32_t *__restrict__ counts, rtfe_event *__restrict__ events) {
   const DevCfg &cfg = *cfgp;
   const int ntrks = cfg.ntrks, nparm = cfg.nparm, lane = threadIdx.x & 63;
   const long long nlists = (long long)scratch->nbursts * nparm * ntrks;
   for (long long li = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); li < nlists; li += (long long)gridDim.x * 4) {
      const int b = (int)(li / (nparm * ntrks));
      const int wi = (int)(li - (long long)b * nparm * ntrks), p = wi / ntrks, trk = wi - p * ntrks;
      const int rep = cfg.uset_rep[cfg.uset_of[p]];
      if (rep == p || ctl[b].status != kBurstDone) continue;      // (kBurstDone: k_publish's mark on a burst the chains finished)
      const rtfe_burst B = bursts[b];
      const uint32_t n = counts[((size_t)b * nparm + rep) * ntrks + trk];
      const uint4 *src = reinterpret_cast<const uint4 *>(events + B.event_base + (size_t)(rep * ntrks + trk) * B.event_cap);
      uint4 *dst = reinterpret_cast<uint4 *>(events + B.event_base + (size_t)(p * ntrks + trk) * B.event_cap);
      for (uint32_t i = (uint32_t)lane; i < n; i += 64) { uint4 e = src[i]; e.w = (e.w & 0x00ffffffu) | ((uint32_t)p << 24); dst[i] = e; }      // (rtfe_event: trk, flags, left_distance, parmset)
      if (lane == 0) counts[((size_t)b * nparm + p) * ntrks + trk] = n; } }

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
