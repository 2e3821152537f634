// rtfe_kernels.hip — the MI355X (gfx950 / CDNA4) analog front end.
//
// The sample path and its helpers, all integer / fp32 streaming work bound by HBM and LDS, no MFMA:
//
//   k_bursts   turns runs of quiet bits into inter-block zones -> the burst table (one workgroup).
//   k_decode   one lane per (parameter set, track) replays the reference's sequential detector EXACTLY (blind countdown,
//              stale-minimum rescans, AGC schedule, half-sample refinement) on samples in LDS, with a data-parallel
//              sliding-window candidate screen in front of it: PE, GCR, -differentiate, -zeros -differentiate, exact rescans,
//              and the bursts the record chains of the peak path (rtfe_sift.hip, rtfe_gain.hip) hand back.
//
// What is reproduced, and where it lives in the reference (LenShustek/readtape V3.18):
//   sample convert            src/readtape.c:1418-1421      volt()
//   deskew delay line         src/decoder.c:820-830         Tile::y()
//   staggered track start     src/decoder.c:855-861         Walker::start
//   lookfor_peak              src/decoder.c:751-810         slow_step() literal, fast path via screen
//   refine_peak               src/decoder.c:700-749         emit_peak()
//   process_*_transition      src/decoder.c:560-609         agc_after_peak()
//   adjust_agc                src/decoder.c:500-531         adjust_agc()
//   AGC schedule NRZI / GCR   src/decode_nrzi.c:196-197,218-229 ; src/decode_gcr.c:843-864
//   AGC schedule PE           src/decode_pe.c:127-155,175,198
// Compile with -ffp-contract=off: the reference is plain C99 on SSE2 (no FMA), and parity is bit-exact.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtfe_device.h"

namespace rtfe {

// ------------------------------------------------------------------------------------------------
// k_bursts: zones of >= gap_chunks quiet chunks (groups of 64 rows) -> burst table.  Single workgroup of 1024 threads.
// A zone END is a quiet chunk c whose successor is not quiet and whose gap_chunks predecessors
// (itself included) are all quiet; each thread tests the 64 chunks of one word per round.
// ------------------------------------------------------------------------------------------------
constexpr int kScratchBytes = 512;
struct BurstScratch {            // lives in the workspace (first kScratchBytes bytes)
   int   nbursts;                // bursts to decode (owned by this scan)
   int   queue;                  // next burst to decode
   int   nbursts_total;          // owned bursts + (time shards) the first burst of the halo, which only bounds the last owned one
   int   queue_walk;             // the dense path's chain queue (k_dchain)
   int   queue_resume;           // ... of the second k_decode pass
   int   queue_seg, nsegs;       // (unused)
   int   queue_stitch;           // (unused)
   int   seg_failed;             // bursts whose segments did not join (statistics)
   int   hard_count;             // candidates k_sift deferred to k_sift_hard (cleared with the scratch block when rtfe_scan starts)
   int   min_height_key;         // 0x7fffffff - the bits of the smallest v_avg_height any chain of this scan LEARNED (0: none did): what a caller may
                                 // raise rtfe_config::screen_floor_height towards for the tape's next scans (rtfe_scan_stats)
   float floor_used;             // the floor the scan's screen was built for (k_scan_begin: the handle's, or - its first scan - the estimate from the samples)
   int   prep_work;              // records on k_prep's work list (k_clear); zeroed with the block by k_scan_begin, not by k_bursts - that runs beside k_prep
   int   pad[3];
   unsigned long long dbg[8];    // at byte 64: optional per-phase cycle counters of k_decode (DevCfg::debug)
   unsigned long long pool_cursor;   // (unused)
   unsigned long long dbg2[8];   // dbg[8..15] (contiguous with dbg through pool_cursor is NOT assumed: indexed separately)
   unsigned long long why[8];    // optimistic-walk failure reasons (debug)
   unsigned long long scr[8];    // k_sift per-phase counters (RTFE_DEBUG=3)
};
static_assert(sizeof(BurstScratch) <= kScratchBytes, "scratch region too small");

// The candidate screen's thresholds for an assumed floor of the learned peak height (v_avg_height >= hfloor, agc_gain <= 2): the loosest thresholds a
// chain can ask for.  ONE definition for rtfe_create (host) and k_adapt_floor (device): src/decoder.c:785-786 scaled by (hfloor / 4) / 2.
__host__ __device__ inline void screen_thresholds(DevCfg &d, float hfloor) {
   const double lsb_per_volt = 32767.0 / (double)d.maxvolts;
   const float scale = (hfloor < 4.0f ? hfloor : 4.0f) / 4.0f / 2.0f;
   // (the new thresholds are made in locals and stored once: a kernel of another stream that reads the live configuration never sees a half-made
   //  screen - ADVICE r5; a handle's scans are still to be serialised on one stream, include/rt_frontend.h)
   int rise_i[kMaxScreens], minpk_i[kMaxScreens];
   for (int s = 0; s < kMaxScreens; ++s) { rise_i[s] = 1 << 30; minpk_i[s] = 1 << 30; }
   for (int p = 0; p < d.nparm; ++p) {
      DevParm &dp = d.parm[p];
      const float srv = dp.rise * scale, smv = dp.min_peak * scale;
      int ri = (int)floor((double)srv * lsb_per_volt * (1.0 - 1e-5)) - 2;
      const int mi = dp.min_peak > 0 ? (int)floor((double)smv * lsb_per_volt * (1.0 - 1e-5)) - 2 : -1;
      if (ri < -1) ri = -1;
      dp.screen_rise_v = srv;
      dp.screen_minpk_v = smv;
      if (ri < rise_i[dp.screen]) rise_i[dp.screen] = ri;
      if (mi < minpk_i[dp.screen]) minpk_i[dp.screen] = mi; }
   for (int s = 0; s < d.nscreens; ++s) { d.screen[s].rise_i = rise_i[s]; d.screen[s].minpk_i = minpk_i[s] < 0 ? -1 : minpk_i[s]; }
   d.floor_now = hfloor; }

// Behind a scan of the peak path: the chains have learned how high this tape's peaks are (BurstScratch::min_height_key); the next scan of the handle
// screens its candidates against half the smallest such height instead of the 1 V every tape clears - on a parameter sweep with low rise thresholds the
// default lets every noise wiggle through (NRZI -m: 15 ms of k_sift against 2.4).  Never wrong: a chain whose thresholds fall below the screen's is
// flagged RTFE_F_SCREEN_UNDERFLOW (update_thresholds reads the same fields) and rescanned exactly; a scan in which that happened moves the floor down again.
__global__ void k_adapt_floor(DevCfg *cfg, const BurstScratch *scratch) {
   if (threadIdx.x != 0 || blockIdx.x != 0 || !cfg->adapt_floor || cfg->differentiate || cfg->find_zeros) return;
   const int key = scratch->min_height_key;
   if (key <= 0) return;
   cfg->floor_probed = 1;                                   // (what the chains learned stands from here on: no estimate from the samples replaces it)
   float want = 0.5f * __uint_as_float((unsigned)(0x7fffffff - key));
   if (want > 4.0f) want = 4.0f;
   if (want < cfg->floor_cfg) want = cfg->floor_cfg;
   if (want == cfg->floor_now) return;
   screen_thresholds(*cfg, want); }

// rtfe_reset_floor: the handle's screen as rtfe_create left it (a new tape; bench.py's "first scan of a tape" lines)
__global__ void k_reset_floor(DevCfg *cfg) {
   if (threadIdx.x != 0 || blockIdx.x != 0) return;
   cfg->floor_probed = 0; cfg->probe_min = 0x7fffffff; cfg->probe_ticket = 0;
   if (cfg->floor_now != cfg->floor_cfg) screen_thresholds(*cfg, cfg->floor_cfg); }

// ------------------------------------------------------------------------------------------------
// k_scan_begin: the head of a peak-path scan, ONE launch for what used to be two memsets - the scratch block and the deferred candidates' per-list
// counts cleared - and, while the handle's candidate screen still stands at the floor it was made with (a tape's FIRST scan), an estimate of how high
// this tape's peaks are, from the samples themselves: the screen is then built for THAT before k_sift reads it.
//
// Why: the screen must pass every sample a chain could accept given v_avg_height >= floor (src/decoder.c:785-786, agc_gain <= 2).  At the 1 V every
// tape clears it lets noise wiggles of a low-rise parameter set through by the million (the reference's default -m on a clean tape: 24 ms instead of 6;
// a tape with 60 mV rms: 144 instead of 12).  Round 5 learned the floor BEHIND a scan (k_adapt_floor) - a tape's first scan paid the default.
// The estimate: kProbeWindows windows of kProbeRows rows spread evenly over the tape; a window counts if it lies inside a block (every 64-row group
// of it is outside the quiet band); per track of such a window the peak-to-peak range max - min, if the track shows both polarities; the smallest of
// all of them.  v_avg_height is the mean of eleven top-to-bottom heights at a block's start (src/decode_nrzi.c:218-229), the range of 25 bit cells is
// an upper bound of any single one: the floor is put at 0.45 x the estimate (k_adapt_floor behind the scan: 0.5 x the smallest height actually learned).
// Exactness does not rest on any of this: a chain whose thresholds fall below the screen's flags RTFE_F_SCREEN_UNDERFLOW as before.
// ------------------------------------------------------------------------------------------------
constexpr int kProbeRows = 512, kProbeWindows = 4096;
__global__ void __launch_bounds__(256) k_scan_begin(DevCfg *cfg, BurstScratch *scratch, uint4 *extra, long long extra_vecs, const int16_t *__restrict__ rows, long long nrows) {
   __shared__ int s_mx[RTFE_MAXTRKS], s_mn[RTFE_MAXTRKS];
   __shared__ unsigned int s_noisy;
   const int tid = threadIdx.x;
   for (long long v = (long long)blockIdx.x * 256 + tid; v < extra_vecs; v += (long long)gridDim.x * 256) extra[v] = make_uint4(0, 0, 0, 0);
   if (blockIdx.x == 0) for (int i = tid; i < kScratchBytes / 4; i += 256) reinterpret_cast<int *>(scratch)[i] = 0;
   // (the same answer in every workgroup of the launch: the flag is only written behind the last ticket)
   const bool probe = cfg->adapt_floor && !cfg->floor_probed && cfg->probe_on && !cfg->differentiate && !cfg->find_zeros && nrows >= 4 * kChunkRows;
   if (!probe) {
      __syncthreads();
      if (blockIdx.x == 0 && tid == 0) scratch->floor_used = cfg->floor_now;
      return; }
   const int ntrks = cfg->ntrks, q = cfg->quiet_i;
   const long long nwin = nrows / kProbeRows < 1 ? 1 : (nrows / kProbeRows > kProbeWindows ? kProbeWindows : nrows / kProbeRows);
   // a thread a (track, every nstr-th row) of the window: its own maximum, minimum and which 64-row groups it saw outside the quiet band, in registers;
   // three LDS atomics a thread at the end (round 6's first cut had every thread at the tracks' nine cells for every sample: 0.27 ms a scan)
   const int nstr = 256 / ntrks, trk = tid % ntrks, rg = tid / ntrks;
   for (long long w = blockIdx.x; w < nwin; w += gridDim.x) {
      const long long r0 = ((w * nrows) / nwin) & ~63ll;
      const int len = (int)(nrows - r0 < kProbeRows ? nrows - r0 : kProbeRows);
      if (tid < ntrks) { s_mx[tid] = -0x10000; s_mn[tid] = 0x10000; }
      if (tid == 0) s_noisy = 0;
      __syncthreads();
      if (rg < nstr) {
         int mx = -0x10000, mn = 0x10000;
         unsigned int loud = 0;
         const int16_t *col = rows + r0 * ntrks + trk;
         for (int i = rg; i < len; i += nstr) {
            const int x = col[(long long)i * ntrks];
            mx = x > mx ? x : mx; mn = x < mn ? x : mn;
            if (x > q || x < -q) loud |= 1u << (i >> 6); }
         atomicMax(&s_mx[trk], mx); atomicMin(&s_mn[trk], mn);
         if (loud) atomicOr(&s_noisy, loud); }
      __syncthreads();
      if (tid == 0) {
         const int ngroups = len >> 6;
         const unsigned all = (1u << ngroups) - 1u;
         if (ngroups >= 4 && (s_noisy & all) == all) {
            int est = 0x7fffffff;
            for (int t = 0; t < ntrks; ++t) if (s_mx[t] > q && s_mn[t] < -q && s_mx[t] - s_mn[t] < est) est = s_mx[t] - s_mn[t];
            if (est != 0x7fffffff) atomicMin(&cfg->probe_min, est); } }
      __syncthreads(); }
   __syncthreads();                                                    // (workgroup 0: the scratch block is cleared before its ticket says so)
   if (tid != 0) return;
   __threadfence();
   if (atomicAdd(&cfg->probe_ticket, 1u) != gridDim.x - 1) return;
   __threadfence();
   const int m = atomicMin(&cfg->probe_min, 0x7fffffff);              // (every workgroup's estimate is in)
   cfg->probe_min = 0x7fffffff; cfg->probe_ticket = 0;
   if (m != 0x7fffffff) {                                             // (else: no window inside a block - the next scan looks again)
      cfg->floor_probed = 1;
      float want = 0.45f * ((float)m / 32767.0f * cfg->maxvolts);
      if (want > 4.0f) want = 4.0f;
      if (want < cfg->floor_cfg) want = cfg->floor_cfg;
      if (want != cfg->floor_now) screen_thresholds(*cfg, want); }
   scratch->floor_used = cfg->floor_now; }

__device__ __forceinline__ bool quiet_at(const u64 *q, long long c, long long nchunks) {
   return c >= 0 && c < nchunks && ((q[c >> 6] >> (c & 63)) & 1); }

__device__ __forceinline__ int block_excl_scan_1024(int v, int *lds, int *total) {
   // exclusive scan of one int per thread over a 1024-thread block (16 waves)
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   int x = v;
   #pragma unroll
   for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(x, o); if (lane >= o) x += y; }
   if (lane == 63) lds[wave] = x;
   __syncthreads();
   if (threadIdx.x == 0) { int s = 0; for (int i = 0; i < 16; ++i) { int t = lds[i]; lds[i] = s; s += t; } lds[16] = s; }
   __syncthreads();
   const int r = x - v + lds[wave];
   *total = lds[16];
   __syncthreads();
   return r; }

// The zone search a round (4096 words of the quiet map = 2^18 chunks = 2^24 rows) at a time; a round's window goes through LDS with 64 words
// in front of it (a zone may begin there; further back the walk reads HBM): the tests are chains of dependent reads, a microsecond each
// from HBM.  One workgroup does all rounds in turn (k_bursts), or - long tapes - a workgroup per round in two passes (k_bursts_cnt: the
// rounds' counts; k_bursts_emit: the entries, each round behind the sum of the counts in front of it) and k_bursts_tail.
constexpr int kBrPer = 4;                                           // consecutive words per thread per round (fewer block scans)
constexpr int kBrWords = 1024 * kBrPer;
constexpr int kBrBack = 64, kBrWin = kBrBack + kBrWords + 1;
struct BrWin {
   const u64 *qwords; long long nwords, nchunks, wbase; const u64 *s_q;
   __device__ __forceinline__ u64 qword(long long w) const {
      if (w < 0 || w >= nwords) return 0;
      const long long k = w - wbase;
      return (k >= 0 && k < kBrWin) ? s_q[k] : qwords[w]; }
   __device__ __forceinline__ bool quiet(long long c) const { return c >= 0 && c < nchunks && ((qword(c >> 6) >> (c & 63)) & 1); } };
// the window of the round that begins at word w0 (all threads; barriers inside)
__device__ __forceinline__ BrWin br_window(const u64 *__restrict__ qwords, long long nwords, long long nchunks, long long w0, u64 *s_q) {
   BrWin bw; bw.qwords = qwords; bw.nwords = nwords; bw.nchunks = nchunks; bw.wbase = w0 - kBrBack; bw.s_q = s_q;
   __syncthreads();
   {  u64 t[(kBrWin + 1023) / 1024];                                // (all loads in flight before the first is waited for)
      #pragma unroll
      for (int j = 0; j < (kBrWin + 1023) / 1024; ++j) { const long long w = bw.wbase + j * 1024 + (int)threadIdx.x; t[j] = (w >= 0 && w < nwords) ? qwords[w] : 0; }
      #pragma unroll
      for (int j = 0; j < (kBrWin + 1023) / 1024; ++j) { const int k = j * 1024 + (int)threadIdx.x; if (k < kBrWin) s_q[k] = t[j]; } }
   __syncthreads();
   return bw; }
// a tape (or shard) that does not begin inside a qualifying zone gets an exact-start burst at row 0 (one thread, the window of round 0)
__device__ __forceinline__ bool br_exact_start(const BrWin &bw, int gap_chunks, int first_is_start, long long max_bursts) {
   bool starts_quiet = true;
   for (int c = 0; c < gap_chunks; ++c) if (!bw.quiet(c)) { starts_quiet = false; break; }
   return first_is_start && !starts_quiet && max_bursts > 0; }
// this thread's words of the round [w0, w0 + wpr): the zone ends in them (a bit per chunk); returns their number
__device__ __forceinline__ int br_ends(const BrWin &bw, long long w0, int wpr, int gap_chunks, u64 (&ends)[kBrPer]) {
   int cnt = 0;
   #pragma unroll
   for (int j = 0; j < kBrPer; ++j) {
      const int wi = (int)threadIdx.x * kBrPer + j;
      const long long w = w0 + wi;
      ends[j] = 0;
      if (wi < wpr && w < bw.nwords) {
         const u64 q = bw.s_q[kBrBack + wi];                          // (the round's own words, straight from the window: words behind the map read 0)
         const u64 qn = bw.s_q[kBrBack + wi + 1];
         const u64 next = (q >> 1) | (qn << 63);                    // bit c = quiet[c+1]
         u64 cand = q & ~next;                                      // quiet and successor not quiet
         const u64 qp = bw.s_q[kBrBack + wi - 1];                     // (word -1 of the map reads 0: not quiet)
         while (cand) {
            const int bit = __ffsll((long long)cand) - 1;
            cand &= cand - 1;
            bool ok = true;
            if (gap_chunks <= 64) {
               // chunks c, c - 1, ... from bit 63 downwards: the gap_chunks uppermost must all be quiet
               const u64 v = bit == 63 ? q : ((q << (63 - bit)) | (qp >> (bit + 1)));
               ok = ((~v) >> (64 - gap_chunks)) == 0; }
            else {
               const long long c = w * 64 + bit;
               for (int k = 1; k < gap_chunks; ++k) if (!bw.quiet(c - k)) { ok = false; break; } }
            if (ok) ends[j] |= 1ull << bit; } }
      cnt += __popcll(ends[j]); }
   return cnt; }
// ... and their table entries, from index idx0 on
__device__ __forceinline__ void br_emit(const BrWin &bw, long long w0, const u64 (&ends)[kBrPer], long long idx0, long long nrows, rtfe_burst *__restrict__ bursts, long long max_bursts) {
   long long idx = idx0;
   #pragma unroll
   for (int j = 0; j < kBrPer; ++j) {
      const long long w = w0 + (long long)threadIdx.x * kBrPer + j;
      u64 e = ends[j];
      while (e) {
         const int bit = __ffsll((long long)e) - 1;
         e &= e - 1;
         const long long c1 = w * 64 + bit + 1;                     // one past the last quiet chunk
         // zone start: walk back over quiet chunks a 64-bit word at a time
         long long c0 = c1 - 1;
         for (;;) {
            if (c0 == 0) break;
            const long long pw = (c0 - 1) >> 6; const int pb = (int)((c0 - 1) & 63);
            // bits pb..0 of word pw, shifted so that bit pb becomes bit 63: count the leading run of ones
            const u64 run = ~(bw.qword(pw) << (63 - pb));
            const int ones = run ? __clzll((long long)run) : 64;
            const int take = ones < pb + 1 ? ones : pb + 1;
            c0 -= take;
            if (take < pb + 1) break; }
         if (idx < max_bursts) {
            rtfe_burst b = {};
            long long zf = c0 * kChunkRows;                          // first row of the zone (chunks are groups of 64 rows)
            long long ze = c1 * kChunkRows;                          // one past its last row
            if (ze > nrows) ze = nrows & ~63ll;
            b.zone_first = zf; b.zone_end = ze; b.reset_sample = -1; b.safe_last = -1;
            bursts[idx] = b; }
         ++idx; } } }
// behind the search: which bursts this scan owns (time shards), their coarse extents, event capacities and region bases (all threads of one workgroup)
__device__ __forceinline__ void br_tail(int nb_found, long long nrows, long long own_rows, int ntrks, int gap_chunks, float cap_frac, int nparm, long long event_capacity,
                                        rtfe_burst *__restrict__ bursts, long long max_bursts, BurstScratch *__restrict__ scratch, int32_t *__restrict__ nbursts_out, int *lds) {
   __shared__ u64 s_ebase;
   __shared__ int s_owned;
   int nb = nb_found;
   if (nb > max_bursts) nb = (int)max_bursts;
   // time shards: keep the bursts that start in the owned rows, plus one more as the bound of the last of them
   if (threadIdx.x == 0) s_owned = nb;
   __syncthreads();
   for (int b0 = 0; b0 < nb; b0 += 1024) {
      const int b = b0 + threadIdx.x;
      // a zone that ends fewer than gap_chunks chunks behind the seam cannot qualify in the right neighbour's slice (it sees
      // fewer than gap_chunks of its chunks), so it belongs here; one that ends later qualifies there and belongs there
      if (b < nb && own_rows < nrows && bursts[b].zone_end >= own_rows + (long long)gap_chunks * kChunkRows && !(bursts[b].flags & RTFE_F_EXACT_START)) atomicMin(&s_owned, b); }
   __syncthreads();
   const int n_owned = s_owned;
   if (nb > n_owned + 1) nb = n_owned + 1;
   // ---- coarse extents, event capacities and region bases (parallel prefix sum) ----
   if (threadIdx.x == 0) s_ebase = 0;
   __syncthreads();
   for (int b0 = 0; b0 < nb; b0 += 1024) {
      const int b = b0 + threadIdx.x;
      long long cap = 0;
      if (b < nb) {
         const long long start = (bursts[b].flags & RTFE_F_EXACT_START) ? 0 : bursts[b].zone_end - kMarginRows;
         const long long end = (b + 1 < nb) ? bursts[b + 1].zone_end : nrows;
         long long len = end - start;
         if (len < 0) len = 0;
         cap = (long long)((float)len * cap_frac) + 64;
         bursts[b].end_sample = end;             // provisional: the decode kernel replaces it by the next reset
         bursts[b].event_cap = (uint32_t)cap; }
      // 64-bit scan done as two 32-bit scans would overflow; regions are < 2^31 events each, so scan in units of 64 events
      int total;
      const int units = (int)((cap * nparm * ntrks + 63) >> 6);
      const int eoff = block_excl_scan_1024(units, lds, &total);
      if (b < nb) {
         const u64 base = s_ebase + ((u64)eoff << 6);
         if ((long long)(base + (u64)cap * nparm * ntrks) > event_capacity) {
            bursts[b].event_cap = 0; bursts[b].flags |= RTFE_F_EVENT_OVERFLOW; }
         bursts[b].event_base = base; }
      __syncthreads();
      if (threadIdx.x == 0) s_ebase += (u64)total << 6;
      __syncthreads(); }
   if (threadIdx.x == 0) {
      scratch->nbursts = n_owned; scratch->nbursts_total = nb; scratch->queue = 0; scratch->queue_walk = 0; scratch->queue_resume = 0; scratch->queue_seg = 0; scratch->nsegs = 0; scratch->queue_stitch = 0; scratch->seg_failed = 0; *nbursts_out = n_owned;
      if (n_owned == nb && own_rows < nrows && n_owned > 0) bursts[n_owned - 1].flags |= RTFE_F_TRUNCATED; }
   if (threadIdx.x < 8) scratch->dbg[threadIdx.x] = 0;
   if (threadIdx.x >= 16 && threadIdx.x < 24) scratch->dbg2[threadIdx.x - 16] = 0;
   if (threadIdx.x >= 24 && threadIdx.x < 32) scratch->why[threadIdx.x - 24] = 0; }

__global__ void __launch_bounds__(1024) k_bursts(const u64 *__restrict__ qwords, long long nwords, long long nchunks,
                                                 long long nrows, long long own_rows, int ntrks, int gap_chunks, int first_is_start,
                                                 float cap_frac, int nparm, long long event_capacity,
                                                 rtfe_burst *__restrict__ bursts, long long max_bursts,
                                                 BurstScratch *__restrict__ scratch, int32_t *__restrict__ nbursts_out, int prof) {
   __shared__ int lds[32];
   long long pk0 = prof ? clock64() : 0, pk1 = 0, pk2 = 0;
   __shared__ int s_base;
   __shared__ u64 s_q[kBrWin];
   if (threadIdx.x == 0) s_base = 0;
   for (long long w0 = 0; w0 < nwords; w0 += kBrWords) {
      const BrWin bw = br_window(qwords, nwords, nchunks, w0, s_q);
      if (w0 == 0 && threadIdx.x == 0 && br_exact_start(bw, gap_chunks, first_is_start, max_bursts)) {
         rtfe_burst b = {};
         b.zone_first = 0; b.zone_end = 0; b.reset_sample = 0; b.safe_last = 0; b.flags = RTFE_F_EXACT_START;
         bursts[0] = b;
         s_base = 1; }
      __syncthreads();
      u64 ends[kBrPer];
      const int cnt = br_ends(bw, w0, kBrWords, gap_chunks, ends);
      int total;
      const int off = block_excl_scan_1024(cnt, lds, &total);
      const int base = s_base;
      br_emit(bw, w0, ends, (long long)base + off, nrows, bursts, max_bursts);
      __syncthreads();
      if (threadIdx.x == 0) s_base = base + total; }
   __syncthreads();
   if (prof) pk1 = clock64();
   br_tail(s_base, nrows, own_rows, ntrks, gap_chunks, cap_frac, nparm, event_capacity, bursts, max_bursts, scratch, nbursts_out, lds);
   if (prof && threadIdx.x == 0) { pk2 = clock64(); scratch->scr[0] = (unsigned long long)(pk1 - pk0); scratch->scr[1] = (unsigned long long)(pk2 - pk1); }      // (RTFE_DEBUG=5)
   }       // (scratch->scr: cleared with the rest of the scratch block when rtfe_scan starts)

// The same search with a workgroup per round of wpr words (4096; fewer in tests).  k_bursts_cnt: rtot[r] = the zone ends of round r (+ the
// exact-start burst in round 0); k_bursts_emit: round r's entries from index sum(rtot[0 .. r)) on; k_bursts_tail: the rest.
__global__ void __launch_bounds__(1024) k_bursts_cnt(const u64 *__restrict__ qwords, long long nwords, long long nchunks, int wpr, int gap_chunks, int first_is_start,
                                                     long long max_bursts, uint32_t *__restrict__ rtot) {
   __shared__ int lds[32];
   __shared__ u64 s_q[kBrWin];
   const long long w0 = (long long)blockIdx.x * wpr;
   const BrWin bw = br_window(qwords, nwords, nchunks, w0, s_q);
   u64 ends[kBrPer];
   const int cnt = br_ends(bw, w0, wpr, gap_chunks, ends);
   int total;
   (void)block_excl_scan_1024(cnt, lds, &total);
   if (threadIdx.x == 0) rtot[blockIdx.x] = (uint32_t)total + ((blockIdx.x == 0 && br_exact_start(bw, gap_chunks, first_is_start, max_bursts)) ? 1u : 0u); }
__global__ void __launch_bounds__(1024) k_bursts_emit(const u64 *__restrict__ qwords, long long nwords, long long nchunks, long long nrows, int wpr, int gap_chunks, int first_is_start,
                                                      rtfe_burst *__restrict__ bursts, long long max_bursts, const uint32_t *__restrict__ rtot) {
   __shared__ int lds[32];
   __shared__ u64 s_q[kBrWin];
   const long long w0 = (long long)blockIdx.x * wpr;
   // the entries in front of this round's: the counts of the rounds before it (a sum over at most a few thousand numbers)
   long long base;
   {  long long mine = 0;
      for (int r = threadIdx.x; r < (int)blockIdx.x; r += 1024) mine += rtot[r];
      int total;
      (void)block_excl_scan_1024((int)(mine > 0x3fffffll ? 0x3fffffll : mine), lds, &total);      // (saturates far beyond any table: nothing is written at or behind max_bursts)
      base = total; }
   const BrWin bw = br_window(qwords, nwords, nchunks, w0, s_q);
   if (blockIdx.x == 0 && br_exact_start(bw, gap_chunks, first_is_start, max_bursts)) {      // (every thread decides alike)
      if (threadIdx.x == 0) {
         rtfe_burst b = {};
         b.zone_first = 0; b.zone_end = 0; b.reset_sample = 0; b.safe_last = 0; b.flags = RTFE_F_EXACT_START;
         bursts[0] = b; }
      base = 1; }
   u64 ends[kBrPer];
   const int cnt = br_ends(bw, w0, wpr, gap_chunks, ends);
   int total;
   const int off = block_excl_scan_1024(cnt, lds, &total);
   br_emit(bw, w0, ends, base + off, nrows, bursts, max_bursts); }
__global__ void __launch_bounds__(1024) k_bursts_tail(int nrounds, const uint32_t *__restrict__ rtot, long long nrows, long long own_rows, int ntrks, int gap_chunks,
                                                      float cap_frac, int nparm, long long event_capacity, rtfe_burst *__restrict__ bursts, long long max_bursts,
                                                      BurstScratch *__restrict__ scratch, int32_t *__restrict__ nbursts_out) {
   __shared__ int lds[32];
   long long mine = 0;
   for (int r = threadIdx.x; r < nrounds; r += 1024) mine += rtot[r];
   int total;
   (void)block_excl_scan_1024((int)(mine > 0x3fffffll ? 0x3fffffll : mine), lds, &total);      // (1024 x 2^22 stays inside an int; the table is shorter than that anyway)
   br_tail(total, nrows, own_rows, ntrks, gap_chunks, cap_frac, nparm, event_capacity, bursts, max_bursts, scratch, nbursts_out, lds); }

// ------------------------------------------------------------------------------------------------
// k_decode
// ------------------------------------------------------------------------------------------------
struct Tile {
   int16_t *x;            // LDS: [halo + tile rows][ntrks] the tape's rows as they are (head order, after -invert): a flat copy
   int      halo;         // rows kept in front of the tile (DevCfg::halo_rows)
   int      ldw;          // (unused)
   const int *colof;      // track -> head column (DevCfg::trk_to_head)
   long long row0;        // first row of the tile proper
   int      nrows;        // rows in the tile proper
   long long reset;       // burst restart row (deskew FIFO restarts there, src/decoder.c:415)
   // the bitmaps and left_distance maps also cover the kScreenHalo rows in front of the tile (word -1 / rows -64..-1)
   unsigned char *bits;   // LDS: [nscreens][5][ntrks][bstride]   0=top 1=bot 2=rescan ("A-sync") 3/4=run starts (k_screen)
   int      bstride;      // bytes per bitmap row = (tile_rows + kScreenHalo) / 8 + 8 (one spare word)
   unsigned char *ldpos;  // LDS: [nscreens][2][ntrks][ldstride] left_distance of the first window max (0) / min (1)
   int      ldstride;     // tile_rows + kScreenHalo
   int      ntrks;
   const int *skew;
   float   *fd;           // LDS (-differentiate peak path only): differentiate()'s output for every element of x
   __device__ __forceinline__ float fy(int t, long long n) const {       // the detector's input at row n: differentiated, then deskewed
      const int d = skew[t];
      const long long m = (n - reset < d) ? n : n - d;
      return fd[((int)(m - row0) + halo) * ntrks + colof[t]]; }
   __device__ __forceinline__ int xi(int t, long long n) const { return x[((int)(n - row0) + halo) * ntrks + colof[t]]; }
   // v_now of track t at row n in int16 units, with the deskew FIFO exactly as the reference runs it
   // from the restart row: undelayed until the FIFO has filled (src/decoder.c:825-827), then delayed
   __device__ __forceinline__ int y(int t, long long n) const {
      const int d = skew[t];
      return xi(t, (n - reset < d) ? n : n - d); }
   __device__ __forceinline__ unsigned char *ldmap(int screen, int kind, int t) const {       // [row], rows >= -kScreenHalo
      return ldpos + (screen * 2 + kind) * (ntrks * ldstride) + t * ldstride + kScreenHalo; }    // (32-bit LDS offsets; ntrks * stride is loop invariant)
   __device__ __forceinline__ const u64 *map(int screen, int kind, int t) const {               // [word], words >= -1
      return reinterpret_cast<const u64 *>(bits + (screen * 5 + kind) * (ntrks * bstride) + t * bstride + kScreenHalo / 8); }
};

// one track's column of the sample tile, indexed by tile-relative row (after the deskew delay)
struct Col {
   const int16_t *p; int P;
   __device__ __forceinline__ int operator[](int i) const { return p[i * P]; } };
__device__ __forceinline__ Col tile_col(const Tile &tl, int trk, int delay) {
   Col c; c.P = tl.ntrks; c.p = tl.x + (tl.halo - delay) * tl.ntrks + tl.colof[trk]; return c; }
// the same column where the caller knows that the tile's rows lie in HBM (k_zeros reads the tape in place): loads through a generic
// pointer are FLAT instructions - they count against both the vector-memory and the LDS counter and may return out of order, so every
// use waits for ALL of them and nothing can be kept in flight across a batch of detector steps; through a global pointer they are
// global_load with in-order returns (s_waitcnt vmcnt(n))
#ifdef RTFE_CPU_EMUL
typedef const int16_t *gptr16;
#else
typedef const __attribute__((address_space(1))) int16_t *gptr16;
#endif
struct GCol {             // a uniform base pointer (the tile in HBM) + a 32-bit element offset per lane: global_load with a scalar base, one add per load
   gptr16 base; int off, P;
   __device__ __forceinline__ int operator[](int i) const {
#ifdef RTFE_CPU_EMUL
      return base[off + i * P];
#else
      // (a 32-bit unsigned BYTE offset from the uniform base: the address mode "scalar base + vector offset" of global_load)
      return *(gptr16)((const __attribute__((address_space(1))) char *)base + (unsigned)((off + i * P) << 1));
#endif
   } };
template <bool kGlobal> __device__ __forceinline__ auto zc_col(const Tile &tl, int trk, int delay) {
   if constexpr (kGlobal) { GCol g; g.base = (gptr16)tl.x; g.off = (tl.halo - delay) * tl.ntrks + tl.colof[trk]; g.P = tl.ntrks; return g; }
   else return tile_col(tl, trk, delay); }

__device__ __forceinline__ float volt(int i, float maxvolts) {      // src/readtape.c:1420
   return (float)i / 32767 * maxvolts; }

struct Walker {            // one per (parameter set, track); lives in registers
   // detector state
   long long start;        // row at which this track's window is seeded (restart + trknum, src/decoder.c:855-861)
   long long next;         // next row the detector will look at with the fast path
   long long blind_until;  // rows <= blind_until are inside pkww_countdown
   int   minv;             // the (possibly stale) window minimum, int16 units (src/decoder.c:765-775)
   long long qtrig;        // row at which the sample equal to minv leaves the window (forces a rescan)
   long long cpos;         // the stale-min state is valid after processing this row
   bool  chain_pending;    // ... except that the forced rescan AT row cpos has not been carried out yet (lazy)
   int   slow_max, slow_countdown;   // literal state while the window is filling
   bool  fast;             // window full and in the regular deskew regime: use the screen
   long long trust_from;   // first row from which k_screen's view of this track (regular regime, full window) is the detector's
   // AGC / block-decoder mirror
   float agc_gain, v_avg_height, v_avg_height_sum;
   int   v_avg_height_count, peakcount, heightndx;
   float v_top, v_bot, v_lasttop, v_lastbot;
   // zero-crossing detector (-zeros), int16 codes (src/decoder.c:617-649)
   int   z_prev, z_top, z_bot;
   float zf_top, zf_bot, zf_lastraw;     // differentiated variant (src/decoder.c:654-683): volts
   long long z_firstzero, z_lastzero;   // rows of the first / last exact zero since the last arming (-1: none)
   bool  z_up_pending, z_dn_pending;
   long long z_ttop_row, z_tbot_row;
   // PE preamble tracking
   bool  datablock, bit1_up;
   double t_lastpeak;
   // thresholds, recomputed whenever the AGC state changes (src/decoder.c:785-786)
   float rise, reqmin;
   bool  thr_dirty;            // rise/reqmin (exact floats) are stale: only the integer bands were refreshed (optimistic walk)
   int   rise_lo, rise_hi;     // integer guard bands around rise / reqmin (int16 units): a difference <= lo fails
   int   min_lo, min_hi;       //   the float test for sure, >= hi passes for sure, in between the float test decides
   // output
   unsigned int nevents;
   unsigned int flags;
};


// A detection whose (cheap-to-defer) half-sample refinement, volt conversion and event store are done
// after the walk by all lanes (finalize_tile): the sequential walker keeps only what feeds back.
struct alignas(8) Rec { unsigned int idx; unsigned short n_rel; unsigned char ld, kind; float g; short val, prev, next, pad; int pad2; };   // 24 bytes


struct Ctx {               // per-workgroup constants for the walkers
   const DevCfg *cfg;
   Tile tile;
   long long row_base;     // absolute row of d_rows[0]
   float *heights;         // LDS [walkers][10]
   rtfe_event *events;     // this burst's regions
   unsigned int cap;
   struct Rec *recs;       // LDS [rec_cap] deferred events of this lane's walker for the current tile
   int     rec_cap;
   int     nrec;           // records queued in this tile
};

__device__ __forceinline__ double time_of(const DevCfg *c, long long abs_row) {     // src/readtape.c:1423
   return (double)(c->tstart_ns + abs_row * c->tdelta_ns) / 1e9; }

__device__ __forceinline__ void adjust_agc(Walker &w, const DevParm &P, float *heights) {    // src/decoder.c:500-531
   float gain, lastheight;
   if (P.agc_alpha) {
      lastheight = w.v_lasttop - w.v_lastbot;
      if (lastheight > 0) {
         gain = w.v_avg_height / lastheight;
         gain = P.agc_alpha * gain + (1 - P.agc_alpha) * w.agc_gain;
         if (gain > 2.0f) gain = 2.0f;
         w.agc_gain = gain; } }
   if (P.agc_window) {
      lastheight = w.v_lasttop - w.v_lastbot;
      if (lastheight > 0) {
         heights[w.heightndx] = lastheight;
         if (++w.heightndx >= P.agc_window) w.heightndx = 0;
         float minheight = 99;
         for (int i = 0; i < P.agc_window; ++i) if (heights[i] < minheight) minheight = heights[i];
         gain = w.v_avg_height / minheight;
         if (gain > 2.0f) gain = 2.0f;
         w.agc_gain = gain; } } }

// what the block decoder's callback does to the state the detector reads back, then the
// post-callback bookkeeping of process_up/down_transition (src/decoder.c:587-590, 605-609)
__device__ __forceinline__ void agc_after_peak_m(Walker &w, int mode, int agc_off, const DevParm &P, float *heights, bool is_top, double t_peak);
__device__ __forceinline__ void agc_after_peak(Walker &w, const DevCfg *cfg, const DevParm &P, float *heights, bool is_top, double t_peak) {
   agc_after_peak_m(w, cfg->mode, cfg->agc_off, P, heights, is_top, t_peak); }
// (mode and agc_off by value: a caller that keeps them in registers does not go back to the configuration block in HBM per detection)
__device__ __forceinline__ void agc_after_peak_m(Walker &w, int mode, int agc_off, const DevParm &P, float *heights, bool is_top, double t_peak) {
   ++w.peakcount;                                               // src/decoder.c:561
   if (agc_off) { }                                              // density detection: no decoder callback (src/decoder.c:578-581)
   else if (mode == RTFE_PE) {
      if (w.datablock) adjust_agc(w, P, heights);               // src/decode_pe.c:175,198
      else {                                                    // pe_preamble_peak, src/decode_pe.c:127-155
         if (w.peakcount == 1) w.bit1_up = !is_top;
         if (w.peakcount > 70 && w.bit1_up == is_top && t_peak - w.t_lastpeak > P.t_clkwindow) {
            w.datablock = true;
            w.v_avg_height = w.v_avg_height_sum / w.v_avg_height_count; }
         else if (w.peakcount >= 5 && w.peakcount <= 15 && w.v_top > w.v_bot) {
            w.v_avg_height_sum += w.v_top - w.v_bot;
            ++w.v_avg_height_count;
            heights[w.heightndx] = w.v_top - w.v_bot;
            if (++w.heightndx >= P.agc_window) w.heightndx = 0; } } }
   else if (mode == RTFE_WW) adjust_agc(w, P, heights);           // src/decode_ww.c:171,190
   else {                                                       // NRZI and GCR share the schedule
      if (is_top) {                                             // src/decode_nrzi.c:218-229, src/decode_gcr.c:853-864
         if (w.peakcount >= 5 && w.peakcount <= 15) {
            w.v_avg_height_sum += w.v_top - w.v_bot;
            ++w.v_avg_height_count;
            heights[w.heightndx] = w.v_top - w.v_bot;
            if (++w.heightndx >= P.agc_window) w.heightndx = 0; }
         else if (w.peakcount > 15) {
            if (w.v_avg_height_count) {
               w.v_avg_height = w.v_avg_height_sum / w.v_avg_height_count;
               w.v_avg_height_count = 0; }
            else adjust_agc(w, P, heights); } }
      else if (w.peakcount > 15 && w.v_avg_height_count == 0) adjust_agc(w, P, heights); }   // src/decode_nrzi.c:196-197
   if (is_top) w.v_lasttop = w.v_top; else w.v_lastbot = w.v_bot;
   w.t_lastpeak = t_peak; }

__device__ __forceinline__ void update_thresholds(Walker &w, const DevParm &P, float lsb_per_volt) {
   w.rise = P.rise * (w.v_avg_height / 4.0f) / w.agc_gain;       // src/decoder.c:785-786
   w.reqmin = P.min_peak * (w.v_avg_height / 4.0f) / w.agc_gain;
   if (w.rise < P.screen_rise_v || (P.min_peak != 0 && w.reqmin < P.screen_minpk_v)) w.flags |= RTFE_F_SCREEN_UNDERFLOW;
   // |(vM - vL) - (M - L) * volts_per_lsb| < 0.02 lsb for |v| <= maxvolts (three fp32 roundings), so one lsb
   // of guard on either side of the converted threshold is ample
   const int r = (int)floorf(w.rise * lsb_per_volt);
   w.rise_lo = r - 1; w.rise_hi = r + 2;
   const int m = (int)floorf(w.reqmin * lsb_per_volt);
   w.min_lo = m - 1; w.min_hi = m + 2;
   w.thr_dirty = false; }

#ifdef RTFE_CPU_EMUL
static inline float fast_rcp(float x) { return 1.0f / x; }
#else
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#endif
// the optimistic walk only needs the integer bands: thresholds from a 1-ulp reciprocal instead of two IEEE divisions,
// with one more lsb of guard on either side (the approximation moves the converted threshold by << 0.01 lsb).
// Returns false when the threshold is too close to the candidate screen's to rule out an underflow here.
__device__ __forceinline__ bool approx_thresholds(Walker &w, const DevParm &P, float lsb_per_volt) {
   const float s = w.v_avg_height * 0.25f * fast_rcp(w.agc_gain);
   const float ra = P.rise * s, ma = P.min_peak * s;
   if (ra < P.screen_rise_v * 1.001f || (P.min_peak != 0 && ma < P.screen_minpk_v * 1.001f)) return false;
   const int r = (int)(ra * lsb_per_volt), m = (int)(ma * lsb_per_volt);
   w.rise_lo = r - 2; w.rise_hi = r + 3;
   w.min_lo = m - 2; w.min_hi = m + 3;
   w.reqmin = P.min_peak == 0 ? 0.0f : 1.0f;                       // only "is there a min_peak test" is read before the next exact refresh
   w.thr_dirty = true;
   return true; }


// 1 = passes, 0 = fails: "v(a) > v(b) + thr" decided on the int16 codes when clear, else in floats
__device__ __forceinline__ bool above_by(int a, int b, float thr, int lo, int hi, float mv) {
   const int d = a - b;
   if (d >= hi) return true;
   if (d <= lo) return false;
   return volt(a, mv) > volt(b, mv) + thr; }
__device__ __forceinline__ bool below_by(int a, int b, float thr, int lo, int hi, float mv) {   // v(a) < v(b) - thr
   const int d = b - a;
   if (d >= hi) return true;
   if (d <= lo) return false;
   return volt(a, mv) < volt(b, mv) - thr; }


// the half-sample refinement of refine_peak (src/decoder.c:712-731): 0 none, 1 = -0.5, 2 = +0.5
__device__ __forceinline__ int refine_code(const DevCfg *cfg, int val_i, int iprev, int inext, float agc_gain, bool is_top) {
   const float mv = cfg->maxvolts;
   const float val = volt(val_i, mv);
   const float thr = 0.005f / agc_gain;                            // PEAK_THRESHOLD / agc_gain (src/decoder.c:715,724)
   const int ti = (int)floorf(thr * cfg->lsb_per_volt);
   // "close" = within thr of the extreme, "far" = beyond it; decided on the codes unless within a guard band
   const int dp = is_top ? val_i - iprev : iprev - val_i, dn = is_top ? val_i - inext : inext - val_i;
   bool pclose, pfar, nclose, nfar;
   if (dp <= ti - 1) { pclose = true; pfar = false; } else if (dp >= ti + 2) { pclose = false; pfar = true; }
   else { const float lim_v = is_top ? val - thr : val + thr, vp = volt(iprev, mv);
          pclose = is_top ? vp > lim_v : vp < lim_v; pfar = is_top ? vp < lim_v : vp > lim_v; }
   if (dn <= ti - 1) { nclose = true; nfar = false; } else if (dn >= ti + 2) { nclose = false; nfar = true; }
   else { const float lim_v = is_top ? val - thr : val + thr, vn = volt(inext, mv);
          nclose = is_top ? vn > lim_v : vn < lim_v; nfar = is_top ? vn < lim_v : vn > lim_v; }
   if (pclose && nfar) return 1;                                   // src/decoder.c:716-721, 725-730
   if (nclose && pfar) return 2;
   return 0; }

__device__ __forceinline__ void store_event(const Ctx &cx, int pidx, int trk, unsigned int idx, long long n, float val, float g,
                                   bool is_top, int adjcode, int left_distance) {
   rtfe_event e;
   e.sample = (uint32_t)(n - cx.tile.reset);
   // -invert negates the VOLTAGE (src/readtape.c:1421): a zero sample is -0.0f there (the differentiator's dead band makes it +0 again)
   e.v_peak = (cx.cfg->invert && !cx.cfg->differentiate && val == 0.0f) ? -0.0f : val;
   e.agc_gain = g;
   e.trk = (uint8_t)trk;
   e.flags = (uint8_t)((is_top ? 0 : 1) | (adjcode << 1));
   e.left_distance = (uint8_t)left_distance;
   e.parmset = (uint8_t)pidx;
   cx.events[(size_t)(pidx * cx.cfg->ntrks + trk) * cx.cap + idx] = e; }

// src/decoder.c:782: the reference asserts agc_gain > 0 whenever it looks at the window's shape, i.e. from the first row behind the
// blind countdown on - and the assert is fatal for the whole run.  The track's event list gets a marker at that row (RTFE_EV_FATAL);
// the host replay delivers everything that comes before it in (row, track) order and then stops as the reference does.
__device__ __forceinline__ bool agc_fatal(Walker &w, Ctx &cx, int pidx, int trk, long long death_row) {
   if (w.agc_gain > 0) return false;
   w.flags |= RTFE_F_AGC_FATAL;
   if (w.nevents < cx.cap) {
      rtfe_event e = {};
      e.sample = (uint32_t)(death_row - cx.tile.reset); e.trk = (uint8_t)trk; e.flags = RTFE_EV_FATAL; e.parmset = (uint8_t)pidx;
      cx.events[(size_t)(pidx * cx.cfg->ntrks + trk) * cx.cap + w.nevents] = e; }
   else w.flags |= RTFE_F_EVENT_OVERFLOW;
   ++w.nevents;
   return true; }

// refine_peak (src/decoder.c:700-749) + event emission + AGC mirror.  `lo` = first row of the window,
// `p` = row of the first window element equal to the extreme.  With defer != 0 the refinement and the
// event store are queued for finalize_tile() (possible whenever nothing downstream needs the peak time).
__device__ __forceinline__ void emit_peak(Walker &w, Ctx &cx, int pidx, int trk, const DevParm &P, long long n,
                                 long long lo, long long p, float val, int val_i, bool is_top, bool defer, const short *nbr = nullptr) {
   const DevCfg *cfg = cx.cfg;
   const int left_distance = (int)(p - lo) + 1;
   // PE decides the end of the preamble from peak times (src/decode_pe.c:136-138): only then is the time needed here
   const bool need_time = cfg->mode == RTFE_PE && !w.datablock && w.peakcount >= 68;
   double t_peak = 0;
   if (w.nevents >= cx.cap) w.flags |= RTFE_F_EVENT_OVERFLOW;
   else if (defer && !need_time && cx.nrec < cx.rec_cap) {
      Rec r; r.idx = w.nevents; r.n_rel = (unsigned short)(n - cx.tile.row0); r.ld = (unsigned char)left_distance;
      r.kind = is_top ? 0 : 1; r.g = w.agc_gain; r.val = (short)val_i; r.pad = 0; r.pad2 = 0;
      if (nbr) { r.prev = nbr[0]; r.next = nbr[1]; } else { r.prev = (short)cx.tile.y(trk, p - 1); r.next = (short)cx.tile.y(trk, p + 1); }
      cx.recs[cx.nrec++] = r; }
   else {
      const int adjcode = refine_code(cfg, val_i, cx.tile.y(trk, p - 1), cx.tile.y(trk, p + 1), w.agc_gain, is_top);
      const float adj = adjcode == 1 ? -0.5f : (adjcode == 2 ? 0.5f : 0.0f);
      if (cfg->mode == RTFE_PE)
         t_peak = time_of(cfg, cx.row_base + n) - ((float)(P.W - left_distance) - adj) * cfg->sample_deltat;
      store_event(cx, pidx, trk, w.nevents, n, val, w.agc_gain, is_top, adjcode, left_distance); }
   if (is_top) w.v_top = val; else w.v_bot = val;
   ++w.nevents;
   agc_after_peak(w, cfg, P, cx.heights, is_top, t_peak);
   if (agc_fatal(w, cx, pidx, trk, n + left_distance + 1)) { w.blind_until = 1ll << 60; w.slow_countdown = 1 << 30; return; }
   update_thresholds(w, P, cfg->lsb_per_volt);
   w.blind_until = n + left_distance; }                          // pkww_countdown = left_distance (src/decoder.c:741)

// all lanes: turn this tile's queued records of one walker into events (no tile data needed)
__device__ __forceinline__ void finalize_records(const Ctx &cx, const Rec *recs, int nrec, int pidx, int trk, int lane, int nlanes) {
   const DevCfg *cfg = cx.cfg;
   for (int i = lane; i < nrec; i += nlanes) {
      const Rec r = recs[i];
      const bool is_top = r.kind == 0;
      const int adjcode = refine_code(cfg, r.val, r.prev, r.next, r.g, is_top);
      store_event(cx, pidx, trk, r.idx, cx.tile.row0 + r.n_rel, volt(r.val, cfg->maxvolts), r.g, is_top, adjcode, r.ld); } }


// exact window minimum and its first position (the rescan of src/decoder.c:767-775)
__device__ __forceinline__ void rescan_min(const Tile &tl, int trk, long long lo, long long hi, int &mn, long long &pos) {
   mn = 0x7fffffff; pos = lo;
   for (long long j = lo; j <= hi; ++j) { const int v = tl.y(trk, j); if (v < mn) { mn = v; pos = j; } } }

// last forced rescan ("window maximum leaves the window") in rows (after, upto] of the current tile, or -1
__device__ __forceinline__ long long last_forced_rescan(const Tile &tl, int screen, int trk, long long after, long long upto) {
   const u64 *am = tl.map(screen, 2, trk);
   long long lo = after + 1 - tl.row0, hi = upto - tl.row0;          // tile-relative, inclusive
   if (lo < 0) lo = 0;
   if (hi < lo) return -1;
   for (long long wd = hi >> 6; wd >= (lo >> 6); --wd) {
      u64 m = am[wd];
      if (wd == (hi >> 6) && (hi & 63) != 63) m &= (1ull << ((hi & 63) + 1)) - 1;
      if (wd == (lo >> 6)) m &= ~0ull << (lo & 63);
      if (m) return tl.row0 + wd * 64 + (63 - __clzll((long long)m)); }
   return -1; }

// bring the stale-minimum state forward to "after row n" using the rescan bitmap of this tile
__device__ __forceinline__ void advance_chain(Walker &w, const Tile &tl, int screen, int trk, int W, long long n) {
   if (n <= w.cpos && !w.chain_pending) return;
   const long long a = last_forced_rescan(tl, screen, trk, w.cpos, n);
   long long pos;
   if (a >= 0) { rescan_min(tl, trk, a - W + 1, a, w.minv, pos); w.cpos = a; w.qtrig = pos + W; }
   else if (w.chain_pending) { rescan_min(tl, trk, w.cpos - W + 1, w.cpos, w.minv, pos); w.qtrig = pos + W; }
   w.chain_pending = false;
   while (w.qtrig <= n) {                                         // the stale minimum itself leaves the window
      const long long t = w.qtrig;
      rescan_min(tl, trk, t - W + 1, t, w.minv, pos);
      w.cpos = t; w.qtrig = pos + W; }
   if (n > w.cpos) w.cpos = n; }

// literal lookfor_peak for one row while the window is still filling or the deskew FIFO is in its
// start-up regime (src/decoder.c:751-810 with the state of src/decoder.c:855-861).
__device__ __forceinline__ void slow_step(Walker &w, Ctx &cx, int pidx, int trk, const DevParm &P, long long n) {
   const Tile &tl = cx.tile;
   const DevCfg *cfg = cx.cfg;
   const int W = P.W;
   if (n == w.start) {                                            // seed the window, src/decoder.c:855-861
      const int v = tl.y(trk, n);
      w.slow_max = v; w.minv = v; w.slow_countdown = 0;
      w.t_lastpeak = time_of(cfg, cx.row_base + n);
      return; }
   const long long nin = n - w.start + 1;                         // rows seen including this one
   const bool popped = nin > W;
   const long long lo = popped ? n - W + 1 : w.start;
   const int vnow = tl.y(trk, n);
   const int old_left = popped ? tl.y(trk, n - W) : 0;            // "float old_left = 0" when nothing is popped
   if (vnow > w.slow_max) w.slow_max = vnow;
   if (old_left == w.slow_max || old_left == w.minv) {             // exact == on floats is == on the int16 codes
      int mx = -0x7fffffff, mn = 0x7fffffff;
      for (long long j = lo; j <= n; ++j) { const int v = tl.y(trk, j); mx = max(mx, v); mn = min(mn, v); }
      w.slow_max = mx; w.minv = mn; }
   if (w.slow_countdown) { --w.slow_countdown; return; }
   const float rise = w.rise, reqmin = w.reqmin;
   const float mv = cfg->maxvolts;
   const float vl = volt(tl.y(trk, lo), mv), vr = volt(vnow, mv);
   const float vmax = volt(w.slow_max, mv), vmin = volt(w.minv, mv);
   bool top = vmax > vl + rise && vmax > vr + rise && (reqmin == 0 || vmax > reqmin);
   bool bot = !top && vmin < vl - rise && vmin < vr - rise && (reqmin == 0 || vmin < -reqmin);
   if (top || bot) {
      const int val = top ? w.slow_max : w.minv;
      long long p = lo;
      while (p <= n && tl.y(trk, p) != val) ++p;
      if (p > n || p == lo || p == n) { w.flags |= RTFE_F_DETECTOR_FATAL; return; }   // src/decoder.c:709-710,748
      // refine_peak's time formula and countdown use W even when the window is not full
      emit_peak(w, cx, pidx, trk, P, n, lo, p, volt(val, mv), val, top, false);
      if (!(w.flags & RTFE_F_AGC_FATAL)) w.slow_countdown = (int)(p - lo) + 1; } }

// switch from the literal path to the screened path: derive the lazy stale-min state
__device__ __forceinline__ void enter_fast(Walker &w, const Tile &tl, int trk, int W, long long n_first_fast) {
   const long long last = n_first_fast - 1;
   long long pos = last - W + 1;
   while (pos <= last && tl.y(trk, pos) != w.minv) ++pos;
   w.qtrig = pos + W;                                             // (pos <= last always: the stale min is a window element)
   w.cpos = last; w.chain_pending = false;
   w.blind_until = last + w.slow_countdown;
   w.next = n_first_fast;
   w.trust_from = n_first_fast;
   w.fast = true; }

__device__ __forceinline__ int stale_ld(const u64 *am, const unsigned char *ldb, int q);
// exact evaluation of one candidate row (the flat body of the screened walker); returns true on a detection
// (the detection itself - emit_peak - is left to the caller: walk() runs it for all lanes of a wave together)
__device__ __forceinline__ bool eval_at(Walker &w, Ctx &cx, int pidx, int trk, const DevParm &P, int n, bool ctop, bool cbot, bool async, bool trusted,
                                        bool &is_top_out, int &pos_out, int &val_out) {
   const DevCfg *cfg = cx.cfg;
   const Tile &tl = cx.tile;
   const int W = P.W;
   const float mv = cfg->maxvolts;
   const Col yb = tile_col(tl, trk, cfg->skew[trk]);
   const int lo = n - W + 1;
   const int vl = yb[lo], vr = yb[n];
   bool hit = false, is_top = false;
   int pos = 0, val = 0;
   if (ctop) {
      pos = lo + tl.ldmap(P.screen, 0, trk)[n] - 1;               // first window maximum, from the screen
      val = yb[pos];
      hit = above_by(val, vl, w.rise, w.rise_lo, w.rise_hi, mv) && above_by(val, vr, w.rise, w.rise_lo, w.rise_hi, mv)
            && (w.reqmin == 0 || (val >= w.min_hi) || (val > w.min_lo && volt(val, mv) > w.reqmin));
      is_top = hit; }
   if (!hit && cbot) {
      if (async) {
         // the window maximum left the window at this very row: the reference rescans here, so its
         // minimum is the true window minimum (src/decoder.c:767-775), whose position the screen recorded
         pos = lo + tl.ldmap(P.screen, 1, trk)[n] - 1;
         val = yb[pos];
         w.minv = val; w.cpos = tl.row0 + n; w.qtrig = tl.row0 + pos + W; w.chain_pending = false; }
      else {
         // the reference's (possibly stale) minimum: from the last A-sync row within reach of the tile's screened halo
         // (stale_ld), else from the walker's own chain state
         const int l = trusted ? stale_ld(tl.map(P.screen, 2, trk), tl.ldmap(P.screen, 1, trk), n) : 0;
         if (l) { pos = lo + l - 1; val = yb[pos]; }
         else { advance_chain(w, tl, P.screen, trk, W, tl.row0 + n); val = w.minv; pos = -1; } }
      hit = below_by(val, vl, w.rise, w.rise_lo, w.rise_hi, mv) && below_by(val, vr, w.rise, w.rise_lo, w.rise_hi, mv)
            && (w.reqmin == 0 || (-val >= w.min_hi) || (-val > w.min_lo && volt(val, mv) < -w.reqmin));
      if (hit && pos < 0) { pos = lo; while (pos <= n && yb[pos] != val) ++pos; }
      if (hit && (pos > n || pos == lo || pos == n)) { w.flags |= RTFE_F_DETECTOR_FATAL; hit = false; } }
   is_top_out = is_top; pos_out = pos; val_out = val;
   return hit; }

// one (parameter set, track) detector over rows [.., limit): the exact path on the samples in LDS
__device__ __forceinline__ void walk(Walker &w, Ctx &cx, int pidx, int trk, long long limit) {
   const DevCfg *cfg = cx.cfg;
   const DevParm &P = cfg->parm[pidx];
   const Tile &tl = cx.tile;
   const int W = P.W;
   const long long tile_end = tl.row0 + tl.nrows;
   if (limit > tile_end) limit = tile_end;
   if (w.thr_dirty) update_thresholds(w, P, cfg->lsb_per_volt);
   // ---- literal start-up path ----
   if (!w.fast) {
      const long long fast_from = tl.reset + W + max(trk, cfg->skew[trk]) + 1;
      while (w.next < limit && w.next < fast_from) { if (w.next >= w.start) slow_step(w, cx, pidx, trk, P, w.next); ++w.next; }
      if (w.next < fast_from) return;
      enter_fast(w, tl, trk, W, w.next); }
   // ---- screened path: every candidate row that is not inside a blind countdown, exactly ----
   const u64 *tm = tl.map(P.screen, 0, trk), *bm = tl.map(P.screen, 1, trk), *am = tl.map(P.screen, 2, trk);
   const int lim = (int)(limit - tl.row0);
   long long n64 = max(w.next, w.blind_until + 1);
   int n = (n64 - tl.row0 > lim) ? lim : (int)(n64 - tl.row0);       // first row not yet looked at
   // Two-phase rounds keep the walker lanes of a wave together: (A) every lane scans to its next detection, (B) the lanes
   // that found one do the detection bookkeeping (event, AGC mirror, thresholds) side by side.  In dense formats this is
   // the difference between one bookkeeping pass per round and one per lane per detection.
   const float mv = cfg->maxvolts;
   const bool trusted = tl.row0 - kScreenHalo >= w.trust_from;       // (the screened halo lies in the regular regime)
   #pragma nounroll
   for (;;) {
      bool hit = false, is_top = false;
      int pos = 0, val = 0;
      #pragma nounroll
      while (n < lim) {                                             // (A)
         int wd = n >> 6;
         const u64 c = (tm[wd] | bm[wd]) >> (n & 63);
         if (!c) { n = (wd + 1) << 6; continue; }
         n += __ffsll((long long)c) - 1;
         if (n >= lim) break;
         const int bit = n & 63;
         const bool ctop = (tm[wd] >> bit) & 1, cbot = (bm[wd] >> bit) & 1;
         hit = eval_at(w, cx, pidx, trk, P, n, ctop, cbot, (am[wd] >> bit) & 1, trusted, is_top, pos, val);
         if (hit) break;
         ++n; }
      if (!hit) break;
      emit_peak(w, cx, pidx, trk, P, tl.row0 + n, tl.row0 + n - W + 1, tl.row0 + pos, volt(val, mv), val, is_top, true);     // (B)
      const long long nb = w.blind_until + 1 - tl.row0;             // (blind for ever after the reference's AGC assert, emit_peak: not an int any more)
      n = nb > lim ? lim : (int)nb; }
   n64 = tl.row0 + n;
   w.next = n64 < limit ? n64 : limit;
   // keep the stale-min state inside the reach of the next tile's halo, lazily: remember the last forced
   // rescan of this tile (its window is re-read only if a later bottom needs it); walk the chain eagerly
   // only when that rescan lies too far back (no falling slope for ~100 rows: rare)
   if (limit - 1 > w.cpos) {
      const long long a = last_forced_rescan(tl, P.screen, trk, w.cpos, limit - 1);
      if (a >= 0) { w.cpos = a; w.chain_pending = true; }
      if (limit - 1 - w.cpos > tl.halo - 2 * W - 8 - cfg->maxskew) advance_chain(w, tl, P.screen, trk, W, limit - 1); } }

// lookfor_zerocrossing (src/decoder.c:617-649) on the int16 codes: every row, one lane per track.  Emits every
// CONFIRMED crossing; the slope gate of :629/:643 needs the decoder's clock average and is applied by the host
// replay, which gets the crossing row as (confirmation row - delay).  Event: sample = confirmation row,
// v_peak = the new extreme, agc_gain bits = delay in rows, left_distance = min(delay, 255).
// the detector's state between rows, and one row of it
struct ZcState { int prev, top, bot; bool up, dn; long long ttop, tbot; };
template <class WT> __device__ __forceinline__ void zc_load(ZcState &z, const WT &w) {
   z.prev = w.z_prev; z.top = w.z_top; z.bot = w.z_bot; z.up = w.z_up_pending; z.dn = w.z_dn_pending; z.ttop = w.z_ttop_row; z.tbot = w.z_tbot_row; }
template <class WT> __device__ __forceinline__ void zc_store(WT &w, const ZcState &z) {
   w.z_prev = z.prev; w.z_top = z.top; w.z_bot = z.bot; w.z_up_pending = z.up; w.z_dn_pending = z.dn; w.z_ttop_row = z.ttop; w.z_tbot_row = z.tbot; }
// returns true when row n (code v) confirms a crossing: `up` its direction, `cross` the row of the sign change
__device__ __forceinline__ bool zc_row(ZcState &z, int v, long long n, int P, bool &up, long long &cross) {
   bool emit = false;
   if (v > 0) {
      z.dn = false;
      if (z.top < v) {
         z.top = v;
         if (z.up && z.top >= P) { z.up = false; z.bot = 0; emit = true; up = true; cross = z.ttop; } }
      if (z.prev < 0 && z.bot <= -P) { z.ttop = n; z.up = true; } }
   else if (v < 0) {
      z.up = false;
      if (z.bot > v) {
         z.bot = v;
         if (z.dn && z.bot <= -P) { z.dn = false; z.top = 0; emit = true; up = false; cross = z.tbot; } }
      if (z.prev > 0 && z.top >= P) { z.tbot = n; z.dn = true; } }
   z.prev = v;
   return emit; }
__device__ __forceinline__ void zc_event(const Ctx &cx, int trk, unsigned int idx, long long n, int v, bool up, long long cross) {
   const unsigned int delay = (unsigned int)(n - cross);
   rtfe_event e;
   e.sample = (uint32_t)(n - cx.tile.reset);
   e.v_peak = (cx.cfg->invert && v == 0) ? -0.0f : volt(v, cx.cfg->maxvolts);
   e.agc_gain = __uint_as_float(delay);
   e.trk = (uint8_t)trk;
   e.flags = (uint8_t)(up ? 0 : 1);
   e.left_distance = (uint8_t)(delay < 255 ? delay : 255);
   e.parmset = 0;
   cx.events[(size_t)trk * cx.cap + idx] = e; }

template <class WT> __device__ __forceinline__ void walk_zeros(WT &w, Ctx &cx, int trk, long long limit) {
   const DevCfg *cfg = cx.cfg;
   const Tile &tl = cx.tile;
   const long long tile_end = tl.row0 + tl.nrows;
   if (limit > tile_end) limit = tile_end;
   const int P = cfg->zc_peak_i;
   long long n = w.next;
   if (n <= w.start) n = w.start + 1;                              // row `start` only seeds the track (src/decoder.c:855-861)
   ZcState z; zc_load(z, w);
   for (; n < limit; ++n) {
      const int v = tl.y(trk, n);
      bool up = false; long long cross = 0;
      if (zc_row(z, v, n, P, up, cross)) {
         if (w.nevents < cx.cap) zc_event(cx, trk, w.nevents, n, v, up, cross);
         else w.flags |= RTFE_F_EVENT_OVERFLOW;
         ++w.nevents; } }
   zc_store(w, z);
   w.next = n; }

// -zeros, one whole tile, all tracks: the 512 rows of a track are cut into sub-segments of kZcSub rows that run
// concurrently, one lane each.  Sub-segment 0 continues from the walker's true state; the others start kZcWarm rows early
// from a fresh state (what a restart would be) and note the state they reach at their first own row.  The detector
// forgets: after a confirmed crossing in each direction its state is a function of the samples since.  A track's result
// stands only if every sub-segment's noted state equals its predecessor's final state in every field; otherwise that
// track is walked again sequentially (walk_zeros) from the untouched walker.  ok[trk] = 1 where the result stands.
constexpr int kZcSub = 64, kZcWarm = 64, kZcMaxEv = 8;
// lookfor_zerocrossing's step (zc_row) on 32-bit state with selects instead of branches: rows are tile-relative (q).  The order of
// the reference's statements is kept (clear the other side's pending flag, new extreme and confirmation, arming), so the state after
// every row is zc_row's.  The two pending flags live in the crossing rows themselves (kZcNone = not pending: a crossing row is only
// ever read while its flag is set), and top >= 0 >= bot always, so "new extreme" is a plain max / min and "new extreme that reaches
// the threshold" one compare against max(top, P - 1): 23 vector operations per row instead of 31.
// Returns 0 / 1 (up confirmed) / 2 (down confirmed); cross = tile-relative row of the sign change.
constexpr int kZcNone = -(1 << 30);
struct Zc32 { int prev, top, bot, ttop, tbot; };
__device__ __forceinline__ int zc_step32(Zc32 &z, int v, int q, int P, int &cross) {
   const bool pos = v > 0, neg = v < 0;
   z.tbot = pos ? kZcNone : z.tbot;
   z.ttop = neg ? kZcNone : z.ttop;
   const bool e_up = v > max(z.top, P - 1) && z.ttop != kZcNone;        // (v > top >= 0: a positive sample)
   const bool e_dn = v < min(z.bot, 1 - P) && z.tbot != kZcNone;
   z.top = max(z.top, v);
   z.bot = min(z.bot, v);
   cross = e_up ? z.ttop : z.tbot;
   z.ttop = e_up ? kZcNone : z.ttop;   z.bot = e_up ? 0 : z.bot;
   z.tbot = e_dn ? kZcNone : z.tbot;   z.top = e_dn ? 0 : z.top;
   const bool arm_up = pos && z.prev < 0 && z.bot <= -P, arm_dn = neg && z.prev > 0 && z.top >= P;
   z.ttop = arm_up ? q : z.ttop;
   z.tbot = arm_dn ? q : z.tbot;
   z.prev = v;
   return e_up ? 1 : (e_dn ? 2 : 0); }
__device__ __forceinline__ Zc32 zc_to32(const ZcState &s, long long row0) {
   Zc32 z; z.prev = s.prev; z.top = s.top; z.bot = s.bot;
   const long long a = s.ttop - row0, b2 = s.tbot - row0;               // (a pending crossing is recent; older ones are clamped above the marker)
   z.ttop = s.up ? (int)(a <= kZcNone ? kZcNone + 1 : a) : kZcNone;
   z.tbot = s.dn ? (int)(b2 <= kZcNone ? kZcNone + 1 : b2) : kZcNone;
   return z; }
__device__ __forceinline__ ZcState zc_from32(const Zc32 &z, long long row0) {
   ZcState s; s.prev = z.prev; s.top = z.top; s.bot = z.bot; s.up = z.ttop != kZcNone; s.dn = z.tbot != kZcNone; s.ttop = row0 + z.ttop; s.tbot = row0 + z.tbot;
   return s; }

// one sub-segment's record: the state it started from (as assumed) and ended in, its events (ev: n_rel | code << 16 , delay | up << 31;
// slot kZcMaxEv takes what does not fit: count > kZcMaxEv says so)
struct ZcLane { Zc32 start, end; int count, bad; unsigned int ev[kZcMaxEv + 1][2]; };
__device__ __forceinline__ bool zc_same(const Zc32 &a, const Zc32 &b) {         // (a crossing row that is not pending is kZcNone on both sides)
   return a.prev == b.prev && a.top == b.top && a.bot == b.bot && a.ttop == b.ttop && a.tbot == b.tbot; }
// the own rows of sub-segment j from state z: events into the lane's record, the end state.  The samples are read kZcAhead batches of
// eight rows ahead of the steps that use them (pre = the batches already on their way, or nullptr): a batch's load latency hides
// behind the dependent steps of the batches in front of it instead of stalling every lane of the wave.
constexpr int kZcAhead = 2;
template <class ColT> __device__ __forceinline__ void zc_load8(int (&v)[8], const ColT &yb, int q) {
   #pragma unroll
   for (int k = 0; k < 8; ++k) v[k] = yb[q + k]; }
struct ZcAhead { int v[kZcAhead][8]; };
template <class ColT> __device__ __forceinline__ void zc_own_rows(ZcLane &me, const Zc32 z0, const ColT &yb, int j, int P, const ZcAhead *pre = nullptr) {
   int cnt = 0;
   Zc32 z = z0;
   const int qend = (j + 1) * kZcSub;
   ZcAhead nx;
   if (pre) nx = *pre;
   else {
      #pragma unroll
      for (int a = 0; a < kZcAhead; ++a) zc_load8(nx.v[a], yb, j * kZcSub + 8 * a); }
   #pragma nounroll
   for (int q = j * kZcSub; q < qend; q += 8) {
      int v8[8];
      #pragma unroll
      for (int k = 0; k < 8; ++k) v8[k] = nx.v[0][k];
      #pragma unroll
      for (int a = 0; a + 1 < kZcAhead; ++a) {
         #pragma unroll
         for (int k = 0; k < 8; ++k) nx.v[a][k] = nx.v[a + 1][k]; }
      zc_load8(nx.v[kZcAhead - 1], yb, min(q + 8 * kZcAhead, qend - 8));       // (behind the last batch: that batch again - no branch, no row outside the tile)
      #pragma unroll
      for (int k = 0; k < 8; ++k) {
         int cross;
         const int e = zc_step32(z, v8[k], q + k, P, cross);
         if (e) {
            const int slot = cnt < kZcMaxEv ? cnt : kZcMaxEv;
            me.ev[slot][0] = (unsigned)(q + k) | ((unsigned)(v8[k] & 0xffff) << 16); me.ev[slot][1] = (unsigned)(q + k - cross) | (e == 1 ? 0x80000000u : 0u);
            ++cnt; } } }
   me.end = z; me.count = cnt; }

// 0: sub-segment L of the lanes starts where its predecessor ended; 1: it does not; 2: it holds more events than its record can
__device__ __forceinline__ int zc_join_verdict(const ZcLane *lanes, int L, int j) {
   const ZcLane &me = lanes[L];
   if (me.count > kZcMaxEv) return 2;
   return (j > 0 && !zc_same(me.start, lanes[L - 1].end)) ? 1 : 0; }

template <class WT, bool kGlobal = false> __device__ __forceinline__ void zeros_tile_parallel(Ctx &cx, WT *walkers, ZcLane *lanes, int *ok, long long stop, unsigned long long *dbgp = nullptr) {
   long long k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0;
   if (dbgp) k0 = clock64();
   const DevCfg *cfg = cx.cfg;
   const Tile &tl = cx.tile;
   const int ntrks = cfg->ntrks, nsub = tl.nrows / kZcSub;
   const int P = cfg->zc_peak_i;
   // thread -> (sub-segment, track), the tracks of a sub-segment side by side: neighbouring lanes then read the 18 bytes of one row
   // (one or two cache lines per sub-segment and load, not one per lane); the lanes' records stay track-major (L)
   const int T = threadIdx.x, j = T / ntrks, trk = T - j * ntrks;
   const bool mine = T < ntrks * nsub;
   const int L = trk * nsub + j;
   if (T < ntrks) {                                                  // a whole tile in the regular regime?
      const WT &w = walkers[T];
      ok[T] = (tl.nrows % kZcSub == 0 && nsub >= 2 && stop >= tl.row0 + tl.nrows && w.next == tl.row0 && w.start < tl.row0
               && tl.row0 - kZcWarm - 1 - tl.reset >= cfg->skew[T]          // every row read is behind the deskew FIFO's start-up
               && w.nevents + (unsigned)(nsub * kZcMaxEv) < cx.cap) ? 1 : 0; }
   __syncthreads();
   if (mine && ok[trk]) {
      ZcLane &me = lanes[L];
      Zc32 z;
      const auto yb = zc_col<kGlobal>(tl, trk, cfg->skew[trk]);              // y(n) = yb[n - row0] in the regular regime
      ZcAhead nx;
      if (j == 0) {
         ZcState zs; zc_load(zs, walkers[trk]); z = zc_to32(zs, tl.row0);
         #pragma unroll
         for (int a = 0; a < kZcAhead; ++a) zc_load8(nx.v[a], yb, 8 * a); }
      else {
         const int q0 = j * kZcSub - cfg->zc_warm;
         Zc32 zw; zw.prev = yb[q0 - 1]; zw.top = 0; zw.bot = 0; zw.ttop = kZcNone; zw.tbot = kZcNone;
         #pragma unroll
         for (int a = 0; a < kZcAhead; ++a) zc_load8(nx.v[a], yb, q0 + 8 * a);
         #pragma nounroll
         for (int q = q0; q < j * kZcSub; q += 8) {                   // (the batches behind this one are in flight during its eight dependent steps; the ones read last are the first of the own rows)
            int v8[8];
            #pragma unroll
            for (int k = 0; k < 8; ++k) v8[k] = nx.v[0][k];
            #pragma unroll
            for (int a = 0; a + 1 < kZcAhead; ++a) {
               #pragma unroll
               for (int k = 0; k < 8; ++k) nx.v[a][k] = nx.v[a + 1][k]; }
            zc_load8(nx.v[kZcAhead - 1], yb, q + 8 * kZcAhead);
            #pragma unroll
            for (int k = 0; k < 8; ++k) { int cross; (void)zc_step32(zw, v8[k], q + k, P, cross); } }
         z = zw; }
      me.start = z;
      zc_own_rows(me, z, yb, j, P, &nx); }
   __syncthreads();
   if (dbgp) k1 = clock64();
   // Does every sub-segment start where its predecessor ended?  Where one does not, it alone is run again (its own 64 rows) from
   // the predecessor's end state - which is the true state, by induction from sub-segment 0 - and the check moves on to the next
   // join; a track costs one repair per join that failed, not a sequential walk of the whole tile.
   int *first_bad = ok + RTFE_MAXTRKS;                               // [ntrks] sub-segment to repair in this round (nsub: none)
   for (int round = 0; round < nsub; ++round) {
      if (mine && ok[trk]) {                                         // every lane checks its own join
         ZcLane &me = lanes[L];
         me.bad = zc_join_verdict(lanes, L, j); }
      __syncthreads();
      if (T < ntrks) {
         int fb = nsub;
         if (ok[T]) for (int k = nsub - 1; k >= 0; --k) { const int bd = lanes[T * nsub + k].bad; if (bd == 2) ok[T] = 0; else if (bd == 1) fb = k; }
         first_bad[T] = fb; }
      __syncthreads();
      bool any = false;
      for (int t = 0; t < ntrks; ++t) any = any || (ok[t] && first_bad[t] < nsub);
      if (!any) break;
      if (mine && ok[trk] && first_bad[trk] == j) {
         ZcLane &me = lanes[L];
         const auto yb = zc_col<kGlobal>(tl, trk, cfg->skew[trk]);
         const Zc32 zp = lanes[L - 1].end;
         me.start = zp;
         zc_own_rows(me, zp, yb, j, P); }
      __syncthreads(); }
   if (dbgp) k2 = clock64();
   if (mine && ok[trk]) {                                            // events in row order; the walker moves to the tile's end
      const ZcLane &me = lanes[L];
      unsigned int idx = walkers[trk].nevents;
      for (int k = 0; k < j; ++k) idx += (unsigned)lanes[L - j + k].count;
      for (int k = 0; k < me.count && k < kZcMaxEv; ++k) {
         const long long n = tl.row0 + (long long)(me.ev[k][0] & 0xffff);
         const int v = (int)(short)(me.ev[k][0] >> 16);
         zc_event(cx, trk, idx + k, n, v, (me.ev[k][1] >> 31) != 0, n - (long long)(me.ev[k][1] & 0x7fffffffu)); } }
   __syncthreads();
   if (T < ntrks && ok[T]) {
      WT &w = walkers[T];
      unsigned int total = 0;
      for (int k = 0; k < nsub; ++k) total += (unsigned)lanes[T * nsub + k].count;
      zc_store(w, zc_from32(lanes[T * nsub + nsub - 1].end, tl.row0));
      w.nevents += total; w.next = tl.row0 + tl.nrows; }
   if (dbgp) k3 = clock64();
   __syncthreads();
   if (dbgp && threadIdx.x == 0) { k4 = clock64(); atomicAdd(&dbgp[0], (unsigned long long)(k1 - k0)); atomicAdd(&dbgp[1], (unsigned long long)(k2 - k1)); atomicAdd(&dbgp[2], (unsigned long long)(k3 - k2)); atomicAdd(&dbgp[3], (unsigned long long)(k4 - k3)); } }


// differentiate() (src/readtape.c:1383-1394) for every element of the tile, all lanes: delta against the previous row of
// the same head (0 at the burst's restart row: v_last_raw is zeroed by init_trackstate, src/decoder.c:437), the
// +-0.05 V dead band, then x 0.4f x samples_per_bit (float x float, then x int).  Rows in front of the restart row
// are never read by the walker.
__device__ __forceinline__ void differentiate_tile(const DevCfg *cfg, const Tile &tl) {
   const int ntrks = cfg->ntrks, nelem = (tl.halo + tl.nrows) * ntrks;
   const int reset_row = (int)(tl.reset - (tl.row0 - tl.halo));        // tile-LDS row of the restart (may lie outside)
   const float mv = cfg->maxvolts;
   const int spb = cfg->samples_per_bit;
   for (int i = threadIdx.x; i < nelem; i += blockDim.x) {
      const int row = i / ntrks;
      const float v = volt(tl.x[i], mv);
      const float vprev = (row == reset_row || i < ntrks) ? 0.0f : volt(tl.x[i - ntrks], mv);
      float delta = v - vprev;
      if (delta < 0.05f && delta > -0.05f) delta = 0;
      tl.fd[i] = delta * 0.4f * spb; } }

// lookfor_peak + refine_peak (src/decoder.c:700-810) on the differentiated signal (-differentiate without -zeros): the
// literal per-row detector on the float tile - the window is rows [n-W+1, n] of the tile, maximum exact, minimum stale
// (src/decoder.c:765), rescan when the leaving sample equals either, blind countdown, top before bottom.
__device__ __forceinline__ void walk_diffpeak(Walker &w, Ctx &cx, int pidx, int trk, long long limit) {
   const DevCfg *cfg = cx.cfg;
   const DevParm &P = cfg->parm[pidx];
   const Tile &tl = cx.tile;
   const int W = P.W;
   const long long tile_end = tl.row0 + tl.nrows;
   if (limit > tile_end) limit = tile_end;
   long long n = w.next;
   if (n < w.start) n = w.start;
   for (; n < limit; ++n) {
      const float vnow = tl.fy(trk, n);
      if (n == w.start) {                                          // seed the window, src/decoder.c:855-861
         w.zf_top = vnow; w.zf_bot = vnow; w.slow_countdown = 0;
         w.t_lastpeak = time_of(cfg, cx.row_base + n);
         continue; }
      const bool popped = n - w.start + 1 > W;
      const long long lo = popped ? n - W + 1 : w.start;
      const float old_left = popped ? tl.fy(trk, n - W) : 0.0f;
      if (vnow > w.zf_top) w.zf_top = vnow;
      if (old_left == w.zf_top || old_left == w.zf_bot) {
         float mx = tl.fy(trk, lo), mn = mx;
         for (long long j = lo + 1; j <= n; ++j) { const float v = tl.fy(trk, j); if (v > mx) mx = v; if (v < mn) mn = v; }
         w.zf_top = mx; w.zf_bot = mn; }
      if (w.slow_countdown) { --w.slow_countdown; continue; }
      const float rise = w.rise, reqmin = w.reqmin;
      const float vl = tl.fy(trk, lo);
      const bool top = w.zf_top > vl + rise && w.zf_top > vnow + rise && (reqmin == 0 || w.zf_top > reqmin);
      const bool bot = !top && w.zf_bot < vl - rise && w.zf_bot < vnow - rise && (reqmin == 0 || w.zf_bot < -reqmin);
      if (!(top || bot)) continue;
      const float val = top ? w.zf_top : w.zf_bot;
      long long p = lo;
      while (p <= n && tl.fy(trk, p) != val) ++p;
      if (p > n || p == lo || p == n) { w.flags |= RTFE_F_DETECTOR_FATAL; continue; }      // src/decoder.c:709-710,748
      const int left_distance = (int)(p - lo) + 1;
      const float prev = tl.fy(trk, p - 1), next = tl.fy(trk, p + 1);
      int adjcode = 0;                                             // 1 = -0.5, 2 = +0.5 (src/decoder.c:712-731)
      if (top) {
         const float lim = val - 0.005f / w.agc_gain;
         if (prev > lim && next < lim) adjcode = 1; else if (next > lim && prev < lim) adjcode = 2; }
      else {
         const float lim = val + 0.005f / w.agc_gain;
         if (prev < lim && next > lim) adjcode = 1; else if (next < lim && prev > lim) adjcode = 2; }
      const float adj = adjcode == 1 ? -0.5f : (adjcode == 2 ? 0.5f : 0.0f);
      const double t_peak = time_of(cfg, cx.row_base + n) - ((float)(W - left_distance) - adj) * cfg->sample_deltat;
      if (w.nevents >= cx.cap) w.flags |= RTFE_F_EVENT_OVERFLOW;
      else store_event(cx, pidx, trk, w.nevents, n, val, w.agc_gain, top, adjcode, left_distance);
      if (top) w.v_top = val; else w.v_bot = val;
      ++w.nevents;
      agc_after_peak(w, cfg, P, cx.heights, top, t_peak);
      if (agc_fatal(w, cx, pidx, trk, n + left_distance + 1)) { w.slow_countdown = 1 << 30; continue; }
      update_thresholds(w, P, cfg->lsb_per_volt);
      w.slow_countdown = left_distance; }
   w.next = n; }

// lookfor_differentiated_zerocrossing (src/decoder.c:654-683) on differentiate()'s output (src/readtape.c:1383-1388),
// every row, one lane per track.  The differentiator restarts against 0 at the burst's restart row.
// Event: sample = row at which the pending crossing is confirmed, v_peak = v_top / v_bot at that moment,
// agc_gain bits = (rows back to the first exact zero) << 16 | (rows back to the last one); 0 = no zero seen: the
// crossing then lies half a sample before the confirmation row (src/decoder.c:658-660).
__device__ __forceinline__ void walk_diffzeros(Walker &w, Ctx &cx, int trk, long long limit) {
   const DevCfg *cfg = cx.cfg;
   const Tile &tl = cx.tile;
   const long long tile_end = tl.row0 + tl.nrows;
   if (limit > tile_end) limit = tile_end;
   const float mv = cfg->maxvolts;
   const int d = cfg->skew[trk];
   const int spb = cfg->samples_per_bit;
   long long n = w.next;
   if (n <= w.start) n = w.start + 1;                              // row `start` only seeds the track (src/decoder.c:855-861)
   for (; n < limit; ++n) {
      // what the detector sees at row n: the differentiated sample of row src (deskew FIFO, src/decoder.c:825-828)
      const long long src = (n - tl.reset < d) ? n : n - d;
      const float vraw = volt(tl.xi(trk, src), mv);
      const float vprev = src == tl.reset ? 0.0f : volt(tl.xi(trk, src - 1), mv);   // v_last_raw = 0 at the restart (src/decoder.c:437)
      float delta = vraw - vprev;
      if (delta < 0.05f && delta > -0.05f) delta = 0;
      const float v = delta * 0.4f * spb;
      bool emit = false, up = false; float vpk = 0; unsigned int d1 = 0, d2 = 0;
      if (v > 0) {
         if (w.zf_top < v) w.zf_top = v;
         if (w.z_up_pending) {
            if (w.z_firstzero >= 0) { d1 = (unsigned int)(n - w.z_firstzero); d2 = (unsigned int)(n - w.z_lastzero); }
            w.z_up_pending = false; w.z_firstzero = -1; emit = true; up = true; vpk = w.zf_top; }
         if (v > 0.2f) { w.z_dn_pending = true; w.z_firstzero = -1; w.zf_bot = 0; } }
      else if (v < 0) {
         if (w.zf_bot > v) w.zf_bot = v;
         if (w.z_dn_pending) {
            if (w.z_firstzero >= 0) { d1 = (unsigned int)(n - w.z_firstzero); d2 = (unsigned int)(n - w.z_lastzero); }
            w.z_dn_pending = false; w.z_firstzero = -1; emit = true; up = false; vpk = w.zf_bot; }
         if (v < -0.2f) { w.z_up_pending = true; w.z_firstzero = -1; w.zf_top = 0; } }
      else { w.z_lastzero = n; if (w.z_firstzero < 0) w.z_firstzero = n; }
      if (emit) {
         if (w.nevents < cx.cap && d1 < 65536u) {
            rtfe_event e;
            e.sample = (uint32_t)(n - tl.reset);
            e.v_peak = vpk;
            e.agc_gain = __uint_as_float((d1 << 16) | d2);
            e.trk = (uint8_t)trk;
            e.flags = (uint8_t)(up ? 0 : 1);
            e.left_distance = (uint8_t)(d1 < 255 ? d1 : 255);
            e.parmset = 0;
            cx.events[(size_t)trk * cx.cap + w.nevents] = e; }
         else w.flags |= RTFE_F_EVENT_OVERFLOW;
         ++w.nevents; } }
   w.next = n; }

// ---- the reference's (possibly stale) window minimum at a row ----
// The detector's min/max tracking does not depend on its decisions (src/decoder.c:760-775 runs before the countdown
// test): the maximum is always exact, and the minimum is refreshed exactly at the rows where the sample leaving the
// window equals the tracked maximum ("A-sync" rows: data alone decides, the screen's bitmap 2) or the tracked minimum.
// Between two A-sync rows the minimum therefore follows a chain that starts from the true minimum at the first of
// them: it stays the same sample until that sample leaves the window, where a rescan makes it the true minimum again.
// the left_distance of the reference's minimum at row q (0 = unknown: no A-sync row within reach of the tile's halo)
__device__ __forceinline__ int stale_ld(const u64 *am, const unsigned char *ldb, int q) {
   int wd = q >> 6;
   u64 m = am[wd] & (~0ull >> (63 - (q & 63)));                     // A-sync rows <= q in q's word
   #pragma nounroll
   while (!m) { if (--wd < -1) return 0; m = am[wd]; }
   int h = wd * 64 + 63 - __clzll((long long)m);                     // last A-sync row: the minimum is the true one there
   #pragma nounroll
   for (;;) {
      const int l = ldb[h];                                          // (a rescan row: its byte is never rewritten with another value)
#ifdef RTFE_CPU_EMUL
      if (l == 0) { fprintf(stderr, "stale_ld: zero left_distance at rescan row %d (asked for row %d)\n", h, q); abort(); }
#endif
      if (h + l > q) return l - (q - h);                             // still the same sample at row q
      h += l; } }                                                    // it left the window at row h + l: rescan there

struct FastDiv {
   unsigned n, M;
   __device__ __forceinline__ explicit FastDiv(int d) : n((unsigned)d), M(d > 1 ? 0xFFFFFFFFu / (unsigned)d + 1u : 0u) {}
   __device__ __forceinline__ int div(int i) const { return n > 1 ? (int)__umulhi((unsigned)i, M) : i; } };

// ---- candidate screen: one thread = one strip of 8 consecutive rows of one track ----
// window max/min by prefix/suffix decomposition around the strip start (van Herk with one block edge)
// mmax (optional): the largest margin any candidate of the strip has - the window maximum above both edges, or both edges above the TRUE
// window minimum (the reference's stale minimum is no lower): an upper bound of every margin a walker of these rows can meet (rtfe_dense.hip)
__device__ __forceinline__ int screen_strip(const Tile &tl, const DevScreen &sc, int screen, int trk, int strip, int *mmax = nullptr) {
   // Keys carry the position so that max/min also yield the FIRST window element equal to the extreme
   // (what refine_peak looks for, src/decoder.c:707-708): r = index relative to the strip's leftmost
   // window element (s0 - W + 1);  kmax = v<<8 | (255 - r)  (max -> largest v, then smallest r),
   // kmin = v<<8 | r  (min -> smallest v, then smallest r).
   const int W = sc.W;
   const int d = tl.skew[trk];
   const Col base = tile_col(tl, trk, d);                          // y(n) = base[n - row0] in the regular regime
   const int s0 = strip * kStrip;
   int mm = 0;
   int v[kStrip], L[kStrip];
   int topb = 0, botb = 0, resb = 0;
   u64 ldt = 0, ldb = 0;
   if (W > kStrip) {
      // the W + 8 samples s0-W .. s0+7 are read in ascending order through one running pointer (the column stride is
      // a run-time value: one multiply for the start, additions from there)
      const int16_t *rp = base.p + (s0 - W) * base.P;
      const int P = base.P;
      int popped = *rp; rp += P;                                   // the sample that leaves the window at row s0 (then L[i-1])
      #pragma unroll
      for (int i = 0; i < kStrip; ++i) { L[i] = *rp; rp += P; }
      // keys: kmin = v << 8 | r, kmax = kmin ^ 0xff (= v << 8 | (255 - r)): one shift-or per sample serves both
      int smx[kStrip], smn[kStrip];
      int amx = (int)0x80000000, amn = 0x7fffffff;
      for (int r = kStrip; r < W - 1; ++r) {                         // the samples between L[7] and v[0]
         const int kk = ((int)*rp << 8) | r; rp += P;
         amx = max(amx, kk ^ 0xff); amn = min(amn, kk); }
      #pragma unroll
      for (int i = 0; i < kStrip; ++i) { v[i] = *rp; rp += P; }
      #pragma unroll
      for (int i = kStrip - 1; i >= 0; --i) {
         const int kk = (L[i] << 8) | i;
         amx = max(amx, kk ^ 0xff); amn = min(amn, kk); smx[i] = amx; smn[i] = amn; }
      int pmx = (int)0x80000000, pmn = 0x7fffffff;
      #pragma unroll
      for (int i = 0; i < kStrip; ++i) {
         const int kk = (v[i] << 8) | (W - 1 + i);
         pmx = max(pmx, kk ^ 0xff); pmn = min(pmn, kk);
         const int kx = max(smx[i], pmx), kn = min(smn[i], pmn);
         const int mx = kx >> 8, mn = kn >> 8;
         const bool t = (mx - L[i] > sc.rise_i) && (mx - v[i] > sc.rise_i) && (sc.minpk_i < 0 || mx > sc.minpk_i);
         const bool b = (L[i] - mn > sc.rise_i) && (v[i] - mn > sc.rise_i) && (sc.minpk_i < 0 || mn < -sc.minpk_i);
         topb |= (int)t << i; botb |= (int)b << i; resb |= (int)(popped >= mx) << i;
         if (mmax) { if (t) mm = max(mm, min(mx - L[i], mx - v[i])); if (b) mm = max(mm, min(L[i] - mn, v[i] - mn)); }
         ldt |= (u64)((255 - (kx & 255)) - i + 1) << (8 * i);        // left_distance of the first maximum
         ldb |= (u64)((kn & 255) - i + 1) << (8 * i);                 // ... and of the first (true) minimum
         popped = L[i]; } }
   else {
      #pragma unroll
      for (int i = 0; i < kStrip; ++i) { v[i] = base[s0 + i]; L[i] = base[s0 + i - W + 1]; }
      #pragma unroll
      for (int i = 0; i < kStrip; ++i) {
         int kx = (int)0x80000000, kn = 0x7fffffff;
         for (int j = s0 + i - W + 1; j <= s0 + i; ++j) {
            const int u = base[j], r = j - (s0 + i - W + 1);
            kx = max(kx, (u << 8) | (255 - r)); kn = min(kn, (u << 8) | r); }
         const int mx = kx >> 8, mn = kn >> 8;
         const int popped = base[s0 + i - W];
         const bool t = (mx - L[i] > sc.rise_i) && (mx - v[i] > sc.rise_i) && (sc.minpk_i < 0 || mx > sc.minpk_i);
         const bool b = (L[i] - mn > sc.rise_i) && (v[i] - mn > sc.rise_i) && (sc.minpk_i < 0 || mn < -sc.minpk_i);
         topb |= (int)t << i; botb |= (int)b << i; resb |= (int)(popped >= mx) << i;
         if (mmax) { if (t) mm = max(mm, min(mx - L[i], mx - v[i])); if (b) mm = max(mm, min(L[i] - mn, v[i] - mn)); }
         ldt |= (u64)((255 - (kx & 255)) + 1) << (8 * i);
         ldb |= (u64)((kn & 255) + 1) << (8 * i); } }
   const int ntb2 = tl.ntrks * tl.bstride;
   unsigned char *o = tl.bits + (screen * 5) * ntb2 + trk * tl.bstride + kScreenHalo / 8 + strip;     // strip >= -kScreenHalo/8
   o[0] = (unsigned char)topb;
   o[ntb2] = (unsigned char)botb;
   o[2 * ntb2] = (unsigned char)resb;
   reinterpret_cast<u64 *>(tl.ldmap(screen, 0, trk))[strip] = ldt;
   reinterpret_cast<u64 *>(tl.ldmap(screen, 1, trk))[strip] = ldb;
   if (mmax) *mmax = mm;
   return topb | botb; }

// cooperative tile load: rows [row0 - halo, row0 + nrows) of the AoS payload -> LDS, as they are (16-byte vectors;
// the tile starts on a multiple of 8 rows, so vectors are aligned in HBM and in LDS).  -invert negates on the way.
__device__ __forceinline__ void load_tile(const DevCfg *cfg, Tile &tl, const int16_t *__restrict__ rows, long long total_rows) {
   const int ntrks = tl.ntrks;      // (the caller may know it at compile time)
   const long long first = tl.row0 - tl.halo;
   const int nelem = (tl.halo + tl.nrows) * ntrks;
   const int nvec = (nelem + 7) >> 3;
   const long long total_elem = total_rows * ntrks;
   const long long e_first = first * ntrks;
   const bool inv = cfg->invert != 0;
   int4 *dst = reinterpret_cast<int4 *>(tl.x);
   constexpr int kBatch = 6;                                      // independent 16-B loads in flight per lane
   for (int vbase = 0; vbase < nvec; vbase += kBatch * (int)blockDim.x) {
      int4 q[kBatch];
      #pragma unroll
      for (int k = 0; k < kBatch; ++k) {
         const int vi = vbase + k * (int)blockDim.x + (int)threadIdx.x;
         const long long ge = e_first + (long long)vi * 8;
         q[k] = make_int4(0, 0, 0, 0);
         if (vi < nvec) {
            if (ge >= 0 && ge + 8 <= total_elem) q[k] = *reinterpret_cast<const int4 *>(rows + ge);
            else {                                                    // tape ends: sample by sample, zeros outside
               int e[8];
               #pragma unroll
               for (int j = 0; j < 8; ++j) { const long long g = ge + j; e[j] = (g >= 0 && g < total_elem) ? (int)(unsigned short)rows[g] : 0; }
               q[k] = make_int4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)); } } }
      #pragma unroll
      for (int k = 0; k < kBatch; ++k) {
         const int vi = vbase + k * (int)blockDim.x + (int)threadIdx.x;
         if (vi >= nvec) continue;
         int4 v = q[k];
         if (inv) {                                                   // two int16 per dword: 0 - x, halves independently
            v.x = (int)((((unsigned)(-(v.x << 16))) >> 16) | ((unsigned)(-(v.x >> 16)) << 16));
            v.y = (int)((((unsigned)(-(v.y << 16))) >> 16) | ((unsigned)(-(v.y >> 16)) << 16));
            v.z = (int)((((unsigned)(-(v.z << 16))) >> 16) | ((unsigned)(-(v.z >> 16)) << 16));
            v.w = (int)((((unsigned)(-(v.w << 16))) >> 16) | ((unsigned)(-(v.w >> 16)) << 16)); }
         dst[vi] = v; } } }

// the candidate screen of every (screen, track) over the tile in LDS (with_halo: the kScreenHalo rows in front of it, too)
__device__ __forceinline__ void run_screens(const DevCfg *cfg, const Tile &tl, bool with_halo) {
   const int hs = with_halo ? kScreenHalo / kStrip : 0;
   const int nstrips = (tl.nrows + kStrip - 1) / kStrip + hs;
   const int per_screen = nstrips * cfg->ntrks;
   const FastDiv fd(cfg->ntrks);
   for (int s = 0; s < cfg->nscreens; ++s)
      for (int i = (int)threadIdx.x; i < per_screen; i += blockDim.x) {   // consecutive lanes = the tracks of one strip: consecutive LDS words
         const int q = fd.div(i);
         (void)screen_strip(tl, cfg->screen[s], s, i - q * cfg->ntrks, q - hs); } }

// restart row for the zone whose last kMarginRows rows are the current tile:
// for every parameter set and track take the last forced rescan inside the zone; any restart at or
// before (that row - W - max(trk, skew) - 1) has a full, regular window when the rescan happens.
__device__ __forceinline__ long long find_reset(const DevCfg *cfg, const Tile &tl, long long *lds_min) {
   if (threadIdx.x == 0) *lds_min = 0x7fffffffffffffffll;
   __syncthreads();
   const int nw = cfg->nparm * cfg->ntrks;
   for (int i = threadIdx.x; i < nw; i += blockDim.x) {
      const int p = i / cfg->ntrks, t = i - p * cfg->ntrks;
      const DevParm &P = cfg->parm[p];
      const u64 *am = tl.map(P.screen, 2, t);
      long long a = -1;
      for (int wd = (tl.nrows - 1) >> 6; wd >= 0; --wd) if (am[wd]) { a = tl.row0 + wd * 64 + (63 - __clzll((long long)am[wd])); break; }
      long long hi = a < 0 ? -1 : a - P.W - max(t, cfg->skew[t]) - 2;
      atomicMin((unsigned long long *)lds_min, (unsigned long long)(hi < 0 ? 0 : hi));
      if (hi < tl.row0) atomicMin((unsigned long long *)lds_min, 0ull); }
   __syncthreads();
   const long long r = *lds_min;
   __syncthreads();
   return r; }          // 0 => no provably safe restart row inside the margin

__host__ __device__ inline unsigned lds_bstride(int tile_rows) { unsigned v = (unsigned)(tile_rows + kScreenHalo) / 8 + 8; while ((v / 4) % 32 != 22) v += 8; return v; }
__host__ __device__ inline unsigned lds_ldstride(int tile_rows) { unsigned v = (unsigned)(tile_rows + kScreenHalo); while ((v / 4) % 32 != 18) v += 8; return v; }
struct LdsLayout {
   unsigned bits, ldpos, heights, recs, nrec, walkers, fdiff, total; };
__host__ __device__ inline unsigned lds_align16(unsigned v) { return (v + 15u) & ~15u; }
__host__ __device__ inline LdsLayout lds_layout(const DevCfg &c) {
   LdsLayout L;
   const unsigned ntrks = (unsigned)c.ntrks, nst = (unsigned)c.nscreens * ntrks, nwalk = (unsigned)c.nparm * ntrks;
   const unsigned T = (unsigned)c.tile_rows;
   unsigned off = lds_align16(ntrks * (unsigned)c.ldw * 2u + 16u);          // (ldw rows of ntrks samples, + one vector of slack)
   const bool zeros = c.find_zeros && !c.differentiate;                      // no screen: the space holds the sub-segment records of zeros_tile_parallel
   L.bits = off;      off = lds_align16(off + (zeros ? ntrks * (T / (unsigned)kZcSub) * (unsigned)sizeof(ZcLane) : nst * 5u * lds_bstride((int)T)));
   L.ldpos = off;     off = lds_align16(off + (zeros ? 0u : nst * 2u * lds_ldstride((int)T)));
   L.fdiff = off;     if (c.differentiate && !c.find_zeros) off = lds_align16(off + ntrks * (unsigned)c.ldw * 4u + 32u);
   L.heights = off;   off = lds_align16(off + nwalk * 10u * 4u);
   L.recs = off;      off = lds_align16(off + nwalk * (unsigned)c.rec_cap * (unsigned)sizeof(Rec));
   L.nrec = off;      off = lds_align16(off + nwalk * 4u);
   L.walkers = off;   off = lds_align16(off + nwalk * (unsigned)sizeof(Walker));
   L.total = off;
   return L; }

__global__ void __launch_bounds__(kDecodeThreads, 2) k_decode(const DevCfg *__restrict__ cfgp, const int16_t *__restrict__ rows,
                                                           long long nrows, long long row_base,
                                                           rtfe_burst *__restrict__ bursts, BurstScratch *__restrict__ scratch,
                                                           uint32_t *__restrict__ counts, rtfe_event *__restrict__ events,
                                                           uint32_t parmset_mask, int screen_off, int single_exact,
                                                           int mode, BurstCtl *__restrict__ ctl) {
#ifdef RTFE_CPU_EMUL
   unsigned char *smem = g_dyn_smem;            // tests/cpu_emul only
#else
   extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
   if (mode == kDecodeRedo && scratch->seg_failed == 0) return;      // nothing for the sample path to redo (k_publish counted): the usual case costs one load
   __shared__ DevCfg cfg;
   __shared__ int s_burst;
   __shared__ long long s_min;
   __shared__ unsigned int s_flags;
   __shared__ int s_off[kMaxScreens * RTFE_MAXTRKS + 1];
   for (int i = threadIdx.x; i < (int)(sizeof(DevCfg) / 4); i += blockDim.x) reinterpret_cast<int *>(&cfg)[i] = reinterpret_cast<const int *>(cfgp)[i];
   __syncthreads();
   if (screen_off && threadIdx.x == 0)
      for (int s = 0; s < cfg.nscreens; ++s) { cfg.screen[s].rise_i = -70000; cfg.screen[s].minpk_i = -1; }
   if (screen_off) for (int i = threadIdx.x; i < cfg.nparm; i += blockDim.x) { cfg.parm[i].screen_rise_v = -1; cfg.parm[i].screen_minpk_v = -1; }
   __syncthreads();
   const int ntrks = cfg.ntrks;
   const int ldw = cfg.ldw;
   Ctx cx;
   cx.cfg = &cfg;
   cx.row_base = row_base;
   cx.tile.x = reinterpret_cast<int16_t *>(smem);
   cx.tile.ldw = ldw; cx.tile.halo = cfg.halo_rows; cx.tile.colof = cfg.trk_to_head;
   cx.tile.ntrks = ntrks;
   cx.tile.skew = cfg.skew;
   const LdsLayout L = lds_layout(cfg);
   const bool diffpeak = cfg.differentiate && !cfg.find_zeros;     // -differentiate without -zeros: literal float detector
   cx.tile.bits = smem + L.bits;
   cx.tile.fd = reinterpret_cast<float *>(smem + L.fdiff);
   cx.tile.bstride = (int)lds_bstride(cfg.tile_rows);
   cx.tile.ldpos = smem + L.ldpos; cx.tile.ldstride = (int)lds_ldstride(cfg.tile_rows);
   float *heights_all = reinterpret_cast<float *>(smem + L.heights);
   // walker w of this workgroup -> thread: spread over the 4 waves so every SIMD issues for some walkers
   const int nwalk = cfg.nparm * ntrks;
   const int nwaves = blockDim.x >> 6;
   const int my_w = (threadIdx.x & 63) * nwaves + (threadIdx.x >> 6);
   const bool is_walker = my_w < nwalk;
   const int pidx = is_walker ? my_w / ntrks : 0, trk = is_walker ? my_w - pidx * ntrks : 0;
   const bool active = is_walker && ((parmset_mask >> pidx) & 1);
   cx.heights = heights_all + (size_t)(is_walker ? my_w : 0) * 10;
   Rec *recs_all = reinterpret_cast<Rec *>(smem + L.recs);
   int *nrec_all = reinterpret_cast<int *>(smem + L.nrec);
   Walker *walkers = reinterpret_cast<Walker *>(smem + L.walkers);       // [nwalk]
   cx.rec_cap = cfg.rec_cap;
   cx.recs = recs_all + (size_t)(is_walker ? my_w : 0) * cfg.rec_cap;

   // restart row of a zone-started burst: load the last kMarginRows rows of its zone, screen them, find the last
   // forced rescan per (parmset, track) (find_reset).  Both the burst itself and its predecessor (which must know
   // where to stop) evaluate this, on the same rows, so they agree.  Returns -1 when no safe row exists.
   auto zone_reset = [&](const rtfe_burst &Z) -> long long {
      if (Z.zone_end - Z.zone_first < kMarginRows + 64) return -1;
      if (cfg.find_zeros || cfg.differentiate) return Z.zone_end - kMarginRows;       // any restart inside the zone is equivalent
      cx.tile.row0 = Z.zone_end - kMarginRows; cx.tile.nrows = kMarginRows; cx.tile.reset = -(1ll << 40);
      __syncthreads();
      load_tile(&cfg, cx.tile, rows, nrows);
      __syncthreads();
      run_screens(&cfg, cx.tile, false);
      __syncthreads();
      const long long r = find_reset(&cfg, cx.tile, &s_min);
      return (r <= 0 || r < Z.zone_first) ? -1 : r; };

   // mode: kDecodeAll = every burst of the table (PE, GCR, -differentiate, exact scans); kDecodeRedo = the bursts the record chains gave up
   for (;;) {
      if (threadIdx.x == 0) { s_burst = atomicAdd(mode == kDecodeRedo ? &scratch->queue_resume : &scratch->queue, 1); s_flags = 0; }
      __syncthreads();
      const int b = s_burst;
      if (b >= scratch->nbursts) break;
      if (mode == kDecodeRedo && ctl[b].status != kBurstNeedsFull) { __syncthreads(); continue; }
      const int nb = scratch->nbursts_total;
      const rtfe_burst B = bursts[b];
      const bool exact = B.flags & RTFE_F_EXACT_START;
      const bool last = b + 1 >= nb;
      const bool has_tail = !last && !single_exact;
      cx.events = events + B.event_base;
      cx.cap = B.event_cap;
      unsigned int bflags = B.flags;
      long long reset, stop;
      const long long hard_end = single_exact ? (B.end_sample < nrows ? B.end_sample : nrows) : nrows;
      long long g_first;
      {
         // ---- where this burst restarts, and where the next one does (= where this one stops) ----
         reset = B.reset_sample;
         if (mode == kDecodeRedo) { reset = ctl[b].reset; bflags = ctl[b].bflags; }          // k_zones decided (the chains' neighbours rely on it)
         else if (!exact) {
            reset = zone_reset(B);
            if (reset < 0) { reset = B.zone_end - kMarginRows; bflags |= RTFE_F_UNSAFE; } }
         stop = hard_end;
         if (mode == kDecodeRedo) stop = ctl[b].stop;
         else if (has_tail) {
            const rtfe_burst NB = bursts[b + 1];
            stop = zone_reset(NB);
            if (stop < 0) stop = NB.zone_end - kMarginRows;
            // the dead part of the gap: nothing a block decoder still listens to lies more than tail_rows into the quiet zone
            if (cfg.tail_rows > 0 && NB.zone_first + cfg.tail_rows < stop) stop = NB.zone_first + cfg.tail_rows; }
         // ---- walker init: init_trackstate + init_trackpeak_state (src/decoder.c:413-455) ----
         // (the walker's state lives in LDS between tiles so that the all-lane phases do not carry it in registers)
         if (is_walker) {
            Walker w = {};
            w.start = reset + trk; w.next = reset; w.blind_until = -1; w.fast = false;
            w.agc_gain = 1.0f; w.v_avg_height = 4.0f; w.z_firstzero = -1; w.z_lastzero = -1;
            update_thresholds(w, cfg.parm[pidx], cfg.lsb_per_volt);
            walkers[my_w] = w;
            for (int i = 0; i < 10; ++i) cx.heights[i] = 0; }
         g_first = reset / cfg.tile_rows; }
      cx.tile.reset = reset;
      // ---- tiles of the tape-global grid that intersect [reset, stop) ----
      const long long T = cfg.tile_rows;
      for (long long g = g_first; g * T < stop; ++g) {
         const long long tile0 = g * T;
         const long long tn = (tile0 + T <= nrows) ? T : nrows - tile0;
         if (tn <= 0) break;
         long long c0 = 0, c1 = 0, c2 = 0;
         cx.tile.row0 = tile0; cx.tile.nrows = (int)tn;
         __syncthreads();
         if (cfg.debug) c0 = clock64();
         // ---- full path: samples into LDS, screen, exact walkers ----
         load_tile(&cfg, cx.tile, rows, nrows);
         __syncthreads();
         if (cfg.debug) c1 = clock64();
         if (diffpeak) differentiate_tile(&cfg, cx.tile);         // (no candidate screen on the differentiated signal)
         else if (!cfg.find_zeros) run_screens(&cfg, cx.tile, true);       // (nor for the zero-crossing detectors)
         __syncthreads();
         if (cfg.debug) c2 = clock64();
         long long c2c = 0;
         cx.nrec = 0;
         const bool zc_par = cfg.find_zeros && !cfg.differentiate && cfg.zc_parallel && (parmset_mask & 1u)
                             && (size_t)ntrks * (cfg.tile_rows / kZcSub) * sizeof(ZcLane) <= (size_t)(L.heights - L.bits)
                             && ntrks * (cfg.tile_rows / kZcSub) <= (int)blockDim.x;
         // (the sub-segment records live where the peak path keeps its screen maps: -zeros has no screen)
         if (zc_par) zeros_tile_parallel(cx, walkers, reinterpret_cast<ZcLane *>(smem + L.bits), s_off, stop, cfg.debug ? scratch->dbg2 : nullptr);     // (s_off: per-track verdicts)
         if (zc_par && cfg.debug && (int)threadIdx.x < ntrks) { atomicAdd(&scratch->dbg2[4], 1ull); if (s_off[threadIdx.x]) atomicAdd(&scratch->dbg2[5], 1ull); }
         if (active) {
            Walker w = walkers[my_w];
            if (cfg.find_zeros) { if (pidx == 0 && !(zc_par && s_off[trk])) { if (cfg.differentiate) walk_diffzeros(w, cx, trk, stop); else walk_zeros(w, cx, trk, stop); } }
            else if (diffpeak) walk_diffpeak(w, cx, pidx, trk, stop);
            else walk(w, cx, pidx, trk, stop);
            walkers[my_w] = w; }
         if (is_walker) nrec_all[my_w] = cx.nrec;
         __syncthreads();
         if (cfg.debug) c2c = clock64();
         for (int w2 = 0; w2 < nwalk; ++w2)                      // all lanes: refinement, volt conversion, event stores
            finalize_records(cx, recs_all + (size_t)w2 * cfg.rec_cap, nrec_all[w2], w2 / ntrks, w2 % ntrks, threadIdx.x, blockDim.x);
         if (cfg.debug) {
            __syncthreads();
            if (threadIdx.x == 0) {
               const long long c3 = clock64();
               atomicAdd(&scratch->dbg[0], (unsigned long long)(c1 - c0)); atomicAdd(&scratch->dbg[1], (unsigned long long)(c2 - c1));
               atomicAdd(&scratch->dbg[2], (unsigned long long)(c3 - c2)); atomicAdd(&scratch->dbg[3], 1ull);
               atomicAdd(&scratch->dbg[5], (unsigned long long)(c2c - c2));
               atomicAdd(&scratch->dbg[6], (unsigned long long)(c3 - c2c)); } } }
      // ---- publish ----
      if (mode != kDecodeAll && threadIdx.x == 0) ctl[b].status = kBurstDone;
      if (is_walker) {
         const unsigned int ne = walkers[my_w].nevents;
         unsigned int wf = walkers[my_w].flags;
         if (cfg.find_zeros && pidx == 0 && active) {                  // history a restart would not have
            const Walker &wz = walkers[my_w];
            const bool dirty = cfg.differentiate ? (wz.z_up_pending || wz.z_dn_pending)
                                                 : (wz.z_up_pending || wz.z_dn_pending || wz.z_top >= cfg.zc_peak_i || wz.z_bot <= -cfg.zc_peak_i);
            if (dirty) wf |= RTFE_F_STATE_AT_END; }
         counts[((size_t)b * cfg.nparm + pidx) * ntrks + trk] = active ? (ne < cx.cap ? ne : cx.cap) : 0;
         if (wf) atomicOr(&s_flags, wf); }
      __syncthreads();
      if (threadIdx.x == 0) {
         bursts[b].reset_sample = reset;
         // last attempt-start row this burst stands for.  Peak detector: the restart row itself.  -zeros: any row of the
         // zone (no window).  Differentiated peaks: the zone is exact zeros after the dead band, so any start that leaves the
         // window time to fill (the timestamp formula depends on a full window, src/decoder.c:732) before the zone ends.
         int wmax = 0;
         for (int sidx = 0; sidx < cfg.nscreens; ++sidx) wmax = max(wmax, cfg.screen[sidx].W);
         bursts[b].safe_last = (bflags & (RTFE_F_UNSAFE)) ? -1
                             : ((cfg.find_zeros && !exact) ? B.zone_end - ntrks - 2
                             : ((diffpeak && !exact) ? max((long long)reset, (long long)B.zone_end - (wmax + cfg.maxskew + ntrks + 4)) : reset));
         bursts[b].end_sample = stop < hard_end ? stop : hard_end;
         bursts[b].flags = bflags | s_flags; }
      __syncthreads(); } }

}  // namespace rtfe
