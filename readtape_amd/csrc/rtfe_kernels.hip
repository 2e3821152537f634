// rtfe_kernels.hip — the MI355X (gfx950 / CDNA4) analog front end.
//
// Seven kernels, all integer / fp32 streaming work bound by HBM and LDS, no MFMA:
//
//   k_quiet    one pass over the interleaved int16 TBIN payload: 1 bit per KiB of payload that says
//              "every sample of every track is inside the quiet band".              [HBM-bound]
//   k_bursts   turns runs of quiet bits into inter-block zones -> the burst table; marks the tiles deep inside a gap
//              that nothing will ever walk (one workgroup).
//   k_screen   dense and stateless, one workgroup per 512-row tile: flat copy of the rows into LDS, a data-
//              parallel sliding-window max/min "candidate screen", the reference's stale window minimum,
//              candidate RUNS (everything the sequential detector reads, as int16 codes) -> HBM.
//   k_decode   (a) burst heads: restart row inside the quiet zone, the literal start-up path, hand-over to
//              k_walk; (c) whatever k_walk gives back; and the whole job for PE / GCR, -zeros, -differentiate and exact
//              re-scans: one lane per (parameter set, track) replays the reference's sequential detector EXACTLY
//              (blind countdown, stale-minimum rescans, AGC schedule, half-sample refinement) on samples in LDS.
//              -zeros: a lane per (track, 64-row sub-segment) with a verified warm-up instead.
//   k_walk     (b) the sequential pass in the common case: one small workgroup per burst SEGMENT walks the candidate
//              runs tile after tile - decisions for all runs in parallel against wide threshold bands, the countdown
//              chain and the three-flop AGC recurrence per walker, events by all lanes.
//   k_segs     cuts long bursts into 48-tile segments that k_walk walks concurrently from guessed states,
//   k_stitch   accepts them where each segment's start state equals its predecessor's end state bit for bit, and
//              compacts their event slots.
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
   int   queue_walk;             // ... of k_walk
   int   queue_resume;           // ... of the second k_decode pass
   int   queue_seg, nsegs;       // segment queue of k_walk (segment mode), segments of this scan (k_segs)
   int   queue_stitch;           // ... of k_stitch
   int   seg_failed;             // bursts whose segments did not join (statistics)
   int   hard_count;             // candidates k_sift deferred to k_sift_hard (cleared with the scratch block when rtfe_scan starts)
   int   pad[6];
   unsigned long long dbg[8];    // at byte 64: optional per-phase cycle counters of k_decode (DevCfg::debug)
   unsigned long long pool_cursor;   // at byte 128: next free PackedRun of the pool (k_screen)
   unsigned long long dbg2[8];   // dbg[8..15] (contiguous with dbg through pool_cursor is NOT assumed: indexed separately)
   unsigned long long why[8];    // optimistic-walk failure reasons (debug)
   unsigned long long scr[8];    // k_screen per-phase cycle counters (debug)
};
static_assert(sizeof(BurstScratch) <= kScratchBytes, "scratch region too small");

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

__global__ void __launch_bounds__(1024) k_bursts(const u64 *__restrict__ qwords, long long nwords, long long nchunks,
                                                 long long nrows, long long own_rows, int ntrks, int gap_chunks, int first_is_start,
                                                 float cap_frac, int nparm, long long event_capacity,
                                                 rtfe_burst *__restrict__ bursts, long long max_bursts,
                                                 BurstScratch *__restrict__ scratch, int32_t *__restrict__ nbursts_out,
                                                 unsigned int *__restrict__ dead, long long ntiles, int tile_rows, int tail_rows) {
   __shared__ int lds[32];
   for (long long i = threadIdx.x; i < (ntiles + 31) / 32; i += blockDim.x) dead[i] = 0;
   __shared__ int s_base;
   __shared__ u64 s_ebase;
   if (threadIdx.x == 0) {
      s_base = 0;
      // a tape (or shard) that does not begin inside a qualifying zone gets an exact-start burst at row 0
      bool starts_quiet = true;
      for (int c = 0; c < gap_chunks; ++c) if (!quiet_at(qwords, c, nchunks)) { starts_quiet = false; break; }
      if (first_is_start && !starts_quiet && max_bursts > 0) {
         rtfe_burst b = {};
         b.zone_first = 0; b.zone_end = 0; b.reset_sample = 0; b.safe_last = 0; b.flags = RTFE_F_EXACT_START;
         bursts[0] = b;
         s_base = 1; } }
   __syncthreads();
   constexpr int kPer = 4;                                          // consecutive words per thread per round (fewer block scans)
   for (long long w0 = 0; w0 < nwords; w0 += 1024 * kPer) {
      u64 ends[kPer];
      int cnt = 0;
      #pragma unroll
      for (int j = 0; j < kPer; ++j) {
         const long long w = w0 + (long long)threadIdx.x * kPer + j;
         ends[j] = 0;
         if (w < nwords) {
            const u64 q = qwords[w];
            const u64 qn = (w + 1 < nwords) ? qwords[w + 1] : 0;
            const u64 next = (q >> 1) | (qn << 63);         // bit c = quiet[c+1]
            u64 cand = q & ~next;                           // quiet and successor not quiet
            while (cand) {
               const int bit = __ffsll((long long)cand) - 1;
               cand &= cand - 1;
               const long long c = w * 64 + bit;
               bool ok = true;
               for (int k = 1; k < gap_chunks; ++k) if (!quiet_at(qwords, c - k, nchunks)) { ok = false; break; }
               if (ok) ends[j] |= 1ull << bit; } }
         cnt += __popcll(ends[j]); }
      int total;
      int off = block_excl_scan_1024(cnt, lds, &total);
      const int base = s_base;
      #pragma unroll
      for (int j = 0; j < kPer; ++j) {
         const long long w = w0 + (long long)threadIdx.x * kPer + j;
         u64 e = ends[j];
         while (e) {
            const int bit = __ffsll((long long)e) - 1;
            e &= e - 1;
            const long long c1 = w * 64 + bit + 1;          // one past the last quiet chunk
            // zone start: walk back over quiet chunks a 64-bit word at a time
            long long c0 = c1 - 1;
            for (;;) {
               if (c0 == 0) break;
               const long long pw = (c0 - 1) >> 6; const int pb = (int)((c0 - 1) & 63);
               // bits pb..0 of word pw, shifted so that bit pb becomes bit 63: count the leading run of ones
               const u64 run = ~(qwords[pw] << (63 - pb));
               const int ones = run ? __clzll((long long)run) : 64;
               const int take = ones < pb + 1 ? ones : pb + 1;
               c0 -= take;
               if (take < pb + 1) break; }
            const long long idx = (long long)base + off++;
            if (idx < max_bursts) {
               rtfe_burst b = {};
               long long zf = c0 * kChunkRows;                          // first row of the zone (chunks are groups of 64 rows)
               long long ze = c1 * kChunkRows;                          // one past its last row
               if (ze > nrows) ze = nrows & ~63ll;
               b.zone_first = zf; b.zone_end = ze; b.reset_sample = -1; b.safe_last = -1;
               bursts[idx] = b; } } }
      __syncthreads();
      if (threadIdx.x == 0) s_base = base + total;
      __syncthreads(); }
   int nb = s_base;
   if (nb > max_bursts) nb = (int)max_bursts;
   // time shards: keep the bursts that start in the owned rows, plus one more as the bound of the last of them
   __shared__ int s_owned;
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
   // drop zones too short to hold a head tile (zone_end - zone_first < margin + 64): mark by flags later in decode
   // ---- second pass: coarse extents, event capacities and region bases (parallel prefix sum) ----
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
         bursts[b].event_cap = (uint32_t)cap;
         // the dead part of the gap in front of this burst: the previous burst's walkers stop tail_rows into the zone
         // (k_decode), this burst's restart lies in the zone's last kMarginRows rows; tiles entirely in between are
         // never walked, so k_screen need not screen them
         if (tail_rows > 0 && !(bursts[b].flags & RTFE_F_EXACT_START)) {
            const long long T = tile_rows;
            const long long g0 = (bursts[b].zone_first + tail_rows + T - 1) / T, g1 = (bursts[b].zone_end - kMarginRows) / T;   // [g0, g1)
            for (long long g = g0; g < g1 && g < ntiles; ++g) atomicOr(&dead[g >> 5], 1u << (g & 31)); } }
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
   if (threadIdx.x >= 24 && threadIdx.x < 32) scratch->why[threadIdx.x - 24] = 0;
   }       // (scratch->scr: cleared with the rest of the scratch block when rtfe_scan starts)

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

// what a burst's walkers are between the kernels of one scan (workspace, one per (burst, walker))
struct WalkState { Walker w; float heights[10]; };

// A detection whose (cheap-to-defer) half-sample refinement, volt conversion and event store are done
// after the walk by all lanes (finalize_tile): the sequential walker keeps only what feeds back.
struct alignas(8) Rec { unsigned int idx; unsigned short n_rel; unsigned char ld, kind; float g; short val, prev, next, pad; int pad2; };   // 24 bytes

// the record walk's deferred detections: 16 bytes, event index = (walker's event count at the tile's start) + position
struct alignas(8) Rec16 { unsigned short n_rel; unsigned char ld, kind; float g; short val, prev, next, pad; };

struct Ctx {               // per-workgroup constants for the walkers
   const DevCfg *cfg;
   Tile tile;
   long long row_base;     // absolute row of d_rows[0]
   float *heights;         // LDS [walkers][10]
   rtfe_event *events;     // this burst's regions
   unsigned int cap;
   struct Rec *recs;       // LDS [rec_cap] deferred events of this lane's walker for the current tile
   int     rec_cap;
   int     rec_cap16;      // ... in Rec16 units (the same LDS space)
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

// all lanes: the record walk's queued detections of ALL walkers of one tile -> events.  off[w] = first record of walker w
// in the flattened numbering (off[nwalk] = total), idx0[w] = its event count at the tile's start.
__device__ __forceinline__ void finalize_records16(const Ctx &cx, const unsigned char *recs_all, int stride_bytes, const int *off, const int *idx0,
                                                   int nwalk, int lane, int nlanes) {
   const DevCfg *cfg = cx.cfg;
   const int total = off[nwalk];
   for (int i = lane; i < total; i += nlanes) {
      int w2 = 0;
      while (off[w2 + 1] <= i) ++w2;
      const int k = i - off[w2];
      const unsigned long long *src = reinterpret_cast<const unsigned long long *>(recs_all + (size_t)w2 * stride_bytes) + 2 * k;
      const unsigned long long r0 = src[0], r1 = src[1];
      const int n_rel = (int)(r0 & 0xffff), ld = (int)((r0 >> 16) & 0xff);
      const bool is_top = ((r0 >> 24) & 0xff) == 0;
      const float g = __uint_as_float((unsigned)(r0 >> 32));
      const int val = (int)(short)(r1 & 0xffff), prev = (int)(short)((r1 >> 16) & 0xffff), next = (int)(short)((r1 >> 32) & 0xffff);
      const int pidx = w2 / cfg->ntrks, trk = w2 - pidx * cfg->ntrks;
      const int adjcode = refine_code(cfg, val, prev, next, g, is_top);
      store_event(cx, pidx, trk, (unsigned)(idx0[w2] + k), cx.tile.row0 + n_rel, volt(val, cfg->maxvolts), g, is_top, adjcode, ld); } }

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

// ---- k_screen only: the reference's (possibly stale) window minimum at every row ----
// The detector's min/max tracking does not depend on its decisions (src/decoder.c:760-775 runs before the countdown
// test): the maximum is always exact, and the minimum is refreshed exactly at the rows where the sample leaving the
// window equals the tracked maximum ("A-sync" rows: data alone decides, the screen's bitmap 2) or the tracked minimum.
// Between two A-sync rows the minimum therefore follows a chain that starts from the true minimum at the first of
// them: it stays the same sample until that sample leaves the window, where a rescan makes it the true minimum again.
// This pass rewrites ldmap(screen,1,trk)[row] from "left_distance of the true minimum" into "left_distance of the
// reference's minimum" for the rows between A-sync rows (0 where no A-sync row lies within reach: unknown).
// One call handles the gaps that START in one 8-row strip.
__device__ __forceinline__ u64 bits_from(const u64 *map, int n) {       // bit k of the result = bit (n + k) of the bitmap, n >= -kScreenHalo
   const int wd = n >> 6, sh = n & 63;
   const u64 lo = map[wd] >> sh;
   return sh ? (lo | (map[wd + 1] << (64 - sh))) : lo; }              // (one spare word behind every bitmap row)

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
      if (h + l > q) return l - (q - h);                             // still the same sample at row q
      h += l; } }                                                    // it left the window at row h + l: rescan there

// rewrite the bytes of the bottom-candidate rows of one strip that are not A-sync rows
__device__ __forceinline__ void fill_stale(const Tile &tl, int screen, int trk, int strip) {
   const u64 *am = tl.map(screen, 2, trk), *bm = tl.map(screen, 1, trk);
   const int sh = (strip & 7) * 8;
   unsigned need = (unsigned)((bm[strip >> 3] & ~am[strip >> 3]) >> sh) & 0xffu;
   if (strip * 8 + 8 > tl.nrows) need &= (1u << (tl.nrows - strip * 8)) - 1;
   unsigned char *ldb = tl.ldmap(screen, 1, trk);
   #pragma nounroll
   while (need) {
      const int j = __ffs((int)need) - 1;
      need &= need - 1;
      const int n = strip * 8 + j;
      ldb[n] = (unsigned char)stale_ld(am, ldb, n); } }

// ---- k_screen only: run starts.  A RUN = consecutive candidate rows of one kind sharing the same extreme; a row where
// both kinds are candidates is a one-row run of each kind.  One call = the 8 rows of one strip of one (screen, track):
// start bytes into the two extra bitmaps (kinds 3 and 4).
__device__ __forceinline__ void run_starts(const Tile &tl, int screen, int trk, int strip) {
   const int ntb = tl.ntrks * tl.bstride;
   unsigned char *row = tl.bits + (screen * 5) * ntb + trk * tl.bstride + kScreenHalo / 8 + strip;
   const int r0 = strip * 8;
   const int nvalid = tl.nrows - r0 >= 8 ? 8 : tl.nrows - r0;
   const unsigned vm = (1u << nvalid) - 1;
   const unsigned t = row[0] & vm, b = row[ntb] & vm;
   // bit j+1 of t9/b9 = row r0+j, bit 0 = the row in front of the strip (never a predecessor at the tile's first row)
   const unsigned t9 = (t << 1) | (strip > 0 ? (row[-1] >> 7) & 1u : 0u), b9 = (b << 1) | (strip > 0 ? (row[ntb - 1] >> 7) & 1u : 0u);
   const unsigned ot = t9 & ~b9, ob = b9 & ~t9;                     // single-kind rows
   unsigned st = t, sb = b;                                         // a candidate row starts a run unless it continues one
   unsigned ct = (ot >> 1) & ot & 0xffu, cb = (ob >> 1) & ob & 0xffu;    // bit j: rows r0+j-1 and r0+j are single-kind candidates
   if (ct | cb) {
      const unsigned char *ldt = tl.ldmap(screen, 0, trk) + r0, *ldb = tl.ldmap(screen, 1, trk) + r0;
      #pragma nounroll
      while (ct) { const int j = __ffs((int)ct) - 1; ct &= ct - 1; if (ldt[j] != 0 && ldt[j] == ldt[j - 1] - 1) st &= ~(1u << j); }
      #pragma nounroll
      while (cb) { const int j = __ffs((int)cb) - 1; cb &= cb - 1; if (ldb[j] != 0 && ldb[j] == ldb[j - 1] - 1) sb &= ~(1u << j); } }
   row[3 * ntb] = (unsigned char)st;
   row[4 * ntb] = (unsigned char)sb; }

// rows in the run that starts at row n of kind `kind` (cm = candidate bitmap of that kind, sm = its start bitmap)
__device__ __forceinline__ int run_length(const u64 *cm, const u64 *sm, int n, int nrows) {
   const u64 c = bits_from(cm, n + 1) & ~bits_from(sm, n + 1);       // continuation rows behind n
   int len = 1 + (~c ? __ffsll((long long)~c) - 1 : 64);
   if (n + len > nrows) len = nrows - n;
   return len; }

// ---- k_screen only: the runs that start in one strip of one (screen, track), in the detector's order (row, top
// before bottom): one descriptor each into the tile's run table, and the strip's unit count as the result.
// descriptor: st | kind << 8 | n << 16 | (u64)nr << 32 | (u64)relrun << 40 | (u64)relmarg << 50   (run / margin-unit index inside the strip)
// result: runs | margin units << 16
__device__ __forceinline__ int list_runs(const Tile &tl, int st, int screen, int trk, int strip, u64 *runtab, int *nruns, int tabcap) {
   const u64 *tm = tl.map(screen, 0, trk), *bm = tl.map(screen, 1, trk), *stm = tl.map(screen, 3, trk), *sbm = tl.map(screen, 4, trk);
   const int sh = (strip & 7) * 8;
   const unsigned stt = (unsigned)(stm[strip >> 3] >> sh) & 0xffu, sb = (unsigned)(sbm[strip >> 3] >> sh) & 0xffu;
   unsigned any = stt | sb;
   int relrun = 0, relmarg = 0;
   #pragma nounroll
   while (any) {
      const int j = __ffs((int)any) - 1;
      any &= any - 1;
      const int n = strip * 8 + j;
      #pragma nounroll
      for (int kind = (stt >> j) & 1 ? 0 : 1; kind <= (int)((sb >> j) & 1); ++kind) {
         const int nr = run_length(kind ? bm : tm, kind ? sbm : stm, n, tl.nrows);
         const int slot = atomicAdd(nruns, 1);
         if (slot < tabcap) runtab[slot] = (u64)(unsigned)(st | (kind << 8) | (n << 16)) | ((u64)nr << 32) | ((u64)relrun << 40) | ((u64)relmarg << 50);
         ++relrun; relmarg += (nr + 2) >> 2; } }
   return relrun | (relmarg << 16); }

// ---- k_screen only: the units of one run -> its place in the list (HBM) ----
// Four lanes share a run: item 0 = the header, item k = margin unit k-1 (rows n + 4k-3 .. n + 4k); lane `sub` builds
// items sub, sub+4, ...
__device__ __forceinline__ void build_run(const Tile &tl, const DevCfg *cfg, int screen, int trk, int W, int n, int kind, int nr,
                                          int4 *hdr, int moff, int4 *marg, int sub) {
   const Col yb = tile_col(tl, trk, cfg->skew[trk]);
   const int ld0 = tl.ldmap(screen, kind, trk)[n];
   const int m = ld0 ? yb[n - W + ld0] : 0;                         // the extreme: lo + left_distance - 1
   const int nitems = 1 + ((nr + 2) >> 2);
   #pragma nounroll
   for (int it = sub; it < nitems; it += 4) {
      int pr[4] = {0, 0, 0, 0};
      const int j0 = it == 0 ? 0 : 4 * it - 3, nj = it == 0 ? 1 : min(4, nr - j0);
      if (ld0) {
         const int16_t *pr_ = yb.p + (n + j0) * yb.P, *pl_ = pr_ - (W - 1) * yb.P;      // right / left window edge at the item's first row
         #pragma unroll
         for (int c = 0; c < 4; ++c) {
            if (c < nj) {
               const int L = *pl_, R = *pr_;
               pl_ += yb.P; pr_ += yb.P;
               int dl = kind ? L - m : m - L, dr = kind ? R - m : m - R;
               dl = dl < 0 ? 0 : dl; dr = dr < 0 ? 0 : dr;
               pr[c] = dl | (dr << 16); } } }
      if (it == 0) {
         int prev = 0, next = 0;
         if (ld0) { const int p = n - W + ld0; prev = yb[p - 1]; next = yb[p + 1]; }
         int4 q;
         q.x = n | (nr << 11) | (kind << 17) | (moff << 18);
         q.y = (m & 0xffff) | (ld0 << 16);
         q.z = (prev & 0xffff) | (next << 16);
         q.w = pr[0];
         *hdr = q; }
      else marg[it - 1] = make_int4(pr[0], pr[1], pr[2], pr[3]); } }

// i / n for the item loops (n fixed per tile, i < 2^24): one multiply instead of an integer division
struct FastDiv {
   unsigned n, M;
   __device__ __forceinline__ explicit FastDiv(int d) : n((unsigned)d), M(d > 1 ? 0xFFFFFFFFu / (unsigned)d + 1u : 0u) {}
   __device__ __forceinline__ int div(int i) const { return n > 1 ? (int)__umulhi((unsigned)i, M) : i; } };

// ---- candidate screen: one thread = one strip of 8 consecutive rows of one track ----
// window max/min by prefix/suffix decomposition around the strip start (van Herk with one block edge)
__device__ __forceinline__ int screen_strip(const Tile &tl, const DevScreen &sc, int screen, int trk, int strip) {
   // Keys carry the position so that max/min also yield the FIRST window element equal to the extreme
   // (what refine_peak looks for, src/decoder.c:707-708): r = index relative to the strip's leftmost
   // window element (s0 - W + 1);  kmax = v<<8 | (255 - r)  (max -> largest v, then smallest r),
   // kmin = v<<8 | r  (min -> smallest v, then smallest r).
   const int W = sc.W;
   const int d = tl.skew[trk];
   const Col base = tile_col(tl, trk, d);                          // y(n) = base[n - row0] in the regular regime
   const int s0 = strip * kStrip;
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
         ldt |= (u64)((255 - (kx & 255)) + 1) << (8 * i);
         ldb |= (u64)((kn & 255) + 1) << (8 * i); } }
   const int ntb2 = tl.ntrks * tl.bstride;
   unsigned char *o = tl.bits + (screen * 5) * ntb2 + trk * tl.bstride + kScreenHalo / 8 + strip;     // strip >= -kScreenHalo/8
   o[0] = (unsigned char)topb;
   o[ntb2] = (unsigned char)botb;
   o[2 * ntb2] = (unsigned char)resb;
   reinterpret_cast<u64 *>(tl.ldmap(screen, 0, trk))[strip] = ldt;
   reinterpret_cast<u64 *>(tl.ldmap(screen, 1, trk))[strip] = ldb;
   return topb | botb; }

// cooperative tile load: rows [row0 - halo, row0 + nrows) of the AoS payload -> LDS, as they are (16-byte vectors;
// the tile starts on a multiple of 8 rows, so vectors are aligned in HBM and in LDS).  -invert negates on the way.
__device__ __forceinline__ void load_tile(const DevCfg *cfg, Tile &tl, const int16_t *__restrict__ rows, long long total_rows) {
   const int ntrks = cfg->ntrks;
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

// act / nact (k_screen only): the strips of the tile proper that hold candidates, compacted (st << 16 | strip) so that
// the sparse passes behind the screen keep every lane busy; stripcnt of every strip is cleared on the way
__device__ __forceinline__ void run_screens(const DevCfg *cfg, const Tile &tl, bool with_halo, unsigned int *act = nullptr, int *nact = nullptr,
                                            unsigned int *stripcnt = nullptr, int smax = 0, int only = -1) {      // only >= 0: that screen alone, into LDS slot 0 (k_screen)
   const int hs = with_halo ? kScreenHalo / kStrip : 0;            // k_screen also screens the rows in front of the tile
   const int nstrips = (tl.nrows + kStrip - 1) / kStrip + hs;
   const int per_screen = nstrips * cfg->ntrks;
   const FastDiv fd(cfg->ntrks);
   for (int s = only >= 0 ? only : 0; s < (only >= 0 ? only + 1 : cfg->nscreens); ++s)
      for (int i0 = 0; i0 < per_screen; i0 += blockDim.x) {         // (uniform trip count: the ballot below needs whole waves)
         const int i = i0 + (int)threadIdx.x;
         const bool valid = i < per_screen;
         int t = 0, strip = 0, any = 0;
         if (valid) {                                               // consecutive lanes = the tracks of one strip: consecutive LDS words
            const int q = fd.div(i);
            t = i - q * cfg->ntrks; strip = q - hs;
            any = screen_strip(tl, cfg->screen[s], only >= 0 ? 0 : s, t, strip); }
         if (act) {
            const int st = (only >= 0 ? 0 : s) * cfg->ntrks + t;
            const bool mine = valid && strip >= 0;
            if (mine) stripcnt[st * smax + strip] = 0;
            const bool on = mine && any != 0;
            const u64 bal = __ballot(on);
            const int lane = threadIdx.x & 63;
            int base = 0;
            if (lane == 0 && bal) base = atomicAdd(nact, __popcll(bal));
            base = __shfl(base, 0);
            if (on) act[base + __popcll(bal & ((1ull << lane) - 1))] = ((unsigned)st << 16) | (unsigned)strip; } } }

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

// ---- record walk: the tile's candidate records come from HBM (k_screen wrote them); no sample is in LDS.
// The same sequential detector as walk(), one iteration per candidate (row, kind): rows inside a blind countdown
// are skipped, the others are decided from the integer edge distances against the guard-banded thresholds and,
// inside the guard band, by re-evaluating the reference's float comparisons on the reconstructed codes.
// Returns false (the caller discards the state) only when the records cannot describe what the detector would see:
// the literal start-up path, a minimum k_screen could not derive, a full event list.
// kDirect (k_lwalk: one lane per walker, lists read straight from HBM): a detection's event is refined and stored at once,
// into [.., ev_limit) of the walker's region, instead of being queued in LDS for finalize_records16.
template <bool kDirect = false>
__device__ __forceinline__ bool walk_records(Walker &w, Ctx &cx, int pidx, int trk, long long limit,
                                             const CandUnit *recs, int nrecs, const TileDir &td, int &why, unsigned int ev_limit = 0xffffffffu) {
   const DevCfg *cfg = cx.cfg;
   const DevParm &P = cfg->parm[pidx];
   const Tile &tl = cx.tile;
   const int W = P.W;
   why = 0;
   if (!w.fast || tl.row0 - kScreenHalo < w.trust_from) { why = 1; return false; }
   const long long tile_end = tl.row0 + tl.nrows;
   const bool whole = limit >= tile_end;
   if (limit > tile_end) limit = tile_end;
   if (whole && td.end_ld == 0) { why = 7; return false; }        // the next tile needs the minimum's state
   const int lim = (int)(limit - tl.row0);
   long long n64 = max(w.next, w.blind_until + 1);
   int cur = (n64 - tl.row0 > lim) ? lim : (int)(n64 - tl.row0);
   if (cur < 0) cur = 0;
   const float mv = cfg->maxvolts, lsb = cfg->lsb_per_volt;
   const bool pe = cfg->mode == RTFE_PE;
   const int4 *units = reinterpret_cast<const int4 *>(recs);
   int i = 0;
   // Two-phase rounds keep the walker lanes of a wave together: (A) every lane scans its runs up to its next detection
   // (a few integer compares per row), (B) the lanes that found one do the detection bookkeeping side by side.
   #pragma nounroll
   for (;;) {
      int hit_n = -1, hit_ld = 0, hit_m = 0, hit_z = 0;
      bool hit_top = false, fail = false;
      #pragma nounroll
      while (i < nrecs) {                                           // (A)
         const int4 A = units[i];                                   // run header
         const int n_s = A.x & 0x7ff, nr = (A.x >> 11) & 0x3f;
         const int ibase = nrecs + (int)((unsigned)A.x >> 18);      // first margin unit of this run (behind the headers)
         ++i;
         if (n_s >= lim) { i = nrecs; break; }
         int k = cur > n_s ? cur - n_s : 0;                          // rows inside the countdown of the last detection are skipped
         if (k >= nr) continue;
         const int ld0 = (A.y >> 16) & 0xff;
         const bool is_top = ((A.x >> 17) & 1) == 0;
         const int m = (int)(short)(A.y & 0xffff);
         if (ld0 == 0) { why = 4; fail = true; break; }
         if (w.reqmin != 0) {                                       // min_peak test: the same for every row of the run
            const int a = is_top ? m : -m;
            if (a <= w.min_lo) continue;
            if (a < w.min_hi) {
               if (w.thr_dirty) update_thresholds(w, P, lsb);
               if (!(is_top ? volt(m, mv) > w.reqmin : volt(m, mv) < -w.reqmin)) continue; } }
         int hit = -1;
         int4 M = make_int4(0, 0, 0, 0);
         int mu = -1;
         #pragma nounroll
         for (; k < nr && n_s + k < lim; ++k) {
            int pr;
            if (k == 0) pr = A.w;
            else {
               const int u = (k - 1) >> 2, c = (k - 1) & 3;
               if (u != mu) { M = units[ibase + u]; mu = u; }
               pr = c == 0 ? M.x : (c == 1 ? M.y : (c == 2 ? M.z : M.w)); }
            const int dl = pr & 0xffff, dr = (int)((unsigned)pr >> 16);
            const int mg = min(dl, dr);
            if (mg <= w.rise_lo) continue;                          // fails for sure
            if (mg < w.rise_hi) {                                   // guard band: the reference's own comparison
               if (w.thr_dirty) update_thresholds(w, P, lsb);
               const float vm = volt(m, mv);
               const float vl = volt(is_top ? m - dl : m + dl, mv), vr = volt(is_top ? m - dr : m + dr, mv);
               if (!(is_top ? (vm > vl + w.rise && vm > vr + w.rise) : (vm < vl - w.rise && vm < vr - w.rise))) continue; }
            hit = k; break; }
         if (hit < 0) continue;
         hit_n = n_s + hit; hit_ld = ld0 - hit; hit_m = m; hit_z = A.z; hit_top = is_top;
         break; }
      if (fail) return false;
      if (hit_n < 0) break;
      {                                                             // (B) a detection at row hit_n: the bookkeeping of emit_peak
         const int n = hit_n, ld = hit_ld, m = hit_m;
         const bool is_top = hit_top;
         const float v = volt(m, mv);
         double t_peak = 0;
         if (pe && !w.datablock && w.peakcount >= 68) {            // the end of the PE preamble is decided on peak times
            const int adjcode = refine_code(cfg, m, (int)(short)(hit_z & 0xffff), hit_z >> 16, w.agc_gain, is_top);
            const float adj = adjcode == 1 ? -0.5f : (adjcode == 2 ? 0.5f : 0.0f);
            t_peak = time_of(cfg, cx.row_base + tl.row0 + n) - ((float)(W - ld) - adj) * cfg->sample_deltat; }
         if (w.nevents >= cx.cap) w.flags |= RTFE_F_EVENT_OVERFLOW;
         else if (kDirect) {
            if (w.nevents >= ev_limit) { why = 6; return false; }
            const int adjcode = refine_code(cfg, m, (int)(short)(hit_z & 0xffff), hit_z >> 16, w.agc_gain, is_top);
            store_event(cx, pidx, trk, w.nevents, tl.row0 + n, v, w.agc_gain, is_top, adjcode, ld); }
         else {
            if (cx.nrec >= cx.rec_cap16) { why = 6; return false; }
            // Rec16 as two 8-byte LDS stores: {n_rel|ld|kind, g} {val|prev, next}
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(reinterpret_cast<Rec16 *>(cx.recs) + cx.nrec++);
            const unsigned int w1 = (unsigned)n | ((unsigned)ld << 16) | ((unsigned)(is_top ? 0 : 1) << 24);
            const unsigned int w3 = (unsigned)(m & 0xffff) | ((unsigned)hit_z << 16);        // val | prev
            const unsigned int w4 = ((unsigned)hit_z >> 16);                                   // next
            dst[0] = (unsigned long long)w1 | ((unsigned long long)__float_as_uint(w.agc_gain) << 32);
            dst[1] = (unsigned long long)w3 | ((unsigned long long)w4 << 32); }
         if (is_top) w.v_top = v; else w.v_bot = v;
         ++w.nevents;
         agc_after_peak(w, cfg, P, cx.heights, is_top, t_peak);
         if (!(w.agc_gain > 0)) { why = 7; return false; }         // (src/decoder.c:782 is about to fire: the sample path writes the marker)
         if (!approx_thresholds(w, P, lsb)) update_thresholds(w, P, lsb);
         w.blind_until = tl.row0 + n + ld;                         // pkww_countdown = left_distance (src/decoder.c:741)
         cur = n + ld + 1; } }
   n64 = tl.row0 + lim;
   w.next = n64 < limit ? n64 : limit;
   if (whole) {                                                     // the minimum's state after the tile's last row, from k_screen
      const int last = tl.nrows - 1;
      w.minv = td.end_min; w.cpos = tl.row0 + last; w.qtrig = tl.row0 + last + td.end_ld; w.chain_pending = false; }
   return true; }

// LDS carve of k_screen / k_decode.  ONE definition, used by the kernels and by the host when it sizes the dynamic
// LDS allocation (rtfe_api.hip): an under-sized allocation does not fault on the GPU, out-of-range LDS reads return 0.
// row strides of the screen's bitmaps / left_distance maps: consecutive tracks' rows start an odd number of 8-byte words
// apart (mod the 32 LDS banks), so that the per-track 8-byte stores of one strip's lanes do not pile up on two banks
__host__ __device__ inline unsigned lds_bstride(int tile_rows) { unsigned v = (unsigned)(tile_rows + kScreenHalo) / 8 + 8; while ((v / 4) % 32 != 22) v += 8; return v; }
__host__ __device__ inline unsigned lds_ldstride(int tile_rows) { unsigned v = (unsigned)(tile_rows + kScreenHalo); while ((v / 4) % 32 != 18) v += 8; return v; }
struct LdsLayout {
   unsigned bits, ldpos, heights, recs, nrec, runs, runcnt, runtab, act, walkers, walkers_next, heights_bak, fdiff, total; };
__host__ __device__ inline unsigned lds_runtab_cap(const DevCfg &c) {        // run descriptors of one (tile, screen) (k_screen)
   const unsigned n = (unsigned)(c.ntrks * c.tile_rows) / 8u;
   return n > 2048u ? 2048u : n; }
__host__ __device__ inline unsigned lds_align16(unsigned v) { return (v + 15u) & ~15u; }
__host__ __device__ inline LdsLayout lds_layout(const DevCfg &c, bool decode) {
   LdsLayout L;
   const unsigned ntrks = (unsigned)c.ntrks, nst = (unsigned)c.nscreens * ntrks, nwalk = (unsigned)c.nparm * ntrks;
   const unsigned T = (unsigned)c.tile_rows;
   unsigned off = lds_align16(ntrks * (unsigned)c.ldw * 2u + 16u);          // (ldw rows of ntrks samples, + one vector of slack)
   const unsigned nsl = decode ? nst : ntrks;                                  // k_screen works through the screens one at a time
   const bool zeros = decode && c.find_zeros && !c.differentiate;             // no screen: the space holds the sub-segment records of zeros_tile_parallel
   L.bits = off;      off = lds_align16(off + (zeros ? ntrks * (T / (unsigned)kZcSub) * (unsigned)sizeof(ZcLane) : nsl * 5u * lds_bstride((int)T)));
   L.ldpos = off;     off = lds_align16(off + (zeros ? 0u : nsl * 2u * lds_ldstride((int)T)));
   // k_decode: the candidate records of a tile share the space of the sample tile (a tile is decided either from
   // its records or from its samples, never both)
   L.runs = 0;
   if (decode) { const unsigned r = lds_align16((unsigned)c.lds_units * (unsigned)sizeof(CandUnit)); if (r > off) off = r; }
   L.runcnt = off;    if (!decode) off = lds_align16(off + ntrks * (T / 8) * 4u);
   L.runtab = off;    if (!decode) off = lds_align16(off + lds_runtab_cap(c) * 8u);
   L.act = off;       if (!decode) off = lds_align16(off + ntrks * (T / 8) * 4u);
   L.fdiff = off;     if (decode && c.differentiate && !c.find_zeros) off = lds_align16(off + ntrks * (unsigned)c.ldw * 4u + 32u);
   L.heights = off;   if (decode) off = lds_align16(off + nwalk * 10u * 4u);
   L.recs = off;      if (decode) off = lds_align16(off + nwalk * (unsigned)c.rec_cap * (unsigned)sizeof(Rec));
   L.nrec = off;      if (decode) off = lds_align16(off + nwalk * 4u);
   L.walkers = off;   if (decode) off = lds_align16(off + nwalk * (unsigned)sizeof(Walker));
   // (the optimistic copies exist for the record tiles of k_decode only: without a record path a sweep's workgroup is
   //  20 KB lighter and two fit a CU)
   L.walkers_next = off; if (decode && c.record_path) off = lds_align16(off + nwalk * (unsigned)sizeof(Walker));
   L.heights_bak = off;  if (decode && c.record_path) off = lds_align16(off + nwalk * 10u * 4u);
   L.total = off;
   return L; }

struct WalkLds { unsigned units, heights, heights_bak, recs, nrec, idx0, band, pmoff, hoff, wst, pm, pmmap, hmap, wtab, total; };
__host__ __device__ inline WalkLds lds_layout_walk(const DevCfg &c) {
   WalkLds L;
   const unsigned nwalk = (unsigned)(c.nparm * c.ntrks);
   unsigned off = 0;
   L.units = off;        off = lds_align16(off + (unsigned)c.lds_units * (unsigned)sizeof(CandUnit));
   L.heights = off;      off = lds_align16(off + nwalk * 10u * 4u);
   L.heights_bak = off;  off = lds_align16(off + nwalk * 10u * 4u);
   L.recs = off;         off = lds_align16(off + nwalk * (unsigned)c.rec_cap16 * (unsigned)sizeof(Rec16));
   L.nrec = off;         off = lds_align16(off + (nwalk + 1) * 4u);
   L.idx0 = off;         off = lds_align16(off + nwalk * 4u);
   L.band = off;         off = lds_align16(off + nwalk * 16u);
   L.pmoff = off;        off = lds_align16(off + (nwalk + 1) * 4u);
   L.hoff = off;         off = lds_align16(off + (nwalk + 1) * 4u);
   L.wst = off;          off = lds_align16(off + nwalk * 16u);
   L.pm = off;           off = lds_align16(off + (unsigned)c.pm_cap * 2u);
   L.pmmap = off;        off = lds_align16(off + (unsigned)c.pm_cap);
   L.hmap = off;         off = lds_align16(off + nwalk * (unsigned)c.rec_cap16);
   L.wtab = off;         off = lds_align16(off + nwalk * 4u);
   L.total = off;
   return L; }

// ------------------------------------------------------------------------------------------------
// k_screen: the dense, stateless pass.  One workgroup per tile of the tape-global grid: coalesced loads of the
// AoS rows -> SoA LDS tile, sliding-window screen (tile + kScreenHalo rows in front), the stale-minimum chain,
// candidate records -> HBM (TileDir + a fixed slot of run_cap records per (tile, screen, track)).
// Everything the sequential pass needs in the common case; it never has to touch the samples again.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 4) k_screen(const DevCfg *__restrict__ cfgp, const int16_t *__restrict__ rows, long long nrows,
                                                TileDir *__restrict__ dir, CandUnit *__restrict__ pool, long long ntiles,
                                                unsigned long long *__restrict__ scr, const unsigned int *__restrict__ dead) {
#ifdef RTFE_CPU_EMUL
   unsigned char *smem = g_dyn_smem;
#else
   extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
   __shared__ DevCfg cfg;
   for (int i = threadIdx.x; i < (int)(sizeof(DevCfg) / 4); i += blockDim.x) reinterpret_cast<int *>(&cfg)[i] = reinterpret_cast<const int *>(cfgp)[i];
   __syncthreads();
   const int ntrks = cfg.ntrks, nst = cfg.nscreens * ntrks;
   const LdsLayout L = lds_layout(cfg, false);
   Tile tl;
   tl.x = reinterpret_cast<int16_t *>(smem);
   tl.ldw = cfg.ldw; tl.halo = cfg.halo_rows; tl.colof = cfg.trk_to_head;
   tl.ntrks = ntrks; tl.skew = cfg.skew; tl.reset = -(1ll << 40);
   tl.bits = smem + L.bits; tl.bstride = (int)lds_bstride(cfg.tile_rows);
   tl.ldpos = smem + L.ldpos; tl.ldstride = (int)lds_ldstride(cfg.tile_rows);
   unsigned int *stripcnt = reinterpret_cast<unsigned int *>(smem + L.runcnt);         // [nst][tile_rows / 8] runs | margin units << 16 per strip
   u64 *runtab = reinterpret_cast<u64 *>(smem + L.runtab);           // the tile's run descriptors
   const int tabcap = (int)lds_runtab_cap(cfg);
   __shared__ int s_nruns, s_nact;
   unsigned int *act = reinterpret_cast<unsigned int *>(smem + L.act);                  // strips with candidates
   __shared__ int s_total[kMaxScreens * RTFE_MAXTRKS], s_lbase[kMaxScreens * RTFE_MAXTRKS];
   const long long T = cfg.tile_rows;
   for (long long g = blockIdx.x; g < ntiles; g += gridDim.x) {
      tl.row0 = g * T; tl.nrows = (int)((tl.row0 + T <= nrows) ? T : nrows - tl.row0);
      if ((dead[g >> 5] >> (g & 31)) & 1) {                           // deep inside an inter-block gap: no walker comes here (k_bursts)
         if ((int)threadIdx.x < nst) { TileDir d0; d0.count = 0xFFFF; d0.nruns = 0; d0.end_ld = 0; d0.pad = 0; d0.end_min = 0; dir[g * nst + threadIdx.x] = d0; }   // (if one ever did: "no list" = sample path)
         continue; }
      __syncthreads();
      long long k0 = 0, k1 = 0, k2 = 0, k3 = 0;
      if (cfg.debug) k0 = clock64();
      load_tile(&cfg, tl, rows, nrows);
      __syncthreads();
      if (cfg.debug) k1 = clock64();
      if (cfg.cut == 1) continue;
      // the screens (distinct window widths of a parameter-set sweep) one after the other through the same LDS arrays:
      // a sweep then costs occupancy no more than a single set does.  Lists are packed screen-major, track-minor.
      int units_before = 0;
      long long k2a = 0, k2b = 0;
      for (int sc_real = 0; sc_real < cfg.nscreens; ++sc_real) {
      const int W_sc = cfg.screen[sc_real].W;
      __syncthreads();
      if (threadIdx.x == 0) { s_nact = 0; s_nruns = 0; }
      __syncthreads();
      run_screens(&cfg, tl, true, act, &s_nact, stripcnt, cfg.tile_rows / kStrip, sc_real);
      __syncthreads();
      if (cfg.debug) k2 = clock64();
      if (cfg.cut == 2) continue;
      const int nstrips = (tl.nrows + kStrip - 1) / kStrip;
      const int smax = cfg.tile_rows / kStrip;
      const int nactive = s_nact;
      for (int k = threadIdx.x; k < nactive; k += blockDim.x) {      // the reference's minimum at the bottom candidates
         const unsigned a = act[k];
         fill_stale(tl, 0, (int)(a >> 16), (int)(a & 0xffff)); }
      __syncthreads();
      if (cfg.cut == 3) continue;
      if (cfg.debug) k2a = clock64();
      for (int k = threadIdx.x; k < nactive; k += blockDim.x) {
         const unsigned a = act[k];
         run_starts(tl, 0, (int)(a >> 16), (int)(a & 0xffff)); }
      __syncthreads();
      if (cfg.debug) k2b = clock64();
      if (cfg.cut == 4) continue;
      for (int k = threadIdx.x; k < nactive; k += blockDim.x) {
         const unsigned a = act[k];
         const int lt = (int)(a >> 16), strip = (int)(a & 0xffff);
         stripcnt[lt * smax + strip] = (unsigned int)list_runs(tl, lt, 0, lt, strip, runtab, &s_nruns, tabcap); }
      __syncthreads();
      // exclusive scan of the strips' unit counts, list by list (one wave per list at a time)
      {
         const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
         for (int lt = wave; lt < ntrks; lt += nwaves) {
            int carry = 0;
            for (int s0 = 0; s0 < nstrips; s0 += 64) {
               const int v = s0 + lane < nstrips ? (int)stripcnt[lt * smax + s0 + lane] : 0;
               int x = v;
               #pragma unroll
               for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
               if (s0 + lane < nstrips) stripcnt[lt * smax + s0 + lane] = (unsigned int)(carry + x - v);
               carry += __shfl(x, 63); }
            if (lane == 0) s_total[lt] = carry; } }
      __syncthreads();
      if (cfg.debug) k3 = clock64();
      if (cfg.cut == 5) continue;
      int tile_units = units_before;
      for (int s2 = 0; s2 < ntrks; ++s2) { if (threadIdx.x == 0) s_lbase[s2] = tile_units; tile_units += (s_total[s2] & 0xffff) + (s_total[s2] >> 16); }
      __syncthreads();
      const bool tab_ok = s_nruns <= tabcap && tile_units <= nst * cfg.run_cap;
      if (tab_ok)
         for (int r = threadIdx.x >> 2; r < s_nruns; r += blockDim.x >> 2) {
            const u64 d = runtab[r];
            const int lt = (int)(d & 0xff), kind = (int)((d >> 8) & 1), n = (int)((d >> 16) & 0xffff), nr = (int)((d >> 32) & 0xff);
            const int relrun = (int)((d >> 40) & 0x3ff), relmarg = (int)(d >> 50);
            const unsigned sb = stripcnt[lt * smax + (n >> 3)];
            int4 *list = reinterpret_cast<int4 *>(pool) + (size_t)g * nst * cfg.run_cap + s_lbase[lt];      // the tile's lists are packed one behind the other
            const int moff = (int)(sb >> 16) + relmarg;
            build_run(tl, &cfg, 0, lt, W_sc, n, kind, nr, list + (sb & 0xffff) + relrun, moff,
                      list + (s_total[lt] & 0xffff) + moff, (int)(threadIdx.x & 3)); }
      if ((int)threadIdx.x < ntrks) {
         const int trk = threadIdx.x;
         TileDir d;
         const int units_l = (s_total[trk] & 0xffff) + (s_total[trk] >> 16);
         d.count = (!tab_ok || units_l >= 0xFFFF || (s_total[trk] >> 16) >= (1 << 14)) ? (uint16_t)0xFFFF : (uint16_t)units_l;
         d.nruns = (uint16_t)(s_total[trk] & 0xffff);
         const int last = tl.nrows - 1;
         const int eld = stale_ld(tl.map(0, 2, trk), tl.ldmap(0, 1, trk), last);
         d.end_ld = (uint8_t)eld; d.pad = 0;
         d.end_min = eld ? (int16_t)tile_col(tl, trk, cfg.skew[trk])[last - W_sc + eld] : (int16_t)0;
         dir[g * nst + sc_real * ntrks + trk] = d; }
      units_before = tab_ok ? tile_units : nst * cfg.run_cap + 1; }      // (an overflowing screen poisons the rest of the tile's slot)
      if (cfg.debug) {
         __syncthreads();
         if (threadIdx.x == 0) {
            const long long k4 = clock64();
            atomicAdd(&scr[0], (unsigned long long)(k1 - k0)); atomicAdd(&scr[1], (unsigned long long)(k2 - k1));
            atomicAdd(&scr[2], (unsigned long long)(k3 - k2)); atomicAdd(&scr[3], (unsigned long long)(k4 - k3)); atomicAdd(&scr[4], 1ull);
            atomicAdd(&scr[5], (unsigned long long)(k2a - k2)); atomicAdd(&scr[6], (unsigned long long)(k2b - k2a)); } } } }

// ------------------------------------------------------------------------------------------------
// k_walk: the sequential pass in the common case.  One small workgroup per burst (a lane per (parameter set, track)),
// tile after tile of the tape-global grid: the tile's candidate runs HBM -> LDS, walk_records(), events.  No samples,
// little LDS, few registers: many bursts resident per CU.  A tile the records cannot decide hands the burst (state
// as of that tile's start) to the second k_decode pass.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_walk_fields(Walker &w, const Walker &s) {
   w.fast = s.fast; w.trust_from = s.trust_from; w.next = s.next; w.blind_until = s.blind_until;
   w.agc_gain = s.agc_gain; w.v_avg_height = s.v_avg_height; w.v_avg_height_sum = s.v_avg_height_sum;
   w.v_avg_height_count = s.v_avg_height_count; w.peakcount = s.peakcount; w.heightndx = s.heightndx;
   w.v_top = s.v_top; w.v_bot = s.v_bot; w.v_lasttop = s.v_lasttop; w.v_lastbot = s.v_lastbot;
   w.datablock = s.datablock; w.bit1_up = s.bit1_up; w.t_lastpeak = s.t_lastpeak;
   w.rise = s.rise; w.reqmin = s.reqmin; w.thr_dirty = s.thr_dirty;
   w.rise_lo = s.rise_lo; w.rise_hi = s.rise_hi; w.min_lo = s.min_lo; w.min_hi = s.min_hi;
   w.minv = s.minv; w.qtrig = s.qtrig; w.cpos = s.cpos; w.chain_pending = s.chain_pending;
   w.nevents = s.nevents; w.flags = s.flags; }

__global__ void __launch_bounds__(kDecodeThreads) k_walk(const DevCfg *__restrict__ cfgp, long long nrows, long long row_base,
                                                       rtfe_burst *__restrict__ bursts, BurstScratch *__restrict__ scratch,
                                                       uint32_t *__restrict__ counts, rtfe_event *__restrict__ events,
                                                       const TileDir *__restrict__ dir, const CandUnit *__restrict__ pool,
                                                       BurstCtl *__restrict__ ctl, WalkState *__restrict__ wstate,
                                                       int walk_mode, const SegTab *__restrict__ segtab, const int *__restrict__ segburst,
                                                       WalkState *__restrict__ seg_start, WalkState *__restrict__ seg_end, int *__restrict__ seg_status) {
#ifdef RTFE_CPU_EMUL
   unsigned char *smem = g_dyn_smem;            // tests/cpu_emul only
#else
   extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
   __shared__ DevCfg cfg;
   __shared__ int s_burst;
   __shared__ unsigned int s_flags;
   __shared__ int s_needfull;
   __shared__ TileDir s_dir[kMaxScreens * RTFE_MAXTRKS];
   __shared__ int s_off[kMaxScreens * RTFE_MAXTRKS + 1];
   for (int i = threadIdx.x; i < (int)(sizeof(DevCfg) / 4); i += blockDim.x) reinterpret_cast<int *>(&cfg)[i] = reinterpret_cast<const int *>(cfgp)[i];
   __syncthreads();
   const int ntrks = cfg.ntrks, nst = cfg.nscreens * ntrks, nwalk = cfg.nparm * ntrks;
   const WalkLds L = lds_layout_walk(cfg);
   int4 *units = reinterpret_cast<int4 *>(smem + L.units);
   float *heights_all = reinterpret_cast<float *>(smem + L.heights), *heights_bak = reinterpret_cast<float *>(smem + L.heights_bak);
   unsigned char *recs_all = smem + L.recs;                          // [nwalk][rec_cap16] Rec16
   int *nrec_all = reinterpret_cast<int *>(smem + L.nrec);          // [nwalk + 1] -> exclusive scan
   int *idx0_all = reinterpret_cast<int *>(smem + L.idx0);          // [nwalk]
   int *band = reinterpret_cast<int *>(smem + L.band);              // [nwalk][4] parallel path: rise lo/hi, min_peak lo/hi (wide)
   int *pmoff = reinterpret_cast<int *>(smem + L.pmoff);            // [nwalk + 1]
   int *hoff = reinterpret_cast<int *>(smem + L.hoff);              // [nwalk + 1]
   float *wst = reinterpret_cast<float *>(smem + L.wst);            // [nwalk][4] v_lasttop, v_lastbot, v_avg_height, alpha
   unsigned short *pm = reinterpret_cast<unsigned short *>(smem + L.pm);     // [pm_cap] last sure row + 1 | (last possible row + 1) << 8
   unsigned char *pmmap = smem + L.pmmap;                           // [pm_cap] walker of a verdict slot
   unsigned int *s_wtab = reinterpret_cast<unsigned int *>(smem + L.wtab);      // [nwalk] walker -> list (screen, track) | parameter set << 8 | track << 16
   unsigned char *hmap = smem + L.hmap;                             // [nwalk * rec_cap16] walker of a detection slot
   __shared__ unsigned int s_seq;
   // the parallel tile path covers the NRZI / GCR AGC schedule with either AGC flavour (alpha filter, or the minimum of the
   // last agc_window heights); PE's preamble logic walks sequentially
   bool par_mode = cfg.mode != RTFE_PE;
   for (int p = 0; p < cfg.nparm; ++p) if ((cfg.parm[p].agc_alpha != 0) == (cfg.parm[p].agc_window != 0)) par_mode = false;
   const int rstride = cfg.rec_cap16 * (int)sizeof(Rec16);
   const int nwaves = blockDim.x >> 6;
   const int my_w = (threadIdx.x & 63) * nwaves + (threadIdx.x >> 6);
   const bool is_walker = my_w < nwalk;
   const int pidx = is_walker ? my_w / ntrks : 0, trk = is_walker ? my_w - pidx * ntrks : 0;
   if (is_walker) s_wtab[my_w] = (unsigned)(cfg.parm[pidx].screen * ntrks + trk) | ((unsigned)pidx << 8) | ((unsigned)trk << 16);
   __syncthreads();
   Ctx cx;
   cx.cfg = &cfg;
   cx.row_base = row_base;
   cx.tile.x = nullptr; cx.tile.ldw = 0; cx.tile.halo = 0; cx.tile.colof = cfg.trk_to_head; cx.tile.ntrks = ntrks; cx.tile.skew = cfg.skew;
   cx.tile.bits = nullptr; cx.tile.bstride = 0; cx.tile.ldpos = nullptr; cx.tile.ldstride = 0;
   cx.heights = heights_all + (size_t)(is_walker ? my_w : 0) * 10;
   cx.rec_cap = 0; cx.rec_cap16 = cfg.rec_cap16;
   cx.recs = reinterpret_cast<Rec *>(recs_all + (size_t)(is_walker ? my_w : 0) * rstride);
   const long long T = cfg.tile_rows;
   // walk_mode: kWalkWhole = every ready burst from its hand-over tile to its end; kWalkPre = the same, but only until
   // every walker has left the AGC start-up (then the burst goes back to "ready" for the segment pass); kWalkSegs = the
   // work items are segments (k_segs): tiles [t0 + s*seg_tiles, ...) of a burst, walked from a guessed state after
   // kSegWarmup tiles of warm-up (segment 0: from the burst's true state)
   for (;;) {
      if (threadIdx.x == 0) { s_burst = atomicAdd(walk_mode == kWalkSegs ? &scratch->queue_seg : &scratch->queue_walk, 1); s_flags = 0; }
      __syncthreads();
      const int item = s_burst;
      if (item >= (walk_mode == kWalkSegs ? scratch->nsegs : scratch->nbursts)) break;
      const int b = walk_mode == kWalkSegs ? segburst[item] : item;
      if (ctl[b].status != kBurstReady) { __syncthreads(); continue; }
      const rtfe_burst B = bursts[b];
      cx.events = events + B.event_base;
      cx.cap = B.event_cap;
      const long long reset = ctl[b].reset, stop = ctl[b].stop;
      const unsigned int bflags = ctl[b].bflags;
      cx.tile.reset = reset;
      Walker w;                                                        // (only the fields of load_walk_fields are ever touched)
      load_walk_fields(w, wstate[(size_t)b * nwalk + (is_walker ? my_w : 0)].w);
      if (is_walker) {
         const WalkState &ws = wstate[(size_t)b * nwalk + my_w];
         for (int i = 0; i < 10; ++i) cx.heights[i] = ws.heights[i]; }
      long long g = ctl[b].next_tile;
      bool give_back = false;
      long long acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0, ntl = 0;
      long long pa[6] = {0, 0, 0, 0, 0, 0};
      TileDir nd; nd.count = 0; nd.nruns = 0; nd.end_ld = 0; nd.pad = 0; nd.end_min = 0;
      // ---- segment mode: which tiles, from which state, into which slot of the burst's event regions ----
      long long g_hi = (stop + T - 1) / T, seg_lo = g;              // tiles [g, g_hi); seg_lo = first tile whose events count
      int seg = 0, nseg = 1;
      unsigned int ev_base = 0, ev_limit = cx.cap;                  // this segment's slot [ev_base, ev_limit) of the walker's event region
      bool pre_done = false;
      if (walk_mode == kWalkSegs) {
         const SegTab stb = segtab[b];
         seg = item - stb.first; nseg = stb.nseg;
         seg_lo = stb.t0 + (long long)seg * cfg.seg_tiles;
         g_hi = seg + 1 < nseg ? seg_lo + cfg.seg_tiles : stb.tend;
         g = seg_lo;
         if (nseg > 1) {
            ev_base = w.nevents + (unsigned)seg * (unsigned)cfg.seg_evcap; ev_limit = ev_base + (unsigned)cfg.seg_evcap;
            if (ev_limit > cx.cap) ev_limit = cx.cap;
            if (seg > 0) {
               // a guess of the state kSegWarmup tiles ahead of the segment: the burst's steady state as of its hand-over, no
               // countdown pending.  Decisions re-join the true sequence at the first stretch of W+1 rows without a candidate,
               // the peak memory after two detections, the AGC filter geometrically - k_stitch verifies that all did.
               g = seg_lo - cfg.seg_warm;
               w.blind_until = -1; w.next = g * T; w.trust_from = -(1ll << 40); w.flags = 0;
               w.cpos = g * T - 1; w.chain_pending = false;
               w.nevents = ev_base; } } }
      const long long g_first = g;
      for (; g < g_hi && g * T < stop; ++g) {
         const long long tile0 = g * T;
         const long long tn = (tile0 + T <= nrows) ? T : nrows - tile0;
         if (tn <= 0) break;
         cx.tile.row0 = tile0; cx.tile.nrows = (int)tn;
         __syncthreads();
         if (walk_mode == kWalkSegs && seg > 0 && g == seg_lo && is_walker) {
            // end of the warm-up: this is the state the segment really starts from (its detections so far are dropped)
            w.nevents = ev_base;
            const DevParm &P = cfg.parm[pidx];
            update_thresholds(w, P, cfg.lsb_per_volt);               // (exact thresholds are a function of the AGC state: normal form)
            WalkState &ws = seg_start[(size_t)item * nwalk + my_w];
            load_walk_fields(ws.w, w);
            for (int i = 0; i < 10; ++i) ws.heights[i] = cx.heights[i]; }
         if (walk_mode == kWalkPre && g > g_first) {                 // has every walker left the AGC start-up (or never entered it)?
            if (threadIdx.x == 0) s_needfull = 0;
            __syncthreads();
            if (is_walker && !((w.peakcount >= 16 && w.v_avg_height_count == 0) || w.peakcount == 0)) atomicOr((unsigned int *)&s_needfull, 1u);
            __syncthreads();
            if (!s_needfull || g - g_first >= 6) { pre_done = true; break; }
            __syncthreads(); }
         if (walk_mode == kWalkSegs && nseg > 1) {                    // room left in this segment's event slot?
            if (threadIdx.x == 0) s_needfull = 0;
            __syncthreads();
            if (is_walker && w.nevents + 2u * (unsigned)cfg.rec_cap16 + 8u >= ev_limit) atomicOr((unsigned int *)&s_needfull, 1u);
            __syncthreads();
            if (s_needfull) { give_back = true; break; }
            __syncthreads(); }
         long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
         if (cfg.debug) t0 = clock64();
         if (threadIdx.x == 0) s_needfull = 0;
         if (threadIdx.x < nst) {
            s_dir[threadIdx.x] = g == g_first ? dir[g * nst + threadIdx.x] : nd;
            if ((g + 1) * T < nrows) nd = dir[(g + 1) * nst + threadIdx.x]; }      // the next tile's entry, in flight during this tile
         __syncthreads();
         if (threadIdx.x <= nst) {                                  // where each list starts inside the tile's slot of the pool
            int o = 0; bool bad = false;
            for (int st = 0; st < (int)threadIdx.x; ++st) { if (s_dir[st].count == 0xFFFF) bad = true; o += s_dir[st].count; }
            s_off[threadIdx.x] = bad ? (1 << 30) : o; }
         __syncthreads();
         if (s_off[nst] >= (1 << 30)) { give_back = true; break; }
         if (cfg.debug) t1 = clock64();
         // the lists go through LDS in groups of consecutive lists that fit (usually one group = the whole tile); the
         // walkers are independent of each other, so each group is walked to the end of the tile before the next
         for (int st_lo = 0, st_hi; st_lo < nst && !give_back; st_lo = st_hi) {
         st_hi = st_lo + 1;
         while (st_hi < nst && s_off[st_hi + 1] - s_off[st_lo] <= cfg.lds_units) ++st_hi;
         const int gbase = s_off[st_lo];
         const int my_st0 = cfg.parm[pidx].screen * ntrks + trk;
         const bool active = is_walker && my_st0 >= st_lo && my_st0 < st_hi;
         __syncthreads();
         {
            // the tile's lists are contiguous in the pool: independent 16-byte loads, eight in flight per lane
            const int4 *src = reinterpret_cast<const int4 *>(pool) + (size_t)g * nst * cfg.run_cap + gbase;
            const int total = s_off[st_hi] - gbase;
            for (int i0 = threadIdx.x; i0 < total; i0 += 8 * (int)blockDim.x) {
               int4 q0, q1, q2, q3, q4, q5, q6, q7;
               const int bd = (int)blockDim.x;
               q0 = src[i0];
               q1 = i0 + bd < total ? src[i0 + bd] : q0;
               q2 = i0 + 2 * bd < total ? src[i0 + 2 * bd] : q0;
               q3 = i0 + 3 * bd < total ? src[i0 + 3 * bd] : q0;
               q4 = i0 + 4 * bd < total ? src[i0 + 4 * bd] : q0;
               q5 = i0 + 5 * bd < total ? src[i0 + 5 * bd] : q0;
               q6 = i0 + 6 * bd < total ? src[i0 + 6 * bd] : q0;
               q7 = i0 + 7 * bd < total ? src[i0 + 7 * bd] : q0;
               units[i0] = q0;
               if (i0 + bd < total) units[i0 + bd] = q1;
               if (i0 + 2 * bd < total) units[i0 + 2 * bd] = q2;
               if (i0 + 3 * bd < total) units[i0 + 3 * bd] = q3;
               if (i0 + 4 * bd < total) units[i0 + 4 * bd] = q4;
               if (i0 + 5 * bd < total) units[i0 + 5 * bd] = q5;
               if (i0 + 6 * bd < total) units[i0 + 6 * bd] = q6;
               if (i0 + 7 * bd < total) units[i0 + 7 * bd] = q7; } }
         __syncthreads();
         if (cfg.debug) t2 = clock64();
         if (is_walker) idx0_all[my_w] = (int)w.nevents;
         // ================= parallel tile path ("decisions first, gains after") =================
         // In the steady state of a block the detector's decisions depend on the AGC only through thresholds that move
         // slowly, and WHICH row of a run fires changes nothing downstream: the countdown ends when the extreme leaves the
         // window (row n_s + ld0 whatever the row), and the AGC sees the extreme's value.  So:
         // (1) every lane classifies runs against WIDE bands around the thresholds at the tile's start: the last row that
         //     passes for every threshold inside the band, and the last row that passes for some;
         // (2) one lane per walker follows the countdown chain: a run fires if a sure row is still ahead of the countdown;
         // (3) all lanes prepare volt() and the AGC quotients of the detections;
         // (4) one lane per walker runs the three-flop gain recurrence and checks that every threshold it produced stayed
         //     inside the band;
         // (5) all lanes find each detection's row with the reference's own comparisons at the now known gain, and write
         //     the events.
         // Any doubt (a run that fires for some thresholds of the band only, a walker not in steady state, a band left)
         // redoes the tile with the sequential walk below: the result is the reference's either way.
         bool done_par = false;
         if (par_mode) {
            const long long tile_end = tile0 + tn;
            const long long limit = stop < tile_end ? stop : tile_end;
            const int lim = (int)(limit - tile0);
            const bool whole = stop >= tile_end;
            const float lsb = cfg.lsb_per_volt;
            if (threadIdx.x == 0) s_seq = 0;
            __syncthreads();
            int my_st = 0, my_nruns = 0;
            if (is_walker) hoff[my_w] = 0;
            if (active) {
               const DevParm &P = cfg.parm[pidx];
               my_st = my_st0; my_nruns = s_dir[my_st].nruns;
               bool ok = w.fast && tile0 - kScreenHalo >= w.trust_from && whole && s_dir[my_st].end_ld != 0
                         && ((w.peakcount >= 16 && w.v_avg_height_count == 0) || my_nruns == 0)       // steady state, or nothing to decide (a silent track)
                         && w.agc_gain > 0 && w.nevents + (unsigned)cfg.rec_cap16 < cx.cap;
               if (ok) {
                  const float s4 = w.v_avg_height * 0.25f * fast_rcp(w.agc_gain);
                  const float rv = P.rise * s4, mv4 = P.min_peak * s4;
                  if (rv < P.screen_rise_v * 1.25f || (P.min_peak != 0 && mv4 < P.screen_minpk_v * 1.25f)) ok = false;   // near the screen's own thresholds
                  band[my_w * 4 + 0] = (int)(rv * lsb * 0.875f) - 4; band[my_w * 4 + 1] = (int)(rv * lsb * 1.125f) + 5;
                  band[my_w * 4 + 2] = P.min_peak != 0 ? (int)(mv4 * lsb * 0.875f) - 4 : -1;
                  band[my_w * 4 + 3] = (int)(mv4 * lsb * 1.125f) + 5;
                  wst[my_w * 4 + 0] = w.v_lasttop; wst[my_w * 4 + 1] = w.v_lastbot; wst[my_w * 4 + 2] = w.v_avg_height; wst[my_w * 4 + 3] = P.agc_alpha; }
               if (!ok) atomicOr(&s_seq, 1u);
               hoff[my_w] = my_nruns; }                                       // (hoff doubles as scratch for the scan)
            __syncthreads();
            if (!s_seq && is_walker) {                                        // exclusive scan, every walker lane for itself (idle walkers: 0 runs)
               int o = 0;
               for (int w2 = 0; w2 < my_w; ++w2) o += hoff[w2];
               pmoff[my_w] = o;
               if (my_w == nwalk - 1) { pmoff[nwalk] = o + my_nruns; if (o + my_nruns > cfg.pm_cap) atomicOr(&s_seq, 1u); }
               if (o + my_nruns <= cfg.pm_cap) for (int r = 0; r < my_nruns; ++r) pmmap[o + r] = (unsigned char)my_w; }
            __syncthreads();
            long long p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0;
            if (cfg.debug) p0 = clock64();
            if (!s_seq) {
               // ---- (1) all lanes: the pass mask of every (walker, run) ----
               const int total = pmoff[nwalk];
               for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
                  const int w2 = pmmap[idx];
                  const int r = idx - pmoff[w2];
                  const int st2 = (int)(s_wtab[w2] & 0xff);
                  const int4 *lst = units + (s_off[st2] - gbase);
                  const int4 A = lst[r];
                  const int nr = (A.x >> 11) & 0x3f, ld0 = (A.y >> 16) & 0xff;
                  const bool is_top = ((A.x >> 17) & 1) == 0;
                  const int m = (int)(short)(A.y & 0xffff);
                  const int lo = band[w2 * 4], hi = band[w2 * 4 + 1], mlo = band[w2 * 4 + 2], mhi = band[w2 * 4 + 3];
                  int ks = 0, km = 0;                                          // (last sure row + 1), (last possible row + 1)
                  bool doubt = ld0 == 0, pk = true;
                  if (mlo >= 0) { const int a = is_top ? m : -m; if (a <= mlo) pk = false; else if (a < mhi) doubt = true; }
                  if (pk && !doubt) {
                     const int4 *mg4 = lst + s_dir[st2].nruns + (int)((unsigned)A.x >> 18);
                     int4 M = make_int4(0, 0, 0, 0);
                     #pragma nounroll
                     for (int k = 0; k < nr; ++k) {
                        int pr;
                        if (k == 0) pr = A.w;
                        else {
                           const int c = (k - 1) & 3;
                           if (c == 0) M = mg4[(k - 1) >> 2];
                           pr = c == 0 ? M.x : (c == 1 ? M.y : (c == 2 ? M.z : M.w)); }
                        const int mg = min(pr & 0xffff, (int)((unsigned)pr >> 16));
                        if (mg >= hi) ks = k + 1;
                        if (mg > lo) km = k + 1; } }
                  if (doubt) km = 255;                                         // undecidable here whatever the countdown
                  pm[idx] = (unsigned short)(ks | (km << 8)); } }
            __syncthreads();
            if (cfg.debug) p1 = clock64();
            int nh = 0, last_blind = -1;
            if (!s_seq) {
               // ---- (2) one lane per walker: the countdown chain over the pass masks -> this tile's detections ----
               if (is_walker) nrec_all[my_w] = 0;
               if (active) {
                  long long n64 = max(w.next, w.blind_until + 1);
                  int cur = (n64 - tile0 > lim) ? lim : (int)(n64 - tile0);
                  if (cur < 0) cur = 0;
                  const int2 *lst = reinterpret_cast<const int2 *>(units + (s_off[my_st] - gbase));
                  unsigned int *hits = reinterpret_cast<unsigned int *>(recs_all + (size_t)my_w * rstride);      // 16-byte slots: rk, v, a, gain before
                  const unsigned short *mypm = pm + pmoff[my_w];
                  // (the next run's header and verdict are fetched while this one is looked at)
                  int2 A; A.x = 0; A.y = 0;
                  int v2 = 0;
                  if (my_nruns > 0) { A = lst[0]; v2 = mypm[0]; }
                  #pragma nounroll
                  for (int r = 0; r < my_nruns; ++r) {
                     int2 An = A; int v2n = 0;
                     if (r + 1 < my_nruns) { An = lst[2 * (r + 1)]; v2n = mypm[r + 1]; }
                     const int n_s = A.x & 0x7ff, nr = (A.x >> 11) & 0x3f;
                     const int k0 = cur > n_s ? cur - n_s : 0;
                     if (k0 < nr) {
                        if ((v2 & 0xff) <= k0) {                            // no sure row ahead of the countdown
                           if ((v2 >> 8) > k0) { atomicOr(&s_seq, 1u); break; } }   // ... but a possible one: undecidable here
                        else {
                           if (nh >= cfg.rec_cap16) { atomicOr(&s_seq, 1u); break; }
                           hits[nh * 4] = (unsigned)r | ((unsigned)k0 << 16) | ((unsigned)((A.x >> 17) & 1) << 31);
                           ++nh;
                           last_blind = n_s + ((A.y >> 16) & 0xff);         // row n + left_distance, whichever row of the run fires
                           cur = last_blind + 1; } }
                     A = An; v2 = v2n; }
                  nrec_all[my_w] = nh; } }
            __syncthreads();
            if (!s_seq && is_walker) {
               int o = 0;
               for (int w2 = 0; w2 < my_w; ++w2) o += nrec_all[w2];
               hoff[my_w] = o;
               if (my_w == nwalk - 1) hoff[nwalk] = o + nh;
               for (int j = 0; j < nh; ++j) hmap[o + j] = (unsigned char)my_w; }
            __syncthreads();
            if (cfg.debug) p2 = clock64();
            if (!s_seq) {
               // ---- (3a) all lanes: volt() of every detection ----
               const int total = hoff[nwalk];
               for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
                  const int w2 = hmap[idx];
                  const int st2 = (int)(s_wtab[w2] & 0xff);
                  unsigned int *slot = reinterpret_cast<unsigned int *>(recs_all + (size_t)w2 * rstride) + 4 * (idx - hoff[w2]);
                  const int r = (int)(slot[0] & 0xffff);
                  const int m = (int)(short)(units[s_off[st2] - gbase + r].y & 0xffff);
                  slot[1] = __float_as_uint(volt(m, cfg.maxvolts)); } }
            __syncthreads();
            if (!s_seq) {
               // ---- (3b) all lanes: what adjust_agc (src/decoder.c:500-531) will blend in at every detection ----
               const int total = hoff[nwalk];
               for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
                  const int w2 = hmap[idx];
                  const int j = idx - hoff[w2];
                  unsigned int *base = reinterpret_cast<unsigned int *>(recs_all + (size_t)w2 * rstride);
                  float vt = wst[w2 * 4 + 0], vb = wst[w2 * 4 + 1];                  // v_lasttop / v_lastbot as of this detection
                  bool ft = false, fb = false;
                  #pragma nounroll
                  for (int jj = j - 1; jj >= 0 && !(ft && fb); --jj) {
                     const bool bot = base[4 * jj] >> 31;
                     if (bot && !fb) { vb = __uint_as_float(base[4 * jj + 1]); fb = true; }
                     if (!bot && !ft) { vt = __uint_as_float(base[4 * jj + 1]); ft = true; } }
                  const float lastheight = vt - vb;
                  float a = -1.0f;
                  if (lastheight > 0) {
                     const float alpha = wst[w2 * 4 + 3];
                     if (alpha != 0) { const float gq = wst[w2 * 4 + 2] / lastheight; a = alpha * gq; }
                     else a = lastheight; }                                   // window AGC: the height itself goes into the ring
                  base[4 * j + 2] = __float_as_uint(a); } }
            __syncthreads();
            if (cfg.debug) p3 = clock64();
            float g_end = 0, vt_last = 0, vb_last = 0;
            bool any_t = false, any_b = false;
            int ndx_end = 0;
            if (!s_seq && active) {
               // ---- (4) one lane per walker: the gain recurrence; did every threshold stay inside the bands? ----
               const DevParm &P = cfg.parm[pidx];
               unsigned int *hits = reinterpret_cast<unsigned int *>(recs_all + (size_t)my_w * rstride);
               const float c1 = 1 - P.agc_alpha;
               const int N = P.agc_window;
               float *ring = heights_bak + my_w * 10;                         // window AGC: a working copy of the ring (committed with the tile)
               int ndx = w.heightndx;
               if (N) for (int i = 0; i < 10; ++i) ring[i] = cx.heights[i];
               float g = w.agc_gain, gmin = g, gmax = g;
               uint4 q; q.x = 0; q.y = 0; q.z = 0; q.w = 0;
               if (nh > 0) q = *reinterpret_cast<const uint4 *>(hits);                       // rk, v, a, -
               #pragma nounroll
               for (int j = 0; j < nh; ++j) {
                  uint4 qn = q;
                  if (j + 1 < nh) qn = *reinterpret_cast<const uint4 *>(hits + 4 * (j + 1));
                  hits[4 * j + 3] = __float_as_uint(g);
                  const float a = __uint_as_float(q.z);
                  if (a >= 0) {
                     if (N) {                                                 // src/decoder.c:519-529
                        ring[ndx] = a;
                        if (++ndx >= N) ndx = 0;
                        float minheight = 99;
                        for (int i = 0; i < N; ++i) if (ring[i] < minheight) minheight = ring[i];
                        g = w.v_avg_height / minheight; }
                     else g = a + c1 * g;                                     // src/decoder.c:511-512
                     if (g > 2.0f) g = 2.0f; }
                  gmin = fminf(gmin, g); gmax = fmaxf(gmax, g);
                  if (q.x >> 31) { vb_last = __uint_as_float(q.y); any_b = true; } else { vt_last = __uint_as_float(q.y); any_t = true; }
                  q = qn; }
               g_end = g; ndx_end = ndx;
               const float s_hi = w.v_avg_height * 0.25f * fast_rcp(gmin), s_lo = w.v_avg_height * 0.25f * fast_rcp(gmax);
               bool ok = gmin > 0 && (int)(P.rise * s_hi * lsb) + 4 <= band[my_w * 4 + 1] && (int)(P.rise * s_lo * lsb) - 3 >= band[my_w * 4 + 0]
                         && P.rise * s_lo >= P.screen_rise_v * 1.01f;
                if (P.min_peak != 0) ok = ok && (int)(P.min_peak * s_hi * lsb) + 4 <= band[my_w * 4 + 3] && (int)(P.min_peak * s_lo * lsb) - 3 >= band[my_w * 4 + 2]
                                          && P.min_peak * s_lo >= P.screen_minpk_v * 1.01f;
               if (!ok) atomicOr(&s_seq, 1u); }
            __syncthreads();
            if (cfg.debug) p4 = clock64();
            if (!s_seq) {
               // ---- (5) all lanes: the events; walker lanes: the state after the tile ----
               const int total = hoff[nwalk];
               for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
                  const int w2 = hmap[idx];
                  const int j = idx - hoff[w2];
                  const unsigned wt = s_wtab[w2];
                  const int st2 = (int)(wt & 0xff), p2 = (int)((wt >> 8) & 0xff), t2 = (int)(wt >> 16);
                  const uint4 sl = *reinterpret_cast<const uint4 *>(recs_all + (size_t)w2 * rstride + 16 * j);
                  const int r = (int)(sl.x & 0xffff), k0 = (int)((sl.x >> 16) & 0x3f);
                  const bool is_top = !(sl.x >> 31);
                  const int4 *lst = units + (s_off[st2] - gbase);
                  const int4 A = lst[r];
                  const int nr = (A.x >> 11) & 0x3f, m = (int)(short)(A.y & 0xffff);
                  const float gb = __uint_as_float(sl.w);
                  // the thresholds in effect at this detection (src/decoder.c:785-786), and the first row from k0 on that passes
                  const DevParm &P2 = cfg.parm[p2];
                  const float rise = P2.rise * (wst[w2 * 4 + 2] / 4.0f) / gb;
                  const int ri = (int)floorf(rise * lsb);
                  const int4 *mg4 = lst + s_dir[st2].nruns + (int)((unsigned)A.x >> 18);
                  int k = k0;
                  #pragma nounroll
                  for (; k < nr - 1; ++k) {
                     const int pr = k == 0 ? A.w : reinterpret_cast<const int *>(mg4)[k - 1];
                     const int dl = pr & 0xffff, dr = (int)((unsigned)pr >> 16);
                     const int mg = min(dl, dr);
                     if (mg <= ri - 1) continue;
                     if (mg >= ri + 2) break;
                     const float vm = volt(m, cfg.maxvolts);
                     const float vl = volt(is_top ? m - dl : m + dl, cfg.maxvolts), vr = volt(is_top ? m - dr : m + dr, cfg.maxvolts);
                     if (is_top ? (vm > vl + rise && vm > vr + rise) : (vm < vl - rise && vm < vr - rise)) break; }
                  const int n = (A.x & 0x7ff) + k, ld = ((A.y >> 16) & 0xff) - k;
                  const int adjcode = refine_code(&cfg, m, (int)(short)(A.z & 0xffff), A.z >> 16, gb, is_top);
                  store_event(cx, p2, t2, (unsigned)(idx0_all[w2] + j), tile0 + n, __uint_as_float(sl.y), gb, is_top, adjcode, ld); }
               if (active) {
                  const DevParm &P = cfg.parm[pidx];
                  if (nh > 0) {
                     if (any_t) { w.v_top = vt_last; w.v_lasttop = vt_last; }
                     if (any_b) { w.v_bot = vb_last; w.v_lastbot = vb_last; }
                     w.peakcount += nh; w.nevents += (unsigned)nh; w.agc_gain = g_end; w.t_lastpeak = 0;
                     if (P.agc_window) { w.heightndx = ndx_end; for (int i = 0; i < 10; ++i) cx.heights[i] = heights_bak[my_w * 10 + i]; }
                     w.blind_until = tile0 + last_blind;
                     if (!approx_thresholds(w, P, lsb)) update_thresholds(w, P, lsb); }
                  w.next = limit;
                  if (whole) {
                     const int last = (int)tn - 1;
                     w.minv = s_dir[my_st].end_min; w.cpos = tile0 + last; w.qtrig = tile0 + last + s_dir[my_st].end_ld; w.chain_pending = false; } }
               done_par = true;
               if (cfg.debug) { const long long p5 = clock64(); pa[0] += p0 - t2; pa[1] += p1 - p0; pa[2] += p2 - p1; pa[3] += p3 - p2; pa[4] += p4 - p3; pa[5] += p5 - p4; } }
            if (cfg.debug && threadIdx.x == 0) atomicAdd(&scratch->why[done_par ? 0 : 1], 1ull); }
         if (done_par) {
            if (cfg.debug) { const long long t4 = clock64(); acc0 += t1 - t0; acc1 += t2 - t1; acc2 += t4 - t2; ++ntl; }
            continue; }                                               // next group of lists / next tile
         // ================= sequential walk =================
         cx.nrec = 0;
         Walker w0;
         load_walk_fields(w0, w);
         if (is_walker) nrec_all[my_w] = 0;
         if (active) {
            const int st = my_st0;
            for (int i = 0; i < 10; ++i) heights_bak[my_w * 10 + i] = cx.heights[i];
            int why = 0;
            if (!walk_records(w, cx, pidx, trk, stop, reinterpret_cast<const CandUnit *>(units + (s_off[st] - gbase)), s_dir[st].nruns, s_dir[st], why))
               atomicOr((unsigned int *)&s_needfull, 1u);
            nrec_all[my_w] = cx.nrec; }
         __syncthreads();
         if (cfg.debug) t3 = clock64();
         if (s_needfull) {
            if (active) { load_walk_fields(w, w0); for (int i = 0; i < 10; ++i) cx.heights[i] = heights_bak[my_w * 10 + i]; }
            give_back = true;
            break; }
         if (threadIdx.x == 0) { int o = 0; for (int w2 = 0; w2 < nwalk; ++w2) { const int c = nrec_all[w2]; nrec_all[w2] = o; o += c; } nrec_all[nwalk] = o; }
         __syncthreads();
         finalize_records16(cx, recs_all, rstride, nrec_all, idx0_all, nwalk, threadIdx.x, blockDim.x);
         if (cfg.debug) { const long long t4 = clock64(); acc0 += t1 - t0; acc1 += t2 - t1; acc2 += t3 - t2; acc3 += t4 - t3; ++ntl; } }     // groups
         if (give_back) break; }
      if (cfg.debug && threadIdx.x == 0) {
         atomicAdd(&scratch->dbg2[0], (unsigned long long)acc0); atomicAdd(&scratch->dbg2[1], (unsigned long long)acc1);
         atomicAdd(&scratch->dbg2[2], (unsigned long long)acc2); atomicAdd(&scratch->dbg2[3], (unsigned long long)acc3);
         atomicAdd(&scratch->dbg[7], (unsigned long long)ntl);
         for (int i = 0; i < 6; ++i) atomicAdd(&scratch->why[2 + i], (unsigned long long)pa[i]); }
      if (walk_mode == kWalkSegs && nseg > 1) {                       // one of several segments: k_stitch joins them (or rejects them all)
         if (is_walker) {
            const DevParm &P = cfg.parm[pidx];
            if (!give_back) update_thresholds(w, P, cfg.lsb_per_volt);
            WalkState &ws = seg_end[(size_t)item * nwalk + my_w];
            load_walk_fields(ws.w, w);
            for (int i = 0; i < 10; ++i) ws.heights[i] = cx.heights[i]; }
         if (threadIdx.x == 0) seg_status[item] = give_back ? 1 : 0;
         __syncthreads();
         continue; }
      if (give_back || pre_done) {                                    // the state as of the start of tile g
         if (is_walker) {
            WalkState &ws = wstate[(size_t)b * nwalk + my_w];
            load_walk_fields(ws.w, w);
            for (int i = 0; i < 10; ++i) ws.heights[i] = cx.heights[i]; }
         if (threadIdx.x == 0) { ctl[b].next_tile = (int)g; if (give_back) ctl[b].status = kBurstNeedsFull; }
         __syncthreads();
         continue; }
      // ---- publish (as k_decode does) ----
      if (is_walker) {
         counts[((size_t)b * cfg.nparm + pidx) * ntrks + trk] = w.nevents < cx.cap ? w.nevents : cx.cap;
         if (w.flags) atomicOr(&s_flags, w.flags); }
      __syncthreads();
      if (threadIdx.x == 0) {
         bursts[b].reset_sample = reset;
         bursts[b].safe_last = (bflags & RTFE_F_UNSAFE) ? -1 : reset;
         bursts[b].end_sample = stop < nrows ? stop : nrows;
         bursts[b].flags = bflags | s_flags;
         ctl[b].status = kBurstDone; }
      __syncthreads(); } }

// ------------------------------------------------------------------------------------------------
// k_segs: cuts the bursts that are ready for the record walk into segments (single workgroup, one lane per burst).
// A burst is cut only if every walker has left the AGC start-up (the guess of a later segment's state copies the
// steady baseline) and the segments' event slots fit the burst's event regions.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_segs(const DevCfg *__restrict__ cfgp, long long nrows, const rtfe_burst *__restrict__ bursts,
                                               BurstScratch *__restrict__ scratch, const BurstCtl *__restrict__ ctl,
                                               const WalkState *__restrict__ wstate, SegTab *__restrict__ segtab,
                                               int *__restrict__ segburst, int *__restrict__ seg_status, long long max_segs) {
   __shared__ int lds[32];
   __shared__ int s_base;
   const DevCfg &cfg = *cfgp;
   const int nwalk = cfg.nparm * cfg.ntrks, S = cfg.seg_tiles;
   const long long T = cfg.tile_rows;
   if (threadIdx.x == 0) s_base = 0;
   __syncthreads();
   const int nb = scratch->nbursts;
   for (int b0 = 0; b0 < nb; b0 += 1024) {
      const int b = b0 + threadIdx.x;
      int nseg = 0; SegTab st; st.first = 0; st.nseg = 0; st.t0 = 0; st.tend = 0;
      if (b < nb && ctl[b].status == kBurstReady) {
         long long tend = (ctl[b].stop + T - 1) / T;
         const long long tmax = (nrows + T - 1) / T;
         if (tend > tmax) tend = tmax;
         st.t0 = ctl[b].next_tile; st.tend = (int)tend;
         const long long ntl = tend - st.t0;
         nseg = 1;
         if (S > 0 && cfg.mode != RTFE_PE && ntl >= S + S / 2) {
            int want = (int)((ntl + S - 1) / S);
            for (int w2 = 0; w2 < nwalk && want > 1; ++w2) {
               const Walker &w = wstate[(size_t)b * nwalk + w2].w;
               if (!((w.peakcount >= 16 && w.v_avg_height_count == 0) || w.peakcount == 0) || !w.fast) want = 1;
               const unsigned long long last_cap = (unsigned long long)((float)((ntl - (long long)(want - 1) * S) * T) * cfg.cap_frac) + 16;     // the last slot only needs room for its own tiles
               if ((unsigned long long)w.nevents + (unsigned long long)(want - 1) * (unsigned)cfg.seg_evcap + last_cap > bursts[b].event_cap) want = 1; }
            nseg = want; } }
      int total;
      const int off = block_excl_scan_1024(nseg, lds, &total);
      const int base = s_base;
      if (nseg > 0) {
         if ((long long)base + off + nseg > max_segs) nseg = 0;      // (cannot happen: max_segs covers one segment per seg_tiles tiles + one per burst)
         st.first = base + off; st.nseg = nseg;
         for (int k = 0; k < nseg; ++k) { segburst[st.first + k] = b; seg_status[st.first + k] = 1; } }
      if (b < nb) segtab[b] = st;
      __syncthreads();
      if (threadIdx.x == 0) s_base = base + total;
      __syncthreads(); }
   if (threadIdx.x == 0) { scratch->nsegs = s_base < max_segs ? s_base : (int)max_segs; scratch->queue_seg = 0; scratch->queue_stitch = 0; } }

// two walker states are "the same state" if every field the continuation reads is bit-identical (the event index and
// the accumulated flags are bookkeeping; the transition count only matters below 16, src/decode_nrzi.c:196-229)
__device__ __forceinline__ unsigned int walk_state_diff(const WalkState &a, const WalkState &b) {      // 0 = the same state
   const Walker &x = a.w, &y = b.w;
   unsigned int d = 0;
   if (x.next != y.next) d |= 1u;
   if (x.blind_until != y.blind_until) d |= 2u;
   if (__float_as_uint(x.agc_gain) != __float_as_uint(y.agc_gain)) d |= 4u;
   if (__float_as_uint(x.v_avg_height) != __float_as_uint(y.v_avg_height) || __float_as_uint(x.v_avg_height_sum) != __float_as_uint(y.v_avg_height_sum)
       || x.v_avg_height_count != y.v_avg_height_count || x.heightndx != y.heightndx) d |= 8u;
   if (__float_as_uint(x.v_top) != __float_as_uint(y.v_top) || __float_as_uint(x.v_bot) != __float_as_uint(y.v_bot)
       || __float_as_uint(x.v_lasttop) != __float_as_uint(y.v_lasttop) || __float_as_uint(x.v_lastbot) != __float_as_uint(y.v_lastbot)) d |= 16u;
   if (__float_as_uint(x.rise) != __float_as_uint(y.rise) || __float_as_uint(x.reqmin) != __float_as_uint(y.reqmin) || x.thr_dirty != y.thr_dirty
       || x.rise_lo != y.rise_lo || x.rise_hi != y.rise_hi || x.min_lo != y.min_lo || x.min_hi != y.min_hi) d |= 32u;
   if (x.minv != y.minv || x.qtrig != y.qtrig || x.cpos != y.cpos || x.chain_pending != y.chain_pending) d |= 64u;
   if (x.datablock != y.datablock || x.bit1_up != y.bit1_up || x.fast != y.fast
       || !((x.peakcount >= 16 && y.peakcount >= 16) || x.peakcount == y.peakcount)) d |= 128u;
   for (int i = 0; i < 10; ++i) if (__float_as_uint(a.heights[i]) != __float_as_uint(b.heights[i])) d |= 128u;
   return d; }

// ------------------------------------------------------------------------------------------------
// k_stitch: one 256-lane workgroup per segmented burst.  Accepts the segments only if each one's state at its first own
// tile equals its predecessor's final state (then, by induction from segment 0, every segment ran from the true
// state); moves the events of segments 1.. down behind their predecessors' and publishes the burst.  Otherwise the
// burst goes to the second k_decode pass with the state it had when it was cut.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_stitch(const DevCfg *__restrict__ cfgp, long long nrows, rtfe_burst *__restrict__ bursts,
                                               BurstScratch *__restrict__ scratch, uint32_t *__restrict__ counts, rtfe_event *__restrict__ events,
                                               BurstCtl *__restrict__ ctl, WalkState *__restrict__ wstate, const SegTab *__restrict__ segtab,
                                               const WalkState *__restrict__ seg_start, const WalkState *__restrict__ seg_end,
                                               const int *__restrict__ seg_status) {
   __shared__ int s_burst, s_firstbad;
   __shared__ unsigned int s_flags;
   const DevCfg &cfg = *cfgp;
   const int ntrks = cfg.ntrks, nwalk = cfg.nparm * ntrks;
   for (;;) {
      if (threadIdx.x == 0) { s_burst = atomicAdd(&scratch->queue_stitch, 1); s_firstbad = 0x7fffffff; s_flags = 0; }
      __syncthreads();
      const int b = s_burst;
      if (b >= scratch->nbursts) break;
      const SegTab st = segtab[b];
      if (ctl[b].status != kBurstReady || st.nseg <= 1) { __syncthreads(); continue; }
      // ---- the first segment that did not finish, or whose start is not its predecessor's end ----
      for (int i = threadIdx.x; i < st.nseg * nwalk; i += blockDim.x) {
         const int sg = i / nwalk, w2 = i - sg * nwalk;
         unsigned int d = seg_status[st.first + sg] != 0 ? 256u : 0u;
         if (sg > 0) d |= walk_state_diff(seg_end[(size_t)(st.first + sg - 1) * nwalk + w2], seg_start[(size_t)(st.first + sg) * nwalk + w2]);
         if (d) {
            atomicMin(&s_firstbad, sg);
            for (int k = 0; k < 8; ++k) if ((d >> k) & 1) atomicAdd(&scratch->why[k], 1ull);          // statistics (tools/gpu_segs.py)
            if (d & 256u) atomicAdd(&scratch->dbg2[7], 1ull); } }
      __syncthreads();
      const int ngood = s_firstbad < st.nseg ? s_firstbad : st.nseg;  // segments 0 .. ngood-1 ran from the true state
      if (ngood == 0) {                                               // (segment 0 itself gave up: the burst as it was cut)
         if (threadIdx.x == 0) { ctl[b].status = kBurstNeedsFull; atomicAdd(&scratch->seg_failed, 1); }
         __syncthreads();
         continue; }
      // ---- the events of segments 1 .. ngood-1 move down behind their predecessors' ----
      const rtfe_burst B = bursts[b];
      for (int w2 = 0; w2 < nwalk; ++w2) {
         const unsigned int n0 = wstate[(size_t)b * nwalk + w2].w.nevents;
         rtfe_event *reg = events + B.event_base + (size_t)w2 * B.event_cap;       // (w2 = parmset * ntrks + track)
         unsigned int dst = seg_end[(size_t)st.first * nwalk + w2].w.nevents;
         unsigned int fl = seg_end[(size_t)st.first * nwalk + w2].w.flags;
         for (int sg = 1; sg < ngood; ++sg) {
            const Walker &e = seg_end[(size_t)(st.first + sg) * nwalk + w2].w;
            const unsigned int base = n0 + (unsigned)sg * (unsigned)cfg.seg_evcap;
            const unsigned int cnt = e.nevents - base;
            fl |= e.flags;
            for (unsigned int k0 = 0; k0 < cnt; k0 += blockDim.x) {               // left to right, a row of lanes at a time: dst < base always
               const unsigned int k = k0 + threadIdx.x;
               rtfe_event ev;
               if (k < cnt) ev = reg[base + k];
               __syncthreads();
               if (k < cnt) reg[dst + k] = ev;
               __syncthreads(); }
            dst += cnt; }
         if (threadIdx.x == 0) {
            if (ngood == st.nseg) { counts[(size_t)b * nwalk + w2] = dst < B.event_cap ? dst : B.event_cap; if (fl) s_flags |= fl; }
            else {                                                    // the second k_decode pass continues behind the last good segment
               const WalkState &src = seg_end[(size_t)(st.first + ngood - 1) * nwalk + w2];
               WalkState &out = wstate[(size_t)b * nwalk + w2];          // (the fields the record walk never touches stay as they were)
               const long long trust = out.w.trust_from;
               load_walk_fields(out.w, src.w);
               for (int i = 0; i < 10; ++i) out.heights[i] = src.heights[i];
               out.w.nevents = dst; out.w.flags = fl; out.w.trust_from = trust; } } }
      __syncthreads();
      if (threadIdx.x == 0) {
         if (ngood == st.nseg) {
            const long long stop = ctl[b].stop;
            bursts[b].reset_sample = ctl[b].reset;
            bursts[b].safe_last = (ctl[b].bflags & RTFE_F_UNSAFE) ? -1 : ctl[b].reset;
            bursts[b].end_sample = stop < nrows ? stop : nrows;
            bursts[b].flags = ctl[b].bflags | s_flags;
            ctl[b].status = kBurstDone; }
         else { ctl[b].next_tile = st.t0 + ngood * cfg.seg_tiles; ctl[b].status = kBurstNeedsFull; atomicAdd(&scratch->seg_failed, 1); } }
      __syncthreads(); } }

__global__ void __launch_bounds__(kDecodeThreads, 2) k_decode(const DevCfg *__restrict__ cfgp, const int16_t *__restrict__ rows,
                                                           long long nrows, long long row_base,
                                                           rtfe_burst *__restrict__ bursts, BurstScratch *__restrict__ scratch,
                                                           uint32_t *__restrict__ counts, rtfe_event *__restrict__ events,
                                                           uint32_t parmset_mask, int screen_off, int single_exact,
                                                           const TileDir *__restrict__ dir, const CandUnit *__restrict__ pool,
                                                           int mode, BurstCtl *__restrict__ ctl, WalkState *__restrict__ wstate) {
#ifdef RTFE_CPU_EMUL
   unsigned char *smem = g_dyn_smem;            // tests/cpu_emul only
#else
   extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
   __shared__ DevCfg cfg;
   __shared__ int s_burst;
   __shared__ long long s_min;
   __shared__ unsigned int s_flags;
   __shared__ int s_needfull;
   __shared__ TileDir s_dir[kMaxScreens * RTFE_MAXTRKS];
   __shared__ int s_off[kMaxScreens * RTFE_MAXTRKS + 1];
   __shared__ int s_recoff[kDecodeThreads + 1], s_idx0[kDecodeThreads];
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
   const LdsLayout L = lds_layout(cfg, true);
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
   CandUnit *runs_all = reinterpret_cast<CandUnit *>(smem + L.runs);     // the tile's lists, packed (lds_units), over the sample tile
   Walker *walkers = reinterpret_cast<Walker *>(smem + L.walkers);       // [nwalk]
   Walker *walkers_next = reinterpret_cast<Walker *>(smem + L.walkers_next);  // [nwalk] result of an optimistic tile, committed only if all lanes agree
   float *heights_bak = reinterpret_cast<float *>(smem + L.heights_bak);     // [nwalk][10]
   cx.rec_cap = cfg.rec_cap; cx.rec_cap16 = cfg.rec_cap * (int)sizeof(Rec) / (int)sizeof(Rec16);
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

   // mode: kDecodeAll = whole bursts (exact scans, -zeros); kDecodeHead = start every burst and hand it to k_walk as
   // soon as its walkers are on the screened path; kDecodeResume = finish the bursts k_walk had to give back
   for (;;) {
      if (threadIdx.x == 0) { s_burst = atomicAdd((mode == kDecodeResume || mode == kDecodeRedo) ? &scratch->queue_resume : &scratch->queue, 1); s_flags = 0; }
      __syncthreads();
      const int b = s_burst;
      if (b >= scratch->nbursts) break;
      if ((mode == kDecodeResume || mode == kDecodeRedo) && ctl[b].status != kBurstNeedsFull) { __syncthreads(); continue; }
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
      if (mode == kDecodeResume) {
         reset = ctl[b].reset; stop = ctl[b].stop; bflags = ctl[b].bflags; g_first = ctl[b].next_tile;
         if (is_walker) {
            const WalkState &ws = wstate[(size_t)b * nwalk + my_w];
            walkers[my_w] = ws.w;
            for (int i = 0; i < 10; ++i) cx.heights[i] = ws.heights[i]; } }
      else {
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
      bool handed_over = false;
      // ---- tiles of the tape-global grid that intersect [reset, stop) ----
      const long long T = cfg.tile_rows;
      for (long long g = g_first; g * T < stop; ++g) {
         const long long tile0 = g * T;
         const long long tn = (tile0 + T <= nrows) ? T : nrows - tile0;
         if (tn <= 0) break;
         long long c0 = 0, c1 = 0, c2 = 0;
         cx.tile.row0 = tile0; cx.tile.nrows = (int)tn;
         __syncthreads();
         if (mode == kDecodeHead) {                               // can k_walk take over from this tile on?
            if (threadIdx.x == 0) s_needfull = 0;
            __syncthreads();
            if (active && !(walkers[my_w].fast && tile0 - kScreenHalo >= walkers[my_w].trust_from)) atomicOr((unsigned int *)&s_needfull, 1u);
            __syncthreads();
            if (!s_needfull) {
               if (is_walker) {
                  WalkState &ws = wstate[(size_t)b * nwalk + my_w];
                  ws.w = walkers[my_w];
                  for (int i = 0; i < 10; ++i) ws.heights[i] = cx.heights[i]; }
               if (threadIdx.x == 0) {
                  BurstCtl c; c.reset = reset; c.stop = stop; c.next_tile = (int)g; c.status = kBurstReady; c.bflags = bflags; c.pad = 0;
                  ctl[b] = c; }
               handed_over = true;
               break; }
            __syncthreads(); }
         if (cfg.debug) c0 = clock64();
         // ---- decide the whole tile from the candidate records k_screen left in HBM ----
         bool done_tile = false;
         if (dir && !screen_off && !cfg.find_zeros && mode != kDecodeHead) {     // (the heads may run beside k_screen: no lists yet)
            const int nst = cfg.nscreens * ntrks;
            if (threadIdx.x == 0) s_needfull = 0;
            if (threadIdx.x < nst) s_dir[threadIdx.x] = dir[g * nst + threadIdx.x];
            __syncthreads();
            long long o1 = 0, o2 = 0, o3 = 0;
            if (cfg.debug) o1 = clock64();
            if (threadIdx.x <= nst) {                                  // where each list goes in LDS (packed)
               int o = 0; bool bad = false;
               for (int st = 0; st < (int)threadIdx.x; ++st) { if (s_dir[st].count == 0xFFFF) bad = true; o += s_dir[st].count; }
               s_off[threadIdx.x] = bad ? (1 << 30) : o; }
            __syncthreads();
            const bool avail = s_off[nst] <= cfg.lds_units;
            if (avail) {
               const int4 *src = reinterpret_cast<const int4 *>(pool) + (size_t)g * nst * cfg.run_cap;
               for (int i = threadIdx.x; i < s_off[nst]; i += blockDim.x) reinterpret_cast<int4 *>(runs_all)[i] = src[i];
               __syncthreads();
               if (cfg.debug) o2 = clock64();
               cx.nrec = 0;
               if (active) {
                  const int st = cfg.parm[pidx].screen * ntrks + trk;
                  Walker w = walkers[my_w];
                  for (int i = 0; i < 10; ++i) heights_bak[my_w * 10 + i] = cx.heights[i];     // part of the walker's state
                  s_idx0[my_w] = (int)w.nevents;
                  int why = 0;
                  if (walk_records(w, cx, pidx, trk, stop, runs_all + s_off[st], s_dir[st].nruns, s_dir[st], why))
                     walkers_next[my_w] = w;
                  else { atomicOr((unsigned int *)&s_needfull, 1u); if (cfg.debug) atomicAdd(&scratch->why[why & 7], 1ull); } }
               if (is_walker) nrec_all[my_w] = cx.nrec;
               __syncthreads();
               if (cfg.debug) o3 = clock64();
               if (s_needfull && active) for (int i = 0; i < 10; ++i) cx.heights[i] = heights_bak[my_w * 10 + i];
               if (!s_needfull) {
                  if (is_walker && active) walkers[my_w] = walkers_next[my_w];
                  if (threadIdx.x == 0) { int o = 0; for (int w2 = 0; w2 < nwalk; ++w2) { const int c = (parmset_mask >> (w2 / ntrks)) & 1 ? nrec_all[w2] : 0; s_recoff[w2] = o; o += c; } s_recoff[nwalk] = o; }
                  __syncthreads();
                  finalize_records16(cx, reinterpret_cast<const unsigned char *>(recs_all), cfg.rec_cap * (int)sizeof(Rec), s_recoff, s_idx0, nwalk, threadIdx.x, blockDim.x);
                  done_tile = true;
                  if (cfg.debug && threadIdx.x == 0) {
                     const long long o4 = clock64();
                     atomicAdd(&scratch->dbg[7], 1ull);
                     atomicAdd(&scratch->dbg2[0], (unsigned long long)(o1 - c0)); atomicAdd(&scratch->dbg2[1], (unsigned long long)(o2 - o1));
                     atomicAdd(&scratch->dbg2[2], (unsigned long long)(o3 - o2)); atomicAdd(&scratch->dbg2[3], (unsigned long long)(o4 - o3)); } }
               __syncthreads(); } }
         if (done_tile) continue;
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
      if (handed_over) { __syncthreads(); continue; }
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
