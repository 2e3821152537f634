// rtfe_zeros.hip — k_zeros: the -zeros front end (lookfor_zerocrossing, src/decoder.c:617-649) as a kernel of its own.
//
// The zero-crossing detector has no AGC feedback and no window: k_decode's general walker (250 VGPRs: two waves per SIMD) was
// carrying it as one of its modes.  Here it has a state of seven words per track, so eight waves fit a SIMD and the latency of
// the per-row chain (read -> a dozen selects -> next row) is covered by other waves; and, eight rows being in flight per lane
// anyway, the samples are read where they lie in HBM (the nine tracks of a row are 18 contiguous bytes, a 64-byte line serves
// 3.5 steps): no tile in LDS, nine workgroups per CU instead of four.  Same algorithm, same functions
// (zeros_tile_parallel: a lane per (track, 64-row sub-segment) with verified warm-up and in-place repair; walk_zeros: the sequential
// walk of a burst's first and last tile), same publication as k_decode(kDecodeAll): one workgroup per burst, tiles of the
// tape-global grid.  Exact scans of single attempts (rtfe_scan_exact) stay with k_decode.  Included behind rtfe_kernels.hip.

namespace rtfe {

struct ZWalker {           // what lookfor_zerocrossing keeps per track (+ where the walk stands)
   long long start, next;
   int   z_prev, z_top, z_bot;
   bool  z_up_pending, z_dn_pending;
   long long z_ttop_row, z_tbot_row;
   unsigned int nevents, flags;
};

struct ZerosLds { unsigned tile, lanes, walkers, total; };
__host__ __device__ inline ZerosLds lds_layout_zeros(const DevCfg &c) {
   ZerosLds L;
   L.tile = 0;
   // (-invert negates on the way into LDS; without it the lanes read the tape's rows where they lie and no tile is staged)
   unsigned off = c.invert ? lds_align16((unsigned)c.ntrks * (unsigned)c.ldw * 2u + 16u) : 0u;
   L.lanes = off;   off = lds_align16(off + (unsigned)c.ntrks * ((unsigned)c.tile_rows / (unsigned)kZcSub) * (unsigned)sizeof(ZcLane));
   L.walkers = off; off = lds_align16(off + (unsigned)c.ntrks * (unsigned)sizeof(ZWalker));
   L.total = off;
   return L; }

__global__ void __launch_bounds__(128, 4) k_zeros(const DevCfg *__restrict__ cfgp, const int16_t *__restrict__ rows, long long nrows, long long row_base,
                                                  rtfe_burst *__restrict__ bursts, BurstScratch *__restrict__ scratch,
                                                  uint32_t *__restrict__ counts, rtfe_event *__restrict__ events) {
#ifdef RTFE_CPU_EMUL
   unsigned char *smem = g_dyn_smem;            // tests/cpu_emul only
#else
   extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
   __shared__ DevCfg cfg;
   __shared__ int s_burst;
   __shared__ unsigned int s_flags;
   __shared__ int s_ok[2 * RTFE_MAXTRKS + 2];
   for (int i = threadIdx.x; i < (int)(sizeof(DevCfg) / 4); i += blockDim.x) reinterpret_cast<int *>(&cfg)[i] = reinterpret_cast<const int *>(cfgp)[i];
   __syncthreads();
   const int ntrks = cfg.ntrks;
   const ZerosLds L = lds_layout_zeros(cfg);
   Ctx cx;
   cx.cfg = &cfg;
   cx.row_base = row_base;
   cx.tile.x = reinterpret_cast<int16_t *>(smem + L.tile);
   cx.tile.ldw = cfg.ldw; cx.tile.halo = cfg.halo_rows; cx.tile.colof = cfg.trk_to_head; cx.tile.ntrks = ntrks; cx.tile.skew = cfg.skew;
   cx.tile.bits = nullptr; cx.tile.bstride = 0; cx.tile.ldpos = nullptr; cx.tile.ldstride = 0; cx.tile.fd = nullptr;
   cx.heights = nullptr; cx.recs = nullptr; cx.rec_cap = 0; cx.nrec = 0;
   ZcLane *lanes = reinterpret_cast<ZcLane *>(smem + L.lanes);
   ZWalker *walkers = reinterpret_cast<ZWalker *>(smem + L.walkers);
   const int trk = threadIdx.x;
   const bool is_walker = trk < ntrks;
   const long long T = cfg.tile_rows;
   const bool par = cfg.zc_parallel && ntrks * (cfg.tile_rows / kZcSub) <= (int)blockDim.x;
   for (;;) {
      if (threadIdx.x == 0) { s_burst = atomicAdd(&scratch->queue, 1); s_flags = 0; }
      __syncthreads();
      const int b = s_burst;
      if (b >= scratch->nbursts) break;
      const int nb = scratch->nbursts_total;
      const rtfe_burst B = bursts[b];
      const bool exact = B.flags & RTFE_F_EXACT_START;
      cx.events = events + B.event_base;
      cx.cap = B.event_cap;
      // any restart inside the zone is equivalent for this detector: the zone's last kMarginRows rows
      long long reset = B.reset_sample;
      unsigned int bflags = B.flags;
      if (!exact) {
         if (B.zone_end - B.zone_first < kMarginRows + 64) { reset = B.zone_end - kMarginRows; bflags |= RTFE_F_UNSAFE; }
         else reset = B.zone_end - kMarginRows; }
      long long stop = nrows;
      if (b + 1 < nb) {
         const rtfe_burst NB = bursts[b + 1];
         stop = NB.zone_end - kMarginRows;
         if (cfg.tail_rows > 0 && NB.zone_first + cfg.tail_rows < stop) stop = NB.zone_first + cfg.tail_rows; }
      if (is_walker) {
         ZWalker w = {};
         w.start = reset + trk; w.next = reset; w.z_ttop_row = 0; w.z_tbot_row = 0;
         walkers[trk] = w; }
      cx.tile.reset = reset;
      for (long long g = reset / T; g * T < stop; ++g) {
         const long long tile0 = g * T;
         const long long tn = (tile0 + T <= nrows) ? T : nrows - tile0;
         if (tn <= 0) break;
         cx.tile.row0 = tile0; cx.tile.nrows = (int)tn;
         __syncthreads();
         if (cfg.invert) { load_tile(&cfg, cx.tile, rows, nrows); __syncthreads(); }
         else cx.tile.x = const_cast<int16_t *>(rows) + (tile0 - cx.tile.halo) * ntrks;      // (rows in front of the restart row are never read)
         if (par) {
            if (cfg.invert) zeros_tile_parallel<ZWalker, false>(cx, walkers, lanes, s_ok, stop, cfg.debug ? scratch->dbg2 : (unsigned long long *)nullptr);
            else zeros_tile_parallel<ZWalker, true>(cx, walkers, lanes, s_ok, stop, cfg.debug ? scratch->dbg2 : (unsigned long long *)nullptr); }      // (the rows where they lie in HBM)
         if (par) {
            if (cfg.debug && (int)threadIdx.x < ntrks) { atomicAdd(&scratch->dbg2[4], 1ull); if (s_ok[threadIdx.x]) atomicAdd(&scratch->dbg2[5], 1ull); if (threadIdx.x == 0) atomicAdd(&scratch->dbg2[6], 1ull); } }      // (RTFE_DEBUG=1: tools/gpu_zeros_phase.py)
         else { if (threadIdx.x < (unsigned)ntrks) s_ok[threadIdx.x] = 0; __syncthreads(); }
         if (is_walker && !s_ok[trk]) {                              // the burst's first and last tile, partial tiles: row by row
            ZWalker w = walkers[trk];
            walk_zeros(w, cx, trk, stop);
            walkers[trk] = w; } }
      __syncthreads();
      // ---- publish (as k_decode does for this detector) ----
      if (is_walker) {
         const ZWalker &w = walkers[trk];
         unsigned int wf = w.flags;
         // history a restart would not have
         if (w.z_up_pending || w.z_dn_pending || w.z_top >= cfg.zc_peak_i || w.z_bot <= -cfg.zc_peak_i) wf |= RTFE_F_STATE_AT_END;
         counts[((size_t)b * cfg.nparm + 0) * ntrks + trk] = w.nevents < cx.cap ? w.nevents : cx.cap;
         if (wf) atomicOr(&s_flags, wf); }
      for (int i = threadIdx.x; i < (cfg.nparm - 1) * ntrks; i += blockDim.x) counts[((size_t)b * cfg.nparm + 1) * ntrks + i] = 0;      // (the detector does not depend on the parameter set: set 0 only)
      __syncthreads();
      if (threadIdx.x == 0) {
         const long long hard_end = nrows;
         bursts[b].reset_sample = reset;
         bursts[b].safe_last = (bflags & RTFE_F_UNSAFE) ? -1 : (!exact ? B.zone_end - ntrks - 2 : reset);
         bursts[b].end_sample = stop < hard_end ? stop : hard_end;
         bursts[b].flags = bflags | s_flags; }
      __syncthreads(); } }

}  // namespace rtfe
