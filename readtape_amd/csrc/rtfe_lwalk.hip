// rtfe_lwalk.hip — k_lwalk: the record walk with ONE LANE PER WALKER (round 2).
//
// k_walk gives a burst segment a whole workgroup: its nine walkers (tracks) share staged lists in LDS and meet at eleven
// barriers per tile, the other lanes help with the parallel tile path.  Measured, that is latency x residency (24 % VALU busy).
// A walker's record walk needs no help: its list of one tile is ~46 sixteen-byte units that it reads once, front to back.
// Here a lane IS a walker: it keeps its state in registers, walks its own list, stores its events as it detects them; 64 / nwalk
// work items (segments, or whole short bursts) share a wave.  Per round (one tile of every item) the walkers of an item copy
// the tile's lists - contiguous in the pool - into the item's LDS region, sixteen 16-byte loads in flight per lane (a walker
// reading its list unit by unit from HBM pays one load latency per unit: measured 90 us per tile), then every lane walks.
// No barriers but the one behind the copy, no cross-lane phases.  The work items, the guessed start states, the event slots
// and the joins are k_walk's (k_segs / k_stitch are unchanged): only who walks differs.
// Included behind rtfe_kernels.hip.  Used when nwalk = parameter sets x tracks <= 32 (two or more items per wave).

namespace rtfe {

// the walkers of one item are lanes [sub * nwalk, (sub + 1) * nwalk) of the wave: any / all over them
__device__ __forceinline__ bool item_any(bool pred, u64 item_mask) { return (__ballot(pred ? 1 : 0) & item_mask) != 0; }

__global__ void __launch_bounds__(64) k_lwalk(const DevCfg *__restrict__ cfgp, long long nrows, long long row_base,
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
   __shared__ float s_heights[64 * 10], s_heights_bak[64 * 10];
   __shared__ int s_base;
   __shared__ TileDir s_dir[64];                                    // the round's directory entries: item sub's at [sub * nwalk, + nst)
   __shared__ DevCfg cfg;                                           // (through the pointer every field read would be a global load)
   int4 *lds_units = reinterpret_cast<int4 *>(smem);                // [items per wave][lds_units]
   for (int i = threadIdx.x; i < (int)(sizeof(DevCfg) / 4); i += blockDim.x) reinterpret_cast<int *>(&cfg)[i] = reinterpret_cast<const int *>(cfgp)[i];
   __syncthreads();
   const int ntrks = cfg.ntrks, nst = cfg.nscreens * ntrks, nwalk = cfg.nparm * ntrks;
   const int ipw = 64 / nwalk;                                       // items per wave
   const int lane = threadIdx.x;
   const int sub = lane / nwalk, my_w = lane - sub * nwalk;
   const bool has_slot = sub < ipw;
   const int pidx = my_w / ntrks, trk = my_w - pidx * ntrks;
   const u64 item_mask = has_slot ? (((nwalk >= 64 ? ~0ull : ((1ull << nwalk) - 1ull))) << (sub * nwalk)) : 0ull;
   const long long T = cfg.tile_rows;
   const DevParm &P = cfg.parm[pidx];
   const int my_st = P.screen * ntrks + trk;
   Ctx cx;
   cx.cfg = &cfg;
   cx.row_base = row_base;
   cx.tile.x = nullptr; cx.tile.ldw = 0; cx.tile.halo = 0; cx.tile.colof = cfg.trk_to_head; cx.tile.ntrks = ntrks; cx.tile.skew = cfg.skew;
   cx.tile.bits = nullptr; cx.tile.bstride = 0; cx.tile.ldpos = nullptr; cx.tile.ldstride = 0; cx.tile.fd = nullptr;
   cx.heights = s_heights + lane * 10;
   float *hbak = s_heights_bak + lane * 10;
   cx.rec_cap = 0; cx.rec_cap16 = 0; cx.recs = nullptr; cx.nrec = 0;
   const bool segs = walk_mode == kWalkSegs;
   for (;;) {
      if (lane == 0) s_base = atomicAdd(segs ? &scratch->queue_seg : &scratch->queue_walk, ipw);
      __syncthreads();
      const int base = s_base;
      __syncthreads();
      const int nitems = segs ? scratch->nsegs : scratch->nbursts;
      if (base >= nitems) break;
      const int item = base + sub;
      int b = -1;
      if (has_slot && item < nitems) { b = segs ? segburst[item] : item; if (ctl[b].status != kBurstReady) b = -1; }
      bool live = b >= 0;                                           // this lane walks
      // ---- the item: which tiles, from which state, into which slot of the walker's event region ----
      Walker w = {};
      long long reset = 0, stop = 0, g = 0, g_hi = 0, seg_lo = 0;
      int seg = 0, nseg = 1;
      unsigned int ev_base = 0, ev_limit = 0, bflags = 0;
      if (live) {
         const rtfe_burst B = bursts[b];
         cx.events = events + B.event_base;
         cx.cap = B.event_cap;
         reset = ctl[b].reset; stop = ctl[b].stop; bflags = ctl[b].bflags;
         const WalkState &ws = wstate[(size_t)b * nwalk + my_w];
         load_walk_fields(w, ws.w);
         for (int i = 0; i < 10; ++i) cx.heights[i] = ws.heights[i];
         g = ctl[b].next_tile;
         g_hi = (stop + T - 1) / T; seg_lo = g;
         ev_limit = cx.cap;
         if (segs) {
            const SegTab stb = segtab[b];
            seg = item - stb.first; nseg = stb.nseg;
            seg_lo = stb.t0 + (long long)seg * cfg.seg_tiles;
            g_hi = seg + 1 < nseg ? seg_lo + cfg.seg_tiles : stb.tend;
            g = seg_lo;
            if (nseg > 1) {
               ev_base = w.nevents + (unsigned)seg * (unsigned)cfg.seg_evcap; ev_limit = ev_base + (unsigned)cfg.seg_evcap;
               if (ev_limit > cx.cap) ev_limit = cx.cap;
               if (seg > 0) {                                        // the guessed state seg_warm tiles ahead of the segment (see k_walk)
                  g = seg_lo - cfg.seg_warm;
                  w.blind_until = -1; w.next = g * T; w.trust_from = -(1ll << 40); w.flags = 0;
                  w.cpos = g * T - 1; w.chain_pending = false;
                  w.nevents = ev_base; } } } }
      cx.tile.reset = reset;
      const long long g_first = g;
      // every lane of the wave runs the same number of rounds (the wave intrinsics below want all of them)
      int my_rounds = 0;
      if (live) { long long hi = g_hi; const long long by_stop = (stop + T - 1) / T; if (by_stop < hi) hi = by_stop; my_rounds = hi > g ? (int)(hi - g) : 0; }
      int rounds = my_rounds;
      #pragma unroll
      for (int o = 32; o >= 1; o >>= 1) { const int other = __shfl(rounds, (lane + o) & 63); rounds = max(rounds, other); }
      bool give_back = false, pre_done = false;                     // (item-wide verdicts: the same in all its lanes)
      long long g_verdict = 0;                                      // ... reached at the start of this tile
      for (int r = 0; r < rounds; ++r, ++g) {
         const bool act = live && !give_back && !pre_done && g < g_hi && g * T < stop && (g * T < nrows);
         if (__ballot(act ? 1 : 0) == 0) break;
         const long long tile0 = g * T;
         const long long tn = (tile0 + T <= nrows) ? T : nrows - tile0;
         if (act) { cx.tile.row0 = tile0; cx.tile.nrows = (int)tn; }
         if (act && segs && seg > 0 && g == seg_lo) {
            // end of the warm-up: the state the segment really starts from (its detections so far are dropped)
            w.nevents = ev_base;
            update_thresholds(w, P, cfg.lsb_per_volt);
            WalkState &ws = seg_start[(size_t)item * nwalk + my_w];
            load_walk_fields(ws.w, w);
            for (int i = 0; i < 10; ++i) ws.heights[i] = cx.heights[i]; }
         if (walk_mode == kWalkPre) {                               // has every walker of the burst left the AGC start-up?
            const bool unsettled = act && g > g_first && !((w.peakcount >= 16 && w.v_avg_height_count == 0) || w.peakcount == 0);
            const bool any_unsettled = item_any(unsettled, item_mask);
            if (act && g > g_first && (!any_unsettled || g - g_first >= 6)) { pre_done = true; g_verdict = g; } }
         const bool walk_now = act && !pre_done;
         // ---- the tile's lists of every item -> LDS (in groups of consecutive lists that fit the item's region: usually one),
         //      copied by the item's own walkers from its slot of the pool; then every walker whose list is there walks ----
         bool ok = true;
         Walker w0;
         if (walk_now) {
            load_walk_fields(w0, w);
            for (int i = 0; i < 10; ++i) hbak[i] = cx.heights[i]; }
         if (walk_now && my_w < nst) s_dir[sub * nwalk + my_w] = dir[(size_t)g * nst + my_w];       // (one load latency, not nst of them)
         __syncthreads();
         const TileDir *d = s_dir + sub * nwalk;
         int4 *mine = lds_units + (size_t)sub * cfg.lds_units;
         long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
         if (cfg.debug) c0 = clock64();
         int lists_done = 0;                                        // lists [0, lists_done) of the item's tile have been walked
         for (int pass = 0; pass < nst; ++pass) {
            const bool item_bad = item_any(walk_now && !ok, item_mask);       // (the copy below is the whole item's job: all of its walkers, or none)
            const bool more = walk_now && !item_bad && lists_done < nst;
            if (__ballot(more ? 1 : 0) == 0) break;
            int pre = 0, gtot = 0, hi = lists_done, my_goff = -1;
            TileDir td; td.count = 0; td.nruns = 0; td.end_ld = 0; td.pad = 0; td.end_min = 0;
            if (more) {
               for (int s2 = 0; s2 < nst; ++s2) {
                  const TileDir e = d[s2];
                  if (e.count == 0xFFFF) ok = false;
                  if (s2 < lists_done) pre += e.count;
                  else if (s2 == hi && gtot + (int)e.count <= cfg.lds_units) { if (s2 == my_st) { my_goff = gtot; td = e; } gtot += e.count; ++hi; } }
               if (hi == lists_done) ok = false; }                  // (one list alone does not fit: the sample path takes the burst)
            if (more && ok) {
               const int4 *src = reinterpret_cast<const int4 *>(pool) + (size_t)g * nst * cfg.run_cap + pre;
               for (int i0 = my_w; i0 < gtot; i0 += 16 * nwalk) {
                  int4 q[16];
                  #pragma unroll
                  for (int k = 0; k < 16; ++k) { const int i = i0 + k * nwalk; q[k] = src[i < gtot ? i : i0]; }
                  #pragma unroll
                  for (int k = 0; k < 16; ++k) { const int i = i0 + k * nwalk; if (i < gtot) mine[i] = q[k]; } } }
            __syncthreads();
            if (cfg.debug) c1 = clock64();
            if (more && ok && my_goff >= 0) {
               int why = 0;
               ok = walk_records<true>(w, cx, pidx, trk, stop, reinterpret_cast<const CandUnit *>(mine + my_goff), td.nruns, td, why, (segs && nseg > 1) ? ev_limit : 0xffffffffu);
               if (!ok && cfg.debug) atomicAdd(&scratch->why[why & 7], 1ull); }
            if (cfg.debug) c2 = clock64();
            __syncthreads();
            if (cfg.debug) { c3 = clock64(); if (lane == 0) { atomicAdd(&scratch->dbg2[0], (unsigned long long)(c1 - c0)); atomicAdd(&scratch->dbg2[1], (unsigned long long)(c2 - c1));
                                                              atomicAdd(&scratch->dbg2[2], (unsigned long long)(c3 - c2)); atomicAdd(&scratch->dbg[7], 1ull); } c0 = c3; }
            if (more) lists_done = hi; }
         // one walker that cannot go on stops the whole item at the start of this tile (the give-back is a burst's, not a walker's)
         const bool any_bad = item_any(walk_now && !ok, item_mask);
         if (walk_now && any_bad) {
            load_walk_fields(w, w0);
            for (int i = 0; i < 10; ++i) cx.heights[i] = hbak[i];
            give_back = true; g_verdict = g; } }
      // ---- what becomes of the item ----
      const bool finished = live && !give_back && !pre_done;
      unsigned int fl = 0;
      if (finished && !(segs && nseg > 1)) fl = w.flags;
      // OR of the walkers' flags over the item (wave rotation; lanes of other items contribute to their own)
      unsigned int item_flags = 0;
      for (int k = 0; k < nwalk; ++k) {
         const unsigned int other = (unsigned int)__shfl((int)fl, sub * nwalk + k < 64 ? sub * nwalk + k : lane);
         item_flags |= other; }
      if (!live) continue;
      if (segs && nseg > 1) {                                        // one of several segments: k_stitch joins them (or rejects them all)
         if (!give_back) update_thresholds(w, P, cfg.lsb_per_volt);
         WalkState &ws = seg_end[(size_t)item * nwalk + my_w];
         load_walk_fields(ws.w, w);
         for (int i = 0; i < 10; ++i) ws.heights[i] = cx.heights[i];
         if (my_w == 0) seg_status[item] = give_back ? 1 : 0;
         continue; }
      if (give_back || pre_done) {                                    // the state as of the start of tile g_verdict: the burst goes on elsewhere
         WalkState &ws = wstate[(size_t)b * nwalk + my_w];
         load_walk_fields(ws.w, w);
         for (int i = 0; i < 10; ++i) ws.heights[i] = cx.heights[i];
         if (my_w == 0) { ctl[b].next_tile = (int)g_verdict; if (give_back) ctl[b].status = kBurstNeedsFull; }
         continue; }
      // ---- publish (as k_walk does) ----
      counts[((size_t)b * cfg.nparm + pidx) * ntrks + trk] = w.nevents < cx.cap ? w.nevents : cx.cap;
      if (my_w == 0) {
         bursts[b].reset_sample = reset;
         bursts[b].safe_last = (bflags & RTFE_F_UNSAFE) ? -1 : reset;
         bursts[b].end_sample = stop < nrows ? stop : nrows;
         bursts[b].flags = bflags | item_flags;
         ctl[b].status = kBurstDone; } } }

}  // namespace rtfe
