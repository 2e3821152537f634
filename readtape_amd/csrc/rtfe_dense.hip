// rtfe_dense.hip — the sample path in two passes, for the formats whose window always holds a top AND a bottom (PE, GCR):
//
//   k_dseg    dense and stateless.  The tape is cut into SUB-SEGMENTS of kDsSub rows on a fixed grid; a lane per (distinct
//             parameter set, track, sub-segment) walks the detector of src/decoder.c:751-810 over the rows of its sub-segment on
//             a tile in LDS (the candidate screen of rtfe_kernels.hip in front of it), started a warm-up early with no countdown
//             pending.  What makes that possible: everything slow in the detector's state - the AGC gain, the baseline - only
//             enters through two thresholds (src/decoder.c:785-786), and WHICH row of an extreme fires changes nothing
//             downstream (the countdown ends when the extreme leaves the window, src/decoder.c:741; the AGC sees the extreme's
//             value).  So the lane decides against a BAND of thresholds [s_lo, s_hi] x (rise, min_peak), taken from the signal's
//             local amplitude: a margin above the band fires for sure, one below it cannot, one inside it is a "maybe" that the
//             next rows usually settle (the same extreme, now for sure: the record says from which row on it may have fired).
//             Whatever the band cannot settle is a DOUBT: the list ends there.  The state that is left - the countdown, and the
//             reference's stale window minimum, a function of the samples (rtfe_kernels.hip: stale_ld) - is noted at the
//             sub-segment's first own row.  Output per lane: a fixed slot {count, countdown at the start, doubt row, records}.
//   k_dchain  a lane per chain (burst, distinct parameter set, track), sequential: the AGC schedule of the block decoders
//             (src/decode_gcr.c:843-864, src/decode_pe.c:127-198, src/decode_nrzi.c:196-229), the thresholds, the events.  It
//             runs the LITERAL detector on the samples in HBM from the burst's restart row (window filling, staggered start,
//             deskew FIFO start-up: src/decoder.c:820-861) until, at a sub-segment boundary, its own state equals what the
//             sub-segment's lane noted (countdown, a forced rescan seen) and its thresholds lie inside the sub-segment's band;
//             from there it consumes records - per record: is the band still right, which of the maybe rows fires (the
//             reference's float comparison on the two samples), refine_peak's half-sample code from the neighbour distances,
//             the event, the AGC mirror, the new thresholds - and drops back to the literal detector wherever a list ends in a
//             doubt, a join fails, or the thresholds leave the band (exact either way; the data decides only the speed).
//
// Parameter sets that differ only in what the HOST decoders read (clk_window, clk_alpha, pulse_adj, z1pt ...; src/parmsets.c:77-110)
// are one chain: its events are stored into every such set's region (DevCfg::uset_*).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtfe {

constexpr int kDsSub = 128;                    // rows of a sub-segment
constexpr int kDsJ = 8;                        // sub-segments per tile
constexpr int kDsTile = kDsSub * kDsJ;         // own rows of a tile
constexpr int kDsThreads = 256;
constexpr int kDsMaxMaybe = 15;
constexpr int kDsNoJoin = 0xff, kDsNoDoubt = 0xff;
struct DsHdr { uint8_t count, start_blind, doubt, flags; float s_lo, s_hi; uint32_t pad; };      // 16 bytes in front of a slot's records; [s_lo, s_hi]: the band the lane decided against
// a record (8 bytes): w0 = nf (first row the extreme may have fired at, relative to the sub-segment's first row: 0..127) | nmaybe << 8
//                          (rows nf .. nf + nmaybe - 1 are "maybe", row nf + nmaybe fires for sure) | kind << 12 (1 = bottom)
//                          | ld << 13 (left_distance at row nf: 2 .. W - 1) | dprev << 19 >> ... (see ds_pack)
//                     w1 = val (int16 code of the extreme) | dprev << 16 | dnext << 24 (|val - neighbour| towards "beyond the extreme", 0..255)
struct DsRec { uint32_t w0, w1; };
__device__ __forceinline__ DsRec ds_pack(int nf, int nm, int kind, int ld, int val, int dp, int dn) {
   DsRec r; r.w0 = (uint32_t)nf | ((uint32_t)nm << 8) | ((uint32_t)kind << 12) | ((uint32_t)ld << 13);
   r.w1 = (uint32_t)(uint16_t)val | ((uint32_t)dp << 16) | ((uint32_t)dn << 24); return r; }

struct DsLds { unsigned bits, ldpos, band, total; };
__host__ __device__ inline DsLds ds_lds_layout(int ntrks, int halo_rows, int tile_rows) {
   DsLds L;
   unsigned off = lds_align16((unsigned)ntrks * (unsigned)(halo_rows + tile_rows + 8) * 2u + 16u);
   L.bits = off;  off = lds_align16(off + (unsigned)ntrks * 5u * lds_bstride(tile_rows));       // (one screen at a time: five kinds as run_screens lays them out, three used)
   L.ldpos = off; off = lds_align16(off + (unsigned)ntrks * 2u * lds_ldstride(tile_rows));
   L.band = off;  off = lds_align16(off + (unsigned)kDsJ * RTFE_MAXTRKS * 12u);
   L.total = off;
   return L; }

enum { kDsMiss = 0, kDsMaybe = 1, kDsSure = 2 };

// ------------------------------------------------------------------------------------------------
// k_dseg
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kDsThreads) k_dseg(const DevCfg *__restrict__ cfgp, const int16_t *__restrict__ rows, long long nrows, long long ntiles,
                                                     unsigned char *__restrict__ dead, float2 *__restrict__ band, unsigned char *__restrict__ slots,
                                                     unsigned long long *__restrict__ dbg) {
#ifdef RTFE_CPU_EMUL
   unsigned char *smem = g_dyn_smem;
#else
   extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
   __shared__ DevCfg cfg;
   __shared__ int s_any, s_amp[RTFE_MAXTRKS];
   for (int i = threadIdx.x; i < (int)(sizeof(DevCfg) / 4); i += blockDim.x) reinterpret_cast<int *>(&cfg)[i] = reinterpret_cast<const int *>(cfgp)[i];
   __syncthreads();
   const int ntrks = cfg.ntrks, pad = cfg.ds_pad, T = pad + kDsTile, nu = cfg.nuset;
   const DsLds L = ds_lds_layout(ntrks, cfg.halo_rows, T);
   Tile tl;
   tl.x = reinterpret_cast<int16_t *>(smem); tl.halo = cfg.halo_rows; tl.ldw = 0; tl.colof = cfg.trk_to_head; tl.ntrks = ntrks; tl.skew = cfg.skew;
   tl.bits = smem + L.bits; tl.bstride = (int)lds_bstride(T); tl.ldpos = smem + L.ldpos; tl.ldstride = (int)lds_ldstride(T); tl.fd = nullptr;
   tl.reset = -(1ll << 40);                                           // (the regular deskew regime everywhere: k_dchain joins only behind the start-up rows)
   float2 *s_band = reinterpret_cast<float2 *>(smem + L.band);
   int *s_ampj = reinterpret_cast<int *>(smem + L.band + kDsJ * RTFE_MAXTRKS * 8);
   const float lsb = cfg.lsb_per_volt;
   const int slot_bytes = cfg.ds_slot, cap = cfg.ds_cap;
   const FastDiv fdn(ntrks);
   long long t_load = 0, t_scr = 0, t_walk = 0, tq = 0;
   const bool prof = cfg.debug == 7;
   for (long long g = blockIdx.x; g < ntiles; g += gridDim.x) {
      tl.row0 = g * kDsTile - pad; tl.nrows = T;
      __syncthreads();
      if (prof) tq = clock64();
      if (threadIdx.x < RTFE_MAXTRKS) s_amp[threadIdx.x] = 0;
      load_tile(&cfg, tl, rows, nrows);
      __syncthreads();
      // ---- the band of every (sub-segment, track): from the amplitude of the rows the sub-segment's lanes can see ----
      {  const int wmax = cfg.halo_rows - kScreenHalo;                  // (>= widest window + 1 + max skew)
         for (int i = threadIdx.x; i < kDsJ * ntrks; i += blockDim.x) {
            const int j = fdn.div(i), t = i - j * ntrks;
            const Col yb = tile_col(tl, t, cfg.skew[t]);
            int mx = -40000, mn = 40000;
            #pragma nounroll
            for (int q = j * kDsSub - wmax; q < pad + (j + 1) * kDsSub; ++q) { const int v = yb[q]; mx = max(mx, v); mn = min(mn, v); }
            const int amp = mx - mn;
            atomicMax(&s_amp[t], amp);
            const float av = (float)amp / lsb;                          // volts, peak to peak
            // the AGC makes the thresholds follow the signal: (v_avg_height / 4) / agc_gain ~ (recent peak-to-peak height) / 4 <= amplitude / 4
            float hi = av * cfg.ds_band_hi * 0.25f;
            float lo = hi * cfg.ds_band_lo; if (lo < cfg.ds_sfloor) lo = cfg.ds_sfloor;
            if (hi < lo) hi = lo;
            s_band[j * RTFE_MAXTRKS + t] = make_float2(lo, hi); s_ampj[j * RTFE_MAXTRKS + t] = amp; } }
      __syncthreads();
      if (prof) { const long long t2 = clock64(); t_load += t2 - tq; tq = t2; }
      bool wrote_any = false;
      for (int s = 0; s < cfg.nscreens; ++s) {
         const DevScreen S = cfg.screen[s];
         // nothing in this tile can rise above the screen: no list (k_dchain reads the flag, not the slots)
         bool flat = true;
         for (int t = 0; t < ntrks; ++t) if (s_amp[t] > S.rise_i) flat = false;
         if (flat) { if (threadIdx.x == 0) dead[g * cfg.nscreens + s] = 1; continue; }
         if (threadIdx.x == 0) s_any = 0;
         __syncthreads();
         {  const int hs = kScreenHalo / kStrip, nstrips = T / kStrip + hs, per = nstrips * ntrks;
            int any = 0;
            for (int i = (int)threadIdx.x; i < per; i += blockDim.x) { const int q = fdn.div(i); any |= screen_strip(tl, S, 0, i - q * ntrks, q - hs); }
            if (any) s_any = 1; }
         __syncthreads();
         if (prof) { const long long t2 = clock64(); t_scr += t2 - tq; tq = t2; }
         if (!s_any) { if (threadIdx.x == 0) dead[g * cfg.nscreens + s] = 1; __syncthreads(); continue; }
         if (threadIdx.x == 0) dead[g * cfg.nscreens + s] = 0;
         wrote_any = true;
         // ---- the lanes: (distinct set of this width, sub-segment, track) ----
         int us[RTFE_MAXPARMSETS], nus = 0;
         for (int u = 0; u < nu; ++u) if (cfg.parm[cfg.uset_rep[u]].screen == s) us[nus++] = u;
         const int W = S.W, warm = cfg.ds_warm[s];
         const int tile_lim = (nrows - tl.row0 < (long long)T) ? (int)(nrows - tl.row0) : T;
         for (int task = threadIdx.x; task < nus * kDsJ * ntrks; task += blockDim.x) {
            const int q = fdn.div(task), t = task - q * ntrks, ul = q / kDsJ, j = q - ul * kDsJ;
            const int u = us[ul];
            const DevParm P = cfg.parm[cfg.uset_rep[u]];
            float2 bd = s_band[j * RTFE_MAXTRKS + t];
            const bool amp_on = P.min_peak != 0;
            const u64 *tm = tl.map(0, 0, t), *bm = tl.map(0, 1, t), *am = tl.map(0, 2, t);
            const unsigned char *ldt = tl.ldmap(0, 0, t), *ldb = tl.ldmap(0, 1, t);
            const Col yb = tile_col(tl, t, cfg.skew[t]);
            const int o0 = pad + j * kDsSub, o1 = o0 + kDsSub;
            const int lim = o1 < tile_lim ? o1 : tile_lim;
            // A sub-segment of small signal (a gap, a block's first or last rows): the chain that passes through has thresholds from
            // elsewhere.  If nothing here rises above a level that such thresholds clear, the band is [that level, infinity): no record.
            if ((float)s_ampj[j * RTFE_MAXTRKS + t] < P.rise * cfg.ds_quiet_s * lsb * 2.0f) {
               int mmax = 0;
               #pragma nounroll
               for (int q = o0 - warm; q < lim; ) {
                  const int wd = q >> 6;
                  const u64 c = (tm[wd] | bm[wd]) >> (q & 63);
                  if (!c) { q = (wd + 1) << 6; continue; }
                  q += __ffsll((long long)c) - 1;
                  if (q >= lim) break;
                  const int lo = q - W + 1, vl = yb[lo], vr = yb[q];
                  if ((tm[q >> 6] >> (q & 63)) & 1) mmax = max(mmax, (int)yb[lo + ldt[q] - 1] - max(vl, vr));
                  if ((bm[q >> 6] >> (q & 63)) & 1) mmax = max(mmax, min(vl, vr) - (int)yb[lo + ldb[q] - 1]);      // (the true minimum: the stale one's margin is no larger)
                  ++q; }
               const float lo_s = (float)(mmax + 4) / (P.rise * lsb);
               if (lo_s <= cfg.ds_quiet_s) bd = make_float2(lo_s > cfg.ds_sfloor ? lo_s : cfg.ds_sfloor, 3.0e38f); }
            // margins in int16 codes: >= r_hi passes for every threshold of the band, <= r_lo for none (k_dchain's exact test has a
            // guard band of floor(thr * lsb) - 1 .. + 2 around every threshold; one code more on either side here)
            const float fr = P.rise * bd.y * lsb, fq = P.min_peak * bd.y * lsb;
            const int r_hi = fr > 1.0e9f ? 0x3fffffff : (int)floorf(fr) + 3, r_lo = (int)floorf(P.rise * bd.x * lsb) - 2;
            const int q_hi = fq > 1.0e9f ? 0x3fffffff : (int)floorf(fq) + 3, q_lo = (int)floorf(P.min_peak * bd.x * lsb) - 2;
            auto cls = [&](int m, int v) -> int {
               if (m <= r_lo || (amp_on && v <= q_lo)) return kDsMiss;
               return (m >= r_hi && (!amp_on || v >= q_hi)) ? kDsSure : kDsMaybe; };
            unsigned char *slot = slots + (((size_t)(g * kDsJ + j) * nu + u) * ntrks + t) * (size_t)slot_bytes;
            DsRec *recs = reinterpret_cast<DsRec *>(slot + sizeof(DsHdr));
            int n = o0 - warm, blind_until = n - 1;
            int pk = -1, ppos = 0, pfirst = 0;                            // a pending "maybe": kind, the extreme's row, the first maybe row
            int count = 0, doubt = -1, start_blind = 0;
            bool stop = false;
            #pragma nounroll
            for (int phase = 0; phase < 2 && !stop; ++phase) {
               const int plim = phase == 0 ? (o0 < lim ? o0 : lim) : lim;
               #pragma nounroll
               while (!stop) {
                  if (n <= blind_until) n = blind_until + 1;
                  if (n >= plim) break;
                  const int wd = n >> 6;
                  const u64 c = (tm[wd] | bm[wd]) >> (n & 63);
                  if (!c) { n = (wd + 1) << 6; continue; }
                  n += __ffsll((long long)c) - 1;
                  if (n >= plim) break;
                  const int bit = n & 63, wd2 = n >> 6;
                  const bool ctop = (tm[wd2] >> bit) & 1, cbot = (bm[wd2] >> bit) & 1;
                  const int lo = n - W + 1;
                  const int vl = yb[lo], vr = yb[n];
                  int tcls = kDsMiss, bcls = kDsMiss, tpos = 0, tval = 0, bpos = 0, bval = 0;
                  bool unknown = false;
                  // (the warm-up rows only have to bring the countdown into step - the join checks that they did: a maybe there counts as a hit)
                  if (ctop) { tpos = lo + ldt[n] - 1; tval = yb[tpos]; tcls = cls(tval - max(vl, vr), tval); if (phase == 0 && tcls == kDsMaybe) tcls = kDsSure; }
                  if (tcls != kDsSure && cbot) {
                     const int l = stale_ld(am, ldb, n);                 // the reference's (possibly stale) minimum, from the samples alone
                     if (l == 0) unknown = phase != 0;
                     else { bpos = lo + l - 1; bval = yb[bpos]; bcls = cls(min(vl, vr) - bval, -bval);
                            if (phase == 0 && bcls == kDsMaybe) bcls = kDsSure;
                            if (bcls != kDsMiss && (bpos <= lo || bpos >= n)) { unknown = phase != 0; bcls = kDsMiss; } } }      // (refine_peak's assert: the literal detector flags it)
                  int fire = -1, fpos = 0, fval = 0, dbt = -1;
                  if (unknown) dbt = pk >= 0 ? pfirst : n;
                  else if (tcls == kDsSure) { fire = 0; fpos = tpos; fval = tval; }
                  else if (tcls == kDsMaybe) {
                     if (bcls != kDsMiss || (pk >= 0 && (pk != 0 || ppos != tpos))) dbt = pk >= 0 ? pfirst : n;
                     else if (pk < 0) { pk = 0; ppos = tpos; pfirst = n; } }
                  else if (bcls == kDsSure) { fire = 1; fpos = bpos; fval = bval; }
                  else if (bcls == kDsMaybe) {
                     if (pk >= 0 && (pk != 1 || ppos != bpos)) dbt = pfirst;
                     else if (pk < 0) { pk = 1; ppos = bpos; pfirst = n; } }
                  if (fire >= 0) {
                     const int nf = pk >= 0 ? pfirst : n;
                     if (pk >= 0 && (pk != fire || ppos != fpos)) dbt = pfirst;
                     else if (n - nf > kDsMaxMaybe) dbt = nf;
                     else if (phase == 1 && count >= cap) dbt = nf;
                     else {
                        if (phase == 1) {
                           const int pv = yb[fpos - 1], nx = yb[fpos + 1];
                           int dp = fire == 0 ? fval - pv : pv - fval, dn = fire == 0 ? fval - nx : nx - fval;
                           dp = dp < 0 ? 0 : (dp > 255 ? 255 : dp); dn = dn < 0 ? 0 : (dn > 255 ? 255 : dn);
                           recs[count++] = ds_pack(nf - o0, n - nf, fire, fpos - nf + W, fval, dp, dn); }
                        blind_until = fpos + W; pk = -1; } }
#ifdef RTFE_CPU_EMUL
                  if (dbt >= 0 && getenv("RTFE_DS_TRACE")) fprintf(stderr, "doubt tile %lld s %d u %d trk %d j %d phase %d row %lld (n %d): unknown %d tcls %d bcls %d pk %d fire %d count %d r %d..%d q %d..%d tm %d bm %d\n", g, s, u, t, j, phase, tl.row0 + dbt, n, (int)unknown, tcls, bcls, pk, fire, count, r_lo, r_hi, q_lo, q_hi,
                                                                    ctop ? tval - max(vl, vr) : -1, (cbot && !unknown) ? min(vl, vr) - bval : -1);
#endif
                  if (dbt >= 0) { doubt = dbt; stop = true; break; }
                  ++n; }
               if (phase == 0) {
                  start_blind = blind_until + 1 - o0; if (start_blind < 0) start_blind = 0;
                  if (stop || pk >= 0 || start_blind >= kDsNoJoin) { start_blind = kDsNoJoin; stop = true; doubt = -1; } } }
            if (!stop && pk >= 0) doubt = pfirst;                        // a maybe that the sub-segment's rows did not settle
            DsHdr h; h.count = (uint8_t)count; h.start_blind = (uint8_t)start_blind;
            h.doubt = (uint8_t)((doubt >= o0 && start_blind != kDsNoJoin) ? doubt - o0 : kDsNoDoubt); h.flags = 0; h.pad = 0; h.s_lo = bd.x; h.s_hi = bd.y;
            if (doubt >= 0 && doubt < o0) h.start_blind = kDsNoJoin;      // (cannot happen behind phase 0; belt and braces)
            *reinterpret_cast<DsHdr *>(slot) = h; }
         __syncthreads();
         if (prof) { const long long t2 = clock64(); t_walk += t2 - tq; tq = t2; } }
      if (wrote_any)
         for (int i = threadIdx.x; i < kDsJ * ntrks; i += blockDim.x) { const int j = fdn.div(i), t = i - j * ntrks; band[(size_t)(g * kDsJ + j) * ntrks + t] = s_band[j * RTFE_MAXTRKS + t]; } }
   if (prof && threadIdx.x == 0) { atomicAdd(&dbg[0], (unsigned long long)t_load); atomicAdd(&dbg[1], (unsigned long long)t_scr); atomicAdd(&dbg[2], (unsigned long long)t_walk); } }

// ------------------------------------------------------------------------------------------------
// k_dchain
// ------------------------------------------------------------------------------------------------
struct DcRows {                // the detector's input straight from HBM: row n of track t after -invert and the deskew FIFO (src/decoder.c:820-830)
   const int16_t *col; int P, sgn, d; long long reset;
   __device__ __forceinline__ int operator()(long long n) const {
      const long long m = (n - reset < d) ? n : n - d;
      return sgn * (int)col[m * P]; } };

__global__ void __launch_bounds__(64) k_dchain(const DevCfg *__restrict__ cfgp, const int16_t *__restrict__ rows, long long nrows, long long row_base,
                                               const rtfe_burst *__restrict__ bursts, BurstScratch *__restrict__ scratch, BurstCtl *__restrict__ ctl,
                                               uint32_t *__restrict__ counts, rtfe_event *__restrict__ events,
                                               const unsigned char *__restrict__ dead, const float2 *__restrict__ band, const unsigned char *__restrict__ slots, long long ntiles) {
   __shared__ float s_heights[64 * 10];
   const DevCfg &cfg = *cfgp;
   const int ntrks = cfg.ntrks, nu = cfg.nuset, nwalk = nu * ntrks;
   const int lane = threadIdx.x;
   const float mv = cfg.maxvolts, lsb = cfg.lsb_per_volt;
   const int nchains = scratch->nbursts * nwalk;
   float *heights = s_heights + lane * 10;
   const int slot_bytes = cfg.ds_slot;
   for (int cbase = blockIdx.x * 64; cbase < nchains; cbase += gridDim.x * 64) {
      const int ci = cbase + lane < nchains ? cbase + lane : nchains - 1;
      const int b = ci / nwalk;
      const int wi = ci - b * nwalk, u = wi / ntrks, trk = wi - u * ntrks;
      const bool active = cbase + lane < nchains && ctl[b].status == kBurstReady;
      const rtfe_burst B = bursts[b];
      const int pidx = cfg.uset_rep[u];
      const unsigned pmask = cfg.uset_mask[u];
      const DevParm P = cfg.parm[pidx];
      const int cmode = cfg.mode, agc_off = cfg.agc_off;
      const int W = P.W, sc = P.screen;
      const bool amp_on = P.min_peak != 0;
      const long long reset = ctl[b].reset;
      const long long stop = chain_stop(cfg, bursts, ctl, b, scratch->nbursts_total, nrows);
      DcRows y; y.col = rows + cfg.trk_to_head[trk]; y.P = ntrks; y.sgn = cfg.invert ? -1 : 1; y.d = cfg.skew[trk]; y.reset = reset;
      const long long start = reset + trk;
      const long long fast_from = reset + W + max(trk, y.d) + 1;
      Walker w = {};
      w.agc_gain = 1.0f; w.v_avg_height = 4.0f;
      update_thresholds(w, P, lsb);
      for (int i = 0; i < 10; ++i) heights[i] = 0;
      const unsigned int cap = B.event_cap;
      rtfe_event *evb = events + B.event_base;
      // ---- one detection: refine_peak's code, the event (into the region of every parameter set this chain stands for), the AGC mirror ----
      bool done = !active, failed = false, dead_chain = false;
      auto fire = [&](long long n, int ld, int val, bool is_top, int iprev, int inext) {
         const int adjcode = refine_code(&cfg, val, iprev, inext, w.agc_gain, is_top);
         double t_peak = 0;
         if (cmode == RTFE_PE) {
            const float adj = adjcode == 1 ? -0.5f : (adjcode == 2 ? 0.5f : 0.0f);
            t_peak = time_of(&cfg, row_base + n) - ((float)(W - ld) - adj) * cfg.sample_deltat; }
         const float vp = volt(val, mv);
         if (w.nevents >= cap) w.flags |= RTFE_F_EVENT_OVERFLOW;
         else {
            rtfe_event e;
            e.sample = (uint32_t)(n - reset); e.v_peak = (cfg.invert && vp == 0.0f) ? -0.0f : vp; e.agc_gain = w.agc_gain;
            e.trk = (uint8_t)trk; e.flags = (uint8_t)((is_top ? 0 : 1) | (adjcode << 1)); e.left_distance = (uint8_t)ld;
            for (unsigned m = pmask; m; m &= m - 1) { const int p = __ffs((int)m) - 1; e.parmset = (uint8_t)p; evb[(size_t)(p * ntrks + trk) * cap + w.nevents] = e; } }
         if (is_top) w.v_top = vp; else w.v_bot = vp;
         ++w.nevents;
         agc_after_peak_m(w, cmode, agc_off, P, heights, is_top, t_peak);
         if (!(w.agc_gain > 0)) {                                        // src/decoder.c:782, at the first row behind the countdown (agc_fatal)
            w.flags |= RTFE_F_AGC_FATAL;
            if (w.nevents < cap) {
               rtfe_event e = {};
               e.sample = (uint32_t)(n + ld + 1 - reset); e.trk = (uint8_t)trk; e.flags = RTFE_EV_FATAL;
               for (unsigned m = pmask; m; m &= m - 1) { const int p = __ffs((int)m) - 1; e.parmset = (uint8_t)p; evb[(size_t)(p * ntrks + trk) * cap + w.nevents] = e; } }
            else w.flags |= RTFE_F_EVENT_OVERFLOW;
            ++w.nevents; dead_chain = true; return; }
         update_thresholds(w, P, lsb); };
      // ---- the literal detector (src/decoder.c:751-810 on the samples in HBM): state after the last row it processed ----
      int lmx = 0, lmn = 0, lcd = 0;
      bool synced = false;                                               // a forced rescan of a full, regular window was seen: from there the stale minimum is what k_dseg derives
      auto rescan = [&](long long lo, long long hi) { int mx = -0x7fffffff, mn = 0x7fffffff; for (long long j = lo; j <= hi; ++j) { const int v = y(j); mx = max(mx, v); mn = min(mn, v); } lmx = mx; lmn = mn; };
      auto lit_step = [&](long long n) {
         if (n < start) return;
         if (n == start) { const int v = y(n); lmx = v; lmn = v; lcd = 0; w.t_lastpeak = time_of(&cfg, row_base + n); return; }      // src/decoder.c:855-861
         const bool popped = n - start + 1 > W;
         const long long lo = popped ? n - W + 1 : start;
         const int vnow = y(n);
         const int old_left = popped ? y(n - W) : 0;
         if (vnow > lmx) lmx = vnow;
         if (old_left == lmx || old_left == lmn) { if (popped && old_left == lmx && n >= fast_from) synced = true; rescan(lo, n); }
         if (lcd) { --lcd; return; }
         const int vli = y(lo);
         // the reference's comparisons (src/decoder.c:788-805): on the codes where clear, in floats inside the guard band
         bool top = above_by(lmx, vli, w.rise, w.rise_lo, w.rise_hi, mv) && above_by(lmx, vnow, w.rise, w.rise_lo, w.rise_hi, mv)
                    && (w.reqmin == 0 || lmx >= w.min_hi || (lmx > w.min_lo && volt(lmx, mv) > w.reqmin));
         bool bot = !top && below_by(lmn, vli, w.rise, w.rise_lo, w.rise_hi, mv) && below_by(lmn, vnow, w.rise, w.rise_lo, w.rise_hi, mv)
                    && (w.reqmin == 0 || -lmn >= w.min_hi || (-lmn > w.min_lo && volt(lmn, mv) < -w.reqmin));
         if (top || bot) {
            const int val = top ? lmx : lmn;
            long long p = lo;
            while (p <= n && y(p) != val) ++p;
            if (p > n || p == lo || p == n) { w.flags |= RTFE_F_DETECTOR_FATAL; return; }      // src/decoder.c:709-710,748
            const int ld = (int)(p - lo) + 1;
            fire(n, ld, val, top, y(p - 1), y(p + 1));
            lcd = ld; } };
      // the literal state behind row n0 - 1 from the samples alone: the last forced rescan in front of it (the sample leaving the window
      // is its maximum), the bookkeeping of src/decoder.c:757-775 from there
      auto resync = [&](long long n0, long long blind_until) -> bool {
         long long lowlim = n0 - 1 - (4 * W + 64);
         if (lowlim < fast_from) lowlim = fast_from;
         long long h = n0 - 1;
         for (; h >= lowlim; --h) {
            const int v = y(h - W);
            bool dom = true;
            for (int k = 1; k <= W; ++k) if (y(h - W + k) > v) { dom = false; break; }
            if (dom) break; }
         if (h < lowlim) return false;
         rescan(h - W + 1, h);
         for (long long r = h + 1; r <= n0 - 1; ++r) {
            const int vnow = y(r), old_left = y(r - W);
            if (vnow > lmx) lmx = vnow;
            if (old_left == lmx || old_left == lmn) rescan(r - W + 1, r); }
         const long long c = blind_until - (n0 - 1);
         lcd = c > 0 ? (c > 0x3fffffff ? 0x3fffffff : (int)c) : 0;
         synced = true;
         return true; };
      auto in_band = [&](const float2 bd) -> bool {
         return w.rise >= P.rise * bd.x && w.rise <= P.rise * bd.y && (!amp_on || (w.reqmin >= P.min_peak * bd.x && w.reqmin <= P.min_peak * bd.y)); };
      long long cur = reset;                                              // next row to process
      long long blind_until = -1;                                         // (record mode)
      bool lit = true;
      unsigned n_lit_rows = 0, n_rec_ev = 0, n_doubt = 0, n_nojoin = 0;
      // (every lane of the wave goes through the same rounds: the emulator's ballot needs all of them)
      #pragma nounroll
      while (__ballot(!done) != 0ull) {
         if (done) continue;
         if (dead_chain || cur >= stop) { done = true; continue; }
         const bool at_bnd = (cur % kDsSub) == 0;
         const long long seg = cur / kDsSub, tile = seg / kDsJ;
         bool join = false;
         const DsHdr *hp = nullptr;
         float2 bd = make_float2(cfg.ds_sfloor, 3.0e38f);
         bool tile_dead = false;
         if (at_bnd && cur >= fast_from && (synced || !lit) && tile < ntiles) {
            tile_dead = dead[tile * cfg.nscreens + sc] != 0;
            if (tile_dead) join = in_band(bd);                             // nothing can rise above the screen: no countdown to agree on
            else {
               hp = reinterpret_cast<const DsHdr *>(slots + (((size_t)seg * nu + u) * ntrks + trk) * (size_t)slot_bytes);
               bd = make_float2(hp->s_lo, hp->s_hi);
               long long mb = lit ? (long long)lcd : blind_until - cur + 1;
               if (mb < 0) mb = 0;
               join = hp->start_blind != kDsNoJoin && mb == (long long)hp->start_blind && in_band(bd);
               if (!join) ++n_nojoin;
#ifdef RTFE_CPU_EMUL
               if (!join && getenv("RTFE_DS_TRACE")) fprintf(stderr, "nojoin b %d u %d trk %d row %lld lit %d: start_blind %d mine %lld band %.3f..%.3f rise %.4f (%.4f..%.4f)\n", b, u, trk, cur, (int)lit, (int)hp->start_blind, mb, bd.x, bd.y, w.rise, P.rise * bd.x, P.rise * bd.y);
#endif
               } }
         if (join) {
            if (lit) { blind_until = cur - 1 + lcd; lit = false; }
            const long long r0 = cur;
            long long next = r0 + kDsSub;                                 // where the chain goes on
            bool to_lit = false;
            if (!tile_dead) {
               const DsHdr h = *hp;
               const DsRec *recs = reinterpret_cast<const DsRec *>(reinterpret_cast<const unsigned char *>(hp) + sizeof(DsHdr));
               #pragma nounroll
               for (int k = 0; k < (int)h.count; ++k) {
                  const DsRec rc = recs[k];
                  const int nfr = (int)(rc.w0 & 0xff), nm = (int)((rc.w0 >> 8) & 0xf), kind = (int)((rc.w0 >> 12) & 1), ld0 = (int)((rc.w0 >> 13) & 0x3f);
                  const int val = (int)(int16_t)(rc.w1 & 0xffff), dp = (int)((rc.w1 >> 16) & 0xff), dn = (int)(rc.w1 >> 24);
                  const long long nf = r0 + nfr;
                  if (nf >= stop) { next = stop; break; }
                  // which of the maybe rows fires: the reference's comparison on the two samples at the window's edges
                  long long n = nf + nm;
                  for (int m = 0; m < nm; ++m) {
                     const long long q = nf + m;
                     const int vl = y(q - W + 1), vr = y(q);
                     const bool hit = kind == 0
                        ? (above_by(val, vl, w.rise, w.rise_lo, w.rise_hi, mv) && above_by(val, vr, w.rise, w.rise_lo, w.rise_hi, mv)
                           && (w.reqmin == 0 || val >= w.min_hi || (val > w.min_lo && volt(val, mv) > w.reqmin)))
                        : (below_by(val, vl, w.rise, w.rise_lo, w.rise_hi, mv) && below_by(val, vr, w.rise, w.rise_lo, w.rise_hi, mv)
                           && (w.reqmin == 0 || -val >= w.min_hi || (-val > w.min_lo && volt(val, mv) < -w.reqmin)));
                     if (hit) { n = q; break; } }
                  if (n >= stop) { next = stop; break; }
                  const int ld = ld0 - (int)(n - nf);
                  const long long p = n - W + ld;
                  // refine_peak's neighbours from their distances; a clamped distance only matters if the threshold reaches it
                  int iprev = kind == 0 ? val - dp : val + dp, inext = kind == 0 ? val - dn : val + dn;
                  const int ti = (int)floorf(0.005f / w.agc_gain * lsb);
                  if (ti + 2 >= 255 || ti < 2) { iprev = y(p - 1); inext = y(p + 1); }
                  fire(n, ld, val, kind == 0, iprev, inext);
                  ++n_rec_ev;
                  blind_until = n + ld;
                  if (dead_chain) break;
                  if (!in_band(bd)) { next = n + 1; to_lit = true; break; } }   // the thresholds left the band: what this list says about the rows behind n is not proven
               if (!dead_chain && !to_lit && next == r0 + kDsSub && h.doubt != kDsNoDoubt) { next = r0 + h.doubt; to_lit = true; ++n_doubt; } }
            cur = next;
            if (to_lit && cur < stop) {
               if (!resync(cur, blind_until)) { failed = true; done = true; }
               lit = true; }
            continue; }
         // ---- literal rows up to the next sub-segment boundary ----
         if (!lit) { if (!resync(cur, blind_until)) { failed = true; done = true; continue; } lit = true; }
         long long end = (seg + 1) * kDsSub;
         if (end > stop) end = stop;
         #pragma nounroll
         for (long long n = cur; n < end && !dead_chain; ++n) lit_step(n);
         n_lit_rows += (unsigned)(end - cur);
         cur = end; }
      // ---- publish ----
      if (!active) continue;
      if (failed) atomicExch(&ctl[b].status, (int)kBurstNeedsFull);
      const unsigned int ne = w.nevents < cap ? w.nevents : cap;
      for (unsigned m = pmask; m; m &= m - 1) { const int p = __ffs((int)m) - 1; counts[((size_t)b * cfg.nparm + p) * ntrks + trk] = ne; }
      if (w.flags) atomicOr(&ctl[b].bflags, w.flags);
      if (n_lit_rows) atomicAdd(&scratch->dbg[0], (unsigned long long)n_lit_rows);
      if (n_rec_ev) atomicAdd(&scratch->dbg[1], (unsigned long long)n_rec_ev);
      if (n_doubt) atomicAdd(&scratch->why[0], (unsigned long long)n_doubt);
      if (n_nojoin) atomicAdd(&scratch->why[1], (unsigned long long)n_nojoin);
      if (failed) atomicAdd(&scratch->why[2], 1ull); } }

}  // namespace rtfe
