// rtfe_api.hip — host side of the C ABI declared in include/rt_frontend.h (librtfe.so).
// Validates the configuration, derives window widths / thresholds exactly as the reference does
// (src/readtape.c:1402,1455-1457; src/decoder.c:448-449) and launches the kernels of a scan.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "rt_frontend.h"
#include "rtfe_device.h"

#include "rtfe_sift.hip"      // single translation unit: kernels + host API
#include "rtfe_kernels.hip"
#include "rtfe_zeros.hip"
#include "rtfe_ww.hip"
#include "rtfe_gain.hip"
#include "rtfe_dense.hip"
#include "rtfe_pack.hip"

namespace rtfe {
__global__ void k_setup_exact(rtfe_burst *burst, BurstScratch *scratch, long long reset_row, long long end_row,
                              unsigned long long cap) {
   rtfe_burst b = {};
   b.zone_first = reset_row; b.zone_end = reset_row; b.reset_sample = reset_row; b.safe_last = reset_row;
   b.end_sample = end_row; b.event_base = 0; b.event_cap = (uint32_t)cap; b.flags = RTFE_F_EXACT_START;
   *burst = b;
   scratch->nbursts = 1; scratch->nbursts_total = 1; scratch->queue = 0; }
}  // namespace rtfe

using namespace rtfe;

constexpr int kGraphCache = 8;
struct rtfe_handle {
   rtfe_config cfg;
   DevCfg dev;
   DevCfg *d_dev;
   int lds_bytes;
   int num_cus;
   int timing;
   hipEvent_t (*ev0)[12], (*ev1)[12];    // timing: a ring of kTimingRing sets of start / stop events, a set per scan (on the stream it ran on)
   int ev_next, ev_pending;             // the set the next scan records into; sets recorded since rtfe_kernel_ms last looked
   int zeros_kernel;                   // -zeros scans run k_zeros (RTFE_ZEROS_KERNEL=0: k_decode's zero-crossing mode, kept for tests)
   hipStream_t side;                   // the peak path's quiet map -> bursts -> restart rows beside its lists -> streams (both only need k_sift): RTFE_OVERLAP=0 keeps them in line
   hipEvent_t ev_fork, ev_join, ev_fork2, ev_join2;
   int overlap;
   int bursts_wpr;                     // RTFE_BURSTS_WPR: words of the quiet map per round of the zone search (tests: many rounds on a short tape); 0 = 4096
   int sfs_occ[kMaxScreens];           // k_sift_s: workgroups of a screen's instantiation a CU holds at once (the occupancy API, at create)
   long long work_cap;      // RTFE_WORK_CAP
   int ds_order, dchain_wgs, prep_wgs, dense_stop, dseg_wgs, dseg_threads;      // RTFE_DS_ORDER (0: chains in burst order), RTFE_DCHAIN_WGS / RTFE_PREP_WGS / RTFE_DSEG_WGS (workgroups per CU), RTFE_DENSE_STOP (debugging): read once, at create
   // rtfe_set_graphs: a scan's launches (about twenty, on two streams) captured once per set of arguments into a HIP graph and replayed - what a scan of the same
   // buffers costs the host, and the gaps between its kernels on the device, shrink to one launch
   int graphs;
   struct GraphEnt { unsigned long long key[13]; hipGraphExec_t exec; unsigned long long stamp; } gcache[kGraphCache];
   unsigned long long gstamp;
};

static thread_local char g_err[512] = "";

static void drop_graphs(rtfe_handle *h) {
   for (int i = 0; i < kGraphCache; ++i) if (h->gcache[i].exec) { (void)hipGraphExecDestroy(h->gcache[i].exec); h->gcache[i].exec = nullptr; } }


// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel on the CURRENT DEVICE, not of a handle: a second handle with a smaller LDS
// layout must not lower what a first one's launches need, a handle made on a second device must set it there too, and handles are made
// from several threads (ingest.py) - a table of the largest size asked for so far per (device, kernel), behind a mutex.
static void raise_dynamic_lds(const void *kernel, int bytes) {
   static std::mutex mu;
   static struct { const void *k; int dev, b; } seen[256];
   static int nseen = 0;
   int dev = 0;
   (void)hipGetDevice(&dev);
   std::lock_guard<std::mutex> lock(mu);
   for (int i = 0; i < nseen; ++i) if (seen[i].k == kernel && seen[i].dev == dev) {
      if (bytes > seen[i].b) seen[i].b = bytes;
      (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, seen[i].b); return; }
   if (nseen < 256) { seen[nseen].k = kernel; seen[nseen].dev = dev; seen[nseen].b = bytes; ++nseen; }
   (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
static int fail(int code, const char *fmt, ...) {
   va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
   return code; }

// the k_sift instantiation for a window width, a workgroup size and the vectors a thread prefetches
typedef void (*sf_kernel_t)(const DevCfg *, const int16_t *, long long, long long, uint16_t *, PeakDir *, unsigned char *, SfHard *, int, int *, unsigned long long *);
typedef void (*sfs_kernel_t)(const SfArgs);
constexpr int kSfWps = 5;          // waves per SIMD k_sift's register allocation is held to: four 5-wave workgroups per CU
constexpr int kSfsWps = 6;         // ... and k_sift_s': six four-wave workgroups per CU, 80 registers (round 6: 21.5 KB of LDS a workgroup - no staging slots - would let a seventh reside; at
                                   // 72 registers a prefetched vector spills, and the seventh workgroup is worth under 1 %: measured 0.6376 / 0.6428 ms)
template <int WM, int MAXT> static sf_kernel_t sf_kernel_nv(int nv) { return nv <= 4 ? k_sift<WM, MAXT, 4, kSfWps> : k_sift<WM, MAXT, 6, kSfWps>; }
template <int WM> static sf_kernel_t sf_kernel_t2(int threads, int nv) { return threads <= 320 ? sf_kernel_nv<WM, 320>(nv) : sf_kernel_nv<WM, 640>(nv); }
static sf_kernel_t sf_kernel(int wmax, int threads, int nv) {
   return wmax <= 18 ? sf_kernel_t2<18>(threads, nv) : sf_kernel_t2<50>(threads, nv); }
// one screen of a width and a track count the lean kernel is built for, a sure level that fits 16 bits: k_sift_s
template <int NT, bool PL> static sfs_kernel_t sfs_kernel_w(int w) {
   switch (w) {
      case 6:  return k_sift_s<6, NT, kSfsWps, PL>;
      case 7:  return k_sift_s<7, NT, kSfsWps, PL>;
      case 8:  return k_sift_s<8, NT, kSfsWps, PL>;
      case 9:  return k_sift_s<9, NT, kSfsWps, PL>;
      case 10: return k_sift_s<10, NT, kSfsWps, PL>;
      case 11: return k_sift_s<11, NT, kSfsWps, PL>;
      case 12: return k_sift_s<12, NT, kSfsWps, PL>;
      case 13: return k_sift_s<13, NT, kSfsWps, PL>;
      case 14: return k_sift_s<14, NT, kSfsWps, PL>;
      case 15: return k_sift_s<15, NT, kSfsWps, PL>;
      case 16: return k_sift_s<16, NT, kSfsWps, PL>;
      case 17: return k_sift_s<17, NT, kSfsWps, PL>;
      default: return nullptr; } }
// (every window width the packed derivation holds - 6 .. 17 samples - for nine and seven tracks: 800 / 556 BPI NRZI at 781 kHz are 13 / 19,
//  at half that rate 6 / 9; wider windows, several widths and other track counts take the general kernel: 2.1 instead of 0.93 ms on C2)
static sfs_kernel_t sfs_kernel(int w, int ntrks, bool plain) {
   if (plain) return ntrks == 9 ? sfs_kernel_w<9, true>(w) : (ntrks == 7 ? sfs_kernel_w<7, true>(w) : nullptr);
   return ntrks == 9 ? sfs_kernel_w<9, false>(w) : (ntrks == 7 ? sfs_kernel_w<7, false>(w) : nullptr); }
// k_dseg with the track count at compile time for the usual tapes
typedef void (*ds_kernel_t)(const DevCfg *, const int16_t *, long long, long long, unsigned char *, unsigned char *, unsigned char *, unsigned long long *);
static ds_kernel_t ds_kernel(int ntrks) { return getenv("RTFE_DSEG_GENERIC") ? k_dseg<0> : (ntrks == 9 ? k_dseg<9> : (ntrks == 7 ? k_dseg<7> : k_dseg<0>)); }
static int sf_wmax(const DevCfg &d) { int w = 0; for (int s = 0; s < d.nscreens; ++s) if (d.screen[s].W > w) w = d.screen[s].W; return w; }
static int sf_threads(const DevCfg &d) { return 64 * ((d.ntrks + 1) / 2); }
static int sf_nvec(const DevCfg &d) { return (d.pk_hl + kSfTile + d.pk_hr) * d.ntrks / 8; }
static int sf_nv(const DevCfg &d) { return (sf_nvec(d) + sf_threads(d) - 1) / sf_threads(d); }
// (several window widths - a parameter sweep -: a launch of the lean kernel per screen, if every screen has one: NRZI -m is three widths, 1.5 ms each where the
//  general kernel's loop over the screens took 15.8)
static sfs_kernel_t sf_special_sc(const DevCfg &d, int sc) { return (d.screen[sc].sure_i <= 32767 && !getenv("RTFE_SIFT_GENERIC")) ? sfs_kernel(d.screen[sc].W, d.ntrks, d.pk_plain != 0) : nullptr; }
static sfs_kernel_t sf_special(const DevCfg &d) {
   if (d.nscreens > 1 && getenv("RTFE_SIFT_PER_SCREEN") && atoi(getenv("RTFE_SIFT_PER_SCREEN")) == 0) return nullptr;
   for (int sc = 0; sc < d.nscreens; ++sc) if (!sf_special_sc(d, sc)) return nullptr;
   return d.nscreens >= 1 ? sf_special_sc(d, 0) : nullptr; }
// k_sift_s has a geometry of its own: a wave per pair of heads with an odd last head's tile split among them, only the window's rows in front of a tile
static int sfs_threads(const DevCfg &d) { return 64 * sfs_waves(d.ntrks); }
static int sfs_lds_sc(const DevCfg &d, int sc) { return (int)sfs_lds_layout(d.ntrks, d.screen[sc].W, d.pk_wave_cap, d.pk_slot).total + 64; }
static int sfs_lds(const DevCfg &d) { int m = 0; for (int sc = 0; sc < d.nscreens; ++sc) { const int v = sfs_lds_sc(d, sc); if (v > m) m = v; } return m; }

extern "C" int rtfe_abi_version(void) { return RTFE_ABI_VERSION; }
extern "C" const char *rtfe_last_error(void) { return g_err; }

// The timed spans of one rtfe_scan (rtfe_kernel_ms), in launch order.  The peak path (NRZI peak detection) runs
//   k_sift | k_prep [k_sift_hard, k_pscan1/2, k_prep] beside k_bursts [k_qpack, k_bursts, k_zones: on a stream of the handle's own] | k_gain [the chains' heads] | k_gain_s [k_gain_seg, k_gain_join] |
//   k_gain_tail | k_emit [k_emit_seg, k_emit, k_publish] | k_decode [the bursts the chains gave up, on the samples]
// the sample path (PE, GCR, differentiated peaks, density detection, parameter-set sweeps with too many widths) k_quiet | k_bursts | k_decode,
// -zeros k_quiet | k_bursts | k_zeros.  A span a scan does not run reads 0.
// the dense sample path (PE, GCR peak detection): k_dseg [quiet map folded in] | k_bursts [k_bursts, k_zones] | k_dchain [k_dorder, k_dchain, k_publish] | k_decode [what the chains gave up]
static const char *KNAMES[] = {"k_quiet", "k_sift", "k_prep", "k_bursts", "k_gain", "k_gain_s", "k_gain_tail", "k_emit", "k_decode", "k_zeros", "k_dseg", "k_dchain"};
enum { kTQuiet, kTSift, kTPrep, kTBursts, kTGain, kTGainS, kTGainTail, kTEmit, kTDecode, kTZeros, kTDseg, kTDchain };
constexpr int kNumKernels = 12;
static_assert(kNumKernels <= 12, "rtfe_handle::ev0 / ev1 hold 12 events a set");
extern "C" int rtfe_kernel_count(void) { return kNumKernels; }
extern "C" const char *rtfe_kernel_name(int i) { return (i >= 0 && i < kNumKernels) ? KNAMES[i] : ""; }

static int create_impl(const rtfe_config *c, rtfe_handle **out, int tile_override);
static int sample_path_workgroups_per_cu(const rtfe_handle *h) {     // as rtfe_scan sizes k_decode's grid
   const int nwalk = h->dev.nparm * h->dev.ntrks;
   const int threads = nwalk <= 64 ? 64 : (nwalk <= 128 ? 128 : 256);
   int per_cu = (160 * 1024) / (h->lds_bytes + 1024);
   const int wave_lim = 8 / (threads / 64);
   return per_cu > wave_lim ? wave_lim : (per_cu < 1 ? 1 : per_cu); }

extern "C" int rtfe_create(const rtfe_config *c, rtfe_handle **out) {
   if (!c || !out) return fail(-1, "null argument");
   rtfe_handle *h = nullptr;
   int rc = create_impl(c, &h, 0);
   if (rc != 0) return rc;
   // The sample path (PE, GCR, parameter-set sweeps with several window widths) keeps a tile, its screen maps and its walkers in
   // LDS: with 512-row tiles a sweep's workgroup can be the only one on its CU.  Shorter tiles that let more workgroups reside win
   // (C4: 98 KB -> 75 KB, one -> two per CU, +35 %); the peak path and -zeros have their own tile sizes.
   // (only then: where several workgroups already share a CU the longer tile is better - GCR, one set: 18.4 vs 22.9 ms per 5e7 rows)
   if (!getenv("RTFE_TILE_ROWS") && !h->dev.find_zeros && h->dev.mode != RTFE_WW && sample_path_workgroups_per_cu(h) == 1) {
      rtfe_handle *h2 = nullptr;
      if (create_impl(c, &h2, kMarginRows) == 0) {
         if (sample_path_workgroups_per_cu(h2) > sample_path_workgroups_per_cu(h)) { rtfe_destroy(h); h = h2; }
         else rtfe_destroy(h2); } }
   *out = h;
   return 0; }

static int create_impl(const rtfe_config *c, rtfe_handle **out, int tile_override) {
   if (c->ntrks < 1 || c->ntrks > RTFE_MAXTRKS) return fail(-2, "ntrks %d out of range", c->ntrks);
   if (c->mode != RTFE_NRZI && c->mode != RTFE_PE && c->mode != RTFE_GCR && c->mode != RTFE_WW)
      return fail(-3, "mode %d not known", c->mode);
   if (c->mode == RTFE_WW && (c->nparmsets != 1 || c->find_zeros || c->differentiate))
      return fail(-3, "Whirlwind: one parameter set, peak detection on the undifferentiated signal (rtfe_ww_scan)");
   if (c->nparmsets < 1 || c->nparmsets > RTFE_MAXPARMSETS) return fail(-5, "nparmsets %d out of range", c->nparmsets);
   if (c->nparmsets * c->ntrks > kDecodeThreads) return fail(-6, "nparmsets*ntrks > %d", kDecodeThreads);
   if (!(c->bpi >= 0) || !(c->ips > 0) || c->tdelta_ns <= 0 || !(c->maxvolts > 0)) return fail(-7, "ips, tdelta_ns and maxvolts must be positive, bpi >= 0");
   // bpi == 0: the density is unknown - the front end of the reference's density pre-pass (src/readtape.c:1656-1672):
   // window of 8 samples, no AGC / baseline feedback (no decoder runs).  Sizing heuristics then assume 12 samples per bit.
   const bool density_mode = c->bpi == 0;
   if (density_mode && (c->find_zeros || c->mode != RTFE_NRZI)) return fail(-7, "bpi = 0 (density detection) is built for the NRZI peak detector only");
   const float bpi_s = density_mode ? 1.0f / (c->ips * ((float)c->tdelta_ns / 1e9f) * 12.0f) : c->bpi;
   rtfe_handle *h = new rtfe_handle();
   h->cfg = *c;
   h->timing = 0; h->ev0 = nullptr; h->ev1 = nullptr; h->ev_next = 0; h->ev_pending = 0;
   DevCfg &d = h->dev;
   memset(&d, 0, sizeof d);
   d.mode = c->mode; d.ntrks = c->ntrks; d.invert = c->invert != 0; d.nparm = c->nparmsets;
   d.find_zeros = c->find_zeros != 0;
   d.differentiate = c->differentiate != 0;
   d.agc_off = density_mode;
   bool seen[RTFE_MAXTRKS] = {false};
   for (int i = 0; i < c->ntrks; ++i) {
      int t = c->head_to_trk[i];
      if (t < 0 || t >= c->ntrks || seen[t]) { delete h; return fail(-8, "head_to_trk is not a permutation"); }
      seen[t] = true; d.head_to_trk[i] = t; d.trk_to_head[t] = i;
      int s = c->skew_delaycnt[i];
      if (s < 0 || s > 50) { delete h; return fail(-9, "skew %d out of range 0..50 (MAXSKEWSAMP)", s); }
      d.skew[i] = s; if (s > d.maxskew) d.maxskew = s; }
   d.maxvolts = c->maxvolts;
   d.sample_deltat = (float)c->tdelta_ns / 1e9f;                       // src/readtape.c:1345
   d.tdelta_ns = c->tdelta_ns; d.tstart_ns = c->tstart_ns;
   const float bitspace = density_mode ? 0.0f : 1 / (c->bpi * c->ips);  // src/decoder.c:448 (not initialised during density detection)
   const float hfloor = c->screen_floor_height > 0 ? c->screen_floor_height : 1.0f;
   const double lsb_per_volt = 32767.0 / (double)c->maxvolts;
   float quiet_v = 1e9f;
   for (int p = 0; p < c->nparmsets; ++p) {
      const rtfe_parmset &ps = c->parmset[p];
      DevParm &dp = d.parm[p];
      int W = density_mode ? 8 : (int)(ps.pkww_bitfrac / (c->bpi * c->ips * d.sample_deltat));     // src/readtape.c:1455-1457
      if (W > 50) W = 50;
      if (W < 2) { delete h; return fail(-10, "parmset %d: window of %d samples is too small", p, W); }
      if (ps.agc_window < 0 || ps.agc_window > 10) { delete h; return fail(-11, "parmset %d: agc_window out of range", p); }
      if (ps.agc_window && ps.agc_alpha != 0) { delete h; return fail(-12, "parmset %d: inconsistent AGC parameters", p); }   // src/decoder.c:502
      dp.W = W; dp.rise = ps.pkww_rise; dp.min_peak = ps.min_peak; dp.agc_alpha = ps.agc_alpha; dp.agc_window = ps.agc_window;
      dp.t_clkwindow = bitspace / 2 * ps.clk_factor;                      // src/decoder.c:449
      int s;
      for (s = 0; s < d.nscreens; ++s) if (d.screen[s].W == W) break;
      if (s == d.nscreens) {
         if (s == kMaxScreens) { delete h; return fail(-13, "more than %d distinct window widths", kMaxScreens); }
         d.screen[s].W = W; d.screen[s].rise_i = 1 << 30; d.screen[s].minpk_i = 1 << 30; ++d.nscreens; }
      dp.screen = s;
      // dead-quiet band: no detection is possible from a freshly reset detector (agc 1, baseline 4 V) while
      // |v| < min_peak, or while the signal range 2|v| < pkww_rise
      float q = ps.min_peak > ps.pkww_rise / 2 ? ps.min_peak : ps.pkww_rise / 2;
      if (q < quiet_v) quiet_v = q; }
   // the candidate screens: the loosest thresholds the AGC can ever ask for, given v_avg_height >= hfloor and agc_gain <= 2 (screen_thresholds, rtfe_kernels.hip).
   // A floor the caller gave stands; the default (1 V: every tape clears it) is where the handle starts, and behind each scan of the peak path the
   // floor moves to half the smallest peak height the scan's chains learned (k_adapt_floor; RTFE_ADAPT_FLOOR=0: never)
   d.floor_cfg = hfloor; d.floor_probed = 0; d.probe_min = 0x7fffffff; d.probe_ticket = 0;
   d.probe_on = !(getenv("RTFE_FLOOR_PROBE") && atoi(getenv("RTFE_FLOOR_PROBE")) == 0);
   d.adapt_floor = !(c->screen_floor_height > 0) && !(getenv("RTFE_ADAPT_FLOOR") && atoi(getenv("RTFE_ADAPT_FLOOR")) == 0);
   screen_thresholds(d, hfloor);
   // k_peaks: a row whose margin reaches sure_i passes the rise test for every threshold the chains accept without asking
   // (tightest: baseline 1.5 x full scale at half gain); k_chain hands a burst whose threshold climbs beyond that to the sample path
   for (int s = 0; s < d.nscreens; ++s) {
      float hi_v = 0;
      for (int p = 0; p < c->nparmsets; ++p) if (d.parm[p].screen == s) { const float v = c->parmset[p].pkww_rise * (1.5f * c->maxvolts / 4.0f) / 0.5f; if (v > hi_v) hi_v = v; }
      long long si = (long long)ceil((double)hi_v * lsb_per_volt) + 3;
      if (si > 65535) si = 65535;
      if (si <= d.screen[s].rise_i + 1) si = d.screen[s].rise_i + 2;
      d.screen[s].sure_i = (int)si; }
   if (d.find_zeros) {
      // the zero-crossing detector has no amplitude feedback and no parameter-set dependence (adjust_agc returns
      // at once, src/decoder.c:501): one walker per track; nothing can become pending while |v| <= 0.2 V
      quiet_v = 0.2f;
      d.samples_per_bit = (int)(1 / (bpi_s * c->ips * d.sample_deltat));          // src/readtape.c:1402
      // differentiated: a restart differentiates against 0 and consecutive samples differ by up to 2q, and neither
      // may reach 0.2 V after the x0.4 x samples_per_bit scaling (src/readtape.c:1388)
      if (d.differentiate) quiet_v = 0.24f / (float)(d.samples_per_bit > 0 ? d.samples_per_bit : 1);
      int code = 1;
      while (code < 32767 && !((float)code / 32767 * c->maxvolts > 0.2f)) ++code;     // exact, same expression as the device's volt()
      d.zc_peak_i = code; }
   if (d.differentiate && !d.find_zeros) {
      // peak detection on the differentiated signal (src/readtape.c:1383-1394): a zone is safe where the dead band
      // (|delta| < 0.05 V -> 0) makes the detector's input exact zeros: |v| < 0.024 V covers both the restart (delta
      // against 0) and consecutive samples.  No candidate screen runs in this mode.
      if (quiet_v > 0.024f) quiet_v = 0.024f;
      d.samples_per_bit = (int)(1 / (bpi_s * c->ips * d.sample_deltat));          // src/readtape.c:1402
      for (int p = 0; p < c->nparmsets; ++p) { d.parm[p].screen_rise_v = -1; d.parm[p].screen_minpk_v = -1; } }
   if (c->quiet_volts > 0 && c->quiet_volts < quiet_v) quiet_v = c->quiet_volts;
   d.quiet_i = (int)floor(quiet_v * 0.98 * lsb_per_volt) - 1;
   if (d.quiet_i < 0) d.quiet_i = 0;
   const int spb = (int)(1 / (bpi_s * c->ips * d.sample_deltat));
   int gap = c->gap_min_samples > 0 ? c->gap_min_samples : 32 * (spb > 0 ? spb : 1);
   if (gap < kMarginRows + 128) gap = kMarginRows + 128;
   d.gap_chunks = (gap + kChunkRows - 1) / kChunkRows + 1;                 // quiet-map chunks are groups of 64 rows
   d.zc_parallel = 1;
   if (const char *e = getenv("RTFE_ZC_PARALLEL")) d.zc_parallel = atoi(e) != 0;
   {  // rows a -zeros sub-segment starts early from a fresh state: two bit cells hold a confirmed crossing in each direction wherever
      // the signal is live; where that is not enough the join check sees it and the sub-segment is run again (exact either way)
      const float spbw = (bpi_s > 0 && c->ips > 0) ? 1.0f / (bpi_s * c->ips * d.sample_deltat) : 32.0f;
      int w = ((int)(2.0f * spbw) + 7) & ~7;
      d.zc_warm = w < 16 ? 16 : (w > 64 ? 64 : w); }
   if (const char *e = getenv("RTFE_ZC_WARM")) { const int v = atoi(e) & ~7; if (v >= 8 && v <= 64) d.zc_warm = v; }
   d.tail_rows = 48 * (spb > 0 ? spb : 1);                              // 48 bit cells of silence: every format has ended its block (NRZI ~10, PE 2.5, GCR 6)
   if (const char *e = getenv("RTFE_TAIL_ROWS")) d.tail_rows = atoi(e);   // (tests: 0 = walk the whole gap)
   d.cap_frac = c->events_per_sample_cap > 0 ? c->events_per_sample_cap : 0.125f;
   {
      const char *e = getenv("RTFE_TILE_ROWS");            // tuning knob; the default is what bench.py measures
      // (-zeros in k_decode's zero-crossing mode - k_zeros does not tile -: a lane per (track, 64-row sub-segment), 14 sub-segments x 9 tracks fill two waves)
      int tr = tile_override > 0 ? tile_override : (e ? atoi(e) : ((c->find_zeros && !c->differentiate) ? 64 * (128 / (c->ntrks > 0 ? c->ntrks : 9)) : 512));
      tr = (tr / 64) * 64;
      if (tr < kMarginRows) tr = kMarginRows;
      if (tr > kMaxTileRows) tr = kMaxTileRows;
      d.tile_rows = tr; }
   {
      int wmax = 0;
      for (int sidx = 0; sidx < d.nscreens; ++sidx) if (d.screen[sidx].W > wmax) wmax = d.screen[sidx].W;
      d.halo_rows = (kScreenHalo + wmax + 1 + d.maxskew + 7) & ~7;
      if (d.halo_rows > kMaxHaloRows) d.halo_rows = kMaxHaloRows;
      d.ldw = d.halo_rows + d.tile_rows + 8; }
   d.lsb_per_volt = (float)(32767.0 / (double)c->maxvolts);
   d.debug = getenv("RTFE_DEBUG") ? atoi(getenv("RTFE_DEBUG")) : 0;
   d.cut = getenv("RTFE_CUT") ? atoi(getenv("RTFE_CUT")) : 0;
   // (k_sift_s's plain build: none of the knobs it would otherwise carry as run-time values; RTFE_SIFT_PLAIN=0: the general build everywhere - tests)
   d.pk_plain = !d.invert && d.cut == 0 && d.debug != 3 && !(getenv("RTFE_SIFT_PLAIN") && atoi(getenv("RTFE_SIFT_PLAIN")) == 0);
   {  // The peak path (k_sift -> k_gain -> k_emit): peak detection on the undifferentiated signal.  It pays where flux transitions are a
      // bit cell apart (NRZI): most peaks then have the window to themselves and the chains stay on their steady path.  PE and GCR put
      // a top and a bottom into one window; their bursts take the sample path (k_decode) - RTFE_PEAK_PATH=0/1 overrides (tests keep both
      // paths covered for every format).
      d.peak_path = !d.find_zeros && !d.differentiate && d.mode == RTFE_NRZI;
      if (const char *e = getenv("RTFE_PEAK_PATH")) d.peak_path = !d.find_zeros && !d.differentiate && d.mode != RTFE_WW && atoi(e) != 0;
      d.pk_fast = 1;
      if (const char *e = getenv("RTFE_GAIN_FAST")) d.pk_fast = atoi(e) != 0;
      // the chains' steady stretches in segments (k_gain_seg): 256 records each (measured on C2: 64 / 128 / 256 -> 2.61 / 2.43 / 2.39 ms per scan); the warm-up from the alpha filter's memory
      d.pk_seg_recs = 256;
      d.pk_mar = kPkMar;
      if (const char *e = getenv("RTFE_PK_MAR")) { const int v = atoi(e); if (v >= 0 && v <= kPkMar) d.pk_mar = v; }      // (tests: margins from the samples)
      if (const char *e = getenv("RTFE_SEG_RECS")) { const int v = atoi(e) & ~7; if (v >= 8 && v <= 4096) d.pk_seg_recs = v; }
      d.pk_rejoin = getenv("RTFE_SEG_REJOIN") ? atoi(getenv("RTFE_SEG_REJOIN")) != 0 : 1;      // (0: a chain that breaks is walked to its end by one lane, as before round 5's last part; tests)
      for (int p = 0; p < c->nparmsets; ++p) {
         const float a = c->parmset[p].agc_alpha;
         // (1 - alpha)^n < 2^-26 brings two gains within an ulp; 64 records more for the last ulp to collapse under the filter's own
         // rounding (round 2 measured the same on the old walk: ~50 records left 3 % of the joins a few ulps apart, ~200 none in 8 000)
         int wm = ((a > 0 && a < 1) ? (int)ceil(log(1.0 / 67108864.0) / log(1.0 - (double)a)) : 16) + 64;
         wm = wm > 1024 ? 1024 : wm;
         if (const char *e = getenv("RTFE_SEG_WARM")) { const int v = atoi(e); if (v >= 0 && v <= 4096) wm = v; }
         d.parm[p].seg_warm = wm; }
      const int wmax = sf_wmax(d);
      d.pk_hl = (kPkBack + 2 * wmax + 6 + 7) & ~7;
      d.pk_hr = (wmax + 2 + 7) & ~7;
      // pool slots: flux transitions per tile and head from the bit cell (PE: up to two per cell), 16 bytes each (a record and its margin
      // block), half as much again for noise and weak peaks.  RTFE_PK_SLOT: tests force the capacity path.
      const float spbf = 1.0f / (bpi_s * c->ips * d.sample_deltat);
      const float ppb = c->mode == RTFE_PE ? 2.0f : 1.0f;
      int slot = (int)((float)kSfTile / (spbf > 2 ? spbf : 2) * ppb * 16.0f * 1.5f) + 160;
      // a screen below the noise floor - no amplitude test, or one as low as the rise test: noise wiggles become runs, every row explicit
      bool noisy_screen = false;
      for (int sidx = 0; sidx < d.nscreens; ++sidx) {
         const bool rise_low = (double)d.screen[sidx].rise_i / lsb_per_volt < 0.03, amp_low = d.screen[sidx].minpk_i < 0 || (double)d.screen[sidx].minpk_i / lsb_per_volt < 0.03;
         if (rise_low && amp_low) noisy_screen = true; }
      if (noisy_screen) slot *= 3;
      d.pk_wave_cap = noisy_screen ? 1792 : (getenv("RTFE_PK_WAVECAP") ? atoi(getenv("RTFE_PK_WAVECAP")) : 384);                          // (a wave's two heads have 2 x 896 samples)
      if (const char *e = getenv("RTFE_PK_SLOT")) { const int v = atoi(e); if (v >= 32 && v <= 65536) slot = v; }
      d.pk_slot = (slot + 15) & ~15;
      d.pk_lds = (int)sf_lds_layout(c->ntrks, d.pk_hl, d.pk_hr, d.pk_wave_cap, d.pk_slot).total + 64;
      if (d.pk_lds > 150 * 1024 || sf_nv(d) > 6) d.peak_path = 0; }
   {  // ---- the dense sample path (rtfe_dense.hip) ----
      // parameter sets the front end cannot tell apart are one chain (src/parmsets.c:77-110: four of the five GCR defaults, five of the eight PE ones)
      d.nuset = 0;
      for (int p = 0; p < c->nparmsets; ++p) {
         int u;
         for (u = 0; u < d.nuset; ++u) {
            const DevParm &a = d.parm[d.uset_rep[u]], &b = d.parm[p];
            if (a.W == b.W && a.rise == b.rise && a.min_peak == b.min_peak && a.agc_alpha == b.agc_alpha && a.agc_window == b.agc_window
                && (c->mode != RTFE_PE || a.t_clkwindow == b.t_clkwindow)) break; }
         if (u == d.nuset) { d.uset_rep[u] = p; d.uset_mask[u] = 0; ++d.nuset; }
         d.uset_of[p] = u; d.uset_mask[u] |= 1u << p; }
      if (getenv("RTFE_DENSE_DEDUP") && atoi(getenv("RTFE_DENSE_DEDUP")) == 0) {      // (tests: every set its own chain)
         d.nuset = c->nparmsets;
         for (int p = 0; p < c->nparmsets; ++p) { d.uset_of[p] = p; d.uset_rep[p] = p; d.uset_mask[p] = 1u << p; } }
      // The default for PE and GCR peak detection (round 4): measured against k_decode on 1e7 .. 2.7e8 rows it is 1.05 - 2.4 x faster (DESIGN.md 4d;
      // most where several parameter sets share a window width); RTFE_DENSE_PATH=0 keeps k_decode (the tests run both against each other)
      d.dense_path = !d.peak_path && !d.find_zeros && !d.differentiate && !d.agc_off && (d.mode == RTFE_PE || d.mode == RTFE_GCR);
      if (const char *e = getenv("RTFE_DENSE_PATH")) d.dense_path = !d.peak_path && !d.find_zeros && !d.differentiate && !d.agc_off && d.mode != RTFE_WW && atoi(e) != 0;
      int wpad = 64;
      for (int sidx = 0; sidx < d.nscreens; ++sidx) {
         // (two windows and a little: a lane's warm-up only has to see one detection the true chain sees too; the join checks that it did -
         //  measured on C4: 74 -> 60 rows makes k_dseg 9 % faster and 0.4 % more joins fail)
         int wm = 2 * d.screen[sidx].W + 16; if (wm < 48) wm = 48;
         if (const char *e = getenv("RTFE_DS_WARM")) { const int v = atoi(e); if (v >= 0 && v <= 192) wm = v; }      // (tests: joins that fail)
         d.ds_warm[sidx] = wm; if (wm > wpad) wpad = wm; }
      d.ds_pad = (wpad + 63) & ~63;
      const float spbf = 1.0f / (bpi_s * c->ips * d.sample_deltat);
      const float expect = (float)kDsSub * (c->mode == RTFE_PE ? 2.0f : 1.0f) / (spbf > 2 ? spbf : 2);
      d.ds_cap = expect * 1.3f + 2 <= 15 ? 15 : (expect * 1.3f + 2 <= 31 ? 31 : 63);
      if (const char *e = getenv("RTFE_DS_CAP")) { const int v = atoi(e); if (v >= 1 && v <= 63) d.ds_cap = v; }      // (tests: lists that run full)
      d.ds_slot = ((int)sizeof(DsHdr) + d.ds_cap * (int)sizeof(DsRec) + 15) & ~15;
      {  // sets per classification pass: three where a width carries three or more (one pass over the tile's margins instead of two: PE's default
         // sweep 26.5 -> 24.3 ms per 6.5e7 rows, bench C4 - its width 1.5 carries three distinct sets - 308 -> 280 ms) and two workgroups of 512
         // still fit a CU's 160 KB of LDS.  (With tiles of 512 rows three sets cost a workgroup per CU - three instead of four - and lost on C4.)
         int per[kMaxScreens] = {0, 0, 0, 0}, mx = 0;
         for (int u = 0; u < d.nuset; ++u) { const int sidx = d.parm[d.uset_rep[u]].screen; if (++per[sidx] > mx) mx = per[sidx]; }
         d.ds_up = mx >= 3 ? 3 : 2;
         if (d.ds_up == 3 && (size_t)ds_lds_layout(c->ntrks, d.halo_rows, d.ds_pad + kDsTile + kDsRight, 3).total + sizeof(DevCfg) + 1024 > 80 * 1024) d.ds_up = 2;
         if (const char *e = getenv("RTFE_DS_UP")) { const int v = atoi(e); if (v >= 1 && v <= 3) d.ds_up = v; } }
      d.ds_sfloor = (hfloor < 4.0f ? hfloor : 4.0f) / 4.0f / 2.0f;
      d.ds_lean = getenv("RTFE_DS_LEAN") ? atoi(getenv("RTFE_DS_LEAN")) != 0 : 1;
      d.ds_quiet_s = getenv("RTFE_DS_QUIET_S") ? (float)atof(getenv("RTFE_DS_QUIET_S")) : 0.6f;
      d.ds_band_hi = getenv("RTFE_DS_BAND_HI") ? (float)atof(getenv("RTFE_DS_BAND_HI")) : 1.25f;
      d.ds_band_lo = getenv("RTFE_DS_BAND_LO") ? (float)atof(getenv("RTFE_DS_BAND_LO")) : 1.0f / 3.0f;
      if ((int)ds_lds_layout(c->ntrks, d.halo_rows, d.ds_pad + kDsTile + kDsRight, d.ds_up).total + 64 > 150 * 1024) d.dense_path = 0;
      if ((d.ds_pad + kDsTile + kDsRight) / kStrip * c->ntrks > 8 * kDsThreads) d.dense_path = 0; }      // (k_dseg keeps a lane's strips of the stale-minimum map in eight registers)
   {
      const int nwalk = c->nparmsets * c->ntrks;
      int rc = (24 * 1024) / (nwalk * 24);
      d.rec_cap = rc > 64 ? 64 : (rc < 8 ? 8 : rc);
      // deferred detections per walker and tile (beyond that they are finished on the spot) - sized below, once the rest of
      // k_decode's LDS is known, so that one more workgroup fits a CU where a smaller buffer buys that
      if (const char *e = getenv("RTFE_REC_CAP")) { const int v = atoi(e); if (v >= 4 && v <= d.rec_cap) d.rec_cap = v; } }      // (experiments: LDS per k_decode workgroup)
   if (d.find_zeros) d.rec_cap = 8;                                   // (the zero-crossing walkers store their events at once)
   if (!d.find_zeros) {
      // k_decode is latency bound: workgroups per CU are what counts.  A smaller record buffer is taken only if it lets one more workgroup reside.
      auto per_cu = [&](int rc) { DevCfg t = d; t.rec_cap = rc; return (160 * 1024) / ((int)lds_layout(t).total + 64 + 5500); };      // (static LDS ~ 4 KB, allocation granularity, margin)
      int best = d.rec_cap;
      for (int rc = d.rec_cap; rc >= 8; --rc) if (per_cu(rc) > per_cu(best)) best = rc;
      d.rec_cap = best; }
   h->lds_bytes = (int)lds_layout(d).total + 64;
   if (h->lds_bytes > 160 * 1024 - 2048) { delete h; return fail(-14, "configuration needs %d bytes of LDS", h->lds_bytes); }
   if (getenv("RTFE_VERBOSE")) {
      const LdsLayout Ld = lds_layout(d);
      fprintf(stderr, "rtfe: LDS k_decode %d (tile..bits %u, bits..ldpos %u, ldpos..heights %u, recs %u, walkers %u)\n", h->lds_bytes, Ld.bits, Ld.ldpos - Ld.bits,
              Ld.heights - Ld.ldpos, Ld.nrec - Ld.recs, Ld.fdiff - Ld.walkers);
      fprintf(stderr, "rtfe: peak path %d, k_sift LDS %d (halo %d/%d rows, slots %d bytes, %d vectors per thread)\n", d.peak_path, d.pk_lds, d.pk_hl, d.pk_hr, d.pk_slot, sf_nv(d));
      for (int sidx = 0; sidx < d.nscreens; ++sidx) fprintf(stderr, "rtfe: screen %d W %d rise_i %d minpk_i %d sure_i %d\n", sidx, d.screen[sidx].W, d.screen[sidx].rise_i, d.screen[sidx].minpk_i, d.screen[sidx].sure_i); }
   hipDeviceProp_t prop;
   int dev = 0;
   if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { delete h; return fail(-20, "no HIP device"); }
   h->num_cus = prop.multiProcessorCount;
   if (hipMalloc(&h->d_dev, sizeof(DevCfg)) != hipSuccess) { delete h; return fail(-21, "hipMalloc failed"); }
   if (hipMemcpy(h->d_dev, &d, sizeof(DevCfg), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(h->d_dev); delete h; return fail(-22, "hipMemcpy failed"); }
   raise_dynamic_lds(reinterpret_cast<const void *>(k_decode), h->lds_bytes);
   h->zeros_kernel = getenv("RTFE_ZEROS_KERNEL") ? atoi(getenv("RTFE_ZEROS_KERNEL")) != 0 : 1;
   h->side = nullptr; h->overlap = getenv("RTFE_OVERLAP") ? atoi(getenv("RTFE_OVERLAP")) != 0 : 1;
   h->graphs = getenv("RTFE_GRAPHS") ? atoi(getenv("RTFE_GRAPHS")) != 0 : 0;
   memset(h->gcache, 0, sizeof h->gcache); h->gstamp = 0;
   h->bursts_wpr = getenv("RTFE_BURSTS_WPR") ? atoi(getenv("RTFE_BURSTS_WPR")) : 0;
   h->ds_order = getenv("RTFE_DS_ORDER") ? atoi(getenv("RTFE_DS_ORDER")) : 1;
   h->dchain_wgs = getenv("RTFE_DCHAIN_WGS") ? atoi(getenv("RTFE_DCHAIN_WGS")) : 16;
   h->prep_wgs = getenv("RTFE_PREP_WGS") ? atoi(getenv("RTFE_PREP_WGS")) : 32;
   h->work_cap = getenv("RTFE_WORK_CAP") ? atoll(getenv("RTFE_WORK_CAP")) : -1;
   h->dense_stop = getenv("RTFE_DENSE_STOP") ? atoi(getenv("RTFE_DENSE_STOP")) : 99;
   h->dseg_wgs = getenv("RTFE_DSEG_WGS") ? atoi(getenv("RTFE_DSEG_WGS")) : 0;
   h->dseg_threads = getenv("RTFE_DSEG_THREADS") ? atoi(getenv("RTFE_DSEG_THREADS")) : 0;
   // (k_zeros packs two tracks' 16-bit states into a lane and reads the rows where they lie: no -invert, no deskew delays, a threshold inside int16)
   if (d.invert || d.maxskew > 0 || d.ntrks < 2 || d.zc_peak_i < 1 || d.zc_peak_i > 32767 || !d.zc_parallel) h->zeros_kernel = 0;      // (RTFE_ZC_PARALLEL=0: k_decode's sequential walk, for the tests)
   if (d.peak_path) {                 // (wide rows - 16 tracks and more - do not fit k_sift's tile into LDS: peak_path is off then and the kernel is never launched)
      raise_dynamic_lds(reinterpret_cast<const void *>(sf_kernel(sf_wmax(d), sf_threads(d), sf_nv(d))), d.pk_lds);
      for (int sc = 0; sc < kMaxScreens; ++sc) h->sfs_occ[sc] = 0;
      if (sf_special(d)) for (int sc = 0; sc < d.nscreens; ++sc) {
         raise_dynamic_lds(reinterpret_cast<const void *>(sf_special_sc(d, sc)), sfs_lds_sc(d, sc));
         int nb = 0;
         if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(sf_special_sc(d, sc)), sfs_threads(d), (size_t)sfs_lds_sc(d, sc)) == hipSuccess && nb >= 1) h->sfs_occ[sc] = nb; } }
   if (d.dense_path) raise_dynamic_lds(reinterpret_cast<const void *>(ds_kernel(d.ntrks)), (int)ds_lds_layout(d.ntrks, d.halo_rows, d.ds_pad + kDsTile + kDsRight, d.ds_up).total + 64);
   (void)hipGetLastError();          // (a refused attribute must not linger as the process' "last error": the caller's runtime would report it as its own)
   if (getenv("RTFE_VERBOSE") && d.peak_path) {
      int nb = -1;
      const int thr = sf_special(d) ? sfs_threads(d) : sf_threads(d), lds = sf_special(d) ? sfs_lds(d) : d.pk_lds;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sf_special(d) ? reinterpret_cast<const void *>(sf_special(d)) : reinterpret_cast<const void *>(sf_kernel(sf_wmax(d), sf_threads(d), sf_nv(d))), thr, (size_t)lds);
      fprintf(stderr, "rtfe: k_sift%s %d threads, %d bytes of LDS: %d workgroups per CU (occupancy API), %d CUs\n", sf_special(d) ? "_s" : "", thr, lds, nb, h->num_cus); }
   *out = h;
   return 0; }

constexpr int kTimingRing = 64;
static void timing_free(rtfe_handle *h) {
   if (!h->ev0) return;
   for (int r = 0; r < kTimingRing; ++r) for (int i = 0; i < kNumKernels; ++i) { (void)hipEventDestroy(h->ev0[r][i]); (void)hipEventDestroy(h->ev1[r][i]); }
   delete[] h->ev0; delete[] h->ev1; h->ev0 = nullptr; h->ev1 = nullptr; }

extern "C" int rtfe_set_timing(rtfe_handle *h, int enable) {
   if (!h) return fail(-1, "null argument");
   if (enable && !h->timing) {
      h->ev0 = new hipEvent_t[kTimingRing][12]; h->ev1 = new hipEvent_t[kTimingRing][12];
      for (int r = 0; r < kTimingRing; ++r) for (int i = 0; i < kNumKernels; ++i)
      {
         const bool ok0 = hipEventCreate(&h->ev0[r][i]) == hipSuccess;
         const bool ok1 = ok0 && hipEventCreate(&h->ev1[r][i]) == hipSuccess;
         if (!ok1) {
            // (no half-made ring left behind - the events made so far are destroyed with it, the lone ev0 of this slot too)
            if (ok0) (void)hipEventDestroy(h->ev0[r][i]);
            for (int r2 = 0; r2 <= r; ++r2) for (int i2 = 0; i2 < (r2 < r ? kNumKernels : i); ++i2) { (void)hipEventDestroy(h->ev0[r2][i2]); (void)hipEventDestroy(h->ev1[r2][i2]); }
            delete[] h->ev0; delete[] h->ev1; h->ev0 = nullptr; h->ev1 = nullptr;
            return fail(-40, "hipEventCreate failed"); } }
      h->ev_next = 0; h->ev_pending = 0; }
   if (!enable && h->timing) timing_free(h);
   h->timing = enable != 0;
   return 0; }

extern "C" int rtfe_kernel_ms(rtfe_handle *h, float *out) {
   if (!h || !out || !h->timing) return fail(-41, "timing is not enabled");
   const int n = h->ev_pending < kTimingRing ? h->ev_pending : kTimingRing;
   for (int i = 0; i < kNumKernels; ++i) out[i] = 0;
   for (int k = 0; k < n; ++k) {
      const int r = ((h->ev_next - 1 - k) % kTimingRing + kTimingRing) % kTimingRing;
      for (int i = 0; i < kNumKernels; ++i) {
         float ms = 0;
         if (hipEventSynchronize(h->ev1[r][i]) != hipSuccess) return fail(-42, "hipEventSynchronize failed");
         if (hipEventElapsedTime(&ms, h->ev0[r][i], h->ev1[r][i]) != hipSuccess) return fail(-43, "hipEventElapsedTime failed");
         out[i] += ms; } }
   h->ev_pending = 0;
   return n; }

extern "C" void rtfe_destroy(rtfe_handle *h) {
   if (!h) return;
   if (h->timing) timing_free(h);
   drop_graphs(h);
   if (h->side) { (void)hipEventDestroy(h->ev_fork); (void)hipEventDestroy(h->ev_join); (void)hipEventDestroy(h->ev_fork2); (void)hipEventDestroy(h->ev_join2); (void)hipStreamDestroy(h->side); }
   (void)hipFree(h->d_dev);
   delete h; }

extern "C" int rtfe_pkww_width(const rtfe_handle *h, int parmset) {
   return (h && parmset >= 0 && parmset < h->dev.nparm) ? h->dev.parm[parmset].W : -1; }

static long long nwords_for(const rtfe_handle *h, int64_t nrows) {
   (void)h;
   const long long nchunks = nrows / kChunkRows + 1;
   return (nchunks + 63) / 64 + 1; }

static long long ntiles_for(const rtfe_handle *h, int64_t nrows) { return (nrows + h->dev.tile_rows - 1) / h->dev.tile_rows; }
// workspace: [0,kScratchBytes) scratch | quiet words | the zone search's counts per round | burst control blocks | the peak path's pieces
static long long pk_tiles_for(int64_t nrows) { return (nrows + kSfTile - 1) / kSfTile; }
constexpr int kBrRoundsMax = 1024;                                  // k_bursts_cnt / _emit: a workgroup per round of the zone search, at most this many (else one workgroup does them all)
static size_t ws_rtot_off(const rtfe_handle *h, int64_t nrows) { return kScratchBytes + (size_t)nwords_for(h, nrows) * 8; }      // ... | the rounds' counts
static size_t ws_ctl_off(const rtfe_handle *h, int64_t nrows) { return (ws_rtot_off(h, nrows) + (size_t)kBrRoundsMax * 4 + 255) & ~(size_t)255; }

extern "C" int64_t rtfe_max_bursts(const rtfe_handle *h, int64_t nrows) {
   return (nrows / kChunkRows) / h->dev.gap_chunks + 4; }

// ... | peak path: directory of the tiles' lists | per-chain baseline (k_gain -> k_emit) | record pool
static size_t ws_pkdir_off(const rtfe_handle *h, int64_t nrows) {
   return (ws_ctl_off(h, nrows) + (size_t)rtfe_max_bursts(h, nrows) * sizeof(BurstCtl) + 255) & ~(size_t)255; }
static size_t pk_dir_bytes(const rtfe_handle *h, int64_t nrows) { return ((size_t)(pk_tiles_for(nrows) + 1) * h->dev.nscreens * h->dev.ntrks * sizeof(PeakDir) + 255) & ~(size_t)255; }
static size_t pk_chain_bytes(const rtfe_handle *h, int64_t nrows) { return ((size_t)rtfe_max_bursts(h, nrows) * h->dev.nparm * h->dev.ntrks * sizeof(float) + 255) & ~(size_t)255; }
static size_t ws_pkchain_off(const rtfe_handle *h, int64_t nrows) { return ws_pkdir_off(h, nrows) + (h->dev.peak_path ? pk_dir_bytes(h, nrows) : 0); }
static size_t ws_pkpool_off(const rtfe_handle *h, int64_t nrows) { return ws_pkchain_off(h, nrows) + (h->dev.peak_path ? pk_chain_bytes(h, nrows) : 0); }
static size_t pk_pool_bytes(const rtfe_handle *h, int64_t nrows) {      // one slot per (tile, screen, head)
   if (!h->dev.peak_path) return 0;
   return (((size_t)pk_tiles_for(nrows) * h->dev.nscreens * h->dev.ntrks * (size_t)h->dev.pk_slot) + 255) & ~(size_t)255; }
// ... | the candidates k_sift deferred (k_sift_hard) | their overflow slots
// (two per tile and list: a clean NRZI tape defers 0.05 % of its candidates, a noisy parameter sweep with wide windows one or two per tile and list;
//  past the capacity a candidate becomes a "minimum unknown" record, and the chain that gets there gives up)
static long long pk_hard_cap(const rtfe_handle *h, int64_t nrows) {
   const long long wgs = pk_tiles_for(nrows) < (long long)h->num_cus * 8 ? pk_tiles_for(nrows) : (long long)h->num_cus * 8;      // (k_sift_s: a wave takes the list's places a chunk at a time)
   const long long c = pk_tiles_for(nrows) * h->dev.nscreens * h->dev.ntrks * 2 + 4096 + wgs * 4 * kSfHardChunk * h->dev.nscreens;
   return c > 0x3fffffffll ? 0x3fffffffll : c; }
static size_t ws_pkhard_off(const rtfe_handle *h, int64_t nrows) { return ws_pkpool_off(h, nrows) + pk_pool_bytes(h, nrows); }
static size_t ws_pkqtile_off(const rtfe_handle *h, int64_t nrows);
static size_t ws_pkovf_off(const rtfe_handle *h, int64_t nrows) { return ws_pkhard_off(h, nrows) + (h->dev.peak_path ? (((size_t)pk_hard_cap(h, nrows) * sizeof(SfHard) + 255) & ~(size_t)255) : 0); }
static size_t pk_ovf_bytes(const rtfe_handle *h, int64_t nrows) { return h->dev.peak_path ? (size_t)pk_hard_cap(h, nrows) * kSfOvfBytes : 0; }
// ... | the tiles' quiet bits (k_sift -> k_qpack)
static size_t ws_pkqtile_off(const rtfe_handle *h, int64_t nrows) { return (ws_pkovf_off(h, nrows) + pk_ovf_bytes(h, nrows) + 255) & ~(size_t)255; }
static size_t pk_qtile_bytes(const rtfe_handle *h, int64_t nrows) { return h->dev.peak_path ? (((size_t)pk_tiles_for(nrows) * 2 + 255) & ~(size_t)255) : 0; }
// ... | the streams' tile offsets and totals (k_pscan) | the streams (k_prep): 16-byte records, entry references
static long long pk_hard_cap(const rtfe_handle *h, int64_t nrows);
// (a stream's capacity: every slot's worth of records and a marker per tile.  Deferred candidates add up to three records each; a stream
//  that outgrows the capacity through them - its slots would have to be full of records without margins as well - is not built, and its
//  chains give up)
static long long pk_ccap(const rtfe_handle *h, int64_t nrows) { return pk_tiles_for(nrows) * (h->dev.pk_slot / 8 + 1) + 64; }
static size_t ws_pktstart_off(const rtfe_handle *h, int64_t nrows) { return ws_pkqtile_off(h, nrows) + pk_qtile_bytes(h, nrows); }
static size_t pk_tstart_bytes(const rtfe_handle *h, int64_t nrows) {      // tile offsets | chunk totals | chunk offsets | stream totals
   const size_t nl = (size_t)h->dev.nscreens * h->dev.ntrks, nch = (size_t)(pk_tiles_for(nrows) + 1023) / 1024;
   return h->dev.peak_path ? ((((size_t)pk_tiles_for(nrows) + 2 * nch + 2) * nl * 4 + 255) & ~(size_t)255) : 0; }
static size_t ws_pkcrec_off(const rtfe_handle *h, int64_t nrows) { return ws_pktstart_off(h, nrows) + pk_tstart_bytes(h, nrows); }
static size_t pk_crec_bytes(const rtfe_handle *h, int64_t nrows) { return h->dev.peak_path ? (((size_t)pk_ccap(h, nrows) * h->dev.nscreens * h->dev.ntrks * sizeof(CRec) + 255) & ~(size_t)255) : 0; }
static size_t ws_pkeref_off(const rtfe_handle *h, int64_t nrows) { return ws_pkcrec_off(h, nrows) + pk_crec_bytes(h, nrows); }
// ... | how many records a list's deferred candidates add to (or take from) its stream (k_sift_hard -> k_pscan)
static size_t ws_pkextra_off(const rtfe_handle *h, int64_t nrows);
static size_t pk_extra_bytes(const rtfe_handle *h, int64_t nrows) { return h->dev.peak_path ? (((size_t)pk_tiles_for(nrows) * h->dev.nscreens * h->dev.ntrks * 4 + 255) & ~(size_t)255) : 0; }
static size_t pk_eref_bytes(const rtfe_handle *h, int64_t nrows) { return h->dev.peak_path ? (((size_t)pk_ccap(h, nrows) * h->dev.nscreens * h->dev.ntrks * 8 + 255) & ~(size_t)255) : 0; }      // (a margin block per record, in stream order)

static size_t ws_pkextra_off(const rtfe_handle *h, int64_t nrows) { return ws_pkeref_off(h, nrows) + pk_eref_bytes(h, nrows); }
// ... | the chains between k_gain (heads), k_gain_s (steady stretches) and k_gain (tails)
static size_t ws_pkcst_off(const rtfe_handle *h, int64_t nrows) { return ws_pkextra_off(h, nrows) + pk_extra_bytes(h, nrows); }
static size_t pk_cst_bytes(const rtfe_handle *h, int64_t nrows) { return h->dev.peak_path ? (((size_t)rtfe_max_bursts(h, nrows) * h->dev.nparm * h->dev.ntrks * sizeof(ChainSt) + 255) & ~(size_t)255) : 0; }

// ... | the segments of the chains' steady stretches (k_segplan -> k_gain_seg -> k_gain_join)
static long long pk_seg_cap(const rtfe_handle *h, int64_t nrows) {
   if (!h->dev.peak_path) return 0;
   if (const char *e = getenv("RTFE_SEG_CAP")) { const long long v = atoll(e); if (v >= 1) return v; }      // (tests: a table that runs full)
   const long long S = h->dev.pk_seg_recs;
   const long long per_stream = pk_ccap(h, nrows) / S + 1;
   return per_stream * h->dev.nscreens * h->dev.ntrks * (h->dev.nparm < 2 ? 1 : 2) + rtfe_max_bursts(h, nrows) * h->dev.nparm * h->dev.ntrks + 64; }
static size_t ws_pksegs_off(const rtfe_handle *h, int64_t nrows) { return ws_pkcst_off(h, nrows) + pk_cst_bytes(h, nrows); }
static size_t pk_segs_bytes(const rtfe_handle *h, int64_t nrows) { return ((size_t)pk_seg_cap(h, nrows) * sizeof(GsSeg) + 255) & ~(size_t)255; }

// ... | per record of a segment: the gain in force if it fired (k_gain_seg -> k_emit_seg)
static size_t ws_pkgfire_off(const rtfe_handle *h, int64_t nrows) { return ws_pksegs_off(h, nrows) + pk_segs_bytes(h, nrows); }
static size_t pk_gfire_bytes(const rtfe_handle *h, int64_t nrows) { return ((size_t)pk_seg_cap(h, nrows) * (size_t)(h->dev.pk_seg_recs > 0 ? h->dev.pk_seg_recs : 0) * 4 + 255) & ~(size_t)255; }

// (k_prep's work list for k_clear lives in the gains' region - nothing reads it behind k_clear; 8 bytes a place, counted in an int)
static long long pk_work_cap(const rtfe_handle *h, int64_t nrows) {
   long long c = (long long)(pk_gfire_bytes(h, nrows) / 8);
   if (h->work_cap >= 0 && h->work_cap < c) c = h->work_cap;      // (RTFE_WORK_CAP, tests: a list that runs full - what does not fit stays unmarked, the general step's)
   return c > 0x7ffffe00ll ? 0x7ffffe00ll : c; }

// ... | the dense sample path: per (tile, width) "nothing above the screen" | per (sub-segment, track) the band | the slots
static long long ds_tiles_for(int64_t nrows) { return (nrows + kDsTile - 1) / kDsTile; }
static size_t ws_dsdead_off(const rtfe_handle *h, int64_t nrows) { return (ws_pkgfire_off(h, nrows) + pk_gfire_bytes(h, nrows) + 255) & ~(size_t)255; }
static size_t ds_dead_bytes(const rtfe_handle *h, int64_t nrows) { return h->dev.dense_path ? (((size_t)ds_tiles_for(nrows) * h->dev.nscreens + 255) & ~(size_t)255) : 0; }
static size_t ws_dsband_off(const rtfe_handle *h, int64_t nrows) { return ws_dsdead_off(h, nrows) + ds_dead_bytes(h, nrows); }
static size_t ds_band_bytes(const rtfe_handle *h, int64_t nrows) { (void)h; (void)nrows; return 0; }      // (the bands travel in the slots' headers: the region of their own is gone - ADVICE r4)
static size_t ws_dsslot_off(const rtfe_handle *h, int64_t nrows) { return ws_dsband_off(h, nrows) + ds_band_bytes(h, nrows); }
static size_t ds_slot_bytes(const rtfe_handle *h, int64_t nrows) { return h->dev.dense_path ? (((size_t)ds_tiles_for(nrows) * kDsJ * h->dev.nuset * h->dev.ntrks * (size_t)h->dev.ds_slot + 511) & ~(size_t)255) : 0; }      // (+ a slot's worth: k_dchain reads nine 16-byte units of the last slot)

extern "C" size_t rtfe_workspace_bytes(const rtfe_handle *h, int64_t nrows) {
   return ws_dsslot_off(h, nrows) + ds_slot_bytes(h, nrows) + 256; }

extern "C" int64_t rtfe_event_capacity(const rtfe_handle *h, int64_t nrows) {
   const double per_track = (double)nrows * h->dev.cap_frac + 128.0 * (double)rtfe_max_bursts(h, nrows) + (double)kMarginRows;
   return (int64_t)(per_track * h->dev.nparm * h->dev.ntrks) + 4096; }

static int launch_check(const char *what) {
   hipError_t e = hipGetLastError();
   if (e != hipSuccess) return fail(-30, "%s: %s", what, hipGetErrorString(e));
   return 0; }

// quiet map -> burst table.  Long maps: a workgroup per round of 2^24 rows in two passes and a tail (k_bursts_cnt / _emit / _tail); short ones
// (or absurdly long ones): the single workgroup that does the rounds in turn.  RTFE_BURSTS_WPR: words per round (tests: many rounds on a short tape).
static void launch_bursts(const rtfe_handle *h, hipStream_t s, const unsigned long long *qwords, long long nwords, long long nchunks, int64_t nrows, int64_t own_rows,
                          int first_is_tape_start, int64_t event_capacity, rtfe_burst *d_bursts, long long maxb, BurstScratch *scratch, int32_t *d_nbursts, uint32_t *rtot) {
   const int wpr_env = h->bursts_wpr;
   const int wpr = (wpr_env >= 1 && wpr_env <= kBrWords) ? wpr_env : kBrWords;
   const long long rounds = (nwords + wpr - 1) / wpr;
   if ((wpr_env > 0 || rounds >= 3) && rounds <= kBrRoundsMax && h->dev.debug != 5) {
      hipLaunchKernelGGL(k_bursts_cnt, dim3((unsigned)rounds), dim3(1024), 0, s, qwords, nwords, nchunks, wpr, h->dev.gap_chunks, first_is_tape_start, maxb, rtot);
      hipLaunchKernelGGL(k_bursts_emit, dim3((unsigned)rounds), dim3(1024), 0, s, qwords, nwords, nchunks, (long long)nrows, wpr, h->dev.gap_chunks, first_is_tape_start, d_bursts, maxb, (const uint32_t *)rtot);
      hipLaunchKernelGGL(k_bursts_tail, dim3(1), dim3(1024), 0, s, (int)rounds, (const uint32_t *)rtot, (long long)nrows, (long long)own_rows, h->dev.ntrks, h->dev.gap_chunks,
                         h->dev.cap_frac, h->dev.nparm, (long long)event_capacity, d_bursts, maxb, scratch, d_nbursts); }
   else
      hipLaunchKernelGGL(k_bursts, dim3(1), dim3(1024), 0, s, qwords, nwords, nchunks, (long long)nrows, (long long)own_rows, h->dev.ntrks,
                         h->dev.gap_chunks, first_is_tape_start, h->dev.cap_frac, h->dev.nparm, (long long)event_capacity, d_bursts, maxb, scratch, d_nbursts, h->dev.debug == 5 ? 1 : 0); }

// the handle's side stream (made at the first scan that wants it - outside any stream capture)
static void ensure_side(rtfe_handle *h) {
   if (h->side || !h->overlap) return;
   // (the side stream's kernels are a few workgroups each, the main stream's fill the chip: at the default priority k_bursts_tail sat behind k_prep's
   //  workgroups for 0.25 ms once k_sift_hard - fewer deferred candidates under the raised floor - no longer held k_prep back)
   int pr_least = 0, pr_greatest = 0;
   (void)hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
   if (hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, pr_greatest) != hipSuccess) { (void)hipGetLastError(); h->side = nullptr; h->overlap = 0; }
   else { (void)hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming); (void)hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming);
          (void)hipEventCreateWithFlags(&h->ev_fork2, hipEventDisableTiming); (void)hipEventCreateWithFlags(&h->ev_join2, hipEventDisableTiming); } }

static int scan_launch(rtfe_handle *h, const int16_t *d_rows, int64_t nrows, int64_t own_rows, int64_t row_base, int first_is_tape_start,
                       void *d_workspace, size_t workspace_bytes,
                       rtfe_burst *d_bursts, int64_t max_bursts, int32_t *d_nbursts,
                       uint32_t *d_counts, rtfe_event *d_events, int64_t event_capacity, void *stream) {
   if (!h || !d_rows || !d_workspace || !d_bursts || !d_nbursts || !d_counts || !d_events) return fail(-1, "null argument");
   if (h->dev.mode == RTFE_WW) return fail(-44, "Whirlwind tapes have no independent bursts: use rtfe_ww_scan");
   if (((uintptr_t)d_rows & 15) != 0) return fail(-31, "d_rows must be 16-byte aligned");
   if (workspace_bytes < rtfe_workspace_bytes(h, nrows)) return fail(-32, "workspace too small");
   // the peak path addresses rows with 32 bits and a record's margin entries by a 32-bit index (2-byte units) into pool + overflow slots
   if (h->dev.peak_path && (nrows >= 0x7ff00000ll || (ws_pkqtile_off(h, nrows) - ws_pkpool_off(h, nrows)) / 2 >= 0xffffffffull))
      return fail(-36, "%lld rows are too many for one rtfe_scan on the peak path: scan the tape in fragments (own_rows)", (long long)nrows);
   if (nrows <= 0 || max_bursts < 1) return fail(-33, "nothing to scan");
   if (own_rows <= 0 || own_rows > nrows) return fail(-35, "own_rows must be in (0, nrows]");
   hipStream_t st = (hipStream_t)stream;
   const long long nchunks = nrows / kChunkRows;                      // complete groups of 64 rows
   const long long nwords = nwords_for(h, nrows);
   BurstScratch *scratch = reinterpret_cast<BurstScratch *>(d_workspace);
   unsigned long long *qwords = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(d_workspace) + kScratchBytes);
   int grid = (int)(nwords < (long long)h->num_cus * 8 ? nwords : (long long)h->num_cus * 8);
   BurstCtl *ctlp = reinterpret_cast<BurstCtl *>(reinterpret_cast<char *>(d_workspace) + ws_ctl_off(h, nrows));
   char *wsb = reinterpret_cast<char *>(d_workspace);
   // one wave per 64 walkers: k_decode's walk phase is latency bound, so small workgroups (many resident per CU)
   // beat wide ones; it holds ~250 VGPRs => 2 waves/SIMD => 8 waves per CU
   const int nwalk = h->dev.nparm * h->dev.ntrks;
   int threads = nwalk <= 64 ? 64 : (nwalk <= 128 ? 128 : 256);
   if (h->dev.find_zeros && !h->dev.differentiate) {                 // -zeros through k_decode (rtfe_zeros.hip says when): a lane per (track, 64-row sub-segment) of the tile
      const int lanes = h->dev.ntrks * (h->dev.tile_rows / 64);
      while (threads < lanes && threads < 256) threads *= 2; }
   int per_cu = (160 * 1024) / (h->lds_bytes + 1024);
   const int wave_lim = 8 / (threads / 64);
   if (per_cu > wave_lim) per_cu = wave_lim;
   if (per_cu < 1) per_cu = 1;
   const int dgrid = h->num_cus * per_cu;
   // every span's events are recorded by every scan (a span that does not run reads ~0)
   bool ran[kNumKernels] = {false};
   const int evset = h->timing ? h->ev_next : 0;
   if (h->timing) { h->ev_next = (h->ev_next + 1) % kTimingRing; ++h->ev_pending; }
   auto t0s = [&](int k, hipStream_t s2) { ran[k] = true; if (h->timing) (void)hipEventRecord(h->ev0[evset][k], s2); };
   auto t1s = [&](int k, hipStream_t s2) { if (h->timing) (void)hipEventRecord(h->ev1[evset][k], s2); };
   auto t0 = [&](int k) { t0s(k, st); };
   auto t1 = [&](int k) { t1s(k, st); };
   auto skip_rest = [&]() { for (int k = 0; k < kNumKernels; ++k) if (!ran[k]) { t0(k); t1(k); } };
   if (h->dev.peak_path) {
      // ---- the peak path: k_sift (quiet map + records) -> k_bursts -> k_zones -> k_gain -> k_emit -> k_publish -> whatever the chains gave up ----
      PeakDir *dirm = reinterpret_cast<PeakDir *>(wsb + ws_pkdir_off(h, nrows));
      float *chainh = reinterpret_cast<float *>(wsb + ws_pkchain_off(h, nrows));
      unsigned char *pkpool = reinterpret_cast<unsigned char *>(wsb + ws_pkpool_off(h, nrows));
      SfHard *hardp = reinterpret_cast<SfHard *>(wsb + ws_pkhard_off(h, nrows));
      unsigned char *ovfp = reinterpret_cast<unsigned char *>(wsb + ws_pkovf_off(h, nrows));
      const int hard_cap = (int)pk_hard_cap(h, nrows);
      const long long ptiles = pk_tiles_for(nrows);
      // the scratch block and the deferred candidates' counts cleared, and - a handle's first scan - the screen's floor estimated from the samples (k_scan_begin)
      hipLaunchKernelGGL(k_scan_begin, dim3(h->num_cus * 4), dim3(256), 0, st, h->d_dev, scratch, reinterpret_cast<uint4 *>(wsb + ws_pkextra_off(h, nrows)), (long long)(pk_extra_bytes(h, nrows) / 16), d_rows, (long long)nrows);
      const int stop_after = getenv("RTFE_PEAK_STOP") ? atoi(getenv("RTFE_PEAK_STOP")) : 99;      // (debugging: launch only the first n kernels of the path)
      t0(kTSift);
      const sfs_kernel_t sfs = sf_special(h->dev);
      const int pthreads = sfs ? sfs_threads(h->dev) : sf_threads(h->dev), plds = sfs ? sfs_lds(h->dev) : h->dev.pk_lds;
      int spc = (160 * 1024) / (plds + 512);
      if (spc * (pthreads / 64) > 32) spc = 32 / (pthreads / 64);
      if (const char *e = getenv("RTFE_SIFT_WGS")) { const int v = atoi(e); if (v >= 1 && v < spc) spc = v; }
      if (spc < 1) spc = 1;
      long long pgrid = (long long)h->num_cus * spc;
      if (pgrid > ptiles) pgrid = ptiles;
      uint16_t *qtile = reinterpret_cast<uint16_t *>(wsb + ws_pkqtile_off(h, nrows));
      if (sfs) {
         for (int sc = 0; sc < h->dev.nscreens; ++sc) {
            SfArgs a;
            a.rows = d_rows; a.nrows = nrows; a.ntiles = (int)ptiles; a.qtile = sc == 0 ? qtile : nullptr; a.dir = dirm; a.pool = pkpool; a.hard = hardp; a.hard_cap = hard_cap;
            a.hard_count = &scratch->hard_count; a.dbg = scratch->scr; a.hcap = h->dev.pk_slot; a.wave_cap = h->dev.pk_wave_cap; a.invert = h->dev.invert;
            a.quiet_i = h->dev.quiet_i; a.cfg = h->d_dev; a.hi_i = h->dev.screen[sc].sure_i;
            a.cut = h->dev.cut; a.debug = h->dev.debug; a.defer = 1; a.nscreens = h->dev.nscreens; a.sc = sc;
            const int lds_sc = sfs_lds_sc(h->dev, sc);
            int spc_sc = (160 * 1024) / (lds_sc + 512);
            if (spc_sc > spc) spc_sc = spc;
            if (h->sfs_occ[sc] >= 1 && spc_sc > h->sfs_occ[sc]) spc_sc = h->sfs_occ[sc];      // (persistent workgroups: never more than are resident at once - the registers may allow fewer than the LDS)
            if (spc_sc < 1) spc_sc = 1;
            long long grid_sc = (long long)h->num_cus * spc_sc;
            if (grid_sc > ptiles) grid_sc = ptiles;
            hipLaunchKernelGGL(sf_special_sc(h->dev, sc), dim3((unsigned)grid_sc), dim3(pthreads), lds_sc, st, a); } }
      else {
         const sf_kernel_t sfk = sf_kernel(sf_wmax(h->dev), pthreads, sf_nv(h->dev));
         hipLaunchKernelGGL(sfk, dim3((unsigned)pgrid), dim3(pthreads), h->dev.pk_lds, st, (const DevCfg *)h->d_dev, d_rows, (long long)nrows, ptiles,
                            qtile, dirm, pkpool, hardp, hard_cap, &scratch->hard_count, scratch->scr); }
      t1(kTSift);
      // Two things only need k_sift's output and not each other: (a) quiet map -> burst table -> restart rows (k_qpack, k_bursts, k_zones:
      // latency-bound, a workgroup or a wave per burst) and (b) the lists -> the streams (k_sift_hard, k_pscan, k_prep).  (a) runs on a
      // stream of the handle's own beside (b) and joins in front of the chains (RTFE_OVERLAP=0: in line; its span then is work, else the
      // time it shared the device).
      hipStream_t sa = st;
      if (h->overlap && stop_after >= 99) {
         ensure_side(h);
         if (h->side) { sa = h->side; (void)hipEventRecord(h->ev_fork, st); (void)hipStreamWaitEvent(sa, h->ev_fork, 0); } }
      t0s(kTBursts, sa);
      hipLaunchKernelGGL(k_qpack, dim3(h->num_cus < 64 ? h->num_cus : 64), dim3(256), 0, sa, (const uint16_t *)qtile, ptiles, qwords, nwords);
      launch_bursts(h, sa, qwords, nwords, nchunks, nrows, own_rows, first_is_tape_start, event_capacity, d_bursts,
                    (long long)(max_bursts < rtfe_max_bursts(h, nrows) ? max_bursts : rtfe_max_bursts(h, nrows)), scratch, d_nbursts, reinterpret_cast<uint32_t *>(wsb + ws_rtot_off(h, nrows)));
      if (stop_after >= 3)
         hipLaunchKernelGGL(k_zones, dim3(h->num_cus * 8), dim3(64), 0, sa, h->d_dev, d_rows, (long long)nrows, (const rtfe_burst *)d_bursts,
                            (const BurstScratch *)scratch, ctlp);
      t1s(kTBursts, sa);
      if (sa != st) (void)hipEventRecord(h->ev_join, sa);
      // the lists -> one stream of 16-byte records per (screen, head): the deferred candidates resolved (k_sift_hard), the streams' tile
      // offsets (k_pscan), the records copied over with their absolute rows, volts and entry references (k_prep)
      t0(kTPrep);
      const int nlists = h->dev.nscreens * h->dev.ntrks;
      uint32_t *tstartp = reinterpret_cast<uint32_t *>(wsb + ws_pktstart_off(h, nrows));
      const int nsc = (int)((ptiles + 1023) / 1024);                    // chunks of 1024 tiles (k_pscan)
      uint32_t *ctotcp = tstartp + (size_t)ptiles * nlists, *coffp = ctotcp + (size_t)nsc * nlists, *ctotp = coffp + (size_t)nsc * nlists;
      CRec *crecp = reinterpret_cast<CRec *>(wsb + ws_pkcrec_off(h, nrows));
      uint2 *erefp = reinterpret_cast<uint2 *>(wsb + ws_pkeref_off(h, nrows));
      int *extrap = reinterpret_cast<int *>(wsb + ws_pkextra_off(h, nrows));
      const long long ccap = pk_ccap(h, nrows);
      hipLaunchKernelGGL(k_sift_hard, dim3(h->num_cus * 16), dim3(256), 0, st, (const DevCfg *)h->d_dev, d_rows, (long long)nrows, (const SfHard *)hardp, hard_cap,
                         (const int *)&scratch->hard_count, ovfp, extrap, scratch->dbg2);
      hipLaunchKernelGGL(k_pscan1, dim3(nsc), dim3(1024), 0, st, (const PeakDir *)dirm, (const int *)extrap, (int)ptiles, nlists, tstartp, ctotcp);
      hipLaunchKernelGGL(k_pscan2, dim3(1), dim3(1024), 0, st, nsc, nlists, (const uint32_t *)ctotcp, coffp, ctotp);
      PrepArgs ppa; ppa.nlists = nlists; ppa.ntrks = h->dev.ntrks; ppa.hcap = h->dev.pk_slot; ppa.mv = h->dev.maxvolts;
      // (workgroups per CU: 4 / 8 / 16 measured 0.51 / 0.47 / 0.42 ms for the span on C2 - half a wave per list, the more lists in flight the better)
      hipLaunchKernelGGL(k_prep, dim3(h->num_cus * (h->prep_wgs >= 1 && h->prep_wgs <= 4096 ? h->prep_wgs : 32)), dim3(256), 0, st, ppa, (const PeakDir *)dirm, (const unsigned char *)pkpool, (const unsigned char *)ovfp,
                         (const uint32_t *)tstartp, (const uint32_t *)coffp, (const uint32_t *)ctotp, ptiles, ccap, crecp, erefp,
                         reinterpret_cast<unsigned long long *>(wsb + ws_pkgfire_off(h, nrows)), pk_work_cap(h, nrows), &scratch->prep_work);
      // (the work list lives where k_gain_seg's gains will: nothing reads it behind k_clear)
      hipLaunchKernelGGL(k_clear, dim3(h->num_cus * 4), dim3(256), 0, st, (const unsigned long long *)(wsb + ws_pkgfire_off(h, nrows)), pk_work_cap(h, nrows), (const int *)&scratch->prep_work,
                         (const uint32_t *)ctotp, ccap, crecp);
#ifdef RTFE_CPU_EMUL
      if (getenv("RTFE_PREP_CHECK")) hipLaunchKernelGGL(k_prep_check, dim3(1), dim3(64), 0, st, (const DevCfg *)h->d_dev, (const uint32_t *)ctotp, ccap, (const CRec *)crecp);
#endif
      t1(kTPrep);
      if (sa != st) (void)hipStreamWaitEvent(st, h->ev_join, 0);        // join: the chains need both
      if (stop_after < 3) { skip_rest(); return launch_check("rtfe_scan"); }
      t0(kTGain);
      ChainSt *cstp = reinterpret_cast<ChainSt *>(wsb + ws_pkcst_off(h, nrows));
      // the chains: from the restart row until the baseline is fixed (k_gain, mode 0), the steady stretch (k_gain_s), whatever that stopped at (k_gain, mode 1)
      for (int mode = 0; mode < 2; ++mode) {
         if (mode == 1) t0(kTGainTail);
         hipLaunchKernelGGL(k_gain, dim3(h->num_cus * 4), dim3(64), 0, st, (const DevCfg *)h->d_dev, mode, cstp, (long long)nrows, (long long)row_base, (const rtfe_burst *)d_bursts,
                            scratch, ctlp, d_counts, d_events, chainh, (const CRec *)crecp, (const uint2 *)erefp, (const uint32_t *)tstartp, (const uint32_t *)coffp, (const uint32_t *)ctotp, ccap,
                            (const unsigned char *)pkpool, ptiles, reinterpret_cast<GsSeg *>(wsb + ws_pksegs_off(h, nrows)), pk_seg_cap(h, nrows), (const int16_t *)d_rows);
         if (mode == 0) {
            t1(kTGain); t0(kTGainS);
            // the steady stretches: in segments, every one on its own, joined where the states agree bit for bit (rtfe_gain.hip)
            GsSeg *segp = reinterpret_cast<GsSeg *>(wsb + ws_pksegs_off(h, nrows));
            float *gfirep = reinterpret_cast<float *>(wsb + ws_pkgfire_off(h, nrows));
            hipLaunchKernelGGL(k_gain_seg, dim3(h->num_cus * 16), dim3(64), 0, st, (const DevCfg *)h->d_dev, (const ChainSt *)cstp, scratch, (const CRec *)crecp, ccap, segp, (const int *)&scratch->nsegs, pk_seg_cap(h, nrows), gfirep);
            hipLaunchKernelGGL(k_gain_join, dim3(h->num_cus * 2), dim3(64), 0, st, (const DevCfg *)h->d_dev, cstp, (const rtfe_burst *)d_bursts, scratch, (const BurstCtl *)ctlp, d_counts, chainh, segp);
            t1(kTGainS); } }
      t1(kTGainTail);
      if (stop_after < 4) { skip_rest(); return launch_check("rtfe_scan"); }
      t0(kTEmit);
      // the heads' and tails' noted events (k_emit: a wave per chain at a time - a chain's head and tail hold a few dozen) are finished on the side stream
      // beside the segments' events (k_emit_seg): disjoint parts of the event lists, both only need the chains
      hipStream_t se = st;
      if (sa != st) { se = sa; (void)hipEventRecord(h->ev_fork2, st); (void)hipStreamWaitEvent(se, h->ev_fork2, 0); }
      hipLaunchKernelGGL(k_emit, dim3(h->num_cus * 32), dim3(64), 0, se,
                         h->d_dev, (const rtfe_burst *)d_bursts, (const BurstScratch *)scratch,
                         (const BurstCtl *)ctlp, (const uint32_t *)d_counts, d_events, (const float *)chainh, (const CRec *)crecp, (const uint2 *)erefp, ccap, (const unsigned char *)pkpool, (const ChainSt *)cstp, d_rows, (long long)nrows, (const GsSeg *)(wsb + ws_pksegs_off(h, nrows)));
      if (se != st) (void)hipEventRecord(h->ev_join2, se);
      hipLaunchKernelGGL(k_emit_seg, dim3(h->num_cus * 16), dim3(256), 0, st, (const DevCfg *)h->d_dev, (const ChainSt *)cstp, (const BurstCtl *)ctlp, d_events, (const CRec *)crecp, (const uint2 *)erefp, ccap,
                         (const unsigned char *)pkpool, (const GsSeg *)(wsb + ws_pksegs_off(h, nrows)), (const int *)&scratch->nsegs, pk_seg_cap(h, nrows), (const float *)(wsb + ws_pkgfire_off(h, nrows)), (int)(rtfe_max_bursts(h, nrows) * h->dev.nparm * h->dev.ntrks), d_rows, (long long)nrows);
      if (se != st) (void)hipStreamWaitEvent(st, h->ev_join2, 0);
      hipLaunchKernelGGL(k_publish, dim3(h->num_cus < 64 ? h->num_cus : 64), dim3(256), 0, st, h->d_dev, (long long)nrows, d_bursts, scratch, ctlp);
      if (h->dev.nuset < h->dev.nparm)      // parameter sets the front end cannot tell apart were ONE chain: its events into the other sets' regions
         hipLaunchKernelGGL(k_dup_sets, dim3(h->num_cus * 8), dim3(256), 0, st, (const DevCfg *)h->d_dev, (const rtfe_burst *)d_bursts, (const BurstScratch *)scratch, (const BurstCtl *)ctlp, d_counts, d_events);
      t1(kTEmit);
      if (stop_after < 5) { skip_rest(); return launch_check("rtfe_scan"); }
      t0(kTDecode);
      hipLaunchKernelGGL(k_decode, dim3(dgrid), dim3(threads), h->lds_bytes, st, h->d_dev, d_rows, (long long)nrows,
                         (long long)row_base, d_bursts, scratch, d_counts, d_events, 0xffffffffu, 0, 0, (int)kDecodeRedo, ctlp);
      t1(kTDecode);
      if (h->dev.adapt_floor) hipLaunchKernelGGL(k_adapt_floor, dim3(1), dim3(1), 0, st, h->d_dev, (const BurstScratch *)scratch);      // (the NEXT scan's screen)
      skip_rest();
      return launch_check("rtfe_scan"); }
   // ---- the sample path: quiet map -> bursts -> every burst in one pass over its samples ----
   (void)hipMemsetAsync(scratch, 0, kScratchBytes, st);
   const long long dtiles = ds_tiles_for(nrows);
   unsigned char *deadp = reinterpret_cast<unsigned char *>(wsb + ws_dsdead_off(h, nrows));
   unsigned char *slotp = reinterpret_cast<unsigned char *>(wsb + ws_dsslot_off(h, nrows));
   if (h->dev.dense_path) {
      // PE, GCR peak detection (rtfe_dense.hip): the dense pass first - it reads every row anyway, and leaves the quiet map's bits as well
      // (k_quiet folded in: one pass over the tape less); bursts and restart rows behind it, then a lane per chain
      (void)hipMemsetAsync(qwords, 0, (size_t)nwords * 8, st);         // (a tile writes its own byte: the map's tail stays "not quiet")
      const int dlds = (int)ds_lds_layout(h->dev.ntrks, h->dev.halo_rows, h->dev.ds_pad + kDsTile + kDsRight, h->dev.ds_up).total + 64;
      int dpc = (160 * 1024) / (dlds + 1024);
      if (dpc > 8) dpc = 8;
      if (h->dseg_wgs >= 1 && h->dseg_wgs < dpc) dpc = h->dseg_wgs;
      if (dpc < 1) dpc = 1;
      long long dg = (long long)h->num_cus * dpc;
      if (dg > dtiles) dg = dtiles;
      t0(kTDseg);
      // (threads: fewer waves that fill the rounds of the all-lane phases better - 448 / 384 / 320 threads for 9 tracks - are SLOWER: G1's k_dseg 84.8 / 92.1 / 93.8 ms
      //  against 78.3 with 512: the kernel lives on its resident waves, not on its lane-task count; RTFE_DSEG_THREADS for experiments)
      int dthreads = kDsThreads;
      if (h->dseg_threads >= 64 && h->dseg_threads <= kDsThreads) dthreads = h->dseg_threads / 64 * 64;
      hipLaunchKernelGGL(ds_kernel(h->dev.ntrks), dim3((unsigned)dg), dim3(dthreads), dlds, st, (const DevCfg *)h->d_dev, d_rows, (long long)nrows, dtiles, deadp, reinterpret_cast<unsigned char *>(qwords), slotp, scratch->scr);
      t1(kTDseg); }
   else {
      t0(kTQuiet);
      hipLaunchKernelGGL(k_quiet, dim3(grid), dim3(256), 0, st, d_rows, (long long)nrows, h->dev.ntrks, h->dev.quiet_i, qwords, nwords);
      t1(kTQuiet); }
   t0(kTBursts);
   launch_bursts(h, st, qwords, nwords, nchunks, nrows, own_rows, first_is_tape_start, event_capacity, d_bursts,
                 (long long)(max_bursts < rtfe_max_bursts(h, nrows) ? max_bursts : rtfe_max_bursts(h, nrows)), scratch, d_nbursts, reinterpret_cast<uint32_t *>(wsb + ws_rtot_off(h, nrows)));
   if (h->dev.dense_path)                                              // the restart rows of all bursts (rtfe_gain.hip)
      hipLaunchKernelGGL(k_zones, dim3(h->num_cus * 8), dim3(64), 0, st, h->d_dev, d_rows, (long long)nrows, (const rtfe_burst *)d_bursts,
                         (const BurstScratch *)scratch, ctlp);
   t1(kTBursts);
   if (h->dev.find_zeros && !h->dev.differentiate && h->zeros_kernel) {          // -zeros: the lean kernel of its own (rtfe_zeros.hip)
      t0(kTZeros);
      const dim3 zg(h->num_cus * (RTFE_ZP_WPS * 256 / kZpThreads)), zb(kZpThreads);                     // persistent workgroups, a burst at a time
      if (h->dev.ntrks == 9) hipLaunchKernelGGL(k_zeros<9>, zg, zb, 0, st, (const DevCfg *)h->d_dev, d_rows, (long long)nrows, (long long)row_base, d_bursts, scratch, d_counts, d_events);
      else if (h->dev.ntrks == 7) hipLaunchKernelGGL(k_zeros<7>, zg, zb, 0, st, (const DevCfg *)h->d_dev, d_rows, (long long)nrows, (long long)row_base, d_bursts, scratch, d_counts, d_events);
      else hipLaunchKernelGGL(k_zeros<0>, zg, zb, 0, st, (const DevCfg *)h->d_dev, d_rows, (long long)nrows, (long long)row_base, d_bursts, scratch, d_counts, d_events);
      t1(kTZeros); }
   else if (h->dev.dense_path) {                                      // PE, GCR peak detection: sub-segment lists, then a lane per chain (rtfe_dense.hip)
      const int dstop = h->dense_stop;      // (debugging: 1 = stop behind k_dseg, 2 = behind k_dchain)
      if (dstop < 2) { skip_rest(); return launch_check("rtfe_scan"); }
      t0(kTDchain);
      const int ds_order = h->ds_order;      // (0: the chains in burst order)
      if (ds_order) hipLaunchKernelGGL(k_dorder, dim3(1), dim3(1024), 0, st, (const rtfe_burst *)d_bursts, (const BurstScratch *)scratch, ctlp, (long long)nrows);
      // (LDS per wave decides how many chains run side by side - the kernel holds ~240 VGPRs, eight waves per CU: the literal rows' cache is a chunk and the
      //  widest window + 2, not the 88 rows the widest window the library accepts would need)
      int wmax = 1;
      for (int i = 0; i < h->dev.nparm; ++i) if (h->dev.parm[i].W > wmax) wmax = h->dev.parm[i].W;
      const int dcache = wmax + 2 + kDcChunk < kDcCache ? ((wmax + 2 + kDcChunk + 7) & ~7) : kDcCache;
      const int dc_wgs = h->dchain_wgs;      // (waves per CU that take chains from the queue)
      hipLaunchKernelGGL(k_dchain, dim3(h->num_cus * (dc_wgs >= 1 && dc_wgs <= 64 ? dc_wgs : 16)), dim3(64), (size_t)(h->dev.ds_slot < 144 ? 144 : h->dev.ds_slot) * 64 + (size_t)dcache * 64 * 2, st, (const DevCfg *)h->d_dev, d_rows, (long long)nrows, (long long)row_base, (const rtfe_burst *)d_bursts,
                         scratch, ctlp, d_counts, d_events, (const unsigned char *)deadp, (const unsigned char *)slotp, dtiles, ds_order);
      hipLaunchKernelGGL(k_publish, dim3(h->num_cus < 64 ? h->num_cus : 64), dim3(256), 0, st, h->d_dev, (long long)nrows, d_bursts, scratch, ctlp);
      t1(kTDchain);
      if (dstop < 3) { skip_rest(); return launch_check("rtfe_scan"); }
      t0(kTDecode);
      hipLaunchKernelGGL(k_decode, dim3(dgrid), dim3(threads), h->lds_bytes, st, h->d_dev, d_rows, (long long)nrows,
                         (long long)row_base, d_bursts, scratch, d_counts, d_events, 0xffffffffu, 0, 0, (int)kDecodeRedo, ctlp);
      t1(kTDecode); }
   else {                                                             // differentiated peaks, density detection (and PE / GCR with RTFE_DENSE_PATH=0)
      t0(kTDecode);
      hipLaunchKernelGGL(k_decode, dim3(dgrid), dim3(threads), h->lds_bytes, st, h->d_dev, d_rows, (long long)nrows,
                         (long long)row_base, d_bursts, scratch, d_counts, d_events, 0xffffffffu, 0, 0, (int)kDecodeAll, ctlp);
      t1(kTDecode); }
   skip_rest();
   return launch_check("rtfe_scan"); }

extern "C" int rtfe_set_graphs(rtfe_handle *h, int enable) {
   if (!h) return fail(-1, "null argument");
   h->graphs = enable != 0;
   return 0; }

extern "C" int rtfe_scan(rtfe_handle *h, const int16_t *d_rows, int64_t nrows, int64_t own_rows, int64_t row_base, int first_is_tape_start,
                         void *d_workspace, size_t workspace_bytes,
                         rtfe_burst *d_bursts, int64_t max_bursts, int32_t *d_nbursts,
                         uint32_t *d_counts, rtfe_event *d_events, int64_t event_capacity, void *stream) {
   // (per-kernel timing records events between the launches, the debug counters are read by tools between scans, the legacy default stream cannot be
   //  captured: those scans are launched directly)
   if (!h || !h->graphs || h->timing || h->dev.debug || !stream)
      return scan_launch(h, d_rows, nrows, own_rows, row_base, first_is_tape_start, d_workspace, workspace_bytes, d_bursts, max_bursts, d_nbursts, d_counts, d_events, event_capacity, stream);
   hipStream_t st = (hipStream_t)stream;
   const unsigned long long key[13] = {(unsigned long long)(uintptr_t)d_rows, (unsigned long long)nrows, (unsigned long long)own_rows, (unsigned long long)row_base, (unsigned long long)first_is_tape_start,
                                       (unsigned long long)(uintptr_t)d_workspace, (unsigned long long)workspace_bytes, (unsigned long long)(uintptr_t)d_bursts, (unsigned long long)max_bursts,
                                       (unsigned long long)(uintptr_t)d_nbursts, (unsigned long long)(uintptr_t)d_counts, (unsigned long long)(uintptr_t)d_events, (unsigned long long)event_capacity};
   for (int i = 0; i < kGraphCache; ++i) {
      rtfe_handle::GraphEnt &e = h->gcache[i];
      if (e.exec && memcmp(e.key, key, sizeof key) == 0) {
         e.stamp = ++h->gstamp;
         if (hipGraphLaunch(e.exec, st) != hipSuccess) return fail(-30, "rtfe_scan: hipGraphLaunch: %s", hipGetErrorString(hipGetLastError()));
         return 0; } }
   int slot = 0;                                                       // an empty place, else the least recently used
   for (int i = 0; i < kGraphCache; ++i) {
      if (!h->gcache[i].exec) { slot = i; break; }
      if (h->gcache[i].stamp < h->gcache[slot].stamp) slot = i; }
   // a new set of arguments: its launches captured (nothing runs while they are), instantiated, launched - into the least recently used place
   ensure_side(h);
   if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      (void)hipGetLastError(); h->graphs = 0;
      return scan_launch(h, d_rows, nrows, own_rows, row_base, first_is_tape_start, d_workspace, workspace_bytes, d_bursts, max_bursts, d_nbursts, d_counts, d_events, event_capacity, stream); }
   const int rc = scan_launch(h, d_rows, nrows, own_rows, row_base, first_is_tape_start, d_workspace, workspace_bytes, d_bursts, max_bursts, d_nbursts, d_counts, d_events, event_capacity, stream);
   hipGraph_t g = nullptr;
   const hipError_t ec = hipStreamEndCapture(st, &g);
   if (rc != 0) { if (g) (void)hipGraphDestroy(g); (void)hipGetLastError(); return rc; }      // (an argument the scan refuses: nothing was enqueued, the message stands)
   hipGraphExec_t ex = nullptr;
   if (ec != hipSuccess || !g || hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
      // (a runtime that cannot capture this sequence: the scan is launched directly, now and from here on)
      if (g) (void)hipGraphDestroy(g);
      (void)hipGetLastError(); h->graphs = 0;
      return scan_launch(h, d_rows, nrows, own_rows, row_base, first_is_tape_start, d_workspace, workspace_bytes, d_bursts, max_bursts, d_nbursts, d_counts, d_events, event_capacity, stream); }
   (void)hipGraphDestroy(g);
   rtfe_handle::GraphEnt &e = h->gcache[slot];
   if (e.exec) (void)hipGraphExecDestroy(e.exec);
   memcpy(e.key, key, sizeof key); e.exec = ex; e.stamp = ++h->gstamp;
   if (hipGraphLaunch(ex, st) != hipSuccess) return fail(-30, "rtfe_scan: hipGraphLaunch: %s", hipGetErrorString(hipGetLastError()));
   return 0; }

// Synchronous (copies three words back): what the last rtfe_scan on this workspace did.
extern "C" int rtfe_pack_events(rtfe_handle *h, const rtfe_burst *d_bursts, const int32_t *d_nbursts, int64_t max_bursts, const uint32_t *d_counts,
                                const rtfe_event *d_events, rtfe_event *d_packed, uint64_t packed_capacity, rtfe_pack_entry *d_plan, void *stream) {
   if (!h || !d_bursts || !d_nbursts || !d_counts || !d_events || !d_packed || !d_plan || max_bursts < 1) return fail(-2, "rtfe_pack_events: null argument");
   hipStream_t st = (hipStream_t)stream;
   const int lists = h->cfg.nparmsets * h->cfg.ntrks;
   hipLaunchKernelGGL(k_pack_plan, dim3(1), dim3(1024), 0, st, d_bursts, d_nbursts, (long long)max_bursts, d_counts, lists, d_plan);
   hipLaunchKernelGGL(k_pack_copy, dim3((unsigned)(h->num_cus * 8)), dim3(256), 0, st, d_bursts, d_nbursts, (long long)max_bursts, d_counts, lists, (const rtfe_pack_entry *)d_plan,
                      reinterpret_cast<const uint4 *>(d_events), reinterpret_cast<uint4 *>(d_packed), (unsigned long long)packed_capacity);
   const hipError_t e = hipGetLastError();
   if (e != hipSuccess) return fail(-30, "rtfe_pack_events: %s", hipGetErrorString(e));
   return 0; }

extern "C" int rtfe_find_end_mark(rtfe_handle *h, const int16_t *d_rows, int64_t nrows, int64_t *d_first, void *stream) {
   if (!h || !d_rows || !d_first || nrows < 0) return fail(-2, "rtfe_find_end_mark: bad argument");
   hipStream_t st = (hipStream_t)stream;
   hipLaunchKernelGGL(k_end_mark_init, dim3(1), dim3(1), 0, st, reinterpret_cast<long long *>(d_first));
   if (nrows > 0) {
      long long blocks = (nrows + 255) / 256;
      if (blocks > (long long)h->num_cus * 16) blocks = (long long)h->num_cus * 16;
      hipLaunchKernelGGL(k_end_mark, dim3((unsigned)blocks), dim3(256), 0, st, d_rows, (long long)nrows, h->cfg.ntrks, reinterpret_cast<long long *>(d_first)); }
   const hipError_t e = hipGetLastError();
   if (e != hipSuccess) return fail(-30, "rtfe_find_end_mark: %s", hipGetErrorString(e));
   return 0; }

extern "C" int rtfe_reset_floor(rtfe_handle *h, void *stream) {
   if (!h) return fail(-1, "null argument");
   hipLaunchKernelGGL(k_reset_floor, dim3(1), dim3(1), 0, (hipStream_t)stream, h->d_dev);
   return launch_check("rtfe_reset_floor"); }

extern "C" int rtfe_scan_stats(rtfe_handle *h, const void *d_workspace, int64_t *out) {
   if (!h || !d_workspace || !out) return fail(-1, "null argument");
   BurstScratch sc;
   if (hipMemcpy(&sc, d_workspace, sizeof sc, hipMemcpyDeviceToHost) != hipSuccess) return fail(-44, "hipMemcpy failed");
   out[0] = sc.nbursts; out[1] = sc.seg_failed; out[2] = (int64_t)sc.scr[3]; out[3] = (int64_t)sc.dbg[0]; out[4] = (int64_t)sc.dbg[1];
   for (int i = 0; i < 8; ++i) out[5 + i] = (int64_t)sc.why[i];
   out[21] = (int64_t)sc.min_height_key;      // 0x7fffffff - float bits of the smallest learned v_avg_height, 0: none (the Python binding turns it back)
   {  float fl = 0;                                // the floor the handle's NEXT scan screens against (k_adapt_floor has moved it behind this one), as float bits
      if (hipMemcpy(&fl, &h->d_dev->floor_now, sizeof fl, hipMemcpyDeviceToHost) != hipSuccess) return fail(-44, "hipMemcpy failed");
      uint32_t fb; memcpy(&fb, &fl, 4); out[22] = (int64_t)fb;
      memcpy(&fb, &sc.floor_used, 4); out[23] = (int64_t)fb; }      // the floor this scan's screen was built for
   for (int i = 0; i < 8; ++i) out[13 + i] = (int64_t)((h->dev.debug == 4 || h->dev.debug == 6 || h->dev.debug == 8 || h->dev.debug == 9) ? sc.dbg2[i] : sc.scr[i]);      // (RTFE_DEBUG=4: k_gain's cycle counters instead)      // RTFE_DEBUG=3: k_sift cycles per phase (copy, dense, owners, record bytes, hard candidates, rounds, rounds with one, tiles)
   return 0; }

extern "C" int rtfe_scan_exact(rtfe_handle *h, const int16_t *d_rows, int64_t nrows, int64_t row_base,
                               int64_t reset_row, int64_t end_row, uint32_t parmset_mask, int screen_off,
                               void *d_workspace, size_t workspace_bytes,
                               rtfe_burst *d_burst, uint32_t *d_counts, rtfe_event *d_events, int64_t event_capacity,
                               void *stream) {
   if (!h || !d_rows || !d_workspace || !d_burst || !d_counts || !d_events) return fail(-1, "null argument");
   if (((uintptr_t)d_rows & 15) != 0) return fail(-31, "d_rows must be 16-byte aligned");
   if (workspace_bytes < kScratchBytes) return fail(-32, "workspace too small");
   if (reset_row < 0 || reset_row >= nrows || end_row <= reset_row) return fail(-34, "bad row range");
   if (end_row > nrows) end_row = nrows;
   hipStream_t st = (hipStream_t)stream;
   BurstScratch *scratch = reinterpret_cast<BurstScratch *>(d_workspace);
   const unsigned long long cap = (unsigned long long)(event_capacity / ((int64_t)h->dev.nparm * h->dev.ntrks));
   hipLaunchKernelGGL(k_setup_exact, dim3(1), dim3(1), 0, st, d_burst, scratch, (long long)reset_row, (long long)end_row,
                      cap > 0xffffffffull ? 0xffffffffull : cap);
   const int nwalk = h->dev.nparm * h->dev.ntrks;
   hipLaunchKernelGGL(k_decode, dim3(1), dim3(nwalk <= 64 ? 64 : (nwalk <= 128 ? 128 : 256)), h->lds_bytes, st, h->d_dev, d_rows, (long long)nrows,
                      (long long)row_base, d_burst, scratch, d_counts, d_events, parmset_mask, screen_off, 1, (int)kDecodeAll, (BurstCtl *)nullptr);
   return launch_check("rtfe_scan_exact"); }

// ---- Whirlwind (include/rt_frontend.h) ----
extern "C" void rtfe_ww_initial_state(rtfe_ww_track *tracks, int ntrks) {
   memset(tracks, 0, sizeof(rtfe_ww_track) * (size_t)ntrks);
   for (int t = 0; t < ntrks; ++t) { tracks[t].agc_gain = 1.0f; tracks[t].v_avg_height = 4.0f; } }

extern "C" int rtfe_ww_scan(rtfe_handle *h, const int16_t *d_rows, int64_t nrows, int64_t row_base, int64_t first_row, int64_t nscan, int64_t seed_row0,
                            const rtfe_ww_track *d_state_in, rtfe_ww_track *d_state_out, uint32_t *d_counts, rtfe_event *d_events, int64_t event_capacity,
                            uint32_t *d_flags, void *stream) {
   if (!h || !d_rows || !d_state_in || !d_state_out || !d_counts || !d_events || !d_flags) return fail(-1, "null argument");
   if (h->dev.mode != RTFE_WW) return fail(-40, "rtfe_ww_scan: the handle was not made for mode RTFE_WW");
   if (h->dev.find_zeros || h->dev.differentiate || h->dev.maxskew > 0 || h->dev.nparm != 1) return fail(-41, "rtfe_ww_scan: peak detection, one parameter set, deskew delays in the state (not in the configuration)");
   if (h->dev.parm[0].W > kWwRing) return fail(-42, "window wider than the state's ring");
   if (first_row < 0 || nscan <= 0 || first_row >= nrows || seed_row0 > first_row || event_capacity < 1) return fail(-43, "bad row range");
   hipLaunchKernelGGL(k_ww, dim3(1), dim3(64), 0, (hipStream_t)stream, (const DevCfg *)h->d_dev, d_rows, (long long)nrows, (long long)row_base,
                      (long long)first_row, (long long)nscan, (long long)seed_row0, d_state_in, d_state_out, d_counts, d_events, (long long)event_capacity, d_flags);
   return launch_check("rtfe_ww_scan"); }
