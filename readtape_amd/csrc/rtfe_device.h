// rtfe_device.h — device-side configuration shared by the kernels and the host API (gfx950 only).
#pragma once
#include <stdint.h>
#include "rt_frontend.h"

namespace rtfe {

typedef unsigned long long u64;

constexpr int kChunkRows   = 64;     // granularity of the quiet map (rows per bit)
constexpr int kMaxTileRows = 2048;   // upper bound of DevCfg::tile_rows (rows per LDS tile of the decode kernel)
constexpr int kMaxHaloRows = 176;    // upper bound of DevCfg::halo_rows = kScreenHalo + W + 1 + max skew, rounded up to 8  (W<=50, skew<=50)
constexpr int kMarginRows  = 256;    // head/tail tile length at a burst boundary (multiple of 64)
constexpr int kStrip       = 8;      // samples per screen strip (one bitmap byte)
constexpr int kDecodeThreads = 256;
constexpr int kMaxScreens  = 4;      // distinct window widths handled in one scan

struct DevParm {
   int   W;             // pkww_width (src/readtape.c:1456)
   float rise;          // PARM.pkww_rise
   float min_peak;      // PARM.min_peak
   float agc_alpha;
   int   agc_window;
   float t_clkwindow;   // PE: clkavg.t_bitspaceavg / 2 * clk_factor (src/decoder.c:449)
   int   screen;        // index into DevCfg::screen
   float screen_rise_v; // the screen's loosest thresholds in volts (for the underflow check)
   float screen_minpk_v;
   int   seg_warm;      // records a segment of the chains' steady stretch starts early (the alpha filter forgets its start value: (1 - alpha)^n < 2^-26)
};

struct DevScreen {
   int W;
   int rise_i;          // candidate if (max - edge) > rise_i on both edges          (int16 units)
   int minpk_i;         // ... and max > minpk_i (top) / min < -minpk_i (bottom); -1 = no min_peak test
   int sure_i;          // k_sift: a row whose margin is >= sure_i passes the rise test for every threshold k_gain accepts without asking
};

struct DevCfg {
   int   mode, ntrks, invert, nparm, nscreens;
   int   find_zeros;              // -zeros: zero-crossing detector instead of the peak detector (src/decoder.c:863-865)
   int   agc_off;                 // density detection (bpi unknown, src/decoder.c:578,596): no decoder runs, so nothing ever
                                  // adjusts the AGC or the baseline; the window is 8 samples (src/readtape.c:1457)
   int   differentiate;           // -differentiate (only with -zeros on the device: src/decoder.c:654-683)
   int   samples_per_bit;         // (int)(1/(bpi*ips*sample_deltat)), src/readtape.c:1402
   int   zc_peak_i;               // smallest positive int16 code c with volt(c) > ZEROCROSS_PEAK (0.2 V, src/decoder.h:138)
   int   head_to_trk[RTFE_MAXTRKS];   // TBIN column -> track (src/readtape.c:1419)
   int   trk_to_head[RTFE_MAXTRKS];   // ... and back: the column of the sample tile that holds a track
   int   skew[RTFE_MAXTRKS];
   int   maxskew;
   float maxvolts;
   float sample_deltat;           // (float)tdelta_ns / 1e9f (src/readtape.c:1345)
   long long tdelta_ns, tstart_ns;
   int   quiet_i;                 // |x| <= quiet_i on every track  <=> row is "quiet"
   int   gap_chunks;              // quiet chunks that make an inter-block zone
   int   zc_parallel;             // -zeros: concurrent sub-segments per tile (0 = one lane per track, sequential)
   int   zc_warm;                 // ... rows a sub-segment starts early from a fresh state (multiple of 8, <= 64; RTFE_ZC_WARM)
   int   tail_rows;               // a burst's walkers stop this many rows into the next zone (the block decoders have long ended
                                  // the block by then; an attempt that has not falls back to an exact rescan in the replay)
   float cap_frac;                // event capacity per track as a fraction of burst length
   int   tile_rows;               // rows per LDS tile (multiple of 64, kMarginRows..kMaxTileRows)
   int   halo_rows;               // rows kept in front of a tile: kScreenHalo + widest window + 1 + max skew, rounded up to 8
   int   ldw;                     // rows of the LDS sample tile (halo_rows + tile_rows + 8)
   float lsb_per_volt;            // 32767 / maxvolts, for the walkers' integer guard bands
   int   rec_cap;                 // deferred-event records per walker per tile (LDS)
   int   debug;                   // RTFE_DEBUG=1: per-phase cycle counters in the workspace (tools/ only)
   int   cut;                     // RTFE_CUT: k_sift stops after a phase (timing experiments, tools/ only; results are then garbage)
   int   pk_plain;                // k_sift_s: the build without -invert, cut-offs, counters and with deferred copy-out (host side only: picks the instantiation)
   int   peak_path;               // k_sift -> k_gain -> k_emit serve rtfe_scan (peak detection on the undifferentiated signal)
   int   pk_hl, pk_hr;            // rows kept in front of / behind a k_sift tile in LDS (multiples of 8)
   int   pk_slot;                 // bytes of a pool slot: the list of one (tile, screen, head); multiple of 16
   int   pk_wave_cap;             // candidates of one wave (two heads of a tile) k_sift can list in LDS; beyond: lists unavailable
   int   pk_lds;                  // dynamic LDS bytes of k_sift
   int   pk_fast;                 // k_gain: the steady-state fast path (0: every detection through the general step; tests)
   int   pk_seg_recs;             // records per segment of a chain's steady stretch (k_gain_seg); 0: a chain is one segment
   int   pk_rejoin;               // k_gain (mode 1): a chain that broke inside its segmented stretch re-joins the segments behind the break where the states agree (1)
   int   pk_mar;                  // rows of a record's margin block the walkers read (kPkMar; RTFE_PK_MAR < 4: the rest is made from the samples - tests)
   // the dense sample path (rtfe_dense.hip): k_dseg -> k_dchain, for peak detection where a window holds a top and a bottom (PE, GCR)
   int   dense_path;
   int   nuset;                              // parameter sets that differ in what the FRONT END reads (window, thresholds, AGC flavour; PE: the clock window)
   int   uset_of[RTFE_MAXPARMSETS];          // parameter set -> distinct set
   int   uset_rep[RTFE_MAXPARMSETS];         // distinct set -> its first parameter set
   unsigned uset_mask[RTFE_MAXPARMSETS];     // distinct set -> the parameter sets it stands for (their event regions all receive its events)
   int   ds_pad;                             // rows in front of a k_dseg tile's own rows that only serve as warm-up (multiple of 64)
   int   ds_warm[kMaxScreens];               // rows a sub-segment's lane starts early, per window width (2 W + 16, at least 48)
   int   ds_up;                              // distinct sets of one width k_dseg classifies in one pass over a tile (2 or 3)
   int   ds_cap, ds_slot;                    // records per slot; bytes of a slot (header + records)
   float ds_band_hi, ds_band_lo;             // a sub-segment's band: [ds_band_lo, 1] x ds_band_hi x (its peak-to-peak amplitude / 4)
   int   ds_lean;                            // k_dchain: the steady-state record as straight-line code (0: every record through the general step; tests)
   float ds_quiet_s;                         // a small-signal sub-segment may use the band [its largest margin, infinity) if that starts at or below this scale
   float ds_sfloor;                          // lower end of every band: the scale (v_avg_height / 4) / agc_gain the candidate screen is built for
   // the candidate screen's floor follows the tape (k_adapt_floor, behind a scan of the peak path): floor_now = what the screen thresholds above stand for
   // at the moment (the assumed lower bound of a learned peak height), floor_cfg = what the handle was made with, adapt_floor = 1: it may rise
   float floor_cfg, floor_now;
   int   adapt_floor;
   // ... and is first ESTIMATED from the samples, at the head of a handle's first scan (k_scan_begin): floor_probed = 1 once an estimate or a learned height
   // moved the floor; probe_min / probe_ticket: the probe's reduction over its workgroups (left as they were found: INT_MAX / 0)
   int   floor_probed, probe_min, probe_on;      // (probe_on = 0: RTFE_FLOOR_PROBE=0, round 5's behaviour - tests, experiments)
   unsigned probe_ticket;
   DevParm   parm[RTFE_MAXPARMSETS];
   DevScreen screen[kMaxScreens];
};

// ---- burst hand-over between the kernels of one scan (workspace) ----
enum { kBurstNew = 0, kBurstNeedsFull = 1, kBurstReady = 2, kBurstDone = 3 };
enum { kDecodeAll = 0, kDecodeRedo = 3 };     // kDecodeRedo: whole bursts the record chains gave up (restart / stop rows from k_zones)
struct BurstCtl {              // 32 bytes per burst
   long long reset, stop;      // restart row / first row of the next burst's span
   int       next_tile;        // tile of the tape-global grid to continue with
   int       status;           // kBurst*
   unsigned  bflags;           // burst flags collected so far
   int       pad;
};
constexpr int kScreenHalo = 64;      // rows in front of a tile that the screen also covers (one bitmap word)

// ---- the peak path: k_sift (dense, stateless) -> k_zones -> k_gain (one LANE per (burst, parmset, track)) -> k_emit ----
constexpr int kSfStrip   = 14;       // rows per lane strip: 14 rows x 2 ntrks bytes is an odd number of dwords for 9 tracks (no LDS bank conflicts)
constexpr int kSfTile    = 64 * kSfStrip;      // 896 rows per k_sift tile
constexpr int kSfGroups  = kSfTile / 64;       // quiet-map groups of 64 rows per tile
constexpr int kSfPosBias = 128;      // a record's owner may lie up to kPkBack + W rows in front of its tile
constexpr int kPkBack    = 64;       // rows k_sift looks back for the last forced rescan in front of a bottom candidate
// One candidate RUN = the rows at which one sample (the "owner": the window maximum, or the reference's
// possibly stale window minimum) is what lookfor_peak would test (src/decoder.c:788-805).
//   w0  bits  0-10  pos - tile row0 + kSfPosBias   the owner's row (the run of a stale minimum may be owned by a sample of the tile in front)
//       bit   11    kind  0 top / 1 bottom
//       bits 12-17  f - pos                         first row of the run with a margin above the screen (1 .. W-1)
//       bits 18-21  nlead                           rows f .. f+nlead-1 carry explicit margins (uint16 entries)
//       bits 22-27  nsure                           the next nsure rows all have margins >= DevScreen::sure_i
//       bits 28-31  ntail                           the next ntail rows carry explicit margins again; no row behind them passes the screen
//                   nsure == 63: every row explicit, (nlead << 4 | ntail) of them from f
//   w1  val (int16) | clamp(d(prev), -1, 253) + 1 (8 bits) | clamp(d(next), -1, 253) + 1 (8 bits)       d = |val - neighbour| signed towards "beyond the extreme"
//       0xffff8000: k_sift could not derive the reference's minimum; rows f .. f+nsure-1 are undecidable from the record
// margin of a row = val - max(left edge, right edge) (tops) / min(edges) - val (bottoms), int16 code differences.
// A list holds the records of the candidates of ONE tile in candidate order; their rows may run on into the next tile.
struct PeakRec { uint32_t w0, w1; };
// The pool: one fixed slot of pk_slot bytes per (tile, screen, head), 16 bytes a record: the PeakRec and, behind it, its MARGIN BLOCK - the margins
// (uint16, clamped at 0) of the run's first kPkMar = 4 rows from f on, entry j (row f + j) at block end - 2 (j + 1).  A list that does not fit its
// slot is marked unavailable in the directory.  (Rounds 3 / 4: 8-byte records from the slot's front, one 2-byte entry per lead and tail row from its
// back - variable per record: a second prefix sum and a predicated store per window row in k_sift, an entry reference per record downstream.)
//   w1 == 0xffff8001: a candidate k_sift deferred; w0 = its index in the hard list = its overflow slot (k_sift_hard)
struct SfHard { uint32_t tile; uint16_t pos; uint8_t head, screen; };      // a deferred candidate: tile, row within it, head, screen
constexpr int kSfOvfBytes = 128;     // an overflow slot: int32 records, pad, <= 4 records of 16 bytes (record + margin block) from byte 8
struct PeakDir {               // per (tile, screen, head): 4 bytes
   uint16_t nrec;              // 0xFFFF: not available (capacity)
   uint16_t nent;
};
}  // namespace rtfe
