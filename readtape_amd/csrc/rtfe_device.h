// rtfe_device.h — device-side configuration shared by the kernels and the host API (gfx950 only).
#pragma once
#include <stdint.h>
#include "rt_frontend.h"

namespace rtfe {

constexpr int kChunkRows   = 64;     // granularity of the quiet map (rows per bit)
constexpr int kMaxTileRows = 2048;   // upper bound of DevCfg::tile_rows (rows per LDS tile of the decode kernel)
constexpr int kHaloRows    = 160;    // rows kept in front of a tile: >= 2*W + max skew + 8  (W<=50, skew<=50)
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
};

struct DevScreen {
   int W;
   int rise_i;          // candidate if (max - edge) > rise_i on both edges          (int16 units)
   int minpk_i;         // ... and max > minpk_i (top) / min < -minpk_i (bottom); -1 = no min_peak test
};

struct DevCfg {
   int   mode, ntrks, invert, nparm, nscreens;
   int   find_zeros;              // -zeros: zero-crossing detector instead of the peak detector (src/decoder.c:863-865)
   int   differentiate;           // -differentiate (only with -zeros on the device: src/decoder.c:654-683)
   int   samples_per_bit;         // (int)(1/(bpi*ips*sample_deltat)), src/readtape.c:1402
   int   zc_peak_i;               // smallest positive int16 code c with volt(c) > ZEROCROSS_PEAK (0.2 V, src/decoder.h:138)
   int   head_to_trk[RTFE_MAXTRKS];   // TBIN column -> track (src/readtape.c:1419)
   int   skew[RTFE_MAXTRKS];
   int   maxskew;
   float maxvolts;
   float sample_deltat;           // (float)tdelta_ns / 1e9f (src/readtape.c:1345)
   long long tdelta_ns, tstart_ns;
   int   quiet_i;                 // |x| <= quiet_i on every track  <=> row is "quiet"
   int   gap_chunks;              // quiet chunks that make an inter-block zone
   float cap_frac;                // event capacity per track as a fraction of burst length
   int   tile_rows;               // rows per LDS tile (multiple of 64, kMarginRows..kMaxTileRows)
   float lsb_per_volt;            // 32767 / maxvolts, for the walkers' integer guard bands
   int   run_cap;                 // candidate-run records per (screen, track) per tile (LDS)
   int   rec_cap;                 // deferred-event records per walker per tile (LDS)
   int   debug;                   // RTFE_DEBUG=1: per-phase cycle counters in the workspace (tools/ only)
   DevParm   parm[RTFE_MAXPARMSETS];
   DevScreen screen[kMaxScreens];
};

// ---- what the dense screen pass (k_screen) leaves in HBM for the sequential pass (k_decode) ----
struct PackedRun {             // one candidate run of one (screen, track) of one tile: 20 bytes
   uint16_t n_s;               // first candidate row, tile-relative
   uint16_t len;               // candidate rows in the run
   int16_t  m, prev, next;     // the extreme and its two neighbours (int16 codes)
   uint8_t  ld;                // left_distance of the extreme at row n_s
   uint8_t  kindfast;          // bit 7: 0 top / 1 bottom; bits 0-3: rows n_s+k decidable from marg[k]
   int16_t  marg[4];
};
struct TileDir {               // per (tile, screen, track): 8 bytes
   uint32_t offset;            // index of the first PackedRun in the pool
   uint16_t count;             // 0xFFFF: not available (pool full / more runs than the tile list holds)
   int16_t  last_rescan;       // tile-relative row of the last forced rescan ("window maximum leaves"), -1 if none
};

}  // namespace rtfe
