/* TEST INFRASTRUCTURE — not product code.
 *
 * Event-dump shim linked into the *unmodified* reference objects (oracle/Makefile, target
 * _ref/readtape_evt) with  -Wl,--wrap=<sym>.  It records what the reference's analog front end
 * hands to its per-format bit decoders, so that the restatement in oracle/ and the HIP path can be
 * checked event for event.
 *
 * Wrapped seams (all are calls that cross translation units in the reference, so --wrap sees them):
 *   {nrzi,pe,gcr,ww}_{top,bot}(struct trkstate_t*)   called from src/decoder.c:583-586, 601-604
 *   init_trackstate()                                called from src/readtape.c:1665,1693,1760,1859
 *   ww_init_blockstate()                             called from src/readtape.c:1692,1759
 *
 * Output: binary records appended to the file named by $RT_EVENT_DUMP (nothing if unset).
 * Record layout (little endian, 48 bytes) — see tools/refdump.py for the reader:
 *   u32 kind      0=top 1=bot 2=block-attempt-start
 *   u32 trk
 *   i32 peakcount (already incremented by process_transition, src/decoder.c:561)
 *   i32 parmset   (block.parmset)
 *   f64 t_peak    (t_top or t_bot; for kind 2: timenow)
 *   i64 timenow_ns  at the call. For kind 0/1 this is one tdelta AFTER the sample being processed
 *                   (src/readtape.c:1424 advances it before process_sample); for kind 2 it is the
 *                   time of the first sample the attempt will read.
 *   f32 v_peak    (v_top or v_bot)
 *   f32 agc_gain
 *   f32 v_avg_height
 *   u32 pad
 */
#include "decoder.h"

static FILE *dumpf;
static int dump_tried;

struct rec {
   uint32_t kind, trk; int32_t peakcount, parmset;
   double t_peak; int64_t timenow_ns;
   float v_peak, agc_gain, v_avg_height; uint32_t pad; };

static void put(uint32_t kind, struct trkstate_t *t) {
   if (!dump_tried) {
      const char *name = getenv("RT_EVENT_DUMP");
      dump_tried = 1;
      if (name) dumpf = fopen(name, "wb"); }
   if (!dumpf) return;
   struct rec r;
   memset(&r, 0, sizeof r);
   r.kind = kind;
   r.parmset = block.parmset;
   r.timenow_ns = timenow_ns;
   if (t) {
      r.trk = (uint32_t)t->trknum;
      r.peakcount = t->peakcount;
      r.t_peak = kind == 0 ? t->t_top : t->t_bot;
      r.v_peak = kind == 0 ? t->v_top : t->v_bot;
      r.agc_gain = t->agc_gain;
      r.v_avg_height = t->v_avg_height; }
   else r.t_peak = timenow;
   fwrite(&r, sizeof r, 1, dumpf);
   fflush(dumpf); }

#define WRAP_PEAK(name, kind) \
   void __real_##name(struct trkstate_t *t); \
   void __wrap_##name(struct trkstate_t *t) { put(kind, t); __real_##name(t); }
WRAP_PEAK(nrzi_top, 0) WRAP_PEAK(nrzi_bot, 1)
WRAP_PEAK(pe_top, 0)   WRAP_PEAK(pe_bot, 1)
WRAP_PEAK(gcr_top, 0)  WRAP_PEAK(gcr_bot, 1)
WRAP_PEAK(ww_top, 0)   WRAP_PEAK(ww_bot, 1)

void __real_init_trackstate(void);
void __wrap_init_trackstate(void) { put(2, NULLP); __real_init_trackstate(); }
void __real_ww_init_blockstate(void);
void __wrap_ww_init_blockstate(void) { put(2, NULLP); __real_ww_init_blockstate(); }
