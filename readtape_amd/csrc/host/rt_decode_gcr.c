/* rt_decode_gcr.c — 6250 BPI GCR bit recovery (placeholder until the GCR row lands).
 * The entry points exist so the library links; calling them reports the block as unusable. */
#include "rt_decode.h"

void rt_gcr_preprocess(struct rt_dec *d) { (void)d; }
void rt_gcr_top(struct rt_dec *d, struct rt_trk *t) { (void)d; (void)t; }
void rt_gcr_bot(struct rt_dec *d, struct rt_trk *t) { (void)d; (void)t; }
void rt_gcr_end_of_block(struct rt_dec *d) {
   if (d->endblock_done) return;
   d->endblock_done = 1;
   d->results[d->parmset].blktype = RT_BS_BADBLOCK; }
