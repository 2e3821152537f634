/* TEST INFRASTRUCTURE — see oracle_fe.c. */
#ifndef ORACLE_FE_H
#define ORACLE_FE_H
#include "rt_decode.h"

struct ofe_det {               /* detector-private per-track state (src/decoder.h:196,198,212-226) */
   float v_last_raw, v_prev;
   uint8_t zerocross_up_pending, zerocross_dn_pending;
   double t_firstzero, t_lastzero;
   float pkww_v[RT_PKWW_MAX_WIDTH];
   float pkww_minv, pkww_maxv;
   int   pkww_left, pkww_right, pkww_countdown;
   int   last_left_distance, last_adj2;    /* what refine_peak decided for the latest peak (for event dumps) */
};
struct ofe_skew {              /* src/decoder.c:227-231 */
   float vdelayed[RT_MAXSKEWSAMP];
   int ndx_next, slots_filled;
};
struct ofe {
   struct rt_dec *dec;
   const int16_t *rows; int64_t nrows; int nheads;
   float maxvolts;
   int64_t tstart_ns, timenow_ns;
   int64_t pos, saved_pos, saved_time_ns; double saved_time;
   int head_to_trk[RT_MAXTRKS];
   int invert;
   int skew_delaycnt[RT_MAXTRKS];
   int pkww_width;
   long long lines_in, numsamples;
   int64_t eob_row;            /* row whose processing ended the current attempt's block (-1: not yet) */
   int fatal;
   struct ofe_det det[RT_MAXTRKS];
   struct ofe_skew skew[RT_MAXTRKS];
   void (*on_attempt_start)(struct ofe *fe, int64_t first_row);
   void (*on_attempt_end)(struct ofe *fe);
   void *user;
};
struct ofe *ofe_new(struct rt_dec *dec, const int16_t *rows, int64_t nrows, int nheads, float maxvolts, int64_t tstart_ns);
void ofe_free(struct ofe *fe);
int  ofe_readblock(void *ctx, int retry);
void ofe_save_pos(void *ctx);
void ofe_restore_pos(void *ctx);
#endif
