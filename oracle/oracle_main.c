/* TEST INFRASTRUCTURE — command-line driver of the parity oracle (see oracle_fe.c).
 *
 *   oracle_readtape [options] tape.tbin
 *     -nrzi|-pe|-gcr  -ntrks=N -bpi=N -ips=N   override the TBIN header (src/readtape.c:1330-1343)
 *     -zeros -differentiate -invert -correct -m -even -revparity=N
 *     -skew=a,b,..      per-track deskew delays in SAMPLES
 *     -deskew           calibrate the delays on the first blocks (NRZI, GCR, Whirlwind)
 *     -parms=FILE       parameter sets in the reference's .parms grammar
 *     -out=BASE         writes BASE.tap (SIMH) and BASE.log
 *     -evt=FILE         event dump, same 48-byte records as oracle/ref_event_shim.c
 *     -evtend           also dump a kind-3 record when an attempt ends (timenow_ns = time of the next
 *                       row it would read); the reference shim cannot produce these
 *     -time             print front-end wall time and Msamples/s (the bench's cpu_baseline)
 *     -blklimit=N
 *     -subsample=N   use only every Nth row (src/readtape.c:1407: reads N rows, keeps the last; the time base is NOT stretched)
 */
#include "oracle_fe.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

struct rec {
   uint32_t kind, trk; int32_t peakcount, parmset;
   double t_peak; int64_t timenow_ns;
   float v_peak, agc_gain, v_avg_height; uint32_t pad; };

static FILE *evtf;
static int evt_ends;
static struct ofe *g_fe;

static void on_transition(struct rt_dec *d, struct rt_trk *t, int is_top, void *user) {
   (void)user;
   if (!evtf) return;
   struct rec r; memset(&r, 0, sizeof r);
   r.kind = is_top ? 0 : 1;
   r.trk = (uint32_t)t->trknum;
   r.peakcount = t->peakcount;
   r.parmset = d->parmset;
   r.t_peak = is_top ? t->t_top : t->t_bot;
   r.timenow_ns = g_fe->timenow_ns;
   r.v_peak = is_top ? t->v_top : t->v_bot;
   r.agc_gain = t->agc_gain;
   r.v_avg_height = t->v_avg_height;
   fwrite(&r, sizeof r, 1, evtf); }

static void on_attempt(struct rt_dec *d, void *user) {
   (void)user;
   if (!evtf) return;
   struct rec r; memset(&r, 0, sizeof r);
   r.kind = 2;
   r.parmset = d->parmset;
   r.t_peak = d->timenow;
   r.timenow_ns = g_fe->timenow_ns;
   fwrite(&r, sizeof r, 1, evtf); }

static void on_attempt_end(struct ofe *fe) {
   if (!evtf || !evt_ends) return;
   struct rec r; memset(&r, 0, sizeof r);
   r.kind = 3;
   r.trk = (uint32_t)(fe->eob_row >= 0 ? fe->pos - fe->eob_row : 1);   /* rows from the block-ending row to the next attempt's first row */
   r.parmset = fe->dec->parmset;
   r.peakcount = (int32_t)fe->dec->results[fe->dec->parmset].blktype;
   r.t_peak = fe->dec->timenow;
   r.timenow_ns = fe->timenow_ns;
   fwrite(&r, sizeof r, 1, evtf); }

static unsigned char *slurp(const char *path, size_t *len) {
   FILE *f = fopen(path, "rb");
   if (!f) return NULL;
   fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
   unsigned char *b = (unsigned char *)malloc((size_t)n + 16);
   if (fread(b, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(b); return NULL; }
   fclose(f); *len = (size_t)n; return b; }

static uint32_t rd32(const unsigned char *p) { return p[0] | p[1] << 8 | p[2] << 16 | (uint32_t)p[3] << 24; }
static float rdf(const unsigned char *p) { uint32_t u = rd32(p); float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char **argv) {
   struct rt_options opt; memset(&opt, 0, sizeof opt);
   opt.specified_parity = 1;
   const char *infile = NULL, *outbase = NULL, *evtname = NULL, *parmfile = NULL, *skewarg = NULL, *orderarg = NULL;
   int ntrks_arg = 0, invert = 0, timing = 0, blklimit = 0x7fffffff, subsample = 1, deskew = 0;
   float bpi_arg = -1, ips_arg = -1;
   int mode_arg = 0;
   for (int i = 1; i < argc; ++i) {
      const char *a = argv[i];
      if (!strcmp(a, "-nrzi")) mode_arg = RT_NRZI;
      else if (!strcmp(a, "-pe")) mode_arg = RT_PE;
      else if (!strcmp(a, "-gcr")) mode_arg = RT_GCR;
      else if (!strcmp(a, "-whirlwind")) { mode_arg = RT_WW; bpi_arg = 100; }          /* src/readtape.c:947-948 */
      else if (!strcmp(a, "-fluxdir=pos")) opt.ww_fluxdir = RT_FLUX_POS;
      else if (!strcmp(a, "-fluxdir=neg")) opt.ww_fluxdir = RT_FLUX_NEG;
      else if (!strcmp(a, "-fluxdir=auto")) opt.ww_fluxdir = RT_FLUX_AUTO;
      else if (!strcmp(a, "-reverse")) opt.ww_reverse = 1;
      else if (!strncmp(a, "-ntrks=", 7)) ntrks_arg = atoi(a + 7);
      else if (!strncmp(a, "-bpi=", 5)) bpi_arg = (float)atof(a + 5);
      else if (!strncmp(a, "-ips=", 5)) ips_arg = (float)atof(a + 5);
      else if (!strcmp(a, "-zeros")) opt.find_zeros = 1;
      else if (!strcmp(a, "-differentiate")) opt.do_differentiate = 1;
      else if (!strcmp(a, "-invert")) invert = 1;
      else if (!strcmp(a, "-correct")) opt.do_correction = 1;
      else if (!strcmp(a, "-m")) opt.multiple_tries = 1;
      else if (!strcmp(a, "-even")) opt.specified_parity = 0;
      else if (!strncmp(a, "-revparity=", 11)) opt.revparity = atoi(a + 11);
      else if (!strcmp(a, "-v")) opt.verbose = 1;
      else if (!strcmp(a, "-tap")) opt.tap_format = 1;
      else if (!strcmp(a, "-time")) timing = 1;
      else if (!strncmp(a, "-skew=", 6)) skewarg = a + 6;
      else if (!strcmp(a, "-deskew")) deskew = 1;
      else if (!strncmp(a, "-order=", 7)) orderarg = a + 7;
      else if (!strncmp(a, "-parms=", 7)) parmfile = a + 7;
      else if (!strncmp(a, "-out=", 5)) outbase = a + 5;
      else if (!strncmp(a, "-evt=", 5)) evtname = a + 5;
      else if (!strcmp(a, "-evtend")) evt_ends = 1;
      else if (!strncmp(a, "-blklimit=", 10)) blklimit = atoi(a + 10);
      else if (!strncmp(a, "-subsample=", 11)) { subsample = atoi(a + 11); if (subsample < 1) subsample = 1; }
      else if (a[0] == '-') { fprintf(stderr, "unknown option %s\n", a); return 2; }
      else infile = a; }
   if (!infile) { fprintf(stderr, "usage: oracle_readtape [options] tape.tbin\n"); return 2; }
   opt.tap_format = 1;

   size_t len; unsigned char *buf = slurp(infile, &len);
   if (!buf || len < 256 || memcmp(buf, "TBINHDR", 8)) { fprintf(stderr, "not a .tbin file: %s\n", infile); return 2; }
   uint32_t flags = rd32(buf + 204), ntrks = rd32(buf + 208), tdelta = rd32(buf + 212);
   float maxvolts = rdf(buf + 216);
   uint32_t mode = rd32(buf + 228); float bpi = rdf(buf + 232), ips = rdf(buf + 236);
   size_t off = 240;
   if (flags & 2) off += 28;
   if (memcmp(buf + off, "DAT", 4)) { fprintf(stderr, "missing DAT tag\n"); return 2; }
   uint64_t tstart; memcpy(&tstart, buf + off + 8, 8);
   off += 16;
   opt.mode = mode_arg ? (enum rt_mode)mode_arg : (enum rt_mode)mode;
   opt.ntrks = ntrks_arg ? ntrks_arg : (int)ntrks;
   opt.bpi = bpi_arg >= 0 ? bpi_arg : bpi;
   opt.ips = ips_arg >= 0 ? ips_arg : ips;
   if (opt.ips == 0) opt.ips = 50;
   if (opt.mode == RT_GCR) opt.bpi = 9042;                   /* src/readtape.c:1652-1654 */
   int nheads = (int)ntrks;
   const int16_t *rows = (const int16_t *)(buf + off);       /* off is even: 256 or 284 */
   int64_t navail = (int64_t)((len - off) / 2 / (size_t)nheads), nrows = navail;
   for (int64_t i = 0; i < navail; ++i) if (rows[i * nheads] == -32768) { nrows = i; break; }
   if ((len - off) / 2 > (size_t)(navail * nheads) && nrows == navail) {
      /* the end marker is the lone int16 after the last full row */ }

   if (subsample > 1) {                                       /* src/readtape.c:1407-1414: of every `subsample` rows the last one is used */
      const int64_t nkeep = nrows / subsample;
      int16_t *dec = (int16_t *)malloc((size_t)(nkeep > 0 ? nkeep : 1) * (size_t)nheads * 2);
      for (int64_t k = 0; k < nkeep; ++k) memcpy(dec + k * nheads, rows + ((k + 1) * subsample - 1) * nheads, (size_t)nheads * 2);
      rows = dec; nrows = nkeep; }
   if (opt.mode == RT_WW) {                                  /* the roles of the heads: -order= or the TBINORD extension (src/readtape.c:883-902) */
      char hdr_order[21] = {0};
      if (flags & 2) memcpy(hdr_order, buf + 240 + 8, 20);
      const char *ord = orderarg ? orderarg : (hdr_order[0] ? hdr_order : "CMLcml");
      if ((int)strlen(ord) != nheads) { fprintf(stderr, "Whirlwind -order must name every head\n"); return 2; }
      char used_order[21] = {0};
      if (strchr(ord, 'x')) {                                /* heads that are not tracks (src/readtape.c:891 parks their samples in a spare track nobody reads): left out here */
         int used[RT_MAXTRKS], nu = 0;
         for (int h = 0; h < nheads; ++h) if (ord[h] != 'x') { used_order[nu] = ord[h]; used[nu++] = h; }
         int16_t *pk = (int16_t *)malloc((size_t)(nrows > 0 ? nrows : 1) * (size_t)(nu > 0 ? nu : 1) * 2);
         for (int64_t k = 0; k < nrows; ++k) for (int j = 0; j < nu; ++j) pk[k * nu + j] = rows[k * nheads + used[j]];
         rows = pk; nheads = nu; ord = used_order; }
      snprintf(opt.ww_order, sizeof opt.ww_order, "%s", ord);
      opt.ntrks = nheads;
      orderarg = NULL; flags &= ~3u;                         /* (no permutation below: head h is track h) */
      if (opt.bpi == 0) opt.bpi = 100; }
   float sample_deltat = (float)(int64_t)tdelta / 1e9f;      /* src/readtape.c:1345 */
   struct rt_dec *d = rt_dec_new(&opt, sample_deltat, (int64_t)tdelta);
   if (!d) { fprintf(stderr, "bad decoder options\n"); return 2; }
   if (parmfile) {
      size_t plen; unsigned char *ptxt = slurp(parmfile, &plen);
      if (!ptxt) { fprintf(stderr, "can't read %s\n", parmfile); return 2; }
      ptxt[plen] = 0;
      if (rt_parse_parms_text(opt.mode, (const char *)ptxt, d->parmsets) <= 0) { fprintf(stderr, "bad parms file\n"); return 2; }
      free(ptxt); }
   struct ofe *fe = ofe_new(d, rows, nrows, nheads, maxvolts, (int64_t)tstart);
   g_fe = fe;
   fe->invert = invert;
   {  /* head -> track permutation: "-order=" wins over the TBINORD header extension (src/readtape.c:877-915, 1346-1355).
         One character per head: a digit = that track (0 = msb), p/P = the parity track (last). */
      char hdr_order[21] = {0};
      if (flags & 2) memcpy(hdr_order, buf + 240 + 8, 20);
      const char *ord = orderarg ? orderarg : (hdr_order[0] ? hdr_order : NULL);
      /* a file without TBIN_NO_REORDER (0x01) "had a permutation applied to it": every track order is ignored (src/readtape.c:1646-1648) */
      if (ord && (flags & 1)) {
         const int n = (int)strlen(ord);
         unsigned seen = 0;
         if (n != nheads) { fprintf(stderr, "-order length doesn't match the number of heads\n"); return 99; }
         for (int i = 0; i < n; ++i) {
            int t = (ord[i] == 'p' || ord[i] == 'P') ? n - 1 : ((ord[i] >= '0' && ord[i] <= '9' && ord[i] - '0' <= n - 2) ? ord[i] - '0' : -1);
            if (t < 0) { fprintf(stderr, "bad -order string\n"); return 99; }
            fe->head_to_trk[i] = t; seen |= 1u << t; }
         if (seen + 1 != (1u << n)) { fprintf(stderr, "-order is not a permutation\n"); return 99; } } }
   if (skewarg) {
      int t = 0; const char *p = skewarg;
      while (*p && t < RT_MAXTRKS) { fe->skew_delaycnt[t++] = atoi(p); while (*p && *p != ',') ++p; if (*p == ',') ++p; } }
   char name[1024];
   if (outbase) {
      snprintf(name, sizeof name, "%s.tap", outbase); d->tapf = fopen(name, "wb");
      snprintf(name, sizeof name, "%s.log", outbase); d->logf = fopen(name, "w"); }
   if (evtname) { evtf = fopen(evtname, "wb"); d->on_transition = on_transition; d->on_attempt = on_attempt; fe->on_attempt_end = on_attempt_end; }

   struct rt_reader rd = { ofe_readblock, ofe_save_pos, ofe_restore_pos, fe };
   if (d->opt.bpi == 0) {                                    /* src/readtape.c:1656-1672: density from the first transitions, then rewind */
      float implied; int nb, hit_end;
      ofe_save_pos(fe);
      const float found = rt_density_prepass(d, &rd, &implied, &nb, &hit_end);
      if (found < 0) { fprintf(stderr, "density detection: a transition distance was not positive, or too many distinct ones\n"); return 99; }
      if (found == 0) {
         fprintf(stderr, "The detected density of %.0f is non-standard; please specify it.\n", implied); return 99; }
      ofe_restore_pos(fe); }
   if (deskew && opt.mode != RT_PE && !skewarg) {            /* src/readtape.c:1675-1717: calibrate on the first blocks, then rewind */
      int delays[RT_MAXTRKS] = {0}, hit_end = 0;
      ofe_save_pos(fe);
      if (rt_deskew_prepass(d, &rd, delays, &hit_end) < 0) { fprintf(stderr, "Some tracks have no transitions\n"); return 99; }
      ofe_restore_pos(fe);
      for (int t = 0; t < opt.ntrks; ++t) fe->skew_delaycnt[t] = delays[t];
      if (opt.mode == RT_WW) {                                /* init_trackpeak_state (src/readtape.c:1707, src/decoder.c:413-423): the delay lines and
                                                                 the windows' indices, extremes and countdown - not the rings, not the AGC */
         memset(fe->skew, 0, sizeof fe->skew);
         for (int t = 0; t < opt.ntrks; ++t) {
            fe->det[t].pkww_left = fe->det[t].pkww_right = fe->det[t].pkww_countdown = 0;
            fe->det[t].pkww_minv = fe->det[t].pkww_maxv = 0; } } }
   struct timespec t0, t1;
   clock_gettime(CLOCK_MONOTONIC, &t0);
   int ok = rt_process_blocks(d, &rd, blklimit);
   clock_gettime(CLOCK_MONOTONIC, &t1);
   double secs = (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
   if (d->logf) fprintf(d->logf, "decoded %d tape marks and %d blocks with %lld bytes; %d with errors, %d with warnings\n",
                        d->numtapemarks, d->numblks, d->numdatabytes, d->numblks_err, d->numblks_warn);
   if (timing) printf("{\"samples\": %lld, \"seconds\": %.6f, \"msamples_per_s\": %.4f, \"blocks\": %d, \"tapemarks\": %d, \"bytes\": %lld, \"ok\": %d}\n",
                      fe->lines_in, secs, fe->lines_in / secs / 1e6, d->numblks, d->numtapemarks, d->numdatabytes, ok);
   if (d->tapf) fclose(d->tapf);
   if (d->logf) fclose(d->logf);
   if (evtf) fclose(evtf);
   return (fe->fatal || d->fatal) ? 99 : 0; }
