/* abi_client.c — TEST INFRASTRUCTURE: a plain C translation unit that includes include/rt_frontend.h and drives librtfe.so the way the
 * reference-side stub of INTEGRATION.md would (the seam of src/readtape.c:1396 readblock / src/decoder.c:817 process_sample): create a
 * front end from a configuration, hand it the TBIN payload in device memory, scan, and write what came back - burst table, counts,
 * events - to a file.  tests/test_abi.py compiles it with gcc (the header is C), runs it on the GPU box and compares the file with what the
 * Python binding (ctypes mirrors of the same structs) got for the same tape, byte for byte: the struct layouts of the header are the
 * contract, not the mirror.
 *
 *   abi_client <config.txt> <rows.bin> <out.bin>
 * config.txt: "key value" lines (mode ntrks maxvolts bpi ips tdelta_ns tstart_ns invert differentiate find_zeros nparmsets,
 *             parmset <bitfrac> <rise> <min_peak> <agc_alpha> <agc_window> <clk_factor>, head <i> <trk>, skew <i> <n>)
 * rows.bin:   int16 rows, ntrks per row
 * out.bin:    int32 nbursts | rtfe_burst[nbursts] | uint32 counts[nbursts][nparmsets][ntrks] | per burst, set, track: rtfe_event[count]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include <hip/hip_runtime_api.h>

#include "rt_frontend.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

int main(int argc, char **argv) {
   if (argc != 4) { fprintf(stderr, "usage: abi_client config.txt rows.bin out.bin\n"); return 2; }
   if (rtfe_abi_version() != RTFE_ABI_VERSION) { fprintf(stderr, "ABI version %d, header says %d\n", rtfe_abi_version(), RTFE_ABI_VERSION); return 2; }
   rtfe_config cfg;
   memset(&cfg, 0, sizeof cfg);
   FILE *f = fopen(argv[1], "r");
   if (!f) { perror(argv[1]); return 2; }
   char key[64];
   int np = 0;
   while (fscanf(f, "%63s", key) == 1) {
      if (!strcmp(key, "mode")) fscanf(f, "%d", &cfg.mode);
      else if (!strcmp(key, "ntrks")) { fscanf(f, "%d", &cfg.ntrks); for (int i = 0; i < cfg.ntrks; ++i) cfg.head_to_trk[i] = i; }
      else if (!strcmp(key, "maxvolts")) fscanf(f, "%f", &cfg.maxvolts);
      else if (!strcmp(key, "bpi")) fscanf(f, "%f", &cfg.bpi);
      else if (!strcmp(key, "ips")) fscanf(f, "%f", &cfg.ips);
      else if (!strcmp(key, "tdelta_ns")) { long long v; fscanf(f, "%lld", &v); cfg.tdelta_ns = v; }
      else if (!strcmp(key, "tstart_ns")) { long long v; fscanf(f, "%lld", &v); cfg.tstart_ns = v; }
      else if (!strcmp(key, "invert")) fscanf(f, "%d", &cfg.invert);
      else if (!strcmp(key, "differentiate")) fscanf(f, "%d", &cfg.differentiate);
      else if (!strcmp(key, "find_zeros")) fscanf(f, "%d", &cfg.find_zeros);
      else if (!strcmp(key, "head")) { int i, t; fscanf(f, "%d %d", &i, &t); if (i >= 0 && i < RTFE_MAXTRKS) cfg.head_to_trk[i] = t; }
      else if (!strcmp(key, "skew")) { int i, n; fscanf(f, "%d %d", &i, &n); if (i >= 0 && i < RTFE_MAXTRKS) cfg.skew_delaycnt[i] = n; }
      else if (!strcmp(key, "parmset")) {
         rtfe_parmset *p = &cfg.parmset[np < RTFE_MAXPARMSETS ? np : RTFE_MAXPARMSETS - 1];
         fscanf(f, "%f %f %f %f %d %f", &p->pkww_bitfrac, &p->pkww_rise, &p->min_peak, &p->agc_alpha, &p->agc_window, &p->clk_factor);
         ++np; }
      else { fprintf(stderr, "unknown key %s\n", key); return 2; } }
   fclose(f);
   cfg.nparmsets = np;
   /* the rows */
   f = fopen(argv[2], "rb");
   if (!f) { perror(argv[2]); return 2; }
   fseek(f, 0, SEEK_END);
   const long bytes = ftell(f);
   fseek(f, 0, SEEK_SET);
   const int64_t nrows = bytes / (2 * cfg.ntrks);
   int16_t *rows = (int16_t *)malloc((size_t)bytes);
   if (fread(rows, 1, (size_t)bytes, f) != (size_t)bytes) { fprintf(stderr, "short read\n"); return 2; }
   fclose(f);

   rtfe_handle *h = NULL;
   if (rtfe_create(&cfg, &h) != 0) { fprintf(stderr, "rtfe_create: %s\n", rtfe_last_error()); return 4; }
   const size_t ws_bytes = rtfe_workspace_bytes(h, nrows);
   const int64_t max_bursts = rtfe_max_bursts(h, nrows), cap = rtfe_event_capacity(h, nrows);
   int16_t *d_rows; void *d_ws; rtfe_burst *d_bursts; int32_t *d_nb; uint32_t *d_counts; rtfe_event *d_events;
   CHECK(hipMalloc((void **)&d_rows, (size_t)bytes + 64));
   CHECK(hipMalloc(&d_ws, ws_bytes));
   CHECK(hipMalloc((void **)&d_bursts, (size_t)max_bursts * sizeof(rtfe_burst)));
   CHECK(hipMalloc((void **)&d_nb, 16));
   CHECK(hipMalloc((void **)&d_counts, (size_t)max_bursts * np * cfg.ntrks * 4));
   CHECK(hipMalloc((void **)&d_events, (size_t)cap * sizeof(rtfe_event)));
   CHECK(hipMemcpy(d_rows, rows, (size_t)bytes, hipMemcpyHostToDevice));
   if (rtfe_scan(h, d_rows, nrows, nrows, 0, 1, d_ws, ws_bytes, d_bursts, max_bursts, d_nb, d_counts, d_events, cap, NULL) != 0) {
      fprintf(stderr, "rtfe_scan: %s\n", rtfe_last_error()); return 5; }
   CHECK(hipDeviceSynchronize());
   int32_t nb = 0;
   CHECK(hipMemcpy(&nb, d_nb, 4, hipMemcpyDeviceToHost));
   rtfe_burst *bursts = (rtfe_burst *)malloc((size_t)(nb > 0 ? nb : 1) * sizeof(rtfe_burst));
   uint32_t *counts = (uint32_t *)malloc((size_t)(nb > 0 ? nb : 1) * np * cfg.ntrks * 4);
   CHECK(hipMemcpy(bursts, d_bursts, (size_t)nb * sizeof(rtfe_burst), hipMemcpyDeviceToHost));
   CHECK(hipMemcpy(counts, d_counts, (size_t)nb * np * cfg.ntrks * 4, hipMemcpyDeviceToHost));
   f = fopen(argv[3], "wb");
   if (!f) { perror(argv[3]); return 2; }
   fwrite(&nb, 4, 1, f);
   fwrite(bursts, sizeof(rtfe_burst), (size_t)nb, f);
   fwrite(counts, 4, (size_t)nb * np * cfg.ntrks, f);
   long nev = 0;
   for (int b = 0; b < nb; ++b)
      for (int p = 0; p < np; ++p)
         for (int t = 0; t < cfg.ntrks; ++t) {
            /* the region of (burst b, parmset p, track t), include/rt_frontend.h */
            const uint32_t n = counts[((size_t)b * np + p) * cfg.ntrks + t];
            if (!n) continue;
            rtfe_event *ev = (rtfe_event *)malloc((size_t)n * sizeof(rtfe_event));
            CHECK(hipMemcpy(ev, d_events + bursts[b].event_base + (uint64_t)(p * cfg.ntrks + t) * bursts[b].event_cap, (size_t)n * sizeof(rtfe_event), hipMemcpyDeviceToHost));
            fwrite(ev, sizeof(rtfe_event), n, f);
            nev += n;
            free(ev); }
   fclose(f);
   printf("abi_client: %d bursts, %ld events, W(set 0) = %d\n", nb, nev, rtfe_pkww_width(h, 0));
   rtfe_destroy(h);
   return 0; }
