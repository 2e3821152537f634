/* rt_parmsets.c — the built-in parameter sets as plain data, and the .parms text grammar.
 * Values restate src/parmsets.c:77-118 (they are parsed there with sscanf("%f"), which yields the
 * same float as the literal with an f suffix). */
#include "rt_decode.h"

#include <ctype.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/*                    clk_window clk_alpha agc_window agc_alpha min_peak clk_factor pulse_adj bitfrac  rise   midbit  z1pt   z2pt */
static const struct rt_parms PE_SETS[] = {
   {1, 0, 0.2f, 5, 0.0f, 0.0f, 1.50f, 0.4f, 0.7f, 0.10f, 0, 0, 0, 0, 0},
   {1, 0, 0.2f, 5, 0.0f, 0.1f, 1.50f, 0.4f, 0.7f, 0.10f, 0, 0, 0, 0, 0},
   {1, 3, 0.0f, 5, 0.0f, 0.0f, 1.40f, 0.0f, 0.7f, 0.10f, 0, 0, 0, 0, 0},
   {1, 3, 0.0f, 5, 0.0f, 0.0f, 1.40f, 0.2f, 0.7f, 0.10f, 0, 0, 0, 0, 0},
   {1, 5, 0.0f, 5, 0.0f, 0.0f, 1.40f, 0.0f, 0.7f, 0.10f, 0, 0, 0, 0, 0},
   {1, 5, 0.0f, 5, 0.0f, 0.0f, 1.50f, 0.2f, 0.7f, 0.10f, 0, 0, 0, 0, 0},
   {1, 5, 0.0f, 5, 0.0f, 0.0f, 1.40f, 0.4f, 0.7f, 0.10f, 0, 0, 0, 0, 0},
   {1, 3, 0.0f, 5, 0.0f, 0.0f, 1.40f, 0.2f, 0.7f, 0.10f, 0, 0, 0, 0, 0} };
static const struct rt_parms NRZI_SETS[] = {
   {1, 0, 0.200f, 0, 0.300f, 1.000f, 0, 0.300f, 0.700f, 0.200f, 0.500f, 0, 0, 0, 0},
   {1, 0, 0.300f, 0, 0.300f, 1.000f, 0, 0.400f, 0.600f, 0.200f, 0.500f, 0, 0, 0, 0},
   {1, 2, 0.000f, 0, 0.300f, 1.000f, 0, 0.400f, 0.700f, 0.200f, 0.500f, 0, 0, 0, 0},
   {1, 0, 0.600f, 0, 0.300f, 1.000f, 0, 0.400f, 0.600f, 0.200f, 0.500f, 0, 0, 0, 0},
   {1, 2, 0.000f, 1, 0.000f, 0.500f, 0, 0.500f, 0.900f, 0.050f, 0.500f, 0, 0, 0, 0},
   {1, 0, 0.200f, 1, 0.000f, 1.000f, 0, 0.500f, 0.700f, 0.050f, 0.500f, 0, 0, 0, 0},
   {1, 2, 0.000f, 1, 0.000f, 0.500f, 0, 0.500f, 0.700f, 0.050f, 0.500f, 0, 0, 0, 0},
   {1, 0, 0.600f, 1, 0.000f, 0.500f, 0, 0.500f, 0.600f, 0.050f, 0.500f, 0, 0, 0, 0} };
static const struct rt_parms GCR_SETS[] = {
   {1,  0, 0.015f, 0, 0.500f, 0.200f, 0, 0.300f, 1.500f, 0.200f, 0, 1.450f, 2.350f, 0, 0},
   {1,  0, 0.020f, 0, 0.500f, 0.200f, 0, 0.300f, 1.500f, 0.200f, 0, 1.450f, 2.350f, 0, 0},
   {1,  0, 0.010f, 0, 0.500f, 0.200f, 0, 0.300f, 1.500f, 0.200f, 0, 1.450f, 2.350f, 0, 0},
   {1, 10, 0.000f, 0, 0.500f, 0.000f, 0, 0.600f, 1.500f, 0.140f, 0, 1.400f, 2.300f, 0, 0},
   {1,  0, 0.020f, 0, 0.500f, 0.200f, 0, 0.300f, 1.500f, 0.200f, 0, 1.480f, 2.350f, 0, 0} };
static const struct rt_parms WW_SETS[] = {
   {1, 0, 0.050f, 0, 0.500f, 1.000f, 0, 0, 0.400f, 0.200f, 0, 0, 0, 0, 0},
   {1, 0, 0.020f, 0, 0.500f, 0.050f, 0, 0, 0.200f, 0.200f, 0, 0, 0, 0, 0} };

void rt_default_parmsets(enum rt_mode mode, struct rt_parms out[RT_MAXPARMSETS]) {
   const struct rt_parms *src = NULL; size_t n = 0;
   memset(out, 0, sizeof(struct rt_parms) * RT_MAXPARMSETS);
   switch (mode) {
   case RT_PE:   src = PE_SETS;   n = sizeof PE_SETS / sizeof PE_SETS[0]; break;
   case RT_NRZI: src = NRZI_SETS; n = sizeof NRZI_SETS / sizeof NRZI_SETS[0]; break;
   case RT_GCR:  src = GCR_SETS;  n = sizeof GCR_SETS / sizeof GCR_SETS[0]; break;
   case RT_WW:   src = WW_SETS;   n = sizeof WW_SETS / sizeof WW_SETS[0]; break;
   default: return; }
   memcpy(out, src, n * sizeof(struct rt_parms)); }

/* ---- .parms text: "parms name, name, ..." then "{ v, v, ..., PRM }" lines; "//" comments.
 * Unknown (obsolete) names are skipped with their values; names missing from the file take the
 * value of the FIRST built-in set in every parsed set (src/parmsets.c:311-327). ---- */
struct pdesc { const char *name; int is_int; size_t off; float min, max; };      /* value ranges: src/parmsets.c:61-73 */
#define PD(n, i, lo, hi) {#n, i, offsetof(struct rt_parms, n), lo, hi}
static const struct pdesc PDESC[] = {
   PD(active, 1, 0, 1), PD(clk_window, 1, 0, 50), PD(clk_alpha, 0, 0, 1), PD(agc_window, 1, 0, 10), PD(agc_alpha, 0, 0, 1), PD(min_peak, 0, 0, 5),
   PD(clk_factor, 0, 0, 2), PD(pulse_adj, 0, 0, 1), PD(pkww_bitfrac, 0, 0, 2), PD(pkww_rise, 0, 0, 5), PD(midbit, 0, 0, 1), PD(z1pt, 0, 1, 2), PD(z2pt, 0, 2, 3) };
#define NPDESC ((int)(sizeof PDESC / sizeof PDESC[0]))

static void skipb(const char **p) { while (**p == ' ' || **p == '\t') ++*p; }

int rt_parse_parms_text(enum rt_mode mode, const char *text, struct rt_parms out[RT_MAXPARMSETS]) {
   struct rt_parms defaults[RT_MAXPARMSETS];
   int map[32], nfile = 0, given[32] = {0}, nsets = 0, got_names = 0;
   rt_default_parmsets(mode, defaults);
   memset(out, 0, sizeof(struct rt_parms) * RT_MAXPARMSETS);
   const char *p = text;
   while (*p) {
      const char *eol = strchr(p, '\n');
      size_t len = eol ? (size_t)(eol - p) : strlen(p);
      char line[512];
      if (len >= sizeof line) len = sizeof line - 1;
      memcpy(line, p, len); line[len] = 0;
      p += eol ? len + 1 : len;
      const char *q = line; skipb(&q);
      if (*q == 0 || *q == '\r' || (q[0] == '/' && q[1] == '/')) continue;
      if (strncasecmp(q, "readtape", 8) == 0) continue;      /* embedded CLI options: not handled here */
      if (strncasecmp(q, "parms", 5) == 0) {
         q += 5; skipb(&q); if (*q == ':') ++q;
         nfile = 0;
         for (;;) {
            char name[64]; int n = 0;
            skipb(&q);
            while ((isalnum((unsigned char)*q) || *q == '_') && n < 63) name[n++] = *q++;
            name[n] = 0;
            if (!n || nfile >= 31) break;
            map[nfile] = -1;
            for (int i = 0; i < NPDESC; ++i) if (strcmp(PDESC[i].name, name) == 0) { map[nfile] = i; given[i] = 1; }
            if (strcmp(name, "id") == 0) map[nfile] = -2;
            ++nfile;
            skipb(&q); if (*q == ',') ++q; else break; }
         got_names = 1;
         continue; }
      if (*q == '{') {
         if (!got_names || nsets >= RT_MAXPARMSETS - 1) return -1;
         ++q;
         struct rt_parms *s = &out[nsets];
         for (int f = 0; f < nfile; ++f) {
            skipb(&q);
            if (map[f] == -2) break;            /* the "PRM" id terminates the values */
            char *end; float v = strtof(q, &end);
            if (end == q) return -1;
            q = end;
            /* a value outside its range is fatal in the reference - also for a parameter the mode does not use, and 0..99 for
             * names it no longer knows (src/parmsets.c:286-297) */
            if (map[f] >= 0 ? (v < PDESC[map[f]].min || v > PDESC[map[f]].max) : (v < 0 || v > 99)) return -1;
            if (map[f] >= 0) {
               if (PDESC[map[f]].is_int) *(int *)((char *)s + PDESC[map[f]].off) = (int)v;
               else *(float *)((char *)s + PDESC[map[f]].off) = v; }
            skipb(&q); if (*q == ',') ++q; }
         ++nsets;
         continue; }
      return -1; }
   if (nsets == 0) return -1;
   for (int i = 0; i < NPDESC; ++i) if (!given[i])
      for (int sidx = 0; sidx < nsets; ++sidx) {
         if (PDESC[i].is_int) *(int *)((char *)&out[sidx] + PDESC[i].off) = *(const int *)((const char *)&defaults[0] + PDESC[i].off);
         else *(float *)((char *)&out[sidx] + PDESC[i].off) = *(const float *)((const char *)&defaults[0] + PDESC[i].off); }
   return nsets; }
