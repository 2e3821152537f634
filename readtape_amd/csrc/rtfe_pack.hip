// rtfe_pack.hip — the event arena of a scan, packed on the device before it crosses PCIe (rtfe_pack_events, include/rt_frontend.h).
//
// rtfe_scan lays the arena out for the worst case: every (burst, parameter set, track) list has event_cap records of room, a fixed share of
// the burst's rows (C2: 38 MB of arena per 2^21 rows hold 6.7 MB of events).  The host replay reads lists, not the arena: what it needs is the
// same addressing rule - events[event_base + (p * ntrks + t) * event_cap + i] - over a smaller buffer.  Two kernels behind the scan, on its stream:
//   k_pack_plan   one workgroup: per burst the longest of its lists (cap2), the bursts' packed bases by a prefix sum, the total
//   k_pack_copy   a workgroup per list, 16 bytes a lane: the list's records to base + list * cap2
// The plan goes to the host with the burst table (16 bytes a burst); the device's burst table is left alone.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_frontend.h"

namespace rtfe {

__global__ void __launch_bounds__(1024) k_pack_plan(const rtfe_burst *__restrict__ bursts, const int32_t *__restrict__ nbursts, long long max_bursts,
                                                    const uint32_t *__restrict__ counts, int lists, rtfe_pack_entry *__restrict__ plan) {
   __shared__ unsigned long long s_v[1024];
   __shared__ unsigned long long s_carry;
   const int tid = threadIdx.x;
   long long nb = *nbursts;
   if (nb > max_bursts) nb = max_bursts;
   if (nb < 0) nb = 0;
   if (tid == 0) s_carry = 0;
   __syncthreads();
   for (long long b0 = 0; b0 < nb; b0 += 1024) {
      const long long b = b0 + tid;
      uint32_t cap2 = 0;
      if (b < nb) {
         const uint32_t *c = counts + (size_t)b * lists;
         for (int l = 0; l < lists; ++l) cap2 = c[l] > cap2 ? c[l] : cap2;
         if (cap2 < 1) cap2 = 1;
         if (cap2 > bursts[b].event_cap) cap2 = bursts[b].event_cap; }        // (a list that overflowed its region: RTFE_F_EVENT_OVERFLOW - the count says what there would have been)
      const unsigned long long n = (unsigned long long)cap2 * (unsigned long long)lists;
      s_v[tid] = n;                                                          // inclusive prefix sum over the workgroup (one workgroup, a few rounds per scan: nothing to optimise)
      __syncthreads();
      for (int s = 1; s < 1024; s <<= 1) {
         const unsigned long long y = tid >= s ? s_v[tid - s] : 0;
         __syncthreads();
         s_v[tid] += y;
         __syncthreads(); }
      const unsigned long long before = s_carry;
      if (b < nb) { rtfe_pack_entry e; e.event_base = before + s_v[tid] - n; e.event_cap = cap2; e.reserved = 0; plan[b] = e; }
      __syncthreads();
      if (tid == 1023) s_carry = before + s_v[1023];
      __syncthreads(); }
   if (tid == 0) { rtfe_pack_entry e; e.event_base = s_carry; e.event_cap = 0; e.reserved = (uint32_t)nb; plan[nb] = e; } }

__global__ void __launch_bounds__(256) k_pack_copy(const rtfe_burst *__restrict__ bursts, const int32_t *__restrict__ nbursts, long long max_bursts,
                                                   const uint32_t *__restrict__ counts, int lists, const rtfe_pack_entry *__restrict__ plan,
                                                   const uint4 *__restrict__ events, uint4 *__restrict__ out, unsigned long long out_cap) {
   long long nb = *nbursts;
   if (nb > max_bursts) nb = max_bursts;
   if (nb <= 0 || plan[nb].event_base > out_cap) return;                        // (does not fit: the host sees the total and fetches the arena as it is)
   const long long npairs = nb * lists;
   for (long long pr = blockIdx.x; pr < npairs; pr += gridDim.x) {
      const long long b = pr / lists;
      const int l = (int)(pr - b * lists);
      const rtfe_pack_entry e = plan[b];
      uint32_t n = counts[pr];
      if (n > e.event_cap) n = e.event_cap;
      const uint4 *src = events + bursts[b].event_base + (unsigned long long)l * bursts[b].event_cap;
      uint4 *dst = out + e.event_base + (unsigned long long)l * e.event_cap;
      for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i]; } }

// The end of the data (src/readtape.c:1410: the reader stops at the first row whose head-0 sample is 0x8000), looked for where the rows already
// are: a strided pass over a window on the host costs as much as reading it.  first[0] = the smallest such row, INT64_MAX if there is none.
__global__ void k_end_mark_init(long long *first) { *first = 0x7fffffffffffffffll; }
__global__ void __launch_bounds__(256) k_end_mark(const int16_t *__restrict__ rows, long long nrows, int ntrks, long long *__restrict__ first) {
   for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (long long)gridDim.x * blockDim.x)
      if (rows[r * ntrks] == (int16_t)0x8000) atomicMin(reinterpret_cast<unsigned long long *>(first), (unsigned long long)r); }

}  // namespace rtfe
