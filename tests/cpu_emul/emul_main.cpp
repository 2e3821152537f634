// TEST INFRASTRUCTURE — builds the kernel sources against tests/cpu_emul/hip/hip_runtime.h (see there).
#include <hip/hip_runtime.h>
thread_local dim3 threadIdx, blockIdx;
dim3 blockDim, gridDim;
unsigned char g_dyn_smem[160 * 1024] __attribute__((aligned(64)));
namespace hipemu {
Barrier g_block_barrier;
Barrier g_wave_barrier[16];
unsigned long long g_wave_scratch[16][64];
}
#include "rtfe_api.hip"
