// TEST INFRASTRUCTURE — builds the kernel sources against tests/cpu_emul/hip/hip_runtime.h (see there).
#include <hip/hip_runtime.h>
#include <mutex>
dim3 threadIdx, blockIdx;
dim3 blockDim, gridDim;
unsigned char g_dyn_smem[160 * 1024] __attribute__((aligned(64)));
namespace hipemu {
Barrier g_block_barrier;
Barrier g_wave_barrier[16];
unsigned long long g_wave_scratch[16][64];

// The fibers of the block that is running: one stack each (kept across launches), run round robin by run_grid on the caller's
// thread.  A fiber leaves the processor only in fiber_yield() (a barrier it cannot pass yet) or by finishing the kernel.
constexpr size_t kStackBytes = 512 * 1024;
struct Fiber { ucontext_t ctx; char *stack = nullptr; bool done = false; };
static std::vector<Fiber> g_fibers;
static ucontext_t g_main;
static int g_cur = -1;
static const std::function<void()> *g_body = nullptr;
static unsigned g_block = 0;

void fiber_yield() { swapcontext(&g_fibers[g_cur].ctx, &g_main); }
static void fiber_entry() {
   (*g_body)();
   g_fibers[g_cur].done = true;
   swapcontext(&g_fibers[g_cur].ctx, &g_main); }

static std::mutex g_run_mutex;      // the fibers' state is process-global: one emulated launch at a time (ctypes releases the GIL - ADVICE r3)
void run_grid(dim3 grid, dim3 block, const std::function<void()> &body) {
   std::lock_guard<std::mutex> lock(g_run_mutex);
   gridDim = grid; blockDim = block;
   if (g_fibers.size() < block.x) g_fibers.resize(block.x);
   g_body = &body;
   for (unsigned b = 0; b < grid.x; ++b) {
      g_block = b;
      g_block_barrier.init(block.x);
      for (unsigned w = 0; w * 64 < block.x; ++w) g_wave_barrier[w].init(std::min(64u, block.x - w * 64));
      for (unsigned t = 0; t < block.x; ++t) {
         Fiber &f = g_fibers[t];
         if (!f.stack) f.stack = (char *)malloc(kStackBytes);
         getcontext(&f.ctx);
         f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = kStackBytes; f.ctx.uc_link = &g_main;
         makecontext(&f.ctx, fiber_entry, 0);
         f.done = false; }
      for (unsigned left = block.x; left;) {
         unsigned ran = 0;
         for (unsigned t = 0; t < block.x; ++t) {
            Fiber &f = g_fibers[t];
            if (f.done) continue;
            g_cur = (int)t; threadIdx = dim3(t); blockIdx = dim3(b);
            swapcontext(&g_main, &f.ctx);
            ++ran;
            if (f.done) --left; }
         if (!ran) break; } }
   g_cur = -1; g_body = nullptr; }
}
#include "rtfe_api.hip"
