// fiber scheduler of the CPU debug emulator (see hip/hip_runtime.h): x86-64 user-space context switch
#include <hip/hip_runtime.h>
// void hipemu_switch(void **save_sp, void *load_sp): save callee-saved registers on the current stack,
// store the stack pointer, switch to load_sp and restore.
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipemu_switch,.-hipemu_switch
)");
namespace hipemu {
Block *g_block; Fiber *g_cur; void *g_sched_sp; dim3 g_blockIdx, g_blockDim, g_gridDim;
static std::function<void()> *g_body;
static void trampoline() { (*g_body)(); g_cur->done = true; for (;;) hipemu_switch(&g_cur->sp, g_sched_sp); }
void launch(std::function<void()> body, dim3 grid, dim3 block)
{
  const size_t STK = 256 * 1024;
  g_body = &body; g_blockDim = block; g_gridDim = grid;
  for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bx = 0; bx < grid.x; bx++) {
    Block b; b.n = block.x; b.arrived = 0; b.gen = 0;
    memset(b.warrived, 0, sizeof b.warrived); memset(b.wgen, 0, sizeof b.wgen);
    b.fibers.resize(b.n);
    g_block = &b; g_blockIdx = dim3(bx, by, bz);
    for (unsigned t = 0; t < b.n; t++) {
      Fiber &f = b.fibers[t]; f.stack = (char *) malloc(STK); f.tidx = dim3(t); f.done = false;
      // initial frame: six zeroed callee-saved registers, then the return address = trampoline; keep the
      // ABI alignment (rsp % 16 == 8 at function entry)
      uintptr_t top = ((uintptr_t) (f.stack + STK)) & ~(uintptr_t) 15;
      void **sp = (void **) top;
      *--sp = nullptr;                       // fake return address of trampoline (never used)
      *--sp = (void *) trampoline;
      for (int i = 0; i < 6; i++) *--sp = nullptr;
      f.sp = sp;
    }
    unsigned alive = b.n;
    while (alive) {
      alive = 0;
      for (unsigned t = 0; t < b.n; t++) {
        Fiber &f = b.fibers[t]; if (f.done) continue;
        g_cur = &f; hipemu_switch(&g_sched_sp, f.sp);
        if (!f.done) alive++;
      }
    }
    for (unsigned t = 0; t < b.n; t++) free(b.fibers[t].stack);
  }
}
}
