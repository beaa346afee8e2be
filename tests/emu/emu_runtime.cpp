// TEST INFRASTRUCTURE ONLY -- see tests/emu/include/hip/hip_runtime.h.
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#include <ucontext.h>

const char* emu_current_kernel = nullptr;
thread_local emu_uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

namespace emu {

thread_local Wave* cur_wave = nullptr;
thread_local unsigned cur_lane = 0;
thread_local bool in_coop = false;

namespace {

constexpr size_t STACK_BYTES = 256 * 1024;

// Context switch between fibers. x86-64: the callee-saved registers and the stack pointer, in user space (glibc's swapcontext also
// saves the signal mask: two system calls per switch, which is most of the time of a launch of thousands of one-wave blocks);
// elsewhere: ucontext.
#if defined(__x86_64__)
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.hidden emu_switch
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch,.-emu_switch
)");
struct Context {
  void* sp = nullptr;
};
void fiber_main();
inline void ctx_init(Context& c, void* stack, size_t bytes) {
  uintptr_t top = ((uintptr_t)stack + bytes) & ~(uintptr_t)15;
  void** a = (void**)(top - 16);  // the `ret` of emu_switch pops this: fiber_main starts with the stack pointer 8 off a 16-byte boundary, as after a call
  a[0] = (void*)&fiber_main;
  a[1] = nullptr;
  for (int k = 1; k <= 6; k++) a[-k] = nullptr;
  c.sp = (void*)(a - 6);
}
inline void ctx_switch(Context& from, Context& to) { emu_switch(&from.sp, to.sp); }
#else
struct Context {
  ucontext_t uc;
};
void fiber_main();
inline void ctx_init(Context& c, void* stack, size_t bytes) {
  getcontext(&c.uc);
  c.uc.uc_stack.ss_sp = stack;
  c.uc.uc_stack.ss_size = bytes;
  c.uc.uc_link = nullptr;
  makecontext(&c.uc, fiber_main, 0);
}
inline void ctx_switch(Context& from, Context& to) { swapcontext(&from.uc, &to.uc); }
#endif

struct Fiber {
  Context ctx;
  void* stack = nullptr;
  bool done = true;
  emu_uint3 tidx;
};

// one cooperative launch at a time (the kernels' __shared__ arrays are statics anyway)
std::mutex g_launch;
Fiber g_fibers[Runtime::MAXT];
Context g_main;
unsigned g_nt = 0, g_cur = 0, g_live = 0;
const std::function<void()>* g_body = nullptr;

void enter(unsigned tid) {  // (a fiber starts or resumes: the per-thread variables of the kernel language are its own again)
  g_cur = tid;
  threadIdx = g_fibers[tid].tidx;
  cur_wave = &rt().waves[tid / 64];
  cur_lane = tid % 64;
}

void fiber_main() {
  unsigned tid = g_cur;
  enter(tid);
  (*g_body)();
  g_fibers[tid].done = true;
  g_live--;
  if (g_live == 0) { Context dead; ctx_switch(dead, g_main); abort(); }
  yield();  // never comes back: finished fibers are skipped
  abort();
}

}  // namespace

void stuck(const char* what, unsigned arrived, unsigned n) {
  fprintf(stderr, "emu: %s stuck in kernel %s (arrived %u of %u)\n", what, emu_current_kernel ? emu_current_kernel : "?", arrived, n);
  abort();
}

void yield() {
  unsigned me = g_cur, nx = me;
  do nx = nx + 1 == g_nt ? 0 : nx + 1; while (g_fibers[nx].done && nx != me);
  if (nx == me) {
    if (g_fibers[me].done) stuck("scheduler (no live fiber)", 0, g_nt);
    return;  // the only live fiber: whoever it waits for will never arrive; the caller's watchdog reports it
  }
  g_cur = nx;
  ctx_switch(g_fibers[me].ctx, g_fibers[nx].ctx);
  enter(me);
}

Runtime& rt() {
  static Runtime r;
  return r;
}

void Runtime::run(dim3 g, dim3 b, const std::function<void()>& fn) {
  unsigned nt = b.x * b.y * b.z;
  if (nt == 0 || nt > MAXT || nt % 64 != 0) { fprintf(stderr, "emu: block size %u unsupported (multiple of 64, <= %u)\n", nt, MAXT); abort(); }
  unsigned total_blocks = g.x * g.y * g.z;
  if (total_blocks == 0) return;
  std::lock_guard<std::mutex> lk(g_launch);
  for (unsigned t = 0; t < nt; t++)
    if (!g_fibers[t].stack) {
      void* p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
      if (p == MAP_FAILED) { perror("emu: mmap"); abort(); }
      g_fibers[t].stack = p;
    }
  const dim3 save_grid = gridDim, save_block = blockDim;
  const emu_uint3 save_tidx = threadIdx, save_bidx = blockIdx;
  g_nt = nt;
  g_body = &fn;
  gridDim = g;
  blockDim = b;
  in_coop = true;
  for (unsigned blk = 0; blk < total_blocks; blk++) {
    blockIdx = {blk % g.x, (blk / g.x) % g.y, blk / (g.x * g.y)};
    block_bar.reset(nt);
    for (unsigned w = 0; w < nt / 64; w++) { waves[w].bar.reset(64); waves[w].lanes = 64; }
    for (unsigned t = 0; t < nt; t++) {
      Fiber& f = g_fibers[t];
      ctx_init(f.ctx, f.stack, STACK_BYTES);
      f.done = false;
      f.tidx = {t % b.x, (t / b.x) % b.y, t / (b.x * b.y)};
    }
    g_live = nt;
    g_cur = 0;
    ctx_switch(g_main, g_fibers[0].ctx);  // back here when the last fiber of the block has finished
  }
  in_coop = false;
  gridDim = save_grid;
  blockDim = save_block;
  threadIdx = save_tidx;
  blockIdx = save_bidx;
}

}  // namespace emu

// ---- test hook (tests/test_pinflate.py): the chunked DEFLATE decoder of am355_pinflate.cpp by itself ----
#include "am355_pinflate.h"
extern "C" long am355_emu_pinflate(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap, size_t chunk_bytes, unsigned n_threads) {
  std::vector<uint8_t> v;
  if (am355::inflate_raw_parallel(in, in_len, v, out_cap, chunk_bytes, n_threads) != 0) return -1;
  if (v.size() > out_cap) return -1;
  if (!v.empty()) memcpy(out, v.data(), v.size());
  return (long)v.size();
}
