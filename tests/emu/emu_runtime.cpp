// TEST INFRASTRUCTURE ONLY -- see tests/emu/include/hip/hip_runtime.h.
#include <hip/hip_runtime.h>

const char* emu_current_kernel = nullptr;
thread_local emu_uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

namespace emu {

thread_local Wave* cur_wave = nullptr;
thread_local unsigned cur_lane = 0;
thread_local bool in_coop = false;

static std::mutex g_mu;
static std::condition_variable g_cv;
static unsigned long long g_launch_gen = 0;  // bumped once per cooperative launch

Runtime& rt() {
  static Runtime r;
  return r;
}

Runtime::Runtime() {
  for (unsigned t = 0; t < MAXT; t++) workers.emplace_back([this, t]() { worker(t); });
}

Runtime::~Runtime() {
  {
    std::lock_guard<std::mutex> lk(g_mu);
    stop = true;
    g_launch_gen++;
  }
  g_cv.notify_all();
  for (auto& w : workers) w.join();
}

void Runtime::worker(unsigned tid) {
  unsigned long long seen = 0;
  for (;;) {
    // launch parameters are read under the same lock that publishes them: a worker that wakes up late for a launch it
    // is not part of must never pair the old generation with the next launch's parameters
    dim3 g, blk;
    unsigned nt;
    {
      std::unique_lock<std::mutex> lk(g_mu);
      g_cv.wait(lk, [&]() { return g_launch_gen != seen; });
      seen = g_launch_gen;
      if (stop) return;
      g = grid;
      blk = block;
      nt = n_threads;
    }
    if (tid >= nt) continue;  // not part of this launch
    in_coop = true;
    unsigned total_blocks = g.x * g.y * g.z;
    for (unsigned b = 0; b < total_blocks; b++) {
      start.wait();
      gridDim = g;
      blockDim = blk;
      blockIdx = {b % g.x, (b / g.x) % g.y, b / (g.x * g.y)};
      threadIdx = {tid % blk.x, (tid / blk.x) % blk.y, tid / (blk.x * blk.y)};
      cur_wave = &waves[tid / 64];
      cur_lane = tid % 64;
      body();
      finish.wait();
    }
    in_coop = false;
  }
}

void Runtime::run(dim3 g, dim3 b, const std::function<void()>& fn) {
  unsigned nt = b.x * b.y * b.z;
  if (nt == 0 || nt > MAXT || nt % 64 != 0) { fprintf(stderr, "emu: block size %u unsupported (multiple of 64, <= %u)\n", nt, MAXT); abort(); }
  unsigned total_blocks = g.x * g.y * g.z;
  if (total_blocks == 0) return;
  start.reset(nt + 1);
  finish.reset(nt + 1);
  block_bar.reset(nt);
  for (unsigned w = 0; w < nt / 64; w++) { waves[w].bar.reset(64); waves[w].lanes = 64; }
  {
    std::lock_guard<std::mutex> lk(g_mu);
    body = fn;
    grid = g;
    block = b;
    n_threads = nt;
    g_launch_gen++;
  }
  g_cv.notify_all();
  for (unsigned blk = 0; blk < total_blocks; blk++) {
    start.wait();
    finish.wait();
  }
}

}  // namespace emu
