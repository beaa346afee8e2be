// micro-benchmark: host memcpy rate INTO pinned memory by allocation flag, 1 and N threads (what bounds the arena gather)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double run(uint8_t* dst, const uint8_t* src, size_t n, int threads) {
  double best = 1e9;
  for (int rep = 0; rep < 5; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    size_t per = n / threads;
    for (int t = 0; t < threads; t++) th.emplace_back([=]() { memcpy(dst + t * per, src + t * per, per); });
    for (auto& x : th) x.join();
    best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  return best;
}
int main() {
  size_t n = 16 << 20;
  uint8_t* src = (uint8_t*)malloc(n);
  memset(src, 1, n);
  struct { const char* name; unsigned flags; } kinds[] = {{"default", hipHostMallocDefault}, {"noncoherent", hipHostMallocNonCoherent}, {"coherent", hipHostMallocCoherent},
                                                          {"numa-user", hipHostMallocNumaUser}, {"portable", hipHostMallocPortable}};
  for (auto& k : kinds) {
    void* p = nullptr;
    if (hipHostMalloc(&p, n, k.flags) != hipSuccess) { printf("%s: alloc failed\n", k.name); continue; }
    memset(p, 0, n);
    printf("%-12s 1 thread %.3f ms, 8 threads %.3f ms, 32 threads %.3f ms (16 MiB)\n", k.name, run((uint8_t*)p, src, n, 1), run((uint8_t*)p, src, n, 8), run((uint8_t*)p, src, n, 32));
    void* d = nullptr;
    hipMalloc(&d, n);
    hipStream_t st; hipStreamCreate(&st);
    double best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, st);
      hipStreamSynchronize(st);
      best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("%-12s H2D 16 MiB %.3f ms\n", k.name, best);
    hipFree(d); hipStreamDestroy(st); hipHostFree(p);
  }
  uint8_t* m = (uint8_t*)malloc(n); memset(m, 0, n);
  printf("%-12s 1 thread %.3f ms, 8 threads %.3f ms, 32 threads %.3f ms\n", "malloc", run(m, src, n, 1), run(m, src, n, 8), run(m, src, n, 32));
  return 0;
}
