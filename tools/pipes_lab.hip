// How many kernels of DIFFERENT streams run at the same time?  Each stream gets a chain of kernels that do nothing but
// wait `us` microseconds in one workgroup; the achieved concurrency is (streams x launches x us) / wall time.
// usage: GPU_MAX_HW_QUEUES=24 ./pipes_lab      (prints one JSON line per configuration)
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__global__ void k_wait(unsigned long long ticks, unsigned long long *out) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = t0;
}

int main() {
  unsigned long long *d = nullptr;
  CK(hipMalloc(&d, 8));
  const int L = 200;
  for (int us : {20, 100}) {
    for (int blocks : {1, 256}) {
      for (int S : {1, 2, 3, 4, 5, 6, 8, 12, 16}) {
        std::vector<hipStream_t> st(S);
        for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (auto &s : st) hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, s, 100ull, d);
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < L; ++i)
          for (auto &s : st) hipLaunchKernelGGL(k_wait, dim3(blocks), dim3(64), 0, s, (unsigned long long)us * 100ull, d);
        CK(hipDeviceSynchronize());
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("{\"kernel_us\": %d, \"workgroups\": %d, \"streams\": %d, \"launches_per_stream\": %d, \"wall_ms\": %.2f, "
               "\"concurrency\": %.2f, \"launches_per_s\": %.0f}\n",
               us, blocks, S, L, wall * 1e3, S * L * us * 1e-6 / wall, S * L / wall);
        fflush(stdout);
        for (auto &s : st) CK(hipStreamDestroy(s));
      }
    }
  }
  // The same with one host thread PER stream (is the ceiling above the device's or the enqueueing thread's?): the
  // time the slowest thread needs to queue its launches is reported next to the wall time of the whole thing.
  for (int us : {5, 20}) {
    for (int S : {1, 4, 8, 12, 16}) {
      std::vector<hipStream_t> st(S);
      for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      for (auto &s : st) hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, s, 100ull, d);
      CK(hipDeviceSynchronize());
      std::vector<double> enq(S, 0.0);
      std::atomic<int> go{0};
      std::vector<std::thread> th;
      const auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < S; ++k)
        th.emplace_back([&, k] {
          go.fetch_add(1);
          while (go.load() < S) {
          }
          const auto a = std::chrono::steady_clock::now();
          for (int i = 0; i < L; ++i)
            hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, st[k], (unsigned long long)us * 100ull, d);
          enq[k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
        });
      for (auto &t : th) t.join();
      CK(hipDeviceSynchronize());
      const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      double worst = 0;
      for (double e : enq) worst = e > worst ? e : worst;
      printf("{\"host_threads\": %d, \"kernel_us\": %d, \"streams\": %d, \"launches_per_stream\": %d, \"wall_ms\": %.2f, "
             "\"slowest_enqueue_ms\": %.2f, \"launches_per_s\": %.0f, \"chain_us_per_launch\": %.1f}\n",
             S, us, S, L, wall * 1e3, worst * 1e3, S * L / wall, wall * 1e6 / L);
      fflush(stdout);
      for (auto &s : st) CK(hipStreamDestroy(s));
    }
  }
  return 0;
}
