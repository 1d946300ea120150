// Shader clock a LONE workgroup runs at (the sliding window's one-window host tick, the node callback): a dependent fp64 FMA chain in one
// wave, s_memtime (shader cycles) against s_memrealtime (100 MHz), launch after launch with a host synchronisation in between -- the way
// a per-tick caller drives the GPU -- and the same chain with the other CUs kept busy by a second kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/lone_wg_clock.hip -o /tmp/lone_wg_clock && /tmp/lone_wg_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void chain(double *out, unsigned long long *clk, int iters) {
  double x = 1.0 + threadIdx.x * 1e-9, a = 1.0000001, b = 1e-9;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) x = __builtin_fma(x, a, b);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
__global__ void busy(double *out, int iters) {
  double x = 1.0 + threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
  if (x == 123.0) out[0] = x;
}
int main() {
  double *out; unsigned long long *clk, h[2];
  hipMalloc(&out, 4096); hipMalloc(&clk, 16);
  hipStream_t s, s2; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  const int iters = 1500;   // 24 000 dependent FMAs: ~100 us
  auto series = [&](const char *name, int n, int gap_us) {
    std::vector<double> ghz, us, fmacyc;
    for (int i = 0; i < n; ++i) {
      hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s, out, clk, iters);
      hipStreamSynchronize(s);
      hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      ghz.push_back((double)h[0] / ((double)h[1] * 10.0));   // cycles per ns
      us.push_back((double)h[1] / 100.0);
      fmacyc.push_back((double)h[0] / (iters * 16.0));
      if (gap_us) { timespec ts{0, gap_us * 1000}; nanosleep(&ts, nullptr); }
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("%-58s first launch %.2f GHz (%.0f us), median %.2f GHz (%.0f us), min %.2f max %.2f; %.2f cycles per dependent fp64 FMA\n", name, ghz[0], us[0], med(ghz), med(us),
           *std::min_element(ghz.begin(), ghz.end()), *std::max_element(ghz.begin(), ghz.end()), med(fmacyc));
  };
  series("lone wave, launches back to back with a host sync", 300, 0);
  series("lone wave, one launch per 20 ms (a 50 Hz caller)", 40, 20000);
  series("lone wave, one launch per 1 ms", 200, 1000);
  hipLaunchKernelGGL(busy, dim3(2048), dim3(256), 0, s2, out, 40000000);   // the rest of the chip busy meanwhile (~ a second)
  series("lone wave beside 2048 busy workgroups", 300, 0);
  hipDeviceSynchronize();
  return 0;
}
