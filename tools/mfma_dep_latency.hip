// Issue interval of v_mfma_f64_16x16x4_f64 seen by ONE wave: chains of dependent MFMAs (same accumulator) against 2 / 4 independent
// accumulators, with 1, 4 (one per SIMD), 8 (two per SIMD) or 16 (four per SIMD) waves of a workgroup issuing at the same time.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_dep_latency.hip -o /tmp/mfma_dep && /tmp/mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC> __global__ __launch_bounds__(1024) void k(double *out, long long *cyc, int iters, int active) {
  const int wave = threadIdx.x >> 6;
  if (wave >= active) return;
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j % NACC], 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
int main() {
  double *out;
  long long *cyc, h[16];
  hipMalloc(&out, 1024 * 8);
  hipMalloc(&cyc, 128);
  const int iters = 2000;
  for (int nacc : {1, 2, 4})
    for (int active : {1, 4, 8, 16}) {
      for (int rep = 0; rep < 2; ++rep) {
        if (nacc == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(1024), 0, 0, out, cyc, iters, active);
        if (nacc == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(1024), 0, 0, out, cyc, iters, active);
        if (nacc == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(1024), 0, 0, out, cyc, iters, active);
        hipDeviceSynchronize();
      }
      hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
      long long mx = 0;
      for (int w = 0; w < active; ++w) mx = h[w] > mx ? h[w] : mx;
      printf("accumulators %d, waves %2d: wave 0 %.1f ticks per MFMA, slowest wave %.1f (the oldest wave wins the issue arbitration)\n", nacc, active,
             (double)h[0] / (iters * 8.0), (double)mx / (iters * 8.0));
    }
  return 0;
}
