// Bare MFMA issue-rate probe for gfx950: the local guide has no FP64 matrix peak, so the roofline
// denominators quoted in DESIGN.md / bench.py are re-measured here.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC> __global__ __launch_bounds__(256) void k64(double *out, int iters, double a, double b) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double x = a + threadIdx.x * 1e-9, y = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> __global__ __launch_bounds__(256) void k32(float *out, int iters, float a, float b) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
  float x = a + threadIdx.x * 1e-6f, y = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void kfma64(double *out, int iters, double a, double b) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> double timeit(F f) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("device %s CUs %d clock %d kHz\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate);
  double *o; hipMalloc(&o, 8 * 256 * 4096);
  const int iters = 20000;
  for (int wpc : {1, 2, 4}) {   // workgroups per CU (each 4 waves = one per SIMD)
    int grid = pr.multiProcessorCount * wpc;
    double ms = timeit([&] { hipLaunchKernelGGL(k64<4>, dim3(grid), dim3(256), 0, 0, o, iters, 1.0, 0.5); });
    double fl = (double)grid * 4 * iters * 4 * 2048.0;
    printf("f64 mfma 16x16x4  4 acc  %d wg/cu : %.3f ms  %.1f TFLOP/s  (%.1f cyc/mfma/SIMD @2.4GHz)\n", wpc, ms, fl / ms * 1e-9,
           ms * 1e-3 * 2.4e9 / (iters * 4.0 * wpc));
    ms = timeit([&] { hipLaunchKernelGGL(k64<16>, dim3(grid), dim3(256), 0, 0, o, iters / 4, 1.0, 0.5); });
    fl = (double)grid * 4 * (iters / 4) * 16 * 2048.0;
    printf("f64 mfma 16x16x4 16 acc  %d wg/cu : %.3f ms  %.1f TFLOP/s\n", wpc, ms, fl / ms * 1e-9);
    ms = timeit([&] { hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, (float *)o, iters, 1.0f, 0.5f); });
    fl = (double)grid * 4 * iters * 4 * 2048.0;
    printf("f32 mfma 16x16x4  4 acc  %d wg/cu : %.3f ms  %.1f TFLOP/s\n", wpc, ms, fl / ms * 1e-9);
    ms = timeit([&] { hipLaunchKernelGGL(kfma64, dim3(grid), dim3(256), 0, 0, o, iters, 1.0000001, 1e-9); });
    fl = (double)grid * 256 * iters * 16 * 2.0;
    printf("f64 v_fma        16 acc  %d wg/cu : %.3f ms  %.1f TFLOP/s\n", wpc, ms, fl / ms * 1e-9);
  }
  return 0;
}
