// Bare MFMA issue-rate probe for gfx950: the local guide has no FP64 matrix peak, so the roofline
// denominator quoted in DESIGN.md / bench.py is re-measured here.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak
// Reports wall-clock TFLOP/s and, from s_memtime, shader cycles per MFMA per wave (clock-independent).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC> __global__ __launch_bounds__(256) void k64(double *out, long long *cyc, int iters, double a, double b) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double x = a + threadIdx.x * 1e-9, y = b;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
  }
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC> __global__ __launch_bounds__(256) void k32(float *out, long long *cyc, int iters, float a, float b) {
  f4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
  float x = a + threadIdx.x * 1e-6f, y = b;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
  }
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ __launch_bounds__(256) void kfma64(double *out, long long *cyc, int iters, double a, double b) {
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = i + threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <typename F> double timeit(F f, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("device %s CUs %d clock %d kHz\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate);
  double *o; hipMalloc(&o, 8 * 256 * 4096);
  long long *cyc; hipMalloc(&cyc, 8);
  long long hc;
  const int iters = 20000;
  auto report = [&](const char *name, int nacc, int wpc, double ms, double flops, double n_inst_per_wave) {
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    // s_memtime ticks at 100 MHz on gfx9; convert with the wall time of the same launch instead
    printf("%-22s acc %2d  %d wg/cu : %8.3f ms  %7.1f TFLOP/s   %.1f ns per inst per wave\n", name, nacc, wpc, ms, flops / ms * 1e-9,
           ms * 1e6 / n_inst_per_wave);
  };
  for (int wpc : {1, 2, 4}) {
    int grid = pr.multiProcessorCount * wpc;
    double ms;
    ms = timeit([&] { hipLaunchKernelGGL(k64<4>, dim3(grid), dim3(256), 0, 0, o, cyc, iters, 1.0, 0.5); }, 10);
    report("f64 mfma 16x16x4", 4, wpc, ms, (double)grid * 4 * iters * 4 * 2048.0, iters * 4.0);
    ms = timeit([&] { hipLaunchKernelGGL(k64<8>, dim3(grid), dim3(256), 0, 0, o, cyc, iters, 1.0, 0.5); }, 10);
    report("f64 mfma 16x16x4", 8, wpc, ms, (double)grid * 4 * iters * 8 * 2048.0, iters * 8.0);
    ms = timeit([&] { hipLaunchKernelGGL(k64<16>, dim3(grid), dim3(256), 0, 0, o, cyc, iters / 2, 1.0, 0.5); }, 10);
    report("f64 mfma 16x16x4", 16, wpc, ms, (double)grid * 4 * (iters / 2) * 16 * 2048.0, (iters / 2) * 16.0);
    ms = timeit([&] { hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, (float *)o, cyc, iters, 1.0f, 0.5f); }, 10);
    report("f32 mfma 16x16x4", 4, wpc, ms, (double)grid * 4 * iters * 4 * 2048.0, iters * 4.0);
    ms = timeit([&] { hipLaunchKernelGGL(k32<16>, dim3(grid), dim3(256), 0, 0, (float *)o, cyc, iters / 2, 1.0f, 0.5f); }, 10);
    report("f32 mfma 16x16x4", 16, wpc, ms, (double)grid * 4 * (iters / 2) * 16 * 2048.0, (iters / 2) * 16.0);
    ms = timeit([&] { hipLaunchKernelGGL(kfma64, dim3(grid), dim3(256), 0, 0, o, cyc, iters, 1.0000001, 1e-9); }, 10);
    report("f64 v_fma", 16, wpc, ms, (double)grid * 256 * iters * 16 * 2.0, iters * 16.0);
  }
  return 0;
}
