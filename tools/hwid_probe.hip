// Where do the waves of a workgroup land?  Records HW_ID (SIMD, CU, SE) and XCC_ID of every wave of every
// workgroup of a launch shaped like k_window_ticks (256 threads, 2 workgroups per CU by LDS):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/hwid_probe tools/hwid_probe.hip && /tmp/hwid_probe [blocks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void probe(unsigned *out, int spin) {
  extern __shared__ char lds[];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
    out[(blockIdx.x * 4 + wave) * 2] = hw;
    out[(blockIdx.x * 4 + wave) * 2 + 1] = xcc;
  }
  // stay resident long enough for the whole grid's first round to be co-resident
  long long t0 = clock64();
  while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 1000) lds[0] = 1;
}

int main(int argc, char **argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 512;
  unsigned *d;
  hipMalloc(&d, blocks * 8 * sizeof(unsigned));
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024);
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 70 * 1024, 0, d, 2000000);
  std::vector<unsigned> h(blocks * 8);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> per_cu;   // (xcc, se, cu) -> blocks
  int same_simd_pairs = 0, pairs = 0;
  for (int b = 0; b < blocks; ++b) {
    const unsigned hw = h[(b * 4) * 2], xcc = h[(b * 4) * 2 + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
    if (b < 12) {
      printf("block %3d: xcc %u se %u sh %u cu %2u  simd of waves 0..3:", b, xcc, se, sh, cu);
      for (int w = 0; w < 4; ++w) printf(" %u", (h[(b * 4 + w) * 2] >> 4) & 3);
      printf("\n");
    }
  }
  for (auto &kv : per_cu) {
    const auto &v = kv.second;
    for (size_t i = 0; i + 1 < v.size(); i += 2) {
      ++pairs;
      const unsigned s0 = (h[(v[i] * 4) * 2] >> 4) & 3, s1 = (h[(v[i + 1] * 4) * 2] >> 4) & 3;
      same_simd_pairs += s0 == s1;
      if (pairs <= 6) printf("CU key %05x hosts blocks %d and %d (id distance %d): wave-0 SIMDs %u / %u\n", kv.first, v[i], v[i + 1], v[i + 1] - v[i], s0, s1);
    }
  }
  printf("distinct CUs used: %zu; co-resident pairs: %d, of which wave 0 on the same SIMD: %d\n", per_cu.size(), pairs, same_simd_pairs);
  return 0;
}
