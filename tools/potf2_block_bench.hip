// Microbenchmark of the 16x16 diagonal-block factor + inverse that sits on the serial critical path
// of potf2 (csrc/cgp_kernels.hpp, factor_block16): one wavefront, block in LDS, s_memtime per call.
//   hipcc --offload-arch=gfx950 -O3 -I corenav_gp_amd/csrc tools/potf2_block_bench.hip -o tools/potf2_block_bench
#include "cgp_kernels.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace cgp;

// readlane of a double (lane index wave-uniform): what the DPP form replaced (variants 1, 2, 5 keep it for the record)
__device__ __forceinline__ double rdlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

template <int V> __device__ __forceinline__ void variant(double (&a)[DB], double (&w)[DB], int &bad, int l15);

template <> __device__ __forceinline__ void variant<0>(double (&a)[DB], double (&w)[DB], int &bad, int l15) {
  factor_block16<double>(a, w, bad, 0, l15);   // the shipped code (DPP form)
}
// V5: the v_readlane form the DPP one replaced
template <> __device__ __forceinline__ void variant<5>(double (&a)[DB], double (&w)[DB], int &bad, int l15) {
  using P = Prec<double>;
  double rinv[DB];
#pragma unroll
  for (int j = 0; j < DB; ++j) {
    double dj = rdlane(a[j], j);
    if (!(dj > 0.0)) { if (bad == 0) bad = j + 1; dj = 1.0; }
    const double rs = P::rsqrt_(dj);
    rinv[j] = rs;
    const double l = (l15 == j) ? dj * rs : a[j] * rs;
    a[j] = l;
#pragma unroll
    for (int c = j + 1; c < DB; ++c) a[c] -= l * rdlane(l, c);
  }
#pragma unroll
  for (int i = 0; i < DB; ++i) {
    double s2 = 0;
#pragma unroll
    for (int q = 0; q < DB; ++q)
      if (q < i) s2 += rdlane(a[q], i) * w[q];
    w[i] = (i < l15) ? 0.0 : ((i == l15) ? rinv[i] : -s2 * rinv[i]);
  }
}
// V1: inverse by right-looking substitution (independent FMAs per column step instead of a dot-product chain)
template <> __device__ __forceinline__ void variant<1>(double (&a)[DB], double (&w)[DB], int &bad, int l15) {
  using P = Prec<double>;
  double rinv[DB];
#pragma unroll
  for (int j = 0; j < DB; ++j) {
    double dj = rdlane(a[j], j);
    if (!(dj > 0.0)) { if (bad == 0) bad = j + 1; dj = 1.0; }
    const double rs = P::rsqrt_(dj);
    rinv[j] = rs;
    const double l = (l15 == j) ? dj * rs : a[j] * rs;
    a[j] = l;
#pragma unroll
    for (int c = j + 1; c < DB; ++c) a[c] -= l * rdlane(l, c);
  }
  double t[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i) t[i] = (i == l15) ? 1.0 : 0.0;
#pragma unroll
  for (int q = 0; q < DB; ++q) {
    w[q] = t[q] * rinv[q];
#pragma unroll
    for (int i = q + 1; i < DB; ++i) t[i] -= rdlane(a[q], i) * w[q];
  }
#pragma unroll
  for (int i = 0; i < DB; ++i) w[i] = (i < l15) ? 0.0 : w[i];
}
// V2: V1 + the column is scaled off the critical path: the rank-1 update uses the unnormalised column
// u and 1/d = rs^2, so the broadcasts of u do not wait for the rsqrt chain
template <> __device__ __forceinline__ void variant<2>(double (&a)[DB], double (&w)[DB], int &bad, int l15) {
  using P = Prec<double>;
  double rinv[DB];
#pragma unroll
  for (int j = 0; j < DB; ++j) {
    double dj = rdlane(a[j], j);
    if (!(dj > 0.0)) { if (bad == 0) bad = j + 1; dj = 1.0; }
    const double u = a[j];
    double uc[DB];
#pragma unroll
    for (int c = j + 1; c < DB; ++c) uc[c] = rdlane(u, c);
    const double rs = P::rsqrt_(dj);
    rinv[j] = rs;
    const double v = u * (rs * rs);
    a[j] = (l15 == j) ? dj * rs : u * rs;
#pragma unroll
    for (int c = j + 1; c < DB; ++c) a[c] -= v * uc[c];
  }
  double t[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i) t[i] = (i == l15) ? 1.0 : 0.0;
#pragma unroll
  for (int q = 0; q < DB; ++q) {
    w[q] = t[q] * rinv[q];
#pragma unroll
    for (int i = q + 1; i < DB; ++i) t[i] -= rdlane(a[q], i) * w[q];
  }
#pragma unroll
  for (int i = 0; i < DB; ++i) w[i] = (i < l15) ? 0.0 : w[i];
}

// V3: 64-bit DPP row_newbcast (gfx90a+): "multiply by lane c's value" is ONE v_fmac_f64_dpp instead of
// two v_readlane_b32 and an fma -- the four 16-lane rows hold identical copies, so a row-local
// broadcast is the right one.  A DPP read of a VGPR needs 2 wait states after the VALU write.
template <> __device__ __forceinline__ void variant<3>(double (&a)[DB], double (&w)[DB], int &bad, int l15) {
  using P = Prec<double>;
  double rinv[DB];
  static_for<0, DB>([&](auto jc) {
    constexpr int J = decltype(jc)::value;
    double dj = mov_bcast<J>(a[J]);
    if (!(dj > 0.0)) { if (bad == 0) bad = J + 1; dj = 1.0; }
    const double rs = P::rsqrt_(dj);
    rinv[J] = rs;
    const double l = (l15 == J) ? dj * rs : a[J] * rs;
    a[J] = l;
    const double nl = -l;
    static_for<J + 1, DB>([&](auto cc) {
      constexpr int C = decltype(cc)::value;
      fmac_bcast<C, C == J + 1>(a[C], l, nl);
    });
  });
  double t[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i) t[i] = (i == l15) ? 1.0 : 0.0;
  static_for<0, DB>([&](auto qc) {
    constexpr int Q = decltype(qc)::value;
    w[Q] = t[Q] * rinv[Q];
    const double nw = -w[Q];
    static_for<Q + 1, DB>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      fmac_bcast<I, false>(t[I], a[Q], nw);
    });
  });
#pragma unroll
  for (int i = 0; i < DB; ++i) w[i] = (i < l15) ? 0.0 : w[i];
}

// V4: V3 + no select on the diagonal lane (a[J] there IS the pivot) + one third-order rsqrt step
template <> __device__ __forceinline__ void variant<4>(double (&a)[DB], double (&w)[DB], int &bad, int l15) {
  double rinv[DB];
  static_for<0, DB>([&](auto jc) {
    constexpr int J = decltype(jc)::value;
    double dj = mov_bcast<J>(a[J]);
    const bool ok = dj > 0.0;
    if (!ok && bad == 0) bad = J + 1;
    dj = ok ? dj : 1.0;
    const double rs = rsqrt3(dj);
    rinv[J] = rs;
    const double l = (ok ? a[J] : ((l15 == J) ? 1.0 : a[J])) * rs;
    a[J] = l;
    const double nl = -l;
    static_for<J + 1, DB>([&](auto cc) {
      constexpr int C = decltype(cc)::value;
      fmac_bcast<C, C == J + 1>(a[C], l, nl);
    });
  });
  double t[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i) t[i] = (i == l15) ? 1.0 : 0.0;
  static_for<0, DB>([&](auto qc) {
    constexpr int Q = decltype(qc)::value;
    w[Q] = t[Q] * rinv[Q];
    const double nw = -w[Q];
    static_for<Q + 1, DB>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      fmac_bcast<I, false>(t[I], a[Q], nw);
    });
  });
#pragma unroll
  for (int i = 0; i < DB; ++i) w[i] = (i < l15) ? 0.0 : w[i];
}


// V6: V4 rescheduled, same arithmetic in the same order per register (bitwise V4's results):
//  * the trailing updates of step J-1 to the columns >= J+1 are issued AFTER the broadcast of pivot J, i.e. under the
//    rsqrt chain of step J instead of in front of it (only column J had to be current for that pivot);
//  * the inverse's forward-substitution step J (independent of the factor chain once column J is final) is issued inside
//    the factor loop too, so its 120 fmacs fill the chain's issue bubbles instead of forming a second serial phase.
template <> __device__ __forceinline__ void variant<6>(double (&a)[DB], double (&w)[DB], int &bad, int l15) {
  double t[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i) t[i] = (i == l15) ? 1.0 : 0.0;
  double lp = 0.0, nlp = 0.0;
  static_for<0, DB>([&](auto jc) {
    constexpr int J = decltype(jc)::value;
    double dj = mov_bcast<J>(a[J]);
    if constexpr (J > 0) {
      static_for<J + 1, DB>([&](auto cc) {
        constexpr int C = decltype(cc)::value;
        fmac_bcast<C, false>(a[C], lp, nlp);
      });
    }
    const bool ok = dj > 0.0;
    if (!ok && bad == 0) bad = J + 1;
    dj = ok ? dj : 1.0;
    const double rs = rsqrt3(dj);
    const double l = (ok ? a[J] : ((l15 == J) ? 1.0 : a[J])) * rs;
    a[J] = l;
    const double nl = -l;
    if constexpr (J + 1 < DB) fmac_bcast<J + 1, true>(a[J + 1], l, nl);
    w[J] = t[J] * rs;
    const double nw = -w[J];
    static_for<J + 1, DB>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      fmac_bcast<I, false>(t[I], a[J], nw);
    });
    lp = l;
    nlp = nl;
  });
#pragma unroll
  for (int i = 0; i < DB; ++i) w[i] = (i < l15) ? 0.0 : w[i];
}

template <int V> __global__ __launch_bounds__(64) void kbench(const double *A, double *Lout, double *Wout, long long *cyc, int iters) {
  __shared__ double blk[DB * DB];
  const int lane = threadIdx.x, l15 = lane & 15;
  for (int i = lane; i < DB * DB; i += 64) blk[i] = A[i];
  __syncthreads();
  double a[DB], w[DB];
  int bad = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < DB; ++c) a[c] = blk[c * DB + l15];
    variant<V>(a, w, bad, l15);
    if (it + 1 < iters) asm volatile("" ::: "memory");
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (lane < DB && blockIdx.x == 0) {
    for (int c = 0; c < DB; ++c) {
      Lout[c * DB + l15] = (l15 >= c) ? a[c] : 0.0;
      Wout[l15 * DB + c] = w[c];  // W[c][l15]
    }
  }
  if (lane == 0 && blockIdx.x == 0) cyc[0] = t1 - t0 + bad;
}

template <int V> void run(const double *dA, double *dL, double *dW, long long *dc, const std::vector<double> &A) {
  const int iters = 20000;
  kbench<V><<<2048, 64>>>(dA, dL, dW, dc, iters);
  hipDeviceSynchronize();
  hipEvent_t ev0, ev1;
  hipEventCreate(&ev0); hipEventCreate(&ev1);
  hipEventRecord(ev0);
  kbench<V><<<2048, 64>>>(dA, dL, dW, dc, iters);
  hipEventRecord(ev1);
  hipEventSynchronize(ev1);
  float ms = 0; hipEventElapsedTime(&ms, ev0, ev1);
  printf("variant %d: kernel %.3f ms for %d iterations of 2 waves/SIMD -> %.1f ns per block-iteration wall\n", V, ms, iters, ms * 1e6 / iters);
  long long c; std::vector<double> L(256), W(256);
  hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  hipMemcpy(L.data(), dL, 2048, hipMemcpyDeviceToHost);
  hipMemcpy(W.data(), dW, 2048, hipMemcpyDeviceToHost);
  // residuals: L L^T - A and W L - I   (L[c*16 + r] column-major, W[col*16 + row]: W[row][col] at W[col*16+row])
  double e1 = 0, e2 = 0;
  for (int r = 0; r < 16; ++r) for (int c2 = 0; c2 <= r; ++c2) {
    double s = 0; for (int q = 0; q < 16; ++q) s += L[q * 16 + r] * L[q * 16 + c2];
    e1 = fmax(e1, fabs(s - A[c2 * 16 + r]));
    double s2 = 0; for (int q = 0; q < 16; ++q) s2 += W[q * 16 + r] * L[c2 * 16 + q];   // (W L)[r][c2]
    e2 = fmax(e2, fabs(s2 - (r == c2 ? 1.0 : 0.0)));
  }
  printf("variant %d: %.1f ns per block (s_memtime 100 MHz)   |LL^T-A| %.2e   |WL-I| %.2e\n", V, c * 10.0 / iters, e1, e2);
  // one wave per CU (the latency schedule's situation: the factor chain is alone on its SIMD)
  kbench<V><<<256, 64>>>(dA, dL, dW, dc, iters);
  hipDeviceSynchronize();
  hipEventRecord(ev0);
  kbench<V><<<256, 64>>>(dA, dL, dW, dc, iters);
  hipEventRecord(ev1);
  hipEventSynchronize(ev1);
  hipEventElapsedTime(&ms, ev0, ev1);
  printf("variant %d: lone wave per CU: %.1f ns per block wall\n", V, ms * 1e6 / iters);
}

int main() {
  std::vector<double> A(256);
  for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) A[c * 16 + r] = exp(-0.05 * (r - c) * (r - c)) + (r == c ? 0.3 : 0.0);
  double *dA, *dL, *dW; long long *dc;
  hipMalloc(&dA, 2048); hipMalloc(&dL, 2048); hipMalloc(&dW, 2048); hipMalloc(&dc, 8);
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
  run<0>(dA, dL, dW, dc, A);
  run<1>(dA, dL, dW, dc, A);
  run<2>(dA, dL, dW, dc, A);
  run<3>(dA, dL, dW, dc, A);
  run<4>(dA, dL, dW, dc, A);
  run<5>(dA, dL, dW, dc, A);
  run<6>(dA, dL, dW, dc, A);
  return 0;
}
