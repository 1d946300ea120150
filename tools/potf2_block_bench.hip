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
  factor_block16<double>(a, w, bad, 0, l15, [] {});   // the shipped code (= variant 8 + the repairing fallback)
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

// V7: the four 16-lane rows stop computing four copies.  Row 0 of the wave factors, row 1 runs the inverse's forward
// substitution AT THE SAME TIME with the SAME instructions: factor step J, column C and inverse step J, element C are both
// "x[C] += (row C of L's column J) * own" -- same broadcast lane, same broadcast source once column J of L is in both
// rows (one v_permlane16_swap per 32-bit half, gfx950), only the accumulator (a[C] / t[C]) and `own` (-L_rJ / -w[J])
// differ, and those are per-lane registers anyway.  120 DPP fmacs instead of 240.  Rows 2, 3 mirror rows 0, 1.
__device__ __forceinline__ double merge_rows(double v) {   // rows 1, 3 take the value rows 0, 2 hold
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]);
}
template <> __device__ __forceinline__ void variant<7>(double (&x)[DB], double (&w)[DB], int &bad, int l15) {
  const bool inv = (threadIdx.x >> 4) & 1;
#pragma unroll
  for (int i = 0; i < DB; ++i) x[i] = inv ? ((i == l15) ? 1.0 : 0.0) : x[i];
  double lp = 0.0, nlp = 0.0;
  static_for<0, DB>([&](auto jc) {
    constexpr int J = decltype(jc)::value;
    const double colj = merge_rows(x[J]);
    double dj = mov_bcast<J>(colj);
    if constexpr (J > 0) {
      static_for<J + 1, DB>([&](auto cc) {
        constexpr int C = decltype(cc)::value;
        fmac_bcast<C, false>(x[C], lp, nlp);
      });
    }
    const bool ok = dj > 0.0;
    if (!ok && bad == 0) bad = J + 1;
    dj = ok ? dj : 1.0;
    const double rs = rsqrt3(dj);
    const double l = (ok ? colj : ((l15 == J) ? 1.0 : colj)) * rs;            // column J of L, in every row
    x[J] = (ok ? x[J] : ((l15 == J && !inv) ? 1.0 : x[J])) * rs;              // row 0: = l; row 1: w[J] = t[J] / L_JJ
    const double own = -x[J];
    if constexpr (J + 1 < DB) fmac_bcast<J + 1, true>(x[J + 1], l, own);
    lp = l;
    nlp = own;
  });
#pragma unroll
  for (int i = 0; i < DB; ++i) w[i] = (i < l15) ? 0.0 : x[i];   // meaningful in rows 1, 3
}

// V8: instruction count is what a lone wave pays (every VALU instruction, DPP or not, is 4 cycles of issue; V6's reordering
// bought 3 %), and V4 spends more instructions around the 240 fmacs than on them.  Here: no per-pivot repair of a
// non-positive pivot (a bad pivot poisons every later one with NaN, so ONE test of the last pivot tells, and the rare
// case re-runs the repairing form), the sign of `own` as the fmac's neg modifier instead of an xor + mov per step, the
// Newton step folded into l (l = l0 + l0 q with l0 = a r, instead of a (r + r q)), no final mask (exact zeros stay zero).
template <> __device__ __forceinline__ void variant<8>(double (&a)[DB], double (&w)[DB], int &bad, int l15) {
  double t[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i) t[i] = (i == l15) ? 1.0 : 0.0;
  double lp = 0.0, wp = 0.0, last = 0.0;
  static_for<0, DB>([&](auto jc) {
    constexpr int J = decltype(jc)::value;
    const double dj = mov_bcast<J>(a[J]);
    if constexpr (J > 0) {
      static_for<J + 1, DB>([&](auto cc) {
        constexpr int C = decltype(cc)::value;
        fnmac_bcast<C, false>(a[C], lp, lp);
      });
      static_for<J, DB>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        fnmac_bcast<I, false>(t[I], lp, wp);
      });
    }
    const double r = __builtin_amdgcn_rsq(dj);
    const double l0 = a[J] * r;
    const double e = __builtin_fma(-(dj * r), r, 1.0);
    const double q = __builtin_fma(0.375, e, 0.5) * e;
    const double l = __builtin_fma(l0, q, l0);
    a[J] = l;
    if constexpr (J + 1 < DB) fnmac_bcast<J + 1, true>(a[J + 1], l, l);
    const double rs = __builtin_fma(r, q, r);
    w[J] = t[J] * rs;
    lp = l;
    wp = w[J];
    last = dj;
  });
  if (!(last > 0.0)) bad = DB;   // some pivot was not positive: the caller re-runs the repairing form on the saved block
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
  if (lane < DB && blockIdx.x == 0)
    for (int c = 0; c < DB; ++c) Lout[c * DB + l15] = (l15 >= c) ? a[c] : 0.0;
  if (lane >= (V == 7 ? DB : 0) && lane < (V == 7 ? 2 * DB : DB) && blockIdx.x == 0)
    for (int c = 0; c < DB; ++c) Wout[l15 * DB + c] = w[c];  // W[c][l15]
  if (lane == 0 && blockIdx.x == 0) cyc[0] = t1 - t0 + bad;
}

// the whole of wave 0's F phase in potf2_tile: block out of the LDP-strided tile, factor, L block and Dinv back to LDS
__global__ __launch_bounds__(64) void kbench_io(const double *A, long long *cyc, int iters) {
  __shared__ double At[DB * LDP + 2 * DB * DB];
  double *Dv = At + DB * LDP;
  const int lane = threadIdx.x, l15 = lane & 15;
  for (int i = lane; i < DB * DB; i += 64) At[(i >> 4) * LDP + (i & 15)] = A[i];
  for (int i = lane; i < DB * DB; i += 64) Dv[DB * DB + i] = A[i];
  __syncthreads();
  int bad = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    double a[DB], w[DB];
#pragma unroll
    for (int c = 0; c < DB; ++c) a[c] = At[c * LDP + l15];
    factor_block16<double>(a, w, bad, 0, l15, [&] {
#pragma unroll
      for (int c = 0; c < DB; ++c) a[c] = Dv[DB * DB + c * DB + l15];
    });
    if (lane < DB) {
#pragma unroll
      for (int i = 0; i < DB; ++i) Dv[l15 * DB + i] = w[i];
    }
    __builtin_amdgcn_s_barrier();
    // put the block back (what the next iteration factors) -- costs what the L-block store costs
    if (lane < DB) {
#pragma unroll
      for (int c = 0; c < DB; ++c) At[c * LDP + l15] = Dv[DB * DB + c * DB + l15] + 0.0 * a[c];
    }
    __builtin_amdgcn_s_barrier();
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0 && blockIdx.x == 0) cyc[0] = t1 - t0 + bad;
}

// the shipped fp32 form, lone wave per CU
__global__ __launch_bounds__(64) void kbench_f32(const double *A, float *Lout, float *Wout, int iters) {
  __shared__ float blk[DB * DB];
  const int lane = threadIdx.x, l15 = lane & 15;
  for (int i = lane; i < DB * DB; i += 64) blk[i] = (float)A[i];
  __syncthreads();
  float a[DB], w[DB];
  int bad = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < DB; ++c) a[c] = blk[c * DB + l15];
    factor_block16<float>(a, w, bad, 0, l15, [] {});
    if (it + 1 < iters) asm volatile("" ::: "memory");
  }
  if (lane < DB && blockIdx.x == 0)
    for (int c = 0; c < DB; ++c) { Lout[c * DB + l15] = (l15 >= c) ? a[c] : 0.f; Wout[l15 * DB + c] = w[c] + bad; }
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
  run<7>(dA, dL, dW, dc, A);
  run<8>(dA, dL, dW, dc, A);
  {
    const int iters = 20000;
    hipEvent_t ev0, ev1; hipEventCreate(&ev0); hipEventCreate(&ev1);
    kbench_io<<<256, 64>>>(dA, dc, iters); hipDeviceSynchronize();
    hipEventRecord(ev0); kbench_io<<<256, 64>>>(dA, dc, iters); hipEventRecord(ev1); hipEventSynchronize(ev1);
    float ms = 0; hipEventElapsedTime(&ms, ev0, ev1);
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("wave 0's F phase (load, factor, L and Dinv to LDS), lone wave per CU: %.1f ns per block, %.0f s_memtime ticks\n", ms * 1e6 / iters, (double)c / iters);
  }
  {
    const int iters = 20000;
    float *fL, *fW; hipMalloc(&fL, 1024); hipMalloc(&fW, 1024);
    hipEvent_t ev0, ev1; hipEventCreate(&ev0); hipEventCreate(&ev1);
    kbench_f32<<<256, 64>>>(dA, fL, fW, iters); hipDeviceSynchronize();
    hipEventRecord(ev0); kbench_f32<<<256, 64>>>(dA, fL, fW, iters); hipEventRecord(ev1); hipEventSynchronize(ev1);
    float ms = 0; hipEventElapsedTime(&ms, ev0, ev1);
    std::vector<float> L(256), W(256);
    hipMemcpy(L.data(), fL, 1024, hipMemcpyDeviceToHost); hipMemcpy(W.data(), fW, 1024, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int r = 0; r < 16; ++r) for (int c2 = 0; c2 <= r; ++c2) {
      double s1 = 0, s2 = 0;
      for (int q = 0; q < 16; ++q) { s1 += (double)L[q * 16 + r] * L[q * 16 + c2]; s2 += (double)W[q * 16 + r] * L[c2 * 16 + q]; }
      e1 = fmax(e1, fabs(s1 - A[c2 * 16 + r])); e2 = fmax(e2, fabs(s2 - (r == c2 ? 1.0 : 0.0)));
    }
    printf("shipped fp32 form, lone wave per CU: %.1f ns per block   |LL^T-A| %.2e   |WL-I| %.2e\n", ms * 1e6 / iters, e1, e2);
  }
  return 0;
}
