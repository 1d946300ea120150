// cgp_refine.hpp -- mixed-precision iterative refinement of alpha = Ky^-1 y and of the predictive mean of an fp32 fit.
//
// What it is for: on dense low-dimensional windows the single-precision factorisation leaves logML and
// the predictive variance at single-precision LAPACK's level, but the predictive MEAN -- mean = (L^-1 K*)^T (L^-1 y), both
// factors carrying the forward error of the tile solves amplified by the cancellation of the Schur complements -- scatters
// around 1e-3 (tests/fuzz/d1_fp32_error.py).  The classical remedy: keep the fp32 factor as the SOLVER and take the residual in
// double precision from a Gram matrix that is never stored,
//     alpha_0 = L^-T z                      z = the y row of the factor panel                    k_refine_solve, mode 0
//     r       = y - Ky alpha                Ky entries evaluated in fp64 from X on the fly       k_refine_gemv<.., false>
//     delta   = L^-T L^-1 r                 fp32 factor and W_t = L(t,t)^-1 images, fp64 sums    k_refine_solve, mode 1
//     alpha  += delta                       kept in fp64
//     mean    = K*^T alpha                  K* entries in fp64 on the fly                         k_refine_gemv<.., true>
// (reference behaviour being matched: gp_slip_node.py:48 `m.predict` -> mu = k*^T alpha, GPy's woodbury_vector.)
// Measured (tests/fuzz/d1_fp32_error.py, the windows round 5's sweep flagged): mean error 1.0e-3 -> 5e-7 (d = 1), 7e-4 -> 1.5e-7
// (d = 2), 3e-4 -> 1e-7 (d = 3) in ONE step; variance and logML are the factor's (single-precision LAPACK's level).
// Cost: N^2 + M N covariance entries in fp64 on the VALU (25 instructions each at d = 1) and three passes over the factor:
// +40 % on an fp32 fit + predict of N = 1024, M = 599 (tools/refine_cost.py), which is why the engine only takes it where it is
// needed (cgp_set_refine: every fit at d <= 3, the fits k_finalize marks as dense beyond -- RF_RHO).
#pragma once
#include "cgp_kernels.hpp"

namespace cgp {

struct RefineArgs {
  double *r;       // [batch][stride] residual / right-hand side of the correction solve
  double *alpha;   // [batch][stride] alpha in double precision
  size_t stride;   // elements per fit (NTmax * 128)
  const int *flag; // [batch] or null: only the fits marked by k_finalize (FitArgs::rflag) are worked on
};

constexpr int RF_ROWS = 64;   // rows (training points: residual; test points: mean) per workgroup, one per lane
constexpr int RF_CW = 256;    // columns staged per chunk

// One covariance-weighted row sum  out_i = sum_c k(x_i, x_c) a_c  per lane, columns dealt to the four waves.
//   MEAN = false: rows are the training points, out = the residual  r_i = y_i - sum_c Ky_ic a_c  (diagonal addend included)
//   MEAN = true : rows are the test points,     out = the predictive mean
// DD = compiled input dimension (0: runtime p.d), BROWN = the reference's RBF x Brownian product kernel (d == 1).
// Every entry is evaluated from coordinate DIFFERENCES in fp64 (inputs are the fp32 device copies, exact in double).
template <typename T, int DD, bool BROWN, bool MEAN>
__global__ __launch_bounds__(256) void k_refine_gemv(FitArgs p, RefineArgs q) {
  constexpr int DM = DD ? DD : MAXD;
  __shared__ double xc[DM][RF_CW];
  __shared__ double ac[RF_CW];
  __shared__ double red[4][RF_ROWS];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  if (q.flag && !q.flag[b]) return;   // (workgroup-uniform)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = p.N, M = p.M, d = DD ? DD : p.d;
  const double *__restrict__ pr = p.prep + (size_t)b * PREP_N;
  const T *__restrict__ Xb = reinterpret_cast<const T *>(p.X) + (size_t)b * d * N;
  const T *__restrict__ Rb = MEAN ? reinterpret_cast<const T *>(p.Xs) + (size_t)b * d * M : Xb;
  const int rows = MEAN ? M : N;
  const int row = blockIdx.x * RF_ROWS + lane;
  const bool rok = row < rows;
  ExpC ec;
  ec.load();
  double xr[DM];
#pragma unroll
  for (int j = 0; j < DM; ++j) {
    // BROWN keeps the raw coordinate (the Brownian factor needs it); the SE kernels pre-scale by 1 / ell
    const double v = (j < d && rok) ? (double)Rb[(size_t)j * rows + row] : 0.0;
    xr[j] = BROWN ? v : v * pr[j];
  }
  const double inv_ell = pr[0], amp = pr[9], amp_b = pr[10], diag_add = pr[11];
  const double *__restrict__ a64 = q.alpha + (size_t)b * q.stride;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int c0 = 0; c0 < N; c0 += RF_CW) {
    {
      const int c = c0 + tid;
      const bool cok = c < N;
#pragma unroll
      for (int j = 0; j < DM; ++j) {
        const double v = (j < d && cok) ? (double)Xb[(size_t)j * N + c] : 0.0;
        xc[j][tid] = BROWN ? v : v * pr[j];
      }
      ac[tid] = cok ? a64[c] : 0.0;
    }
    __syncthreads();
    // wave w takes columns w, w + 4, ... of the chunk: every LDS read below is a broadcast (one address per wave)
#pragma unroll 1
    for (int cc = wave; cc < RF_CW; cc += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cl = cc + 4 * u;
        double s = 0.0, kb = 1.0;
        if constexpr (BROWN) {
          const double x = xr[0], xp = xc[0][cl];
          const double df = (x - xp) * inv_ell;
          s = df * df;
          const int sx = (x > 0.0) - (x < 0.0), sp = (xp > 0.0) - (xp < 0.0);
          kb = (sx == sp) ? amp_b * fmin(fabs(x), fabs(xp)) : 0.0;
        } else {
#pragma unroll
          for (int j = 0; j < DM; ++j) {
            if (j < d) {
              const double df = xr[j] - xc[j][cl];
              s = __builtin_fma(df, df, s);
            }
          }
        }
        const double e = exp_nonpos(-0.5 * s, ec);
        acc[u] = __builtin_fma(BROWN ? e * kb : e, ac[cl], acc[u]);
      }
    }
    __syncthreads();
  }
  red[wave][lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (wave == 0 && rok) {
    const double ka = amp * ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
    if constexpr (MEAN) {
      reinterpret_cast<T *>(p.mean)[(size_t)b * M + row] = (T)ka;
    } else {
      const double ar = a64[row];
      const double yv = (double)reinterpret_cast<const T *>(p.y)[(size_t)b * N + row];
      q.r[(size_t)b * q.stride + row] = yv - ka - diag_add * ar;
    }
  }
}

// The triangular solves through the fp32 factor panel and the W_t = L(t,t)^-1 images (negated block image, WIMG), every sum in
// fp64.  One workgroup of 1024 threads per fit -- these solves are chains of NT tile steps, each step a matrix-vector product
// with a slab of the factor, and a lone 256-thread workgroup with a handful of loads in flight per thread ran them at the
// latency of one HBM round trip per 16 columns (k_alpha: 218 us for N = 1024; first version of this kernel: 424 us).  Here a
// tile step has a thousand 16-byte loads of its slab in flight at once (1024 threads x four) and the W image of the step is staged
// into LDS, transposed and padded, while the slab streams.
//   mode 0:  alpha = L^-T z            z = the y row of the factor panel (what k_alpha computes, kept in double)
//   mode 1:  alpha += L^-T L^-1 r      the correction solve of a refinement step
//   forward,  tile t:  rhs = r_t - L(t, < t) w(< t)            thread = (4 rows, one of 32 column groups)
//                      w_t = W_t rhs
//   backward, tile t:  rhs = w_t - L(> t, t)^T delta(> t)      wave = 8 columns, lanes over rows (16-byte loads), shuffle reduction
//                      delta_t = W_t^T rhs                       (overwrites w_t: nothing reads it again)
constexpr int RS_THREADS = 1024;
#ifndef CGP_RS_MINW
#define CGP_RS_MINW 4   // waves per SIMD the solve kernel is compiled for: 4 = one workgroup per CU, no spill; 8 = two per CU at 64 VGPRs with 14 spilled -- measured equal or 1-4 % slower (`make variant` A/B, tools/refine_cost.py)
#endif
// LDS copy of the W image: the 36 lower 16x16 blocks as in HBM ([q][c] = -W[cb 16 + c][qb 16 + q]) with rows padded to 17 --
// 39 KB, so that two workgroups share a CU (64 KB each at N = 1024).  Bank = (16 blk + 17 q + c) mod 64: the forward product
// (lanes = 16 rows c of four consecutive row blocks cb: blk mod 4 distinct) and the backward one (lanes = 16 columns q of four
// consecutive column blocks qb) are both conflict-free.
constexpr int RS_WB = DB * (DB + 1);   // floats per padded block
__host__ __device__ inline size_t refine_solve_lds_bytes(int NT) {
  return (size_t)(NT * TS + TS) * sizeof(double) + (size_t)16 * TS * sizeof(double) + (size_t)(WIMG / (DB * DB)) * RS_WB * sizeof(float);
}
constexpr int kRefineMaxNT = (160 * 1024 - (TS + 16 * TS) * 8 - (WIMG / (DB * DB)) * RS_WB * 4) / (TS * 8);   // block steps the LDS holds

template <typename T>
__device__ __forceinline__ void rs_stage_w(const T *__restrict__ Wt, float *__restrict__ WB, int tid) {
  for (int e = tid; e < WIMG / 4; e += RS_THREADS) {   // 2304 groups of four entries
    const float4 v = *reinterpret_cast<const float4 *>(Wt + 4 * (size_t)e);
    float *dst = WB + (e >> 6) * RS_WB + ((e >> 2) & 15) * (DB + 1) + (e & 3) * 4;
    dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
  }
}
// -W[row][col] of the staged image
__device__ __forceinline__ float rs_w(const float *__restrict__ WB, int row, int col) {
  return WB[(wimg_blk(row >> 4, col >> 4) / (DB * DB)) * RS_WB + (col & 15) * (DB + 1) + (row & 15)];
}

// r: the right-hand side of mode 1; a64: alpha, read-modified-written (mode 0: written) -- global or LDS (the one-launch form)
template <typename T>
__device__ __forceinline__ void rs_solve_body(const FitArgs &p, int mode, int b, char *smem_raw, const double *r, double *a64) {
  static_assert(sizeof(T) == 4, "the refinement is for the fp32 factor");
  const int NP = p.NT * TS, N = p.N;
  double *w = reinterpret_cast<double *>(smem_raw);   // [NP]
  double *rhs = w + NP;                               // [128]
  double *red = rhs + TS;                             // [16][128]
  float *WT = reinterpret_cast<float *>(red + 16 * TS);   // W image of the tile step, padded blocks (rs_stage_w)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T *__restrict__ Lw = reinterpret_cast<const T *>(p.Lw) + (size_t)b * p.lw_stride;
  const T *__restrict__ Winv = reinterpret_cast<const T *>(p.Winv) + (size_t)b * p.winv_stride;
  const int ld = p.ld;
  if (mode == 0) {
    // w = z: the y row of the panel, extra row index M
    const size_t zrow = (size_t)p.NT * TS + p.M;
    for (int j = tid; j < NP; j += RS_THREADS) w[j] = (double)Lw[(size_t)j * ld + zrow];
    __syncthreads();
  } else {
    const int rq = tid & 31, cgp = tid >> 5;   // rows c0 + 4 rq .. + 3, columns cgp, cgp + 32, ...
    for (int tb = 0; tb < p.NT; ++tb) {
      const int c0 = tb * TS;
      const T *__restrict__ Lr = Lw + c0 + 4 * rq;
      double s[4] = {0.0, 0.0, 0.0, 0.0};
      int c = cgp;
      for (; c + 3 * 32 < c0; c += 4 * 32) {   // four 16-byte loads in flight (64 VGPRs: two workgroups per CU hide each other's latency)
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(Lr + (size_t)(c + 32 * u) * ld);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double x = w[c + 32 * u];
          s[0] = __builtin_fma((double)v[u].x, x, s[0]);
          s[1] = __builtin_fma((double)v[u].y, x, s[1]);
          s[2] = __builtin_fma((double)v[u].z, x, s[2]);
          s[3] = __builtin_fma((double)v[u].w, x, s[3]);
        }
      }
      const T *__restrict__ Wt = Winv + (size_t)tb * WIMG;
      rs_stage_w<T>(Wt, WT, tid);
      // the two column groups of a wave first (lanes l and l ^ 32 hold the same rows), then 16 partial sums per row through LDS
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += __shfl_xor(s[j], 32);
      if (lane < 32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave * TS + 4 * rq + j] = s[j];
      }
      __syncthreads();
      if (tid < TS) {
        double a = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) a += red[g * TS + tid];
        rhs[tid] = ((c0 + tid < N) ? r[c0 + tid] : 0.0) - a;
      }
      __syncthreads();
      {
        // w_t = W_t rhs: row i, columns g, g + 8, ... <= the end of the row's diagonal block (exact zeros above the diagonal)
        const int i = tid & (TS - 1), g = tid >> 7;
        double t0 = 0.0;
        const int cend = ((i >> 4) + 1) * DB;
        for (int col = g; col < cend; col += 8) t0 = __builtin_fma((double)rs_w(WT, i, col), rhs[col], t0);
        red[g * TS + i] = t0;
      }
      __syncthreads();
      if (tid < TS) {
        double a = 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) a += red[g * TS + tid];
        w[c0 + tid] = -a;
      }
      __syncthreads();
    }
  }
  for (int tb = p.NT - 1; tb >= 0; --tb) {
    const int c0 = tb * TS, rbelow = c0 + TS;
    const T *__restrict__ Wt = Winv + (size_t)tb * WIMG;
    rs_stage_w<T>(Wt, WT, tid);
    // wave `wave` owns columns 8 wave .. + 7 of the tile, four at a time; a lane covers rows rbelow + 4 lane + 256 j
    for (int half = 0; half < 2; ++half) {
      const int cl = wave * 8 + half * 4;
      double s[4] = {0.0, 0.0, 0.0, 0.0};
      // four columns = four 16-byte loads in flight per lane; rows in whole groups of four (NP is a multiple of 128)
      for (int rr = rbelow + 4 * lane; rr < NP; rr += 256) {
        float4 v[4];
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) v[jc] = *reinterpret_cast<const float4 *>(Lw + (size_t)(c0 + cl + jc) * ld + rr);
        const double d0 = w[rr], d1 = w[rr + 1], d2 = w[rr + 2], d3 = w[rr + 3];
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) {
          s[jc] = __builtin_fma((double)v[jc].x, d0, s[jc]);
          s[jc] = __builtin_fma((double)v[jc].y, d1, s[jc]);
          s[jc] = __builtin_fma((double)v[jc].z, d2, s[jc]);
          s[jc] = __builtin_fma((double)v[jc].w, d3, s[jc]);
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) s[jc] += __shfl_xor(s[jc], off);
      if (lane == 0) {
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) rhs[cl + jc] = w[c0 + cl + jc] - s[jc];
      }
    }
    __syncthreads();
    {
      // delta_t = W_t^T rhs: column i, rows g, g + 8, ... from the start of the column's diagonal block
      const int i = tid & (TS - 1), g = tid >> 7;
      double t0 = 0.0;
      for (int row = (i & ~15) + g; row < TS; row += 8) t0 = __builtin_fma((double)rs_w(WT, row, i), rhs[row], t0);
      red[g * TS + i] = t0;
    }
    __syncthreads();
    if (tid < TS) {
      double a = 0.0;
#pragma unroll
      for (int g = 0; g < 8; ++g) a += red[g * TS + tid];
      w[c0 + tid] = -a;
    }
    __syncthreads();
  }
  for (int j = tid; j < NP; j += RS_THREADS) a64[j] = (j < N) ? (mode == 0 ? 0.0 : a64[j]) + w[j] : 0.0;
}
template <typename T>
__global__ __launch_bounds__(RS_THREADS, CGP_RS_MINW) void k_refine_solve(FitArgs p, RefineArgs q, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int b = blockIdx.x;
  if (q.flag && !q.flag[b]) return;   // (workgroup-uniform)
  rs_solve_body<T>(p, mode, b, smem_raw, q.r + (size_t)b * q.stride, q.alpha + (size_t)b * q.stride);
}

// --------------------------------------------------------------------------------------------------
// The whole refinement of ONE fit in one workgroup: what a call of more than three input dimensions launches under the default
// setting, where k_finalize marks the dense fits (FitArgs::rflag) and almost always marks none -- ONE launch whose workgroups
// return at once instead of four (BASELINE configs[2]: +1.5 us per call instead of +6).  A marked fit runs the four stages back to
// back on its CU: alpha and r live in LDS between them, the covariance sums take one row per thread (1024 rows per pass, every column
// chunk staged once): ~270 us at N = 1024, M = 599 against ~250 us of the four-launch form -- the stages of the one fit are a chain
// either way.  Runtime input dimension, SE kernels only (d <= 3 and the RBF x Brownian kernel take the four launches, every fit).
// --------------------------------------------------------------------------------------------------
inline size_t refine_gated_lds_bytes(int NT) { return refine_solve_lds_bytes(NT) + (size_t)2 * NT * TS * sizeof(double); }
constexpr int kRefineGatedMaxNT = (160 * 1024 - (TS + 16 * TS) * 8 - (WIMG / (DB * DB)) * RS_WB * 4) / (3 * TS * 8);
template <typename T, bool MEAN>
__device__ __forceinline__ void rf_rowsum_wg(const FitArgs &p, int b, double *stage, const double *alpha, double *r_out) {
  // stage: [MAXD][RF_CW] scaled column points + [RF_CW] alpha (the solve's reduction / W-image area, free between the solves)
  const int tid = threadIdx.x, N = p.N, M = p.M, d = p.d;
  const int rows = MEAN ? M : N;
  const double *__restrict__ pr = p.prep + (size_t)b * PREP_N;
  const T *__restrict__ Xb = reinterpret_cast<const T *>(p.X) + (size_t)b * d * N;
  const T *__restrict__ Rb = MEAN ? reinterpret_cast<const T *>(p.Xs) + (size_t)b * d * M : Xb;
  double *xc = stage, *ac = stage + MAXD * RF_CW;
  ExpC ec;
  ec.load();
  const double amp = pr[9], diag_add = pr[11];
  for (int row0 = 0; row0 < rows; row0 += RS_THREADS) {
    const int row = row0 + tid;
    const bool rok = row < rows;
    double xr[MAXD];
#pragma unroll
    for (int j = 0; j < MAXD; ++j) xr[j] = (j < d && rok) ? (double)Rb[(size_t)j * rows + row] * pr[j] : 0.0;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int c0 = 0; c0 < N; c0 += RF_CW) {
      __syncthreads();
      if (tid < RF_CW) {
        const int c = c0 + tid;
        const bool cok = c < N;
#pragma unroll
        for (int j = 0; j < MAXD; ++j) xc[j * RF_CW + tid] = (j < d && cok) ? (double)Xb[(size_t)j * N + c] * pr[j] : 0.0;
        ac[tid] = cok ? alpha[c] : 0.0;
      }
      __syncthreads();
#pragma unroll 1
      for (int cc = 0; cc < RF_CW; cc += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          double sq = 0.0;
#pragma unroll
          for (int j = 0; j < MAXD; ++j) {
            if (j < d) {
              const double df = xr[j] - xc[j * RF_CW + cc + u];
              sq = __builtin_fma(df, df, sq);
            }
          }
          acc[u] = __builtin_fma(exp_nonpos(-0.5 * sq, ec), ac[cc + u], acc[u]);
        }
      }
    }
    if (rok) {
      const double ka = amp * ((acc[0] + acc[1]) + (acc[2] + acc[3]));
      if constexpr (MEAN) reinterpret_cast<T *>(p.mean)[(size_t)b * M + row] = (T)ka;
      else r_out[row] = (double)reinterpret_cast<const T *>(p.y)[(size_t)b * N + row] - ka - diag_add * alpha[row];
    }
  }
  __syncthreads();
}
template <typename T>
__global__ __launch_bounds__(RS_THREADS, CGP_RS_MINW) void k_refine_gated(FitArgs p, RefineArgs q, int steps, int want_mean) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int b = blockIdx.x;
  if (q.flag && !q.flag[b]) return;   // (workgroup-uniform)
  const int NP = p.NT * TS;
  double *al = reinterpret_cast<double *>(smem_raw + refine_solve_lds_bytes(p.NT));   // [NP] alpha, then [NP] r, behind the solve's arrays
  double *rr = al + NP;
  double *stage = reinterpret_cast<double *>(smem_raw) + NP + TS;                     // the solve's red / W-image area
  static_assert((MAXD + 1) * RF_CW * 8 <= 16 * TS * 8 + (WIMG / (DB * DB)) * RS_WB * 4, "the column stage fits the solve's scratch");
  rs_solve_body<T>(p, 0, b, smem_raw, rr, al);
  __syncthreads();
  for (int st = 0; st < steps; ++st) {
    rf_rowsum_wg<T, false>(p, b, stage, al, rr);
    rs_solve_body<T>(p, 1, b, smem_raw, rr, al);
    __syncthreads();
  }
  if (want_mean && p.M > 0) rf_rowsum_wg<T, true>(p, b, stage, al, nullptr);
  double *a64 = q.alpha + (size_t)b * q.stride;
  for (int j = threadIdx.x; j < NP; j += RS_THREADS) a64[j] = al[j];
}

}  // namespace cgp
