// cgp_window.hpp -- online sliding-window GP (BASELINE.json configs[3]): per odometry tick the oldest
// sample leaves the window and a new one enters, and the Cholesky factor is maintained by a rank-1
// update instead of a refit.  Not reference behaviour (the reference refits once per 150-tick
// window, CoreNav.cpp:289-305); the oracle for it is "refit from scratch on the current window".
//
// One workgroup owns one window and walks a whole block of ticks inside ONE launch (state lives in
// HBM: no per-tick launch, no host round trip).  Per tick, with the window [x_0 .. x_{n-1}]:
//   drop x_0 :  Ky = [[a, b'],[b, C]] = L L'  ->  chol(C) = chol(L22 L22' + l21 l21')   rank-1 UPDATE
//               by Givens-like rotations (c_j, s_j) column by column; z = L^-1 y rides along as one
//               more row.
//   add x_new:  l = L^-1 k(X, x_new) (forward substitution), d = sqrt(k** + noise - |l|^2),
//               z_new = (y_new - l'z)/d.  l'z and k** - |l|^2 are also the one-step-ahead
//               predictive mean / variance of y_new BEFORE it is added, so they are the tick's output.
// Both sweeps are fused into one pass over L in 32-column panels: wave 0 does the sequential part
// of a panel (32x32 diagonal block in registers, v_readlane broadcasts), then every thread applies
// the panel's 32 rotations and the substitution update to its own rows below.  L is read and
// written exactly once per tick: ~ n^2/2 * 8 B * 2 of HBM/L2 traffic, the bound of this kernel.
// Storage: column-major, capacity 2N x 2N; the window origin slides down the diagonal and is moved
// back every N ticks.
#pragma once
#include "cgp_kernels.hpp"

namespace cgp {

constexpr int WPB = 32;  // panel width of the window sweep

struct WindowArgs {
  double *L;        // [nwin][CAP*CAP] column-major
  double *z;        // [nwin][CAP]
  double *xw;       // [nwin][d][CAP] window inputs (same index space as L)
  double *yw;       // [nwin][CAP]
  int *state;       // [nwin][4] = {origin, n, info, ticks_done}
  const double *prep;   // [nwin][PREP_N]
  const double *theta;  // [nwin][MAX_THETA]
  const double *xs;     // [nwin][T][d] the block of ticks
  const double *ys;     // [nwin][T]
  double *pred_mean, *pred_var, *logml;  // [nwin][T]
  int N, CAP, d, kernel_id, T, include_noise;
};

// Covariance of two points (raw coordinates), direct formulas.
__device__ __forceinline__ double win_cov(int kid, int d, const double *pr, const double *a, int as, const double *b,
                                          int bs, bool same) {
  if (kid != K_RBF_BROWNIAN) {
    double d2 = 0;
    for (int q = 0; q < d; ++q) {
      const double df = (a[q * as] - b[q * bs]) * pr[q];
      d2 += df * df;
    }
    return pr[9] * exp(-0.5 * d2);
  }
  const double x = a[0], xp = b[0];
  double r2 = same ? 0.0 : (-2.0 * x * xp + (x * x + xp * xp));
  r2 = r2 < 0.0 ? 0.0 : r2;
  const double rr = sqrt(r2) * pr[0];
  const int sx = (x > 0) - (x < 0), sp = (xp > 0) - (xp < 0);
  const double kb = (sx == sp) ? pr[10] * fmin(fabs(x), fabs(xp)) : 0.0;
  return pr[9] * exp(-0.5 * rr * rr) * kb;
}

__global__ __launch_bounds__(256, 2) void k_window_ticks(WindowArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *vv = reinterpret_cast<double *>(smem_raw);  // [N] rank-1 vector
  double *kk = vv + p.N;                               // [N] right-hand side of the append solve
  double *ll = kk + p.N;                               // [N] solution l
  double *cs = ll + p.N;                               // [3][WPB] c, s, 1/c of the current panel
  double *xn = cs + 3 * WPB;                           // [MAXD] the incoming point
  double *red = xn + MAXD;                             // [8] scalars handed from wave 0 to the block
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = p.N, CAP = p.CAP, d = p.d, kid = p.kernel_id;
  double *L = p.L + (size_t)w * CAP * CAP;
  double *z = p.z + (size_t)w * CAP;
  double *xw = p.xw + (size_t)w * d * CAP;
  double *yw = p.yw + (size_t)w * CAP;
  int *st = p.state + w * 4;
  const double *pr = p.prep + (size_t)w * PREP_N;
  const double *th = p.theta + (size_t)w * MAX_THETA;
  const int nth = (kid == K_SE_ISO) ? 3 : (kid == K_SE_ARD ? d + 2 : 4);
  const double noise = th[nth - 1];
  int o = st[0], n = st[1], bad = st[2];

  for (int t = 0; t < p.T; ++t) {
    // ---- make room: move the window back to the origin when it reached the end of the buffer
    if (o + n >= CAP) {
      for (int c = 0; c < n; ++c)
        for (int r = c + tid; r < n; r += 256) L[(size_t)c * CAP + r] = L[(size_t)(o + c) * CAP + o + r];
      for (int i = tid; i < n; i += 256) {
        z[i] = z[o + i];
        yw[i] = yw[o + i];
        for (int q = 0; q < d; ++q) xw[q * CAP + i] = xw[q * CAP + o + i];
      }
      __syncthreads();
      o = 0;
    }
    const bool drop = n >= N;
    const int o2 = drop ? o + 1 : o, n2 = drop ? n - 1 : n;  // window after the drop
    if (tid < d) xn[tid] = p.xs[((size_t)w * p.T + t) * d + tid];
    __syncthreads();
    const double ynew = p.ys[(size_t)w * p.T + t];
    for (int i = tid; i < n2; i += 256) {
      vv[i] = drop ? L[(size_t)o * CAP + o2 + i] : 0.0;
      kk[i] = win_cov(kid, d, pr, xw + o2 + i, CAP, xn, 1, false);
    }
    double vz = drop ? z[o] : 0.0;  // the dropped sample's component of z (uniform)
    double sl2 = 0, slz = 0, slog = 0, szz = 0;
    __syncthreads();

    for (int p0 = 0; p0 < n2; p0 += WPB) {
      const int nb = min(WPB, n2 - p0);
      double *Lp = L + (size_t)(o2 + p0) * CAP + o2;  // column p0 of the window, row 0 of the window
      if (wave == 0) {
        // ---- phase A: 32x32 diagonal block, lane = row (replicated in the upper half-wave)
        const int i = lane & 31;
        double a[WPB];
#pragma unroll
        for (int j = 0; j < WPB; ++j) a[j] = (j <= i && i < nb && j < nb) ? Lp[(size_t)j * CAP + p0 + i] : (i == j ? 1.0 : 0.0);
        double vi = i < nb ? vv[p0 + i] : 0.0, ki = i < nb ? kk[p0 + i] : 0.0, zi = i < nb ? z[o2 + p0 + i] : 0.0;
#pragma unroll
        for (int j = 0; j < WPB; ++j) {
          if (j < nb) {
            // rotation (c, s) that folds v_j into the diagonal: one reciprocal and one rsqrt, both
            // hardware-seeded + Newton (this chain is the serial critical path of the tick)
            const double ljj = rdlane(a[j], j), vj = rdlane(vi, j);
            double il = __builtin_amdgcn_rcp(ljj);
            il = __builtin_fma(il, __builtin_fma(-ljj, il, 1.0), il);
            il = __builtin_fma(il, __builtin_fma(-ljj, il, 1.0), il);
            const double s = vj * il;
            const double q2 = __builtin_fma(s, s, 1.0);
            const double ci = Prec<double>::rsqrt_(q2);
            const double c = q2 * ci;
            const double tv = (a[j] + s * vi) * ci;
            if (i > j) {
              vi = c * vi - s * tv;
              a[j] = tv;
            } else if (i == j) a[j] = ljj * c;
            const double zj = rdlane(zi, j);
            const double zn = (zj + s * vz) * ci;
            vz = c * vz - s * zn;
            if (i == j) zi = zn;
            szz += zn * zn;
            if (lane == 0) {
              cs[j] = c;
              cs[WPB + j] = s;
              cs[2 * WPB + j] = ci;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // diagonal of the finished block: logs and reciprocals once per lane, in parallel
        double dg = 1.0;
#pragma unroll
        for (int j = 0; j < WPB; ++j) dg = (i == j) ? a[j] : dg;
        double lg = (lane < nb) ? log(dg) : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lg += __shfl_xor(lg, off);
        slog += lg;
        const double idg = 1.0 / dg;
        // forward substitution inside the block for the incoming point
#pragma unroll
        for (int q = 0; q < WPB; ++q) {
          if (q < nb) {
            const double lq = rdlane(ki, q) * rdlane(idg, q);
            if (i > q) ki -= a[q] * lq;
            sl2 += lq * lq;
            slz += lq * rdlane(zi, q);
            if (lane == 0) ll[p0 + q] = lq;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (lane < nb) {
#pragma unroll
          for (int j = 0; j < WPB; ++j)
            if (j <= i && j < nb) Lp[(size_t)j * CAP + p0 + i] = a[j];
          z[o2 + p0 + i] = zi;
        }
      }
      __syncthreads();
      // ---- phase B: every row below the panel takes the panel's rotations and the solve update
      for (int i = p0 + WPB + tid; i < n2; i += 256) {
        asm volatile("" ::: "memory");  // keep the panel's 128 LDS scalars from being hoisted into registers
        double a[WPB];
#pragma unroll
        for (int j = 0; j < WPB; ++j) a[j] = Lp[(size_t)j * CAP + i];
        double vi = vv[i], ki = kk[i];
#pragma unroll
        for (int j = 0; j < WPB; ++j) {
          const double tv = (a[j] + cs[WPB + j] * vi) * cs[2 * WPB + j];
          vi = cs[j] * vi - cs[WPB + j] * tv;
          a[j] = tv;
          ki -= tv * ll[p0 + j];
        }
#pragma unroll
        for (int j = 0; j < WPB; ++j) Lp[(size_t)j * CAP + i] = a[j];
        vv[i] = vi;
        kk[i] = ki;
      }
      __syncthreads();
    }

    // ---- append the new sample as the last row of the factor
    if (tid == 0) {
      red[0] = sl2; red[1] = slz; red[2] = slog; red[3] = szz;
    }
    __syncthreads();
    sl2 = red[0]; slz = red[1]; slog = red[2]; szz = red[3];
    const double kss = (kid == K_RBF_BROWNIAN) ? pr[9] * pr[10] * fabs(xn[0]) : pr[9];
    double d2 = kss + noise + 1e-8 - sl2;
    if (!(d2 > 0.0)) {
      if (bad == 0) bad = t + 1;
      d2 = 1e-300;
    }
    const double dd = sqrt(d2);
    const double znew = (ynew - slz) / dd;
    double *Lrow = L + (size_t)o2 * CAP + o2 + n2;  // row n2 of the window, column 0
    for (int j = tid; j < n2; j += 256) Lrow[(size_t)j * CAP] = ll[j];
    if (tid == 0) {
      Lrow[(size_t)n2 * CAP] = dd;
      z[o2 + n2] = znew;
      yw[o2 + n2] = ynew;
      for (int q = 0; q < d; ++q) xw[q * CAP + o2 + n2] = xn[q];
      const size_t oi = (size_t)w * p.T + t;
      double pv = kss - sl2;
      pv = pv < 1e-15 ? 1e-15 : pv;
      p.pred_mean[oi] = slz;
      p.pred_var[oi] = p.include_noise ? pv + noise : pv;
      p.logml[oi] = -0.5 * (szz + znew * znew) - (slog + log(dd)) - 0.5 * (double)(n2 + 1) * 1.8378770664093453;
    }
    o = o2;
    n = n2 + 1;
    __syncthreads();
  }
  if (tid == 0) {
    st[0] = o;
    st[1] = n;
    st[2] = bad;
    st[3] += p.T;
  }
}

}  // namespace cgp
