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
// Both sweeps are fused into one pass over L in 16-column panels: wave 0 does the sequential part
// of a panel (16x16 diagonal block in registers, DPP row_newbcast broadcasts), then every thread applies
// the panel's 16 rotations and the substitution update to its own rows below.  L is read and
// written exactly once per tick: ~ n^2/2 * 8 B * 2 of HBM/L2 traffic, the bound of this kernel.
// Storage: column-major, capacity 2N x 2N; the window origin slides down the diagonal and is moved
// back every N ticks.
#pragma once
#include "cgp_kernels.hpp"

namespace cgp {

#ifndef WIN_WPB
#define WIN_WPB 16
#endif
constexpr int WPB = WIN_WPB;  // panel width of the window sweep
static_assert(WPB == 16, "phase A broadcasts with DPP row_newbcast: one 16-lane row = one diagonal block");
#ifndef WIN_OCC
#define WIN_OCC 2
#endif

#ifndef WIN_BPREFETCH
#define WIN_BPREFETCH 0  // sweep waves: next trip's row in flight during the current one (+32 VGPRs: measured)
#endif

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

__global__ __launch_bounds__(256, WIN_OCC) void k_window_ticks(WindowArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *vv = reinterpret_cast<double *>(smem_raw);  // [N] rank-1 vector
  double *kk = vv + p.N;                               // [N] right-hand side of the append solve
  double *ll = kk + p.N;                               // [N] solution l
  double *cs = ll + p.N;                               // [2][2][WPB] c, s of a panel's rotations, double-buffered
  double *xn = cs + 4 * WPB;                           // [MAXD] the incoming point
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
    // sum of log(diag) without a log on the serial path: every lane of wave 0's first row keeps the running
    // product of its diagonal entries as (mantissa in [0.5, 1), exponent); ONE log per lane and tick at the end
    double pmant = 1.0;
    int pexp = 0;
    __syncthreads();

    // Panel pipeline.  Per 16-column panel the work is  A(p): wave 0 rotates and solves the 16x16
    // diagonal block (serial, in registers)  and  B(p): every row below takes the panel's 16
    // rotations and the substitution update.  A(p+1) only needs B(p) on the 16 rows of the next
    // diagonal block, so wave 0 applies B(p) to those rows itself and goes straight on to A(p+1) while
    // waves 1-3 sweep the remaining rows: one barrier per panel, serial part and sweep side by side.
    // (c, s) of a panel live in cs[panel & 1] (double buffer), l in ll[p0 ..].
    const int i = lane & (WPB - 1);
    const int npan = (n2 + WPB - 1) / WPB;
    double vi = 0, ki = 0, zi = 0;  // wave 0: the diagonal block's rows of v, k, z (registers across panels)
    // Addressing of the factor: a buffer descriptor on the window's slab + one 32-bit byte offset per lane
    // + one scalar offset per column, instead of 64-bit pointer arithmetic per entry (the sweep is bound by
    // instruction issue, not by HBM: DESIGN.md section 9).
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(L, 0, (int)((size_t)CAP * CAP * sizeof(double)), 0x00020000);
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    auto ld64 = [&](unsigned off, int soff) {
      const u2 q = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, soff, 0);
      return __hiloint2double((int)q[1], (int)q[0]);
    };
    auto st64 = [&](double x, unsigned off, int soff) {
      u2 q;
      q[0] = (unsigned)__double2loint(x);
      q[1] = (unsigned)__double2hiint(x);
      __builtin_amdgcn_raw_buffer_store_b64(q, rsrc, off, soff, 0);
    };
    const int colb = CAP * (int)sizeof(double);  // bytes between columns
    // the 16x16 diagonal block of panel p0 (lane = row), identity-padded; issued early by the caller
    auto load_diag = [&](int p0, int nb, double (&a)[WPB]) {
      const unsigned off = (unsigned)(((o2 + p0) * CAP + o2 + p0 + i) * (int)sizeof(double));
#pragma unroll
      for (int j = 0; j < WPB; ++j) a[j] = (j <= i && i < nb && j < nb) ? ld64(off, j * colb) : (i == j ? 1.0 : 0.0);
      zi = i < nb ? z[o2 + p0 + i] : 0.0;
    };
    auto phase_a = [&](int p0, int nb, double *csb, double (&a)[WPB]) {
      // Givens rotation (c, s) = (l_jj, v_j) / sqrt(l_jj^2 + v_j^2) that folds v_j into the diagonal:
      // one hardware-seeded rsqrt with a third-order step; this chain is the serial critical path
      // of the tick.  Broadcasts of lane j's values are 64-bit DPP row_newbcast moves (the four
      // 16-lane rows hold identical copies).
      double idg = 1.0;  // 1 / (new diagonal entry of this lane's row): the rotation's own rsqrt, no division
      double mine = 0.0;
      // No guard on J < nb: beyond the window the block is identity-padded with v = z = k = 0, so those steps
      // are exact no-ops (c = 1, s = 0) -- 32 fewer branches per panel on the serial path.
      static_for<0, WPB>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        {
          const double ljj = mov_bcast<J>(a[J]), vj = mov_bcast<J>(vi), zj = mov_bcast<J>(zi);
          const double r2 = __builtin_fma(vj, vj, ljj * ljj);
          const double ri = rsqrt3(r2);
          const double c = ljj * ri, sn = vj * ri;
          const double aj = a[J];
          const double tv = __builtin_fma(sn, vi, c * aj);
          const double nv = __builtin_fma(c, vi, -(sn * aj));
          if (i > J) {
            a[J] = tv;
            vi = nv;
          } else if (i == J) {
            a[J] = r2 * ri;  // sqrt(l_jj^2 + v_j^2)
            idg = ri;
          }
          const double zn = __builtin_fma(sn, vz, c * zj);
          vz = __builtin_fma(c, vz, -(sn * zj));
          if (i == J) zi = zn;
          szz += zn * zn;
          csb[J] = c;           // every lane holds the same (c, s): one uniform-address LDS write, no exec mask dance
          csb[WPB + J] = sn;
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      // diagonal of the finished block into the running product (off the critical path: nothing below reads it)
      if (lane < nb) {
        double dg = 1.0;
#pragma unroll
        for (int j = 0; j < WPB; ++j) dg = (i == j) ? a[j] : dg;
        pmant *= dg;
        pexp += __builtin_amdgcn_frexp_exp(pmant);
        pmant = __builtin_amdgcn_frexp_mant(pmant);
      }
      // forward substitution inside the block for the incoming point
      static_for<0, WPB>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        {
          const double lq = mov_bcast<Q>(ki * idg);   // 0 on the padded rows
          if (i > Q) ki = __builtin_fma(-a[Q], lq, ki);
          sl2 = __builtin_fma(lq, lq, sl2);
          fmac_bcast<Q, true>(slz, zi, lq);
          mine = (i == Q) ? lq : mine;   // lane q keeps l_q
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      if (lane < nb) ll[p0 + i] = mine;  // l of the panel: one predicated LDS write instead of sixteen
      if (lane < nb) {
        const unsigned off = (unsigned)(((o2 + p0) * CAP + o2 + p0 + i) * (int)sizeof(double));
#pragma unroll
        for (int j = 0; j < WPB; ++j)
          if (j <= i && j < nb) st64(a[j], off, j * colb);
        z[o2 + p0 + i] = zi;
      }
    };
    // one row of the sweep: a[] = the row's 16 panel columns; returns the row's updated v and k
    auto sweep_row = [&](double (&a)[WPB], double &v, double &k, const double *csb, int p0) {
#pragma unroll
      for (int j = 0; j < WPB; ++j) {
        const double c = csb[j], sn = csb[WPB + j], aj = a[j];
        const double tv = __builtin_fma(sn, v, c * aj);
        v = __builtin_fma(c, v, -(sn * aj));
        a[j] = tv;
        k = __builtin_fma(-tv, ll[p0 + j], k);
      }
    };

    if (wave == 0 && npan > 0) {
      const int nb = min(WPB, n2);
      vi = i < nb ? vv[i] : 0.0;
      ki = i < nb ? kk[i] : 0.0;
      double ad[WPB];
      load_diag(0, nb, ad);
      phase_a(0, nb, cs, ad);
    }
    for (int pi = 0; pi < npan; ++pi) {
      lds_barrier();  // A(pi) and B(pi-1) are complete.  LDS only: within a tick no thread reads factor entries another
                      // thread wrote (a panel's entries have one owner), so the sweep's HBM stores stay in flight
      const int p0 = pi * WPB;
      const double *csb = cs + (pi & 1) * 2 * WPB;
      if (wave == 0) {
        if (pi + 1 < npan) {
          // B(pi) on the rows of the next diagonal block, then A(pi + 1) with v, k still in registers
          const int nb1 = min(WPB, n2 - (p0 + WPB));
          const int r = p0 + WPB + i;
          double ad[WPB];
          load_diag(p0 + WPB, nb1, ad);  // in flight while the 16 rows below take B(pi)
          double a[WPB];
          const unsigned offr = (unsigned)(((o2 + p0) * CAP + o2 + r) * (int)sizeof(double));
#pragma unroll
          for (int j = 0; j < WPB; ++j) a[j] = i < nb1 ? ld64(offr, j * colb) : 0.0;
          vi = i < nb1 ? vv[r] : 0.0;
          ki = i < nb1 ? kk[r] : 0.0;
          sweep_row(a, vi, ki, csb, p0);
          if (lane < nb1) {
#pragma unroll
            for (int j = 0; j < WPB; ++j) st64(a[j], offr, j * colb);
          }
          phase_a(p0 + WPB, nb1, cs + ((pi + 1) & 1) * 2 * WPB, ad);
        }
      } else {
        // ---- B(pi): rows below the next diagonal block, three waves
        for (int r = p0 + 2 * WPB + (tid - 64); r < n2; r += 192) {
          asm volatile("" ::: "memory");  // keep the panel's LDS scalars from being hoisted across rows
          const unsigned offr = (unsigned)(((o2 + p0) * CAP + o2 + r) * (int)sizeof(double));
          double a[WPB];
#pragma unroll
          for (int j = 0; j < WPB; ++j) a[j] = ld64(offr, j * colb);
          double v = vv[r], k = kk[r];
          sweep_row(a, v, k, csb, p0);
#pragma unroll
          for (int j = 0; j < WPB; ++j) st64(a[j], offr, j * colb);
          vv[r] = v;
          kk[r] = k;
        }
      }
    }
    __syncthreads();

    // ---- append the new sample as the last row of the factor
    if (wave == 0) {
      double lg = log(pmant) + (double)pexp * 0.6931471805599453;  // lanes that never multiplied: log(1) + 0
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) lg += __shfl_xor(lg, off);
      slog = lg;
    }
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
