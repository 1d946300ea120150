// cgp_window.hpp -- online sliding-window GP (BASELINE.json configs[3]): per odometry tick the oldest
// sample leaves the window and a new one enters, and the Cholesky factor is maintained by a rank-1
// update instead of a refit.  Not reference behaviour (the reference refits once per 150-tick
// window, CoreNav.cpp:289-305); the oracle for it is "refit from scratch on the current window".
//
// One workgroup owns one window and walks a whole block of ticks inside ONE launch (state lives in
// HBM: no per-tick launch, no host round trip).  Per tick, with the window [x_0 .. x_{n-1}]:
//   drop x_0 :  Ky = [[a, b'],[b, C]] = L L'  ->  chol(C) = chol(L22 L22' + l21 l21')   rank-1 UPDATE;
//               z = L^-1 y is updated with it.
//   add x_new:  l = L^-1 k(X, x_new) (forward substitution), d = sqrt(k** + noise - |l|^2),
//               z_new = (y_new - l'z)/d.  l'z and k** - |l|^2 are also the one-step-ahead
//               predictive mean / variance of y_new BEFORE it is added, so they are the tick's output.
// Since the end of round 6 the update is NOT a chain of rotations: with w = L22^-1 l21 and t_j = 1 + sum_{k<=j} w_k^2 the
// rotation of column j is c_j = sqrt(t_{j-1}/t_j), s_j = w_j/sqrt(t_j), so the only serial work is a forward substitution
// with the OLD factor -- two right-hand sides, l21 and k(X, x_new) -- and everything else (c, s, the new l and z, the new
// diagonal block) follows from prefix sums, lane-parallel (k_window_ticks, solve_block of k_window_pairs).
// Both sweeps are fused into one pass over L in 16-column panels: wave 0 does the sequential part
// of a panel (16x16 diagonal block in registers, DPP row_newbcast), then every thread takes the panel's columns
// (c, s', w, q) into its own rows below.  L is read and written exactly once per tick (once per TWO ticks in
// k_window_pairs, once per FOUR in k_window_multi<4>): n^2/2 * 8 B * 2 of HBM traffic per pass.
// Storage: column-major, capacity 2N x 2N; the window origin slides down the diagonal and is moved
// back every N ticks.  The strict upper triangle of the slab is never read (the kernels store zeros / by-products there).
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

#ifndef WIN_PROBE
#define WIN_PROBE 0   // `make variant` measurement build (tools/window_tick_phases.py): k_window_ticks returns wave 0's per-phase clock sums instead of the tick's outputs
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
  int t0, nt;       // this launch handles ticks [t0, t0 + nt) of the block (k_window_pairs: nt even)
  int *info_out;    // [nwin] or nullptr: a copy of state[2] (first failing tick) where the host can read it without a copy command
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

// 64-bit DPP row shift right by S lanes inside every 16-lane row, zero fill (prefix sums over a diagonal block's lanes)
template <int S> __device__ __forceinline__ double win_row_shr(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + S, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + S, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// exclusive prefix sum over the 16 lanes of a row
__device__ __forceinline__ double win_row_scan_excl(double x) {
  double s = win_row_shr<1>(x);
  s += win_row_shr<1>(s);
  s += win_row_shr<2>(s);
  s += win_row_shr<4>(s);
  s += win_row_shr<8>(s);
  return s;
}

// The single-tick kernel.  Round 6 (end): the rank-1 update is no longer a chain of rotations.  With w = L22^-1 v (v = the dropped
// sample's column) and t_j = 1 + sum_{k<=j} w_k^2 the rotation that folds v_j into column j is c_j = sqrt(t_{j-1} / t_j),
// s_j = w_j / sqrt(t_j), and the vector it acts on is the substitution's residual u^(j) = v - L[:, :j] w[:j] over sqrt(t_{j-1}):
//   L'_ij = c_j L_ij + s'_j u^(j)_i,  s'_j = w_j / sqrt(t_j t_{j-1})            (i >= j; the diagonal comes out as L_jj / c_j)
// and with L' = L22 G, G = chol(I + w w') (semiseparable: G_jj = 1 / c_j, G_ij = w_i s'_j) the append solve and the z row follow
// from substitutions with the OLD factor, q = L22^-1 k, qy = L22^-1 y2 = z2 + w z0, and running sums:
//   l_j = (q_j - w_j S_j / t_{j-1}) c_j,  S_j = sum_{k<j} w_k q_k               (the same with qy for z')
// So the only serial work of a panel is a two-right-hand-side forward substitution of its 16 x 16 block (two DPP instructions per
// step and right-hand side); sixteen rsqrt, the prefix sums and the new block are lane-parallel.  Measured (tools/window_host_tick.py,
// one N = 512 window, T = 1 pushes): see DESIGN.md section 9.
constexpr int WIN_STG = 2 * WPB * WPB + 2 * WPB;   // doubles of one staging buffer of wave 0's inputs: 16 rows of the panel, the next diagonal block, z
template <int NTH>   // threads of the workgroup: 256 (two workgroups per CU, many windows) or 512 (no more windows than CUs: seven sweep waves per window)
__global__ __launch_bounds__(NTH, NTH == 256 ? WIN_OCC : 1) void k_window_ticks(WindowArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *vv = reinterpret_cast<double *>(smem_raw);  // [N] residual u of the substitution with v (the rank-1 vector)
  double *kk = vv + p.N;                               // [N] residual of the substitution with k (the append solve)
  double *ll = kk + p.N;                               // [N] solution l
  double *cs = ll + p.N;                               // [2][WPB][4] c, s', w, q of a panel's columns, double-buffered
  double *xn = cs + 8 * WPB;                           // [MAXD] the incoming point
  double *red = xn + MAXD;                             // [8] scalars handed from wave 0 to the block
  double *stg = red + 8;                               // [2][WIN_STG] wave 0's inputs of the next panel step, staged by wave 1 (LDS-DMA)
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

  for (int t = p.t0; t < p.t0 + p.nt; ++t) {
    // ---- make room: move the window back to the origin when it reached the end of the buffer
    if (o + n >= CAP) {
      for (int c = 0; c < n; ++c)
        for (int r = c + tid; r < n; r += NTH) L[(size_t)c * CAP + r] = L[(size_t)(o + c) * CAP + o + r];
      for (int i = tid; i < n; i += NTH) {
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
    for (int i = tid; i < n2; i += NTH) {
      vv[i] = drop ? L[(size_t)o * CAP + o2 + i] : 0.0;
      kk[i] = win_cov(kid, d, pr, xw + o2 + i, CAP, xn, 1, false);
    }
    // the dropped sample's component of z: uniform, moved to scalar registers HERE so that its load is known to have landed
    // before the panel loop (left in a VGPR the compiler waits for it at its first use inside the loop with vmcnt(0), i.e.
    // for every store wave 0 has in flight, once per panel)
    const double z0v = drop ? z[o] : 0.0;
    const double z0 = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(z0v)), __builtin_amdgcn_readfirstlane(__double2loint(z0v)));
    double sl2 = 0, slz = 0, slog = 0, szz = 0;   // wave 0: per-lane partial sums over the panels (reduced once per tick)
    // sum of log(diag) without a log on the serial path: every lane of wave 0's first row keeps the running
    // product of its diagonal entries as (mantissa in [0.5, 1), exponent); ONE log per lane and tick at the end
    double pmant = 1.0;
    int pexp = 0;
    __syncthreads();

    // Panel pipeline.  Per 16-column panel the work is  A(p): wave 0 solves the 16x16 diagonal block for w and q
    // (serial, in registers) and forms the panel's c, s', l, z' and the new block  and  B(p): every row below takes
    // the panel's columns.  A(p+1) only needs B(p) on the 16 rows of the next diagonal block, so wave 0 applies B(p)
    // to those rows itself and goes straight on to A(p+1) while waves 1-3 sweep the remaining rows: one barrier per
    // panel, serial part and sweep side by side.  (c, s', w, q) of a panel live in cs[panel & 1], l in ll[p0 ..].
    const int i = lane & (WPB - 1);
    const int npan = (n2 + WPB - 1) / WPB;
    double ui = 0, ki = 0, zi = 0;  // wave 0: the diagonal block's rows of u, k, z (registers across panels)
    double Tb = 1.0, Skb = 0.0, Syb = 0.0;   // wave 0: t, sum w q, sum w qy up to the panel's first column (uniform)
    double cg = 1.0, csp = 0.0, cw = 0.0, cq = 0.0;   // wave 0: lane j keeps c_j, s'_j, w_j, q_j of the panel it solved last (its own sweep takes them by DPP)
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
    // the 16x16 diagonal block of panel p0 (lane = row): strictly lower part in a[] (zero elsewhere), the lane's own
    // diagonal entry in dg (1 on the padded rows); issued early by the caller
    auto load_diag = [&](int p0, int nb, double (&a)[WPB], double &dg) {
      const unsigned off = (unsigned)(((o2 + p0) * CAP + o2 + p0 + i) * (int)sizeof(double));
#pragma unroll
      for (int j = 0; j < WPB; ++j) a[j] = (j < i && i < nb) ? ld64(off, j * colb) : 0.0;
      dg = i < nb ? ld64(off + (unsigned)(i * colb), 0) : 1.0;
      zi = i < nb ? z[o2 + p0 + i] : 0.0;
    };
    auto phase_a = [&](int p0, int nb, double *csb, double (&a)[WPB], double dg) {
      // 1 / diagonal (off the chain: the block arrived a panel ago): hardware seed and two Newton steps
      double idg = __builtin_amdgcn_rcp(dg);
      idg = __builtin_fma(__builtin_fma(-dg, idg, 1.0), idg, idg);
      idg = __builtin_fma(__builtin_fma(-dg, idg, 1.0), idg, idg);
      // forward substitution of the block, both right-hand sides: after step J lane J's residual is final (its own
      // diagonal is not in a[]), so w_J = u_J / L_JJ is lane J's product, broadcast and subtracted in ONE DPP instruction
      static_for<0, WPB - 1>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        const double wv = ui * idg, qv = ki * idg;
        fnmac_bcast<J, true>(ui, wv, a[J]);
        fnmac_bcast<J, true>(ki, qv, a[J]);
      });
      const double wm = ui * idg, qm = ki * idg;          // w_i, q_i of this lane's column
      const double qy = __builtin_fma(wm, z0, zi);        // (L22^-1 y2)_i
      const double w2 = wm * wm;
      const double Tj = Tb + win_row_scan_excl(w2), Tj1 = Tj + w2;
      const double Sk = Skb + win_row_scan_excl(wm * qm), Sy = Syb + win_row_scan_excl(wm * qy);
      const double rj = rsqrt3(Tj), rj1 = rsqrt3(Tj1);
      const double g = Tj * rj * rj1;                      // c_j = sqrt(t_{j-1} / t_j)
      const double h = wm * (rj * rj);                     // w_j / t_{j-1}
      const double sp = h * g;                             // s'_j
      const double lj = __builtin_fma(-h, Sk, qm) * g;
      const double zn = __builtin_fma(-h, Sy, qy) * g;
      sl2 = __builtin_fma(lj, lj, sl2);
      slz = __builtin_fma(lj, zn, slz);
      szz = __builtin_fma(zn, zn, szz);
      const double dnew = dg * (Tj1 * rj1 * rj);           // L_jj / c_j
      if (lane < WPB) {
        csb[4 * i + 0] = g;
        csb[4 * i + 1] = sp;
        csb[4 * i + 2] = wm;
        csb[4 * i + 3] = qm;
      }
      if (lane < nb) {
        pmant *= dnew;
        pexp += __builtin_amdgcn_frexp_exp(pmant);
        pmant = __builtin_amdgcn_frexp_mant(pmant);
        ll[p0 + i] = lj;
        z[o2 + p0 + i] = zn;
      }
      // the new block, column by column from the right: u^(k)_i = L_ii w_i + sum_{m=k}^{i-1} L_im w_m (the substitution's
      // residual before column k, as a sum of what is left of it), L'_ik = c_k L_ik + s'_k u^(k)_i
      // (all lanes active for the DPP reads; the broadcast operands wm, g, sp were written long before: no hazard nop.  Columns right
      // of a lane's diagonal get s'_k u_i -- they land in the strict upper triangle of the slab, which nothing reads -- so the sixteen
      // stores need one predicate, the lane's, not one each)
      double acc = dg * wm;
      const unsigned off = (unsigned)(((o2 + p0) * CAP + o2 + p0 + i) * (int)sizeof(double));
      double nw[WPB - 1];
      static_for<0, WPB - 1>([&](auto kc) {
        constexpr int K = WPB - 2 - decltype(kc)::value;   // 14 .. 0
        fmac_bcast<K, K == WPB - 2>(acc, wm, a[K]);
        nw[K] = 0.0;
        fmac_bcast<K, K == WPB - 2>(nw[K], g, a[K]);
        fmac_bcast<K, K == WPB - 2>(nw[K], sp, acc);
      });
      if (lane < nb) {
#pragma unroll
        for (int K = 0; K < WPB - 1; ++K)
          if (K < nb) st64(nw[K], off, K * colb);   // (uniform: a column past the window may lie past the slab)
        st64(dnew, off + (unsigned)(i * colb), 0);
      }
      cg = g; csp = sp; cw = wm; cq = qm;
      // carry t and the running sums to the next panel (lane 15 of the row holds the block's totals)
      Tb = mov_bcast<WPB - 1>(Tj1);
      Skb = mov_bcast<WPB - 1>(__builtin_fma(wm, qm, Sk));
      Syb = mov_bcast<WPB - 1>(__builtin_fma(wm, qy, Sy));
    };
    // one row of the sweep: a[] = the row's 16 panel columns (old factor); u, k = the row's residuals
    auto sweep_row = [&](double (&a)[WPB], double &u, double &k, const double *csb) {
#pragma unroll
      for (int j = 0; j < WPB; ++j) {
        const double c = csb[4 * j], sp = csb[4 * j + 1], wj = csb[4 * j + 2], qj = csb[4 * j + 3], aj = a[j];
        a[j] = __builtin_fma(sp, u, c * aj);
        u = __builtin_fma(-aj, wj, u);
        k = __builtin_fma(-aj, qj, k);
      }
    };

    // Wave 0's inputs of panel step q -- the 16 rows under panel q's diagonal block in panel q's columns, the diagonal block
    // of panel q + 1 and its rows of z -- are this tick's OLD values until wave 0 itself rewrites them, so wave 1 requests them
    // one step ahead straight into LDS (dword LDS-DMA: the rows are only 8-byte aligned; 64 lanes = two columns of 16 rows);
    // rows / columns past the window are clamped (never used: wave 0 masks them).  Wave 0 then has no load on its path.
    auto stage_inputs = [&](int q) {
      typedef __attribute__((address_space(3))) void lds_void;
      typedef const __attribute__((address_space(1))) void gbl_void;
      const int p0q = q * WPB, r0 = p0q + WPB;
      double *sb = stg + (q & 1) * WIN_STG;
      const int rl = min(r0 + ((lane & 31) >> 1), n2 - 1), hf = lane & 1;
#pragma unroll
      for (int c2 = 0; c2 < WPB / 2; ++c2) {
        const int col = p0q + 2 * c2 + (lane >> 5);
        const char *g = reinterpret_cast<const char *>(L + (size_t)(o2 + col) * CAP + o2 + rl) + 4 * hf;
        __builtin_amdgcn_global_load_lds((gbl_void *)g, (lds_void *)(sb + c2 * 2 * WPB), 4, 0, 0);
      }
#pragma unroll
      for (int c2 = 0; c2 < WPB / 2; ++c2) {
        const int col = min(r0 + 2 * c2 + (lane >> 5), n2 - 1);
        const char *g = reinterpret_cast<const char *>(L + (size_t)(o2 + col) * CAP + o2 + rl) + 4 * hf;
        __builtin_amdgcn_global_load_lds((gbl_void *)g, (lds_void *)(sb + WPB * WPB + c2 * 2 * WPB), 4, 0, 0);
      }
      {
        const char *g = reinterpret_cast<const char *>(z + o2 + rl) + 4 * hf;   // lanes 32 .. 63: the same 16 rows again
        __builtin_amdgcn_global_load_lds((gbl_void *)g, (lds_void *)(sb + 2 * WPB * WPB), 4, 0, 0);
      }
    };
    // wave 0's own rows: the panel's c, s', w, q are still in its registers (lane j: column j) -- four DPP instructions per column, no LDS read
    auto sweep_row_w0 = [&](double (&a)[WPB], double &u, double &k) {
      static_for<0, WPB>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        double out = 0.0;
        fmac_bcast<J, J == 0>(out, cg, a[J]);
        fmac_bcast<J, J == 0>(out, csp, u);
        fnmac_bcast<J, J == 0>(u, cw, a[J]);
        fnmac_bcast<J, J == 0>(k, cq, a[J]);
        a[J] = out;
      });
    };
    if (wave == 1 && npan > 1) {
      stage_inputs(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (wave == 0 && npan > 0) {
      const int nb = min(WPB, n2);
      ui = i < nb ? vv[i] : 0.0;
      ki = i < nb ? kk[i] : 0.0;
      double ad[WPB], dg;
      load_diag(0, nb, ad, dg);
      phase_a(0, nb, cs, ad, dg);
    }
#if WIN_PROBE
    long long wp[4] = {0, 0, 0, 0}, wpt = __builtin_amdgcn_s_memtime();
    const long long wp0 = wpt;
#define WIN_LAP(k) { const long long nn_ = __builtin_amdgcn_s_memtime(); wp[k] += nn_ - wpt; wpt = nn_; }
#else
#define WIN_LAP(k)
#endif
    for (int pi = 0; pi < npan; ++pi) {
      WIN_LAP(3)
      lds_barrier();  // A(pi) and B(pi-1) are complete.  LDS only: within a tick no thread reads factor entries another
                      // thread wrote (a panel's entries have one owner), so the sweep's HBM stores stay in flight
      const int p0 = pi * WPB;
      const double *csb = cs + (pi & 1) * 4 * WPB;
      if (wave == 0) {
        WIN_LAP(0)
        if (pi + 1 < npan) {
          // B(pi) on the rows of the next diagonal block, then A(pi + 1) with u, k still in registers
          const int nb1 = min(WPB, n2 - (p0 + WPB));
          const int r = p0 + WPB + i;
          double ad[WPB], dg, a[WPB];
          const double *sb = stg + (pi & 1) * WIN_STG;
          const unsigned offr = (unsigned)(((o2 + p0) * CAP + o2 + r) * (int)sizeof(double));
          if (nb1 == WPB) {   // every panel but the last: no row of the block lies past the window
#pragma unroll
            for (int j = 0; j < WPB; ++j) a[j] = sb[j * WPB + i];
#pragma unroll
            for (int j = 0; j < WPB; ++j) ad[j] = j < i ? sb[WPB * WPB + j * WPB + i] : 0.0;
            dg = sb[WPB * WPB + i * WPB + i];
            zi = sb[2 * WPB * WPB + i];
            ui = vv[r];
            ki = kk[r];
          } else {
#pragma unroll
            for (int j = 0; j < WPB; ++j) a[j] = i < nb1 ? sb[j * WPB + i] : 0.0;
#pragma unroll
            for (int j = 0; j < WPB; ++j) ad[j] = (j < i && i < nb1) ? sb[WPB * WPB + j * WPB + i] : 0.0;
            dg = i < nb1 ? sb[WPB * WPB + i * WPB + i] : 1.0;
            zi = i < nb1 ? sb[2 * WPB * WPB + i] : 0.0;
            ui = i < nb1 ? vv[r] : 0.0;
            ki = i < nb1 ? kk[r] : 0.0;
          }
#if WIN_PROBE
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          WIN_LAP(1)
#endif
          sweep_row_w0(a, ui, ki);
          if (lane < nb1) {
#pragma unroll
            for (int j = 0; j < WPB; ++j) st64(a[j], offr, j * colb);
          }
          WIN_LAP(2)
          phase_a(p0 + WPB, nb1, cs + ((pi + 1) & 1) * 4 * WPB, ad, dg);
        }
      } else {
        if (wave == 1 && pi + 2 < npan) stage_inputs(pi + 1);
        // ---- B(pi): rows below the next diagonal block, the other waves
        for (int r = p0 + 2 * WPB + (tid - 64); r < n2; r += NTH - 64) {
          asm volatile("" ::: "memory");  // keep the panel's LDS scalars from being hoisted across rows
          const unsigned offr = (unsigned)(((o2 + p0) * CAP + o2 + r) * (int)sizeof(double));
          double a[WPB];
#pragma unroll
          for (int j = 0; j < WPB; ++j) a[j] = ld64(offr, j * colb);
          double u = vv[r], k = kk[r];
          sweep_row(a, u, k, csb);
#pragma unroll
          for (int j = 0; j < WPB; ++j) st64(a[j], offr, j * colb);
          vv[r] = u;
          kk[r] = k;
        }
        if (wave == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the staged inputs have landed before the barrier publishes them
      }
    }
    __syncthreads();

    // ---- append the new sample as the last row of the factor
    if (wave == 0) {
      double lg = log(pmant) + (double)pexp * 0.6931471805599453;  // lanes that never multiplied: log(1) + 0
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) lg += __shfl_xor(lg, off);
      slog = lg;
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {   // the 16 lanes of a row (the four rows hold identical copies)
        sl2 += __shfl_xor(sl2, off);
        slz += __shfl_xor(slz, off);
        szz += __shfl_xor(szz, off);
      }
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
    for (int j = tid; j < n2; j += NTH) Lrow[(size_t)j * CAP] = ll[j];
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
#if WIN_PROBE   // measurement build: the tick's outputs are wave 0's clock sums (barrier wait, inputs read, row sweep + stores, serial block)
      p.logml[oi] = (double)wp[0] * 67108864.0 + (double)wp[1];
      p.pred_mean[oi] = (double)wp[2] * 67108864.0 + (double)wp[3];
      p.pred_var[oi] = (double)(__builtin_amdgcn_s_memtime() - wp0);
#endif
    }
    o = o2;
    n = n2 + 1;
    __syncthreads();
  }
  if (tid == 0) {
    st[0] = o;
    st[1] = n;
    st[2] = bad;
    st[3] += p.nt;
    if (p.info_out) {
      __threadfence_system();   // the tick's outputs (this thread's stores, possibly to pinned host memory) are visible before the status word:
      p.info_out[w] = bad;      // a one-launch host push polls that word instead of synchronising the stream
    }
  }
}


// --------------------------------------------------------------------------------------------------
// k_window_pairs: TWO ticks per pass over the factor.  Under load the single-tick kernel is bound by its factor traffic
// (every entry read once and written once per tick, 4.1-4.2 TB/s whatever the chain does), so the
// steady-state ticks (full window, no ring compaction inside the pair) are taken two at a time: a panel's rows are loaded
// once, take tick t's rotations and substitution update, then tick t + 1's, and are stored once.
// With o1 = o + 1 the origin after the first drop, m = N - 1 and rows / columns counted from o1:
//   tick t    : v1 = column -1 (the dropped one), window [0, m), new row m = (l1, d1)
//   tick t + 1: v2 = column 0 AFTER tick t's update (row m contributes l1[0]), window [1, m], new row m + 1 = (l2, d2)
// Both ticks share the panel grid (16 columns from o1): per panel wave 0 runs A1 (tick t: rotations + substitution of the
// diagonal block) and then A2 (tick t + 1) on the block it still holds; in panel 0 lane 0 (row 0, dropped by tick t + 1) is
// inert for A2 and every row's v2 is its freshly rotated column-0 entry.  Tick t's new row m is born panel by panel (l1 of a
// panel is final after A1): until it reaches the diagonal block in the last panel it is swept like any other row -- by
// tick t + 1's rotations only -- with its entries taken from LDS instead of memory.  Same arithmetic as two single ticks
// (up to the order in which a row's two rotation sets are interleaved with other rows').
// --------------------------------------------------------------------------------------------------
// WPW windows per workgroup: with the traffic halved the pass is bound by wave 0's chains, which use ONE 16-lane row (the
// other three rows of the wave held copies); with WPW = 2 / 4 the rows serve different windows (row g -> window g WPW / 4 of
// the workgroup; DPP row_newbcast is row-local, nothing in the chain crosses rows), so one pass of the chains advances WPW
// windows, and waves 1-3 sweep the rows of all of them.  Windows of a context advance in lock-step: origin, size and panel
// count are workgroup-uniform.
template <int WPW>
__global__ __launch_bounds__(256, WIN_OCC) void k_window_pairs(WindowArgs p) {
  static_assert(WPW == 1 || WPW == 2 || WPW == 4, "rows of wave 0 per window: 4, 2 or 1");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int N = p.N, CAP = p.CAP, d = p.d, kid = p.kernel_id;
  const int NS = (N + 2 + 1) & ~1;                       // per-vector LDS stride (rows 0 .. m + 1)
  const int WS = 6 * NS + 16 * WPB + 2 * MAXD + 16;      // LDS doubles per window
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;
  const int wl0 = (g * WPW) >> 2;                        // wave 0: the window this lane's row serves
  const bool lead = (g & (4 / WPW - 1)) == 0;            // first row of its window (the others hold copies)
  double *lds = reinterpret_cast<double *>(smem_raw);
  // per-window LDS blocks: vv1 kk1 ll1 vv2 kk2 ll2 [NS each] | cs1 cs2 [2][WPB][4] (c, s', w, q per column) | xn [2][MAXD] | red [16]
  auto blk = [&](int wl) { return lds + wl * WS; };
  double *stg = lds + WPW * WS + (wave > 0 ? wave - 1 : 0) * (WPB * 64);   // sweep waves: the next trip's 64 rows x 16 columns, staged by LDS-DMA
  const int w0 = blockIdx.x * WPW;
  const size_t LWs = (size_t)CAP * CAP;
  double *Lg = p.L + (size_t)w0 * LWs;
  const int nth = (kid == K_SE_ISO) ? 3 : (kid == K_SE_ARD ? d + 2 : 4);
  // wave 0: this lane's window
  double *vv1 = blk(wl0), *kk1 = vv1 + NS, *ll1 = kk1 + NS, *vv2 = ll1 + NS, *kk2 = vv2 + NS, *ll2 = kk2 + NS;
  double *cs1 = ll2 + NS, *cs2 = cs1 + 8 * WPB, *xn = cs2 + 8 * WPB, *red = xn + 2 * MAXD;
  double *z = p.z + (size_t)(w0 + wl0) * CAP;
  const double *pr = p.prep + (size_t)(w0 + wl0) * PREP_N;
  const double noise = p.theta[(size_t)(w0 + wl0) * MAX_THETA + nth - 1];
  const unsigned woff = (unsigned)(wl0 * (int)(LWs * sizeof(double)));
  int o = p.state[w0 * 4], bad = tid < WPW ? p.state[(w0 + tid) * 4 + 2] : 0;
  // The host cuts a push into launches from its MIRROR of the windows' origin and size; this kernel is only correct for full
  // windows with room for two more rows.  A window whose device state says otherwise (the mirror went out of step: a failed
  // launch, an earlier push that returned early) is flagged (state[2] = -1) and left untouched instead of being swept.
  if (p.state[w0 * 4 + 1] != N || o + N + 1 >= CAP) {
    if (tid < WPW && p.state[(w0 + tid) * 4 + 2] == 0) p.state[(w0 + tid) * 4 + 2] = -1;
    if (tid < WPW && p.info_out) p.info_out[w0 + tid] = p.state[(w0 + tid) * 4 + 2];
    return;
  }
  const int m = N - 1;
  const int npan = (m + 1 + WPB - 1) / WPB;   // panels of tick t + 1 (columns 1 .. m); tick t uses columns 0 .. m - 1
  const int i = lane & (WPB - 1);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(Lg, 0, (int)(WPW * LWs * sizeof(double)), 0x00020000);
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
  const int colb = CAP * (int)sizeof(double);

  for (int t = p.t0; t < p.t0 + p.nt; t += 2) {
    const int o1 = o + 1;
    if (tid < 2 * d * WPW) {
      const int wl = tid / (2 * d), r2 = tid - wl * 2 * d, tk = r2 / d, q = r2 - tk * d;
      (blk(wl) + 6 * NS + 16 * WPB)[tk * MAXD + q] = p.xs[((size_t)(w0 + wl) * p.T + t + tk) * d + q];
    }
    __syncthreads();
    for (int idx = tid; idx < WPW * (m + 1); idx += 256) {
      const int wl = idx / (m + 1), rr = idx - wl * (m + 1);
      double *b = blk(wl);
      const double *xnw = b + 6 * NS + 16 * WPB, *prw = p.prep + (size_t)(w0 + wl) * PREP_N;
      const double *xww = p.xw + (size_t)(w0 + wl) * d * CAP;
      if (rr < m) {
        b[rr] = Lg[wl * LWs + (size_t)o * CAP + o1 + rr];                                      // vv1
        b[NS + rr] = win_cov(kid, d, prw, xww + o1 + rr, CAP, xnw, 1, false);                  // kk1
        b[4 * NS + rr] = win_cov(kid, d, prw, xww + o1 + rr, CAP, xnw + MAXD, 1, false);       // kk2
      } else {
        b[4 * NS + m] = win_cov(kid, d, prw, xnw, 1, xnw + MAXD, 1, false);   // kk2[m]: the two incoming points
        b[m] = 0.0;            // vv1[m]
        b[NS + m] = 0.0;       // kk1[m]
        b[2 * NS + m] = 0.0;   // ll1[m]
      }
    }
    double vz1 = z[o], vz2 = 0.0;   // the dropped samples' components of z
    double sl2a = 0, slza = 0, szza = 0, sl2b = 0, slzb = 0, szzb = 0;
    double pma = 1.0, pmb = 1.0;    // running products of the diagonals (mantissa, exponent), one per tick
    int pea = 0, peb = 0;
    double d1 = 1.0, znew1 = 0.0;
    int bad1 = 0;   // tick t's pivot check, seen by wave 0 (handed to the window's tail thread through LDS)
    __syncthreads();

    // element (row rr, column cc) of the factor, both counted from o1
    auto eoff = [&](int rr, int cc) { return (unsigned)(((o1 + cc) * CAP + o1 + rr) * (int)sizeof(double)); };   // + the window's slab
    // One tick's work on a diagonal block (lane = row), the substitution form of k_window_ticks: a[] = the block's strictly
    // lower part (zero elsewhere), dg = the lane's diagonal entry (1 on padded rows), u / k = the rows' residuals of the rank-1
    // vector and of the incoming point, zi = the rows' z, z0 = the dropped sample's z.  Leaves the new block in a[] / dg, the new
    // z in zi, the lane's column values (c, s', w, q) in co[], its l in lj, and carries t / the running sums on.
    auto solve_block = [&](double (&a)[WPB], double &dg, double &u, double &k, double &zi, const double z0, double (&car)[3],
                           double &sl2, double &slz, double &szz, double (&co)[4], double &lj) {
      double idg = __builtin_amdgcn_rcp(dg);
      idg = __builtin_fma(__builtin_fma(-dg, idg, 1.0), idg, idg);
      idg = __builtin_fma(__builtin_fma(-dg, idg, 1.0), idg, idg);
      static_for<0, WPB - 1>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        const double wv = u * idg, qv = k * idg;
        fnmac_bcast<J, true>(u, wv, a[J]);
        fnmac_bcast<J, true>(k, qv, a[J]);
      });
      const double wm = u * idg, qm = k * idg;
      const double qy = __builtin_fma(wm, z0, zi);
      const double w2 = wm * wm;
      const double Tj = car[0] + win_row_scan_excl(w2), Tj1 = Tj + w2;
      const double Sk = car[1] + win_row_scan_excl(wm * qm), Sy = car[2] + win_row_scan_excl(wm * qy);
      const double rj = rsqrt3(Tj), rj1 = rsqrt3(Tj1);
      const double g = Tj * rj * rj1;
      const double h = wm * (rj * rj);
      const double sp = h * g;
      lj = __builtin_fma(-h, Sk, qm) * g;
      const double zn = __builtin_fma(-h, Sy, qy) * g;
      sl2 = __builtin_fma(lj, lj, sl2);
      slz = __builtin_fma(lj, zn, slz);
      szz = __builtin_fma(zn, zn, szz);
      double acc = dg * wm;
      dg = dg * (Tj1 * rj1 * rj);
      zi = zn;
      static_for<0, WPB - 1>([&](auto kc) {
        constexpr int K = WPB - 2 - decltype(kc)::value;   // 14 .. 0
        fmac_bcast<K, K == WPB - 2>(acc, wm, a[K]);
        double nw = 0.0;
        fmac_bcast<K, K == WPB - 2>(nw, g, a[K]);
        fmac_bcast<K, K == WPB - 2>(nw, sp, acc);
        a[K] = (K < i) ? nw : 0.0;
      });
      co[0] = g; co[1] = sp; co[2] = wm; co[3] = qm;
      car[0] = mov_bcast<WPB - 1>(Tj1);
      car[1] = mov_bcast<WPB - 1>(__builtin_fma(wm, qm, Sk));
      car[2] = mov_bcast<WPB - 1>(__builtin_fma(wm, qy, Sy));
    };
    // one row of the sweep with one tick's columns (c, s', w, q from LDS)
    auto sweep_row = [&](double (&a)[WPB], double &u, double &k, const double *csb) {
#pragma unroll
      for (int j = 0; j < WPB; ++j) {
        const double c = csb[4 * j], sp = csb[4 * j + 1], wj = csb[4 * j + 2], qj = csb[4 * j + 3], aj = a[j];
        a[j] = __builtin_fma(sp, u, c * aj);
        u = __builtin_fma(-aj, wj, u);
        k = __builtin_fma(-aj, qj, k);
      }
    };
    // wave 0's own rows: the columns' values are in its registers (lane j of the row: column j), four DPP instructions per column
    auto sweep_row_w0 = [&](double (&a)[WPB], double &u, double &k, const double (&co)[4]) {
      static_for<0, WPB>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        double out = 0.0;
        fmac_bcast<J, J == 0>(out, co[0], a[J]);
        fmac_bcast<J, J == 0>(out, co[1], u);
        fnmac_bcast<J, J == 0>(u, co[2], a[J]);
        fnmac_bcast<J, J == 0>(k, co[3], a[J]);
        a[J] = out;
      });
    };
    // a row below the diagonal block of panel pi through both ticks: tick t (rows < m), then tick t + 1
    auto both_ticks = [&](double (&a)[WPB], int rr, int pi, double &v1, double &k1, double &v2, double &k2, const double *b) {
      const int cso = (pi & 1) * 4 * WPB;
      if (rr < m) sweep_row(a, v1, k1, b + 6 * NS + cso);                 // cs1
      if (pi == 0) v2 = a[0];   // column 0 after tick t: this row's entry of tick t + 1's rank-1 vector
      sweep_row(a, v2, k2, b + 6 * NS + 8 * WPB + cso);                   // cs2
    };
    // the 16 panel entries of row rr: memory, or -- tick t's new row m, which exists only as l1 so far -- LDS
    auto load_row = [&](double (&a)[WPB], int rr, int p0, bool live, unsigned wo, const double *b) {
      const unsigned off = wo + eoff(live ? rr : 0, p0);
#pragma unroll
      for (int j = 0; j < WPB; ++j) {
        const double x = ld64(off, j * colb);
        a[j] = !live ? 0.0 : (rr == m ? b[2 * NS + p0 + j] : x);   // ll1
      }
    };

    // ---- wave 0 state: the diagonal block's rows of u, k, z for both ticks (registers across panels)
    double v1i = 0, k1i = 0, v2i = 0, k2i = 0, zi = 0;
    double car1[3] = {1.0, 0.0, 0.0}, car2[3] = {1.0, 0.0, 0.0};   // t, sum w q, sum w qy up to the panel, per tick
    double co1[4] = {1.0, 0.0, 0.0, 0.0}, co2[4] = {1.0, 0.0, 0.0, 0.0};   // lane j: c, s', w, q of column j of the panel solved last
    auto row_sum = [&](double x) {   // over the 16 lanes of a row
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) x += __shfl_xor(x, off);
      return x;
    };
    auto block_ab = [&](int pi, double (&a)[WPB], double dg) {   // A1 then A2 on the diagonal block of panel pi (a[], dg, zi, v1i .. k2i loaded)
      const int p0 = pi * WPB;
      const int nb1 = max(0, min(WPB, m - p0));        // rows / columns of tick t in this block
      const int nb2 = min(WPB, m + 1 - p0);            // of tick t + 1 (the block's last row may be row m)
      double *c1 = cs1 + (pi & 1) * 4 * WPB, *c2 = cs2 + (pi & 1) * 4 * WPB;
      // ---- A1
      {
        double lj;
        solve_block(a, dg, v1i, k1i, zi, vz1, car1, sl2a, slza, szza, co1, lj);
        if (lead) {
          c1[4 * i + 0] = co1[0]; c1[4 * i + 1] = co1[1]; c1[4 * i + 2] = co1[2]; c1[4 * i + 3] = co1[3];
          if (i < nb1) {
            pma *= dg;
            pea += __builtin_amdgcn_frexp_exp(pma);
            pma = __builtin_amdgcn_frexp_mant(pma);
            ll1[p0 + i] = lj;
          }
        }
      }
      if (p0 + WPB > m) {
        // the last panel: tick t's new row m = (l1, d1) joins the block as row m - p0, with its z
        const double kss1 = (kid == K_RBF_BROWNIAN) ? pr[9] * pr[10] * fabs(xn[0]) : pr[9];
        const double s2 = row_sum(sl2a), sz = row_sum(slza);
        double dd = kss1 + noise + 1e-8 - s2;
        if (!(dd > 0.0)) {
          bad1 = t + 1;
          dd = 1e-300;
        }
        d1 = sqrt(dd);
        znew1 = (p.ys[(size_t)(w0 + wl0) * p.T + t] - sz) / d1;
        const int im = m - p0;
        if (i == im) {
#pragma unroll
          for (int j = 0; j < WPB; ++j) a[j] = j < im ? ll1[p0 + j] : 0.0;
          dg = d1;
          zi = znew1;
        }
      }
      // ---- A2
      {
        const bool first = pi == 0;
        if (first) {
          vz2 = mov_bcast<0>(zi);           // row 0's z after tick t: the component tick t + 1 drops
          v2i = i == 0 ? 0.0 : a[0];        // column 0 after tick t; row 0 itself is inert from here on
          zi = i == 0 ? 0.0 : zi;
          k2i = i == 0 ? 0.0 : k2i;
          dg = i == 0 ? 1.0 : dg;
          a[0] = 0.0;                       // column 0 is not part of tick t + 1
        }
        double lj;
        solve_block(a, dg, v2i, k2i, zi, vz2, car2, sl2b, slzb, szzb, co2, lj);
        if (lead) {
          c2[4 * i + 0] = co2[0]; c2[4 * i + 1] = co2[1]; c2[4 * i + 2] = co2[2]; c2[4 * i + 3] = co2[3];
          if (i < nb2) {
            if (!(first && i == 0)) {
              pmb *= dg;
              peb += __builtin_amdgcn_frexp_exp(pmb);
              pmb = __builtin_amdgcn_frexp_mant(pmb);
            }
            ll2[p0 + i] = (first && i == 0) ? 0.0 : lj;
          }
        }
      }
      if (lead && i < nb2) {
        // columns right of the diagonal hold zeros: they land in the strict upper triangle of the slab, which nothing reads
        const unsigned off = woff + eoff(p0 + i, p0);
#pragma unroll
        for (int j = 0; j < WPB - 1; ++j)
          if (j < nb2) st64(a[j], off, j * colb);
        st64(dg, off + (unsigned)(i * colb), 0);
        z[o1 + p0 + i] = zi;
      }
    };

    // sweep waves: trips of a panel (wave-uniform), the (window, row) of a trip's slot, and the request of a trip's entries
    auto trips_of = [&](int pi) {
      const int tot = WPW * (m + 1 - (pi * WPB + 2 * WPB)) - (wave - 1) * 64;
      return tot > 0 ? (tot + 191) / 192 : 0;
    };
    auto slot_row = [&](int pi, int n, int r, int &wl, int &rr, bool &ok) {
      const int R = m + 1 - (pi * WPB + 2 * WPB);
      const int idx = (wave - 1) * 64 + r + 192 * n;
      ok = idx < WPW * R;
      wl = (WPW == 1 || !ok) ? 0 : idx / R;
      rr = pi * WPB + 2 * WPB + (ok ? idx - wl * R : 0);   // a slot past the rows: any row of the slab (never used)
    };
    auto issue_stage = [&](int pi, int n) {
      typedef __attribute__((address_space(3))) void lds_void;
#pragma unroll
      for (int h = 0; h < 2; ++h) {   // lane l: dword l & 1 of row slot 32 h + (l >> 1)
        int wl, rr;
        bool ok;
        slot_row(pi, n, 32 * h + (lane >> 1), wl, rr, ok);
        const unsigned vo = (unsigned)(wl * (int)(LWs * sizeof(double))) + eoff(rr, pi * WPB) + 4u * (lane & 1);
#pragma unroll
        for (int c = 0; c < WPB; ++c)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(stg + c * 64 + 32 * h), 4, vo, c * colb, 0, 0);
      }
    };
    bool first_trip = true;
    if (wave > 0) {
      int pi0 = 0;
      while (pi0 < npan && trips_of(pi0) == 0) ++pi0;
      if (pi0 < npan) issue_stage(pi0, 0);
    }
    if (wave == 0) {
      double ad[WPB], dg;
      {
        const int nb = min(WPB, m + 1);
        const unsigned off = woff + eoff(i, 0);
        const bool in = i < nb && i < m;
#pragma unroll
        for (int j = 0; j < WPB; ++j) ad[j] = (j < i && in) ? ld64(off, j * colb) : 0.0;
        dg = in ? ld64(off + (unsigned)(i * colb), 0) : 1.0;
        zi = in ? z[o1 + i] : 0.0;
        v1i = i < nb ? vv1[i] : 0.0;
        k1i = i < nb ? kk1[i] : 0.0;
        k2i = i < nb ? kk2[i] : 0.0;
      }
      block_ab(0, ad, dg);
    }
    for (int pi = 0; pi < npan; ++pi) {
      lds_barrier();  // A1 / A2 of panel pi and the sweep of panel pi - 1 are complete
      const int p0 = pi * WPB;
      if (wave == 0) {
        if (pi + 1 < npan) {
          // the rows of the next diagonal block through panel pi, then A1 / A2 of panel pi + 1
          const int nbn = min(WPB, m + 1 - (p0 + WPB));   // rows of the next block (row m may be its last)
          const int rr = p0 + WPB + i;
          const bool live = i < nbn, old = live && rr < m;   // old: a row of the factor (not tick t's new row)
          double ad[WPB], dg;
          {
            const unsigned off = woff + eoff(old ? rr : 0, p0 + WPB);
#pragma unroll
            for (int j = 0; j < WPB; ++j) {
              const double x = ld64(off, j * colb);
              ad[j] = (j < i && old) ? x : 0.0;
            }
            const double xd = ld64(woff + eoff(old ? rr : 0, old ? rr : 0), 0);
            dg = old ? xd : 1.0;
          }
          const double zn = old ? z[o1 + rr] : 0.0;
          double a[WPB];
          {
            const unsigned off = woff + eoff(old ? rr : 0, p0);
#pragma unroll
            for (int j = 0; j < WPB; ++j) {
              const double x = ld64(off, j * colb);
              a[j] = old ? x : 0.0;
            }
          }
          v1i = old ? vv1[rr] : 0.0;
          k1i = old ? kk1[rr] : 0.0;
          v2i = (live && pi > 0) ? vv2[rr] : 0.0;
          k2i = live ? kk2[rr] : 0.0;
          // every lane takes both sweeps (the DPP reads need the whole row active): rows past the factor carry zeros through tick
          // t; tick t's new row m, which exists only as l1 so far, gets its entries from LDS between the two
          sweep_row_w0(a, v1i, k1i, co1);
          if (p0 + 2 * WPB > m) {
#pragma unroll
            for (int j = 0; j < WPB; ++j) a[j] = (live && rr == m) ? ll1[p0 + j] : a[j];
          }
          if (pi == 0) v2i = live ? a[0] : 0.0;
          sweep_row_w0(a, v2i, k2i, co2);
          if (lead && i < nbn) {
            const unsigned off = woff + eoff(rr, p0);
#pragma unroll
            for (int j = 0; j < WPB; ++j) st64(a[j], off, j * colb);
          }
          zi = zn;
          block_ab(pi + 1, ad, dg);
        }
      } else {
        // ---- the rows below the next diagonal block, three waves.  A trip = 64 rows x the panel's 16 columns per wave.  Its entries
        // were requested ONE TRIP AGO, in front of that trip's stores, straight into LDS (dword LDS-DMA through the buffer descriptor:
        // the rows are only 8-byte aligned): vmcnt counts loads and stores in one sequence, so a load issued after a trip's stores
        // could only be waited for together with them -- the store round trip on every trip's path (measured: 3.2 M ticks/s without
        // the arithmetic, 4.8 M without the stores).  Requested first, the wait leaves exactly the sixteen stores outstanding.
        const int nt = trips_of(pi);
        for (int n = 0; n < nt; ++n) {
          if (first_trip) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
          first_trip = false;
          int wl, rr;
          bool ok;
          slot_row(pi, n, lane, wl, rr, ok);
          double *b = blk(wl);
          const unsigned wo = (unsigned)(wl * (int)(LWs * sizeof(double)));
          double a[WPB];
#pragma unroll
          for (int j = 0; j < WPB; ++j) a[j] = stg[j * 64 + lane];
          if (rr == m) {   // tick t's new row exists only as l1 so far
#pragma unroll
            for (int j = 0; j < WPB; ++j) a[j] = b[2 * NS + p0 + j];
          }
          double v1 = b[rr], k1 = b[NS + rr], v2 = pi > 0 ? b[3 * NS + rr] : 0.0, k2 = b[4 * NS + rr];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the stage has been read: the next trip may land in it
          {
            int npi = pi, nn = n + 1;
            if (nn >= nt) {
              nn = 0;
              do ++npi; while (npi < npan && trips_of(npi) == 0);
            }
            if (npi < npan) issue_stage(npi, nn);
          }
          if (ok) {
            both_ticks(a, rr, pi, v1, k1, v2, k2, b);
            const unsigned off = wo + eoff(rr, p0);
#pragma unroll
            for (int j = 0; j < WPB; ++j) st64(a[j], off, j * colb);
            b[rr] = v1;
            b[NS + rr] = k1;
            b[3 * NS + rr] = v2;
            b[4 * NS + rr] = k2;
          }
        }
      }
    }
    __syncthreads();

    // ---- outputs of both ticks; tick t + 1's new row m + 1
    if (wave == 0) {
      double la = log(pma) + (double)pea * 0.6931471805599453, lb = log(pmb) + (double)peb * 0.6931471805599453;
#pragma unroll
      for (int off = 32 / WPW; off > 0; off >>= 1) {   // within the window's rows
        la += __shfl_xor(la, off);
        lb += __shfl_xor(lb, off);
      }
      sl2a = row_sum(sl2a); slza = row_sum(slza); szza = row_sum(szza);   // per-lane partial sums over the panels
      sl2b = row_sum(sl2b); slzb = row_sum(slzb); szzb = row_sum(szzb);
      if (lead && i == 0) {
        red[0] = sl2a; red[1] = slza; red[2] = la; red[3] = szza; red[4] = d1; red[5] = znew1; red[6] = (double)bad1;
        red[8] = sl2b; red[9] = slzb; red[10] = lb; red[11] = szzb;
      }
    }
    __syncthreads();
    for (int idx = tid; idx < WPW * m; idx += 256) {
      const int wl = idx / m, cc = 1 + (idx - wl * m);
      Lg[wl * LWs + (size_t)(o1 + cc) * CAP + o1 + m + 1] = (blk(wl) + 5 * NS)[cc];   // ll2
    }
    if (tid < WPW) {
      const int wl = tid, w = w0 + wl;
      double *b = blk(wl);
      const double *xn = b + 6 * NS + 16 * WPB, *red = xn + 2 * MAXD;
      const double *pr = p.prep + (size_t)w * PREP_N;
      const double noise = p.theta[(size_t)w * MAX_THETA + nth - 1];
      double *L = Lg + wl * LWs, *z = p.z + (size_t)w * CAP, *yw = p.yw + (size_t)w * CAP, *xw = p.xw + (size_t)w * d * CAP;
      if (bad == 0 && red[6] != 0.0) bad = (int)red[6];
      const double kss1 = (kid == K_RBF_BROWNIAN) ? pr[9] * pr[10] * fabs(xn[0]) : pr[9];
      const double kss2 = (kid == K_RBF_BROWNIAN) ? pr[9] * pr[10] * fabs(xn[MAXD]) : pr[9];
      const double y1 = p.ys[(size_t)w * p.T + t], y2 = p.ys[(size_t)w * p.T + t + 1];
      const size_t oi = (size_t)w * p.T + t;
      double pv1 = kss1 - red[0];
      pv1 = pv1 < 1e-15 ? 1e-15 : pv1;
      p.pred_mean[oi] = red[1];
      p.pred_var[oi] = p.include_noise ? pv1 + noise : pv1;
      p.logml[oi] = -0.5 * (red[3] + red[5] * red[5]) - (red[2] + log(red[4])) - 0.5 * (double)N * 1.8378770664093453;
      double dd = kss2 + noise + 1e-8 - red[8];
      if (!(dd > 0.0)) {
        if (bad == 0) bad = t + 2;
        dd = 1e-300;
      }
      const double d2 = sqrt(dd), znew2 = (y2 - red[9]) / d2;
      double pv2 = kss2 - red[8];
      pv2 = pv2 < 1e-15 ? 1e-15 : pv2;
      p.pred_mean[oi + 1] = red[9];
      p.pred_var[oi + 1] = p.include_noise ? pv2 + noise : pv2;
      p.logml[oi + 1] = -0.5 * (red[11] + znew2 * znew2) - (red[10] + log(d2)) - 0.5 * (double)N * 1.8378770664093453;
      L[(size_t)(o1 + m + 1) * CAP + o1 + m + 1] = d2;
      z[o1 + m + 1] = znew2;
      yw[o1 + m] = y1;
      yw[o1 + m + 1] = y2;
      for (int q = 0; q < d; ++q) {
        xw[q * CAP + o1 + m] = xn[q];
        xw[q * CAP + o1 + m + 1] = xn[MAXD + q];
      }
    }
    o += 2;   // (the status `bad` is per thread; thread 0 sits in wave 0, so its copy carries both ticks' checks)
    __syncthreads();
  }
  if (tid < WPW) {
    int *st = p.state + (w0 + tid) * 4;
    st[0] = o;
    st[1] = N;
    st[2] = bad;
    st[3] += p.nt;
    if (p.info_out) p.info_out[w0 + tid] = bad;
  }
}

// --------------------------------------------------------------------------------------------------
// k_window_multi<TK>: TK ticks per pass over the factor (TK = 4 ships), the paired kernel's scheme carried on.  With the
// substitution form a tick's serial part is short (solve_block), the pass is bound by the factor's bytes, and TK ticks read
// and write them once.  Rows / columns counted from o1 = o + 1, m = N - 1, tick q = 0 .. TK - 1:
//   drops row q - 1; its rank-1 vector is column q - 1 AFTER ticks < q (tick 0: column -1 from memory); its window is rows and
//   columns [q, m + q); its new row is m + q = (l_q, d_q).
// Rows m .. m + TK - 2 are born inside the pass: row m + q exists only as l_q (LDS) until tick q + 1 takes it, is swept by ticks
// > q only, and is stored like any other row; row m + TK - 1 is written at the end.  In panel 0 lane q - 1 of the diagonal block
// goes inert before tick q and column q - 1 leaves the block.  One window per workgroup (four lane rows of wave 0 hold copies).
// --------------------------------------------------------------------------------------------------
template <int TK>
__global__ __launch_bounds__(256, WIN_OCC) void k_window_multi(WindowArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int N = p.N, CAP = p.CAP, d = p.d, kid = p.kernel_id;
  const int NS = (N + TK + 3) & ~1;                       // per-vector LDS stride (rows 0 .. m + TK - 1)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool lead = (lane >> 4) == 0;
  const int i = lane & (WPB - 1);
  double *lds = reinterpret_cast<double *>(smem_raw);
  // LDS: vv[TK] kk[TK] ll[TK] (NS each) | cs[TK][2][WPB][4] | xn[TK][MAXD] | red[TK][8] | the sweep waves' stages 3 x [16][64]
  double *VV = lds, *KK = VV + TK * NS, *LL = KK + TK * NS, *CS = LL + TK * NS, *XN = CS + TK * 8 * WPB, *RED = XN + TK * MAXD;
  double *stg = RED + TK * 8 + (wave > 0 ? wave - 1 : 0) * (WPB * 64);
  const int w = blockIdx.x;
  double *Lg = p.L + (size_t)w * CAP * CAP;
  double *z = p.z + (size_t)w * CAP, *xw = p.xw + (size_t)w * d * CAP, *yw = p.yw + (size_t)w * CAP;
  const double *pr = p.prep + (size_t)w * PREP_N;
  const int nth = (kid == K_SE_ISO) ? 3 : (kid == K_SE_ARD ? d + 2 : 4);
  const double noise = p.theta[(size_t)w * MAX_THETA + nth - 1];
  int o = p.state[w * 4], bad = p.state[w * 4 + 2];
  // only correct for full windows with room for TK more rows (the host cuts a push from its mirror; see k_window_pairs)
  if (p.state[w * 4 + 1] != N || o + N + TK - 1 >= CAP) {
    if (tid == 0 && p.state[w * 4 + 2] == 0) p.state[w * 4 + 2] = -1;
    if (tid == 0 && p.info_out) p.info_out[w] = p.state[w * 4 + 2];
    return;
  }
  const int m = N - 1;
  const int nrow = m + TK - 1;                 // rows 0 .. nrow - 1 pass through the sweeps (old rows and the rows born by ticks < TK - 1)
  const int npan = (nrow + WPB - 1) / WPB;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(Lg, 0, (int)((size_t)CAP * CAP * sizeof(double)), 0x00020000);
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
  const int colb = CAP * (int)sizeof(double);

  for (int t = p.t0; t < p.t0 + p.nt; t += TK) {
    const int o1 = o + 1;
    if (tid < TK * d) XN[(tid / d) * MAXD + tid % d] = p.xs[((size_t)w * p.T + t + tid / d) * d + tid % d];
    __syncthreads();
    for (int rr = tid; rr < nrow; rr += 256) {
#pragma unroll
      for (int q = 0; q < TK; ++q) {
        double kv = 0.0;
        if (rr < m) kv = win_cov(kid, d, pr, xw + o1 + rr, CAP, XN + q * MAXD, 1, false);
        else if (rr - m < q) kv = win_cov(kid, d, pr, XN + (rr - m) * MAXD, 1, XN + q * MAXD, 1, false);   // two incoming points
        KK[q * NS + rr] = kv;
        VV[q * NS + rr] = (q == 0 && rr < m) ? Lg[(size_t)o * CAP + o1 + rr] : 0.0;
        LL[q * NS + rr] = 0.0;
      }
    }
    double z0q[TK], uq[TK], kq[TK], car[TK][3], co[TK][4], sl2[TK], slz[TK], szz[TK], pm[TK], dq[TK], znq[TK];
    int pe[TK], badq = 0;
#pragma unroll
    for (int q = 0; q < TK; ++q) {
      z0q[q] = 0.0; uq[q] = 0.0; kq[q] = 0.0; car[q][0] = 1.0; car[q][1] = 0.0; car[q][2] = 0.0;
      co[q][0] = 1.0; co[q][1] = 0.0; co[q][2] = 0.0; co[q][3] = 0.0;
      sl2[q] = 0.0; slz[q] = 0.0; szz[q] = 0.0; pm[q] = 1.0; pe[q] = 0; dq[q] = 1.0; znq[q] = 0.0;
    }
    z0q[0] = z[o];
    __syncthreads();

    auto eoff = [&](int rr, int cc) { return (unsigned)(((o1 + cc) * CAP + o1 + rr) * (int)sizeof(double)); };
    auto solve_block = [&](double (&a)[WPB], double &dg, double &u, double &k, double &zi, const double z0, double (&cr)[3],
                           double &s2, double &sz, double &szq, double (&cq)[4], double &lj) {
      double idg = __builtin_amdgcn_rcp(dg);
      idg = __builtin_fma(__builtin_fma(-dg, idg, 1.0), idg, idg);
      idg = __builtin_fma(__builtin_fma(-dg, idg, 1.0), idg, idg);
      static_for<0, WPB - 1>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        const double wv = u * idg, qv = k * idg;
        fnmac_bcast<J, true>(u, wv, a[J]);
        fnmac_bcast<J, true>(k, qv, a[J]);
      });
      const double wm = u * idg, qm = k * idg;
      const double qy = __builtin_fma(wm, z0, zi);
      const double w2 = wm * wm;
      const double Tj = cr[0] + win_row_scan_excl(w2), Tj1 = Tj + w2;
      const double Sk = cr[1] + win_row_scan_excl(wm * qm), Sy = cr[2] + win_row_scan_excl(wm * qy);
      const double rj = rsqrt3(Tj), rj1 = rsqrt3(Tj1);
      const double g = Tj * rj * rj1;
      const double h = wm * (rj * rj);
      const double sp = h * g;
      lj = __builtin_fma(-h, Sk, qm) * g;
      const double zn = __builtin_fma(-h, Sy, qy) * g;
      s2 = __builtin_fma(lj, lj, s2);
      sz = __builtin_fma(lj, zn, sz);
      szq = __builtin_fma(zn, zn, szq);
      double acc = dg * wm;
      dg = dg * (Tj1 * rj1 * rj);
      zi = zn;
      static_for<0, WPB - 1>([&](auto kc) {
        constexpr int K = WPB - 2 - decltype(kc)::value;   // 14 .. 0
        fmac_bcast<K, K == WPB - 2>(acc, wm, a[K]);
        double nw = 0.0;
        fmac_bcast<K, K == WPB - 2>(nw, g, a[K]);
        fmac_bcast<K, K == WPB - 2>(nw, sp, acc);
        a[K] = (K < i) ? nw : 0.0;
      });
      cq[0] = g; cq[1] = sp; cq[2] = wm; cq[3] = qm;
      cr[0] = mov_bcast<WPB - 1>(Tj1);
      cr[1] = mov_bcast<WPB - 1>(__builtin_fma(wm, qm, Sk));
      cr[2] = mov_bcast<WPB - 1>(__builtin_fma(wm, qy, Sy));
    };
    auto sweep_row = [&](double (&a)[WPB], double &u, double &k, const double *csb) {
#pragma unroll
      for (int j = 0; j < WPB; ++j) {
        const double c = csb[4 * j], sp = csb[4 * j + 1], wj = csb[4 * j + 2], qj = csb[4 * j + 3], aj = a[j];
        a[j] = __builtin_fma(sp, u, c * aj);
        u = __builtin_fma(-aj, wj, u);
        k = __builtin_fma(-aj, qj, k);
      }
    };
    auto sweep_row_w0 = [&](double (&a)[WPB], double &u, double &k, const double (&cq)[4]) {
      static_for<0, WPB>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        double out = 0.0;
        fmac_bcast<J, J == 0>(out, cq[0], a[J]);
        fmac_bcast<J, J == 0>(out, cq[1], u);
        fnmac_bcast<J, J == 0>(u, cq[2], a[J]);
        fnmac_bcast<J, J == 0>(k, cq[3], a[J]);
        a[J] = out;
      });
    };
    auto row_sum = [&](double x) {
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) x += __shfl_xor(x, off);
      return x;
    };
    double zi = 0.0;
    // A_0 .. A_{TK-1} on the diagonal block of panel pi (a[] strictly lower, dg, zi, uq[], kq[] loaded)
    auto block_all = [&](int pi, double (&a)[WPB], double dg) {
      const int p0 = pi * WPB, col = p0 + i;
      static_for<0, TK>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        if constexpr (Q >= 1) {
          if (pi == 0) {   // row Q - 1 leaves: its z is what tick Q drops, column Q - 1 (after ticks < Q) is tick Q's rank-1 vector
            z0q[Q] = mov_bcast<Q - 1>(zi);
            uq[Q] = i >= Q ? a[Q - 1] : 0.0;
            kq[Q] = i >= Q ? kq[Q] : 0.0;
            zi = i == Q - 1 ? 0.0 : zi;
            dg = i == Q - 1 ? 1.0 : dg;
            a[Q - 1] = 0.0;
          }
        }
        double lj;
        solve_block(a, dg, uq[Q], kq[Q], zi, z0q[Q], car[Q], sl2[Q], slz[Q], szz[Q], co[Q], lj);
        if (lead) {
          double *c = CS + Q * 8 * WPB + (pi & 1) * 4 * WPB;
          c[4 * i + 0] = co[Q][0]; c[4 * i + 1] = co[Q][1]; c[4 * i + 2] = co[Q][2]; c[4 * i + 3] = co[Q][3];
          if (col >= Q && col < m + Q) {
            pm[Q] *= dg;
            pe[Q] += __builtin_amdgcn_frexp_exp(pm[Q]);
            pm[Q] = __builtin_amdgcn_frexp_mant(pm[Q]);
            LL[Q * NS + col] = lj;
          }
        }
        if constexpr (Q + 1 < TK) {
          const int im = m + Q - p0;
          if (im >= 0 && im < WPB) {   // tick Q's new row m + Q = (l_Q, d_Q) joins the block, with its z
            const double kss = (kid == K_RBF_BROWNIAN) ? pr[9] * pr[10] * fabs(XN[Q * MAXD]) : pr[9];
            const double s2 = row_sum(sl2[Q]), sz = row_sum(slz[Q]);
            double dd = kss + noise + 1e-8 - s2;
            if (!(dd > 0.0)) {
              if (badq == 0) badq = t + Q + 1;
              dd = 1e-300;
            }
            dq[Q] = sqrt(dd);
            znq[Q] = (p.ys[(size_t)w * p.T + t + Q] - sz) / dq[Q];
            if (i == im) {
#pragma unroll
              for (int j = 0; j < WPB; ++j) a[j] = j < im ? LL[Q * NS + p0 + j] : 0.0;
              dg = dq[Q];
              zi = znq[Q];
            }
          }
        }
      });
      if (lead && col < nrow) {
        const unsigned off = eoff(col, p0);
#pragma unroll
        for (int j = 0; j < WPB - 1; ++j)
          if (p0 + j < nrow) st64(a[j], off, j * colb);
        st64(dg, off + (unsigned)(i * colb), 0);
        z[o1 + col] = zi;
      }
    };
    // sweep waves: trips of a panel (wave-uniform), a slot's row, the request of a trip
    auto trips_of = [&](int pi) {
      const int tot = nrow - (pi * WPB + 2 * WPB) - (wave - 1) * 64;
      return tot > 0 ? (tot + 191) / 192 : 0;
    };
    auto slot_row = [&](int pi, int n, int r, int &rr, bool &ok) {
      const int R = nrow - (pi * WPB + 2 * WPB);
      const int idx = (wave - 1) * 64 + r + 192 * n;
      ok = idx < R;
      rr = pi * WPB + 2 * WPB + (ok ? idx : 0);
    };
    auto issue_stage = [&](int pi, int n) {
      typedef __attribute__((address_space(3))) void lds_void;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int rr;
        bool ok;
        slot_row(pi, n, 32 * h + (lane >> 1), rr, ok);
        const unsigned vo = eoff(rr, pi * WPB) + 4u * (lane & 1);
#pragma unroll
        for (int c = 0; c < WPB; ++c)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(stg + c * 64 + 32 * h), 4, vo, c * colb, 0, 0);
      }
    };
    bool first_trip = true;
    if (wave > 0) {
      int pi0 = 0;
      while (pi0 < npan && trips_of(pi0) == 0) ++pi0;
      if (pi0 < npan) issue_stage(pi0, 0);
    }
    if (wave == 0) {
      double ad[WPB], dg;
      {
        const unsigned off = eoff(i, 0);
        const bool in = i < m;
#pragma unroll
        for (int j = 0; j < WPB; ++j) ad[j] = (j < i && in) ? ld64(off, j * colb) : 0.0;
        dg = in ? ld64(off + (unsigned)(i * colb), 0) : 1.0;
        zi = in ? z[o1 + i] : 0.0;
        uq[0] = in ? VV[i] : 0.0;
#pragma unroll
        for (int q = 0; q < TK; ++q) kq[q] = in ? KK[q * NS + i] : 0.0;
      }
      block_all(0, ad, dg);
    }
    for (int pi = 0; pi < npan; ++pi) {
      lds_barrier();
      const int p0 = pi * WPB;
      if (wave == 0) {
        if (pi + 1 < npan) {
          const int rr = p0 + WPB + i;
          const bool old = rr < m;
          double ad[WPB], dg, a[WPB];
          {
            const unsigned off = eoff(old ? rr : 0, p0 + WPB);
#pragma unroll
            for (int j = 0; j < WPB; ++j) {
              const double x = ld64(off, j * colb);
              ad[j] = (j < i && old) ? x : 0.0;
            }
            const double xd = ld64(eoff(old ? rr : 0, old ? rr : 0), 0);
            dg = old ? xd : 1.0;
          }
          const double zn = old ? z[o1 + rr] : 0.0;
          {
            const unsigned off = eoff(old ? rr : 0, p0);
#pragma unroll
            for (int j = 0; j < WPB; ++j) {
              const double x = ld64(off, j * colb);
              a[j] = old ? x : 0.0;
            }
          }
          // residuals of the rows that exist for a tick (row rr takes part in tick q while rr < m + q)
#pragma unroll
          for (int q = 0; q < TK; ++q) {
            uq[q] = (rr < m + q && (q == 0 || pi > 0)) ? VV[q * NS + rr] : 0.0;
            kq[q] = rr < m + q ? KK[q * NS + rr] : 0.0;
          }
          // every lane takes every tick's sweep (the DPP reads need the whole row active): rows that do not exist yet carry zeros;
          // a born row's entries come from LDS in front of the first tick that takes it
          static_for<0, TK>([&](auto qc) {
            constexpr int Q = decltype(qc)::value;
            sweep_row_w0(a, uq[Q], kq[Q], co[Q]);
            if constexpr (Q + 1 < TK) {
              if (p0 + 2 * WPB > m + Q && p0 + WPB <= m + Q) {   // the row tick Q bears is one of these sixteen
#pragma unroll
                for (int j = 0; j < WPB; ++j) a[j] = rr == m + Q ? LL[Q * NS + p0 + j] : a[j];
              }
              if (pi == 0) uq[Q + 1] = rr < m + Q + 1 ? a[Q] : 0.0;
            }
          });
          if (lead && rr < nrow) {
            const unsigned off = eoff(rr, p0);
#pragma unroll
            for (int j = 0; j < WPB; ++j) st64(a[j], off, j * colb);
          }
          zi = zn;
          block_all(pi + 1, ad, dg);
        }
      } else {
        const int nt = trips_of(pi);
        for (int n = 0; n < nt; ++n) {
          if (first_trip) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
          first_trip = false;
          int rr;
          bool ok;
          slot_row(pi, n, lane, rr, ok);
          double a[WPB];
#pragma unroll
          for (int j = 0; j < WPB; ++j) a[j] = stg[j * 64 + lane];
          if (rr >= m) {   // a row born inside the pass has no memory yet
#pragma unroll
            for (int j = 0; j < WPB; ++j) a[j] = 0.0;
          }
          double u[TK], k[TK];
#pragma unroll
          for (int q = 0; q < TK; ++q) {
            u[q] = (q == 0 || pi > 0) ? VV[q * NS + rr] : 0.0;
            k[q] = KK[q * NS + rr];
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the stage has been read: the next trip may land in it
          {
            int npi = pi, nn = n + 1;
            if (nn >= nt) {
              nn = 0;
              do ++npi; while (npi < npan && trips_of(npi) == 0);
            }
            if (npi < npan) issue_stage(npi, nn);
          }
          if (ok) {
#pragma unroll
            for (int q = 0; q < TK; ++q) {
              if (rr < m + q) sweep_row(a, u[q], k[q], CS + q * 8 * WPB + (pi & 1) * 4 * WPB);
              if (q + 1 < TK && rr == m + q) {   // the row tick q bears: its entries are l_q, in front of the first tick that takes it
#pragma unroll
                for (int j = 0; j < WPB; ++j) a[j] = LL[q * NS + p0 + j];
              }
              if (pi == 0 && q + 1 < TK) u[q + 1] = a[q];
            }
            const unsigned off = eoff(rr, p0);
#pragma unroll
            for (int j = 0; j < WPB; ++j) st64(a[j], off, j * colb);
#pragma unroll
            for (int q = 0; q < TK; ++q) {
              VV[q * NS + rr] = u[q];
              KK[q * NS + rr] = k[q];
            }
          }
        }
      }
    }
    __syncthreads();

    // ---- outputs of the TK ticks; the last tick's new row
    if (wave == 0) {
#pragma unroll
      for (int q = 0; q < TK; ++q) {
        double la = log(pm[q]) + (double)pe[q] * 0.6931471805599453;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) la += __shfl_xor(la, off);
        const double s2 = row_sum(sl2[q]), sz = row_sum(slz[q]), szq = row_sum(szz[q]);
        if (tid == 0) {
          double *r = RED + q * 8;
          r[0] = s2; r[1] = sz; r[2] = la; r[3] = szq; r[4] = dq[q]; r[5] = znq[q];
        }
      }
    }
    __syncthreads();
    for (int cc = TK - 1 + tid; cc < nrow; cc += 256)
      Lg[(size_t)(o1 + cc) * CAP + o1 + nrow] = LL[(TK - 1) * NS + cc];
    if (tid == 0) {
      if (bad == 0 && badq != 0) bad = badq;
#pragma unroll
      for (int q = 0; q < TK; ++q) {
        const double *r = RED + q * 8;
        const double kss = (kid == K_RBF_BROWNIAN) ? pr[9] * pr[10] * fabs(XN[q * MAXD]) : pr[9];
        const double yq = p.ys[(size_t)w * p.T + t + q];
        const size_t oi = (size_t)w * p.T + t + q;
        double dd = r[4], zn = r[5];
        if (q == TK - 1) {
          double d2 = kss + noise + 1e-8 - r[0];
          if (!(d2 > 0.0)) {
            if (bad == 0) bad = t + q + 1;
            d2 = 1e-300;
          }
          dd = sqrt(d2);
          zn = (yq - r[1]) / dd;
          Lg[(size_t)(o1 + nrow) * CAP + o1 + nrow] = dd;
          z[o1 + nrow] = zn;
        }
        double pv = kss - r[0];
        pv = pv < 1e-15 ? 1e-15 : pv;
        p.pred_mean[oi] = r[1];
        p.pred_var[oi] = p.include_noise ? pv + noise : pv;
        p.logml[oi] = -0.5 * (r[3] + zn * zn) - (r[2] + log(dd)) - 0.5 * (double)N * 1.8378770664093453;
        yw[o1 + m + q] = yq;
        for (int c = 0; c < d; ++c) xw[c * CAP + o1 + m + q] = XN[q * MAXD + c];
      }
    }
    o += TK;
    __syncthreads();
  }
  if (tid == 0) {
    int *st = p.state + w * 4;
    st[0] = o;
    st[1] = N;
    st[2] = bad;
    st[3] += p.nt;
    if (p.info_out) p.info_out[w] = bad;
  }
}


}  // namespace cgp
