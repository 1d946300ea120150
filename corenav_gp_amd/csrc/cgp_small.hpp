// cgp_small.hpp -- the reference's own operating point in ONE launch.
//
// gp_slip_node.py:27-36 fits a window of at most 134 kept ticks and runs m.optimize() on it: a few dozen evaluations of
// -logML and its gradient on a matrix that is 72 KB as a packed fp64 lower triangle -- it fits one CU's LDS.  Up to
// round 3 every evaluation was seven launches of the large-window machinery (128 x 128 tiles, factor panel in HBM) and a
// host round trip: 0.22 ms each, 5.9 ms per callback, all of it launch and PCIe latency.  Here one workgroup per
// window does everything without leaving the CU:
//
//   Gram (RBF x Brownian, SE-iso, SE-ARD; + (sigma_n^2 + 1e-8 + jitter) I)          packed lower 16 x 16 blocks in LDS
//   Cholesky + W = L^-1 in place (GPy: jitchol / dpotrf, dtrtri)                     factor_block16 on wave 0, MFMA
//                                                                                    f64 16x16x4 panels / updates /
//                                                                                    inverse rows on the other waves
//   z = W y, alpha = W^T z, logML = -z'z/2 - sum log L_ii - N/2 log 2 pi              (GPy: dpotrs, the likelihood)
//   Ky^-1 = W^T W block by block in registers, contracted with alpha alpha^T - Ky^-1 and dK/dtheta from the inputs
//                                                                                    (GPy: dL_dK = (alpha alpha^T - Ky^-1)/2)
//   GPy's jitter ladder (jitchol: mean(diag) 1e-6 10^k, k = 0..4) around the evaluation
//   L-BFGS on the Logexp-transformed parameters (lbfgs_core.hpp, lane 0)             (paramz: m.optimize())
//
// so cgp_optimize / cgp_optimize_batch / cgp_nll_grad of a window of N <= 160 samples are one launch and one copy
// back.  fp64 only (the reference's arithmetic; RBF x Brownian on raw ticks is not a single-precision problem).
//
// Block storage: block (bi, bj), bi >= bj, at sm_tri(bi, bj); element (r, c) of a block at c * 17 + r.  The odd column
// stride makes BOTH orientations of an MFMA operand read conflict-free (lanes along r: contiguous; lanes along c: stride
// 17 doubles = 34 banks, sixteen distinct even banks), which is what lets the inverse and W^T W read blocks transposed.
// fp64 v_mfma_f64_16x16x4: lane (l15 = lane & 15, lq = lane >> 4) supplies A[m = l15][k = 4 ks + lq], B[k = 4 ks + lq][n = l15]
// and holds D[m = lq + 4 reg][n = l15] -- register `reg` of D is exactly the B operand of k-step `reg`, so products chain
// in registers.
#pragma once
#include "cgp_kernels.hpp"
#include "lbfgs_core.hpp"

namespace cgp {

constexpr int SM_MAX_NB = 10;              // block rows at most: windows of up to 160 samples
constexpr int SM_MAX_N = SM_MAX_NB * DB;
constexpr int SM_LD = DB + 1;
constexpr int SM_BLK = DB * SM_LD;         // elements per block
constexpr int SM_THREADS = 512;
constexpr int SM_WAVES = SM_THREADS / 64;
#ifndef SM_HELPERS
#define SM_HELPERS 7                       // `make variant` A/B: 6 = the chain wave's SIMD partner (wave 4) stays idle during the factor phases
#endif
constexpr int SM_NH = SM_HELPERS;          // helper waves beside the factor chain.  Measured (cycles per evaluation, N = 134): with 6 the chain
                                           // issues alone on its SIMD (31 k) but the helpers end later (factor phases 48 k); with 7 the chain
                                           // slows to 38 k and the phases end at 43.5 k
constexpr int SM_TAB_MAX = 1024;           // entries of the tick-grid RBF table (8 KB)
constexpr int SM_DEAL = 16;                // entries of a helper's work list: count + at most 15 items (ten block rows, six helpers: 52 items of a step, at most 12 on one)
constexpr int SM_OUT = 48;                 // doubles per window in SmallArgs::out ([32, 48): phase clocks of a -DCGP_ABLATION build)
enum { SM_MODE_EVAL = 0, SM_MODE_OPT = 1 };
// out[]: 0 logML (at theta / at the optimum), 1 evaluations, 2 L-BFGS status, 3 iterations, 4 info (first non-positive
// pivot after the jitter ladder, 0 = ok), 5 jitter of the last evaluation, 8.. d(-logML)/dtheta (natural parameters; OPT: wrt
// the Logexp variables, at the optimum), 20.. theta (OPT: the optimum)
enum { SMO_LOGML = 0, SMO_EVALS = 1, SMO_STATUS = 2, SMO_ITERS = 3, SMO_INFO = 4, SMO_JITTER = 5, SMO_GRAD = 8, SMO_THETA = 20 };

__host__ __device__ __forceinline__ constexpr int sm_tri(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * SM_BLK; }

struct SmallArgs {
  const double *X;   // [windows][d][N]
  const double *y;   // [windows][N]
  double *theta;     // [windows][MAX_THETA] natural parameters: evaluation point (EVAL) / start in, optimum out (OPT)
  double *out;       // [windows][SM_OUT]
  int N, d, kernel_id, nth, mode, max_evals;
  double pgtol, factr;
  // element (point r, coordinate q) of X at X[q * x_sq + r * x_sr], of Xs at Xs[q * xs_sq + r * xs_sr]: the engine's resident
  // layout is coordinate-major ({N, 1} / {M, 1}); a lone window is read as the caller staged it, point-major ({1, d}), in place
  int x_sq, x_sr, xs_sq, xs_sr;
  const unsigned short *deal;   // [NB][SM_NH][SM_DEAL] helper work lists of this NB (sm_build_deal, built by the host)
  // k_small_predict: test points [windows][d][M], outputs [windows][M], `parts` workgroups per window (each refits, each predicts a slice)
  const double *Xs;
  double *mean, *var;
  int M, include_noise, parts;
  // RBF x Brownian on a tick grid: every input (window and test points) integer-valued, |x| <= 2^26, spread < tab_n <= SM_TAB_MAX:
  // r^2 = -2 x x' + (x^2 + x'^2) is then the exact integer (x - x')^2 and the RBF factor takes tab_n values per evaluation, which
  // the kernels keep in LDS (same expression, same bits as the direct evaluation).  0: no table.  The host decides (it has the
  // arrays: node callbacks, cgp_nll_grad, cgp_optimize); device-resident batches run without.
  int tab_n;
  const double *jitter;   // per window or NULL: the jitter to start from
  int ladder;             // 1: GPy's jitchol ladder inside the launch; 0: one attempt at the given jitter, the pivot reported
  double *logml;          // per window or NULL (beside the record)
  int *info;
  // k_small_predict, the node callback: the LAST workgroup to finish (done_count, device, returns to 0) stores done_seq to done_flag
  // (pinned host memory) behind a system-scope fence -- the host polls that word instead of synchronising the stream.  NULL: off.
  int *done_count, *done_flag;
  int done_seq;
};

// Block rows of k_small_predict's V = W K* phase -> waves: row bi costs bi + 1 products.  The two waves of a SIMD (w and w + 4) share
// one fp64 MFMA pipe (one MFMA per 64 cycles, the older wave first: tools/mfma_dep_latency.hip), so rows are dealt longest first to
// the SIMD with the least work and, inside it, to its wave with the least (round-robin rows {w, w + 8} left SIMD 0 with 15 of the
// 45 products of nine block rows and its wave 0 with 10 in a row).  One 16-bit row mask per wave, behind the work lists.
__host__ __device__ inline void sm_build_rowmap(unsigned short *map, int NB) {
  int load[SM_WAVES];
  for (int w = 0; w < SM_WAVES; ++w) load[w] = 0, map[w] = 0;
  for (int bi = NB - 1; bi >= 0; --bi) {
    int sd = 0;
    for (int q = 1; q < 4; ++q) sd = load[q] + load[q + 4] < load[sd] + load[sd + 4] ? q : sd;
    const int w = load[sd + 4] < load[sd] ? sd + 4 : sd;
    load[w] += bi + 1;
    map[w] = (unsigned short)(map[w] | (1u << bi));
  }
}
// per-NB table the host builds once per context: [NB][SM_NH][SM_DEAL] work lists, then SM_WAVES row masks
__host__ __device__ __forceinline__ constexpr size_t small_table_elems(int NB) { return (size_t)NB * SM_NH * SM_DEAL + SM_WAVES; }

// extra dynamic LDS of k_small_predict: one 16-column chunk of K* (NB blocks), its test points, partial sums
__host__ __device__ __forceinline__ constexpr size_t small_predict_lds_extra(int NB, int d) {
  return sizeof(double) * ((size_t)NB * SM_BLK + (size_t)d * DB + (size_t)NB * DB + SM_WAVES * DB + 16);
}
// dynamic LDS of k_small for a window of NB block rows and d input dimensions
__host__ __device__ __forceinline__ constexpr size_t small_lds_bytes(int NB, int d) {
  return sizeof(double) * ((size_t)NB * (NB + 1) / 2 * SM_BLK + (size_t)(d + 7) * NB * DB + SM_WAVES * GRAD_N + 64) +
         sizeof(corenav::LbfgsCore) + sizeof(int) * (8 + NB * (NB + 1) / 2) + 2 * (size_t)NB * SM_NH * SM_DEAL + 64;
}

struct SmallLds {
  double *Bk;    // packed blocks
  double *xr;    // [d][NP] raw inputs
  double *yv, *zv, *al, *ldg;   // [NP] each: y, z = W y, alpha, diag(L)
  double *tmp;   // [3][NP] partial sums of z / alpha
  double *red;   // [SM_WAVES][GRAD_N] reductions; between evaluations lane 0's gradient vectors
  double *sc;    // 64 scalars: [0..9] theta, [10..17] 1/ell_q, 18 amp, 19 amp_b, 20 diag add, 21 jitter, 22 logML, 23 mean |x|,
                 //             [24..35] gradient sums, [36..45] dtheta/dx of the Logexp transform, [48..63] phase clocks
  corenav::LbfgsCore *lb;
  int *flag;     // [0] first non-positive pivot of the running evaluation, [1] optimiser finished, [2] ladder attempt
  int *tb;       // [blocks] packed block index -> block row | block column << 8
  unsigned short *deal;   // [NB][SM_NH][SM_DEAL] helper work lists (sm_build_deal)
  double *etab;           // [tab_n] sigma_r^2 exp(-(i / ell)^2 / 2) of the tick-grid form, rebuilt per evaluation; tab_n = 0: none
  int tab_n;
};

// Phase clocks (-DCGP_ABLATION builds only): lane 0 adds the s_memtime ticks since the previous lap to sc[48 + slot]; k_small
// copies them to out[32..].  Slots: 0 constants, 1 Gram, 2 F phases, 3 P, 4 U, 5 last row of W, 6 z / alpha / logML, 7 gradient
// sums, 8 lane-0 step (gradient, L-BFGS), 9 evaluations.  tools/small_phases.py.
struct SmClock {
#ifdef CGP_ABLATION
  long long t;
  __device__ __forceinline__ void start() { t = __builtin_amdgcn_s_memtime(); }
  __device__ __forceinline__ void lap(const double *sc_, int slot, int tid) {
    const long long n = __builtin_amdgcn_s_memtime();
    if (tid == 0) const_cast<double *>(sc_)[48 + slot] += (double)(n - t);
    t = n;
  }
#else
  __device__ __forceinline__ void start() {}
  __device__ __forceinline__ void lap(const double *, int, int) {}
#endif
};

// Who does what beside the factor chain (sm_eval, phase F): for every step jb the items -- blocks (jp, j) of row jp = jb - 1
// of W, cost jp - j + 1 products, code 0x100 | j; trailing blocks (bi, bj), bi > jb, bj >= jb, with panel jp, one product, code bi << 4 | bj --
// are dealt to the SM_NH helper waves longest first, each to the helper with the least work so far.  Lists
// deal[jb][helper] = {count, items...}; depends on the number of block rows only: the host builds the table of every NB once
// per context (a thread building its list in the kernel indexes load[] / cnt[] dynamically, i.e. in scratch memory: 39 us of
// a 77 us launch when it was done there) and a launch copies its 2 KB into LDS.
// Returns false when the deal does not fit what sm_eval can hold -- a list longer than SM_DEAL - 1 items, or more than two W
// blocks (code 0x100 | j) for one helper: sm_eval keeps two pending W blocks per helper (wres[2] / wj[2]).  Holds for every
// NB <= SM_MAX_NB with the shipped SM_HELPERS; cgp_create checks it for the tables it builds and refuses the context
// otherwise (SM_HELPERS is a `make variant` knob), instead of a wrong factor with no error.
__host__ __device__ inline bool sm_build_deal(unsigned short *deal, int NB, int jb) {
  int load[SM_NH], cnt[SM_NH], nw[SM_NH];
  bool fits = true;
  for (int h = 0; h < SM_NH; ++h) load[h] = cnt[h] = nw[h] = 0;
  unsigned short *base = deal + jb * SM_NH * SM_DEAL;
  auto give = [&](int code, int cost) {
    int h = 0;
    for (int hh = 1; hh < SM_NH; ++hh) h = load[hh] < load[h] ? hh : h;
    load[h] += cost;
    if (code & 0x100) fits = fits && ++nw[h] <= 2;
    if (cnt[h] < SM_DEAL - 1) base[h * SM_DEAL + 1 + cnt[h]++] = (unsigned short)code;
    else fits = false;
  };
  const int jp = jb - 1;
  for (int j = 0; j < jp; ++j) give(0x100 | j, jp - j + 1);
  for (int bi = jb + 1; bi < NB; ++bi)          // panel jp applied to every block below / right of the diagonal block jb,
    for (int bj = jb; bj <= bi; ++bj) give(bi << 4 | bj, 1);   // column jb (rows >= jb + 1) included: block (jb, jb) is the chain's
  for (int h = 0; h < SM_NH; ++h) base[h * SM_DEAL] = (unsigned short)cnt[h];
  return fits;
}

// Cross-lane sums on the VALU's DPP path instead of ds_bpermute (what __shfl_xor compiles to: an LDS round trip per step, ~100
// cycles each when the steps depend on each other -- 72 of them closed every evaluation's gradient phase, ~50 every L-BFGS
// step).  A double moves as two 32-bit halves.  sm_row_sum: every lane gets the total of its row of 16 (rotations by 8, 4, 2,
// 1: the same tree in every lane up to the order of each addition's two operands, so all lanes hold identical bits);
// sm_wave_sum: the four row totals added in fixed order through scalar registers, every lane gets the wave's total.
template <int CTRL> __device__ __forceinline__ double sm_dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sm_row_sum(double v) {
  v += sm_dpp_mov<0x128>(v);   // row_ror:8
  v += sm_dpp_mov<0x124>(v);   // row_ror:4
  v += sm_dpp_mov<0x122>(v);   // row_ror:2
  v += sm_dpp_mov<0x121>(v);   // row_ror:1
  return v;
}
__device__ __forceinline__ double sm_row_max(double v) {
  v = __builtin_fmax(v, sm_dpp_mov<0x128>(v));
  v = __builtin_fmax(v, sm_dpp_mov<0x124>(v));
  v = __builtin_fmax(v, sm_dpp_mov<0x122>(v));
  v = __builtin_fmax(v, sm_dpp_mov<0x121>(v));
  return v;
}
__device__ __forceinline__ double sm_lane_value(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double sm_wave_sum(double v) {
  v = sm_row_sum(v);
  return ((sm_lane_value(v, 0) + sm_lane_value(v, 16)) + sm_lane_value(v, 32)) + sm_lane_value(v, 48);
}

// mean |x| of the first input (jitchol's mean(diag) for the Brownian factor) by one wave, straight from the window in HBM
__device__ __forceinline__ void sm_mean_abs(double *dst, const double *x, int N, int stride, int lane) {
  double sa = 0.0;
  for (int i = lane; i < N; i += 64) sa += fabs(x[(size_t)i * stride]);
  sa = sm_wave_sum(sa);
  if (lane == 0) *dst = sa / (double)N;
}

// K_ij of two window points (no noise term) with the per-evaluation constants in registers.  BROWN: the reference's
// RBF(1) x Brownian(1) (gp_slip_node.py:31) in GPy's form -- r^2 = -2 x x' + (x^2 + x'^2), clipped at 0, forced 0 on the
// diagonal; sigma_b^2 min(|x|, |x'|) where the signs agree -- else SE-iso / SE-ARD over d <= DMAX length-scaled
// differences.  dq2[q] = the length-scaled squared difference of coordinate q (what dK/d ell_q multiplies K by, up to 1 / ell_q).
template <bool BROWN, int DMAX> struct SmKern {
  double amp, amp_b, iell[DMAX];
  __device__ __forceinline__ void load(const double *sc) {
    amp = sc[18];
    amp_b = sc[19];
#pragma unroll
    for (int q = 0; q < DMAX; ++q) iell[q] = sc[10 + q];
  }
  // k(training point gi, test point c of the staged chunk xt[q * 16 + c])
  __device__ __forceinline__ double cross(const double *xr, const double *xt, int d, int NP, int gi, int c, const ExpC &ec) const {
    if constexpr (BROWN) {
      const double x = xr[gi], xp = xt[c];
      double r2 = -2.0 * x * xp + (x * x + xp * xp);
      r2 = r2 < 0.0 ? 0.0 : r2;
      const bool same = (x > 0.0 && xp > 0.0) || (x < 0.0 && xp < 0.0) || (x == 0.0 && xp == 0.0);
      const double ax = __builtin_fabs(x), ap = __builtin_fabs(xp);
      const double kb = same ? amp_b * (ax < ap ? ax : ap) : 0.0;
      return amp * exp_nonpos(-0.5 * r2 * (iell[0] * iell[0]), ec) * kb;
    } else {
      double d2 = 0.0;
#pragma unroll
      for (int q = 0; q < DMAX; ++q)
        if (q < d) {
          const double df = (xr[q * NP + gi] - xt[q * DB + c]) * iell[q];
          d2 += df * df;
        }
      return amp * exp_nonpos(-0.5 * d2, ec);
    }
  }
  // tick-grid forms (SmallArgs::tab_n): the RBF factor from the table, index |x - x'| (an exact integer)
  __device__ __forceinline__ double brownian(double x, double xp) const {
    const bool same = (x > 0.0 && xp > 0.0) || (x < 0.0 && xp < 0.0) || (x == 0.0 && xp == 0.0);
    const double ax = __builtin_fabs(x), ap = __builtin_fabs(xp);
    return same ? amp_b * (ax < ap ? ax : ap) : 0.0;
  }
  // (tmax = tab_n - 1 as a double: padded rows / columns are evaluated and discarded, their index must stay inside the table)
  __device__ __forceinline__ double eval_tab(const double *xr, const double *etab, double tmax, int gi, int gj) const {
    const double x = xr[gi], xp = xr[gj];
    return etab[(int)__builtin_fmin(__builtin_fabs(x - xp), tmax)] * brownian(x, xp);
  }
  __device__ __forceinline__ double cross_tab(const double *xr, const double *xt, const double *etab, double tmax, int gi, int c) const {
    const double x = xr[gi], xp = xt[c];
    return etab[(int)__builtin_fmin(__builtin_fabs(x - xp), tmax)] * brownian(x, xp);
  }
  // the table itself: entry i = amp exp(-0.5 (i^2 / ell^2)), the expression eval / cross evaluate for r^2 = i^2
  __device__ __forceinline__ void fill_tab(double *etab, int tab_n, int tid, const ExpC &ec) const {
    for (int i = tid; i < tab_n; i += SM_THREADS) {
      const double r2 = (double)i * (double)i;
      etab[i] = amp * exp_nonpos(-0.5 * (r2 * (iell[0] * iell[0])), ec);
    }
  }
  // the length-scaled squared differences alone (the gradient phase has K itself from the Gram phase, in registers)
  __device__ __forceinline__ void diffs(const double *xr, int d, int NP, int gi, int gj, double (&dq2)[DMAX]) const {
    if constexpr (BROWN) {
      const double x = xr[gi], xp = xr[gj];
      double r2 = (gi == gj) ? 0.0 : (-2.0 * x * xp + (x * x + xp * xp));
      r2 = r2 < 0.0 ? 0.0 : r2;
      dq2[0] = r2 * (iell[0] * iell[0]);
    } else {
#pragma unroll
      for (int q = 0; q < DMAX; ++q) {
        double v = 0.0;
        if (q < d) {
          const double df = (xr[q * NP + gi] - xr[q * NP + gj]) * iell[q];
          v = df * df;
        }
        dq2[q] = v;
      }
    }
  }
  __device__ __forceinline__ double eval(const double *xr, int d, int NP, int gi, int gj, const ExpC &ec, double (&dq2)[DMAX]) const {
    if constexpr (BROWN) {
      const double x = xr[gi], xp = xr[gj];
      double r2 = (gi == gj) ? 0.0 : (-2.0 * x * xp + (x * x + xp * xp));
      r2 = r2 < 0.0 ? 0.0 : r2;
      const double q2 = r2 * (iell[0] * iell[0]);
      const bool same = (x > 0.0 && xp > 0.0) || (x < 0.0 && xp < 0.0) || (x == 0.0 && xp == 0.0);
      const double ax = __builtin_fabs(x), ap = __builtin_fabs(xp);
      const double kb = same ? amp_b * (ax < ap ? ax : ap) : 0.0;
      dq2[0] = q2;
      return amp * exp_nonpos(-0.5 * q2, ec) * kb;
    } else {
      double d2 = 0.0;
#pragma unroll
      for (int q = 0; q < DMAX; ++q) {
        double v = 0.0;
        if (q < d) {
          const double df = (xr[q * NP + gi] - xr[q * NP + gj]) * iell[q];
          v = df * df;
        }
        dq2[q] = v;
        d2 += v;
      }
      return amp * exp_nonpos(-0.5 * d2, ec);
    }
  }
};

// One evaluation at the natural parameters in s.sc[0..9] with jitter s.sc[21] (1 / ell_q, amplitudes and the diagonal
// addend already in s.sc[10..20]: sm_prepare): on return (after the final barrier) s.flag[0] = first non-positive
// pivot (0 = positive definite), s.sc[22] = logML, s.sc[24..35] = gradient sums (k_grad's layout: [0] amplitude,
// [1..8] length-scales, [9] noise).  Every thread of the workgroup calls it.
template <bool BROWN, int DMAX, bool GRAD = true>
__device__ __forceinline__ void sm_eval(const SmallLds &s, int d, int N, int NB, int tid) {
  using P = Prec<double>;
  using acc_t = P::acc_t;
  const int NP = NB * DB, nblk = NB * (NB + 1) / 2;
  const int lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double *Bk = s.Bk;
  ExpC ec;
  ec.load_literals();
  SmClock ck;
  ck.start();
  SmKern<BROWN, DMAX> kern;
  kern.load(s.sc);
  const bool tab = BROWN && s.tab_n > 0;
  if constexpr (BROWN) {
    if (tab) {
      kern.fill_tab(s.etab, s.tab_n, tid, ec);
      __syncthreads();
    }
  }
  // ---- Gram, in the mapping the gradient phase uses (wave w: blocks w, w + 8, ...; lane (l15, lq): entries (row lq + 4 r, column
  // l15)), so that the covariances stay in registers (kv) for the contraction at the end instead of being evaluated twice
  constexpr int KVB = (SM_MAX_NB * (SM_MAX_NB + 1) / 2 + SM_WAVES - 1) / SM_WAVES;   // blocks per wave at most
  double kv[KVB][4];
  {
    const double diag_add = s.sc[20];
#pragma unroll
    for (int i = 0; i < KVB; ++i) {
      const int blk = wave + SM_WAVES * i;
      if (blk < nblk) {
        const int t = s.tb[blk], bi = t & 255, bj = t >> 8;
        const int gj = bj * DB + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gi = bi * DB + lq + 4 * r;
          double dq2[DMAX];
          double k0 = 0.0, g;
          if (gi < N && gj < N) {
            if constexpr (BROWN) k0 = tab ? kern.eval_tab(s.xr, s.etab, (double)(s.tab_n - 1), gi, gj) : kern.eval(s.xr, d, NP, gi, gj, ec, dq2);
            else k0 = kern.eval(s.xr, d, NP, gi, gj, ec, dq2);
            g = gi == gj ? k0 + diag_add : k0;
          } else g = (gi == gj) ? 1.0 : 0.0;   // identity padding keeps the factor well defined
          kv[i][r] = GRAD ? k0 : 0.0;
          Bk[blk * SM_BLK + l15 * SM_LD + lq + 4 * r] = g;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) kv[i][r] = 0.0;
      }
    }
  }
  __syncthreads();
  ck.lap(s.sc, 1, tid);

  // ---- Cholesky and W = L^-1 in place, organised around the one serial chain (the NB diagonal blocks on wave 0):
  //   F(jb)  wave 0: factor + invert diagonal block jb in registers
  //          the other waves, meanwhile: row jb-1 of W (W_ij = -Dinv_i sum_{j <= q < i} L_iq W_qj; kept in registers
  //          until the barrier: it overwrites L_ij, which other blocks of the row still read) and the trailing update
  //          with panel jb-1 of everything below / right of block (jb, jb)
  //   P(jb)  panel L(i,jb) = A(i,jb) Dinv_jb^T; the chain wave also updates the NEXT diagonal block with it
  auto inverse_block = [&](int i, int j) -> acc_t {   // W(i, j), i > j, returned in the accumulator layout
    acc_t t0 = acc_t{0, 0, 0, 0}, t1 = t0;
    int q = j;
    for (; q + 1 < i; q += 2) {   // two products per trip, every operand read before the first MFMA, two accumulator chains
      const double *la = Bk + sm_tri(i, q), *wa = Bk + sm_tri(q, j), *lb = Bk + sm_tri(i, q + 1), *wb = Bk + sm_tri(q + 1, j);
      double fa[4], fb[4], ga[4], gb[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        fa[ks] = la[(4 * ks + lq) * SM_LD + l15];   // L(i,q)[m = l15][k]
        fb[ks] = wa[l15 * SM_LD + 4 * ks + lq];     // W(q,j)[k][n = l15]  (q = j: Dinv_j)
        ga[ks] = lb[(4 * ks + lq) * SM_LD + l15];
        gb[ks] = wb[l15 * SM_LD + 4 * ks + lq];
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        t0 = P::mfma(fa[ks], fb[ks], t0);
        t1 = P::mfma(ga[ks], gb[ks], t1);
      }
    }
    if (q < i) {
      const double *la = Bk + sm_tri(i, q), *wa = Bk + sm_tri(q, j);
      double fa[4], fb[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        fa[ks] = la[(4 * ks + lq) * SM_LD + l15];
        fb[ks] = wa[l15 * SM_LD + 4 * ks + lq];
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) t0 = P::mfma(fa[ks], fb[ks], t0);
    }
    const double *di = Bk + sm_tri(i, i);
    double ga[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ga[ks] = -di[(4 * ks + lq) * SM_LD + l15];   // -Dinv_i[m = l15][k]
    acc_t w = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w = P::mfma(ga[ks], t0[ks] + t1[ks], w);      // T's register ks IS the B operand of k-step ks
    return w;
  };
  auto store_acc = [&](double *blk, const acc_t &a) {   // block[row = lq + 4 reg][col = l15]
#pragma unroll
    for (int r = 0; r < 4; ++r) blk[l15 * SM_LD + lq + 4 * r] = a[r];
  };
  auto trailing_block = [&](int bi, int bj, int jp) {   // C(bi,bj) -= L(bi,jp) L(bj,jp)^T
    double *cb = Bk + sm_tri(bi, bj);
    const double *la = Bk + sm_tri(bj, jp), *lb = Bk + sm_tri(bi, jp);
    acc_t acc;
    double fa[4], fb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = cb[(lq + 4 * r) * SM_LD + l15];        // C[row = l15][col = lq + 4 r]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fa[ks] = -la[(4 * ks + lq) * SM_LD + l15];
      fb[ks] = lb[(4 * ks + lq) * SM_LD + l15];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = P::mfma(fa[ks], fb[ks], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) cb[(lq + 4 * r) * SM_LD + l15] = acc[r];
  };
  // P[n = l15][col lq + 4 reg] of  P = A(bi,jb) Dinv_jb^T  (block bi of panel jb), in the accumulator layout
  auto panel_block = [&](int bi, int jb) -> acc_t {
    const double *dj = Bk + sm_tri(jb, jb), *ab = Bk + sm_tri(bi, jb);
    double fa[4], fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fa[ks] = dj[(4 * ks + lq) * SM_LD + l15];        // Dinv_jb[m = l15][k]
      fb[ks] = ab[(4 * ks + lq) * SM_LD + l15];        // A(bi,jb)[n = l15][k]
    }
    acc_t acc = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = P::mfma(fa[ks], fb[ks], acc);
    return acc;
  };
  // Helpers during F: every wave but 0 (SM_HELPERS = 6: not its SIMD partner 4 either).  What each of them does in
  // step jb comes from a list built once per launch (sm_build_deal): the blocks (jp, j) of row jp = jb - 1 of W cost
  // jp - j + 1 products each, the trailing blocks one each.
  const int helper = SM_NH == 7 ? wave - 1 : (wave == 0 || wave == 4 ? -1 : (wave < 4 ? wave - 1 : wave - 2));
  for (int jb = 0; jb < NB; ++jb) {
    acc_t wres[2];
    int wj[2] = {-1, -1};
    if (wave == 0) {
      double *dblk = Bk + sm_tri(jb, jb);
      double a[DB], w[DB];
      int bad = 0;
#pragma unroll
      for (int c = 0; c < DB; ++c) a[c] = dblk[c * SM_LD + l15];
      factor_block16<double>(a, w, bad, jb * DB, l15, [&] {
#pragma unroll
        for (int c = 0; c < DB; ++c) a[c] = dblk[c * SM_LD + l15];
      });
      if (lane < DB) {
        double dg = 1.0;
#pragma unroll
        for (int c = 0; c < DB; ++c) dg = (c == l15) ? a[c] : dg;
        s.ldg[jb * DB + l15] = dg;
#pragma unroll
        for (int i = 0; i < DB; ++i) dblk[l15 * SM_LD + i] = w[i];   // Dinv_jb[i][l15] replaces the diagonal block
      }
      if (bad != 0 && lane == 0 && s.flag[0] == 0) s.flag[0] = bad;
#ifdef CGP_ABLATION
      if (tid == 0) s.sc[48 + 10] += (double)(__builtin_amdgcn_s_memtime() - ck.t);   // the chain alone, without the wait for the helpers
#endif
    } else if (helper >= 0 && jb > 0) {
      const int jp = jb - 1;
      const unsigned short *lst = s.deal + (jb * SM_NH + helper) * SM_DEAL;
      const int n = lst[0];
      for (int k = 1; k <= n; ++k) {
        const int it = lst[k];
        if (it & 0x100) {
          const int slot = wj[0] < 0 ? 0 : 1;
          wres[slot] = inverse_block(jp, it & 0xff);
          wj[slot] = it & 0xff;
        } else trailing_block(it >> 4, it & 15, jp);
      }
    }
    lds_barrier();
    ck.lap(s.sc, 2, tid);
    if (wj[0] >= 0) store_acc(Bk + sm_tri(jb - 1, wj[0]), wres[0]);
    if (wj[1] >= 0) store_acc(Bk + sm_tri(jb - 1, wj[1]), wres[1]);
    // ---- P(jb): L(bi,jb) = A(bi,jb) Dinv_jb^T, one block row per wave, stored in place; the chain wave owns block row jb + 1
    // and updates the next diagonal block C(jb+1,jb+1) -= L(jb+1,jb) L(jb+1,jb)^T straight from the accumulator (register
    // `reg` of a product is the operand of k-step `reg` of the next, file header).  Every other block of the trailing
    // matrix takes panel jb from the helpers during the next factor phase.
    for (int bi = jb + 1 + wave; bi < NB; bi += SM_WAVES) {
      const acc_t pb = panel_block(bi, jb);
      double *ab = Bk + sm_tri(bi, jb);
      if (bi == jb + 1) {
        double *cb = Bk + sm_tri(bi, bi);
        acc_t acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = cb[(lq + 4 * r) * SM_LD + l15];   // C[row = l15][col = lq + 4 r]
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = P::mfma(-pb[ks], pb[ks], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) cb[(lq + 4 * r) * SM_LD + l15] = acc[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) ab[(lq + 4 * r) * SM_LD + l15] = pb[r];      // L(bi,jb)[row = l15][col = lq + 4 r]
    }
    lds_barrier();
    ck.lap(s.sc, 3, tid);
  }
  {  // the last row of W
    acc_t wres[2];
    int wj[2] = {-1, -1};
    for (int j = wave; j < NB - 1; j += SM_WAVES) {
      const int slot = wj[0] < 0 ? 0 : 1;
      wres[slot] = inverse_block(NB - 1, j);
      wj[slot] = j;
    }
    lds_barrier();
    if (wj[0] >= 0) store_acc(Bk + sm_tri(NB - 1, wj[0]), wres[0]);
    if (wj[1] >= 0) store_acc(Bk + sm_tri(NB - 1, wj[1]), wres[1]);
  }
  __syncthreads();
  ck.lap(s.sc, 5, tid);

  // ---- z = W y, alpha = W^T z, logML: a row's (column's) blocks dealt over three threads, partial sums added in fixed order
  {
    const int part = tid / NP, i = tid - part * NP;
    if (part < 3) {
      const int bi = i >> 4, r = i & 15;
      double z = 0.0;
      for (int bj = part; bj <= bi; bj += 3) {
        const double *wb = Bk + sm_tri(bi, bj) + r;
#pragma unroll
        for (int c = 0; c < DB; ++c) z = __builtin_fma(wb[c * SM_LD], s.yv[bj * DB + c], z);
      }
      s.tmp[part * NP + i] = z;
    }
    __syncthreads();
    if (tid < NP) s.zv[tid] = (s.tmp[tid] + s.tmp[NP + tid]) + s.tmp[2 * NP + tid];
    __syncthreads();
    if (part < 3) {
      const int bj = i >> 4, c = i & 15;
      double a = 0.0;
      for (int bi = bj + part; bi < NB; bi += 3) {
        const double *wb = Bk + sm_tri(bi, bj) + c * SM_LD;
#pragma unroll
        for (int r = 0; r < DB; ++r) a = __builtin_fma(wb[r], s.zv[bi * DB + r], a);
      }
      s.tmp[part * NP + i] = a;
    }
    __syncthreads();
  }
  double lsum = 0.0;
  if (tid < NP) {
    s.al[tid] = (s.tmp[tid] + s.tmp[NP + tid]) + s.tmp[2 * NP + tid];
    const double z = s.zv[tid];
    lsum = -0.5 * z * z - (tid < N ? log(s.ldg[tid]) : 0.0);
  }
  lsum = sm_wave_sum(lsum);
  if (lane == 0) s.red[wave * GRAD_N] = lsum;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < SM_WAVES; ++w) t += s.red[w * GRAD_N];
    s.sc[22] = t - 0.5 * (double)N * 1.8378770664093453;
  }
  __syncthreads();
  ck.lap(s.sc, 6, tid);

  if constexpr (!GRAD) return;   // fit + predict (k_small_predict): no gradient
  // ---- gradient sums: Ky^-1(bi,bj) = sum_{k >= bi} W(k,bi)^T W(k,bj) in registers, contracted on the spot
  double s_amp = 0.0, s_noise = 0.0, s_ell[DMAX];
#pragma unroll
  for (int q = 0; q < DMAX; ++q) s_ell[q] = 0.0;
#pragma unroll
  for (int i = 0; i < KVB; ++i) {
    const int blk = wave + SM_WAVES * i;
    if (blk >= nblk) continue;
    const int t = s.tb[blk], bi = t & 255, bj = t >> 8;
    acc_t a0 = acc_t{0, 0, 0, 0}, a1 = a0;
    int k = bi;
#if defined(SM_PROBE) && SM_PROBE == 2   // timing probe (wrong results): no W^T W products in the gradient phase
    k = NB;
#endif
    for (; k + 1 < NB; k += 2) {   // two products per trip, operands read before the first MFMA, two accumulator chains
      const double *wa = Bk + sm_tri(k, bi), *wb = Bk + sm_tri(k, bj), *wc = Bk + sm_tri(k + 1, bi), *wd = Bk + sm_tri(k + 1, bj);
      double fa[4], fb[4], fc[4], fd[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        fa[ks] = wa[l15 * SM_LD + 4 * ks + lq];   // W(k,bi)[kk][m = l15]
        fb[ks] = wb[l15 * SM_LD + 4 * ks + lq];   // W(k,bj)[kk][n = l15]
        fc[ks] = wc[l15 * SM_LD + 4 * ks + lq];
        fd[ks] = wd[l15 * SM_LD + 4 * ks + lq];
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        a0 = P::mfma(fa[ks], fb[ks], a0);
        a1 = P::mfma(fc[ks], fd[ks], a1);
      }
    }
    if (k < NB) {
      const double *wa = Bk + sm_tri(k, bi), *wb = Bk + sm_tri(k, bj);
      double fa[4], fb[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        fa[ks] = wa[l15 * SM_LD + 4 * ks + lq];
        fb[ks] = wb[l15 * SM_LD + 4 * ks + lq];
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a0 = P::mfma(fa[ks], fb[ks], a0);
    }
    const double wgt = (bi == bj) ? 1.0 : 2.0;
    const int gj = bj * DB + l15;
    const double alj = s.al[gj];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gi = bi * DB + lq + 4 * r;
      if (gi < N && gj < N) {
        double dq2[DMAX];
        kern.diffs(s.xr, d, NP, gi, gj, dq2);
        const double kvv = kv[i][r];
        const double w = s.al[gi] * alj - (a0[r] + a1[r]);
        const double wk = wgt * w * kvv;
        s_amp += wk;
#pragma unroll
        for (int q = 0; q < DMAX; ++q) s_ell[q] += wk * dq2[q];
        if (gi == gj) s_noise += w;
      }
    }
  }
  {
    double vals[GRAD_N];
    vals[0] = s_amp;
#pragma unroll
    for (int q = 0; q < MAXD; ++q) vals[1 + q] = q < DMAX ? s_ell[q < DMAX ? q : 0] : 0.0;
    vals[9] = s_noise;
    vals[10] = vals[11] = 0.0;
#pragma unroll
    for (int i = 0; i < GRAD_N; ++i) {
      const bool used = i == 0 || i == 9 || (i >= 1 && i <= DMAX);   // the other slots are zeros by construction
      const double v = used ? sm_wave_sum(vals[i]) : 0.0;
      if (lane == 0) s.red[wave * GRAD_N + i] = v;
    }
  }
  __syncthreads();
  if (tid < GRAD_N) {
    double t = 0.0;
    for (int w = 0; w < SM_WAVES; ++w) t += s.red[w * GRAD_N + tid];   // fixed order
    s.sc[24 + tid] = t;
  }
  __syncthreads();
  ck.lap(s.sc, 7, tid);
#ifdef CGP_ABLATION
  if (tid == 0) s.sc[48 + 9] += 1.0;
#endif
}

// Logexp (GPy paramz.transformations.Logexp): theta = log(1 + exp(x))
__device__ __forceinline__ double sm_to_theta(double x) { return x > 35.0 ? x : log1p(exp(x)); }
__device__ __forceinline__ double sm_to_x(double th) { return th > 35.0 ? th : log(expm1(th)); }
// The trial point's Logexp pair with ONE exponential and one logarithm: theta = softplus(x) = max(x, 0) + log1p(e), e = exp(-|x|),
// and dtheta/dx = 1 - exp(-theta) = sigmoid(x) = (x >= 0 ? 1 : e) / (1 + e).  log1p(e) as log(w) e / (w - 1), w = 1 + e (exact
// to rounding where 1 + e rounds; w == 1: e itself).  The library's log1p(exp(x)) and expm1(-theta) were 5.4 k cycles of
// every evaluation on the wave that also runs the L-BFGS step.
__device__ __forceinline__ double sm_softplus(double x, double &slope) {
  if (x > 35.0) {
    slope = 1.0;
    return x;
  }
  ExpC ec;
  ec.load_literals();
  const double e = exp_nonpos(-__builtin_fabs(x), ec), w = 1.0 + e;
  const double l = w == 1.0 ? e : log(w) * (e / (w - 1.0));
  slope = (x >= 0.0 ? 1.0 : e) / w;
  return __builtin_fmax(x, 0.0) + l;
}

// Before an evaluation, one lane per parameter (the transcendental functions of the Logexp transform are a few hundred
// instructions each: in parallel they cost one of them, on lane 0 they cost nth of them): theta from the trial point
// (OPT) or as given (EVAL) -> sc[0..9], dtheta/dx = 1 - exp(-theta) -> sc[36..45]; the jitter ladder starts over.
__device__ __forceinline__ void sm_trial_point(const SmallLds &s, const SmallArgs &p, int nth, const double *thb, int tid) {
  if (tid < MAX_THETA) {
    double th = 0.0, dth = 0.0;
    if (tid < nth) {
      if (p.mode == SM_MODE_OPT) {
        const double x = s.lb->xn[tid];
        th = fmax(sm_softplus(x, dth), 1e-300);
      } else {
        th = thb[tid];
        dth = 1.0;
      }
    }
    s.sc[tid] = th;
    s.sc[36 + tid] = dth;
  }
  if (tid == 0) {
    s.sc[21] = 0.0;
    s.flag[2] = 0;
  }
}

// The per-evaluation constants from theta and the jitter: 1 / ell_q (one lane per dimension), amplitudes, diagonal addend.
// Called by the first wave right after sm_trial_point (same wave: its LDS writes are visible to its later reads in order).
__device__ __forceinline__ void sm_constants(const SmallLds &s, int kid, int d, int nth, int tid) {
  if (tid < MAXD) s.sc[10 + tid] = tid < d ? ((kid == K_SE_ARD) ? 1.0 / s.sc[1 + tid] : 1.0 / s.sc[1]) : 0.0;
  if (tid == MAXD) {
    s.sc[18] = s.sc[0];
    s.sc[19] = (kid == K_RBF_BROWNIAN) ? s.sc[2] : 0.0;
    s.sc[20] = s.sc[nth - 1] + 1e-8 + s.sc[21];
    s.flag[0] = 0;
  }
}

// ---- L-BFGS step on wave 0, one lane per parameter -------------------------------------------------------------------
// lbfgs_core.hpp's state machine (same decisions, same state struct in LDS) with the vectors spread over the lanes:
// lane j holds component j of x, g, the trial point and the direction in registers, dot products are four row-local
// shuffles, and the history pairs are read from LDS once per pair instead of once per component.  On one lane the same
// step was ~24 k cycles of dependent LDS round trips per evaluation; the decisions are wave-uniform (they derive from
// the reduced scalars), so there is no divergence.  Sums run as a tree over the lanes instead of left to right: results
// agree with the host stepper to rounding, not bitwise.
__device__ __forceinline__ double lb_dot(double a, double b, bool on) {
  double p = on ? a * b : 0.0;
  return sm_row_sum(p);
}
__device__ __forceinline__ double lb_amax(double a, bool on) {
  double p = on ? __builtin_fabs(a) : 0.0;
  return sm_row_max(p);
}

__device__ __attribute__((noinline)) void sm_lbfgs_tell_wave(corenav::LbfgsCore &c, double fv, double gv, int lane) {
  using corenav::LB_M;
  const int j = lane & 15, n = c.n;
  const bool on = j < n, wr = lane < 16;
  double xj = c.x[j], gj = c.g[j], xnj = c.xn[j], gnj = c.gn[j], dj = c.dir[j];
  double f = c.f, fn = c.fn, dg0 = c.dg0, t = c.t;
  corenav::MtSearch mt = c.mt;
  int hist = c.hist, ls = c.ls, iters = c.iters, first = c.first;
  const int evals = c.evals + 1, max_evals = c.max_evals;
  const double pgtol = c.pgtol, ftol = c.ftol;
  bool do_start = false;
  int fin = -1;
  const bool feas = __builtin_isfinite(fv);
  if (!feas) fv = corenav::LB_INFEASIBLE;
  if (first) {
    first = 0;
    f = fv;
    gj = gv;
    if (!feas) fin = 3;
    else if (lb_amax(gj, on) <= pgtol) fin = 0;
    else do_start = true;
  } else {
    fn = fv;
    gnj = feas ? gv : 0.0;
    // one call of the More-Thuente search (lbfgs_core.hpp: MtSearch, the same code as the host stepper)
    const double gd = lb_dot(gnj, dj, on);
    double tt = t;
    const int task = mt.step(fn, gd, tt);
    if (task == 0) {
      if (evals >= max_evals) fin = 2;
      else if (ls >= corenav::LB_MAXLS) {   // the search failed: back to x, without memory if there was any
        if (hist == 0) fin = 3;
        else {
          hist = 0;
          do_start = true;
        }
      } else {
        ++ls;
        t = tt;
        xnj = xj + tt * dj;
      }
    } else {   // accept the step
      const double sj = xnj - xj, yj = gnj - gj;
      const double sy = lb_dot(sj, yj, on), fold = f;
      xj = xnj;
      gj = gnj;
      f = fn;
      ++iters;
      if (lb_amax(gj, on) <= pgtol) fin = 0;
      else if ((fold - f) <= ftol * __builtin_fmax(__builtin_fmax(__builtin_fabs(fold), __builtin_fabs(f)), 1.0)) fin = 1;
      else {
        if (sy > corenav::LB_EPS * (-dg0 * t)) {
          if (hist == LB_M) {  // drop the oldest pair
            if (wr) {
#pragma unroll
              for (int i = 1; i < LB_M; ++i) {
                c.S[i - 1][j] = c.S[i][j];
                c.Y[i - 1][j] = c.Y[i][j];
              }
            }
            if (lane == 0) {
#pragma unroll
              for (int i = 1; i < LB_M; ++i) c.rho[i - 1] = c.rho[i];
            }
            --hist;
          }
          if (wr) {
            c.S[hist][j] = sj;
            c.Y[hist][j] = yj;
          }
          if (lane == 0) c.rho[hist] = 1.0 / sy;
          ++hist;
        }
        do_start = true;
      }
    }
  }
  if (do_start) {
    if (evals >= max_evals) fin = 2;
    else {
      double q = gj, al[LB_M], sv[LB_M], yv[LB_M], rh[LB_M];   // two-loop recursion; every pair read from LDS once
#pragma unroll
      for (int i = 0; i < LB_M; ++i) {
        const bool live = i < hist;
        sv[i] = live ? c.S[i][j] : 0.0;
        yv[i] = live ? c.Y[i][j] : 0.0;
        rh[i] = live ? c.rho[i] : 0.0;
      }
#pragma unroll
      for (int i = LB_M - 1; i >= 0; --i) {
        if (i < hist) {
          al[i] = rh[i] * lb_dot(sv[i], q, on);
          q -= al[i] * yv[i];
        } else al[i] = 0.0;
      }
      if (hist > 0) {
        double sl = 0.0, yl = 0.0;
#pragma unroll
        for (int i = 0; i < LB_M; ++i) {
          sl = (i == hist - 1) ? sv[i] : sl;
          yl = (i == hist - 1) ? yv[i] : yl;
        }
        q *= lb_dot(sl, yl, on) / lb_dot(yl, yl, on);
      }
#pragma unroll
      for (int i = 0; i < LB_M; ++i) {
        if (i < hist) {
          const double be = rh[i] * lb_dot(yv[i], q, on);
          q += sv[i] * (al[i] - be);
        }
      }
      dj = -q;
      dg0 = lb_dot(gj, dj, on);
      bool descent = dg0 < 0;
      if (!descent && hist > 0) {  // ascent direction: drop the memory, restart from steepest descent
        hist = 0;
        dj = -gj;
        dg0 = lb_dot(gj, dj, on);
        descent = dg0 < 0;
      }
      if (!descent) fin = 3;
      else {
        const double dnorm = sqrt(lb_dot(dj, dj, on));
        t = iters == 0 ? __builtin_fmin(1.0 / dnorm, corenav::LB_STPMAX) : 1.0;
        ls = 1;
        mt.start(f, dg0, t);
        xnj = xj + t * dj;
      }
    }
  }
  if (fin >= 0) xnj = xj;
  if (wr) {
    c.x[j] = xj;
    c.g[j] = gj;
    c.xn[j] = xnj;
    c.gn[j] = gnj;
    c.dir[j] = dj;
  }
  if (lane == 0) {
    c.f = f; c.fn = fn; c.dg0 = dg0; c.t = t;
    c.mt = mt;
    c.hist = hist; c.ls = ls; c.iters = iters; c.first = first; c.evals = evals;
    if (fin >= 0) {
      c.status = fin;
      c.finished = 1;
    }
  }
}

// Wave 0 after an evaluation, one lane per parameter: the gradient component from the sums (the host twin is
// grad_from_sums, cgp_engine.hip), then either the L-BFGS step (OPT) or the outputs (EVAL).
__device__ __forceinline__ void sm_wave0_tell(const SmallLds &s, const SmallArgs &p, int kid, int d, int nth, double *ob, int lane) {
  const bool ok = s.flag[0] == 0;
  const int j = lane & 15;
  const double *sums = s.sc + 24;
  double num = 0.0, den = 1.0;   // d(-logML)/dtheta_j = -0.5 num / den
  if (j < nth) {
    if (j == nth - 1) num = sums[9];
    else if (j == 0) {
      num = sums[0];
      den = s.sc[0];
    } else if (kid == K_SE_ISO) {
      for (int q = 0; q < d; ++q) num += sums[1 + q];
      den = s.sc[1];
    } else if (kid == K_SE_ARD) {
      num = sums[j];
      den = s.sc[j];
    } else {   // RBF x Brownian: theta = (sigma_r^2, ell, sigma_b^2, sigma_n^2); both amplitudes scale K
      num = j == 1 ? sums[1] : sums[0];
      den = s.sc[j];
    }
  }
  const double g = ok && j < nth ? -0.5 * num / den : 0.0;
  if (p.mode == SM_MODE_OPT) {
    sm_lbfgs_tell_wave(*s.lb, ok ? -s.sc[22] : INFINITY, g * s.sc[36 + j], lane);
    if (lane == 0) s.flag[1] = s.lb->finished ? 1 : 0;
  } else {
    if (lane < MAX_THETA) {
      ob[SMO_GRAD + lane] = g;
      ob[SMO_THETA + lane] = s.sc[lane];
    }
    if (lane == 0) {
      ob[SMO_LOGML] = s.sc[22];
      ob[SMO_EVALS] = 1.0;
      ob[SMO_STATUS] = 0.0;
      ob[SMO_ITERS] = 0.0;
      ob[SMO_INFO] = (double)s.flag[0];
      ob[SMO_JITTER] = ok ? s.sc[21] : 0.0;
      s.flag[1] = 1;
    }
  }
}

// BROWN: RBF x Brownian (d = 1); otherwise SE-iso / SE-ARD with d <= DMAX (the host picks the smallest build that fits)
template <bool BROWN, int DMAX>
__global__ __launch_bounds__(SM_THREADS) void k_small(SmallArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int N = p.N, d = p.d, kid = p.kernel_id, nth = p.nth;
  const int NB = (N + DB - 1) / DB, NP = NB * DB, nblk = NB * (NB + 1) / 2;
  SmallLds s;
  s.Bk = reinterpret_cast<double *>(smem_raw);
  s.xr = s.Bk + (size_t)nblk * SM_BLK;
  s.yv = s.xr + (size_t)d * NP;
  s.zv = s.yv + NP;
  s.al = s.zv + NP;
  s.ldg = s.al + NP;
  s.tmp = s.ldg + NP;
  s.red = s.tmp + 3 * NP;
  s.sc = s.red + SM_WAVES * GRAD_N;
  s.lb = reinterpret_cast<corenav::LbfgsCore *>(s.sc + 64);
  s.flag = reinterpret_cast<int *>(reinterpret_cast<char *>(s.lb) + sizeof(corenav::LbfgsCore));
  s.tb = s.flag + 8;
  s.deal = reinterpret_cast<unsigned short *>(s.tb + nblk);
  s.etab = reinterpret_cast<double *>(smem_raw + ((small_lds_bytes(NB, d) + 15) & ~(size_t)15));   // behind everything else
  s.tab_n = BROWN ? p.tab_n : 0;
  const double *Xb = p.X + (size_t)b * d * N, *yb = p.y + (size_t)b * N;
  double *thb = p.theta + (size_t)b * MAX_THETA, *ob = p.out + (size_t)b * SM_OUT;
  for (int i = tid; i < d * NP; i += SM_THREADS) {
    const int q = i / NP, r = i - q * NP;
    s.xr[i] = r < N ? Xb[(size_t)q * p.x_sq + (size_t)r * p.x_sr] : 0.0;
  }
  for (int i = tid; i < NP; i += SM_THREADS) s.yv[i] = i < N ? yb[i] : 0.0;
  for (int i = tid; i < nblk; i += SM_THREADS) {   // packed index -> (block row, block column)
    int bi = 0, rem = i;
    while (rem > bi) {
      rem -= bi + 1;
      ++bi;
    }
    s.tb[i] = bi | (rem << 8);
  }
  for (int i = tid; i < NB * SM_NH * SM_DEAL; i += SM_THREADS) s.deal[i] = p.deal[i];
  if (tid >= 64 && tid < 128) sm_mean_abs(s.sc + 23, Xb, N, p.x_sr, tid - 64);
  __syncthreads();
  if (tid == 0) {
    s.flag[1] = 0;
    for (int i = 0; i < 16; ++i) s.sc[48 + i] = 0.0;
    if (p.mode == SM_MODE_OPT) {
      double *x0 = s.red;
      for (int i = 0; i < nth; ++i) x0[i] = sm_to_x(thb[i]);
      s.lb->init(x0, nth, p.max_evals, p.pgtol, p.factr);
    }
  }
  __syncthreads();
  const int max_rounds = p.mode == SM_MODE_OPT ? p.max_evals + 64 : 1;
  // the first trial point and its constants by the first wave; every later one right behind the L-BFGS step, on the same
  // wave, without a workgroup barrier in between (LDS accesses of one wave are in order)
  if (tid < 64) {
    sm_trial_point(s, p, nth, thb, tid);
    sm_constants(s, kid, d, nth, tid);
  }
  __syncthreads();
  for (int round = 0; round < max_rounds; ++round) {
    SmClock ck;
    ck.start();
    // GPy jitchol: retry a matrix that is not positive definite with jitter mean(diag) 1e-6 10^k, k = 0..4
    for (;;) {
      sm_eval<BROWN, DMAX>(s, d, N, NB, tid);
      const int bad = s.flag[0];
      const int attempt = s.flag[2];
      // every wave has latched the two flags before anybody rewrites them: wave 0 runs ahead from here (L-BFGS step, next
      // trial point, which reset flag[0] / flag[2]), and a wave still reading would otherwise see the old pivot with the
      // new attempt count on an exhausted ladder and take the retry branch alone (barrier mismatch)
      __syncthreads();
      if (bad == 0 || attempt >= 5) break;
      if (tid == 0) {
        const double noise = s.sc[nth - 1] + 1e-8;
        const double md = BROWN ? s.sc[0] * s.sc[2] * s.sc[23] + noise : s.sc[0] + noise;
        s.sc[21] = attempt == 0 ? md * 1e-6 : s.sc[21] * 10.0;
        s.sc[20] = noise + s.sc[21];   // the diagonal addend is the only constant the jitter changes
        s.flag[2] = attempt + 1;
        s.flag[0] = 0;
      }
      __syncthreads();
      ck.start();
    }
    ck.start();
    if (tid < 64) {
      sm_wave0_tell(s, p, kid, d, nth, ob, tid);
      if (p.mode == SM_MODE_OPT && !s.lb->finished) {
        sm_trial_point(s, p, nth, thb, tid);
        sm_constants(s, kid, d, nth, tid);
      }
    }
    ck.lap(s.sc, 8, tid);
    __syncthreads();
    if (s.flag[1]) break;
  }
#ifdef CGP_ABLATION
  if (tid == 0)
    for (int i = 0; i < 16; ++i) ob[32 + i] = s.sc[48 + i];
#endif
  if (p.mode == SM_MODE_OPT) {
    const corenav::LbfgsCore &lb = *s.lb;
    if (tid < MAX_THETA) ob[SMO_THETA + tid] = thb[tid] = tid < nth ? sm_to_theta(lb.x[tid]) : 0.0;
    if (tid == 0) {
      ob[SMO_LOGML] = -lb.f;
      ob[SMO_EVALS] = (double)lb.evals;
      ob[SMO_STATUS] = (double)(lb.done() ? lb.status : 2);
      ob[SMO_ITERS] = (double)lb.iters;
      ob[SMO_INFO] = __builtin_isfinite(lb.f) ? 0.0 : 1.0;
      ob[SMO_JITTER] = 0.0;
      for (int i = 0; i < MAX_THETA; ++i) ob[SMO_GRAD + i] = i < nth ? lb.g[i] : 0.0;   // wrt x, at the optimum
    }
  }
}

// --------------------------------------------------------------------------------------------------
// k_small_predict: the reference node's fixed-theta work item (gp_slip_node.py:35,45-49,57-61: GPRegression at theta, then
// mean and variance on the 599 ticks after the window) for a short window in ONE launch.  `parts` workgroups per window:
// each repeats the fit in its own LDS (Gram, Cholesky, W = L^-1, alpha, logML, GPy's jitter ladder: ~40 us, cheaper
// than handing a factor from CU to CU) and predicts its slice of the test points, 16 at a time:
//   K* chunk (NB blocks of 16 x 16) -> LDS;  mean = K*^T alpha;  V = W K* block row by block row on MFMA;
//   var = k** - |V|^2 (clipped at 1e-15, + sigma_n^2 on request).
// Part 0 writes the window's record (logML, info, jitter).
// --------------------------------------------------------------------------------------------------
// Operands of one 16 x 16 x 16 fp64 product of k_small_predict's V phase: the A side (a block of W, element [m = l15][k = 4 ks + lq],
// column stride 17) and the B side (a block of K*, element [k][n = l15]).  The V loop requests the NEXT product's operands
// unconditionally before multiplying the current one, in two identical halves, behind a sched_barrier and with carried
// addresses: conditional request blocks make the compiler's wait-count pass wait for ALL outstanding LDS loads before the first
// MFMA of every product, plain requests are sunk behind the MFMAs by the scheduler, and addresses recomputed per product land
// in the other operand set's destination registers (a full wait again).
struct SmOperands {
  double a[4], b[4];
};
__device__ __forceinline__ void sm_req(SmOperands &o, const double *w, const double *k) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    o.a[ks] = w[4 * ks * SM_LD];
    o.b[ks] = k[4 * ks];
  }
}

template <bool BROWN, int DMAX>
__global__ __launch_bounds__(SM_THREADS) void k_small_predict(SmallArgs p) {
  using P = Prec<double>;
  using acc_t = P::acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, b = blockIdx.x / p.parts, part = blockIdx.x % p.parts;
  const int lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = p.N, d = p.d, kid = p.kernel_id, nth = p.nth, M = p.M;
  const int NB = (N + DB - 1) / DB, NP = NB * DB, nblk = NB * (NB + 1) / 2;
  SmallLds s;
  s.Bk = reinterpret_cast<double *>(smem_raw);
  s.xr = s.Bk + (size_t)nblk * SM_BLK;
  s.yv = s.xr + (size_t)d * NP;
  s.zv = s.yv + NP;
  s.al = s.zv + NP;
  s.ldg = s.al + NP;
  s.tmp = s.ldg + NP;
  s.red = s.tmp + 3 * NP;
  s.sc = s.red + SM_WAVES * GRAD_N;
  s.lb = reinterpret_cast<corenav::LbfgsCore *>(s.sc + 64);
  s.flag = reinterpret_cast<int *>(reinterpret_cast<char *>(s.lb) + sizeof(corenav::LbfgsCore));
  s.tb = s.flag + 8;
  s.deal = reinterpret_cast<unsigned short *>(s.tb + nblk);
  // behind k_small's layout (small_lds_bytes): the prediction scratch
  double *Ks = reinterpret_cast<double *>(smem_raw + ((small_lds_bytes(NB, d) + 15) & ~(size_t)15));
  double *xt = Ks + (size_t)NB * SM_BLK;     // [d][16] test points of the chunk
  double *mpart = xt + (size_t)d * DB;       // [NB][16] partial means by block row
  double *vpart = mpart + (size_t)NB * DB;   // [waves][16] partial |V|^2
  s.etab = vpart + SM_WAVES * DB + 16;       // small_predict_lds_extra ends here
  s.tab_n = BROWN ? p.tab_n : 0;
  const double *Xb = p.X + (size_t)b * d * N, *yb = p.y + (size_t)b * N, *Xsb = p.Xs + (size_t)b * d * M;
  SmClock ck;   // CGP_ABLATION builds: slots 10 staging, 11 K* chunk, 12 means + V, 13 outputs (the fit laps slots 1..7 itself)
  ck.start();
  double *thb = p.theta + (size_t)b * MAX_THETA, *ob = p.out + (size_t)b * SM_OUT;
  for (int i = tid; i < d * NP; i += SM_THREADS) {
    const int q = i / NP, r = i - q * NP;
    s.xr[i] = r < N ? Xb[(size_t)q * p.x_sq + (size_t)r * p.x_sr] : 0.0;
  }
  for (int i = tid; i < NP; i += SM_THREADS) s.yv[i] = i < N ? yb[i] : 0.0;
  for (int i = tid; i < nblk; i += SM_THREADS) {
    int bi = 0, rem = i;
    while (rem > bi) {
      rem -= bi + 1;
      ++bi;
    }
    s.tb[i] = bi | (rem << 8);
  }
  for (int i = tid; i < NB * SM_NH * SM_DEAL; i += SM_THREADS) s.deal[i] = p.deal[i];
  const unsigned rowmask = __builtin_amdgcn_readfirstlane((unsigned)p.deal[NB * SM_NH * SM_DEAL + wave]);   // sm_build_rowmap
  if (tid >= 64 && tid < 128) sm_mean_abs(s.sc + 23, Xb, N, p.x_sr, tid - 64);
  __syncthreads();
  if (tid < 16) s.sc[48 + tid] = 0.0;
  if (tid < MAX_THETA) s.sc[tid] = tid < nth ? thb[tid] : 0.0;
  if (tid == 0) {
    s.sc[21] = p.jitter ? p.jitter[b] : 0.0;
    s.flag[2] = 0;
  }
  __syncthreads();
  if (tid < 64) sm_constants(s, kid, d, nth, tid);
  __syncthreads();
  ck.lap(s.sc, 10, tid);
  for (;;) {   // GPy jitchol ladder (every part of a window climbs it identically)
    sm_eval<BROWN, DMAX, false>(s, d, N, NB, tid);
    const int bad = s.flag[0], attempt = s.flag[2];
    if (bad == 0 || attempt >= 5 || !p.ladder) break;
    __syncthreads();
    if (tid == 0) {
      const double noise = s.sc[nth - 1] + 1e-8;
      const double md = BROWN ? s.sc[0] * s.sc[2] * s.sc[23] + noise : s.sc[0] + noise;
      s.sc[21] = attempt == 0 ? md * 1e-6 : s.sc[21] * 10.0;
      s.sc[20] = noise + s.sc[21];
      s.flag[2] = attempt + 1;
      s.flag[0] = 0;
    }
    __syncthreads();
  }
  const bool ok = s.flag[0] == 0;
  if (part == 0 && tid == 0) {
    ob[SMO_LOGML] = s.sc[22];
    ob[SMO_EVALS] = 1.0;
    ob[SMO_STATUS] = 0.0;
    ob[SMO_ITERS] = 0.0;
    ob[SMO_INFO] = (double)s.flag[0];
    ob[SMO_JITTER] = ok ? s.sc[21] : 0.0;
    if (p.logml) p.logml[b] = s.sc[22];
    if (p.info) p.info[b] = s.flag[0];
  }
  if (!ok || M <= 0) return;
  ck.start();
  // ---- this part's slice of the test points, 16 at a time
  ExpC ec;
  ec.load_literals();
  SmKern<BROWN, DMAX> kern;
  kern.load(s.sc);
  const double noise_add = p.include_noise ? s.sc[nth - 1] : 0.0;
  const int nchunk = (M + DB - 1) / DB, per = (nchunk + p.parts - 1) / p.parts;
  double *Bk = s.Bk;
  for (int ch = part * per; ch < min(nchunk, (part + 1) * per); ++ch) {
    const int m0 = ch * DB;
    if (tid < d * DB) {
      const int q = tid / DB, c = tid - q * DB;
      xt[tid] = m0 + c < M ? Xsb[(size_t)q * p.xs_sq + (size_t)(m0 + c) * p.xs_sr] : 0.0;
    }
    __syncthreads();
    {   // K* blocks: two per pass, one entry per thread; entry (training row r of block bj, test column c) at Ks[bj][c * 17 + r]
      const int e = tid & 255, r = e & 15, c = e >> 4;
      for (int b0 = 0; b0 < NB; b0 += 2) {
        const int bj = b0 + (tid >> 8);
        if (bj < NB) {
          const int gi = bj * DB + r;
          double kv;
          if constexpr (BROWN) kv = s.tab_n > 0 ? kern.cross_tab(s.xr, xt, s.etab, (double)(s.tab_n - 1), gi, c) : kern.cross(s.xr, xt, d, NP, gi, c, ec);
          else kv = kern.cross(s.xr, xt, d, NP, gi, c, ec);
          Ks[bj * SM_BLK + c * SM_LD + r] = (gi < N && m0 + c < M) ? kv : 0.0;
        }
      }
    }
    __syncthreads();
    ck.lap(s.sc, 11, tid);
    if (tid >= 256 && tid < 256 + NB * DB) {   // partial means by block row, on the waves with the short V rows (sm_build_rowmap)
      const int bj = (tid - 256) >> 4, c = tid & 15;
      const double *kb = Ks + bj * SM_BLK + c * SM_LD;
      double a = 0.0;
#pragma unroll
      for (int r = 0; r < DB; ++r) a = __builtin_fma(kb[r], s.al[bj * DB + r], a);
      mpart[bj * DB + c] = a;
    }
    double q2 = 0.0;   // lanes of the wave: sum over this wave's block rows of V[row][col l15]^2, rows lq + 4 r
    for (unsigned rows = rowmask; rows;) {
      const int bi = 31 - __builtin_clz(rows);
      rows &= ~(1u << bi);
      acc_t a0 = acc_t{0, 0, 0, 0}, a1 = a0;
      // the next product's operands are requested (always: a repeat of the last one's when there is none) before this one's
      // MFMAs, in two identical halves with the addresses carried along -- see SmOperands
      const double *wp = Bk + sm_tri(bi, 0) + lq * SM_LD + l15, *kp = Ks + l15 * SM_LD + lq;
      SmOperands o0, o1;
      sm_req(o0, wp, kp);
      for (int bj = 0;; bj += 2) {
        const int s1 = bj + 1 <= bi ? SM_BLK : 0;
        wp += s1;
        kp += s1;
        sm_req(o1, wp, kp);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a0 = P::mfma(o0.a[ks], o0.b[ks], a0);
        if (bj + 1 > bi) break;
        const int s2 = bj + 2 <= bi ? SM_BLK : 0;
        wp += s2;
        kp += s2;
        sm_req(o0, wp, kp);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a1 = P::mfma(o1.a[ks], o1.b[ks], a1);
        if (bj + 2 > bi) break;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = a0[r] + a1[r];
        q2 = __builtin_fma(v, v, q2);
      }
    }
    q2 += __shfl_xor(q2, 16);
    q2 += __shfl_xor(q2, 32);
    if (lane < DB) vpart[wave * DB + lane] = q2;
    __syncthreads();
    ck.lap(s.sc, 12, tid);
    if (tid < DB && m0 + tid < M) {
      double mu = 0.0, q = 0.0;
      for (int bj = 0; bj < NB; ++bj) mu += mpart[bj * DB + tid];
      for (int w = 0; w < SM_WAVES; ++w) q += vpart[w * DB + tid];
      const double kss = BROWN ? s.sc[18] * s.sc[19] * fabs(xt[tid]) : s.sc[18];
      double v = kss - q;
      v = v < 1e-15 ? 1e-15 : v;
      p.mean[(size_t)b * M + m0 + tid] = mu;
      p.var[(size_t)b * M + m0 + tid] = v + noise_add;
    }
    __syncthreads();
    ck.lap(s.sc, 13, tid);
  }
#ifdef CGP_ABLATION
  if (part == 0 && tid == 0)
    for (int i = 0; i < 16; ++i) ob[32 + i] = s.sc[48 + i];
#endif
  if (p.done_flag) {
    __syncthreads();   // every thread's stores of this workgroup have been issued and waited for
    if (tid == 0) {
      __threadfence_system();
      if (atomicAdd(p.done_count, 1) == (int)gridDim.x - 1) {
        *p.done_count = 0;
        __threadfence_system();
        *reinterpret_cast<volatile int *>(p.done_flag) = p.done_seq;
      }
    }
  }
}

}  // namespace cgp
