// cgp_kernels.hpp -- hand-written gfx950 (CDNA4) kernels of the slip-GP fit/predict.
//
// One fixed-theta fit + predict is ONE blocked left-looking Cholesky of the augmented trapezoid
//
//        [ Ky  ]  N_pad rows   (Ky = k(X,X) + (sigma_n^2 + 1e-8 + jitter) I, never materialised)
//        [ K*^T]  M rows       (cross-covariances of the test points)
//        [ y^T ]  1 row
//
// whose factor panel Lw holds  L  (rows < N_pad),  V^T = (L^-1 K*)^T  and  z^T = (L^-1 y)^T.
// Then  mean = V^T z,  var = k** - |V_m|^2,  logML = -0.5 z'z - sum log L_ii - N/2 log 2pi.
// This replaces GPy's kern.K / jitchol(dpotrf) / dpotrs / dpotri / predict chain that
// gp_slip_node.py:31-49 reaches with three kernels per 128-column block step:
//   k_update : S(i,k) = Gram(i,k) - sum_{j<k} L(i,j) L(k,j)^T     fp64/fp32 MFMA 16x16x4, LDS-tiled
//   k_potf2  : S(k,k) = L(k,k) L(k,k)^T  and  W_k = L(k,k)^-1        one workgroup per fit, LDS-resident
//   k_trmm   : L(i,k) = S(i,k) W_k^T                                   triangular MFMA product
// and k_finalize for mean / variance / log marginal likelihood, k_alpha for alpha = L^-T z.
//
// Storage: Lw is column-major, leading dimension ld (multiple of 128), one slab per fit.
// Inputs are SoA per fit: X[d][N], Xs[d][M], y[N].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace cgp {

constexpr int TS = 128;    // tile edge (rows and columns)
constexpr int KT = 16;     // k-chunk staged through LDS per barrier
constexpr int LDST = 144;  // LDS row stride of a staged chunk (elements): (144*8) % 256 == 128 and
                           // (144*4) % 128 == 64, so the 4 k-groups of an MFMA operand read hit
                           // disjoint bank halves for both fp64 (ds_read_b64) and fp32 (ds_read_b32)
constexpr int DB = 16;     // diagonal sub-block of potf2 / trsm
constexpr int MAXD = 8;
constexpr int MAX_THETA = MAXD + 2;
constexpr int PREP_N = 16;  // prep[0..7] 1/ell_q, [8] log amp (SE) or 0, [9] amp, [10] amp_b, [11] diag add
constexpr int GRAD_N = 12;  // k_grad sums: [0] amplitude, [1..8] length-scales, [9] noise
constexpr int LDP = TS + 2;   // LDS leading dimension of the potf2 tile (2-way conflicts at most)

enum { K_SE_ISO = 0, K_SE_ARD = 1, K_RBF_BROWNIAN = 2 };

struct FitArgs {
  void *Lw;              // [batch][NT*128 cols][ld rows]
  size_t lw_stride;      // elements per fit
  int ld;
  const void *X;         // [batch][d][N]
  const void *Xs;        // [batch][d][M]
  const void *y;         // [batch][N]
  const double *theta;   // [batch][MAX_THETA]
  const double *jitter;  // [batch] or nullptr
  const double *prep;    // [batch][PREP_N] per-fit derived constants written by k_prep
  void *Winv;            // [batch][NTmax][128*128]  W_k = L(k,k)^-1, column-major, lower triangular
  size_t winv_stride;    // elements per fit
  int *info;             // [batch]
  void *mean, *var;      // [batch][M]
  double *logml;         // [batch]
  void *alpha;           // [batch][NT*128]
  size_t alpha_stride;
  int N, d, M, NT, ET, kernel_id, include_noise;
  int rows_from_extra;   // 1: only the extra (test/y) row tiles are processed (predict after fit)
  double *macc, *vacc;   // [batch][M] running sums V_m . z and |V_m|^2 over the block columns already
                         // final (fp64 throughput schedule: accumulated inside k_panel); null = k_finalize
                         // reads the whole of V
  int tile_off;          // first block-tile index of this launch (split panel launches)
  int xid;               // 1: the M (= N) "test rows" are the identity, so the extra block becomes (L^-1)^T (gradient mode)
  double *gpart;         // [batch][pairs][GRAD_N] per-tile-pair partial sums of k_grad
  long long *dbgbuf;     // 64 slots of s_memtime stamps / per-phase cycle sums (-DCGP_ABLATION builds)
  int dbg;               // timing ablations (env CGP_DBG); only a -DCGP_ABLATION build reads it
};

// Timing ablations skip parts of the arithmetic (results are WRONG by design), so the shipped library
// does not contain them: CGP_DBG_ON(p, bit) is a compile-time 0 unless built with -DCGP_ABLATION.
#ifdef CGP_ABLATION
#define CGP_DBG_ON(p, bit) (((p).dbg & (bit)) != 0)
#else
#define CGP_DBG_ON(p, bit) false
#endif
constexpr int DBG_SLOTS = 512;  // cgp_debug_read: [0, 8) potf2 phases, [64 + 8 k, 64 + 8 k + 8) k_panel phases of step k

// Per-phase cycle sums of a kernel (CGP_DBG & 1024 in a -DCGP_ABLATION build): thread 0 of every
// workgroup adds the s_memtime ticks since the previous lap to dbgbuf[slot]; slot + 7 of a step counts
// the workgroups.  Compiles to nothing in the shipped library.
struct PhaseClock {
#ifdef CGP_ABLATION
  long long t;
  bool on;
  __device__ __forceinline__ void start(const FitArgs &p, int tid) {
    on = CGP_DBG_ON(p, 1024) && tid == 0;
    t = __builtin_amdgcn_s_memtime();
  }
  __device__ __forceinline__ void lap(const FitArgs &p, int slot) {
    const long long n = __builtin_amdgcn_s_memtime();
    if (on) atomicAdd(reinterpret_cast<unsigned long long *>(p.dbgbuf) + slot, (unsigned long long)(n - t));
    t = n;
  }
  __device__ __forceinline__ void count(const FitArgs &p, int slot) {
    if (on) atomicAdd(reinterpret_cast<unsigned long long *>(p.dbgbuf) + slot, 1ull);
  }
#else
  __device__ __forceinline__ void start(const FitArgs &, int) {}
  __device__ __forceinline__ void lap(const FitArgs &, int) {}
  __device__ __forceinline__ void count(const FitArgs &, int) {}
#endif
};

template <typename T> struct Prec;
template <> struct Prec<double> {
  using acc_t = double __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: n = lane & 15, m = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int drow(int lane, int reg) { return (lane >> 4) + 4 * reg; }
  static __device__ __forceinline__ double exp_(double x) { return exp(x); }
  static __device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
  static __device__ __forceinline__ double log_(double x) { return log(x); }
  // 1/sqrt(x), x > 0 and normal: hardware seed + two Newton steps (shorter dependent chain than
  // sqrt followed by a division; this sits on the serial critical path of the diagonal blocks)
  static __device__ __forceinline__ double rsqrt_(double x) {
    double r = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    r = __builtin_fma(r, __builtin_fma(-h * r, r, 0.5), r);
    r = __builtin_fma(r, __builtin_fma(-h * r, r, 0.5), r);
    return r;
  }
};
template <> struct Prec<float> {
  using acc_t = float __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f32_16x16x4_f32: n = lane & 15, m = (lane >> 4) * 4 + reg
  static __device__ __forceinline__ int drow(int lane, int reg) { return (lane >> 4) * 4 + reg; }
  static __device__ __forceinline__ float exp_(float x) { return expf(x); }
  static __device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }
  static __device__ __forceinline__ float log_(float x) { return logf(x); }
  static __device__ __forceinline__ float rsqrt_(float x) {
    float r = __builtin_amdgcn_rsqf(x);
    return __builtin_fmaf(r, __builtin_fmaf(-0.5f * x * r, r, 0.5f), r);
  }
};

// Row-tile index (in units of 128 rows of Lw) handled by block t of a step-k launch.
// In-matrix tiles first (first_in .. NT-1), then the ET extra tiles, which start at row NT*128.
__device__ __forceinline__ int row_tile_of(int t, int first_in, int NT, int rows_from_extra) {
  if (rows_from_extra) return NT + t;
  const int nin = NT - first_in;
  return (t < nin) ? first_in + t : NT + (t - nin);
}

// XCD-aware block -> (tile slot t, fit b) map.  The dispatcher is observed to place consecutive
// workgroups on consecutive XCDs (8 private L2s); all tiles of one fit share the tile-k column
// panel, so a fit's tiles are steered to one XCD to make those re-reads L2 hits.  Speed only: any
// placement is correct.  Needs gridDim.y % 8 == 0, otherwise the identity map is used.
__device__ __forceinline__ void tile_fit_of_block(int &t, int &b) {
  const int T = gridDim.x, B = gridDim.y;
  t = blockIdx.x;
  b = blockIdx.y;
  if ((B & 7) == 0) {
    const int lin = blockIdx.y * T + blockIdx.x;
    const int xcd = lin & 7, slot = lin >> 3;
    t = slot % T;
    b = (slot / T) * 8 + xcd;
  }
}

// Same, for launches whose tile slot 0 is a long-running workgroup (k_panel with the next diagonal
// tile fused in): the slot-0 workgroup of every fit gets the lowest linear ids, i.e. is dispatched
// first, and the other slots follow in the XCD-steered order.
__device__ __forceinline__ void tile_fit_of_block_first(int &t, int &b) {
  const int T = gridDim.x, B = gridDim.y;
  const int lin = blockIdx.y * T + blockIdx.x;
  if (lin < B) {
    t = 0;
    b = lin;
    return;
  }
  const int l2 = lin - B;
  if ((B & 7) == 0) {
    const int xcd = l2 & 7, slot = l2 >> 3;
    t = 1 + slot % (T - 1);
    b = (slot / (T - 1)) * 8 + xcd;
  } else {
    t = 1 + l2 % (T - 1);
    b = l2 / (T - 1);
  }
}

// readlane for scalars of either precision (lane index must be wave-uniform)
__device__ __forceinline__ float rdlane(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ double rdlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

#ifdef CGP_AB  // first-generation three-launch schedule (k_update -> k_potf2 -> k_trmm), A/B builds only
// --------------------------------------------------------------------------------------------------
// The MFMA inner loop shared by k_update and k_trmm:
//   acc[i][j] (+)= sum_q  Cop[cl][q] * Rop[rl][q]      over nchunk chunks of KT columns
// Rop = "row panel" (128 rows x K), Cop = "column panel" (128 rows x K), both column-major in global
// memory (K runs along columns, leading dimensions ldR / ldC).  256 threads = 4 waves as 2x2, each
// wave owns a 64x64 block of the 128x128 result as 4x4 MFMA 16x16x4 accumulators:
//   acc[i][j][reg] = C[row = wr*64 + j*16 + (lane&15)][col = wc*64 + i*16 + drow(lane,reg)]
// (MFMA "A" operand = Cop, "B" operand = Rop, so lane&15 runs along result rows, which are
// contiguous in the column-major destination).  Chunks are staged global -> registers -> LDS with
// one chunk of prefetch; LDS rows are padded to LDST so the four k-groups of an operand read fall in
// disjoint bank halves.  TRI: Cop is lower-triangular in (cl, q) (a 128x128 inverse factor), so
// 16-column fragments whose every entry has q > cl are skipped (half the MFMAs).
// --------------------------------------------------------------------------------------------------
template <typename T, bool TRI>
__device__ __forceinline__ void mfma_panel_loop(typename Prec<T>::acc_t (&acc)[4][4], const T *gR, size_t ldR,
                                                const T *gC, size_t ldC, int nchunk, T *smem, int tid, int dbg = 0) {
  using P = Prec<T>;
  using vec8 = T __attribute__((ext_vector_type(8)));
  constexpr int CH = KT * LDST;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  auto compute = [&](const T *cur, int c) {
#pragma unroll
    for (int ks = 0; ks < KT / 4; ++ks) {
      T fa[4], fb[4];
      const T *ra = cur + CH + (ks * 4 + lq) * LDST + wc * 64 + l15;  // column panel -> result columns
      const T *rb = cur + (ks * 4 + lq) * LDST + wr * 64 + l15;       // row panel    -> result rows
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = ra[i * 16];
        fb[i] = rb[i * 16];
      }
      const int q0 = c * KT + ks * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (TRI && q0 > wc * 64 + i * 16 + 15) continue;  // wave-uniform: whole fragment is zero
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = P::mfma(fa[i], fb[j], acc[i][j]);
      }
    }
  };

  if constexpr (sizeof(T) == 8) {
    // fp64: global -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no ds_write).  One
    // wave-instruction moves one 1 KiB column (128 rows): lane -> rows 2*lane, 2*lane+1; the LDS
    // destination is wave-uniform base + lane*16 B, which the column-padded image satisfies.
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    auto stage = [&](T *buf, int chunk) {
#pragma unroll
      for (int i = 0; i < KT / 4; ++i) {
        const int col = wave * (KT / 4) + i;
        __builtin_amdgcn_global_load_lds((gbl_void *)(gR + (size_t)(chunk * KT + col) * ldR + lane * 2),
                                         (lds_void *)(buf + col * LDST), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void *)(gC + (size_t)(chunk * KT + col) * ldC + lane * 2),
                                         (lds_void *)(buf + CH + col * LDST), 16, 0, 0);
      }
    };
    if (nchunk > 0) stage(smem, 0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
      if (c + 1 < nchunk && !(dbg & 1)) stage(smem + ((c + 1) & 1) * 2 * CH, c + 1);
      compute(smem + (c & 1) * 2 * CH, c);
      if (!(dbg & 4)) __syncthreads();
    }
  } else {
    // fp32: register-staged (a 512-byte column does not fill a 1 KiB LDS-DMA wave-instruction)
    const int sc = tid >> 4, sr = (tid & 15) * 8;  // staging: column sc of the chunk, rows sr..sr+7
    gR += sr;
    gC += sr;
    vec8 pr, pc;
    if (nchunk > 0) {
      pr = *reinterpret_cast<const vec8 *>(gR + (size_t)sc * ldR);
      pc = *reinterpret_cast<const vec8 *>(gC + (size_t)sc * ldC);
      *reinterpret_cast<vec8 *>(smem + sc * LDST + sr) = pr;
      *reinterpret_cast<vec8 *>(smem + CH + sc * LDST + sr) = pc;
    }
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
      if (c + 1 < nchunk) {
        pr = *reinterpret_cast<const vec8 *>(gR + (size_t)((c + 1) * KT + sc) * ldR);
        pc = *reinterpret_cast<const vec8 *>(gC + (size_t)((c + 1) * KT + sc) * ldC);
      }
      compute(smem + (c & 1) * 2 * CH, c);
      if (c + 1 < nchunk) {
        T *nxt = smem + ((c + 1) & 1) * 2 * CH;
        *reinterpret_cast<vec8 *>(nxt + sc * LDST + sr) = pr;
        *reinterpret_cast<vec8 *>(nxt + CH + sc * LDST + sr) = pc;
      }
      __syncthreads();
    }
  }
}

#endif  // CGP_AB

// exp(x) for x <= 0 (every covariance exponent is -0.5 r^2): n = rint(x log2 e), r = x - n ln2 in
// two pieces, degree-13 Horner polynomial on |r| <= ln2/2, v_ldexp for 2^n (denormal-exact).  The
// argument is clamped at -800 (result 0) instead of being special-cased.  Coefficients live in
// constant memory so they are fetched once into SGPRs and used as v_fma source operands; as
// immediates every v_fmac would need two v_mov to materialise its addend.
__constant__ double kExpC[16] = {1.6059043836821613e-10, 2.08767569878681e-09,  2.505210838544172e-08,
                                 2.755731922398589e-07,  2.7557319223985893e-06, 2.48015873015873e-05,
                                 1.984126984126984e-04,  1.3888888888888889e-03, 8.333333333333333e-03,
                                 4.1666666666666664e-02, 1.6666666666666666e-01, 0.5,
                                 1.4426950408889634,     -6.93147180369123816490e-01, -1.90821492927058770002e-10,
                                 -800.0};
struct ExpC {
  double c[16];
  __device__ __forceinline__ void load() {
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = kExpC[i];
  }
};
__device__ __forceinline__ double exp_nonpos(double x, const ExpC &e) {
  x = __builtin_fmax(x, e.c[15]);
  const double n = __builtin_rint(x * e.c[12]);
  double r = __builtin_fma(n, e.c[13], x);
  r = __builtin_fma(n, e.c[14], r);
  double q = e.c[0];
#pragma unroll
  for (int i = 1; i < 12; ++i) q = __builtin_fma(q, r, e.c[i]);
  q = __builtin_fma(q, r, 1.0);
  q = __builtin_fma(q, r, 1.0);
  return __builtin_amdgcn_ldexp(q, (int)n);
}
__device__ __forceinline__ float exp_nonpos(float x, const ExpC &) { return __expf(fmaxf(x, -104.f)); }

#ifdef CGP_AB
// Gram tile G(rt, k) evaluated from the inputs, then S = G - acc, stored to the factor panel.
// Column points (16 per lane) are the outer static loops, the 4 row points the inner one.
// FAST: interior tile -- every row and column is a real point and no diagonal / y-row entry is in
// it, so the padding and noise selects vanish; the other tiles take the general path.
template <typename T, bool BROWN, bool FAST>
__device__ __forceinline__ void gram_tile(const FitArgs &p, typename Prec<T>::acc_t (&acc)[4][4], T *__restrict__ out,
                                          const T *__restrict__ xr, const T *__restrict__ xc, const T *__restrict__ yc,
                                          bool extra, int rowbase, int colbase, T amp, T inv_ell, T amp_b, T diag_add,
                                          int lane, int wr, int wc) {
  using P = Prec<T>;
  const int N = p.N, M = p.M, ld = p.ld, l15 = lane & 15;
  ExpC ec;
  ec.load();
  T xrow[4][MAXD];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int q = 0; q < MAXD; ++q) xrow[j][q] = (BROWN && q > 0) ? T(0) : xr[(wr * 64 + j * 16 + l15) * MAXD + q];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cl = wc * 64 + i * 16 + P::drow(lane, r);  // local result column
      const int gcol = colbase + cl;
      T xcol[MAXD];
#pragma unroll
      for (int q = 0; q < MAXD; ++q) xcol[q] = (BROWN && q > 0) ? T(0) : xc[cl * MAXD + q];
      const T ycl = FAST ? T(0) : yc[cl];
      const bool colok = gcol < N;
      T *__restrict__ ocol = out + (size_t)cl * ld + wr * 64 + l15;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int grow = rowbase + wr * 64 + j * 16 + l15;
        T g;
        if (!BROWN) {
          T d2 = 0;
#pragma unroll
          for (int q = 0; q < MAXD; ++q) {
            const T df = xrow[j][q] - xcol[q];
            d2 = __builtin_fma(df, df, d2);
          }
          g = CGP_DBG_ON(p, 32) ? d2 : amp * exp_nonpos(T(-0.5) * d2, ec);
        } else {
          const T x = xrow[j][0], xp = xcol[0];
          const bool same = !FAST && !extra && grow == gcol;  // GPy forces r^2 = 0 on the auto-covariance diagonal
          T r2 = same ? T(0) : (T(-2) * x * xp + (x * x + xp * xp));
          r2 = r2 < T(0) ? T(0) : r2;
          const T rr = P::sqrt_(r2) * inv_ell;
          const int sx = (x > T(0)) - (x < T(0)), sp = (xp > T(0)) - (xp < T(0));
          const T ax = x < T(0) ? -x : x, ap = xp < T(0) ? -xp : xp;
          const T kb = (sx == sp) ? amp_b * (ax < ap ? ax : ap) : T(0);
          g = amp * exp_nonpos(T(-0.5) * rr * rr, ec) * kb;
        }
        if (!FAST) {
          if (!extra) {
            const bool dg = grow == gcol;
            g = dg ? g + diag_add : g;
            g = (grow < N && colok) ? g : (dg ? T(1) : T(0));  // identity padding keeps the factor well defined
          } else {
            g = (grow == M) ? ycl : g;
            g = (!colok || grow > M) ? T(0) : g;
          }
        }
        if (!CGP_DBG_ON(p, 16) || g == T(12345)) ocol[j * 16] = g - acc[i][j][r];
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the 16 column points from being software-pipelined into spills
    }
  }
}

template <typename T, bool BROWN>
__device__ __forceinline__ void gram_epilogue(const FitArgs &p, typename Prec<T>::acc_t (&acc)[4][4],
                                              T *__restrict__ Lw, T *__restrict__ smem, int b, int k, int rt, int tid) {
  const double *__restrict__ th = p.theta + (size_t)b * MAX_THETA;
  const int kid = p.kernel_id, d = p.d, N = p.N, M = p.M, ld = p.ld;
  const bool extra = rt >= p.NT;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  T *xr = smem;                    // [128][MAXD] rows of this tile (training or test points), zero padded
  T *xc = smem + TS * MAXD;        // [128][MAXD] columns = training points of tile k
  T *yc = smem + 2 * TS * MAXD;    // [128] y of the tile-k columns (only the y row uses it)
  const T *__restrict__ Xb = reinterpret_cast<const T *>(p.X) + (size_t)b * d * N;
  const T *__restrict__ Xsb = reinterpret_cast<const T *>(p.Xs) + (size_t)b * d * M;
  const T *__restrict__ yb = reinterpret_cast<const T *>(p.y) + (size_t)b * N;
  for (int idx = tid; idx < MAXD * TS; idx += 256) {
    const int q = idx >> 7, r = idx & 127;
    T vc = T(0), vr = T(0);
    if (q < d) {
      T sc_q = T(1);
      if (kid == K_SE_ISO) sc_q = T(1.0 / th[1]);
      else if (kid == K_SE_ARD) sc_q = T(1.0 / th[1 + q]);
      const int gc = k * TS + r;
      if (gc < N) vc = Xb[(size_t)q * N + gc] * sc_q;
      if (!extra) {
        const int gr = rt * TS + r;
        if (gr < N) vr = Xb[(size_t)q * N + gr] * sc_q;
      } else {
        const int e = (rt - p.NT) * TS + r;
        if (e < M) vr = Xsb[(size_t)q * M + e] * sc_q;
      }
    }
    xc[r * MAXD + q] = vc;
    xr[r * MAXD + q] = vr;
  }
  if (tid < TS) {
    const int gc = k * TS + tid;
    yc[tid] = (gc < N) ? yb[gc] : T(0);
  }
  __syncthreads();
  const T amp = T(th[0]);
  const T inv_ell = BROWN ? T(1.0 / th[1]) : T(1);
  const T amp_b = BROWN ? T(th[2]) : T(0);
  const int nth = (kid == K_SE_ISO) ? 3 : (kid == K_SE_ARD ? d + 2 : 4);
  const T diag_add = T(th[nth - 1] + 1e-8 + (p.jitter ? p.jitter[b] : 0.0));
  const int rowbase = extra ? (rt - p.NT) * TS : rt * TS;  // global row index of local row 0
  const int colbase = k * TS;
  T *__restrict__ out = Lw + (size_t)rt * TS + (size_t)colbase * ld;
  const bool cols_full = colbase + TS <= N;
  const bool fast = cols_full && (extra ? (rowbase + TS <= M) : (rt != k && rowbase + TS <= N));
  if (fast) gram_tile<T, BROWN, true>(p, acc, out, xr, xc, yc, extra, rowbase, colbase, amp, inv_ell, amp_b, diag_add, lane, wr, wc);
  else gram_tile<T, BROWN, false>(p, acc, out, xr, xc, yc, extra, rowbase, colbase, amp, inv_ell, amp_b, diag_add, lane, wr, wc);
}

// --------------------------------------------------------------------------------------------------
// k_update: S(rt, k) = Gram(rt, k) - sum_{j < k} L(rt, j) L(k, j)^T   (a2 gram + a3 syrk/gemm + a8)
// grid (row tiles, batch).  The Gram tile is evaluated from the inputs in the epilogue, so Ky and
// K* never exist in HBM.
// --------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void k_update(FitArgs p, int k) {
  using P = Prec<T>;
  using acc_t = typename P::acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);

  int bt, b;
  tile_fit_of_block(bt, b);
  const int rt = row_tile_of(bt, k, p.NT, p.rows_from_extra);
  T *Lw = reinterpret_cast<T *>(p.Lw) + (size_t)b * p.lw_stride;
  const int ld = p.ld;
  const int tid = threadIdx.x;

  acc_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

  mfma_panel_loop<T, false>(acc, Lw + (size_t)rt * TS, (size_t)ld, Lw + (size_t)k * TS, (size_t)ld,
                            (k * TS) / KT, smem, tid, p.dbg);
  if (CGP_DBG_ON(p, 8)) return;

  // ---- epilogue: Gram tile from the inputs, S = G - acc ----
  if (p.kernel_id == K_RBF_BROWNIAN) gram_epilogue<T, true>(p, acc, Lw, smem, b, k, rt, tid);
  else gram_epilogue<T, false>(p, acc, Lw, smem, b, k, rt, tid);
}

// --------------------------------------------------------------------------------------------------
// k_trmm: L(rt, k) = S(rt, k) W_k^T with W_k = L(k,k)^-1 from k_potf2 (a3 "trsm_panel" and, for the
// extra tiles, a8 "trsm_var", done as a triangular MFMA product instead of a substitution).
// grid (row tiles below k, batch).  In place: a workgroup reads only its own tile before writing it.
// --------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void k_trmm(FitArgs p, int k) {
  using P = Prec<T>;
  using acc_t = typename P::acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  int bt, b;
  tile_fit_of_block(bt, b);
  const int rt = row_tile_of(bt, k + 1, p.NT, p.rows_from_extra);
  T *Lw = reinterpret_cast<T *>(p.Lw) + (size_t)b * p.lw_stride;
  const int ld = p.ld;
  const T *Wk = reinterpret_cast<const T *>(p.Winv) + (size_t)b * p.winv_stride + (size_t)k * TS * TS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15;
  acc_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
  T *tile = Lw + (size_t)(k * TS) * ld + (size_t)rt * TS;
  mfma_panel_loop<T, true>(acc, tile, (size_t)ld, Wk, (size_t)TS, TS / KT, smem, tid);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rl = wr * 64 + j * 16 + l15;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cl = wc * 64 + i * 16 + P::drow(lane, r);
        tile[(size_t)cl * ld + rl] = acc[i][j][r];
      }
  }
}

#endif  // CGP_AB

// One 16x16 diagonal block in the registers of a wavefront: lane holds row (lane & 15) of the block
// in a[] (replicated over the four 16-lane groups).  On return a[] holds the row of the Cholesky
// factor and w[i] = Dinv[i][lane & 15] (column (lane & 15) of the block's inverse, by forward
// substitution).  A non-positive pivot is replaced by 1 and reported in `bad` (1-based global index).
// This is the serial critical path of the factorisation (128 dependent pivots per tile) and it is
// bound by instruction issue, not latency (tools/potf2_block_bench.hip), so the fp64 form uses the
// 64-bit DPP row_newbcast of gfx90a+: "times lane c's value" is ONE v_fmac_f64_dpp instead of two
// v_readlane_b32 and an fma (the four rows hold identical copies, a row-local broadcast is the right
// one).  3.76 -> 2.42 us per block with two such waves per SIMD.  A DPP read of a VGPR needs two wait
// states after the VALU write: the first DPP instruction after each producer carries an s_nop 1.
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
template <int C, bool NOP> __device__ __forceinline__ void fmac_bcast(float &acc, float src, float own) {
  if constexpr (NOP) asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(C));
  else asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(C));
}
template <int C> __device__ __forceinline__ float mov_bcast(float src) {
  float d;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(src), "n"(C));
  return d;
}
template <int C, bool NOP> __device__ __forceinline__ void fmac_bcast(double &acc, double src, double own) {  // acc += src[lane C of the row] * own
  if constexpr (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(C));
  else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(C));
}
template <int C> __device__ __forceinline__ double mov_bcast(double src) {
  double d;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(src), "n"(C));
  return d;
}
// 1/sqrt(x): hardware seed (~2^-23 relative) and one third-order step, error ~ e^3
__device__ __forceinline__ double rsqrt3(double x) {
  const double r = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(-(x * r), r, 1.0);
  const double q = __builtin_fma(0.375, e, 0.5) * e;
  return __builtin_fma(r, q, r);
}

template <typename T>
__device__ __forceinline__ void factor_block16(T (&a)[DB], T (&w)[DB], int &bad, int pivot_base, int l15) {
  using P = Prec<T>;
  T rinv[DB];
  if constexpr (true) {
    static_for<0, DB>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      T dj = mov_bcast<J>(a[J]);
      const bool ok = dj > T(0);
      if (!ok && bad == 0) bad = pivot_base + J + 1;
      dj = ok ? dj : T(1);
      T rs;
      if constexpr (sizeof(T) == 8) rs = rsqrt3(dj);
      else rs = P::rsqrt_(dj);
      rinv[J] = rs;
      const T l = (ok ? a[J] : ((l15 == J) ? T(1) : a[J])) * rs;   // lane J's a[J] is the pivot itself
      a[J] = l;
      const T nl = -l;
      static_for<J + 1, DB>([&](auto cc) {
        constexpr int C = decltype(cc)::value;
        fmac_bcast<C, C == J + 1>(a[C], l, nl);
      });
    });
    // right-looking forward substitution for column l15 of the inverse: independent updates per step
    T t[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i) t[i] = (i == l15) ? T(1) : T(0);
    static_for<0, DB>([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      w[Q] = t[Q] * rinv[Q];
      const T nw = -w[Q];
      static_for<Q + 1, DB>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        fmac_bcast<I, false>(t[I], a[Q], nw);
      });
    });
#pragma unroll
    for (int i = 0; i < DB; ++i) w[i] = (i < l15) ? T(0) : w[i];
  } else {
#pragma unroll
    for (int j = 0; j < DB; ++j) {
      T dj = rdlane(a[j], j);
      if (!(dj > T(0))) {
        if (bad == 0) bad = pivot_base + j + 1;
        dj = T(1);
      }
      const T rs = P::rsqrt_(dj);
      rinv[j] = rs;
      const T l = (l15 == j) ? dj * rs : a[j] * rs;
      a[j] = l;
#pragma unroll
      for (int c = j + 1; c < DB; ++c) a[c] -= l * rdlane(l, c);
    }
#pragma unroll
    for (int i = 0; i < DB; ++i) {
      T s = 0;
#pragma unroll
      for (int q = 0; q < DB; ++q)
        if (q < i) s += rdlane(a[q], i) * w[q];
      w[i] = (i < l15) ? T(0) : ((i == l15) ? rinv[i] : -s * rinv[i]);
    }
  }
}

// --------------------------------------------------------------------------------------------------
// k_potf2: factor the 128x128 diagonal tile (a3 "potf2_diag") and invert the factor, one workgroup
// per fit, everything resident in LDS:
//   for each 16-column panel:  (a) wave 0 factors the 16x16 diagonal block and inverts it in
//       registers (lane = row, v_readlane broadcasts, no barriers);
//       (b) panel rows below  P = A Dinv^T          -- MFMA 16x16x4, one 16-row block per wave-slot
//       (c) trailing update   C -= P P^T            -- MFMA, lower block pairs round-robin over waves
//   then (d) W = L^-1 by block levels (MFMA), kept transposed in the unused upper triangle.
// LDS tile is column-major with leading dimension LDP.  info = first non-positive pivot (1-based).
// --------------------------------------------------------------------------------------------------
// Phases (a)-(d) on an LDS-resident tile: factor it in place (lower triangle), leave W = L^-1
// transposed in the strict upper triangle and the inverted 16x16 diagonal blocks in Dv.
template <typename T>
__device__ __forceinline__ void potf2_lds_body(T *At, T *Dv, T *Ts, int *flag, int k, int tid, long long *dbgbuf = nullptr,
                                               int live = TS) {
  using P = Prec<T>;
  using acc_t = typename P::acc_t;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  long long tA = 0, tB = 0, tC = 0, t0 = __builtin_amdgcn_s_memtime();
  for (int jb = 0; jb < TS / DB; ++jb) {
    const int j0 = jb * DB;
    long long s0 = __builtin_amdgcn_s_memtime();
    if (wave == 0) {
      // (a) lane holds row (lane & 15) of the diagonal block (replicated over the four 16-lane groups)
      T a[DB], w[DB];
      int bad = 0;
      if (j0 < live) {
#pragma unroll
        for (int c = 0; c < DB; ++c) a[c] = At[(j0 + c) * LDP + j0 + l15];
        factor_block16<T>(a, w, bad, k * TS + j0, l15);
      } else {  // identity padding beyond the window (N not a multiple of 128): nothing to factor
#pragma unroll
        for (int c = 0; c < DB; ++c) a[c] = w[c] = (c == l15) ? T(1) : T(0);
      }
      if (bad != 0 && lane == 0 && *flag == 0) *flag = bad;
      if (lane < DB) {
#pragma unroll
        for (int c = 0; c < DB; ++c)
          if (l15 >= c) At[(j0 + c) * LDP + j0 + l15] = a[c];
      }
      if (lane < DB) {
#pragma unroll
        for (int i = 0; i < DB; ++i) Dv[jb * DB * DB + l15 * DB + i] = w[i];
      }
    }
    __syncthreads();
    long long s1 = __builtin_amdgcn_s_memtime();
    tA += s1 - s0;
    // (b) panel: rows of block bi below the diagonal block, P[r][c] = sum_q A[r][q] Dinv[c][q]
    for (int bi = jb + 1 + wave; bi < TS / DB; bi += 4) {
      acc_t acc = acc_t{0, 0, 0, 0};
      T fa[4], fb[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        fa[ks] = Dv[jb * DB * DB + (ks * 4 + lq) * DB + l15];
        fb[ks] = At[(j0 + ks * 4 + lq) * LDP + bi * DB + l15];
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = P::mfma(fa[ks], fb[ks], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) At[(j0 + P::drow(lane, r)) * LDP + bi * DB + l15] = acc[r];
    }
    __syncthreads();
    long long s2 = __builtin_amdgcn_s_memtime();
    tB += s2 - s1;
    // (c) trailing update of the lower block pairs (bi >= bj > jb)
    const int nb = TS / DB - jb - 1;
    for (int idx = wave; idx < nb * (nb + 1) / 2; idx += 4) {
      int bj = 0, rem = idx;
      while (rem >= nb - bj) {
        rem -= nb - bj;
        ++bj;
      }
      const int bi = bj + rem + jb + 1;
      bj += jb + 1;
      acc_t acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = At[(bj * DB + P::drow(lane, r)) * LDP + bi * DB + l15];
      T fa[4], fb[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        fa[ks] = -At[(j0 + ks * 4 + lq) * LDP + bj * DB + l15];
        fb[ks] = At[(j0 + ks * 4 + lq) * LDP + bi * DB + l15];
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = P::mfma(fa[ks], fb[ks], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) At[(bj * DB + P::drow(lane, r)) * LDP + bi * DB + l15] = acc[r];
    }
    __syncthreads();
    tC += __builtin_amdgcn_s_memtime() - s2;
  }
  long long t1 = __builtin_amdgcn_s_memtime();

  // (d) W = L^-1: block (i, j), i > j:  W_ij = -Dinv_i * sum_{kk=j}^{i-1} L_{i,kk} W_{kk,j}.
  // Blocks with the same i - j are independent (one level per barrier).  W_ij is stored transposed
  // at the upper-triangle position, i.e. W_ij[r][c] at At[(16 i + r) * LDP + 16 j + c].
  T *tsw = Ts + wave * DB * DB;
  for (int lev = 1; lev < TS / DB; ++lev) {
    for (int j = wave; j + lev < TS / DB; j += 4) {
      const int i = j + lev;
      acc_t acc = acc_t{0, 0, 0, 0};
      for (int kk = j; kk < i; ++kk) {
        T fa[4], fb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          fa[ks] = At[(kk * DB + ks * 4 + lq) * LDP + i * DB + l15];  // L_{i,kk}[r = l15][q]
          fb[ks] = (kk == j) ? Dv[j * DB * DB + l15 * DB + ks * 4 + lq]  // Dinv_j[q][c = l15]
                             : At[(kk * DB + ks * 4 + lq) * LDP + j * DB + l15];  // W_{kk,j}[q][c]
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = P::mfma(fa[ks], fb[ks], acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) tsw[P::drow(lane, r) * DB + l15] = acc[r];  // T[r][c]
      acc_t acc2 = acc_t{0, 0, 0, 0};
      T ga[4], gb[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        ga[ks] = -Dv[i * DB * DB + (ks * 4 + lq) * DB + l15];  // -Dinv_i[r = l15][q]
        gb[ks] = tsw[(ks * 4 + lq) * DB + l15];                // T[q][c = l15]
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc2 = P::mfma(ga[ks], gb[ks], acc2);
#pragma unroll
      for (int r = 0; r < 4; ++r) At[(i * DB + P::drow(lane, r)) * LDP + j * DB + l15] = acc2[r];
    }
    __syncthreads();
  }
  if (dbgbuf && tid == 0 && blockIdx.x == 0) {
    dbgbuf[0] = tA;
    dbgbuf[1] = tB;
    dbgbuf[2] = tC;
    dbgbuf[3] = t1 - t0;
    dbgbuf[4] = __builtin_amdgcn_s_memtime() - t1;
  }
}

// Phase (e): write L (upper triangle zeroed) to the factor panel and W_k (column-major 128 x 128).
template <typename T>
__device__ __forceinline__ void potf2_store(const FitArgs &p, const T *At, const T *Dv, const int *flag, T *tile,
                                            int ld, int b, int k, int tid) {
  if (tid == 0 && *flag != 0 && p.info[b] == 0) p.info[b] = *flag;
  T *Wk = reinterpret_cast<T *>(p.Winv) + (size_t)b * p.winv_stride + (size_t)k * TS * TS;
  for (int idx = tid; idx < TS * TS; idx += 256) {
    const int c = idx >> 7, r = idx & 127;
    tile[(size_t)c * ld + r] = (r >= c) ? At[c * LDP + r] : T(0);
    T w = T(0);
    if (r >= c) {
      if ((r >> 4) == (c >> 4)) w = Dv[(r >> 4) * DB * DB + (c & 15) * DB + (r & 15)];
      else w = At[r * LDP + c];
    }
    Wk[(size_t)c * TS + r] = w;
  }
}

#ifdef CGP_AB
template <typename T>
__global__ __launch_bounds__(256) void k_potf2(FitArgs p, int k) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *At = reinterpret_cast<T *>(smem_raw);  // element (r, c) at At[c * LDP + r]
  T *Dv = At + TS * LDP;                     // Dv[jb][q][x] = Dinv_jb[x][q]
  T *Ts = Dv + 8 * DB * DB;                  // per-wave 16x16 scratch
  int *flag = reinterpret_cast<int *>(Ts + 4 * DB * DB);
  const int b = blockIdx.x, tid = threadIdx.x;
  T *Lw = reinterpret_cast<T *>(p.Lw) + (size_t)b * p.lw_stride;
  const int ld = p.ld;
  T *tile = Lw + (size_t)(k * TS) * ld + (size_t)k * TS;
  for (int idx = tid; idx < TS * TS; idx += 256) {
    const int c = idx >> 7, r = idx & 127;
    At[c * LDP + r] = tile[(size_t)c * ld + r];
  }
  if (tid == 0) *flag = 0;
  __syncthreads();
  potf2_lds_body<T>(At, Dv, Ts, flag, k, tid, nullptr, p.N - k * TS);
  potf2_store<T>(p, At, Dv, flag, tile, ld, b, k, tid);
}

#endif  // CGP_AB

// --------------------------------------------------------------------------------------------------
// k_finalize: mean_m = V_m . z ; var_m = k** - |V_m|^2 (clip 1e-15, + sigma_n^2) ;
// logML = -0.5 z'z - sum log L_ii - N/2 log 2pi   (a5 z-part, a6, a8).  grid (ceil(M/64)+1, batch):
// the last block of each fit does the scalar reductions.
// --------------------------------------------------------------------------------------------------
template <typename T, int RB = 64>
__global__ __launch_bounds__(256) void k_finalize(FitArgs p, int do_logml) {
  using P = Prec<T>;
  __shared__ double red[2][256];
  const int b = blockIdx.y, tid = threadIdx.x;
  const T *Lw = reinterpret_cast<const T *>(p.Lw) + (size_t)b * p.lw_stride;
  const int ld = p.ld, N = p.N, M = p.M, NP = p.NT * TS;
  const size_t rb = (size_t)p.NT * TS;  // first extra row
  const double *th = p.theta + (size_t)b * MAX_THETA;
  const int kid = p.kernel_id;
  const int nth = (kid == K_SE_ISO) ? 3 : (kid == K_SE_ARD ? p.d + 2 : 4);
  // RB test rows per block, 256 / RB column groups (RB = 16 for the latency schedule: 4x the blocks)
  constexpr int NG = 256 / RB;
  const int nmb = (M + RB - 1) / RB;
  if ((int)blockIdx.x < nmb) {
    const int ml = tid % RB, g = tid / RB;
    const int m = blockIdx.x * RB + ml;
    double smu = 0, sq = 0;
    if (m < M) {
      // with accumulators only the last block column is still to be added
      const int c0 = p.macc ? (p.NT - 1) * TS : 0;
      if (p.macc && g == 0) {
        smu = p.macc[(size_t)b * M + m];
        sq = p.vacc[(size_t)b * M + m];
      }
      for (int c = c0 + g; c < NP; c += NG) {
        const double v = (double)Lw[(size_t)c * ld + rb + m];
        const double z = (double)Lw[(size_t)c * ld + rb + M];
        smu += v * z;
        sq += v * v;
      }
    }
    red[0][tid] = smu;
    red[1][tid] = sq;
    __syncthreads();
    if (g == 0 && m < M) {
      double mu, q;
      if constexpr (NG == 4) {
        mu = (red[0][ml] + red[0][64 + ml]) + (red[0][128 + ml] + red[0][192 + ml]);
        q = (red[1][ml] + red[1][64 + ml]) + (red[1][128 + ml] + red[1][192 + ml]);
      } else {
        mu = q = 0;
#pragma unroll
        for (int g2 = 0; g2 < NG; ++g2) {
          mu += red[0][g2 * RB + ml];
          q += red[1][g2 * RB + ml];
        }
      }
      double kss;
      if (kid == K_RBF_BROWNIAN) {
        const double xs = (double)reinterpret_cast<const T *>(p.Xs)[(size_t)b * p.d * M + m];
        kss = th[0] * th[2] * fabs(xs);
      } else {
        kss = th[0];
      }
      double v = kss - q;
      v = v < 1e-15 ? 1e-15 : v;
      if (p.include_noise) v += th[nth - 1];
      reinterpret_cast<T *>(p.mean)[(size_t)b * M + m] = (T)mu;
      reinterpret_cast<T *>(p.var)[(size_t)b * M + m] = (T)v;
    }
  } else if (do_logml) {
    double sl = 0, sz = 0;
    for (int c = tid; c < NP; c += 256) {
      if (c < N) sl += (double)P::log_(Lw[(size_t)c * ld + c]);
      const double z = (double)Lw[(size_t)c * ld + rb + M];
      sz += z * z;
    }
    red[0][tid] = sl;
    red[1][tid] = sz;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) {
        red[0][tid] += red[0][tid + s];
        red[1][tid] += red[1][tid + s];
      }
      __syncthreads();
    }
    if (tid == 0) p.logml[b] = -0.5 * red[1][0] - red[0][0] - 0.5 * (double)N * 1.8378770664093453;
  }
}

// --------------------------------------------------------------------------------------------------
// k_alpha: alpha = L^-T z (a5 "potrs" back substitution).  One workgroup per fit; z is the y row
// of the factor panel.  Per 128-tile (last to first):
//   rhs      = z_tile - L(below, tile)^T alpha(below)      one wave per column, shuffle reduction
//   alpha_t  = L(t,t)^-T rhs = W_t^T rhs                    W_t = L(t,t)^-1 is already in Winv (the panel
// kernels multiply by it), so the in-tile back substitution is a triangular matrix-vector product with
// no sequential dependency: two barriers per tile instead of one per column.  Only entries r >= c of a
// W_t column are read (at 16-row block granularity: the strictly upper 16x16 blocks are never written).
// --------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_alpha(FitArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *al = reinterpret_cast<double *>(smem_raw);  // [NT*128]
  double *rhs = al + p.NT * TS;                        // [128]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T *Lw = reinterpret_cast<const T *>(p.Lw) + (size_t)b * p.lw_stride;
  const T *Winv = reinterpret_cast<const T *>(p.Winv) + (size_t)b * p.winv_stride;
  const int ld = p.ld, NP = p.NT * TS, M = p.M;
  const size_t rb = (size_t)p.NT * TS;
  for (int tb = p.NT - 1; tb >= 0; --tb) {
    const int c0 = tb * TS, rbelow = c0 + TS;
    for (int cl = wave; cl < TS; cl += 4) {
      const T *col = Lw + (size_t)(c0 + cl) * ld;
      double s = 0;
      for (int r = rbelow + lane; r < NP; r += 64) s += (double)col[r] * al[r];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
      if (lane == 0) rhs[cl] = (double)col[rb + M] - s;
    }
    __syncthreads();
    const T *Wt = Winv + (size_t)tb * TS * TS;  // column-major: W[r][c] at Wt[c * 128 + r], zero for r < c
    for (int cl = wave; cl < TS; cl += 4) {
      const T *wc = Wt + (size_t)cl * TS;
      double s = 0;
      for (int r = (cl & ~(DB - 1)) + lane; r < TS; r += 64) s += (double)wc[r] * rhs[r];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
      if (lane == 0) al[c0 + cl] = s;
    }
    __syncthreads();
  }
  T *out = reinterpret_cast<T *>(p.alpha) + (size_t)b * p.alpha_stride;
  for (int i = tid; i < NP; i += 256) out[i] = (T)al[i];
}

// (n, d) row-major fp64 -> SoA [d][n] in the device dtype, one fit per blockIdx.y (the host-buffer entry
// points stage the caller's arrays untouched; transposition and conversion happen here).
template <typename T>
__global__ void k_pack_soa(const double *__restrict__ src, T *__restrict__ dst, int n, int d) {
  const size_t b = blockIdx.y;
  const double *s = src + b * (size_t)n * d;
  T *o = dst + b * (size_t)n * d;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * d; i += gridDim.x * blockDim.x) {
    const int r = i / d, q = i - r * d;
    o[(size_t)q * n + r] = (T)s[i];
  }
}

}  // namespace cgp
