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
// gp_slip_node.py:31-49 reaches.  This header holds the shared pieces (precision traits, the
// 16x16 diagonal-block factorisation, potf2_tile, k_finalize, k_alpha, k_pack_soa); the schedules -- per
// 128-column block step  S(i,k) = Gram(i,k) - sum_{j<k} L(i,j) L(k,j)^T  kept in MFMA accumulators,
// L(i,k) = S(i,k) W_k^T in registers, and the diagonal tile's potf2 + inverse W_k = L(k,k)^-1 -- are in
// cgp_kernels_fused.hpp (k_panel, k_diag_lean, k_tile_sk, k_trmm_sk, k_grad).
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
constexpr int NCB = TS / DB;  // 8 column blocks of 16
// W_k = L(k,k)^-1 is kept in HBM as the exact LDS image the in-register trmm reads: its 36 lower 16x16
// blocks, NEGATED (the panel accumulators hold -S),
//   Wimg[blk(cb, qb)][q][c] = -W[cb*16 + c][qb*16 + q],   blk(cb, qb) = cb (cb + 1) / 2 + qb,  qb <= cb,
// so staging it is a straight 16-byte-per-lane LDS-DMA copy of 36 (fp32) / 72 (fp64) KiB.
constexpr int WIMG = NCB * (NCB + 1) / 2 * DB * DB;
// fp32 fits of more than three input dimensions are refined (cgp_refine.hpp) per fit, when the factor itself says the window is
// dense: rho = (sigma_f^2 + sigma_n^2) / geometric mean of L_ii^2 >= RF_RHO -- how much of every Schur complement is cancellation.
// The unrefined mean's error against the oracle follows it (tests/fuzz/rho_vs_error.py, 192 fits of N = 1100, d = 3 ... 6, mean / worst
// error by rho: [0, 4) 2.8e-5 / 8.6e-5, [4, 8) 5.7e-5 / 1.7e-4, [8, 12) 9.8e-5 / 2.3e-4, [12, 16) 2.1e-4 / 7.6e-4, [16, 24) 3.0e-4 /
// 9.4e-4, >= 24 5.4e-4 / 1.6e-3): 12 keeps the worst unrefined fit a factor of four inside 1e-3.  BASELINE configs[2]'s 512
// windows sit at rho 2.9 ... 11.5: none is marked.
constexpr double RF_RHO = 12.0;
__host__ __device__ __forceinline__ constexpr int wimg_blk(int cb, int qb) { return (cb * (cb + 1) / 2 + qb) * DB * DB; }

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
  void *Winv;            // [batch][NTmax][WIMG]  negated block image of W_k = L(k,k)^-1 (see WIMG)
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
  int tile_off;          // first block-tile index of this launch (split panel launches, A/B overlap schedule)
  void *dpart;           // [batch][2][DPART] register images of pre-updated diagonal tiles (see diag_next)
  void *pimg;            // [batch][2][DPART] register images of pre-updated kind-A panel tiles (k_panel<T, true>, kind C)
  int diag_slots;        // k_panel<T, true>: 1 = the launch finishes diagonal tile k+1, 2 = pre-updates tile k+2,
                         // 4 = pre-updates the next launch's kind-A tile (k+2, k+1), 8 = kind A starts from that image
  int diag_stride;       // k_panel<T, true>: workgroups per CU (one block of 256 ids in `stride` holds the finishers)
  int img_slots;         // register-image slots per fit in dpart / pimg: 0 = 2 (by parity of the tile index), else one per tile index (k_sched)
  int xid;               // 1: the M (= N) "test rows" are the identity, so the extra block becomes (L^-1)^T (gradient mode)
  double *gpart;         // [batch][pairs][GRAD_N] per-tile-pair partial sums of k_grad
  int *rflag;            // [batch] or null: k_finalize marks the fits whose fp32 mean wants the refinement (cgp_refine.hpp, RF_RHO)
  void *Lp;              // fp32 mid-size calls: the panel tiles a second time, split into three bf16 planes (bx6p_loop); null otherwise
  size_t lp_stride;      // bf16 elements per fit (3 planes of lw_stride elements)
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
// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence on ALL memory: hipcc puts
// s_waitcnt vmcnt(0) in front of s_barrier, so a phase that also issued fire-and-forget HBM stores (a finished
// column of L, a block of the W image, a swept row of a window) would wait for their acknowledgements --
// microseconds -- at every barrier.  Use it only where no thread reads, before the next __syncthreads() or the
// end of the kernel, global memory another thread of the workgroup wrote since the last one.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// Execution span of a launch (CGP_DBG & 2048, -DCGP_ABLATION): thread 0 of every workgroup stamps the 100 MHz
// real-time counter at entry and exit; dbgbuf[384 + k] = 2^62 - earliest entry, dbgbuf[416 + k] = latest exit of
// block step k's k_panel launch (atomicMax both).  tools/launch_spans.py.  Nothing in the shipped library.
struct SpanClock {
#ifdef CGP_ABLATION
  unsigned long long r0, c0;
#endif
  __device__ __forceinline__ void enter(const FitArgs &p, int k, int tid) {
#ifdef CGP_ABLATION
    if (CGP_DBG_ON(p, 2048) && tid == 0) {
      r0 = __builtin_amdgcn_s_memrealtime();
      c0 = __builtin_amdgcn_s_memtime();
      atomicMax(reinterpret_cast<unsigned long long *>(p.dbgbuf) + 384 + (k & 31), (1ull << 62) - r0);
    }
#endif
  }
  // also sums, per block step, the workgroups' residence in real-time ticks (448 + k) and in s_memtime ticks
  // (480 + k): their ratio is the shader clock, the first over span x slots the occupancy actually reached
  __device__ __forceinline__ void leave(const FitArgs &p, int k, int tid) {
#ifdef CGP_ABLATION
    if (CGP_DBG_ON(p, 2048) && tid == 0) {
      const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
      unsigned long long *d = reinterpret_cast<unsigned long long *>(p.dbgbuf);
      atomicMax(d + 416 + (k & 31), r1);
      atomicAdd(d + 448 + (k & 31), r1 - r0);
      atomicAdd(d + 480 + (k & 31), c1 - c0);
    }
#endif
  }
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
  // the same coefficients as literals in the instruction stream: for a kernel that is ONE short launch, where the first
  // touch of the constant segment is a cold miss the whole workgroup waits on (k_small: 36 k cycles of a 130 k launch)
  __device__ __forceinline__ void load_literals() {
    c[0] = 1.6059043836821613e-10, c[1] = 2.08767569878681e-09, c[2] = 2.505210838544172e-08, c[3] = 2.755731922398589e-07;
    c[4] = 2.7557319223985893e-06, c[5] = 2.48015873015873e-05, c[6] = 1.984126984126984e-04, c[7] = 1.3888888888888889e-03;
    c[8] = 8.333333333333333e-03, c[9] = 4.1666666666666664e-02, c[10] = 1.6666666666666666e-01, c[11] = 0.5;
    c[12] = 1.4426950408889634, c[13] = -6.93147180369123816490e-01, c[14] = -1.90821492927058770002e-10, c[15] = -800.0;
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

// One 16x16 diagonal block in the registers of a wavefront: lane holds row (lane & 15) of the block
// in a[] (replicated over the four 16-lane groups).  On return a[] holds the row of the Cholesky
// factor and w[i] = Dinv[i][lane & 15] (column (lane & 15) of the block's inverse, by forward
// substitution).  A non-positive pivot is replaced by 1 and reported in `bad` (1-based global index).
// This is the serial critical path of the factorisation (128 dependent pivots per tile) and a lone wave
// pays for it by INSTRUCTION COUNT: every VALU instruction, DPP or not, is 4 cycles of issue, dependent
// ones issue back to back, and reordering alone bought 3 % (tools/potf2_block_bench.hip, variants 4 / 6).  So:
//  * "times lane c's value" is ONE v_fmac_f64_dpp (64-bit DPP row_newbcast of gfx90a+; the four rows hold
//    identical copies, a row-local broadcast is the right one) with the sign as its neg modifier;
//  * the Newton step of 1/sqrt(pivot) is folded into the column: l = l0 + l0 q with l0 = a r, not a (r + r q);
//  * no per-pivot repair on the fast path: a non-positive pivot turns every later pivot into NaN, so ONE test
//    of the last pivot tells, and the rare block that fails it is reloaded and redone by the repairing form;
//  * the inverse's substitution step J rides in the factor loop (column J of L is final there), no masks:
//    the exact zeros above the diagonal of the inverse stay exact zeros.
// 1.51 -> 1.02 us per fp64 block for a lone wave.  A DPP read of a VGPR needs two wait states after the
// VALU write: the first DPP instruction after each producer carries an s_nop 1.
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

template <int C, bool NOP> __device__ __forceinline__ void fnmac_bcast(float &acc, float src, float own) {  // acc -= src[lane C of the row] * own
  if constexpr (NOP) asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(C));
  else asm volatile("v_fmac_f32_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(C));
}
template <int C, bool NOP> __device__ __forceinline__ void fnmac_bcast(double &acc, double src, double own) {
  if constexpr (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(C));
  else asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(C));
}
// 1/sqrt(x) = r + r q with r the hardware seed: q of one third-order step (fp64, seed ~2^-23: error ~ e^3) or of
// one Newton step (fp32)
__device__ __forceinline__ double rsqrt_corr(double x, double r) {
  const double e = __builtin_fma(-(x * r), r, 1.0);
  return __builtin_fma(0.375, e, 0.5) * e;
}
__device__ __forceinline__ float rsqrt_corr(float x, float r) { return __builtin_fmaf(-0.5f * x * r, r, 0.5f); }
__device__ __forceinline__ double rsq_seed(double x) { return __builtin_amdgcn_rsq(x); }
__device__ __forceinline__ float rsq_seed(float x) { return __builtin_amdgcn_rsqf(x); }

// the repairing form: a non-positive pivot is replaced by 1 and reported
template <typename T>
__device__ __forceinline__ void factor_block16_repair(T (&a)[DB], T (&w)[DB], int &bad, int pivot_base, int l15) {
  T rinv[DB];
  static_for<0, DB>([&](auto jc) {
    constexpr int J = decltype(jc)::value;
    T dj = mov_bcast<J>(a[J]);
    const bool ok = dj > T(0);
    if (!ok && bad == 0) bad = pivot_base + J + 1;
    dj = ok ? dj : T(1);
    const T r = rsq_seed(dj), q = rsqrt_corr(dj, r);
    const T l0 = (ok ? a[J] : ((l15 == J) ? T(1) : a[J])) * r;   // lane J's a[J] is the pivot itself
    const T l = __builtin_fma(l0, q, l0);
    rinv[J] = __builtin_fma(r, q, r);
    a[J] = l;
    static_for<J + 1, DB>([&](auto cc) {
      constexpr int C = decltype(cc)::value;
      fnmac_bcast<C, C == J + 1>(a[C], l, l);
    });
  });
  // right-looking forward substitution for column l15 of the inverse: independent updates per step
  T t[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i) t[i] = (i == l15) ? T(1) : T(0);
  static_for<0, DB>([&](auto qc) {
    constexpr int Q = decltype(qc)::value;
    w[Q] = t[Q] * rinv[Q];
    static_for<Q + 1, DB>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      fnmac_bcast<I, false>(t[I], a[Q], w[Q]);
    });
  });
#pragma unroll
  for (int i = 0; i < DB; ++i) w[i] = (i < l15) ? T(0) : w[i];
}

// `reload` refills a[] with the block as it was (only called when a pivot failed)
template <typename T, typename Reload>
__device__ __forceinline__ void factor_block16(T (&a)[DB], T (&w)[DB], int &bad, int pivot_base, int l15, Reload &&reload) {
  T t[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i) t[i] = (i == l15) ? T(1) : T(0);
  T lp = T(0), wp = T(0), last = T(0);
  static_for<0, DB>([&](auto jc) {
    constexpr int J = decltype(jc)::value;
    const T dj = mov_bcast<J>(a[J]);
    if constexpr (J > 0) {   // step J-1's updates of the columns that pivot J does not need, and its substitution step
      static_for<J + 1, DB>([&](auto cc) {
        constexpr int C = decltype(cc)::value;
        fnmac_bcast<C, false>(a[C], lp, lp);
      });
      static_for<J, DB>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        fnmac_bcast<I, false>(t[I], lp, wp);
      });
    }
    const T r = rsq_seed(dj), q = rsqrt_corr(dj, r);
    const T l0 = a[J] * r;
    const T l = __builtin_fma(l0, q, l0);
    a[J] = l;
    if constexpr (J + 1 < DB) fnmac_bcast<J + 1, true>(a[J + 1], l, l);
    w[J] = t[J] * __builtin_fma(r, q, r);
    lp = l;
    wp = w[J];
    last = dj;
  });
  if (!(last > T(0))) {   // wave-uniform and rare: some pivot was not positive (every later one is NaN then)
    reload();
    factor_block16_repair<T>(a, w, bad, pivot_base, l15);
  }
}

// --------------------------------------------------------------------------------------------------
// potf2_tile: factor the LDS-resident 128x128 diagonal tile (a3 "potf2_diag"), invert the factor, and write
// L(k,k) and the block image of -W_k to HBM -- the latency schedule's per-step critical path, so it is
// organised around its one serial chain, the eight 16x16 diagonal blocks (factor_block16, wave 0):
//   F(jb)  wave 0: factor + invert diagonal block jb in registers (lane = row, DPP broadcasts)
//          waves 1-3, meanwhile: everything that is NOT on the chain -- the trailing update with panel
//          jb-1 of the block columns >= jb+1, row jb-1 of W = L^-1 (the inverse grows with the factor:
//          W_ij = -Dinv_i sum_{j <= q < i} L_iq W_qj only needs rows < i of W and Dinv_i; its image blocks go to
//          HBM straight from the accumulators)
//   P(jb)  all waves: panel  L(i,jb) = A(i,jb) Dinv_jb^T,  i > jb                       (MFMA 16x16x4)
//   U(jb)  all waves: update of block column jb+1 ONLY with panel jb -- all the next diagonal block needs
// so the chain is 8 x (factor + two short MFMA phases) and the ~60 % of the tile's work that used to
// follow the last pivot (inverse by levels, stores) or sit between pivots (whole trailing update) runs
// beside it.  LDS tile column-major with leading dimension LDP; W_ij is kept transposed in the unused upper
// triangle; Dv[jb] = Dinv_jb.  info = first non-positive pivot (1-based).
// --------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void potf2_tile(const FitArgs &p, T *At, T *Dv, T *Ts, int *flag, T *__restrict__ tile, int ld,
                                           int b, int k, int tid) {
  using P = Prec<T>;
  using acc_t = typename P::acc_t;
  constexpr int NB = TS / DB;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int live = p.N - k * TS;
  T *__restrict__ Wk = reinterpret_cast<T *>(p.Winv) + (size_t)b * p.winv_stride + (size_t)k * WIMG;
  T *tsw = Ts + wave * DB * DB;

  // C(bi, bj) -= L(bi, jp) L(bj, jp)^T   (bi >= bj > jp)
  auto trailing_block = [&](int bi, int bj, int jp) {
    acc_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = At[(bj * DB + P::drow(lane, r)) * LDP + bi * DB + l15];
    T fa[4], fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fa[ks] = -At[(jp * DB + ks * 4 + lq) * LDP + bj * DB + l15];
      fb[ks] = At[(jp * DB + ks * 4 + lq) * LDP + bi * DB + l15];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = P::mfma(fa[ks], fb[ks], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) At[(bj * DB + P::drow(lane, r)) * LDP + bi * DB + l15] = acc[r];
  };
  // N blocks of one block row at once, C(bi, bj0 + m) -= L(bi, jp) L(bj0 + m, jp)^T: the row's panel block is loaded
  // once, every LDS read is in flight before the first MFMA, N MFMA chains interleaved (the helper waves are bound
  // by LDS and MFMA latency, not by the pipes: one block at a time ran at a sixth of the MFMA rate).  Per block
  // the arithmetic is trailing_block's.
  auto trailing_multi = [&](auto nc, int bi, int bj0, int jp) {
    constexpr int N = decltype(nc)::value;
    acc_t a[N];
    T fa[N][4], fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb[ks] = At[(jp * DB + ks * 4 + lq) * LDP + bi * DB + l15];
#pragma unroll
    for (int m = 0; m < N; ++m) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fa[m][ks] = -At[(jp * DB + ks * 4 + lq) * LDP + (bj0 + m) * DB + l15];
#pragma unroll
      for (int r = 0; r < 4; ++r) a[m][r] = At[((bj0 + m) * DB + P::drow(lane, r)) * LDP + bi * DB + l15];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int m = 0; m < N; ++m) a[m] = P::mfma(fa[m][ks], fb[ks], a[m]);
#pragma unroll
    for (int m = 0; m < N; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) At[((bj0 + m) * DB + P::drow(lane, r)) * LDP + bi * DB + l15] = a[m][r];
  };
  // blocks (bi, bj0 .. bi) of block row bi
  auto trailing_row = [&](int bi, int bj0, int jp) {
    using std::integral_constant;
    switch (bi - bj0 + 1) {
      case 1: trailing_multi(integral_constant<int, 1>{}, bi, bj0, jp); break;
      case 2: trailing_multi(integral_constant<int, 2>{}, bi, bj0, jp); break;
      case 3: trailing_multi(integral_constant<int, 3>{}, bi, bj0, jp); break;
      case 4: trailing_multi(integral_constant<int, 4>{}, bi, bj0, jp); break;
      case 5: trailing_multi(integral_constant<int, 3>{}, bi, bj0, jp); trailing_multi(integral_constant<int, 2>{}, bi, bj0 + 3, jp); break;
      case 6: trailing_multi(integral_constant<int, 3>{}, bi, bj0, jp); trailing_multi(integral_constant<int, 3>{}, bi, bj0 + 3, jp); break;
      default: break;
    }
  };
  // W(i, j) = -Dinv_i sum_{j <= q < i} L(i, q) W(q, j), written to LDS (transposed, upper triangle) and, negated,
  // to its block of the HBM image straight from the accumulator.  The sum runs on two accumulators (even / odd
  // q) so that consecutive 4-MFMA products do not wait for each other.
  auto inverse_block = [&](int i, int j) {
    constexpr int MAXT = NB - 1;
    const int n = i - j;   // terms, 1 .. 7; every operand is read before the first MFMA
    T fa[MAXT][4], fb[MAXT][4], ga[4];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
      if (t < n) {
        const int kk = j + t;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          fa[t][ks] = At[(kk * DB + ks * 4 + lq) * LDP + i * DB + l15];  // L_{i,kk}[r = l15][q]
          fb[t][ks] = (t == 0) ? Dv[j * DB * DB + l15 * DB + ks * 4 + lq]  // Dinv_j[q][c = l15]
                               : At[(kk * DB + ks * 4 + lq) * LDP + j * DB + l15];  // W_{kk,j}[q][c]
        }
      }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ga[ks] = -Dv[i * DB * DB + (ks * 4 + lq) * DB + l15];  // -Dinv_i[r = l15][q]
    acc_t acc0 = acc_t{0, 0, 0, 0}, acc1 = acc0;
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
      if (t < n) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (t & 1) acc1 = P::mfma(fa[t][ks], fb[t][ks], acc1);
          else acc0 = P::mfma(fa[t][ks], fb[t][ks], acc0);
        }
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) tsw[P::drow(lane, r) * DB + l15] = acc0[r] + acc1[r];  // T[r][c] (wave-private scratch)
    acc_t acc2 = acc_t{0, 0, 0, 0};
    T gb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) gb[ks] = tsw[(ks * 4 + lq) * DB + l15];                // T[q][c = l15]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc2 = P::mfma(ga[ks], gb[ks], acc2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      At[(i * DB + P::drow(lane, r)) * LDP + j * DB + l15] = acc2[r];          // W_ij[row drow][col l15]
      Wk[wimg_blk(i, j) + l15 * DB + P::drow(lane, r)] = -acc2[r];             // image [q = col][c = row]
    }
  };
  // The blocks of a row of W cost i - j + 1 products each (j = 0 the longest): dealt in snake order over
  // `nw` waves (0 1 2 2 1 0 0 1 ...) the loads come out equal.
  auto snake = [](int idx, int nw) {
    const int q = idx / nw, r = idx % nw;
    return (q & 1) ? nw - 1 - r : r;
  };
  // Column blocks [jb0, jb1) of L (from the top of each diagonal block down; zero above the diagonal inside it) and
  // the images of their Dinv blocks -> HBM, by waves w0 .. w0 + nw - 1: one column per wave-instruction, two rows
  // per lane (16-byte accesses), the LDS reads of eight columns batched ahead of their stores.
  auto store_blocks = [&](int jb0, int jb1, int w0, int nw) {
    using vec2 = T __attribute__((ext_vector_type(2)));
    const int wl = wave - w0;
#pragma unroll 1
    for (int c0 = jb0 * DB + wl * 8; c0 < jb1 * DB; c0 += nw * 8) {
      vec2 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = c0 + q, r = 2 * lane;   // rows r, r + 1 of column c: one 16-byte LDS read, one 16-byte store
        vec2 x = *reinterpret_cast<const vec2 *>(At + c * LDP + r);
        x[0] = r >= c ? x[0] : T(0);
        x[1] = r + 1 >= c ? x[1] : T(0);
        v[q] = x;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = c0 + q, r = 2 * lane;
        if (r + 1 >= (c & ~(DB - 1))) *reinterpret_cast<vec2 *>(tile + (size_t)c * ld + r) = v[q];
      }
    }
    for (int e = jb0 * DB * DB + wl * 64 + lane; e < jb1 * DB * DB; e += nw * 64) Wk[wimg_blk(e >> 8, e >> 8) + (e & 255)] = -Dv[e];
  };

#ifdef CGP_ABLATION
  long long tF = 0, tP = 0, tU = 0, tm = __builtin_amdgcn_s_memtime();
#define POTF2_LAP(x) { const long long nn = __builtin_amdgcn_s_memtime(); x += nn - tm; tm = nn; }
#else
#define POTF2_LAP(x)
#endif
  // A ragged last tile (the window ends inside it) is factored on its live block rows only: rows and columns beyond
  // the window are the identity (off-diagonal blocks exactly zero: the Gram tile pads so and the updates add zero rows), so
  // every product that involves them is zero and every loop below runs over nbl instead of NB block rows.  What the
  // skipped steps would have produced is written here once: identity diagonal blocks of L and of W, zero blocks of the
  // W image.  (A window of 134 ticks used to pay a whole tile's factorisation for its six rows beyond 128.)
  const int nbl = min(NB, (live + DB - 1) / DB);
  if (nbl < NB) {
    for (int e = nbl * DB * DB + tid; e < NB * DB * DB; e += 256) {
      const int jb = e >> 8, c = (e >> 4) & 15, r = e & 15;
      const T v = (r == c) ? T(1) : T(0);
      At[(jb * DB + c) * LDP + jb * DB + r] = v;
      Dv[e] = v;   // Dv[jb][r][c]: the identity either way
    }
    for (int i = nbl; i < NB; ++i)
      for (int e = tid; e < i * DB * DB; e += 256) Wk[wimg_blk(i, e >> 8) + (e & 255)] = T(0);
  }
  for (int jb = 0; jb < nbl; ++jb) {
    const int j0 = jb * DB;
    // ---- F(jb)
    if (wave == 0) {
      // lane holds row (lane & 15) of the diagonal block (replicated over the four 16-lane groups)
      T a[DB], w[DB];
      int bad = 0;
      if (j0 < live) {
#pragma unroll
        for (int c = 0; c < DB; ++c) a[c] = At[(j0 + c) * LDP + j0 + l15];
#ifdef CGP_ABLATION
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long tl = __builtin_amdgcn_s_memtime();
#endif
        factor_block16<T>(a, w, bad, k * TS + j0, l15, [&] {
#pragma unroll
          for (int c = 0; c < DB; ++c) a[c] = At[(j0 + c) * LDP + j0 + l15];
        });
#ifdef CGP_ABLATION
        if (CGP_DBG_ON(p, 1024) && tid == 0) {
          asm volatile("" : "+v"(w[15]), "+v"(a[15]));
          const long long tf = __builtin_amdgcn_s_memtime();
          unsigned long long *d = reinterpret_cast<unsigned long long *>(p.dbgbuf);
          atomicAdd(d + 42, (unsigned long long)(tl - tm));
          atomicAdd(d + 43, (unsigned long long)(tf - tl));
        }
#endif
        if (bad != 0 && lane == 0 && *flag == 0) *flag = bad;
      } else {  // identity padding beyond the window (N not a multiple of 128): nothing to factor
#pragma unroll
        for (int c = 0; c < DB; ++c) a[c] = w[c] = (c == l15) ? T(1) : T(0);
      }
      if (lane < DB) {   // the block's strictly upper triangle in LDS is read by nobody (store_blocks masks it)
#pragma unroll
        for (int c = 0; c < DB; ++c) At[(j0 + c) * LDP + j0 + l15] = a[c];
#pragma unroll
        for (int i = 0; i < DB; ++i) Dv[jb * DB * DB + l15 * DB + i] = w[i];
      }
#ifdef CGP_ABLATION
      if (CGP_DBG_ON(p, 1024) && tid == 0) atomicAdd(reinterpret_cast<unsigned long long *>(p.dbgbuf) + 37, (unsigned long long)(__builtin_amdgcn_s_memtime() - tm));
#endif
    } else {
      if (jb > 0) {
        const int jp = jb - 1, nb = nbl - 1 - jb;  // block columns jb+1 .. nbl-1 still take panel jp
#ifdef CGP_ABLATION
        long long h0 = __builtin_amdgcn_s_memtime();
#endif
        for (int j = 0; j < jp; ++j)   // row jp of W, snake order over the three helper waves
          if (snake(j, 3) == wave - 1) inverse_block(jp, j);
#ifdef CGP_ABLATION
        long long h1 = __builtin_amdgcn_s_memtime();
#endif
        // block rows jb+1 .. 7 of the trailing matrix have 1 .. nb blocks; dealt so that the three helpers get
        // 7 7 7 / 5 5 5 / 4 3 3 / 3 2 1 / 2 1 / 1 blocks: rows nb - h and nb - 5 + h (1-based row lengths), helper h
        const int h = wave - 1;
        const int r1 = nb - h, r2 = nb - 5 + h;
        if (r1 >= 1) trailing_row(jb + r1, jb + 1, jp);
        if (r2 >= 1 && r2 < r1) trailing_row(jb + r2, jb + 1, jp);
#ifdef CGP_ABLATION
        if (CGP_DBG_ON(p, 1024) && tid == 64) {
          unsigned long long *d = reinterpret_cast<unsigned long long *>(p.dbgbuf);
          atomicAdd(d + 40, (unsigned long long)(h1 - h0));
          atomicAdd(d + 41, (unsigned long long)(__builtin_amdgcn_s_memtime() - h1));
        }
#endif
      }
#ifdef CGP_ABLATION
      if (CGP_DBG_ON(p, 1024) && tid == 64) atomicAdd(reinterpret_cast<unsigned long long *>(p.dbgbuf) + 38, (unsigned long long)(__builtin_amdgcn_s_memtime() - tm));
#endif
    }
    lds_barrier();  // HBM stores of the helpers stay in flight
    POTF2_LAP(tF)
    // ---- P(jb): rows of block bi below the diagonal block, P[r][c] = sum_q A[r][q] Dinv[c][q]
    for (int bi = jb + 1 + wave; bi < nbl; bi += 4) {
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
    lds_barrier();
    POTF2_LAP(tP)
    // ---- U(jb): block column jb + 1 with panel jb
    for (int bi = jb + 1 + wave; bi < nbl; bi += 4) trailing_block(bi, jb + 1, jb);
    lds_barrier();
    POTF2_LAP(tU)
  }
  // ---- tail: the last live row of W and the last column
  for (int j = 0; j < nbl - 1; ++j)
    if (snake(j + 1, 4) == wave) inverse_block(nbl - 1, j);   // offset 1: wave 0 comes out of the last factor block last
  // L and the Dinv blocks of the image go to HBM once, here, by all four waves (16-byte stores: the tail is
  // store-issue-bound).  Spreading them over the last factor phases, where the helper waves have few trailing
  // blocks left, was measured: the phases grew by more than the tail shrank.  The strictly upper 16x16 blocks of
  // the tile in HBM are never read by anybody and are left alone.
  store_blocks(0, NB, 0, 4);
  if (tid == 0 && *flag != 0 && p.info[b] == 0) p.info[b] = *flag;
#ifdef CGP_ABLATION
  if (CGP_DBG_ON(p, 1024) && tid == 0) {
    unsigned long long *d = reinterpret_cast<unsigned long long *>(p.dbgbuf);
    atomicAdd(d + 32, (unsigned long long)tF);
    atomicAdd(d + 33, (unsigned long long)tP);
    atomicAdd(d + 34, (unsigned long long)tU);
    atomicAdd(d + 35, (unsigned long long)(__builtin_amdgcn_s_memtime() - tm));
    atomicAdd(d + 36, 1ull);
  }
#endif
#undef POTF2_LAP
}

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
    if (tid == 0) {
      p.logml[b] = -0.5 * red[1][0] - red[0][0] - 0.5 * (double)N * 1.8378770664093453;
      // rho = prior variance / geometric mean of the pivots L_ii^2 (the conditional variances): how much of every Schur
      // complement is cancellation -- the quantity the fp32 mean's error follows
      // (the unrefined error at a given rho grows with the window's length: N = 1100 / 2200 / 4400, worst fit with rho in [8, 12):
      // 2.3e-4 / 2.0e-4 / 3.1e-4, in [12, 16): 7.6e-4 / 7.2e-4 / 1.3e-3 -- longer windows are marked earlier)
      const double rho_min = p.NT > 48 ? 5.0 : (p.NT > 24 ? 8.0 : RF_RHO);
      if (p.rflag) p.rflag[b] = (th[0] + th[nth - 1]) * exp(-2.0 * red[0][0] / (double)N) >= rho_min ? 1 : 0;
    }
  }
}

// --------------------------------------------------------------------------------------------------
// k_alpha: alpha = L^-T z (a5 "potrs" back substitution).  One workgroup per fit; z is the y row
// of the factor panel.  Per 128-tile (last to first):
//   rhs      = z_tile - L(below, tile)^T alpha(below)      one wave per column, shuffle reduction
//   alpha_t  = L(t,t)^-T rhs = W_t^T rhs                    W_t = L(t,t)^-1 is already in Winv (the panel
// kernels multiply by it; negated block image, see WIMG), so the in-tile back substitution is a
// triangular matrix-vector product with no sequential dependency: two barriers per tile instead of
// one per column.
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
  // CG columns per wave at a time: their loads are all in flight before the first reduction (one column at a time
  // was one HBM round trip per column, 32 in a row per wave and phase); per column the sums are formed in the same order.
  constexpr int CG = 8;
  for (int tb = p.NT - 1; tb >= 0; --tb) {
    const int c0 = tb * TS, rbelow = c0 + TS;
    for (int cg = wave * CG; cg < TS; cg += 4 * CG) {
      double s[CG];
#pragma unroll
      for (int j = 0; j < CG; ++j) s[j] = 0;
      for (int r = rbelow + lane; r < NP; r += 64) {
        const double a = al[r];
#pragma unroll
        for (int j = 0; j < CG; ++j) s[j] += (double)Lw[(size_t)(c0 + cg + j) * ld + r] * a;
      }
      double z[CG];
#pragma unroll
      for (int j = 0; j < CG; ++j) z[j] = (double)Lw[(size_t)(c0 + cg + j) * ld + rb + M];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int j = 0; j < CG; ++j) s[j] += __shfl_xor(s[j], off);
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < CG; ++j) rhs[cg + j] = z[j] - s[j];
      }
    }
    __syncthreads();
    const T *Wt = Winv + (size_t)tb * WIMG;
    for (int cg = wave * CG; cg < TS; cg += 4 * CG) {
      const int qb = cg >> 4;   // the CG columns share their 16-column block (CG divides 16)
      double s[CG];
#pragma unroll
      for (int j = 0; j < CG; ++j) s[j] = 0;
      for (int r = qb * DB + lane; r < TS; r += 64) {  // -W[r][cl] at block (r >> 4, qb), entry [q][r & 15]
        const double x = rhs[r];
#pragma unroll
        for (int j = 0; j < CG; ++j) s[j] -= (double)Wt[wimg_blk(r >> 4, qb) + ((cg + j) & 15) * DB + (r & 15)] * x;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int j = 0; j < CG; ++j) s[j] += __shfl_xor(s[j], off);
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < CG; ++j) al[c0 + cg + j] = s[j];
      }
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

// The three k_pack_soa of a host-buffer call, the copy of theta to its own buffer and the zeroing of the jitter in ONE
// launch (a small call is a handful of short launches: five of them were staging).  raw = [X | y | Xs | theta] as staged.
template <typename T>
__global__ void k_pack_call(const double *__restrict__ raw, T *__restrict__ dX, T *__restrict__ dy, T *__restrict__ dXs,
                            double *__restrict__ dtheta, double *__restrict__ djitter, int batch, int N, int d, int M) {
  const size_t b = blockIdx.y, B = batch;
  const double *sx = raw + b * (size_t)N * d, *sy = raw + B * N * d + b * (size_t)N;
  const double *sxs = raw + B * N * d + B * N + b * (size_t)M * d, *sth = raw + B * N * d + B * N + B * (size_t)M * d + b * MAX_THETA;
  T *ox = dX + b * (size_t)N * d, *oy = dy + b * (size_t)N, *oxs = dXs + b * (size_t)M * d;
  const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = t0; i < N * d; i += stride) {
    const int r = i / d, q = i - r * d;
    ox[(size_t)q * N + r] = (T)sx[i];
  }
  for (int i = t0; i < N; i += stride) oy[i] = (T)sy[i];
  for (int i = t0; i < M * d; i += stride) {
    const int r = i / d, q = i - r * d;
    oxs[(size_t)q * M + r] = (T)sxs[i];
  }
  if (t0 < MAX_THETA) dtheta[b * MAX_THETA + t0] = sth[t0];
  if (t0 == 0) djitter[b] = 0.0;
}

}  // namespace cgp
