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
//   k_potf2  : S(k,k) = L(k,k) L(k,k)^T  + inverses of its eight 16x16 diagonal blocks
//   k_trsm   : L(i,k) = S(i,k) L(k,k)^-T   (blocked substitution)
// and k_finalize for mean / variance / log marginal likelihood, k_alpha for alpha = L^-T z.
//
// Storage: Lw is column-major, leading dimension ld (multiple of 128), one slab per fit.
// Inputs are SoA per fit: X[d][N], Xs[d][M], y[N].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cgp {

constexpr int TS = 128;    // tile edge (rows and columns)
constexpr int KT = 16;     // k-chunk staged through LDS per barrier
constexpr int LDST = 144;  // LDS row stride of a staged chunk (elements): (144*8) % 256 == 128 and
                           // (144*4) % 128 == 64, so the 4 k-groups of an MFMA operand read hit
                           // disjoint bank halves for both fp64 (ds_read_b64) and fp32 (ds_read_b32)
constexpr int DB = 16;     // diagonal sub-block of potf2 / trsm
constexpr int MAXD = 8;
constexpr int MAX_THETA = MAXD + 2;
constexpr int LDA_P = TS + 1;  // padded LDS leading dimension of the potf2 / trsm tile

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
  void *Dinv;            // [batch][NTmax][8][16*16]
  size_t dinv_stride;    // elements per fit
  int *info;             // [batch]
  void *mean, *var;      // [batch][M]
  double *logml;         // [batch]
  void *alpha;           // [batch][NT*128]
  size_t alpha_stride;
  int N, d, M, NT, ET, kernel_id, include_noise;
  int rows_from_extra;   // 1: only the extra (test/y) row tiles are processed (predict after fit)
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
};

// Row-tile index (in units of 128 rows of Lw) handled by block t of a step-k launch.
// In-matrix tiles first (first_in .. NT-1), then the ET extra tiles, which start at row NT*128.
__device__ __forceinline__ int row_tile_of(int t, int first_in, int NT, int rows_from_extra) {
  if (rows_from_extra) return NT + t;
  const int nin = NT - first_in;
  return (t < nin) ? first_in + t : NT + (t - nin);
}

// --------------------------------------------------------------------------------------------------
// Covariance entry.  SE kernels get inputs pre-divided by the length-scales (xr, xc are scaled);
// RBF x Brownian follows GPy's r^2 = x^2 + x'^2 - 2xx' clipped at 0 (diag forced to 0) on raw x.
// --------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T cov_entry(int kid, int d, const T *xr, int rs, const T *xc, int cs, T amp,
                                       T inv_ell, T amp_b, bool same) {
  if (kid != K_RBF_BROWNIAN) {
    T d2 = 0;
#pragma unroll 1
    for (int q = 0; q < d; ++q) {
      const T df = xr[q * rs] - xc[q * cs];
      d2 += df * df;
    }
    return amp * Prec<T>::exp_(T(-0.5) * d2);
  }
  const T x = xr[0], xp = xc[0];
  T r2 = same ? T(0) : (T(-2) * x * xp + (x * x + xp * xp));
  r2 = r2 < T(0) ? T(0) : r2;
  const T r = Prec<T>::sqrt_(r2) * inv_ell;
  const T krbf = amp * Prec<T>::exp_(T(-0.5) * r * r);
  const int sx = (x > T(0)) - (x < T(0)), sp = (xp > T(0)) - (xp < T(0));
  const T ax = x < T(0) ? -x : x, ap = xp < T(0) ? -xp : xp;
  const T kb = (sx == sp) ? amp_b * (ax < ap ? ax : ap) : T(0);
  return krbf * kb;
}

// --------------------------------------------------------------------------------------------------
// k_update: S(rt, k) = Gram(rt, k) - sum_{j < k} L(rt, j) L(k, j)^T   (a2 gram + a3 syrk/gemm + a8)
// grid (row tiles, batch), 256 threads = 4 waves as 2 (rows) x 2 (cols), each wave a 64x64 block
// of 4x4 MFMA 16x16x4 accumulators.  MFMA "A" operand = rows of tile k (S columns), "B" operand =
// rows of tile rt (S rows): lane&15 then runs along S rows, which are contiguous in memory.
// --------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void k_update(FitArgs p, int k) {
  using P = Prec<T>;
  using acc_t = typename P::acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  // layout: buf[2] x { rowsChunk[KT][LDST], colsChunk[KT][LDST] }
  constexpr int CH = KT * LDST;

  const int b = blockIdx.y;
  const int rt = row_tile_of(blockIdx.x, k, p.NT, p.rows_from_extra);
  T *Lw = reinterpret_cast<T *>(p.Lw) + (size_t)b * p.lw_stride;
  const int ld = p.ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  acc_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

  const int nchunk = (k * TS) / KT;
  // staging map: thread -> (column sc of the chunk, 8 consecutive rows from sr)
  const int sc = tid >> 4, sr = (tid & 15) * 8;
  const T *gR = Lw + (size_t)rt * TS + sr;  // rows of tile rt
  const T *gC = Lw + (size_t)k * TS + sr;   // rows of tile k
  using vec8 = T __attribute__((ext_vector_type(8)));
  vec8 pr, pc;
  if (nchunk > 0) {
    pr = *reinterpret_cast<const vec8 *>(gR + (size_t)sc * ld);
    pc = *reinterpret_cast<const vec8 *>(gC + (size_t)sc * ld);
    *reinterpret_cast<vec8 *>(smem + sc * LDST + sr) = pr;
    *reinterpret_cast<vec8 *>(smem + CH + sc * LDST + sr) = pc;
  }
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const T *cur = smem + (c & 1) * 2 * CH;
    if (c + 1 < nchunk) {
      const size_t off = (size_t)((c + 1) * KT + sc) * ld;
      pr = *reinterpret_cast<const vec8 *>(gR + off);
      pc = *reinterpret_cast<const vec8 *>(gC + off);
    }
#pragma unroll
    for (int ks = 0; ks < KT / 4; ++ks) {
      T fa[4], fb[4];
      const T *ra = cur + CH + (ks * 4 + lq) * LDST + wc * 64 + l15;  // tile-k rows  -> S columns
      const T *rb = cur + (ks * 4 + lq) * LDST + wr * 64 + l15;       // tile-rt rows -> S rows
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = ra[i * 16];
        fb[i] = rb[i * 16];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = P::mfma(fa[i], fb[j], acc[i][j]);
    }
    if (c + 1 < nchunk) {
      T *nxt = smem + ((c + 1) & 1) * 2 * CH;
      *reinterpret_cast<vec8 *>(nxt + sc * LDST + sr) = pr;
      *reinterpret_cast<vec8 *>(nxt + CH + sc * LDST + sr) = pc;
    }
    __syncthreads();
  }

  // ---- epilogue: Gram tile from the inputs, S = G - acc ----
  const double *th = p.theta + (size_t)b * MAX_THETA;
  const int kid = p.kernel_id, d = p.d, N = p.N, M = p.M;
  const bool extra = rt >= p.NT;
  T *xr = smem;             // [d][128] rows of this tile (training rows or test rows)
  T *xc = smem + MAXD * TS; // [d][128] columns = training rows of tile k
  T *yc = smem + 2 * MAXD * TS;  // [128] y of tile k columns (only for the y row)
  const T *Xb = reinterpret_cast<const T *>(p.X) + (size_t)b * d * N;
  const T *Xsb = reinterpret_cast<const T *>(p.Xs) + (size_t)b * d * M;
  const T *yb = reinterpret_cast<const T *>(p.y) + (size_t)b * N;
  for (int idx = tid; idx < d * TS; idx += 256) {
    const int q = idx >> 7, r = idx & 127;
    T sc_q = T(1);
    if (kid == K_SE_ISO) sc_q = T(1.0 / th[1]);
    else if (kid == K_SE_ARD) sc_q = T(1.0 / th[1 + q]);
    const int gc = k * TS + r;
    xc[q * TS + r] = (gc < N) ? Xb[(size_t)q * N + gc] * sc_q : T(0);
    T v = T(0);
    if (!extra) {
      const int gr = rt * TS + r;
      if (gr < N) v = Xb[(size_t)q * N + gr] * sc_q;
    } else {
      const int e = (rt - p.NT) * TS + r;
      if (e < M) v = Xsb[(size_t)q * M + e] * sc_q;
    }
    xr[q * TS + r] = v;
  }
  if (tid < TS) {
    const int gc = k * TS + tid;
    yc[tid] = (gc < N) ? yb[gc] : T(0);
  }
  __syncthreads();
  const T amp = T(th[0]);
  const T inv_ell = (kid == K_RBF_BROWNIAN) ? T(1.0 / th[1]) : T(1);
  const T amp_b = (kid == K_RBF_BROWNIAN) ? T(th[2]) : T(0);
  const int nth = (kid == K_SE_ISO) ? 3 : (kid == K_SE_ARD ? d + 2 : 4);
  const T diag_add = T(th[nth - 1] + 1e-8 + (p.jitter ? p.jitter[b] : 0.0));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rl = wr * 64 + j * 16 + l15;  // local S row
    const int grow = extra ? (rt - p.NT) * TS + rl : rt * TS + rl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cl = wc * 64 + i * 16 + P::drow(lane, r);  // local S column
        const int gcol = k * TS + cl;
        T g;
        if (!extra) {
          if (grow < N && gcol < N) {
            g = cov_entry<T>(kid, d, xr + rl, TS, xc + cl, TS, amp, inv_ell, amp_b, grow == gcol);
            if (grow == gcol) g += diag_add;
          } else {
            g = (grow == gcol) ? T(1) : T(0);  // identity padding keeps the factor well defined
          }
        } else {
          if (gcol >= N || grow > M) g = T(0);
          else if (grow == M) g = yc[cl];
          else g = cov_entry<T>(kid, d, xr + rl, TS, xc + cl, TS, amp, inv_ell, amp_b, false);
        }
        Lw[(size_t)gcol * ld + (size_t)rt * TS + rl] = g - acc[i][j][r];
      }
    }
  }
}

// --------------------------------------------------------------------------------------------------
// k_potf2: factor the 128x128 diagonal tile in LDS (a3 "potf2_diag") and invert its eight 16x16
// diagonal blocks for k_trsm.  One workgroup per fit.  info = first non-positive pivot (1-based).
// --------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_potf2(FitArgs p, int k) {
  using P = Prec<T>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *A = reinterpret_cast<T *>(smem_raw);  // A[r * LDA_P + c]
  const int b = blockIdx.x, tid = threadIdx.x;
  T *Lw = reinterpret_cast<T *>(p.Lw) + (size_t)b * p.lw_stride;
  const int ld = p.ld;
  T *tile = Lw + (size_t)(k * TS) * ld + (size_t)k * TS;
  for (int idx = tid; idx < TS * TS; idx += 256) {
    const int c = idx >> 7, r = idx & 127;
    A[r * LDA_P + c] = tile[(size_t)c * ld + r];
  }
  int bad = 0;
  for (int jb = 0; jb < TS / DB; ++jb) {
    const int j0 = jb * DB;
    for (int j = j0; j < j0 + DB; ++j) {
      __syncthreads();
      T ajj = A[j * LDA_P + j];
      if (!(ajj > T(0))) {
        if (bad == 0) bad = k * TS + j + 1;
        ajj = T(1);
      }
      const T dj = P::sqrt_(ajj);
      const T dinv = T(1) / dj;
      __syncthreads();
      if (tid < TS) {
        if (tid > j) A[tid * LDA_P + j] *= dinv;
        else if (tid == j) A[j * LDA_P + j] = dj;
      }
      __syncthreads();
      const int r = tid & 127, half = tid >> 7;
      if (r > j) {
        const T lrj = A[r * LDA_P + j];
        for (int c = j + 1 + half; c < j0 + DB; c += 2)
          if (r >= c) A[r * LDA_P + c] -= lrj * A[c * LDA_P + j];
      }
    }
    __syncthreads();
    // trailing update of everything right of the 16-wide panel (lower triangle only)
    const int t0 = j0 + DB, nrem = TS - t0;
    for (int idx = tid; idx < nrem * nrem; idx += 256) {
      const int rr = idx % nrem, cc = idx / nrem;
      if (rr >= cc) {
        const T *ar = A + (t0 + rr) * LDA_P + j0, *ac = A + (t0 + cc) * LDA_P + j0;
        T s = 0;
#pragma unroll
        for (int q = 0; q < DB; ++q) s += ar[q] * ac[q];
        A[(t0 + rr) * LDA_P + t0 + cc] -= s;
      }
    }
  }
  __syncthreads();
  if (tid == 0 && bad != 0 && p.info[b] == 0) p.info[b] = bad;
  // inverses of the eight 16x16 diagonal blocks: thread -> (block, column), forward substitution
  if (tid < TS) {
    const int blk = tid >> 4, col = tid & 15, base = blk * DB;
    T w[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i) {
      const T *ai = A + (base + i) * LDA_P + base;
      T s = 0;
#pragma unroll
      for (int q = 0; q < DB; ++q)
        if (q < i) s += (q >= col ? ai[q] * w[q] : T(0));
      const T dinv = T(1) / ai[i];
      w[i] = (i < col) ? T(0) : ((i == col) ? dinv : -s * dinv);
    }
    T *Di = reinterpret_cast<T *>(p.Dinv) + (size_t)b * p.dinv_stride + ((size_t)k * 8 + blk) * (DB * DB);
#pragma unroll
    for (int i = 0; i < DB; ++i) Di[i * DB + col] = w[i];
  }
  for (int idx = tid; idx < TS * TS; idx += 256) {
    const int c = idx >> 7, r = idx & 127;
    tile[(size_t)c * ld + r] = (r >= c) ? A[r * LDA_P + c] : T(0);
  }
}

// --------------------------------------------------------------------------------------------------
// k_trsm: L(rt, k) = S(rt, k) L(k,k)^-T by block forward substitution over the eight 16-column
// blocks (a3 "trsm_panel" and, for the extra tiles, a8 "trsm_var").  grid (row tiles below k, batch)
// --------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_trsm(FitArgs p, int k) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *Xs = reinterpret_cast<T *>(smem_raw);  // Xs[r * LDA_P + c]
  const int b = blockIdx.y, tid = threadIdx.x;
  const int rt = row_tile_of(blockIdx.x, k + 1, p.NT, p.rows_from_extra);
  T *Lw = reinterpret_cast<T *>(p.Lw) + (size_t)b * p.lw_stride;
  const int ld = p.ld;
  T *tile = Lw + (size_t)(k * TS) * ld + (size_t)rt * TS;
  const T *Lkk = Lw + (size_t)(k * TS) * ld + (size_t)k * TS;  // L(k,k)[c][q] at Lkk[q*ld + c]
  const T *Di = reinterpret_cast<const T *>(p.Dinv) + (size_t)b * p.dinv_stride + (size_t)k * 8 * (DB * DB);
  for (int idx = tid; idx < TS * TS; idx += 256) {
    const int c = idx >> 7, r = idx & 127;
    Xs[r * LDA_P + c] = tile[(size_t)c * ld + r];
  }
  __syncthreads();
  const int r = tid & 127;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 7);  // wave-uniform half: columns h*8 .. h*8+7
  T *row = Xs + r * LDA_P;
  for (int cb = 0; cb < TS / DB; ++cb) {
    const int c0 = cb * DB + h * 8;
    T t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = row[c0 + u];
    for (int q = 0; q < cb * DB; ++q) {
      const T xq = row[q];
      const T *lq = Lkk + (size_t)q * ld + c0;
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] -= xq * lq[u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) row[c0 + u] = t[u];
    __syncthreads();
    T tt[DB];
#pragma unroll
    for (int q = 0; q < DB; ++q) tt[q] = row[cb * DB + q];
    T x[8];
    const T *dcb = Di + cb * (DB * DB);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = h * 8 + u;
      T s = 0;
#pragma unroll
      for (int q = 0; q < DB; ++q)
        if (q <= c) s += tt[q] * dcb[c * DB + q];
      x[u] = s;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) row[c0 + u] = x[u];
    __syncthreads();
  }
  for (int idx = tid; idx < TS * TS; idx += 256) {
    const int c = idx >> 7, rr = idx & 127;
    tile[(size_t)c * ld + rr] = Xs[rr * LDA_P + c];
  }
}

// --------------------------------------------------------------------------------------------------
// k_finalize: mean_m = V_m . z ; var_m = k** - |V_m|^2 (clip 1e-15, + sigma_n^2) ;
// logML = -0.5 z'z - sum log L_ii - N/2 log 2pi   (a5 z-part, a6, a8).  grid (ceil(M/64)+1, batch):
// the last block of each fit does the scalar reductions.
// --------------------------------------------------------------------------------------------------
template <typename T>
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
  const int nmb = (M + 63) / 64;
  if ((int)blockIdx.x < nmb) {
    const int ml = tid & 63, g = tid >> 6;
    const int m = blockIdx.x * 64 + ml;
    double smu = 0, sq = 0;
    if (m < M) {
      for (int c = g; c < NP; c += 4) {
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
      const double mu = (red[0][ml] + red[0][64 + ml]) + (red[0][128 + ml] + red[0][192 + ml]);
      const double q = (red[1][ml] + red[1][64 + ml]) + (red[1][128 + ml] + red[1][192 + ml]);
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
// of the factor panel.  Per 128-tile (last to first): rhs = z - L(below,tile)^T alpha(below) by one
// wave per column with a shuffle reduction, then a 128-step in-tile back substitution.
// --------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_alpha(FitArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *al = reinterpret_cast<double *>(smem_raw);  // [NT*128]
  double *rhs = al + p.NT * TS;                        // [128]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T *Lw = reinterpret_cast<const T *>(p.Lw) + (size_t)b * p.lw_stride;
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
    for (int c = TS - 1; c >= 0; --c) {
      // alpha_c = rhs_c / L_cc ; then rhs_q -= L[c][q] alpha_c for q < c   (L[c][q] at col q, row c0+c)
      const double ac = rhs[c] / (double)Lw[(size_t)(c0 + c) * ld + c0 + c];
      if (tid == 0) al[c0 + c] = ac;
      if (tid < c) rhs[tid] -= (double)Lw[(size_t)(c0 + tid) * ld + c0 + c] * ac;
      __syncthreads();
    }
  }
  T *out = reinterpret_cast<T *>(p.alpha) + (size_t)b * p.alpha_stride;
  for (int i = tid; i < NP; i += 256) out[i] = (T)al[i];
}

}  // namespace cgp
