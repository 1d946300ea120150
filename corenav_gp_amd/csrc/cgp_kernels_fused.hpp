// cgp_kernels_fused.hpp -- the schedules the engine runs by default (DESIGN.md sections 2 and 4):
// throughput (more than 16 fits per call in fp64, 24 in fp32), two launches per block step k
// (one when the diagonal tile rides in the panel launch: fp32, and fp64 below 512 fits per call)
//   k_diag_lean(k) / k_diag(k) : one workgroup per fit.  S(k,k) = G(k,k) - sum_j L(k,j) L(k,j)^T on a
//                triangular MFMA loop, then potf2 + inverse of that tile without leaving the CU (L(k,k),
//                W_k written).  _lean: packed lower-triangle factorisation in 78 KB of LDS, two per CU.
//   k_panel(k) : every row tile below k and every extra tile.  S(i,k) as above, kept in the MFMA
//                accumulators, multiplied by W_k^T IN REGISTERS and stored once as L(i,k): the S
//                tile never goes to HBM and there is no separate trsm/trmm launch.
// latency (up to 16 / 24 fits per call)
//   k_tile_sk(k) + k_trmm_sk(k) : diagonal and panel tiles of a step in one launch, inner dimension
//                split over several workgroups with a ticketed, fixed-order reduction.
// and k_grad (hyper-parameter gradients).
// The in-register product works because the C/D layout of v_mfma_*_16x16x4 is also a valid B-operand
// layout: accumulator register r of lane (n = lane&15, kq = lane>>4) holds S[row n][col drow(lane,r)],
// which is B[k = kq][n] of an MFMA whose four k-values are the columns {drow(lane', r)}; the A operand
// supplies W[c][same columns].  Waves are laid out 4 x 1 (32 rows x 128 columns each) so every
// column block a wave needs for the triangular product is in its own registers.
#pragma once
#include "cgp_kernels.hpp"

namespace cgp {

#ifndef CGP_SKIP_DEAD_WAVE
#define CGP_SKIP_DEAD_WAVE 1   // fp32 full-batch tiles: a wave whose rows are all padding issues no MFMA (round 5, same box, three alternations:
                               // 102.2-102.5 k -> 103.5-104.1 k fits/s at 512 x N = 1024, k_panel 0.676 -> 0.688 of peak; `make variant EXTRA=-DCGP_SKIP_DEAD_WAVE=0`)
#endif
constexpr bool kSkipDeadWave = CGP_SKIP_DEAD_WAVE != 0;

// acc[cb][j][reg] = C[row = wave*32 + 2*(lane&15) + j][col = cb*16 + drow(lane, reg)].
// Rows are interleaved (2*l15 + j) so one lane owns two adjacent rows: 16-byte stores, and the two
// row fragments of a k-step come out of one 16-byte LDS read.
// Chunk 0 of both panels into LDS buffer 0 (issued before the Gram phase so it lands under it).
template <typename T>
__device__ __forceinline__ void stage_first_chunk(const T *gR, size_t ldR, const T *gC, size_t ldC, T *smem, int tid) {
  constexpr int CH = KT * LDST;
  if constexpr (sizeof(T) == 8) {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int i = 0; i < KT / 4; ++i) {
      const int col = wave * (KT / 4) + i;
      __builtin_amdgcn_global_load_lds((gbl_void *)(gR + (size_t)col * ldR + lane * 2), (lds_void *)(smem + col * LDST), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void *)(gC + (size_t)col * ldC + lane * 2), (lds_void *)(smem + CH + col * LDST), 16, 0, 0);
    }
  } else {
    using vec8 = T __attribute__((ext_vector_type(8)));
    const int sc = tid >> 4, sr = (tid & 15) * 8;
    *reinterpret_cast<vec8 *>(smem + sc * LDST + sr) = *reinterpret_cast<const vec8 *>(gR + sr + (size_t)sc * ldR);
    *reinterpret_cast<vec8 *>(smem + CH + sc * LDST + sr) = *reinterpret_cast<const vec8 *>(gC + sr + (size_t)sc * ldC);
  }
}

// zs / ms (optional): z of the newest block column (the last 8 chunks) and the running sums
// ms[j] += V z, ms[2 + j] += V^2 for rows 2 l15 + j, as in rdirect_step.
// live = false (wave-uniform): this wave's 32 rows are padding beyond the last extra row (CGP_SKIP_DEAD_WAVE) -- it stages and
// keeps the barriers, and issues no MFMA.
template <typename T, bool PRESTAGED = false>
__device__ __forceinline__ void mfma_rowpanel_loop(typename Prec<T>::acc_t (&acc)[NCB][2], const T *gR, size_t ldR,
                                                   const T *gC, size_t ldC, int nchunk, T *smem, int tid,
                                                   const T *zs = nullptr, T *ms = nullptr, bool live = true) {
  using P = Prec<T>;
  using vec8 = T __attribute__((ext_vector_type(8)));
  using vec2 = T __attribute__((ext_vector_type(2)));
  constexpr int CH = KT * LDST;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;

  auto compute = [&](const T *cur, int c) {
    if (!live) return;
    const T *zc = (zs && c >= nchunk - TS / KT) ? zs + (c - (nchunk - TS / KT)) * KT : nullptr;  // block-uniform
#pragma unroll
    for (int ks = 0; ks < KT / 4; ++ks) {
      T fa[NCB];
      const T *ra = cur + CH + (ks * 4 + lq) * LDST + l15;  // column panel -> result columns
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) fa[cb] = ra[cb * DB];
      const vec2 fb = *reinterpret_cast<const vec2 *>(cur + (ks * 4 + lq) * LDST + wave * 32 + 2 * l15);
      if (zc) {
        const T zv = zc[ks * 4 + lq];
        ms[0] = __builtin_fma(fb[0], zv, ms[0]);
        ms[1] = __builtin_fma(fb[1], zv, ms[1]);
        ms[2] = __builtin_fma(fb[0], fb[0], ms[2]);
        ms[3] = __builtin_fma(fb[1], fb[1], ms[3]);
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        acc[cb][0] = P::mfma(fa[cb], fb[0], acc[cb][0]);
        acc[cb][1] = P::mfma(fa[cb], fb[1], acc[cb][1]);
      }
    }
  };

  if constexpr (sizeof(T) == 8) {
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
    if (nchunk > 0 && !PRESTAGED) stage(smem, 0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
      if (c + 1 < nchunk) stage(smem + ((c + 1) & 1) * 2 * CH, c + 1);
      compute(smem + (c & 1) * 2 * CH, c);
      __syncthreads();
    }
  } else {
    const int sc = tid >> 4, sr = (tid & 15) * 8;
    gR += sr;
    gC += sr;
    vec8 pr, pc;
    if (nchunk > 0 && !PRESTAGED) {
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

// --------------------------------------------------------------------------------------------------
// fp32 products on the bf16 matrix cores (CGP_F32_BF16X6, default on: the tile loops of the full-batch and mid-size fp32 builds).
// gfx950 runs an fp32-input MFMA at the VECTOR rate -- 1/16 of the bf16 MFMA rate -- and has no xf32 form.  An fp32 value splits
// EXACTLY into three bf16 values by truncation, x = x0 + x1 + x2 (8 + 8 + 8 mantissa bits, same exponent range), and
//     a b = a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0) + O(2^-24 |a b|)
// so SIX bf16 products with fp32 accumulation reproduce the fp32 product to fp32's own rounding level (every partial product of two
// 8-bit mantissas is exact in fp32; what is dropped -- a1 b2, a2 b1, a2 b2 -- is at most 2^-21, typically 2^-23 of |a b|: a 16-term
// inner product comes out within 1.0-1.5 x the error of sixteen chained fp32 roundings, tests/test_bf16x6_split_cpu.py).  Terms are
// added smallest first.  The split happens once, when a chunk is staged: three bf16 planes per panel in LDS (bx_pos: unpadded 32-byte rows, halves
// swizzled); bx6_compute pairs the terms into three K = 32 MFMAs per 16 x 16 block (48 cycles against the fp32 form's 128).
// Measured (round 5, configs[3], 512 fits): 104 k -> 123 k fits/s; the 64-fit call 0.89 -> 0.73 ms.  What bounds the loops now is
// HBM: a 128 x 128 tile reads two 128 x K panels for 2 x 128 x 128 x K flops -- 32 flop per byte, 64 with the column panel served by
// L2 -- and the panel launches run at 3.2-3.3 TB/s (profiles/r05_pmc_summary_f32.json) of the ~6.3 TB/s a streaming kernel reaches:
// a second register set, an L2 prefetch three chunks ahead, a second plane buffer and the row panel loaded straight into B-operand
// registers were each measured and none was faster (docs/negatives.md, round 5).
// --------------------------------------------------------------------------------------------------
#ifndef CGP_F32_BF16X6
#define CGP_F32_BF16X6 1
#endif
constexpr bool kF32Bf16x6 = CGP_F32_BF16X6 != 0;
#ifndef CGP_BX_MID_SETS
#define CGP_BX_MID_SETS 4
#endif
constexpr int kBxMidSets = CGP_BX_MID_SETS;   // chunks in flight (register sets) of the mid-size build's bf16-plane loop
#ifndef CGP_BX_TRMM
#define CGP_BX_TRMM 1
#endif
constexpr bool kBxTrmm = CGP_BX_TRMM != 0;    // the in-register triangular product L = S W^T of the fp32 panel tiles on the bf16 matrix cores too
#ifndef CGP_BX_TRI
#define CGP_BX_TRI 1
#endif
constexpr bool kBxTri = CGP_BX_TRI != 0;      // the diagonal-tile update of the deep fp32 builds on the bf16 matrix cores too
constexpr int BXS = 16;                      // bf16 elements per staged row (32 bytes, no padding: the two 16-byte halves are swizzled instead)
constexpr int BX_PLANE = TS * BXS;           // bf16 elements of one plane of one panel
constexpr int BX_FLOATS = 6 * BX_PLANE / 2;  // floats the six planes occupy
typedef unsigned bxu4 __attribute__((ext_vector_type(4)));
// Where the 8 k-values `half` (0: k = 0 .. 7, 1: k = 8 .. 15) of staged row `slot` live, in bf16 elements from the plane's start.
// A lane's operand is one 16-byte read; ds_read_b128 is served in four groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31} and the same + 32: MI355X_MICROARCH.md, LDS) which between them take rows l15 = 0 .. 15 with half 0 for eight of them and
// half 1 for the other eight: the halves of a row swap places with bit 2 of the row so that such a group covers all 64 banks
// once, and with bit 4 so that the eight consecutive rows a ds_write_b128 group stores cover the 32 store banks once, also for
// the row panel, whose rows are stored in the order the accumulator layout reads them (bx_row_slot).
__device__ __forceinline__ int bx_pos(int slot, int half) { return slot * BXS + 8 * ((half ^ (slot >> 2) ^ (slot >> 4)) & 1); }
// row panel: a lane's two rows are 2 l15 + {0, 1} of its wave's 32; stored as slot 16 j + l15 so that one read's rows are consecutive
__device__ __forceinline__ int bx_row_slot(int r) { return (r & ~31) | ((r & 1) << 4) | ((r >> 1) & 15); }

// eight floats (consecutive k of one row) -> three planes of eight bf16 each, written as 16-byte rows.  Split by truncation:
// x0 = the top 16 bits of x, x1 = the top 16 bits of x - x0, x2 = those of x - x0 - x1 (both differences exact); v_perm_b32 packs
// the high halves of two words.
#ifndef CGP_BX_RNE
#define CGP_BX_RNE 0
#endif
typedef __bf16 bxbf2 __attribute__((ext_vector_type(2)));
typedef float bxf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bx_split8(const float (&x)[8], bxu4 (&w)[3]) {
#if CGP_BX_RNE
  // split by rounding to nearest (v_cvt_pk_bf16_f32 rounds and packs a pair): the residuals are signed, x - x0 - x1 still has at
  // most 8 significant bits, and the three products the 6-term form drops (x1 y2, x2 y1, x2 y2) are 8 x smaller and of either sign
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const bxf2 a = {x[i], x[i + 1]};
    const unsigned p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bxbf2));
    const bxf2 ra = {a.x - __uint_as_float(p0 << 16), a.y - __uint_as_float(p0 & 0xffff0000u)};
    const unsigned p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(ra, bxbf2));
    const bxf2 rb = {ra.x - __uint_as_float(p1 << 16), ra.y - __uint_as_float(p1 & 0xffff0000u)};
    w[0][i >> 1] = p0;
    w[1][i >> 1] = p1;
    w[2][i >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(rb, bxbf2));
  }
  return;
#endif
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    unsigned hi[2][3];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float a = x[i + u];
      const unsigned a0 = __float_as_uint(a) & 0xffff0000u;
      const float ra = a - __uint_as_float(a0);             // exact: a0 is a prefix of a's mantissa
      const unsigned a1 = __float_as_uint(ra) & 0xffff0000u;
      const float rb = ra - __uint_as_float(a1);            // exact; at most 8 significant bits are left
      hi[u][0] = a0;
      hi[u][1] = a1;
      hi[u][2] = __float_as_uint(rb);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) w[pl][i >> 1] = __builtin_amdgcn_perm(hi[1][pl], hi[0][pl], 0x07060302u);   // {hi[1] >> 16, hi[0] >> 16}
  }
}
__device__ __forceinline__ void bx_split_store(const float (&x)[8], unsigned short *plane0, int off) {
  bxu4 w[3];
  bx_split8(x, w);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bxu4 *>(plane0 + pl * BX_PLANE + off) = w[pl];
}
template <int D> struct BxStage {   // what a thread stages per chunk: row (tid & 127) of both panels, columns 8 (tid >> 7) .. + 7; D chunks in flight
  // buffer loads: the row of the chunk goes into the scalar offset, the lane's place in it is a constant VGPR -- no per-load
  // address arithmetic on the VALU, which the split already fills
  __amdgpu_buffer_rsrc_t rR, rC;
  int ldR4, ldC4;         // leading dimensions in bytes
  int vR, vC;             // this lane's byte offset inside a chunk
  int offR, offC;         // where its 8 values go in a plane
  float xr[D][8], xc[D][8];
  float pz = 0.f, pv = 0.f;   // running V z and V^2 of this thread's row over its 8 of every 16 columns (newest block column only)
  __device__ __forceinline__ void init(const float *gR, size_t ldR, const float *gC, size_t ldC, int nchunk, int tid) {
    const int r = tid & (TS - 1), h = tid >> 7;
    const int bytesR = (int)(((size_t)nchunk * KT * ldR) * sizeof(float)), bytesC = (int)(((size_t)nchunk * KT * ldC) * sizeof(float));
    rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(gR), 0, bytesR, 0x00020000);
    rC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(gC), 0, bytesC, 0x00020000);
    ldR4 = (int)ldR * 4;
    ldC4 = (int)ldC * 4;
    vR = 4 * r + 8 * h * ldR4;
    vC = 4 * r + 8 * h * ldC4;
    offR = bx_pos(bx_row_slot(r), h);
    offC = bx_pos(r, h);
  }
  template <int S> __device__ __forceinline__ void load(int chunk) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      xr[S][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rR, vR, (chunk * KT + i) * ldR4, 0));
      xc[S][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rC, vC, (chunk * KT + i) * ldC4, 0));
    }
  }
  template <int S> __device__ __forceinline__ void store(float *planes, const float *z8 = nullptr) {
    unsigned short *sp = reinterpret_cast<unsigned short *>(planes);
    if (z8) {   // wave-uniform address: one broadcast read
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        pz = __builtin_fmaf(xr[S][i], z8[i], pz);
        pv = __builtin_fmaf(xr[S][i], xr[S][i], pv);
      }
    }
    bx_split_store(xr[S], sp, offR);
#if defined(CGP_PROBE_COLSPLIT) && CGP_PROBE_COLSPLIT == 0
    // timing probe (WRONG results; `make variant`): the column panel's split and plane stores are skipped -- the bound of what sharing
    // the column-panel stage between row tiles (a 256 x 128 row-pair tile, or more) could save
    if (xc[S][0] == 12345.678f) bx_split_store(xc[S], sp + 3 * BX_PLANE, offC);
#elif defined(CGP_PROBE_COLSPLIT) && CGP_PROBE_COLSPLIT == 2
    if ((threadIdx.x & 64) == 0) bx_split_store(xc[S], sp + 3 * BX_PLANE, offC);   // probe: half of it (what a row PAIR saves per tile)
#else
    bx_split_store(xc[S], sp + 3 * BX_PLANE, offC);
#endif
  }
};
// One chunk of products out of the planes.  The full-rate bf16 MFMA of gfx950 is the K = 32 form (v_mfma_f32_16x16x32_bf16: 16
// cycles for 16 k-values more than the legacy K = 16 form takes for 16), and a chunk has 16 columns -- so TWO terms share one
// instruction, concatenated along K: lanes 0-31 (k = 0 .. 15) carry one plane pair, lanes 32-63 (k = 16 .. 31) the other:
//     [a0 | a2] . [b2 | b0] = a0 b2 + a2 b0        [a0 | a1] . [b1 | b0] = a0 b1 + a1 b0        [a0 | a1] . [b0 | b1] = a0 b0 + a1 b1
// three MFMAs (48 cycles) per 16 x 16 block and chunk against the fp32 form's four (128 cycles), smallest terms first.  A lane's
// operand is 8 consecutive bf16 of ONE plane (which one depends on its half of the wave): one 16-byte LDS read.
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf8 bx_ld8(const unsigned short *p) { return __builtin_bit_cast(bf8, *reinterpret_cast<const bxu4 *>(p)); }
// pa[cb & 1] + cb DB BXS / pb[j]: this lane's 8 k-values of plane 0, column block cb / row block j (the swizzle depends on the parity
// of the 16-row block); hi = lq >> 1 selects the second plane of a pair
__device__ __forceinline__ void bx6_compute(Prec<float>::acc_t (&acc)[NCB][2], const unsigned short *const (&pa)[2], const unsigned short *const (&pb)[2],
                                            int hi) {
  const int p01 = hi ? BX_PLANE : 0, p02 = hi ? 2 * BX_PLANE : 0;          // [x0 | x1], [x0 | x2]
  const int p10 = hi ? 0 : BX_PLANE, p20 = hi ? 0 : 2 * BX_PLANE;          // [x1 | x0], [x2 | x0]
  // pass 1: the smallest pair of terms, [a0 | a2] . [b2 | b0]; pass 2: [a0 | a1] . [b1 | b0] and [a0 | a1] . [b0 | b1].  Two passes
  // so that only the B operands of a pass are live.  The A operands are read one pair of column blocks AHEAD of the pair being
  // multiplied (the mid-size build has one wave per SIMD: nobody else covers an LDS read issued right in front of its MFMA).
  bf8 a[2][2];
  a[0][0] = bx_ld8(pa[0] + p02);
  a[0][1] = bx_ld8(pa[1] + p02 + DB * BXS);
  const bf8 b20[2] = {bx_ld8(pb[0] + p20), bx_ld8(pb[1] + p20)};
  bf8 b10[2], b01[2];
#pragma unroll
  for (int cb = 0; cb < NCB; cb += 2) {
    const int u = (cb >> 1) & 1;
    if (cb + 2 < NCB) {
      a[u ^ 1][0] = bx_ld8(pa[0] + p02 + (cb + 2) * DB * BXS);
      a[u ^ 1][1] = bx_ld8(pa[1] + p02 + (cb + 3) * DB * BXS);
    } else {   // the first pair of pass 2 and its B operands
      a[u ^ 1][0] = bx_ld8(pa[0] + p01);
      a[u ^ 1][1] = bx_ld8(pa[1] + p01 + DB * BXS);
      b10[0] = bx_ld8(pb[0] + p10);
      b10[1] = bx_ld8(pb[1] + p10);
      b01[0] = bx_ld8(pb[0] + p01);
      b01[1] = bx_ld8(pb[1] + p01);
    }
    acc[cb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][0], b20[0], acc[cb][0], 0, 0, 0);
    acc[cb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][0], b20[1], acc[cb][1], 0, 0, 0);
    acc[cb + 1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][1], b20[0], acc[cb + 1][0], 0, 0, 0);
    acc[cb + 1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][1], b20[1], acc[cb + 1][1], 0, 0, 0);
  }
#pragma unroll
  for (int cb = 0; cb < NCB; cb += 2) {
    const int u = (cb >> 1) & 1;   // (pass 1 left the first pair of pass 2 in a[0])
    if (cb + 2 < NCB) {
      a[u ^ 1][0] = bx_ld8(pa[0] + p01 + (cb + 2) * DB * BXS);
      a[u ^ 1][1] = bx_ld8(pa[1] + p01 + (cb + 3) * DB * BXS);
    }
    acc[cb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][0], b10[0], acc[cb][0], 0, 0, 0);
    acc[cb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][0], b10[1], acc[cb][1], 0, 0, 0);
    acc[cb + 1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][1], b10[0], acc[cb + 1][0], 0, 0, 0);
    acc[cb + 1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][1], b10[1], acc[cb + 1][1], 0, 0, 0);
    acc[cb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][0], b01[0], acc[cb][0], 0, 0, 0);
    acc[cb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][0], b01[1], acc[cb][1], 0, 0, 0);
    acc[cb + 1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][1], b01[0], acc[cb + 1][0], 0, 0, 0);
    acc[cb + 1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][1], b01[1], acc[cb + 1][1], 0, 0, 0);
  }
}
// The same products for a workgroup that has its CU to itself (mid-size build: one wave per SIMD, nobody else to cover an LDS read
// issued right in front of its MFMA -- which is where the compiler puts every one of them, re-using one register quad).  Reads and
// waits are written out: up to three pairs of A operands in flight, `s_waitcnt lgkmcnt` counted by hand (LDS reads return in
// order; the "+v" operands tie each wait to the MFMAs that consume what it waited for).
__device__ __forceinline__ bxu4 bx_rd(unsigned addr, int off) {
  bxu4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
  return v;
}
__device__ __forceinline__ void bx_tie(bxu4 &x) { asm volatile("" : "+v"(x)); }   // a use of x stays behind the (volatile) wait in front of this
#define BX_WAIT2(n, x, y) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(n))
#define BX_WAIT4(n, x, y, z, w) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "n"(n))
#define BX_WAIT6(n, x, y, z, w, u, t) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(x), "+v"(y), "+v"(z), "+v"(w), "+v"(u), "+v"(t) : "n"(n))
__device__ __forceinline__ void bx_mm4(Prec<float>::acc_t (&acc)[NCB][2], int cb, const bxu4 &a0, const bxu4 &a1, const bxu4 &b0, const bxu4 &b1) {
  const bf8 x0 = __builtin_bit_cast(bf8, a0), x1 = __builtin_bit_cast(bf8, a1), y0 = __builtin_bit_cast(bf8, b0), y1 = __builtin_bit_cast(bf8, b1);
  acc[cb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0, y0, acc[cb][0], 0, 0, 0);
  acc[cb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0, y1, acc[cb][1], 0, 0, 0);
  acc[cb + 1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, y0, acc[cb + 1][0], 0, 0, 0);
  acc[cb + 1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, y1, acc[cb + 1][1], 0, 0, 0);
}
template <int BO>   // BO: byte offset of the plane buffer
__device__ __forceinline__ void bx6_compute_pipe(Prec<float>::acc_t (&acc)[NCB][2], const unsigned (&la)[2], const unsigned (&lb)[2], int hi) {
  constexpr int PB = 2 * BX_PLANE, CBB = 2 * DB * BXS;   // bytes of a plane, of a column block
  // the plane a lane reads is part of its address: [x0 | x2] / [x0 | x1] for A, [x2 | x0] / [x1 | x0] / [x0 | x1] for B
  const unsigned a2[2] = {la[0] + (hi ? 2 * PB : 0), la[1] + (hi ? 2 * PB : 0)}, a1[2] = {la[0] + (hi ? PB : 0), la[1] + (hi ? PB : 0)};
  const unsigned q20[2] = {lb[0] + (hi ? 0 : 2 * PB), lb[1] + (hi ? 0 : 2 * PB)}, q10[2] = {lb[0] + (hi ? 0 : PB), lb[1] + (hi ? 0 : PB)};
  const unsigned q01[2] = {lb[0] + (hi ? PB : 0), lb[1] + (hi ? PB : 0)};
  bxu4 b20a = bx_rd(q20[0], BO), b20b = bx_rd(q20[1], BO);
  bxu4 p0 = bx_rd(a2[0], BO + 0 * CBB), p1 = bx_rd(a2[1], BO + 1 * CBB);
  bxu4 p2 = bx_rd(a2[0], BO + 2 * CBB), p3 = bx_rd(a2[1], BO + 3 * CBB);
  bxu4 p4 = bx_rd(a2[0], BO + 4 * CBB), p5 = bx_rd(a2[1], BO + 5 * CBB);
  BX_WAIT4(4, b20a, b20b, p0, p1);
  bx_mm4(acc, 0, p0, p1, b20a, b20b);
  bxu4 p6 = bx_rd(a2[0], BO + 6 * CBB), p7 = bx_rd(a2[1], BO + 7 * CBB);
  BX_WAIT2(4, p2, p3);
  bx_mm4(acc, 2, p2, p3, b20a, b20b);
  bxu4 b10a = bx_rd(q10[0], BO), b10b = bx_rd(q10[1], BO), b01a = bx_rd(q01[0], BO), b01b = bx_rd(q01[1], BO);
  bxu4 r0 = bx_rd(a1[0], BO + 0 * CBB), r1 = bx_rd(a1[1], BO + 1 * CBB);
  BX_WAIT2(8, p4, p5);
  bx_mm4(acc, 4, p4, p5, b20a, b20b);
  bxu4 r2 = bx_rd(a1[0], BO + 2 * CBB), r3 = bx_rd(a1[1], BO + 3 * CBB);
  BX_WAIT2(8, p6, p7);
  bx_mm4(acc, 6, p6, p7, b20a, b20b);
  bxu4 r4 = bx_rd(a1[0], BO + 4 * CBB), r5 = bx_rd(a1[1], BO + 5 * CBB);
  BX_WAIT6(4, b10a, b10b, b01a, b01b, r0, r1);
  bx_mm4(acc, 0, r0, r1, b10a, b10b);
  bx_mm4(acc, 0, r0, r1, b01a, b01b);
  bxu4 r6 = bx_rd(a1[0], BO + 6 * CBB), r7 = bx_rd(a1[1], BO + 7 * CBB);
  BX_WAIT2(4, r2, r3);
  bx_mm4(acc, 2, r2, r3, b10a, b10b);
  bx_mm4(acc, 2, r2, r3, b01a, b01b);
  BX_WAIT2(2, r4, r5);
  bx_mm4(acc, 4, r4, r5, b10a, b10b);
  bx_mm4(acc, 4, r4, r5, b01a, b01b);
  BX_WAIT2(0, r6, r7);
  bx_mm4(acc, 6, r6, r7, b10a, b10b);
  bx_mm4(acc, 6, r6, r7, b01a, b01b);
}
// `st` arrives with chunks 0 .. D - 1 in flight (bx6_prologue, before the Gram phase); set c % D holds chunk c.  D = 1, one plane
// buffer, two barriers per chunk: the full-batch build (four workgroups per CU at 128 VGPRs cover the rest of the latency; a second
// set spilled even at three per CU -- 168 VGPRs -- and a spill's scratch traffic drains the load queue at every reload; a second plane buffer measured the
// same).  D = 4, two plane buffers, one barrier per chunk: the mid-size build, whose CUs hold one or two workgroups (chunk c + 1 is
// split into the buffer chunk c - 1 was read from, which every wave left before the barrier of iteration c; the Gram inputs lie
// over buffer 1, first overwritten after the barrier of iteration 0).
template <int D> __device__ __forceinline__ void bx6_prologue(BxStage<D> &st, int nchunk) {
  static_assert(D == 1 || D == 2 || D == 4, "register sets");
  st.template load<0>(0);
  if (D > 1) st.template load<1 % D>(1);
  if (D > 2) st.template load<2 % D>(2);
  if (D > 2) st.template load<3 % D>(3);
}
template <int D, bool DBL, int S>   // S = c % D: the set the split of chunk c emptied takes chunk c + D
__device__ __forceinline__ void bx6_iter(Prec<float>::acc_t (&acc)[NCB][2], BxStage<D> &st, int c, int nchunk, float *smem, const unsigned short *const (&pa)[2],
                                         const unsigned short *const (&pb)[2], int hi, const float *zs, int zfirst, int h8, bool live) {
  // UNCONDITIONAL.  Past the last chunk the loads are harmless either way: the chunk's row goes into the SCALAR offset, which the
  // descriptor's bounds check may or may not cover (the range check is defined on the vector offset), so what makes the look-ahead
  // safe is the LAYOUT, not the descriptor: a look-ahead reaches at most D chunks = 64 columns past column 128 k of a panel that has
  // NT x 128 >= 128 (k + 1) columns inside the same fit's slab (run_schedule asserts k < NT; cgp_create sizes the slab by NTmax), the
  // values are never stored, and where the check does fire they are zeros without a memory access.  A load behind a branch
  // is one the compiler cannot count on when it computes the vmcnt of an OLDER load's wait -- every wait then drained the queue
  // (vmcnt(0) at each split: one chunk in flight however many sets there were).
  st.template load<S>(c + D);
  lds_barrier();                         // chunk c is in its planes (DBL: and chunk c - 1 is done with)
  if constexpr (DBL) {   // (the set index and the buffer of a chunk have the same parity: D is even)
    static_assert(!DBL || D % 2 == 0, "plane buffer by set parity");
    const unsigned la[2] = {(unsigned)(size_t)pa[0], (unsigned)(size_t)pa[1]}, lb[2] = {(unsigned)(size_t)pb[0], (unsigned)(size_t)pb[1]};
    if (live) bx6_compute_pipe<(S & 1) * 12 * BX_PLANE>(acc, la, lb, hi);
  } else if (live) bx6_compute(acc, pa, pb, hi);
  if (!DBL) lds_barrier();               // every wave is done with it
  if (c + 1 < nchunk)
    st.template store<(S + 1) % D>(smem + (DBL ? ((c + 1) & 1) * BX_FLOATS : 0), (zs && c + 1 >= zfirst) ? zs + (c + 1 - zfirst) * KT + h8 : nullptr);
}
template <int D, bool DBL>
__device__ __forceinline__ void bx6_loop(Prec<float>::acc_t (&acc)[NCB][2], BxStage<D> &st, int nchunk, float *smem, int tid, const float *zs, float *ms,
                                         bool live) {
  if (nchunk <= 0) return;
  const unsigned short *sp = reinterpret_cast<const unsigned short *>(smem);
  const int lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lq >> 1;
  // this lane's fragments of plane 0: row blocks j = 0, 1 of the wave's slab; column blocks of either parity, less the block's own offset
  const unsigned short *const pb[2] = {sp + bx_pos(wave * 32 + l15, lq & 1), sp + bx_pos(wave * 32 + DB + l15, lq & 1)};
  const unsigned short *const pa[2] = {sp + 3 * BX_PLANE + bx_pos(l15, lq & 1), sp + 3 * BX_PLANE + bx_pos(DB + l15, lq & 1) - DB * BXS};
  const int h8 = 8 * __builtin_amdgcn_readfirstlane(tid >> 7), zfirst = nchunk - TS / KT;   // the newest block column = the last 8 chunks
  st.template store<0>(smem, (zs && zfirst <= 0) ? zs + (0 - zfirst) * KT + h8 : nullptr);   // chunk 0
  for (int c0 = 0; c0 < nchunk; c0 += D) {   // nchunk is a multiple of 8 (whole block columns): the body is straight-line code
    bx6_iter<D, DBL, 0>(acc, st, c0, nchunk, smem, pa, pb, hi, zs, zfirst, h8, live);
    if (D > 1) bx6_iter<D, DBL, 1 % D>(acc, st, c0 + 1, nchunk, smem, pa, pb, hi, zs, zfirst, h8, live);
    if (D > 2) bx6_iter<D, DBL, 2 % D>(acc, st, c0 + 2, nchunk, smem, pa, pb, hi, zs, zfirst, h8, live);
    if (D > 2) bx6_iter<D, DBL, 3 % D>(acc, st, c0 + 3, nchunk, smem, pa, pb, hi, zs, zfirst, h8, live);
  }
  __syncthreads();
  if (zs) {   // hand the row sums over in the layout the fp32 loop leaves them in: lanes 0-15 of wave w, rows 32 w + 2 lane + {0, 1}
    smem[tid] = st.pz;
    smem[2 * TS + tid] = st.pv;
    __syncthreads();
    if (lane < 16) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = wave * 32 + 2 * lane + j;
        ms[j] = smem[r] + smem[TS + r];
        ms[2 + j] = smem[2 * TS + r] + smem[3 * TS + r];
      }
    }
    __syncthreads();
  }
}

// --------------------------------------------------------------------------------------------------
// The mid-size build's tile loop on panels that arrive ALREADY SPLIT (round 6).  Inside a call that leaves a CU one or two
// workgroups a tile's duration is its loop's latency, and the register-staged split above costs a wave as many issue cycles as the
// products themselves (16 loads, ~90 VALU, 6 ds_write per chunk against 48 MFMAs: 3 000 ticks per chunk measured, 768 of them MFMA).
// So the workgroup that PRODUCES a tile of L writes it twice: as fp32 (everything else reads that) and as three bf16 planes,
//     Lp[plane][16-column chunk][row][16]        (one chunk of one tile = 4 KB contiguous per plane)
// and a consumer's staging is six LDS-DMA instructions per wave and chunk -- no VGPR, no VALU, no ds_write: lane i of wave w
// fetches the 16 bytes that belong at LDS unit 64 w + i of the plane (the swizzle of bx_pos and the row order of bx_row_slot are a
// permutation INSIDE a wave's 1 KB, applied on the global side).  Three chunk buffers of 24 KB, chunks c + 1 and c + 2 in flight
// while chunk c is multiplied (hand-counted `s_waitcnt vmcnt(6)`), one barrier per chunk.  The products, their order and the planes'
// values are those of bx6_loop: results are bitwise the register-staged form's.  1.5 x the panel bytes from HBM -- which is why the
// full-batch build (HBM at 3.5 TB/s) would keep the register-staged split and only calls of at most 96 fits take this form.
// --------------------------------------------------------------------------------------------------
// MEASURED, NOT SHIPPED (round 6, docs/negatives.md): the loop itself is 22-27 % faster per tile (k = 5: 119 k -> 93 k ticks, k = 7:
// 142 k -> 104 k; 1 850 ticks per chunk) and every result is bitwise the register-staged build's, but a tile's store phase grows
// from 3 k to 10 k ticks (48 eight-byte plane stores per lane) -- on the kind-A chain too -- and the late launches of a 64-fit call
// are bound by 576-704 workgroups meeting 512 slots, not by the tile's length: 0.704 ms per call against 0.697 (two stream groups),
// 0.770 against 0.718 (one).  `make variant EXTRA=-DCGP_MID_PLANES=1` builds it (the context then allocates the planes).
#ifndef CGP_MID_PLANES
#define CGP_MID_PLANES 0
#endif
constexpr bool kMidPlanes = CGP_MID_PLANES != 0 && kF32Bf16x6;
constexpr int BXP_BUF = 6 * BX_PLANE * 2;   // bytes of one chunk buffer: six planes of 4 KB
constexpr int BXP_NBUF = 3;
struct BxPlaneSrc {
  const unsigned short *gR, *gC;   // this lane's 16 bytes of plane 0, chunk 0 of the row / column panel
  size_t chunk_stride, plane_stride;   // bf16 elements
  __device__ __forceinline__ void init(const unsigned short *planes, size_t plane_elems, int ld, int chunk0, int row_tile, int col_tile, int tid) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = 32 * wave + (lane >> 1);
    const int h = ((lane & 1) ^ (slot >> 2) ^ (slot >> 4)) & 1;                       // bx_pos: which half of the row sits at this unit
    const int rowR = (slot & ~31) | ((slot & 15) << 1) | ((slot >> 4) & 1);          // inverse of bx_row_slot
    chunk_stride = (size_t)ld * BXS;
    plane_stride = plane_elems;
    const unsigned short *base = planes + (size_t)chunk0 * chunk_stride;
    gR = base + ((size_t)row_tile * TS + rowR) * BXS + 8 * h;
    gC = base + ((size_t)col_tile * TS + slot) * BXS + 8 * h;
  }
  // chunk c of both panels -> buffer `buf` (byte offset): 6 wave-instructions of 1 KB each
  __device__ __forceinline__ void issue(int c, unsigned char *smem_bytes, int buf, int wave) const {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    const unsigned short *r = gR + (size_t)c * chunk_stride, *q = gC + (size_t)c * chunk_stride;
    unsigned char *dst = smem_bytes + buf + wave * 1024;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      __builtin_amdgcn_global_load_lds((gbl_void *)(r + pl * plane_stride), (lds_void *)(dst + pl * 2 * BX_PLANE), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void *)(q + pl * plane_stride), (lds_void *)(dst + (3 + pl) * 2 * BX_PLANE), 16, 0, 0);
    }
  }
};
// the tile's own planes, written next to store_tile's fp32 copy: lane (l15, lq) holds k = 4 lq + r of rows 2 l15 + {0, 1} of its wave's slab
__device__ __forceinline__ void store_tile_planes(const Prec<float>::acc_t (&acc)[NCB][2], unsigned short *planes, size_t plane_elems, int ld, int rt, int k,
                                                  int tid) {
  const int lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  typedef unsigned bxu2 __attribute__((ext_vector_type(2)));
  unsigned short *out = planes + ((size_t)(k * (TS / KT)) * ld + (size_t)rt * TS + wave * 32 + 2 * l15) * BXS + 4 * lq;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const float v[8] = {acc[cb][0][0], acc[cb][0][1], acc[cb][0][2], acc[cb][0][3], acc[cb][1][0], acc[cb][1][1], acc[cb][1][2], acc[cb][1][3]};
    bxu4 w[3];
    bx_split8(v, w);
    unsigned short *o = out + (size_t)cb * ld * BXS;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      *reinterpret_cast<bxu2 *>(o + pl * plane_elems) = bxu2{w[pl][0], w[pl][1]};           // row 2 l15
      *reinterpret_cast<bxu2 *>(o + pl * plane_elems + BXS) = bxu2{w[pl][2], w[pl][3]};     // row 2 l15 + 1
    }
  }
}
// zs != null (extra rows, running predictive sums): V z and V^2 over the newest block column (the last 8 chunks), from the fp32
// copy -- the planes never pass through registers.  ms as bx6_loop leaves it.
__device__ __forceinline__ void bx6p_loop(Prec<float>::acc_t (&acc)[NCB][2], const BxPlaneSrc &src, int nchunk, float *smem, int tid, const float *zs, float *ms,
                                          bool live, const float *gR32, size_t ldR32) {
  if (nchunk <= 0) return;
  unsigned char *sb = reinterpret_cast<unsigned char *>(smem);
  const int lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lq >> 1;
  const unsigned base = (unsigned)(size_t)smem;
  // this lane's fragments of plane 0 of buffer 0 (bx6_loop's pa / pb as LDS byte addresses)
  const unsigned lb0[2] = {base + 2 * bx_pos(wave * 32 + l15, lq & 1), base + 2 * bx_pos(wave * 32 + DB + l15, lq & 1)};
  const unsigned la0[2] = {base + 2 * (3 * BX_PLANE + bx_pos(l15, lq & 1)), base + 2 * (3 * BX_PLANE + bx_pos(DB + l15, lq & 1) - DB * BXS)};
  // (chunks 0 and 1 were issued by the caller before the Gram phase: bx6p_prologue)
  int buf = 0;
  for (int c = 0; c < nchunk; ++c) {
    // everything but the six DMAs of chunk c + 1 has landed, i.e. chunk c is in its buffer -- for this wave's share; the barrier
    // makes it everybody's, and says every wave is done with chunk c - 1, whose buffer chunk c + 2 takes
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    lds_barrier();
    const int nb = buf == 0 ? 2 * BXP_BUF : buf - BXP_BUF;   // buffer of chunk c + 2 = (c - 1) mod 3
    src.issue(c + 2 < nchunk ? c + 2 : nchunk - 1, sb, nb, wave);   // past the end: the last chunk again (never read), so the count holds
    if (live) {
      const unsigned la[2] = {la0[0] + buf, la0[1] + buf}, lb[2] = {lb0[0] + buf, lb0[1] + buf};
      bx6_compute_pipe<0>(acc, la, lb, hi);
    }
    buf = buf == 2 * BXP_BUF ? 0 : buf + BXP_BUF;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (zs) {
    // row r = tid & 127, columns 8 h .. 8 h + 7 of each of the newest block column's eight chunks
    const int r = tid & (TS - 1), h8 = 8 * (tid >> 7), zfirst = nchunk - TS / KT;
    float pz = 0.f, pv = 0.f;
    const float *g = gR32 + (size_t)(zfirst * KT + h8) * ldR32 + r;
#pragma unroll
    for (int c = 0; c < TS / KT; ++c) {
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = g[(size_t)(c * KT + i) * ldR32];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        pz = __builtin_fmaf(x[i], zs[c * KT + h8 + i], pz);
        pv = __builtin_fmaf(x[i], x[i], pv);
      }
    }
    smem[tid] = pz;
    smem[2 * TS + tid] = pv;
    __syncthreads();
    if (lane < 16) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rr = wave * 32 + 2 * lane + j;
        ms[j] = smem[rr] + smem[TS + rr];
        ms[2 + j] = smem[2 * TS + rr] + smem[3 * TS + rr];
      }
    }
    __syncthreads();
  }
}

// --------------------------------------------------------------------------------------------------
// fp64 panel loop with the row panel kept out of LDS.  A wave's B operand is its own 32 rows of the
// row panel and nobody else reads them, so they are loaded straight into registers (one 16-byte
// load per lane and k-step: rows 2 l15 + {0,1}, column 4 ks + lq -- 256-byte runs per 16 lanes);
// only the column panel, which all four waves share, goes through LDS (LDS-DMA, 4 chunk buffers).
// Both streams run two chunks ahead; the waits are hand-placed (`s_waitcnt vmcnt(8)`: everything but
// the 4 + 4 loads of chunk c + 1 has landed) because a __syncthreads() would drain the queue.
//   prologue (before the Gram phase):  C(0) R(0) C(1) R(1)
//   iteration c: wait, barrier, issue C(c+2) R(c+2), multiply chunk c.
// nchunk is a multiple of 4 (k * 128 / 16).  The Gram staging area (smem + 2 chunk buffers) is
// buffer 2 / 3, first written after the barrier of iteration 0, i.e. after every wave's Gram phase.
// --------------------------------------------------------------------------------------------------
// R = slots of the chunk ring (row fragments in registers, column-panel chunks in LDS), D = chunks in flight ahead of the one
// being multiplied: (4, 2).  The loop is written for any (R, D) with 8 % R == 0, D < R, (D - 1) x loads per chunk <= 63; (8, 5)
// in the fp32 mid-size build was measured in round 5 (tools/phase_clock.py had shown 3.6-4.6 k ticks per chunk inside a 64-fit
// call against 2.05 k of MFMA time): 24 / 32 / 48 / 64 / 96 fits 0.608 / 0.671 / 0.767 / 1.035 / 1.293 ms per call against 0.63 /
// 0.684 / 0.770 / 0.985 / 1.27 -- those ticks are two co-resident workgroups sharing the MFMA pipes (2 x 2.05 k), not memory latency.
template <typename T, int R = 4> struct RowFrag {
  using vec2 = T __attribute__((ext_vector_type(2)));
  vec2 r[R][KT / 4];
};
#ifndef CGP_MID_RING
#define CGP_MID_RING 4   // `make variant` A/B: 8 = ring of eight / five chunks ahead in the fp32 mid-size build (measured: no gain)
#endif
template <typename T, bool MID> constexpr int deep_ring() { return (MID && sizeof(T) == 4) ? CGP_MID_RING : 4; }
template <int R> constexpr int deep_dist() { return R == 4 ? 2 : 5; }

template <typename T, int SLOT, int R>
__device__ __forceinline__ void rfrag_load(RowFrag<T, R> &f, const T *gRl, size_t ldR, int chunk, int lq) {
  // Issued as raw instructions: with an LDS-DMA load in flight the compiler's waitcnt pass treats
  // every later use of a loaded register as "flat pending" and inserts vmcnt(0), which would wait
  // for the whole prefetch queue.  The hand-placed vmcnt waits in rdirect_step cover these loads.
#pragma unroll
  for (int ks = 0; ks < KT / 4; ++ks) {
    const T *src = gRl + (size_t)(chunk * KT + ks * 4 + lq) * ldR;
    if constexpr (sizeof(T) == 8) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(f.r[SLOT][ks]) : "v"(src));
    else asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(f.r[SLOT][ks]) : "v"(src));
  }
}

// One chunk (KT columns of 128 rows) of the column panel, HBM -> LDS by LDS-DMA.  fp64: a column is
// one 1 KiB dwordx4 wave-instruction; fp32: two 256-byte dword wave-instructions (gfx950 has no
// 8-byte LDS-DMA).  Loads per wave and chunk: CLOADS.
template <typename T> constexpr int cpanel_loads() { return sizeof(T) == 8 ? KT / 4 : KT / 2; }
template <typename T>
__device__ __forceinline__ void cpanel_stage(const T *gC, size_t ldC, int chunk, T *buf, int lane, int wave) {
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void gbl_void;
#pragma unroll
  for (int i = 0; i < KT / 4; ++i) {
    const int col = wave * (KT / 4) + i;
    const T *src = gC + (size_t)(chunk * KT + col) * ldC;
    if constexpr (sizeof(T) == 8) {
      __builtin_amdgcn_global_load_lds((gbl_void *)(src + lane * 2), (lds_void *)(buf + col * LDST), 16, 0, 0);
    } else {
      __builtin_amdgcn_global_load_lds((gbl_void *)(src + lane), (lds_void *)(buf + col * LDST), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void *)(src + 64 + lane), (lds_void *)(buf + col * LDST + 64), 4, 0, 0);
    }
  }
}

template <typename T, int R = 4>
__device__ __forceinline__ void rdirect_prologue(RowFrag<T, R> &f, const T *gR, size_t ldR, const T *gC, size_t ldC,
                                                 int nchunk, T *smem, int tid) {
  constexpr int CH = KT * LDST;
  if (nchunk <= 0) return;   // (otherwise nchunk >= 8 > D)
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T *gRl = gR + wave * 32 + 2 * (lane & 15);
  static_for<0, deep_dist<R>()>([&](auto jc) {
    constexpr int J = decltype(jc)::value;
    cpanel_stage<T>(gC, ldC, J, smem + J * CH, lane, wave);
    rfrag_load<T, J>(f, gRl, ldR, J, lane >> 4);
  });
}

// ISSUE / LAST are compile-time so that no data-dependent branch sits between a load and its use:
// the compiler's own s_waitcnt insertion then counts the loads in flight exactly (with a branch it
// falls back to vmcnt(0) at the first use of the row fragments, which serialises the pipeline).
// ACC: the chunk belongs to the newest block column; besides feeding the MFMAs its row fragments are
// folded into the running sums  ms[j] += V z,  ms[2 + j] += V^2  (rows 2 l15 + j; zs = z of that block
// column, 16 values per chunk) -- the predictive mean / variance accumulate here instead of in a
// separate pass over V.
template <typename T, int R, int S, bool ISSUE, int WAITN, bool ACC = false>
__device__ __forceinline__ void rdirect_step(typename Prec<T>::acc_t (&acc)[NCB][2], RowFrag<T, R> &f, const T *gRl, size_t ldR,
                                             const T *gC, size_t ldC, int c, T *smem, int lane, int wave,
                                             const T *zs = nullptr, T *ms = nullptr) {
  using P = Prec<T>;
  constexpr int CH = KT * LDST;
  constexpr int D = deep_dist<R>();
  const int l15 = lane & 15, lq = lane >> 4;
  // everything but the loads of the WAITN / (cpanel_loads + 4 row fragments) chunks behind chunk c has landed
  static_assert(WAITN >= 0 && WAITN <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WAITN) : "memory");
  // The row fragments were loaded by raw instructions D steps ago; to the compiler they were ready
  // at once.  Pin every use of this step's fragments behind the wait above (volatile asm statements
  // keep their order), or a use that depends on nothing else -- the V^2 sums -- is hoisted over it.
#pragma unroll
  for (int ks = 0; ks < KT / 4; ++ks) asm volatile("" : "+v"(f.r[S][ks]));
  if constexpr (ISSUE) {
    cpanel_stage<T>(gC, ldC, c + D, smem + ((S + D) % R) * CH, lane, wave);
    rfrag_load<T, (S + D) % R>(f, gRl, ldR, c + D, lq);
  }
  // column fragments one k-step ahead of the MFMAs that consume them (two register sets)
  const T *cur = smem + S * CH + lq * LDST + l15;
  T fa[2][NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) fa[0][cb] = cur[cb * DB];
#pragma unroll
  for (int ks = 0; ks < KT / 4; ++ks) {
    if (ks + 1 < KT / 4) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) fa[(ks + 1) & 1][cb] = cur[(ks + 1) * 4 * LDST + cb * DB];
    }
    const typename RowFrag<T, R>::vec2 fb = f.r[S][ks];
    if constexpr (ACC) {
      const T zv = zs[ks * 4 + lq];
      ms[0] = __builtin_fma(fb[0], zv, ms[0]);
      ms[1] = __builtin_fma(fb[1], zv, ms[1]);
      ms[2] = __builtin_fma(fb[0], fb[0], ms[2]);
      ms[3] = __builtin_fma(fb[1], fb[1], ms[3]);
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      acc[cb][0] = P::mfma(fa[ks & 1][cb], fb[0], acc[cb][0]);
      acc[cb][1] = P::mfma(fa[ks & 1][cb], fb[1], acc[cb][1]);
    }
  }
}

// nchunk is a multiple of 8 (k * 128 / 16, or the 8 chunks of the newest block column): whole groups of R steps that all
// issue, then the last 8 chunks -- the newest block column, whose sums ride along -- with the issue and wait pattern of a
// draining queue (chunk j of the tail has min(D - 1, 7 - j) chunks in flight behind it).
template <typename T, int R = 4>
__device__ __forceinline__ void mfma_rowpanel_loop_rdirect(typename Prec<T>::acc_t (&acc)[NCB][2], RowFrag<T, R> &f, const T *gR,
                                                           size_t ldR, const T *gC, size_t ldC, int nchunk, T *smem, int tid,
                                                           const T *zs, T (&ms)[4]) {
  if (nchunk <= 0) return;
  constexpr int D = deep_dist<R>();
  constexpr int L = cpanel_loads<T>() + KT / 4;   // vector-memory instructions per wave and chunk
  static_assert((D - 1) * L <= 63 && D < R && 8 % R == 0, "ring / distance / vmcnt");
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T *gRl = gR + wave * 32 + 2 * (lane & 15);
  int c = 0;
  for (; c + 8 < nchunk; c += R) {
    static_for<0, R>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      rdirect_step<T, R, J, true, (D - 1) * L>(acc, f, gRl, ldR, gC, ldC, c + J, smem, lane, wave);
    });
  }
  static_for<0, 8>([&](auto jc) {
    constexpr int J = decltype(jc)::value;
    constexpr int behind = (D - 1) < (7 - J) ? (D - 1) : (7 - J);
    rdirect_step<T, R, J % R, (J + D < 8), behind * L, true>(acc, f, gRl, ldR, gC, ldC, c + J, smem, lane, wave, zs + J * KT, ms);
  });
}

// --------------------------------------------------------------------------------------------------
// Diagonal-tile update: C(kk) -= sum_j L(k,j) L(k,j)^T is symmetric and only its lower 16x16 blocks
// are factored, so the row panel is staged ONCE per chunk (it is both operands) and wave w owns the
// 16-row blocks w and 7 - w: block row rb needs column blocks 0..rb, i.e. (w + 1) + (8 - w) = 9
// MFMA tiles per k-step on every wave instead of 16.
//   acc[cb][j][reg] = C[row = rb_j*16 + (lane&15)][col = cb*16 + drow(lane, reg)],  rb_0 = w, rb_1 = 7-w,
// defined for cb <= rb_j only.
// --------------------------------------------------------------------------------------------------
// LDS ring of the triangular loop: three chunk buffers; chunks 0 and 1 are in flight while the Gram
// tile is built.  The Gram staging area starts at 2 * KT * LDST elements, i.e. on buffer 2, which is
// first written in iteration 0 after the barrier every wave reaches only when its Gram tile is done.
constexpr int TRI_PD = 2;  // chunks in flight ahead of the one being multiplied
__device__ __forceinline__ constexpr int tri_buf(int i) { return i * KT * LDST; }

// DEEP: the chunk goes HBM -> LDS by LDS-DMA (no VGPR staging), which is what lets the loop run TRI_PD chunks
// ahead.  fp64: one 1 KiB dwordx4 wave-instruction per column; fp32: two 256-byte dword wave-instructions (a
// 512-byte column does not fill a dwordx4 one and the padded LDS columns are not contiguous).  tri_dma_loads<T>()
// wave-instructions per wave and chunk.
template <typename T> constexpr int tri_dma_loads() { return sizeof(T) == 8 ? KT / 4 : KT / 2; }
template <typename T, bool DEEP = sizeof(T) == 8>
__device__ __forceinline__ void stage_chunk_tri(const T *gR, size_t ldR, int chunk, T *buf, int tid) {
  if constexpr (DEEP) {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int i = 0; i < KT / 4; ++i) {
      const int col = wave * (KT / 4) + i;
      const T *src = gR + (size_t)(chunk * KT + col) * ldR;
      if constexpr (sizeof(T) == 8) {
        __builtin_amdgcn_global_load_lds((gbl_void *)(src + lane * 2), (lds_void *)(buf + col * LDST), 16, 0, 0);
      } else {
        __builtin_amdgcn_global_load_lds((gbl_void *)(src + lane), (lds_void *)(buf + col * LDST), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void *)(src + 64 + lane), (lds_void *)(buf + col * LDST + 64), 4, 0, 0);
      }
    }
  } else {
    // fp32: a 512-byte column does not fill a 1 KiB LDS-DMA wave-instruction; 32 bytes per thread through registers
    using vec8 = T __attribute__((ext_vector_type(8)));
    const int sc = tid >> 4, sr = (tid & 15) * 8;
    *reinterpret_cast<vec8 *>(buf + sc * LDST + sr) = *reinterpret_cast<const vec8 *>(gR + sr + (size_t)(chunk * KT + sc) * ldR);
  }
}

// One wavefront's share, W = wave index as a compile-time constant so the MFMA set is static.
template <typename T, int W, bool DEEP>
__device__ __forceinline__ void syrk_tri_wave(typename Prec<T>::acc_t (&acc)[NCB][2], const T *gR, size_t ldR, int nchunk,
                                              T *smem, int tid) {
  using P = Prec<T>;
  constexpr int RB0 = W, RB1 = NCB - 1 - W;
  const int lane = tid & 63;
  const int l15 = lane & 15, lq = lane >> 4;

  auto compute = [&](const T *cur) {
#pragma unroll
    for (int ks = 0; ks < KT / 4; ++ks) {
      const T *ra = cur + (ks * 4 + lq) * LDST + l15;
      T fa[RB1 + 1];
#pragma unroll
      for (int cb = 0; cb <= RB1; ++cb) fa[cb] = ra[cb * DB];  // block cb of the panel = columns of C; rows RB0 / RB1
#pragma unroll
      for (int cb = 0; cb <= RB1; ++cb) {
        if (cb <= RB0) acc[cb][0] = P::mfma(fa[cb], fa[RB0], acc[cb][0]);
        acc[cb][1] = P::mfma(fa[cb], fa[RB1], acc[cb][1]);
      }
    }
  };

  if constexpr (DEEP) {
    // prefetch distance TRI_PD over a ring of TRI_PD + 1 (LDS-DMA): with one or two workgroups per CU nothing
    // else hides the HBM latency
    for (int c = 0; c < nchunk; ++c) {
      // chunk c has landed when at most the loads of chunk c + 1 (4 fp64 / 8 fp32) are still in flight
      if (c + 1 < nchunk) {
        if constexpr (sizeof(T) == 8) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      static_assert(tri_dma_loads<double>() == 4 && tri_dma_loads<float>() == 8, "vmcnt constants");
      if (c + TRI_PD < nchunk) stage_chunk_tri<T, true>(gR, ldR, c + TRI_PD, smem + tri_buf((c + TRI_PD) % (TRI_PD + 1)), tid);
      compute(smem + tri_buf(c % (TRI_PD + 1)));
    }
  } else {
    // fp32: register-staged, two buffers, one chunk ahead (chunk 0 was written by the prologue); four
    // workgroups per CU cover the latency
    using vec8 = T __attribute__((ext_vector_type(8)));
    const int sc = tid >> 4, sr = (tid & 15) * 8;
    const T *src = gR + sr + (size_t)sc * ldR;
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
      vec8 pr;
      if (c + 1 < nchunk) pr = *reinterpret_cast<const vec8 *>(src + (size_t)((c + 1) * KT) * ldR);
      compute(smem + tri_buf(c & 1));
      if (c + 1 < nchunk) *reinterpret_cast<vec8 *>(smem + tri_buf((c + 1) & 1) + sc * LDST + sr) = pr;
      __syncthreads();
    }
  }
}

// DEEP: chunks 0 .. TRI_PD-1 must have been issued with stage_chunk_tri<T, true> before the call (tri_prologue);
// otherwise chunk 0 must have been written by stage_chunk_tri<T, false>.
template <typename T, bool DEEP = sizeof(T) == 8>
__device__ __forceinline__ void mfma_syrk_tri_loop(typename Prec<T>::acc_t (&acc)[NCB][2], const T *gR, size_t ldR,
                                                   int nchunk, T *smem, int tid) {
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  switch (wave) {
    case 0: syrk_tri_wave<T, 0, DEEP>(acc, gR, ldR, nchunk, smem, tid); break;
    case 1: syrk_tri_wave<T, 1, DEEP>(acc, gR, ldR, nchunk, smem, tid); break;
    case 2: syrk_tri_wave<T, 2, DEEP>(acc, gR, ldR, nchunk, smem, tid); break;
    default: syrk_tri_wave<T, 3, DEEP>(acc, gR, ldR, nchunk, smem, tid); break;
  }
}
template <typename T, bool DEEP = sizeof(T) == 8>
__device__ __forceinline__ void tri_prologue(const T *gR, size_t ldR, int nchunk, T *smem, int tid) {
  if (nchunk > 0) stage_chunk_tri<T, DEEP>(gR, ldR, 0, smem + tri_buf(0), tid);
  if (DEEP && nchunk > 1) stage_chunk_tri<T, DEEP>(gR, ldR, 1, smem + tri_buf(1), tid);
}

// --------------------------------------------------------------------------------------------------
// The triangular (diagonal-tile) update on the bf16 matrix cores -- bx6_loop's arithmetic (see there) for mfma_syrk_tri_loop's
// accumulator layout, for the builds whose diagonal tiles sit on the critical path of a block step (mid-size fp32: kind A's
// finish is 16 chunks behind the tile it has just stored, a lone workgroup on its CU).  One panel, three planes per chunk
// buffer, two buffers, D chunks in flight in registers (unconditional buffer loads, zeros past the end), one barrier per chunk.
// Row block rb's [x0 | x1] B operand is the [x0 | x1] A operand of column block rb: 2 + (RB1 + 1) + 2 + (RB1 + 1) reads of 16
// bytes per lane and chunk for 27 MFMAs (9 blocks x 3), issued ahead of the MFMAs with hand-counted waits (bx6_compute_pipe).
// --------------------------------------------------------------------------------------------------
constexpr int BXT_FLOATS = 3 * BX_PLANE / 2;   // floats of one triangular chunk buffer (three planes)
template <int D> struct BxTriStage {
  __amdgpu_buffer_rsrc_t rR;
  int ldR4, vR, offR;
  float xr[D][8];
  __device__ __forceinline__ void init(const float *gR, size_t ldR, int nchunk, int tid) {
    const int r = tid & (TS - 1), h = tid >> 7;
    rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(gR), 0, (int)(((size_t)(nchunk > 0 ? nchunk : 0) * KT * ldR) * sizeof(float)), 0x00020000);
    ldR4 = (int)ldR * 4;
    vR = 4 * r + 8 * h * ldR4;
    offR = bx_pos(r, h);
  }
  template <int S> __device__ __forceinline__ void load(int chunk) {
#pragma unroll
    for (int i = 0; i < 8; ++i) xr[S][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rR, vR, (chunk * KT + i) * ldR4, 0));
  }
  template <int S> __device__ __forceinline__ void store(float *planes) { bx_split_store(xr[S], reinterpret_cast<unsigned short *>(planes), offR); }
};
template <int D> __device__ __forceinline__ void bx6_tri_prologue(BxTriStage<D> &st) {
  static_assert(D == 2 || D == 4, "register sets (the plane buffer of a chunk follows the parity of its set)");
  st.template load<0>(0);
  st.template load<1>(1);
  if (D > 2) st.template load<2 % D>(2);
  if (D > 2) st.template load<3 % D>(3);
}
template <int W, int BO>
__device__ __forceinline__ void bx6_tri_compute(Prec<float>::acc_t (&acc)[NCB][2], const unsigned (&e)[2], bool hi) {
  constexpr int RB0 = W, RB1 = NCB - 1 - W, PB = 2 * BX_PLANE, CBB = 2 * DB * BXS;
  const unsigned o02 = hi ? 2 * PB : 0, o01 = hi ? PB : 0, o20 = hi ? 0 : 2 * PB, o10 = hi ? 0 : PB;
  bxu4 a02[RB1 + 1], a01[RB1 + 1];
  bxu4 b20[2] = {bx_rd(e[RB0 & 1] + o20, BO + RB0 * CBB), bx_rd(e[RB1 & 1] + o20, BO + RB1 * CBB)};
  static_for<0, RB1 + 1>([&](auto cc) {
    constexpr int CB = decltype(cc)::value;
    a02[CB] = bx_rd(e[CB & 1] + o02, BO + CB * CBB);
  });
  bxu4 b10[2] = {bx_rd(e[RB0 & 1] + o10, BO + RB0 * CBB), bx_rd(e[RB1 & 1] + o10, BO + RB1 * CBB)};
  asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(b20[0]), "+v"(b20[1]));
  static_for<0, RB1 + 1>([&](auto cc) { bx_tie(a02[decltype(cc)::value]); });
  static_for<0, RB1 + 1>([&](auto cc) {
    constexpr int CB = decltype(cc)::value;
    a01[CB] = bx_rd(e[CB & 1] + o01, BO + CB * CBB);
  });
  const bf8 c20[2] = {__builtin_bit_cast(bf8, b20[0]), __builtin_bit_cast(bf8, b20[1])};
  static_for<0, RB1 + 1>([&](auto cc) {   // the smallest pair of terms first
    constexpr int CB = decltype(cc)::value;
    const bf8 a = __builtin_bit_cast(bf8, a02[CB]);
    if constexpr (CB <= RB0) acc[CB][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, c20[0], acc[CB][0], 0, 0, 0);
    acc[CB][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, c20[1], acc[CB][1], 0, 0, 0);
  });
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b10[0]), "+v"(b10[1]));
  static_for<0, RB1 + 1>([&](auto cc) { bx_tie(a01[decltype(cc)::value]); });
  const bf8 c10[2] = {__builtin_bit_cast(bf8, b10[0]), __builtin_bit_cast(bf8, b10[1])};
  const bf8 c01[2] = {__builtin_bit_cast(bf8, a01[RB0]), __builtin_bit_cast(bf8, a01[RB1])};   // [x0 | x1] of the row blocks
  static_for<0, RB1 + 1>([&](auto cc) {
    constexpr int CB = decltype(cc)::value;
    const bf8 a = __builtin_bit_cast(bf8, a01[CB]);
    if constexpr (CB <= RB0) acc[CB][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, c10[0], acc[CB][0], 0, 0, 0);
    acc[CB][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, c10[1], acc[CB][1], 0, 0, 0);
  });
  static_for<0, RB1 + 1>([&](auto cc) {
    constexpr int CB = decltype(cc)::value;
    const bf8 a = __builtin_bit_cast(bf8, a01[CB]);
    if constexpr (CB <= RB0) acc[CB][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, c01[0], acc[CB][0], 0, 0, 0);
    acc[CB][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, c01[1], acc[CB][1], 0, 0, 0);
  });
}
template <int D, int W, int S>   // S = c % D
__device__ __forceinline__ void bx6_tri_iter(Prec<float>::acc_t (&acc)[NCB][2], BxTriStage<D> &st, int c, int nchunk, float *smem, const unsigned (&e)[2], bool hi) {
  st.template load<S>(c + D);            // unconditional (bx6_iter)
  lds_barrier();                         // chunk c is in its planes, chunk c - 1 is done with
  bx6_tri_compute<W, (S & 1) * 6 * BX_PLANE>(acc, e, hi);
  if (c + 1 < nchunk) st.template store<(S + 1) % D>(smem + ((S + 1) & 1) * BXT_FLOATS);
}
template <int D, int W>
__device__ __forceinline__ void bx6_tri_wave(Prec<float>::acc_t (&acc)[NCB][2], BxTriStage<D> &st, int nchunk, float *smem, int tid) {
  const int lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
  const unsigned base = (unsigned)(size_t)smem;
  const unsigned e[2] = {base + 2 * bx_pos(l15, lq & 1), base + 2 * (bx_pos(DB + l15, lq & 1) - DB * BXS)};
  st.template store<0>(smem);            // chunk 0
  for (int c0 = 0; c0 < nchunk; c0 += D) {   // nchunk is a multiple of 8
    bx6_tri_iter<D, W, 0>(acc, st, c0, nchunk, smem, e, lq >> 1);
    bx6_tri_iter<D, W, 1>(acc, st, c0 + 1, nchunk, smem, e, lq >> 1);
    if (D > 2) bx6_tri_iter<D, W, 2 % D>(acc, st, c0 + 2, nchunk, smem, e, lq >> 1);
    if (D > 2) bx6_tri_iter<D, W, 3 % D>(acc, st, c0 + 3, nchunk, smem, e, lq >> 1);
  }
}
template <int D>
__device__ __forceinline__ void bx6_syrk_tri_loop(Prec<float>::acc_t (&acc)[NCB][2], BxTriStage<D> &st, int nchunk, float *smem, int tid) {
  if (nchunk <= 0) return;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  switch (wave) {
    case 0: bx6_tri_wave<D, 0>(acc, st, nchunk, smem, tid); break;
    case 1: bx6_tri_wave<D, 1>(acc, st, nchunk, smem, tid); break;
    case 2: bx6_tri_wave<D, 2>(acc, st, nchunk, smem, tid); break;
    default: bx6_tri_wave<D, 3>(acc, st, nchunk, smem, tid); break;
  }
}

// General exp (|x| small enough not to overflow): the covariance exponent with log(amplitude)
// folded in can be slightly positive.  Same reduction / polynomial as exp_nonpos.
__device__ __forceinline__ double exp_gen(double x, const ExpC &e) { return exp_nonpos(x, e); }
__device__ __forceinline__ float exp_gen(float x, const ExpC &) { return __expf(fmaxf(x, -104.f)); }

constexpr int GK = 12;  // augmented point length: MAXD coordinates, 2 norm slots, padded to a multiple of 4
// floats in front of the newest block column's z in the bf16-plane loops: the plane buffer and the Gram inputs behind it, or (DBL) two
// plane buffers, the Gram inputs over the second
template <bool DBL> constexpr int bx_z_offset() { return DBL ? 2 * BX_FLOATS : BX_FLOATS + 2 * GK * TS + 3 * TS; }
static_assert(2 * GK * TS + 3 * TS <= BX_FLOATS, "Gram inputs fit the second plane buffer");

// acc <- -G, in registers (the MFMA loop then adds L L^T, so acc ends as -S).  The covariance exponent e_ij = -0.5 |a_i - b_j|^2 (+ log amplitude)
// is itself an inner product of augmented points
//     a' = (a, -0.5|a|^2 + log amp, 1)        b' = (b, 1, -0.5|b|^2)
// so it is produced by three MFMA 16x16x4 per 16x16 block straight into the accumulator layout; the
// VALU then only evaluates exp.  (This is the x^2 + x'^2 - 2xx' expansion GPy itself uses for r^2;
// its rounding error is ~1e-16 |x|^2 absolute in the exponent.)  FAST = interior tile: no selects.
// NR = 16-row blocks per wave: 2 (rows 2 l15 + j of the wave's 32-row slab; TRI: block rows `wave` and `7 - wave`) or 1 (row
// l15 of the wave's 16-row slab: the 64-row tiles of k_rows64).
// DIFF1 (fp32, one input dimension, SE kernels): the exponent comes from the DIFFERENCE of the two raw coordinates, as the
// Brownian form's does.  The inner-product expansion carries ~1e-7 |x|^2 / ell^2 of absolute error in the exponent in single
// precision -- 60 x the rounding of a Gram entry -- and dense one-dimensional inputs are exactly the windows whose conditioning
// amplifies it: tests/fuzz/fuzz_parity.py found 1 window in 2 000 (d = 1, N = 255 ... 1000) at 1.05-1.3 of the 1e-3 bar where
// single-precision LAPACK on the exact Gram matrix is at 5e-5.  For d >= 2 the expansion stays (d subtractions and FMAs per
// entry on the VALU would cost more than the Gram phase has; no window of the sweeps needs it).
template <typename T, bool BROWN, bool FAST, bool TRI = false, int NR = 2, bool DIFF1 = false>
__device__ __forceinline__ void gram_apply_tile(const FitArgs &p, typename Prec<T>::acc_t (&acc)[NCB][NR],
                                                const T *__restrict__ xrT, const T *__restrict__ xcT,
                                                const T *__restrict__ xraw, const T *__restrict__ craw,
                                                const T *__restrict__ yc, bool extra, int rowbase, int colbase, T amp,
                                                T amp_b, T diag_add, int lane, int wave, T inv_ell = T(0)) {
  using P = Prec<T>;
  using acc_t = typename P::acc_t;
  const int N = p.N, M = p.M, l15 = lane & 15, lq = lane >> 4;
  ExpC ec;
  ec.load();
  // local rows of this lane: 2 l15 + {0, 1} of the wave's 32-row slab, or (TRI, diagonal tile) row
  // l15 of the 16-row blocks `wave` and `7 - wave`, of which only column blocks <= the row block exist
  static_assert(NR == 2 || (NR == 1 && !TRI), "row blocks per wave");
  const int rbj[2] = {TRI ? wave : 0, TRI ? NCB - 1 - wave : 0};
  const int rl[2] = {TRI ? rbj[0] * DB + l15 : (NR == 2 ? wave * 32 + 2 * l15 : wave * DB + l15),
                     TRI ? rbj[1] * DB + l15 : wave * 32 + 2 * l15 + 1};
  T fb[NR][GK / 4];
#pragma unroll
  for (int j = 0; j < NR; ++j)
#pragma unroll
    for (int s = 0; s < GK / 4; ++s) fb[j][s] = xrT[(4 * s + lq) * TS + rl[j]];
  static_assert(!DIFF1 || (!BROWN && sizeof(T) == 4), "the difference form is for the fp32 SE kernels");
  T xrw[2] = {T(0), T(0)};
  if (BROWN || DIFF1) {
    xrw[0] = xraw[rl[0]];
    if (NR == 2) xrw[1] = xraw[rl[1]];
  }
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    if (TRI && cb > rbj[0] && cb > rbj[1]) continue;
    T fa[GK / 4];
#pragma unroll
    for (int s = 0; s < GK / 4; ++s) fa[s] = xcT[(4 * s + lq) * TS + cb * DB + l15];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      if (TRI && cb > rbj[j]) continue;
      acc_t e = acc_t{0, 0, 0, 0};
      if constexpr (!DIFF1) {
#pragma unroll
        for (int s = 0; s < GK / 4; ++s) e = P::mfma(fa[s], fb[j][s], e);
      }
      const int grow = rowbase + rl[j];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cl = cb * DB + P::drow(lane, r);
        const int gcol = colbase + cl;
        T g;
        if constexpr (DIFF1) {
          const T df = (xrw[j] - craw[cl]) * inv_ell;
          g = amp * exp_gen(T(-0.5) * df * df, ec);
        } else if (!BROWN) {
          g = CGP_DBG_ON(p, 128) ? e[r] : exp_gen(e[r], ec);
        } else {
          const T x = xrw[j], xp = craw[cl];
          const bool same = !FAST && !extra && grow == gcol;  // GPy forces r^2 = 0 on the auto-covariance diagonal
          T ee = e[r] > T(0) ? T(0) : e[r];                   // r^2 clipped at 0
          if constexpr (sizeof(T) == 4) {
            // GPy's r^2 = x^2 + x'^2 - 2 x x' on raw tick counts (what e[] holds, reproduced as is in fp64) loses
            // the RBF factor itself in single precision (ticks ~ 1e3: 0.06 absolute in r^2, and the window's
            // conditioning amplifies a 1e-4 perturbation of K to per cent); the difference of two ticks is exact
            const T df = (x - xp) * inv_ell;
            ee = T(-0.5) * df * df;
          }
          ee = same ? T(0) : ee;
          const int sx = (x > T(0)) - (x < T(0)), sp = (xp > T(0)) - (xp < T(0));
          const T ax = x < T(0) ? -x : x, ap = xp < T(0) ? -xp : xp;
          const T kb = (sx == sp) ? amp_b * (ax < ap ? ax : ap) : T(0);
          g = amp * exp_gen(ee, ec) * kb;
        }
        if (!FAST) {
          const bool colok = gcol < N;
          if (!extra) {
            const bool dg = grow == gcol;
            g = dg ? g + diag_add : g;
            g = (grow < N && colok) ? g : (dg ? T(1) : T(0));  // identity padding keeps the factor well defined
          } else {
            if (p.xid) g = (grow == gcol) ? T(1) : T(0);  // gradient mode: "test rows" = identity
            g = (grow == M) ? yc[cl] : g;
            g = (!colok || grow > M) ? T(0) : g;
          }
        }
        acc[cb][j][r] = -g;
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // two 16x16 blocks (8 independent exp chains) at a time
  }
}

// Per-fit derived constants (inverse length-scales, log amplitude, diagonal addend), computed once
// per schedule so the tile kernels read them with scalar loads instead of dividing per thread.
__global__ void k_prep(FitArgs p, int batch, double *prep, int zero_info, int *wready) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  if (zero_info) p.info[b] = 0;      // a fit schedule starts with a clean status (no separate memset node)
  if (wready) wready[b] = 0;
  const double *th = p.theta + (size_t)b * MAX_THETA;
  double *o = prep + (size_t)b * PREP_N;
  const int kid = p.kernel_id, d = p.d;
  for (int q = 0; q < MAXD; ++q) {
    double s = 0.0;
    if (q < d) s = (kid == K_SE_ARD) ? 1.0 / th[1 + q] : 1.0 / th[1];
    o[q] = s;
  }
  const bool brown = kid == K_RBF_BROWNIAN;
  o[8] = brown ? 0.0 : log(th[0]);
  o[9] = th[0];
  o[10] = brown ? th[2] : 0.0;
  const int nth = (kid == K_SE_ISO) ? 3 : (kid == K_SE_ARD ? d + 2 : 4);
  o[11] = th[nth - 1] + 1e-8 + (p.jitter ? p.jitter[b] : 0.0);
  for (int q = 12; q < PREP_N; ++q) o[q] = 0.0;
}

// The raw coordinates of the point a thread prepares for the Gram tile (thread t < 128: row point t,
// t >= 128: column point t - 128), fetched at kernel entry so their HBM latency is hidden behind
// the MFMA loop.
template <typename T> struct GramPre {
  T v[MAXD];
  T yv;
};

// row0 >= 0: the tile's first row inside its block (matrix rows, or extra rows when rt >= NT) instead of the 128-row tile rt's
template <typename T>
__device__ __forceinline__ void gram_prefetch(const FitArgs &p, int b, int k, int rt, int tid, GramPre<T> &g, int row0 = -1) {
  const int d = p.d, N = p.N, M = p.M;
  const bool extra = rt >= p.NT;
  const T *__restrict__ Xb = reinterpret_cast<const T *>(p.X) + (size_t)b * d * N;
  const T *__restrict__ Xsb = reinterpret_cast<const T *>(p.Xs) + (size_t)b * d * M;
  const T *__restrict__ yb = reinterpret_cast<const T *>(p.y) + (size_t)b * N;
  const bool isrow = tid < TS;
  const int r = tid & 127;
  const T *src = Xb;
  int idx, len = N;
  if (isrow) {
    if (!extra) idx = (row0 >= 0 ? row0 : rt * TS) + r;
    else { src = Xsb; idx = (row0 >= 0 ? row0 : (rt - p.NT) * TS) + r; len = M; }
  } else idx = k * TS + r;
  const bool ok = idx < len;
#pragma unroll
  for (int q = 0; q < MAXD; ++q) g.v[q] = (q < d && ok) ? src[(size_t)q * len + idx] : T(0);
  g.yv = (!isrow && ok) ? yb[idx] : T(0);
}

// Stages the augmented points of tile (rt, k) in LDS ([GK][128], component-major) and applies
// acc <- G - acc.
template <typename T, bool TRI = false, int NR = 2>
__device__ __forceinline__ void gram_apply(const FitArgs &p, typename Prec<T>::acc_t (&acc)[NCB][NR], T *__restrict__ smem,
                                           int b, int k, int rt, int tid, const GramPre<T> &g, PhaseClock *pc = nullptr,
                                           int slot = 0, int row0 = -1, bool live = true) {
  const double *__restrict__ pr = p.prep + (size_t)b * PREP_N;
  const int kid = p.kernel_id, N = p.N, M = p.M;
  const bool extra = rt >= p.NT;
  const bool brown = kid == K_RBF_BROWNIAN;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  T *xrT = smem;                 // [GK][128] augmented row points
  T *xcT = smem + GK * TS;       // [GK][128] augmented column points
  T *xraw = smem + 2 * GK * TS;  // [128] raw first coordinate of the row points (Brownian factor)
  T *craw = xraw + TS;           // [128] raw first coordinate of the column points
  T *yc = craw + TS;             // [128] y of the tile-k columns (only the y row uses it)
  {
    const bool isrow = tid < TS;
    const int r = tid & 127;
    T *dst = isrow ? xrT : xcT;
    T nrm = 0;
#pragma unroll
    for (int q = 0; q < MAXD; ++q) {
      const T v = g.v[q] * T(pr[q]);
      nrm = __builtin_fma(v, v, nrm);
      dst[q * TS + r] = v;
    }
    dst[MAXD * TS + r] = isrow ? (T(-0.5) * nrm + T(pr[8])) : T(1);
    dst[(MAXD + 1) * TS + r] = isrow ? T(1) : T(-0.5) * nrm;
    dst[(MAXD + 2) * TS + r] = T(0);
    dst[(MAXD + 3) * TS + r] = T(0);
    if (isrow) xraw[r] = g.v[0];
    else {
      craw[r] = g.v[0];
      yc[r] = g.yv;
    }
  }
  __syncthreads();
  if (pc) pc->lap(p, slot);  // inputs fetched and staged
  const T amp = T(pr[9]), amp_b = T(pr[10]), diag_add = T(pr[11]);
  const int rowbase = row0 >= 0 ? row0 : (extra ? (rt - p.NT) * TS : rt * TS);
  const int colbase = k * TS;
  constexpr int ROWS = NR == 2 ? TS : TS / 2;   // rows of the tile
  const bool fast = (colbase + TS <= N) && (extra ? (rowbase + ROWS <= M && !p.xid) : (rt != k && rowbase + ROWS <= N));
  if (!live) return;   // (wave-uniform; no barrier below)
  if constexpr (sizeof(T) == 4) {
    if (!brown && p.d == 1) {   // fp32, one input dimension: exponent from the coordinate difference (DIFF1)
      if constexpr (TRI) gram_apply_tile<T, false, false, true, NR, true>(p, acc, xrT, xcT, xraw, craw, yc, extra, rowbase, colbase, amp, amp_b, diag_add, lane, wave, T(pr[0]));
      else if (fast) gram_apply_tile<T, false, true, false, NR, true>(p, acc, xrT, xcT, xraw, craw, yc, extra, rowbase, colbase, amp, amp_b, diag_add, lane, wave, T(pr[0]));
      else gram_apply_tile<T, false, false, false, NR, true>(p, acc, xrT, xcT, xraw, craw, yc, extra, rowbase, colbase, amp, amp_b, diag_add, lane, wave, T(pr[0]));
      return;
    }
  }
  if constexpr (TRI) {  // diagonal tile: never "fast" (it carries the noise diagonal)
    if (brown) gram_apply_tile<T, true, false, true>(p, acc, xrT, xcT, xraw, craw, yc, extra, rowbase, colbase, amp, amp_b, diag_add, lane, wave, T(pr[0]));
    else gram_apply_tile<T, false, false, true>(p, acc, xrT, xcT, xraw, craw, yc, extra, rowbase, colbase, amp, amp_b, diag_add, lane, wave);
  } else if (brown) {
    if (fast) gram_apply_tile<T, true, true, false, NR>(p, acc, xrT, xcT, xraw, craw, yc, extra, rowbase, colbase, amp, amp_b, diag_add, lane, wave, T(pr[0]));
    else gram_apply_tile<T, true, false, false, NR>(p, acc, xrT, xcT, xraw, craw, yc, extra, rowbase, colbase, amp, amp_b, diag_add, lane, wave, T(pr[0]));
  } else {
    if (fast) gram_apply_tile<T, false, true, false, NR>(p, acc, xrT, xcT, xraw, craw, yc, extra, rowbase, colbase, amp, amp_b, diag_add, lane, wave);
    else gram_apply_tile<T, false, false, false, NR>(p, acc, xrT, xcT, xraw, craw, yc, extra, rowbase, colbase, amp, amp_b, diag_add, lane, wave);
  }
}

// --------------------------------------------------------------------------------------------------
// Diagonal tile inside the panel launch (schedule "fused", the default).  The workgroup that owns
// tile (k+1, k) of step k is the only one the next diagonal tile waits for, so it carries on and
// updates, factors and inverts tile (k+1, k+1) itself while the rest of the step-k launch keeps the
// other CUs busy: no diagonal launch sits between two panel launches any more.  To share a CU with a
// second panel workgroup it has to live in the panel's LDS budget, so the tile is held as its 36
// lower 16x16 blocks only (72 KB) and L^-1 grows in the blocks the factorisation has finished with:
//   step jb:  (a)  wave 0: L_jj, Dinv_jb in registers            | waves 1-3: (e) of step jb-1
//             (b)  L_i,jb = A_i,jb Dinv_jb^T      (i > jb)        (b') W_jb,j = -Dinv_jb T_jb,j   (j < jb)
//             (c)  A_bi,bj -= L_bi,jb L_bj,jb^T   (bi >= bj > jb) (c') T_i,j += L_i,jb W_jb,j     (i > jb > j)
//             (e)  L_i,jb -> HBM, then T_i,jb = L_i,jb Dinv_jb overwrites it
// with T_i,j = sum_{j <= q < current step} L_i,q W_q,j the running right-looking sum of the forward
// substitution L W = I; when step i arrives it is complete and (b') turns it into W_i,j.  A blocks
// are stored [col][row], T / W blocks [row][col]: every MFMA operand read and accumulator access
// below is then a contiguous 512-byte run per wavefront (no LDS bank conflicts).
// --------------------------------------------------------------------------------------------------
__device__ __forceinline__ constexpr int tri_blk(int i, int j) { return (i * (i + 1) / 2 + j) * DB * DB; }
template <typename T> constexpr int diag_lds_elems() { return 36 * DB * DB + 3 * DB * DB; }

template <typename T>
__device__ __forceinline__ void diag_factor_packed(const FitArgs &p, T *__restrict__ Bk, T *__restrict__ tile, int ld,
                                                   T *__restrict__ Wk, int b, int k, int tid) {
  using P = Prec<T>;
  using acc_t = typename P::acc_t;
  constexpr int NB = TS / DB;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int live = p.N - k * TS;  // rows / columns of this tile that belong to the window
  T *Dv = Bk + 36 * DB * DB;  // Dv[q][x]  = Dinv_jb[x][q]
  T *DvT = Dv + DB * DB;      // DvT[jb & 1][q][c] = Dinv_jb[q][c]
  int bad = 0;
  for (int jb = 0; jb < NB; ++jb) {
    const int j0 = jb * DB;
    if (wave == 0) {
      T *Djj = Bk + tri_blk(jb, jb);
      T a[DB], w[DB];
      if (j0 < live) {
#pragma unroll
        for (int c = 0; c < DB; ++c) a[c] = Djj[c * DB + l15];
        factor_block16<T>(a, w, bad, k * TS + j0, l15, [&] {
#pragma unroll
          for (int c = 0; c < DB; ++c) a[c] = Djj[c * DB + l15];
        });
      } else {  // identity padding beyond the window (N not a multiple of 128): nothing to factor
#pragma unroll
        for (int c = 0; c < DB; ++c) a[c] = w[c] = (c == l15) ? T(1) : T(0);
      }
      if (lane < DB) {
        T *dvt = DvT + (jb & 1) * DB * DB;
#pragma unroll
        for (int i = 0; i < DB; ++i) {
          Dv[l15 * DB + i] = w[i];
          dvt[i * DB + l15] = w[i];
          Wk[wimg_blk(jb, jb) + l15 * DB + i] = -w[i];  // w[i] = W[j0 + i][j0 + l15]
        }
#pragma unroll
        for (int c = 0; c < DB; ++c) tile[(size_t)(j0 + c) * ld + j0 + l15] = (l15 >= c) ? a[c] : T(0);
      }
    } else if (jb > 0) {
      const int pj = jb - 1;
      const T *dvt = DvT + (pj & 1) * DB * DB;
      for (int i = jb + wave - 1; i < NB; i += 3) {
        T *blk = Bk + tri_blk(i, pj);
        T fa[4], fb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          fa[ks] = blk[(ks * 4 + lq) * DB + l15];  // L_i,pj[r = l15][q]
          fb[ks] = dvt[(ks * 4 + lq) * DB + l15];  // Dinv_pj[q][c = l15]
          tile[(size_t)(pj * DB + ks * 4 + lq) * ld + i * DB + l15] = fa[ks];
        }
        acc_t acc = acc_t{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = P::mfma(fa[ks], fb[ks], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) blk[P::drow(lane, r) * DB + l15] = acc[r];  // T_i,pj[r][c]
      }
    }
    lds_barrier();
    // (b) and (b'): NB - 1 independent 16x16 products
    for (int t = wave; t < NB - 1; t += 4) {
      const bool below = t < NB - 1 - jb;
      T *blk = below ? Bk + tri_blk(jb + 1 + t, jb) : Bk + tri_blk(jb, t - (NB - 1 - jb));
      T fa[4], fb[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        fa[ks] = Dv[(ks * 4 + lq) * DB + l15];   // Dinv_jb[x = l15][q]
        fb[ks] = blk[(ks * 4 + lq) * DB + l15];  // A_i,jb[r = l15][q]   or   T_jb,j[q][c = l15]
      }
      acc_t acc = acc_t{0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = P::mfma(fa[ks], fb[ks], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) blk[P::drow(lane, r) * DB + l15] = below ? acc[r] : -acc[r];
    }
    lds_barrier();
    // (c) and (c')
    const int nb = NB - 1 - jb;
    const int nc = nb * (nb + 1) / 2;
    for (int idx = wave; idx < nc + nb * jb; idx += 4) {
      T *cblk;
      const T *ablk, *bblk;
      bool neg;
      if (idx < nc) {
        int bj = 0, rem = idx;
        while (rem >= nb - bj) {
          rem -= nb - bj;
          ++bj;
        }
        const int bi = bj + rem + jb + 1;
        bj += jb + 1;
        cblk = Bk + tri_blk(bi, bj);
        ablk = Bk + tri_blk(bj, jb);
        bblk = Bk + tri_blk(bi, jb);
        neg = true;
      } else {
        const int e = idx - nc;
        const int i = jb + 1 + e / jb, j = e % jb;
        cblk = Bk + tri_blk(i, j);
        ablk = Bk + tri_blk(i, jb);
        bblk = Bk + tri_blk(jb, j);
        neg = false;
      }
      acc_t acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = cblk[P::drow(lane, r) * DB + l15];
      T fa[4], fb[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const T av = ablk[(ks * 4 + lq) * DB + l15];
        fa[ks] = neg ? -av : av;
        fb[ks] = bblk[(ks * 4 + lq) * DB + l15];
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = P::mfma(fa[ks], fb[ks], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) cblk[P::drow(lane, r) * DB + l15] = acc[r];
    }
    lds_barrier();
  }
  if (tid == 0 && bad != 0 && p.info[b] == 0) p.info[b] = bad;
  // strictly lower blocks of W -> the negated block image in HBM (Bk holds W_ij[r][c], the image [q = c][r])
  for (int idx = tid; idx < 28 * DB * DB; idx += 256) {
    const int blk = idx >> 8, e = idx & 255;
    int i = 1, rem = blk;
    while (rem >= i) {
      rem -= i;
      ++i;
    }
    const int j = rem, c = e >> 4, r = e & 15;
    Wk[wimg_blk(i, j) + c * DB + r] = -Bk[tri_blk(i, j) + r * DB + c];
  }
}

// The diagonal tile kn in <= 78 KB of LDS: S(kn,kn) = G(kn,kn) - sum_{j<kn} L(kn,j) L(kn,j)^T, then the packed
// factorisation (L(kn,kn), -W_kn written).  TRI (fp64): triangular update loop, see mfma_syrk_tri_loop.
//
// In the throughput schedule the tile never gets a launch of its own (after step 0): its update is cut at
// block column kn - 2, and both pieces ride in panel launches whose other workgroups keep the CUs busy
// ("diagonal tile inside the panel launches"):
//   DIAG_PARTIAL  launch kn - 2, one extra workgroup per fit: acc = -G + sum_{j < kn-2} L L^T (every operand
//                 is final since launch kn - 3), dumped as a raw register image to p.dpart[kn & 1];
//   DIAG_FINISH   launch kn - 1, by the workgroup that has just stored tile (kn, kn-1): reloads the image
//                 (kn <= 2: starts from the Gram tile instead), adds block columns kn-2 and kn-1, factors.
// so every dependency of the diagonal tile crosses a launch boundary except the workgroup's own tile,
// which it fences.  DIAG_FULL is the whole thing in one go (k_diag_lean: step 0, A/B schedules).
enum { DIAG_FULL = 0, DIAG_PARTIAL = 1, DIAG_FINISH = 2 };
// Register images of pre-updated tiles are kept by parity of the tile index (2 slots per fit: a slot is rewritten two
// launches after it was read, the launch boundaries order that) -- or, one-launch schedule (k_sched), one slot per tile
// index: every address is then written once per call.
__device__ __forceinline__ int img_slots(const FitArgs &p) { return p.img_slots > 0 ? p.img_slots : 2; }
constexpr bool kTriDiag = true;  // diagonal tile: only its lower 16x16 blocks are updated (9 of 16 MFMA tiles per k-step)
constexpr int DPART = 2 * NCB * 4 * 256;  // elements of one register image: 64 accumulator values x 256 threads

template <typename T, bool TRI, bool STORE>
__device__ __forceinline__ void acc_image(typename Prec<T>::acc_t (&acc)[NCB][2], T *__restrict__ img, int tid) {
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (TRI && cb > (j ? NCB - 1 - wave : wave)) continue;  // wave-uniform: block not part of the triangle
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T *q = img + ((cb * 2 + j) * 4 + r) * 256 + tid;
        if constexpr (STORE) *q = acc[cb][j][r];
        else acc[cb][j][r] = *q;
      }
    }
}

// FAT: the launch carries potf2_lds_elems<T>() of LDS, so the tile is factored by potf2_tile (factor chain on wave 0 beside
// the trailing updates and the inverse on waves 1-3; in-kernel clocks, fp32, lone workgroup: 67 k ticks against the packed
// form's 89 k, 76 k against 124 k with 64 fits on the chip) -- the packed form exists to fit TWO diagonal workgroups on a CU.
template <typename T> constexpr int potf2_lds_elems() { return TS * LDP + 8 * DB * DB + 4 * DB * DB + 8; }
template <typename T, int MODE, bool TRI, bool DEEP = sizeof(T) == 8, bool FAT = false>
__device__ __forceinline__ void diag_next(const FitArgs &p, typename Prec<T>::acc_t (&acc)[NCB][2], T *__restrict__ smem,
                                          T *__restrict__ Lw, int b, int kn, int tid, PhaseClock *pc = nullptr, bool img_ready = false) {
  using P = Prec<T>;
  using vec2 = T __attribute__((ext_vector_type(2)));
  const int ld = p.ld;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  constexpr int CH2 = 2 * KT * LDST;
  if constexpr (MODE == DIAG_FINISH) {
    // Tile (kn, kn-1), written by THIS workgroup a moment ago, is the last block column of the row panel.
    // Workgroup scope is enough and is all that is wanted: the waves of a workgroup share their CU's
    // write-through L1, nobody else reads that tile in this launch and this CU never held its lines
    // before, so "every wave's stores have left the CU" (vmcnt(0)) + a barrier orders them before the
    // loads below.  An agent-scope pair here would write back the whole XCD L2 and drop the CU's L1 once
    // per fit in every launch -- measured: it costs more than the diagonal launches it replaces.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  // block columns [c_first, c_last) of the row panel are applied here
  const bool from_image = MODE == DIAG_FINISH && kn >= 3;
  const int c_first = from_image ? (kn - 2) * (TS / KT) : 0;
  const int c_last = (MODE == DIAG_PARTIAL ? kn - 2 : kn) * (TS / KT);
  const int nchunk = c_last - c_first;
  const T *gR = Lw + (size_t)kn * TS + (size_t)c_first * KT * ld;
  T *img = reinterpret_cast<T *>(p.dpart) + ((size_t)b * img_slots(p) + kn % img_slots(p)) * DPART;
  // fp32 builds with the deep loops (mid-size; the fat k_diag_lean): the update on the bf16 matrix cores (bx6_syrk_tri_loop)
  // (the full-batch build keeps the fp32-input MFMA here: with this loop inlined its kernel -- then at 168 VGPRs, 128 now -- spilled 220 registers)
  constexpr bool BXT = kF32Bf16x6 && kBxTri && sizeof(T) == 4 && TRI && DEEP;
  BxTriStage<kBxMidSets> bxt;   // (only BXT uses it)
  {
    GramPre<T> gp;
    if (!from_image) gram_prefetch<T>(p, b, kn, kn, tid, gp);
    if constexpr (BXT) {
      bxt.init(gR, (size_t)ld, nchunk, tid);
      bx6_tri_prologue(bxt);   // (nchunk = 0: the descriptor is empty, zeros without a memory access)
    } else if constexpr (TRI) tri_prologue<T, DEEP>(gR, (size_t)ld, nchunk, smem, tid);  // DEEP: DMA ring, 2 chunks ahead
    else if (nchunk > 0) stage_first_chunk<T>(gR, (size_t)ld, gR, (size_t)ld, smem, tid);
    if (from_image) {
      if (!img_ready) acc_image<T, TRI, false>(acc, img, tid);   // (img_ready: the caller requested it before the fence, see panel_tile_body)
    } else gram_apply<T, TRI>(p, acc, smem + (BXT ? BXT_FLOATS : CH2), b, kn, kn, tid, gp);   // BXT: behind plane buffer 0, over buffer 1
  }
  if (pc) pc->lap(p, 344);  // finisher: fence + image / Gram tile (measurement build; slots 344.. = all steps summed)
  if constexpr (BXT) bx6_syrk_tri_loop(acc, bxt, nchunk, smem, tid);
  else if constexpr (TRI) mfma_syrk_tri_loop<T, DEEP>(acc, gR, (size_t)ld, nchunk, smem, tid);
  else mfma_rowpanel_loop<T, true>(acc, gR, (size_t)ld, gR, (size_t)ld, nchunk, smem, tid);
  if (pc) pc->lap(p, 345);  // the two newest block columns
  if constexpr (MODE == DIAG_PARTIAL) {
    acc_image<T, TRI, true>(acc, img, tid);
    return;
  }
  __syncthreads();
  T *tile = Lw + (size_t)(kn * TS) * ld + (size_t)kn * TS;
  if constexpr (FAT) {
    static_assert(!FAT || TRI, "the fat form takes the triangular accumulator layout");
    T *At = smem;  // element (r, c) at At[c * LDP + r]; lower 16x16 blocks only (wave w: block rows w and 7 - w)
    T *Dv = At + TS * LDP;
    T *Ts = Dv + 8 * DB * DB;
    int *flag = reinterpret_cast<int *>(Ts + 4 * DB * DB);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rb = j ? NCB - 1 - wave : wave;
        if (cb <= rb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) At[(cb * DB + P::drow(lane, r)) * LDP + rb * DB + l15] = -acc[cb][j][r];
        }
      }
    if (tid == 0) *flag = 0;
    __syncthreads();
    if (pc) pc->lap(p, 346);  // tile into LDS
    potf2_tile<T>(p, At, Dv, Ts, flag, tile, ld, b, kn, tid);
    if (pc) {
      pc->lap(p, 347);        // factorisation + inverse + stores
      pc->count(p, 351);
    }
    return;
  }
  // S = -acc -> the 36 lower blocks
  if constexpr (TRI) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rb = j ? NCB - 1 - wave : wave;
        if (cb <= rb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) smem[tri_blk(rb, cb) + P::drow(lane, r) * DB + l15] = -acc[cb][j][r];
        }
      }
  } else {
    // rows 32 wave + 2 l15 + {0, 1} live in block row 2 wave + (l15 >> 3)
    const int rb = 2 * wave + (l15 >> 3), rin = (2 * l15) & 15;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
      if (cb <= rb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          vec2 v;
          v[0] = -acc[cb][0][r];
          v[1] = -acc[cb][1][r];
          *reinterpret_cast<vec2 *>(smem + tri_blk(rb, cb) + P::drow(lane, r) * DB + rin) = v;
        }
      }
  }
  // (the strictly upper 16x16 blocks of the tile in HBM are never read by anybody and are left alone)
  __syncthreads();
  T *Wk = reinterpret_cast<T *>(p.Winv) + (size_t)b * p.winv_stride + (size_t)kn * WIMG;
  if (pc) pc->lap(p, 346);  // tile into LDS
  diag_factor_packed<T>(p, smem, tile, ld, Wk, b, kn, tid);
  if (pc) {
    pc->lap(p, 347);        // packed factorisation + inverse + stores
    pc->count(p, 351);
  }
}

// k_diag_lean: the diagonal tile in the LDS budget of a panel workgroup (two per CU), so that with
// two or more fits per CU one workgroup's factorisation latency runs under the other's MFMA loop.
template <typename T, bool FAT = false>
__global__ __launch_bounds__(256, 2) void k_diag_lean(FitArgs p, int k) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const int b = blockIdx.x;
  T *Lw = reinterpret_cast<T *>(p.Lw) + (size_t)b * p.lw_stride;
  typename Prec<T>::acc_t acc[NCB][2];
  diag_next<T, DIAG_FULL, kTriDiag, sizeof(T) == 8 || FAT, FAT>(p, acc, smem, Lw, b, k, threadIdx.x);
}

// --------------------------------------------------------------------------------------------------
// k_panel: L(rt, k) = (G(rt,k) - sum_{j<k} L(rt,j) L(k,j)^T) W_k^T   (a2 + a3 syrk/gemm + trsm + a8)
// --------------------------------------------------------------------------------------------------
// L(:, k) = S W_k^T on a tile held as acc = -S (layout of mfma_rowpanel_loop), W_k staged in LDS.
// Every wave must be done with `smem` before the call.
template <typename T>
__device__ __forceinline__ void trmm_in_registers(const FitArgs &p, typename Prec<T>::acc_t (&acc)[NCB][2], T *__restrict__ smem,
                                                  int b, int k, int tid, PhaseClock *pc = nullptr, int wslot = 0, bool live = true) {
  using P = Prec<T>;
  using acc_t = typename P::acc_t;
  const int lane = tid & 63, l15 = lane & 15;
  // -W_k's block image (WIMG) -> LDS, a straight copy by LDS-DMA: 1 KiB per wave-instruction, no VGPR
  // staging, no ds_write.  Wl[blk(cb,qb)][q][c] = -W[cb*16 + c][qb*16 + q].
  const T *__restrict__ Wk = reinterpret_cast<const T *>(p.Winv) + (size_t)b * p.winv_stride + (size_t)k * WIMG;
  {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int PER = 1024 / (int)sizeof(T);            // elements per wave-instruction
    constexpr int NI = WIMG / PER / 4;                     // wave-instructions per wave: 9 (fp32) / 18 (fp64)
    static_assert(WIMG % (4 * PER) == 0, "image is a whole number of 4 KiB rounds");
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int off = (i * 4 + wave) * PER;
      __builtin_amdgcn_global_load_lds((gbl_void *)(Wk + off + lane * (16 / (int)sizeof(T))), (lds_void *)(smem + off), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (pc) pc->lap(p, wslot);  // W_k staged
  if (!live) return;

  // in-register triangular product, descending column blocks so L(:, cb) may overwrite S(:, cb)
#pragma unroll
  for (int cb = NCB - 1; cb >= 0; --cb) {
    acc_t t0 = acc_t{0, 0, 0, 0}, t1 = t0;
    const T *wrow = smem + (cb * (cb + 1) / 2) * DB * DB + l15;
#pragma unroll
    for (int qb = 0; qb <= cb; ++qb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const T a = wrow[qb * DB * DB + P::drow(lane, r) * DB];  // -W[cb*16 + l15][qb*16 + drow(lane, r)] (acc = -S)
        t0 = P::mfma(a, acc[qb][0][r], t0);
        t1 = P::mfma(a, acc[qb][1][r], t1);
      }
    }
    acc[cb][0] = t0;
    acc[cb][1] = t1;
    __builtin_amdgcn_sched_barrier(0);  // one column block at a time: bounds the live W fragments
  }
}

// The same triangular product on the bf16 matrix cores (fp32 bf16-plane builds; see bx6_loop for the arithmetic).  A K = 32 MFMA
// takes 8 k-values per lane: lane (l15, g) of the B operand supplies its OWN accumulator registers of a PAIR of column blocks --
// acc[2m][j][0..3] and acc[2m + 1][j][0..3], i.e. k = (block 2m, q = 4g + r), (block 2m + 1, q = 4g + r) -- split into three
// planes in registers; the A operand of (column block cb, pair m) is -W's entries in that k order, split and laid out per lane by a
// conversion pass that all four waves share (fp32 image in global memory -> registers -> three planes in LDS, one 16-byte read per
// lane, plane and operand).  Eight terms (see trmm_bx6_pair), each its own instruction here (the pairing of bx6_compute needs 16
// k-values per term): 20 (cb, m) operands x 8 x 2 row blocks = 320 MFMAs of 16 cycles against 288 of 32.  Pairs in DESCENDING order: an input block
// belongs to one pair only, so once pair m's B planes are taken, acc[2m] and acc[2m + 1] are free to become the outputs L(:, 2m),
// L(:, 2m + 1), which later (lower) pairs keep adding to -- in place, as trmm_in_registers.  Two rounds (pairs 3, 2, 1: 12 operands
// = 36 KB of planes; pair 0: 8 = 24 KB) so that the planes fit the panel kernels' LDS.
// --------------------------------------------------------------------------------------------------
__device__ __forceinline__ void trmm_bx6_operand(int round, int ci, int &m, int &cb) {   // the ci-th (pair, column block) of a round
  if (round) { m = 0; cb = ci; }
  else if (ci < 2) { m = 3; cb = 6 + ci; }
  else if (ci < 6) { m = 2; cb = 2 + ci; }
  else { m = 1; cb = ci - 4; }
}
template <int M, int CI0>   // pair M, whose operands start at index CI0 of the round's planes
__device__ __forceinline__ void trmm_bx6_pair(Prec<float>::acc_t (&acc)[NCB][2], const unsigned short *planes, int lane) {
  using acc_t = Prec<float>::acc_t;
  bf8 bp[2][3];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float v[8] = {acc[2 * M][j][0], acc[2 * M][j][1], acc[2 * M][j][2], acc[2 * M][j][3],
                        acc[2 * M + 1][j][0], acc[2 * M + 1][j][1], acc[2 * M + 1][j][2], acc[2 * M + 1][j][3]};
    bxu4 w[3];
    bx_split8(v, w);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) bp[j][pl] = __builtin_bit_cast(bf8, w[pl]);
    acc[2 * M][j] = acc_t{0, 0, 0, 0};
    acc[2 * M + 1][j] = acc_t{0, 0, 0, 0};
  }
  // two column blocks at a time: four accumulators between two MFMAs on the same one
  static_for<0, (NCB - 2 * M) / 2>([&](auto hc) {
    constexpr int H = decltype(hc)::value, CB = 2 * M + 2 * H, CI = CI0 + 2 * H;
    bf8 ap[2][3];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) ap[u][pl] = bx_ld8(planes + (((CI + u) * 3 + pl) * 64 + lane) * 8);
    // EIGHT terms here, smallest first: a1 b2, a2 b1 (the two the tile loop drops), a0 b2, a2 b0, a1 b1, a0 b1, a1 b0, a0 b0 -- all but
    // a2 b2 (2^-32).  W is an explicit inverse: on the reference's RBF x Brownian windows (cond ~ 1e6) the product S W^T cancels
    // heavily, and with six terms the dropped 2^-22 |s| |w| per product put 3 of 2 600 fuzz cases past their fp32 bar (1.1-1.35 x).
    // With eight every product is exact to 2^-32 and only the fp32 accumulation rounds.
    // (Six terms for the SE kernels only, whose windows hold their bar with six, would be 2 % faster at 512 fits -- 134.9 k against
    // 132.0 k fits/s -- but neither form of the choice was usable: a per-term branch halved the kernel's speed, two straight-line
    // bodies behind one branch spilled 318 registers.)
    constexpr int NT8 = 8;
    constexpr int TA[NT8] = {1, 2, 0, 2, 1, 0, 1, 0}, TB[NT8] = {2, 1, 2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int t = 0; t < NT8; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[CB + u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[u][TA[t]], bp[j][TB[t]], acc[CB + u][j], 0, 0, 0);
  });
}
__device__ __forceinline__ void trmm_bx6(const FitArgs &p, Prec<float>::acc_t (&acc)[NCB][2], float *__restrict__ smem, int b, int k, int tid,
                                         PhaseClock *pc, int wslot, bool live) {
  const int lane = tid & 63, c = lane & 15, g = lane >> 4;
  const float *__restrict__ Wk = reinterpret_cast<const float *>(p.Winv) + (size_t)b * p.winv_stride + (size_t)k * WIMG;
  unsigned short *planes = reinterpret_cast<unsigned short *>(smem);
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    // conversion: operand ci of the round, lane (c, g): -W[cb*16 + c][(2m)*16 + 4g + r], -W[cb*16 + c][(2m+1)*16 + 4g + r], r = 0..3
    // (image: Wl[blk(cb, qb)][q][c]); block (cb, 2m + 1) of an even cb = 2m lies above the diagonal: zeros
    // 12 / 8 operands over four waves: three / two per wave, all their loads (L2 hits, ~1 us each way) issued before the first split
    constexpr int NOPW[2] = {3, 2};
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    float v[3][8];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i >= NOPW[round]) continue;
      int m, cb;
      trmm_bx6_operand(round, wv + 4 * i, m, cb);
      const float *w0 = Wk + (cb * (cb + 1) / 2 + 2 * m) * DB * DB + (4 * g) * DB + c;
      const bool second = 2 * m + 1 <= cb;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[i][r] = w0[r * DB];
        v[i][4 + r] = second ? w0[DB * DB + r * DB] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i >= NOPW[round]) continue;
      bxu4 w[3];
      bx_split8(v[i], w);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bxu4 *>(planes + (((wv + 4 * i) * 3 + pl) * 64 + lane) * 8) = w[pl];
    }
    __syncthreads();
    if (pc && round == 0) pc->lap(p, wslot);  // W_k staged (first round)
    if (live) {
      if (round == 0) {
        trmm_bx6_pair<3, 0>(acc, planes, lane);
        trmm_bx6_pair<2, 2>(acc, planes, lane);
        trmm_bx6_pair<1, 6>(acc, planes, lane);
      } else trmm_bx6_pair<0, 0>(acc, planes, lane);
    }
    if (round == 0) __syncthreads();   // every wave is done with the first round's planes
  }
}

// acc (rows 32 wave + 2 l15 + {0,1}, columns cb*16 + drow) <-> a column-major 128 x 128 tile, 16-byte accesses
template <typename T>
__device__ __forceinline__ void store_tile(const typename Prec<T>::acc_t (&acc)[NCB][2], T *__restrict__ tile, int ld, int tid) {
  using P = Prec<T>;
  using vec2 = T __attribute__((ext_vector_type(2)));
  const int lane = tid & 63, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  T *__restrict__ out = tile + wave * 32 + 2 * l15;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      vec2 v;
      v[0] = acc[cb][0][r];
      v[1] = acc[cb][1][r];
      *reinterpret_cast<vec2 *>(out + (size_t)(cb * DB + P::drow(lane, r)) * ld) = v;
    }
}
template <typename T, bool ADD>
__device__ __forceinline__ void load_tile(typename Prec<T>::acc_t (&acc)[NCB][2], const T *tile, int ld, int tid) {
  using P = Prec<T>;
  using vec2 = T __attribute__((ext_vector_type(2)));
  const int lane = tid & 63, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T *src = tile + wave * 32 + 2 * l15;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const vec2 v = *reinterpret_cast<const vec2 *>(src + (size_t)(cb * DB + P::drow(lane, r)) * ld);
      acc[cb][0][r] = ADD ? acc[cb][0][r] + v[0] : v[0];
      acc[cb][1][r] = ADD ? acc[cb][1][r] + v[1] : v[1];
    }
}

// k_panel fp64: row panel straight to registers (mfma_rowpanel_loop_rdirect).  The loop is written for
// both precisions, but fp32 measured 4.5 % slower with it (85.1k vs 89.1k fits/s at N = 1024: 156 VGPRs
// cost an occupancy step and a 512-byte column needs two 256-byte LDS-DMA instructions), so fp32
// keeps the register-staged loop.

// Kind C of k_panel<T, true, DEEP, MID> in launch k: tile (k + 2, k + 1) -- the NEXT launch's kind-A tile -- with the
// block columns < k (final since the previous launch): acc = -G + sum_{j<k} L(k+2, j) L(k+1, j)^T, left as a raw
// register image in p.pimg[(k + 1) & 1].  Same arithmetic, in the same order, as the one-workgroup form.
template <typename T, bool DEEP>
__device__ __forceinline__ void panel_partial(const FitArgs &p, typename Prec<T>::acc_t (&acc)[NCB][2], T *__restrict__ smem, int b,
                                              int k, int tid) {
  constexpr int CH2 = 2 * KT * LDST;
  const int kc = k + 1, rt = k + 2, ld = p.ld, nchunk = k * (TS / KT);
  const T *Lw = reinterpret_cast<const T *>(p.Lw) + (size_t)b * p.lw_stride;
  const T *gR = Lw + (size_t)rt * TS, *gC = Lw + (size_t)kc * TS;
  GramPre<T> gp;
  gram_prefetch<T>(p, b, kc, rt, tid, gp);
  if constexpr (DEEP) {
    constexpr int R = deep_ring<T, true>();   // kind C only exists in the mid-size build
    RowFrag<T, R> rf;
    T *zs = smem + R * KT * LDST;  // the loop folds its newest block column into running sums nobody reads here
    if (tid < TS) zs[tid] = T(0);
    T ms[4] = {T(0), T(0), T(0), T(0)};
    rdirect_prologue<T, R>(rf, gR, (size_t)ld, gC, (size_t)ld, nchunk, smem, tid);
    // the Gram inputs are staged in the ring slots the prologue leaves free (first written after iteration 0's barrier)
    gram_apply<T>(p, acc, smem + (R - 2) * KT * LDST, b, kc, rt, tid, gp);
    mfma_rowpanel_loop_rdirect<T, R>(acc, rf, gR, (size_t)ld, gC, (size_t)ld, nchunk, smem, tid, zs, ms);
  } else {
    if (nchunk > 0) stage_first_chunk<T>(gR, (size_t)ld, gC, (size_t)ld, smem, tid);
    gram_apply<T>(p, acc, smem + CH2, b, kc, rt, tid, gp);
    mfma_rowpanel_loop<T, true>(acc, gR, (size_t)ld, gC, (size_t)ld, nchunk, smem, tid);
  }
  acc_image<T, false, true>(acc, reinterpret_cast<T *>(p.pimg) + ((size_t)b * img_slots(p) + kc % img_slots(p)) * DPART, tid);
}

#ifndef CGP_IMG_EARLY
#define CGP_IMG_EARLY 1   // mid-size build: the finisher's image is requested in front of its fence (`make variant` A/B: 0)
#endif
template <typename T, bool MID> constexpr bool mid_fat() { return MID && sizeof(T) == 4; }
// One tile of block step k: L(rt, k) = (G(rt,k) - sum_{c_first*16 <= col < 128 k} L(rt,:) L(k,:)^T [+ image]) W_k^T, stored; then, kind A
// (finish_next), the next diagonal tile.  The body of k_panel after its role decode -- also what a tile task of the
// one-launch schedule (k_sched) runs.
template <typename T, bool DIAGNEXT, bool DEEP, bool MID>
__device__ __forceinline__ void panel_tile_body(const FitArgs &p, int k, int b, int rt, int c_first, bool finish_next, bool from_image,
                                                T *smem, int tid) {
  using P = Prec<T>;
  using acc_t = typename P::acc_t;
  acc_t acc[NCB][2];
  T *Lw = reinterpret_cast<T *>(p.Lw) + (size_t)b * p.lw_stride;
  const int ld = p.ld;
  PhaseClock pc;
  pc.start(p, tid);
  const int ps = finish_next ? 336 : 64 + 8 * (k & 31);  // debug slots of this step (kind A: its own group, all steps summed)

  // Gram first: acc = -G(rt, kc) while nothing else is live in the register file, with chunk 0 of the
  // panels already in flight into LDS buffer 0 (the Gram inputs are staged in buffer 1's space).
  constexpr int CH2 = 2 * KT * LDST;
  const T *gR = Lw + (size_t)rt * TS + (size_t)(c_first * KT) * ld, *gC = Lw + (size_t)k * TS + (size_t)(c_first * KT) * ld;
  // timing probes (wrong results): 4096 = every tile reads fit (b & 7)'s first row panel (L2-resident row
  // panels), 8192 = the same for the column panel
  if (CGP_DBG_ON(p, 4096)) gR = reinterpret_cast<const T *>(p.Lw) + (size_t)(b & 7) * p.lw_stride + (size_t)p.NT * TS;
  if (CGP_DBG_ON(p, 8192)) gC = reinterpret_cast<const T *>(p.Lw) + (size_t)(b & 7) * p.lw_stride + (size_t)k * TS;
  const int nchunk = k * (TS / KT) - c_first;
  const T *pimg = reinterpret_cast<const T *>(p.pimg) + ((size_t)b * img_slots(p) + k % img_slots(p)) * DPART;  // image of tile (k + 1, k)
  // running predictive sums (extra tiles, throughput schedule): z of the newest block column
  // (k - 1) goes to LDS behind the chunk ring; zeros when nothing is to be accumulated
  const bool accm = p.macc != nullptr && rt >= p.NT && k > 0;
  // the last extra tile of a fit is partly padding (601 rows in five tiles of 128): a wave whose 32 rows all lie beyond the y row
  // has nothing to compute.  Register-staged fp32 loop only (four workgroups per CU: the SIMD the idle wave frees serves three
  // other waves; in fp64, two per CU, skipping it measured 1.2 % SLOWER in round 2).
  const bool live = !(kSkipDeadWave && !DEEP && sizeof(T) == 4 && rt >= p.NT && !p.xid &&
                      (rt - p.NT) * TS + __builtin_amdgcn_readfirstlane(tid >> 6) * 32 > p.M);
  constexpr int R = DEEP ? deep_ring<T, MID>() : 4;   // chunk ring of the deep loop (the register-staged loop has two buffers)
  // fp32, full-batch and mid-size builds: products on the bf16 matrix cores (bx6_loop); BXD chunks in flight, two plane buffers in the mid-size build
  constexpr bool BX6 = kF32Bf16x6 && sizeof(T) == 4 && (!DEEP || MID);
  constexpr bool BXDBL = DEEP;
  constexpr int BXD = DEEP ? kBxMidSets : 1;
  constexpr bool PL = kMidPlanes && BX6 && MID;                 // panels arrive as bf16 planes by LDS-DMA (bx6p_loop): three chunk buffers
  constexpr int GRAM_OFF = PL ? 2 * BXP_BUF / 4 : (BX6 ? BX_FLOATS : CH2);   // where the Gram inputs are staged (BXDBL: over the second plane buffer; PL: the third)
  T *zs = smem + (PL ? BXP_NBUF * BXP_BUF / 4 : (BX6 ? bx_z_offset<BXDBL>() : R * KT * LDST));
  if (tid < TS) zs[tid] = accm ? Lw[(size_t)((k - 1) * TS + tid) * ld + (size_t)p.NT * TS + p.M] : T(0);
  T ms[4] = {T(0), T(0), T(0), T(0)};
  if constexpr (DEEP && !BX6) {
    RowFrag<T, R> rf;
    {
      GramPre<T> gp;
      if (!(MID && from_image)) gram_prefetch<T>(p, b, k, rt, tid, gp);
      rdirect_prologue<T, R>(rf, gR, (size_t)ld, gC, (size_t)ld, nchunk, smem, tid);
      if (MID && from_image) acc_image<T, false, false>(acc, const_cast<T *>(pimg), tid);
      else gram_apply<T>(p, acc, smem + (R - 2) * KT * LDST, b, k, rt, tid, gp, &pc, ps + 6);   // ring slots the prologue leaves free
    }
    pc.lap(p, ps + 0);
    mfma_rowpanel_loop_rdirect<T, R>(acc, rf, gR, (size_t)ld, gC, (size_t)ld, nchunk, smem, tid, zs, ms);
  } else {
    BxStage<PL ? 1 : BXD> bxs;   // (only the register-staged bf16-plane build uses it)
    BxPlaneSrc bps;              // (only the DMA build)
    {
      GramPre<T> gp;
      if (!(MID && from_image)) gram_prefetch<T>(p, b, k, rt, tid, gp);
      if (nchunk > 0) {
        if constexpr (PL) {
          bps.init(reinterpret_cast<const unsigned short *>(p.Lp) + (size_t)b * p.lp_stride, p.lp_stride / 3, ld, c_first, rt, k, tid);
          const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
          bps.issue(0, reinterpret_cast<unsigned char *>(smem), 0, wv);        // chunks 0 and 1: in flight through the Gram phase
          bps.issue(1, reinterpret_cast<unsigned char *>(smem), BXP_BUF, wv);
        } else if constexpr (BX6) {
          bxs.init(gR, (size_t)ld, gC, (size_t)ld, nchunk, tid);
          bx6_prologue(bxs, nchunk);   // in flight through the Gram phase
        } else stage_first_chunk<T>(gR, (size_t)ld, gC, (size_t)ld, smem, tid);
      }
      if (CGP_DBG_ON(p, 16384)) {  // timing probe: no Gram tile
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[cb][j][r] = gp.v[0];
      } else if (MID && from_image) acc_image<T, false, false>(acc, const_cast<T *>(pimg), tid);
      else gram_apply<T>(p, acc, smem + GRAM_OFF, b, k, rt, tid, gp, &pc, ps + 6, -1, live);
    }
    pc.lap(p, ps + 0);
    if constexpr (PL) bx6p_loop(acc, bps, nchunk, smem, tid, accm ? zs : nullptr, ms, live, gR, (size_t)ld);
    else if constexpr (BX6) bx6_loop<BXD, BXDBL>(acc, bxs, nchunk, smem, tid, accm ? zs : nullptr, ms, live);
    else mfma_rowpanel_loop<T, true>(acc, gR, (size_t)ld, gC, (size_t)ld, nchunk, smem, tid, accm ? zs : nullptr, ms, live);
  }
  pc.lap(p, ps + 1);
  if (accm) {
    // a row's 128 columns are spread over the four 16-lane groups: fold them, then the lanes of
    // group 0 add to the fit's accumulators (this workgroup is the only writer of its rows)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ms[q] += __shfl_xor(ms[q], 16);
      ms[q] += __shfl_xor(ms[q], 32);
    }
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (lane < 16) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = (rt - p.NT) * TS + wave * 32 + 2 * lane + j;
        if (m < p.M) {
          // agent-scope accesses: neighbouring rows' sums share cache lines with other workgroups' rows, and in the one-launch
          // schedule the previous block step's writer of THIS row may have run on another XCD (whose L2 is not coherent with ours)
          // block step 1 starts the sums (nothing zeroed them), later steps add
          double *pm = p.macc + (size_t)b * p.M + m, *pv = p.vacc + (size_t)b * p.M + m;
          const double om = k == 1 ? 0.0 : __hip_atomic_load(pm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const double ov = k == 1 ? 0.0 : __hip_atomic_load(pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(pm, om + (double)ms[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(pv, ov + (double)ms[2 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
  __syncthreads();  // every wave is done with the staged inputs before W_k overwrites them
  pc.lap(p, ps + 2);
  if (!CGP_DBG_ON(p, 64)) {
    if constexpr (kF32Bf16x6 && kBxTrmm && sizeof(T) == 4) trmm_bx6(p, acc, smem, b, k, tid, &pc, ps + 3, live);
    else trmm_in_registers<T>(p, acc, smem, b, k, tid, &pc, ps + 3, live);
  }
  pc.lap(p, ps + 4);
  if (live && (!CGP_DBG_ON(p, 32768) || acc[0][0][0] == T(12345.678)))  // timing probe: no store
  store_tile<T>(acc, Lw + (size_t)(k * TS) * ld + (size_t)rt * TS, ld, tid);
  if constexpr (PL) {   // ... and as bf16 planes, for the tiles that take this one as a panel (rows beyond the window's y row are never read)
    if (live) store_tile_planes(acc, reinterpret_cast<unsigned short *>(p.Lp) + (size_t)b * p.lp_stride, p.lp_stride / 3, ld, rt, k, tid);
  }
  pc.lap(p, ps + 5);
  pc.count(p, ps + 7);
  if constexpr (DIAGNEXT) {
    if (finish_next) {
      // The pre-updated image of the diagonal tile (kind B, two launches ago) does not depend on the tile this workgroup has just
      // stored: mid-size build, it is requested HERE, in front of the finish's workgroup fence, whose wait for the tile's stores then
      // covers its latency too (the accumulators are free once the stores are issued).
      bool img_ready = false;
      if constexpr (MID && CGP_IMG_EARLY) {
        if (k + 1 >= 3) {
          const T *img = reinterpret_cast<const T *>(p.dpart) + ((size_t)b * img_slots(p) + (k + 1) % img_slots(p)) * DPART;
          acc_image<T, kTriDiag, false>(acc, const_cast<T *>(img), tid);
          img_ready = true;
        }
      }
      diag_next<T, DIAG_FINISH, kTriDiag, DEEP, mid_fat<T, MID>()>(p, acc, smem, Lw, b, k + 1, tid, &pc, img_ready);
    }
  }
}

// DIAGNEXT (throughput schedule, fit launches): besides the row tiles below the diagonal and the extra
// tiles the launch carries, per fit, the workgroup that finishes the next diagonal tile (kind A: row tile
// k + 1, then DIAG_FINISH of tile k + 1), the one that pre-updates the one after (kind B: DIAG_PARTIAL of
// tile k + 2) and -- calls too small to fill the chip, where a launch lasts as long as its kind-A chain -- the
// one that pre-updates the NEXT launch's kind-A tile (kind C: tile (k + 2, k + 1) with the block columns < k,
// left as a register image in p.pimg; the next launch's kind A then only adds block column k, so the chain of
// a block step no longer grows with k).  Kinds A, B, C get the lowest linear block ids, i.e. are dispatched
// first, so their longer chains end inside the launch.  grid.x = A + B + C + other tiles;
// p.diag_slots = {1: has A, 2: has B, 4: has C, 8: kind A starts from the image of the previous launch's C}.
// DEEP: row panel straight to registers + LDS-DMA column panel two chunks ahead (fp64 always; fp32 for calls
// that leave most CUs with one or two workgroups, where nothing else hides the memory latency -- with four
// workgroups per CU the register-staged loop is 4.5 % faster: 128 instead of 156 VGPRs).
// MID: kinds C / image-A compiled in (their branches cost registers the full-batch build cannot spare).  In fp32 the
// mid-size build also takes the fat form of the diagonal tile (79 KB of LDS: two workgroups per CU, which these calls do
// not fill anyway); fp64 would need 147 KB, one workgroup per CU, and keeps the packed form.
#ifndef CGP_KINDA_PRIO
#define CGP_KINDA_PRIO 0   // s_setprio of the kind-A workgroup's waves in the mid-size build (0: none; `make variant` A/B)
#endif
#ifndef CGP_F32_FULL_OCC
#define CGP_F32_FULL_OCC 4   // workgroups per CU the register-staged fp32 build is compiled for: 128 VGPRs -- the tile loop has none spilled (the
                             // Gram / diagonal-tile code around it has: 85 in the fused kernel) -- and 38 KB of LDS; at three per CU (153 VGPRs, no
                             // spill) the bf16-plane build measured 5 % slower (121.8 k against 127.9 k fits/s)
#endif
constexpr int F32_FULL_OCC = CGP_F32_FULL_OCC;
template <typename T, bool DIAGNEXT = false, bool DEEP = sizeof(T) == 8, bool MID = false>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? (DEEP ? ((MID && (CGP_MID_RING > 4 || CGP_F32_BF16X6)) ? 2 : 3) : F32_FULL_OCC) : 2) void k_panel(FitArgs p, int k) {
  using P = Prec<T>;
  using acc_t = typename P::acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const int tid = threadIdx.x;
  int bt, b, rt;
  int c_first = 0;             // first 16-column chunk of the inner dimension this workgroup applies
  bool finish_next = false, from_image = false;
  acc_t acc[NCB][2];
  SpanClock sc;
  sc.enter(p, k, tid);
  if constexpr (DIAGNEXT) {
    const int Tg = gridDim.x, B = gridDim.y;
    const int lin = blockIdx.y * Tg + blockIdx.x;
    const int nA = (p.diag_slots & 1) ? B : 0, nB = (p.diag_slots & 2) ? B : 0, nC = (MID && (p.diag_slots & 4)) ? B : 0;
    // Dispatch follows the linear block id and an XCD hands consecutive workgroups to different CUs, so ids
    // are dealt in blocks of 256 (8 XCDs x 32 CUs): every `stride`-th block (stride = workgroups per CU)
    // is a block of kind-A workgroups and the blocks between hold the other tiles.  A CU then hosts one
    // latency-bound diagonal factorisation next to MFMA-bound tiles instead of several at once.  Speed only.
    int ia = -1, io = lin;  // index among the A kinds / among everything else
    const int stride = p.diag_stride, ablk = nA >> 8, oblk = (Tg * B - nA) >> 8;
    if (nA > 0 && (B & 255) == 0 && stride > 0 && (ablk - 1) * stride < ablk + oblk) {
      const int j = lin >> 8, r = lin & 255;
      const int before = min((j + stride - 1) / stride, ablk);  // A blocks at positions < j
      if (j % stride == 0 && j / stride < ablk) ia = (j / stride) * 256 + r;
      else io = (j - before) * 256 + r;
    } else if (lin < nA) ia = lin;
    else io = lin - nA;
    if (CGP_DBG_ON(p, 65536) && (ia >= 0 || io < nB)) {  // timing probe: no diagonal work in the launch
      if (ia < 0) return;
    }
    if (ia >= 0) {
      b = ia;
      rt = k + 1;
      finish_next = !CGP_DBG_ON(p, 65536);
      // The kind-A workgroup IS the block step's critical path in a call that does not fill the chip (tile (k + 1, k), then the
      // diagonal tile: 77 us alone, 86 ... 95 us beside the bulk tile its CU also hosts); its waves take the issue slots first.
      if constexpr (MID && CGP_KINDA_PRIO > 0) __builtin_amdgcn_s_setprio(CGP_KINDA_PRIO);
      if (MID && (p.diag_slots & 8)) {  // tile (k + 1, k) minus block column k - 1 is waiting in the image
        from_image = true;
        c_first = (k - 1) * (TS / KT);
      }
    } else if (io < nB) {
      b = io;
      T *LwB = reinterpret_cast<T *>(p.Lw) + (size_t)b * p.lw_stride;
      diag_next<T, DIAG_PARTIAL, kTriDiag, DEEP>(p, acc, smem, LwB, b, k + 2, tid);
      sc.leave(p, k, tid);
      return;
    } else if (MID && io < nB + nC) {
      panel_partial<T, DEEP>(p, acc, smem, io - nB, k, tid);  // kind C (a path of its own, as kind B: the main path keeps its registers)
      sc.leave(p, k, tid);
      return;
    } else {
      const int l2 = io - nB - nC, To = Tg - (nA ? 1 : 0) - (nB ? 1 : 0) - (nC ? 1 : 0);  // the other tiles, XCD-steered
      if ((B & 7) == 0) {
        const int xcd = l2 & 7, slot = l2 >> 3;
        bt = slot % To;
        b = (slot / To) * 8 + xcd;
      } else {
        bt = l2 % To;
        b = l2 / To;
      }
      const int first = k + 1 + (nA ? 1 : 0);  // first in-matrix row tile among the others
      const int nin = p.NT - first > 0 ? p.NT - first : 0;
      rt = bt < nin ? first + bt : p.NT + (bt - nin);
    }
  } else {
    tile_fit_of_block(bt, b);
    rt = row_tile_of(bt + p.tile_off, k + 1, p.NT, p.rows_from_extra);
  }
  panel_tile_body<T, DIAGNEXT, DEEP, MID>(p, k, b, rt, c_first, finish_next, from_image, smem, tid);
  sc.leave(p, k, tid);
}

// --------------------------------------------------------------------------------------------------
// k_rows64: the EXTRA rows (test points and y) of block step k in 64-row tiles -- four waves of 16 rows -- for the fp32
// mid-size calls (BASELINE configs[2] as sharded: 64 fits per GPU), whose extra-row launches E(k) run on a second stream
// beside the factorisation launches (run_schedule).  A mid-size launch is a handful of equal workgroups per CU and lasts as
// long as the CU with the most of them: E(k) as 128-row tiles is 5 x 64 = 320 workgroups on 256 CUs -- 64 CUs take two, the
// launch lasts two tile times for 1.25 tile times of work (E(7): 143 us against an MFMA floor of 67; round 5 timeline,
// profiles/r05_mid_timelines.txt).  Half-height tiles halve the quantum: 640 workgroups, at most three per CU.
// Same arithmetic per element, in the same order, as panel_tile_body (rows are independent): bitwise the 128-row result.
//   acc[cb][reg] = C[row = 16 wave + (lane & 15)][col = 16 cb + drow(lane, reg)]
// Column panel through the LDS-DMA ring (4 slots, two chunks ahead), a wave's own 16 rows straight to registers.
// --------------------------------------------------------------------------------------------------
constexpr int HR = TS / 2;   // rows of a half tile
template <typename T, int SLOT>
__device__ __forceinline__ void r16_load(T (&f)[4][KT / 4], const T *gRl, size_t ldR, int chunk, int lq) {
#pragma unroll
  for (int ks = 0; ks < KT / 4; ++ks) {
    const T *src = gRl + (size_t)(chunk * KT + ks * 4 + lq) * ldR;
    if constexpr (sizeof(T) == 8) asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(f[SLOT][ks]) : "v"(src));
    else asm volatile("global_load_dword %0, %1, off" : "=&v"(f[SLOT][ks]) : "v"(src));
  }
}
template <typename T, int S, bool ISSUE, int WAITN, bool ACC>
__device__ __forceinline__ void r16_step(typename Prec<T>::acc_t (&acc)[NCB][1], T (&f)[4][KT / 4], const T *gRl, size_t ldR, const T *gC,
                                         size_t ldC, int c, T *smem, int lane, int wave, const T *zs, T *ms) {
  using P = Prec<T>;
  constexpr int CH = KT * LDST;
  const int l15 = lane & 15, lq = lane >> 4;
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WAITN) : "memory");
#pragma unroll
  for (int ks = 0; ks < KT / 4; ++ks) asm volatile("" : "+v"(f[S][ks]));
  if constexpr (ISSUE) {
    cpanel_stage<T>(gC, ldC, c + 2, smem + ((S + 2) & 3) * CH, lane, wave);
    r16_load<T, (S + 2) & 3>(f, gRl, ldR, c + 2, lq);
  }
  const T *cur = smem + S * CH + lq * LDST + l15;
  T fa[2][NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) fa[0][cb] = cur[cb * DB];
#pragma unroll
  for (int ks = 0; ks < KT / 4; ++ks) {
    if (ks + 1 < KT / 4) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) fa[(ks + 1) & 1][cb] = cur[(ks + 1) * 4 * LDST + cb * DB];
    }
    const T fb = f[S][ks];
    if constexpr (ACC) {
      const T zv = zs[ks * 4 + lq];
      ms[0] = __builtin_fma(fb, zv, ms[0]);
      ms[1] = __builtin_fma(fb, fb, ms[1]);
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb][0] = P::mfma(fa[ks & 1][cb], fb, acc[cb][0]);
  }
}

template <typename T>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? 4 : 2) void k_rows64(FitArgs p, int k) {
  using P = Prec<T>;
  using acc_t = typename P::acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int h, b;
  tile_fit_of_block(h, b);                     // a fit's half tiles on one XCD: they share the column panel
  const int row0 = h * HR;                     // first extra row of this half tile (rows > M are padding, row M is y)
  T *Lw = reinterpret_cast<T *>(p.Lw) + (size_t)b * p.lw_stride;
  const int ld = p.ld;
  constexpr int CH = KT * LDST, L = cpanel_loads<T>() + KT / 4;
  const size_t rb = (size_t)p.NT * TS + row0;  // first row of the tile in the factor panel
  const T *gR = Lw + rb, *gC = Lw + (size_t)k * TS;
  const int nchunk = k * (TS / KT);
  const bool accm = p.macc != nullptr && k > 0;
  T *zs = smem + 4 * CH;
  if (tid < TS) zs[tid] = accm ? Lw[(size_t)((k - 1) * TS + tid) * ld + (size_t)p.NT * TS + p.M] : T(0);
  T ms[2] = {T(0), T(0)};
  acc_t acc[NCB][1];
  T rf[4][KT / 4];
  const T *gRl = gR + wave * DB + l15;
  {
    GramPre<T> gp;
    gram_prefetch<T>(p, b, k, p.NT, tid, gp, row0);
    if (nchunk > 0) {
      cpanel_stage<T>(gC, (size_t)ld, 0, smem, lane, wave);
      r16_load<T, 0>(rf, gRl, (size_t)ld, 0, lq);
      cpanel_stage<T>(gC, (size_t)ld, 1, smem + CH, lane, wave);
      r16_load<T, 1>(rf, gRl, (size_t)ld, 1, lq);
    }
    gram_apply<T, false, 1>(p, acc, smem + 2 * CH, b, k, p.NT, tid, gp, nullptr, 0, row0);
  }
  if (nchunk > 0) {
    int c = 0;
    for (; c + 8 < nchunk; c += 4) {
      r16_step<T, 0, true, L, false>(acc, rf, gRl, (size_t)ld, gC, (size_t)ld, c, smem, lane, wave, nullptr, nullptr);
      r16_step<T, 1, true, L, false>(acc, rf, gRl, (size_t)ld, gC, (size_t)ld, c + 1, smem, lane, wave, nullptr, nullptr);
      r16_step<T, 2, true, L, false>(acc, rf, gRl, (size_t)ld, gC, (size_t)ld, c + 2, smem, lane, wave, nullptr, nullptr);
      r16_step<T, 3, true, L, false>(acc, rf, gRl, (size_t)ld, gC, (size_t)ld, c + 3, smem, lane, wave, nullptr, nullptr);
    }
    static_for<0, 8>([&](auto jc) {   // the newest block column: the predictive sums ride along
      constexpr int J = decltype(jc)::value;
      r16_step<T, J & 3, (J + 2 < 8), (J < 7 ? L : 0), true>(acc, rf, gRl, (size_t)ld, gC, (size_t)ld, c + J, smem, lane, wave, zs + J * KT, ms);
    });
  }
  if (accm) {
    ms[0] += __shfl_xor(ms[0], 16);
    ms[0] += __shfl_xor(ms[0], 32);
    ms[1] += __shfl_xor(ms[1], 16);
    ms[1] += __shfl_xor(ms[1], 32);
    const int m = row0 + wave * DB + lane;
    if (lane < 16 && m < p.M) {
      double *pm = p.macc + (size_t)b * p.M + m, *pv = p.vacc + (size_t)b * p.M + m;
      const double om = k == 1 ? 0.0 : __hip_atomic_load(pm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double ov = k == 1 ? 0.0 : __hip_atomic_load(pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(pm, om + (double)ms[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(pv, ov + (double)ms[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();   // every wave is done with the staged inputs before W_k overwrites them
  // L(:, k) = S W_k^T in registers (trmm_in_registers with one row block per wave)
  const T *__restrict__ Wk = reinterpret_cast<const T *>(p.Winv) + (size_t)b * p.winv_stride + (size_t)k * WIMG;
  {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    constexpr int PER = 1024 / (int)sizeof(T), NI = WIMG / PER / 4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int off = (i * 4 + wave) * PER;
      __builtin_amdgcn_global_load_lds((gbl_void *)(Wk + off + lane * (16 / (int)sizeof(T))), (lds_void *)(smem + off), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
#pragma unroll
  for (int cb = NCB - 1; cb >= 0; --cb) {
    acc_t t0 = acc_t{0, 0, 0, 0};
    const T *wrow = smem + (cb * (cb + 1) / 2) * DB * DB + l15;
#pragma unroll
    for (int qb = 0; qb <= cb; ++qb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) t0 = P::mfma(wrow[qb * DB * DB + P::drow(lane, r) * DB], acc[qb][0][r], t0);
    }
    acc[cb][0] = t0;
    __builtin_amdgcn_sched_barrier(0);
  }
  T *__restrict__ out = Lw + (size_t)(k * TS) * ld + rb + wave * DB + l15;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(size_t)(cb * DB + P::drow(lane, r)) * ld] = acc[cb][0][r];
}

#ifdef CGP_AB
// --------------------------------------------------------------------------------------------------
// k_sched: a whole mid-size call (a few dozen fits: BASELINE configs[2] as sharded over 8 GPUs is 64 per GPU) in ONE launch.
// As NT + 1 launches such a call is bound by what happens at the launch boundaries: every launch lasts as long as its
// longest workgroup chain, carries an integer number of equal workgroups per CU (768 ... 320 tiles on 256 CUs), and all
// its workgroups enter their input / Gram phases together and leave through trmm / store together, so the matrix
// pipes idle at both ends of all eight launches (MFMA floor of the 64-fit fp32 call 0.46 ms, as
// launches 0.98).  Here the same tile programs run as TASKS of one persistent launch: workgroups pull the next task of
// a list ordered by block step (kinds: A tile (k+1,k) + diagonal tile k+1; B / C the pre-updates
// of k_panel<T, true, DEEP, true>; T every other tile), wait for what it reads -- per fit: `prog[rt]` = block columns of
// row tile rt stored so far, `wdone` = diagonal tiles factored, one flag per pre-update image -- and publish what they
// wrote.  A fit's steps no longer wait for the other fits' (the chain of fit b's diagonal tiles runs beside the tiles of
// the others), a CU's workgroups drift out of phase, and there is one wind-down per call.
// Ordering: every task a task waits for is EARLIER in the list, and a workgroup only holds a task while it is resident,
// so the earliest unfinished task can always run -- no assumption on residency, placement or dispatch order.  Hand-offs
// follow the guide's counter form: producer  s_waitcnt vmcnt(0) (every wave) -> barrier -> lane 0: agent-scope release,
// s_waitcnt vmcnt(0), relaxed agent-scope flag stores;  consumer  lane 0 polls with relaxed agent-scope loads
// (s_sleep between polls, bounded: a timeout raises `abort`, every workgroup leaves and info reports it), one
// agent-scope acquire, barrier, plain loads.  Pre-update images take one slot per tile index (FitArgs::img_slots), so
// every address of the call is written once.  Same arithmetic in the same order as the launches: bitwise the same results.
// --------------------------------------------------------------------------------------------------
enum { SCHED_T = 0, SCHED_A = 1, SCHED_B = 2, SCHED_C = 3 };
struct SchedArgs {
  const int4 *tasks;   // {kind, block step k, fit b, row tile rt}
  int ntasks;
  int *head;           // next task
  int *abort;          // a wait timed out: everybody leaves
  int *prog;           // [fits][prog_stride] block columns stored per row tile
  int prog_stride;
  int *wdone;          // [fits] diagonal tiles factored (W_k published for k < wdone)
  int *bflag, *cflag;  // [fits][flag_stride] pre-update images written (kind B: by diagonal tile index, kind C: by block column)
  int flag_stride;
};
constexpr unsigned SCHED_SPIN_LIMIT = 1u << 19;   // polls before a wait gives up (~ a second; a call is milliseconds)

__device__ __forceinline__ bool sched_wait_ge(int *addr, int want, int *abort) {
  for (unsigned spins = 0;; ++spins) {
    if (__hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
    if ((spins & 31) == 31 && __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    if (spins >= SCHED_SPIN_LIMIT) {
      __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(64);
  }
}

template <typename T>
__device__ __attribute__((noinline)) void sched_run_tile(const FitArgs &p, int k, int b, int rt, int c_first, int finish, int img, T *smem, int tid) {
  panel_tile_body<T, true, true, true>(p, k, b, rt, c_first, finish != 0, img != 0, smem, tid);
}
template <typename T>
__device__ __attribute__((noinline)) void sched_run_diag_partial(const FitArgs &p, T *smem, int b, int kn, int tid) {
  typename Prec<T>::acc_t acc[NCB][2];
  diag_next<T, DIAG_PARTIAL, kTriDiag, true>(p, acc, smem, reinterpret_cast<T *>(p.Lw) + (size_t)b * p.lw_stride, b, kn, tid);
}
template <typename T>
__device__ __attribute__((noinline)) void sched_run_panel_partial(const FitArgs &p, T *smem, int b, int k, int tid) {
  typename Prec<T>::acc_t acc[NCB][2];
  panel_partial<T, true>(p, acc, smem, b, k, tid);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void k_sched(FitArgs p, SchedArgs q) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  __shared__ int s_task, s_ok;
  __shared__ FitArgs sp;   // the arguments where an out-of-line tile program can reach them (a kernel argument has no address)
  if (threadIdx.x == 0) sp = p;
  for (;;) {
    // the thread index is re-read as an opaque value every trip: otherwise everything the tile programs derive from it (lane
    // offsets, operand addresses) is hoisted out of the task loop and kept live across it -- 326 (fp32) / 856 (fp64) spilled VGPRs
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    __syncthreads();   // everybody is done with the previous task (LDS, s_task)
    if (tid == 0) {
      int t = q.ntasks;
      if (__hip_atomic_load(q.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        t = __hip_atomic_fetch_add(q.head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_task = t;
    }
    __syncthreads();
    const int t = s_task;
    if (t >= q.ntasks) return;
    const int4 tk = q.tasks[t];
    const int kind = tk.x, k = tk.y, b = tk.z, rt = tk.w;
    int *prog = q.prog + (size_t)b * q.prog_stride, *wdone = q.wdone + b;
    int *bflag = q.bflag + (size_t)b * q.flag_stride, *cflag = q.cflag + (size_t)b * q.flag_stride;
    if (tid == 0) {
      // `wdone` counts the diagonal tiles factored INSIDE this launch (tile 0 has a launch of its own ahead of it), so
      // W_k is published when wdone >= k
      bool ok = true;
      if (kind == SCHED_T) {
        ok = sched_wait_ge(wdone, k, q.abort) && (k == 0 || sched_wait_ge(prog + rt, k, q.abort));
        // an extra tile with running predictive sums also reads z of block column k - 1: the y row's tile
        if (ok && k > 0 && rt >= p.NT && p.macc != nullptr) ok = sched_wait_ge(prog + p.NT + p.M / TS, k, q.abort);
      } else if (kind == SCHED_A) {
        ok = sched_wait_ge(wdone, k, q.abort) && (k == 0 || sched_wait_ge(prog + k + 1, k, q.abort)) &&
             (k < 1 || sched_wait_ge(cflag + k, 1, q.abort)) && (k < 2 || sched_wait_ge(bflag + k + 1, 1, q.abort));
      } else if (kind == SCHED_B) {
        ok = sched_wait_ge(prog + k + 2, k, q.abort);
      } else {
        ok = k == 0 || (sched_wait_ge(prog + k + 2, k, q.abort) && sched_wait_ge(prog + k + 1, k, q.abort));
      }
      if (ok) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // this CU's L1 holds nothing older than what the flags announced
      else atomicCAS(p.info + b, 0, -(1000 + k));                 // reported: the call fails with a negative info, it does not hang
      s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    if (!s_ok) return;
    // the tile programs are OUT OF LINE (sched_run_*): inlined into the task loop they shared one register allocation with it and
    // with each other (127 fp32 / 568 fp64 spilled VGPRs, some inside the hand-counted `s_waitcnt vmcnt(n)` loops, whose counts a
    // spill's scratch access silently breaks); as functions each is allocated as its own kernel is
    if (kind == SCHED_B) sched_run_diag_partial<T>(sp, smem, b, k + 2, tid);
    else if (kind == SCHED_C) sched_run_panel_partial<T>(sp, smem, b, k, tid);
    else {
      const bool chain = kind == SCHED_A, img = chain && k >= 1;
      sched_run_tile<T>(sp, k, b, rt, img ? (k - 1) * (TS / KT) : 0, chain ? 1 : 0, img ? 1 : 0, smem, tid);
    }
    // publish: every wave's stores have left the CU, then ONE lane's agent-scope release ahead of the flags
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (kind == SCHED_T) __hip_atomic_store(prog + rt, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (kind == SCHED_A) {
        __hip_atomic_store(prog + k + 1, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(wdone, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (kind == SCHED_B) __hip_atomic_store(bflag + k + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(cflag + k + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

#endif  // CGP_AB (k_sched)

#ifdef CGP_AB
// --------------------------------------------------------------------------------------------------
// k_diag: the diagonal tile of block step k, updated, factored and inverted by one workgroup.
// --------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_diag(FitArgs p, int k) {
  using P = Prec<T>;
  using acc_t = typename P::acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const int b = blockIdx.x;
  T *Lw = reinterpret_cast<T *>(p.Lw) + (size_t)b * p.lw_stride;
  const int ld = p.ld;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;

  const long long tstart = __builtin_amdgcn_s_memtime();
  constexpr int CH2 = 2 * KT * LDST;
  acc_t acc[NCB][2];
  const T *gR = Lw + (size_t)k * TS;
  const int nchunk = (k * TS) / KT;
  // fp64: triangular update (9 of 16 MFMA tiles per wave, panel staged once, prefetch distance 2).
  // fp32: the full-tile loop -- its MFMAs are half as long, the loop is bound by the staging
  // latency either way and the triangular form measured 13 % slower there.
  constexpr bool TRI = sizeof(T) == 8;
  {
    GramPre<T> gp;
    gram_prefetch<T>(p, b, k, k, tid, gp);
    if constexpr (TRI) {
      if (nchunk > 0) stage_chunk_tri<T>(gR, (size_t)ld, 0, smem + tri_buf(0), tid);
      if (nchunk > 1) stage_chunk_tri<T>(gR, (size_t)ld, 1, smem + tri_buf(1), tid);
    } else if (nchunk > 0) stage_first_chunk<T>(gR, (size_t)ld, gR, (size_t)ld, smem, tid);
    gram_apply<T, TRI>(p, acc, smem + CH2, b, k, k, tid, gp);
  }
  if constexpr (TRI) mfma_syrk_tri_loop<T>(acc, gR, (size_t)ld, nchunk, smem, tid);
  else mfma_rowpanel_loop<T, true>(acc, gR, (size_t)ld, gR, (size_t)ld, nchunk, smem, tid);
  __syncthreads();

  T *At = smem;                // element (r, c) at At[c * LDP + r]
  T *Dv = At + TS * LDP;
  T *Ts = Dv + 8 * DB * DB;
  int *flag = reinterpret_cast<int *>(Ts + 4 * DB * DB);
  if constexpr (TRI) {
    // lower 16x16 blocks only (wave w: block rows w and 7 - w); the factorisation never reads the others
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rb = j ? NCB - 1 - wave : wave;
        if (cb <= rb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) At[(cb * DB + P::drow(lane, r)) * LDP + rb * DB + l15] = -acc[cb][j][r];
        }
      }
  } else {
    using vec2 = T __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        vec2 v;
        v[0] = -acc[cb][0][r];
        v[1] = -acc[cb][1][r];
        *reinterpret_cast<vec2 *>(At + (cb * DB + P::drow(lane, r)) * LDP + wave * 32 + 2 * l15) = v;
      }
  }
  if (tid == 0) *flag = 0;
  __syncthreads();
  const long long tq = __builtin_amdgcn_s_memtime();
  potf2_tile<T>(p, At, Dv, Ts, flag, Lw + (size_t)(k * TS) * ld + (size_t)k * TS, ld, b, k, tid);
  if (CGP_DBG_ON(p, 512) && tid == 0 && b == 0) { p.dbgbuf[5] = __builtin_amdgcn_s_memtime() - tq; p.dbgbuf[6] = tq - tstart; }
}

#endif  // CGP_AB

// --------------------------------------------------------------------------------------------------
// Latency schedule (a handful of fits: the throughput schedule would leave most CUs idle and walk
// every tile's inner dimension in one workgroup).  Per block step two launches:
//   k_tile_sk(k) : grid (1 + tiles, fits, SK).  Tile slot 0 is the diagonal tile, the others the
//                  panel / extra tiles of step k; the inner dimension (128 k columns) is cut into SK
//                  ranges, one workgroup each.  Range 0 starts from -G (Gram first), the others from
//                  0.  Partial tiles go to a scratch slab; the workgroup that arrives last (atomic
//                  ticket) adds them in fixed order 0..SK-1 -- results do not depend on arrival
//                  order -- and finishes the tile: the diagonal tile is factored and inverted in
//                  LDS (potf2_tile), the others are stored as -S into their slot of the panel.
//                  The diagonal update and the panel updates of a step thus run side by side.
//   k_trmm_sk(k) : L(i,k) = S(i,k) W_k^T in registers, as the tail of k_panel.
//                  (CGP_SK_TRMM=fused: the other finishers of k_tile_sk wait -- bounded -- for W_k and
//                  apply it themselves, one launch per step; 2 % faster, kept for A/B only.)
// --------------------------------------------------------------------------------------------------
struct SplitArgs {
  void *part;       // [fits][slots][SK_MAX][128*128] partial tiles
  int *ticket;      // [fits][slots], zero between launches (the finishing workgroup resets it)
  int slots;        // tile slots per fit in `part` / `ticket`
  int sk;           // ranges the inner dimension is cut into (1..SK_MAX)
  int has_diag;     // fit: slot 0 is the diagonal tile; predict-only: extra tiles only
  int *wready;      // [fits] number of block steps whose W_k is published (fit schedule); see k_tile_sk
  void *diag_img;   // [fits][2][LAT_IMG_MAX][DPART] pre-updated diagonal tiles (register images), by parity of the tile
  int fuse_trmm;    // finishers of the non-diagonal tiles wait for W_k and apply it (one launch per step)
};
constexpr int SK_MAX = 8;

// Chunks of one pre-update workgroup of the diagonal tile (latency schedule): 3 block columns, so that it is
// shorter than the factorisation it hides behind.
constexpr int LAT_IMG_CHUNKS = 3 * (TS / KT);
constexpr int LAT_IMG_MAX = 6;  // images per diagonal tile: 3 block columns each, the last one is added by the finisher: N <= (LAT_IMG_MAX * 3 + 2) * 128 = 2560
__host__ __device__ __forceinline__ constexpr int lat_images(int kn) {  // images of diagonal tile kn: columns < (kn-1)*128
  return kn >= 2 ? ((kn - 1) * (TS / KT) + LAT_IMG_CHUNKS - 1) / LAT_IMG_CHUNKS : 0;
}

template <typename T>
__global__ __launch_bounds__(256) void k_tile_sk(FitArgs p, SplitArgs q, int k) {
  using P = Prec<T>;
  using acc_t = typename P::acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  __shared__ int s_last;
  const int t = blockIdx.x, b = blockIdx.y, sp = blockIdx.z;
  const bool diag = q.has_diag && t == 0;
  const int rt = diag ? k : row_tile_of(t - (q.has_diag ? 1 : 0), k + 1, p.NT, p.rows_from_extra);
  T *Lw = reinterpret_cast<T *>(p.Lw) + (size_t)b * p.lw_stride;
  const int ld = p.ld;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  constexpr int CH2 = 2 * KT * LDST;

  // phase clock of the workgroup that finishes the DIAGONAL tile (CGP_DBG & 1024, measurement build):
  // slots 8.. = Gram / images + last block column, -, -, LDS fill, factorisation + stores; slot 15 = count
  PhaseClock pc;
  pc.start(p, tid);
  acc_t acc[NCB][2];

  if (diag) {
    // ---- the diagonal tile is the per-step critical path (its factorisation is serial), so its update never
    // goes through the split-K slabs: the block columns < k - 1 were pre-applied during the PREVIOUS step's
    // launch (workgroups z >= 1 below, hidden behind that step's factorisation) and left as ONE register image
    // (their fixed-order sum); this workgroup adds block column k - 1 (final since the trmm launch in between)
    // on the triangular loop and factors.  Same arithmetic order whatever shares the launch.
    T *imgs = reinterpret_cast<T *>(q.diag_img) + (size_t)b * 2 * LAT_IMG_MAX * DPART;
    if (sp >= 1) {
      const int kn = k + 1, s = sp - 1;  // pre-update of the NEXT diagonal tile, image s
      if (kn >= p.NT || s >= lat_images(kn)) return;
      const int c0 = s * LAT_IMG_CHUNKS, c1 = min((s + 1) * LAT_IMG_CHUNKS, (kn - 1) * (TS / KT));
      const T *gR = Lw + (size_t)kn * TS + (size_t)(c0 * KT) * ld;
      const int nchunk = c1 - c0;
      GramPre<T> gp;
      if (s == 0) gram_prefetch<T>(p, b, kn, kn, tid, gp);
      stage_chunk_tri<T>(gR, (size_t)ld, 0, smem + tri_buf(0), tid);
      if (sizeof(T) == 8 && nchunk > 1) stage_chunk_tri<T>(gR, (size_t)ld, 1, smem + tri_buf(1), tid);
      if (s == 0) gram_apply<T, true>(p, acc, smem + CH2, b, kn, kn, tid, gp);
      else {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb][0] = acc[cb][1] = acc_t{0, 0, 0, 0};
      }
      mfma_syrk_tri_loop<T>(acc, gR, (size_t)ld, nchunk, smem, tid);
      T *im = imgs + (size_t)(kn & 1) * LAT_IMG_MAX * DPART;
      acc_image<T, true, true>(acc, im + (size_t)s * DPART, tid);
      const int nimg_next = lat_images(kn);
      if (nimg_next > 1) {
        // the last pre-update workgroup to arrive folds the images into image 0 in fixed order 0, 1, 2, ... (still
        // hidden behind this step's factorisation), so the next step's diagonal workgroup loads ONE image
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          int *tk = q.ticket + b * q.slots;  // slot 0 = the diagonal tile, which no longer uses the slab tickets
          const int old = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_last = old == nimg_next - 1;
          if (s_last) {
            __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          }
        }
        __syncthreads();
        if (s_last) {
          acc_image<T, true, false>(acc, im, tid);
          for (int s2 = 1; s2 < nimg_next; ++s2) {
            acc_t add[NCB][2];
            acc_image<T, true, false>(add, im + (size_t)s2 * DPART, tid);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                if (cb <= (j ? NCB - 1 - wave : wave)) acc[cb][j] += add[cb][j];
          }
          acc_image<T, true, true>(acc, im, tid);
        }
      }
      pc.lap(p, 24);  // a pre-update workgroup, whole life
      pc.count(p, 25);
      return;
    }
    const int nimg = lat_images(k);
    const int c0 = k >= 1 ? (k - 1) * (TS / KT) : 0, nchunk = k * (TS / KT) - c0;
    const T *gR = Lw + (size_t)k * TS + (size_t)(c0 * KT) * ld;
    {
      GramPre<T> gp;
      if (nimg == 0) gram_prefetch<T>(p, b, k, k, tid, gp);
      if (nchunk > 0) stage_chunk_tri<T>(gR, (size_t)ld, 0, smem + tri_buf(0), tid);
      if (sizeof(T) == 8 && nchunk > 1) stage_chunk_tri<T>(gR, (size_t)ld, 1, smem + tri_buf(1), tid);
      if (nimg == 0) gram_apply<T, true>(p, acc, smem + CH2, b, k, k, tid, gp);
      else {
        acc_image<T, true, false>(acc, imgs + (size_t)(k & 1) * LAT_IMG_MAX * DPART, tid);  // already the fixed-order sum
      }
    }
    mfma_syrk_tri_loop<T>(acc, gR, (size_t)ld, nchunk, smem, tid);
    __syncthreads();
    pc.lap(p, 8);
    T *At = smem;  // element (r, c) at At[c * LDP + r]
    T *Dv = At + TS * LDP;
    T *Ts = Dv + 8 * DB * DB;
    int *flag = reinterpret_cast<int *>(Ts + 4 * DB * DB);
    // lower 16x16 blocks only (wave w: block rows w and 7 - w); the factorisation never reads the others
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rb = j ? NCB - 1 - wave : wave;
        if (cb <= rb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) At[(cb * DB + P::drow(lane, r)) * LDP + rb * DB + l15] = -acc[cb][j][r];
        }
      }
    if (tid == 0) *flag = 0;
    __syncthreads();
    pc.lap(p, 11);  // tile into LDS
    potf2_tile<T>(p, At, Dv, Ts, flag, Lw + (size_t)(k * TS) * ld + (size_t)k * TS, ld, b, k, tid);
    pc.lap(p, 12);  // factorisation, inverse, stores
    pc.count(p, 15);
    __threadfence();   // W_k (and L(k,k)) visible device-wide before the step is announced
    __syncthreads();
    if (tid == 0) __hip_atomic_store(q.wready + b, k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }

  // ---- panel / extra tiles: inner dimension split over q.sk workgroups, ticketed fixed-order reduction
  if (sp >= q.sk) return;  // the grid's z extent also covers the diagonal tile's pre-update workgroups
  if (sp == 0) {
    GramPre<T> gp;
    gram_prefetch<T>(p, b, k, rt, tid, gp);
    gram_apply<T>(p, acc, smem + CH2, b, k, rt, tid, gp);
    __syncthreads();
  } else {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb][0] = acc[cb][1] = acc_t{0, 0, 0, 0};
  }
  // Ranges of the inner dimension: range 0 also builds the Gram tile, which costs about as much as
  // GRAM_UNITS chunks, so the n chunks + GRAM_UNITS are split evenly and range 0 gets that many fewer
  // chunks.  A function of (k, sk) only: the summation order does not depend on what shares the launch.
  const int n = (k * TS) / KT;
  constexpr int GRAM_UNITS = 4;
  auto bound = [&](int i) {
    if (q.sk == 1) return i == 0 ? 0 : n;
    const int u = (int)((long long)i * (n + GRAM_UNITS) / q.sk) - GRAM_UNITS;
    return i >= q.sk ? n : (u < 0 ? 0 : u);
  };
  const int c0 = sp == 0 ? 0 : bound(sp), c1 = bound(sp + 1);
  const T *gR = Lw + (size_t)rt * TS + (size_t)(c0 * KT) * ld, *gC = Lw + (size_t)k * TS + (size_t)(c0 * KT) * ld;
  mfma_rowpanel_loop<T, false>(acc, gR, (size_t)ld, gC, (size_t)ld, c1 - c0, smem, tid);
  pc.lap(p, sp == 0 ? 16 : 17);  // panel tiles: Gram + update (range 0) / update (other ranges), every workgroup
  pc.count(p, sp == 0 ? 22 : 23);

  if (q.sk > 1) {
    T *slab = reinterpret_cast<T *>(q.part) + ((size_t)(b * q.slots + t) * SK_MAX) * TS * TS;
    store_tile<T>(acc, slab + (size_t)sp * TS * TS, TS, tid);
    // publish: every wave's stores have left the CU, then ONE lane's agent-scope release ahead of the ticket
    // (the guide's counter form of the hand-off; 256 threads fencing cost 2-4x one lane's)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int old = __hip_atomic_fetch_add(q.ticket + b * q.slots + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = old == q.sk - 1;
      if (s_last) {
        __hip_atomic_store(q.ticket + b * q.slots + t, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the last arriver reads the other slabs below
      }
    }
    __syncthreads();
    if (!s_last) return;
    pc.lap(p, 18);  // last arriver: slab store + fence + ticket
    // fixed-order sum 0 .. sk-1 with the next partial tile's loads in flight while the current one is added
    using vec2 = T __attribute__((ext_vector_type(2)));
    const T *src = slab + wave * 32 + 2 * l15;
    vec2 nx[NCB][4];
    auto fetch = [&](int s2) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          nx[cb][r] = *reinterpret_cast<const vec2 *>(src + (size_t)s2 * TS * TS + (size_t)(cb * DB + P::drow(lane, r)) * TS);
    };
    fetch(0);
    for (int s2 = 0; s2 < q.sk; ++s2) {
      vec2 cur[NCB][4];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) cur[cb][r] = nx[cb][r];
      if (s2 + 1 < q.sk) fetch(s2 + 1);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[cb][0][r] = s2 ? acc[cb][0][r] + cur[cb][r][0] : cur[cb][r][0];
          acc[cb][1][r] = s2 ? acc[cb][1][r] + cur[cb][r][1] : cur[cb][r][1];
        }
    }
  }
  __syncthreads();
  pc.lap(p, 19);  // reduction

  if (!q.fuse_trmm) {
    store_tile<T>(acc, Lw + (size_t)(k * TS) * ld + (size_t)rt * TS, ld, tid);
    pc.lap(p, 20);
    pc.count(p, 21);
    return;
  }
  // L(rt, k) = S W_k^T here as well (CGP_SK_TRMM=fused, measurement build): W_k comes from the diagonal tile's
  // workgroup of THIS launch.  Its workgroup has been dispatched by now or will be without this one's help
  // (at most one waiting workgroup per tile, far fewer than CUs), so waiting cannot deadlock; the wait is
  // bounded all the same and a timeout is reported through info[] instead of hanging the GPU.
  if (q.has_diag) {
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(q.wready + b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) <= k && ++spins < (1 << 22))
        __builtin_amdgcn_s_sleep(8);
      if (spins >= (1 << 22)) atomicCAS(p.info + b, 0, -(k + 1));
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  trmm_in_registers<T>(p, acc, smem, b, k, tid);
  store_tile<T>(acc, Lw + (size_t)(k * TS) * ld + (size_t)rt * TS, ld, tid);
}

// k_trmm_sk: L(i,k) = S(i,k) W_k^T for the tiles of step k, on the latency schedule's critical path (every tile
// of the next step needs its result), so a tile is cut into TRMM_SPLIT row slabs, one workgroup (one CU's MFMA
// pipe) each, 16 rows per wave, and the W image's LDS-DMA is issued first, in flight while the S rows are
// loaded into the accumulators.  In-kernel clocks of the one-workgroup form: loads 6.7 k, product 13.4 k (the
// tile's 1152 MFMAs on one CU), stores 6.1 k cycles.
#ifndef CGP_TRMM_SPLIT
#define CGP_TRMM_SPLIT 2
#endif
constexpr int TRMM_SPLIT = CGP_TRMM_SPLIT;
constexpr int TRMM_WAVES = TS / DB / TRMM_SPLIT;  // 4 waves of 16 rows per workgroup
template <typename T>
__global__ __launch_bounds__(64 * TRMM_WAVES) void k_trmm_sk(FitArgs p, int k) {
  using P = Prec<T>;
  using acc_t = typename P::acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slab = blockIdx.x % TRMM_SPLIT;
  const int rt = row_tile_of(blockIdx.x / TRMM_SPLIT, k + 1, p.NT, p.rows_from_extra);
  T *Lw = reinterpret_cast<T *>(p.Lw) + (size_t)b * p.lw_stride;
  const int ld = p.ld;
  T *tile = Lw + (size_t)(k * TS) * ld + (size_t)rt * TS + (slab * TRMM_WAVES + wave) * DB + l15;  // this lane's row
  const T *__restrict__ Wk = reinterpret_cast<const T *>(p.Winv) + (size_t)b * p.winv_stride + (size_t)k * WIMG;
  PhaseClock pc;  // measurement build: slots 44.. = loads landed, product, stores issued; 47 = count
  pc.start(p, tid);
  {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    constexpr int PER = 1024 / (int)sizeof(T);
#pragma unroll
    for (int i = 0; i < (WIMG / PER + TRMM_WAVES - 1) / TRMM_WAVES; ++i) {
      const int blk = i * TRMM_WAVES + wave;
      if (blk < WIMG / PER)
        __builtin_amdgcn_global_load_lds((gbl_void *)(Wk + blk * PER + lane * (16 / (int)sizeof(T))), (lds_void *)(smem + blk * PER), 16, 0, 0);
    }
  }
  acc_t acc[NCB];  // acc[cb][r] = -S[row][cb*16 + drow(lane, r)]
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[cb][r] = tile[(size_t)(cb * DB + P::drow(lane, r)) * ld];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  pc.lap(p, 44);
#pragma unroll
  for (int cb = NCB - 1; cb >= 0; --cb) {
    acc_t t0 = acc_t{0, 0, 0, 0};
    const T *wrow = smem + (cb * (cb + 1) / 2) * DB * DB + l15;
#pragma unroll
    for (int qb = 0; qb <= cb; ++qb)
#pragma unroll
      for (int r = 0; r < 4; ++r) t0 = P::mfma(wrow[qb * DB * DB + P::drow(lane, r) * DB], acc[qb][r], t0);
    acc[cb] = t0;
    __builtin_amdgcn_sched_barrier(0);
  }
  pc.lap(p, 45);
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[(size_t)(cb * DB + P::drow(lane, r)) * ld] = acc[cb][r];
  pc.lap(p, 46);
  pc.count(p, 47);
}

// --------------------------------------------------------------------------------------------------
// k_grad (GPy's dL_dK = 0.5 (alpha alpha^T - Ky^-1) contracted with dK/dtheta):
// after a gradient-mode factorisation (xid = 1, M = N) the extra block of the factor panel holds
// Wt = (L^-1)^T, so Ky^-1 = Wt Wt^T is a syrk of its row panels -- the same MFMA loop as the
// update.  Workgroup (pair, fit) forms the 128x128 tile (ti >= tj) of Ky^-1 in registers and reduces
//     S_amp  = sum w_ij K_ij        S_ell[q] = sum w_ij K_ij d_q^2        S_noise = sum_i w_ii
// with w = alpha_i alpha_j - Ky^-1_ij, d_q the length-scaled coordinate difference; off-diagonal
// tiles count twice.  Wt[e][c] = 0 for c < e, so the inner dimension starts at column block ti.
// --------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void k_grad(FitArgs p, int npairs) {
  using P = Prec<T>;
  using acc_t = typename P::acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const int b = blockIdx.y, pair = blockIdx.x;
  int ti = 0, rem = pair;
  while (rem > ti) {
    rem -= ti + 1;
    ++ti;
  }
  const int tj = rem;  // ti >= tj
  const T *Lw = reinterpret_cast<const T *>(p.Lw) + (size_t)b * p.lw_stride;
  const int ld = p.ld, N = p.N, d = p.d, kid = p.kernel_id;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  const size_t rb = (size_t)p.NT * TS;

  acc_t acc[NCB][2];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) acc[cb][0] = acc[cb][1] = acc_t{0, 0, 0, 0};
  const T *gR = Lw + rb + (size_t)ti * TS + (size_t)(ti * TS) * ld;
  const T *gC = Lw + rb + (size_t)tj * TS + (size_t)(ti * TS) * ld;
  mfma_rowpanel_loop<T, false>(acc, gR, (size_t)ld, gC, (size_t)ld, (p.NT - ti) * (TS / KT), smem, tid);

  // inputs of the tile: scaled coordinates [MAXD][128] of rows and columns, alpha of both
  const double *__restrict__ pr = p.prep + (size_t)b * PREP_N;
  T *xr = smem, *xc = smem + MAXD * TS, *ar = smem + 2 * MAXD * TS, *ac = ar + TS;
  const T *__restrict__ Xb = reinterpret_cast<const T *>(p.X) + (size_t)b * d * N;
  const T *__restrict__ al = reinterpret_cast<const T *>(p.alpha) + (size_t)b * p.alpha_stride;
  const bool brown = kid == K_RBF_BROWNIAN;
  for (int idx = tid; idx < MAXD * TS; idx += 256) {
    const int q = idx >> 7, r = idx & 127;
    const T sc = brown ? T(1) : T(pr[q]);  // Brownian keeps the raw tick (GPy's r^2 expansion)
    const int gi = ti * TS + r, gj = tj * TS + r;
    xr[idx] = (q < d && gi < N) ? Xb[(size_t)q * N + gi] * sc : T(0);
    xc[idx] = (q < d && gj < N) ? Xb[(size_t)q * N + gj] * sc : T(0);
  }
  if (tid < TS) {
    const int gi = ti * TS + tid, gj = tj * TS + tid;
    ar[tid] = gi < N ? al[gi] : T(0);
    ac[tid] = gj < N ? al[gj] : T(0);
  }
  __syncthreads();
  const T amp = T(pr[9]), amp_b = T(pr[10]);
  const T inv_ell = T(pr[0]);
  double s_amp = 0, s_noise = 0, s_ell[MAXD];
#pragma unroll
  for (int q = 0; q < MAXD; ++q) s_ell[q] = 0;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cl = cb * DB + P::drow(lane, r);
      const int gcol = tj * TS + cl;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rl = wave * 32 + 2 * l15 + j;
        const int grow = ti * TS + rl;
        const T kinv = acc[cb][j][r];
        if (grow < N && gcol < N) {
          const T w = ar[rl] * ac[cl] - kinv;
          T kv, dq2[MAXD];
          if (!brown) {
            T d2 = 0;
#pragma unroll
            for (int q = 0; q < MAXD; ++q) {
              const T df = xr[q * TS + rl] - xc[q * TS + cl];
              dq2[q] = df * df;
              d2 += dq2[q];
            }
            kv = amp * P::exp_(T(-0.5) * d2);
          } else {
            const T x = xr[rl], xp = xc[cl];
            T r2 = (grow == gcol) ? T(0) : (T(-2) * x * xp + (x * x + xp * xp));
            r2 = r2 < T(0) ? T(0) : r2;
            const T rr = P::sqrt_(r2) * inv_ell;
            const int sx = (x > T(0)) - (x < T(0)), sp = (xp > T(0)) - (xp < T(0));
            const T ax = x < T(0) ? -x : x, ap = xp < T(0) ? -xp : xp;
            const T kb = (sx == sp) ? amp_b * (ax < ap ? ax : ap) : T(0);
            kv = amp * P::exp_(T(-0.5) * rr * rr) * kb;
#pragma unroll
            for (int q = 0; q < MAXD; ++q) dq2[q] = T(0);
            dq2[0] = rr * rr;
          }
          const double wk = (double)w * (double)kv;
          s_amp += wk;
#pragma unroll
          for (int q = 0; q < MAXD; ++q) s_ell[q] += wk * (double)dq2[q];
          if (grow == gcol) s_noise += (double)w;
        }
      }
    }
  }
  // workgroup reduction: wave shuffles, then LDS
  __syncthreads();
  double *red = reinterpret_cast<double *>(smem_raw);  // [4][GRAD_N]
  double vals[GRAD_N];
  vals[0] = s_amp;
#pragma unroll
  for (int q = 0; q < MAXD; ++q) vals[1 + q] = s_ell[q];
  vals[9] = s_noise;
  vals[10] = vals[11] = 0;
#pragma unroll
  for (int i = 0; i < GRAD_N; ++i) {
    double v = vals[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) red[wave * GRAD_N + i] = v;
  }
  __syncthreads();
  if (tid < GRAD_N) {
    const double wgt = (ti == tj) ? 1.0 : 2.0;
    const double v = (red[tid] + red[GRAD_N + tid]) + (red[2 * GRAD_N + tid] + red[3 * GRAD_N + tid]);
    p.gpart[((size_t)b * npairs + pair) * GRAD_N + tid] = wgt * v;
  }
}

}  // namespace cgp
