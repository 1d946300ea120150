// cgp_lookahead.hpp -- batched stop-time look-ahead on the GPU: the loop of
// GpPredictor::GPCallBack (gp_predictor/src/gp_predictor.cpp:58-130) for a whole Monte-Carlo
// ensemble at once, one 64-lane wave per trajectory, the 15x15 filter matrices in LDS.
// The host class (csrc/gp_predictor_core.cpp) stays the single-trajectory path and the checker's
// counterpart; both follow the same statement order.
//
// The covariance propagation P <- F P F' + Q (4 of every 5 steps are nothing else) runs on the matrix
// cores without touching LDS: with the 15x15 matrices padded to 16x16, the C/D layout of
// v_mfma_f64_16x16x4_f64 (lane (n, g) register r holds element [g + 4r][n]) is, for a SYMMETRIC
// matrix, also its A-operand layout, and for any matrix the B-operand layout of its transpose's
// rows -- so  T = P F'  and  P = F T + Q  are 4 + 4 MFMAs on registers that never leave the wave
// (P: 4 accumulator registers, F: 4 operand registers, Q: 4).  Only the measurement update of every
// 5th step goes through LDS.
// Per IMU step (5 per odometry tick, :64):  P <- F P F' + Q (:66); every 5th step the unscented
// transform of 0.8/(1-slip) at {mu, mu+-sigma} (:69-78) gives R (:80-88), K = P H'(H P H' + R)^-1 and
// the Joseph update (:90-91); then the +3 sigma LLH point goes through llh_to_enu (:95-99) and the
// loop stops at the first step whose horizontal error exceeds the threshold (:102-121).
#pragma once
#include <hip/hip_runtime.h>

namespace cgp {

struct LookaheadArgs {
  const double *mean, *sigma;  // [ntraj][M]
  const double *P, *Q, *STM;   // [ntraj][225] row-major
  const double *Hvec;          // [ntraj][60]  SetStopping.HvecData
  const double *pos;           // [ntraj][3]   LLH of the rover at the end of the window
  const double *trig;          // [ntraj][4]   sin/cos(lat), sin/cos(lon) of pos (host libm)
  const double *arrival, *now; // [ntraj]
  double *stop_cmd, *xy_err;   // [ntraj]
  int *fired, *i_out;          // [ntraj]
  int ntraj, M, h_bug_compatible;
  double threshold;
  double init_llh[3], init_ecef[3];
  double origin_trig[4];       // sin/cos(init lat), sin/cos(init lon)
};

constexpr int LA_NS = 15, LA_NM = 4, LA_WAVES = 4;
constexpr int LA_PER_WAVE = 4 * 225 + 4 * 60 + 64;  // P, F, Q, T | H, PHt, K, KR | small scratch

// llh_to_enu (gp_predictor.cpp:144-178) of a point displaced by (dphi, dlam) from the trajectory's
// base position: the base angles' sines / cosines are computed once per trajectory, the displaced ones
// by angle addition with a short series for the small displacement (3 sigma of the position error,
// ~1e-6 rad; exact to 1e-18 up to 0.5 rad, clamped beyond -- the horizontal error is then thousands
// of kilometres and the threshold test fires either way).  The double-precision libm sin / cos / tan
// this replaces cost 256 VGPRs + 100 AGPRs + 272 B of scratch (large-argument reduction) and held the
// kernel at one wave per SIMD.
struct LaGeo {
  double sphi, cphi, slam, clam;  // base position
  double sP, cP, sL, cL;          // ENU origin (init_llh)
};
__device__ __forceinline__ void la_sincos_small(double d, double &sd, double &cd) {
  d = fmin(fmax(d, -0.5), 0.5);
  const double d2 = d * d;
  double ps = -1.0 / 1307674368000.0;  // -1/15!
  ps = __builtin_fma(ps, d2, 1.0 / 6227020800.0);
  ps = __builtin_fma(ps, d2, -1.0 / 39916800.0);
  ps = __builtin_fma(ps, d2, 1.0 / 362880.0);
  ps = __builtin_fma(ps, d2, -1.0 / 5040.0);
  ps = __builtin_fma(ps, d2, 1.0 / 120.0);
  ps = __builtin_fma(ps, d2, -1.0 / 6.0);
  sd = __builtin_fma(ps * d2, d, d);
  double pc = 1.0 / 20922789888000.0;  // 1/16!
  pc = __builtin_fma(pc, d2, -1.0 / 87178291200.0);
  pc = __builtin_fma(pc, d2, 1.0 / 479001600.0);
  pc = __builtin_fma(pc, d2, -1.0 / 3628800.0);
  pc = __builtin_fma(pc, d2, 1.0 / 40320.0);
  pc = __builtin_fma(pc, d2, -1.0 / 720.0);
  pc = __builtin_fma(pc, d2, 1.0 / 24.0);
  pc = __builtin_fma(pc, d2, -0.5);
  cd = __builtin_fma(pc, d2, 1.0);
}
__device__ __forceinline__ void la_enu_displaced(const LaGeo &g, double dphi, double dlam, double h, const double *iecef,
                                                 double &e0, double &e1) {
  const double a = 6378137.0000, b = 6356752.3142;
  const double e2 = 1.0 - (b / a) * (b / a);
  double sdp, cdp, sdl, cdl;
  la_sincos_small(dphi, sdp, cdp);
  la_sincos_small(dlam, sdl, cdl);
  const double sinphi = __builtin_fma(g.sphi, cdp, g.cphi * sdp), cosphi = __builtin_fma(g.cphi, cdp, -(g.sphi * sdp));
  const double sinlam = __builtin_fma(g.slam, cdl, g.clam * sdl), coslam = __builtin_fma(g.clam, cdl, -(g.slam * sdl));
  const double tanphi = sinphi / cosphi;
  const double tmp2 = 1.0 - e2;
  const double tmpden = sqrt(1.0 + tmp2 * tanphi * tanphi);
  const double x1 = (a * coslam) / tmpden + h * coslam * cosphi;
  const double y1 = (a * sinlam) / tmpden + h * sinlam * cosphi;
  const double tmp3 = sqrt(1.0 - e2 * sinphi * sinphi);
  const double z1 = (a * tmp2 * sinphi) / tmp3 + h * sinphi;
  const double dx = x1 - iecef[0], dy = y1 - iecef[1], dz = z1 - iecef[2];
  e0 = -g.sL * dx + g.cL * dy;
  e1 = -g.sP * g.cL * dx - g.sP * g.sL * dy + g.cP * dz;
}

__device__ __forceinline__ double la_rdlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// C(n x m) = A(n x k) B(k x m) or A B^T, row-major in LDS, outputs strided over the wave's lanes.
// All lanes of the wave call it; the trailing wave barrier orders the LDS traffic.
template <bool BT>
__device__ __forceinline__ void la_mm(const double *A, const double *B, double *C, int n, int k, int m, int lane,
                                      const double *add = nullptr, double scale = 1.0) {
  for (int o = lane; o < n * m; o += 64) {
    const int r = o / m, c = o - r * m;
    double s = 0.0;
    for (int q = 0; q < k; ++q) s += A[r * k + q] * (BT ? B[c * k + q] : B[q * m + c]);
    C[o] = scale * s + (add ? add[o] : 0.0);
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

__global__ __launch_bounds__(64 * LA_WAVES) void k_lookahead(LookaheadArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tr = blockIdx.x * LA_WAVES + wave;
  if (tr >= p.ntraj) return;
  double *base = reinterpret_cast<double *>(smem_raw) + (size_t)wave * LA_PER_WAVE;
  double *P = base, *F = P + 225, *Q = F + 225, *T = Q + 225;
  double *H = T + 225, *PHt = H + 60, *K = PHt + 60, *KR = K + 60, *sc = KR + 60;  // sc: 64 scratch doubles
  for (int o = lane; o < 225; o += 64) {
    P[o] = p.P[(size_t)tr * 225 + o];
    F[o] = p.STM[(size_t)tr * 225 + o];
    Q[o] = p.Q[(size_t)tr * 225 + o];
  }
  if (lane < 60) {
    const int r = lane / LA_NS, c = lane - r * LA_NS;
    H[lane] = p.Hvec[(size_t)tr * 60 + (p.h_bug_compatible ? r * 4 + c : r * LA_NS + c)];  // :38-42
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const double *mean = p.mean + (size_t)tr * p.M, *sigma = p.sigma + (size_t)tr * p.M;
  const double hgt = p.pos[tr * 3 + 2];
  LaGeo geo;
  geo.sphi = p.trig[tr * 4]; geo.cphi = p.trig[tr * 4 + 1]; geo.slam = p.trig[tr * 4 + 2]; geo.clam = p.trig[tr * 4 + 3];
  geo.sP = p.origin_trig[0]; geo.cP = p.origin_trig[1]; geo.sL = p.origin_trig[2]; geo.cL = p.origin_trig[3];
  double e00, e01;
  la_enu_displaced(geo, 0.0, 0.0, hgt, p.init_ecef, e00, e01);  // :95, loop invariant
  const double R1[16] = {0.5, 0.5, 0.0, 0.0, 1 / 0.685, -1 / 0.685, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0};
  int i = 0, fired = 0;
  double xy = 0.0, cmd = 0.0;
  // register-resident 16x16 (zero padded) forms, see the header comment
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int l15 = lane & 15, lg = lane >> 4;
  double Fop[4];
  d4 Pd, Qd;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kk = 4 * r + lg, row = lg + 4 * r;
    Fop[r] = (l15 < LA_NS && kk < LA_NS) ? F[l15 * LA_NS + kk] : 0.0;        // F[l15][4r + lg]
    Pd[r] = (row < LA_NS && l15 < LA_NS) ? P[row * LA_NS + l15] : 0.0;       // P[lg + 4r][l15]
    Qd[r] = (row < LA_NS && l15 < LA_NS) ? Q[row * LA_NS + l15] : 0.0;
  }
  for (int slip_i = 0; slip_i < 5 * p.M; ++slip_i) {   // :64
    {                                                   // :66  P = F P F' + Q
      d4 Td = d4{0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) Td = __builtin_amdgcn_mfma_f64_16x16x4f64(Pd[ks], Fop[ks], Td, 0, 0, 0);  // T = P F'
      d4 Pn = Qd;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) Pn = __builtin_amdgcn_mfma_f64_16x16x4f64(Fop[ks], Td[ks], Pn, 0, 0, 0);  // F T + Q
      Pd = Pn;
    }
    if (slip_i % 5 == 0) {                              // :67
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = lg + 4 * r;
        if (row < LA_NS && l15 < LA_NS) P[row * LA_NS + l15] = Pd[r];
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const double c0 = mean[i], c1 = mean[i] + sigma[i], c2 = mean[i] - sigma[i];
      const double o0 = 0.8 / (1.0 - c0), o1 = 0.8 / (1.0 - c1), o2 = 0.8 / (1.0 - c2);
      const double est = (o0 + o1 + o2) / 3.0;
      const double cov = ((o0 - est) * (o0 - est) + (o1 - est) * (o1 - est) + (o2 - est) * (o2 - est)) / 3.0;
      double R2[4] = {fmax(0.03 * 0.03, cov * cov), fmax(0.03 * 0.03, cov * cov), fmax(0.05 * 0.05, cov * cov), 0.05 * 0.05};
      double R[16];  // R = 25 R1 R2 R1'  (every lane computes it: 4x4)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double s = 0.0;
#pragma unroll
          for (int q = 0; q < 4; ++q) s += R1[r * 4 + q] * R2[q] * R1[c * 4 + q];
          R[r * 4 + c] = 25.0 * s;
        }
      la_mm<true>(P, H, PHt, LA_NS, LA_NS, LA_NM, lane);        // P H'   (15 x 4)
      la_mm<false>(H, PHt, sc, LA_NM, LA_NS, LA_NM, lane);      // H P H' (4 x 4) -> sc[0..15]
      // S^-1 by Gauss-Jordan with partial pivoting, redundantly in every lane's registers
      double a[4][8];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          a[r][c] = sc[r * 4 + c] + R[r * 4 + c];
          a[r][4 + c] = (r == c) ? 1.0 : 0.0;
        }
      // fully unrolled with compile-time indices (the pivot row is swapped in by selects), so the
      // 4 x 8 tableau stays in registers instead of scratch memory
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        int piv = c;
        double best = fabs(a[c][c]);
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
          const double v = fabs(a[r][c]);
          if (v > best) {   // same rule as the host path: first strictly larger magnitude wins
            best = v;
            piv = r;
          }
        }
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
          const bool sw = piv == r;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const double t = a[c][j], u = a[r][j];
            a[c][j] = sw ? u : t;
            a[r][j] = sw ? t : u;
          }
        }
        const double d = 1.0 / a[c][c];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[c][j] *= d;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (r == c) continue;
          const double f = a[r][c];
#pragma unroll
          for (int j = 0; j < 8; ++j) a[r][j] = (f != 0.0) ? a[r][j] - f * a[c][j] : a[r][j];
        }
      }
      __builtin_amdgcn_wave_barrier();
      {
        double siv = 0.0, rv = 0.0;   // lane e < 16 picks Si[e / 4][e % 4] and R[e] with static indices
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          siv = (lane == e) ? a[e >> 2][4 + (e & 3)] : siv;
          rv = (lane == e) ? R[e] : rv;
        }
        if (lane < 16) {
          sc[16 + lane] = siv;  // Si
          sc[32 + lane] = rv;
        }
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      la_mm<false>(PHt, sc + 16, K, LA_NS, LA_NM, LA_NM, lane);   // K = P H' S^-1  (15 x 4)   :90
      la_mm<false>(K, H, T, LA_NS, LA_NM, LA_NS, lane);           // K H
      for (int o = lane; o < 225; o += 64) T[o] = ((o / LA_NS) == (o % LA_NS) ? 1.0 : 0.0) - T[o];  // I - K H
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      la_mm<false>(K, sc + 32, KR, LA_NS, LA_NM, LA_NM, lane);    // K R
      // Joseph form (:91):  P = G P G' + (K R) K'  with G = I - K H, on the matrix cores like the
      // propagation: T' = P G' (P symmetric: its C/D registers are the A operand), then G T' on top of
      // the one-k-step product (K R) K'.
      {
        double Gop[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kk = 4 * r + lg;
          Gop[r] = (l15 < LA_NS && kk < LA_NS) ? T[l15 * LA_NS + kk] : 0.0;   // G[l15][4r + lg]
        }
        const double kra = l15 < LA_NS ? KR[l15 * LA_NM + lg] : 0.0;           // (K R)[l15][lg]
        const double kb = l15 < LA_NS ? K[l15 * LA_NM + lg] : 0.0;             // K[l15][lg]
        d4 Pn = __builtin_amdgcn_mfma_f64_16x16x4f64(kra, kb, d4{0, 0, 0, 0}, 0, 0, 0);
        d4 Td = d4{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) Td = __builtin_amdgcn_mfma_f64_16x16x4f64(Pd[ks], Gop[ks], Td, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) Pn = __builtin_amdgcn_mfma_f64_16x16x4f64(Gop[ks], Td[ks], Pn, 0, 0, 0);
        Pd = Pn;
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      ++i;                                                // :92
    }
    // P[6][6], P[7][7], P[8][8]: element [g + 4r][n] lives in lane 16 g + n, register r
    const double p66 = la_rdlane(Pd[1], 2 * 16 + 6), p77 = la_rdlane(Pd[1], 3 * 16 + 7), p88 = la_rdlane(Pd[2], 8);
    const double s6 = 3.0 * sqrt(fabs(p66)), s7 = 3.0 * sqrt(fabs(p77)), s8 = 3.0 * sqrt(fabs(p88));
    double e30, e31;
    la_enu_displaced(geo, s6, s7, hgt + s8, p.init_ecef, e30, e31);  // :97
    xy = sqrt((e30 - e00) * (e30 - e00) + (e31 - e01) * (e31 - e01));                  // :99
    if (xy > p.threshold) {                               // :102  (wave-uniform)
      const double dt = p.arrival[tr] + i / 10.0 - p.now[tr];
      cmd = dt < 0.0 ? 0.5 : dt;                          // :107-117
      fired = 1;
      break;
    }
  }
  if (lane == 0) {
    p.fired[tr] = fired;
    p.stop_cmd[tr] = cmd;
    p.i_out[tr] = i;
    p.xy_err[tr] = xy;
  }
}

}  // namespace cgp
