// cgp_lookahead.hpp -- batched stop-time look-ahead on the GPU: the loop of
// GpPredictor::GPCallBack (gp_predictor/src/gp_predictor.cpp:58-130) for a whole Monte-Carlo
// ensemble at once, one 64-lane wave per trajectory, the 15x15 filter matrices in LDS.
// The host class (csrc/gp_predictor_core.cpp) stays the single-trajectory path and the checker's
// counterpart; both follow the same statement order.
//
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
  const double *arrival, *now; // [ntraj]
  double *stop_cmd, *xy_err;   // [ntraj]
  int *fired, *i_out;          // [ntraj]
  int ntraj, M, h_bug_compatible;
  double threshold;
  double init_llh[3], init_ecef[3];
};

constexpr int LA_NS = 15, LA_NM = 4, LA_WAVES = 4;
constexpr int LA_PER_WAVE = 4 * 225 + 4 * 60 + 64;  // P, F, Q, T | H, PHt, K, KR | small scratch

__device__ __forceinline__ void la_llh_to_enu(double lat, double lon, double h, const double *illh, const double *iecef,
                                              double &e0, double &e1) {
  const double a = 6378137.0000, b = 6356752.3142;
  const double e = sqrt(1.0 - (b / a) * (b / a));
  const double sinphi = sin(lat), cosphi = cos(lat), coslam = cos(lon), sinlam = sin(lon);
  const double tanphi = tan(lat);
  const double tmp2 = 1.0 - e * e;
  const double tmpden = sqrt(1.0 + tmp2 * tanphi * tanphi);
  const double x1 = (a * coslam) / tmpden + h * coslam * cosphi;
  const double y1 = (a * sinlam) / tmpden + h * sinlam * cosphi;
  const double tmp3 = sqrt(1.0 - e * e * sinphi * sinphi);
  const double z1 = (a * tmp2 * sinphi) / tmp3 + h * sinphi;
  const double dx = x1 - iecef[0], dy = y1 - iecef[1], dz = z1 - iecef[2];
  const double sP = sin(illh[0]), cP = cos(illh[0]), sL = sin(illh[1]), cL = cos(illh[1]);
  e0 = -sL * dx + cL * dy;
  e1 = -sP * cL * dx - sP * sL * dy + cP * dz;
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
  const double lat = p.pos[tr * 3], lon = p.pos[tr * 3 + 1], hgt = p.pos[tr * 3 + 2];
  double e00, e01;
  la_llh_to_enu(lat, lon, hgt, p.init_llh, p.init_ecef, e00, e01);  // :95, loop invariant
  const double R1[16] = {0.5, 0.5, 0.0, 0.0, 1 / 0.685, -1 / 0.685, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0};
  int i = 0, fired = 0;
  double xy = 0.0, cmd = 0.0;
  for (int slip_i = 0; slip_i < 5 * p.M; ++slip_i) {   // :64
    la_mm<false>(F, P, T, LA_NS, LA_NS, LA_NS, lane);   // :66  P = F P F' + Q
    la_mm<true>(T, F, P, LA_NS, LA_NS, LA_NS, lane, Q);
    if (slip_i % 5 == 0) {                              // :67
      const double c0 = mean[i], c1 = mean[i] + sigma[i], c2 = mean[i] - sigma[i];
      const double o0 = 0.8 / (1.0 - c0), o1 = 0.8 / (1.0 - c1), o2 = 0.8 / (1.0 - c2);
      const double est = (o0 + o1 + o2) / 3.0;
      const double cov = ((o0 - est) * (o0 - est) + (o1 - est) * (o1 - est) + (o2 - est) * (o2 - est)) / 3.0;
      double R2[4] = {fmax(0.03 * 0.03, cov * cov), fmax(0.03 * 0.03, cov * cov), fmax(0.05 * 0.05, cov * cov), 0.05 * 0.05};
      double R[16];  // R = 25 R1 R2 R1'  (every lane computes it: 4x4)
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
          double s = 0.0;
          for (int q = 0; q < 4; ++q) s += R1[r * 4 + q] * R2[q] * R1[c * 4 + q];
          R[r * 4 + c] = 25.0 * s;
        }
      la_mm<true>(P, H, PHt, LA_NS, LA_NS, LA_NM, lane);        // P H'   (15 x 4)
      la_mm<false>(H, PHt, sc, LA_NM, LA_NS, LA_NM, lane);      // H P H' (4 x 4) -> sc[0..15]
      // S^-1 by Gauss-Jordan with partial pivoting, redundantly in every lane's registers
      double a[4][8];
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
          a[r][c] = sc[r * 4 + c] + R[r * 4 + c];
          a[r][4 + c] = (r == c) ? 1.0 : 0.0;
        }
      for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r)
          if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (piv != c)
          for (int j = 0; j < 8; ++j) {
            const double t = a[c][j];
            a[c][j] = a[piv][j];
            a[piv][j] = t;
          }
        const double d = 1.0 / a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] *= d;
        for (int r = 0; r < 4; ++r) {
          if (r == c) continue;
          const double f = a[r][c];
          if (f != 0.0)
            for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < 16) sc[16 + lane] = a[lane >> 2][4 + (lane & 3)];  // Si
      if (lane < 16) sc[32 + lane] = R[lane];
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      la_mm<false>(PHt, sc + 16, K, LA_NS, LA_NM, LA_NM, lane);   // K = P H' S^-1  (15 x 4)   :90
      la_mm<false>(K, H, T, LA_NS, LA_NM, LA_NS, lane);           // K H
      for (int o = lane; o < 225; o += 64) T[o] = ((o / LA_NS) == (o % LA_NS) ? 1.0 : 0.0) - T[o];  // I - K H
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      la_mm<false>(K, sc + 32, KR, LA_NS, LA_NM, LA_NM, lane);    // K R
      // Joseph form (:91):  P = (I-KH) P (I-KH)' + K R K'.  F is needed again next step, so the
      // intermediate goes through Q's neighbour: use PHt-free space?  15x15 temp = reuse `sc`-less: two passes via registers
      double tmp[4];
      for (int u = 0; u < 4; ++u) {
        const int o = lane + 64 * u;
        double s = 0.0;
        if (o < 225) {
          const int r = o / LA_NS, c = o - r * LA_NS;
          for (int q = 0; q < LA_NS; ++q) s += T[r * LA_NS + q] * P[q * LA_NS + c];
        }
        tmp[u] = s;
      }
      __builtin_amdgcn_wave_barrier();
      for (int u = 0; u < 4; ++u) {
        const int o = lane + 64 * u;
        if (o < 225) P[o] = tmp[u];                                 // P <- (I-KH) P
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      for (int u = 0; u < 4; ++u) {
        const int o = lane + 64 * u;
        double s = 0.0;
        if (o < 225) {
          const int r = o / LA_NS, c = o - r * LA_NS;
          for (int q = 0; q < LA_NS; ++q) s += P[r * LA_NS + q] * T[c * LA_NS + q];
          for (int q = 0; q < LA_NM; ++q) s += KR[r * LA_NM + q] * K[c * LA_NM + q];
        }
        tmp[u] = s;
      }
      __builtin_amdgcn_wave_barrier();
      for (int u = 0; u < 4; ++u) {
        const int o = lane + 64 * u;
        if (o < 225) P[o] = tmp[u];
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      ++i;                                                // :92
    }
    const double s6 = 3.0 * sqrt(fabs(P[6 * LA_NS + 6])), s7 = 3.0 * sqrt(fabs(P[7 * LA_NS + 7])),
                 s8 = 3.0 * sqrt(fabs(P[8 * LA_NS + 8]));
    double e30, e31;
    la_llh_to_enu(lat + s6, lon + s7, hgt + s8, p.init_llh, p.init_ecef, e30, e31);  // :97
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
