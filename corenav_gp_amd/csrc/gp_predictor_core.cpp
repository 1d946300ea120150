// gp_predictor_core.cpp -- see header.  Every block cites the reference line it reproduces.
#include "gp_predictor_core.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace corenav {

namespace {
constexpr int NS = 15;  // error-state dimension
constexpr int NM = 4;   // measurement dimension

// C(n x m) = A(n x k) * B(k x m)
inline void mm(const double *A, const double *B, double *C, int n, int k, int m) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      double s = 0.0;
      for (int q = 0; q < k; ++q) s += A[i * k + q] * B[q * m + j];
      C[i * m + j] = s;
    }
}
// C(n x m) = A(n x k) * B(m x k)^T
inline void mmT(const double *A, const double *B, double *C, int n, int k, int m) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      double s = 0.0;
      for (int q = 0; q < k; ++q) s += A[i * k + q] * B[j * k + q];
      C[i * m + j] = s;
    }
}
// general 4x4 inverse by Gauss-Jordan with partial pivoting (Eigen's .inverse() on a 4x4)
inline void inv4(const double *A, double *Ai) {
  double a[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      a[i][j] = A[i * 4 + j];
      a[i][4 + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    for (int r = c + 1; r < 4; ++r)
      if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
    if (piv != c)
      for (int j = 0; j < 8; ++j) std::swap(a[c][j], a[piv][j]);
    const double d = 1.0 / a[c][c];
    for (int j = 0; j < 8; ++j) a[c][j] *= d;
    for (int r = 0; r < 4; ++r) {
      if (r == c) continue;
      const double f = a[r][c];
      if (f != 0.0)
        for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
    }
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) Ai[i * 4 + j] = a[i][4 + j];
}
}  // namespace

void llh_to_enu(double lat, double lon, double h, const double init_llh[3], const double init_ecef[3],
                double enu[3]) {
  const double a = 6378137.0000, b = 6356752.3142;            // :150-151
  const double e = std::sqrt(1.0 - std::pow(b / a, 2));        // :152
  const double sinphi = std::sin(lat), cosphi = std::cos(lat);
  const double coslam = std::cos(lon), sinlam = std::sin(lon);
  const double tan2phi = std::pow(std::tan(lat), 2);
  const double tmp2 = 1.0 - e * e;
  const double tmpden = std::sqrt(1.0 + tmp2 * tan2phi);
  const double x1 = (a * coslam) / tmpden + h * coslam * cosphi;  // :161
  const double y1 = (a * sinlam) / tmpden + h * sinlam * cosphi;
  const double tmp3 = std::sqrt(1.0 - e * e * sinphi * sinphi);
  const double z1 = (a * tmp2 * sinphi) / tmp3 + h * sinphi;       // :164
  const double dx = x1 - init_ecef[0], dy = y1 - init_ecef[1], dz = z1 - init_ecef[2];
  const double sP = std::sin(init_llh[0]), cP = std::cos(init_llh[0]);
  const double sL = std::sin(init_llh[1]), cL = std::cos(init_llh[1]);
  enu[0] = -sL * dx + cL * dy;                                     // :173 R_dist row 0
  enu[1] = -sP * cL * dx - sP * sL * dy + cP * dz;
  enu[2] = cP * cL * dx + cP * sL * dy + sP * dz;
}

void unpack_H(const double *HvecData, bool bug_compatible, double H[60]) {
  for (int r = 0; r < NM; ++r)
    for (int c = 0; c < NS; ++c) H[r * NS + c] = HvecData[bug_compatible ? r * 4 + c : r * NS + c];  // :38-42
}

StopPrediction predict_stop(const double *mean, const double *sigma, int M, const double *PvecData,
                            const double *QvecData, const double *STMvecData, const double *HvecData,
                            const double pos_llh[3], double arrival_time, double now, double threshold,
                            bool h_bug_compatible, const double init_llh[3], const double init_ecef[3]) {
  StopPrediction out;
  double P[NS * NS], Q[NS * NS], F[NS * NS], H[NM * NS], T[NS * NS], T2[NS * NS];
  std::memcpy(P, PvecData, sizeof(P));    // :30-36 row-major unpack
  std::memcpy(Q, QvecData, sizeof(Q));
  std::memcpy(F, STMvecData, sizeof(F));
  unpack_H(HvecData, h_bug_compatible, H);
  const double R1[16] = {0.5, 0.5, 0.0, 0.0, 1 / 0.685, -1 / 0.685, 0.0, 0.0,
                         0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0};  // :84-87
  int i = 0;
  double e0[3];
  llh_to_enu(pos_llh[0], pos_llh[1], pos_llh[2], init_llh, init_ecef, e0);  // :95 (loop-invariant)
  for (int slip_i = 0; slip_i < 5 * M; ++slip_i) {                          // :64
    mm(F, P, T, NS, NS, NS);                                                // :66  P = F P F' + Q
    mmT(T, F, T2, NS, NS, NS);
    for (int q = 0; q < NS * NS; ++q) P[q] = T2[q] + Q[q];
    if (slip_i % 5 == 0) {                                                  // :67
      const double c0 = mean[i], c1 = mean[i] + sigma[i], c2 = mean[i] - sigma[i];   // :69-71
      const double o0 = 0.8 / (1.0 - c0), o1 = 0.8 / (1.0 - c1), o2 = 0.8 / (1.0 - c2);  // :73-75
      const double est = (o0 + o1 + o2) / 3.0;                                       // :77
      const double cov = ((o0 - est) * (o0 - est) + (o1 - est) * (o1 - est) + (o2 - est) * (o2 - est)) / 3.0;  // :78
      double R2[16] = {0};
      R2[0] = std::max(0.03 * 0.03, cov * cov);                                       // :80-83
      R2[5] = std::max(0.03 * 0.03, cov * cov);
      R2[10] = std::max(0.05 * 0.05, cov * cov);
      R2[15] = 0.05 * 0.05;
      double A4[16], R[16];
      mm(R1, R2, A4, 4, 4, 4);                                                        // :88
      mmT(A4, R1, R, 4, 4, 4);
      for (double &v : R) v *= 25.0;
      double PHt[NS * NM], S[16], Si[16], K[NS * NM];
      mmT(P, H, PHt, NS, NS, NM);                                                     // :90
      mm(H, PHt, S, NM, NS, NM);
      for (int q = 0; q < 16; ++q) S[q] += R[q];
      inv4(S, Si);
      mm(PHt, Si, K, NS, NM, NM);
      double IKH[NS * NS], KR[NS * NM], KRK[NS * NS];
      mm(K, H, IKH, NS, NM, NS);                                                      // :91 Joseph form
      for (int r = 0; r < NS; ++r)
        for (int cidx = 0; cidx < NS; ++cidx) IKH[r * NS + cidx] = (r == cidx ? 1.0 : 0.0) - IKH[r * NS + cidx];
      mm(IKH, P, T, NS, NS, NS);
      mmT(T, IKH, T2, NS, NS, NS);
      mm(K, R, KR, NS, NM, NM);
      mmT(KR, K, KRK, NS, NM, NS);
      for (int q = 0; q < NS * NS; ++q) P[q] = T2[q] + KRK[q];
      ++i;                                                                            // :92
    }
    const double s6 = 3.0 * std::sqrt(std::fabs(P[6 * NS + 6]));
    const double s7 = 3.0 * std::sqrt(std::fabs(P[7 * NS + 7]));
    const double s8 = 3.0 * std::sqrt(std::fabs(P[8 * NS + 8]));
    double e3[3];
    llh_to_enu(pos_llh[0] + s6, pos_llh[1] + s7, pos_llh[2] + s8, init_llh, init_ecef, e3);  // :97
    out.xy_err = std::sqrt((e3[0] - e0[0]) * (e3[0] - e0[0]) + (e3[1] - e0[1]) * (e3[1] - e0[1]));  // :99
    if (out.xy_err > threshold) {                                                     // :102
      const double dt = arrival_time + i / 10.0 - now;                                // :107,:114
      out.stop_cmd = (dt < 0.0) ? 0.5 : dt;
      out.fired = true;
      break;                                                                          // :121
    }
  }
  out.i = i;
  return out;
}

}  // namespace corenav
