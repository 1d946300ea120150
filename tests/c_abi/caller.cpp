// The same boundary from C++ (the reference's host language: gp_predictor/src/gp_predictor.cpp is a C++ ROS node): the header
// inside a C++ translation unit, RAII around the context, the node's own work item (GP_Input window -> GP_Output arrays,
// gp_slip_node.py:16-63) through cgp_slip_node_callback, then GpPredictor's stop-time look-ahead on the result
// (gp_predictor.cpp:58-130) through cgp_predict_stop.   caller_cpp <window.bin>
// window.bin: n, then time[n], slip[n], theta[4], then m and the expected mean[m], sigma[m] (doubles).
#include <cmath>
#include <cstdio>
#include <memory>
#include <vector>

#include "corenav_gp.h"

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  auto rd = [&](double *p, size_t n) { return std::fread(p, sizeof(double), n, f) == n; };
  double nn = 0, mm = 0;
  if (!rd(&nn, 1)) return 2;
  const int n = (int)nn;
  std::vector<double> t(n), s(n), theta(4);
  if (!rd(t.data(), n) || !rd(s.data(), n) || !rd(theta.data(), 4) || !rd(&mm, 1)) return 2;
  const int m = (int)mm;
  std::vector<double> emean(m), esigma(m);
  if (!rd(emean.data(), m) || !rd(esigma.data(), m)) return 2;
  std::fclose(f);
  int status = 0;
  std::unique_ptr<cgp_ctx, void (*)(cgp_ctx *)> ctx(cgp_create_ex(0, 256, 1024, 1, 1, CGP_F64, &status), cgp_destroy);
  if (!ctx) {
    std::fprintf(stderr, "caller.cpp: cgp_create_ex: %s\n", cgp_strerror(status));
    return 3;
  }
  std::vector<double> mean(1024), sigma(1024);
  int m_out = 0;
  int rc = cgp_slip_node_callback(ctx.get(), t.data(), s.data(), n, CGP_KERNEL_RBF_BROWNIAN, theta.data(), mean.data(), sigma.data(), 1024, &m_out);
  if (rc != 0 || m_out != m) {
    std::fprintf(stderr, "caller.cpp: callback rc %d, %d entries (expected %d)\n", rc, m_out, m);
    return 1;
  }
  double em = 0, es = 0, sc = 0;
  for (int i = 0; i < m; ++i) {
    em = std::fmax(em, std::fabs(mean[i] - emean[i]));
    sc = std::fmax(sc, std::fabs(emean[i]));
    es = std::fmax(es, std::fabs(sigma[i] - esigma[i]) / esigma[i]);
  }
  std::printf("node callback from C++: %d predictions, mean err %.2e, sigma err %.2e\n", m, em / sc, es);
  if (!(em / sc < 1e-6 && es < 1e-6)) return 1;
  // GpPredictor's look-ahead on these arrays with a filter snapshot of its own (values as tests/test_host_abi.py's smallest case)
  std::vector<double> P(225, 0.0), Q(225, 0.0), STM(225, 0.0), H(60, 0.0);
  for (int i = 0; i < 15; ++i) {
    P[i * 15 + i] = i < 3 ? 1e-6 : (i < 6 ? 2e-3 : 1e-8);
    Q[i * 15 + i] = i < 3 ? 1e-9 : (i < 6 ? 3e-5 : 1e-12);
    STM[i * 15 + i] = 1.0;
  }
  STM[6 * 15 + 3] = 1.6e-9, STM[7 * 15 + 4] = 2.0e-9;
  const double llh[3] = {0.693457963620326, -1.39498384275845, 334.99}, init_llh[3] = {0.693457963620326, -1.39498384275845, 334.99};
  const double init_ecef[3] = {856503.0, -4843015.0, 4047267.0};
  int fired = 0, iout = 0;
  double cmd = 0, xy = 0;
  rc = cgp_predict_stop(mean.data(), sigma.data(), m, P.data(), Q.data(), STM.data(), H.data(), llh, 10.0, 10.5, 3.0, 1, init_llh, init_ecef, &fired, &cmd,
                        &iout, &xy);
  if (rc != 0 || !(xy >= 0.0) || iout < 0) return 1;
  std::printf("look-ahead from C++: fired %d after %d odometry steps, xy_err %.3f, stop_cmd %.2f\ncaller.cpp ok\n", fired, iout, xy, cmd);
  return 0;
}
