// Drives cgp_sweep_* over the engine double: many calls, several "devices", both entry points; exits non-zero on a wrong result.
#include "../../include/corenav_gp.h"

#include <cstdio>
#include <vector>
extern "C" long cgp_double_calls(const cgp_ctx *c);

int main() {
  const int ndev = 5, N = 3, d = 2, M = 4, theta_stride = 4;
  int devices[ndev] = {0, 3, 7, 3, 9};   // a device named twice gets two contexts
  for (int round = 0; round < 3; ++round) {   // create / destroy: worker threads start and stop cleanly
    cgp_sweep *sw = cgp_sweep_create(devices, ndev, N, M, d, 64, CGP_F64);
    if (!sw || cgp_sweep_ndev(sw) != ndev) return 1;
    for (int call = 0; call < 400; ++call) {
      const int batch = 1 + (call * 7) % 64;   // also batches smaller than the device count: empty shards
      std::vector<double> X((size_t)batch * N * d), y((size_t)batch * N), Xs((size_t)batch * M * d), th((size_t)batch * theta_stride, 1.0);
      for (int f = 0; f < batch; ++f) X[(size_t)f * N * d] = f;
      std::vector<double> mean((size_t)batch * M), var((size_t)batch * M), logml(batch), summ((size_t)batch * 3);
      std::vector<int> info(batch, -1);
      const int rc = cgp_sweep_fit_predict(sw, batch, N, d, M, 0, X.data(), y.data(), Xs.data(), th.data(), theta_stride, 1, mean.data(),
                                           var.data(), logml.data(), info.data(), summ.data());
      bool expect17 = false;
      for (int i = 0; i < ndev; ++i) {
        int a, b;
        if (cgp_sweep_shard(sw, batch, i, &a, &b) != CGP_OK) return 2;
        expect17 = expect17 || b - a == 13;
        for (int f = a; f < b; ++f)
          if (logml[f] != f + 1000.0 * devices[i] || info[f] != 0 || summ[3 * (size_t)f] != logml[f]) {   // the shard's own slice of X (X[f] = f), on the shard's own context
            std::fprintf(stderr, "call %d fit %d: logml %g\n", call, f, logml[f]);
            return 3;
          }
      }
      if (rc != (expect17 ? 17 : CGP_OK)) return 4;
      // device entry: per-shard pointers
      std::vector<std::vector<double>> dX(ndev), dl(ndev);
      std::vector<std::vector<int>> di(ndev);
      std::vector<const void *> pX(ndev), py(ndev);
      std::vector<const double *> pth(ndev);
      std::vector<double *> pl(ndev);
      std::vector<int *> pi(ndev);
      std::vector<void *> pm(ndev), pv(ndev);
      for (int i = 0; i < ndev; ++i) {
        int a, b;
        cgp_sweep_shard(sw, batch, i, &a, &b);
        dX[i].assign(b - a + 1, 0.0);
        for (int f = 0; f < b - a; ++f) dX[i][f] = 10.0 * f;
        dl[i].assign(b - a + 1, -1.0);
        di[i].assign(b - a + 1, -1);
        pX[i] = py[i] = dX[i].data();
        pth[i] = dX[i].data();
        pl[i] = dl[i].data();
        pi[i] = di[i].data();
        pm[i] = pv[i] = dl[i].data();
      }
      if (cgp_sweep_fit_predict_device(sw, batch, N, d, M, 0, pX.data(), py.data(), pX.data(), pth.data(), nullptr, 1, pm.data(), pv.data(),
                                       pl.data(), pi.data(), nullptr) != CGP_OK)
        return 5;
      if (cgp_sweep_synchronize(sw) != CGP_OK) return 6;
      for (int i = 0; i < ndev; ++i) {
        int a, b;
        cgp_sweep_shard(sw, batch, i, &a, &b);
        for (int f = 0; f < b - a; ++f)
          if (dl[i][f] != 10.0 * f + 1000.0 * devices[i] || di[i][f] != 0) return 7;
      }
    }
    long total = 0;
    for (int i = 0; i < ndev; ++i) total += cgp_double_calls(cgp_sweep_context(sw, i));
    if (total <= 0) return 8;
    cgp_sweep_destroy(sw);
  }
  std::puts("sweep threads ok");
  return 0;
}
