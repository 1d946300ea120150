// cgp_sweep.cpp -- multi-device sweep of independent GP fits behind the C ABI (SURVEY.md 8b
// "cgp_fit_predict_batch(ctx[], ...)", 8e "one host thread + one stream-set per device").
//
// The path shards across fits only: a batch of windows (one per Monte-Carlo trajectory / terrain
// segment) is cut into contiguous per-device blocks -- the same partition as
// corenav_gp_amd/sharding.py::shard_range -- every device runs its block through its own engine context
// on its own host thread, and the only exchange is the per-fit summary table gathered on the host
// (a few KB; there is no data-path collective, so no RCCL call is needed inside one process).  This is
// what lets the C++ ROS host of the reference (gp_predictor) shard an ensemble without Python / torch.
// Pure host code on top of include/corenav_gp.h: no HIP call of its own.
#include "../../include/corenav_gp.h"

#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

struct cgp_sweep {
  std::vector<cgp_ctx *> ctx;
  std::vector<int> device;
  int per_dev_cap = 0;
};

namespace {
// contiguous block partition: shards [0, batch % ndev) get one extra fit
void shard_range(int batch, int i, int ndev, int *start, int *stop) {
  const int base = batch / ndev, extra = batch % ndev;
  *start = i * base + std::min(i, extra);
  *stop = *start + base + (i < extra ? 1 : 0);
}
}  // namespace

extern "C" {

cgp_sweep *cgp_sweep_create(const int *devices, int ndev, int max_n, int max_m, int max_d, int max_batch_total, int dtype) {
  if (!devices || ndev < 1 || ndev > 64 || max_batch_total < 1) return nullptr;
  cgp_sweep *sw = new cgp_sweep();
  sw->per_dev_cap = (max_batch_total + ndev - 1) / ndev;
  for (int i = 0; i < ndev; ++i) {
    cgp_ctx *c = cgp_create(devices[i], max_n, max_m, max_d, sw->per_dev_cap, dtype);
    if (!c) {
      cgp_sweep_destroy(sw);
      return nullptr;
    }
    sw->ctx.push_back(c);
    sw->device.push_back(devices[i]);
  }
  return sw;
}

void cgp_sweep_destroy(cgp_sweep *sw) {
  if (!sw) return;
  for (cgp_ctx *c : sw->ctx) cgp_destroy(c);
  delete sw;
}

int cgp_sweep_ndev(const cgp_sweep *sw) { return sw ? (int)sw->ctx.size() : 0; }

int cgp_sweep_shard(const cgp_sweep *sw, int batch, int i, int *start, int *stop) {
  if (!sw || batch < 0 || i < 0 || i >= (int)sw->ctx.size() || !start || !stop) return CGP_EINVAL;
  shard_range(batch, i, (int)sw->ctx.size(), start, stop);
  return CGP_OK;
}

int cgp_sweep_fit_predict(cgp_sweep *sw, int batch, int N, int d, int M, int kid, const double *X, const double *y,
                          const double *Xs, const double *theta, int theta_stride, int include_noise, double *mean,
                          double *var, double *logml, int *info, double *summary) {
  if (!sw || batch < 1 || !X || !y || !theta || !logml || !info || (M > 0 && (!Xs || !mean || !var))) return CGP_EINVAL;
  const int ndev = (int)sw->ctx.size();
  if ((batch + ndev - 1) / ndev > sw->per_dev_cap) return CGP_ECAPACITY;
  std::vector<int> rcs(ndev, CGP_OK);
  std::vector<std::thread> th;
  th.reserve(ndev);
  for (int i = 0; i < ndev; ++i) {
    int a, b;
    shard_range(batch, i, ndev, &a, &b);
    if (b == a) continue;
    th.emplace_back([=, &rcs]() {
      const size_t o = (size_t)a;
      rcs[i] = cgp_fit_predict_batch(sw->ctx[i], b - a, N, d, M, kid, X + o * N * d, y + o * N,
                                     Xs ? Xs + o * M * d : nullptr, theta + o * theta_stride, theta_stride, include_noise,
                                     mean ? mean + o * M : nullptr, var ? var + o * M : nullptr, logml + o, info + o);
    });
  }
  for (std::thread &t : th) t.join();
  int first = CGP_OK;
  for (int i = 0; i < ndev; ++i) {
    if (rcs[i] < 0) return rcs[i];  // argument / runtime error of a shard
    if (first == CGP_OK && rcs[i] > 0) first = rcs[i];
  }
  // the gather of SURVEY.md 8e: {logml, max sigma = 2 sqrt(max var), info} per fit, global fit order
  if (summary) {
    for (int f = 0; f < batch; ++f) {
      double vmax = 0.0;
      for (int m = 0; m < M; ++m) vmax = std::max(vmax, var[(size_t)f * M + m]);
      summary[3 * (size_t)f + 0] = logml[f];
      summary[3 * (size_t)f + 1] = 2.0 * std::sqrt(vmax);
      summary[3 * (size_t)f + 2] = (double)info[f];
    }
  }
  return first;
}

}  // extern "C"
