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
//
// Threads: shard 0 runs on the CALLING thread, every other shard on a persistent worker thread that was started by
// cgp_sweep_create and sleeps on a condition variable between calls (a sweep over one device is therefore exactly one
// context's call, no hand-off; round 4 created and joined ndev std::threads inside every call, ~60 us of a 0.9 ms
// shard).  cgp_sweep_fit_predict_device is the device-resident form: per-shard device pointers and streams, work enqueued
// without synchronising, so a shard pays neither PCIe nor a host round trip inside the call.
#include "../../include/corenav_gp.h"

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace {
// One persistent worker: runs the job it is handed, signals completion.
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::function<void()> job;
  bool has_job = false, busy = false, quit = false;

  void start() {
    th = std::thread([this] {
      std::unique_lock<std::mutex> lk(mu);
      for (;;) {
        cv_job.wait(lk, [this] { return has_job || quit; });
        if (quit) return;
        std::function<void()> j = std::move(job);
        has_job = false;
        lk.unlock();
        j();
        lk.lock();
        busy = false;
        cv_done.notify_all();
      }
    });
  }
  void post(std::function<void()> j) {
    {
      std::lock_guard<std::mutex> lk(mu);
      job = std::move(j);
      has_job = busy = true;
    }
    cv_job.notify_one();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [this] { return !busy; });
  }
  void stop() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    cv_job.notify_one();
    if (th.joinable()) th.join();
  }
};
}  // namespace

struct cgp_sweep {
  std::vector<cgp_ctx *> ctx;
  std::vector<int> device;
  std::vector<Worker *> worker;   // worker[i] serves shard i + 1 (shard 0 runs on the caller's thread)
  int per_dev_cap = 0;
};

namespace {
// contiguous block partition: shards [0, batch % ndev) get one extra fit
void shard_range(int batch, int i, int ndev, int *start, int *stop) {
  const int base = batch / ndev, extra = batch % ndev;
  *start = i * base + std::min(i, extra);
  *stop = *start + base + (i < extra ? 1 : 0);
}

// fn(i, start, stop) -> status, for every non-empty shard: shards 1.. on their workers, shard 0 here, then wait for all
template <typename F> int for_each_shard(cgp_sweep *sw, int batch, F &&fn) {
  const int ndev = (int)sw->ctx.size();
  std::vector<int> rcs(ndev, CGP_OK);
  for (int i = 1; i < ndev; ++i) {
    int a, b;
    shard_range(batch, i, ndev, &a, &b);
    if (b > a) sw->worker[i - 1]->post([&rcs, &fn, i, a, b] { rcs[i] = fn(i, a, b); });
  }
  {
    int a, b;
    shard_range(batch, 0, ndev, &a, &b);
    if (b > a) rcs[0] = fn(0, a, b);
  }
  for (int i = 1; i < ndev; ++i) sw->worker[i - 1]->wait();
  int first = CGP_OK;
  for (int i = 0; i < ndev; ++i) {
    if (rcs[i] < 0) return rcs[i];  // argument / runtime error of a shard
    if (first == CGP_OK && rcs[i] > 0) first = rcs[i];
  }
  return first;
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
  for (int i = 1; i < ndev; ++i) {
    Worker *w = new Worker();
    w->start();
    sw->worker.push_back(w);
  }
  return sw;
}

void cgp_sweep_destroy(cgp_sweep *sw) {
  if (!sw) return;
  for (Worker *w : sw->worker) {
    w->stop();
    delete w;
  }
  for (cgp_ctx *c : sw->ctx) cgp_destroy(c);
  delete sw;
}

int cgp_sweep_ndev(const cgp_sweep *sw) { return sw ? (int)sw->ctx.size() : 0; }

cgp_ctx *cgp_sweep_context(const cgp_sweep *sw, int i) { return sw && i >= 0 && i < (int)sw->ctx.size() ? sw->ctx[i] : nullptr; }

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
  const int rc = for_each_shard(sw, batch, [=](int i, int a, int b) {
    const size_t o = (size_t)a;
    return cgp_fit_predict_batch(sw->ctx[i], b - a, N, d, M, kid, X + o * N * d, y + o * N, Xs ? Xs + o * M * d : nullptr,
                                 theta + o * theta_stride, theta_stride, include_noise, mean ? mean + o * M : nullptr,
                                 var ? var + o * M : nullptr, logml + o, info + o);
  });
  if (rc < 0) return rc;
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
  return rc;
}

int cgp_sweep_fit_predict_device(cgp_sweep *sw, int batch, int N, int d, int M, int kid, const void *const *dX,
                                 const void *const *dy, const void *const *dXs, const double *const *dtheta,
                                 const double *const *djitter, int include_noise, void *const *dmean, void *const *dvar,
                                 double *const *dlogml, int *const *dinfo, void *const *hip_streams) {
  if (!sw || batch < 1 || !dX || !dy || !dtheta || !dlogml || !dinfo || (M > 0 && (!dXs || !dmean || !dvar))) return CGP_EINVAL;
  const int ndev = (int)sw->ctx.size();
  if ((batch + ndev - 1) / ndev > sw->per_dev_cap) return CGP_ECAPACITY;
  return for_each_shard(sw, batch, [=](int i, int a, int b) {
    return cgp_fit_predict_batch_device(sw->ctx[i], b - a, N, d, M, kid, dX[i], dy[i], dXs ? dXs[i] : nullptr, dtheta[i],
                                        djitter ? djitter[i] : nullptr, include_noise, dmean ? dmean[i] : nullptr,
                                        dvar ? dvar[i] : nullptr, dlogml[i], dinfo[i], hip_streams ? hip_streams[i] : CGP_STREAM_CTX);
  });
}

int cgp_sweep_synchronize(cgp_sweep *sw) {
  if (!sw) return CGP_EINVAL;
  int first = CGP_OK;
  for (cgp_ctx *c : sw->ctx) {
    const int rc = cgp_synchronize(c);
    if (first == CGP_OK && rc != CGP_OK) first = rc;
  }
  return first;
}

}  // extern "C"
