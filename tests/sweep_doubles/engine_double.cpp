// Test double of the engine entry points cgp_sweep.cpp calls (tests/test_sweep_threads_cpu.py): no GPU, no HIP.  A "context" is a
// counter; a "fit" writes a value that depends on the context, the call number and the fit index, so the driver can check that
// every shard ran exactly once per call, on its own context, with its own slice -- under ThreadSanitizer.
#include "../../include/corenav_gp.h"

#include <atomic>
#include <chrono>
#include <thread>

struct cgp_ctx {
  int device;
  long calls;            // touched only by the thread that runs this context's shard: a race here is a sweep bug
  std::atomic<long> pending;
};

extern "C" {
cgp_ctx *cgp_create(int device, int, int, int, int, int) { return device >= 100 ? nullptr : new cgp_ctx{device, 0, {0}}; }
cgp_ctx *cgp_create_ex(int device, int a, int b, int c, int d, int e, int *st) {
  cgp_ctx *x = cgp_create(device, a, b, c, d, e);
  if (st) *st = x ? CGP_OK : CGP_ENODEVICE;
  return x;
}
void cgp_destroy(cgp_ctx *c) { delete c; }
int cgp_synchronize(cgp_ctx *c) {
  c->pending.store(0);
  return CGP_OK;
}
int cgp_fit_predict_batch(cgp_ctx *c, int batch, int N, int d, int M, int, const double *X, const double *, const double *, const double *,
                          int, int, double *mean, double *var, double *logml, int *info) {
  ++c->calls;
  if (c->device == 7) std::this_thread::sleep_for(std::chrono::microseconds(50));   // a slow device: the caller must still wait for it
  for (int f = 0; f < batch; ++f) {
    logml[f] = X[(size_t)f * N * d] + 1000.0 * c->device;
    info[f] = 0;
    for (int m = 0; m < M; ++m) mean[(size_t)f * M + m] = (double)c->calls, var[(size_t)f * M + m] = 1.0 + m;
  }
  return batch == 13 ? 17 : CGP_OK;   // a per-fit status (not positive definite) must come back as the call's status
}
int cgp_fit_predict_batch_device(cgp_ctx *c, int batch, int, int, int, int, const void *dX, const void *, const void *, const double *,
                                 const double *, int, void *, void *, double *dlogml, int *dinfo, void *) {
  ++c->calls;
  c->pending.fetch_add(1);
  for (int f = 0; f < batch; ++f) {
    dlogml[f] = static_cast<const double *>(dX)[f] + 1000.0 * c->device;
    dinfo[f] = 0;
  }
  return CGP_OK;
}
long cgp_double_calls(const cgp_ctx *c) { return c->calls; }
}
