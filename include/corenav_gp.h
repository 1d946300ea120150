/* corenav_gp.h -- C ABI of the MI355X-native slip-GP engine (libcorenav_gp.so).
 *
 * The reference has no C ABI on this path: its operator boundary is the ROS message pair
 * core_nav/GP_Input -> core_nav/GP_Output (core_navigation/msg/GP_Input.msg:1-3,
 * core_navigation/msg/GP_Output.msg:1-3) produced by core_navigation/script/gp_slip_node.py and
 * consumed by gp_predictor/src/gp_predictor.cpp.  The entry points below are what a binding for
 * that path has to call; each one names the reference lines it replaces.  INTEGRATION.md shows the
 * ctypes stub for gp_slip_node.py and the C++ call for gp_predictor.
 *
 * Conventions: plain C, no exceptions cross the boundary, caller owns every host pointer, the
 * library owns device memory inside the context.  A context is NOT thread-safe: one context per
 * host thread / HIP stream.  Return value: 0 ok, < 0 argument/runtime error (cgp_strerror), > 0 a
 * LAPACK-style `info` = 1-based index of the first non-positive pivot after the GPy jitter policy
 * (mean(diag)*1e-6*10^k, k = 0..4) is exhausted.
 */
#ifndef CORENAV_GP_H_
#define CORENAV_GP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cgp_ctx cgp_ctx;

/* kernel ids; theta layouts (all fp64, natural -- not log -- parameters):
 *   SE_ISO        [sigma_f^2, ell, sigma_n^2]
 *   SE_ARD        [sigma_f^2, ell_1 .. ell_d, sigma_n^2]
 *   RBF_BROWNIAN  [sigma_rbf^2, ell, sigma_brownian^2, sigma_n^2]   d == 1
 * RBF_BROWNIAN is `GPy.kern.RBF(1) * GPy.kern.Brownian(1)` of gp_slip_node.py:31. */
enum { CGP_KERNEL_SE_ISO = 0, CGP_KERNEL_SE_ARD = 1, CGP_KERNEL_RBF_BROWNIAN = 2 };
enum { CGP_F64 = 0, CGP_F32 = 1 };
enum {
  CGP_OK = 0, CGP_EINVAL = -1, CGP_ENOMEM = -2, CGP_EHIP = -3, CGP_ESTATE = -4, CGP_ENODEVICE = -5,
  CGP_ECAPACITY = -6
};
#define CGP_MAX_D 8
#define CGP_MAX_THETA (CGP_MAX_D + 2)

/* ---- lifetime ------------------------------------------------------------------------------- */
/* Allocates every device buffer for up to `max_batch` simultaneous fits of at most max_n training
 * points, max_m test points, max_d input dimensions.  dtype = CGP_F64 | CGP_F32 is the arithmetic
 * type of the device path; host buffers are always fp64 (the messages are float64[]).
 * Returns NULL on failure (no usable gfx950 device, out of memory): there is no CPU fallback. */
cgp_ctx *cgp_create(int device, int max_n, int max_m, int max_d, int max_batch, int dtype);
void cgp_destroy(cgp_ctx *ctx);
const char *cgp_strerror(int code);
/* Text of the last HIP error seen by this context ("" if none). */
const char *cgp_last_error(const cgp_ctx *ctx);
int cgp_abi_version(void);

/* ---- single window, host buffers -------------------------------------------------------------
 * cgp_fit replaces `GPy.models.GPRegression(x_train, y_train, kernel)` + the exact-inference pass
 * inside it (gp_slip_node.py:35; Gram build, +(sigma_n^2 + 1e-8) I, jitchol, dpotrs, log marginal
 * likelihood) at the fixed hyper-parameters `theta`.  X is (N, d) row-major, y is (N).
 * logml may be NULL. */
int cgp_fit(cgp_ctx *ctx, const double *X, const double *y, int N, int d, int kernel_id,
            const double *theta, double *logml);
/* cgp_predict replaces the `m.predict(np.array([[x]]))` loop (gp_slip_node.py:45-49) for all M
 * test points at once.  Xs is (M, d) row-major.  var is the latent variance clipped at 1e-15 and,
 * when include_noise != 0, plus sigma_n^2 (GPy predict(include_likelihood=True)).  Needs a prior
 * successful cgp_fit on this context. */
int cgp_predict(cgp_ctx *ctx, const double *Xs, int M, int include_noise, double *mean, double *var);
/* alpha = Ky^-1 y of the last fit (GPy `woodbury_vector`, dpotrs).  alpha has N entries. */
int cgp_get_alpha(cgp_ctx *ctx, double *alpha);
/* Lower Cholesky factor of the last fit, (N, N) row-major, upper triangle zero (GPy `LW`). */
int cgp_get_factor(cgp_ctx *ctx, double *L);
/* Jitter that was added to the diagonal for the last fit to succeed (0 if none). */
double cgp_last_jitter(const cgp_ctx *ctx);

/* ---- hyper-parameter optimisation (the reference's `m.optimize()`, gp_slip_node.py:36) ----------
 * cgp_nll_grad: value and gradient of the NEGATIVE log marginal likelihood at theta (natural
 * parameters, same layout as cgp_fit).  GPy's ExactGaussianInference: dL/dK = 0.5 (alpha alpha^T -
 * Ky^-1) contracted with dK/dtheta; Ky^-1 is formed on the device as a syrk of L^-1.  Needs a
 * context created with max_m >= N.  grad has ntheta entries.  Leaves the context fitted at theta. */
int cgp_nll_grad(cgp_ctx *ctx, const double *X, const double *y, int N, int d, int kernel_id,
                 const double *theta, double *nll, double *grad);
/* cgp_optimize: minimises the negative log marginal likelihood over the Logexp-transformed
 * parameters theta = log(1 + exp(x)) (GPy's default positivity constraint) with L-BFGS, starting from
 * theta_inout (GPy starts every parameter at 1.0), at most max_evals objective evaluations (GPy:
 * 1000).  Writes the optimum to theta_inout, its log marginal likelihood to *logml, the number of
 * evaluations to *n_evals, and leaves the context fitted at the optimum (cgp_predict may follow). */
int cgp_optimize(cgp_ctx *ctx, const double *X, const double *y, int N, int d, int kernel_id,
                 double *theta_inout, int max_evals, double *logml, int *n_evals);

/* cgp_optimize_batch: the same optimisation for `batch` windows of identical shape at once.  Every
 * L-BFGS round evaluates value + gradient of ALL windows in one batched device schedule (each window
 * keeps its own line-search / history state on the host); a window whose matrix is not positive
 * definite at a trial point treats it as infeasible (no jitter retry in the batched path).
 * X (batch, N, d), y (batch, N), theta_inout (batch, theta_stride); logml / n_evals (batch) may be
 * NULL.  Needs max_batch >= batch and max_m >= N.  Follow with cgp_fit_predict_batch at the optima. */
int cgp_optimize_batch(cgp_ctx *ctx, int batch, int N, int d, int kernel_id, const double *X, const double *y,
                       double *theta_inout, int theta_stride, int max_evals, double *logml, int *n_evals);

/* Host-only self-test of the L-BFGS used by cgp_optimize: minimises the n-dimensional Rosenbrock
 * function from x0 (n <= 16); writes the minimiser, returns the number of evaluations (< 0 on
 * failure).  Lets the optimiser be tested without a GPU. */
int cgp_selftest_lbfgs(double *x_inout, int n, int max_evals, double *f_out);

/* ---- the node callback in one call -------------------------------------------------------------
 * Everything gp_slip_node.py:16-63 computes between "GP Input Arrived" and pub.publish(), at fixed
 * theta: first int(0.9 n) samples train (:27-29), grid arange(min, max + 600, 1) (:45), output
 * mean = means[n:], sigma = 2 sqrt(var[n:]) (:59-61).  Writes at most `cap` entries; *m_out gets
 * the number of entries the reference would publish.  cgp_slip_node_callback_opt additionally runs
 * cgp_optimize on the training window first (max_evals <= 0: fixed theta), returning theta. */
int cgp_slip_node_callback_opt(cgp_ctx *ctx, const double *time_array, const double *slip_array, int n,
                               int kernel_id, double *theta_inout, int max_evals, double *mean,
                               double *sigma, int cap, int *m_out);
int cgp_slip_node_callback(cgp_ctx *ctx, const double *time_array, const double *slip_array, int n,
                           int kernel_id, const double *theta, double *mean, double *sigma, int cap,
                           int *m_out);

/* ---- batch of independent windows, host buffers ------------------------------------------------
 * `batch` fits of identical shape (one per Monte-Carlo trajectory / terrain segment).
 * X (batch, N, d), y (batch, N), Xs (batch, M, d), theta (batch, theta_stride) row-major; outputs
 * mean/var (batch, M), logml (batch), info (batch; per-fit status as the return-value convention).
 * Returns 0 if every fit succeeded, else the first non-zero per-fit status. */
int cgp_fit_predict_batch(cgp_ctx *ctx, int batch, int N, int d, int M, int kernel_id,
                          const double *X, const double *y, const double *Xs, const double *theta,
                          int theta_stride, int include_noise, double *mean, double *var,
                          double *logml, int *info);

/* ---- batch, device-resident buffers (the measured path) ---------------------------------------
 * All pointers are DEVICE pointers in the context's dtype (fp64 or fp32), SoA per fit:
 *   dX (batch, d, N), dy (batch, N), dXs (batch, d, M), dtheta (batch, CGP_MAX_THETA) fp64,
 *   djitter (batch) fp64 or NULL, dmean/dvar (batch, M), dlogml (batch) fp64, dinfo (batch) int32.
 * Work is enqueued on `hip_stream` (a hipStream_t; NULL = the context's own stream) and the call
 * returns without synchronising.  No jitter retry happens here: read dinfo and re-submit the failed
 * fits with djitter set (cgp_fit_predict_batch does exactly that). */
int cgp_fit_predict_batch_device(cgp_ctx *ctx, int batch, int N, int d, int M, int kernel_id,
                                 const void *dX, const void *dy, const void *dXs, const double *dtheta,
                                 const double *djitter, int include_noise, void *dmean, void *dvar,
                                 double *dlogml, int *dinfo, void *hip_stream);

/* Number of worker streams a batch is spread over (1..8, default 4): the batch is cut into that
 * many groups whose launch schedules run concurrently (HIP streams + events, forked from and
 * joined to the caller's stream), so latency-bound launches of one group overlap MFMA-bound
 * launches of another. */
int cgp_set_streams(cgp_ctx *ctx, int n);

/* Development aid: 64 in-kernel s_memtime stamps (100 MHz ticks) written when env CGP_DBG & 512. */
int cgp_debug_read(cgp_ctx *ctx, long long out[64]);

/* ---- online sliding-window GP (BASELINE configs[3]; not reference behaviour) -------------------
 * `nwin` independent windows of at most N samples each live on the device.  cgp_window_push feeds
 * T ticks to every window in ONE launch: per tick the oldest sample leaves a full window (rank-1
 * Cholesky update), the new one enters (forward substitution), and the tick's outputs are the
 * one-step-ahead predictive mean / variance of the incoming y BEFORE it is added, and the log
 * marginal likelihood of the window after it.  theta (nwin, theta_stride) is fixed per window.
 * xs (nwin, T, d), ys (nwin, T); outputs (nwin, T).  Returns 0, or the 1-based tick at which a window
 * lost positive definiteness. */
int cgp_window_init(cgp_ctx *ctx, int nwin, int N, int d, int kernel_id, const double *theta, int theta_stride);
int cgp_window_push(cgp_ctx *ctx, int T, const double *xs, const double *ys, int include_noise,
                    double *pred_mean, double *pred_var, double *logml);
/* Device-resident variant for streaming benchmarks: dxs/dys/outputs are device pointers, enqueued on
 * hip_stream without synchronising. */
int cgp_window_push_device(cgp_ctx *ctx, int T, const double *dxs, const double *dys, int include_noise,
                           double *dpred_mean, double *dpred_var, double *dlogml, void *hip_stream);
/* Current size of window `w` and the first failing tick (0 = none). */
int cgp_window_state(cgp_ctx *ctx, int w, int *n, int *info);

/* ---- per-kernel timing for the roofline line (bench.py) --------------------------------------
 * When enabled, every launch is bracketed by hipEvents on its stream; cgp_profile_read drains them.
 * kernel index: 0 update(syrk/gemm+gram) 1 potf2(+inverse) 2 trmm 3 finalize(mean/var/logml) 4 alpha.
 * flops = algorithmic flops issued by those launches (DESIGN.md section "Kernels"). */
#define CGP_PROF_KERNELS 5
int cgp_profile_enable(cgp_ctx *ctx, int on);
int cgp_profile_read(cgp_ctx *ctx, double ms[CGP_PROF_KERNELS], double flops[CGP_PROF_KERNELS],
                     long long launches[CGP_PROF_KERNELS]);

/* ---- GpPredictor host arithmetic (gp_predictor/src/gp_predictor.cpp) --------------------------
 * cgp_llh_to_enu: GpPredictor::llh_to_enu (gp_predictor.cpp:144-178). */
int cgp_llh_to_enu(double lat, double lon, double h, const double init_llh[3], const double init_ecef[3],
                   double enu[3]);
/* cgp_predict_stop: the covariance look-ahead of GpPredictor::GPCallBack (gp_predictor.cpp:58-130)
 * on the SetStopping response arrays (core_navigation/srv/SetStopping.srv:3-7).  HvecData has 60
 * entries; h_bug_compatible != 0 unpacks it with the reference's r*4+c indexing
 * (gp_predictor.cpp:38-42), 0 with r*15+c.  Outputs: *fired (threshold crossed), *stop_cmd (the
 * Float64 published on stop_cmd, :107-118), *i_out (odometry steps consumed), *xy_err. */
int cgp_predict_stop(const double *mean, const double *sigma, int M, const double *PvecData,
                     const double *QvecData, const double *STMvecData, const double *HvecData,
                     const double pos_llh[3], double arrival_time, double now, double threshold,
                     int h_bug_compatible, const double init_llh[3], const double init_ecef[3],
                     int *fired, double *stop_cmd, int *i_out, double *xy_err);

/* cgp_predict_stop_batch: the same look-ahead for `ntraj` trajectories at once ON THE DEVICE (one
 * wavefront per trajectory): mean/sigma (ntraj, M), P/Q/STM (ntraj, 225), HvecData
 * (ntraj, 60), pos_llh (ntraj, 3), arrival_time/now (ntraj); outputs (ntraj) each.  Host buffers. */
int cgp_predict_stop_batch(cgp_ctx *ctx, int ntraj, int M, const double *mean, const double *sigma,
                           const double *PvecData, const double *QvecData, const double *STMvecData,
                           const double *HvecData, const double *pos_llh, const double *arrival_time,
                           const double *now, double threshold, int h_bug_compatible, const double init_llh[3],
                           const double init_ecef[3], int *fired, double *stop_cmd, int *i_out, double *xy_err);

/* One GpPredictor::GPCallBack (gp_predictor.cpp:17-132) through the C++ class in
 * csrc/gp_predictor.h with an in-process NodeHandle: the SetStopping service answers with the given
 * arrays, the clock returns `arrival_time` on the first read (:22) and `now` afterwards (:107), and
 * whatever the node publishes on stop_cmd is returned.  Used by the replay harness and the tests. */
int cgp_gppredictor_callback(const double *mean, const double *sigma, int M, const double *PvecData,
                             const double *QvecData, const double *STMvecData, const double *HvecData,
                             const double pos_llh[3], double arrival_time, double now,
                             int h_bug_compatible, int *published, double *stop_cmd);

/* ---- producer side: slip + recording-window state machine of CoreNav::Update -----------------
 * (core_navigation/src/CoreNav.cpp:176,244-330; stopCallback :755-759; getCmdData :794-816).
 * cgp_recorder_update = one 10 Hz odometry update: wheel ground speeds {FL, FR, BL, BR}, INS forward
 * speed, commanded speed.  Returns 1 when a GP_Input window is published this tick; it is then
 * copied to time_out / slipwin_out (at most cap entries, *n_out = its length).  *slip_out = the
 * tick's slip value (may be NULL). */
typedef struct cgp_recorder cgp_recorder;
cgp_recorder *cgp_recorder_create(void);
void cgp_recorder_destroy(cgp_recorder *rec);
int cgp_recorder_update(cgp_recorder *rec, const double wheel_vel[4], double vlin, double cmd_x, double *slip_out,
                        double *time_out, double *slipwin_out, int cap, int *n_out);
void cgp_recorder_stop_cmd(cgp_recorder *rec, double cmd_stop);
void cgp_recorder_cmd(cgp_recorder *rec, double cmd_x);
/* state[8] = {odomUptCount, startRecording, stopRecording, gp_flag, first_driving_flag,
 *             new_stop_data_arrived_, skipped_windows, cmd_stop_} */
void cgp_recorder_state(const cgp_recorder *rec, double state[8]);

#ifdef __cplusplus
}
#endif
#endif /* CORENAV_GP_H_ */
