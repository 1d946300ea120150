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
/* `hip_stream` arguments take a hipStream_t.  NULL is the legacy default stream itself (the value
 * torch.cuda.current_stream().cuda_stream has for the default stream): the enqueued work is ordered
 * after the caller's earlier default-stream work (e.g. the kernels that produced dX) and before its
 * later work.  CGP_STREAM_CTX selects the context's private non-blocking stream; the caller then
 * orders it against its own streams with events or cgp_synchronize. */
#define CGP_STREAM_CTX ((void *)(size_t)-1)

/* ---- lifetime ------------------------------------------------------------------------------- */
/* Allocates every device buffer for up to `max_batch` simultaneous fits of at most max_n training
 * points, max_m test points, max_d input dimensions.  dtype = CGP_F64 | CGP_F32 is the arithmetic
 * type of the device path; host buffers are always fp64 (the messages are float64[]).
 * CGP_F64 is the reference's arithmetic and meets 1e-6 against it on every kernel.  CGP_F32 is for the SE kernels on
 * standardised inputs (BASELINE configs[2], 1e-3).  Its contract, checked by tests/fuzz/fuzz_parity.py (one bar, no second class):
 *   - predictive mean: refined against a double-precision residual (cgp_set_refine below; by default every window of d <= 3
 *     input dimensions and, beyond, every fit whose factor shows a dense window) -- 5e-5 of the oracle or better where it is
 *     refined (typically 1e-6), 1e-3 where it is not;
 *   - variance and logML come from the single-precision factor: max(1e-3, 10 x the error of spotrf / strtrs on the same
 *     window) -- the second term only matters for windows that are ill-conditioned in single precision (dense
 *     one-dimensional inputs), where no single-precision factorisation holds 1e-3;
 *   - the reference's RBF x Brownian kernel on raw tick counts (cond(Ky) ~ 1e6, prior variance 1000 x the posterior one) is an
 *     fp64 path, as in the reference: in CGP_F32 its mean is refined like any d = 1 window, its variance is held to
 *     max(3e-3, 30 x that LAPACK error) only (its banded factor meets the bf16 matrix cores' truncating sums: a bias of
 *     +1.6e-3 +- 5e-4 at a thousand samples, one sweep window at 3.29e-3).  Use CGP_F64 for that kernel.
 * Returns NULL on failure (device index out of range, device is not gfx950 -- the architecture name
 * is checked: the code object holds gfx950 kernels only -- or out of memory): no CPU fallback. */
cgp_ctx *cgp_create(int device, int max_n, int max_m, int max_d, int max_batch, int dtype);
/* The same, telling WHY it failed: *status (may be NULL) = CGP_OK, CGP_EINVAL (an argument out of range), CGP_ENODEVICE
 * (device index out of range or not a gfx950 part), CGP_ENOMEM (a device allocation failed: the context is sized by
 * max_batch x max_n^2) or CGP_EHIP (any other runtime failure). */
cgp_ctx *cgp_create_ex(int device, int max_n, int max_m, int max_d, int max_batch, int dtype, int *status);
void cgp_destroy(cgp_ctx *ctx);
const char *cgp_strerror(int code);
/* Text of the last HIP error seen by this context ("" if none). */
const char *cgp_last_error(const cgp_ctx *ctx);
/* ABI revision of the library that is loaded; compare with CGP_ABI_VERSION of the header a client was built
 * against.  2: cgp_debug_read writes CGP_DEBUG_SLOTS = 512 slots (version 1: 64), and a NULL `hip_stream` is the legacy
 * default stream (version 1: the context's private stream, now CGP_STREAM_CTX).  3 (this header): cgp_set_streams accepts 0
 * (the engine decides; the default, was one group) and cgp_create_ex, cgp_lbfgs_minimize, cgp_sweep_fit_predict_device,
 * cgp_sweep_synchronize, cgp_sweep_context, cgp_set_refine exist; fp32 windows of d <= 3 get a refined mean by default. */
#define CGP_ABI_VERSION 3
int cgp_abi_version(void);
/* How the library was built: 0 for the shipped library.  CGP_BUILD_ABLATION (-DCGP_ABLATION): env
 * CGP_DBG is read and can skip parts of the arithmetic for timing ablations -- outputs are WRONG by
 * design, bench.py refuses to report such a build as a measurement.  CGP_BUILD_AB (-DCGP_AB): the
 * alternative schedules of DESIGN.md section 13 are compiled in and selectable by environment.
 * CGP_BUILD_F32_NATIVE (-DCGP_F32_BF16X6=0): the fp32 tile loops use the fp32-input MFMA instead of the shipped form (every
 * fp32 product as six bf16 products on the bf16 matrix cores, same fp32 rounding level: DESIGN.md section 4). */
enum { CGP_BUILD_ABLATION = 1, CGP_BUILD_AB = 2, CGP_BUILD_F32_NATIVE = 4 };
int cgp_build_flags(void);
/* Blocks until everything enqueued on the context's private stream has finished. */
int cgp_synchronize(cgp_ctx *ctx);

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
 * evaluations to *n_evals, and leaves the context fitted at the optimum (cgp_predict may follow).
 * *n_evals: windows of at most 160 samples (one-launch device optimiser) report the optimiser's own evaluations (what
 * scipy reports as nfev for the same run); longer windows (host optimiser over device gradients) report those + 1,
 * the refit at the optimum that leaves the factor panel resident.  The optimiser is scipy's L-BFGS-B without bounds
 * (csrc/lbfgs_core.hpp): same line search, same stopping tests, same trajectory to rounding. */
int cgp_optimize(cgp_ctx *ctx, const double *X, const double *y, int N, int d, int kernel_id,
                 double *theta_inout, int max_evals, double *logml, int *n_evals);

/* cgp_optimize_batch: the same optimisation for `batch` windows of identical shape at once.  Every
 * L-BFGS round evaluates value + gradient of ALL windows in one batched device schedule (each window
 * keeps its own line-search / history state on the host); a window whose matrix is not positive
 * definite at a trial point is re-evaluated with GPy's jitter ladder (mean(diag) 1e-6 10^k, k = 0..4,
 * the windows that failed only), exactly as jitchol does inside m.optimize(); a point that still
 * fails is infeasible (+inf) for the line search.
 * X (batch, N, d), y (batch, N), theta_inout (batch, theta_stride); logml / n_evals (batch) may be
 * NULL.  Needs max_batch >= batch and max_m >= N.  Follow with cgp_fit_predict_batch at the optima. */
int cgp_optimize_batch(cgp_ctx *ctx, int batch, int N, int d, int kernel_id, const double *X, const double *y,
                       double *theta_inout, int theta_stride, int max_evals, double *logml, int *n_evals);

/* Host-only self-test of the L-BFGS used by cgp_optimize: minimises the n-dimensional Rosenbrock
 * function from x0 (n <= 16); writes the minimiser, returns the number of evaluations (< 0 on
 * failure).  Lets the optimiser be tested without a GPU. */
int cgp_selftest_lbfgs(double *x_inout, int n, int max_evals, double *f_out);

/* Host-only: the optimiser state machine of cgp_optimize / cgp_optimize_batch (and, lane-parallel, of the one-launch
 * short-window kernel) driven on a caller-supplied objective -- the role scipy.optimize.fmin_l_bfgs_b plays under
 * m.optimize() (gp_slip_node.py:36).  fn(x, grad, n, user) returns f and writes the gradient; a non-finite f marks an
 * infeasible point.  x_inout (n <= 16) holds the start and receives the best point.  pgtol / factr as in scipy (GPy:
 * 1e-5, 1e7).  Returns 0, or CGP_EINVAL; *status: 0 gradient test, 1 function-decrease test, 2 max_evals, 3 line search
 * failed.  Lets tests compare the optimiser with scipy on the oracle's objective without a GPU. */
typedef double (*cgp_objective_fn)(const double *x, double *grad, int n, void *user);
int cgp_lbfgs_minimize(cgp_objective_fn fn, void *user, double *x_inout, int n, int max_evals, double pgtol, double factr,
                       double *f_out, int *n_evals, int *n_iters, int *status);

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
 * Returns 0 if every fit succeeded, else the first non-zero per-fit status.
 * Short fp64 windows (N <= 144 for any d, N <= 160 at d <= 2; M > 0) run as ONE launch with the factor in LDS
 * (csrc/cgp_small.hpp: k_small_predict), the same form the node callbacks take; longer ones on the tiled schedules.  Which
 * form runs depends on (N, d, M) only, never on `batch` or the slot; cgp_fit + cgp_predict of the same window (always the
 * tiled schedules: the factor stays resident for cgp_get_factor) agree with it to rounding, not bitwise. */
int cgp_fit_predict_batch(cgp_ctx *ctx, int batch, int N, int d, int M, int kernel_id,
                          const double *X, const double *y, const double *Xs, const double *theta,
                          int theta_stride, int include_noise, double *mean, double *var,
                          double *logml, int *info);

/* ---- batch, device-resident buffers (the measured path) ---------------------------------------
 * All pointers are DEVICE pointers in the context's dtype (fp64 or fp32), SoA per fit:
 *   dX (batch, d, N), dy (batch, N), dXs (batch, d, M), dtheta (batch, CGP_MAX_THETA) fp64,
 *   djitter (batch) fp64 or NULL, dmean/dvar (batch, M), dlogml (batch) fp64, dinfo (batch) int32.
 * Work is enqueued on `hip_stream` (see CGP_STREAM_CTX above: NULL = the legacy default stream,
 * CGP_STREAM_CTX = the context's own stream) and the call returns without synchronising.  No jitter retry happens here: read dinfo and re-submit the failed
 * fits with djitter set (cgp_fit_predict_batch does exactly that). */
int cgp_fit_predict_batch_device(cgp_ctx *ctx, int batch, int N, int d, int M, int kernel_id,
                                 const void *dX, const void *dy, const void *dXs, const double *dtheta,
                                 const double *djitter, int include_noise, void *dmean, void *dvar,
                                 double *dlogml, int *dinfo, void *hip_stream);

/* ---- multi-device sweep (SURVEY.md 8b "cgp_fit_predict_batch(ctx[], ...)", 8e) --------------------
 * One engine context and one host thread per listed device; a batch of independent windows is cut into
 * contiguous per-device blocks (device i gets fits [start_i, stop_i), the first batch % ndev devices
 * one fit more -- cgp_sweep_shard returns the range), every block runs cgp_fit_predict_batch on its own
 * device concurrently, and the per-fit summaries {logml, max sigma = 2 sqrt(max var), info} are gathered
 * on the host in global fit order (`summary` is (batch, 3), may be NULL).  No data-path collective:
 * the path shards across fits only, a single fit is never split.  This is the entry point that lets
 * the reference's C++ ROS host (gp_predictor) shard a Monte-Carlo ensemble without Python; the
 * one-process-per-GPU form over RCCL is corenav_gp_amd/sharding.py + bench.py --gpus N.  `devices` may
 * name a device more than once (several contexts on one GPU: the self-test of a one-GPU box).
 * Argument meaning, outputs and return value as cgp_fit_predict_batch; max_batch_total = the largest
 * batch a call will carry.  Returns NULL / CGP_ECAPACITY like cgp_create / cgp_fit_predict_batch.
 * Reproducibility across shardings: a fit's result does not depend on its slot in a call or on its neighbours, and
 * in CGP_F64 not on how many fits share the call either -- a sweep equals one context running the whole batch BITWISE.
 * In CGP_F32 calls of up to 96 fits factor the 128 x 128 diagonal tile in a different (fatter) form than larger calls
 * do, so a fit's fp32 result depends on the size of the call it rides in to single-precision rounding (inside the
 * 1e-3 bar): a 512-fit sweep over 8 shards of 64 agrees with the 512-fit call to rounding, not bitwise
 * (tests/test_gpu_parity.py::test_sweep_fp32_matches_single_context_to_rounding). */
typedef struct cgp_sweep cgp_sweep;
cgp_sweep *cgp_sweep_create(const int *devices, int ndev, int max_n, int max_m, int max_d, int max_batch_total,
                            int dtype);
void cgp_sweep_destroy(cgp_sweep *sweep);
int cgp_sweep_ndev(const cgp_sweep *sweep);
int cgp_sweep_shard(const cgp_sweep *sweep, int batch, int i, int *start, int *stop);
int cgp_sweep_fit_predict(cgp_sweep *sweep, int batch, int N, int d, int M, int kernel_id, const double *X,
                          const double *y, const double *Xs, const double *theta, int theta_stride,
                          int include_noise, double *mean, double *var, double *logml, int *info,
                          double *summary);
/* Device-resident form: shard i's inputs already live on device i in the layout of cgp_fit_predict_batch_device
 * (its (stop_i - start_i) fits only), every argument an array of ndev per-shard device pointers -- dX[i] (n_i, d, N),
 * dy[i], dXs[i], dtheta[i] (n_i, CGP_MAX_THETA) fp64, djitter[i] or a NULL array, outputs dmean[i], dvar[i], dlogml[i],
 * dinfo[i]; hip_streams[i] the stream of device i to enqueue on (a NULL ARRAY = every context's own stream, an element
 * follows the hip_stream convention above).  Returns when every shard's work has been ENQUEUED, without synchronising:
 * no PCIe copy and no host round trip inside the call (a 64-fit shard is 0.8 ms of device time).  Wait with the streams
 * you passed, or cgp_sweep_synchronize for the contexts' own streams.  No jitter retry, as cgp_fit_predict_batch_device.
 * Threads: shard 0 is issued by the calling thread, the others by persistent worker threads created with the sweep (no
 * thread is created per call); a sweep over ONE device is exactly that context's call.  Current device: every entry point
 * that takes a context makes that context's device current on the thread it runs on (hipSetDevice) and leaves it so; after a
 * cgp_sweep_* call the CALLING thread's current device is therefore devices[0] -- a caller with its own work on another device
 * sets it again. */
int cgp_sweep_fit_predict_device(cgp_sweep *sweep, int batch, int N, int d, int M, int kernel_id, const void *const *dX,
                                 const void *const *dy, const void *const *dXs, const double *const *dtheta,
                                 const double *const *djitter, int include_noise, void *const *dmean, void *const *dvar,
                                 double *const *dlogml, int *const *dinfo, void *const *hip_streams);
int cgp_sweep_synchronize(cgp_sweep *sweep);
/* The engine context of shard i (owned by the sweep): for cgp_set_streams, cgp_set_refine, cgp_last_error, cgp_profile_* on a shard. */
cgp_ctx *cgp_sweep_context(const cgp_sweep *sweep, int i);

/* Stream groups of a batch.  n = 0 (the default): the engine decides -- an fp32 call of 56 ... 96 fits (BASELINE configs[2] as
 * sharded over 8 GPUs: 64 per GPU) is cut into TWO groups whose launch schedules run concurrently, group 0 on the caller's
 * stream and group 1 on one of the context's worker streams, forked from / joined to the caller's stream with events, so
 * the chain-bound early launches of one group run beside the MFMA-bound ones of the other (0.99 -> 0.90 ms per 64-fit
 * call on one context); every other call is one group.  Whether two streams really overlap depends on how the runtime mapped
 * them onto its hardware queues (the process's history), so the engine MEASURES: per (caller stream, fits, block steps), after two
 * warm-up calls, four such calls run as two groups and four as one, bracketed by events on the caller's stream; the faster form is
 * kept, measured again after 24, 48, ... 256 calls (a context's first calls run on a part still coming out of idle, where both
 * forms measure alike) and whenever three monitored calls in a row come out 1.3 x slower than the chosen form measured.  No call
 * blocks for this: every decision is read with hipEventQuery (a call whose answer is not in yet runs as two groups), and a stream
 * that is being captured into a hipGraph is never touched with a timing event (the call takes the form already decided, or two
 * groups forked / joined with plain events, which the capture records as edges).  A caller that wants the form fixed passes n = 1
 * or n = 2.  n = 1: always one group.  n = 2..8: up to n groups for full-batch
 * calls too (hundreds of fits gain nothing measurable).  Results do not depend on the setting (a fit's arithmetic is the
 * same in any group). */
int cgp_set_streams(cgp_ctx *ctx, int n);

/* Development aid (-DCGP_ABLATION builds; all zero otherwise): in-kernel s_memtime sums.  [0, 8) potf2
 * phases of block 0 (CGP_DBG & 512); [64 + 8k, 64 + 8k + 8) per-phase sums of k_panel at block step k
 * over all workgroups, slot 7 of each group = workgroup count (CGP_DBG & 1024).  Reading resets them. */
#define CGP_DEBUG_SLOTS 512
int cgp_debug_read(cgp_ctx *ctx, long long out[CGP_DEBUG_SLOTS]);
/* Development aid: the raw result record of the last short-window launch (cgp_nll_grad / cgp_optimize* of a window of at
 * most 160 samples in an fp64 context, csrc/cgp_small.hpp): [0] logML, [1] evaluations, [2] L-BFGS status, [3] iterations,
 * [4] info, [5] jitter, [8..18) gradient, [20..30) theta, [32..48) per-phase s_memtime sums (-DCGP_ABLATION builds; zero
 * otherwise: tools/small_phases.py). */
#define CGP_SMALL_OUT 48
int cgp_debug_small(cgp_ctx *ctx, double out[CGP_SMALL_OUT]);
/* Development aid: device addresses and byte sizes of the context's large buffers, as pairs
 * out[2 i] = address, out[2 i + 1] = bytes for i = 0 factor panels (Lw), 1 W images (Winv), 2 diagonal-tile images,
 * 3 panel-tile images, 4 inputs X, 5 running predictive sums, 6 latency partial tiles, 7 latency images
 * (tools/ctx_placement.py prints them next to the timings of a context). */
#define CGP_DEBUG_BUFFERS 8
int cgp_debug_buffers(cgp_ctx *ctx, unsigned long long out[2 * CGP_DEBUG_BUFFERS]);

/* ---- online sliding-window GP (BASELINE configs[3]; not reference behaviour) -------------------
 * `nwin` independent windows of at most N samples each live on the device.  cgp_window_push feeds
 * T ticks to every window in ONE launch: per tick the oldest sample leaves a full window (rank-1
 * Cholesky update), the new one enters (forward substitution), and the tick's outputs are the
 * one-step-ahead predictive mean / variance of the incoming y BEFORE it is added, and the log
 * marginal likelihood of the window after it.  theta (nwin, theta_stride) is fixed per window.
 * xs (nwin, T, d), ys (nwin, T); outputs (nwin, T).  Returns 0, or the 1-based tick at which a window
 * lost positive definiteness.  cgp_window_push blocks until the outputs are in the caller's arrays (a small push is read and
 * written by the kernels in pinned host memory; a one-tick push waits on the windows' status words there rather than on the
 * stream: 74 us per tick of one N = 512 window from a C caller).  Steady-state ticks of a longer push go two per pass over the
 * factors, four from 512 windows: the outputs are those of the tick-by-tick stream to rounding. */
int cgp_window_init(cgp_ctx *ctx, int nwin, int N, int d, int kernel_id, const double *theta, int theta_stride);
int cgp_window_push(cgp_ctx *ctx, int T, const double *xs, const double *ys, int include_noise,
                    double *pred_mean, double *pred_var, double *logml);
/* Device-resident variant for streaming benchmarks: dxs/dys/outputs are device pointers, enqueued on
 * hip_stream (NULL = legacy default stream, CGP_STREAM_CTX = the context's own) without synchronising. */
int cgp_window_push_device(cgp_ctx *ctx, int T, const double *dxs, const double *dys, int include_noise,
                           double *dpred_mean, double *dpred_var, double *dlogml, void *hip_stream);
/* Current size of window `w` and the first failing tick (0 = none). */
int cgp_window_state(cgp_ctx *ctx, int w, int *n, int *info);

/* ---- fp32 contexts: mixed-precision refinement of alpha and the predictive mean -----------------
 * After the single-precision factorisation: alpha_0 = L^-T L^-1 y from the factor, then `steps` times
 *   r = y - Ky alpha   in DOUBLE precision, Ky entries re-evaluated from X on the fly (never stored),
 *   alpha += L^-T L^-1 r   through the fp32 factor, alpha kept in double,
 * and mean = K*^T alpha with K* evaluated in double -- GPy's own form of the mean (gp_slip_node.py:48 m.predict: mu = k*^T
 * woodbury_vector).  One step takes the mean of a dense one- or two-dimensional window from ~1e-3 of the oracle to ~1e-6
 * (tests/fuzz/d1_fp32_error.py); variance and logML come from the factor as before.  steps = -1 (the default): the engine decides --
 * one step (two for windows of more than 1 024 samples, where a step contracts less); for every fit of a window of d <= 3 input dimensions (the RBF x Brownian kernel included: +40 % per call at
 * N = 1024, M = 599), and for d > 3 only for the fits whose factor shows a dense window (prior variance / geometric mean of
 * the pivots L_ii^2 >= 12: the unrefined mean's error follows that ratio, tests/fuzz/rho_vs_error.py) -- BASELINE configs[2] (d = 6,
 * ratio 3 ... 11.5) has no such fit and pays one launch whose workgroups return at once (not measurable: 0.696 ms per 64-fit
 * call either way); a call with such a fit pays the latency of one refinement (~0.27 ms at N = 1024) whatever their number.  0: never; 1..3: that many
 * steps for every fit of every fp32 call.  cgp_get_alpha then returns the refined alpha (double precision).  No effect on
 * CGP_F64 contexts, nor on windows of more than 13 000 samples (the solve keeps the window's alpha in LDS). */
int cgp_set_refine(cgp_ctx *ctx, int steps);

/* ---- per-kernel timing for the roofline line (bench.py) --------------------------------------
 * on = 1: every launch is bracketed by hipEvents on its stream; on = 2 + k: only the update launch of block
 * step k (the other launches of the schedule stay back to back, so the bracketed one runs as it does in an
 * untimed step); on = 0: off.  cgp_profile_read drains the events.
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
 * copied to time_out / slipwin_out and *n_out = its length.  If the window is longer than `cap` the
 * first cap entries are copied, *n_out still holds the full length and CGP_ECAPACITY is returned
 * (never a silent truncation).  *slip_out = the tick's slip value (may be NULL). */
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
