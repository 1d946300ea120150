// cgp_engine.hip -- context, launch schedule and the C ABI of include/corenav_gp.h.
// Host side of the slip-GP hot path; the arithmetic is in cgp_kernels.hpp (HIP, gfx950 only).
// There is no CPU fallback anywhere in this file: without a usable device cgp_create returns NULL.
#include "../../include/corenav_gp.h"
#include "cgp_kernels.hpp"
#include "cgp_kernels_fused.hpp"
#include "cgp_window.hpp"
#include "cgp_lookahead.hpp"
#include "cgp_small.hpp"
#include "cgp_refine.hpp"
#include "gp_predictor_core.hpp"
#include "gp_predictor.h"
#include "lbfgs.hpp"
#include "slip_recorder.hpp"

#include <algorithm>
#include <atomic>
#include <climits>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace cgp;

static_assert(CGP_MAX_D == MAXD, "header / kernel MAXD mismatch");
static_assert(CGP_DEBUG_SLOTS == DBG_SLOTS, "header / kernel debug slot mismatch");
static_assert(CGP_MAX_THETA == MAX_THETA, "header / kernel MAX_THETA mismatch");
static_assert(CGP_SMALL_OUT == SM_OUT, "header / kernel short-window record mismatch");

namespace {

#ifdef CGP_AB
constexpr bool kAbBuild = true;
#else
constexpr bool kAbBuild = false;
#endif

struct ProfRec {
  int kernel;
  hipEvent_t a, b;
  double flops;
};

}  // namespace

struct cgp_ctx {
  int device = 0, dtype = CGP_F64;
  int max_n = 0, max_m = 0, max_d = 0, max_batch = 0;
  int NTmax = 0, ETmax = 0, ld = 0;
  size_t esz = 8;
  hipStream_t stream = nullptr;
  // worker streams: a batch is cut into groups whose schedules run concurrently, so the
  // one-workgroup-per-fit potf2 launches of one group overlap the MFMA launches of the others
  static constexpr int kMaxStreams = 8;
  hipStream_t wstream[kMaxStreams] = {nullptr};
  hipStream_t hstream = nullptr;   // highest stream priority: for a latency chain that must be dispatched AHEAD of bulk work on the caller's stream
  hipEvent_t ev_fork = nullptr, ev_join[kMaxStreams] = {nullptr};
  // Two stream groups or one for an fp32 call of 56 ... 96 fits?  Decided per caller stream by timing one call each way (run_schedule).
  struct GroupTune {
    hipStream_t stream = nullptr;
    int batch = 0, NT = 0;     // the entry is for calls of this many fits and block steps on `stream`
    int state = -1;            // -1 free, else eligible calls seen on this stream so far (kTuneDecide and beyond: decided)
    int groups = 2;
    float ms[2] = {0.f, 0.f};
    hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t mon[2] = {nullptr, nullptr};   // after the decision: one call in a while bracketed, read later with hipEventQuery
    bool mon_pending = false;
    int slow = 0;                              // consecutive monitored calls well above what the chosen form measured
    int retune = 24;                           // decided calls until the two forms are measured again: 24, 48, ... kRetune
    unsigned long long used = 0;
  } tune[4];
  unsigned long long tune_clock = 0;
  hipEvent_t ev_look[3 * 64] = {nullptr};  // look-ahead schedule: diag / P1 / P2 completion per step
  int nstreams = 0;   // cgp_set_streams: 0 = the engine decides (two groups for fp32 calls of 56 ... 96 fits, else one); n >= 1 = as told
  // device buffers
  void *Lw = nullptr, *Winv = nullptr, *dX = nullptr, *dXs = nullptr, *dy = nullptr;
  void *Lp = nullptr;   // fp32: the panel tiles of a mid-size call as bf16 planes [mid_cap][3][lw_stride] (cgp_kernels_fused.hpp, bx6p_loop)
  void *dmean = nullptr, *dvar = nullptr, *dalpha = nullptr;
  // fp32 contexts: mixed-precision refinement of alpha and the predictive mean (cgp_refine.hpp, cgp_set_refine)
  double *dref_r = nullptr, *dref_a = nullptr;   // [max_batch][alpha_stride] residual, alpha in double precision
  int *dref_flag = nullptr;                      // [max_batch] fits k_finalize marked as dense (d > 3 under the default setting)
  int refine = -1;        // -1: the engine decides (one step: every fit at d <= 3, the dense ones beyond), 0: never, 1..3: that many steps for every call
  bool refined = false;   // dref_a holds the refined alpha of the fits of the last fit call (cgp_predict / cgp_get_alpha use it)
  double *dtheta = nullptr, *djitter = nullptr, *dlogml = nullptr, *dprep = nullptr;
  long long *ddbg = nullptr;
  double *dgpart = nullptr;
  void *dpart = nullptr;   // latency schedule: partial tiles [lat_cap][slots][SK_MAX][128*128]
  int *dticket = nullptr;  //                   arrival tickets [lat_cap][slots]
  int lat_cap = 0;         //                   fits the latency slabs are sized for: min(LAT_FITS_ALLOC, max_batch)
  int *dwready = nullptr;  //                   published block steps [lat_cap]
  void *dlatimg = nullptr;  //                  pre-updated diagonal tiles [lat_cap][2][LAT_IMG_MAX][DPART]
  double *dmacc = nullptr;  // [max_batch][2][max_m] running predictive sums (throughput schedule, fp64)
  void *ddiagimg = nullptr;  // [max_batch][2][DPART] pre-updated diagonal tiles (throughput schedule, diag_next)
  void *dpanimg = nullptr;   // [min(max_batch, mid cap)][2][DPART] pre-updated kind-A panel tiles (k_panel kind C)
  int mid_cap = 0;
  // one-launch schedule of a mid-size call (k_sched): task list of the last (fits, NT, ET) shape, flag words, one image slot per tile index
  void *dsched_tasks = nullptr, *dsched_flags = nullptr, *dsched_bimg = nullptr, *dsched_cimg = nullptr;
  size_t sched_tasks_cap = 0, sched_flags_cap = 0, sched_bimg_cap = 0, sched_cimg_cap = 0;
  int sched_key[3] = {0, 0, 0}, sched_ntasks = 0, sched_grid = 0, n_cu = 0;
  // cgp_fit_predict_batch staging, grown on demand and kept: pinned host buffers (hipHostMalloc) so the
  // H2D / D2H copies are real asynchronous DMA, and a raw fp64 device buffer the pack kernels read
  void *pin_in = nullptr, *pin_out = nullptr, *draw = nullptr;
  size_t pin_in_cap = 0, pin_out_cap = 0, draw_cap = 0;
  double *la_buf = nullptr;  // cgp_predict_stop_batch staging, grown on demand
  int *la_ibuf = nullptr;
  size_t la_nd = 0, la_ni = 0;
  int sk_slots = 0;
  size_t lat_units = 0;   // fits x tile slots the latency schedule's partial-tile slab holds
  int lat_img_fits = 0;   // fits the pre-update image buffer of the latency schedule holds (only windows of >= 3 block steps use it)
  // sliding windows (cgp_window_*)
  WindowArgs win{};
  int nwin = 0;
  int win_o = 0, win_n = 0;   // origin / size of the windows (they advance in lock-step), mirrored on the host to cut a push into launches
  void *winbuf[8] = {nullptr};
  // cgp_window_push staging, grown on demand and kept: one pinned host block and one device block per direction
  void *win_pin = nullptr, *win_dev = nullptr;
  size_t win_pin_cap = 0, win_dev_cap = 0;
  // gradient-mode evaluations (cgp_nll_grad / cgp_optimize*): theta + jitter out, partial sums + logML + info back, one pinned block
  void *opt_pin = nullptr;
  size_t opt_pin_cap = 0;
  int *dinfo = nullptr;
  double *dsmall = nullptr;   // [max_batch][SM_OUT] results of the one-launch short-window kernel (k_small, fp64 contexts)
  unsigned short *dsmdeal = nullptr;   // k_small's helper work lists (sm_build_deal) of every NB <= SM_MAX_NB, at smdeal_off[NB]
  int *dsmdone = nullptr;              // k_small_predict's count of finished workgroups (the node callback polls a pinned word the last one writes)
  int small_seq = 0;
  size_t smdeal_off[SM_MAX_NB + 1] = {0};
  std::vector<double> lazy_win;   // [X (N, d) | y] of the window a short-window kernel evaluated in place (ensure_fitted uploads it)
  int pending_tab = 0;               // tick-grid table size cgp_fit_predict_batch found for the batch it is about to submit (0: none)
  bool last_small_dev = false;       // ... or window 0's record of the last batched fit + predict launch, still in dsmall
  double last_small[SM_OUT] = {0};   // the last single-window short-window record (the kernels write it to pinned host memory)
  size_t lw_stride = 0, winv_stride = 0, alpha_stride = 0;
  // state of the last single fit (cgp_fit -> cgp_predict)
  bool have_fit = false;
  bool lazy_fit = false;      // the window, theta and jitter of the last short-window evaluation are on the device but the factor
                              // panel is not: cgp_predict / cgp_get_alpha / cgp_get_factor run the fit schedule first (ensure_fitted)
  double f_meandiag_x = 0.0;  // mean |x| of that window's first input (GPy jitchol's mean(diag) for the Brownian factor)
  int fN = 0, fd = 0, fkernel = 0;
  double ftheta[CGP_MAX_THETA] = {0};
  double fjitter = 0.0;
  // profiling
  bool prof = false;
  int prof_step = -1;  // >= 0: only the update launch of this block step is bracketed (cgp_profile_enable(2 + k))
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  double prof_ms[CGP_PROF_KERNELS] = {0}, prof_flops[CGP_PROF_KERNELS] = {0};
  long long prof_n[CGP_PROF_KERNELS] = {0};
  std::string err;
};

namespace {

int ensure_fitted(cgp_ctx *c);   // below, with the short-window paths
bool small_batch_predict_ok(const cgp_ctx *c, int batch, int N, int d, int M);
int tick_table_entries_batch(int kid, int d, int batch, const double *X, int N, const double *Xs, int M);
struct SmallRaw;
struct SmallDev {   // a batch resident in the caller's device buffers (cgp_fit_predict_batch_device): no ladder, jitter as given
  const double *X, *y, *Xs, *theta, *jitter;
  double *logml;
  int *info;
};
int small_predict_launch(cgp_ctx *c, int batch, int N, int d, int M, int kid, int include_noise, double *dmean, double *dvar, double *dout,
                         hipStream_t s, const SmallRaw *raw = nullptr, const SmallDev *dev = nullptr, int tab_n = 0, int *done_flag = nullptr,
                         int done_seq = 0);
int node_callback_small(cgp_ctx *c, const double *time_array, const double *slip_array, int n, int kid, double *theta, int max_evals,
                        double *mean, double *sigma, int cap, int *m_out);
bool grow_pinned(void *&p, size_t &cap, size_t bytes);
bool grow_device(void *&p, size_t &cap, size_t bytes);

inline int ntheta(int kid, int d) { return kid == CGP_KERNEL_SE_ISO ? 3 : (kid == CGP_KERNEL_SE_ARD ? d + 2 : 4); }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

bool hip_ok(cgp_ctx *c, hipError_t e, const char *what) {
  if (e == hipSuccess) return true;
  if (c) c->err = std::string(what) + ": " + hipGetErrorString(e);
  return false;
}
#define HIP_TRY(ctx, expr)                                   \
  do {                                                       \
    if (!hip_ok((ctx), (expr), #expr)) return CGP_EHIP;      \
  } while (0)

hipEvent_t get_event(cgp_ctx *c) {
  if (!c->pool.empty()) {
    hipEvent_t e = c->pool.back();
    c->pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

struct Launcher {
  cgp_ctx *c;
  hipStream_t s;
  int pending = -1;
  void begin(int kernel, double flops, int step = -1) {
    if (!c->prof) return;
    if (c->prof_step >= 0 && !(kernel == 0 && step == c->prof_step)) return;  // one update launch per schedule
    ProfRec r{kernel, get_event(c), get_event(c), flops};
    (void)hipEventRecord(r.a, s);
    c->recs.push_back(r);
    pending = (int)c->recs.size() - 1;
  }
  void end() {
    if (!c->prof || pending < 0) return;
    (void)hipEventRecord(c->recs[pending].b, s);
    pending = -1;
  }
};

// Batches up to this size take the latency schedule: the measured crossover against the (mid-size) throughput
// schedule (tools/lat_crossover.sh, round 3: N = 2048 fp64 8 fits 1.87 vs 2.64 ms, 12 fits 2.72 vs 2.66; N = 1024 fp32
// 20 fits 0.641 vs 0.672 ms, 24 fits 0.769 vs 0.712) is 11 fits in fp64 and 20 in fp32 (16 / 24 before the mid-size form).
constexpr int LAT_FITS_F64 = 11, LAT_FITS_F32 = 20;
// The shorter the window, the larger the call the latency schedule still wins (tools/lat_crossover.sh by window length, end of
// round 3, ms per call latency vs throughput).  fp64: N = 256 32 fits 0.171 vs 0.240, 48 fits 0.225 vs 0.245; N = 512 28 fits 0.417 vs
// 0.422 (no extra-row split), 32 fits 0.464 vs 0.456; N = 768 20 fits 0.607 vs 0.657, 24 fits 0.681 vs 0.681; N = 1024 16 fits 0.835 vs
// 0.920, 20 fits 1.010 vs 0.954; N = 1536 12 fits 1.487 vs 1.635, 16 fits 1.926 vs 1.682; N = 2048 11 (above).  fp32: N = 256 32 fits
// 0.124 vs 0.133, 48 fits 0.154 vs 0.138; N = 512 24 fits 0.261 vs 0.272, 28 fits 0.290 vs 0.276; N = 768 20 fits 0.412 vs 0.414; N = 1024
// 20 (above).  The slabs are sized for LAT_FITS_SHORT fits.
// (Round 4, tools/check_crossovers.py: fp64 N = 256 33 fits 0.168 latency vs 0.205 throughput -- the limit of 32 was the slab's,
// not the crossover's; with round 3's 48 fits 0.225 vs 0.245 the fp64 limit for one and two block steps is 48.)
constexpr int LAT_FITS_SHORT = 32, LAT_FITS_SHORT64 = 48;
inline int lat_fits_by_steps(bool f64, int NT) {
  if (NT <= 2) return f64 ? LAT_FITS_SHORT64 : LAT_FITS_SHORT;
  if (f64) return NT <= 4 ? 28 : NT <= 6 ? 22 : NT <= 8 ? 18 : NT <= 12 ? 13 : LAT_FITS_F64;
  return NT <= 4 ? 24 : LAT_FITS_F32;
}
constexpr int FUSED64_BELOW = 512;                // fp64 throughput schedule: diagonal tiles inside the panel launches below this batch
// Calls too small to fill the chip (a launch then lasts as long as its longest workgroup chain): the next launch's
// kind-A tile is pre-updated by a kind-C workgroup (k_panel), and fp32 takes the deep-prefetch loops (DEEP).
// Measured crossover (tools/r3_mid2.sh, same box, CGP_MID_FITS either side): fp32 N = 1024: 32 fits +14 %, 48 +13 %,
// 64 +9 %, 96 +2 %, 128 -4 %; fp64 N = 2048: 24 fits +18 %, 32 +11 %, 48 +4 %, 64 -2 %.
constexpr int MID_FITS_F64 = 48, MID_FITS_F32 = 96;
#ifndef CGP_NO_EXTRA_SPLIT
#define CGP_NO_EXTRA_SPLIT 0   // `make variant` A/B: mid-size calls keep the extra rows inside the factorisation launches
#endif
constexpr bool kNoExtraSplit = CGP_NO_EXTRA_SPLIT != 0;
#ifndef CGP_LAT_MIN_NT
#define CGP_LAT_MIN_NT 1   // block steps from which a handful of fits takes the latency schedule (3 until the end of round 3: `make variant` A/B)
#endif
constexpr size_t kOptPinIn = CGP_MAX_THETA + 8;   // doubles at the head of the gradient-mode pinned block: theta, jitter
#ifndef CGP_WIN_PAIRS
#define CGP_WIN_PAIRS 1   // sliding window: steady-state ticks two per pass over the factor (`make variant`: 0 = every tick on its own)
#endif
constexpr bool kWinPairs = CGP_WIN_PAIRS != 0;
constexpr size_t kWinZeroCopyBytes = 16 * 1024;   // cgp_window_push blocks up to this size are read / written in pinned host memory by the kernels
constexpr int kWinPackLds = 72 * 1024;       // pack windows into a workgroup only while two workgroups still fit a CU's LDS ...
#ifndef CGP_WIN_MULTI
#define CGP_WIN_MULTI 4   // steady-state ticks per pass over the factor where the window is long enough (k_window_multi); 0 = pairs only
#endif
constexpr int kWinMulti = CGP_WIN_MULTI > 2 ? CGP_WIN_MULTI : 4;
constexpr bool kWinUseMulti = CGP_WIN_MULTI > 2;
constexpr int kWinMultiMinWindows = 512;
constexpr size_t kWinPairStage = 3 * WPB * 64 * sizeof(double);   // k_window_pairs: the three sweep waves' staged trips (24 KB)
constexpr int kWinPackMinGroups = 512;
constexpr int kWinWideMax = 256;            // single-tick kernel: up to this many windows 512 threads per window       // ... and the chip still gets two workgroups per CU
// fp64 mid-size calls put their extra rows on a second stream when there is enough of them: fits x block steps >= this
// (tools/r3_xs_n.sh, same box, with / without, ms per call: N = 2048 28 fits 3.77 / 3.70, 36 fits 4.25 / 4.45, 48 fits 5.01 / 5.70;
// N = 1536 36 fits 2.37 / 2.38, 48 fits 2.71 / 3.01; N = 1024 36 fits 1.27 / 1.17, 48 fits 1.29 / 1.31; N = 512 28 fits 0.52 / 0.42)
constexpr int XSPLIT64_WORK = 480;
#ifndef CGP_XSPLIT32_FITS
#define CGP_XSPLIT32_FITS 0    // fp32 mid-size calls from this many fits: extra rows on the second stream as 64-row tiles (k_rows64); 0 = never.
                               // Measured (round 5, ms per call without / with): 24 fits 0.629 / 0.647, 32 0.686 / 0.777, 48 0.770 / 0.791, 64 0.994 / 0.954,
                               // 96 1.282 / 1.270 -- bitwise the one-stream results, but two stream GROUPS (0.895 at 64 fits) beat it: not taken
#endif
constexpr int XSPLIT32_FITS = CGP_XSPLIT32_FITS;
constexpr int MID_FITS_ALLOC = kAbBuild ? 512 : (MID_FITS_F64 > MID_FITS_F32 ? MID_FITS_F64 : MID_FITS_F32);
// fp64 by window length (tools/r3_mid_n2.sh, end of round 3, with the extra-row split by work; mid-size form off / on, ms per call):
// N = 2048 64 fits 7.26 / 6.87, 96 fits 9.80 / 9.67; N = 1536 96 fits 5.14 / 4.95; N = 1024 96 fits 2.22 / 2.10; N = 768 64 fits 1.06 /
// 1.01, 80 fits 1.16 / 1.16, 96 fits 1.27 / 1.28; N = 512 96 fits 0.645 / 0.663 -- up to 96 fits from eight block steps, 64 from six.
template <typename T> inline int mid_fits(int NT) {     // ablation build: CGP_MID_FITS moves the crossover (measurement)
  if constexpr (kAbBuild) {
    const char *e = getenv("CGP_MID_FITS");
    if (e) return std::max(0, std::min(atoi(e), MID_FITS_ALLOC));
  }
  if (sizeof(T) == 8) return NT >= 8 ? MID_FITS_F32 : NT >= 6 ? 64 : MID_FITS_F64;
  return MID_FITS_F32;
}
constexpr int LAT_FITS_ALLOC = kAbBuild ? 64 : LAT_FITS_SHORT64;  // slabs are sized for min(this, max_batch) fits
template <typename T> inline int lat_fits(int NT) {     // ablation build: CGP_LAT_FITS moves the crossover (measurement)
  if constexpr (kAbBuild) {
    const char *e = getenv("CGP_LAT_FITS");
    if (e) return std::max(0, std::min(atoi(e), LAT_FITS_ALLOC));
  }
  return lat_fits_by_steps(sizeof(T) == 8, NT);
}

inline size_t alpha_lds_bytes(int NT) { return (size_t)(NT * TS + TS) * sizeof(double); }
template <typename T> constexpr int upd_lds_bytes() { return 4 * KT * LDST * (int)sizeof(T); }
template <typename T> constexpr int panel_lds_bytes() {   // + z of one block column; fp32 with the bf16 planes: planes + Gram inputs + z
  return (kF32Bf16x6 && sizeof(T) == 4) ? (bx_z_offset<false>() + TS) * 4 : upd_lds_bytes<T>() + TS * (int)sizeof(T);
}
static_assert(panel_lds_bytes<float>() >= WIMG * 4 && panel_lds_bytes<float>() >= upd_lds_bytes<float>() + TS * 4,
              "the panel kernels stage W_k's image (and k_rows64 its chunk buffers) over the loop's LDS");
template <typename T> constexpr int paneldiag_lds_bytes() { return std::max(panel_lds_bytes<T>(), diag_lds_elems<T>() * (int)sizeof(T)); }
// mid-size build: fp32 factors the diagonal tile in the fat form (potf2_tile), see mid_fat
template <typename T> constexpr int paneldiag_mid_lds_bytes() {
  return mid_fat<T, true>() ? std::max(paneldiag_lds_bytes<T>(), potf2_lds_elems<T>() * (int)sizeof(T)) : paneldiag_lds_bytes<T>();
}
static_assert(!kF32Bf16x6 || (bx_z_offset<true>() + TS) * 4 <= paneldiag_mid_lds_bytes<float>(), "mid-size build: the two plane buffers do not fit its LDS");
// the deep loop's chunk ring + z of one block column must fit what the mid-size build's launches carry
static_assert(deep_ring<float, true>() * KT * LDST * 4 + TS * 4 <= paneldiag_mid_lds_bytes<float>() &&
              deep_ring<double, true>() * KT * LDST * 8 + TS * 8 <= paneldiag_mid_lds_bytes<double>(), "mid-size build: ring does not fit its LDS");
template <typename T> constexpr int potf2_lds_bytes() {
  return (TS * LDP + 8 * DB * DB + 4 * DB * DB) * (int)sizeof(T) + 16;
}

#ifndef CGP_F32_FULL_DEEP
#define CGP_F32_FULL_DEEP 0   // `make variant` A/B: the deep-prefetch fp32 loops at full batch too (measured slower: 128 -> 168 VGPRs)
#endif
// hipFuncSetAttribute applies to the CURRENT device's function object: called from cgp_create after
// hipSetDevice, once per (device, dtype).
// (two contexts may be created from two threads at once: the per-device "done" marks are guarded by a mutex)
std::mutex g_attr_mutex;
template <typename T> int set_lds_attrs(int device) {
  static bool done[64] = {false};
  std::lock_guard<std::mutex> lk(g_attr_mutex);
  if (device >= 0 && device < 64 && done[device]) return 0;
  const int upd = upd_lds_bytes<T>(), tile = potf2_lds_bytes<T>();
  auto set = [](const void *fn, int bytes) { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess; };
  bool ok = true;
  ok = ok && set(reinterpret_cast<const void *>(&k_panel<T, false>), panel_lds_bytes<T>());
  ok = ok && set(reinterpret_cast<const void *>(&k_grad<T>), upd);
  ok = ok && set(reinterpret_cast<const void *>(&k_tile_sk<T>), tile);
  ok = ok && set(reinterpret_cast<const void *>(&k_trmm_sk<T>), upd);
  ok = ok && set(reinterpret_cast<const void *>(&k_diag_lean<T>), paneldiag_lds_bytes<T>());
  ok = ok && set(reinterpret_cast<const void *>(&k_panel<T, true>), paneldiag_lds_bytes<T>());
  ok = ok && set(reinterpret_cast<const void *>(&k_panel<T, true, true, true>), paneldiag_mid_lds_bytes<T>());
#ifdef CGP_AB
  ok = ok && set(reinterpret_cast<const void *>(&k_sched<T>), paneldiag_mid_lds_bytes<T>());
#endif
  if constexpr (sizeof(T) == 4) ok = ok && set(reinterpret_cast<const void *>(&k_refine_solve<float>), 160 * 1024);
  if constexpr (sizeof(T) == 4) ok = ok && set(reinterpret_cast<const void *>(&k_refine_gated<float>), 160 * 1024);
  if constexpr (mid_fat<T, true>()) ok = ok && set(reinterpret_cast<const void *>(&k_diag_lean<T, true>), paneldiag_mid_lds_bytes<T>());
  if constexpr (CGP_F32_FULL_DEEP && sizeof(T) == 4) ok = ok && set(reinterpret_cast<const void *>(&k_panel<T, true, true, false>), paneldiag_lds_bytes<T>());
#ifdef CGP_AB
  ok = ok && set(reinterpret_cast<const void *>(&k_diag<T>), tile);
#endif
  if (!ok) return -1;
  if (device >= 0 && device < 64) done[device] = true;
  return 0;
}

constexpr size_t kLdsPerWorkgroup = 160 * 1024;   // gfx950
// k_small<BROWN, DMAX>: the reference's kernel, and SE kernels compiled for d <= 1 / 3 / 8 (the smallest that fits is launched)
template <typename F> int for_each_small_kernel(F &&f) {
  int rc = f(reinterpret_cast<const void *>(&k_small<true, 1>));
  if (rc == 0) rc = f(reinterpret_cast<const void *>(&k_small<false, 1>));
  if (rc == 0) rc = f(reinterpret_cast<const void *>(&k_small<false, 3>));
  if (rc == 0) rc = f(reinterpret_cast<const void *>(&k_small<false, 8>));
  return rc;
}
inline size_t small_predict_lds(int NB, int d) { return ((small_lds_bytes(NB, d) + 15) & ~(size_t)15) + small_predict_lds_extra(NB, d); }
int set_small_attr(int device) {
  static bool done[64] = {false};
  std::lock_guard<std::mutex> lk(g_attr_mutex);
  if (device >= 0 && device < 64 && done[device]) return 0;
  const int rc = for_each_small_kernel([](const void *fn) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerWorkgroup) == hipSuccess ? 0 : -1;
  });
  if (rc != 0) return -1;
  const void *pk[] = {reinterpret_cast<const void *>(&k_small_predict<true, 1>), reinterpret_cast<const void *>(&k_small_predict<false, 1>),
                      reinterpret_cast<const void *>(&k_small_predict<false, 3>), reinterpret_cast<const void *>(&k_small_predict<false, 8>)};
  for (const void *fn : pk)
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerWorkgroup) != hipSuccess) return -1;
  if (device >= 0 && device < 64) done[device] = true;
  return 0;
}

// Algorithmic flops of the update launches of block step k (DESIGN.md "Kernels"): lower-trapezoid
// entries of block column k times a 2*(k*128)-flop inner product, plus (3d+2) per Gram entry.
double trsm_flops(int N, int M, int k, bool in_rows, int batch) {
  const double w = std::min(TS, N - k * TS);
  const double rows = (in_rows ? std::max(0, N - (k + 1) * TS) : 0) + (M + 1);
  return batch * rows * w * w;  // w^2/2 multiply-adds per row
}

// Fused schedule: k_panel = off-diagonal + extra rows of block column k (update + Gram + in-register
// trmm); k_diag = the diagonal tile (update + Gram + potf2 + inverse).
double panel_flops(int N, int M, int d, int k, bool in_rows, int batch) {
  const double w = std::min(TS, N - k * TS);
  const double rows = (in_rows ? std::max(0, N - (k + 1) * TS) : 0) + (M + 1);
  return batch * (rows * w * 2.0 * (double)(k * TS) + rows * w * w + (3.0 * d + 2.0) * (rows - 1) * w);
}
double diag_flops(int N, int d, int k, int batch) {
  const double w = std::min(TS, N - k * TS);
  const double tri = w * (w + 1) / 2.0;
  return batch * (tri * 2.0 * (double)(k * TS) + (3.0 * d + 2.0) * tri + w * w * w / 3.0 + w * w * w / 3.0);
}

// Offsets every per-fit pointer of `a` by g0 fits (group view of a batch).
template <typename T> FitArgs group_view(const FitArgs &a, int g0) {
  FitArgs v = a;
  auto adv = [](const void *p, size_t elems) -> const void * {
    return p ? static_cast<const void *>(static_cast<const T *>(p) + elems) : nullptr;
  };
  v.Lw = const_cast<void *>(adv(a.Lw, (size_t)g0 * a.lw_stride));
  v.Winv = const_cast<void *>(adv(a.Winv, (size_t)g0 * a.winv_stride));
  v.Lp = a.Lp ? static_cast<void *>(static_cast<unsigned short *>(a.Lp) + (size_t)g0 * a.lp_stride) : nullptr;
  v.dpart = const_cast<void *>(adv(a.dpart, (size_t)g0 * 2 * DPART));
  v.pimg = const_cast<void *>(adv(a.pimg, (size_t)g0 * 2 * DPART));
  v.X = adv(a.X, (size_t)g0 * a.d * a.N);
  v.Xs = adv(a.Xs, (size_t)g0 * a.d * a.M);
  v.y = adv(a.y, (size_t)g0 * a.N);
  v.mean = const_cast<void *>(adv(a.mean, (size_t)g0 * a.M));
  v.var = const_cast<void *>(adv(a.var, (size_t)g0 * a.M));
  v.alpha = const_cast<void *>(adv(a.alpha, (size_t)g0 * a.alpha_stride));
  v.theta = a.theta + (size_t)g0 * MAX_THETA;
  v.jitter = a.jitter ? a.jitter + g0 : nullptr;
  v.prep = a.prep ? a.prep + (size_t)g0 * PREP_N : nullptr;
  v.logml = a.logml ? a.logml + g0 : nullptr;
  v.info = a.info ? a.info + g0 : nullptr;
  v.rflag = a.rflag ? a.rflag + g0 : nullptr;
  return v;
}

// A/B switches of the schedule.  The shipped library has ONE schedule pair (throughput: k_diag_lean +
// k_panel with running predictive sums; latency: k_tile_sk + k_trmm_sk for <= LAT_FITS_F64 / LAT_FITS_F32 fits); the
// alternatives measured in DESIGN.md (one diagonal launch per step, two-stream overlap,
// fat diagonal, finalize without accumulators, fused trmm) exist only in a -DCGP_AB build, where the
// environment selects them once per process.
struct SchedSwitches {
  bool no_latency = false, split_diag = false, fused_diag = false, overlap = false, fat_diag = false, acc_off = false,
       sk_fused_trmm = false;
};
const SchedSwitches &sched_switches() {
  static const SchedSwitches sw = [] {
    SchedSwitches w;
    // CGP_SCHED=throughput is honoured by every build: the tests use it to run the throughput schedule
    // on a handful of fits.
    const char *e = getenv("CGP_SCHED");
    const std::string sch = e ? e : "";
    w.no_latency = sch == "throughput";
#ifdef CGP_AB
    w.no_latency = w.no_latency || sch == "overlap" || sch == "splitdiag";
    w.split_diag = sch == "splitdiag";
    w.fused_diag = sch == "fuseddiag";
    w.no_latency = w.no_latency || w.fused_diag;
    w.overlap = sch == "overlap";
    const char *dg = getenv("CGP_DIAG");
    w.fat_diag = dg && std::string(dg) == "fat";
    const char *ac = getenv("CGP_ACC");
    w.acc_off = ac && std::string(ac) == "off";
    const char *tk = getenv("CGP_SK_TRMM");
    w.sk_fused_trmm = tk && std::string(tk) == "fused";
#endif
    return w;
  }();
  return sw;
}

// k_panel<T, true> has two builds: the full-batch one, and the one for calls that leave CUs underfilled (kinds C /
// image-A compiled in; fp32: deep-prefetch loops)
template <typename T> void launch_panel_diag(bool mid, dim3 grid, hipStream_t s, const FitArgs &a, int k) {
  if (mid) {
    hipLaunchKernelGGL((k_panel<T, true, true, true>), grid, dim3(256), paneldiag_mid_lds_bytes<T>(), s, a, k);
    return;
  }
  if constexpr (CGP_F32_FULL_DEEP && sizeof(T) == 4) hipLaunchKernelGGL((k_panel<T, true, true, false>), grid, dim3(256), paneldiag_lds_bytes<T>(), s, a, k);
  else hipLaunchKernelGGL((k_panel<T, true>), grid, dim3(256), paneldiag_lds_bytes<T>(), s, a, k);
}

// The extra-row tiles of block step k alone (rows_from_extra), in the loop flavour of the mid-size build
template <typename T> void launch_panel_rows(dim3 grid, hipStream_t s, const FitArgs &a, int k) {
  hipLaunchKernelGGL((k_panel<T, false, true, false>), grid, dim3(256), panel_lds_bytes<T>(), s, a, k);
}

template <typename T> void launch_diag(const FitArgs &a, int nfits, int k, bool fat, hipStream_t s, bool mid = false) {
  if constexpr (mid_fat<T, true>()) {
    if (mid) {
      hipLaunchKernelGGL((k_diag_lean<T, true>), dim3(nfits), dim3(256), paneldiag_mid_lds_bytes<T>(), s, a, k);
      return;
    }
  }
#ifdef CGP_AB
  if (fat) {
    hipLaunchKernelGGL(k_diag<T>, dim3(nfits), dim3(256), potf2_lds_bytes<T>(), s, a, k);
    return;
  }
#endif
  (void)fat;
  hipLaunchKernelGGL(k_diag_lean<T>, dim3(nfits), dim3(256), paneldiag_lds_bytes<T>(), s, a, k);
}

// A mid-size fit call as ONE persistent launch (k_sched, cgp_kernels_fused.hpp): the tile programs of the NT + 1 launches
// become tasks of a list ordered by block step; per-fit progress counters replace the launch boundaries.  Bitwise the
// launches' results; measured slower than them (round 4), so it exists in the -DCGP_AB library only.
bool sched_enabled() {   // measurement builds only (-DCGP_AB), CGP_SCHED=onelaunch: as measured it does not beat the launches
  if constexpr (!kAbBuild) return false;
  static const bool on = [] {
    const char *e = getenv("CGP_SCHED");
    return e && std::string(e) == "onelaunch";
  }();
  return on;
}
#ifdef CGP_AB
template <typename T>
int run_sched(cgp_ctx *c, FitArgs a, int batch, hipStream_t s) {
  const int NT = a.NT, ET = a.ET;
  if (c->sched_key[0] != batch || c->sched_key[1] != NT || c->sched_key[2] != ET) {
    std::vector<int4> tasks;
    auto put = [&](int kind, int k, int b, int rt) { tasks.push_back(make_int4(kind, k, b, rt)); };
    // Order: by block step; inside a step the chain first (A, then the pre-updates C and B), then the matrix rows' tiles (the
    // next step's chain reads them), and only then the EXTRA rows' tiles of the step BEFORE -- they feed nothing but their own
    // next block column, so they lag one step behind and fill in beside the chain.  CGP_SCHED_LAG=0: no lag (measurement).
    static const int lag = [] { const char *e = getenv("CGP_SCHED_LAG"); return e ? atoi(e) : 1; }();
    auto extra = [&](int k) {
      for (int e = 0; e < ET; ++e)
        for (int b = 0; b < batch; ++b) put(SCHED_T, k, b, NT + e);
    };
    for (int k = 0; k < NT; ++k) {
      if (k + 1 < NT) for (int b = 0; b < batch; ++b) put(SCHED_A, k, b, k + 1);
      if (k + 2 < NT) for (int b = 0; b < batch; ++b) put(SCHED_C, k, b, k + 2);
      if (k + 2 < NT && k >= 1) for (int b = 0; b < batch; ++b) put(SCHED_B, k, b, k + 2);
      for (int rt = k + 2; rt < NT; ++rt)
        for (int b = 0; b < batch; ++b) put(SCHED_T, k, b, rt);
      if (k - lag >= 0) extra(k - lag);
    }
    for (int k = std::max(0, NT - lag); k < NT; ++k) extra(k);
    if (!grow_device(c->dsched_tasks, c->sched_tasks_cap, tasks.size() * sizeof(int4))) return CGP_ENOMEM;
    HIP_TRY(c, hipMemcpyAsync(c->dsched_tasks, tasks.data(), tasks.size() * sizeof(int4), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));   // `tasks` is a local (once per shape)
    c->sched_ntasks = (int)tasks.size();
    c->sched_key[0] = batch;
    c->sched_key[1] = NT;
    c->sched_key[2] = ET;
    int occ = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_sched<T>, 256, paneldiag_mid_lds_bytes<T>()));
    c->sched_grid = std::max(1, std::min(c->sched_ntasks, std::max(1, occ) * std::max(1, c->n_cu)));
  }
  const int pstride = c->NTmax + c->ETmax, fstride = c->NTmax + 2;
  const size_t nflag = 2 + (size_t)batch * (pstride + 1 + 2 * fstride);
  const size_t img_bytes = (size_t)batch * c->NTmax * DPART * sizeof(T);
  if (!grow_device(c->dsched_flags, c->sched_flags_cap, nflag * sizeof(int)) || !grow_device(c->dsched_bimg, c->sched_bimg_cap, img_bytes) ||
      !grow_device(c->dsched_cimg, c->sched_cimg_cap, img_bytes))
    return CGP_ENOMEM;
  HIP_TRY(c, hipMemsetAsync(c->dsched_flags, 0, nflag * sizeof(int), s));
  int *fl = static_cast<int *>(c->dsched_flags);
  SchedArgs q{};
  q.tasks = static_cast<const int4 *>(c->dsched_tasks);
  q.ntasks = c->sched_ntasks;
  q.head = fl;
  q.abort = fl + 1;
  q.prog = fl + 2;
  q.prog_stride = pstride;
  q.wdone = q.prog + (size_t)batch * pstride;
  q.bflag = q.wdone + batch;
  q.cflag = q.bflag + (size_t)batch * fstride;
  q.flag_stride = fstride;
  a.dpart = c->dsched_bimg;
  a.pimg = c->dsched_cimg;
  a.img_slots = c->NTmax;
  a.diag_slots = 0;
  a.diag_stride = 0;
  hipLaunchKernelGGL(k_sched<T>, dim3(c->sched_grid), dim3(256), paneldiag_mid_lds_bytes<T>(), s, a, q);
  return CGP_OK;
}
#else
template <typename T> int run_sched(cgp_ctx *, FitArgs, int, hipStream_t) { return CGP_EINVAL; }
#endif

// Refinement of an fp32 fit (cgp_refine.hpp): how many steps this call takes, and whether every fit takes them or only the
// ones k_finalize marks as dense (FitArgs::rflag, RF_RHO).  -1 (the default) = one step; every fit of a window of at most three
// input dimensions (measured, 40 fits of N = 1000, mean error against the oracle without / with: d = 1 1.0e-3 / 6e-7, d = 2
// 7e-4 / 1.5e-7, d = 3 3.1e-4 with a worst fit at 1.1e-3 / 1e-7), the marked fits beyond (d = 4: two windows of 43 000 fuzz cases
// at 1.1 and 1.3e-3, rho = 21; d = 6: 7e-5, worst 1.8e-4, rho 3 ... 8): BASELINE configs[2] (d = 6, rho 3 ... 11.5) pays one
// launch whose workgroups return at once (k_refine_gated).
constexpr int kRefineAutoMaxD = 3;
inline int refine_steps(const cgp_ctx *c, const FitArgs &a) {
  if (c->dtype != CGP_F32 || !c->dref_a || a.xid || a.NT > kRefineMaxNT) return 0;
  // one step contracts the mean's error by ~1e-3 on most windows (N = 1536 / 2048 / 3000, d = 1, a lone fit: 2.4e-3 / 8.8e-4 / 2.4e-3 ->
  // 1.8e-6 / 6.0e-6 / 1.9e-6) but by as little as 2e-2 on some -- 6.7e-3 -> 1.4e-4 -> 2.6e-6 at N = 4 500; in batched calls of N = 1 100
  // the sweep found single fits at 3 ... 7e-5 after one step (six in ~3 000 refined cases) -- so windows of more than 1 024 samples
  // take two (the contract for a refined mean is 5e-5: include/corenav_gp.h)
  if (c->refine < 0) return a.NT > 8 ? 2 : 1;
  return std::min(c->refine, 3);
}
inline bool refine_gated(const cgp_ctx *c, const FitArgs &a) { return c->refine < 0 && a.d > kRefineAutoMaxD; }
template <bool MEAN> void launch_refine_gemv(int rows, int nfits, hipStream_t s, const FitArgs &a, const RefineArgs &q) {
  const dim3 grid(cdiv(rows, RF_ROWS), nfits);
  if (a.kernel_id == K_RBF_BROWNIAN) hipLaunchKernelGGL((k_refine_gemv<float, 1, true, MEAN>), grid, dim3(256), 0, s, a, q);
  else if (a.d == 1) hipLaunchKernelGGL((k_refine_gemv<float, 1, false, MEAN>), grid, dim3(256), 0, s, a, q);
  else if (a.d == 2) hipLaunchKernelGGL((k_refine_gemv<float, 2, false, MEAN>), grid, dim3(256), 0, s, a, q);
  else if (a.d == 3) hipLaunchKernelGGL((k_refine_gemv<float, 3, false, MEAN>), grid, dim3(256), 0, s, a, q);
  else hipLaunchKernelGGL((k_refine_gemv<float, 0, false, MEAN>), grid, dim3(256), 0, s, a, q);
}
// The launches that follow k_finalize of `nfits` fits starting at fit g0 of the call: alpha_0 = L^-T z in double (in place of
// k_alpha), `steps` correction steps (solve = false: alpha of the last fit call is already in dref_a -- predict after fit),
// then the mean of the M test points.
inline void launch_refine(cgp_ctx *c, const FitArgs &a, int nfits, int g0, hipStream_t s, int steps, bool solve, bool alpha_for_all) {
  const RefineArgs q{c->dref_r + (size_t)g0 * c->alpha_stride, c->dref_a + (size_t)g0 * c->alpha_stride, c->alpha_stride, a.rflag};
  if (solve && q.flag && !alpha_for_all && a.NT <= kRefineGatedMaxNT) {
    // d > 3 under the default setting: k_finalize marked the dense fits, almost always none -- one launch, not four
    hipLaunchKernelGGL(k_refine_gated<float>, dim3(nfits), dim3(RS_THREADS), refine_gated_lds_bytes(a.NT), s, a, q, steps, a.M > 0 ? 1 : 0);
    return;
  }
  if (solve) {
    const size_t lds = refine_solve_lds_bytes(a.NT);
    RefineArgs q0 = q;
    if (alpha_for_all) q0.flag = nullptr;   // the caller asked for alpha (cgp_get_alpha): alpha_0 of every fit, marked or not
    hipLaunchKernelGGL(k_refine_solve<float>, dim3(nfits), dim3(RS_THREADS), lds, s, a, q0, 0);
    for (int st = 0; st < steps; ++st) {
      launch_refine_gemv<false>(a.N, nfits, s, a, q);
      hipLaunchKernelGGL(k_refine_solve<float>, dim3(nfits), dim3(RS_THREADS), lds, s, a, q, 1);
    }
  }
  if (a.M > 0) launch_refine_gemv<true>(a.M, nfits, s, a, q);
}
inline double refine_flops(const FitArgs &a, int nfits, int steps, bool solve) {   // covariance entries at (3 d + 20) flops + the two passes over the factor
  const double ent = (solve ? (double)steps * a.N * a.N : 0.0) + (double)a.M * a.N;
  return nfits * (ent * (3.0 * a.d + 20.0) + (solve ? (steps + 0.5) * 2.0 * a.N * a.N : 0.0));
}

// Enqueue the whole schedule for `batch` fits.  The batch is cut into up to c->nstreams contiguous
// groups, one worker stream each, forked from / joined to the caller's stream `s` with events; the
// launches are issued step-interleaved so every stream always has work queued.
// in_rows = false: predict after fit (only the extra row tiles).
template <typename T>
int run_schedule(cgp_ctx *c, FitArgs a, int batch, bool in_rows, bool want_alpha, hipStream_t s) {
  const SchedSwitches &sw = sched_switches();
  a.rows_from_extra = in_rows ? 0 : 1;
  // fp32: alpha and the mean refined against a double-precision residual (cgp_refine.hpp).  A fit call computes alpha for it;
  // a predict-after-fit call (in_rows = false) reuses the alpha the fit call left in dref_a.
  int rsteps = 0;
  if constexpr (sizeof(T) == 4) rsteps = refine_steps(c, a);
  const bool rsolve = rsteps > 0 && (in_rows || want_alpha);
  if (rsteps > 0 && !rsolve && !c->refined) rsteps = 0;
  if (in_rows || want_alpha) c->refined = rsolve;
  const bool ralpha_all = rsolve && want_alpha;
  if (rsolve) want_alpha = false;   // alpha_0 comes from k_refine_solve (double precision, dref_a): no k_alpha launch
  if constexpr (sizeof(T) == 4) a.rflag = rsteps > 0 && refine_gated(c, a) ? c->dref_flag : nullptr;
  const int upd_lds = upd_lds_bytes<T>();
  const int tile_lds = potf2_lds_bytes<T>();
  const bool auto_groups = c->nstreams == 0;
  int G = auto_groups ? 1 : std::max(1, std::min(std::min(c->nstreams, (int)cgp_ctx::kMaxStreams), batch));
  // latency schedule: a handful of fits, windows long enough for splitting to pay and short enough for the
  // diagonal tile's pre-update images (N <= 2560); anything else takes the throughput schedule
  // (the slab / image capacities are part of the predicate: a call they cannot hold -- only reachable with the measurement
  // build's CGP_LAT_FITS override -- takes the mid-size or throughput schedule instead of failing after k_prep was queued)
  const bool latency = !sw.no_latency && batch <= std::min(lat_fits<T>(a.NT), c->lat_cap) && a.NT >= CGP_LAT_MIN_NT && lat_images(a.NT - 1) <= LAT_IMG_MAX &&
                       (size_t)batch * (a.NT + a.ET + 1) <= c->lat_units && (a.NT < 3 || batch <= c->lat_img_fits);
  const bool mid = batch <= std::min(mid_fits<T>(a.NT), c->mid_cap);  // the whole call (the images are indexed by fit)
  // A latency-schedule call is one chain; an fp64 mid-size call has its own concurrency (factorisation || extra rows, below).
  // An fp32 mid-size call honours cgp_set_streams: cut into groups on worker streams, the chain-bound early launches of
  // one group run beside the others' (64 x N=1024: 0.99 -> 0.92 ms per call at two groups) -- as long as every group stays
  // above the latency schedule's range (a group that small would switch schedule: four groups of 16 measured 1.50 ms).
  // Two groups, from 56 fits (measured, ms per call at 1 / 2 / 3 groups: 48 fits 0.77 / 0.83 / 0.83, 64 fits 0.99 / 0.92 / 1.48,
  // 96 fits 1.24 / 1.17 / 1.66).
  // Since round 5 the engine takes that split by itself (cgp_set_streams(0), the default): every caller of a 56 ... 96-fit fp32
  // call -- cgp_sweep_*, a plain C host -- gets the 0.90 ms, not only a caller that knew to ask; cgp_set_streams(1) keeps one group.
  if (latency || (mid && sizeof(T) == 8)) G = 1;
  cgp_ctx::GroupTune *tune = nullptr;   // non-null: this call belongs to the tuning runs
  hipEvent_t tune_ev0 = nullptr, tune_ev1 = nullptr;   // events to record on the caller's stream before / after this call
  if (mid && sizeof(T) == 4 && (G > 1 || auto_groups)) {
    G = batch >= 56 && batch / 2 > lat_fits<T>(a.NT) ? 2 : 1;
    // Whether the two groups really run side by side depends on how the runtime mapped the caller's stream and the worker
    // stream onto its few hardware queues -- the process's history, which nothing reports: the same 64-fit call took 0.89 ms
    // on the legacy stream and 1.28 ms (the two groups in order: worse than one group's 0.98) on a fresh torch stream or on a
    // context created later in the process (tools/sweep_probe.py, stream_probe.py; a probe with spin kernels did not predict
    // it, both groups on the context's own streams was no better, and a single timed call does not show it: it is a
    // steady-state effect of calls issued back to back).  So the default does not assume: per caller stream, after two
    // warm-up calls, four eligible calls run as two groups and four as one, each run bracketed by events on the caller's
    // stream, and from then on the faster form is used.  Results are bitwise the same either way.
    // Every kRetune calls the two runs are repeated (a caller that first synchronises after every call and later issues calls
    // back to back sees the other behaviour: sweep_probe.py's 1.24 ms): 8 calls in 264, half of them at the slower setting.
    // The measurement never blocks the caller and never touches a stream that is being captured: the decision is read with
    // hipEventQuery (until the last timed call has finished the call keeps two groups and asks again next time), an entry is
    // keyed by (stream, fits, block steps) -- a caller that mixes shapes on one stream gets one decision per shape --, a pair
    // of spans that differ by more than 4 x is the caller's own host gaps, not the schedules (discarded, measured again), and
    // while the stream is capturing (cgp_set_streams documents fixing the form for graphs; this keeps the default safe) the
    // call takes the form already decided, or two groups, without recording or reading any timing event.
    constexpr int kWarm = 2, kRun = 4, kTuneDecide = kWarm + 2 * kRun, kRetune = 256;
    if (G == 2 && auto_groups && in_rows && !c->prof) {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      const bool capturing = hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
      if (capturing) (void)hipGetLastError();
      cgp_ctx::GroupTune *t = nullptr, *lru = &c->tune[0];
      for (auto &e : c->tune) {
        if (e.state >= 0 && e.stream == s && e.batch == batch && e.NT == a.NT) t = &e;
        if (e.used < lru->used) lru = &e;
      }
      if (!t && !capturing) {
        t = lru;
        t->stream = s;
        t->batch = batch;
        t->NT = a.NT;
        t->state = 0;
        t->groups = 2;
        for (auto &e : t->e)
          if (!e && hipEventCreate(&e) != hipSuccess) {   // no events: keep the two groups, never measure
            (void)hipGetLastError();
            t->state = kTuneDecide + 1;
          }
        for (auto &e : t->mon)
          if (!e && hipEventCreate(&e) != hipSuccess) (void)hipGetLastError();   // (no monitor then)
        t->mon_pending = false;
        t->slow = 0;
        t->retune = 24;
      }
      if (capturing) {
        G = t && t->state > kTuneDecide ? t->groups : 2;
        t = nullptr;
      }
      if (t) {
        t->used = ++c->tune_clock;
        // (the first decisions are re-checked early: a context's first calls run while the part is still coming out of idle -- both forms
      // then measure alike, 4.70 / 4.72 ms per four calls where warm they are 4.5 / 2.9 -- and the decision is a coin toss)
      if (t->state > kTuneDecide + t->retune) {
        t->state = kWarm;
        t->retune = std::min(2 * t->retune, kRetune);
      }
        if (t->state == kTuneDecide) {
          float m2 = 0.f, m1 = 0.f;
          const hipError_t qe = hipEventQuery(t->e[3]);
          if (qe == hipSuccess && hipEventElapsedTime(&m2, t->e[0], t->e[1]) == hipSuccess &&
              hipEventElapsedTime(&m1, t->e[2], t->e[3]) == hipSuccess) {
            t->ms[0] = m2;
            t->ms[1] = m1;
            if (m2 > 4.f * m1 || m1 > 4.f * m2) t->state = kWarm - 1;   // host gaps, not schedules: measure again
            else t->groups = m2 <= m1 ? 2 : 1;
            ++t->state;
            static const bool trace = getenv("CGP_TUNE_TRACE") != nullptr;   // development aid: what was measured and decided
            if (trace) fprintf(stderr, "cgp tune: ctx %p stream %p fits %d NT %d: two groups %.3f ms, one group %.3f ms per %d calls -> %s\n", (void *)c, (void *)s,
                               batch, a.NT, m2, m1, kRun, t->state == kWarm ? "measure again" : (t->groups == 2 ? "two groups" : "one group"));
          } else {
            (void)hipGetLastError();   // not ready (or a failed query): nothing sticks, two groups for this call, ask again
            if (qe != hipErrorNotReady) t->state = kTuneDecide + 1;
          }
        }
        if (t->state == kTuneDecide) {
          G = 2;
        } else if (t->state > kTuneDecide) {
          G = t->groups;
          ++t->state;
          // The stream -> hardware-queue mapping the decision rests on can change under a long-lived caller (other contexts and
          // streams come and go): one call in a while is bracketed by two events that a LATER call reads without blocking; three
          // such calls in a row at more than 1.3 x what the chosen form measured send the entry back to measuring.
          if (t->mon[0] && t->mon[1]) {
            if (t->mon_pending) {
              const hipError_t qm = hipEventQuery(t->mon[1]);
              if (qm == hipSuccess) {
                float ms1 = 0.f;
                const float base = t->ms[t->groups == 2 ? 0 : 1] / (float)kRun;
                if (hipEventElapsedTime(&ms1, t->mon[0], t->mon[1]) == hipSuccess && base > 0.f) t->slow = ms1 > 1.3f * base ? t->slow + 1 : 0;
                t->mon_pending = false;
                if (t->slow >= 3) {
                  t->slow = 0;
                  t->state = kWarm;
                  static const bool trace2 = getenv("CGP_TUNE_TRACE") != nullptr;
                  if (trace2) fprintf(stderr, "cgp tune: ctx %p stream %p: three monitored calls above 1.3 x %.3f ms (last %.3f): measuring again\n", (void *)c, (void *)s, base, ms1);
                }
              } else (void)hipGetLastError();
            } else if ((t->state & 7) == 0) {
              tune_ev0 = t->mon[0];
              tune_ev1 = t->mon[1];
              t->mon_pending = true;
            }
          }
        } else {
          G = t->state < kWarm + kRun ? 2 : 1;
          tune = t;
          if (t->state == kWarm) tune_ev0 = t->e[0];
          if (t->state == kWarm + kRun - 1) tune_ev1 = t->e[1];
          if (t->state == kWarm + kRun) tune_ev0 = t->e[2];
          if (t->state == kTuneDecide - 1) tune_ev1 = t->e[3];
        }
      }
    }
  }
  if (c->prof || !in_rows) G = 1;  // per-kernel timing wants isolated launches
  if (tune_ev0) HIP_TRY(c, hipEventRecord(tune_ev0, s));
  const bool fused64 = sw.fused_diag || batch < FUSED64_BELOW;
  const bool split_diag = sw.split_diag || !in_rows || (sizeof(T) == 8 && !fused64);
  // Mid-size calls: the extra rows (test points and y: M + 1 of them against N - 128 (k + 1) matrix rows, i.e. most of
  // the update work) do not feed the factorisation, only their own next block column.  They get launches of their own on
  // a second stream -- E(k), after the launch that finished diagonal tile k -- so the chain-bound factorisation launches
  // A(k) (few workgroups, long chains) run beside the MFMA-bound extra-row launches instead of in lock-step with them:
  //     s :  diag(0)  A(0)  A(1)  A(2) ...            A(k): matrix tiles of step k + kinds A / B / C
  //     sE:           E(0)  E(1)  E(2) ...            E(k) waits for A(k - 1) (W_k, row panel k) and follows E(k - 1)
  // Measured (tools/r3_xsplit.sh, same box, variant without the split): fp64 N = 2048 48 fits 5.58 -> 4.99 ms, 32 fits 4.03 ->
  // 3.92, 24 fits 3.48 -> 3.51, 12 fits 2.62 -> 3.45 (the call is one chain then: nothing to run beside it); fp32 N = 1024
  // 64 fits 0.99 -> 1.01, 32 fits 0.69 -> 0.76: its launches last as long as kind A's chain at every k, so the extra rows'
  // last launches only queue up behind it, and two contexts overlap worse (0.76 -> 0.96 ms per call).  fp64 from 28 fits.
  // (not under cgp_profile_enable: per-launch events of two concurrent streams would overlap in time and their sum overstate the kernel)
  // fp32: the extra rows go to the second stream as 64-row tiles (k_rows64) from XSPLIT32_FITS fits per call
  int x32 = XSPLIT32_FITS;
  if constexpr (kAbBuild) {   // measurement: CGP_XSPLIT32=<min fits> moves the threshold (0: never)
    static const int x32env = [] { const char *e = getenv("CGP_XSPLIT32"); return e ? atoi(e) : -1; }();
    if (x32env >= 0) x32 = x32env;
  }
  const bool xsplit_on = sizeof(T) == 8 ? batch * a.NT >= XSPLIT64_WORK : (x32 > 0 && batch >= x32);
  const bool xsplit = mid && !latency && xsplit_on && in_rows && !split_diag && G == 1 && a.ET > 0 && !kNoExtraSplit && !c->prof;
  constexpr bool own_main = false;   // (both halves of a concurrent call on the context's own streams: measured worse, docs/negatives.md round 5)
  std::vector<FitArgs> ga(G);
  std::vector<int> gb(G);
  std::vector<hipStream_t> gs(G);
  for (int g = 0, g0 = 0; g < G; ++g) {
    gb[g] = batch / G + (g < batch % G ? 1 : 0);
    if constexpr (kAbBuild) {   // measurement: CGP_GROUP0 = fits of group 0 of two (uneven halves run out of step with each other)
      const char *e = getenv("CGP_GROUP0");
      if (e && G == 2 && atoi(e) > 0 && atoi(e) < batch) gb[g] = g == 0 ? atoi(e) : batch - atoi(e);
    }
    ga[g] = group_view<T>(a, g0);
    // group 0 stays on the caller's stream, the others go to worker streams: the workers live in another priority pool
    // (cgp_create), so they never share a hardware queue with the caller's stream -- two WORKER streams may share one
    // (the pool has few queues and every context creates eight streams: a 64-fit call as two groups on two workers ran
    // 1.37 instead of 0.92 ms whenever other contexts existed in the process)
    gs[g] = g == 0 ? s : c->wstream[g - 1];
    g0 += gb[g];
  }
  hipLaunchKernelGGL(k_prep, dim3(cdiv(batch, 64)), dim3(64), 0, s, a, batch, c->dprep, in_rows ? 1 : 0,
                     (latency && in_rows) ? c->dwready : nullptr);
  if (G > 1 || own_main) {
    HIP_TRY(c, hipEventRecord(c->ev_fork, s));
    for (int g = 0; g < G; ++g)
      if (gs[g] != s) HIP_TRY(c, hipStreamWaitEvent(gs[g], c->ev_fork, 0));
  }
  std::vector<Launcher> L;
  for (int g = 0; g < G; ++g) L.push_back(Launcher{c, gs[g]});
  // throughput schedule: the predictive sums V z and |V|^2 accumulate inside k_panel (block column
  // k - 1 while it streams through the row fragments of step k); k_finalize then only adds the last
  // block column instead of reading all of V.
  // (No memset of the sums: block step 1 -- the first that has a finished block column behind it -- WRITES them, later steps add;
  // the two fill launches were 10 us at the head of every call.  A window of one block step has nothing to accumulate.)
  const bool use_acc = !latency && !sw.acc_off && a.M > 0 && !a.xid && a.NT >= 2;
  if (use_acc) {
    const size_t half = (size_t)batch * a.M;
    for (int g = 0, g0 = 0; g < G; g0 += gb[g], ++g) {
      ga[g].macc = c->dmacc + (size_t)g0 * a.M;
      ga[g].vacc = c->dmacc + half + (size_t)g0 * a.M;
    }
  }
  const int panel_lds = panel_lds_bytes<T>();
  if (mid && !latency && in_rows && !c->prof && a.NT >= 2 && sched_enabled() && !sw.split_diag && !sw.overlap) {
    launch_diag<T>(ga[0], batch, 0, sw.fat_diag, s, true);   // diagonal tile 0: nothing to run beside it yet
    int rc = run_sched<T>(c, ga[0], batch, s);
    if (rc != CGP_OK) return rc;
    hipLaunchKernelGGL((k_finalize<T, 64>), dim3(cdiv(a.M, 64) + 1, batch), dim3(256), 0, s, ga[0], 1);
    if (want_alpha) hipLaunchKernelGGL(k_alpha<T>, dim3(batch), dim3(256), alpha_lds_bytes(a.NT), s, ga[0]);
    if (rsteps > 0) launch_refine(c, ga[0], batch, 0, s, rsteps, rsolve, ralpha_all);
    HIP_TRY(c, hipGetLastError());
    return CGP_OK;
  }
#ifdef CGP_AB
  if (sw.overlap && in_rows && G == 1 && !c->prof && a.NT >= 2) {
    // Look-ahead schedule on two streams: the panel launch of step k is cut into P1 = the tile right
    // below the diagonal (the only one the next diagonal tile needs) and P2 = all the others, and
    //     sA:  P1(k) -> diag(k+1)            sB:  P2(k)
    // run concurrently; events carry the cross dependencies.  Measured: no gain.
    // sA = the context's HIGH-priority stream (the chain P1 -> diag must be dispatched ahead of the bulk P2 queued on the
    // caller's stream; round 1-3 ran both on equal-priority worker streams, which may even have shared a hardware queue)
    hipStream_t sA = c->hstream, sB = s;
    auto ev = [&](int i) { return c->ev_look[i]; };
    HIP_TRY(c, hipEventRecord(c->ev_fork, s));
    HIP_TRY(c, hipStreamWaitEvent(sA, c->ev_fork, 0));
    FitArgs a1 = ga[0], a2 = ga[0];
    a1.tile_off = 0;
    a2.tile_off = 1;
    const int B = gb[0], NT = a.NT;
    launch_diag<T>(ga[0], B, 0, sw.fat_diag, sA);
    HIP_TRY(c, hipEventRecord(ev(0), sA));                         // evD[0]
    for (int k = 0; k < NT; ++k) {
      const int nin = NT - k - 1;                                  // in-matrix tiles below the diagonal
      const bool has_p1 = nin >= 1;
      const int n2 = (has_p1 ? nin - 1 : 0) + a.ET;                // P2: everything but the first tile
      if (has_p1) {
        if (k > 0) HIP_TRY(c, hipStreamWaitEvent(sA, ev(3 * (k - 1) + 2), 0));  // P2(k-1)
        hipLaunchKernelGGL((k_panel<T, false>), dim3(1, B), dim3(256), panel_lds, sA, a1, k);
        HIP_TRY(c, hipEventRecord(ev(3 * k + 1), sA));             // evP1[k]
      }
      HIP_TRY(c, hipStreamWaitEvent(sB, ev(3 * k), 0));            // diag(k)
      if (k > 0 && NT - k >= 1) HIP_TRY(c, hipStreamWaitEvent(sB, ev(3 * (k - 1) + 1), 0));  // P1(k-1)
      FitArgs ap = has_p1 ? a2 : a1;
      hipLaunchKernelGGL((k_panel<T, false>), dim3(n2, B), dim3(256), panel_lds, sB, ap, k);
      HIP_TRY(c, hipEventRecord(ev(3 * k + 2), sB));               // evP2[k]
      if (k + 1 < NT) {
        if (!has_p1 && k > 0) HIP_TRY(c, hipStreamWaitEvent(sA, ev(3 * (k - 1) + 2), 0));
        launch_diag<T>(ga[0], B, k + 1, sw.fat_diag, sA);
        HIP_TRY(c, hipEventRecord(ev(3 * (k + 1)), sA));           // evD[k+1]
      }
    }
    HIP_TRY(c, hipStreamWaitEvent(sA, ev(3 * (NT - 1) + 2), 0));
    hipLaunchKernelGGL((k_finalize<T, 64>), dim3(cdiv(a.M, 64) + 1, B), dim3(256), 0, sA, ga[0], 1);
    if (want_alpha) hipLaunchKernelGGL(k_alpha<T>, dim3(B), dim3(256), (a.NT * TS + TS) * sizeof(double), sA, ga[0]);
    if (rsteps > 0) launch_refine(c, ga[0], B, 0, sA, rsteps, rsolve, ralpha_all);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev_join[0], sA));
    HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join[0], 0));
    return CGP_OK;
  }
#endif
  // Latency schedule for a handful of fits (k_tile_sk): the diagonal tile and
  // the panel tiles of a step in one launch, inner dimension split over up to SK_MAX workgroups.
  if (latency) {
    const bool split_trmm = !sw.sk_fused_trmm;
    // tile slots per fit of THIS call (slab and tickets are indexed with it; the tickets are zero between launches whatever the stride)
    const int call_slots = a.NT + a.ET + 1;
    if ((size_t)batch * call_slots > c->lat_units || (a.NT >= 3 && batch > c->lat_img_fits)) return CGP_ECAPACITY;
    SplitArgs q{c->dpart, c->dticket, call_slots, 1, in_rows ? 1 : 0, c->dwready, c->dlatimg, split_trmm ? 0 : 1};
    for (int k = 0; k < a.NT; ++k) {
      const int tiles = (in_rows ? a.NT - k - 1 : 0) + a.ET;
      const int nslots = tiles + (in_rows ? 1 : 0);
      // >= 8 chunks of 16 columns per range; a function of k only, so a fit's result does not
      // depend on how many fits share the call (the partial sums are added in range order)
      const int sk = std::max(1, std::min(SK_MAX, k));
      q.sk = sk;
      L[0].begin(0, panel_flops(a.N, a.M, a.d, k, in_rows, batch) + (in_rows ? diag_flops(a.N, a.d, k, batch) : 0.0), k);
      // z also carries the pre-update workgroups of the NEXT diagonal tile (1 + images of tile k + 1)
      const int gz = in_rows ? std::max(sk, 1 + (k + 1 < a.NT ? lat_images(k + 1) : 0)) : sk;
      hipLaunchKernelGGL(k_tile_sk<T>, dim3(nslots, batch, gz), dim3(256), in_rows ? tile_lds : upd_lds, s, ga[0], q, k);
      L[0].end();
      if (split_trmm) {
        L[0].begin(2, trsm_flops(a.N, a.M, k, in_rows, batch));
        hipLaunchKernelGGL(k_trmm_sk<T>, dim3(tiles * TRMM_SPLIT, batch), dim3(64 * TRMM_WAVES), upd_lds, s, ga[0], k);
        L[0].end();
      }
    }
    L[0].begin(3, batch * (4.0 * a.M * a.N + 2.0 * a.N));
    hipLaunchKernelGGL((k_finalize<T, 8>), dim3(cdiv(a.M, 8) + 1, batch), dim3(256), 0, s, ga[0], in_rows ? 1 : 0);
    L[0].end();
    if (want_alpha) {
      L[0].begin(4, batch * (double)a.N * a.N);
      hipLaunchKernelGGL(k_alpha<T>, dim3(batch), dim3(256), alpha_lds_bytes(a.NT), s, ga[0]);
      L[0].end();
    }
    if (rsteps > 0) {
      L[0].begin(4, refine_flops(a, batch, rsteps, rsolve));
      launch_refine(c, ga[0], batch, 0, s, rsteps, rsolve, ralpha_all);
      L[0].end();
    }
    HIP_TRY(c, hipGetLastError());
    return CGP_OK;
  }
  // Fit launches: the diagonal tile k + 1 is finished inside the panel launch of step k and tile k + 2 is
  // pre-updated there (k_panel<T, true>, diag_next), so only tile 0 has a launch of its own: NT + 1
  // launches per fit schedule instead of 2 NT, and no launch in which one workgroup per fit factors a tile
  // while the rest of the chip waits.  -DCGP_AB, CGP_SCHED=splitdiag: one k_diag_lean launch per step.
  // Measured (same box, alternating libraries): fp32 N = 1024 +2.6 % fits/s, fp64 N = 2048 -0.7 % -- in fp64
  // the diagonal launches are already MFMA-bound in their syrk part and two workgroups per CU leave the
  // finisher's factorisation chain nothing to hide behind -- so at the bench batch fp64 keeps one k_diag_lean
  // launch per step (CGP_SCHED=fuseddiag in a -DCGP_AB build selects the fused form there too).
  // fp64: a diagonal launch is one workgroup per fit on a latency chain whose length does not depend on the
  // batch (3.3 ms per N = 2048 schedule), so below FUSED64_BELOW fits per call the fused form wins there too
  // (batch 32 +26 %, 64 +15 %, 128 +6 %, 256 +0.2 %, 512 -0.7 %).
  // (fused64 / split_diag / xsplit are decided above, with the stream groups)
  hipStream_t sE = c->wstream[0];
  Launcher LE{c, sE};
  for (int k = 0; k < a.NT; ++k) {
    const int gx_t = (in_rows ? a.NT - k - 1 : 0) + (xsplit ? 0 : a.ET);
    for (int g = 0; g < G; ++g) {
      if (in_rows && (split_diag || k == 0)) {
        L[g].begin(1, diag_flops(a.N, a.d, k, gb[g]));
        launch_diag<T>(ga[g], gb[g], k, sw.fat_diag, gs[g], mid && !split_diag);
        L[g].end();
      }
      if (xsplit) {
        // E(k) needs W_k and the row panel k: diag(0) for k = 0, else everything up to A(k - 1), the last thing queued on s
        HIP_TRY(c, hipEventRecord(c->ev_look[k], gs[g]));
        HIP_TRY(c, hipStreamWaitEvent(sE, c->ev_look[k], 0));
        FitArgs ae = ga[g];
        ae.rows_from_extra = 1;
        LE.begin(0, panel_flops(a.N, a.M, a.d, k, false, gb[g]), k);
        if constexpr (sizeof(T) == 8) launch_panel_rows<T>(dim3(a.ET, gb[g]), sE, ae, k);
        else hipLaunchKernelGGL(k_rows64<T>, dim3(cdiv(a.M + 1, HR), gb[g]), dim3(256), panel_lds_bytes<T>(), sE, ae, k);
        LE.end();
      }
      if (split_diag) {
        L[g].begin(0, panel_flops(a.N, a.M, a.d, k, in_rows, gb[g]), k);
        hipLaunchKernelGGL((k_panel<T, false>), dim3(gx_t, gb[g]), dim3(256), panel_lds, gs[g], ga[g], k);
        L[g].end();
        continue;
      }
      const bool hasA = k + 1 < a.NT, hasB = k + 2 < a.NT && k >= 1;
      const bool hasC = mid && k + 2 < a.NT, imgA = mid && hasA && k >= 1;  // launch k - 1 had a kind C iff k + 1 < NT
      FitArgs ak = ga[g];
      ak.diag_slots = (hasA ? 1 : 0) | (hasB ? 2 : 0) | (hasC ? 4 : 0) | (imgA ? 8 : 0);
      ak.diag_stride = sizeof(T) == 8 || mid ? 2 : (CGP_F32_FULL_DEEP ? 3 : F32_FULL_OCC);  // workgroups per CU of k_panel<T, true> (LDS / VGPR bound)
      const int gx = gx_t + (hasB ? 1 : 0) + (hasC ? 1 : 0);  // gx_t already counts row tile k + 1 (kind A)
      // algorithmic flops of THIS launch: kind C does the part of tile (k + 2, k + 1) that kind A of launch k + 1 no longer does
      double fl = panel_flops(a.N, a.M, a.d, k, true, gb[g]) + (hasA ? diag_flops(a.N, a.d, k + 1, gb[g]) : 0.0);
      if (xsplit) fl -= panel_flops(a.N, a.M, a.d, k, false, gb[g]);
      auto tile_part = [&](int kc, int ncols) {  // Gram + `ncols` inner columns of one 128-row tile of block column kc, all fits
        const double w = std::min(TS, a.N - kc * TS), rows = std::min(TS, a.N - (kc + 1) * TS);
        return gb[g] * rows * w * (2.0 * ncols + (3.0 * a.d + 2.0));
      };
      if (imgA) fl -= tile_part(k, (k - 1) * TS);
      if (hasC) fl += tile_part(k + 1, k * TS);
      if (gx < 1) continue;  // (mid-size call, last block step: the extra rows are all that is left)
      L[g].begin(0, fl, k);
      launch_panel_diag<T>(mid, dim3(gx, gb[g]), gs[g], ak, k);
      L[g].end();
    }
  }
  if (xsplit) {  // join: k_finalize reads the extra rows
    HIP_TRY(c, hipEventRecord(c->ev_join[0], sE));
    HIP_TRY(c, hipStreamWaitEvent(gs[0], c->ev_join[0], 0));
  }
  for (int g = 0, g0 = 0; g < G; g0 += gb[g], ++g) {
    L[g].begin(3, gb[g] * (4.0 * a.M * a.N + 2.0 * a.N));
    hipLaunchKernelGGL((k_finalize<T, 64>), dim3(cdiv(a.M, 64) + 1, gb[g]), dim3(256), 0, gs[g], ga[g], in_rows ? 1 : 0);
    L[g].end();
    if (want_alpha) {
      L[g].begin(4, gb[g] * (double)a.N * a.N);
      hipLaunchKernelGGL(k_alpha<T>, dim3(gb[g]), dim3(256), alpha_lds_bytes(a.NT), gs[g], ga[g]);
      L[g].end();
    }
    if (rsteps > 0) {
      L[g].begin(4, refine_flops(a, gb[g], rsteps, rsolve));
      launch_refine(c, ga[g], gb[g], g0, gs[g], rsteps, rsolve, ralpha_all);
      L[g].end();
    }
  }
  HIP_TRY(c, hipGetLastError());
  for (int g = 0; g < G; ++g) {
    if (gs[g] == s) continue;   // ran on the caller's stream itself
    hipEvent_t ej = c->ev_join[(g + 1) % cgp_ctx::kMaxStreams];   // ([0] is the extra rows' join)
    HIP_TRY(c, hipEventRecord(ej, gs[g]));
    HIP_TRY(c, hipStreamWaitEvent(s, ej, 0));
  }
  if (tune_ev1) HIP_TRY(c, hipEventRecord(tune_ev1, s));
  if (tune) ++tune->state;
  return CGP_OK;
}

int run(cgp_ctx *c, const FitArgs &a, int batch, bool in_rows, bool want_alpha, hipStream_t s) {
  return c->dtype == CGP_F64 ? run_schedule<double>(c, a, batch, in_rows, want_alpha, s)
                             : run_schedule<float>(c, a, batch, in_rows, want_alpha, s);
}

FitArgs base_args(cgp_ctx *c, int N, int d, int M, int kid, int include_noise) {
  FitArgs a{};
  a.Lw = c->Lw;
  a.lw_stride = c->lw_stride;
  a.ld = c->ld;
  a.Winv = c->Winv;
  a.winv_stride = c->winv_stride;
  a.Lp = c->Lp;
  a.lp_stride = 3 * c->lw_stride;
  a.alpha = c->dalpha;
  a.alpha_stride = c->alpha_stride;
  a.prep = c->dprep;
  a.dpart = c->ddiagimg;
  a.pimg = c->dpanimg;
  a.dbgbuf = c->ddbg;
  a.N = N;
  a.d = d;
  a.M = M;
  a.NT = cdiv(N, TS);
  a.ET = cdiv(M + 1, TS);
  // the tile loops read up to four chunks (64 columns) past block column k - 1 of a panel without a branch (bx6_iter): every block
  // step k < NT of a call stays inside the NTmax x 128 columns of the fit's slab
  static_assert(4 * KT <= TS, "the unconditional look-ahead of the tile loops stays inside the next block column");
  a.kernel_id = kid;
  a.include_noise = include_noise;
#ifdef CGP_ABLATION
  static const int dbg_env = [] { const char *e = getenv("CGP_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg_env;  // timing ablations: results are wrong on purpose, cgp_build_flags() reports the build
#endif
  return a;
}

int check_shape(const cgp_ctx *c, int batch, int N, int d, int M, int kid) {
  if (!c) return CGP_EINVAL;
  if (batch < 1 || N < 1 || d < 1 || M < 0) return CGP_EINVAL;
  if (kid < 0 || kid > 2) return CGP_EINVAL;
  if (kid == CGP_KERNEL_RBF_BROWNIAN && d != 1) return CGP_EINVAL;
  if (batch > c->max_batch || N > c->max_n || M > c->max_m || d > c->max_d) return CGP_ECAPACITY;
  return CGP_OK;
}

// hip_stream argument of the device entry points: NULL is the legacy default stream itself (what
// torch.cuda.current_stream().cuda_stream is for the default stream), so the work is ordered with the
// caller's other default-stream work; CGP_STREAM_CTX selects the context's private stream.
inline hipStream_t pick_stream(cgp_ctx *c, void *hip_stream) {
  return hip_stream == CGP_STREAM_CTX ? c->stream : (hipStream_t)hip_stream;
}

// Pinned staging blocks that kernels also access IN PLACE (the one-launch short-window paths, the per-tick window push) must not be
// freed and allocated again between two such launches (see cgp_window_push: stores lost under load): a block starts at 64 KB -- more
// than any in-place use needs -- and grows geometrically, so the in-place paths never see a second allocation and the DMA-staged
// ones (cgp_fit_predict_batch's megabytes) see few.
bool grow_pinned(void *&p, size_t &cap, size_t bytes) {
  if (bytes <= cap) return true;
  bytes = std::max({bytes, (size_t)64 * 1024, 2 * cap});
  if (p) (void)hipHostFree(p);
  p = nullptr;
  cap = 0;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
    p = nullptr;
    return false;
  }
  cap = bytes;
  return true;
}
bool grow_device(void *&p, size_t &cap, size_t bytes) {
  if (bytes <= cap) return true;
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    p = nullptr;
    return false;
  }
  cap = bytes;
  return true;
}

// (n, d) row-major fp64  ->  SoA [d][n] in the device dtype, staged in `tmp`
void pack_soa(const double *src, int n, int d, int dtype, std::vector<char> &tmp, size_t off_elems) {
  if (dtype == CGP_F64) {
    double *o = reinterpret_cast<double *>(tmp.data()) + off_elems;
    for (int q = 0; q < d; ++q)
      for (int i = 0; i < n; ++i) o[(size_t)q * n + i] = src[(size_t)i * d + q];
  } else {
    float *o = reinterpret_cast<float *>(tmp.data()) + off_elems;
    for (int q = 0; q < d; ++q)
      for (int i = 0; i < n; ++i) o[(size_t)q * n + i] = (float)src[(size_t)i * d + q];
  }
}
void pack_vec(const double *src, size_t n, int dtype, std::vector<char> &tmp, size_t off_elems) {
  if (dtype == CGP_F64) memcpy(reinterpret_cast<double *>(tmp.data()) + off_elems, src, n * sizeof(double));
  else {
    float *o = reinterpret_cast<float *>(tmp.data()) + off_elems;
    for (size_t i = 0; i < n; ++i) o[i] = (float)src[i];
  }
}
void unpack_vec(const std::vector<char> &tmp, size_t off_elems, size_t n, int dtype, double *dst) {
  if (dtype == CGP_F64) memcpy(dst, reinterpret_cast<const double *>(tmp.data()) + off_elems, n * sizeof(double));
  else {
    const float *o = reinterpret_cast<const float *>(tmp.data()) + off_elems;
    for (size_t i = 0; i < n; ++i) dst[i] = (double)o[i];
  }
}

// mean of diag(Ky) for GPy's jitter policy (jitchol: jitter = diagA.mean() * 1e-6)
double mean_diag(int kid, const double *theta, int d, const double *X, int N) {
  const double noise = theta[ntheta(kid, d) - 1] + 1e-8;
  if (kid != CGP_KERNEL_RBF_BROWNIAN) return theta[0] + noise;
  double s = 0;
  for (int i = 0; i < N; ++i) s += std::fabs(X[i]);
  return theta[0] * theta[2] * s / N + noise;
}

int upload_theta(cgp_ctx *c, const double *theta, int theta_stride, int nth, int batch, hipStream_t s,
                 std::vector<double> &stage) {
  stage.assign((size_t)batch * CGP_MAX_THETA, 0.0);
  for (int b = 0; b < batch; ++b)
    for (int q = 0; q < nth; ++q) stage[(size_t)b * CGP_MAX_THETA + q] = theta[(size_t)b * theta_stride + q];
  HIP_TRY(c, hipMemcpyAsync(c->dtheta, stage.data(), stage.size() * sizeof(double), hipMemcpyHostToDevice, s));
  return CGP_OK;
}

}  // namespace

extern "C" {

int cgp_abi_version(void) { return CGP_ABI_VERSION; }

const char *cgp_strerror(int code) {
  switch (code) {
    case CGP_OK: return "ok";
    case CGP_EINVAL: return "invalid argument";
    case CGP_ENOMEM: return "out of device memory";
    case CGP_EHIP: return "HIP runtime error (see cgp_last_error)";
    case CGP_ESTATE: return "no fitted model in this context";
    case CGP_ENODEVICE: return "no usable gfx950 device";
    case CGP_ECAPACITY: return "problem exceeds the context capacity given to cgp_create";
    default: return code > 0 ? "matrix not positive definite after the jitter policy (value = failing pivot)" : "unknown error";
  }
}

const char *cgp_last_error(const cgp_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }
double cgp_last_jitter(const cgp_ctx *ctx) { return ctx ? ctx->fjitter : 0.0; }

cgp_ctx *cgp_create(int device, int max_n, int max_m, int max_d, int max_batch, int dtype) {
  return cgp_create_ex(device, max_n, max_m, max_d, max_batch, dtype, nullptr);
}

cgp_ctx *cgp_create_ex(int device, int max_n, int max_m, int max_d, int max_batch, int dtype, int *status) {
  auto fail = [status](int code) -> cgp_ctx * {
    if (status) *status = code;
    return nullptr;
  };
  if (status) *status = CGP_OK;
  if (max_n < 1 || max_m < 0 || max_d < 1 || max_d > CGP_MAX_D || max_batch < 1) return fail(CGP_EINVAL);
  if (dtype != CGP_F64 && dtype != CGP_F32) return fail(CGP_EINVAL);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(CGP_ENODEVICE);
  if (hipSetDevice(device) != hipSuccess) return fail(CGP_ENODEVICE);
  // the code object holds gfx950 kernels only (MFMA f64 16x16x4, 64-bit DPP row_newbcast, 16-byte LDS-DMA)
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(CGP_ENODEVICE);
  if ((dtype == CGP_F64 ? set_lds_attrs<double>(device) : set_lds_attrs<float>(device)) != 0) return fail(CGP_EHIP);
  (void)hipGetLastError();   // a clean slate: what the allocations below leave behind tells out-of-memory from anything else
  cgp_ctx *c = new cgp_ctx();
  c->n_cu = prop.multiProcessorCount;
  c->device = device;
  c->dtype = dtype;
  c->esz = dtype == CGP_F64 ? 8 : 4;
  c->max_n = max_n;
  c->max_m = max_m;
  c->max_d = max_d;
  c->max_batch = max_batch;
  c->NTmax = cdiv(max_n, TS);
  c->ETmax = cdiv(max_m + 1, TS);
  c->ld = (c->NTmax + c->ETmax) * TS;
  c->lw_stride = (size_t)c->ld * c->NTmax * TS;
  c->winv_stride = (size_t)c->NTmax * WIMG;
  c->alpha_stride = (size_t)c->NTmax * TS;
  const size_t B = max_batch;
  bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
  // Worker streams carry launches that must run BESIDE the caller's stream (the extra-row launches E(k) of a mid-size
  // call, the groups of cgp_set_streams).  The runtime multiplexes HIP streams onto a few hardware queues per priority
  // level, and two streams that land on one queue run in order: a 64-fit fp64 call took 8.34 instead of 6.87 ms whenever
  // its worker stream shared the legacy default stream's queue -- which depended on how many streams the process had used
  // before (tools/ctx_placement.py, profiles/r04_ctx_placement.txt: one throw-away stream created in between restored
  // 6.89 ms; memory placement played no part).  Queues are pooled per priority, so the workers are created at the LOWEST
  // priority: they never share a queue with a normal-priority caller stream, and the chain-bound factorisation launches
  // on the caller's stream are dispatched ahead of the extra rows that fill in beside them.
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  for (int i = 0; i < cgp_ctx::kMaxStreams; ++i) {
    ok = ok && hipStreamCreateWithPriority(&c->wstream[i], hipStreamNonBlocking, prio_least) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming) == hipSuccess;
  }
  ok = ok && hipStreamCreateWithPriority(&c->hstream, hipStreamNonBlocking, prio_greatest) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
  for (auto &e : c->ev_look) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipMalloc(&c->Lw, B * c->lw_stride * c->esz) == hipSuccess;
  ok = ok && hipMalloc(&c->Winv, B * c->winv_stride * c->esz) == hipSuccess;
  ok = ok && hipMalloc(&c->dX, B * max_d * max_n * c->esz) == hipSuccess;
  ok = ok && hipMalloc(&c->dXs, B * max_d * (size_t)std::max(max_m, 1) * c->esz) == hipSuccess;
  ok = ok && hipMalloc(&c->dy, B * max_n * c->esz) == hipSuccess;
  ok = ok && hipMalloc(&c->dmean, B * (size_t)std::max(max_m, 1) * c->esz) == hipSuccess;
  ok = ok && hipMalloc(&c->dvar, B * (size_t)std::max(max_m, 1) * c->esz) == hipSuccess;
  ok = ok && hipMalloc(&c->dalpha, B * c->alpha_stride * c->esz) == hipSuccess;
  if (dtype == CGP_F32) {
    ok = ok && hipMalloc((void **)&c->dref_r, B * c->alpha_stride * sizeof(double)) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->dref_a, B * c->alpha_stride * sizeof(double)) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->dref_flag, B * sizeof(int)) == hipSuccess;
    ok = ok && hipMemset(c->dref_flag, 0, B * sizeof(int)) == hipSuccess;
  }
  ok = ok && hipMalloc((void **)&c->dtheta, B * CGP_MAX_THETA * sizeof(double)) == hipSuccess;
  ok = ok && hipMalloc((void **)&c->djitter, B * sizeof(double)) == hipSuccess;
  ok = ok && hipMalloc((void **)&c->dgpart, (B * c->NTmax * (c->NTmax + 1) / 2 * GRAD_N + 2) * sizeof(double)) == hipSuccess;  // + logML, info of a single evaluation
  c->sk_slots = c->NTmax + c->ETmax + 1;
  c->lat_cap = std::min(LAT_FITS_ALLOC, max_batch);
  // partial-tile slab: the latency schedule takes MORE fits the fewer block steps a window has (lat_fits_by_steps), and a
  // call of NT block steps uses NT + ET + 1 tile slots per fit, so the slab is sized for the largest fits x slots product over
  // the block-step counts this context can see (a max_n = 2048 fp64 context: 11 fits x 22 slots, not 32 x 22) and a call
  // indexes it with its OWN slot count (SplitArgs::slots)
  size_t lat_units = 0;
  for (int nt = 1; nt <= c->NTmax; ++nt) {
    const int fits = std::min(c->lat_cap, kAbBuild ? LAT_FITS_ALLOC : lat_fits_by_steps(dtype == CGP_F64, nt));
    lat_units = std::max(lat_units, (size_t)fits * (nt + c->ETmax + 1));
  }
  c->lat_units = lat_units;
  ok = ok && hipMalloc(&c->dpart, lat_units * SK_MAX * TS * TS * c->esz) == hipSuccess;
  ok = ok && hipMalloc((void **)&c->dticket, sizeof(int) * c->lat_cap * c->sk_slots) == hipSuccess;
  ok = ok && hipMemset(c->dticket, 0, sizeof(int) * c->lat_cap * c->sk_slots) == hipSuccess;
  ok = ok && hipMalloc((void **)&c->dwready, sizeof(int) * c->lat_cap) == hipSuccess;
  // pre-update images exist from three block steps on (lat_images): sized for the most fits the latency schedule takes there
  int img_fits = 1;
  for (int nt = 3; nt <= std::max(3, c->NTmax); ++nt)
    img_fits = std::max(img_fits, std::min(c->lat_cap, kAbBuild ? LAT_FITS_ALLOC : lat_fits_by_steps(dtype == CGP_F64, nt)));
  c->lat_img_fits = c->NTmax >= 3 ? img_fits : 1;
  ok = ok && hipMalloc(&c->dlatimg, (size_t)c->lat_img_fits * 2 * LAT_IMG_MAX * DPART * c->esz) == hipSuccess;
  ok = ok && hipMalloc((void **)&c->dmacc, sizeof(double) * 2 * B * std::max(c->max_m, 1)) == hipSuccess;
  ok = ok && hipMalloc(&c->ddiagimg, B * 2 * DPART * c->esz) == hipSuccess;
  c->mid_cap = std::min(MID_FITS_ALLOC, max_batch);
  ok = ok && hipMalloc(&c->dpanimg, (size_t)c->mid_cap * 2 * DPART * c->esz) == hipSuccess;
  if (dtype == CGP_F32 && kMidPlanes) ok = ok && hipMalloc(&c->Lp, (size_t)c->mid_cap * 3 * c->lw_stride * sizeof(unsigned short)) == hipSuccess;
  ok = ok && hipMalloc((void **)&c->ddbg, DBG_SLOTS * sizeof(long long)) == hipSuccess;
  ok = ok && hipMemset(c->ddbg, 0, DBG_SLOTS * sizeof(long long)) == hipSuccess;
  ok = ok && hipMalloc((void **)&c->dprep, B * PREP_N * sizeof(double)) == hipSuccess;
  ok = ok && hipMalloc((void **)&c->dlogml, B * sizeof(double)) == hipSuccess;
  ok = ok && hipMalloc((void **)&c->dinfo, B * sizeof(int)) == hipSuccess;
  if (dtype == CGP_F64) {
    ok = ok && hipMalloc((void **)&c->dsmall, B * SM_OUT * sizeof(double)) == hipSuccess;
    {
      std::vector<unsigned short> tab;
      for (int NB = 1; NB <= SM_MAX_NB; ++NB) {
        c->smdeal_off[NB] = tab.size();
        tab.resize(tab.size() + small_table_elems(NB), 0);
        for (int jb = 0; jb < NB; ++jb) ok = sm_build_deal(tab.data() + c->smdeal_off[NB], NB, jb) && ok;   // false: the deal outgrew sm_eval's lists
        sm_build_rowmap(tab.data() + c->smdeal_off[NB] + (size_t)NB * SM_NH * SM_DEAL, NB);
      }
      ok = ok && hipMalloc((void **)&c->dsmdeal, tab.size() * sizeof(unsigned short)) == hipSuccess;
      ok = ok && hipMemcpy(c->dsmdeal, tab.data(), tab.size() * sizeof(unsigned short), hipMemcpyHostToDevice) == hipSuccess;
      ok = ok && hipMalloc((void **)&c->dsmdone, sizeof(int)) == hipSuccess && hipMemset(c->dsmdone, 0, sizeof(int)) == hipSuccess;
    }
    ok = ok && set_small_attr(device) == 0;
  }
  // the fills above (tickets, debug slots, refinement flags) are asynchronous to the host and ordered on the legacy default stream;
  // the context's work runs on non-blocking streams that do not wait for it (see cgp_window_init)
  ok = ok && hipDeviceSynchronize() == hipSuccess;
  if (!ok) {
    const hipError_t e = hipGetLastError();
    cgp_destroy(c);
    return fail(e == hipErrorOutOfMemory ? CGP_ENOMEM : CGP_EHIP);
  }
  return c;
}

void cgp_destroy(cgp_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto &r : c->recs) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  for (auto e : c->pool) (void)hipEventDestroy(e);
  void *bufs[] = {c->Lw, c->Winv, c->dX, c->dXs, c->dy, c->dmean, c->dvar, c->dalpha, c->dtheta, c->djitter, c->dlogml, c->dinfo, c->dprep, c->ddbg, c->dgpart, c->dpart, c->dticket, c->dwready, c->dlatimg, c->la_buf, c->la_ibuf, c->dmacc, c->ddiagimg, c->dpanimg, c->dsmall, c->dsmdeal, c->dsmdone, c->dref_r, c->dref_a, c->dref_flag, c->Lp};
  for (void *p : bufs)
    if (p) (void)hipFree(p);
  for (int i = 0; i < cgp_ctx::kMaxStreams; ++i) {
    if (c->wstream[i]) {
      (void)hipStreamSynchronize(c->wstream[i]);
      (void)hipStreamDestroy(c->wstream[i]);
    }
    if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
  }
  if (c->hstream) {
    (void)hipStreamSynchronize(c->hstream);
    (void)hipStreamDestroy(c->hstream);
  }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  for (auto &t : c->tune) {
    for (auto e : t.e)
      if (e) (void)hipEventDestroy(e);
    for (auto e : t.mon)
      if (e) (void)hipEventDestroy(e);
  }
  for (auto e : c->ev_look)
    if (e) (void)hipEventDestroy(e);
  for (void *wb : c->winbuf)
    if (wb) (void)hipFree(wb);
  if (c->win_pin) (void)hipHostFree(c->win_pin);
  if (c->opt_pin) (void)hipHostFree(c->opt_pin);
  if (c->win_dev) (void)hipFree(c->win_dev);
  if (c->pin_in) (void)hipHostFree(c->pin_in);
  if (c->pin_out) (void)hipHostFree(c->pin_out);
  if (c->draw) (void)hipFree(c->draw);
  for (void *sb : {c->dsched_tasks, c->dsched_flags, c->dsched_bimg, c->dsched_cimg})
    if (sb) (void)hipFree(sb);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int cgp_debug_read(cgp_ctx *c, long long out[CGP_DEBUG_SLOTS]) {
  if (!c || !out) return CGP_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipDeviceSynchronize());
  HIP_TRY(c, hipMemcpy(out, c->ddbg, DBG_SLOTS * sizeof(long long), hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemset(c->ddbg, 0, DBG_SLOTS * sizeof(long long)));  // the sums restart
  HIP_TRY(c, hipDeviceSynchronize());                                 // (the fill is asynchronous to the host: see cgp_window_init)
  return CGP_OK;
}

int cgp_debug_buffers(cgp_ctx *c, unsigned long long out[2 * CGP_DEBUG_BUFFERS]) {
  if (!c || !out) return CGP_EINVAL;
  const size_t B = c->max_batch;
  const void *p[CGP_DEBUG_BUFFERS] = {c->Lw, c->Winv, c->ddiagimg, c->dpanimg, c->dX, c->dmacc, c->dpart, c->dlatimg};
  const size_t n[CGP_DEBUG_BUFFERS] = {B * c->lw_stride * c->esz,
                                       B * c->winv_stride * c->esz,
                                       B * 2 * DPART * c->esz,
                                       (size_t)c->mid_cap * 2 * DPART * c->esz,
                                       B * c->max_d * c->max_n * c->esz,
                                       sizeof(double) * 2 * B * std::max(c->max_m, 1),
                                       c->lat_units * SK_MAX * TS * TS * c->esz,
                                       (size_t)c->lat_img_fits * 2 * LAT_IMG_MAX * DPART * c->esz};
  for (int i = 0; i < CGP_DEBUG_BUFFERS; ++i) {
    out[2 * i] = (unsigned long long)(uintptr_t)p[i];
    out[2 * i + 1] = n[i];
  }
  return CGP_OK;
}

int cgp_debug_small(cgp_ctx *c, double out[CGP_SMALL_OUT]) {
  if (!c || !out) return CGP_EINVAL;
  if (!c->dsmall) return CGP_ESTATE;
  HIP_TRY(c, hipSetDevice(c->device));
  if (c->last_small_dev) {
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(out, c->dsmall, SM_OUT * sizeof(double), hipMemcpyDeviceToHost));
  } else memcpy(out, c->last_small, SM_OUT * sizeof(double));
  return CGP_OK;
}

int cgp_build_flags(void) {
  int f = 0;
#ifdef CGP_ABLATION
  f |= CGP_BUILD_ABLATION;
#endif
#ifdef CGP_AB
  f |= CGP_BUILD_AB;
#endif
  if (!kF32Bf16x6) f |= CGP_BUILD_F32_NATIVE;
  return f;
}

int cgp_synchronize(cgp_ctx *c) {
  if (!c) return CGP_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return CGP_OK;
}

int cgp_set_streams(cgp_ctx *c, int n) {
  if (!c || n < 0 || n > cgp_ctx::kMaxStreams) return CGP_EINVAL;
  c->nstreams = n;
  return CGP_OK;
}

int cgp_set_refine(cgp_ctx *c, int steps) {
  if (!c || steps < -1 || steps > 3) return CGP_EINVAL;
  c->refine = steps;
  return CGP_OK;
}

int cgp_profile_enable(cgp_ctx *c, int on) {
  if (!c) return CGP_EINVAL;
  c->prof = on != 0;
  c->prof_step = on >= 2 ? on - 2 : -1;
  return CGP_OK;
}

int cgp_profile_read(cgp_ctx *c, double ms[CGP_PROF_KERNELS], double flops[CGP_PROF_KERNELS],
                     long long launches[CGP_PROF_KERNELS]) {
  if (!c) return CGP_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  for (auto &r : c->recs) {
    HIP_TRY(c, hipEventSynchronize(r.b));
    float t = 0;
    HIP_TRY(c, hipEventElapsedTime(&t, r.a, r.b));
    static const bool dump = getenv("CGP_PROF_DUMP") != nullptr;
    if (dump) fprintf(stderr, "cgp prof: kernel %d  %.4f ms\n", r.kernel, t);
    c->prof_ms[r.kernel] += t;
    c->prof_flops[r.kernel] += r.flops;
    c->prof_n[r.kernel] += 1;
    c->pool.push_back(r.a);
    c->pool.push_back(r.b);
  }
  c->recs.clear();
  for (int i = 0; i < CGP_PROF_KERNELS; ++i) {
    if (ms) ms[i] = c->prof_ms[i];
    if (flops) flops[i] = c->prof_flops[i];
    if (launches) launches[i] = c->prof_n[i];
    c->prof_ms[i] = c->prof_flops[i] = 0;
    c->prof_n[i] = 0;
  }
  return CGP_OK;
}

int cgp_fit_predict_batch_device(cgp_ctx *c, int batch, int N, int d, int M, int kid, const void *dX,
                                 const void *dy, const void *dXs, const double *dtheta, const double *djitter,
                                 int include_noise, void *dmean, void *dvar, double *dlogml, int *dinfo,
                                 void *hip_stream) {
  int rc = check_shape(c, batch, N, d, M, kid);
  if (rc != CGP_OK) return rc;
  if (!dX || !dy || !dtheta || !dlogml || !dinfo || (M > 0 && (!dXs || !dmean || !dvar))) return CGP_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  FitArgs a = base_args(c, N, d, M, kid, include_noise);
  a.X = dX;
  a.Xs = dXs;
  a.y = dy;
  a.theta = dtheta;
  a.jitter = djitter;
  a.mean = dmean;
  a.var = dvar;
  a.logml = dlogml;
  a.info = dinfo;
  c->have_fit = false;
  c->lazy_fit = false;
  if (small_batch_predict_ok(c, batch, N, d, M)) {   // short windows: fit + predictions of the whole batch in ONE launch, factors in LDS
    const SmallDev dev{static_cast<const double *>(dX), static_cast<const double *>(dy), static_cast<const double *>(dXs), dtheta, djitter, dlogml, dinfo};
    const int tab = c->pending_tab;   // only the host-buffer entry point knows whether the inputs are tick counts
    c->pending_tab = 0;
    return small_predict_launch(c, batch, N, d, M, kid, include_noise, static_cast<double *>(dmean), static_cast<double *>(dvar), c->dsmall,
                                pick_stream(c, hip_stream), nullptr, &dev, tab);
  }
  return run(c, a, batch, true, false, pick_stream(c, hip_stream));
}

int cgp_fit_predict_batch(cgp_ctx *c, int batch, int N, int d, int M, int kid, const double *X, const double *y,
                          const double *Xs, const double *theta, int theta_stride, int include_noise,
                          double *mean, double *var, double *logml, int *info) {
  int rc = check_shape(c, batch, N, d, M, kid);
  if (rc != CGP_OK) return rc;
  if (!X || !y || !theta || (M > 0 && (!Xs || !mean || !var))) return CGP_EINVAL;
  const int nth = ntheta(kid, d);
  if (theta_stride < nth) return CGP_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const size_t esz = c->esz;
  const bool trace = kAbBuild && getenv("CGP_TRACE_E2E");
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t_start = now(), t_last = t_start;
  auto lap = [&](const char *what) {
    if (!trace) return;
    auto t = now();
    fprintf(stderr, "[e2e] %-18s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };
  // Host side of the boundary (gp_slip_node.py:19-25 "list -> (n, 1) fp64"): the caller's row-major fp64
  // arrays go through ONE pinned staging block and ONE H2D DMA; the (n, d) -> SoA [d][n] transposition
  // and the fp64 -> device dtype conversion run on the device (k_pack_soa), not on a host core.
  const size_t B = batch, nX = B * N * d, ny = B * N, nXs = B * (size_t)M * d, nTh = B * CGP_MAX_THETA;
  const size_t in_bytes = (nX + ny + nXs + nTh) * sizeof(double);
  const size_t out_elems = 2 * B * (size_t)M;                    // mean, var in the device dtype
  const size_t out_bytes = out_elems * esz + B * sizeof(double) + B * sizeof(int);
  if (!grow_pinned(c->pin_in, c->pin_in_cap, in_bytes) || !grow_pinned(c->pin_out, c->pin_out_cap, out_bytes) ||
      !grow_device(c->draw, c->draw_cap, in_bytes))
    return CGP_ENOMEM;
  double *hin = static_cast<double *>(c->pin_in);
  double *hth = hin + nX + ny + nXs;
  for (size_t b = 0; b < B; ++b)
    for (int q = 0; q < CGP_MAX_THETA; ++q) hth[b * CGP_MAX_THETA + q] = q < nth ? theta[b * theta_stride + q] : 0.0;
  const double *draw = static_cast<const double *>(c->draw);
  // Staging copy (pageable -> pinned) is the longest host step of a large call (66 MB for the headline batch:
  // 6.6 ms on one core against 1.3 ms of DMA): ranges of fits are copied by up to 4 threads and each range's
  // DMA is queued as soon as it is staged, so the copy engine works while the later ranges are still copied.
  const int nthr = in_bytes > (4u << 20) ? (int)std::min<size_t>({4, B, std::max(1u, std::thread::hardware_concurrency())}) : 1;
  auto stage = [&](size_t b0, size_t b1) {
    memcpy(hin + b0 * N * d, X + b0 * N * d, (b1 - b0) * N * d * sizeof(double));
    memcpy(hin + nX + b0 * N, y + b0 * N, (b1 - b0) * N * sizeof(double));
    if (nXs) memcpy(hin + nX + ny + b0 * M * d, Xs + b0 * (size_t)M * d, (b1 - b0) * (size_t)M * d * sizeof(double));
  };
  auto dma = [&](size_t b0, size_t b1) -> hipError_t {
    char *dd = static_cast<char *>(c->draw);
    auto one = [&](size_t off, size_t cnt) {
      return cnt ? hipMemcpyAsync(dd + off * sizeof(double), hin + off, cnt * sizeof(double), hipMemcpyHostToDevice, s) : hipSuccess;
    };
    hipError_t e = one(b0 * N * d, (b1 - b0) * N * d);
    if (e == hipSuccess) e = one(nX + b0 * N, (b1 - b0) * N);
    if (e == hipSuccess) e = one(nX + ny + b0 * M * d, (b1 - b0) * (size_t)M * d);
    return e;
  };
  if (nthr == 1) {
    // a small call: the whole staged block [X | y | Xs | theta] in ONE DMA, unpacked by ONE launch
    stage(0, B);
    HIP_TRY(c, hipMemcpyAsync(c->draw, hin, in_bytes, hipMemcpyHostToDevice, s));
    lap("stage + queue DMA");
    const dim3 grid(std::min(64, cdiv(std::max(N, M) * d, 256)), batch);
    if (c->dtype == CGP_F64)
      hipLaunchKernelGGL(k_pack_call<double>, grid, dim3(256), 0, s, draw, static_cast<double *>(c->dX), static_cast<double *>(c->dy),
                         static_cast<double *>(c->dXs), c->dtheta, c->djitter, batch, N, d, M);
    else
      hipLaunchKernelGGL(k_pack_call<float>, grid, dim3(256), 0, s, draw, static_cast<float *>(c->dX), static_cast<float *>(c->dy),
                         static_cast<float *>(c->dXs), c->dtheta, c->djitter, batch, N, d, M);
  } else {
    std::vector<std::thread> th;
    auto lo = [&](int w) { return B * w / nthr; };
    for (int w = 1; w < nthr; ++w) th.emplace_back(stage, lo(w), lo(w + 1));
    stage(0, lo(1));
    hipError_t e = dma(0, lo(1));
    for (int w = 1; w < nthr; ++w) {
      th[w - 1].join();
      if (e == hipSuccess) e = dma(lo(w), lo(w + 1));
    }
    HIP_TRY(c, e);
    lap("stage + queue DMA");
    HIP_TRY(c, hipMemcpyAsync(static_cast<char *>(c->draw) + (nX + ny + nXs) * sizeof(double), hth, nTh * sizeof(double),
                              hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->dtheta, draw + nX + ny + nXs, nTh * sizeof(double), hipMemcpyDeviceToDevice, s));
    auto pack = [&](const double *src, void *dst, int n, int dd) {
      const dim3 grid(std::min(64, cdiv(n * dd, 256)), batch);
      if (c->dtype == CGP_F64) hipLaunchKernelGGL(k_pack_soa<double>, grid, dim3(256), 0, s, src, static_cast<double *>(dst), n, dd);
      else hipLaunchKernelGGL(k_pack_soa<float>, grid, dim3(256), 0, s, src, static_cast<float *>(dst), n, dd);
    };
    pack(draw, c->dX, N, d);
    pack(draw + nX, c->dy, N, 1);
    if (M > 0) pack(draw + nX + ny, c->dXs, M, d);
    HIP_TRY(c, hipMemsetAsync(c->djitter, 0, sizeof(double) * batch, s));
  }
  // (the scan that establishes "tick counts" is 1.5 ns per input on the host: 0.3 ms for 256 windows with their 599 test points,
  // more than the launch it would shorten by a tenth -- ensembles beyond 20 k inputs take the direct evaluation)
  const int tick_tab = small_batch_predict_ok(c, batch, N, d, M) && (size_t)batch * (N + M) <= 20000
                           ? tick_table_entries_batch(kid, d, batch, X, N, Xs, M) : 0;
  c->pending_tab = tick_tab;
  rc = cgp_fit_predict_batch_device(c, batch, N, d, M, kid, c->dX, c->dy, c->dXs, c->dtheta, c->djitter,
                                    include_noise, c->dmean, c->dvar, c->dlogml, c->dinfo, CGP_STREAM_CTX);
  c->pending_tab = 0;
  if (rc != CGP_OK) return rc;
  char *hout = static_cast<char *>(c->pin_out);
  double *hl = reinterpret_cast<double *>(hout + out_elems * esz);
  int *hinfo = reinterpret_cast<int *>(hl + B);
  lap("queue schedule");
  // results are requested together with the status: one synchronisation per call unless a fit needs the jitter ladder
  auto queue_results = [&]() -> hipError_t {
    hipError_t e = hipSuccess;
    if (M > 0) {
      e = hipMemcpyAsync(hout, c->dmean, B * M * esz, hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipMemcpyAsync(hout + B * M * esz, c->dvar, B * M * esz, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hl, c->dlogml, sizeof(double) * batch, hipMemcpyDeviceToHost, s);
    return e;
  };
  HIP_TRY(c, hipMemcpyAsync(hinfo, c->dinfo, sizeof(int) * batch, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, queue_results());
  HIP_TRY(c, hipStreamSynchronize(s));
  lap("device done (info + results)");
  bool retried = false;
  // GPy jitchol policy for the fits that failed: jitter = mean(diag) * 1e-6 * 10^k, k = 0..4,
  // re-submitted one fit at a time (rare path) into the same device slots.
  std::vector<double> hjit(batch, 0.0);
  for (int b = 0; b < batch; ++b) {
    if (hinfo[b] == 0) continue;
    double jit = mean_diag(kid, theta + (size_t)b * theta_stride, d, X + (size_t)b * N * d, N) * 1e-6;
    retried = true;
    for (int attempt = 0; attempt < 5 && hinfo[b] != 0; ++attempt, jit *= 10.0) {
      HIP_TRY(c, hipMemcpyAsync(c->djitter + b, &jit, sizeof(double), hipMemcpyHostToDevice, s));
      c->pending_tab = tick_tab;
      rc = cgp_fit_predict_batch_device(
          c, 1, N, d, M, kid, (char *)c->dX + (size_t)b * d * N * esz, (char *)c->dy + (size_t)b * N * esz,
          (char *)c->dXs + (size_t)b * d * M * esz, c->dtheta + (size_t)b * CGP_MAX_THETA, c->djitter + b,
          include_noise, (char *)c->dmean + (size_t)b * M * esz, (char *)c->dvar + (size_t)b * M * esz,
          c->dlogml + b, c->dinfo + b, CGP_STREAM_CTX);
      if (rc != CGP_OK) return rc;
      HIP_TRY(c, hipMemcpyAsync(&hinfo[b], c->dinfo + b, sizeof(int), hipMemcpyDeviceToHost, s));
      HIP_TRY(c, hipStreamSynchronize(s));
      hjit[b] = jit;
    }
  }
  c->fjitter = hjit[0];
  if (retried) {
    HIP_TRY(c, queue_results());
    HIP_TRY(c, hipStreamSynchronize(s));
  }
  lap("D2H");
  if (M > 0) {
    if (c->dtype == CGP_F64) {
      memcpy(mean, hout, B * M * sizeof(double));
      memcpy(var, hout + B * M * esz, B * M * sizeof(double));
    } else {
      const float *fm = reinterpret_cast<const float *>(hout), *fv = fm + B * M;
      for (size_t i = 0; i < B * M; ++i) {
        mean[i] = (double)fm[i];
        var[i] = (double)fv[i];
      }
    }
  }
  int first = 0;
  for (int b = 0; b < batch; ++b) {
    if (logml) logml[b] = hl[b];
    if (info) info[b] = hinfo[b];
    if (first == 0 && hinfo[b] != 0) first = hinfo[b];
  }
  lap("copy out");
  return first;
}

int cgp_fit(cgp_ctx *c, const double *X, const double *y, int N, int d, int kid, const double *theta, double *logml) {
  int rc = check_shape(c, 1, N, d, 0, kid);
  if (rc != CGP_OK) return rc;
  if (!X || !y || !theta) return CGP_EINVAL;
  const int nth = ntheta(kid, d);
  int info = 0;
  double l = 0;
  rc = cgp_fit_predict_batch(c, 1, N, d, 0, kid, X, y, nullptr, theta, nth, 0, nullptr, nullptr, &l, &info);
  if (rc < 0) return rc;
  c->have_fit = (info == 0);
  c->fN = N;
  c->fd = d;
  c->fkernel = kid;
  memcpy(c->ftheta, theta, sizeof(double) * nth);
  if (logml) *logml = l;
  return info;
}

int cgp_predict(cgp_ctx *c, const double *Xs, int M, int include_noise, double *mean, double *var) {
  if (!c || !Xs || !mean || !var || M < 1) return CGP_EINVAL;
  if (!c->have_fit) return CGP_ESTATE;
  if (M > c->max_m) return CGP_ECAPACITY;
  HIP_TRY(c, hipSetDevice(c->device));
  if (int fr = ensure_fitted(c)) return fr;
  hipStream_t s = c->stream;
  const size_t esz = c->esz;
  std::vector<char> hxs((size_t)c->fd * M * esz);
  pack_soa(Xs, M, c->fd, c->dtype, hxs, 0);
  HIP_TRY(c, hipMemcpyAsync(c->dXs, hxs.data(), hxs.size(), hipMemcpyHostToDevice, s));
  FitArgs a = base_args(c, c->fN, c->fd, M, c->fkernel, include_noise);
  a.X = c->dX;
  a.Xs = c->dXs;
  a.y = c->dy;
  a.theta = c->dtheta;
  a.jitter = c->djitter;
  a.mean = c->dmean;
  a.var = c->dvar;
  a.logml = c->dlogml;
  a.info = c->dinfo;
  int rc = run(c, a, 1, false, false, s);
  if (rc != CGP_OK) return rc;
  std::vector<char> hm((size_t)M * esz), hv((size_t)M * esz);
  HIP_TRY(c, hipMemcpyAsync(hm.data(), c->dmean, hm.size(), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipMemcpyAsync(hv.data(), c->dvar, hv.size(), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  unpack_vec(hm, 0, M, c->dtype, mean);
  unpack_vec(hv, 0, M, c->dtype, var);
  return CGP_OK;
}

int cgp_get_alpha(cgp_ctx *c, double *alpha) {
  if (!c || !alpha) return CGP_EINVAL;
  if (!c->have_fit) return CGP_ESTATE;
  HIP_TRY(c, hipSetDevice(c->device));
  if (int fr = ensure_fitted(c)) return fr;
  hipStream_t s = c->stream;
  // z is the y row of the factor panel; it sits at extra row index M of the LAST run.  Re-run the
  // y row alone (M = 0) so its position is known, then back-substitute.
  FitArgs a = base_args(c, c->fN, c->fd, 0, c->fkernel, 0);
  a.X = c->dX;
  a.Xs = c->dXs;
  a.y = c->dy;
  a.theta = c->dtheta;
  a.jitter = c->djitter;
  a.mean = c->dmean;
  a.var = c->dvar;
  a.logml = c->dlogml;
  a.info = c->dinfo;
  int rc = run(c, a, 1, false, true, s);
  if (rc != CGP_OK) return rc;
  if (c->dtype == CGP_F32 && c->refined) {   // the refined alpha is kept in double precision (cgp_refine.hpp)
    HIP_TRY(c, hipMemcpyAsync(alpha, c->dref_a, (size_t)c->fN * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return CGP_OK;
  }
  std::vector<char> h((size_t)c->fN * c->esz);
  HIP_TRY(c, hipMemcpyAsync(h.data(), c->dalpha, h.size(), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  unpack_vec(h, 0, c->fN, c->dtype, alpha);
  return CGP_OK;
}

int cgp_get_factor(cgp_ctx *c, double *L) {
  if (!c || !L) return CGP_EINVAL;
  if (!c->have_fit) return CGP_ESTATE;
  HIP_TRY(c, hipSetDevice(c->device));
  if (int fr = ensure_fitted(c)) return fr;
  const int N = c->fN;
  std::vector<char> h((size_t)N * N * c->esz);
  // columns 0..N-1, rows 0..N-1 of the column-major panel -> dense (N x N) column-major staging
  HIP_TRY(c, hipMemcpy2DAsync(h.data(), (size_t)N * c->esz, c->Lw, (size_t)c->ld * c->esz, (size_t)N * c->esz, N,
                              hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      double v = 0;
      if (j <= i) v = c->dtype == CGP_F64 ? reinterpret_cast<double *>(h.data())[(size_t)j * N + i]
                                          : (double)reinterpret_cast<float *>(h.data())[(size_t)j * N + i];
      L[(size_t)i * N + j] = v;
    }
  return CGP_OK;
}

int cgp_slip_node_callback(cgp_ctx *c, const double *time_array, const double *slip_array, int n, int kid,
                           const double *theta, double *mean, double *sigma, int cap, int *m_out) {
  if (!c || !time_array || !slip_array || !theta || !mean || !sigma || n < 2 || cap < 0) return CGP_EINVAL;
  // gp_slip_node.py:27-29  per = 0.9 ; x_train = X[:int(per*len(X))]
  const int ntr = (int)(0.9 * (double)n);
  if (ntr < 1) return CGP_EINVAL;
  // gp_slip_node.py:45  X_ = np.arange(X.min(), X.max() + 600, 1)
  double xmin = time_array[0], xmax = time_array[0];
  for (int i = 1; i < n; ++i) {
    xmin = std::min(xmin, time_array[i]);
    xmax = std::max(xmax, time_array[i]);
  }
  const long long glen = (long long)std::ceil((xmax + 600.0 - xmin) / 1.0);
  // gp_slip_node.py:59-61  means[len(X):]  (index slice)
  const long long mo = std::max(0LL, glen - n);
  if (m_out) *m_out = (int)mo;
  const int M = (int)std::min<long long>(mo, cap);
  if (M == 0) return CGP_OK;
  {   // a short fp64 window (the reference's 149-tick GP_Input is one): fit and predictions in one launch
    double th[CGP_MAX_THETA];
    std::copy(theta, theta + ntheta(kid, 1), th);
    const int rs = node_callback_small(c, time_array, slip_array, n, kid, th, 0, mean, sigma, cap, m_out);
    if (rs != CGP_ESTATE) return rs;
  }
  std::vector<double> xs(M), var(M);
  for (int m = 0; m < M; ++m) xs[m] = xmin + (double)(n + m);
  // GPRegression(...) and the m.predict loop (gp_slip_node.py:35,45-49) in ONE factorisation pass: the
  // prediction points ride as extra rows (one schedule, one host round trip instead of two)
  double logml = 0.0;
  int info = 0;
  int rc = cgp_fit_predict_batch(c, 1, ntr, 1, M, kid, time_array, slip_array, xs.data(), theta, ntheta(kid, 1), 1, mean,
                                 var.data(), &logml, &info);
  if (rc != CGP_OK) return rc;
  if (info != 0) return info;  // not positive definite even with GPy's jitter ladder (LinAlgError there)
  for (int m = 0; m < M; ++m) sigma[m] = 2.0 * std::sqrt(var[m]);  // gp_slip_node.py:61
  return CGP_OK;
}

int cgp_llh_to_enu(double lat, double lon, double h, const double init_llh[3], const double init_ecef[3], double enu[3]) {
  if (!init_llh || !init_ecef || !enu) return CGP_EINVAL;
  corenav::llh_to_enu(lat, lon, h, init_llh, init_ecef, enu);
  return CGP_OK;
}

int cgp_predict_stop(const double *mean, const double *sigma, int M, const double *P, const double *Q,
                     const double *STM, const double *Hvec, const double pos_llh[3], double arrival_time, double now,
                     double threshold, int h_bug_compatible, const double init_llh[3], const double init_ecef[3],
                     int *fired, double *stop_cmd, int *i_out, double *xy_err) {
  if (!mean || !sigma || M < 0 || !P || !Q || !STM || !Hvec || !pos_llh || !init_llh || !init_ecef) return CGP_EINVAL;
  corenav::StopPrediction r = corenav::predict_stop(mean, sigma, M, P, Q, STM, Hvec, pos_llh, arrival_time, now,
                                                    threshold, h_bug_compatible != 0, init_llh, init_ecef);
  if (fired) *fired = r.fired ? 1 : 0;
  if (stop_cmd) *stop_cmd = r.stop_cmd;
  if (i_out) *i_out = r.i;
  if (xy_err) *xy_err = r.xy_err;
  return CGP_OK;
}

int cgp_gppredictor_callback(const double *mean, const double *sigma, int M, const double *P, const double *Q,
                             const double *STM, const double *Hvec, const double pos_llh[3], double arrival_time,
                             double now, int h_bug_compatible, int *published, double *stop_cmd) {
  if (!mean || !sigma || M < 0 || !P || !Q || !STM || !Hvec || !pos_llh) return CGP_EINVAL;
  corenav::NodeHandle nh;
  int clock_reads = 0, npub = 0;
  double last = 0.0;
  nh.now = [&]() { return clock_reads++ == 0 ? arrival_time : now; };
  nh.call_set_stopping = [&](corenav_pod::core_nav::SetStopping &srv) {
    std::copy(P, P + 225, srv.response.PvecData.begin());
    std::copy(Q, Q + 225, srv.response.QvecData.begin());
    std::copy(STM, STM + 225, srv.response.STMvecData.begin());
    std::copy(Hvec, Hvec + 60, srv.response.HvecData.begin());
    srv.response.PosData.x = pos_llh[0];
    srv.response.PosData.y = pos_llh[1];
    srv.response.PosData.z = pos_llh[2];
    return srv.request.stopping;
  };
  nh.publish_stop_cmd = [&](const corenav_pod::std_msgs::Float64 &m) {
    ++npub;
    last = m.data;
  };
  GpPredictor node(nh);
  node.h_bug_compatible = h_bug_compatible != 0;
  auto msg = std::make_shared<corenav_pod::core_nav::GP_Output>();
  msg->mean.assign(mean, mean + M);
  msg->sigma.assign(sigma, sigma + M);
  node.GPCallBack(msg);
  if (published) *published = npub;
  if (stop_cmd) *stop_cmd = last;
  return CGP_OK;
}

}  // extern "C"

namespace {
// One gradient-mode evaluation on the device for the window already uploaded to slot 0.
template <typename T>
int grad_eval(cgp_ctx *c, int N, int d, int kid, double *logml, double sums[GRAD_N], int *info) {
  hipStream_t s = c->stream;
  FitArgs a = base_args(c, N, d, /*M=*/N, kid, 0);
  a.xid = 1;
  a.X = c->dX;
  a.Xs = c->dX;  // unused: the "test rows" are the identity
  a.y = c->dy;
  a.theta = c->dtheta;
  a.jitter = c->djitter;
  a.mean = c->dmean;
  a.var = c->dvar;
  // logML and info of this one window live right behind its partial sums: [sums | logml | info] comes back in ONE copy,
  // into the pinned block (a real asynchronous copy, no pageable staging)
  const int npairs = a.NT * (a.NT + 1) / 2;
  const size_t npart = (size_t)npairs * GRAD_N;
  a.gpart = c->dgpart;
  a.logml = c->dgpart + npart;
  a.info = reinterpret_cast<int *>(c->dgpart + npart + 1);
  int rc = run(c, a, 1, true, true, s);
  if (rc != CGP_OK) return rc;
  hipLaunchKernelGGL(k_grad<T>, dim3(npairs, 1), dim3(256), upd_lds_bytes<T>(), s, a, npairs);
  HIP_TRY(c, hipGetLastError());
  if (!grow_pinned(c->opt_pin, c->opt_pin_cap, (kOptPinIn + npart + 2) * sizeof(double))) return CGP_ENOMEM;
  double *part = static_cast<double *>(c->opt_pin) + kOptPinIn, *hl = part + npart;
  int *hi = reinterpret_cast<int *>(hl + 1);
  HIP_TRY(c, hipMemcpyAsync(part, c->dgpart, (npart + 2) * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  *logml = *hl;
  *info = *hi;
  for (int i = 0; i < GRAD_N; ++i) sums[i] = 0.0;
  for (int pz = 0; pz < npairs; ++pz)  // fixed order: deterministic
    for (int i = 0; i < GRAD_N; ++i) sums[i] += part[(size_t)pz * GRAD_N + i];
  return CGP_OK;
}

void grad_from_sums(int kid, int d, const double *theta, const double *sums, double *grad);   // below, with the batched optimiser

// The window of a gradient-mode evaluation -> slot 0 (once per cgp_nll_grad call, once per cgp_optimize run)
int upload_window(cgp_ctx *c, const double *X, const double *y, int N, int d, hipStream_t s) {
  std::vector<char> hx((size_t)d * N * c->esz), hy((size_t)N * c->esz);
  pack_soa(X, N, d, c->dtype, hx, 0);
  pack_vec(y, N, c->dtype, hy, 0);
  HIP_TRY(c, hipMemcpyAsync(c->dX, hx.data(), hx.size(), hipMemcpyHostToDevice, s));   // pageable source: staged by the
  HIP_TRY(c, hipMemcpyAsync(c->dy, hy.data(), hy.size(), hipMemcpyHostToDevice, s));   // runtime before the call returns
  return CGP_OK;
}

// -logML and its gradient at theta for the window in slot 0 (upload_window); X only feeds the jitter ladder's mean diagonal
int nll_grad_resident(cgp_ctx *c, const double *X, int N, int d, int kid, const double *theta, double *nll, double *grad) {
  const int nth = ntheta(kid, d);
  hipStream_t s = c->stream;
  // the whole block -- [theta | jitter | partial sums, logML, info] -- before any copy is queued: grad_eval must never
  // reallocate it under a copy in flight (hipHostFree would have to synchronise the device)
  const size_t nt_ = cdiv(N, TS), need = kOptPinIn + std::max<size_t>(64, nt_ * (nt_ + 1) / 2 * GRAD_N + 2);
  if (!grow_pinned(c->opt_pin, c->opt_pin_cap, need * sizeof(double))) return CGP_ENOMEM;
  double *hin = static_cast<double *>(c->opt_pin);   // [theta (CGP_MAX_THETA) | jitter]
  double jit = 0.0, logml = 0.0, sums[GRAD_N];
  int info = 0, rc = CGP_OK;
  for (int attempt = 0; attempt <= 5; ++attempt) {  // GPy jitchol policy
    hin = static_cast<double *>(c->opt_pin);
    for (int q = 0; q < CGP_MAX_THETA; ++q) hin[q] = q < nth ? theta[q] : 0.0;
    hin[CGP_MAX_THETA] = jit;
    HIP_TRY(c, hipMemcpyAsync(c->dtheta, hin, sizeof(double) * CGP_MAX_THETA, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->djitter, hin + CGP_MAX_THETA, sizeof(double), hipMemcpyHostToDevice, s));
    rc = c->dtype == CGP_F64 ? grad_eval<double>(c, N, d, kid, &logml, sums, &info)
                             : grad_eval<float>(c, N, d, kid, &logml, sums, &info);
    if (rc != CGP_OK) return rc;
    if (info == 0) break;
    jit = (attempt == 0) ? mean_diag(kid, theta, d, X, N) * 1e-6 : jit * 10.0;
  }
  c->fjitter = (info == 0) ? jit : 0.0;
  c->have_fit = (info == 0);
  c->fN = N;
  c->fd = d;
  c->fkernel = kid;
  memcpy(c->ftheta, theta, sizeof(double) * nth);
  if (info != 0) return info;
  *nll = -logml;
  grad_from_sums(kid, d, theta, sums, grad);   // dlogML/dtheta = 0.5 * sum_ij w_ij dK_ij/dtheta -> gradient of the NEGATIVE log likelihood
  return CGP_OK;
}
}  // namespace


// ---- short windows (N <= SM_MAX_N, fp64): one launch per evaluation / per whole optimisation (cgp_small.hpp) -------------
namespace {
bool small_enabled() {   // CGP_SMALL=off: the large-window machinery for every size (A/B, and the tests that cover it at small N)
  static const bool off = [] {
    const char *e = getenv("CGP_SMALL");
    return e && std::string(e) == "off";
  }();
  return !off;
}
inline bool small_ok(const cgp_ctx *c, int N) { return c->dtype == CGP_F64 && N <= SM_MAX_N && c->dsmall && small_enabled(); }

// Tick-grid table of the short-window kernels (SmallArgs::tab_n): entries needed when every input is integer-valued and small
// enough for r^2 to be exact, 0 otherwise.  ad hoc off-switch for A/B and the bitwise test: CGP_TICKTAB=off.
int tick_table_entries(int kid, int d, const double *X, int N, const double *Xs, int M) {
  static const bool off = [] {
    const char *e = getenv("CGP_TICKTAB");
    return e && std::string(e) == "off";
  }();
  if (off || kid != CGP_KERNEL_RBF_BROWNIAN || d != 1) return 0;
  double lo = X[0], hi = X[0];
  auto scan = [&](const double *v, int n) {   // branch-free body (vectorises): this runs in front of launches that take 50 us
    double l = lo, h = hi;
    int bad = 0;
    for (int i = 0; i < n; ++i) {
      const double x = v[i];
      const double c = x < -67108864.0 ? -67108864.0 : (x > 67108864.0 ? 67108864.0 : x);   // NaN falls through as itself
      bad |= (double)(int)c != x;
      l = x < l ? x : l;
      h = x > h ? x : h;
    }
    lo = l;
    hi = h;
    return bad == 0;
  };
  if (!scan(X, N) || (Xs && !scan(Xs, M))) return 0;
  const double spread = hi - lo;
  return spread < (double)SM_TAB_MAX ? (int)spread + 1 : 0;
}

// ... of a batch of windows (and their test points): the largest of the windows' tables, 0 as soon as one window is off the grid
int tick_table_entries_batch(int kid, int d, int batch, const double *X, int N, const double *Xs, int M) {
  int n = 0;
  for (int b = 0; b < batch; ++b) {
    const int nb = tick_table_entries(kid, d, X + (size_t)b * N * d, N, Xs ? Xs + (size_t)b * M * d : nullptr, M);
    if (nb == 0) return 0;
    n = std::max(n, nb);
  }
  return n;
}

double mean_abs_first(const double *X, int N, int d) {
  double s = 0;
  for (int i = 0; i < N; ++i) s += std::fabs(X[(size_t)i * d]);
  return s / N;
}
double mean_diag_from(int kid, const double *theta, int d, double meanabs) {   // mean_diag with the window's mean |x| precomputed
  const double noise = theta[ntheta(kid, d) - 1] + 1e-8;
  return kid == CGP_KERNEL_RBF_BROWNIAN ? theta[0] * theta[2] * meanabs + noise : theta[0] + noise;
}

// ONE window (X (N, d) row-major, y, theta) and optionally M test points staged in the pinned block [X | y | Xs | theta],
// which the short-window kernels read in place: no copy command and no unpack launch in front of them (8 KB over the
// host link inside a launch costs less than either)
struct SmallRaw {
  const double *X, *y, *Xs;
  double *theta;
};
int small_stage_window(cgp_ctx *c, const double *X, const double *y, int N, int d, const double *Xs, int M, const double *theta, int nth,
                       SmallRaw &r) {
  const size_t nX = (size_t)N * d, nXs = (size_t)M * d, in_bytes = (nX + N + nXs + CGP_MAX_THETA) * sizeof(double);
  if (!grow_pinned(c->pin_in, c->pin_in_cap, in_bytes)) return CGP_ENOMEM;
  double *hin = static_cast<double *>(c->pin_in);
  memcpy(hin, X, nX * sizeof(double));
  memcpy(hin + nX, y, (size_t)N * sizeof(double));
  if (Xs) memcpy(hin + nX + N, Xs, nXs * sizeof(double));
  for (int q = 0; q < CGP_MAX_THETA; ++q) hin[nX + N + nXs + q] = q < nth ? theta[q] : 0.0;
  r = SmallRaw{hin, hin + nX, hin + nX + N, hin + nX + N + nXs};
  return CGP_OK;
}

int small_launch(cgp_ctx *c, int batch, int N, int d, int kid, int mode, int max_evals, hipStream_t s, double *out = nullptr,
                 const SmallRaw *raw = nullptr, int tab_n = 0) {
  SmallArgs a{};
  a.tab_n = tab_n;
  a.X = raw ? raw->X : static_cast<const double *>(c->dX);
  a.y = raw ? raw->y : static_cast<const double *>(c->dy);
  a.theta = raw ? raw->theta : c->dtheta;
  a.x_sq = raw ? 1 : N;
  a.x_sr = raw ? d : 1;
  a.out = out ? out : c->dsmall;
  a.deal = c->dsmdeal + c->smdeal_off[cdiv(N, DB)];
  a.N = N;
  a.d = d;
  a.kernel_id = kid;
  a.nth = ntheta(kid, d);
  a.mode = mode;
  a.max_evals = max_evals > 0 ? max_evals : 1000;
  a.pgtol = 1e-5;   // scipy fmin_l_bfgs_b as paramz calls it: pgtol 1e-5, factr 1e7
  a.factr = 1e7;
  const size_t lds = ((small_lds_bytes(cdiv(N, DB), d) + 15) & ~(size_t)15) + (size_t)tab_n * sizeof(double);
  if (kid == CGP_KERNEL_RBF_BROWNIAN) hipLaunchKernelGGL((k_small<true, 1>), dim3(batch), dim3(SM_THREADS), lds, s, a);
  else if (d <= 1) hipLaunchKernelGGL((k_small<false, 1>), dim3(batch), dim3(SM_THREADS), lds, s, a);
  else if (d <= 3) hipLaunchKernelGGL((k_small<false, 3>), dim3(batch), dim3(SM_THREADS), lds, s, a);
  else hipLaunchKernelGGL((k_small<false, 8>), dim3(batch), dim3(SM_THREADS), lds, s, a);
  HIP_TRY(c, hipGetLastError());
  return CGP_OK;
}

// fit at the theta slots + prediction of M test points per window in ONE launch (k_small_predict); `parts` workgroups per window
inline bool small_predict_ok(const cgp_ctx *c, int N, int d, int M) {
  return small_ok(c, N) && M > 0 && small_predict_lds(cdiv(N, DB), d) <= kLdsPerWorkgroup;
}
bool small_batch_predict_ok(const cgp_ctx *c, int batch, int N, int d, int M) {
  static const bool off = [] {
    const char *e = kAbBuild ? getenv("CGP_SMALLPRED") : nullptr;
    return e && std::string(e) == "off";
  }();
  return !off && batch <= c->max_batch && small_predict_ok(c, N, d, M);
}
int small_predict_launch(cgp_ctx *c, int batch, int N, int d, int M, int kid, int include_noise, double *dmean, double *dvar, double *dout,
                         hipStream_t s, const SmallRaw *raw, const SmallDev *dev, int tab_n, int *done_flag, int done_seq) {
  SmallArgs a{};
  a.ladder = 1;
  a.done_flag = (done_flag && c->dsmdone) ? done_flag : nullptr;
  a.done_count = c->dsmdone;
  a.done_seq = done_seq;
  a.X = raw ? raw->X : static_cast<const double *>(c->dX);
  a.y = raw ? raw->y : static_cast<const double *>(c->dy);
  a.theta = raw ? raw->theta : c->dtheta;
  a.x_sq = raw ? 1 : N;
  a.x_sr = raw ? d : 1;
  a.xs_sq = raw ? 1 : M;
  a.xs_sr = raw ? d : 1;
  a.out = dout;
  a.deal = c->dsmdeal + c->smdeal_off[cdiv(N, DB)];
  a.N = N;
  a.d = d;
  a.kernel_id = kid;
  a.nth = ntheta(kid, d);
  a.Xs = raw ? raw->Xs : static_cast<const double *>(c->dXs);
  if (dev) {
    a.X = dev->X, a.y = dev->y, a.Xs = dev->Xs, a.theta = const_cast<double *>(dev->theta);
    a.jitter = dev->jitter, a.logml = dev->logml, a.info = dev->info, a.ladder = 0;
  }
  a.mean = dmean;
  a.var = dvar;
  a.M = M;
  a.include_noise = include_noise;
  const int nchunk = cdiv(M, DB);
  // every workgroup repeats its window's fit (one workgroup per CU: the factor fills the LDS), so: as many parts as keep the
  // launch to ONE round of workgroups -- a lone window: a chunk of 16 test points each; >= n_cu windows: one workgroup each
  a.parts = std::max(1, std::min(nchunk, std::max(c->n_cu, 1) / batch));
  a.parts = cdiv(nchunk, cdiv(nchunk, a.parts));   // no part without a chunk (8 windows x 38 chunks: 19 parts of two, not 32)
  size_t lds = small_predict_lds(cdiv(N, DB), d);
  if (lds + (size_t)tab_n * sizeof(double) > kLdsPerWorkgroup) tab_n = 0;   // no room beside the factor and the K* chunk: direct evaluation
  a.tab_n = tab_n;
  lds += (size_t)tab_n * sizeof(double);
  c->last_small_dev = dout == c->dsmall;
  const dim3 grid(batch * a.parts), block(SM_THREADS);
  if (kid == CGP_KERNEL_RBF_BROWNIAN) hipLaunchKernelGGL((k_small_predict<true, 1>), grid, block, lds, s, a);
  else if (d <= 1) hipLaunchKernelGGL((k_small_predict<false, 1>), grid, block, lds, s, a);
  else if (d <= 3) hipLaunchKernelGGL((k_small_predict<false, 3>), grid, block, lds, s, a);
  else hipLaunchKernelGGL((k_small_predict<false, 8>), grid, block, lds, s, a);
  HIP_TRY(c, hipGetLastError());
  return CGP_OK;
}

// the single-window record: the kernel writes it to the pinned block (behind its input part); one synchronisation
double *small_record_slot(cgp_ctx *c) {
  if (!grow_pinned(c->opt_pin, c->opt_pin_cap, (kOptPinIn + 64 + SM_OUT) * sizeof(double))) return nullptr;
  return static_cast<double *>(c->opt_pin) + kOptPinIn;
}
int small_read_one(cgp_ctx *c, const double *slot, double out[SM_OUT], hipStream_t s) {
  HIP_TRY(c, hipStreamSynchronize(s));
  memcpy(out, slot, SM_OUT * sizeof(double));
  memcpy(c->last_small, slot, SM_OUT * sizeof(double));
  c->last_small_dev = false;
  return CGP_OK;
}

void small_mark_fitted(cgp_ctx *c, const double *X, const double *y, int N, int d, int kid, const double *theta, double jitter, bool ok) {
  c->have_fit = ok;
  c->lazy_fit = ok;
  if (ok) {
    c->lazy_win.resize((size_t)N * d + N);
    memcpy(c->lazy_win.data(), X, (size_t)N * d * sizeof(double));
    memcpy(c->lazy_win.data() + (size_t)N * d, y, (size_t)N * sizeof(double));
  }
  c->f_meandiag_x = mean_abs_first(X, N, d);
  c->fjitter = ok ? jitter : 0.0;
  c->fN = N;
  c->fd = d;
  c->fkernel = kid;
  memcpy(c->ftheta, theta, sizeof(double) * ntheta(kid, d));
}

// cgp_predict / cgp_get_alpha / cgp_get_factor after a short-window evaluation: the window, theta (and the jitter the
// evaluation needed) are on the device, the factor panel is not -- run the fit schedule once, GPy's jitter ladder around it
int ensure_fitted(cgp_ctx *c) {
  if (!c->lazy_fit) return CGP_OK;
  // nothing counts as fitted until the schedule below has succeeded: an early return (upload, copy, launch error) must not
  // leave have_fit set over a factor panel that was never built for this window
  c->lazy_fit = false;
  c->have_fit = false;
  hipStream_t s = c->stream;
  {   // the window was evaluated where the caller's copy was staged: bring it (and theta) to slot 0 now
    int rc = upload_window(c, c->lazy_win.data(), c->lazy_win.data() + (size_t)c->fN * c->fd, c->fN, c->fd, s);
    if (rc != CGP_OK) return rc;
    double th[CGP_MAX_THETA] = {0};
    std::copy(c->ftheta, c->ftheta + ntheta(c->fkernel, c->fd), th);
    HIP_TRY(c, hipMemcpyAsync(c->dtheta, th, sizeof th, hipMemcpyHostToDevice, s));   // pageable source: staged before the call returns
  }
  FitArgs a = base_args(c, c->fN, c->fd, 0, c->fkernel, 0);
  a.X = c->dX;
  a.Xs = c->dXs;
  a.y = c->dy;
  a.theta = c->dtheta;
  a.jitter = c->djitter;
  a.mean = c->dmean;
  a.var = c->dvar;
  a.logml = c->dlogml;
  a.info = c->dinfo;
  double jit = c->fjitter;
  int info = 0;
  // GPy's jitchol: one attempt without jitter, then five rungs mean(diag) 1e-6 10^r.  The short-window kernel may already
  // have climbed to rung `rung` (its jitter is kept bit for bit); the refit continues from there and never passes rung 5.
  const double base = mean_diag_from(c->fkernel, c->ftheta, c->fd, c->f_meandiag_x) * 1e-6;
  int rung = jit > 0.0 && base > 0.0 ? 1 + (int)std::lround(std::log10(jit / base)) : 0;
  for (; rung <= 5; ++rung) {
    HIP_TRY(c, hipMemcpyAsync(c->djitter, &jit, sizeof(double), hipMemcpyHostToDevice, s));
    int rc = run(c, a, 1, true, false, s);
    if (rc != CGP_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(&info, c->dinfo, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (info == 0) break;
    jit = jit == 0.0 ? base : jit * 10.0;
  }
  c->have_fit = info == 0;
  c->fjitter = info == 0 ? jit : 0.0;
  return info;
}

int small_nll_grad(cgp_ctx *c, const double *X, const double *y, int N, int d, int kid, const double *theta, double *nll, double *grad) {
  hipStream_t s = c->stream;
  const int nth = ntheta(kid, d);
  double *slot = small_record_slot(c);
  if (!slot) return CGP_ENOMEM;
  SmallRaw raw;
  int rc = small_stage_window(c, X, y, N, d, nullptr, 0, theta, nth, raw);
  if (rc == CGP_OK) rc = small_launch(c, 1, N, d, kid, SM_MODE_EVAL, 1, s, slot, &raw, tick_table_entries(kid, d, X, N, nullptr, 0));
  double out[SM_OUT];
  if (rc == CGP_OK) rc = small_read_one(c, slot, out, s);
  if (rc != CGP_OK) return rc;
  const int info = (int)out[SMO_INFO];
  small_mark_fitted(c, X, y, N, d, kid, theta, out[SMO_JITTER], info == 0);
  if (info != 0) return info;
  *nll = -out[SMO_LOGML];
  for (int i = 0; i < nth; ++i) grad[i] = out[SMO_GRAD + i];
  return CGP_OK;
}

// m.optimize() of ONE short window: stage, one launch (the L-BFGS loop runs on the device), one copy back
int small_optimize(cgp_ctx *c, const double *X, const double *y, int N, int d, int kid, double *theta, int max_evals, double *logml,
                   int *n_evals) {
  hipStream_t s = c->stream;
  const int nth = ntheta(kid, d);
  for (int i = 0; i < nth; ++i)
    if (!(theta[i] > 0.0)) return CGP_EINVAL;
  double *slot = small_record_slot(c);
  if (!slot) return CGP_ENOMEM;
  SmallRaw raw;
  int rc = small_stage_window(c, X, y, N, d, nullptr, 0, theta, nth, raw);
  if (rc == CGP_OK) rc = small_launch(c, 1, N, d, kid, SM_MODE_OPT, max_evals, s, slot, &raw, tick_table_entries(kid, d, X, N, nullptr, 0));
  double out[SM_OUT];
  if (rc == CGP_OK) rc = small_read_one(c, slot, out, s);
  if (rc != CGP_OK) return rc;
  if (out[SMO_INFO] != 0.0) {   // the start itself is not positive definite even with the jitter ladder
    c->have_fit = false;
    return 1;
  }
  for (int i = 0; i < nth; ++i) theta[i] = out[SMO_THETA + i];
  small_mark_fitted(c, X, y, N, d, kid, theta, 0.0, true);
  if (logml) *logml = out[SMO_LOGML];
  if (n_evals) *n_evals = (int)out[SMO_EVALS];
  return CGP_OK;
}
}  // namespace

extern "C" int cgp_nll_grad(cgp_ctx *c, const double *X, const double *y, int N, int d, int kid, const double *theta,
                            double *nll, double *grad) {
  int rc = check_shape(c, 1, N, d, N, kid);
  if (rc != CGP_OK) return rc;
  if (!X || !y || !theta || !nll || !grad) return CGP_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  if (small_ok(c, N)) return small_nll_grad(c, X, y, N, d, kid, theta, nll, grad);
  c->lazy_fit = false;
  rc = upload_window(c, X, y, N, d, c->stream);
  if (rc != CGP_OK) return rc;
  return nll_grad_resident(c, X, N, d, kid, theta, nll, grad);
}

extern "C" int cgp_optimize(cgp_ctx *c, const double *X, const double *y, int N, int d, int kid, double *theta,
                            int max_evals, double *logml, int *n_evals) {
  int rc = check_shape(c, 1, N, d, N, kid);
  if (rc != CGP_OK) return rc;
  if (!X || !y || !theta) return CGP_EINVAL;
  const int nth = ntheta(kid, d);
  // Logexp transform (GPy paramz.transformations.Logexp): theta = log(1 + exp(x))
  auto to_theta = [](double x) { return x > 35.0 ? x : std::log1p(std::exp(x)); };
  auto to_x = [](double th) { return th > 35.0 ? th : std::log(std::expm1(th)); };
  std::vector<double> x(nth), th(nth), g(nth);
  for (int i = 0; i < nth; ++i) {
    if (!(theta[i] > 0.0)) return CGP_EINVAL;
    x[i] = to_x(theta[i]);
  }
  int hard_error = CGP_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  if (small_ok(c, N)) return small_optimize(c, X, y, N, d, kid, theta, max_evals, logml, n_evals);
  c->lazy_fit = false;
  rc = upload_window(c, X, y, N, d, c->stream);   // the window does not change between evaluations: only theta travels
  if (rc != CGP_OK) return rc;
  auto fg = [&](const std::vector<double> &xx, std::vector<double> &gx) -> double {
    for (int i = 0; i < nth; ++i) th[i] = std::max(to_theta(xx[i]), 1e-300);
    double nll = 0.0;
    const int r = nll_grad_resident(c, X, N, d, kid, th.data(), &nll, g.data());
    if (r < 0) hard_error = r;
    if (r != 0) return INFINITY;  // not PD even with jitter: infeasible point
    for (int i = 0; i < nth; ++i) gx[i] = g[i] * (xx[i] > 35.0 ? 1.0 : -std::expm1(-th[i]));  // dtheta/dx = 1 - exp(-theta)
    return nll;
  };
  corenav::LbfgsResult res = corenav::lbfgs_minimize(fg, x, max_evals > 0 ? max_evals : 1000);
  if (hard_error != CGP_OK) return hard_error;
  for (int i = 0; i < nth; ++i) theta[i] = to_theta(x[i]);
  // leave the context fitted at the optimum
  double nll = 0.0;
  rc = nll_grad_resident(c, X, N, d, kid, theta, &nll, g.data());
  if (rc != CGP_OK) return rc;
  if (logml) *logml = -nll;
  if (n_evals) *n_evals = res.evals + 1;
  return CGP_OK;
}


namespace {
// The reference node's whole callback (gp_slip_node.py:16-63) for a short window in ONE host round trip: stage the window,
// the prediction ticks and theta (one H2D), then queue  unpack -> [k_small: m.optimize() on the device, optimum left in the
// context's theta slot] -> k_small_predict (fit at that slot with GPy's jitter ladder + the predictions) -> one D2H, and
// synchronise once.  max_evals == 0: fixed theta.  CGP_ESTATE: not a short fp64 window -- the caller's general path.
int node_callback_small(cgp_ctx *c, const double *time_array, const double *slip_array, int n, int kid, double *theta, int max_evals,
                        double *mean, double *sigma, int cap, int *m_out) {
  const int ntr = (int)(0.9 * (double)n);  // gp_slip_node.py:27-29
  if (ntr < 1 || !small_ok(c, ntr) || check_shape(c, 1, ntr, 1, ntr, kid) != CGP_OK) return CGP_ESTATE;
  const int nth = ntheta(kid, 1);
  for (int i = 0; i < nth; ++i)
    if (!(theta[i] > 0.0)) return max_evals > 0 ? CGP_EINVAL : CGP_ESTATE;
  double xmin = time_array[0], xmax = time_array[0];   // gp_slip_node.py:45  X_ = np.arange(X.min(), X.max() + 600, 1)
  for (int i = 1; i < n; ++i) {
    xmin = std::min(xmin, time_array[i]);
    xmax = std::max(xmax, time_array[i]);
  }
  const long long glen = (long long)std::ceil((xmax + 600.0 - xmin) / 1.0);
  const long long mo = std::max(0LL, glen - n);         // gp_slip_node.py:59-61  means[len(X):]
  const int M = (int)std::min<long long>(mo, cap);
  if (M == 0 || M > c->max_m || !small_predict_ok(c, ntr, 1, M)) return CGP_ESTATE;
  if (m_out) *m_out = (int)mo;
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const size_t nout = 2 * (size_t)M + 2 * SM_OUT, out_bytes = (nout + 1) * sizeof(double);   // mean, var, the two kernels' records, the completion word
  if (!grow_pinned(c->pin_in, c->pin_in_cap, (2 * (size_t)ntr + M + CGP_MAX_THETA) * sizeof(double)) ||
      !grow_pinned(c->pin_out, c->pin_out_cap, out_bytes))
    return CGP_ENOMEM;
  // no copy commands and no unpack launch: the kernels read the pinned block in place and write their results (10 KB) to
  // pinned host memory, visible when the stream has drained
  SmallRaw raw;
  int rc = small_stage_window(c, time_array, slip_array, ntr, 1, nullptr, 0, theta, nth, raw);
  if (rc != CGP_OK) return rc;
  {   // the prediction ticks behind y, theta behind them
    double *hin = static_cast<double *>(c->pin_in), *xs = hin + 2 * (size_t)ntr;
    for (int m = 0; m < M; ++m) xs[m] = xmin + (double)(n + m);
    for (int q = 0; q < CGP_MAX_THETA; ++q) xs[M + q] = q < nth ? theta[q] : 0.0;
    raw.Xs = xs;
    raw.theta = xs + M;
  }
  double *hout = static_cast<double *>(c->pin_out);
  double *rec = hout + 2 * (size_t)M, *opt = rec + SM_OUT;
  if (max_evals > 0) {   // leaves the optimum in raw.theta, where the next launch reads it
    rc = small_launch(c, 1, ntr, 1, kid, SM_MODE_OPT, max_evals, s, opt, &raw, tick_table_entries(kid, 1, time_array, ntr, nullptr, 0));
    if (rc != CGP_OK) return rc;
  }
  // the last workgroup of the prediction launch stores the call's sequence number to a word of the pinned block (behind a system-scope
  // fence): the host polls it instead of synchronising the stream, and falls back to the synchronisation after 2 ms (a long
  // optimisation, a faulting kernel)
  volatile int *flag = reinterpret_cast<volatile int *>(hout + nout);
  const int seq = (c->small_seq = c->small_seq % 1000000 + 1);
  *flag = 0;
  rc = small_predict_launch(c, 1, ntr, 1, M, kid, 1, hout, hout + M, rec, s, &raw, nullptr, tick_table_entries(kid, 1, time_array, ntr, raw.Xs, M),
                            const_cast<int *>(reinterpret_cast<volatile int *>(flag)), seq);
  if (rc != CGP_OK) return rc;
  bool done = false;
  if (c->dsmdone) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int spin = 0;; ++spin) {
      if (*flag == seq) { done = true; break; }
      if ((spin & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  if (!done) HIP_TRY(c, hipStreamSynchronize(s));
  memcpy(c->last_small, max_evals > 0 ? opt : rec, SM_OUT * sizeof(double));
  if (max_evals > 0) {
    if (opt[SMO_INFO] != 0.0) {   // the start values themselves are not positive definite even with the jitter ladder
      c->have_fit = false;
      c->lazy_fit = false;
      return 1;
    }
    for (int i = 0; i < nth; ++i) theta[i] = opt[SMO_THETA + i];
  }
  const int info = (int)rec[SMO_INFO];
  small_mark_fitted(c, time_array, slip_array, ntr, 1, kid, theta, rec[SMO_JITTER], info == 0);
  if (info != 0) return info;  // not positive definite even with GPy's jitter ladder (LinAlgError there)
  memcpy(mean, hout, (size_t)M * sizeof(double));
  for (int m = 0; m < M; ++m) sigma[m] = 2.0 * std::sqrt(hout[M + m]);   // gp_slip_node.py:61
  return CGP_OK;
}
}  // namespace

extern "C" int cgp_slip_node_callback_opt(cgp_ctx *c, const double *time_array, const double *slip_array, int n,
                                          int kid, double *theta, int max_evals, double *mean, double *sigma, int cap,
                                          int *m_out) {
  if (!c || !time_array || !slip_array || !theta || n < 2) return CGP_EINVAL;
  if (max_evals > 0 && mean && sigma && cap >= 0) {
    const int rc = node_callback_small(c, time_array, slip_array, n, kid, theta, max_evals, mean, sigma, cap, m_out);
    if (rc != CGP_ESTATE) return rc;   // CGP_ESTATE: not a short fp64 window -- the two-call path below
  }
  if (max_evals > 0) {
    const int ntr = (int)(0.9 * (double)n);  // gp_slip_node.py:27-29
    int rc = cgp_optimize(c, time_array, slip_array, ntr, 1, kid, theta, max_evals, nullptr, nullptr);  // :36
    if (rc != CGP_OK) return rc;
  }
  return cgp_slip_node_callback(c, time_array, slip_array, n, kid, theta, mean, sigma, cap, m_out);
}

extern "C" int cgp_window_init(cgp_ctx *c, int nwin, int N, int d, int kid, const double *theta, int theta_stride) {
  if (!c || nwin < 1 || N < 2 || N > 2048 || d < 1 || d > CGP_MAX_D || !theta || kid < 0 || kid > 2) return CGP_EINVAL;
  if (kid == CGP_KERNEL_RBF_BROWNIAN && d != 1) return CGP_EINVAL;
  const int nth = ntheta(kid, d);
  if (theta_stride < nth) return CGP_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  // the two-ticks-per-pass kernel keeps six window-length vectors in LDS: 98 KB at N = 2048
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_window_pairs<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void *>(&k_window_pairs<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void *>(&k_window_pairs<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void *>(&k_window_multi<kWinMulti>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
    return CGP_EHIP;
  // the old windows are gone from here on: a failure below must leave the context without windows,
  // not with stale pointers (cgp_window_push checks nwin)
  c->nwin = 0;
  c->win = WindowArgs{};
  auto drop = [c]() {
    for (void *&wb : c->winbuf) {
      if (wb) (void)hipFree(wb);
      wb = nullptr;
    }
  };
  drop();
#ifndef CGP_WIN_CAP_PAD
#define CGP_WIN_CAP_PAD 0
#endif
  const int CAP = 2 * N + CGP_WIN_CAP_PAD;   // ring capacity = leading dimension of the windows' slabs
  const size_t W = nwin;
  size_t sizes[6] = {W * CAP * CAP * 8, W * CAP * 8, W * d * CAP * 8, W * CAP * 8, W * 4 * sizeof(int),
                     W * (PREP_N + MAX_THETA) * 8};
  for (int i = 0; i < 6; ++i)
    if (hipMalloc(&c->winbuf[i], sizes[i]) != hipSuccess) {
      c->winbuf[i] = nullptr;
      drop();
      return CGP_ENOMEM;
    }
  WindowArgs wa{};
  wa.L = (double *)c->winbuf[0];
  wa.z = (double *)c->winbuf[1];
  wa.xw = (double *)c->winbuf[2];
  wa.yw = (double *)c->winbuf[3];
  wa.state = (int *)c->winbuf[4];
  double *pt = (double *)c->winbuf[5];
  wa.prep = pt;
  wa.theta = pt + W * PREP_N;
  wa.N = N;
  wa.CAP = CAP;
  wa.d = d;
  wa.kernel_id = kid;
  std::vector<double> h(W * (PREP_N + MAX_THETA), 0.0);
  for (size_t w = 0; w < W; ++w) {
    const double *th = theta + w * theta_stride;
    double *o = h.data() + w * PREP_N;
    for (int q = 0; q < d; ++q) o[q] = (kid == CGP_KERNEL_SE_ARD) ? 1.0 / th[1 + q] : 1.0 / th[1];
    o[9] = th[0];
    o[10] = (kid == CGP_KERNEL_RBF_BROWNIAN) ? th[2] : 0.0;
    for (int q = 0; q < nth; ++q) h[W * PREP_N + w * MAX_THETA + q] = th[q];
  }
  // hipMemset of device memory is ASYNCHRONOUS to the host and ordered on the legacy default stream only -- the pushes run on the
  // context's non-blocking stream (or the caller's), which does not wait for it: without the synchronisation below the first
  // push could read the windows' state words before they were zeroed (found by round 6's sweep under load: fourteen processes
  // sharing the GPU -- a memory access fault, or garbage for one window; never seen on an idle GPU, where the fill is over
  // before the first launch is issued).
  if (!hip_ok(c, hipMemcpy(pt, h.data(), h.size() * 8, hipMemcpyHostToDevice), "window theta H2D") ||
      !hip_ok(c, hipMemset(wa.state, 0, W * 4 * sizeof(int)), "window state memset") ||
      !hip_ok(c, hipDeviceSynchronize(), "window init synchronise")) {
    drop();
    return CGP_EHIP;
  }
  c->win = wa;  // published only when every allocation and copy has succeeded
  c->nwin = nwin;
  c->win_o = c->win_n = 0;
  return CGP_OK;
}

namespace {
int window_push_impl(cgp_ctx *c, int T, const double *dxs, const double *dys, int include_noise, double *dpm, double *dpv, double *dl,
                     int *info_out, hipStream_t ws);
}
extern "C" int cgp_window_push_device(cgp_ctx *c, int T, const double *dxs, const double *dys, int include_noise,
                                      double *dpm, double *dpv, double *dl, void *hip_stream) {
  if (!c || c->nwin < 1) return CGP_ESTATE;
  if (T < 1 || !dxs || !dys || !dpm || !dpv || !dl) return CGP_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  return window_push_impl(c, T, dxs, dys, include_noise, dpm, dpv, dl, nullptr, pick_stream(c, hip_stream));
}
namespace {
// info_out: [nwin] ints the kernels mirror every window's status word into (pinned host memory for the per-tick entry), or null
int window_push_impl(cgp_ctx *c, int T, const double *dxs, const double *dys, int include_noise, double *dpm, double *dpv, double *dl,
                     int *info_out, hipStream_t ws) {
  WindowArgs a = c->win;
  a.info_out = info_out;
  a.xs = dxs;
  a.ys = dys;
  a.pred_mean = dpm;
  a.pred_var = dpv;
  a.logml = dl;
  a.T = T;
  a.include_noise = include_noise;
  // The T ticks are cut into launches: runs of steady-state ticks (full windows, no ring compaction inside) go two per
  // pass over the factor (k_window_pairs), everything else -- filling, the tick that compacts the ring, an odd one out --
  // through the single-tick kernel.  Origin and size of the windows are deterministic and identical for every window of
  // the context, so the host mirrors them instead of reading them back.
  const size_t lds1 = (size_t)(3 * a.N + 8 * WPB + MAXD + 8 + 2 * WIN_STG) * sizeof(double);
  const size_t lds2 = (size_t)(6 * ((a.N + 3) & ~1) + 16 * WPB + 2 * MAXD + 16) * sizeof(double);   // per window
  // windows per workgroup of the paired kernel (rows of wave 0 per window: 4 / wpw)
  // measured (tools/r3_winpack.sh, N = 512): 1024 windows 2.20 / 2.65 / 2.03 M ticks/s at 1 / 2 / 4 per workgroup, 512 windows
  // 2.18 / 1.81 / 1.20 -- two per workgroup once that still leaves two workgroups per CU, four never
  int wpw = 1;
  if (c->nwin % 2 == 0 && 2 * lds2 + kWinPairStage <= (size_t)kWinPackLds + 8 * 1024 && c->nwin / 2 >= kWinPackMinGroups) wpw = 2;
  if constexpr (kAbBuild) {
    const char *e = getenv("CGP_WIN_WPW");
    const int v = e ? atoi(e) : 0;
    if ((v == 1 || v == 2 || v == 4) && c->nwin % v == 0 && v * lds2 + kWinPairStage <= 150 * 1024) wpw = v;
  }
  const int N = a.N, CAP = a.CAP;
  if (c->win_o < 0) {   // the mirror was invalidated by a failed push: read the windows' state back (they advance in lock-step)
    int st[4];
    HIP_TRY(c, hipStreamSynchronize(ws));
    HIP_TRY(c, hipMemcpy(st, c->win.state, sizeof(st), hipMemcpyDeviceToHost));
    c->win_o = st[0];
    c->win_n = st[1];
  }
  int o = c->win_o, n = c->win_n;
  auto one_tick = [&](int &oo, int &nn) {   // k_window_ticks, one tick
    if (oo + nn >= CAP) oo = 0;
    const bool drop = nn >= N;
    oo = drop ? oo + 1 : oo;
    nn = (drop ? nn - 1 : nn) + 1;
  };
  auto pair_ok = [&](int oo, int nn, int left) { return kWinPairs && N >= 2 * WPB && nn == N && left >= 2 && oo + N + 1 < CAP; };
  const int NSm = (N + kWinMulti + 3) & ~1;
  const size_t ldsm = (size_t)(3 * kWinMulti * NSm + kWinMulti * (8 * WPB + MAXD + 8)) * sizeof(double) + kWinPairStage;
  auto multi_ok = [&](int oo, int nn, int left) {
    // (measured, N = 512: 512 windows 3.99 M ticks/s against 3.54 M two per pass, 1 024 windows 3.97 against 3.41; 256 windows 2.92 against 3.38 --
    // one window per workgroup leaves half of a small call's lanes idle: from kWinMultiMinWindows windows)
    return kWinUseMulti && kWinPairs && c->nwin >= kWinMultiMinWindows && N >= 4 * WPB && ldsm <= 80 * 1024 && nn == N && left >= kWinMulti &&
           oo + N + kWinMulti - 1 < CAP;
  };
  for (int t = 0; t < T;) {
    a.t0 = t;
    int nm = 0;
    for (int oo = o; multi_ok(oo, n, T - t - kWinMulti * nm); oo += kWinMulti) ++nm;
    if (nm > 0) {
      a.nt = kWinMulti * nm;
      hipLaunchKernelGGL(k_window_multi<kWinMulti>, dim3(c->nwin), dim3(256), ldsm, ws, a);
      o += kWinMulti * nm;
      t += kWinMulti * nm;
      continue;
    }
    int np = 0;
    for (int oo = o; pair_ok(oo, n, T - t - 2 * np); oo += 2) ++np;
    if (np > 0) {
      a.nt = 2 * np;
      if (wpw == 4) hipLaunchKernelGGL(k_window_pairs<4>, dim3(c->nwin / 4), dim3(256), 4 * lds2 + kWinPairStage, ws, a);
      else if (wpw == 2) hipLaunchKernelGGL(k_window_pairs<2>, dim3(c->nwin / 2), dim3(256), 2 * lds2 + kWinPairStage, ws, a);
      else hipLaunchKernelGGL(k_window_pairs<1>, dim3(c->nwin), dim3(256), lds2 + kWinPairStage, ws, a);
      o += 2 * np;
      t += 2 * np;
      continue;
    }
    int ns = 0;
    do {
      one_tick(o, n);
      ++ns;
    } while (t + ns < T && !pair_ok(o, n, T - t - ns));
    a.nt = ns;
    // threads per window: with no more windows than CUs a workgroup sweeps with seven waves instead of three (1024 threads: 128 VGPRs
    // per lane, the serial wave spills -- 251 us per host tick against 117)
    int wth = c->nwin <= kWinWideMax ? 512 : 256;
    if constexpr (kAbBuild) {
      const char *e = getenv("CGP_WIN_THREADS");
      const int v = e ? atoi(e) : 0;
      if (v == 256 || v == 512) wth = v;
    }
    if (wth == 512) hipLaunchKernelGGL(k_window_ticks<512>, dim3(c->nwin), dim3(512), lds1, ws, a);
    else hipLaunchKernelGGL(k_window_ticks<256>, dim3(c->nwin), dim3(256), lds1, ws, a);
    t += ns;
  }
  if (!hip_ok(c, hipGetLastError(), "window launches")) {
    c->win_o = c->win_n = -1;   // what reached the device is unknown: the next push re-reads the state
    return CGP_EHIP;
  }
  c->win_o = o;   // committed only once every launch of the push was accepted
  c->win_n = n;
  return CGP_OK;
}
}  // namespace

extern "C" int cgp_window_push(cgp_ctx *c, int T, const double *xs, const double *ys, int include_noise, double *pm,
                               double *pv, double *logml) {
  if (!c || c->nwin < 1) return CGP_ESTATE;
  if (T < 1 || !xs || !ys || !pm || !pv || !logml) return CGP_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  // The per-tick host entry (configs[3] is "streamed per IMU tick": T = 1 is the common call): no allocation and no
  // pageable copy on the path.  One pinned block [xs | ys | pm pv logml | state] and its device twin live in the
  // context; per call: stage in, ONE H2D, the launch, TWO D2H (outputs, window states), one synchronisation.
  const size_t W = c->nwin, nx = W * T * c->win.d, ny = W * T;
  const size_t ndbl = nx + 4 * ny, bytes = ndbl * 8 + W * 4 * sizeof(int);
  // (the pinned block is never smaller than the largest push the kernels access in place: a block that was freed and allocated again
  // between two small pushes -- the first pushes of a stream grow -- is what round 6's sweep caught under load, fourteen processes on
  // the GPU: about one push in a hundred came back with its outputs untouched, the kernel's stores having gone to the pages of the
  // block just freed.  One allocation for the context's lifetime takes the window out of that path.)
  if (!grow_pinned(c->win_pin, c->win_pin_cap, std::max(bytes, kWinZeroCopyBytes)) || !grow_device(c->win_dev, c->win_dev_cap, ndbl * 8)) return CGP_ENOMEM;
  double *h = static_cast<double *>(c->win_pin), *d = static_cast<double *>(c->win_dev);
  int *hst = reinterpret_cast<int *>(h + ndbl);
  memcpy(h, xs, nx * 8);
  memcpy(h + nx, ys, ny * 8);
  hipStream_t s = c->stream;
  if (bytes <= kWinZeroCopyBytes) {
    // A tick or a handful of them (configs[3] is "streamed per IMU tick"): no copy command at all.  The kernels read the
    // samples where they were staged (pinned host memory is device-visible at its host address) and write the tick's
    // outputs and every window's status word back there; one synchronisation.  (Round 4: one H2D, two D2H, 176 us per tick of
    // one N = 512 window, ~30 us of it the three copy commands; the tick itself is 147 us of one workgroup's serial chains.)
    // T = 1 is ONE launch of the single-tick kernel, whose last store is the window's status word (after a system-scope fence): the host
    // polls those words in the pinned block instead of synchronising the stream (a few microseconds of the runtime's wake-up), and
    // falls back to the synchronisation if they do not arrive (which is also where a faulting kernel's error surfaces)
    constexpr int kPending = INT_MIN;
    const bool poll = T == 1;
    for (size_t w = 0; w < W; ++w) hst[w] = poll ? kPending : 0;
    int rc0 = window_push_impl(c, T, h, h + nx, include_noise, h + nx + ny, h + nx + 2 * ny, h + nx + 3 * ny, hst, s);
    if (rc0 != CGP_OK) return rc0;
    bool done = false;
    if (poll) {
      volatile int *v = hst;
      const auto t0 = std::chrono::steady_clock::now();
      for (int spin = 0;; ++spin) {
        size_t w = 0;
        while (w < W && v[w] != kPending) ++w;
        if (w == W) { done = true; break; }
        if ((spin & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
      }
      std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!done) HIP_TRY(c, hipStreamSynchronize(s));
    memcpy(pm, h + nx + ny, ny * 8);
    memcpy(pv, h + nx + 2 * ny, ny * 8);
    memcpy(logml, h + nx + 3 * ny, ny * 8);
    for (size_t w = 0; w < W; ++w)
      if (hst[w] != 0) return hst[w];
    return CGP_OK;
  }
  HIP_TRY(c, hipMemcpyAsync(d, h, (nx + ny) * 8, hipMemcpyHostToDevice, s));
  int rc = cgp_window_push_device(c, T, d, d + nx, include_noise, d + nx + ny, d + nx + 2 * ny, d + nx + 3 * ny, s);
  if (rc != CGP_OK) return rc;
  HIP_TRY(c, hipMemcpyAsync(h + nx + ny, d + nx + ny, 3 * ny * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipMemcpyAsync(hst, c->win.state, W * 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  memcpy(pm, h + nx + ny, ny * 8);
  memcpy(pv, h + nx + 2 * ny, ny * 8);
  memcpy(logml, h + nx + 3 * ny, ny * 8);
  for (size_t w = 0; w < W; ++w)
    if (hst[w * 4 + 2] != 0) return hst[w * 4 + 2];
  return CGP_OK;
}

extern "C" int cgp_window_state(cgp_ctx *c, int w, int *n, int *info) {
  if (!c || c->nwin < 1) return CGP_ESTATE;
  if (w < 0 || w >= c->nwin) return CGP_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipDeviceSynchronize());
  int st[4];
  HIP_TRY(c, hipMemcpy(st, c->win.state + w * 4, sizeof(st), hipMemcpyDeviceToHost));
  if (n) *n = st[1];
  if (info) *info = st[2];
  return CGP_OK;
}

struct cgp_recorder {
  corenav::SlipWindowRecorder r;
};
extern "C" cgp_recorder *cgp_recorder_create(void) { return new cgp_recorder(); }
extern "C" void cgp_recorder_destroy(cgp_recorder *rec) { delete rec; }
extern "C" int cgp_recorder_update(cgp_recorder *rec, const double wv[4], double vlin, double cmd_x, double *slip_out,
                                   double *time_out, double *slipwin_out, int cap, int *n_out) {
  if (!rec || !wv) return CGP_EINVAL;
  const bool pub = rec->r.Update(wv[0], wv[1], wv[2], wv[3], vlin, cmd_x);
  if (slip_out) *slip_out = rec->r.slip;
  if (pub) {
    const int n = (int)rec->r.time_array.size();
    if (n_out) *n_out = n;
    for (int i = 0; i < std::min(n, cap); ++i) {
      if (time_out) time_out[i] = rec->r.time_array[i];
      if (slipwin_out) slipwin_out[i] = rec->r.slip_array[i];
    }
    if (n > cap && (time_out || slipwin_out)) return CGP_ECAPACITY;  // *n_out = the size the window needs
  }
  return pub ? 1 : 0;
}
extern "C" void cgp_recorder_stop_cmd(cgp_recorder *rec, double cmd_stop) {
  if (rec) rec->r.stopCallback(cmd_stop);
}
extern "C" void cgp_recorder_cmd(cgp_recorder *rec, double cmd_x) {
  if (rec) rec->r.CmdCallBack(cmd_x);
}
extern "C" void cgp_recorder_state(const cgp_recorder *rec, double st[8]) {
  if (!rec || !st) return;
  const auto &r = rec->r;
  st[0] = r.odomUptCount; st[1] = r.startRecording; st[2] = r.stopRecording; st[3] = r.gp_flag;
  st[4] = r.first_driving_flag; st[5] = r.new_stop_data_arrived_; st[6] = r.skipped_windows; st[7] = r.cmd_stop_;
}

extern "C" int cgp_predict_stop_batch(cgp_ctx *c, int ntraj, int M, const double *mean, const double *sigma,
                                      const double *P, const double *Q, const double *STM, const double *Hvec,
                                      const double *pos, const double *arrival, const double *now, double threshold,
                                      int h_bug, const double init_llh[3], const double init_ecef[3], int *fired,
                                      double *stop_cmd, int *i_out, double *xy_err) {
  if (!c || ntraj < 1 || M < 0 || !mean || !sigma || !P || !Q || !STM || !Hvec || !pos || !arrival || !now ||
      !init_llh || !init_ecef || !fired || !stop_cmd || !i_out || !xy_err)
    return CGP_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t T = ntraj;
  const size_t nd = T * (2 * (size_t)M + 3 * 225 + 60 + 3 + 4 + 2 + 2);  // doubles in (incl. 4 trig values), 2 double outputs
  // staging buffers live in the context and only grow (the replay loop calls this once per tick group)
  if (nd > c->la_nd) {
    if (c->la_buf) (void)hipFree(c->la_buf);
    c->la_buf = nullptr;
    c->la_nd = 0;
    if (hipMalloc((void **)&c->la_buf, nd * sizeof(double)) != hipSuccess) return CGP_ENOMEM;
    c->la_nd = nd;
  }
  if (T * 2 > c->la_ni) {
    if (c->la_ibuf) (void)hipFree(c->la_ibuf);
    c->la_ibuf = nullptr;
    c->la_ni = 0;
    if (hipMalloc((void **)&c->la_ibuf, T * 2 * sizeof(int)) != hipSuccess) return CGP_ENOMEM;
    c->la_ni = T * 2;
  }
  double *buf = c->la_buf;
  int *ibuf = c->la_ibuf;
  hipStream_t s = c->stream;
  double *d = buf;
  LookaheadArgs a{};
  auto put = [&](const double *src, size_t n) {
    double *dst = d;
    (void)hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyHostToDevice, s);
    d += n;
    return dst;
  };
  a.mean = put(mean, T * M);
  a.sigma = put(sigma, T * M);
  a.P = put(P, T * 225);
  a.Q = put(Q, T * 225);
  a.STM = put(STM, T * 225);
  a.Hvec = put(Hvec, T * 60);
  a.pos = put(pos, T * 3);
  // sines / cosines of the base positions and of the ENU origin on the host: libm's large-argument
  // paths would otherwise set the kernel's register and scratch footprint
  std::vector<double> trig(T * 4);
  for (size_t t = 0; t < T; ++t) {
    trig[4 * t + 0] = std::sin(pos[3 * t]);
    trig[4 * t + 1] = std::cos(pos[3 * t]);
    trig[4 * t + 2] = std::sin(pos[3 * t + 1]);
    trig[4 * t + 3] = std::cos(pos[3 * t + 1]);
  }
  a.trig = put(trig.data(), T * 4);
  a.origin_trig[0] = std::sin(init_llh[0]);
  a.origin_trig[1] = std::cos(init_llh[0]);
  a.origin_trig[2] = std::sin(init_llh[1]);
  a.origin_trig[3] = std::cos(init_llh[1]);
  a.arrival = put(arrival, T);
  a.now = put(now, T);
  a.stop_cmd = d;
  a.xy_err = d + T;
  a.fired = ibuf;
  a.i_out = ibuf + T;
  a.ntraj = ntraj;
  a.M = M;
  a.h_bug_compatible = h_bug;
  a.threshold = threshold;
  for (int i = 0; i < 3; ++i) {
    a.init_llh[i] = init_llh[i];
    a.init_ecef[i] = init_ecef[i];
  }
  hipLaunchKernelGGL(k_lookahead, dim3(cdiv(ntraj, LA_WAVES)), dim3(64 * LA_WAVES),
                     (size_t)LA_WAVES * LA_PER_WAVE * sizeof(double), s, a);
  int rc = CGP_OK;
  if (!hip_ok(c, hipGetLastError(), "k_lookahead")) rc = CGP_EHIP;
  std::vector<int> hi(T * 2);
  if (rc == CGP_OK && !hip_ok(c, hipMemcpyAsync(stop_cmd, a.stop_cmd, T * 8, hipMemcpyDeviceToHost, s), "D2H")) rc = CGP_EHIP;
  if (rc == CGP_OK && !hip_ok(c, hipMemcpyAsync(xy_err, a.xy_err, T * 8, hipMemcpyDeviceToHost, s), "D2H")) rc = CGP_EHIP;
  if (rc == CGP_OK && !hip_ok(c, hipMemcpyAsync(hi.data(), ibuf, T * 2 * sizeof(int), hipMemcpyDeviceToHost, s), "D2H")) rc = CGP_EHIP;
  if (!hip_ok(c, hipStreamSynchronize(s), "sync")) rc = CGP_EHIP;
  if (rc != CGP_OK) return rc;
  for (size_t i = 0; i < T; ++i) {
    fired[i] = hi[i];
    i_out[i] = hi[T + i];
  }
  return CGP_OK;
}

extern "C" int cgp_selftest_lbfgs(double *x, int n, int max_evals, double *f_out) {
  if (!x || n < 2 || n > 16) return CGP_EINVAL;
  std::vector<double> xv(x, x + n);
  auto fg = [n](const std::vector<double> &v, std::vector<double> &g) {
    double f = 0.0;
    for (int i = 0; i < n; ++i) g[i] = 0.0;
    for (int i = 0; i + 1 < n; ++i) {
      const double a = v[i + 1] - v[i] * v[i], b = 1.0 - v[i];
      f += 100.0 * a * a + b * b;
      g[i] += -400.0 * a * v[i] - 2.0 * b;
      g[i + 1] += 200.0 * a;
    }
    return f;
  };
  corenav::LbfgsResult r = corenav::lbfgs_minimize(fg, xv, max_evals > 0 ? max_evals : 1000, 1e-8, 10.0);
  for (int i = 0; i < n; ++i) x[i] = xv[i];
  if (f_out) *f_out = r.f;
  return r.status == 3 ? -1 : r.evals;
}

extern "C" int cgp_lbfgs_minimize(cgp_objective_fn fn, void *user, double *x, int n, int max_evals, double pgtol, double factr,
                                  double *f_out, int *n_evals, int *n_iters, int *status) {
  if (!fn || !x || n < 1 || n > corenav::LB_N) return CGP_EINVAL;
  std::vector<double> xv(x, x + n);
  auto fg = [&](const std::vector<double> &v, std::vector<double> &g) { return fn(v.data(), g.data(), n, user); };
  corenav::LbfgsResult r = corenav::lbfgs_minimize(fg, xv, max_evals > 0 ? max_evals : 1000, pgtol, factr);
  for (int i = 0; i < n; ++i) x[i] = xv[i];
  if (f_out) *f_out = r.f;
  if (n_evals) *n_evals = r.evals;
  if (n_iters) *n_iters = r.iters;
  if (status) *status = r.status;
  return CGP_OK;
}

namespace {
// Gradient-mode evaluation of `batch` windows already uploaded to slots 0..batch-1 (theta in dtheta).
template <typename T>
int grad_eval_batch(cgp_ctx *c, int batch, int N, int d, int kid, std::vector<double> &logml, std::vector<double> &sums,
                    std::vector<int> &info) {
  hipStream_t s = c->stream;
  FitArgs a = base_args(c, N, d, /*M=*/N, kid, 0);
  a.xid = 1;
  a.X = c->dX;
  a.Xs = c->dX;
  a.y = c->dy;
  a.theta = c->dtheta;
  a.jitter = c->djitter;  // per window; zero unless the jitter ladder of cgp_optimize_batch is climbing
  a.mean = c->dmean;
  a.var = c->dvar;
  a.logml = c->dlogml;
  a.info = c->dinfo;
  a.gpart = c->dgpart;
  int rc = run(c, a, batch, true, true, s);
  if (rc != CGP_OK) return rc;
  const int npairs = a.NT * (a.NT + 1) / 2;
  hipLaunchKernelGGL(k_grad<T>, dim3(npairs, batch), dim3(256), upd_lds_bytes<T>(), s, a, npairs);
  HIP_TRY(c, hipGetLastError());
  std::vector<double> part((size_t)batch * npairs * GRAD_N);
  logml.resize(batch);
  info.resize(batch);
  HIP_TRY(c, hipMemcpyAsync(part.data(), c->dgpart, part.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipMemcpyAsync(logml.data(), c->dlogml, sizeof(double) * batch, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipMemcpyAsync(info.data(), c->dinfo, sizeof(int) * batch, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  sums.assign((size_t)batch * GRAD_N, 0.0);
  for (int b = 0; b < batch; ++b)
    for (int pz = 0; pz < npairs; ++pz)
      for (int i = 0; i < GRAD_N; ++i) sums[(size_t)b * GRAD_N + i] += part[((size_t)b * npairs + pz) * GRAD_N + i];
  return CGP_OK;
}

// d(-logML)/dtheta from the k_grad sums (same formulas as cgp_nll_grad)
void grad_from_sums(int kid, int d, const double *theta, const double *sums, double *grad) {
  if (kid == CGP_KERNEL_SE_ISO) {
    double se = 0;
    for (int q = 0; q < d; ++q) se += sums[1 + q];
    grad[0] = -0.5 * sums[0] / theta[0];
    grad[1] = -0.5 * se / theta[1];
    grad[2] = -0.5 * sums[9];
  } else if (kid == CGP_KERNEL_SE_ARD) {
    grad[0] = -0.5 * sums[0] / theta[0];
    for (int q = 0; q < d; ++q) grad[1 + q] = -0.5 * sums[1 + q] / theta[1 + q];
    grad[d + 1] = -0.5 * sums[9];
  } else {
    grad[0] = -0.5 * sums[0] / theta[0];
    grad[1] = -0.5 * sums[1] / theta[1];
    grad[2] = -0.5 * sums[0] / theta[2];
    grad[3] = -0.5 * sums[9];
  }
}
}  // namespace

extern "C" int cgp_optimize_batch(cgp_ctx *c, int batch, int N, int d, int kid, const double *X, const double *y,
                                  double *theta, int theta_stride, int max_evals, double *logml_out, int *n_evals) {
  int rc = check_shape(c, batch, N, d, N, kid);
  if (rc != CGP_OK) return rc;
  if (!X || !y || !theta) return CGP_EINVAL;
  const int nth = ntheta(kid, d);
  if (theta_stride < nth) return CGP_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const size_t esz = c->esz;
  std::vector<char> hx((size_t)batch * d * N * esz), hy((size_t)batch * N * esz);
  for (int b = 0; b < batch; ++b) {
    pack_soa(X + (size_t)b * N * d, N, d, c->dtype, hx, (size_t)b * d * N);
    pack_vec(y + (size_t)b * N, N, c->dtype, hy, (size_t)b * N);
  }
  HIP_TRY(c, hipMemcpyAsync(c->dX, hx.data(), hx.size(), hipMemcpyHostToDevice, s));
  HIP_TRY(c, hipMemcpyAsync(c->dy, hy.data(), hy.size(), hipMemcpyHostToDevice, s));
  c->lazy_fit = false;
  if (small_ok(c, N)) {
    // short windows: ONE launch, one workgroup per window, each running its own L-BFGS loop on the device (cgp_small.hpp)
    for (int b = 0; b < batch; ++b)
      for (int i = 0; i < nth; ++i)
        if (!(theta[(size_t)b * theta_stride + i] > 0.0)) return CGP_EINVAL;
    std::vector<double> hth;
    rc = upload_theta(c, theta, theta_stride, nth, batch, s, hth);
    if (rc == CGP_OK) rc = small_launch(c, batch, N, d, kid, SM_MODE_OPT, max_evals, s, nullptr, nullptr, tick_table_entries_batch(kid, d, batch, X, N, nullptr, 0));
    c->last_small_dev = true;
    if (rc != CGP_OK) return rc;
    std::vector<double> out((size_t)batch * SM_OUT);
    HIP_TRY(c, hipMemcpyAsync(out.data(), c->dsmall, out.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->have_fit = false;
    for (int b = 0; b < batch; ++b) {
      const double *ob = out.data() + (size_t)b * SM_OUT;
      for (int i = 0; i < nth; ++i) theta[(size_t)b * theta_stride + i] = ob[SMO_THETA + i];
      if (logml_out) logml_out[b] = ob[SMO_LOGML];
      if (n_evals) n_evals[b] = (int)ob[SMO_EVALS];
    }
    return CGP_OK;
  }
  auto to_theta = [](double x) { return x > 35.0 ? x : std::log1p(std::exp(x)); };
  auto to_x = [](double th) { return th > 35.0 ? th : std::log(std::expm1(th)); };
  std::vector<corenav::LbfgsStepper> st;
  st.reserve(batch);
  for (int b = 0; b < batch; ++b) {
    std::vector<double> x0(nth);
    for (int i = 0; i < nth; ++i) {
      if (!(theta[(size_t)b * theta_stride + i] > 0.0)) return CGP_EINVAL;
      x0[i] = to_x(theta[(size_t)b * theta_stride + i]);
    }
    st.emplace_back(x0, max_evals > 0 ? max_evals : 1000, 1e-5, 1e7);
  }
  std::vector<double> th((size_t)batch * nth), hth, lml, sums, g(nth), gx(nth), jit(batch, 0.0), lml2, sums2;
  std::vector<int> info, info2;
  c->have_fit = false;
  auto eval = [&](std::vector<double> &l, std::vector<double> &sm, std::vector<int> &inf) {
    return c->dtype == CGP_F64 ? grad_eval_batch<double>(c, batch, N, d, kid, l, sm, inf)
                               : grad_eval_batch<float>(c, batch, N, d, kid, l, sm, inf);
  };
  for (int round = 0; round < (max_evals > 0 ? max_evals : 1000) + 40; ++round) {
    bool any = false;
    for (int b = 0; b < batch; ++b) {
      any = any || !st[b].done();
      const std::vector<double> &xx = st[b].done() ? st[b].best() : st[b].trial();
      for (int i = 0; i < nth; ++i) th[(size_t)b * nth + i] = std::max(to_theta(xx[i]), 1e-300);
    }
    if (!any) break;
    rc = upload_theta(c, th.data(), nth, nth, batch, s, hth);
    if (rc != CGP_OK) return rc;
    HIP_TRY(c, hipMemsetAsync(c->djitter, 0, sizeof(double) * batch, s));
    rc = eval(lml, sums, info);
    if (rc != CGP_OK) return rc;
    // GPy jitchol inside m.optimize(): a trial point whose matrix is not positive definite is retried
    // with jitter mean(diag) 1e-6 10^k, k = 0..4.  Only the windows that failed climb the ladder (the
    // others keep jitter 0 and their first results); the re-evaluation is one more batched schedule.
    for (int attempt = 0; attempt < 5; ++attempt) {
      bool any_bad = false;
      for (int b = 0; b < batch; ++b) {
        if (st[b].done() || info[b] == 0) continue;
        any_bad = true;
        jit[b] = attempt == 0 ? mean_diag(kid, th.data() + (size_t)b * nth, d, X + (size_t)b * N * d, N) * 1e-6 : jit[b] * 10.0;
      }
      if (!any_bad) break;
      HIP_TRY(c, hipMemcpyAsync(c->djitter, jit.data(), sizeof(double) * batch, hipMemcpyHostToDevice, s));
      rc = eval(lml2, sums2, info2);
      if (rc != CGP_OK) return rc;
      for (int b = 0; b < batch; ++b) {
        if (st[b].done() || info[b] == 0) continue;
        info[b] = info2[b];
        lml[b] = lml2[b];
        std::copy(sums2.begin() + (size_t)b * GRAD_N, sums2.begin() + (size_t)(b + 1) * GRAD_N, sums.begin() + (size_t)b * GRAD_N);
      }
    }
    std::fill(jit.begin(), jit.end(), 0.0);
    for (int b = 0; b < batch; ++b) {
      if (st[b].done()) continue;
      const double *tb = th.data() + (size_t)b * nth;
      double f = INFINITY;
      if (info[b] == 0) {
        grad_from_sums(kid, d, tb, sums.data() + (size_t)b * GRAD_N, g.data());
        const std::vector<double> &xx = st[b].trial();
        for (int i = 0; i < nth; ++i) gx[i] = g[i] * (xx[i] > 35.0 ? 1.0 : -std::expm1(-tb[i]));
        f = -lml[b];
      }
      st[b].tell(f, gx);
    }
  }
  for (int b = 0; b < batch; ++b) {
    const std::vector<double> &xb = st[b].best();
    for (int i = 0; i < nth; ++i) theta[(size_t)b * theta_stride + i] = to_theta(xb[i]);
    const corenav::LbfgsResult r = st[b].result();
    if (logml_out) logml_out[b] = -r.f;
    if (n_evals) n_evals[b] = r.evals;
  }
  return CGP_OK;
}
