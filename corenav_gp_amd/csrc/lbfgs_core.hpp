// lbfgs_core.hpp -- the optimiser under GPy's `m.optimize()` (core_navigation/script/gp_slip_node.py:36; paramz 'lbfgsb' =
// scipy.optimize.fmin_l_bfgs_b, no bounds, m = 10, factr 1e7, pgtol 1e-5, maxfun 1000) restated as ONE fixed-size struct
// that compiles for the host and for the device.  Without bounds L-BFGS-B (Byrd, Lu, Nocedal, Zhu 1995; Zhu et al. 1997,
// version 3.0) is:
//   direction  d = -H g, H the limited-memory BFGS inverse with H0 = (s'y / y'y) I of the newest pair (its generalised
//              Cauchy point is x itself and the subspace step is the full quasi-Newton step; first direction -g);
//   step       Moré & Thuente's search (MINPACK-2 dcsrch / dcstep, ACM TOMS 20, 1994) with ftol 1e-3, gtol 0.9, xtol 0.1,
//              first trial 1 / |d| in the first iteration and 1 afterwards, at most 20 evaluations; a warning exit of the
//              search is accepted like a converged one; a failed search or an ascent direction drops the memory and
//              restarts from steepest descent, and stops the run if the memory was already empty;
//   update     the pair is skipped when s'y <= eps * (-g'd) * step;
//   stop       max |g| <= pgtol, then (f_old - f) <= factr * eps * max(|f_old|, |f|, 1) -- tested in that order after
//              every accepted step.
// With the same line search the run follows scipy's trajectory (same trial points to rounding, same number of
// evaluations: tests/test_optimizer_cpu.py drives this struct and scipy on the oracle's objective).  The two-loop
// recursion replaces L-BFGS-B's compact representation: the same matrix, different rounding.
// Ask / tell: evaluate f and its gradient at xn, call tell(); repeat until done.  The host optimisers (LbfgsStepper,
// cgp_optimize_batch) and the one-launch device optimiser of short windows (cgp_small.hpp, one lane per parameter) run
// this same state machine; the line search (MtSearch) is the same code in both.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define CGP_HD __host__ __device__
#else
#define CGP_HD
#endif

namespace corenav {

constexpr int LB_N = 16;      // parameters at most (CGP_MAX_THETA = 10; cgp_selftest_lbfgs takes up to 16)
constexpr int LB_M = 10;      // history pairs
constexpr int LB_MAXLS = 20;  // evaluations per line search (scipy: maxls)
constexpr double LB_EPS = 2.220446049250313e-16;
constexpr double LB_STPMAX = 1e10;   // L-BFGS-B: `big` for a problem without bounds
constexpr double LB_INFEASIBLE = 1e300;   // value given to a point whose matrix stays indefinite (oracle/gp_oracle.py: optimize)

// One line search.  start() = the 'START' call of dcsrch, step() = every later call: 0 -> evaluate at the new stp,
// 1 -> converged (sufficient decrease and curvature), 2 -> one of dcsrch's warnings (the caller accepts the point).
struct MtSearch {
  int brackt, stage;
  double finit, ginit, gtest, width, width1, stx, fx, gx, sty, fy, gy, stmin, stmax;

  static CGP_HD double mx(double a, double b) { return a > b ? a : b; }
  static CGP_HD double mn(double a, double b) { return a < b ? a : b; }
  static CGP_HD double ab(double a) { return a < 0 ? -a : a; }

  CGP_HD void start(double f, double g, double stp) {
    brackt = 0;
    stage = 1;
    finit = f;
    ginit = g;
    gtest = 1e-3 * g;
    width = LB_STPMAX;
    width1 = 2.0 * LB_STPMAX;
    stx = sty = 0.0;
    fx = fy = f;
    gx = gy = g;
    stmin = 0.0;
    stmax = stp + 4.0 * stp;
  }

  CGP_HD int step(double f, double g, double &stp) {
    const double gtol = 0.9, xtol = 0.1, stpmin = 0.0, stpmax = LB_STPMAX;
    const double ftest = finit + stp * gtest;
    if (stage == 1 && f <= ftest && g >= 0.0) stage = 2;
    int task = 0;
    if (brackt && (stp <= stmin || stp >= stmax)) task = 2;        // rounding errors prevent progress
    if (brackt && stmax - stmin <= xtol * stmax) task = 2;         // xtol test satisfied
    if (stp == stpmax && f <= ftest && g <= gtest) task = 2;
    if (stp == stpmin && (f > ftest || g >= gtest)) task = 2;
    if (f <= ftest && ab(g) <= gtol * (-ginit)) task = 1;
    if (task) return task;
    if (stage == 1 && f <= fx && f > ftest) {   // the modified function of the first stage
      double fxm = fx - stx * gtest, fym = fy - sty * gtest, gxm = gx - gtest, gym = gy - gtest;
      trial(stx, fxm, gxm, sty, fym, gym, stp, f - stp * gtest, g - gtest);
      fx = fxm + stx * gtest;
      fy = fym + sty * gtest;
      gx = gxm + gtest;
      gy = gym + gtest;
    } else {
      trial(stx, fx, gx, sty, fy, gy, stp, f, g);
    }
    if (brackt) {
      if (ab(sty - stx) >= 0.66 * width1) stp = stx + 0.5 * (sty - stx);
      width1 = width;
      width = ab(sty - stx);
      stmin = mn(stx, sty);
      stmax = mx(stx, sty);
    } else {
      stmin = stp + 1.1 * (stp - stx);
      stmax = stp + 4.0 * (stp - stx);
    }
    stp = mn(mx(stp, stpmin), stpmax);
    if ((brackt && (stp <= stmin || stp >= stmax)) || (brackt && stmax - stmin <= xtol * stmax)) stp = stx;
    return 0;
  }

 private:
  // dcstep: the safeguarded cubic / quadratic trial step and the update of the interval [stx, sty]
  CGP_HD void trial(double &sx, double &fxv, double &dx, double &sy, double &fyv, double &dy, double &stp, double fp, double dp) {
    const double sgnd = dp * (dx / ab(dx));
    double stpf;
    if (fp > fxv) {   // higher value: the minimum is bracketed
      const double theta = 3.0 * (fxv - fp) / (stp - sx) + dx + dp;
      const double s = mx(mx(ab(theta), ab(dx)), ab(dp));
      double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
      if (stp < sx) gamma = -gamma;
      const double r = ((gamma - dx) + theta) / (((gamma - dx) + gamma) + dp);
      const double stpc = sx + r * (stp - sx);
      const double stpq = sx + ((dx / ((fxv - fp) / (stp - sx) + dx)) / 2.0) * (stp - sx);
      stpf = ab(stpc - sx) < ab(stpq - sx) ? stpc : stpc + (stpq - stpc) / 2.0;
      brackt = 1;
    } else if (sgnd < 0.0) {   // lower value, derivatives of opposite sign: bracketed
      const double theta = 3.0 * (fxv - fp) / (stp - sx) + dx + dp;
      const double s = mx(mx(ab(theta), ab(dx)), ab(dp));
      double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
      if (stp > sx) gamma = -gamma;
      const double r = ((gamma - dp) + theta) / (((gamma - dp) + gamma) + dx);
      const double stpc = stp + r * (sx - stp);
      const double stpq = stp + (dp / (dp - dx)) * (sx - stp);
      stpf = ab(stpc - stp) > ab(stpq - stp) ? stpc : stpq;
      brackt = 1;
    } else if (ab(dp) < ab(dx)) {   // lower value, same sign, the derivative shrinks
      const double theta = 3.0 * (fxv - fp) / (stp - sx) + dx + dp;
      const double s = mx(mx(ab(theta), ab(dx)), ab(dp));
      double gamma = s * sqrt(mx(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
      if (stp > sx) gamma = -gamma;
      const double r = ((gamma - dp) + theta) / ((gamma + (dx - dp)) + gamma);
      double stpc;
      if (r < 0.0 && gamma != 0.0) stpc = stp + r * (sx - stp);
      else stpc = stp > sx ? stmax : stmin;
      const double stpq = stp + (dp / (dp - dx)) * (sx - stp);
      if (brackt) {
        stpf = ab(stpc - stp) < ab(stpq - stp) ? stpc : stpq;
        stpf = stp > sx ? mn(stp + 0.66 * (sy - stp), stpf) : mx(stp + 0.66 * (sy - stp), stpf);
      } else {
        stpf = ab(stpc - stp) > ab(stpq - stp) ? stpc : stpq;
        stpf = mx(stmin, mn(stmax, stpf));
      }
    } else {   // lower value, same sign, the derivative does not shrink
      if (brackt) {
        const double theta = 3.0 * (fp - fyv) / (sy - stp) + dy + dp;
        const double s = mx(mx(ab(theta), ab(dy)), ab(dp));
        double gamma = s * sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
        if (stp > sy) gamma = -gamma;
        const double r = ((gamma - dp) + theta) / (((gamma - dp) + gamma) + dy);
        stpf = stp + r * (sy - stp);
      } else {
        stpf = stp > sx ? stmax : stmin;
      }
    }
    if (fp > fxv) {
      sy = stp;
      fyv = fp;
      dy = dp;
    } else {
      if (sgnd < 0.0) {
        sy = sx;
        fyv = fxv;
        dy = dx;
      }
      sx = stp;
      fxv = fp;
      dx = dp;
    }
    stp = stpf;
  }
};

struct LbfgsCore {
  int n, max_evals, hist, ls, evals, iters, status;  // status: 0 converged (gradient), 1 converged (function decrease), 2 max evals, 3 line search failed
  int first, finished;
  double pgtol, ftol;
  double x[LB_N], g[LB_N], xn[LB_N], gn[LB_N], dir[LB_N];
  double S[LB_M][LB_N], Y[LB_M][LB_N], rho[LB_M];
  double wa[LB_M], ws[LB_N], wy[LB_N];  // work vectors (members, so that on the device they live with the state in LDS, not in scratch memory)
  double f, fn, dg0, t;
  MtSearch mt;

  CGP_HD void init(const double *x0, int n_, int max_evals_, double pgtol_, double factr) {
    n = n_;
    max_evals = max_evals_;
    pgtol = pgtol_;
    ftol = factr * LB_EPS;
    hist = ls = evals = iters = status = 0;
    finished = 0;
    first = 1;
    f = fn = dg0 = t = 0.0;
    mt.start(0.0, 0.0, 0.0);
    for (int i = 0; i < LB_N; ++i) {
      x[i] = xn[i] = i < n ? x0[i] : 0.0;
      g[i] = gn[i] = dir[i] = 0.0;
    }
  }
  CGP_HD bool done() const { return finished != 0; }

  // Feed f(xn) and its gradient.  A non-finite f marks an infeasible point: it enters the search as LB_INFEASIBLE with a
  // zero gradient.
  CGP_HD void tell(double fv, const double *gv) {
    ++evals;
    const bool feas = isfin(fv);
    if (!feas) fv = LB_INFEASIBLE;
    if (first) {
      first = 0;
      f = fv;
      for (int i = 0; i < n; ++i) g[i] = gv[i];
      if (!feas) return finish(3);
      if (gmax(g) <= pgtol) return finish(0);
      return start_iteration();
    }
    fn = fv;
    for (int i = 0; i < n; ++i) gn[i] = feas ? gv[i] : 0.0;
    line_search_step();
  }

 private:
  static CGP_HD bool isfin(double v) { return __builtin_isfinite(v); }
  static CGP_HD double amax(double a, double b) { return a > b ? a : b; }
  static CGP_HD double amin(double a, double b) { return a < b ? a : b; }
  CGP_HD double dot(const double *a, const double *b) const {
    double s = 0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
  }
  CGP_HD double gmax(const double *a) const {
    double s = 0;
    for (int i = 0; i < n; ++i) s = amax(s, fabs(a[i]));
    return s;
  }
  CGP_HD void finish(int st) {
    status = st;
    finished = 1;
    for (int i = 0; i < n; ++i) xn[i] = x[i];
  }
  CGP_HD void set_trial(double tt) {
    t = tt;
    for (int j = 0; j < n; ++j) xn[j] = x[j] + tt * dir[j];
  }

  CGP_HD void start_iteration() {
    for (;;) {
      if (evals >= max_evals) return finish(2);
      for (int j = 0; j < n; ++j) dir[j] = g[j];  // two-loop recursion
      const int k = hist;
      double *a = wa;
      for (int i = k - 1; i >= 0; --i) {
        a[i] = rho[i] * dot(S[i], dir);
        for (int j = 0; j < n; ++j) dir[j] -= a[i] * Y[i][j];
      }
      if (k > 0) {
        const double gam = dot(S[k - 1], Y[k - 1]) / dot(Y[k - 1], Y[k - 1]);
        for (int j = 0; j < n; ++j) dir[j] *= gam;
      }
      for (int i = 0; i < k; ++i) {
        const double be = rho[i] * dot(Y[i], dir);
        for (int j = 0; j < n; ++j) dir[j] += S[i][j] * (a[i] - be);
      }
      for (int j = 0; j < n; ++j) dir[j] = -dir[j];
      dg0 = dot(g, dir);
      if (dg0 < 0) break;
      if (hist == 0) return finish(3);   // -g is not a descent direction: nothing left to try
      hist = 0;                          // ascent direction: drop the memory, restart from steepest descent
    }
    const double dnorm = sqrt(dot(dir, dir));
    const double t0 = iters == 0 ? amin(1.0 / dnorm, LB_STPMAX) : 1.0;
    ls = 1;
    mt.start(f, dg0, t0);
    set_trial(t0);
  }

  CGP_HD void line_search_step() {
    const double gd = dot(gn, dir);
    double tt = t;
    const int task = mt.step(fn, gd, tt);
    if (task == 0) {
      if (ls >= LB_MAXLS || evals >= max_evals) {   // the search failed: back to x, without memory if there was any
        if (evals >= max_evals) return finish(2);
        if (hist == 0) return finish(3);
        hist = 0;
        return start_iteration();
      }
      ++ls;
      return set_trial(tt);
    }
    // accept the step
    double *s = ws, *yv = wy;
    for (int j = 0; j < n; ++j) {
      s[j] = xn[j] - x[j];
      yv[j] = gn[j] - g[j];
    }
    const double sy = dot(s, yv), fold = f;
    for (int j = 0; j < n; ++j) {
      x[j] = xn[j];
      g[j] = gn[j];
    }
    f = fn;
    ++iters;
    if (gmax(g) <= pgtol) return finish(0);
    if ((fold - f) <= ftol * amax(amax(fabs(fold), fabs(f)), 1.0)) return finish(1);
    if (sy > LB_EPS * (-dg0 * t)) {
      if (hist == LB_M) {  // drop the oldest pair
        for (int i = 1; i < LB_M; ++i) {
          for (int j = 0; j < n; ++j) {
            S[i - 1][j] = S[i][j];
            Y[i - 1][j] = Y[i][j];
          }
          rho[i - 1] = rho[i];
        }
        --hist;
      }
      for (int j = 0; j < n; ++j) {
        S[hist][j] = s[j];
        Y[hist][j] = yv[j];
      }
      rho[hist] = 1.0 / sy;
      ++hist;
    }
    start_iteration();
  }
};

}  // namespace corenav
