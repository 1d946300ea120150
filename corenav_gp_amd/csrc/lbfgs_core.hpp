// lbfgs_core.hpp -- the L-BFGS state machine of lbfgs.hpp as ONE fixed-size struct that compiles for the host and for
// the device: unconstrained limited-memory BFGS (m = 10) with a strong-Wolfe line search, the role
// scipy.optimize.fmin_l_bfgs_b (no bounds) plays under GPy's `m.optimize()`
// (core_navigation/script/gp_slip_node.py:36; paramz 'lbfgsb': factr 1e7, pgtol 1e-5, maxfun 1000).
// Ask / tell: evaluate f and its gradient at xn, call tell(); repeat until done.  The host optimisers
// (LbfgsStepper, cgp_optimize_batch) and the one-launch device optimiser of short windows (cgp_small.hpp: lane 0 of the
// window's workgroup) run this same code.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define CGP_HD __host__ __device__
#else
#define CGP_HD
#endif

namespace corenav {

constexpr int LB_N = 16;  // parameters at most (CGP_MAX_THETA = 10; cgp_selftest_lbfgs takes up to 16)
constexpr int LB_M = 10;  // history pairs

struct LbfgsCore {
  int n, max_evals, hist, ls, evals, iters, status;  // status: 0 converged (gradient), 1 converged (function decrease), 2 max evals, 3 line search failed
  int have_hi, first, finished;
  double pgtol, ftol;
  double x[LB_N], g[LB_N], xn[LB_N], gn[LB_N], dir[LB_N];
  double S[LB_M][LB_N], Y[LB_M][LB_N], rho[LB_M];
  double wa[LB_M], ws[LB_N], wy[LB_N];  // work vectors (members, so that on the device they live with the state in LDS, not in scratch memory)
  double f, fn, dg0, t, t_lo, f_lo, dg_lo, t_hi, f_hi, t_prev, f_prev;

  CGP_HD void init(const double *x0, int n_, int max_evals_, double pgtol_, double factr) {
    n = n_;
    max_evals = max_evals_;
    pgtol = pgtol_;
    ftol = factr * 2.220446049250313e-16;
    hist = ls = evals = iters = status = 0;
    have_hi = finished = 0;
    first = 1;
    f = fn = dg0 = t = t_lo = f_lo = dg_lo = t_hi = f_hi = t_prev = f_prev = 0.0;
    for (int i = 0; i < LB_N; ++i) {
      x[i] = xn[i] = i < n ? x0[i] : 0.0;
      g[i] = gn[i] = dir[i] = 0.0;
    }
  }
  CGP_HD bool done() const { return finished != 0; }

  // Feed f(xn) and its gradient.  Non-finite f marks an infeasible point.
  CGP_HD void tell(double fv, const double *gv) {
    ++evals;
    if (!isfin(fv)) fv = INFINITY;
    if (first) {
      first = 0;
      f = fv;
      for (int i = 0; i < n; ++i) g[i] = gv[i];
      if (!isfin(f)) return finish(3);
      return start_iteration();
    }
    fn = fv;
    for (int i = 0; i < n; ++i) gn[i] = gv[i];
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
    if (gmax(g) <= pgtol) return finish(0);
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
    if (!(dg0 < 0)) {  // not a descent direction: restart from steepest descent
      hist = 0;
      for (int j = 0; j < n; ++j) dir[j] = -g[j];
      dg0 = dot(g, dir);
    }
    // line search state (Nocedal & Wright alg. 3.5 / 3.6, c1 = 1e-4, c2 = 0.9)
    t_lo = 0;
    f_lo = f;
    dg_lo = dg0;
    t_hi = f_hi = 0;
    have_hi = 0;
    t_prev = 0;
    f_prev = f;
    ls = 0;
    set_trial(iters == 0 ? amin(1.0, 1.0 / amax(gmax(g), 1e-300)) : 1.0);
  }

  CGP_HD void line_search_step() {
    const double c1 = 1e-4, c2 = 0.9;
    const double tt = t, fnv = fn;
    const double dgn = isfin(fnv) ? dot(gn, dir) : 0.0;
    bool ok = false, give_up = false;
    if (!have_hi) {
      if (fnv > f + c1 * tt * dg0 || (ls > 0 && fnv >= f_prev)) {
        t_lo = t_prev;
        f_lo = f_prev;
        t_hi = tt;
        f_hi = fnv;
        have_hi = 1;
      } else if (fabs(dgn) <= -c2 * dg0) {
        ok = true;
      } else if (dgn >= 0) {
        t_hi = t_prev;
        f_hi = f_prev;
        t_lo = tt;
        f_lo = fnv;
        dg_lo = dgn;
        have_hi = 1;
      } else {
        t_prev = tt;
        f_prev = fnv;
        dg_lo = dgn;
        ++ls;
        if (ls >= 30 || evals >= max_evals) give_up = true;
        else return set_trial(2.0 * tt);
      }
    } else {
      if (fnv > f + c1 * tt * dg0 || fnv >= f_lo) {
        t_hi = tt;
        f_hi = fnv;
      } else {
        if (fabs(dgn) <= -c2 * dg0) ok = true;
        else {
          if (dgn * (t_hi - t_lo) >= 0) {
            t_hi = t_lo;
            f_hi = f_lo;
          }
          t_lo = tt;
          f_lo = fnv;
          dg_lo = dgn;
        }
      }
    }
    if (!ok && !give_up) {
      ++ls;
      const double lo = amin(t_lo, t_hi), hi = amax(t_lo, t_hi);
      if (ls >= 30 || evals >= max_evals || fabs(hi - lo) < 1e-16 * amax(1.0, fabs(lo))) give_up = true;
      else {
        const double dt = t_hi - t_lo;  // quadratic interpolation, safeguarded by bisection
        double tq = t_lo - 0.5 * dg_lo * dt * dt / (f_hi - f_lo - dg_lo * dt);
        if (!isfin(tq) || tq <= lo + 0.1 * (hi - lo) || tq >= hi - 0.1 * (hi - lo)) tq = 0.5 * (lo + hi);
        return set_trial(tq);
      }
    }
    if (!ok) {  // accept a sufficient-decrease point if the last trial is one, else stop
      if (isfin(fnv) && fnv <= f + c1 * tt * dg0 && fnv < f) ok = true;
      else return finish(evals >= max_evals ? 2 : 3);
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
    f = fnv;
    ++iters;
    if (sy > 1e-10 * dot(yv, yv)) {
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
    if ((fold - f) <= ftol * amax(amax(fabs(fold), fabs(f)), 1.0)) return finish(1);
    start_iteration();
  }
};

}  // namespace corenav
