// lbfgs.hpp -- unconstrained limited-memory BFGS (m = 10) with a strong-Wolfe line search, the
// role scipy.optimize.fmin_l_bfgs_b (no bounds) plays under GPy's `m.optimize()`
// (core_navigation/script/gp_slip_node.py:36; paramz 'lbfgsb': factr 1e7, pgtol 1e-5, maxfun 1000).
#pragma once
#include <algorithm>
#include <cmath>
#include <functional>
#include <vector>

namespace corenav {

struct LbfgsResult {
  double f = 0.0;
  int evals = 0, iters = 0;
  int status = 0;  // 0 converged (gradient), 1 converged (function decrease), 2 max evals, 3 line search failed
};

// fg(x, grad) -> f.  Returns non-finite f to signal an infeasible point (treated as +inf).
inline LbfgsResult lbfgs_minimize(const std::function<double(const std::vector<double> &, std::vector<double> &)> &fg,
                                  std::vector<double> &x, int max_evals = 1000, double pgtol = 1e-5,
                                  double factr = 1e7) {
  const int n = (int)x.size(), m = 10;
  const double ftol = factr * 2.220446049250313e-16;
  LbfgsResult res;
  std::vector<double> g(n), xn(n), gn(n), dir(n);
  std::vector<std::vector<double>> S, Y;
  std::vector<double> rho;
  double f = fg(x, g);
  res.evals = 1;
  auto dot = [&](const std::vector<double> &a, const std::vector<double> &b) {
    double s = 0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
  };
  auto gmax = [&](const std::vector<double> &a) {
    double s = 0;
    for (double v : a) s = std::max(s, std::fabs(v));
    return s;
  };
  if (!std::isfinite(f)) {
    res.f = f;
    res.status = 3;
    return res;
  }
  while (true) {
    if (gmax(g) <= pgtol) { res.status = 0; break; }
    if (res.evals >= max_evals) { res.status = 2; break; }
    // two-loop recursion
    dir = g;
    const int k = (int)S.size();
    std::vector<double> a(k);
    for (int i = k - 1; i >= 0; --i) {
      a[i] = rho[i] * dot(S[i], dir);
      for (int j = 0; j < n; ++j) dir[j] -= a[i] * Y[i][j];
    }
    if (k > 0) {
      const double gam = dot(S[k - 1], Y[k - 1]) / dot(Y[k - 1], Y[k - 1]);
      for (double &v : dir) v *= gam;
    }
    for (int i = 0; i < k; ++i) {
      const double be = rho[i] * dot(Y[i], dir);
      for (int j = 0; j < n; ++j) dir[j] += S[i][j] * (a[i] - be);
    }
    for (double &v : dir) v = -v;
    double dg0 = dot(g, dir);
    if (!(dg0 < 0)) {  // not a descent direction: restart from steepest descent
      S.clear(); Y.clear(); rho.clear();
      for (int j = 0; j < n; ++j) dir[j] = -g[j];
      dg0 = dot(g, dir);
    }
    // line search: bracketing + zoom (Nocedal & Wright alg. 3.5 / 3.6), c1 = 1e-4, c2 = 0.9
    const double c1 = 1e-4, c2 = 0.9;
    double t = (res.iters == 0) ? std::min(1.0, 1.0 / std::max(gmax(g), 1e-300)) : 1.0;
    double t_lo = 0, f_lo = f, dg_lo = dg0, t_hi = 0, f_hi = 0;
    bool have_hi = false, ok = false;
    double fn = f;
    auto eval = [&](double step) {
      for (int j = 0; j < n; ++j) xn[j] = x[j] + step * dir[j];
      double v = fg(xn, gn);
      ++res.evals;
      if (!std::isfinite(v)) v = INFINITY;
      return v;
    };
    double t_prev = 0, f_prev = f;
    for (int ls = 0; ls < 30 && res.evals < max_evals; ++ls) {
      fn = eval(t);
      const double dgn = std::isfinite(fn) ? dot(gn, dir) : 0.0;
      if (!have_hi) {
        if (fn > f + c1 * t * dg0 || (ls > 0 && fn >= f_prev)) {
          t_lo = t_prev; f_lo = f_prev; t_hi = t; f_hi = fn; have_hi = true;
          // dg_lo stays the slope at t_prev (recomputed below if needed)
        } else if (std::fabs(dgn) <= -c2 * dg0) { ok = true; break; }
        else if (dgn >= 0) { t_hi = t_prev; f_hi = f_prev; t_lo = t; f_lo = fn; dg_lo = dgn; have_hi = true; }
        else { t_prev = t; f_prev = fn; dg_lo = dgn; t *= 2.0; continue; }
      } else {
        if (fn > f + c1 * t * dg0 || fn >= f_lo) { t_hi = t; f_hi = fn; }
        else {
          if (std::fabs(dgn) <= -c2 * dg0) { ok = true; break; }
          if (dgn * (t_hi - t_lo) >= 0) { t_hi = t_lo; f_hi = f_lo; }
          t_lo = t; f_lo = fn; dg_lo = dgn;
        }
      }
      // next trial inside the bracket: quadratic interpolation, safeguarded by bisection
      const double dt = t_hi - t_lo;
      double tq = t_lo - 0.5 * dg_lo * dt * dt / (f_hi - f_lo - dg_lo * dt);
      const double lo = std::min(t_lo, t_hi), hi = std::max(t_lo, t_hi);
      if (!std::isfinite(tq) || tq <= lo + 0.1 * (hi - lo) || tq >= hi - 0.1 * (hi - lo)) tq = 0.5 * (lo + hi);
      t = tq;
      if (std::fabs(hi - lo) < 1e-16 * std::max(1.0, std::fabs(lo))) break;
    }
    if (!ok) {
      // accept the best sufficient-decrease point found, else stop
      if (std::isfinite(fn) && fn <= f + c1 * t * dg0 && fn < f) ok = true;
      else if (have_hi && t_lo > 0 && f_lo < f) {
        fn = eval(t_lo);
        ok = std::isfinite(fn) && fn < f;
      }
      if (!ok) { res.status = 3; break; }
    }
    // update
    std::vector<double> s(n), yv(n);
    for (int j = 0; j < n; ++j) { s[j] = xn[j] - x[j]; yv[j] = gn[j] - g[j]; }
    const double sy = dot(s, yv);
    const double fold = f;
    x = xn; g = gn; f = fn;
    ++res.iters;
    if (sy > 1e-10 * dot(yv, yv)) {
      if ((int)S.size() == m) { S.erase(S.begin()); Y.erase(Y.begin()); rho.erase(rho.begin()); }
      S.push_back(s); Y.push_back(yv); rho.push_back(1.0 / sy);
    }
    if ((fold - f) <= ftol * std::max({std::fabs(fold), std::fabs(f), 1.0})) { res.status = 1; break; }
  }
  res.f = f;
  return res;
}

}  // namespace corenav
