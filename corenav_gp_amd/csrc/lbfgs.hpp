// lbfgs.hpp -- unconstrained limited-memory BFGS (m = 10) with a strong-Wolfe line search, the
// role scipy.optimize.fmin_l_bfgs_b (no bounds) plays under GPy's `m.optimize()`
// (core_navigation/script/gp_slip_node.py:36; paramz 'lbfgsb': factr 1e7, pgtol 1e-5, maxfun 1000).
// Written as an ask/tell stepper so that many independent problems can share one batched objective
// evaluation per round (cgp_optimize_batch); lbfgs_minimize drives a single stepper.
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

class LbfgsStepper {
 public:
  LbfgsStepper(const std::vector<double> &x0, int max_evals, double pgtol, double factr)
      : n_((int)x0.size()), max_evals_(max_evals), pgtol_(pgtol), ftol_(factr * 2.220446049250313e-16), x_(x0),
        g_(x0.size()), xn_(x0), gn_(x0.size()), dir_(x0.size()) {}

  bool done() const { return done_; }
  const std::vector<double> &trial() const { return xn_; }  // the point to evaluate next
  const std::vector<double> &best() const { return x_; }
  LbfgsResult result() const {
    LbfgsResult r = res_;
    r.f = f_;
    return r;
  }

  // Feed f(trial()) and its gradient.  Non-finite f marks an infeasible point.
  void tell(double fv, const std::vector<double> &gv) {
    ++res_.evals;
    if (!std::isfinite(fv)) fv = INFINITY;
    if (first_) {
      first_ = false;
      f_ = fv;
      g_ = gv;
      if (!std::isfinite(f_)) return finish(3);
      return start_iteration();
    }
    fn_ = fv;
    gn_ = gv;
    line_search_step();
  }

 private:
  double dot(const std::vector<double> &a, const std::vector<double> &b) const {
    double s = 0;
    for (int i = 0; i < n_; ++i) s += a[i] * b[i];
    return s;
  }
  static double gmax(const std::vector<double> &a) {
    double s = 0;
    for (double v : a) s = std::max(s, std::fabs(v));
    return s;
  }
  void finish(int status) {
    res_.status = status;
    done_ = true;
    xn_ = x_;
  }
  void set_trial(double t) {
    t_ = t;
    for (int j = 0; j < n_; ++j) xn_[j] = x_[j] + t * dir_[j];
  }

  void start_iteration() {
    if (gmax(g_) <= pgtol_) return finish(0);
    if (res_.evals >= max_evals_) return finish(2);
    dir_ = g_;  // two-loop recursion
    const int k = (int)S_.size();
    std::vector<double> a(k);
    for (int i = k - 1; i >= 0; --i) {
      a[i] = rho_[i] * dot(S_[i], dir_);
      for (int j = 0; j < n_; ++j) dir_[j] -= a[i] * Y_[i][j];
    }
    if (k > 0) {
      const double gam = dot(S_[k - 1], Y_[k - 1]) / dot(Y_[k - 1], Y_[k - 1]);
      for (double &v : dir_) v *= gam;
    }
    for (int i = 0; i < k; ++i) {
      const double be = rho_[i] * dot(Y_[i], dir_);
      for (int j = 0; j < n_; ++j) dir_[j] += S_[i][j] * (a[i] - be);
    }
    for (double &v : dir_) v = -v;
    dg0_ = dot(g_, dir_);
    if (!(dg0_ < 0)) {  // not a descent direction: restart from steepest descent
      S_.clear();
      Y_.clear();
      rho_.clear();
      for (int j = 0; j < n_; ++j) dir_[j] = -g_[j];
      dg0_ = dot(g_, dir_);
    }
    // line search state (Nocedal & Wright alg. 3.5 / 3.6, c1 = 1e-4, c2 = 0.9)
    t_lo_ = 0;
    f_lo_ = f_;
    dg_lo_ = dg0_;
    t_hi_ = f_hi_ = 0;
    have_hi_ = false;
    t_prev_ = 0;
    f_prev_ = f_;
    ls_ = 0;
    set_trial(res_.iters == 0 ? std::min(1.0, 1.0 / std::max(gmax(g_), 1e-300)) : 1.0);
  }

  void line_search_step() {
    const double c1 = 1e-4, c2 = 0.9;
    const double t = t_, fn = fn_;
    const double dgn = std::isfinite(fn) ? dot(gn_, dir_) : 0.0;
    bool ok = false, give_up = false;
    if (!have_hi_) {
      if (fn > f_ + c1 * t * dg0_ || (ls_ > 0 && fn >= f_prev_)) {
        t_lo_ = t_prev_;
        f_lo_ = f_prev_;
        t_hi_ = t;
        f_hi_ = fn;
        have_hi_ = true;
      } else if (std::fabs(dgn) <= -c2 * dg0_) {
        ok = true;
      } else if (dgn >= 0) {
        t_hi_ = t_prev_;
        f_hi_ = f_prev_;
        t_lo_ = t;
        f_lo_ = fn;
        dg_lo_ = dgn;
        have_hi_ = true;
      } else {
        t_prev_ = t;
        f_prev_ = fn;
        dg_lo_ = dgn;
        ++ls_;
        if (ls_ >= 30 || res_.evals >= max_evals_) give_up = true;
        else return set_trial(2.0 * t);
      }
    } else {
      if (fn > f_ + c1 * t * dg0_ || fn >= f_lo_) {
        t_hi_ = t;
        f_hi_ = fn;
      } else {
        if (std::fabs(dgn) <= -c2 * dg0_) ok = true;
        else {
          if (dgn * (t_hi_ - t_lo_) >= 0) {
            t_hi_ = t_lo_;
            f_hi_ = f_lo_;
          }
          t_lo_ = t;
          f_lo_ = fn;
          dg_lo_ = dgn;
        }
      }
    }
    if (!ok && !give_up) {
      ++ls_;
      const double lo = std::min(t_lo_, t_hi_), hi = std::max(t_lo_, t_hi_);
      if (ls_ >= 30 || res_.evals >= max_evals_ || std::fabs(hi - lo) < 1e-16 * std::max(1.0, std::fabs(lo))) give_up = true;
      else {
        const double dt = t_hi_ - t_lo_;  // quadratic interpolation, safeguarded by bisection
        double tq = t_lo_ - 0.5 * dg_lo_ * dt * dt / (f_hi_ - f_lo_ - dg_lo_ * dt);
        if (!std::isfinite(tq) || tq <= lo + 0.1 * (hi - lo) || tq >= hi - 0.1 * (hi - lo)) tq = 0.5 * (lo + hi);
        return set_trial(tq);
      }
    }
    if (!ok) {  // accept a sufficient-decrease point if the last trial is one, else stop
      if (std::isfinite(fn) && fn <= f_ + c1 * t * dg0_ && fn < f_) ok = true;
      else return finish(res_.evals >= max_evals_ ? 2 : 3);
    }
    // accept the step
    std::vector<double> s(n_), yv(n_);
    for (int j = 0; j < n_; ++j) {
      s[j] = xn_[j] - x_[j];
      yv[j] = gn_[j] - g_[j];
    }
    const double sy = dot(s, yv), fold = f_;
    x_ = xn_;
    g_ = gn_;
    f_ = fn;
    ++res_.iters;
    if (sy > 1e-10 * dot(yv, yv)) {
      if ((int)S_.size() == 10) {
        S_.erase(S_.begin());
        Y_.erase(Y_.begin());
        rho_.erase(rho_.begin());
      }
      S_.push_back(s);
      Y_.push_back(yv);
      rho_.push_back(1.0 / sy);
    }
    if ((fold - f_) <= ftol_ * std::max({std::fabs(fold), std::fabs(f_), 1.0})) return finish(1);
    start_iteration();
  }

  int n_, max_evals_;
  double pgtol_, ftol_;
  std::vector<double> x_, g_, xn_, gn_, dir_;
  std::vector<std::vector<double>> S_, Y_;
  std::vector<double> rho_;
  double f_ = 0, fn_ = 0, dg0_ = 0, t_ = 0;
  double t_lo_ = 0, f_lo_ = 0, dg_lo_ = 0, t_hi_ = 0, f_hi_ = 0, t_prev_ = 0, f_prev_ = 0;
  bool have_hi_ = false, first_ = true, done_ = false;
  int ls_ = 0;
  LbfgsResult res_;
};

// fg(x, grad) -> f.  Returns non-finite f to signal an infeasible point (treated as +inf).
inline LbfgsResult lbfgs_minimize(const std::function<double(const std::vector<double> &, std::vector<double> &)> &fg,
                                  std::vector<double> &x, int max_evals = 1000, double pgtol = 1e-5,
                                  double factr = 1e7) {
  LbfgsStepper st(x, max_evals, pgtol, factr);
  std::vector<double> g(x.size());
  while (!st.done()) {
    const double f = fg(st.trial(), g);
    st.tell(f, g);
  }
  x = st.best();
  return st.result();
}

}  // namespace corenav
