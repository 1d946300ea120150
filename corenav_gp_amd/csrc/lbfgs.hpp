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

#include "lbfgs_core.hpp"

namespace corenav {

struct LbfgsResult {
  double f = 0.0;
  int evals = 0, iters = 0;
  int status = 0;  // 0 converged (gradient), 1 converged (function decrease), 2 max evals, 3 line search failed
};

// vector-facing wrapper of LbfgsCore (lbfgs_core.hpp: the state machine itself, shared with the device optimiser)
class LbfgsStepper {
 public:
  LbfgsStepper(const std::vector<double> &x0, int max_evals, double pgtol, double factr) : xn_(x0), x_(x0) {
    c_.init(x0.data(), (int)x0.size(), max_evals, pgtol, factr);
  }

  bool done() const { return c_.done(); }
  const std::vector<double> &trial() const { return xn_; }  // the point to evaluate next
  const std::vector<double> &best() const { return x_; }
  LbfgsResult result() const {
    LbfgsResult r;
    r.f = c_.f;
    r.evals = c_.evals;
    r.iters = c_.iters;
    r.status = c_.status;
    return r;
  }

  // Feed f(trial()) and its gradient.  Non-finite f marks an infeasible point.
  void tell(double fv, const std::vector<double> &gv) {
    c_.tell(fv, gv.data());
    std::copy(c_.xn, c_.xn + c_.n, xn_.begin());
    std::copy(c_.x, c_.x + c_.n, x_.begin());
  }

 private:
  LbfgsCore c_;
  std::vector<double> xn_, x_;
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
