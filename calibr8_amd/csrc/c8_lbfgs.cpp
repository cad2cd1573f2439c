// c8_lbfgs.cpp -- bound-constrained limited-memory BFGS for the outer calibration loop.
//
// The reference drives its inverse problems with ROL: a line-search step with a limited-memory BFGS secant
// (storage 20) on the canonical variables, bounds [-1, 1], status tests on iterations / gradient norm / step
// norm and a cap on the function evaluations per line search (main_inverse.cpp:21-28, :83-120); its Python
// package uses SciPy's L-BFGS-B for the same job.  Neither library is part of this build; this is a small
// projected L-BFGS with the same controls.  It does not reproduce ROL's iterates, only its contract: minimise
// f(x) over lo <= x <= hi from gradients, stop on the same three tests.  Pure host C++.
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#include "../../include/c8.h"

namespace {

struct Pair { std::vector<double> s, y; double rho; };

double dot(std::vector<double> const& a, std::vector<double> const& b) {
  double s = 0.;
  for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
  return s;
}

}  // namespace

extern "C" int c8_lbfgs_minimize(int n, double* x, const double* lo, const double* hi, c8_objective_fn fn, void* user,
                                 const c8_lbfgs_opts* opts, c8_lbfgs_result* res) {
  if (n <= 0 || !x || !fn || !opts) return C8_ERR_ARG;
  int const memory = opts->memory > 0 ? opts->memory : 20;
  int const max_ls = opts->max_ls_evals > 0 ? opts->max_ls_evals : 5;
  auto L = [&](int i) { return lo ? lo[i] : -std::numeric_limits<double>::infinity(); };
  auto H = [&](int i) { return hi ? hi[i] : std::numeric_limits<double>::infinity(); };
  auto clip = [&](std::vector<double>& v) {
    for (int i = 0; i < n; ++i) v[i] = std::min(std::max(v[i], L(i)), H(i));
  };
  std::vector<double> xc(x, x + n), g(n), xn(n), gn(n), d(n), pg(n);
  clip(xc);
  double f = 0.;
  int evals = 1, iters = 0, status = C8_LBFGS_ITERATION_LIMIT;
  if (fn(user, n, xc.data(), &f, g.data()) != 0) return C8_ERR_ARG;  // the starting point must be evaluable
  std::vector<Pair> mem;
  // a variable is held at its bound while the gradient pushes it outwards
  auto active = [&](int i, std::vector<double> const& xx, std::vector<double> const& gg) {
    return (xx[i] <= L(i) && gg[i] > 0.) || (xx[i] >= H(i) && gg[i] < 0.);
  };
  auto projected_gradient = [&]() {
    for (int i = 0; i < n; ++i) pg[i] = active(i, xc, g) ? 0. : g[i];
    return std::sqrt(dot(pg, pg));
  };
  double pgn = projected_gradient();
  while (iters < opts->max_iters) {
    if (pgn < opts->grad_tol) { status = C8_LBFGS_GRADIENT_TOL; break; }
    // two-loop recursion on the projected gradient
    std::vector<double> q(pg), alpha(mem.size());
    for (int k = (int)mem.size() - 1; k >= 0; --k) {
      alpha[k] = mem[k].rho * dot(mem[k].s, q);
      for (int i = 0; i < n; ++i) q[i] -= alpha[k] * mem[k].y[i];
    }
    double gamma = 1.;
    if (!mem.empty()) gamma = dot(mem.back().s, mem.back().y) / dot(mem.back().y, mem.back().y);
    for (int i = 0; i < n; ++i) q[i] *= gamma;
    for (size_t k = 0; k < mem.size(); ++k) {
      double const beta = mem[k].rho * dot(mem[k].y, q);
      for (int i = 0; i < n; ++i) q[i] += (alpha[k] - beta) * mem[k].s[i];
    }
    for (int i = 0; i < n; ++i) d[i] = active(i, xc, g) ? 0. : -q[i];
    double slope = dot(g, d);
    if (!(slope < 0.)) {  // not a descent direction: steepest descent on the free variables
      for (int i = 0; i < n; ++i) d[i] = -pg[i];
      slope = -pgn * pgn;
      mem.clear();
    }
    // backtracking along the projected path, Armijo on the actual displacement
    double t = mem.empty() ? std::min(1., 1. / pgn) : 1.;
    bool accepted = false;
    double fn_new = f;
    for (int ls = 0; ls < max_ls; ++ls, t *= 0.5) {
      for (int i = 0; i < n; ++i) xn[i] = xc[i] + t * d[i];
      clip(xn);
      double dec = 0.;
      for (int i = 0; i < n; ++i) dec += g[i] * (xn[i] - xc[i]);
      ++evals;
      if (fn(user, n, xn.data(), &fn_new, gn.data()) != 0) continue;  // evaluation failed there: shorter step
      if (fn_new <= f + 1e-4 * dec) { accepted = true; break; }
    }
    ++iters;
    if (!accepted) { status = C8_LBFGS_LINE_SEARCH_FAILED; break; }
    Pair p;
    p.s.resize(n);
    p.y.resize(n);
    for (int i = 0; i < n; ++i) { p.s[i] = xn[i] - xc[i]; p.y[i] = gn[i] - g[i]; }
    double const sy = dot(p.s, p.y), sn = std::sqrt(dot(p.s, p.s));
    if (sy > 1e-12 * sn * std::sqrt(dot(p.y, p.y))) {
      p.rho = 1. / sy;
      mem.push_back(p);
      if ((int)mem.size() > memory) mem.erase(mem.begin());
    }
    xc = xn;
    g = gn;
    f = fn_new;
    pgn = projected_gradient();
    if (sn < opts->step_tol) { status = C8_LBFGS_STEP_TOL; break; }
  }
  if (status == C8_LBFGS_ITERATION_LIMIT && pgn < opts->grad_tol) status = C8_LBFGS_GRADIENT_TOL;
  std::copy(xc.begin(), xc.end(), x);
  if (res) {
    res->iters = iters;
    res->evals = evals;
    res->status = status;
    res->f = f;
    res->projected_gradient_norm = pgn;
  }
  return C8_OK;
}
