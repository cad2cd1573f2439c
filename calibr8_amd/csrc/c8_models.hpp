// c8_models.hpp -- device-side residual plug-ins.
//
// Same plug-in surface as the reference, templated on the scalar T in
// {double, Dual}, stateless apart from the per-point registers they own:
//
//   LocalResidual concept  (local_residual.hpp:62-158)
//     NLOC, NPARAMS, FINITE_DEF, HAS_LOCAL, name()
//     initial_guess(g)            first half of solve_nonlinear
//     evaluate(g, force, path)    fills R[], returns ELASTIC/PLASTIC
//     cauchy / dev_cauchy / hydro_cauchy / pressure_scale_factor
//   GlobalResidual concept (global_residual.hpp:125-130)
//     Mechanics::flux(local, g, ...)  the integrand of the stabilised mixed
//     weak form as point fluxes: R(i,n,eq) += [V(i,eq) N_n + G(i,eq,:).grad N_n] w dv
//
// A new constitutive model is one struct implementing the LocalResidual
// concept plus one line in C8_FOR_EACH_MODEL (c8_kernels.hip).
//
// Semantics that must not be "simplified" (SURVEY.md section 10): the yield function is
// scaled by val(mu), not mu (small_J2.cpp:208, hyper_J2.cpp:264); the branch
// test is f > tol || |f| < tol (small_J2.cpp:215); n = s/|s| is 0/0 at zero
// strain and may only be read on the plastic branch.
#pragma once

#include "c8_math.hpp"

namespace c8 {

enum { C8_ELASTIC_PATH = 0, C8_PLASTIC_PATH = 1 };


C8_HD void set_val(double& x, double v) { x = v; }
C8_HD void set_val(Dual& x, double v) { x.v = v; }  // keeps the seeding (local_residual.cpp:293-296)

// interpolated global state at a quadrature point: the GlobalResidual accessors
// scalar_x(1), grad_scalar_x(1), vector_x(0), grad_vector_x(0), grad_vector_x_prev(0)
template <class T> struct PointState {
  T u[3];
  T p;
  T grad_p[3];
  Tens3<T> grad_u;
  Tens3<T> grad_u_prev;
};

template <class T> C8_HD T compute_mu(T const& E, T const& nu) { return E / (2. * (1. + nu)); }       // material_params.hpp:12
template <class T> C8_HD T compute_kappa(T const& E, T const& nu) { return E / (3. * (1. - 2. * nu)); }  // :20

template <class T> C8_HD Tens3<T> small_strain(Tens3<T> const& g) {
  Tens3<T> const gt = transpose(g);
  return scale(0.5, g + gt);
}

// Point quantities of a local model that depend on the global state and the previous local state only, not on the
// unknowns: a model may compute them once per point (trial) and reuse them over the Newton iterations
// (evaluate(g, tol, trial)).  Models without such quantities use NoTrial.
struct NoTrial {};

// ---- elastic.cpp:76-136 -----------------------------------------------------
template <class T> struct Elastic {
  static constexpr int NLOC = 1, NPARAMS = 4;
  static constexpr bool FINITE_DEF = false, HAS_LOCAL = false;
  // wave kernels: 256-thread workgroups per CU that the register budget is set for (2 -> 256 registers, 1 -> 512),
  // for the two Jacobian kernels and for the local-adjoint / parameter-gradient kernels
  static constexpr int WAVE_BLOCKS_PER_CU = 2, WAVE_BLOCKS_PER_CU_ADJ = 2;
  static constexpr int WAVE_BLOCKS_PER_CU_K4 = 2;  // waves per SIMD of the local-adjoint wave kernel
  static constexpr bool GJ_XLANE_JAC = false, GJ_XLANE_K4 = false;  // pivot-column hand-over of the local solves (gj_solve_cols), as measured
  static constexpr bool NEWTON_MATRIX_IN_LDS = false;  // local Newton of the wave kernel: matrix columns in registers
  using Trial = NoTrial;
  C8_HD Trial trial(PointState<T> const&) const { return {}; }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, Trial const&) { return evaluate(g, abs_tol); }
  T params[NPARAMS];  // E nu cte delta_T
  T xi[NLOC], xi_prev[NLOC], R[NLOC];
  C8_HD static void init_variables(double* xi0) { xi0[0] = 0.; }
  C8_HD void initial_guess(PointState<T> const&) { set_val(xi[0], 0.); }
  C8_HD int evaluate(PointState<T> const&, double, bool = false, int = 0) { return 0; }
  C8_HD Tens3<T> dev_cauchy(PointState<T> const& g) const {
    T const mu = compute_mu(params[0], params[1]);
    Tens3<T> const eps = small_strain(g.grad_u);
    return scale(2. * mu, dev(eps));
  }
  C8_HD Tens3<T> cauchy(PointState<T> const& g) const {
    Tens3<T> s = dev_cauchy(g);
    s.xx = s.xx - g.p; s.yy = s.yy - g.p; s.zz = s.zz - g.p;
    return s;
  }
  C8_HD T hydro_cauchy(PointState<T> const& g) const {
    T const E = params[0], nu = params[1];
    T const kappa = compute_kappa(E, nu);
    return kappa * trace(small_strain(g.grad_u)) - params[2] * params[3] * E / (1. - 2. * nu);
  }
  C8_HD T pressure_scale_factor() const { return compute_kappa(params[0], params[1]); }
};

// ---- isotropic_elastic.cpp (mixed formulation): the local unknown is the Cauchy stress ---------------------
template <class T> C8_HD T compute_lambda(T const& E, T const& nu) { return E * nu / ((1. + nu) * (1. - 2. * nu)); }  // material_params.hpp:28
template <class T> struct IsotropicElastic {
  static constexpr int NLOC = 6, NPARAMS = 2;
  static constexpr bool FINITE_DEF = false, HAS_LOCAL = true;
  static constexpr int WAVE_BLOCKS_PER_CU = 2, WAVE_BLOCKS_PER_CU_ADJ = 2;
  static constexpr int WAVE_BLOCKS_PER_CU_K4 = 2;  // waves per SIMD of the local-adjoint wave kernel
  static constexpr bool GJ_XLANE_JAC = false, GJ_XLANE_K4 = false;  // pivot-column hand-over of the local solves (gj_solve_cols), as measured
  static constexpr bool NEWTON_MATRIX_IN_LDS = false;  // local Newton of the wave kernel: matrix columns in registers
  using Trial = NoTrial;
  C8_HD Trial trial(PointState<T> const&) const { return {}; }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, Trial const&) { return evaluate(g, abs_tol); }
  T params[NPARAMS];  // E nu  (isotropic_elastic.cpp:61-76)
  T xi[NLOC], xi_prev[NLOC], R[NLOC];  // cauchy(00,01,02,11,12,22)
  C8_HD static void init_variables(double* xi0) { C8_UNROLL for (int k = 0; k < NLOC; ++k) xi0[k] = 0.; }
  C8_HD Tens3<T> hooke(PointState<T> const& g) const {  // lambda tr(eps) I + 2 mu eps
    T const mu = compute_mu(params[0], params[1]), lambda = compute_lambda(params[0], params[1]);
    Tens3<T> const eps = small_strain(g.grad_u);
    Tens3<T> s = scale(2. * mu, eps);
    T const lt = lambda * trace(eps);
    s.xx = s.xx + lt; s.yy = s.yy + lt; s.zz = s.zz + lt;
    return s;
  }
  // the reference starts from the exact stress and takes one Newton step (:99-121); from that start the residual
  // vanishes, so the generic Newton loop stops at its first test with the same state
  C8_HD void initial_guess(PointState<T> const& g) {
    T sv[6];
    pack_sym6(hooke(g), sv);
    C8_UNROLL
    for (int k = 0; k < NLOC; ++k) set_val(xi[k], val(sv[k]));
  }
  C8_HD int evaluate(PointState<T> const& g, double, bool = false, int = 0) {  // :128-152
    pack_sym6(sym6(xi) - hooke(g), R);
    return 0;
  }
  C8_HD T hydro_cauchy(PointState<T> const&) const { return (xi[0] + xi[3] + xi[5]) / 3.; }  // :170-181, 3-D
  C8_HD Tens3<T> dev_cauchy(PointState<T> const& g) const {  // :162-168
    Tens3<T> s = sym6(xi);
    T const h = hydro_cauchy(g);
    s.xx = s.xx - h; s.yy = s.yy - h; s.zz = s.zz - h;
    return s;
  }
  C8_HD Tens3<T> cauchy(PointState<T> const& g) const {  // cauchy_mixed :191-197
    Tens3<T> s = dev_cauchy(g);
    s.xx = s.xx - g.p; s.yy = s.yy - g.p; s.zz = s.zz - g.p;
    return s;
  }
  C8_HD T pressure_scale_factor() const { return compute_kappa(params[0], params[1]); }
};

// ---- small_J2.cpp -------------------------------------------------------------
// The reference's class serves 3-D and 2-D meshes (ndims = m_num_dims, eye<T>(ndims), small_J2.cpp:186-187): in 2-D it
// works on 2 x 2 tensors -- the in-plane deviatoric stress without its out-of-plane entry -- with 3 + 1 local unknowns.
template <class T, int DIM> struct SmallJ2Dim {
  static constexpr int NSYM = (DIM == 3) ? 6 : 3;
  static constexpr int NLOC = NSYM + 1, NPARAMS = 6;
  static constexpr bool FINITE_DEF = false, HAS_LOCAL = true;
  static constexpr int WAVE_BLOCKS_PER_CU = 2, WAVE_BLOCKS_PER_CU_ADJ = 2;
  static constexpr int WAVE_BLOCKS_PER_CU_K4 = 3;  // 162 registers: three waves per SIMD (3.1 against 3.5 ms per million elements)
  static constexpr bool GJ_XLANE_JAC = false, GJ_XLANE_K4 = true;  // pivot-column hand-over of the local solves (gj_solve_cols), as measured
  static constexpr bool GJ_XLANE_NEWTON = true;  // K1: DPP hand-over in the Newton solve, LDS in the inverse (11.35 against 11.47 ms per assembly, profiles/README round 2)
  static constexpr bool NEWTON_MATRIX_IN_LDS = false;  // local Newton of the wave kernel: matrix columns in registers
  using Trial = NoTrial;
  C8_HD Trial trial(PointState<T> const&) const { return {}; }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, Trial const&) { return evaluate(g, abs_tol); }
  T params[NPARAMS];  // E nu K Y cte delta_T  (small_J2.cpp:70-75)
  T xi[NLOC], xi_prev[NLOC], R[NLOC];  // pstrain (00,01,02,11,12,22) or (00,01,11), alpha
  C8_HD static void init_variables(double* xi0) { C8_UNROLL for (int k = 0; k < NLOC; ++k) xi0[k] = 0.; }
  C8_HD void initial_guess(PointState<T> const&) {  // :127-135
    C8_UNROLL
    for (int k = 0; k < NLOC; ++k) set_val(xi[k], val(xi_prev[k]));
  }
  C8_HD Tens3<T> dev_cauchy(PointState<T> const& g) const {  // :266-277
    T const mu = compute_mu(params[0], params[1]);
    Tens3<T> const eps = small_strain(g.grad_u);
    Tens3<T> const pstrain = sym_dim<DIM>(xi);
    Tens3<T> const dev_eps = (DIM == 3) ? dev(eps) : minus_s_eye<DIM>(eps, trace(eps) * (1. / 3.));  // eps - tr(eps)/3 I(ndims)
    return scale(2. * mu, dev_eps - pstrain);
  }
  C8_HD Tens3<T> cauchy(PointState<T> const& g) const {  // :253-263
    return minus_s_eye<DIM>(dev_cauchy(g), g.p);
  }
  C8_HD T hydro_cauchy(PointState<T> const& g) const {  // :280-289
    T const E = params[0], nu = params[1];
    T const kappa = compute_kappa(E, nu);
    return kappa * trace(small_strain(g.grad_u)) - params[4] * params[5] * E / (1. - 2. * nu);
  }
  C8_HD T pressure_scale_factor() const { return compute_kappa(params[0], params[1]); }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, bool force_path = false, int path_in = 0) {  // :181-250
    double const sqrt_23 = 0.81649658092772603273;
    double const sqrt_32 = 1.22474487139158904910;
    T const mu = compute_mu(params[0], params[1]);
    T const K = params[2], Y = params[3];
    T const alpha = xi[NSYM], alpha_old = xi_prev[NSYM];
    Tens3<T> const s = dev_cauchy(g);
    T const s_mag = norm(s);
    T const sigma_yield = Y + K * alpha;
    T const f = (s_mag - sqrt_23 * sigma_yield) / val(mu);
    int path;
    if (!force_path) path = (val(f) > abs_tol || fabs(val(f)) < abs_tol) ? C8_PLASTIC_PATH : C8_ELASTIC_PATH;
    else path = path_in;
    if (path == C8_PLASTIC_PATH) {
      T const dgam = sqrt_32 * (alpha - alpha_old);
      T const c = dgam / s_mag;  // dgam * n = (dgam/|s|) s
      Tens3<T> const Rp = sym_dim<DIM>(xi) - sym_dim<DIM>(xi_prev) - scale(c, s);
      pack_sym_dim<DIM>(Rp, R);
      R[NSYM] = f;
    } else {
      C8_UNROLL
      for (int k = 0; k < NLOC; ++k) R[k] = xi[k] - xi_prev[k];
    }
    return path;
  }
  // ---- the closed form of this model's local equations (3-D) ------------------------------------------------------------
  // With linear hardening the return mapping has an explicit solution (radial return): from the trial stress
  // s_tr = 2 mu (dev eps - pstrain_old), dgam = (|s_tr| - sqrt(2/3)(Y + K alpha_old)) / (2 mu + 2K/3), n = s_tr / |s_tr|,
  // pstrain = pstrain_old + dgam n, alpha = alpha_old + sqrt(2/3) dgam -- the point the reference's Newton iteration
  // (small_J2.cpp:122-173) converges to --, and the derivative of the stress through the local solve, which the
  // reference obtains as dxi/dx = -(dC/dxi)^-1 dC/dx (evaluations.cpp:101-115), is the consistent tangent
  // ds = a dev(sym d grad u) + b n (n : dev(sym d grad u)),  a = 2 mu theta,  b = 2 mu (1 - 2 mu / H - theta),
  // theta = 1 - 2 mu dgam / |s_tr|, H = 2 mu + 2K/3 (elastic: a = 2 mu, b = 0).  The branch is the reference's test at the
  // initial guess (f > tol || |f| < tol).  The forward wave kernel of hex8 uses it (c8_assemble_wave.hpp, CLOSED) in place
  // of the Newton iteration, the inverse of dC/dxi and the AD passes of phase D; every other kernel, and this one when the
  // caller asks for it (C8_KERNEL_WAVE_AD) or allows fewer than eight Newton iterations, runs the AD form above.
  static constexpr bool HAS_CLOSED_FORM = (DIM == 3);
  struct ClosedForm {
    static constexpr int NT = 14;
    double xi[NLOC];   // converged local state
    double F[13];      // flux values: Gu (xx xy xz yx yy yz zx zy zz), Vp (both ip sets), Gp
    double t[NT];      // what the tangent columns need: a, b, tr(n)/3, 1/kappa, tau, n (9)
  };
  // once per point (the kernel runs it on one lane per point and hands t to the lanes that write the columns)
  // pressure_mass: the -p / kappa term of the pressure residual (Mechanics::flux_pressure) rides along -- the wave kernel
  // fuses the two ip sets of hex8; the lane-group kernels leave it to the loop over the second ip set
  C8_HD static void closed_form(double const* prm, double const* q, double const* xi_old, double abs_tol, double h,
                                double stab_mult, ClosedForm& cf, bool pressure_mass) {
    double const sqrt_23 = 0.81649658092772603273;
    double const E = prm[0], nu = prm[1], K = prm[2], Y = prm[3];
    double const mu = E * c8_rcp(2. * (1. + nu)), kappa = E * c8_rcp(3. * (1. - 2. * nu));
    double const inv_mu = c8_rcp(mu);
    // q: grad u (row-major), p, grad p, u
    double eps[9];
    C8_UNROLL
    for (int i = 0; i < 3; ++i)
      C8_UNROLL
      for (int j = 0; j < 3; ++j) eps[3 * i + j] = 0.5 * (q[3 * i + j] + q[3 * j + i]);
    double const tr = eps[0] + eps[4] + eps[8];
    double const th = tr * (1. / 3.);
    double const pold[9] = {xi_old[0], xi_old[1], xi_old[2], xi_old[1], xi_old[3], xi_old[4], xi_old[2], xi_old[4], xi_old[5]};
    double st[9], ss = 0.;
    C8_UNROLL
    for (int k = 0; k < 9; ++k) {
      double const dev = eps[k] - ((k == 0 || k == 4 || k == 8) ? th : 0.);
      st[k] = (2. * mu) * (dev - pold[k]);
      ss += st[k] * st[k];
    }
    double const smag = sqrt(ss);
    double const alpha_old = xi_old[NSYM];
    double const excess = smag - sqrt_23 * (Y + K * alpha_old);
    double const f0 = excess * inv_mu;
    bool const plastic = f0 > abs_tol || fabs(f0) < abs_tol;
    double theta = 1.;
    double* n = cf.t + 5;
    cf.t[1] = 0.;
    C8_UNROLL
    for (int k = 0; k < NLOC; ++k) cf.xi[k] = xi_old[k];
    C8_UNROLL
    for (int k = 0; k < 9; ++k) n[k] = 0.;
    cf.t[2] = 0.;
    if (plastic) {
      double const inv_H = c8_rcp(2. * mu + (2. / 3.) * K);
      double const dgam = excess * inv_H;
      double const inv = c8_rcp(smag);
      C8_UNROLL
      for (int k = 0; k < 9; ++k) n[k] = st[k] * inv;
      cf.t[2] = (n[0] + n[4] + n[8]) * (1. / 3.);
      theta = 1. - (2. * mu) * dgam * inv;
      cf.t[1] = (2. * mu) * (1. - (2. * mu) * inv_H - theta);
      cf.xi[0] = xi_old[0] + dgam * n[0]; cf.xi[1] = xi_old[1] + dgam * n[1]; cf.xi[2] = xi_old[2] + dgam * n[2];
      cf.xi[3] = xi_old[3] + dgam * n[4]; cf.xi[4] = xi_old[4] + dgam * n[5]; cf.xi[5] = xi_old[5] + dgam * n[8];
      cf.xi[NSYM] = alpha_old + sqrt_23 * dgam;
    }
    cf.t[0] = (2. * mu) * theta;
    double const inv_kappa = c8_rcp(kappa);
    double const tau = (stab_mult * 0.5 * h * h) * inv_mu;
    cf.t[3] = pressure_mass ? inv_kappa : 0.;
    cf.t[4] = tau;
    double const p = q[9];
    C8_UNROLL
    for (int k = 0; k < 9; ++k) cf.F[k] = theta * st[k] - ((k == 0 || k == 4 || k == 8) ? p : 0.);
    double const hydro = kappa * tr - prm[4] * prm[5] * E * c8_rcp(1. - 2. * nu);
    cf.F[9] = pressure_mass ? -(hydro * inv_kappa) + -(p * inv_kappa) : -(hydro * inv_kappa);
    cf.F[10] = -(tau * q[10]); cf.F[11] = -(tau * q[11]); cf.F[12] = -(tau * q[12]);
  }
  // (d flux / d q)(d q / d x_b) for one element unknown x_b, 13 entries: x_b = u_k of a node (ek = e_k, isp = 0) or p of a
  // node (ek = 0, isp = 1); g = dN/dx and N = the shape value of that node at the point (a weight may ride on both).
  // With the consistent tangent d s_ij / d eps_kl = a (sym_ijkl - delta_ij delta_kl / 3) + b n_ij (n_kl - tr(n)/3 delta_kl):
  //   rows 0..8   a (1/2 (ek_i g_j + g_i ek_j) - delta_ij g_k / 3) + b n_ij (n_k. g - tr(n)/3 g_k) - delta_ij isp N
  //   row  9      -g_k - isp N / kappa          rows 10..12   -isp tau g_l
  // One instruction stream for all columns: no selects, the flags are factors.
  C8_HD static void closed_form_flux_column(double const* t, double const* ek, double isp, double const* g, double N, double* db) {
    double const a = t[0], b = t[1], trn3 = t[2], inv_kappa = t[3], tau = t[4];
    double const* n = t + 5;
    double const gk = ek[0] * g[0] + ek[1] * g[1] + ek[2] * g[2];
    double ng[3];  // n g
    C8_UNROLL
    for (int i = 0; i < 3; ++i) ng[i] = n[3 * i] * g[0] + n[3 * i + 1] * g[1] + n[3 * i + 2] * g[2];
    double const nkg = ek[0] * ng[0] + ek[1] * ng[1] + ek[2] * ng[2];  // n symmetric: (n g)_k = n_k. g
    double const bnk = b * (nkg - trn3 * gk);
    double const Np = isp * N;
    double const diag = a * gk * (1. / 3.) + Np;
    double const ha = 0.5 * a;
    double hg[3], hek[3];
    C8_UNROLL
    for (int i = 0; i < 3; ++i) { hg[i] = ha * g[i]; hek[i] = ek[i]; }
    C8_UNROLL
    for (int i = 0; i < 3; ++i)
      C8_UNROLL
      for (int j = 0; j < 3; ++j) {
        double const v = hek[i] * hg[j] + hg[i] * hek[j] + n[3 * i + j] * bnk;
        db[3 * i + j] = (i == j) ? v - diag : v;
      }
    db[9] = -gk - Np * inv_kappa;
    double const tp = isp * tau;
    db[10] = -(tp * g[0]); db[11] = -(tp * g[1]); db[12] = -(tp * g[2]);
  }
  // ---- the same tangent, one NODE PAIR at a time (the row-per-node kernel, c8_assemble_node.hpp) -----------------------------
  // With g = dN_a/dx of the row node a and h = dN_m/dx of the column node m at a point, the 4 x 4 block
  // d R_(a,.) / d x_(m,.) of the point follows from the columns above by contracting with the row node's shape entries:
  //   (u_i, u_k)  w [ delta_ik a/2 (g.h) + a/2 g_k h_i - a/3 g_i h_k + b (n g)_i (n' h)_k ],   n' = n - tr(n)/3 I
  //   (u_i, p)   -w g_i N_m        (p, u_k)  -w N_a h_k        (p, p)  -w N_a N_m / kappa - tau w (g.h)
  // closed_form_row gathers what depends on the point and the ROW node only (NROW doubles), closed_form_block adds the block
  // of one column node: 57 operations per point and node pair, no selects.
  static constexpr bool HAS_CLOSED_FORM_ROWS = HAS_CLOSED_FORM;
  static constexpr int NROW = 14;
  // r: w g (3) | w b (n g) (3) | a/2 | w N_a | n' (xx xy xz yy yz zz);   el (the same at every point of an element): tau, 1/kappa
  // TR = true: the record of the TRANSPOSED blocks d R_(m,.) / d x_(a,.) -- what the adjoint assembly scatters
  // (evaluations.cpp:463-465) --; they differ from the blocks above only in which side carries n' and which n
  template <bool TR = false>
  C8_HD static void closed_form_row(double const* t, double w, double const* g, double Na, double* r, double* el) {
    double const a = t[0], b = t[1], trn3 = t[2];
    double const* n = t + 5;
    r[0] = w * g[0]; r[1] = w * g[1]; r[2] = w * g[2];
    double const wb = w * b;
    double const ta = TR ? trn3 : 0., tb = TR ? 0. : trn3;  // n' on the row node's side (TR) or on the column node's
    C8_UNROLL
    for (int i = 0; i < 3; ++i) r[3 + i] = wb * (n[3 * i] * g[0] + n[3 * i + 1] * g[1] + n[3 * i + 2] * g[2] - ta * g[i]);
    r[6] = 0.5 * a;
    r[7] = w * Na;
    r[8] = n[0] - tb; r[9] = n[1]; r[10] = n[2]; r[11] = n[4] - tb; r[12] = n[5]; r[13] = n[8] - tb;
    el[0] = t[4];
    el[1] = t[3];
  }
  // The load term of the calibration objective (calibration.cpp:302-343 re-enters the weak form): J += wc sum_j sigma_(c j) S_j
  // with sigma = 2 mu (dev eps - pstrain) - p I.  Its partial derivatives at the point: dJ_dxi (local unknowns, packed
  // symmetric entries (c, j) of pstrain) and dJ_dq (grad u row-major 0..8, p 9) -- what the dual-number kernels obtain by
  // seeding xi and x in turn (evaluations.cpp:469-481).  c is a run-time index: selected by comparisons, no indexed tables.
  C8_HD static void closed_form_load_term(double const* prm, double wc, int c, double const* S, double* dJ_dxi, double* dJ_dq) {
    double const mu2 = prm[0] * c8_rcp(1. + prm[1]);  // 2 mu
    double const Sc = c == 0 ? S[0] : (c == 1 ? S[1] : S[2]);
    double const k = wc * mu2;
    // packed (00, 01, 02, 11, 12, 22): entry (i, j) takes S_j when c == i and, off the diagonal, S_i when c == j
    dJ_dxi[0] = -k * (c == 0 ? S[0] : 0.);
    dJ_dxi[1] = -k * (c == 0 ? S[1] : (c == 1 ? S[0] : 0.));
    dJ_dxi[2] = -k * (c == 0 ? S[2] : (c == 2 ? S[0] : 0.));
    dJ_dxi[3] = -k * (c == 1 ? S[1] : 0.);
    dJ_dxi[4] = -k * (c == 1 ? S[2] : (c == 2 ? S[1] : 0.));
    dJ_dxi[5] = -k * (c == 2 ? S[2] : 0.);
    dJ_dxi[NSYM] = 0.;
    C8_UNROLL
    for (int a = 0; a < 3; ++a)
      C8_UNROLL
      for (int l = 0; l < 3; ++l)
        dJ_dq[3 * a + l] = k * (0.5 * ((c == a ? S[l] : 0.) + (c == l ? S[a] : 0.)) - (a == l ? Sc * (1. / 3.) : 0.));
    dJ_dq[9] = -wc * Sc;
  }
  // (d xi / d eps)^T gxi at the point, as a symmetric tensor Re (xx xy xz yy yz zz): the term (dxi/dx)^T g of the adjoint
  // right-hand side (evaluations.cpp:486-487) without the elimination of dC/dxi.  From the radial return
  //   d pstrain = (2 mu / H) (n' : d eps) n + (1 - theta) (dev d eps - n (n' : d eps)),   d alpha = sqrt(2/3) (2 mu / H) n' : d eps
  // (elastic points: theta = 1, b = 0, and 2 mu / H below vanishes with them: Re = 0 as dC/dx = 0 gives).
  C8_HD static void closed_form_adjoint(double const* prm, double const* t, double const* gxi, double* Re) {
    double const sqrt_23 = 0.81649658092772603273;
    double const mu = prm[0] * c8_rcp(2. * (1. + prm[1]));
    double const i2mu = c8_rcp(2. * mu);
    double const a = t[0], b = t[1], trn3 = t[2];
    double const* n = t + 5;
    double const omt = 1. - a * i2mu;     // 1 - theta
    double const tmH = omt - b * i2mu;    // 2 mu / H
    double const G[6] = {gxi[0], 0.5 * gxi[1], 0.5 * gxi[2], gxi[3], 0.5 * gxi[4], gxi[5]};  // g . d xi = G : d pstrain
    double const Gn = G[0] * n[0] + G[3] * n[4] + G[5] * n[8] + 2. * (G[1] * n[1] + G[2] * n[2] + G[4] * n[5]);
    double const ca = -(b * i2mu) * Gn + gxi[NSYM] * sqrt_23 * tmH;
    double const trG3 = (G[0] + G[3] + G[5]) * (1. / 3.);
    Re[0] = ca * (n[0] - trn3) + omt * (G[0] - trG3);
    Re[1] = ca * n[1] + omt * G[1];
    Re[2] = ca * n[2] + omt * G[2];
    Re[3] = ca * (n[4] - trn3) + omt * (G[3] - trG3);
    Re[4] = ca * n[5] + omt * G[4];
    Re[5] = ca * (n[8] - trn3) + omt * (G[5] - trG3);
  }
  // The local adjoint solve (evaluations.cpp:528-659) in closed form: phi = (dC/dxi)^-T v with v = g - (dR/dxi)^T z, and
  // the history term of the previous step g' = -(dC/dxi_prev)^T phi.  With e = pstrain (packed 00 01 02 11 12 22; w_k = 2 for
  // the off-diagonal entries, which stand for two tensor entries), s = 2 mu (dev eps - e), n = s / |s|, beta = 2 mu dgam / |s|
  // = (1 - theta) / theta, kap = sqrt(2/3) K / mu, the plastic branch has
  //   dC_e / de [de] = (1 + beta) de - beta n (n : de),  dC_e / dalpha = -sqrt(3/2) n,  dC_alpha / de [de] = -2 n : de,
  //   dC_alpha / dalpha = -kap;      dC_e / de_prev = -I,  dC_e / dalpha_prev = sqrt(3/2) n,  dC_alpha / dxi_prev = 0,
  // so that with nv = sum_k n_k v_k (packed, unweighted; sum_k w_k n_k^2 = 1)
  //   phi_alpha = -(v_alpha + sqrt(3/2) nv) / (sqrt 6 + kap),   a = sum_k n_k phi_k = nv + 2 phi_alpha,
  //   phi_k = (v_k + w_k n_k (beta a + 2 phi_alpha)) / (1 + beta),   g'_k = phi_k,   g'_alpha = -sqrt(3/2) a.
  // Elastic points: C = xi - xi_prev, phi = v, g' = phi.  (dR/dxi)^T z = -2 mu w dv (E_k : grad z_u): the stress is the only
  // flux that sees xi.  t: what closed_form left for the tangent (a = 2 mu theta, n); ZG: grad z_u row-major (0..8).
  C8_HD static void closed_form_local_adjoint(double const* prm, double const* t, double wdv, double const* ZG, double const* g_in,
                                              double* phi, double* g_out) {
    static_assert(DIM == 3, "3-D form");
    double const sqrt_32 = 1.22474487139158904910, sqrt_6 = 2.44948974278317809820, sqrt_23 = 0.81649658092772603273;
    double const mu = prm[0] * c8_rcp(2. * (1. + prm[1]));
    double const c = 2. * mu * wdv;
    double const* nn = t + 5;
    double const n[6] = {nn[0], nn[1], nn[2], nn[4], nn[5], nn[8]};
    double v[NLOC];
    v[0] = g_in[0] + c * ZG[0];
    v[1] = g_in[1] + c * (ZG[1] + ZG[3]);
    v[2] = g_in[2] + c * (ZG[2] + ZG[6]);
    v[3] = g_in[3] + c * ZG[4];
    v[4] = g_in[4] + c * (ZG[5] + ZG[7]);
    v[5] = g_in[5] + c * ZG[8];
    v[NSYM] = g_in[NSYM];
    bool plastic = false;
    C8_UNROLL
    for (int k = 0; k < 9; ++k) plastic = plastic || (nn[k] != 0.);
    if (!plastic) {
      C8_UNROLL
      for (int k = 0; k < NLOC; ++k) { phi[k] = v[k]; g_out[k] = v[k]; }
      return;
    }
    double const theta = t[0] * c8_rcp(2. * mu);
    double const beta = (1. - theta) * c8_rcp(theta);
    double const kap = sqrt_23 * prm[2] * c8_rcp(mu);
    double nv = 0.;
    C8_UNROLL
    for (int k = 0; k < 6; ++k) nv += n[k] * v[k];
    double const pa = -(v[NSYM] + sqrt_32 * nv) * c8_rcp(sqrt_6 + kap);
    double const a = nv + 2. * pa;
    double const f = beta * a + 2. * pa;
    double const ib = c8_rcp(1. + beta);
    C8_UNROLL
    for (int k = 0; k < 6; ++k) {
      double const wk = (k == 1 || k == 2 || k == 4) ? 2. : 1.;
      phi[k] = (v[k] + wk * n[k] * f) * ib;
      g_out[k] = phi[k];
    }
    phi[NSYM] = pa;
    g_out[NSYM] = -sqrt_32 * a;
  }
  // The point's share of the parameter gradient (eval_qoi_gradient, evaluations.cpp:758-925) for parameter `mine` (E nu K Y cte
  // delta_T) at the stored state, in closed form: (dC/dp)^T phi + dJ/dp + (dR/dp)^T z.  The local residual sees the parameters
  // through C_alpha = (|s| - sqrt(2/3)(Y + K alpha)) / val(mu) only (plastic points; n = s / |s| does not depend on mu), the
  // fluxes through mu (stress 2 mu D, D = dev eps - pstrain; stabilisation tau = stab h^2 / (2 mu)), kappa (-p / kappa) and
  // the thermal term cte delta_T E / ((1 - 2 nu) kappa), whose E and nu derivatives cancel; the load term of the
  // calibration objective through the stress.  q: grad u (0..8), p (9), grad p (10..12); ZG likewise for the adjoint z.
  C8_HD static double closed_form_param_gradient(double const* prm, double const* q, double const* xi, double abs_tol, double h,
                                                 double stab_mult, double wdv, double const* ZG, double const* phi, int mine,
                                                 double wload, int comp, double const* S) {
    static_assert(DIM == 3, "3-D form");
    double const sqrt_23 = 0.81649658092772603273;
    double const E = prm[0], nu = prm[1], K = prm[2], Y = prm[3], cte = prm[4], dT = prm[5];
    double const mu = E * c8_rcp(2. * (1. + nu)), kappa = E * c8_rcp(3. * (1. - 2. * nu));
    double const dmu = mine == 0 ? mu * c8_rcp(E) : (mine == 1 ? -mu * c8_rcp(1. + nu) : 0.);
    double const dkap = mine == 0 ? kappa * c8_rcp(E) : (mine == 1 ? 2. * kappa * c8_rcp(1. - 2. * nu) : 0.);
    double const dK = mine == 2 ? 1. : 0., dY = mine == 3 ? 1. : 0., dcte = mine == 4 ? 1. : 0., ddT = mine == 5 ? 1. : 0.;
    double eps[9];
    C8_UNROLL
    for (int i = 0; i < 3; ++i)
      C8_UNROLL
      for (int j = 0; j < 3; ++j) eps[3 * i + j] = 0.5 * (q[3 * i + j] + q[3 * j + i]);
    double const th = (eps[0] + eps[4] + eps[8]) * (1. / 3.);
    double const ep[9] = {xi[0], xi[1], xi[2], xi[1], xi[3], xi[4], xi[2], xi[4], xi[5]};
    double D[9], dd = 0.;
    C8_UNROLL
    for (int k = 0; k < 9; ++k) {
      D[k] = eps[k] - ((k == 0 || k == 4 || k == 8) ? th : 0.) - ep[k];
      dd += D[k] * D[k];
    }
    double const smag = 2. * mu * sqrt(dd);
    double const alpha = xi[NSYM];
    double const f = (smag - sqrt_23 * (Y + K * alpha)) * c8_rcp(mu);
    bool const plastic = f > abs_tol || fabs(f) < abs_tol;
    double s = 0.;
    if (plastic) s = phi[NSYM] * (dmu * smag * c8_rcp(mu) - sqrt_23 * (dY + dK * alpha)) * c8_rcp(mu);
    double DZ = 0.;
    C8_UNROLL
    for (int k = 0; k < 9; ++k) DZ += D[k] * ZG[k];
    double const th3 = E * c8_rcp((1. - 2. * nu) * kappa);  // = 3
    double const dVp = dkap * q[9] * c8_rcp(kappa * kappa) + (dcte * dT + ddT * cte) * th3;
    double const tau = (stab_mult * 0.5 * h * h) * c8_rcp(mu);
    double const dGp = dmu * tau * c8_rcp(mu);
    s += wdv * (2. * dmu * DZ + dVp * ZG[9] + dGp * (q[10] * ZG[10] + q[11] * ZG[11] + q[12] * ZG[12]));
    if (wload != 0.) s += 2. * dmu * wload * (D[3 * comp] * S[0] + D[3 * comp + 1] * S[1] + D[3 * comp + 2] * S[2]);
    return s;
  }
  // J[4 i + k] += block entry (row i of the row node, column k of the column node; 3 = p)
  C8_HD static void closed_form_block(double const* r, double const* el, double const* h, double Nm, double* J) {
    double const gh = r[0] * h[0] + r[1] * h[1] + r[2] * h[2];
    double const nh[3] = {r[8] * h[0] + r[9] * h[1] + r[10] * h[2], r[9] * h[0] + r[11] * h[1] + r[12] * h[2],
                          r[10] * h[0] + r[12] * h[1] + r[13] * h[2]};
    double const p1[3] = {r[6] * r[0], r[6] * r[1], r[6] * r[2]};   // a/2 w g_k
    double const m3 = r[6] * (-2. / 3.);                               // -a/3
    double const p2[3] = {m3 * h[0], m3 * h[1], m3 * h[2]};
    C8_UNROLL
    for (int i = 0; i < 3; ++i)
      C8_UNROLL
      for (int k = 0; k < 3; ++k) J[4 * i + k] = fma(r[3 + i], nh[k], fma(r[i], p2[k], fma(p1[k], h[i], J[4 * i + k])));
    double const d = r[6] * gh;
    J[0] += d; J[5] += d; J[10] += d;
    C8_UNROLL
    for (int i = 0; i < 3; ++i) J[4 * i + 3] = fma(-r[i], Nm, J[4 * i + 3]);
    C8_UNROLL
    for (int k = 0; k < 3; ++k) J[12 + k] = fma(-r[7], h[k], J[12 + k]);
    J[15] = fma(-el[0], gh, fma(-(r[7] * el[1]), Nm, J[15]));
  }
};
template <class T> struct SmallJ2 : SmallJ2Dim<T, 3> {};       // "small_J2" on a 3-D mesh
template <class T> struct SmallJ2Plane : SmallJ2Dim<T, 2> {};  // "small_J2" on a 2-D mesh (notch2D_small_J2.yaml.in)

// ---- small_hill_plane_strain.cpp (2-D meshes): Hill's yield function on the in-plane deviatoric stress completed by
//      s_zz = 2 mu (-tr(eps)/3 + tr(pstrain)) (:226-233), Voce hardening, flow along the in-plane part of the Hill normal
//      (:243-247); R02 = R12 = 1 -----------------------------------------------------------------------------------
template <class T> struct SmallHillPlaneStrain {
  static constexpr int NLOC = 4, NPARAMS = 9;
  static constexpr bool FINITE_DEF = false, HAS_LOCAL = true;
  static constexpr int WAVE_BLOCKS_PER_CU = 2, WAVE_BLOCKS_PER_CU_ADJ = 2;
  static constexpr int WAVE_BLOCKS_PER_CU_K4 = 2;
  static constexpr bool GJ_XLANE_JAC = false, GJ_XLANE_K4 = false;
  static constexpr bool NEWTON_MATRIX_IN_LDS = false;
  using Trial = NoTrial;
  C8_HD Trial trial(PointState<T> const&) const { return {}; }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, Trial const&) { return evaluate(g, abs_tol); }
  T params[NPARAMS];  // E nu Y S D R00 R11 R22 R01  (small_hill_plane_strain.cpp:76-84)
  T xi[NLOC], xi_prev[NLOC], R[NLOC];  // pstrain (00,01,11), alpha
  C8_HD static void init_variables(double* xi0) { C8_UNROLL for (int k = 0; k < NLOC; ++k) xi0[k] = 0.; }
  C8_HD void initial_guess(PointState<T> const&) {  // :140-148
    C8_UNROLL
    for (int k = 0; k < NLOC; ++k) set_val(xi[k], val(xi_prev[k]));
  }
  C8_HD Tens3<T> dev_cauchy(PointState<T> const& g) const {  // :293-304
    T const mu = compute_mu(params[0], params[1]);
    Tens3<T> const eps = small_strain(g.grad_u);
    return scale(2. * mu, minus_s_eye<2>(eps, trace(eps) * (1. / 3.)) - sym_dim<2>(xi));
  }
  C8_HD Tens3<T> cauchy(PointState<T> const& g) const { return minus_s_eye<2>(dev_cauchy(g), g.p); }  // :280-290
  C8_HD T hydro_cauchy(PointState<T> const& g) const {  // :307-315
    return compute_kappa(params[0], params[1]) * trace(small_strain(g.grad_u));
  }
  C8_HD T pressure_scale_factor() const { return compute_kappa(params[0], params[1]); }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, bool force_path = false, int path_in = 0) {  // :193-277
    T const mu = compute_mu(params[0], params[1]);
    T const Y = params[2], S = params[3], D = params[4];
    auto inv2 = [](T const& r) { return 1. / (r * r); };
    T const i00 = inv2(params[5]), i11 = inv2(params[6]), i22 = inv2(params[7]);
    T const F = 0.5 * (i11 + i22 - i00), G = 0.5 * (i22 + i00 - i11), H = 0.5 * (i00 + i11 - i22);  // compute_hill_params
    T const N = 1.5 * inv2(params[8]);  // L, M belong to the out-of-plane shears, which vanish
    T const alpha = xi[3], alpha_old = xi_prev[3];
    Tens3<T> const ps = sym_dim<2>(xi);
    Tens3<T> const s = dev_cauchy(g);
    T const s_zz = (2. * mu) * (-(trace(small_strain(g.grad_u)) * (1. / 3.)) + trace(ps));
    T const d12 = s.yy - s_zz, d20 = s_zz - s.xx, d01 = s.xx - s.yy;
    T const hill = c8_sqrt(F * d12 * d12 + G * d20 * d20 + H * d01 * d01 + 2. * (N * s.xy * s.xy));  // compute_hill_value
    T const sigma_yield = Y + S * (1. - c8_exp(-(D * alpha)));
    T const f = (hill - sigma_yield) / val(mu);
    int path;
    if (!force_path) path = (val(f) > abs_tol || fabs(val(f)) < abs_tol) ? C8_PLASTIC_PATH : C8_ELASTIC_PATH;
    else path = path_in;
    if (path == C8_PLASTIC_PATH) {
      Tens3<T> n = scale(0., ps);  // in-plane part of compute_hill_normal
      n.xx = ((G + H) * s.xx - H * s.yy - G * s_zz) / hill;
      n.yy = ((F + H) * s.yy - H * s.xx - F * s_zz) / hill;
      n.xy = n.yx = N * s.xy / hill;
      T const dgam = alpha - alpha_old;
      pack_sym_dim<2>(ps - sym_dim<2>(xi_prev) - scale(dgam, n), R);
      R[3] = f;
    } else {
      C8_UNROLL
      for (int k = 0; k < NLOC; ++k) R[k] = xi[k] - xi_prev[k];
    }
    return path;
  }
};

// ---- hyper_J2_plane_strain.cpp (2-D meshes): finite-deformation J2 in plane strain.  Local unknowns zeta (00,01,11), Ie,
//      alpha; the in-plane tensors are completed out of plane where the 3-D quantity is needed: zeta_zz = -tr(zeta),
//      be_bar_zz = (zeta_zz + Ie) / det(rF)^(2/3).  The deformation gradient is F = grad u + I with the out-of-plane
//      stretch 1, so the 3 x 3 determinant, inverse and cofactor are the 2 x 2 ones. ------------------------------------
template <class T> struct HyperJ2PlaneStrain {
  static constexpr int NLOC = 5, NPARAMS = 6;
  static constexpr bool FINITE_DEF = true, HAS_LOCAL = true;
  static constexpr int WAVE_BLOCKS_PER_CU = 2, WAVE_BLOCKS_PER_CU_ADJ = 2;
  static constexpr int WAVE_BLOCKS_PER_CU_K4 = 2;
  static constexpr bool GJ_XLANE_JAC = false, GJ_XLANE_K4 = false;
  static constexpr bool NEWTON_MATRIX_IN_LDS = false;
  T params[NPARAMS];  // E nu K Y Y_inf delta  (hyper_J2_plane_strain.cpp:76-81)
  T xi[NLOC], xi_prev[NLOC], R[NLOC];
  C8_HD static void init_variables(double* xi0) {  // :119-131
    C8_UNROLL
    for (int k = 0; k < NLOC; ++k) xi0[k] = 0.;
    xi0[3] = 1.;
  }
  // the trial state depends on F, F_prev and the previous local state only (eval_be_bar_plane_strain :134-156)
  struct Trial { T zeta_trial[3], Ie_trial; };
  C8_HD Trial trial(PointState<T> const& g) const {
    Tens3<T> const I = eye3<T>();
    Tens3<T> const rF = matmul(g.grad_u + I, inverse(g.grad_u_prev + I));
    T const det_rF_13 = c8_cbrt(det(rF));
    Tens3<T> const rF_bar = scale(1. / det_rF_13, rF);
    Tens3<T> inner = sym_dim<2>(xi_prev);  // zeta_old + Ie_old I(2)
    inner.xx = inner.xx + xi_prev[3];
    inner.yy = inner.yy + xi_prev[3];
    Tens3<T> const be2 = matmul(matmul(rF_bar, inner), transpose(rF_bar));
    T const zeta_zz = -(xi_prev[0] + xi_prev[2]);
    T const be_zz = (zeta_zz + xi_prev[3]) / (det_rF_13 * det_rF_13);
    Trial t;
    t.Ie_trial = (be2.xx + be2.yy + be_zz) / 3.;
    t.zeta_trial[0] = be2.xx - t.Ie_trial;
    t.zeta_trial[1] = be2.xy;
    t.zeta_trial[2] = be2.yy - t.Ie_trial;
    return t;
  }
  C8_HD void initial_guess(PointState<T> const& g) {  // :168-184
    Trial const t = trial(g);
    C8_UNROLL
    for (int k = 0; k < 3; ++k) set_val(xi[k], val(t.zeta_trial[k]));
    set_val(xi[3], val(t.Ie_trial));
    set_val(xi[4], val(xi_prev[4]));
  }
  C8_HD Tens3<T> dev_cauchy(PointState<T> const& g) const {  // :340-351
    T const mu = compute_mu(params[0], params[1]);
    T const J = det(g.grad_u + eye3<T>());
    return scale(mu / J, sym_dim<2>(xi));
  }
  C8_HD Tens3<T> cauchy(PointState<T> const& g) const { return minus_s_eye<2>(dev_cauchy(g), g.p); }  // :328-337
  C8_HD T hydro_cauchy(PointState<T> const& g) const {  // :354-365
    T const kappa = compute_kappa(params[0], params[1]);
    T const J = det(g.grad_u + eye3<T>());
    return (kappa * 0.5) * (J - 1. / J);
  }
  C8_HD T pressure_scale_factor() const { return compute_kappa(params[0], params[1]); }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, bool force_path = false, int path_in = 0) {
    return evaluate(g, abs_tol, trial(g), force_path, path_in);
  }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, Trial const& tr, bool force_path = false, int path_in = 0) {  // :231-325
    double const sqrt_23 = 0.81649658092772603273;
    double const sqrt_32 = 1.22474487139158904910;
    T const mu = compute_mu(params[0], params[1]);
    T const K = params[2], Y = params[3], Y_inf = params[4], delta = params[5];
    T const Ie = xi[3], alpha = xi[4], alpha_old = xi_prev[4];
    Tens3<T> const zeta = sym_dim<2>(xi);
    T const zeta_zz = -(xi[0] + xi[2]);
    Tens3<T> s3 = scale(mu, zeta);  // s_3D = mu zeta_3D
    s3.zz = mu * zeta_zz;
    T const s_mag = norm(s3);
    T const sigma_yield = Y + K * alpha + (Y_inf - Y) * (1. - c8_exp(-(delta * alpha)));
    T const f = (s_mag - sqrt_23 * sigma_yield) / val(mu);
    int path;
    if (!force_path) path = (val(f) > abs_tol || fabs(val(f)) < abs_tol) ? C8_PLASTIC_PATH : C8_ELASTIC_PATH;
    else path = path_in;
    Tens3<T> Rz = zeta - sym_dim<2>(tr.zeta_trial);
    if (path == C8_PLASTIC_PATH) {
      T const dgam = sqrt_32 * (alpha - alpha_old);
      T const c = (2. * dgam) * Ie * mu / s_mag;  // 2 dgam Ie n_2D, n_2D = mu zeta / |s_3D|
      Rz = Rz + scale(c, zeta);
      Tens3<T> be = zeta;  // be_bar_3D = zeta_3D + Ie I(3)
      be.xx = be.xx + Ie; be.yy = be.yy + Ie; be.zz = zeta_zz + Ie;
      R[3] = det(be) - 1.;
      R[4] = f;
    } else {
      R[3] = Ie - tr.Ie_trial;
      R[4] = alpha - alpha_old;
    }
    pack_sym_dim<2>(Rz, R);
    return path;
  }
};

// ---- small_hill.cpp (Hill's anisotropic yield function, yield_functions.hpp:34-99; Voce hardening) ---------
template <class T> struct SmallHill {
  static constexpr bool PIN_PHASES_K1 = false, PIN_PHASES_K3 = true;  // jacobian_wave: lane-derived values per phase (pin_phases), as measured
  static constexpr int NLOC = 7, NPARAMS = 11;
  static constexpr bool FINITE_DEF = false, HAS_LOCAL = true;
  static constexpr int WAVE_BLOCKS_PER_CU = 2, WAVE_BLOCKS_PER_CU_ADJ = 2;
  static constexpr int WAVE_BLOCKS_PER_CU_K4 = 2;  // waves per SIMD of the local-adjoint wave kernel
  static constexpr bool GJ_XLANE_JAC = false, GJ_XLANE_K4 = true;  // pivot-column hand-over of the local solves (gj_solve_cols), as measured (round 3: LDS in K1 / K3)
  static constexpr bool NEWTON_MATRIX_IN_LDS = false;  // local Newton of the wave kernel: matrix columns in registers
  using Trial = NoTrial;
  C8_HD Trial trial(PointState<T> const&) const { return {}; }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, Trial const&) { return evaluate(g, abs_tol); }
  T params[NPARAMS];  // E nu Y R00 R11 R22 R01 R02 R12 S D  (small_hill.cpp:78-88)
  T xi[NLOC], xi_prev[NLOC], R[NLOC];  // pstrain(00,01,02,11,12,22), alpha
  C8_HD static void init_variables(double* xi0) { C8_UNROLL for (int k = 0; k < NLOC; ++k) xi0[k] = 0.; }
  C8_HD void initial_guess(PointState<T> const&) {  // :143-151
    C8_UNROLL
    for (int k = 0; k < NLOC; ++k) set_val(xi[k], val(xi_prev[k]));
  }
  C8_HD Tens3<T> dev_cauchy(PointState<T> const& g) const {  // :284-295
    T const mu = compute_mu(params[0], params[1]);
    Tens3<T> const eps = small_strain(g.grad_u);
    Tens3<T> const pstrain = sym6(xi);
    return scale(2. * mu, dev(eps) - pstrain);
  }
  C8_HD Tens3<T> cauchy(PointState<T> const& g) const {  // :271-281
    Tens3<T> s = dev_cauchy(g);
    s.xx = s.xx - g.p; s.yy = s.yy - g.p; s.zz = s.zz - g.p;
    return s;
  }
  C8_HD T hydro_cauchy(PointState<T> const& g) const {  // :298-306
    return compute_kappa(params[0], params[1]) * trace(small_strain(g.grad_u));
  }
  C8_HD T pressure_scale_factor() const { return compute_kappa(params[0], params[1]); }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, bool force_path = false, int path_in = 0) {  // :196-268
    T const mu = compute_mu(params[0], params[1]);
    T const Y = params[2], S = params[9], D = params[10];
    auto inv2 = [](T const& r) { return 1. / (r * r); };
    T const i00 = inv2(params[3]), i11 = inv2(params[4]), i22 = inv2(params[5]);
    T const F = 0.5 * (i11 + i22 - i00), G = 0.5 * (i22 + i00 - i11), H = 0.5 * (i00 + i11 - i22);  // compute_hill_params
    T const L = 1.5 * inv2(params[8]), M = 1.5 * inv2(params[7]), N = 1.5 * inv2(params[6]);
    T const alpha = xi[6], alpha_old = xi_prev[6];
    Tens3<T> const s = dev_cauchy(g);
    T const d12 = s.yy - s.zz, d20 = s.zz - s.xx, d01 = s.xx - s.yy;
    T const hill = c8_sqrt(F * d12 * d12 + G * d20 * d20 + H * d01 * d01 +
                           2. * (L * s.yz * s.yz + M * s.xz * s.xz + N * s.xy * s.xy));  // compute_hill_value
    T const sigma_yield = Y + S * (1. - c8_exp(-(D * alpha)));
    T const f = (hill - sigma_yield) / val(mu);
    int path;
    if (!force_path) path = (val(f) > abs_tol || fabs(val(f)) < abs_tol) ? C8_PLASTIC_PATH : C8_ELASTIC_PATH;
    else path = path_in;
    if (path == C8_PLASTIC_PATH) {
      Tens3<T> n;  // compute_hill_normal
      n.xx = ((G + H) * s.xx - H * s.yy - G * s.zz) / hill;
      n.yy = ((F + H) * s.yy - H * s.xx - F * s.zz) / hill;
      n.zz = ((G + F) * s.zz - G * s.xx - F * s.yy) / hill;
      n.xy = n.yx = N * s.xy / hill;
      n.xz = n.zx = M * s.xz / hill;
      n.yz = n.zy = L * s.yz / hill;
      T const dgam = alpha - alpha_old;
      Tens3<T> const ps = sym6(xi);
      Tens3<T> Rp = ps - sym6(xi_prev) - scale(dgam, n);
      Rp.zz = trace(ps);  // the (2,2) equation is replaced by plastic incompressibility (:236)
      pack_sym6(Rp, R);
      R[6] = f;
    } else {
      C8_UNROLL
      for (int k = 0; k < NLOC; ++k) R[k] = xi[k] - xi_prev[k];
    }
    return path;
  }
};

// ---- hypo_hill.cpp: hypoelastic rate form in the unrotated configuration (local unknown = unrotated Cauchy stress
//      TC), Hill's yield function (yield_functions.hpp:34-99), Voce hardening --------------------------------------
template <class T> struct HypoHill {
  static constexpr bool PIN_PHASES_K1 = true, PIN_PHASES_K3 = true;  // jacobian_wave: lane-derived values per phase (pin_phases), as measured
  static constexpr int NLOC = 7, NPARAMS = 11;
  static constexpr bool FINITE_DEF = true, HAS_LOCAL = true;
#ifndef C8_TUNE_HH_WAVES
#define C8_TUNE_HH_WAVES 2
#endif
  static constexpr int WAVE_BLOCKS_PER_CU = C8_TUNE_HH_WAVES, WAVE_BLOCKS_PER_CU_ADJ = 1;
  static constexpr int WAVE_BLOCKS_PER_CU_K4 = 1;  // waves per SIMD of the local-adjoint wave kernel
  static constexpr bool GJ_XLANE_JAC = false, GJ_XLANE_K4 = false;  // pivot-column hand-over of the local solves (gj_solve_cols), as measured (round 3: LDS in K1 / K3)
#ifdef C8_TUNE_HH_NEWTON_LDS
  static constexpr bool NEWTON_MATRIX_IN_LDS = true;
#else
  static constexpr bool NEWTON_MATRIX_IN_LDS = false;  // with the DPP hand-over K1 takes 32.5 ms (39.2 with the matrix in LDS)
#endif
  T params[NPARAMS];  // E nu Y R00 R11 R22 R01 R02 R12 S D  (hypo_hill.cpp:84-95)
  T xi[NLOC], xi_prev[NLOC], R[NLOC];  // TC(00,01,02,11,12,22), alpha
  C8_HD static void init_variables(double* xi0) { C8_UNROLL for (int k = 0; k < NLOC; ++k) xi0[k] = 0.; }  // :123-131
  // the unrotated rate of deformation d = R^T sym((F - F_prev) F^-1) R depends on F and F_prev only
  // (eval_d :134-139, hypo_kinematics.hpp:11-18; the reference caches it over the local Newton iteration, :150-151)
  struct Trial { T d[6]; };
  C8_HD Trial trial(PointState<T> const& g) const {
    Tens3<T> const I = eye3<T>();
    Tens3<T> const F = g.grad_u + I;
    Tens3<T> const F_prev = g.grad_u_prev + I;
    Tens3<T> const Rot = polar_rotation(F);
    Tens3<T> const L = matmul(F - F_prev, inverse(F));
    Tens3<T> const D = scale(0.5, L + transpose(L));
    Trial t;
    pack_sym6(matmul(matmul(transpose(Rot), D), Rot), t.d);
    return t;
  }
  C8_HD void initial_guess(PointState<T> const& g) {  // :153-168: elastic predictor, values only
    double const E = val(params[0]), nu = val(params[1]);
    double const lambda = compute_lambda(E, nu), mu = compute_mu(E, nu);
    Trial const t = trial(g);
    double const ltr = lambda * (val(t.d[0]) + val(t.d[3]) + val(t.d[5]));
    C8_UNROLL
    for (int k = 0; k < 6; ++k) set_val(xi[k], val(xi_prev[k]) + 2. * mu * val(t.d[k]) + ((k == 0 || k == 3 || k == 5) ? ltr : 0.));
    set_val(xi[6], val(xi_prev[6]));
  }
  C8_HD Tens3<T> rotated_cauchy(PointState<T> const& g) const {  // :292-298
    Tens3<T> const Rot = polar_rotation(g.grad_u + eye3<T>());
    return matmul(matmul(Rot, sym6(xi)), transpose(Rot));
  }
  C8_HD Tens3<T> dev_cauchy(PointState<T> const& g) const { return dev(rotated_cauchy(g)); }  // :313-316
  C8_HD Tens3<T> cauchy(PointState<T> const& g) const {  // :301-310
    Tens3<T> s = dev_cauchy(g);
    s.xx = s.xx - g.p; s.yy = s.yy - g.p; s.zz = s.zz - g.p;
    return s;
  }
  C8_HD T hydro_cauchy(PointState<T> const& g) const { return trace(rotated_cauchy(g)) / 3.; }  // :319-322
  // both at once (one polar decomposition instead of two); Mechanics::flux_coupled uses it when a model has it
  C8_HD void cauchy_and_hydro(PointState<T> const& g, Tens3<T>& sigma, T& sigma_h) const {
    Tens3<T> const RC = rotated_cauchy(g);
    sigma_h = trace(RC) / 3.;
    sigma = dev(RC);
    sigma.xx = sigma.xx - g.p; sigma.yy = sigma.yy - g.p; sigma.zz = sigma.zz - g.p;
  }
  C8_HD T pressure_scale_factor() const { return compute_kappa(params[0], params[1]); }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, bool force_path = false, int path_in = 0) {
    return evaluate(g, abs_tol, trial(g), force_path, path_in);
  }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, Trial const& tr, bool force_path = false, int path_in = 0) {  // :211-289
    T const lambda = compute_lambda(params[0], params[1]);
    T const mu = compute_mu(params[0], params[1]);
    T const Y = params[2], S = params[9], D = params[10];
    auto inv2 = [](T const& r) { return 1. / (r * r); };
    T const i00 = inv2(params[3]), i11 = inv2(params[4]), i22 = inv2(params[5]);
    T const F = 0.5 * (i11 + i22 - i00), G = 0.5 * (i22 + i00 - i11), H = 0.5 * (i00 + i11 - i22);  // compute_hill_params
    T const L = 1.5 * inv2(params[8]), M = 1.5 * inv2(params[7]), N = 1.5 * inv2(params[6]);
    T const alpha = xi[6], alpha_old = xi_prev[6];
    Tens3<T> const TC = sym6(xi);
    T const d12 = TC.yy - TC.zz, d20 = TC.zz - TC.xx, d01 = TC.xx - TC.yy;
    T const hill = c8_sqrt(F * d12 * d12 + G * d20 * d20 + H * d01 * d01 +
                           2. * (L * TC.yz * TC.yz + M * TC.xz * TC.xz + N * TC.xy * TC.xy));  // compute_hill_value
    T const sigma_yield = Y + S * (1. - c8_exp(-(D * alpha)));
    T const f = (hill - sigma_yield) / val(mu);
    Tens3<T> const d = sym6(tr.d);
    T const ltr = lambda * trace(d);
    Tens3<T> Rt = TC - sym6(xi_prev) - scale(2. * mu, d);
    Rt.xx = Rt.xx - ltr; Rt.yy = Rt.yy - ltr; Rt.zz = Rt.zz - ltr;
    Rt = scale(1. / val(mu), Rt);
    int path;
    if (!force_path) path = (val(f) > abs_tol || fabs(val(f)) < abs_tol) ? C8_PLASTIC_PATH : C8_ELASTIC_PATH;
    else path = path_in;
    if (path == C8_PLASTIC_PATH) {
      Tens3<T> n;  // compute_hill_normal
      n.xx = ((G + H) * TC.xx - H * TC.yy - G * TC.zz) / hill;
      n.yy = ((F + H) * TC.yy - H * TC.xx - F * TC.zz) / hill;
      n.zz = ((G + F) * TC.zz - G * TC.xx - F * TC.yy) / hill;
      n.xy = n.yx = N * TC.xy / hill;
      n.xz = n.zx = M * TC.xz / hill;
      n.yz = n.zy = L * TC.yz / hill;
      T const dgam = alpha - alpha_old;
      Rt = Rt + scale((2. * mu * dgam) / val(mu), n);
      R[6] = f;
    } else {
      R[6] = alpha - alpha_old;
    }
    pack_sym6(Rt, R);
    return path;
  }
};

// ---- hypo_hill_plane_strain.cpp (2-D meshes): the rate form of hypo_hill on in-plane kinematics, the out-of-plane
//      stress as an extra unknown.  Local unknowns TC (00,01,11), alpha, TC_zz; residuals are NOT scaled by 1/mu here.  The
//      polar rotation of F = grad u + I with unit out-of-plane stretch is the in-plane rotation completed by 1. ----------
template <class T> struct HypoHillPlaneStrain {
  static constexpr int NLOC = 5, NPARAMS = 9;
  static constexpr bool FINITE_DEF = true, HAS_LOCAL = true;
  static constexpr int WAVE_BLOCKS_PER_CU = 2, WAVE_BLOCKS_PER_CU_ADJ = 2;
  static constexpr int WAVE_BLOCKS_PER_CU_K4 = 2;
  static constexpr bool GJ_XLANE_JAC = false, GJ_XLANE_K4 = false;
  static constexpr bool NEWTON_MATRIX_IN_LDS = false;
  T params[NPARAMS];  // E nu Y S D R00 R11 R22 R01  (hypo_hill_plane_strain.cpp:84-92)
  T xi[NLOC], xi_prev[NLOC], R[NLOC];
  C8_HD static void init_variables(double* xi0) { C8_UNROLL for (int k = 0; k < NLOC; ++k) xi0[k] = 0.; }  // :124-140
  // d = R^T sym((F - F_prev) F^-1) R depends on F and F_prev only (eval_d :143-156)
  struct Trial { T d[3]; };
  C8_HD Trial trial(PointState<T> const& g) const {
    Tens3<T> const I = eye3<T>();
    Tens3<T> const F = g.grad_u + I;
    Tens3<T> const F_prev = g.grad_u_prev + I;
    Tens3<T> const Rot = polar_rotation(F);
    Tens3<T> const L = matmul(F - F_prev, inverse(F));
    Tens3<T> const D = scale(0.5, L + transpose(L));
    Trial t;
    pack_sym_dim<2>(matmul(matmul(transpose(Rot), D), Rot), t.d);
    return t;
  }
  C8_HD void initial_guess(PointState<T> const& g) {  // :168-186: elastic predictor, values only
    double const E = val(params[0]), nu = val(params[1]);
    double const lambda = compute_lambda(E, nu), mu = compute_mu(E, nu);
    Trial const t = trial(g);
    double const ltr = lambda * (val(t.d[0]) + val(t.d[2]));
    set_val(xi[0], val(xi_prev[0]) + ltr + 2. * mu * val(t.d[0]));
    set_val(xi[1], val(xi_prev[1]) + 2. * mu * val(t.d[1]));
    set_val(xi[2], val(xi_prev[2]) + ltr + 2. * mu * val(t.d[2]));
    set_val(xi[3], val(xi_prev[3]));
    set_val(xi[4], val(xi_prev[4]) + ltr);
  }
  C8_HD Tens3<T> rotated_cauchy(PointState<T> const& g) const {  // :332-341
    Tens3<T> const Rot = polar_rotation(g.grad_u + eye3<T>());
    return matmul(matmul(Rot, sym_dim<2>(xi)), transpose(Rot));
  }
  C8_HD T hydro_cauchy(PointState<T> const& g) const { return (trace(rotated_cauchy(g)) + xi[4]) / 3.; }  // :365-369
  C8_HD Tens3<T> dev_cauchy(PointState<T> const& g) const {  // :355-362
    Tens3<T> const RC = rotated_cauchy(g);
    return minus_s_eye<2>(RC, (trace(RC) + xi[4]) / 3.);
  }
  C8_HD Tens3<T> cauchy(PointState<T> const& g) const { return minus_s_eye<2>(dev_cauchy(g), g.p); }  // :343-352
  // both at once (one polar decomposition instead of two); Mechanics::flux_coupled uses it
  C8_HD void cauchy_and_hydro(PointState<T> const& g, Tens3<T>& sigma, T& sigma_h) const {
    Tens3<T> const RC = rotated_cauchy(g);
    sigma_h = (trace(RC) + xi[4]) / 3.;
    sigma = minus_s_eye<2>(minus_s_eye<2>(RC, sigma_h), g.p);
  }
  C8_HD T pressure_scale_factor() const { return compute_kappa(params[0], params[1]); }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, bool force_path = false, int path_in = 0) {
    return evaluate(g, abs_tol, trial(g), force_path, path_in);
  }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, Trial const& tr, bool force_path = false, int path_in = 0) {  // :232-329
    T const lambda = compute_lambda(params[0], params[1]);
    T const mu = compute_mu(params[0], params[1]);
    T const Y = params[2], S = params[3], D = params[4];
    auto inv2 = [](T const& r) { return 1. / (r * r); };
    T const i00 = inv2(params[5]), i11 = inv2(params[6]), i22 = inv2(params[7]);
    T const F = 0.5 * (i11 + i22 - i00), G = 0.5 * (i22 + i00 - i11), H = 0.5 * (i00 + i11 - i22);  // compute_hill_params
    T const N = 1.5 * inv2(params[8]);
    T const alpha = xi[3], alpha_old = xi_prev[3], TC_zz = xi[4];
    Tens3<T> const TC = sym_dim<2>(xi);
    T const d12 = TC.yy - TC_zz, d20 = TC_zz - TC.xx, d01 = TC.xx - TC.yy;
    T const phi = c8_sqrt(F * d12 * d12 + G * d20 * d20 + H * d01 * d01 + 2. * (N * TC.xy * TC.xy));  // compute_hill_value
    T const sigma_yield = Y + S * (1. - c8_exp(-(D * alpha)));
    T const f = (phi - sigma_yield) / val(mu);
    Tens3<T> const d = sym_dim<2>(tr.d);
    T const ltr = lambda * trace(d);
    Tens3<T> Rt = minus_s_eye<2>(TC - sym_dim<2>(xi_prev), ltr) - scale(2. * mu, d);
    T Rzz = TC_zz - xi_prev[4] - ltr;
    int path;
    if (!force_path) path = (val(f) > abs_tol || fabs(val(f)) < abs_tol) ? C8_PLASTIC_PATH : C8_ELASTIC_PATH;
    else path = path_in;
    if (path == C8_PLASTIC_PATH) {
      Tens3<T> n = scale(0., TC);  // in-plane part of compute_hill_normal
      n.xx = ((G + H) * TC.xx - H * TC.yy - G * TC_zz) / phi;
      n.yy = ((F + H) * TC.yy - H * TC.xx - F * TC_zz) / phi;
      n.xy = n.yx = N * TC.xy / phi;
      T const dgam = alpha - alpha_old;
      Tens3<T> const dp = scale(dgam, n);
      Rt = Rt + scale(2. * mu, dp);
      Rzz = Rzz + (2. * mu) * (-(dp.xx + dp.yy));
      R[3] = f;
    } else {
      R[3] = alpha - alpha_old;
    }
    pack_sym_dim<2>(Rt, R);
    R[4] = Rzz;
    return path;
  }
};


// =====================================================================================================================
// Plane-stress family: the local models that pair with `mechanics_plane_stress` (one global residual, u).  cauchy() is
// the in-plane Cauchy stress -- sigma_zz = 0 is built into the models --; there is no pressure unknown, so g.p is never
// read and pressure_scale_factor() (0 in the reference) is never used.
// =====================================================================================================================

// compute_hill_params / compute_hill_value / in-plane compute_hill_normal with R02 = R12 = 1 (yield_functions.hpp:35-98)
// on a 2-D stress completed by s_zz; params[5..8] = R00 R11 R22 R01
template <class T> struct HillPlane {
  T F, G, H, N;
  C8_HD explicit HillPlane(T const* params) {
    auto inv2 = [](T const& r) { return 1. / (r * r); };
    T const i00 = inv2(params[5]), i11 = inv2(params[6]), i22 = inv2(params[7]);
    F = 0.5 * (i11 + i22 - i00);
    G = 0.5 * (i22 + i00 - i11);
    H = 0.5 * (i00 + i11 - i22);
    N = 1.5 * inv2(params[8]);
  }
  C8_HD T value(Tens3<T> const& s, T const& s_zz) const {
    T const d12 = s.yy - s_zz, d20 = s_zz - s.xx, d01 = s.xx - s.yy;
    return c8_sqrt(F * d12 * d12 + G * d20 * d20 + H * d01 * d01 + 2. * (N * s.xy * s.xy));
  }
  C8_HD Tens3<T> normal(Tens3<T> const& s, T const& s_zz, T const& hill) const {
    Tens3<T> n = scale(0., s);
    n.xx = ((G + H) * s.xx - H * s.yy - G * s_zz) / hill;
    n.yy = ((F + H) * s.yy - H * s.xx - F * s_zz) / hill;
    n.xy = n.yx = N * s.xy / hill;
    return n;
  }
};

// ---- small_hill_plane_stress.cpp: small strain, eps_zz eliminated by sigma_zz = 0 (:318-329) ----------------------------
template <class T> struct SmallHillPlaneStress {
  static constexpr bool PLANE_STRESS = true;  // pairs with mechanics_plane_stress (one global residual)
  static constexpr int NLOC = 4, NPARAMS = 9;
  static constexpr bool FINITE_DEF = false, HAS_LOCAL = true;
  static constexpr int WAVE_BLOCKS_PER_CU = 2, WAVE_BLOCKS_PER_CU_ADJ = 2;
  static constexpr int WAVE_BLOCKS_PER_CU_K4 = 2;
  static constexpr bool GJ_XLANE_JAC = false, GJ_XLANE_K4 = false;
  static constexpr bool NEWTON_MATRIX_IN_LDS = false;
  using Trial = NoTrial;
  C8_HD Trial trial(PointState<T> const&) const { return {}; }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, Trial const&) { return evaluate(g, abs_tol); }
  T params[NPARAMS];  // E nu Y S D R00 R11 R22 R01  (small_hill_plane_stress.cpp:70-78)
  T xi[NLOC], xi_prev[NLOC], R[NLOC];  // pstrain (00,01,11), alpha
  C8_HD static void init_variables(double* xi0) { C8_UNROLL for (int k = 0; k < NLOC; ++k) xi0[k] = 0.; }
  C8_HD void initial_guess(PointState<T> const&) {  // :138-146
    C8_UNROLL
    for (int k = 0; k < NLOC; ++k) set_val(xi[k], val(xi_prev[k]));
  }
  C8_HD Tens3<T> cauchy(PointState<T> const& g) const {  // :278-293 with epsilon_zz of :318-329
    T const mu = compute_mu(params[0], params[1]);
    T const lambda = compute_lambda(params[0], params[1]);
    Tens3<T> const eps = small_strain(g.grad_u);
    Tens3<T> const ps = sym_dim<2>(xi);
    T const eps_zz = -(lambda * trace(eps) + 2. * mu * trace(ps)) / (lambda + 2. * mu);
    T const eps_kk = trace(eps) + eps_zz;
    return minus_s_eye<2>(scale(2. * mu, eps - ps), -(lambda * eps_kk));
  }
  C8_HD T hydro_cauchy(PointState<T> const& g) const { return trace(cauchy(g)) / 3.; }  // :305-309
  C8_HD Tens3<T> dev_cauchy(PointState<T> const& g) const {  // :296-302
    Tens3<T> const c = cauchy(g);
    return minus_s_eye<2>(c, trace(c) / 3.);
  }
  C8_HD T pressure_scale_factor() const { return T(0.); }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, bool force_path = false, int path_in = 0) {  // :192-275
    T const mu = compute_mu(params[0], params[1]);
    T const Y = params[2], S = params[3], D = params[4];
    HillPlane<T> const hp(params);
    T const alpha = xi[3], alpha_old = xi_prev[3];
    Tens3<T> const sigma = cauchy(g);
    T const zero = T(0.);
    T const hill = hp.value(sigma, zero);
    T const sigma_yield = Y + S * (1. - c8_exp(-(D * alpha)));
    T const f = (hill - sigma_yield) / val(mu);
    int path;
    if (!force_path) path = (val(f) > abs_tol || fabs(val(f)) < abs_tol) ? C8_PLASTIC_PATH : C8_ELASTIC_PATH;
    else path = path_in;
    if (path == C8_PLASTIC_PATH) {
      T const dgam = alpha - alpha_old;
      pack_sym_dim<2>(sym_dim<2>(xi) - sym_dim<2>(xi_prev) - scale(dgam, hp.normal(sigma, zero, hill)), R);
      R[3] = f;
    } else {
      C8_UNROLL
      for (int k = 0; k < NLOC; ++k) R[k] = xi[k] - xi_prev[k];
    }
    return path;
  }
};

// ---- hyper_J2_plane_stress.cpp: finite-deformation J2; local unknowns zeta (00,01,11), Ie, the out-of-plane stretch
//      lambda_z and alpha.  F_3D = [F_2D, lambda_z], zeta_zz = -tr(zeta); lambda_z closes sigma_zz = 0 (:302-304) --------
template <class T> struct HyperJ2PlaneStress {
  static constexpr bool PLANE_STRESS = true;  // pairs with mechanics_plane_stress (one global residual)
  static constexpr int NLOC = 6, NPARAMS = 8;
  static constexpr int Z_STRETCH = 4;  // m_z_stretch_idx (:60): position of lambda_z in xi
  static constexpr bool FINITE_DEF = true, HAS_LOCAL = true;
  static constexpr int WAVE_BLOCKS_PER_CU = 2, WAVE_BLOCKS_PER_CU_ADJ = 2;
  static constexpr int WAVE_BLOCKS_PER_CU_K4 = 2;
  static constexpr bool GJ_XLANE_JAC = false, GJ_XLANE_K4 = false;
  static constexpr bool NEWTON_MATRIX_IN_LDS = false;
  using Trial = NoTrial;  // the trial state depends on the unknown lambda_z: nothing to cache over the Newton iteration
  C8_HD Trial trial(PointState<T> const&) const { return {}; }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, Trial const&) { return evaluate(g, abs_tol); }
  T params[NPARAMS];  // E nu Y S D A n K  (hyper_J2_plane_stress.cpp:80-87)
  T xi[NLOC], xi_prev[NLOC], R[NLOC];
  C8_HD static void init_variables(double* xi0) {  // :121-138
    C8_UNROLL
    for (int k = 0; k < NLOC; ++k) xi0[k] = 0.;
    xi0[3] = 1.;
    xi0[4] = 1.;
  }
  // eval_be_bar_plane_stress (:141-169) from the previous zeta, Ie, lambda_z and the current lambda_z; returns the trial
  // values Ie_trial and zeta_trial (00,01,11) and det F_2D
  C8_HD void trial_state(PointState<T> const& g, T const& lambda_z, T* zeta_trial, T& Ie_trial, T& J_2D) const {
    Tens3<T> F = g.grad_u + eye3<T>();
    J_2D = det(F);  // F_zz = 1 here: the 2 x 2 determinant
    Tens3<T> F_prev = g.grad_u_prev + eye3<T>();
    F.zz = lambda_z;
    F_prev.zz = xi_prev[4];
    Tens3<T> const rF = matmul(F, inverse(F_prev));
    T const det_rF_13 = c8_cbrt(det(rF));
    Tens3<T> const rF_bar = scale(1. / det_rF_13, rF);
    Tens3<T> inner = sym_dim<2>(xi_prev);  // zeta_3D + Ie I(3) of the previous step
    inner.zz = -(xi_prev[0] + xi_prev[2]);
    inner.xx = inner.xx + xi_prev[3];
    inner.yy = inner.yy + xi_prev[3];
    inner.zz = inner.zz + xi_prev[3];
    Tens3<T> const be = matmul(matmul(rF_bar, inner), transpose(rF_bar));
    Ie_trial = trace(be) / 3.;
    zeta_trial[0] = be.xx - Ie_trial;
    zeta_trial[1] = be.xy;
    zeta_trial[2] = be.yy - Ie_trial;
  }
  C8_HD void initial_guess(PointState<T> const& g) {  // :183-201: zeta and Ie from the trial state; lambda_z, alpha stay
    T zt[3], Ie_t, J_2D;
    trial_state(g, xi[4], zt, Ie_t, J_2D);
    C8_UNROLL
    for (int k = 0; k < 3; ++k) set_val(xi[k], val(zt[k]));
    set_val(xi[3], val(Ie_t));
  }
  C8_HD T jac(PointState<T> const& g) const { return det(g.grad_u + eye3<T>()) * xi[4]; }
  C8_HD Tens3<T> dev_cauchy(PointState<T> const& g) const {  // :377-389
    return scale(compute_mu(params[0], params[1]) / jac(g), sym_dim<2>(xi));
  }
  C8_HD T hydro_cauchy(PointState<T> const& g) const {  // :392-403
    T const J = jac(g);
    return (compute_kappa(params[0], params[1]) * 0.5) * (J - 1. / J);
  }
  C8_HD Tens3<T> cauchy(PointState<T> const& g) const {  // :361-374
    T const J = jac(g);
    T const kappa = compute_kappa(params[0], params[1]);
    return minus_s_eye<2>(scale(compute_mu(params[0], params[1]) / J, sym_dim<2>(xi)), -((kappa * 0.5) * (J - 1. / J)));
  }
  C8_HD T pressure_scale_factor() const { return T(0.); }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, bool force_path = false, int path_in = 0) {  // :246-358
    double const sqrt_23 = 0.81649658092772603273;
    double const sqrt_32 = 1.22474487139158904910;
    T const mu = compute_mu(params[0], params[1]), kappa = compute_kappa(params[0], params[1]);
    T const Y = params[2], S = params[3], D = params[4], A = params[5], nexp = params[6], K = params[7];
    T const Ie = xi[3], lambda_z = xi[4], alpha = xi[5], alpha_old = xi_prev[5];
    T zt[3], Ie_trial, J_2D;
    trial_state(g, lambda_z, zt, Ie_trial, J_2D);
    Tens3<T> const zeta = sym_dim<2>(xi);
    T const zeta_zz = -(xi[0] + xi[2]);
    Tens3<T> s3 = scale(mu, zeta);  // s = mu zeta_3D
    s3.zz = mu * zeta_zz;
    T const s_mag = norm(s3);
    double const power_law_offset = 1e-12;
    T const sigma_yield = Y + S * (1. - c8_exp(-(D * alpha))) + A * c8_pow(alpha + power_law_offset, nexp) + K * alpha;
    T const f = (s_mag - sqrt_23 * sigma_yield) / val(mu);
    T const mat_factor = kappa / (2. * mu);
    R[4] = lambda_z - c8_sqrt((1. - zeta_zz / mat_factor) / (J_2D * J_2D));
    int path;
    if (!force_path) path = (val(f) > abs_tol || fabs(val(f)) < abs_tol) ? C8_PLASTIC_PATH : C8_ELASTIC_PATH;
    else path = path_in;
    Tens3<T> Rz = zeta - sym_dim<2>(zt);
    if (path == C8_PLASTIC_PATH) {
      T const dgam = sqrt_32 * (alpha - alpha_old);
      T const c = (2. * dgam) * Ie * mu / s_mag;  // 2 dgam Ie n_2D, n_2D = mu zeta / |s|
      Rz = Rz + scale(c, zeta);
      Tens3<T> be = zeta;  // be_bar = zeta_3D + Ie I(3)
      be.xx = be.xx + Ie; be.yy = be.yy + Ie; be.zz = zeta_zz + Ie;
      R[3] = det(be) - 1.;
      R[5] = f;
    } else {
      R[3] = Ie - Ie_trial;
      R[5] = alpha - alpha_old;
    }
    pack_sym_dim<2>(Rz, R);
    return path;
  }
};

// ---- hypo_hill_plane_stress.cpp: hypoelastic rate form, unrotated in-plane Cauchy stress TC (00,01,11), alpha and the
//      out-of-plane stretch lambda_z; the material axes Q rotate the rate of deformation (:164-177) and the stress
//      (:378-388).  The TC rows of the plastic residual are divided by val(mu) on the unforced path only (:303). ---------
template <class T> struct HypoHillPlaneStress {
  static constexpr bool PLANE_STRESS = true;  // pairs with mechanics_plane_stress (one global residual)
  static constexpr int NLOC = 5, NPARAMS = 13;
  static constexpr int Z_STRETCH = 4;  // m_z_stretch_idx (:72)
  static constexpr bool FINITE_DEF = true, HAS_LOCAL = true;
  static constexpr int WAVE_BLOCKS_PER_CU = 2, WAVE_BLOCKS_PER_CU_ADJ = 2;
  static constexpr int WAVE_BLOCKS_PER_CU_K4 = 2;
  static constexpr bool GJ_XLANE_JAC = false, GJ_XLANE_K4 = false;
  static constexpr bool NEWTON_MATRIX_IN_LDS = false;
  T params[NPARAMS];  // E nu Y S D R00 R11 R22 R01 Q00 Q01 Q10 Q11  (hypo_hill_plane_stress.cpp:92-104)
  T xi[NLOC], xi_prev[NLOC], R[NLOC];
  C8_HD static void init_variables(double* xi0) {  // :138-152
    C8_UNROLL
    for (int k = 0; k < NLOC; ++k) xi0[k] = 0.;
    xi0[4] = 1.;
  }
  C8_HD Tens3<T> material_axes() const {  // compute_Q :155-162
    Tens3<T> Q = scale(0., eye3<T>());
    Q.xx = params[9]; Q.xy = params[10]; Q.yx = params[11]; Q.yy = params[12];
    return Q;
  }
  // d = Q^T R^T sym((F - F_prev) F^-1) R Q depends on F and F_prev only (eval_d :164-177)
  struct Trial { T d[3]; };
  C8_HD Trial trial(PointState<T> const& g) const {
    Tens3<T> const I = eye3<T>();
    Tens3<T> const F = g.grad_u + I;
    Tens3<T> const F_prev = g.grad_u_prev + I;
    Tens3<T> const Rot = polar_rotation(F);
    Tens3<T> const L = matmul(F - F_prev, inverse(F));
    Tens3<T> const D = scale(0.5, L + transpose(L));
    Tens3<T> const Q = material_axes();
    Trial t;
    pack_sym_dim<2>(matmul(matmul(matmul(matmul(transpose(Q), transpose(Rot)), D), Rot), Q), t.d);
    return t;
  }
  C8_HD void initial_guess(PointState<T> const& g) {  // :191-211: elastic predictor, values only
    double const E = val(params[0]), nu = val(params[1]);
    double const lambda = compute_lambda(E, nu), mu = compute_mu(E, nu);
    Trial const t = trial(g);
    double const tr_d = val(t.d[0]) + val(t.d[2]);
    double const d_zz = -lambda * tr_d / (lambda + 2. * mu);
    double const ltr = lambda * (tr_d + d_zz);
    set_val(xi[0], val(xi_prev[0]) + ltr + 2. * mu * val(t.d[0]));
    set_val(xi[1], val(xi_prev[1]) + 2. * mu * val(t.d[1]));
    set_val(xi[2], val(xi_prev[2]) + ltr + 2. * mu * val(t.d[2]));
    set_val(xi[3], val(xi_prev[3]));
    set_val(xi[4], val(xi_prev[4]) / (1. - d_zz));
  }
  C8_HD Tens3<T> rotated_cauchy(PointState<T> const& g) const {  // :378-388
    Tens3<T> const Rot = polar_rotation(g.grad_u + eye3<T>());
    Tens3<T> const Q = material_axes();
    return matmul(matmul(matmul(matmul(Rot, Q), sym_dim<2>(xi)), transpose(Q)), transpose(Rot));
  }
  C8_HD Tens3<T> cauchy(PointState<T> const& g) const { return rotated_cauchy(g); }                  // :391-393
  C8_HD T hydro_cauchy(PointState<T> const& g) const { return trace(rotated_cauchy(g)) / 3.; }       // :403-405
  C8_HD Tens3<T> dev_cauchy(PointState<T> const& g) const {                                          // :396-400
    Tens3<T> const RC = rotated_cauchy(g);
    return minus_s_eye<2>(RC, trace(RC) / 3.);
  }
  C8_HD T pressure_scale_factor() const { return T(0.); }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, bool force_path = false, int path_in = 0) {
    return evaluate(g, abs_tol, trial(g), force_path, path_in);
  }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, Trial const& tr, bool force_path = false, int path_in = 0) {  // :256-375
    T const lambda = compute_lambda(params[0], params[1]);
    T const mu = compute_mu(params[0], params[1]);
    T const Y = params[2], S = params[3], D = params[4];
    HillPlane<T> const hp(params);
    T const alpha = xi[3], alpha_old = xi_prev[3], lambda_z = xi[4], lambda_z_old = xi_prev[4];
    Tens3<T> const TC = sym_dim<2>(xi);
    T const zero = T(0.);
    T const phi = hp.value(TC, zero);
    T const sigma_yield = Y + S * (1. - c8_exp(-(D * alpha)));
    T const f = (phi - sigma_yield) / val(mu);
    Tens3<T> const d = sym_dim<2>(tr.d);
    T const d_zz = -(lambda * trace(d)) / (lambda + 2. * mu);
    Tens3<T> Rt = minus_s_eye<2>(TC - sym_dim<2>(xi_prev), lambda * (trace(d) + d_zz)) - scale(2. * mu, d);
    int path;
    if (!force_path) path = (val(f) > abs_tol || fabs(val(f)) < abs_tol) ? C8_PLASTIC_PATH : C8_ELASTIC_PATH;
    else path = path_in;
    if (path == C8_PLASTIC_PATH) {
      T const dgam = alpha - alpha_old;
      Tens3<T> const dp = scale(dgam, hp.normal(TC, zero, phi));
      T const dp_zz = -(dp.xx + dp.yy);
      T const corr_dp_zz = (2. * mu) * dp_zz / (2. * mu + lambda);  // correction from the return map
      Rt.xx = Rt.xx + ((2. * mu) * dp.xx - lambda * corr_dp_zz);
      Rt.yy = Rt.yy + ((2. * mu) * dp.yy - lambda * corr_dp_zz);
      Rt.xy = Rt.xy + (2. * mu) * dp.xy;
      Rt.yx = Rt.xy;
      if (!force_path) Rt = scale(1. / val(mu), Rt);
      R[3] = f;
      R[4] = lambda_z - lambda_z_old / (1. - (d_zz + corr_dp_zz));
    } else {
      R[3] = alpha - alpha_old;
      R[4] = lambda_z - lambda_z_old / (1. - d_zz);
    }
    pack_sym_dim<2>(Rt, R);
    return path;
  }
};


// =====================================================================================================================
// Hosford / Barlat family: yield functions on principal stresses (minitensor::eig_spd_cos, restated in c8_math.hpp), a
// local Newton iteration whose branch is chosen by the first evaluation and forced afterwards, and the cubic line search
// of line_search.hpp after every step (LINE_SEARCH: the assembly kernels run local_newton_line_search for these models).
// =====================================================================================================================
template <class M, class = void> struct uses_line_search : std::false_type {};
template <class M> struct uses_line_search<M, std::enable_if_t<M::LINE_SEARCH>> : std::true_type {};

// small_hosford.cpp:228-265, hypo_hosford.cpp:264-301: phi = vm (1/2 sum |(s_i - s_j) / vm|^a)^(1/a) and its normal
template <class T>
C8_HD void hosford_phi_and_normal(Tens3<T> const& sigma, T const& vm_stress, T const& a, T& phi, Tens3<T>& n) {
  Tens3<T> V;
  T D[3];
  eig_spd_cos(sigma, V, D);
  T const e0 = D[0] / vm_stress, e1 = D[1] / vm_stress, e2 = D[2] / vm_stress;
  phi = vm_stress * c8_pow(0.5 * (c8_pow(c8_abs(e0 - e1), a) + c8_pow(c8_abs(e1 - e2), a) + c8_pow(c8_abs(e2 - e0), a)), 1. / a);
  T const q0 = D[0] / phi, q1 = D[1] / phi, q2 = D[2] / phi;
  T const d01 = q0 - q1, d12 = q1 - q2, d20 = q2 - q0;
  T const am2 = a - 2.;
  T const f01 = d01 * c8_pow(c8_abs(d01), am2), f12 = d12 * c8_pow(c8_abs(d12), am2), f20 = d20 * c8_pow(c8_abs(d20), am2);
  n = scale(0., sigma);
  add_dyad_col(n, f01 - f20, V, 0);
  add_dyad_col(n, f12 - f01, V, 1);
  add_dyad_col(n, f20 - f12, V, 2);
  n = scale(0.5, n);
}

// ---- small_hosford.cpp ---------------------------------------------------------------------------------------------
template <class T> struct SmallHosford {
  static constexpr int NLOC = 7, NPARAMS = 7;
  static constexpr bool FINITE_DEF = false, HAS_LOCAL = true, LINE_SEARCH = true;
  static constexpr int WAVE_BLOCKS_PER_CU = 1, WAVE_BLOCKS_PER_CU_ADJ = 1;
  static constexpr int WAVE_BLOCKS_PER_CU_K4 = 1;
  static constexpr bool GJ_XLANE_JAC = false, GJ_XLANE_K4 = false;
  static constexpr bool NEWTON_MATRIX_IN_LDS = false;
  using Trial = NoTrial;
  C8_HD Trial trial(PointState<T> const&) const { return {}; }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, Trial const&) { return evaluate(g, abs_tol); }
  T params[NPARAMS];  // E nu Y a K S D  (small_hosford.cpp:86-92)
  T xi[NLOC], xi_prev[NLOC], R[NLOC];  // pstrain(00,01,02,11,12,22), alpha
  C8_HD static void init_variables(double* xi0) { C8_UNROLL for (int k = 0; k < NLOC; ++k) xi0[k] = 0.; }
  C8_HD void initial_guess(PointState<T> const&) {  // :143-151
    C8_UNROLL
    for (int k = 0; k < NLOC; ++k) set_val(xi[k], val(xi_prev[k]));
  }
  C8_HD Tens3<T> dev_cauchy(PointState<T> const& g) const {  // :356-368
    T const mu = compute_mu(params[0], params[1]);
    return scale(2. * mu, dev(small_strain(g.grad_u)) - sym6(xi));
  }
  C8_HD Tens3<T> cauchy(PointState<T> const& g) const {  // :343-353
    Tens3<T> s = dev_cauchy(g);
    s.xx = s.xx - g.p; s.yy = s.yy - g.p; s.zz = s.zz - g.p;
    return s;
  }
  C8_HD T hydro_cauchy(PointState<T> const& g) const { return compute_kappa(params[0], params[1]) * trace(small_strain(g.grad_u)); }
  C8_HD T pressure_scale_factor() const { return compute_kappa(params[0], params[1]); }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, bool force_path = false, int path_in = 0) {  // :268-340
    T const mu = compute_mu(params[0], params[1]);
    T const Y = params[2], a = params[3], K = params[4], S = params[5], D = params[6];
    T const alpha = xi[6], alpha_old = xi_prev[6];
    Tens3<T> const s = dev_cauchy(g);
    Tens3<T> sigma = s;
    sigma.xx = s.xx - g.p; sigma.yy = s.yy - g.p; sigma.zz = s.zz - g.p;
    T const vm_stress = 1.22474487139158904910 * norm(s);
    T phi;
    Tens3<T> n;
    hosford_phi_and_normal(sigma, vm_stress, a, phi, n);
    T const flow_stress = Y + K * alpha + S * (1. - c8_exp(-(D * alpha)));
    T const f = (phi - flow_stress) / (2. * val(mu));
    int path;
    if (!force_path) path = (val(f) > abs_tol || fabs(val(f)) < abs_tol) ? C8_PLASTIC_PATH : C8_ELASTIC_PATH;
    else path = path_in;
    if (path == C8_PLASTIC_PATH) {
      T const dgam = alpha - alpha_old;
      pack_sym6(sym6(xi) - sym6(xi_prev) - scale(dgam, n), R);
      R[6] = f;
    } else {
      C8_UNROLL
      for (int k = 0; k < NLOC; ++k) R[k] = xi[k] - xi_prev[k];
    }
    return path;
  }
};

// ---- hypo_hosford.cpp / hypo_barlat.cpp: the hypoelastic frame of hypo_hill.cpp (unrotated Cauchy stress TC, elastic
//      predictor as the initial guess) with residual rows scaled by 1 / (2 mu) -- a differentiable scale here ------------
template <class T, class Derived, int NP> struct HypoPrincipalBase {
  static constexpr int NLOC = 7, NPARAMS = NP;
  static constexpr bool FINITE_DEF = true, HAS_LOCAL = true, LINE_SEARCH = true;
  static constexpr int WAVE_BLOCKS_PER_CU = 1, WAVE_BLOCKS_PER_CU_ADJ = 1;
  static constexpr int WAVE_BLOCKS_PER_CU_K4 = 1;
  static constexpr bool GJ_XLANE_JAC = false, GJ_XLANE_K4 = false;
  static constexpr bool NEWTON_MATRIX_IN_LDS = false;
  T params[NPARAMS];
  T xi[NLOC], xi_prev[NLOC], R[NLOC];  // TC(00,01,02,11,12,22), alpha
  C8_HD static void init_variables(double* xi0) { C8_UNROLL for (int k = 0; k < NLOC; ++k) xi0[k] = 0.; }
  struct Trial { T d[6]; };  // d = R^T sym((F - F_prev) F^-1) R (hypo_kinematics.hpp:11-18; material axes Q = I)
  C8_HD Trial trial(PointState<T> const& g) const {
    Tens3<T> const I = eye3<T>();
    Tens3<T> const F = g.grad_u + I;
    Tens3<T> const F_prev = g.grad_u_prev + I;
    Tens3<T> const Rot = polar_rotation(F);
    Tens3<T> const L = matmul(F - F_prev, inverse(F));
    Tens3<T> const D = scale(0.5, L + transpose(L));
    Trial t;
    pack_sym6(matmul(matmul(transpose(Rot), D), Rot), t.d);
    return t;
  }
  C8_HD void initial_guess(PointState<T> const& g) {  // hypo_hosford.cpp:166-181, hypo_barlat.cpp:336-351
    double const E = val(params[0]), nu = val(params[1]);
    double const lambda = compute_lambda(E, nu), mu = compute_mu(E, nu);
    Trial const t = trial(g);
    double const ltr = lambda * (val(t.d[0]) + val(t.d[3]) + val(t.d[5]));
    C8_UNROLL
    for (int k = 0; k < 6; ++k) set_val(xi[k], val(xi_prev[k]) + ((k == 0 || k == 3 || k == 5) ? ltr : 0.) + 2. * mu * val(t.d[k]));
    set_val(xi[6], val(xi_prev[6]));
  }
  C8_HD Tens3<T> rotated_cauchy(PointState<T> const& g) const {
    Tens3<T> const Rot = polar_rotation(g.grad_u + eye3<T>());
    return matmul(matmul(Rot, sym6(xi)), transpose(Rot));
  }
  C8_HD Tens3<T> dev_cauchy(PointState<T> const& g) const { return dev(rotated_cauchy(g)); }
  C8_HD Tens3<T> cauchy(PointState<T> const& g) const {
    Tens3<T> s = dev_cauchy(g);
    s.xx = s.xx - g.p; s.yy = s.yy - g.p; s.zz = s.zz - g.p;
    return s;
  }
  C8_HD T hydro_cauchy(PointState<T> const& g) const { return trace(rotated_cauchy(g)) / 3.; }
  C8_HD void cauchy_and_hydro(PointState<T> const& g, Tens3<T>& sigma, T& sigma_h) const {
    Tens3<T> const RC = rotated_cauchy(g);
    sigma_h = trace(RC) / 3.;
    sigma = dev(RC);
    sigma.xx = sigma.xx - g.p; sigma.yy = sigma.yy - g.p; sigma.zz = sigma.zz - g.p;
  }
  C8_HD T pressure_scale_factor() const { return compute_kappa(params[0], params[1]); }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, bool force_path = false, int path_in = 0) {
    return evaluate(g, abs_tol, trial(g), force_path, path_in);
  }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, Trial const& tr, bool force_path = false, int path_in = 0) {
    T const lambda = compute_lambda(params[0], params[1]);
    T const mu = compute_mu(params[0], params[1]);
    T const alpha = xi[6], alpha_old = xi_prev[6];
    Tens3<T> const TC = sym6(xi);
    Derived& self = *static_cast<Derived*>(this);
    T phi, flow_stress;
    self.yield(TC, alpha, phi, flow_stress);
    T const scale_factor = 2. * mu;
    T const f = (phi - flow_stress) / scale_factor;
    Tens3<T> const d = sym6(tr.d);
    T const ltr = lambda * trace(d);
    Tens3<T> Rt = TC - sym6(xi_prev);
    Rt.xx = Rt.xx - ltr; Rt.yy = Rt.yy - ltr; Rt.zz = Rt.zz - ltr;
    Rt = Rt - scale(2. * mu, d);
    Rt = scale(1. / scale_factor, Rt);
    int path;
    if (!force_path) path = (val(f) > abs_tol || fabs(val(f)) < abs_tol) ? C8_PLASTIC_PATH : C8_ELASTIC_PATH;
    else path = path_in;
    if (path == C8_PLASTIC_PATH) {
      T const dgam = alpha - alpha_old;
      Rt = Rt + scale(dgam, self.normal(phi));  // the scale factor of the TC rows removes the 2 mu multiplier
      R[6] = f;
    } else {
      R[6] = alpha - alpha_old;
    }
    pack_sym6(Rt, R);
    return path;
  }
};

// hypo_hosford.cpp (params E nu Y a K S D; the flow stress has no K term, :326)
template <class T> struct HypoHosford : HypoPrincipalBase<T, HypoHosford<T>, 7> {
  Tens3<T> n_;
  C8_HD void yield(Tens3<T> const& TC, T const& alpha, T& phi, T& flow_stress) {
    T const vm_stress = 1.22474487139158904910 * norm(dev(TC));
    hosford_phi_and_normal(TC, vm_stress, this->params[3], phi, n_);
    flow_stress = this->params[2] + this->params[5] * (1. - c8_exp(-(this->params[6] * alpha)));
  }
  C8_HD Tens3<T> normal(T const&) const { return n_; }
};

// hypo_barlat.cpp: Barlat's Yld2004-18p (yield_functions.hpp:101-386); params E nu Y a K S D sp_01 sp_02 sp_10 sp_12
// sp_20 sp_21 sp_33 sp_44 sp_55 dp_01 .. dp_55
template <class T> struct HypoBarlat : HypoPrincipalBase<T, HypoBarlat<T>, 25> {
  Tens3<T> Vs_, Vd_;
  T Ds_[3], Dd_[3];
  // L * stress for the transformation built from nine coefficients q (unflatten_barlat_params :162-189); Voigt order
  // (00, 11, 22, 01, 12, 20) (:128-141)
  C8_HD static Tens3<T> transform(T const* q, Tens3<T> const& t) {
    T const L00 = (q[0] + q[1]) / 3., L01 = (-2. * q[0] + q[1]) / 3., L02 = (q[0] - 2. * q[1]) / 3.;
    T const L10 = (-2. * q[2] + q[3]) / 3., L11 = (q[2] + q[3]) / 3., L12 = (q[2] - 2. * q[3]) / 3.;
    T const L20 = (-2. * q[4] + q[5]) / 3., L21 = (q[4] - 2. * q[5]) / 3., L22 = (q[4] + q[5]) / 3.;
    Tens3<T> s;
    s.xx = L00 * t.xx + L01 * t.yy + L02 * t.zz;
    s.yy = L10 * t.xx + L11 * t.yy + L12 * t.zz;
    s.zz = L20 * t.xx + L21 * t.yy + L22 * t.zz;
    s.xy = s.yx = q[6] * t.xy;
    s.yz = s.zy = q[7] * t.yz;
    s.xz = s.zx = q[8] * t.zx;
    return s;
  }
  C8_HD void yield(Tens3<T> const& TC, T const& alpha, T& phi, T& flow_stress) {  // evaluate_barlat_phi :322-366
    T const a = this->params[3];
    double const vm_phi = 1.22474487139158904910 * val(norm(dev(TC)));
    eig_spd_cos(transform(&this->params[7], TC), Vs_, Ds_);
    eig_spd_cos(transform(&this->params[16], TC), Vd_, Dd_);
    T sum = T(0.);
    C8_UNROLL
    for (int i = 0; i < 3; ++i)
      C8_UNROLL
      for (int j = 0; j < 3; ++j) sum = sum + c8_pow(c8_abs(Ds_[i] / vm_phi - Dd_[j] / vm_phi), a);
    sum = 0.25 * sum;
    phi = vm_phi * c8_exp((1.0 / a) * c8_log(sum));
    flow_stress = this->params[2] + this->params[4] * alpha + this->params[5] * (1. - c8_exp(-(this->params[6] * alpha)));
  }
  C8_HD Tens3<T> normal(T const& phi) const {  // evaluate_barlat_normal / compute_barlat_normal :293-320, :369-384
    T const am2 = this->params[3] - 2.;
    Tens3<T> sp_n = scale(0., Vs_), dp_n = scale(0., Vs_);
    C8_UNROLL
    for (int k = 0; k < 3; ++k) {
      T ms = T(0.), md = T(0.);
      C8_UNROLL
      for (int j = 0; j < 3; ++j) {
        T const ds = Ds_[k] / phi - Dd_[j] / phi;
        ms = ms + ds * c8_pow(c8_abs(ds), am2);
        T const dd = Ds_[j] / phi - Dd_[k] / phi;
        md = md + (-dd) * c8_pow(c8_abs(dd), am2);
      }
      add_dyad_col(sp_n, 0.25 * ms, Vs_, k);
      add_dyad_col(dp_n, 0.25 * md, Vd_, k);
    }
    return transform(&this->params[7], sp_n) + transform(&this->params[16], dp_n);
  }
};

// ---- hyper_J2.cpp -------------------------------------------------------------
template <class T> struct HyperJ2 {
  static constexpr bool PIN_PHASES_K1 = true, PIN_PHASES_K3 = true;  // jacobian_wave: lane-derived values per phase (pin_phases), as measured
  static constexpr int NLOC = 8, NPARAMS = 8;
  static constexpr bool FINITE_DEF = true, HAS_LOCAL = true;
  // measured on 1 M hex8 elements: at 256 registers the local-adjoint kernel spills 1 KB per lane (32 ms), at 512
  // registers and half the occupancy it takes 14.8 ms; the Jacobian kernels are faster at 2 workgroups per CU
  static constexpr int WAVE_BLOCKS_PER_CU = 2, WAVE_BLOCKS_PER_CU_ADJ = 1;
  static constexpr int WAVE_BLOCKS_PER_CU_K4 = 1;  // waves per SIMD of the local-adjoint wave kernel
  static constexpr bool GJ_XLANE_JAC = false, GJ_XLANE_K4 = false;  // pivot-column hand-over of the local solves (gj_solve_cols), as measured (round 3: LDS in K1 / K3)
  // local Newton of the wave kernel: matrix columns in registers.  (Round 2: 32.5 ms against 28.8 with the matrix in LDS; since the
  // phases form their lane-derived values themselves (PIN_PHASES_K1) the register form spills 12 B instead of 180: 18.4 against 21.5 ms)
  static constexpr bool NEWTON_MATRIX_IN_LDS = false;
  T params[NPARAMS];  // E nu Y S D A n K  (hyper_J2.cpp:83-90)
  T xi[NLOC], xi_prev[NLOC], R[NLOC];  // zeta(6), Ie, alpha
  C8_HD static void init_variables(double* xi0) {  // :119-134
    C8_UNROLL
    for (int k = 0; k < NLOC; ++k) xi0[k] = 0.;
    xi0[6] = 1.;
  }
  C8_HD Tens3<T> be_bar_trial(PointState<T> const& g) const {  // eval_be_bar(zeta_old, Ie_old) :137-154
    Tens3<T> const I = eye3<T>();
    Tens3<T> const F = g.grad_u + I;
    Tens3<T> const F_prev = g.grad_u_prev + I;
    Tens3<T> const rF = matmul(F, inverse(F_prev));
    T const det_rF_13 = c8_cbrt(det(rF));
    Tens3<T> const rF_bar = scale(1. / det_rF_13, rF);
    Tens3<T> be_old = sym6(xi_prev);
    be_old.xx = be_old.xx + xi_prev[6]; be_old.yy = be_old.yy + xi_prev[6]; be_old.zz = be_old.zz + xi_prev[6];
    return matmul(matmul(rF_bar, be_old), transpose(rF_bar));
  }
  C8_HD void initial_guess(PointState<T> const& g) {  // :167-178
    Tens3<T> const bt = be_bar_trial(g);
    Tens3<T> const z = dev(bt);
    T zv[6];
    pack_sym6(z, zv);
    C8_UNROLL
    for (int k = 0; k < 6; ++k) set_val(xi[k], val(zv[k]));
    set_val(xi[6], val(trace(bt)) / 3.);
    set_val(xi[7], val(xi_prev[7]));
  }
  C8_HD Tens3<T> dev_cauchy(PointState<T> const& g) const {  // :327-338
    T const mu = compute_mu(params[0], params[1]);
    Tens3<T> const F = g.grad_u + eye3<T>();
    T const J = det(F);
    return scale(mu / J, sym6(xi));
  }
  C8_HD Tens3<T> cauchy(PointState<T> const& g) const {  // :316-324
    Tens3<T> s = dev_cauchy(g);
    s.xx = s.xx - g.p; s.yy = s.yy - g.p; s.zz = s.zz - g.p;
    return s;
  }
  C8_HD T hydro_cauchy(PointState<T> const& g) const {  // :341-352
    T const kappa = compute_kappa(params[0], params[1]);
    Tens3<T> const F = g.grad_u + eye3<T>();
    T const J = det(F);
    return (kappa * 0.5) * (J - 1. / J);
  }
  C8_HD T pressure_scale_factor() const { return compute_kappa(params[0], params[1]); }
  // the trial elastic left Cauchy-Green tensor depends on F, F_prev and the previous state only (:137-154)
  struct Trial { T dev_bt[6], tr_bt_3; };
  C8_HD Trial trial(PointState<T> const& g) const {
    Tens3<T> const bt = be_bar_trial(g);
    Trial t;
    pack_sym6(dev(bt), t.dev_bt);
    t.tr_bt_3 = trace(bt) / 3.;
    return t;
  }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, bool force_path = false, int path_in = 0) {
    return evaluate(g, abs_tol, trial(g), force_path, path_in);
  }
  C8_HD int evaluate(PointState<T> const& g, double abs_tol, Trial const& tr, bool force_path = false, int path_in = 0) {  // :226-314
    double const sqrt_23 = 0.81649658092772603273;
    double const sqrt_32 = 1.22474487139158904910;
    T const mu = compute_mu(params[0], params[1]);
    T const Y = params[2], S = params[3], D = params[4], A = params[5], nexp = params[6], K = params[7];
    T const Ie = xi[6], alpha = xi[7], alpha_old = xi_prev[7];
    Tens3<T> const zeta = sym6(xi);
    Tens3<T> const s = scale(mu, zeta);
    T const s_mag = norm(s);
    double const power_law_offset = 1e-12;
    T const sigma_yield = Y + S * (1. - c8_exp(-(D * alpha))) + A * c8_pow(alpha + power_law_offset, nexp) + K * alpha;
    T const f = (s_mag - sqrt_23 * sigma_yield) / val(mu);
    int path;
    if (!force_path) path = (val(f) > abs_tol || fabs(val(f)) < abs_tol) ? C8_PLASTIC_PATH : C8_ELASTIC_PATH;
    else path = path_in;
    Tens3<T> Rz = zeta - sym6(tr.dev_bt);
    if (path == C8_PLASTIC_PATH) {
      T const dgam = sqrt_32 * (alpha - alpha_old);
      T const c = (2. * dgam) * Ie / s_mag;
      Rz = Rz + scale(c, s);
      Tens3<T> be = zeta;
      be.xx = be.xx + Ie; be.yy = be.yy + Ie; be.zz = be.zz + Ie;
      R[6] = det(be) - 1.;
      R[7] = f;
    } else {
      R[6] = Ie - tr.tr_bt_3;
      R[7] = alpha - alpha_old;
    }
    pack_sym6(Rz, R);
    return path;
  }
};

// ---- mechanics.cpp: stabilised mixed u-p weak form, as point fluxes ----------
// ip set 0 (mechanics.cpp:116-145, :169-213):
//   R_u[n,i] += sum_j P_ij dN_n/dx_j w dv            P = sigma (small strain) or sigma cof(F)
//   R_p[n]   -= (sigma_h/kappa) N_n w dv + sum_ij S_ij dp/dx_j dN_n/dx_i w dv,  S = tau I [cof^T cof / det F]
// ip set 1 (:215-223):
//   R_p[n]   -= (p/kappa) N_n w dv
template <class T> struct MechFlux {
  Tens3<T> Gu;  // multiplies grad N in the momentum residual
  T Vp;         // multiplies N in the pressure residual
  T Gp[3];      // multiplies grad N in the pressure residual
};

// optional member of a local model: Cauchy stress and its hydrostatic part from one evaluation
template <class L, class T, class = void> struct has_stress_pair : std::false_type {};
template <class L, class T>
struct has_stress_pair<L, T, decltype(std::declval<L const&>().cauchy_and_hydro(std::declval<PointState<T> const&>(),
                                                                              std::declval<Tens3<T>&>(), std::declval<T&>()))>
    : std::true_type {};

struct Mechanics {
  static constexpr bool USES_U = false;  // the integrand reads grad u, p, grad p but not u (mechanics.cpp:116-227)
  template <class T, class Local>
  C8_HD static void flux_coupled(Local const& local, PointState<T> const& g, double h, double stab_mult, MechFlux<T>& f) {
    Tens3<T> stress;
    T hydro;
    if constexpr (has_stress_pair<Local, T>::value) {
      local.cauchy_and_hydro(g, stress, hydro);
    } else {
      stress = local.cauchy(g);
      hydro = local.hydro_cauchy(g);
    }
    T const mu = compute_mu(local.params[0], local.params[1]);
    T const psf = local.pressure_scale_factor();
    T const tau = (stab_mult * 0.5 * h * h) / mu;  // mechanics.cpp:197
    f.Vp = -(hydro / psf);
    if (Local::FINITE_DEF) {
      Tens3<T> const F = g.grad_u + eye3<T>();
      Tens3<T> const C = cofactor(F);
      T const detF = det(F);
      stress = matmul(stress, C);  // PK1 = sigma cof(F)  (:133)
      // (tau / det F) C^T C grad p (:198-202), as two matrix-vector products instead of forming C^T C
      T const v0 = C.xx * g.grad_p[0] + C.xy * g.grad_p[1] + C.xz * g.grad_p[2];
      T const v1 = C.yx * g.grad_p[0] + C.yy * g.grad_p[1] + C.yz * g.grad_p[2];
      T const v2 = C.zx * g.grad_p[0] + C.zy * g.grad_p[1] + C.zz * g.grad_p[2];
      T const tj = tau / detF;
      f.Gp[0] = -(tj * (C.xx * v0 + C.yx * v1 + C.zx * v2));
      f.Gp[1] = -(tj * (C.xy * v0 + C.yy * v1 + C.zy * v2));
      f.Gp[2] = -(tj * (C.xz * v0 + C.yz * v1 + C.zz * v2));
    } else {
      f.Gp[0] = -(tau * g.grad_p[0]);
      f.Gp[1] = -(tau * g.grad_p[1]);
      f.Gp[2] = -(tau * g.grad_p[2]);
    }
    f.Gu = stress;
  }
  // the displacement flux alone (sigma, or PK1 = sigma cof F): R_u[n][i] = sum_j Gu[i][j] dN_n/dx_j w dv (:136-144)
  template <class T, class Local> C8_HD static Tens3<T> flux_u(Local const& local, PointState<T> const& g) {
    Tens3<T> stress = local.cauchy(g);
    if (Local::FINITE_DEF) stress = matmul(stress, cofactor(g.grad_u + eye3<T>()));
    return stress;
  }
  template <class T, class Local>
  C8_HD static T flux_pressure(Local const& local, PointState<T> const& g) {  // :215-223
    return -(g.p / local.pressure_scale_factor());
  }
};

// MechanicsPlaneStress::evaluate (mechanics_plane_stress.cpp:47-95): the momentum balance with the local model's in-plane
// Cauchy stress; under finite deformation PK1 = lambda_z J sigma F^-T with the model's out-of-plane stretch (:66-82);
// everything times the thickness (:90).  No pressure residual.
struct MechanicsPlaneStress {
  template <class T, class Local>
  C8_HD static void flux(Local const& local, PointState<T> const& g, double thickness, MechFlux<T>& f) {
    Tens3<T> stress = local.cauchy(g);
    if constexpr (Local::FINITE_DEF) {
      Tens3<T> const F = g.grad_u + eye3<T>();  // F_zz = 1: determinant and inverse are the 2 x 2 ones
      Tens3<T> const F_invT = transpose(inverse(F));
      T const J = det(F);
      stress = matmul(scale(local.xi[Local::Z_STRETCH] * J, stress), F_invT);
    }
    f.Gu = scale(thickness, stress);
    f.Vp = T(0.);
    f.Gp[0] = f.Gp[1] = f.Gp[2] = T(0.);
  }
};

// ---- QoI concept (qoi.hpp): value_pt of the objective integrand at a coupled point ---------------
// One integrand with run-time coefficients covers the reference's two point-wise objectives, so that the
// adjoint kernels are compiled once:
//   "average displacement" (avg_disp.cpp:16-33):       c_avg  * sum_i u_i w dv / ndims
//   the load term of "calibration" (calibration.cpp:302-343, :468-472): the internal force of the coupled weak
//   form summed over the element's nodes on the load plane, sum_n R_u[n][comp] = w dv Gu[comp][.] . S with
//   S = sum_n grad N_n, times c_load = balance * dt/T * load_mismatch.
// (The displacement-mismatch term of "calibration" lives on faces and is linear-quadratic in the nodal values:
// c8_primal.hip adds it and its derivative directly.)
struct QoiArgs {
  double c_avg, c_load;
  int comp;
  double const* S;  // [nelems][coupled points][3], null when c_load == 0
  double ndims = 3.;  // "average displacement" divides by the number of dimensions (avg_disp.cpp:27)
  double thickness = 1.;  // mechanics_plane_stress: the internal force of the load term carries it (mechanics_plane_stress.cpp:90)
};
template <class L, class = void> struct is_plane_stress : std::false_type {};
template <class L> struct is_plane_stress<L, std::enable_if_t<L::PLANE_STRESS>> : std::true_type {};
struct PointQoi {
  template <class T, class Local>
  C8_HD static T evaluate(PointState<T> const& g, Local const& local, double wdv, QoiArgs const& qa, size_t qp) {
    T v = (g.u[0] + g.u[1] + g.u[2]) * (qa.c_avg * wdv / qa.ndims);
    if (qa.c_load != 0.) {  // uniform over the launch
      Tens3<T> Gu;  // the displacement flux of the element's global residual (compute_load re-enters it, calibration.cpp:334)
      if constexpr (is_plane_stress<Local>::value) {
        MechFlux<T> f;
        MechanicsPlaneStress::flux(local, g, qa.thickness, f);
        Gu = f.Gu;
      } else {
        Gu = Mechanics::flux_u(local, g);
      }
      double const s0 = qa.S[qp * 3] * (qa.c_load * wdv), s1 = qa.S[qp * 3 + 1] * (qa.c_load * wdv),
                   s2 = qa.S[qp * 3 + 2] * (qa.c_load * wdv);
      T const r0 = Gu.xx * s0 + Gu.xy * s1 + Gu.xz * s2;
      T const r1 = Gu.yx * s0 + Gu.yy * s1 + Gu.yz * s2;
      T const r2 = Gu.zx * s0 + Gu.zy * s1 + Gu.zz * s2;
      v = v + (qa.comp == 0 ? r0 : (qa.comp == 1 ? r1 : r2));
    }
    return v;
  }
};

}  // namespace c8
