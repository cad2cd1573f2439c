// c8_assemble_adjoint.hpp -- K2..K6: residual-only assembly and the adjoint-sensitivity
// kernels, in the same SPMD form and lane mapping as c8_assemble.hpp (lane k of an
// element's group owns derivative slot k).
//
//   K2 residual_element            eval_global_residual   evaluations.cpp:156-259
//   K3 adjoint_jacobian_element    eval_adjoint_jacobian  evaluations.cpp:349-526
//   K4 adjoint_local_element       solve_adjoint_local    evaluations.cpp:528-659
//   K5 param_gradient_element      eval_qoi_gradient      evaluations.cpp:758-925
//   K6 qoi_element                 eval_qoi               evaluations.cpp:662-756
//
// Transposed products with the adjoint vectors never form the transposed matrices:
// lane k owns column k of each AD Jacobian, so (J^T v)[k] is a lane-local dot product.
#pragma once

#include "c8_assemble.hpp"

namespace c8 {

// own-row entry of the element residual from point fluxes (any scalar type)
template <class E, class SH, class T>
C8_HD double residual_entry(SH const& sh, int pt, int k, MechFlux<T> const& f) {
  double const wdv = sh.wdv[pt];
  int ik, nk, eqk;
  slot_to_dof<E>(k, ik, nk, eqk);
  double const d0 = sh.dN[pt][nk][0] * wdv, d1 = sh.dN[pt][nk][1] * wdv, d2 = sh.dN[pt][nk][2] * wdv;
  double const r0 = val(f.Gu.xx) * d0 + val(f.Gu.xy) * d1 + val(f.Gu.xz) * d2;
  double const r1 = val(f.Gu.yx) * d0 + val(f.Gu.yy) * d1 + val(f.Gu.yz) * d2;
  double const r2 = val(f.Gu.zx) * d0 + val(f.Gu.zy) * d1 + val(f.Gu.zz) * d2;
  double const rp = val(f.Vp) * (sh.N[pt][nk] * wdv) + val(f.Gp[0]) * d0 + val(f.Gp[1]) * d1 + val(f.Gp[2]) * d2;
  return (ik == 1) ? rp : (eqk == 0 ? r0 : (eqk == 1 ? r1 : r2));
}

// (dR/ds)^T z for the tangent direction s carried by the flux: w dv [ dGu : grad z_u + dVp z_p + dGp . grad z_p ]
template <class E, class SH>
C8_HD double flux_dot_adjoint(SH const& sh, int pt, MechFlux<Dual> const& f, bool coupled) {
  double zpv = 0., zpg[3] = {0., 0., 0.}, zg[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}};
  C8_UNROLL
  for (int n = 0; n < E::NN; ++n) {
    double const d0 = sh.dN[pt][n][0], d1 = sh.dN[pt][n][1], d2 = sh.dN[pt][n][2];
    if constexpr (E::NRES == 2) {
      double const zp = sh.z[E::DIM * E::NN + n];
      zpv += zp * sh.N[pt][n];
      zpg[0] += zp * d0; zpg[1] += zp * d1; zpg[2] += zp * d2;
    }
    C8_UNROLL
    for (int i = 0; i < E::DIM; ++i) {
      double const zu = sh.z[E::DIM * n + i];
      zg[i][0] += zu * d0; zg[i][1] += zu * d1; zg[i][2] += zu * d2;
    }
  }
  double s = f.Vp.d * zpv;
  if (coupled) {
    s += f.Gp[0].d * zpg[0] + f.Gp[1].d * zpg[1] + f.Gp[2].d * zpg[2];
    s += f.Gu.xx.d * zg[0][0] + f.Gu.xy.d * zg[0][1] + f.Gu.xz.d * zg[0][2];
    s += f.Gu.yx.d * zg[1][0] + f.Gu.yy.d * zg[1][1] + f.Gu.yz.d * zg[1][2];
    s += f.Gu.zx.d * zg[2][0] + f.Gu.zy.d * zg[2][1] + f.Gu.zz.d * zg[2][2];
  }
  return s * sh.wdv[pt];
}

template <class E, class EX, class SH>
C8_HD void load_adjoint(EX& ex, SH& sh, AdjointArgs const& aa) {
  ex.each([&](int k) {
    int ik, nk, eqk;
    slot_to_dof<E>(k, ik, nk, eqk);
    sh.z[k] = (ik == 0) ? aa.z_u[(size_t)sh.node[nk] * E::DIM + eqk] : aa.z_p[sh.node[nk]];
  });
  ex.sync();
}

// =====================================================================================
// K2: residual only, from the stored local state (no AD, no local solve).
// Deviation from the reference recorded in SURVEY.md section 10: the reference gathers local
// point 0 for every point (evaluations.cpp:227-229); with several coupled points per
// element this gathers point `pt`.  Identical on tet4.
// =====================================================================================
template <class E, template <class> class ModelT> struct ResidualLane {
  double Rk;
  ModelT<double> m;
  PointState<double> g;
};

template <class E, template <class> class ModelT, class EX>
C8_HD void residual_element(EX& ex, GroupShared<E, ModelT<Dual>::NLOC>& sh, MeshTables const& mt,
                            ModelSettings const& ms, FieldArgs const& fa, SystemArgs const& sa, int e) {
  using Model = ModelT<double>;
  constexpr int NL = Model::NLOC;
  constexpr bool PREV = Model::FINITE_DEF;
  load_element<E>(ex, sh, mt, fa, e, PREV);
  ex.each([&](int k) {
    auto& r = ex.lane(k);
    r.Rk = 0.;
    int const es = mt.elem_set ? mt.elem_set[e] : 0;
    C8_UNROLL
    for (int q = 0; q < Model::NPARAMS; ++q) r.m.params[q] = mt.params[es * Model::NPARAMS + q];
    if (k == 0) sh.h = elem_size<E>(sh);
  });
  ex.sync();
  for (int ip_set = 0; ip_set < E::NSETS; ++ip_set) {
    if (ip_set == 0 || !E::SAME_POINTS) shape_tables<E>(ex, sh, ip_set);
    int const npts = ip_set == 0 ? E::NP0 : E::NP1;
    for (int pt = 0; pt < npts; ++pt) {
      ex.each([&](int k) {
        auto& r = ex.lane(k);
        if (ip_set == 0) {
          size_t const q = ((size_t)e * E::NP0 + pt) * NL;
          interpolate_values<E, double, PREV>(sh, pt, r.g);
          C8_UNROLL
          for (int j = 0; j < NL; ++j) { r.m.xi[j] = fa.xi[q + j]; r.m.xi_prev[j] = fa.xi_prev[q + j]; }
          MechFlux<double> f;
          global_flux<E>(r.m, r.g, sh.h, ms, f);
          r.Rk += residual_entry<E>(sh, pt, k, f);
        } else if constexpr (E::NRES == 2) {
          interpolate_values<E, double, false>(sh, pt, r.g);
          int ik, nk, eqk;
          slot_to_dof<E>(k, ik, nk, eqk);
          double const Vp = Mechanics::flux_pressure(r.m, r.g);
          if (ik == 1) r.Rk += Vp * (sh.N[pt][nk] * sh.wdv[pt]);
        }
      });
    }
  }
  ex.sync();
  scatter_rhs<E>(ex, sh, sa, [&](int k) { return ex.lane(k).Rk; });
}

// =====================================================================================
// K3: adjoint Jacobian and right-hand side.  No local solve: the stored xi is used
// (evaluations.cpp:442-446).  QoI is a template parameter (the reference's QoI<T> plug-in).
// =====================================================================================
template <class E, template <class> class ModelT> struct AdjointLane : ForwardLane<E, ModelT> {
  double rhs;  // entry k of the element adjoint right-hand side
};

template <class E, template <class> class ModelT, class QoI, class EX>
C8_HD void adjoint_jacobian_element(EX& ex, GroupShared<E, ModelT<Dual>::NLOC>& sh, MeshTables const& mt,
                                    ModelSettings const& ms, FieldArgs const& fa, AdjointArgs const& aa,
                                    SystemArgs const& sa, int e) {
  using Model = ModelT<Dual>;
  constexpr int NL = Model::NLOC;
  constexpr bool PREV = Model::FINITE_DEF;
  load_element<E>(ex, sh, mt, fa, e, PREV);
  ex.each([&](int k) {
    auto& r = ex.lane(k);
    C8_UNROLL
    for (int a = 0; a < E::NDOF; ++a) r.Jcol[a] = 0.;
    r.rhs = 0.;
    r.failed = false;
    load_params(r.m, mt, e);
    if (k == 0) sh.h = elem_size<E>(sh);
  });
  ex.sync();
  for (int ip_set = 0; ip_set < E::NSETS; ++ip_set) {
    if (ip_set == 0 || !E::SAME_POINTS) shape_tables<E>(ex, sh, ip_set);
    int const npts = ip_set == 0 ? E::NP0 : E::NP1;
    for (int pt = 0; pt < npts; ++pt) {
      if (ip_set == 0) {
        size_t const qp = (size_t)e * E::NP0 + pt;
        // seed xi, evaluate -> dC/dxi ; unseed, seed x, evaluate -> dC/dx  (:442-453)
        ex.each([&](int k) {
          auto& r = ex.lane(k);
          interpolate_values<E, Dual, PREV>(sh, pt, r.g);
          C8_UNROLL
          for (int j = 0; j < NL; ++j) {
            r.m.xi_prev[j] = Dual(fa.xi_prev[qp * NL + j]);
            r.m.xi[j] = Dual(fa.xi[qp * NL + j], (j == k) ? 1. : 0.);
            r.m.R[j] = Dual(0.);
          }
          if (Model::HAS_LOCAL) {
            r.m.evaluate(r.g, ms.abs_tol);
            if (k < NL) {
              C8_UNROLL
              for (int j = 0; j < NL; ++j) sh.M[j][k] = r.m.R[j].d;
            }
          }
          C8_UNROLL
          for (int j = 0; j < NL; ++j) r.m.xi[j].d = 0.;
          seed_x<E>(sh, pt, k, r.g);
          C8_UNROLL
          for (int j = 0; j < NL; ++j) r.b[j] = 0.;
          if (Model::HAS_LOCAL) {
            r.m.evaluate(r.g, ms.abs_tol);
            C8_UNROLL
            for (int j = 0; j < NL; ++j) r.b[j] = -r.m.R[j].d;
          }
        });
        if (Model::HAS_LOCAL) {
          ex.sync();
          bool const ok = gj_solve<NL>(ex, sh, [&](int k) { return ex.lane(k).b; });
          ex.each([&](int k) { if (!ok) ex.lane(k).failed = true; });
        }
        // dtotal column k (:459-465), dJ/dx_k (:469-471), dJ/dxi_k (:474-478), g -= dJ/dxi (:481)
        ex.each([&](int k) {
          auto& r = ex.lane(k);
          C8_UNROLL
          for (int j = 0; j < NL; ++j) r.m.xi[j].d = r.b[j];
          MechFlux<Dual> f;
          global_flux<E>(r.m, r.g, sh.h, ms, f);
          double dummy = 0.;
          accumulate_coupled<E>(sh, pt, k, f, r.Jcol, dummy);
          C8_UNROLL
          for (int j = 0; j < NL; ++j) r.m.xi[j].d = 0.;
          double const dJ_dx = QoI::evaluate(r.g, r.m, sh.wdv[pt], aa.qoi, qp).d;
          unseed(r.g);
          C8_UNROLL
          for (int j = 0; j < NL; ++j) r.m.xi[j].d = (j == k) ? 1. : 0.;
          double const dJ_dxi = QoI::evaluate(r.g, r.m, sh.wdv[pt], aa.qoi, qp).d;
          C8_UNROLL
          for (int j = 0; j < NL; ++j) r.m.xi[j].d = 0.;
          if (k < NL) {
            double const gk = aa.g[qp * NL + k] - dJ_dxi;
            aa.g[qp * NL + k] = gk;
            sh.vec[k] = gk;
          }
          r.rhs += -dJ_dx + aa.f[qp * E::NDOF + k];
        });
        ex.sync();
        ex.each([&](int k) {  // rhs += (dxi/dx)^T g  (:486-487)
          auto& r = ex.lane(k);
          double s = 0.;
          C8_UNROLL
          for (int j = 0; j < NL; ++j) s += r.b[j] * sh.vec[j];
          r.rhs += s;
        });
      } else if constexpr (E::NRES == 2) {
        ex.each([&](int k) {
          auto& r = ex.lane(k);
          interpolate_values<E, Dual, false>(sh, pt, r.g);
          seed_x<E>(sh, pt, k, r.g);
          Dual const Vp = Mechanics::flux_pressure(r.m, r.g);
          double const wdv = sh.wdv[pt];
          C8_UNROLL
          for (int n = 0; n < E::NN; ++n) r.Jcol[E::DIM * E::NN + n] += Vp.d * (sh.N[pt][n] * wdv);
        });
      }
    }
  }
  ex.sync();
  if (E::NDOF <= 16) {
    // lane k holds column k of dtotal; the assembled matrix is dtotal^T.  Scattering column k as row k
    // would send the 16 lanes of an instruction to 16 different CSR rows (measured 2-3x slower), so
    // transpose through LDS: afterwards lane k holds row k of dtotal = column k of dtotal^T.
    constexpr int NJ = (E::NDOF <= 16) ? E::NDOF : 1;
    ex.each([&](int k) {
      auto& r = ex.lane(k);
      C8_UNROLL
      for (int a = 0; a < NJ; ++a) sh.JT[k % NJ][a] = r.Jcol[a];
    });
    ex.sync();
    ex.each([&](int k) {
      auto& r = ex.lane(k);
      C8_UNROLL
      for (int a = 0; a < NJ; ++a) r.Jcol[a] = sh.JT[a][k % NJ];
    });
    scatter_lhs<E>(ex, sh, mt, sa, e, false, [&](int k) { return ex.lane(k).Jcol; });
  } else {
    scatter_lhs<E>(ex, sh, mt, sa, e, true, [&](int k) { return ex.lane(k).Jcol; });
  }
  scatter_rhs<E>(ex, sh, sa, [&](int k) { return ex.lane(k).rhs; }, e);
  ex.each([&](int k) {
    if (k == 0 && ex.lane(k).failed) ex.flag(sa.status);
  });
}

// =====================================================================================
// K4: local adjoint phi and the history vectors f, g of the previous step.
// =====================================================================================
template <class E, template <class> class ModelT, class EX>
C8_HD void adjoint_local_element(EX& ex, GroupShared<E, ModelT<Dual>::NLOC>& sh, MeshTables const& mt,
                                 ModelSettings const& ms, FieldArgs const& fa, AdjointArgs const& aa,
                                 SystemArgs const& sa, int e) {
  using Model = ModelT<Dual>;
  constexpr int NL = Model::NLOC;
  constexpr bool PREV = Model::FINITE_DEF;
  load_element<E>(ex, sh, mt, fa, e, PREV);
  load_adjoint<E>(ex, sh, aa);
  ex.each([&](int k) {
    auto& r = ex.lane(k);
    r.failed = false;
    load_params(r.m, mt, e);
    if (k == 0) sh.h = elem_size<E>(sh);
  });
  ex.sync();
  shape_tables<E>(ex, sh, 0);
  for (int pt = 0; pt < E::NP0; ++pt) {
    size_t const qp = (size_t)e * E::NP0 + pt;
    if (!Model::HAS_LOCAL) {  // dC/dxi = 0: Eigen's rank-0 solve returns phi = 0, so f = g = 0
      ex.each([&](int k) {
        if (k < NL) { aa.phi[qp * NL + k] = 0.; aa.g[qp * NL + k] = 0.; }
        aa.f[qp * E::NDOF + k] = 0.;
      });
      continue;
    }
    // xi seeded: dR/dxi (global) and dC/dxi (local); phi = (dC/dxi)^-T (g - (dR/dxi)^T z)  (:613-625)
    ex.each([&](int k) {
      auto& r = ex.lane(k);
      interpolate_values<E, Dual, PREV>(sh, pt, r.g);
      C8_UNROLL
      for (int j = 0; j < NL; ++j) {
        r.m.xi_prev[j] = Dual(fa.xi_prev[qp * NL + j]);
        r.m.xi[j] = Dual(fa.xi[qp * NL + j], (j == k) ? 1. : 0.);
        r.m.R[j] = Dual(0.);
      }
      MechFlux<Dual> f;
      global_flux<E>(r.m, r.g, sh.h, ms, f);
      double const dRz = flux_dot_adjoint<E>(sh, pt, f, true);
      r.m.evaluate(r.g, ms.abs_tol);
      if (k < NL) {
        C8_UNROLL
        for (int j = 0; j < NL; ++j) sh.M[k][j] = r.m.R[j].d;  // transposed fill: row k = column k of dC/dxi
        sh.vec[k] = aa.g[qp * NL + k] - dRz;
      }
    });
    ex.sync();
    ex.each([&](int k) {
      auto& r = ex.lane(k);
      C8_UNROLL
      for (int j = 0; j < NL; ++j) r.b[j] = sh.vec[j];
    });
    bool const ok = gj_solve<NL>(ex, sh, [&](int k) { return ex.lane(k).b; });
    // x_prev seeded: f = -(dC/dx_prev)^T phi (:628-633); xi_prev seeded: g = -(dC/dxi_prev)^T phi (:636-642)
    ex.each([&](int k) {
      auto& r = ex.lane(k);
      if (!ok) r.failed = true;
      if (k == 0) {
        C8_UNROLL
        for (int j = 0; j < NL; ++j) aa.phi[qp * NL + j] = r.b[j];
      }
      C8_UNROLL
      for (int j = 0; j < NL; ++j) r.m.xi[j].d = 0.;
      double fk = 0.;
      if (PREV) {
        seed_x<E>(sh, pt, k, r.g, true);
        r.m.evaluate(r.g, ms.abs_tol);
        C8_UNROLL
        for (int j = 0; j < NL; ++j) fk -= r.m.R[j].d * r.b[j];
        unseed(r.g);
      }
      aa.f[qp * E::NDOF + k] = fk;
      C8_UNROLL
      for (int j = 0; j < NL; ++j) r.m.xi_prev[j].d = (j == k) ? 1. : 0.;
      r.m.evaluate(r.g, ms.abs_tol);
      double gk = 0.;
      C8_UNROLL
      for (int j = 0; j < NL; ++j) gk -= r.m.R[j].d * r.b[j];
      if (k < NL) aa.g[qp * NL + k] = gk;
    });
  }
  ex.each([&](int k) {
    if (k == 0 && ex.lane(k).failed) ex.flag(sa.status);
  });
}

// =====================================================================================
// K5: parameter gradient.  Lane k < n_active carries d/d(param active[k]); the per-lane
// sums are accumulated over the elements a group visits (acc) and added to grad once.
// =====================================================================================
template <class E, template <class> class ModelT> struct GradLane {
  ModelT<Dual> m;
  PointState<Dual> g;
  double acc;
  int slot;  // position in grad, or -1
};

template <class E, template <class> class ModelT, class QoI, class EX>
C8_HD void param_gradient_element(EX& ex, GroupShared<E, ModelT<Dual>::NLOC>& sh, MeshTables const& mt,
                                  ModelSettings const& ms, FieldArgs const& fa, AdjointArgs const& aa, int e) {
  using Model = ModelT<Dual>;
  constexpr int NL = Model::NLOC;
  constexpr bool PREV = Model::FINITE_DEF;
  load_element<E>(ex, sh, mt, fa, e, PREV);
  load_adjoint<E>(ex, sh, aa);
  int const es = mt.elem_set ? mt.elem_set[e] : 0;
  int32_t const* act = aa.active + es * 10;
  ex.each([&](int k) {
    auto& r = ex.lane(k);
    int const nact = act[1];
    int const mine = (k < nact) ? act[2 + k] : -1;
    // a group may move between element sets: flush the sum of the previous set
    int const slot = (k < nact) ? act[0] + k : -1;
    if (slot != r.slot) {
      if (r.slot >= 0) ex.add(aa.out + r.slot, r.acc, 1);
      r.acc = 0.;
      r.slot = slot;
    }
    C8_UNROLL
    for (int q = 0; q < Model::NPARAMS; ++q)  // seed_wrt_params, local_residual.cpp:812-819
      r.m.params[q] = Dual(mt.params[es * Model::NPARAMS + q], (q == mine) ? 1. : 0.);
    if (k == 0) sh.h = elem_size<E>(sh);
  });
  ex.sync();
  for (int ip_set = 0; ip_set < E::NSETS; ++ip_set) {
    if (ip_set == 0 || !E::SAME_POINTS) shape_tables<E>(ex, sh, ip_set);
    int const npts = ip_set == 0 ? E::NP0 : E::NP1;
    for (int pt = 0; pt < npts; ++pt) {
      ex.each([&](int k) {
        auto& r = ex.lane(k);
        if (r.slot < 0) return;
        if (ip_set == 0) {
          size_t const qp = (size_t)e * E::NP0 + pt;
          interpolate_values<E, Dual, PREV>(sh, pt, r.g);
          C8_UNROLL
          for (int j = 0; j < NL; ++j) {
            r.m.xi_prev[j] = Dual(fa.xi_prev[qp * NL + j]);
            r.m.xi[j] = Dual(fa.xi[qp * NL + j]);
            r.m.R[j] = Dual(0.);
          }
          r.m.evaluate(r.g, ms.abs_tol);
          double s = 0.;
          C8_UNROLL
          for (int j = 0; j < NL; ++j) s += r.m.R[j].d * aa.phi[qp * NL + j];  // (dC/dp)^T phi (:864-866)
          s += QoI::evaluate(r.g, r.m, sh.wdv[pt], aa.qoi, qp).d;               // dJ/dp (:869-871)
          MechFlux<Dual> f;
          global_flux<E>(r.m, r.g, sh.h, ms, f);
          s += flux_dot_adjoint<E>(sh, pt, f, true);                         // (dR/dp)^T z (:883-886)
          r.acc += s;
        } else if constexpr (E::NRES == 2) {
          interpolate_values<E, Dual, false>(sh, pt, r.g);
          MechFlux<Dual> f;
          f.Vp = Mechanics::flux_pressure(r.m, r.g);
          r.acc += flux_dot_adjoint<E>(sh, pt, f, false);
        }
      });
    }
  }
  ex.sync();
}

template <class EX> C8_HD void param_gradient_flush(EX& ex, AdjointArgs const& aa) {
  ex.each([&](int k) {
    auto& r = ex.lane(k);
    if (r.slot >= 0) ex.add(aa.out + r.slot, r.acc, 1);
    r.slot = -1;
    r.acc = 0.;
  });
}

// =====================================================================================
// K6: QoI value.  Lane pt of the group evaluates coupled point pt; sums like K5.
// =====================================================================================
template <class E, template <class> class ModelT> struct QoiLane {
  ModelT<double> m;
  PointState<double> g;
  double acc;
};

template <class E, template <class> class ModelT, class QoI, class EX>
C8_HD void qoi_element(EX& ex, GroupShared<E, ModelT<Dual>::NLOC>& sh, MeshTables const& mt, FieldArgs const& fa,
                       QoiArgs const& qa, int e) {
  using Model = ModelT<double>;
  constexpr int NL = Model::NLOC;
  load_element<E>(ex, sh, mt, fa, e, false);
  shape_tables<E>(ex, sh, 0);
  ex.each([&](int k) {
    auto& r = ex.lane(k);
    if (k >= E::NP0) return;
    int const es = mt.elem_set ? mt.elem_set[e] : 0;
    C8_UNROLL
    for (int q = 0; q < Model::NPARAMS; ++q) r.m.params[q] = mt.params[es * Model::NPARAMS + q];
    size_t const qp = (size_t)e * E::NP0 + k;
    interpolate_values<E, double, false>(sh, k, r.g);
    C8_UNROLL
    for (int j = 0; j < NL; ++j) {
      r.m.xi[j] = fa.xi ? fa.xi[qp * NL + j] : 0.;
      r.m.xi_prev[j] = fa.xi_prev ? fa.xi_prev[qp * NL + j] : 0.;
    }
    r.acc += QoI::evaluate(r.g, r.m, sh.wdv[k], qa, qp);
  });
  ex.sync();
}

template <class E, class EX> C8_HD void qoi_flush(EX& ex, double* out) {
  ex.each([&](int k) {
    auto& r = ex.lane(k);
    if (k < E::NP0) ex.add(out, r.acc, 1);
    r.acc = 0.;
  });
}

}  // namespace c8
