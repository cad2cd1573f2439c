// c8_assemble.hpp -- per-element assembly algorithms, written once in SPMD form.
//
// An element is processed by a LANE GROUP of E::NDOF lanes (32 for hex8, 16 for
// tet4): lane k of the group owns derivative slot k of the element's forward-AD
// dual numbers (slot order = element DOF order, global_residual.cpp:21-23), so
// after the point loop lane k holds column k of the element Jacobian dR/dx and
// entry k of the element residual.  A 64-lane wavefront carries 64/NDOF groups.
//
// The algorithm is the reference's per-point sequence (evaluations.cpp:79-134):
//   local Newton solve with xi seeded      -> converged xi, dC/dxi
//   re-evaluate with x seeded              -> dC/dx
//   dxi/dx = -(dC/dxi)^-1 dC/dx            -> chain-rule seeds for xi
//   weak-form integrand with both seeded   -> R_e, dR_e/dx
// with three implementation differences that do not change the mathematics:
// the weak form is evaluated as point fluxes and contracted with grad N
// (c8_models.hpp), element contributions are summed over points before the
// single scatter, and the 7x7/8x8 solves are one cooperative Gauss-Jordan with
// row pivoting per group instead of a full-pivot LU per right-hand side.
//
// The code is generic over an executor `EX` so that exactly this source runs
//   * inside the HIP kernels (c8_kernels.hip): each() runs the body once for
//     the calling lane, shared state lives in LDS, sync() is a wave-level fence;
//   * in tests/emul (CPU): each() loops the lanes of one group serially.
// The CPU instantiation exists only to unit-test kernel logic without a GPU;
// the product never executes it.
#pragma once

#include <stdint.h>

#include "c8_element.hpp"
#include "c8_models.hpp"

namespace c8 {

// ---- arguments shared by all assembly kernels (device pointers on the GPU) ----
struct MeshTables {
  int32_t const* conn;       // [nelems][NN]
  double const* coords;      // [nnodes][3]
  int32_t const* nodeptr;    // [nnodes+1] node-graph row offsets (sorted neighbour lists)
  uint8_t const* pos;        // [nelems][NN(col node)][NN(row node)] position of col node in row node's list
  int32_t const* elem_set;   // [nelems] or null (single set)
  int32_t const* order;      // element processing order (colour-sorted) or null
  double const* params;      // [nsets][NPARAMS]
  // cached shape tables of the wave kernels (hex8), [nelems][SHAPE_STRIDE], or null: the geometry is static, so
  // dN/dx, w dv and the element size are computed once per context (store_shape_tables) instead of per call
  double const* shape = nullptr;
};
struct ModelSettings {
  double stab_mult;
  double abs_tol, rel_tol;
  int max_iters;
  double thickness = 1.;  // mechanics_plane_stress.cpp:22
  // `line search:` sublist of the local residual (line_search.hpp:28-49; Hosford / Barlat models)
  double ls_c1 = 1.e-4, ls_bmin = 0.5, ls_bmax = 0.9;
  int ls_max_evals = 4;
  int closed_form = 0;  // forward wave kernel: the model's closed form where it has one (c8_set_kernel_variant)
  int closed_form_slot = 0;  // the same in the lane-group kernel (C8_KERNEL_AUTO only: an explicit C8_KERNEL_SLOT iterates)
};
struct FieldArgs {
  double const* u;        // [nnodes][3]
  double const* p;        // [nnodes]; not read under mechanics_plane_stress (one residual)
  double const* u_prev;
  double const* p_prev;
  double const* xi_prev;  // [nelems][NP0][NLOC]
  double* xi;             // [nelems][NP0][NLOC]
};
// extra arguments of the adjoint kernels (K3-K6)
struct AdjointArgs {
  double* g;           // local history  [nelems][NP0][NLOC]   (adjoint.cpp:52-74)
  double* f;           // global history [nelems][NP0][NDOF]
  double const* z_u;   // global adjoint solution, u block [nnodes][3]
  double const* z_p;   // p block [nnodes]
  double* phi;         // local adjoint [nelems][NP0][NLOC]
  double* out;         // grad [n_active] (K5) or J [1] (K6)
  int32_t const* active;  // [nsets][2 + 8]: {offset into grad, n_active, param indices...}
  QoiArgs qoi;            // objective integrand (default: average displacement)
};
struct SystemArgs {
  double* A[2][2];  // CSR values of the four blocks
  double* b[2];     // residual vectors
  int* status;      // device int: set nonzero when a local Newton solve fails
  int atomic;       // 1: atomic adds; 0: plain read-modify-write (colour-batched launch)
  double* stage;    // not null: element matrices are stored element-major here and summed by gather_node_rows
  int stage_ring;   // element slots in the stage: element e uses slot e % stage_ring
#ifdef C8_STAMPS
  unsigned long long* stamps;  // diagnostic build only: [4096][16] s_memtime stamps of sampled elements
#endif
};

// Staged assembly (scatter mode GATHER): the assembly kernels store every element matrix, as it stands in
// registers, to stage[e][STAGE_STRIDE] with fully coalesced stores; gather_node_rows then sums the rows of each
// node over the node's elements in a fixed order and adds them to the CSR values.  No atomics, no colouring,
// bitwise reproducible.
struct GatherArgs {
  int32_t const* nodeptr;       // node graph
  uint8_t const* pos;           // [nelems][NN(col node)][NN(row node)]
  int32_t const* nodeelem_ptr;  // [nnodes+1]
  int32_t const* nodeelem;      // (element << 3) | local node
  double const* stage;          // [stage_ring][stage_stride]
  int stage_ring;               // element e is in slot e % stage_ring
  int32_t const* node_order;    // nodes to process (StagePlan::node_order)
  double* A[2][2];
  double* b[2];
  int assign = 0;               // 1: the rows are assigned (A = sum, b = sum) instead of added to (c8_set_assign_mode)
#ifdef C8_STAMPS
  unsigned long long* stamps = nullptr;  // diagnostic build only: [4096][16] s_memtime stamps of sampled nodes (row-per-node kernel)
#endif
};
// stage[e]: for every element node n the four rows (u_0, u_1, u_2, p) of that node, NDOF columns each, in element
// DOF order -- the 4*NDOF values a node's row sum needs from this element are contiguous -- then the NDOF entries of
// the element residual.
template <class E> constexpr int stage_stride() { return E::NN * 4 * E::NDOF + E::NDOF; }
template <class E> C8_HD int stage_row(int n, int rr) { return (n * 4 + rr) * E::NDOF; }  // rr: 0..2 = u_rr, 3 = p
constexpr int GATHER_MAX_DEGREE = 64;  // node-graph rows the gather kernel's LDS accumulator can hold

// ---- per-group shared scratch (LDS) -------------------------------------------
template <class E, int NL> struct GroupShared {
  static constexpr int NPT = (E::NP0 > E::NP1) ? E::NP0 : E::NP1;
  double X[E::NN][3];
  double u[E::NN][3], p[E::NN];
  double u_prev[E::NN][3];
  double N[NPT][E::NN];
  double dN[NPT][E::NN][3];
  double wdv[NPT];
  double M[NL][NL + 1];
  // adjoint assembly of small elements: element matrix transposed through LDS before the scatter, so
  // that the lanes of one scatter instruction share a CSR row (hex8 uses the wave kernel instead)
  static constexpr int NJT = (E::NDOF <= 16) ? E::NDOF : 1;
  double JT[NJT][NJT + 1];
  double vec[NL + 1];      // right-hand side / history exchange between lanes
  double z[E::NDOF];       // element adjoint solution (gather_adjoint, global_residual.cpp:423-438)
  double h;
  int32_t node[E::NN];
  int32_t nptr[E::NN], deg[E::NN];
};

// ---- shape tables: N, dN/dx, w*detJ for every point of an ip set ---------------
template <class E, class SH> C8_HD void shape_entry(SH& sh, int ip_set, int pt, int n0, int n1) {
  double xi[3], w;
  E::point(ip_set, pt, xi, w);
  // J(a,b) = d x_b / d xi_a
  double J[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}};
  C8_UNROLL
  for (int n = 0; n < E::NN; ++n) {
    double g[3];
    E::dNdxi(n, xi, g);
    C8_UNROLL
    for (int a = 0; a < 3; ++a)
      C8_UNROLL
      for (int b = 0; b < 3; ++b) J[a][b] += g[a] * sh.X[n][b];
  }
  if (E::DIM == 2) J[2][2] = 1.;  // 2-D: the in-plane 2 x 2 Jacobian, completed by the unit out-of-plane direction
  Tens3<double> Jt;
  Jt.xx = J[0][0]; Jt.xy = J[0][1]; Jt.xz = J[0][2];
  Jt.yx = J[1][0]; Jt.yy = J[1][1]; Jt.yz = J[1][2];
  Jt.zx = J[2][0]; Jt.zy = J[2][1]; Jt.zz = J[2][2];
  double const dJ = det(Jt);
  Tens3<double> const Ji = inverse(Jt);
  if (n0 == 0) sh.wdv[pt] = w * dJ;
  C8_UNROLL
  for (int n = n0; n < n1; ++n) {
    double g[3];
    E::dNdxi(n, xi, g);
    sh.N[pt][n] = E::N(n, xi);
    sh.dN[pt][n][0] = Ji.xx * g[0] + Ji.xy * g[1] + Ji.xz * g[2];
    sh.dN[pt][n][1] = Ji.yx * g[0] + Ji.yy * g[1] + Ji.yz * g[2];
    sh.dN[pt][n][2] = Ji.zx * g[0] + Ji.zy * g[1] + Ji.zz * g[2];
  }
}

template <class E, class EX, class SH> C8_HD void shape_tables(EX& ex, SH& sh, int ip_set) {
  int const npts = ip_set == 0 ? E::NP0 : E::NP1;
  int const lanes_per_pt = E::NDOF / npts;                                // hex8: 4, tet4: 16 / 4
  int const nodes_per_lane = (E::NN + lanes_per_pt - 1) / lanes_per_pt;   // hex8: 2, tet4: 1
  ex.each([&](int k) {
    int const pt = k / lanes_per_pt, sub = k % lanes_per_pt;
    int const n0 = sub * nodes_per_lane;
    int const n1 = (n0 + nodes_per_lane < E::NN) ? n0 + nodes_per_lane : E::NN;
    if (pt < npts && n0 < E::NN) shape_entry<E>(sh, ip_set, pt, n0, n1);
  });
  ex.sync();
}

// mean-square edge length, mechanics.cpp:103-113
template <class E, class SH> C8_HD double elem_size(SH const& sh) {
  double h = 0.;
  C8_UNROLL
  for (int e = 0; e < E::NEDGES; ++e) {
    int a, b;
    E::edge(e, a, b);
    double const dx = sh.X[b][0] - sh.X[a][0], dy = sh.X[b][1] - sh.X[a][1], dz = sh.X[b][2] - sh.X[a][2];
    h += dx * dx + dy * dy + dz * dz;
  }
  return sqrt(h / E::NEDGES);
}

// ---- interpolation (global_residual.cpp:289-332) --------------------------------
// Values are the same in every lane; the tangent of lane k (x seeded along element
// DOF k) is a single shape-function entry, no sum needed.
template <class E, class T, bool PREV, class SH>
C8_HD void interpolate_values(SH const& sh, int pt, PointState<T>& g) {
  double u[3] = {0., 0., 0.}, gu[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}};
  double p = 0., gp[3] = {0., 0., 0.};
  double gup[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}};
  C8_UNROLL
  for (int n = 0; n < E::NN; ++n) {
    double const Nn = sh.N[pt][n];
    double const d0 = sh.dN[pt][n][0], d1 = sh.dN[pt][n][1], d2 = sh.dN[pt][n][2];
    C8_UNROLL
    for (int i = 0; i < 3; ++i) {
      double const un = sh.u[n][i];
      u[i] += un * Nn;
      gu[i][0] += un * d0; gu[i][1] += un * d1; gu[i][2] += un * d2;
      if (PREV) {
        double const upn = sh.u_prev[n][i];
        gup[i][0] += upn * d0; gup[i][1] += upn * d1; gup[i][2] += upn * d2;
      }
    }
    double const pn = sh.p[n];
    p += pn * Nn;
    gp[0] += pn * d0; gp[1] += pn * d1; gp[2] += pn * d2;
  }
  C8_UNROLL
  for (int i = 0; i < 3; ++i) { g.u[i] = T(u[i]); g.grad_p[i] = T(gp[i]); }
  g.p = T(p);
  g.grad_u.xx = T(gu[0][0]); g.grad_u.xy = T(gu[0][1]); g.grad_u.xz = T(gu[0][2]);
  g.grad_u.yx = T(gu[1][0]); g.grad_u.yy = T(gu[1][1]); g.grad_u.yz = T(gu[1][2]);
  g.grad_u.zx = T(gu[2][0]); g.grad_u.zy = T(gu[2][1]); g.grad_u.zz = T(gu[2][2]);
  g.grad_u_prev.xx = T(gup[0][0]); g.grad_u_prev.xy = T(gup[0][1]); g.grad_u_prev.xz = T(gup[0][2]);
  g.grad_u_prev.yx = T(gup[1][0]); g.grad_u_prev.yy = T(gup[1][1]); g.grad_u_prev.yz = T(gup[1][2]);
  g.grad_u_prev.zx = T(gup[2][0]); g.grad_u_prev.zy = T(gup[2][1]); g.grad_u_prev.zz = T(gup[2][2]);
}

// seed_wrt_x for lane k (global_residual.cpp:206-216): d(interpolant)/d(x_k)
template <class E, class SH> C8_HD void seed_x(SH const& sh, int pt, int k, PointState<Dual>& g, bool prev = false) {
  int i, n, eq;
  slot_to_dof<E>(k, i, n, eq);
  double const Nn = sh.N[pt][n];
  double const d0 = sh.dN[pt][n][0], d1 = sh.dN[pt][n][1], d2 = sh.dN[pt][n][2];
  bool const isu = (i == 0), isp = (i == 1);
  Tens3<Dual>& G = prev ? g.grad_u_prev : g.grad_u;
  if (!prev) {
    g.u[0].d = (isu && eq == 0) ? Nn : 0.;
    g.u[1].d = (isu && eq == 1) ? Nn : 0.;
    g.u[2].d = (isu && eq == 2) ? Nn : 0.;
    g.p.d = isp ? Nn : 0.;
    g.grad_p[0].d = isp ? d0 : 0.;
    g.grad_p[1].d = isp ? d1 : 0.;
    g.grad_p[2].d = isp ? d2 : 0.;
  }
  G.xx.d = (isu && eq == 0) ? d0 : 0.; G.xy.d = (isu && eq == 0) ? d1 : 0.; G.xz.d = (isu && eq == 0) ? d2 : 0.;
  G.yx.d = (isu && eq == 1) ? d0 : 0.; G.yy.d = (isu && eq == 1) ? d1 : 0.; G.yz.d = (isu && eq == 1) ? d2 : 0.;
  G.zx.d = (isu && eq == 2) ? d0 : 0.; G.zy.d = (isu && eq == 2) ? d1 : 0.; G.zz.d = (isu && eq == 2) ? d2 : 0.;
}
C8_HD void unseed(PointState<Dual>& g) {
  C8_UNROLL
  for (int i = 0; i < 3; ++i) { g.u[i].d = 0.; g.grad_p[i].d = 0.; }
  g.p.d = 0.;
  Tens3<Dual>* t[2] = {&g.grad_u, &g.grad_u_prev};
  C8_UNROLL
  for (int q = 0; q < 2; ++q) {
    t[q]->xx.d = 0.; t[q]->xy.d = 0.; t[q]->xz.d = 0.;
    t[q]->yx.d = 0.; t[q]->yy.d = 0.; t[q]->yz.d = 0.;
    t[q]->zx.d = 0.; t[q]->zy.d = 0.; t[q]->zz.d = 0.;
  }
}

// ---- cooperative Gauss-Jordan with row pivoting ----------------------------------
// M (NL x NL, in group-shared memory) is destroyed; every lane passes its own
// right-hand side b[NL] (registers) and gets its solution back in b.  Lane c < NL
// owns column c of M.  Replaces Eigen fullPivLu().solve() (evaluations.cpp:112,
// small_J2.cpp:157); returns false on a zero pivot.
template <int NL, class EX, class SH, class GetB>
C8_HD bool gj_solve(EX& ex, SH& sh, GetB getb) {
  bool ok = true;
  C8_UNROLL
  for (int s = 0; s < NL; ++s) {
    ex.each([&](int k) {
      double* b = getb(k);
      double col[NL];
      C8_UNROLL
      for (int r = 0; r < NL; ++r) col[r] = sh.M[r][s];
      int rstar = s;
      double big = fabs(col[s]);
      C8_UNROLL
      for (int r = s + 1; r < NL; ++r) {
        double const a = fabs(col[r]);
        if (a > big) { big = a; rstar = r; }
      }
      if (!(big > 0.)) ok = false;
      // swap rows s <-> rstar in this lane's copies (compile-time slots, run-time rstar)
      double const cs = col[s], bs = b[s];
      double cpiv = cs, bpiv = bs;
      static_for<NL>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        bool const hit = (r > s) && (r == rstar);
        cpiv = hit ? col[r] : cpiv;
        bpiv = hit ? b[r] : bpiv;
        col[r] = hit ? cs : col[r];
        b[r] = hit ? bs : b[r];
      });
      double const inv = 1. / cpiv;
      double const bsn = bpiv * inv;
      b[s] = bsn;
      C8_UNROLL
      for (int r = 0; r < NL; ++r) if (r != s) b[r] -= col[r] * bsn;
      if (k > s && k < NL) {  // column owner updates M[:, k]
        double const ms = sh.M[s][k], mr = sh.M[rstar][k];
        sh.M[rstar][k] = ms;
        double const msn = mr * inv;
        sh.M[s][k] = msn;
        C8_UNROLL
        for (int r = 0; r < NL; ++r) if (r != s) sh.M[r][k] -= col[r] * msn;
      }
    });
    ex.sync();
  }
  return ok;
}

// ---- lane state for the forward (primal) assembly -----------------------------------
template <class E, template <class> class ModelT> struct ForwardLane {
  using Model = ModelT<Dual>;
  double Jcol[E::NDOF];  // column k of dR_e/dx
  double Rk;             // entry k of R_e
  Model m;
  PointState<Dual> g;
  double b[Model::NLOC];
  int iter;
  double R_norm_0;
  bool converged, failed;
  // models with a local line search (uses_line_search): branch of the first evaluation, state of the search
  int path;
  double dxi[Model::NLOC];
  double ls_alpha, ls_applied, ls_best_alpha, ls_best_phi, ls_phi0, ls_phi;
  int ls_n;
  bool ls_done;
};

template <class Model> C8_HD void load_params(Model& m, MeshTables const& mt, int e) {
  int const es = mt.elem_set ? mt.elem_set[e] : 0;
  C8_UNROLL
  for (int q = 0; q < Model::NPARAMS; ++q) m.params[q] = Dual(mt.params[es * Model::NPARAMS + q]);
}

template <class E, class EX, class SH>
C8_HD void load_element(EX& ex, SH& sh, MeshTables const& mt, FieldArgs const& fa, int e, bool prev) {
  ex.each([&](int k) {
    int i, n, eq;
    slot_to_dof<E>(k, i, n, eq);
    int const node = mt.conn[e * E::NN + n];
    if (i == 0) {
      sh.X[n][eq] = mt.coords[(size_t)node * 3 + eq];
      sh.u[n][eq] = fa.u[(size_t)node * E::DIM + eq];
      if (prev) sh.u_prev[n][eq] = fa.u_prev[(size_t)node * E::DIM + eq];
    }
    // once per node: by its pressure slot, or (one residual: no such slot) by its first displacement slot
    if ((E::NRES == 2) ? (i == 1) : (eq == 0)) {
      if (E::DIM == 2) {  // out-of-plane entries of the 3-wide containers
        sh.X[n][2] = 0.;
        sh.u[n][2] = 0.;
        if (prev) sh.u_prev[n][2] = 0.;
      }
      sh.p[n] = (E::NRES == 2) ? fa.p[node] : 0.;
      sh.node[n] = node;
      if (mt.nodeptr) {
        int const a = mt.nodeptr[node];
        sh.nptr[n] = a;
        sh.deg[n] = mt.nodeptr[node + 1] - a;
      }
    }
  });
  ex.sync();
}

// the coupled point fluxes of the element's global residual: `mechanics`, or `mechanics_plane_stress` on Tri3PlaneStress
template <class E, class T, class Local>
C8_HD void global_flux(Local const& local, PointState<T> const& g, double h, ModelSettings const& ms, MechFlux<T>& f) {
  if constexpr (E::NRES == 2) Mechanics::flux_coupled(local, g, h, ms.stab_mult, f);
  else MechanicsPlaneStress::flux(local, g, ms.thickness, f);
}

// add the weak-form contribution of one point to lane k's Jacobian column / residual entry
template <class E, class SH>
C8_HD void accumulate_coupled(SH const& sh, int pt, int k, MechFlux<Dual> const& f, double* Jcol, double& Rk) {
  double const wdv = sh.wdv[pt];
  int ik, nk, eqk;
  slot_to_dof<E>(k, ik, nk, eqk);
  constexpr int D = E::DIM;
  C8_UNROLL
  for (int n = 0; n < E::NN; ++n) {
    double const d0 = sh.dN[pt][n][0] * wdv, d1 = sh.dN[pt][n][1] * wdv, d2 = sh.dN[pt][n][2] * wdv;
    // fused multiply-adds chained through the accumulator (one instruction per product)
    Jcol[D * n + 0] = fma(f.Gu.xz.d, d2, fma(f.Gu.xy.d, d1, fma(f.Gu.xx.d, d0, Jcol[D * n + 0])));
    Jcol[D * n + 1] = fma(f.Gu.yz.d, d2, fma(f.Gu.yy.d, d1, fma(f.Gu.yx.d, d0, Jcol[D * n + 1])));
    if (D == 3) Jcol[D * n + 2] = fma(f.Gu.zz.d, d2, fma(f.Gu.zy.d, d1, fma(f.Gu.zx.d, d0, Jcol[D * n + 2])));
    if constexpr (E::NRES == 2)
      Jcol[D * E::NN + n] = fma(f.Gp[2].d, d2, fma(f.Gp[1].d, d1, fma(f.Gp[0].d, d0, fma(f.Vp.d, sh.N[pt][n] * wdv, Jcol[D * E::NN + n]))));
  }
  double const d0 = sh.dN[pt][nk][0] * wdv, d1 = sh.dN[pt][nk][1] * wdv, d2 = sh.dN[pt][nk][2] * wdv;
  double const r0 = f.Gu.xx.v * d0 + f.Gu.xy.v * d1 + f.Gu.xz.v * d2;
  double const r1 = f.Gu.yx.v * d0 + f.Gu.yy.v * d1 + f.Gu.yz.v * d2;
  double const r2 = f.Gu.zx.v * d0 + f.Gu.zy.v * d1 + f.Gu.zz.v * d2;
  double const rp = f.Vp.v * (sh.N[pt][nk] * wdv) + f.Gp[0].v * d0 + f.Gp[1].v * d1 + f.Gp[2].v * d2;
  Rk += (ik == 1) ? rp : (eqk == 0 ? r0 : (eqk == 1 ? r1 : r2));
}

// ---- scatter (global_residual.cpp:463-479 scatter_rhs, :556-586 scatter_lhs) ------
// The CSR position of (row dof, col dof) follows from the node graph alone, because
// every block stores all equations of a neighbour node contiguously and sorted:
//   rowptr_ij[n*neq_i + eq_i] = nodeptr[n]*neq_i*neq_j + eq_i*deg[n]*neq_j
//   offset = rowptr + pos(row node, col node)*neq_j + eq_j
// so the reference's 4 KB/element scatter_offsets table (disc.cpp:414-459) is replaced
// by 64 bytes of positions per element.  If `transpose`, lane k holds ROW k.
template <class E, class EX, class SH, class GetJ>
C8_HD void scatter_lhs(EX& ex, SH const& sh, MeshTables const& mt, SystemArgs const& sa, int e, bool transpose, GetJ getj) {
  if (sa.stage && !transpose) {  // staged assembly: column k of the element matrix into the element's stage
    ex.each([&](int k) {
      double const* Jc = getj(k);
      double* const st = sa.stage + (size_t)(e % sa.stage_ring) * stage_stride<E>();
      C8_UNROLL
      for (int a = 0; a < E::NDOF; ++a) {
        int ia, na, eqa;
        slot_to_dof<E>(a, ia, na, eqa);
        st[stage_row<E>(na, ia == 0 ? eqa : 3) + k] = Jc[a];
      }
    });
    return;
  }
  ex.each([&](int k) {
    double const* Jc = getj(k);
    int ik, nk, eqk;
    slot_to_dof<E>(k, ik, nk, eqk);
    int const neqk = ik == 0 ? E::DIM : 1;
    uint8_t const* posk = mt.pos + ((size_t)e * E::NN + nk) * E::NN;  // pos[e][col node nk][row node]
    C8_UNROLL
    for (int a = 0; a < E::NDOF; ++a) {
      int ia, na, eqa;
      slot_to_dof<E>(a, ia, na, eqa);
      int const neqa = ia == 0 ? E::DIM : 1;
      size_t off;
      double* vals;
      if (!transpose) {  // entry (row a, col k)
        off = (size_t)sh.nptr[na] * (neqa * neqk) + (size_t)eqa * sh.deg[na] * neqk + (size_t)posk[na] * neqk + eqk;
        vals = sa.A[ia][ik];
      } else {           // entry (row k, col a): position of node na in row node nk's list
        uint8_t const pka = mt.pos[((size_t)e * E::NN + na) * E::NN + nk];
        off = (size_t)sh.nptr[nk] * (neqk * neqa) + (size_t)eqk * sh.deg[nk] * neqa + (size_t)pka * neqa + eqa;
        vals = sa.A[ik][ia];
      }
      ex.add(vals + off, Jc[a], sa.atomic);
    }
  });
}

template <class E, class EX, class SH, class GetR>
C8_HD void scatter_rhs(EX& ex, SH const& sh, SystemArgs const& sa, GetR getr, int e = -1) {
  if (sa.stage && e >= 0) {  // staged assembly: the element residual behind the element matrix
    ex.each([&](int k) { sa.stage[(size_t)(e % sa.stage_ring) * stage_stride<E>() + E::NN * 4 * E::NDOF + k] = getr(k); });
    return;
  }
  ex.each([&](int k) {
    int ik, nk, eqk;
    slot_to_dof<E>(k, ik, nk, eqk);
    int const neqk = ik == 0 ? E::DIM : 1;
    ex.add(sa.b[ik] + (size_t)sh.node[nk] * neqk + eqk, getr(k), sa.atomic);
  });
}

// ---- local Newton iteration with line search (small_hosford.cpp:147-218, hypo_hosford.cpp:183-254, hypo_barlat.cpp:
// 353-432): the branch is chosen by the first evaluation and forced afterwards; every Newton step is followed by the
// backtracking search of line_search.hpp:85-135 on the merit 1/2 |C|^2 with slope C . (J dxi).  All values are
// replicated over the lanes of a group, so every lane takes the same decisions; lane c < NL holds column c of J = dC/dxi
// in its tangents and hands it to the group through sh.M.  On exit sh.M holds dC/dxi of the last evaluation at the
// converged state, as after the plain iteration.
template <int NL, class EX, class SH>
C8_HD void local_newton_line_search(EX& ex, SH& sh, ModelSettings const& ms) {
  auto active = [&](int k) { auto& r = ex.lane(k); return (r.iter <= ms.max_iters) && !r.converged; };
  auto searching = [&](int k) { return active(k) && !ex.lane(k).ls_done; };
  while (ex.any(active)) {
    ex.each([&](int k) {
      auto& r = ex.lane(k);
      if (!active(k)) return;
      if (r.iter == 1) r.path = r.m.evaluate(r.g, ms.abs_tol);
      else r.m.evaluate(r.g, ms.abs_tol, true, r.path);
      double nrm = 0.;
      C8_UNROLL
      for (int j = 0; j < NL; ++j) nrm += r.m.R[j].v * r.m.R[j].v;
      double const C_norm = sqrt(nrm);
      if (r.iter == 1) r.R_norm_0 = C_norm;
      double const C_norm_rel = C_norm / r.R_norm_0;
      if ((C_norm_rel < ms.rel_tol) || (C_norm < ms.abs_tol)) r.converged = true;
      if (k < NL) C8_UNROLL for (int j = 0; j < NL; ++j) sh.M[j][k] = r.m.R[j].d;
      C8_UNROLL
      for (int j = 0; j < NL; ++j) r.b[j] = -r.m.R[j].v;
      r.ls_phi0 = 0.5 * C_norm * C_norm;
    });
    ex.sync();
    if (!ex.any(active)) break;
    bool const ok = gj_solve<NL>(ex, sh, [&](int k) { return ex.lane(k).b; });
    ex.each([&](int k) {
      auto& r = ex.lane(k);
      if (!active(k)) return;
      if (!ok) { r.failed = true; r.iter = ms.max_iters + 1; r.ls_done = true; return; }
      C8_UNROLL
      for (int j = 0; j < NL; ++j) { r.dxi[j] = r.b[j]; r.m.xi[j].v += r.b[j]; }  // the full Newton step
      r.ls_alpha = 1.;
      r.ls_applied = 1.;
      r.ls_best_alpha = 1.;
      r.ls_best_phi = 1.7976931348623157e308;
      r.ls_n = 1;
      r.ls_done = false;
    });
    while (ex.any(searching)) {
      ex.each([&](int k) {  // eval(alpha): move to the trial step, evaluate on the forced branch
        auto& r = ex.lane(k);
        if (!searching(k)) return;
        double const diff = r.ls_alpha - r.ls_applied;
        r.ls_applied = r.ls_alpha;
        C8_UNROLL
        for (int j = 0; j < NL; ++j) r.m.xi[j].v += diff * r.dxi[j];
        r.path = r.m.evaluate(r.g, ms.abs_tol, true, r.path);
        double nrm = 0.;
        C8_UNROLL
        for (int j = 0; j < NL; ++j) nrm += r.m.R[j].v * r.m.R[j].v;
        double const C_alpha = sqrt(nrm);
        r.ls_phi = 0.5 * C_alpha * C_alpha;
        if (k < NL) C8_UNROLL for (int j = 0; j < NL; ++j) sh.M[j][k] = r.m.R[j].d;
      });
      ex.sync();
      ex.each([&](int k) {
        auto& r = ex.lane(k);
        if (!searching(k)) return;
        double slope = 0.;  // phi'(alpha) = C . (J dxi)
        C8_UNROLL
        for (int i = 0; i < NL; ++i) {
          double Jd = 0.;
          C8_UNROLL
          for (int c = 0; c < NL; ++c) Jd += sh.M[i][c] * r.dxi[c];
          slope += r.m.R[i].v * Jd;
        }
        double const phi_0 = r.ls_phi0, dphi_0 = -2. * phi_0;
        if (r.ls_phi < r.ls_best_phi) { r.ls_best_phi = r.ls_phi; r.ls_best_alpha = r.ls_alpha; }
        if (r.ls_phi <= phi_0 + r.ls_alpha * (ms.ls_c1 * dphi_0)) { r.ls_done = true; return; }  // sufficient decrease
        // minimiser of the cubic through (0, phi_0, dphi_0) and (alpha, phi, slope), safeguarded (line_search.hpp:56-66,:121-123)
        double const a = r.ls_alpha;
        double const d1 = dphi_0 + slope - 3. * (phi_0 - r.ls_phi) / (0. - a);
        double const radicand = d1 * d1 - dphi_0 * slope;
        double alpha_model = 0.5 * a;
        if (!(radicand < 0.)) {
          double const d2 = sqrt(radicand);
          double const denom = slope - dphi_0 + 2. * d2;
          if (denom != 0.) alpha_model = a - a * (slope + d2 - d1) / denom;
        }
        double const lo = ms.ls_bmin * a, hi = ms.ls_bmax * a;
        r.ls_alpha = fmin(fmax(alpha_model, lo), hi);
        r.ls_n++;
        if (r.ls_n > ms.ls_max_evals) { r.ls_alpha = r.ls_best_alpha; r.ls_done = true; }  // the lowest-merit step
      });
      ex.sync();
    }
    ex.each([&](int k) {  // move the local state to the accepted step
      auto& r = ex.lane(k);
      if (!active(k) || r.failed) return;
      double const diff = r.ls_alpha - r.ls_applied;
      C8_UNROLL
      for (int j = 0; j < NL; ++j) r.m.xi[j].v += diff * r.dxi[j];
      r.iter++;
    });
  }
  ex.each([&](int k) {
    auto& r = ex.lane(k);
    if ((r.iter > ms.max_iters) && !r.converged) r.failed = true;
  });
}

// =====================================================================================
// K1: eval_forward_jacobian (evaluations.cpp:12-154) for one element.
// =====================================================================================
template <class M, class = void> struct has_closed_form : std::false_type {};
template <class M> struct has_closed_form<M, std::enable_if_t<M::HAS_CLOSED_FORM>> : std::true_type {};

// CLOSED: the instantiation that runs a model's closed form (Model::HAS_CLOSED_FORM) in place of the local Newton iteration
// and the AD passes of the first ip set -- a kernel of its own, so that neither form carries the other's registers
template <class E, template <class> class ModelT, bool CLOSED = false, class EX>
C8_HD void forward_jacobian_element(EX& ex, GroupShared<E, ModelT<Dual>::NLOC>& sh, MeshTables const& mt,
                                    ModelSettings const& ms, FieldArgs const& fa, SystemArgs const& sa, int e) {
  using Model = ModelT<Dual>;
  constexpr int NL = Model::NLOC;
  constexpr bool PREV = Model::FINITE_DEF;
  using Lane = ForwardLane<E, ModelT>;

  load_element<E>(ex, sh, mt, fa, e, PREV);
  ex.each([&](int k) {
    Lane& r = ex.lane(k);
    C8_UNROLL
    for (int a = 0; a < E::NDOF; ++a) r.Jcol[a] = 0.;
    r.Rk = 0.;
    r.failed = false;
    load_params(r.m, mt, e);
    if (k == 0) sh.h = elem_size<E>(sh);
  });
  ex.sync();

  for (int ip_set = 0; ip_set < E::NSETS; ++ip_set) {
    if (ip_set == 0 || !E::SAME_POINTS) shape_tables<E>(ex, sh, ip_set);
    int const npts = ip_set == 0 ? E::NP0 : E::NP1;
    for (int pt = 0; pt < npts; ++pt) {
      if constexpr (CLOSED) if (ip_set == 0) {
        // the model's closed form (Model::closed_form: converged state, fluxes and what the consistent tangent needs) in
        // place of the local Newton iteration and the two AD passes; every lane of the group evaluates it (the values are
        // replicated over the group as in the iterated form) and takes the column of its own element unknown from it
        // (Model::closed_form_flux_column).  The pressure term of the second ip set stays with that set's loop below.
        if constexpr (has_closed_form<Model>::value) {
          size_t const q = ((size_t)e * E::NP0 + pt) * NL;
          ex.each([&](int k) {
            Lane& r = ex.lane(k);
            interpolate_values<E, Dual, PREV>(sh, pt, r.g);
            double const qv[16] = {r.g.grad_u.xx.v, r.g.grad_u.xy.v, r.g.grad_u.xz.v, r.g.grad_u.yx.v, r.g.grad_u.yy.v,
                                   r.g.grad_u.yz.v, r.g.grad_u.zx.v, r.g.grad_u.zy.v, r.g.grad_u.zz.v, r.g.p.v,
                                   r.g.grad_p[0].v, r.g.grad_p[1].v, r.g.grad_p[2].v, r.g.u[0].v, r.g.u[1].v, r.g.u[2].v};
            double xo[NL];
            C8_UNROLL
            for (int j = 0; j < NL; ++j) xo[j] = fa.xi_prev[q + j];
            int const es = mt.elem_set ? mt.elem_set[e] : 0;
            typename Model::ClosedForm cf;
            Model::closed_form(mt.params + (size_t)es * Model::NPARAMS, qv, xo, ms.abs_tol, sh.h, ms.stab_mult, cf, false);
            if (k == 0) {
              C8_UNROLL
              for (int j = 0; j < NL; ++j) fa.xi[q + j] = cf.xi[j];
            }
            int ik, nk, eqk;
            slot_to_dof<E>(k, ik, nk, eqk);
            double const w = sh.wdv[pt];
            double const ek[3] = {(ik == 0 && eqk == 0) ? 1. : 0., (ik == 0 && eqk == 1) ? 1. : 0., (ik == 0 && eqk == 2) ? 1. : 0.};
            double const isp = ik == 1 ? 1. : 0.;
            double const b0 = sh.dN[pt][nk][0], b1 = sh.dN[pt][nk][1], b2 = sh.dN[pt][nk][2], bN = sh.N[pt][nk];
            double const gk[3] = {w * b0, w * b1, w * b2};
            double db[13];
            Model::closed_form_flux_column(cf.t, ek, isp, gk, w * bN, db);
            constexpr int D = E::DIM;
            C8_UNROLL
            for (int n = 0; n < E::NN; ++n) {
              double const a0 = sh.dN[pt][n][0], a1 = sh.dN[pt][n][1], a2 = sh.dN[pt][n][2], aN = sh.N[pt][n];
              r.Jcol[D * n + 0] = fma(a2, db[2], fma(a1, db[1], fma(a0, db[0], r.Jcol[D * n + 0])));
              r.Jcol[D * n + 1] = fma(a2, db[5], fma(a1, db[4], fma(a0, db[3], r.Jcol[D * n + 1])));
              r.Jcol[D * n + 2] = fma(a2, db[8], fma(a1, db[7], fma(a0, db[6], r.Jcol[D * n + 2])));
              r.Jcol[D * E::NN + n] = fma(aN, db[9], fma(a2, db[12], fma(a1, db[11], fma(a0, db[10], r.Jcol[D * E::NN + n]))));
            }
            double const* F = cf.F;
            // (selects, not an index: a register array indexed at run time would live in scratch memory)
            double const r0 = F[0] * b0 + F[1] * b1 + F[2] * b2;
            double const r1 = F[3] * b0 + F[4] * b1 + F[5] * b2;
            double const r2 = F[6] * b0 + F[7] * b1 + F[8] * b2;
            double const rp = F[9] * bN + F[10] * b0 + F[11] * b1 + F[12] * b2;
            r.Rk += w * (ik == 1 ? rp : (eqk == 0 ? r0 : (eqk == 1 ? r1 : r2)));
          });
        }
      }
      if (ip_set == 0) {
        if constexpr (!CLOSED) {
        size_t const q = ((size_t)e * E::NP0 + pt) * NL;
        // --- local->gather, seed_wrt_xi, solve_nonlinear (small_J2.cpp:122-173) ---
        ex.each([&](int k) {
          Lane& r = ex.lane(k);
          interpolate_values<E, Dual, PREV>(sh, pt, r.g);
          C8_UNROLL
          for (int j = 0; j < NL; ++j) {
            r.m.xi_prev[j] = Dual(fa.xi_prev[q + j]);
            r.m.xi[j] = Dual(fa.xi[q + j], (j == k) ? 1. : 0.);
            r.m.R[j] = Dual(0.);
          }
          r.m.initial_guess(r.g);
          r.iter = 1;
          r.R_norm_0 = 1.;
          r.converged = !Model::HAS_LOCAL;
        });
        if constexpr (uses_line_search<Model>::value) {
          local_newton_line_search<NL>(ex, sh, ms);
        } else if (Model::HAS_LOCAL) {
          while (ex.any([&](int k) { Lane& r = ex.lane(k); return (r.iter <= ms.max_iters) && !r.converged; })) {
            ex.each([&](int k) {
              Lane& r = ex.lane(k);
              if (!((r.iter <= ms.max_iters) && !r.converged)) return;
              r.m.evaluate(r.g, ms.abs_tol);
              double nrm = 0.;
              C8_UNROLL
              for (int j = 0; j < NL; ++j) nrm += r.m.R[j].v * r.m.R[j].v;
              double const R_norm = sqrt(nrm);
              if (r.iter == 1) r.R_norm_0 = R_norm;
              double const R_norm_rel = R_norm / r.R_norm_0;  // NaN on elastic points: the abs test decides
              if ((R_norm_rel < ms.rel_tol) || (R_norm < ms.abs_tol)) r.converged = true;
              if (k < NL) C8_UNROLL for (int j = 0; j < NL; ++j) sh.M[j][k] = r.m.R[j].d;
              C8_UNROLL
              for (int j = 0; j < NL; ++j) r.b[j] = -r.m.R[j].v;
            });
            ex.sync();
            if (!ex.any([&](int k) { Lane& r = ex.lane(k); return (r.iter <= ms.max_iters) && !r.converged; })) break;
            bool const ok = gj_solve<NL>(ex, sh, [&](int k) { return ex.lane(k).b; });
            ex.each([&](int k) {
              Lane& r = ex.lane(k);
              if (!ok) { r.failed = true; r.iter = ms.max_iters + 1; return; }
              C8_UNROLL
              for (int j = 0; j < NL; ++j) r.m.xi[j].v += r.b[j];
              r.iter++;
            });
          }
          ex.each([&](int k) {
            Lane& r = ex.lane(k);
            if ((r.iter > ms.max_iters) && !r.converged) r.failed = true;
          });
        }
        // --- local->scatter; dC/dxi is in sh.M from the last evaluate; unseed xi; seed x;
        //     evaluate -> dC/dx; dxi/dx = -(dC/dxi)^-1 dC/dx  (evaluations.cpp:101-115) ---
        ex.each([&](int k) {
          Lane& r = ex.lane(k);
          if (k == 0) {  // values are replicated over the group: one lane stores them
            C8_UNROLL
            for (int j = 0; j < NL; ++j) fa.xi[q + j] = r.m.xi[j].v;
          }
          C8_UNROLL
          for (int j = 0; j < NL; ++j) r.m.xi[j].d = 0.;
          seed_x<E>(sh, pt, k, r.g);
          if (Model::HAS_LOCAL) {
            r.m.evaluate(r.g, ms.abs_tol);
            C8_UNROLL
            for (int j = 0; j < NL; ++j) r.b[j] = -r.m.R[j].d;
          }
        });
        if (Model::HAS_LOCAL) {
          ex.sync();
          bool const ok = gj_solve<NL>(ex, sh, [&](int k) { return ex.lane(k).b; });
          ex.each([&](int k) {
            Lane& r = ex.lane(k);
            if (!ok) r.failed = true;
            C8_UNROLL
            for (int j = 0; j < NL; ++j) r.m.xi[j].d = r.b[j];  // local->seed_wrt_x(dxi_dx)
          });
        }
        // --- global->evaluate + accumulate (evaluations.cpp:126-131) ---
        ex.each([&](int k) {
          Lane& r = ex.lane(k);
          MechFlux<Dual> f;
          global_flux<E>(r.m, r.g, sh.h, ms, f);
          accumulate_coupled<E>(sh, pt, k, f, r.Jcol, r.Rk);
        });
        }  // !CLOSED
      } else if constexpr (E::NRES == 2) {
        ex.each([&](int k) {
          Lane& r = ex.lane(k);
          interpolate_values<E, Dual, false>(sh, pt, r.g);
          seed_x<E>(sh, pt, k, r.g);
          Dual const Vp = Mechanics::flux_pressure(r.m, r.g);
          double const wdv = sh.wdv[pt];
          int ik, nk, eqk;
          slot_to_dof<E>(k, ik, nk, eqk);
          C8_UNROLL
          for (int n = 0; n < E::NN; ++n) r.Jcol[E::DIM * E::NN + n] += Vp.d * (sh.N[pt][n] * wdv);
          if (ik == 1) r.Rk += Vp.v * (sh.N[pt][nk] * wdv);
        });
      }
    }
  }
  ex.sync();
  scatter_lhs<E>(ex, sh, mt, sa, e, false, [&](int k) { return ex.lane(k).Jcol; });
  scatter_rhs<E>(ex, sh, sa, [&](int k) { return ex.lane(k).Rk; }, e);
  ex.each([&](int k) {
    if (k == 0 && ex.lane(k).failed) ex.flag(sa.status);
  });
}

}  // namespace c8
