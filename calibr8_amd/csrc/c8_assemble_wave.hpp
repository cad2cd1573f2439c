// c8_assemble_wave.hpp -- K1 for hex8 with ONE WAVEFRONT PER ELEMENT.
//
// Same mathematics as forward_jacobian_element (c8_assemble.hpp, evaluations.cpp:12-154) but the
// 64 lanes are re-assigned phase by phase so that every lane carries a useful tangent:
//
//   phase N  local Newton          lane = point*8 + xi-direction        all 8 coupled points at once
//   phase D  point derivatives     lane = (point%4)*16 + q-direction    4 points per pass, 2 passes
//            q = the 16 interpolated point quantities (grad u 9, p 1, grad p 3, u 3) that the
//            residuals depend on; lane c gets dC/dq_c, solves dxi/dq_c, and column c of the
//            13 x 16 matrix D = d(point fluxes)/dq with the local state condensed
//   phase P  contraction           lane = column b (32) x row half (2)
//            J_e += w dv  W^T D B   with B = dq/dx_e and W = d(R_e)/d(flux): the shape-function
//            tables, both sparse (3-4 non-zeros per row/column)
//
// Against the slot-per-lane kernel this runs the expensive AD passes 2 (instead of 8) times
// per element and the Newton iterations once (instead of 8 times), and needs no 32-entry
// Jacobian column in registers while the constitutive code is live.  The chain rule through q
// is exact: the result is the same dR/dx to rounding.
//
// Requires an element with 32 DOFs, 8 coupled points and identical point sets for both ip
// sets (hex8): the two ip sets are fused by adding the pressure-mass flux to V_p.
#pragma once
// Diagnostic build only (-DC8_STAMPS, tools/stamp_phases.py): s_memtime stamps at phase boundaries.
#ifdef C8_STAMPS
#define C8_STAMP(i) ex.stamp(sa, e, i)
#else
#define C8_STAMP(i)
#endif

#include "c8_assemble.hpp"

namespace c8 {

constexpr int WQ = 16;   // point quantities: grad_u (0..8 row-major), p (9), grad_p (10..12), u (13..15)
constexpr int WF = 13;   // point fluxes: Gu (0..8 row-major), Vp (9), Gp (10..12)

// ---- cached shape tables (static geometry) --------------------------------------------------------------------
// Per element: dN[pt][n][3] (lane = pt*8 + n reads its three values contiguously), w dv [pt], element size h.
constexpr int SHAPE_STRIDE = 208;  // 192 + 8 + 1, padded to a multiple of 64 bytes
constexpr int SHAPE_WDV = 192, SHAPE_H = 200;
// dN/dx of (point pt, node m), component d, within an element's cached table.  The 8 x 8 entries are stored SKEWED: row
// r = (pt + m) mod 8, column m -- per row first the components 0 and 1 of the eight columns (16 doubles), then their
// components 2 (8 doubles).  The eight lanes of an element then read ONE row of 192 contiguous bytes per step whether a lane
// owns a point and walks the nodes (lane pt at step i: node (i - pt) mod 8) or owns a node and walks the points (lane m at
// step i: point (i - m) mod 8): the row-per-node kernels (c8_assemble_node.hpp) are bound by the number of 64-byte segments
// their loads touch, and a row-major table costs a lane-per-point reader eight times the segments.
C8_HD constexpr int shape_dn_offset(int pt, int m, int d) { return ((pt + m) & 7) * 24 + (d < 2 ? m * 2 + d : 16 + m); }
template <class E> struct ShapeShared {
  double X[E::NN][3];
  double N[E::NP0][E::NN];
  double dN[E::NP0][E::NN][3];
  double wdv[E::NP0];
  double h;
};
struct ShapeLane {};
// one wavefront per element, the arithmetic of the kernels' own shape phase (shape_entry, elem_size)
template <class E, class EX> C8_HD void store_shape_tables(EX& ex, ShapeShared<E>& sh, MeshTables const& mt, double* tab, int e) {
  static_assert(E::NN == 8 && E::NP0 == 8, "hex8-like element");
  ex.each([&](int lane) {
    if (lane < 3 * E::NN) {
      int const n = lane / 3, d = lane - 3 * n;
      sh.X[n][d] = mt.coords[(size_t)mt.conn[e * E::NN + n] * 3 + d];
    }
  });
  ex.sync();
  ex.each([&](int lane) {
    shape_entry<E>(sh, 0, lane >> 3, lane & 7, (lane & 7) + 1);
    if (lane == 0) sh.h = elem_size<E>(sh);
  });
  ex.sync();
  ex.each([&](int lane) {
    double* const t = tab + (size_t)e * SHAPE_STRIDE;
    int const pt = lane >> 3, n = lane & 7;
    t[shape_dn_offset(pt, n, 0)] = sh.dN[pt][n][0];
    t[shape_dn_offset(pt, n, 1)] = sh.dN[pt][n][1];
    t[shape_dn_offset(pt, n, 2)] = sh.dN[pt][n][2];
    if (lane < E::NP0) t[SHAPE_WDV + lane] = sh.wdv[lane];
    if (lane == E::NP0) t[SHAPE_H] = sh.h;
    if (lane > E::NP0 && lane < SHAPE_STRIDE - SHAPE_H + E::NP0) t[SHAPE_H + lane - E::NP0] = 0.;  // padding
  });
  ex.sync();
}
// the lane's share of the cached tables: issued with the first loads of an element ...
template <class E, class R> C8_HD void load_cached_shape(R& r, MeshTables const& mt, int e, int lane) {
  double const* const t = mt.shape + (size_t)e * SHAPE_STRIDE;
  // lane = (row, column) of the skewed table, i.e. the entry of point (row - column) mod 8 and node `column`: eight lanes
  // read one row (commit_cached_shape puts the entry where it belongs)
  int const row = lane >> 3, col = lane & 7;
  r.dn[0] = t[row * 24 + col * 2 + 0];
  r.dn[1] = t[row * 24 + col * 2 + 1];
  r.dn[2] = t[row * 24 + 16 + col];
  r.sx = (lane <= E::NP0) ? t[SHAPE_WDV + lane] : 0.;  // lanes 0..7: w dv of point `lane`; lane 8: h
}
// ... and committed to LDS beside the nodal data (N is a constant of the reference element)
template <class E, class R, class SH> C8_HD void commit_cached_shape(R const& r, SH& sh, int lane) {
  int const n = lane & 7, pt = ((lane >> 3) - n) & 7;  // the entry this lane fetched (load_cached_shape)
  double xi[3], w;
  E::point(0, pt, xi, w);
  sh.N[pt][n] = E::N(n, xi);
  sh.dN[pt][n][0] = r.dn[0];
  sh.dN[pt][n][1] = r.dn[1];
  sh.dN[pt][n][2] = r.dn[2];
  if (lane < E::NP0) sh.wdv[lane] = r.sx;
  if (lane == E::NP0) sh.h = r.sx;
}

// ADJ / PREV: the arrays only the adjoint assembly / only finite-deformation models use are left out of the other
// instantiations (one entry instead): 17.7 KB instead of 19.2 KB for the forward assembly of a small-strain model.
// NOSOLVE: the closed-form forward kernel has no local elimination and leaves out the matrices M (4.6 KB).
template <class E, int NL, bool ADJ = true, bool PREV = true, bool NOSOLVE = false> struct WaveShared {
  static constexpr int NLP = 8;  // lanes per point in phase N
  double X[E::NN][3];
  double u[E::NN][3], p[E::NN];
  double u_prev[PREV ? E::NN : 1][3];
  double N[E::NP0][E::NN];
  double dN[E::NP0][E::NN][3];
  double wdv[E::NP0];
  double M[NOSOLVE ? 1 : E::NP0][NOSOLVE ? 1 : NLP][NLP + 1];   // dC/dxi per point
  alignas(16) double q[E::NP0][WQ]; // interpolated values
  // closed-form forward kernel: (dN/dx_0, dN/dx_1, dN/dx_2, N) of every (point, node) side by side, so that phase P reads
  // a node's four shape entries with two 16-byte LDS loads
  alignas(16) double G4[NOSOLVE ? E::NP0 : 1][NOSOLVE ? E::NN : 1][4];
  double qprev[PREV ? E::NP0 : 1][9];  // grad_u at the previous step (finite deformation)
  double xi[E::NP0][NLP];           // converged local state
  double xip[E::NP0][NLP];          // previous local state
  double D[NOSOLVE ? 1 : 4][WF][WQ + 1];  // dflux/dq of the 4 points of a pass
  double F[E::NP0][WF];             // flux values
  double gh[ADJ ? E::NP0 : 1][NLP]; // adjoint: local history g at each point
  double rq[ADJ ? 4 : 1][WQ + 1];   // adjoint: -dJ/dq + (dxi/dq)^T g of the 4 points of a pass
  double h;
  int32_t node[E::NN];
  int32_t nptr[E::NN], deg[E::NN];
  int32_t failed;
};

template <template <class> class ModelT> struct WaveLane {
  using Model = ModelT<Dual>;
  Model m;
  typename Model::Trial trial;
  PointState<Dual> g;
  double b[Model::NLOC];
  double J[16];   // phase P: rows of this lane's half, column b
  double J1[16];  // closed-form forward kernel: the rows of flux group 1 (phase P there), otherwise unused
  double R, Rx;
  // issued with the first loads of the element, used much later: the previous / current local state of this lane's
  // (point, direction) and the eight CSR positions of this lane's column node (their round trips would otherwise
  // be exposed after the shape tables and in front of the scatter)
  double xi_pre, xip_pre;
  unsigned long long pos8;
  double dn[3], sx;  // cached shape tables: this lane's dN/dx entry; w dv (lanes 0..7) or h (lane 8)
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

template <class SH> C8_HD void load_point(SH const& sh, int pt, PointState<Dual>& g, bool prev) {
  g.grad_u.xx = Dual(sh.q[pt][0]); g.grad_u.xy = Dual(sh.q[pt][1]); g.grad_u.xz = Dual(sh.q[pt][2]);
  g.grad_u.yx = Dual(sh.q[pt][3]); g.grad_u.yy = Dual(sh.q[pt][4]); g.grad_u.yz = Dual(sh.q[pt][5]);
  g.grad_u.zx = Dual(sh.q[pt][6]); g.grad_u.zy = Dual(sh.q[pt][7]); g.grad_u.zz = Dual(sh.q[pt][8]);
  g.p = Dual(sh.q[pt][9]);
  g.grad_p[0] = Dual(sh.q[pt][10]); g.grad_p[1] = Dual(sh.q[pt][11]); g.grad_p[2] = Dual(sh.q[pt][12]);
  g.u[0] = Dual(sh.q[pt][13]); g.u[1] = Dual(sh.q[pt][14]); g.u[2] = Dual(sh.q[pt][15]);
  if (prev) {
    g.grad_u_prev.xx = Dual(sh.qprev[pt][0]); g.grad_u_prev.xy = Dual(sh.qprev[pt][1]); g.grad_u_prev.xz = Dual(sh.qprev[pt][2]);
    g.grad_u_prev.yx = Dual(sh.qprev[pt][3]); g.grad_u_prev.yy = Dual(sh.qprev[pt][4]); g.grad_u_prev.yz = Dual(sh.qprev[pt][5]);
    g.grad_u_prev.zx = Dual(sh.qprev[pt][6]); g.grad_u_prev.zy = Dual(sh.qprev[pt][7]); g.grad_u_prev.zz = Dual(sh.qprev[pt][8]);
  } else {
    g.grad_u_prev = scale(0., eye3<Dual>());
  }
}

// seed point quantity c
C8_HD void seed_q(PointState<Dual>& g, int c) {
  g.grad_u.xx.d = (c == 0) ? 1. : 0.; g.grad_u.xy.d = (c == 1) ? 1. : 0.; g.grad_u.xz.d = (c == 2) ? 1. : 0.;
  g.grad_u.yx.d = (c == 3) ? 1. : 0.; g.grad_u.yy.d = (c == 4) ? 1. : 0.; g.grad_u.yz.d = (c == 5) ? 1. : 0.;
  g.grad_u.zx.d = (c == 6) ? 1. : 0.; g.grad_u.zy.d = (c == 7) ? 1. : 0.; g.grad_u.zz.d = (c == 8) ? 1. : 0.;
  g.p.d = (c == 9) ? 1. : 0.;
  g.grad_p[0].d = (c == 10) ? 1. : 0.; g.grad_p[1].d = (c == 11) ? 1. : 0.; g.grad_p[2].d = (c == 12) ? 1. : 0.;
  g.u[0].d = (c == 13) ? 1. : 0.; g.u[1].d = (c == 14) ? 1. : 0.; g.u[2].d = (c == 15) ? 1. : 0.;
}

// interpolated point quantities (global_residual.cpp:289-332).  The 16 point quantities are the 4 x 4 products
// (nodal value a) x (shape entry b) summed over the nodes:
//   a in {u_0, u_1, u_2, p},  b in {dN/dx_0, dN/dx_1, dN/dx_2, N};  q index: grad_u[a][b] = 3a+b, p = 9 (a=b=3),
//   grad_p[b] = 10+b (a=3), u[a] = 13+a (b=3).
// One code path for every lane (the index only selects base address and stride of the two LDS operands); same operand
// order and the same sequential sum over the nodes as the reference's loop.
C8_HD int q_index(int a, int b) { return (a < 3) ? ((b < 3) ? 3 * a + b : 13 + a) : ((b < 3) ? 10 + b : 9); }
template <class E, class SH>
C8_HD double interp_ab(SH const& sh, int pt, int a, int b, double const* u3 /* [NN][3] */, double const* p1 /* [NN] */) {
  double const* const Ap = (a < 3) ? u3 + a : p1;
  int const sa = (a < 3) ? 3 : 1;
  double const* const Bp = (b < 3) ? &sh.dN[pt][0][b] : &sh.N[pt][0];
  int const sb = (b < 3) ? 3 : 1;
  double s = 0.;
  C8_UNROLL
  for (int n = 0; n < E::NN; ++n) s += Ap[n * sa] * Bp[n * sb];
  return s;
}

#ifdef C8_TUNE_ALWAYS_SWAP  // tuning build (same results): the row-exchange selects of the local solves run unconditionally
#define C8_ALWAYS_SWAP true
#else
#define C8_ALWAYS_SWAP false
#endif
// Gauss-Jordan with row pivoting inside lane groups of G lanes (see gj_solve in c8_assemble.hpp for the slot kernels).
// Lane cg (< NL) of a group owns COLUMN cg of the group's matrix in its registers (col(lane, r), r = 0..NL-1) and every
// lane of the group carries its own right-hand side b (registers).  Step s: the owner of column s publishes it through
// the group's LDS matrix Mg (column s of Mg, one write and one read per entry: the only LDS traffic of a step), every
// lane finds the same pivot row, swaps, eliminates its right-hand side and -- lanes cg > s -- its own column.  The
// arithmetic is the one of an elimination on a shared matrix, entry by entry.
template <int NL, int G, class EX, class GetM, class Col, class GetB, class Active>
C8_HD bool gj_solve_cols(EX& ex, GetM getm, Col col, GetB getb, Active active) {
  bool ok = true;
  C8_UNROLL
  for (int s = 0; s < NL; ++s) {
    ex.each([&](int lane) {
      if (!active(lane)) return;
      if (lane % G == s) {
        auto* M = getm(lane);  // double (*)[G + 1]
        static_for<NL>([&](auto rc) { constexpr int r = decltype(rc)::value; M[r][s] = col(lane, r); });
      }
    });
    ex.sync();
    ex.each([&](int lane) {
      if (!active(lane)) return;
      int const cg = lane % G;
      auto* M = getm(lane);
      double* b = getb(lane);
      double pc[NL];  // the pivot column
      C8_UNROLL
      for (int r = 0; r < NL; ++r) pc[r] = M[r][s];
      int rstar = s;
      double big = fabs(pc[s]);
      C8_UNROLL
      for (int r = s + 1; r < NL; ++r) {
        double const a = fabs(pc[r]);
        if (a > big) { big = a; rstar = r; }
      }
      if (!(big > 0.)) ok = false;
      bool const mine = cg > s && cg < NL;  // this lane still has a column to eliminate
      double const cs = pc[s], bs = b[s], ms = col(lane, s);
      double cpiv = cs, bpiv = bs, mpiv = ms;
      if (ex.uniform_any(rstar != s) || C8_ALWAYS_SWAP) {  // row exchanges are rare (never needed by the J2 models): skipped wave-wide when no group pivots
        static_for<NL>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          bool const hit = (r > s) && (r == rstar);
          cpiv = hit ? pc[r] : cpiv;
          bpiv = hit ? b[r] : bpiv;
          mpiv = hit ? col(lane, r) : mpiv;
          pc[r] = hit ? cs : pc[r];
          b[r] = hit ? bs : b[r];
          col(lane, r) = hit ? ms : col(lane, r);
        });
      }
      double const inv = c8_rcp(cpiv);
      double const bsn = bpiv * inv;
      b[s] = bsn;
      C8_UNROLL
      for (int r = 0; r < NL; ++r) if (r != s) b[r] -= pc[r] * bsn;
      if (mine) {
        double const msn = mpiv * inv;
        static_for<NL>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          if (r == s) col(lane, r) = msn;
          else col(lane, r) -= pc[r] * msn;
        });
      }
    });
  }
  return ok;
}

// The same elimination with the pivot column handed over by a cross-lane broadcast (ex.bcast8: DPP moves on the GPU) --
// no LDS traffic and no barrier in a step, but six VALU operations per value.  Which hand-over is faster depends on the
// registers the surrounding code leaves free: a per-model, per-kernel trait (GJ_XLANE_JAC, GJ_XLANE_K4), as measured.
template <int NL, class EX, class Col, class GetB, class Active>
C8_HD bool gj_solve_xlane(EX& ex, Col col, GetB getb, Active active) {
  bool ok = true;
  static_for<NL>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    ex.each([&](int lane) {
      if (!active(lane)) return;
      int const cg = lane & 7;
      double* b = getb(lane);
      double pc[NL];  // the pivot column
      static_for<NL>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        pc[r] = ex.template bcast8<s>(lane, [&](int l) -> double { return col(l, r); });
      });
      int rstar = s;
      double big = fabs(pc[s]);
      C8_UNROLL
      for (int r = s + 1; r < NL; ++r) {
        double const a = fabs(pc[r]);
        if (a > big) { big = a; rstar = r; }
      }
      if (!(big > 0.)) ok = false;
      bool const mine = cg > s && cg < NL;  // this lane still has a column to eliminate
      double const cs = pc[s], bs = b[s], ms = col(lane, s);
      double cpiv = cs, bpiv = bs, mpiv = ms;
      if (ex.uniform_any(rstar != s) || C8_ALWAYS_SWAP) {  // see gj_solve_cols
        static_for<NL>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          bool const hit = (r > s) && (r == rstar);
          cpiv = hit ? pc[r] : cpiv;
          bpiv = hit ? b[r] : bpiv;
          mpiv = hit ? col(lane, r) : mpiv;
          pc[r] = hit ? cs : pc[r];
          b[r] = hit ? bs : b[r];
          col(lane, r) = (hit && mine) ? ms : col(lane, r);  // the owner of the pivot column leaves it as broadcast
        });
      }
      double const inv = c8_rcp(cpiv);
      double const bsn = bpiv * inv;
      b[s] = bsn;
      C8_UNROLL
      for (int r = 0; r < NL; ++r) if (r != s) b[r] -= pc[r] * bsn;
      if (mine) {
        double const msn = mpiv * inv;
        static_for<NL>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          if (r == s) col(lane, r) = msn;
          else col(lane, r) -= pc[r] * msn;
        });
      }
    });
  });
  return ok;
}

// hand-over of the Newton solve of K1: GJ_XLANE_NEWTON where a model names it, else the model's GJ_XLANE_JAC
template <class M, class = void> struct xlane_newton : std::integral_constant<bool, M::GJ_XLANE_JAC> {};
template <class M> struct xlane_newton<M, std::void_t<decltype(M::GJ_XLANE_NEWTON)>> : std::integral_constant<bool, M::GJ_XLANE_NEWTON> {};
#ifdef C8_TUNE_XL  // tuning build (same results): hand-over of the Newton solve (bit 0) and of the inverse (bit 1), every model
#define C8_XL_NEWTON(M) ((C8_TUNE_XL & 1) != 0)
#define C8_XL_INVERSE(M) ((C8_TUNE_XL & 2) != 0)
#else
#define C8_XL_NEWTON(M) xlane_newton<M>::value
#define C8_XL_INVERSE(M) M::GJ_XLANE_JAC
#endif
// the local solve of the wave kernels: column d of the matrix is lane d's R[.].d, the right-hand side is lane.b
template <int NL, bool XLANE, class EX, class SH, class Active> C8_HD bool local_solve(EX& ex, SH& sh, Active active) {
  auto col = [&](int lane, int j) -> double& { return ex.lane(lane).m.R[j].d; };
  auto rhs = [&](int lane) { return ex.lane(lane).b; };
  if constexpr (XLANE) return gj_solve_xlane<NL>(ex, col, rhs, active);
  else return gj_solve_cols<NL, 8>(ex, [&](int lane) { return sh.M[lane >> 3]; }, col, rhs, active);
}

// The same elimination with the whole matrix in the group's LDS matrix Mg (lane cg updates column cg there): three
// dependent LDS round trips per step instead of one, but no column held in registers -- for models whose local Newton
// iteration has no registers to spare (Model::NEWTON_MATRIX_IN_LDS).
template <int NL, int G, class EX, class GetM, class GetB, class Active>
C8_HD bool gj_solve_grouped(EX& ex, GetM getm, GetB getb, Active active) {
  bool ok = true;
  C8_UNROLL
  for (int s = 0; s < NL; ++s) {
    ex.each([&](int lane) {
      if (!active(lane)) return;
      int const cg = lane % G;
      auto* M = getm(lane);  // double (*)[9]
      double* b = getb(lane);
      double col[NL];
      C8_UNROLL
      for (int r = 0; r < NL; ++r) col[r] = M[r][s];
      int rstar = s;
      double big = fabs(col[s]);
      C8_UNROLL
      for (int r = s + 1; r < NL; ++r) {
        double const a = fabs(col[r]);
        if (a > big) { big = a; rstar = r; }
      }
      if (!(big > 0.)) ok = false;
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
      double const inv = c8_rcp(cpiv);
      double const bsn = bpiv * inv;
      b[s] = bsn;
      C8_UNROLL
      for (int r = 0; r < NL; ++r) if (r != s) b[r] -= col[r] * bsn;
      if (cg > s && cg < NL) {
        double const ms = M[s][cg], mr = M[rstar][cg];
        M[rstar][cg] = ms;
        double const msn = mr * inv;
        M[s][cg] = msn;
        C8_UNROLL
        for (int r = 0; r < NL; ++r) if (r != s) M[r][cg] -= col[r] * msn;
      }
    });
    ex.sync();
  }
  return ok;
}

// Model::trial for the local Newton iteration: its arguments (the point quantities, the previous local state, the parameters)
// carry no tangent there -- the seeds sit in xi --, so it is evaluated in plain doubles (the model's double instantiation, the
// arithmetic of the dual numbers' value parts) instead of dual numbers whose tangent halves are all zero.
template <template <class> class ModelT> C8_HD void model_values(ModelT<Dual> const& m, PointState<Dual> const& g, ModelT<double>& md, PointState<double>& gd) {
  C8_UNROLL
  for (int k = 0; k < ModelT<Dual>::NPARAMS; ++k) md.params[k] = m.params[k].v;
  C8_UNROLL
  for (int k = 0; k < ModelT<Dual>::NLOC; ++k) { md.xi[k] = m.xi[k].v; md.xi_prev[k] = m.xi_prev[k].v; }
  C8_UNROLL
  for (int k = 0; k < 3; ++k) { gd.u[k] = g.u[k].v; gd.grad_p[k] = g.grad_p[k].v; }
  gd.p = g.p.v;
  auto vals = [](Tens3<Dual> const& a) {
    Tens3<double> b;
    b.xx = a.xx.v; b.xy = a.xy.v; b.xz = a.xz.v; b.yx = a.yx.v; b.yy = a.yy.v; b.yz = a.yz.v; b.zx = a.zx.v; b.zy = a.zy.v; b.zz = a.zz.v;
    return b;
  };
  gd.grad_u = vals(g.grad_u);
  gd.grad_u_prev = vals(g.grad_u_prev);
}
template <template <class> class ModelT> C8_HD typename ModelT<Dual>::Trial trial_of(ModelT<double> const& md, PointState<double> const& gd) {
  using TD = typename ModelT<Dual>::Trial;
  if constexpr (std::is_same<TD, NoTrial>::value) {
    return TD{};
  } else {
    using T1 = typename ModelT<double>::Trial;
    static_assert(sizeof(TD) == sizeof(T1) / sizeof(double) * sizeof(Dual), "Trial: an aggregate of T");
    T1 const t = md.trial(gd);
    TD out;
    double const* const src = reinterpret_cast<double const*>(&t);
    Dual* const dst = reinterpret_cast<Dual*>(&out);
    C8_UNROLL
    for (int k = 0; k < (int)(sizeof(T1) / sizeof(double)); ++k) dst[k] = Dual(src[k]);
    return out;
  }
}
template <template <class> class ModelT> C8_HD typename ModelT<Dual>::Trial trial_values(ModelT<Dual> const& m, PointState<Dual> const& g) {
  if constexpr (std::is_same<typename ModelT<Dual>::Trial, NoTrial>::value) {
    return {};
  } else {
    ModelT<double> md;
    PointState<double> gd;
    model_values<ModelT>(m, g, md, gd);
    return trial_of<ModelT>(md, gd);
  }
}
// The start of the local Newton iteration: the model's initial guess (values only by definition: solve_nonlinear of every
// model sets the VALUES of xi) and its trial state, both from one pass through the model's double instantiation -- the
// reference's guesses of the finite-deformation models are their trial states, which the dual-number form evaluated twice.
template <template <class> class ModelT> C8_HD typename ModelT<Dual>::Trial guess_and_trial_values(ModelT<Dual>& m, PointState<Dual> const& g) {
  ModelT<double> md;
  PointState<double> gd;
  model_values<ModelT>(m, g, md, gd);
  md.initial_guess(gd);
  C8_UNROLL
  for (int k = 0; k < ModelT<Dual>::NLOC; ++k) m.xi[k].v = md.xi[k];
  return trial_of<ModelT>(md, gd);
}

// ---- local Newton iteration with line search in the 8-lanes-per-point layout (the lane-group form with its references:
// local_newton_line_search, c8_assemble.hpp): lane = point*8 + d holds column d of J = dC/dxi in its tangents and hands it
// to the point's eight lanes through sh.M[pt]; the values are replicated over the eight lanes, so they take the same
// decisions.  The elimination works on the matrix in LDS (gj_solve_grouped): these models have no registers to spare.  On
// exit sh.M[pt] holds dC/dxi of the last evaluation at the converged state, as after the plain iteration.
template <int NL, bool PIN, template <class> class ModelT, class EX, class SH>
C8_HD void local_newton_line_search_wave(EX& ex, SH& sh, ModelSettings const& ms) {
  // the kinematic part of the residual (Model::trial) once per point, in plain doubles (trial_values), not in every evaluation
  constexpr bool CACHED = !std::is_same<typename ModelT<Dual>::Trial, NoTrial>::value;  // r.trial: set with the initial guess
  auto eval = [&](auto& r, bool force) __attribute__((always_inline)) {
    if constexpr (CACHED) return r.m.evaluate(r.g, ms.abs_tol, r.trial, force, r.path);
    else return r.m.evaluate(r.g, ms.abs_tol, force, r.path);
  };
  auto active = [&](int lane) { auto& r = ex.lane(lane); return (r.iter <= ms.max_iters) && !r.converged; };
  auto searching = [&](int lane) { return active(lane) && !ex.lane(lane).ls_done; };
  while (ex.any(active)) {
    ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
      auto& r = ex.lane(lane);
      if (!active(lane)) return;
      int const pt = lane >> 3, d = lane & 7;
      if (r.iter == 1) { r.path = 0; r.path = eval(r, false); }
      else eval(r, true);
      double nrm = 0.;
      C8_UNROLL
      for (int j = 0; j < NL; ++j) nrm += r.m.R[j].v * r.m.R[j].v;
      double const C_norm = sqrt(nrm);
      if (r.iter == 1) r.R_norm_0 = C_norm;
      double const C_norm_rel = C_norm / r.R_norm_0;
      if ((C_norm_rel < ms.rel_tol) || (C_norm < ms.abs_tol)) r.converged = true;
      if (d < NL) C8_UNROLL for (int j = 0; j < NL; ++j) sh.M[pt][j][d] = r.m.R[j].d;
      C8_UNROLL
      for (int j = 0; j < NL; ++j) r.b[j] = -r.m.R[j].v;
      r.ls_phi0 = 0.5 * C_norm * C_norm;
    });
    ex.sync();
    if (!ex.any(active)) break;
    bool const ok = gj_solve_grouped<NL, 8>(ex, [&](int lane) { return sh.M[lane >> 3]; },
                                            [&](int lane) { return ex.lane(lane).b; }, active);
    ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
      auto& r = ex.lane(lane);
      if (!active(lane)) return;
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
      ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);  // eval(alpha): the trial step on the forced branch
        auto& r = ex.lane(lane);
        if (!searching(lane)) return;
        int const pt = lane >> 3, d = lane & 7;
        double const diff = r.ls_alpha - r.ls_applied;
        r.ls_applied = r.ls_alpha;
        C8_UNROLL
        for (int j = 0; j < NL; ++j) r.m.xi[j].v += diff * r.dxi[j];
        r.path = eval(r, true);
        double nrm = 0.;
        C8_UNROLL
        for (int j = 0; j < NL; ++j) nrm += r.m.R[j].v * r.m.R[j].v;
        double const C_alpha = sqrt(nrm);
        r.ls_phi = 0.5 * C_alpha * C_alpha;
        if (d < NL) C8_UNROLL for (int j = 0; j < NL; ++j) sh.M[pt][j][d] = r.m.R[j].d;
      });
      ex.sync();
      ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
        auto& r = ex.lane(lane);
        if (!searching(lane)) return;
        int const pt = lane >> 3;
        double slope = 0.;  // phi'(alpha) = C . (J dxi)
        C8_UNROLL
        for (int i = 0; i < NL; ++i) {
          double Jd = 0.;
          C8_UNROLL
          for (int c = 0; c < NL; ++c) Jd += sh.M[pt][i][c] * r.dxi[c];
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
    ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);  // move the local state to the accepted step
      auto& r = ex.lane(lane);
      if (!active(lane) || r.failed) return;
      double const diff = r.ls_alpha - r.ls_applied;
      C8_UNROLL
      for (int j = 0; j < NL; ++j) r.m.xi[j].v += diff * r.dxi[j];
      r.iter++;
    });
  }
}

// ADJOINT = false: eval_forward_jacobian (evaluations.cpp:12-154)
// ADJOINT = true : eval_adjoint_jacobian (evaluations.cpp:349-526): no local solve (stored xi), the element
//                  matrix is scattered transposed, and the right-hand side is -dJ/dx + f + (dxi/dx)^T g with
//                  g -= dJ/dxi updated in place; every x-derivative goes through the point quantities q.
template <class M, bool ADJ, class = void> struct pin_phases : std::false_type {};
template <class M> struct pin_phases<M, false, std::enable_if_t<M::PIN_PHASES_K1>> : std::true_type {};
template <class M> struct pin_phases<M, true, std::enable_if_t<M::PIN_PHASES_K3>> : std::true_type {};

template <class E, template <class> class ModelT, class QoI, bool ADJOINT, bool CLOSED = false, class EX, class SH>
C8_HD void jacobian_wave(EX& ex, SH& sh, MeshTables const& mt, ModelSettings const& ms,
                         FieldArgs const& fa, AdjointArgs const& aa, SystemArgs const& sa, int e) {
  using Model = ModelT<Dual>;
  constexpr int NL = Model::NLOC;
  constexpr bool PREV = Model::FINITE_DEF;
  static_assert(E::NDOF == 32 && E::NP0 == 8 && E::SAME_POINTS, "wave kernel needs a hex8-like element");
  static_assert(NL <= 8, "at most 8 local unknowns per point");
  // lane-derived values (DOF slots, seeds, table offsets) formed in the phase that uses them instead of at the top of the
  // kernel (C8_PIN on the lane number at the head of every phase): a register-allocation switch per model and kernel, as
  // measured (hyper_J2 K1 23.5 -> 21.5 ms, K3 16.1 -> 13.9; hypo_hill 30.9 -> 29.8, 32.6 -> 29.2; small_hill K3 12.7 -> 11.3;
  // small_J2 and small_hill K1 lose 3-7 % with it)
  constexpr bool PIN = pin_phases<Model, ADJOINT>::value;

  C8_STAMP(0);
  // ---- load: lanes 0..31 = element DOF slots; shape tables: lane = point*8 + node ----------
  ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
    auto& r = ex.lane(lane);
    r.failed = false;
    r.R = 0.;
    if constexpr (!CLOSED) {  // (the closed-form phase P assigns its accumulators at its first point)
      C8_UNROLL
      for (int a = 0; a < 16; ++a) r.J[a] = r.J1[a] = 0.;
    }
    if (lane == 0) sh.failed = 0;
    {
      int const pt = lane >> 3, d = lane & 7;
      r.xi_pre = r.xip_pre = 0.;
      if (d < NL) {
        size_t const q = ((size_t)e * E::NP0 + pt) * NL;
        r.xip_pre = fa.xi_prev[q + d];
        r.xi_pre = fa.xi[q + d];
      }
      int ib, nb, eqb;
      slot_to_dof<E>(lane & 31, ib, nb, eqb);
      r.pos8 = *reinterpret_cast<unsigned long long const*>(mt.pos + ((size_t)e * E::NN + nb) * E::NN);  // pos[e][col node nb][0..7]
      if (mt.shape) load_cached_shape<E>(r, mt, e, lane);
    }
    if (lane < E::NDOF) {
      int i, n, eq;
      slot_to_dof<E>(lane, i, n, eq);
      int const node = mt.conn[e * E::NN + n];
      if (i == 0) {
        if (!mt.shape) sh.X[n][eq] = mt.coords[(size_t)node * 3 + eq];  // only the shape tables read the coordinates
        sh.u[n][eq] = fa.u[(size_t)node * 3 + eq];
        if (PREV) sh.u_prev[n][eq] = fa.u_prev[(size_t)node * 3 + eq];
      } else {
        sh.p[n] = fa.p[node];
        sh.node[n] = node;
        int const a = mt.nodeptr[node];
        sh.nptr[n] = a;
        sh.deg[n] = mt.nodeptr[node + 1] - a;
      }
    }
    if (mt.shape) commit_cached_shape<E>(r, sh, lane);
  });
  ex.sync();
  C8_STAMP(10);
  if (!mt.shape) {
    ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
      shape_entry<E>(sh, 0, lane >> 3, lane & 7, (lane & 7) + 1);
      if (lane == 0) sh.h = elem_size<E>(sh);
    });
    ex.sync();
  }
  C8_STAMP(11);
  // ---- interpolation: lane (pt, d) computes quantities 2d and 2d+1 (and grad_u_prev) --------------
  ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
    int const pt = lane >> 3, d = lane & 7;
    {
      int const k0 = 2 * d, k1 = 2 * d + 1;  // products (a, b) = (k >> 2, k & 3)
      sh.q[pt][q_index(k0 >> 2, k0 & 3)] = interp_ab<E>(sh, pt, k0 >> 2, k0 & 3, &sh.u[0][0], &sh.p[0]);
      sh.q[pt][q_index(k1 >> 2, k1 & 3)] = interp_ab<E>(sh, pt, k1 >> 2, k1 & 3, &sh.u[0][0], &sh.p[0]);
    }
    if (PREV) {  // grad_u at the previous step: entry d = (a, b) with a = d / 3 (and entry 8 on lane d = 0)
      int const a = (d >= 3) + (d >= 6), b = d - 3 * a;
      sh.qprev[pt][d] = interp_ab<E>(sh, pt, a, b, &sh.u_prev[0][0], &sh.p[0]);
      if (d == 0) sh.qprev[pt][8] = interp_ab<E>(sh, pt, 2, 2, &sh.u_prev[0][0], &sh.p[0]);
    }
    if (d < NL) {
      auto& r = ex.lane(lane);
      sh.xip[pt][d] = r.xip_pre;
      sh.xi[pt][d] = r.xi_pre;
    }
    if constexpr (CLOSED) {  // lane = (point, node d)
      sh.G4[pt][d][0] = sh.dN[pt][d][0];
      sh.G4[pt][d][1] = sh.dN[pt][d][1];
      sh.G4[pt][d][2] = sh.dN[pt][d][2];
      sh.G4[pt][d][3] = sh.N[pt][d];
    }
  });
  ex.sync();

  C8_STAMP(1);
  {
  bool need_inverse = false;
  if constexpr (!CLOSED) {  // CLOSED: the model's closed form replaces phases N, the inverse and the AD passes of phase D
  // ---- phase N: local Newton at all 8 points (small_J2.cpp:122-173) ---------------------------
  ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
    auto& r = ex.lane(lane);
    int const pt = lane >> 3, d = lane & 7;
    load_params(r.m, mt, e);
    load_point(sh, pt, r.g, PREV);
    C8_UNROLL
    for (int j = 0; j < NL; ++j) {
      r.m.xi_prev[j] = Dual(sh.xip[pt][j]);
      r.m.xi[j] = Dual(sh.xi[pt][j], (j == d) ? 1. : 0.);
      r.m.R[j] = Dual(0.);
    }
    if (!ADJOINT) r.trial = guess_and_trial_values<ModelT>(r.m, r.g);  // the guess, and the trial state of the iteration below
    r.iter = 1;
    r.R_norm_0 = 1.;
    r.converged = !Model::HAS_LOCAL;
    if (ADJOINT) {  // evaluate at the stored state with xi seeded -> dC/dxi; dJ/dxi; g -= dJ/dxi  (:442-446, :474-481)
      r.converged = true;
      if (Model::HAS_LOCAL) {
        r.m.evaluate(r.g, ms.abs_tol, trial_values<ModelT>(r.m, r.g));  // xi seeded: the trial state carries no tangent
        if (d < NL) {
          C8_UNROLL
          for (int j = 0; j < NL; ++j) sh.M[pt][j][d] = r.m.R[j].d;
        }
      }
      double const dJ_dxi = QoI::evaluate(r.g, r.m, sh.wdv[pt], aa.qoi, (size_t)e * E::NP0 + pt).d;
      if (d < NL) {
        size_t const qg = ((size_t)e * E::NP0 + pt) * NL + d;
        double const gk = aa.g[qg] - dJ_dxi;
        aa.g[qg] = gk;
        sh.gh[pt][d] = gk;
      }
    }
  });
  if constexpr (uses_line_search<Model>::value) {
    if (!ADJOINT) local_newton_line_search_wave<NL, PIN, ModelT>(ex, sh, ms);
  } else if (Model::HAS_LOCAL && !ADJOINT) {
    auto running = [&](int lane) { auto& r = ex.lane(lane); return (r.iter <= ms.max_iters) && !r.converged; };
    while (ex.any(running)) {
      ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
        auto& r = ex.lane(lane);
        if (!running(lane)) return;
        int const pt = lane >> 3, d = lane & 7;
        r.m.evaluate(r.g, ms.abs_tol, r.trial);
        double nrm = 0.;
        C8_UNROLL
        for (int j = 0; j < NL; ++j) nrm += r.m.R[j].v * r.m.R[j].v;
        double const R_norm = sqrt(nrm);
        if (r.iter == 1) r.R_norm_0 = R_norm;
        double const R_norm_rel = R_norm / r.R_norm_0;  // NaN on elastic points: the abs test decides
        if ((R_norm_rel < ms.rel_tol) || (R_norm < ms.abs_tol)) r.converged = true;
        if (d < NL) {
          C8_UNROLL
          for (int j = 0; j < NL; ++j) sh.M[pt][j][d] = r.m.R[j].d;
        }
        C8_UNROLL
        for (int j = 0; j < NL; ++j) r.b[j] = -r.m.R[j].v;
      });
      ex.sync();
      if (!ex.any(running)) break;
      // column d of dC/dxi is this lane's own tangent: the elimination works on R[.].d in place
      bool ok;
      if constexpr (Model::NEWTON_MATRIX_IN_LDS)
        ok = gj_solve_grouped<NL, 8>(ex, [&](int lane) { return sh.M[lane >> 3]; },
                                     [&](int lane) { return ex.lane(lane).b; }, running);
      else
        ok = local_solve<NL, C8_XL_NEWTON(Model)>(ex, sh, running);
      ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
        auto& r = ex.lane(lane);
        if (!running(lane)) return;
        if (!ok) { r.failed = true; r.iter = ms.max_iters + 1; return; }
        C8_UNROLL
        for (int j = 0; j < NL; ++j) r.m.xi[j].v += r.b[j];
        r.iter++;
      });
    }
  }
  ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
    auto& r = ex.lane(lane);
    int const pt = lane >> 3, d = lane & 7;
    if ((r.iter > ms.max_iters) && !r.converged) r.failed = true;
    if (d == 0 && !ADJOINT) {  // local->scatter (local_residual.cpp:624-631)
      size_t const q = ((size_t)e * E::NP0 + pt) * NL;
      C8_UNROLL
      for (int j = 0; j < NL; ++j) { fa.xi[q + j] = r.m.xi[j].v; sh.xi[pt][j] = r.m.xi[j].v; }
      if (r.failed) sh.failed = 1;
    }
  });
  ex.sync();
  C8_STAMP(2);
  // ---- (dC/dxi)^-1 once per point, in the 8-lane layout: lane d solves for unit vector e_d.  Both passes
  //      of phase D then apply it as a matrix-vector product instead of two more eliminations.  When every
  //      point of the element took the elastic branch dC/dxi is exactly I (R = xi - xi_trial) and the
  //      elimination is skipped: solving I x = b returns b unchanged, so the result is bitwise the same.
  if (Model::HAS_LOCAL) {
    need_inverse = ex.any_wave([&](int lane) {
      int const pt = lane >> 3, d = lane & 7;
      bool notI = false;
      if (d < NL) {
        C8_UNROLL
        for (int j = 0; j < NL; ++j) notI = notI || (sh.M[pt][j][d] != ((j == d) ? 1. : 0.));
      }
      return notI;
    });
    if (need_inverse) {
      ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
        auto& r = ex.lane(lane);
        int const d = lane & 7;
        int const pt = lane >> 3;
        C8_UNROLL
        for (int j = 0; j < NL; ++j) {
          r.b[j] = (j == d) ? 1. : 0.;
          r.m.R[j].d = (d < NL) ? sh.M[pt][j][d] : 0.;  // this lane's column of dC/dxi
        }
      });
      bool const ok = local_solve<NL, C8_XL_INVERSE(Model)>(ex, sh, [](int) { return true; });
      ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
        auto& r = ex.lane(lane);
        int const pt = lane >> 3, d = lane & 7;
        if (d < NL) {
          C8_UNROLL
          for (int j = 0; j < NL; ++j) sh.M[pt][j][d] = r.b[j];
        }
        if (!ok && lane == 0) sh.failed = 1;
      });
      ex.sync();
    }
  }

  }  // !CLOSED
  if constexpr (CLOSED) {
    // the model's closed form (Model::closed_form) once per point, on the first lane of the point's eight: it reads the
    // point quantities and the previous state, stores the converged state and the weighted fluxes, and leaves what the
    // tangent columns need IN PLACE OF the point quantities (sh.q[pt][0 .. NT-1]; nothing reads q afterwards)
    static_assert(Model::ClosedForm::NT <= WQ, "the tangent data take the place of the point quantities");
    ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
      int const pt = lane >> 3, d = lane & 7;
      if (d != 0) return;
      int const es = mt.elem_set ? mt.elem_set[e] : 0;
      typename Model::ClosedForm cf;
      Model::closed_form(mt.params + (size_t)es * Model::NPARAMS, sh.q[pt], sh.xip[pt], ms.abs_tol, sh.h, ms.stab_mult, cf, true);
      size_t const q0 = ((size_t)e * E::NP0 + pt) * NL;
      C8_UNROLL
      for (int j = 0; j < NL; ++j) fa.xi[q0 + j] = cf.xi[j];
      double const w = sh.wdv[pt];
      double* Fp = sh.F[pt];
      C8_UNROLL
      for (int rr = 0; rr < WF; ++rr) Fp[rr] = w * cf.F[rr];
      C8_UNROLL
      for (int i = 0; i < Model::ClosedForm::NT; ++i) sh.q[pt][i] = cf.t[i];
    });
    ex.sync();
  }
  C8_STAMP(3);
  if constexpr (CLOSED) {
    // ---- phase P of the closed-form kernel: lane = (point half hf, column b), no D in LDS.  The model gives (D B)[.][b],
    //      the 13 flux derivatives with respect to this lane's element unknown, directly from the point's tangent data and
    //      the column node's shape entries (Model::closed_form_flux_column); the lane then takes ALL 32 rows of its column
    //      for the four points of its half, so the eight nodes' shape entries (the same for every lane of a half) are read
    //      once per point for 13 products.  J holds the rows (n, u_0), (n, u_1) (flux group 0), J1 the rows (n, u_2),
    //      (n, p) (group 1), in the order the scatter expects; the halves exchange their partial sums below.
    static_assert(!ADJOINT && !Mechanics::USES_U, "the closed-form kernel is the forward assembly of a weak form without u terms");
    ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
      auto& r = ex.lane(lane);
      int const b = lane & 31, hf = lane >> 5;
      bool const bu = b < 3 * E::NN;
      int const m = bu ? b / 3 : b - 3 * E::NN;
      int const k = bu ? b - 3 * m : 3;
      double const ek[3] = {k == 0 ? 1. : 0., k == 1 ? 1. : 0., k == 2 ? 1. : 0.};
      double const isp = bu ? 0. : 1.;
      int const cg = bu ? 3 * k : 10;
      // the first point of a half assigns the accumulators (a product instead of the first fused multiply-add: no pass that
      // zeroes 32 accumulators), the other three add to them
      auto point = [&](int q4, auto first) {
        constexpr bool FIRST = decltype(first)::value;
        int const pt = 4 * hf + q4;
        double const w = sh.wdv[pt];
        double const* const Gb = sh.G4[pt][m];
        double const b0 = Gb[0], b1 = Gb[1], b2 = Gb[2], bN = Gb[3];
        double const g[3] = {w * b0, w * b1, w * b2};
        double tn[Model::ClosedForm::NT];
        C8_UNROLL
        for (int i = 0; i < Model::ClosedForm::NT; ++i) tn[i] = sh.q[pt][i];
        double db[WF];
        Model::closed_form_flux_column(tn, ek, isp, g, w * bN, db);
        C8_UNROLL
        for (int n = 0; n < E::NN; ++n) {
          double const* const Ga = sh.G4[pt][n];
          double const a0 = Ga[0], a1 = Ga[1], a2 = Ga[2], aN = Ga[3];
          // one fused multiply-add per product, chained through the accumulator (a sum of products added afterwards costs
          // a multiplication and an addition more per entry: 17 instead of 13 instructions per node)
          r.J[2 * n] = fma(a2, db[2], fma(a1, db[1], FIRST ? a0 * db[0] : fma(a0, db[0], r.J[2 * n])));
          r.J[2 * n + 1] = fma(a2, db[5], fma(a1, db[4], FIRST ? a0 * db[3] : fma(a0, db[3], r.J[2 * n + 1])));
          r.J1[2 * n] = fma(a2, db[8], fma(a1, db[7], FIRST ? a0 * db[6] : fma(a0, db[6], r.J1[2 * n])));
          r.J1[2 * n + 1] = fma(aN, db[9], fma(a2, db[12], fma(a1, db[11], FIRST ? a0 * db[10] : fma(a0, db[10], r.J1[2 * n + 1]))));
        }
        double const* Fp = sh.F[pt];
        r.R += Fp[cg] * b0 + Fp[cg + 1] * b1 + Fp[cg + 2] * b2 + Fp[9] * (isp * bN);
      };
      point(0, std::true_type{});
      C8_NOUNROLL
      for (int q4 = 1; q4 < 4; ++q4) point(q4, std::false_type{});
    });
  } else
  // ---- phases D and P, 4 points per pass -------------------------------------------------------------
  for (int t = 0; t < 2; ++t) {
    ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
      auto& r = ex.lane(lane);
      int const ql = lane >> 4, c = lane & 15, pt = 4 * t + ql;
      load_params(r.m, mt, e);
      load_point(sh, pt, r.g, PREV);
      seed_q(r.g, c);
      C8_UNROLL
      for (int j = 0; j < NL; ++j) {
        r.m.xi_prev[j] = Dual(sh.xip[pt][j]);
        r.m.xi[j] = Dual(sh.xi[pt][j]);
        r.m.R[j] = Dual(0.);
        r.b[j] = 0.;
      }
      if (Model::HAS_LOCAL) {  // dC/dq_c with xi unseeded (evaluations.cpp:105-109)
        r.m.evaluate(r.g, ms.abs_tol);
        C8_UNROLL
        for (int j = 0; j < NL; ++j) r.b[j] = -r.m.R[j].d;
      }
      double dJ_dq = 0.;
      if (ADJOINT) dJ_dq = QoI::evaluate(r.g, r.m, sh.wdv[pt], aa.qoi, (size_t)e * E::NP0 + pt).d;  // x seeded, xi plain (:469-471)
      // dxi/dq_c = -(dC/dxi)^-1 dC/dq_c  (evaluations.cpp:112; local->seed_wrt_x, chain rule through q)
      if (Model::HAS_LOCAL && need_inverse) {
        C8_UNROLL
        for (int i = 0; i < NL; ++i) {
          double sacc = 0.;
          C8_UNROLL
          for (int j = 0; j < NL; ++j) sacc += sh.M[pt][i][j] * r.b[j];
          r.m.xi[i].d = sacc;
        }
      } else {
        C8_UNROLL
        for (int j = 0; j < NL; ++j) r.m.xi[j].d = r.b[j];
      }
      if (ADJOINT) {
        double v = -dJ_dq;
        C8_UNROLL
        for (int j = 0; j < NL; ++j) v += r.m.xi[j].d * sh.gh[pt][j];
        sh.rq[ql][c] = v;
      }
      MechFlux<Dual> f;
      Mechanics::flux_coupled(r.m, r.g, sh.h, ms.stab_mult, f);
      f.Vp = f.Vp + Mechanics::flux_pressure(r.m, r.g);  // ip set 1 has the same points and weights
      // the quadrature weight w*dv goes into D and F here (13 products per lane and pass) rather than
      // into every product of phase P
      double const w = sh.wdv[pt];
      double* Dc = &sh.D[ql][0][c];
      constexpr int LD = WQ + 1;
      Dc[0 * LD] = w * f.Gu.xx.d; Dc[1 * LD] = w * f.Gu.xy.d; Dc[2 * LD] = w * f.Gu.xz.d;
      Dc[3 * LD] = w * f.Gu.yx.d; Dc[4 * LD] = w * f.Gu.yy.d; Dc[5 * LD] = w * f.Gu.yz.d;
      Dc[6 * LD] = w * f.Gu.zx.d; Dc[7 * LD] = w * f.Gu.zy.d; Dc[8 * LD] = w * f.Gu.zz.d;
      Dc[9 * LD] = w * f.Vp.d;
      Dc[10 * LD] = w * f.Gp[0].d; Dc[11 * LD] = w * f.Gp[1].d; Dc[12 * LD] = w * f.Gp[2].d;
      if (c == 0 && !ADJOINT) {
        double* Fp = sh.F[pt];
        Fp[0] = w * f.Gu.xx.v; Fp[1] = w * f.Gu.xy.v; Fp[2] = w * f.Gu.xz.v;
        Fp[3] = w * f.Gu.yx.v; Fp[4] = w * f.Gu.yy.v; Fp[5] = w * f.Gu.yz.v;
        Fp[6] = w * f.Gu.zx.v; Fp[7] = w * f.Gu.zy.v; Fp[8] = w * f.Gu.zz.v;
        Fp[9] = w * f.Vp.v;
        Fp[10] = w * f.Gp[0].v; Fp[11] = w * f.Gp[1].v; Fp[12] = w * f.Gp[2].v;
      }
    });
    ex.sync();
    C8_STAMP(4 + 2 * t);
    // phase P: lane = (g, column b).  The rows of the element matrix are split by flux group, so that no
    // product of the contraction W^T (D B) is formed twice:
    //   g = 0: entries (node n, u_0) from flux rows 0..2 and (n, u_1) from flux rows 3..5,   n = 0..7
    //   g = 1: entries (node n, u_2) from flux rows 6..8 and (n, p)   from flux rows 9..12
    // Both halves run one instruction stream: entry E0 takes rows r0..r0+2, entry E1 rows r1..r1+2 plus
    // row 9 with weight zf (zero for g = 0).  J[2n] = E0 of node n, J[2n+1] = E1 of node n.
    ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
      auto& r = ex.lane(lane);
      int const b = lane & 31, g = lane >> 5;
      bool const bu = b < 3 * E::NN;
      int const m = bu ? b / 3 : b - 3 * E::NN;
      int const k = bu ? b - 3 * m : 0;
      int const cg = bu ? 3 * k : 10;       // first gradient column of D that x_b drives
      int const cv = bu ? 13 + k : 9;       // value column of D that x_b drives
      // the weak form does not read u itself: the u-value column is multiplied by zero instead of being
      // skipped under a lane-dependent branch (a branch per row serialises the LDS reads of this loop)
      bool const has_cv = Mechanics::USES_U || !bu;
      int const cvl = has_cv ? cv : 9;
      int const r0 = 6 * g, r1 = 3 + 7 * g;
      double const zf = g ? 1. : 0.;
      C8_NOUNROLL
      for (int ql = 0; ql < 4; ++ql) {
        int const pt = 4 * t + ql;
        double const bN = sh.N[pt][m], b0 = sh.dN[pt][m][0], b1 = sh.dN[pt][m][1], b2 = sh.dN[pt][m][2];
        double const bNv = has_cv ? bN : 0.;
        double const bNp = bu ? 0. : bN;
        double s0, s1, s2, u0, u1, u2, u3, v0 = 0., v1 = 0.;
        if (!ADJOINT) {
          // (D B)[rr][b] = sum_c D[rr][c] dq_c/dx_b   (D already carries w*dv)
          auto DB = [&](int rr) {
            double const* Dr = sh.D[ql][rr];
            return Dr[cg] * b0 + Dr[cg + 1] * b1 + Dr[cg + 2] * b2 + Dr[cvl] * bNv;
          };
          s0 = DB(r0); s1 = DB(r0 + 1); s2 = DB(r0 + 2);
          u0 = DB(r1); u1 = DB(r1 + 1); u2 = DB(r1 + 2);
          u3 = zf * DB(9);
        } else {
          // transposed element matrix computed directly, so that the lanes of one scatter instruction
          // still share a CSR row: lane = element ROW a (flux side), entries = element COLUMNS (q side):
          // (W^T D)[a][c] = sum_r dR_a/dflux_r D[r][c]
          auto WD = [&](int c) {
            return sh.D[ql][cg][c] * b0 + sh.D[ql][cg + 1][c] * b1 + sh.D[ql][cg + 2][c] * b2 + sh.D[ql][9][c] * bNp;
          };
          s0 = WD(r0); s1 = WD(r0 + 1); s2 = WD(r0 + 2);
          u0 = WD(r1); u1 = WD(r1 + 1); u2 = WD(r1 + 2);
          u3 = zf * WD(9);
          if (Mechanics::USES_U) { v0 = WD(13 + 2 * g); v1 = (1. - zf) * WD(14); }
        }
        C8_UNROLL
        for (int n = 0; n < E::NN; ++n) {
          double const a0 = sh.dN[pt][n][0], a1 = sh.dN[pt][n][1], a2 = sh.dN[pt][n][2], aN = sh.N[pt][n];
          // fused multiply-adds chained through the accumulator (see the closed-form phase P)
          double j0 = fma(a2, s2, fma(a1, s1, fma(a0, s0, r.J[2 * n])));
          double j1 = fma(aN, u3, fma(a2, u2, fma(a1, u1, fma(a0, u0, r.J[2 * n + 1]))));
          if constexpr (ADJOINT && Mechanics::USES_U) {
            j0 = fma(aN, v0, j0);
            j1 = fma(aN, v1, j1);
          }
          r.J[2 * n] = j0;
          r.J[2 * n + 1] = j1;
        }
        if (!ADJOINT) {  // residual entry b from the flux values (both halves compute it, half 0 stores it)
          double const* Fp = sh.F[pt];
          r.R += Fp[cg] * b0 + Fp[cg + 1] * b1 + Fp[cg + 2] * b2 + Fp[9] * bNp;
        } else {  // rhs_b = sum_c [-dJ/dq_c + (dxi/dq_c).g] dq_c/dx_b + f_b  (:486-487)
          double const* rq = sh.rq[ql];
          r.R += rq[cg] * b0 + rq[cg + 1] * b1 + rq[cg + 2] * b2 + rq[cv] * bN +
                 aa.f[((size_t)e * E::NP0 + pt) * E::NDOF + b];
        }
      }
    });
    ex.sync();
    C8_STAMP(5 + 2 * t);
  }
  if constexpr (CLOSED) {
    // the halves exchange their partial sums: half 0 completes group 0 (its J and the other half's J), half 1 group 1 (the
    // other half's J1 and its own), both into J, which the scatter reads.  One lane-pair exchange per accumulator
    // (EX::pair_sum32: v_permlane32_swap on the device, no LDS traffic and no selects)
    ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
      auto& r = ex.lane(lane);
      C8_UNROLL
      for (int k = 0; k < 16; ++k)
        r.J[k] = ex.pair_sum32(lane, [&](int l) { return ex.lane(l).J[k]; }, [&](int l) { return ex.lane(l).J1[k]; });
      double const Rr = ex.xor32(lane, [&](int l) { return ex.lane(l).R; });
      r.Rx = Rr;
    });
    ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
      auto& r = ex.lane(lane);
      r.R += r.Rx;
    });
  }

  }
  C8_STAMP(8);
  // ---- scatter: lane (g, b) holds column b of the rows (n, u_2g) [J[2n]] and (n, u_1) or (n, p) [J[2n+1]];
  //      the adjoint assembly stores them transposed (evaluations.cpp:463-465) ------------------------------
  ex.each([&](int lane_) { int lane = lane_; if constexpr (PIN) C8_PIN(lane);
    auto& r = ex.lane(lane);
    int const b = lane & 31, g = lane >> 5;
    int ib, nb, eqb;
    slot_to_dof<E>(b, ib, nb, eqb);
    int const neqb = ib == 0 ? 3 : 1;
    // forward: J[(n,i)][b]; adjoint: the lane already holds the transposed entries J[b][(n,i)], so in both
    // cases this is assembled entry (row (n,i), column b) and the lanes of one instruction share 2 rows.
    // One uniform branch selects atomic or plain adds for the whole batch (a branch per add would keep the
    // adds from being issued back to back).
    double* const A0 = sa.A[0][ib];                 // u rows: block (0, ib)
    double* const A1 = g ? sa.A[1][ib] : A0;        // second entry: p row (block (1, ib)) or u row 1
    auto scatter_all = [&](auto mode) {
      constexpr int ATOMIC = decltype(mode)::value;
      C8_UNROLL
      for (int n = 0; n < E::NN; ++n) {
        size_t const nptr = (size_t)sh.nptr[n], deg = (size_t)sh.deg[n], pos = (size_t)((r.pos8 >> (8 * n)) & 0xff);  // pos[e][nb][row node n]
        size_t const in_row = pos * neqb + eqb;
        size_t const urow0 = nptr * (3 * neqb);     // first u row of node n in block (0, ib)
        ex.add(A0 + urow0 + (size_t)(2 * g) * deg * neqb + in_row, r.J[2 * n], ATOMIC);
        ex.add(A1 + (g ? nptr * neqb : urow0 + deg * neqb) + in_row, r.J[2 * n + 1], ATOMIC);
      }
      if (g == 0) ex.add(sa.b[ib] + (size_t)sh.node[nb] * neqb + eqb, r.R, ATOMIC);
    };
    if (sa.stage) {  // staged assembly: registers -> stage[e]
      double* const st = sa.stage + (size_t)(e % sa.stage_ring) * stage_stride<E>();
      C8_UNROLL
      for (int n = 0; n < E::NN; ++n) {  // two runs of 32 contiguous values per store
        C8_STREAM_STORE(st + stage_row<E>(n, 2 * g) + b, r.J[2 * n]);            // row u_2g of node n
        C8_STREAM_STORE(st + stage_row<E>(n, g ? 3 : 1) + b, r.J[2 * n + 1]);    // row p or u_1
      }
      if (g == 0) st[E::NN * 4 * E::NDOF + b] = r.R;
    } else if (sa.atomic) {
      scatter_all(std::integral_constant<int, 1>{});
    } else {
      scatter_all(std::integral_constant<int, 0>{});
    }
    if (lane == 0 && sh.failed) ex.flag(sa.status);
  });
  C8_STAMP(9);
}

// ---- staged assembly, second half: the rows of one node summed over the node's elements ---------------------
// One wavefront per node, any element type.  stage[e] holds, per element node, that node's four rows (u_0, u_1,
// u_2, p; NDOF columns each) contiguously (stage_row).  The wave adds the contributions of the node's elements,
// in ascending element order, into acc[pos][row][col] (LDS: pos = position of the column node in this node's graph
// row) and then adds the finished rows to the four CSR blocks and the node's residual entries to b with
// contiguous accesses.
template <class E, int MAXDEG> struct GatherShared {
  // one double of padding per accumulator row: 80 % of the LDS cycles of the unpadded form were bank conflicts; it is
  // worth 0.6 % of an assembly (the kernel waits on HBM, not on LDS)
  static constexpr int LD = 17;
  double acc[MAXDEG][LD];
};

template <class E, int MAXDEG> struct GatherLane {
  static constexpr int N00 = (9 * MAXDEG + 63) / 64, N01 = (3 * MAXDEG + 63) / 64;
  static constexpr int NLD = (4 * E::NDOF + 63) / 64;  // loads per lane and element: hex8 2, tet4 1
#ifdef C8_TUNE_GATHER_CH  // tuning build (same results)
  static constexpr int CH = C8_TUNE_GATHER_CH;
#else
  // elements whose rows are in flight together in one wavefront.  Two, not all eight: 72 registers and seven waves per
  // SIMD hide the latency better than eight loads in flight at four waves per SIMD (11.13 against 11.39 ms per assembly)
  static constexpr int CH = 2;
#endif
  double v[CH][NLD], rv[CH];
  int pos[CH][NLD];
  double a00[N00], a01[N01], a10[N01], a11;  // current values of this lane's CSR entries
  double bold;                               // ... and of its residual entry (lanes 0..3)
  double rsum;
};

// value index (row, column of the node's 4 x NDOF block) of load j of a lane
template <int NLD> C8_HD int gather_idx(int lane, int j) {
  return NLD == 2 ? 2 * lane + j : lane + 64 * j;  // two values per lane: adjacent, fetched by one 16-byte load
}

template <class E, int MAXDEG, class EX>
C8_HD void gather_node_rows(EX& ex, GatherShared<E, MAXDEG>& sh, GatherArgs const& ga, int node) {
  using GL = GatherLane<E, MAXDEG>;
  constexpr int CH = GL::CH, NLD = GL::NLD, NB = 4 * E::NDOF;
  int const nptr = ga.nodeptr[node], deg = ga.nodeptr[node + 1] - nptr;
  int const e0 = ga.nodeelem_ptr[node], e1 = ga.nodeelem_ptr[node + 1];
  size_t const np = (size_t)nptr;
  int const n3 = 3 * deg;
  ex.each([&](int lane) {
    auto& r = ex.lane(lane);
    // the CSR entries this lane will update: loaded now, so that their round trip overlaps the stage loads.
    // Block (0,0): the node's three u rows are contiguous (3*deg entries each), one run of 9*deg values;
    // blocks (0,1) and (1,0): runs of 3*deg values; block (1,1): deg values.
    // (assign mode: nothing to read, the rows start from zero -- uniform over the launch)
    C8_UNROLL
    for (int it = 0; it < GL::N00; ++it) {
      int const j = lane + 64 * it;
      r.a00[it] = 0.;
      if (!ga.assign && j < 9 * deg) r.a00[it] = ga.A[0][0][np * 9 + j];
    }
    C8_UNROLL
    for (int it = 0; it < GL::N01; ++it) {
      int const j = lane + 64 * it;
      r.a01[it] = r.a10[it] = 0.;
      if (!ga.assign && j < n3) { r.a01[it] = ga.A[0][1][np * 3 + j]; r.a10[it] = ga.A[1][0][np * 3 + j]; }
    }
    r.a11 = 0.;
    if (!ga.assign && lane < deg) r.a11 = ga.A[1][1][np + lane];
    r.bold = 0.;
    if (!ga.assign && lane < 4) r.bold = lane < 3 ? ga.b[0][(size_t)node * 3 + lane] : ga.b[1][node];
    constexpr int LDA = GatherShared<E, MAXDEG>::LD;
    C8_UNROLL
    for (int it = 0; it < (MAXDEG * LDA + 63) / 64; ++it) {
      int const q = lane + 64 * it;
      if (q < deg * LDA) (&sh.acc[0][0])[q] = 0.;
    }
    r.rsum = 0.;
  });
  ex.sync();
  for (int c0 = e0; c0 < e1; c0 += CH) {
    // all loads of the chunk first (independent HBM round trips), then the adds in element order
    ex.each([&](int lane) {
      auto& r = ex.lane(lane);
      C8_UNROLL
      for (int k = 0; k < CH; ++k) {
        if (c0 + k < e1) {
          int const packed = ga.nodeelem[c0 + k];
          int const e = packed >> 3, ln = packed & 7;
          double const* const st = ga.stage + (size_t)(e % ga.stage_ring) * stage_stride<E>();
          C8_UNROLL
          for (int j = 0; j < NLD; ++j) {
            int const idx = gather_idx<NLD>(lane, j);  // (row rr, column c) of the node's block: idx = rr * NDOF + c
            r.pos[k][j] = -1;
            if (idx < NB) {
              int const c = idx % E::NDOF;
              int const m = c < 3 * E::NN ? c / 3 : c - 3 * E::NN;
              r.pos[k][j] = ga.pos[((size_t)e * E::NN + m) * E::NN + ln];
            }
          }
          if (NLD == 2) {  // one 16-byte load: values 2 lane and 2 lane + 1 (two 8-byte loads per lane: +2 % per assembly)
            C8_STREAM_LOAD2(st + stage_row<E>(ln, 0) + 2 * lane, r.v[k][0], r.v[k][NLD - 1]);
          } else if (lane < NB) {
            r.v[k][0] = C8_STREAM_LOAD(st + stage_row<E>(ln, 0) + lane);
          }
          r.rv[k] = (lane < 4) ? st[E::NN * 4 * E::NDOF + (lane < 3 ? 3 * ln + lane : 3 * E::NN + ln)] : 0.;
        }
      }
    });
    C8_UNROLL
    for (int k = 0; k < CH; ++k) {
      if (c0 + k >= e1) break;
      ex.each([&](int lane) {
        auto& r = ex.lane(lane);
        C8_UNROLL
        for (int j = 0; j < NLD; ++j) {
          int const idx = gather_idx<NLD>(lane, j);
          if (idx < NB) {
            int const rr = idx / E::NDOF, c = idx % E::NDOF;
            int const col = c < 3 * E::NN ? c % 3 : 3;
            sh.acc[r.pos[k][j]][rr * 4 + col] += r.v[k][j];  // distinct addresses within one instruction
          }
        }
        r.rsum += r.rv[k];
      });
      ex.sync();
    }
  }
  ex.each([&](int lane) {
    auto& r = ex.lane(lane);
    // all fetched values complete HERE, once, outside the conditional stores below: with the waits inside the branches the
    // compiler's count of outstanding memory operations is lost at every join, and it then drains the counter -- the
    // previous store included -- in front of each store
    C8_UNROLL
    for (int it = 0; it < GL::N00; ++it) C8_PIN(r.a00[it]);
    C8_UNROLL
    for (int it = 0; it < GL::N01; ++it) { C8_PIN(r.a01[it]); C8_PIN(r.a10[it]); }
    C8_PIN(r.a11);
    C8_PIN(r.bold);
    C8_UNROLL
    for (int it = 0; it < GL::N00; ++it) {
      int const j = lane + 64 * it;
      if (j < 9 * deg) {
        int const i = (j >= n3) + (j >= 2 * n3), jj = j - i * n3, pos = jj / 3, col = jj - 3 * pos;  // i = j / n3 < 3
        ga.A[0][0][np * 9 + j] = r.a00[it] + sh.acc[pos][i * 4 + col];
      }
    }
    C8_UNROLL
    for (int it = 0; it < GL::N01; ++it) {
      int const j = lane + 64 * it;
      if (j < n3) {
        int const i = (j >= deg) + (j >= 2 * deg), pos = j - i * deg;  // block (0,1): rows u_i = j / deg < 3, deg entries each
        ga.A[0][1][np * 3 + j] = r.a01[it] + sh.acc[pos][i * 4 + 3];
        int const pos2 = j / 3, col = j - 3 * pos2;        // block (1,0): the p row, 3*deg entries
        ga.A[1][0][np * 3 + j] = r.a10[it] + sh.acc[pos2][3 * 4 + col];
      }
    }
    if (lane < deg) ga.A[1][1][np + lane] = r.a11 + sh.acc[lane][15];
    if (lane < 3) ga.b[0][(size_t)node * 3 + lane] = r.bold + r.rsum;
    if (lane == 3) ga.b[1][node] = r.bold + r.rsum;
  });
  ex.sync();
}

template <class E, template <class> class ModelT, class EX, class SH>
C8_HD void forward_jacobian_wave(EX& ex, SH& sh, MeshTables const& mt,
                                 ModelSettings const& ms, FieldArgs const& fa, SystemArgs const& sa, int e) {
  jacobian_wave<E, ModelT, PointQoi, false>(ex, sh, mt, ms, fa, AdjointArgs{}, sa, e);
}
// K1 with the model's closed form in place of the local Newton iteration and the AD passes (Model::HAS_CLOSED_FORM)
template <class E, template <class> class ModelT, class EX, class SH>
C8_HD void forward_jacobian_wave_closed(EX& ex, SH& sh, MeshTables const& mt,
                                        ModelSettings const& ms, FieldArgs const& fa, SystemArgs const& sa, int e) {
  jacobian_wave<E, ModelT, PointQoi, false, true>(ex, sh, mt, ms, fa, AdjointArgs{}, sa, e);
}

template <class E, template <class> class ModelT, class QoI, class EX, class SH>
C8_HD void adjoint_jacobian_wave(EX& ex, SH& sh, MeshTables const& mt,
                                 ModelSettings const& ms, FieldArgs const& fa, AdjointArgs const& aa,
                                 SystemArgs const& sa, int e) {
  jacobian_wave<E, ModelT, QoI, true>(ex, sh, mt, ms, fa, aa, sa, e);
}

// =====================================================================================
// K4 / K5 with one wavefront per element: lane = point*8 + direction, all 8 coupled points at once.
// =====================================================================================
template <class E, int NL> struct WaveSharedA {
  static constexpr int NLP = 8;
  double X[E::NN][3];
  double u[E::NN][3], p[E::NN];
  double u_prev[E::NN][3];
  double N[E::NP0][E::NN];
  double dN[E::NP0][E::NN][3];
  double wdv[E::NP0];
  double M[E::NP0][NLP][NLP + 1];
  double q[E::NP0][WQ];
  double qprev[E::NP0][9];
  double xi[E::NP0][NLP];
  double xip[E::NP0][NLP];
  double z[E::NDOF];                // element adjoint solution, slot order
  double zq[E::NP0][WQ];            // the adjoint field interpolated like q: grad z_u (0..8), z_p (9), grad z_p (10..12)
  double vec[E::NP0][NLP];          // right-hand side / phi exchange
  double wq[E::NP0][WQ];            // (dC/dq_prev)^T phi per point
  double h;
  int32_t node[E::NN];
  int32_t nptr[E::NN], deg[E::NN];
  int32_t failed;
};

template <template <class> class ModelT> struct WaveLaneA {
  using Model = ModelT<Dual>;
  Model m;
  PointState<Dual> g;
  double b[Model::NLOC];
  double acc;
  int slot;
  double xi_pre, xip_pre;  // local state of this lane's (point, direction), loaded with the first loads of the element
  double dn[3], sx;        // cached shape tables: this lane's dN/dx entry; w dv (lanes 0..7) or h (lane 8)
};


// (d flux / ds) . (interpolated adjoint) for the tangent s carried by f: the point form of (dR/ds)^T z
template <class SH> C8_HD double flux_dot_zq(SH const& sh, int pt, MechFlux<Dual> const& f) {
  double const* z = sh.zq[pt];
  double s = f.Vp.d * z[9] + f.Gp[0].d * z[10] + f.Gp[1].d * z[11] + f.Gp[2].d * z[12];
  s += f.Gu.xx.d * z[0] + f.Gu.xy.d * z[1] + f.Gu.xz.d * z[2];
  s += f.Gu.yx.d * z[3] + f.Gu.yy.d * z[4] + f.Gu.yz.d * z[5];
  s += f.Gu.zx.d * z[6] + f.Gu.zy.d * z[7] + f.Gu.zz.d * z[8];
  return s * sh.wdv[pt];
}

// common prologue: nodal data, shape tables, point quantities, adjoint point quantities, local state
template <class E, int NL, bool PREV, class EX, class SH>
C8_HD void wave_prologue(EX& ex, SH& sh, MeshTables const& mt, FieldArgs const& fa, AdjointArgs const& aa, int e) {
  ex.each([&](int lane) {
    if (lane == 0) sh.failed = 0;
    {
      auto& r = ex.lane(lane);
      int const pt = lane >> 3, d = lane & 7;
      r.xi_pre = r.xip_pre = 0.;
      if (d < NL) {
        size_t const q = ((size_t)e * E::NP0 + pt) * NL;
        r.xip_pre = fa.xi_prev[q + d];
        r.xi_pre = fa.xi[q + d];
      }
      if (mt.shape) load_cached_shape<E>(r, mt, e, lane);
    }
    if (lane < E::NDOF) {
      int i, n, eq;
      slot_to_dof<E>(lane, i, n, eq);
      int const node = mt.conn[e * E::NN + n];
      if (i == 0) {
        if (!mt.shape) sh.X[n][eq] = mt.coords[(size_t)node * 3 + eq];  // only the shape tables read the coordinates
        sh.u[n][eq] = fa.u[(size_t)node * 3 + eq];
        if (PREV) sh.u_prev[n][eq] = fa.u_prev[(size_t)node * 3 + eq];
        sh.z[lane] = aa.z_u[(size_t)node * 3 + eq];
      } else {
        sh.p[n] = fa.p[node];
        sh.node[n] = node;
        sh.z[lane] = aa.z_p[node];
      }
    }
    if (mt.shape) commit_cached_shape<E>(ex.lane(lane), sh, lane);
  });
  ex.sync();
  if (!mt.shape) {
    ex.each([&](int lane) {
      shape_entry<E>(sh, 0, lane >> 3, lane & 7, (lane & 7) + 1);
      if (lane == 0) sh.h = elem_size<E>(sh);
    });
    ex.sync();
  }
  ex.each([&](int lane) {
    int const pt = lane >> 3, d = lane & 7;
    C8_UNROLL
    for (int h = 0; h < 2; ++h) {
      int const k = 2 * d + h, a = k >> 2, b = k & 3;  // product (a, b), see interp_ab
      sh.q[pt][q_index(a, b)] = interp_ab<E>(sh, pt, a, b, &sh.u[0][0], &sh.p[0]);
      // the adjoint nodal values, same products (z = [u part: 3n+i | p part: 3 NN + n]); u[a] has no adjoint twin
      double const zv = interp_ab<E>(sh, pt, a, b, &sh.z[0], &sh.z[3 * E::NN]);
      sh.zq[pt][q_index(a, b)] = (a < 3 && b == 3) ? 0. : zv;
    }
    if (PREV) {
      int const a = (d >= 3) + (d >= 6), b = d - 3 * a;
      sh.qprev[pt][d] = interp_ab<E>(sh, pt, a, b, &sh.u_prev[0][0], &sh.p[0]);
      if (d == 0) sh.qprev[pt][8] = interp_ab<E>(sh, pt, 2, 2, &sh.u_prev[0][0], &sh.p[0]);
    }
    if (d < NL) {
      auto& r = ex.lane(lane);
      sh.xip[pt][d] = r.xip_pre;
      sh.xi[pt][d] = r.xi_pre;
    }
  });
  ex.sync();
}

// K4: solve_adjoint_local (evaluations.cpp:528-659)
template <class E, template <class> class ModelT, class EX>
C8_HD void adjoint_local_wave(EX& ex, WaveSharedA<E, ModelT<Dual>::NLOC>& sh, MeshTables const& mt, ModelSettings const& ms,
                              FieldArgs const& fa, AdjointArgs const& aa, SystemArgs const& sa, int e) {
  using Model = ModelT<Dual>;
  constexpr int NL = Model::NLOC;
  constexpr bool PREV = Model::FINITE_DEF;
  static_assert(E::NDOF == 32 && E::NP0 == 8, "wave kernel needs a hex8-like element");
  if (!Model::HAS_LOCAL) {  // dC/dxi = 0: Eigen's rank-0 solve gives phi = 0, hence f = g = 0
    ex.each([&](int lane) {
      int const pt = lane >> 3, d = lane & 7;
      size_t const qp = (size_t)e * E::NP0 + pt;
      if (d < NL) { aa.phi[qp * NL + d] = 0.; aa.g[qp * NL + d] = 0.; }
      C8_UNROLL
      for (int t = 0; t < 4; ++t) aa.f[(size_t)e * E::NP0 * E::NDOF + t * 64 + lane] = 0.;
    });
    return;
  }
  wave_prologue<E, NL, PREV>(ex, sh, mt, fa, aa, e);
  // xi seeded: (dR/dxi)^T z and dC/dxi^T; rhs = g - (dR/dxi)^T z  (:613-622)
  ex.each([&](int lane) {
    auto& r = ex.lane(lane);
    int const pt = lane >> 3, d = lane & 7;
    size_t const qp = (size_t)e * E::NP0 + pt;
    load_params(r.m, mt, e);
    load_point(sh, pt, r.g, PREV);
    C8_UNROLL
    for (int j = 0; j < NL; ++j) {
      r.m.xi_prev[j] = Dual(sh.xip[pt][j]);
      r.m.xi[j] = Dual(sh.xi[pt][j], (j == d) ? 1. : 0.);
      r.m.R[j] = Dual(0.);
    }
    MechFlux<Dual> f;
    Mechanics::flux_coupled(r.m, r.g, sh.h, ms.stab_mult, f);
    double const dRz = flux_dot_zq(sh, pt, f);
    r.m.evaluate(r.g, ms.abs_tol, trial_values<ModelT>(r.m, r.g));  // xi seeded: the trial state carries no tangent
    if (d < NL) {
      C8_UNROLL
      for (int j = 0; j < NL; ++j) sh.M[pt][d][j] = r.m.R[j].d;  // transposed fill
      sh.vec[pt][d] = aa.g[qp * NL + d] - dRz;
    }
  });
  ex.sync();
  ex.each([&](int lane) {
    auto& r = ex.lane(lane);
    int const pt = lane >> 3, d = lane & 7;
    C8_UNROLL
    for (int j = 0; j < NL; ++j) {
      r.b[j] = sh.vec[pt][j];
      r.m.R[j].d = (d < NL) ? sh.M[pt][j][d] : 0.;  // this lane's column of (dC/dxi)^T
    }
  });
  bool const ok = local_solve<NL, Model::GJ_XLANE_K4>(ex, sh, [](int) { return true; });
  // phi; xi_prev seeded: g = -(dC/dxi_prev)^T phi (:636-642); x_prev seeded through q_prev: w = (dC/dq_prev)^T phi
  ex.each([&](int lane) {
    auto& r = ex.lane(lane);
    int const pt = lane >> 3, d = lane & 7;
    size_t const qp = (size_t)e * E::NP0 + pt;
    if (d == 0) {
      C8_UNROLL
      for (int j = 0; j < NL; ++j) aa.phi[qp * NL + j] = r.b[j];
      if (!ok) sh.failed = 1;
    }
    C8_UNROLL
    for (int j = 0; j < NL; ++j) { r.m.xi[j].d = 0.; r.m.xi_prev[j].d = (j == d) ? 1. : 0.; }
    r.m.evaluate(r.g, ms.abs_tol);
    double gk = 0.;
    C8_UNROLL
    for (int j = 0; j < NL; ++j) gk -= r.m.R[j].d * r.b[j];
    if (d < NL) aa.g[qp * NL + d] = gk;
    if (PREV) {
      C8_UNROLL
      for (int j = 0; j < NL; ++j) r.m.xi_prev[j].d = 0.;
      C8_UNROLL
      for (int round = 0; round < 2; ++round) {
        int const c = (round == 0) ? d : 8;
        Tens3<Dual>& G = r.g.grad_u_prev;
        G.xx.d = (c == 0) ? 1. : 0.; G.xy.d = (c == 1) ? 1. : 0.; G.xz.d = (c == 2) ? 1. : 0.;
        G.yx.d = (c == 3) ? 1. : 0.; G.yy.d = (c == 4) ? 1. : 0.; G.yz.d = (c == 5) ? 1. : 0.;
        G.zx.d = (c == 6) ? 1. : 0.; G.zy.d = (c == 7) ? 1. : 0.; G.zz.d = (c == 8) ? 1. : 0.;
        r.m.evaluate(r.g, ms.abs_tol);
        double wc = 0.;
        C8_UNROLL
        for (int j = 0; j < NL; ++j) wc += r.m.R[j].d * r.b[j];
        if (round == 0 || d == 0) sh.wq[pt][c] = wc;
      }
    }
  });
  ex.sync();
  // f = -(dC/dx_prev)^T phi = -B_prev^T w  (:628-633): 8 points x 32 element DOFs, 4 per lane
  ex.each([&](int lane) {
    C8_UNROLL
    for (int t = 0; t < 4; ++t) {
      int const idx = t * 64 + lane, pt = idx >> 5, b = idx & 31;
      double fv = 0.;
      if (PREV && b < 3 * E::NN) {
        int const m = b / 3, k = b - 3 * m;
        fv = -(sh.wq[pt][3 * k] * sh.dN[pt][m][0] + sh.wq[pt][3 * k + 1] * sh.dN[pt][m][1] + sh.wq[pt][3 * k + 2] * sh.dN[pt][m][2]);
      }
      aa.f[(size_t)e * E::NP0 * E::NDOF + idx] = fv;
    }
    if (lane == 0 && sh.failed) ex.flag(sa.status);
  });
}

// geometry and interpolation at point pt of element slot el of an eight-element group: stores (dx/dxi)^-1 and w dv,
// fills the point state (values only).  ZG, when not null, receives the interpolated adjoint quantities in the layout of
// the point quantities q: grad z_u (0..8), z_p (9), grad z_p (10..12).
template <class E, bool PREV, class SH>
C8_HD void group_point_state(SH& sh, int el, int pt, PointState<double>& gq, double* ZG) {
  double xi[3], w;
  E::point(0, pt, xi, w);
  double J[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}};           // J(a,b) = dx_b / dxi_a
  double Gu[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}};          // du_i / dxi_a
  double Gup[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}};
  double Gz[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}};
  double Gp[3] = {0., 0., 0.}, pv = 0., uv[3] = {0., 0., 0.}, Gzp[3] = {0., 0., 0.}, zpv = 0.;
  C8_UNROLL
  for (int n = 0; n < E::NN; ++n) {
    double g[3];
    E::dNdxi(n, xi, g);
    double const Nn = E::N(n, xi), pn = sh.p[el][n];
    pv += pn * Nn;
    if (ZG) zpv += sh.z[el][n][3] * Nn;
    C8_UNROLL
    for (int a = 0; a < 3; ++a) {
      Gp[a] += pn * g[a];
      if (ZG) Gzp[a] += sh.z[el][n][3] * g[a];
      C8_UNROLL
      for (int b = 0; b < 3; ++b) {
        J[a][b] += g[a] * sh.X[el][n][b];
        Gu[b][a] += sh.u[el][n][b] * g[a];
        if (PREV) Gup[b][a] += sh.u_prev[el][n][b] * g[a];
        if (ZG) Gz[b][a] += sh.z[el][n][b] * g[a];
      }
    }
    C8_UNROLL
    for (int b = 0; b < 3; ++b) uv[b] += sh.u[el][n][b] * Nn;
  }
  Tens3<double> Jt;
  Jt.xx = J[0][0]; Jt.xy = J[0][1]; Jt.xz = J[0][2];
  Jt.yx = J[1][0]; Jt.yy = J[1][1]; Jt.yz = J[1][2];
  Jt.zx = J[2][0]; Jt.zy = J[2][1]; Jt.zz = J[2][2];
  double const dJ = det(Jt);
  Tens3<double> const Ji = inverse(Jt);
  double const ji[9] = {Ji.xx, Ji.xy, Ji.xz, Ji.yx, Ji.yy, Ji.yz, Ji.zx, Ji.zy, Ji.zz};
  C8_UNROLL
  for (int q = 0; q < 9; ++q) sh.Ji[el][pt][q] = ji[q];
  sh.wdv[el][pt] = w * dJ;
  // d/dx_l = sum_a Ji[3l+a] d/dxi_a
  auto phys = [&](double const* gx, int l) { return ji[3 * l] * gx[0] + ji[3 * l + 1] * gx[1] + ji[3 * l + 2] * gx[2]; };
  gq.grad_u.xx = phys(Gu[0], 0); gq.grad_u.xy = phys(Gu[0], 1); gq.grad_u.xz = phys(Gu[0], 2);
  gq.grad_u.yx = phys(Gu[1], 0); gq.grad_u.yy = phys(Gu[1], 1); gq.grad_u.yz = phys(Gu[1], 2);
  gq.grad_u.zx = phys(Gu[2], 0); gq.grad_u.zy = phys(Gu[2], 1); gq.grad_u.zz = phys(Gu[2], 2);
  if (PREV) {
    gq.grad_u_prev.xx = phys(Gup[0], 0); gq.grad_u_prev.xy = phys(Gup[0], 1); gq.grad_u_prev.xz = phys(Gup[0], 2);
    gq.grad_u_prev.yx = phys(Gup[1], 0); gq.grad_u_prev.yy = phys(Gup[1], 1); gq.grad_u_prev.yz = phys(Gup[1], 2);
    gq.grad_u_prev.zx = phys(Gup[2], 0); gq.grad_u_prev.zy = phys(Gup[2], 1); gq.grad_u_prev.zz = phys(Gup[2], 2);
  } else {
    gq.grad_u_prev = scale(0., eye3<double>());
  }
  gq.p = pv;
  C8_UNROLL
  for (int l = 0; l < 3; ++l) { gq.grad_p[l] = phys(Gp, l); gq.u[l] = uv[l]; }
  if (ZG) {
    C8_UNROLL
    for (int i = 0; i < 3; ++i)
      C8_UNROLL
      for (int l = 0; l < 3; ++l) ZG[3 * i + l] = phys(Gz[i], l);
    ZG[9] = zpv;
    C8_UNROLL
    for (int l = 0; l < 3; ++l) ZG[10 + l] = phys(Gzp, l);
  }
}
// mean-square edge length of element slot el (mechanics.cpp:103-113)
template <class E, class SH> C8_HD double group_elem_size(SH const& sh, int el) {
  double hh = 0.;
  C8_UNROLL
  for (int ed = 0; ed < E::NEDGES; ++ed) {
    int a, b;
    E::edge(ed, a, b);
    double const dx = sh.X[el][b][0] - sh.X[el][a][0], dy = sh.X[el][b][1] - sh.X[el][a][1], dz = sh.X[el][b][2] - sh.X[el][a][2];
    hh += dx * dx + dy * dy + dz * dz;
  }
  return sqrt(hh / E::NEDGES);
}

// =====================================================================================
// K2 for hex8, eight elements per wavefront: eval_global_residual (evaluations.cpp:156-259), no AD.
// The slot-per-lane kernel evaluates every point on all 32 lanes of an element; here lane (element, point)
// evaluates the constitutive model and the point fluxes once, and lane (element, node) contracts them with the
// shape-function gradients of its node, rebuilt from the stored inverse Jacobians (9 values per point instead of
// a 24-value gradient table).  Adds into b are atomic (4 per lane).
// =====================================================================================
template <class E> struct ResidualWaveShared {
  static constexpr int NE = 8;  // elements per wavefront
  double X[NE][E::NN][3], u[NE][E::NN][3], p[NE][E::NN], u_prev[NE][E::NN][3];
  double z[1][1][4];              // (adjoint nodal values: used by the gradient kernel's layout only)
  double Ji[NE][E::NP0][9];       // (dx/dxi)^-1, row-major: dN/dx_l = sum_a Ji[3l+a] dN/dxi_a
  double wdv[NE][E::NP0];
  double F[NE][E::NP0][WF + 1];   // point fluxes, both ip sets fused (same points on hex8)
  double h[NE];
  int32_t node[NE][E::NN];
};
template <template <class> class ModelT> struct ResidualWaveLane {
  ModelT<double> m;
  PointState<double> g;
};

template <class E, template <class> class ModelT, class EX>
C8_HD void residual_wave8(EX& ex, ResidualWaveShared<E>& sh, MeshTables const& mt, ModelSettings const& ms,
                          FieldArgs const& fa, SystemArgs const& sa, int e0, int count) {
  using Model = ModelT<double>;
  constexpr int NL = Model::NLOC;
  constexpr bool PREV = Model::FINITE_DEF;
  static_assert(E::NN == 8 && E::NP0 == 8 && E::SAME_POINTS, "eight nodes, eight points, one point set");
  // ---- lane (element, node): nodal data ----
  ex.each([&](int lane) {
    int const el = lane >> 3, n = lane & 7;
    if (el >= count) return;
    int const node = mt.conn[(size_t)(e0 + el) * E::NN + n];
    sh.node[el][n] = node;
    C8_UNROLL
    for (int d = 0; d < 3; ++d) {
      sh.X[el][n][d] = mt.coords[(size_t)node * 3 + d];
      sh.u[el][n][d] = fa.u[(size_t)node * 3 + d];
      if (PREV) sh.u_prev[el][n][d] = fa.u_prev[(size_t)node * 3 + d];
    }
    sh.p[el][n] = fa.p[node];
  });
  ex.sync();
  // ---- lane (element, point): geometry, interpolation, model, fluxes ----
  ex.each([&](int lane) {
    int const el = lane >> 3, pt = lane & 7;
    if (el >= count) return;
    auto& r = ex.lane(lane);
    int const e = e0 + el;
    group_point_state<E, PREV>(sh, el, pt, r.g, nullptr);
    if (pt == 0) sh.h[el] = group_elem_size<E>(sh, el);
    int const es = mt.elem_set ? mt.elem_set[e] : 0;
    C8_UNROLL
    for (int q = 0; q < Model::NPARAMS; ++q) r.m.params[q] = mt.params[es * Model::NPARAMS + q];
    size_t const qp = ((size_t)e * E::NP0 + pt) * NL;
    C8_UNROLL
    for (int j = 0; j < NL; ++j) { r.m.xi[j] = fa.xi[qp + j]; r.m.xi_prev[j] = fa.xi_prev[qp + j]; }
  });
  ex.sync();
  ex.each([&](int lane) {
    int const el = lane >> 3, pt = lane & 7;
    if (el >= count) return;
    auto& r = ex.lane(lane);
    MechFlux<double> f;
    Mechanics::flux_coupled(r.m, r.g, sh.h[el], ms.stab_mult, f);
    f.Vp = f.Vp + Mechanics::flux_pressure(r.m, r.g);
    double* Fp = sh.F[el][pt];
    Fp[0] = f.Gu.xx; Fp[1] = f.Gu.xy; Fp[2] = f.Gu.xz;
    Fp[3] = f.Gu.yx; Fp[4] = f.Gu.yy; Fp[5] = f.Gu.yz;
    Fp[6] = f.Gu.zx; Fp[7] = f.Gu.zy; Fp[8] = f.Gu.zz;
    Fp[9] = f.Vp;
    Fp[10] = f.Gp[0]; Fp[11] = f.Gp[1]; Fp[12] = f.Gp[2];
  });
  ex.sync();
  // ---- lane (element, node): R_u[n][i] = sum_pt w dv Gu[i][.] . grad N_n ; R_p[n] likewise (residual_entry) ----
  ex.each([&](int lane) {
    int const el = lane >> 3, n = lane & 7;
    if (el >= count) return;
    double R[4] = {0., 0., 0., 0.};
    C8_UNROLL
    for (int pt = 0; pt < E::NP0; ++pt) {
      double xi[3], w, g[3];
      E::point(0, pt, xi, w);
      E::dNdxi(n, xi, g);
      double const* ji = sh.Ji[el][pt];
      double const wdv = sh.wdv[el][pt];
      double const d0 = (ji[0] * g[0] + ji[1] * g[1] + ji[2] * g[2]) * wdv, d1 = (ji[3] * g[0] + ji[4] * g[1] + ji[5] * g[2]) * wdv,
                   d2 = (ji[6] * g[0] + ji[7] * g[1] + ji[8] * g[2]) * wdv;
      double const* Fp = sh.F[el][pt];
      R[0] += Fp[0] * d0 + Fp[1] * d1 + Fp[2] * d2;
      R[1] += Fp[3] * d0 + Fp[4] * d1 + Fp[5] * d2;
      R[2] += Fp[6] * d0 + Fp[7] * d1 + Fp[8] * d2;
      R[3] += Fp[9] * (E::N(n, xi) * wdv) + Fp[10] * d0 + Fp[11] * d1 + Fp[12] * d2;
    }
    int const node = sh.node[el][n];
    C8_UNROLL
    for (int i = 0; i < 3; ++i) ex.add(sa.b[0] + (size_t)node * 3 + i, R[i], 1);
    ex.add(sa.b[1] + node, R[3], 1);
  });
  ex.sync();
}

// =====================================================================================
// K5 for hex8, eight elements per wavefront: eval_qoi_gradient (evaluations.cpp:758-925).  Lane (element, point)
// walks the active parameters of its element set one after the other (one dual-number evaluation each):
// (dC/dp)^T phi + dJ/dp + (dR/dp)^T z at its point.  The one-element-per-wavefront kernel (param_gradient_wave) uses
// 8 lanes per point, of which only the active parameters (4 in the reference decks) work.
// =====================================================================================
template <class E> struct GradWaveShared {
  static constexpr int NE = 8;
  double X[NE][E::NN][3], u[NE][E::NN][3], p[NE][E::NN], u_prev[NE][E::NN][3];
  double z[NE][E::NN][4];
  double Ji[NE][E::NP0][9];
  double wdv[NE][E::NP0];
  double h[NE];
  double red[64];
};
template <template <class> class ModelT> struct GradWaveLane {
  ModelT<Dual> m;
  PointState<Dual> g;
  double acc[8];   // per active-parameter slot of this lane's element set
  int slot0;       // first gradient entry of that set (-1: none yet)
};

template <class EX> C8_HD void param_gradient_wave8_flush(EX& ex, double* red, AdjointArgs const& aa) {
  // one set in the whole wavefront (the usual case): sum over the lanes, one add per parameter; else per lane
  // lanes that have not met an element yet (slot0 < 0) hold zeros and do not count as another set
  int const s0 = ex.first_lane([&](int lane) { return ex.lane(lane).slot0; });
  bool const mixed = ex.any_wave([&](int lane) { int const q = ex.lane(lane).slot0; return q >= 0 && q != s0; }) || s0 < 0;
  static_for<8>([&](auto ac) {  // static index into acc
    constexpr int a = decltype(ac)::value;
    if (mixed) {
      ex.each([&](int lane) {
        auto& r = ex.lane(lane);
        if (r.slot0 >= 0 && r.acc[a] != 0.) ex.add(aa.out + r.slot0 + a, r.acc[a], 1);
      });
    } else {
      ex.each([&](int lane) { red[lane] = ex.lane(lane).acc[a]; });
      ex.sync();
      ex.each([&](int lane) {
        if (lane == 0) {
          double t = 0.;
          for (int k = 0; k < 64; ++k) t += red[k];
          if (t != 0.) ex.add(aa.out + s0 + a, t, 1);
        }
      });
      ex.sync();
    }
  });
  ex.each([&](int lane) {
    auto& r = ex.lane(lane);
    r.slot0 = -1;
    C8_UNROLL
    for (int a = 0; a < 8; ++a) r.acc[a] = 0.;
  });
}

template <class E, template <class> class ModelT, class QoI, bool CLOSED = false, class EX>
C8_HD void param_gradient_wave8(EX& ex, GradWaveShared<E>& sh, MeshTables const& mt, ModelSettings const& ms,
                                FieldArgs const& fa, AdjointArgs const& aa, int e0, int count) {
  using Model = ModelT<Dual>;
  constexpr int NL = Model::NLOC;
  constexpr bool PREV = Model::FINITE_DEF;
  static_assert(E::NN == 8 && E::NP0 == 8 && E::SAME_POINTS, "eight nodes, eight points, one point set");
  ex.each([&](int lane) {
    int const el = lane >> 3, n = lane & 7;
    if (el >= count) return;
    int const node = mt.conn[(size_t)(e0 + el) * E::NN + n];
    C8_UNROLL
    for (int d = 0; d < 3; ++d) {
      sh.X[el][n][d] = mt.coords[(size_t)node * 3 + d];
      sh.u[el][n][d] = fa.u[(size_t)node * 3 + d];
      if (PREV) sh.u_prev[el][n][d] = fa.u_prev[(size_t)node * 3 + d];
      sh.z[el][n][d] = aa.z_u[(size_t)node * 3 + d];
    }
    sh.p[el][n] = fa.p[node];
    sh.z[el][n][3] = aa.z_p[node];
  });
  ex.sync();
  ex.each([&](int lane) {
    int const el = lane >> 3, pt = lane & 7;
    if (el >= count || pt != 0) return;
    sh.h[el] = group_elem_size<E>(sh, el);
  });
  ex.sync();
  // a lane whose element belongs to another set than its previous element's: the sums so far go out first
  bool const moved = ex.any_wave([&](int lane) {
    int const el = lane >> 3;
    if (el >= count) return false;
    int const es = mt.elem_set ? mt.elem_set[e0 + el] : 0;
    int const s = ex.lane(lane).slot0;
    return s >= 0 && s != aa.active[es * 10];
  });
  if (moved) param_gradient_wave8_flush(ex, sh.red, aa);
  ex.each([&](int lane) {
    int const el = lane >> 3, pt = lane & 7;
    if (el >= count) return;
    auto& r = ex.lane(lane);
    int const e = e0 + el;
    int const es = mt.elem_set ? mt.elem_set[e] : 0;
    int32_t const* act = aa.active + es * 10;
    int const nact = act[1];
    r.slot0 = act[0];
    PointState<double> gq;
    double ZG[13];
    group_point_state<E, PREV>(sh, el, pt, gq, ZG);
    double const wdv = sh.wdv[el][pt];
    size_t const qp = (size_t)e * E::NP0 + pt;
    // point state as dual numbers without tangents
    r.g.grad_u.xx = Dual(gq.grad_u.xx); r.g.grad_u.xy = Dual(gq.grad_u.xy); r.g.grad_u.xz = Dual(gq.grad_u.xz);
    r.g.grad_u.yx = Dual(gq.grad_u.yx); r.g.grad_u.yy = Dual(gq.grad_u.yy); r.g.grad_u.yz = Dual(gq.grad_u.yz);
    r.g.grad_u.zx = Dual(gq.grad_u.zx); r.g.grad_u.zy = Dual(gq.grad_u.zy); r.g.grad_u.zz = Dual(gq.grad_u.zz);
    r.g.grad_u_prev.xx = Dual(gq.grad_u_prev.xx); r.g.grad_u_prev.xy = Dual(gq.grad_u_prev.xy); r.g.grad_u_prev.xz = Dual(gq.grad_u_prev.xz);
    r.g.grad_u_prev.yx = Dual(gq.grad_u_prev.yx); r.g.grad_u_prev.yy = Dual(gq.grad_u_prev.yy); r.g.grad_u_prev.yz = Dual(gq.grad_u_prev.yz);
    r.g.grad_u_prev.zx = Dual(gq.grad_u_prev.zx); r.g.grad_u_prev.zy = Dual(gq.grad_u_prev.zy); r.g.grad_u_prev.zz = Dual(gq.grad_u_prev.zz);
    r.g.p = Dual(gq.p);
    C8_UNROLL
    for (int l = 0; l < 3; ++l) { r.g.grad_p[l] = Dual(gq.grad_p[l]); r.g.u[l] = Dual(gq.u[l]); }
    double qv[WQ], xiv[NL], phv[NL];  // CLOSED: the model's closed form (Model::closed_form_param_gradient) instead of dual numbers
    if constexpr (CLOSED) {
      double const qq[WQ] = {gq.grad_u.xx, gq.grad_u.xy, gq.grad_u.xz, gq.grad_u.yx, gq.grad_u.yy, gq.grad_u.yz, gq.grad_u.zx, gq.grad_u.zy,
                             gq.grad_u.zz, gq.p, gq.grad_p[0], gq.grad_p[1], gq.grad_p[2], gq.u[0], gq.u[1], gq.u[2]};
      C8_UNROLL
      for (int k = 0; k < WQ; ++k) qv[k] = qq[k];
      C8_UNROLL
      for (int j = 0; j < NL; ++j) { xiv[j] = fa.xi[qp * NL + j]; phv[j] = aa.phi[qp * NL + j]; }
    }
    C8_NOUNROLL
    for (int a = 0; a < nact; ++a) {
      int const mine = act[2 + a];
      double s = 0.;
      if constexpr (CLOSED) {
        double const wload = aa.qoi.c_load * wdv;
        s = Model::closed_form_param_gradient(mt.params + (size_t)es * Model::NPARAMS, qv, xiv, ms.abs_tol, sh.h[el], ms.stab_mult, wdv, ZG,
                                              phv, mine, wload, aa.qoi.comp, wload != 0. ? aa.qoi.S + qp * 3 : qv);
      } else {
      C8_UNROLL
      for (int q = 0; q < Model::NPARAMS; ++q)
        r.m.params[q] = Dual(mt.params[es * Model::NPARAMS + q], (q == mine) ? 1. : 0.);
      C8_UNROLL
      for (int j = 0; j < NL; ++j) {
        r.m.xi_prev[j] = Dual(fa.xi_prev[qp * NL + j]);
        r.m.xi[j] = Dual(fa.xi[qp * NL + j]);
        r.m.R[j] = Dual(0.);
      }
      r.m.evaluate(r.g, ms.abs_tol);
      C8_UNROLL
      for (int j = 0; j < NL; ++j) s += r.m.R[j].d * aa.phi[qp * NL + j];      // (dC/dp)^T phi (:864-866)
      s += QoI::evaluate(r.g, r.m, wdv, aa.qoi, qp).d;                       // dJ/dp (:869-871)
      MechFlux<Dual> f;
      Mechanics::flux_coupled(r.m, r.g, sh.h[el], ms.stab_mult, f);
      f.Vp = f.Vp + Mechanics::flux_pressure(r.m, r.g);                      // both ip sets (same points)
      double t = f.Vp.d * ZG[9] + f.Gp[0].d * ZG[10] + f.Gp[1].d * ZG[11] + f.Gp[2].d * ZG[12];
      t += f.Gu.xx.d * ZG[0] + f.Gu.xy.d * ZG[1] + f.Gu.xz.d * ZG[2];
      t += f.Gu.yx.d * ZG[3] + f.Gu.yy.d * ZG[4] + f.Gu.yz.d * ZG[5];
      t += f.Gu.zx.d * ZG[6] + f.Gu.zy.d * ZG[7] + f.Gu.zz.d * ZG[8];
      s += t * wdv;                                                          // (dR/dp)^T z (:883-886)
      }
      // branch-free with static indices: a conditional update makes the compiler index acc dynamically (scratch)
      static_for<8>([&](auto ac) {
        constexpr int k = decltype(ac)::value;
        r.acc[k] += (k == a) ? s : 0.;
      });
    }
  });
  ex.sync();
}

// =====================================================================================
// K4 for hex8 models with a closed form of their local equations (Model::closed_form_local_adjoint), eight elements per
// wavefront: solve_adjoint_local (evaluations.cpp:528-659) without dual numbers and without the elimination of dC/dxi.
// Lane (element, point) rebuilds the point's geometry from the coordinates, interpolates grad u and grad z, takes the state
// and the tangent data from the model's closed form (as the row-per-node assemblies do) and writes phi and g; f, which a
// small-strain model leaves zero, is cleared by the wavefront as one contiguous block.
// =====================================================================================
template <class E, template <class> class ModelT, class EX>
C8_HD void adjoint_local_closed_wave8(EX& ex, GradWaveShared<E>& sh, MeshTables const& mt, ModelSettings const& ms,
                                      FieldArgs const& fa, AdjointArgs const& aa, int e0, int count) {
  using Model = ModelT<Dual>;
  constexpr int NL = Model::NLOC;
  static_assert(E::NN == 8 && E::NP0 == 8 && E::SAME_POINTS && !Model::FINITE_DEF, "hex8, small strain");
  ex.each([&](int lane) {
    int const el = lane >> 3, n = lane & 7;
    if (el >= count) return;
    int const node = mt.conn[(size_t)(e0 + el) * E::NN + n];
    C8_UNROLL
    for (int d = 0; d < 3; ++d) {
      sh.X[el][n][d] = mt.coords[(size_t)node * 3 + d];
      sh.u[el][n][d] = fa.u[(size_t)node * 3 + d];
      sh.z[el][n][d] = aa.z_u[(size_t)node * 3 + d];
    }
    sh.p[el][n] = fa.p[node];
    sh.z[el][n][3] = aa.z_p[node];
  });
  ex.sync();
  ex.each([&](int lane) {
    int const el = lane >> 3, pt = lane & 7;
    if (el >= count || pt != 0) return;
    sh.h[el] = group_elem_size<E>(sh, el);
  });
  ex.sync();
  ex.each([&](int lane) {
    int const el = lane >> 3, pt = lane & 7;
    // f = -(dC/dx_prev)^T phi = 0: the wavefront's elements are consecutive, their f one block of count * NP0 * NDOF doubles
    double* const fb = aa.f + (size_t)e0 * E::NP0 * E::NDOF;
    C8_UNROLL
    for (int k = 0; k < E::NP0 * E::NDOF / 8; ++k) {
      int const idx = k * 64 + lane;
      if (idx < count * E::NP0 * E::NDOF) fb[idx] = 0.;
    }
    if (el >= count) return;
    int const e = e0 + el;
    int const es = mt.elem_set ? mt.elem_set[e] : 0;
    double const* const prm = mt.params + (size_t)es * Model::NPARAMS;
    PointState<double> gq;
    double ZG[13];
    group_point_state<E, false>(sh, el, pt, gq, ZG);
    double const q[WQ] = {gq.grad_u.xx, gq.grad_u.xy, gq.grad_u.xz, gq.grad_u.yx, gq.grad_u.yy, gq.grad_u.yz, gq.grad_u.zx, gq.grad_u.zy,
                          gq.grad_u.zz, gq.p, gq.grad_p[0], gq.grad_p[1], gq.grad_p[2], gq.u[0], gq.u[1], gq.u[2]};
    size_t const qp = (size_t)e * E::NP0 + pt;
    double xi_old[NL], g_in[NL], phi[NL], g_out[NL];
    C8_UNROLL
    for (int j = 0; j < NL; ++j) { xi_old[j] = fa.xi_prev[qp * NL + j]; g_in[j] = aa.g[qp * NL + j]; }
    typename Model::ClosedForm cf;
    Model::closed_form(prm, q, xi_old, ms.abs_tol, sh.h[el], ms.stab_mult, cf, true);
    Model::closed_form_local_adjoint(prm, cf.t, sh.wdv[el][pt], ZG, g_in, phi, g_out);
    C8_UNROLL
    for (int j = 0; j < NL; ++j) { aa.phi[qp * NL + j] = phi[j]; aa.g[qp * NL + j] = g_out[j]; }
  });
  ex.sync();
}

// =====================================================================================
// K6 for hex8, eight elements per wavefront: eval_qoi (evaluations.cpp:662-756) and the load sum of preprocess_qoi
// (:262-347): lane (element, point) evaluates the objective integrand once; one add per wavefront at the end.
// =====================================================================================
template <template <class> class ModelT> struct QoiWaveLane {
  ModelT<double> m;
  PointState<double> g;
  double acc;
};

template <class E, template <class> class ModelT, class QoI, class EX>
C8_HD void qoi_wave8(EX& ex, GradWaveShared<E>& sh, MeshTables const& mt, FieldArgs const& fa, QoiArgs const& qa, int e0, int count) {
  using Model = ModelT<double>;
  constexpr int NL = Model::NLOC;
  constexpr bool PREV = Model::FINITE_DEF;
  ex.each([&](int lane) {
    int const el = lane >> 3, n = lane & 7;
    if (el >= count) return;
    int const node = mt.conn[(size_t)(e0 + el) * E::NN + n];
    C8_UNROLL
    for (int d = 0; d < 3; ++d) {
      sh.X[el][n][d] = mt.coords[(size_t)node * 3 + d];
      sh.u[el][n][d] = fa.u[(size_t)node * 3 + d];
      if (PREV) sh.u_prev[el][n][d] = fa.u_prev ? fa.u_prev[(size_t)node * 3 + d] : 0.;
    }
    sh.p[el][n] = fa.p[node];
  });
  ex.sync();
  ex.each([&](int lane) {
    int const el = lane >> 3, pt = lane & 7;
    if (el >= count) return;
    auto& r = ex.lane(lane);
    int const e = e0 + el;
    group_point_state<E, PREV>(sh, el, pt, r.g, nullptr);
    int const es = mt.elem_set ? mt.elem_set[e] : 0;
    C8_UNROLL
    for (int q = 0; q < Model::NPARAMS; ++q) r.m.params[q] = mt.params[es * Model::NPARAMS + q];
    size_t const qp = (size_t)e * E::NP0 + pt;
    C8_UNROLL
    for (int j = 0; j < NL; ++j) {
      r.m.xi[j] = fa.xi ? fa.xi[qp * NL + j] : 0.;
      r.m.xi_prev[j] = fa.xi_prev ? fa.xi_prev[qp * NL + j] : 0.;
    }
    r.acc += QoI::evaluate(r.g, r.m, sh.wdv[el][pt], qa, qp);
  });
  ex.sync();
}
template <class EX> C8_HD void qoi_wave8_flush(EX& ex, double* red, double* out) {
  ex.each([&](int lane) { red[lane] = ex.lane(lane).acc; ex.lane(lane).acc = 0.; });
  ex.sync();
  ex.each([&](int lane) {
    if (lane == 0) {
      double t = 0.;
      for (int k = 0; k < 64; ++k) t += red[k];
      ex.add(out, t, 1);
    }
  });
  ex.sync();
}

}  // namespace c8
