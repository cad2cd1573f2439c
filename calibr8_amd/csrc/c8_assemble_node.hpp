// c8_assemble_node.hpp -- K1 for hex8 models with a closed form: ONE WAVEFRONT PER NODE, no element stage.
//
// eval_forward_jacobian (evaluations.cpp:12-154) assembles element by element and scatters each element matrix into
// the rows of its eight nodes (global_residual.cpp:556-586).  The staged kernels of c8_assemble_wave.hpp keep that
// shape -- element matrices go to an element-major stage (8.4 KB per hex8 element), a second kernel sums the rows of
// each node -- and move six times the bytes the assembly needs.  Here the loop is turned inside out: a wavefront OWNS
// the four CSR rows (u_0, u_1, u_2, p) of one node, forms them from the node's (up to eight) elements and writes each
// row once.  No stage, no atomics, every sum in a fixed order: bitwise reproducible.
//
// What makes it affordable is the model's closed form (Model::closed_form, Model::closed_form_row / _block): the
// tangent data of a point are a handful of doubles, so the eight nodes of an element each recompute them (on all 64
// lanes at once, one lane per (element, point)) instead of sharing them through memory, and the contraction with
// the shape functions -- the bulk of the arithmetic -- is split over the nodes without redundancy: every 4 x 4
// block of the element matrix is formed exactly once, by the wavefront of its row node.
//
//   phase A  lane = (element s of the node, point)      interpolate the point quantities from the element's nodal
//            values and cached shape tables, closed form (radial return, consistent tangent), the row node's record of
//            the point into LDS, the point's share of the node's residual; the lane of the element's local node 0
//            stores the converged local state
//   phase B  lane = (element s of the node, column node m)   the 4 x 4 block d R_(node,.) / d x_(m,.) summed over the
//            element's eight points: 16 accumulators
//   phase C  element by element, in ascending element order: blocks added into the node's row accumulator in LDS
//            (acc[position of the column node in the node's graph row][16])
//   phase D  the finished rows added to (or assigned to) the four CSR blocks, the residual entries to b
//
// Requires an element with 8 nodes, 8 coupled points and identical point sets for both ip sets (hex8), the cached
// shape tables (c8_set_shape_cache) and xi != xi_prev (an element's previous state is read by eight wavefronts).
#pragma once

#include "c8_assemble_wave.hpp"

namespace c8 {

template <class M, class = void> struct has_closed_form_rows : std::false_type {};
template <class M> struct has_closed_form_rows<M, std::enable_if_t<M::HAS_CLOSED_FORM_ROWS>> : std::true_type {};

template <class E, class ModelD, int MAXDEG> struct NodeShared {
  static constexpr int NR = ModelD::NROW;
  static constexpr int LDR = (NR + 4) | 1;  // record + the point's four residual contributions; odd stride
  static constexpr int LDA = 17;            // one double of padding per accumulator row (see GatherShared)
  double rec[8][E::NP0][LDR];
  double acc[MAXDEG][LDA];
  double Ntab[E::NP0][E::NN];               // shape values of the reference element at the points
  double bsum[4];
};

template <int MAXDEG> struct NodeLane {
  static constexpr int N00 = (9 * MAXDEG + 63) / 64, N01 = (3 * MAXDEG + 63) / 64;
  double J[16];
  double a00[N00], a01[N01], a10[N01], a11;  // current values of this lane's CSR entries
  double rs;
  int e, a, pos;
  bool valid;
};

template <class E, template <class> class ModelT, int MAXDEG, class EX>
C8_HD void node_rows_closed(EX& ex, NodeShared<E, ModelT<Dual>, MAXDEG>& sh, MeshTables const& mt, ModelSettings const& ms,
                            FieldArgs const& fa, GatherArgs const& ga, int node) {
  using Model = ModelT<Dual>;
  using SH = NodeShared<E, Model, MAXDEG>;
  using NL_ = NodeLane<MAXDEG>;
  constexpr int NL = Model::NLOC, NR = SH::NR;
  static_assert(E::NN == 8 && E::NP0 == 8 && E::SAME_POINTS, "row-per-node kernel: hex8-like element");
  static_assert(!Mechanics::USES_U && !Model::FINITE_DEF, "row-per-node kernel: small-strain weak form without u terms");
  int const nptr = ga.nodeptr[node], deg = ga.nodeptr[node + 1] - nptr;
  int const e0 = ga.nodeelem_ptr[node], e1 = ga.nodeelem_ptr[node + 1];
  size_t const np = (size_t)nptr;
  int const n3 = 3 * deg;
  ex.each([&](int lane) {
    auto& r = ex.lane(lane);
    C8_UNROLL
    for (int it = 0; it < (MAXDEG * SH::LDA + 63) / 64; ++it) {
      int const q = lane + 64 * it;
      if (q < deg * SH::LDA) (&sh.acc[0][0])[q] = 0.;
    }
    {
      double xi[3], w;
      E::point(0, lane >> 3, xi, w);
      sh.Ntab[lane >> 3][lane & 7] = E::N(lane & 7, xi);
    }
    if (lane < 4) sh.bsum[lane] = 0.;
  });
  ex.sync();
  for (int c0 = e0; c0 < e1; c0 += 8) {  // eight elements of the node at a time (a node of a hex8 mesh rarely has more)
    int const ne = (e1 - c0 < 8) ? e1 - c0 : 8;
    // ---- phase A: lane = (element s, point) ------------------------------------------------------------------------
    ex.each([&](int lane) {
      auto& r = ex.lane(lane);
      int const s = lane >> 3, pt = lane & 7;
      r.valid = s < ne;
      r.rs = 0.;
      if (!r.valid) return;
      int const packed = ga.nodeelem[c0 + s];
      int const e = packed >> 3, a = packed & 7;
      r.e = e;
      r.a = a;
      double const* const t = mt.shape + (size_t)e * SHAPE_STRIDE;
      int32_t const* const cn = mt.conn + (size_t)e * E::NN;
      // phase B's operand of this lane (column node m = lane & 7): position of m in the node's graph row
      r.pos = ga.pos[((size_t)e * E::NN + pt) * E::NN + a];
      // interpolation (global_residual.cpp:289-332): the same sequential sums over the nodes as interp_ab
      double q[WQ];
      C8_UNROLL
      for (int c = 0; c < WQ; ++c) q[c] = 0.;
      C8_UNROLL
      for (int m = 0; m < E::NN; ++m) {
        int const nd = cn[m];
        double const d0 = t[(pt * E::NN + m) * 3 + 0], d1 = t[(pt * E::NN + m) * 3 + 1], d2 = t[(pt * E::NN + m) * 3 + 2];
        double const Nm = sh.Ntab[pt][m];
        double const u0 = fa.u[(size_t)nd * 3 + 0], u1 = fa.u[(size_t)nd * 3 + 1], u2 = fa.u[(size_t)nd * 3 + 2], pm = fa.p[nd];
        q[0] += u0 * d0; q[1] += u0 * d1; q[2] += u0 * d2;
        q[3] += u1 * d0; q[4] += u1 * d1; q[5] += u1 * d2;
        q[6] += u2 * d0; q[7] += u2 * d1; q[8] += u2 * d2;
        q[9] += pm * Nm;
        q[10] += pm * d0; q[11] += pm * d1; q[12] += pm * d2;
      }
      size_t const q0 = ((size_t)e * E::NP0 + pt) * NL;
      double xi_old[NL];
      C8_UNROLL
      for (int j = 0; j < NL; ++j) xi_old[j] = fa.xi_prev[q0 + j];
      int const es = mt.elem_set ? mt.elem_set[e] : 0;
      typename Model::ClosedForm cf;
      Model::closed_form(mt.params + (size_t)es * Model::NPARAMS, q, xi_old, ms.abs_tol, t[SHAPE_H], ms.stab_mult, cf, true);
      if (a == 0) {  // local->scatter (local_residual.cpp:624-631): once per element, by the wavefront of its first node
        C8_UNROLL
        for (int j = 0; j < NL; ++j) fa.xi[q0 + j] = cf.xi[j];
      }
      double const w = t[SHAPE_WDV + pt];
      double const g[3] = {t[(pt * E::NN + a) * 3 + 0], t[(pt * E::NN + a) * 3 + 1], t[(pt * E::NN + a) * 3 + 2]};
      double const Na = sh.Ntab[pt][a];
      double* const rc = sh.rec[s][pt];
      Model::closed_form_row(cf.t, w, g, Na, rc);
      // the point's share of R_(node,.): fluxes contracted with the row node's shape entries
      C8_UNROLL
      for (int i = 0; i < 3; ++i) rc[NR + i] = w * (cf.F[3 * i] * g[0] + cf.F[3 * i + 1] * g[1] + cf.F[3 * i + 2] * g[2]);
      rc[NR + 3] = w * (cf.F[9] * Na + cf.F[10] * g[0] + cf.F[11] * g[1] + cf.F[12] * g[2]);
    });
    ex.sync();
    // ---- phase B: lane = (element s, column node m) ----------------------------------------------------------------
    ex.each([&](int lane) {
      auto& r = ex.lane(lane);
      C8_UNROLL
      for (int j = 0; j < 16; ++j) r.J[j] = 0.;
      if (!r.valid) return;
      int const s = lane >> 3, m = lane & 7;
      double const* const t = mt.shape + (size_t)r.e * SHAPE_STRIDE;
      double h[E::NP0][3];  // dN_m/dx of this lane's column node at the eight points (the element's table is in L1 / L2 by now)
      C8_UNROLL
      for (int q = 0; q < E::NP0; ++q) {
        h[q][0] = t[(q * E::NN + m) * 3 + 0];
        h[q][1] = t[(q * E::NN + m) * 3 + 1];
        h[q][2] = t[(q * E::NN + m) * 3 + 2];
      }
      C8_UNROLL
      for (int pt = 0; pt < E::NP0; ++pt) {
        Model::closed_form_block(sh.rec[s][pt], h[pt], sh.Ntab[pt][m], r.J);
        // one point's record in registers at a time: without the two lines below the compiler fetches the records of all
        // eight points first (136 doubles) and spills
        C8_UNROLL
        for (int j = 0; j < 16; ++j) C8_PIN(r.J[j]);
        C8_SCHED_FENCE();
      }
      if (m < 4) {
        double v = 0.;
        C8_UNROLL
        for (int pt = 0; pt < E::NP0; ++pt) v += sh.rec[s][pt][NR + m];
        r.rs = v;
      }
    });
    // ---- phase C: the elements' blocks into the row accumulator, one element after the other ------------------------
    for (int s2 = 0; s2 < ne; ++s2) {
      ex.each([&](int lane) {
        auto& r = ex.lane(lane);
        if ((lane >> 3) != s2) return;
        double* const ac = sh.acc[r.pos];
        C8_UNROLL
        for (int j = 0; j < 16; ++j) ac[j] += r.J[j];   // distinct column nodes: distinct addresses within the instruction
        if ((lane & 7) < 4) sh.bsum[lane & 7] += r.rs;
      });
      ex.sync();
    }
  }
  // ---- phase D: rows out (the tail of gather_node_rows) -------------------------------------------------------------
  ex.each([&](int lane) {
    auto& r = ex.lane(lane);
    // the CSR entries this lane updates (assign mode: nothing to read, the rows start from zero -- uniform over the launch)
    C8_UNROLL
    for (int it = 0; it < NL_::N00; ++it) {
      int const j = lane + 64 * it;
      r.a00[it] = 0.;
      if (!ga.assign && j < 9 * deg) r.a00[it] = ga.A[0][0][np * 9 + j];
    }
    C8_UNROLL
    for (int it = 0; it < NL_::N01; ++it) {
      int const j = lane + 64 * it;
      r.a01[it] = r.a10[it] = 0.;
      if (!ga.assign && j < n3) { r.a01[it] = ga.A[0][1][np * 3 + j]; r.a10[it] = ga.A[1][0][np * 3 + j]; }
    }
    r.a11 = 0.;
    if (!ga.assign && lane < deg) r.a11 = ga.A[1][1][np + lane];
    C8_UNROLL
    for (int it = 0; it < NL_::N00; ++it) {
      int const j = lane + 64 * it;
      if (j < 9 * deg) {
        int const i = (j >= n3) + (j >= 2 * n3), jj = j - i * n3, pos = jj / 3, col = jj - 3 * pos;
        ga.A[0][0][np * 9 + j] = r.a00[it] + sh.acc[pos][i * 4 + col];
      }
    }
    C8_UNROLL
    for (int it = 0; it < NL_::N01; ++it) {
      int const j = lane + 64 * it;
      if (j < n3) {
        int const i = (j >= deg) + (j >= 2 * deg), pos = j - i * deg;
        ga.A[0][1][np * 3 + j] = r.a01[it] + sh.acc[pos][i * 4 + 3];
        int const pos2 = j / 3, col = j - 3 * pos2;
        ga.A[1][0][np * 3 + j] = r.a10[it] + sh.acc[pos2][3 * 4 + col];
      }
    }
    if (lane < deg) ga.A[1][1][np + lane] = r.a11 + sh.acc[lane][15];
    if (lane < 3) ga.b[0][(size_t)node * 3 + lane] = (ga.assign ? 0. : ga.b[0][(size_t)node * 3 + lane]) + sh.bsum[lane];
    if (lane == 3) ga.b[1][node] = (ga.assign ? 0. : ga.b[1][node]) + sh.bsum[3];
  });
  ex.sync();
}

}  // namespace c8
