// c8_assemble_node.hpp -- K1 for hex8 models with a closed form: ONE WAVEFRONT PER NODE, no element stage.
//
// eval_forward_jacobian (evaluations.cpp:12-154) assembles element by element and scatters each element matrix into
// the rows of its eight nodes (global_residual.cpp:556-586).  The staged kernels of c8_assemble_wave.hpp keep that
// shape -- element matrices go to an element-major stage (8.4 KB per hex8 element), a second kernel sums the rows of
// each node -- and move six times the bytes the assembly needs.  Here the loop is turned inside out: a wavefront OWNS
// the four CSR rows (u_0, u_1, u_2, p) of one node, forms them from the node's (up to eight) elements and writes each
// row once.  No stage, no atomics on global memory, every sum in a fixed order: bitwise reproducible.
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
//   phase C  blocks added into the node's row accumulator in LDS, every entry summed in ascending element order
//            (acc[position of the column node in the node's graph row][16])
//   phase D  the finished rows added to (or assigned to) the four CSR blocks, the residual entries to b
//
// Requires an element with 8 nodes, 8 coupled points and identical point sets for both ip sets (hex8), the cached
// shape tables (c8_set_shape_cache) and xi != xi_prev (an element's previous state is read by eight wavefronts).
#pragma once

#include "c8_assemble_wave.hpp"

// Diagnostic build only (-DC8_STAMPS, tools/stamp_node.py): s_memtime stamps at the phase boundaries of sampled nodes.
#ifdef C8_STAMPS
#define C8_NSTAMP(i) ex.stamp_node(ga.stamps, node, i)
#else
#define C8_NSTAMP(i)
#endif

namespace c8 {

template <class M, class = void> struct has_closed_form_rows : std::false_type {};
template <class M> struct has_closed_form_rows<M, std::enable_if_t<M::HAS_CLOSED_FORM_ROWS>> : std::true_type {};

// MANY = false (every node of the mesh has at most eight elements): the row accumulator takes the place of the point
// records once phase B has read them; MANY = true: a node's elements go through phases A-C eight at a time and the
// accumulator has its own storage.
template <class E, class ModelD, int MAXDEG, bool MANY> struct NodeShared {
  static constexpr int NR = ModelD::NROW;
  // record + the point's four residual shares.  Even (16-byte LDS reads) and such that the records of two elements lie
  // 32 banks apart modulo the 64 banks (8 points * LDR * 8 bytes = 128 modulo 256): the two element groups of a
  // 16-lane LDS access then never meet in a bank
  static constexpr int LDR = NR + 4;
  static_assert(LDR % 2 == 0 && (E::NP0 * LDR * 8) % 256 == 128, "record stride: 16-byte aligned, element groups 32 banks apart");
  static constexpr int LDA = 18;            // accumulator row: 16 entries + padding, 16-byte aligned
  static constexpr int NREC = 8 * E::NP0 * LDR, NACC = MAXDEG * LDA;
  alignas(16) double buf[MANY ? NREC + NACC : (NREC > NACC ? NREC : NACC)];
  alignas(16) double nodal[8][E::NN][4];    // (u_0, u_1, u_2, p) of the nodes of the eight elements, loaded one node per lane
  double el[8][2];                          // per element: the model's per-element scalars (closed_form_row)
  double bsum[4];
  C8_HD double* rec(int s, int pt) { return buf + (s * E::NP0 + pt) * LDR; }
  C8_HD double* acc(int pos) { return buf + (MANY ? NREC : 0) + pos * LDA; }
};

#ifndef C8_TUNE_NODE_HAHEAD
#define C8_TUNE_NODE_HAHEAD 3
#endif

template <int MAXDEG> struct NodeLane {
  static constexpr int N00 = (9 * MAXDEG + 63) / 64, N01 = (3 * MAXDEG + 63) / 64;
  double J[16];
  double a00[N00], a01[N01], a10[N01], a11;  // current values of this lane's CSR entries
  double bold;                               // ... and of its residual entry (lanes 0..3)
  double rs;
  int e, a, pos;
  bool valid;
};

// shape values of the reference hex8 at its Gauss points: E::N(n, xi) operation for operation, 0.125 (1 + sx xi_0)
// (1 + sy xi_1) (1 + sz xi_2), with either the point (of_point) or the node (of_node) fixed per lane and the other index a
// compile-time constant or a lane value; signs are selected as values (no indexed tables: they would live in scratch)
template <class E> struct HexShape {
  double c[3];  // of_point: the point's coordinates; of_node: the node's signs
  C8_HD static HexShape of_point(int pt) {
    double w;
    HexShape h;
    E::point(0, pt, h.c, w);
    return h;
  }
  C8_HD double at_node(int n) const {
    double const sx = ((n ^ (n >> 1)) & 1) ? 1. : -1., sy = ((n >> 1) & 1) ? 1. : -1., sz = ((n >> 2) & 1) ? 1. : -1.;  // E::sign
    return 0.125 * (1. + sx * c[0]) * (1. + sy * c[1]) * (1. + sz * c[2]);
  }
  C8_HD static HexShape of_node(int n) {
    HexShape h;
    E::sign(n, h.c[0], h.c[1], h.c[2]);
    return h;
  }
  C8_HD double at_point(int pt) const {
    double xi[3], w;
    E::point(0, pt, xi, w);
    return 0.125 * (1. + c[0] * xi[0]) * (1. + c[1] * xi[1]) * (1. + c[2] * xi[2]);
  }
};

// ADJ = false: eval_forward_jacobian (evaluations.cpp:12-154): A += dR/dx, b += R, xi = the converged local state.
// ADJ = true:  eval_adjoint_jacobian (evaluations.cpp:349-526) for the objective "average displacement": A += (dR/dx)^T at
//              the stored state, b += -dJ/dx + f + (dxi/dx)^T g; g is left as it is (dJ/dxi = 0 for this objective) and
//              no state is written.  The transposed blocks come from the same code with another record
//              (Model::closed_form_row<true>), (dxi/dx)^T g from Model::closed_form_adjoint.
template <class E, template <class> class ModelT, int MAXDEG, bool MANY, bool ADJ = false, class EX>
C8_HD void node_rows_closed(EX& ex, NodeShared<E, ModelT<Dual>, MAXDEG, MANY>& sh, MeshTables const& mt, ModelSettings const& ms,
                            FieldArgs const& fa, GatherArgs const& ga, int node, AdjointArgs const& aa = AdjointArgs{}) {
  using Model = ModelT<Dual>;
  using SH = NodeShared<E, Model, MAXDEG, MANY>;
  using NL_ = NodeLane<MAXDEG>;
  constexpr int NL = Model::NLOC, NR = SH::NR;
  static_assert(E::NN == 8 && E::NP0 == 8 && E::SAME_POINTS, "row-per-node kernel: hex8-like element");
  static_assert(!Mechanics::USES_U && !Model::FINITE_DEF, "row-per-node kernel: small-strain weak form without u terms");
  C8_NSTAMP(6);
  int const nptr = ga.nodeptr[node], deg = ga.nodeptr[node + 1] - nptr;
  int const e0 = ga.nodeelem_ptr[node], e1 = ga.nodeelem_ptr[node + 1];
  size_t const np = (size_t)nptr;
  int const n3 = 3 * deg;
  if (e0 == e1) return;  // a node without elements (phantom columns of a mesh part): its rows are not touched
  C8_NSTAMP(0);
  auto zero_acc = [&](int lane) {
    C8_UNROLL
    for (int it = 0; it < (MAXDEG * SH::LDA + 63) / 64; ++it) {
      int const q = lane + 64 * it;
      if (q < deg * SH::LDA) sh.acc(0)[q] = 0.;
    }
    if (lane < 4) sh.bsum[lane] = 0.;
  };
  if (MANY) {
    ex.each([&](int lane) { zero_acc(lane); });
    ex.sync();
  }
  for (int c0 = e0; c0 < e1; c0 += 8) {  // eight elements of the node at a time (MANY = false: exactly one round)
    int const ne = (e1 - c0 < 8) ? e1 - c0 : 8;
    // ---- phase A: nodal values, lane = (element s, node m): every lane fetches one node of one element, the eight lanes
    //      of an element then read all eight from LDS (a lane per point fetching its own copies costs eight times the
    //      loads and their registers) ------------------------------------------------------------------------------------
    ex.each([&](int lane) {
      auto& r = ex.lane(lane);
      int const s = lane >> 3, m = lane & 7;
      r.valid = s < ne;
      r.rs = 0.;
      if (!r.valid) return;
      int const packed = ga.nodeelem[c0 + s];
      int const e = packed >> 3, a = packed & 7;
      r.e = e;
      r.a = a;
      // phase B's operand of this lane (column node m): position of m in the node's graph row
      r.pos = ga.pos[((size_t)e * E::NN + m) * E::NN + a];
      int const nd = mt.conn[(size_t)e * E::NN + m];
      double* const nv = sh.nodal[s][m];
      nv[0] = fa.u[(size_t)nd * 3 + 0]; nv[1] = fa.u[(size_t)nd * 3 + 1]; nv[2] = fa.u[(size_t)nd * 3 + 2];
      nv[3] = fa.p[nd];
    });
    ex.sync();
    // ---- phase A, continued: lane = (element s, point) ---------------------------------------------------------------
    ex.each([&](int lane_) {
      auto& r = ex.lane(lane_);
      int lane = lane_;
      C8_PIN(lane);  // the shape values below are formed here, not at the top of the kernel and kept in registers
      int const s = lane >> 3, pt = lane & 7;
      if (!r.valid) return;
      int const e = r.e, a = r.a;
      double const* const t = mt.shape + (size_t)e * SHAPE_STRIDE;
      HexShape<E> const Np = HexShape<E>::of_point(pt);
      // adjoint assembly: this point's history entries (f at the node's four rows, g), fetched under the interpolation
      double fh4[4] = {0., 0., 0., 0.}, gx[NL];
      C8_UNROLL
      for (int j = 0; j < NL; ++j) gx[j] = 0.;
      if constexpr (ADJ) {
        double const* const fh = aa.f + ((size_t)e * E::NP0 + pt) * E::NDOF;
        fh4[0] = fh[3 * a + 0]; fh4[1] = fh[3 * a + 1]; fh4[2] = fh[3 * a + 2]; fh4[3] = fh[3 * E::NN + a];
        C8_UNROLL
        for (int j = 0; j < NL; ++j) gx[j] = aa.g[((size_t)e * E::NP0 + pt) * NL + j];
      }
      // interpolation (global_residual.cpp:289-332).  The table is skewed (shape_dn_offset): at step i the element's eight
      // lanes read row i, this lane the entry of node m = (i - pt) mod 8 -- the sum over the nodes starts at node -pt mod 8
      // and wraps; the row node's own entry (g below) is kept when it comes by.  Four nodes' operands in flight at a time
      double q[WQ];
      C8_UNROLL
      for (int c = 0; c < WQ; ++c) q[c] = 0.;
#ifndef C8_TUNE_NODE_GSEL
#define C8_TUNE_NODE_GSEL 0
#endif
      double g[3] = {0., 0., 0.};
      C8_UNROLL
      for (int i = 0; i < E::NN; ++i) {
        int const m = (i - pt) & 7;
        double const* const row = t + i * 24;
        double const d0 = row[m * 2 + 0], d1 = row[m * 2 + 1], d2 = row[16 + m];
        if (C8_TUNE_NODE_GSEL) {
          bool const own = m == a;
          g[0] = own ? d0 : g[0]; g[1] = own ? d1 : g[1]; g[2] = own ? d2 : g[2];
        }
        double const Nm = Np.at_node(m);
        double const* const nv = sh.nodal[s][m];
        double const u0 = nv[0], u1 = nv[1], u2 = nv[2], pm = nv[3];
        q[0] += u0 * d0; q[1] += u0 * d1; q[2] += u0 * d2;
        q[3] += u1 * d0; q[4] += u1 * d1; q[5] += u1 * d2;
        q[6] += u2 * d0; q[7] += u2 * d1; q[8] += u2 * d2;
        q[9] += pm * Nm;
        q[10] += pm * d0; q[11] += pm * d1; q[12] += pm * d2;
#ifndef C8_TUNE_NODE_AGROUP
#define C8_TUNE_NODE_AGROUP 8
#endif
#ifndef C8_TUNE_NODE_AGROUP_ADJ
#define C8_TUNE_NODE_AGROUP_ADJ 2
#endif
        // nodes whose operands are in flight together: as measured (forward 8: 4.51 ms, 4: 4.70, 2: 4.57; adjoint 2: 5.39, 4: 5.55, 8: 5.66)
        if ((i + 1) % (ADJ ? C8_TUNE_NODE_AGROUP_ADJ : C8_TUNE_NODE_AGROUP) == 0) {
          C8_UNROLL
          for (int c = 0; c < 13; ++c) C8_PIN(q[c]);
          C8_SCHED_FENCE();
        }
      }
      C8_NSTAMP(7);
      size_t const q0 = ((size_t)e * E::NP0 + pt) * NL;
      double xi_old[NL];
      C8_UNROLL
      for (int j = 0; j < NL; ++j) xi_old[j] = fa.xi_prev[q0 + j];
      int const es = mt.elem_set ? mt.elem_set[e] : 0;
      typename Model::ClosedForm cf;
      Model::closed_form(mt.params + (size_t)es * Model::NPARAMS, q, xi_old, ms.abs_tol, t[SHAPE_H], ms.stab_mult, cf, true);
      if (!ADJ && a == 0) {  // local->scatter (local_residual.cpp:624-631): once per element, by the wavefront of its first node
        C8_UNROLL
        for (int j = 0; j < NL; ++j) fa.xi[q0 + j] = cf.xi[j];
      }
      C8_NSTAMP(8);
      double const w = t[SHAPE_WDV + pt];
      if (!C8_TUNE_NODE_GSEL) { g[0] = t[shape_dn_offset(pt, a, 0)]; g[1] = t[shape_dn_offset(pt, a, 1)]; g[2] = t[shape_dn_offset(pt, a, 2)]; }
      double const Na = Np.at_node(a);
      double* const rc = sh.rec(s, pt);
      double el[2];
      Model::template closed_form_row<ADJ>(cf.t, w, g, Na, rc, el);
      if (pt == 0) { sh.el[s][0] = el[0]; sh.el[s][1] = el[1]; }
      if constexpr (!ADJ) {
        // the point's share of R_(node,.): fluxes contracted with the row node's shape entries
        C8_UNROLL
        for (int i = 0; i < 3; ++i) rc[NR + i] = w * (cf.F[3 * i] * g[0] + cf.F[3 * i + 1] * g[1] + cf.F[3 * i + 2] * g[2]);
        rc[NR + 3] = w * (cf.F[9] * Na + cf.F[10] * g[0] + cf.F[11] * g[1] + cf.F[12] * g[2]);
      } else {
        // the point's share of the adjoint right-hand side at the node's rows (:486-487): [-dJ/dq + (dxi/dq)^T g] dq/dx + f
        // with dJ/dq = c_avg w / ndims on the displacement itself (avg_disp.cpp:16-33) and (dxi/dq)^T g = Re on grad u
        double Re[6];
        double dq[10];  // dJ/dq of the objective's load term at fixed xi (grad u 0..8, p 9); zero for "average displacement"
        C8_UNROLL
        for (int k = 0; k < 10; ++k) dq[k] = 0.;
        if (aa.qoi.c_load != 0.) {  // calibration objective (uniform over the launch): g -= dJ/dxi first (:474-481); the
          double dxi[NL];           // updated g is stored by node_rows_update_g once every wavefront has read the old one
          Model::closed_form_load_term(mt.params + (size_t)es * Model::NPARAMS, aa.qoi.c_load * w, aa.qoi.comp,
                                       aa.qoi.S + ((size_t)e * E::NP0 + pt) * 3, dxi, dq);
          C8_UNROLL
          for (int j = 0; j < NL; ++j) gx[j] -= dxi[j];
        }
        Model::closed_form_adjoint(mt.params + (size_t)es * Model::NPARAMS, cf.t, gx, Re);
        double const dJ = aa.qoi.c_avg * w / aa.qoi.ndims * Na;
        rc[NR + 0] = (Re[0] - dq[0]) * g[0] + (Re[1] - dq[1]) * g[1] + (Re[2] - dq[2]) * g[2] - dJ + fh4[0];
        rc[NR + 1] = (Re[1] - dq[3]) * g[0] + (Re[3] - dq[4]) * g[1] + (Re[4] - dq[5]) * g[2] - dJ + fh4[1];
        rc[NR + 2] = (Re[2] - dq[6]) * g[0] + (Re[4] - dq[7]) * g[1] + (Re[5] - dq[8]) * g[2] - dJ + fh4[2];
        rc[NR + 3] = -dq[9] * Na + fh4[3];
      }
    });
    ex.sync();
    C8_NSTAMP(1);
    // ---- phase B: lane = (element s, column node m) ----------------------------------------------------------------
    ex.each([&](int lane_) {
      auto& r = ex.lane(lane_);
      int lane = lane_;
      C8_PIN(lane);
      C8_UNROLL
      for (int j = 0; j < 16; ++j) r.J[j] = 0.;
      if (!r.valid) return;
      int const s = lane >> 3, m = lane & 7;
      double const* const t = mt.shape + (size_t)r.e * SHAPE_STRIDE;
      HexShape<E> const Nn = HexShape<E>::of_node(m);
      double const el[2] = {sh.el[s][0], sh.el[s][1]};
      // dN_m/dx of this lane's column node: at step i the element's eight lanes read row i of the skewed table, this lane
      // the entry of point (i - m) mod 8; C8_TUNE_NODE_HAHEAD steps ahead of the arithmetic (the element's table is in L1 /
      // L2 by now)
      constexpr int AH = C8_TUNE_NODE_HAHEAD;
      double hq[AH + 1][3];
      C8_UNROLL
      for (int k = 0; k < AH; ++k) { hq[k][0] = t[k * 24 + m * 2 + 0]; hq[k][1] = t[k * 24 + m * 2 + 1]; hq[k][2] = t[k * 24 + 16 + m]; }
      C8_UNROLL
      for (int i = 0; i < E::NP0; ++i) {
        int const pt = (i - m) & 7;
        hq[AH][0] = hq[AH][1] = hq[AH][2] = 0.;
        if (i + AH < E::NP0) {
          hq[AH][0] = t[(i + AH) * 24 + m * 2 + 0];
          hq[AH][1] = t[(i + AH) * 24 + m * 2 + 1];
          hq[AH][2] = t[(i + AH) * 24 + 16 + m];
        }
        double const hc[3] = {hq[0][0], hq[0][1], hq[0][2]};
        Model::closed_form_block(sh.rec(s, pt), el, hc, Nn.at_point(pt), r.J);
        // one point's record in registers at a time: without the lines below the compiler fetches the records of all
        // eight points first (over a hundred doubles) and spills
        C8_UNROLL
        for (int j = 0; j < 16; ++j) C8_PIN(r.J[j]);
        C8_SCHED_FENCE();
        C8_UNROLL
        for (int k = 0; k < AH; ++k) { hq[k][0] = hq[k + 1][0]; hq[k][1] = hq[k + 1][1]; hq[k][2] = hq[k + 1][2]; }
      }
      if (m < 4) {
        double v = 0.;
        C8_UNROLL
        for (int pt = 0; pt < E::NP0; ++pt) v += sh.rec(s, pt)[NR + m];
        r.rs = v;
      }
    });
    C8_NSTAMP(2);
    if (c0 == e0) {
      // the CSR entries this lane updates in phase D, fetched while phase C runs (assign mode: nothing to read, the rows start from
      // zero -- uniform over the launch).  Not earlier: the counter of outstanding loads is in order, so the first wait of
      // phase B for a shape entry out of L2 would also wait for these, which come from HBM
      ex.each([&](int lane_) {
        auto& r = ex.lane(lane_);
        int lane = lane_;
        C8_PIN(lane);  // what follows is derived from the lane number here, not at the top of the kernel and kept in registers
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
        r.bold = 0.;
        if (!ga.assign && lane < 4) r.bold = lane < 3 ? ga.b[0][(size_t)node * 3 + lane] : ga.b[1][node];
      });
    }
    if (!MANY) {  // the accumulator takes the records' place
      ex.sync();
      ex.each([&](int lane) { zero_acc(lane); });
    }
    ex.sync();
    C8_NSTAMP(3);
    // ---- phase C: the elements' blocks into the row accumulator (ds_add_f64), the sums into one entry taken in ascending
    //      element order --------------------------------------------------------------------------------------------
#ifndef C8_TUNE_NODE_C_BY_ELEMENT
    // all elements in one pass.  Lanes of one instruction that add to the same entry (a column node shared by several of the
    // node's elements) are served by the LDS unit in ascending lane order, i.e. in ascending element order: the results are
    // bitwise those of the element-by-element form below (measured on jiggled meshes, forward and adjoint, tools/README.md),
    // which took eight times the LDS instructions (K1 4.68 -> 4.46 ms)
    ex.each([&](int lane) {
      auto& r = ex.lane(lane);
      if ((lane >> 3) >= ne) return;
      double* const ac = sh.acc(r.pos);
      C8_UNROLL
      for (int j = 0; j < 16; ++j) ex.lds_add(ac + j, r.J[j]);
      if ((lane & 7) < 4) ex.lds_add(&sh.bsum[lane & 7], r.rs);
    });
#else  // one element after the other: within one element the column nodes are distinct and so are the addresses
    for (int s2 = 0; s2 < ne; ++s2) {
      ex.each([&](int lane) {
        auto& r = ex.lane(lane);
        if ((lane >> 3) != s2) return;
        double* const ac = sh.acc(r.pos);
        C8_UNROLL
        for (int j = 0; j < 16; ++j) ex.lds_add(ac + j, r.J[j]);
        if ((lane & 7) < 4) ex.lds_add(&sh.bsum[lane & 7], r.rs);
      });
    }
#endif
    ex.sync();
    C8_NSTAMP(4);
  }
  // ---- phase D: rows out (the tail of gather_node_rows) -------------------------------------------------------------
  ex.each([&](int lane_) {
    auto& r = ex.lane(lane_);
    int lane = lane_;
    C8_PIN(lane);
    // all fetched values complete HERE, once, outside the conditional stores below: with the waits inside the branches the
    // compiler's count of outstanding memory operations is lost at every join, and it then drains the counter -- the
    // previous store included -- in front of each store (ten serialised write round trips per node)
    C8_UNROLL
    for (int it = 0; it < NL_::N00; ++it) C8_PIN(r.a00[it]);
    C8_UNROLL
    for (int it = 0; it < NL_::N01; ++it) { C8_PIN(r.a01[it]); C8_PIN(r.a10[it]); }
    C8_PIN(r.a11);
    C8_PIN(r.bold);
    C8_UNROLL
    for (int it = 0; it < NL_::N00; ++it) {
      int const j = lane + 64 * it;
      if (j < 9 * deg) {
        int const i = (j >= n3) + (j >= 2 * n3), jj = j - i * n3, pos = jj / 3, col = jj - 3 * pos;
        ga.A[0][0][np * 9 + j] = r.a00[it] + sh.acc(pos)[i * 4 + col];
      }
    }
    C8_UNROLL
    for (int it = 0; it < NL_::N01; ++it) {
      int const j = lane + 64 * it;
      if (j < n3) {
        int const i = (j >= deg) + (j >= 2 * deg), pos = j - i * deg;
        ga.A[0][1][np * 3 + j] = r.a01[it] + sh.acc(pos)[i * 4 + 3];
        int const pos2 = j / 3, col = j - 3 * pos2;
        ga.A[1][0][np * 3 + j] = r.a10[it] + sh.acc(pos2)[3 * 4 + col];
      }
    }
    if (lane < deg) ga.A[1][1][np + lane] = r.a11 + sh.acc(lane)[15];
    if (lane < 3) ga.b[0][(size_t)node * 3 + lane] = r.bold + sh.bsum[lane];
    if (lane == 3) ga.b[1][node] = r.bold + sh.bsum[3];
  });
  ex.sync();
  C8_NSTAMP(5);
}

// The adjoint assembly's update of the local history, g -= dJ/dxi (evaluations.cpp:474-481), for the objective's load term:
// one lane per (element, point), run AFTER the row-per-node launches of the call (every wavefront reads the old g of its
// elements' points; eight wavefronts read each).  The shape-table weight and the model's derivative are those of phase A.
template <class E, template <class> class ModelT>
C8_HD void node_rows_update_g(MeshTables const& mt, AdjointArgs const& aa, size_t qp) {
  using Model = ModelT<Dual>;
  int const e = (int)(qp / E::NP0), pt = (int)(qp % E::NP0);
  int const es = mt.elem_set ? mt.elem_set[e] : 0;
  double const w = mt.shape[(size_t)e * SHAPE_STRIDE + SHAPE_WDV + pt];
  double dxi[Model::NLOC], dq[10];
  Model::closed_form_load_term(mt.params + (size_t)es * Model::NPARAMS, aa.qoi.c_load * w, aa.qoi.comp, aa.qoi.S + qp * 3, dxi, dq);
  C8_UNROLL
  for (int j = 0; j < Model::NLOC; ++j) aa.g[qp * Model::NLOC + j] -= dxi[j];
}

}  // namespace c8
