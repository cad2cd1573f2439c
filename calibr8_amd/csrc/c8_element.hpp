// c8_element.hpp -- element kit: shape functions, quadrature, DOF order.
//
// Restates what the reference takes from apf (weight.cpp:9-12 getBF/getGradBF,
// evaluations.cpp:82-85 getIntPoint/getIntWeight/getDV) for tet4 -- the element
// the reference runs (disc.cpp:165) -- and extends it to hex8 (BASELINE.json).
// Element DOF order is residual-major, node-minor (global_residual.cpp:21-23):
// slot = offset[i] + node*neq[i] + eq with residual 0 = u (DIM eqs), 1 = p (1 eq).
// tri3 is the reference's 2-D element (disc.cpp:165): DIM = 2; nodal coordinates stay [n][3] (z = 0), point
// quantities stay 3 x 3 containers whose out-of-plane entries are zero.
#pragma once

#include "c8_math.hpp"

namespace c8 {

enum { C8_TRI3 = 3, C8_TET4 = 4, C8_HEX8 = 8 };

template <int ET> struct Elem;

template <> struct Elem<C8_HEX8> {
  static constexpr int TYPE = C8_HEX8;
  static constexpr int NRES = 2;       // global residuals of `mechanics`: u and p; two ip sets
  static constexpr int NSETS = 2;
  static constexpr int DIM = 3;
  static constexpr int NN = 8;         // nodes
  static constexpr int NDOF = 32;      // 3*NN + NN
  static constexpr int NP0 = 8;        // coupled points (ip set 0): 2x2x2 Gauss
  static constexpr int NP1 = 8;        // pressure points (ip set 1): 2x2x2 Gauss
  static constexpr bool SAME_POINTS = true;
  static constexpr int NEDGES = 12;
  C8_HD static void point(int /*ip_set*/, int pt, double* xi, double& w) {
    double const g = 0.5773502691896257645;
    xi[0] = (pt & 1) ? g : -g;
    xi[1] = (pt & 2) ? g : -g;
    xi[2] = (pt & 4) ? g : -g;
    w = 1.;
  }
  C8_HD static void sign(int n, double& sx, double& sy, double& sz) {
    sx = ((n ^ (n >> 1)) & 1) ? 1. : -1.;
    sy = ((n >> 1) & 1) ? 1. : -1.;
    sz = ((n >> 2) & 1) ? 1. : -1.;
  }
  C8_HD static double N(int n, double const* xi) {
    double sx, sy, sz;
    sign(n, sx, sy, sz);
    return 0.125 * (1. + sx * xi[0]) * (1. + sy * xi[1]) * (1. + sz * xi[2]);
  }
  C8_HD static void dNdxi(int n, double const* xi, double* g) {
    double sx, sy, sz;
    sign(n, sx, sy, sz);
    double const a = 1. + sx * xi[0], b = 1. + sy * xi[1], c = 1. + sz * xi[2];
    g[0] = 0.125 * sx * b * c;
    g[1] = 0.125 * sy * a * c;
    g[2] = 0.125 * sz * a * b;
  }
  C8_HD static void edge(int e, int& a, int& b) {
    // (0,1)(1,2)(2,3)(3,0) (4,5)(5,6)(6,7)(7,4) (0,4)(1,5)(2,6)(3,7)
    if (e < 4) { a = e; b = (e + 1) & 3; }
    else if (e < 8) { a = e; b = 4 + ((e + 1) & 3); }
    else { a = e - 8; b = e - 4; }
  }
};

template <> struct Elem<C8_TET4> {
  static constexpr int TYPE = C8_TET4;
  static constexpr int NRES = 2;       // global residuals of `mechanics`: u and p; two ip sets
  static constexpr int NSETS = 2;
  static constexpr int DIM = 3;
  static constexpr int NN = 4;
  static constexpr int NDOF = 16;
  static constexpr int NP0 = 1;        // order 1 (mechanics.cpp:45)
  static constexpr int NP1 = 4;        // order 2 (mechanics.cpp:46)
  static constexpr bool SAME_POINTS = false;
  static constexpr int NEDGES = 6;
  C8_HD static void point(int ip_set, int pt, double* xi, double& w) {
    if (ip_set == 0) {
      xi[0] = xi[1] = xi[2] = 0.25;
      w = 1. / 6.;
    } else {
      double const a = 0.138196601125011, b = 0.585410196624969;
      xi[0] = (pt == 1) ? b : a;
      xi[1] = (pt == 2) ? b : a;
      xi[2] = (pt == 3) ? b : a;
      w = 1. / 24.;
    }
  }
  C8_HD static double N(int n, double const* xi) {
    return n == 0 ? 1. - xi[0] - xi[1] - xi[2] : (n == 1 ? xi[0] : (n == 2 ? xi[1] : xi[2]));
  }
  C8_HD static void dNdxi(int n, double const*, double* g) {
    g[0] = (n == 0) ? -1. : (n == 1 ? 1. : 0.);
    g[1] = (n == 0) ? -1. : (n == 2 ? 1. : 0.);
    g[2] = (n == 0) ? -1. : (n == 3 ? 1. : 0.);
  }
  C8_HD static void edge(int e, int& a, int& b) {
    // (0,1)(1,2)(2,0)(0,3)(1,3)(2,3)
    if (e < 3) { a = e; b = (e + 1) % 3; }
    else { a = e - 3; b = 3; }
  }
};

template <> struct Elem<C8_TRI3> {
  static constexpr int TYPE = C8_TRI3;
  static constexpr int NRES = 2;       // global residuals of `mechanics`: u and p; two ip sets
  static constexpr int NSETS = 2;
  static constexpr int DIM = 2;
  static constexpr int NN = 3;
  static constexpr int NDOF = 9;       // 2*NN + NN
  static constexpr int NP0 = 1;        // order 1 (mechanics.cpp:45): centroid
  static constexpr int NP1 = 3;        // order 2 (mechanics.cpp:46): three interior points
  static constexpr bool SAME_POINTS = false;
  static constexpr int NEDGES = 3;
  C8_HD static void point(int ip_set, int pt, double* xi, double& w) {
    xi[2] = 0.;
    if (ip_set == 0) {
      xi[0] = xi[1] = 1. / 3.;
      w = 0.5;
    } else {
      double const a = 1. / 6., b = 2. / 3.;
      xi[0] = (pt == 0) ? b : a;
      xi[1] = (pt == 1) ? b : a;
      w = 1. / 6.;
    }
  }
  C8_HD static double N(int n, double const* xi) { return n == 0 ? 1. - xi[0] - xi[1] : (n == 1 ? xi[0] : xi[1]); }
  C8_HD static void dNdxi(int n, double const*, double* g) {
    g[0] = (n == 0) ? -1. : (n == 1 ? 1. : 0.);
    g[1] = (n == 0) ? -1. : (n == 2 ? 1. : 0.);
    g[2] = 0.;
  }
  C8_HD static void edge(int e, int& a, int& b) { a = e; b = (e + 1) % 3; }  // (0,1)(1,2)(2,0)
};

// tri3 under `mechanics_plane_stress` (mechanics_plane_stress.cpp:24-38): ONE global residual (u, 2 equations per node),
// one ip set (order 1); element DOFs are the six displacement slots, so a wavefront carries ten lane groups
struct Tri3PlaneStress : Elem<C8_TRI3> {
  static constexpr int NRES = 1;
  static constexpr int NSETS = 1;
  static constexpr int NDOF = 6;
  static constexpr int NP1 = 0;
};

// which residual / node / equation an element DOF slot addresses (dx_idx inverse)
template <class E> C8_HD void slot_to_dof(int k, int& i, int& n, int& eq) {
  if (k < E::DIM * E::NN) { i = 0; n = k / E::DIM; eq = k - E::DIM * n; }
  else { i = 1; n = k - E::DIM * E::NN; eq = 0; }
}

}  // namespace c8
