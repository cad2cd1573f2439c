// c8_qoi_host.hpp -- the parts of the Calibration objective (calibration.cpp, 3-D form) that are not device kernels,
// shared by c8_qoi.hip and by the CPU emulator of the tests: the face quadrature, the face term of one face, and the
// set-up tables (which element faces lie on the displacement side set, which element nodes on the load plane).
#pragma once

#include <algorithm>
#include <cmath>
#include <set>
#include <string>
#include <vector>

#include "c8_assemble.hpp"
#include "c8_host.hpp"

namespace c8 {

// element faces in local node ids (this library's numbering of the downward faces; the reference only uses
// the face to find its nodes)
static int const TET_FACES[4][3] = {{0, 1, 2}, {0, 1, 3}, {1, 2, 3}, {0, 2, 3}};
static int const HEX_FACES[6][4] = {{0, 1, 2, 3}, {0, 1, 5, 4}, {1, 2, 6, 5}, {2, 3, 7, 6}, {3, 0, 4, 7}, {4, 5, 6, 7}};

// order-2 rule on a face (calibration.cpp:262-266): tri3 3 points, quad4 2x2 Gauss; Nf = face shape functions,
// wdv = weight * getDV.  Returns the number of points.
C8_HD int face_rule(int nf, double const X[][3], double Nf[][4], double* wdv) {
  if (nf == 3) {
    double a[3], b[3];
    for (int d = 0; d < 3; ++d) { a[d] = X[1][d] - X[0][d]; b[d] = X[2][d] - X[0][d]; }
    double const c0 = a[1] * b[2] - a[2] * b[1], c1 = a[2] * b[0] - a[0] * b[2], c2 = a[0] * b[1] - a[1] * b[0];
    double const dv = sqrt(c0 * c0 + c1 * c1 + c2 * c2);
    double const st[3][2] = {{1. / 6., 1. / 6.}, {2. / 3., 1. / 6.}, {1. / 6., 2. / 3.}};
    for (int q = 0; q < 3; ++q) {
      Nf[q][0] = 1. - st[q][0] - st[q][1]; Nf[q][1] = st[q][0]; Nf[q][2] = st[q][1]; Nf[q][3] = 0.;
      wdv[q] = dv / 6.;
    }
    return 3;
  }
  double const gp = 0.5773502691896257645;
  double const sn[4] = {-1., 1., 1., -1.}, tn[4] = {-1., -1., 1., 1.};
  int q = 0;
  for (int j = 0; j < 2; ++j)
    for (int i = 0; i < 2; ++i, ++q) {
      double const s = (i ? gp : -gp), t = (j ? gp : -gp);
      double ds[3] = {0., 0., 0.}, dt[3] = {0., 0., 0.};
      for (int k = 0; k < 4; ++k) {
        Nf[q][k] = 0.25 * (1. + sn[k] * s) * (1. + tn[k] * t);
        for (int d = 0; d < 3; ++d) {
          ds[d] += 0.25 * sn[k] * (1. + tn[k] * t) * X[k][d];
          dt[d] += 0.25 * tn[k] * (1. + sn[k] * s) * X[k][d];
        }
      }
      double const c0 = ds[1] * dt[2] - ds[2] * dt[1], c1 = ds[2] * dt[0] - ds[0] * dt[2], c2 = ds[0] * dt[1] - ds[1] * dt[0];
      wdv[q] = sqrt(c0 * c0 + c1 * c1 + c2 * c2);
    }
  return 4;
}
// face area by the one-point rule of calibration.cpp:122-126
inline double face_area(int nf, double const X[][3]) {
  if (nf == 3) {
    double Nf[4][4], wdv[4];
    face_rule(3, X, Nf, wdv);
    return 3. * wdv[0];
  }
  double const sn[4] = {-1., 1., 1., -1.}, tn[4] = {-1., -1., 1., 1.};
  double ds[3] = {0., 0., 0.}, dt[3] = {0., 0., 0.};
  for (int k = 0; k < 4; ++k)
    for (int d = 0; d < 3; ++d) { ds[d] += 0.25 * sn[k] * X[k][d]; dt[d] += 0.25 * tn[k] * X[k][d]; }
  double const c0 = ds[1] * dt[2] - ds[2] * dt[1], c1 = ds[2] * dt[0] - ds[0] * dt[2], c2 = ds[0] * dt[1] - ds[1] * dt[0];
  return 4. * sqrt(c0 * c0 + c1 * c1 + c2 * c2);
}

// compute_surface_mismatch (calibration.cpp:225-300) for one face of the side set.  At a face point the element's
// shape functions reduce to the face's own, so the interpolation of the element field at boundaryToElementXi(point)
// is the face interpolation of the nodal values.  Returns the unscaled value 1/2 sum_q sum_d w_d (u_d - u_meas_d)^2
// w dv and its derivative with respect to the nodal displacements of the face.
// With ndims = 2 the "face" is a tri3 ELEMENT of a 2-D mesh and the same integral is compute_disp_mismatch
// (calibration.cpp:163-222): its order-2 rule is the three-point rule of the triangle.
C8_HD double surface_mismatch_face(int nf, int32_t const* fn, double const* coords, double const* u, double const* u_meas,
                                   double const* wt, double grad[4][3], int ndims = 3) {
  double X[4][3], Nf[4][4], wdv[4], du[4][3];
  for (int k = 0; k < nf; ++k)
    for (int d = 0; d < 3; ++d) {
      X[k][d] = coords[(size_t)fn[k] * 3 + d];
      du[k][d] = d < ndims ? u[(size_t)fn[k] * ndims + d] - u_meas[(size_t)fn[k] * ndims + d] : 0.;
      grad[k][d] = 0.;
    }
  int const nq = face_rule(nf, X, Nf, wdv);
  double val = 0.;
  for (int q = 0; q < nq; ++q)
    for (int d = 0; d < ndims; ++d) {
      double diff = 0.;
      for (int k = 0; k < nf; ++k) diff += du[k][d] * Nf[q][k];
      val += 0.5 * wt[d] * diff * diff * wdv[q];
      for (int k = 0; k < nf; ++k) grad[k][d] += wt[d] * diff * Nf[q][k] * wdv[q];
    }
  return val;
}

// S[e][pt][j] = sum over the element's nodes on the load plane of dN_n/dx_j at coupled point pt
template <class E> inline void load_plane_sums(HostMesh const& m, std::vector<unsigned> const& mask, std::vector<double>& S) {
  S.assign((size_t)m.nelems * E::NP0 * 3, 0.);
  GroupShared<E, 1> sh;
  for (int e = 0; e < m.nelems; ++e) {
    if (!mask[e]) continue;
    for (int n = 0; n < E::NN; ++n)
      for (int d = 0; d < 3; ++d) sh.X[n][d] = m.coords[(size_t)m.conn[(size_t)e * E::NN + n] * 3 + d];
    for (int pt = 0; pt < E::NP0; ++pt) {
      shape_entry<E>(sh, 0, pt, 0, E::NN);
      for (int n = 0; n < E::NN; ++n)
        if (mask[e] & (1u << n))
          for (int j = 0; j < 3; ++j) S[((size_t)e * E::NP0 + pt) * 3 + j] += sh.dN[pt][n][j];
    }
  }
}


// Calibration::before_elems (calibration.cpp:55-160) + setup_coord_based_node_mapping (qoi.cpp:160-198)
struct CalibrationTables {
  std::vector<int32_t> faces;   // [n][4] node ids of the element faces on the side set (-1 padding)
  std::vector<unsigned> mask;   // [nelems] bit n: local node n lies on the load plane
  std::vector<double> S;        // [nelems][coupled points][3]
  double area = 0.;
  int nfn = 0;                  // nodes per face
};
inline void calibration_tables(HostMesh const& mesh, int num_faces, int32_t const* side_faces, int coord_idx, double coord_value,
                               double coord_tol, CalibrationTables& t) {
  if (mesh.nn == 3) {
    // 2-D branch (calibration.cpp:76-104): the displacement term is integrated over the elements themselves -- all of
    // them, or (a distance field with a threshold in the reference) the ones listed in side_faces as element ids --,
    // every "face" here is a tri3 element; the area is the sum of the element areas
    t.nfn = 3;
    t.faces.clear();
    t.mask.assign((size_t)mesh.nelems, 0u);
    t.area = 0.;
    std::set<int32_t> listed(side_faces, side_faces + (side_faces ? num_faces : 0));
    for (int e = 0; e < mesh.nelems; ++e) {
      int32_t const* en = &mesh.conn[(size_t)e * 3];
      if (num_faces == 0 || listed.count(e)) {
        double X[4][3];
        for (int k = 0; k < 3; ++k)
          for (int q = 0; q < 3; ++q) X[k][q] = mesh.coords[(size_t)en[k] * 3 + q];
        t.area += face_area(3, X);
        for (int k = 0; k < 4; ++k) t.faces.push_back(k < 3 ? en[k] : -1);
      }
      for (int n = 0; n < 3; ++n)
        if (std::abs(mesh.coords[(size_t)en[n] * 3 + coord_idx] - coord_value) < coord_tol) t.mask[e] |= 1u << n;
    }
    load_plane_sums<Elem<C8_TRI3>>(mesh, t.mask, t.S);
    return;
  }
  int const nn = mesh.nn, nfn = (nn == 4) ? 3 : 4, nfe = (nn == 4) ? 4 : 6;
  t.nfn = nfn;
  std::set<std::vector<int32_t>> side;
  for (int f = 0; f < num_faces; ++f) {
    std::vector<int32_t> key(side_faces + (size_t)f * nfn, side_faces + (size_t)(f + 1) * nfn);
    std::sort(key.begin(), key.end());
    side.insert(key);
  }
  // m_mapping_disp (calibration.cpp:98-135): one face per element, a later downward face overwrites an earlier
  // one, every match adds its area; m_mapping_load (qoi.cpp:160-198)
  t.faces.clear();
  t.mask.assign((size_t)mesh.nelems, 0u);
  t.area = 0.;
  for (int e = 0; e < mesh.nelems; ++e) {
    int32_t const* en = &mesh.conn[(size_t)e * nn];
    int hit = -1;
    for (int dn = 0; dn < nfe; ++dn) {
      int const* loc = (nn == 4) ? TET_FACES[dn] : HEX_FACES[dn];
      std::vector<int32_t> key(nfn);
      for (int k = 0; k < nfn; ++k) key[k] = en[loc[k]];
      std::sort(key.begin(), key.end());
      if (!side.count(key)) continue;
      hit = dn;
      double X[4][3];
      for (int k = 0; k < nfn; ++k)
        for (int q = 0; q < 3; ++q) X[k][q] = mesh.coords[(size_t)en[loc[k]] * 3 + q];
      t.area += face_area(nfn, X);
    }
    if (hit >= 0) {
      int const* loc = (nn == 4) ? TET_FACES[hit] : HEX_FACES[hit];
      for (int k = 0; k < 4; ++k) t.faces.push_back(k < nfn ? en[loc[k]] : -1);
    }
    for (int n = 0; n < nn; ++n)
      if (std::abs(mesh.coords[(size_t)en[n] * 3 + coord_idx] - coord_value) < coord_tol) t.mask[e] |= 1u << n;
  }
  if (nn == 4) load_plane_sums<Elem<C8_TET4>>(mesh, t.mask, t.S);
  else load_plane_sums<Elem<C8_HEX8>>(mesh, t.mask, t.S);
}

}  // namespace c8
