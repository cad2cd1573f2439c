// c8_qoi.hip -- the objective (QoI) beside the adjoint hot path: "average displacement"
// (avg_disp.cpp) and "calibration" (calibration.cpp, 3-D form: surface displacement mismatch on a side set plus
// the mismatch of the reaction load on a coordinate plane).
//
// The point-wise parts (average displacement; the load term, which re-enters the weak form) are the run-time
// integrand PointQoi of c8_models.hpp inside the adjoint kernels.  This file holds what lives outside them:
// the face integral of the displacement mismatch and its derivative (quadratic in the nodal values, no AD),
// preprocess_qoi (evaluations.cpp:262-347 -> total load), postprocess (calibration.cpp:374-381) and the set-up
// (calibration.cpp:13-50, :55-160; qoi.cpp:160-198).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <set>
#include <string>
#include <vector>

#include "../../include/c8.h"
#include "c8_api_internal.hpp"

using namespace c8;

#define QH(call)                                                                                   \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) return c8_fail(C8_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

namespace {

// element faces in local node ids (this library's numbering of the downward faces; the reference only uses
// the face to find its nodes)
int const TET_FACES[4][3] = {{0, 1, 2}, {0, 1, 3}, {1, 2, 3}, {0, 2, 3}};
int const HEX_FACES[6][4] = {{0, 1, 2, 3}, {0, 1, 5, 4}, {1, 2, 6, 5}, {2, 3, 7, 6}, {3, 0, 4, 7}, {4, 5, 6, 7}};

// order-2 rule on a face (calibration.cpp:262-266): tri3 3 points, quad4 2x2 Gauss; Nf = face shape functions,
// wdv = weight * getDV.  Returns the number of points.
__host__ __device__ inline int face_rule(int nf, double const X[][3], double Nf[][4], double* wdv) {
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
double face_area(int nf, double const X[][3]) {
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
  return 4. * std::sqrt(c0 * c0 + c1 * c1 + c2 * c2);
}

// compute_surface_mismatch (calibration.cpp:225-300), one thread per (element, face) pair of the side set.
// At a face point the element's shape functions reduce to the face's own, so the interpolation of the element
// field at boundaryToElementXi(point) is the face interpolation of the nodal values.  `scale` carries
// mult * dt/T / area, mult = the number of coupled points (the reference adds the face integral at every one).
// J (if not null) += value; b0 (if not null) -= d value / d u   (the adjoint right-hand side is -dJ/dx).
__global__ void k_surface_mismatch(int n, int nf, int32_t const* face_nodes, double const* coords, double const* u,
                                   double const* u_meas, double w0, double w1, double w2, double scale, double* J,
                                   double* b0) {
  int const f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  int32_t const* fn = face_nodes + (size_t)f * 4;
  double X[4][3], Nf[4][4], wdv[4], du[4][3];
  for (int k = 0; k < nf; ++k)
    for (int d = 0; d < 3; ++d) {
      X[k][d] = coords[(size_t)fn[k] * 3 + d];
      du[k][d] = u[(size_t)fn[k] * 3 + d] - u_meas[(size_t)fn[k] * 3 + d];
    }
  int const nq = face_rule(nf, X, Nf, wdv);
  double const wt[3] = {w0, w1, w2};
  double val = 0., grad[4][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}};
  for (int q = 0; q < nq; ++q)
    for (int d = 0; d < 3; ++d) {
      double diff = 0.;
      for (int k = 0; k < nf; ++k) diff += du[k][d] * Nf[q][k];
      val += 0.5 * wt[d] * diff * diff * wdv[q];
      for (int k = 0; k < nf; ++k) grad[k][d] += wt[d] * diff * Nf[q][k] * wdv[q];
    }
  if (J) unsafeAtomicAdd(J, val * scale);
  if (b0)
    for (int k = 0; k < nf; ++k)
      for (int d = 0; d < 3; ++d) unsafeAtomicAdd(&b0[(size_t)fn[k] * 3 + d], -(grad[k][d] * scale));
}

__global__ void k_add_scalar(double* x, double v) { *x += v; }

// S[e][pt][j] = sum over the element's nodes on the load plane of dN_n/dx_j at coupled point pt
template <class E> void load_plane_sums(HostMesh const& m, std::vector<unsigned> const& mask, std::vector<double>& S) {
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

}  // namespace

// ---- used by c8_api.hip -------------------------------------------------------------------------------------
// the point integrand of the adjoint kernels (K3, K5): x, xi or parameter derivatives of the objective
QoiArgs c8_qoi_args(c8_ctx const* c) {
  if (c->qoi_kind == 0) return QoiArgs{1., 0., 0, nullptr};
  return QoiArgs{0., c->cal_balance * c->cal_dt_over_T * c->cal_load_mismatch, c->cal_comp, c->d_cal_S};
}

// preprocess_qoi: the total reaction load of the step and its mismatch with the measured load
int c8_qoi_prepare(c8_ctx* c, FieldArgs const& fa) {
  if (c->qoi_kind == 0) return C8_OK;
  if (!c->d_u_meas) return c8_fail(C8_ERR_ARG, "calibration objective: c8_set_measured has not been called");
  QH(hipMemsetAsync(c->d_scalar, 0, sizeof(double), c->stream));
  AdjointArgs aa{nullptr, nullptr, nullptr, nullptr, nullptr, c->d_scalar, c->d_active, QoiArgs{0., 1., c->cal_comp, c->d_cal_S}};
  MeshTables mt{c->d_conn, c->d_coords, c->d_nodeptr, c->d_pos, c->d_elem_set, nullptr, c->d_params};
  LaunchArgs a{mt, c->ms, fa, aa, SystemArgs{}, 0, c->mesh.nelems, c->stream};
  QH(c->ks.qoi(a));
  double total = 0.;
  QH(hipMemcpyAsync(&total, c->d_scalar, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  QH(hipStreamSynchronize(c->stream));
  double sums[2] = {c->cal_area_local, total};
  if (c->allreduce) c->allreduce(c->allreduce_user, sums, 2);  // PCU_Add_Double (calibration.cpp:138, :351)
  c->cal_area = sums[0];
  c->cal_total_load = sums[1];
  c->cal_load_mismatch = sums[1] - c->cal_load_meas;
  return C8_OK;
}

// the face term: J (nullable) += value, b0 (nullable) -= d value / d u
int c8_qoi_surface(c8_ctx* c, double const* u, double* J, double* b0) {
  if (c->qoi_kind == 0 || c->cal_nfaces == 0) return C8_OK;
  if (!(c->cal_area > 0.)) return c8_fail(C8_ERR_ARG, "calibration objective: the displacement side set has no area");
  double const scale = (double)c->npts0 * c->cal_dt_over_T / c->cal_area;
  int const n = c->cal_nfaces;
  hipLaunchKernelGGL(k_surface_mismatch, dim3((n + 127) / 128), dim3(128), 0, c->stream, n, c->cal_nf, c->d_cal_faces,
                     c->d_coords, u, c->d_u_meas, c->cal_w[0], c->cal_w[1], c->cal_w[2], scale, J, b0);
  QH(hipGetLastError());
  return C8_OK;
}

// Calibration::postprocess on one rank: J += 1/2 balance dt/T load_mismatch^2
int c8_qoi_postprocess(c8_ctx* c, double* J) {
  if (c->qoi_kind == 0) return C8_OK;
  // every part adds the load term; the caller's sum over the parts counts it once (J /= PCU_Comm_Peers(), :378)
  double const Jf = 0.5 * c->cal_balance * c->cal_dt_over_T * c->cal_load_mismatch * c->cal_load_mismatch / c->num_parts;
  hipLaunchKernelGGL(k_add_scalar, dim3(1), dim3(1), 0, c->stream, J, Jf);
  QH(hipGetLastError());
  return C8_OK;
}

extern "C" {

int c8_set_qoi_avg_disp(c8_ctx* c) {
  if (!c) return c8_fail(C8_ERR_ARG, "c8_set_qoi_avg_disp: null ctx");
  c->qoi_kind = 0;
  return C8_OK;
}

int c8_set_qoi_calibration(c8_ctx* c, const c8_calibration_desc* d) {
  if (!c || !d || d->num_faces < 0 || (d->num_faces > 0 && !d->faces)) return c8_fail(C8_ERR_ARG, "c8_set_qoi_calibration: bad argument");
  int const nn = c->mesh.nn, nfn = (nn == 4) ? 3 : 4, nfe = (nn == 4) ? 4 : 6;
  if (d->num_faces > 0 && d->nodes_per_face != nfn) return c8_fail(C8_ERR_ARG, "c8_set_qoi_calibration: faces must have 3 nodes (tet4) or 4 (hex8)");
  if (d->coord_idx < 0 || d->coord_idx > 2 || d->reaction_comp < 0 || d->reaction_comp > 2) return c8_fail(C8_ERR_ARG, "c8_set_qoi_calibration: coordinate index / component out of range");
  std::set<std::vector<int32_t>> side;
  for (int f = 0; f < d->num_faces; ++f) {
    std::vector<int32_t> key(d->faces + (size_t)f * nfn, d->faces + (size_t)(f + 1) * nfn);
    std::sort(key.begin(), key.end());
    side.insert(key);
  }
  // m_mapping_disp (calibration.cpp:98-135): one face per element, a later downward face overwrites an earlier
  // one, every match adds its area; m_mapping_load (qoi.cpp:160-198)
  std::vector<int32_t> faces;
  std::vector<unsigned> mask((size_t)c->mesh.nelems, 0u);
  double area = 0.;
  for (int e = 0; e < c->mesh.nelems; ++e) {
    int32_t const* en = &c->mesh.conn[(size_t)e * nn];
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
        for (int q = 0; q < 3; ++q) X[k][q] = c->mesh.coords[(size_t)en[loc[k]] * 3 + q];
      area += face_area(nfn, X);
    }
    if (hit >= 0) {
      int const* loc = (nn == 4) ? TET_FACES[hit] : HEX_FACES[hit];
      for (int k = 0; k < 4; ++k) faces.push_back(k < nfn ? en[loc[k]] : -1);
    }
    for (int n = 0; n < nn; ++n)
      if (std::abs(c->mesh.coords[(size_t)en[n] * 3 + d->coord_idx] - d->coord_value) < d->coord_tol) mask[e] |= 1u << n;
  }
  if (d->num_faces > 0 && !(area > 0.) && !c->allreduce) return c8_fail(C8_ERR_ARG, "c8_set_qoi_calibration: no element face lies on the displacement side set");
  std::vector<double> S;
  if (nn == 4) load_plane_sums<Elem<C8_TET4>>(c->mesh, mask, S);
  else load_plane_sums<Elem<C8_HEX8>>(c->mesh, mask, S);
  (void)hipFree(c->d_cal_faces);
  (void)hipFree(c->d_cal_S);
  c->d_cal_faces = nullptr;
  c->d_cal_S = nullptr;
  if (!faces.empty()) {
    QH(hipMalloc((void**)&c->d_cal_faces, faces.size() * sizeof(int32_t)));
    QH(hipMemcpy(c->d_cal_faces, faces.data(), faces.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  QH(hipMalloc((void**)&c->d_cal_S, S.size() * sizeof(double)));
  QH(hipMemcpy(c->d_cal_S, S.data(), S.size() * sizeof(double), hipMemcpyHostToDevice));
  c->cal_nfaces = (int)(faces.size() / 4);
  c->cal_nf = nfn;
  c->cal_area = c->cal_area_local = area;  // summed over the parts at the next preprocess
  for (int k = 0; k < 3; ++k) c->cal_w[k] = d->weights[k];
  c->cal_balance = d->balance_factor;
  c->cal_comp = d->reaction_comp;
  c->cal_dt_over_T = d->dt_over_total_time;
  c->cal_load_meas = c->cal_total_load = c->cal_load_mismatch = 0.;
  c->d_u_meas = nullptr;
  c->qoi_kind = 1;
  return C8_OK;
}

int c8_set_allreduce(c8_ctx* c, c8_allreduce_fn fn, void* user, int num_parts) {
  if (!c || num_parts < 1 || (num_parts > 1 && !fn)) return c8_fail(C8_ERR_ARG, "c8_set_allreduce: bad argument");
  c->allreduce = fn;
  c->allreduce_user = user;
  c->num_parts = num_parts;
  return C8_OK;
}

int c8_set_measured(c8_ctx* c, const double* u_meas, double load_meas) {
  if (!c || !u_meas) return c8_fail(C8_ERR_ARG, "c8_set_measured: null argument");
  c->d_u_meas = u_meas;
  c->cal_load_meas = load_meas;
  return C8_OK;
}

int c8_qoi_preprocess(c8_ctx* c, const c8_state* st, double* out) {
  if (!c || !st || !st->x[0] || !st->x[1] || !st->xi || !st->xi_prev) return c8_fail(C8_ERR_ARG, "c8_qoi_preprocess: null argument");
  FieldArgs fa{st->x[0], st->x[1], st->x_prev[0], st->x_prev[1], st->xi_prev, st->xi};
  int const rc = c8_qoi_prepare(c, fa);
  if (rc) return rc;
  if (out) { out[0] = c->cal_area; out[1] = c->cal_total_load; out[2] = c->cal_load_mismatch; }
  return C8_OK;
}

}  // extern "C"
