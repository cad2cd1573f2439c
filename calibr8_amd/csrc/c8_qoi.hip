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
#include "c8_qoi_host.hpp"

using namespace c8;

#define QH(call)                                                                                   \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) return c8_fail(C8_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

namespace {

// the face term, one thread per (element, face) pair of the side set.  `scale` carries mult * dt/T / area, mult = the
// number of coupled points (the reference adds the face integral at every one).
// J (if not null) += value; b0 (if not null) -= d value / d u   (the adjoint right-hand side is -dJ/dx).
__global__ void k_surface_mismatch(int n, int nf, int32_t const* face_nodes, double const* coords, double const* u,
                                   double const* u_meas, double w0, double w1, double w2, double scale, double* J,
                                   double* b0, int ndims) {
  int const f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  int32_t const* fn = face_nodes + (size_t)f * 4;
  double const wt[3] = {w0, w1, w2};
  double grad[4][3];
  double const val = surface_mismatch_face(nf, fn, coords, u, u_meas, wt, grad, ndims);
  if (J) unsafeAtomicAdd(J, val * scale);
  if (b0)
    for (int k = 0; k < nf; ++k)
      for (int d = 0; d < ndims; ++d) unsafeAtomicAdd(&b0[(size_t)fn[k] * ndims + d], -(grad[k][d] * scale));
}

__global__ void k_add_scalar(double* x, double v) { *x += v; }

}  // namespace

// ---- used by c8_api.hip -------------------------------------------------------------------------------------
// the point integrand of the adjoint kernels (K3, K5): x, xi or parameter derivatives of the objective
QoiArgs c8_qoi_args(c8_ctx const* c) {
  if (c->qoi_kind == 0) return QoiArgs{1., 0., 0, nullptr, (double)c->ndims};
  return QoiArgs{0., c->cal_balance * c->cal_dt_over_T * c->cal_load_mismatch, c->cal_comp, c->d_cal_S, (double)c->ndims, c->ms.thickness};
}

// preprocess_qoi: the total reaction load of the step and its mismatch with the measured load
int c8_qoi_prepare(c8_ctx* c, FieldArgs const& fa) {
  if (c->qoi_kind == 0) return C8_OK;
  if (!c->d_u_meas) return c8_fail(C8_ERR_ARG, "calibration objective: c8_set_measured has not been called");
  QH(hipMemsetAsync(c->d_scalar, 0, sizeof(double), c->stream));
  AdjointArgs aa{nullptr, nullptr, nullptr, nullptr, nullptr, c->d_scalar, c->d_active, QoiArgs{0., 1., c->cal_comp, c->d_cal_S, (double)c->ndims, c->ms.thickness}};
  MeshTables mt{c->d_conn, c->d_coords, c->d_nodeptr, c->d_pos, c->d_elem_set, nullptr, c->d_params};
  LaunchArgs a{mt, c->ms, fa, aa, SystemArgs{}, 0, c->mesh.nelems, c->stream};
  QH(c->ks.qoi(a));
  double total = 0.;
  QH(hipMemcpyAsync(&total, c->d_scalar, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  QH(hipStreamSynchronize(c->stream));
  double sums[2] = {c->cal_area_local, total};
  int const rca = c8_parts_allreduce(c, sums, 2);  // PCU_Add_Double (calibration.cpp:138, :351)
  if (rca) return rca;
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
                     c->d_coords, u, c->d_u_meas, c->cal_w[0], c->cal_w[1], c->cal_w[2], scale, J, b0, c->ndims);
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
  int const nn = c->mesh.nn, nfn = (nn == 8) ? 4 : 3;
  // 3-D: faces of the displacement side set; 2-D (calibration.cpp:76-104): every element, or the listed element ids
  if (d->num_faces > 0 && d->nodes_per_face != (nn == 3 ? 1 : nfn))
    return c8_fail(C8_ERR_ARG, "c8_set_qoi_calibration: faces must have 3 nodes (tet4) or 4 (hex8); on a tri3 mesh the list holds element ids (nodes_per_face = 1)");
  if (nn == 3)
    for (int f = 0; f < d->num_faces; ++f)
      if (d->faces[f] < 0 || d->faces[f] >= c->mesh.nelems) return c8_fail(C8_ERR_ARG, "c8_set_qoi_calibration: element id out of range");
  if (d->coord_idx < 0 || d->coord_idx >= c->ndims || d->reaction_comp < 0 || d->reaction_comp >= c->ndims) return c8_fail(C8_ERR_ARG, "c8_set_qoi_calibration: coordinate index / component out of range");
  CalibrationTables t;
  calibration_tables(c->mesh, d->num_faces, d->faces, d->coord_idx, d->coord_value, d->coord_tol, t);
  std::vector<int32_t> const& faces = t.faces;
  std::vector<double> const& S = t.S;
  double const area = t.area;
  if ((d->num_faces > 0 || nn == 3) && !(area > 0.) && !c->allreduce && !c->halo) return c8_fail(C8_ERR_ARG, "c8_set_qoi_calibration: no element face lies on the displacement side set");
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
  if (!c || !st || !st->x[0] || (c->nres == 2 && !st->x[1]) || !st->xi || !st->xi_prev) return c8_fail(C8_ERR_ARG, "c8_qoi_preprocess: null argument");
  FieldArgs fa{st->x[0], st->x[1], st->x_prev[0], st->x_prev[1], st->xi_prev, st->xi};
  int const rc = c8_qoi_prepare(c, fa);
  if (rc) return rc;
  if (out) { out[0] = c->cal_area; out[1] = c->cal_total_load; out[2] = c->cal_load_mismatch; }
  return C8_OK;
}

}  // extern "C"
