// c8_primal.hip -- next to the hot path: Dirichlet and traction boundary
// conditions on the assembled device system, y = A x, and the Newton / line-search step driver
// that calls the assembly.  O(boundary) kernels and host control flow; the sparse linear solve is a
// caller-supplied callback (out of scope, linear_solve.cpp).
#include <hip/hip_runtime.h>

#include <cmath>
#include <limits>
#include <string>
#include <vector>

#include "../../include/c8.h"
#include "c8_api_internal.hpp"

using namespace c8;

namespace {

constexpr int TPB = 256;
__host__ __device__ inline int neq_of(int i, int ndims) { return i == 0 ? ndims : 1; }

// dbcs.cpp:68-118: one thread per constrained row
__global__ void k_dirichlet(int n, int resid, int eq, int32_t const* nodes, double const* values, double const* x,
                            int32_t const* nodeptr, int32_t const* nodeadj, double* A_i0, double* A_i1, double* b,
                            int is_adjoint, int nowned, int ndims, int nres) {
  int const t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  int const node = nodes[t];
  if (node >= nowned) return;  // multi-part mesh: the owner applies the condition to its OWNED row (node sets hold owned nodes, dbcs.cpp:66)
  int const ni = neq_of(resid, ndims);
  int const row = node * ni + eq;
  int64_t const np = nodeptr[node], deg = nodeptr[node + 1] - np;
  double diag = 0.;
  for (int j = 0; j < nres; ++j) {
    int const nj = neq_of(j, ndims);
    double* vals = (j == 0 ? A_i0 : A_i1) + np * ni * nj + (int64_t)eq * deg * nj;  // start of this CSR row
    for (int64_t k = 0; k < deg; ++k)
      for (int e = 0; e < nj; ++e) {
        bool const is_diag = (j == resid) && (nodeadj[np + k] == node) && (e == eq);
        if (is_diag) diag = vals[k * nj + e];
        else vals[k * nj + e] = 0.;
      }
  }
  b[row] = is_adjoint ? 0. : diag * (x[row] - values[t]);
}

// tbcs.cpp:46-78: one thread per face
__global__ void k_traction(int n, int npf, int32_t const* faces, double const* traction, double const* coords, double* b) {
  int const f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  int32_t const* fn = faces + (size_t)f * npf;
  double X[4][3];
  for (int a = 0; a < npf; ++a)
    for (int d = 0; d < 3; ++d) X[a][d] = coords[(size_t)fn[a] * 3 + d];
  if (npf == 2) {  // sides of a 2-D mesh: edges, order-1 rule: midpoint, w = 2, dv = length / 2, N = 1/2; two equations per node
    double const ex = X[1][0] - X[0][0], ey = X[1][1] - X[0][1];
    double const len = sqrt(ex * ex + ey * ey);
    double const* T = traction + (size_t)f * 3;
    for (int a = 0; a < 2; ++a)
      for (int d = 0; d < 2; ++d) unsafeAtomicAdd(&b[(size_t)fn[a] * 2 + d], -(T[d] * 0.5 * len));
  } else if (npf == 3) {  // order-1 rule: centroid, w = 1/2, dv = 2 * area, N = 1/3
    double const e1[3] = {X[1][0] - X[0][0], X[1][1] - X[0][1], X[1][2] - X[0][2]};
    double const e2[3] = {X[2][0] - X[0][0], X[2][1] - X[0][1], X[2][2] - X[0][2]};
    double const cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
    double const dv = sqrt(cx * cx + cy * cy + cz * cz);
    double const* T = traction + (size_t)f * 3;
    for (int a = 0; a < 3; ++a)
      for (int d = 0; d < 3; ++d) unsafeAtomicAdd(&b[(size_t)fn[a] * 3 + d], -(T[d] * (1. / 3.) * 0.5 * dv));
  } else {  // quad4, 2x2 Gauss
    double const g = 0.5773502691896257645;
    for (int q = 0; q < 4; ++q) {
      double const xi = (q & 1) ? g : -g, eta = (q & 2) ? g : -g;
      double const sx[4] = {-1., 1., 1., -1.}, sy[4] = {-1., -1., 1., 1.};
      double N[4], dx[3] = {0., 0., 0.}, dy[3] = {0., 0., 0.};
      for (int a = 0; a < 4; ++a) {
        N[a] = 0.25 * (1. + sx[a] * xi) * (1. + sy[a] * eta);
        double const dNx = 0.25 * sx[a] * (1. + sy[a] * eta), dNy = 0.25 * sy[a] * (1. + sx[a] * xi);
        for (int d = 0; d < 3; ++d) { dx[d] += dNx * X[a][d]; dy[d] += dNy * X[a][d]; }
      }
      double const cx = dx[1] * dy[2] - dx[2] * dy[1], cy = dx[2] * dy[0] - dx[0] * dy[2], cz = dx[0] * dy[1] - dx[1] * dy[0];
      double const dv = sqrt(cx * cx + cy * cy + cz * cz);
      double const* T = traction + ((size_t)f * 4 + q) * 3;
      for (int a = 0; a < 4; ++a)
        for (int d = 0; d < 3; ++d) unsafeAtomicAdd(&b[(size_t)fn[a] * 3 + d], -(T[d] * N[a] * dv));
    }
  }
}

// y_i[row] (+)= sum_k A_ij[row][k] x_j[col k]; one thread per row
__global__ void k_spmv(int nnodes, int i, int j, int32_t const* nodeptr, int32_t const* nodeadj, double const* vals,
                       double const* x, double* y, int accumulate, int ndims) {
  int const row = blockIdx.x * blockDim.x + threadIdx.x;
  int const ni = neq_of(i, ndims), nj = neq_of(j, ndims);
  if (row >= nnodes * ni) return;
  int const node = row / ni, eq = row - node * ni;
  int64_t const np = nodeptr[node], deg = nodeptr[node + 1] - np;
  double const* v = vals + np * ni * nj + (int64_t)eq * deg * nj;
  double s = 0.;
  for (int64_t k = 0; k < deg; ++k) {
    int const cn = nodeadj[np + k];
    for (int e = 0; e < nj; ++e) s += v[k * nj + e] * x[(size_t)cn * nj + e];
  }
  y[row] = accumulate ? y[row] + s : s;
}

__global__ void k_axpy(size_t n, double a, double const* x, double* y) {
  size_t const t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t < n) y[t] += a * x[t];
}
__global__ void k_scale(size_t n, double a, double* y) {
  size_t const t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t < n) y[t] *= a;
}
__global__ void k_dot(size_t n, double const* x, double const* y, double* out) {
  __shared__ double sm[TPB / 64];
  double s = 0.;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) s += x[t] * y[t];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.;
    for (int w = 0; w < TPB / 64; ++w) tot += sm[w];
    unsafeAtomicAdd(out, tot);
  }
}

int grid_of(size_t n) { return (int)((n + TPB - 1) / TPB); }

}  // namespace

#define C8P_HIP(call)                                                                                                   \
  do {                                                                                                                  \
    hipError_t err__ = (call);                                                                                          \
    if (err__ != hipSuccess) return c8_fail(C8_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(err__));       \
  } while (0)

static int dot(c8_ctx* c, size_t n, double const* x, double const* y, double* result) {
  C8P_HIP(hipMemsetAsync(c->d_scalar, 0, sizeof(double), c->stream));
  int const g = std::min(grid_of(n), 1024);
  hipLaunchKernelGGL(k_dot, dim3(g), dim3(TPB), 0, c->stream, n, x, y, c->d_scalar);
  C8P_HIP(hipMemcpyAsync(result, c->d_scalar, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  C8P_HIP(hipStreamSynchronize(c->stream));
  return C8_OK;
}

extern "C" {

int c8_apply_dirichlet(c8_ctx* c, int n, const c8_dbc* dbcs, const double* const x[2], const c8_system* sys, int is_adjoint) {
  if (!c || n < 0 || (n > 0 && !dbcs) || !x || !sys) return c8_fail(C8_ERR_ARG, "c8_apply_dirichlet: null argument");
  for (int q = 0; q < n; ++q) {  // in deck order: later conditions overwrite earlier ones on shared rows
    c8_dbc const& d = dbcs[q];
    if (d.resid < 0 || d.resid >= c->nres || d.eq < 0 || d.eq >= neq_of(d.resid, c->ndims)) return c8_fail(C8_ERR_ARG, "c8_apply_dirichlet: bad residual/equation index");
    if (d.n <= 0) continue;
    hipLaunchKernelGGL(k_dirichlet, dim3(grid_of(d.n)), dim3(TPB), 0, c->stream, d.n, d.resid, d.eq, d.nodes, d.values,
                       x[d.resid], c->d_nodeptr, c->d_nodeadj, sys->A[d.resid][0], sys->A[d.resid][1], sys->b[d.resid], is_adjoint,
                       c->halo ? c8_halo_num_owned(c->halo) : c->mesh.nnodes, c->ndims, c->nres);
    C8P_HIP(hipGetLastError());
  }
  return C8_OK;
}

int c8_apply_traction(c8_ctx* c, int n, const c8_tbc* tbcs, const c8_system* sys) {
  if (!c || n < 0 || (n > 0 && !tbcs) || !sys) return c8_fail(C8_ERR_ARG, "c8_apply_traction: null argument");

  for (int q = 0; q < n; ++q) {
    c8_tbc const& t = tbcs[q];
    bool const side_ok = c->ndims == 2 ? t.nodes_per_face == 2 : (t.nodes_per_face == 3 || t.nodes_per_face == 4);
    if (t.resid != 0 || !side_ok) return c8_fail(C8_ERR_ARG, "c8_apply_traction: tractions act on residual 0 over the sides of the mesh (edges of a 2-D mesh, tri3 / quad4 faces of a 3-D mesh)");
    if (t.n <= 0) continue;
    hipLaunchKernelGGL(k_traction, dim3(grid_of(t.n)), dim3(TPB), 0, c->stream, t.n, t.nodes_per_face, t.faces, t.traction,
                       c->d_coords, sys->b[0]);
    C8P_HIP(hipGetLastError());
  }
  return C8_OK;
}

int c8_face_points(int npf, int n, const double* coords, const int32_t* faces, double* xyz) {
  if ((npf < 2 || npf > 4) || n < 0 || !coords || !faces || !xyz) return c8_fail(C8_ERR_ARG, "c8_face_points: bad argument");
  double const g = 0.5773502691896257645;
  for (int f = 0; f < n; ++f) {
    int32_t const* fn = faces + (size_t)f * npf;
    if (npf == 2) {
      for (int d = 0; d < 3; ++d) xyz[(size_t)f * 3 + d] = 0.5 * (coords[(size_t)fn[0] * 3 + d] + coords[(size_t)fn[1] * 3 + d]);
    } else if (npf == 3) {
      for (int d = 0; d < 3; ++d)
        xyz[(size_t)f * 3 + d] = (coords[(size_t)fn[0] * 3 + d] + coords[(size_t)fn[1] * 3 + d] + coords[(size_t)fn[2] * 3 + d]) / 3.;
    } else {
      double const sx[4] = {-1., 1., 1., -1.}, sy[4] = {-1., -1., 1., 1.};
      for (int q = 0; q < 4; ++q) {
        double const xi = (q & 1) ? g : -g, eta = (q & 2) ? g : -g;
        for (int d = 0; d < 3; ++d) {
          double s = 0.;
          for (int a = 0; a < 4; ++a) s += 0.25 * (1. + sx[a] * xi) * (1. + sy[a] * eta) * coords[(size_t)fn[a] * 3 + d];
          xyz[((size_t)f * 4 + q) * 3 + d] = s;
        }
      }
    }
  }
  return C8_OK;
}

int c8_apply_A(c8_ctx* c, const c8_system* sys, const double* const x[2], double* const y[2]) {
  if (!c || !sys || !x || !y) return c8_fail(C8_ERR_ARG, "c8_apply_A: null argument");
  for (int i = 0; i < c->nres; ++i)
    for (int j = 0; j < c->nres; ++j) {
      hipLaunchKernelGGL(k_spmv, dim3(grid_of((size_t)c->mesh.nnodes * neq_of(i, c->ndims))), dim3(TPB), 0, c->stream, c->mesh.nnodes, i, j,
                         c->d_nodeptr, c->d_nodeadj, sys->A[i][j], x[j], y[i], j, c->ndims);
      C8P_HIP(hipGetLastError());
    }
  return C8_OK;
}

}  // extern "C"

// ---- step drivers ------------------------------------------------------------------------------------------------
// Shared plumbing of the two drivers.  With a halo attached to the context the step runs over all parts: the assembled
// system is gathered to its owners, boundary conditions, norms and dot products act on the OWNED rows (the first
// `nowned` node rows) and are summed over the parts, and what the linear solve returns on the owned nodes is imported to
// the ghost and phantom copies before it is used.
namespace {

struct StepSystem {
  c8_ctx* c;
  const c8_system* sys;
  size_t nloc[2];      // local rows per block (all local nodes)
  size_t nown[2];      // owned rows per block
  size_t nnz[2][2];
  bool parts;
  int nres;            // blocks in use: 2, or 1 under mechanics_plane_stress
  StepSystem(c8_ctx* ctx, const c8_system* s) : c(ctx), sys(s) {
    parts = ctx->halo != nullptr;
    nres = ctx->nres;
    int const no = parts ? c8_halo_num_owned(ctx->halo) : ctx->mesh.nnodes;
    for (int i = 0; i < 2; ++i) {
      nloc[i] = (size_t)ctx->mesh.nnodes * neq_of(i, ctx->ndims);
      nown[i] = (size_t)no * neq_of(i, ctx->ndims);
      for (int j = 0; j < 2; ++j) nnz[i][j] = (size_t)ctx->graph.nodeptr[ctx->mesh.nnodes] * neq_of(i, ctx->ndims) * neq_of(j, ctx->ndims);
    }
  }
  int zero() const {  // la->zero_all (linear_alg.cpp:118-129)
    for (int i = 0; i < nres; ++i) {
      C8P_HIP(hipMemsetAsync(sys->b[i], 0, nloc[i] * sizeof(double), c->stream));
      for (int j = 0; j < nres; ++j) C8P_HIP(hipMemsetAsync(sys->A[i][j], 0, nnz[i][j] * sizeof(double), c->stream));
    }
    return C8_OK;
  }
  // The status of an assembly agreed between the parts (PCU_Add_Int, primal.cpp:100,164): every part learns whether
  // ANY local solve failed, or any part hit an error, and all take the same branch.
  int agree(int rc) const {
    if (!parts) return rc;
    double v[2] = {rc == C8_LOCAL_SOLVE_FAILED ? 1. : 0., (rc != C8_OK && rc != C8_LOCAL_SOLVE_FAILED) ? 1. : 0.};
    int const ra = c8_parts_allreduce(c, v, 2);
    if (ra != C8_OK) return ra;
    if (rc != C8_OK && rc != C8_LOCAL_SOLVE_FAILED) return rc;
    if (v[1] > 0.) return c8_fail(C8_ERR_DEVICE, "step driver: the assembly failed on another part");
    if (v[0] > 0.) return rc == C8_OK ? c8_fail(C8_LOCAL_SOLVE_FAILED, "a local constitutive Newton solve did not converge on another part") : rc;
    return C8_OK;
  }
  // gather_A / gather_b (primal.cpp:110-111), overlapped with the owned rows' sums when the row sums run in two parts
  int gather() const {
    int rc = C8_OK;
    if (parts) rc = c8_halo_gather_start(c->halo, sys, C8_HALO_A | C8_HALO_B);
    if (rc == C8_OK) rc = c8_gather_finish(c);  // two-part row sums (c8_set_gather_early_nodes): the rest of the rows
    if (rc == C8_OK && parts) rc = c8_halo_gather_finish(c->halo, sys);
    return rc;
  }
  // sum over the parts of x . y on the owned rows of both blocks (LinearAlg::norm_b, linear_alg.cpp:138-146)
  int dot_owned(double* const x[2], double* const y[2], double* out) const {
    double s[2] = {0., 0.};
    int rc;
    for (int i = 0; i < nres; ++i)
      if ((rc = dot(c, nown[i], x[i], y[i], &s[i])) != C8_OK) return rc;
    double v = s[0] + s[1];
    if (parts && (rc = c8_parts_allreduce(c, &v, 1)) != C8_OK) return rc;
    *out = v;
    return C8_OK;
  }
};

// Two-point cubic backtracking (line_search.hpp:56-135 with the reference's safeguards): the model through
// (0, f0, g0) and (t, ft, gt) has its minimiser at t - t (gt + w - v) / (gt - g0 + 2 w), v = g0 + gt - 3 (f0 - ft) / (0 - t),
// w = sqrt(v^2 - g0 gt); without a real interior minimiser the step is halved.  The arithmetic is kept operation for
// operation so that Newton histories equal the reference's.
double cubic_step(double f0, double g0, double t, double ft, double gt) {
  double const v = g0 + gt - 3. * (f0 - ft) / (0. - t);
  double const disc = v * v - g0 * gt;
  if (disc < 0.) return 0.5 * t;
  double const w = std::sqrt(disc);
  double const q = gt - g0 + 2. * w;
  if (q == 0.) return 0.5 * t;
  return t - t * (gt + w - v) / q;
}

}  // namespace

extern "C" {

int c8_primal_solve_step(c8_ctx* c, const c8_state* st, const c8_system* sys, int ndbc, const c8_dbc* dbcs, int ntbc,
                         const c8_tbc* tbcs, const c8_newton_opts* o, c8_linear_solve_fn solve, void* user, int32_t* iters_out) {
  if (!c || !st || !sys || !o || !solve) return c8_fail(C8_ERR_ARG, "c8_primal_solve_step: null argument");
  StepSystem S(c, sys);
  if (!c->d_work[0]) {  // (all four also under mechanics_plane_stress, where the p-sized ones stay unused)
    for (int k = 0; k < 4; ++k) C8P_HIP(hipMalloc((void**)&c->d_work[k], S.nloc[k & 1] * sizeof(double)));
  }
  double* dx[2] = {c->d_work[0], c->d_work[1]};
  double* Adx[2] = {c->d_work[2], c->d_work[3]};
  double* x[2] = {const_cast<double*>(st->x[0]), const_cast<double*>(st->x[1])};
  double* b[2] = {sys->b[0], sys->b[1]};
  int const saved_async = c->async;

  // zero_all + eval_forward_jacobian + status over the parts + tbcs + gather_A/gather_b + dbcs (primal.cpp:97-114)
  auto assemble = [&]() -> int {
    int rc = S.zero();
    if (rc != C8_OK) return rc;
    c->async = 0;
    rc = S.agree(c8_assemble_forward_jacobian(c, st, sys));
    c->async = saved_async;
    if (rc != C8_OK) {
      // a failed assembly leaves A and b undefined (evaluations.cpp:95-97): the second part of two-part row sums
      // (c8_set_gather_early_nodes) is dropped with it, so that the line search can contract and assemble again
      c->gather_pending = c->pending_node_rows = false;
      return rc;
    }
    // tractions are added to the GHOST-distributed residual before the gather, as in the reference; in assign mode the
    // owned rows' sums would overwrite them, so those run first
    if (ntbc > 0 && c->assign_mode && (rc = c8_gather_finish(c)) != C8_OK) return rc;
    if ((rc = c8_apply_traction(c, ntbc, tbcs, sys)) != C8_OK) return rc;
    if ((rc = S.gather()) != C8_OK) return rc;
    return c8_apply_dirichlet(c, ndbc, dbcs, st->x, sys, 0);
  };
  auto residual_norm = [&](double* out) -> int {
    double s;
    int const rc = S.dot_owned(b, b, &s);
    *out = std::sqrt(s);
    return rc;
  };
  // Disc::add_to_soln(x, dx, alpha) (disc.cpp:893-949): dx has been imported to the ghost and phantom copies, so every
  // copy of a node takes the same update and stays equal to its owner's value bit for bit
  auto move = [&](double alpha) {
    for (int i = 0; i < S.nres; ++i)
      hipLaunchKernelGGL(k_axpy, dim3(grid_of(S.nloc[i])), dim3(TPB), 0, c->stream, S.nloc[i], alpha, dx[i], x[i]);
  };

  int iter = 1;
  bool converged = false;
  double r_first = 1.;
  int rc = C8_OK;
  while ((iter <= o->max_iters) && !converged) {
    rc = assemble();
    if (rc != C8_OK) break;  // local solve failed at the base point (primal.cpp:101-104), or an error
    double r_abs;
    if ((rc = residual_norm(&r_abs)) != C8_OK) break;
    if (iter == 1) r_first = r_abs;
    if ((r_abs < o->abs_tol) || (r_abs / r_first < o->rel_tol)) { converged = true; break; }
    for (int i = 0; i < S.nres; ++i)  // la->scale_b(-1.)
      hipLaunchKernelGGL(k_scale, dim3(grid_of(S.nloc[i])), dim3(TPB), 0, c->stream, S.nloc[i], -1., sys->b[i]);
    C8P_HIP(hipStreamSynchronize(c->stream));
    if (solve(user, sys, dx) != 0) { rc = c8_fail(C8_ERR_ARG, "c8_primal_solve_step: linear solve callback failed"); break; }
    if (S.parts && (rc = c8_halo_scatter_x(c->halo, dx)) != C8_OK) break;
    move(1.);
    if (o->line_search) {  // primal.cpp:139-197: merit 1/2 |R|^2, slope at 0 = -|R_0|^2, slope at t = R(t) . (A dx)
      double const f0 = 0.5 * r_abs * r_abs, g0 = -2. * f0;
      // every trial starts its local solves from the local state of the base point (primal.cpp:146-156), so the merit is
      // one fixed function of the step and a trial whose local solves diverged leaves nothing behind
      size_t const xi_bytes = (size_t)c->mesh.nelems * c->npts0 * c->nloc * sizeof(double);
      if (!c->d_xi_saved) C8P_HIP(hipMalloc((void**)&c->d_xi_saved, xi_bytes));
      C8P_HIP(hipMemcpyAsync(c->d_xi_saved, st->xi, xi_bytes, hipMemcpyDeviceToDevice, c->stream));
      double t = 1., t_now = 1., t_best = 1., f_best = std::numeric_limits<double>::max();
      bool any_assembled = false, accepted = false;
      for (int trial = 1; trial <= o->max_evals && !accepted; ++trial) {
        move(t - t_now);
        t_now = t;
        C8P_HIP(hipMemcpyAsync(st->xi, c->d_xi_saved, xi_bytes, hipMemcpyDeviceToDevice, c->stream));
        int const arc = assemble();
        if (arc == C8_LOCAL_SOLVE_FAILED) { t *= 0.5; continue; }  // a local solve diverged: contract and retry
        if (arc != C8_OK) { rc = arc; break; }
        double r_t;
        if ((rc = residual_norm(&r_t)) != C8_OK) break;
        double const ft = 0.5 * r_t * r_t;
        any_assembled = true;
        if (ft < f_best) { f_best = ft; t_best = t; }
        if (ft <= f0 + t * (o->sufficient_decrease * g0)) { accepted = true; break; }
        const double* cdx[2] = {dx[0], dx[1]};
        if ((rc = c8_apply_A(c, sys, cdx, Adx)) != C8_OK) break;
        double gt;
        if ((rc = S.dot_owned(b, Adx, &gt)) != C8_OK) break;
        double const t_model = cubic_step(f0, g0, t, ft, gt);
        t = std::min(std::max(t_model, o->min_backtrack * t), o->max_backtrack * t);
      }
      if (rc != C8_OK) break;
      if (!accepted) {
        if (!any_assembled) { rc = c8_fail(C8_LOCAL_SOLVE_FAILED, "line search could not assemble at any trial step"); break; }
        t = t_best;
      }
      move(t - t_now);
    }
    iter++;
  }
  C8P_HIP(hipStreamSynchronize(c->stream));
  if (iters_out) *iters_out = iter;
  if (rc != C8_OK) return rc;
  if (!converged) return c8_fail(C8_NOT_CONVERGED, "Newton's method failed in the allowed iterations");
  return C8_OK;
}

int c8_adjoint_solve_step(c8_ctx* c, const c8_state* st, const c8_system* sys, int ndbc, const c8_dbc* dbcs,
                          c8_linear_solve_fn solve, void* user, double* const z[2], double* phi, double* g, double* f,
                          double* grad) {
  if (!c || !st || !sys || !solve || !z || !z[0] || (c->nres == 2 && !z[1]) || !phi || !g || !f || !grad)
    return c8_fail(C8_ERR_ARG, "c8_adjoint_solve_step: null argument");
  StepSystem S(c, sys);
  int rc = S.zero();  // la->zero_all (adjoint.cpp:118)
  if (rc != C8_OK) return rc;
  int const saved_async = c->async;
  c->async = 0;
  rc = S.agree(c8_assemble_adjoint_jacobian(c, st, g, f, sys));
  if (rc != C8_OK) c->gather_pending = c->pending_node_rows = false;  // nothing is left waiting for c8_gather_finish after a failed assembly
  if (rc == C8_OK) rc = S.gather();  // gather_A / gather_b (adjoint.cpp:128-129)
  const double* zc[2] = {z[0], z[1]};
  if (rc == C8_OK) rc = c8_apply_dirichlet(c, ndbc, dbcs, zc, sys, 1);  // apply_primal_dbcs(..., is_adjoint) (adjoint.cpp:137)
  if (rc == C8_OK) {
    if (hipStreamSynchronize(c->stream) != hipSuccess) rc = c8_fail(C8_ERR_DEVICE, "c8_adjoint_solve_step: sync failed");
    else if (solve(user, sys, z) != 0) rc = c8_fail(C8_ERR_ARG, "c8_adjoint_solve_step: linear solve callback failed");
  }
  if (rc == C8_OK && S.parts) rc = c8_halo_scatter_x(c->halo, z);           // the adjoint field is synchronised (adjoint.cpp:148)
  if (rc == C8_OK) rc = S.agree(c8_solve_adjoint_local(c, st, zc, phi, g, f));  // adjoint.cpp:182
  if (rc == C8_OK) rc = c8_param_gradient(c, st, zc, phi, grad);            // adjoint_objective.cpp:90-93 (this part's share)
  c->async = saved_async;
  return rc;
}

int c8_transform_params(int n, const double* values, const int32_t* kind, const double* a, const double* b,
                        int from_canonical, double* out) {
  if (n < 0 || !values || !kind || !a || !b || !out) return c8_fail(C8_ERR_ARG, "c8_transform_params: null argument");
  for (int i = 0; i < n; ++i) {
    double const v = values[i];
    if (kind[i] == C8_SCALE_NONE) out[i] = v;
    else if (kind[i] == C8_SCALE_LOG) out[i] = from_canonical ? a[i] * std::exp(v) : std::log(v / a[i]);
    else if (kind[i] == C8_SCALE_BOUNDS) {
      double const span = 0.5 * (b[i] - a[i]), mean = 0.5 * (a[i] + b[i]);
      if (from_canonical) out[i] = span * v + mean;
      else {
        double const cl = v < a[i] ? a[i] : (v > b[i] ? b[i] : v);
        out[i] = (cl - mean) / span;
      }
    } else return c8_fail(C8_ERR_ARG, "c8_transform_params: unknown scale kind");
  }
  return C8_OK;
}

int c8_transform_gradient(int n, const double* grad, const double* values, const int32_t* kind, const double* a,
                          const double* b, double* out) {
  if (n < 0 || !grad || !values || !kind || !a || !b || !out) return c8_fail(C8_ERR_ARG, "c8_transform_gradient: null argument");
  for (int i = 0; i < n; ++i) {
    if (kind[i] == C8_SCALE_NONE) out[i] = grad[i];
    else if (kind[i] == C8_SCALE_LOG) out[i] = grad[i] * values[i];
    else if (kind[i] == C8_SCALE_BOUNDS) out[i] = grad[i] * 0.5 * (b[i] - a[i]);
    else return c8_fail(C8_ERR_ARG, "c8_transform_gradient: unknown scale kind");
  }
  return C8_OK;
}

}  // extern "C"
