/* tests/abi_client/abi_client.c -- TEST INFRASTRUCTURE: a compiled client of include/c8.h, no Python and no ctypes between
 * the caller and libc8.so.  Builds as C11 (gcc) and as C++17 (g++), links libc8.so and the HIP runtime, and walks the
 * sequence of INTEGRATION.md sections 1-3 on a small hex8 brick the way a CALIBR8 maintainer's binding would
 * (evaluations.hpp:23-84 -> c8_assemble_*; primal.cpp:96-100; linear_alg.cpp:53-86 -> c8_halo_*):
 *   c8_create -> c8_graph -> c8_init_variables -> c8_assemble_forward_jacobian (twice: accumulate-into) ->
 *   c8_assemble_residual -> c8_comm_create_host / c8_halo_build / c8_halo_attach / c8_halo_gather / c8_halo_scatter_x
 *   (one part: empty exchange lists) -> c8_comm_allreduce_sum -> c8_destroy.
 * Writes inputs and outputs to argv[1] as raw doubles; tests/test_abi.py runs the same inputs through the Python layer
 * and compares bit for bit. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "c8.h"

#define CHECK(call)                                                                   \
  do {                                                                                \
    int rc__ = (call);                                                                \
    if (rc__ != C8_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc__, c8_last_error()); return 2; } \
  } while (0)
#define HIP(call)                                                                     \
  do {                                                                                \
    hipError_t e__ = (call);                                                          \
    if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e__)); return 3; } \
  } while (0)

static int host_exchange(void* user, const double* send, const int64_t* send_counts, double* recv, const int64_t* recv_counts) {
  (void)user; (void)send; (void)recv;
  return (send_counts[0] == 0 && recv_counts[0] == 0) ? 0 : 1;  /* one part: nothing travels */
}
static int host_allreduce(void* user, double* values, int n) { (void)user; (void)values; (void)n; return 0; }

static double* dev_copy(const double* h, size_t n) {
  double* d = NULL;
  if (hipMalloc((void**)&d, (n ? n : 1) * sizeof(double)) != hipSuccess) return NULL;
  if (h) { if (hipMemcpy(d, h, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return NULL; }
  else if (hipMemset(d, 0, (n ? n : 1) * sizeof(double)) != hipSuccess) return NULL;
  return d;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: abi_client <output file>\n"); return 1; }
  enum { NX = 5, NY = 4, NZ = 3 };
  int const nnodes = (NX + 1) * (NY + 1) * (NZ + 1), nelems = NX * NY * NZ;
  double* coords = (double*)malloc(sizeof(double) * 3 * (size_t)nnodes);
  int32_t* conn = (int32_t*)malloc(sizeof(int32_t) * 8 * (size_t)nelems);
  CHECK(c8_brick_mesh(NX, NY, NZ, 1.0, 0.8, 0.6, coords, conn));
  double params[6] = {1000.0, 0.25, 100.0, 2.0, 0.0, 0.0};  /* small_J2: E nu K Y cte delta_T */
  c8_mesh_desc md;
  memset(&md, 0, sizeof md);
  md.elem_type = C8_ELEM_HEX8; md.num_nodes = nnodes; md.num_elems = nelems; md.num_elem_sets = 1;
  md.coords = coords; md.conn = conn;
  c8_model_desc mo;
  memset(&mo, 0, sizeof mo);
  mo.global_type = "mechanics"; mo.local_type = "small_J2"; mo.stabilization_multiplier = 1.0;
  mo.local_max_iters = 500; mo.local_abs_tol = 1e-12; mo.local_rel_tol = 1e-12; mo.num_params = 6; mo.params = params;
  c8_ctx* ctx = NULL;
  CHECK(c8_create(&md, &mo, &ctx));
  if (c8_num_local_dofs(ctx) != 7 || c8_num_local_points(ctx) != 8 || c8_num_dims(ctx) != 3 || c8_num_residuals(ctx) != 2) return 4;
  int64_t nnz[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) nnz[i][j] = c8_graph_nnz(ctx, i, j);
  int64_t* rowptr = (int64_t*)malloc(sizeof(int64_t) * ((size_t)nnodes + 1));
  int32_t* colidx = (int32_t*)malloc(sizeof(int32_t) * (size_t)nnz[1][1]);
  CHECK(c8_graph(ctx, 1, 1, rowptr, colidx));  /* the node graph (block p-p) */
  /* prescribed state: a ramped stretch along y, a part of the bar plastic */
  size_t const nu = 3 * (size_t)nnodes, nxi = (size_t)nelems * 8 * 7;
  double* u = (double*)calloc(nu, sizeof(double));
  double* p = (double*)calloc((size_t)nnodes, sizeof(double));
  for (int n = 0; n < nnodes; ++n) {
    double const x = coords[3 * n], y = coords[3 * n + 1], z = coords[3 * n + 2], e = 0.006 * y / 0.8;
    u[3 * n + 0] = -0.25 * e * x + 1e-5 * sin(7.0 * y + 3.0 * z);
    u[3 * n + 1] = 0.5 * 0.006 * y * y / 0.8 + 1e-5 * sin(5.0 * x);
    u[3 * n + 2] = -0.25 * e * z + 1e-5 * cos(4.0 * x + y);
    p[n] = -(1000.0 / 1.5) * 0.5 * e;
  }
  double* xi0 = (double*)malloc(sizeof(double) * nxi);
  CHECK(c8_init_variables(ctx, xi0));
  double* d_u = dev_copy(u, nu); double* d_p = dev_copy(p, (size_t)nnodes);
  double* d_u0 = dev_copy(NULL, nu); double* d_p0 = dev_copy(NULL, (size_t)nnodes);
  double* d_xip = dev_copy(xi0, nxi); double* d_xi = dev_copy(xi0, nxi);
  double* d_A[2][2]; double* d_b[2];
  for (int i = 0; i < 2; ++i) {
    d_b[i] = dev_copy(NULL, (size_t)nnodes * (i == 0 ? 3 : 1));
    for (int j = 0; j < 2; ++j) d_A[i][j] = dev_copy(NULL, (size_t)nnz[i][j]);
  }
  if (!d_u || !d_p || !d_u0 || !d_p0 || !d_xip || !d_xi || !d_b[0] || !d_b[1] || !d_A[0][0] || !d_A[0][1] || !d_A[1][0] || !d_A[1][1]) return 5;
  c8_state st;
  st.x[0] = d_u; st.x[1] = d_p; st.x_prev[0] = d_u0; st.x_prev[1] = d_p0; st.xi_prev = d_xip; st.xi = d_xi;
  c8_system sys;
  for (int i = 0; i < 2; ++i) { sys.b[i] = d_b[i]; for (int j = 0; j < 2; ++j) sys.A[i][j] = d_A[i][j]; }
  CHECK(c8_assemble_forward_jacobian(ctx, &st, &sys));  /* 0 or -1 as evaluations.cpp:95-97 */
  CHECK(c8_assemble_forward_jacobian(ctx, &st, &sys));  /* accumulates: the system is now doubled */
  CHECK(c8_assemble_residual(ctx, &st, &sys));          /* b tripled */
  /* one part: halo with empty exchange lists through the host transport */
  c8_comm* comm = NULL;
  CHECK(c8_comm_create_host(0, 1, host_exchange, host_allreduce, NULL, &comm));
  int64_t const zero2[2] = {0, 0};
  c8_halo_desc hd;
  memset(&hd, 0, sizeof hd);
  hd.num_owned = nnodes; hd.num_touched = nnodes;
  hd.send_ptr = zero2; hd.recv_ptr = zero2; hd.recv_col_ptr = zero2; hd.import_ptr = zero2; hd.export_ptr = zero2;
  c8_halo* halo = NULL;
  CHECK(c8_halo_build(nnodes, rowptr, colidx, &hd, 0, 1, &halo));
  CHECK(c8_halo_attach(halo, ctx, comm));
  CHECK(c8_halo_gather(halo, &sys, C8_HALO_A | C8_HALO_B));
  double* xs[2] = {d_u, d_p};
  CHECK(c8_halo_scatter_x(halo, xs));
  double red[3] = {1.5, -2.0, 3.25};
  CHECK(c8_comm_allreduce_sum(comm, red, 3));
  if (red[0] != 1.5 || red[2] != 3.25 || c8_comm_size(comm) != 1 || c8_halo_send_bytes(halo, C8_HALO_A | C8_HALO_B) != 0) return 6;
  CHECK(c8_status(ctx));
  HIP(hipDeviceSynchronize());
  /* outputs */
  FILE* f = fopen(argv[1], "wb");
  if (!f) return 7;
  double head[8] = {(double)nnodes, (double)nelems, (double)nnz[0][0], (double)nnz[0][1], (double)nnz[1][0], (double)nnz[1][1], (double)NX, (double)NY};
  fwrite(head, sizeof(double), 8, f);
  fwrite(u, sizeof(double), nu, f);
  fwrite(p, sizeof(double), (size_t)nnodes, f);
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) {
      double* h = (double*)malloc(sizeof(double) * (size_t)nnz[i][j]);
      HIP(hipMemcpy(h, d_A[i][j], sizeof(double) * (size_t)nnz[i][j], hipMemcpyDeviceToHost));
      fwrite(h, sizeof(double), (size_t)nnz[i][j], f);
      free(h);
    }
  for (int i = 0; i < 2; ++i) {
    size_t const n = (size_t)nnodes * (i == 0 ? 3 : 1);
    double* h = (double*)malloc(sizeof(double) * n);
    HIP(hipMemcpy(h, d_b[i], sizeof(double) * n, hipMemcpyDeviceToHost));
    fwrite(h, sizeof(double), n, f);
    free(h);
  }
  {
    double* h = (double*)malloc(sizeof(double) * nxi);
    HIP(hipMemcpy(h, d_xi, sizeof(double) * nxi, hipMemcpyDeviceToHost));
    fwrite(h, sizeof(double), nxi, f);
    free(h);
  }
  fclose(f);
  c8_halo_destroy(halo);
  c8_comm_destroy(comm);
  c8_destroy(ctx);
  printf("abi_client ok: %d nodes, %d elements, build %s\n", nnodes, nelems, c8_build_info());
  return 0;
}
