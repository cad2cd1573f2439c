// c8_api_internal.hpp -- the context behind the opaque c8_ctx of include/c8.h, shared by the
// translation units that implement the C ABI.
#pragma once

#include <hip/hip_runtime_api.h>

#include <string>
#include <vector>

#include "../../include/c8.h"
#include "c8_host.hpp"
#include "c8_kernels.hpp"

int c8_fail(int code, std::string const& msg);  // records c8_last_error() and returns code

struct c8_ctx {
  // (namespace c8 types)
  c8::HostMesh mesh;
  c8::HostGraph graph;
  std::vector<int32_t> order, color_off;
  int model = c8::MODEL_NONE;
  int nloc = 0, nparams = 0, npts0 = 0;
  int ndims = 3;                      // 3, or 2 on tri3 meshes: u has ndims equations per node
  int nres = 2;                       // global residuals: 2 (`mechanics`: u, p), 1 (`mechanics_plane_stress`: u)
  c8::ModelSettings ms{};
  std::vector<double> params;
  std::vector<std::vector<int32_t>> active;
  c8::KernelSet ks{};
  // device mirrors
  int32_t* d_conn = nullptr;
  double* d_coords = nullptr;
  int32_t* d_nodeptr = nullptr;
  int32_t* d_nodeadj = nullptr;   // node-graph columns (boundary conditions, A x)
  double* d_scalar = nullptr;     // reduction result
  double* d_work[4] = {nullptr, nullptr, nullptr, nullptr};  // Newton driver: dx[2], A dx[2]
  double* d_xi_saved = nullptr;   // Newton driver: local state at the base point of a line search
  uint8_t* d_pos = nullptr;
  int32_t* d_elem_set = nullptr;
  int32_t* d_order = nullptr;
  int32_t* d_nodeelem_ptr = nullptr;  // node -> elements (staged assembly)
  int32_t* d_nodeelem = nullptr;
  double* d_stage = nullptr;          // [ring][stage_stride], allocated at the first staged assembly
  int32_t* d_node_order = nullptr;
  c8::StagePlan plan;                 // staged assembly: chunks, ring, node order
  int stage_min_chunk = 0;            // 0: automatic (c8_api.hip: stage_setup)
  // staged assembly with the row sums of chunk k beside the assembly of chunk k + 1 (c8_set_stage_overlap): the row-sum
  // launches go to sum_stream, events order the two streams both ways
  int stage_overlap = 0;
  hipStream_t sum_stream = nullptr;
  std::vector<hipEvent_t> ev_asm, ev_sum;
  // staged assembly in two parts (c8_set_gather_early_nodes): the rows of nodes [early_begin, early_end) are summed by
  // the assembly call, the other rows by c8_gather_finish
  int early_begin = 0, early_end = 0, early_count = 0;
  bool gather_pending = false;
  c8::GatherArgs pending_ga{};
  // row-per-node forward assembly (C8_KERNEL_NODE) in two parts: the fields the second part reads
  bool pending_node_rows = false, pending_adjoint = false;
  c8::FieldArgs pending_fa{};
  c8::AdjointArgs pending_aa{};
  double* d_shape = nullptr;          // cached shape tables of the wave kernels, [nelems][ks.shape_stride] (hex8; null: computed per call)
  double* d_params = nullptr;
  int32_t* d_active = nullptr;  // [nsets][10]: {grad offset, n_active, indices...}
  int* d_status = nullptr;
  unsigned long long* d_stamps = nullptr;  // -DC8_STAMPS diagnostic build only
  // objective: 0 = average displacement, 1 = calibration (c8_qoi.hip)
  int qoi_kind = 0;
  int32_t* d_cal_faces = nullptr;   // [cal_nfaces][4] node ids of the element faces on the displacement side set
  double* d_cal_S = nullptr;        // [nelems][coupled points][3] load-plane sums of grad N
  double const* d_u_meas = nullptr; // caller's measured displacement of the current step (device)
  int cal_nfaces = 0, cal_nf = 0, cal_comp = 0;
  double cal_area = 0., cal_w[3] = {1., 1., 1.}, cal_balance = 0., cal_dt_over_T = 1.;
  double cal_load_meas = 0., cal_total_load = 0., cal_load_mismatch = 0.;
  double cal_area_local = 0.;       // this part's share of the side-set area
  c8_allreduce_fn allreduce = nullptr;  // SUM over the parts (null: one part)
  void* allreduce_user = nullptr;
  int num_parts = 1;
  c8_halo* halo = nullptr;              // multi-part mesh: exchanges and reductions (c8_halo_attach), not owned
  hipStream_t stream = nullptr;
  int scatter_mode = C8_SCATTER_COLORED;
  bool scatter_auto = false;         // the mode was chosen by c8_create, not by the caller (see run() in c8_api.hip)
  int kernel_variant = C8_KERNEL_AUTO;
  int assign_mode = 0;               // staged Jacobian assemblies assign their outputs instead of adding to them
  int async = 0;
  int32_t const* subset = nullptr;   // set for the duration of a *_subset call
  int subset_count = 0;
};


// c8_qoi.hip
c8::QoiArgs c8_qoi_args(c8_ctx const* c);
int c8_qoi_prepare(c8_ctx* c, c8::FieldArgs const& fa);
int c8_qoi_surface(c8_ctx* c, double const* u, double* J, double* b0);
int c8_qoi_postprocess(c8_ctx* c, double* J);
// c8_halo.hip: multi-part helpers for the other translation units
int c8_parts_allreduce(c8_ctx* c, double* values, int n);  // SUM over the parts: caller's callback, else the halo's communicator; no-op on one part
int c8_halo_num_owned(c8_halo const* h);
void c8_halo_detach_ctx(c8_ctx* c);  // c8_destroy: the halo attached to c (if any) forgets the context and its communicator
