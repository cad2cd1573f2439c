// c8_api.hip -- implementation of the C ABI in include/c8.h.
//
// The context owns the host tables (node graph, colouring) and their device mirrors;
// every assembly call is a (colour-batched) sequence of kernel launches on the context's
// stream.  There is no CPU execution path here: without a HIP device c8_create() fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/c8.h"
#include "c8_host.hpp"
#include "c8_kernels.hpp"
#include "c8_api_internal.hpp"

using namespace c8;

thread_local std::string g_c8_last_error;
int c8_fail(int code, std::string const& msg) {
  g_c8_last_error = msg;
  return code;
}
static int fail(int code, std::string const& msg) { return c8_fail(code, msg); }
#define C8_HIP(call)                                                                               \
  do {                                                                                             \
    hipError_t err__ = (call);                                                                     \
    if (err__ != hipSuccess) return fail(C8_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(err__)); \
  } while (0)

template <class T> static int upload(T** dptr, std::vector<T> const& h) {
  if (h.empty()) { *dptr = nullptr; return C8_OK; }
  C8_HIP(hipMalloc((void**)dptr, h.size() * sizeof(T)));
  C8_HIP(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return C8_OK;
}

static int model_id(char const* name, int ndims, int* nloc, int* nparams) {
  std::string const s = name ? name : "";
  if (ndims == 2) {  // the models of the reference's 2-D decks that run on `mechanics` (2 + 1 equations per node)
    if (s == "small_J2") { *nloc = SmallJ2Plane<double>::NLOC; *nparams = SmallJ2Plane<double>::NPARAMS; return MODEL_SMALL_J2; }
    if (s == "small_hill_plane_strain") { *nloc = SmallHillPlaneStrain<double>::NLOC; *nparams = SmallHillPlaneStrain<double>::NPARAMS; return MODEL_SMALL_HILL_PLANE_STRAIN; }
    if (s == "hypo_hill_plane_strain") { *nloc = HypoHillPlaneStrain<double>::NLOC; *nparams = HypoHillPlaneStrain<double>::NPARAMS; return MODEL_HYPO_HILL_PLANE_STRAIN; }
    if (s == "hyper_J2_plane_strain") { *nloc = HyperJ2PlaneStrain<double>::NLOC; *nparams = HyperJ2PlaneStrain<double>::NPARAMS; return MODEL_HYPER_J2_PLANE_STRAIN; }
    // the models of `mechanics_plane_stress` (2 equations per node, no pressure)
    if (s == "small_hill_plane_stress") { *nloc = SmallHillPlaneStress<double>::NLOC; *nparams = SmallHillPlaneStress<double>::NPARAMS; return MODEL_SMALL_HILL_PLANE_STRESS; }
    if (s == "hyper_J2_plane_stress") { *nloc = HyperJ2PlaneStress<double>::NLOC; *nparams = HyperJ2PlaneStress<double>::NPARAMS; return MODEL_HYPER_J2_PLANE_STRESS; }
    if (s == "hypo_hill_plane_stress") { *nloc = HypoHillPlaneStress<double>::NLOC; *nparams = HypoHillPlaneStress<double>::NPARAMS; return MODEL_HYPO_HILL_PLANE_STRESS; }
    return MODEL_NONE;
  }
  if (s == "elastic") { *nloc = Elastic<double>::NLOC; *nparams = Elastic<double>::NPARAMS; return MODEL_ELASTIC; }
  if (s == "small_J2") { *nloc = SmallJ2<double>::NLOC; *nparams = SmallJ2<double>::NPARAMS; return MODEL_SMALL_J2; }
  if (s == "hyper_J2") { *nloc = HyperJ2<double>::NLOC; *nparams = HyperJ2<double>::NPARAMS; return MODEL_HYPER_J2; }
  if (s == "isotropic_elastic") { *nloc = IsotropicElastic<double>::NLOC; *nparams = IsotropicElastic<double>::NPARAMS; return MODEL_ISOTROPIC_ELASTIC; }
  if (s == "hypo_hill") { *nloc = HypoHill<double>::NLOC; *nparams = HypoHill<double>::NPARAMS; return MODEL_HYPO_HILL; }
  if (s == "small_hill") { *nloc = SmallHill<double>::NLOC; *nparams = SmallHill<double>::NPARAMS; return MODEL_SMALL_HILL; }
  if (s == "small_hosford") { *nloc = SmallHosford<double>::NLOC; *nparams = SmallHosford<double>::NPARAMS; return MODEL_SMALL_HOSFORD; }
  if (s == "hypo_hosford") { *nloc = HypoHosford<double>::NLOC; *nparams = HypoHosford<double>::NPARAMS; return MODEL_HYPO_HOSFORD; }
  if (s == "hypo_barlat") { *nloc = HypoBarlat<double>::NLOC; *nparams = HypoBarlat<double>::NPARAMS; return MODEL_HYPO_BARLAT; }
  return MODEL_NONE;
}

static void stage_release(c8_ctx* c);
static int upload_active(c8_ctx* c) {
  std::vector<int32_t> tab((size_t)c->mesh.nsets * 10, 0);
  int ofs = 0;
  for (int es = 0; es < c->mesh.nsets; ++es) {
    int const n = (int)c->active[es].size();
    tab[(size_t)es * 10 + 0] = ofs;
    tab[(size_t)es * 10 + 1] = n;
    for (int k = 0; k < n; ++k) tab[(size_t)es * 10 + 2 + k] = c->active[es][k];
    ofs += n;
  }
  if (!c->d_active) C8_HIP(hipMalloc((void**)&c->d_active, tab.size() * sizeof(int32_t)));
  C8_HIP(hipMemcpy(c->d_active, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  return C8_OK;
}

extern "C" {

const char* c8_last_error(void) { return g_c8_last_error.c_str(); }

#ifndef C8_BUILD_ID
#define C8_BUILD_ID "unknown"
#endif
#ifndef C8_BUILD_FLAGS
#define C8_BUILD_FLAGS "unknown"
#endif
const char* c8_build_info(void) { return "id=" C8_BUILD_ID " flags=" C8_BUILD_FLAGS; }

int c8_create(const c8_mesh_desc* md, const c8_model_desc* mo, c8_ctx** out) {
  if (!md || !mo || !out) return fail(C8_ERR_ARG, "c8_create: null argument");
  *out = nullptr;
  if (md->elem_type != C8_ELEM_TET4 && md->elem_type != C8_ELEM_HEX8 && md->elem_type != C8_ELEM_TRI3)
    return fail(C8_ERR_UNSUPPORTED, "c8_create: elem_type must be C8_ELEM_TRI3, C8_ELEM_TET4 or C8_ELEM_HEX8");
  int const ndims = md->elem_type == C8_ELEM_TRI3 ? 2 : 3;
  if (md->num_nodes <= 0 || md->num_elems <= 0 || md->num_elem_sets <= 0 || !md->coords || !md->conn)
    return fail(C8_ERR_ARG, "c8_create: empty mesh");
  std::string const global_type = mo->global_type ? mo->global_type : "";
  if (global_type != "mechanics" && global_type != "mechanics_plane_stress")
    return fail(C8_ERR_UNSUPPORTED, "c8_create: global residual must be 'mechanics' (mixed formulation) or 'mechanics_plane_stress'");
  int nloc = 0, nparams = 0;
  int const model = model_id(mo->local_type, ndims, &nloc, &nparams);
  if (model == MODEL_NONE) return fail(C8_ERR_UNSUPPORTED, std::string("c8_create: unknown local residual name") + (ndims == 2 ? " for a 2-D mesh: " : ": ") + (mo->local_type ? mo->local_type : "(null)"));
  // the plane-stress models carry sigma_zz = 0 and no pressure: they belong to mechanics_plane_stress and only to it
  if (model_is_plane_stress(model) != (global_type == "mechanics_plane_stress"))
    return fail(C8_ERR_UNSUPPORTED, "c8_create: 'mechanics_plane_stress' (tri3 meshes) takes the *_plane_stress local residuals, 'mechanics' the others");
  if (mo->thickness < 0.) return fail(C8_ERR_ARG, "c8_create: negative thickness");
  if (mo->num_params != nparams || !mo->params) return fail(C8_ERR_ARG, "c8_create: wrong number of material parameters for this model");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(C8_ERR_DEVICE, "c8_create: no HIP device available (this library has no CPU path)");

  c8_ctx* c = new c8_ctx();
  c->mesh.elem_type = md->elem_type;
  c->ndims = ndims;
  c->mesh.nn = md->elem_type;  // 3, 4, 8: the type ids are the node counts
  c->mesh.nnodes = md->num_nodes;
  c->mesh.nelems = md->num_elems;
  c->mesh.nsets = md->num_elem_sets;
  c->mesh.coords.assign(md->coords, md->coords + (size_t)md->num_nodes * 3);
  c->mesh.conn.assign(md->conn, md->conn + (size_t)md->num_elems * c->mesh.nn);
  if (md->elem_set) {
    c->mesh.elem_set.assign(md->elem_set, md->elem_set + md->num_elems);
    for (int32_t s : c->mesh.elem_set)
      if (s < 0 || s >= md->num_elem_sets) { delete c; return fail(C8_ERR_ARG, "c8_create: elem_set id out of range"); }
  } else if (md->num_elem_sets != 1) {
    delete c;
    return fail(C8_ERR_ARG, "c8_create: elem_set is required when num_elem_sets > 1");
  }
  if (md->num_extra_pairs > 0) {
    if (!md->extra_pairs) { delete c; return fail(C8_ERR_ARG, "c8_create: extra_pairs is null"); }
    c->mesh.extra_pairs.assign(md->extra_pairs, md->extra_pairs + (size_t)md->num_extra_pairs * 2);
  }
  std::string err = build_node_graph(c->mesh, c->graph);
  if (err.empty()) err = color_elements(c->mesh, c->order, c->color_off);
  if (!err.empty()) { delete c; return fail(C8_ERR_ARG, "c8_create: " + err); }
  c->model = model;
  c->nloc = nloc;
  c->nparams = nparams;
  c->npts0 = (md->elem_type == C8_ELEM_HEX8) ? Elem<C8_HEX8>::NP0 : (md->elem_type == C8_ELEM_TET4 ? Elem<C8_TET4>::NP0 : Elem<C8_TRI3>::NP0);
  c->ms = ModelSettings{mo->stabilization_multiplier, mo->local_abs_tol, mo->local_rel_tol, mo->local_max_iters,
                        mo->thickness > 0. ? mo->thickness : 1.};
  c->nres = model_is_plane_stress(model) ? 1 : 2;
  c->ms.closed_form = c->ms.closed_form_slot = c->ms.max_iters >= 8 ? 1 : 0;  // see c8_set_kernel_variant
  if (mo->ls_max_evals < 0 || mo->ls_sufficient_decrease < 0. || mo->ls_min_backtrack < 0. || mo->ls_max_backtrack < 0.) {
    delete c;
    return fail(C8_ERR_ARG, "c8_create: negative line-search setting");
  }
  if (mo->ls_sufficient_decrease > 0.) c->ms.ls_c1 = mo->ls_sufficient_decrease;  // 0: the defaults of line_search.hpp:28-35
  if (mo->ls_min_backtrack > 0.) c->ms.ls_bmin = mo->ls_min_backtrack;
  if (mo->ls_max_backtrack > 0.) c->ms.ls_bmax = mo->ls_max_backtrack;
  if (mo->ls_max_evals > 0) c->ms.ls_max_evals = mo->ls_max_evals;
  c->params.assign(mo->params, mo->params + (size_t)md->num_elem_sets * nparams);
  c->active.assign(md->num_elem_sets, std::vector<int32_t>());
  c->active[0].push_back(0);  // default: E of element set 0 (small_J2.cpp:96-98)
  c->ks = get_kernels(md->elem_type, model);
  int rc = C8_OK;
  if ((rc = upload(&c->d_conn, c->mesh.conn)) || (rc = upload(&c->d_coords, c->mesh.coords)) ||
      (rc = upload(&c->d_nodeptr, c->graph.nodeptr)) || (rc = upload(&c->d_nodeadj, c->graph.nodeadj)) || (rc = upload(&c->d_pos, c->graph.pos)) ||
      (rc = upload(&c->d_elem_set, c->mesh.elem_set)) || (rc = upload(&c->d_order, c->order)) ||
      (rc = upload(&c->d_nodeelem_ptr, c->graph.nodeelem_ptr)) || (rc = upload(&c->d_nodeelem, c->graph.nodeelem)) ||
      (rc = upload(&c->d_params, c->params)) || (rc = upload_active(c))) {
    c8_destroy(c);
    return rc;
  }
  if (hipMalloc((void**)&c->d_scalar, sizeof(double)) != hipSuccess || hipMalloc((void**)&c->d_status, sizeof(int)) != hipSuccess || hipMemset(c->d_status, 0, sizeof(int)) != hipSuccess) {
    c8_destroy(c);
    return fail(C8_ERR_DEVICE, "c8_create: status allocation failed");
  }
  // cached shape tables of the wave kernels; if they do not fit, the kernels compute them per call (same values)
  (void)c8_set_shape_cache(c, 1);
  // default scatter mode: the staged assembly (fastest, reproducible) where the node degrees allow it; its stage is
  // allocated at the first Jacobian assembly and, should that fail, the context falls back to colour batches
  c->scatter_auto = true;
  c->scatter_mode = (c->ks.can_stage && c->graph.max_degree <= c8::GATHER_MAX_DEGREE) ? C8_SCATTER_GATHER : C8_SCATTER_COLORED;
  g_c8_last_error.clear();
  *out = c;
  return C8_OK;
}

// The geometry of a context is static: dN/dx, w dv and the element size of every element are computed once and read
// back by the wave kernels (1.7 KB per hex8 element) instead of being recomputed by every call.  on = 0 frees the
// tables (the kernels then compute them per call, same values); element types without wave kernels have none.
int c8_set_shape_cache(c8_ctx* c, int on) {
  if (!c) return fail(C8_ERR_ARG, "c8_set_shape_cache: null ctx");
  if (!on || !c->ks.shape_tables) {
    (void)hipFree(c->d_shape);
    c->d_shape = nullptr;
    return C8_OK;
  }
  if (c->d_shape) return C8_OK;
  size_t const bytes = (size_t)c->mesh.nelems * c->ks.shape_stride * sizeof(double);
  if (bytes == 0) return C8_OK;
  if (hipMalloc((void**)&c->d_shape, bytes) != hipSuccess) {
    c->d_shape = nullptr;
    return fail(C8_ERR_DEVICE, "c8_set_shape_cache: cannot allocate the shape tables (c8_set_shape_cache(ctx, 0) runs without them)");
  }
  MeshTables const mt{c->d_conn, c->d_coords, c->d_nodeptr, c->d_pos, c->d_elem_set, nullptr, c->d_params};
  if (c->ks.shape_tables(mt, c->d_shape, c->mesh.nelems, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
    (void)hipFree(c->d_shape);
    c->d_shape = nullptr;
    return fail(C8_ERR_DEVICE, "c8_set_shape_cache: shape-table kernel failed");
  }
  return C8_OK;
}

void c8_destroy(c8_ctx* c) {
  if (!c) return;
  c8_halo_detach_ctx(c);  // an attached halo outlives the context detached (c8_halo_destroy must not touch freed memory)
  stage_release(c);
  for (hipEvent_t e : c->ev_asm) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev_sum) (void)hipEventDestroy(e);
  if (c->sum_stream) (void)hipStreamDestroy(c->sum_stream);
  void* bufs[] = {c->d_shape, c->d_cal_faces, c->d_cal_S, c->d_nodeelem_ptr, c->d_nodeelem, c->d_nodeadj, c->d_scalar, c->d_xi_saved, c->d_work[0], c->d_work[1], c->d_work[2], c->d_work[3], c->d_conn, c->d_coords, c->d_nodeptr, c->d_pos, c->d_elem_set, c->d_order, c->d_params, c->d_active, c->d_status};
  for (void* b : bufs) (void)hipFree(b);
  delete c;
}

int c8_num_local_dofs(const c8_ctx* c) { return c ? c->nloc : C8_ERR_ARG; }
int c8_num_dims(const c8_ctx* c) { return c ? c->ndims : C8_ERR_ARG; }
int c8_num_residuals(const c8_ctx* c) { return c ? c->nres : C8_ERR_ARG; }
int c8_num_local_points(const c8_ctx* c) { return c ? c->npts0 : C8_ERR_ARG; }
int c8_num_colors(const c8_ctx* c) { return c ? (int)c->color_off.size() - 1 : C8_ERR_ARG; }
int64_t c8_graph_nnz(const c8_ctx* c, int i, int j) {
  if (!c || i < 0 || i > 1 || j < 0 || j > 1) return C8_ERR_ARG;
  return block_nnz(c->graph, c->mesh.nnodes, i, j, c->ndims);
}
int c8_graph(const c8_ctx* c, int i, int j, int64_t* rowptr, int32_t* colidx) {
  if (!c || i < 0 || i > 1 || j < 0 || j > 1 || !rowptr || !colidx) return fail(C8_ERR_ARG, "c8_graph: bad argument");
  block_csr(c->graph, c->mesh.nnodes, i, j, rowptr, colidx, c->ndims);
  return C8_OK;
}
int c8_init_variables(const c8_ctx* c, double* xi) {
  if (!c || !xi) return fail(C8_ERR_ARG, "c8_init_variables: bad argument");
  size_t const npt = (size_t)c->mesh.nelems * c->npts0;
  for (size_t q = 0; q < npt; ++q) {
    double* x = xi + q * c->nloc;
    switch (c->model) {  // init_variables_impl of each model
      case MODEL_ELASTIC: Elastic<double>::init_variables(x); break;
      case MODEL_SMALL_J2: if (c->ndims == 2) SmallJ2Plane<double>::init_variables(x); else SmallJ2<double>::init_variables(x); break;
      case MODEL_SMALL_HILL_PLANE_STRAIN: SmallHillPlaneStrain<double>::init_variables(x); break;
      case MODEL_HYPER_J2_PLANE_STRAIN: HyperJ2PlaneStrain<double>::init_variables(x); break;
      case MODEL_HYPO_HILL_PLANE_STRAIN: HypoHillPlaneStrain<double>::init_variables(x); break;
      case MODEL_SMALL_HILL_PLANE_STRESS: SmallHillPlaneStress<double>::init_variables(x); break;
      case MODEL_HYPER_J2_PLANE_STRESS: HyperJ2PlaneStress<double>::init_variables(x); break;
      case MODEL_HYPO_HILL_PLANE_STRESS: HypoHillPlaneStress<double>::init_variables(x); break;
      case MODEL_HYPER_J2: HyperJ2<double>::init_variables(x); break;
      case MODEL_SMALL_HILL: SmallHill<double>::init_variables(x); break;
      case MODEL_ISOTROPIC_ELASTIC: IsotropicElastic<double>::init_variables(x); break;
      case MODEL_HYPO_HILL: HypoHill<double>::init_variables(x); break;
      case MODEL_SMALL_HOSFORD: SmallHosford<double>::init_variables(x); break;
      case MODEL_HYPO_HOSFORD: HypoHosford<double>::init_variables(x); break;
      case MODEL_HYPO_BARLAT: HypoBarlat<double>::init_variables(x); break;
      default: return fail(C8_ERR_UNSUPPORTED, "c8_init_variables: unknown model");
    }
  }
  return C8_OK;
}

int c8_set_params(c8_ctx* c, const double* params) {
  if (!c || !params) return fail(C8_ERR_ARG, "c8_set_params: bad argument");
  c->params.assign(params, params + c->params.size());
  C8_HIP(hipMemcpyAsync(c->d_params, c->params.data(), c->params.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  C8_HIP(hipStreamSynchronize(c->stream));
  return C8_OK;
}
int c8_set_active_params(c8_ctx* c, int es, int n, const int32_t* idx) {
  if (!c || es < 0 || es >= c->mesh.nsets || n < 0 || (n > 0 && !idx)) return fail(C8_ERR_ARG, "c8_set_active_params: bad argument");
  for (int k = 0; k < n; ++k)
    if (idx[k] < 0 || idx[k] >= c->nparams) return fail(C8_ERR_ARG, "c8_set_active_params: parameter index out of range");
  if (n > 8 || (c->nres == 1 && n > 6))  // one lane of an element's group per active parameter (six lanes under mechanics_plane_stress)
    return fail(C8_ERR_ARG, "c8_set_active_params: at most 8 active parameters per element set (6 under mechanics_plane_stress)");
  c->active[es].assign(idx, idx + n);
  return upload_active(c);
}
int c8_num_active_params(const c8_ctx* c) {
  if (!c) return C8_ERR_ARG;
  int n = 0;
  for (auto const& a : c->active) n += (int)a.size();
  return n;
}
int c8_set_stream(c8_ctx* c, void* s) {
  if (!c) return fail(C8_ERR_ARG, "c8_set_stream: null ctx");
  c->stream = (hipStream_t)s;
  return C8_OK;
}
int c8_set_scatter_mode(c8_ctx* c, int mode) {
  if (!c || (mode != C8_SCATTER_ATOMIC && mode != C8_SCATTER_COLORED && mode != C8_SCATTER_GATHER)) return fail(C8_ERR_ARG, "c8_set_scatter_mode: bad argument");
  if (mode == C8_SCATTER_GATHER) {
    if (!c->ks.can_stage) return fail(C8_ERR_UNSUPPORTED, "c8_set_scatter_mode: staged (gather) assembly is built for 3-D elements");
    if (c->graph.max_degree > c8::GATHER_MAX_DEGREE) return fail(C8_ERR_UNSUPPORTED, "c8_set_scatter_mode: node degree too large for staged (gather) assembly");
  }
  if (c->gather_pending) return fail(C8_ERR_ARG, "c8_set_scatter_mode: a staged assembly is waiting for c8_gather_finish");
  c->scatter_mode = mode;
  c->scatter_auto = false;
  return C8_OK;
}
int c8_get_scatter_mode(const c8_ctx* c) { return c ? c->scatter_mode : C8_ERR_ARG; }
int c8_set_stage_chunk(c8_ctx* c, int min_chunk) {
  if (!c || min_chunk < 1) return fail(C8_ERR_ARG, "c8_set_stage_chunk: bad argument");
  C8_HIP(hipDeviceSynchronize());
  stage_release(c);  // the plan is rebuilt at the next staged assembly
  c->stage_min_chunk = min_chunk;
  return C8_OK;
}
int c8_set_stage_overlap(c8_ctx* c, int on) {
  if (!c) return fail(C8_ERR_ARG, "c8_set_stage_overlap: null ctx");
  if (c->gather_pending) return fail(C8_ERR_ARG, "c8_set_stage_overlap: a staged assembly is waiting for c8_gather_finish");
  C8_HIP(hipDeviceSynchronize());
  c->stage_overlap = on ? 1 : 0;
  return C8_OK;
}
int c8_set_assign_mode(c8_ctx* c, int on) {
  if (!c) return fail(C8_ERR_ARG, "c8_set_assign_mode: null ctx");
  if (c->gather_pending) return fail(C8_ERR_ARG, "c8_set_assign_mode: a staged assembly is waiting for c8_gather_finish");
  c->assign_mode = on ? 1 : 0;
  return C8_OK;
}
int c8_set_gather_early_nodes(c8_ctx* c, int node_begin, int node_end) {
  if (!c || node_begin < 0 || node_end < node_begin || node_end > c->mesh.nnodes) return fail(C8_ERR_ARG, "c8_set_gather_early_nodes: bad argument");
  if (c->gather_pending) return fail(C8_ERR_ARG, "c8_set_gather_early_nodes: a staged assembly is waiting for c8_gather_finish");
  C8_HIP(hipDeviceSynchronize());
  stage_release(c);  // the node order is rebuilt at the next staged assembly
  c->early_begin = node_begin;
  c->early_end = node_end;
  return C8_OK;
}
int c8_gather_finish(c8_ctx* c) {
  if (!c) return fail(C8_ERR_ARG, "c8_gather_finish: null ctx");
  if (!c->gather_pending) return C8_OK;
  c->gather_pending = false;
  if (c->pending_node_rows) {
    c->pending_node_rows = false;
    MeshTables const mt{c->d_conn, c->d_coords, c->d_nodeptr, c->d_pos, c->d_elem_set, nullptr, c->d_params, c->d_shape};
    AdjointArgs const* const paa = c->pending_adjoint ? &c->pending_aa : nullptr;
    C8_HIP(c->ks.node_rows(mt, c->ms, c->pending_fa, paa, c->pending_ga, 0, c->early_begin, c->graph.max_degree, c->graph.max_node_elems, c->stream));
    C8_HIP(c->ks.node_rows(mt, c->ms, c->pending_fa, paa, c->pending_ga, c->early_end, c->mesh.nnodes - c->early_end, c->graph.max_degree, c->graph.max_node_elems, c->stream));
    if (paa && paa->qoi.c_load != 0.)
      C8_HIP(c->ks.node_rows(mt, c->ms, c->pending_fa, paa, c->pending_ga, -1, c->mesh.nelems, c->graph.max_degree, c->graph.max_node_elems, c->stream));
    return C8_OK;
  }
  int const total = (int)c->plan.node_order.size();
  C8_HIP(c->ks.gather_rows(c->pending_ga, c->early_count, total - c->early_count, c->graph.max_degree, c->stream));
  if (c->async) return C8_OK;
  return c8_status(c);
}
int c8_set_kernel_variant(c8_ctx* c, int variant) {
  if (!c || variant < C8_KERNEL_AUTO || variant > C8_KERNEL_NODE) return fail(C8_ERR_ARG, "c8_set_kernel_variant: bad argument");
  if ((variant == C8_KERNEL_WAVE || variant == C8_KERNEL_WAVE_AD) && !c->ks.forward_jacobian_wave)
    return fail(C8_ERR_UNSUPPORTED, "c8_set_kernel_variant: the wave-per-element kernels exist for hex8 elements and the models without a local line search");
  if (variant == C8_KERNEL_NODE && (!c->ks.node_rows || c->ms.max_iters < 8))
    return fail(C8_ERR_UNSUPPORTED, "c8_set_kernel_variant: the row-per-node kernel exists for hex8 elements and models with a closed form (small_J2), with local_max_iters >= 8");
  if (c->gather_pending) return fail(C8_ERR_ARG, "c8_set_kernel_variant: an assembly is waiting for c8_gather_finish");
  c->kernel_variant = variant;
  // a model's closed form (small_J2) runs in the forward wave kernel unless the caller asks for the iterated AD form or
  // gives the local Newton iteration a budget in which it may fail: the failure (-1) is the iterated form's to report
  c->ms.closed_form = (variant != C8_KERNEL_WAVE_AD && c->ms.max_iters >= 8) ? 1 : 0;
  c->ms.closed_form_slot = (variant == C8_KERNEL_AUTO && c->ms.max_iters >= 8) ? 1 : 0;  // an explicit C8_KERNEL_SLOT iterates
  return C8_OK;
}
#ifdef C8_STAMPS
int c8_debug_stamps(c8_ctx* c, unsigned long long* out) {  // diagnostic build only, not in c8.h
  C8_HIP(hipMemcpy(out, c->d_stamps, 4096 * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return C8_OK;
}
#endif
int c8_set_async(c8_ctx* c, int async) {
  if (!c) return fail(C8_ERR_ARG, "c8_set_async: null ctx");
  c->async = async ? 1 : 0;
  return C8_OK;
}
int c8_status(c8_ctx* c) {
  if (!c) return fail(C8_ERR_ARG, "c8_status: null ctx");
  int h = 0;
  C8_HIP(hipMemcpyAsync(&h, c->d_status, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  C8_HIP(hipStreamSynchronize(c->stream));
  if (h != 0) {
    C8_HIP(hipMemsetAsync(c->d_status, 0, sizeof(int), c->stream));
    return fail(C8_LOCAL_SOLVE_FAILED, "a local constitutive Newton solve did not converge");
  }
  return C8_OK;
}

}  // extern "C"

static MeshTables tables(c8_ctx const* c, bool colored) {
  return MeshTables{c->d_conn, c->d_coords, c->d_nodeptr, c->d_pos, c->d_elem_set, colored ? c->d_order : nullptr, c->d_params, c->d_shape};
}

// Staged (gather) assembly on the caller's stream: chunk k of the elements is assembled into the stage ring,
// then the rows of the nodes whose last element lies in chunk k are summed.  A node's elements lie in at most
// two consecutive chunks (StagePlan), so a ring of three chunks is enough; chunking only bounds the size of the
// stage (8.4 KB per element), the two kernels of a chunk run one after the other (the assembly kernel fills
// the register file and LDS of every CU, so the row sums cannot run beside it).
static int stage_setup(c8_ctx* c) {
  if (c->d_stage) return C8_OK;
  // default (stage_min_chunk == 0): one chunk while the whole stage stays under 12 GB (a chunk boundary drains the
  // GPU: 12.4 ms in one chunk against 12.7 ms in eight on a million hex8 elements), else chunks of 262144 elements
  int min_chunk = c->stage_min_chunk;
  if (min_chunk <= 0)
    min_chunk = ((double)c->mesh.nelems * c->ks.stage_stride * sizeof(double) <= 12e9) ? c->mesh.nelems : 262144;
  plan_staged_assembly(c->mesh, c->graph, min_chunk, min_chunk >= 256 ? 256 : 4, c->plan);
  c->early_count = 0;
  if (c->early_end > c->early_begin) {  // the early nodes' rows first (stable: both parts keep ascending node order)
    if (c->plan.nchunks != 1) return fail(C8_ERR_UNSUPPORTED, "staged assembly in two parts needs the whole mesh in one staged chunk");
    auto const early = [&](int32_t n) { return n >= c->early_begin && n < c->early_end; };
    c->early_count = (int)(std::stable_partition(c->plan.node_order.begin(), c->plan.node_order.end(), early) - c->plan.node_order.begin());
  }
  size_t const bytes = (size_t)c->plan.ring * c->ks.stage_stride * sizeof(double);
  if (hipMalloc((void**)&c->d_stage, bytes) != hipSuccess) return fail(C8_ERR_DEVICE, "staged assembly: cannot allocate the element stage");
  return upload(&c->d_node_order, c->plan.node_order);
}
static void stage_release(c8_ctx* c) {
  (void)hipFree(c->d_stage);
  (void)hipFree(c->d_node_order);
  c->d_stage = nullptr;
  c->d_node_order = nullptr;
}

// Row-per-node forward assembly (c8_assemble_node.hpp): whether this call takes it, and its launches.
static bool node_rows_applies(c8_ctx const* c, FieldArgs const& fa) {
  return c->ks.node_rows && c->ms.closed_form && c->d_shape && fa.xi != fa.xi_prev && c->graph.max_degree <= c8::GATHER_MAX_DEGREE &&
         (c->kernel_variant == C8_KERNEL_AUTO || c->kernel_variant == C8_KERNEL_NODE);
}
// nodes [0, nnodes) in one launch; with an early node range set, that range now and the two ranges around it in
// c8_gather_finish (the kernel takes a contiguous range of node numbers: no order table between the launch and the node)
static int run_node_rows(c8_ctx* c, FieldArgs const& fa, SystemArgs const& sa, AdjointArgs const* aa = nullptr) {
  if (c->gather_pending) return fail(C8_ERR_ARG, "row-per-node assembly: c8_gather_finish has not been called for the previous assembly");
  GatherArgs ga{c->d_nodeptr, c->d_pos, c->d_nodeelem_ptr, c->d_nodeelem, nullptr, 0, nullptr,
                {{sa.A[0][0], sa.A[0][1]}, {sa.A[1][0], sa.A[1][1]}}, {sa.b[0], sa.b[1]}, c->assign_mode};
  MeshTables const mt{c->d_conn, c->d_coords, c->d_nodeptr, c->d_pos, c->d_elem_set, nullptr, c->d_params, c->d_shape};
#ifdef C8_STAMPS
  if (!c->d_stamps) C8_HIP(hipMalloc((void**)&c->d_stamps, 4096 * 16 * sizeof(unsigned long long)));
  ga.stamps = c->d_stamps;
#endif
  if (c->early_end > c->early_begin) {  // two parts: the early rows now, the rest in c8_gather_finish
    C8_HIP(c->ks.node_rows(mt, c->ms, fa, aa, ga, c->early_begin, c->early_end - c->early_begin, c->graph.max_degree, c->graph.max_node_elems, c->stream));
    c->pending_ga = ga;
    c->pending_fa = fa;
    c->pending_adjoint = aa != nullptr;
    if (aa) c->pending_aa = *aa;
    c->pending_node_rows = true;
    c->gather_pending = true;
    return C8_OK;  // the closed form has no failing local solve: nothing to read back
  }
  C8_HIP(c->ks.node_rows(mt, c->ms, fa, aa, ga, 0, c->mesh.nnodes, c->graph.max_degree, c->graph.max_node_elems, c->stream));
  if (aa && aa->qoi.c_load != 0.)  // calibration objective: g -= dJ/dxi once every wavefront has read the old g
    C8_HIP(c->ks.node_rows(mt, c->ms, fa, aa, ga, -1, c->mesh.nelems, c->graph.max_degree, c->graph.max_node_elems, c->stream));
  return C8_OK;
}

static MeshTables tables(c8_ctx const* c, bool colored);
static int run_staged(c8_ctx* c, LaunchFn fn, FieldArgs const& fa, AdjointArgs const& aa, SystemArgs sa) {
  int rc = stage_setup(c);
  if (rc) return rc;
  StagePlan const& pl = c->plan;
  sa.status = c->d_status;
  sa.atomic = 0;
  sa.stage = c->d_stage;
  sa.stage_ring = pl.ring;
  GatherArgs ga{c->d_nodeptr, c->d_pos, c->d_nodeelem_ptr, c->d_nodeelem, c->d_stage, pl.ring, c->d_node_order,
                {{sa.A[0][0], sa.A[0][1]}, {sa.A[1][0], sa.A[1][1]}}, {sa.b[0], sa.b[1]}, c->assign_mode};
  if (c->gather_pending) return fail(C8_ERR_ARG, "staged assembly: c8_gather_finish has not been called for the previous assembly");
  if (c->early_end > c->early_begin) {  // two parts: the early rows now, the rest in c8_gather_finish
    LaunchArgs a{tables(c, false), c->ms, fa, aa, sa, 0, c->mesh.nelems, c->stream};
    C8_HIP(fn(a));
    C8_HIP(c->ks.gather_rows(ga, 0, c->early_count, c->graph.max_degree, c->stream));
    c->pending_ga = ga;
    c->pending_node_rows = false;
    c->gather_pending = true;
    if (c->async) return C8_OK;
    return c8_status(c);
  }
  if (c->stage_overlap && pl.nchunks > 1) {
    // the row sums of chunk k (bound by HBM) run on a second stream beside the assembly of chunk k + 1 (bound by
    // instruction issue).  Order: row sums k after assembly k; assembly k after row sums k - 2, whose ring slot it
    // overwrites (row sums k read the slots of chunks k - 1 and k); the caller's stream continues after the last row sums.
    if (!c->sum_stream) {  // highest priority: the row-sum waves take the slots the assembly waves free as they retire
      int lo = 0, hi = 0;
      C8_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      char const* pr = getenv("C8_SUM_STREAM_PRIORITY");
      C8_HIP(hipStreamCreateWithPriority(&c->sum_stream, hipStreamNonBlocking, pr ? atoi(pr) : hi));
    }
    while ((int)c->ev_asm.size() < pl.nchunks) {
      hipEvent_t e0, e1;
      C8_HIP(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
      C8_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
      c->ev_asm.push_back(e0);
      c->ev_sum.push_back(e1);
    }
    for (int k = 0; k < pl.nchunks; ++k) {
      if (k >= 2) C8_HIP(hipStreamWaitEvent(c->stream, c->ev_sum[k - 2], 0));
      LaunchArgs a{tables(c, false), c->ms, fa, aa, sa, k * pl.chunk, std::min(pl.chunk, c->mesh.nelems - k * pl.chunk), c->stream};
      C8_HIP(fn(a));
      C8_HIP(hipEventRecord(c->ev_asm[k], c->stream));
      C8_HIP(hipStreamWaitEvent(c->sum_stream, c->ev_asm[k], 0));
      C8_HIP(c->ks.gather_rows(ga, pl.node_off[k], pl.node_off[k + 1] - pl.node_off[k], c->graph.max_degree, c->sum_stream));
      C8_HIP(hipEventRecord(c->ev_sum[k], c->sum_stream));
    }
    C8_HIP(hipStreamWaitEvent(c->stream, c->ev_sum[pl.nchunks - 1], 0));
  } else
  for (int k = 0; k < pl.nchunks; ++k) {
    LaunchArgs a{tables(c, false), c->ms, fa, aa, sa, k * pl.chunk, std::min(pl.chunk, c->mesh.nelems - k * pl.chunk), c->stream};
    C8_HIP(fn(a));
    C8_HIP(c->ks.gather_rows(ga, pl.node_off[k], pl.node_off[k + 1] - pl.node_off[k], c->graph.max_degree, c->stream));
  }
  if (c->async) return C8_OK;
  return c8_status(c);
}

// run one launcher over the whole mesh: one launch per colour, or one atomic launch.
// `scatters` = the kernel adds into shared A/b entries (needs colouring or atomics).
static int run(c8_ctx* c, LaunchFn fn, FieldArgs const& fa, AdjointArgs const& aa, SystemArgs sa, bool scatters, char const* what) {
  if (!fn) return fail(C8_ERR_UNSUPPORTED, std::string(what) + ": not available for this element/model");
  // staged assembly: the two Jacobian assemblies; everything else (residual-only assembly: NDOF adds per element)
  // keeps atomic adds
  bool const staged = scatters && c->scatter_mode == C8_SCATTER_GATHER && sa.A[0][0] && !c->subset &&
                      (fn == c->ks.forward_jacobian_wave || fn == c->ks.adjoint_jacobian_wave || fn == c->ks.forward_jacobian ||
                       (fn == c->ks.adjoint_jacobian && c->ks.adjoint_slot_stages));
  // a Jacobian kernel that cannot stage (hex8 slot-per-lane adjoint kernel; element subsets): colour batches / atomic
  // adds for this call while the context is in its DEFAULT mode, an error when the caller asked for GATHER
  bool const unstaged_default = scatters && c->scatter_mode == C8_SCATTER_GATHER && sa.A[0][0] && !staged && c->scatter_auto &&
                                !c->assign_mode;
  bool const colored = scatters && !c->subset && (c->scatter_mode == C8_SCATTER_COLORED || unstaged_default);
  sa.status = c->d_status;
  sa.atomic = colored ? 0 : 1;
#ifdef C8_STAMPS
  if (!c->d_stamps) C8_HIP(hipMalloc((void**)&c->d_stamps, 4096 * 16 * sizeof(unsigned long long)));
  sa.stamps = c->d_stamps;
#endif
  if (c->subset && scatters && c->scatter_mode != C8_SCATTER_ATOMIC && !c->scatter_auto)
    return fail(C8_ERR_ARG, std::string(what) + ": element subsets need C8_SCATTER_ATOMIC");
  if (scatters && c->scatter_mode == C8_SCATTER_GATHER && sa.A[0][0] && !staged && !unstaged_default)
    return fail(C8_ERR_UNSUPPORTED, std::string(what) + ": staged (gather) assembly of hex8 adjoint Jacobians needs the wave-per-element kernel");
  if (c->assign_mode && scatters && sa.A[0][0] && !staged)
    return fail(C8_ERR_UNSUPPORTED, std::string(what) + ": assign mode (c8_set_assign_mode) needs the staged Jacobian assembly (C8_SCATTER_GATHER)");
  if (c->kernel_variant == C8_KERNEL_NODE && fn == c->ks.forward_jacobian_wave && !(staged && node_rows_applies(c, fa)))
    return fail(C8_ERR_UNSUPPORTED, std::string(what) + ": C8_KERNEL_NODE needs C8_SCATTER_GATHER, the shape-table cache and distinct xi / xi_prev arrays");
  if (staged && fn == c->ks.forward_jacobian_wave && node_rows_applies(c, fa)) return run_node_rows(c, fa, sa);
  // the adjoint assembly in the same form (both objectives: the point integrands have closed derivatives)
  if (staged && fn == c->ks.adjoint_jacobian_wave && node_rows_applies(c, fa)) return run_node_rows(c, fa, sa, &aa);
  if (staged) {
    int const rc = run_staged(c, fn, fa, aa, sa);
    if (rc == C8_ERR_DEVICE && c->scatter_auto && !c->d_stage && !c->assign_mode && c->early_end <= c->early_begin) {
      // the default mode could not get its stage: colour batches from here on (noted in c8_last_error)
      (void)hipGetLastError();
      c->scatter_mode = C8_SCATTER_COLORED;
      c->scatter_auto = false;
      int const rc2 = run(c, fn, fa, aa, sa, scatters, what);
      if (rc2 == C8_OK) g_c8_last_error = "note: the element stage of C8_SCATTER_GATHER could not be allocated; this context now assembles in C8_SCATTER_COLORED mode";
      return rc2;
    }
    return rc;
  }
  LaunchArgs a{tables(c, colored), c->ms, fa, aa, sa, 0, 0, c->stream};
  if (c->subset) {
    a.mt.order = c->subset;
    a.count = c->subset_count;
    C8_HIP(fn(a));
  } else if (colored) {
    int const nc = (int)c->color_off.size() - 1;
    for (int k = 0; k < nc; ++k) {
      a.first = c->color_off[k];
      a.count = c->color_off[k + 1] - c->color_off[k];
      C8_HIP(fn(a));
    }
  } else {
    a.first = 0;
    a.count = c->mesh.nelems;
    C8_HIP(fn(a));
  }
  if (c->async) return C8_OK;
  return c8_status(c);
}

// under mechanics_plane_stress (one residual) the entries [1] of a state, a system and an adjoint vector are ignored
static bool check_state(const c8_ctx* c, const c8_state* st) {
  if (!st || !st->xi_prev || !st->xi) return false;
  for (int i = 0; i < c->nres; ++i)
    if (!st->x[i] || !st->x_prev[i]) return false;
  return true;
}
static bool check_z(const c8_ctx* c, const double* const z[2]) { return z && z[0] && (c->nres == 1 || z[1]); }
static FieldArgs field_args(const c8_state* st) {
  return FieldArgs{st->x[0], st->x[1], st->x_prev[0], st->x_prev[1], st->xi_prev, st->xi};
}

extern "C" {

int c8_assemble_forward_jacobian(c8_ctx* c, const c8_state* st, const c8_system* sys) {
  if (!c || !check_state(c, st) || !sys) return fail(C8_ERR_ARG, "c8_assemble_forward_jacobian: null argument");
  for (int i = 0; i < c->nres; ++i) {
    if (!sys->b[i]) return fail(C8_ERR_ARG, "c8_assemble_forward_jacobian: null b");
    for (int j = 0; j < c->nres; ++j)
      if (!sys->A[i][j]) return fail(C8_ERR_ARG, "c8_assemble_forward_jacobian: null A block");
  }
  SystemArgs sa{{{sys->A[0][0], sys->A[0][1]}, {sys->A[1][0], sys->A[1][1]}}, {sys->b[0], sys->b[1]}, nullptr, 0};
  LaunchFn fn = c->ks.forward_jacobian;
  if (c->ks.forward_jacobian_wave && c->kernel_variant != C8_KERNEL_SLOT) fn = c->ks.forward_jacobian_wave;
  return run(c, fn, field_args(st), AdjointArgs{}, sa, true, "c8_assemble_forward_jacobian");
}

int c8_assemble_forward_jacobian_subset(c8_ctx* c, const c8_state* st, const c8_system* sys, const int32_t* elems, int count) {
  if (!c || count < 0 || (count > 0 && !elems)) return fail(C8_ERR_ARG, "c8_assemble_forward_jacobian_subset: bad argument");
  if (count == 0) return C8_OK;
  c->subset = elems;
  c->subset_count = count;
  int const rc = c8_assemble_forward_jacobian(c, st, sys);
  c->subset = nullptr;
  c->subset_count = 0;
  return rc;
}

int c8_assemble_residual(c8_ctx* c, const c8_state* st, const c8_system* sys) {
  if (!c || !check_state(c, st) || !sys || !sys->b[0] || (c->nres == 2 && !sys->b[1])) return fail(C8_ERR_ARG, "c8_assemble_residual: null argument");
  SystemArgs sa{{{nullptr, nullptr}, {nullptr, nullptr}}, {sys->b[0], sys->b[1]}, nullptr, 0};
  // hex8, natural element order with atomic adds (whole mesh, not colour-batched): eight elements per wavefront
  if (c->ks.residual_wave && c->kernel_variant != C8_KERNEL_SLOT && c->scatter_mode != C8_SCATTER_COLORED && !c->subset) {
    sa.status = c->d_status;
    sa.atomic = 1;
    LaunchArgs a{tables(c, false), c->ms, field_args(st), AdjointArgs{}, sa, 0, c->mesh.nelems, c->stream};
    C8_HIP(c->ks.residual_wave(a));
    return c->async ? C8_OK : c8_status(c);
  }
  return run(c, c->ks.residual, field_args(st), AdjointArgs{}, sa, true, "c8_assemble_residual");
}

int c8_assemble_adjoint_jacobian(c8_ctx* c, const c8_state* st, double* g, const double* f, const c8_system* sys) {
  if (!c || !check_state(c, st) || !sys || !g || !f) return fail(C8_ERR_ARG, "c8_assemble_adjoint_jacobian: null argument");
  for (int i = 0; i < c->nres; ++i) {
    if (!sys->b[i]) return fail(C8_ERR_ARG, "c8_assemble_adjoint_jacobian: null b");
    for (int j = 0; j < c->nres; ++j)
      if (!sys->A[i][j]) return fail(C8_ERR_ARG, "c8_assemble_adjoint_jacobian: null A block");
  }
  SystemArgs sa{{{sys->A[0][0], sys->A[0][1]}, {sys->A[1][0], sys->A[1][1]}}, {sys->b[0], sys->b[1]}, nullptr, 0};
  // the face term of the calibration objective is added to b after the row sums: with assign mode AND the row sums in
  // two parts, the second part would overwrite it
  if (c->assign_mode && c->qoi_kind == 1 && c->early_end > c->early_begin && c->scatter_mode == C8_SCATTER_GATHER)
    return fail(C8_ERR_UNSUPPORTED, "c8_assemble_adjoint_jacobian: assign mode with two-part row sums and the calibration objective");
  int rc = c8_qoi_prepare(c, field_args(st));  // preprocess_qoi (evaluations.cpp:365)
  if (rc) return rc;
  AdjointArgs aa{g, const_cast<double*>(f), nullptr, nullptr, nullptr, nullptr, c->d_active, c8_qoi_args(c)};
  LaunchFn fn = c->ks.adjoint_jacobian;
  if (c->ks.adjoint_jacobian_wave && c->kernel_variant != C8_KERNEL_SLOT) fn = c->ks.adjoint_jacobian_wave;
  int const async = c->async;
  c->async = 1;  // the face term of the objective is enqueued before the status is read back
  rc = run(c, fn, field_args(st), aa, sa, true, "c8_assemble_adjoint_jacobian");
  c->async = async;
  if (rc) return rc;
  if ((rc = c8_qoi_surface(c, st->x[0], nullptr, sys->b[0]))) return rc;
  return async ? C8_OK : c8_status(c);
}

int c8_solve_adjoint_local(c8_ctx* c, const c8_state* st, const double* const z[2], double* phi, double* g, double* f) {
  if (!c || !check_state(c, st) || !check_z(c, z) || !phi || !g || !f) return fail(C8_ERR_ARG, "c8_solve_adjoint_local: null argument");
  AdjointArgs aa{g, f, z[0], z[1], phi, nullptr, c->d_active, c8_qoi_args(c)};
  LaunchFn fn = c->ks.adjoint_local;
  if (c->ks.adjoint_local_wave && c->kernel_variant != C8_KERNEL_SLOT) fn = c->ks.adjoint_local_wave;
  // the model's closed form of the local adjoint solve (hex8 small_J2), where the caller leaves the kernel choice to the library
  if (c->ks.adjoint_local_closed && c->ms.closed_form && !c->subset && st->xi != st->xi_prev &&
      (c->kernel_variant == C8_KERNEL_AUTO || c->kernel_variant == C8_KERNEL_NODE))
    fn = c->ks.adjoint_local_closed;
  return run(c, fn, field_args(st), aa, SystemArgs{}, false, "c8_solve_adjoint_local");
}

int c8_param_gradient(c8_ctx* c, const c8_state* st, const double* const z[2], const double* phi, double* grad) {
  if (!c || !check_state(c, st) || !check_z(c, z) || !phi || !grad) return fail(C8_ERR_ARG, "c8_param_gradient: null argument");
  int const rcq = c8_qoi_prepare(c, field_args(st));  // preprocess_qoi (evaluations.cpp:780)
  if (rcq) return rcq;
  AdjointArgs aa{nullptr, nullptr, z[0], z[1], const_cast<double*>(phi), grad, c->d_active, c8_qoi_args(c)};
  LaunchFn fn = c->ks.param_gradient;
  if (c->ks.param_gradient_wave && c->kernel_variant != C8_KERNEL_SLOT) fn = c->ks.param_gradient_wave;
  // the model's closed form (hex8 small_J2), where the caller leaves the kernel choice to the library
  if (c->ks.param_gradient_closed && c->ms.closed_form && !c->subset && (c->kernel_variant == C8_KERNEL_AUTO || c->kernel_variant == C8_KERNEL_NODE))
    fn = c->ks.param_gradient_closed;
  return run(c, fn, field_args(st), aa, SystemArgs{}, false, "c8_param_gradient");
}

int c8_eval_qoi(c8_ctx* c, const c8_state* st, double* J) {
  if (!c || !st || !st->x[0] || (c->nres == 2 && !st->x[1]) || !J) return fail(C8_ERR_ARG, "c8_eval_qoi: null argument");
  FieldArgs fa{st->x[0], st->x[1], st->x_prev[0], st->x_prev[1], st->xi_prev, st->xi};
  if (c->qoi_kind == 0) {
    AdjointArgs aa{nullptr, nullptr, nullptr, nullptr, nullptr, J, c->d_active, c8_qoi_args(c)};
    return run(c, c->kernel_variant == C8_KERNEL_SLOT ? c->ks.qoi_slot : c->ks.qoi, fa, aa, SystemArgs{}, false, "c8_eval_qoi");
  }
  // calibration (Calibration<double>::evaluate + postprocess): preprocess_qoi (evaluations.cpp:674), the face
  // term, and 1/2 balance dt/T load_mismatch^2
  if (!st->xi || !st->xi_prev) return fail(C8_ERR_ARG, "c8_eval_qoi: the calibration objective needs the local state");
  int rc = c8_qoi_prepare(c, fa);
  if (rc || (rc = c8_qoi_surface(c, st->x[0], J, nullptr)) || (rc = c8_qoi_postprocess(c, J))) return rc;
  return c->async ? C8_OK : c8_status(c);
}

int c8_brick_mesh(int nx, int ny, int nz, double lx, double ly, double lz, double* coords, int32_t* conn) {
  if (nx <= 0 || ny <= 0 || nz <= 0 || !coords || !conn) return fail(C8_ERR_ARG, "c8_brick_mesh: bad argument");
  HostMesh m;
  make_brick(nx, ny, nz, lx, ly, lz, m);
  std::memcpy(coords, m.coords.data(), m.coords.size() * sizeof(double));
  std::memcpy(conn, m.conn.data(), m.conn.size() * sizeof(int32_t));
  return C8_OK;
}
int c8_brick_partition(int nx, int ny, int nz, int px, int py, int pz, int32_t* elem_part) {
  if (nx <= 0 || ny <= 0 || nz <= 0 || px <= 0 || py <= 0 || pz <= 0 || !elem_part) return fail(C8_ERR_ARG, "c8_brick_partition: bad argument");
  std::vector<int32_t> part;
  brick_partition(nx, ny, nz, px, py, pz, part);
  std::memcpy(elem_part, part.data(), part.size() * sizeof(int32_t));
  return C8_OK;
}

}  // extern "C"
