/* c8.h -- C ABI of the MI355X assembly / adjoint-sensitivity path of CALIBR8.
 *
 * This is the drop-in boundary.  Each entry point replaces one
 * of the reference's `eval_*` free functions (source/calibr8/src/evaluations.hpp:23-84);
 * the arrays it reads and writes are the raw arrays those functions reach through
 * Tpetra / apf today:
 *   nodal fields x[i]     disc->primal(step).global[i]          (evaluations.cpp:24-27)
 *   point fields xi       disc->primal(step).local[model_form]  (apf IP fields)
 *   A[i][j]               la->A[GHOST][i][j] local CSR `values` (global_residual.cpp:529-537)
 *   b[i]                  la->b[GHOST][i]  get1dViewNonConst()  (global_residual.cpp:469)
 * Residual index 0 = u (3 equations per node), 1 = p (1 equation per node):
 * `mechanics` with the mixed formulation (mechanics.cpp:16-55).  Local row id of
 * (node, eq) in block i is node*neq_i + eq (disc.cpp:263-265 get_dof).
 *
 * Conventions
 *   - No exceptions, no C++ types.  Every function returns an int status:
 *       C8_OK (0); C8_LOCAL_SOLVE_FAILED (-1) = some local Newton solve did not converge,
 *       outputs undefined -- the reference's convention (evaluations.hpp:19,
 *       evaluations.cpp:95-97); < -1 = API misuse or device error, see c8_last_error().
 *   - c8_mesh_desc / c8_model_desc hold HOST pointers, copied at c8_create().
 *   - c8_state / c8_system and all other array arguments of the assembly calls are
 *     DEVICE pointers (HBM).  The caller owns them.
 *   - Outputs A, b, grad are ACCUMULATED INTO (+=), like scatter_lhs/scatter_rhs
 *     (global_residual.cpp:463-479,556-586); the caller zeroes first (primal.cpp:98).
 *   - One in-flight call per ctx; different ctxs (different GPUs/streams) are independent.
 */
#ifndef C8_H
#define C8_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct c8_ctx c8_ctx;

enum { C8_ELEM_TRI3 = 3, C8_ELEM_TET4 = 4, C8_ELEM_HEX8 = 8 };
enum { C8_OK = 0, C8_LOCAL_SOLVE_FAILED = -1, C8_ERR_ARG = -2, C8_ERR_DEVICE = -3, C8_ERR_UNSUPPORTED = -4 };
/* ATOMIC: one launch, f64 atomic adds.  COLORED: one launch per element colour, plain adds, reproducible.
 * GATHER: element matrices are staged element-major in a context-owned buffer (8.4 KB per hex8, 2.2 KB per
 * tet4 element) and a second kernel sums each node's rows in ascending element order: no atomics, bitwise
 * reproducible; the fastest mode on both element types and the DEFAULT of c8_create wherever no node has more than
 * 64 neighbours.  If the stage cannot be allocated at the first Jacobian assembly, a context still in its default
 * mode switches to COLORED and says so in c8_last_error(). */
enum { C8_SCATTER_ATOMIC = 0, C8_SCATTER_COLORED = 1, C8_SCATTER_GATHER = 2 };
enum { C8_KERNEL_AUTO = 0, C8_KERNEL_SLOT = 1, C8_KERNEL_WAVE = 2, C8_KERNEL_WAVE_AD = 3, C8_KERNEL_NODE = 4 };

/* One mesh part (what Disc holds after loadMdsMesh, disc.cpp:31-39): all nodes that touch a
 * local element, local (GHOST) numbering. */
typedef struct {
  int32_t elem_type;       /* C8_ELEM_TET4 / C8_ELEM_TRI3 (what the reference runs in 3-D / 2-D, disc.cpp:165) or
                              C8_ELEM_HEX8.  On a tri3 mesh `mechanics` has 2 + 1 equations per node (mechanics.cpp:34):
                              every u array is [num_nodes*2]; coords stay [num_nodes][3] with z = 0. */
  int32_t num_nodes;
  int32_t num_elems;
  int32_t num_elem_sets;   /* material blocks, disc->num_elem_sets() */
  const double* coords;    /* [num_nodes][3] */
  const int32_t* conn;     /* [num_elems][nodes per element], local node ids */
  const int32_t* elem_set; /* [num_elems] element-set id, or NULL when num_elem_sets == 1 */
  /* Extra (row node, col node) couplings to reserve in the graphs besides those of the local
   * elements: on a multi-part mesh, the off-part columns of owned interface rows, so that the owned
   * rows already have the union pattern of compute_owned_graph (disc.cpp:389-398) and the halo
   * ADD of linear_alg.cpp:53-63 can land in place.  NULL / 0 on a single part. */
  int32_t num_extra_pairs;
  const int32_t* extra_pairs; /* [num_extra_pairs][2] */
} c8_mesh_desc;

/* The `residuals:` block of a deck (global_residual.cpp:620-630, local_residual.cpp:893-933). */
typedef struct {
  const char* global_type;          /* "mechanics" (mixed u-p formulation, two residuals), or on tri3 meshes
                                       "mechanics_plane_stress" (mechanics_plane_stress.cpp: ONE residual, u; it pairs
                                       with the *_plane_stress local residuals and only with them) */
  const char* local_type;           /* "elastic" | "small_J2" | "hyper_J2" | "small_hill" | "isotropic_elastic" |
                                       "hypo_hill" | "small_hosford" | "hypo_hosford" | "hypo_barlat"; on tri3 meshes "small_J2" | "small_hill_plane_strain" |
                                       "hyper_J2_plane_strain" | "hypo_hill_plane_strain", and under
                                       mechanics_plane_stress "small_hill_plane_stress" | "hyper_J2_plane_stress" |
                                       "hypo_hill_plane_stress" (the names of local_residual.cpp:893-933) */
  double stabilization_multiplier;  /* mechanics.cpp:47 */
  int32_t local_max_iters;          /* "nonlinear max iters" of the local residual */
  double local_abs_tol;             /* "nonlinear absolute tol" */
  double local_rel_tol;             /* "nonlinear relative tol" */
  int32_t num_params;               /* elastic 4 (E nu cte delta_T), small_J2 6 (E nu K Y cte delta_T),
                                       hyper_J2 8 (E nu Y S D A n K), small_hill / hypo_hill 11 (E nu Y R00 R11 R22
                                       R01 R02 R12 S D), isotropic_elastic 2 (E nu), small_hosford / hypo_hosford 7 (E nu Y a K S D),
                                       hypo_barlat 25 (those seven + sp_01 sp_02 sp_10 sp_12 sp_20 sp_21 sp_33 sp_44 sp_55
                                       and dp_01 .. dp_55), small_hill_plane_strain /
                                       hypo_hill_plane_strain 9 (E nu Y S D R00 R11 R22 R01), hyper_J2_plane_strain 6 (E nu K Y Y_inf delta),
                                       small_hill_plane_stress 9 (as plane strain), hyper_J2_plane_stress 8 (as hyper_J2),
                                       hypo_hill_plane_stress 13 (the nine + Q00 Q01 Q10 Q11) */
  const double* params;             /* [num_elem_sets][num_params] */
  double thickness;                 /* mechanics_plane_stress.cpp:22 "thickness"; 0 = the reference's default 1 */
  /* `line search:` sublist of the local residual (line_search.hpp:40-49), used by the local Newton iteration of
   * small_hosford / hypo_hosford / hypo_barlat; 0 in a field = the reference's default (1e-4, 0.5, 0.9, 4 evaluations) */
  double ls_sufficient_decrease;
  double ls_min_backtrack;
  double ls_max_backtrack;
  int32_t ls_max_evals;
} c8_model_desc;

/* Primal state at one load step (DEVICE pointers).  Under mechanics_plane_stress (c8_num_residuals() == 1) the entries
 * [1] of a state, of a system (A[i][j] with i or j = 1, b[1]) and of an adjoint vector z are not read and may be NULL. */
typedef struct {
  const double* x[2];      /* x[0] = u [num_nodes*3], x[1] = p [num_nodes]  at step n   */
  const double* x_prev[2]; /* same at step n-1                                           */
  const double* xi_prev;   /* [num_elems][c8_num_local_points][c8_num_local_dofs], step n-1 */
  double* xi;              /* same at step n: written by the forward assembly, read by the others */
} c8_state;

/* Linear system in GHOST distribution (DEVICE pointers). */
typedef struct {
  double* A[2][2]; /* CSR values over the graphs reported by c8_graph() */
  double* b[2];    /* b[0] [num_nodes*3], b[1] [num_nodes] */
} c8_system;

/* ---- lifetime ------------------------------------------------------------------------- */
/* Builds host tables (node graph, element colouring) and their device mirrors on the current
 * HIP device.  Replaces State::State + Disc::build_data for this path (state.cpp:33-46,
 * disc.cpp:563-583). */
int c8_create(const c8_mesh_desc* mesh, const c8_model_desc* model, c8_ctx** out);
void c8_destroy(c8_ctx* ctx);
const char* c8_last_error(void);
/* "id=<sha of sources and flags> flags=<compiler flags>" of this library build (calibr8_amd/build.py): profiles carry
 * the id of the build they were measured on. */
const char* c8_build_info(void);

/* ---- discretisation queries (host arrays) ------------------------------------------------ */
int c8_num_local_dofs(const c8_ctx* ctx);    /* LocalResidual::num_dofs: 1 / 7 / 8 (3-D), 4 (2-D) */
int c8_num_dims(const c8_ctx* ctx);          /* 3, or 2 on a tri3 mesh: equations per node of residual 0 */
int c8_num_residuals(const c8_ctx* ctx);     /* GlobalResidual::num_residuals: 2 (mechanics: u, p) or 1 (mechanics_plane_stress: u) */
int c8_num_local_points(const c8_ctx* ctx);  /* points of the local-state field per element */
int c8_num_colors(const c8_ctx* ctx);
/* Block (i,j) CSR graph = m_graphs[GHOST][i][j] (disc.cpp:356-387): sorted columns. */
int64_t c8_graph_nnz(const c8_ctx* ctx, int i, int j);
int c8_graph(const c8_ctx* ctx, int i, int j, int64_t* rowptr, int32_t* colidx);
/* LocalResidual::init_variables (local_residual.cpp:35-74): initial local state, HOST array
 * [num_elems][points][dofs]. */
int c8_init_variables(const c8_ctx* ctx, double* xi_host);

/* ---- run-time settings ------------------------------------------------------------------- */
int c8_set_params(c8_ctx* ctx, const double* params_host); /* LocalResidual::set_params; [sets][num_params] */
/* Active (differentiated) parameters of one element set, LocalResidual::m_active_indices
 * (small_J2.cpp:96-98 default = {0}; objective.cpp:110-115 overrides from the inverse block). */
int c8_set_active_params(c8_ctx* ctx, int elem_set, int n, const int32_t* param_idx);
int c8_num_active_params(const c8_ctx* ctx); /* total over element sets = length of grad */
int c8_set_stream(c8_ctx* ctx, void* hip_stream);
int c8_set_scatter_mode(c8_ctx* ctx, int mode); /* C8_SCATTER_GATHER (default, see above), C8_SCATTER_ATOMIC or C8_SCATTER_COLORED */
int c8_get_scatter_mode(const c8_ctx* ctx);
/* C8_SCATTER_GATHER tuning: elements are staged in chunks of at least `min_chunk` elements (the chunk is never
 * smaller than the element bandwidth of the mesh) through a ring of three chunks.  Default: the whole mesh in one
 * chunk while its stage stays under 12 GB (8.4 KB per hex8, 2.2 KB per tet4 element), else chunks of 262144. */
int c8_set_stage_chunk(c8_ctx* ctx, int min_chunk);
/* C8_SCATTER_GATHER in several chunks: on = 1 issues the row sums of chunk k on a stream of the context's own, beside
 * the assembly of chunk k + 1 on the caller's stream (events order the two both ways; the caller's stream continues
 * only after the last row sums, so the call keeps its stream semantics).  Same values bit for bit. */
int c8_set_stage_overlap(c8_ctx* ctx, int on);
/* C8_SCATTER_GATHER in two parts, for a caller that exchanges ghost rows (LinearAlg::gather_A / gather_b,
 * linear_alg.cpp:53-86): with an early node range set, a Jacobian assembly stages every element and sums the rows of
 * the nodes [node_begin, node_end) only -- the ghost rows, which the caller can then pack and send -- and
 * c8_gather_finish sums the rows of all other nodes while the exchange is in flight.  Every assembly must then be
 * followed by c8_gather_finish before the system is used.  Needs the whole mesh in one staged chunk; an empty range
 * (the default) turns the split off. */
int c8_set_gather_early_nodes(c8_ctx* ctx, int node_begin, int node_end);
int c8_gather_finish(c8_ctx* ctx);
/* C8_SCATTER_GATHER, assign mode (default off = the reference's accumulate-into semantics): the two Jacobian assemblies
 * ASSIGN the rows of A (all four blocks) and b of every node that has elements instead of adding to them, i.e.
 * la->zero_all() followed by eval_forward_jacobian / eval_adjoint_jacobian (primal.cpp:98-99, adjoint.cpp:123-125) in one
 * call: the caller's zeroing pass and the row sums' read of the old values (3.5 of 15.9 GB on a million hex8 elements)
 * both go away.  Rows of nodes without elements are not touched.  The other entry points (residual-only assembly,
 * boundary conditions, ...) keep adding; a Jacobian assembly in another scatter mode is refused while the mode is on. */
int c8_set_assign_mode(c8_ctx* ctx, int on);
/* Shape-table cache (default on; hex8 wave kernels): the geometry of a context is static, so dN/dx, w dv and the element
 * size are computed once at c8_create (1.7 KB per element) instead of by every call (weight.cpp:5-25 recomputes them for
 * every AD pass).  on = 0 frees the tables; results are the same either way. */
int c8_set_shape_cache(c8_ctx* ctx, int on);
/* Forward-assembly kernel: C8_KERNEL_SLOT = one lane group per element (any element type);
 * C8_KERNEL_WAVE = one wavefront per element (hex8); C8_KERNEL_AUTO picks WAVE where available.  Where a model has a
 * closed form of its local equations (small_J2 on 3-D meshes: radial return and consistent tangent), the forward WAVE
 * kernel -- and, under C8_KERNEL_AUTO, the lane-group kernel of an element type without a wave kernel (tet4) -- uses it
 * in place of the local Newton iteration and the automatic-differentiation passes: the same converged state and
 * Jacobian as the iterated form to within the local Newton tolerance (2e-13 measured).  C8_KERNEL_WAVE_AD (hex8) and an
 * explicit C8_KERNEL_SLOT keep the iterated, automatically differentiated form; it also runs whenever
 * local_max_iters < 8, so that a local solve that cannot converge within the caller's budget still reports
 * C8_LOCAL_SOLVE_FAILED as the reference does.
 * C8_KERNEL_NODE (hex8, models with a closed form, C8_SCATTER_GATHER): eval_forward_jacobian with one wavefront per NODE --
 * the wavefront forms the node's four CSR rows from the node's elements (closed form recomputed per node, every 4 x 4
 * block of an element matrix formed once, by its row node) and writes them once: no element stage, no atomics, bitwise
 * reproducible, a third of the staged form's memory traffic.  C8_KERNEL_AUTO takes it where it applies (cached shape
 * tables present, st->xi and st->xi_prev distinct arrays); C8_KERNEL_WAVE keeps the staged one-wavefront-per-element form.
 * It honours c8_set_assign_mode and c8_set_gather_early_nodes / c8_gather_finish like the staged form; between the
 * assembly call and c8_gather_finish the arrays of `st` must stay as they were (the second part reads them).
 * Under C8_KERNEL_AUTO / C8_KERNEL_NODE c8_assemble_adjoint_jacobian takes the same form and c8_solve_adjoint_local the
 * model's closed form of the local adjoint solve (no dual numbers, no elimination of dC/dxi; same results to rounding);
 * c8_param_gradient likewise takes the closed form of the point's share of the gradient. */
int c8_set_kernel_variant(c8_ctx* ctx, int variant);
/* async = 1: assembly calls only enqueue and return C8_OK; c8_status() then synchronises the
 * stream and reports C8_OK / C8_LOCAL_SOLVE_FAILED for everything enqueued since the last call. */
int c8_set_async(c8_ctx* ctx, int async);
int c8_status(c8_ctx* ctx);

/* ---- the objective (QoI) of the adjoint path ---------------------------------------------------------
 * Default: "average displacement" (avg_disp.cpp).  "calibration" (calibration.cpp, 3-D form):
 *   J_step = dt/T [ 1/(2 area) int_side sum_d w_d (u_d - u_meas_d)^2 dS * (coupled points per element)
 *                   + 1/2 balance (load - load_meas)^2 ],
 * load = internal force component `reaction_comp` summed over the nodes with |x[coord_idx] - coord_value| <
 * coord_tol.  The factor "coupled points per element" is the reference's: it adds the face integral at every
 * coupled integration point of the element (1 for tet4, 8 for this library's hex8 extension).  On a tri3 mesh the
 * displacement term is the reference's 2-D branch (calibration.cpp:76-104, :163-222): the integral runs over the
 * ELEMENTS -- all of them (num_faces = 0), or the listed ones (a distance field with a threshold in the reference) --
 * and `area` is the sum of their areas. */
typedef struct {
  int32_t num_faces, nodes_per_face;  /* displacement side set: faces as node ids, 3 per face (tet4) or 4 (hex8); on a
                                         tri3 mesh a list of ELEMENT ids (nodes_per_face = 1), or 0 for every element */
  const int32_t* faces;               /* HOST array [num_faces][nodes_per_face] */
  double weights[3];                  /* "displacement weights" */
  double balance_factor;
  int32_t coord_idx;                  /* "coordinate index" / "coordinate value" / "coordinate tolerance" */
  double coord_value, coord_tol;
  int32_t reaction_comp;              /* "reaction force component" */
  double dt_over_total_time;          /* m_dt / m_total_time */
} c8_calibration_desc;
/* Multi-part meshes: the calibration objective sums the side-set area and the reaction load over all parts
 * (PCU_Add_Double, calibration.cpp:138, :351) and shares the load term of J between the parts (:375-378).  The
 * caller supplies the SUM all-reduce over its communicator (HOST doubles, in place) and the number of parts; the
 * library calls it from c8_qoi_preprocess and from the entry points that run preprocess_qoi. */
typedef void (*c8_allreduce_fn)(void* user, double* values, int n);
int c8_set_allreduce(c8_ctx* ctx, c8_allreduce_fn fn, void* user, int num_parts);
int c8_set_qoi_avg_disp(c8_ctx* ctx);
int c8_set_qoi_calibration(c8_ctx* ctx, const c8_calibration_desc* desc);
/* measured data of the current step: nodal displacements (DEVICE array [nodes][3], kept by reference) and load */
int c8_set_measured(c8_ctx* ctx, const double* u_meas, double load_meas);
/* preprocess_qoi (evaluations.cpp:262-347); the adjoint entry points below run it themselves, this call only
 * reports: out (HOST, 3 doubles, may be NULL) = {side-set area, total load, load mismatch}. */
int c8_qoi_preprocess(c8_ctx* ctx, const c8_state* st, double* out);

/* ---- the hot path (all array arguments are DEVICE pointers) --------------------------------- */
/* eval_forward_jacobian (evaluations.cpp:12-154): R and dR/dx with the local state condensed;
 * writes the converged local state to st->xi. */
int c8_assemble_forward_jacobian(c8_ctx* ctx, const c8_state* st, const c8_system* sys);
/* The same over a subset of the elements (`elems`: DEVICE array of `count` distinct element ids), with atomic
 * adds.  For overlapping the halo exchange with the assembly: assemble the elements that touch
 * ghost nodes, start the exchange, assemble the rest.  The context must be in C8_SCATTER_ATOMIC mode. */
int c8_assemble_forward_jacobian_subset(c8_ctx* ctx, const c8_state* st, const c8_system* sys, const int32_t* elems, int count);
/* eval_global_residual (evaluations.cpp:156-259): R only, from the stored local state. */
int c8_assemble_residual(c8_ctx* ctx, const c8_state* st, const c8_system* sys);
/* eval_adjoint_jacobian (evaluations.cpp:349-526) for the context's objective:
 * A += (dR/dx total)^T, b += -dJ/dx + f + (dxi/dx)^T g, and g -= dJ/dxi in place.
 * g [elems][points][local dofs], f [elems][points][element dofs]. */
int c8_assemble_adjoint_jacobian(c8_ctx* ctx, const c8_state* st, double* g, const double* f, const c8_system* sys);
/* solve_adjoint_local (evaluations.cpp:528-659): phi from the global adjoint z, then the
 * history vectors f, g for the previous step (overwritten). */
int c8_solve_adjoint_local(c8_ctx* ctx, const c8_state* st, const double* const z[2], double* phi, double* g, double* f);
/* eval_qoi_gradient (evaluations.cpp:758-925): grad[c8_num_active_params] +=
 * sum_e sum_pt (dC/dp)^T phi + dJ/dp + (dR/dp)^T z. */
int c8_param_gradient(c8_ctx* ctx, const c8_state* st, const double* const z[2], const double* phi, double* grad);
/* eval_qoi (evaluations.cpp:662-756), "average displacement" (avg_disp.cpp:16-33): *J += value. */
int c8_eval_qoi(c8_ctx* ctx, const c8_state* st, double* J);

/* ---- next to the hot path: boundary conditions and the Newton step driver ----- */
/* One Dirichlet condition = one entry of the deck's `dirichlet bcs` block (dbcs.cpp:59-66): residual
 * index, equation, node set, and the prescribed value at every node of the set (the caller evaluates
 * the expression or reads the measured field, dbcs.cpp:76 / :183-185).  DEVICE pointers. */
typedef struct {
  int32_t resid, eq, n;
  const int32_t* nodes;   /* [n] local node ids */
  const double* values;   /* [n] */
} c8_dbc;
/* One traction condition (tbcs.cpp:17-86) on the sides of the mesh: tri3 faces of tet4 meshes and the edges of tri3
 * meshes with the reference's 1-point rule, quad4 faces of hex8 meshes with the 2x2 rule.  DEVICE pointers. */
typedef struct {
  int32_t resid, n, nodes_per_face;  /* 2 (edges of a 2-D mesh), 3 or 4 */
  const int32_t* faces;    /* [n][nodes_per_face] local node ids */
  const double* traction;  /* [n][points][3], points = 1 (edge, tri3) or 4 (quad4); on a 2-D mesh the third entry is ignored */
} c8_tbc;
/* apply_primal_dbcs (dbcs.cpp:28-121): for every constrained row keep the diagonal entry, zero the rest
 * of the row in every block, b[row] = diag * (x[row] - value), or 0 when is_adjoint. */
int c8_apply_dirichlet(c8_ctx* ctx, int n, const c8_dbc* dbcs, const double* const x[2], const c8_system* sys, int is_adjoint);
/* apply_primal_tbcs (tbcs.cpp:88-98): b[row(node, d)] -= T_d N_node w dv over the faces. */
int c8_apply_traction(c8_ctx* ctx, int n, const c8_tbc* tbcs, const c8_system* sys);
/* Integration points of boundary sides (HOST arrays; coords [.][3]): xyz [n][points][3], for evaluating traction expressions. */
int c8_face_points(int nodes_per_face, int n, const double* coords, const int32_t* faces, double* xyz);
/* LinearAlg::apply_A (linear_alg.cpp:158-175): y = A x over the four blocks (DEVICE pointers). */
int c8_apply_A(c8_ctx* ctx, const c8_system* sys, const double* const x[2], double* const y[2]);

/* The linear solve is out of scope (Belos/Teko/MueLu in the reference, linear_solve.cpp): the driver
 * calls back with the assembled system (b already scaled to -R, primal.cpp:131) and device arrays to
 * receive dx.  Return 0 on success. */
typedef int (*c8_linear_solve_fn)(void* user, const c8_system* sys, double* const dx[2]);
typedef struct {
  int32_t max_iters;            /* "nonlinear max iters" of the global residual */
  double abs_tol, rel_tol;      /* "nonlinear absolute/relative tol" */
  int32_t line_search;          /* 1 = Armijo/cubic backtracking (line_search.hpp), 0 = full steps */
  double sufficient_decrease;   /* 1e-4 */
  double min_backtrack, max_backtrack; /* 0.5, 0.9 */
  int32_t max_evals;            /* 4 */
} c8_newton_opts;
/* Primal::solve_at_step (primal.cpp:31-209): Newton iterations on st->x (updated in place, st->xi receives the
 * converged local state) with the reference's convergence tests and line search.  With a halo attached to the context
 * (c8_halo_attach) the step runs over all parts: every rank calls it collectively, the assembly is followed by the
 * status all-reduce (primal.cpp:100,164), gather_A / gather_b (:110-111), boundary conditions and norms on the OWNED
 * rows, and the solution increment is imported to the ghost copies (disc.cpp:944-947); the callback then solves the
 * distributed owned system and must fill dx on the owned nodes.  Returns C8_OK, C8_LOCAL_SOLVE_FAILED (base point
 * or every line-search trial failed, on any part), or C8_NOT_CONVERGED; *iters = Newton iterations taken. */
enum { C8_NOT_CONVERGED = -5 };
int c8_primal_solve_step(c8_ctx* ctx, const c8_state* st, const c8_system* sys, int ndbc, const c8_dbc* dbcs, int ntbc,
                         const c8_tbc* tbcs, const c8_newton_opts* opts, c8_linear_solve_fn solve, void* user,
                         int32_t* iters);

/* Adjoint::solve_at_step (adjoint.cpp:76-189) followed by eval_qoi_gradient (adjoint_objective.cpp:86-94)
 * for one load step of one part: zero + c8_assemble_adjoint_jacobian, adjoint Dirichlet rows, the
 * caller's linear solve (system passed as assembled: A = (dR/dx)^T, b = right-hand side, NOT negated;
 * the solution is written to z), c8_solve_adjoint_local, and grad += this step's contribution.
 * March the steps from the last to the first with the same g, f arrays (zero before the last step). */
int c8_adjoint_solve_step(c8_ctx* ctx, const c8_state* st, const c8_system* sys, int ndbc, const c8_dbc* dbcs,
                          c8_linear_solve_fn solve, void* user, double* const z[2], double* phi, double* g, double* f,
                          double* grad);

/* ---- multi-part meshes: owned/ghost halo and reductions ---------------------------------
 * One process per GPU, one mesh part per process, elements not ghosted, nodes on part boundaries shared -- the
 * reference's MPI scheme (disc.cpp:237,297).  Local node numbering of a part: OWNED nodes, then GHOST nodes (touched
 * by a local element, owned elsewhere), then PHANTOM nodes (not touched locally: columns of owned interface rows,
 * reserved in the graphs through c8_mesh_desc.extra_pairs), so that the OWNED matrix and vectors are prefix views of
 * the local arrays and every exchange lands in place.  What replaces what:
 *   C1  c8_halo_gather (C8_HALO_B)      LinearAlg::gather_b, Tpetra Export ADD          linear_alg.cpp:78-86
 *   C2  c8_halo_gather (C8_HALO_A)      LinearAlg::gather_A                             linear_alg.cpp:53-63
 *   C3  c8_halo_scatter_x               apf::synchronize in Disc::add_to_soln / Import  disc.cpp:944-947, :1001
 *   C4  c8_comm_allreduce_sum           PCU_Add_Doubles(grad)                           adjoint_objective.cpp:109
 *   C5  c8_comm_allreduce_sum           PCU_Add_Double(J), PCU_Add_Int(status)          adjoint_objective.cpp:39,99; primal.cpp:100,164
 * Transport: RCCL point-to-point over xGMI (grouped ncclSend/ncclRecv, one message per neighbour and exchange, on a
 * stream of its own; pack and unpack-add are HIP kernels on the context's stream), or a caller-supplied host exchange
 * (an MPI host without device-aware transport; several ranks sharing one card, which RCCL refuses).  The received
 * contributions of a value are added in ascending source-rank order: the gathered system is bitwise reproducible. */
typedef struct c8_comm c8_comm;
typedef struct c8_halo c8_halo;
enum { C8_COMM_ID_BYTES = 128 };
/* RCCL: rank 0 calls c8_comm_rccl_id, the caller broadcasts the 128 bytes by whatever it has (MPI_Bcast in the
 * reference's host code, torch.distributed in this repo's bench), every rank calls c8_comm_create_rccl with its HIP
 * device current.  librccl is loaded at the first of these calls (C8_RCCL_LIB overrides the search). */
int c8_comm_rccl_id(void* id_out);
int c8_comm_create_rccl(const void* id, int rank, int nranks, c8_comm** out);
/* Host transport.  exchange: chunk r of `send` (send_counts[r] doubles, chunks back to back in rank order) goes to
 * rank r, chunk r of `recv` (recv_counts[r] doubles) comes from rank r; HOST buffers; blocking.  allreduce: in-place
 * SUM of n HOST doubles.  Both return 0 on success. */
typedef int (*c8_host_exchange_fn)(void* user, const double* send, const int64_t* send_counts, double* recv,
                                   const int64_t* recv_counts);
typedef int (*c8_host_allreduce_fn)(void* user, double* values, int n);
int c8_comm_create_host(int rank, int nranks, c8_host_exchange_fn exchange, c8_host_allreduce_fn allreduce, void* user,
                        c8_comm** out);
void c8_comm_destroy(c8_comm* comm);
int c8_comm_rank(const c8_comm* comm);
int c8_comm_size(const c8_comm* comm);
/* C4 / C5: in-place SUM over the ranks of n HOST doubles (gradient, objective, failure flag packed by the caller). */
int c8_comm_allreduce_sum(c8_comm* comm, double* values, int n);

/* The exchange lists of one part (HOST arrays, copied).  What Tpetra derives from the OWNED and GHOST maps
 * (disc.cpp:316-332); the caller gets them from its mesh database (PUMI's remote copies in the reference).
 * Rows travel whole, in the SENDER's graph order; the receiver needs the sender's column lists once. */
typedef struct {
  int32_t num_owned, num_touched;   /* local nodes [0, num_owned) owned, [num_owned, num_touched) ghost, rest phantom */
  /* export, C1/C2: my ghost rows, grouped by owner rank; send_ptr [nranks+1] */
  const int64_t* send_ptr;
  const int32_t* send_nodes;        /* local (ghost) node ids */
  /* rows I own that rank r sends me, in r's order, with r's column list of each row in MY local ids */
  const int64_t* recv_ptr;          /* [nranks+1] */
  const int32_t* recv_nodes;        /* local (owned) node ids */
  const int64_t* recv_col_ptr;      /* [recv_ptr[nranks] + 1] */
  const int32_t* recv_cols;         /* local node ids (owned, ghost or phantom) */
  /* import, C3: nodes whose values I receive (every ghost and phantom node), grouped by owner rank; and the owned
   * nodes I send to each importing rank, in the importer's order */
  const int64_t* import_ptr;        /* [nranks+1] */
  const int32_t* import_nodes;
  const int64_t* export_ptr;        /* [nranks+1] */
  const int32_t* export_nodes;
  /* shape of the systems the tables are for: equations per node of residual 0 (c8_num_dims: 3, or 2 on tri3 meshes) and
   * the number of residuals (c8_num_residuals: 2, or 1 under mechanics_plane_stress); 0 = 3 and 2 */
  int32_t num_dims, num_residuals;
} c8_halo_desc;
/* Index tables of the exchanges, built on the host from the part's node graph (= block (1,1) of c8_graph(): one row
 * per node, sorted neighbour ids) -- needs no device. */
int c8_halo_build(int32_t num_nodes, const int64_t* node_rowptr, const int32_t* node_colidx, const c8_halo_desc* desc,
                  int rank, int nranks, c8_halo** out);
/* Device mirrors of the tables, message buffers, events, on the context's device; the context's graph must be the one
 * the tables were built from.  Also registers the halo with the context: the step drivers (c8_primal_solve_step,
 * c8_adjoint_solve_step) and the calibration objective then work over all parts. */
int c8_halo_attach(c8_halo* halo, c8_ctx* ctx, c8_comm* comm);
void c8_halo_destroy(c8_halo* halo);
enum { C8_HALO_B = 1, C8_HALO_A = 2 };
/* C1 and/or C2 in ONE message per neighbour: ghost rows of b (C8_HALO_B), of the four blocks of A (C8_HALO_A) or both
 * are packed on the context's stream and sent; c8_halo_gather_finish adds what arrived into the owned rows, on the
 * context's stream.  Work enqueued on the context's stream between the two calls (interior elements, the owned rows'
 * sums: c8_gather_finish) overlaps the exchange.  Afterwards the first num_owned node rows hold the OWNED system;
 * ghost rows are scratch.  c8_halo_gather = start + finish. */
int c8_halo_gather_start(c8_halo* halo, const c8_system* sys, int what);
int c8_halo_gather_finish(c8_halo* halo, const c8_system* sys);
int c8_halo_gather(c8_halo* halo, const c8_system* sys, int what);
/* C3: owner values of a nodal field pair x = {u [nodes*num_dims], p [nodes]} (x[1] unused with one residual) copied to
 * every ghost and phantom copy. */
int c8_halo_scatter_x(c8_halo* halo, double* const x[2]);
/* bytes this rank sends per exchange (what = C8_HALO_A | C8_HALO_B, or 0 for the C3 import) */
int64_t c8_halo_send_bytes(const c8_halo* halo, int what);
/* Diagnostic / test access to the host tables (tests/ replay the exchanges in numpy without a device):
 * which = 0 gather send codes, 1 gather send counts per rank, 2 gather recv counts, 3 unpack destinations (codes),
 * 4 unpack source offsets, 5 unpack source list; 10..15 the same for C8_HALO_B alone; 20 import send codes (C3),
 * 21 its send counts, 22 its recv counts, 23 its destination codes.  A code is (segment << 56) | offset with segment
 * 0..3 = A00 A01 A10 A11, 4 = b[0] / x[0], 5 = b[1] / x[1]. */
int c8_halo_table(const c8_halo* halo, int which, int64_t* n, const int64_t** data);

/* ---- canonical optimisation variables (HOST arrays) -------------------------------
 * Objective::transform_params / transform_gradient (objective.cpp:41-61,125-137) and their Python twins
 * (python/calibr8/util/parameter_transforms.py:31-59).  kind[i]: C8_SCALE_NONE (identity),
 * C8_SCALE_LOG (a = reference value: canonical = log(value / a)), C8_SCALE_BOUNDS (a = lower, b = upper:
 * canonical = (clip(value) - mean) / span in [-1, 1]). */
enum { C8_SCALE_NONE = 0, C8_SCALE_LOG = 1, C8_SCALE_BOUNDS = 2 };
int c8_transform_params(int n, const double* values, const int32_t* kind, const double* a, const double* b,
                        int from_canonical, double* out);
/* d(objective)/d(canonical) from d(objective)/d(physical); `values` are the CANONICAL values for log
 * scales, as in grad_transform (parameter_transforms.py:53-59). */
int c8_transform_gradient(int n, const double* grad, const double* values, const int32_t* kind, const double* a,
                          const double* b, double* out);

/* ---- host helpers for synthetic problems -------------------------------- */
/* Structured hex8 brick; coords [(nx+1)(ny+1)(nz+1)][3], conn [nx*ny*nz][8] (host arrays). */
int c8_brick_mesh(int nx, int ny, int nz, double lx, double ly, double lz, double* coords, int32_t* conn);
/* Block partition px*py*pz of that brick: part id per element (host array). */
int c8_brick_partition(int nx, int ny, int nz, int px, int py, int pz, int32_t* elem_part);

/* ---- outer optimiser: bound-constrained L-BFGS on the canonical variables, with the controls of
 * the reference's ROL set-up (main_inverse.cpp:21-28, :83-120: secant storage, iteration limit, gradient and step
 * tolerances, function evaluations per line search).  HOST arrays; `fn` returns 0, or non-zero when the objective
 * cannot be evaluated at x (the line search then shortens the step). */
typedef int (*c8_objective_fn)(void* user, int n, const double* x, double* f, double* grad);
typedef struct {
  int32_t max_iters;      /* "iteration limit" (20) */
  double grad_tol;        /* "gradient tolerance" (1e-12), on the projected gradient */
  double step_tol;        /* "step tolerance" (1e-12) */
  int32_t max_ls_evals;   /* "max line search evals" (5) */
  int32_t memory;         /* secant storage (20) */
} c8_lbfgs_opts;
enum { C8_LBFGS_ITERATION_LIMIT = 0, C8_LBFGS_GRADIENT_TOL = 1, C8_LBFGS_STEP_TOL = 2, C8_LBFGS_LINE_SEARCH_FAILED = 3 };
typedef struct {
  int32_t iters, evals, status;
  double f, projected_gradient_norm;
} c8_lbfgs_result;
int c8_lbfgs_minimize(int n, double* x, const double* lo, const double* hi, c8_objective_fn fn, void* user,
                      const c8_lbfgs_opts* opts, c8_lbfgs_result* res);

#ifdef __cplusplus
}
#endif
#endif /* C8_H */
