// c8_kernels.hpp -- launcher table between the C ABI (c8_api.hip) and the kernels.
#pragma once

#include <hip/hip_runtime_api.h>

#include "c8_assemble_adjoint.hpp"

namespace c8 {

enum { MODEL_NONE = -1, MODEL_ELASTIC = 0, MODEL_SMALL_J2 = 1, MODEL_HYPER_J2 = 2, MODEL_SMALL_HILL = 3, MODEL_ISOTROPIC_ELASTIC = 4, MODEL_HYPO_HILL = 5,
       MODEL_SMALL_HILL_PLANE_STRAIN = 6, MODEL_HYPER_J2_PLANE_STRAIN = 7, MODEL_HYPO_HILL_PLANE_STRAIN = 8,
       MODEL_SMALL_HILL_PLANE_STRESS = 9, MODEL_HYPER_J2_PLANE_STRESS = 10, MODEL_HYPO_HILL_PLANE_STRESS = 11,
       MODEL_SMALL_HOSFORD = 12, MODEL_HYPO_HOSFORD = 13, MODEL_HYPO_BARLAT = 14 };
inline bool model_is_plane_stress(int m) { return m >= MODEL_SMALL_HILL_PLANE_STRESS && m <= MODEL_HYPO_HILL_PLANE_STRESS; }

struct LaunchArgs {
  MeshTables mt;
  ModelSettings ms;
  FieldArgs fa;
  AdjointArgs aa;
  SystemArgs sa;
  int first, count;  // range of the element order to process
  hipStream_t stream;
};

typedef hipError_t (*LaunchFn)(LaunchArgs const&);
typedef hipError_t (*GatherFn)(GatherArgs const&, int first, int count, int max_degree, hipStream_t);
// aa = null: forward assembly; aa != null: adjoint assembly (objective "average displacement")
typedef hipError_t (*NodeRowsFn)(MeshTables const&, ModelSettings const&, FieldArgs const&, AdjointArgs const* aa, GatherArgs const&, int first,
                                 int count, int max_degree, int max_node_elems, hipStream_t);

struct KernelSet {
  LaunchFn forward_jacobian;   // K1, one lane group (NDOF lanes) per element
  LaunchFn forward_jacobian_wave;  // K1, one wavefront per element (hex8 only, else null)
  LaunchFn residual;           // K2
  LaunchFn residual_wave;      // K2, eight hex8 elements per wavefront, atomic adds (hex8 only, else null)
  LaunchFn adjoint_jacobian;   // K3
  LaunchFn adjoint_jacobian_wave;  // K3, one wavefront per element (hex8 only, else null)
  LaunchFn adjoint_local;      // K4 (per-point outputs only: one launch, no colouring)
  LaunchFn param_gradient;     // K5 (grid-stride, one atomic per lane at the end)
  LaunchFn adjoint_local_wave;     // K4, one wavefront per element (hex8 only, else null)
  LaunchFn adjoint_local_closed;   // K4 in the model's closed form, eight elements per wavefront (hex8 models with one, else null)
  LaunchFn param_gradient_closed;  // K5 likewise
  LaunchFn param_gradient_wave;    // K5, one wavefront per element (hex8 only, else null)
  LaunchFn qoi;                // K6 (hex8: eight elements per wavefront)
  LaunchFn qoi_slot;           // K6, one lane per point of a lane group (any element type)
  hipError_t (*shape_tables)(MeshTables const&, double* tab, int nelems, hipStream_t);  // cached shape tables of the wave kernels (hex8, else null)
  int shape_stride;            // doubles per element in that table
  GatherFn gather_rows;        // staged assembly: node rows summed from the element-major stage
  NodeRowsFn node_rows;        // K1 and K3, one wavefront per node, no stage (hex8 models with a closed form, else null)
  int stage_stride;            // doubles per element in the stage
  bool adjoint_slot_stages;    // the slot-per-lane K3 can store into the stage (it transposes through LDS first)
  bool can_stage;              // staged (gather) assembly available for this element type
};

// registry keyed like the reference's string factories
// (global_residual.cpp:620-630, local_residual.cpp:893-933)
KernelSet get_kernels(int elem_type, int model);

}  // namespace c8
