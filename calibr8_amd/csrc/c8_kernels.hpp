// c8_kernels.hpp -- launcher table between the C ABI (c8_api.hip) and the kernels.
#pragma once

#include <hip/hip_runtime_api.h>

#include "c8_assemble.hpp"

namespace c8 {

enum { MODEL_NONE = -1, MODEL_ELASTIC = 0, MODEL_SMALL_J2 = 1, MODEL_HYPER_J2 = 2 };

struct LaunchArgs {
  MeshTables mt;
  ModelSettings ms;
  FieldArgs fa;
  SystemArgs sa;
  int first, count;  // range of the element order to process
  hipStream_t stream;
};

typedef hipError_t (*LaunchFn)(LaunchArgs const&);

struct KernelSet {
  LaunchFn forward_jacobian;
};

// registry keyed like the reference's string factories
// (global_residual.cpp:620-630, local_residual.cpp:893-933)
KernelSet get_kernels(int elem_type, int model);

}  // namespace c8
