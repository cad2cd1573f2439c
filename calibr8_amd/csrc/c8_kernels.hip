// c8_kernels.hip -- gfx950 kernels: one lane group (E::NDOF lanes) per element,
// 64/NDOF elements per wavefront, 4 wavefronts per workgroup.  No workgroup barrier is
// ever used: groups are independent and local Newton iteration counts differ, so all
// cooperation is wave-synchronous through LDS (in-order per wave) with compiler fences.
#include <hip/hip_runtime.h>

#include "c8_assemble.hpp"
#include "c8_kernels.hpp"

namespace c8 {

constexpr int BLOCK = 256;

template <class Lane> struct GpuExec {
  int k;
  Lane& L;
  __device__ __forceinline__ GpuExec(int k_, Lane& l) : k(k_), L(l) {}
  template <class F> __device__ __forceinline__ void each(F f) { f(k); }
  __device__ __forceinline__ Lane& lane(int) { return L; }
  template <class F> __device__ __forceinline__ bool any(F f) { return f(k); }
  __device__ __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  __device__ __forceinline__ void add(double* p, double v, int atomic) {
    if (atomic) unsafeAtomicAdd(p, v);  // global_atomic_add_f64
    else *p += v;
  }
  __device__ __forceinline__ void flag(int* s) { atomicOr(s, 1); }
};

// XCD-aware block remap: the dispatcher deals workgroups round-robin over the 8 XCDs
// (blocks b and b+8 share an XCD, MI355X_MICROARCH.md), so give each XCD one contiguous
// chunk of the element order: neighbouring elements then meet in the same L2.
__device__ __forceinline__ int xcd_block(int b, int nblocks) {
  int const chunk = (nblocks + 7) >> 3;
  return (b & 7) * chunk + (b >> 3);
}

template <class E, template <class> class ModelT>
__global__ void __launch_bounds__(BLOCK) k_forward_jacobian(MeshTables mt, ModelSettings ms, FieldArgs fa, SystemArgs sa,
                                                          int first, int count, int nblocks) {
  constexpr int GPB = BLOCK / E::NDOF;
  using Lane = ForwardLane<E, ModelT>;
  __shared__ GroupShared<E, ModelT<Dual>::NLOC> shs[GPB];
  int const lb = xcd_block(blockIdx.x, nblocks);
  if (lb >= nblocks) return;
  int const gib = threadIdx.x / E::NDOF, k = threadIdx.x % E::NDOF;
  int const gi = lb * GPB + gib;
  if (gi >= count) return;
  int const e = mt.order ? mt.order[first + gi] : first + gi;
  Lane L;
  GpuExec<Lane> ex(k, L);
  forward_jacobian_element<E, ModelT>(ex, shs[gib], mt, ms, fa, sa, e);
}

template <class E, template <class> class ModelT>
static hipError_t launch_forward(LaunchArgs const& a) {
  constexpr int GPB = BLOCK / E::NDOF;
  int const nblocks = (a.count + GPB - 1) / GPB;
  int const grid = ((nblocks + 7) / 8) * 8;
  if (a.count <= 0) return hipSuccess;
  hipLaunchKernelGGL((k_forward_jacobian<E, ModelT>), dim3(grid), dim3(BLOCK), 0, a.stream, a.mt, a.ms, a.fa, a.sa,
                     a.first, a.count, nblocks);
  return hipGetLastError();
}

template <class E> static KernelSet kernel_set_for(int model) {
  KernelSet ks{};
  switch (model) {
    case MODEL_ELASTIC: ks.forward_jacobian = &launch_forward<E, Elastic>; break;
    case MODEL_SMALL_J2: ks.forward_jacobian = &launch_forward<E, SmallJ2>; break;
    case MODEL_HYPER_J2: ks.forward_jacobian = &launch_forward<E, HyperJ2>; break;
  }
  return ks;
}

KernelSet get_kernels(int elem_type, int model) {
  if (elem_type == C8_HEX8) return kernel_set_for<Elem<C8_HEX8>>(model);
  if (elem_type == C8_TET4) return kernel_set_for<Elem<C8_TET4>>(model);
  return KernelSet{};
}

}  // namespace c8
