// c8_kernels.hip -- gfx950 kernels: one lane group (E::NDOF lanes) per element,
// 64/NDOF elements per wavefront, 4 wavefronts per workgroup.  No workgroup barrier is
// ever used: groups are independent and local Newton iteration counts differ, so all
// cooperation is wave-synchronous through LDS (in-order per wave) with compiler fences.
#include <hip/hip_runtime.h>

#include "c8_assemble_adjoint.hpp"
#include "c8_assemble_wave.hpp"
#include "c8_assemble_node.hpp"
#include "c8_kernels.hpp"

namespace c8 {

#ifdef C8_TUNE_VECTOR_WIB  // tuning build (same results): the wave's index in its block as a vector value
#define C8_WAVE_IN_BLOCK(WPB) (threadIdx.x >> 6)
#else
#define C8_WAVE_IN_BLOCK(WPB) ((WPB) == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6))
#endif
#ifndef C8_BLOCK
#define C8_BLOCK 64
#endif
constexpr int BLOCK = C8_BLOCK;

struct NoLane {};  // kernels whose lanes keep nothing between the phases
template <class Lane> struct GpuExec {
  int k;
  Lane& L;
  __device__ __forceinline__ GpuExec(int k_, Lane& l) : k(k_), L(l) {}
  template <class F> __device__ __forceinline__ void each(F f) { f(k); }
  __device__ __forceinline__ Lane& lane(int) { return L; }
  template <class F> __device__ __forceinline__ bool any(F f) { return f(k); }
  template <class F> __device__ __forceinline__ bool any_wave(F f) { return __any(f(k)) != 0; }
  // callable INSIDE each(): true in every active lane if x is true in any of them -- a scalar branch around work that
  // is a no-op for the lanes where x is false (the emulator simply returns x)
  __device__ __forceinline__ bool uniform_any(bool x) { return __any(x) != 0; }
  // the value of f at the first active lane, in every lane
  template <class F> __device__ __forceinline__ int first_lane(F f) { return __builtin_amdgcn_readfirstlane(f(k)); }
  // Lanes of one wavefront exchange data through LDS.  LDS instructions of a wave execute in order, so
  // only the compiler has to be kept from moving LDS accesses across the exchange point: a
  // wavefront-scope fence does that without draining the memory counters (a workgroup-scope fence
  // also waits for outstanding global loads/stores/atomics).
  __device__ __forceinline__ void sync() {
#ifdef C8_SYNC_WORKGROUP
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
  }
  __device__ __forceinline__ void add(double* p, double v, int atomic) {
    if (atomic) unsafeAtomicAdd(p, v);  // global_atomic_add_f64
    else *p += v;
  }
  __device__ __forceinline__ void flag(int* s) { atomicOr(s, 1); }
  // *p += v for p in LDS, without a return value: ds_add_f64, no vector ALU instruction.  Adds of one wavefront to one
  // address take effect in program order.
  __device__ __forceinline__ void lds_add(double* p, double v) { unsafeAtomicAdd(p, v); }
  // get(l) evaluated in lane S of the caller's group of 8 lanes (lanes 8g .. 8g+7), returned to every lane of the
  // group: two DPP row broadcasts (lane S and lane 8+S of each row of 16) and a select per 32-bit half, no LDS and no
  // barrier.  The source lane must be active whenever a reader is (a group is active or inactive as a whole wherever
  // this is used).  (ds_bpermute_b32 instead of DPP: measured slower on every model.)
  template <int S, class F> __device__ __forceinline__ double bcast8(int lane, F get) {
    double const x = get(lane);
    int const lo = __double2loint(x), hi = __double2hiint(x);
    int const a_lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + S, 0xf, 0xf, false);  // row_newbcast:S
    int const a_hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + S, 0xf, 0xf, false);
    int const b_lo = __builtin_amdgcn_update_dpp(0, lo, 0x158 + S, 0xf, 0xf, false);  // row_newbcast:8+S
    int const b_hi = __builtin_amdgcn_update_dpp(0, hi, 0x158 + S, 0xf, 0xf, false);
    bool const upper = (k & 8) != 0;
    return __hiloint2double(upper ? b_hi : a_hi, upper ? b_lo : a_lo);
  }
  // get(l) of lane l ^ 32 (the other half of the wavefront)
  template <class F> __device__ __forceinline__ double xor32(int lane, F get) { return __shfl_xor(get(lane), 32); }
  // lanes 0..31: getA(lane) + getA(lane + 32); lanes 32..63: getB(lane - 32) + getB(lane).  v_permlane32_swap (gfx950)
  // exchanges the upper half of one register with the lower half of another: two swaps per double, then one add.
  template <class FA, class FB> __device__ __forceinline__ double pair_sum32(int lane, FA getA, FB getB) {
    double const a = getA(lane), b = getB(lane);
#ifdef C8_TUNE_NO_SWAP32  // tuning build (same results): the exchange through ds_bpermute and selects
    double const recv = __shfl_xor(lane < 32 ? b : a, 32);
    return lane < 32 ? a + __shfl_xor(a, 32) + 0. * recv : __shfl_xor(b, 32) + b;
#endif
    auto const lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto const hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
  }
#ifdef C8_STAMPS
  // row-per-node kernel: node n is sampled when n % 244 == 0; the stamp does not wait for outstanding memory operations
  __device__ __forceinline__ void stamp_node(unsigned long long* stamps, int n, int i) {
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    if (k == 0 && stamps && (n % 244) == 0 && n / 244 < 4096) stamps[(size_t)(n / 244) * 16 + i] = t;
  }
  // element e is sampled when e % 244 == 0 (4096 samples over a 1M-element mesh)
  __device__ __forceinline__ void stamp(SystemArgs const& sa, int e, int i) {
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    if (k == 0 && sa.stamps && (e % 244) == 0 && e / 244 < 4096) sa.stamps[(size_t)(e / 244) * 16 + i] = t;
  }
#endif
};

// XCD-aware block remap: the dispatcher deals workgroups round-robin over the 8 XCDs
// (blocks b and b+8 share an XCD, MI355X_MICROARCH.md), so give each XCD one contiguous
// chunk of the element order: neighbouring elements then meet in the same L2.
// stripe > 0: XCD x takes every 8th STRIPE of `stripe` consecutive blocks instead of one contiguous eighth -- the element
// order is then dealt out evenly and no XCD is left with the expensive (plastic) end of the mesh; the grid must be a
// multiple of 8 * stripe (stripe_grid).  The kernels that add into shared rows with atomics keep the contiguous eighths
// (their neighbours meet in one L2); the staged Jacobian kernels and the per-element ones use stripes: 11.02 against
// 11.23-11.32 ms per assembly (gpurun_out/tune_xcd.log).
#ifndef C8_TUNE_STAGE_STRIPE
#define C8_TUNE_STAGE_STRIPE 32
#endif
constexpr int STAGE_STRIPE = C8_TUNE_STAGE_STRIPE;
inline int stripe_grid(int nblocks, int stripe) { return stripe ? ((nblocks + 8 * stripe - 1) / (8 * stripe)) * (8 * stripe) : ((nblocks + 7) / 8) * 8; }
__device__ __forceinline__ int xcd_stripe(int b, int stripe) {
  int const x = b & 7, k = b >> 3;  // k-th block of XCD x
  int const s = stripe > 0 ? stripe : 1;
  return ((k / s) * 8 + x) * s + (k % s);
}
__device__ __forceinline__ int xcd_block(int b, int nblocks) {
#if defined(C8_TUNE_NO_XCD_REMAP)   // tuning build (same results): blocks in launch order, round-robin over the XCDs
  return b;
#elif defined(C8_TUNE_XCD_STRIPE)   // tuning build (same results): XCD x takes every 8th stripe of C8_TUNE_XCD_STRIPE blocks
  int const S = C8_TUNE_XCD_STRIPE;
  int const x = b & 7, k = b >> 3;        // k-th block of XCD x
  return ((k / S) * 8 + x) * S + (k % S);
#else
  int const chunk = (nblocks + 7) >> 3;
  return (b & 7) * chunk + (b >> 3);
#endif
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
  if (gib >= GPB) return;  // lanes left over when NDOF does not divide the block (tri3: 7 groups of 9)
  int const gi = lb * GPB + gib;
  if (gi >= count) return;
  int const e = mt.order ? mt.order[first + gi] : first + gi;
  Lane L;
  GpuExec<Lane> ex(k, L);
  forward_jacobian_element<E, ModelT>(ex, shs[gib], mt, ms, fa, sa, e);
}

// K1 in lane groups with the model's closed form (Model::HAS_CLOSED_FORM) in place of the local Newton iteration and the AD
// passes of the first ip set
template <class E, template <class> class ModelT>
__global__ void __launch_bounds__(BLOCK) k_forward_jacobian_closed(MeshTables mt, ModelSettings ms, FieldArgs fa, SystemArgs sa,
                                                                 int first, int count, int nblocks) {
  constexpr int GPB = BLOCK / E::NDOF;
  using Lane = ForwardLane<E, ModelT>;
  __shared__ GroupShared<E, ModelT<Dual>::NLOC> shs[GPB];
  int const lb = xcd_block(blockIdx.x, nblocks);
  if (lb >= nblocks) return;
  int const gib = threadIdx.x / E::NDOF, k = threadIdx.x % E::NDOF;
  if (gib >= GPB) return;
  int const gi = lb * GPB + gib;
  if (gi >= count) return;
  int const e = mt.order ? mt.order[first + gi] : first + gi;
  Lane L;
  GpuExec<Lane> ex(k, L);
  forward_jacobian_element<E, ModelT, true>(ex, shs[gib], mt, ms, fa, sa, e);
}

template <class E, template <class> class ModelT>
static hipError_t launch_forward(LaunchArgs const& a) {
  constexpr int GPB = BLOCK / E::NDOF;
  int const nblocks = (a.count + GPB - 1) / GPB;
  int const grid = ((nblocks + 7) / 8) * 8;
  if (a.count <= 0) return hipSuccess;
  if constexpr (has_closed_form<ModelT<Dual>>::value) {
    if (a.ms.closed_form_slot) {
      hipLaunchKernelGGL((k_forward_jacobian_closed<E, ModelT>), dim3(grid), dim3(BLOCK), 0, a.stream, a.mt, a.ms, a.fa, a.sa,
                         a.first, a.count, nblocks);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL((k_forward_jacobian<E, ModelT>), dim3(grid), dim3(BLOCK), 0, a.stream, a.mt, a.ms, a.fa, a.sa,
                     a.first, a.count, nblocks);
  return hipGetLastError();
}

// One wavefront per workgroup for the wave Jacobian kernels: a workgroup's registers and LDS are released only when
// its last wavefront retires, and element durations differ (elastic / plastic points, Newton iterations), so with four
// wavefronts per workgroup finished slots idle until the slowest is done (12.5 ms against 12.7-12.9 ms).
#ifndef C8_JBLOCK
#define C8_JBLOCK 64
#endif
constexpr int JBLOCK = C8_JBLOCK;
// K1, one wavefront per hex8 element (c8_assemble_wave.hpp)
template <class E, template <class> class ModelT>
__global__ void __launch_bounds__(JBLOCK) __attribute__((amdgpu_waves_per_eu(ModelT<Dual>::WAVE_BLOCKS_PER_CU, ModelT<Dual>::WAVE_BLOCKS_PER_CU)))
k_forward_jacobian_wave(MeshTables mt, ModelSettings ms, FieldArgs fa, SystemArgs sa,
                                                               int first, int count, int nblocks) {
  constexpr int WPB = JBLOCK / 64;
  using Lane = WaveLane<ModelT>;
  __shared__ WaveShared<E, ModelT<Dual>::NLOC, false, ModelT<Dual>::FINITE_DEF> shs[WPB];
  int const lb = (sa.stage && STAGE_STRIPE) ? xcd_stripe(blockIdx.x, STAGE_STRIPE) : xcd_block(blockIdx.x, nblocks);
  if (lb >= nblocks) return;
  // the wave's index in the block is wave-uniform: as a scalar it keeps the element / node number and every address
  // derived from it in SGPRs (one wave per block: zero)
  int const wib = C8_WAVE_IN_BLOCK(WPB), lane = threadIdx.x & 63;
  int const gi = lb * WPB + wib;
  if (gi >= count) return;
  int const e = mt.order ? mt.order[first + gi] : first + gi;
  Lane L;
  GpuExec<Lane> ex(lane, L);
  forward_jacobian_wave<E, ModelT>(ex, shs[wib], mt, ms, fa, sa, e);
}

#ifndef C8_CLOSED_WAVES
#define C8_CLOSED_WAVES 3  // 156 registers, 13.5 KB of LDS per wave: three waves per SIMD measured 0.6 % ahead of two
#endif
// K1 with the model's closed form (Model::HAS_CLOSED_FORM): no local Newton iteration, no local elimination, no AD passes
template <class E, template <class> class ModelT>
__global__ void __launch_bounds__(JBLOCK) __attribute__((amdgpu_waves_per_eu(C8_CLOSED_WAVES, C8_CLOSED_WAVES)))
k_forward_jacobian_wave_closed(MeshTables mt, ModelSettings ms, FieldArgs fa, SystemArgs sa,
                                                                         int first, int count, int nblocks) {
  constexpr int WPB = JBLOCK / 64;
  using Lane = WaveLane<ModelT>;
  __shared__ WaveShared<E, ModelT<Dual>::NLOC, false, ModelT<Dual>::FINITE_DEF, true> shs[WPB];
  int const lb = (sa.stage && STAGE_STRIPE) ? xcd_stripe(blockIdx.x, STAGE_STRIPE) : xcd_block(blockIdx.x, nblocks);
  if (lb >= nblocks) return;
  // the wave's index in the block is wave-uniform: as a scalar it keeps the element / node number and every address
  // derived from it in SGPRs (one wave per block: zero)
  int const wib = C8_WAVE_IN_BLOCK(WPB), lane = threadIdx.x & 63;
  int const gi = lb * WPB + wib;
  if (gi >= count) return;
  int const e = mt.order ? mt.order[first + gi] : first + gi;
  Lane L;
  GpuExec<Lane> ex(lane, L);
  forward_jacobian_wave_closed<E, ModelT>(ex, shs[wib], mt, ms, fa, sa, e);
}

template <class E, template <class> class ModelT>
static hipError_t launch_forward_wave(LaunchArgs const& a) {
  constexpr int WPB = JBLOCK / 64;
  int const nblocks = (a.count + WPB - 1) / WPB;
  int const grid = a.sa.stage ? stripe_grid(nblocks, STAGE_STRIPE) : ((nblocks + 7) / 8) * 8;
  if (a.count <= 0) return hipSuccess;
  if constexpr (has_closed_form<ModelT<Dual>>::value) {
    if (a.ms.closed_form) {
      hipLaunchKernelGGL((k_forward_jacobian_wave_closed<E, ModelT>), dim3(grid), dim3(JBLOCK), 0, a.stream, a.mt, a.ms, a.fa, a.sa,
                         a.first, a.count, nblocks);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL((k_forward_jacobian_wave<E, ModelT>), dim3(grid), dim3(JBLOCK), 0, a.stream, a.mt, a.ms, a.fa, a.sa,
                     a.first, a.count, nblocks);
  return hipGetLastError();
}

template <class E, template <class> class ModelT>
__global__ void __launch_bounds__(JBLOCK) __attribute__((amdgpu_waves_per_eu(ModelT<Dual>::WAVE_BLOCKS_PER_CU, ModelT<Dual>::WAVE_BLOCKS_PER_CU)))
k_adjoint_jacobian_wave(MeshTables mt, ModelSettings ms, FieldArgs fa, AdjointArgs aa,
                                                                  SystemArgs sa, int first, int count, int nblocks) {
  constexpr int WPB = JBLOCK / 64;
  using Lane = WaveLane<ModelT>;
  __shared__ WaveShared<E, ModelT<Dual>::NLOC, true, ModelT<Dual>::FINITE_DEF> shs[WPB];
  int const lb = (sa.stage && STAGE_STRIPE) ? xcd_stripe(blockIdx.x, STAGE_STRIPE) : xcd_block(blockIdx.x, nblocks);
  if (lb >= nblocks) return;
  // the wave's index in the block is wave-uniform: as a scalar it keeps the element / node number and every address
  // derived from it in SGPRs (one wave per block: zero)
  int const wib = C8_WAVE_IN_BLOCK(WPB), lane = threadIdx.x & 63;
  int const gi = lb * WPB + wib;
  if (gi >= count) return;
  int const e = mt.order ? mt.order[first + gi] : first + gi;
  Lane L;
  GpuExec<Lane> ex(lane, L);
  adjoint_jacobian_wave<E, ModelT, PointQoi>(ex, shs[wib], mt, ms, fa, aa, sa, e);
}

template <class E, template <class> class ModelT>
static hipError_t launch_adjoint_jacobian_wave(LaunchArgs const& a) {
  constexpr int WPB = JBLOCK / 64;
  int const nblocks = (a.count + WPB - 1) / WPB;
  int const grid = a.sa.stage ? stripe_grid(nblocks, STAGE_STRIPE) : ((nblocks + 7) / 8) * 8;
  if (a.count <= 0) return hipSuccess;
  hipLaunchKernelGGL((k_adjoint_jacobian_wave<E, ModelT>), dim3(grid), dim3(JBLOCK), 0, a.stream, a.mt, a.ms, a.fa, a.aa,
                     a.sa, a.first, a.count, nblocks);
  return hipGetLastError();
}

#ifdef C8_TUNE_K4_WAVES  // tuning build (same results): waves per SIMD of the local-adjoint kernel, every model
#define C8_K4_WAVES(M) C8_TUNE_K4_WAVES
#else
#define C8_K4_WAVES(M) M::WAVE_BLOCKS_PER_CU_K4
#endif
template <class E, template <class> class ModelT>
__global__ void __launch_bounds__(JBLOCK) __attribute__((amdgpu_waves_per_eu(C8_K4_WAVES(ModelT<Dual>), C8_K4_WAVES(ModelT<Dual>))))
k_adjoint_local_wave(MeshTables mt, ModelSettings ms, FieldArgs fa, AdjointArgs aa,
                                                               SystemArgs sa, int first, int count, int nblocks) {
  constexpr int WPB = JBLOCK / 64;
  using Lane = WaveLaneA<ModelT>;
  __shared__ WaveSharedA<E, ModelT<Dual>::NLOC> shs[WPB];
  int const lb = STAGE_STRIPE ? xcd_stripe(blockIdx.x, STAGE_STRIPE) : xcd_block(blockIdx.x, nblocks);  // per-element outputs only: stripes
  if (lb >= nblocks) return;
  // the wave's index in the block is wave-uniform: as a scalar it keeps the element / node number and every address
  // derived from it in SGPRs (one wave per block: zero)
  int const wib = C8_WAVE_IN_BLOCK(WPB), lane = threadIdx.x & 63;
  int const gi = lb * WPB + wib;
  if (gi >= count) return;
  Lane L;
  GpuExec<Lane> ex(lane, L);
  adjoint_local_wave<E, ModelT>(ex, shs[wib], mt, ms, fa, aa, sa, first + gi);
}

// CLOSED: the model's closed form of the point's share (Model::closed_form_param_gradient) instead of dual numbers
template <class E, template <class> class ModelT, bool CLOSED = false>
__global__ void __launch_bounds__(BLOCK, CLOSED ? 2 : ModelT<Dual>::WAVE_BLOCKS_PER_CU_ADJ) k_param_gradient_wave(MeshTables mt, ModelSettings ms, FieldArgs fa, AdjointArgs aa, int count) {
  constexpr int WPB = BLOCK / 64;
  using Lane = GradWaveLane<ModelT>;
  __shared__ GradWaveShared<E> shs[WPB];
  // the wave's index in the block is wave-uniform: as a scalar it keeps the element / node number and every address
  // derived from it in SGPRs (one wave per block: zero)
  int const wib = C8_WAVE_IN_BLOCK(WPB), lane = threadIdx.x & 63;
  Lane L;
  L.slot0 = -1;
  C8_UNROLL
  for (int a = 0; a < 8; ++a) L.acc[a] = 0.;
  GpuExec<Lane> ex(lane, L);
  int const ngroups = (count + 7) / 8;
  for (int gidx = blockIdx.x * WPB + wib; gidx < ngroups; gidx += gridDim.x * WPB)
    param_gradient_wave8<E, ModelT, PointQoi, CLOSED>(ex, shs[wib], mt, ms, fa, aa, gidx * 8, (count - gidx * 8 < 8) ? count - gidx * 8 : 8);
  param_gradient_wave8_flush(ex, shs[wib].red, aa);
}

// K4 in closed form (adjoint_local_closed_wave8): eight consecutive elements per wavefront
template <class E, template <class> class ModelT>
__global__ void __launch_bounds__(BLOCK) k_adjoint_local_closed(MeshTables mt, ModelSettings ms, FieldArgs fa, AdjointArgs aa, int first, int count) {
  constexpr int WPB = BLOCK / 64;
  __shared__ GradWaveShared<E> shs[WPB];
  int const wib = C8_WAVE_IN_BLOCK(WPB), lane = threadIdx.x & 63;
  int const gidx = blockIdx.x * WPB + wib;
  if (gidx * 8 >= count) return;
  NoLane L;
  GpuExec<NoLane> ex(lane, L);
  adjoint_local_closed_wave8<E, ModelT>(ex, shs[wib], mt, ms, fa, aa, first + gidx * 8, (count - gidx * 8 < 8) ? count - gidx * 8 : 8);
}
template <class E, template <class> class ModelT> static hipError_t launch_adjoint_local_closed(LaunchArgs const& a) {
  constexpr int WPB = BLOCK / 64;
  if (a.count <= 0) return hipSuccess;
  int const ngroups = (a.count + 7) / 8;
  hipLaunchKernelGGL((k_adjoint_local_closed<E, ModelT>), dim3((ngroups + WPB - 1) / WPB), dim3(BLOCK), 0, a.stream, a.mt, a.ms, a.fa, a.aa, a.first, a.count);
  return hipGetLastError();
}
template <class E, template <class> class ModelT> static hipError_t launch_adjoint_local_wave(LaunchArgs const& a) {
  constexpr int WPB = JBLOCK / 64;
  int const nblocks = (a.count + WPB - 1) / WPB;
  int const grid = stripe_grid(nblocks, STAGE_STRIPE);
  if (a.count <= 0) return hipSuccess;
  hipLaunchKernelGGL((k_adjoint_local_wave<E, ModelT>), dim3(grid), dim3(JBLOCK), 0, a.stream, a.mt, a.ms, a.fa, a.aa, a.sa,
                     a.first, a.count, nblocks);
  return hipGetLastError();
}
template <class E, template <class> class ModelT> static hipError_t launch_param_gradient_wave(LaunchArgs const& a) {
  constexpr int WPB = BLOCK / 64;
  if (a.count <= 0) return hipSuccess;
  int const ngroups = (a.count + 7) / 8;
  int const nblocks = (ngroups + WPB - 1) / WPB;
  int const grid = nblocks < 2048 ? nblocks : 2048;
  hipLaunchKernelGGL((k_param_gradient_wave<E, ModelT>), dim3(grid), dim3(BLOCK), 0, a.stream, a.mt, a.ms, a.fa, a.aa, a.count);
  return hipGetLastError();
}
template <class E, template <class> class ModelT> static hipError_t launch_param_gradient_closed(LaunchArgs const& a) {
  constexpr int WPB = BLOCK / 64;
  if (a.count <= 0) return hipSuccess;
  int const ngroups = (a.count + 7) / 8;
  int const nblocks = (ngroups + WPB - 1) / WPB;
  int const grid = nblocks < 2048 ? nblocks : 2048;
  hipLaunchKernelGGL((k_param_gradient_wave<E, ModelT, true>), dim3(grid), dim3(BLOCK), 0, a.stream, a.mt, a.ms, a.fa, a.aa, a.count);
  return hipGetLastError();
}

// cached shape tables of the wave kernels: one wavefront per element, once per context
template <class E> __global__ void __launch_bounds__(64) k_shape_tables(MeshTables mt, double* tab, int nelems) {
  __shared__ ShapeShared<E> sh;
  int const e = blockIdx.x;
  if (e >= nelems) return;
  ShapeLane L;
  GpuExec<ShapeLane> ex(threadIdx.x, L);
  store_shape_tables<E>(ex, sh, mt, tab, e);
}
template <class E> static hipError_t launch_shape_tables(MeshTables const& mt, double* tab, int nelems, hipStream_t stream) {
  if (nelems <= 0) return hipSuccess;
  hipLaunchKernelGGL((k_shape_tables<E>), dim3(nelems), dim3(64), 0, stream, mt, tab, nelems);
  return hipGetLastError();
}

// K2 for hex8: eight elements per wavefront, 32 per workgroup (residual_wave8)
template <class E, template <class> class ModelT>
__global__ void __launch_bounds__(BLOCK, 2) k_residual_wave(MeshTables mt, ModelSettings ms, FieldArgs fa, SystemArgs sa,
                                                          int count, int nblocks) {
  constexpr int WPB = BLOCK / 64;
  using Lane = ResidualWaveLane<ModelT>;
  __shared__ ResidualWaveShared<E> shs[WPB];
  int const lb = xcd_block(blockIdx.x, nblocks);
  if (lb >= nblocks) return;
  // the wave's index in the block is wave-uniform: as a scalar it keeps the element / node number and every address
  // derived from it in SGPRs (one wave per block: zero)
  int const wib = C8_WAVE_IN_BLOCK(WPB), lane = threadIdx.x & 63;
  int const e0 = (lb * WPB + wib) * 8;
  if (e0 >= count) return;
  Lane L;
  GpuExec<Lane> ex(lane, L);
  residual_wave8<E, ModelT>(ex, shs[wib], mt, ms, fa, sa, e0, (count - e0 < 8) ? count - e0 : 8);
}
template <class E, template <class> class ModelT> static hipError_t launch_residual_wave(LaunchArgs const& a) {
  constexpr int EPB = (BLOCK / 64) * 8;
  if (a.count <= 0) return hipSuccess;
  int const nblocks = (a.count + EPB - 1) / EPB;
  int const grid = ((nblocks + 7) / 8) * 8;
  hipLaunchKernelGGL((k_residual_wave<E, ModelT>), dim3(grid), dim3(BLOCK), 0, a.stream, a.mt, a.ms, a.fa, a.sa, a.count, nblocks);
  return hipGetLastError();
}

// staged assembly: one wavefront per node sums the node's rows from the element-major stage
template <class E, int MAXDEG>
__global__ void __launch_bounds__(BLOCK) k_gather_rows(GatherArgs ga, int first, int count, int nblocks) {
  constexpr int WPB = BLOCK / 64;
  __shared__ GatherShared<E, MAXDEG> shs[WPB];
  int const lb = xcd_block(blockIdx.x, nblocks);
  if (lb >= nblocks) return;
  // the wave's index in the block is wave-uniform: as a scalar it keeps the element / node number and every address
  // derived from it in SGPRs (one wave per block: zero)
  int const wib = C8_WAVE_IN_BLOCK(WPB), lane = threadIdx.x & 63;
  int const gi = lb * WPB + wib;
  if (gi >= count) return;
  GatherLane<E, MAXDEG> L;
  GpuExec<GatherLane<E, MAXDEG>> ex(lane, L);
  gather_node_rows<E, MAXDEG>(ex, shs[wib], ga, ga.node_order[first + gi]);
}
template <class E> static hipError_t launch_gather_rows(GatherArgs const& ga, int first, int count, int max_degree, hipStream_t stream) {
  constexpr int WPB = BLOCK / 64;
  if (count <= 0) return hipSuccess;
  int const nblocks = (count + WPB - 1) / WPB;
  int const grid = ((nblocks + 7) / 8) * 8;
  if (max_degree <= 32)
    hipLaunchKernelGGL((k_gather_rows<E, 32>), dim3(grid), dim3(BLOCK), 0, stream, ga, first, count, nblocks);
  else if (max_degree <= GATHER_MAX_DEGREE)
    hipLaunchKernelGGL((k_gather_rows<E, GATHER_MAX_DEGREE>), dim3(grid), dim3(BLOCK), 0, stream, ga, first, count, nblocks);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

// K1, one wavefront per node (c8_assemble_node.hpp): the node's rows formed from its elements with the model's closed form
// and written once -- no element stage.  XCD x takes every 8th stripe of NODE_STRIPE consecutive nodes: neighbouring nodes
// share their elements' shape tables and state in one L2, and the eight XCDs work on one region of the mesh at a time.
// Measured on the 100^3 brick (K1 ms / K3 ms / HBM-side GB per assembly with FETCH_SIZE doubled): stripes of 64: 4.37 / 5.20 /
// 16.9; 256: 4.33 / 5.12 / 13.9; 512: 4.36 / 5.21 / 13.0; 1024: 4.36 / 5.30 / 12.5; 2048: 4.36 / 5.28 / 12.3; 16384 and one
// contiguous eighth of the nodes per XCD: 4.71 / - / 12.1 (profiles/README.md, round 3).
#ifndef C8_NODE_WAVES
#define C8_NODE_WAVES 3
#endif
#ifndef C8_TUNE_NODE_STRIPE
#define C8_TUNE_NODE_STRIPE 256
#endif
constexpr int NODE_STRIPE = C8_TUNE_NODE_STRIPE;
template <class E, template <class> class ModelT, int MAXDEG, bool MANY>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MANY ? 2 : C8_NODE_WAVES, 4)))
k_node_rows_closed(MeshTables mt, ModelSettings ms, FieldArgs fa, GatherArgs ga, int first, int count, int nblocks) {
  using Lane = NodeLane<MAXDEG>;
  __shared__ NodeShared<E, ModelT<Dual>, MAXDEG, MANY> sh;
  int const lb = xcd_stripe(blockIdx.x, NODE_STRIPE);
  if (lb >= count) return;
  Lane L;
  GpuExec<Lane> ex(threadIdx.x, L);
  node_rows_closed<E, ModelT, MAXDEG, MANY>(ex, sh, mt, ms, fa, ga, first + lb);  // nodes [first, first + count)
}
// K3 in the same form (objective "average displacement"): the transposed rows and the adjoint right-hand side
template <class E, template <class> class ModelT, int MAXDEG, bool MANY>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MANY ? 2 : C8_NODE_WAVES, 4)))
k_node_rows_closed_adjoint(MeshTables mt, ModelSettings ms, FieldArgs fa, AdjointArgs aa, GatherArgs ga, int first, int count, int nblocks) {
  using Lane = NodeLane<MAXDEG>;
  __shared__ NodeShared<E, ModelT<Dual>, MAXDEG, MANY> sh;
  int const lb = xcd_stripe(blockIdx.x, NODE_STRIPE);
  if (lb >= count) return;
  Lane L;
  GpuExec<Lane> ex(threadIdx.x, L);
  node_rows_closed<E, ModelT, MAXDEG, MANY, true>(ex, sh, mt, ms, fa, ga, first + lb, aa);
}
// max_node_elems <= 8 (every hex8 mesh cut out of a structured one, most others): the lean form; otherwise a node's
// elements go through the kernel eight at a time
template <class E, template <class> class ModelT>
__global__ void __launch_bounds__(256) k_node_rows_update_g(MeshTables mt, AdjointArgs aa, size_t npoints) {
  size_t const qp = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (qp < npoints) node_rows_update_g<E, ModelT>(mt, aa, qp);
}
// first = -1: the update of g that closes an adjoint assembly under the calibration objective (count = elements)
template <class E, template <class> class ModelT>
static hipError_t launch_node_rows(MeshTables const& mt, ModelSettings const& ms, FieldArgs const& fa, AdjointArgs const* aa, GatherArgs const& ga,
                                   int first, int count, int max_degree, int max_node_elems, hipStream_t stream) {
  if (count <= 0) return hipSuccess;
  if (first < 0) {
    size_t const npoints = (size_t)count * E::NP0;
    hipLaunchKernelGGL((k_node_rows_update_g<E, ModelT>), dim3((unsigned)((npoints + 255) / 256)), dim3(256), 0, stream, mt, *aa, npoints);
    return hipGetLastError();
  }
  int const nblocks = count;
  int const grid = stripe_grid(nblocks, NODE_STRIPE);
  bool const lean = max_degree <= 32 && max_node_elems <= 8;
  if (max_degree > GATHER_MAX_DEGREE) return hipErrorInvalidValue;
  if (aa) {  // adjoint assembly
    if (lean) hipLaunchKernelGGL((k_node_rows_closed_adjoint<E, ModelT, 32, false>), dim3(grid), dim3(64), 0, stream, mt, ms, fa, *aa, ga, first, count, nblocks);
    else hipLaunchKernelGGL((k_node_rows_closed_adjoint<E, ModelT, GATHER_MAX_DEGREE, true>), dim3(grid), dim3(64), 0, stream, mt, ms, fa, *aa, ga, first, count, nblocks);
  } else {
    if (lean) hipLaunchKernelGGL((k_node_rows_closed<E, ModelT, 32, false>), dim3(grid), dim3(64), 0, stream, mt, ms, fa, ga, first, count, nblocks);
    else hipLaunchKernelGGL((k_node_rows_closed<E, ModelT, GATHER_MAX_DEGREE, true>), dim3(grid), dim3(64), 0, stream, mt, ms, fa, ga, first, count, nblocks);
  }
  return hipGetLastError();
}
template <class E, template <class> class ModelT, class = void> struct NodeKernel {
  static NodeRowsFn get() { return nullptr; }
  static LaunchFn get_adjoint_local() { return nullptr; }
  static LaunchFn get_param_gradient() { return nullptr; }
};
template <template <class> class ModelT>
struct NodeKernel<Elem<C8_HEX8>, ModelT, std::enable_if_t<has_closed_form_rows<ModelT<Dual>>::value>> {
  static NodeRowsFn get() { return &launch_node_rows<Elem<C8_HEX8>, ModelT>; }
  static LaunchFn get_adjoint_local() { return &launch_adjoint_local_closed<Elem<C8_HEX8>, ModelT>; }
  static LaunchFn get_param_gradient() { return &launch_param_gradient_closed<Elem<C8_HEX8>, ModelT>; }
};

// group index -> element for the colour-batched / atomic element-parallel kernels
#define C8_GROUP_PROLOGUE(E)                                                  \
  constexpr int GPB = BLOCK / E::NDOF;                                        \
  int const lb = xcd_block(blockIdx.x, nblocks);                              \
  if (lb >= nblocks) return;                                                  \
  int const gib = threadIdx.x / E::NDOF, k = threadIdx.x % E::NDOF;           \
  if (gib >= GPB) return; /* lanes left over when NDOF does not divide the block (tri3: 7 groups of 9) */ \
  int const gi = lb * GPB + gib;                                              \
  if (gi >= count) return;                                                    \
  int const e = mt.order ? mt.order[first + gi] : first + gi;

template <class E, template <class> class ModelT>
__global__ void __launch_bounds__(BLOCK) k_residual(MeshTables mt, ModelSettings ms, FieldArgs fa, SystemArgs sa,
                                                  int first, int count, int nblocks) {
  C8_GROUP_PROLOGUE(E)
  using Lane = ResidualLane<E, ModelT>;
  __shared__ GroupShared<E, ModelT<Dual>::NLOC> shs[GPB];
  Lane L;
  GpuExec<Lane> ex(k, L);
  residual_element<E, ModelT>(ex, shs[gib], mt, ms, fa, sa, e);
}

template <class E, template <class> class ModelT>
__global__ void __launch_bounds__(BLOCK) k_adjoint_jacobian(MeshTables mt, ModelSettings ms, FieldArgs fa, AdjointArgs aa,
                                                          SystemArgs sa, int first, int count, int nblocks) {
  C8_GROUP_PROLOGUE(E)
  using Lane = AdjointLane<E, ModelT>;
  __shared__ GroupShared<E, ModelT<Dual>::NLOC> shs[GPB];
  Lane L;
  GpuExec<Lane> ex(k, L);
  adjoint_jacobian_element<E, ModelT, PointQoi>(ex, shs[gib], mt, ms, fa, aa, sa, e);
}

template <class E, template <class> class ModelT>
__global__ void __launch_bounds__(BLOCK) k_adjoint_local(MeshTables mt, ModelSettings ms, FieldArgs fa, AdjointArgs aa,
                                                       SystemArgs sa, int first, int count, int nblocks) {
  C8_GROUP_PROLOGUE(E)
  using Lane = AdjointLane<E, ModelT>;
  __shared__ GroupShared<E, ModelT<Dual>::NLOC> shs[GPB];
  Lane L;
  GpuExec<Lane> ex(k, L);
  adjoint_local_element<E, ModelT>(ex, shs[gib], mt, ms, fa, aa, sa, e);
}

// K5/K6: each group walks the elements with a grid stride and adds its sums once at the end
template <class E, template <class> class ModelT>
__global__ void __launch_bounds__(BLOCK) k_param_gradient(MeshTables mt, ModelSettings ms, FieldArgs fa, AdjointArgs aa, int count) {
  constexpr int GPB = BLOCK / E::NDOF;
  using Lane = GradLane<E, ModelT>;
  __shared__ GroupShared<E, ModelT<Dual>::NLOC> shs[GPB];
  int const gib = threadIdx.x / E::NDOF, k = threadIdx.x % E::NDOF;
  if (gib >= GPB) return;
  Lane L;
  L.slot = -1;
  L.acc = 0.;
  GpuExec<Lane> ex(k, L);
  for (int e = blockIdx.x * GPB + gib; e < count; e += gridDim.x * GPB)
    param_gradient_element<E, ModelT, PointQoi>(ex, shs[gib], mt, ms, fa, aa, e);
  param_gradient_flush(ex, aa);
}

template <class E, template <class> class ModelT>
__global__ void __launch_bounds__(BLOCK) k_qoi(MeshTables mt, FieldArgs fa, AdjointArgs aa, int count) {
  constexpr int GPB = BLOCK / E::NDOF;
  using Lane = QoiLane<E, ModelT>;
  __shared__ GroupShared<E, ModelT<Dual>::NLOC> shs[GPB];
  int const gib = threadIdx.x / E::NDOF, k = threadIdx.x % E::NDOF;
  if (gib >= GPB) return;
  Lane L;
  L.acc = 0.;
  GpuExec<Lane> ex(k, L);
  for (int e = blockIdx.x * GPB + gib; e < count; e += gridDim.x * GPB)
    qoi_element<E, ModelT, PointQoi>(ex, shs[gib], mt, fa, aa.qoi, e);
  qoi_flush<E>(ex, aa.out);
}

// K6 for hex8: eight elements per wavefront (qoi_wave8), grid-stride, one add per wavefront
template <class E, template <class> class ModelT>
__global__ void __launch_bounds__(BLOCK, 2) k_qoi_wave(MeshTables mt, FieldArgs fa, AdjointArgs aa, int count) {
  constexpr int WPB = BLOCK / 64;
  using Lane = QoiWaveLane<ModelT>;
  __shared__ GradWaveShared<E> shs[WPB];
  // the wave's index in the block is wave-uniform: as a scalar it keeps the element / node number and every address
  // derived from it in SGPRs (one wave per block: zero)
  int const wib = C8_WAVE_IN_BLOCK(WPB), lane = threadIdx.x & 63;
  Lane L;
  L.acc = 0.;
  GpuExec<Lane> ex(lane, L);
  int const ngroups = (count + 7) / 8;
  for (int gidx = blockIdx.x * WPB + wib; gidx < ngroups; gidx += gridDim.x * WPB)
    qoi_wave8<E, ModelT, PointQoi>(ex, shs[wib], mt, fa, aa.qoi, gidx * 8, (count - gidx * 8 < 8) ? count - gidx * 8 : 8);
  qoi_wave8_flush(ex, shs[wib].red, aa.out);
}
template <class E, template <class> class ModelT> static hipError_t launch_qoi_wave(LaunchArgs const& a) {
  constexpr int WPB = BLOCK / 64;
  if (a.count <= 0) return hipSuccess;
  int const ngroups = (a.count + 7) / 8;
  int const nblocks = (ngroups + WPB - 1) / WPB;
  int const grid = nblocks < 2048 ? nblocks : 2048;
  hipLaunchKernelGGL((k_qoi_wave<E, ModelT>), dim3(grid), dim3(BLOCK), 0, a.stream, a.mt, a.fa, a.aa, a.count);
  return hipGetLastError();
}

template <class E> static void grid_for(LaunchArgs const& a, int& nblocks, int& grid) {
  constexpr int GPB = BLOCK / E::NDOF;
  nblocks = (a.count + GPB - 1) / GPB;
  grid = ((nblocks + 7) / 8) * 8;
}

template <class E, template <class> class ModelT> static hipError_t launch_residual(LaunchArgs const& a) {
  int nblocks, grid;
  grid_for<E>(a, nblocks, grid);
  if (a.count <= 0) return hipSuccess;
  hipLaunchKernelGGL((k_residual<E, ModelT>), dim3(grid), dim3(BLOCK), 0, a.stream, a.mt, a.ms, a.fa, a.sa, a.first, a.count, nblocks);
  return hipGetLastError();
}
template <class E, template <class> class ModelT> static hipError_t launch_adjoint_jacobian(LaunchArgs const& a) {
  int nblocks, grid;
  grid_for<E>(a, nblocks, grid);
  if (a.count <= 0) return hipSuccess;
  hipLaunchKernelGGL((k_adjoint_jacobian<E, ModelT>), dim3(grid), dim3(BLOCK), 0, a.stream, a.mt, a.ms, a.fa, a.aa, a.sa, a.first, a.count, nblocks);
  return hipGetLastError();
}
template <class E, template <class> class ModelT> static hipError_t launch_adjoint_local(LaunchArgs const& a) {
  int nblocks, grid;
  grid_for<E>(a, nblocks, grid);
  if (a.count <= 0) return hipSuccess;
  hipLaunchKernelGGL((k_adjoint_local<E, ModelT>), dim3(grid), dim3(BLOCK), 0, a.stream, a.mt, a.ms, a.fa, a.aa, a.sa, a.first, a.count, nblocks);
  return hipGetLastError();
}
template <class E, template <class> class ModelT> static hipError_t launch_param_gradient(LaunchArgs const& a) {
  int nblocks, grid;
  grid_for<E>(a, nblocks, grid);
  if (a.count <= 0) return hipSuccess;
  grid = nblocks < 2048 ? nblocks : 2048;
  hipLaunchKernelGGL((k_param_gradient<E, ModelT>), dim3(grid), dim3(BLOCK), 0, a.stream, a.mt, a.ms, a.fa, a.aa, a.count);
  return hipGetLastError();
}
template <class E, template <class> class ModelT> static hipError_t launch_qoi(LaunchArgs const& a) {
  int nblocks, grid;
  grid_for<E>(a, nblocks, grid);
  if (a.count <= 0) return hipSuccess;
  grid = nblocks < 2048 ? nblocks : 2048;
  hipLaunchKernelGGL((k_qoi<E, ModelT>), dim3(grid), dim3(BLOCK), 0, a.stream, a.mt, a.fa, a.aa, a.count);
  return hipGetLastError();
}

template <class E, template <class> class ModelT> struct WaveKernel {
  static LaunchFn get() { return nullptr; }
  static LaunchFn get_adjoint() { return nullptr; }
  static LaunchFn get_adjoint_local() { return nullptr; }
  static LaunchFn get_param_gradient() { return nullptr; }
  static LaunchFn get_residual() { return nullptr; }
  static LaunchFn get_qoi() { return nullptr; }
  static constexpr hipError_t (*shape_tables)(MeshTables const&, double*, int, hipStream_t) = nullptr;
};
template <template <class> class ModelT> struct WaveKernel<Elem<C8_HEX8>, ModelT> {
  static LaunchFn get() { return &launch_forward_wave<Elem<C8_HEX8>, ModelT>; }
  static LaunchFn get_adjoint() { return &launch_adjoint_jacobian_wave<Elem<C8_HEX8>, ModelT>; }
  static LaunchFn get_adjoint_local() { return &launch_adjoint_local_wave<Elem<C8_HEX8>, ModelT>; }
  static LaunchFn get_param_gradient() { return &launch_param_gradient_wave<Elem<C8_HEX8>, ModelT>; }
  static LaunchFn get_residual() { return &launch_residual_wave<Elem<C8_HEX8>, ModelT>; }
  static LaunchFn get_qoi() { return &launch_qoi_wave<Elem<C8_HEX8>, ModelT>; }
  static constexpr hipError_t (*shape_tables)(MeshTables const&, double*, int, hipStream_t) = &launch_shape_tables<Elem<C8_HEX8>>;
};

template <class E, template <class> class ModelT> static KernelSet kernel_set() {
  using WK = WaveKernel<E, ModelT>;
  KernelSet ks;
  ks.forward_jacobian = &launch_forward<E, ModelT>;
  ks.forward_jacobian_wave = WK::get();
  ks.adjoint_jacobian_wave = WK::get_adjoint();
  ks.adjoint_local_wave = WK::get_adjoint_local();
  ks.param_gradient_wave = WK::get_param_gradient();
  ks.residual = &launch_residual<E, ModelT>;
  ks.residual_wave = WK::get_residual();
  ks.adjoint_jacobian = &launch_adjoint_jacobian<E, ModelT>;
  ks.adjoint_local = &launch_adjoint_local<E, ModelT>;
  ks.param_gradient = &launch_param_gradient<E, ModelT>;
  ks.qoi = WK::get_qoi() ? WK::get_qoi() : &launch_qoi<E, ModelT>;
  ks.qoi_slot = &launch_qoi<E, ModelT>;
  ks.shape_tables = WK::shape_tables;
  ks.shape_stride = SHAPE_STRIDE;
  ks.gather_rows = &launch_gather_rows<E>;
  ks.node_rows = NodeKernel<E, ModelT>::get();
  ks.adjoint_local_closed = NodeKernel<E, ModelT>::get_adjoint_local();
  ks.param_gradient_closed = NodeKernel<E, ModelT>::get_param_gradient();
  ks.stage_stride = stage_stride<E>();
  ks.adjoint_slot_stages = E::NDOF <= 16;  // the slot-per-lane adjoint kernel holds assembled columns only for small elements
  ks.can_stage = E::DIM == 3;              // the stage and the row-sum kernel are laid out for 3 + 1 equations per node
  return ks;
}

// The registry: one line per constitutive model (local_residual.cpp:893-933).  The instantiations are compiled in
// PARTS -- this file is compiled once per part with -DC8_KERNEL_PART=n (calibr8_amd/build.py, in parallel) and once as a
// whole by the tools that read its assembly; every part is a function the registry of part 0 dispatches to.
#ifndef C8_KERNEL_PART
#define C8_KERNEL_PART -1  // all parts in this translation unit
#endif
#define C8_PART(n) (C8_KERNEL_PART == -1 || C8_KERNEL_PART == (n))
KernelSet kernels_hex8_a(int model);       // part 0: elastic, small_J2, isotropic_elastic (+ get_kernels)
KernelSet kernels_hex8_b(int model);       // part 1: hyper_J2, small_hill
KernelSet kernels_hex8_c(int model);       // part 2: hypo_hill
KernelSet kernels_tet4(int model);         // part 3: the six models on tet4
KernelSet kernels_2d(int model);           // part 4: tri3 under mechanics and mechanics_plane_stress
KernelSet kernels_line_search(int elem_type, int model);  // part 5: Hosford / Barlat on tet4 and hex8

#if C8_PART(0)
KernelSet kernels_hex8_a(int model) {
  using E = Elem<C8_HEX8>;
  switch (model) {
    case MODEL_ELASTIC: return kernel_set<E, Elastic>();
    case MODEL_SMALL_J2: return kernel_set<E, SmallJ2>();
    case MODEL_ISOTROPIC_ELASTIC: return kernel_set<E, IsotropicElastic>();
  }
  return KernelSet{};
}
#endif
#if C8_PART(1)
KernelSet kernels_hex8_b(int model) {
  using E = Elem<C8_HEX8>;
  switch (model) {
    case MODEL_HYPER_J2: return kernel_set<E, HyperJ2>();
    case MODEL_SMALL_HILL: return kernel_set<E, SmallHill>();
  }
  return KernelSet{};
}
#endif
#if C8_PART(2)
KernelSet kernels_hex8_c(int model) {
  if (model == MODEL_HYPO_HILL) return kernel_set<Elem<C8_HEX8>, HypoHill>();
  return KernelSet{};
}
#endif
#if C8_PART(3)
KernelSet kernels_tet4(int model) {
  using E = Elem<C8_TET4>;
  switch (model) {
    case MODEL_ELASTIC: return kernel_set<E, Elastic>();
    case MODEL_SMALL_J2: return kernel_set<E, SmallJ2>();
    case MODEL_HYPER_J2: return kernel_set<E, HyperJ2>();
    case MODEL_SMALL_HILL: return kernel_set<E, SmallHill>();
    case MODEL_ISOTROPIC_ELASTIC: return kernel_set<E, IsotropicElastic>();
    case MODEL_HYPO_HILL: return kernel_set<E, HypoHill>();
  }
  return KernelSet{};
}
#endif
#if C8_PART(4)
// 2-D meshes: the models the reference's 2-D decks run on `mechanics` with 2 + 1 equations per node ...
KernelSet kernels_2d(int model) {
  switch (model) {
    case MODEL_SMALL_J2: return kernel_set<Elem<C8_TRI3>, SmallJ2Plane>();
    case MODEL_SMALL_HILL_PLANE_STRAIN: return kernel_set<Elem<C8_TRI3>, SmallHillPlaneStrain>();
    case MODEL_HYPER_J2_PLANE_STRAIN: return kernel_set<Elem<C8_TRI3>, HyperJ2PlaneStrain>();
    case MODEL_HYPO_HILL_PLANE_STRAIN: return kernel_set<Elem<C8_TRI3>, HypoHillPlaneStrain>();
    // ... and `mechanics_plane_stress`: one residual, six element DOFs
    case MODEL_SMALL_HILL_PLANE_STRESS: return kernel_set<Tri3PlaneStress, SmallHillPlaneStress>();
    case MODEL_HYPER_J2_PLANE_STRESS: return kernel_set<Tri3PlaneStress, HyperJ2PlaneStress>();
    case MODEL_HYPO_HILL_PLANE_STRESS: return kernel_set<Tri3PlaneStress, HypoHillPlaneStress>();
  }
  return KernelSet{};
}
#endif
#if C8_PART(5)
template <class E> static KernelSet line_search_set(int model) {
  switch (model) {
    case MODEL_SMALL_HOSFORD: return kernel_set<E, SmallHosford>();
    case MODEL_HYPO_HOSFORD: return kernel_set<E, HypoHosford>();
    case MODEL_HYPO_BARLAT: return kernel_set<E, HypoBarlat>();
  }
  return KernelSet{};
}
KernelSet kernels_line_search(int elem_type, int model) {
  return elem_type == C8_HEX8 ? line_search_set<Elem<C8_HEX8>>(model) : line_search_set<Elem<C8_TET4>>(model);
}
#endif

#if C8_PART(0)
KernelSet get_kernels(int elem_type, int model) {
  if (elem_type == C8_TRI3) return kernels_2d(model);
  if (model >= MODEL_SMALL_HILL_PLANE_STRAIN && model <= MODEL_HYPO_HILL_PLANE_STRESS) return KernelSet{};  // the plane models exist on 2-D meshes only
  if (elem_type != C8_HEX8 && elem_type != C8_TET4) return KernelSet{};
  if (model >= MODEL_SMALL_HOSFORD) return kernels_line_search(elem_type, model);
  if (elem_type == C8_TET4) return kernels_tet4(model);
  KernelSet ks = kernels_hex8_a(model);
  if (!ks.forward_jacobian) ks = kernels_hex8_b(model);
  if (!ks.forward_jacobian) ks = kernels_hex8_c(model);
  return ks;
}
#endif

}  // namespace c8
