// tests/emul/c8_emul.cpp -- TEST INFRASTRUCTURE.
//
// Runs the product's kernel source (calibr8_amd/csrc/c8_assemble*.hpp) on the CPU by
// instantiating it with a serial executor: each() loops the lanes of one lane group,
// shared (LDS) state is a stack object, atomics are plain adds.  This lets the CPU test
// suite check the kernel algorithms against the oracle without a GPU.  It is not a
// fallback: nothing under calibr8_amd/ can reach this file.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../calibr8_amd/csrc/c8_assemble_adjoint.hpp"
#include "../../calibr8_amd/csrc/c8_assemble_wave.hpp"
#include "../../calibr8_amd/csrc/c8_assemble_node.hpp"
#include "../../calibr8_amd/csrc/c8_host.hpp"
#include "../../calibr8_amd/csrc/c8_qoi_host.hpp"

using namespace c8;

template <class Lane, int NDOF> struct CpuExec {
  Lane lanes[NDOF];
  // inside each(), a lane may only touch its own registers: the GPU executor's lane(k) ignores k
  int cur = -1;
  template <class F> void each(F f) { for (int k = 0; k < NDOF; ++k) { cur = k; f(k); } cur = -1; }
  Lane& lane(int k) {
    if (cur >= 0 && k != cur) { std::fprintf(stderr, "c8emu: lane %d touched the registers of lane %d\n", cur, k); std::abort(); }
    return lanes[k];
  }
  template <class F> bool any(F f) { bool a = false; for (int k = 0; k < NDOF; ++k) { cur = k; a = f(k) || a; } cur = -1; return a; }
  template <class F> bool any_wave(F f) { return any(f); }
  bool uniform_any(bool x) { return x; }
  template <class F> int first_lane(F f) { cur = 0; int const v = f(0); cur = -1; return v; }
  void sync() {}
  void add(double* p, double v, int) { *p += v; }
  void flag(int* s) { *s = 1; }
  void lds_add(double* p, double v) { *p += v; }
  // the value of get() in lane S of the caller's group of 8 lanes (GpuExec: DPP broadcast).  Sound in this serial
  // emulation only where lane S does not change the value during the same each(): true for the pivot column of an
  // elimination step, which its owner leaves alone in that step.
  template <int S, class F> double bcast8(int lane, F get) {
    int const saved = cur;
    int const src = (lane & ~7) + S;
    cur = src;
    double const v = get(src);
    cur = saved;
    return v;
  }
  // get() of lane ^ 32 (GpuExec: a shuffle).  Sound here where no lane changes during an each() what its partner reads.
  // lanes 0..31: getA(lane) + getA(lane + 32); lanes 32..63: getB(lane - 32) + getB(lane) (GpuExec: v_permlane32_swap).
  // Sound where no lane overwrites during the each() what a LATER lane reads: the caller stores the result in the A
  // register, which only lanes below 32 read from their partners, and those have run by then.
  template <class FA, class FB> double pair_sum32(int lane, FA getA, FB getB) {
    int const saved = cur, other = lane ^ 32;
    double r;
    if (lane < 32) {
      double const own = getA(lane);
      cur = other;
      r = own + getA(other);
    } else {
      cur = other;
      double const o = getB(other);
      cur = saved;
      r = o + getB(lane);
    }
    cur = saved;
    return r;
  }
  template <class F> double xor32(int lane, F get) {
    int const saved = cur;
    cur = lane ^ 32;
    double const v = get(lane ^ 32);
    cur = saved;
    return v;
  }
};

enum { K_GRAD_CLOSED = 17, K_ADJ_LOCAL_CLOSED = 16, K_ADJ_JAC_NODE = 15, K_FORWARD_NODE = 14, K_QOI_PREPROCESS = 13, K_QOI_WAVE = 12, K_RESIDUAL_WAVE = 11, K_ADJ_LOCAL_WAVE = 9, K_GRAD_WAVE = 10, K_ADJ_JAC_WAVE = 8, K_FORWARD_WAVE = 7, K_FORWARD = 1, K_RESIDUAL = 2, K_ADJ_JAC = 3, K_ADJ_LOCAL = 4, K_GRAD = 5, K_QOI = 6 };

// objective configuration for the next calls (what c8_set_qoi_calibration / c8_set_measured keep in the context)
struct EmuQoi {
  int kind = 0;  // 0 average displacement, 1 calibration
  std::vector<int32_t> side_faces;
  int nfaces = 0, coord_idx = 0, comp = 0;
  double coord_value = 0., coord_tol = 0., w[3] = {1., 1., 1.}, balance = 0., dt_over_T = 1.;
  std::vector<double> u_meas;
  double load_meas = 0., load_mismatch = 0., total_load = 0., area = 0.;
};
static EmuQoi g_qoi;
extern "C" void c8emu_set_qoi_avg_disp() { g_qoi = EmuQoi(); }
extern "C" void c8emu_set_qoi_calibration(int nfaces, int npf, int const* faces, double const* weights, double balance,
                                          int coord_idx, double coord_value, double coord_tol, int comp, double dt_over_T) {
  g_qoi = EmuQoi();
  g_qoi.kind = 1;
  g_qoi.nfaces = nfaces;
  g_qoi.side_faces.assign(faces, faces + (size_t)nfaces * npf);
  for (int d = 0; d < 3; ++d) g_qoi.w[d] = weights[d];
  g_qoi.balance = balance; g_qoi.coord_idx = coord_idx; g_qoi.coord_value = coord_value; g_qoi.coord_tol = coord_tol;
  g_qoi.comp = comp; g_qoi.dt_over_T = dt_over_T;
}
extern "C" void c8emu_set_measured(int n, double const* u_meas, double load_meas) {
  g_qoi.u_meas.assign(u_meas, u_meas + n);
  g_qoi.load_meas = load_meas;
}
extern "C" void c8emu_qoi_info(double* out) { out[0] = g_qoi.area; out[1] = g_qoi.total_load; out[2] = g_qoi.load_mismatch; }

static int g_shape_cache = 1;  // the wave kernels read cached shape tables (c8_set_shape_cache) or compute them per call
extern "C" void c8emu_set_shape_cache(int on) { g_shape_cache = on; }
static int g_node_many = 0;  // row-per-node kernel: the form with separate accumulator storage also where no node has more than eight elements
extern "C" void c8emu_set_node_many(int on) { g_node_many = on; }
static int g_last_nchunks = 0;
extern "C" int c8emu_last_nchunks() { return g_last_nchunks; }

struct Call {
  int what;
  int staged;  // wave Jacobian kernels: staged (gather) assembly instead of direct adds
  int assign;  // staged assembly: the row sums assign A and b instead of adding (c8_set_assign_mode)
  int closed;  // forward wave kernel: the model's closed form (what the library runs by default) instead of the AD / Newton form
  int nnodes;
  HostGraph const* graph;
  HostMesh const* mesh;
  int* nchunks_out;
  int nelems;
  MeshTables mt;
  ModelSettings ms;
  FieldArgs fa;
  AdjointArgs aa;
  SystemArgs sa;
};

// staged assembly in the order the chunk loop of c8_api.hip produces: chunk k into the ring, then the rows of the
// nodes that chunk k completes; a small minimum chunk so that test meshes go round the ring
template <class E, class Assemble> static void run_staged(Call const& c, Assemble assemble) {
  StagePlan pl;
  plan_staged_assembly(*c.mesh, *c.graph, 1, 1, pl);
  std::vector<double> stage((size_t)pl.ring * stage_stride<E>(), 0.);
  SystemArgs sa = c.sa;
  sa.stage = stage.data();
  sa.stage_ring = pl.ring;
  auto* gsh = new GatherShared<E, GATHER_MAX_DEGREE>();
  auto* gex = new CpuExec<GatherLane<E, GATHER_MAX_DEGREE>, 64>();
  GatherArgs ga{c.mt.nodeptr, c.mt.pos, c.graph->nodeelem_ptr.data(), c.graph->nodeelem.data(), stage.data(), pl.ring,
                pl.node_order.data(), {{sa.A[0][0], sa.A[0][1]}, {sa.A[1][0], sa.A[1][1]}}, {sa.b[0], sa.b[1]}, c.assign};
  for (int k = 0; k < pl.nchunks; ++k) {
    int const e1 = std::min(c.nelems, (k + 1) * pl.chunk);
    for (int e = k * pl.chunk; e < e1; ++e) assemble(sa, e);
    for (int q = pl.node_off[k]; q < pl.node_off[k + 1]; ++q) gather_node_rows<E, GATHER_MAX_DEGREE>(*gex, *gsh, ga, pl.node_order[q]);
  }
  if (c.nchunks_out) *c.nchunks_out = pl.nchunks;
  delete gex;
  delete gsh;
}

template <class E, template <class> class ModelT> static void run(Call const& c) {
  using SH = GroupShared<E, ModelT<Dual>::NLOC>;
  SH sh;
  if (c.what == K_FORWARD) {
    auto* ex = new CpuExec<ForwardLane<E, ModelT>, E::NDOF>();
    auto fwd = [&](SystemArgs const& sa, int e) {  // the library's launcher: the closed-form instantiation where the model has one
      if constexpr (has_closed_form<ModelT<Dual>>::value) {
        if (c.ms.closed_form_slot) { forward_jacobian_element<E, ModelT, true>(*ex, sh, c.mt, c.ms, c.fa, sa, e); return; }
      }
      forward_jacobian_element<E, ModelT>(*ex, sh, c.mt, c.ms, c.fa, sa, e);
    };
    if (c.staged) run_staged<E>(c, fwd);
    else for (int e = 0; e < c.nelems; ++e) fwd(c.sa, e);
    delete ex;
  } else if (c.what == K_RESIDUAL) {
    auto* ex = new CpuExec<ResidualLane<E, ModelT>, E::NDOF>();
    for (int e = 0; e < c.nelems; ++e) residual_element<E, ModelT>(*ex, sh, c.mt, c.ms, c.fa, c.sa, e);
    delete ex;
  } else if (c.what == K_ADJ_JAC) {
    auto* ex = new CpuExec<AdjointLane<E, ModelT>, E::NDOF>();
    if (c.staged && E::NDOF <= 16) run_staged<E>(c, [&](SystemArgs const& sa, int e) { adjoint_jacobian_element<E, ModelT, PointQoi>(*ex, sh, c.mt, c.ms, c.fa, c.aa, sa, e); });
    else for (int e = 0; e < c.nelems; ++e) adjoint_jacobian_element<E, ModelT, PointQoi>(*ex, sh, c.mt, c.ms, c.fa, c.aa, c.sa, e);
    delete ex;
  } else if (c.what == K_ADJ_LOCAL) {
    auto* ex = new CpuExec<AdjointLane<E, ModelT>, E::NDOF>();
    for (int e = 0; e < c.nelems; ++e) adjoint_local_element<E, ModelT>(*ex, sh, c.mt, c.ms, c.fa, c.aa, c.sa, e);
    delete ex;
  } else if (c.what == K_GRAD) {
    auto* ex = new CpuExec<GradLane<E, ModelT>, E::NDOF>();
    for (int k = 0; k < E::NDOF; ++k) { ex->lanes[k].slot = -1; ex->lanes[k].acc = 0.; }
    for (int e = 0; e < c.nelems; ++e) param_gradient_element<E, ModelT, PointQoi>(*ex, sh, c.mt, c.ms, c.fa, c.aa, e);
    param_gradient_flush(*ex, c.aa);
    delete ex;
  } else if (c.what == K_QOI) {
    auto* ex = new CpuExec<QoiLane<E, ModelT>, E::NDOF>();
    for (int k = 0; k < E::NDOF; ++k) ex->lanes[k].acc = 0.;
    for (int e = 0; e < c.nelems; ++e) qoi_element<E, ModelT, PointQoi>(*ex, sh, c.mt, c.fa, c.aa.qoi, e);
    qoi_flush<E>(*ex, c.aa.out);
    delete ex;
  }
}

template <template <class> class ModelT> static void run_wave_adjoint(Call const& c) {
  using E = Elem<C8_HEX8>;
  if (c.what == K_ADJ_LOCAL_WAVE) {
    auto* sh = new WaveSharedA<E, ModelT<Dual>::NLOC>();
    auto* ex = new CpuExec<WaveLaneA<ModelT>, 64>();
    for (int e = 0; e < c.nelems; ++e) adjoint_local_wave<E, ModelT>(*ex, *sh, c.mt, c.ms, c.fa, c.aa, c.sa, e);
    delete ex;
    delete sh;
    return;
  }
  // K5: one "wavefront" walks all groups of eight elements, as one wave of the grid-stride kernel does
  auto* sh = new GradWaveShared<E>();
  auto* ex = new CpuExec<GradWaveLane<ModelT>, 64>();
  for (int k = 0; k < 64; ++k) { ex->lanes[k].slot0 = -1; for (int a = 0; a < 8; ++a) ex->lanes[k].acc[a] = 0.; }
  for (int e0 = 0; e0 < c.nelems; e0 += 8)
    param_gradient_wave8<E, ModelT, PointQoi>(*ex, *sh, c.mt, c.ms, c.fa, c.aa, e0, std::min(8, c.nelems - e0));
  param_gradient_wave8_flush(*ex, sh->red, c.aa);
  delete ex;
  delete sh;
}

// K4 in the model's closed form, eight elements per "wavefront" (adjoint_local_closed_wave8)
template <template <class> class ModelT> static void run_adjoint_local_closed(Call const& c) {
  using E = Elem<C8_HEX8>;
  auto* sh = new GradWaveShared<E>();
  struct NoLane {};
  auto* ex = new CpuExec<NoLane, 64>();
  for (int e0 = 0; e0 < c.nelems; e0 += 8)
    adjoint_local_closed_wave8<E, ModelT>(*ex, *sh, c.mt, c.ms, c.fa, c.aa, e0, std::min(8, c.nelems - e0));
  delete ex;
  delete sh;
}

// K5 in the model's closed form (param_gradient_wave8<CLOSED>)
template <template <class> class ModelT> static void run_param_gradient_closed(Call const& c) {
  using E = Elem<C8_HEX8>;
  auto* sh = new GradWaveShared<E>();
  auto* ex = new CpuExec<GradWaveLane<ModelT>, 64>();
  for (int k = 0; k < 64; ++k) { ex->lanes[k].slot0 = -1; for (int a = 0; a < 8; ++a) ex->lanes[k].acc[a] = 0.; }
  for (int e0 = 0; e0 < c.nelems; e0 += 8)
    param_gradient_wave8<E, ModelT, PointQoi, true>(*ex, *sh, c.mt, c.ms, c.fa, c.aa, e0, std::min(8, c.nelems - e0));
  param_gradient_wave8_flush(*ex, sh->red, c.aa);
  delete ex;
  delete sh;
}

template <template <class> class ModelT> static void run_wave(Call const& c) {
  using E = Elem<C8_HEX8>;
  auto* sh = new WaveShared<E, ModelT<Dual>::NLOC>();
  // the closed-form kernel has its own shared layout, as on the device (c8_kernels.hip: k_forward_jacobian_wave_closed)
  auto* shc = new WaveShared<E, ModelT<Dual>::NLOC, false, ModelT<Dual>::FINITE_DEF, true>();
  auto* ex = new CpuExec<WaveLane<ModelT>, 64>();
  auto one = [&](SystemArgs const& sa, int e) {
    if (c.what == K_FORWARD_WAVE) {
      if constexpr (has_closed_form<ModelT<Dual>>::value) {
        if (c.closed) { forward_jacobian_wave_closed<E, ModelT>(*ex, *shc, c.mt, c.ms, c.fa, sa, e); return; }
      }
      forward_jacobian_wave<E, ModelT>(*ex, *sh, c.mt, c.ms, c.fa, sa, e);
    }
    else adjoint_jacobian_wave<E, ModelT, PointQoi>(*ex, *sh, c.mt, c.ms, c.fa, c.aa, sa, e);
  };
  if (c.staged) run_staged<E>(c, one);
  else for (int e = 0; e < c.nelems; ++e) one(c.sa, e);
  delete ex;
  delete sh;
  delete shc;
}

// K1, one wavefront per node (c8_assemble_node.hpp): every node that has elements, in ascending order -- or, with
// c.staged set, the upper half of the nodes first and the lower half afterwards, as the two-part form of the library does
// (c8_set_gather_early_nodes + c8_gather_finish)
template <template <class> class ModelT, bool MANY> static void run_node_rows_as(Call const& c) {
  using E = Elem<C8_HEX8>;
  auto* sh = new NodeShared<E, ModelT<Dual>, GATHER_MAX_DEGREE, MANY>();
  auto* ex = new CpuExec<NodeLane<GATHER_MAX_DEGREE>, 64>();
  GatherArgs ga{c.mt.nodeptr, c.mt.pos, c.graph->nodeelem_ptr.data(), c.graph->nodeelem.data(), nullptr, 0, nullptr,
                {{c.sa.A[0][0], c.sa.A[0][1]}, {c.sa.A[1][0], c.sa.A[1][1]}}, {c.sa.b[0], c.sa.b[1]}, c.assign};
  auto one = [&](int n) {
    if (c.what == K_ADJ_JAC_NODE) node_rows_closed<E, ModelT, GATHER_MAX_DEGREE, MANY, true>(*ex, *sh, c.mt, c.ms, c.fa, ga, n, c.aa);
    else node_rows_closed<E, ModelT, GATHER_MAX_DEGREE, MANY>(*ex, *sh, c.mt, c.ms, c.fa, ga, n);
  };
  int const half = c.staged ? c.nnodes / 2 : 0;
  for (int n = half; n < c.nnodes; ++n) one(n);
  for (int n = 0; n < half; ++n) one(n);
  if (c.what == K_ADJ_JAC_NODE && c.aa.qoi.c_load != 0.)  // the library's closing launch: g -= dJ/dxi (launch_node_rows, first = -1)
    for (size_t qp = 0; qp < (size_t)c.nelems * E::NP0; ++qp) node_rows_update_g<E, ModelT>(c.mt, c.aa, qp);
  delete ex;
  delete sh;
}
template <template <class> class ModelT> static int run_node_rows(Call const& c) {
  if constexpr (has_closed_form_rows<ModelT<Dual>>::value) {
    if (!c.mt.shape || c.graph->max_degree > GATHER_MAX_DEGREE) return -4;
    // the library's choice (launch_node_rows); g_node_many forces the form for nodes with more than eight elements
    if (c.graph->max_node_elems > 8 || g_node_many) run_node_rows_as<ModelT, true>(c);
    else run_node_rows_as<ModelT, false>(c);
    return 0;
  } else {
    return -4;
  }
}

template <template <class> class ModelT> static void run_residual_wave(Call const& c) {
  using E = Elem<C8_HEX8>;
  auto* sh = new ResidualWaveShared<E>();
  auto* ex = new CpuExec<ResidualWaveLane<ModelT>, 64>();
  for (int e0 = 0; e0 < c.nelems; e0 += 8) residual_wave8<E, ModelT>(*ex, *sh, c.mt, c.ms, c.fa, c.sa, e0, std::min(8, c.nelems - e0));
  delete ex;
  delete sh;
}

template <template <class> class ModelT> static void run_qoi_wave(Call const& c) {
  using E = Elem<C8_HEX8>;
  auto* sh = new GradWaveShared<E>();
  auto* ex = new CpuExec<QoiWaveLane<ModelT>, 64>();
  for (int k = 0; k < 64; ++k) ex->lanes[k].acc = 0.;
  for (int e0 = 0; e0 < c.nelems; e0 += 8) qoi_wave8<E, ModelT, PointQoi>(*ex, *sh, c.mt, c.fa, c.aa.qoi, e0, std::min(8, c.nelems - e0));
  qoi_wave8_flush(*ex, sh->red, c.aa.out);
  delete ex;
  delete sh;
}

// 2-D meshes (tri3): the slot kernels with the 2-D models
static int dispatch_2d(std::string const& model, Call const& c) {
  using E = Elem<C8_TRI3>;
  if (c.what != K_FORWARD && c.what != K_RESIDUAL && c.what != K_ADJ_JAC && c.what != K_ADJ_LOCAL && c.what != K_GRAD && c.what != K_QOI) return -4;
  if (c.staged) return -4;
  if (model == "small_J2") run<E, SmallJ2Plane>(c);
  else if (model == "small_hill_plane_strain") run<E, SmallHillPlaneStrain>(c);
  else if (model == "hyper_J2_plane_strain") run<E, HyperJ2PlaneStrain>(c);
  else if (model == "hypo_hill_plane_strain") run<E, HypoHillPlaneStrain>(c);
  // `mechanics_plane_stress`: one residual, six element DOFs
  else if (model == "small_hill_plane_stress") run<Tri3PlaneStress, SmallHillPlaneStress>(c);
  else if (model == "hyper_J2_plane_stress") run<Tri3PlaneStress, HyperJ2PlaneStress>(c);
  else if (model == "hypo_hill_plane_stress") run<Tri3PlaneStress, HypoHillPlaneStress>(c);
  else return -2;
  return 0;
}

template <class E> static int dispatch(std::string const& model, Call const& c) {
  if (c.what == K_FORWARD_NODE || c.what == K_ADJ_JAC_NODE) {
    if (E::TYPE != C8_HEX8) return -4;
    if (model == "small_J2") return run_node_rows<SmallJ2>(c);
    return -4;
  }
  if (c.what == K_ADJ_LOCAL_CLOSED || c.what == K_GRAD_CLOSED) {
    if (E::TYPE != C8_HEX8) return -4;
    if constexpr (E::TYPE == C8_HEX8) {
      if (model == "small_J2") {
        if (c.what == K_GRAD_CLOSED) run_param_gradient_closed<SmallJ2>(c);
        else run_adjoint_local_closed<SmallJ2>(c);
        return 0;
      }
    }
    return -4;
  }
  if (c.what == K_QOI_WAVE) {
    if (E::TYPE != C8_HEX8) return -4;
    if (model == "elastic") run_qoi_wave<Elastic>(c);
    else if (model == "small_J2") run_qoi_wave<SmallJ2>(c);
    else if (model == "hyper_J2") run_qoi_wave<HyperJ2>(c);
    else if (model == "small_hill") run_qoi_wave<SmallHill>(c);
    else if (model == "isotropic_elastic") run_qoi_wave<IsotropicElastic>(c);
    else if (model == "hypo_hill") run_qoi_wave<HypoHill>(c);
    else if (model == "small_hosford") run_qoi_wave<SmallHosford>(c);
    else if (model == "hypo_hosford") run_qoi_wave<HypoHosford>(c);
    else if (model == "hypo_barlat") run_qoi_wave<HypoBarlat>(c);
    else return -2;
    return 0;
  }
  if (c.what == K_RESIDUAL_WAVE) {
    if (E::TYPE != C8_HEX8) return -4;
    if (model == "elastic") run_residual_wave<Elastic>(c);
    else if (model == "small_J2") run_residual_wave<SmallJ2>(c);
    else if (model == "hyper_J2") run_residual_wave<HyperJ2>(c);
    else if (model == "small_hill") run_residual_wave<SmallHill>(c);
    else if (model == "isotropic_elastic") run_residual_wave<IsotropicElastic>(c);
    else if (model == "hypo_hill") run_residual_wave<HypoHill>(c);
    else if (model == "small_hosford") run_residual_wave<SmallHosford>(c);
    else if (model == "hypo_hosford") run_residual_wave<HypoHosford>(c);
    else if (model == "hypo_barlat") run_residual_wave<HypoBarlat>(c);
    else return -2;
    return 0;
  }
  if (c.what == K_ADJ_LOCAL_WAVE || c.what == K_GRAD_WAVE) {
    if (E::TYPE != C8_HEX8) return -4;
    if (model == "elastic") run_wave_adjoint<Elastic>(c);
    else if (model == "small_J2") run_wave_adjoint<SmallJ2>(c);
    else if (model == "hyper_J2") run_wave_adjoint<HyperJ2>(c);
    else if (model == "small_hill") run_wave_adjoint<SmallHill>(c);
    else if (model == "isotropic_elastic") run_wave_adjoint<IsotropicElastic>(c);
    else if (model == "hypo_hill") run_wave_adjoint<HypoHill>(c);
    else if (model == "small_hosford") run_wave_adjoint<SmallHosford>(c);
    else if (model == "hypo_hosford") run_wave_adjoint<HypoHosford>(c);
    else if (model == "hypo_barlat") run_wave_adjoint<HypoBarlat>(c);
    else return -2;
    return 0;
  }
  if (c.what == K_FORWARD_WAVE || c.what == K_ADJ_JAC_WAVE) {
    if (E::TYPE != C8_HEX8) return -4;
    if (model == "elastic") run_wave<Elastic>(c);
    else if (model == "small_J2") run_wave<SmallJ2>(c);
    else if (model == "hyper_J2") run_wave<HyperJ2>(c);
    else if (model == "small_hill") run_wave<SmallHill>(c);
    else if (model == "isotropic_elastic") run_wave<IsotropicElastic>(c);
    else if (model == "hypo_hill") run_wave<HypoHill>(c);
    else if (model == "small_hosford") run_wave<SmallHosford>(c);
    else if (model == "hypo_hosford") run_wave<HypoHosford>(c);
    else if (model == "hypo_barlat") run_wave<HypoBarlat>(c);
    else return -2;
    return 0;
  }
  if (model == "elastic") run<E, Elastic>(c);
  else if (model == "small_J2") run<E, SmallJ2>(c);
  else if (model == "hyper_J2") run<E, HyperJ2>(c);
  else if (model == "small_hill") run<E, SmallHill>(c);
  else if (model == "isotropic_elastic") run<E, IsotropicElastic>(c);
  else if (model == "hypo_hill") run<E, HypoHill>(c);
  else if (model == "small_hosford") run<E, SmallHosford>(c);
  else if (model == "hypo_hosford") run<E, HypoHosford>(c);
  else if (model == "hypo_barlat") run<E, HypoBarlat>(c);
  else return -2;
  return 0;
}

// the `line search:` settings of the local residual (c8_model_desc.ls_*): one configuration, like the objective
static double g_ls_c1 = 1.e-4, g_ls_bmin = 0.5, g_ls_bmax = 0.9;
static int g_ls_max_evals = 4;
extern "C" void c8emu_set_local_line_search(double c1, double bmin, double bmax, int max_evals) {
  g_ls_c1 = c1; g_ls_bmin = bmin; g_ls_bmax = bmax; g_ls_max_evals = max_evals;
}

// ptrs: 0 u, 1 p, 2 u_prev, 3 p_prev, 4 xi_prev, 5 xi, 6..9 A00 A01 A10 A11, 10 b0, 11 b1,
//       12 g, 13 f, 14 z_u, 15 z_p, 16 phi, 17 out
extern "C" int c8emu_call(int what, int elem_type, int nnodes, int nelems, double const* coords, int const* conn,
                          int const* elem_set, int nsets, char const* local_type, double stab_mult, int max_iters,
                          double abs_tol, double rel_tol, double const* params, int const* active, double** ptrs) {
  HostMesh mesh;
  HostGraph graph;
  mesh.elem_type = elem_type;
  mesh.nn = elem_type;
  mesh.nnodes = nnodes;
  mesh.nelems = nelems;
  mesh.nsets = nsets;
  mesh.coords.assign(coords, coords + (size_t)nnodes * 3);
  mesh.conn.assign(conn, conn + (size_t)nelems * elem_type);
  std::string const err = build_node_graph(mesh, graph);
  if (!err.empty()) { std::fprintf(stderr, "c8emu: %s\n", err.c_str()); return -3; }
  int status = 0;
  Call c;
  c.what = what & 0xff;
  c.staged = (what >> 8) & 1;
  c.assign = (what >> 9) & 1;
  c.closed = (what >> 10) & 1;
  c.nnodes = nnodes;
  c.graph = &graph;
  c.mesh = &mesh;
  c.nchunks_out = &g_last_nchunks;
  c.nelems = nelems;
  c.mt = MeshTables{mesh.conn.data(), mesh.coords.data(), graph.nodeptr.data(), graph.pos.data(), elem_set, nullptr, params};
  std::vector<double> shape_tab;
  if (g_shape_cache && elem_type == C8_HEX8) {  // what c8_set_shape_cache builds at c8_create, same source
    using E8 = Elem<C8_HEX8>;
    shape_tab.assign((size_t)nelems * SHAPE_STRIDE, 0.);
    auto* ssh = new ShapeShared<E8>();
    auto* sex = new CpuExec<ShapeLane, 64>();
    for (int e = 0; e < nelems; ++e) store_shape_tables<E8>(*sex, *ssh, c.mt, shape_tab.data(), e);
    delete sex;
    delete ssh;
    c.mt.shape = shape_tab.data();
  }
  c.ms = ModelSettings{stab_mult, abs_tol, rel_tol, max_iters};
  c.ms.closed_form_slot = c.closed;  // the lane-group kernel's closed-form path (the library: C8_KERNEL_AUTO)
  c.ms.ls_c1 = g_ls_c1; c.ms.ls_bmin = g_ls_bmin; c.ms.ls_bmax = g_ls_bmax; c.ms.ls_max_evals = g_ls_max_evals;
  c.fa = FieldArgs{ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], ptrs[5]};
  c.sa = SystemArgs{{{ptrs[6], ptrs[7]}, {ptrs[8], ptrs[9]}}, {ptrs[10], ptrs[11]}, &status, 0};
  c.aa = AdjointArgs{ptrs[12], ptrs[13], ptrs[14], ptrs[15], ptrs[16], ptrs[17], active, QoiArgs{1., 0., 0, nullptr, elem_type == C8_TRI3 ? 2. : 3.}};
  auto run_it = [&]() {
    if (elem_type == C8_TRI3) return dispatch_2d(local_type, c);
    return (elem_type == C8_HEX8) ? dispatch<Elem<C8_HEX8>>(local_type, c) : dispatch<Elem<C8_TET4>>(local_type, c);
  };
  int const base = c.what;
  bool const is_qoi = base == K_QOI || base == K_QOI_WAVE, is_k3 = base == K_ADJ_JAC || base == K_ADJ_JAC_WAVE || base == K_ADJ_JAC_NODE,
             is_k5 = base == K_GRAD || base == K_GRAD_WAVE || base == K_GRAD_CLOSED, is_pre = base == K_QOI_PREPROCESS;
  if (g_qoi.kind == 1 && (is_qoi || is_k3 || is_k5 || is_pre)) {
    // the sequence of c8_qoi.hip: tables (set-up), preprocess_qoi, then the entry point with the point integrand
    CalibrationTables t;
    calibration_tables(mesh, g_qoi.nfaces, g_qoi.side_faces.data(), g_qoi.coord_idx, g_qoi.coord_value, g_qoi.coord_tol, t);
    g_qoi.area = t.area;
    int const npts0 = (elem_type == C8_HEX8) ? Elem<C8_HEX8>::NP0 : (elem_type == C8_TET4 ? Elem<C8_TET4>::NP0 : Elem<C8_TRI3>::NP0);
    int const nd = elem_type == C8_TRI3 ? 2 : 3;
    double total = 0.;
    {
      Call pc = c;
      pc.what = (elem_type == C8_HEX8 && (base == K_QOI_WAVE || base == K_ADJ_JAC_WAVE || base == K_GRAD_WAVE || base == K_GRAD_CLOSED)) ? K_QOI_WAVE : K_QOI;
      pc.staged = 0;
      pc.aa.out = &total;
      pc.aa.qoi = QoiArgs{0., 1., g_qoi.comp, t.S.data(), (double)nd};
      Call const saved = c;
      c = pc;
      int const rc0 = run_it();
      c = saved;
      if (rc0 != 0) return rc0;
    }
    g_qoi.total_load = total;
    g_qoi.load_mismatch = total - g_qoi.load_meas;
    if (is_pre) return 0;
    double const scale = (double)npts0 * g_qoi.dt_over_T / t.area;
    auto surface = [&](double* J, double* b0) {
      for (size_t f = 0; f < t.faces.size() / 4; ++f) {
        int32_t const* fn = &t.faces[f * 4];
        double grad[4][3];
        double const val = surface_mismatch_face(t.nfn, fn, mesh.coords.data(), ptrs[0], g_qoi.u_meas.data(), g_qoi.w, grad, nd);
        if (J) *J += val * scale;
        if (b0)
          for (int k = 0; k < t.nfn; ++k)
            for (int d = 0; d < nd; ++d) b0[(size_t)fn[k] * nd + d] -= grad[k][d] * scale;
      }
    };
    if (is_qoi) {  // Calibration<double>::evaluate + postprocess
      surface(ptrs[17], nullptr);
      *ptrs[17] += 0.5 * g_qoi.balance * g_qoi.dt_over_T * g_qoi.load_mismatch * g_qoi.load_mismatch;
      return 0;
    }
    c.aa.qoi = QoiArgs{0., g_qoi.balance * g_qoi.dt_over_T * g_qoi.load_mismatch, g_qoi.comp, t.S.data(), (double)nd};
    int const rc = run_it();
    if (rc != 0) return rc;
    if (is_k3) surface(nullptr, ptrs[10]);
    return status ? -1 : 0;
  }
  if (is_pre) return 0;
  int const rc = run_it();
  if (rc != 0) return rc;
  return status ? -1 : 0;
}

// G5: the dual-number intrinsics and 3x3 tensor helpers of c8_math.hpp, one operation at a time:
// out = {value, derivative} of op(a, b) with da, db the tangents of the arguments
extern "C" int c8emu_dual_op(int op, double a, double da, double b, double db, double* out) {
  Dual const A(a, da), B(b, db);
  Dual r(0.);
  switch (op) {
    case 0: r = A + B; break;
    case 1: r = A - B; break;
    case 2: r = A * B; break;
    case 3: r = A / B; break;
    case 4: r = c8_sqrt(A); break;
    case 5: r = c8_cbrt(A); break;
    case 6: r = c8_exp(A); break;
    case 7: r = c8_pow(A, B); break;
    case 8: r = b / A; break;         // double / Dual
    case 9: r = A / b; break;         // Dual / double
    case 10: {                          // det and norm of a 3x3 tensor whose entries all move with the same tangent
      Tens3<Dual> t;
      t.xx = A; t.xy = A * 0.5; t.xz = B; t.yx = B * 2.; t.yy = A + 1.; t.yz = A - B; t.zx = B; t.zy = A * B; t.zz = A + 2.;
      r = det(t) + norm(t);
      break;
    }
    case 11: {                          // trace of inverse
      Tens3<Dual> t;
      t.xx = A + 3.; t.xy = B; t.xz = A * 0.1; t.yx = B * 0.2; t.yy = A + 4.; t.yz = B; t.zx = A * 0.3; t.zy = B * 0.1; t.zz = A + 5.;
      r = trace(inverse(t));
      break;
    }
    default: return -1;
  }
  out[0] = r.v;
  out[1] = r.d;
  return 0;
}

