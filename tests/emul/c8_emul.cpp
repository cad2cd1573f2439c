// tests/emul/c8_emul.cpp -- TEST INFRASTRUCTURE.
//
// Runs the product's kernel source (calibr8_amd/csrc/c8_assemble.hpp) on the CPU by
// instantiating it with a serial executor: each() loops the lanes of one lane group,
// shared (LDS) state is a stack object, atomics are plain adds.  This lets the CPU test
// suite check the kernel algorithm against the oracle without a GPU.  It is not a
// fallback: nothing under calibr8_amd/ can reach this file.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../calibr8_amd/csrc/c8_assemble.hpp"
#include "../../calibr8_amd/csrc/c8_host.hpp"

using namespace c8;

template <class Lane, int NDOF> struct CpuExec {
  Lane lanes[NDOF];
  template <class F> void each(F f) { for (int k = 0; k < NDOF; ++k) f(k); }
  Lane& lane(int k) { return lanes[k]; }
  template <class F> bool any(F f) { return f(0); }
  void sync() {}
  void add(double* p, double v, int) { *p += v; }
  void flag(int* s) { *s = 1; }
};

struct Problem {
  HostMesh mesh;
  HostGraph graph;
  MeshTables mt;
  ModelSettings ms;
};

template <class E, template <class> class ModelT>
static void run_forward(Problem& pb, FieldArgs const& fa, SystemArgs const& sa) {
  using Lane = ForwardLane<E, ModelT>;
  auto* ex = new CpuExec<Lane, E::NDOF>();
  GroupShared<E, ModelT<Dual>::NLOC> sh;
  for (int e = 0; e < pb.mesh.nelems; ++e) forward_jacobian_element<E, ModelT>(*ex, sh, pb.mt, pb.ms, fa, sa, e);
  delete ex;
}

template <class E> static int dispatch_forward(std::string const& model, Problem& pb, FieldArgs const& fa, SystemArgs const& sa) {
  if (model == "elastic") run_forward<E, Elastic>(pb, fa, sa);
  else if (model == "small_J2") run_forward<E, SmallJ2>(pb, fa, sa);
  else if (model == "hyper_J2") run_forward<E, HyperJ2>(pb, fa, sa);
  else return -2;
  return 0;
}

extern "C" int c8emu_forward_jacobian(int elem_type, int nnodes, int nelems, double const* coords, int const* conn,
                                      int const* elem_set, int nsets, char const* local_type, double stab_mult,
                                      int max_iters, double abs_tol, double rel_tol, double const* params,
                                      double const* u, double const* p, double const* u_prev, double const* p_prev,
                                      double const* xi_prev, double* xi, double* A00, double* A01, double* A10,
                                      double* A11, double* b0, double* b1) {
  Problem pb;
  pb.mesh.elem_type = elem_type;
  pb.mesh.nn = elem_type;
  pb.mesh.nnodes = nnodes;
  pb.mesh.nelems = nelems;
  pb.mesh.nsets = nsets;
  pb.mesh.coords.assign(coords, coords + (size_t)nnodes * 3);
  pb.mesh.conn.assign(conn, conn + (size_t)nelems * elem_type);
  std::string const err = build_node_graph(pb.mesh, pb.graph);
  if (!err.empty()) { std::fprintf(stderr, "c8emu: %s\n", err.c_str()); return -3; }
  pb.mt = MeshTables{pb.mesh.conn.data(), pb.mesh.coords.data(), pb.graph.nodeptr.data(), pb.graph.pos.data(),
                     elem_set, nullptr, params};
  pb.ms = ModelSettings{stab_mult, abs_tol, rel_tol, max_iters};
  FieldArgs fa{u, p, u_prev, p_prev, xi_prev, xi};
  int status = 0;
  SystemArgs sa{{{A00, A01}, {A10, A11}}, {b0, b1}, &status, 0};
  int rc = (elem_type == C8_HEX8) ? dispatch_forward<Elem<C8_HEX8>>(local_type, pb, fa, sa)
                                  : dispatch_forward<Elem<C8_TET4>>(local_type, pb, fa, sa);
  if (rc != 0) return rc;
  return status ? -1 : 0;
}
