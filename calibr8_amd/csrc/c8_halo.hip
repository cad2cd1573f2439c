// c8_halo.hip -- multi-part meshes: the owned/ghost halo exchanges and the small reductions
// of the path, behind the C ABI.
//
//   C1/C2  ghost rows of b and of the four CSR blocks of A  -> owner, ADD   (LinearAlg::gather_b / gather_A,
//          linear_alg.cpp:53-86: Tpetra Export with the exporters of disc.cpp:316-332)
//   C3     owner values of a nodal field -> ghost and phantom copies, COPY  (apf::synchronize, disc.cpp:944-947)
//   C4/C5  SUM all-reduce of a few doubles                                  (PCU_Add_*, primal.cpp:100)
//
// Design.  All index work is done ONCE on the host (c8_halo_build): per exchange a list of value positions to pack
// (send order = rank-major, rows whole, in the sender's graph order) and, for the receiver, for every destination value
// the list of message positions that add into it, in ascending source-rank order.  At run time an exchange is
//   pack kernel (context stream)  ->  grouped ncclSend/ncclRecv, one message per neighbour (comm stream)  ->
//   unpack kernel (context stream): one thread per destination value sums its contributions in that fixed order.
// No atomics anywhere: the gathered system is bitwise reproducible whatever the arrival order.  Events order the
// two streams; the host never waits (RCCL transport).  xGMI is point-to-point: a 2x2x2 block partition has 7
// neighbours per part = 7 links, so every message has a link of its own.
//
// RCCL is loaded with dlopen at the first use, so the library itself has no link-time dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/c8.h"
#include "c8_api_internal.hpp"

namespace {

constexpr int TPB = 256;
constexpr int NSEG = 6;                       // A00 A01 A10 A11 b0|x0 b1|x1
constexpr int64_t OFF_MASK = ((int64_t)1 << 56) - 1;
inline int64_t code(int seg, int64_t off) { return ((int64_t)seg << 56) | off; }

#define C8H_HIP(call)                                                                                             \
  do {                                                                                                            \
    hipError_t err__ = (call);                                                                                    \
    if (err__ != hipSuccess) return c8_fail(C8_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(err__)); \
  } while (0)

struct Segs { double* p[NSEG]; };

__global__ void k_pack(int64_t n, int64_t const* idx, Segs s, double* out) {
  int64_t const t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n) return;
  int64_t const c = idx[t];
  out[t] = s.p[c >> 56][c & OFF_MASK];
}
// one thread per destination value: its contributions summed in ascending source-rank order, then added
__global__ void k_unpack_add(int64_t n, int64_t const* dst, int64_t const* src_ptr, int64_t const* src, double const* in, Segs s) {
  int64_t const t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n) return;
  double v = 0.;
  for (int64_t k = src_ptr[t]; k < src_ptr[t + 1]; ++k) v += in[src[k]];
  int64_t const c = dst[t];
  s.p[c >> 56][c & OFF_MASK] += v;
}
__global__ void k_unpack_store(int64_t n, int64_t const* dst, double const* in, Segs s) {
  int64_t const t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n) return;
  int64_t const c = dst[t];
  s.p[c >> 56][c & OFF_MASK] = in[t];
}
inline int grid_of(int64_t n) { return (int)((n + TPB - 1) / TPB); }

// ---- RCCL through dlopen ---------------------------------------------------------------------------------------
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};
Rccl& rccl() {
  static Rccl r;
  if (r.lib || !r.error.empty()) return r;
  std::vector<std::string> names;
  if (char const* e = std::getenv("C8_RCCL_LIB")) names.push_back(e);
  names.push_back("librccl.so.1");
  names.push_back("librccl.so");
  names.push_back("/opt/rocm/lib/librccl.so.1");
  // a copy the process has loaded already comes first (under PyTorch that is the one built against the HIP runtime in
  // the process; loading a second RCCL beside it would give the process two)
  if (!std::getenv("C8_RCCL_LIB"))
    for (char const* n : {"librccl.so", "librccl.so.1"}) {
      r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
      if (r.lib) break;
    }
  if (!r.lib)
    for (auto const& n : names) {
      r.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
  if (!r.lib) { r.error = std::string("cannot load librccl (set C8_RCCL_LIB): ") + dlerror(); return r; }
  auto sym = [&](char const* name) -> void* {
    void* p = dlsym(r.lib, name);
    if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + name;
    return p;
  };
  r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
  r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
  r.Send = (decltype(r.Send))sym("ncclSend");
  r.Recv = (decltype(r.Recv))sym("ncclRecv");
  r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
  r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  if (!r.error.empty()) { dlclose(r.lib); r.lib = nullptr; }
  return r;
}
#define C8H_NCCL(call)                                                                                         \
  do {                                                                                                         \
    ncclResult_t res__ = (call);                                                                               \
    if (res__ != ncclSuccess) return c8_fail(C8_ERR_DEVICE, std::string(#call) + ": " + rccl().GetErrorString(res__)); \
  } while (0)

// one exchange pattern: what to pack, how much goes where, where what arrives is added / stored
struct Exchange {
  std::vector<int64_t> send_idx, send_counts, recv_counts;       // codes; per-rank counts
  std::vector<int64_t> dst, src_ptr, src;                        // unpack: destination codes (unique), contributions
  int64_t nsend = 0, nrecv = 0;
  int64_t *d_send_idx = nullptr, *d_dst = nullptr, *d_src_ptr = nullptr, *d_src = nullptr;
  bool add = true;                                               // ADD (C1/C2) or COPY (C3: dst has one source each, in message order)
};

}  // namespace

struct c8_comm {
  int rank = 0, nranks = 1;
  ncclComm_t nccl = nullptr;
  hipStream_t stream = nullptr;          // RCCL transport: the stream the messages travel on
  double* d_small = nullptr;             // device scratch of the small all-reduce
  c8_host_exchange_fn host_exchange = nullptr;
  c8_host_allreduce_fn host_allreduce = nullptr;
  void* user = nullptr;
};

struct c8_halo {
  int rank = 0, nranks = 1;
  int32_t nnodes = 0, nowned = 0, ntouched = 0;
  int ndims = 3, nres = 2;               // equations per node of residual 0; residuals in the systems
  std::vector<int64_t> nodeptr;          // of the graph the tables were built from (checked at attach)
  Exchange full, bonly, aonly, import;   // C1+C2, C1, C2, C3
  // attached state
  c8_ctx* ctx = nullptr;
  c8_comm* comm = nullptr;
  double *d_sendbuf = nullptr, *d_recvbuf = nullptr;   // sized for the largest exchange
  double *h_sendbuf = nullptr, *h_recvbuf = nullptr;   // pinned, host transport only
  hipEvent_t ev_packed = nullptr, ev_arrived = nullptr;
  Exchange* pending = nullptr;
};

namespace {

// receiver side of an ADD exchange: group the message positions by destination, sources in ascending position
// (= ascending source rank, the message is rank-major)
void group_by_destination(std::vector<int64_t> const& dst_of_pos, Exchange& x) {
  size_t const n = dst_of_pos.size();
  std::vector<int64_t> order(n);
  std::iota(order.begin(), order.end(), (int64_t)0);
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return dst_of_pos[a] < dst_of_pos[b]; });
  x.dst.clear();
  x.src_ptr.assign(1, 0);
  x.src.resize(n);
  for (size_t k = 0; k < n; ++k) {
    int64_t const d = dst_of_pos[order[k]];
    if (x.dst.empty() || x.dst.back() != d) {
      if (!x.dst.empty()) x.src_ptr.push_back((int64_t)k);
      x.dst.push_back(d);
    }
    x.src[k] = order[k];
  }
  x.src_ptr.push_back((int64_t)n);
  if (x.dst.empty()) x.src_ptr.assign(1, 0);
}

int upload64(int64_t** d, std::vector<int64_t> const& h) {
  *d = nullptr;
  if (h.empty()) return C8_OK;
  C8H_HIP(hipMalloc((void**)d, h.size() * sizeof(int64_t)));
  C8H_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  return C8_OK;
}
int upload_exchange(Exchange& x) {
  int rc;
  if ((rc = upload64(&x.d_send_idx, x.send_idx)) || (rc = upload64(&x.d_dst, x.dst))) return rc;
  if (x.add && ((rc = upload64(&x.d_src_ptr, x.src_ptr)) || (rc = upload64(&x.d_src, x.src)))) return rc;
  return C8_OK;
}
void free_exchange(Exchange& x) {
  (void)hipFree(x.d_send_idx); (void)hipFree(x.d_dst); (void)hipFree(x.d_src_ptr); (void)hipFree(x.d_src);
  x.d_send_idx = x.d_dst = x.d_src_ptr = x.d_src = nullptr;
}

// pack on the context's stream, move the message, leave it in d_recvbuf with ev_arrived recorded (RCCL) or copied (host)
int start_exchange(c8_halo* h, Exchange& x, Segs const& segs) {
  c8_ctx* c = h->ctx;
  c8_comm* cm = h->comm;
  if (h->pending) return c8_fail(C8_ERR_ARG, "halo: an exchange is already in flight");
  // a part without neighbours has nothing to move -- but the host transport's callback may be a collective over all
  // ranks (torch.distributed all_to_all), so it is entered with all-zero counts; RCCL point-to-point needs no call
  if (x.nsend == 0 && x.nrecv == 0 && cm->nccl) return C8_OK;
  if (x.nsend > 0) {
    hipLaunchKernelGGL(k_pack, dim3(grid_of(x.nsend)), dim3(TPB), 0, c->stream, x.nsend, x.d_send_idx, segs, h->d_sendbuf);
    C8H_HIP(hipGetLastError());
  }
  if (cm->nccl) {
    Rccl& R = rccl();
    C8H_HIP(hipEventRecord(h->ev_packed, c->stream));
    C8H_HIP(hipStreamWaitEvent(cm->stream, h->ev_packed, 0));
    C8H_NCCL(R.GroupStart());
    int64_t so = 0, ro = 0;
    for (int r = 0; r < h->nranks; ++r) {
      if (x.send_counts[r] > 0) C8H_NCCL(R.Send(h->d_sendbuf + so, (size_t)x.send_counts[r], ncclDouble, r, cm->nccl, cm->stream));
      if (x.recv_counts[r] > 0) C8H_NCCL(R.Recv(h->d_recvbuf + ro, (size_t)x.recv_counts[r], ncclDouble, r, cm->nccl, cm->stream));
      so += x.send_counts[r];
      ro += x.recv_counts[r];
    }
    C8H_NCCL(R.GroupEnd());
    C8H_HIP(hipEventRecord(h->ev_arrived, cm->stream));
  } else {
    if (x.nsend > 0) C8H_HIP(hipMemcpyAsync(h->h_sendbuf, h->d_sendbuf, (size_t)x.nsend * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    C8H_HIP(hipStreamSynchronize(c->stream));
    if (cm->host_exchange(cm->user, h->h_sendbuf, x.send_counts.data(), h->h_recvbuf, x.recv_counts.data()) != 0)
      return c8_fail(C8_ERR_ARG, "halo: the host exchange callback failed");
  }
  h->pending = &x;
  return C8_OK;
}
int finish_exchange(c8_halo* h, Segs const& segs) {
  c8_ctx* c = h->ctx;
  Exchange* x = h->pending;
  if (!x) return C8_OK;
  h->pending = nullptr;
  if (h->comm->nccl) {
    C8H_HIP(hipStreamWaitEvent(c->stream, h->ev_arrived, 0));
  } else if (x->nrecv > 0) {
    C8H_HIP(hipMemcpyAsync(h->d_recvbuf, h->h_recvbuf, (size_t)x->nrecv * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  int64_t const nd = (int64_t)x->dst.size();
  if (nd > 0) {
    if (x->add) hipLaunchKernelGGL(k_unpack_add, dim3(grid_of(nd)), dim3(TPB), 0, c->stream, nd, x->d_dst, x->d_src_ptr, x->d_src, h->d_recvbuf, segs);
    else hipLaunchKernelGGL(k_unpack_store, dim3(grid_of(nd)), dim3(TPB), 0, c->stream, nd, x->d_dst, h->d_recvbuf, segs);
    C8H_HIP(hipGetLastError());
  }
  // buffer reuse: the context's stream has waited for ev_arrived (sends and receives of this message are done before
  // the next pack overwrites d_sendbuf); the comm stream waits for the next ev_packed, which follows this unpack
  return C8_OK;
}

Segs system_segs(const c8_system* sys) {
  return Segs{{sys->A[0][0], sys->A[0][1], sys->A[1][0], sys->A[1][1], sys->b[0], sys->b[1]}};
}

}  // namespace

int c8_parts_allreduce(c8_ctx* c, double* values, int n) {
  if (c->allreduce) { c->allreduce(c->allreduce_user, values, n); return C8_OK; }
  if (c->halo && c->halo->comm) return c8_comm_allreduce_sum(c->halo->comm, values, n);
  return C8_OK;
}
int c8_halo_num_owned(c8_halo const* h) { return h->nowned; }
void c8_halo_detach_ctx(c8_ctx* c) {
  c8_halo* h = c->halo;
  if (!h) return;
  if (h->comm && h->comm->stream) (void)hipStreamSynchronize(h->comm->stream);
  (void)hipStreamSynchronize(c->stream);
  h->pending = nullptr;
  h->ctx = nullptr;
  h->comm = nullptr;
  c->halo = nullptr;
}

extern "C" {

// ---- communicator ----------------------------------------------------------------------------------------------
int c8_comm_rccl_id(void* id_out) {
  if (!id_out) return c8_fail(C8_ERR_ARG, "c8_comm_rccl_id: null argument");
  Rccl& R = rccl();
  if (!R.lib) return c8_fail(C8_ERR_UNSUPPORTED, "c8_comm_rccl_id: " + R.error);
  static_assert(sizeof(ncclUniqueId) == C8_COMM_ID_BYTES, "id size");
  ncclUniqueId id;
  C8H_NCCL(R.GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof(id));
  return C8_OK;
}

int c8_comm_create_rccl(const void* id_in, int rank, int nranks, c8_comm** out) {
  if (!id_in || !out || nranks < 1 || rank < 0 || rank >= nranks) return c8_fail(C8_ERR_ARG, "c8_comm_create_rccl: bad argument");
  *out = nullptr;
  Rccl& R = rccl();
  if (!R.lib) return c8_fail(C8_ERR_UNSUPPORTED, "c8_comm_create_rccl: " + R.error);
  ncclUniqueId id;
  std::memcpy(&id, id_in, sizeof(id));
  c8_comm* cm = new c8_comm();
  cm->rank = rank;
  cm->nranks = nranks;
  ncclResult_t const res = R.CommInitRank(&cm->nccl, nranks, id, rank);
  if (res != ncclSuccess) {
    delete cm;
    return c8_fail(C8_ERR_DEVICE, std::string("c8_comm_create_rccl: ncclCommInitRank: ") + R.GetErrorString(res));
  }
  if (hipStreamCreateWithFlags(&cm->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&cm->d_small, 64 * sizeof(double)) != hipSuccess) {
    c8_comm_destroy(cm);
    return c8_fail(C8_ERR_DEVICE, "c8_comm_create_rccl: stream / scratch allocation failed");
  }
  *out = cm;
  return C8_OK;
}

int c8_comm_create_host(int rank, int nranks, c8_host_exchange_fn ex, c8_host_allreduce_fn ar, void* user, c8_comm** out) {
  if (!out || !ex || !ar || nranks < 1 || rank < 0 || rank >= nranks) return c8_fail(C8_ERR_ARG, "c8_comm_create_host: bad argument");
  c8_comm* cm = new c8_comm();
  cm->rank = rank;
  cm->nranks = nranks;
  cm->host_exchange = ex;
  cm->host_allreduce = ar;
  cm->user = user;
  *out = cm;
  return C8_OK;
}

void c8_comm_destroy(c8_comm* cm) {
  if (!cm) return;
  if (cm->nccl) (void)rccl().CommDestroy(cm->nccl);
  if (cm->stream) (void)hipStreamDestroy(cm->stream);
  (void)hipFree(cm->d_small);
  delete cm;
}
int c8_comm_rank(const c8_comm* cm) { return cm ? cm->rank : C8_ERR_ARG; }
int c8_comm_size(const c8_comm* cm) { return cm ? cm->nranks : C8_ERR_ARG; }

int c8_comm_allreduce_sum(c8_comm* cm, double* values, int n) {
  if (!cm || !values || n < 0) return c8_fail(C8_ERR_ARG, "c8_comm_allreduce_sum: bad argument");
  if (n == 0) return C8_OK;
  if (!cm->nccl) {
    if (cm->nranks == 1) return C8_OK;
    if (cm->host_allreduce(cm->user, values, n) != 0) return c8_fail(C8_ERR_ARG, "c8_comm_allreduce_sum: the host all-reduce callback failed");
    return C8_OK;
  }
  for (int o = 0; o < n; o += 64) {  // latency-bound: one packed message of at most 64 doubles at a time
    int const m = std::min(64, n - o);
    C8H_HIP(hipMemcpyAsync(cm->d_small, values + o, m * sizeof(double), hipMemcpyHostToDevice, cm->stream));
    C8H_NCCL(rccl().AllReduce(cm->d_small, cm->d_small, (size_t)m, ncclDouble, ncclSum, cm->nccl, cm->stream));
    C8H_HIP(hipMemcpyAsync(values + o, cm->d_small, m * sizeof(double), hipMemcpyDeviceToHost, cm->stream));
    C8H_HIP(hipStreamSynchronize(cm->stream));
  }
  return C8_OK;
}

// ---- index tables (host only) ------------------------------------------------------------------------------------
int c8_halo_build(int32_t num_nodes, const int64_t* rowptr, const int32_t* colidx, const c8_halo_desc* d, int rank, int nranks, c8_halo** out) {
  if (!out) return c8_fail(C8_ERR_ARG, "c8_halo_build: null argument");
  *out = nullptr;
  if (num_nodes <= 0 || !rowptr || !colidx || !d || nranks < 1 || rank < 0 || rank >= nranks)
    return c8_fail(C8_ERR_ARG, "c8_halo_build: bad argument");
  if (d->num_owned < 0 || d->num_owned > d->num_touched || d->num_touched > num_nodes) return c8_fail(C8_ERR_ARG, "c8_halo_build: owned <= touched <= nodes violated");
  if (!d->send_ptr || !d->recv_ptr || !d->recv_col_ptr || !d->import_ptr || !d->export_ptr) return c8_fail(C8_ERR_ARG, "c8_halo_build: null list offsets");
  auto deg = [&](int32_t n) { return rowptr[n + 1] - rowptr[n]; };
  c8_halo* h = new c8_halo();
  h->rank = rank;
  h->nranks = nranks;
  h->nnodes = num_nodes;
  h->nowned = d->num_owned;
  h->ntouched = d->num_touched;
  h->ndims = d->num_dims == 0 ? 3 : d->num_dims;
  h->nres = d->num_residuals == 0 ? 2 : d->num_residuals;
  if ((h->ndims != 2 && h->ndims != 3) || (h->nres != 1 && h->nres != 2)) { delete h; return c8_fail(C8_ERR_ARG, "c8_halo_build: num_dims must be 2 or 3, num_residuals 1 or 2"); }
  int const nres = h->nres, ndims = h->ndims;
  auto neq_of = [ndims](int i) { return i == 0 ? ndims : 1; };
  h->nodeptr.assign(rowptr, rowptr + num_nodes + 1);
  auto bad = [&](std::string const& m) { delete h; return c8_fail(C8_ERR_ARG, "c8_halo_build: " + m); };

  // ---- export (C1, C2, C1+C2) ----
  // message to rank r: [b0 rows][b1 rows][A00][A01][A10][A11] of my ghost rows owned by r, rows in send order, a row's
  // values in CSR order (equation a, then graph position k, then equation b of the column node)
  int const blk[4][2] = {{0, 0}, {0, 1}, {1, 0}, {1, 1}};
  for (int what = 1; what <= 3; ++what) {
    Exchange& x = what == 3 ? h->full : (what == 1 ? h->bonly : h->aonly);
    x.add = true;
    x.send_counts.assign(nranks, 0);
    x.recv_counts.assign(nranks, 0);
    std::vector<int64_t> dst_of_pos;
    for (int r = 0; r < nranks; ++r) {
      // what I send to r
      int64_t const s0 = d->send_ptr[r], s1 = d->send_ptr[r + 1];
      if (s1 < s0 || (s1 > s0 && !d->send_nodes)) return bad("send lists");
      size_t const before = x.send_idx.size();
      if (what & C8_HALO_B)
        for (int i = 0; i < nres; ++i)
          for (int64_t q = s0; q < s1; ++q) {
            int32_t const n = d->send_nodes[q];
            if (n < d->num_owned || n >= d->num_touched) return bad("a send node is not a ghost node");
            for (int a = 0; a < neq_of(i); ++a) x.send_idx.push_back(code(4 + i, (int64_t)n * neq_of(i) + a));
          }
      if (what & C8_HALO_A)
        for (int b = 0; b < 4; ++b) {
          if (blk[b][0] >= nres || blk[b][1] >= nres) continue;
          int const ni = neq_of(blk[b][0]), nj = neq_of(blk[b][1]);
          for (int64_t q = s0; q < s1; ++q) {
            int32_t const n = d->send_nodes[q];
            if (n < d->num_owned || n >= d->num_touched) return bad("a send node is not a ghost node");
            int64_t const base = rowptr[n] * ni * nj, len = (int64_t)ni * deg(n) * nj;
            for (int64_t k = 0; k < len; ++k) x.send_idx.push_back(code(b, base + k));
          }
        }
      x.send_counts[r] = (int64_t)(x.send_idx.size() - before);
      // what r sends me
      int64_t const r0 = d->recv_ptr[r], r1 = d->recv_ptr[r + 1];
      if (r1 < r0 || (r1 > r0 && (!d->recv_nodes || !d->recv_cols))) return bad("recv lists");
      size_t const before_r = dst_of_pos.size();
      if (what & C8_HALO_B)
        for (int i = 0; i < nres; ++i)
          for (int64_t q = r0; q < r1; ++q) {
            int32_t const n = d->recv_nodes[q];
            if (n < 0 || n >= d->num_owned) return bad("a recv node is not an owned node");
            for (int a = 0; a < neq_of(i); ++a) dst_of_pos.push_back(code(4 + i, (int64_t)n * neq_of(i) + a));
          }
      if (what & C8_HALO_A)
        for (int b = 0; b < 4; ++b) {
          if (blk[b][0] >= nres || blk[b][1] >= nres) continue;
          int const ni = neq_of(blk[b][0]), nj = neq_of(blk[b][1]);
          for (int64_t q = r0; q < r1; ++q) {
            int32_t const n = d->recv_nodes[q];
            if (n < 0 || n >= d->num_owned) return bad("a recv node is not an owned node");
            int64_t const c0 = d->recv_col_ptr[q], c1 = d->recv_col_ptr[q + 1];  // the sender's columns of this row
            int32_t const* rb = colidx + rowptr[n];
            int32_t const* re = colidx + rowptr[n + 1];
            int64_t const dn = deg(n);
            for (int a = 0; a < ni; ++a)
              for (int64_t k = c0; k < c1; ++k) {
                int32_t const* it = std::lower_bound(rb, re, d->recv_cols[k]);
                if (it == re || *it != d->recv_cols[k]) return bad("a received column is missing from the local graph (c8_mesh_desc.extra_pairs)");
                int64_t const pos = it - rb;
                for (int e = 0; e < nj; ++e) dst_of_pos.push_back(code(b, rowptr[n] * ni * nj + a * dn * nj + pos * nj + e));
              }
          }
        }
      x.recv_counts[r] = (int64_t)(dst_of_pos.size() - before_r);
    }
    x.nsend = (int64_t)x.send_idx.size();
    x.nrecv = (int64_t)dst_of_pos.size();
    group_by_destination(dst_of_pos, x);
  }
  // ---- import (C3): owners send the values of export_nodes, importers store them at import_nodes ----
  {
    Exchange& x = h->import;
    x.add = false;
    x.send_counts.assign(nranks, 0);
    x.recv_counts.assign(nranks, 0);
    for (int r = 0; r < nranks; ++r) {
      int64_t const s0 = d->export_ptr[r], s1 = d->export_ptr[r + 1], r0 = d->import_ptr[r], r1 = d->import_ptr[r + 1];
      if (s1 < s0 || r1 < r0 || (s1 > s0 && !d->export_nodes) || (r1 > r0 && !d->import_nodes)) return bad("import / export lists");
      for (int i = 0; i < nres; ++i) {
        for (int64_t q = s0; q < s1; ++q) {
          int32_t const n = d->export_nodes[q];
          if (n < 0 || n >= d->num_owned) return bad("an export node is not an owned node");
          for (int a = 0; a < neq_of(i); ++a) x.send_idx.push_back(code(4 + i, (int64_t)n * neq_of(i) + a));
        }
        for (int64_t q = r0; q < r1; ++q) {
          int32_t const n = d->import_nodes[q];
          if (n < d->num_owned || n >= num_nodes) return bad("an import node is an owned node");
          for (int a = 0; a < neq_of(i); ++a) x.dst.push_back(code(4 + i, (int64_t)n * neq_of(i) + a));
        }
      }
      int const per_node = ndims + (nres == 2 ? 1 : 0);
      x.send_counts[r] = (s1 - s0) * per_node;
      x.recv_counts[r] = (r1 - r0) * per_node;
    }
    x.nsend = (int64_t)x.send_idx.size();
    x.nrecv = (int64_t)x.dst.size();
  }
  *out = h;
  return C8_OK;
}

int c8_halo_table(const c8_halo* h, int which, int64_t* n, const int64_t** data) {
  if (!h || !n || !data) return c8_fail(C8_ERR_ARG, "c8_halo_table: null argument");
  Exchange const* x = which < 10 ? &h->full : (which < 20 ? &h->bonly : &h->import);
  std::vector<int64_t> const* v = nullptr;
  switch (which % 10) {
    case 0: v = &x->send_idx; break;
    case 1: v = &x->send_counts; break;
    case 2: v = &x->recv_counts; break;
    case 3: v = &x->dst; break;
    case 4: v = &x->src_ptr; break;
    case 5: v = &x->src; break;
    default: return c8_fail(C8_ERR_ARG, "c8_halo_table: unknown table");
  }
  *n = (int64_t)v->size();
  *data = v->data();
  return C8_OK;
}

int64_t c8_halo_send_bytes(const c8_halo* h, int what) {
  if (!h) return C8_ERR_ARG;
  Exchange const& x = what == 0 ? h->import : (what == 3 ? h->full : (what == 1 ? h->bonly : h->aonly));
  return x.nsend * (int64_t)sizeof(double);
}

// ---- attach to a context / communicator ---------------------------------------------------------------------------
int c8_halo_attach(c8_halo* h, c8_ctx* c, c8_comm* cm) {
  if (!h || !c || !cm) return c8_fail(C8_ERR_ARG, "c8_halo_attach: null argument");
  if (h->ctx) return c8_fail(C8_ERR_ARG, "c8_halo_attach: already attached");
  if (cm->rank != h->rank || cm->nranks != h->nranks) return c8_fail(C8_ERR_ARG, "c8_halo_attach: communicator rank / size differ from the halo's");
  if (c->mesh.nnodes != h->nnodes) return c8_fail(C8_ERR_ARG, "c8_halo_attach: the context has another number of nodes");
  if (c->ndims != h->ndims || c->nres != h->nres)
    return c8_fail(C8_ERR_ARG, "c8_halo_attach: the tables were built for other systems (c8_halo_desc.num_dims / num_residuals against c8_num_dims / c8_num_residuals)");
  for (int32_t n = 0; n <= h->nnodes; ++n)
    if ((int64_t)c->graph.nodeptr[n] != h->nodeptr[n]) return c8_fail(C8_ERR_ARG, "c8_halo_attach: the context's graph is not the one the tables were built from");
  int rc;
  for (Exchange* x : {&h->full, &h->bonly, &h->aonly, &h->import})
    if ((rc = upload_exchange(*x))) return rc;  // the caller destroys the halo
  int64_t const ns = std::max(std::max(h->full.nsend, h->import.nsend), (int64_t)1);
  int64_t const nr = std::max(std::max(h->full.nrecv, h->import.nrecv), (int64_t)1);
  C8H_HIP(hipMalloc((void**)&h->d_sendbuf, (size_t)ns * sizeof(double)));
  C8H_HIP(hipMalloc((void**)&h->d_recvbuf, (size_t)nr * sizeof(double)));
  if (!cm->nccl) {
    C8H_HIP(hipHostMalloc((void**)&h->h_sendbuf, (size_t)ns * sizeof(double), hipHostMallocDefault));
    C8H_HIP(hipHostMalloc((void**)&h->h_recvbuf, (size_t)nr * sizeof(double), hipHostMallocDefault));
  }
  C8H_HIP(hipEventCreateWithFlags(&h->ev_packed, hipEventDisableTiming));
  C8H_HIP(hipEventCreateWithFlags(&h->ev_arrived, hipEventDisableTiming));
  h->ctx = c;
  h->comm = cm;
  c->halo = h;
  c->num_parts = cm->nranks;
  return C8_OK;
}

void c8_halo_destroy(c8_halo* h) {
  if (!h) return;
  if (h->ctx && h->ctx->halo == h) h->ctx->halo = nullptr;
  // an exchange may still be in flight on the communicator's stream or waiting to be unpacked on the context's
  if (h->comm && h->comm->stream) (void)hipStreamSynchronize(h->comm->stream);
  if (h->ctx) (void)hipStreamSynchronize(h->ctx->stream);
  for (Exchange* x : {&h->full, &h->bonly, &h->aonly, &h->import}) free_exchange(*x);
  (void)hipFree(h->d_sendbuf);
  (void)hipFree(h->d_recvbuf);
  if (h->h_sendbuf) (void)hipHostFree(h->h_sendbuf);
  if (h->h_recvbuf) (void)hipHostFree(h->h_recvbuf);
  if (h->ev_packed) (void)hipEventDestroy(h->ev_packed);
  if (h->ev_arrived) (void)hipEventDestroy(h->ev_arrived);
  delete h;
}

// ---- run time -------------------------------------------------------------------------------------------------
int c8_halo_gather_start(c8_halo* h, const c8_system* sys, int what) {
  if (!h || !h->ctx || !sys) return c8_fail(C8_ERR_ARG, "c8_halo_gather_start: null argument or halo not attached");
  if (what < 1 || what > 3) return c8_fail(C8_ERR_ARG, "c8_halo_gather_start: what = C8_HALO_B | C8_HALO_A");
  for (int i = 0; i < h->nres; ++i) {
    if ((what & C8_HALO_B) && !sys->b[i]) return c8_fail(C8_ERR_ARG, "c8_halo_gather_start: null b");
    for (int j = 0; j < h->nres; ++j)
      if ((what & C8_HALO_A) && !sys->A[i][j]) return c8_fail(C8_ERR_ARG, "c8_halo_gather_start: null A block");
  }
  return start_exchange(h, what == 3 ? h->full : (what == 1 ? h->bonly : h->aonly), system_segs(sys));
}
int c8_halo_gather_finish(c8_halo* h, const c8_system* sys) {
  if (!h || !h->ctx || !sys) return c8_fail(C8_ERR_ARG, "c8_halo_gather_finish: null argument or halo not attached");
  return finish_exchange(h, system_segs(sys));
}
int c8_halo_gather(c8_halo* h, const c8_system* sys, int what) {
  int const rc = c8_halo_gather_start(h, sys, what);
  return rc ? rc : c8_halo_gather_finish(h, sys);
}
int c8_halo_scatter_x(c8_halo* h, double* const x[2]) {
  if (!h || !h->ctx || !x || !x[0] || (h->nres == 2 && !x[1])) return c8_fail(C8_ERR_ARG, "c8_halo_scatter_x: null argument or halo not attached");
  Segs const segs{{nullptr, nullptr, nullptr, nullptr, x[0], x[1]}};
  int const rc = start_exchange(h, h->import, segs);
  return rc ? rc : finish_exchange(h, segs);
}

}  // extern "C"
