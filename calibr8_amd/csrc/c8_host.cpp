// c8_host.cpp -- see c8_host.hpp
#include "c8_host.hpp"

#include <algorithm>
#include <cstring>

namespace c8 {

std::string build_node_graph(HostMesh const& m, HostGraph& g) {
  int const nn = m.nn;
  // count-then-fill adjacency with duplicates, then sort/unique per row
  std::vector<int64_t> cnt((size_t)m.nnodes + 1, 0);
  for (int e = 0; e < m.nelems; ++e)
    for (int a = 0; a < nn; ++a) {
      int const na = m.conn[(size_t)e * nn + a];
      if (na < 0 || na >= m.nnodes) return "connectivity entry out of range";
      cnt[(size_t)na + 1] += nn;
    }
  size_t const nextra = m.extra_pairs.size() / 2;
  for (size_t q = 0; q < nextra; ++q) {
    int const r = m.extra_pairs[2 * q], c = m.extra_pairs[2 * q + 1];
    if (r < 0 || r >= m.nnodes || c < 0 || c >= m.nnodes) return "extra graph pair out of range";
    cnt[(size_t)r + 1] += 1;
  }
  for (int n = 0; n < m.nnodes; ++n) cnt[n + 1] += cnt[n];
  std::vector<int32_t> raw((size_t)cnt[m.nnodes]);
  std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
  for (int e = 0; e < m.nelems; ++e)
    for (int a = 0; a < nn; ++a) {
      int const na = m.conn[(size_t)e * nn + a];
      for (int b = 0; b < nn; ++b) raw[(size_t)fill[na]++] = m.conn[(size_t)e * nn + b];
    }
  for (size_t q = 0; q < nextra; ++q) raw[(size_t)fill[m.extra_pairs[2 * q]]++] = m.extra_pairs[2 * q + 1];
  g.nodeptr.assign((size_t)m.nnodes + 1, 0);
  g.nodeadj.clear();
  g.nodeadj.reserve(raw.size() / 2);
  for (int n = 0; n < m.nnodes; ++n) {
    int32_t* b = raw.data() + cnt[n];
    int32_t* e = raw.data() + cnt[n + 1];
    std::sort(b, e);
    e = std::unique(b, e);
    if (e - b > 255) return "node degree exceeds 255 (position table is uint8)";
    g.nodeadj.insert(g.nodeadj.end(), b, e);
    if (g.nodeadj.size() > (size_t)0x7fffffff / 9) return "node graph too large for int32 CSR offsets";
    g.nodeptr[(size_t)n + 1] = (int32_t)g.nodeadj.size();
  }
  g.max_degree = 0;
  for (int n = 0; n < m.nnodes; ++n) g.max_degree = std::max(g.max_degree, (int)(g.nodeptr[n + 1] - g.nodeptr[n]));
  // node -> elements (gather-mode assembly sums a node's rows over its elements, in ascending element order)
  if (nn > 8 || m.nelems >= (1 << 28)) return "node-to-element table needs nn <= 8 and fewer than 2^28 elements";
  g.nodeelem_ptr.assign((size_t)m.nnodes + 1, 0);
  for (size_t q = 0; q < m.conn.size(); ++q) g.nodeelem_ptr[(size_t)m.conn[q] + 1]++;
  g.max_node_elems = 0;
  for (int n = 0; n < m.nnodes; ++n) g.max_node_elems = std::max(g.max_node_elems, (int)g.nodeelem_ptr[n + 1]);
  for (int n = 0; n < m.nnodes; ++n) g.nodeelem_ptr[n + 1] += g.nodeelem_ptr[n];
  g.nodeelem.assign(m.conn.size(), 0);
  {
    std::vector<int32_t> fill2(g.nodeelem_ptr.begin(), g.nodeelem_ptr.end() - 1);
    for (int e = 0; e < m.nelems; ++e)
      for (int a = 0; a < nn; ++a) g.nodeelem[(size_t)fill2[m.conn[(size_t)e * nn + a]]++] = (e << 3) | a;
  }
  g.pos.assign((size_t)m.nelems * nn * nn, 0);
  for (int e = 0; e < m.nelems; ++e)
    for (int r = 0; r < nn; ++r) {
      int const nr = m.conn[(size_t)e * nn + r];
      int32_t const* rb = g.nodeadj.data() + g.nodeptr[nr];
      int32_t const* re = g.nodeadj.data() + g.nodeptr[nr + 1];
      for (int c = 0; c < nn; ++c) {
        int const nc = m.conn[(size_t)e * nn + c];
        g.pos[((size_t)e * nn + c) * nn + r] = (uint8_t)(std::lower_bound(rb, re, nc) - rb);
      }
    }
  return "";
}

void plan_staged_assembly(HostMesh const& m, HostGraph const& g, int min_chunk, int align, StagePlan& plan) {
  int bw = 1;
  std::vector<int32_t> last((size_t)m.nnodes, -1);
  for (int n = 0; n < m.nnodes; ++n) {
    int const a = g.nodeelem_ptr[n], b = g.nodeelem_ptr[n + 1];
    if (a == b) continue;
    int const lo = g.nodeelem[a] >> 3, hi = g.nodeelem[b - 1] >> 3;  // lists are ascending
    bw = std::max(bw, hi - lo + 1);
    last[n] = hi;
  }
  int chunk = std::max(bw, min_chunk);
  chunk = ((chunk + align - 1) / align) * align;
  if ((int64_t)chunk * 4 >= m.nelems) {  // not worth a ring
    plan.chunk = m.nelems;
    plan.nchunks = 1;
    plan.ring = m.nelems;
  } else {
    plan.chunk = chunk;
    plan.nchunks = (m.nelems + chunk - 1) / chunk;
    plan.ring = 3 * chunk;
  }
  plan.node_off.assign((size_t)plan.nchunks + 1, 0);
  for (int n = 0; n < m.nnodes; ++n)
    if (last[n] >= 0) plan.node_off[(size_t)(last[n] / plan.chunk) + 1]++;
  for (int k = 0; k < plan.nchunks; ++k) plan.node_off[k + 1] += plan.node_off[k];
  plan.node_order.assign((size_t)plan.node_off[plan.nchunks], 0);
  std::vector<int32_t> fill(plan.node_off.begin(), plan.node_off.end() - 1);
  for (int n = 0; n < m.nnodes; ++n)
    if (last[n] >= 0) plan.node_order[(size_t)fill[last[n] / plan.chunk]++] = n;
}


int64_t block_nnz(HostGraph const& g, int nnodes, int i, int j, int ndims) {
  int const neq[2] = {ndims, 1};
  return (int64_t)g.nodeptr[nnodes] * neq[i] * neq[j];
}

void block_csr(HostGraph const& g, int nnodes, int i, int j, int64_t* rowptr, int32_t* colidx, int ndims) {
  int const neq[2] = {ndims, 1};
  int const ni = neq[i], nj = neq[j];
  int64_t w = 0;
  rowptr[0] = 0;
  for (int n = 0; n < nnodes; ++n)
    for (int ei = 0; ei < ni; ++ei) {
      for (int32_t k = g.nodeptr[n]; k < g.nodeptr[n + 1]; ++k)
        for (int ej = 0; ej < nj; ++ej) colidx[w++] = g.nodeadj[k] * nj + ej;
      rowptr[(size_t)n * ni + ei + 1] = w;
    }
}

std::string color_elements(HostMesh const& m, std::vector<int32_t>& order, std::vector<int32_t>& offsets) {
  int const nn = m.nn;
  std::vector<uint64_t> used((size_t)m.nnodes, 0);
  std::vector<uint8_t> color((size_t)m.nelems, 0);
  int ncolors = 0;
  for (int e = 0; e < m.nelems; ++e) {
    uint64_t mask = 0;
    for (int a = 0; a < nn; ++a) mask |= used[m.conn[(size_t)e * nn + a]];
    if (~mask == 0) return "more than 64 element colours needed";
    int c = 0;
    while ((mask >> c) & 1) ++c;
    color[e] = (uint8_t)c;
    for (int a = 0; a < nn; ++a) used[m.conn[(size_t)e * nn + a]] |= (uint64_t)1 << c;
    if (c + 1 > ncolors) ncolors = c + 1;
  }
  offsets.assign((size_t)ncolors + 1, 0);
  for (int e = 0; e < m.nelems; ++e) offsets[(size_t)color[e] + 1]++;
  for (int c = 0; c < ncolors; ++c) offsets[c + 1] += offsets[c];
  order.resize((size_t)m.nelems);
  std::vector<int32_t> fill(offsets.begin(), offsets.end() - 1);
  for (int e = 0; e < m.nelems; ++e) order[(size_t)fill[color[e]]++] = e;
  return "";
}

void make_brick(int nx, int ny, int nz, double lx, double ly, double lz, HostMesh& m) {
  m.elem_type = 8;
  m.nn = 8;
  m.nnodes = (nx + 1) * (ny + 1) * (nz + 1);
  m.nelems = nx * ny * nz;
  m.nsets = 1;
  m.elem_set.clear();
  m.coords.resize((size_t)m.nnodes * 3);
  m.conn.resize((size_t)m.nelems * 8);
  auto nid = [&](int i, int j, int k) { return (k * (ny + 1) + j) * (nx + 1) + i; };
  for (int k = 0; k <= nz; ++k)
    for (int j = 0; j <= ny; ++j)
      for (int i = 0; i <= nx; ++i) {
        size_t const n = (size_t)nid(i, j, k);
        m.coords[n * 3 + 0] = lx * i / nx;
        m.coords[n * 3 + 1] = ly * j / ny;
        m.coords[n * 3 + 2] = lz * k / nz;
      }
  size_t e = 0;
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i, ++e) {
        int32_t* c = &m.conn[e * 8];
        c[0] = nid(i, j, k); c[1] = nid(i + 1, j, k); c[2] = nid(i + 1, j + 1, k); c[3] = nid(i, j + 1, k);
        c[4] = nid(i, j, k + 1); c[5] = nid(i + 1, j, k + 1); c[6] = nid(i + 1, j + 1, k + 1); c[7] = nid(i, j + 1, k + 1);
      }
}

void brick_partition(int nx, int ny, int nz, int px, int py, int pz, std::vector<int32_t>& elem_part) {
  elem_part.resize((size_t)nx * ny * nz);
  size_t e = 0;
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i, ++e) {
        int const a = (int)((int64_t)i * px / nx), b = (int)((int64_t)j * py / ny), c = (int)((int64_t)k * pz / nz);
        elem_part[e] = (c * py + b) * px + a;
      }
}

}  // namespace c8
