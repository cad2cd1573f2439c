// c8_host.hpp -- host-side tables that feed the assembly kernels.
//
// The minimal mesh/graph layer the kernels need, standing in for the parts of
// the reference's Disc that the hot path consumes (disc.cpp:263-265 get_dof,
// :356-387 compute_ghost_graph, :414-459 compute_scatter_offsets, :461-484
// compute_elem_lids).  Pure C++, no device code.
#pragma once

#include <stdint.h>

#include <string>
#include <vector>

namespace c8 {

struct HostMesh {
  int elem_type = 0;  // C8_TET4 / C8_HEX8
  int nn = 0;
  int nnodes = 0, nelems = 0, nsets = 1;
  std::vector<double> coords;     // [nnodes][3]
  std::vector<int32_t> conn;      // [nelems][nn]
  std::vector<int32_t> elem_set;  // [nelems] (empty = one set)
  std::vector<int32_t> extra_pairs;  // [n][2] extra (row node, col node) graph entries
};

struct HostGraph {
  std::vector<int32_t> nodeptr;   // [nnodes+1]
  std::vector<int32_t> nodeadj;   // sorted neighbour node ids (a node neighbours itself)
  std::vector<uint8_t> pos;       // [nelems][nn(col node)][nn(row node)]
  std::vector<int32_t> nodeelem_ptr;  // [nnodes+1] offsets into nodeelem
  std::vector<int32_t> nodeelem;      // elements of a node, ascending, packed (element << 3) | local node index
  int max_degree = 0;                 // longest row of the node graph
  int max_node_elems = 0;             // most elements around one node
};

// Node-to-node graph with sorted rows, plus the per-element position table.
// Returns an empty string on success, otherwise an error message.
std::string build_node_graph(HostMesh const& m, HostGraph& g);

// Plan of the staged (gather) assembly: elements are processed in chunks of `chunk` consecutive elements whose
// matrices go to a ring of `ring` element slots (slot = element mod ring); the rows of a node are summed once
// the chunk holding its last element is done.  `chunk` is at least the element bandwidth of the mesh (largest
// spread of element ids around one node), so a node's elements lie in at most two consecutive chunks and a ring
// of three chunks is enough; meshes too small or too scattered for that get one chunk and ring = nelems.
struct StagePlan {
  int chunk = 0, nchunks = 0, ring = 0;
  std::vector<int32_t> node_order;  // nodes with elements, sorted by the chunk of their last element
  std::vector<int32_t> node_off;    // [nchunks+1] offsets into node_order
};
void plan_staged_assembly(HostMesh const& m, HostGraph const& g, int min_chunk, int align, StagePlan& plan);

// CSR of block (i,j), i,j in {0:u (ndims eqs: 3, or 2 on tri3 meshes), 1:p (1 eq)}: row dof = node*neq_i+eq_i,
// columns sorted, all equations of a neighbour node contiguous -- the layout
// Tpetra builds in compute_ghost_graph.
int64_t block_nnz(HostGraph const& g, int nnodes, int i, int j, int ndims = 3);
void block_csr(HostGraph const& g, int nnodes, int i, int j, int64_t* rowptr, int32_t* colidx, int ndims = 3);

// Greedy element colouring: elements of one colour share no node, so a launch
// over one colour can read-modify-write CSR values and residual entries without
// atomics.  `order` lists elements colour by colour; `offsets` has ncolors+1 entries.
std::string color_elements(HostMesh const& m, std::vector<int32_t>& order, std::vector<int32_t>& offsets);

// Structured hex8 brick, x-fastest node numbering (SURVEY.md section 8d synthetic meshes).
void make_brick(int nx, int ny, int nz, double lx, double ly, double lz, HostMesh& m);

// Block partition of a structured brick into px*py*pz parts (stands in for the
// reference's offline ParMETIS/Zoltan split, which is not available here).
void brick_partition(int nx, int ny, int nz, int px, int py, int pz, std::vector<int32_t>& elem_part);

}  // namespace c8
