"""Owned/ghost halo exchange and reductions for the multi-GPU assembly.

One process per GPU, one mesh part per process.  Elements are not ghosted; nodes on part
boundaries are shared; every rank assembles its own elements into GHOST-distributed A and b
(all local nodes) and the ghost rows are then ADDed into their owners -- exactly the reference's
MPI scheme, with torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests) in place of Tpetra Export/Import and PCU:

  C1  gather_b      LinearAlg::gather_b   linear_alg.cpp:78-86   ghost residual rows -> owner, ADD
  C2  gather_A      LinearAlg::gather_A   linear_alg.cpp:53-63   ghost Jacobian rows -> owner, ADD
  C3  scatter_x     apf::synchronize in Disc::add_to_soln, disc.cpp:944-947   owner -> ghosts, COPY
  C4  allreduce     PCU_Add_Doubles(grad) adjoint_objective.cpp:109; PCU_Add_Double(J) :39,:99;
  C5                PCU_Add_Int(status)   primal.cpp:100,164 -- packed into one buffer

Layout.  Local node numbering of a part: OWNED nodes, then GHOST nodes (touched by a local element,
owned elsewhere), then PHANTOM nodes (not touched locally; they only appear as columns of owned
interface rows), each group sorted by global id.  The phantom columns are reserved in the local
graphs at c8_create (c8_mesh_desc.extra_pairs), so the first `nowned` node rows of the local arrays
already have the union pattern the reference builds in compute_owned_graph (disc.cpp:389-398): the
OWNED matrix / vector are PREFIX VIEWS of the local arrays and the halo ADD lands in place, with no
second copy of a multi-GB matrix.  Owner of a shared node = lowest part id (PUMI's rule is not
visible in the reference; any deterministic rule is equivalent).

The exchanges are neighbour exchanges (one grouped all_to_all with per-neighbour split sizes:
point-to-point over xGMI links, not a ring).  All index tables are built once at setup on the host.
"""
import numpy as np

NEQ = (3, 1)


class Part:
    """One rank's mesh part before the phantom columns are known."""

    def __init__(self, rank, world, conn, node_gid, node_owner, coords_of, num_global_nodes):
        self.rank, self.world = rank, world
        self.conn = np.ascontiguousarray(conn, dtype=np.int32)
        self.node_gid = np.ascontiguousarray(node_gid, dtype=np.int64)       # owned, then ghosts
        self.node_owner = np.ascontiguousarray(node_owner, dtype=np.int32)
        self.coords_of = coords_of
        self.num_global_nodes = int(num_global_nodes)
        self.nowned = int((self.node_owner == rank).sum())
        assert (self.node_owner[:self.nowned] == rank).all(), "owned nodes must come first"
        self.ntouched = len(self.node_gid)


def _finish_part(rank, world, coords_of, gids, owner_of, conn_g, nglobal):
    owner = owner_of(gids)
    node_gid = np.concatenate([np.sort(gids[owner == rank]), np.sort(gids[owner != rank])])
    order = np.argsort(node_gid)
    conn = order[np.searchsorted(node_gid[order], conn_g)].astype(np.int32)
    return Part(rank, world, conn, node_gid, owner_of(node_gid), coords_of, nglobal)


def part_from_global(coords, conn, elem_part, rank, world):
    """Cut rank's part out of a global mesh (small meshes / tests).  Stands in for the reference's
    offline `split` (disc.cpp:31-39 loads one pre-split part per rank)."""
    coords, conn, elem_part = np.asarray(coords, dtype=np.float64), np.asarray(conn), np.asarray(elem_part)
    node_owner = np.full(len(coords), world, dtype=np.int32)
    for r in range(world - 1, -1, -1):
        node_owner[np.unique(conn[elem_part == r])] = r  # lowest part id wins
    mine = conn[elem_part == rank]
    gids = np.unique(mine).astype(np.int64)
    return _finish_part(rank, world, lambda g: coords[g], gids, lambda g: node_owner[g], mine.astype(np.int64),
                        len(coords))


def brick_part(rank, pdims, n, edge=1.0):
    """Rank's n^3-element block of a (px*n, py*n, pz*n) hex8 brick with element edge `edge`/n, built
    without the global mesh (the weak-scaling workload: BASELINE.json config 5)."""
    px, py, pz = pdims
    world = px * py * pz
    bx, by, bz = rank % px, (rank // px) % py, rank // (px * py)
    NX, NY, NZ = px * n + 1, py * n + 1, pz * n + 1
    i = np.arange(n + 1) + bx * n
    j = np.arange(n + 1) + by * n
    k = np.arange(n + 1) + bz * n
    K, J, I = np.meshgrid(k, j, i, indexing="ij")
    g = (K * NY + J) * NX + I

    def coords_of(gid):
        gi, gj, gk = gid % NX, (gid // NX) % NY, gid // (NX * NY)
        h = edge / n
        return np.stack([gi * h, gj * h, gk * h], axis=1).astype(np.float64)

    def owner_of(gid):
        gi, gj, gk = gid % NX, (gid // NX) % NY, gid // (NX * NY)
        # blocks containing a node: the one it lies in and, on a lower face, the previous one;
        # rank is monotone in each block coordinate, so the lowest sharer takes every "previous"
        ox = np.minimum(gi // n, px - 1) - ((gi % n == 0) & (gi > 0) & (gi < px * n)).astype(np.int64)
        oy = np.minimum(gj // n, py - 1) - ((gj % n == 0) & (gj > 0) & (gj < py * n)).astype(np.int64)
        oz = np.minimum(gk // n, pz - 1) - ((gk % n == 0) & (gk > 0) & (gk < pz * n)).astype(np.int64)
        return ((oz * py + oy) * px + ox).astype(np.int32)

    kk, jj, ii = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    ii, jj, kk = ii.ravel(), jj.ravel(), kk.ravel()
    conn_g = np.stack([g[kk, jj, ii], g[kk, jj, ii + 1], g[kk, jj + 1, ii + 1], g[kk, jj + 1, ii],
                       g[kk + 1, jj, ii], g[kk + 1, jj, ii + 1], g[kk + 1, jj + 1, ii + 1], g[kk + 1, jj + 1, ii]],
                      axis=1).astype(np.int64)
    return _finish_part(rank, world, coords_of, np.unique(g).astype(np.int64), owner_of, conn_g, NX * NY * NZ)


def pdims_for(world):
    """Block grid for `world` parts: 1, 2x1x1, 2x2x1, 2x2x2, ..."""
    d = [1, 1, 1]
    k = 0
    while d[0] * d[1] * d[2] < world:
        d[k % 3] *= 2
        k += 1
    assert d[0] * d[1] * d[2] == world, "world size must be a power of two"
    return tuple(d)


class HaloPlan:
    """Step 1 of the setup (before the assembler exists): exchange the column lists of ghost rows,
    find the phantom nodes and the extra graph pairs.  `dist` is an initialised torch.distributed
    module, or None for a single part."""

    def __init__(self, part, dist=None):
        self.part, self.dist = part, dist
        p = part
        self.world = p.world if dist is not None else 1
        nn = p.conn.shape[1]
        # element adjacency of ghost rows only (gids, in ascending LOCAL id order = the row order of the graph)
        ghost_lo = p.nowned
        touching = np.nonzero((p.conn >= ghost_lo).any(axis=1))[0]
        # elements that add into ghost rows: assembled first, so that the exchange can overlap the rest
        self.interface_elems = touching.astype(np.int32)
        mask = np.ones(len(p.conn), dtype=bool)
        mask[touching] = False
        self.interior_elems = np.nonzero(mask)[0].astype(np.int32)
        rows = np.repeat(p.conn[touching], nn, axis=1).reshape(-1)          # row node of every (a, b) pair
        cols = np.tile(p.conn[touching], (1, nn)).reshape(-1)
        keep = rows >= ghost_lo
        key = np.unique(rows[keep].astype(np.int64) * (p.ntouched + 1) + cols[keep])
        g_rows, g_cols = key // (p.ntouched + 1), key % (p.ntouched + 1)    # sorted by (row, col local id)
        ghosts = np.arange(p.nowned, p.ntouched)
        start = np.searchsorted(g_rows, ghosts)
        end = np.searchsorted(g_rows, ghosts, side="right")
        owners = p.node_owner[ghosts]
        self.send_rows = [ghosts[owners == r] for r in range(self.world)]   # local ghost ids per owner rank
        msgs = []
        for r in range(self.world):
            sel = np.nonzero(owners == r)[0]
            col_lists = [p.node_gid[g_cols[start[s]:end[s]]] for s in sel]
            msgs.append((p.node_gid[ghosts[sel]], col_lists))
        if self.world > 1:
            gathered = [None] * self.world
            dist.all_gather_object(gathered, msgs)
            self.incoming = [gathered[src][p.rank] for src in range(self.world)]
        else:
            self.incoming = [msgs[0]]
        # phantom nodes: received column gids that are not local
        local = {int(g): n for n, g in enumerate(p.node_gid)}
        phantom = set()
        for src in range(self.world):
            if src == p.rank:
                continue
            for cl in self.incoming[src][1]:
                for g in cl.tolist():
                    if g not in local:
                        phantom.add(g)
        self.phantom_gid = np.array(sorted(phantom), dtype=np.int64)
        for k, g in enumerate(self.phantom_gid):
            local[int(g)] = p.ntouched + k
        self.gid2local = local
        self.nnodes = p.ntouched + len(self.phantom_gid)
        self.node_gid = np.concatenate([p.node_gid, self.phantom_gid])
        self.coords = p.coords_of(self.node_gid)
        # extra graph pairs: every received (row, col) coupling (duplicates of local ones are harmless)
        pairs = []
        self.recv_rows_local, self.recv_cols_local = [], []
        for src in range(self.world):
            if src == p.rank or self.world == 1:
                self.recv_rows_local.append(np.zeros(0, dtype=np.int64))
                self.recv_cols_local.append([])
                continue
            row_gids, col_lists = self.incoming[src]
            rl = np.array([local[int(g)] for g in row_gids], dtype=np.int64)
            cls = [np.array([local[int(g)] for g in cl.tolist()], dtype=np.int64) for cl in col_lists]
            self.recv_rows_local.append(rl)
            self.recv_cols_local.append(cls)
            for r, cl in zip(rl, cls):
                pairs.append(np.stack([np.full(len(cl), r), cl], axis=1))
        self.extra_pairs = np.concatenate(pairs).astype(np.int32) if pairs else np.zeros((0, 2), dtype=np.int32)


class Halo:
    """Step 2: index tables against the assembler's graphs, and the runtime exchanges.
    rowptr_uu / colidx_uu = block (0,0) of c8_graph() of the assembler created with plan.extra_pairs."""

    def __init__(self, plan, rowptr_uu, colidx_uu, device="cpu"):
        import torch
        self.torch, self.plan, self.dist = torch, plan, plan.dist
        self.device = torch.device(device)
        self.world, p = plan.world, plan.part
        self.nowned = p.nowned
        rp = np.asarray(rowptr_uu)[::3] // 9                      # node-level row offsets
        deg = np.diff(rp)
        adj = (np.asarray(colidx_uu)[np.repeat(rp[:-1] * 9, deg) + _ramp(deg) * 3] // 3).astype(np.int64)
        self.nodeptr = rp.astype(np.int64)
        # send: all node pairs of my ghost rows, row by row, in graph order
        self.send_pairs = [np.concatenate([np.arange(rp[n], rp[n + 1]) for n in rows]) if len(rows) else
                           np.zeros(0, dtype=np.int64) for rows in plan.send_rows]
        # recv: position of each received (row, col) in my graph
        self.recv_pairs = []
        for src in range(self.world):
            dst = [rp[r] + np.searchsorted(adj[rp[r]:rp[r + 1]], cl)
                   for r, cl in zip(plan.recv_rows_local[src], plan.recv_cols_local[src])]
            self.recv_pairs.append(np.concatenate(dst).astype(np.int64) if dst else np.zeros(0, dtype=np.int64))
        T = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int64), device=self.device)
        W = range(self.world)
        self.A_send = [[[T(_expand(self.send_pairs[r], self.nodeptr, i, j)) for r in W] for j in range(2)] for i in range(2)]
        self.A_recv = [[[T(_expand(self.recv_pairs[r], self.nodeptr, i, j)) for r in W] for j in range(2)] for i in range(2)]
        dof = lambda rows, i: (np.asarray(rows, dtype=np.int64)[:, None] * NEQ[i] + np.arange(NEQ[i])[None, :]).reshape(-1)
        self.b_send = [[T(dof(plan.send_rows[r], i)) for r in W] for i in range(2)]
        self.b_recv = [[T(dof(plan.recv_rows_local[r], i)) for r in W] for i in range(2)]
        self.neighbours = [r for r in W if len(plan.send_rows[r]) or len(plan.recv_rows_local[r])]
        self.bytes_per_gather_A = 8 * sum(int(self.A_send[i][j][r].numel()) for i in range(2) for j in range(2) for r in W)
        self.bytes_per_gather_b = 8 * sum(int(self.b_send[i][r].numel()) for i in range(2) for r in W)

    def _exchange(self, send_chunks, recv_sizes):
        """One grouped neighbour exchange: chunk r goes to rank r; returns the received chunks."""
        t = self.torch
        sbuf = t.cat(send_chunks)
        out_sizes, in_sizes = [int(s) for s in recv_sizes], [int(c.numel()) for c in send_chunks]
        if sbuf.is_cuda and self.dist.get_backend() == "gloo":
            # rehearsal mode (several ranks sharing one GPU, no RCCL): stage through the host
            rh = t.empty(sum(out_sizes), dtype=sbuf.dtype)
            self.dist.all_to_all_single(rh, sbuf.cpu(), out_sizes, in_sizes)
            rbuf = rh.to(sbuf.device)
        else:
            rbuf = t.empty(sum(out_sizes), dtype=sbuf.dtype, device=sbuf.device)
            self.dist.all_to_all_single(rbuf, sbuf, out_sizes, in_sizes)
        return t.split(rbuf, out_sizes)

    def gather_b(self, b):
        """C1, in place: afterwards b[i][: nowned*neq_i] is the OWNED residual."""
        if self.world == 1:
            return b
        W = range(self.world)
        send = [b[i][self.b_send[i][r]] for i in range(2) for r in W]
        sizes = [self.b_recv[i][r].numel() for i in range(2) for r in W]
        # one exchange for both blocks: chunks ordered (rank-major inside each block) -> regroup by rank
        got = self._exchange_blocks(send, sizes, 2)
        for i in range(2):
            for r in W:
                if self.b_recv[i][r].numel():
                    b[i].index_add_(0, self.b_recv[i][r], got[i][r])
        return b

    def gather_A(self, A):
        """C2, in place: afterwards the rows of the first `nowned` nodes of every block hold the OWNED
        Jacobian (union pattern, columns in local numbering; local->global ids = plan.node_gid)."""
        if self.world == 1:
            return A
        W = range(self.world)
        blocks = [(i, j) for i in range(2) for j in range(2)]
        send = [A[i][j][self.A_send[i][j][r]] for (i, j) in blocks for r in W]
        sizes = [self.A_recv[i][j][r].numel() for (i, j) in blocks for r in W]
        got = self._exchange_blocks(send, sizes, 4)
        for q, (i, j) in enumerate(blocks):
            for r in W:
                if self.A_recv[i][j][r].numel():
                    A[i][j].index_add_(0, self.A_recv[i][j][r], got[q][r])
        return A

    def _exchange_blocks(self, send, sizes, nblk):
        """send/sizes are block-major lists of per-rank chunks; do ONE all_to_all for all blocks."""
        W = self.world
        by_rank_send = [self.torch.cat([send[q * W + r] for q in range(nblk)]) for r in range(W)]
        by_rank_sizes = [sum(sizes[q * W + r] for q in range(nblk)) for r in range(W)]
        got = self._exchange(by_rank_send, by_rank_sizes)
        out = [[None] * W for _ in range(nblk)]
        for r in range(W):
            parts = self.torch.split(got[r], [int(sizes[q * W + r]) for q in range(nblk)])
            for q in range(nblk):
                out[q][r] = parts[q]
        return out

    # ---- C1 + C2 as ONE exchange that overlaps the assembly of the interior elements ----------------------
    def flat_tables(self, offsets):
        """Index tables into a LinearSystem.flat laid out as A00 A01 A10 A11 b0 b1 (`offsets` = its 7 offsets):
        send indices in rank-major order (chunk r = everything rank r receives from me) and the matching
        receive indices."""
        t, W = self.torch, range(self.world)
        off = [int(o) for o in offsets]
        blocks = [(i, j) for i in range(2) for j in range(2)]
        send, recv = [], []
        self._flat_in, self._flat_out = [], []
        for r in W:
            s_r = [self.A_send[i][j][r] + off[q] for q, (i, j) in enumerate(blocks)] + [self.b_send[i][r] + off[4 + i] for i in range(2)]
            r_r = [self.A_recv[i][j][r] + off[q] for q, (i, j) in enumerate(blocks)] + [self.b_recv[i][r] + off[4 + i] for i in range(2)]
            send.append(t.cat(s_r))
            recv.append(t.cat(r_r))
            self._flat_in.append(int(send[-1].numel()))
            self._flat_out.append(int(recv[-1].numel()))
        self._flat_send, self._flat_recv = t.cat(send), t.cat(recv)

    def start_gather(self, ls):
        """Pack the ghost rows of A and b (one gather) and start the exchange; returns a handle.  Everything
        that adds into ghost rows must have been enqueued before; work enqueued after this call (the assembly
        of the interior elements, which only touches owned rows) runs while the exchange is in flight."""
        if self.world == 1:
            return None
        t = self.torch
        if not hasattr(self, "_flat_send"):
            self.flat_tables(ls.offsets)
        sbuf = ls.flat[self._flat_send]
        if sbuf.is_cuda and self.dist.get_backend() == "gloo":  # rehearsal: several ranks on one GPU, via the host
            sh = sbuf.cpu()
            rbuf = t.empty(sum(self._flat_out), dtype=sbuf.dtype)
            work = self.dist.all_to_all_single(rbuf, sh, self._flat_out, self._flat_in, async_op=True)
            return (work, rbuf, sh)
        rbuf = t.empty(sum(self._flat_out), dtype=sbuf.dtype, device=sbuf.device)
        work = self.dist.all_to_all_single(rbuf, sbuf, self._flat_out, self._flat_in, async_op=True)
        return (work, rbuf, sbuf)

    def finish_gather(self, ls, handle):
        """Wait for the exchange and ADD the received ghost contributions into the owned rows (one index_add;
        atomic adds, so it may run while other kernels still add into the same rows)."""
        if handle is None:
            return ls
        work, rbuf, _keep = handle
        work.wait()
        ls.flat.index_add_(0, self._flat_recv, rbuf.to(ls.flat.device))
        return ls

    def scatter_x(self, x):
        """C3, in place: owner values of a nodal field pair x = [u, p] copied to the ghost copies."""
        if self.world == 1:
            return x
        W = range(self.world)
        send = [x[i][self.b_recv[i][r]] for i in range(2) for r in W]
        sizes = [self.b_send[i][r].numel() for i in range(2) for r in W]
        got = self._exchange_blocks(send, sizes, 2)
        for i in range(2):
            for r in W:
                if self.b_send[i][r].numel():
                    x[i][self.b_send[i][r]] = got[i][r]
        return x

    def allreduce(self, values):
        """C4/C5: one SUM all-reduce of a small float64 tensor (gradient, objective, failure flag packed)."""
        if self.world > 1:
            if values.is_cuda and self.dist.get_backend() == "gloo":
                h = values.cpu()
                self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM)
                values.copy_(h)
            else:
                self.dist.all_reduce(values, op=self.dist.ReduceOp.SUM)
        return values


def _expand(pairs, nodeptr, i, j):
    """dof-level value indices of node pairs in block (i,j): row n, eq a, pair position k, eq b ->
    nodeptr[n]*ni*nj + a*deg[n]*nj + k*nj + b  (the layout of c8_graph)."""
    ni, nj = NEQ[i], NEQ[j]
    pairs = np.asarray(pairs, dtype=np.int64)
    if len(pairs) == 0:
        return np.zeros(0, dtype=np.int64)
    row = np.searchsorted(nodeptr, pairs, side="right") - 1
    k = pairs - nodeptr[row]
    deg = nodeptr[row + 1] - nodeptr[row]
    a = np.arange(ni)[None, :, None]
    b = np.arange(nj)[None, None, :]
    idx = (nodeptr[row] * ni * nj)[:, None, None] + a * (deg * nj)[:, None, None] + (k * nj)[:, None, None] + b
    return idx.reshape(-1)


def _ramp(deg):
    """concatenate(arange(d) for d in deg)"""
    deg = np.asarray(deg, dtype=np.int64)
    total = int(deg.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    return np.arange(total, dtype=np.int64) - np.repeat(np.cumsum(deg) - deg, deg)
