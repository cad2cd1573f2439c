"""Mesh parts and the binding of the C-ABI halo (include/c8.h, "multi-part meshes").

One process per GPU, one mesh part per process.  Elements are not ghosted; nodes on part boundaries are shared; every
rank assembles its own elements into GHOST-distributed A and b (all local nodes) and the ghost rows are then ADDed into
their owners -- the reference's MPI scheme.  The exchanges themselves live in libc8.so (csrc/c8_halo.hip: HIP pack /
unpack kernels, RCCL point-to-point messages or a host transport):

  C1  Halo.gather(ls, B)   LinearAlg::gather_b   linear_alg.cpp:78-86   ghost residual rows -> owner, ADD
  C2  Halo.gather(ls, A)   LinearAlg::gather_A   linear_alg.cpp:53-63   ghost Jacobian rows -> owner, ADD
  C3  Halo.scatter_x       apf::synchronize in Disc::add_to_soln, disc.cpp:944-947   owner -> copies, COPY
  C4  Comm.allreduce       PCU_Add_Doubles(grad) adjoint_objective.cpp:109; PCU_Add_Double(J) :39,:99;
  C5                       PCU_Add_Int(status)   primal.cpp:100,164 -- packed into one buffer by the caller

What stays here is what the reference's mesh database does at load time (disc.cpp:31-39, :316-332): cutting parts, and
the one-off exchange of the column lists of ghost rows from which the phantom nodes and the exchange lists follow
(numpy, vectorised; `dist` = torch.distributed, used for this set-up traffic and as the host transport of the tests).

Layout.  Local node numbering of a part: OWNED nodes, then GHOST nodes (touched by a local element, owned elsewhere),
then PHANTOM nodes (not touched locally; they only appear as columns of owned interface rows), each group sorted by
global id.  The phantom columns are reserved in the local graphs at c8_create (c8_mesh_desc.extra_pairs), so the first
`nowned` node rows of the local arrays already have the union pattern the reference builds in compute_owned_graph
(disc.cpp:389-398): the OWNED matrix / vector are PREFIX VIEWS of the local arrays and the halo ADD lands in place.
Owner of a shared node = lowest part id (PUMI's rule is not visible in the reference; any deterministic rule is
equivalent).
"""
import ctypes as C

import numpy as np

from . import lib as _l

NEQ = (3, 1)


class Part:
    """One rank's mesh part before the phantom columns are known."""

    def __init__(self, rank, world, conn, node_gid, node_owner, coords_of, num_global_nodes, owner_of):
        self.rank, self.world = rank, world
        self.conn = np.ascontiguousarray(conn, dtype=np.int32)
        self.node_gid = np.ascontiguousarray(node_gid, dtype=np.int64)       # owned, then ghosts
        self.node_owner = np.ascontiguousarray(node_owner, dtype=np.int32)
        self.coords_of = coords_of
        self.owner_of = owner_of                                              # gids -> owner rank (any node of the mesh)
        self.num_global_nodes = int(num_global_nodes)
        self.nowned = int((self.node_owner == rank).sum())
        assert (self.node_owner[:self.nowned] == rank).all(), "owned nodes must come first"
        self.ntouched = len(self.node_gid)


def _finish_part(rank, world, coords_of, gids, owner_of, conn_g, nglobal):
    owner = owner_of(gids)
    node_gid = np.concatenate([np.sort(gids[owner == rank]), np.sort(gids[owner != rank])])
    order = np.argsort(node_gid)
    conn = order[np.searchsorted(node_gid[order], conn_g)].astype(np.int32)
    return Part(rank, world, conn, node_gid, owner_of(node_gid), coords_of, nglobal, owner_of)


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
    if np.isscalar(n):
        n = (int(n),) * 3
    nx, ny, nz = (int(v) for v in n)
    px, py, pz = pdims
    world = px * py * pz
    bx, by, bz = rank % px, (rank // px) % py, rank // (px * py)
    NX, NY, NZ = px * nx + 1, py * ny + 1, pz * nz + 1
    i = np.arange(nx + 1) + bx * nx
    j = np.arange(ny + 1) + by * ny
    k = np.arange(nz + 1) + bz * nz
    K, J, I = np.meshgrid(k, j, i, indexing="ij")
    g = (K * NY + J) * NX + I
    h = edge / nx

    def coords_of(gid):
        gi, gj, gk = gid % NX, (gid // NX) % NY, gid // (NX * NY)
        return np.stack([gi * h, gj * h, gk * h], axis=1).astype(np.float64)

    def owner_of(gid):
        gi, gj, gk = gid % NX, (gid // NX) % NY, gid // (NX * NY)
        # blocks containing a node: the one it lies in and, on a lower face, the previous one;
        # rank is monotone in each block coordinate, so the lowest sharer takes every "previous"
        ox = np.minimum(gi // nx, px - 1) - ((gi % nx == 0) & (gi > 0) & (gi < px * nx)).astype(np.int64)
        oy = np.minimum(gj // ny, py - 1) - ((gj % ny == 0) & (gj > 0) & (gj < py * ny)).astype(np.int64)
        oz = np.minimum(gk // nz, pz - 1) - ((gk % nz == 0) & (gk > 0) & (gk < pz * nz)).astype(np.int64)
        return ((oz * py + oy) * px + ox).astype(np.int32)

    kk, jj, ii = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
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


def _ramp(deg):
    """concatenate(arange(d) for d in deg)"""
    deg = np.asarray(deg, dtype=np.int64)
    total = int(deg.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    return np.arange(total, dtype=np.int64) - np.repeat(np.cumsum(deg) - deg, deg)


def _exchange_objects(dist, world, rank, msgs):
    """msgs[r] goes to rank r; returns what every rank sent to this one (set-up traffic only)."""
    if world == 1 or dist is None:
        return [msgs[0]]
    gathered = [None] * world
    dist.all_gather_object(gathered, msgs)
    return [gathered[src][rank] for src in range(world)]


class HaloPlan:
    """The exchange lists of one part (c8_halo_desc) and the phantom columns (c8_mesh_desc.extra_pairs), from one
    exchange of the column lists of the ghost rows and one of the import requests.  `dist` is an initialised
    torch.distributed module, or None for a single part."""

    def __init__(self, part, dist=None):
        self.part, self.dist = part, dist
        p = part
        W = self.world = p.world if dist is not None else 1
        nn = p.conn.shape[1]
        ghost_lo, nt = p.nowned, p.ntouched
        touching = np.nonzero((p.conn >= ghost_lo).any(axis=1))[0]
        # elements that add into ghost rows: assembled first when the exchange overlaps the interior assembly
        self.interface_elems = touching.astype(np.int32)
        mask = np.ones(len(p.conn), dtype=bool)
        mask[touching] = False
        self.interior_elems = np.nonzero(mask)[0].astype(np.int32)
        # (row, col) couplings of the ghost rows, sorted by (row, col) local id = the order of the graph rows
        rows = np.repeat(p.conn[touching], nn, axis=1).reshape(-1)
        cols = np.tile(p.conn[touching], (1, nn)).reshape(-1)
        keep = rows >= ghost_lo
        key = np.unique(rows[keep].astype(np.int64) * (nt + 1) + cols[keep])
        g_rows, g_cols = key // (nt + 1), key % (nt + 1)
        ghosts = np.arange(p.nowned, nt)
        start = np.searchsorted(g_rows, ghosts)
        end = np.searchsorted(g_rows, ghosts, side="right")
        owners = p.node_owner[ghosts]
        # ---- export lists: my ghost rows grouped by owner; each owner learns the rows' column lists (global ids) ----
        order = np.argsort(owners, kind="stable")
        self.send_nodes = ghosts[order].astype(np.int32)
        self.send_ptr = np.searchsorted(owners[order], np.arange(W + 1)).astype(np.int64)
        msgs = []
        for r in range(W):
            sel = order[self.send_ptr[r]:self.send_ptr[r + 1]]
            deg = end[sel] - start[sel]
            idx = np.repeat(start[sel], deg) + _ramp(deg)
            msgs.append((p.node_gid[ghosts[sel]], np.concatenate([[0], np.cumsum(deg)]).astype(np.int64), p.node_gid[g_cols[idx]]))
        incoming = _exchange_objects(dist, W, p.rank, msgs)
        # ---- phantom nodes: received column gids that are not local ----
        all_cols = np.concatenate([m[2] for m in incoming]) if incoming else np.zeros(0, dtype=np.int64)
        self.phantom_gid = np.setdiff1d(np.unique(all_cols), p.node_gid).astype(np.int64)
        self.node_gid = np.concatenate([p.node_gid, self.phantom_gid])
        self.nnodes = len(self.node_gid)
        self.coords = p.coords_of(self.node_gid)
        self._sorted = np.argsort(self.node_gid)
        self._sorted_gid = self.node_gid[self._sorted]
        # ---- receive lists in local ids, and the extra graph pairs (duplicates of local couplings are harmless) ----
        rn, rcp, rc, pairs, rptr, ncols = [], [np.zeros(1, dtype=np.int64)], [], [], [0], 0
        for src in range(W):
            row_gids, col_ptr, col_gids = incoming[src]
            rl, cl = self.local_of(row_gids), self.local_of(col_gids)
            rn.append(rl)
            rcp.append(col_ptr[1:] + ncols)
            ncols += len(cl)
            rc.append(cl)
            pairs.append(np.stack([np.repeat(rl, np.diff(col_ptr)), cl], axis=1))
            rptr.append(rptr[-1] + len(rl))
        self.recv_ptr = np.array(rptr, dtype=np.int64)
        self.recv_nodes = np.concatenate(rn).astype(np.int32)
        self.recv_col_ptr = np.concatenate(rcp).astype(np.int64)
        self.recv_cols = np.concatenate(rc).astype(np.int32)
        self.extra_pairs = np.concatenate(pairs).astype(np.int32) if pairs else np.zeros((0, 2), dtype=np.int32)
        # ---- import lists (C3): every ghost and phantom node from its owner; the owners learn who wants what ----
        copies = np.arange(p.nowned, self.nnodes)
        cown = np.concatenate([owners, p.owner_of(self.phantom_gid).astype(np.int32)]) if len(copies) else np.zeros(0, dtype=np.int32)
        order = np.argsort(cown, kind="stable")
        self.import_nodes = copies[order].astype(np.int32)
        self.import_ptr = np.searchsorted(cown[order], np.arange(W + 1)).astype(np.int64)
        wanted = _exchange_objects(dist, W, p.rank, [self.node_gid[self.import_nodes[self.import_ptr[r]:self.import_ptr[r + 1]]] for r in range(W)])
        self.export_nodes = np.concatenate([self.local_of(w) for w in wanted]).astype(np.int32)
        self.export_ptr = np.concatenate([[0], np.cumsum([len(w) for w in wanted])]).astype(np.int64)
        assert (self.export_nodes < p.nowned).all(), "a rank asked for a node this rank does not own"

    def local_of(self, gids):
        """local ids of global node ids (all must be local: owned, ghost or phantom)"""
        gids = np.asarray(gids, dtype=np.int64)
        if len(gids) == 0:
            return np.zeros(0, dtype=np.int64)
        k = np.searchsorted(self._sorted_gid, gids)
        assert (k < len(self._sorted_gid)).all() and (self._sorted_gid[np.minimum(k, len(self._sorted_gid) - 1)] == gids).all()
        return self._sorted[k].astype(np.int64)

    def desc(self, ndims=3, nres=2):
        """c8_halo_desc over this plan's arrays (which it keeps alive), for systems with `ndims` equations per node in
        residual 0 and `nres` residuals"""
        a = lambda v, t: np.ascontiguousarray(v, dtype=t)
        keep = [a(self.send_ptr, np.int64), a(self.send_nodes, np.int32), a(self.recv_ptr, np.int64), a(self.recv_nodes, np.int32),
                a(self.recv_col_ptr, np.int64), a(self.recv_cols, np.int32), a(self.import_ptr, np.int64), a(self.import_nodes, np.int32),
                a(self.export_ptr, np.int64), a(self.export_nodes, np.int32)]
        p64 = lambda v: v.ctypes.data_as(_l.i64p)
        p32 = lambda v: v.ctypes.data_as(_l.i32p)
        d = _l.HaloDesc(self.part.nowned, self.part.ntouched, p64(keep[0]), p32(keep[1]), p64(keep[2]), p32(keep[3]), p64(keep[4]),
                        p32(keep[5]), p64(keep[6]), p32(keep[7]), p64(keep[8]), p32(keep[9]), int(ndims), int(nres))
        d._keep = keep
        return d


class Comm:
    """c8_comm: the communicator of the exchanges and small reductions.  Comm.rccl(dist) -- RCCL over xGMI, one rank
    per GPU (dist only carries the 128-byte id from rank 0 to the others); Comm.host(dist) -- host transport through
    torch.distributed on CPU tensors (gloo): the CPU-rendezvous tests and several ranks sharing one card."""

    def __init__(self, handle, rank, world, keep=()):
        self.L = _l.load_library()
        self.h, self.rank, self.world, self._keep = handle, rank, world, keep

    @classmethod
    def rccl(cls, dist, rank, world):
        L = _l.load_library()
        ident = C.create_string_buffer(_l.C8_COMM_ID_BYTES)
        if rank == 0:
            _l.check(L.c8_comm_rccl_id(ident))
        box = [bytes(ident.raw)]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        h = C.c_void_p()
        _l.check(L.c8_comm_create_rccl(C.create_string_buffer(box[0], _l.C8_COMM_ID_BYTES), rank, world, C.byref(h)))
        return cls(h, rank, world)

    @classmethod
    def host(cls, dist, rank, world, group=None):
        import torch
        L = _l.load_library()

        def exchange(_user, send, send_counts, recv, recv_counts):
            try:
                sc = [int(send_counts[r]) for r in range(world)]
                rc = [int(recv_counts[r]) for r in range(world)]
                s = torch.from_numpy(np.ctypeslib.as_array(send, shape=(max(1, sum(sc)),)))[:sum(sc)]
                r = torch.from_numpy(np.ctypeslib.as_array(recv, shape=(max(1, sum(rc)),)))[:sum(rc)]
                if world > 1:
                    dist.all_to_all_single(r, s, rc, sc, group=group)
                else:
                    r.copy_(s)
                return 0
            except Exception as e:  # an exception must not cross the C boundary
                print("c8 host exchange failed:", e)
                return 1

        def allreduce(_user, vals, n):
            try:
                if world > 1:
                    t = torch.from_numpy(np.ctypeslib.as_array(vals, shape=(n,)))
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                return 0
            except Exception as e:
                print("c8 host all-reduce failed:", e)
                return 1

        ex, ar = _l.HOST_EXCHANGE_FN(exchange), _l.HOST_ALLREDUCE_FN(allreduce)
        h = C.c_void_p()
        _l.check(L.c8_comm_create_host(rank, world, ex, ar, None, C.byref(h)))
        return cls(h, rank, world, keep=(ex, ar))

    def allreduce(self, values):
        """C4 / C5: in-place SUM over the ranks of a float64 numpy array (gradient, objective, failure flag packed)."""
        v = np.ascontiguousarray(values, dtype=np.float64)
        _l.check(self.L.c8_comm_allreduce_sum(self.h, v.ctypes.data_as(_l.dp), v.size))
        if v is not values:
            values[...] = v
        return values

    def close(self):
        if self.h:
            self.L.c8_comm_destroy(self.h)
            self.h = None


class Halo:
    """c8_halo: index tables built from a HaloPlan and the part's node graph (block (1,1) of the assembler's graph),
    attached to an Assembler and a Comm for the run-time exchanges."""
    B, A = _l.C8_HALO_B, _l.C8_HALO_A

    def __init__(self, plan, node_rowptr, node_colidx, asm=None, comm=None, ndims=None, nres=None):
        self.L = _l.load_library()
        if ndims is None:  # the shape of the assembler's systems: 3 + 1 equations per node unless told otherwise
            ndims = getattr(asm, "ndims", 3)
        if nres is None:
            nres = getattr(asm, "nres", 2)
        self.plan, self.world = plan, plan.world
        self.nowned = plan.part.nowned
        rp = np.ascontiguousarray(node_rowptr, dtype=np.int64)
        ci = np.ascontiguousarray(node_colidx, dtype=np.int32)
        assert len(rp) == plan.nnodes + 1
        d = plan.desc(ndims, nres)
        h = C.c_void_p()
        _l.check(self.L.c8_halo_build(plan.nnodes, rp.ctypes.data_as(_l.i64p), ci.ctypes.data_as(_l.i32p), C.byref(d),
                                      plan.part.rank if plan.world > 1 else 0, plan.world, C.byref(h)))
        self.h, self.asm, self.comm = h, None, None
        if asm is not None:
            self.attach(asm, comm)

    def attach(self, asm, comm):
        _l.check(self.L.c8_halo_attach(self.h, asm.h, comm.h))
        self.asm, self.comm = asm, comm

    def close(self):
        if self.h:
            self.L.c8_halo_destroy(self.h)
            self.h = None

    def table(self, which):
        """host copy of one index table (c8_halo_table)"""
        n, ptr = C.c_int64(), _l.i64p()
        _l.check(self.L.c8_halo_table(self.h, which, C.byref(n), C.byref(ptr)))
        return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy() if n.value else np.zeros(0, dtype=np.int64)

    def send_bytes(self, what):
        return int(self.L.c8_halo_send_bytes(self.h, what))

    def gather_start(self, ls, what=3):
        """Pack the ghost rows (what = Halo.A | Halo.B) on the assembler's stream and start the exchange.  Everything
        that adds into ghost rows must have been enqueued before; work enqueued after this call (the interior elements,
        the owned rows' sums) runs while the messages travel."""
        sy = ls.c_struct()
        _l.check(self.L.c8_halo_gather_start(self.h, C.byref(sy), what))

    def gather_finish(self, ls):
        """ADD what arrived into the owned rows (ascending source rank: reproducible)."""
        sy = ls.c_struct()
        _l.check(self.L.c8_halo_gather_finish(self.h, C.byref(sy)))

    def gather(self, ls, what=3):
        """C1 / C2, in place: afterwards the rows of the first `nowned` nodes hold the OWNED system (union pattern,
        columns in local numbering; local->global ids = plan.node_gid)."""
        self.gather_start(ls, what)
        self.gather_finish(ls)
        return ls

    def scatter_x(self, x):
        """C3, in place: owner values of a nodal field pair x = [u, p] copied to the ghost and phantom copies."""
        xs = (C.c_void_p * 2)(x[0].data_ptr(), x[1].data_ptr())
        _l.check(self.L.c8_halo_scatter_x(self.h, xs))
        return x
