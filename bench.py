"""bench.py -- element Jacobian assemblies / second on synthetic hex8 J2-plasticity bricks.

One "step" = one pass of the hot path (c8_assemble_forward_jacobian: residual + Jacobian with the
local return-mapping solves and the CSR scatter) over the rank's mesh part, inputs resident in HBM.
Workload (BASELINE.json metric: "1M hex8 J2-plasticity fp64, 1/2/4/8 GPUs"): a 100x100x100 hex8 brick,
small_J2 (E 1000, nu 0.25, K 100, Y 2), prescribed mixed elastic/plastic state of SURVEY.md 8d.

N = 1: the whole brick on one GPU.
N > 1 (one line, both numbers): `value` is the STRONG-scaling rate -- the one 100^3 brick split N ways (BASELINE.json
config 4: 2x1x1, 2x2x1, 2x2x2 blocks) -- and `also.weak` holds the weak-scaling rate -- one 100^3 block per GPU of a
(px*100, py*100, pz*100) brick (config 5's shape).  A step is then the assembly plus the owned/ghost halo ADD of the
Jacobian and residual (LinearAlg::gather_A/gather_b): HIP pack / unpack kernels and grouped RCCL point-to-point messages
of libc8.so (c8_halo_*), no torch op on the path.  Both are reported kernel-only (HIP events around the assembly
launches) and as the whole step (kernel + C1 + C2), whole job = all elements of all ranks / max-over-ranks wall time.
The exchange overlaps the assembly: the ghost rows are formed first and travel while the owned rows are formed (with
--scatter atomic: the elements that add into ghost rows are assembled first).  Every rank's owned rows are checked
against a single-part assembly of a sampled sub-brick around a corner it shares with its neighbours.  torch.distributed
(gloo) only carries the rendezvous, the one-off exchange lists and the timing reduction.

Launch: python bench.py [--gpus N --steps K --warmup W]; for N > 1 under torch.distributed.run.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

J2 = [1000.0, 0.25, 100.0, 2.0, 0.0, 0.0]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy)


def algorithmic_bytes(nelems, nnodes, nnz_total, nen=8, nqp=8, nloc=7, ndofn=4):
    """every datum once. connectivity + xi_prev read + xi write; coords + x + x_prev read
    + R write; each CSR value written once."""
    return nelems * (4 * nen + 8 * nqp * nloc * 2) + nnodes * 8 * (3 + 2 * ndofn + ndofn) + 8 * nnz_total


def pmc_profile(edge, scatter, kernel, build_id):
    """The committed rocprofv3 --pmc passes of this same command (profiles/*traffic*.json, collected by
    tools/collect_pmc.py as MI355X_MICROARCH.md prescribes; PMC counters cannot be read from inside the timed process).
    Returns (profile of THIS library build or None, newest profile of another build or None): a figure measured on
    other kernels is reported apart, never as `traffic`."""
    import glob
    same = other = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        c = d.get("config", {})
        if c.get("edge") == edge and c.get("scatter") == scatter and c.get("kernel") == kernel:
            d["file"] = os.path.basename(f)
            if d.get("build_id") == build_id:
                same = d
            else:
                other = d
    return same, other


def valu_utilisation(profile, kernel_ms):
    """FP64-VALU issue utilisation of the assembly (SURVEY.md 8d asks for it beside the HBM figure: it is what binds):
    wave-level VALU instructions of the assembly's kernels per launch / their measured duration, against the issue peak
    of 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 FP64 instruction."""
    peak = 256 * 4 * 2.4e9 / 4.0
    insts = sum(profile["per_launch_mean"][k].get("SQ_INSTS_VALU", 0.0) for k in profile.get("launches_per_assembly", [])
                if k in profile["per_launch_mean"])
    if not insts:
        return None
    rate = insts / (kernel_ms * 1e-3)
    return {"wave_valu_insts": insts, "issue_peak_per_s": peak, "achieved_per_s": rate, "frac": rate / peak}


def cpu_baseline(n, nthreads):
    """Oracle (CPU restatement of the reference algorithm) timed on an n^3 sample of the same workload."""
    import oracle_lib as ol
    from meshes import brick, prescribed_fields
    c, conn, _ = brick(n, n, n)
    orc = ol.Oracle(ol.HEX8, c, conn, "small_J2", J2)
    u, p = prescribed_fields(c, 0.004, ramp=True)
    z, zp = np.zeros_like(u), np.zeros_like(p)
    ls, xi = orc.new_linsys(), orc.new_state()
    t0 = time.perf_counter()
    rc = orc.forward_jacobian(u, p, z, zp, orc.new_state(), xi, ls, nthreads=nthreads)
    dt = time.perf_counter() - t0
    assert rc == 0
    return len(conn) / dt, orc, (u, p, z, zp), ls, xi


def consistent_fields(X, extent, eps=0.004, E=1000.0, nu=0.25):
    """The ramped uniaxial state of SURVEY.md 8d as a function of the GLOBAL coordinates alone (the perturbation is a
    sine instead of a random number), so that every part and a single-part assembly of any sub-brick see the same field"""
    x, y, z = X[:, 0], X[:, 1], X[:, 2]
    ly = extent[1]
    e = eps * y / ly
    wob = np.sin(37.0 * x + 23.0 * y + 31.0 * z)
    u = np.zeros_like(X)
    u[:, 1] = 0.5 * eps * y * y / ly
    u[:, 0] = -nu * e * x
    u[:, 2] = -nu * e * z
    u += 2e-4 * eps * np.stack([wob, np.cos(29.0 * x - 11.0 * y + 17.0 * z), np.sin(19.0 * x + 41.0 * y - 13.0 * z)], axis=1)
    kappa = E / (3 * (1 - 2 * nu))
    p = -kappa * (1 - 2 * nu) * e * (1.0 + 1e-2 * wob)
    return np.ascontiguousarray(u.ravel()), np.ascontiguousarray(p)


class Case:
    """One mesh configuration of the job: this rank's part, its context, halo tables and prescribed state, and the step."""

    def __init__(self, args, scaling, world, rank, dist, dev, comm):
        import torch
        from calibr8_amd import Assembler
        from calibr8_amd import distributed as D
        from meshes import prescribed_fields
        self.args, self.scaling, self.world, self.rank, self.dist, self.dev = args, scaling, world, rank, dist, dev
        self.torch = torch
        n = args.n
        pdims = D.pdims_for(world)
        if scaling == "strong":
            assert all(n % q == 0 for q in pdims), "strong scaling: the edge must divide by the part grid"
            block = tuple(n // q for q in pdims)
        else:
            block = (n, n, n)
        self.elem_edge = 1.0 / n  # element edge 1/n everywhere
        if args.workload == "notch":
            # about n^3 elements per GPU: a (4 : 1 : 1) bar, 15 % of whose elements the two notches remove; slabs along the bar
            from meshes import notched_bar
            assert scaling == "weak", "--workload notch is weak-scaled"
            ny = max(4, int(round(0.64 * n)))
            nx = max(8, int(round(world * n ** 3 / (0.85 * ny * ny))))
            gc, gconn, _ = notched_bar(nx, ny, ny)
            order = np.argsort(gc[gconn].mean(axis=1)[:, 0], kind="stable")
            elem_part = np.empty(len(gconn), dtype=np.int32)
            elem_part[order] = (np.arange(len(gconn)) * world) // len(gconn)
            part = D.part_from_global(gc, gconn, elem_part, rank, world)
            block, pdims = (nx, ny, ny), (world, 1, 1)
            del gc, gconn, order, elem_part
        else:
            part = D.brick_part(rank, pdims, block, edge=block[0] / n)
        self.block, self.pdims, self.part = block, pdims, part
        self.plan = plan = D.HaloPlan(part, dist if world > 1 else None)
        coords = plan.coords
        self.asm = asm = Assembler(8, coords, part.conn, "small_J2", J2, device=str(dev), scatter=args.scatter,
                                   extra_pairs=plan.extra_pairs)
        asm.set_kernel(args.kernel)
        if args.stage_chunk > 0:
            asm.set_stage_chunk(args.stage_chunk)
        if args.stage_overlap >= 0:
            asm.set_stage_overlap(args.stage_overlap)
        self.halo = D.Halo(plan, asm.rowptr[1][1], asm.colidx[1][1], asm, comm) if world > 1 else None
        # prescribed state of SURVEY.md 8d on the rank's own block (local coordinates of the block)
        origin = coords[: part.ntouched].min(axis=0)
        u_h, p_h = prescribed_fields(coords - origin, 0.004, ramp=True, seed=1234 + rank)
        self.u, self.p = asm.dev(u_h), asm.dev(p_h)
        self.u0, self.p0 = torch.zeros_like(self.u), torch.zeros_like(self.p)
        self.xi_prev, self.xi = asm.new_state(), asm.new_state()
        self.ls = asm.new_linsys()
        asm.set_async(True)
        self.overlap = world > 1 and args.scatter == "atomic" and not args.no_overlap
        self.split = world > 1 and args.scatter == "gather" and not args.no_overlap
        if self.overlap:
            self.e_if = torch.as_tensor(plan.interface_elems, device=dev)
            self.e_in = torch.as_tensor(plan.interior_elems, device=dev)
        if args.assign:
            asm.set_assign_mode(True)
        if self.split:  # assembly in two parts: the ghost rows (local nodes nowned .. ntouched) first
            if args.stage_chunk <= 0:
                asm.set_stage_chunk(asm.nelems)  # (staged kernels: the two-part row sums need the whole part in one chunk)
            asm.set_gather_early_nodes(part.nowned, part.ntouched)

    def step(self, ev=None, fields=None):
        """eval_forward_jacobian, then la->gather_A / gather_b (primal.cpp:99,110-111).  With more than one rank the
        exchange of the ghost rows (one message per neighbour) runs beside assembly work; `ev` = HIP-event pairs around
        the assembly launches."""
        asm, ls, halo = self.asm, self.ls, self.halo
        u, p = fields if fields else (self.u, self.p)
        u0, p0, xi_prev, xi = self.u0, self.p0, self.xi_prev, self.xi
        if self.split:  # ghost rows first; the other rows are formed while the ghost rows travel
            if ev:
                ev[0][0].record()
            asm.forward_jacobian(u, p, u0, p0, xi_prev, xi, ls)
            if ev:
                ev[0][1].record()
            halo.gather_start(ls)
            if ev:
                ev[1][0].record()
            asm.gather_finish()
            if ev:
                ev[1][1].record()
            halo.gather_finish(ls)
            return
        if not self.overlap:
            if ev:
                ev[0][0].record()
            asm.forward_jacobian(u, p, u0, p0, xi_prev, xi, ls)
            if ev:
                ev[0][1].record()
            if self.world > 1:
                halo.gather(ls)
            return
        if ev:
            ev[0][0].record()
        asm.forward_jacobian_subset(u, p, u0, p0, xi_prev, xi, ls, self.e_if)
        if ev:
            ev[0][1].record()
        halo.gather_start(ls)
        if ev:
            ev[1][0].record()
        asm.forward_jacobian_subset(u, p, u0, p0, xi_prev, xi, ls, self.e_in)
        if ev:
            ev[1][1].record()
        halo.gather_finish(ls)

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def allmax(self, v):
        t = self.torch.tensor([v], dtype=self.torch.float64)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def allsum(self, v):
        t = self.torch.tensor([float(v)], dtype=self.torch.float64)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def run(self, steps, warmup):
        """W untimed steps, then EXACTLY K steps between barrier + synchronize on both sides; max over the ranks."""
        torch = self.torch
        for _ in range(warmup):
            self.step()
        assert self.asm.status() == 0
        E = lambda: torch.cuda.Event(enable_timing=True)
        ev = [[(E(), E()), (E(), E())] for _ in range(steps)]  # HIP events on the stream the kernels are launched on
        self.barrier()
        t0 = time.perf_counter()
        for pairs in ev:
            self.step(pairs)
        self.barrier()
        dt = time.perf_counter() - t0
        assert self.asm.status() == 0
        npairs = 2 if (self.overlap or self.split) else 1
        per_step = np.array([sum(a.elapsed_time(b) for a, b in pairs[:npairs]) for pairs in ev])  # assembly launches only
        dt = self.allmax(dt)
        nelems_total = int(self.allsum(self.asm.nelems))
        kmean = self.allmax(float(per_step.mean()))
        return {"dt": dt, "ms_per_step": 1e3 * dt / steps, "value": nelems_total * steps / dt, "nelems_total": nelems_total,
                "kernel_ms_per_step": kmean,
                "kernel_ms_median_min_max": [self.allmax(float(np.median(per_step))), -self.allmax(-float(per_step.min())),
                                             self.allmax(float(per_step.max()))],
                "elements_per_s_kernel_only": nelems_total / (1e-3 * kmean)}

    def timed(self, fn, reps):
        fn()  # untimed: a mode's first call may allocate (the element stage of the staged kernels)
        self.barrier()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        self.barrier()
        return 1e3 * self.allmax(time.perf_counter() - t) / reps

    def overlap_check(self):
        """untimed: the overlapped step against the blocking exchange after a whole assembly (owned rows)"""
        if not (self.overlap or self.split):
            return None
        asm, ls, part = self.asm, self.ls, self.part
        ls.zero()
        self.step()
        ref_flat = ls.flat.clone()
        ls.zero()
        if self.split:
            asm.set_gather_early_nodes(0, 0)
        asm.forward_jacobian(self.u, self.p, self.u0, self.p0, self.xi_prev, self.xi, ls)
        self.halo.gather(ls)
        if self.split:
            asm.set_gather_early_nodes(part.nowned, part.ntouched)
        no = part.nowned  # owned rows only: ghost rows are scratch after the exchange
        d = 0.0
        for k, neq in ((4, 3), (5, 1)):
            lo = int(ls.offsets[k])
            d = max(d, float((ls.flat[lo:lo + no * neq] - ref_flat[lo:lo + no * neq]).abs().max() /
                             ls.flat[lo:lo + no * neq].abs().max()))
        lo, hi = 0, int(asm.rowptr[0][0][no * 3])
        d = max(d, float((ls.flat[lo:hi] - ref_flat[lo:hi]).abs().max() / ls.flat[lo:hi].abs().max()))
        d = self.allmax(d)
        assert d < 1e-12, d
        return d

    def owned_rows_check(self, half=3):
        """untimed: this rank's OWNED rows after assembly + halo ADD against a single-part assembly of a sub-brick of the
        global mesh (2*half elements per direction, around a corner this block shares with its neighbours), with a state
        that is a function of the global coordinates.  Returns (max relative difference over the sampled rows, all ranks;
        rows compared, all ranks)."""
        from calibr8_amd import Assembler
        from meshes import brick
        asm, plan, part = self.asm, self.plan, self.part
        torch = self.torch
        px, py, pz = self.pdims
        nb = self.block
        NE = (px * nb[0], py * nb[1], pz * nb[2])  # elements of the global brick per direction
        h = self.elem_edge
        extent = tuple(v * h for v in NE)
        bidx = (self.rank % px, (self.rank // px) % py, self.rank // (px * py))
        # the state, on every local node (ghost and phantom copies included: the same function everywhere)
        u_h, p_h = consistent_fields(plan.coords, extent)
        fields = (asm.dev(u_h), asm.dev(p_h))
        self.ls.zero()
        self.step(fields=fields)
        self.barrier()
        # the sub-brick: around the block corner towards the next block in every direction that has one
        lo, hi = [], []
        for d in range(3):
            c = (bidx[d] + 1) * nb[d] if bidx[d] + 1 < self.pdims[d] else bidx[d] * nb[d]
            lo.append(max(0, c - half))
            hi.append(min(NE[d], c + half))
        m = [hi[d] - lo[d] for d in range(3)]
        sc, sconn, _ = brick(m[0], m[1], m[2], m[0] * h, m[1] * h, m[2] * h)
        sc = sc + np.array([lo[0] * h, lo[1] * h, lo[2] * h])
        sub = Assembler(8, sc, sconn, "small_J2", J2, device=str(self.dev), scatter=self.args.scatter)
        sub.set_kernel(self.args.kernel)
        su, sp = consistent_fields(sc, extent)
        sls = sub.new_linsys()
        du, dp = sub.dev(su), sub.dev(sp)
        assert sub.forward_jacobian(du, dp, torch.zeros_like(du), torch.zeros_like(dp), sub.new_state(), sub.new_state(), sls) == 0
        # global ids of the sub-brick's nodes; the complete ones (every element around them is inside the sub-brick)
        NX, NY = NE[0] + 1, NE[1] + 1
        ii, jj, kk = np.meshgrid(np.arange(m[0] + 1), np.arange(m[1] + 1), np.arange(m[2] + 1), indexing="ij")
        sid = (kk * (m[1] + 1) + jj) * (m[0] + 1) + ii
        gi, gj, gk = ii + lo[0], jj + lo[1], kk + lo[2]
        gid_of_sub = np.empty(sid.size, dtype=np.int64)
        gid_of_sub[sid.ravel()] = ((gk * NY + gj) * NX + gi).ravel()
        inner = np.ones(ii.shape, dtype=bool)
        for d, (g, l, hgh) in enumerate(((gi, lo[0], hi[0]), (gj, lo[1], hi[1]), (gk, lo[2], hi[2]))):
            inner &= ((g > l) | (g == 0)) & ((g < hgh) | (g == NE[d]))
        cand = sid[inner].ravel()
        owned_sorted = part.node_gid[:part.nowned]  # sorted by construction
        pos = np.searchsorted(owned_sorted, gid_of_sub[cand])
        ok = (pos < len(owned_sorted)) & (owned_sorted[np.minimum(pos, len(owned_sorted) - 1)] == gid_of_sub[cand])
        sub_nodes, loc_nodes = cand[ok], pos[ok]
        A_loc = [[self.ls.A[i][j].cpu().numpy() for j in range(2)] for i in range(2)]
        A_sub = [[sls.A[i][j].cpu().numpy() for j in range(2)] for i in range(2)]
        b_loc, b_sub = [v.cpu().numpy() for v in self.ls.b], [v.cpu().numpy() for v in sls.b]
        neq = (3, 1)
        worst, nrows = 0.0, 0
        for sn, ln in zip(sub_nodes, loc_nodes):
            for i in range(2):
                for a in range(neq[i]):
                    rs, rl = sn * neq[i] + a, ln * neq[i] + a
                    for j in range(2):
                        r0, r1 = sub.rowptr[i][j][rs], sub.rowptr[i][j][rs + 1]
                        cs = sub.colidx[i][j][r0:r1]
                        ref = dict(zip((gid_of_sub[cs // neq[j]] * neq[j] + cs % neq[j]).tolist(), A_sub[i][j][r0:r1].tolist()))
                        q0, q1 = asm.rowptr[i][j][rl], asm.rowptr[i][j][rl + 1]
                        cl = asm.colidx[i][j][q0:q1]
                        got = dict(zip((plan.node_gid[cl // neq[j]] * neq[j] + cl % neq[j]).tolist(), A_loc[i][j][q0:q1].tolist()))
                        rowmax = max(max(abs(v) for v in ref.values()), 1e-300)
                        for key, v in ref.items():
                            worst = max(worst, abs(got.get(key, float("nan")) - v) / rowmax)
                        for key, v in got.items():
                            if key not in ref:
                                worst = max(worst, abs(v) / rowmax)
                    worst = max(worst, abs(b_loc[i][rl] - b_sub[i][rs]) / max(np.abs(b_sub[i]).max(), 1e-300))
                    nrows += 1
        del sub, sls
        worst = float("inf") if worst != worst else worst  # a missing entry is a failure
        return self.allmax(worst), int(self.allsum(nrows))

    def close(self):
        if self.halo is not None:
            self.halo.close()
        self.halo = None


def main():
    # stdout carries ONE line, the JSON result: everything else that writes to file descriptor 1 during the run (gloo's
    # connection messages, library banners of the ranks) goes to stderr
    real_stdout = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--edge", dest="n", type=int, default=100, help="brick edge (elements): per GPU (weak) / of the whole brick (strong)")
    ap.add_argument("--scatter", default="gather", choices=["colored", "atomic", "gather"],
                    help="gather (default): rows owned by nodes (row-per-node kernel, or staged assembly + row sums), no atomics, bitwise reproducible, fastest")
    ap.add_argument("--cpu-sample", type=int, default=32, help="edge of the CPU-baseline sample brick")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = host cores available, max 16)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "slot", "wave", "wave_ad", "node"])
    ap.add_argument("--stage-overlap", type=int, default=-1, help="staged kernels: 1 = row sums of a chunk beside the assembly of the next (second stream), 0 = one after the other; default: the library's choice")
    ap.add_argument("--stage-chunk", type=int, default=0, help="staged kernels: minimum elements per staged chunk")
    ap.add_argument("--assign", action="store_true", help="scatter=gather: c8_set_assign_mode (zero_all + assembly in one call); "
                    "NOT the default: the contract metric is the accumulate-into assembly")
    ap.add_argument("--workload", default="brick", choices=["brick", "notch"],
                    help="brick (default: the contract workload); notch = BASELINE config 3's geometry, a double-edge-notched "
                         "hex8 bar of about edge^3 elements per GPU (tests/meshes.py notched_bar), cut into slabs along the bar for N > 1")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: exchange the ghost rows after the whole assembly")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="the timed steps only (no `also` timings, other entry points, CPU baseline): what tools/collect_pmc.py profiles")
    ap.add_argument("--scaling", default="both", choices=["both", "weak", "strong"],
                    help="N > 1: both (default) = strong scaling as `value` and weak scaling under `also.weak`; weak / strong = that one only")
    ap.add_argument("--transport", "--backend", dest="transport", default="rccl", choices=["rccl", "host", "nccl", "gloo"],
                    help="rccl (default): grouped ncclSend/ncclRecv over xGMI, one rank per GPU; host: rehearsal of the N > 1 "
                         "path with all ranks on one GPU (RCCL refuses that), messages through the host")
    ap.add_argument("--require-rccl", action="store_true", help="N > 1: exit non-zero instead of falling back to the host transport when the RCCL communicator fails")
    args = ap.parse_args()
    args.transport = {"nccl": "rccl", "gloo": "host"}.get(args.transport, args.transport)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node N for --gpus N"
    ndev = torch.cuda.device_count()
    fallback = None
    if world > ndev and args.transport == "rccl":
        # fewer cards than ranks (a rehearsal on a shared card): RCCL admits one rank per card
        if args.require_rccl:
            raise SystemExit("bench.py: %d ranks on %d GPU(s) and --require-rccl" % (world, ndev))
        print("rank %d: %d ranks on %d GPU(s): host transport" % (rank, world, ndev), file=sys.stderr)
        args.transport, fallback = "host", "%d ranks share %d GPU(s): RCCL admits one rank per card" % (world, ndev)
    if args.transport == "host":
        local_rank = local_rank % max(ndev, 1) if world > ndev else 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("gloo")  # control plane only: rendezvous, exchange lists, timing; the data path is libc8.so's

    from calibr8_amd import distributed as D
    from calibr8_amd import lib as c8lib

    transport = args.transport
    comm = None
    if world > 1:
        if transport == "rccl":
            err = ""
            try:
                comm = D.Comm.rccl(dist, rank, world)
                # pre-flight: one small all-reduce through the new communicator; a wrong sum or an error sends every rank
                # to the host transport below (said in `transport_fallback` and config.transport) or, with --require-rccl, out
                chk = comm.allreduce(np.array([1.0, float(rank)]))
                if abs(chk[0] - world) > 1e-12 or abs(chk[1] - 0.5 * world * (world - 1)) > 1e-9:
                    raise RuntimeError("all-reduce through the RCCL communicator returned %r" % (chk,))
                ok = 1.0
            except Exception as e:  # every rank must take the same branch
                print("rank %d: RCCL communicator failed (%s)" % (rank, e), file=sys.stderr)
                ok, err = 0.0, str(e)
            t = torch.tensor([ok])
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if float(t.item()) == 0.0:
                errs = [None] * world
                dist.all_gather_object(errs, err)
                why = next((e for e in errs if e), "unknown")
                if args.require_rccl:
                    raise SystemExit("bench.py: the RCCL communicator could not be created (%s) and --require-rccl" % why)
                if comm is not None:
                    comm.close()
                comm, transport, fallback = None, "host", "RCCL communicator could not be created: %s" % why[:300]
        if comm is None:
            comm = D.Comm.host(dist, rank, world)

    scalings = ["weak"] if world == 1 else (["strong", "weak"] if args.scaling == "both" else [args.scaling])
    if args.workload == "notch":
        scalings = ["weak"]
    head = Case(args, scalings[0], world, rank, dist, dev, comm)
    res = head.run(args.steps, args.warmup)
    asm = head.asm
    reps = max(2, min(5, args.steps))
    also = {}
    if not args.headline_only:
        # beside the headline (untimed above): the reference's step is zero_all + eval_forward_jacobian (primal.cpp:98-99)
        def step_with_zero():
            head.ls.zero()
            head.step()

        also["ms_per_step_with_zero_all"] = head.timed(step_with_zero, reps)
        if args.scatter == "gather" and not args.assign:
            asm.set_assign_mode(True)  # zero_all + assembly in one call (c8_set_assign_mode): no zeroing pass, no read of the old values
            also["ms_per_step_assign_mode"] = head.timed(head.step, reps)
            asm.set_assign_mode(False)
        if args.kernel in ("auto", "node") and args.scatter == "gather":
            # the staged one-wavefront-per-element form of the same closed form (round 2's default): stage + row sums
            asm.set_kernel("wave")
            also["ms_per_step_staged_closed_form"] = head.timed(head.step, reps)
            asm.set_kernel(args.kernel)
        if args.kernel in ("auto", "wave", "node"):
            # the same step with the local Newton iteration and the AD passes kept in the kernel (C8_KERNEL_WAVE_AD): what every
            # model without a closed form runs, and what round 1 timed
            asm.set_kernel("wave_ad")
            also["ms_per_step_iterated_ad_form"] = head.timed(head.step, reps)
            also["elements_per_s_iterated_ad_form"] = res["nelems_total"] / (1e-3 * also["ms_per_step_iterated_ad_form"])
            asm.set_kernel(args.kernel)
        assert asm.status() == 0
    overlap_check = head.overlap_check()
    rows_check = head.owned_rows_check() if (world > 1 and args.workload == "brick") else None
    if rows_check is not None:
        assert rows_check[0] < 1e-12, rows_check
    halo_bytes = int(head.allmax(float(head.halo.send_bytes(3)) if world > 1 else 0.0))
    plastic_frac = float((head.xi[:, :, 6] > 0).double().mean().item())
    build_info = c8lib.load_library().c8_build_info().decode()
    build_id = build_info.split()[0].split("=")[1]
    block, pdims = head.block, head.pdims
    node_path = args.kernel in ("auto", "node") and args.scatter == "gather"
    kkey = args.kernel if args.kernel in ("slot", "wave_ad") else ("node" if node_path else "wave")
    n = args.n

    def part_label(c):
        return "%dx%dx%d hex8 block per GPU (%d elements, %d nodes)" % (tuple(c.block) + (c.asm.nelems, c.asm.nnodes))

    out = {
        "metric": "element Jacobian assemblies/sec, 1M hex8 J2-plasticity fp64",
        "value": res["value"], "unit": "elements/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": scalings[0] if world > 1 else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("double-edge-notched hex8 bar (BASELINE config 3's geometry: the %dx%dx%d brick minus two V-notches; "
                                "%d elements, %d nodes on this GPU%s), small_J2 E1000 nu0.25 K100 Y2, prescribed ramped uniaxial state "
                                "eps=0.004 seed 1234, residual+Jacobian assembly"
                                % (block[0], block[1], block[2], asm.nelems, asm.nnodes, ", slabs along the bar" if world > 1 else ""))
                               if args.workload == "notch" else
                               "%s%s, small_J2 E1000 nu0.25 K100 Y2, prescribed ramped uniaxial state eps=0.004 seed 1234, "
                               "residual+Jacobian assembly; a structured brick stands in for the notched specimen "
                               "(SURVEY.md 8d; --workload notch times that one)"
                               % (part_label(head), " = the %d^3 brick (BASELINE config 4) split %dx%dx%d" % ((n,) + tuple(pdims))
                                  if scalings[0] == "strong" and world > 1 else ""),
                   "elements_per_gpu": asm.nelems, "elements_total": res["nelems_total"], "plastic_fraction": plastic_frac,
                   "scatter": args.scatter, "kernel": args.kernel,
                   "local_solve": "iterated: local Newton + forward-mode AD in the kernel (C8_KERNEL_WAVE_AD)" if args.kernel in ("slot", "wave_ad")
                                  else "closed form of small_J2 (radial return + consistent tangent, same state and Jacobian "
                                       "to 2e-13; the library's default); also.ms_per_step_iterated_ad_form times the "
                                       "Newton + AD form",
                   "shape_tables_cached": True, "shape_tables_bytes": int(asm.nelems * 208 * 8),
                   "shape_tables_note": "dN/dx, w dv and h of the static mesh are computed once per context (c8_set_shape_cache, k_shape_tables) "
                                        "outside the timed call; the reference recomputes them for every AD pass (weight.cpp:5-25)",
                   "colors": asm.ncolors, "part_grid": list(pdims),
                   "parallelism": "one element block per GPU; ghost rows of A and b ADDed into their owners: HIP pack kernel, "
                                  "one grouped ncclSend/ncclRecv message per neighbour (RCCL over xGMI), HIP unpack-add kernel "
                                  "(c8_halo_gather_start / _finish); block partition (ParMETIS is not in the image)",
                   "transport": transport if world > 1 else None,
                   "halo_send_bytes_per_step_max_rank": halo_bytes,
                   "halo_overlapped_with_interior_assembly": bool(head.overlap),
                   "halo_overlapped_with_owned_rows": bool(head.split), "assign_mode": bool(args.assign),
                   "overlap_vs_blocking_max_rel_diff": overlap_check,
                   "owned_rows_vs_single_part_sub_brick": None if rows_check is None else {"max_rel_diff": rows_check[0], "rows_compared_all_ranks": rows_check[1]},
                   "library_build": build_info},
        "kernel_only": {"ms_per_step": res["kernel_ms_per_step"], "elements_per_s": res["elements_per_s_kernel_only"],
                        "ms_median_min_max_over_steps": res["kernel_ms_median_min_max"],
                        "note": "HIP events around the assembly launches of each of the K timed steps (max over ranks of the per-rank "
                                "mean / median / max, min over ranks of the min); ms_per_step above is the whole step: kernel + C1 + C2 for N > 1"},
        "also": also,
    }
    if fallback:
        out["transport_fallback"] = fallback  # NOT an RCCL measurement: the messages went through the host
    nnz_total = sum(asm.nnz[i][j] for i in range(2) for j in range(2))
    head_nelems, head_nnodes = asm.nelems, asm.nnodes
    kernel_ms = res["kernel_ms_per_step"]

    # ---- the other scaling of the same job (N > 1): weak beside strong ------------------------------------------------
    if world > 1 and len(scalings) > 1:
        head.close()
        head.ls = head.xi = head.xi_prev = head.u = head.p = head.u0 = head.p0 = None
        torch.cuda.empty_cache()
        other = Case(args, scalings[1], world, rank, dist, dev, comm)
        r2 = other.run(args.steps, args.warmup)
        oc2 = other.overlap_check()
        rc2 = other.owned_rows_check() if args.workload == "brick" else None
        if rc2 is not None:
            assert rc2[0] < 1e-12, rc2
        out["also"][scalings[1]] = {
            "scaling": scalings[1], "value": r2["value"], "unit": "elements/s", "ms_per_step": r2["ms_per_step"],
            "kernel_only_ms_per_step": r2["kernel_ms_per_step"], "kernel_only_elements_per_s": r2["elements_per_s_kernel_only"],
            "kernel_ms_median_min_max_over_steps": r2["kernel_ms_median_min_max"],
            "workload": part_label(other) + (" of a %dx%dx%d-element brick" % tuple(other.block[d] * other.pdims[d] for d in range(3))),
            "elements_total": r2["nelems_total"], "halo_send_bytes_per_step_max_rank": int(other.allmax(float(other.halo.send_bytes(3)))),
            "overlap_vs_blocking_max_rel_diff": oc2,
            "owned_rows_vs_single_part_sub_brick": None if rc2 is None else {"max_rel_diff": rc2[0], "rows_compared_all_ranks": rc2[1]}}
        other.close()
        del other
    if rank == 0:
        balg = algorithmic_bytes(head_nelems, head_nnodes, nnz_total)
        achieved = balg / (kernel_ms * 1e-3) / 1e9
        prof, prof_other = pmc_profile(block[0] if block[0] == block[1] == block[2] and args.workload == "brick" else -1, args.scatter, kkey, build_id)
        kname = {"slot": "k_forward_jacobian<hex8,small_J2>", "wave_ad": "k_forward_jacobian_wave<hex8,small_J2>",
                 "wave": "k_forward_jacobian_wave_closed<hex8,small_J2>", "node": "k_node_rows_closed<hex8,small_J2>"}[kkey]
        if kkey == "node":
            kname += " (one wavefront per node: the node's four CSR rows formed from its elements and written once; no stage, no second kernel)"
        elif args.scatter == "gather":
            kname += " into the element stage + k_gather_rows, per chunk of elements"
        else:
            kname += " (%d launches per step)" % (asm.ncolors if args.scatter == "colored" else 1)
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS,
                           "traffic": prof.get("traffic_bytes_per_launch") if prof else None,
                           "algorithmic_bytes_per_step": balg, "kernel_ms_per_step": kernel_ms, "kernel": kname}
        # what binds: FP64 VALU issue, from the same PMC profile
        if prof:
            out["roofline"]["valu"] = valu_utilisation(prof, kernel_ms)
            out["roofline"]["traffic_profile"] = prof["file"]
            if prof.get("traffic_bytes_per_launch"):
                # beside the contract's fraction (algorithmic bytes / time / peak): the rate at which the kernels really
                # move their HBM bytes, and how many bytes they move per algorithmic byte
                tr = float(prof["traffic_bytes_per_launch"])
                out["roofline"]["traffic_rate"] = {"GB_per_s": tr / (kernel_ms * 1e-3) / 1e9,
                                                   "frac_of_peak": tr / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                   "bytes_per_algorithmic_byte": tr / balg}
        elif prof_other:
            out["roofline"]["profile_of_another_build"] = {
                "file": prof_other["file"], "build_id": prof_other.get("build_id"),
                "traffic": prof_other.get("traffic_bytes_per_launch"), "valu": valu_utilisation(prof_other, kernel_ms),
                "note": "PMC passes measured on another build of the library: reported for orientation only"}
    if world == 1 and not args.headline_only:
        # beside the headline (untimed above): the other entry points of the path on the same mesh and state,
        # HIP-event milliseconds per call (BASELINE config 3: primal + adjoint dR/dp assembly on one GPU)
        u, p, u0, p0, xi_prev, xi, ls = head.u, head.p, head.u0, head.p0, head.xi_prev, head.xi, head.ls
        asm.set_active(0, [0, 1, 2, 3])
        g_h = torch.zeros(asm.nelems, asm.npts, asm.nloc, dtype=torch.float64, device=dev)
        f_h = torch.zeros(asm.nelems, asm.npts, asm.ndofs, dtype=torch.float64, device=dev)
        phi = torch.zeros_like(g_h)
        z_u, z_p = torch.randn_like(u) * 1e-3, torch.randn_like(p) * 1e-3
        grad = torch.zeros(4, dtype=torch.float64, device=dev)
        Jq = torch.zeros(1, dtype=torch.float64, device=dev)

        def ms_of(fn, reps=3):
            fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / reps

        out["other_entry_points_ms"] = {
            "c8_assemble_adjoint_jacobian": ms_of(lambda: asm.adjoint_jacobian(u, p, u0, p0, xi_prev, xi, g_h, f_h, ls)),
            "c8_solve_adjoint_local": ms_of(lambda: asm.solve_adjoint_local(u, p, u0, p0, xi_prev, xi, z_u, z_p, phi, g_h, f_h)),
            "c8_param_gradient": ms_of(lambda: asm.qoi_gradient(u, p, u0, p0, xi_prev, xi, z_u, z_p, phi, grad)),
            "c8_assemble_residual": ms_of(lambda: asm.global_residual(u, p, u0, p0, xi_prev, xi, ls)),
            "c8_eval_qoi": ms_of(lambda: asm.eval_qoi(u, p, Jq)),
        }
        assert asm.status() == 0
        del g_h, f_h, phi
    if world == 1 and not args.no_cpu and not args.headline_only:
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        nthreads = args.cpu_threads if args.cpu_threads > 0 else max(1, min(16, avail))
        v1 = cpu_baseline(12, 1)[0]  # single-core rate on a small sample, for reference
        v, orc, (cu, cp, cz, czp), ls_o, xi_o = cpu_baseline(args.cpu_sample, nthreads)
        out["cpu_baseline"] = {"value": v, "unit": "elements/s", "cores": nthreads, "kind": "port",
                               "sample": "%d^3 hex8 brick (%d elements) of the same workload (the full mesh would take minutes); "
                                         "oracle/libc8oracle.so (CPU restatement of the reference algorithm, g++ -O2), %d threads, "
                                         "one element slice per thread and colour" % (args.cpu_sample, args.cpu_sample ** 3, nthreads),
                               "value_1core": v1}
        # parity gate on the sample: the same sub-problem through the HIP path
        from gpu_backend import GpuBackend
        from parity import compare_systems, rel_vec
        g = GpuBackend(8, orc.coords, orc.conn, "small_J2", J2, scatter=args.scatter, device=str(dev), kernel=args.kernel)
        ls_g, xi_g = g.new_linsys(), g.new_state()
        assert g.forward_jacobian(cu, cp, cz, czp, g.new_state(), xi_g, ls_g) == 0
        errs = compare_systems(orc, ls_g, ls_o)
        errs["xi"] = rel_vec(xi_g, xi_o)
        out["parity_max_rel_err"] = max(errs.values())
        out["speedup_vs_cpu_baseline"] = out["value"] / v
    if rank == 0:
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if world > 1:
        head.close()
        comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
