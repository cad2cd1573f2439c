"""bench.py -- element Jacobian assemblies / second on synthetic hex8 J2-plasticity bricks.

One "step" = one pass of the hot path (c8_assemble_forward_jacobian: residual + Jacobian with the
local return-mapping solves and the CSR scatter) over the rank's mesh part, inputs resident in HBM.
Workload (BASELINE.json metric: "1M hex8 J2-plasticity fp64"): a 100x100x100 hex8 brick per GPU,
small_J2 (E 1000, nu 0.25, K 100, Y 2), prescribed mixed elastic/plastic state of SURVEY.md 8d.
For N > 1 the workload is weak-scaled by default: rank r owns one 100^3 block of a (px*100, py*100, pz*100) brick
(2x2x2 blocks of an 8M-element brick at N = 8, BASELINE.json config 5); --scaling strong splits the one 100^3 brick
N ways instead (BASELINE.json config 4).  A step is then the assembly plus the owned/ghost halo ADD of the Jacobian
and residual (LinearAlg::gather_A/gather_b): HIP pack / unpack kernels and grouped RCCL point-to-point messages of
libc8.so (c8_halo_*), no torch op on the path.  `value` = all elements of all ranks / max-over-ranks wall time.  The
exchange overlaps the assembly: in the default staged mode the ghost rows are summed first and travel while the owned
rows are summed; with --scatter atomic the elements that add into ghost rows are assembled first.  torch.distributed
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


def main():
    # stdout carries ONE line, the JSON result: everything else that writes to file descriptor 1 during the run (gloo's
    # connection messages, library banners of the ranks) goes to stderr
    real_stdout = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--edge", dest="n", type=int, default=100, help="brick edge (elements) per GPU")
    ap.add_argument("--scatter", default="gather", choices=["colored", "atomic", "gather"],
                    help="gather (default): staged assembly + row sums, no atomics, bitwise reproducible, fastest")
    ap.add_argument("--cpu-sample", type=int, default=32, help="edge of the CPU-baseline sample brick")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = host cores available, max 16)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "slot", "wave", "wave_ad", "node"])
    ap.add_argument("--stage-overlap", type=int, default=-1, help="scatter=gather: 1 = row sums of a chunk beside the assembly of the next (second stream), 0 = one after the other; default: the library's choice")
    ap.add_argument("--stage-chunk", type=int, default=0, help="scatter=gather: minimum elements per staged chunk")
    ap.add_argument("--assign", action="store_true", help="scatter=gather: c8_set_assign_mode (zero_all + assembly in one call); "
                    "NOT the default: the contract metric is the accumulate-into assembly")
    ap.add_argument("--workload", default="brick", choices=["brick", "notch"],
                    help="brick (default: the contract workload); notch = BASELINE config 3's geometry, a double-edge-notched "
                         "hex8 bar of about edge^3 elements per GPU (tests/meshes.py notched_bar), cut into slabs along the bar for N > 1")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: exchange the ghost rows after the whole assembly")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = one edge^3 block per GPU (default); strong = the one edge^3 brick split N ways")
    ap.add_argument("--transport", "--backend", dest="transport", default="rccl", choices=["rccl", "host", "nccl", "gloo"],
                    help="rccl (default): grouped ncclSend/ncclRecv over xGMI, one rank per GPU; host: rehearsal of the N > 1 "
                         "path with all ranks on one GPU (RCCL refuses that), messages through the host")
    args = ap.parse_args()
    args.transport = {"nccl": "rccl", "gloo": "host"}.get(args.transport, args.transport)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node N for --gpus N"
    ndev = torch.cuda.device_count()
    if world > ndev and args.transport == "rccl":
        # fewer cards than ranks (a rehearsal on a shared card): RCCL admits one rank per card, so the messages go through
        # the host; same on every rank (one node), and said in config.transport
        print("rank %d: %d ranks on %d GPU(s): host transport" % (rank, world, ndev), file=sys.stderr)
        args.transport = "host"
    if args.transport == "host":
        local_rank = local_rank % max(ndev, 1) if world > ndev else 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("gloo")  # control plane only: rendezvous, exchange lists, timing; the data path is libc8.so's

    from calibr8_amd import Assembler
    from calibr8_amd import distributed as D
    from calibr8_amd import lib as c8lib
    from meshes import prescribed_fields

    # mesh part of this rank.  weak: an n^3 block of the (px*n, py*n, pz*n) brick; strong: the (n/px, n/py, n/pz) block
    # of the one n^3 brick.  Element edge 1/n everywhere.
    n = args.n
    pdims = D.pdims_for(world)
    if args.scaling == "strong":
        assert all(n % q == 0 for q in pdims), "--scaling strong: the edge must divide by the part grid"
        block = tuple(n // q for q in pdims)
    else:
        block = (n, n, n)
    if args.workload == "notch":
        # about n^3 elements per GPU: a (4 : 1 : 1) bar, 15 % of whose elements the two notches remove; slabs along the bar
        from meshes import notched_bar
        assert args.scaling == "weak", "--workload notch is weak-scaled"
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
    plan = D.HaloPlan(part, dist if world > 1 else None)
    coords = plan.coords
    asm = Assembler(8, coords, part.conn, "small_J2", J2, device=str(dev), scatter=args.scatter,
                    extra_pairs=plan.extra_pairs)
    asm.set_kernel(args.kernel)
    if args.stage_chunk > 0:
        asm.set_stage_chunk(args.stage_chunk)
    if args.stage_overlap >= 0:
        asm.set_stage_overlap(args.stage_overlap)
    transport = args.transport
    comm = halo = None
    if world > 1:
        if transport == "rccl":
            try:
                comm = D.Comm.rccl(dist, rank, world)
                # pre-flight: one small all-reduce through the new communicator; a wrong sum or an error sends every rank
                # to the host transport below
                chk = comm.allreduce(np.array([1.0, float(rank)]))
                if abs(chk[0] - world) > 1e-12 or abs(chk[1] - 0.5 * world * (world - 1)) > 1e-9:
                    raise RuntimeError("all-reduce through the RCCL communicator returned %r" % (chk,))
                ok = 1.0
            except Exception as e:  # every rank must take the same branch
                print("rank %d: RCCL communicator failed (%s)" % (rank, e), file=sys.stderr)
                ok = 0.0
            t = torch.tensor([ok])
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if float(t.item()) == 0.0:
                if comm is not None:
                    comm.close()
                comm, transport = None, "host (RCCL communicator could not be created)"
        if comm is None:
            comm = D.Comm.host(dist, rank, world)
        halo = D.Halo(plan, asm.rowptr[1][1], asm.colidx[1][1], asm, comm)
    # prescribed state of SURVEY.md 8d on the rank's own block (local coordinates of the block)
    origin = coords[: part.ntouched].min(axis=0)
    u_h, p_h = prescribed_fields(coords - origin, 0.004, ramp=True, seed=1234 + rank)
    u, p = asm.dev(u_h), asm.dev(p_h)
    u0, p0 = torch.zeros_like(u), torch.zeros_like(p)
    xi_prev, xi = asm.new_state(), asm.new_state()
    ls = asm.new_linsys()
    asm.set_async(True)

    overlap = world > 1 and args.scatter == "atomic" and not args.no_overlap
    split = world > 1 and args.scatter == "gather" and not args.no_overlap
    if overlap:
        e_if = torch.as_tensor(plan.interface_elems, device=dev)
        e_in = torch.as_tensor(plan.interior_elems, device=dev)
    if args.assign:
        asm.set_assign_mode(True)
    if split:  # staged assembly in two parts: the ghost rows (local nodes nowned .. ntouched) are summed first
        if args.stage_chunk <= 0:
            asm.set_stage_chunk(asm.nelems)  # the two-part row sums need the whole part in one staged chunk
        asm.set_gather_early_nodes(part.nowned, part.ntouched)

    def step(ev=None):
        """eval_forward_jacobian, then la->gather_A / gather_b (primal.cpp:99,110-111).  With more than one rank the
        exchange of the ghost rows (one message per neighbour) runs beside assembly work; `ev` = HIP-event pairs around
        the assembly launches."""
        if split:  # every element staged, ghost rows summed; the other rows are summed while the ghost rows travel
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
        if not overlap:
            if ev:
                ev[0][0].record()
            asm.forward_jacobian(u, p, u0, p0, xi_prev, xi, ls)
            if ev:
                ev[0][1].record()
            if world > 1:
                halo.gather(ls)
            return
        if ev:
            ev[0][0].record()
        asm.forward_jacobian_subset(u, p, u0, p0, xi_prev, xi, ls, e_if)
        if ev:
            ev[0][1].record()
        halo.gather_start(ls)
        if ev:
            ev[1][0].record()
        asm.forward_jacobian_subset(u, p, u0, p0, xi_prev, xi, ls, e_in)
        if ev:
            ev[1][1].record()
        halo.gather_finish(ls)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    assert asm.status() == 0
    E = lambda: torch.cuda.Event(enable_timing=True)
    ev = [[(E(), E()), (E(), E())] for _ in range(args.steps)]  # HIP events on the stream the kernels are launched on
    barrier()
    t0 = time.perf_counter()
    for pairs in ev:
        step(pairs)
    barrier()
    dt = time.perf_counter() - t0
    assert asm.status() == 0
    # assembly kernels only (both launches of a step when the exchange is overlapped)
    kernel_ms = float(np.mean([sum(a.elapsed_time(b) for a, b in pairs[:2 if (overlap or split) else 1]) for pairs in ev]))
    tmax = torch.tensor([dt], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    def timed(fn, reps):
        barrier()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        barrier()
        tt = torch.tensor([time.perf_counter() - t], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return 1e3 * float(tt.item()) / reps

    # beside the headline (untimed above): the reference's step is zero_all + eval_forward_jacobian (primal.cpp:98-99)
    reps = max(2, min(5, args.steps))

    def step_with_zero():
        ls.zero()
        step()

    also = {"ms_per_step_with_zero_all": timed(step_with_zero, reps)}
    if args.scatter == "gather" and not args.assign:
        asm.set_assign_mode(True)  # zero_all + assembly in one call (c8_set_assign_mode): no zeroing pass, no read of the old values
        also["ms_per_step_assign_mode"] = timed(step, reps)
        asm.set_assign_mode(False)
    if args.kernel in ("auto", "wave"):
        # the same step with the local Newton iteration and the AD passes kept in the kernel (C8_KERNEL_WAVE_AD): what every
        # model without a closed form runs, and what round 1 timed
        asm.set_kernel("wave_ad")
        also["ms_per_step_iterated_ad_form"] = timed(step, reps)
        asm.set_kernel(args.kernel)
    assert asm.status() == 0

    overlap_check = None
    if overlap or split:  # untimed: the overlapped step against the blocking exchange after a whole assembly
        ls.zero()
        step()
        ref_flat = ls.flat.clone()
        ls.zero()
        if split:
            asm.set_gather_early_nodes(0, 0)
        asm.forward_jacobian(u, p, u0, p0, xi_prev, xi, ls)
        halo.gather(ls)
        no = part.nowned  # owned rows only: ghost rows are scratch after the exchange
        d = 0.0
        for k, neq in ((4, 3), (5, 1)):
            lo = int(ls.offsets[k])
            d = max(d, float((ls.flat[lo:lo + no * neq] - ref_flat[lo:lo + no * neq]).abs().max() /
                             ls.flat[lo:lo + no * neq].abs().max()))
        lo, hi = 0, int(asm.rowptr[0][0][no * 3])
        d = max(d, float((ls.flat[lo:hi] - ref_flat[lo:hi]).abs().max() / ls.flat[lo:hi].abs().max()))
        chk = torch.tensor([d], dtype=torch.float64)
        dist.all_reduce(chk, op=dist.ReduceOp.MAX)
        overlap_check = float(chk.item())
        assert overlap_check < 1e-12, overlap_check
    hb = torch.tensor([float(halo.send_bytes(3)) if world > 1 else 0.0], dtype=torch.float64)
    ne = torch.tensor([float(asm.nelems)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(hb, op=dist.ReduceOp.MAX)
        dist.all_reduce(ne, op=dist.ReduceOp.SUM)
    halo_bytes = int(hb.item())
    nelems_total = int(ne.item())
    value = nelems_total * args.steps / dt
    if "ms_per_step_iterated_ad_form" in also:  # the same metric for the iterated AD form (whole job, elements per second)
        also["elements_per_s_iterated_ad_form"] = nelems_total / (1e-3 * also["ms_per_step_iterated_ad_form"])
    plastic_frac = float((xi[:, :, 6] > 0).double().mean().item())
    build_info = c8lib.load_library().c8_build_info().decode()
    build_id = build_info.split()[0].split("=")[1]

    out = {
        "metric": "element Jacobian assemblies/sec, 1M hex8 J2-plasticity fp64",
        "value": value, "unit": "elements/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("double-edge-notched hex8 bar (BASELINE config 3's geometry: the %dx%dx%d brick minus two V-notches; "
                                "%d elements, %d nodes on this GPU%s), small_J2 E1000 nu0.25 K100 Y2, prescribed ramped uniaxial state "
                                "eps=0.004 seed 1234, residual+Jacobian assembly"
                                % (block[0], block[1], block[2], asm.nelems, asm.nnodes, ", slabs along the bar" if world > 1 else ""))
                               if args.workload == "notch" else
                               "%dx%dx%d hex8 brick per GPU (%d elements, %d nodes)%s, small_J2 E1000 nu0.25 K100 Y2, "
                               "prescribed ramped uniaxial state eps=0.004 seed 1234, residual+Jacobian assembly; a "
                               "structured brick stands in for the notched specimen (SURVEY.md 8d; --workload notch times that one)"
                               % (block[0], block[1], block[2], asm.nelems, asm.nnodes,
                                  " = the %d^3 brick (BASELINE config 4) split %dx%dx%d" % ((n,) + tuple(pdims)) if args.scaling == "strong" and world > 1 else ""),
                   "elements_per_gpu": asm.nelems, "elements_total": nelems_total, "plastic_fraction": plastic_frac,
                   "scatter": args.scatter, "kernel": args.kernel,
                   "local_solve": "iterated: local Newton + forward-mode AD in the kernel (C8_KERNEL_WAVE_AD)" if args.kernel in ("slot", "wave_ad")
                                  else "closed form of small_J2 (radial return + consistent tangent, same state and Jacobian "
                                       "to 2e-13; the library's default); also.ms_per_step_iterated_ad_form times the "
                                       "Newton + AD form of the same kernel",
                   "colors": asm.ncolors, "part_grid": list(pdims),
                   "parallelism": "one element block per GPU; ghost rows of A and b ADDed into their owners: HIP pack kernel, "
                                  "one grouped ncclSend/ncclRecv message per neighbour (RCCL over xGMI), HIP unpack-add kernel "
                                  "(c8_halo_gather_start / _finish)",
                   "transport": transport if world > 1 else None,
                   "halo_send_bytes_per_step_max_rank": halo_bytes,
                   "halo_overlapped_with_interior_assembly": bool(overlap),
                   "halo_overlapped_with_owned_row_sums": bool(split), "assign_mode": bool(args.assign),
                   "overlap_vs_blocking_max_rel_diff": overlap_check, "library_build": build_info},
        "also": also,
    }
    if rank == 0:
        nnz_total = sum(asm.nnz[i][j] for i in range(2) for j in range(2))
        balg = algorithmic_bytes(asm.nelems, asm.nnodes, nnz_total)
        achieved = balg / (kernel_ms * 1e-3) / 1e9
        prof, prof_other = pmc_profile(block[0] if block[0] == block[1] == block[2] and args.workload == "brick" else -1, args.scatter,
                                       args.kernel if args.kernel in ("slot", "wave_ad") else "wave", build_id)
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS,
                           "traffic": prof.get("traffic_bytes_per_launch") if prof else None,
                           "algorithmic_bytes_per_step": balg, "kernel_ms_per_step": kernel_ms,
                           "kernel": {"slot": "k_forward_jacobian", "wave_ad": "k_forward_jacobian_wave"}.get(args.kernel, "k_forward_jacobian_wave_closed") +
                                     ("<hex8,small_J2> into the element stage + k_gather_rows, per chunk of elements"
                                      if args.scatter == "gather" else "<hex8,small_J2> (%d launches per step)"
                                      % (asm.ncolors if args.scatter == "colored" else 1))}
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
                "note": "PMC passes measured on other kernels than this library build: reported for orientation only"}
        if world == 1:
            # beside the headline (untimed above): the other entry points of the path on the same mesh and state,
            # HIP-event milliseconds per call (BASELINE config 2: primal + adjoint dR/dp assembly on one GPU)
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
        if not args.no_cpu and world == 1:
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            nthreads = args.cpu_threads if args.cpu_threads > 0 else max(1, min(16, avail))
            v1 = cpu_baseline(12, 1)[0]  # single-core rate on a small sample, for reference
            v, orc, (cu, cp, cz, czp), ls_o, xi_o = cpu_baseline(args.cpu_sample, nthreads)
            out["cpu_baseline"] = {"value": v, "unit": "elements/s", "cores": nthreads, "kind": "port",
                                   "sample": "%d^3 hex8 brick (%d elements) of the same workload; oracle/libc8oracle.so "
                                             "(CPU restatement of the reference algorithm, g++ -O2), %d threads, one "
                                             "element slice per thread and colour" % (args.cpu_sample, args.cpu_sample ** 3, nthreads),
                                   "value_1core": v1}
            # parity gate on the sample: the same sub-problem through the HIP path
            from gpu_backend import GpuBackend
            from parity import compare_systems, rel_vec
            g = GpuBackend(8, orc.coords, orc.conn, "small_J2", J2, scatter=args.scatter, device=str(dev))
            ls_g, xi_g = g.new_linsys(), g.new_state()
            assert g.forward_jacobian(cu, cp, cz, czp, g.new_state(), xi_g, ls_g) == 0
            errs = compare_systems(orc, ls_g, ls_o)
            errs["xi"] = rel_vec(xi_g, xi_o)
            out["parity_max_rel_err"] = max(errs.values())
            out["speedup_vs_cpu_baseline"] = value / v
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if world > 1:
        halo.close()
        comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
