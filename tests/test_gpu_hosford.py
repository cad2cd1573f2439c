"""`-m gpu`: the Hosford / Barlat family on the device: `small_hosford`, `hypo_hosford`,
`hypo_barlat` through the lane-group kernels on tet4 and hex8 and through the wave-per-element kernels on hex8 (local Newton
iteration with forced branch and line search inside the assembly kernel), every entry point against the oracle at 1e-12, and
the reference's two decks end to end."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from parity_cases import (BARLAT, CASES_LINE_SEARCH, HOSFORD_100, LOCAL_LINE_SEARCH, check_adjoint_chain, check_forward,
                          check_residual, mesh_of)

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("scatter", ["colored", "atomic", "default"])
@pytest.mark.parametrize("kind", ["tet4", "hex8"])
@pytest.mark.parametrize("model,params,eps", CASES_LINE_SEARCH)
def test_line_search_models_match_oracle(model, params, eps, kind, scatter):
    from gpu_backend import GpuBackend
    et, c, conn = mesh_of(kind)
    orc = ol.Oracle(et, c, conn, model, params)
    orc.set_local_line_search(*LOCAL_LINE_SEARCH)
    # "default": the library's own choice -- the staged assembly of K1 and K3; on hex8 through the wave-per-element kernels
    gpu = GpuBackend(et, c, conn, model, params, scatter=None if scatter == "default" else scatter, line_search=LOCAL_LINE_SEARCH)
    check_forward(orc, gpu, c, model, eps, 1e-12)
    check_residual(orc, gpu, c, eps, 1e-12)
    check_adjoint_chain(orc, gpu, c, model, eps, 1e-12)


@pytest.mark.parametrize("scatter", ["colored", "default"])
@pytest.mark.parametrize("model,params,eps", CASES_LINE_SEARCH)
def test_hex8_lane_group_kernels_match_oracle(model, params, eps, scatter):
    # `slot`: one thread per point and DOF slot, the kernels these models ran through before they had wave-per-element ones
    # (their adjoint Jacobian kernel cannot stage on hex8: colour batches in the default mode)
    from gpu_backend import GpuBackend
    et, c, conn = mesh_of("hex8")
    orc = ol.Oracle(et, c, conn, model, params)
    orc.set_local_line_search(*LOCAL_LINE_SEARCH)
    gpu = GpuBackend(et, c, conn, model, params, scatter=None if scatter == "default" else scatter, line_search=LOCAL_LINE_SEARCH)
    gpu.asm.set_kernel("slot")
    check_forward(orc, gpu, c, model, eps, 1e-12)
    check_residual(orc, gpu, c, eps, 1e-12)
    check_adjoint_chain(orc, gpu, c, model, eps, 1e-12)


def test_hex8_wave_and_lane_group_kernels_agree():
    # the same Newton + line-search decisions in both layouts: state, matrices and right-hand side to rounding
    import torch
    from calibr8_amd import Assembler
    from meshes import prescribed_fields
    et, c, conn = mesh_of("hex8")
    asm = Assembler(8, c, conn, "hypo_barlat", BARLAT, line_search=LOCAL_LINE_SEARCH)
    assert asm.scatter == "gather" and asm.nloc == 7
    u_h, p_h = prescribed_fields(c, 0.006, ramp=True, perturb=5e-2)
    u, p = asm.dev(u_h), asm.dev(p_h)
    z = torch.zeros_like(u), torch.zeros_like(p)
    out = {}
    for k in ("wave", "slot"):
        asm.set_kernel(k)
        xi0, xi, ls = asm.new_state(), asm.new_state(), asm.new_linsys()
        asm.forward_jacobian(u, p, z[0], z[1], xi0, xi, ls)
        assert asm.status() == 0
        out[k] = (xi.clone(), ls.flat.clone())
    assert float((out["wave"][0][:, :, 6] > 0).double().mean()) > 0.3
    assert float((out["wave"][0] - out["slot"][0]).abs().max()) < 1e-12
    assert float((out["wave"][1] - out["slot"][1]).abs().max()) < 1e-11 * float(out["slot"][1].abs().max())


def notch():
    d = json.load(open(os.path.join(HERE, "golden", "notch_tet4.json")))
    return np.array(d["coords"]), np.array(d["conn"], dtype=np.int32), {k: np.array(v, dtype=np.int32) for k, v in d["node_sets"].items()}


@pytest.mark.parametrize("deck", ["notch_small_hosford", "notch_hypo_barlat"])
def test_notch_hosford_barlat_regressions_with_device_newton_driver(deck):
    # test/primal/notch_small_hosford.yaml.in and notch_hypo_barlat.yaml.in end to end on the device
    from calibr8_amd import Assembler
    from calibr8_amd.primal import PrimalDriver
    c, conn, ns = notch()
    if deck == "notch_small_hosford":
        model, params, rate, pin, tol = "small_hosford", HOSFORD_100, 0.001, 1.4447629888205869e-04, 1e-7
    else:
        model, params, rate, pin, tol = "hypo_barlat", [v if k != 4 else 0.0 for k, v in enumerate(BARLAT)], 0.01, 1.3989452247489746e-03, 1e-10
    asm = Assembler(4, c, conn, model, params, line_search=LOCAL_LINE_SEARCH)
    dbcs = [(0, 0, ns["xmin"], lambda x, y, z, t: 0.0), (0, 1, ns["ymin"], lambda x, y, z, t: 0.0),
            (0, 2, ns["zmin"], lambda x, y, z, t: 0.0), (0, 1, ns["ymax"], lambda x, y, z, t: rate * t)]
    drv = PrimalDriver(asm, dbcs, max_iters=15, abs_tol=1e-8, rel_tol=1e-8).solve(4)
    J = drv.qoi()
    assert abs(J - pin) / pin < tol, (deck, J, pin)
    assert float(drv.xi[-1][:, :, 6].max()) > 1e-2
