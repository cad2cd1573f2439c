"""The kernel source (calibr8_amd/csrc/c8_assemble.hpp) run lane-by-lane on the CPU must agree
with the oracle to 1e-12 (BASELINE.json tolerance).  This is the no-GPU rehearsal of the
`-m gpu` parity tests in test_gpu_parity.py."""
import json
import os

import numpy as np
import pytest

import emul_lib as em
import oracle_lib as ol
from meshes import brick, jiggle, prescribed_fields
from parity import compare_systems, rel_vec

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-12
J2 = [1000.0, 0.25, 100.0, 2.0, 0.0, 0.0]
HJ2 = [1000.0, 0.25, 2.0, 1.0, 5.0, 0.5, 0.5, 100.0]
EL = [1000.0, 0.25, 1e-3, 10.0]


def hex_case(model, params, eps, ramp=True):
    c, conn, sets = brick(4, 3, 3, 1.0, 0.8, 0.7)
    c = jiggle(c, sets, 0.04)
    orc = ol.Oracle(ol.HEX8, c, conn, model, params)
    u, p = prescribed_fields(c, eps, ramp=ramp, perturb=5e-2)
    return orc, u, p


def tet_case(model, params, eps):
    d = json.load(open(os.path.join(HERE, "golden", "cube_tet4.json")))
    c, conn = np.array(d["coords"]), np.array(d["conn"], dtype=np.int32)
    orc = ol.Oracle(ol.TET4, c, conn, model, params)
    u, p = prescribed_fields(c, eps, ramp=True, perturb=5e-2)
    return orc, u, p


def run_both(orc, model, u, p, u_prev, p_prev, xi_prev):
    ls_o, xi_o = orc.new_linsys(), orc.new_state()
    assert orc.forward_jacobian(u, p, u_prev, p_prev, xi_prev, xi_o, ls_o) == 0
    ls_e, xi_e = orc.new_linsys(), orc.new_state()
    assert em.forward_jacobian(orc, u, p, u_prev, p_prev, xi_prev, xi_e, ls_e, model) == 0
    return ls_o, xi_o, ls_e, xi_e


@pytest.mark.parametrize("mesh", ["hex8", "tet4"])
@pytest.mark.parametrize("model,params,eps", [("small_J2", J2, 0.001), ("small_J2", J2, 0.004),
                                              ("elastic", EL, 0.002), ("hyper_J2", HJ2, 0.004)])
def test_forward_jacobian_emulated_kernel_matches_oracle(mesh, model, params, eps):
    orc, u, p = (hex_case if mesh == "hex8" else tet_case)(model, params, eps)
    u0, p0 = np.zeros_like(u), np.zeros_like(p)
    xi_prev = orc.new_state()
    ls_o, xi_o, ls_e, xi_e = run_both(orc, model, u, p, u0, p0, xi_prev)
    errs = compare_systems(orc, ls_e, ls_o)
    errs["xi"] = rel_vec(xi_e, xi_o)
    assert max(errs.values()) < TOL, errs
    # second load step from the converged state (non-trivial xi_prev, u_prev)
    u2, p2 = 1.5 * u, 1.5 * p
    ls_o, xi_o2, ls_e, xi_e2 = run_both(orc, model, u2, p2, u, p, xi_o)
    errs = compare_systems(orc, ls_e, ls_o)
    errs["xi"] = rel_vec(xi_e2, xi_o2)
    assert max(errs.values()) < TOL, errs
    if model == "small_J2" and eps > 0.003:
        assert (xi_o2[:, :, 6] > 0).mean() > 0.3  # the plastic branch really ran
