"""The 2-D path (tri3, `mechanics` with 2 + 1 equations per node) on the CPU: the kernel source
run by the lane emulator against the oracle, every entry point, on a jiggled structured triangle mesh and on the
reference's notch2D mesh.  The oracle's 2-D path is pinned to the reference in test_oracle_pins.py."""
import numpy as np
import pytest

import emul_lib as em
import oracle_lib as ol
from parity_cases import ACTIVE, CASES_2D, CASES_PLANE_STRESS, check_adjoint_chain, check_forward, check_residual, mesh_2d


@pytest.mark.parametrize("mesh", ["structured", "notch2D"])
@pytest.mark.parametrize("model,params,eps", CASES_2D)
def test_emulated_2d_kernels_match_oracle(model, params, eps, mesh):
    et, c, conn = mesh_2d(mesh)
    orc = ol.Oracle(et, c, conn, model, params)
    dut = em.Emul(et, c, conn, model, params)
    assert orc.ndims == 2 and orc.nloc in (4, 5) and orc.ndofs == 9
    check_forward(orc, dut, c, model, eps, 1e-12)
    check_residual(orc, dut, c, eps, 1e-12)
    check_adjoint_chain(orc, dut, c, model, eps, 1e-12)


@pytest.mark.parametrize("mesh", ["structured", "notch2D"])
@pytest.mark.parametrize("model,params,eps", CASES_PLANE_STRESS)
def test_emulated_plane_stress_kernels_match_oracle(model, params, eps, mesh):
    # `mechanics_plane_stress`: one global residual (u), one ip set, six element DOFs; the p arrays and the other three
    # blocks of the test containers must come back untouched (compare_systems checks them against the oracle's zeros)
    et, c, conn = mesh_2d(mesh)
    orc = ol.Oracle(et, c, conn, model, params)
    dut = em.Emul(et, c, conn, model, params)
    assert orc.nres == 1 and orc.ndims == 2 and orc.nloc in (4, 5, 6) and orc.ndofs == 6
    check_forward(orc, dut, c, model, eps, 1e-12)
    check_residual(orc, dut, c, eps, 1e-12)
    check_adjoint_chain(orc, dut, c, model, eps, 1e-12)


def test_plane_stress_plastic_branch_ran():
    from parity_cases import two_steps
    et, c, conn = mesh_2d("structured")
    for model, params, eps in CASES_PLANE_STRESS[1:]:
        st = two_steps(ol.Oracle(et, c, conn, model, params), c, eps)
        alpha = st[2][2][:, :, 3 if model != "hyper_J2_plane_stress" else 5]
        assert (alpha > 0).mean() > 0.3, model


def test_2d_plastic_branch_ran():
    from parity_cases import J2, two_steps
    et, c, conn = mesh_2d("structured")
    orc = ol.Oracle(et, c, conn, "small_J2", J2)
    st = two_steps(orc, c, 0.004)
    assert (st[2][2][:, :, 3] > 0).mean() > 0.3
