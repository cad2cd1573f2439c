"""The random parity sweep of test_gpu_fuzz.py with the kernel source run on the CPU (no GPU): same cases, the
emulated kernels against the oracle."""
import pytest

import emul_lib as em
import oracle_lib as ol
from parity_cases import check_adjoint_chain, check_forward, check_residual
from test_gpu_fuzz import random_case

pytestmark = []  # CPU test (the module it imports the case generator from is GPU-marked)


@pytest.mark.parametrize("seed", range(16))
def test_random_case_emulated_kernels_match_oracle(seed):
    model, params, kind, c, conn, eps, scatter, kernel = random_case(seed)
    et = ol.HEX8 if kind == "hex8" else ol.TET4
    orc = ol.Oracle(et, c, conn, model, params)
    dut = em.Emul(et, c, conn, model, params)
    dut.wave = kind == "hex8" and kernel == "auto"
    dut.staged = scatter == "gather"
    tol = 1e-12
    check_forward(orc, dut, c, model, eps, tol)
    check_residual(orc, dut, c, eps, tol)
    if not (kind == "hex8" and not dut.wave and dut.staged):
        check_adjoint_chain(orc, dut, c, model, eps, tol)


@pytest.mark.parametrize("seed", [445, 544])
def test_hyper_j2_onset_deviation_is_state_sensitivity_not_the_solver(seed):
    """Sweep seeds where the forward Jacobians of kernel and oracle differ by 1e-11 (hyper_J2, first plastic increment
    ~1e-8): the kernel's Jacobian equals the oracle's Jacobian evaluated AT THE KERNEL'S converged local state to
    machine precision, the two states agree to one unit in the last place, and the oracle's own Jacobian moves by as
    much as the observed deviation between those two states.  So the deviation is the conditioning of the map state ->
    Jacobian (curvature of the power-law flow stress), not the row-pivoted elimination (a complete-pivoting elimination
    in the reference's pivot order was tried: same 1.65e-11)."""
    import numpy as np
    from parity import compare_systems, rel_csr_rows, rel_vec
    from parity_cases import jacobian_at_state, two_steps
    model, params, kind, c, conn, eps, scatter, kernel = random_case(seed)
    assert model == "hyper_J2"
    et = ol.HEX8 if kind == "hex8" else ol.TET4
    orc = ol.Oracle(et, c, conn, model, params)
    dut = em.Emul(et, c, conn, model, params)
    st = two_steps(orc, c, eps)
    worst_direct = 0.0
    for n in (1, 2):
        (u, p, _), (up, pp, xip) = st[n], st[n - 1]
        ls_o, ls_d, xo, xd = orc.new_linsys(), dut.new_linsys(), orc.new_state(), dut.new_state()
        assert orc.forward_jacobian(u, p, up, pp, xip, xo, ls_o) == 0
        assert dut.forward_jacobian(u, p, up, pp, xip, xd, ls_d) == 0
        direct = compare_systems(orc, ls_d, ls_o)["A00"]
        at_dut, at_orc = jacobian_at_state(orc, u, p, up, pp, xip, xd), jacobian_at_state(orc, u, p, up, pp, xip, xo)
        same_state = rel_csr_rows(ls_d.A[0][0], at_dut[0][0], orc.rowptr[0][0])
        oracle_moves = rel_csr_rows(at_dut[0][0], at_orc[0][0], orc.rowptr[0][0])
        assert rel_vec(xd, xo) < 5e-15
        assert same_state < 1e-14
        if direct > 1e-12:
            assert oracle_moves > 0.2 * direct
        worst_direct = max(worst_direct, direct)
    assert worst_direct > 1e-12  # the seeds were picked because the direct comparison exceeds the bar
