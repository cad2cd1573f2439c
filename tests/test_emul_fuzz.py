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
    tol = 1e-10 if model == "hyper_J2" else 1e-12  # see test_gpu_fuzz.py
    check_forward(orc, dut, c, model, eps, tol)
    check_residual(orc, dut, c, eps, tol)
    if not (kind == "hex8" and not dut.wave and dut.staged):
        check_adjoint_chain(orc, dut, c, model, eps, tol)
