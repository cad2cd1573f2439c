"""The kernel source (calibr8_amd/csrc/c8_assemble*.hpp) run lane-by-lane on the CPU must agree
with the oracle to 1e-12 (BASELINE.json tolerance).  This is the no-GPU rehearsal of the
`-m gpu` parity tests in test_gpu_parity.py: the same cases, the same checker."""
import numpy as np
import pytest

import emul_lib as em
from parity_cases import (CASES, MESHES, check_adjoint_chain, check_forward, check_residual, check_tiny_and_ragged,
                          check_two_element_sets, make_pair)

TOL = 1e-12


def factory(et, c, conn, model, params, **kw):
    return em.Emul(et, c, conn, model, params, **kw)


@pytest.mark.parametrize("mesh", MESHES)
@pytest.mark.parametrize("model,params,eps", CASES)
def test_forward_jacobian(mesh, model, params, eps):
    orc, dut, c = make_pair(factory, mesh, model, params)
    check_forward(orc, dut, c, model, eps, TOL)


@pytest.mark.parametrize("mesh", MESHES)
@pytest.mark.parametrize("model,params,eps", CASES)
def test_residual_only(mesh, model, params, eps):
    orc, dut, c = make_pair(factory, mesh, model, params)
    check_residual(orc, dut, c, eps, TOL)


@pytest.mark.parametrize("model,params,eps", CASES)
def test_residual_only_wave_kernel(model, params, eps):
    # eight hex8 elements per wavefront (residual_wave8); the 4 x 3 x 3 mesh ends with a partial group of 4
    orc, dut, c = make_pair(factory, "hex8", model, params)
    dut.wave = True
    check_residual(orc, dut, c, eps, TOL)


@pytest.mark.parametrize("mesh", MESHES)
@pytest.mark.parametrize("model,params,eps", CASES)
def test_adjoint_chain(mesh, model, params, eps):
    # K3 -> K4 -> K5 on two consecutive load steps with non-trivial history vectors
    orc, dut, c = make_pair(factory, mesh, model, params)
    check_adjoint_chain(orc, dut, c, model, eps, TOL)


@pytest.mark.parametrize("model,params,eps", CASES)
def test_forward_jacobian_wave_kernel(model, params, eps):
    # the one-wavefront-per-element hex8 kernel (c8_assemble_wave.hpp)
    orc, dut, c = make_pair(factory, "hex8", model, params)
    dut.wave = True
    check_forward(orc, dut, c, model, eps, TOL)


@pytest.mark.parametrize("model,params,eps", CASES)
def test_adjoint_chain_wave_kernel(model, params, eps):
    orc, dut, c = make_pair(factory, "hex8", model, params)
    dut.wave = True
    check_adjoint_chain(orc, dut, c, model, eps, TOL)


@pytest.mark.parametrize("model,params,eps", CASES)
def test_staged_gather_assembly(model, params, eps):
    # staged assembly: element matrices stored element-major, rows summed per node (gather_node_rows)
    orc, dut, c = make_pair(factory, "hex8", model, params)
    dut.wave = dut.staged = True
    check_forward(orc, dut, c, model, eps, TOL)
    check_adjoint_chain(orc, dut, c, model, eps, TOL)


@pytest.mark.parametrize("kind", ["tet4", "hex8"])
def test_staged_gather_assembly_slot_kernels(kind):
    # staged assembly through the slot-per-lane kernels: tet4 K1 and K3 (transposed through LDS), hex8 K1
    from parity_cases import J2
    orc, dut, c = make_pair(factory, kind, "small_J2", J2)
    dut.staged = True
    check_forward(orc, dut, c, "small_J2", 0.004, TOL)
    if kind == "tet4":
        check_adjoint_chain(orc, dut, c, "small_J2", 0.004, TOL)


def test_staged_assembly_goes_round_the_ring():
    # a long thin brick: 16 chunks through a ring of three, node rows summed as their last chunk completes
    import emul_lib
    import oracle_lib as ol
    from meshes import brick, jiggle
    from parity_cases import J2
    c, conn, sets = brick(2, 2, 16, 0.4, 0.4, 3.0)
    c = jiggle(c, sets, 0.03)
    orc, dut = ol.Oracle(ol.HEX8, c, conn, "small_J2", J2), factory(ol.HEX8, c, conn, "small_J2", J2)
    dut.wave = dut.staged = True
    check_forward(orc, dut, c, "small_J2", 0.004, TOL)
    assert emul_lib.lib().c8emu_last_nchunks() > 4
    check_adjoint_chain(orc, dut, c, "small_J2", 0.004, TOL)


@pytest.mark.parametrize("kind,wave", [("hex8", False), ("hex8", True), ("tet4", False)])
def test_two_element_sets(kind, wave):
    check_two_element_sets(factory, kind, TOL, wave=wave)


def test_tiny_and_ragged_meshes():
    check_tiny_and_ragged(factory, TOL)
