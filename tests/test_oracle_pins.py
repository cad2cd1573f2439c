"""Pin the CPU oracle against the reference's own regression values.

Each case reproduces a deck under /root/reference/source/calibr8/test/primal/ on the
shipped cube mesh (tests/golden/cube_tet4.json, generated from test/mesh/cube/cube.msh) or the
shipped notch mesh (tests/golden/notch_tet4.json, read out of test/mesh/notch/notch0.smb by
tests/golden/make_notch_fixture.py, whose .smb reader is checked against the cube's gmsh file):
same material, boundary conditions, load steps and tolerances; the expected value and
its tolerance are the deck's `regression:` block.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from fe_driver import Dbc, Primal, Tbc

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def cube():
    d = json.load(open(os.path.join(HERE, "golden", "cube_tet4.json")))
    d["coords"] = np.array(d["coords"], dtype=np.float64)
    d["conn"] = np.array(d["conn"], dtype=np.int32)
    return d


def sym_dbcs(cube):
    ns = cube["node_sets"]
    return [Dbc(0, 0, ns["xmin"], lambda x, y, z, t: 0.0),
            Dbc(0, 1, ns["ymin"], lambda x, y, z, t: 0.0),
            Dbc(0, 2, ns["zmin"], lambda x, y, z, t: 0.0)]


def rel(a, b):
    return abs((a - b) / b)


def test_cube_elastic(cube):
    # primal/cube_elastic.yaml.in:21-27 (material), :30-34 (bcs), :40-41 (pin); also analytic:
    # free thermal expansion eps = cte*dT = 0.01 => integral of (ux+uy+uz)/3 over the unit cube = 0.005
    be = ol.Oracle(ol.TET4, cube["coords"], cube["conn"], "elastic", [1000.0, 0.25, 1.0e-3, 10.0])
    pr = Primal(be, cube["coords"], sym_dbcs(cube), max_iters=15, abs_tol=1e-8, rel_tol=1e-8).solve(1)
    assert rel(pr.qoi(), 5.00000000000000184e-3) < 1.0e-6
    assert rel(pr.qoi(), 5.0e-3) < 1.0e-10


def hyper_j2(cube, Y):
    # E nu Y S D A n K  (hyper_J2.cpp:83-90)
    return ol.Oracle(ol.TET4, cube["coords"], cube["conn"], "hyper_J2",
                     [1000.0, 0.25, Y, 0.0, 0.0, 0.0, 0.0, 100.0], max_iters=30, abs_tol=1e-12, rel_tol=1e-12)


def test_cube_hyper_J2(cube):
    # primal/cube_hyper_J2.yaml.in: 10 steps, ymax pulled by 0.01*t, pin :48-49
    be = hyper_j2(cube, 10.0)
    dbcs = sym_dbcs(cube) + [Dbc(0, 1, cube["node_sets"]["ymax"], lambda x, y, z, t: 0.01 * t)]
    pr = Primal(be, cube["coords"], dbcs, max_iters=15).solve(10)
    assert rel(pr.qoi(), 1.57817536611772440e-02) < 1.0e-4
    # plasticity was active: equivalent plastic strain alpha > 0 at the last step
    assert pr.xi[-1][:, :, 7].max() > 1e-3


def test_cube_hyperelasticity(cube):
    # primal/cube_hyperelasticity.yaml.in: 4 steps, Y = 1e5 (never yields), max 3 Newton iters, pin :48-49
    be = hyper_j2(cube, 100000.0)
    dbcs = sym_dbcs(cube) + [Dbc(0, 1, cube["node_sets"]["ymax"], lambda x, y, z, t: 0.001 * t)]
    pr = Primal(be, cube["coords"], dbcs, max_iters=3).solve(4)
    assert rel(pr.qoi(), 8.34720846455980019e-04) < 1.0e-4


def test_cube_hyperelasticity_traction(cube):
    # primal/cube_hyperelasticity_traction.yaml.in: clamp ymin, traction 0.1*t on ymax, pin :51-52
    be = hyper_j2(cube, 100000.0)
    ns = cube["node_sets"]
    dbcs = [Dbc(0, d, ns["ymin"], lambda x, y, z, t: 0.0) for d in range(3)]
    tbcs = [Tbc(0, cube["side_sets"]["ymax"], lambda x, y, z, t: (0.0, 0.1 * t, 0.0))]
    pr = Primal(be, cube["coords"], dbcs, tbcs, max_iters=10).solve(4)
    assert rel(pr.qoi(), 1.61757374785081228e-04) < 1.0e-4


@pytest.fixture(scope="module")
def notch():
    d = json.load(open(os.path.join(HERE, "golden", "notch_tet4.json")))
    d["coords"] = np.array(d["coords"], dtype=np.float64)
    d["conn"] = np.array(d["conn"], dtype=np.int32)
    return d


def test_notch_small_J2_deck_is_small_hill(notch):
    # primal/notch_small_J2.yaml.in: 1550 tets, 4 steps, ymax pulled by 0.001*t; the deck selects `small_hill` (:21)
    # with all R = 1 and Voce hardening Y 2, S 10, D 2 (:26-37); pin :50-51.  The 3-D small-strain plasticity pin.
    be = ol.Oracle(ol.TET4, notch["coords"], notch["conn"], "small_hill",
                   [1000.0, 0.25, 2.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 10.0, 2.0], max_iters=500, abs_tol=1e-12, rel_tol=1e-12)
    dbcs = sym_dbcs(notch) + [Dbc(0, 1, notch["node_sets"]["ymax"], lambda x, y, z, t: 0.001 * t)]
    pr = Primal(be, notch["coords"], dbcs, max_iters=15, abs_tol=1e-8, rel_tol=1e-8).solve(4)
    assert rel(pr.qoi(), 1.4622046563394649e-04) < 1.0e-11  # deck tolerance 1e-4; measured 3.7e-16
    assert pr.xi[-1][:, :, 6].max() > 1e-4  # the notch root has yielded


def test_notch_hyper_J2(notch):
    # primal/notch_hyper_J2.yaml.in: 4 steps, ymax pulled by 0.005*t, Y 10, K 100 (:26-34); pin :47-48
    be = ol.Oracle(ol.TET4, notch["coords"], notch["conn"], "hyper_J2",
                   [1000.0, 0.25, 10.0, 0.0, 0.0, 0.0, 0.0, 100.0], max_iters=500, abs_tol=1e-12, rel_tol=1e-12)
    dbcs = sym_dbcs(notch) + [Dbc(0, 1, notch["node_sets"]["ymax"], lambda x, y, z, t: 0.005 * t)]
    pr = Primal(be, notch["coords"], dbcs, max_iters=15, abs_tol=1e-8, rel_tol=1e-8).solve(4)
    assert rel(pr.qoi(), 7.0080671510235862e-04) < 1.0e-11  # deck tolerance 1e-4; measured 1.7e-15
    assert pr.xi[-1][:, :, 7].max() > 1e-4


def test_notch_hypo_J2_deck_is_hypo_hill(notch):
    # primal/notch_hypo_J2.yaml.in: `hypo_hill` (:21) with R = 1, Y 2, S 10, D 2, 4 steps, ymax pulled by 0.005*t; pin :50-51.
    # Exercises the polar rotation (minitensor::polar_rotation, restated) and the unrotated rate of deformation.
    be = ol.Oracle(ol.TET4, notch["coords"], notch["conn"], "hypo_hill",
                   [1000.0, 0.25, 2.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 10.0, 2.0], max_iters=500, abs_tol=1e-12, rel_tol=1e-12)
    dbcs = sym_dbcs(notch) + [Dbc(0, 1, notch["node_sets"]["ymax"], lambda x, y, z, t: 0.005 * t)]
    pr = Primal(be, notch["coords"], dbcs, max_iters=15, abs_tol=1e-8, rel_tol=1e-8).solve(4)
    assert rel(pr.qoi(), 7.5441386985803955e-04) < 1.0e-9  # deck tolerance 1e-4; measured 1.4e-11
    assert pr.xi[-1][:, :, 6].max() > 1e-2


def test_baseline_config1_hex8_elastic_bar():
    # BASELINE.json configs[0] (SURVEY.md 8d "Config 1 (plumbing)"): 10 x 10 x 10 hex8 unit bar, `elastic` with the
    # material of cube_elastic.yaml.in:21-27, symmetric BCs on xmin/ymin/zmin: free thermal expansion,
    # eps = cte * dT = 0.01, so the average-displacement QoI is the deck's pin 5e-3 (times the unit volume)
    from meshes import brick
    c, conn, sets = brick(10, 10, 10)
    be = ol.Oracle(ol.HEX8, c, conn, "elastic", [1000.0, 0.25, 1e-3, 10.0])
    dbcs = [Dbc(0, k, sets[s], lambda x, y, z, t: 0.0) for k, s in enumerate(["xmin", "ymin", "zmin"])]
    pr = Primal(be, c, dbcs).solve(1)
    assert rel(pr.qoi(), 5.0e-3) < 1e-10
    assert np.abs(pr.u[1].reshape(-1, 3) - 0.01 * c).max() < 1e-12  # u = eps * x exactly
