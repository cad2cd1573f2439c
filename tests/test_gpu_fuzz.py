"""Seeded random sweep of the parity tests: element type x model x scatter mode x kernel variant x mesh size x state
(elastic to fully plastic, two load steps with history) against the oracle, bar 1e-12 on every
quantity of every model.  C8_FUZZ_SEEDS=n widens the sweep (tools/emul_sweep.py runs the same cases on the CPU emulator:
3000 seeds, none above 1e-12).  Deterministic: every case is a function of its seed."""
import numpy as np
import pytest

import oracle_lib as ol
from meshes import brick, jiggle
from parity_cases import ACTIVE, AUDIT, CASES, check_adjoint_chain, check_forward, check_residual

pytestmark = pytest.mark.gpu
PARAMS = {m: p for m, p, _ in CASES}


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    model = list(PARAMS)[rng.integers(len(PARAMS))]
    kind = ["hex8", "tet4"][rng.integers(2)]
    n = tuple(int(v) for v in rng.integers(2, 5, 3))
    c, conn, sets = brick(n[0], n[1], n[2], *(0.5 + rng.random(3)))
    c = jiggle(c, sets, 0.02 + 0.1 * rng.random(), seed=seed)
    if kind == "tet4":
        tets = [[0, 1, 2, 6], [0, 2, 3, 6], [0, 3, 7, 6], [0, 7, 4, 6], [0, 4, 5, 6], [0, 5, 1, 6]]
        conn = np.concatenate([conn[:, t] for t in tets]).astype(np.int32)
        X = c[conn]
        vol = np.einsum("ij,ij->i", np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), X[:, 3] - X[:, 0])
        assert (vol > 0).all()
    eps = float(10 ** rng.uniform(-3.3, -2.2))
    scatter = ["colored", "atomic", "gather"][rng.integers(3)]
    # hex8: the library's choice, the lane-group kernel, the wave kernel with the iterated AD form (what every model without a
    # closed form runs), and for small_J2 also its two closed-form kernels by name (staged wave kernel; row-per-node kernel)
    kernels = ["auto", "slot", "wave_ad"] if kind == "hex8" else ["auto"]
    if kind == "hex8" and model == "small_J2":
        kernels += ["wave"] + (["node"] if scatter == "gather" else [])
    kernel = kernels[rng.integers(len(kernels))]
    # perturbed parameters (+-20 %), kept admissible
    p = np.array(PARAMS[model], dtype=np.float64)
    p = p * (1.0 + 0.2 * (rng.random(len(p)) - 0.5) * (np.abs(p) > 0))
    if model in ("elastic", "small_J2", "hyper_J2", "small_hill", "isotropic_elastic", "hypo_hill"):
        p[1] = min(p[1], 0.4)  # Poisson's ratio
    return model, list(p), kind, c, conn, eps, scatter, kernel


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("C8_FUZZ_SEEDS", "24"))))
def test_random_case_matches_oracle(seed):
    from gpu_backend import GpuBackend
    model, params, kind, c, conn, eps, scatter, kernel = random_case(seed)
    et = ol.HEX8 if kind == "hex8" else ol.TET4
    orc = ol.Oracle(et, c, conn, model, params)
    gpu = GpuBackend(et, c, conn, model, params, scatter=scatter, kernel=kernel)
    # 1e-12 for every model and every quantity.  Two things are measured on their proper scale (parity_cases.py):
    # the Jacobian of a point whose state-to-Jacobian map is ill-conditioned (hyper_J2 at plastic onset) is compared
    # with the oracle's Jacobian AT THE SAME converged local state (the states themselves agree to 1e-15), and a
    # parameter-gradient component against the sum of the magnitudes of the products summed into it.
    tol = 1e-12
    AUDIT.ctx = "gpu %s %s %s" % (kind, kernel, scatter)
    check_forward(orc, gpu, c, model, eps, tol)
    check_residual(orc, gpu, c, eps, tol)
    if not (kind == "hex8" and kernel == "slot" and scatter == "gather"):  # that adjoint kernel cannot stage (refused)
        check_adjoint_chain(orc, gpu, c, model, eps, tol)


# ---- the model families added in round 2: 2-D plane strain / plane stress on tri3, Hosford / Barlat on tet4 and hex8 -------
def random_case_r2(seed):
    from meshes import jiggle_2d, tri_mesh
    from parity_cases import CASES_2D, CASES_LINE_SEARCH, CASES_PLANE_STRESS
    rng = np.random.default_rng(5000 + seed)
    pool = [(m, p, "2d") for m, p, _ in CASES_2D[1:]] + [(m, p, "2d") for m, p, _ in CASES_PLANE_STRESS[1:]] + \
           [(m, p, "3d") for m, p, _ in CASES_LINE_SEARCH]
    model, p0, fam = pool[rng.integers(len(pool))]
    p = np.array(p0, dtype=np.float64)
    scale = 1.0 + 0.2 * (rng.random(len(p)) - 0.5) * (np.abs(p) > 0)
    if model == "hypo_hill_plane_stress":
        scale[9:] = 1.0  # the material axes stay a rotation
    if model in ("small_hosford", "hypo_hosford", "hypo_barlat"):
        scale[3] = 1.0   # the exponent a
    p = p * scale
    p[1] = min(p[1], 0.4)
    eps = float(10 ** rng.uniform(-3.0, -2.2))
    if fam == "2d":
        nx, ny = (int(v) for v in rng.integers(3, 8, 2))
        c, conn, sets = tri_mesh(nx, ny, *(0.5 + rng.random(2)))
        c = jiggle_2d(c, sets, 0.02 + 0.05 * rng.random(), seed=seed)
        return model, list(p), ol.TRI3, c, conn, eps, ["colored", "atomic"][rng.integers(2)]
    kind = ["hex8", "tet4"][rng.integers(2)]
    n = tuple(int(v) for v in rng.integers(2, 4, 3))
    c, conn, sets = brick(n[0], n[1], n[2], *(0.5 + rng.random(3)))
    c = jiggle(c, sets, 0.02 + 0.06 * rng.random(), seed=seed)
    if kind == "tet4":
        tets = [[0, 1, 2, 6], [0, 2, 3, 6], [0, 3, 7, 6], [0, 7, 4, 6], [0, 4, 5, 6], [0, 5, 1, 6]]
        conn = np.concatenate([conn[:, t] for t in tets]).astype(np.int32)
    return model, list(p), (ol.HEX8 if kind == "hex8" else ol.TET4), c, conn, eps, ["colored", "atomic", None][rng.integers(3)]


def run_case_r2(factory, seed, tol=1e-12):
    from parity_cases import LOCAL_LINE_SEARCH
    model, params, et, c, conn, eps, scatter = random_case_r2(seed)
    ls = model in ("small_hosford", "hypo_hosford", "hypo_barlat")
    orc = ol.Oracle(et, c, conn, model, params)
    if ls:
        orc.set_local_line_search(*LOCAL_LINE_SEARCH)
    dut = factory(et, c, conn, model, params, scatter, LOCAL_LINE_SEARCH if ls else None)
    AUDIT.ctx = "%s elem%d %s" % (type(dut).__name__, et, scatter)
    check_forward(orc, dut, c, model, eps, tol)
    check_residual(orc, dut, c, eps, tol)
    check_adjoint_chain(orc, dut, c, model, eps, tol)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("C8_FUZZ_SEEDS_R2", "24"))))
def test_random_case_of_the_round2_families_matches_oracle(seed):
    from gpu_backend import GpuBackend

    def factory(et, c, conn, model, params, scatter, ls):
        return GpuBackend(et, c, conn, model, params, scatter=scatter, **({"line_search": ls} if ls else {}))

    run_case_r2(factory, seed)
