"""c8_lbfgs_minimize (host C++ in libc8.so): the outer optimiser of the calibration loop, against SciPy's L-BFGS-B
(what the reference's Python package uses) on bounded test problems.  No GPU needed."""
import numpy as np
import scipy.optimize as so

from calibr8_amd.inverse import lbfgs_minimize


def rosen(x):
    return so.rosen(x), so.rosen_der(x)


def test_unconstrained_rosenbrock_reaches_the_minimum():
    x, info = lbfgs_minimize(rosen, [-1.2, 1.0, -0.5, 0.8], max_iters=200, grad_tol=1e-10, max_ls_evals=20)
    assert info["status"] == "gradient tolerance", info
    assert np.abs(x - 1.0).max() < 1e-8


def test_bounded_problem_matches_scipy_lbfgsb():
    lo, hi = np.array([-1.0, -1.0, -1.0]), np.array([0.5, 1.0, 0.2])
    x0 = np.array([-0.8, -0.5, 0.1])
    ref = so.minimize(rosen, x0, jac=True, method="L-BFGS-B", bounds=list(zip(lo, hi)), options={"ftol": 1e-16, "gtol": 1e-12})
    x, info = lbfgs_minimize(rosen, x0, lo, hi, max_iters=500, grad_tol=1e-10, max_ls_evals=30)
    assert np.abs(x - ref.x).max() < 1e-6, (x, ref.x, info)
    assert abs(info["f"] - ref.fun) < 1e-10
    assert (x >= lo - 1e-15).all() and (x <= hi + 1e-15).all()
    assert np.isclose(x[0], 0.5)  # the first bound is active at the solution


def test_controls_and_failed_evaluations():
    # iteration limit
    x, info = lbfgs_minimize(rosen, [-1.2, 1.0], max_iters=3, max_ls_evals=20)
    assert info["iters"] == 3 and info["status"] == "iteration limit"
    # the objective refuses to evaluate outside a disc: the line search backs off and stays inside
    def guarded(x):
        if x @ x > 4.0:
            return None
        return rosen(x)
    x, info = lbfgs_minimize(guarded, [-1.2, 1.0], max_iters=300, grad_tol=1e-9, max_ls_evals=30)
    assert np.abs(x - 1.0).max() < 1e-6, (x, info)
    # a quadratic is solved to the step tolerance
    A = np.diag([1.0, 10.0, 100.0])
    x, info = lbfgs_minimize(lambda v: (0.5 * v @ A @ v, A @ v), [1.0, 1.0, 1.0], max_iters=100, grad_tol=1e-14, max_ls_evals=20)
    assert np.abs(x).max() < 1e-10
