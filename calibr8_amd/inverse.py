"""Outer calibration loop: Adjoint_Objective (adjoint_objective.cpp:27-118) on the
canonical variables of objective.cpp:41-61 / :125-137, minimised by the library's bound-constrained L-BFGS
(c8_lbfgs_minimize, the stand-in for ROL's line-search step with an L-BFGS secant, main_inverse.cpp:21-28,
:83-120).  Host control flow only: every primal solve, adjoint step and objective evaluation runs through the
C ABI on the device."""
import ctypes as C

import numpy as np

from . import lib as _l

STATUS = {0: "iteration limit", 1: "gradient tolerance", 2: "step tolerance", 3: "line search failed"}


def lbfgs_minimize(fun, x0, lo=None, hi=None, max_iters=20, grad_tol=1e-12, step_tol=1e-12, max_ls_evals=5, memory=20):
    """fun(x) -> (f, grad) or None when the objective cannot be evaluated at x.  Returns (x, info dict).
    Defaults are the reference's ("iteration limit", "gradient tolerance", "step tolerance",
    "max line search evals", Maximum Storage 20)."""
    L = _l.load_library()
    x = np.ascontiguousarray(x0, dtype=np.float64).copy()
    n = len(x)
    dp = C.POINTER(C.c_double)
    lo_a = None if lo is None else np.ascontiguousarray(np.broadcast_to(lo, n), dtype=np.float64)
    hi_a = None if hi is None else np.ascontiguousarray(np.broadcast_to(hi, n), dtype=np.float64)
    err = []

    def cb(_user, nn, xp, fp, gp):
        try:
            r = fun(np.ctypeslib.as_array(xp, shape=(nn,)).copy())
            if r is None:
                return 1
            f, g = r
            fp[0] = float(f)
            np.ctypeslib.as_array(gp, shape=(nn,))[:] = g
            return 0
        except Exception as e:  # an exception must not unwind through the C frames
            err.append(e)
            return 1

    fn = _l.OBJECTIVE_FN(cb)
    opts = _l.LbfgsOpts(max_iters, grad_tol, step_tol, max_ls_evals, memory)
    res = _l.LbfgsResult()
    rc = L.c8_lbfgs_minimize(n, x.ctypes.data_as(dp), None if lo_a is None else lo_a.ctypes.data_as(dp),
                             None if hi_a is None else hi_a.ctypes.data_as(dp), fn, None, C.byref(opts), C.byref(res))
    if rc != 0:
        raise err[0] if err else RuntimeError("c8_lbfgs_minimize: the objective could not be evaluated at the starting point")
    return x, {"iters": res.iters, "evals": res.evals, "status": STATUS[res.status], "f": res.f,
               "projected_gradient_norm": res.projected_gradient_norm, "last_exception": err[-1] if err else None}


class InverseProblem:
    """Adjoint_Objective: value = sum over steps of eval_qoi after a primal solve with the trial parameters, gradient =
    the adjoint march, both in canonical variables with bound scaling to [-1, 1] (objective.cpp:41-61, :125-137).
    `make_primal(params)` returns a solved PrimalDriver for the full physical parameter vector of element set 0 (with
    measured data attached when the objective needs it).  On a multi-part mesh every rank runs the same loop on its part
    (`make_primal` builds the part's driver with the distributed linear solve) and `comm` (distributed.Comm) sums the
    objective and the gradient over the parts (PCU_Add_Double(J), adjoint_objective.cpp:39,99; PCU_Add_Doubles(grad),
    :109): all ranks then take the same L-BFGS steps."""

    def __init__(self, make_primal, base_params, active, bounds, comm=None):
        self.comm = comm
        # one forward problem, or several that share the parameters (the "problems" of the reference's input,
        # objective.cpp:16-39: J and the gradient are summed over them)
        self.problems = list(make_primal) if isinstance(make_primal, (list, tuple)) else [make_primal]
        self.make_primal = self.problems[0]
        self.base = np.array(base_params, dtype=np.float64)
        self.active = [int(a) for a in active]
        b = np.asarray(bounds, dtype=np.float64)
        self.lo, self.hi = np.ascontiguousarray(b[:, 0]), np.ascontiguousarray(b[:, 1])
        self.kind = np.full(len(self.active), _l.C8_SCALE_BOUNDS, dtype=np.int32)
        self.history = []

    def _tr(self, v, from_canonical):
        L = _l.load_library()
        v = np.ascontiguousarray(v, dtype=np.float64)
        out = np.zeros_like(v)
        _l.check(L.c8_transform_params(len(v), v.ctypes.data_as(_l.dp), self.kind.ctypes.data_as(_l.i32p),
                                       self.lo.ctypes.data_as(_l.dp), self.hi.ctypes.data_as(_l.dp), int(from_canonical),
                                       out.ctypes.data_as(_l.dp)))
        return out

    def to_canonical(self, physical_active):
        return self._tr(physical_active, False)

    def to_physical(self, canonical):
        return self._tr(canonical, True)

    def value_and_gradient(self, canonical):
        from .primal import adjoint_gradient
        L = _l.load_library()
        phys = self.to_physical(canonical)
        params = self.base.copy()
        params[self.active] = phys
        J, g = 0.0, np.zeros(len(self.active))
        for make in self.problems:
            try:
                pr = make(params)
            except (RuntimeError, _l.C8Error):
                pr = None  # the forward problem failed at these parameters (adjoint_objective.cpp lets ROL back off)
            if self.comm is not None:  # every part backs off together
                if self.comm.allreduce(np.array([0.0 if pr is not None else 1.0]))[0] > 0.0:
                    return None
            elif pr is None:
                return None
            pr.asm.set_active(0, self.active)
            J += pr.qoi()
            g += np.ascontiguousarray(adjoint_gradient(pr, len(self.active)))
            del pr
        if self.comm is not None:
            red = self.comm.allreduce(np.concatenate([[J], g]))
            J, g = float(red[0]), np.ascontiguousarray(red[1:])
        canon = np.ascontiguousarray(canonical, dtype=np.float64)
        gc = np.zeros_like(g)
        _l.check(L.c8_transform_gradient(len(g), g.ctypes.data_as(_l.dp), canon.ctypes.data_as(_l.dp),
                                         self.kind.ctypes.data_as(_l.i32p), self.lo.ctypes.data_as(_l.dp),
                                         self.hi.ctypes.data_as(_l.dp), gc.ctypes.data_as(_l.dp)))
        self.history.append((phys.copy(), float(J)))
        return J, gc

    def solve(self, initial_active, **opts):
        x0 = self.to_canonical(initial_active)
        x, info = lbfgs_minimize(self.value_and_gradient, x0, -1.0, 1.0, **opts)
        return self.to_physical(x), info


class FEMUProblem(InverseProblem):
    """FEMU_Objective (femu_objective.cpp:13-36): the value is the same sum over problems and steps of eval_qoi after the
    primal solves, kept for the last parameter vector (`param_diff`, objective.cpp:139-151); there is no adjoint -- the
    reference leaves the gradient to the optimiser's finite differences, restated here as forward differences of the
    value in the canonical variables (step `fd_step` times max(1, |x_i|), stepping inwards at the upper bound)."""

    def __init__(self, make_primal, base_params, active, bounds, comm=None, fd_step=1e-6):
        super().__init__(make_primal, base_params, active, bounds, comm)
        self.fd_step = float(fd_step)
        self._last = (None, None)

    def value(self, canonical):
        canonical = np.ascontiguousarray(canonical, dtype=np.float64)
        if self._last[0] is not None and np.array_equal(self._last[0], canonical):
            return self._last[1]
        params = self.base.copy()
        params[self.active] = self.to_physical(canonical)
        J = 0.0
        for make in self.problems:
            try:
                pr = make(params)
            except (RuntimeError, _l.C8Error):
                pr = None
            if self.comm is not None:
                if self.comm.allreduce(np.array([0.0 if pr is not None else 1.0]))[0] > 0.0:
                    return None
            elif pr is None:
                return None
            J += pr.qoi()
            del pr
        if self.comm is not None:
            J = float(self.comm.allreduce(np.array([J]))[0])
        self._last = (canonical.copy(), J)
        return J

    def value_and_gradient(self, canonical):
        canonical = np.ascontiguousarray(canonical, dtype=np.float64)
        J = self.value(canonical)
        if J is None:
            return None
        g = np.zeros_like(canonical)
        for i in range(len(canonical)):
            h = self.fd_step * max(1.0, abs(canonical[i]))
            if canonical[i] + h > 1.0:
                h = -h
            x = canonical.copy()
            x[i] += h
            Jh = self.value(x)
            if Jh is None:
                return None
            g[i] = (Jh - J) / h
        self._last = (canonical.copy(), J)
        self.history.append((self.to_physical(canonical), float(J)))
        return J, g
