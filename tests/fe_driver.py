"""Small host-side primal/adjoint drivers used by the tests (TEST INFRASTRUCTURE).

They restate the callers on either side of the hot path so that the assembly
backends (the CPU oracle, or the HIP product through its C ABI) can be run to the
reference's end-to-end regression values:

  Newton loop + Armijo/cubic line search   primal.cpp:91-199, line_search.hpp:56-135
  Dirichlet row replacement                 dbcs.cpp:28-121
  traction surface integrals                tbcs.cpp:17-86
  adjoint step + history bookkeeping        adjoint.cpp:52-189, adjoint_objective.cpp:48-125

The sparse linear solves (Belos/Teko/MueLu in the reference, out of scope) are a
direct SciPy solve here.  A backend needs: nnodes, rowptr, colidx, new_linsys(),
new_state(), forward_jacobian(), and for the adjoint adjoint_jacobian(),
solve_adjoint_local(), qoi_gradient(), eval_qoi().
"""
import math

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

def neq_of(be):
    """equations per node of the two residuals: u has ndims (3, or 2 on tri3 meshes), p has 1"""
    return [getattr(be, "ndims", 3), 1]


def nres_of(be):
    """global residuals of the backend: 2 (`mechanics`: u, p) or 1 (`mechanics_plane_stress`: u only; the p arrays and
    the other three blocks exist in the test containers but stay untouched)"""
    return getattr(be, "nres", 2)


def rhs_of(be, ls):
    return np.concatenate(ls.b[:nres_of(be)])


def block_matrix(be, ls):
    n = be.nnodes
    NEQ = neq_of(be)
    if nres_of(be) == 1:
        return sp.csr_matrix((ls.A[0][0], be.colidx[0][0], be.rowptr[0][0]), shape=(n * NEQ[0], n * NEQ[0]))
    blocks = [[None, None], [None, None]]
    for i in range(2):
        for j in range(2):
            blocks[i][j] = sp.csr_matrix((ls.A[i][j], be.colidx[i][j], be.rowptr[i][j]),
                                         shape=(n * NEQ[i], n * NEQ[j]))
    return sp.bmat(blocks, format="csr")


class Dbc:
    """[resid_idx, eq, node list, value(x, y, z, t)]"""

    def __init__(self, resid, eq, nodes, fn):
        self.resid, self.eq, self.nodes, self.fn = resid, eq, list(nodes), fn


class Tbc:
    """[resid_idx, list of boundary triangles (node triples), traction(x, y, z, t) -> 3-vector]"""

    def __init__(self, resid, sides, fn):
        self.resid, self.sides, self.fn = resid, sides, fn


def apply_dbcs(be, ls, dbcs, x, coords, t, is_adjoint=False):
    """dbcs.cpp:28-121: keep the diagonal, zero the rest of the row in every block,
    R[row] = diag * (sol - v) (0 for the adjoint system)."""
    NEQ = neq_of(be)
    for bc in dbcs:
        i = bc.resid
        for node in bc.nodes:
            row = node * NEQ[i] + bc.eq
            sol = x[i][row]
            v = bc.fn(coords[node, 0], coords[node, 1], coords[node, 2], t)
            for j in range(nres_of(be)):
                rp, ci, vals = be.rowptr[i][j], be.colidx[i][j], ls.A[i][j]
                lo, hi = rp[row], rp[row + 1]
                if i == j:
                    diag = 0.0
                    for k in range(lo, hi):
                        if ci[k] == row:
                            diag = vals[k]
                        else:
                            vals[k] = 0.0
                    ls.b[i][row] = 0.0 if is_adjoint else diag * (sol - v)
                else:
                    vals[lo:hi] = 0.0


def apply_tbcs(ls, tbcs, coords, t):
    """tbcs.cpp:17-86 on tri3 boundary faces with the order-1 rule (centroid, N = 1/3); on a 2-D mesh the sides are
    edges (midpoint, N = 1/2, w dv = length) and a node has two equations."""
    for bc in tbcs:
        for tri in bc.sides:
            if len(tri) == 2:
                X = coords[list(tri)]
                length = np.linalg.norm(X[1] - X[0])
                xc = X.mean(axis=0)
                T = bc.fn(xc[0], xc[1], xc[2], t)
                for n in tri:
                    for d in range(2):
                        ls.b[bc.resid][n * 2 + d] -= T[d] * 0.5 * length
                continue
            X = coords[list(tri)]
            area = 0.5 * np.linalg.norm(np.cross(X[1] - X[0], X[2] - X[0]))
            xc = X.mean(axis=0)
            T = bc.fn(xc[0], xc[1], xc[2], t)
            for n in tri:
                for d in range(3):
                    ls.b[bc.resid][n * 3 + d] -= T[d] * (1.0 / 3.0) * 0.5 * (2.0 * area)


def _cubic_min(phi_0, dphi_0, a, phi, slope_a):
    d1 = dphi_0 + slope_a - 3.0 * (phi_0 - phi) / (0.0 - a)
    rad = d1 * d1 - dphi_0 * slope_a
    if rad < 0.0:
        return 0.5 * a
    d2 = math.sqrt(rad)
    den = slope_a - dphi_0 + 2.0 * d2
    if den == 0.0:
        return 0.5 * a
    return a - a * (slope_a + d2 - d1) / den


def line_search(phi_0, dphi_0, evaluate, c1=1e-4, bmin=0.5, bmax=0.9, max_evals=4):
    """line_search.hpp:85-135"""
    armijo = c1 * dphi_0
    alpha, best_alpha, best_phi, any_ok = 1.0, 1.0, float("inf"), False
    for _ in range(max_evals):
        ok, phi, slope = evaluate(alpha)
        if not ok:
            alpha *= 0.5
            continue
        any_ok = True
        if phi < best_phi:
            best_phi, best_alpha = phi, alpha
        if phi <= phi_0 + alpha * armijo:
            return alpha, True
        am = _cubic_min(phi_0, dphi_0, alpha, phi, slope)
        alpha = min(max(am, bmin * alpha), bmax * alpha)
    return best_alpha, any_ok


class Primal:
    """primal.cpp:31-209 for one part.  Keeps the per-step fields for the adjoint."""

    def __init__(self, be, coords, dbcs, tbcs=(), max_iters=15, abs_tol=1e-8, rel_tol=1e-8, step_size=1.0,
                 use_line_search=True):
        self.be, self.coords = be, np.asarray(coords, dtype=np.float64)
        self.dbcs, self.tbcs = dbcs, tbcs
        self.max_iters, self.abs_tol, self.rel_tol = max_iters, abs_tol, rel_tol
        self.step_size = step_size
        self.use_line_search = use_line_search
        n = be.nnodes
        self.u = [np.zeros(n * neq_of(be)[0])]
        self.p = [np.zeros(n)]
        self.xi = [be.new_state()]
        self.ls = be.new_linsys()
        self.newton_iters = []
        self.measured = None  # Calibration QoI: ([u_meas per step], [load_meas per step]), index 0 unused

    def _assemble(self, step, x, xi):
        be, ls = self.be, self.ls
        ls.zero()
        rc = be.forward_jacobian(x[0], x[1], self.u[step - 1], self.p[step - 1], self.xi[step - 1], xi, ls)
        if rc != 0:
            return rc
        t = step * self.step_size
        apply_tbcs(ls, self.tbcs, self.coords, t)
        apply_dbcs(be, ls, self.dbcs, x, self.coords, t)
        return 0

    def solve_at_step(self, step):
        be, ls = self.be, self.ls
        assert len(self.u) == step
        x = [self.u[step - 1].copy(), self.p[step - 1].copy()]
        xi = self.xi[step - 1].copy()
        n3 = be.nnodes * neq_of(be)[0]
        two = nres_of(be) == 2
        it, converged, r0 = 1, False, 1.0
        while it <= self.max_iters and not converged:
            if self._assemble(step, x, xi) != 0:
                raise RuntimeError("local solve failed at the base point")
            R = rhs_of(be, ls)
            rn = float(np.sqrt(sum(np.sum(ls.b[i] ** 2) for i in range(nres_of(be)))))
            if it == 1:
                r0 = rn
            if rn < self.abs_tol or rn / r0 < self.rel_tol:
                converged = True
                break
            A = block_matrix(be, ls)
            dx = spla.spsolve(A.tocsc(), -R)
            x[0] += dx[:n3]
            if two:
                x[1] += dx[n3:]
            if self.use_line_search:
                psi_0 = 0.5 * rn * rn
                dpsi_0 = -2.0 * psi_0
                state = {"applied": 1.0}
                xi_saved = xi.copy()  # primal.cpp:146-156: every trial warm-starts its local solves from the base state

                def evaluate(alpha):
                    xi[...] = xi_saved
                    x[0] += (alpha - state["applied"]) * dx[:n3]
                    if two:
                        x[1] += (alpha - state["applied"]) * dx[n3:]
                    state["applied"] = alpha
                    if self._assemble(step, x, xi) != 0:
                        return False, 0.0, 0.0
                    Ra = rhs_of(be, ls)
                    phi = 0.5 * float(Ra @ Ra)
                    slope = float(Ra @ (block_matrix(be, ls) @ dx))
                    return True, phi, slope

                alpha, ok = line_search(psi_0, dpsi_0, evaluate)
                if not ok:
                    raise RuntimeError("line search could not assemble at any trial step")
                x[0] += (alpha - state["applied"]) * dx[:n3]
                if two:
                    x[1] += (alpha - state["applied"]) * dx[n3:]
            it += 1
        if not converged:
            raise RuntimeError("Newton's method failed in %d iterations" % self.max_iters)
        self.newton_iters.append(it)
        self.u.append(x[0])
        self.p.append(x[1])
        self.xi.append(xi)

    def solve(self, nsteps):
        for s in range(1, nsteps + 1):
            self.solve_at_step(s)
        return self

    def begin_qoi_step(self, s):
        """measured data of the step + preprocess_qoi (adjoint_objective.cpp:33-35, :86-88) for QoIs that need it"""
        if getattr(self, "measured", None) is not None:
            self.be.set_measured(self.measured[0][s], self.measured[1][s])
            self.be.qoi_preprocess(self.u[s], self.p[s], self.u[s - 1], self.p[s - 1], self.xi[s - 1], self.xi[s])

    def qoi(self):
        """sum over steps of eval_qoi (adjoint_objective.cpp:36-37; main_primal.cpp sums steps too)."""
        J = 0.0
        for s in range(1, len(self.u)):
            self.begin_qoi_step(s)
            J += self.be.eval_qoi(self.u[s], self.p[s])
        return J


def adjoint_gradient(primal, nparams):
    """adjoint_objective.cpp:83-95 + adjoint.cpp:76-189: march backwards, return dJ/dp."""
    be = primal.be
    nsteps = len(primal.u) - 1
    nd = (neq_of(be)[0] + (1 if nres_of(be) == 2 else 0)) * be.nn
    g = np.zeros((be.nelems, be.npts, be.nloc))
    f = np.zeros((be.nelems, be.npts, nd))
    grad = np.zeros(nparams)
    n3 = be.nnodes * neq_of(be)[0]
    ls = be.new_linsys()
    for step in range(nsteps, 0, -1):
        u, p, xi = primal.u[step], primal.p[step], primal.xi[step]
        up, pp, xip = primal.u[step - 1], primal.p[step - 1], primal.xi[step - 1]
        primal.begin_qoi_step(step)
        ls.zero()
        be.adjoint_jacobian(u, p, up, pp, xip, xi, g, f, ls)
        z = [np.zeros(n3), np.zeros(be.nnodes)]
        apply_dbcs(be, ls, primal.dbcs, z, primal.coords, 0.0, is_adjoint=True)
        A = block_matrix(be, ls)
        zz = spla.spsolve(A.tocsc(), rhs_of(be, ls))
        z_u = np.ascontiguousarray(zz[:n3])
        z_p = np.ascontiguousarray(zz[n3:]) if nres_of(be) == 2 else np.zeros(be.nnodes)
        phi = np.zeros((be.nelems, be.npts, be.nloc))
        be.solve_adjoint_local(u, p, up, pp, xip, xi, z_u, z_p, phi, g, f)
        grad += be.qoi_gradient(u, p, up, pp, xip, xi, z_u, z_p, phi, nparams)
    return grad
