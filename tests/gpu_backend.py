"""Adapter that lets the test drivers (fe_driver.py) run on the HIP product through its C ABI:
numpy in/out, device tensors inside.  Used only by `-m gpu` tests."""
import numpy as np

from calibr8_amd import Assembler


class HostLinSys:
    def __init__(self, gb):
        self.A = [[np.zeros(gb.asm.nnz[i][j]) for j in range(2)] for i in range(2)]
        self.b = [np.zeros(gb.nnodes * gb.ndims), np.zeros(gb.nnodes)]

    def zero(self):
        for i in range(2):
            self.b[i][:] = 0.0
            for j in range(2):
                self.A[i][j][:] = 0.0


class GpuBackend:
    def __init__(self, elem_type, coords, conn, local_type, params, scatter=None, kernel="auto", **kw):
        self.asm = Assembler(elem_type, coords, conn, local_type, params, scatter=scatter, **kw)
        if not (kernel == "wave" and elem_type != 8):
            self.asm.set_kernel(kernel)
        a = self.asm
        self.nnodes, self.nelems, self.nn = a.nnodes, a.nelems, a.nn
        self.ndims = a.ndims
        self.npts, self.nloc = a.npts, a.nloc
        self.rowptr, self.colidx = a.rowptr, a.colidx
        self._ls = a.new_linsys()

    def new_linsys(self):
        return HostLinSys(self)

    def new_state(self):
        return self.asm.new_state().cpu().numpy()

    def set_active(self, es, idx):
        self.asm.set_active(es, idx)

    def set_params(self, p):
        self.asm.set_params(p)

    def _up(self, ls):
        d = self._ls
        for i in range(2):
            d.b[i].copy_(self.asm.dev(ls.b[i]))
            for j in range(2):
                d.A[i][j].copy_(self.asm.dev(ls.A[i][j]))
        return d

    def _down(self, d, ls):
        for i in range(2):
            ls.b[i][:] = d.b[i].cpu().numpy()
            for j in range(2):
                ls.A[i][j][:] = d.A[i][j].cpu().numpy()

    def forward_jacobian(self, u, p, u_prev, p_prev, xi_prev, xi, ls):
        a = self.asm
        d = self._up(ls)
        dxi = a.dev(xi)
        rc = a.forward_jacobian(a.dev(u), a.dev(p), a.dev(u_prev), a.dev(p_prev), a.dev(xi_prev), dxi, d)
        self._down(d, ls)
        xi[...] = dxi.cpu().numpy()
        return rc

    def global_residual(self, u, p, u_prev, p_prev, xi_prev, xi, ls):
        a = self.asm
        d = self._up(ls)
        rc = a.global_residual(a.dev(u), a.dev(p), a.dev(u_prev), a.dev(p_prev), a.dev(xi_prev), a.dev(xi), d)
        self._down(d, ls)
        return rc

    def adjoint_jacobian(self, u, p, u_prev, p_prev, xi_prev, xi, g, f, ls):
        a = self.asm
        d = self._up(ls)
        dg = a.dev(g)
        rc = a.adjoint_jacobian(a.dev(u), a.dev(p), a.dev(u_prev), a.dev(p_prev), a.dev(xi_prev), a.dev(xi), dg,
                                a.dev(f), d)
        self._down(d, ls)
        g[...] = dg.cpu().numpy()
        return rc

    def solve_adjoint_local(self, u, p, u_prev, p_prev, xi_prev, xi, z_u, z_p, phi, g, f):
        a = self.asm
        dphi, dg, df = a.dev(phi), a.dev(g), a.dev(f)
        rc = a.solve_adjoint_local(a.dev(u), a.dev(p), a.dev(u_prev), a.dev(p_prev), a.dev(xi_prev), a.dev(xi),
                                   a.dev(z_u), a.dev(z_p), dphi, dg, df)
        phi[...] = dphi.cpu().numpy()
        g[...] = dg.cpu().numpy()
        f[...] = df.cpu().numpy()
        return rc

    def qoi_gradient(self, u, p, u_prev, p_prev, xi_prev, xi, z_u, z_p, phi, nparams):
        a = self.asm
        grad = a.dev(np.zeros(nparams))
        a.qoi_gradient(a.dev(u), a.dev(p), a.dev(u_prev), a.dev(p_prev), a.dev(xi_prev), a.dev(xi), a.dev(z_u),
                       a.dev(z_p), a.dev(phi), grad)
        return grad.cpu().numpy()

    def eval_qoi(self, u, p):
        a = self.asm
        J = a.dev(np.zeros(1))
        st = getattr(self, "_qoi_state", None)
        if st is None:
            a.eval_qoi(a.dev(u), a.dev(p), J)
        else:  # calibration: the state of the step given to qoi_preprocess
            a.eval_qoi(a.dev(u), a.dev(p), J, xi_prev=st[4], xi=st[5], u_prev=st[2], p_prev=st[3])
        return float(J.cpu().numpy()[0])

    # ---- Calibration objective ----
    def set_calibration(self, faces, **kw):
        self.asm.set_qoi_calibration(faces, **kw)
        self.calibration = True

    def set_measured(self, u_meas, load_meas):
        self.asm.set_measured(self.asm.dev(np.ascontiguousarray(u_meas, dtype=np.float64)), load_meas)

    def qoi_preprocess(self, u, p, u_prev, p_prev, xi_prev, xi):
        a = self.asm
        self._qoi_state = [a.dev(v) for v in (u, p, u_prev, p_prev, xi_prev, xi)]
        return np.array(a.qoi_preprocess(*self._qoi_state))
