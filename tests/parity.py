"""Parity metrics used by every GPU-vs-oracle and emulator-vs-oracle test:
  residual: max |R - R_ref| / ||R_ref||_inf
  Jacobian: max over nonzeros of |A - A_ref| / ||row of A_ref||_inf   (entries that are sums of
            cancelling terms are judged against their row, not against themselves)
  state:    max |xi - xi_ref| / max(1, ||xi_ref||_inf)
"""
import numpy as np


def rel_vec(a, ref):
    s = np.abs(ref).max()
    return float(np.abs(a - ref).max() / (s if s > 0 else 1.0))


def rel_csr_rows(vals, ref, rowptr):
    worst = 0.0
    rp = np.asarray(rowptr)
    nrows = len(rp) - 1
    # row-wise inf-norm of the reference
    rownorm = np.maximum.reduceat(np.abs(ref), rp[:-1]) if len(ref) else np.zeros(nrows)
    lens = np.diff(rp)
    scale = np.repeat(np.where(rownorm > 0, rownorm, 1.0), lens)
    if len(ref):
        worst = float((np.abs(vals - ref) / scale).max())
    return worst


def compare_systems(be, ls, ls_ref):
    out = {}
    out["b_u"] = rel_vec(ls.b[0], ls_ref.b[0])
    out["b_p"] = rel_vec(ls.b[1], ls_ref.b[1])
    for i in range(2):
        for j in range(2):
            out["A%d%d" % (i, j)] = rel_csr_rows(ls.A[i][j], ls_ref.A[i][j], be.rowptr[i][j])
    return out
