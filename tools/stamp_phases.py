"""Diagnostic only (build with C8_STAMPS=1 python -m calibr8_amd.build): per-phase s_memtime shares of the wave kernel.
The stamps are written over a slice of the p-residual output, so the outputs of this build are invalid."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from calibr8_amd import Assembler, brick_mesh
from meshes import prescribed_fields

from parity_cases import HILL, HJ2, J2
model = os.environ.get("MODEL", "small_J2")  # MODEL=hyper_J2 | small_hill | hypo_hill: the iterated kernel of that model
n = 100
coords, conn = brick_mesh(n, n, n)
asm = Assembler(8, coords, conn, model, {"small_J2": J2, "hyper_J2": HJ2, "small_hill": HILL, "hypo_hill": HILL}[model],
                scatter=sys.argv[1] if len(sys.argv) > 1 else "atomic")
asm.set_kernel("wave_ad")  # the stamps sit in the iterated kernel
u_h, p_h = prescribed_fields(coords, 0.004, ramp=True)
u, p = asm.dev(u_h), asm.dev(p_h)
u0, p0 = torch.zeros_like(u), torch.zeros_like(p)
xi0, xi = asm.new_state(), asm.new_state()
ls = asm.new_linsys()
for _ in range(3):
    asm.forward_jacobian(u, p, u0, p0, xi0, xi, ls)
torch.cuda.synchronize()
import ctypes as C
buf = np.zeros(4096 * 16, dtype=np.uint64)
asm.L.c8_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
asm.L.c8_debug_stamps(asm.h, buf.ctypes.data_as(C.c_void_p))
raw = buf.reshape(4096, 16)[:, [0, 10, 11, 1, 2, 3, 4, 5, 6, 7, 8, 9]].astype(np.int64)  # stamps in program order
d = np.diff(raw, axis=1)
names = ["loads (conn, then nodal)", "shape tables", "interpolation", "newton", "inverse", "D pass0", "P pass0", "D pass1", "P pass1", "(loop end)", "scatter"]
plastic = (xi[::244][:4096, :, -1] > 0).any(dim=1).cpu().numpy()
ok = raw[:, 11] > raw[:, 0]
raw, d, plastic = raw[ok], d[ok], plastic[ok]
for label, sel in (("all", np.ones(len(raw), bool)), ("elastic elems", ~plastic), ("plastic elems", plastic)):
    if sel.sum() == 0:
        continue
    tot = (raw[sel, 11] - raw[sel, 0]).mean()
    print("%s (%d): total %.0f cycles" % (label, sel.sum(), tot))
    for k, nm in enumerate(names):
        print("   %-20s %8.0f  %5.1f %%" % (nm, d[sel, k].mean(), 100 * d[sel, k].mean() / tot))
