"""Diagnostic only (build with C8_STAMPS=1 python -m calibr8_amd.build): s_memtime shares of the phases of the row-per-node
kernel (c8_assemble_node.hpp) over 4096 sampled nodes of the bench workload."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from calibr8_amd import Assembler, brick_mesh
from meshes import prescribed_fields

n = 100
coords, conn = brick_mesh(n, n, n)
asm = Assembler(8, coords, conn, "small_J2", [1000.0, 0.25, 100.0, 2.0, 0.0, 0.0], scatter="gather")
asm.set_kernel("node")
u_h, p_h = prescribed_fields(coords, 0.004, ramp=True)
u, p = asm.dev(u_h), asm.dev(p_h)
u0, p0 = torch.zeros_like(u), torch.zeros_like(p)
xi0, xi = asm.new_state(), asm.new_state()
ls = asm.new_linsys()
for _ in range(3):
    asm.forward_jacobian(u, p, u0, p0, xi0, xi, ls)
torch.cuda.synchronize()
buf = np.zeros(4096 * 16, dtype=np.uint64)
asm.L.c8_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
asm.L.c8_debug_stamps(asm.h, buf.ctypes.data_as(C.c_void_p))
order = [6, 0, 7, 8, 1, 2, 3, 4, 5]
names = ["scalar loads (node -> graph row, element list)", "A: element list, connectivity, nodal values, interpolation",
         "A: closed form", "A: row record, LDS", "B: blocks over 8 points", "fetch of the old CSR values issued, zero acc", "C: blocks into the accumulator",
         "D: rows out"]
raw = buf.reshape(4096, 16)[:, order].astype(np.int64)
ok = (raw[:, -1] > raw[:, 0]) & (raw[:, 0] > 0)
raw = raw[ok]
d = np.diff(raw, axis=1)
tot = (raw[:, -1] - raw[:, 0]).mean()
print("nodes sampled %d, total %.0f cycles per node (median %.0f)" % (len(raw), tot, np.median(raw[:, -1] - raw[:, 0])))
for k, nm in enumerate(names):
    print("   %-70s %8.0f  %5.1f %%   (median %.0f)" % (nm, d[:, k].mean(), 100 * d[:, k].mean() / tot, np.median(d[:, k])))
