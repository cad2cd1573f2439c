"""SHA-256 of A and b after K1 (three calls) and K3 on a jiggled 40^3 brick: run with two builds of libc8.so to compare them bit for bit
(tools/ab_libs.sh style: copy a build over calibr8_amd/libc8.so, run, compare the printed hashes)."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from calibr8_amd import Assembler, brick_mesh
from meshes import prescribed_fields, jiggle, brick
n = 40
coords, conn = brick_mesh(n, n, n)
import numpy as np
rng = np.random.default_rng(1); coords = coords + 0.2 / n * (rng.random(coords.shape) - 0.5)
asm = Assembler(8, coords, conn, "small_J2", [1000.0, 0.25, 100.0, 2.0, 0.0, 0.0], scatter="gather")
u_h, p_h = prescribed_fields(coords, 0.004, ramp=True, perturb=5e-2)
u, p = asm.dev(u_h), asm.dev(p_h)
z, zp = torch.zeros_like(u), torch.zeros_like(p)
hs = []
for rep in range(3):
    xi0, xi, ls = asm.new_state(), asm.new_state(), asm.new_linsys()
    asm.forward_jacobian(u, p, z, zp, xi0, xi, ls)
    hs.append(hashlib.sha256(ls.flat.cpu().numpy().tobytes()).hexdigest()[:16])
g = torch.randn(asm.nelems, asm.npts, asm.nloc, dtype=torch.float64, device=asm.device, generator=torch.Generator(device=asm.device).manual_seed(1)) * 1e-3
f = torch.randn(asm.nelems, asm.npts, asm.ndofs, dtype=torch.float64, device=asm.device, generator=torch.Generator(device=asm.device).manual_seed(2)) * 1e-3
ls = asm.new_linsys()
asm.adjoint_jacobian(u, p, z, zp, xi0, xi, g.clone(), f, ls)
hs.append(hashlib.sha256(ls.flat.cpu().numpy().tobytes()).hexdigest()[:16])
print(sys.argv[1] if len(sys.argv) > 1 else "", hs)
