"""TEST INFRASTRUCTURE: the halo exchanges of libc8.so replayed in numpy from its host index tables (c8_halo_table),
with torch.distributed (gloo) as the transport -- so that the CPU suite checks the tables c8_halo_build produces, the
message layout and the reproducible unpack order without a device.  On a GPU the same tables drive the HIP pack / unpack
kernels of csrc/c8_halo.hip (tests/test_gpu_distributed.py)."""
import numpy as np
import torch

MASK = (1 << 56) - 1


class HaloReplay:
    def __init__(self, halo, dist, world):
        self.dist, self.world = dist, world
        self.t = {base: [halo.table(base + k) for k in range(6 if base < 20 else 4)] for base in (0, 10, 20)}

    @staticmethod
    def _pick(segs, codes):
        out = np.empty(len(codes))
        seg, off = codes >> 56, codes & MASK
        for s in np.unique(seg):
            m = seg == s
            out[m] = segs[int(s)][off[m]]
        return out

    def _exchange(self, sbuf, sc, rc):
        r = torch.empty(int(rc.sum()), dtype=torch.float64)
        if self.world > 1:
            self.dist.all_to_all_single(r, torch.from_numpy(sbuf), [int(v) for v in rc], [int(v) for v in sc])
        return r.numpy()

    def gather(self, segs, b_only=False):
        """segs = [A00, A01, A10, A11, b0, b1] numpy arrays, updated in place (k_pack, exchange, k_unpack_add)."""
        send_idx, sc, rc, dst, src_ptr, src = self.t[10 if b_only else 0]
        rbuf = self._exchange(self._pick(segs, send_idx), sc, rc)
        v = np.zeros(len(dst))
        n = np.diff(src_ptr)
        for k in range(int(n.max()) if len(n) else 0):  # contribution k of every destination: ascending source rank
            m = n > k
            v[m] += rbuf[src[src_ptr[:-1][m] + k]]
        seg, off = dst >> 56, dst & MASK
        for s in np.unique(seg):
            m = seg == s
            segs[int(s)][off[m]] += v[m]

    def scatter_x(self, x):
        """x = [u, p]: owner values to the ghost and phantom copies (k_pack, exchange, k_unpack_store)."""
        segs = [None, None, None, None, x[0], x[1]]
        send_idx, sc, rc, dst = self.t[20]
        rbuf = self._exchange(self._pick(segs, send_idx), sc, rc)
        seg, off = dst >> 56, dst & MASK
        for s in np.unique(seg):
            m = seg == s
            segs[int(s)][off[m]] = rbuf[m]
