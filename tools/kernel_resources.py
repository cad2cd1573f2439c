"""Print VGPR / scratch / occupancy / LDS of every kernel in c8_kernels.hip (hipcc resource-usage remarks)."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "calibr8_amd", "csrc", "c8_kernels.hip")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
t = subprocess.run(cmd, capture_output=True, text=True).stderr
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for blk in t.split("Function Name: ")[1:]:
    name = blk.split()[0]
    if pat not in name:
        continue
    g = lambda k: re.search(k + r": (\d+)", blk).group(1)
    short = re.sub(r"^_ZN2c8\d+", "", name)[:70]
    print("%-70s VGPR %3s AGPR %3s scratch %4s occ %s LDS %s" % (
        short, g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
        g(r"LDS Size \[bytes/block\]")))
