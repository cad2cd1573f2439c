"""Static instruction histogram of one kernel of c8_kernels.hip (gfx950 ISA), split at s_memtime stamps if present.
usage: python tools/isa_histogram.py <substring of the mangled kernel name> [asm file]"""
import collections, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "gpurun_out", "c8_kernels.s")
if len(sys.argv) <= 2:
    os.makedirs(os.path.dirname(asm), exist_ok=True)
    flags = os.environ.get("C8_EXTRA_FLAGS", "").split()
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-S",
                    "--cuda-device-only", "-o", asm, os.path.join(root, "calibr8_amd", "csrc", "c8_kernels.hip")] + flags,
                   check=True, capture_output=True)
t = open(asm).read()
pat = sys.argv[1]
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", t, re.S | re.M):
    if pat not in m.group(1):
        continue
    ops = collections.Counter()
    for line in m.group(2).split("\n"):
        line = line.strip()
        if not line or line.startswith((".", ";")) or line.endswith(":"):
            continue
        ops[line.split()[0]] += 1
    print(m.group(1)[:90], "static instructions:", sum(ops.values()))
    cls = collections.Counter()
    for k, v in ops.items():
        c = "valu f64" if k.startswith("v_") and "f64" in k else "valu other" if k.startswith("v_") else \
            "lds" if k.startswith("ds_") else "salu" if k.startswith("s_") else "vmem"
        cls[c] += v
    print("  ", dict(cls))
    for k, v in ops.most_common(40):
        print("   %-28s %5d" % (k, v))
    # regions between s_memtime stamps (diagnostic -DC8_STAMPS build): static counts per class
    parts = re.split(r"s_memtime[^\n]*\n", m.group(2))
    if len(parts) > 1:
        print("   regions between stamps (static):")
        for i, part in enumerate(parts):
            c = collections.Counter()
            for line in part.split("\n"):
                line = line.strip()
                if not line or line.startswith((".", ";")) or line.endswith(":"):
                    continue
                k = line.split()[0]
                c["valu" if k.startswith("v_") else "lds" if k.startswith("ds_") else "salu" if k.startswith("s_") else "vmem"] += 1
                if k.startswith("v_div_scale"): c["div"] += 0.5
                if k.startswith("v_cndmask"): c["cndmask"] += 1
                if k.startswith("s_cbranch"): c["branch"] += 1
            print("     region %2d: %s" % (i, dict(c)))
