#!/usr/bin/env python3
"""Static instruction mix of the kernels of one csrc/*.hip translation unit (hipcc -S, no GPU):

    python scripts/isa_mix.py fishing_step 'lean<float, 4, 258>' ['lean<float, 1, 2>' ...]

Counts VALU / SALU / VMEM / LDS instructions of each kernel whose demangled name contains a pattern, and its most
frequent VALU opcodes.  Straight-line kernels (one tile per workgroup), so static counts ~ executed counts."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_fishing_amd.build import TU_FLAGS  # noqa: E402  (the per-unit flags of the product build: -fno-slp-vectorize, kernarg preload)


def mixes(tu, extra=()):
    src = os.path.join(ROOT, "gym_fishing_amd", "csrc", tu + ".hip")
    asm = "/tmp/%s.%d.s" % (tu, os.getpid())
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-fno-gpu-rdc",
                    "-S", "--cuda-device-only", src, "-o", asm] + TU_FLAGS.get(tu + ".hip", []) + list(extra), check=True, stderr=subprocess.DEVNULL)
    txt = open(asm).read()
    os.remove(asm)
    out = {}
    parts = re.split(r"\n(_ZN7fishing\w+):[^\n]*\n", txt)
    for i in range(1, len(parts), 2):
        body = parts[i + 1].split(".Lfunc_end")[0]
        dem = subprocess.run(["c++filt", parts[i]], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
        ins = [ln.split()[0] for ln in body.splitlines() if ln.startswith("\t") and not ln.strip().startswith((".", ";"))]
        c = collections.Counter("valu" if x.startswith("v_") else "salu" if x.startswith("s_") else
                                "vmem" if x.startswith(("global_", "buffer_", "flat_")) else "lds" if x.startswith("ds_") else "other"
                                for x in ins)
        out[dem] = (c, collections.Counter(x for x in ins if x.startswith("v_")))
    return out


if __name__ == "__main__":
    m = mixes(sys.argv[1])
    for pat in sys.argv[2:]:
        for k in sorted(m):
            if pat in k:
                c, ops = m[k]
                print(k, dict(c))
                print("    ", ops.most_common(16))
