#!/usr/bin/env python3
"""Where the waits sit in the lean step kernels (hipcc -S of csrc/fishing_step.hip with the product's flags, no GPU):

    python scripts/isa_shape.py ['lean<float, 1, 12294, 4>' ...]        (no pattern: every exact lean kernel)

Per kernel: the index of the first global load, how many `s_waitcnt lgkmcnt` a wave passes before it (scalar-load round
trips in front of the tile's loads: round 4's run-time walk flag had made it four), the index of the first `s_waitcnt vmcnt`
and how many of the noise generator's 32 x 32 -> 64-bit multiplies were issued before it (LLVM sinks the generator behind
that wait wherever control flow follows it: fishing-v4's derivation, Beverton-Holt / Myers / May), scalar loads behind
the first vector wait.  tests/test_isa_shape.py holds the product to these shapes."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_fishing_amd.build import HIPCC_FLAGS, TU_FLAGS, hipcc_path  # noqa: E402


def listing(tu="fishing_step", extra=()):
    """{demangled kernel name: [instruction text, ...]} of one translation unit."""
    src = os.path.join(ROOT, "gym_fishing_amd", "csrc", tu + ".hip")
    asm = "/tmp/%s.shape.%d.s" % (tu, os.getpid())
    flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.run([hipcc_path()] + flags + ["-S", "--cuda-device-only", src, "-o", asm] + TU_FLAGS.get(tu + ".hip", []) + list(extra),
                   check=True, stderr=subprocess.DEVNULL)
    txt = open(asm).read()
    os.remove(asm)
    parts = re.split(r"\n(_ZN7fishing\w+):[^\n]*\n", txt)
    names = subprocess.run(["c++filt"] + parts[1::2], capture_output=True, text=True).stdout.strip().splitlines()
    out = {}
    for name, body in zip(names, parts[2::2]):
        body = body.split(".Lfunc_end")[0]
        out[name.split("(")[0].replace("void ", "")] = [ln.strip() for ln in body.splitlines()
                                                        if ln.startswith("\t") and not ln.strip().startswith((".", ";"))]
    return out


def shape(ins):
    first_load = next((i for i, t in enumerate(ins) if t.startswith("global_load")), None)
    first_vm = next((i for i, t in enumerate(ins) if t.startswith("s_waitcnt") and "vmcnt" in t), None)
    # (instructions 0-3 are the kernarg-preload prologue firmware without the feature runs: not on the path)
    lg = [i for i, t in enumerate(ins[:first_load or 0]) if i > 3 and t.startswith("s_waitcnt") and "lgkmcnt" in t]
    mul = [i for i, t in enumerate(ins) if t.startswith(("v_mad_u64_u32", "v_mul_hi_u32"))]
    return {"instructions": len(ins), "first_global_load": first_load, "lgkm_waits_before_first_load": len(lg),
            "first_vmcnt_wait": first_vm, "wide_multiplies_before_first_vmcnt_wait": sum(1 for i in mul if first_vm is None or i < first_vm),
            "wide_multiplies": len(mul),
            "s_loads_behind_first_vmcnt_wait": sum(1 for i, t in enumerate(ins) if first_vm is not None and i > first_vm and t.startswith("s_load"))}


def is_exact_lean(name):
    m = re.match(r"fishing::step_kernel_lean<(float|double), (\d+), (\d+), (\d+)>$", name)
    return bool(m) and not int(m.group(3)) & 1024       # (feat::OPT = 1 << 10: the catch-alls)


if __name__ == "__main__":
    ks = listing()
    pats = sys.argv[1:]
    for k in sorted(ks):
        if (any(p in k for p in pats) if pats else is_exact_lean(k)):
            print(k, shape(ks[k]))
