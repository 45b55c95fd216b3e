#!/usr/bin/env python3
"""Per-kernel fingerprint of the ISA the product build generates (no GPU needed): hipcc -S of every csrc/*.hip with the
build's own flags, each kernel's instruction stream (comments stripped, branch labels renumbered per kernel) hashed.

    python scripts/isa_hash.py profiles/r06_isa_hashes.json            # write the table
    python scripts/isa_hash.py --diff profiles/r05_isa_hashes.json profiles/r06_isa_hashes.json

Round 6 used it to show that taking the experiment switches out of csrc/ changed no product kernel: r05 -> r06 differ in
the four reset kernels (the device-resident reset counter), the math test hook and three new kernels, nothing else."""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_fishing_amd import build  # noqa: E402


def table():
    out, procs = {}, []
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    for src in build.sources():
        asm = "/tmp/isa_hash.%d.%s.s" % (os.getpid(), os.path.basename(src))
        cmd = [build.hipcc_path()] + flags + build.TU_FLAGS.get(os.path.basename(src), []) + ["-S", "--cuda-device-only", src, "-o", asm]
        procs.append((asm, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
    for asm, p in procs:
        if p.wait() != 0:
            raise SystemExit("hipcc -S failed for %s" % asm)
        parts = re.split(r"\n(_ZN7fishing\w+):[^\n]*\n", open(asm).read())
        os.remove(asm)
        for i in range(1, len(parts), 2):
            body = parts[i + 1].split(".Lfunc_end")[0]
            ins = [re.sub(r"\s*;.*$", "", ln.strip()) for ln in body.splitlines()
                   if ln.startswith("\t") and not ln.strip().startswith((".", ";"))]
            ins = [re.sub(r"\.LBB\d+_(\d+)", r".LBB_\1", x) for x in ins]
            name = subprocess.run(["c++filt", parts[i]], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
            out[name] = {"instructions": len(ins), "sha1": hashlib.sha1("\n".join(ins).encode()).hexdigest()[:16]}
    return out


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--diff":
        a, b = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
        for k in sorted(set(a) | set(b)):
            if a.get(k) != b.get(k):
                print("%-100s %s -> %s" % (k[:100], a.get(k), b.get(k)))
        print("%d kernels before, %d after, %d differ" % (len(a), len(b), sum(1 for k in set(a) | set(b) if a.get(k) != b.get(k))))
    else:
        t = table()
        with open(sys.argv[1], "w") as f:
            json.dump(t, f, indent=0, sort_keys=True)
        print("%d kernels" % len(t))
