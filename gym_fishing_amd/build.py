"""Build recipe for libfishing_hip.so (gfx950 only, in-tree).

    python -m gym_fishing_amd.build        # (re)build if sources are newer

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting gym_fishing_amd/_lib/libfishing_hip.so is git-ignored but travels to
the GPU box with the gpurun snapshot.

-ffp-contract=off is part of the numerical contract: the reference's NumPy
arithmetic rounds every operation separately, so the kernels must not fuse
a*b+c into an FMA ("Bit-exactness vs NumPy").
"""
import glob
import json
import os
import re
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "_lib")
LIB_PATH = os.path.join(LIB_DIR, "libfishing_hip.so")
# Per-kernel register / LDS table of the library that was built (hipcc's own remarks, written next to the .so).  bench.py
# quotes a committed rocprofv3 --pmc figure only when the kernel it ran still has the resources of the kernel that
# was profiled (scripts/summarize_profile.py stores them in the record): a changed kernel reports no stale counters.
RESOURCES_PATH = os.path.join(LIB_DIR, "kernel_resources.json")
ARCH = "gfx950"

HIPCC_FLAGS = ["-O3", "--offload-arch=" + ARCH, "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
               "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# Per translation unit.  fishing_step.hip: no SLP vectorizer -- it splits the tile's 16-byte loads into 8-byte halves
# and sinks them next to their first use, behind the Philox block and (the return accumulator) behind the first
# s_waitcnt; without it the loads are issued together at the top of the tile, ahead of the scheduling fence, as the
# kernel is written (N = 2^21 with returns: 9.8 -> 8.7 us per step; profiles/r03_small_n/).
# ... and kernarg preload: the lean kernel's four leading pointer arguments arrive in SGPRs at wave launch (see the kernel).
# fishing_rollout.hip: the fused kernels are VALU-bound and still run 2.5-8 % faster without the SLP vectorizer (fused
# step 6.70 -> 6.40 us per step at N = 2^22, 1.25 -> 1.13 at 2^18; in-kernel-policy rollouts +2.5-6 %:
# profiles/r03_rollout_no_slp.jsonl) -- its packed f32 operations save fewer issue slots than its re-ordering costs.
TU_FLAGS = {"fishing_step.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=10"],
            "fishing_rollout.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [
        os.path.join(os.path.dirname(PKG), "include", "fishing_hip.h")]


def is_stale():
    # (a missing kernel_resources.json next to a valid prebuilt library only means "no PMC quote" to bench.py: not stale)
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(s) > t for s in _deps() if os.path.exists(s))


def hipcc_path():
    return shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


def build(force=False, verbose=False, extra_flags=(), out=None):
    """Compile every csrc/*.hip into one shared library.  Returns its path.
    `out` + `extra_flags` build an experimental variant next to the default library
    (scripts/build_variants.py: today the host-side UBSan build; the product always loads LIB_PATH)."""
    if out is not None:
        return _compile(out, verbose, extra_flags)
    if not force and not is_stale():
        return LIB_PATH
    return _compile(LIB_PATH, verbose, extra_flags)


def _compile(out_path, verbose, extra_flags):
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libfishing_hip.so")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    tmp = out_path + ".tmp.%d" % os.getpid()
    srcs = sources()
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + ["-Rpass-analysis=kernel-resource-usage"] + list(extra_flags)
    objs = ["%s.%d.o" % (tmp, i) for i in range(len(srcs))]
    # one hipcc per translation unit, side by side (the two kernel files take about as long as each other),
    # then one link step
    tu_flags = TU_FLAGS
    cmds = [[hipcc] + flags + tu_flags.get(os.path.basename(src), []) + ["-c", src, "-o", obj] for src, obj in zip(srcs, objs)]
    procs = []
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate()[0] for p in procs]
    try:
        for p, o in zip(procs, outs):
            if p.returncode != 0:
                raise RuntimeError("hipcc failed (%d):\n%s" % (p.returncode, o))
            if verbose:
                rest = "\n".join(ln for ln in o.splitlines() if "remark:" not in ln and "kernel-resource-usage" not in ln)
                if rest.strip():
                    print(rest)
        link = [hipcc] + HIPCC_FLAGS + list(extra_flags) + objs + ["-o", tmp]
        if verbose:
            print(" ".join(link), flush=True)
        proc = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if proc.returncode != 0:
            raise RuntimeError("hipcc link failed (%d):\n%s" % (proc.returncode, proc.stdout))
        os.replace(tmp, out_path)
        if out_path == LIB_PATH:
            with open(RESOURCES_PATH, "w") as f:
                json.dump(parse_resources("\n".join(outs)), f, indent=0, sort_keys=True)
    finally:
        for f in objs + [tmp]:
            if os.path.exists(f):
                os.remove(f)
    return out_path


def parse_resources(remarks):
    """hipcc -Rpass-analysis=kernel-resource-usage output -> {demangled kernel name: {vgpr, sgpr, lds, scratch, occupancy}}."""
    blocks = re.split(r"remark: [^\n]*Function Name: ", remarks)[1:]
    names = [b.split()[0] for b in blocks]
    dem = names
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        dem = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    except Exception:  # noqa: BLE001 - mangled names are still unique keys
        pass
    table = {}
    for blk, name in zip(blocks, dem):
        def g(key, blk=blk):
            m = re.search(re.escape(key) + r": (\d+)", blk)
            return int(m.group(1)) if m else None
        short = re.sub(r"^void ", "", name).split("(")[0]
        table[short] = {"vgpr": g("VGPRs"), "sgpr": g("TotalSGPRs"), "lds": g("LDS Size [bytes/block]"),
                        "scratch": g("ScratchSize [bytes/lane]"), "occupancy": g("Occupancy [waves/SIMD]")}
    return table


def kernel_resources(kernel=None):
    """The table written by the last build of LIB_PATH ({} when absent); `kernel` -> that kernel's entry or None."""
    try:
        with open(RESOURCES_PATH) as f:
            table = json.load(f)
    except Exception:  # noqa: BLE001
        table = {}
    return table if kernel is None else table.get(kernel)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
