#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one csrc/*.hip translation unit (cross-compiles, no GPU):

    python scripts/kernel_resources.py fishing_step [-D...]

hipcc -Rpass-analysis=kernel-resource-usage, demangled with c++filt."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table(tu, extra=()):
    src = os.path.join(ROOT, "gym_fishing_amd", "csrc", tu + ".hip")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-fno-gpu-rdc",
           "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + list(extra)
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows = []
    for blk in re.split(r"remark: Function Name: ", txt)[1:]:
        name = blk.split()[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"^void ", "", dem).split("(")[0]
        g = lambda k: int(re.search(re.escape(k) + r": (\d+)", blk).group(1))  # noqa: E731
        rows.append(dict(kernel=dem, vgpr=g("VGPRs"), sgpr=g("TotalSGPRs"), occupancy=g("Occupancy [waves/SIMD]"),
                         scratch=g("ScratchSize [bytes/lane]"), sgpr_spill=g("SGPRs Spill"), vgpr_spill=g("VGPRs Spill"),
                         lds=g("LDS Size [bytes/block]")))
    return rows


if __name__ == "__main__":
    rows = table(sys.argv[1], sys.argv[2:])
    print("%-70s %5s %5s %4s %7s %6s %6s %5s" % ("kernel", "vgpr", "sgpr", "occ", "scratch", "sspill", "vspill", "lds"))
    for r in sorted(rows, key=lambda r: r["kernel"]):
        print("%-70s %5d %5d %4d %7d %6d %6d %5d" % (r["kernel"], r["vgpr"], r["sgpr"], r["occupancy"], r["scratch"],
                                                      r["sgpr_spill"], r["vgpr_spill"], r["lds"]))
    print("%d kernels; %d with spills" % (len(rows), sum(1 for r in rows if r["sgpr_spill"] or r["vgpr_spill"] or r["scratch"])))
