#!/usr/bin/env python3
"""profiles/pmc_latest.json from the committed per-workload summaries (profiles/<prefix>_step_*_summary.json): one PMC traffic record
per (kernel, N) with the kernel's compile-time resources as the build in this tree reports them -- run it right after copying a
profile session's summaries into profiles/, BEFORE changing the kernels (bench.py quotes a record as roofline.traffic only while the
built kernel still has these resources).  A session that spans several gpurun calls writes one pmc_latest.json per call; this
rebuilds the whole table.

    python scripts/rebuild_pmc_latest.py r05
"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_fishing_amd import build  # noqa: E402


def main(prefix):
    recs = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "%s_step_*_summary.json" % prefix))):
        s = json.load(open(f))
        if "hbm_bytes_per_launch" not in s:
            continue
        res = build.kernel_resources(s["kernel"])
        if res is None:
            raise SystemExit("%s: the built library has no kernel %s" % (os.path.basename(f), s["kernel"]))
        recs.append({"kernel": s["kernel"], "n_envs": s["n_envs"], "hbm_bytes_per_launch": s["hbm_bytes_per_launch"],
                     "kernel_resources": res,
                     "source": os.path.basename(f) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                                                     "FETCH_SIZE x2 gfx950 correction)"})
    with open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w") as out:
        json.dump(recs, out, indent=1)
    print("%d records" % len(recs))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r05")
