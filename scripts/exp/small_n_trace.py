"""Group a rocprofv3 --kernel-trace CSV by (kernel, grid): median / mean dispatch duration in us.

    python scripts/exp/small_n_trace.py <dir with *_kernel_trace.csv> [name substring ...] > out.jsonl

Used with scripts/exp/small_n_shapes (kernel names carry body / threads / envs per thread; the grid gives N)."""
import csv
import glob
import json
import os
import re
import statistics
import sys


def main():
    d = sys.argv[1]
    subs = sys.argv[2:]
    hits = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not hits:
        sys.exit("no kernel_trace.csv under " + d)
    path = max(hits, key=os.path.getmtime)
    groups = {}
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if subs and not any(s in name for s in subs):
            continue
        grid = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
        wg = int(r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or 0)
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        groups.setdefault((name, grid, wg), []).append(dur)
    for (name, grid, wg), v in sorted(groups.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        v2 = v[len(v) // 5:]          # drop the warm-up launches of each case
        rec = {"kernel": name, "grid_threads": grid, "workgroup": wg, "calls": len(v), "us_median": round(statistics.median(v2), 3),
               "us_mean": round(statistics.fmean(v2), 3), "us_min": round(min(v2), 3)}
        m = re.match(r"shape_kernel<(\d+), (\d+), (\d+), (true|false)>", name)
        if m:
            body, thr, ept, ret = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4) == "true"
            rec.update(body=["empty", "copy", "step"][body], threads=thr, ept=ept, ret=ret, n_envs=grid * ept)
        print(json.dumps(rec))


if __name__ == "__main__":
    main()
