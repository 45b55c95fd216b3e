#!/usr/bin/env python3
"""Turn the raw rocprofv3 CSVs of scripts/profile_bench.sh into the committed summaries:

    python scripts/summarize_profile.py gpurun_out/<tag> profiles/<name> [--latest]

The kernel, N and the algorithmic bytes per env-step are read from the bench line the traced run printed
(gpurun_out/<tag>/bench_trace.json: roofline.kernel, config.envs_per_gpu, roofline.bytes_per_env_step).

Writes <name>_kernel_stats.csv (rocprofv3 --stats table, our kernels + top others),
<name>_summary.json (avg duration, PMC bytes per launch with the gfx950 correction:
FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports exactly half of a wide coalesced
read stream on gfx950 -- MI355X_MICROARCH.md section HBM -- so read bytes = 2 * FETCH_SIZE * 1024;
WRITE_SIZE is exact for 16-byte-per-lane stores), and with --latest profiles/pmc_latest.json
(read by bench.py for roofline.traffic).
"""
import argparse
import csv
import glob
import json
import os
import statistics
import sys


def find(d, pattern):
    hits = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None       # (a re-run merges into the same directory: newest wins)


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:100]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("raw")
    ap.add_argument("out")
    ap.add_argument("--kernel", default=None, help="substring of the kernel name (default: roofline.kernel of the bench line)")
    ap.add_argument("--n-envs", type=int, default=None)
    ap.add_argument("--bytes", type=int, default=None, help="algorithmic bytes per env-step")
    ap.add_argument("--grid", type=int, default=None,
                    help="only launches with this Grid_Size_X (work-items): for a kernel that the traced program runs at several "
                         "sizes; durations then come from the kernel trace, not from the --stats table")
    ap.add_argument("--latest", action="store_true")
    a = ap.parse_args()
    full = None
    line = None
    bj = os.path.join(a.raw, "bench_trace.json")
    if os.path.exists(bj):
        try:
            line = json.loads(open(bj).read().strip().splitlines()[-1])
        except Exception:  # noqa: BLE001
            pass
    if line:
        a.kernel = a.kernel or line["roofline"]["kernel"]
        a.n_envs = a.n_envs or line["config"]["envs_per_gpu"]
        a.bytes = a.bytes or line["roofline"]["bytes_per_env_step"]
    a.kernel = a.kernel or "step_kernel"
    a.n_envs = a.n_envs or (1 << 22)
    a.bytes = a.bytes or 33
    summ = {"raw_dir": a.raw, "kernel_filter": a.kernel, "n_envs": a.n_envs, "bytes_per_env_step": a.bytes,
            "workload": line["config"]["workload"] if line else None}

    stats = find(os.path.join(a.raw, "trace"), "*kernel_stats.csv")
    if stats:
        rows = list(csv.DictReader(open(stats)))
        with open(a.out + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows[:40]:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"], r["StdDev"]])
        for r in rows:
            if a.kernel in r["Name"]:
                full = r["Name"]       # the most expensive match: every later filter uses THIS kernel only
                summ["kernel"] = short(r["Name"])
                summ["calls"] = int(r["Calls"])
                summ["avg_ns"] = float(r["AverageNs"])
                summ["min_ns"] = float(r["MinNs"])
                summ["max_ns"] = float(r["MaxNs"])
                break

    def same(name):
        return short(name) == short(full) if full else a.kernel in name

    trace = find(os.path.join(a.raw, "trace"), "*kernel_trace.csv")
    if trace:
        rows = [r for r in csv.DictReader(open(trace)) if same(r["Kernel_Name"]) and (a.grid is None or int(r["Grid_Size_X"]) == a.grid)]
        if rows:
            d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
            summ["trace_median_ns"] = statistics.median(d)
            if a.grid is not None:      # the --stats table mixes this kernel's sizes: take the figures from the trace
                summ.update(calls=len(d), avg_ns=statistics.fmean(d), min_ns=float(min(d)), max_ns=float(max(d)), grid_filter=a.grid)
            summ["grid"] = int(rows[0]["Grid_Size_X"])
            summ["workgroup"] = int(rows[0]["Workgroup_Size_X"])
            summ["vgpr"] = int(rows[0]["VGPR_Count"])
            summ["sgpr"] = int(rows[0]["SGPR_Count"])
            summ["lds_bytes"] = int(rows[0]["LDS_Block_Size"])
    for key, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        f = find(os.path.join(a.raw, key), "*counter_collection.csv")
        if not f:
            continue
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                if same(r["Kernel_Name"]) and r["Counter_Name"] == counter and (a.grid is None or int(r["Grid_Size"]) == a.grid)]
        if vals:
            summ[counter + "_KiB_median"] = statistics.median(vals)
            summ[counter + "_launches"] = len(vals)
    if "FETCH_SIZE_KiB_median" in summ and "WRITE_SIZE_KiB_median" in summ:
        rd = 2.0 * summ["FETCH_SIZE_KiB_median"] * 1024.0      # gfx950: FETCH_SIZE = half the coalesced read bytes
        wr = summ["WRITE_SIZE_KiB_median"] * 1024.0
        summ["read_bytes_per_launch_corrected"] = rd
        summ["write_bytes_per_launch"] = wr
        summ["hbm_bytes_per_launch"] = rd + wr
        per = a.bytes
        summ["algorithmic_bytes_per_launch"] = per * a.n_envs
        summ["traffic_over_algorithmic"] = (rd + wr) / (per * a.n_envs)
        if "avg_ns" in summ:        # achieved HBM rate from rocprofv3 alone: bytes per launch / average duration
            summ["achieved_GBps_algorithmic"] = per * a.n_envs / summ["avg_ns"]
            summ["achieved_GBps_pmc_traffic"] = (rd + wr) / summ["avg_ns"]
            # the roof, per regime: streams that fit the 256 MiB Infinity Cache (the bench line says so) can be served faster
            # than HBM could -- a figure above 1 is then written as hbm_spec_ratio, never as a fraction of "the" roof
            ratio = summ["achieved_GBps_algorithmic"] / 8000.0
            resident = line["roofline"].get("cache_resident") if line and line.get("roofline") else None
            if resident is None:        # (no bench line: the f-row kernels -- state streams + an 8-row action ring at this N)
                resident = a.n_envs * (a.bytes + 4 * 8) < 256 * 2 ** 20
            summ["cache_resident"] = bool(resident)
            summ["frac_of_8TBps_peak"] = ratio if ratio <= 1.0 else None
            if ratio > 1.0:
                summ["hbm_spec_ratio"] = ratio
    if line:
        summ["bench_line_under_profiler"] = line
    with open(a.out + "_summary.json", "w") as f:
        json.dump(summ, f, indent=1)
    if a.latest and "hbm_bytes_per_launch" in summ:
        # profiles/pmc_latest.json: a list, one record per (kernel, N), read by bench.py for roofline.traffic
        path = os.path.join(os.path.dirname(a.out), "pmc_latest.json")
        try:
            recs = json.load(open(path))
            recs = recs if isinstance(recs, list) else []
        except Exception:  # noqa: BLE001
            recs = []
        recs = [r for r in recs if not (r.get("kernel") == summ.get("kernel") and r.get("n_envs") == a.n_envs)]
        # the profiled kernel's compile-time resources (gym_fishing_amd/_lib/kernel_resources.json of the library in this
        # tree -- run this script before rebuilding): bench.py drops the record once the built kernel differs
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from gym_fishing_amd import build as _build
        recs.append({"kernel": summ.get("kernel"), "n_envs": a.n_envs, "hbm_bytes_per_launch": summ["hbm_bytes_per_launch"],
                     "kernel_resources": _build.kernel_resources(summ.get("kernel")),
                     "source": os.path.basename(a.out) + "_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                               "separate passes; FETCH_SIZE x2 gfx950 correction)"})
        with open(path, "w") as f:
            json.dump(recs, f, indent=1)
    print(json.dumps({k: v for k, v in summ.items() if k != "bench_line_under_profiler"}, indent=1))


if __name__ == "__main__":
    main()
