#!/usr/bin/env python3
"""scripts/profile_f_rows.sh's raw CSVs -> one profiles/<prefix>_<name>_summary.json per 8(f) kernel, in the schema of
scripts/summarize_profile.py (which does the work: this only walks the rows the traced run printed), plus the SQ shares.

    python scripts/summarize_f_rows.py gpurun_out/<tag> profiles/r04
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    raw, prefix = sys.argv[1], sys.argv[2]
    rows = [json.loads(ln) for ln in open(os.path.join(raw, "rows.jsonl")) if ln.startswith("{")]
    sq = {}
    sq_json = prefix + "_sq_f_rows.json"
    if os.path.isdir(os.path.join(raw, "pmc_sq")):
        subprocess.run([sys.executable, os.path.join(HERE, "summarize_sq.py"), raw, sq_json], check=True, stdout=subprocess.DEVNULL)
        sq = json.load(open(sq_json))
    index = []
    for r in rows:
        if r["row"] == "f4 zoo step":
            name = "zoo_%s_%s" % (r["id"].replace("fishing-", ""), "f64" if r["dtype"] == "float64" else "f32")
        elif r["row"] == "f1 fused step":
            name = "fused_step_%s_2p%d" % (r["id"].replace("fishing-", ""), r.get("log2_n", 20))
        else:
            name = "rollout_%s_%s" % (r["id"].replace("fishing-", ""), r["policy"])
        out = "%s_%s" % (prefix, name)
        # (a thread steps four envs: Grid_Size_X = n / 4 work-items; the fused kernel runs at two sizes in this trace)
        subprocess.run([sys.executable, os.path.join(HERE, "summarize_profile.py"), raw, out, "--kernel", r["kernel"],
                        "--n-envs", str(r["env_steps_per_launch"]), "--bytes", str(max(r["bytes_per_env_step"], 1))]
                       + (["--grid", str(r["n_envs"] // 4)] if r["row"] == "f1 fused step" else []),
                       check=True, stdout=subprocess.DEVNULL)
        s = json.load(open(out + "_summary.json"))
        shared = prefix + "_f_rows_kernel_stats.csv"    # (one shared rocprofv3 --stats table instead of a copy per kernel)
        if os.path.exists(shared):
            os.remove(out + "_kernel_stats.csv")
        else:
            os.rename(out + "_kernel_stats.csv", shared)
        s["row"], s["workload_line"] = r["row"], r
        s["env_steps_per_launch"] = s.pop("n_envs")
        if r["bytes_per_env_step"] == 0:               # no per-step HBM traffic by construction: the byte figures mean nothing
            for k in ("algorithmic_bytes_per_launch", "traffic_over_algorithmic", "achieved_GBps_algorithmic", "frac_of_8TBps_peak"):
                s.pop(k, None)
            s["bytes_per_env_step"] = 0
        if "avg_ns" in s:
            s["env_steps_per_s_rocprofv3"] = r["env_steps_per_launch"] / s["avg_ns"] * 1e9
        for k, v in sq.items():
            if s.get("kernel") and k.startswith(s["kernel"][:100]) and r["row"] != "f1 fused step":     # (per-kernel medians: sizes mixed)
                s["sq"] = {a: b for a, b in v.items() if a.startswith("share") or a in ("launches", "valu_issue_utilisation", "median_duration_ns")}
        json.dump(s, open(out + "_summary.json", "w"), indent=1)
        index.append({"file": os.path.basename(out) + "_summary.json", "kernel": s.get("kernel"), "avg_us": round(s.get("avg_ns", 0) / 1e3, 2),
                      "frac_of_8TBps_peak": s.get("frac_of_8TBps_peak"), "traffic_over_algorithmic": s.get("traffic_over_algorithmic"),
                      "env_steps_per_s": s.get("env_steps_per_s_rocprofv3"), "sq": s.get("sq")})
    json.dump(index, open(prefix + "_f_rows_index.json", "w"), indent=1)
    for i in index:
        print(i)


if __name__ == "__main__":
    main()
