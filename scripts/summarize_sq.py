#!/usr/bin/env python3
"""Per-kernel SQ counter shares from scripts/profile_sq.sh:

    python scripts/summarize_sq.py gpurun_out/<tag> profiles/<name>.json

For each of our kernels: launches, and per launch the medians of the SQ counters plus the shares of
SQ_WAVE_CYCLES spent parked (SQ_WAIT_ANY: s_waitcnt / barrier), issue-stalled (SQ_WAIT_INST_ANY) and issuing
(SQ_ACTIVE_INST_ANY), and the VALU share (SQ_ACTIVE_INST_VALU)."""
import collections, csv, glob, json, os, statistics, sys


def main():
    raw, out = sys.argv[1], sys.argv[2]
    f = max(glob.glob(os.path.join(raw, "pmc_sq", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if "fishing::" not in name:
            continue
        per[name][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    # kernel durations of the same pass -> VALU issue utilisation: wave64 VALU instructions per SIMD per
    # (4 cycles at 2.4 GHz), 1024 SIMDs
    dur = {}
    tf = glob.glob(os.path.join(raw, "pmc_sq", "**", "*kernel_trace.csv"), recursive=True)
    if tf:
        for r in csv.DictReader(open(max(tf, key=os.path.getmtime))):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    res = {}
    for name, disp in per.items():
        cols = collections.defaultdict(list)
        for d in disp.values():
            for k, v in d.items():
                cols[k].append(v)
        med = {k: statistics.median(v) for k, v in cols.items()}
        row = {"launches": len(disp)}
        row.update({k: med[k] for k in sorted(med)})
        wc = med.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            for k, tag in (("SQ_WAIT_ANY", "share_parked_waitcnt"), ("SQ_WAIT_INST_ANY", "share_issue_stalled"),
                           ("SQ_ACTIVE_INST_ANY", "share_issuing"), ("SQ_ACTIVE_INST_VALU", "share_valu")):
                if k in med:
                    row[tag] = round(med[k] / wc, 4)
        util = [d["SQ_INSTS_VALU"] / (dur[i] * 1e-9) / 1024 / 6.0e8 for i, d in disp.items()
                if i in dur and "SQ_INSTS_VALU" in d and dur[i] > 0]
        if util:
            row["median_duration_ns"] = statistics.median(dur[i] for i in disp if i in dur)
            row["valu_issue_utilisation"] = round(statistics.median(util), 3)
        res[name[:110]] = row
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    for k, v in res.items():
        print(k, {a: b for a, b in v.items() if a.startswith("share") or a in ("launches", "valu_issue_utilisation")})


if __name__ == "__main__":
    main()
