"""Table of a build-variant A/B session of scripts/exp/small_n_shapes (trace medians per round):
    python scripts/exp/small_n_ab_table.py gpurun_out/r03_s05 base noslp ..."""
import json, sys, os
d = sys.argv[1]; vs = sys.argv[2:]
def load(f): return [json.loads(l) for l in open(f)] if os.path.exists(f) else []
keys = ["copy_bare", "step_bare", "prod_bare", "copy_ret", "step_rec1", "step_rec11", "prod_ret"]
for ln in (19, 20, 21, 22):
    print("N=2^%d  trace median us per round" % ln)
    for v in vs:
        cols = {k: [] for k in keys}
        for rnd in (1, 2, 3):
            for r in load(os.path.join(d, "trace_%s_%d.jsonl" % (v, rnd))):
                if r["grid_threads"] * 4 != 1 << ln: continue
                k = r["kernel"]
                name = {"shape_kernel<1, 256, 4, false, 0>": "copy_bare", "shape_kernel<1, 256, 4, true, 0>": "copy_ret",
                        "shape_kernel<2, 256, 4, false, 0>": "step_bare", "shape_kernel<2, 256, 4, true, 1>": "step_rec1",
                        "shape_kernel<2, 256, 4, true, 11>": "step_rec11"}.get(k)
                if "lean<float, 1, 4098>" in k or "lean<float, 1, 12290>" in k: name = "prod_bare"
                if "lean<float, 1, 4102>" in k or "lean<float, 1, 12294>" in k: name = "prod_ret"
                if name: cols[name].append(r["us_median"])
        print("  %-9s" % v + "  ".join("%s %s" % (k, "/".join("%.2f" % x for x in cols[k])) for k in keys))
