#!/usr/bin/env python3
"""A/B compile-time variants of libfishing_hip.so (built into gym_fishing_amd/_lib/variants/):
each variant runs in its own process (FISHING_HIP_LIB selects the library), two rounds
interleaved across variants so device drift shows up as spread rather than as a winner."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, statistics, sys, torch
sys.path.insert(0, %r)
import gym_fishing_amd as gf
res = {}
for key, env_id, ln, ret, dtype in (("v1_22", "fishing-v1", 22, False, torch.float32), ("v1_22_ret", "fishing-v1", 22, True, torch.float32),
                                    ("v1_24", "fishing-v1", 24, False, torch.float32), ("v1_20", "fishing-v1", 20, False, torch.float32),
                                    ("v4_22", "fishing-v4", 22, False, torch.float32), ("v1_22_f64", "fishing-v1", 22, False, torch.float64)):
    n = 1 << ln
    env = gf.make(env_id, sigma=0.1, num_envs=n, seed=1, track_returns=ret, dtype=dtype)
    env.reset()
    acts = torch.rand((8, n), device="cuda") * 2 - 1
    steps = 300 if ln <= 22 else 80
    env.step_many(acts, steps)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(acts, steps); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / steps)
    res[key] = round(statistics.median(ts), 2)
    del env, acts
    torch.cuda.empty_cache()
print(json.dumps(res))
''' % ROOT


def main():
    libs = sorted(glob.glob(os.path.join(ROOT, "gym_fishing_amd", "_lib", "variants", "*.so")))
    only = sys.argv[1:]
    for rnd in range(2):
        for lib in libs:
            tag = os.path.basename(lib)[len("libfishing_hip_"):-3]
            if only and tag not in only:
                continue
            env = dict(os.environ, FISHING_HIP_LIB=lib)
            p = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ("ERR " + p.stderr[-300:])
            print(json.dumps({"round": rnd, "variant": tag, "us_per_launch": line}), flush=True)


if __name__ == "__main__":
    main()
