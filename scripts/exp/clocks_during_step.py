"""Is the fast / slow state of the N = 2^26 step a memory-clock (DPM) state?  While step_many runs, a thread samples the
current sclk / mclk / fclk / socclk levels from sysfs (pp_dpm_*: the line marked '*') and `rocm-smi`-style power numbers
where readable; one JSON line per trial with the timing and the levels seen.

    python scripts/exp/clocks_during_step.py [log2_n] [trials]
"""
import glob
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import gym_fishing_amd as gf  # noqa: E402


def dpm_files():
    out = {}
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        for k in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
            f = os.path.join(d, k)
            if os.path.exists(f):
                out.setdefault(d, {})[k] = f
    return out


def current(f):
    try:
        for ln in open(f).read().splitlines():
            if ln.rstrip().endswith("*"):
                return ln.strip()
    except OSError as e:
        return "unreadable: %s" % e.strerror
    return None


def main():
    ln = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    n = 1 << ln
    files = dpm_files()
    print(json.dumps({"dpm_files": {d: sorted(v) for d, v in files.items()}}), flush=True)
    cfg = bench.CONFIGS["v1"]
    acts = bench.make_actions(torch, cfg, n, 2)
    for trial in range(trials):
        env = bench.make_env(gf, torch, "v1", n, 0, False)
        env.reset()
        env.step_many(acts, 16)
        seen = {}
        stop = threading.Event()

        def sample():
            while not stop.is_set():
                for d, fs in files.items():
                    for k, f in fs.items():
                        seen.setdefault(os.path.basename(os.path.dirname(d)) + ":" + k, set()).add(current(f))
                time.sleep(0.002)
        th = threading.Thread(target=sample)
        th.start()
        us = min(bench.timed_steps(torch, env, acts, 80, spin_ms=30.0)[0] for _ in range(2))
        stop.set()
        th.join()
        print(json.dumps({"log2_n": ln, "trial": trial, "us": round(us, 1), "levels": {k: sorted(str(x) for x in v) for k, v in seen.items()}}),
              flush=True)
        del env
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
