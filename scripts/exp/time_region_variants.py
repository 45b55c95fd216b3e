"""The driver's 20-step timed region (bench.py) in variants, 40 repeats each, median wall us:
  as_is      = ev0; step_many(20); ev1; episode_record(); synchronize          (bench.py's region)
  no_events  = step_many(20); episode_record(); synchronize
  no_record  = ev0; step_many(20); ev1; synchronize
  bare       = step_many(20); synchronize
"""
import gc
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import gym_fishing_amd as gf  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = 1 << 22
    cfg = bench.CONFIGS["v1"]
    env = bench.make_env(gf, torch, "v1", n, 0, True)
    env.reset()
    actions = bench.make_actions(torch, cfg, n, bench.RING)
    bench.spin_up(torch, env, actions, 300.0)
    env.episode_stats()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def region(events, record):
        if events:
            ev0.record()
        env.step_many(actions, K)
        if events:
            ev1.record()
        if record:
            env.episode_record()
        torch.cuda.synchronize()
    res = {"K": K}
    gc.collect()
    gc.disable()
    for name, ev, rec in (("as_is", True, True), ("no_events", False, True), ("no_record", True, False), ("bare", False, False),
                          ("as_is_again", True, True)):
        walls = []
        for _ in range(40):
            region(ev, rec)                      # the dress rehearsal: its closing synchronize is the opening bracket
            t0 = time.perf_counter()
            region(ev, rec)
            walls.append((time.perf_counter() - t0) * 1e6)
        res[name] = {"wall_us": round(statistics.median(walls), 1), "min_us": round(min(walls), 1), "per_step_us": round(statistics.median(walls) / K, 3)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
