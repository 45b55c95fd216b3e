"""Where the wall time of the driver's command (`bench.py --steps 20 --warmup 5`) goes: the timed region is
20 launches (~426 us of kernel time) and its wall clock reads ~500 us.  Variants of the same region, 60 repeats each,
medians in us:

  region          = sync; t0; step_many(20); episode_record(); sync                      (bench.py's region)
  no_record       = the same without the record's reduce kernel
  poll            = region, but the host spins on an event query before the closing synchronize
  enqueue_only    = host time of step_many(20) alone (no wait)
  events          = HIP events around the 20 launches of `region`

    python scripts/exp/time_k20_region.py [K]
"""
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
    R = 60
    res = {}

    def run(name, record, poll):
        walls, evs, enq = [], [], []
        for _ in range(R):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e2 = torch.cuda.Event()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record()
            env.step_many(actions, K)
            e1.record()
            if record:
                env.episode_record()
            t1 = time.perf_counter()
            if poll:
                e2.record()
                while not e2.query():
                    pass
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            walls.append((t2 - t0) * 1e6)
            enq.append((t1 - t0) * 1e6)
            evs.append(e0.elapsed_time(e1) * 1e3)
        res[name] = {"wall_us": statistics.median(walls), "wall_min_us": min(walls), "enqueue_us": statistics.median(enq),
                     "events_us": statistics.median(evs), "per_step_wall_us": statistics.median(walls) / K}

    run("region", True, False)
    run("no_record", False, False)
    run("poll", True, True)
    run("poll_no_record", False, True)
    # back-to-back: the same K launches with the device already busy (what the roofline uses)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    env.step_many(actions, 16)
    e0.record()
    env.step_many(actions, 256)
    e1.record()
    torch.cuda.synchronize()
    res["steady_us_per_launch"] = e0.elapsed_time(e1) * 1e3 / 256
    res["K"] = K
    print(json.dumps(res))


if __name__ == "__main__":
    main()
