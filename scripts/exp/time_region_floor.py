"""What the runtime charges a "synchronize; K launches; synchronize" region whatever the kernels do: the driver's timed region
with the step kernel, with the library's empty kernel of the same grid and argument shape (fishing_step_floor_f32 mode 0)
and with the copy floor (mode 1), K = 1 / 20, 60 repeats each, median wall us.  region(step) - K * steady launch time and
region(empty) - K * empty launch time are the fixed cost F_region of DESIGN.md section 6.

    python scripts/exp/time_region_floor.py > profiles/r06_region_fixed_cost.json
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
from gym_fishing_amd import _capi  # noqa: E402


def main():
    n = 1 << 22
    env = bench.make_env(gf, torch, "v1", n, 0, True)
    env.reset()
    actions = bench.make_actions(torch, bench.CONFIGS["v1"], n, bench.RING)
    bench.spin_up(torch, env, actions, 300.0)
    env.episode_stats()
    lib, bufs, stream = env._lib, env._c_buffers(actions[0]), env._stream()
    floor = lambda mode, k: _capi.check(lib.fishing_step_floor_f32(mode, n, bufs, k, stream), "floor")   # noqa: E731
    res = {"n_envs": n}
    gc.collect()
    gc.disable()
    for K in (1, 20):
        for name, fn in (("step", lambda: env.step_many(actions, K)), ("step_and_record", lambda: (env.step_many(actions, K), env.episode_record())),
                         ("empty", lambda: floor(0, K)), ("copy", lambda: floor(1, K))):
            walls = []
            for _ in range(60):
                fn()
                torch.cuda.synchronize()             # (the rehearsal: its closing synchronize is the opening bracket)
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                walls.append((time.perf_counter() - t0) * 1e6)
            res["K%d_%s" % (K, name)] = {"wall_us": round(statistics.median(walls), 2), "min_us": round(min(walls), 2)}
    res["steady_step_us"] = round(bench.steady_launch_us(torch, env, actions, 256, spin_ms=30.0), 3)
    res["steady_empty_us"] = round(bench.floor_launch_us(torch, env, actions, n, 0), 3)
    res["steady_copy_us"] = round(bench.floor_launch_us(torch, env, actions, n, 1), 3)
    for K in (1, 20):
        res["K%d_fixed_cost_us" % K] = {"step": round(res["K%d_step" % K]["wall_us"] - K * res["steady_step_us"], 2),
                                        "step_and_record": round(res["K%d_step_and_record" % K]["wall_us"] - K * res["steady_step_us"], 2),
                                        "empty": round(res["K%d_empty" % K]["wall_us"] - K * res["steady_empty_us"], 2),
                                        "copy": round(res["K%d_copy" % K]["wall_us"] - K * res["steady_copy_us"], 2)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
