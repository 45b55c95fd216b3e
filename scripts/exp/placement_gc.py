"""placement_large.py mode 2 (only the env re-created per trial), with and without a gc.collect() between dropping the old
env and building the new one: is the fast / slow alternation a matter of WHEN the old arena is returned to the driver?

    python scripts/exp/placement_gc.py [log2_n] [trials] [collect 0/1]
"""
import gc
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import gym_fishing_amd as gf  # noqa: E402


def main():
    ln = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    collect = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
    n = 1 << ln
    cfg = bench.CONFIGS["v1"]
    acts = bench.make_actions(torch, cfg, n, 2)
    env = None
    for trial in range(trials):
        env = None
        if collect:
            gc.collect()
        torch.cuda.empty_cache()
        before = torch.cuda.memory_reserved()
        env = bench.make_env(gf, torch, "v1", n, 0, False)
        env.reset()
        env.step_many(acts, 16)
        us = min(bench.timed_steps(torch, env, acts, 40, spin_ms=15.0)[0] for _ in range(2))
        print(json.dumps({"log2_n": ln, "collect": collect, "trial": trial, "us": round(us, 1), "arena": hex(env._arena.data_ptr()),
                          "reserved_before_MB": before >> 20, "reserved_now_MB": torch.cuda.memory_reserved() >> 20,
                          "gc_count": gc.get_count()}), flush=True)


if __name__ == "__main__":
    main()
