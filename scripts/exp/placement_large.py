"""Does the step kernel's speed at an HBM-resident size depend on WHERE the allocations landed?  Same layout every
trial; the arena and the action ring are re-allocated each time (optionally with a dummy allocation of a varying size
held in between to move them); full device addresses logged next to the timing.

    python scripts/exp/placement_large.py [log2_n] [trials] [returns 0/1] [mode]
      mode 0: plain re-allocation; mode 1: a dummy of (trial % 8) * 32 MiB + 2 MiB allocated first and kept
      mode 2: actions allocated once, only the env re-allocated; mode 3: env once, actions re-allocated
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import gym_fishing_amd as gf  # noqa: E402


def main():
    ln = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    ret = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
    mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    n = 1 << ln
    cfg = bench.CONFIGS["v1"]
    b = 33 if ret else 25
    acts = env = None
    for trial in range(trials):
        dummy = None
        if mode == 1:
            dummy = torch.empty(((trial % 8) * 32 + 2) << 20, dtype=torch.uint8, device="cuda")
        if acts is None or mode in (0, 1, 3):
            acts = None
            torch.cuda.empty_cache()
            acts = bench.make_actions(torch, cfg, n, 2)
        if env is None or mode in (0, 1, 2):
            env = None
            torch.cuda.empty_cache()
            env = bench.make_env(gf, torch, "v1", n, 0, ret)
            env.reset()
        env.step_many(acts, 16)
        us = min(bench.timed_steps(torch, env, acts, 40, spin_ms=15.0)[0] for _ in range(2))
        print(json.dumps({"log2_n": ln, "returns": ret, "mode": mode, "trial": trial, "us": us,
                          "arena": hex(env._arena.data_ptr()), "acts": hex(acts.data_ptr()),
                          "TBps": n * b / us / 1e6}), flush=True)
        if mode in (0, 1):
            env = acts = None
        del dummy
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
