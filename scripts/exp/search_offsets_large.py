"""Random search over the relative placement of the env arena's streams (and of the action rows) at an HBM-resident
size.  Each trial: gaps after obs / t / reward / done (bytes, multiples of 256 below 64 KiB... plus the stream sizes)
and the action row padding; the step kernel timed over 2 x 40 launches.  One JSON line per trial.

    python scripts/exp/search_offsets_large.py [log2_n] [trials] [returns 0/1] [seed]
"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import gym_fishing_amd as gf  # noqa: E402
from gym_fishing_amd import envs as E  # noqa: E402


def main():
    ln = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    ret = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
    rng = random.Random(int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    n = 1 << ln
    cfg = bench.CONFIGS["v1"]
    b = 33 if ret else 25
    acts_cache = {}
    for trial in range(trials):
        if trial == 0:
            gaps, pad = (12288, 24576, 36864, 49152, 0), 3072
        elif trial == 1:
            gaps, pad = (0, 0, 0, 0, 0), 3072
        else:
            gaps = tuple(256 * rng.randrange(0, 256) for _ in range(4)) + (0,)
            pad = 64 * rng.randrange(0, 256)          # elements: 256-byte steps below 64 KiB
        if pad not in acts_cache:
            acts_cache.clear()
            torch.cuda.empty_cache()
            acts_cache[pad] = bench.make_actions(torch, cfg, n, 2, pad=pad)
        acts = acts_cache[pad]
        E.BaseFishingEnv._STREAM_STAGGER = gaps
        env = bench.make_env(gf, torch, "v1", n, 0, ret)
        env.reset()
        env.step_many(acts, 16)
        us = min(bench.timed_steps(torch, env, acts, 40, spin_ms=15.0)[0] for _ in range(2))
        print(json.dumps({"log2_n": ln, "returns": ret, "gaps": gaps, "action_pad": pad, "us": us,
                          "arena_base_mod_2M": env._arena.data_ptr() % (1 << 21), "act_base_mod_2M": acts.data_ptr() % (1 << 21),
                          "TBps": n * b / us / 1e6}), flush=True)
        del env
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
