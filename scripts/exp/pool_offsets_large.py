"""Stream placement INSIDE one fixed allocation: the env arena is carved out of a pool that is allocated once (so the
physical placement stays put), at a varying base offset and with coarse gaps between the streams (multiples of 64 KiB
below 64 MiB).  Separates 'which physical pages' from 'which relative offsets'.

    python scripts/exp/pool_offsets_large.py [log2_n] [trials] [returns 0/1] [seed]
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
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    ret = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
    rng = random.Random(int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    n = 1 << ln
    cfg = bench.CONFIGS["v1"]
    b = 33 if ret else 25
    acts = bench.make_actions(torch, cfg, n, 2)
    coarse = os.environ.get("POOL_COARSE") == "1"          # gaps in multiples of 16 MiB below 1 GiB instead
    pool = torch.zeros(n * 17 + ((6 << 30) if coarse else (512 << 20)), dtype=torch.uint8, device="cuda")
    real_zeros = torch.zeros
    state = {"base": 0}

    def pool_zeros(size, *a, **kw):
        if isinstance(size, int) and size >= (64 << 20) and kw.get("dtype") == torch.uint8 and "cuda" in str(kw.get("device")):
            assert state["base"] + size <= pool.numel(), (state["base"], size, pool.numel())
            v = pool[state["base"]:state["base"] + size]
            v.zero_()
            return v
        return real_zeros(size, *a, **kw)

    for trial in range(trials):
        if trial < 8:
            gaps, base = (12288, 24576, 36864, 49152, 0), (trial % 4) * (16 << 20) * (trial // 4 + 1)
        else:
            unit, cnt = ((16 << 20), 64) if coarse else ((64 << 10), 1024)
            gaps = tuple(unit * rng.randrange(0, cnt) for _ in range(4)) + (0,)
            base = unit * rng.randrange(0, cnt)
        state["base"] = base
        E.BaseFishingEnv._STREAM_STAGGER = gaps
        torch.zeros = pool_zeros
        try:
            env = bench.make_env(gf, torch, "v1", n, 0, ret)
        finally:
            torch.zeros = real_zeros
        assert env._arena.data_ptr() == pool.data_ptr() + base
        env.reset()
        env.step_many(acts, 16)
        us = min(bench.timed_steps(torch, env, acts, 40, spin_ms=15.0)[0] for _ in range(2))
        print(json.dumps({"log2_n": ln, "returns": ret, "trial": trial, "gaps": gaps, "base": base, "us": us,
                          "pool": hex(pool.data_ptr()), "TBps": n * b / us / 1e6}), flush=True)
        del env


if __name__ == "__main__":
    main()
