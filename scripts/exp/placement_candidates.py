"""N = 2^26 bare fishing-v1 step: K candidate arenas ALIVE AT ONCE (K distinct physical placements), each timed with the
same action ring -- how many of them are 'fast'?  Then each is timed again (is fast / slow a stable property of the
allocation?).

    python scripts/exp/placement_candidates.py [log2_n] [candidates] [returns 0/1]
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
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    ret = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
    n = 1 << ln
    cfg = bench.CONFIGS["v1"]
    b = 33 if ret else 25
    acts = bench.make_actions(torch, cfg, n, 2)
    envs = []
    for i in range(k):
        env = bench.make_env(gf, torch, "v1", n, 0, ret)
        env.reset()
        envs.append(env)
    for rnd in range(3):
        for i, env in enumerate(envs):
            env.step_many(acts, 16)
            us = min(bench.timed_steps(torch, env, acts, 40, spin_ms=15.0)[0] for _ in range(2))
            print(json.dumps({"log2_n": ln, "returns": ret, "round": rnd, "candidate": i, "us": round(us, 1),
                              "arena": hex(env._arena.data_ptr()), "TBps": round(n * b / us / 1e6, 2)}), flush=True)


if __name__ == "__main__":
    main()
