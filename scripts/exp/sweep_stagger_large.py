"""Stream stagger (bytes added between the env arena's stream starts) at HBM-resident sizes: the default 12 KiB was
chosen at N = 2^22 (inside the Infinity Cache).  fishing-v1, with and without the return accumulator.

    python scripts/exp/sweep_stagger_large.py [log2_n]
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import gym_fishing_amd as gf  # noqa: E402
from gym_fishing_amd import envs as E  # noqa: E402


def main():
    ln = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    n = 1 << ln
    cfg = bench.CONFIGS["v1"]
    for pad in (3072, 0, 5120 + 64):
        acts = bench.make_actions(torch, cfg, n, 4, pad=pad)
        for ret in (True, False):
            for stagger in (12288, 0, 256, 4096, 8192 + 256, 12288 + 256, 20480, 65536 + 4096, (1 << 20) + 12288,
                            (1 << 21) + 4096 + 256):
                if pad != 3072 and stagger not in (12288, 0):
                    continue
                E.BaseFishingEnv._STREAM_STAGGER = stagger
                env = bench.make_env(gf, torch, "v1", n, 0, ret)
                env.reset()
                env.step_many(acts, 24)
                best = []
                for _ in range(3):
                    us, _ = bench.timed_steps(torch, env, acts, 60, spin_ms=30.0)
                    best.append(us)
                b = 33 if ret else 25
                print(json.dumps({"log2_n": ln, "returns": ret, "stagger": stagger, "action_pad": pad, "us": sorted(best),
                                  "TBps_best": n * b / min(best) / 1e6}), flush=True)
                del env
                torch.cuda.empty_cache()
        del acts
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
