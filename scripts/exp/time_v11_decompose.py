#!/usr/bin/env python3
"""Where fishing-v11's step time goes, by elimination (N = 2^22, float32 / float64, HIP events over 400 launches):
model lists of 1 / 2 / 5 growth functions (one kind = no regroup inefficiency), with / without the return record, with / without
auto-reset (no auto-reset = no model redraw), against fishing-v10 (the same 41 bytes per env-step) and fishing-v9 (33)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf  # noqa: E402


def events(env, acts, launches=400):
    env.step_many(acts, 60)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    env.step_many(acts, launches)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / launches


def main():
    n = 1 << 22
    acts = torch.empty((8, n + 3072), device="cuda")[:, :n]
    acts.copy_(torch.rand((8, n), device="cuda") * 2 - 1)
    low = torch.empty((8, n + 3072), device="cuda")[:, :n]
    low.copy_(torch.rand((8, n), device="cuda") * 0.1 - 1.0)        # small quotas: long episodes, few redraws
    for dtype in (torch.float32, torch.float64):
        for name, kw, a in (
                ("v11 5 models", dict(), acts), ("v11 2 models (allen, ricker)", dict(models=("allen", "ricker")), acts),
                ("v11 1 model (allen)", dict(models=("allen",)), acts), ("v11 1 model (may)", dict(models=("may",)), acts),
                ("v11 5 models, no returns", dict(track_returns=False), acts),
                ("v11 5 models, no auto-reset (no redraw; finished envs stepped on)", dict(auto_reset=False), acts),
                ("v11 5 models, small quotas (mean episode ~100 steps)", dict(), low)):
            k = dict(num_envs=n, seed=1, track_returns=True, dtype=dtype)
            k.update(kw)
            env = gf.make("fishing-v11", **k)
            for d in env.model_params.values():
                d["sigma"] = 0.1
            env.reset()
            us = events(env, a)
            print(json.dumps(dict(id="fishing-v11", dtype=str(dtype)[6:], case=name, us_per_launch=round(us, 2), kernel=env.step_kernel_name())), flush=True)
            del env
        for idn in ("fishing-v10", "fishing-v9"):
            env = gf.make(idn, num_envs=n, seed=1, track_returns=True, dtype=dtype, sigma=0.1)
            env.reset()
            print(json.dumps(dict(id=idn, dtype=str(dtype)[6:], case="reference point", us_per_launch=round(events(env, acts), 2),
                                  kernel=env.step_kernel_name())), flush=True)
            del env


if __name__ == "__main__":
    main()
