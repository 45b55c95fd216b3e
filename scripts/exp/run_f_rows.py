"""The SURVEY 8(f) kernels under one roof, for rocprofv3 (scripts/profile_f_rows.sh): the growth zoo's step kernels (one per
growth function, float32 with returns; three in float64), the fused K-step kernel and the in-kernel-policy rollouts.
Prints one JSON line per workload: the kernel rocprofv3 will name, N, env-steps per launch, algorithmic bytes per env-step,
and the rate by HIP events (back-to-back launches behind a spin-up)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf  # noqa: E402

QUICK = "--quick" in sys.argv        # (the PMC passes: fewer launches, same kernels)
ZOO_ONLY = "--zoo-only" in sys.argv  # (A/B of growth-function evaluations: the step kernels alone, every id in both layouts)
V11_ONLY = "--v11-only" in sys.argv  # (A/B of fishing-v11's forms: its step kernels in both layouts and its random-policy rollout)


def events(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps          # us per call


def make(idn, n, dtype=torch.float32, returns=True):
    kw = {} if idn == "fishing-v11" else dict(sigma=0.1)
    if idn == "fishing-v4":
        kw = dict(sigma=0.05, sigma_p=0.1)
    env = gf.make(idn, num_envs=n, seed=1, track_returns=returns, dtype=dtype, **kw)
    if idn == "fishing-v11":
        for d in env.model_params.values():
            d["sigma"] = 0.1
    env.reset()
    return env


def main():
    n = 1 << 22
    ring = torch.empty((8, n + 3072), device="cuda")
    acts = ring[:, :n]
    acts.copy_(torch.rand((8, n), device="cuda") * 2 - 1)
    launches = 60 if QUICK else 400
    ids32 = ids64 = (5, 6, 7, 8, 9, 10, 11)
    if V11_ONLY:
        ids32 = ids64 = (11,)
    for idn, dtype in [("fishing-v%d" % k, torch.float32) for k in ids32] + [("fishing-v%d" % k, torch.float64) for k in ids64]:
        env = make(idn, n, dtype)
        env.step_many(acts, 50)
        torch.cuda.synchronize()
        us = events(lambda: env.step_many(acts, launches), 1) / launches
        w = 4 if dtype == torch.float32 else 8
        # R obs + action 4 + t 4, W obs + reward + done 1 + t 4, returns R + W; v10: r R + W; v11: model_idx R 4 + W 4 (on this
        # random-policy workload practically every lane redraws a model every step: PMC traffic 1.11 x the figure without the write)
        byt = 13 + 3 * w + 2 * w + (2 * w if idn == "fishing-v10" else 0) + (8 if idn == "fishing-v11" else 0)
        print(json.dumps(dict(row="f4 zoo step", id=idn, dtype=str(dtype)[6:], kernel=env.step_kernel_name(), n_envs=n, env_steps_per_launch=n,
                              bytes_per_env_step=byt, us_per_launch=round(us, 2), frac_of_8TBps=round(n * byt / us / 8e6, 3))), flush=True)
        del env
    if ZOO_ONLY:
        return
    # fused K-step kernel (caller's actions): the launch-bound regime's tool (N = 2^20) and at the metric's size (2^22); 101 steps per
    # launch, reward / done rows out
    for ln in (() if V11_ONLY else (20, 22)):
        nn = 1 << ln
        a2 = torch.empty((8, nn + 3072), device="cuda")[:, :nn]
        a2.copy_(torch.rand((8, nn), device="cuda") * 2 - 1)
        rows_r = torch.empty((101, nn), dtype=torch.float32, device="cuda")
        rows_d = torch.empty((101, nn), dtype=torch.uint8, device="cuda")
        for idn in ("fishing-v1", "fishing-v4"):
            env = make(idn, nn)
            f = lambda: env.step_many(a2, 101, fused=True, rewards_out=rows_r, dones_out=rows_d)  # noqa: E731
            f()
            torch.cuda.synchronize()
            us = events(f, (6 if QUICK else 40) if ln == 20 else (3 if QUICK else 12))
            print(json.dumps(dict(row="f1 fused step", id=idn, log2_n=ln, kernel="fishing::step_fused_kernel<float, %d" % (1 if idn == "fishing-v1" else 4),
                                  n_envs=nn, env_steps_per_launch=nn * 101, bytes_per_env_step=9, us_per_launch=round(us, 1),
                                  env_steps_per_s="%.3e" % (nn * 101 / us * 1e6))), flush=True)
            del env
        del a2, rows_r, rows_d
        torch.cuda.empty_cache()
    # in-kernel-policy rollouts: no action traffic at all; VALU-bound
    for idn, pol, param, tag in (("fishing-v1", "random", 0.0, "1, 0, true, true, false>"), ("fishing-v1", "escapement", 0.5, "1, 2, true, true, false>"),
                                 ("fishing-v4", "random", 0.0, "4, 0, true, false, false>"),
                                 ("fishing-v11", "random", 0.0, "105, -1, true, false, false>")):
        if V11_ONLY and idn != "fishing-v11":
            continue
        # (template arguments 5 and 6: the compile-time power-of-two-K twin the dispatch picks for fishing-v0/v1/v2 at K = 1; one policy
        # parameter per env -- fishing_rollout_params_*)
        env = make(idn, n)
        T = 505
        env.rollout(T, policy=pol, param=param)        # (the warm-up launch has the timed launches' length: rocprofv3's per-kernel average mixes them)
        torch.cuda.synchronize()
        us = events(lambda: env.rollout(T, policy=pol, param=param), 2 if QUICK else 4)
        print(json.dumps(dict(row="f1 rollout", id=idn, policy=pol, kernel="fishing::rollout_kernel<float, " + tag, n_envs=n,
                              env_steps_per_launch=n * T, bytes_per_env_step=0, us_per_launch=round(us, 1),
                              env_steps_per_s="%.3e" % (n * T / us * 1e6))), flush=True)
        del env


if __name__ == "__main__":
    main()
