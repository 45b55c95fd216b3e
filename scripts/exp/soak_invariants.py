#!/usr/bin/env python3
"""Long auto-resetting runs of every env id in both layouts, invariants checked on the device at EVERY step:

    python scripts/exp/soak_invariants.py [--log2-n 20] [--steps 20000] > profiles/r04_soak_invariants.jsonl

N = 2^log2_n + 4 envs (a ragged tile), in-kernel noise, actions alternating between a conservative phase (quota below a
tenth of K: long episodes, stocks near their equilibria) and a random phase (quota up to 1.2 K: collapses and early resets).
Per step, accumulated in device counters (no host synchronisation inside the run): observations finite and >= -1
(population >= 0), year counter in [0, Tmax], reward finite, >= 0 and <= the stock before the step, `done` exactly where
the year counter was reset.  At the end: the episodic-return record counts exactly the dones seen, its summed lengths are
the steps of the finished episodes, and a second run from the same seed ends in the same bits.  One JSON line per (id, layout).
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf  # noqa: E402

IDS = ["fishing-v0", "fishing-v1", "fishing-v2", "fishing-v4", "fishing-v5", "fishing-v6", "fishing-v7", "fishing-v8",
       "fishing-v9", "fishing-v10", "fishing-v11"]


def run(env_id, dtype, n, steps, Tmax, check=True):
    kw = dict(num_envs=n, seed=5, Tmax=Tmax, track_returns=True, dtype=dtype)
    if env_id != "fishing-v11":
        kw["sigma"] = 0.1
    env = gf.make(env_id, **kw)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(17)
    ring = 16
    if env_id == "fishing-v0":
        low = torch.randint(0, 10, (ring, n), device="cuda", generator=g, dtype=torch.int32)
        rnd = torch.randint(0, 120, (ring, n), device="cuda", generator=g, dtype=torch.int32)
    else:
        low = torch.rand((ring, n), device="cuda", generator=g) * 0.1 - 1.0
        rnd = torch.rand((ring, n), device="cuda", generator=g) * 1.3 - 1.05
    bad = torch.zeros(6, dtype=torch.int64, device="cuda")      # obs, t, reward sign, reward <= stock, done/t, finite reward
    dones = torch.zeros((), dtype=torch.int64, device="cuda")
    lengths = torch.zeros((), dtype=torch.int64, device="cuda")
    per_env_K = env_id == "fishing-v4"
    K = None if per_env_K else float(env.params["K"])
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    for s in range(steps):
        a = (low if (s // 500) % 2 == 0 else rnd)[s % ring]
        if check:
            prev_obs = env._obs.clone()
            prev_t = env._t.clone()
            Kn = env.K.to(env._obs.dtype).clone() if per_env_K else K        # the K in force BEFORE the step (auto-reset redraws it)
        obs, rew, done, _ = env.step(a)
        if check:
            o, t = env._obs, env._t
            d = done.bool().reshape(-1)
            bad[0] += (~torch.isfinite(o) | (o < -1.0)).sum()
            bad[1] += ((t < 0) | (t > Tmax)).sum()
            bad[2] += (rew < 0).sum()
            # fishing-v4's reset observation is x0 - 1 whatever K (quirk B8): the stock is (obs + 1) * K of the episode's K
            bad[3] += (rew > (prev_obs + 1.0) * Kn + tol).sum()
            bad[4] += (d != (t == 0)).sum() + (~d & (t != prev_t + 1)).sum()
            bad[5] += (~torch.isfinite(rew)).sum()
            dones += d.sum()
            lengths += torch.where(d, prev_t + 1, torch.zeros_like(prev_t)).sum()
    torch.cuda.synchronize()
    st = env.episode_stats()
    return env, bad.cpu().tolist(), int(dones), int(lengths), st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-n", type=int, default=20)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--tmax", type=int, default=100)
    a = ap.parse_args()
    n = (1 << a.log2_n) + 4
    worst = 0
    for env_id in IDS:
        for dtype in (torch.float32, torch.float64):
            t0 = time.time()
            env, bad, dones, lengths, st = run(env_id, dtype, n, a.steps, a.tmax)
            rec_len = float(st["sum_length"])
            twin, _, _, _, st2 = run(env_id, dtype, n, a.steps, a.tmax, check=False)
            same = bool(torch.equal(env._obs.view(torch.int32 if dtype == torch.float32 else torch.int64),
                                    twin._obs.view(torch.int32 if dtype == torch.float32 else torch.int64))
                        and torch.equal(env._t, twin._t)) and st2["n_episodes"] == st["n_episodes"]
            line = dict(env_id=env_id, layout="f32" if dtype == torch.float32 else "f64", n_envs=n, steps=a.steps, Tmax=a.tmax,
                        env_steps=n * a.steps, violations=dict(zip(("obs", "t", "reward_sign", "reward_le_stock", "done_vs_t",
                                                                    "reward_finite"), bad)),
                        dones_seen=dones, record_episodes=int(st["n_episodes"]), lengths_seen=lengths,
                        record_lengths=round(rec_len), mean_return=st.get("mean_return"),
                        second_run_same_bits=same, seconds=round(time.time() - t0, 1))
            ok = (sum(bad) == 0 and dones == int(st["n_episodes"]) and abs(lengths - rec_len) <= 1e-6 * max(lengths, 1) and same)
            line["ok"] = ok
            worst |= 0 if ok else 1
            print(json.dumps(line), flush=True)
    return worst


if __name__ == "__main__":
    sys.exit(main())
