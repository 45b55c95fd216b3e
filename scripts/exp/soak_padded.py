"""One-off soak of the padded-tile path at the env level: random ids / options / batch sizes (not multiples of 1024), the
padded lean launch against the same env forced onto the general kernel (launch_threads=128); every visible stream and
the episode statistics must agree bit for bit over 40 steps (mixed step / step_many / fused)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf

def bits(x):
    it = {1: torch.uint8, 4: torch.int32, 8: torch.int64}[x.element_size()]
    return x.contiguous().view(it)

rng = np.random.default_rng(2024)
ids = ["fishing-v0", "fishing-v1", "fishing-v2", "fishing-v4", "fishing-v5", "fishing-v7", "fishing-v10", "fishing-v11"]
fails = 0
for trial in range(60):
    env_id = ids[trial % len(ids)]
    n = int(rng.choice([rng.integers(1, 250) * 4, 1024 * int(rng.integers(1, 6)) + 4 * int(rng.integers(1, 255))]))
    dtype = torch.float32 if rng.random() < 0.7 else torch.float64
    kw = dict(num_envs=n, seed=int(rng.integers(1, 1 << 30)), Tmax=int(rng.integers(2, 9)), dtype=dtype,
              track_returns=bool(rng.random() < 0.6), record_terminal_obs=bool(rng.random() < 0.3),
              done_bits=bool(rng.random() < 0.3), auto_reset=bool(rng.random() < 0.8))
    if env_id != "fishing-v11":
        kw["sigma"] = 0.1 if rng.random() < 0.8 else torch.full((n,), 0.07)
    if env_id in ("fishing-v0", "fishing-v1", "fishing-v2") and rng.random() < 0.3:
        kw["compact"] = True
    if env_id == "fishing-v1" and rng.random() < 0.5:
        kw["K"] = float(rng.choice([1.5, 2.0])); kw["init_state"] = 0.75 * kw["K"]
    A = gf.make(env_id, **kw)
    B = gf.make(env_id, launch_threads=128, **kw)
    if env_id == "fishing-v11":
        for e in (A, B):
            for d in e.model_params.values():
                d["sigma"] = 0.1
    assert A._padded == (n % 1024 != 0), (n, A._padded)
    g = torch.Generator(device="cuda").manual_seed(trial)
    if env_id == "fishing-v0":
        acts = torch.randint(0, 100, (5, n + 4), device="cuda", generator=g, dtype=torch.int32)[:, :n]
    else:
        acts = (torch.rand((5, n + 4), device="cuda", generator=g) * 1.4 - 1.2)[:, :n]
    for e in (A, B):
        e.reset()
    fused_ok = not (kw["record_terminal_obs"] or kw["done_bits"])
    for phase in range(4):
        for e in (A, B):
            if phase == 0:
                for k in range(7):
                    e.step(acts[k % 5])
            elif phase == 1:
                e.step_many(acts, 11)
            elif phase == 2 and fused_ok:
                e.step_many(acts, 13, fused=(e is A))
            else:
                e.step_many(acts, 9)
        torch.cuda.synchronize()
        names = ["_obs", "_t", "_reward", "_done"] + [k for k in ("_ep_return", "_terminal_obs", "_done_bits", "_r_arr", "_K_arr", "_model_idx")
                                                       if getattr(A, k) is not None and getattr(B, k) is not None]
        for name in names:
            if not torch.equal(bits(getattr(A, name)), bits(getattr(B, name))):
                print("MISMATCH", trial, env_id, n, kw, phase, name, flush=True)
                fails += 1
        if env_id == "fishing-v4":
            if not (torch.equal(bits(A.K), bits(B.K)) and torch.equal(bits(A.r), bits(B.r))):
                print("MISMATCH K/r", trial, n, phase, flush=True); fails += 1
    if kw["track_returns"]:
        sa, sb = A.episode_stats(), B.episode_stats()
        if sa["n_episodes"] != sb["n_episodes"] or abs(sa["mean_return"] - sb["mean_return"]) > 1e-9 * max(1.0, abs(sb["mean_return"])):
            print("MISMATCH stats", trial, env_id, n, sa, sb, flush=True); fails += 1
    print("trial", trial, env_id, n, str(dtype)[6:], "ok" if not fails else "FAILS %d" % fails, flush=True)
    del A, B
print("soak", "ok" if not fails else "FAILED %d" % fails)
sys.exit(1 if fails else 0)
