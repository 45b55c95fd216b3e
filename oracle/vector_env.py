"""NumPy-vectorised (N,) CPU baseline -- TEST / BENCH INFRASTRUCTURE ONLY (never imported by the product).

The reference's step() arithmetic (oracle/fishing_oracle.py: step, the restatement pinned to the golden vectors)
applied to N envs at once in float64, the way the reference itself vectorises population_draw() inside BMSY()
(models/policies.py:59-63): one np.random.normal(0, 1, N) per step from the legacy global stream, uniform random
actions, reset on done.  BASELINE.md section 4.2(b).
"""
import time

import numpy as np

from . import fishing_oracle as fo

MODEL_OF = {"fishing-v0": fo.MODEL_V0, "fishing-v1": fo.MODEL_V1, "fishing-v2": fo.MODEL_V2, "fishing-v4": fo.MODEL_V4}


def time_vectorised_rollout(env_id="fishing-v1", n=1 << 16, steps=101, seed=0, sigma=0.1, r=0.3, K=1.0, C=0.5, x0=0.75,
                            Tmax=100, n_actions=100, K_mean=1.0, r_mean=0.3, sigma_p=0.1, action_low=-1.0, action_high=1.0):
    """Wall-clock of `steps` vectorised step() calls over `n` envs, one process.  Returns
    (env_steps_per_second, total_reward)."""
    model = MODEL_OF[env_id]
    np.random.seed(seed)
    arng = np.random.RandomState(seed + 1)
    Ka = np.full(n, float(K))
    ra = np.full(n, float(r))
    if model == fo.MODEL_V4:
        Ka = np.clip(np.random.normal(K_mean, sigma_p, n), 0, 1e6)
        ra = np.clip(np.random.normal(r_mean, sigma_p, n), 0, 1e6)
    obs = fo.reset_obs(model, x0, Ka, np.float64)
    t = np.zeros(n, np.int32)
    total = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        if model == fo.MODEL_V0:
            a = arng.randint(0, n_actions, n).astype(np.int32)
        else:
            a = arng.uniform(action_low, action_high, n).astype(np.float32)
        z = np.random.normal(0, 1, n)
        o2, rew, done, t2, _ = fo.step(model, obs, t, a, z, ra, Ka, sigma, C=C, Tmax=Tmax, n_actions=n_actions)
        total += float(rew.sum())
        zK = zr = None
        if model == fo.MODEL_V4:
            zK, zr = np.random.normal(0, 1, n), np.random.normal(0, 1, n)
        obs, t, Ka, ra = fo.auto_reset(model, o2, done, t2, Ka, ra, x0, zK=zK, zr=zr, K_mean=K_mean, r_mean=r_mean,
                                       sigma_p=sigma_p)
    dt = time.perf_counter() - t0
    return n * steps / dt, total
