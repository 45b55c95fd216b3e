#!/usr/bin/env python3
"""fishing-v4 (per-episode parameter uncertainty, envs/fishing_model_error.py) at scale: 2^18 stocks, each with its own
drawn (K, r).  Every env computes its OWN escapement level -- the reference's escapement(env) / BMSY(env), run per env under
the pair that env holds -- and the whole batch is rolled out inside one fused kernel launch with one policy parameter per
env (fishing_rollout_params_*).  Beside it: the same stocks managed with the single level of the mean parameters."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_fishing_amd as gf  # noqa: E402
from gym_fishing_amd.policies import escapement  # noqa: E402

N, EPISODES = 1 << 18, 5
kw = dict(num_envs=N, sigma=0.05, sigma_p=0.15, seed=3, track_returns=True)

env = gf.make("fishing-v4", **kw)
env.reset()
model = escapement(env)                                   # BMSY() per env: a tensor of N escapement levels
S = model.S
print("escapement levels: min %.3f  mean %.3f  max %.3f" % (float(S.min()), float(S.mean()), float(S.max())))
env.rollout(EPISODES * (env.Tmax + 1), policy=model.kernel_policy)      # one launch, env i escapes to S[i]
own = env.episode_stats()

ref = gf.make("fishing-v4", **kw)
ref.reset()
ref.rollout(EPISODES * (ref.Tmax + 1), policy="escapement", param=0.5 * ref.K_mean)   # the level of the mean parameters
shared = ref.episode_stats()

for tag, st in (("own level per env  ", own), ("one level (K_mean/2)", shared)):
    print("%s: %d episodes, mean return %.4f +- %.4f" % (tag, int(st["n_episodes"]), st["mean_return"], st["std_return"]))
# (each episode redraws (K, r) while a policy object keeps the level it computed at construction -- as in the reference -- so the
# per-env levels are no better informed than the shared one; the point is that N policy objects' parameters run in one launch)
assert own["n_episodes"] >= N * EPISODES and shared["n_episodes"] >= N * EPISODES
