#!/usr/bin/env python3
"""The SB3-shaped boundary: NumPy in, NumPy out, a list of info dicts.  What the reference's SB3 scripts
do with `DummyVecEnv([lambda: gym.make("fishing-v1")] * n)` (tests/test-PPO.py:10-21) runs here against N envs
advanced by one kernel launch per step; `model` is any object with SB3's `predict(obs) -> (actions, state)`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_fishing_amd.vec_env import make_vec_env  # noqa: E402


class EscapementModel:
    """Stand-in for a trained SB3 model: the constant-escapement rule on a batch of observations."""

    def __init__(self, S=0.5, K=1.0):
        self.S, self.K = S, K

    def predict(self, obs, state=None, mask=None, deterministic=True):
        x = (obs[:, 0] + 1.0) * self.K
        quota = np.maximum(x - self.S, 0.0)
        return (quota / self.K - 1.0).astype(np.float32).reshape(-1, 1), state


n_envs = 256
venv = make_vec_env("fishing-v1", n_envs, sigma=0.05, seed=0)
model = EscapementModel()
obs = venv.reset()
returns, running = [], np.zeros(n_envs)
for _ in range(3 * 101):                      # SB3's evaluate_policy loop, batched
    actions, _ = model.predict(obs, deterministic=True)
    obs, rewards, dones, infos = venv.step(actions)
    running += rewards
    for i in np.flatnonzero(dones):
        assert "terminal_observation" in infos[i]
        returns.append(running[i])
        running[i] = 0.0
print("episodes %d, mean reward %.4f +/- %.4f" % (len(returns), np.mean(returns), np.std(returns)))
venv.close()
