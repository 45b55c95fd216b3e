#!/usr/bin/env python3
"""The reference's examples/const_escapement.py on the MI355X env: the constant-escapement
rule (harvest everything above K/2) for one episode, first through the reference's scalar
protocol, then for 2^20 stochastic replicates in one fused kernel launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_fishing_amd as gf  # noqa: E402
from gym_fishing_amd.policies import escapement  # noqa: E402

r, K = 0.3, 1
env = gf.make("fishing-v1", r=r, K=K, sigma=0.0)
env.reset()
obs, rows = env.state, []
for t in range(env.Tmax):
    fish_population = env.get_fish_population(obs)
    Q = max(fish_population - K / 2, 0)             # the escapement rule
    action = env.get_action(Q)
    quota = env.get_quota(action)
    obs, reward, done, info = env.step(action)
    rows.append([t, fish_population, quota, reward, 0])
print("scalar protocol: %d steps, return %.6f" % (len(rows), sum(x[3] for x in rows)))

# the same rule for 2^20 noisy stocks at once, evaluated inside the rollout kernel
venv = gf.make("fishing-v1", r=r, K=K, sigma=0.1, num_envs=1 << 20, seed=0, track_returns=True)
venv.reset()
venv.rollout(10 * 101, policy="escapement", param=K / 2)
print("2^20 envs x 1010 steps:", venv.episode_stats())

# the reference's table for a policy object (models/policies.py) -> pandas DataFrame
df = env.simulate(escapement(env), reps=2)
print(df.head())
