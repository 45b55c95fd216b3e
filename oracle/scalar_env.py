"""Scalar, one-env-at-a-time CPU restatement of the reference env -- TEST INFRASTRUCTURE ONLY.

This is the "reference-equivalent scalar NumPy step()" of BASELINE.md section 4: one Python
object per fish stock, the same NumPy calls per step as the reference makes (np.clip on
a 1-element array, np.random.normal(0, 1) from the global legacy stream, np.maximum,
np.array([...]) for the returned state), in the same order, so that (a) seeded like the
reference (np.random.seed) it reproduces the golden trajectories bit-for-bit with no
externally supplied noise (tests/test_oracle_golden.py::test_scalar_env_*), and (b) its
wall-clock per step is what the reference costs on a host core.  bench.py times it as
`cpu_baseline` (kind "port").  The product never imports it.

Follows (paths relative to /root/reference/gym_fishing/envs/):
  base_fishing_env.py:19-58 ctor, :60-81 step, :83-91 reset, :112-164 helpers;
  fishing_env.py:7-24 (v0), fishing_cts_env.py:5-12 (v1), fishing_tipping_env.py:7-35 (v2),
  fishing_model_error.py:9-48 (v4).
"""
import numpy as np

_DEFAULTS = {"r": 0.3, "K": 1, "sigma": 0.0, "init_state": 0.75, "Tmax": 100, "n_actions": 100, "C": 0.5,
             "K_mean": 1.0, "r_mean": 0.3, "sigma_p": 0.1}


class ScalarFishingEnv:
    def __init__(self, env_id="fishing-v1", **kw):
        unknown = set(kw) - set(_DEFAULTS)
        if unknown:
            raise TypeError("unexpected kwargs %s" % sorted(unknown))
        cfg = dict(_DEFAULTS, **kw)
        self.env_id = env_id
        self.discrete = env_id == "fishing-v0"
        self.tipping = env_id == "fishing-v2"
        self.model_error = env_id == "fishing-v4"
        self.sigma = cfg["sigma"]
        self.x0 = cfg["init_state"]
        self.Tmax = cfg["Tmax"]
        self.n_actions = cfg["n_actions"]
        self.C = cfg["C"]
        self.low = np.array([-1], dtype=np.float32)
        self.high = np.array([1], dtype=np.float32)
        if self.model_error:
            self.K_mean, self.r_mean, self.sigma_p = cfg["K_mean"], cfg["r_mean"], cfg["sigma_p"]
            self.state = np.array([self.x0 / self.K_mean - 1])      # base ctor runs first (:46)
            self._draw_params()                                     # fishing_model_error.py:37-38
        else:
            self.K, self.r = cfg["K"], cfg["r"]
            self.state = np.array([self.x0 / self.K - 1])
        self.t = 0

    def _draw_params(self):
        # K first, then r; each clipped to [0, 1e6]
        self.K = np.clip(np.random.normal(self.K_mean, self.sigma_p), 0, 1e6)
        self.r = np.clip(np.random.normal(self.r_mean, self.sigma_p), 0, 1e6)

    def reset(self):
        if self.model_error:                                        # fishing_model_error.py:41-48
            self._draw_params()
            self.state = np.array([self.x0])                        # un-normalised (quirk B8)
        else:                                                       # base_fishing_env.py:83-91
            self.state = np.array([self.x0 / self.K - 1])
        self.t = 0
        return self.state

    def quota(self, action):
        if self.discrete:
            return (action / self.n_actions) * self.K               # :140
        a = np.clip(action, self.low, self.high)[0]                 # :143-145
        return (a + 1) * self.K                                     # :146

    def step(self, action):
        K, r = self.K, self.r
        q = self.quota(action)
        x = (self.state[0] + 1) * K                                 # :159
        h = min(x, q)                                               # :117
        x = max(x - h, 0.0)                                         # :118
        if self.tipping:                                            # fishing_tipping_env.py:25-34
            x = np.maximum(x * np.exp(r * (1 - x / K) * (x - self.C) + x * self.sigma * np.random.normal(0, 1)), 0)
        else:                                                       # :125-131
            x = np.maximum(x + r * x * (1.0 - x / K) + x * self.sigma * np.random.normal(0, 1), 0.0)
        self.state = np.array([x / K - 1])                          # :163
        reward = max(h, 0.0)                                        # :74
        self.t += 1
        done = bool(self.t > self.Tmax)                             # :76
        if x <= 0.0:                                                # :78-79
            done = True
        return self.state, reward, done, {}


def time_random_rollout(env_id="fishing-v1", n_env_steps=200_000, seed=0, **kw):
    """Wall-clock of `n_env_steps` step() calls with a uniform random policy and reset on
    done (one env, one core).  Returns (env_steps_per_second, total_reward)."""
    import time
    np.random.seed(seed)
    env = ScalarFishingEnv(env_id, **kw)
    env.reset()
    arng = np.random.RandomState(seed + 1)
    if env.discrete:
        acts = arng.randint(0, env.n_actions, n_env_steps)
    else:
        acts = arng.uniform(-1, 1, (n_env_steps, 1)).astype(np.float32).astype(np.float64)
    total = 0.0
    t0 = time.perf_counter()
    for k in range(n_env_steps):
        _, rew, done, _ = env.step(acts[k])
        total += rew
        if done:
            env.reset()
    dt = time.perf_counter() - t0
    return n_env_steps / dt, total
