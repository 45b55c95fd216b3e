"""The reference's non-learned policies (gym_fishing/models/policies.py:4-67) on top of the
HIP env: MSY constant quota, constant escapement, and the BMSY growth-curve sweep.

All three return `(action, obs)` from `predict(obs)` SB3-style, work with the scalar
protocol and with N-env tensors, and carry `kernel_policy = (policy_id, param)` so that
`env.simulate(model)` / `env.rollout(policy=...)` can run them inside the fused rollout
kernel (csrc/fishing_rollout.hip) instead of calling predict() per step.
"""
import numpy as np
import torch

from ._capi import POLICY_ESCAPEMENT, POLICY_MSY
from .spaces import is_discrete


def _is_zoo(env):
    return env.MODEL not in (0, 1, 2, 4)


def _growth_args(env):
    """Parameters of the one-step growth BMSY / msy evaluate.  The reference sets `env.sigma = 0` around the
    call (models/policies.py:10-13,60-64); the logistic / tipping models read `self.sigma`, the zoo's growth
    functions read `params["sigma"]` and never see that override (growth_models.py:208-261) -- reproduced.
    fishing-v4 evaluates with the (K, r) currently drawn (the N-env form: per env, see BMSY)."""
    kw = {} if _is_zoo(env) else {"sigma": 0.0}
    if env.MODEL == 4 and env._scalar:
        kw.update(K=float(env.K), r=float(env.r))
    return kw


def _sweep_dtype(env):
    """The dtype BMSY / msy evaluate in: that of the observation Box's grid -- float32 by default, as the reference's
    sweep is (a float32 grid times Python-float parameters stays float32 in NumPy); float64 for a caller who widened
    the Box (env.observation_space.dtype = np.float64), as the reference's linspace would follow."""
    return torch.float64 if np.dtype(env.observation_space.dtype) == np.float64 else torch.float32


def _per_env_models(env):
    """fishing-v11 with N envs: every env runs its own growth function, so BMSY / msy are per-env quantities."""
    return env.MODEL == 11 and not env._scalar


def _per_env_params(env):
    """fishing-v4 with N envs: every env runs on the (K, r) it drew, so BMSY / msy are per-env quantities."""
    return env.MODEL == 4 and not env._scalar


def BMSY(env, n=10001):
    """models/policies.py:51-67: sweep n states of the observation Box through one population_draw() on the
    device and return the population with the largest growth.  Like the reference, this resets the
    environment.  The sweep is evaluated in the grid's dtype whatever the env's layout (_sweep_dtype): with the
    default float32 Box S = 0.4996 K for the flat logistic maximum, not K / 2.
    fishing-v11 with N envs: what N reference envs would return, one S per env -- that of the growth function in force in
    that env when BMSY is called (growth_models.py:190-194) -- as a [N] tensor; one sweep per growth function of the
    model list, all in one launch.
    fishing-v4 with N envs: likewise one S per env, each swept under the (K, r) that env has drawn
    (fishing_bmsy_sweep_*: N x n growth evaluations in one launch) -- what N reference envs return."""
    grid = np.linspace(env.observation_space.low, env.observation_space.high, num=n,
                       dtype=env.observation_space.dtype).reshape(-1)
    dt = _sweep_dtype(env)
    state = torch.as_tensor(grid, device=env.device).to(dt)
    kw = _growth_args(env)
    K = kw.get("K", float(env.params["K"]))
    x0 = (state + 1.0) * K                                       # get_fish_population :158-160
    if _per_env_params(env):
        S = env.bmsy_sweep(state, env.K, env.r, dtype=dt)        # (env.K / env.r: the pairs in force, before the reset below redraws them)
        env.reset()
        return S
    if _per_env_models(env):
        kinds = sorted({int(k) for k in env._c_params().kinds[:len(env.models)]})
        X = x0.repeat(len(kinds))
        idx = torch.tensor(kinds, dtype=torch.int32, device=env.device).repeat_interleave(n)
        growth = (env.population_draw(X, dtype=dt, model_idx=idx) - X).reshape(len(kinds), n)
        S_kind = torch.zeros(5, dtype=dt, device=env.device)
        S_kind[torch.tensor(kinds, device=env.device)] = x0[torch.argmax(growth, dim=1)]
        S = S_kind[env.model_idx.long()]                         # the model in force per env, BEFORE the reset below redraws it
        env.reset()
        return S
    growth = env.population_draw(x0, dtype=dt, **kw) - x0
    S = float(x0[int(torch.argmax(growth))])
    env.reset()
    return S


class msy:
    """models/policies.py:4-19: harvest the constant quota MSY = f(BMSY) - BMSY."""

    def __init__(self, env, **kwargs):
        self.env = env
        self.S = BMSY(env)
        dt = _sweep_dtype(env)
        if _per_env_models(env) or _per_env_params(env):
            # one quota per env: f(S_i) - S_i under the growth function / the (K, r) in force in env i NOW -- BMSY's reset has
            # redrawn them, exactly as the reference's msy evaluates population_draw() after BMSY's env.reset() (:7-13)
            kw = dict(sigma=0.0, K=env.K, r=env.r) if _per_env_params(env) else {}
            self.msy = env.population_draw(self.S, dtype=dt, **kw) - self.S
            env.reset()
            self.kernel_policy = (POLICY_MSY, self.msy)          # one quota per env: fishing_rollout_params_* (ABI 7)
            return
        x = torch.tensor([self.S], dtype=dt, device=env.device)
        self.msy = float(env.population_draw(x, dtype=dt, **_growth_args(env))[0] - x[0])
        env.reset()
        self.kernel_policy = (POLICY_MSY, self.msy)

    def predict(self, obs, **kwargs):
        action = self.env.get_action(self.msy)
        if isinstance(obs, torch.Tensor) and obs.dim() >= 1 and obs.shape[0] == self.env.num_envs and not self.env._scalar:
            dt = torch.int64 if is_discrete(self.env.action_space) else torch.float32
            action = torch.as_tensor(action, device=obs.device).to(dt).expand(self.env.num_envs).reshape(-1, 1)
        return action, obs


class escapement:
    """models/policies.py:22-31: harvest everything above the escapement level S = BMSY."""

    def __init__(self, env, **kwargs):
        self.env = env
        self.S = BMSY(env)
        self.kernel_policy = (POLICY_ESCAPEMENT, self.S)         # (a tensor S -- one per env -- goes to fishing_rollout_params_*)

    def predict(self, obs, **kwargs):
        pop = self.env.get_fish_population(obs)
        if isinstance(pop, torch.Tensor):
            quota = torch.clamp(pop - self.S, min=0.0)
            action = self.env.get_action(quota)
            if not is_discrete(self.env.action_space):
                action = action.to(torch.float32)
            return action.reshape(-1, 1), obs
        quota = max(pop - self.S, 0.0)
        return self.env.get_action(quota), obs


class user_action:
    """models/policies.py:34-47: ask the person at the terminal for this step's quota."""

    def __init__(self, env, **kwargs):
        self.env = env

    def predict(self, obs, **kwargs):
        pop = self.env.get_fish_population(obs)
        quota = float(input("fish population: %s. Your harvest quota: " % (pop,)))
        return self.env.get_action(quota), obs
