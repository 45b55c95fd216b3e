"""The gymnasium-shaped API: reset() -> (obs, info), step() -> (obs, reward, terminated, truncated, info).

The reference registers with the old `gym` only and speaks its 4-tuple protocol (gym_fishing/envs/__init__.py:17-35,
base_fishing_env.py:60-81), which gym_fishing_amd.envs keeps.  gymnasium's checker, wrappers and every tool written
against it expect the 5-tuple, so that registry gets THIS class (gym_fishing_amd.register_with_gym), and
`make(id, api="gymnasium", ...)` returns it directly.  It owns a 4-tuple env and splits the reference's

    done = years_passed > Tmax  or  fish_population <= 0            (base_fishing_env.py:76-79)

into truncated (the horizon) and terminated (the stock is gone); both can be true on the same step.  Nothing new runs
on the device for the one-env protocol; with num_envs the split costs two or three small elementwise torch ops per
step (and, under fused auto-reset, a copy of the year counters before the step: the kernel has already reset them by
the time step() returns) -- this is the convenience surface, bench.py measures the 4-tuple path underneath.
"""
import torch


def _base():
    try:
        import gymnasium
        return gymnasium.Env
    except Exception:  # noqa: BLE001 - optional dependency
        return object


class GymnasiumFishingEnv(_base()):
    metadata = {"render_modes": ["human"]}

    def __init__(self, env_id=None, env=None, render_mode=None, **kwargs):
        if (env is None) == (env_id is None):
            raise ValueError("pass an env id (plus constructor kwargs) or a constructed 4-tuple env")
        if env is None:
            from . import make
            if kwargs.get("num_envs") is not None and kwargs.get("auto_reset", True):
                kwargs.setdefault("record_terminal_obs", True)      # terminated is read off the pre-reset observation
            env = make(env_id, **kwargs)
        elif not env._scalar and env.auto_reset and env._terminal_obs is None:
            raise ValueError("an auto-resetting N-env needs record_terminal_obs=True for the terminated / truncated split")
        self.env = env
        self.render_mode = render_mode
        self.observation_space = env.observation_space
        self.action_space = env.action_space

    # Everything else (Tmax, K, r, sigma, init_state, state, simulate, state_dict, ...) is the wrapped env's -- reads AND
    # writes: `env.unwrapped.Tmax = 50`, `.sigma = ...`, `.K = ...` are legal in the reference (its helpers and SB3
    # callbacks do exactly that) and must reach the object whose attributes feed the kernels' parameter struct, not
    # shadow it on this wrapper.
    _OWN = frozenset(("env", "render_mode", "observation_space", "action_space", "spec", "metadata", "np_random",
                      "_np_random", "_np_random_seed"))

    def __getattr__(self, name):
        if name == "env":
            raise AttributeError(name)
        return getattr(self.env, name)

    def __setattr__(self, name, value):
        if name in self._OWN or "env" not in self.__dict__:
            object.__setattr__(self, name, value)
        else:
            setattr(self.env, name, value)
        if name in ("observation_space", "action_space") and "env" in self.__dict__:
            setattr(self.env, name, value)        # (BMSY's sweep follows the env's observation Box dtype)

    @property
    def unwrapped(self):
        """gymnasium's contract: the innermost gymnasium.Env -- this object (the 4-tuple env underneath is not one).
        Attribute writes on it are forwarded (see __setattr__), so `env.unwrapped.Tmax = 50` still reaches the kernels."""
        return self

    def _conform(self, obs):
        """One env: the observation in the dtype of the declared Box (float32).  The reference hands out float64 against a
        float32 Box (quirk B6) and old gym's Box.contains never looked at the dtype; gymnasium's does -- its env checker and
        every wrapper that validates observations reject a float64 array for a float32 space -- so THIS flavour rounds the
        scalar protocol's observation once (<= 6e-8; the wrapped 4-tuple env keeps the reference's float64: `env.env`)."""
        want = getattr(self.observation_space, "dtype", None)
        return obs.astype(want) if want is not None and hasattr(obs, "astype") and obs.dtype != want else obs

    def reset(self, *, seed=None, options=None):
        mask = (options or {}).get("mask") if isinstance(options, dict) else None
        obs = self.env.reset(mask, seed=seed)
        return (self._conform(obs) if self.env._scalar else obs), {}

    def step(self, action):
        env = self.env
        if env._scalar:
            obs, reward, done, info = env.step(action)
            terminated = bool(env.fish_population <= 0.0)
            truncated = bool(env.years_passed > env.Tmax)
            return self._conform(obs), reward, terminated, truncated, info
        resetting = bool(env.auto_reset)
        t_prev = env._t.clone() if resetting else None
        obs, reward, done, info = env.step(action)
        done = done.to(torch.bool) if done.dtype != torch.bool else done
        # x <= 0  <=>  x / K - 1 <= -1 for any K > 0 (the observation map, base_fishing_env.py:162-164): the episode's
        # last observation says whether the stock is gone, whatever the env's (possibly redrawn) K
        # (fishing-v4's RESET observation is un-normalised, quirk B8; terminal_observation is always a step() output)
        last_obs = (info["terminal_observation"] if resetting else obs).reshape(-1)
        terminated = done & (last_obs <= -1.0)
        years = (t_prev.to(torch.int64) + 1) if resetting else env._t.to(torch.int64)
        truncated = done & (years > int(env.Tmax))
        return obs, reward, terminated, truncated, info

    def render(self, *args, **kwargs):
        return self.env.render(*args, **kwargs)

    def close(self):
        return self.env.close()
