"""NumPy-facing VecEnv adapter: the stable-baselines3 `VecEnv` protocol around an N-env fishing env.

The reference trains and evaluates its agents through SB3 (tests/test-PPO.py:10-21,
examples/PPO.py:7-16) and codes its own helpers against the VecEnv shape
(gym_fishing/envs/shared_env.py:15-26,57-79).  The tensor protocol of `BaseFishingEnv` keeps
everything on the device; this wrapper is the boundary for callers that want what SB3's
`DummyVecEnv` hands them:

    obs      float32 ndarray [N, 1]   (the post-reset observation where an env just finished)
    rewards  float32 ndarray [N]
    dones    bool    ndarray [N]
    infos    list of N dicts; infos[i]["terminal_observation"] (ndarray [1]) where dones[i]

The hot path is unchanged -- one kernel launch per step with the fused auto-reset.  Up to 8192 envs
(`make_vec_env`'s default, `host_mapped=True`) the env's streams live in pinned, device-mapped host memory: the kernel
reads the actions and writes the results over PCIe and a step is launch + stream sync with no copies; larger batches
keep their state in HBM and come down as two asynchronous copies into pinned staging buffers and one stream sync.  When stable_baselines3 is importable the class derives from its
`VecEnv`, so `PPO("MlpPolicy", FishingVecEnv(env))` accepts it as is; without SB3 it is a plain
class with the same methods.
"""
import numpy as np
import torch


def _vec_env_base():
    try:
        from stable_baselines3.common.vec_env import VecEnv
        return VecEnv
    except Exception:  # noqa: BLE001 -- SB3 is optional
        return object


_BASE = _vec_env_base()


class FishingVecEnv(_BASE):
    """`env` must be an N-env (tensor protocol) env built with auto_reset=True and
    record_terminal_obs=True; `make_vec_env(id, n_envs, **kwargs)` builds one."""

    def __init__(self, env):
        if getattr(env, "_scalar", True):
            raise ValueError("FishingVecEnv wraps the N-env protocol: construct the env with num_envs=N")
        if not env.auto_reset or env._terminal_obs is None:
            raise ValueError("FishingVecEnv needs auto_reset=True and record_terminal_obs=True "
                             "(SB3 semantics: the returned obs is the post-reset one, the terminal obs goes to info)")
        self.env = env
        if _BASE is not object:
            _BASE.__init__(self, env.num_envs, env.observation_space, env.action_space)
        else:
            self.num_envs = env.num_envs
            self.observation_space = env.observation_space
            self.action_space = env.action_space
        self._obs_dtype = np.dtype(getattr(env.observation_space, "dtype", np.float32))
        # Pinned staging: the actions go up and the env's state arena (obs | t | reward | done, one
        # allocation) plus the terminal observations come down as three asynchronous copies around the
        # step kernel, followed by ONE stream sync -- a .cpu() per output would sync four times and an
        # upload from pageable memory once more.
        n = env.num_envs
        self._host_mapped = bool(getattr(env, "_host_mapped", False))
        esz = torch.empty(0, dtype=env.dtype).element_size()
        offs = env._arena_offs
        self._arena_end = offs[3] + n
        self._h_arena = torch.empty(self._arena_end, dtype=torch.uint8).pin_memory()
        self._hv_obs = self._h_arena[offs[0]:offs[0] + n * esz].view(env.dtype).numpy()
        self._hv_rew = self._h_arena[offs[2]:offs[2] + n * esz].view(env.dtype).numpy()
        self._hv_done = self._h_arena[offs[3]:offs[3] + n].numpy()
        self._h_term = torch.empty(n, dtype=env.dtype).pin_memory()
        self._hv_term = self._h_term.numpy()
        if self._host_mapped:       # zero-copy: NumPy views of the env's own pinned streams
            self._hv_obs, self._hv_rew = env._obs.numpy(), env._reward.numpy()
            self._hv_done, self._hv_term = env._done.numpy(), env._terminal_obs.numpy()
        self._h_act = torch.empty(n, dtype=env._want).pin_memory()
        self._hv_act = self._h_act.numpy()
        self._d_act = torch.empty(n, dtype=env._want, device=env.device)
        self._pending = None
        self.reset_infos = [{} for _ in range(env.num_envs)]      # SB3 >= 2.0 reads this after reset()
        self.render_mode = None
        self.metadata = dict(getattr(env, "metadata", {}))

    # ------------------------------------------------------------------ VecEnv protocol
    def _download(self, with_terminal):
        env = self.env
        if self._host_mapped:       # the env's streams ARE host memory and step() / reset() already waited
            return
        self._h_arena.copy_(env._arena[:self._arena_end], non_blocking=True)
        if with_terminal:
            self._h_term.copy_(env._terminal_obs, non_blocking=True)
        torch.cuda.current_stream(env.device).synchronize()

    def reset(self):
        self.env.reset()
        self._download(False)
        return self._hv_obs.astype(self._obs_dtype).reshape(self.num_envs, 1)

    def step_async(self, actions):
        self._pending = actions

    def step_wait(self):
        if self._host_mapped:
            self.env.step(np.asarray(self._pending))
            self._pending = None
        else:
            self._hv_act[:] = np.asarray(self._pending).reshape(self.num_envs)
            self._pending = None
            self._d_act.copy_(self._h_act, non_blocking=True)
            self.env.step(self._d_act)
            self._download(True)
        n = self.num_envs
        obs_h = self._hv_obs.astype(self._obs_dtype).reshape(n, 1)      # copies: callers may keep them across steps
        rew_h = self._hv_rew.astype(np.float32)
        done_h = self._hv_done.astype(bool)
        infos = [{} for _ in range(n)]
        idx = np.flatnonzero(done_h)
        if idx.size:
            term = self._hv_term.astype(self._obs_dtype)
            for i in idx:
                infos[i]["terminal_observation"] = term[i:i + 1]
        return obs_h, rew_h, done_h, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.env.close()

    def seed(self, seed=None):
        self.env.seed(seed)
        return [None if seed is None else int(seed) + i for i in range(self.num_envs)]

    def get_attr(self, attr_name, indices=None):
        return list(self.env.get_attr(attr_name, self._indices(indices)))

    def set_attr(self, attr_name, value, indices=None):
        self.env.set_attr(attr_name, value, self._indices(indices))

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        return list(self.env.env_method(method_name, *method_args, indices=self._indices(indices), **method_kwargs))

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * len(self._indices(indices))

    def get_images(self):
        return [None] * self.num_envs

    def render(self, mode="human"):
        return self.env.render(mode)

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        if isinstance(indices, (int, np.integer)):
            return [int(indices)]
        return [int(i) for i in indices]

    @property
    def unwrapped(self):
        return self


def make_vec_env(env_id, n_envs, **kwargs):
    """`stable_baselines3.common.env_util.make_vec_env(env_id, n_envs)` for the fishing ids: N envs on the
    device behind the NumPy VecEnv protocol."""
    from . import make
    kwargs.setdefault("auto_reset", True)
    kwargs.setdefault("host_mapped", int(n_envs) <= 8192)      # small batches: state in pinned host memory, no copies
    kwargs["record_terminal_obs"] = True
    return FishingVecEnv(make(env_id, num_envs=int(n_envs), **kwargs))
