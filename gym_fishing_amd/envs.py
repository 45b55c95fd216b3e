"""Host-side mirror of the reference's env classes, backed by libfishing_hip.so.

Same class names, constructor kwargs, defaults, attributes and methods as
gym_fishing/envs/{base_fishing_env,fishing_env,fishing_cts_env,fishing_tipping_env,
fishing_model_error}.py, so it is a drop-in for the rollout path:

* ``num_envs=None`` (default)  -> the reference's scalar protocol: ``reset() -> ndarray(1,)``,
  ``step(a) -> (ndarray(1,) float64, float, bool, {})`` computed by the fp64 parity kernel
  on one env (BASELINE config 1, plumbing).
* ``num_envs=N``              -> N envs in lockstep with the SB3-VecEnv shape the reference's
  own helpers code against (shared_env.py:15-26,57-79): ``reset() -> obs[N,1]``,
  ``step(actions[N,1]) -> (obs[N,1], rewards[N], dones[N], info)``; tensors stay on the
  GPU (torch, zero-copy views of the env's buffers, valid until the next step/reset).

Random numbers: ``rng="philox"`` (default with num_envs) = the in-kernel counter-based streams keyed
by ``seed`` and the global env index; ``rng="numpy"`` (default for the scalar protocol) = the reference's
own draws from NumPy's global legacy stream, in its order, handed to the kernel as external noise -- so
``np.random.seed(s)`` reproduces the reference's trajectories.

All arithmetic happens in the HIP kernels (csrc/); this file only owns buffers, counters
and argument plumbing.  Three concerns live in mixins of their own: fishing-v4's parameter-mode
machine (v4_params.py), checkpoints / graph replay (checkpoint.py), the reference's helper
methods -- get_quota ... population_draw, simulate, plot -- (reference_helpers.py).  No CPU fallback: without the library or a HIP device the
constructor raises FishingLibraryError.
"""
import csv
import ctypes

import numpy as np
import torch

from . import _capi
from ._capi import (FLAG_AUTO_RESET, FLAG_PADDED_TILES, FLAG_RESET_COUNTER_ON_DEVICE, FLAG_T_U8, FLAG_V4_DERIVED, KIND_OF_NAME, MODEL_V0, MODEL_V1, MODEL_V2, MODEL_V4, MODEL_V5, MODEL_V6,
                    MODEL_V7, MODEL_V8, MODEL_V9, MODEL_V10, MODEL_V11, POLICY_CONSTANT, POLICY_ESCAPEMENT,
                    POLICY_MSY, POLICY_RANDOM, FishingLibraryError)
from .spaces import space_classes
from .checkpoint import STATE_FORMAT, V4_PARAM_STREAM, CheckpointAndReplay  # noqa: F401  (re-exported)
from .reference_helpers import ReferenceHelpers
from .v4_params import V4ParameterModes

POLICIES = {"random": POLICY_RANDOM, "constant": POLICY_CONSTANT, "escapement": POLICY_ESCAPEMENT,
            "msy": POLICY_MSY}


# raw hipStream_t of torch's current stream: the private fast path Inductor / Triton use (0.2 us) instead
# of building a torch.cuda.Stream object per call (2.6 us of a 10 us step() on the host)
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _current_stream_ptr(device_index):
    if _RAW_STREAM is not None:
        return _RAW_STREAM(device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


def _require_device(device):
    if not torch.cuda.is_available():
        raise FishingLibraryError(
            "gym_fishing_amd needs a HIP device (MI355X): torch.cuda.is_available() is False. "
            "There is no CPU backend; the CPU restatement under oracle/ is test infrastructure.")
    dev = torch.device("cuda" if device is None else device)
    if dev.type != "cuda":
        raise FishingLibraryError("device must be a HIP ('cuda') device, got %r" % (device,))
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


class _Repeat:
    """What VecEnv.get_attr returns for a value shared by all N envs (no N-long list)."""

    def __init__(self, value, n):
        self.value, self.n = value, n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self.value] * len(range(*i.indices(self.n)))
        if not -self.n <= i < self.n:
            raise IndexError(i)
        return self.value

    def __iter__(self):
        return (self.value for _ in range(self.n))


def _gym_env_base():
    """`gym.Env` when the (optional) gym package is importable, so isinstance checks of tooling
    built on the reference's gym 0.17 API (SB3's DummyVecEnv / check_env) accept these classes;
    plain `object` otherwise.  gymnasium is not used as a base: its 5-tuple step API differs
    from the reference's 4-tuple one, which this package keeps."""
    try:
        import gym
        return gym.Env
    except Exception:  # noqa: BLE001 - optional dependency
        return object


class BaseFishingEnv(V4ParameterModes, CheckpointAndReplay, ReferenceHelpers, _gym_env_base()):
    """base_fishing_env.py:16-164, vectorised.  See the module docstring."""

    metadata = {"render.modes": ["human"]}
    MODEL = MODEL_V1
    # what FishingParams is built from beyond the base fields, per class: attributes / entries of `params` the model's kernels
    # read (_param_key compares them on every step(): only what can matter is looked at)
    _KEY_ATTRS = ()
    _KEY_PARAMS = ()
    _STREAM_STAGGER = 12288      # bytes between the arena's stream starts beyond their sizes (see __init__)

    def __init__(self, params=None, Tmax=100, file=None, *, num_envs=None, device=None, seed=0,
                 dtype=None, auto_reset=None, env_offset=0, record_terminal_obs=False,
                 track_returns=False, done_bits=False, launch_blocks=0, launch_threads=0, compact=False, rng=None,
                 host_mapped=False, derived_params=None):
        params = dict({"r": 0.3, "K": 1, "sigma": 0.0, "x0": 0.75} if params is None else params)
        self.params = params
        self.Tmax = int(Tmax)
        self.file = file
        self.init_state = params["x0"]
        self._scalar = num_envs is None
        # host_mapped (N-env protocol, meant for small N behind the NumPy VecEnv adapter): the env's streams live
        # in pinned, device-mapped host memory like the scalar protocol's, the kernels read and write them over
        # PCIe, step() / reset() wait for the stream and hand back CPU tensors -- no staging copies
        self._host_mapped = self._scalar or bool(host_mapped)
        self.num_envs = 1 if self._scalar else int(num_envs)
        if self.num_envs < 1:
            raise ValueError("num_envs must be >= 1")
        if env_offset % 4 or env_offset < 0:
            raise ValueError("env_offset must be a non-negative multiple of 4 (noise quads / 16-byte rows)")
        self.env_offset = int(env_offset)
        # compact layout: years_passed as one byte per env instead of four (19 instead of 25 bytes per
        # env-step in the fp32 layout); needs Tmax <= 254
        self.compact = bool(compact)
        if self.compact and (num_envs is None or int(Tmax) > 254):
            raise ValueError("compact=True is for the N-env protocol with Tmax <= 254")
        if dtype is None:
            dtype = torch.float64 if self._scalar else torch.float32
        if dtype not in (torch.float32, torch.float64):
            raise ValueError("dtype must be torch.float32 (fast layout) or torch.float64 (parity layout)")
        self.dtype = dtype
        self.auto_reset = (not self._scalar) if auto_reset is None else bool(auto_reset)
        # Where the random numbers come from.  "philox": the counter-based in-kernel streams keyed by `seed`.
        # "numpy" (scalar protocol only, its default): exactly the reference's draws from NumPy's global legacy
        # stream -- np.random.normal(0, 1) once per step() also at sigma = 0 (base_fishing_env.py:130),
        # np.random.normal(mean, sigma_p) for fishing-v4's K then r (fishing_model_error.py:37-43),
        # np.random.choice(models) for fishing-v11 (growth_models.py:187,200) -- handed to the kernel as external
        # noise, so `np.random.seed(s)` reproduces the reference's trajectory.
        # With N envs "numpy" draws np.random.normal(0, 1, N) per step -- the order in which SB3's DummyVecEnv
        # steps N reference envs one after the other -- for the ids whose reset() draws nothing (not v4 / v11,
        # whose per-episode draws interleave with the step draws env by env).
        rng = ("numpy" if self._scalar else "philox") if rng is None else rng
        if rng not in ("numpy", "philox") or (rng == "numpy" and not self._scalar and self.MODEL in (MODEL_V4, MODEL_V11)):
            raise ValueError("rng must be 'philox' or 'numpy' ('numpy' with num_envs is not available for "
                             "fishing-v4 / fishing-v11)")
        self._np_rng = rng == "numpy"
        self._seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._step_count = 0
        self._reset_count = 0

        # spaces (base_fishing_env.py:49-58): Box(-1, 1, (1,), float32) for both
        Box, _ = space_classes()
        self.action_space = Box(np.array([-1], dtype=np.float32), np.array([1], dtype=np.float32),
                                dtype=np.float32)
        self.observation_space = Box(np.array([-1], dtype=np.float32), np.array([1], dtype=np.float32),
                                     dtype=np.float32)

        self._lib = _capi.lib()                      # raises if the HIP library is missing
        self.device = _require_device(device)
        self._dev_index = self.device.index
        self._suffix = "f32" if dtype == torch.float32 else "f64"
        self._fn_step = getattr(self._lib, "fishing_step_" + self._suffix)
        self._fn_reset = getattr(self._lib, "fishing_reset_" + self._suffix)
        self._fn_rollout = getattr(self._lib, "fishing_rollout_" + self._suffix)
        self._fn_rollout_params = getattr(self._lib, "fishing_rollout_params_" + self._suffix)
        self._fn_step_many = getattr(self._lib, "fishing_step_many_" + self._suffix)
        self._fn_step_fused = getattr(self._lib, "fishing_step_fused_" + self._suffix)

        N, dev = self.num_envs, self.device
        self._per_env = self.MODEL == MODEL_V4
        # The four streams every step touches live in one arena, each start staggered by a further
        # 12 KiB: streams that are walked in lockstep from power-of-two-spaced bases collide in the
        # HBM channel hash (measured at N = 2^22: 16.6 us unstaggered -> 16.1 us; profiles/
        # r01c_stream_stagger_experiment.jsonl).
        esz = torch.empty(0, dtype=dtype).element_size()
        # FISHING_FLAG_PADDED_TILES: with room for whole 1024-env tiles behind every state stream, a batch that is not a
        # multiple of 1024 envs steps in ONE launch instead of two (N = 10^6: 9.3 -> 5.1 us per step).  The envs behind
        # the N-th are scratch; every tensor the env hands out is the [:N] view.
        self._cap = N
        if not self._host_mapped and N % 1024 and N % 4 == 0:      # (also batches below one tile: the lean kernel then)
            self._cap = (N + 1023) // 1024 * 1024
        self._padded = self._cap != N
        C = self._cap
        sizes = [(dtype, C * esz), (torch.uint8, C) if self.compact else (torch.int32, C * 4), (dtype, C * esz),
                 (torch.uint8, C)]
        if track_returns:
            sizes.append((dtype, C * esz))
        stagger = 0 if self._host_mapped else self._STREAM_STAGGER  # host-mapped: nothing to de-alias, keep the arena small
        offs, off = [], 0
        for k, (_, nbytes) in enumerate(sizes):
            offs.append(off)
            gap = stagger[k] if isinstance(stagger, (tuple, list)) else stagger * (k + 1)
            off = (off + nbytes + gap + 255) & ~255
        self._arena_offs = offs
        if self._host_mapped:
            # scalar protocol: the single env's streams (+ its action) live in pinned, device-mapped
            # host memory.  The kernels read and write it directly over PCIe (a few bytes), so a step
            # is launch + stream sync with no staging copies: 17 us instead of ~80 us per step.
            self._action_off = off
            self._arena = torch.zeros(off + 256, dtype=torch.uint8).pin_memory()
            self._arena_np = self._arena.numpy()
        else:
            self._arena = torch.zeros(off, dtype=torch.uint8, device=dev)
        views = [self._arena[o:o + nb].view(dt)[:N] for o, (dt, nb) in zip(offs, sizes)]
        self._obs, self._t, self._reward, self._done = views[:4]
        self._r_arr = self._K_arr = self._sigma_arr = None
        sigma = params["sigma"]
        if isinstance(sigma, (torch.Tensor, np.ndarray, list, tuple)):
            self._sigma_arr = self._per_env_buffer(dtype)
            self._sigma_arr.copy_(torch.as_tensor(sigma).to(device=dev, dtype=dtype).reshape(N))
            self._sigma_scalar = float(self._sigma_arr[0])
        else:
            self._sigma_scalar = float(sigma)
        # fishing-v4: which parameter mode the env starts in (v4_params.py: derived from the Philox streams / stored arrays)
        self._init_v4_modes(derived_params)
        if self.MODEL == MODEL_V10:      # the drifting growth rate is per-env state (growth_models.py:151)
            self._r_arr = self._per_env_buffer(dtype, float(params["r"]))
        self._model_idx = self._per_env_buffer(torch.int32) if self.MODEL == MODEL_V11 else None
        self._terminal_obs = None
        if record_terminal_obs:
            self._terminal_obs = (torch.empty(N, dtype=dtype).pin_memory() if self._host_mapped
                                  else self._per_env_buffer(dtype))
        self._done_bits = (torch.zeros((self._cap + 63) // 64, dtype=torch.int64, device=dev)[:(N + 63) // 64]
                           if done_bits else None)
        self._ep_return = self._partials = self._record = None
        if track_returns:
            self._ep_return = views[4]
            # one slot (4 doubles) per workgroup of the widest launch this batch can get: 4096 up to N = 2^22
            self._partial_slots = int(self._lib.fishing_partials_slots(max(N, self._cap)))
            self._partials = torch.zeros(4 * self._partial_slots, dtype=torch.float64, device=dev)
            self._record = torch.zeros(4, dtype=torch.float64, device=dev)
        self._action_buf = None
        self._scalar_views = None
        self._last_action = None
        self._cparams = self._pkey = self._cbuf = self._cparams_ref = self._cbuf_ref = None
        self._counter = None          # device-resident step counter (graph-replay mode), else host int
        self._want = torch.int32 if self.MODEL == MODEL_V0 else torch.float32
        if self._scalar:
            self._host_action = self._arena[self._action_off:self._action_off + 4].view(self._want)
            self._host_action_np = self._host_action.numpy()
            self._host_z = self._arena[self._action_off + 16:self._action_off + 16 + esz].view(dtype)
            self._host_z_np = self._host_z.numpy()
        elif self._host_mapped:
            self._host_action = torch.zeros(N, dtype=self._want).pin_memory()
            self._host_action_np = self._host_action.numpy()
        self._obs_view = self._obs.view(N, 1)
        self._done_view = self._done.view(torch.bool)
        self._info = {}
        if self._terminal_obs is not None:
            self._info["terminal_observation"] = self._terminal_obs.view(N, 1)
        if self._done_bits is not None:
            self._info["done_bits"] = self._done_bits
        self._launch = (int(launch_blocks), int(launch_threads))

        # reference attributes (base_fishing_env.py:27-46)
        self.reward = 0
        self.harvest = 0
        self.write_obj = open(file, "w+") if file is not None else None
        self._set_initial_state()

    # ------------------------------------------------------------------ parameters
    @property
    def years_passed(self):
        """base_fishing_env.py:75.  One env: the int.  num_envs: the live int32 year-counter tensor -- except for
        fishing-v4 in the derived-parameter mode, where the counter also dates each env's episode and with it its (K, r):
        there a COPY is returned, and an assignment (env.years_passed = ..., set_attr) first switches the env to stored
        r / K arrays, so that an outside write can move the Tmax check but never silently re-key an env's parameters."""
        if self._scalar:
            return self._years_scalar
        return self._t.clone() if self._derived else self._t

    @years_passed.setter
    def years_passed(self, v):
        if self._scalar:
            self._years_scalar = v
            return
        if v is self._t:
            return
        self._leave_derived_mode()
        self._t.copy_(torch.as_tensor(v).to(device=self.device, dtype=self._t.dtype).reshape(self.num_envs))

    @property
    def K(self):
        """Carrying capacity.  fishing-v4 with num_envs: the per-env tensor -- in the default derived-parameter mode a
        SNAPSHOT materialised by this access (the env keeps no K / r arrays: in-place edits of the returned tensor are
        lost, and every access costs a small kernel); assign through `env.K = ...` / `env.r = ...`, which switches the
        env to stored arrays until the next full reset()."""
        return self._K_view() if self._per_env else self.params["K"]

    @K.setter
    def K(self, v):
        self._set_param("K", v)

    @property
    def r(self):
        return self._r_view() if self._per_env else self.params["r"]

    @r.setter
    def r(self, v):
        self._set_param("r", v)

    @property
    def sigma(self):
        return self._sigma_arr if self._sigma_arr is not None else self._sigma_scalar

    @sigma.setter
    def sigma(self, v):
        if isinstance(v, (torch.Tensor, np.ndarray, list, tuple)):
            self._sigma_arr = self._per_env_buffer(self.dtype)
            self._sigma_arr.copy_(torch.as_tensor(v).to(device=self.device, dtype=self.dtype).reshape(self.num_envs))
        else:
            self._sigma_arr = None
            self._sigma_scalar = float(v)
            if self.MODEL in (MODEL_V0, MODEL_V1, MODEL_V2, MODEL_V4):
                # (the zoo's growth functions read their own params dict, which env.sigma never reaches in the
                # reference: growth_models.py:208-261)
                self.params["sigma"] = v

    def _per_env_buffer(self, dtype, fill=None):
        """A [N] device tensor with room for whole 1024-env tiles behind it (FISHING_FLAG_PADDED_TILES; self._cap == N
        when the batch needs no padding).  Every per-env stream handed to the kernels comes from here or from the arena."""
        buf = (torch.zeros(self._cap, dtype=dtype, device=self.device) if fill is None
               else torch.full((self._cap,), fill, dtype=dtype, device=self.device))
        return buf[:self.num_envs]

    def _set_param(self, name, v):
        if self._per_env:
            self._leave_derived_mode()
            arr = self._K_arr if name == "K" else self._r_arr
            if isinstance(v, (torch.Tensor, np.ndarray, list, tuple)):
                arr.copy_(torch.as_tensor(v).to(device=self.device, dtype=self.dtype).reshape(self.num_envs))
            else:
                arr.fill_(float(v))
        else:
            self.params[name] = v

    def _param_key(self):
        p = self.params
        if self.compact and self.Tmax > 254:
            raise ValueError("compact layout needs Tmax <= 254")
        ka, kp = self._KEY_ATTRS, self._KEY_PARAMS
        return (self.Tmax, self.init_state, self.auto_reset, self._derived, None if self._counter is not None else self._origin,
                self._sigma_scalar, p["r"], p["K"], self._launch,
                tuple([getattr(self, k) for k in ka]) if ka else ka, tuple([p.get(k) for k in kp]) if kp else kp)

    def _c_params(self):
        """FishingParams for the next call; rebuilt only when a source attribute changed
        (env.Tmax = ..., env.sigma = ..., env.init_state = ... are all legal in the reference)."""
        key = self._param_key()
        if key == self._pkey:
            return self._cparams
        p = self.params
        cp = _capi.FishingParams()
        cp.model = self.MODEL
        cp.n_actions = int(getattr(self, "n_actions", 0) or 0)
        cp.Tmax = int(self.Tmax)
        cp.flags = ((FLAG_AUTO_RESET if self.auto_reset else 0) | (FLAG_T_U8 if self.compact else 0)
                    | (FLAG_V4_DERIVED if self._derived else 0) | (FLAG_PADDED_TILES if self._padded else 0)
                    | (FLAG_RESET_COUNTER_ON_DEVICE if self._counter is not None else 0))
        cp.v4_origin_step, cp.v4_origin_counter = self._origin
        cp.r = float(p["r"])
        cp.K = float(p["K"])
        cp.sigma = self._sigma_scalar
        cp.C = float(getattr(self, "C", p.get("C", 0.5)))
        cp.x0 = float(self.init_state)
        cp.r_mean = float(getattr(self, "r_mean", p.get("r_mean", p["r"])))
        cp.K_mean = float(getattr(self, "K_mean", p.get("K_mean", p["K"])))
        cp.sigma_p = float(getattr(self, "sigma_p", p.get("sigma_p", 0.0)))
        cp.launch_blocks, cp.launch_threads = self._launch
        for k in ("M", "theta", "q", "b", "a", "alpha"):
            setattr(cp, k, float(p.get(k, 0.0) or 0.0))
        if self.MODEL == MODEL_V11:
            cp.n_models = len(self.models)
            for i, name in enumerate(self.models):
                cp.kinds[i] = KIND_OF_NAME[name]
            for name, d in self.model_params.items():
                g = cp.zoo[KIND_OF_NAME[name]]
                for k in ("r", "K", "sigma", "C", "M", "theta", "q", "b", "a"):
                    setattr(g, k, float(d.get(k, 0.0) or 0.0))
        self._cparams, self._pkey = cp, key
        self._cparams_ref = ctypes.byref(cp)        # (step() passes this: no temporary pointer object per call)
        return cp

    def _c_buffers(self, action=None, z_ext=None, with_outputs=True):
        ptr = lambda t: (t.data_ptr() if t is not None else None)  # noqa: E731
        return _capi.make_buffers(
            obs=ptr(self._obs), action=ptr(action), reward=ptr(self._reward) if with_outputs else None,
            done=ptr(self._done) if with_outputs else None, done_bits=ptr(self._done_bits), t=ptr(self._t),
            r=ptr(self._r_arr), K=ptr(self._K_arr), sigma=ptr(self._sigma_arr), z_ext=ptr(z_ext),
            terminal_obs=ptr(self._terminal_obs), ep_return=ptr(self._ep_return),
            return_partials=ptr(self._partials), model_idx=ptr(self._model_idx), counter=ptr(self._counter),
            v4_stamp=ptr(self._stamp))

    def _step_buffers(self, action_ptr, z_ptr):
        """The step() FishingBuffers: built once (the env's tensors never move), only the
        action / z_ext / sigma pointers are refreshed per call."""
        b = self._cbuf
        if b is None:
            b = self._cbuf = self._c_buffers()
            self._cbuf_ref = ctypes.byref(b)
        b.action = action_ptr
        b.z_ext = z_ptr
        b.sigma = self._sigma_arr.data_ptr() if self._sigma_arr is not None else None
        return self._cbuf_ref

    def _stream(self):
        return _current_stream_ptr(self.device.index)

    # ------------------------------------------------------------------ state
    def _set_initial_state(self):
        """Constructor state (base_fishing_env.py:33-46): obs = x0 / K - 1, t = 0."""
        self._obs.fill_(float(self.init_state) / float(self.params["K"]) - 1.0)
        self._t.zero_()
        self._publish_scalar_state()

    def _host_sync(self):
        rc = self._lib.fishing_stream_synchronize(self._stream())
        if rc:
            _capi.check(rc, "fishing_stream_synchronize")

    def _numpy_redraw(self):
        """rng="numpy": the per-episode draws of fishing-v4 / fishing-v11 from NumPy's global stream, in the
        reference's order, replacing what the reset kernel drew from Philox."""
        if not self._np_rng:
            return
        if self.MODEL == MODEL_V4:
            K = float(np.clip(np.random.normal(self.K_mean, self.sigma_p), 0, 1e6))
            r = float(np.clip(np.random.normal(self.r_mean, self.sigma_p), 0, 1e6))
            self._K_arr.fill_(K)
            self._r_arr.fill_(r)
        elif self.MODEL == MODEL_V11:
            self._model_idx.fill_(KIND_OF_NAME[str(np.random.choice(self.models))])

    def _read_scalar(self):
        """obs, t, reward, done of the single env, read straight from the pinned arena."""
        rc = self._lib.fishing_stream_synchronize(self._stream())   # the kernels wrote pinned host memory
        if rc:
            _capi.check(rc, "fishing_stream_synchronize")
        v = self._scalar_views
        if v is None:       # NumPy views of the env's four cells in the pinned arena, made once
            host = self._arena_np
            o_obs, o_t, o_rew, o_done = self._arena_offs[:4]
            ftype = np.float64 if self.dtype == torch.float64 else np.float32
            w = np.dtype(ftype).itemsize
            v = self._scalar_views = (host[o_obs:o_obs + w].view(ftype), host[o_t:o_t + 4].view(np.int32),
                                      host[o_rew:o_rew + w].view(ftype), host[o_done:o_done + 1])
        return v[0].astype(np.float64), int(v[1][0]), float(v[2][0]), bool(v[3][0])

    def _publish_scalar_state(self):
        if self._scalar:
            self.state, self._years_scalar, self._last_reward, self._last_done = self._read_scalar()
            self.fish_population = float((self.state[0] + 1.0) * float(self._K_view() if self._per_env else self.params["K"]))
        else:
            self.state = self._obs_view
            self.reward = self._reward

    def seed(self, seed=None):
        """The reference has no seed() (base_fishing_env.py:13); this keys the Philox streams and, for
        rng="numpy", seeds NumPy's global stream the way a user of the reference would (np.random.seed)."""
        self._leave_derived_mode()          # the parameters in force were drawn under the old seed / counters
        self._seed = int(0 if seed is None else seed) & 0xFFFFFFFFFFFFFFFF
        if self._np_rng and seed is not None:
            np.random.seed(int(seed) & 0xFFFFFFFF)
        self._step_count = 0
        self._reset_count = 0
        if self._counter is not None:
            self._counter.zero_()
        return [self._seed]

    def reset(self, mask=None, *, seed=None, options=None):
        """base_fishing_env.py:83-91 (v4: fishing_model_error.py:41-48).  `mask` (bool[N]) resets
        a subset -- what a VecEnv wrapper without in-kernel auto-reset would call.  `seed` / `options`
        are accepted for callers written against the newer gym signature (seed -> self.seed(seed));
        the return value stays the reference's: the observation alone."""
        if seed is not None:
            self.seed(seed)
        m = None
        if mask is not None:
            self._begin_masked_reset()          # (fishing-v4 derived: per-env origin stamps from here on)
            m = torch.as_tensor(mask).to(device=self.device).reshape(self.num_envs).to(torch.uint8).contiguous()
        elif self._derived_capable:
            self._enter_derived_mode_at_full_reset()
        with torch.cuda.device(self.device):
            # (graph-replay mode: the reset counter is the device word counter[3], which the library reads and bumps itself -- a
            # captured reset() draws fresh fishing-v4 parameters / fishing-v11 models at every replay)
            rc = self._fn_reset(self._c_params(), self.num_envs, self.env_offset,
                                self._c_buffers(with_outputs=False), m.data_ptr() if m is not None else None,
                                self._seed, 0 if self._counter is not None else self._reset_count, self._stream())
        _capi.check(rc, "fishing_reset")
        self._reset_count += 1
        if mask is None and self._stamp is not None:        # (cleared by the kernel: the launches go back to the stamp-free forms)
            self._stamp, self._cbuf = None, None
        self._numpy_redraw()
        if self._host_mapped and not self._scalar:
            self._host_sync()
        if self._scalar:
            self.reward = 0 if self.MODEL != MODEL_V4 else self.reward   # v4 leaves it (quirk B8)
            self.harvest = 0
        self._publish_scalar_state()
        return self.state

    # ------------------------------------------------------------------ step
    def _prepare_action(self, action):
        N = self.num_envs
        want = self._want
        if self._scalar:                      # write the one action into the pinned buffer the kernel reads
            if isinstance(action, torch.Tensor):
                action = action.detach().cpu().numpy()
            a = np.asarray(action).reshape(-1)
            if a.size != 1:
                raise ValueError("expected 1 action, got shape %s" % (np.shape(action),))
            self._host_action_np[0] = a[0]
            return self._host_action
        if self._host_mapped:                 # N actions into the pinned buffer the kernel reads over PCIe
            if isinstance(action, torch.Tensor):
                action = action.detach().cpu().numpy()
            a = np.asarray(action)
            if a.size != N:
                raise ValueError("expected %d actions, got shape %s" % (N, np.shape(action)))
            self._host_action_np[:] = a.reshape(N)
            return self._host_action
        if isinstance(action, torch.Tensor):
            a = action
            # fast path: already the stream the kernel reads
            if a.dtype == want and a.device == self.device and a.numel() == N and a.is_contiguous() \
                    and not a.data_ptr() & 15:
                return a
            if a.device != self.device:
                a = a.to(self.device)
        else:
            a = torch.as_tensor(np.asarray(action), device=self.device)
        if a.numel() != N:
            raise ValueError("expected %d actions, got shape %s" % (N, tuple(a.shape)))
        a = a.reshape(N)
        if a.dtype != want:
            a = a.to(want)
        if not a.is_contiguous() or a.data_ptr() % 16:
            if self._action_buf is None:
                self._action_buf = torch.empty(N, dtype=want, device=self.device)
            self._action_buf.copy_(a)
            a = self._action_buf
        return a

    def step(self, action, noise=None):
        """base_fishing_env.py:60-81.  `noise` (optional, [N] standard normals) replaces the
        in-kernel Philox stream -- the external-noise parity mode."""
        a = self._prepare_action(action)
        z = None
        if noise is not None:
            z = torch.as_tensor(noise).to(device=self.device, dtype=self.dtype).reshape(self.num_envs).contiguous()
        elif self._np_rng:
            if self._scalar:
                self._host_z_np[0] = np.random.normal(0, 1)   # the reference's draw, from the global stream
                z = self._host_z
            else:                                             # N reference envs stepped in order (DummyVecEnv)
                z = torch.as_tensor(np.random.normal(0, 1, self.num_envs)).to(device=self.device, dtype=self.dtype)
        bufs = self._step_buffers(a.data_ptr(), z.data_ptr() if z is not None else None)
        on_device = self._counter is not None
        host_count = 0 if on_device else self._step_count
        self._c_params()                      # (rebuilt only when a source attribute changed; the call passes its cached reference)
        dev = self._dev_index
        if torch.cuda.current_device() == dev:
            stream = _current_stream_ptr(dev)
            rc = self._fn_step(self._cparams_ref, self.num_envs, self.env_offset, bufs, self._seed, host_count, stream)
            if on_device and not rc:
                rc = self._lib.fishing_counter_add(self._counter.data_ptr(), 1, stream)
        else:
            with torch.cuda.device(self.device):
                stream = self._stream()
                rc = self._fn_step(self._cparams_ref, self.num_envs, self.env_offset, bufs, self._seed, host_count,
                                   stream)
                if on_device and not rc:
                    rc = self._lib.fishing_counter_add(self._counter.data_ptr(), 1, stream)
        if rc:
            _capi.check(rc, "fishing_step")
        self._step_count += 1
        self._last_action = a
        if self._scalar:
            return self._step_result()
        if self._host_mapped:                 # the results are in host memory: valid once the stream has drained
            self._host_sync()
        return self._obs_view, self._reward, self._done_view, self._info

    def _step_result(self):
        if self._scalar:
            self._publish_scalar_state()
            self.reward = self._last_reward
            self.harvest = self.reward
            return self.state, self.reward, self._last_done, {}
        return self._obs_view, self._reward, self._done_view, self._info

    def step_many(self, actions, n_steps=None, fused=False, rewards_out=None, dones_out=None):
        """n_steps consecutive step() calls enqueued by one C call; `actions` is [R, N]
        (a ring of R action batches, cycled).  Returns the last step's result.
        fused=True runs them in ONE kernel launch (fishing_step_fused_*: state in registers, action rows
        prefetched) -- bit-identical results, and the fast path while a launch per step is latency-bound
        (N <= 2^20); `rewards_out` [n_steps, N] (env dtype) / `dones_out` [n_steps, N] (uint8 or bool) then receive every
        step's reward / done rows.  Not available with terminal-observation / done_bits records or rng="numpy"."""
        want = torch.int32 if self.MODEL == MODEL_V0 else torch.float32
        if not (isinstance(actions, torch.Tensor) and actions.device == self.device and actions.dtype == want
                and actions.dim() == 2 and actions.shape[1] == self.num_envs and actions.stride(1) == 1
                and actions.stride(0) >= self.num_envs and actions.stride(0) % 4 == 0 and actions.data_ptr() % 16 == 0):
            raise ValueError("actions must be a [R, %d] %s tensor on %s with unit inner stride and a row stride "
                             "that is a multiple of 4 elements" % (self.num_envs, want, self.device))
        R = actions.shape[0]
        row_stride = actions.stride(0) if R > 1 else self.num_envs
        n_steps = R if n_steps is None else int(n_steps)
        if self._np_rng:                    # the draws come from NumPy's host-side stream: one step() per step
            for k in range(n_steps):
                self.step(actions[k % R])
            return self._step_result()
        if (rewards_out is not None or dones_out is not None) and not fused:
            raise ValueError("rewards_out / dones_out need fused=True")
        with torch.cuda.device(self.device):
            count = 0 if self._counter is not None else self._step_count
            if fused:
                out_stride = 0
                for name, o, ok in (("rewards_out", rewards_out, (self.dtype,)), ("dones_out", dones_out, (torch.uint8, torch.bool))):
                    if o is None:
                        continue
                    if not (isinstance(o, torch.Tensor) and o.device == self.device and o.dtype in ok and o.dim() == 2
                            and o.shape[0] >= n_steps and o.shape[1] == self.num_envs and o.stride(1) == 1
                            and o.stride(0) % 16 == 0 and o.data_ptr() % 16 == 0
                            and (out_stride in (0, o.stride(0)))):
                        raise ValueError("%s must be a [>= %d, %d] device tensor with unit inner stride and a row stride "
                                         "that is a multiple of 16 elements (the same for both outputs)"
                                         % (name, n_steps, self.num_envs))
                    out_stride = o.stride(0)
                rc = self._fn_step_fused(self._c_params(), self.num_envs, self.env_offset, self._c_buffers(actions),
                                         row_stride, R, n_steps, rewards_out.data_ptr() if rewards_out is not None else None,
                                         dones_out.data_ptr() if dones_out is not None else None, out_stride,
                                         self._seed, count, self._stream())
            else:
                rc = self._fn_step_many(self._c_params(), self.num_envs, self.env_offset, self._c_buffers(actions),
                                        row_stride, R, n_steps, self._seed, count, self._stream())
            if self._counter is not None and not rc:
                rc = self._lib.fishing_counter_add(self._counter.data_ptr(), n_steps, self._stream())
        _capi.check(rc, "fishing_step_fused" if fused else "fishing_step_many")
        self._step_count += n_steps
        self._last_action = actions[(n_steps - 1) % R] if n_steps else self._last_action
        return self._step_result()

    # SB3 VecEnv protocol pieces the reference's helpers use (shared_env.py:15-26,57-79)
    def step_async(self, actions):
        self._pending = actions

    def step_wait(self):
        return self.step(self._pending)

    def get_attr(self, name, indices=None):
        v = getattr(self, name)
        if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == self.num_envs:
            idx = range(self.num_envs) if indices is None else ([indices] if isinstance(indices, int) else indices)
            return [v[i] for i in idx]
        n = self.num_envs if indices is None else (1 if isinstance(indices, int) else len(indices))
        return _Repeat(v, n)

    def set_attr(self, name, value, indices=None):
        setattr(self, name, value)

    def env_method(self, name, *args, indices=None, **kwargs):
        out = getattr(self, name)(*args, **kwargs)
        n = self.num_envs if indices is None else (1 if isinstance(indices, int) else len(indices))
        return _Repeat(out, n)

    # ------------------------------------------------------------------ fused rollout
    def rollout(self, n_steps, policy="random", param=0.0, record=False):
        """n_steps of step() inside one kernel with an in-kernel policy (csrc/fishing_rollout.hip).
        record=True returns the [n_steps, 4, N] table {obs_in, action, reward, done}.  The fused kernel always
        draws from the Philox streams keyed by `seed` (also for an env built with rng="numpy": a kernel cannot
        consume NumPy's host-side stream).
        `param` as a tensor of N values: one policy parameter per env (fishing_rollout_params_*) -- env i escapes to
        S = param[i] / fishes the quota param[i] / takes the action param[i]: N fishing-v4 or fishing-v11 envs, each with
        the S its own BMSY() found (policies.escapement(env).kernel_policy hands that tensor over).  Needs auto-reset."""
        if isinstance(policy, tuple):                 # ("constant", a) or a policies.* kernel_policy pair
            policy, param = policy
        pol = POLICIES[policy] if isinstance(policy, str) else int(policy)
        per_env = None
        if isinstance(param, torch.Tensor) and param.numel() > 1:
            if param.numel() != self.num_envs:
                raise ValueError("a per-env policy parameter needs one value per env (%d), got %d" % (self.num_envs, param.numel()))
            if not self.auto_reset:
                raise ValueError("a per-env policy parameter needs auto_reset=True (the frozen-episode rollout takes a scalar)")
            per_env = param.to(device=self.device, dtype=self.dtype).reshape(-1).contiguous()
            if per_env.data_ptr() % 16:         # (a view into the middle of a tensor: the ABI wants 16-byte alignment)
                per_env = per_env.clone()
            param = 0.0
        elif isinstance(param, torch.Tensor):
            param = float(param.reshape(-1)[0])
        if not self.auto_reset:
            # a rollout without auto-reset freezes finished envs: their year counters stop dating their episodes
            self._leave_derived_mode()
        traj = None
        if record:
            if self.num_envs % 4:
                raise ValueError("record=True needs num_envs % 4 == 0")
            traj = torch.empty((int(n_steps), 4, self.num_envs), dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            if per_env is not None:
                rc = self._fn_rollout_params(self._c_params(), self.num_envs, self.env_offset, self._c_buffers(), pol,
                                             per_env.data_ptr(), int(n_steps), traj.data_ptr() if traj is not None else None,
                                             self._seed, 0 if self._counter is not None else self._step_count, self._stream())
            else:
                rc = self._fn_rollout(self._c_params(), self.num_envs, self.env_offset, self._c_buffers(),
                                      pol, float(param), int(n_steps), traj.data_ptr() if traj is not None else None,
                                      self._seed, 0 if self._counter is not None else self._step_count, self._stream())
            if self._counter is not None and not rc:
                rc = self._lib.fishing_counter_add(self._counter.data_ptr(), int(n_steps), self._stream())
        _capi.check(rc, "fishing_rollout")
        self._step_count += int(n_steps)
        if self._scalar:
            self._publish_scalar_state()
        return traj

    def episode_record(self, all_reduce=True, copy=False):
        """The episodic-return record {sum R, sum R^2, n, sum length} as a 4-double DEVICE tensor: reduced on the
        device in slot order and -- when torch.distributed is initialised -- summed across ranks (one RCCL
        all-reduce).  Everything is enqueued on the current stream; nothing here waits for the GPU (with a
        non-RCCL backend the all-reduce goes through a 32-byte host copy).
        The tensor returned is the env's own scratch: VALID UNTIL THE NEXT episode_record() / episode_stats() call,
        which rewrites it.  `copy=True` returns a private copy (one 32-byte device copy on the same stream) for
        callers that keep records of several points of a run side by side."""
        if self._partials is None:
            raise RuntimeError("construct the env with track_returns=True")
        with torch.cuda.device(self.device):
            rc = self._lib.fishing_reduce_returns_slots(self._partials.data_ptr(), self._partial_slots, self._record.data_ptr(),
                                                        self._stream())
        _capi.check(rc, "fishing_reduce_returns_slots")
        from .sharding import all_reduce_record
        rec = self._record              # scratch: rewritten from the partials by every call, so reduced in place
        if all_reduce:
            rec = all_reduce_record(rec)
        return rec.clone() if copy else rec

    def episode_stats(self, all_reduce=True):
        """episode_record() read back to the host, with mean / std of the return and mean episode length."""
        from .sharding import summarize_record
        return summarize_record(self.episode_record(all_reduce))

    def step_kernel_name(self, actions=None):
        """Diagnostic: the kernel (as rocprofv3 names it) that step() launches for this env's whole tiles."""
        import ctypes
        out = ctypes.create_string_buffer(160)
        a = self._obs if actions is None else actions         # any aligned device pointer: nothing is launched
        fn = getattr(self._lib, "fishing_step_kernel_name_" + self._suffix)
        _capi.check(fn(self._c_params(), self.num_envs, self._c_buffers(a), out, 160), "fishing_step_kernel_name")
        return out.value.decode()

    # ------------------------------------------------------------------ render / close
    def render(self, mode="human", index=0):
        """base_fishing_env.py:93-94 -> shared_env.py:8-12.  The reference's version raises
        (self.action is never assigned, quirk B10); this one reports the last action."""
        a = None if self._last_action is None else self._last_action[index].item()
        rew = self._reward[index].item()
        row = [int(self._t[index]), float(self._obs[index]), a, rew]
        if self.write_obj is not None:
            csv.writer(self.write_obj).writerow(row)
        return row

    def close(self):
        if self.write_obj is not None:
            self.write_obj.close()
            self.write_obj = None


class FishingEnv(BaseFishingEnv):
    """fishing-v0 (fishing_env.py:6-24): Discrete(n_actions), quota = a / n_actions * K."""
    MODEL = MODEL_V0
    _KEY_ATTRS = ("n_actions",)

    def __init__(self, r=0.3, K=1, sigma=0.0, n_actions=100, init_state=0.75, Tmax=100, file=None, **vec):
        self.n_actions = int(n_actions)
        super().__init__(params={"r": r, "K": K, "sigma": sigma, "x0": init_state}, Tmax=Tmax, file=file, **vec)
        _, Discrete = space_classes()
        self.action_space = Discrete(self.n_actions)


class FishingCtsEnv(BaseFishingEnv):
    """fishing-v1 (fishing_cts_env.py:4-12): continuous action, logistic growth."""
    MODEL = MODEL_V1

    def __init__(self, r=0.3, K=1, sigma=0.0, init_state=0.75, Tmax=100, file=None, **vec):
        super().__init__(params={"r": r, "K": K, "sigma": sigma, "x0": init_state}, Tmax=Tmax, file=file, **vec)


class FishingTippingEnv(BaseFishingEnv):
    """fishing-v2 (fishing_tipping_env.py:6-35): tipping-point growth with parameter C."""
    MODEL = MODEL_V2
    _KEY_ATTRS = ("C",)

    def __init__(self, r=0.3, K=1, C=0.5, sigma=0.0, init_state=0.75, Tmax=100, file=None, **vec):
        self.C = C
        super().__init__(params={"r": r, "K": K, "sigma": sigma, "C": C, "x0": init_state}, Tmax=Tmax,
                         file=file, **vec)


class FishingModelError(BaseFishingEnv):
    """fishing-v4 (fishing_model_error.py:6-48): K, r ~ N(mean, sigma_p) clipped to [0, 1e6],
    redrawn per env at construction and at every reset; reset obs is x0 un-normalised."""
    MODEL = MODEL_V4
    _KEY_ATTRS = ("K_mean", "r_mean", "sigma_p")

    def __init__(self, K_mean=1.0, r_mean=0.3, price=1.0, sigma=0.0, sigma_p=0.1, init_state=0.75, Tmax=100,
                 file=None, **vec):
        self.K_mean, self.r_mean, self.sigma_p, self.price = K_mean, r_mean, sigma_p, price
        super().__init__(params={"r": r_mean, "K": K_mean, "sigma": sigma, "r_mean": r_mean, "K_mean": K_mean,
                                 "sigma_p": sigma_p, "x0": init_state}, Tmax=Tmax, file=file, **vec)

    def _set_initial_state(self):
        # constructor: draw (K, r) once (fishing_model_error.py:37-38) but keep the base
        # class's obs = x0 / K_mean - 1 (base_fishing_env.py:46) until the first reset()
        self._set_origin(self._step_count, self._reset_count)
        with torch.cuda.device(self.device):
            rc = self._fn_reset(self._c_params(), self.num_envs, self.env_offset,
                                self._c_buffers(with_outputs=False), None, self._seed, self._reset_count,
                                self._stream())
        _capi.check(rc, "fishing_reset")
        self._reset_count += 1
        self._numpy_redraw()
        if self._host_mapped:  # the arena is host memory here: let the reset kernel land before overwriting
            torch.cuda.current_stream(self.device).synchronize()
        self._obs.fill_(float(self.init_state) / float(self.K_mean) - 1.0)
        self._publish_scalar_state()


# ---------------------------------------------------------------------------------------------
# Growth-model zoo, fishing-v5..v11 (gym_fishing/envs/growth_models.py:6-204): same env core,
# lognormal process noise; one kernel instantiation per growth function (per-env switch for
# fishing-v11).  step(), reset(), rollout() and simulate() all work as for fishing-v0..v4.
# ---------------------------------------------------------------------------------------------
class Allen(BaseFishingEnv):
    """fishing-v5 (growth_models.py:6-25; allen() :208-217)."""
    MODEL = MODEL_V5
    _KEY_PARAMS = ("C",)

    def __init__(self, r=0.3, K=1, C=0.5, sigma=0.0, init_state=0.75, Tmax=100, file=None, **vec):
        super().__init__(params={"r": r, "K": K, "sigma": sigma, "C": C, "x0": init_state}, Tmax=Tmax, file=file, **vec)


class BevertonHolt(BaseFishingEnv):
    """fishing-v6 (growth_models.py:28-40; beverton_holt() :220-226)."""
    MODEL = MODEL_V6

    def __init__(self, r=0.3, K=1, sigma=0.0, init_state=0.75, Tmax=100, file=None, **vec):
        super().__init__(params={"r": r, "K": K, "sigma": sigma, "x0": init_state}, Tmax=Tmax, file=file, **vec)


class Myers(BaseFishingEnv):
    """fishing-v8 (growth_models.py:43-70; myers() :247-255)."""
    MODEL = MODEL_V8
    _KEY_PARAMS = ("theta", "M")

    def __init__(self, r=1.0, K=1.0, M=1.0, theta=3.0, sigma=0.0, init_state=1.5, Tmax=100, file=None, **vec):
        super().__init__(params={"r": r, "K": K, "sigma": sigma, "theta": theta, "M": M, "x0": init_state},
                         Tmax=Tmax, file=file, **vec)


class May(BaseFishingEnv):
    """fishing-v7 (growth_models.py:75-108; may() :229-242)."""
    MODEL = MODEL_V7
    _KEY_PARAMS = ("q", "b", "a", "M")

    def __init__(self, r=0.7, K=1.5, M=1.5, q=3, b=0.15, sigma=0.0, a=0.2, init_state=0.75, Tmax=100, file=None,
                 **vec):
        super().__init__(params={"r": r, "K": K, "sigma": sigma, "q": q, "b": b, "a": a, "M": M, "x0": init_state},
                         Tmax=Tmax, file=file, **vec)


class Ricker(BaseFishingEnv):
    """fishing-v9 (growth_models.py:111-123; ricker() :258-261)."""
    MODEL = MODEL_V9

    def __init__(self, r=0.3, K=1, sigma=0.0, init_state=0.75, Tmax=100, file=None, **vec):
        super().__init__(params={"r": r, "K": K, "sigma": sigma, "x0": init_state}, Tmax=Tmax, file=file, **vec)


class NonStationary(BaseFishingEnv):
    """fishing-v10 (growth_models.py:126-154): Beverton-Holt whose r drifts by alpha at every
    population draw and is never restored by reset() -- r is per-env state here."""
    MODEL = MODEL_V10
    _KEY_PARAMS = ("alpha",)

    def __init__(self, r=0.8, K=1, sigma=0.0, alpha=-0.007, init_state=0.75, Tmax=100, file=None, **vec):
        super().__init__(params={"r": r, "K": K, "sigma": sigma, "alpha": alpha, "x0": init_state}, Tmax=Tmax,
                         file=file, **vec)

    @property
    def r(self):
        return float(self._r_arr[0]) if self._scalar else self._r_arr

    @r.setter
    def r(self, v):
        if isinstance(v, (torch.Tensor, np.ndarray, list, tuple)):
            self._r_arr.copy_(torch.as_tensor(v).to(device=self.device, dtype=self.dtype).reshape(self.num_envs))
        else:
            self._r_arr.fill_(float(v))


_V11_DEFAULT_PARAMS = {
    "allen": {"r": 0.3, "K": 1.0, "sigma": 0.0, "C": 0.5, "x0": 0.75},
    "beverton_holt": {"r": 0.3, "K": 1, "sigma": 0.0, "x0": 0.75},
    "myers": {"r": 1.0, "K": 1.0, "M": 1.0, "theta": 3.0, "sigma": 0.0, "x0": 1.5},
    "may": {"r": 0.7, "K": 1.5, "M": 1.5, "q": 3, "b": 0.15, "sigma": 0.0, "a": 0.2, "x0": 0.75},
    "ricker": {"r": 0.3, "K": 1, "sigma": 0.0, "x0": 0.75},
}


class ModelUncertainty(BaseFishingEnv):
    """fishing-v11 (growth_models.py:157-204): the growth function is one of `models`, redrawn
    per env at every reset.  The env core keeps the base defaults (K = 1, x0 = 0.75) for the
    obs / quota maps, as the reference does; `model_params` holds the per-model dicts (the
    reference's `self.params`), `model` / `model_idx` the kind in force."""
    MODEL = MODEL_V11

    def __init__(self, models=("allen", "beverton_holt", "myers", "may", "ricker"), params=None, Tmax=100,
                 file=None, **vec):
        self.models = list(models)
        if not 1 <= len(self.models) <= 5 or any(m not in KIND_OF_NAME for m in self.models):
            raise ValueError("models must be 1..5 of %s" % sorted(KIND_OF_NAME))
        self.model_params = {k: dict(v) for k, v in (_V11_DEFAULT_PARAMS if params is None else params).items()}
        super().__init__(Tmax=Tmax, file=file, **vec)

    def _param_key(self):
        # (dict order, no sorting: GraphedSteps.replay() compares this on every replay; a reordered dict only costs a re-capture)
        return super()._param_key() + (tuple(self.models), tuple((k, tuple(v.items())) for k, v in self.model_params.items()))

    def _set_initial_state(self):
        # constructor: choose a model (growth_models.py:187); obs as the base class sets it
        with torch.cuda.device(self.device):
            rc = self._fn_reset(self._c_params(), self.num_envs, self.env_offset,
                                self._c_buffers(with_outputs=False), None, self._seed, self._reset_count,
                                self._stream())
        _capi.check(rc, "fishing_reset")
        self._reset_count += 1
        self._numpy_redraw()
        self._publish_scalar_state()

    @property
    def model_idx(self):
        return self._model_idx

    @property
    def model(self):
        names = {v: k for k, v in KIND_OF_NAME.items()}
        if self._scalar:
            return names[int(self._model_idx[0])]
        return [names[int(k)] for k in self._model_idx.tolist()]
