"""gym_fishing_amd -- MI355X-native vectorised fisheries gym.

Drop-in for the rollout path of boettiger-lab/gym_fishing: the ids fishing-v0/v1/v2/v4,
the gym.Env reset()/step()/render() surface and the constructor kwargs of the reference
(gym_fishing/envs/__init__.py:17-71; v5..v11 = the growth-model zoo), with step()/reset() executed by hand-written HIP
kernels for gfx950 behind a C ABI (include/fishing_hip.h).

    import gym_fishing_amd as gf
    env = gf.make("fishing-v1", sigma=0.1, num_envs=1 << 22)   # N envs in lockstep on the GPU
    obs = env.reset()
    obs, reward, done, info = env.step(actions)                # torch tensors, on device

    env = gf.make("fishing-v1")                                # the reference's scalar protocol

Importing this package does not load the HIP library; constructing an env does, and
raises FishingLibraryError if the library or a HIP device is missing.
"""
import os as _os


def ensure_dmabuf_ipc():
    """RCCL (and any cross-process sharing of device memory) needs dmabuf IPC on this driver: HSA_ENABLE_IPC_MODE_LEGACY=0,
    read when HIP initialises.  Called by the multi-rank entry points (sharding.make_sharded, bench.py) -- importing the
    package does not touch the process environment.  An explicit choice of the caller wins; returns False, with a warning,
    when the variable had to be set but HIP is already up (too late for this process: export it before starting)."""
    if "HSA_ENABLE_IPC_MODE_LEGACY" in _os.environ:
        return True
    _os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    try:
        import torch
        late = torch.cuda.is_initialized()
    except Exception:  # noqa: BLE001
        late = False
    if late:
        import warnings
        warnings.warn("HSA_ENABLE_IPC_MODE_LEGACY=0 set after HIP was initialised: it does not take effect in this process; "
                      "export it before launching multi-GPU ranks")
    return not late


from ._capi import FishingLibraryError  # noqa: E402, F401

__version__ = "0.1.0"

# id -> class name in .envs (gym_fishing/envs/__init__.py:17-71)
ENTRY_POINTS = {
    "fishing-v0": "FishingEnv",
    "fishing-v1": "FishingCtsEnv",
    "fishing-v2": "FishingTippingEnv",
    "fishing-v4": "FishingModelError",
    "fishing-v5": "Allen",
    "fishing-v6": "BevertonHolt",
    "fishing-v7": "May",
    "fishing-v8": "Myers",
    "fishing-v9": "Ricker",
    "fishing-v10": "NonStationary",
    "fishing-v11": "ModelUncertainty",
}
ENV_IDS = tuple(ENTRY_POINTS)


def env_class(env_id):
    if env_id not in ENTRY_POINTS:
        raise KeyError("unknown env id %r; this build provides %s" % (env_id, ", ".join(ENV_IDS)))
    from . import envs
    return getattr(envs, ENTRY_POINTS[env_id])


def make(env_id, api="gym", **kwargs):
    """gym.make(id, **ctor_kwargs) equivalent.  Extra kwargs: num_envs, device, seed, dtype,
    auto_reset, env_offset, record_terminal_obs, track_returns, done_bits, compact.
    api="gym" (default): the reference's protocol -- reset() -> obs, step() -> (obs, reward, done, info).
    api="gymnasium": reset() -> (obs, info), step() -> (obs, reward, terminated, truncated, info) with
    truncated = years_passed > Tmax and terminated = fish_population <= 0 (gym_fishing_amd.gymnasium_api)."""
    if api == "gym":
        return env_class(env_id)(**kwargs)
    if api != "gymnasium":
        raise ValueError("api must be 'gym' or 'gymnasium', not %r" % (api,))
    env_class(env_id)       # unknown ids fail here, with the list of ids
    from .gymnasium_api import GymnasiumFishingEnv
    return GymnasiumFishingEnv(env_id, **kwargs)


def register_with_gym():
    """Register the ids with gym and / or gymnasium when importable (never required): the old `gym` gets the
    reference's 4-tuple classes (gym_fishing/envs/__init__.py:17-35), gymnasium -- whose checker and wrappers reject a
    4-tuple step() -- the 5-tuple GymnasiumFishingEnv around them."""
    done = []
    for mod in ("gymnasium", "gym"):
        try:
            reg = __import__(mod + ".envs.registration", fromlist=["register"])
        except Exception:  # noqa: BLE001 - optional dependency
            continue
        for env_id, cls in ENTRY_POINTS.items():
            try:
                if mod == "gymnasium":
                    reg.register(id=env_id, entry_point="gym_fishing_amd.gymnasium_api:GymnasiumFishingEnv",
                                 kwargs={"env_id": env_id})
                else:
                    reg.register(id=env_id, entry_point="gym_fishing_amd.envs:" + cls)
                done.append((mod, env_id))
            except Exception:  # noqa: BLE001 - already registered
                pass
    return done


# like the reference (gym_fishing/__init__.py:2), importing the package registers the ids -- but
# only if a gym is there to register with; never a hard dependency
try:
    register_with_gym()
except Exception:  # noqa: BLE001
    pass


def __getattr__(name):
    if name in ("FishingEnv", "FishingCtsEnv", "FishingTippingEnv", "FishingModelError", "BaseFishingEnv", "Allen",
                "BevertonHolt", "May", "Myers", "Ricker", "NonStationary", "ModelUncertainty"):
        from . import envs
        return getattr(envs, name)
    if name in ("make_vec_env", "FishingVecEnv"):       # NumPy / SB3 VecEnv adapter
        from . import vec_env
        return getattr(vec_env, name)
    raise AttributeError(name)
