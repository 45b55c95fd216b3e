"""ctypes loader for the plain-C oracle (oracle/fishing_oracle.c) -- TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# FISHING_ORACLE_LIB: another build of the same source, e.g. `make -C oracle sanitize` (ASan + UBSan; run the tests with
# LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0)
LIB = os.environ.get("FISHING_ORACLE_LIB") or os.path.join(HERE, "_build", "libfishing_oracle.so")

c_i32, c_i64, c_u64, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_void_p
_lib = None


def build(force=False):
    src = os.path.join(HERE, "fishing_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B"] if force else ["make", "-C", HERE], check=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        h = ctypes.CDLL(LIB)
        for name, real in (("oracle_step_f64", ctypes.c_double), ("oracle_step_f32", ctypes.c_float)):
            fn = getattr(h, name)
            fn.restype = None
            fn.argtypes = [c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, real, real, real, real,
                           c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]
        h.oracle_noise_f32.restype = None
        h.oracle_noise_f32.argtypes = [c_i64, c_u64, c_u64, c_u64, c_vp, c_vp]
        h.oracle_rollout_random_f32.restype = ctypes.c_double
        h.oracle_rollout_random_f32.argtypes = [c_i32, c_i64, c_u64, c_i32, c_vp, c_vp, ctypes.c_float,
                                                ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                c_i32, c_i32, c_u64, c_u64, c_i32]
        h.oracle_max_threads.restype = c_i32
        _lib = h
    return _lib


def _p(a):
    return a.ctypes.data_as(c_vp) if a is not None else None


def step(model, obs, t, action, z, r, K, sigma, C=0.5, Tmax=100, n_actions=100, dtype=np.float64):
    """Same contract as fishing_oracle.step (arrays r/K/sigma or scalars)."""
    dtype = np.dtype(dtype)
    n = len(obs)
    obs = np.ascontiguousarray(obs, dtype=dtype)
    t = np.ascontiguousarray(t, dtype=np.int32)
    action = np.ascontiguousarray(action, dtype=np.int32 if model == 0 else np.float32)
    z = np.ascontiguousarray(z, dtype=dtype)
    arr = lambda v: np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=dtype), (n,)))  # noqa: E731
    r_a, K_a, s_a = arr(r), arr(K), arr(sigma)
    o = np.empty(n, dtype)
    rew = np.empty(n, dtype)
    done = np.empty(n, np.uint8)
    t2 = np.empty(n, np.int32)
    fn = lib().oracle_step_f64 if dtype == np.float64 else lib().oracle_step_f32
    fn(model, n, _p(obs), _p(t), _p(action), _p(z), _p(r_a), _p(K_a), _p(s_a), 0.0, 0.0, 0.0, float(C), Tmax,
       n_actions, _p(o), _p(rew), _p(done), _p(t2))
    return o, rew, done, t2


def noise(n, env_offset, seed, counter):
    z = np.empty(n, np.float32)
    a = np.empty(n, np.float32)
    lib().oracle_noise_f32(n, env_offset, seed, counter, _p(z), _p(a))
    return z, a


def rollout_random_f32(model, n, T, threads=0, env_offset=0, r=0.3, K=1.0, sigma=0.1, C=0.5, x0=0.75, Tmax=100,
                       n_actions=100, seed=1234, step_counter=0, obs=None, t=None):
    obs = np.full(n, x0 / K - 1.0, np.float32) if obs is None else obs
    t = np.zeros(n, np.int32) if t is None else t
    total = lib().oracle_rollout_random_f32(model, n, env_offset, T, _p(obs), _p(t), r, K, sigma, C, x0, Tmax,
                                            n_actions, seed, step_counter, threads)
    return total, obs, t
