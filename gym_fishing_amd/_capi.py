"""ctypes binding of libfishing_hip.so (include/fishing_hip.h).

There is exactly one compute backend: the HIP library.  If it cannot be loaded the
product raises FishingLibraryError -- no CPU fallback exists anywhere in this package
(the CPU restatement lives in oracle/ and is test infrastructure only).
"""
import ctypes
import os

c_i32, c_i64, c_u32, c_u64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint64
c_dbl, c_vp = ctypes.c_double, ctypes.c_void_p

ABI_VERSION = 9

MODEL_V0, MODEL_V1, MODEL_V2, MODEL_V4 = 0, 1, 2, 4
MODEL_V5, MODEL_V6, MODEL_V7, MODEL_V8, MODEL_V9, MODEL_V10, MODEL_V11 = 5, 6, 7, 8, 9, 10, 11
KIND_ALLEN, KIND_BEVERTON_HOLT, KIND_MYERS, KIND_MAY, KIND_RICKER = 0, 1, 2, 3, 4
KIND_OF_NAME = {"allen": KIND_ALLEN, "beverton_holt": KIND_BEVERTON_HOLT, "myers": KIND_MYERS, "may": KIND_MAY,
                "ricker": KIND_RICKER}
N_KINDS = 5
FLAG_AUTO_RESET = 1
FLAG_T_U8 = 4
FLAG_V4_DERIVED = 8                  # fishing-v4: (K, r) re-derived in-kernel, no r / K arrays
FLAG_PADDED_TILES = 16               # state buffers hold whole 1024-env tiles: a ragged batch steps in one launch
FLAG_RESET_COUNTER_ON_DEVICE = 32    # counter is u64[4]; fishing_reset_* reads and bumps the reset counter (word 3) itself
FLAG_GENERAL_KERNEL = 0x80000000     # FISHING_FLAG_DIAG_GENERAL_KERNEL (tests, A/B timing)
POLICY_RANDOM, POLICY_CONSTANT, POLICY_ESCAPEMENT, POLICY_MSY = 0, 1, 2, 3
STREAM_NOISE, STREAM_AUTORESET, STREAM_RESET, STREAM_POLICY = 0, 1, 2, 3


class FishingLibraryError(RuntimeError):
    pass


class FishingGrowthParams(ctypes.Structure):
    _fields_ = [(k, c_dbl) for k in ("r", "K", "sigma", "C", "M", "theta", "q", "b", "a")]


class FishingParams(ctypes.Structure):
    _fields_ = [("model", c_i32), ("n_actions", c_i32), ("Tmax", c_i32), ("flags", c_u32),
                ("r", c_dbl), ("K", c_dbl), ("sigma", c_dbl), ("C", c_dbl), ("x0", c_dbl),
                ("r_mean", c_dbl), ("K_mean", c_dbl), ("sigma_p", c_dbl),
                ("launch_blocks", c_i32), ("launch_threads", c_i32),
                ("M", c_dbl), ("theta", c_dbl), ("q", c_dbl), ("b", c_dbl), ("a", c_dbl), ("alpha", c_dbl),
                ("n_models", c_i32), ("kinds", c_i32 * 5), ("zoo", FishingGrowthParams * 5),
                ("v4_origin_step", c_u64), ("v4_origin_counter", c_u64)]


BUFFER_FIELDS = ("obs", "action", "reward", "done", "done_bits", "t", "r", "K", "sigma", "z_ext",
                 "terminal_obs", "ep_return", "return_partials", "model_idx", "counter", "v4_stamp")


class FishingBuffers(ctypes.Structure):
    _fields_ = [(name, c_vp) for name in BUFFER_FIELDS]


# symbol -> (restype, argtypes); must list every function include/fishing_hip.h declares
_PP = ctypes.POINTER(FishingParams)
_BP = ctypes.POINTER(FishingBuffers)
SIGNATURES = {
    "fishing_abi_version": (c_i32, []),
    "fishing_error_string": (ctypes.c_char_p, [c_i32]),
    "fishing_partials_len": (c_i64, []),
    "fishing_partials_slots": (c_i64, [c_i64]),
    "fishing_step_f32": (c_i32, [_PP, c_i64, c_i64, _BP, c_u64, c_u64, c_vp]),
    "fishing_step_f64": (c_i32, [_PP, c_i64, c_i64, _BP, c_u64, c_u64, c_vp]),
    "fishing_step_many_f32": (c_i32, [_PP, c_i64, c_i64, _BP, c_i64, c_i32, c_i32, c_u64, c_u64, c_vp]),
    "fishing_step_many_f64": (c_i32, [_PP, c_i64, c_i64, _BP, c_i64, c_i32, c_i32, c_u64, c_u64, c_vp]),
    "fishing_reset_f32": (c_i32, [_PP, c_i64, c_i64, _BP, c_vp, c_u64, c_u64, c_vp]),
    "fishing_reset_f64": (c_i32, [_PP, c_i64, c_i64, _BP, c_vp, c_u64, c_u64, c_vp]),
    "fishing_step_fused_f32": (c_i32, [_PP, c_i64, c_i64, _BP, c_i64, c_i32, c_i32, c_vp, c_vp, c_i64, c_u64, c_u64, c_vp]),
    "fishing_step_fused_f64": (c_i32, [_PP, c_i64, c_i64, _BP, c_i64, c_i32, c_i32, c_vp, c_vp, c_i64, c_u64, c_u64, c_vp]),
    "fishing_v4_params_f32": (c_i32, [_PP, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_u64, c_u64, c_vp]),
    "fishing_v4_params_f64": (c_i32, [_PP, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_u64, c_u64, c_vp]),
    "fishing_step_kernel_name_f32": (c_i32, [_PP, c_i64, _BP, ctypes.c_char_p, c_i64]),
    "fishing_step_kernel_name_f64": (c_i32, [_PP, c_i64, _BP, ctypes.c_char_p, c_i64]),
    "fishing_rollout_f32": (c_i32, [_PP, c_i64, c_i64, _BP, c_i32, c_dbl, c_i32, c_vp, c_u64, c_u64, c_vp]),
    "fishing_rollout_f64": (c_i32, [_PP, c_i64, c_i64, _BP, c_i32, c_dbl, c_i32, c_vp, c_u64, c_u64, c_vp]),
    "fishing_rollout_params_f32": (c_i32, [_PP, c_i64, c_i64, _BP, c_i32, c_vp, c_i32, c_vp, c_u64, c_u64, c_vp]),
    "fishing_rollout_params_f64": (c_i32, [_PP, c_i64, c_i64, _BP, c_i32, c_vp, c_i32, c_vp, c_u64, c_u64, c_vp]),
    "fishing_reduce_returns": (c_i32, [c_vp, c_vp, c_vp]),
    "fishing_reduce_returns_slots": (c_i32, [c_vp, c_i64, c_vp, c_vp]),
    "fishing_counter_add": (c_i32, [c_vp, c_u64, c_vp]),
    "fishing_population_draw_f32": (c_i32, [_PP, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "fishing_population_draw_f64": (c_i32, [_PP, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "fishing_bmsy_sweep_f32": (c_i32, [_PP, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "fishing_bmsy_sweep_f64": (c_i32, [_PP, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "fishing_stream_synchronize": (c_i32, [c_vp]),
    "fishing_noise_f32": (c_i32, [c_i64, c_i64, c_u64, c_u64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "fishing_step_normals_f32": (c_i32, [c_i64, c_i64, c_u64, c_u64, c_vp, c_vp]),
    "fishing_reset_normals_f32": (c_i32, [c_i64, c_i64, c_u64, c_u64, c_i32, c_vp, c_vp, c_vp]),
    "fishing_math_f64": (c_i32, [c_i64, c_i32, c_vp, c_vp, c_vp]),
    "fishing_step_floor_f32": (c_i32, [c_i32, c_i64, _BP, c_i32, c_vp]),
}

_lib = None


def library_path():
    from . import build as _build       # lazy: keeps `python -m gym_fishing_amd.build` warning-free
    return os.environ.get("FISHING_HIP_LIB", _build.LIB_PATH)


def lib():
    """Load (once) and return the ctypes handle.  Raises FishingLibraryError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise FishingLibraryError(
            "libfishing_hip.so not found at %s -- build it with `python -m gym_fishing_amd.build` "
            "(needs hipcc; cross-compiles for gfx950 without a GPU). There is no CPU fallback." % path)
    try:
        handle = ctypes.CDLL(path)
    except OSError as e:  # pragma: no cover - depends on the host's ROCm install
        raise FishingLibraryError("cannot load %s: %s" % (path, e)) from e
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise FishingLibraryError("%s does not export %s (stale build?)" % (path, name)) from e
        fn.restype = restype
        fn.argtypes = argtypes
    got = handle.fishing_abi_version()
    if got != ABI_VERSION:
        raise FishingLibraryError("ABI version mismatch: library %d, binding %d" % (got, ABI_VERSION))
    _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().fishing_error_string(int(rc))
        raise FishingLibraryError("%s failed: %s (code %d)" % (what, msg.decode() if msg else "?", rc))


def make_buffers(**ptrs):
    """FishingBuffers from {field: int device pointer or None}."""
    b = FishingBuffers()
    for k, v in ptrs.items():
        if k not in BUFFER_FIELDS:
            raise KeyError(k)
        setattr(b, k, v if v else None)
    return b
