"""The C-ABI library loads on a CPU-only host and exports every symbol include/fishing_hip.h
declares; argument errors are reported without touching a GPU.  No compute calls here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from gym_fishing_amd import _capi, build

HEADER = os.path.join(ROOT, "include", "fishing_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fishing_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(build.LIB_PATH):
        build.build()
    return _capi.lib()


def test_header_and_binding_list_the_same_functions():
    assert declared_functions() == sorted(_capi.SIGNATURES)


def test_library_exports_every_declared_symbol(lib):
    raw = ctypes.CDLL(build.LIB_PATH)
    for name in declared_functions():
        assert hasattr(raw, name), name


def test_abi_version_and_struct_layout(lib):
    assert lib.fishing_abi_version() == _capi.ABI_VERSION
    # FishingParams: 4 x i32, 8 x f64, 2 x i32 (88) + 6 x f64 + 6 x i32 (160) + 5 x 9 x f64 (520) + 2 x u64 -> 536 bytes
    assert ctypes.sizeof(_capi.FishingParams) == 536
    assert ctypes.sizeof(_capi.FishingBuffers) == 16 * ctypes.sizeof(ctypes.c_void_p)       # (v4_stamp: ABI 6)
    hdr = open(HEADER).read()
    body = hdr[hdr.index("typedef struct FishingBuffers {"):hdr.index("} FishingBuffers;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(\w+)\s*;", body)
    assert tuple(fields) == _capi.BUFFER_FIELDS
    assert lib.fishing_partials_len() == 65536 * 4      # a workgroup per 1024-env tile up to N = 2^26 (ABI 5)
    # slots a batch can touch: whole 4096-slot passes of the reduction, one per 2^22 envs
    assert [lib.fishing_partials_slots(n) for n in (0, 1, 1 << 22, (1 << 22) + 1, 1 << 24, (1 << 26) - 5, 1 << 26, 1 << 30)] == \
        [4096, 4096, 4096, 8192, 16384, 65536, 65536, 65536]


def test_error_strings(lib):
    assert lib.fishing_error_string(0) == b"ok"
    for code in (-1, -2, -3, -4, -5, -6, -7, -8):
        assert lib.fishing_error_string(code) not in (b"ok", b"unknown error")


def test_argument_errors_need_no_gpu(lib):
    """Validation happens before any launch, so it is checkable on a CPU-only host."""
    p = _capi.FishingParams()
    p.model, p.n_actions, p.Tmax = _capi.MODEL_V1, 100, 100
    b = _capi.FishingBuffers()
    assert lib.fishing_step_f32(None, 4, 0, b, 0, 0, None) == -1          # NULL params
    assert lib.fishing_step_f32(p, 4, 0, b, 0, 0, None) == -1             # NULL obs / t
    b = _capi.make_buffers(obs=4096, t=8192, action=12288)
    assert lib.fishing_step_f32(p, -3, 0, b, 0, 0, None) == -4            # n < 0
    assert lib.fishing_step_f32(p, 4, 6, b, 0, 0, None) == -4             # env_offset % 4 != 0
    assert lib.fishing_step_f32(p, 0, 0, b, 0, 0, None) == 0              # n == 0: nothing to do
    b = _capi.make_buffers(obs=4100, t=8192, action=12288)
    assert lib.fishing_step_f64(p, 4, 0, b, 0, 0, None) == -3             # misaligned obs
    p.model = 3                                                           # there is no fishing-v3
    b = _capi.make_buffers(obs=4096, t=8192, action=12288)
    assert lib.fishing_step_f32(p, 4, 0, b, 0, 0, None) == -2             # unknown model
    p.model = _capi.MODEL_V4
    assert lib.fishing_reset_f32(p, 4, 0, b, None, 0, 0, None) == -1      # v4 needs r, K arrays
    bv4 = _capi.make_buffers(obs=4096, t=8192, action=12288, r=16384, K=20480)
    p.K_mean, p.r_mean, p.sigma_p = float("nan"), 0.3, 0.1
    assert lib.fishing_reset_f32(p, 4, 0, bv4, None, 0, 0, None) == -8    # fishing-v4's means must be finite
    p.K_mean, p.sigma_p = 1.0, float("inf")
    assert lib.fishing_step_f32(p, 4, 0, bv4, 0, 0, None) == -8
    p.sigma_p = 0.1
    p.flags = _capi.FLAG_V4_DERIVED | _capi.FLAG_T_U8
    assert lib.fishing_reset_f32(p, 4, 0, b, None, 0, 0, None) == -7      # derived parameters need the int32 year counter
    p.flags = _capi.FLAG_V4_DERIVED
    assert lib.fishing_reset_f32(p, 4, 0, b, 24576, 0, 0, None) == -7     # ... and, for a masked reset, the origin stamps (v4_stamp)
    bs = _capi.make_buffers(obs=4096, t=8192, action=12288, v4_stamp=28672)
    assert lib.fishing_reset_f32(p, 4, 0, bs, 24576, 0, 0x7FFFFFFF, None) == -4      # a stamp holds 31 bits of reset counter + 1
    p.flags = 0
    bs = _capi.make_buffers(obs=4096, t=8192, action=12288, r=16384, K=20480, v4_stamp=28672)
    assert lib.fishing_step_f32(p, 4, 0, bs, 0, 0, None) == -7            # stamps belong to the derived mode
    assert lib.fishing_v4_params_f32(p, 4, 0, None, None, None, None, 0, 0, None) == -1
    p.model = _capi.MODEL_V1
    assert lib.fishing_v4_params_f32(p, 4, 0, 4096, None, None, None, 0, 0, None) == -2     # fishing-v4 only
    assert lib.fishing_step_fused_f32(p, 4, 0, b, 4, 0, 3, None, None, 0, 0, 0, None) == -4   # ring_len <= 0
    assert lib.fishing_step_fused_f32(p, 4, 0, b, 4, 2, 3, 4096, None, 2, 0, 0, None) == -3   # out_stride < n
    bz = _capi.make_buffers(obs=4096, t=8192, action=12288, z_ext=16384)
    assert lib.fishing_step_fused_f32(p, 4, 0, bz, 4, 2, 3, None, None, 0, 0, 0, None) == -7  # external noise: step() only
    name = ctypes.create_string_buffer(128)
    p.K = 1.0
    bo = _capi.make_buffers(obs=4096, t=8192, action=12288, reward=16384, done=20480)
    assert lib.fishing_step_kernel_name_f32(p, 1 << 22, bo, name, 128) == 0
    assert name.value == b"fishing::step_kernel_lean<float, 1, 11391, 4>"     # sigma = 0 (no generator): the catch-all
    p.sigma = 0.1
    bo.ep_return = 24576
    assert lib.fishing_step_kernel_name_f32(p, 1 << 22, bo, name, 128) == 0
    assert name.value == b"fishing::step_kernel_lean<float, 1, 11391, 4>"     # a record without auto-reset: the catch-all's latch
    p.flags = _capi.FLAG_AUTO_RESET
    assert lib.fishing_step_kernel_name_f32(p, 1 << 22, bo, name, 128) == 0
    assert name.value == b"fishing::step_kernel_lean<float, 1, 12294, 4>"    # (last argument: envs per thread) Philox (2) | RET (4) | KP2 (4096) | ONE (8192: a tile per workgroup): K = 1
    p.K = 1.5
    assert lib.fishing_step_kernel_name_f32(p, 1 << 22, bo, name, 128) == 0
    assert name.value == b"fishing::step_kernel_lean<float, 1, 8198, 4>"     # K not a power of two: the true division
    p.K = 1.0
    assert lib.fishing_step_kernel_name_f32(p, 1000, bo, name, 128) == 0
    assert name.value == b"fishing::step_kernel<float, 1>"                   # below one tile: the general kernel
    p.sigma = 0.0
    assert lib.fishing_rollout_f32(p, 4, 0, b, 17, 0.0, 3, None, 0, 0, None) == -5
    assert lib.fishing_rollout_f32(p, 4, 0, b, 0, 0.0, -1, None, 0, 0, None) == -4
    assert lib.fishing_reduce_returns(None, None, None) == -1
    assert lib.fishing_reduce_returns_slots(None, 4096, None, None) == -1
    one = ctypes.c_double(0.0)
    for bad in (0, 4095, 6144, 65536 + 4096):       # whole passes only, within the buffer
        assert lib.fishing_reduce_returns_slots(ctypes.byref(one), bad, ctypes.byref(one), None) == -4
    assert lib.fishing_noise_f32(-1, 0, 0, 0, 0, None, None, None, None) == -4
    p.launch_threads = 100
    assert lib.fishing_step_f32(p, 4, 0, b, 0, 0, None) == -4             # threads not a multiple of 64


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setenv("FISHING_HIP_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_capi, "_lib", None)
    with pytest.raises(_capi.FishingLibraryError, match="no CPU fallback"):
        _capi.lib()


def test_header_is_valid_c_and_cxx(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 and as C++ with no HIP headers."""
    import shutil
    import subprocess
    inc = os.path.join(ROOT, "include")
    src_c = tmp_path / "t.c"
    src_c.write_text('#include "fishing_hip.h"\nint main(void){FishingParams p; FishingBuffers b; (void)p; (void)b; '
                     'return sizeof(FishingParams) == 536 ? 0 : 1;}\n')
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, str(src_c), "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0
    if shutil.which("g++"):
        src_cc = tmp_path / "t.cc"
        src_cc.write_text('#include "fishing_hip.h"\nint main(){return fishing_abi_version == nullptr;}\n')
        subprocess.run(["g++", "-std=c++11", "-Wall", "-fsyntax-only", "-I", inc, str(src_cc)], check=True)
