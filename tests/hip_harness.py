"""Thin test harness: call the C ABI (include/fishing_hip.h) with NumPy inputs.

Device memory comes from torch (plumbing); every compute call goes through the ctypes
binding gym_fishing_amd._capi, i.e. through libfishing_hip.so.
"""
import numpy as np
import torch

from gym_fishing_amd import _capi

TORCH_OF = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}


def dev(a, dtype=None):
    t = torch.as_tensor(np.array(a))   # copy: inputs may be read-only broadcasts
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def params(model, r=0.3, K=1.0, sigma=0.0, C=0.5, x0=0.75, Tmax=100, n_actions=100, K_mean=1.0, r_mean=0.3,
           sigma_p=0.1, auto_reset=False, launch_blocks=0, launch_threads=0, M=0.0, theta=0.0, q=0.0, b=0.0, a=0.0,
           alpha=0.0, models=None, zoo_table=None, general=False, t_u8=False, derived=False, origin=(0, 0), padded=False):
    p = _capi.FishingParams()
    p.M, p.theta, p.q, p.b, p.a, p.alpha = M, theta, q, b, a, alpha
    if models is not None:                      # fishing-v11: list of kind indices + per-kind dicts
        p.n_models = len(models)
        for i, k in enumerate(models):
            p.kinds[i] = k
        for k, d in enumerate(zoo_table):
            for name in ("r", "K", "sigma", "C", "M", "theta", "q", "b", "a"):
                setattr(p.zoo[k], name, float(d.get(name, 0.0)))
    p.model, p.n_actions, p.Tmax = model, n_actions, Tmax
    p.flags = ((_capi.FLAG_AUTO_RESET if auto_reset else 0) | (_capi.FLAG_GENERAL_KERNEL if general else 0)
               | (_capi.FLAG_T_U8 if t_u8 else 0) | (_capi.FLAG_V4_DERIVED if derived else 0)
               | (_capi.FLAG_PADDED_TILES if padded else 0))
    p.v4_origin_step, p.v4_origin_counter = origin
    p.r, p.K, p.sigma, p.C, p.x0 = r, K, sigma, C, x0
    p.r_mean, p.K_mean, p.sigma_p = r_mean, K_mean, sigma_p
    p.launch_blocks, p.launch_threads = launch_blocks, launch_threads
    return p


class State:
    """Device buffers of one shard, created from host arrays."""

    def __init__(self, n, dtype, model, obs, t=None, r=None, K=None, sigma=None, ep_return=False,
                 terminal=False, done_bits=False, model_idx=None, t_u8=False, stamp=None):
        self.n, self.np_dtype, self.model = n, np.dtype(dtype), model
        td = TORCH_OF[self.np_dtype]
        self.obs = dev(np.broadcast_to(np.asarray(obs, dtype=dtype), (n,)))
        self.t = dev(np.broadcast_to(np.asarray(0 if t is None else t, dtype=np.uint8 if t_u8 else np.int32), (n,)))
        self.reward = torch.zeros(n, dtype=td, device="cuda")
        self.done = torch.zeros(n, dtype=torch.uint8, device="cuda")
        self.r = dev(np.broadcast_to(np.asarray(r, dtype=dtype), (n,))) if r is not None else None
        self.K = dev(np.broadcast_to(np.asarray(K, dtype=dtype), (n,))) if K is not None else None
        self.sigma = dev(np.broadcast_to(np.asarray(sigma, dtype=dtype), (n,))) if sigma is not None else None
        self.terminal = torch.zeros(n, dtype=td, device="cuda") if terminal else None
        self.ep_return = torch.zeros(n, dtype=td, device="cuda") if ep_return else None
        self.partials = (torch.zeros(int(_capi.lib().fishing_partials_len()), dtype=torch.float64, device="cuda")
                         if ep_return else None)
        self.done_bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda") if done_bits else None
        self.model_idx = (dev(np.broadcast_to(np.asarray(model_idx, dtype=np.int32), (n,)))
                          if model_idx is not None else None)
        self.stamp = dev(np.broadcast_to(np.asarray(stamp, dtype=np.int32), (n,))) if stamp is not None else None

    def buffers(self, action=None, z_ext=None):
        p = lambda x: x.data_ptr() if x is not None else None  # noqa: E731
        return _capi.make_buffers(obs=p(self.obs), action=p(action), reward=p(self.reward), done=p(self.done),
                                  done_bits=p(self.done_bits), t=p(self.t), r=p(self.r), K=p(self.K),
                                  sigma=p(self.sigma), z_ext=p(z_ext), terminal_obs=p(self.terminal),
                                  ep_return=p(self.ep_return), return_partials=p(self.partials),
                                  model_idx=p(self.model_idx), v4_stamp=p(self.stamp))

    @property
    def suffix(self):
        return "f32" if self.np_dtype == np.float32 else "f64"

    def action_tensor(self, action):
        return dev(action, torch.int32 if self.model == _capi.MODEL_V0 else torch.float32)

    def step(self, p, action, z=None, seed=0, step_counter=0, env_offset=0, n=None, expect=0):
        a = self.action_tensor(action)
        zt = dev(np.asarray(z, dtype=self.np_dtype)) if z is not None else None
        fn = getattr(_capi.lib(), "fishing_step_" + self.suffix)
        rc = fn(p, self.n if n is None else n, env_offset, self.buffers(a, zt), seed, step_counter, None)
        assert rc == expect, "fishing_step rc=%d (%s)" % (rc, _capi.lib().fishing_error_string(rc))
        torch.cuda.synchronize()
        return self.host()

    def reset(self, p, mask=None, seed=0, counter=0, env_offset=0, expect=0):
        m = dev(np.asarray(mask, dtype=np.uint8)) if mask is not None else None
        fn = getattr(_capi.lib(), "fishing_reset_" + self.suffix)
        rc = fn(p, self.n, env_offset, self.buffers(), m.data_ptr() if m is not None else None, seed, counter, None)
        assert rc == expect, rc
        torch.cuda.synchronize()

    def rollout(self, p, policy, param, T, seed=0, step_counter=0, env_offset=0, record=False):
        traj = (torch.zeros((T, 4, self.n), dtype=TORCH_OF[self.np_dtype], device="cuda") if record else None)
        fn = getattr(_capi.lib(), "fishing_rollout_" + self.suffix)
        rc = fn(p, self.n, env_offset, self.buffers(), policy, float(param), T,
                traj.data_ptr() if record else None, seed, step_counter, None)
        assert rc == 0, "fishing_rollout rc=%d" % rc
        torch.cuda.synchronize()
        return traj.cpu().numpy() if record else None

    def ring_tensor(self, actions):
        """[R, n] host actions -> device ring whose rows start 16-byte aligned (row stride padded to 4 elements);
        returns the [R, n] view."""
        a = np.asarray(actions)
        R = a.shape[0]
        stride = (self.n + 3) // 4 * 4
        buf = torch.zeros((R, stride), dtype=torch.int32 if self.model == _capi.MODEL_V0 else torch.float32, device="cuda")
        view = buf[:, :self.n]
        view.copy_(torch.as_tensor(a).to(buf.dtype))
        return view

    def step_fused(self, p, actions, n_steps, seed=0, step_counter=0, env_offset=0, per_step=True, expect=0):
        """fishing_step_fused_*: `actions` a [R, n] host array (the ring); returns (reward_steps, done_steps) host
        arrays [n_steps, n] when per_step."""
        a = self.ring_tensor(actions)
        R = a.shape[0]
        td = TORCH_OF[self.np_dtype]
        stride = (self.n + 15) // 16 * 16
        rs = torch.zeros((n_steps, stride), dtype=td, device="cuda") if per_step else None
        ds = torch.zeros((n_steps, stride), dtype=torch.uint8, device="cuda") if per_step else None
        fn = getattr(_capi.lib(), "fishing_step_fused_" + self.suffix)
        rc = fn(p, self.n, env_offset, self.buffers(a), a.stride(0), R, n_steps,
                rs.data_ptr() if per_step else None, ds.data_ptr() if per_step else None, stride if per_step else 0,
                seed, step_counter, None)
        assert rc == expect, "fishing_step_fused rc=%d (%s)" % (rc, _capi.lib().fishing_error_string(rc))
        torch.cuda.synchronize()
        if per_step:
            return rs[:, :self.n].cpu().numpy(), ds[:, :self.n].cpu().numpy()
        return None, None

    def v4_params(self, p, seed=0, step_counter=0, env_offset=0):
        """(K, r) in force under FISHING_FLAG_V4_DERIVED, materialised by fishing_v4_params_*."""
        td = TORCH_OF[self.np_dtype]
        K = torch.zeros(self.n, dtype=td, device="cuda")
        r = torch.zeros(self.n, dtype=td, device="cuda")
        fn = getattr(_capi.lib(), "fishing_v4_params_" + self.suffix)
        rc = fn(p, self.n, env_offset, self.t.data_ptr(), self.stamp.data_ptr() if self.stamp is not None else None,
                K.data_ptr(), r.data_ptr(), seed, step_counter, None)
        assert rc == 0, rc
        torch.cuda.synchronize()
        return K.cpu().numpy(), r.cpu().numpy()

    def host(self):
        return (self.obs.cpu().numpy(), self.reward.cpu().numpy(), self.done.cpu().numpy(), self.t.cpu().numpy())

    def record(self):
        out = torch.zeros(4, dtype=torch.float64, device="cuda")
        rc = _capi.lib().fishing_reduce_returns(self.partials.data_ptr(), out.data_ptr(), None)
        assert rc == 0
        torch.cuda.synchronize()
        return out.cpu().numpy()


def kernel_name(p, n, buffers, dtype=np.float32, raw=False):
    """Name of the kernel step() would launch for the whole tiles of this request (fishing_step_kernel_name_*), as
    rocprofv3 prints it.  The lean kernel's last template argument is its envs per thread: the usual 4 is dropped here
    unless `raw` (tests name the feature mask; the float64 layout's 2-per-thread form keeps its ", 2>")."""
    import ctypes
    out = ctypes.create_string_buffer(160)
    fn = getattr(_capi.lib(), "fishing_step_kernel_name_" + ("f32" if np.dtype(dtype) == np.float32 else "f64"))
    rc = fn(p, n, buffers, out, 160)
    assert rc == 0, rc
    name = out.value.decode()
    if not raw and "step_kernel_lean<" in name and name.endswith(", 4>"):
        name = name[:-4] + ">"
    return name


def device_noise(n, seed, counter, stream_tag=0, env_offset=0):
    """(words[n,4], z0[n], z1[n]) of Philox index env_offset + i from the device generator; on the reset
    streams z0 / z1 are the (zK, zr) normals of ENV env_offset + i (its own Philox2x32 block), words stay the
    Philox4x32 block of that index."""
    words = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    z0 = torch.zeros(n, dtype=torch.float32, device="cuda")
    z1 = torch.zeros(n, dtype=torch.float32, device="cuda")
    rc = _capi.lib().fishing_noise_f32(n, env_offset, seed, counter, stream_tag, words.data_ptr(), z0.data_ptr(),
                                      z1.data_ptr(), None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    if stream_tag != _capi.STREAM_NOISE:
        # fishing-v4 (K, r) normals: one block per env PAIR on the reset streams (fishing_hip.h)
        rc = _capi.lib().fishing_reset_normals_f32(n, env_offset, seed, counter, stream_tag, z0.data_ptr(),
                                                  z1.data_ptr(), None)
        assert rc == 0, rc
        torch.cuda.synchronize()
    return words.cpu().numpy().view(np.uint32), z0.cpu().numpy(), z1.cpu().numpy()


def device_step_noise(n, seed, step_counter, env_offset=0):
    """The float32 z the step kernel uses for envs env_offset .. env_offset+n-1 (quad scheme)."""
    z = torch.zeros(max(n, 1), dtype=torch.float32, device="cuda")
    rc = _capi.lib().fishing_step_normals_f32(n, env_offset, seed, step_counter, z.data_ptr(), None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return z.cpu().numpy()[:n]


def ulp_diff(a, b):
    """Distance in units in the last place between two float arrays of the same dtype."""
    a = np.asarray(a)
    b = np.asarray(b)
    it = np.int64 if a.dtype == np.float64 else np.int32
    ai = a.view(it).astype(np.int64)
    bi = b.view(it).astype(np.int64)
    sign = np.int64(np.iinfo(it).min)
    ai = np.where(ai < 0, sign - ai, ai)
    bi = np.where(bi < 0, sign - bi, bi)
    d = np.abs(ai - bi)
    return np.where(np.isnan(a) & np.isnan(b), 0, d)
