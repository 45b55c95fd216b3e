"""Guard bands around every buffer handed to the C ABI.

Each case carves its streams out of ONE device arena, every stream flush against a 4 KiB band of a known byte on
either side (the band behind a stream begins at the byte after its last element), drives the entry points of
include/fishing_hip.h over it -- reset (all / masked), step, step_many, step_fused, rollout (with and without the
trajectory rows, with one policy parameter per env), the fishing-v4 parameter materialisation, population_draw, the BMSY sweep, the generator hooks, the
return reduction -- and then reads the arena back: no byte outside the streams may have changed.  Run twice with two
different band bytes, the streams themselves must also come out identical: a read behind a stream that reached a result
would show there.  Sizes: one env, sub-quad, sub-tile, whole tiles, tiles + ragged tails, and the padded-tile contract
(FISHING_FLAG_PADDED_TILES: state streams hold whole 1024-env tiles, the action stream exactly n elements).

The kernels' out-of-bounds behaviour is otherwise invisible: torch's allocator rounds every tensor up to 512 bytes and
packs small ones into shared segments, so a stray 16-byte store lands in slack or in a neighbour and nothing faults.
"""
import numpy as np
import pytest

from oracle import fishing_oracle as fo     # (model ids and fishing-v11's parameter table only)

pytestmark = pytest.mark.gpu
GUARD = 4096
TILE = 1024
ERR_UNSUPPORTED = -7            # include/fishing_hip.h: FISHING_ERR_UNSUPPORTED
MATH_FUNCTIONS = (0, 1)  # FISHING_MATH_LOG_F64, FISHING_MATH_EXP_F64


@pytest.fixture(scope="module")
def hh():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device; none visible")
    import hip_harness
    return hip_harness


class Arena:
    """One uint8 device tensor filled with `fill`; alloc() hands out 256-byte-aligned views with GUARD bytes between."""

    def __init__(self, fill, nbytes=4 << 20):
        import torch
        self.torch = torch
        self.fill = fill
        self.buf = torch.full((nbytes,), fill, dtype=torch.uint8, device="cuda")
        self.cur = GUARD
        self.live = []          # (name, offset, nbytes)
        self.item = {}          # name -> element size

    def alloc(self, name, count, dtype, init=None):
        torch = self.torch
        item = torch.empty((), dtype=dtype).element_size()
        nbytes = int(count) * item
        off = (self.cur + 255) // 256 * 256
        assert off + nbytes + GUARD <= self.buf.numel(), "arena too small for %s" % name
        self.cur = off + nbytes + GUARD
        self.live.append((name, off, nbytes))
        self.item[name] = item
        view = self.buf[off:off + nbytes].view(dtype)
        if init is None:
            view.zero_()
        else:
            view.copy_(torch.as_tensor(np.ascontiguousarray(init)).to(dtype))
        return view

    def check(self, what):
        """Every byte outside the streams still holds the band byte; returns the streams' bytes (host) by name."""
        self.torch.cuda.synchronize()
        host = self.buf.cpu().numpy()
        inside = np.zeros(host.size, dtype=bool)
        out = {}
        for name, off, nbytes in self.live:
            inside[off:off + nbytes] = True
            out[name] = host[off:off + nbytes].copy()
        bad = np.flatnonzero(~inside & (host != self.fill))
        if bad.size:
            first = int(bad[0])
            near = min(self.live, key=lambda s: min(abs(first - s[1]), abs(first - (s[1] + s[2]))))
            where = "before" if first < near[1] else "%d bytes behind the end of" % (first - (near[1] + near[2]))
            raise AssertionError("%s: %d guard bytes overwritten, first %s stream '%s' (%d bytes long)"
                                 % (what, bad.size, where, near[0], near[2]))
        return out


# (name, model, params, what the state needs)
CASES = [
    ("v0", fo.MODEL_V0, dict(sigma=0.1), {}),
    ("v1", fo.MODEL_V1, dict(sigma=0.1), {}),
    ("v1_K3_sigma_array", fo.MODEL_V1, dict(sigma=0.1, K=3.0, x0=2.25), dict(sigarr=True)),
    ("v1_byte_years", fo.MODEL_V1, dict(sigma=0.1, t_u8=True), dict(t8=True)),
    ("v2", fo.MODEL_V2, dict(sigma=0.1), {}),
    ("v4_stored", fo.MODEL_V4, dict(sigma=0.1, sigma_p=0.2), dict(rk=True)),
    ("v4_derived", fo.MODEL_V4, dict(sigma=0.1, sigma_p=0.2, derived=True), {}),
    ("v4_stamped", fo.MODEL_V4, dict(sigma=0.1, sigma_p=0.2, derived=True), dict(stamp=True)),
    ("v5", fo.MODEL_V5, dict(sigma=0.1, r=0.3, C=0.5), {}),
    ("v7", fo.MODEL_V7, dict(sigma=0.1, r=0.7, K=1.5, M=1.5, q=3.0, b=0.15, a=0.2), {}),
    ("v8", fo.MODEL_V8, dict(sigma=0.1, r=1.0, M=1.0, theta=3.0, x0=1.5), {}),
    ("v9", fo.MODEL_V9, dict(sigma=0.1), {}),
    ("v10", fo.MODEL_V10, dict(sigma=0.1, r=0.8, alpha=-0.01), dict(r_only=True)),
    ("v11", fo.MODEL_V11, dict(), dict(mixed=True)),
]
SIZES = [1, 3, 4, 7, 64, 1000, 1023, 1024, 1027, 2 * TILE + 4, 4 * TILE + 3]
PADDED_SIZES = [4, 1000, TILE + 4, 4 * TILE + 612]
PADDED_STATE_STREAMS = ("obs", "t", "reward", "done", "r", "K", "sigma", "model_idx", "v4_stamp", "terminal_obs", "ep_return")


class Guarded:
    """The streams of one shard inside an Arena (the layout of hip_harness.State, guard bands between)."""

    def __init__(self, hh, arena, n, cap, dtype, model, need, optional, n_rows):
        import torch
        from gym_fishing_amd import _capi
        self.hh, self.ar, self.n, self.cap, self.model = hh, arena, n, cap, model
        self.np_dtype = np.dtype(dtype)
        td = hh.TORCH_OF[self.np_dtype]
        self.td = td
        A = arena.alloc
        self.obs = A("obs", cap, td)
        self.t = A("t", cap, torch.uint8 if need.get("t8") else torch.int32)
        self.reward = A("reward", cap, td)
        self.done = A("done", cap, torch.uint8)
        self.r = A("r", cap, td, np.full(cap, 0.8 if need.get("r_only") else 0.3)) if need.get("rk") or need.get("r_only") else None
        self.K = A("K", cap, td, np.full(cap, 1.0)) if need.get("rk") else None
        self.sigma = A("sigma", cap, td, np.linspace(0.02, 0.2, cap)) if need.get("sigarr") else None
        self.model_idx = A("model_idx", cap, torch.int32) if need.get("mixed") else None
        self.stamp = A("v4_stamp", cap, torch.int32) if need.get("stamp") else None
        self.terminal = A("terminal_obs", cap, td) if optional else None
        self.done_bits = A("done_bits", (cap + 63) // 64, torch.int64) if optional else None
        self.ep_return = A("ep_return", cap, td) if optional else None
        self.slots = int(_capi.lib().fishing_partials_slots(n))
        self.partials = A("return_partials", 4 * self.slots, torch.float64) if optional else None
        # caller-owned inputs: exactly n elements, whatever the state streams hold
        rng = np.random.default_rng(11)
        self.stride = (n + 3) // 4 * 4
        if model == fo.MODEL_V0:
            ring = rng.integers(0, 100, (n_rows, n)).astype(np.int32)
        else:
            ring = rng.uniform(-1.15, 0.3, (n_rows, n)).astype(np.float32)
        # the ring's rows start 16-byte aligned; its LAST row ends with its n-th element
        self.ring = A("action_ring", (n_rows - 1) * self.stride + n, torch.int32 if model == fo.MODEL_V0 else torch.float32)
        for k in range(n_rows):
            self.ring[k * self.stride:k * self.stride + n].copy_(torch.as_tensor(ring[k]))
        self.z = A("z_ext", n, td, rng.standard_normal(n))
        self.mask = A("reset_mask", n, torch.uint8, (np.arange(n) % 3 == 0).astype(np.uint8))
        self.out4 = A("record", 4, torch.float64)

    def buffers(self, row=None, z=False, fusedable=False):
        from gym_fishing_amd import _capi
        p = lambda x: x.data_ptr() if x is not None else None  # noqa: E731
        act = self.ring[row * self.stride:].data_ptr() if row is not None else None
        return _capi.make_buffers(obs=p(self.obs), action=act, reward=p(self.reward), done=p(self.done),
                                  done_bits=None if fusedable else p(self.done_bits), t=p(self.t), r=p(self.r), K=p(self.K),
                                  sigma=p(self.sigma), z_ext=p(self.z) if z else None,
                                  terminal_obs=None if fusedable else p(self.terminal), ep_return=p(self.ep_return),
                                  return_partials=p(self.partials), model_idx=p(self.model_idx), v4_stamp=p(self.stamp))


def drive(hh, fill, case, dtype, n, padded, optional, auto):
    """One arena, every entry point that takes this shard; returns the streams' bytes."""
    import torch
    from gym_fishing_amd import _capi
    lib = _capi.lib()
    name, model, kw, need = case
    sfx = "f32" if np.dtype(dtype) == np.float32 else "f64"
    cap = (n + TILE - 1) // TILE * TILE if padded else n
    n_rows, T, off, seed = 3, 4, 8, 123
    kw = dict(dict(Tmax=3, auto_reset=auto, padded=padded), **kw)
    if need.get("mixed"):
        kw.update(models=[2, 0, 4, 1], zoo_table=[dict(d, sigma=0.1) for d in fo.V11_TABLE])
    derived = bool(kw.get("derived"))
    p = hh.params(model, origin=(5, 0), **kw)
    ar = Arena(fill)
    g = Guarded(hh, ar, n, cap, dtype, model, need, optional, n_rows)
    what = "%s %s n=%d padded=%s optional=%s auto=%s" % (name, sfx, n, padded, optional, auto)

    def ok(rc, call, allowed=()):
        assert rc == 0 or rc in allowed, "%s: %s returned %d (%s)" % (what, call, rc, lib.fishing_error_string(rc))
        return rc == 0

    ok(getattr(lib, "fishing_reset_" + sfx)(p, n, off, g.buffers(), None, seed, 0, None), "reset(all)")
    step = getattr(lib, "fishing_step_" + sfx)
    c = 5
    for k in range(3):
        ok(step(p, n, off, g.buffers(k % n_rows), seed, c, None), "step")
        c += 1
    if not need.get("mixed"):       # (external noise; fishing-v11 takes it too, but one path is enough there)
        ok(step(p, n, off, g.buffers(0, z=True), seed, c, None), "step(z_ext)")
        c += 1
    # a masked reset: stored-array fishing-v4 redraws the masked envs' (K, r); the derived mode needs the stamps
    rc = getattr(lib, "fishing_reset_" + sfx)(p, n, off, g.buffers(), g.mask.data_ptr(), seed, 1, None)
    masked_ok = ok(rc, "reset(mask)", allowed=(ERR_UNSUPPORTED,) if derived and not need.get("stamp") else ())
    assert masked_ok or (derived and g.stamp is None)
    ok(step(p, n, off, g.buffers(1), seed, c, None), "step after reset(mask)")
    c += 1
    ok(getattr(lib, "fishing_step_many_" + sfx)(p, n, off, g.buffers(0), g.stride, n_rows, T, seed, c, None), "step_many")
    c += T
    # fused steps with per-step rows (rows padded to 16 elements, the last row ends with ITS 16-element stride)
    ostride = (n + 15) // 16 * 16
    rs = ar.alloc("reward_steps", T * ostride, g.td)
    ds = ar.alloc("done_steps", T * ostride, torch.uint8)
    ok(getattr(lib, "fishing_step_fused_" + sfx)(p, n, off, g.buffers(0, fusedable=True), g.stride, n_rows, T, rs.data_ptr(),
                                                  ds.data_ptr(), ostride, seed, c, None), "step_fused(rows)")
    c += T
    ok(getattr(lib, "fishing_step_fused_" + sfx)(p, n, off, g.buffers(0, fusedable=True), g.stride, n_rows, T, None, None, 0,
                                                  seed, c, None), "step_fused")
    c += T
    # fused rollouts: every in-kernel policy; the trajectory rows need n % 4 == 0
    roll = getattr(lib, "fishing_rollout_" + sfx)
    unsupported = (ERR_UNSUPPORTED,) if derived and not auto else ()
    for policy, param in ((_capi.POLICY_RANDOM, 0.0), (_capi.POLICY_ESCAPEMENT, 0.5), (_capi.POLICY_MSY, 0.05),
                          (_capi.POLICY_CONSTANT, -0.9)):
        if ok(roll(p, n, off, g.buffers(fusedable=True), policy, param, T, None, seed, c, None), "rollout", unsupported):
            c += T
    # ... and with one parameter per env (ABI 7): exactly n of them
    pparams = ar.alloc("policy_params", n, g.td, np.linspace(0.0, 0.9, n))
    if ok(getattr(lib, "fishing_rollout_params_" + sfx)(p, n, off, g.buffers(fusedable=True), _capi.POLICY_ESCAPEMENT,
                                                        pparams.data_ptr(), T, None, seed, c, None), "rollout_params",
          (ERR_UNSUPPORTED,) if not auto else ()):
        c += T
    if n % 4 == 0:
        traj = ar.alloc("traj", T * 4 * n, g.td)
        if ok(roll(p, n, off, g.buffers(fusedable=True), _capi.POLICY_RANDOM, 0.0, T, traj.data_ptr(), seed, c, None),
              "rollout(traj)", unsupported):
            c += T
    if derived:
        Ko, ro = ar.alloc("K_out", n, g.td), ar.alloc("r_out", n, g.td)
        ok(getattr(lib, "fishing_v4_params_" + sfx)(p, n, off, g.t.data_ptr(), g.stamp.data_ptr() if g.stamp is not None else None,
                                                    Ko.data_ptr(), ro.data_ptr(), seed, c, None), "v4_params")
    if optional:
        ok(lib.fishing_reduce_returns_slots(g.partials.data_ptr(), g.slots, g.out4.data_ptr(), None), "reduce_returns_slots")
    # population_draw over n populations (per element: fishing-v11's growth function, fishing-v4's (r, K))
    x_in = ar.alloc("x_in", n, g.td, np.linspace(0.0, 1.5, n))
    x_out = ar.alloc("x_out", n, g.td)
    midx = ar.alloc("draw_model_idx", n, torch.int32, np.arange(n) % 5) if need.get("mixed") else None
    ok(getattr(lib, "fishing_population_draw_" + sfx)(p, n, x_in.data_ptr(), g.z.data_ptr(),
                                                      midx.data_ptr() if midx is not None else None,
                                                      g.r.data_ptr() if need.get("rk") else None,
                                                      g.K.data_ptr() if need.get("rk") else None, x_out.data_ptr(), None),
       "population_draw")
    if model in (fo.MODEL_V0, fo.MODEL_V1, fo.MODEL_V2, fo.MODEL_V4):
        n_states = 37
        states = ar.alloc("states", n_states, g.td, np.linspace(-1.0, 1.0, n_states))
        S = ar.alloc("S_out", n, g.td)
        ok(getattr(lib, "fishing_bmsy_sweep_" + sfx)(p, n, g.K.data_ptr() if need.get("rk") else None,
                                                     g.r.data_ptr() if need.get("rk") else None, states.data_ptr(), n_states,
                                                     S.data_ptr(), None), "bmsy_sweep")
    return what, ar.check(what), ar.item


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_no_entry_point_writes_outside_its_streams(hh, case, dtype):
    """Every size x {exact streams, padded tiles} x {all optional streams, none}: the guard bands survive, and the streams
    do not depend on what the bands hold."""
    runs = [(n, False) for n in SIZES] + [(n, True) for n in PADDED_SIZES]
    for i, (n, padded) in enumerate(runs):
        optional, auto = bool(i & 1), bool((i >> 1) & 1) or bool(case[2].get("derived"))
        what, a, item = drive(hh, 0xA5, case, dtype, n, padded, optional, auto)
        _, b, _ = drive(hh, 0x3C, case, dtype, n, padded, optional, auto)
        assert a.keys() == b.keys()
        for name in a:
            live = a[name].size
            if padded and name in PADDED_STATE_STREAMS:       # behind the n-th element a padded state stream is scratch
                live = n * item[name]
            elif padded and name == "done_bits":
                live = n // 64 * 8
            assert np.array_equal(a[name][:live], b[name][:live]), (what, name)


def test_generator_and_math_hooks_stay_inside_their_outputs(hh):
    """The diagnostic entry points (Philox words / normals, the elementary functions) at sizes around a wave and a tile."""
    import torch
    from gym_fishing_amd import _capi
    lib = _capi.lib()
    for n in (1, 3, 63, 64, 65, 1023, 1025, 4099):
        ar = Arena(0xA5, nbytes=2 << 20)
        words = ar.alloc("words", 4 * n, torch.int32)
        z0, z1 = ar.alloc("z0", n, torch.float32), ar.alloc("z1", n, torch.float32)
        for tag in (_capi.STREAM_NOISE, 1, 2, 3):
            assert lib.fishing_noise_f32(n, 4, 99, 7, tag, words.data_ptr(), z0.data_ptr(), z1.data_ptr(), None) == 0
        assert lib.fishing_step_normals_f32(n, 4, 99, 7, z0.data_ptr(), None) == 0
        for tag in (1, 2):
            assert lib.fishing_reset_normals_f32(n, 4, 99, 7, tag, z0.data_ptr(), z1.data_ptr(), None) == 0
        xin = ar.alloc("in", n, torch.float64, np.linspace(0.1, 3.0, n))
        xout = ar.alloc("out", n, torch.float64)
        for fn in MATH_FUNCTIONS:
            assert lib.fishing_math_f64(n, fn, xin.data_ptr(), xout.data_ptr(), None) == 0
        counter = ar.alloc("counter", 1, torch.int64)
        assert lib.fishing_counter_add(counter.data_ptr(), 5, None) == 0
        got = ar.check("hooks n=%d" % n)
        assert int(got["counter"].view(np.int64)[0]) == 5
        assert np.isfinite(got["out"].view(np.float64)).all()


def test_full_partials_buffer_and_both_reductions(hh):
    """fishing_reduce_returns reads fishing_partials_len() doubles and writes four; the slots form reads 4 * slots."""
    import torch
    from gym_fishing_amd import _capi
    lib = _capi.lib()
    ar = Arena(0xA5, nbytes=4 << 20)
    L = int(lib.fishing_partials_len())
    partials = ar.alloc("partials", L, torch.float64, np.arange(L, dtype=np.float64) % 7)
    out = ar.alloc("out4", 4, torch.float64)
    assert lib.fishing_reduce_returns(partials.data_ptr(), out.data_ptr(), None) == 0
    full = ar.check("reduce_returns")["out4"].view(np.float64).copy()
    want = (np.arange(L, dtype=np.float64) % 7).reshape(-1, 4).sum(axis=0)
    assert np.array_equal(full, want)
    slots = int(lib.fishing_partials_slots(1 << 20))
    assert lib.fishing_reduce_returns_slots(partials.data_ptr(), slots, out.data_ptr(), None) == 0
    part = ar.check("reduce_returns_slots")["out4"].view(np.float64)
    assert np.array_equal(part, (np.arange(4 * slots, dtype=np.float64) % 7).reshape(-1, 4).sum(axis=0))


def test_the_guard_bands_do_catch_an_overrun(hh):
    """Negative control: streams sized for 1024 envs, a step over 1028 -- the four envs too many land in the bands (inside
    the arena: nothing else is touched) and check() must say so."""
    import torch
    from gym_fishing_amd import _capi
    lib = _capi.lib()
    ar = Arena(0xA5, nbytes=1 << 20)
    n = 1024
    obs, t = ar.alloc("obs", n, torch.float32), ar.alloc("t", n, torch.int32)
    reward, done = ar.alloc("reward", n, torch.float32), ar.alloc("done", n, torch.uint8)
    action = ar.alloc("action", n + 4, torch.float32)
    b = _capi.make_buffers(obs=obs.data_ptr(), action=action.data_ptr(), reward=reward.data_ptr(), done=done.data_ptr(),
                           t=t.data_ptr())
    p = hh.params(fo.MODEL_V1, sigma=0.1)
    assert lib.fishing_step_f32(p, n, 0, b, 1, 0, None) == 0
    ar.check("in bounds")
    assert lib.fishing_step_f32(p, n + 4, 0, b, 1, 1, None) == 0
    with pytest.raises(AssertionError, match="guard bytes overwritten"):
        ar.check("four envs too many")
