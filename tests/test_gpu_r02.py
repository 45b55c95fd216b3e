"""GPU parity tests of the round-2 kernels, all through the C ABI (tests/hip_harness.py):

* fishing-v4 with derived parameters (FISHING_FLAG_V4_DERIVED: no r / K arrays, every kernel re-derives an
  env's (K, r) from the Philox2x32 block its year counter points at) == the stored-array mode, bit for bit;
* fishing_step_fused_* (K steps per launch, state in registers) == K fishing_step_* calls, bit for bit;
* the episodic-return record counts an episode once, however long a finished env is stepped on;
* fishing_step_kernel_name_* names the instantiation the dispatch picks.
"""
import numpy as np
import pytest

from oracle import fishing_oracle as fo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hh():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import hip_harness
    return hip_harness


def same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    it = {4: np.uint32, 8: np.uint64, 1: np.uint8}[a.dtype.itemsize]
    assert a.dtype == b.dtype and a.shape == b.shape, what
    bad = np.flatnonzero(a.view(it) != b.view(it))
    assert bad.size == 0, "%s: %d differing, first at %d: %r vs %r" % (what, bad.size, bad[0], a.flat[bad[0]], b.flat[bad[0]])


# ------------------------------------------------------------------ fishing-v4: derived == stored parameters
@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("kernel", ["lean", "general"])
@pytest.mark.parametrize("sigarr", [False, True], ids=["sigma_scalar", "sigma_array"])
def test_v4_derived_parameters_equal_stored_arrays(hh, dtype, kernel, sigarr):
    """Two fishing-v4 batches (N = 3 * 1024 + 77, env_offset 8), same seed and actions: one keeps r / K arrays
    (the redraw stores into them), the other runs under FISHING_FLAG_V4_DERIVED with NO arrays.  Over 220
    auto-resetting steps -- with a second full reset() at step 120, so both origin rules are exercised away from
    zero -- obs / reward / done / t / ep_return are bit-identical every step, and the (K, r) fishing_v4_params_*
    materialises from the year counters equal the stored arrays."""
    import torch
    from gym_fishing_amd import _capi
    lib = _capi.lib()
    n, off, seed = 3 * 1024 + 77, 8, 0xFEEDF00D12
    general = kernel == "general"
    kw = dict(sigma=0.1, Tmax=7, K_mean=1.0, r_mean=0.3, sigma_p=0.2, auto_reset=True, general=general)
    sig = np.random.default_rng(3).uniform(0.02, 0.2, n) if sigarr else None
    fn = lib.fishing_step_f32 if dtype == np.float32 else lib.fishing_step_f64
    S = hh.State(n, dtype, fo.MODEL_V4, np.zeros(n), r=np.full(n, 0.3), K=np.full(n, 1.0), sigma=sig, ep_return=True)
    D = hh.State(n, dtype, fo.MODEL_V4, np.zeros(n), sigma=sig, ep_return=True)
    assert D.r is None and D.K is None
    g = torch.Generator(device="cuda").manual_seed(11)
    origin, resets = (0, 0), 0
    ps = hh.params(fo.MODEL_V4, **kw)
    pd = hh.params(fo.MODEL_V4, derived=True, origin=origin, **kw)
    S.reset(ps, seed=seed, counter=resets, env_offset=off)
    D.reset(pd, seed=seed, counter=resets, env_offset=off)
    finished = 0
    for s in range(220):
        if s == 120:        # a reset of all envs in mid-run: new origin (step count 120, reset counter 1)
            resets = 1
            origin = (s, resets)
            pd = hh.params(fo.MODEL_V4, derived=True, origin=origin, **kw)
            S.reset(ps, seed=seed, counter=resets, env_offset=off)
            D.reset(pd, seed=seed, counter=resets, env_offset=off)
        # a wide action range: many envs fish themselves out early, others run to Tmax + 1
        a = (torch.rand(n, device="cuda", generator=g) * 1.3 - 1.15).float()
        Kd, rd = D.v4_params(pd, seed=seed, step_counter=s, env_offset=off)      # in force BEFORE the step
        same(Kd, S.K.cpu().numpy(), "K before step %d" % s)
        same(rd, S.r.cpu().numpy(), "r before step %d" % s)
        assert fn(ps, n, off, S.buffers(a), seed, s, None) == 0
        assert fn(pd, n, off, D.buffers(a), seed, s, None) == 0
        torch.cuda.synchronize()
        for name in ("obs", "reward", "done", "t", "ep_return"):
            assert torch.equal(getattr(S, name), getattr(D, name)), (name, s)
        finished += int(S.done.sum())
    assert finished > 20 * n                 # plenty of redraws happened
    assert len(np.unique(S.K.cpu().numpy())) > n // 2
    rs, rd_ = S.record(), D.record()
    assert rs[2] == rd_[2] == finished and np.array_equal(rs, rd_)
    if not general:
        want = fo_mask(derived=True, sigarr=sigarr, ret=True, one=True) if dtype == np.float32 else None
        name = hh.kernel_name(pd, n, D.buffers(a), dtype)
        assert name.startswith("fishing::step_kernel_lean<%s, 4, " % ("float" if dtype == np.float32 else "double")), name
        if want is not None:
            assert name.endswith(", %d>" % want), (name, want)


def fo_mask(noise=2, ret=False, sigarr=False, t8=False, term=False, bits=False, zz=False, derived=False, drift=False, one=False):
    """Feature mask of step_kernel_lean (csrc/fishing_step.hip: namespace feat); `one` = a tile per workgroup (grid == tiles)."""
    return (noise | (4 if ret else 0) | (8 if sigarr else 0) | (16 if t8 else 0) | (32 if term else 0) | (64 if bits else 0)
            | (128 if zz else 0) | (256 if derived else 0) | (512 if drift else 0) | (8192 if one else 0))


def test_v4_derived_mode_guards_its_year_counter_and_hands_out_snapshots(hh):
    """In the derived mode the year counter dates each env's episode and with it its (K, r).  So env.years_passed hands
    out a COPY there (an in-place edit of it changes nothing), an ASSIGNMENT first moves the env to stored r / K arrays
    (the parameters in force stay what they were; only the Tmax check follows the new counter), and env.K / env.r are
    snapshots: editing one in place does not touch the env.  Outside the derived mode years_passed is the live tensor."""
    import torch
    import gym_fishing_amd as gf
    n = 2048
    env = gf.make("fishing-v4", sigma=0.05, sigma_p=0.2, num_envs=n, seed=3, Tmax=6)
    env.reset()
    acts = torch.rand((3, n), device="cuda") * 1.4 - 1.2
    env.step_many(acts, 5)
    assert env._derived
    K0, r0 = env.K.clone(), env.r.clone()
    yp = env.years_passed
    assert yp.data_ptr() != env._t.data_ptr() and torch.equal(yp, env._t)
    yp.zero_()                                   # an outside in-place edit of the copy
    env.K.fill_(7.0)                             # ... and of a K snapshot
    assert env._derived and torch.equal(env.K, K0) and torch.equal(env.r, r0) and int(env._t.max()) > 0
    twin = gf.make("fishing-v4", sigma=0.05, sigma_p=0.2, num_envs=n, seed=3, Tmax=6)
    twin.reset()
    twin.step_many(acts, 5)
    env.years_passed = torch.zeros(n, dtype=torch.int32, device="cuda")        # an assignment: stored arrays from here on
    assert not env._derived and torch.equal(env.K, K0) and torch.equal(env.r, r0) and int(env._t.max()) == 0
    assert env.years_passed.data_ptr() == env._t.data_ptr()                    # the live tensor again
    # the parameters in force did not move: the next step's observations equal the twin's wherever the twin's env does not
    # finish on it (there the two differ only by the year counter the caller rewrote)
    oa, _, da, _ = env.step(acts[0])
    ob, _, db, _ = twin.step(acts[0])
    same = ~(da.bool() | db.bool())
    assert int(same.sum()) > n // 4 and torch.equal(oa[same], ob[same])


def test_v4_derived_parameters_against_the_oracle(hh):
    """The derived mode end to end against the oracle: the oracle dates every env's episode with v4_origin() and
    draws (K, r) from reset_normals() (its own Philox2x32), the device's Box-Muller being within 2e-5 of libm's;
    so K / r agree to 1e-5 and the float64 trajectories stay within 1e-4 over 40 steps."""
    n, off, seed = 2048 + 12, 4, 77
    kw = dict(sigma=0.05, Tmax=5, K_mean=1.0, r_mean=0.3, sigma_p=0.1, auto_reset=True)
    p = hh.params(fo.MODEL_V4, derived=True, origin=(0, 0), **kw)
    D = hh.State(n, np.float64, fo.MODEL_V4, np.zeros(n))
    D.reset(p, seed=seed, counter=0, env_offset=off)
    env = np.arange(off, off + n, dtype=np.uint64)
    rng = np.random.default_rng(1)
    t = np.zeros(n, np.int64)
    for s in range(40):
        stream, counter = fo.v4_origin(s, t, 0, 0)
        zK = np.empty(n, np.float32)
        zr = np.empty(n, np.float32)
        for st_, c in set(zip(stream.tolist(), counter.tolist())):
            m = (stream == st_) & (counter == c)
            zK[m], zr[m] = fo.reset_normals(seed, env[m], c, st_)
        K, r = fo.draw_model_error_params(zK, zr, 1.0, 0.3, 0.1, np.float64)
        Kd, rd = D.v4_params(p, seed=seed, step_counter=s, env_offset=off)
        assert np.abs(Kd - K).max() < 1e-5 and np.abs(rd - r).max() < 1e-5, s
        a = rng.uniform(-1.1, 0.1, n).astype(np.float32)
        obs_in = D.obs.cpu().numpy()
        o, rew, done, t2 = D.step(p, a, seed=seed, step_counter=s, env_offset=off)
        z = hh.device_step_noise(n, seed, s, off).astype(np.float64)
        eo, er, ed, et, _ = fo.step(fo.MODEL_V4, obs_in, t.astype(np.int32), a, z, rd, Kd, 0.05, Tmax=5)
        same(rew, er, "reward step %d" % s)
        assert np.array_equal(done, ed)
        exp_obs = np.where(ed.astype(bool), 0.75, eo)        # fishing-v4 restarts at x0 un-normalised (quirk B8)
        same(o, exp_obs, "obs step %d" % s)
        t = np.where(ed.astype(bool), 0, et).astype(np.int64)
        assert np.array_equal(t2, t)


# ------------------------------------------------------------------ fused step_many == per-step launches
FUSED_CASES = [
    ("v0", fo.MODEL_V0, dict(sigma=0.1, n_actions=100), False),
    ("v1", fo.MODEL_V1, dict(sigma=0.1), False),
    ("v1_K2", fo.MODEL_V1, dict(sigma=0.1, K=2.0, r=0.5, x0=1.1), False),      # power-of-two K: the exact x * (1/K)
    ("v1_K3", fo.MODEL_V1, dict(sigma=0.1, K=3.0), False),                     # ... and a K that keeps the division
    ("v1_quiet", fo.MODEL_V1, dict(sigma=0.0), False),
    ("v2", fo.MODEL_V2, dict(sigma=0.1, C=0.5), False),
    ("v4_stored", fo.MODEL_V4, dict(sigma=0.05, sigma_p=0.2), False),
    ("v4_derived", fo.MODEL_V4, dict(sigma=0.05, sigma_p=0.2), True),
    ("v6", fo.MODEL_V6, dict(sigma=0.1), False),
    ("v7", fo.MODEL_V7, dict(sigma=0.1, r=0.7, K=1.5, M=1.5, q=3.0, b=0.15, a=0.2), False),
    ("v10", fo.MODEL_V10, dict(sigma=0.1, r=0.8, alpha=-0.007), False),
    ("v11", fo.MODEL_V11, dict(sigma=0.0), False),             # growth function per env, redrawn per episode
    ("v11_sigma_array", fo.MODEL_V11, dict(sigma=0.0), False),   # ... with a per-env sigma (the per-lane switch path)
]


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("auto", [True, False], ids=["auto_reset", "no_reset"])
@pytest.mark.parametrize("case", FUSED_CASES, ids=[c[0] for c in FUSED_CASES])
def test_fused_step_many_equals_per_step_launches(hh, case, auto, dtype):
    """fishing_step_fused_*: 23 steps in ONE launch (action ring of 5 rows, so it wraps; N = 2 * 1024 + 37, ragged;
    env_offset 12; start counter 100) against 23 fishing_step_* launches with the same counters.  Every per-step
    reward / done row, the final obs / t / ep_return / (K, r) and the return record are bit-identical -- for
    every model family, with and without auto-reset (step() semantics: a finished env is stepped on)."""
    import torch
    from gym_fishing_amd import _capi
    lib = _capi.lib()
    case_name, model, kw, derived = case
    n, off, seed, T, R, c0 = 2 * 1024 + 37, 12, 4242, 23, 5, 100
    per_env = model == fo.MODEL_V4
    drift = model == fo.MODEL_V10
    mixed = model == fo.MODEL_V11
    kw = dict(kw, Tmax=6, auto_reset=auto)
    if mixed:
        kw.update(models=[4, 0, 3, 1, 2], zoo_table=[dict(d, sigma=0.1) for d in fo.V11_TABLE])
    pk = dict(derived=True, origin=(c0, 0)) if derived else {}
    p = hh.params(model, **kw, **pk)
    rng = np.random.default_rng(5)
    if model == fo.MODEL_V0:
        ring = rng.integers(0, 100, (R, n)).astype(np.int32)
    else:
        ring = rng.uniform(-1.1, 0.2, (R, n)).astype(np.float32)

    def mk():
        st = hh.State(n, dtype, model, np.zeros(n), r=(np.full(n, kw.get("r", 0.3)) if (per_env and not derived) or drift else None),
                      K=np.full(n, 1.0) if per_env and not derived else None, ep_return=True,
                      model_idx=np.zeros(n, np.int32) if mixed else None,
                      sigma=np.linspace(0.02, 0.2, n) if case_name == "v11_sigma_array" else None)
        st.reset(p, seed=seed, counter=0, env_offset=off)
        return st
    A, B = mk(), mk()
    fn = lib.fishing_step_f32 if dtype == np.float32 else lib.fishing_step_f64
    rows_r, rows_d = [], []
    ring_dev = A.ring_tensor(ring)
    for s in range(T):
        assert fn(p, n, off, A.buffers(ring_dev[s % R]), seed, c0 + s, None) == 0
        torch.cuda.synchronize()
        rows_r.append(A.reward.cpu().numpy())
        rows_d.append(A.done.cpu().numpy())
    rs, ds = B.step_fused(p, ring, T, seed=seed, step_counter=c0, env_offset=off)
    for s in range(T):
        same(rs[s], rows_r[s], "reward row %d" % s)
        assert np.array_equal(ds[s], rows_d[s]), "done row %d" % s
    names = (["obs", "t", "reward", "done", "ep_return"] + (["K", "r"] if per_env and not derived else []) + (["r"] if drift else [])
             + (["model_idx"] if mixed else []))
    for name in names:
        assert torch.equal(getattr(A, name), getattr(B, name)), name
    ra, rb = A.record(), B.record()
    assert ra[2] == rb[2] and ra[3] == rb[3] and np.allclose(ra[:2], rb[:2], rtol=1e-12)
    assert ra[2] > 0
    if derived:
        Ka, ra_ = A.v4_params(p, seed=seed, step_counter=c0 + T, env_offset=off)
        Kb, rb_ = B.v4_params(p, seed=seed, step_counter=c0 + T, env_offset=off)
        same(Ka, Kb, "derived K after the run")
        same(ra_, rb_, "derived r after the run")
    # ... and without the per-step rows (only the last step's reward / done are written)
    C = mk()
    C.step_fused(p, ring, T, seed=seed, step_counter=c0, env_offset=off, per_step=False)
    for name in ("obs", "t", "reward", "done", "ep_return") + (("model_idx",) if mixed else ()):
        assert torch.equal(getattr(A, name), getattr(C, name)), name


def test_fused_step_many_at_the_launch_bound_sizes(hh):
    """BASELINE configs 2 and 4's per-GPU shard (N = 2^20 fishing-v1, 2^19 fishing-v2): 101 fused steps == 101
    launches on all envs, plus the compact (uint8 year counter) layout."""
    import torch
    from gym_fishing_amd import _capi
    lib = _capi.lib()
    for model, n, t8 in ((fo.MODEL_V1, 1 << 20, False), (fo.MODEL_V2, 1 << 19, False), (fo.MODEL_V1, 1 << 18, True)):
        p = hh.params(model, sigma=0.1, C=0.5, auto_reset=True, t_u8=t8)
        g = torch.Generator(device="cuda").manual_seed(n)
        ring = (torch.rand((8, n), device="cuda", generator=g) * 2 - 1).float()
        A = hh.State(n, np.float32, model, np.full(n, -0.25), ep_return=True, t_u8=t8)
        B = hh.State(n, np.float32, model, np.full(n, -0.25), ep_return=True, t_u8=t8)
        assert lib.fishing_step_many_f32(p, n, 0, A.buffers(ring), n, 8, 101, 7, 0, None) == 0
        assert lib.fishing_step_fused_f32(p, n, 0, B.buffers(ring), n, 8, 101, None, None, 0, 7, 0, None) == 0
        torch.cuda.synchronize()
        for name in ("obs", "t", "reward", "done", "ep_return"):
            assert torch.equal(getattr(A, name), getattr(B, name)), (model, name)
        ra, rb = A.record(), B.record()
        assert ra[2] == rb[2] > n // 2 and ra[3] == rb[3] and np.allclose(ra[:2], rb[:2], rtol=1e-12)


# ------------------------------------------------------------------ the return record counts an episode once
@pytest.mark.parametrize("kernel", ["lean", "general", "fused"])
def test_finished_envs_stepped_on_are_recorded_once(hh, kernel):
    """Without auto-reset a finished env keeps being stepped (the reference allows it: quirk B7) and stays done;
    the episodic-return record must hold each env's episode once -- at the step its done flag first rose."""
    import torch
    n, Tmax, T = 2048, 5, 14
    p = hh.params(fo.MODEL_V1, sigma=0.1, Tmax=Tmax, auto_reset=False, general=(kernel == "general"))
    st = hh.State(n, np.float32, fo.MODEL_V1, np.full(n, -0.25), ep_return=True)
    rng = np.random.default_rng(0)
    ring = rng.uniform(-1.0, -0.2, (T, n)).astype(np.float32)
    ring[:, ::3] = 1.0                       # every third env takes the whole stock at once: done at step 0
    first_done = np.full(n, -1)
    ret_at_done = np.zeros(n, np.float32)
    running = np.zeros(n, np.float32)
    if kernel == "fused":
        rs, ds = st.step_fused(p, ring, T, seed=3)
        for s in range(T):
            running = (running + rs[s]).astype(np.float32)
            new = (ds[s] == 1) & (first_done < 0)
            first_done[new] = s
            ret_at_done[new] = running[new]
    else:
        for s in range(T):
            _, rew, done, _ = st.step(p, ring[s], seed=3, step_counter=s)
            running = (running + rew).astype(np.float32)
            new = (done == 1) & (first_done < 0)
            first_done[new] = s
            ret_at_done[new] = running[new]
    assert (first_done >= 0).all() and (first_done[::3] == 0).all() and first_done.max() == Tmax
    rec = st.record()
    assert rec[2] == n, rec                                   # one episode per env, not one per step after the end
    assert rec[3] == (first_done + 1).sum()
    assert np.isclose(rec[0], ret_at_done.astype(np.float64).sum(), rtol=1e-6)


# ------------------------------------------------------------------ which kernel runs what
def test_kernel_names_follow_the_dispatch(hh):
    """fishing_step_kernel_name_* reports the instantiation the launch code picks: exact masks for the hot
    requests, the catch-all of the (T, MODEL) for everything else, the general kernel for fishing-v11 in float64,
    batches below one tile and the diagnostic flag."""
    n = 1 << 22
    st = hh.State(4096, np.float32, fo.MODEL_V1, np.zeros(4096), ep_return=True, terminal=True, done_bits=True)
    full = st.buffers(st.action_tensor(np.zeros(4096, np.float32)))

    def name(p, n=n, dtype=np.float32, **drop):
        from gym_fishing_amd import _capi
        b = _capi.FishingBuffers.from_buffer_copy(full)
        for k in ("terminal_obs", "done_bits", "ep_return", "return_partials"):
            if not drop.get(k, False):
                setattr(b, k, None)
        return hh.kernel_name(p, n, b, dtype)
    p1 = hh.params(fo.MODEL_V1, sigma=0.1, auto_reset=True)
    assert name(p1) == "fishing::step_kernel_lean<float, 1, 12290>"                # Philox (2) | KP2 (4096): K = 1 | ONE (8192): a tile per workgroup
    assert name(p1, ep_return=True, return_partials=True) == "fishing::step_kernel_lean<float, 1, 12294>"
    # a workgroup per tile up to 65536 tiles (round 3: return_partials has that many slots); the walk direction of the
    # one-tile forms is a run-time flag
    assert name(p1, n=1 << 25) == "fishing::step_kernel_lean<float, 1, 12290>"
    assert name(p1, n=1 << 26) == "fishing::step_kernel_lean<float, 1, 12290>"
    assert name(p1, n=(1 << 22) + 1024) == "fishing::step_kernel_lean<float, 1, 12290>"
    assert name(p1, n=(1 << 26) + 1024) == "fishing::step_kernel_lean<float, 1, 12290>"      # (beyond: ranges of 2^26 envs)
    # on an explicitly capped grid: the tile loop (round 3's compile-time zig-zag twins, mask bit 128, are gone: every form
    # takes its walk direction from a run-time flag)
    pc = hh.params(fo.MODEL_V1, sigma=0.1, auto_reset=True, launch_blocks=4096)
    assert name(pc, n=1 << 24) == "fishing::step_kernel<float, 1>"                  # ... the general kernel's
    assert name(pc, n=1 << 22) == "fishing::step_kernel_lean<float, 1, 12290>"      # (4096 workgroups cover 4096 tiles one to one)
    assert name(p1, n=1 << 20) == "fishing::step_kernel_lean<float, 1, 12290>"
    assert name(p1, n=1 << 24, ep_return=True, return_partials=True) == "fishing::step_kernel_lean<float, 1, 12294>"
    assert name(p1, terminal_obs=True) == "fishing::step_kernel_lean<float, 1, 11391>"
    assert name(p1, terminal_obs=True, done_bits=True) == "fishing::step_kernel_lean<float, 1, 11391>"
    assert name(p1, dtype=np.float64) == "fishing::step_kernel_lean<double, 1, 12290, 2>"       # float64, cache-resident: 2 envs per thread (512-thread workgroups), exact
    assert name(hh.params(fo.MODEL_V1, sigma=0.1, K=1.5, auto_reset=True), dtype=np.float64) == "fishing::step_kernel_lean<double, 1, 11391, 2>"
    assert name(hh.params(fo.MODEL_V1, sigma=0.1, K=1.5, auto_reset=True), n=1 << 20, dtype=np.float64) == "fishing::step_kernel_lean<double, 1, 11391, 2>"
    assert name(p1, n=1 << 24, dtype=np.float64) == "fishing::step_kernel_lean<double, 1, 11391>"
    assert name(hh.params(fo.MODEL_V4, sigma=0.1, derived=True), dtype=np.float64) == "fishing::step_kernel_lean<double, 4, 28031, 2>"     # (its catch-all: DERIVED | STAMP "may be there")
    assert name(hh.params(fo.MODEL_V1, sigma=0.1, t_u8=True)) == "fishing::step_kernel_lean<float, 1, 12306>"
    assert name(hh.params(fo.MODEL_V0, sigma=0.1)) == "fishing::step_kernel_lean<float, 0, 12290>"
    # a K that is not a power of two keeps the correctly rounded division
    assert name(hh.params(fo.MODEL_V1, sigma=0.1, K=1.5)) == "fishing::step_kernel_lean<float, 1, 8194>"
    assert name(hh.params(fo.MODEL_V1, sigma=0.1, K=0.25)) == "fishing::step_kernel_lean<float, 1, 12290>"
    assert name(hh.params(fo.MODEL_V9, sigma=0.1)) == "fishing::step_kernel_lean<float, 104, 8194>"
    assert name(hh.params(fo.MODEL_V1, sigma=0.1, general=True)) == "fishing::step_kernel<float, 1>"
    # the return record without auto-reset needs the latch, which only the catch-all carries
    assert name(hh.params(fo.MODEL_V1, sigma=0.1, auto_reset=False), ep_return=True, return_partials=True) == \
        "fishing::step_kernel_lean<float, 1, 11391>"
    assert name(p1, n=1000) == "fishing::step_kernel<float, 1>"
    assert name(hh.params(fo.MODEL_V4, sigma=0.1, derived=True)) == "fishing::step_kernel_lean<float, 4, 8450>"
    # fishing-v11 (growth function per env): the lean kernel in both layouts, exact instantiations (float64: round 4)
    p11 = hh.params(fo.MODEL_V11, sigma=0.1, models=[0, 1, 2, 3, 4], zoo_table=[dict(d, sigma=0.1) for d in fo.V11_TABLE], auto_reset=True)
    b11 = hh.State(4096, np.float32, fo.MODEL_V11, np.zeros(4096), model_idx=np.zeros(4096, np.int32), ep_return=True)
    full11 = b11.buffers(b11.action_tensor(np.zeros(4096, np.float32)))
    assert hh.kernel_name(p11, n, full11) == "fishing::step_kernel_lean<float, 105, 8198>"
    assert hh.kernel_name(p11, n, full11, np.float64) == "fishing::step_kernel_lean<double, 105, 8198>"


# ------------------------------------------------------------------ the host mirror in the derived mode
def test_env_v4_derived_mode_equals_the_stored_mode_and_survives_its_exits(hh):
    """make("fishing-v4", num_envs=N) keeps no r / K arrays (derived_params defaults to on for the Philox streams);
    derived_params=False keeps them.  Same seed => same trajectories and the same env.K / env.r, through step(),
    step_many(), the fused rollout, a mid-run full reset(), masked resets (per-env origin stamps: the derived mode
    stays), and the exits from the derived mode: env.K = ..., seed()."""
    import torch
    import gym_fishing_amd as gf
    n = 4096 + 8
    mk = lambda derived: gf.make("fishing-v4", num_envs=n, sigma=0.05, sigma_p=0.2, Tmax=6, seed=5, env_offset=16,   # noqa: E731
                                 track_returns=True, derived_params=derived)
    D, S = mk(None), mk(False)
    assert D._derived and D._K_arr is None and not S._derived and S._K_arr is not None
    g = torch.Generator(device="cuda").manual_seed(0)

    def check(tag):
        torch.cuda.synchronize()
        for name in ("_obs", "_t", "_reward", "_done", "_ep_return"):
            assert torch.equal(getattr(D, name), getattr(S, name)), (tag, name)
        assert torch.equal(D.K, S.K) and torch.equal(D.r, S.r), tag
    check("constructor")                                         # the constructor's draw (reset counter 0)
    for e in (D, S):
        e.reset()
    check("reset")
    for s in range(30):
        a = torch.rand(n, device="cuda", generator=g) * 1.3 - 1.15
        for e in (D, S):
            e.step(a)
        check("step %d" % s)
    ring = torch.rand((4, n), device="cuda", generator=g) * 1.3 - 1.15
    for e in (D, S):
        e.step_many(ring, 11)
    check("step_many")
    for e in (D, S):
        e.step_many(ring, 9, fused=True)
    check("fused step_many")
    for e in (D, S):
        e.rollout(13, policy="random")
    check("fused rollout")
    for e in (D, S):
        e.reset()                                                # a full reset in mid-run: new origin
    assert D._derived and D._origin == (63, 2)
    for e in (D, S):
        e.step_many(ring, 7)
    check("after the second reset")
    sd = D.state_dict()                                          # checkpoint in the derived mode
    mask = torch.zeros(n, dtype=torch.bool, device="cuda")
    mask[::5] = True
    for e in (D, S):
        e.reset(mask)                                            # envs restart at different times -> per-env origin stamps
    # (round 4: the masked reset keeps the derived mode -- no r / K arrays -- on the catch-all's stamped form, 45 B per env-step)
    assert D._derived and D._K_arr is None and D._stamp is not None
    assert D.step_kernel_name(ring[0]) == "fishing::step_kernel_lean<float, 4, 28031, 4>"
    assert int((D._stamp != 0).sum()) == int(mask.sum()) and int(D._stamp.max()) == D._reset_count
    check("right after a masked reset")
    for s in range(9):
        for e in (D, S):
            e.step(ring[s % 4])
        check("step %d after a masked reset" % s)
    for e in (D, S):
        e.step_many(ring, 8)
    check("step_many after a masked reset")
    for e in (D, S):
        e.step_many(ring, 7, fused=True)
    check("fused step_many after a masked reset")
    mask2 = torch.zeros(n, dtype=torch.bool, device="cuda")
    mask2[3::7] = True
    for e in (D, S):
        e.reset(mask2)                                           # a second masked reset: later stamps over earlier ones
        e.rollout(9, policy="random")
    check("fused rollout after a second masked reset")
    sd_stamped = D.state_dict()
    assert "_stamp" in sd_stamped and sd_stamped["v4_derived"]
    R2 = mk(None)
    R2.load_state_dict(sd_stamped)                               # a checkpoint taken in the stamped mode resumes in it
    assert R2._derived and R2._stamp is not None and torch.equal(R2._stamp, D._stamp) and torch.equal(R2.K, D.K)
    for e in (D, S, R2):
        e.step_many(ring, 6)
    check("after the stamped checkpoint")
    assert torch.equal(R2._obs, D._obs) and torch.equal(R2._t, D._t)
    for e in (D, S):
        e.reset()
    assert D._derived and D._K_arr is None and D._stamp is None  # a full reset clears the stamps: the stamp-free kernels again
    assert D.step_kernel_name(ring[0]) == "fishing::step_kernel_lean<float, 4, 8454, 4>"
    for e in (D, S):
        e.step_many(ring, 5)
        e.K = 1.25                                               # user-supplied parameters -> arrays
        e.step_many(ring, 5)
    assert not D._derived
    check("after env.K = 1.25")
    for e in (D, S):
        e.seed(77)                                               # new stream, parameters in force stay
        e.step_many(ring, 4)
        e.reset()
        e.step_many(ring, 6)
    assert D._derived
    check("after seed()")
    sa, sb = D.episode_stats(), S.episode_stats()
    assert sa["n_episodes"] == sb["n_episodes"] > n and sa["sum_return"] == sb["sum_return"]
    # resume from the checkpoint taken in the derived mode
    R = mk(None)
    R.load_state_dict(sd)
    assert R._derived and R._origin == (63, 2)
    D2 = mk(None)
    D2.load_state_dict(sd)
    for e in (R, D2):
        e.step_many(ring, 12)
    torch.cuda.synchronize()
    assert torch.equal(R._obs, D2._obs) and torch.equal(R.K, D2.K)


@pytest.mark.parametrize("trial", range(12))
def test_v4_random_operation_sequences_derived_equals_stored(hh, trial):
    """The fixed walk above, randomised: 12 seeds x 60 operations drawn from step / step_many / fused step_many / fused rollout
    (random, escapement) / full reset / masked reset (random mask, sometimes empty or all) / env.K read / env.sigma write /
    seed() / env.K write / checkpoint-and-restore into a fresh pair, in any order -- the derived batch (no r / K arrays, origin
    stamps after masked resets, arrays after the exits) and the stored-array batch agree bit for bit after every operation,
    whatever parameter mode the sequence has put the derived one in; graph replay of the derived batch follows as a third."""
    import torch
    import gym_fishing_amd as gf
    from gym_fishing_amd.graphs import GraphedSteps
    rng = np.random.default_rng(4100 + trial)
    n = int(rng.choice([1024, 2048 + 4, 4096 + 8, 1000]))
    mk = lambda derived: gf.make("fishing-v4", num_envs=n, sigma=0.05, sigma_p=0.2, Tmax=int(rng_T), seed=9 + trial,   # noqa: E731
                                 env_offset=16, track_returns=True, derived_params=derived)
    rng_T = rng.integers(3, 9)
    D, S, G = mk(None), mk(False), mk(None)
    g = torch.Generator(device="cuda").manual_seed(trial)
    ring = torch.rand((4, n), device="cuda", generator=g) * 1.3 - 1.15
    graph = None
    modes = set()

    def check(tag):
        torch.cuda.synchronize()
        for name in ("_obs", "_t", "_ep_return"):
            assert torch.equal(getattr(D, name), getattr(S, name)), (trial, tag, name)
            assert torch.equal(getattr(G, name), getattr(S, name)), (trial, tag, name, "graph")
        assert torch.equal(D.K, S.K) and torch.equal(D.r, S.r) and torch.equal(G.K, S.K), (trial, tag)
        modes.add("stored" if not D._derived else ("stamped" if D._stamp is not None else "derived"))

    for e in (D, S, G):
        e.reset()
    check("reset")
    ops = ["step", "step", "step_many", "fused", "rollout_random", "rollout_escapement", "reset", "mask", "mask", "read_K",
           "sigma", "seed", "write_K", "checkpoint", "graph", "graph"]
    for k in range(60):
        op = str(rng.choice(ops))
        if op == "step":
            for e in (D, S, G):
                e.step(ring[k % 4])
        elif op == "step_many":
            m = int(rng.integers(1, 9))
            for e in (D, S, G):
                e.step_many(ring, m)
        elif op == "fused":
            m = int(rng.integers(1, 9))
            for e in (D, S, G):
                e.step_many(ring, m, fused=True)
        elif op.startswith("rollout"):
            m = int(rng.integers(1, 12))
            pol = dict(policy="random") if op.endswith("random") else dict(policy="escapement", param=0.4)
            for e in (D, S, G):
                e.rollout(m, **pol)
        elif op == "reset":
            for e in (D, S, G):
                e.reset()
        elif op == "mask":
            kind = rng.random()
            mask = torch.zeros(n, dtype=torch.bool, device="cuda") if kind < 0.15 else (
                torch.ones(n, dtype=torch.bool, device="cuda") if kind < 0.3 else
                torch.as_tensor(rng.random(n) < rng.uniform(0.01, 0.6), device="cuda"))
            for e in (D, S, G):
                e.reset(mask)
        elif op == "read_K":
            assert torch.equal(D.K, S.K) and torch.equal(D.r, S.r)
        elif op == "sigma":
            v = float(rng.uniform(0.0, 0.1))
            for e in (D, S, G):
                e.sigma = v
        elif op == "seed":
            v = int(rng.integers(1, 1 << 30))
            for e in (D, S, G):
                e.seed(v)
        elif op == "write_K":
            v = float(rng.choice([1.25, 0.5, 2.0]))
            for e in (D, S, G):
                e.K = v
        elif op == "checkpoint":
            sds = [e.state_dict() for e in (D, S, G)]
            D, S, G = mk(None), mk(False), mk(None)
            for e, sd in zip((D, S, G), sds):
                e.load_state_dict(sd)
            graph = None
        elif op == "graph":
            # the third batch takes this operation as graph replays (captured once, re-captured when its launch signature moved);
            # the other two as plain step_many
            if graph is None or graph.env is not G:
                graph = GraphedSteps(G, ring, n_steps=3)
            reps = int(rng.integers(1, 4))
            for _ in range(reps):
                graph.replay()
                for e in (D, S):
                    e.step_many(ring, 3)         # (a call starts at the ring's first row, like a replay)
        check("%d %s" % (k, op))
    sa, sb = D.episode_stats(), S.episode_stats()
    assert sa["n_episodes"] == sb["n_episodes"] and sa["sum_return"] == sb["sum_return"]
    assert modes            # (which parameter modes the derived batch went through depends on the sequence; all three occur over the trials)


@pytest.mark.parametrize("trial", range(5))
@pytest.mark.parametrize("env_id", ["fishing-v0", "fishing-v1", "fishing-v2", "fishing-v5", "fishing-v7", "fishing-v8", "fishing-v10", "fishing-v11"])
def test_random_operation_sequences_every_family_three_ways(hh, env_id, trial):
    """The same walk for the other families: a batch on the kernels the dispatch picks, one forced onto the general kernel
    (launch_threads=128) and one whose plain steps run as hipGraph replays take 50 random operations -- step / step_many /
    fused step_many / fused rollouts / full and masked resets / env.sigma and env.Tmax writes / seed() / checkpoint-and-restore
    -- and agree bit for bit after every one (float32 or float64, whole tiles or a ragged padded batch, by the trial)."""
    import torch
    import gym_fishing_amd as gf
    from gym_fishing_amd.graphs import GraphedSteps
    rng = np.random.default_rng(5200 + 17 * trial + int(env_id.split("-v")[1]))
    n = int(rng.choice([1024, 3 * 1024 + 100, 4096, 1000]))
    dtype = torch.float32 if rng.random() < 0.6 else torch.float64
    kw = dict(num_envs=n, seed=3 + trial, Tmax=int(rng.integers(3, 9)), track_returns=True, dtype=dtype, env_offset=8)
    if env_id != "fishing-v11":
        kw["sigma"] = 0.1
    A, B, G = gf.make(env_id, **kw), gf.make(env_id, launch_threads=128, **kw), gf.make(env_id, **kw)
    g = torch.Generator(device="cuda").manual_seed(trial)
    if env_id == "fishing-v0":
        ring = torch.randint(0, 110, (4, n), device="cuda", generator=g, dtype=torch.int32)
    else:
        ring = torch.rand((4, n), device="cuda", generator=g) * 1.3 - 1.15
    graph = None

    def check(tag):
        torch.cuda.synchronize()
        for name in ("_obs", "_t", "_ep_return", "_r_arr", "_model_idx"):
            a = getattr(A, name)
            if a is None:
                continue
            for other, what in ((B, "general kernel"), (G, "graph")):
                o = getattr(other, name)
                assert torch.equal(a.view(torch.uint8), o.view(torch.uint8)), (env_id, trial, tag, name, what)

    for e in (A, B, G):
        e.reset()
    check("reset")
    ops = ["step", "step", "step_many", "fused", "rollout_random", "rollout_msy", "reset", "mask", "sigma", "Tmax", "seed",
           "checkpoint", "graph", "graph"]
    for k in range(50):
        op = str(rng.choice(ops))
        if op == "step":
            for e in (A, B, G):
                e.step(ring[k % 4])
        elif op == "step_many":
            m = int(rng.integers(1, 9))
            for e in (A, B, G):
                e.step_many(ring, m)
        elif op == "fused":
            m = int(rng.integers(1, 9))
            for e in (A, B, G):          # (the fused kernel keeps its own launch shape: the general-kernel batch steps one by one)
                e.step_many(ring, m, fused=e is not B)
        elif op.startswith("rollout"):
            m = int(rng.integers(1, 12))
            pol = dict(policy="random") if op.endswith("random") else dict(policy="msy", param=0.05)
            for e in (A, B, G):
                e.rollout(m, **pol)
        elif op == "reset":
            for e in (A, B, G):
                e.reset()
        elif op == "mask":
            mask = torch.as_tensor(rng.random(n) < rng.uniform(0.0, 0.7), device="cuda")
            for e in (A, B, G):
                e.reset(mask)
        elif op == "sigma" and env_id != "fishing-v11":
            v = float(rng.uniform(0.0, 0.15))
            for e in (A, B, G):
                e.sigma = v
        elif op == "Tmax":
            v = int(rng.integers(2, 10))
            for e in (A, B, G):
                e.Tmax = v
        elif op == "seed":
            v = int(rng.integers(1, 1 << 30))
            for e in (A, B, G):
                e.seed(v)
        elif op == "checkpoint":
            sds = [e.state_dict() for e in (A, B, G)]
            A, B, G = gf.make(env_id, **kw), gf.make(env_id, launch_threads=128, **kw), gf.make(env_id, **kw)
            for e, sd in zip((A, B, G), sds):
                e.load_state_dict(sd)
            graph = None
        elif op == "graph":
            if graph is None or graph.env is not G:
                graph = GraphedSteps(G, ring, n_steps=3)
            for _ in range(int(rng.integers(1, 4))):
                graph.replay()
                for e in (A, B):
                    e.step_many(ring, 3)
        check("%d %s" % (k, op))
    sa, sb, sg = A.episode_stats(), B.episode_stats(), G.episode_stats()
    assert sa["n_episodes"] == sb["n_episodes"] == sg["n_episodes"]
    assert abs(sa["sum_return"] - sb["sum_return"]) <= 1e-9 * max(1.0, abs(sa["sum_return"]))


def test_state_dict_round_trip_carries_sigma_and_scalar_attributes(hh):
    """load_state_dict() restores what FishingParams is built from: sigma changed after construction (env.sigma
    = ...), n_actions, C, the fishing-v4 means -- a freshly built env resumes bit for bit."""
    import torch
    import gym_fishing_amd as gf
    n = 2048
    g = torch.Generator(device="cuda").manual_seed(1)
    for env_id, kw, attr in (("fishing-v1", {}, None), ("fishing-v2", dict(C=0.4), "C"), ("fishing-v0", dict(n_actions=50), "n_actions")):
        A = gf.make(env_id, num_envs=n, sigma=0.0, seed=3, **kw)
        A.reset()
        A.sigma = 0.2                                            # after construction
        if attr == "C":
            A.C = 0.45
        acts = (torch.randint(0, 50, (3, n), device="cuda", generator=g, dtype=torch.int32) if env_id == "fishing-v0"
                else torch.rand((3, n), device="cuda", generator=g) - 1.0)
        A.step_many(acts, 5)
        sd = A.state_dict()
        B = gf.make(env_id, num_envs=n, sigma=0.0, seed=3)       # built with the defaults
        B.load_state_dict(sd)
        assert B.sigma == 0.2 and (attr is None or getattr(B, attr) == getattr(A, attr))
        A.step_many(acts, 6)
        B.step_many(acts, 6)
        torch.cuda.synchronize()
        assert torch.equal(A._obs, B._obs) and torch.equal(A._reward, B._reward), env_id


def test_v10_population_draw_drifts_r_like_the_reference(hh):
    """NonStationary.population_draw (growth_models.py:148-154) moves params['r'] by alpha on EVERY call -- the
    calls BMSY() / msy() make included -- and evaluates Beverton-Holt with the moved value."""
    import gym_fishing_amd as gf
    env = gf.make("fishing-v10", sigma=0.0, rng="philox")
    env.reset()
    r0, alpha = 0.8, -0.007
    x = np.array([0.3, 0.6, 0.9])
    for k in range(1, 4):
        got = env.population_draw(x, noise=np.zeros(3))
        r = r0 + k * alpha
        want = fo.zoo_population_draw(fo.KIND_OF_MODEL[fo.MODEL_V10], x, np.zeros(3), dict(r=r, K=1.0, sigma=0.0))
        assert np.allclose(got, want, rtol=1e-12, atol=0), (k, got, want)
        assert np.isclose(env.r, r)


# ------------------------------------------------------------------ simulate_mdp_vec, row for row
from conftest import load_vec_sims  # noqa: E402


@pytest.mark.parametrize("case", load_vec_sims(), ids=lambda c: c["key"])
def test_simulate_mdp_vec_reproduces_the_reference_table(hh, case):
    """shared_env.py:57-79 driven unmodified over N reference envs (tests/golden/reference_vec_sims.npz) against
    rollout.simulate_mdp_vec over the N-env batch seeded the same way (rng="numpy": one np.random.normal(0, 1, N) per
    step is the order in which a DummyVecEnv steps N reference envs): same row count and order (Tmax + 1 rows per env
    and batch, no break on done, auto-reset mid-table), same numbers -- bit for bit for fishing-v1, within the
    transcendental tolerance for fishing-v2 and the zoo (round 4: fishing-v5 / v7 / v9)."""
    import gym_fishing_amd as gf
    from gym_fishing_amd import policies, rollout
    env = gf.make(case["id"], num_envs=case["num_envs"], rng="numpy", dtype=__import__("torch").float64, **case["kwargs"])
    if case["policy"] == "constant":
        class Const:
            def predict(self, obs, **kw):
                import torch
                return torch.full((env.num_envs, 1), -0.45, dtype=torch.float32), obs
        model = Const()
    else:
        zoo = case["id"] not in ("fishing-v1", "fishing-v2")
        if zoo:
            # the zoo's growth functions read params["sigma"], not the env.sigma BMSY() zeroes: the sweep is noisy, S depends on
            # the stream -- the fixture seeds it on its own (tests/golden/make_golden.py)
            np.random.seed(5)
        model = getattr(policies, case["policy"])(env)
        if zoo:
            # float32 sweep on the device vs NumPy's log / exp: S (decided by the sweep's noise) is the reference's, msy to 1e-6;
            # then the reference's own numbers go in, so that the table compares the rollout and not the sweep
            assert model.S == case["S"], (model.S, case["S"])
            if case["msy"] is not None:
                assert abs(model.msy - case["msy"]) <= 1e-6
                model.msy = case["msy"]
        elif case["id"] == "fishing-v2":
            # the tipping-point growth curve is flat at its maximum and the device's exp differs from np.exp in the last
            # bit: the float32 sweep's argmax lands a few grid points (of 10001) away.  Take the reference's S so that
            # the table compares the rollout, not the sweep.
            assert abs(model.S - case["S"]) < 2e-3
            model.S = case["S"]
        else:
            assert model.S == case["S"]
        if case["msy"] is not None and not zoo:
            assert model.msy == case["msy"]
    np.random.seed(case["seed"])
    df = rollout.simulate_mdp_vec(env, model, case["n_eval_episodes"])
    got = df.to_numpy(dtype=np.float64) if hasattr(df, "to_numpy") else np.stack([df[c] for c in rollout.COLUMNS], 1)
    want = case["table"]
    assert got.shape == want.shape == (case["n_eval_episodes"] * (case["kwargs"]["Tmax"] + 1), 5)
    assert np.array_equal(got[:, [0, 4]], want[:, [0, 4]])              # time and rep columns: the row order
    if case["id"] != "fishing-v1":          # exp (fishing-v2) / log + exp (the zoo) on the device vs NumPy's, float64
        assert np.allclose(got, want, rtol=0, atol=1e-9)
    else:
        same(got, want, case["key"])


def test_simulate_mdp_vec_fused_path_equals_the_step_loop(hh):
    """With the Philox streams a model that names a kernel policy runs inside the fused rollout kernel; the table
    must equal the one the step-by-step loop builds from the same seed (same counters, same arithmetic)."""
    import gym_fishing_amd as gf
    from gym_fishing_amd import policies, rollout
    tabs = []
    for fused in (True, False):
        env = gf.make("fishing-v1", num_envs=8, sigma=0.1, Tmax=12, seed=3)
        model = policies.escapement(env)
        if not fused:
            del model.kernel_policy
        df = rollout.simulate_mdp_vec(env, model, 16)
        tabs.append(df.to_numpy(dtype=np.float64))
    assert tabs[0].shape == (16 * 13, 5)
    same(tabs[0], tabs[1], "fused vs step loop")


def test_simulate_mdp_vec_fishing_v4_rows_use_the_K_in_force(hh):
    """fishing-v4 redraws K at every reset, and the reference's table asks the env itself for the population of each row
    (df_entry_vec -> env_method("get_fish_population"), shared_env.py:15-26): a row after an auto-reset inside the table
    must use the NEW K.  A constant action that fishes the stock out every step makes every env reset every step; the
    table's state column must equal (obs + 1) * K with the K a twin env reports at that moment."""
    import torch
    import gym_fishing_amd as gf
    from gym_fishing_amd import rollout
    n, Tmax = 8, 5
    mk = lambda: gf.make("fishing-v4", num_envs=n, sigma=0.05, sigma_p=0.3, Tmax=Tmax, seed=21)  # noqa: E731
    env, twin = mk(), mk()
    df = rollout.simulate_mdp_vec(env, ("constant", 0.9), n)
    tab = df.to_numpy(dtype=np.float64)
    assert tab.shape == (n * (Tmax + 1), 5)
    twin.auto_reset = True
    twin.reset()
    a = torch.full((n,), 0.9, dtype=torch.float32, device="cuda")
    Ks, want = [], []
    for t in range(Tmax + 1):
        K = twin.K.to(torch.float64).reshape(-1).clone()
        Ks.append(K.cpu().numpy())
        want.append(((twin.state.reshape(-1).to(torch.float64) + 1.0) * K).cpu().numpy())
        if t < Tmax:
            twin.step(a)
    got = tab[:, 1].reshape(Tmax + 1, n)
    assert np.array_equal(got, np.stack(want))
    assert not np.array_equal(Ks[0], Ks[1]) and not np.array_equal(Ks[1], Ks[2])      # K really changed inside the table
    assert np.array_equal(tab[:, 0].reshape(Tmax + 1, n)[:, 0], np.arange(Tmax + 1))
    assert np.array_equal(tab[n:, 2], np.full(n * Tmax, np.float64(np.float32(0.9))))       # the raw action of the previous step


# ------------------------------------------------------------------ FISHING_FLAG_PADDED_TILES: a ragged batch in one launch
PADDED_CASES = [("v1", fo.MODEL_V1, {}), ("v1_ext_noise", fo.MODEL_V1, {}), ("v0", fo.MODEL_V0, {}), ("v2", fo.MODEL_V2, {}), ("v1_K3", fo.MODEL_V1, dict(K=3.0)),
                ("v4_stored", fo.MODEL_V4, {}), ("v4_derived", fo.MODEL_V4, dict(derived=True, origin=(7, 0))),
                ("v9", fo.MODEL_V9, {}), ("v10", fo.MODEL_V10, dict(r=0.8, alpha=-0.01)), ("v11", fo.MODEL_V11, {})]


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("case", PADDED_CASES, ids=[c[0] for c in PADDED_CASES])
def test_padded_tiles_flag_steps_a_ragged_batch_like_the_two_launch_path(hh, case, dtype):
    """With FISHING_FLAG_PADDED_TILES (state buffers hold whole 1024-env tiles) a batch of 4 * 1024 + 612 envs takes ONE
    lean launch; without it, the lean launch plus a one-workgroup launch of the general kernel for the tail.  Same bits
    for the n envs on every stream, same return record (the scratch envs behind the n-th never finish), 9 auto-resetting
    steps, every model family; the action tensor holds exactly n elements."""
    import torch
    from gym_fishing_amd import _capi
    lib = _capi.lib()
    _, model, kw = case
    n, cap, off, seed, c0 = 4 * 1024 + 612, 5 * 1024, 8, 77, 7
    per_env, drift, mixed = model == fo.MODEL_V4, model == fo.MODEL_V10, model == fo.MODEL_V11
    derived = kw.get("derived", False)
    kw = dict(dict(sigma=0.1, Tmax=3, auto_reset=True, sigma_p=0.2), **kw)
    if mixed:
        kw.update(models=[2, 0, 4], zoo_table=[dict(d, sigma=0.1) for d in fo.V11_TABLE])
    fn = lib.fishing_step_f32 if dtype == np.float32 else lib.fishing_step_f64
    rng = np.random.default_rng(3)
    acts = [rng.integers(0, 100, n).astype(np.int32) if model == fo.MODEL_V0 else rng.uniform(-1.1, 0.3, n).astype(np.float32)
            for _ in range(9)]
    zs = [rng.standard_normal(n) for _ in range(9)]
    outs = []
    for padded in (True, False):
        p = hh.params(model, padded=padded, **kw)
        st = hh.State(cap, dtype, model, np.zeros(cap), r=(np.full(cap, kw.get("r", 0.3)) if (per_env and not derived) or drift else None),
                      K=np.full(cap, 1.0) if per_env and not derived else None, ep_return=True, terminal=True,
                      model_idx=np.zeros(cap, np.int32) if mixed else None)
        assert getattr(lib, "fishing_reset_" + st.suffix)(p, n, off, st.buffers(), None, seed, 0, None) == 0
        for s_, a in enumerate(acts):
            at = torch.as_tensor(a).cuda()                   # exactly n elements: nothing may be read behind them
            zt = hh.dev(zs[s_].astype(dtype)) if case[0] == "v1_ext_noise" else None       # (the same for external noise)
            assert fn(p, n, off, st.buffers(at, zt), seed, c0 + s_, None) == 0
        torch.cuda.synchronize()
        outs.append(st)
    A, B = outs
    names = ["obs", "t", "reward", "done", "ep_return", "terminal"] + (["K", "r"] if per_env and not derived else []) + \
        (["r"] if drift else []) + (["model_idx"] if mixed else [])
    for name in names:
        assert _bits_equal(getattr(A, name)[:n], getattr(B, name)[:n]), (name, case[0])
    ra, rb = A.record(), B.record()
    assert ra[2] == rb[2] > 0 and ra[3] == rb[3] and np.allclose(ra[:2], rb[:2], rtol=1e-12, equal_nan=True)


@pytest.mark.parametrize("env_id", ["fishing-v1", "fishing-v4", "fishing-v11"])
def test_env_pads_its_streams_so_that_any_batch_size_steps_in_one_launch(hh, env_id):
    """make(id, num_envs=N) with N not a multiple of 1024 allocates room for whole tiles behind every per-env stream and
    sets FISHING_FLAG_PADDED_TILES; what the caller sees (shapes, results, episode statistics, state_dict round trip)
    is unchanged: equal to an env forced onto the general kernel (launch_threads=128), which needs no padding."""
    import torch
    import gym_fishing_amd as gf
    n = 3 * 1024 + 100
    kw = dict(num_envs=n, seed=9, Tmax=5, track_returns=True)
    if env_id != "fishing-v11":
        kw["sigma"] = 0.1
    A = gf.make(env_id, **kw)
    B = gf.make(env_id, launch_threads=128, **kw)
    assert A._padded and A._cap == 4 * 1024 and A._obs.shape == (n,) and A.state.shape == (n, 1)
    assert A._c_params().flags & 16 and B._c_params().flags & 16          # (the flag is the env's; the general kernel ignores it)
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = torch.rand((7, n), device="cuda", generator=g) * 1.4 - 1.2
    for e in (A, B):
        e.reset()
    for k in range(23):
        oa, ra, da, _ = A.step(acts[k % 7])
        ob, rb, db, _ = B.step(acts[k % 7])
        assert oa.shape == (n, 1) and ra.shape == (n,) and da.shape == (n,)
        assert _bits_equal(oa.reshape(-1), ob.reshape(-1)) and _bits_equal(ra, rb) and torch.equal(da, db), k
    sa, sb = A.episode_stats(), B.episode_stats()
    assert sa["n_episodes"] == sb["n_episodes"] > n and abs(sa["mean_return"] - sb["mean_return"]) < 1e-9
    # a padded env resumes from its own checkpoint
    sd = A.state_dict()
    C = gf.make(env_id, **kw)
    C.load_state_dict(sd)
    for k in range(5):
        oa, _, _, _ = A.step(acts[k])
        oc, _, _, _ = C.step(acts[k])
        assert _bits_equal(oa.reshape(-1), oc.reshape(-1)), k
    A.step_many(acts, 9, fused=True)
    C.step_many(acts, 9)
    assert _bits_equal(A.state.reshape(-1), C.state.reshape(-1))


# ------------------------------------------------------------------ randomised differential test of the three step paths
def _bits_equal(x, y):
    import torch
    it = {1: torch.uint8, 4: torch.int32, 8: torch.int64}[x.element_size()]
    return torch.equal(x.view(it), y.view(it))


@pytest.mark.parametrize("trial", range(48))
def test_randomised_requests_agree_across_dispatch_general_and_fused(hh, trial):
    """48 random requests -- model (v0 / v1 / v2 / v4 stored / v4 derived / three zoo kinds incl. the drifting v10),
    layout, N in [1, 7000] (whole tiles + ragged tails + sub-tile batches), auto-reset, return record, sigma array,
    one-byte year counter, noise mode (Philox / none / external) -- each stepped 7 times three ways: whatever
    instantiation the dispatch picks, the general kernel (diagnostic flag), and -- where it applies -- ONE fused
    launch.  Every stream must agree bit for bit (NaNs included), the return records to double rounding."""
    import torch
    from gym_fishing_amd import _capi
    lib = _capi.lib()
    rng = np.random.default_rng(9000 + trial)
    kind = ["v0", "v1", "v2", "v4s", "v4d", "v6", "v9", "v10"][trial % 8]
    model = {"v0": fo.MODEL_V0, "v1": fo.MODEL_V1, "v2": fo.MODEL_V2, "v4s": fo.MODEL_V4, "v4d": fo.MODEL_V4, "v6": fo.MODEL_V6,
             "v9": fo.MODEL_V9, "v10": fo.MODEL_V10}[kind]
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    n = int(rng.choice([rng.integers(1, 1024), 1024, 2048, rng.integers(1025, 7000), 4096 + 3]))
    auto = bool(rng.random() < 0.6)
    ret = bool(rng.random() < 0.6)
    sigarr = bool(rng.random() < 0.35)
    derived = kind == "v4d"
    t8 = bool(rng.random() < 0.25) and not derived
    noise = rng.choice(["philox", "philox", "none", "ext"])
    T, off, seed, c0 = 7, 4 * int(rng.integers(0, 50)), int(rng.integers(1, 1 << 40)), int(rng.integers(0, 300))
    kw = dict(sigma=0.0 if noise == "none" else 0.12, C=0.5, Tmax=4, sigma_p=0.15, auto_reset=auto, t_u8=t8)
    if kind in ("v0", "v1", "v2"):      # K = 2^k takes the exact-multiply instantiations, any other K the true division
        kw["K"] = float(rng.choice([1.0, 1.0, 2.0, 0.5, 1.5, 3.0]))
        kw["x0"] = 0.75 * kw["K"]
    if kind == "v10":
        kw.update(r=0.8, alpha=-0.01)
    per_env, drift = model == fo.MODEL_V4, model == fo.MODEL_V10
    sig = rng.uniform(0.02, 0.2, n) if sigarr else None
    zz = [rng.standard_normal(n) for _ in range(T)] if noise == "ext" else None
    if model == fo.MODEL_V0:
        ring = rng.integers(0, 100, (T, n)).astype(np.int32)
    else:
        ring = rng.uniform(-1.15, 0.3, (T, n)).astype(np.float32)
    fn = lib.fishing_step_f32 if dtype == np.float32 else lib.fishing_step_f64

    def run(mode):
        p = hh.params(model, general=(mode == "general"), derived=derived, origin=(c0, 0), **kw)
        st = hh.State(n, dtype, model, np.zeros(n), r=(np.full(n, kw.get("r", 0.3)) if (per_env and not derived) or drift else None),
                      K=np.full(n, 1.0) if per_env and not derived else None, sigma=sig, ep_return=ret, t_u8=t8)
        st.reset(p, seed=seed, counter=0, env_offset=off)
        if mode == "fused":
            st.step_fused(p, ring, T, seed=seed, step_counter=c0, env_offset=off, per_step=bool(trial & 1))
        else:
            dev_ring = st.ring_tensor(ring)
            for s in range(T):
                z = hh.dev(zz[s].astype(dtype)) if zz is not None else None
                assert fn(p, n, off, st.buffers(dev_ring[s], z), seed, c0 + s, None) == 0
            torch.cuda.synchronize()
        return st
    A, B = run("dispatch"), run("general")
    names = ["obs", "t", "reward", "done"] + (["ep_return"] if ret else []) + (["K", "r"] if per_env and not derived else []) + (["r"] if drift else [])
    what = (kind, np.dtype(dtype).name, n, auto, ret, sigarr, t8, noise)
    for name in names:
        assert _bits_equal(getattr(A, name), getattr(B, name)), (name, "dispatch vs general") + what
    if ret:
        ra, rb = A.record(), B.record()
        assert ra[2] == rb[2] and ra[3] == rb[3] and np.allclose(ra[:2], rb[:2], rtol=1e-12, equal_nan=True), what
    if noise != "ext":
        C = run("fused")
        for name in names:
            assert _bits_equal(getattr(A, name), getattr(C, name)), (name, "dispatch vs fused") + what
        if ret:
            rc = C.record()
            assert ra[2] == rc[2] and ra[3] == rc[3] and np.allclose(ra[:2], rc[:2], rtol=1e-12, equal_nan=True), what


@pytest.mark.parametrize("trial", range(10))
def test_randomised_requests_beyond_4096_tiles(hh, trial):
    """Round 3: a workgroup per tile at every size.  Ten random requests at N in (2^22, 1.5 * 2^23] -- 4097 .. 12288 tiles
    + a ragged tail: the exact one-tile forms, the catch-alls' (terminal observations, ballot words, sigma array),
    float64 on two envs per thread (up to ~6600 tiles) and on four -- three steps each (the XCD-aware zig-zag walks an
    odd and an even one) against the general kernel: every stream bit for bit, the return records to double rounding."""
    import torch
    from gym_fishing_amd import _capi
    lib = _capi.lib()
    rng = np.random.default_rng(7700 + trial)
    kind = ["v1", "v0", "v2", "v4d", "v9", "v1", "v4s", "v1", "v2", "v10"][trial]
    model = {"v0": fo.MODEL_V0, "v1": fo.MODEL_V1, "v2": fo.MODEL_V2, "v4s": fo.MODEL_V4, "v4d": fo.MODEL_V4, "v9": fo.MODEL_V9,
             "v10": fo.MODEL_V10}[kind]
    dtype = np.float64 if trial in (5, 7, 8) else np.float32
    n = int(rng.integers((1 << 22) + 1, 3 << 22))
    ret, sigarr = bool(rng.random() < 0.7), bool(rng.random() < 0.3)
    term, bits = bool(rng.random() < 0.4), bool(rng.random() < 0.3)
    derived = kind == "v4d"
    T, off, seed, c0 = 3, 4 * int(rng.integers(0, 50)), int(rng.integers(1, 1 << 40)), int(rng.integers(0, 300))
    kw = dict(sigma=0.12, C=0.5, Tmax=2, sigma_p=0.15, auto_reset=True)
    if kind in ("v0", "v1", "v2"):
        kw["K"] = float(rng.choice([1.0, 2.0, 1.5]))
        kw["x0"] = 0.75 * kw["K"]
    if kind == "v10":
        kw.update(r=0.8, alpha=-0.01)
    per_env, drift = model == fo.MODEL_V4, model == fo.MODEL_V10
    sig = rng.uniform(0.02, 0.2, n) if sigarr else None
    g = torch.Generator(device="cuda").manual_seed(trial)
    row = -(-n // 4) * 4            # (every action batch 16-byte aligned)
    ring = (torch.randint(0, 100, (T, row), device="cuda", generator=g, dtype=torch.int32) if model == fo.MODEL_V0
            else (torch.rand((T, row), device="cuda", generator=g) * 1.45 - 1.15).float())[:, :n]
    fn = lib.fishing_step_f32 if dtype == np.float32 else lib.fishing_step_f64
    outs, names_seen = [], []
    for general in (False, True):
        p = hh.params(model, general=general, derived=derived, origin=(c0, 0), **kw)
        st = hh.State(n, dtype, model, np.zeros(n), r=(np.full(n, kw.get("r", 0.3)) if (per_env and not derived) or drift else None),
                      K=np.full(n, 1.0) if per_env and not derived else None, sigma=sig, ep_return=ret, terminal=term, done_bits=bits)
        st.reset(p, seed=seed, counter=0, env_offset=off)
        names_seen.append(hh.kernel_name(p, n, st.buffers(ring[0]), dtype))
        for s in range(T):
            assert fn(p, n, off, st.buffers(ring[s]), seed, c0 + s, None) == 0
        torch.cuda.synchronize()
        outs.append(st)
    A, B = outs
    what = (kind, np.dtype(dtype).name, n, ret, sigarr, term, bits, names_seen)
    assert "step_kernel_lean" in names_seen[0] and "step_kernel<" in names_seen[1], what
    names = (["obs", "t", "reward", "done"] + (["ep_return"] if ret else []) + (["K", "r"] if per_env and not derived else [])
             + (["r"] if drift else []) + (["terminal"] if term else []) + (["done_bits"] if bits else []))
    for name in names:
        assert _bits_equal(getattr(A, name), getattr(B, name)), (name,) + what
    if ret:
        ra, rb = A.record(), B.record()
        assert ra[2] == rb[2] > 0 and ra[3] == rb[3] and np.allclose(ra[:2], rb[:2], rtol=1e-12, equal_nan=True), what
    del A, B, outs
    torch.cuda.empty_cache()


@pytest.mark.parametrize("where", ["step_counter_crosses_2^32", "env_index_crosses_2^32", "both_far_beyond_2^32"])
def test_v4_derived_parameters_across_the_32_bit_boundaries(hh, where):
    """The derivation does its integer work in 32 bits while every counter and env index of a tile fits, in 64 bits
    otherwise (fishing_common.h: derive_fits_32) -- the two must be the same function.  Stored vs derived parameters
    over 160 auto-resetting steps with the step counter running through 2^32, with the shard's env indices
    straddling 2^32 (some tiles narrow, some wide, in one launch), and with both far beyond."""
    import torch
    from gym_fishing_amd import _capi
    lib = _capi.lib()
    n, seed = 4 * 1024 + 40, 31337
    c0, off = {"step_counter_crosses_2^32": ((1 << 32) - 70, 16), "env_index_crosses_2^32": (5, (1 << 32) - 2048),
               "both_far_beyond_2^32": ((1 << 40) + 11, (1 << 36) + 4096)}[where]
    kw = dict(sigma=0.1, Tmax=6, K_mean=1.0, r_mean=0.3, sigma_p=0.2, auto_reset=True)
    ps = hh.params(fo.MODEL_V4, **kw)
    pd = hh.params(fo.MODEL_V4, derived=True, origin=(c0, 3), **kw)
    S = hh.State(n, np.float32, fo.MODEL_V4, np.zeros(n), r=np.full(n, 0.3), K=np.full(n, 1.0), ep_return=True)
    D = hh.State(n, np.float32, fo.MODEL_V4, np.zeros(n), ep_return=True)
    S.reset(ps, seed=seed, counter=3, env_offset=off)
    D.reset(pd, seed=seed, counter=3, env_offset=off)
    g = torch.Generator(device="cuda").manual_seed(2)
    for s in range(160):
        a = (torch.rand(n, device="cuda", generator=g) * 1.3 - 1.15).float()
        if s % 20 == 0:
            Kd, rd = D.v4_params(pd, seed=seed, step_counter=c0 + s, env_offset=off)
            same(Kd, S.K.cpu().numpy(), "K before step %d" % s)
            same(rd, S.r.cpu().numpy(), "r before step %d" % s)
        assert lib.fishing_step_f32(ps, n, off, S.buffers(a), seed, c0 + s, None) == 0
        assert lib.fishing_step_f32(pd, n, off, D.buffers(a), seed, c0 + s, None) == 0
        torch.cuda.synchronize()
        for name in ("obs", "reward", "done", "t", "ep_return"):
            assert _bits_equal(getattr(S, name), getattr(D, name)), (name, s, where)
    assert S.record()[2] == D.record()[2] > 10 * n
    # the oracle's restatement of the block (param_words) agrees with the device's on these indices too
    env = np.arange(off, off + 64, dtype=np.uint64)
    _, zK, zr = hh.device_noise(64, seed, c0 + 17, fo.STREAM_AUTORESET, off)
    eK, er = fo.reset_normals(seed, env, c0 + 17, fo.STREAM_AUTORESET)
    assert np.abs(zK - eK).max() < 2e-5 and np.abs(zr - er).max() < 2e-5


def test_v4_rollout_without_auto_reset_leaves_the_derived_mode(hh):
    """A fused rollout without auto-reset freezes finished envs (simulate_mdp's `break`): their year counters stop,
    so the rule that dates an episode from them no longer holds.  The C ABI refuses that combination; the host
    mirror stores the parameters first.  env.simulate() over a fishing-v4 batch then gives the same table from a
    derived-mode env and from a stored-mode env, and env.K stays what it was for the frozen envs."""
    import torch
    import gym_fishing_amd as gf
    from gym_fishing_amd import _capi, policies
    p = hh.params(fo.MODEL_V4, sigma=0.05, derived=True, auto_reset=False)
    st = hh.State(2048, np.float32, fo.MODEL_V4, np.zeros(2048))
    rc = _capi.lib().fishing_rollout_f32(p, 2048, 0, st.buffers(), _capi.POLICY_RANDOM, 0.0, 5, None, 0, 0, None)
    assert rc == -7                                      # FISHING_ERR_UNSUPPORTED
    tabs, Ks = [], []
    for derived in (None, False):
        env = gf.make("fishing-v4", num_envs=64, sigma=0.05, sigma_p=0.2, Tmax=12, seed=4, derived_params=derived)
        # (a policy the fused kernel runs: since round 4 escapement / msy on an N-env fishing-v4 batch carry one S per env and
        # are driven step by step -- tests/test_gpu_envs.py::test_v4_num_envs_bmsy_and_msy_follow_each_envs_parameters)
        model = ("constant", -0.85)
        df = env.simulate(model, reps=2)
        tabs.append(df.to_numpy(dtype=np.float64))
        Ks.append(env.K.clone())
        assert env._derived is False                 # (the no-auto-reset rollout stored the parameters)
        env.reset()
        assert env._derived is (derived is None)     # a full reset returns to the derived mode
    same(tabs[0], tabs[1], "simulate table: derived vs stored")
    assert torch.equal(Ks[0], Ks[1])


# ------------------------------------------------------------------ estimate_policyfn against the reference's table
def _policyfn_cases():
    z = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "reference_policyfn.npz"))
    return [(k, z[k]) for k in sorted(z.files)]


@pytest.mark.parametrize("key,table", _policyfn_cases(), ids=[k for k, _ in _policyfn_cases()])
def test_policyfn_reproduces_the_reference_table(hh, key, table):
    """env.policyfn(model, reps=2) (shared_env.py:82-102) on the reference with msy / escapement policies, for
    fishing-v0 / v1 with default and non-default parameters, against the same call here (scalar protocol, fp64
    kernels): 100 rows [population, quota, rep] bit for bit.  Both sides use a float64 observation grid (see
    tests/golden/make_golden.py for why)."""
    import gym_fishing_amd as gf
    from gym_fishing_amd import policies
    _, tag, pname = key.split("_", 2) if key.count("_") == 2 else (None, key.split("_")[1] + "_params", key.split("_")[3])
    env_id = "fishing-" + tag[:2]
    kw = {"v1_params": {"r": 0.5, "K": 2.0, "init_state": 1.1}, "v0_params": {"n_actions": 37, "r": 0.4}}.get(tag, {})
    env = gf.make(env_id, sigma=0.0, **kw)
    model = getattr(policies, pname)(env)
    env.observation_space.dtype = np.dtype(np.float64)
    df = env.policyfn(model, reps=2)
    got = df.to_numpy(dtype=np.float64) if hasattr(df, "to_numpy") else np.stack([df[c] for c in ("state", "action", "rep")], 1)
    same(got, table, key)


def test_v4_origin_stamps_through_the_c_abi(hh):
    """FishingBuffers.v4_stamp (ABI 6): a masked fishing_reset_* under FISHING_FLAG_V4_DERIVED stamps the masked envs with
    its reset counter + 1; from there the derived batch -- general kernel (ragged size), lean catch-all, fused kernel --
    equals a stored-array batch reset with the same mask bit for bit, fishing_v4_params_* shows the stored (K, r), an
    auto-reset clears an env's stamp, and a reset of every env clears them all.  Without the buffer the masked reset is
    refused (FISHING_ERR_UNSUPPORTED), as before."""
    import torch
    from gym_fishing_amd import _capi
    lib = _capi.lib()
    n, off, seed = 5 * 1024 + 12, 8, 41
    kw = dict(sigma=0.05, sigma_p=0.2, Tmax=5, auto_reset=True)
    pS = hh.params(fo.MODEL_V4, **kw)
    S = hh.State(n, np.float32, fo.MODEL_V4, np.zeros(n), r=np.zeros(n), K=np.ones(n), ep_return=True)
    D = hh.State(n, np.float32, fo.MODEL_V4, np.zeros(n), ep_return=True, stamp=np.zeros(n, np.int32))
    bare = hh.State(n, np.float32, fo.MODEL_V4, np.zeros(n))
    rng = np.random.default_rng(2)
    step_count, reset_count = 0, 3
    pD = hh.params(fo.MODEL_V4, derived=True, origin=(step_count, reset_count), **kw)
    S.reset(pS, seed=seed, counter=reset_count, env_offset=off)
    D.reset(pD, seed=seed, counter=reset_count, env_offset=off)
    reset_count += 1

    def same(tag):
        for name in ("obs", "t", "reward", "done", "ep_return"):
            a, b = getattr(S, name), getattr(D, name)
            assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b), (tag, name)
        K, r = D.v4_params(pD, seed=seed, step_counter=step_count, env_offset=off)
        assert np.array_equal(K, S.K.cpu().numpy()) and np.array_equal(r, S.r.cpu().numpy()), tag

    def steps(k, fused=False):
        nonlocal step_count
        acts = rng.uniform(-1.1, 0.2, (k, n)).astype(np.float32)
        if fused:
            S.step_fused(pS, acts, k, seed=seed, step_counter=step_count, env_offset=off, per_step=False)
            D.step_fused(pD, acts, k, seed=seed, step_counter=step_count, env_offset=off, per_step=False)
            step_count += k
        else:
            for i in range(k):
                S.step(pS, acts[i], seed=seed, step_counter=step_count, env_offset=off)
                D.step(pD, acts[i], seed=seed, step_counter=step_count, env_offset=off)
                step_count += 1
    steps(4)
    same("before the masked reset")
    mask = (rng.random(n) < 0.3).astype(np.uint8)
    bare.reset(pD, mask=mask, seed=seed, counter=reset_count, env_offset=off, expect=-7)     # no stamps to write
    S.reset(pS, mask=mask, seed=seed, counter=reset_count, env_offset=off)
    D.reset(pD, mask=mask, seed=seed, counter=reset_count, env_offset=off)
    assert np.array_equal(D.stamp.cpu().numpy(), np.where(mask, reset_count + 1, 0))
    reset_count += 1
    same("after the masked reset")
    steps(1)
    same("one step on")
    st = D.stamp.cpu().numpy()
    done = D.done.cpu().numpy().astype(bool)
    assert (st[done] == 0).all() and (st[~done & mask.astype(bool)] == reset_count).all()    # cleared exactly where auto-reset
    steps(5)
    same("per-step launches")
    mask2 = (rng.random(n) < 0.2).astype(np.uint8)
    S.reset(pS, mask=mask2, seed=seed, counter=reset_count, env_offset=off)
    D.reset(pD, mask=mask2, seed=seed, counter=reset_count, env_offset=off)
    reset_count += 1
    steps(6, fused=True)
    same("fused launch after a second masked reset")
    ra, rb = S.record(), D.record()
    assert ra[2] == rb[2] > 0 and ra[3] == rb[3]
    # a reset of every env: all stamps cleared, the origin words date the episodes again
    pD = hh.params(fo.MODEL_V4, derived=True, origin=(step_count, reset_count), **kw)
    S.reset(pS, seed=seed, counter=reset_count, env_offset=off)
    D.reset(pD, seed=seed, counter=reset_count, env_offset=off)
    assert int(D.stamp.abs().sum()) == 0
    steps(3)
    same("after the reset of every env")
