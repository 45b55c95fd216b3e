"""fishing-v4's per-episode (K, r) (fishing_model_error.py:37-48), all through the C ABI (tests/hip_harness.py) and the
env class: the derived-parameter mode (FISHING_FLAG_V4_DERIVED: no r / K arrays, every kernel re-derives an env's (K, r) from
the Philox2x32 block its year counter -- or its origin stamp -- points at) == the stored-array mode == the oracle, bit for
bit; the mode's exits and re-entries; random operation sequences three ways; the 32-bit boundaries of the counters."""
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

def _bits_equal(x, y):
    import torch
    it = {1: torch.uint8, 4: torch.int32, 8: torch.int64}[x.element_size()]
    return torch.equal(x.view(it), y.view(it))


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

    def same_bits(x, y):     # (NaN == NaN: a drawn K of 0 -- five standard deviations out, once in ~50 of these sequences -- makes the
        it = {1: torch.uint8, 4: torch.int32, 8: torch.int64}[x.element_size()]     # env's observations NaN in both batches)
        return x.shape == y.shape and torch.equal(x.contiguous().view(it), y.contiguous().view(it))

    def check(tag):
        torch.cuda.synchronize()
        for name in ("_obs", "_t", "_ep_return"):
            assert same_bits(getattr(D, name), getattr(S, name)), (trial, tag, name)
            assert same_bits(getattr(G, name), getattr(S, name)), (trial, tag, name, "graph")
        assert same_bits(D.K, S.K) and same_bits(D.r, S.r) and same_bits(G.K, S.K), (trial, tag)
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
            assert same_bits(D.K, S.K) and same_bits(D.r, S.r)
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
    assert sa["n_episodes"] == sb["n_episodes"] and np.array_equal(sa["sum_return"], sb["sum_return"], equal_nan=True)
    assert modes            # (which parameter modes the derived batch went through depends on the sequence; all three occur over the trials)

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
