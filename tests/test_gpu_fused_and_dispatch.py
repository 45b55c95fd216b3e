"""The launch forms of step(), all through the C ABI (tests/hip_harness.py): fishing_step_fused_* (K steps per launch, state
in registers) == K fishing_step_* calls, bit for bit; the episodic-return record counts an episode once however long a
finished env is stepped on; fishing_step_kernel_name_* names the instantiation the dispatch picks; FISHING_FLAG_PADDED_TILES;
randomised requests through dispatch / general kernel / fused launch; random operation sequences of every env family."""
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

def fo_mask(noise=2, ret=False, sigarr=False, t8=False, term=False, bits=False, zz=False, derived=False, drift=False, one=False):
    """Feature mask of step_kernel_lean (csrc/fishing_step.hip: namespace feat); `one` = a tile per workgroup (grid == tiles)."""
    return (noise | (4 if ret else 0) | (8 if sigarr else 0) | (16 if t8 else 0) | (32 if term else 0) | (64 if bits else 0)
            | (128 if zz else 0) | (256 if derived else 0) | (512 if drift else 0) | (8192 if one else 0))

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
    # (... and the other configs' real sizes: the metric's 2^22, config 3's fishing-v0 at 2^22, config 5's fishing-v4 shard at 2^21
    # on derived parameters with its sigma array)
    for model, n, t8 in ((fo.MODEL_V1, 1 << 20, False), (fo.MODEL_V2, 1 << 19, False), (fo.MODEL_V1, 1 << 18, True),
                         (fo.MODEL_V1, 1 << 22, False), (fo.MODEL_V0, 1 << 22, False), (fo.MODEL_V4, 1 << 21, False)):
        v4 = model == fo.MODEL_V4
        p = hh.params(model, sigma=0.05 if v4 else 0.1, C=0.5, auto_reset=True, t_u8=t8, derived=v4)
        g = torch.Generator(device="cuda").manual_seed(n)
        if model == fo.MODEL_V0:
            ring = torch.randint(0, 100, (8, n), device="cuda", generator=g, dtype=torch.int32)
        else:
            ring = (torch.rand((8, n), device="cuda", generator=g) * 2 - 1).float()
        mk = lambda: hh.State(n, np.float32, model, np.full(n, 0.75 if v4 else -0.25, np.float32), ep_return=True, t_u8=t8,   # noqa: E731
                              sigma=np.float32(0.05) if v4 else None)
        A, B = mk(), mk()
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
    assert hh.kernel_name(p11, n, full11, np.float64) == "fishing::step_kernel_lean<double, 105, 8198, 2>"     # (two envs per thread: round 5)


# ------------------------------------------------------------------ the host mirror in the derived mode

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
    kind = ["v1", "v0", "v2", "v4d", "v9", "v1", "v4s", "v1", "v2", "v10"][trial % 10]     # (tests/fuzz_differential.py runs trials beyond 9)
    model = {"v0": fo.MODEL_V0, "v1": fo.MODEL_V1, "v2": fo.MODEL_V2, "v4s": fo.MODEL_V4, "v4d": fo.MODEL_V4, "v9": fo.MODEL_V9,
             "v10": fo.MODEL_V10}[kind]
    dtype = np.float64 if trial % 10 in (5, 7, 8) else np.float32
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
