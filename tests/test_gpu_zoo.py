"""GPU parity of the growth-model zoo fishing-v5..v11 (growth_models.py).

These models go through log / exp / pow, which are not bit-reproducible between NumPy/libm
and the device math library, so parity is tolerance-based:
  * fp64 layout: on the POPULATION x = (obs+1) K, |dx| <= 2e-14 * x per step against the golden vectors
    captured from the reference (a few ulp of exp(mu), mu = O(1)); reward, done, t exact.  Round 5: the float64 kernels
    evaluate the algebraically equal form on a < 1-ulp exp (fishing_common.h: zoo_draw_f64); population_draw -- which hands
    populations out itself -- follows the reference's own round trip where ITS rounding exceeds this bar (stocks outside
    [2^-30, 2^30], results outside [2^-92, 2^92]), the step kernels -- whose obs = x / K - 1 cannot carry that difference
    (test_zoo_f64_step_outputs_do_not_depend_on_how_a_far_stock_is_evaluated) -- do not: same tolerance, measured maxima in
    profiles/r05_zoo_f64_error.json;
  * fp32 layout: the north star's bar -- |obs - ref| <= 1e-6 and |reward - ref| <= 1e-6 per step, absolute,
    against the reference's float64 numbers (round 4: the float32 kernels evaluate the growth function in the
    algebraically equal form without the log / exp round trip, fishing_common.h: zoo_draw_f32; measured maxima
    per growth function in profiles/r04_zoo_f32_error.json).
"""
import numpy as np
import pytest

from conftest import load_zoo_cases
from oracle import fishing_oracle as fo
from test_oracle_golden import ZOO_DEFAULTS

pytestmark = pytest.mark.gpu
ZOO = load_zoo_cases()
F64_RTOL = 2e-14
F32_ATOL = 1e-6         # on obs and on reward, absolute (BASELINE.json north_star)


@pytest.fixture(scope="module")
def hh():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device; none visible")
    import hip_harness
    return hip_harness


def zoo_kw(c):
    P = dict(ZOO_DEFAULTS[c.id])
    P.update(c.kwargs)
    return P


def hip_params(hh, c, **over):
    model = fo.MODEL_OF_ID[c.id]
    P = zoo_kw(c)
    kw = dict(r=float(P.get("r", 0.3)), K=float(P["K"]), sigma=float(P.get("sigma", 0.0)), C=float(P.get("C", 0.5)),
              x0=float(P["init_state"]), Tmax=int(P.get("Tmax", 100)), M=float(P.get("M", 0.0)),
              theta=float(P.get("theta", 0.0)), q=float(P.get("q", 0.0)), b=float(P.get("b", 0.0)),
              a=float(P.get("a", 0.0)), alpha=float(P.get("alpha", 0.0)))
    if model == fo.MODEL_V11:
        kw.update(models=[0, 1, 2, 3, 4], zoo_table=fo.V11_TABLE)
    kw.update(over)
    return hh.params(model, **kw)


def obs_close_f32(obs_dev, obs_ref):
    a, b = np.asarray(obs_dev, dtype=np.float64), np.asarray(obs_ref, dtype=np.float64)
    assert np.asarray(obs_dev).dtype == np.float32
    bad = ~((np.abs(a - b) <= F32_ATOL) | (np.isnan(a) & np.isnan(b)))
    assert not bad.any(), "max |obs - ref| %.3e at %s" % (np.nanmax(np.abs(a - b)), np.argwhere(bad)[0])


def pop_close(obs_dev, obs_ref, K, rtol):
    if np.asarray(obs_dev).dtype == np.float32:
        return obs_close_f32(obs_dev, obs_ref)
    a = (np.asarray(obs_dev, dtype=np.float64) + 1.0) * K
    b = (np.asarray(obs_ref, dtype=np.float64) + 1.0) * K
    # the state that is carried is obs = x/K - 1: near extinction (obs -> -1) its spacing, not the
    # transcendental error, bounds the population's accuracy -> 2 ulp of obs as absolute slack
    eps = 1.2e-7 if np.asarray(obs_dev).dtype == np.float32 else 2.3e-16
    bad = ~((np.abs(a - b) <= rtol * np.abs(b) + 2 * eps * K) | (np.isnan(a) & np.isnan(b)))
    assert not bad.any(), "max rel err %.3e at %s" % (np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)),
                                                       np.argwhere(bad)[0])


def t_in_of(c):
    t_in = np.where(np.arange(c.nsteps)[None, :] == 0, 0, np.roll(c.t, 1, axis=1))
    prev_done = np.roll(c.done, 1, axis=1).astype(bool)
    prev_done[:, 0] = False
    return np.where(prev_done, 0, t_in)


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("c", ZOO, ids=[c.name for c in ZOO])
def test_zoo_golden_single_steps(hh, c, dtype):
    """Every recorded reference step (all envs x steps as one batch), external noise."""
    model = fo.MODEL_OF_ID[c.id]
    K = float(zoo_kw(c)["K"])
    n = c.obs.size
    st = hh.State(n, dtype, model, c.obs_in.reshape(-1), t=t_in_of(c).reshape(-1),
                  r=c.params_r.reshape(-1) if model == fo.MODEL_V10 else None,
                  model_idx=c.model_idx.reshape(-1) if model == fo.MODEL_V11 else None)
    obs, rew, done, t = st.step(hip_params(hh, c), c.action.reshape(-1), z=c.z.reshape(-1))
    pop_close(obs, c.obs.reshape(-1), K, F64_RTOL)
    if dtype == np.float64:
        assert np.array_equal(rew, c.reward.reshape(-1), equal_nan=True)      # (v8_myers_r_below_minus_one: NaN stock, NaN harvest)
        assert (done == c.done.reshape(-1)).all()
    else:
        ref = c.reward.reshape(-1)
        assert (np.isnan(rew) == np.isnan(ref)).all() and (np.abs(rew - ref)[~np.isnan(ref)] <= F32_ATOL).all()
        # an f32 population within 1e-6 of zero may flip the extinction flag: none in the fixtures
        assert (done == c.done.reshape(-1)).all()
    assert (t == c.t.reshape(-1)).all()
    if model == fo.MODEL_V10:     # r drifted by alpha and was written back
        want = (c.params_r.reshape(-1) + zoo_kw(c)["alpha"]).astype(dtype)
        got = st.r.cpu().numpy()
        assert np.array_equal(got, want) if dtype == np.float64 else np.allclose(got, want, rtol=1e-6)


@pytest.mark.parametrize("env_id", ["fishing-v5", "fishing-v6", "fishing-v7", "fishing-v8", "fishing-v9", "fishing-v10",
                                    "fishing-v11"])
def test_zoo_scalar_protocol_follows_reference(env_id):
    """The Python classes (Allen ... ModelUncertainty) through the reference's scalar protocol,
    fed the reference's recorded normals; fishing-v11's model choice is installed from the
    fixture after every reset (the reference draws it from the global MT19937 stream)."""
    import torch
    import gym_fishing_amd as gf
    for c in [c for c in ZOO if c.id == env_id]:
        K = float(zoo_kw(c)["K"])
        for e in range(2):
            env = gf.make(env_id, **c.kwargs)
            obs = env.reset()
            assert obs.shape == (1,) and obs[0] == c.reset_obs[e, 0]
            for s in range(c.nsteps):
                if env_id == "fishing-v11":
                    env.model_idx.fill_(int(c.model_idx[e, s]))
                if env_id == "fishing-v10":
                    assert abs(env.r - c.params_r[e, s]) < 1e-12
                obs, rew, done, info = env.step(np.array([c.action[e, s]], dtype=np.float32), noise=[c.z[e, s]])
                if np.isnan(c.obs[e, s]):         # (v8_myers_r_below_minus_one: the reference's stock is NaN, and so must this one's be)
                    assert np.isnan(obs[0]), (name, e, s)
                else:
                    assert abs((obs[0] + 1) * K - (c.obs[e, s] + 1) * K) <= 1e-12 * max(1.0, abs(c.obs[e, s] + 1) * K)
                assert (np.isnan(rew) and np.isnan(c.reward[e, s])) or abs(rew - c.reward[e, s]) <= 1e-12, (name, e, s)
                assert done == bool(c.done[e, s])
                # keep following the reference exactly so ulp-level differences cannot accumulate
                env._obs.fill_(float(c.obs[e, s]))
                if done:
                    obs = env.reset()
                    assert obs[0] == c.reset_obs[e, s + 1]
            if env_id == "fishing-v11":
                assert env.model in ("allen", "beverton_holt", "myers", "may", "ricker")


def test_v11_model_draw_is_the_oracles_and_uniform_to_two_to_the_minus_sixteen(hh):
    """growth_models.py:187,200: np.random.choice(models) per episode.  The device draws it from one Philox2x32-10 block per
    env quad, 16 bits per env (fishing_common.h: redraw_kinds): the reset kernel's and the step kernel's draws are the
    oracle's bit for bit -- any env offset, both reset streams, model lists of 1 .. 5 -- and over 2^20 envs x 3 draws the
    model frequencies pass a chi-square test at 1 / n each (the scheme's own bias is <= 2^-16 per model: invisible here)."""
    n, seed = 1 << 20, 4242
    p = hh.params(fo.MODEL_V11, sigma=0.0, Tmax=0, auto_reset=True, models=[0, 1, 2, 3, 4], zoo_table=[dict(d) for d in fo.V11_TABLE])
    counts = np.zeros(5)
    for off, counter in ((0, 0), (12, 5), (1 << 20, 1 << 20)):
        st = hh.State(n, np.float32, fo.MODEL_V11, np.zeros(n), model_idx=np.zeros(n, np.int32))
        st.reset(p, seed=seed, counter=counter, env_offset=off)
        got = st.model_idx.cpu().numpy()
        env = np.arange(off, off + n, dtype=np.uint64)
        assert np.array_equal(got, fo.model_draw(seed, env, counter, fo.STREAM_RESET, [0, 1, 2, 3, 4]))
        counts += np.bincount(got, minlength=5)
        # Tmax = 0: every env finishes on its first step -> every env redraws on the auto-reset stream, keyed by the step counter
        st.step(p, np.full(n, -1.0, np.float32), seed=seed, step_counter=counter + 7, env_offset=off)
        again = st.model_idx.cpu().numpy()
        assert np.array_equal(again, fo.model_draw(seed, env, counter + 7, fo.STREAM_AUTORESET, [0, 1, 2, 3, 4]))
        assert (again != got).mean() > 0.7          # (another block: 4 in 5 envs change their model)
    expect = counts.sum() / 5
    chi2 = ((counts - expect) ** 2 / expect).sum()
    assert chi2 < 18.5, (chi2, counts)              # 4 degrees of freedom: P(chi2 > 18.5) = 1e-3
    for models in ([2], [4, 1], [3, 3, 0], [4, 0, 3, 1]):     # shorter lists, any order, repeats allowed
        pm = hh.params(fo.MODEL_V11, sigma=0.0, Tmax=5, models=models, zoo_table=[dict(d) for d in fo.V11_TABLE])
        st = hh.State(4099, np.float64, fo.MODEL_V11, np.zeros(4099), model_idx=np.zeros(4099, np.int32))
        st.reset(pm, seed=9, counter=3, env_offset=8)
        assert np.array_equal(st.model_idx.cpu().numpy(),
                              fo.model_draw(9, np.arange(8, 8 + 4099, dtype=np.uint64), 3, fo.STREAM_RESET, models))


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("model", [fo.MODEL_V5, fo.MODEL_V7, fo.MODEL_V8, fo.MODEL_V10, fo.MODEL_V11])
def test_zoo_philox_auto_reset_vs_oracle(hh, model, dtype):
    """In-kernel noise + fused auto-reset at N = 4099: the oracle is fed the device's normals
    and (fishing-v11) the device's model draws; populations agree within the transcendental
    tolerance, the drift / model bookkeeping exactly."""
    _zoo_philox_auto_reset_vs_oracle(hh, model, dtype, n=4099, off=8, T=24)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("model", [fo.MODEL_V5, fo.MODEL_V6, fo.MODEL_V7, fo.MODEL_V8, fo.MODEL_V9, fo.MODEL_V10, fo.MODEL_V11],
                         ids=["v5", "v6", "v7", "v8", "v9", "v10", "v11"])
def test_zoo_full_size_batch_vs_oracle(hh, model, dtype):
    """The same comparison over EVERY env of a BASELINE-sized batch, N = 2^22 -- every id of the zoo in both layouts
    (fishing-v10: the drifting r stream; fishing-v11: growth function per env from the LDS table, model redraws on the
    auto-reset stream): three steps, each output of each env against the
    oracle fed the device's normals -- the zig-zag walk, the tile boundaries and the quad-indexed generators are
    index-dependent code that a window in the middle of the batch does not reach.  Tolerances unchanged (float32: 1e-6 on
    obs and reward; float64: 2e-14 of the population); the oracle's exp is NumPy's here (1 ulp from libm's: see simd_exp)."""
    _zoo_philox_auto_reset_vs_oracle(hh, model, dtype, n=1 << 22, off=1 << 20, T=3, simd_exp=True)


def _zoo_philox_auto_reset_vs_oracle(hh, model, dtype, n, off, T, simd_exp=False):
    seed = 777
    env_id = {v: k for k, v in fo.MODEL_OF_ID.items()}[model]
    P = dict(ZOO_DEFAULTS[env_id])
    P.update(sigma=0.1)
    K, x0, Tmax = float(P["K"]), float(P["init_state"]), 7
    kw = dict(r=float(P.get("r", 0.3)), K=K, sigma=0.1, C=float(P.get("C", 0.5)), x0=x0, Tmax=Tmax,
              M=float(P.get("M", 0.0)), theta=float(P.get("theta", 0.0)), q=float(P.get("q", 0.0)),
              b=float(P.get("b", 0.0)), a=float(P.get("a", 0.0)), alpha=float(P.get("alpha", 0.0)), auto_reset=True)
    table = [dict(d, sigma=0.1) for d in fo.V11_TABLE]
    if model == fo.MODEL_V11:
        kw.update(models=[4, 0, 3], zoo_table=table)          # a 3-model list in a non-default order
    p = hh.params(model, **kw)
    st = hh.State(n, dtype, model, np.zeros(n), r=np.full(n, P.get("r", 0.3)) if model == fo.MODEL_V10 else None,
                  model_idx=np.zeros(n, np.int32) if model == fo.MODEL_V11 else None, terminal=True)
    st.reset(p, seed=seed, counter=0, env_offset=off)
    rng = np.random.default_rng(3)
    obs = np.full(n, x0 / K - 1.0, dtype)
    assert np.array_equal(st.obs.cpu().numpy(), obs)
    t = np.zeros(n, np.int32)
    r_arr = np.full(n, P.get("r", 0.3), dtype)
    kind = st.model_idx.cpu().numpy() if model == fo.MODEL_V11 else None
    if model == fo.MODEL_V11:
        want = fo.model_draw(seed, np.arange(off, off + n, dtype=np.uint64), 0, fo.STREAM_RESET, [4, 0, 3])
        # (one Philox2x32-10 block per env quad, four 16-bit draws: oracle.model_words mirrors fishing_common.h: model_block)
        assert np.array_equal(kind, want) and set(np.unique(kind)) == {0, 3, 4}
    rtol = F64_RTOL
    for s in range(T):
        a = rng.uniform(-1, -0.6, n).astype(np.float32)
        o, rew, done, t2 = st.step(p, a, seed=seed, step_counter=s, env_offset=off)
        z = hh.device_step_noise(n, seed, s, off).astype(np.float64)
        Pstep = dict(P)
        if model == fo.MODEL_V10:
            r_arr = (r_arr + dtype(P["alpha"])).astype(dtype)
            Pstep["r"] = r_arr.astype(np.float64)
            got_r = st.r.cpu().numpy()
            assert np.array_equal(got_r, r_arr)
        eo, er, ed, et, ex = fo.step_zoo(model, obs.astype(np.float64), t, a, z, table if model == fo.MODEL_V11 else Pstep,
                                         K, Tmax=Tmax, kind=kind, simd_exp=simd_exp)
        term = st.terminal.cpu().numpy()
        pop_close(term, eo, K, rtol)
        assert np.abs(rew.astype(np.float64) - er).max() <= (0 if dtype == np.float64 else F32_ATOL)
        # extinction flag: x <= 0 is decided on the population before it is folded into obs; only a
        # population within rounding distance of zero may be classified differently
        differ = done != ed
        assert (ex[differ] < 1e-6).all()
        # follow the device state (errors must not compound in the comparison)
        ed = done
        assert (t2 == np.where(ed.astype(bool), 0, et)).all()
        m = ed.astype(bool)
        obs = np.where(m, dtype(x0 / K - 1.0), term).astype(dtype)
        assert np.array_equal(o, obs)
        t = np.where(m, 0, et).astype(np.int32)
        if model == fo.MODEL_V11 and m.any():
            draw = fo.model_draw(seed, np.arange(off, off + n, dtype=np.uint64), s, fo.STREAM_AUTORESET, [4, 0, 3])
            kind = np.where(m, draw, kind).astype(np.int32)
            assert np.array_equal(st.model_idx.cpu().numpy(), kind)


def test_zoo_bmsy_and_policies_run(hh):
    """msy / escapement + simulate on zoo envs (the reference's own tests do exactly this:
    tests/test-envs.py:33-139), via population_draw on the device and the step-by-step path."""
    import gym_fishing_amd as gf
    from gym_fishing_amd import policies
    for env_id in ("fishing-v5", "fishing-v6", "fishing-v7", "fishing-v8", "fishing-v9"):
        env = gf.make(env_id, sigma=0)
        S = policies.BMSY(env)
        P = ZOO_DEFAULTS[env_id]
        grid = (np.linspace(-1, 1, 10001, dtype=np.float32).astype(np.float64) + 1) * float(P["K"])
        Pq = dict(P, sigma=0.0)
        g = fo.zoo_population_draw(fo.KIND_OF_MODEL[fo.MODEL_OF_ID[env_id]], grid, np.zeros_like(grid), Pq) - grid
        assert abs(S - grid[np.nanargmax(g)]) <= 3 * 2e-4 * float(P["K"])
        df = env.simulate(policies.escapement(env))
        assert len(df) == 100 and df["reward"].sum() > 0
        df2 = env.simulate(policies.msy(env))
        assert 1 <= len(df2) <= 100


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("policy", ["random", "constant", "escapement", "msy"])
@pytest.mark.parametrize("env_id", ["fishing-v5", "fishing-v6", "fishing-v7", "fishing-v8", "fishing-v9", "fishing-v10",
                                    "fishing-v11"])
def test_zoo_fused_rollout_equals_stepwise(hh, env_id, policy, dtype):
    """The fused rollout of a zoo env == T step() calls fed the policy's float32 actions:
    both run the same device arithmetic and Philox blocks, so bit-for-bit (incl. the
    fishing-v10 drift of r and the fishing-v11 model redraws)."""
    from gym_fishing_amd import _capi
    model = fo.MODEL_OF_ID[env_id]
    n, off, seed, T = 2052, 4, 31, 22
    P = dict(ZOO_DEFAULTS[env_id])
    K, x0 = float(P["K"]), float(P["init_state"])
    kw = dict(r=float(P.get("r", 0.3)), K=K, sigma=0.1, C=float(P.get("C", 0.5)), x0=x0, Tmax=6,
              M=float(P.get("M", 0.0)), theta=float(P.get("theta", 0.0)), q=float(P.get("q", 0.0)),
              b=float(P.get("b", 0.0)), a=float(P.get("a", 0.0)), alpha=float(P.get("alpha", 0.0)), auto_reset=True)
    if model == fo.MODEL_V11:
        kw.update(models=[0, 1, 2, 3, 4], zoo_table=[dict(d, sigma=0.1) for d in fo.V11_TABLE])
    p = hh.params(model, **kw)
    pol = {"random": _capi.POLICY_RANDOM, "constant": _capi.POLICY_CONSTANT, "escapement": _capi.POLICY_ESCAPEMENT,
           "msy": _capi.POLICY_MSY}[policy]
    param = {"random": 0.0, "constant": -0.8125, "escapement": 0.5 * K, "msy": 0.05 * K}[policy]
    mk = lambda: hh.State(n, dtype, model, np.zeros(n), r=np.full(n, P.get("r", 0.3)) if model == fo.MODEL_V10 else None,   # noqa: E731
                          model_idx=np.zeros(n, np.int32) if model == fo.MODEL_V11 else None, ep_return=True)
    A, B = mk(), mk()
    A.reset(p, seed=seed, env_offset=off)
    B.reset(p, seed=seed, env_offset=off)
    traj = A.rollout(p, pol, param, T, seed=seed, env_offset=off, record=True)
    env = np.arange(off, off + n)
    for s in range(T):
        obs = B.obs.cpu().numpy()
        if policy == "random":
            a = fo.policy_random_action(fo.MODEL_V1, seed, env, s)
        elif policy == "constant":
            a = np.full(n, param, np.float32)
        else:
            a = fo.policy_action(policy, param, fo.MODEL_V1, obs, dtype(K), 100, dtype)
        assert np.array_equal(traj[s, 0], obs), (s, "obs_in")
        assert np.array_equal(traj[s, 1], a.astype(dtype)), (s, "action")
        _, rew, done, _ = B.step(p, a, seed=seed, step_counter=s, env_offset=off)
        assert np.array_equal(traj[s, 2], rew) and (traj[s, 3].astype(np.uint8) == done).all(), s
    assert np.array_equal(A.obs.cpu().numpy(), B.obs.cpu().numpy())
    assert (A.t.cpu().numpy() == B.t.cpu().numpy()).all()
    if model == fo.MODEL_V10:
        assert np.array_equal(A.r.cpu().numpy(), B.r.cpu().numpy())
    if model == fo.MODEL_V11:
        assert np.array_equal(A.model_idx.cpu().numpy(), B.model_idx.cpu().numpy())
    ra, rb = A.record(), B.record()
    assert ra[2] == rb[2] and ra[2] >= n and np.allclose(ra, rb, rtol=1e-12)


def test_zoo_simulate_uses_the_fused_rollout(hh):
    import torch
    import gym_fishing_amd as gf
    from gym_fishing_amd import policies
    venv = gf.make("fishing-v9", sigma=0.0, num_envs=8, dtype=torch.float64)
    esc = policies.escapement(venv)
    fused = venv.simulate(esc).to_numpy(dtype=np.float64)

    class Plain:      # no kernel_policy -> step-by-step path
        def predict(self, obs, **kw):
            return esc.predict(obs, **kw)
    stepwise = venv.simulate(Plain()).to_numpy(dtype=np.float64)
    assert fused.shape == stepwise.shape == (800, 5) and np.array_equal(fused, stepwise)


@pytest.mark.parametrize("env_id", ["fishing-v6", "fishing-v10", "fishing-v11"])
def test_zoo_env_step_many_fused_equals_launch_per_step(hh, env_id):
    """env.step_many(fused=True) for the zoo, fishing-v11 (growth function per env, redrawn at every auto-reset inside
    the launch) included: every per-step reward / done row and the final state equal a launch per step."""
    import torch
    import gym_fishing_amd as gf
    n, K = 3 * 1024 + 5, 37
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    buf = torch.rand((6, n + 7), device="cuda", generator=g) * 1.2 - 1.1      # row stride: a multiple of 4 elements
    acts = buf[:, :n]

    def mk():
        kw = {} if env_id == "fishing-v11" else dict(sigma=0.1)
        env = gf.make(env_id, num_envs=n, seed=11, Tmax=9, track_returns=True, **kw)
        if env_id == "fishing-v11":
            for d in env.model_params.values():
                d["sigma"] = 0.1
        env.reset()
        return env
    A, B = mk(), mk()
    rows_r, rows_d = [], []
    for k in range(K):
        _, r, d, _ = A.step(acts[k % 6])
        rows_r.append(r.clone())
        rows_d.append(d.clone())
    stride = (n + 15) // 16 * 16
    rr = torch.empty((K, stride), device="cuda")[:, :n]
    dd = torch.empty((K, stride), dtype=torch.uint8, device="cuda")[:, :n]
    B.step_many(acts, K, fused=True, rewards_out=rr, dones_out=dd)
    for k in range(K):
        assert torch.equal(rr[k].view(torch.int32), rows_r[k].view(torch.int32)), (env_id, k)
        assert torch.equal(dd[k].bool(), rows_d[k].bool()), (env_id, k)
    for key in ("_obs", "_t"):
        assert torch.equal(getattr(A, key).view(torch.int32), getattr(B, key).view(torch.int32)), key
    if env_id == "fishing-v11":
        assert torch.equal(A._model_idx, B._model_idx)
        assert len(set(A._model_idx.cpu().tolist())) == 5
    ea, eb = A.episode_stats(), B.episode_stats()
    assert ea["n_episodes"] == eb["n_episodes"] > n


@pytest.mark.parametrize("model", [fo.MODEL_V5, fo.MODEL_V6, fo.MODEL_V7, fo.MODEL_V8, fo.MODEL_V9, fo.MODEL_V10])
@pytest.mark.parametrize("ret", [False, True], ids=["plain", "returns"])
def test_zoo_lean_and_general_kernels_agree(hh, model, ret):
    """fishing-v5..v10 in float32 take the lean step kernel for whole 1024-env tiles (one growth
    function; v10 adds its drifting per-env r stream); FISHING_FLAG_GENERAL_KERNEL forces the general
    kernel.  Same bits on every stream over 12 auto-resetting steps, sigma > 0 and sigma = 0, ragged
    tail included."""
    import torch
    env_id = {v: k for k, v in fo.MODEL_OF_ID.items()}[model]
    P = dict(ZOO_DEFAULTS[env_id])
    n = 1024 * 5 + 7
    lib = __import__("gym_fishing_amd")._capi.lib()
    for sigma in (0.1, 0.0):
        kw = dict(r=float(P.get("r", 0.3)), K=float(P["K"]), sigma=sigma, C=float(P.get("C", 0.5)),
                  x0=float(P["init_state"]), Tmax=4, M=float(P.get("M", 0.0)), theta=float(P.get("theta", 0.0)),
                  q=float(P.get("q", 0.0)), b=float(P.get("b", 0.0)), a=float(P.get("a", 0.0)),
                  alpha=float(P.get("alpha", 0.0)), auto_reset=True)
        pa, pb = hh.params(model, **kw), hh.params(model, general=True, **kw)
        drift = model == fo.MODEL_V10
        A, B = (hh.State(n, np.float32, model, np.zeros(n), ep_return=ret,
                         r=np.full(n, kw["r"]) if drift else None) for _ in range(2))
        A.reset(pa, seed=5, env_offset=12)
        B.reset(pb, seed=5, env_offset=12)
        g = torch.Generator(device="cuda").manual_seed(n)
        for s in range(12):
            a = (torch.rand(n, device="cuda", generator=g) * 1.4 - 1.2).float()
            for st, p in ((A, pa), (B, pb)):
                assert lib.fishing_step_f32(p, n, 12, st.buffers(a), 5, s, None) == 0
            torch.cuda.synchronize()
            for name in ("obs", "reward", "done", "t") + (("ep_return",) if ret else ()) + (("r",) if drift else ()):
                assert torch.equal(getattr(A, name), getattr(B, name)), (name, s, sigma)
        if drift:
            assert np.allclose(A.r.cpu().numpy(), kw["r"] + 12 * kw["alpha"], rtol=1e-5)
        assert int(A.done.sum()) >= 0 and bool(torch.isfinite(A.obs).all())
        if ret:
            ra, rb = A.record(), B.record()
            assert ra[2] == rb[2] and ra[3] == rb[3] and np.allclose(ra[:2], rb[:2], rtol=1e-12) and ra[2] > 0


def test_zoo_f64_step_outputs_do_not_depend_on_how_a_far_stock_is_evaluated(hh):
    """The float64 growth functions hand far stocks / far results over to the reference's own log / exp round trip only in
    population_draw; the step kernels evaluate the algebraic form everywhere (fishing_step.hip, top).  Why that is no loss of
    parity: a step's state leaves as obs = x' / K - 1, and for stocks of 2^-31 ... 2^-50 K -- far below the hand-over line -- the
    float64 oracle (the reference's round trip) and the kernel produce the SAME obs bits, the same reward and done, for every
    growth function: the difference between the two evaluations (< 1e-13 of x') is far below what obs can hold (1.1e-16 absolute)."""
    n = 4096
    rng = np.random.default_rng(11)
    expo = rng.integers(31, 51, n)
    obs = (np.ldexp(1.0, -expo) - 1.0).astype(np.float64)           # x = (obs + 1) K = 2^-e exactly (K = 1)
    assert ((obs + 1.0) == np.ldexp(1.0, -expo)).all()
    t = rng.integers(0, 50, n).astype(np.int32)
    a = np.full(n, -1.0, np.float32)                                # quota 0: the stock after harvest is x itself
    z = rng.standard_normal(n)
    table = [dict(d, sigma=0.1, K=1.0) for d in fo.V11_TABLE]
    model_of_kind = {fo.KIND_ALLEN: fo.MODEL_V5, fo.KIND_BH: fo.MODEL_V6, fo.KIND_MYERS: fo.MODEL_V8, fo.KIND_MAY: fo.MODEL_V7,
                     fo.KIND_RICKER: fo.MODEL_V9}
    for kind, model in model_of_kind.items():
        P = table[kind]
        pk = hh.params(model, r=float(P["r"]), K=1.0, sigma=0.1, C=float(P.get("C", 0.5)), M=float(P.get("M", 0.0)),
                       theta=float(P.get("theta", 0.0)), q=float(P.get("q", 0.0)), b=float(P.get("b", 0.0)), a=float(P.get("a", 0.0)),
                       x0=0.75, Tmax=100)
        st = hh.State(n, np.float64, model, obs, t=t)
        o, rew, done, t2 = st.step(pk, a, z=z)
        eo, er, ed, et, ex = fo.step_zoo(model, obs, t, a, z, dict(P, init_state=0.75), 1.0, Tmax=100)
        assert (ex > 0).all() and (ex < 2.0 ** -25).all()           # tiny, live stocks -- the regime the hand-over exists for
        assert np.array_equal(o.view(np.uint64), np.asarray(eo, np.float64).view(np.uint64)), kind
        assert np.array_equal(rew, er) and np.array_equal(done, ed) and np.array_equal(t2, et), kind
    kinds = rng.integers(0, 5, n).astype(np.int32)
    p11 = hh.params(fo.MODEL_V11, sigma=0.0, K=1.0, x0=0.75, Tmax=100, models=[0, 1, 2, 3, 4], zoo_table=table)
    st = hh.State(n, np.float64, fo.MODEL_V11, obs, t=t, model_idx=kinds)
    o, rew, done, t2 = st.step(p11, a, z=z)
    eo, er, ed, et, ex = fo.step_zoo(fo.MODEL_V11, obs, t, a, z, table, 1.0, Tmax=100, kind=kinds)
    assert np.array_equal(o.view(np.uint64), np.asarray(eo, np.float64).view(np.uint64))
    assert np.array_equal(rew, er) and np.array_equal(done, ed)


def test_zoo_f64_fused_and_rollout_kernels_on_tiny_stocks(hh):
    """The same argument for the other two translation-unit-local copies of the float64 growth functions (fishing_rollout.hip
    is built with FISHING_ZOO_F64_FAR 0 like fishing_step.hip): the fused K-step kernel and the in-kernel-policy rollout kernel,
    one step each from stocks of 2^-31 ... 2^-53 K -- everything obs can hold below the hand-over line; a stock of 1e-30 K has no
    obs of its own, obs = -1 is the extinct stock -- with in-kernel noise, against the float64 oracle (the reference's round
    trip) fed the device's normals: the same obs bits, reward and done for every growth function and for fishing-v11."""
    from gym_fishing_amd import _capi
    n, seed = 4096, 31
    rng = np.random.default_rng(12)
    expo = rng.integers(31, 54, n)
    obs = (np.ldexp(1.0, -expo) - 1.0).astype(np.float64)
    assert ((obs + 1.0) == np.ldexp(1.0, -expo)).all()
    t = rng.integers(0, 50, n).astype(np.int32)
    a = np.full(n, -1.0, np.float32)                                # quota 0
    z = hh.device_step_noise(n, seed, 5, 0).astype(np.float64)
    table = [dict(d, sigma=0.1, K=1.0) for d in fo.V11_TABLE]
    kinds = rng.integers(0, 5, n).astype(np.int32)
    cases = [(fo.MODEL_V5, fo.KIND_ALLEN), (fo.MODEL_V6, fo.KIND_BH), (fo.MODEL_V8, fo.KIND_MYERS), (fo.MODEL_V7, fo.KIND_MAY),
             (fo.MODEL_V9, fo.KIND_RICKER), (fo.MODEL_V11, None)]
    for model, kind in cases:
        if model == fo.MODEL_V11:
            pk = hh.params(model, sigma=0.0, K=1.0, x0=0.75, Tmax=100, models=[0, 1, 2, 3, 4], zoo_table=table)
            eo, er, ed, et, ex = fo.step_zoo(model, obs, t, a, z, table, 1.0, Tmax=100, kind=kinds)
        else:
            P = table[kind]
            pk = hh.params(model, r=float(P["r"]), K=1.0, sigma=0.1, C=float(P.get("C", 0.5)), M=float(P.get("M", 0.0)),
                           theta=float(P.get("theta", 0.0)), q=float(P.get("q", 0.0)), b=float(P.get("b", 0.0)),
                           a=float(P.get("a", 0.0)), x0=0.75, Tmax=100)
            eo, er, ed, et, ex = fo.step_zoo(model, obs, t, a, z, dict(P, init_state=0.75), 1.0, Tmax=100)
        assert (ex > 0).all() and (ex < 2.0 ** -25).all()
        mk = lambda: hh.State(n, np.float64, model, obs, t=t, model_idx=kinds if model == fo.MODEL_V11 else None)      # noqa: E731
        st = mk()
        rs, ds = st.step_fused(pk, a[None, :], 1, seed=seed, step_counter=5)
        o, _, _, t2 = st.host()
        assert np.array_equal(o.view(np.uint64), np.asarray(eo, np.float64).view(np.uint64)), ("fused", model)
        assert np.array_equal(rs[0], er) and np.array_equal(ds[0], ed) and np.array_equal(t2, et), ("fused", model)
        st = mk()
        st.rollout(pk, _capi.POLICY_CONSTANT, -1.0, 1, seed=seed, step_counter=5)
        o, rew, done, t2 = st.host()
        assert np.array_equal(o.view(np.uint64), np.asarray(eo, np.float64).view(np.uint64)), ("rollout", model)
        assert np.array_equal(rew, er) and np.array_equal(done, ed) and np.array_equal(t2, et), ("rollout", model)


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("powers", [(3.0, 3.0), (2.0, 4.0), (2.5, 3.0), (3.0, 1.7), (2.5, 1.7)],
                         ids=["theta3_q3", "theta2_q4", "theta2p5_q3", "theta3_q1p7", "theta2p5_q1p7"])
def test_v11_coefficient_table_equals_the_single_kind_kernels_bit_for_bit(hh, dtype, powers):
    """fishing-v11's step evaluates its env's growth function from a per-kind coefficient table (fishing_common.h:
    zoo_draw_lut_one): one division and one exp per env, the zero coefficients of the other kinds contributing exact zeros.  Held
    here to the kernels that carry ONE growth function as a compile-time fact (fishing-v5 / v6 / v8 / v7 / v9 with the same
    parameter set): all envs of a batch assigned kind k must come out with the same bits as that model's kernel on the same
    states, actions and noise -- obs, reward, done -- for every kind, with Myers' theta and May's q equal small integers (one shared
    product), different small integers, and non-integers (exp2(e log2 x) / exp(e log x) per lane); and against the oracle within
    the layout's tolerance."""
    theta, q = powers
    n = 4096 + 8
    rng = np.random.default_rng(int(10 * theta + 100 * q))
    table = [dict(d, sigma=0.07 + 0.01 * k) for k, d in enumerate(fo.V11_TABLE)]
    table[fo.KIND_MYERS]["theta"] = theta
    table[fo.KIND_MAY]["q"] = q
    table[fo.KIND_MAY]["K"] = 1.0         # (May's growth reads M, not K; K = 1 gives its own kernel fishing-v11's obs / quota maps)
    obs = rng.uniform(-1.0, 0.9, n).astype(dtype)
    obs[::61] = -1.0
    t = rng.integers(0, 90, n).astype(np.int32)
    a = rng.uniform(-1.1, 0.3, n).astype(np.float32)
    z = rng.standard_normal(n).astype(dtype)
    model_of_kind = {fo.KIND_ALLEN: fo.MODEL_V5, fo.KIND_BH: fo.MODEL_V6, fo.KIND_MYERS: fo.MODEL_V8, fo.KIND_MAY: fo.MODEL_V7,
                     fo.KIND_RICKER: fo.MODEL_V9}
    p11 = hh.params(fo.MODEL_V11, sigma=0.0, K=1.0, x0=0.75, Tmax=100, models=[0, 1, 2, 3, 4], zoo_table=table)
    for kind, model in model_of_kind.items():
        P = table[kind]
        one = hh.State(n, dtype, model, obs, t=t)
        assert float(P["K"]) == 1.0     # (the single-kind env's obs / quota maps use ITS K, fishing-v11's the env core's K = 1)
        pk = hh.params(model, r=float(P["r"]), K=1.0, sigma=float(P["sigma"]), C=float(P.get("C", 0.5)), M=float(P.get("M", 0.0)),
                       theta=float(P.get("theta", 0.0)), q=float(P.get("q", 0.0)), b=float(P.get("b", 0.0)), a=float(P.get("a", 0.0)),
                       x0=0.75, Tmax=100)
        o1, r1, d1, t1 = one.step(pk, a, z=z)
        mixed = hh.State(n, dtype, fo.MODEL_V11, obs, t=t, model_idx=np.full(n, kind, np.int32))
        o2, r2, d2, t2 = mixed.step(p11, a, z=z)
        it = {4: np.uint32, 8: np.uint64}[np.dtype(dtype).itemsize]
        assert np.array_equal(o1.view(it), o2.view(it)), ("obs", kind, powers)
        assert np.array_equal(r1.view(it), r2.view(it)) and np.array_equal(d1, d2) and np.array_equal(t1, t2), (kind, powers)
    # every kind at once, against the oracle (float64 arithmetic of the reference's round trip)
    kinds = rng.integers(0, 5, n).astype(np.int32)
    mixed = hh.State(n, dtype, fo.MODEL_V11, obs, t=t, model_idx=kinds)
    o, rew, done, t2 = mixed.step(p11, a, z=z)
    eo, er, ed, et, ex = fo.step_zoo(fo.MODEL_V11, obs.astype(np.float64), t, a, z.astype(np.float64), table, 1.0, Tmax=100, kind=kinds)
    pop_close(o, eo, 1.0, F64_RTOL)
    assert np.abs(rew.astype(np.float64) - er).max() <= (0 if dtype == np.float64 else F32_ATOL)
    assert (ex[done != ed] < 1e-6).all() and (t2 == et).all()


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("model", [fo.MODEL_V5, fo.MODEL_V6, fo.MODEL_V7, fo.MODEL_V8, fo.MODEL_V9])
def test_zoo_random_parameter_sets_match_oracle(hh, model, dtype, seed_offset=0):
    """The zoo's counterpart of test_gpu_parity.py::test_random_parameter_sets_match_oracle: 24 random parameter sets per growth
    function -- r (Ricker: negative too), K, sigma, Allen's C, May's M / q / b / a, Myers' M / theta (integer and non-integer
    exponents) far from the defaults --, stocks from extinct to 1.6 K, 4 auto-resetting steps with external noise, every output of
    every step against the oracle's float64 evaluation of growth_models.py:208-261: float64 within 2e-14 of the population and the
    reward exact, float32 within (1e-6 + 2e-6 * population) * max(1, e^|r|) (the north star's absolute 1e-6 is a statement at the
    defaults -- K = 1, r = 0.3; a stock of 3 K has a float32 spacing of 2.4e-7 itself, and what is left of it after a harvest that
    nearly takes it all carries that spacing into a growth step that multiplies a small stock by up to e^r: 1.4e-6 was seen at
    r = 1.56 under Ricker), `done` equal except where the population is within the bar of zero.  (`seed_offset`: tests/fuzz_differential.py runs the same body from other seeds.)"""
    rng = np.random.default_rng(8800 + model + 1000 * seed_offset)
    n = 1027
    kind = fo.KIND_OF_MODEL[model]
    for trial in range(24):
        K = float(rng.choice([0.5, 1.0, 1.0, 2.0, 1.5]))
        P = dict(r=float(rng.uniform(0.05, 1.6)), K=K, sigma=float(rng.choice([0.0, 0.05, 0.2])), C=float(rng.uniform(0.05, 0.8)) * K,
                 M=float(rng.uniform(0.6, 2.0)) * K, theta=float(rng.choice([1.0, 2.0, 3.0, 1.5, 2.5])),
                 q=float(rng.choice([1.0, 2.0, 3.0, 2.5])), b=float(rng.uniform(0.1, 0.4)) * K, a=float(rng.uniform(0.0, 0.2)) * K)
        if kind == fo.KIND_RICKER and trial % 4 == 0:
            P["r"] = float(rng.uniform(-0.4, 0.0))
        x0, Tmax = 0.75 * K, int(rng.integers(2, 7))
        p = hh.params(model, r=P["r"], K=K, sigma=P["sigma"], C=P["C"], M=P["M"], theta=P["theta"], q=P["q"], b=P["b"], a=P["a"],
                      x0=x0, Tmax=Tmax, auto_reset=True)
        obs = rng.uniform(-1.0, 0.6, n).astype(dtype)
        obs[::97] = -1.0                                        # extinct stocks: log(0) inside every growth function
        t = rng.integers(0, Tmax + 1, n).astype(np.int32)
        st = hh.State(n, dtype, model, obs, t=t, terminal=True)
        for s in range(4):
            a = rng.uniform(-1.2, 0.2, n).astype(np.float32)
            z = rng.standard_normal(n).astype(dtype)
            o, rew, done, t2 = st.step(p, a, z=z, seed=trial, step_counter=s)
            eo, er, ed, et, ex = fo.step_zoo(model, obs.astype(np.float64), t, a, z.astype(np.float64), P, K, Tmax=Tmax)
            term = st.terminal.cpu().numpy().astype(np.float64)
            pop, ref = (term + 1.0) * K, (eo + 1.0) * K
            # (a Ricker stock above K under a NEGATIVE rate runs away -- 1e185 fish after four steps were seen: the reference's
            # exp(mu) then carries the rounding of mu itself, half an ulp of |mu| = 426, which the algebraic form does not
            # (fishing_common.h: zoo_draw_f64) -- the bar grows by that much; float32 overflows to inf beyond 3.4e38)
            with np.errstate(all="ignore"):
                mu_ulp = 2.3e-16 * np.abs(np.log(np.abs(ref)))
            mu_ulp = np.where(np.isfinite(mu_ulp), mu_ulp, 0.0)
            if dtype == np.float64:
                bad = ~((np.abs(pop - ref) <= (F64_RTOL + mu_ulp) * np.abs(ref) + 2 * 2.3e-16 * K) | (np.isnan(pop) & np.isnan(ref)))
                assert not bad.any(), (kind, trial, s, P, np.argwhere(bad)[0], pop[bad][0], ref[bad][0])
                assert np.array_equal(rew, er, equal_nan=True), (kind, trial, s)
            else:
                amp = max(1.0, float(np.exp(abs(P["r"]))))
                bad = ~((np.abs(pop - ref) <= (1e-6 + 2e-6 * np.abs(ref)) * amp) | (np.isnan(pop) & np.isnan(ref))
                        | (np.isinf(pop) & (np.abs(ref) > 1e38) & (np.sign(pop) == np.sign(ref))))
                assert not bad.any(), (kind, trial, s, P, np.argwhere(bad)[0], pop[bad][0], ref[bad][0])
                assert ((np.abs(rew.astype(np.float64) - er) <= 1e-6 + 2e-6 * np.abs(er)) | (np.isnan(rew) & np.isnan(er))
                        | (np.isinf(rew) & (np.abs(er) > 1e38))).all(), (kind, trial, s)
            differ = done != ed
            assert (np.abs(ex[differ]) <= 1e-6 * max(1.0, float(np.exp(abs(P["r"]))))).all() and (t2 == np.where(done.astype(bool), 0, et)).all(), (kind, trial, s)
            m = done.astype(bool)                               # follow the device's state: errors must not compound in the comparison
            obs = np.where(m, dtype(x0 / K - 1.0), st.terminal.cpu().numpy()).astype(dtype)
            assert np.array_equal(o, obs, equal_nan=True), (kind, trial, s)
            t = np.where(m, 0, et).astype(np.int32)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("n", [4096, 4099], ids=["lean", "general"])
def test_v11_index_outside_the_zoo_steps_as_beverton_holt(hh, n, dtype):
    """include/fishing_hip.h, FishingBuffers.model_idx: an index outside [0, FISHING_N_KINDS) steps as Beverton-Holt -- in the lean
    kernels (which clamp where they load it, round 5) and in the general kernel alike -- and whatever is written back for it is
    Beverton-Holt's index or the index itself."""
    rng = np.random.default_rng(11)
    table = [dict(d, sigma=0.1) for d in fo.V11_TABLE]
    p11 = hh.params(fo.MODEL_V11, sigma=0.0, K=1.0, x0=0.75, Tmax=100, models=[0, 1, 2, 3, 4], zoo_table=table)
    obs = rng.uniform(-0.9, 0.4, n).astype(dtype)
    t = rng.integers(0, 90, n).astype(np.int32)
    a = rng.uniform(-1.1, 0.3, n).astype(np.float32)
    z = rng.standard_normal(n).astype(dtype)
    kinds = rng.integers(0, 5, n).astype(np.int32)
    wild = kinds.copy()
    where = rng.random(n) < 0.2
    wild[where] = rng.choice(np.array([-7, -1, 5, 6, 1 << 20, -(1 << 31)], np.int64), int(where.sum())).astype(np.int32)
    tame = np.where(where, fo.KIND_BH, kinds).astype(np.int32)
    A = hh.State(n, dtype, fo.MODEL_V11, obs, t=t, model_idx=wild)
    B = hh.State(n, dtype, fo.MODEL_V11, obs, t=t, model_idx=tame)
    it = {4: np.uint32, 8: np.uint64}[np.dtype(dtype).itemsize]
    for k in range(3):      # (auto-reset on: finished envs redraw their index, neighbours in the quad are written back with them)
        oa, ra, da, ta = A.step(p11, a, z=z, seed=3, step_counter=k)
        ob, rb, db, tb = B.step(p11, a, z=z, seed=3, step_counter=k)
        assert np.array_equal(oa.view(it), ob.view(it)) and np.array_equal(ra.view(it), rb.view(it)), k
        assert np.array_equal(da, db) and np.array_equal(ta, tb), k
        ka, kb = A.model_idx.cpu().numpy(), B.model_idx.cpu().numpy()
        assert ((ka == kb) | ((ka == wild) & (kb == fo.KIND_BH))).all(), k


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("ret", [False, True], ids=["plain", "returns"])
def test_v11_lean_and_general_kernels_agree(hh, ret, dtype):
    """fishing-v11 takes the lean step kernel too (growth function per env: its coefficients come from a table in LDS -- rounds
    2-4 regrouped the wave's envs by kind --; the kinds are redrawn at every auto-reset) -- exact instantiations in both layouts (float64: round 4; the catch-all's
    one-tile form in round 3, the general kernel before): same bits as the general kernel on every stream and on the kind array
    over 14 auto-resetting steps, three-model list in a non-default order, ragged tail included."""
    import torch
    n = 1024 * 6 + 13
    lib = __import__("gym_fishing_amd")._capi.lib()
    table = [dict(d, sigma=0.1) for d in fo.V11_TABLE]
    kw = dict(sigma=0.1, Tmax=4, auto_reset=True, models=[4, 0, 3], zoo_table=table)
    pa, pb = hh.params(fo.MODEL_V11, **kw), hh.params(fo.MODEL_V11, general=True, **kw)
    A, B = (hh.State(n, dtype, fo.MODEL_V11, np.zeros(n), ep_return=ret, model_idx=np.zeros(n, np.int32)) for _ in range(2))
    f32 = dtype == np.float32
    # (float64: two envs per thread at cache-resident sizes, like every other model of the layout -- round 5, once the
    # growth function's coefficients came from an LDS table instead of a regroup of the lane's four envs)
    assert hh.kernel_name(pa, n, A.buffers(A.obs), dtype) == "fishing::step_kernel_lean<%s, 105, %d%s>" % (
        "float" if f32 else "double", 8198 if ret else 8194, "" if f32 else ", 2")
    assert hh.kernel_name(pb, n, B.buffers(B.obs), dtype) == "fishing::step_kernel<%s, 105>" % ("float" if f32 else "double")
    step = lib.fishing_step_f32 if f32 else lib.fishing_step_f64
    A.reset(pa, seed=5, env_offset=12)
    B.reset(pb, seed=5, env_offset=12)
    assert len(set(A.model_idx.cpu().tolist())) == 3
    g = torch.Generator(device="cuda").manual_seed(n)
    for s in range(14):
        a = (torch.rand(n, device="cuda", generator=g) * 1.4 - 1.2).float()
        for st, p in ((A, pa), (B, pb)):
            assert step(p, n, 12, st.buffers(a), 5, s, None) == 0
        torch.cuda.synchronize()
        for name in ("obs", "reward", "done", "t", "model_idx") + (("ep_return",) if ret else ()):
            x, y = getattr(A, name), getattr(B, name)
            it = {1: torch.uint8, 4: torch.int32, 8: torch.int64}[x.element_size()]
            assert torch.equal(x.view(it), y.view(it)), (name, s)
    if ret:
        ra, rb = A.record(), B.record()
        assert ra[2] == rb[2] and ra[3] == rb[3] and np.allclose(ra[:2], rb[:2], rtol=1e-12) and ra[2] > n


ZOO_BY_NAME = {c.name: c for c in load_zoo_cases()}


@pytest.mark.parametrize("name", sorted(ZOO_BY_NAME))
def test_zoo_scalar_protocol_seeded_like_the_reference(name):
    """np.random.seed(s) + the drop-in scalar env, driven as the fixtures were captured: the env consumes NumPy's
    global stream like the reference (one lognormal's normal per step, np.random.choice(models) per fishing-v11
    episode), so the whole free-running trajectory follows the reference -- exactly for rewards taken from an
    unchanged stock, the growth-function picks of fishing-v11 and fishing-v10's drifting r, within the
    accumulated log / exp rounding for the populations."""
    import gym_fishing_amd as gf
    c = ZOO_BY_NAME[name]
    names = ["allen", "beverton_holt", "myers", "may", "ricker"]
    for e, seed in enumerate(c.meta["seeds"][:3]):
        np.random.seed(seed)
        env = gf.make(c.id, **c.kwargs)
        K = float(env.params["K"])
        obs = env.reset()
        assert obs[0] == c.reset_obs[e, 0]
        for s in range(c.nsteps):
            if c.id == "fishing-v11":
                assert env.model == names[c.model_idx[e, s]], (name, e, s)
            a = np.array([c.action[e, s]], dtype=np.float32)
            obs, rew, done, _ = env.step(a)
            if c.id == "fishing-v10":
                assert float(env.r) == c.params_r[e, s] + c.kwargs["alpha"]      # drifted once more by this step
            assert np.isclose((obs[0] + 1.0) * K, (c.obs[e, s] + 1.0) * K, rtol=1e-9, atol=1e-12, equal_nan=True), (name, e, s)
            assert np.isclose(rew, c.reward[e, s], rtol=1e-9, atol=1e-12, equal_nan=True) and done == bool(c.done[e, s])
            if done:
                obs = env.reset()
                assert obs[0] == c.reset_obs[e, s + 1]
        env.close()


# ------------------------------------------------------------------ fishing-v11: population_draw / BMSY with N envs (ABI 6)
@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
def test_v11_population_draw_selects_the_growth_function_per_element(hh, dtype):
    """fishing_population_draw_* with model_idx (ABI 6; ModelUncertainty.population_draw, growth_models.py:190-194): element i
    grows under growth function model_idx[i] with THAT function's parameter set -- against the oracle's
    zoo_population_draw per kind over one mixed array; extinct stocks and an out-of-range kind (-> Beverton-Holt, as in
    the step kernels) included."""
    import torch
    from gym_fishing_amd import _capi
    lib = _capi.lib()
    n = 5 * 4001 + 3
    rng = np.random.default_rng(11)
    kinds = rng.integers(0, 5, n).astype(np.int32)
    kinds[-3:] = [7, -1, 99]
    table = [dict(d, sigma=0.1) for d in fo.V11_TABLE]
    x = (rng.uniform(0.0, 2.0, n)).astype(dtype)
    x[::501] = 0.0
    z = rng.standard_normal(n).astype(dtype)
    p = hh.params(fo.MODEL_V11, models=[0, 1, 2, 3, 4], zoo_table=table)
    xt, zt, kt = hh.dev(x), hh.dev(z), hh.dev(kinds)
    out = torch.empty_like(xt)
    fn = lib.fishing_population_draw_f32 if dtype == np.float32 else lib.fishing_population_draw_f64
    assert fn(p, n, xt.data_ptr(), zt.data_ptr(), kt.data_ptr(), None, None, out.data_ptr(), None) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float64)
    eff = np.where((kinds >= 0) & (kinds < 5), kinds, fo.KIND_BH)
    want = np.zeros(n)
    for k in range(5):
        m = eff == k
        want[m] = fo.zoo_population_draw(k, x[m].astype(np.float64), z[m].astype(np.float64), table[k])
    assert (np.isnan(got) == np.isnan(want)).all()
    ok = ~np.isnan(want)
    if dtype == np.float64:
        assert (np.abs(got - want)[ok] <= F64_RTOL * np.abs(want[ok])).all()
    else:
        assert np.abs(got - want)[ok].max() <= F32_ATOL * 1.5           # (populations, in units of the largest K = 1.5)
    assert (got[x == 0] == 0).all()
    # without the selector fishing-v11 has no growth function to apply; with any other model there is nothing to select
    assert fn(p, n, xt.data_ptr(), zt.data_ptr(), None, None, None, out.data_ptr(), None) == -1        # FISHING_ERR_NULL
    p9 = hh.params(fo.MODEL_V9, sigma=0.1)
    assert fn(p9, n, xt.data_ptr(), zt.data_ptr(), kt.data_ptr(), None, None, out.data_ptr(), None) == -7      # FISHING_ERR_UNSUPPORTED


def test_v11_num_envs_population_draw_and_bmsy_follow_each_envs_model(hh):
    """An N-env fishing-v11 batch: population_draw(x) grows env i's stock under env i's model in force (round 3 raised
    NotImplementedError here), BMSY(env) returns one S per env -- that of its growth function, equal to what the scalar
    protocol's BMSY returns with that model in force (models/policies.py:51-67) -- and msy / escapement built on it
    drive the batch through simulate()."""
    import torch
    import gym_fishing_amd as gf
    from gym_fishing_amd import policies
    n = 64
    env = gf.make("fishing-v11", num_envs=n, seed=3)
    env.reset()
    idx = env.model_idx.clone()
    assert len(set(idx.cpu().tolist())) == 5
    x = torch.linspace(0.1, 1.4, n, device="cuda", dtype=torch.float64)
    got = env.population_draw(x, dtype=torch.float64).cpu().numpy()
    want = np.zeros(n)
    for k in range(5):
        m = idx.cpu().numpy() == k
        want[m] = fo.zoo_population_draw(k, x.cpu().numpy()[m], np.zeros(m.sum()), fo.V11_TABLE[k])
    assert np.allclose(got, want, rtol=F64_RTOL, atol=0)
    with pytest.raises(ValueError):
        env.population_draw(x[:5])                       # neither one per env nor a model_idx
    S = policies.BMSY(env)
    assert isinstance(S, torch.Tensor) and S.shape == (n,)
    names = ["allen", "beverton_holt", "myers", "may", "ricker"]
    per_kind = {}
    for k, name in enumerate(names):
        one = gf.make("fishing-v11", models=[name])      # scalar protocol, this growth function in force
        per_kind[k] = policies.BMSY(one)
    want_S = np.array([per_kind[int(k)] for k in idx.cpu().tolist()], dtype=np.float32)
    assert np.array_equal(S.cpu().numpy(), want_S)
    esc = policies.escapement(env)
    assert esc.kernel_policy[1] is esc.S and esc.S.shape == (n,)        # (one S per env: fishing_rollout_params_*, ABI 7)
    a, _ = esc.predict(env.state)
    assert a.shape == (n, 1) and a.dtype == torch.float32
    df = env.simulate(esc)
    assert len(df) > n and df["reward"].sum() > 0
    m = policies.msy(env)
    # (f(S_i) - S_i under the model drawn by BMSY's reset -- the reference's order of events -- may well be negative)
    assert m.msy.shape == (n,) and bool(torch.isfinite(m.msy).all())
    a, _ = m.predict(env.state)
    assert a.shape == (n, 1)


def test_zoo_f64_log_exp_are_within_one_ulp(hh):
    """The float64 parity layout's own log / exp (csrc/fishing_common.h: log_f64 / exp_f64, the msun argument reductions
    and coefficients with the divisions done by Newton steps) against libm, element by element: <= 1 ulp over the
    populations and exponents the growth functions see and far beyond, special values exact; any other function id is refused."""
    import math
    import torch
    from gym_fishing_amd import _capi
    lib = _capi.lib()

    def run(fn, v):
        t = hh.dev(np.asarray(v, dtype=np.float64))
        o = torch.empty_like(t)
        assert lib.fishing_math_f64(t.numel(), fn, t.data_ptr(), o.data_ptr(), None) == 0
        torch.cuda.synchronize()
        return o.cpu().numpy()
    rng = np.random.default_rng(8)
    v = np.concatenate([np.exp(rng.uniform(-40, 40, 200000)), 1.0 + rng.uniform(-0.3, 0.45, 100000), rng.uniform(0, 3, 100000),
                        [5e-324, 2.2250738585072014e-308, 1.0, 0.5, 2.0, 1e308]])
    got = run(0, v)
    want = np.array([math.log(x) for x in v])
    assert hh.ulp_diff(got, want).max() <= 1
    special = run(0, [0.0, -0.0, -1.0, np.inf, np.nan])
    assert special[0] == -np.inf and special[1] == -np.inf and np.isnan(special[2]) and special[3] == np.inf and np.isnan(special[4])
    y = np.concatenate([rng.uniform(-745, 709, 200000), rng.uniform(-3, 3, 200000), [0.0, -0.0, 1.0, -1.0, 709.78, -745.13]])
    got = run(1, y)
    want = np.array([math.exp(x) for x in y])
    assert hh.ulp_diff(got, want).max() <= 1
    special = run(1, [-np.inf, np.inf, np.nan, 710.0, -746.0])
    assert special[0] == 0.0 and special[1] == np.inf and np.isnan(special[2]) and special[3] == np.inf and special[4] == 0.0
    t = hh.dev(np.ones(4))
    for fn in (-1, 2, 3, 4):        # (2-4: the polynomial forms of builds that no longer exist)
        assert lib.fishing_math_f64(4, fn, t.data_ptr(), t.data_ptr(), None) == -4      # FISHING_ERR_SIZE


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
def test_zoo_special_values_follow_the_reference(hh, dtype):
    """The growth functions at the edges of their domain, through fishing_population_draw_*, against the oracle's float64
    evaluation of the reference's log / exp round trip on the same inputs: extinct, tiny, huge, infinite and NaN stocks
    under zero, large, infinite and NaN noise.  Both layouts evaluate an algebraically equal form WITHOUT the round
    trip (fishing_common.h: zoo_draw_f32, zoo_draw_f64) -- this is where "equal" is checked value by value: the same
    NaNs, the same zeros, the same infinities, finite values within the layout's tolerance (x' up to 1e6 here: relative).
    The float64 layout's hand-over to the reference's own round trip (far stocks, far results) is crossed in both directions:
    x = 1e-12 and 1e12 lie beyond it, Allen / Ricker at x = 1e3 produce results beyond it from a stock inside."""
    import torch
    from gym_fishing_amd import _capi
    lib = _capi.lib()
    tiny = 1e-30
    xs = np.array([0.0, tiny, 1e-12, 1e-3, 0.4, 1.0, 2.5, 1e3, 1e6, 1e12, np.inf, np.nan])
    zs = np.array([0.0, 1.0, -1.0, 6.5, -6.5, np.inf, -np.inf, np.nan])
    X, Z = (a.reshape(-1) for a in np.meshgrid(xs, zs, indexing="ij"))
    n = X.size
    fn = lib.fishing_population_draw_f32 if dtype == np.float32 else lib.fishing_population_draw_f64
    # (the last row: Myers with r < -1 -- log(r + 1) is NaN in the reference and so is every population; an algebraic form
    # that carried A = r + 1 < 0 through max(0, .) would report an extinct stock instead)
    for env_id, override in (("fishing-v5", {}), ("fishing-v6", {}), ("fishing-v7", {}), ("fishing-v8", {}), ("fishing-v9", {}),
                             ("fishing-v8", {"r": -1.5})):
        model = fo.MODEL_OF_ID[env_id]
        P = dict(ZOO_DEFAULTS[env_id], sigma=0.2, **override)
        p = hh.params(model, r=float(P.get("r", 0.3)), K=float(P["K"]), sigma=0.2, C=float(P.get("C", 0.5)), M=float(P.get("M", 0.0)),
                      theta=float(P.get("theta", 0.0)), q=float(P.get("q", 0.0)), b=float(P.get("b", 0.0)), a=float(P.get("a", 0.0)))
        xt, zt = hh.dev(X.astype(dtype)), hh.dev(Z.astype(dtype))
        out = torch.empty_like(xt)
        assert fn(p, n, xt.data_ptr(), zt.data_ptr(), None, None, None, out.data_ptr(), None) == 0
        torch.cuda.synchronize()
        got = out.cpu().numpy().astype(np.float64)
        with np.errstate(all="ignore"):
            want = fo.zoo_population_draw(fo.KIND_OF_MODEL[model], X.astype(dtype).astype(np.float64), Z.astype(dtype).astype(np.float64), P)
            want = want.astype(dtype).astype(np.float64)          # (what the layout can hold: float32 overflows to inf, underflows to 0)
        bad = []
        for i in range(n):
            if dtype == np.float32 and np.isinf(Z[i]) and 0.0 < X[i] < 1e-10:
                # x ** 3 underflows float32 (1e-90): the algebraic form sees an extinct stock (0 * inf = NaN), the round
                # trip's log does not.  Infinite noise never comes out of the generator (|z| <= 6.76).
                continue
            if np.isnan(want[i]) or np.isnan(got[i]):
                ok = np.isnan(want[i]) and np.isnan(got[i])
            elif np.isinf(want[i]) or want[i] == 0.0:
                # (a float32 result within a rounding of the range's end may land on either side of it)
                ok = got[i] == want[i] or (dtype == np.float32 and (got[i] > 1e38 or got[i] < 1e-37))
            else:
                ok = abs(got[i] - want[i]) <= (2e-14 if dtype == np.float64 else 2e-6) * abs(want[i]) + 1e-300
            if not ok:
                bad.append((env_id, float(X[i]), float(Z[i]), float(got[i]), float(want[i])))
        assert not bad, bad


@pytest.mark.parametrize("env_id", ["fishing-v9", "fishing-v10", "fishing-v11"])
def test_zoo_shards_reproduce_the_whole_batch_at_full_size(hh, env_id):
    """SURVEY 8(e) for the zoo at a BASELINE-sized batch: N = 2^22 envs stepped as one batch and as 8 shards (env_offset =
    the shard's first global index) land on the same bits -- noise, fishing-v11's model draws (reset and auto-reset) and
    fishing-v10's drifting r are functions of the GLOBAL env index -- and two runs of the whole batch are identical."""
    import torch
    import gym_fishing_amd as gf
    n, shards, steps = 1 << 22, 8, 6
    g = torch.Generator(device="cuda").manual_seed(5)
    acts = torch.rand((steps, n), device="cuda", generator=g) * 1.6 - 1.2

    def mk(count, off):
        kw = {} if env_id == "fishing-v11" else dict(sigma=0.1)
        env = gf.make(env_id, num_envs=count, env_offset=off, seed=17, Tmax=4, **kw)
        if env_id == "fishing-v11":
            for d in env.model_params.values():
                d["sigma"] = 0.1
        env.reset()
        return env

    def run(env, a):
        for s in range(steps):
            env.step(a[s])
        torch.cuda.synchronize()
        extra = env._model_idx if env_id == "fishing-v11" else (env._r_arr if env_id == "fishing-v10" else None)
        return env._obs.clone(), env._t.clone(), None if extra is None else extra.clone()
    whole = run(mk(n, 0), acts)
    again = run(mk(n, 0), acts)
    for a, b in zip(whole, again):
        assert a is None or torch.equal(a, b)
    per = n // shards
    for k in range(shards):
        part = run(mk(per, k * per), acts[:, k * per:(k + 1) * per].contiguous())
        for a, b in zip(whole, part):
            assert a is None or torch.equal(a[k * per:(k + 1) * per], b), (env_id, k)
    assert bool(torch.isfinite(whole[0]).all())
    if env_id == "fishing-v11":
        assert len(set(whole[2][:4096].cpu().tolist())) == 5


# ------------------------------------------------------------------ the module-level growth functions (growth_models.py:208-269)
def _growth_close(got, want, dtype):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape
    both_nan = np.isnan(got) & np.isnan(want)
    if dtype == "f64":      # a few ulp of exp(mu): the zoo's float64 bar (2e-14 of the population)
        ok = np.abs(got - want) <= F64_RTOL * np.abs(want)
    else:                   # float32: 1e-6 absolute up to a population of 1, relative above
        ok = np.abs(got - want) <= F32_ATOL * np.maximum(1.0, np.abs(want))
    assert (ok | both_nan).all(), (np.argwhere(~(ok | both_nan))[0], got[~(ok | both_nan)][0], want[~(ok | both_nan)][0])


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_module_level_growth_functions_follow_the_reference(dtype):
    """allen / beverton_holt / may / myers / ricker (x, params), as a user of the reference calls them: seeded like the
    reference's run (np.random.seed, then a vector, a matrix and a scalar of populations -- each call consumes one legacy
    standard normal per element), NumPy in -> float64 NumPy of the same shape out, scalar in -> scalar out; the same
    through population_model[name]; and on device tensors with the recorded normals passed explicitly."""
    import torch
    from conftest import load_growth_function_cases
    from gym_fishing_amd import growth_models as gm
    td = torch.float64 if dtype == "f64" else torch.float32
    assert list(gm.population_model) == ["allen", "beverton_holt", "myers", "may", "ricker"]
    for c in load_growth_function_cases():
        f = getattr(gm, c["name"])
        assert gm.population_model[c["name"]] is f
        np.random.seed(c["seed"])
        for shape_tag, x, z, want in c["calls"]:
            arg = np.float64(x) if shape_tag == "scalar" else x
            got = f(arg, c["params"], dtype=td)
            assert isinstance(got, np.ndarray if shape_tag != "scalar" else np.floating) and np.asarray(got).dtype == np.float64
            _growth_close(got, want, dtype)
        # ... and the stream stands where the reference's stands after the three calls
        st = np.random.get_state()
        np.random.seed(c["seed"])
        np.random.normal(0, 1, sum(call[1].size for call in c["calls"]))
        assert all(np.array_equal(a, b) for a, b in zip(st[1:], np.random.get_state()[1:]) if isinstance(a, np.ndarray)) \
            and st[2:] == np.random.get_state()[2:]
        shape_tag, x, z, want = c["calls"][1]
        got = f(torch.as_tensor(x, device="cuda", dtype=td), c["params"], noise=torch.as_tensor(z, device="cuda"), dtype=td)
        assert isinstance(got, torch.Tensor) and got.is_cuda and got.dtype == td and tuple(got.shape) == x.shape
        _growth_close(got.cpu().numpy(), want, dtype)
    # the reference's KeyError for a parameter the function reads
    with pytest.raises(KeyError):
        gm.allen(0.5, {"r": 0.3, "K": 1.0, "sigma": 0.0})
    with pytest.raises(ValueError):
        gm.ricker(np.ones(4), {"r": 0.3, "K": 1.0, "sigma": 0.0}, noise=np.zeros(3))


def test_v11_fused_rollout_under_each_envs_own_escapement_level():
    """N fishing-v11 envs, each with the S that BMSY() found for the growth function in force there (policies.escapement:
    a tensor S), rolled out inside the fused kernel (fishing_rollout_params_*, ABI 7) == the step loop driven by the
    policy's predict() on a twin restored from the same checkpoint -- observations, actions, rewards, dones, the models
    redrawn at the auto-resets: bit for bit in the float64 layout; and simulate_mdp_vec takes that fused path."""
    import torch
    import gym_fishing_amd as gf
    from gym_fishing_amd import policies, rollout
    n, T = 1024, 30
    mk = lambda: gf.make("fishing-v11", num_envs=n, seed=21, Tmax=9, dtype=torch.float64)      # noqa: E731
    A = mk()
    for d in A.model_params.values():
        d["sigma"] = 0.05
    A.reset()
    model = policies.escapement(A)
    S = model.kernel_policy[1]
    assert isinstance(S, torch.Tensor) and S.shape == (n,) and len(set(S.cpu().tolist())) >= 3     # one level per growth function
    B = mk()
    for d in B.model_params.values():
        d["sigma"] = 0.05
    B.load_state_dict(A.state_dict())
    traj = A.rollout(T, policy=model.kernel_policy, record=True)
    model.env = B
    bits = lambda x: x.contiguous().view(torch.int64)          # noqa: E731
    for s in range(T):
        o = B.state.clone()
        a, _ = model.predict(o)
        assert torch.equal(bits(traj[s, 0]), bits(o.reshape(-1))), s
        assert torch.equal(bits(traj[s, 1]), bits(a.reshape(-1).to(traj.dtype))), s
        _, rew, done, _ = B.step(a.reshape(-1))
        assert torch.equal(bits(traj[s, 2]), bits(rew)) and torch.equal(traj[s, 3].bool(), done.bool()), s
    assert torch.equal(bits(A.state), bits(B.state)) and torch.equal(A.model_idx, B.model_idx)
    # simulate_mdp_vec: the fused path (kernel_policy with a tensor) and the step loop (kernel_policy hidden) build the same table
    model.env = A

    class Hidden:
        def __init__(self, m):
            self.m = m

        def predict(self, obs, **kw):
            return self.m.predict(obs)
    C = mk()
    for d in C.model_params.values():
        d["sigma"] = 0.05
    C.load_state_dict(A.state_dict())
    fused = rollout.simulate_mdp_vec(A, model, n)
    model.env = C
    loop = rollout.simulate_mdp_vec(C, Hidden(model), n)
    fa = fused.to_numpy(dtype=np.float64) if hasattr(fused, "to_numpy") else np.stack([fused[c] for c in rollout.COLUMNS], 1)
    la = loop.to_numpy(dtype=np.float64) if hasattr(loop, "to_numpy") else np.stack([loop[c] for c in rollout.COLUMNS], 1)
    assert fa.shape == la.shape == (n * 10, 5) and np.array_equal(fa, la, equal_nan=True)
