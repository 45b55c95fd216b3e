"""The reference's callers of the env on the N-env path (shared_env.py:15-102): simulate_mdp_vec tables and policyfn tables
row for row against the reference-held fixtures, the fused path behind simulate_mdp_vec, fishing-v10's drifting r in
population_draw (growth_models.py:151)."""
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
