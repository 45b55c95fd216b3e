"""GPU tests of the host-side mirror (gym_fishing_amd/envs.py): the reference's gym.Env
protocol, scalar and vectorised, driven through the Python API a user of the reference
would call.  Numerics are checked against the golden vectors / the oracle."""
import numpy as np
import pytest

from conftest import load_golden_cases
from oracle import fishing_oracle as fo

pytestmark = pytest.mark.gpu
CASES = {c.name: c for c in load_golden_cases()}


@pytest.fixture(scope="module")
def gf():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device; none visible")
    import gym_fishing_amd
    return gym_fishing_amd


def bits(x):
    return np.asarray(x, dtype=np.float64).view(np.int64)


@pytest.mark.parametrize("name", sorted(CASES))
def test_scalar_protocol_seeded_like_the_reference_reproduces_it(gf, name):
    """The north star's "seeded identically": np.random.seed(s), then the drop-in env driven exactly like the
    reference was when the fixtures were captured (tests/golden/make_golden.py: construct, reset, step, reset on
    done) -- no recorded noise handed in.  The scalar protocol draws from NumPy's global legacy stream in the
    reference's order (one normal per step also at sigma = 0, K then r for fishing-v4) and the kernels do the
    arithmetic: every obs / reward / done / K / r equals the reference bit-for-bit (fishing-v2: exp tolerance)."""
    c = CASES[name]
    is_v2, is_v4 = c.id == "fishing-v2", c.id == "fishing-v4"
    for e, seed in enumerate(c.meta["seeds"]):
        np.random.seed(seed)
        env = gf.make(c.id, **c.kwargs)
        if c.init_reset:
            obs = env.reset()
            assert bits(obs[0]) == bits(c.reset_obs[e, 0])
        for s in range(c.nsteps):
            if is_v4:
                assert bits(float(env.K)) == bits(c.K[e, s]) and bits(float(env.r)) == bits(c.r[e, s]), (name, e, s)
            a = int(c.action[e, s]) if c.id == "fishing-v0" else np.array([c.action[e, s]], dtype=np.float32)
            obs, rew, done, info = env.step(a)
            if is_v2:
                assert abs(obs[0] - c.obs[e, s]) < 1e-12, (name, e, s)   # free-running: exp ulps accumulate
            else:
                assert bits(obs[0]) == bits(c.obs[e, s]) or (np.isnan(obs[0]) and np.isnan(c.obs[e, s])), (name, e, s)
                assert (bits(rew) == bits(c.reward[e, s]) or (np.isnan(rew) and np.isnan(c.reward[e, s]))) and done == bool(c.done[e, s])     # (a NaN's sign bit is the host's / the device's own)
            assert env.years_passed == c.t[e, s]
            if done and c.auto_reset:
                obs = env.reset()
                assert bits(obs[0]) == bits(c.reset_obs[e, s + 1])
        env.close()
    with pytest.raises(ValueError):
        gf.make("fishing-v4", num_envs=8, rng="numpy")


@pytest.mark.parametrize("env_id", ["fishing-v0", "fishing-v1", "fishing-v2"])
def test_n_env_numpy_rng_equals_a_dummy_vec_env_of_reference_envs(gf, env_id):
    """rng="numpy" with N envs: one np.random.normal(0, 1, N) per step = the draws SB3's DummyVecEnv makes when it
    steps N reference envs one after the other.  Against N reference-equivalent scalar envs (oracle/scalar_env.py,
    pinned to the reference bit-for-bit) sharing the global stream: same obs / reward / done, bit-for-bit in
    float64 (fishing-v2: exp tolerance), over 130 steps with resets on done."""
    import torch
    from oracle.scalar_env import ScalarFishingEnv
    n, T, seed = 8, 130, 77
    kw = dict(sigma=0.1, Tmax=40)
    rng = np.random.RandomState(5)
    acts = (rng.randint(0, 60, (T, n)) if env_id == "fishing-v0" else rng.uniform(-1, -0.3, (T, n)).astype(np.float32))
    np.random.seed(seed)
    refs = [ScalarFishingEnv(env_id, **kw) for _ in range(n)]
    for r in refs:
        r.reset()
    want = []
    for t in range(T):
        row = []
        for i, r in enumerate(refs):
            # float32 action value, float64 arithmetic: the reference-era promotion, as in the fixtures
            a = int(acts[t, i]) if env_id == "fishing-v0" else np.array([acts[t, i]], dtype=np.float32).astype(np.float64)
            o, rew, d, _ = r.step(a)
            row.append((float(o[0]), float(rew), bool(d)))
            if d:
                r.reset()
        want.append(row)
    np.random.seed(seed)
    env = gf.make(env_id, num_envs=n, dtype=torch.float64, rng="numpy", record_terminal_obs=True, **kw)
    env.reset()
    for t in range(T):
        a = torch.as_tensor(acts[t])
        _, rew, done, info = env.step(a)
        term = info["terminal_observation"].cpu().numpy().reshape(-1)
        for i in range(n):
            o, r, d = want[t][i]
            if env_id == "fishing-v2":
                assert abs(term[i] - o) < 1e-11 and abs(float(rew[i]) - r) < 1e-11
            else:
                assert bits(term[i]) == bits(o) and bits(float(rew[i])) == bits(r), (t, i)
            assert bool(done[i]) == d


@pytest.mark.parametrize("name", ["v1_sigma0_const", "v1_sigma01_random", "v1_params", "v1_edge_noreset",
                                  "v0_sigma01_random", "v0_edge", "v2_sigma0_zeroquota", "v4_sigma005",
                                  "v4_noinitreset"])
def test_scalar_protocol_reproduces_reference(gf, name):
    """make(id, **kw) with no num_envs: the reference's scalar protocol, types included
    (base_fishing_env.py:60-91).  Noise = the reference's recorded normals (external-noise
    mode); for fishing-v4 the recorded (K, r) are installed after each reset."""
    c = CASES[name]
    is_v2 = c.id == "fishing-v2"
    for e in range(min(3, c.obs.shape[0])):
        env = gf.make(c.id, **c.kwargs)
        assert env.observation_space.shape == (1,) and env.num_envs == 1
        if c.init_reset:
            obs = env.reset()
            assert isinstance(obs, np.ndarray) and obs.shape == (1,) and obs.dtype == np.float64
        if c.id == "fishing-v4":
            env.K, env.r = c.K[e, 0], c.r[e, 0]
            if c.init_reset:
                assert env.state[0] == c.reset_obs[e, 0]
        for s in range(c.nsteps):
            a = int(c.action[e, s]) if c.id == "fishing-v0" else np.array([c.action[e, s]], dtype=np.float32)
            obs, rew, done, info = env.step(a, noise=[c.z[e, s]])
            assert isinstance(obs, np.ndarray) and obs.shape == (1,) and obs.dtype == np.float64
            assert isinstance(rew, float) and isinstance(done, bool) and info == {}
            if is_v2:
                assert abs(obs[0] - c.obs[e, s]) < 1e-15
            else:
                assert bits(obs[0]) == bits(c.obs[e, s]), (name, e, s)
            assert bits(rew) == bits(c.reward[e, s]) and done == bool(c.done[e, s])
            assert env.years_passed == c.t[e, s]
            if done and c.auto_reset:
                obs = env.reset()
                assert bits(obs[0]) == bits(c.reset_obs[e, s + 1])
                if c.id == "fishing-v4" and s + 1 < c.nsteps:
                    env.K, env.r = c.K[e, s + 1], c.r[e, s + 1]
        env.close()


def test_reference_test_tipping_assertions(gf):
    """tests/test-envs.py:93-106 verbatim against this package."""
    env = gf.make("fishing-v2", sigma=0, init_state=0.75)
    env.reset()
    obs, reward, done, info = env.step(env.get_action(0))
    assert env.get_fish_population(obs) >= 0.75
    assert env.get_fish_population(obs) == 0.7641951637890196 or abs(env.get_fish_population(obs) - 0.7641951637890196) < 1e-15
    env.init_state = 0.3
    env.reset()
    obs, reward, done, info = env.step(env.get_action(0))
    assert env.get_fish_population(obs) <= 0.3


def test_vectorised_protocol_shapes_dtypes_and_auto_reset(gf):
    import torch
    n = 1000
    env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=3, Tmax=4, record_terminal_obs=True, done_bits=True,
                  track_returns=True)
    assert env.num_envs == n and env.auto_reset and env.dtype == torch.float32
    obs = env.reset()
    assert obs.shape == (n, 1) and obs.dtype == torch.float32 and obs.is_cuda
    assert bool((obs == -0.25).all())
    total_done = 0
    for s in range(12):
        a = torch.full((n, 1), -0.9, device="cuda")
        obs, rew, done, info = env.step(a)
        assert obs.shape == (n, 1) and rew.shape == (n,) and done.shape == (n,) and done.dtype == torch.bool
        term = info["terminal_observation"]
        assert term.shape == (n, 1)
        # SB3 semantics: finished envs already show the reset observation, terminal obs is kept
        assert bool((obs[done] == -0.25).all()) and bool((obs[~done] == term[~done]).all())
        bits_ = info["done_bits"].cpu().numpy().view(np.uint64)
        unpacked = ((bits_[:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).reshape(-1)[:n]
        assert (unpacked.astype(bool) == done.cpu().numpy()).all()
        total_done += int(done.sum())
    stats = env.episode_stats()
    assert stats["n_episodes"] == total_done and total_done >= 2 * n    # Tmax=4 -> 5-step episodes
    assert 0 < stats["mean_length"] <= 5
    # numpy / list actions are accepted too (copied to the device)
    obs, rew, done, _ = env.step(np.full((n, 1), -0.9, dtype=np.float32))
    assert obs.shape == (n, 1)
    with pytest.raises(ValueError):
        env.step(np.zeros(n + 1, dtype=np.float32))
    # render() works (the reference's raises, quirk B10) and reports [t, obs, action, reward]
    row = env.render(index=5)
    assert len(row) == 4 and abs(row[2] + 0.9) < 1e-6


def test_vectorised_matches_oracle_through_python_api(gf):
    """fishing-v0 via the class API at N = 4096, in-kernel noise, vs the oracle on the
    device's own normals (read back with the noise hook)."""
    import torch
    import hip_harness as hh
    n, seed = 4096, 42
    env = gf.make("fishing-v0", sigma=0.2, n_actions=50, num_envs=n, seed=seed, auto_reset=False)
    obs = env.reset().clone()
    g = torch.Generator(device="cuda").manual_seed(0)
    t = np.zeros(n, np.int32)
    for s in range(5):
        a = torch.randint(0, 50, (n,), device="cuda", generator=g)
        o_prev = obs.cpu().numpy().reshape(-1)
        obs, rew, done, _ = env.step(a)
        z = hh.device_step_noise(n, seed, s)
        eo, er, ed, t, _ = fo.step(fo.MODEL_V0, o_prev, t, a.cpu().numpy().astype(np.int32), z, 0.3, 1.0, 0.2,
                                   n_actions=50, dtype=np.float32)
        assert np.array_equal(obs.cpu().numpy().reshape(-1), eo) and np.array_equal(rew.cpu().numpy(), er)
        assert np.array_equal(done.cpu().numpy(), ed.astype(bool))
        obs = obs.clone()


def test_masked_reset_and_seed(gf):
    import torch
    n = 256
    env = gf.make("fishing-v4", sigma=0.05, num_envs=n, seed=9, auto_reset=False)
    env.reset()
    K0 = env.K.clone()
    a = torch.full((n,), -0.8, device="cuda")
    env.step(a)
    mask = torch.zeros(n, dtype=torch.bool, device="cuda")
    mask[::2] = True
    before = env.state.clone()
    env.reset(mask=mask)
    st = env.state.reshape(-1)
    assert bool((st[::2] == 0.75).all())                     # v4 reset obs is un-normalised x0 (quirk B8)
    assert bool((st[1::2] == before.reshape(-1)[1::2]).all())
    assert bool((env.K[1::2] == K0[1::2]).all()) and bool((env.K[::2] != K0[::2]).any())
    assert bool((env.years_passed[::2] == 0).all()) and bool((env.years_passed[1::2] == 1).all())
    # the masked reset kept fishing-v4 in the derived mode (per-env origin stamps, no r / K arrays): a DERIVED kernel steps it
    assert env._derived and env._K_arr is None and "28031" in env.step_kernel_name()
    stored = gf.make("fishing-v4", sigma=0.05, num_envs=n, seed=9, auto_reset=False, derived_params=False)
    stored.reset()
    stored.step(a)
    stored.reset(mask=mask)
    for _ in range(3):
        env.step(a)
        stored.step(a)
    assert torch.equal(env.state, stored.state) and torch.equal(env.K, stored.K) and torch.equal(env.r, stored.r)
    # same seed -> same parameter draws
    env2 = gf.make("fishing-v4", sigma=0.05, num_envs=n, seed=9, auto_reset=False)
    env2.reset()
    assert bool((env2.K == K0).all())
    # K, r ~ N(mean, sigma_p) clipped at 0
    big = gf.make("fishing-v4", sigma_p=0.1, num_envs=1 << 16, seed=1)
    big.reset()
    K = big.K.double()
    assert abs(float(K.mean()) - 1.0) < 2e-3 and abs(float(K.std()) - 0.1) < 2e-3 and float(K.min()) >= 0


def test_helpers_match_reference_maps(gf, anchors):
    """get_quota / get_action / get_fish_population / get_state (base_fishing_env.py:135-164)."""
    import torch
    env0 = gf.make("fishing-v0")
    env1 = gf.make("fishing-v1")
    q = np.linspace(0.0, 1.0, 21)
    assert [int(env0.get_action(x)) for x in q] == anchors["get_action_v0"]
    assert [float(env1.get_action(x)) for x in q] == anchors["get_action_v1"]
    assert [float(env0.get_quota(int(a))) for a in range(0, 101, 5)] == anchors["get_quota_v0"]
    assert env1.get_quota(np.array([5.0])) == 2.0 and env1.get_quota(np.array([-5.0])) == 0.0
    assert env1.get_fish_population(np.array([-0.25])) == 0.75
    assert env1.get_state(0.75)[0] == -0.25
    v = gf.make("fishing-v1", K=2.0, num_envs=8)
    pop = v.get_fish_population(torch.full((8, 1), -0.5, device="cuda"))
    assert pop.shape == (8,) and bool((pop == 1.0).all())
    assert bool((v.get_quota(torch.full((8,), 0.5, device="cuda")) == 3.0).all())
    assert v.get_attr("Tmax")[0] == 100 and len(v.get_attr("Tmax")) == 8
    assert v.env_method("get_fish_population", np.array([-0.5]), indices=3)[0] == 1.0


def test_step_equals_the_reference_sequence_of_public_helpers(gf):
    """base_fishing_env.py:60-81 spelled out with the public methods, as a caller of the reference may: get_quota ->
    get_fish_population -> harvest_draw -> population_draw -> get_state reproduces step() -- one env through the scalar
    protocol (harvest_draw / population_draw on self.fish_population, self.harvest kept), and N envs on tensors."""
    import torch
    a, z = np.array([-0.4], dtype=np.float32), 0.3
    stepped, manual = gf.make("fishing-v1", sigma=0.1), gf.make("fishing-v1", sigma=0.1)
    stepped.reset()
    manual.reset()
    obs, rew, done, _ = stepped.step(a, noise=[z])
    quota = manual.get_quota(a)
    manual.get_fish_population(manual.state)
    assert manual.fish_population == 0.75
    h = manual.harvest_draw(quota)
    assert h == manual.harvest == min(0.75, quota) and manual.fish_population == max(0.75 - h, 0.0)
    x1 = manual.population_draw(noise=[z])
    assert manual.get_state(x1)[0] == obs[0] and max(h, 0.0) == rew
    # a quota above the stock takes the stock, nothing is left; NaN follows Python's min / max
    manual.fish_population = 0.2
    assert manual.harvest_draw(0.5) == 0.2 and manual.fish_population == 0.0
    manual.fish_population = 0.2
    assert manual.harvest_draw(float("nan")) == 0.2 and manual.fish_population == 0.0
    # N envs: tensors in, (harvest, population left) out
    n = 64
    v, w = (gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=3, dtype=torch.float64, auto_reset=False) for _ in range(2))
    v.reset()
    w.reset()
    acts = torch.linspace(-1.2, 0.4, n, device="cuda", dtype=torch.float32)
    zs = torch.linspace(-2.0, 2.0, n, device="cuda", dtype=torch.float64)
    obs, rew, done, _ = v.step(acts, noise=zs)
    hv, left = w.harvest_draw(w.get_quota(acts))
    x1 = w.population_draw(left, noise=zs)
    assert torch.equal(w.get_state(x1), obs.double()) and torch.equal(torch.clamp(hv, min=0.0), rew.double())
    assert bool(done.any()) and torch.equal(done.bool(), x1 <= 0.0)          # (the quotas above the stock emptied it)


def test_episode_record_of_a_batch_beyond_4096_tiles(gf):
    """N = 2^22 + 3 tiles: step() runs a workgroup per tile (4099 of them), the env sizes its return_partials for the
    8192 slots such a batch can touch and reduces exactly those -- the record counts every finished episode and sums
    every finished return, checked against the done / reward streams themselves."""
    import torch
    n = (1 << 22) + 3072
    env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=5, Tmax=4, track_returns=True)
    assert env._partial_slots == 8192 and env._partials.numel() == 4 * 8192
    env.reset()
    assert env.step_kernel_name(torch.zeros(n, device="cuda")).endswith("12294, 4>")     # KP2 | RET | ONE
    g = torch.Generator(device="cuda").manual_seed(1)
    running = torch.zeros(n, dtype=torch.float64, device="cuda")
    n_done, sum_ret = 0, 0.0
    for s in range(9):
        a = torch.rand(n, device="cuda", generator=g) * 1.2 - 1.0
        obs, rew, done, _ = env.step(a)
        running += rew.double()
        n_done += int(done.sum())
        sum_ret += float(running[done].sum())
        running[done] = 0.0
    st = env.episode_stats()
    assert st["n_episodes"] == n_done and n_done > n
    assert st["mean_return"] * n_done == pytest.approx(sum_ret, rel=1e-6)
    assert int(torch.count_nonzero(env._partials.view(-1, 4)[4099:])) == 0


def test_fused_rollout_api_and_stats(gf):
    import torch
    n = 1 << 14
    env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=11, track_returns=True)
    env.reset()
    env.rollout(303, policy="escapement", param=0.5)          # 3 full 101-step episodes per env
    s = env.episode_stats()
    assert s["n_episodes"] >= 3 * n - 5 and 90 < s["mean_length"] <= 101
    # constant-escapement is the known optimum: ~ MSY * 100 per episode for r=0.3, K=1
    assert 6.5 < s["mean_return"] < 9.0
    env2 = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=11, track_returns=True)
    env2.reset()
    env2.rollout(303, policy="random")
    s2 = env2.episode_stats()
    assert s2["mean_return"] < s["mean_return"]
    traj = gf.make("fishing-v1", sigma=0.0, num_envs=8).rollout(5, policy="constant", param=-0.9375, record=True)
    assert traj.shape == (5, 4, 8)
    assert float(traj[1, 0, 0]) == pytest.approx(float.fromhex("-0x1.fc00000000000p-3"), abs=1e-7)
    assert bool((traj[:, 2] == 0.0625).all())


def test_step_many_equals_repeated_step(gf):
    import torch
    n = 5000
    acts = torch.rand((4, n), device="cuda") * 2 - 1
    a = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=5)
    b = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=5)
    a.reset()
    b.reset()
    a.step_many(acts, 10)
    for k in range(10):
        b.step(acts[k % 4])
    assert torch.equal(a.state, b.state) and torch.equal(a.years_passed, b.years_passed)


def test_render_writes_csv(gf, tmp_path):
    f = tmp_path / "log.csv"
    env = gf.make("fishing-v1", file=str(f))
    env.reset()
    env.step(np.array([-0.9375], dtype=np.float32))
    row = env.render()
    env.close()
    assert row[0] == 1 and row[3] == 0.0625
    assert f.read_text().strip().split(",")[0] == "1"


# ------------------------------------------------------------------ policies + simulate
from conftest import load_policy_sims  # noqa: E402

SIMS = load_policy_sims()


@pytest.mark.parametrize("tag,env_id,kw", [("v0", "fishing-v0", {}), ("v1", "fishing-v1", {}), ("v2", "fishing-v2", {}),
                                           ("v1_params", "fishing-v1", {"r": 0.5, "K": 2.0, "init_state": 1.1}),
                                           ("v0_params", "fishing-v0", {"n_actions": 37, "r": 0.4})])
def test_bmsy_and_msy_reproduce_reference_values(gf, anchors, tag, env_id, kw):
    """models/policies.py:51-67 on the device.  The sweep is evaluated in float32 whatever the env's layout,
    like the reference's (a float32 grid times Python-float parameters): it reproduces the reference's values
    bit-for-bit, S = 0.4996 for the flat logistic maximum -- from the float32 N-env layout and from the
    float64 scalar protocol alike."""
    import torch
    from gym_fishing_amd.policies import BMSY, msy
    env = gf.make(env_id, sigma=0.0, num_envs=4, dtype=torch.float32, **kw)
    a = anchors["policy_" + tag]
    m = msy(env)
    if env_id == "fishing-v2":
        # v_exp_f32 != np.exp in the last bits and the maximum of the growth curve is flat:
        # the argmax may move a few cells of the 2e-4 grid; the MSY value itself barely changes
        assert abs(m.S - a["BMSY"]) <= 1e-3 and abs(m.msy - a["msy"]) <= 1e-6
    else:
        assert m.S == a["BMSY"] and BMSY(env) == a["BMSY"]
        assert m.msy == a["msy"]
    if env_id != "fishing-v2":
        env64 = gf.make(env_id, sigma=0.0, **kw)
        m64 = msy(env64)
        assert m64.S == a["BMSY"] and m64.msy == a["msy"]


from conftest import load_seeded_sims  # noqa: E402

SEEDED = load_seeded_sims()


@pytest.mark.parametrize("c", SEEDED, ids=[c["key"] for c in SEEDED])
def test_seeded_policy_flows_follow_the_reference(gf, c):
    """A user's script, unchanged: np.random.seed(7); env = make(id, sigma > 0); model = msy(env) or
    escapement(env); df = env.simulate(model, reps=2).  BMSY() and msy() consume the global stream like the
    reference (one normal per population_draw() of the logistic / tipping models, one per grid point for the
    zoo, whose growth functions ignore the sigma = 0 the reference sets on the env) and evaluate in float32 like
    it, so S, msy and every row of the table follow the reference: bit-for-bit for fishing-v0 / v1, within the
    transcendental tolerance for fishing-v2 and the zoo."""
    from gym_fishing_amd.policies import escapement, msy
    np.random.seed(c["seed"])
    env = gf.make(c["id"], **c["kwargs"])
    if c["id"] == "fishing-v4":
        # the fishing-v4 tables were taken with a float64 observation grid (tests/golden/make_golden.py: the float32 grid
        # times np.float64 K / r is float32 under the reference's NumPy 1.19 and float64 under NumPy 2 -- the float64 grid
        # is float64 under both): the flow -- K-then-r draws at the constructor and every reset(), BMSY's and msy's draws,
        # each row on its episode's (K, r) -- is pinned bit for bit
        env.observation_space.dtype = np.dtype(np.float64)
    model = (msy if c["policy"] == "msy" else escapement)(env)
    df = env.simulate(model, reps=c["reps"])
    got, want = df.to_numpy(dtype=np.float64), c["table"]
    exact = c["id"] in ("fishing-v0", "fishing-v1", "fishing-v4")
    if exact:
        assert model.S == c["S"] and (c["msy"] is None or model.msy == c["msy"])
        assert got.shape == want.shape and np.array_equal(got.view(np.int64), np.ascontiguousarray(want).view(np.int64))
    elif c["id"] == "fishing-v2":
        # float32 exp on the device vs np.exp: the flat maximum of the growth curve may move a few grid cells
        assert abs(model.S - c["S"]) <= 2e-3
        assert got.shape[1] == want.shape[1] and abs(got.shape[0] - want.shape[0]) <= 60
    else:
        # float32 log / exp on the device vs NumPy's: where the growth curve is flat (sigma = 0) the argmax may sit
        # a grid cell (2e-4) away; with sigma > 0 the sweep's noise decides it and S is the reference's exactly
        assert abs(model.S - c["S"]) <= 1e-3, (model.S, c["S"])
        same_S = model.S == c["S"]
        assert c["kwargs"].get("sigma", 0.0) == 0.0 or same_S
        tol = 1e-6 if same_S else 5e-3
        assert c["msy"] is None or abs(model.msy - c["msy"]) <= tol * max(1.0, abs(c["msy"]))
        assert got.shape == want.shape and np.allclose(got, want, rtol=10 * tol, atol=tol)


@pytest.mark.parametrize("c", SIMS, ids=[c["key"] for c in SIMS])
def test_env_simulate_reproduces_reference_tables(gf, c):
    """env.simulate(model) (base_fishing_env.py:100-101 -> shared_env.py:29-54) through the
    scalar protocol (Python loop + fp64 kernel) and through the N-env fused rollout; the
    policy constants are the reference's own (tests/golden anchors)."""
    import torch
    from gym_fishing_amd import policies
    kw = dict(r=c["r"], K=c["K"], init_state=c["x0"], sigma=0.0)
    if c["env_id"] == "fishing-v0":
        kw["n_actions"] = c["n_actions"]
    table = c["table"]

    def model_for(env):
        m = policies.msy(env) if c["policy"] == "msy" else policies.escapement(env)
        if c["policy"] == "msy":
            m.msy = c["param"]
        else:
            m.S = c["param"]
        m.kernel_policy = (m.kernel_policy[0], c["param"])
        return m
    is_v2 = c["env_id"] == "fishing-v2"
    # scalar protocol
    env = gf.make(c["env_id"], **kw)
    df = env.simulate(model_for(env), reps=2)
    got = df.to_numpy(dtype=np.float64)
    assert got.shape[0] == 2 * table.shape[0] and list(df.columns) == ["time", "state", "action", "reward", "rep"]
    for rep in range(2):
        blk = got[rep * table.shape[0]:(rep + 1) * table.shape[0]]
        assert (blk[:, 4] == rep).all()
        if is_v2:
            assert np.allclose(blk[:, :4], table, rtol=1e-8, atol=1e-10)   # unstable tipping run amplifies exp ulps
        else:
            assert np.array_equal(blk[:, :4].view(np.int64), table.view(np.int64)), c["key"]
    # N-env fused rollout, fp64: every env is one rep and reproduces the same table
    venv = gf.make(c["env_id"], num_envs=8, dtype=torch.float64, **kw)
    dfv = venv.simulate(model_for(venv))
    gv = dfv.to_numpy(dtype=np.float64)
    assert gv.shape[0] == 8 * table.shape[0]
    for rep in (0, 7):
        blk = gv[gv[:, 4] == rep]
        if is_v2:
            assert np.allclose(blk[:, :4], table, rtol=1e-8, atol=1e-10)   # unstable tipping run amplifies exp ulps
        else:
            assert np.array_equal(blk[:, :4].view(np.int64), np.ascontiguousarray(table).view(np.int64)), c["key"]


def test_simulate_with_generic_model_and_policyfn(gf):
    """A model without kernel_policy is driven through batched predict(); policyfn gives the
    reference's [state, action, rep] sweep (shared_env.py:82-102)."""
    import torch
    from gym_fishing_amd import policies

    class Wrapped:      # hides kernel_policy -> step-by-step path
        def __init__(self, inner):
            self.inner = inner

        def predict(self, obs, **kw):
            return self.inner.predict(obs, **kw)
    for env_id, dt in (("fishing-v1", torch.float64), ("fishing-v0", torch.float64), ("fishing-v2", torch.float32)):
        venv = gf.make(env_id, sigma=0.0, num_envs=8, dtype=dt)
        for pol in (policies.escapement(venv), policies.msy(venv)):
            a = venv.simulate(pol).to_numpy(dtype=np.float64)
            b = venv.simulate(Wrapped(pol)).to_numpy(dtype=np.float64)
            assert a.shape == b.shape and np.array_equal(a, b), (env_id, type(pol).__name__)
    env = gf.make("fishing-v1")
    esc = policies.escapement(env)
    assert esc.S == 0.49959999322891235          # the reference's float32 sweep, also from the float64 scalar env
    pf = env.policyfn(esc)
    assert list(pf.columns) == ["state", "action", "rep"] and len(pf) == 50
    st, ac = pf["state"].to_numpy(), pf["action"].to_numpy()
    assert np.allclose(ac, np.maximum(st - esc.S, 0.0), atol=1e-7)


def test_graph_replay_draws_fresh_noise_and_matches_eager(gf):
    """A captured step() replays with advancing noise keys (device-resident counter) and
    gives exactly what eager stepping gives; same for a captured 5-step step_many."""
    import torch
    from gym_fishing_amd.graphs import GraphedSteps
    n = 4096
    acts = torch.rand((5, n), device="cuda") * 2 - 1
    eager = gf.make("fishing-v1", sigma=0.2, num_envs=n, seed=21)
    eager.reset()
    graphed = gf.make("fishing-v1", sigma=0.2, num_envs=n, seed=21)
    graphed.reset()
    static = acts[0].clone()
    g = GraphedSteps(graphed, static)
    assert int(graphed._counter[0]) == 0 and torch.equal(graphed.state, eager.state)
    prev = None
    for k in range(7):
        static.copy_(acts[k % 5])
        obs, rew, done, _ = g.replay()
        eo, er, ed, _ = eager.step(acts[k % 5])
        assert torch.equal(obs, eo) and torch.equal(rew, er) and torch.equal(done, ed), k
        if prev is not None:
            assert not torch.equal(prev, rew)
        prev = rew.clone()
    assert int(graphed._counter[0]) == 7
    # multi-step graph
    g2 = GraphedSteps(graphed, acts)
    for _ in range(3):
        g2.replay()
        eager.step_many(acts, 5)
    assert torch.equal(graphed.state, eager.state) and int(graphed._counter[0]) == 7 + 15
    # eager calls on a graph-mode env keep the same stream of noise
    graphed.step(acts[1])
    eager.step(acts[1])
    assert torch.equal(graphed.state, eager.state)


def test_graph_replay_at_a_zigzag_size_takes_the_walk_direction_from_the_device_counter(gf):
    """From ~100 MB per step odd steps walk the tiles backwards (in groups of eight).  A captured launch has frozen
    arguments, so in graph-replay mode the step's parity -- like its noise key -- comes from the device-resident counter,
    which the one-tile forms then read BEFORE their loads.  N = 2^22 with returns (138 MB per step), three replays of a
    3-step graph (odd length: the parity of a replay's first step alternates) against eager stepping: bit for bit."""
    import torch
    from gym_fishing_amd.graphs import GraphedSteps
    n = 1 << 22
    acts = torch.rand((3, n), device="cuda") * 1.4 - 1.2
    mk = lambda: gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=9, Tmax=5, track_returns=True)  # noqa: E731
    eager, graphed = mk(), mk()
    eager.reset()
    graphed.reset()
    g = GraphedSteps(graphed, acts)
    assert graphed.step_kernel_name(acts[0]) == "fishing::step_kernel_lean<float, 1, 12294, 4>"
    for rnd in range(3):
        g.replay()
        eager.step_many(acts, 3)
        assert torch.equal(graphed.state, eager.state) and torch.equal(graphed._t, eager._t), rnd
        assert torch.equal(graphed._ep_return, eager._ep_return), rnd
    a, b = graphed.episode_stats(), eager.episode_stats()
    assert a["n_episodes"] == b["n_episodes"] > n and abs(a["sum_return"] - b["sum_return"]) <= 1e-9 * abs(b["sum_return"])
    del eager, graphed, g
    torch.cuda.empty_cache()


def test_graph_replay_of_fishing_v4_survives_a_reset_after_the_capture(gf):
    """fishing-v4 re-derives (K, r) from the origin of the last reset() of all envs.  A captured launch freezes its
    arguments, so in graph-replay mode the origin lives next to the step counter in device memory (FishingBuffers.counter
    = u64[3], ABI 4) and reset() rewrites it: the env STAYS in the derived mode (no r / K arrays), and replays before AND
    after a later reset() equal an eager env bit for bit -- state, and the (K, r) in force."""
    import torch
    from gym_fishing_amd.graphs import GraphedSteps
    n = 2048 + 24
    acts = torch.rand((4, n), device="cuda") * 1.4 - 1.2
    mk = lambda: gf.make("fishing-v4", sigma=0.05, sigma_p=0.2, num_envs=n, seed=5, Tmax=7)  # noqa: E731
    eager, graphed = mk(), mk()
    eager.reset()
    graphed.reset()
    assert eager._derived and graphed._derived
    g = GraphedSteps(graphed, acts)                  # 4 steps per replay
    assert graphed._derived and graphed._K_arr is None and graphed._counter.numel() == 4
    assert graphed.step_kernel_name(acts[0]) == eager.step_kernel_name(acts[0]) == "fishing::step_kernel_lean<float, 4, 8450, 4>"
    for rnd in range(3):
        for _ in range(3):
            g.replay()
            eager.step_many(acts, 4)
        assert torch.equal(graphed.state, eager.state), rnd
        assert torch.equal(graphed.K, eager.K) and torch.equal(graphed.r, eager.r), rnd
        if rnd < 2:                                  # a reset of all envs between replays: a new origin for both
            graphed.reset()
            eager.reset()
            assert eager._derived and graphed._derived
            # (in graph-replay mode reset() moves the origin on the device; the host's copy is read back on demand)
            assert graphed._counter.tolist() == [graphed._step_count, *graphed._host_origin(), eager._reset_count]
            assert graphed._host_origin() == eager._host_origin()
            assert torch.equal(graphed.K, eager.K)
    # a checkpoint of the graph-mode env resumes in an env that never saw the capture
    sd = graphed.state_dict()
    assert sd["format"] == 3 and sd["_counter"].numel() == 4 and sd["reset_count"] == eager._reset_count
    fresh = mk()
    fresh.load_state_dict(sd)
    fresh.step_many(acts, 4)
    g.replay()
    assert torch.equal(fresh.state, graphed.state) and torch.equal(fresh.K, graphed.K)
    # ... and a fishing-v4 state without the parameter-stream tag (format 1) is refused before anything changes
    old = {k: v for k, v in sd.items() if k not in ("format", "v4_param_stream")}
    before = fresh.state.clone()
    with pytest.raises(ValueError, match="parameter stream"):
        fresh.load_state_dict(old)
    assert torch.equal(fresh.state, before) and fresh._derived


def test_graph_replay_follows_a_parameter_mode_switch_after_the_capture(gf):
    """A captured launch has frozen fishing-v4's parameter mode and the addresses of its r / K streams.  GraphedSteps
    re-captures when the env's launch signature moved, and the env never frees the arrays a capture may still write:
    (1) capture in the derived mode, then a masked reset() (per-env episode origins from there on) -> replays equal an
    eager env that did the same; (2) seed() first (stored arrays), capture, then a full reset() (back to derived) ->
    replays equal eager, and the arrays the first graph wrote through are still the env's own."""
    import torch
    from gym_fishing_amd.graphs import GraphedSteps
    n = 3072
    acts = torch.rand((3, n), device="cuda") * 1.4 - 1.2
    mask = torch.rand(n, device="cuda") < 0.3
    mk = lambda: gf.make("fishing-v4", sigma=0.05, sigma_p=0.2, num_envs=n, seed=5, Tmax=6)  # noqa: E731
    # (1)
    eager, graphed = mk(), mk()
    eager.reset()
    graphed.reset()
    g = GraphedSteps(graphed, acts)
    for _ in range(2):
        g.replay()
        eager.step_many(acts, 3)
    assert g.recaptures == 0 and torch.equal(graphed.state, eager.state)
    graphed.reset(mask)
    eager.reset(mask)
    sig = graphed.launch_signature()
    for _ in range(3):
        g.replay()
        eager.step_many(acts, 3)
        assert torch.equal(graphed.state, eager.state) and torch.equal(graphed.K, eager.K) and torch.equal(graphed.r, eager.r)
    assert g.recaptures == 1 and graphed.launch_signature() == sig
    # a scalar of the parameter struct changed behind the capture's back: honoured by the next replay
    graphed.Tmax = eager.Tmax = 3
    g.replay()
    eager.step_many(acts, 3)
    assert g.recaptures == 2 and torch.equal(graphed.state, eager.state) and torch.equal(graphed._t, eager._t)
    # (2)
    eager, graphed = mk(), mk()
    for e in (eager, graphed):
        e.reset()
        e.seed(77)                          # the parameters in force were drawn under the old seed: stored arrays
        assert not e._derived
    g = GraphedSteps(graphed, acts)
    K_ptr = graphed._K_arr.data_ptr()
    g.replay()
    eager.step_many(acts, 3)
    assert torch.equal(graphed.state, eager.state) and torch.equal(graphed.K, eager.K)
    graphed.reset()
    eager.reset()
    assert graphed._derived and graphed._K_arr is None and graphed._K_store.data_ptr() == K_ptr      # kept, not freed
    junk = [torch.full((n,), 7.0, device="cuda") for _ in range(8)]          # whatever would have reused a freed block
    for _ in range(3):
        g.replay()
        eager.step_many(acts, 3)
        assert torch.equal(graphed.state, eager.state) and torch.equal(graphed.K, eager.K)
    assert g.recaptures == 1 and all(bool((j == 7.0).all()) for j in junk)


def test_raw_graph_replays_keep_the_step_count_the_env_acts_on(gf):
    """A caller's own torch.cuda.CUDAGraph of env.step() advances only the device-resident counter.  Everything that
    needs the step count afterwards -- reset()'s episode origin, env.K / env.r of the derived mode, state_dict() -- reads
    it from the device: the run equals eager stepping bit for bit, the checkpoint resumes."""
    import torch
    n = 2048
    a = torch.rand(n, device="cuda") * 1.4 - 1.2
    mk = lambda: gf.make("fishing-v4", sigma=0.05, sigma_p=0.2, num_envs=n, seed=9, Tmax=5)  # noqa: E731
    eager, graphed = mk(), mk()
    eager.reset()
    graphed.reset()
    graphed.enable_graph_replay()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graphed.step(a)                     # warm-up outside the capture (torch's rule); eager takes the same step
    torch.cuda.current_stream().wait_stream(side)
    eager.step(a)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        graphed.step(a)
    host_before = graphed._step_count
    for _ in range(9):
        graph.replay()
        eager.step(a)
    assert graphed._step_count == host_before            # the host never saw the replays ...
    assert torch.equal(graphed.K, eager.K) and torch.equal(graphed.r, eager.r)      # ... env.K asks the device
    assert graphed._step_count == eager._step_count == 10
    sd = graphed.state_dict()
    assert sd["step_count"] == 10
    graph.replay()
    eager.step(a)
    graphed.reset()
    eager.reset()
    assert graphed._host_origin() == eager._host_origin() == (11, 2)
    for _ in range(4):
        graph.replay()
        eager.step(a)
    assert torch.equal(graphed.state, eager.state) and torch.equal(graphed.K, eager.K)
    fresh = mk()
    fresh.load_state_dict(sd)
    again = mk()
    again.reset()
    for _ in range(10):
        again.step(a)
    assert torch.equal(fresh.state, again.state) and torch.equal(fresh.K, again.K)


def test_reset_in_graph_replay_mode_is_capturable_and_never_reads_the_counter_back(gf):
    """In graph-replay mode a reset() of all envs dates fishing-v4's episodes from the DEVICE's step counter, copied device
    word to device word on the current stream: no host read (reset() does not wait for the GPU on the launch-bound path
    this mode exists for) and therefore legal inside a caller's own stream capture.  A captured [reset(), 3 steps] replayed
    three times equals an eager env that resets and steps three times, bit for bit -- state and the (K, r) in force: the reset
    counter is a device word too (FISHING_FLAG_RESET_COUNTER_ON_DEVICE), read and bumped by the reset itself, so every replay
    draws the parameters of ITS reset -- not the captured one's over and over."""
    import torch
    n = 2048
    acts = torch.rand((3, n), device="cuda") * 1.4 - 1.2
    mk = lambda: gf.make("fishing-v4", sigma=0.05, sigma_p=0.2, num_envs=n, seed=21, Tmax=5)  # noqa: E731
    eager, graphed = mk(), mk()
    eager.reset()
    graphed.reset()
    graphed.enable_graph_replay()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graphed.step_many(acts, 3)          # warm-up outside the capture (torch's rule); eager takes the same steps
    torch.cuda.current_stream().wait_stream(side)
    eager.step_many(acts, 3)
    rc = graphed._reset_count
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        graphed.reset()                     # a host read of the device counter would be a synchronisation: illegal in a capture
        graphed.step_many(acts, 3)
    assert graphed._derived
    for rnd in range(3):
        graph.replay()
        eager.reset()
        eager.step_many(acts, 3)
        assert torch.equal(graphed.state, eager.state) and torch.equal(graphed._t, eager._t), rnd
        assert torch.equal(graphed.K, eager.K) and torch.equal(graphed.r, eager.r), rnd
        if rnd:         # the episodes of two replays start from different draws
            assert not torch.equal(graphed.K, K_prev)
        K_prev = graphed.K.clone()
    assert graphed._host_origin() == eager._host_origin() == (3 + 2 * 3, rc + 2)
    assert graphed._current_reset_count() == eager._reset_count == rc + 3
    # ... and fishing-v11's per-episode model choice follows the same word
    mk11 = lambda: gf.make("fishing-v11", num_envs=n, seed=21, Tmax=5)  # noqa: E731
    e11, g11 = mk11(), mk11()
    g11.enable_graph_replay()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g11.reset()
    torch.cuda.current_stream().wait_stream(side)
    e11.reset()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g11.reset()
    seen = []
    for rnd in range(3):
        graph.replay()
        e11.reset()
        assert torch.equal(g11.model_idx, e11.model_idx), rnd
        seen.append(g11.model_idx.clone())
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])


def test_load_state_dict_into_a_graph_replay_env_moves_the_device_counter_and_refuses_stray_stamps(gf):
    """(1) A checkpoint of a host-counter env carries no `_counter`; loaded into an env in graph-replay mode, the device
    word IS the step count from there on and must take the checkpoint's value -- the kernels draw from it, reset() dates
    episodes from it.  (2) Origin stamps belong to fishing-v4's derived mode: a state that carries them without that mode is
    refused before anything changes."""
    import torch
    n = 2048
    acts = torch.rand((4, n), device="cuda") * 1.4 - 1.2
    mk = lambda: gf.make("fishing-v4", sigma=0.05, sigma_p=0.2, num_envs=n, seed=3, Tmax=6)  # noqa: E731
    a = mk()
    a.reset()
    a.step_many(acts, 5)
    sd = a.state_dict()
    assert "_counter" not in sd and sd["step_count"] == 5
    b = mk()
    b.reset()
    b.enable_graph_replay()
    b.step_many(acts, 2)
    b.load_state_dict(sd)
    assert b._counter.tolist() == [5, *sd["v4_origin"], sd["reset_count"]]
    a.step_many(acts, 4)
    b.step_many(acts, 4)
    assert torch.equal(a.state, b.state) and torch.equal(a.K, b.K) and b._current_step_count() == 9
    b.reset()
    a.reset()
    a.step_many(acts, 3)
    b.step_many(acts, 3)
    assert torch.equal(a.state, b.state) and torch.equal(a.K, b.K)
    # (2)
    a.reset(torch.arange(n, device="cuda") % 3 == 0)
    sd = a.state_dict()
    assert "_stamp" in sd and sd["v4_derived"]
    before = (b.state.clone(), b._t.clone(), b._derived, b._seed, b._current_step_count())
    for bad in (dict(sd, v4_derived=False), dict(sd, _stamp=sd["_stamp"][:n // 2].clone())):
        with pytest.raises(ValueError, match="_stamp"):
            b.load_state_dict(bad)
        assert torch.equal(b.state, before[0]) and torch.equal(b._t, before[1])
        assert (b._derived, b._seed, b._current_step_count()) == before[2:]
    with pytest.raises(ValueError, match="_stamp"):
        gf.make("fishing-v1", sigma=0.1, num_envs=n).load_state_dict(dict(sd, v4_derived=True))
    b.load_state_dict(sd)               # ... while the consistent state loads
    a.step_many(acts, 3)
    b.step_many(acts, 3)
    assert torch.equal(a.state, b.state) and torch.equal(a.K, b.K)


def test_v4_state_without_the_stream_tag_loads_where_it_can(gf):
    """load_state_dict refuses a fishing-v4 state written under another parameter stream only where the stream decides
    what the state means: the derived mode always; stored arrays unless strict=False (the (K, r) in force are in the
    state; a warning says redraws will differ); never for rng='numpy' envs, which do not use that stream."""
    import torch
    import warnings
    n = 1024
    env = gf.make("fishing-v4", num_envs=n, seed=1, derived_params=False)
    env.reset()
    env.step(torch.zeros(n, device="cuda") - 0.9)
    sd = {k: v for k, v in env.state_dict().items() if k not in ("format", "v4_param_stream")}
    other = gf.make("fishing-v4", num_envs=n, seed=1, derived_params=False)
    with pytest.raises(ValueError, match="strict=False"):
        other.load_state_dict(sd)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        other.load_state_dict(sd, strict=False)
    assert len(w) == 1 and torch.equal(other.K, env.K) and torch.equal(other.state, env.state)
    one = gf.make("fishing-v4")                         # scalar protocol, rng="numpy"
    one.reset()
    sd1 = {k: v for k, v in one.state_dict().items() if k not in ("format", "v4_param_stream")}
    two = gf.make("fishing-v4")
    two.load_state_dict(sd1)
    assert two.K == one.K and two.r == one.r


def test_v11_state_of_another_model_stream_is_refused_or_warned_about(gf):
    """fishing-v11 redraws its growth function at every reset from a generator that is part of the state's meaning
    (`v11_model_stream`: one Philox2x32-10 block per env quad since ABI 8; a Philox4x32-10 word per env before).  A format-2
    state carries no such tag: strict loading refuses it before anything changes, strict=False loads the models in force
    with a warning; the current format resumes bit for bit; an rng='numpy' env, which never uses that stream, loads either."""
    import torch
    import warnings
    n = 2048
    acts = torch.rand((3, n), device="cuda") * 1.4 - 1.2
    mk = lambda **kw: gf.make("fishing-v11", num_envs=n, seed=4, Tmax=3, **kw)  # noqa: E731
    a = mk()
    a.reset()
    a.step_many(acts, 3)
    sd = a.state_dict()
    assert sd["format"] == 3 and sd["v11_model_stream"] == "philox2x32-10/quad:u16"
    b = mk()
    b.load_state_dict(sd)
    a.step_many(acts, 9)
    b.step_many(acts, 9)
    assert torch.equal(a.state, b.state) and torch.equal(a.model_idx, b.model_idx)
    old = dict({k: v for k, v in sd.items() if k != "v11_model_stream"}, format=2)
    c = mk()
    c.reset()
    before = (c.state.clone(), c.model_idx.clone())
    with pytest.raises(ValueError, match="model stream"):
        c.load_state_dict(old)
    assert torch.equal(c.state, before[0]) and torch.equal(c.model_idx, before[1])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        c.load_state_dict(old, strict=False)
    assert len(w) == 1 and "model stream" in str(w[0].message) and torch.equal(c.model_idx, sd["_model_idx"])
    one = gf.make("fishing-v11")            # scalar protocol, rng="numpy": np.random.choice draws the model
    one.reset()
    sd1 = {k: v for k, v in one.state_dict().items() if k != "v11_model_stream"}
    assert gf.make("fishing-v11").load_state_dict(sd1).model == one.model


@pytest.mark.parametrize("env_id", ["fishing-v1", "fishing-v0", "fishing-v4"])
def test_gymnasium_api_splits_done_into_terminated_and_truncated(gf, env_id):
    """make(id, api="gymnasium"): reset() -> (obs, info), step() -> 5-tuple with truncated = years_passed > Tmax and
    terminated = fish_population <= 0 (the two arms of base_fishing_env.py:76-79), same observations / rewards as the
    4-tuple env underneath -- one env (host floats / bools) and N auto-resetting envs (device tensors)."""
    import torch
    # one env: harvest everything at once -> the stock is gone on step 1 (terminated, not truncated); a patient policy
    # runs into the horizon (truncated, not terminated)
    e5, e4 = gf.make(env_id, api="gymnasium", Tmax=5), gf.make(env_id, Tmax=5)
    obs, info = e5.reset(seed=3)
    # (gymnasium's Box.contains looks at the dtype: this flavour hands the scalar protocol's observation out in the Box's float32)
    assert obs.dtype == np.float32 and np.array_equal(obs, e4.reset(seed=3).astype(np.float32)) and info == {}
    take_all = e4.get_action(10.0)
    o5, r5, term, trunc, _ = e5.step(take_all)
    o4, r4, d4, _ = e4.step(take_all)
    assert o5.dtype == np.float32 and np.array_equal(o5, o4.astype(np.float32)) and r5 == r4 and d4 is True and term is True and trunc is False
    assert e5.env.state.dtype == np.float64                     # the wrapped 4-tuple env keeps the reference's float64
    e5.reset()
    leave = e4.get_action(0.0)
    flags = [e5.step(leave)[2:4] for _ in range(6)]
    assert flags[:5] == [(False, False)] * 5 and flags[5] == (False, True)
    assert e5.Tmax == 5 and e5.unwrapped is e5 and e5.action_space is e5.env.action_space
    # attribute WRITES reach the wrapped env (the reference's callers set env.unwrapped.Tmax / .sigma / .K), not a shadow
    e5.unwrapped.Tmax = 2
    e5.sigma = 0.25
    assert e5.env.Tmax == 2 and e5.env.sigma == 0.25 and "Tmax" not in e5.__dict__ and "sigma" not in e5.__dict__
    e5.reset()
    assert [e5.step(leave)[3] for _ in range(3)] == [False, False, True]          # truncated by the NEW horizon
    e5.render_mode = "human"
    assert e5.__dict__["render_mode"] == "human" and not hasattr(e5.env, "render_mode")
    # N envs with fused auto-reset: both kinds of ending in one batch, and the 4-tuple env's done is their union
    n = 512
    kw = dict(num_envs=n, seed=11, sigma=0.05, Tmax=4)
    v5, v4 = gf.make(env_id, api="gymnasium", **kw), gf.make(env_id, record_terminal_obs=True, **kw)
    o, info = v5.reset()
    assert torch.equal(o, v4.reset()) and info == {}
    g = torch.Generator(device="cuda").manual_seed(1)
    n_term = n_trunc = 0
    for s in range(12):
        if env_id == "fishing-v0":
            a = torch.where(torch.rand(n, device="cuda", generator=g) < 0.15, 99, 0).to(torch.int32)
        else:
            a = torch.where(torch.rand(n, device="cuda", generator=g) < 0.15, 1.0, -1.0)
        o5, r5, term, trunc, i5 = v5.step(a)
        o4, r4, d4, i4 = v4.step(a)
        assert torch.equal(o5, o4) and torch.equal(r5, r4) and torch.equal(term | trunc, d4.bool())
        pop_gone = i4["terminal_observation"].reshape(-1) <= -1.0
        assert torch.equal(term, d4.bool() & pop_gone)
        n_term += int(term.sum())
        n_trunc += int(trunc.sum())
    assert n_term > 50 and n_trunc > 50


def test_bmsy_does_not_disturb_the_env_noise_level(gf):
    """BMSY()/msy() evaluate population_draw at sigma = 0 (models/policies.py:61-65) and must
    leave the env's own sigma in force afterwards."""
    import torch
    from gym_fishing_amd import policies
    a = gf.make("fishing-v1", sigma=0.3, num_envs=256, seed=2)
    b = gf.make("fishing-v1", sigma=0.3, num_envs=256, seed=2)
    policies.msy(a)
    a.reset()
    b.reset()
    act = torch.full((256,), -0.9, device="cuda")
    oa, _, _, _ = a.step(act)
    ob, _, _, _ = b.step(act)
    assert torch.equal(oa, ob) and float(oa.std()) > 0.01


@pytest.mark.parametrize("env_id,kw", [("fishing-v0", {}), ("fishing-v1", {}), ("fishing-v5", {"sigma": 0}),
                                       ("fishing-v6", {"sigma": 0}), ("fishing-v7", {"sigma": 0}),
                                       ("fishing-v8", {"sigma": 0}), ("fishing-v9", {"sigma": 0}),
                                       ("fishing-v2", {"sigma": 0, "init_state": 0.75}),
                                       ("fishing-v10", {"sigma": 0, "alpha": -0.007}), ("fishing-v11", {})])
def test_reference_test_suite_flow(gf, tmp_path, env_id, kw):
    """The body of every test in the reference's tests/test-envs.py (:11-139), minus SB3's
    check_env (not installed): make -> msy -> simulate -> plot -> escapement -> simulate -> plot,
    plus the API-conformance facts check_env would assert."""
    from gym_fishing_amd.policies import escapement, msy, user_action
    env = gf.make(env_id, **kw)
    obs = env.reset()
    assert obs.shape == env.observation_space.shape and isinstance(obs, np.ndarray)
    a = env.action_space.sample()
    o, r, d, info = env.step(a)
    assert o.shape == (1,) and isinstance(r, float) and isinstance(d, bool) and isinstance(info, dict)
    user_action(env)                                       # constructed, never prompted (as in the reference)
    reps = 10 if env_id == "fishing-v11" else 1           # (test-envs.py:136,139: reps=10 for ModelUncertainty)
    np.random.seed(0)                                      # (:8, :94, :130)
    model, model2 = msy(env), escapement(env)              # fishing-v11: BMSY() under the model in force, as the reference's
    for tag, m in (("msy", model), ("escapement", model2)):
        df = env.simulate(m, reps=reps)
        assert list(df.columns) == ["time", "state", "action", "reward", "rep"] and 1 <= len(df) <= 100 * reps
        assert set(df["rep"]) == set(range(reps)) and float(df["reward"].min()) >= 0.0
        out = env.plot(df, str(tmp_path / ("%s_%s.png" % (env_id, tag))))
        assert (tmp_path / ("%s_%s.png" % (env_id, tag))).stat().st_size > 1000 and out.endswith(".png")
    pf = env.policyfn(model2)
    env.plot_policy(pf, str(tmp_path / "policy.png"))
    assert (tmp_path / "policy.png").exists()


def test_c_api_demo_runs_without_python_side_state(gf, tmp_path):
    """examples/c_api_demo.cpp: the library driven from plain C++ (hipMalloc + a stream)."""
    import shutil
    import subprocess
    from conftest import ROOT
    import os
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not on this box")
    exe = tmp_path / "c_api_demo"
    libdir = os.path.join(ROOT, "gym_fishing_amd", "_lib")
    subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "c_api_demo.cpp"), "-L", libdir, "-lfishing_hip",
                    "-Wl,-rpath," + libdir, "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True, timeout=120)
    assert out.returncode == 0, out.stdout
    assert "mean return 6.312500" in out.stdout and "episodes 1048576" in out.stdout


def test_state_dict_resume_is_bit_exact(gf):
    """Checkpoint / resume: a restored env continues exactly like the one that kept running
    (streams, counters and seed are all that is needed; the noise is counter-based)."""
    import torch
    n = 3000
    acts = torch.rand((6, n), device="cuda") * 2 - 1
    a = gf.make("fishing-v4", sigma=0.1, num_envs=n, seed=8, Tmax=5, track_returns=True)
    a.reset()
    a.step_many(acts, 9)
    sd = a.state_dict()
    a.step_many(acts, 11)
    a.rollout(7, policy="random")
    b = gf.make("fishing-v4", sigma=0.1, num_envs=n, seed=123, Tmax=5, track_returns=True)
    b.load_state_dict(sd)
    b.step_many(acts, 11)
    b.rollout(7, policy="random")
    assert torch.equal(a.state, b.state) and torch.equal(a.K, b.K) and torch.equal(a.years_passed, b.years_passed)
    assert a.episode_stats() == b.episode_stats()
    s = gf.make("fishing-v1", sigma=0.3, seed=4)
    s.reset()
    for _ in range(5):
        s.step(np.array([-0.9], dtype=np.float32))
    sd = s.state_dict()
    want = [s.step(np.array([-0.8], dtype=np.float32))[0][0] for _ in range(4)]
    s2 = gf.make("fishing-v1", sigma=0.3, seed=999).load_state_dict(sd)
    got = [s2.step(np.array([-0.8], dtype=np.float32))[0][0] for _ in range(4)]
    assert want == got


def test_load_state_dict_checks_sizes_before_it_changes_anything(gf):
    """A state of another batch size is refused before the first field changes (the env keeps running as it was); a
    return_partials buffer shorter than the env's -- written before the buffer grew with the batch (ABI 5) -- loads
    into the first slots, the rest zero: the record is the sum over slots."""
    import torch
    a = gf.make("fishing-v1", sigma=0.1, num_envs=4096, seed=8, Tmax=5, track_returns=True)
    a.reset()
    acts = torch.rand((4, 4096), device="cuda") * 2 - 1
    a.step_many(acts, 12)
    sd = a.state_dict()
    b = gf.make("fishing-v1", sigma=0.1, num_envs=2048, seed=8, Tmax=5, track_returns=True)
    b.reset()
    before = (b.state.clone(), b._step_count, b._seed)
    with pytest.raises(ValueError, match="elements"):
        b.load_state_dict(sd)
    assert torch.equal(b.state, before[0]) and (b._step_count, b._seed) == before[1:]
    short = dict(sd)
    short["_partials"] = sd["_partials"][:4 * 1024].clone()       # (only the first four slots of a 4096-env batch are ever written)
    c = gf.make("fishing-v1", sigma=0.1, num_envs=4096, seed=1, Tmax=5, track_returns=True)
    c._partials.fill_(7.0)
    c.load_state_dict(short)
    assert c.episode_stats() == a.episode_stats()


def test_bench_contract_json_line(gf):
    """bench.py prints exactly ONE JSON line with the contract's keys (small run, no CPU baseline)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    import os
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "8",
                          "--n-envs", str(1 << 18), "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 8 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    # (N = 2^18: the streams sit in the Infinity Cache, and the record says so in the field itself)
    assert r["bound"] == ("infinity-cache/hbm" if r["cache_resident"] else "hbm") and r["cache_resident"] is True
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert abs(d["value"] - (1 << 18) * 40 / (d["ms_per_step"] * 40 / 1e3)) / d["value"] < 1e-9


@pytest.mark.parametrize("script", ["const_escapement.py", "random_rollout.py", "vec_env_numpy.py", "parameter_uncertainty.py"])
def test_examples_run(gf, script):
    import subprocess
    import sys
    from conftest import ROOT
    import os
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1500:]
    if script == "const_escapement.py":
        assert "scalar protocol: 100 steps, return 7.675000" in out.stdout and "mean_return" in out.stdout
    elif script == "vec_env_numpy.py":
        assert "episodes 768, mean reward" in out.stdout          # 256 envs x three 101-step episodes
    elif script == "parameter_uncertainty.py":
        assert "own level per env" in out.stdout and "escapement levels: min" in out.stdout
    else:
        assert "env-steps/s" in out.stdout


def test_numpy_vec_env_adapter_follows_the_sb3_protocol(gf):
    """gym_fishing_amd.vec_env.FishingVecEnv: what SB3's DummyVecEnv hands its callers --
    float32 obs [N, 1], float32 rewards [N], bool dones [N], a list of N info dicts with the terminal
    observation of every env that just finished, the returned obs already reset -- and the calls the
    reference's own VecEnv helpers make (shared_env.py:15-26,57-79)."""
    import torch
    from gym_fishing_amd.vec_env import FishingVecEnv, make_vec_env
    n, Tmax = 64, 5
    venv = make_vec_env("fishing-v1", n, sigma=0.1, seed=3, Tmax=Tmax)
    twin = gf.make("fishing-v1", num_envs=n, sigma=0.1, seed=3, Tmax=Tmax, record_terminal_obs=True)
    obs = venv.reset()
    twin.reset()
    assert isinstance(obs, np.ndarray) and obs.shape == (n, 1) and obs.dtype == np.float32 and (obs == -0.25).all()
    assert venv.num_envs == n and venv.get_attr("Tmax") == [Tmax] * n and venv.get_attr("Tmax", indices=3) == [Tmax]
    assert venv.env_is_wrapped(object) == [False] * n and venv.action_space.low[0] == -1
    rng = np.random.default_rng(0)
    finished = 0
    for s in range(14):
        a = rng.uniform(-1, -0.5, (n, 1)).astype(np.float32)
        venv.step_async(a)
        obs, rew, done, infos = venv.step_wait()
        o2, r2, d2, i2 = twin.step(torch.as_tensor(a))
        assert obs.shape == (n, 1) and obs.dtype == np.float32 and rew.shape == (n,) and rew.dtype == np.float32
        assert done.shape == (n,) and done.dtype == np.bool_ and isinstance(infos, list) and len(infos) == n
        assert np.array_equal(obs, o2.cpu().numpy()) and np.array_equal(rew, r2.cpu().numpy())
        assert np.array_equal(done, d2.cpu().numpy())
        term = i2["terminal_observation"].cpu().numpy()
        for i in range(n):
            if done[i]:
                finished += 1
                assert infos[i]["terminal_observation"].shape == (1,) and infos[i]["terminal_observation"][0] == term[i, 0]
                assert obs[i, 0] == np.float32(-0.25)                 # already reset
            else:
                assert infos[i] == {}
        # the reference's df_entry_vec idiom (shared_env.py:15-26)
        pop = venv.env_method("get_fish_population", (obs[7],), indices=7)[0][0]
        assert np.isclose(pop, (obs[7, 0] + 1.0) * 1.0)
    assert finished >= n                                              # Tmax = 5: every env finished at least once
    # discrete actions arrive as an int array of shape [N]
    v0 = make_vec_env("fishing-v0", 8, sigma=0.0, seed=1)
    v0.reset()
    obs, rew, done, infos = v0.step(np.full(8, 10))
    assert np.allclose(rew, 0.1) and not done.any() and obs.dtype == np.float32
    # above 8192 envs the adapter keeps the state in HBM and downloads it (two async copies + one sync)
    assert venv.env._host_mapped and venv.env._obs.device.type == "cpu" and venv.env._obs.is_pinned()
    big = make_vec_env("fishing-v1", 9216, sigma=0.1, seed=3, Tmax=Tmax)
    tbig = gf.make("fishing-v1", num_envs=9216, sigma=0.1, seed=3, Tmax=Tmax, record_terminal_obs=True)
    assert not big.env._host_mapped and big.env._obs.device.type == "cuda"
    ob, tb = big.reset(), tbig.reset()
    ab = rng.uniform(-1, -0.5, (9216, 1)).astype(np.float32)
    for _ in range(7):
        ob, rb, db, ib = big.step(ab)
        o2, r2, d2, i2 = tbig.step(torch.as_tensor(ab))
        assert np.array_equal(ob, o2.cpu().numpy()) and np.array_equal(rb, r2.cpu().numpy()) and np.array_equal(db, d2.cpu().numpy())
        for j in np.flatnonzero(db)[:3]:
            assert ib[j]["terminal_observation"][0] == i2["terminal_observation"].cpu().numpy()[j, 0]
    # the fp64 parity layout and per-env parameters behind the same float32 NumPy boundary
    v4 = make_vec_env("fishing-v4", 16, sigma=0.05, seed=2, dtype=torch.float64)
    t4 = gf.make("fishing-v4", num_envs=16, sigma=0.05, seed=2, dtype=torch.float64, record_terminal_obs=True)
    o = v4.reset()
    t4.reset()
    assert o.shape == (16, 1) and o.dtype == np.float32 and (o == np.float32(0.75)).all()      # quirk a9: x0 un-normalised
    a4 = np.full((16, 1), -0.8, np.float32)
    for _ in range(3):
        o, r, d, inf = v4.step(a4)
        o2, r2, d2, _ = t4.step(torch.as_tensor(a4))
        assert np.array_equal(o, o2.cpu().numpy().astype(np.float32)) and np.array_equal(r, r2.cpu().numpy().astype(np.float32))
        assert np.array_equal(d, d2.cpu().numpy())
    with pytest.raises(ValueError):
        FishingVecEnv(gf.make("fishing-v1", num_envs=4))              # no terminal-obs record
    with pytest.raises(ValueError):
        FishingVecEnv(gf.make("fishing-v1"))                          # scalar protocol
    assert venv.seed(5) == [5 + i for i in range(n)]
    venv.close()


def test_two_envs_on_concurrent_streams_match_sequential_runs(gf):
    """The library holds no state and only enqueues on the caller's stream (re-entrant per
    stream): two envs stepped concurrently on two streams give exactly what each gives alone."""
    import torch
    n, K = 1 << 16, 40

    def build(seed):
        env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=seed, track_returns=True)
        g = torch.Generator(device="cuda").manual_seed(seed)
        acts = (torch.rand((4, n), device="cuda", generator=g) * 2 - 1).float()
        env.reset()
        return env, acts
    alone = []
    for seed in (1, 2):
        env, acts = build(seed)
        env.step_many(acts, K)
        torch.cuda.synchronize()
        alone.append((env._obs.clone(), env._t.clone(), env.episode_stats()))
    pairs = [build(1), build(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for k in range(K):                       # interleave the enqueues, one step at a time
        for (env, acts), s in zip(pairs, streams):
            with torch.cuda.stream(s):
                env.step(acts[k % 4])
    torch.cuda.synchronize()
    for (env, _), (obs, t, stats), s in zip(pairs, alone, streams):
        with torch.cuda.stream(s):
            got = env.episode_stats()
        assert torch.equal(env._obs, obs) and torch.equal(env._t, t) and got == stats


@pytest.mark.parametrize("env_id", ["fishing-v0", "fishing-v1", "fishing-v2", "fishing-v4", "fishing-v5", "fishing-v6",
                                    "fishing-v7", "fishing-v8", "fishing-v9", "fishing-v10", "fishing-v11"])
def test_long_run_invariants_every_id(gf, env_id):
    """Size-independent invariants over a long auto-resetting run of every id (N = 4096 + 3, ragged; 600 steps
    of in-kernel noise and random actions): the observation never drops below -1 (population >= 0), stays
    finite, the year counter stays in [0, Tmax], rewards are >= 0 and never exceed the stock, and the
    episodic-return record counts exactly the dones it was shown."""
    import torch
    n, T, Tmax = 4096 + 3, 600, 25
    kw = dict(num_envs=n, seed=11, Tmax=Tmax, track_returns=True)
    if env_id != "fishing-v11":
        kw["sigma"] = 0.1
    env = gf.make(env_id, **kw)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(3)
    dones = 0
    K = float(env.params["K"]) if env_id not in ("fishing-v4",) else None
    for s in range(T):
        if env_id == "fishing-v0":
            a = torch.randint(0, 100, (n,), device="cuda", generator=g, dtype=torch.int32)
        else:
            a = torch.rand(n, device="cuda", generator=g) * 1.2 - 1.0          # quota in [0, 1.2 K)
        prev = env._obs.clone()
        obs, rew, done, _ = env.step(a)
        dones += int(done.sum())
        if s % 50 == 0 or s == T - 1:
            assert bool(torch.isfinite(obs).all()) and bool((obs >= -1.0).all()), (env_id, s)
            assert bool((env._t >= 0).all()) and bool((env._t <= Tmax).all())
            assert bool((rew >= 0).all())
            if K is not None and env_id != "fishing-v11":
                # harvest <= stock before the step: reward <= (obs_prev + 1) * K (+ rounding)
                assert bool((rew <= (prev + 1.0) * K + 1e-5).all()), (env_id, s)
    st = env.episode_stats()
    assert st["n_episodes"] == dones and dones >= n * (T // (Tmax + 1))
    assert st["sum_length"] <= dones * (Tmax + 1) and st["mean_return"] >= 0


def test_v4_num_envs_bmsy_and_msy_follow_each_envs_parameters(gf):
    """policies.BMSY / msy on an N-env fishing-v4 batch (round 4; the parameter MEANS before): one S per env, swept under the
    (K, r) that env has drawn -- what N reference envs return, one BMSY() each (models/policies.py:51-67) -- equal to the
    scalar protocol's BMSY with the same pair in force, and to the sweep written out in NumPy float32; msy's quota is
    f(S_i) - S_i under the pair env i holds AFTER BMSY's reset (the reference's order of events, :7-13)."""
    import torch
    from gym_fishing_amd import policies
    n = 1000
    env = gf.make("fishing-v4", num_envs=n, sigma=0.05, sigma_p=0.25, seed=12)
    env.reset()
    K, r = env.K.clone(), env.r.clone()
    S = policies.BMSY(env)
    assert isinstance(S, torch.Tensor) and S.shape == (n,) and S.dtype == torch.float32
    assert not torch.equal(env.K, K)                              # BMSY reset the env: new draws
    # (the reference's own call: np.linspace with the Box's float32 ARRAY endpoints is not np.linspace(-1.0, 1.0, ...) -- 5734 of
    # the 10001 points differ by an ulp)
    grid = np.linspace(env.observation_space.low, env.observation_space.high, num=10001, dtype=np.float32).reshape(-1)
    Kh, rh, Sh = K.cpu().numpy(), r.cpu().numpy(), S.cpu().numpy()
    one = np.float32(1)
    for i in list(range(0, n, 97)) + [int(np.argmin(Kh)), int(np.argmax(Kh)), int(np.argmin(rh))]:
        with np.errstate(all="ignore"):
            x0 = (grid + one) * Kh[i]
            g = np.maximum((x0 + ((rh[i] * x0) * (one - (x0 / Kh[i])))) + ((x0 * np.float32(0)) * np.float32(0)), np.float32(0)) - x0
        assert Sh[i] == x0[int(np.argmax(g))], (i, Kh[i], rh[i])
    for i in (0, 501, 999):                                       # the scalar protocol with that pair in force
        one_env = gf.make("fishing-v4", sigma=0.05)
        one_env.K, one_env.r = float(Kh[i]), float(rh[i])
        assert policies.BMSY(one_env) == float(Sh[i])
    # escapement / msy on the batch
    esc = policies.escapement(env)
    assert esc.kernel_policy[1] is esc.S and esc.S.shape == (n,)        # (one S per env: the fused kernel takes the tensor, ABI 7)
    a, _ = esc.predict(env.state)
    assert a.shape == (n, 1)
    m = policies.msy(env)
    assert m.msy.shape == (n,) and m.kernel_policy[1] is m.msy
    df = env.simulate(m)
    assert len(df) > n
    # the fused rollout under each env's own S / quota == the step loop driven by predict(), bit for bit (float64 layout: the
    # in-kernel policy then does the host policy's float64 arithmetic; the twin starts from the batch's own checkpoint)
    mk64 = lambda: gf.make("fishing-v4", num_envs=n, sigma=0.05, sigma_p=0.25, seed=12, Tmax=7, dtype=torch.float64)   # noqa: E731
    A = mk64()
    A.reset()
    for cls in (policies.escapement, policies.msy):
        model = cls(A)
        assert isinstance(model.kernel_policy[1], torch.Tensor) and model.kernel_policy[1].shape == (n,)
        B = mk64()
        B.load_state_dict(A.state_dict())
        traj = A.rollout(25, policy=model.kernel_policy, record=True)
        model.env = B
        for s in range(25):
            o = B.state.clone()
            a, _ = model.predict(o)
            # (bit for bit, NaN for NaN: sigma_p = 0.25 now and then clips a K to 0, and 0 / 0 is the action there on both sides)
            bits = lambda x: x.contiguous().view(torch.int64)          # noqa: E731
            assert torch.equal(bits(traj[s, 0]), bits(o.reshape(-1))), s
            assert torch.equal(bits(traj[s, 1]), bits(a.reshape(-1).to(traj.dtype))), s
            _, rew, done, _ = B.step(a.reshape(-1))
            assert torch.equal(bits(traj[s, 2]), bits(rew)) and torch.equal(traj[s, 3].bool(), done.bool()), s
        model.env = A
        assert torch.equal(bits(A.state), bits(B.state)) and torch.equal(A.K, B.K) and torch.equal(A._t, B._t)
        assert len(set(model.kernel_policy[1].cpu().tolist())) > n // 2          # (really one parameter per env)
    # the C entry point's argument checks
    from gym_fishing_amd import _capi
    lib = _capi.lib()
    p = env._c_params()
    st = torch.as_tensor(grid, device="cuda")
    out = torch.empty(n, device="cuda")
    assert lib.fishing_bmsy_sweep_f32(p, n, K.data_ptr(), r.data_ptr(), st.data_ptr(), 0, out.data_ptr(), None) == -4
    assert lib.fishing_bmsy_sweep_f32(p, n, K.data_ptr(), r.data_ptr(), None, 5, out.data_ptr(), None) == -1
    assert lib.fishing_bmsy_sweep_f32(p, n, None, None, st.data_ptr(), 10001, out.data_ptr(), None) == 0          # the struct's K_mean-less scalars
    z = gf.make("fishing-v9", num_envs=8)
    assert lib.fishing_bmsy_sweep_f32(z._c_params(), 8, None, None, st.data_ptr(), 10001, out.data_ptr(), None) == -2
    x = torch.rand(n, device="cuda")
    assert lib.fishing_population_draw_f32(z._c_params(), 8, x.data_ptr(), None, None, r.data_ptr(), K.data_ptr(), out.data_ptr(), None) == -7
