"""Pin the CPU oracle (oracle/fishing_oracle.py) to the reference.

The golden vectors were captured from the unmodified reference by
tests/golden/make_golden.py; every comparison here is bit-for-bit (float64 viewed
as int64), v2's np.exp included (same NumPy on both sides).
"""
import numpy as np
import pytest

from conftest import load_golden_cases
from oracle import fishing_oracle as fo

CASES = load_golden_cases()


def bits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64)).view(np.int64)


def assert_bit_equal(a, b, what):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    assert same.all(), "%s: %d/%d mismatches, first at %s: %r vs %r" % (
        what, (~same).sum(), same.size, np.argwhere(~same)[0], a[~same][0], b[~same][0])


def case_params(c):
    model = fo.MODEL_OF_ID[c.id]
    return dict(model=model, sigma=c.param("sigma"), C=c.param("C"), Tmax=c.param("Tmax"),
                n_actions=c.param("n_actions"))


@pytest.mark.parametrize("c", CASES, ids=[c.name for c in CASES])
def test_single_step_bit_exact(c):
    """Every recorded step, fed the reference's own input state."""
    p = case_params(c)
    t_in = np.where(np.arange(c.nsteps)[None, :] == 0, 0, np.roll(c.t, 1, axis=1))
    # after an auto-reset the counter restarts at 0
    if c.auto_reset:
        prev_done = np.roll(c.done, 1, axis=1).astype(bool)
        prev_done[:, 0] = False
        t_in = np.where(prev_done, 0, t_in)
    obs, rew, done, t, _ = fo.step(p["model"], c.obs_in, t_in, c.action, c.z, c.r, c.K,
                                   p["sigma"], C=p["C"], Tmax=p["Tmax"], n_actions=p["n_actions"])
    assert_bit_equal(obs, c.obs, c.name + " obs")
    assert_bit_equal(rew, c.reward, c.name + " reward")
    assert (done == c.done).all()
    assert (t == c.t).all()


@pytest.mark.parametrize("c", CASES, ids=[c.name for c in CASES])
def test_free_running_trajectory_bit_exact(c):
    """Carry the oracle's own state across steps (and resets), as a rollout does."""
    p = case_params(c)
    model = p["model"]
    E = c.obs.shape[0]
    x0 = c.param("init_state")
    if model == fo.MODEL_V4:
        if c.init_reset:
            K, r = fo.draw_model_error_params(c.zK[:, 0], c.zr[:, 0], c.param("K_mean"),
                                              c.param("r_mean"), c.param("sigma_p"))
            obs = fo.reset_obs(model, x0, K)
        else:  # constructor state: K, r already drawn; obs = x0 / K_mean - 1 (quirk a9)
            K, r = c.K[:, 0].copy(), c.r[:, 0].copy()
            obs = np.full(E, x0 / c.param("K_mean") - 1.0)
    else:
        K = np.full(E, float(c.param("K")))
        r = np.full(E, float(c.param("r")))
        obs = fo.reset_obs(model, x0, K)
    if c.init_reset:
        assert_bit_equal(obs, c.reset_obs[:, 0], c.name + " initial reset obs")
    t = np.zeros(E, dtype=np.int32)
    for s in range(c.nsteps):
        assert_bit_equal(obs, c.obs_in[:, s], "%s obs_in step %d" % (c.name, s))
        obs, rew, done, t, _ = fo.step(model, obs, t, c.action[:, s], c.z[:, s], r, K, p["sigma"],
                                       C=p["C"], Tmax=p["Tmax"], n_actions=p["n_actions"])
        assert_bit_equal(obs, c.obs[:, s], "%s obs step %d" % (c.name, s))
        assert_bit_equal(rew, c.reward[:, s], "%s reward step %d" % (c.name, s))
        assert (done == c.done[:, s]).all()
        assert (t == c.t[:, s]).all()
        if c.auto_reset:
            obs, t, K, r = fo.auto_reset(model, obs, done, t, K, r, x0,
                                         zK=np.nan_to_num(c.zK[:, s + 1]),
                                         zr=np.nan_to_num(c.zr[:, s + 1]),
                                         K_mean=c.param("K_mean"), r_mean=c.param("r_mean"),
                                         sigma_p=c.param("sigma_p"))
            m = done.astype(bool)
            assert_bit_equal(obs[m], c.reset_obs[m, s + 1], "%s reset obs step %d" % (c.name, s))
        if model == fo.MODEL_V4 and s + 1 < c.nsteps:
            assert_bit_equal(K, c.K[:, s + 1], "%s K step %d" % (c.name, s))
            assert_bit_equal(r, c.r[:, s + 1], "%s r step %d" % (c.name, s))


def test_known_answers_from_survey(anchors):
    """SURVEY.md Appendix A.4 anchors + the reference's own test_tipping pins
    (tests/test-envs.py:93-106)."""
    a = anchors["v1_sigma0_const"]
    assert a["first5_obs_hex"][0] == "-0x1.fc00000000000p-3"
    assert a["return"] == 6.3125 and a["n_steps_to_done"] == 101
    # replay with the oracle
    obs = fo.reset_obs(fo.MODEL_V1, 0.75, np.array([1.0]))
    t = np.zeros(1, np.int32)
    ret = 0.0
    for s in range(101):
        obs, rew, done, t, _ = fo.step(fo.MODEL_V1, obs, t, np.float32([-0.9375]), [0.0], 0.3, 1.0, 0.0)
        if s < 5:
            assert float(obs[0]).hex() == a["first5_obs_hex"][s]
        ret += float(rew[0])
        assert bool(done[0]) == (s == 100)
    assert ret == a["return"] and float(obs[0]) == a["final_obs"]
    # reference test_tipping: grows from 0.75, declines from 0.3 (zero quota, sigma = 0)
    for x0, key, cmp in ((0.75, "from_0.75", np.greater_equal), (0.3, "from_0.3", np.less_equal)):
        obs = fo.reset_obs(fo.MODEL_V2, x0, np.array([1.0]))
        obs, *_ , x = fo.step(fo.MODEL_V2, obs, [0], np.float32([-1.0]), [0.0], 0.3, 1.0, 0.0, C=0.5)
        pop = (obs[0] + 1.0) * 1.0
        assert cmp(pop, x0)
        assert pop == anchors["test_tipping"][key]


def test_quota_map_anchors(anchors):
    q = fo.quota_from_action(fo.MODEL_V0, np.arange(0, 101, 5), 1.0, 100)
    assert_bit_equal(q, anchors["get_quota_v0"], "v0 quota map")
    # clip: +5 -> quota 2K, -5 -> 0
    q = fo.quota_from_action(fo.MODEL_V1, np.float32([5.0, -5.0, 0.3337]), 1.0, 100)
    assert q[0] == 2.0 and q[1] == 0.0 and q[2] == 1.3337000012397766


def test_nan_and_zero_K_follow_ieee():
    """K = 0 is reachable in fishing-v4 (clip at 0): 0/0 -> NaN must propagate like NumPy."""
    obs, rew, done, t, x = fo.step(fo.MODEL_V4, [0.75], [0], np.float32([-0.9]), [0.3], 0.3, 0.0, 0.1)
    assert np.isnan(obs[0]) and not done[0]
    g = np.maximum(np.float64(-0.0), 0.0)
    assert not np.signbit(g)


PHILOX_KAT = [  # Random123 kat_vectors: philox4x32-10
    ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


@pytest.mark.parametrize("ctr,key,want", PHILOX_KAT)
def test_philox_known_answers(ctr, key, want):
    got = fo.philox4x32_10(*ctr, *key)
    assert tuple(int(x) for x in got) == want


PHILOX2_KAT = [  # Random123 kat_vectors: philox2x32-10  (ctr0, ctr1), key -> (out0, out1)
    ((0, 0), 0, (0xFF1DAE59, 0x6CD10DF2)),
    ((0xFFFFFFFF, 0xFFFFFFFF), 0xFFFFFFFF, (0x2C3F628B, 0xAB4FD7AD)),
    ((0x243F6A88, 0x85A308D3), 0x13198A2E, (0xDD7CE038, 0xF62A4C12)),
]


@pytest.mark.parametrize("ctr,key,want", PHILOX2_KAT)
def test_philox2x32_known_answers(ctr, key, want):
    """fishing-v4's per-env parameter draws (fishing_common.h: draw_model_error) use Philox2x32-10."""
    got = fo.philox2x32_10(*ctr, key)
    assert tuple(int(x) for x in got) == want


def test_v4_parameter_draw_statistics_and_origin_rule():
    """(zK, zr) of the Philox2x32 parameter stream: standard normal, uncorrelated, distinct per env / counter /
    stream / seed half; and the rule that dates an episode from the env's year counter (v4_origin)."""
    n = 1 << 16
    env = np.arange(n, dtype=np.uint64)
    zK, zr = (x.astype(np.float64) for x in fo.reset_normals(99, env, 5, fo.STREAM_RESET))
    for z in (zK, zr):
        assert abs(z.mean()) < 4.0 / np.sqrt(n) and abs(z.var() - 1.0) < 0.03
    assert abs(np.corrcoef(zK, zr)[0, 1]) < 0.02
    base = fo.param_words(99, env[:64], 5, fo.STREAM_RESET)[0]
    for other in (fo.param_words(99, env[:64], 5, fo.STREAM_AUTORESET), fo.param_words(99, env[:64], 6, fo.STREAM_RESET),
                  fo.param_words(99 + (1 << 32), env[:64], 5, fo.STREAM_RESET),
                  fo.param_words(99, env[:64] + np.uint64(1 << 32), 5, fo.STREAM_RESET),
                  fo.param_words(99, env[:64], 5 + (1 << 32), fo.STREAM_RESET)):
        assert not np.array_equal(base, other[0])
    # the reset() stream and the auto-reset stream can never meet while counters stay below 2^31 (bit 31 of the second
    # counter word is the stream's): not for equal counters, not for the pair round 2's XOR tags made collide
    # (a ^ b == 0x5851F42D ^ 0x2545F491), not across a sample of the range
    rng = np.random.default_rng(0)
    a_cnt = rng.integers(0, 1 << 31, 200)
    for a_, b_ in list(zip(a_cnt, a_cnt)) + [(int(x), int(x) ^ (0x5851F42D ^ 0x2545F491)) for x in a_cnt[:50]]:
        b_ &= 0x7FFFFFFF
        w_r = fo.param_words(99, env[:8], int(a_), fo.STREAM_RESET)
        w_a = fo.param_words(99, env[:8], int(b_), fo.STREAM_AUTORESET)
        assert not np.array_equal(w_r[0], w_a[0]) and not np.array_equal(w_r[1], w_a[1])
    # ... and beyond 2^31 steps the counter's high part moves to the key: still distinct blocks
    assert not np.array_equal(fo.param_words(99, env[:8], 5, fo.STREAM_AUTORESET)[0],
                              fo.param_words(99, env[:8], 5 + (1 << 31), fo.STREAM_AUTORESET)[0])
    # an env that has run since the reset made at step count 40 (reset counter 3); one auto-reset by step 57
    stream, counter = fo.v4_origin(step_counter=60, t=np.array([20, 2]), origin_step=40, origin_counter=3)
    assert stream.tolist() == [fo.STREAM_RESET, fo.STREAM_AUTORESET] and counter.tolist() == [3, 57]


def test_v11_model_draw_scheme():
    """fishing-v11's model choice (growth_models.py:187,200: np.random.choice(models)) as the kernels draw it: one Philox2x32-10
    block per env QUAD, four 16-bit halves.  The index map (half * n) >> 16 splits the 65536 halves into n buckets whose sizes
    differ by at most one -- every model within 2^-16 of 1 / n --, the four envs of a quad take the four halves of ONE block
    (a Random123 known answer), the two reset streams and consecutive counters give unrelated draws."""
    for n in (1, 2, 3, 4, 5):
        sizes = np.bincount((np.arange(65536, dtype=np.uint64) * np.uint64(n) >> np.uint64(16)).astype(int), minlength=n)
        assert sizes.sum() == 65536 and sizes.max() - sizes.min() <= 1
        assert np.abs(sizes / 65536.0 - 1.0 / n).max() <= 2.0 ** -16
    # seed 0, counter 0, auto-reset stream, quad 0: key = 0 ^ tag, counter words (0, 0)
    w0, w1 = fo.philox2x32_10(np.uint32(0), np.uint32(0), np.uint32(0x4D4F444C))
    halves = fo.model_words(0, np.arange(4, dtype=np.uint64), 0, fo.STREAM_AUTORESET)
    assert halves.tolist() == [int(w0) & 0xFFFF, int(w0) >> 16, int(w1) & 0xFFFF, int(w1) >> 16]
    # the block of quad q on the reset stream: c1 carries bit 31
    w0, w1 = fo.philox2x32_10(np.uint32(7), np.uint32(0x80000003), np.uint32(0x4D4F444C ^ 5))
    assert int(fo.model_words(5, np.uint64(4 * 7 + 2), 3, fo.STREAM_RESET)) == int(w1) & 0xFFFF
    env = np.arange(1 << 18, dtype=np.uint64)
    a = fo.model_draw(1, env, 0, fo.STREAM_RESET, [0, 1, 2, 3, 4])
    b = fo.model_draw(1, env, 0, fo.STREAM_AUTORESET, [0, 1, 2, 3, 4])
    c = fo.model_draw(1, env, 1, fo.STREAM_AUTORESET, [0, 1, 2, 3, 4])
    for u, v in ((a, b), (b, c), (a, c)):
        assert abs((u == v).mean() - 0.2) < 0.01            # independent draws agree one time in five
    counts = np.bincount(a, minlength=5)
    assert (((counts - counts.sum() / 5) ** 2) / (counts.sum() / 5)).sum() < 18.5
    # neighbouring envs of one quad are as unrelated as envs of different quads
    assert abs((a[0::4] == a[1::4]).mean() - 0.2) < 0.01 and abs((a[1::4] == a[2::4]).mean() - 0.2) < 0.01


def test_noise_statistics():
    """The Philox + Box-Muller stream is standard normal and independent of sharding."""
    n = 1 << 16
    z = fo.noise_normal(1234, np.arange(n), 7).astype(np.float64)
    assert abs(z.mean()) < 4.0 / np.sqrt(n)
    assert abs(z.var() - 1.0) < 0.03
    from scipy import stats
    assert stats.kstest(z, "norm").pvalue > 1e-3
    # global env index keys the stream: a shard starting at 1000 sees the same numbers
    z2 = fo.noise_normal(1234, np.arange(1000, 1100), 7)
    assert (z2 == z[1000:1100].astype(np.float32)).all()
    # consecutive steps are uncorrelated
    z3 = fo.noise_normal(1234, np.arange(n), 8).astype(np.float64)
    assert abs(np.corrcoef(z, z3)[0, 1]) < 0.02
    a = fo.policy_random_action(fo.MODEL_V1, 1234, np.arange(n), 7)
    assert a.min() >= -1.0 and a.max() <= 1.0 and abs(a.mean()) < 0.01
    ai = fo.policy_random_action(fo.MODEL_V0, 1234, np.arange(n), 7, n_actions=100)
    assert ai.min() == 0 and ai.max() == 99


@pytest.mark.parametrize("c", CASES, ids=[c.name for c in CASES])
def test_scalar_env_reproduces_reference_when_seeded_identically(c):
    """oracle/scalar_env.py consumes the global legacy NumPy stream exactly as the reference
    does (one normal per step, two per fishing-v4 construct / reset), so np.random.seed(s)
    alone -- no externally supplied noise -- reproduces the golden trajectories bit-for-bit."""
    from oracle.scalar_env import ScalarFishingEnv
    for e, seed in enumerate(c.meta["seeds"]):
        np.random.seed(seed)
        env = ScalarFishingEnv(c.id, **c.kwargs)
        if c.init_reset:
            o = env.reset()
            assert bits(o[0]) == bits(c.reset_obs[e, 0])
        for s in range(c.nsteps):
            a = int(c.action[e, s]) if c.id == "fishing-v0" else np.array([c.action[e, s]], np.float32).astype(np.float64)
            obs, rew, done, info = env.step(a)
            assert bits(obs[0]) == bits(c.obs[e, s]), (c.name, e, s)
            assert bits(rew) == bits(c.reward[e, s]), (c.name, e, s)
            assert done == bool(c.done[e, s]) and env.t == c.t[e, s]
            assert isinstance(info, dict) and obs.shape == (1,) and obs.dtype == np.float64
            if done and c.auto_reset:
                o = env.reset()
                assert bits(o[0]) == bits(c.reset_obs[e, s + 1])


from conftest import load_policy_sims  # noqa: E402

SIMS = load_policy_sims()


@pytest.mark.parametrize("c", SIMS, ids=[c["key"] for c in SIMS])
def test_policy_rollouts_reproduce_reference_simulate_tables(c):
    """the msy / escapement rules + step() reproduce env.simulate()'s table
    [time, state, action, reward] of the reference bit-for-bit (shared_env.py:29-54)."""
    model = fo.MODEL_OF_ID[c["env_id"]]
    K, nact = c["K"], c["n_actions"]
    obs = np.array([c["x0"] / K - 1.0])
    t = np.zeros(1, np.int32)
    rows, quota_prev, rew_prev = [], 0.0, 0.0
    for s in range(100):
        rows.append([s, (obs[0] + 1.0) * K, quota_prev, rew_prev])          # record BEFORE acting (:37-38)
        act = fo.policy_action(c["policy"], c["param"], model, obs, K, nact)
        obs, rew, done, t, _ = fo.step(model, obs, t, act, [0.0], c["r"], K, 0.0, n_actions=nact)
        quota_prev, rew_prev = fo.quota_from_action(model, act, K, nact)[0], rew[0]
        if done[0]:
            break
    assert_bit_equal(np.array(rows), c["table"], c["key"])


# ------------------------------------------------------------------ growth-model zoo
from conftest import load_zoo_cases  # noqa: E402

ZOO = load_zoo_cases()
ZOO_DEFAULTS = {   # constructor defaults, growth_models.py:6-154
    "fishing-v5": {"r": 0.3, "K": 1, "C": 0.5, "sigma": 0.0, "init_state": 0.75},
    "fishing-v6": {"r": 0.3, "K": 1, "sigma": 0.0, "init_state": 0.75},
    "fishing-v7": {"r": 0.7, "K": 1.5, "M": 1.5, "q": 3, "b": 0.15, "sigma": 0.0, "a": 0.2, "init_state": 0.75},
    "fishing-v8": {"r": 1.0, "K": 1.0, "M": 1.0, "theta": 3.0, "sigma": 0.0, "init_state": 1.5},
    "fishing-v9": {"r": 0.3, "K": 1, "sigma": 0.0, "init_state": 0.75},
    "fishing-v10": {"r": 0.8, "K": 1, "sigma": 0.0, "alpha": -0.007, "init_state": 0.75},
    "fishing-v11": {"K": 1, "init_state": 0.75},
}


def zoo_params(c):
    P = dict(ZOO_DEFAULTS[c.id])
    P.update({k: v for k, v in c.kwargs.items() if k != "Tmax"})
    return P


@pytest.mark.parametrize("c", ZOO, ids=[c.name for c in ZOO])
def test_zoo_single_step_bit_exact(c):
    """fishing-v5..v11: every recorded reference step replayed by the oracle (same NumPy
    log / exp / power on the same dtype, so bit-for-bit on the CPU)."""
    model = fo.MODEL_OF_ID[c.id]
    P = zoo_params(c)
    Tmax = c.kwargs.get("Tmax", 100)
    t_in = np.where(np.arange(c.nsteps)[None, :] == 0, 0, np.roll(c.t, 1, axis=1))
    prev_done = np.roll(c.done, 1, axis=1).astype(bool)
    prev_done[:, 0] = False
    t_in = np.where(prev_done, 0, t_in)
    kind = None
    if model == fo.MODEL_V10:
        P["r"] = c.params_r + P["alpha"]            # the draw uses r AFTER the += alpha
        assert np.allclose(c.params_r[:, 1:], c.params_r[:, :-1] + P["alpha"], rtol=0, atol=1e-15)
    if model == fo.MODEL_V11:
        kind, P = c.model_idx, None
        assert set(np.unique(kind)) <= {0, 1, 2, 3, 4} and len(np.unique(kind)) >= 3
    obs, rew, done, t, _ = fo.step_zoo(model, c.obs_in, t_in, c.action, c.z, P, c.K, Tmax=Tmax, kind=kind)
    assert_bit_equal(obs, c.obs, c.name + " obs")
    assert_bit_equal(rew, c.reward, c.name + " reward")
    assert (done == c.done).all() and (t == c.t).all()


# ------------------------------------------------------------------ simulate_mdp_vec over fishing-v4 (K per row)
def test_reference_vec_sim_of_fishing_v4_uses_the_K_in_force_at_each_row():
    """tests/golden/reference_vec_sims_v4.npz: the reference's simulate_mdp_vec (shared_env.py:57-79) over three
    FishingModelError envs, with the observation and the env's K logged at every get_fish_population call
    (df_entry_vec, shared_env.py:15-26).  Facts of the reference this pins: a row's population is (obs + 1) * K with
    the K the env holds AT THAT ROW -- after an auto-reset inside the table that is the redrawn K -- and the first obs
    of an episode is x0 un-normalised (quirk B8).  The oracle's scalar envs behind the same harness, seeded the same
    way, reproduce the table bit for bit."""
    import json
    import os

    from conftest import GOLDEN
    from oracle import scalar_env
    z = np.load(os.path.join(GOLDEN, "reference_vec_sims_v4.npz"))
    tab, obs_rows, K_rows = z["v4_constant/table"], z["v4_constant/obs_rows"], z["v4_constant/K_rows"]
    meta = json.loads(str(z["v4_constant/meta"]))
    n, Tmax, reps = meta["num_envs"], meta["kwargs"]["Tmax"], meta["n_eval_episodes"] // meta["num_envs"]
    assert tab.shape == (reps * (Tmax + 1) * n, 5)
    assert np.array_equal(tab[:, 1], (obs_rows + 1.0) * K_rows)
    K = K_rows.reshape(reps, Tmax + 1, n)
    assert (K[:, 1:] != K[:, :-1]).any(axis=(0, 1)).all()          # every env changed its K inside a table
    assert np.array_equal(obs_rows.reshape(reps, Tmax + 1, n)[:, 0], np.full((reps, n), 0.75))
    # the oracle's scalar envs behind a DummyVecEnv-shaped loop (envs stepped in order, finished envs reset at once)
    envs = [scalar_env.ScalarFishingEnv("fishing-v4", **meta["kwargs"]) for _ in range(n)]
    np.random.seed(meta["seed"])
    rows = []
    for rep in range(reps):
        obs = [e.reset() for e in envs]
        action, reward = [-1.0] * n, [0.0] * n
        for t in range(Tmax + 1):
            for i, e in enumerate(envs):
                rows.append([t, (obs[i][0] + 1) * e.K, action[i], reward[i], rep * n + i])
            if t == Tmax:
                break
            for i, e in enumerate(envs):
                a = np.array([0.2], dtype=np.float32).astype(np.float64)
                o, r, d, _ = e.step(a)
                if d:
                    o = e.reset()
                obs[i], action[i], reward[i] = o, 0.2, r
    got = np.array(rows, dtype=np.float64)
    assert np.array_equal(got[:, [0, 4]], tab[:, [0, 4]])
    assert np.array_equal(got[:, 1], tab[:, 1]) and np.array_equal(got[:, 3], tab[:, 3])
    assert np.allclose(got[:, 2], tab[:, 2], rtol=0, atol=1e-7)     # the raw action column (float32(0.2) in the reference's table)


def test_growth_functions_bit_exact():
    """growth_models.py:208-269 called directly -- f(x, params) on a scalar, a vector and a matrix of populations (zero,
    1e-300, far above K, negative; r < 0 for Beverton-Holt's clip): the oracle's zoo_population_draw on the recorded
    normals reproduces every result bit for bit, NaN for NaN."""
    from conftest import load_growth_function_cases
    cases = load_growth_function_cases()
    assert len(cases) == 11 and {c["name"] for c in cases} == {"allen", "beverton_holt", "myers", "may", "ricker"}
    for c in cases:
        for shape_tag, x, z, want in c["calls"]:
            got = fo.zoo_population_draw(c["kind"], x, z, c["params"])
            assert got.shape == want.shape and np.array_equal(got, want, equal_nan=True), (c["tag"], shape_tag)

