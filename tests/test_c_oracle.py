"""Pin the plain-C oracle (oracle/fishing_oracle.c, the `c_port` CPU baseline of bench.py) to the
reference's golden vectors and to the Python oracle: same bits for the step, the same Philox quad
scheme for the noise and the random policy."""
import numpy as np
import pytest

from conftest import load_golden_cases
from oracle import c_oracle as co
from oracle import fishing_oracle as fo

CASES = load_golden_cases()


@pytest.mark.parametrize("c", CASES, ids=[c.name for c in CASES])
def test_c_step_reproduces_reference_vectors(c):
    """Every recorded step of the reference, fed its own input state: float64 bit-for-bit (fishing-v2:
    libm exp vs np.exp may differ in the last place)."""
    model = fo.MODEL_OF_ID[c.id]
    t_in = np.where(np.arange(c.nsteps)[None, :] == 0, 0, np.roll(c.t, 1, axis=1))
    if c.auto_reset:
        prev_done = np.roll(c.done, 1, axis=1).astype(bool)
        prev_done[:, 0] = False
        t_in = np.where(prev_done, 0, t_in)
    flat = lambda a: np.ascontiguousarray(np.asarray(a).reshape(-1))  # noqa: E731
    obs, rew, done, t = co.step(model, flat(c.obs_in), flat(t_in), flat(c.action), flat(c.z), flat(c.r), flat(c.K),
                                c.param("sigma"), C=c.param("C"), Tmax=c.param("Tmax"),
                                n_actions=c.param("n_actions"))
    if model == fo.MODEL_V2:
        # one ulp of exp() on the population, seen through obs = x / K - 1 (absolute, K = 1)
        assert np.abs(obs - flat(c.obs)).max() <= 4.5e-16
    else:
        assert np.array_equal(obs.view(np.int64), flat(c.obs).astype(np.float64).view(np.int64))
        assert (done == flat(c.done)).all()
    assert np.array_equal(rew.view(np.int64), flat(c.reward).astype(np.float64).view(np.int64))
    assert (t == flat(c.t)).all()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_c_step_matches_python_oracle_on_random_inputs(dtype):
    rng = np.random.default_rng(5)
    n = 20000
    for model in (fo.MODEL_V0, fo.MODEL_V1, fo.MODEL_V4):
        obs = rng.uniform(-1, 1, n).astype(dtype)
        t = rng.integers(0, 101, n).astype(np.int32)
        a = rng.integers(0, 100, n).astype(np.int32) if model == fo.MODEL_V0 else rng.uniform(-1.2, 1.2, n).astype(np.float32)
        z = rng.normal(0, 1, n).astype(dtype)
        r = rng.uniform(0.1, 0.9, n).astype(dtype)
        K = rng.uniform(0.5, 2.0, n).astype(dtype)
        eo, er, ed, et, _ = fo.step(model, obs, t, a, z, r, K, 0.1, dtype=dtype)
        o, rew, done, t2 = co.step(model, obs, t, a, z, r, K, 0.1, dtype=dtype)
        it = np.int64 if dtype == np.float64 else np.int32
        assert np.array_equal(o.view(it), eo.view(it)) and np.array_equal(rew.view(it), er.view(it))
        assert (done == ed).all() and (t2 == et).all()


def test_c_noise_follows_the_quad_scheme():
    seed, counter = 0xFEEDFACE12345678, 991
    for off, n in ((0, 4099), ((1 << 35) + 6, 1001)):
        z, a = co.noise(n, off, seed, counter)
        env = np.arange(off, off + n, dtype=np.uint64)
        assert np.array_equal(a, fo.policy_random_action(fo.MODEL_V1, seed, env, counter))
        # libm logf / sqrtf / cosf in float32 vs the float64 restatement rounded to float32
        assert np.abs(z - fo.noise_normal(seed, env, counter)).max() < 2e-6


def test_c_rollout_is_thread_and_shard_invariant():
    """The baseline workload (random policy, auto-reset) gives the same state for any thread count
    and when split into env_offset shards; its reward total matches stepping the Python oracle with
    the C generator's normals."""
    n, T = 1024, 6
    tot1, o1, t1 = co.rollout_random_f32(1, n, T, threads=1, seed=7)
    tot4, o4, t4 = co.rollout_random_f32(1, n, T, threads=4, seed=7)
    assert np.array_equal(o1, o4) and np.array_equal(t1, t4) and np.isclose(tot1, tot4, rtol=1e-12)
    oa = co.rollout_random_f32(1, n // 2, T, threads=2, seed=7)[1]
    ob = co.rollout_random_f32(1, n // 2, T, threads=2, seed=7, env_offset=n // 2)[1]
    assert np.array_equal(np.concatenate([oa, ob]), o1)
    obs, t, total = np.full(n, -0.25, np.float32), np.zeros(n, np.int32), 0.0
    for s in range(T):
        z, a = co.noise(n, 0, 7, s)
        obs, rew, done, t, _ = fo.step(fo.MODEL_V1, obs, t, a, z, 0.3, 1.0, 0.1, dtype=np.float32)
        total += float(rew.astype(np.float64).sum())
        obs = np.where(done.astype(bool), np.float32(-0.25), obs)
        t = np.where(done.astype(bool), 0, t).astype(np.int32)
    assert np.array_equal(obs, o1) and np.array_equal(t, t1) and np.isclose(total, tot1, rtol=1e-9)
