#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the UNMODIFIED reference.

Test infrastructure -- never imported by the product.  Runs only in the build
container, where the upstream reference is mounted read-only at /root/reference;
exits 0 without writing anything when it is absent (e.g. on the GPU box).

How the reference is made importable: the reference
imports ``gym`` but uses only five symbols of it as *containers* -- ``gym.Env``
(a base class), ``gym.spaces.Box`` / ``Discrete`` (attribute bags),
``gym.spaces.discrete.Discrete`` (for an isinstance test) and
``gym.envs.registration.register``.  ``gym`` is not installed and there is no
network, so an in-memory module with exactly those names is placed in
``sys.modules``.  None of the arithmetic on the hot path lives in ``gym``: the
reference's own ``step/reset/population_draw`` run unmodified on NumPy.

What is captured, per case: ctor kwargs, the action fed at every step, the
standard normal ``z`` the reference consumed in that step (peeked from the legacy
global RandomState by get_state / draw / set_state, so the stream the reference
sees is untouched), and the reference's outputs ``obs, reward, done`` plus the
bookkeeping (``years_passed``; for fishing-v4 the ``K, r`` in force).  Only
numbers are stored -- no reference source text.

Actions: the reference is pinned to NumPy 1.19 (examples/requirements.txt:14),
whose value-based promotion evaluates ``(clip(a)+1)*K`` in float64; under
NumPy 2 the same expression on a float32 scalar stays float32 (SURVEY.md
Appendix A.3).  Fixtures therefore feed ``float32 action -> .astype(float64)``
so NumPy 2 reproduces the reference-era float64 arithmetic; the float32 value
is what is stored.

Usage:  python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = HERE      # overridden by `--out DIR` (tests/test_golden_provenance.py regenerates into a temp dir)


def _install_gym_stand_in():
    import numpy as np

    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")
    discrete = types.ModuleType("gym.spaces.discrete")
    envs = types.ModuleType("gym.envs")
    registration = types.ModuleType("gym.envs.registration")

    class Env:
        metadata = {}

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            self.shape = self.low.shape
            self.dtype = np.dtype(dtype)

    class Discrete:
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.dtype(np.int64)

    registry = {}

    def register(id, entry_point, **kw):
        registry[id] = entry_point

    def make(id, **kw):
        import importlib

        mod, cls = registry[id].split(":")
        return getattr(importlib.import_module(mod), cls)(**kw)

    gym.Env = Env
    gym.spaces = spaces
    gym.envs = envs
    gym.make = make
    spaces.Box = Box
    spaces.Discrete = Discrete
    spaces.discrete = discrete
    discrete.Discrete = Discrete
    envs.registration = registration
    registration.register = register
    for m in (gym, spaces, discrete, envs, registration):
        sys.modules[m.__name__] = m
    return gym


def _peek_normal(np):
    """The standard normal the NEXT np.random.normal(0, 1) call will return."""
    st = np.random.get_state()
    z = np.random.normal(0, 1)
    np.random.set_state(st)
    return z


def main():
    if not os.path.isdir(REF):
        print("reference not mounted at %s; nothing generated" % REF)
        return 0
    sys.dont_write_bytecode = True
    import matplotlib

    matplotlib.use("Agg")
    import numpy as np

    gym = _install_gym_stand_in()
    sys.path.insert(0, REF)
    import gym_fishing  # noqa: F401  (registers ids)
    from gym_fishing.models.policies import BMSY, escapement, msy

    out_default = {}
    out = out_default
    anchors = {}

    def as_action(env_id, a):
        if env_id == "fishing-v0":
            return int(a)
        return np.array([a], dtype=np.float32).astype(np.float64)

    ZOO_MODELS = ["allen", "beverton_holt", "myers", "may", "ricker"]

    def run_case(name, env_id, kwargs, seeds, nsteps, action_fn, auto_reset=True,
                 init_reset=True, out=None):
        """One reference env per seed, driven `nsteps` steps.  On done the driver
        calls reset() (what SB3's DummyVecEnv does around the reference) when
        auto_reset, else keeps stepping the finished env (reference allows it)."""
        E = len(seeds)
        is_v0 = env_id == "fishing-v0"
        is_v4 = env_id == "fishing-v4"
        A = np.zeros((E, nsteps), dtype=np.int32 if is_v0 else np.float32)
        Z = np.zeros((E, nsteps))
        OBS_IN = np.zeros((E, nsteps))
        OBS = np.zeros((E, nsteps))
        REW = np.zeros((E, nsteps))
        DONE = np.zeros((E, nsteps), dtype=np.uint8)
        T = np.zeros((E, nsteps), dtype=np.int32)
        KK = np.zeros((E, nsteps))
        RR = np.zeros((E, nsteps))
        ZK = np.full((E, nsteps + 1), np.nan)  # reset draws; column 0 = initial reset
        ZR = np.full((E, nsteps + 1), np.nan)
        RESET_OBS = np.full((E, nsteps + 1), np.nan)
        RP = np.full((E, nsteps), np.nan)                 # params["r"] in force (fishing-v10 drifts it)
        MID = np.full((E, nsteps), -1, dtype=np.int32)    # fishing-v11: index of the model in force
        out = out_default if out is None else out
        for e, seed in enumerate(seeds):
            np.random.seed(seed)
            env = gym.make(env_id, **kwargs)
            arng = np.random.RandomState(10_000 + seed)  # action stream, separate from env noise

            def do_reset(col):
                if is_v4:
                    st = np.random.get_state()
                    ZK[e, col] = np.random.normal(0, 1)
                    ZR[e, col] = np.random.normal(0, 1)
                    np.random.set_state(st)
                o = env.reset()
                RESET_OBS[e, col] = o[0]

            if init_reset:
                do_reset(0)
            for s in range(nsteps):
                a = action_fn(arng, s, e)
                A[e, s] = a
                Z[e, s] = _peek_normal(np)
                OBS_IN[e, s] = env.state[0]
                KK[e, s] = env.K
                RR[e, s] = env.r
                if isinstance(env.params.get("r", None), (int, float)):
                    RP[e, s] = env.params["r"]
                if hasattr(env, "model"):
                    MID[e, s] = ZOO_MODELS.index(env.model)
                obs, rew, done, info = env.step(as_action(env_id, A[e, s]))
                OBS[e, s] = obs[0]
                REW[e, s] = rew
                DONE[e, s] = done
                T[e, s] = env.years_passed
                if done and auto_reset:
                    do_reset(s + 1)
        kw = {k: v for k, v in kwargs.items()}
        out[name + "/meta"] = np.array(json.dumps(
            {"id": env_id, "kwargs": kw, "seeds": list(seeds), "nsteps": nsteps,
             "auto_reset": auto_reset, "init_reset": init_reset}))
        for k, v in (("action", A), ("z", Z), ("obs_in", OBS_IN), ("obs", OBS), ("reward", REW),
                     ("done", DONE), ("t", T), ("K", KK), ("r", RR), ("zK", ZK), ("zr", ZR),
                     ("reset_obs", RESET_OBS), ("params_r", RP), ("model_idx", MID)):
            out[name + "/" + k] = v
        return OBS, REW, DONE

    f32 = np.float32
    # --- BASELINE config 1: fishing-v1, sigma=0, single env, constant dyadic action
    o, r, d = run_case("v1_sigma0_const", "fishing-v1", {"sigma": 0.0}, [0], 101,
                       lambda g, s, e: -0.9375, auto_reset=False)
    anchors["v1_sigma0_const"] = {
        "first5_obs_hex": [float(x).hex() for x in o[0, :5]],
        "return": float(r[0].sum()), "final_obs": float(o[0, -1]),
        "n_steps_to_done": int(np.argmax(d[0]) + 1)}

    # --- fishing-v1 sigma=0.1, random policy U[-1,1) float32 (BASELINE config 2 at toy N)
    run_case("v1_sigma01_random", "fishing-v1", {"sigma": 0.1}, list(range(1, 9)), 130,
             lambda g, s, e: f32(g.uniform(-1, 1)))
    # conservative policy: keeps the stock alive for whole 101-step episodes
    run_case("v1_sigma01_low", "fishing-v1", {"sigma": 0.1}, list(range(11, 17)), 210,
             lambda g, s, e: f32(g.uniform(-1, -0.8)))
    # non-default parameters
    run_case("v1_params", "fishing-v1", {"r": 0.5, "K": 2.0, "sigma": 0.2, "init_state": 1.1,
                                          "Tmax": 17},
             list(range(21, 27)), 60, lambda g, s, e: f32(g.uniform(-1, -0.5)))
    # clip + harvest-all + stepping a finished env without reset (quirk B7)
    edge_actions = [-5.0, -1.0, -0.9375, 0.0, 5.0, 1.0, -0.5, -1.0, 0.25, -0.75]
    run_case("v1_edge_noreset", "fishing-v1", {"sigma": 0.1, "Tmax": 6}, [31, 32], 10,
             lambda g, s, e: f32(edge_actions[s]), auto_reset=False)
    # non-finite and huge actions: np.clip passes NaN through, Python's min(x, nan) keeps x (everything is
    # harvested), +-inf / +-1e30 clip to the Box bounds
    special = [float("nan"), -0.9, float("inf"), -0.95, float("-inf"), -0.9, float("nan"), float("nan"), 0.0, -1.0,
               1e30, -1e30]
    run_case("v1_special_actions", "fishing-v1", {"sigma": 0.1, "Tmax": 9}, [33, 34], 12,
             lambda g, s, e: f32(special[s]))
    # constructor corner values: a stock that starts extinct (done on every first step), a horizon of zero years (done after one
    # step whatever the stock), a negative growth rate
    run_case("v1_starts_extinct", "fishing-v1", {"sigma": 0.1, "init_state": 0.0}, [35, 36], 6,
             lambda g, s, e: f32(g.uniform(-1, 0)))
    run_case("v1_Tmax0", "fishing-v1", {"sigma": 0.1, "Tmax": 0}, [37, 38], 6,
             lambda g, s, e: f32(g.uniform(-1, -0.8)))
    run_case("v1_negative_r", "fishing-v1", {"sigma": 0.1, "r": -0.4, "Tmax": 12}, [39, 40], 30,
             lambda g, s, e: f32(g.uniform(-1, -0.8)))
    # --- fishing-v0 (BASELINE config 3 at toy N)
    run_case("v0_sigma01_random", "fishing-v0", {"sigma": 0.1}, list(range(41, 49)), 130,
             lambda g, s, e: g.randint(0, 100))
    run_case("v0_sigma0_low", "fishing-v0", {"sigma": 0.0, "n_actions": 64}, [51, 52], 105,
             lambda g, s, e: g.randint(0, 8))
    run_case("v0_edge", "fishing-v0", {"sigma": 0.1, "Tmax": 8}, [53], 12,
             lambda g, s, e: [0, 10, 99, 0, 100, 150, 3, 7, 1, 0, 2, 5][s], auto_reset=True)
    # one action only (every index >= 1 is a quota >= K: the whole stock), a thousand actions (a finer quota grid than the default 100)
    run_case("v0_one_action", "fishing-v0", {"sigma": 0.1, "n_actions": 1, "Tmax": 7}, [54, 55], 12,
             lambda g, s, e: [0, 0, 0, 1, 0, 0, 2, 0, 0, 0, 0, 1][s])
    run_case("v0_thousand_actions", "fishing-v0", {"sigma": 0.1, "n_actions": 1000, "Tmax": 30}, [56, 57, 58], 40,
             lambda g, s, e: g.randint(0, 300))
    # --- fishing-v2 tipping point (BASELINE config 4 at toy N)
    run_case("v2_sigma01_low", "fishing-v2", {"sigma": 0.1}, list(range(61, 69)), 130,
             lambda g, s, e: f32(g.uniform(-1, -0.8)))
    run_case("v2_sigma01_random", "fishing-v2", {"sigma": 0.1, "C": 0.4}, list(range(71, 77)), 60,
             lambda g, s, e: f32(g.uniform(-1, 1)))
    run_case("v2_sigma0_zeroquota", "fishing-v2", {"sigma": 0.0, "init_state": 0.75}, [0], 3,
             lambda g, s, e: f32(-1.0), auto_reset=False)
    run_case("v2_sigma0_zeroquota_low", "fishing-v2", {"sigma": 0.0, "init_state": 0.3}, [0], 3,
             lambda g, s, e: f32(-1.0), auto_reset=False)
    # the tipping point at zero (no tipping: the factor (x - C) never changes sign) and above the carrying capacity (every stock below it shrinks)
    run_case("v2_C_zero", "fishing-v2", {"sigma": 0.1, "C": 0.0, "Tmax": 20}, [78, 79], 30,
             lambda g, s, e: f32(g.uniform(-1, -0.7)))
    run_case("v2_C_above_K", "fishing-v2", {"sigma": 0.1, "C": 1.5, "Tmax": 20}, [80, 81], 30,
             lambda g, s, e: f32(g.uniform(-1, -0.7)))
    # --- fishing-v4 per-episode parameter uncertainty (BASELINE config 5 at toy N)
    run_case("v4_sigma005", "fishing-v4", {"sigma": 0.05, "sigma_p": 0.1}, list(range(81, 89)), 240,
             lambda g, s, e: f32(g.uniform(-1, -0.7)))
    run_case("v4_random", "fishing-v4", {"sigma": 0.1, "sigma_p": 0.3, "K_mean": 1.5,
                                          "r_mean": 0.4, "init_state": 0.6, "Tmax": 12},
             list(range(91, 97)), 80, lambda g, s, e: f32(g.uniform(-1, 0.2)))
    # the draws' clip bounds (fishing_model_error.py:37-38, 41-42: np.clip(N(mean, sigma_p), 0, 1e6)).  A mean far below zero pins the
    # drawn value to exactly 0: with K = 0 the stock is 0 * (1 - 0 / 0) = NaN from the first step on, the reward 0 and `done` only ever
    # the year counter's; with r = 0 the stock only shrinks; a mean far above 1e6 pins K to 1e6
    run_case("v4_K_clipped_to_zero", "fishing-v4", {"sigma": 0.05, "K_mean": -5.0, "Tmax": 6}, [101, 102, 103], 16,
             lambda g, s, e: f32(g.uniform(-1, 0.5)))
    run_case("v4_r_clipped_to_zero", "fishing-v4", {"sigma": 0.05, "r_mean": -5.0, "Tmax": 6}, [104, 105, 106], 16,
             lambda g, s, e: f32(g.uniform(-1, -0.6)))
    run_case("v4_K_clipped_to_1e6", "fishing-v4", {"sigma": 0.05, "K_mean": 3.0e6, "sigma_p": 0.2, "Tmax": 6}, [107, 108], 16,
             lambda g, s, e: f32(g.uniform(-1, -0.6)))
    # v4 before the first reset(): constructor draws, obs = x0/K_mean - 1 (quirk a9)
    run_case("v4_noinitreset", "fishing-v4", {"sigma": 0.05}, [97, 98], 20,
             lambda g, s, e: f32(g.uniform(-1, -0.7)), init_reset=False)

    # --- growth-model zoo fishing-v5..v11 (growth_models.py), lognormal noise
    zoo = {}
    low = lambda g, s, e: f32(g.uniform(-1, -0.8))      # noqa: E731
    mid = lambda g, s, e: f32(g.uniform(-1, 0.0))       # noqa: E731
    for tag, env_id, kw, af, n in (
            ("v5_allen", "fishing-v5", {"sigma": 0.1}, low, 120),
            ("v5_allen_s0", "fishing-v5", {"sigma": 0.0, "C": 0.3, "r": 0.5}, mid, 40),
            ("v6_bh", "fishing-v6", {"sigma": 0.1}, low, 120),
            ("v6_bh_params", "fishing-v6", {"sigma": 0.05, "r": 0.6, "K": 2.0, "init_state": 1.0, "Tmax": 20}, mid, 70),
            ("v7_may", "fishing-v7", {"sigma": 0.1}, low, 120),
            ("v7_may_s0", "fishing-v7", {"sigma": 0.0}, mid, 60),
            ("v8_myers", "fishing-v8", {"sigma": 0.1}, low, 120),
            ("v8_myers_s0", "fishing-v8", {"sigma": 0.0, "theta": 2.5, "M": 1.2}, mid, 60),
            ("v9_ricker", "fishing-v9", {"sigma": 0.1}, low, 120),
            ("v9_ricker_params", "fishing-v9", {"sigma": 0.2, "r": 0.8, "K": 1.5, "Tmax": 15}, mid, 60),
            ("v10_nonstat", "fishing-v10", {"sigma": 0.05, "alpha": -0.007}, low, 230),
            ("v10_nonstat_s0", "fishing-v10", {"sigma": 0.0, "alpha": -0.02, "r": 0.5, "Tmax": 40}, low, 90),
            ("v11_uncert", "fishing-v11", {}, low, 330),
            ("v11_uncert_T", "fishing-v11", {"Tmax": 6}, mid, 100)):
        run_case(tag, env_id, kw, list(range(200, 206)), n, af, out=zoo)
    # the zoo under the non-finite and huge actions of v1_special_actions (round 4): a NaN quota harvests the whole stock
    # (Python's min(x, nan) keeps x), the growth functions then see an extinct stock -- log(0) = -inf, exp(-inf) = 0
    for tag, env_id, kw in (("v9_ricker_special_actions", "fishing-v9", {"sigma": 0.1, "Tmax": 9}),
                            ("v7_may_special_actions", "fishing-v7", {"sigma": 0.1, "Tmax": 9}),
                            ("v8_myers_special_actions", "fishing-v8", {"sigma": 0.1, "Tmax": 9}),
                            ("v11_uncert_special_actions", "fishing-v11", {"Tmax": 9})):
        run_case(tag, env_id, kw, [233, 234], 12, lambda g, s, e: f32(special[s]), out=zoo)
    # corner parameters (round 5): Myers with r < -1 (log(r + 1) is NaN: the stock is NaN from the first step), Beverton-Holt with
    # r = 0 (B = K / r: a division by zero inside the reference), Ricker with a negative rate, May with a non-integer exponent
    for tag, env_id, kw in (("v8_myers_r_below_minus_one", "fishing-v8", {"sigma": 0.1, "r": -1.5, "Tmax": 6}),
                            ("v6_bh_r_zero", "fishing-v6", {"sigma": 0.05, "r": 0.0, "Tmax": 8}),
                            ("v9_ricker_negative_r", "fishing-v9", {"sigma": 0.1, "r": -0.3, "Tmax": 10}),
                            ("v7_may_q_two_and_a_half", "fishing-v7", {"sigma": 0.05, "q": 2.5, "Tmax": 12})):
        run_case(tag, env_id, kw, [235, 236], 16, low, out=zoo)
    # --- the module-level growth functions themselves (growth_models.py:208-269), called the way a user of the
    # reference may: f(x, params) on a scalar, a vector and a matrix of populations (zero, tiny, typical, far above K,
    # negative), default and non-default parameter dicts, sigma = 0 and > 0.  np.random.lognormal(mu, sigma) consumes one
    # legacy standard normal per element of mu: recorded by drawing them first and rewinding the stream.
    from gym_fishing.envs import growth_models as gm
    grow = {}
    PKEYS = ("r", "K", "sigma", "C", "M", "theta", "q", "b", "a")
    grid = np.concatenate([[0.0, 1e-300, 1e-12, 1e-3], np.linspace(0.02, 3.0, 37), [7.5, 50.0, 1e6, -0.5]])
    for tag, fname, P in (
            ("allen", "allen", {"r": 0.3, "K": 1.0, "sigma": 0.0, "C": 0.5}),
            ("allen_noise", "allen", {"r": 0.9, "K": 2.0, "sigma": 0.15, "C": 0.2}),
            ("beverton_holt", "beverton_holt", {"r": 0.3, "K": 1, "sigma": 0.0}),
            ("beverton_holt_noise", "beverton_holt", {"r": 0.6, "K": 2.5, "sigma": 0.2}),
            ("beverton_holt_r0", "beverton_holt", {"r": -0.1, "K": 1.0, "sigma": 0.1}),      # clip(r, 0, inf): B = inf
            ("myers", "myers", {"r": 1.0, "K": 1.0, "M": 1.0, "theta": 3.0, "sigma": 0.0}),
            ("myers_noise", "myers", {"r": 0.7, "K": 1.0, "M": 1.3, "theta": 2.5, "sigma": 0.1}),
            ("may", "may", {"r": 0.7, "K": 1.5, "M": 1.5, "q": 3, "b": 0.15, "sigma": 0.0, "a": 0.2}),
            ("may_noise", "may", {"r": 0.5, "K": 1.5, "M": 1.2, "q": 2, "b": 0.2, "sigma": 0.1, "a": 0.1}),
            ("ricker", "ricker", {"r": 0.3, "K": 1, "sigma": 0.0}),
            ("ricker_noise", "ricker", {"r": 1.1, "K": 0.8, "sigma": 0.25})):
        f = gm.population_model[fname]
        seed = 4000 + len(grow)
        np.random.seed(seed)            # the three calls below run on from here, in this order
        for shape_tag, x in (("vector", grid), ("matrix", grid[4:40].reshape(4, 9)), ("scalar", np.float64(0.62))):
            st = np.random.get_state()
            z = np.random.normal(0, 1, np.shape(x))
            np.random.set_state(st)
            y = f(x, P)
            grow["%s/%s/x" % (tag, shape_tag)] = np.asarray(x, dtype=np.float64)
            grow["%s/%s/z" % (tag, shape_tag)] = np.asarray(z, dtype=np.float64)
            grow["%s/%s/out" % (tag, shape_tag)] = np.asarray(y, dtype=np.float64)
        grow[tag + "/kind"] = np.array(ZOO_MODELS.index(fname), dtype=np.int32)
        grow[tag + "/seed"] = np.array(seed, dtype=np.int64)
        grow[tag + "/params"] = np.array([float(P.get(k, np.nan)) for k in PKEYS])
    np.savez_compressed(os.path.join(OUT, "reference_growth_functions.npz"), **grow)

    # --- anchors from the reference's own test (tests/test-envs.py:93-106)
    env = gym.make("fishing-v2", sigma=0, init_state=0.75)
    env.reset()
    obs, _, _, _ = env.step(env.get_action(0))
    hi = float(env.get_fish_population(obs))
    env.init_state = 0.3
    env.reset()
    obs, _, _, _ = env.step(env.get_action(0))
    lo = float(env.get_fish_population(obs))
    anchors["test_tipping"] = {"from_0.75": hi, "from_0.3": lo}
    assert hi >= 0.75 and lo <= 0.3

    # --- "next" rows: BMSY / msy / escapement known answers, sigma=0 simulate tables.
    # The policies hand step() a Python / NumPy float; with the reference's pinned NumPy 1.19
    # np.clip against the float32 Box bounds returns float32 and the quota is then formed in
    # float64 (Appendix A.3).  NumPy 2 would keep float64 through the clip, so the policy is
    # wrapped to round its continuous action to float32 and pass it on as a float64 array --
    # the same "float32 value, float64 arithmetic" convention as the trajectory fixtures.
    class RefEra:
        def __init__(self, model, discrete):
            self.model, self.discrete = model, discrete

        def predict(self, obs, **kw):
            a, st = self.model.predict(obs, **kw)
            if not self.discrete:
                a = np.array([a], dtype=np.float32).astype(np.float64)
            return a, st

    sims = {}
    for env_id, kw in (("fishing-v0", {}), ("fishing-v1", {}), ("fishing-v2", {}),
                       ("fishing-v1", {"r": 0.5, "K": 2.0, "init_state": 1.1}),
                       ("fishing-v0", {"n_actions": 37, "r": 0.4})):
        tag = env_id[-2:] + ("_params" if kw else "")
        env = gym.make(env_id, sigma=0.0, **kw)
        S = float(BMSY(env))
        m = msy(env)
        anchors["policy_" + tag] = {"BMSY": S, "msy": float(m.msy), "kwargs": kw}
        for pname, model in (("msy", m), ("escapement", escapement(env))):
            df = env.simulate(RefEra(model, env_id == "fishing-v0"), reps=1)
            sims["sim_%s_%s" % (tag, pname)] = df.to_numpy(dtype=np.float64)
            anchors["policy_" + tag]["sum_reward_" + pname] = float(df.reward.sum())
            anchors["policy_" + tag]["rows_" + pname] = int(len(df))
    # --- seeded flows at sigma > 0: np.random.seed(s); env = make(...); model = policy(env) (BMSY's and msy's
    # population_draw() calls consume the global stream too); df = env.simulate(model, reps=2).  What a user's
    # script does; the drop-in's scalar protocol must give the same table from the same seed.
    seeded = {}
    for env_id, kw in (("fishing-v1", {"sigma": 0.1}), ("fishing-v0", {"sigma": 0.1}), ("fishing-v2", {"sigma": 0.05}),
                       ("fishing-v5", {"sigma": 0.1}), ("fishing-v9", {"sigma": 0.1}), ("fishing-v11", {}),
                       # (round 4: the rest of the zoo -- fishing-v10's r drifts on EVERY population_draw(), BMSY's and msy's included)
                       ("fishing-v6", {"sigma": 0.1}), ("fishing-v7", {"sigma": 0.1}), ("fishing-v8", {"sigma": 0.1}),
                       ("fishing-v10", {"sigma": 0.05})):
        for pname, cls in (("msy", msy), ("escapement", escapement)):
            np.random.seed(7)
            env = gym.make(env_id, **kw)
            model = cls(env)
            df = env.simulate(RefEra(model, env_id == "fishing-v0"), reps=2)
            key = "%s_%s" % (env_id.replace("fishing-", ""), pname)
            seeded[key + "/table"] = df.to_numpy(dtype=np.float64)
            seeded[key + "/meta"] = np.array(json.dumps({"id": env_id, "kwargs": kw, "policy": pname, "seed": 7, "reps": 2,
                                                         "S": float(model.S),
                                                         "msy": float(model.msy) if pname == "msy" else None}))
    # fishing-v4 (round 3).  Its BMSY sweep multiplies the float32 observation grid by np.float64 scalars (the drawn K, r),
    # which NumPy 2 evaluates in float64 and the reference's NumPy 1.19 in float32: no version-neutral table of THAT.
    # Asked for a float64 grid -- as the policyfn fixtures below do -- every product is float64 under both, and the table
    # pins the flow itself: the constructor's and every reset()'s K-then-r draws, BMSY's and msy's population_draw()
    # draws, all from the one global stream, and the table's rows on the (K, r) of each episode.
    for pname, cls in (("msy", msy), ("escapement", escapement)):
        np.random.seed(7)
        kw = {"sigma": 0.05, "sigma_p": 0.1}
        env = gym.make("fishing-v4", **kw)
        env.observation_space.dtype = np.dtype(np.float64)
        model = cls(env)
        df = env.simulate(RefEra(model, False), reps=2)
        seeded["v4_%s/table" % pname] = df.to_numpy(dtype=np.float64)
        seeded["v4_%s/meta" % pname] = np.array(json.dumps({"id": "fishing-v4", "kwargs": kw, "policy": pname, "seed": 7,
                                                           "reps": 2, "S": float(model.S),
                                                           "msy": float(model.msy) if pname == "msy" else None}))
    np.savez_compressed(os.path.join(OUT, "reference_seeded_sims.npz"), **seeded)
    # --- estimate_policyfn (shared_env.py:82-102): the policy's quota over a grid of 50 observations, through the
    # reference's own env.policyfn(); same action convention as the tables above (RefEra).
    pf = {}
    for env_id, kw in (("fishing-v1", {}), ("fishing-v0", {}), ("fishing-v1", {"r": 0.5, "K": 2.0, "init_state": 1.1}),
                       ("fishing-v0", {"n_actions": 37, "r": 0.4})):
        tag = env_id[-2:] + ("_params" if kw else "")
        for pname, cls in (("msy", msy), ("escapement", escapement)):
            env = gym.make(env_id, sigma=0.0, **kw)
            model = RefEra(cls(env), env_id == "fishing-v0")
            # The grid's dtype comes from the observation Box (float32), and `float32 scalar + Python int` is float32
            # under NumPy 2 but float64 under the reference's NumPy 1.19: asking for a float64 grid makes every sum
            # float64 under both, so the table pins the function (grid, predict, population and quota maps, row
            # layout) and not a NumPy version.
            env.observation_space.dtype = np.dtype(np.float64)
            df = env.policyfn(model, reps=2)
            pf["policyfn_%s_%s" % (tag, pname)] = df.to_numpy(dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "reference_policyfn.npz"), **pf)

    # --- simulate_mdp_vec (shared_env.py:57-79): the reference's only code written against an N-env object.  It is
    # driven UNMODIFIED over N reference envs behind a minimal harness with the SB3-DummyVecEnv behaviour it relies
    # on (envs stepped in order, a finished env reset at once and its post-reset observation returned, env_method /
    # get_attr fan-out).  The policy adapter calls the reference policy per env and keeps the fixtures' action
    # convention (float32 value, float64 arithmetic).  Recorded: the table (Tmax + 1 rows per env and rep, no break
    # on done, raw action of the previous step in the action column).
    from gym_fishing.envs.shared_env import simulate_mdp_vec

    class MiniVecEnv:
        def __init__(self, envs):
            self.envs, self.num_envs = envs, len(envs)
            self.action_space = envs[0].action_space

        def reset(self):
            return np.stack([e.reset() for e in self.envs])

        def step(self, actions):
            obs, rews, dones, infos = [], [], [], []
            for e, a in zip(self.envs, actions):
                o, r, d, info = e.step(a)
                if d:
                    info = dict(info, terminal_observation=o)
                    o = e.reset()
                obs.append(o)
                rews.append(r)
                dones.append(d)
                infos.append(info)
            return np.stack(obs), np.array(rews), np.array(dones), infos

        def env_method(self, name, *args, indices=None, **kw):
            idx = range(self.num_envs) if indices is None else ([indices] if isinstance(indices, int) else indices)
            return [getattr(self.envs[i], name)(*args, **kw) for i in idx]

        def get_attr(self, name, indices=None):
            idx = range(self.num_envs) if indices is None else ([indices] if isinstance(indices, int) else indices)
            return [getattr(self.envs[i], name) for i in idx]

    class VecPolicy:
        def __init__(self, model):
            self.model = model

        def predict(self, obs, state=None, mask=None):
            acts = [np.array([self.model.predict(o)[0]], dtype=np.float32).astype(np.float64) for o in obs]
            return np.stack(acts), state

    class constant:                     # a fixed action: fishes the stock out every other step -> auto-resets mid-table
        S = float("nan")

        def __init__(self, env, a=-0.45):
            self.a = a

        def predict(self, obs, **kw):
            return self.a, obs

    vec = {}
    for env_id, kw, pname, cls, n_envs, episodes in (
            ("fishing-v1", {"sigma": 0.1, "Tmax": 15}, "constant", constant, 4, 4),
            ("fishing-v1", {"sigma": 0.1, "Tmax": 15}, "escapement", escapement, 3, 6),
            ("fishing-v1", {"sigma": 0.1, "Tmax": 15}, "msy", msy, 3, 3),
            ("fishing-v1", {"sigma": 0.2, "Tmax": 9, "r": 0.5, "K": 2.0, "init_state": 1.1}, "escapement", escapement, 4, 8),
            ("fishing-v2", {"sigma": 0.05, "Tmax": 12}, "escapement", escapement, 3, 6),
            # (round 4) the zoo through the same helper.  Its growth functions read params["sigma"], not the env.sigma BMSY()
            # zeroes, so the sweep IS noisy and S depends on the stream: seeded on its own
            ("fishing-v5", {"sigma": 0.1, "Tmax": 12}, "escapement", escapement, 3, 6),
            ("fishing-v9", {"sigma": 0.1, "Tmax": 12}, "msy", msy, 3, 3),
            ("fishing-v7", {"sigma": 0.05, "Tmax": 10}, "escapement", escapement, 4, 4)):
        envs = [gym.make(env_id, **kw) for _ in range(n_envs)]
        if env_id in ("fishing-v5", "fishing-v7", "fishing-v9"):
            np.random.seed(5)
        model = cls(envs[0])            # (logistic / tipping: the sweep runs at sigma = 0, S and msy do not depend on the stream)
        np.random.seed(11)              # from here on the N envs share the global stream, stepped in order
        df = simulate_mdp_vec(MiniVecEnv(envs), VecPolicy(model), n_eval_episodes=episodes)
        key = "%s_%s_%d" % (env_id.replace("fishing-", ""), pname, len(vec) // 2)
        vec[key + "/table"] = df.to_numpy(dtype=np.float64)
        vec[key + "/meta"] = np.array(json.dumps({"id": env_id, "kwargs": kw, "policy": pname, "seed": 11, "num_envs": n_envs,
                                                  "n_eval_episodes": episodes, "S": float(model.S),
                                                  "msy": float(model.msy) if pname == "msy" else None}))
    np.savez_compressed(os.path.join(OUT, "reference_vec_sims.npz"), **vec)

    # fishing-v4 through the same helper: K is redrawn at every reset, and df_entry_vec (shared_env.py:15-26) asks the
    # env itself for each row's population -- so a row after an auto-reset inside the table uses the NEW K.  The harness
    # logs the observation and the env's K at every get_fish_population call next to the table.
    class LoggingVecEnv(MiniVecEnv):
        def __init__(self, envs):
            super().__init__(envs)
            self.log = []

        def env_method(self, name, *args, indices=None, **kw):
            if name == "get_fish_population":
                i = indices if isinstance(indices, int) else indices[0]
                self.log.append((float(np.asarray(args[0]).reshape(-1)[0]), float(self.envs[i].K)))
            return super().env_method(name, *args, indices=indices, **kw)

    kw4 = {"sigma": 0.05, "sigma_p": 0.2, "Tmax": 7}
    envs = [gym.make("fishing-v4", **kw4) for _ in range(3)]
    np.random.seed(11)
    lv = LoggingVecEnv(envs)
    df = simulate_mdp_vec(lv, VecPolicy(constant(envs[0], a=0.2)), n_eval_episodes=6)
    v4 = {"v4_constant/table": df.to_numpy(dtype=np.float64),
          "v4_constant/obs_rows": np.array([o for o, _ in lv.log], dtype=np.float64),
          "v4_constant/K_rows": np.array([k for _, k in lv.log], dtype=np.float64),
          "v4_constant/meta": np.array(json.dumps({"id": "fishing-v4", "kwargs": kw4, "policy": "constant", "seed": 11,
                                                   "num_envs": 3, "n_eval_episodes": 6, "S": float("nan"), "msy": None}))}
    np.savez_compressed(os.path.join(OUT, "reference_vec_sims_v4.npz"), **v4)
    # get_action / get_quota round trips (base_fishing_env.py:135-156)
    env0 = gym.make("fishing-v0")
    env1 = gym.make("fishing-v1")
    q = np.linspace(0.0, 1.0, 21)
    anchors["get_action_v0"] = [int(env0.get_action(x)) for x in q]
    anchors["get_action_v1"] = [float(env1.get_action(x)) for x in q]
    anchors["get_quota_v0"] = [float(env0.get_quota(int(a))) for a in range(0, 101, 5)]
    anchors["np_seed42_normals"] = [float(x) for x in np.random.RandomState(42).normal(0, 1, 3)]
    anchors["versions"] = {"numpy": np.__version__, "python": sys.version.split()[0],
                           "reference_version": open(os.path.join(REF, "gym_fishing/version.txt")).read().strip()}

    np.savez_compressed(os.path.join(OUT, "reference_trajectories.npz"), **out)
    np.savez_compressed(os.path.join(OUT, "reference_policy_sims.npz"), **sims)
    np.savez_compressed(os.path.join(OUT, "reference_zoo_trajectories.npz"), **zoo)
    with open(os.path.join(OUT, "reference_anchors.json"), "w") as f:
        json.dump(anchors, f, indent=1, sort_keys=True)
    print("wrote %d arrays, %d sims, %d anchors" % (len(out), len(sims), len(anchors)))
    return 0


if __name__ == "__main__":
    if "--out" in sys.argv:
        OUT = sys.argv[sys.argv.index("--out") + 1]
    sys.exit(main())
