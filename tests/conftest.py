import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class GoldenCase:
    """One case of tests/golden/reference_trajectories.npz (captured from the reference)."""

    def __init__(self, name, npz):
        self.name = name
        self.meta = json.loads(str(npz[name + "/meta"]))
        for k in ("action", "z", "obs_in", "obs", "reward", "done", "t", "K", "r", "zK", "zr",
                  "reset_obs", "params_r", "model_idx"):
            setattr(self, k, npz[name + "/" + k])
        self.id = self.meta["id"]
        self.kwargs = self.meta["kwargs"]
        self.nsteps = self.meta["nsteps"]
        self.auto_reset = self.meta["auto_reset"]
        self.init_reset = self.meta["init_reset"]

    def param(self, key):
        defaults = {"r": 0.3, "K": 1.0, "sigma": 0.0, "init_state": 0.75, "Tmax": 100,
                    "n_actions": 100, "C": 0.5, "K_mean": 1.0, "r_mean": 0.3, "sigma_p": 0.1}
        return self.kwargs.get(key, defaults[key])


def load_zoo_cases():
    npz = np.load(os.path.join(GOLDEN, "reference_zoo_trajectories.npz"))
    names = sorted({k.split("/")[0] for k in npz.files})
    return [GoldenCase(n, npz) for n in names]


GROWTH_PARAM_KEYS = ("r", "K", "sigma", "C", "M", "theta", "q", "b", "a")
GROWTH_FUNCTION_OF_KIND = ("allen", "beverton_holt", "myers", "may", "ricker")


def load_growth_function_cases():
    """Reference f(x, params) calls of growth_models.py:208-269 (tests/golden/make_golden.py): per case the function, its
    parameter dict, the seed set before the three calls (vector, matrix, scalar -- in that order) and per call the
    populations, the standard normals the reference consumed, and its result."""
    z = np.load(os.path.join(GOLDEN, "reference_growth_functions.npz"))
    out = []
    for tag in sorted({k.split("/")[0] for k in z.files}):
        params = {k: float(v) for k, v in zip(GROWTH_PARAM_KEYS, z[tag + "/params"]) if not np.isnan(v)}
        calls = [(sh, z["%s/%s/x" % (tag, sh)], z["%s/%s/z" % (tag, sh)], z["%s/%s/out" % (tag, sh)])
                 for sh in ("vector", "matrix", "scalar")]
        out.append(dict(tag=tag, kind=int(z[tag + "/kind"]), name=GROWTH_FUNCTION_OF_KIND[int(z[tag + "/kind"])],
                        seed=int(z[tag + "/seed"]), params=params, calls=calls))
    return out


def load_golden_cases():
    npz = np.load(os.path.join(GOLDEN, "reference_trajectories.npz"))
    names = sorted({k.split("/")[0] for k in npz.files})
    return [GoldenCase(n, npz) for n in names]


@pytest.fixture(scope="session")
def golden_cases():
    return load_golden_cases()


@pytest.fixture(scope="session")
def anchors():
    with open(os.path.join(GOLDEN, "reference_anchors.json")) as f:
        return json.load(f)


def load_policy_sims():
    """Reference env.simulate(msy / escapement) tables at sigma = 0 (tests/golden/make_golden.py)."""
    with open(os.path.join(GOLDEN, "reference_anchors.json")) as f:
        anchors = json.load(f)
    sims = np.load(os.path.join(GOLDEN, "reference_policy_sims.npz"))
    out = []
    for key in sorted(sims.files):
        parts = key.split("_")
        pname, tag = parts[-1], "_".join(parts[1:-1])
        a = anchors["policy_" + tag]
        kw = a["kwargs"]
        out.append(dict(key=key, env_id="fishing-" + tag[:2], policy=pname,
                        param=a["BMSY"] if pname == "escapement" else a["msy"], table=sims[key][:, :4],
                        K=float(kw.get("K", 1.0)), r=float(kw.get("r", 0.3)), x0=float(kw.get("init_state", 0.75)),
                        n_actions=int(kw.get("n_actions", 100))))
    return out


def load_seeded_sims():
    """Reference flows at sigma > 0 (tests/golden/make_golden.py): np.random.seed(7); env = make(id, **kw);
    model = msy(env) / escapement(env); env.simulate(model, reps=2) -> the table, the policy's S and msy."""
    z = np.load(os.path.join(GOLDEN, "reference_seeded_sims.npz"))
    out = []
    for key in sorted(k[:-5] for k in z.files if k.endswith("/meta")):
        meta = json.loads(str(z[key + "/meta"]))
        out.append(dict(key=key, table=z[key + "/table"], **meta))
    return out


def load_vec_sims():
    """Reference simulate_mdp_vec tables (tests/golden/make_golden.py): N reference envs behind a DummyVecEnv-shaped
    harness, np.random.seed(11) right before the call."""
    z = np.load(os.path.join(GOLDEN, "reference_vec_sims.npz"))
    out = []
    for key in sorted(k[:-5] for k in z.files if k.endswith("/meta")):
        meta = json.loads(str(z[key + "/meta"]))
        out.append(dict(key=key, table=z[key + "/table"], **meta))
    return out
