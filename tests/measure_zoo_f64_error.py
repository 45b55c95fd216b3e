#!/usr/bin/env python3
"""How far the float64 parity layout of the growth-model zoo (fishing-v5..v11) lands from the reference's arithmetic -- and
from the exact value.  Run ON THE GPU BOX, once per library variant (FISHING_HIP_LIB selects it; `--tag` names it):

    python3 tests/measure_zoo_f64_error.py --tag algebraic >> gpurun_out/zoo_f64_error.jsonl

One JSON line per measurement:
  * "golden": every recorded step of the reference-held fixtures (tests/golden/reference_zoo_trajectories.npz) through
    fishing_step_f64 -> max relative population error against the reference's numbers (the layout's bar: 2e-14).
  * "sweep": 2^20 random (x in [0, 2K], z ~ N(0, 1)) per growth function through fishing_population_draw_f64 against the
    float64 oracle (the reference's log / exp round trip in NumPy) -> max relative error, in units of 2e-14 too.
  * "exact": 20000 of those points evaluated with mpmath at 50 digits -> max relative error of the DEVICE and of the ORACLE
    (= the reference's arithmetic) against the exact value, in float64 ulps of the result: which of the two is closer.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # (lives under tests/: it checks the product against the oracle)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import hip_harness as hh  # noqa: E402
from conftest import load_zoo_cases  # noqa: E402
from gym_fishing_amd import _capi  # noqa: E402
from oracle import fishing_oracle as fo  # noqa: E402
from test_gpu_zoo import hip_params, t_in_of, zoo_kw  # noqa: E402
from test_oracle_golden import ZOO_DEFAULTS  # noqa: E402

KIND_NAMES = ["allen", "beverton_holt", "myers", "may", "ricker"]
BAR = 2e-14


def golden(tag):
    for c in load_zoo_cases():
        model = fo.MODEL_OF_ID[c.id]
        K = float(zoo_kw(c)["K"])
        n = c.obs.size
        st = hh.State(n, np.float64, model, c.obs_in.reshape(-1), t=t_in_of(c).reshape(-1),
                      r=c.params_r.reshape(-1) if model == fo.MODEL_V10 else None,
                      model_idx=c.model_idx.reshape(-1) if model == fo.MODEL_V11 else None)
        obs, rew, done, t = st.step(hip_params(hh, c), c.action.reshape(-1), z=c.z.reshape(-1))
        ref = c.obs.reshape(-1)
        x = (ref + 1.0) * K
        with np.errstate(invalid="ignore", divide="ignore"):
            rel = np.abs((obs + 1.0) * K - x) / np.maximum(np.abs(x), 1e-300)
        live = np.isfinite(x) & (x > 1e-3)          # (obs = x / K - 1 cannot resolve a population near extinction)
        worst = float(rel[live].max()) if live.any() else 0.0       # (v8_myers_r_below_minus_one: every stock NaN, on both sides)
        print(json.dumps(dict(tag=tag, kind="golden", case=c.name, id=c.id, steps=int(n), live_steps=int(live.sum()),
                              nan_pattern_equal=bool(np.array_equal(np.isnan(obs), np.isnan(ref))),
                              max_rel_population=worst, in_units_of_the_bar=worst / BAR,
                              reward_bit_equal=bool(np.array_equal(rew, c.reward.reshape(-1), equal_nan=True)),
                              done_mismatches=int((done != c.done.reshape(-1)).sum()))), flush=True)


def mp_exact(kind, x, z, P):
    import mpmath as mp
    mp.mp.dps = 50
    r, K, s = (mp.mpf(float(P.get(k, 0.0))) for k in ("r", "K", "sigma"))
    C, M, th, q, b, a = (mp.mpf(float(P.get(k, 0.0))) for k in ("C", "M", "theta", "q", "b", "a"))
    out = []
    for xi, zi in zip(x, z):
        xi, zi = mp.mpf(float(xi)), mp.mpf(float(zi))
        if kind == 0:
            v = xi * mp.e ** (r * (1 - xi / K) * (1 - C) / K + s * zi)
        elif kind == 1:
            v = (r + 1) * xi / (1 + xi / (K / r)) * mp.e ** (s * zi)
        elif kind == 2:
            v = (r + 1) * xi ** th / (1 + xi ** th / M) * mp.e ** (s * zi)
        elif kind == 3:
            v = (xi + xi * r * (1 - xi / M) - a * xi ** q / (xi ** q + b ** q)) * mp.e ** (s * zi)
        else:
            v = xi * mp.e ** (r * (1 - xi / K) + s * zi)
        out.append(v)
    return out


def sweep(tag, n=1 << 20, n_exact=20000):
    import mpmath as mp
    rng = np.random.default_rng(2025)
    lib = _capi.lib()
    for env_id in ("fishing-v5", "fishing-v6", "fishing-v7", "fishing-v8", "fishing-v9"):
        model = fo.MODEL_OF_ID[env_id]
        kind = fo.KIND_OF_MODEL[model]
        P = dict(ZOO_DEFAULTS[env_id], sigma=0.1)
        K = float(P["K"])
        x = rng.uniform(0.0, 2.0, n) * K
        x[:64] = 0.0
        z = rng.standard_normal(n)
        p = hh.params(model, r=float(P.get("r", 0.3)), K=K, sigma=0.1, C=float(P.get("C", 0.5)), M=float(P.get("M", 0.0)),
                      theta=float(P.get("theta", 0.0)), q=float(P.get("q", 0.0)), b=float(P.get("b", 0.0)), a=float(P.get("a", 0.0)))
        xt, zt = hh.dev(x), hh.dev(z)
        out = torch.empty_like(xt)
        rc = lib.fishing_population_draw_f64(p, n, xt.data_ptr(), zt.data_ptr(), None, None, None, out.data_ptr(), None)
        assert rc == 0, rc
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        with np.errstate(all="ignore"):
            want = fo.zoo_population_draw(kind, x, z, P)
        ok = np.isfinite(want) & (want > 1e-3)
        rel = np.abs(got - want)[ok] / want[ok]
        i = int(np.flatnonzero(ok)[np.argmax(rel)])
        print(json.dumps(dict(tag=tag, kind="sweep", id=env_id, growth_function=KIND_NAMES[kind], samples=int(ok.sum()),
                              max_rel=float(rel.max()), in_units_of_the_bar=float(rel.max() / BAR), p999_rel=float(np.quantile(rel, 0.999)),
                              worst=dict(x=float(x[i]), z=float(z[i]), got=float(got[i]), want=float(want[i])),
                              nonfinite_and_zero_agree=bool((np.isnan(got) == np.isnan(want)).all() and ((got == 0) == (want == 0)).all()))), flush=True)
        # against the exact value (May is kept away from the zero of its exp_mu, where every float64 evaluation loses digits alike)
        idx = np.flatnonzero(ok)[:n_exact]
        exact = mp_exact(kind, x[idx], z[idx], P)
        ulp = np.spacing(want[idx])
        dev_err = np.array([float(abs(mp.mpf(float(g)) - e)) for g, e in zip(got[idx], exact)]) / ulp
        ref_err = np.array([float(abs(mp.mpf(float(w)) - e)) for w, e in zip(want[idx], exact)]) / ulp
        print(json.dumps(dict(tag=tag, kind="exact", id=env_id, growth_function=KIND_NAMES[kind], samples=int(idx.size),
                              device_max_ulp=float(dev_err.max()), device_mean_ulp=float(dev_err.mean()),
                              reference_arithmetic_max_ulp=float(ref_err.max()), reference_arithmetic_mean_ulp=float(ref_err.mean()))), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default=os.path.basename(os.environ.get("FISHING_HIP_LIB", "default")))
    a = ap.parse_args()
    golden(a.tag)
    sweep(a.tag)
