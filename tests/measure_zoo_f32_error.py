#!/usr/bin/env python3
"""How far the float32 layout of the growth-model zoo (fishing-v5..v11) lands from the reference's float64 arithmetic.

Run ON THE GPU BOX, once per library variant (FISHING_HIP_LIB selects it; `--tag` names it in the record):

    python3 tests/measure_zoo_f32_error.py --tag mid_f64 >> gpurun_out/zoo_f32_error.jsonl

Two measurements per growth function, one JSON line each:
  * "golden": every recorded step of the reference-held fixtures (tests/golden/reference_zoo_trajectories.npz: obs_in, t,
    action, z of the unmodified reference, cast to float32) through fishing_step_f32 -> max |obs - ref|, max |reward - ref|,
    max relative population error, done / t mismatches.  The north star's bar: obs and reward within 1e-6.
  * "sweep": 2^20 random (x, z) per growth function through fishing_population_draw_f32 against the float64 oracle
    (oracle/fishing_oracle.py: zoo_population_draw) on the same float32-representable inputs -> max |x' - ref| / K.
plus "time": microseconds per step of the float32 lean step kernel at N = 2^22, sigma = 0.1, with auto-reset.
"""
import argparse
import json
import os
import statistics
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


def golden(tag):
    for c in load_zoo_cases():
        model = fo.MODEL_OF_ID[c.id]
        K = float(zoo_kw(c)["K"])
        n = c.obs.size
        st = hh.State(n, np.float32, model, c.obs_in.reshape(-1), t=t_in_of(c).reshape(-1),
                      r=c.params_r.reshape(-1) if model == fo.MODEL_V10 else None,
                      model_idx=c.model_idx.reshape(-1) if model == fo.MODEL_V11 else None)
        obs, rew, done, t = st.step(hip_params(hh, c), c.action.reshape(-1), z=c.z.reshape(-1))
        ref = c.obs.reshape(-1)
        dobs = np.abs(obs.astype(np.float64) - ref)
        x = (ref + 1.0) * K
        rel = np.abs((obs.astype(np.float64) + 1.0) * K - x) / np.maximum(x, 1e-300)
        rec = dict(tag=tag, kind="golden", case=c.name, id=c.id, steps=int(n), K=K,
                   max_abs_obs=float(np.nanmax(dobs)), max_abs_reward=float(np.abs(rew - c.reward.reshape(-1)).max()),
                   max_rel_population=float(np.nanmax(np.where(x > 1e-3, rel, 0.0))),
                   done_mismatches=int((done != c.done.reshape(-1)).sum()), t_mismatches=int((t != c.t.reshape(-1)).sum()))
        if model == fo.MODEL_V11:
            per = {}
            for k in range(5):
                m = c.model_idx.reshape(-1) == k
                if m.any():
                    per[KIND_NAMES[k]] = float(np.nanmax(dobs[m]))
            rec["max_abs_obs_by_growth_function"] = per
        print(json.dumps(rec), flush=True)


def sweep(tag, n=1 << 20):
    rng = np.random.default_rng(2024)
    lib = _capi.lib()
    for env_id in ("fishing-v5", "fishing-v6", "fishing-v7", "fishing-v8", "fishing-v9"):
        model = fo.MODEL_OF_ID[env_id]
        P = dict(ZOO_DEFAULTS[env_id], sigma=0.1)
        K = float(P["K"])
        # the stock after harvest: anywhere between extinct and twice the carrying capacity
        x = (rng.uniform(0.0, 2.0, n) * K).astype(np.float32)
        x[:64] = np.float32(0.0)
        z = rng.standard_normal(n).astype(np.float32)
        p = hh.params(model, r=float(P.get("r", 0.3)), K=K, sigma=0.1, C=float(P.get("C", 0.5)), M=float(P.get("M", 0.0)),
                      theta=float(P.get("theta", 0.0)), q=float(P.get("q", 0.0)), b=float(P.get("b", 0.0)), a=float(P.get("a", 0.0)))
        xt, zt = hh.dev(x), hh.dev(z)
        out = torch.empty_like(xt)
        rc = lib.fishing_population_draw_f32(p, n, xt.data_ptr(), zt.data_ptr(), None, None, None, out.data_ptr(), None)
        assert rc == 0, rc
        torch.cuda.synchronize()
        got = out.cpu().numpy().astype(np.float64)
        want = fo.zoo_population_draw(fo.KIND_OF_MODEL[model], x.astype(np.float64), z.astype(np.float64), P)
        ok = np.isfinite(want)
        d = np.abs(got - want)[ok] / K            # in units of the observation (obs = x / K - 1)
        rel = (np.abs(got - want) / np.maximum(want, 1e-300))[ok & (want > 1e-3)]
        i = int(np.argmax(np.abs(got - want) * ok))
        print(json.dumps(dict(tag=tag, kind="sweep", id=env_id, growth_function=KIND_NAMES[fo.KIND_OF_MODEL[model]], samples=int(ok.sum()),
                              max_abs_obs=float(d.max()), p999_abs_obs=float(np.quantile(d, 0.999)), max_rel_population=float(rel.max()),
                              worst=dict(x=float(x[i]), z=float(z[i]), got=float(got[i]), want=float(want[i])),
                              nonfinite_agree=bool((np.isnan(got) == np.isnan(want)).all()))), flush=True)


def timing(tag, n=1 << 22):
    import gym_fishing_amd as gf
    ring = torch.empty((8, n + 3072), device="cuda")
    acts = ring[:, :n]
    acts.copy_(torch.rand((8, n), device="cuda") * 2 - 1)
    runs = [(idn, torch.float32) for idn in ("fishing-v1", "fishing-v5", "fishing-v6", "fishing-v7", "fishing-v8", "fishing-v9",
                                             "fishing-v10", "fishing-v11")]
    runs += [(idn, torch.float64) for idn in ("fishing-v1", "fishing-v8", "fishing-v9", "fishing-v11")]
    for idn, dtype in runs:
        kw = {} if idn == "fishing-v11" else dict(sigma=0.1)
        f32 = dtype == torch.float32
        for returns in (False, True):
            env = gf.make(idn, num_envs=n, seed=1, track_returns=returns, dtype=dtype, **kw)
            if idn == "fishing-v11":
                for d in env.model_params.values():
                    d["sigma"] = 0.1
            env.reset()
            env.step_many(acts, 100)
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                env.step_many(acts, 200)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 5)
            w = 4 if f32 else 8
            byt = (13 + 3 * w) + (2 * w if returns else 0) + (2 * w if idn == "fishing-v10" else 0) + (8 if idn == "fishing-v11" else 0)
            us = statistics.median(ts)
            print(json.dumps(dict(tag=tag, kind="time", id=idn, dtype="float32" if f32 else "float64", n=n, returns=returns,
                                  kernel=env.step_kernel_name(), us_per_step=round(us, 2), bytes_per_env_step=byt,
                                  frac_of_8TBps=round(n * byt / us / 8e6, 3))), flush=True)
            del env


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default=os.path.basename(os.environ.get("FISHING_HIP_LIB", "default")))
    ap.add_argument("--no-time", action="store_true")
    a = ap.parse_args()
    golden(a.tag)
    sweep(a.tag)
    if not a.no_time:
        timing(a.tag)
