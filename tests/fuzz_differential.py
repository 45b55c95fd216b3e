#!/usr/bin/env python3
"""(Test infrastructure -- it drives test bodies and, through them, the oracle: it lives in tests/.)
One-off widening of the randomised differential tests (tests/test_gpu_fused_and_dispatch.py): the same test bodies over trial
numbers far beyond the parametrised ranges -- every trial draws its own request from its number --

    python tests/fuzz_differential.py [--requests 1500] [--big 40] [--sequences 12] > gpurun_out/.../fuzz.jsonl

* requests: test_randomised_requests_agree_across_dispatch_general_and_fused (dispatch vs general kernel vs ONE fused launch),
* big:      test_randomised_requests_beyond_4096_tiles (N in (2^22, 1.5 * 2^23]: the zig-zag walk, ranges),
* sequences: test_random_operation_sequences_every_family_three_ways (50 random operations per env id and trial, three ways).
One JSON line per family: trials run, failures (trial number + the assertion's first line), seconds."""
import argparse
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--requests", type=int, default=1500)
    ap.add_argument("--big", type=int, default=40)
    ap.add_argument("--sequences", type=int, default=12)
    ap.add_argument("--only-v4", action="store_true")
    ap.add_argument("--oracle-sweeps", type=int, default=0, help="further seeds of tests/test_gpu_parity.py::test_random_parameter_sets_match_oracle")
    ap.add_argument("--only-oracle", action="store_true")
    ap.add_argument("--only-zoo", action="store_true", help="with --oracle-sweeps: the zoo's sweep alone")
    a = ap.parse_args()
    import hip_harness as hh
    import test_gpu_fused_and_dispatch as T

    def run(name, fn, trials, **kw):
        t0, bad = time.time(), []
        for k in trials:
            try:
                fn(hh, trial=k, **kw)
            except Exception as e:  # noqa: BLE001  (an assertion = a finding; anything else too)
                bad.append({"trial": k, "error": (str(e) or traceback.format_exc()).splitlines()[0][:300], **kw})
        return {"family": name, "trials": len(list(trials)), "first": trials[0], "last": trials[-1], "failures": bad,
                "seconds": round(time.time() - t0, 1), **kw}

    # (the committed parametrisations end at 48 / 10 / 5 / 12: everything from there on is new ground)
    if not a.only_v4 and not a.only_oracle:
      print(json.dumps(run("requests", T.test_randomised_requests_agree_across_dispatch_general_and_fused, range(48, 48 + a.requests))), flush=True)
      print(json.dumps(run("beyond_4096_tiles", T.test_randomised_requests_beyond_4096_tiles, range(10, 10 + a.big))), flush=True)
    if a.oracle_sweeps:
        import numpy as np
        import test_gpu_parity as P
        from oracle import fishing_oracle as fo
        for model in (() if a.only_zoo else (fo.MODEL_V0, fo.MODEL_V1, fo.MODEL_V2, fo.MODEL_V4)):
            for dtype in (np.float64, np.float32):
                t0, bad = time.time(), []
                for k in range(1, 1 + a.oracle_sweeps):
                    try:
                        P.test_random_parameter_sets_match_oracle(hh, model, dtype, seed_offset=k)
                    except Exception as e:  # noqa: BLE001
                        bad.append({"seed_offset": k, "error": (str(e) or traceback.format_exc()).splitlines()[0][:300]})
                print(json.dumps({"family": "oracle parameter sweep (30 parameter sets x 6 steps x 1003 envs per trial)", "model": int(model),
                                  "dtype": np.dtype(dtype).name, "trials": a.oracle_sweeps, "failures": bad,
                                  "seconds": round(time.time() - t0, 1)}), flush=True)
        if not a.only_zoo:
            for policy in ("random", "constant", "escapement", "msy"):
                t0, bad, cnt = time.time(), [], 0
                for k in range(1, 1 + max(1, a.oracle_sweeps // 4)):
                    for model in (fo.MODEL_V0, fo.MODEL_V1, fo.MODEL_V2, fo.MODEL_V4):
                        for dtype in (np.float64, np.float32):
                            cnt += 1
                            try:
                                P.test_fused_rollout_equals_stepwise(hh, model, dtype, policy, seed_offset=k)
                            except Exception as e:  # noqa: BLE001
                                bad.append({"seed_offset": k, "model": int(model), "dtype": np.dtype(dtype).name,
                                            "error": (str(e) or traceback.format_exc()).splitlines()[0][:300]})
                print(json.dumps({"family": "fused rollout == stepwise under the oracle's policy actions (random K, r, sigma, policy parameter, seed, batch)",
                                  "policy": policy, "trials": cnt, "failures": bad[:20], "n_failures": len(bad), "seconds": round(time.time() - t0, 1)}), flush=True)
        import test_gpu_zoo as Z
        for model in (fo.MODEL_V5, fo.MODEL_V6, fo.MODEL_V7, fo.MODEL_V8, fo.MODEL_V9):
            for dtype in (np.float64, np.float32):
                t0, bad = time.time(), []
                for k in range(1, 1 + a.oracle_sweeps):
                    try:
                        Z.test_zoo_random_parameter_sets_match_oracle(hh, model, dtype, seed_offset=k)
                    except Exception as e:  # noqa: BLE001
                        bad.append({"seed_offset": k, "error": (str(e) or traceback.format_exc()).splitlines()[0][:400]})
                print(json.dumps({"family": "zoo oracle parameter sweep (24 parameter sets x 4 steps x 1027 envs per trial)", "model": int(model),
                                  "dtype": np.dtype(dtype).name, "trials": a.oracle_sweeps, "failures": bad[:20], "n_failures": len(bad),
                                  "seconds": round(time.time() - t0, 1)}), flush=True)
        if a.only_oracle:
            return
    for env_id in ("fishing-v0", "fishing-v1", "fishing-v2", "fishing-v4", "fishing-v5", "fishing-v7", "fishing-v8", "fishing-v10", "fishing-v11"):
        fn = T.test_random_operation_sequences_every_family_three_ways
        if a.only_v4 and env_id != "fishing-v4":
            continue
        if env_id == "fishing-v4":      # (fishing-v4's walk is its own test: derived parameters vs stored arrays vs graph replay)
            import test_gpu_v4_params as V
            print(json.dumps(run("sequences fishing-v4", V.test_v4_random_operation_sequences_derived_equals_stored,
                                 range(12, 12 + a.sequences))), flush=True)
            continue
        print(json.dumps(run("sequences", fn, range(5, 5 + a.sequences), env_id=env_id)), flush=True)


if __name__ == "__main__":
    main()
