#!/usr/bin/env python3
"""The launch-bound sizes (N = 2^19 .. 2^21) split over S concurrent streams: S envs of N / S envs (env_offset keeps the
global env index, so the union is the same batch), every one stepping K times on its own stream.  Does overlapping the
sub-batches' launches hide the dependent-launch floor that bounds the one-stream step there?

    python scripts/exp/time_small_n_streams.py > profiles/r04_small_n_streams.jsonl

The host enqueues in interleaved chunks of `chunk` steps per stream (one fishing_step_many call each), so that no stream
runs dry while another is being fed; `graph`: the same S chains captured into one hipGraph (fork / join by events) and
replayed -- no host enqueue per step at all.
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf  # noqa: E402

K = 2048


def build(env_id, n, S, ret, kw):
    m = n // S
    envs, acts = [], []
    for s in range(S):
        e = gf.make(env_id, num_envs=m, seed=1, env_offset=s * m, track_returns=ret, **kw)
        e.reset()
        ring = torch.empty((8, m + 3072), device="cuda")
        a = ring[:, :m]
        a.copy_(torch.rand((8, m), device="cuda") * 0.2 - 1.0)
        envs.append(e)
        acts.append(a)
    return envs, acts


def time_streams(envs, acts, streams, chunk):
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _c in range(K // chunk):
            for e, a, st in zip(envs, acts, streams):
                with torch.cuda.stream(st):
                    e.step_many(a, chunk)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K * 1e6)
    return best


def time_graph(envs, acts, streams, steps_in_graph=64):
    """One hipGraph: fork from the capturing stream to S streams, `steps_in_graph` dependent steps on each, join."""
    for e in envs:
        e.enable_graph_replay()
    main = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        for e, a in zip(envs, acts):        # warm-up outside the capture
            e.step_many(a, 8)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=main):
        for e, a, st in zip(envs, acts, streams):
            st.wait_stream(main)                      # fork
            with torch.cuda.stream(st):
                e.step_many(a, steps_in_graph)
        for st in streams:
            main.wait_stream(st)                      # join
    best = 1e9
    reps = K // steps_in_graph
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(main):
            for _r in range(reps):
                g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (reps * steps_in_graph) * 1e6)
    return best


def main():
    for env_id, kw, ret, bytes_ in (("fishing-v1", dict(sigma=0.1), True, 33), ("fishing-v2", dict(sigma=0.1), True, 33),
                                    ("fishing-v1", dict(sigma=0.1), False, 25)):
        for log2n in (19, 20, 21):
            n = 1 << log2n
            for S in (1, 2, 4, 8):
                streams = [torch.cuda.Stream() for _ in range(S)]
                envs, acts = build(env_id, n, S, ret, kw)
                line = dict(env_id=env_id, returns=ret, log2_n=log2n, streams=S)
                for chunk in (4, 32):
                    us = time_streams(envs, acts, streams, chunk)
                    line["us_per_step_chunk%d" % chunk] = round(us, 3)
                try:
                    us = time_graph(envs, acts, streams)
                    line["us_per_step_graph"] = round(us, 3)
                except Exception as ex:  # noqa: BLE001
                    line["graph_error"] = repr(ex)[:200]
                best = min(v for k, v in line.items() if k.startswith("us_per_step"))
                line["best_TBps"] = round(n * bytes_ / best / 1e6, 3)
                print(json.dumps(line), flush=True)
                del envs, acts, streams


if __name__ == "__main__":
    main()
