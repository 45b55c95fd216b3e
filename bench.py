#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the vectorised step() at N = 2^22 envs, fishing-v1.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path over one batch: one fishing_step_f32 launch that
advances every env of this rank's shard by one timestep (actions are read from HBM:
a ring of pre-generated random-policy action batches, resident before the clock starts).
Workload (BASELINE.json metric / config 2 at the metric's N): fishing-v1, sigma = 0.1,
r = 0.3, K = 1, x0 = 0.75, Tmax = 100, N = 2^22 envs per GPU, U[-1,1) float32 actions,
in-kernel Philox noise, fused auto-reset, per-env episodic-return accumulation.

Multi-GPU (--gpus N, launched by torch.distributed.run, one rank per GPU): every rank
owns its own 2^22 envs (weak scaling; global env index = rank * 2^22 + i keys the noise),
no data-path collective; one RCCL all-reduce of the 4-double episodic-return record at the
end of the rollout, inside the timed region.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_ENVS = 1 << 22
RING = 8
# algorithmic bytes per env-step: fp32 layout 25 B (R obs 4 + action 4 + t 4,
# W obs 4 + reward 4 + done 1 + t 4) + 8 B for the per-env episodic-return accumulator (R+W 4)
BYTES_STEP = 25
BYTES_STEP_COMPACT = 19     # --compact: years_passed as uint8 (R 1 + W 1 instead of R 4 + W 4)
BYTES_RETURN_ACC = 8
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md chip table)
HBM_COPY_GBS = 6290.0       # measured float4 copy on the same table


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5050)
    ap.add_argument("--warmup", type=int, default=505)
    ap.add_argument("--n-envs", type=int, default=N_ENVS, help="envs per GPU (default 2^22, the metric's N)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-returns", action="store_true", help="pure 25 B step (no episodic-return accumulator)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--compact", action="store_true",
                    help="opt-in compact layout (uint8 year counter, 19 B/env-step); NOT the BASELINE layout")
    ap.add_argument("--extra", action="store_true", help="also time the fused rollout and an L3-spilling N")
    return ap.parse_args()


def cpu_baseline(seconds):
    """Reference-equivalent scalar Python port (oracle/scalar_env.py), one core, bounded sample."""
    from oracle.scalar_env import time_random_rollout
    rate, _ = time_random_rollout("fishing-v1", 20_000, seed=0, sigma=0.1)      # calibrate
    n = int(max(50_000, min(rate * seconds, 5_000_000)))
    rate, _ = time_random_rollout("fishing-v1", n, seed=1, sigma=0.1)
    out = {"value": rate, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": "oracle/scalar_env.py (per-env NumPy step(), same op sequence as the reference): "
                     "%d env-steps of fishing-v1 sigma=0.1, random policy, reset on done, 1 core" % n}
    try:    # the same Python port on every core of the box's CPU share;
        # independent `python -c` workers: nothing here depends on how this file was started
        import subprocess
        procs = max(1, min(os.cpu_count() or 1, 16))
        per = max(20_000, n // 8)
        code = ("import sys; sys.path.insert(0, %r); from oracle.scalar_env import time_random_rollout; "
                "print(time_random_rollout('fishing-v1', %d, seed=int(sys.argv[1]), sigma=0.1)[0])" % (ROOT, per))
        kids = [subprocess.Popen([sys.executable, "-c", code, str(100 + i)], stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, text=True) for i in range(procs)]
        rates = [float(k.communicate(timeout=180)[0].strip().splitlines()[-1]) for k in kids]
        if all(k.returncode == 0 for k in kids):
            out["python_port_all_cores"] = {"value": sum(rates), "unit": "env-steps/s", "cores": procs,
                                            "sample": "%d concurrent processes x %d env-steps each; sum of the "
                                                      "per-process rates" % (procs, per)}
    except Exception as e:  # noqa: BLE001
        out["python_port_all_cores_error"] = repr(e)[:200]
    try:    # stronger CPU figure for context: the plain-C oracle over all host cores
        from oracle import c_oracle
        threads = min(os.cpu_count() or 1, 64)
        c_oracle.rollout_random_f32(1, 1 << 16, 8, threads=threads)             # warm the thread pool
        nn, T = 1 << 20, 32
        t0 = time.perf_counter()
        c_oracle.rollout_random_f32(1, nn, T, threads=threads)
        dt = time.perf_counter() - t0
        out["c_port_all_cores"] = {"value": nn * T / dt, "unit": "env-steps/s", "cores": threads,
                                   "sample": "oracle/fishing_oracle.c float32 + OpenMP, %d envs x %d steps" % (nn, T)}
        t0 = time.perf_counter()
        c_oracle.rollout_random_f32(1, nn // 8, T, threads=1)
        dt = time.perf_counter() - t0
        out["c_port_1_core"] = {"value": nn // 8 * T / dt, "unit": "env-steps/s", "cores": 1}
    except Exception as e:  # noqa: BLE001 - the C port is optional context
        out["c_port_error"] = repr(e)[:200]
    return out


def pmc_traffic(n_envs, with_returns):
    """HBM bytes per launch from a committed rocprofv3 --pmc summary of this same command
    (profiles/pmc_latest.json), or None.  bench.py cannot profile itself."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        if rec.get("n_envs") == n_envs and bool(rec.get("with_returns")) == bool(with_returns):
            return rec.get("hbm_bytes_per_launch"), rec.get("source")
    except Exception:  # noqa: BLE001
        pass
    return None, None


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    # rehearsal knobs (not used by the driver): several ranks on ONE device over gloo, to run the
    # multi-rank control flow on a single-GPU box
    backend = os.environ.get("FISHING_BENCH_BACKEND", "nccl")
    if os.environ.get("FISHING_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK set) the process group is created for any world size, so
    # the single-rank launch exercises the same RCCL init / all-reduce / barrier code as N > 1
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on stdout at communicator creation; stdout must carry
        # exactly one JSON line, so fd 1 points at stderr until the communicator exists.
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
            warm = torch.zeros(4, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    import gym_fishing_amd as gf
    n = args.n_envs
    with_returns = not args.no_returns
    env = gf.make("fishing-v1", sigma=0.1, num_envs=n, env_offset=rank * n, seed=1234,
                  track_returns=with_returns, auto_reset=True, compact=args.compact)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(4321 + rank)
    # ring rows are 12 KiB longer than N so consecutive batches do not start at power-of-two-spaced
    # addresses (same reason the env staggers its own streams)
    ring = torch.empty((RING, n + 3072), device="cuda", dtype=torch.float32)
    actions = ring[:, :n]
    actions.copy_(torch.rand((RING, n), device="cuda", generator=g, dtype=torch.float32) * 2 - 1)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    env.step_many(actions, args.warmup)
    if with_returns:
        env.episode_stats()           # warm the reduce kernel and the RCCL communicator
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sync_all()
    t0 = time.perf_counter()
    ev0.record()
    env.step_many(actions, args.steps)            # K launches on torch's current stream
    ev1.record()
    stats = env.episode_stats() if with_returns else {}
    sync_all()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / max(args.steps, 1)     # HIP events around the K launches
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    total_env_steps = float(n) * world * args.steps
    bytes_per = (BYTES_STEP_COMPACT if args.compact else BYTES_STEP) + (BYTES_RETURN_ACC if with_returns else 0)
    achieved = n * bytes_per / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(n, with_returns)
    resident = n * (4 + (1 if args.compact else 4) + 4 + 1 + (4 if with_returns else 0)) + RING * (n + 3072) * 4
    out = {
        "metric": "env-steps/sec at N=2^22, fishing-v1",
        "value": total_env_steps / elapsed,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "fishing-v1 sigma=0.1 r=0.3 K=1 x0=0.75 Tmax=100, N=2^%d envs per GPU, random-policy "
                               "float32 actions read from HBM (ring of %d batches), in-kernel Philox4x32-10 noise, "
                               "fused auto-reset%s%s; one fishing_step_f32 launch per step" % (
                                   n.bit_length() - 1, RING,
                                   ", per-env episodic-return accumulator + return record" if with_returns else "",
                                   "; COMPACT layout (uint8 year counter)" if args.compact else ""),
                   "envs_per_gpu": n, "global_envs": n * world, "parallelism": "env-shard x%d" % world,
                   "collective": "1 all-reduce of 4 doubles per rollout" if world > 1 else "none"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     # <T, MODEL, NOISE, RET, SIGARR, T8, DRIFT, TERM, BITS> as rocprofv3 prints it
                     "kernel": "fishing::step_kernel_lean<float, 1, 2, %s, false, %s, false, false, false>" % (
                         "true" if with_returns else "false", "true" if args.compact else "false"),
                     "bytes_per_env_step": bytes_per,
                     "avg_launch_us": kernel_ms * 1e3, "frac_of_measured_copy": achieved / HBM_COPY_GBS,
                     "note": "resident arrays %.0f MB (obs, t, reward, done%s + %d action batches): %s the 256 MiB "
                             "Infinity Cache" % (resident / 1e6, ", ep_return" if with_returns else "", RING,
                                                 "fits" if resident < 256 * 2 ** 20 else "exceeds")},
    }
    if stats:
        out["episode_stats"] = {k: stats[k] for k in ("n_episodes", "mean_return", "std_return", "mean_length") if k in stats}

    if with_returns and rank == 0 and world == 1 and not args.compact:
        # for reference on the same line: the bare 25-byte step (no return accumulator), i.e. exactly
        # SURVEY 8(d)'s per-unit figure, measured right after the headline region on the same device
        bare = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1234, auto_reset=True)
        bare.reset()
        bare.step_many(actions, min(args.warmup, 200))
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        kb = max(1, min(args.steps, 2020))
        torch.cuda.synchronize()
        tb = time.perf_counter()
        b0.record()
        bare.step_many(actions, kb)
        b1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - tb
        us = b0.elapsed_time(b1) * 1e3 / kb
        out["bare_step"] = {"value": n * kb / wall, "unit": "env-steps/s", "steps": kb, "bytes_per_env_step": BYTES_STEP,
                            "avg_launch_us": us, "achieved_GBps": n * BYTES_STEP / us / 1e3,
                            "frac": n * BYTES_STEP / us / 1e3 / HBM_PEAK_GBS,
                            "note": "same env family without the per-env episodic-return accumulator"}
        del bare

    if args.extra and rank == 0:
        extra = {}
        del env, actions
        torch.cuda.empty_cache()
        # fused T-step rollout, in-kernel policy (VALU-bound, not HBM-bound): env-steps/s only
        for pol, param in (("random", 0.0), ("escapement", 0.5)):
            env2 = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1234, track_returns=True)
            env2.reset()
            env2.rollout(101, policy=pol, param=param)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            env2.rollout(2020, policy=pol, param=param)
            torch.cuda.synchronize()
            extra["fused_rollout_%s_env_steps_per_s" % pol] = n * 2020 / (time.perf_counter() - t1)
            del env2
        # pure 25 B step at sizes that do / do not fit the Infinity Cache (kernel-only, HIP events)
        for ln in (20, 22, 24, 26):
            nn = 1 << ln
            e3 = gf.make("fishing-v1", sigma=0.1, num_envs=nn, seed=1234)
            e3.reset()
            acts = torch.rand((4, nn), device="cuda") * 2 - 1
            k = max(20, min(400, (1 << 31) // nn))
            e3.step_many(acts, k)
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            e3.step_many(acts, k)
            a1.record()
            torch.cuda.synchronize()
            us = a0.elapsed_time(a1) * 1e3 / k
            extra["step_only_2^%d" % ln] = {"us_per_launch": us, "GBps": nn * BYTES_STEP / us / 1e3,
                                            "env_steps_per_s": nn / us * 1e6}
            del e3, acts
            torch.cuda.empty_cache()
        out["extra"] = extra

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
