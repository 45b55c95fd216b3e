#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the vectorised step() at N = 2^22 envs, fishing-v1.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config v1|v0|v2|v4]

A "step" is one pass of the hot path over one batch: one fishing_step_f32 launch that
advances every env of this rank's shard by one timestep (actions are read from HBM:
a ring of pre-generated random-policy action batches, resident before the clock starts).

Workloads (--config; SURVEY.md section 8d's concrete inputs for BASELINE.json's configs):
  v1 (default, the metric's config)  fishing-v1 sigma=0.1, N = 2^22 per GPU, U[-1,1) float32 actions
  v0 (config 3)  fishing-v0 n_actions=100 sigma=0.1, N = 2^22, uniform int32 actions in [0, 100)
  v2 (config 4)  fishing-v2 C=0.5 sigma=0.1, U[-1,-0.8) actions; N = 2^22 on one GPU, 2^19 per GPU otherwise
  v4 (config 5)  fishing-v4 K_mean=1 r_mean=0.3 sigma_p=0.1, sigma ARRAY filled with 0.05, U[-1,1) actions,
                 N = 2^21 per GPU (2^24 over 8); (K, r) derived in-kernel (--v4-stored: r / K arrays in HBM)
all with in-kernel Philox noise, fused auto-reset, per-env episodic-return accumulation.

Multi-GPU (--gpus N): one rank per GPU over RCCL.  Started bare (no WORLD_SIZE in the environment) this script
launches its N ranks itself as CHILD processes -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 ... bench.py <same flags>` -- before anything touches the GPU, relays rank 0's JSON line
and exits with the children's status; started by torch.distributed.run it is one of those ranks.  Every rank owns
its own envs (weak scaling; global env index = rank * n + i keys the noise), no data-path collective; one
all-reduce of the 4-double episodic-return record per rollout, inside the timed region.

Timed region: a device spin-up (the same launches, >= --spinup-ms, reported), W warm-up steps, a short second spin and one
untimed dress rehearsal of the timed sequence come first;
then EXACTLY K steps between barrier + synchronize on both sides: K launches + the record's reduce kernel
(+ the all-reduce) enqueued, one synchronize.  The record is read back to the host after the clock stops.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings).
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

# dmabuf IPC is what RCCL (and any cross-process device-memory sharing) needs on this driver; set before anything can
# initialise HIP, in EVERY process that runs this file -- a rank started directly by torch.distributed.run (as the
# driver's scaling run does) never passes through self_launch()
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RING = 8
# algorithmic bytes per env-step: fp32 layout 25 B (R obs 4 + action 4 + t 4,
# W obs 4 + reward 4 + done 1 + t 4) + 8 B for the per-env episodic-return accumulator (R+W 4)
BYTES_STEP = 25
BYTES_STEP_COMPACT = 19     # --compact: years_passed as uint8 (R 1 + W 1 instead of R 4 + W 4)
BYTES_RETURN_ACC = 8
BYTES_SIGMA_ARRAY = 4       # fishing-v4: per-env sigma (R 4)
BYTES_V4_STAMP = 8          # fishing-v4 --v4-stamped: per-env origin stamps after a masked reset (R 4 + W 4; int32 in every layout)
BYTES_RK_ARRAYS = 16        # fishing-v4 --v4-stored: per-env r, K read every step (R 4 + 4) and -- on this random-policy workload, where
                            # practically every 128-byte line holds a finished env every step (mean episode length 1.47) -- rewritten by the
                            # redraw (W 4 + 4): rocprofv3 PMC traffic was 1.18 x the 45 B that left the writes out (profiles/r02_step_v4s_*)
BYTES_F64 = 12              # --f64, the parity layout: obs R+W and reward W are 8 bytes wide (25 -> 37 B); the return accumulator +16
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md chip table)
HBM_COPY_GBS = 6290.0       # measured float4 copy on the same table
L2_GBS = 34500.0            # aggregate L2 rate, 8 XCDs x 4 MiB (same guide, "L2 (per XCD)")
L2_BYTES = 32 << 20
INFINITY_CACHE_GBS = 8600.0  # same guide, "Indexed rows": 38 MB table served from the Infinity Cache, 8.6 TB/s chip-wide -- a LOWER bound there
MAX_CPU_WORKERS = 64        # cap of the all-cores CPU baseline (a box without a cgroup quota reports 256 CPUs)

CONFIGS = {
    "v1": dict(env_id="fishing-v1", kwargs=dict(sigma=0.1), actions=("uniform", -1.0, 1.0), log2_n=22, log2_n_multi=22,
               baseline_config=2, what="fishing-v1 sigma=0.1 r=0.3 K=1 x0=0.75 Tmax=100"),
    "v0": dict(env_id="fishing-v0", kwargs=dict(sigma=0.1, n_actions=100), actions=("int", 0, 100), log2_n=22, log2_n_multi=22,
               baseline_config=3, what="fishing-v0 n_actions=100 sigma=0.1 (index->quota map in-kernel)"),
    "v2": dict(env_id="fishing-v2", kwargs=dict(sigma=0.1, C=0.5), actions=("uniform", -1.0, -0.8), log2_n=22, log2_n_multi=19,
               baseline_config=4, what="fishing-v2 tipping point C=0.5 sigma=0.1"),
    "v4": dict(env_id="fishing-v4", kwargs=dict(K_mean=1.0, r_mean=0.3, sigma_p=0.1), actions=("uniform", -1.0, 1.0),
               log2_n=21, log2_n_multi=21, baseline_config=5,
               what="fishing-v4 K_mean=1 r_mean=0.3 sigma_p=0.1, per-env sigma array (0.05), (K, r) redrawn per episode"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5050)
    ap.add_argument("--warmup", type=int, default=505)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="v1", help="workload (default v1 = the metric's)")
    ap.add_argument("--n-envs", type=int, default=0, help="envs per GPU (default: the config's N)")
    ap.add_argument("--spinup-ms", type=float, default=150.0,
                    help="device spin-up ahead of --warmup: the same launches for at least this long (clocks, caches, "
                         "code objects); reported, never part of the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-returns", action="store_true", help="pure 25 B step (no episodic-return accumulator)")
    ap.add_argument("--no-subrecords", action="store_true", help="skip the bare-step / HBM-resident / fused sub-records")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--compact", action="store_true",
                    help="opt-in compact layout (uint8 year counter, 19 B/env-step); NOT the BASELINE layout")
    ap.add_argument("--v4-stored", action="store_true", help="config v4 with r / K arrays in HBM instead of derived (K, r)")
    ap.add_argument("--v4-stamped", action="store_true",
                    help="config v4 after a masked reset(): derived (K, r) with per-env origin stamps (45 B/env-step)")
    ap.add_argument("--f64", action="store_true",
                    help="the float64 parity layout (bit-exact against the reference's arithmetic; 37 B/env-step); NOT the headline layout")
    ap.add_argument("--extra", action="store_true", help="also time the fused rollout and an N sweep")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` sub-record (BASELINE configs 2-5 + the HBM-resident sizes under this run's clock)")
    ap.add_argument("--rehearsal", action="store_true",
                    help="tests only: honour FISHING_BENCH_RUNTIME (a stand-in for the device seam, tests/bench_rehearsal.py); "
                         "without this flag the variable is ignored, so the driver's run can never be re-routed by the environment")
    return ap.parse_args()


# --------------------------------------------------------------------------------------- multi-GPU self-launch
def self_launch(args):
    """--gpus N > 1 without a torch.distributed.run parent: start the N ranks as children (nothing in THIS process
    has touched the GPU), relay rank 0's JSON line, return the children's exit status."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)       # stderr passes through
    line = None
    for out in proc.stdout:
        out = out.strip()
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
    return rc


# --------------------------------------------------------------------------------------- CPU baselines
def host_cores():
    """(cores this process can actually run on, what the box reports): os.cpu_count() is the machine's logical CPU count; the
    container may be pinned to fewer (sched affinity) or granted a CPU-time quota worth fewer (cgroup v2 cpu.max, v1 cfs
    quota) -- a one-GPU box of this pool reports 256 CPUs with a 16-core quota, and 256 workers on a 16-core share measure
    their own start-up contention, not the host.  "All host cores" = min of the three."""
    info = {"os_cpu_count": os.cpu_count(),
            "sched_affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "cgroup_cpu_quota_cores": None}
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        info["cgroup_cpu_quota_cores"] = None if q == "max" else float(q) / float(period)
    except Exception:  # noqa: BLE001
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            info["cgroup_cpu_quota_cores"] = None if q <= 0 else q / period
        except Exception:  # noqa: BLE001
            pass
    limits = [v for v in (info["os_cpu_count"], info["sched_affinity_cpus"]) if v]
    if info["cgroup_cpu_quota_cores"]:
        limits.append(max(1, int(info["cgroup_cpu_quota_cores"] + 0.5)))
    return max(1, min(limits) if limits else 1), info


def cpu_baseline(seconds, cfg_name):
    """Reference-equivalent scalar Python port (oracle/scalar_env.py), one core, bounded sample; beside it the same
    port on every core, the NumPy-vectorised (N,) restatement and the plain-C port."""
    cfg = CONFIGS[cfg_name]
    env_id = cfg["env_id"]
    scalar_id = env_id
    skw = dict(sigma=0.1 if cfg_name != "v4" else 0.05)
    from oracle.scalar_env import time_random_rollout
    rate, _ = time_random_rollout(scalar_id, 20_000, seed=0, **skw)      # calibrate
    n = int(max(50_000, min(rate * seconds, 5_000_000)))
    rate, _ = time_random_rollout(scalar_id, n, seed=1, **skw)
    out = {"value": rate, "unit": "env-steps/s", "cores": 1, "kind": "port", "host_cores": host_cores()[1],
           "sample": "oracle/scalar_env.py (the reference's per-env NumPy step(), helper calls inlined): %d env-steps of %s "
                     "sigma=%g, random policy, reset on done, 1 core" % (n, scalar_id, skw["sigma"]),
           "reference_build_container": {"value": 1.1e5, "unit": "env-steps/s", "cores": 1,
                                         "source": "SURVEY.md 6: the unmodified reference timed in the build container"}}
    try:    # the same Python port on every host core (os.cpu_count() workers, the count reported);
        # independent `python -c` workers: nothing here depends on how this file was started.  Every worker imports first
        # and then waits for one common start time, so the workers really run side by side (a box whose cgroup grants fewer
        # cores than os.cpu_count() shows -- 256 vs a 16-core quota on the round-5 boxes -- gets one worker per USABLE core,
        # host_cores(), and says so): `value` is the aggregate, all env-steps / (last finish - common start); the sum of the
        # workers' own rates is kept beside it.  Bounded: ~2 n env-steps in total, whatever the core count.
        procs, cores_info = host_cores()
        procs = min(procs, MAX_CPU_WORKERS)
        per = max(4_000, (2 * n) // procs)
        t_go = time.time() + 1.5 + 0.05 * procs
        code = ("import sys, time; sys.path.insert(0, %r); from oracle.scalar_env import time_random_rollout\n"
                "while time.time() < %r: time.sleep(0.0005)\n"
                "t0 = time.time(); r = time_random_rollout(%r, %d, seed=int(sys.argv[1]), sigma=%r)[0]; t1 = time.time()\n"
                "print(r, t0, t1)" % (ROOT, t_go, scalar_id, per, skw["sigma"]))
        kids = [subprocess.Popen([sys.executable, "-c", code, str(100 + i)], stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, text=True) for i in range(procs)]
        deadline = t_go + max(30.0, 6.0 * seconds)          # ONE deadline for the whole section; stragglers are killed
        rows = []
        try:
            for k in kids:
                rows.append([float(v) for v in k.communicate(timeout=max(0.1, deadline - time.time()))[0].strip().splitlines()[-1].split()])
        finally:
            for k in kids:
                if k.poll() is None:
                    k.kill()
                    k.communicate()
        if len(rows) == procs and all(k.returncode == 0 for k in kids):
            late = sum(1 for r in rows if r[1] > t_go + 0.05)
            wall = max(r[2] for r in rows) - min(r[1] for r in rows)
            # (a worker that finished importing after the common start skews the aggregate: then only the sum of rates stands)
            out["python_port_all_cores"] = dict({"value": procs * per / wall if late == 0 else None, "unit": "env-steps/s", "cores": procs},
                                                sum_of_worker_rates=sum(r[0] for r in rows), workers_started_late=late,
                                                sample="%d processes (one per usable core, capped at %d) x %d env-steps from one "
                                                       "common start; all env-steps / (last finish - first start)" % (procs, MAX_CPU_WORKERS, per))
    except Exception as e:  # noqa: BLE001
        out["python_port_all_cores_error"] = repr(e)[:200]
    try:    # the NumPy-vectorised (N,) restatement, one process
        from oracle.vector_env import time_vectorised_rollout
        kind, lo, hi = cfg["actions"]
        vkw = dict(cfg["kwargs"])
        vkw.setdefault("sigma", 0.05)
        if kind == "uniform":
            vkw.update(action_low=lo, action_high=hi)
        nv, tv = 1 << 16, 40
        time_vectorised_rollout(env_id, nv, 3, seed=0, **vkw)
        vrate, _ = time_vectorised_rollout(env_id, nv, tv, seed=1, **vkw)
        out["numpy_vectorised"] = {"value": vrate, "unit": "env-steps/s", "cores": 1, "kind": "port",
                                   "sample": "oracle/vector_env.py: float64 step() over (N,) arrays, %d envs x %d steps" % (nv, tv)}
    except Exception as e:  # noqa: BLE001
        out["numpy_vectorised_error"] = repr(e)[:200]
    try:    # stronger CPU figure for context: the plain-C oracle over all host cores
        from oracle import c_oracle
        model = {"v0": 0, "v1": 1, "v2": 2, "v4": 1}[cfg_name]
        threads = host_cores()[0]
        c_oracle.rollout_random_f32(model, 1 << 16, 8, threads=threads)             # warm the thread pool
        nn, T = 1 << 20, 32
        t0 = time.perf_counter()
        c_oracle.rollout_random_f32(model, nn, T, threads=threads)
        dt = time.perf_counter() - t0
        out["c_port_all_cores"] = {"value": nn * T / dt, "unit": "env-steps/s", "cores": threads,
                                   "sample": "oracle/fishing_oracle.c float32 + OpenMP, %d envs x %d steps" % (nn, T)}
        t0 = time.perf_counter()
        c_oracle.rollout_random_f32(model, nn // 8, T, threads=1)
        dt = time.perf_counter() - t0
        out["c_port_1_core"] = {"value": nn // 8 * T / dt, "unit": "env-steps/s", "cores": 1}
    except Exception as e:  # noqa: BLE001 - the C port is optional context
        out["c_port_error"] = repr(e)[:200]
    return out


def pmc_traffic(kernel, n_envs, built=None):
    """HBM bytes per launch from a committed rocprofv3 --pmc summary of this same command
    (profiles/pmc_latest.json), or None.  bench.py cannot profile itself, so the figure is a lookup -- valid only while
    the kernel that runs is the kernel that was profiled: the record carries that kernel's compile-time resources
    (VGPRs / SGPRs / LDS / scratch / occupancy, from the build's own table) and is refused when the library built in
    this tree reports different ones, or none."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if built is None:
        from gym_fishing_amd import build
        built = build.kernel_resources(kernel)
    try:
        with open(path) as f:
            rec = json.load(f)
        for r in rec if isinstance(rec, list) else [rec]:
            if r.get("n_envs") == n_envs and r.get("kernel") == kernel:
                if not built or r.get("kernel_resources") != built:
                    return None, "stale: %s was profiled with kernel resources %s, the built library has %s" % (
                        r.get("source", "the committed record").split(" ")[0], r.get("kernel_resources"), built)
                return r.get("hbm_bytes_per_launch"), r.get("source")
    except Exception:  # noqa: BLE001
        pass
    return None, None


# --------------------------------------------------------------------------------------- the device seam
class HipRuntime:
    """Where a rank's envs live and how it waits for them: the HIP device, one per rank.  (FISHING_BENCH_RUNTIME =
    "module:attr" swaps this object for a stand-in -- tests/bench_rehearsal.py puts the CPU oracle behind the same
    interface so that THIS file's multi-rank control flow -- self-launch, process group, ranks_seen, shard offsets, action
    slices, record all-reduce, barriers, the MAX over ranks, the JSON relay -- runs at world = 8 on a box without eight
    GPUs.  A rehearsal line says so in config.rehearsal and carries no roofline; the driver never sets the variable.)"""
    name = None              # a stand-in names itself here
    cuda = True
    backend = "nccl"
    device = "cuda"

    def claim_device(self, torch, rank, local_rank):
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device")
        single_device = os.environ.get("FISHING_BENCH_SINGLE_DEVICE") == "1"
        # (device_count() does not initialise the GPU on this image; no re-exec and no retry from here on, whatever fails)
        if not single_device and torch.cuda.device_count() < local_rank + 1:
            raise SystemExit("bench.py rank %d: LOCAL_RANK=%d but only %d HIP device(s) visible (HIP_VISIBLE_DEVICES=%r, "
                             "ROCR_VISIBLE_DEVICES=%r): one rank per GPU needs --gpus <= the devices of this node" % (
                                 rank, local_rank, torch.cuda.device_count(), os.environ.get("HIP_VISIBLE_DEVICES"),
                                 os.environ.get("ROCR_VISIBLE_DEVICES")))
        # rehearsal knob (not used by the driver): several ranks on ONE device over gloo, to run the
        # multi-rank control flow on a single-GPU box
        if single_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        return local_rank

    def synchronize(self, torch):
        torch.cuda.synchronize()

    def event(self, torch):
        return torch.cuda.Event(enable_timing=True)

    def make_env(self, gf, torch, cfg_name, n, env_offset, with_returns, compact=False, v4_stored=False, f64=False):
        return make_env(gf, torch, cfg_name, n, env_offset, with_returns, compact, v4_stored, f64)


def load_runtime(rehearsal=False):
    """The HIP runtime -- unless --rehearsal was passed AND FISHING_BENCH_RUNTIME names a stand-in (tests only: the variable
    alone does nothing, and a line produced through a stand-in carries config.rehearsal and no roofline)."""
    spec = os.environ.get("FISHING_BENCH_RUNTIME") if rehearsal else None
    if not spec:
        return HipRuntime()
    import importlib
    mod, _, attr = spec.partition(":")
    return getattr(importlib.import_module(mod), attr or "Runtime")()


# --------------------------------------------------------------------------------------- one rank
ACTION_CHUNK = 1 << 16


def make_actions(torch, cfg, n, rows, env_offset=0, pad=3072, seed=4321, device="cuda"):
    """[rows, n] action ring of the envs [env_offset, env_offset + n).  The random policy's actions are a function
    of the GLOBAL env index -- chunk c = envs [c * 2^16, (c + 1) * 2^16) is drawn from torch's generator seeded
    seed + c -- so a sharded run steps exactly the workload of the single-process run over all envs.  Rows are
    `pad` elements longer than n so consecutive batches do not start at power-of-two-spaced addresses (same
    reason the env staggers its own streams)."""
    kind, lo, hi = cfg["actions"]
    dtype = torch.int32 if kind == "int" else torch.float32
    ring = torch.empty((rows, n + pad), device=device, dtype=dtype)
    view = ring[:, :n]
    g = torch.Generator(device=device)
    first = env_offset // ACTION_CHUNK
    last = (env_offset + n - 1) // ACTION_CHUNK
    for c in range(first, last + 1):
        g.manual_seed(seed + c)
        if kind == "int":
            blk = torch.randint(lo, hi, (rows, ACTION_CHUNK), device=device, generator=g, dtype=torch.int32)
        else:
            blk = torch.rand((rows, ACTION_CHUNK), device=device, generator=g, dtype=torch.float32) * (hi - lo) + lo
        a = max(c * ACTION_CHUNK, env_offset)
        b = min((c + 1) * ACTION_CHUNK, env_offset + n)
        view[:, a - env_offset:b - env_offset] = blk[:, a - c * ACTION_CHUNK:b - c * ACTION_CHUNK]
    return view


def make_env(gf, torch, cfg_name, n, env_offset, with_returns, compact=False, v4_stored=False, f64=False):
    cfg = CONFIGS[cfg_name]
    kw = dict(cfg["kwargs"])
    if f64:
        kw["dtype"] = torch.float64
    if cfg_name == "v4":
        kw["sigma"] = torch.full((n,), 0.05, dtype=torch.float64 if f64 else torch.float32, device="cuda")
        kw["derived_params"] = not v4_stored
    return gf.make(cfg["env_id"], num_envs=n, env_offset=env_offset, seed=1234, track_returns=with_returns,
                   auto_reset=True, compact=compact, **kw)


def bytes_per_env_step(cfg_name, with_returns, compact=False, v4_stored=False, f64=False, v4_stamped=False):
    w = 2 if f64 else 1         # width of the real-valued streams relative to float32
    b = (BYTES_STEP_COMPACT if compact else BYTES_STEP) + (BYTES_F64 if f64 else 0)
    if cfg_name == "v4":
        b += w * (BYTES_SIGMA_ARRAY + (BYTES_RK_ARRAYS if v4_stored else 0)) + (BYTES_V4_STAMP if v4_stamped and not v4_stored else 0)
    return b + (w * BYTES_RETURN_ACC if with_returns else 0)


def spin_up(torch, env, actions, min_ms, rt=None):
    """Run the benchmark's own launches until `min_ms` of wall time has passed (device clocks ramp under load;
    the first launches of a cold device run 10-40 % slow).  Returns (ms, launches)."""
    t0 = time.perf_counter()
    launches = 0
    burst = 256 if rt is None or rt.cuda else 1
    while (time.perf_counter() - t0) * 1e3 < min_ms:
        env.step_many(actions, burst)
        (rt.synchronize(torch) if rt is not None else torch.cuda.synchronize())
        launches += burst
    return (time.perf_counter() - t0) * 1e3, launches


def per_launch_us(torch, env, actions, launches):
    """Event-to-event duration of `launches` single steps (one event pair per launch, same stream): median / mean.
    Includes the event packets' own gaps, so it sits a little above the kernel time rocprofv3 reports."""
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
    R = actions.shape[0]
    evs[0].record()
    for i in range(launches):
        env.step_many(actions[i % R:i % R + 1], 1)
        evs[i + 1].record()
    torch.cuda.synchronize()
    d = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(launches)]
    return statistics.median(d), statistics.fmean(d)


def timed_steps(torch, env, actions, steps, spin_ms=60.0, repeats=1, **kw):
    """HIP events around `steps` launches (or one fused launch of `steps` steps), after `spin_ms` of the same work:
    every sub-record follows allocations and host work during which the device idled and its clocks dropped."""
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < spin_ms:
        env.step_many(actions, steps, **kw)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(repeats):
        env.step_many(actions, steps, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (steps * repeats), time.perf_counter() - t0


def graph_region(torch, gf, args, n, actions):
    """The timed region once more as ONE hipGraph: the same K launches + the record's reduce kernel captured once
    (gym_fishing_amd.graphs.GraphedSteps; the step counter lives in device memory, so every replay draws fresh noise)
    and replayed between synchronize() calls.  Beside the plain figure, never instead of it: what a caller gains who
    can pre-record its loop."""
    from gym_fishing_amd.graphs import GraphedSteps
    try:
        env = make_env(gf, torch, args.config, n, 0, True, False, args.v4_stored)
        env.reset()
        env.step_many(actions, max(args.warmup, 8))
        g = GraphedSteps(env, actions, n_steps=args.steps, record=True)
        walls = []
        for _ in range(12):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
        w = statistics.median(walls[2:])
        rec = g.record.tolist()
        return {"steps": args.steps, "ms_per_step": w / max(args.steps, 1) * 1e3, "value": float(n) * args.steps / w,
                "unit": "env-steps/s", "replays": len(walls), "ms_per_step_first_replay": walls[0] / max(args.steps, 1) * 1e3,
                "kernel": env.step_kernel_name(actions[0]), "episodes_in_record": rec[2],
                "note": "median wall time of one replay of a hipGraph holding the K step launches + the return record's "
                        "reduce kernel, synchronize() on both sides"}
    except Exception as e:  # noqa: BLE001 - a sub-record must not take the headline down
        return {"error": repr(e)[:300]}


def roof(achieved_gbps, cache_resident):
    """One way of writing a memory-roofline figure everywhere on the line.  `frac` is achieved / the 8 TB/s HBM spec and is
    only a fraction of THE roof while HBM is the roof; streams that sit in the 256 MiB Infinity Cache (cache_resident) can be
    served faster than HBM could -- then the figure above 1 is written as `hbm_spec_ratio`, `frac` is null, and no `frac*`
    field of this file ever exceeds 1."""
    f = achieved_gbps / HBM_PEAK_GBS
    out = {"achieved_GBps": achieved_gbps, "frac": f if f <= 1.0 else None, "cache_resident": bool(cache_resident)}
    if f > 1.0:
        out["hbm_spec_ratio"] = f
        out["roof_note"] = "above the HBM spec: the streams are served by the L2s / Infinity Cache, HBM is not the roof at this size"
    return out


def resident_bytes(cfg_name, n, with_returns, rows, f64=False, v4_stored=False, compact=False):
    """State streams + the action ring a stepping loop keeps touching (what has to fit the Infinity Cache to be served by it)."""
    esz = 1 if compact else 4
    rsz = 8 if f64 else 4
    return n * (rsz + esz + rsz + 1 + (rsz if with_returns else 0) + (rsz if cfg_name == "v4" else 0)
                + (2 * rsz if v4_stored and cfg_name == "v4" else 0)) + rows * (n + 3072) * 4


def steady_launch_us(torch, env, actions, launches=256, lead=16, spin_ms=60.0):
    """Average duration of one step launch: HIP events around `launches` back-to-back launches enqueued behind a `lead`-launch
    lead-in (the device is busy when the first event fires), after `spin_ms` of the same work -- the headline's
    roofline.avg_launch_us, for any env."""
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < spin_ms:
        env.step_many(actions, 64)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    env.step_many(actions, lead)
    e0.record()
    env.step_many(actions, launches)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / launches


# BASELINE.json's configs 2-5 at their per-GPU-shard and whole sizes, SURVEY.md section 8(d)'s spill sizes of configs 3 / 4, and the
# metric's workload in the REFERENCE's precision (float64: bit-exact against its NumPy arithmetic; 53 B with the return accumulator,
# 37 B bare): (key, config, log2 N, float64?, per-env returns?)
CONFIG_RECORDS = (
    ("config2_v1_2p20", "v1", 20, False, True),
    ("config3_v0_2p22", "v0", 22, False, True),
    ("config4_v2_2p19_shard", "v2", 19, False, True),
    ("config4_v2_2p22", "v2", 22, False, True),
    ("config5_v4_2p21_shard", "v4", 21, False, True),
    ("config5_v4_2p24", "v4", 24, False, True),
    ("config3_v0_2p26", "v0", 26, False, True),
    ("config4_v2_2p26", "v2", 26, False, True),
    ("metric_v1_2p22_f64", "v1", 22, True, True),
    ("metric_v1_2p22_f64_bare", "v1", 22, True, False),
)


def floor_launch_us(torch, env, acts, n, mode, launches=256, lead=16, spin_ms=10.0):
    """steady_launch_us for the library's floor kernels (fishing_step_floor_f32: the step launch's grid, argument list and
    kernarg preload; mode 0 an empty body, mode 1 a copy over the step's streams -- it clobbers reward / done / ep_return, so
    the env is scratch afterwards)."""
    from gym_fishing_amd import _capi
    lib, bufs, stream = env._lib, env._c_buffers(acts[0]), env._stream()
    run = lambda k: _capi.check(lib.fishing_step_floor_f32(mode, n, bufs, k, stream), "fishing_step_floor_f32")  # noqa: E731
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < spin_ms:
        run(64)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    run(lead)
    e0.record()
    run(launches)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / launches


def latency_floor(torch, env, acts, n, step_bytes, resident, us):
    """The roof of a launch-bound size.  latency_floor_us = the empty kernel of the step's grid and argument shape, measured here,
    + the step's algorithmic bytes / the rate of the cache level that can hold them between two steps, by the guide's figures:
    the aggregate L2 (32 MiB, 34.5 TB/s) serves everything when the whole resident set -- state streams + action ring -- fits it,
    the state streams alone (a tile stays on its XCD from step to step) when only they fit, nothing otherwise; the rest comes from
    the Infinity Cache (8.6 TB/s -- a figure the guide gives as a lower bound).  copy_floor_us = a copy over the fishing-v1
    stream set (33 B per env) in the step's access shape, measured here (frac_of_copy scales its memory part to the config's
    bytes): where it beats the computed floor, the guide's Infinity-Cache rate is the pessimistic term."""
    empty = statistics.median(floor_launch_us(torch, env, acts, n, 0) for _ in range(3))
    copy = statistics.median(floor_launch_us(torch, env, acts, n, 1) for _ in range(3))
    ring = acts.shape[0] * (n + 3072) * 4
    if resident <= L2_BYTES:
        level, l2_bytes = "L2", step_bytes
    elif resident - ring <= L2_BYTES:
        level, l2_bytes = "L2 (state) + infinity-cache (actions)", step_bytes - 4
    else:
        level, l2_bytes = "infinity-cache", 0
    floor = empty + n * l2_bytes / L2_GBS / 1e3 + n * (step_bytes - l2_bytes) / INFINITY_CACHE_GBS / 1e3
    out = {"empty_launch_us": round(empty, 3), "copy_floor_us": round(copy, 3), "copy_bytes_per_env_step": 33,
           "latency_floor_us": round(floor, 3), "floor_level": level, "bytes_from_L2": l2_bytes}
    # (a fraction of a floor is at most 1; where the step beats the estimate -- the guide's Infinity-Cache rate is a lower
    # bound -- the ratio is written as such and the fraction is null: no `frac*` field of this file ever exceeds 1)
    # the measured copy moves fishing-v1's 33 B per env; a config that streams more (fishing-v4: 37 B) scales its memory part
    copy_scaled = empty + max(copy - empty, 0.0) * step_bytes / 33.0
    if step_bytes != 33:
        out["copy_floor_scaled_us"] = round(copy_scaled, 3)
    for name, f in (("floor", floor / us), ("copy", copy_scaled / us)):
        out["frac_of_" + name] = round(f, 4) if f <= 1.0 else None
        if f > 1.0:
            out[name + "_over_launch_ratio"] = round(f, 4)
    return out


def config_records(torch, gf, launches=256):
    """Every BASELINE config under THIS run's clock (the driver times the default command only): per record the kernel the
    dispatch picked, its algorithmic bytes per env-step, the average launch duration (steady_launch_us: HIP events over
    >= 256 launches behind a lead-in) and the roofline figure with its regime; at the launch-bound shard sizes (N <= 2^21) also
    the floor such a launch stands on (latency_floor).  Config 2 additionally carries its in-kernel random-policy rollout
    (env-steps/s; not an HBM-bound kernel)."""
    out = {}
    for key, name, ln, f64, with_returns in CONFIG_RECORDS:
        try:
            n = 1 << ln
            rows = RING if ln <= 22 else 4
            cfg = CONFIGS[name]
            env = make_env(gf, torch, name, n, 0, with_returns, f64=f64)
            env.reset()
            acts = make_actions(torch, cfg, n, rows)
            env.step_many(acts, 24)
            # (the launch-bound sizes swing with the device's power state from one millisecond bracket to the next -- 4.3 / 5.4 us at
            # N = 2^19 on one box within one minute --: five brackets, the median reported, every one listed)
            runs = [steady_launch_us(torch, env, acts, launches, spin_ms=60.0 if i == 0 else 10.0) for i in range(5 if ln <= 21 else 1)]
            us = statistics.median(runs)
            b = bytes_per_env_step(name, with_returns, f64=f64)
            resident = resident_bytes(name, n, with_returns, rows, f64=f64)
            rec = {"n_envs": n, "kernel": env.step_kernel_name(acts[0]).replace("fishing::step_kernel_", ""), "bytes_per_env_step": b,
                   "avg_launch_us": round(us, 3), "env_steps_per_s": n / us * 1e6}
            if f64:
                rec["dtype"] = "f64"
            if len(runs) > 1:
                rec["brackets_us"] = [round(r, 2) for r in runs]
            r = roof(n * b / us / 1e3, resident < 256 * 2 ** 20)
            r.pop("roof_note", None)
            rec.update({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
            if key == "config2_v1_2p20":        # "random-policy rollout": the policy drawn in-kernel, no action traffic at all
                env.rollout(101, policy="random")
                torch.cuda.synchronize()
                r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                r0.record()
                for _ in range(8):
                    env.rollout(101, policy="random")
                r1.record()
                torch.cuda.synchronize()
                us_r = r0.elapsed_time(r1) * 1e3 / (8 * 101)
                rec["random_policy_rollout"] = {"us_per_step": round(us_r, 4), "env_steps_per_s": n / us_r * 1e6}
            if ln <= 21 and not f64 and with_returns:
                rec.update(latency_floor(torch, env, acts, n, b, resident, us))
            out[key] = rec
            del env, acts
        except Exception as e:  # noqa: BLE001 - a sub-record must not take the headline down
            out[key] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
    out["note"] = ("HIP events around %d back-to-back step launches behind a 16-launch lead-in, same process as the headline; frac = of "
                   "the 8 TB/s HBM spec (null + hbm_spec_ratio when cache-resident streams beat it); latency_floor_us = empty kernel of "
                   "the same grid (measured) + bytes / the guide's L2 or Infinity-Cache rate; rocprofv3 twins: profiles/r06_step_*" % launches)
    return out


def python_step_loop(torch, gf, calls=2000):
    """What a drop-in caller of the reference's env.step() runs: a Python loop, one env.step(actions[k]) per step (SURVEY 8d:
    "kernel-only and end-to-end (Python loop / hipGraph)").  enqueue_us = host time per call, nothing waited for; wall_us = the
    loop including the final synchronize, per step; step_many_us = the same launches enqueued by one C call."""
    out = {}
    for ln in (20, 22):
        try:
            n = 1 << ln
            env = make_env(gf, torch, "v1", n, 0, True)
            env.reset()
            acts = make_actions(torch, CONFIGS["v1"], n, RING)
            for k in range(200):
                env.step(acts[k % RING])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(calls):
                env.step(acts[k % RING])
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            many = steady_launch_us(torch, env, acts, 256, spin_ms=10.0)
            out["2^%d" % ln] = {"calls": calls, "enqueue_us": round((t1 - t0) / calls * 1e6, 3), "wall_us": round((t2 - t0) / calls * 1e6, 3),
                                "env_steps_per_s": n * calls / (t2 - t0), "step_many_us": round(many, 3)}
            del env, acts
        except Exception as e:  # noqa: BLE001
            out["2^%d" % ln] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
    return out


def rank_report(torch, dist, rt, backend, rank, world, local_device, env_offset, n, elapsed_local, steps, local_rec, merged_rec):
    """What the first unattended multi-GPU run must say about itself: every rank's device, shard and time, and whether the
    all-reduced return record is the sum of the ranks' own records.  One all_gather of 8 doubles per rank, after the clock
    stopped.  Returns (list of per-rank dicts, ok, message); every rank computes the same verdict from the same gathered data."""
    dev = "cuda" if backend == "nccl" else "cpu"
    mine = torch.tensor([float(rank), float(local_device), float(env_offset), float(n), elapsed_local / max(steps, 1) * 1e3]
                        + [float(x) for x in local_rec.tolist()] + [float(x) for x in merged_rec.tolist()],
                        dtype=torch.float64, device=dev)
    rows = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(rows, mine)
    rows = [r.cpu().tolist() for r in rows]
    ranks = [{"rank": int(r[0]), "device": int(r[1]), "env_offset": int(r[2]), "n_envs": int(r[3]), "ms_per_step": r[4],
              "record": r[5:9]} for r in rows]
    ranks.sort(key=lambda d: d["rank"])
    problems = []
    if [d["rank"] for d in ranks] != list(range(world)):
        problems.append("ranks gathered: %s" % [d["rank"] for d in ranks])
    total = [sum(d["record"][k] for d in ranks) for k in range(4)]
    for r in rows:                       # every rank must hold the same merged record
        if r[9:13] != rows[0][9:13]:
            problems.append("rank %d holds a different all-reduced record than rank %d" % (int(r[0]), int(rows[0][0])))
            break
    merged = rows[0][9:13]
    for k, name in enumerate(("sum_return", "sum_sq_return", "n_episodes", "sum_length")):
        exact = k >= 2                   # counts are integers in doubles: exact; the sums may differ in the last bits by order
        if (merged[k] != total[k]) if exact else (abs(merged[k] - total[k]) > 1e-12 * max(1.0, abs(total[k]))):
            problems.append("all-reduced %s = %r but the ranks' own records sum to %r" % (name, merged[k], total[k]))
    offs = sorted((d["env_offset"], d["n_envs"]) for d in ranks)
    if any(offs[i][0] + offs[i][1] != offs[i + 1][0] for i in range(len(offs) - 1)) or offs[0][0] != 0:
        problems.append("the ranks' env blocks do not tile [0, N): %s" % offs)
    return ranks, not problems, "; ".join(problems)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    wall_s = {}                     # where this run's wall time went (seconds per section; reported as `bench_wall_s`)
    t_mark = [time.perf_counter()]

    def mark(name):
        now = time.perf_counter()
        wall_s[name] = round(wall_s.get(name, 0.0) + now - t_mark[0], 3)
        t_mark[0] = now

    import torch
    import torch.distributed as dist

    rt = load_runtime(args.rehearsal)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d under torch.distributed.run needs %d ranks (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
    # a rank that cannot get its device exits non-zero here, with a one-line diagnosis: no re-exec, no retry
    local_rank = rt.claim_device(torch, rank, local_rank)
    backend = os.environ.get("FISHING_BENCH_BACKEND", rt.backend)
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
            # every rank contributes a one: the sum is the number of ranks the collective library actually joined
            ones = torch.ones(1, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(ones)
            rt.synchronize(torch)
            ranks_seen = int(round(float(ones.item())))
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

        if dist.get_world_size() != world or ranks_seen != world:
            print("bench.py rank %d: the process group has %d ranks and the all-reduce of ones saw %d, WORLD_SIZE says %d"
                  % (rank, dist.get_world_size(), ranks_seen, world), file=sys.stderr, flush=True)
            raise SystemExit(3)
    else:
        ranks_seen = None

    import gym_fishing_amd as gf
    cfg = CONFIGS[args.config]
    n = args.n_envs or (1 << (cfg["log2_n"] if world == 1 else cfg["log2_n_multi"]))
    with_returns = not args.no_returns
    env = rt.make_env(gf, torch, args.config, n, rank * n, with_returns, args.compact, args.v4_stored, args.f64)
    env.reset()
    stamped = args.v4_stamped and args.config == "v4" and not args.v4_stored and rt.cuda
    if stamped:         # one env in eight reset on its own: the batch stays in the derived mode, on per-env origin stamps
        env.reset(torch.arange(n, device="cuda") % 8 == 0)
    actions = make_actions(torch, cfg, n, RING, rank * n, device=rt.device)

    mark("import_init_env_actions")
    barrier_first = use_dist and backend == "nccl" and os.environ.get("FISHING_BENCH_BARRIER_FIRST", "1") == "1"

    def sync_all():
        # barrier + synchronize.  RCCL's barrier is a collective on the device, ordered behind everything this rank has
        # enqueued, and torch blocks the host until it has run: enqueued straight behind the launches its launch latency
        # hides under them (one host <-> device round trip per bracket instead of two).  A host-side barrier (gloo
        # rehearsal) says nothing about the device, so there the device is drained first.
        if use_dist and not barrier_first:
            rt.synchronize(torch)
        if use_dist:
            dist.barrier()
        rt.synchronize(torch)

    spin_ms, spin_launches = spin_up(torch, env, actions, args.spinup_ms, rt)
    env.step_many(actions, args.warmup)
    if with_returns:
        env.episode_stats()           # warm the reduce kernel and the RCCL communicator
    ev0, ev1 = rt.event(torch), rt.event(torch)

    def region(with_events=False):
        """The timed sequence, exactly: K launches + the record's reduce kernel (+ the all-reduce) enqueued, then the
        closing barrier + synchronize.  (`with_events`: the untimed rehearsal brackets its K launches with HIP events --
        roofline.avg_launch_us_rehearsal_region; two event packets cost the 20-step region 9 of its 417 us, so the timed
        pass carries none: profiles/r04_region_variants.jsonl.)"""
        if with_events:
            ev0.record()
        env.step_many(actions, args.steps)            # K launches on torch's current stream
        if with_events:
            ev1.record()
        rec = env.episode_record() if with_returns else None     # reduce kernel (+ all-reduce), enqueued only
        sync_all()
        return rec

    # One untimed dress rehearsal of that very sequence (every lazily initialised path on the host -- event timing,
    # the reduce launch behind a burst of step launches, the barrier -- has then run once: a 20-step region is ~0.45 ms, and a
    # first-use stall of a few hundred microseconds inside it was seen to cost a third of the figure), and no garbage
    # collection inside the region.  Both are reported under `spinup`; neither skips or shortens the K timed steps.
    # (the collection comes BEFORE the rehearsal: nothing but the rehearsal's own closing barrier + synchronize may lie between
    # the last busy moment of the device and the clock's start -- a few idle milliseconds let its clocks drop, and the 20
    # launches that follow run 2-3 us slower each)
    import gc
    gc.collect()
    gc.disable()
    # (... and the collection itself is tens of milliseconds during which the device idles and its clocks drop; the rehearsal's
    # 0.4 ms of launches do not bring them back -- a region timed behind it read 410-416 us where the same region in a loop that
    # keeps the device busy reads 397, profiles/r06_region_fixed_cost.json -- so the spin-up's launches run once more, briefly)
    respin_ms, respin_launches = spin_up(torch, env, actions, min(args.spinup_ms, 30.0), rt) if args.spinup_ms > 0 else (0.0, 0)
    sync_all()
    region(with_events=True)
    kernel_ms = ev0.elapsed_time(ev1) / max(args.steps, 1)     # HIP events around the rehearsal's K launches
    t0 = time.perf_counter()
    record = region()
    elapsed = time.perf_counter() - t0
    gc.enable()
    elapsed_local = elapsed
    ranks = None
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        if record is not None:
            # self-diagnosis, after the clock: the merged record must be the sum of the ranks' own records (the env's scratch
            # record is rewritten by episode_record(): keep the merged one first)
            record = record.clone()
            local_rec = env.episode_record(all_reduce=False).clone()
            ranks, ok, why = rank_report(torch, dist, rt, backend, rank, world, local_rank, rank * n, n, elapsed_local, args.steps,
                                         local_rec, record)
            if not ok:
                print("bench.py rank %d: return-record self-check FAILED: %s" % (rank, why), file=sys.stderr, flush=True)
                raise SystemExit(4)
    stats = {}
    if record is not None:
        from gym_fishing_amd.sharding import summarize_record
        stats = summarize_record(record)
    # The kernel's own duration, for the roofline: the same K launches once more, this time enqueued BEHIND a short
    # lead-in so that the device is already busy when the first event fires -- HIP events around the K launches then
    # hold K kernel durations and nothing else.  (The bracket over the timed region above also contains the idle
    # device's pick-up of the first launch, ~10-25 us once: 0.5 us per step at K = 20, invisible at K = 5050.)
    if rt.cuda:
        k_steady = max(args.steps, 256)          # (at the driver's K = 20 the events' own few us would be 1-2 % of the bracket)
        # (the recipe of every `configs` entry: behind 60 ms of the same work -- host work since the timed region has let the device
        # idle, and 256 launches = 5 ms on a device still ramping read 19.2-19.4 us where rocprofv3 and K = 5050 read 18.7-18.8)
        steady_ms = steady_launch_us(torch, env, actions, k_steady, lead=16, spin_ms=60.0) / 1e3
        med_us, mean_us = per_launch_us(torch, env, actions, max(1, min(args.steps, 200)))
    else:               # a rehearsal measures nothing about a kernel
        k_steady, steady_ms, med_us, mean_us = args.steps, kernel_ms, None, None

    mark("spinup_warmup_timed_region_roofline_events")
    total_env_steps = float(n) * world * args.steps
    bytes_per = bytes_per_env_step(args.config, with_returns, args.compact, args.v4_stored, args.f64, stamped)
    achieved = n * bytes_per / (steady_ms * 1e-3) / 1e9
    kernel = env.step_kernel_name(actions[0])
    traffic, traffic_src = pmc_traffic(kernel, n) if rt.cuda else (None, None)
    resident = resident_bytes(args.config, n, with_returns, RING, args.f64, args.v4_stored, args.compact)
    fits = resident < 256 * 2 ** 20
    head = roof(achieved, fits)
    log2n = n.bit_length() - 1 if n & (n - 1) == 0 else None
    out = {
        "metric": "env-steps/sec at N=2^22, fishing-v1" if args.config == "v1" else
                  "env-steps/sec, BASELINE config %d (%s)" % (cfg["baseline_config"], cfg["env_id"]),
        "value": total_env_steps / elapsed,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
        "ms_per_step_min_over_ranks": min(d["ms_per_step"] for d in ranks) if ranks else elapsed_local / max(args.steps, 1) * 1e3,
        "ms_per_step_max_over_ranks": max(d["ms_per_step"] for d in ranks) if ranks else elapsed_local / max(args.steps, 1) * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64" if args.f64 else "f32",
        "data": "synthetic",
        "config": {"workload": "%s, N=%s envs per GPU, random-policy %s actions read from HBM (ring of %d batches), "
                               "in-kernel Philox4x32-10 noise, fused auto-reset%s%s%s; one fishing_step_%s launch per step" % (
                                   cfg["what"], ("2^%d" % log2n) if log2n is not None else str(n),
                                   "int32 [0,100)" if cfg["actions"][0] == "int" else "float32 U[%g,%g)" % cfg["actions"][1:],
                                   RING, ", per-env episodic-return accumulator + return record" if with_returns else "",
                                   ("; COMPACT layout (uint8 year counter)" if args.compact else "") + ("; FLOAT64 parity layout" if args.f64 else ""),
                                   ("; r / K arrays in HBM" if args.v4_stored else "; (K, r) re-derived in-kernel, no r / K arrays"
                                    + (", per-env origin stamps after a masked reset of one env in eight" if stamped else ""))
                                   if args.config == "v4" else "", "f64" if args.f64 else "f32"),
                   "name": args.config,
                   "baseline_config": cfg["baseline_config"] if not (args.config == "v1" and n != 1 << 20)
                   else "metric (config 2's workload at N = %d)" % n,
                   "envs_per_gpu": n, "global_envs": n * world, "parallelism": "env-shard x%d" % world,
                   "collective": "1 all-reduce of 4 doubles per rollout (%s)" % ("RCCL" if backend == "nccl" else backend)
                                 if world > 1 else "none",
                   # ranks counted by an all-reduce of ones at start-up (null: no process group, i.e. a bare 1-GPU run)
                   ("rccl_ranks_seen" if backend == "nccl" else backend + "_ranks_seen"): ranks_seen,
                   # per rank: device, env block, its own ms_per_step and return record (null: no process group)
                   "ranks": ranks,
                   "record_is_sum_of_rank_records": True if ranks is not None else None,
                   "rehearsal": rt.name},
        "spinup": {"ms": spin_ms, "launches": spin_launches, "respin_ms": respin_ms, "respin_launches": respin_launches,
                   "rehearsal_launches": args.steps,
                   "note": "same launches as the timed region, ahead of --warmup; behind the warm-up a second, short spin (the "
                           "host work in between lets the clocks drop) and one untimed dress rehearsal of the timed sequence "
                           "itself (K launches + record + barrier); none of it timed"},
        "roofline": {"bound": "infinity-cache/hbm" if fits else "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": head["frac"], "hbm_spec_ratio": head.get("hbm_spec_ratio"), "traffic": traffic, "traffic_source": traffic_src,
                     # traffic: a LOOKUP of the committed rocprofv3 --pmc record of this command (bench.py cannot read counters
                     # of its own run), quoted only while the built kernel has the profiled kernel's resources
                     "traffic_is_lookup_of_committed_pmc_record": True,
                     "kernel": kernel, "bytes_per_env_step": bytes_per,
                     "avg_launch_us": steady_ms * 1e3, "avg_launch_launches": k_steady,
                     "avg_launch_us_rehearsal_region": kernel_ms * 1e3,
                     "launch_us_median_event_pairs": med_us, "launch_us_mean_event_pairs": mean_us,
                     "measured_copy_ratio": achieved / HBM_COPY_GBS,
                     "cache_resident": fits,
                     # the same kernel where HBM IS the roof (N = 2^26, filled from the hbm_resident sub-record below)
                     "hbm_resident_frac": None,
                     "note": "avg_launch_us = HIP events around max(K, 256) launches enqueued behind a 16-launch lead-in (device "
                             "busy when the first event fires) after 60 ms of the same launches, per launch -- the recipe of every "
                             "`configs` entry; avg_launch_us_rehearsal_region = the same bracket over the K "
                             "launches of the UNTIMED dress rehearsal of the timed region (includes the idle device's pick-up of the "
                             "first launch; the timed pass itself carries no event packets); "
                             "launch_us_median_event_pairs = median event-to-event time of single launches (each pair adds "
                             "its own ~2.5 us of packet gaps).  Resident arrays %.0f MB "
                             "(state streams + %d action batches): %s the 256 MiB Infinity Cache%s" % (
                                 resident / 1e6, RING, "fits" if fits else "exceeds",
                                 " -- the HBM-resident figure is the hbm_resident sub-record" if fits else "")},
    }
    if stats:
        out["episode_stats"] = {k: stats[k] for k in ("n_episodes", "mean_return", "std_return", "mean_length") if k in stats}
    if not rt.cuda:
        out["roofline"] = None
        args.no_subrecords = args.no_cpu_baseline = True
        args.extra = False
    if stamped:
        args.no_subrecords = True
    if rank == 0 and world == 1 and with_returns and not args.no_subrecords and not args.compact and not args.f64:
        out["graph_region"] = graph_region(torch, gf, args, n, actions)
        mark("graph_region")

    subrecords = rank == 0 and world == 1 and not args.no_subrecords and not args.compact and not args.f64
    if subrecords and with_returns:
        # for reference on the same line: the bare step (no return accumulator), i.e. exactly
        # SURVEY 8(d)'s per-unit figure, measured right after the headline region on the same device
        bare = make_env(gf, torch, args.config, n, 0, False, False, args.v4_stored)
        bare.reset()
        bare.step_many(actions, min(max(args.warmup, 50), 200))
        kb = max(256, min(args.steps, 2020))
        us, wall = timed_steps(torch, bare, actions, kb)
        bb = bytes_per_env_step(args.config, False, False, args.v4_stored)
        out["bare_step"] = {"value": n * kb / wall, "unit": "env-steps/s", "steps": kb, "bytes_per_env_step": bb,
                            "avg_launch_us": us, **roof(n * bb / us / 1e3, resident_bytes(args.config, n, False, RING, False,
                                                                                           args.v4_stored) < 256 * 2 ** 20),
                            "kernel": bare.step_kernel_name(actions[0]),
                            "note": "same env family without the per-env episodic-return accumulator"}
        del bare
        mark("bare_step")
    if subrecords:
        # HBM-resident figure: the same workload at a size whose state streams alone are several times the
        # 256 MiB Infinity Cache
        del env
        torch.cuda.empty_cache()
        big = 1 << (26 if args.config != "v4" else 24)       # (config 5's whole N = 2^24 on one GPU: 620 MB of streams)
        rows = 4
        eb = make_env(gf, torch, args.config, big, 0, with_returns, False, args.v4_stored)
        eb.reset()
        ab = make_actions(torch, cfg, big, rows)
        eb.step_many(ab, 24)
        us, _ = timed_steps(torch, eb, ab, 100)
        # the same env once more in a second allocation (made while the first is alive): round 2 saw +-7 % between
        # allocations at this size, round 3 could not reproduce it; reported beside the first
        eb2 = make_env(gf, torch, args.config, big, 0, with_returns, False, args.v4_stored)
        eb2.reset()
        eb2.step_many(ab, 24)
        us2, _ = timed_steps(torch, eb2, ab, 100)
        del eb2
        out["hbm_resident"] = {"n_envs": big, "steps": 100, "bytes_per_env_step": bytes_per, "avg_launch_us": us,
                               "avg_launch_us_second_allocation": us2,
                               **roof(big * bytes_per / us / 1e3, False),
                               "env_steps_per_s": big / us * 1e6, "kernel": eb.step_kernel_name(ab[0]),
                               "resident_MB": (big * (bytes_per - 1 - 4) / 2 + big * 5 + rows * big * 4) / 1e6,
                               "note": "same workload, N = 2^%d: far outside the Infinity Cache" % (big.bit_length() - 1)}
        out["roofline"]["hbm_resident_frac"] = out["hbm_resident"]["frac"]
        out["roofline"]["hbm_resident_n_envs"] = big
        del eb, ab
        torch.cuda.empty_cache()
        if args.config != "v4":
            # ... and at N = 2^24 (SURVEY.md section 8d names both sizes): 554 MB of streams, twice the Infinity Cache
            mid = 1 << 24
            em = make_env(gf, torch, args.config, mid, 0, with_returns, False, args.v4_stored)
            em.reset()
            am = make_actions(torch, cfg, mid, rows)
            em.step_many(am, 24)
            usm, _ = timed_steps(torch, em, am, 200)
            out["hbm_resident"]["n_2p24"] = {"n_envs": mid, "steps": 200, "avg_launch_us": usm,
                                             **roof(mid * bytes_per / usm / 1e3, False),
                                             "kernel": em.step_kernel_name(am[0])}
            del em, am
            torch.cuda.empty_cache()
        mark("hbm_resident")
        # launch-bound sizes: the fused multi-step kernel (K steps per launch, state in registers) next to
        # the launch-per-step path; env-steps/s only -- its HBM traffic is 9 B/env-step, not the headline's
        fused = {}
        for ln in (19, 20):
            nn = 1 << ln
            ef = make_env(gf, torch, args.config, nn, 0, with_returns, False, args.v4_stored)
            ef.reset()
            af = make_actions(torch, cfg, nn, RING)
            rows_r = torch.empty((101, nn), dtype=torch.float32, device="cuda")
            rows_d = torch.empty((101, nn), dtype=torch.uint8, device="cuda")
            ef.step_many(af, 101, fused=True, rewards_out=rows_r, dones_out=rows_d)
            ef.step_many(af, 202)
            us_l, _ = timed_steps(torch, ef, af, 1010)
            us_f, _ = timed_steps(torch, ef, af, 101, repeats=8, fused=True, rewards_out=rows_r, dones_out=rows_d)
            us_f2, _ = timed_steps(torch, ef, af, 101, repeats=8, fused=True)
            # ... and the fused rollout with the policy drawn in-kernel (no action traffic at all): BASELINE config 2 is
            # "N = 2^20, random-policy rollout"
            ef.rollout(101, policy="random")
            torch.cuda.synchronize()
            r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            r0.record()
            for _ in range(8):
                ef.rollout(101, policy="random")
            r1.record()
            torch.cuda.synchronize()
            us_r = r0.elapsed_time(r1) * 1e3 / (8 * 101)
            fb = 9
            fused["2^%d" % ln] = {
                "per_step_launches": {"us_per_step": us_l, "env_steps_per_s": nn / us_l * 1e6,
                                      "bytes_per_env_step": bytes_per, **roof(nn * bytes_per / us_l / 1e3, True)},
                "fused_with_reward_done_rows": {"us_per_step": us_f, "env_steps_per_s": nn / us_f * 1e6,
                                                "bytes_per_env_step": fb, "achieved_GBps": nn * fb / us_f / 1e3,
                                                "speedup_over_per_step_launches": us_l / us_f},
                "fused_last_step_outputs_only": {"us_per_step": us_f2, "env_steps_per_s": nn / us_f2 * 1e6,
                                                 "bytes_per_env_step": 4, "speedup_over_per_step_launches": us_l / us_f2},
                "fused_rollout_in_kernel_random_policy": {"us_per_step": us_r, "env_steps_per_s": nn / us_r * 1e6,
                                                          "bytes_per_env_step": 0, "speedup_over_per_step_launches": us_l / us_r},
            }
            del ef, af, rows_r, rows_d
            torch.cuda.empty_cache()
        out["fused_step_many"] = dict(fused, note="fishing_step_fused_f32: 101 steps per launch, bit-identical to 101 "
                                                  "fishing_step_f32 launches; VALU-bound (Philox + Box-Muller), not HBM-bound: "
                                                  "reported as env-steps/s, never the headline")
        env = None
        mark("fused_step_many")
        configs = None
        if args.config == "v1" and with_returns and not args.no_configs:
            out["python_step_loop"] = python_step_loop(torch, gf)
            mark("python_step_loop")
            configs = config_records(torch, gf)
            mark("configs")

    if args.extra and rank == 0:
        extra = {}
        if env is not None:
            del env
        del actions
        torch.cuda.empty_cache()
        # fused T-step rollout, in-kernel policy (VALU-bound, not HBM-bound): env-steps/s only
        for pol, param in (("random", 0.0), ("escapement", 0.5)):
            env2 = make_env(gf, torch, args.config, n, 0, True, False, args.v4_stored)
            env2.reset()
            env2.rollout(101, policy=pol, param=param)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            env2.rollout(2020, policy=pol, param=param)
            torch.cuda.synchronize()
            extra["fused_rollout_%s_env_steps_per_s" % pol] = n * 2020 / (time.perf_counter() - t1)
            del env2
        # pure step at sizes that do / do not fit the Infinity Cache (kernel-only, HIP events)
        bb = bytes_per_env_step(args.config, False, False, args.v4_stored)
        for ln in (19, 20, 22, 24, 26):
            nn = 1 << ln
            e3 = make_env(gf, torch, args.config, nn, 0, False, False, args.v4_stored)
            e3.reset()
            acts = make_actions(torch, cfg, nn, 4)
            k = max(20, min(400, (1 << 31) // nn))
            e3.step_many(acts, k)
            us, _ = timed_steps(torch, e3, acts, k)
            extra["step_only_2^%d" % ln] = {"us_per_launch": us, "GBps": nn * bb / us / 1e3, "env_steps_per_s": nn / us * 1e6}
            del e3, acts
            torch.cuda.empty_cache()
        out["extra"] = extra

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.config)
        mark("cpu_baseline")
    elif rank == 0:
        out["cpu_baseline"] = None
    out["bench_wall_s"] = dict(wall_s, total=round(sum(wall_s.values()), 3))
    if subrecords and configs is not None:
        # LAST on the line (the driver keeps the last 8 KB of stdout): the Python loop and every config's own record
        out["python_step_loop"] = out.pop("python_step_loop")
        out["configs"] = configs
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
