"""bench.py's host-side pieces that need no GPU: the sharded action ring is a slice of the global one, the
--gpus N entry point starts child ranks and relays their failure loudly, byte accounting per config."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.mark.parametrize("cfg", ["v1", "v0", "v2"])
def test_sharded_action_rings_are_slices_of_the_global_ring(cfg):
    """The random policy's actions are keyed by the global env index (chunks of 2^16 envs, one generator seed per
    chunk), so 4 ranks x n envs step exactly the workload of 1 rank x 4n envs -- also for shard sizes that are not
    multiples of the chunk."""
    c = bench.CONFIGS[cfg]
    for n in (1 << 16, 3 * (1 << 15), 40000):
        whole = bench.make_actions(torch, c, 4 * n, 3, 0, pad=64, device="cpu")
        for rank in range(4):
            part = bench.make_actions(torch, c, n, 3, rank * n, pad=64, device="cpu")
            assert torch.equal(part, whole[:, rank * n:(rank + 1) * n]), (cfg, n, rank)
        lo, hi = c["actions"][1:]
        assert float(whole.min()) >= lo and float(whole.max()) < hi
        assert whole.stride(0) == 4 * n + 64          # padded rows


def test_bytes_per_env_step_accounting():
    assert bench.bytes_per_env_step("v1", False) == 25 and bench.bytes_per_env_step("v1", True) == 33
    assert bench.bytes_per_env_step("v0", True) == 33 and bench.bytes_per_env_step("v2", True) == 33
    assert bench.bytes_per_env_step("v1", False, compact=True) == 19
    assert bench.bytes_per_env_step("v4", False) == 29              # derived (K, r): sigma array only
    assert bench.bytes_per_env_step("v4", True) == 37
    # stored r / K arrays: read every step AND rewritten by the redraw on this workload (PMC-measured, round 2): 45 / 53 B
    assert bench.bytes_per_env_step("v4", False, v4_stored=True) == 45
    assert bench.bytes_per_env_step("v4", True, v4_stored=True) == 53
    assert bench.bytes_per_env_step("v1", False, f64=True) == 37 and bench.bytes_per_env_step("v1", True, f64=True) == 53
    # fishing-v4 derived with per-env origin stamps (after a masked reset): R 4 + W 4 on top of 37
    assert bench.bytes_per_env_step("v4", True, v4_stamped=True) == 45 and bench.bytes_per_env_step("v4", False, v4_stamped=True) == 37


@pytest.mark.skipif(torch.cuda.is_available(), reason="the failure path: only where no HIP device exists")
def test_gpus_n_self_launch_relays_child_failure():
    """`python bench.py --gpus 2` without a torch.distributed.run parent starts its two ranks as children; here they
    fail (no device), and the parent must exit non-zero with nothing that looks like a result on stdout."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                           "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=300)
    assert proc.returncode != 0
    assert proc.stdout.strip() == ""
    assert "needs a HIP device" in proc.stderr


# ------------------------------------------------------------------ world = 8, rehearsed on the CPU
def _rehearse(gpus, config, n_envs, steps=3, warmup=1, extra_env=None, timeout=600):
    """`python bench.py --gpus N ...` through its own self-launch, with the CPU oracle behind the device seam."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(FISHING_BENCH_RUNTIME="tests.bench_rehearsal:Runtime", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    env.update(extra_env or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", str(steps), "--warmup", str(warmup),
           "--spinup-ms", "0", "--config", config, "--n-envs", str(n_envs), "--rehearsal"]
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("config", ["v2", "v4"])
def test_eight_rank_rehearsal_of_the_bench_rank_path(config):
    """The first 8-GPU run will be unattended.  This is that run's control flow, at world = 8, on the CPU: `bench.py
    --gpus 8` starts eight child ranks (torch.distributed.run, 127.0.0.1), each claims its device, joins the process group
    (gloo here), counts the ranks with an all-reduce of ones, builds its envs at offset rank * n and its slice of the
    global action ring, runs spin-up / warm-up / dress rehearsal / the timed region with the record's all-reduce and the
    closing barrier inside it, the elapsed time is MAX-reduced, rank 0 prints ONE JSON line and the parent relays it.  The
    ORACLE advances the envs (tests/bench_rehearsal.py), so the merged record is checkable: it equals the record of ONE
    rank stepping all 8 n envs -- BASELINE configs 4 (fishing-v2) and 5 (fishing-v4, (K, r) redrawn per episode from the
    global env index)."""
    n, steps, warmup = 2048, 3, 1
    proc = _rehearse(8, config, n, steps, warmup)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, proc.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["gloo_ranks_seen"] == 8 and out["scaling"] == "weak"
    assert out["config"]["global_envs"] == 8 * n and out["config"]["envs_per_gpu"] == n
    assert "rehearsal" in out["config"] and out["roofline"] is None and out["cpu_baseline"] is None
    assert out["config"]["collective"].startswith("1 all-reduce of 4 doubles")
    assert out["value"] == pytest.approx(8 * n * steps / (out["ms_per_step"] * steps * 1e-3), rel=1e-9)
    # the run describes itself: every rank's device, env block, own time and own record; the merged record is their sum
    ranks = out["config"]["ranks"]
    assert [d["rank"] for d in ranks] == list(range(8)) and [d["device"] for d in ranks] == list(range(8))
    assert [(d["env_offset"], d["n_envs"]) for d in ranks] == [(r * n, n) for r in range(8)]
    assert out["config"]["record_is_sum_of_rank_records"] is True
    assert out["ms_per_step_max_over_ranks"] == pytest.approx(out["ms_per_step"], rel=1e-12)     # MAX over ranks IS the line's time
    assert 0 < out["ms_per_step_min_over_ranks"] <= out["ms_per_step_max_over_ranks"]
    assert out["ms_per_step_min_over_ranks"] == min(d["ms_per_step"] for d in ranks)
    assert sum(d["record"][2] for d in ranks) == out["episode_stats"]["n_episodes"]
    assert all(d["record"][2] > 0 for d in ranks)                # every rank finished episodes of its own
    one = _rehearse(1, config, 8 * n, steps, warmup)
    assert one.returncode == 0, one.stderr[-3000:]
    ref = json.loads([ln for ln in one.stdout.splitlines() if ln.strip()][0])
    assert ref["n_gpus"] == 1 and ref["config"]["gloo_ranks_seen"] is None and ref["config"]["ranks"] is None
    a, b = out["episode_stats"], ref["episode_stats"]
    assert a["n_episodes"] == b["n_episodes"] > 0
    for k in ("mean_return", "std_return", "mean_length"):
        assert a[k] == pytest.approx(b[k], rel=1e-9), k


@pytest.mark.timeout(600)
def test_rehearsed_rank_without_a_device_fails_loudly():
    """`--gpus 8` on a node that shows six devices: ranks 6 and 7 exit non-zero with a one-line diagnosis, no re-exec, no
    retry; torch.distributed.run tears the others down and the parent relays the failure -- no JSON line."""
    proc = _rehearse(8, "v1", 1024, extra_env={"FISHING_REHEARSAL_DEVICES": "6"}, timeout=500)
    assert proc.returncode != 0 and proc.stdout.strip() == ""
    assert "only 6 rehearsal device(s)" in proc.stderr


@pytest.mark.timeout(600)
def test_a_rank_whose_record_is_not_in_the_merged_one_fails_the_run():
    """The self-check of the unattended run: a rank whose own record is not what went into the all-reduce (here: rank 1's
    local record is doctored after the merge) makes every rank exit non-zero with the reason, and no JSON line appears."""
    proc = _rehearse(2, "v1", 1024, extra_env={"FISHING_REHEARSAL_CORRUPT_RANK": "1"}, timeout=500)
    assert proc.returncode != 0 and proc.stdout.strip() == ""
    assert "return-record self-check FAILED" in proc.stderr and "ranks' own records sum to" in proc.stderr


@pytest.mark.skipif(torch.cuda.is_available(), reason="the refusal path: only where no HIP device exists")
def test_the_runtime_variable_alone_does_not_reroute_the_bench():
    """FISHING_BENCH_RUNTIME is honoured only together with --rehearsal: the driver's command, whatever its environment,
    runs on the HIP runtime (here: fails for want of a device instead of silently timing the oracle)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FISHING_BENCH_RUNTIME"] = "tests.bench_rehearsal:Runtime"
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--n-envs", "1024",
                           "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert proc.returncode != 0 and proc.stdout.strip() == "" and "needs a HIP device" in proc.stderr


def test_roof_labels_never_put_a_fraction_above_one():
    a = bench.roof(7200.0, True)
    assert a["frac"] == pytest.approx(0.9) and a["cache_resident"] is True and "hbm_spec_ratio" not in a
    b = bench.roof(8055.0, True)
    assert b["frac"] is None and b["hbm_spec_ratio"] == pytest.approx(8055.0 / 8000.0) and b["cache_resident"] is True
    # what decides the regime: state streams + action ring against the 256 MiB Infinity Cache
    assert bench.resident_bytes("v1", 1 << 22, True, bench.RING) < 256 * 2 ** 20 < bench.resident_bytes("v1", 1 << 24, True, 4)
    assert bench.resident_bytes("v4", 1 << 21, True, bench.RING) < 256 * 2 ** 20 < bench.resident_bytes("v4", 1 << 24, True, 4)
    keys = [k for k, *_ in bench.CONFIG_RECORDS]
    assert len(set(keys)) == len(keys) and {c for _, c, *_ in bench.CONFIG_RECORDS} == {"v0", "v1", "v2", "v4"}


def test_config_shards_of_eight_ranks_are_quad_aligned_and_tile():
    """The per-rank blocks the 8-GPU run will own: BASELINE config 4 (2^19 per rank) and config 5 (2^21 per rank) --
    offsets are multiples of 4 (noise quads, 16-byte rows) and of the 2^16-env action chunks, the blocks tile
    [0, 8 n) exactly, and sharding.shard_range gives the same blocks for the global batch."""
    from gym_fishing_amd import sharding
    for name in ("v2", "v4"):
        n = 1 << bench.CONFIGS[name]["log2_n_multi"]
        assert n == (1 << 19 if name == "v2" else 1 << 21)
        blocks = [sharding.shard_range(8 * n, r, 8) for r in range(8)]
        assert blocks == [(r * n, n) for r in range(8)]
        assert all(off % 4 == 0 and off % bench.ACTION_CHUNK == 0 for off, _ in blocks)
