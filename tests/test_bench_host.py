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
