"""The committed fixtures are exactly what tests/golden/make_golden.py produces from the
reference mounted in the build container.  Skipped where the reference is absent (GPU box)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference not mounted (only in the build container)")
def test_fixtures_regenerate_bit_for_bit(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py"), "--out", str(tmp_path)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    for name in ("reference_trajectories.npz", "reference_policy_sims.npz", "reference_zoo_trajectories.npz",
                 "reference_seeded_sims.npz", "reference_vec_sims.npz", "reference_vec_sims_v4.npz", "reference_policyfn.npz",
                 "reference_growth_functions.npz"):
        new, old = np.load(tmp_path / name), np.load(os.path.join(GOLDEN, name))
        assert sorted(new.files) == sorted(old.files), name
        for k in old.files:
            a, b = new[k], old[k]
            assert a.dtype == b.dtype and a.shape == b.shape, (name, k)
            if a.dtype.kind == "f":
                assert np.array_equal(a, b, equal_nan=True), (name, k)
            else:
                assert np.array_equal(a, b), (name, k)
    new = json.load(open(tmp_path / "reference_anchors.json"))
    old = json.load(open(os.path.join(GOLDEN, "reference_anchors.json")))
    assert new == old


def test_fixtures_hold_numbers_only():
    """A fixture is data: no reference source text travels in tests/golden/."""
    for name in os.listdir(GOLDEN):
        if name.endswith(".npz"):
            z = np.load(os.path.join(GOLDEN, name))
            for k in z.files:
                assert z[k].dtype.kind in "fiub" or k.endswith("/meta"), (name, k)
                if k.endswith("/meta"):
                    meta = json.loads(str(z[k]))
                    assert set(meta) in ({"id", "kwargs", "seeds", "nsteps", "auto_reset", "init_reset"},
                                         {"id", "kwargs", "policy", "seed", "reps", "S", "msy"},
                                         {"id", "kwargs", "policy", "seed", "num_envs", "n_eval_episodes", "S", "msy"}), (name, k)
