#!/usr/bin/env python3
"""How close is the device exp() path to the reference's np.exp on the fishing-v2 golden steps?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden_cases
import hip_harness as hh
tot = same = 0
worst = 0
for c in load_golden_cases():
    if c.id != "fishing-v2":
        continue
    t_in = np.where(np.arange(c.nsteps)[None, :] == 0, 0, np.roll(c.t, 1, axis=1))
    prev_done = np.roll(c.done, 1, axis=1).astype(bool); prev_done[:, 0] = False
    t_in = np.where(prev_done & c.auto_reset, 0, t_in)
    p = hh.params(2, r=c.param("r"), K=float(c.param("K")), sigma=c.param("sigma"), C=c.param("C"), x0=c.param("init_state"), Tmax=c.param("Tmax"))
    st = hh.State(c.obs.size, np.float64, 2, c.obs_in.reshape(-1), t=t_in.reshape(-1))
    obs, *_ = st.step(p, c.action.reshape(-1), z=c.z.reshape(-1))
    d = hh.ulp_diff((obs + 1.0), (c.obs.reshape(-1) + 1.0))
    tot += d.size; same += int((obs == c.obs.reshape(-1)).sum()); worst = max(worst, int(d.max()))
print("fishing-v2 fp64 golden steps: %d, bit-identical obs: %d (%.1f%%), worst population difference: %d ulp" % (tot, same, 100.0 * same / tot, worst))
