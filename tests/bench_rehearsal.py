"""CPU stand-in for bench.py's device seam (FISHING_BENCH_RUNTIME="tests.bench_rehearsal:Runtime"): the ORACLE advances
each rank's envs, everything else -- bench.py's self-launch, process group, ranks_seen, per-rank env offsets, action-ring
slices, the product's own record all-reduce (gym_fishing_amd.sharding.all_reduce_record), the barriers, the MAX over ranks,
the JSON relay -- is the code the driver's unattended 8-GPU run will execute.  Test infrastructure: lives under tests/,
never imported by the product; a line produced through it says so (config.rehearsal) and carries no roofline."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import fishing_oracle as fo  # noqa: E402

MODEL = {"v1": fo.MODEL_V1, "v0": fo.MODEL_V0, "v2": fo.MODEL_V2, "v4": fo.MODEL_V4}


class _Clock:
    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class OracleVecEnv:
    """The slice of gym_fishing_amd.envs.BaseFishingEnv that bench.py's timed region touches, on the float32 oracle:
    in-kernel-style noise keyed by (seed, GLOBAL env index, step), fused auto-reset, per-env episodic return and the
    {sum R, sum R^2, n, sum length} record.  fishing-v4: (K, r) redrawn per episode from the parameter stream
    (reset stream at reset(), auto-reset stream keyed by the finishing step), sigma = 0.05 per env."""

    def __init__(self, cfg_name, kwargs, n, env_offset, with_returns, seed=1234):
        self.model, self.n, self.off, self.seed = MODEL[cfg_name], int(n), int(env_offset), seed
        self.kw = dict(kwargs)
        self.with_returns = with_returns
        self.env = np.arange(self.off, self.off + self.n, dtype=np.uint64)
        self.steps = 0
        self.resets = 0
        self.rec = np.zeros(4)
        self.v4 = self.model == fo.MODEL_V4
        self.sigma = np.float32(0.05 if self.v4 else self.kw.get("sigma", 0.0))
        self.K = np.full(self.n, self.kw.get("K_mean", 1.0) if self.v4 else 1.0, np.float32)
        self.r = np.full(self.n, self.kw.get("r_mean", 0.3) if self.v4 else 0.3, np.float32)
        self._draw(fo.STREAM_RESET, self.resets, np.ones(self.n, bool))       # the constructor's draw (fishing_model_error.py:37-38)
        self.resets += 1

    def _draw(self, stream, counter, mask):
        if not self.v4:
            return
        zK, zr = fo.reset_normals(self.seed, self.env, counter, stream)
        K, r = fo.draw_model_error_params(zK, zr, self.kw.get("K_mean", 1.0), self.kw.get("r_mean", 0.3),
                                          self.kw.get("sigma_p", 0.1), np.float32)
        self.K = np.where(mask, K, self.K).astype(np.float32)
        self.r = np.where(mask, r, self.r).astype(np.float32)

    def reset(self):
        self._draw(fo.STREAM_RESET, self.resets, np.ones(self.n, bool))
        self.resets += 1
        self.obs = fo.reset_obs(self.model, 0.75, self.K, np.float32)
        self.t = np.zeros(self.n, np.int32)
        self.ep = np.zeros(self.n, np.float32)

    def step_many(self, actions, k):
        R = actions.shape[0]
        for i in range(int(k)):
            a = actions[i % R].numpy()
            z = fo.noise_normal(self.seed, self.env, self.steps)
            o, rew, d, t2, _ = fo.step(self.model, self.obs, self.t, a, z, self.r, self.K, self.sigma, C=self.kw.get("C", 0.5),
                                       Tmax=100, n_actions=self.kw.get("n_actions", 100), dtype=np.float32)
            m = d.astype(bool)
            if self.with_returns:
                self.ep = (self.ep + rew).astype(np.float32)
                e64 = self.ep[m].astype(np.float64)
                self.rec += [e64.sum(), (e64 * e64).sum(), float(m.sum()), float(t2[m].sum())]
                self.ep = np.where(m, np.float32(0), self.ep)
            self._draw(fo.STREAM_AUTORESET, self.steps, m)
            self.obs = np.where(m, fo.reset_obs(self.model, 0.75, self.K, np.float32), o).astype(np.float32)
            self.t = np.where(m, np.int32(0), t2).astype(np.int32)
            self.steps += 1

    def episode_record(self, all_reduce=True):
        import torch
        from gym_fishing_amd.sharding import all_reduce_record          # the product's merge, over the rehearsal's gloo group
        rec = torch.tensor(self.rec, dtype=torch.float64)
        if not all_reduce and os.environ.get("FISHING_REHEARSAL_CORRUPT_RANK") == os.environ.get("RANK", "0"):
            rec[0] += 1.0              # negative control of bench.py's self-check: this rank's own record is not what was merged
        return all_reduce_record(rec) if all_reduce else rec

    def episode_stats(self, all_reduce=True):
        from gym_fishing_amd.sharding import summarize_record
        return summarize_record(self.episode_record(all_reduce))

    def step_kernel_name(self, actions=None):
        return "oracle/fishing_oracle.py:step (CPU rehearsal)"


class Runtime:
    name = "CPU oracle stand-in (tests/bench_rehearsal.py): control-flow rehearsal, not a measurement"
    cuda = False
    backend = "gloo"
    device = "cpu"

    def claim_device(self, torch, rank, local_rank):
        # the rule the real runtime keeps: a rank that cannot get its device exits non-zero, no re-exec, no retry
        limit = int(os.environ.get("FISHING_REHEARSAL_DEVICES", "1000000"))
        if local_rank + 1 > limit:
            raise SystemExit("bench.py rank %d: LOCAL_RANK=%d but only %d rehearsal device(s)" % (rank, local_rank, limit))
        return local_rank

    def synchronize(self, torch):
        pass

    def event(self, torch):
        return _Clock()

    def make_env(self, gf, torch, cfg_name, n, env_offset, with_returns, compact=False, v4_stored=False, f64=False):
        import bench
        return OracleVecEnv(cfg_name, bench.CONFIGS[cfg_name]["kwargs"], n, env_offset, with_returns)
