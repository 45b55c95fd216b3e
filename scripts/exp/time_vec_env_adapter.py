"""Host time per step of the NumPy VecEnv adapter at SB3-like sizes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gym_fishing_amd.vec_env import make_vec_env
for n in (8, 64, 1024, 16384):
    venv = make_vec_env("fishing-v1", n, sigma=0.1, seed=1)
    venv.reset()
    a = np.random.default_rng(0).uniform(-1, -0.5, (n, 1)).astype(np.float32)
    for _ in range(50):
        venv.step(a)
    K = 1000
    t0 = time.perf_counter()
    for _ in range(K):
        venv.step(a)
    dt = time.perf_counter() - t0
    print("N=%d: %.1f us per step_wait, %.3e env-steps/s" % (n, dt / K * 1e6, n * K / dt), flush=True)
import cProfile, pstats
venv = make_vec_env("fishing-v1", 64, sigma=0.1, seed=1); venv.reset()
a = np.random.default_rng(0).uniform(-1, -0.5, (64, 1)).astype(np.float32)
for _ in range(100):
    venv.step(a)
pr = cProfile.Profile(); pr.enable()
for _ in range(1000):
    venv.step(a)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
