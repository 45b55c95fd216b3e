"""Host-side profile of the scalar gym.Env protocol (BASELINE config 1): one env, one launch + sync per step."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
env = gf.make("fishing-v1", sigma=0.0); env.reset()
a = np.array([-0.9375], dtype=np.float32)
for rep in range(3):
    t0 = time.perf_counter(); k = 0
    for _ in range(20):
        done = False; env.reset(); ret = 0.0
        while not done:
            _, r, done, _ = env.step(a); k += 1; ret += r
    dt = time.perf_counter() - t0
    print("scalar protocol: %.2f us/step, %.0f steps/s, return %.4f" % (dt / k * 1e6, k / dt, ret))
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    done = False; env.reset()
    while not done:
        _, r, done, _ = env.step(a)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(10)
