"""Where the host time of one Python env.step() goes (N small enough that the GPU is never the bottleneck)."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
n = 1 << 12
env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1); env.reset()
acts = torch.rand((8, n), device="cuda") * 2 - 1
rows = [acts[k] for k in range(8)]
for k in range(200):
    env.step(rows[k % 8])
torch.cuda.synchronize()
K = 20000
t0 = time.perf_counter()
for k in range(K):
    env.step(rows[k & 7])
t1 = time.perf_counter()
torch.cuda.synchronize()
print("step(): %.2f us host per call" % ((t1 - t0) / K * 1e6))
# pieces
t0 = time.perf_counter()
for k in range(K):
    env._prepare_action(rows[k & 7])
print("_prepare_action: %.2f us" % ((time.perf_counter() - t0) / K * 1e6))
t0 = time.perf_counter()
for k in range(K):
    env._step_buffers(0, None)
print("_step_buffers: %.2f us" % ((time.perf_counter() - t0) / K * 1e6))
t0 = time.perf_counter()
for k in range(K):
    torch.cuda.current_device()
print("current_device: %.2f us" % ((time.perf_counter() - t0) / K * 1e6))
t0 = time.perf_counter()
for k in range(K):
    torch.cuda.current_stream().cuda_stream
print("current_stream().cuda_stream: %.2f us" % ((time.perf_counter() - t0) / K * 1e6))
t0 = time.perf_counter()
for k in range(K):
    env._c_params()
print("_c_params: %.2f us" % ((time.perf_counter() - t0) / K * 1e6))
p = env._c_params(); b = env._step_buffers(rows[0].data_ptr(), None); s = torch.cuda.current_stream().cuda_stream
t0 = time.perf_counter()
for k in range(K):
    env._fn_step(p, n, 0, b, 1, k, s)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("bare ctypes fishing_step_f32 call: %.2f us" % ((t1 - t0) / K * 1e6))
pr = cProfile.Profile(); pr.enable()
for k in range(5000):
    env.step(rows[k & 7])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
