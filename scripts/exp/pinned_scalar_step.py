import sys, time, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))); sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))), "tests"))
from gym_fishing_amd import _capi
torch.cuda.init()
lib = _capi.lib()
arena = torch.zeros(4096, dtype=torch.uint8).pin_memory()
obs = arena[0:8].view(torch.float64); t = arena[256:260].view(torch.int32); rew = arena[512:520].view(torch.float64); done = arena[768:769]
act = arena[1024:1028].view(torch.float32)
obs[0] = -0.25; act[0] = -0.9375
p = _capi.FishingParams(); p.model, p.Tmax = 1, 100; p.r, p.K, p.sigma, p.C, p.x0 = 0.3, 1.0, 0.0, 0.5, 0.75
b = _capi.make_buffers(obs=obs.data_ptr(), action=act.data_ptr(), reward=rew.data_ptr(), done=done.data_ptr(), t=t.data_ptr())
st = torch.cuda.current_stream()
rc = lib.fishing_step_f64(p, 1, 0, b, 0, 0, st.cuda_stream); st.synchronize()
print("rc", rc, float(obs[0]).hex(), float(rew[0]), int(t[0]), int(done[0]))
K = 3000
t0 = time.perf_counter()
for k in range(K):
    lib.fishing_step_f64(p, 1, 0, b, 0, k, st.cuda_stream); st.synchronize()
    o = float(obs[0])
dt = time.perf_counter() - t0
print("zero-copy scalar step: %.1f us/step" % (dt / K * 1e6))
