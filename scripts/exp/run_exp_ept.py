import ctypes, json, os, statistics, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "exp_ept.so")
lib = ctypes.CDLL(so)
lib.exp_step.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 5 + [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
n = 1 << 22
obs = torch.full((n,), -0.25, device="cuda"); t = torch.zeros(n, dtype=torch.int32, device="cuda")
rew = torch.zeros(n, device="cuda"); done = torch.zeros(n, dtype=torch.uint8, device="cuda")
acts = torch.rand((8, n), device="cuda") * 2 - 1
st = torch.cuda.current_stream().cuda_stream
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import gym_fishing_amd as gf
prod = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1)
prod.reset()
cfgs = [(copy, ept, blocks) for copy in (0, 1) for ept in (4, 8) for blocks in (1024, 2048, 4096) if n // (256 * ept) >= blocks // 4]
res = {c: [] for c in cfgs}
res_prod = []
for rnd in range(5):
    prod.step_many(acts, 20)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); prod.step_many(acts, 200); e1.record(); torch.cuda.synchronize()
    res_prod.append(e0.elapsed_time(e1) * 5.0)
    for c in cfgs:
        copy, ept, blocks = c
        blocks = min(blocks, n // (256 * ept))
        for k in range(20):
            lib.exp_step(ept, copy, blocks, n, obs.data_ptr(), acts[k % 8].data_ptr(), rew.data_ptr(), done.data_ptr(), t.data_ptr(), 1, k, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(200):
            lib.exp_step(ept, copy, blocks, n, obs.data_ptr(), acts[k % 8].data_ptr(), rew.data_ptr(), done.data_ptr(), t.data_ptr(), 1, k, st)
        e1.record(); torch.cuda.synchronize()
        res[c].append(e0.elapsed_time(e1) * 5.0)
print(json.dumps({"product_step_kernel_us": round(statistics.median(res_prod), 2)}), flush=True)
for c in cfgs:
    print(json.dumps({"copy": c[0], "ept": c[1], "blocks": min(c[2], n // (256 * c[1])), "med_us": round(statistics.median(res[c]), 2)}), flush=True)
