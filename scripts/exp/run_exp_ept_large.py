"""exp_ept.hip at N = 2^26 (HBM-resident): 4 vs 8 envs per thread (16 vs 32 contiguous bytes per lane and stream), step and
copy-shaped, a few workgroup counts, next to the product kernel in the same process."""
import ctypes, json, os, statistics, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "exp_ept.so"))
lib.exp_step.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 5 + [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import gym_fishing_amd as gf
n = 1 << 26
obs = torch.full((n,), -0.25, device="cuda"); t = torch.zeros(n, dtype=torch.int32, device="cuda")
rew = torch.zeros(n, device="cuda"); done = torch.zeros(n, dtype=torch.uint8, device="cuda")
ring = torch.empty((2, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((2, n), device="cuda") * 2 - 1)
st = torch.cuda.current_stream().cuda_stream
prod = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1)
prod.reset()
cfgs = [(copy, ept, blocks) for copy in (0, 1) for ept in (4, 8) for blocks in (768, 1536, 4096)]
res = {c: [] for c in cfgs}
res_prod = []
K = 30
for rnd in range(3):
    prod.step_many(acts, 10)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); prod.step_many(acts, K); e1.record(); torch.cuda.synchronize()
    res_prod.append(e0.elapsed_time(e1) * 1e3 / K)
    for c in cfgs:
        copy, ept, blocks = c
        for k in range(6):
            lib.exp_step(ept, copy, blocks, n, obs.data_ptr(), acts[k % 2].data_ptr(), rew.data_ptr(), done.data_ptr(), t.data_ptr(), 1, k, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(K):
            lib.exp_step(ept, copy, blocks, n, obs.data_ptr(), acts[k % 2].data_ptr(), rew.data_ptr(), done.data_ptr(), t.data_ptr(), 1, k, st)
        e1.record(); torch.cuda.synchronize()
        res[c].append(e0.elapsed_time(e1) * 1e3 / K)
print(json.dumps({"product_step_kernel_us (zig-zag walk, 768 workgroups)": round(statistics.median(res_prod), 1)}), flush=True)
for c in cfgs:
    print(json.dumps({"copy": c[0], "ept": c[1], "blocks": c[2], "med_us (forward walk)": round(statistics.median(res[c]), 1)}), flush=True)
