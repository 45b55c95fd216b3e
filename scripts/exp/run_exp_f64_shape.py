"""fp64 parity layout: copy-shaped kernel with 4 (the product's shape) vs 2 envs per thread, next to fishing_step_f64.
Build first (no GPU needed):  hipcc -O3 --offload-arch=gfx950 -shared -fPIC scripts/exp/exp_f64_shape.hip -o scripts/exp/exp_f64_shape.so"""
import ctypes, json, os, statistics, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "exp_f64_shape.so"))
lib.exp_shape.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 6
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import gym_fishing_amd as gf
for ln in (22, 24):
    n = 1 << ln
    obs = torch.full((n,), -0.25, device="cuda", dtype=torch.float64); t = torch.zeros(n, dtype=torch.int32, device="cuda")
    rew = torch.zeros(n, device="cuda", dtype=torch.float64); done = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ring = torch.empty((4, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((4, n), device="cuda") * 2 - 1)
    st = torch.cuda.current_stream().cuda_stream
    prod = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1, dtype=torch.float64)
    prod.reset()
    cfgs = [(ept, blocks, threads) for ept in (4, 2) for threads in (256, 512) for blocks in (1024, 2048, 4096, 8192)]
    res = {c: [] for c in cfgs}
    res_prod = []
    K = 200 if ln == 22 else 60
    for rnd in range(3):
        prod.step_many(acts, 20)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); prod.step_many(acts, K); e1.record(); torch.cuda.synchronize()
        res_prod.append(e0.elapsed_time(e1) * 1e3 / K)
        for c in cfgs:
            ept, blocks, threads = c
            blocks = min(blocks, n // (threads * ept))
            args = lambda k: (ept, blocks, threads, n, obs.data_ptr(), acts[k % 4].data_ptr(), rew.data_ptr(), done.data_ptr(), t.data_ptr(), st)
            for k in range(20):
                lib.exp_shape(*args(k))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(K):
                lib.exp_shape(*args(k))
            e1.record(); torch.cuda.synchronize()
            res[c].append(e0.elapsed_time(e1) * 1e3 / K)
    print(json.dumps({"log2_n": ln, "product_fishing_step_f64_us": round(statistics.median(res_prod), 2),
                      "TBps": round(n * 37 / statistics.median(res_prod) / 1e6, 2)}), flush=True)
    for c in cfgs:
        us = statistics.median(res[c])
        print(json.dumps({"log2_n": ln, "ept": c[0], "blocks": c[1], "threads": c[2], "us": round(us, 2), "TBps": round(n * 37 / us / 1e6, 2)}), flush=True)
    del prod, obs, t, rew, done, ring, acts
    torch.cuda.empty_cache()
