"""One-shot launches (N = 2^19 .. 2^21, one tile per workgroup): the product's bare fishing-v1 step next to a copy-shaped
kernel over the same streams (scripts/exp/exp_ept.hip, copy = 1) -- how much of the small-N launch is the arithmetic?
Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -shared -fPIC scripts/exp/exp_ept.hip -o scripts/exp/exp_ept.so"""
import ctypes, json, os, statistics, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "exp_ept.so"))
lib.exp_step.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 5 + [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import gym_fishing_amd as gf
st = torch.cuda.current_stream().cuda_stream
for ln in (18, 19, 20, 21, 22):
    n = 1 << ln
    obs = torch.full((n,), -0.25, device="cuda"); t = torch.zeros(n, dtype=torch.int32, device="cuda")
    rew = torch.zeros(n, device="cuda"); done = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ring = torch.empty((8, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((8, n), device="cuda") * 2 - 1)
    prod = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1)
    prod.reset()
    blocks = min(4096, n // 1024)
    res = {"product": [], "copy": [], "exp_step": []}
    K = 400
    for rnd in range(5):
        prod.step_many(acts, 50)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); prod.step_many(acts, K); e1.record(); torch.cuda.synchronize()
        res["product"].append(e0.elapsed_time(e1) * 1e3 / K)
        for name, copy in (("copy", 1), ("exp_step", 0)):
            for k in range(50):
                lib.exp_step(4, copy, blocks, n, obs.data_ptr(), acts[k % 8].data_ptr(), rew.data_ptr(), done.data_ptr(), t.data_ptr(), 1, k, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(K):
                lib.exp_step(4, copy, blocks, n, obs.data_ptr(), acts[k % 8].data_ptr(), rew.data_ptr(), done.data_ptr(), t.data_ptr(), 1, k, st)
            e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) * 1e3 / K)
    print(json.dumps({"log2_n": ln, "us_per_launch_back_to_back": {k: round(statistics.median(v), 2) for k, v in res.items()},
                      "note": "host-enqueued from Python for copy / exp_step (may be enqueue-bound at the smallest sizes), C-enqueued for the product"}), flush=True)
    del prod, obs, t, rew, done, ring, acts
