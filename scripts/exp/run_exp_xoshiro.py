"""Counter-based Philox (no state in memory) vs stateful xoshiro128++ (16 B/env state, R+W per step)."""
import ctypes, json, os, statistics, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "exp_ept.so"))
lib.exp_step.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 5 + [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
lib.exp_step_xo.argtypes = [ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 7
n = 1 << 22
obs = torch.full((n,), -0.25, device="cuda"); t = torch.zeros(n, dtype=torch.int32, device="cuda")
rew = torch.zeros(n, device="cuda"); done = torch.zeros(n, dtype=torch.uint8, device="cuda")
state = torch.randint(1, 2 ** 31 - 1, (n, 4), dtype=torch.int32, device="cuda")
acts = torch.rand((8, n), device="cuda") * 2 - 1
st = torch.cuda.current_stream().cuda_stream
res = {"philox_counter_based": [], "xoshiro_state_in_hbm": []}
for rnd in range(5):
    for kind in res:
        def run(k):
            a = acts[k % 8].data_ptr()
            if kind.startswith("philox"):
                lib.exp_step(4, 0, 2048, n, obs.data_ptr(), a, rew.data_ptr(), done.data_ptr(), t.data_ptr(), 1, k, st)
            else:
                lib.exp_step_xo(2048, n, obs.data_ptr(), a, rew.data_ptr(), done.data_ptr(), t.data_ptr(), state.data_ptr(), st)
        for k in range(20):
            run(k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(200):
            run(k)
        e1.record(); torch.cuda.synchronize()
        res[kind].append(e0.elapsed_time(e1) * 5.0)
for kind, bytes_ in (("philox_counter_based", 25), ("xoshiro_state_in_hbm", 57)):
    us = statistics.median(res[kind])
    print(json.dumps({"generator": kind, "bytes_per_env_step": bytes_, "med_us": round(us, 2), "env_steps_per_s": "%.3e" % (n / us * 1e6),
                      "GBps": round(n * bytes_ / us / 1e3)}), flush=True)
z = rew  # sanity: distribution of one step's noise is not checked here (experiment only)
