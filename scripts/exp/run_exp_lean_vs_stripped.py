"""Same buffers (arena, 12 KiB stagger), same launch loop: stripped experiment kernel vs the product's
lean kernel (through the C ABI) vs a same-shape copy."""
import ctypes, json, os, statistics, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from gym_fishing_amd import _capi
lib = ctypes.CDLL(os.path.join(here, "exp_ept.so"))
lib.exp_step.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 5 + [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
prod = _capi.lib()
n = 1 << 22
arena = torch.zeros(512 << 20, dtype=torch.uint8, device="cuda")
base = (arena.data_ptr() + (1 << 21) - 1) & ~((1 << 21) - 1)
ring = torch.empty((8, n + 3072), device="cuda")
acts = ring[:, :n]
acts.copy_(torch.rand((8, n), device="cuda") * 2 - 1)
st = torch.cuda.current_stream().cuda_stream
sizes = [4 * n, 4 * n, 4 * n, n]
ptrs, off = [], 0
for k, sz in enumerate(sizes):
    ptrs.append(base + off)
    off = (off + sz + 12288 * (k + 1) + 255) & ~255
obs, t, rew, done = ptrs
p = _capi.FishingParams()
p.model, p.Tmax, p.flags = 1, 100, 1
p.r, p.K, p.sigma, p.C, p.x0 = 0.3, 1.0, 0.1, 0.5, 0.75
pg = _capi.FishingParams.from_buffer_copy(p)
pg.flags = 3
def run(kind, k):
    a = acts[k % 8].data_ptr()
    if kind == "stripped":
        lib.exp_step(4, 0, 2048, n, obs, a, rew, done, t, 1, k, st)
    elif kind.startswith("pipelined"):
        lib.exp_step(43, 0, int(kind.split("_")[1]), n, obs, a, rew, done, t, 1, k, st)
    elif kind.startswith("copy"):
        lib.exp_step(4, 1, int(kind.split("_")[1]) if "_" in kind else 2048, n, obs, a, rew, done, t, 1, k, st)
    else:
        b = _capi.make_buffers(obs=obs, action=a, reward=rew, done=done, t=t)
        rc = prod.fishing_step_f32(p if kind == "lean" else pg, n, 0, b, 1, k, st)
        assert rc == 0
kinds = ["stripped", "pipelined_2048", "lean", "general", "copy", "copy_4096"]
res = {k: [] for k in kinds}
for rnd in range(6):
    for kind in kinds:
        arena.zero_()
        for k in range(20):
            run(kind, k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(200):
            run(kind, k)
        e1.record(); torch.cuda.synchronize()
        res[kind].append(e0.elapsed_time(e1) * 5.0)
for kind in kinds:
    print(json.dumps({"kernel": kind, "med_us": round(statistics.median(res[kind]), 2), "min_us": round(min(res[kind]), 2)}), flush=True)
