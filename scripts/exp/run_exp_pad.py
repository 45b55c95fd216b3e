"""Does staggering the streams' base addresses matter? (same stripped kernel, arena placement)"""
import ctypes, json, os, statistics, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "exp_ept.so"))
lib.exp_step.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 5 + [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
n = 1 << 22
arena = torch.zeros(512 << 20, dtype=torch.uint8, device="cuda")
base = arena.data_ptr()
base = (base + (1 << 21) - 1) & ~((1 << 21) - 1)
acts = torch.rand((8, n), device="cuda") * 2 - 1
st = torch.cuda.current_stream().cuda_stream
pads = [0, 4096, 8192, 12288, 16384, 20480, 24576, 49152, 12288 + 256]
def place(pad):
    sizes = [4 * n, 4 * n, 4 * n, n]     # obs, reward, t, done
    ptrs, off = [], 0
    for k, sz in enumerate(sizes):
        ptrs.append(base + off)
        off += sz + pad * (k + 1)
        off = (off + 255) & ~255
    return ptrs
res = {p: [] for p in pads}
for p in pads:   # init obs/t
    obs, rew, t, done = place(p)
for rnd in range(6):
    for p in pads:
        obs, rew, t, done = place(p)
        arena.zero_()
        for k in range(20):
            lib.exp_step(4, 0, 2048, n, obs, acts[k % 8].data_ptr(), rew, done, t, 1, k, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(200):
            lib.exp_step(4, 0, 2048, n, obs, acts[k % 8].data_ptr(), rew, done, t, 1, k, st)
        e1.record(); torch.cuda.synchronize()
        res[p].append(e0.elapsed_time(e1) * 5.0)
# tile -> workgroup mapping at the best-known stagger
maps = {4: "strided", 41: "contiguous per block", 42: "contiguous per XCD group"}
resm = {m: [] for m in maps}
obs, rew, t, done = place(12288)
for rnd in range(6):
    for m in maps:
        for blocks in (2048,):
            for k in range(20):
                lib.exp_step(m, 0, blocks, n, obs, acts[k % 8].data_ptr(), rew, done, t, 1, k, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(200):
                lib.exp_step(m, 0, blocks, n, obs, acts[k % 8].data_ptr(), rew, done, t, 1, k, st)
            e1.record(); torch.cuda.synchronize()
            resm[m].append(e0.elapsed_time(e1) * 5.0)
for m, name in maps.items():
    print(json.dumps({"map": name, "med_us": round(statistics.median(resm[m]), 2), "min_us": round(min(resm[m]), 2)}), flush=True)
for p in pads:
    print(json.dumps({"pad_bytes": p, "med_us": round(statistics.median(res[p]), 2), "min_us": round(min(res[p]), 2)}), flush=True)
