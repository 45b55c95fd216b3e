"""N = 2^26 bare fishing-v1 step through the C ABI with every stream in its OWN allocation (obs, t, reward, done:
four hipMallocs, each start offset by k * 12 KiB inside its allocation) instead of one arena; all re-allocated per trial.
Is the fast / slow alternation of the arena (placement_large.py) still there?

    python scripts/exp/placement_separate.py [log2_n] [trials] [mode]
      mode 0: four separate allocations; mode 1: one arena (the env's layout) for comparison; mode 2: separate, allocated
      in reverse order
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
from gym_fishing_amd import _capi  # noqa: E402
import hip_harness as hh  # noqa: E402


def main():
    ln = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    n = 1 << ln
    cfg = bench.CONFIGS["v1"]
    acts = bench.make_actions(torch, cfg, n, 2)
    p = hh.params(_capi.MODEL_V1, sigma=0.1, auto_reset=True)
    lib = _capi.lib()
    stag = 12288
    for trial in range(trials):
        sizes = [n * 4, n * 4, n * 4, n]
        if mode == 1:
            offs, off = [], 0
            for k, nb in enumerate(sizes):
                offs.append(off)
                off = (off + nb + stag * (k + 1) + 255) & ~255
            arena = torch.zeros(off, dtype=torch.uint8, device="cuda")
            raw = [arena[o:o + nb] for o, nb in zip(offs, sizes)]
            holders = [arena]
        else:
            order = range(4) if mode == 0 else reversed(range(4))
            holders = [None] * 4
            for k in order:
                holders[k] = torch.zeros(sizes[k] + stag * 4, dtype=torch.uint8, device="cuda")
            raw = [holders[k][stag * k:stag * k + sizes[k]] for k in range(4)]
        obs, t, rew, done = raw[0].view(torch.float32), raw[1].view(torch.int32), raw[2].view(torch.float32), raw[3]
        obs.fill_(-0.25)
        b = _capi.make_buffers(obs=obs.data_ptr(), action=acts.data_ptr(), reward=rew.data_ptr(), done=done.data_ptr(),
                               t=t.data_ptr())
        st = torch.cuda.current_stream().cuda_stream
        count = 0

        def run(k):
            nonlocal count
            rc = lib.fishing_step_many_f32(p, n, 0, b, acts.stride(0), 2, k, 1234, count, st)
            assert rc == 0, rc
            count += k

        run(40)
        torch.cuda.synchronize()
        best = []
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run(20)
            e0.record()
            run(40)
            e1.record()
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) * 1e3 / 40)
        print(json.dumps({"log2_n": ln, "mode": mode, "trial": trial, "us": min(best), "TBps": n * 25 / min(best) / 1e6,
                          "addresses": [hex(x.data_ptr()) for x in raw]}), flush=True)
        del obs, t, rew, done, raw, holders
        if mode == 1:
            del arena
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
