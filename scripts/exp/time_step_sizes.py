"""Step time of the requests that have NO one-tile form (float64, the catch-alls) over N = 2^22 .. 2^26, back to back,
HIP events -- run once per library variant (FISHING_HIP_LIB).  One JSON line per (case, N)."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
CASES = (("f64_v1", "fishing-v1", dict(sigma=0.1, dtype=torch.float64), (22, 23, 24, 25)),
         ("f64_v1_ret", "fishing-v1", dict(sigma=0.1, dtype=torch.float64, track_returns=True), (22, 23, 24, 25)),
         ("f64_v1_K1.5_ret", "fishing-v1", dict(sigma=0.1, K=1.5, dtype=torch.float64, track_returns=True), (20, 22, 24)),
         ("f32_v1_term_ret", "fishing-v1", dict(sigma=0.1, track_returns=True, record_terminal_obs=True), (22, 23, 24, 25, 26)),
         ("f32_v1_K1.5_ret", "fishing-v1", dict(sigma=0.1, K=1.5, track_returns=True), (23, 24, 26)),
         ("f32_v4_stored_sig_ret", "fishing-v4", dict(sigma=0.05, derived_params=False, track_returns=True), (23, 24)))
only = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] else None
sizes_arg = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else None
for name, env_id, kw, sizes in CASES:
    if only and name not in only:
        continue
    sizes = sizes_arg or sizes
    for ln in sizes:
        n = 1 << ln
        ring = torch.empty((4, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((4, n), device="cuda") * 2 - 1)
        e = gf.make(env_id, num_envs=n, seed=1, **kw)
        e.reset()
        K = 400 if ln <= 21 else 100 if ln <= 23 else 40
        v = []
        for rnd in range(3):
            e.step_many(acts, 20)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e.step_many(acts, K); e1.record(); torch.cuda.synchronize()
            v.append(e0.elapsed_time(e1) * 1e3 / K)
        print(json.dumps(dict(case=name, log2_n=ln, us=round(statistics.median(v), 2), kernel=e.step_kernel_name(acts[0]),
                              lib=os.path.basename(os.path.dirname(os.environ.get("FISHING_HIP_LIB", "product/x"))))), flush=True)
        del e, ring, acts
        torch.cuda.empty_cache()
