"""One-off soak of the round-2 identities at a size and length the test suite does not afford:
(1) fishing-v4 derived vs stored parameters, N = 2^20, 3000 auto-resetting steps (step_many + fused mixed);
(2) fused vs per-step launches, fishing-v1 / v2, N = 2^20, 1500 steps in chunks of uneven length."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
n = 1 << 20
g = torch.Generator(device="cuda").manual_seed(1)
ring = torch.rand((16, n), device="cuda", generator=g) * 1.4 - 1.2
D = gf.make("fishing-v4", num_envs=n, sigma=0.05, sigma_p=0.2, seed=77, track_returns=True)
S = gf.make("fishing-v4", num_envs=n, sigma=0.05, sigma_p=0.2, seed=77, track_returns=True, derived_params=False)
for e in (D, S):
    e.reset()
done = 0
for k, fused in ((700, False), (333, True), (1, False), (966, True), (1000, False)):
    for e in (D, S):
        e.step_many(ring, k, fused=fused)
    done += k
    torch.cuda.synchronize()
    ok = all(torch.equal(getattr(D, a).view(torch.int32) if getattr(D, a).dtype != torch.uint8 else getattr(D, a),
                         getattr(S, a).view(torch.int32) if getattr(S, a).dtype != torch.uint8 else getattr(S, a))
             for a in ("_obs", "_t", "_reward", "_done", "_ep_return"))
    okK = torch.equal(D.K, S.K) and torch.equal(D.r, S.r)
    print("v4 derived vs stored after %d steps: streams %s, K/r %s" % (done, ok, okK), flush=True)
    assert ok and okK
a, b = D.episode_stats(), S.episode_stats()
assert a["n_episodes"] == b["n_episodes"] and a["sum_length"] == b["sum_length"], (a, b)
print("episodes", a["n_episodes"], "mean return", a["mean_return"], b["mean_return"])
for env_id in ("fishing-v1", "fishing-v2"):
    A = gf.make(env_id, num_envs=n, sigma=0.1, seed=5, track_returns=True)
    B = gf.make(env_id, num_envs=n, sigma=0.1, seed=5, track_returns=True)
    for e in (A, B):
        e.reset()
    total = 0
    for k in (101, 7, 500, 1, 390, 501):
        A.step_many(ring, k)
        B.step_many(ring, k, fused=True)
        total += k
        torch.cuda.synchronize()
        ok = all(torch.equal(getattr(A, x), getattr(B, x)) for x in ("_obs", "_t", "_reward", "_done", "_ep_return"))
        print("%s fused vs launches after %d steps: %s" % (env_id, total, ok), flush=True)
        assert ok
    sa, sb = A.episode_stats(), B.episode_stats()
    assert sa["n_episodes"] == sb["n_episodes"] and abs(sa["sum_return"] - sb["sum_return"]) <= 1e-9 * abs(sa["sum_return"])
print("soak ok")
