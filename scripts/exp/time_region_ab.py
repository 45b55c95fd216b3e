"""The driver's 20-step timed region behind different preambles, 40 samples each (median / p10 / p90 wall us): the dress rehearsal
with / without its event packets, with / without reading them, behind 20 ms of host idleness, and behind such a gap followed by
bench.py's short second spin.  Result (profiles/r06_region_preamble_ab.json): 395-417 us whatever the events do; an idle gap
costs ~7 us, the second spin takes it back.

    python scripts/exp/time_region_ab.py > profiles/r06_region_preamble_ab.json
"""
import gc, json, os, statistics, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, bench
import gym_fishing_amd as gf
n = 1 << 22
env = bench.make_env(gf, torch, "v1", n, 0, True); env.reset()
actions = bench.make_actions(torch, bench.CONFIGS["v1"], n, bench.RING)
bench.spin_up(torch, env, actions, 300.0); env.episode_stats()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 20
def region(events):
    if events: ev0.record()
    env.step_many(actions, K)
    if events: ev1.record()
    rec = env.episode_record()
    torch.cuda.synchronize()
    return rec
gc.collect(); gc.disable()
res = {}
for name, ev, read, idle_ms in (("A_events_read", True, True, 0), ("B_noevents", False, False, 0), ("C_events_noread", True, False, 0),
                                ("D_idle20ms_then_rehearsal", True, True, 20), ("E_idle20ms_respin", True, True, -20), ("A2", True, True, 0)):
    walls = []
    for _ in range(40):
        if idle_ms > 0:
            time.sleep(idle_ms / 1e3)
        if idle_ms < 0:
            time.sleep(-idle_ms / 1e3)
            bench.spin_up(torch, env, actions, 30.0)
        torch.cuda.synchronize()
        region(ev)
        if read: ev0.elapsed_time(ev1)
        t0 = time.perf_counter(); region(False); walls.append((time.perf_counter() - t0) * 1e6)
    walls.sort()
    res[name] = {"median": round(statistics.median(walls), 1), "p10": round(walls[4], 1), "p90": round(walls[36], 1)}
print(json.dumps(res))
