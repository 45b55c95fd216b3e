"""Rollout recording = the reference's simulate helpers (gym_fishing/envs/shared_env.py:29-102)
producing the same `[time, state, action, reward, rep]` table.

* scalar protocol: the reference's loop, verbatim in behaviour (record before acting, quota
  and reward of the previous step, break on done, at most Tmax rows per rep).
* N-env tensor protocol: every env is one rep; a model that carries `kernel_policy`
  (policies.msy / policies.escapement, or the strings "random" / ("constant", a)) runs inside
  the fused rollout kernel and the table is cut from its `[T][4][N]` record; any other model
  is driven step by step with batched predict().
* simulate_mdp_vec: the reference's own N-env helper (shared_env.py:57-79), row for row: Tmax + 1 rows per env
  and batch, time-major, no break on done (the batch auto-resets like SB3's VecEnv), the RAW ACTION of the
  previous step in the action column (pinned by tests/golden/reference_vec_sims.npz).
Returns a pandas DataFrame when pandas is importable, else a dict of NumPy columns.
"""
import numpy as np
import torch

from ._capi import POLICY_CONSTANT, POLICY_RANDOM

COLUMNS = ["time", "state", "action", "reward", "rep"]


def _table(rows):
    arr = np.asarray(rows, dtype=np.float64).reshape(-1, 5)
    try:
        from pandas import DataFrame
        df = DataFrame(arr, columns=COLUMNS)
        df["rep"] = df["rep"].astype(int)
        return df
    except ImportError:
        return {c: arr[:, i] for i, c in enumerate(COLUMNS)}


def _kernel_policy(model):
    if isinstance(model, str) and model == "random":
        return POLICY_RANDOM, 0.0
    if isinstance(model, tuple) and len(model) == 2 and model[0] == "constant":
        return POLICY_CONSTANT, float(model[1])
    return getattr(model, "kernel_policy", None)


class _StringPolicy:
    """predict() for the policy strings the fused kernel understands, for the step-by-step path: a constant action, or
    uniform random actions from torch's generator (NOT the in-kernel Philox policy stream)."""

    def __init__(self, env, kp):
        self.env, self.kp = env, kp

    def predict(self, obs, **kw):
        env, (pol, param) = self.env, self.kp
        n = env.num_envs
        if pol == POLICY_CONSTANT:
            if env.MODEL == 0:
                return torch.full((n,), int(param), dtype=torch.int32, device=env.device), None
            return torch.full((n,), float(param), dtype=torch.float32, device=env.device), None
        if env.MODEL == 0:
            return torch.randint(0, env.n_actions, (n,), dtype=torch.int32, device=env.device), None
        return torch.rand(n, device=env.device) * 2.0 - 1.0, None


def simulate_mdp(env, model, reps=1):
    """shared_env.py:29-54.  With an N-env batch every env of every batch is one rep of that table."""
    if not env._scalar:
        return _simulate_mdp_batched(env, model, reps * env.num_envs)
    rows = []
    for rep in range(reps):
        obs = env.reset()
        quota, reward = 0.0, 0.0
        for t in range(env.Tmax):
            rows.append([t, env.get_fish_population(obs), quota, reward, int(rep)])
            action, _ = model.predict(obs)
            obs, reward, done, _ = env.step(action)
            if isinstance(action, np.ndarray):
                action = action.reshape(-1)[0]
            quota = env.get_quota(action)
            if done:
                break
    return _table(rows)


def _cut_tables(env, traj, rep0):
    """[T][4][N] record {obs_in, action, reward, done} -> rows of the simulate table."""
    T, _, N = traj.shape
    obs_in, act, rew, done = (traj[:, k].to(torch.float64) for k in range(4))
    K = env._K_view().to(torch.float64).reshape(1, N) if env._per_env else float(env.params["K"])
    state = (obs_in + 1.0) * K                                   # get_fish_population :158-160
    if env.MODEL == 0:
        quota = (act / env.n_actions) * K                        # get_quota :140
    else:
        quota = (act.clamp(-1.0, 1.0) + 1.0) * K                 # get_quota :143-146 (act is float32-valued)
    zeros = torch.zeros((1, N), dtype=torch.float64, device=traj.device)
    quota_prev = torch.cat([zeros, quota[:-1]])
    rew_prev = torch.cat([zeros, rew[:-1]])
    ended = torch.cumsum(done, 0) - done          # > 0 on rows after the episode's last step
    keep = ended == 0
    t_idx = torch.arange(T, device=traj.device, dtype=torch.float64).reshape(T, 1).expand(T, N)
    rep = (rep0 + torch.arange(N, device=traj.device, dtype=torch.float64)).reshape(1, N).expand(T, N)
    cols = torch.stack([t_idx, state, quota_prev, rew_prev, rep], dim=-1)     # [T, N, 5]
    cols = cols.permute(1, 0, 2)[keep.t()]                                    # rep-major, time order
    return cols.cpu().numpy()


def simulate_mdp_vec(env, model, n_eval_episodes):
    """shared_env.py:57-79, row for row.  `env` is an N-env batch (auto-reset is switched on for the call: the
    reference relies on the VecEnv's).  Per batch ("rep" in the reference's loop) and t in range(Tmax): one row per
    env [t, population of obs_i, action_i, reward_i, rep * N + i] BEFORE acting -- action / reward are the previous
    step's (initially action_space.low[0] and 0) -- then a final block of rows at t = Tmax; no break on done.
    The action column holds the raw action (not the quota simulate_mdp records).  A model with `kernel_policy`
    runs inside the fused rollout kernel; any other -- and fishing-v4, whose K changes at every auto-reset inside the
    table, so that each row needs the K in force at that row (df_entry_vec asks the env itself, shared_env.py:15-26) --
    is driven step by step with batched predict().
    fishing-v0: the reference's version needs action_space.low and fails on Discrete; here the initial action is 0."""
    N = env.num_envs
    if n_eval_episodes % N:
        raise AssertionError("number of evaluations needs to be divisible by the number of parallel environments")
    if env._scalar:
        raise ValueError("simulate_mdp_vec needs an N-env batch (make(id, num_envs=N))")
    batches = n_eval_episodes // N
    Tmax = int(env.get_attr("Tmax")[0])
    kp = _kernel_policy(model)
    discrete = env.MODEL == 0
    a0 = 0.0 if discrete else float(env.action_space.low[0])
    f64 = torch.float64
    out = []
    saved = env.auto_reset
    env.auto_reset = True
    try:
        for b in range(batches):
            obs = env.reset().reshape(-1).to(f64).clone()
            K_rows = None
            if kp is not None and N % 4 == 0 and not env._np_rng and not env._per_env:
                traj = env.rollout(Tmax, policy=kp[0], param=kp[1], record=True).to(f64)     # [T, 4, N]
                obs_rows = torch.cat([traj[:, 0], env.state.reshape(1, N).to(f64)])               # obs before each step + after the last
                act_rows = torch.cat([torch.full((1, N), a0, dtype=f64, device=env.device), traj[:, 1]])
                rew_rows = torch.cat([torch.zeros((1, N), dtype=f64, device=env.device), traj[:, 2]])
            else:
                obs_rows = torch.empty((Tmax + 1, N), dtype=f64, device=env.device)
                act_rows = torch.full((Tmax + 1, N), a0, dtype=f64, device=env.device)
                rew_rows = torch.zeros((Tmax + 1, N), dtype=f64, device=env.device)
                state = None
                done = torch.zeros(N, dtype=torch.bool, device=env.device)
                o = env.state
                if not hasattr(model, "predict"):       # "random" / ("constant", a) outside the fused kernel
                    if kp is None:
                        raise ValueError("model needs a predict() method (or be 'random' / ('constant', a))")
                    model = _StringPolicy(env, kp)
                if env._per_env:    # fishing-v4 redraws K at every (auto-)reset: a row's population uses the K in force THEN
                    K_rows = torch.empty((Tmax + 1, N), dtype=f64, device=env.device)
                    kr_buf = (env._per_env_buffer(env.dtype), env._per_env_buffer(env.dtype))     # one pair for all Tmax + 1 asks
                for t in range(Tmax):
                    obs_rows[t] = o.reshape(-1).to(f64)
                    if K_rows is not None:
                        K_rows[t] = env._K_view(kr_buf)[:N].to(f64)
                    try:
                        action, state = model.predict(o, state=state, mask=done)
                    except TypeError:                   # a predict(obs) without the SB3 keywords
                        action, state = model.predict(o)
                    a = torch.as_tensor(action, device=env.device).reshape(-1)
                    o, rew, done, _ = env.step(a)
                    act_rows[t + 1] = a.to(f64)
                    rew_rows[t + 1] = rew.to(f64)
                obs_rows[Tmax] = o.reshape(-1).to(f64)
                if K_rows is not None:
                    K_rows[Tmax] = env._K_view(kr_buf)[:N].to(f64)
            K = K_rows if K_rows is not None else float(env.params["K"])
            pop = (obs_rows + 1.0) * K                              # get_fish_population :158-160 (the env's K at that row)
            t_idx = torch.arange(Tmax + 1, device=env.device, dtype=f64).reshape(-1, 1).expand(Tmax + 1, N)
            rep = (b * N + torch.arange(N, device=env.device, dtype=f64)).reshape(1, N).expand(Tmax + 1, N)
            out.append(torch.stack([t_idx, pop, act_rows, rew_rows, rep], dim=-1).reshape(-1, 5).cpu().numpy())
    finally:
        env.auto_reset = saved
    return _table(np.concatenate(out) if out else np.zeros((0, 5)))


def _simulate_mdp_batched(env, model, n_eval_episodes):
    """simulate_mdp's table (shared_env.py:29-54: break on done, at most Tmax rows per rep, quota column) from an
    N-env batch: n_eval_episodes must be a multiple of num_envs; each env of each batch is one rep."""
    if n_eval_episodes % env.num_envs:
        raise AssertionError("number of evaluations needs to be divisible by the number of parallel environments")
    batches = n_eval_episodes // env.num_envs
    kp = _kernel_policy(model)
    out = []
    saved = env.auto_reset
    env.auto_reset = False
    try:
        for b in range(batches):
            env.reset()
            # (a per-env parameter tensor needs the auto-resetting kernel; this table freezes finished envs: step loop)
            if kp is not None and env.num_envs % 4 == 0 and not isinstance(kp[1], torch.Tensor):
                traj = env.rollout(env.Tmax, policy=kp[0], param=kp[1], record=True)
            else:
                traj = torch.zeros((env.Tmax, 4, env.num_envs), dtype=env.dtype, device=env.device)
                obs = env.state
                for t in range(env.Tmax):
                    traj[t, 0] = obs.reshape(-1)
                    action, _ = model.predict(obs)
                    a = torch.as_tensor(action, device=env.device).reshape(-1)
                    traj[t, 1] = a.to(env.dtype)
                    obs, rew, done, _ = env.step(a)
                    traj[t, 2], traj[t, 3] = rew, done.to(env.dtype)
            out.append(_cut_tables(env, traj, b * env.num_envs))
    finally:
        env.auto_reset = saved
    return _table(np.concatenate(out) if out else np.zeros((0, 5)))


def estimate_policyfn(env, model, reps=1, n=50):
    """shared_env.py:82-102: the policy's quota over a grid of n observations."""
    grid = np.linspace(env.observation_space.low, env.observation_space.high, num=n,
                       dtype=env.observation_space.dtype)
    rows = []
    for rep in range(reps):
        for obs in grid:
            action, _ = model.predict(obs)
            if isinstance(action, np.ndarray):
                action = action.reshape(-1)[0]
            if isinstance(action, torch.Tensor):
                action = action.reshape(-1)[0].item()
            pop = env.get_fish_population(obs)
            quota = env.get_quota(action)
            rows.append([float(pop), float(quota), rep])
    arr = np.asarray(rows, dtype=np.float64)
    try:
        from pandas import DataFrame
        return DataFrame(arr, columns=["state", "action", "rep"])
    except ImportError:
        return {"state": arr[:, 0], "action": arr[:, 1], "rep": arr[:, 2]}
