"""hipGraph capture of env steps (torch.cuda.CUDAGraph is the capture/replay plumbing).

The kernels are launched on torch's current stream, so they are captured like any torch
op.  What needs care is the noise key: a captured launch freezes its arguments, so the
step counter moves to device memory first (BaseFishingEnv.enable_graph_replay); every
replay then advances it with a captured one-thread kernel.

    g = GraphedSteps(env, actions)      # actions: static [R, N] (or [N]) device tensor
    for _ in range(iters):
        actions.copy_(policy(...))      # write new actions into the static buffer
        obs, reward, done, info = g.replay()
"""
import torch


class GraphedSteps:
    def __init__(self, env, actions, n_steps=None, warmup=1, record=False):
        """`record=True` also captures the reduction of the episodic-return record behind the steps (this rank's record:
        no collective inside the graph); `self.record` is then the 4-double device tensor every replay rewrites.
        A capture freezes the launch arguments: the parameter struct's scalars, fishing-v4's parameter mode and every
        stream's address (env.launch_signature()).  replay() compares that signature with the one captured and captures
        again when it moved -- env.Tmax = ..., env.sigma = ..., env.K = ..., seed(), a masked reset() of fishing-v4,
        load_state_dict() -- so a replay never runs a launch the env has since outgrown (`self.recaptures` counts)."""
        if env._scalar:
            raise ValueError("graph capture is for the N-env tensor protocol")
        self.env = env.enable_graph_replay()
        self.actions = actions
        many = actions.dim() == 2
        self.n_steps = (actions.shape[0] if n_steps is None else int(n_steps)) if many else 1
        self._step = (lambda: env.step_many(actions, self.n_steps)) if many else (lambda: env.step(actions))
        self._with_record = bool(record)
        self._warmup = int(warmup)
        self.record = None
        self.recaptures = -1
        self._capture()

    def _run(self):
        out = self._step()
        if self._with_record:
            self.record = self.env.episode_record(all_reduce=False)
        return out

    def _capture(self):
        env = self.env
        # warm up on a side stream (torch's capture rule), then restore the counter so the
        # captured sequence continues where the caller left off
        start = env._counter.clone()
        side = torch.cuda.Stream(device=env.device)
        side.wait_stream(torch.cuda.current_stream(env.device))
        with torch.cuda.stream(side):
            snap = (env._obs.clone(), env._t.clone())
            state = (env._ep_return, env._partials, env._r_arr, env._K_arr, env._model_idx, env._stamp)
            extra = [t.clone() if t is not None else None for t in state]
            for _ in range(self._warmup):
                self._run()
            env._obs.copy_(snap[0])
            env._t.copy_(snap[1])
            for t, c in zip(state, extra):
                if t is not None:
                    t.copy_(c)
            env._counter.copy_(start)
        torch.cuda.current_stream(env.device).wait_stream(side)
        env._step_count -= self._warmup * self.n_steps
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._run()
        # capture does not execute: the env state and counter are still at `start`
        env._step_count -= self.n_steps
        self._signature = env.launch_signature()
        self.recaptures += 1

    def replay(self):
        if self.env.launch_signature() != self._signature:
            self._capture()
        self.graph.replay()
        self.env._step_count += self.n_steps
        return self.out
