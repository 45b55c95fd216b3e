"""Checkpoint / resume and graph replay of an env (split from envs.py in round 5; the reference has neither -- its env is a few
scalars).  state_dict() / load_state_dict(): everything a rollout needs to resume bit-for-bit -- the per-env streams, the counters
that key the noise, the seed.  enable_graph_replay(): the step counter (and fishing-v4's episode origin) in device memory, so that a
captured hipGraph draws fresh noise on every replay; launch_signature(): what such a capture has frozen.  The mixin reads the env's
buffers and counters and, for fishing-v4, calls into v4_params.V4ParameterModes (_check_v4_state / _adopt_v4_mode / _host_origin)."""
import numpy as np
import torch

from ._capi import MODEL_V11

# state_dict() format.  2: carries `format` and `v4_param_stream` (what draws fishing-v4's (K, r): one Philox2x32-10 block
# per env, fishing_common.h: param_block); _counter holds {step counter, v4 origin step, v4 origin counter}.
# 3: `v11_model_stream` (what draws fishing-v11's per-episode model: one Philox2x32-10 block per env quad, four 16-bit draws,
# fishing_common.h: model_block -- ABI 8; before, a Philox4x32-10 word per env); _counter gains the reset counter as a fourth word.
STATE_FORMAT = 3
V4_PARAM_STREAM = "philox2x32-10/env"
V11_MODEL_STREAM = "philox2x32-10/quad:u16"


class CheckpointAndReplay:
    def _current_step_count(self):
        """How many step() calls this env has made.  In graph-replay mode the device-resident counter is the truth -- a
        replayed hipGraph (GraphedSteps, or a caller's own torch.cuda.CUDAGraph) advances only that -- so the host's
        copy is refreshed from it here (one 8-byte read that waits for the stream: reset(), env.K / env.r, state_dict()
        and leaving the derived mode ask, step() never does)."""
        if self._counter is not None:
            self._step_count = int(self._counter[0].item())
        return self._step_count

    def _current_reset_count(self):
        """How many reset() calls (constructor draw included) this env has made: in graph-replay mode the device word
        counter[3], which fishing_reset_* bumps itself -- also inside a replayed graph, where the host takes no part."""
        if self._counter is not None:
            self._reset_count = int(self._counter[3].item())
        return self._reset_count

    def launch_signature(self):
        """Everything a captured launch has frozen: the parameter struct's source values, fishing-v4's parameter mode and
        the address of every stream.  A hipGraph captured from this env replays correctly only while this value is what it
        was at capture time (GraphedSteps checks it on every replay and re-captures; a caller's own torch.cuda.CUDAGraph
        must do the same): env.Tmax = ..., env.sigma = ..., env.K = ..., seed(), a masked reset() of fishing-v4 and
        load_state_dict() can all change it."""
        ptr = lambda t: (t.data_ptr() if t is not None else 0)  # noqa: E731
        # (in graph-replay mode fishing-v4's episode origin travels in the device-resident counter words: not part of the key)
        return (self._param_key(), self._seed, self._derived,
                tuple(ptr(t) for t in (self._obs, self._t, self._reward, self._done, self._done_bits, self._r_arr, self._K_arr,
                                       self._sigma_arr, self._terminal_obs, self._ep_return, self._partials, self._model_idx,
                                       self._counter, self._stamp)))

    # ------------------------------------------------------------------ checkpoint / resume
    _STATE_TENSORS = ("_obs", "_t", "_reward", "_done", "_r_arr", "_K_arr", "_sigma_arr", "_ep_return", "_partials",
                      "_model_idx", "_counter", "_stamp")
    _STATE_ATTRS = ("_sigma_scalar", "n_actions", "C", "K_mean", "r_mean", "sigma_p")

    def state_dict(self):
        """Everything a rollout needs to resume bit-for-bit: the per-env streams, the counters that
        key the noise, the seed.  (The reference has no checkpointing; its env is a few scalars.)"""
        if self._scalar:
            torch.cuda.current_stream(self.device).synchronize()
        sd = {k: getattr(self, k).clone() for k in self._STATE_TENSORS if getattr(self, k) is not None}
        sd.update(format=STATE_FORMAT, v4_param_stream=V4_PARAM_STREAM, v11_model_stream=V11_MODEL_STREAM,
                  seed=self._seed, step_count=self._current_step_count(), reset_count=self._current_reset_count(),
                  params=dict(self.params), Tmax=self.Tmax, init_state=self.init_state,
                  v4_derived=self._derived, v4_origin=tuple(self._host_origin()), auto_reset=self.auto_reset,
                  attrs={k: getattr(self, k) for k in self._STATE_ATTRS if hasattr(self, k)})
        if self.MODEL == MODEL_V11:
            sd["attrs"].update(models=list(self.models), model_params={k: dict(v) for k, v in self.model_params.items()})
        if self._np_rng:        # rng="numpy": the noise source is NumPy's global stream -- part of the state
            sd["numpy_rng_state"] = np.random.get_state()
        return sd

    def load_state_dict(self, sd, strict=True):
        """Resume from state_dict().  fishing-v4 on the Philox streams redraws (K, r) at every reset from a generator that
        is part of the state's meaning (`v4_param_stream`): a state written under another scheme (round 1: one Philox4x32
        block per env PAIR; no tag at all before format 2) is refused in the derived mode, where the parameters in force
        themselves would come out different.  A stored-array state carries its (K, r) in force, so `strict=False` loads it
        with a warning -- the run continues exactly until the first redraw, which then follows this library's stream.  Envs
        that never use that stream (rng="numpy": the scalar protocol's default) load any fishing-v4 state."""
        # everything that can refuse the state is checked BEFORE the first field changes: a failed load leaves the env as it was
        self._check_v4_state(sd, strict, V4_PARAM_STREAM)
        if self.MODEL == MODEL_V11 and not self._np_rng and sd.get("v11_model_stream") != V11_MODEL_STREAM:
            # the models IN FORCE travel in _model_idx; what the tag decides is every later redraw (reset / auto-reset)
            msg = ("fishing-v11 state was written with model stream %r, this library draws %r: the run would continue with "
                   "other model draws" % (sd.get("v11_model_stream"), V11_MODEL_STREAM))
            if strict:
                raise ValueError(msg + " (strict=False loads the models in force)")
            import warnings
            warnings.warn(msg)
        v4_arrays = self._per_env and not sd.get("v4_derived", False)
        for k in self._STATE_TENSORS:
            if k in sd and getattr(self, k) is None and k not in ("_counter", "_stamp") and not (k in ("_r_arr", "_K_arr") and v4_arrays):
                raise ValueError("state has %s but this env was built without it" % k)
            # sizes too: a state of another batch size must not get as far as the first copy_.  (return_partials grew with
            # ABI 5 for batches beyond 2^22 envs: an older, shorter buffer loads into the first slots -- the record is
            # the sum over slots -- a longer one cannot.)
            if k in sd and getattr(self, k) is not None and k != "_counter":
                have, got = getattr(self, k).numel(), sd[k].numel()
                if got != have and not (k == "_partials" and got < have):
                    raise ValueError("state's %s has %d elements, this env's %d" % (k, got, have))
        if self._host_mapped:
            torch.cuda.current_stream(self.device).synchronize()
        self._adopt_v4_mode(sd)                 # fishing-v4: same parameter mode as the saved env
        for k in self._STATE_TENSORS:
            if k in sd:
                if getattr(self, k) is None and k == "_counter":
                    self.enable_graph_replay()
                if k == "_counter":         # (format 1 kept the step counter alone, format 2 three words; the others follow below)
                    self._counter[:sd[k].numel()].copy_(sd[k])
                elif k == "_partials" and sd[k].numel() < self._partials.numel():
                    self._partials.zero_()
                    self._partials[:sd[k].numel()].copy_(sd[k])
                else:
                    getattr(self, k).copy_(sd[k])
        self._seed, self._step_count, self._reset_count = sd["seed"], sd["step_count"], sd["reset_count"]
        if self._counter is not None and "_counter" not in sd:
            # a state taken from a host-counter env, loaded into an env in graph-replay mode: the device word is this env's
            # step count from here on (_current_step_count() reads it back), so it must not keep the value it had
            self._counter[0].fill_(int(sd["step_count"]))
        if self._counter is not None:       # (the reset counter's device word: absent from a format-2 _counter, or no _counter at all)
            self._counter[3].fill_(int(sd["reset_count"]))
        self.params.update(sd["params"])
        self.Tmax, self.init_state = sd["Tmax"], sd["init_state"]
        self.auto_reset = sd.get("auto_reset", self.auto_reset)
        for k, v in sd.get("attrs", {}).items():          # the scalar attributes FishingParams is built from
            setattr(self, k, [*v] if k == "models" else ({m: dict(d) for m, d in v.items()} if k == "model_params" else v))
        if self._sigma_arr is None and "_sigma_scalar" not in sd.get("attrs", {}):
            self._sigma_scalar = float(self.params["sigma"])
        if self._np_rng and "numpy_rng_state" in sd:
            np.random.set_state(sd["numpy_rng_state"])
        if self._per_env:
            self._set_origin(*self._origin)
        self._publish_scalar_state()
        return self

    def enable_graph_replay(self):
        """Keep the step counter -- and the reset counter -- in device memory from now on.  step() / step_many() / rollout()
        then launch with frozen arguments plus a one-thread counter bump, so a hipGraph that
        captured them (gym_fishing_amd.graphs.GraphedSteps, or a caller's own torch.cuda.CUDAGraph) draws
        fresh noise on every replay; reset() draws fishing-v4's (K, r) / fishing-v11's models with the device's reset counter
        and bumps it itself, so a captured [reset(), K steps] collection loop starts every replayed episode from a fresh draw.
        Same streams as the host-counter mode: an eager env making the same calls lands on the same bits.
        What a capture freezes besides the counter: every scalar of the parameter struct, fishing-v4's parameter mode
        (derived / stored arrays) and every stream's address -- launch_signature().  GraphedSteps re-captures when that
        changes; a caller replaying a torch.cuda.CUDAGraph of its own must compare launch_signature() itself.  The env
        never frees a stream a capture may still reference (fishing-v4's r / K arrays live as long as the env)."""
        if self._counter is None:
            # {step counter, v4 origin step, v4 origin counter}: a captured launch freezes FishingParams, and with them the
            # origin that the derived fishing-v4 parameters date episodes from -- so in this mode the kernels read the
            # origin from these words (include/fishing_hip.h: FishingBuffers.counter), which reset() rewrites.  fishing-v4
            # stays in the derived mode under graph replay (ABI 4; round 2 fell back to r / K arrays for good).
            # ... and the reset counter as a fourth word, read and bumped by fishing_reset_* itself
            # (FISHING_FLAG_RESET_COUNTER_ON_DEVICE, ABI 9): a captured reset() draws fresh parameters at every replay.
            self._counter = torch.tensor([self._step_count, self._origin[0], self._origin[1], self._reset_count],
                                         dtype=torch.int64, device=self.device)
            self._cbuf = None
        return self
