"""fishing-v4's parameter modes (fishing_model_error.py:37-48: K, r ~ N(mean, sigma_p) redrawn per env at every reset).

An N-env fishing-v4 batch on the Philox streams keeps NO r / K arrays: an env's (K, r) are a function of where its episode
began, and every kernel re-derives them from the block that drew them (csrc/fishing_common.h: derive_model_error).  This
mixin is the host side of that mode machine -- which mode the env is in, what dates an episode, when arrays come back --
kept apart from the env class's protocol / buffer plumbing (envs.py):

    derived            the year counter dates every episode from the last reset() of ALL envs (the *origin*: step count +
                       reset counter, in FishingParams -- or, in graph-replay mode, in the device words counter[1..2], which
                       are then the truth: reset() moves them without a host read, a replayed graph moves them alone)
    derived + stamps   after a masked reset(): the masked envs carry a per-env origin stamp (R 4 + W 4 bytes per env-step)
                       until the next reset of all envs
    stored arrays      once something makes the parameters underivable -- env.K = ... / env.r = ..., seed(), an outside write
                       to years_passed, a rollout without auto-reset, rng="numpy", the scalar protocol, compact year counters --
                       (K, r) live in a pair of arrays allocated once and never freed (a captured hipGraph may still write
                       through those addresses); the next reset() of all envs returns to the derived mode.

The mixin owns: _derived_capable, _derived, _origin, _K_store / _r_store, _stamp / _stamp_store; it reads the env's buffers
(_t, _counter, _K_arr, _r_arr), _c_params(), _stream() and counters."""
import torch

from . import _capi


class V4ParameterModes:
    def _init_v4_modes(self, derived_params):
        self._derived_capable = (self._per_env and not self._np_rng and not self._scalar and not self.compact
                                 and (derived_params is None or bool(derived_params)))
        if derived_params and not self._derived_capable:
            raise ValueError("derived_params=True needs fishing-v4 with num_envs, rng='philox' and the int32 year counter")
        self._derived = self._derived_capable
        self._origin = (0, 0)            # (step count, reset counter) of the last reset() of all envs
        self._K_store = self._r_store = None      # (see _param_store)
        # per-env episode origins of envs reset one by one (FishingBuffers.v4_stamp; allocated by the first masked reset(),
        # dropped from the launches again by the next reset of every env, never freed)
        self._stamp = self._stamp_store = None
        if self._per_env and not self._derived:
            self._K_arr, self._r_arr = self._param_store()
            self._r_arr.fill_(float(self.params["r"]))
            self._K_arr.fill_(float(self.params["K"]))

    # ------------------------------------------------------------------ storage that outlives every mode
    def _param_store(self):
        """The (K, r) arrays of a fishing-v4 env, allocated once and kept for the env's lifetime: a launch captured in a
        hipGraph while the env ran on stored arrays keeps reading -- and, on every auto-reset, WRITING -- these addresses,
        so they must never go back to the allocator while the env lives, whatever mode it is in by then."""
        if self._K_store is None:
            self._K_store, self._r_store = self._per_env_buffer(self.dtype), self._per_env_buffer(self.dtype)
        return self._K_store, self._r_store

    def _stamp_buffer(self):
        """fishing-v4's origin stamps (int32, zeroed): one allocation for the env's lifetime, like _param_store."""
        if self._stamp_store is None:
            self._stamp_store = self._per_env_buffer(torch.int32)
        else:
            self._stamp_store.zero_()
        return self._stamp_store

    # ------------------------------------------------------------------ the origin
    def _host_origin(self):
        """The origin as host integers.  In graph-replay mode the device words are the truth -- reset() moves the origin there
        without a host read, and a replayed graph that contains a reset() moves it without the host taking part at all -- so
        it is read back here, every time (state_dict(), env.K / env.r: calls that wait for the stream anyway)."""
        if self._counter is not None:
            words = self._counter.tolist()
            self._origin = (int(words[1]), int(words[2]))
            self._step_count, self._reset_count = int(words[0]), int(words[3])
        return self._origin

    def _set_origin(self, step_count, reset_count):
        """(step count, reset counter) of the reset() of ALL envs that dates every running episode; mirrored into the
        device-resident counter words in graph-replay mode (two fills on the current stream)."""
        self._origin = (int(step_count), int(reset_count))
        if self._counter is not None:
            self._counter[1].fill_(self._origin[0])
            self._counter[2].fill_(self._origin[1])

    # ------------------------------------------------------------------ transitions
    def _enter_derived_mode_at_full_reset(self):
        """reset() of ALL envs: its counters date every episode from here on; arrays and stamps leave the launches."""
        self._derived, self._K_arr, self._r_arr, self._cbuf = True, None, None, None
        if self._counter is None:
            self._set_origin(self._step_count, self._reset_count)
        # (graph-replay mode: fishing_reset_* itself moves the origin, device word to device word behind the reset kernel --
        # counter[1] = the step counter, counter[2] = the reset counter it drew with (FISHING_FLAG_RESET_COUNTER_ON_DEVICE) -- so
        # reset() neither waits for the GPU nor breaks a caller's stream capture, and a REPLAYED reset() dates the episodes from
        # the replay's own counters.  The host's copy of the origin is read back on demand: _host_origin.)

    def _begin_masked_reset(self):
        """Envs reset at different times: in the derived mode each masked env's episode origin goes into its stamp (R 4 + W 4
        per env-step from here on, until the next reset of every env) -- no r / K arrays."""
        if self._derived and self._stamp is None:
            self._stamp, self._cbuf = self._stamp_buffer(), None

    def _leave_derived_mode(self):
        """Store the parameters in force and continue with r / K arrays (until the next full reset())."""
        if self._derived:
            self._K_arr, self._r_arr = self._derive_params(self._param_store())
            self._derived = False
            self._stamp = None              # (origin stamps belong to the derived mode)
            self._cbuf = None

    # ------------------------------------------------------------------ what env.K / env.r show
    def _derive_params(self, out=None):
        """(K, r) tensors of a fishing-v4 env in the derived mode, materialised by fishing_v4_params_* (into the pair
        `out` when given: callers that ask every step reuse one pair instead of allocating two streams per call)."""
        K, r = out if out is not None else (self._per_env_buffer(self.dtype), self._per_env_buffer(self.dtype))
        cp = self._c_params()
        if self._counter is not None:
            # graph-replay mode: the episode origin lives in the device words (reset() moves it there without telling the
            # host) and fishing_v4_params_* takes it from the struct -- read it back for this call (which waits for the
            # stream anyway: it needs the step count)
            cp = _capi.FishingParams.from_buffer_copy(cp)
            cp.v4_origin_step, cp.v4_origin_counter = self._host_origin()
        with torch.cuda.device(self.device):
            rc = getattr(self._lib, "fishing_v4_params_" + self._suffix)(
                cp, self.num_envs, self.env_offset, self._t.data_ptr(),
                self._stamp.data_ptr() if self._stamp is not None else None, K.data_ptr(), r.data_ptr(),
                self._seed, self._current_step_count(), self._stream())
        _capi.check(rc, "fishing_v4_params")
        return K, r

    def _K_view(self, out=None):
        if self._derived:
            return self._derive_params(out)[0]
        return float(self._K_arr[0]) if self._scalar else self._K_arr

    def _r_view(self):
        if self._derived:
            return self._derive_params()[1]
        return float(self._r_arr[0]) if self._scalar else self._r_arr

    # ------------------------------------------------------------------ checkpoints
    def _check_v4_state(self, sd, strict, stream_tag):
        """Everything about a fishing-v4 state that can refuse it, before the first field changes (load_state_dict)."""
        if self._per_env and not self._np_rng and sd.get("v4_param_stream") != stream_tag:
            if strict or sd.get("v4_derived", False):
                raise ValueError("fishing-v4 state was written with parameter stream %r, this library draws %r: it cannot "
                                 "resume bit-for-bit%s" % (sd.get("v4_param_stream"), stream_tag,
                                                           "" if sd.get("v4_derived", False) else " (strict=False loads the stored (K, r))"))
            import warnings
            warnings.warn("fishing-v4 state written with parameter stream %r: the (K, r) in force are loaded, redraws will "
                          "follow %r" % (sd.get("v4_param_stream"), stream_tag))
        if self._per_env and sd.get("v4_derived", False) and not self._derived_capable:
            raise ValueError("state was saved in the derived-parameter mode, which this env cannot run")
        if "_stamp" in sd:      # origin stamps exist only in fishing-v4's derived mode, one per env
            if not (self._per_env and sd.get("v4_derived", False)):
                raise ValueError("state has _stamp (fishing-v4 origin stamps) but was not saved in the derived-parameter mode")
            if sd["_stamp"].numel() != self.num_envs:
                raise ValueError("state's _stamp has %d elements, this env's %d" % (sd["_stamp"].numel(), self.num_envs))

    def _adopt_v4_mode(self, sd):
        """Put the env into the parameter mode the state was saved in (load_state_dict, before the streams are copied)."""
        if not self._per_env:
            return
        if sd.get("v4_derived", False):
            self._derived, self._K_arr, self._r_arr = True, None, None
            self._stamp = self._stamp_buffer() if "_stamp" in sd else None
        elif self._derived:
            self._derived = False
            self._K_arr, self._r_arr = self._param_store()
        if not self._derived:
            self._stamp = None
        self._origin = tuple(sd.get("v4_origin", (0, 0)))
        self._cbuf = None
