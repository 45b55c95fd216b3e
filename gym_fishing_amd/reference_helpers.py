"""The reference's helper methods on the env (base_fishing_env.py:100-164; split from envs.py in round 5): get_quota / get_action /
get_fish_population / get_state, harvest_draw, population_draw (what BMSY() drives, models/policies.py:59-63), the per-env BMSY sweep,
and simulate / policyfn / plot / plot_policy.  Elementwise on tensors for the N-env protocol, the reference's scalars for one env."""
import numpy as np
import torch

from . import _capi
from ._capi import MODEL_V0, MODEL_V1, MODEL_V2, MODEL_V4, MODEL_V10, MODEL_V11
from .spaces import is_discrete


class ReferenceHelpers:
    # ------------------------------------------------------------------ helpers (base_fishing_env.py:135-164)
    def _K_for_math(self):
        if self._per_env:
            return self._K_view()
        return self.params["K"]

    def get_quota(self, action):
        """base_fishing_env.py:135-147 (elementwise on tensors in vec mode)."""
        K = self._K_for_math()
        if is_discrete(self.action_space):
            if isinstance(action, torch.Tensor):
                return (action.to(torch.float64) / self.n_actions) * K
            return (action / self.n_actions) * K
        # the action passes through the float32 action Box (np.clip against its float32 bounds),
        # then the quota is formed in float64
        if isinstance(action, torch.Tensor):
            return (action.to(torch.float32).to(torch.float64).clamp(-1.0, 1.0).reshape(-1) + 1.0) * K
        a = np.clip(np.asarray(action, dtype=np.float32).astype(np.float64), -1.0, 1.0).reshape(-1)[0]
        return (a + 1) * K

    def get_action(self, quota):
        """base_fishing_env.py:149-156."""
        K = self._K_for_math()
        if is_discrete(self.action_space):
            if isinstance(quota, torch.Tensor):
                return torch.round(quota * self.n_actions / K).to(torch.int64)
            return round(quota * self.n_actions / K)
        return quota / K - 1

    def get_fish_population(self, state):
        """base_fishing_env.py:158-160."""
        K = self._K_for_math()
        if isinstance(state, torch.Tensor):
            pop = (state.to(torch.float64).reshape(-1) + 1.0) * K
            return pop
        # (state[0] + 1) * K: a (1,) observation gives a scalar, the reference's VecEnv idiom
        # get_fish_population((obs_i,)) (shared_env.py:17-19) an array of shape (1,)
        s0 = state if np.ndim(state) == 0 else state[0]
        pop = (np.asarray(s0, dtype=np.float64) + 1) * K
        if self._scalar:
            self.fish_population = pop
        return pop

    def get_state(self, fish_population):
        """base_fishing_env.py:162-164."""
        K = self._K_for_math()
        if isinstance(fish_population, torch.Tensor):
            return (fish_population / K - 1.0).reshape(-1, 1)
        return np.array([fish_population / K - 1])

    def harvest_draw(self, quota, x=None):
        """base_fishing_env.py:112-119: harvest = min(population, quota); population = max(population - harvest, 0.0).
        One env: on self.fish_population, which it updates, like the reference (self.harvest too).  Tensors: `x` holds
        the populations (default: the batch's current ones), elementwise with Python's min / max operand order (a NaN
        quota leaves the population's side of min(), as in the step kernels); returns (harvest, population left)."""
        if isinstance(quota, torch.Tensor) or isinstance(x, torch.Tensor) or not self._scalar:
            if x is None:
                x = self.get_fish_population(self._obs_view)
            xt = torch.as_tensor(x, device=self.device)
            q = torch.as_tensor(quota, device=self.device).to(xt.dtype).expand_as(xt) if not isinstance(quota, torch.Tensor) \
                else quota.to(device=self.device, dtype=xt.dtype).reshape(xt.shape)
            h = torch.where(q < xt, q, xt)                       # min(x, q): q only where q < x
            d = xt - h
            return h, torch.where(d.new_zeros(()) > d, d.new_zeros(()), d)     # max(d, 0.0): 0.0 only where 0.0 > d
        pop = self.fish_population if x is None else x
        self.harvest = min(pop, quota)
        self.fish_population = max(pop - self.harvest, 0.0)
        return self.harvest

    def population_draw(self, x=None, noise=None, sigma=None, dtype=None, r=None, K=None, model_idx=None):
        """base_fishing_env.py:121-133 (v2: fishing_tipping_env.py:24-35) over an array of
        populations -- the call BMSY() makes (models/policies.py:59-63).  `x` None uses
        self.fish_population like the reference's zero-argument form (scalar protocol).
        `dtype` picks the arithmetic (default: the env's layout); `sigma`, `r`, `K` override the
        scalar parameters for this call only (`r` / `K` as tensors: one value per population).
        fishing-v11 (growth_models.py:190-194: the growth function in force, with ITS parameter set).  One env: its
        model.  N envs: `x` holds one population per env and env i grows under model_idx[i], the function in force
        there -- or pass `model_idx` (int32, one FISHING_KIND per element of `x`) to choose per element, e.g. one sweep
        per growth function in a single launch (policies.BMSY does)."""
        use_attr = x is None
        if use_attr:
            x = self.fish_population
        dtype = self.dtype if dtype is None else dtype
        xt = torch.as_tensor(x).to(device=self.device, dtype=dtype).reshape(-1).contiguous()
        zt = None
        if noise is not None:
            zt = torch.as_tensor(noise).to(device=self.device, dtype=dtype).reshape(-1).contiguous()
        elif self._np_rng:
            # the reference draws here whatever sigma is: one np.random.normal(0, 1) for the logistic / tipping
            # models (base_fishing_env.py:130), np.random.lognormal(mu, sigma) -- one normal per ELEMENT of mu --
            # for the zoo (growth_models.py:217-261); consume the global stream the same way
            zoo = self.MODEL not in (MODEL_V0, MODEL_V1, MODEL_V2, MODEL_V4)
            z = np.random.normal(0, 1, xt.numel()) if (zoo and xt.numel() > 1) else np.full(xt.numel(), np.random.normal(0, 1))
            zt = torch.as_tensor(z).to(device=self.device, dtype=dtype)
        out = torch.empty_like(xt)
        cp = self._c_params()
        if self.MODEL == MODEL_V10 and self._scalar and r is None:
            # growth_models.py:151: every population_draw() call -- BMSY()'s and msy()'s sweeps included -- first
            # moves r by alpha, and keeps the moved value
            r = float(self._r_arr[0]) + float(self.params.get("alpha", 0.0))
            self._r_arr.fill_(r)
        # `r` / `K` as tensors (one value per population): element i under ITS parameters -- N fishing-v4 envs, each with the
        # pair it drew (policies.msy); scalars override the struct's for this call
        r_arr = K_arr = None
        if isinstance(r, torch.Tensor) or isinstance(K, torch.Tensor):
            if self.MODEL not in (MODEL_V0, MODEL_V1, MODEL_V2, MODEL_V4):
                raise ValueError("per-population r / K are the logistic / tipping models'")
            if isinstance(r, torch.Tensor):
                r_arr, r = r.to(device=self.device, dtype=dtype).reshape(-1).contiguous(), None
            if isinstance(K, torch.Tensor):
                K_arr, K = K.to(device=self.device, dtype=dtype).reshape(-1).contiguous(), None
            if any(a is not None and a.numel() != xt.numel() for a in (r_arr, K_arr)):
                raise ValueError("r / K tensors need one value per population (%d)" % xt.numel())
        if sigma is not None or r is not None or K is not None:     # never edit the cached struct step() uses
            cp = _capi.FishingParams.from_buffer_copy(cp)
            if sigma is not None:
                cp.sigma = float(sigma)
            if r is not None:
                cp.r = float(r)
            if K is not None:
                cp.K = float(K)
        kinds = None
        if model_idx is not None and self.MODEL != MODEL_V11:
            raise ValueError("model_idx selects fishing-v11's growth function per element; %s has one" % type(self).__name__)
        if self.MODEL == MODEL_V11:
            # growth_models.py:190-194: the growth function currently in force, with ITS parameter set
            if model_idx is not None:
                kinds = torch.as_tensor(model_idx).to(device=self.device, dtype=torch.int32).reshape(-1).contiguous()
                if kinds.numel() != xt.numel():
                    raise ValueError("model_idx needs one entry per population (%d), got %d" % (xt.numel(), kinds.numel()))
            elif self._scalar:
                kinds = self._model_idx[:1].expand(xt.numel()).contiguous()      # one env, one model in force
            elif xt.numel() == self.num_envs:
                kinds = self._model_idx                                          # env i under the model in force there
            else:
                raise ValueError("fishing-v11 with num_envs=%d: pass one population per env (each grows under its env's "
                                 "model in force) or model_idx= with one growth-function kind per population"
                                 % self.num_envs)
        fn = getattr(self._lib, "fishing_population_draw_" + ("f32" if dtype == torch.float32 else "f64"))
        with torch.cuda.device(self.device):
            rc = fn(cp, xt.numel(), xt.data_ptr(), zt.data_ptr() if zt is not None else None,
                    kinds.data_ptr() if kinds is not None else None, r_arr.data_ptr() if r_arr is not None else None,
                    K_arr.data_ptr() if K_arr is not None else None, out.data_ptr(), self._stream())
        _capi.check(rc, "fishing_population_draw")
        if isinstance(x, torch.Tensor):
            return out.reshape(x.shape)
        res = out.cpu().numpy().astype(np.float64)
        res = res.reshape(np.shape(x)) if np.ndim(x) else float(res[0])
        if use_attr:
            self.fish_population = res
        return res

    def bmsy_sweep(self, states, K, r, dtype=None):
        """BMSY()'s sweep (models/policies.py:51-67) once per env, each under ITS (K, r) tensors: S[i] = the population
        (states[j] + 1) * K[i] with the largest noise-free one-step growth.  fishing-v0/v1/v2/v4."""
        dtype = self.dtype if dtype is None else dtype
        st = torch.as_tensor(states).to(device=self.device, dtype=dtype).reshape(-1).contiguous()
        Kt, rt = (torch.as_tensor(v).to(device=self.device, dtype=dtype).reshape(-1).contiguous() for v in (K, r))
        if Kt.numel() != rt.numel():
            raise ValueError("K and r need one value per env each")
        out = torch.empty_like(Kt)
        fn = getattr(self._lib, "fishing_bmsy_sweep_" + ("f32" if dtype == torch.float32 else "f64"))
        with torch.cuda.device(self.device):
            rc = fn(self._c_params(), Kt.numel(), Kt.data_ptr(), rt.data_ptr(), st.data_ptr(), st.numel(), out.data_ptr(), self._stream())
        _capi.check(rc, "fishing_bmsy_sweep")
        return out

    # the reference exposes its helpers as methods (base_fishing_env.py:100-110)
    def simulate(self, model, reps=1):
        from .rollout import simulate_mdp
        return simulate_mdp(self, model, reps)

    def policyfn(self, model, reps=1):
        from .rollout import estimate_policyfn
        return estimate_policyfn(self, model, reps)

    def plot(self, df, output="results.png"):
        from .plotting import plot_mdp
        return plot_mdp(df, output)

    def plot_policy(self, df, output="results.png"):
        from .plotting import plot_policyfn
        return plot_policyfn(df, output)
