"""CPU ORACLE (NumPy) for the gym_fishing hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement of what the reference environment
computes in ``step()`` / ``reset()`` for fishing-v0/v1/v2/v4, vectorised over
envs, with every operation individually rounded in the order the reference
evaluates it.  It is the *checker* for the HIP kernels.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
the product package ``gym_fishing_amd`` never does (tests/test_host_logic.py
greps for that).

Parity status: PINNED.  tests/test_oracle_golden.py checks it bit-for-bit
against tests/golden/reference_trajectories.npz, which tests/golden/make_golden.py
captured from the unmodified reference imported in the build container.

Reference citations are relative to /root/reference/gym_fishing/envs/ :
  base_fishing_env.py:60-81    step()               -> step()
  base_fishing_env.py:135-147  get_quota()          -> quota_from_action()
  base_fishing_env.py:158-160  get_fish_population  -> x = (obs + 1) * K
  base_fishing_env.py:112-119  harvest_draw()       -> h, x
  base_fishing_env.py:121-133  population_draw()    -> logistic growth
  fishing_tipping_env.py:24-35 population_draw()    -> tipping-point growth (v2)
  base_fishing_env.py:162-164  get_state()          -> obs' = x / K - 1
  base_fishing_env.py:83-91    reset()              -> reset_obs()
  fishing_model_error.py:37-48 ctor / reset() (v4)  -> draw_model_error_params(), reset_obs()

The in-kernel noise generator (Philox4x32-10 + Box-Muller) has no counterpart in
the reference (which uses the global MT19937 stream, base_fishing_env.py:130); its
restatement here follows the published Philox algorithm (Salmon et al., SC'11,
"Parallel random numbers: as easy as 1, 2, 3") and is pinned by the Random123
known-answer vectors in tests/test_oracle_golden.py.
"""
import numpy as np

# model ids shared with include/fishing_hip.h
MODEL_V0 = 0  # logistic, Discrete(n_actions)           fishing_env.py:7-24
MODEL_V1 = 1  # logistic, continuous                    fishing_cts_env.py:4-12
MODEL_V2 = 2  # tipping point, continuous               fishing_tipping_env.py:6-35
MODEL_V4 = 4  # logistic, per-episode K,r uncertainty   fishing_model_error.py:6-48

# growth-model zoo (growth_models.py:6-270): lognormal noise, x' = max(0, exp(mu(x) + sigma z))
MODEL_V5 = 5    # Allen            growth_models.py:6-25,   allen() :208-217
MODEL_V6 = 6    # Beverton-Holt    :28-40,  beverton_holt() :220-226
MODEL_V7 = 7    # May              :75-108, may() :229-242
MODEL_V8 = 8    # Myers            :43-70,  myers() :247-255
MODEL_V9 = 9    # Ricker           :111-123, ricker() :258-261
MODEL_V10 = 10  # NonStationary    :126-154 (Beverton-Holt with r += alpha every draw)
MODEL_V11 = 11  # ModelUncertainty :157-204 (one of the five per episode)

MODEL_OF_ID = {"fishing-v0": MODEL_V0, "fishing-v1": MODEL_V1,
               "fishing-v2": MODEL_V2, "fishing-v4": MODEL_V4,
               "fishing-v5": MODEL_V5, "fishing-v6": MODEL_V6, "fishing-v7": MODEL_V7,
               "fishing-v8": MODEL_V8, "fishing-v9": MODEL_V9, "fishing-v10": MODEL_V10,
               "fishing-v11": MODEL_V11}

# kinds of growth function, in the order of the reference's default `models` list
# (growth_models.py:160): the index is what fishing-v11 stores per env
KIND_ALLEN, KIND_BH, KIND_MYERS, KIND_MAY, KIND_RICKER = 0, 1, 2, 3, 4
KIND_OF_MODEL = {MODEL_V5: KIND_ALLEN, MODEL_V6: KIND_BH, MODEL_V7: KIND_MAY, MODEL_V8: KIND_MYERS,
                 MODEL_V9: KIND_RICKER, MODEL_V10: KIND_BH}
# fishing-v11's per-model parameter table (growth_models.py:161-186)
V11_TABLE = [
    {"r": 0.3, "K": 1.0, "sigma": 0.0, "C": 0.5},                                        # allen
    {"r": 0.3, "K": 1.0, "sigma": 0.0},                                                  # beverton_holt
    {"r": 1.0, "K": 1.0, "M": 1.0, "theta": 3.0, "sigma": 0.0},                          # myers
    {"r": 0.7, "K": 1.5, "M": 1.5, "q": 3.0, "b": 0.15, "sigma": 0.0, "a": 0.2},         # may
    {"r": 0.3, "K": 1.0, "sigma": 0.0},                                                  # ricker
]

# RNG stream tags (top byte of Philox counter word 1)
STREAM_NOISE = 0      # per-step process noise z
STREAM_AUTORESET = 1  # v4 (K, r) redraw when step() auto-resets a finished env
STREAM_RESET = 2      # v4 (K, r) redraw in an explicit reset()
STREAM_POLICY = 3     # random-policy actions of the fused rollout


def quota_from_action(model, action, K, n_actions, dtype=np.float64):
    """base_fishing_env.py:135-147.  Discrete: (a / n_actions) * K with true
    division (:140).  Continuous: clip to the Box [-1, 1] (:143-145), then
    (a + 1) * K (:146).  The float32 action is widened to ``dtype`` first, which
    is what the reference's pinned NumPy 1.19 does."""
    dt = np.dtype(dtype).type
    if model == MODEL_V0:
        a = np.asarray(action).astype(dtype)       # exact for |a| < 2**24 (f32) / 2**53 (f64)
        return (a / dt(n_actions)) * K
    a = np.asarray(action, dtype=np.float32).astype(dtype)
    a = np.clip(a, dt(-1.0), dt(1.0))
    return (a + dt(1.0)) * K


def step(model, obs, t, action, z, r, K, sigma, C=0.5, Tmax=100, n_actions=100,
         dtype=np.float64):
    """One reference step() for every env.  Returns (obs', reward, done, t', x).

    obs, z : arrays of ``dtype``;  t : int array;  r, K, sigma : scalars or arrays
    of ``dtype`` (arrays for fishing-v4).  ``done`` is uint8.  No auto-reset here:
    the reference has none (base_fishing_env.py:60-81); see auto_reset()."""
    dt = np.dtype(dtype).type
    obs = np.asarray(obs, dtype=dtype)
    z = np.asarray(z, dtype=dtype)
    r = np.asarray(r, dtype=dtype)
    K = np.asarray(K, dtype=dtype)
    sigma = np.asarray(sigma, dtype=dtype)
    one, zero = dt(1.0), dt(0.0)
    with np.errstate(all="ignore"):
        quota = quota_from_action(model, action, K, n_actions, dtype)
        x = (obs + one) * K                                   # :159
        h = np.where(quota < x, quota, x)                     # :117  min(x, quota)
        d = x - h
        x = np.where(zero > d, zero, d)                       # :118  max(x - h, 0.0)
        if model == MODEL_V2:                                 # fishing_tipping_env.py:25-34
            e = ((r * (one - (x / K))) * (x - dt(C))) + ((x * sigma) * z)
            g = x * np.exp(e)
        else:                                                 # :125-131
            g = (x + ((r * x) * (one - (x / K)))) + ((x * sigma) * z)
        x = np.maximum(g, zero)                               # NaN-propagating
        obs_next = x / K - one                                # :163
        reward = np.where(zero > h, zero, h)                  # :74   max(h, 0.0)
        t_next = np.asarray(t, dtype=np.int32) + np.int32(1)  # :75
        done = (t_next > np.int32(Tmax)) | (x <= zero)        # :76-79
    return (obs_next.astype(dtype), reward.astype(dtype), done.astype(np.uint8),
            t_next.astype(np.int32), x.astype(dtype))


def _libm_exp(v):
    """exp() of the C library, element by element.  np.random.lognormal (legacy RandomState)
    is exp(loc + scale * gauss) evaluated with libm's scalar exp(), which differs from NumPy's
    SIMD np.exp by 1 ulp on ~2 % of inputs; float64 parity needs the libm one."""
    import math
    v = np.asarray(v)
    if v.dtype != np.float64:
        return np.exp(v)                     # float32 layout: tolerance-based anyway
    flat = v.reshape(-1)
    out = np.empty_like(flat)
    for i, e in enumerate(flat):
        try:
            out[i] = math.exp(e)
        except OverflowError:
            out[i] = np.inf
    return out.reshape(v.shape)


def zoo_population_draw(kind, x, z, P, dtype=np.float64, simd_exp=False):
    """The five growth functions of growth_models.py:208-261, each followed by
    np.maximum(0, np.random.lognormal(mu, sigma)) = max(0, exp(mu + sigma z)).
    `P`: dict of scalars / arrays (r, K, sigma and, per kind, C | M, theta | M, q, b, a).
    `simd_exp`: np.exp instead of libm's scalar exp (1 ulp apart on ~2 % of inputs) -- for the tolerance-based
    comparisons of millions of envs, where the element-by-element libm loop would take seconds per step."""
    dt = np.dtype(dtype).type
    g = lambda k: np.asarray(P[k], dtype=dtype)   # noqa: E731
    x = np.asarray(x, dtype=dtype)
    z = np.asarray(z, dtype=dtype)
    one, zero = dt(1.0), dt(0.0)
    with np.errstate(all="ignore"):
        if kind == KIND_ALLEN:                                  # :208-217
            mu = np.log(x) + g("r") * (one - x / g("K")) * (one - g("C")) / g("K")
        elif kind == KIND_BH:                                   # :220-226
            x = np.clip(x, zero, dt(np.inf))
            A = np.clip(g("r"), zero, dt(np.inf)) + one
            B = np.clip(g("K"), zero, dt(np.inf)) / np.clip(g("r"), zero, dt(np.inf))
            mu = np.log(A) + np.log(x) - np.log(one + x / B)
        elif kind == KIND_MAY:                                  # :229-242
            xq = np.power(x, g("q"))
            exp_mu = x + x * g("r") * (one - x / g("M")) - g("a") * xq / (xq + np.power(g("b"), g("q")))
            mu = np.log(exp_mu)
        elif kind == KIND_MYERS:                                # :247-255
            A = g("r") + one
            mu = np.log(A) + g("theta") * np.log(x) - np.log(one + np.power(x, g("theta")) / g("M"))
        elif kind == KIND_RICKER:                               # :258-261
            mu = np.log(x) + g("r") * (one - x / g("K"))
        else:
            raise ValueError(kind)
        e = mu + g("sigma") * z
        return np.maximum(zero, np.exp(e) if simd_exp else _libm_exp(e)).astype(dtype)


def step_zoo(model, obs, t, action, z, P, K_obs, Tmax=100, kind=None, dtype=np.float64, simd_exp=False):
    """step() (base_fishing_env.py:60-81) with a zoo population_draw.  K_obs is the env's
    self.K (obs <-> population map, quota); the growth parameters come from P
    (self.params).  fishing-v10: P["r"] is the value AFTER this step's `r += alpha`
    (growth_models.py:151).  fishing-v11: `kind` is the per-env model index array."""
    dt = np.dtype(dtype).type
    obs = np.asarray(obs, dtype=dtype)
    K_obs = np.asarray(K_obs, dtype=dtype)
    one, zero = dt(1.0), dt(0.0)
    with np.errstate(all="ignore"):
        quota = quota_from_action(MODEL_V1, action, K_obs, 0, dtype)
        x = (obs + one) * K_obs
        h = np.where(quota < x, quota, x)
        d = x - h
        x = np.where(zero > d, zero, d)
        if model == MODEL_V11:
            kind = np.asarray(kind)
            xn = np.zeros_like(x)
            for k in range(5):
                m = kind == k
                if m.any():
                    Pk = dict(V11_TABLE[k]) if P is None else dict(P[k])
                    xn[m] = zoo_population_draw(k, x[m], np.asarray(z, dtype=dtype)[m], Pk, dtype, simd_exp)
            x = xn
        else:
            x = zoo_population_draw(KIND_OF_MODEL[model], x, z, P, dtype, simd_exp)
        obs_next = x / K_obs - one
        reward = np.where(zero > h, zero, h)
        t_next = np.asarray(t, dtype=np.int32) + np.int32(1)
        done = (t_next > np.int32(Tmax)) | (x <= zero)
    return (obs_next.astype(dtype), reward.astype(dtype), done.astype(np.uint8), t_next.astype(np.int32),
            x.astype(dtype))


def draw_model_error_params(zK, zr, K_mean, r_mean, sigma_p, dtype=np.float64):
    """fishing_model_error.py:37-38 / :42-43.  np.random.normal(loc, scale) is
    loc + scale * z; K is drawn first, then r; both clipped to [0, 1e6]."""
    dt = np.dtype(dtype).type
    zK = np.asarray(zK, dtype=dtype)
    zr = np.asarray(zr, dtype=dtype)
    K = np.clip(dt(K_mean) + dt(sigma_p) * zK, dt(0.0), dt(1e6))
    r = np.clip(dt(r_mean) + dt(sigma_p) * zr, dt(0.0), dt(1e6))
    return K.astype(dtype), r.astype(dtype)


def reset_obs(model, x0, K, dtype=np.float64):
    """base_fishing_env.py:84 -> x0 / K - 1;  fishing-v4 returns the
    UN-normalised x0 (fishing_model_error.py:44, quirk B8)."""
    dt = np.dtype(dtype).type
    if model == MODEL_V4:
        return np.broadcast_to(dt(x0), np.shape(K)).astype(dtype)
    return (dt(x0) / np.asarray(K, dtype=dtype) - dt(1.0)).astype(dtype)


# --------------------------------------------------------------------------
# Philox4x32-10 (Salmon et al. 2011) + the uniform / normal maps of the kernels
# --------------------------------------------------------------------------
_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32 with 10 rounds.  All inputs uint32 arrays/scalars."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for rnd in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            if rnd != 9:
                k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
                k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def philox_words(seed, env_index, step_counter, stream):
    """Counter layout shared with csrc/fishing_common.h (`env_index` = the Philox index:
    the env QUAD index on the noise and policy streams (noise_normal, policy_random_action); fishing-v4's
    parameter draws use param_words, fishing-v11's model draw model_words instead):
    c0 = index[31:0], c1 = stream<<24 | index[55:32], c2 = step[31:0], c3 = step[63:32];
    key = (seed[31:0], seed[63:32])."""
    env = np.asarray(env_index, dtype=np.uint64)
    step_counter = int(step_counter)
    c0 = (env & _MASK).astype(np.uint32)
    c1 = (((env >> np.uint64(32)) & np.uint64(0xFFFFFF)).astype(np.uint32)
          | np.uint32((int(stream) & 0xFF) << 24))
    c2 = np.uint32(step_counter & 0xFFFFFFFF)
    c3 = np.uint32((step_counter >> 32) & 0xFFFFFFFF)
    seed = int(seed)
    return philox4x32_10(c0, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def uniform_open(w):
    """u in (0, 1]:  float32(w) * 2**-32 + 2**-33  (each op rounded to float32)."""
    return (np.asarray(w, dtype=np.uint32).astype(np.float32) * np.float32(2.0 ** -32)
            + np.float32(2.0 ** -33))


def uniform_turn(w):
    """u in [0, 1]:  float32(w) * 2**-32 (fraction of a full turn)."""
    return np.asarray(w, dtype=np.uint32).astype(np.float32) * np.float32(2.0 ** -32)


def box_muller(w0, w1):
    """Two standard normals from two words, evaluated in float64 and rounded to
    float32 -- the value the device's hardware log2/sqrt/cos/sin approximate."""
    u1 = uniform_open(w0).astype(np.float64)
    u2 = uniform_turn(w1).astype(np.float64)
    rad = np.sqrt(-2.0 * np.log(u1))
    return ((rad * np.cos(2.0 * np.pi * u2)).astype(np.float32),
            (rad * np.sin(2.0 * np.pi * u2)).astype(np.float32))


def _quad_words(seed, env_index, step_counter, stream):
    """(even-leg word, odd-leg word, leg) of env_index within its quad's block."""
    env = np.asarray(env_index, dtype=np.uint64)
    w = philox_words(seed, env >> np.uint64(2), step_counter, stream)
    return env, w


def noise_normal(seed, env_index, step_counter):
    """Process-noise z (float32) of global env `env_index` at global step `step_counter`.
    One Philox block serves an env QUAD (index env >> 2): Box-Muller of words (0, 1) gives the
    cos / sin legs = z of envs 4q, 4q + 1, of words (2, 3) z of envs 4q + 2, 4q + 3
    (fishing_common.h: noise_quad)."""
    env, (w0, w1, w2, w3) = _quad_words(seed, env_index, step_counter, STREAM_NOISE)
    hi = (env & np.uint64(2)).astype(bool)
    zc, zs = box_muller(np.where(hi, w2, w0), np.where(hi, w3, w1))
    return np.where((env & np.uint64(1)).astype(bool), zs, zc).astype(np.float32)


def quad_word(seed, env_index, counter, stream):
    """Word (env & 3) of the Philox block of env quad (env >> 2) on `stream`: the per-env word of
    the policy stream (random actions)."""
    env, ws = _quad_words(seed, env_index, counter, stream)
    leg = (env & np.uint64(3)).astype(np.int64)
    return np.choose(leg, [np.broadcast_to(x, leg.shape) for x in ws]).astype(np.uint32)


def model_words(seed, env_index, counter, stream):
    """The 16-bit half behind fishing-v11's model draw of env `env_index` (fishing_common.h: model_block): ONE Philox2x32-10
    block per env QUAD -- c0 = quad[31:0], c1 = (counter[30:0] | 1 << 31 on the reset stream) ^ quad[63:32] * 0xC2B2AE35, key =
    seed[31:0] ^ seed[63:32] * 0x85EBCA6B ^ 0x4D4F444C ^ counter[62:31] * 0x9E3779B1 -- whose two words give four halves: env 4q + j takes
    w0 & 0xFFFF, w0 >> 16, w1 & 0xFFFF, w1 >> 16 for j = 0 .. 3."""
    env = np.asarray(env_index, dtype=np.uint64)
    quad = env >> np.uint64(2)
    counter, seed, stream = int(counter), int(seed), int(stream)
    m32 = 0xFFFFFFFF
    key = np.uint32(((seed & m32) ^ (((seed >> 32) * 0x85EBCA6B) & m32)) ^ 0x4D4F444C ^ ((((counter >> 31) & m32) * 0x9E3779B1) & m32))
    c0 = (quad & _MASK).astype(np.uint32)
    c1 = (np.uint64((counter & 0x7FFFFFFF) | (0x80000000 if stream == STREAM_RESET else 0))
          ^ (((quad >> np.uint64(32)) * np.uint64(0xC2B2AE35)) & np.uint64(m32))).astype(np.uint32)
    w0, w1 = philox2x32_10(c0, c1, key)
    leg = (env & np.uint64(3)).astype(np.int64)
    w = np.where(leg < 2, w0, w1)
    return np.where(leg & 1, w >> np.uint32(16), w & np.uint32(0xFFFF)).astype(np.uint32)


def model_draw(seed, env_index, counter, stream, kinds):
    """fishing-v11 (growth_models.py:187,200: np.random.choice(models)): index into the model list, (half * n_models) >> 16
    of the env's 16-bit half (model_words), mapped to its FISHING_KIND_*.  Every model's probability is within 2^-16 of
    1 / n_models."""
    h = model_words(seed, env_index, counter, stream).astype(np.uint64)
    idx = ((h * np.uint64(len(kinds))) >> np.uint64(16)).astype(np.int64)
    return np.asarray(kinds, dtype=np.int32)[idx]


def policy_random_action(model, seed, env_index, step_counter, n_actions=100):
    """Random policy sampled in-kernel: word (env & 3) of the quad's block on the policy
    stream: continuous a = float32(w) * 2**-31 - 1 in [-1, 1]; discrete a = (w * n_actions) >> 32."""
    w = quad_word(seed, env_index, step_counter, STREAM_POLICY)
    if model == MODEL_V0:
        return ((w.astype(np.uint64) * np.uint64(n_actions)) >> np.uint64(32)).astype(np.int32)
    return w.astype(np.float32) * np.float32(2.0 ** -31) - np.float32(1.0)


def policy_action(policy, param, model, obs, K, n_actions=100, dtype=np.float64):
    """The reference's non-learned policies as the rollout kernel evaluates them
    (models/policies.py:16-19 msy, :27-31 escapement): quota -> get_action (:149-156).
    Continuous actions are float32 (the action Box dtype the reference clips to); discrete
    ones use Python round() = round-half-even."""
    dt = np.dtype(dtype).type
    obs = np.asarray(obs, dtype=dtype)
    K = np.asarray(K, dtype=dtype)
    if policy == "escapement":
        d = (obs + dt(1.0)) * K - dt(param)
        q = np.where(dt(0.0) > d, dt(0.0), d)                     # max(x - S, 0.0)
    elif policy == "msy":
        q = np.broadcast_to(dt(param), obs.shape)
    else:
        raise ValueError(policy)
    if model == MODEL_V0:
        return np.rint((q * dt(n_actions) / K).astype(np.float64)).astype(np.int32)
    return (q / K - dt(1.0)).astype(np.float32)


_M2 = np.uint64(0xD256D193)


def philox2x32_10(c0, c1, k):
    """Vectorised Philox2x32 with 10 rounds (Salmon et al. 2011): one multiply per round, two words out.
    Inputs uint32 arrays / scalars (the key may be an array: one key per lane)."""
    c0, c1, k = (np.asarray(x, dtype=np.uint32) for x in (c0, c1, k))
    c0, c1, k = np.broadcast_arrays(c0, c1, k)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p = _M2 * c0.astype(np.uint64)
            hi = (p >> np.uint64(32)).astype(np.uint32)
            lo = (p & _MASK).astype(np.uint32)
            c0, c1 = hi ^ k ^ c1, lo
            k = (k.astype(np.uint64) + np.uint64(0x9E3779B9)).astype(np.uint32)
    return c0, c1


def _bitrev32(v):
    v = np.asarray(v, dtype=np.uint32)
    v = ((v >> np.uint32(1)) & np.uint32(0x55555555)) | ((v & np.uint32(0x55555555)) << np.uint32(1))
    v = ((v >> np.uint32(2)) & np.uint32(0x33333333)) | ((v & np.uint32(0x33333333)) << np.uint32(2))
    v = ((v >> np.uint32(4)) & np.uint32(0x0F0F0F0F)) | ((v & np.uint32(0x0F0F0F0F)) << np.uint32(4))
    v = ((v >> np.uint32(8)) & np.uint32(0x00FF00FF)) | ((v & np.uint32(0x00FF00FF)) << np.uint32(8))
    return (v >> np.uint32(16)) | (v << np.uint32(16))


def param_words(seed, env_index, counter, stream):
    """The two words behind fishing-v4's (K, r) draw of env `env_index` (fishing_common.h: param_block):
    Philox2x32-10 with counter words c0 = env[31:0], c1 = counter[30:0] | (1 << 31 on the reset stream) and
    key = seed[31:0] ^ seed[63:32] * 0x85EBCA6B ^ counter[62:31] * 0x9E3779B1 ^ env[63:32] * 0xC2B2AE35 (all mod 2^32):
    injective in (env, counter, stream) while env < 2^32 and counter < 2^31, so the reset() draws and the auto-reset
    draws never share a block."""
    env = np.asarray(env_index, dtype=np.uint64)
    counter, seed, stream = int(counter), int(seed), int(stream)
    m32 = 0xFFFFFFFF
    key0 = ((seed & m32) ^ (((seed >> 32) * 0x85EBCA6B) & m32)) ^ ((((counter >> 31) & m32) * 0x9E3779B1) & m32)
    env_hi = (env >> np.uint64(32)).astype(np.uint64)
    key = (np.uint64(key0) ^ ((env_hi * np.uint64(0xC2B2AE35)) & np.uint64(m32))).astype(np.uint32)
    c0 = (env & _MASK).astype(np.uint32)
    c1 = np.uint32((counter & 0x7FFFFFFF) | (0x80000000 if stream == STREAM_RESET else 0))
    return philox2x32_10(c0, np.broadcast_to(c1, c0.shape), key)


def reset_normals(seed, env_index, counter, stream):
    """(zK, zr) float32 for a fishing-v4 parameter draw: the cos and sin legs of ONE Box-Muller pair, K first
    then r (fishing_model_error.py:42-43 order), from the env's own Philox2x32-10 block (param_words)."""
    w0, w1 = param_words(seed, env_index, counter, stream)
    return box_muller(w0, w1)


def v4_origin(step_counter, t, origin_step, origin_counter):
    """Which (stream, counter) drew the parameters in force for an env with years_passed `t` at global step
    `step_counter` (fishing_common.h: derive_model_error): the last full reset if the env has run since it
    (step_counter - t == origin_step), else the auto-reset of step (step_counter - t - 1)."""
    since = np.asarray(step_counter, dtype=np.int64) - np.asarray(t, dtype=np.int64)
    from_reset = since == int(origin_step)
    counter = np.where(from_reset, int(origin_counter), since - 1)
    stream = np.where(from_reset, STREAM_RESET, STREAM_AUTORESET)
    return stream, counter


def auto_reset(model, obs_next, done, t_next, K, r, x0, zK=None, zr=None, K_mean=1.0,
               r_mean=0.3, sigma_p=0.1, dtype=np.float64):
    """SB3 DummyVecEnv semantics around the reference: where done,
    keep obs_next as the terminal observation and return the reset observation,
    t = 0, and for fishing-v4 freshly drawn (K, r)."""
    done_b = np.asarray(done).astype(bool)
    K = np.broadcast_to(np.asarray(K, dtype=dtype), obs_next.shape).copy()
    r = np.broadcast_to(np.asarray(r, dtype=dtype), obs_next.shape).copy()
    if model == MODEL_V4:
        Kn, rn = draw_model_error_params(zK, zr, K_mean, r_mean, sigma_p, dtype)
        K = np.where(done_b, Kn, K)
        r = np.where(done_b, rn, r)
    obs_out = np.where(done_b, reset_obs(model, x0, K, dtype), obs_next).astype(dtype)
    t_out = np.where(done_b, np.int32(0), t_next).astype(np.int32)
    return obs_out, t_out, K.astype(dtype), r.astype(dtype)
