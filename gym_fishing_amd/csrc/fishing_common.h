// fishing_common.h -- device-side building blocks shared by the step / reset /
// rollout kernels of libfishing_hip.so (gfx950 only).
//
// Arithmetic contract (restated in oracle/fishing_oracle.py):
// every operation of the reference's step() is a separately rounded IEEE op in the
// reference's evaluation order.  The library is built with -ffp-contract=off, so the
// expressions below are NOT fused into FMAs; divisions are correctly rounded
// (hipcc default for f64, -fhip-fp32-correctly-rounded-divide-sqrt default for f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <tuple>
#include <type_traits>

#include "../../include/fishing_hip.h"

namespace fishing {

constexpr int kWave = 64;           // gfx950 wavefront
constexpr int kEnvsPerThread = 4;   // 16-byte accesses on the f32 streams
constexpr int kMaxBlocks = 4096;    // grid cap of every kernel that loops over tiles
#ifndef FISHING_PARTIAL_SLOTS
#define FISHING_PARTIAL_SLOTS 65536
#endif
constexpr int kPartialSlots = FISHING_PARTIAL_SLOTS;   // slots of the return_partials buffer = largest one-tile grid
static_assert(kPartialSlots >= kMaxBlocks, "every looping grid has its slots");
constexpr int kPartialFields = 4;   // {sum R, sum R^2, n_episodes, sum length}

constexpr uint32_t kStreamNoise = 0;      // step noise
constexpr uint32_t kStreamAutoReset = 1;  // v4 (K, r) / v11 model redraw inside step()
constexpr uint32_t kStreamReset = 2;      // v4 (K, r) / v11 model redraw in reset()
constexpr uint32_t kStreamPolicy = 3;     // random-policy actions of the fused rollout

enum NoiseMode { kNoiseNone = 0, kNoiseExt = 1, kNoisePhilox = 2 };

// ---------------------------------------------------------------- Philox4x32-10
// Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3" (SC'11).
// Counter-based: no generator state in HBM or LDS; the round keys are wave-uniform and
// live in SGPRs.  One 32x32->64 multiply (v_mad_u64_u32 / v_mul_hi+lo) per lane pair.
struct Words4 {
    uint32_t w0, w1, w2, w3;
};

// a ^ b ^ c in ONE instruction: gfx950's v_bitop3_b32 with truth table 0x96 (hipcc leaves two v_xor_b32 otherwise).  Every
// Philox round has two (4x32) or one (2x32) such update: 20 / 10 VALU instructions less per block -- 367 -> 347 in the
// headline step kernel, ~60 less in fishing-v4's (one noise block + four parameter blocks per thread).
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

__device__ __forceinline__ Words4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int rnd = 0; rnd < 10; ++rnd) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Words4{c0, c1, c2, c3};
}

// Counter layout (mirrored in oracle/fishing_oracle.py: philox_words):
//   c0 = index[31:0], c1 = stream << 24 | index[55:32], c2 = counter[31:0], c3 = counter[63:32]
// `index` is the global env QUAD index (env >> 2) on the noise and policy streams -- one block
// feeds the four envs of a thread tile: noise (w0, w1) -> Box-Muller cos / sin legs = z of envs
// 4q, 4q+1, (w2, w3) -> z of envs 4q+2, 4q+3; policy word j -> the random action of env 4q+j --,
// and for fishing-v11's model draw on the reset streams (redraw_kinds: word j -> env 4q + j).
// fishing-v4's (K, r) draws use a Philox2x32-10 block per ENV instead (draw_model_error).
__device__ __forceinline__ Words4 philox_block(uint64_t seed, uint64_t index, uint64_t counter,
                                               uint32_t stream) {
    return philox4x32_10((uint32_t)index, (stream << 24) | ((uint32_t)(index >> 32) & 0xFFFFFFu),
                         (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)seed,
                         (uint32_t)(seed >> 32));
}

// Philox2x32-10 (same paper; Random123 known answers in tests/test_oracle_golden.py): one 32x32->64
// multiply per round, two words out -- exactly the (zK, zr) Box-Muller pair of ONE env.  fishing-v4's
// parameter draws use it per env, so that the draw can be re-derived inside step() every step for
// half the multiplies a 4x32 block would cost (derive_model_error below).
__device__ __forceinline__ void philox2x32_10(uint32_t c0, uint32_t c1, uint32_t k, uint32_t& o0, uint32_t& o1) {
#pragma unroll
    for (int rnd = 0; rnd < 10; ++rnd) {
        const uint64_t p = (uint64_t)0xD256D193u * c0;
        c0 = xor3((uint32_t)(p >> 32), c1, k);
        c1 = (uint32_t)p;
        k += 0x9E3779B9u;
    }
    o0 = c0;
    o1 = c1;
}

// Block of the per-env parameter draw (mirrored in oracle/fishing_oracle.py: param_words; round 3 layout):
//   c0  = env[31:0]
//   c1  = counter[30:0] | (stream == kStreamReset ? 1 << 31 : 0)
//   key = seed[31:0] ^ seed[63:32] * 0x85EBCA6B ^ counter[62:31] * 0x9E3779B1 ^ env[63:32] * 0xC2B2AE35
// While env < 2^32 and counter < 2^31 -- every run so far: 2^31 steps are 15 hours at 25 us per step -- the block is an
// injective function of (env, counter, stream): the reset() draws and the auto-reset draws can never meet, whatever the
// two counters (round 2 told the streams apart by XOR-ing a tag into c1, which made reset counter a and step counter b
// share a block whenever a ^ b equalled the difference of the tags, b ~ 2.1e9).  There the key is wave-uniform and its
// ten round keys stay in SGPRs, like the noise block's; beyond, the high parts perturb the key per lane.
// Key space: Philox2x32 takes ONE 32-bit key, so the 64-bit seed is folded -- two seeds share their parameter stream with
// probability 2^-32 (the noise stream, Philox4x32, carries the full 64-bit seed and is not affected).
constexpr uint32_t kParamResetBit = 0x80000000u;
__device__ __forceinline__ uint32_t param_key(uint64_t seed) {
    return (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x85EBCA6Bu);
}
__device__ __forceinline__ void param_block(uint64_t seed, uint64_t env, uint64_t counter, bool reset_stream, uint32_t& w0,
                                            uint32_t& w1) {
    const uint32_t c1 = ((uint32_t)counter & ~kParamResetBit) | (reset_stream ? kParamResetBit : 0u);
    const uint32_t key = param_key(seed) ^ ((uint32_t)(counter >> 31) * 0x9E3779B1u) ^ ((uint32_t)(env >> 32) * 0xC2B2AE35u);
    philox2x32_10((uint32_t)env, c1, key, w0, w1);
}

// Two standard normals from two words.  u1 in (0, 1], u2 = fraction of a turn in [0, 1].
// Hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, v_cos_f32 / v_sin_f32 (argument
// in revolutions, so no 2*pi multiply).  |z| <= sqrt(66 ln 2) = 6.76.
__device__ __forceinline__ void box_muller(uint32_t w0, uint32_t w1, float& zc, float& zs) {
    const float u1 = __builtin_fmaf((float)w0, 0x1p-32f, 0x1p-33f);      // (the product is exact: same bits as mul + add, one instruction)
    const float u2 = (float)w1 * 0x1p-32f;
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    zc = rad * __builtin_amdgcn_cosf(u2);
    zs = rad * __builtin_amdgcn_sinf(u2);
}

// Process noise of the env quad {4 * quad .. 4 * quad + 3} at step `counter`: ONE Philox block and
// two Box-Muller pairs for a thread's four envs (env_offset and tile bases are multiples of 4, so a
// quad never straddles threads or shards).
__device__ __forceinline__ void noise_quad(uint64_t seed, uint64_t quad, uint64_t counter, float (&z)[4]) {
    const Words4 w = philox_block(seed, quad, counter, kStreamNoise);
    box_muller(w.w0, w.w1, z[0], z[1]);
    box_muller(w.w2, w.w3, z[2], z[3]);
}

// random-policy action from one word: continuous a in [-1, 1]; discrete in [0, n_actions)
__device__ __forceinline__ float action_cts_from_word(uint32_t w) { return (float)w * 0x1p-31f - 1.0f; }
__device__ __forceinline__ int32_t action_int_from_word(uint32_t w, int32_t n_actions) {
    return (int32_t)(((uint64_t)w * (uint64_t)(uint32_t)n_actions) >> 32);
}

// ---------------------------------------------------------------- scalar params in T
// template tags of the zoo (fishing-v5..v11): kModelZoo + FISHING_KIND_* = one instantiation per
// growth function (v5..v10: the kind is a compile-time constant, so each kernel carries only
// its own log/exp/pow chain); kModelZooMixed = fishing-v11, kind per env (per-lane switch)
constexpr int kModelZoo = 100;
constexpr int kModelZooMixed = kModelZoo + FISHING_N_KINDS;
constexpr int kModelZooRT = kModelZooMixed + 1;    // general kernel: one growth function, kind decided at run time
constexpr bool is_zoo_tag(int model_tag) { return model_tag >= kModelZoo && model_tag <= kModelZooRT; }

template <typename T>
struct GrowthT {                 // FishingGrowthParams narrowed to T
    T r, K, sigma, C, M, theta, q, b, a;
    // per-launch constants of the growth functions, evaluated once on the host in double
    // (libm) instead of once per env on the device:
    T bq;      // May:            b ** q                       (growth_models.py:238)
    T logA;    // B-H / Myers:    log(clip(r, 0, inf) + 1) / log(r + 1)   (:222,:225 / :248,:251)
    T B;       // Beverton-Holt:  clip(K, 0, inf) / clip(r, 0, inf)       (:224)
};

template <typename T>
inline GrowthT<T> make_growth(double r, double K, double sigma, double C, double M, double theta, double q,
                              double b, double a, bool beverton_holt) {
    GrowthT<T> g{(T)r, (T)K, (T)sigma, (T)C, (T)M, (T)theta, (T)q, (T)b, (T)a, (T)0, (T)0, (T)0};
    g.bq = (T)std::pow(b, q);
    if (beverton_holt) {
        const double rc = r < 0 ? 0 : r, Kc = K < 0 ? 0 : K;
        g.logA = (T)std::log(rc + 1.0);
        g.B = (T)(Kc / rc);
    } else {
        g.logA = (T)std::log(r + 1.0);
    }
    return g;
}

inline bool is_zoo_model(int model) { return model >= FISHING_MODEL_V5 && model <= FISHING_MODEL_V11; }
inline bool is_core_model(int model) {
    return model == FISHING_MODEL_V0 || model == FISHING_MODEL_V1 || model == FISHING_MODEL_V2 || model == FISHING_MODEL_V4;
}

// growth-function kind of a single-kind zoo model (v11 carries it per env)
inline int kind_of_model(int model) {
    switch (model) {
        case FISHING_MODEL_V5: return FISHING_KIND_ALLEN;
        case FISHING_MODEL_V7: return FISHING_KIND_MAY;
        case FISHING_MODEL_V8: return FISHING_KIND_MYERS;
        case FISHING_MODEL_V9: return FISHING_KIND_RICKER;
        default: return FISHING_KIND_BEVERTON_HOLT;   // v6, v10
    }
}

template <typename T>
struct ParamsT {
    int32_t model, n_actions, Tmax;
    uint32_t flags;
    T r, K, sigma, C, x0, r_mean, K_mean, sigma_p;
    T M, theta, q, b, a, alpha;
    GrowthT<T> growth;           // the single growth function of fishing-v5..v10 (+ host constants)
    int32_t kind;                // FISHING_KIND_* of that function (run-time kind of the general kernel)
    uint64_t origin_step, origin_counter;   // fishing-v4 derived parameters: see derive_model_error
    int32_t n_models;
    int32_t kinds[FISHING_N_KINDS];
    GrowthT<T> zoo[FISHING_N_KINDS];
};

template <typename T>
inline ParamsT<T> narrow_params(const FishingParams& p) {
    ParamsT<T> q;
    q.model = p.model;
    q.n_actions = p.n_actions;
    q.Tmax = p.Tmax;
    q.flags = p.flags;
    q.r = (T)p.r;
    q.K = (T)p.K;
    q.sigma = (T)p.sigma;
    q.C = (T)p.C;
    q.x0 = (T)p.x0;
    q.r_mean = (T)p.r_mean;
    q.K_mean = (T)p.K_mean;
    q.sigma_p = (T)p.sigma_p;
    q.M = (T)p.M;
    q.theta = (T)p.theta;
    q.q = (T)p.q;
    q.b = (T)p.b;
    q.a = (T)p.a;
    q.alpha = (T)p.alpha;
    q.n_models = p.n_models;
    q.kind = kind_of_model(p.model);
    q.origin_step = p.v4_origin_step;
    q.origin_counter = p.v4_origin_counter;
    // the host-side constants (pow / log) only where a growth function of the zoo will read them
    q.growth = GrowthT<T>{};
    if (is_zoo_model(p.model) && p.model != FISHING_MODEL_V11)
        q.growth = make_growth<T>(p.r, p.K, p.sigma, p.C, p.M, p.theta, p.q, p.b, p.a,
                                  p.model == FISHING_MODEL_V6 || p.model == FISHING_MODEL_V10);
    for (int k = 0; k < FISHING_N_KINDS; ++k) {
        q.kinds[k] = p.kinds[k];
        q.zoo[k] = GrowthT<T>{};
        if (p.model == FISHING_MODEL_V11) {
            const FishingGrowthParams& g = p.zoo[k];
            q.zoo[k] = make_growth<T>(g.r, g.K, g.sigma, g.C, g.M, g.theta, g.q, g.b, g.a, k == FISHING_KIND_BEVERTON_HOLT);
        }
    }
    return q;
}

template <typename T>
struct BuffersT {
    T* obs;
    const void* action;
    T* reward;
    uint8_t* done;
    uint64_t* done_bits;
    int32_t* t;
    T* r;
    T* K;
    const T* sigma;
    const T* z_ext;
    T* terminal_obs;
    T* ep_return;
    double* partials;
    int32_t* model_idx;
    const uint64_t* counter;
};

template <typename T>
inline BuffersT<T> typed_buffers(const FishingBuffers& b) {
    BuffersT<T> q;
    q.obs = (T*)b.obs;
    q.action = b.action;
    q.reward = (T*)b.reward;
    q.done = b.done;
    q.done_bits = b.done_bits;
    q.t = b.t;
    q.r = (T*)b.r;
    q.K = (T*)b.K;
    q.sigma = (const T*)b.sigma;
    q.z_ext = (const T*)b.z_ext;
    q.terminal_obs = (T*)b.terminal_obs;
    q.ep_return = (T*)b.ep_return;
    q.partials = b.return_partials;
    q.model_idx = b.model_idx;
    q.counter = b.counter;
    return q;
}

// Host-side tag dispatch: calls f(std::integral_constant<int, TAG>{}) with the kernel template tag
// of `model` (the model id itself for fishing-v0/v1/v2/v4; kModelZoo + kind for v5..v10;
// kModelZooMixed for v11).  Keeps the run-time -> compile-time switch in one place.
template <int TAG>
using ModelTag = std::integral_constant<int, TAG>;

template <typename F>
inline int with_model_tag(int model, F&& f) {
    switch (model) {
        case FISHING_MODEL_V0: return f(ModelTag<FISHING_MODEL_V0>{});
        case FISHING_MODEL_V1: return f(ModelTag<FISHING_MODEL_V1>{});
        case FISHING_MODEL_V2: return f(ModelTag<FISHING_MODEL_V2>{});
        case FISHING_MODEL_V4: return f(ModelTag<FISHING_MODEL_V4>{});
        case FISHING_MODEL_V11: return f(ModelTag<kModelZooMixed>{});
        default: break;
    }
    if (!is_zoo_model(model)) return FISHING_ERR_MODEL;
    switch (kind_of_model(model)) {
        case FISHING_KIND_ALLEN: return f(ModelTag<kModelZoo + FISHING_KIND_ALLEN>{});
        case FISHING_KIND_MYERS: return f(ModelTag<kModelZoo + FISHING_KIND_MYERS>{});
        case FISHING_KIND_MAY: return f(ModelTag<kModelZoo + FISHING_KIND_MAY>{});
        case FISHING_KIND_RICKER: return f(ModelTag<kModelZoo + FISHING_KIND_RICKER>{});
        default: return f(ModelTag<kModelZoo + FISHING_KIND_BEVERTON_HOLT>{});
    }
}

// the same for the general step kernel, which keeps the zoo's growth-function kind a run-time value
template <typename F>
inline int with_general_tag(int model, F&& f) {
    switch (model) {
        case FISHING_MODEL_V0: return f(ModelTag<FISHING_MODEL_V0>{});
        case FISHING_MODEL_V1: return f(ModelTag<FISHING_MODEL_V1>{});
        case FISHING_MODEL_V2: return f(ModelTag<FISHING_MODEL_V2>{});
        case FISHING_MODEL_V4: return f(ModelTag<FISHING_MODEL_V4>{});
        case FISHING_MODEL_V11: return f(ModelTag<kModelZooMixed>{});
        default: break;
    }
    if (!is_zoo_model(model)) return FISHING_ERR_MODEL;
    return f(ModelTag<kModelZooRT>{});
}

// Launch status of THIS launch (hipLaunchKernel's own return value), not whatever sticky error an
// unrelated earlier call left on the thread -- and without consuming that state either.
// `dyn_lds` bytes of (unused) dynamic LDS per workgroup cap how many workgroups a CU holds at once -- a launch-time
// occupancy limit (launch_kernel_lds).
template <typename... P, typename... A>
inline int launch_kernel_lds(void (*kernel)(P...), int blocks, int threads, size_t dyn_lds, hipStream_t stream, A&&... args) {
    std::tuple<P...> packed{static_cast<P>(args)...};
    return std::apply(
        [&](auto&... a) {
            void* argv[] = {(void*)&a...};
            return (int)hipLaunchKernel((const void*)kernel, dim3((unsigned)blocks), dim3((unsigned)threads), argv, dyn_lds,
                                        stream);
        },
        packed);
}
template <typename... P, typename... A>
inline int launch_kernel(void (*kernel)(P...), int blocks, int threads, hipStream_t stream, A&&... args) {
    return launch_kernel_lds(kernel, blocks, threads, 0, stream, std::forward<A>(args)...);
}

// ---------------------------------------------------------------- the env arithmetic
template <typename T>
__device__ __forceinline__ T exp_t(T e);
template <>
__device__ __forceinline__ double exp_t<double>(double e) {
    return exp(e);
}
template <>
__device__ __forceinline__ float exp_t<float>(float e) {
    return __expf(e);
}

// get_quota (base_fishing_env.py:135-147).  Continuous: the action (already widened to T;
// a float32 from the caller, or the T-valued output of an in-kernel policy) is clipped to
// [-1, 1] (NaN passes through, as np.clip), then (a + 1) * K.
template <typename T>
__device__ __forceinline__ T quota_cts(T av, T K) {
    av = (av < (T)-1) ? (T)-1 : av;
    av = (av > (T)1) ? (T)1 : av;
    return (av + (T)1) * K;
}
// Discrete: (a / n_actions) * K with true division (:140); no range check (quirk B11).
template <typename T>
__device__ __forceinline__ T quota_int(int32_t a, int32_t n_actions, T K) {
    return ((T)a / (T)n_actions) * K;
}

// get_action (base_fishing_env.py:149-156), used by the in-kernel escapement / MSY policies
// (models/policies.py:17-19, :29-31).  Continuous: quota / K - 1 evaluated in T, then
// rounded to float32 -- the dtype of the action Box the reference clips to
// (base_fishing_env.py:143-145; with the reference's pinned NumPy 1.19 np.clip against the
// float32 bounds returns float32), so an in-kernel policy emits exactly what a caller could
// pass through the float32 action stream.  Discrete: Python round(), i.e. round-half-even,
// of quota * n_actions / K.
struct DivK;   // exact power-of-two shortcut for x / K, defined below
template <typename T>
__device__ __forceinline__ T div_K(T x, T K, const DivK& d);
template <typename T>
__device__ __forceinline__ float action_cts_from_quota(T quota, T K) {
    return (float)(quota / K - (T)1);
}
template <typename T>
__device__ __forceinline__ float action_cts_from_quota(T quota, T K, const DivK& dk) {
    return (float)(div_K<T>(quota, K, dk) - (T)1);
}
template <typename T>
__device__ __forceinline__ int32_t action_int_from_quota(T quota, int32_t n_actions, T K) {
    return (int32_t)__builtin_rint((double)(quota * (T)n_actions / K));
}

template <typename T>
__device__ __forceinline__ T log_t(T v);
template <>
__device__ __forceinline__ double log_t<double>(double v) {
    return log(v);
}
template <>
__device__ __forceinline__ float log_t<float>(float v) {
    return __builtin_amdgcn_logf(v) * 0.6931471805599453f;          // v_log_f32 is log2
}
// x ** e for x >= 0 as exp(e * log x): one log + one exp instead of the library pow (which
// carries full special-case handling and is ~4x the instructions); pow(0, e > 0) = exp(-inf) = 0,
// inf and NaN propagate.  Error ~ |e log x| ulp, far inside the zoo's parity tolerance.
template <typename T>
__device__ __forceinline__ T pow_t(T v, T e);
template <>
__device__ __forceinline__ double pow_t<double>(double v, double e) {
    return exp(e * log(v));
}
template <>
__device__ __forceinline__ float pow_t<float>(float v, float e) {
    return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(v));    // v_exp_f32 is 2**x
}

// The five growth functions of growth_models.py:208-261; each ends in
// np.maximum(0, np.random.lognormal(mu, sigma)) = max(0, exp(mu + sigma z)).  The reference
// really does round-trip through log and exp (also at sigma = 0); so does this.
// RECOMPUTE: P.r changed on the device (fishing-v10 drift) -> logA / B are evaluated here
template <typename T, int KIND = -1, bool RECOMPUTE = false>
__device__ __forceinline__ T zoo_population_draw(int kind_rt, T x, T z, const GrowthT<T>& P) {
    const T inf = (T)__builtin_huge_val();
    const int kind = (KIND >= 0) ? KIND : kind_rt;      // compile-time kind folds the switch away
    T mu;
    switch (kind) {
        case FISHING_KIND_ALLEN:          // :208-217
            mu = log_t<T>(x) + P.r * ((T)1 - x / P.K) * ((T)1 - P.C) / P.K;
            break;
        case FISHING_KIND_MYERS: {        // :247-255   (log(A), A = r + 1, comes from the host)
            const T lx = log_t<T>(x);
            mu = P.logA + P.theta * lx - log_t<T>((T)1 + exp_t<T>(P.theta * lx) / P.M);   // x**theta
            break;
        }
        case FISHING_KIND_MAY: {          // :229-242   (b**q comes from the host)
            const T xq = pow_t<T>(x, P.q);
            const T exp_mu = x + x * P.r * ((T)1 - x / P.M) - P.a * xq / (xq + P.bq);
            mu = log_t<T>(exp_mu);
            break;
        }
        case FISHING_KIND_RICKER:         // :258-261
            mu = log_t<T>(x) + P.r * ((T)1 - x / P.K);
            break;
        default: {                        // Beverton-Holt :220-226 (np.clip(., 0, inf): NaN passes)
            const T xc = (x < (T)0) ? (T)0 : ((x > inf) ? inf : x);
            T logA = P.logA, B = P.B;
            if (RECOMPUTE) {
                const T rc = (P.r < (T)0) ? (T)0 : P.r;
                const T Kc = (P.K < (T)0) ? (T)0 : P.K;
                logA = log_t<T>(rc + (T)1);
                B = Kc / rc;
            }
            mu = logA + log_t<T>(xc) - log_t<T>((T)1 + xc / B);
            break;
        }
    }
    const T g = exp_t<T>(mu + P.sigma * z);
    return (g > (T)0) ? g : ((g != g) ? g : (T)0);    // np.maximum(0, g)
}

// fishing-v11: the growth function differs per env, so a straight per-lane switch runs all five
// functions for every one of a thread's four envs (20 masked passes per wave, most lanes idle in
// each).  Instead each wave regroups its 256 envs BY KIND through a wave-private LDS window, ONCE for
// all kinds: every env gets a slot = start of its kind's segment + its rank inside the kind (ballot +
// mbcnt; segments are padded to whole 64-slot chunks, so a chunk never mixes kinds), (x, z) pairs go
// to their slots in one write phase, the wave evaluates growth function k on the chunks of segment k
// with wave-uniform parameters (ceil(n_k / 64) passes per kind, 5-7 in total), and every env reads
// its result back from its slot: three LDS phases per tile whatever the number of kinds.  Same
// function on the same inputs as zoo_population_draw<T, k>: identical bits.
// Must be called by all 64 lanes of the wave (envs that do not take part pass kind < 0).
// `win` is this wave's window: kZooWindowSlots pairs.  256 envs in 5 padded segments fill at most
// 512 slots (the padded total is a multiple of 64 below 256 + 5 * 63).
constexpr int kZooWindowSlots = 512;
template <typename T>
struct alignas(2 * sizeof(T)) ZooSlot {
    T x, z;
};

template <typename T, int K>
__device__ __forceinline__ void zoo_rank_kind(const int (&kind)[4], int (&slot)[4], int& begin, int& count, int& next) {
    int total = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool mine = kind[j] == K;
        const uint64_t bal = __ballot(mine);
        const int pos = next + total +
                        (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        slot[j] = mine ? pos : slot[j];
        total += __popcll(bal);                   // wave-uniform
    }
    begin = next;
    count = total;
    next += (total + kWave - 1) & ~(kWave - 1);
}

template <typename T, int K>
__device__ __forceinline__ void zoo_eval_kind(ZooSlot<T>* __restrict__ win, int begin, int count, const GrowthT<T>& P,
                                              int lane) {
    for (int c = 0; c < count; c += kWave) {      // wave-uniform trip count
        if (c + lane < count) {
            const ZooSlot<T> v = win[begin + c + lane];
            win[begin + c + lane].x = zoo_population_draw<T, K, false>(K, v.x, v.z, P);
        }
    }
}

template <typename T>
__device__ __forceinline__ void zoo_draw_regrouped(const int (&kind)[4], const T (&x)[4], const T (&z)[4],
                                                   const GrowthT<T> (&zoo)[FISHING_N_KINDS], T (&out)[4],
                                                   ZooSlot<T>* __restrict__ win, int lane) {
    int slot[4] = {-1, -1, -1, -1};
    int begin[FISHING_N_KINDS], count[FISHING_N_KINDS];
    int next = 0;
    zoo_rank_kind<T, FISHING_KIND_ALLEN>(kind, slot, begin[0], count[0], next);
    zoo_rank_kind<T, FISHING_KIND_BEVERTON_HOLT>(kind, slot, begin[1], count[1], next);
    zoo_rank_kind<T, FISHING_KIND_MYERS>(kind, slot, begin[2], count[2], next);
    zoo_rank_kind<T, FISHING_KIND_MAY>(kind, slot, begin[3], count[3], next);
    zoo_rank_kind<T, FISHING_KIND_RICKER>(kind, slot, begin[4], count[4], next);
    static_assert(FISHING_KIND_ALLEN == 0 && FISHING_KIND_BEVERTON_HOLT == 1 && FISHING_KIND_MYERS == 2 &&
                      FISHING_KIND_MAY == 3 && FISHING_KIND_RICKER == 4 && FISHING_N_KINDS == 5,
                  "segment order = kind order");
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (slot[j] >= 0) win[slot[j]] = ZooSlot<T>{x[j], z[j]};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    zoo_eval_kind<T, FISHING_KIND_ALLEN>(win, begin[0], count[0], zoo[FISHING_KIND_ALLEN], lane);
    zoo_eval_kind<T, FISHING_KIND_BEVERTON_HOLT>(win, begin[1], count[1], zoo[FISHING_KIND_BEVERTON_HOLT], lane);
    zoo_eval_kind<T, FISHING_KIND_MYERS>(win, begin[2], count[2], zoo[FISHING_KIND_MYERS], lane);
    zoo_eval_kind<T, FISHING_KIND_MAY>(win, begin[3], count[3], zoo[FISHING_KIND_MAY], lane);
    zoo_eval_kind<T, FISHING_KIND_RICKER>(win, begin[4], count[4], zoo[FISHING_KIND_RICKER], lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = (slot[j] >= 0) ? win[slot[j]].x : out[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();              // the next tile reuses the window
}

// step() with a zoo growth function: quota / obs maps use the env's K (K_obs), the growth its
// own parameter set (self.params in the reference).
template <typename T, int KIND = -1, bool RECOMPUTE = false>
__device__ __forceinline__ void env_step_zoo(T obs, int32_t t, T quota, T z, int kind, const GrowthT<T>& P,
                                             T K_obs, int32_t Tmax, T& obs_next, T& reward, bool& done,
                                             int32_t& t_next) {
    T x = (obs + (T)1) * K_obs;
    const T h = (quota < x) ? quota : x;
    const T d = x - h;
    x = ((T)0 > d) ? (T)0 : d;
    x = zoo_population_draw<T, KIND, RECOMPUTE>(kind, x, z, P);
    obs_next = x / K_obs - (T)1;
    reward = ((T)0 > h) ? (T)0 : h;
    t_next = t + 1;
    done = (t_next > Tmax) || (x <= (T)0);
}

// x / K.  When K is a power of two (the default K = 1 included) the quotient is exact up to the
// final rounding, and x * (1/K) with the exactly representable 1/K rounds to the same bits, so the
// ~12-instruction IEEE division sequence can be one multiply.  KP2 is decided on the host
// (is_pow2) and is wave-uniform; every other K keeps the correctly rounded division.
struct DivK {
    bool pow2;
    float inv_f;
    double inv_d;
};
inline DivK make_divk(double K) {
    int e = 0;
    const bool p2 = K > 0 && std::isfinite(K) && std::frexp(K, &e) == 0.5 && e > -120 && e < 120;
    return DivK{p2, p2 ? (float)(1.0 / K) : 0.0f, p2 ? 1.0 / K : 0.0};
}
template <typename T>
__device__ __forceinline__ T div_K(T x, T K, const DivK& d);
template <>
__device__ __forceinline__ float div_K<float>(float x, float K, const DivK& d) {
    return d.pow2 ? x * d.inv_f : x / K;
}
template <>
__device__ __forceinline__ double div_K<double>(double x, double K, const DivK& d) {
    return d.pow2 ? x * d.inv_d : x / K;
}

// population_draw(): base_fishing_env.py:121-133 (logistic), fishing_tipping_env.py:24-35
// (tipping point; the noise sits inside the exponent, scaled by x -- quirk B9).
template <typename T, int MODEL>
__device__ __forceinline__ T population_draw(T x, T z, T r, T K, T sigma, T C, const DivK& dk = DivK{false, 0.0f, 0.0}) {
    T g;
    const T xk = div_K<T>(x, K, dk);
    if (MODEL == FISHING_MODEL_V2) {
        const T e = ((r * ((T)1 - xk)) * (x - C)) + ((x * sigma) * z);
        g = x * exp_t<T>(e);
    } else {
        g = (x + ((r * x) * ((T)1 - xk))) + ((x * sigma) * z);
    }
    return (g > (T)0) ? g : ((g != g) ? g : (T)0);   // np.maximum(g, 0.0): NaN-propagating, -0 -> +0
}

// One reference step() on one env.  Returns through refs.
template <typename T, int MODEL>
__device__ __forceinline__ void env_step(T obs, int32_t t, T quota, T z, T r, T K, T sigma, T C,
                                         int32_t Tmax, T& obs_next, T& reward, bool& done,
                                         int32_t& t_next, const DivK& dk = DivK{false, 0.0f, 0.0}) {
    T x = (obs + (T)1) * K;                   // get_fish_population  :159
    const T h = (quota < x) ? quota : x;      // min(x, quota)        :117
    const T d = x - h;
    x = ((T)0 > d) ? (T)0 : d;                // max(x - h, 0.0)      :118
    x = population_draw<T, MODEL>(x, z, r, K, sigma, C, dk);
    obs_next = div_K<T>(x, K, dk) - (T)1;     // get_state            :163
    reward = ((T)0 > h) ? (T)0 : h;           // max(harvest, 0.0)    :74
    t_next = t + 1;                           //                      :75
    done = (t_next > Tmax) || (x <= (T)0);    //                      :76-79
}

// fishing_model_error.py:37-38 / :42-43: K = clip(K_mean + sigma_p * zK, 0, 1e6), then r.  The argument is never NaN here:
// zK is finite (|z| <= 6.76) and the host refuses non-finite K_mean / r_mean / sigma_p for fishing-v4 (check_common), so
// np.clip is one v_med3_f32 (float) or max + min (double) instead of two compare + select pairs.
template <typename T>
__device__ __forceinline__ T clip_param(T v);
template <>
__device__ __forceinline__ float clip_param<float>(float v) {
    return __builtin_amdgcn_fmed3f(v, 0.0f, 1e6f);
}
template <>
__device__ __forceinline__ double clip_param<double>(double v) {
    return __builtin_fmin(__builtin_fmax(v, 0.0), 1e6);
}
// fishing-v11 (growth_models.py:187,200): a new growth function for the finished envs of one thread's
// 4-env tile.  One Philox block per env quad on the reset streams, word j -> env 4q + j.  Returns
// whether any kind was redrawn.
__device__ __forceinline__ bool redraw_kinds(uint64_t seed, uint64_t base, uint64_t counter, uint32_t stream,
                                             const int32_t (&kinds)[FISHING_N_KINDS], int32_t n_models,
                                             const bool (&fin)[4], int32_t (&kind)[4]) {
    if (!(fin[0] | fin[1] | fin[2] | fin[3])) return false;
    const Words4 w = philox_block(seed, base >> 2, counter, stream);
    const uint32_t ww[4] = {w.w0, w.w1, w.w2, w.w3};
#pragma unroll
    for (int j = 0; j < 4; ++j) kind[j] = fin[j] ? kinds[action_int_from_word(ww[j], n_models)] : kind[j];
    return true;
}

// reset observation: x0 / K - 1 (base_fishing_env.py:84); v4 returns x0 un-normalised
// (fishing_model_error.py:44, quirk B8).
template <typename T, int MODEL>
__device__ __forceinline__ T reset_obs(T x0, T K) {
    return (MODEL == FISHING_MODEL_V4) ? x0 : (x0 / K - (T)1);
}

// fishing-v4 (K, r) draw of ONE env (fishing_model_error.py:37-38 / :42-43: K first, then r): one
// Philox2x32-10 block keyed by (seed, env, counter, stream) -> Box-Muller (zK, zr).
//   stream kStreamReset,     counter = reset counter          : reset()
//   any other stream tag (kStreamAutoReset), counter = step counter of the step that finished the episode
template <typename T>
__device__ __forceinline__ void draw_model_error_block(uint64_t seed, uint64_t env, uint64_t counter, bool reset_stream,
                                                       T K_mean, T r_mean, T sigma_p, T& K, T& r) {
    uint32_t w0, w1;
    param_block(seed, env, counter, reset_stream, w0, w1);
    float zK, zr;
    box_muller(w0, w1, zK, zr);
    K = clip_param<T>(K_mean + sigma_p * (T)zK);
    r = clip_param<T>(r_mean + sigma_p * (T)zr);
}
template <typename T>
__device__ __forceinline__ void draw_model_error(uint64_t seed, uint64_t env, uint64_t counter,
                                                 uint32_t stream, T K_mean, T r_mean, T sigma_p,
                                                 T& K, T& r) {
    draw_model_error_block<T>(seed, env, counter, stream == kStreamReset, K_mean, r_mean, sigma_p, K, r);
}

// The (K, r) in force for an env are a pure function of where its episode began, and that is readable
// from the env's own year counter: at global step c an env with years_passed t either has run since
// the last full reset() (made at step count `origin_step` with reset counter `origin_counter`), in
// which case c - t == origin_step, or was auto-reset by the step with counter c - t - 1.  So a
// fishing-v4 batch on the Philox streams needs NO r / K arrays in HBM: step() re-derives them
// (FISHING_FLAG_V4_DERIVED; 8 B/env-step of reads and, on the random-policy workload where nearly every
// 128-byte line holds a finished env, 8 B/env-step of redraw writes saved).  Same values bit for bit
// as the stored-array path, which draws from the same blocks at the moment of the reset.
template <typename T, bool NARROW = false>
__device__ __forceinline__ void derive_model_error(uint64_t seed, uint64_t env, uint64_t step_counter, int32_t t,
                                                   uint64_t origin_step, uint64_t origin_counter, T K_mean,
                                                   T r_mean, T sigma_p, T& K, T& r) {
    if constexpr (NARROW) {
        // every env index of the wave fits 32 bits and every counter 31 (derive_fits_32): the same block with half the
        // integer work -- param_block's key terms of the high parts are zero, the key is wave-uniform
        const uint32_t since = (uint32_t)step_counter - (uint32_t)t;
        const bool from_reset = since == (uint32_t)origin_step;
        uint32_t w0, w1;
        philox2x32_10((uint32_t)env, from_reset ? ((uint32_t)origin_counter | kParamResetBit) : since - 1u, param_key(seed), w0, w1);
        float zK, zr;
        box_muller(w0, w1, zK, zr);
        K = clip_param<T>(K_mean + sigma_p * (T)zK);
        r = clip_param<T>(r_mean + sigma_p * (T)zr);
    } else {
        const uint64_t since = step_counter - (uint64_t)(int64_t)t;
        const bool from_reset = since == origin_step;
        draw_model_error_block<T>(seed, env, from_reset ? origin_counter : since - 1, from_reset, K_mean, r_mean, sigma_p, K, r);
    }
}
// FishingBuffers.counter under FISHING_FLAG_V4_DERIVED: {step counter, origin step, origin counter} in device memory, so
// that a launch captured in a hipGraph (frozen arguments) sees the origin of the LAST reset().  Wave-uniform scalar loads.
__device__ __forceinline__ void device_origin(const uint64_t* counter, uint64_t& origin_step, uint64_t& origin_counter) {
    if (counter) {
        origin_step = counter[1];
        origin_counter = counter[2];
    }
}

// wave-uniform: may this launch's derivations of the envs [env_first, env_last] use the 32-bit form?  (An env whose
// years_passed exceeds the step count is out of contract -- the host advances both together -- so since - 1 cannot wrap
// except as the unused arm of the from_reset select.)
__device__ __forceinline__ bool derive_fits_32(uint64_t env_last, uint64_t step_counter, uint64_t origin_step,
                                               uint64_t origin_counter) {
    return ((env_last >> 32) | ((step_counter | origin_step | origin_counter) >> 31)) == 0;
}

// Redraw (K, r) and restart the finished envs of one thread's 4-env tile (`base` = global index of
// its first env).  Returns whether anything was redrawn.
template <typename T, int MODEL, int E>
__device__ __forceinline__ bool redraw_tile(uint64_t seed, uint64_t base, uint64_t counter, uint32_t stream,
                                            T K_mean, T r_mean, T sigma_p, T x0, const bool (&fin)[E],
                                            T (&KK)[E], T (&rr)[E], T (&obs)[E], int32_t (&t)[E]) {
    bool any = false;
#pragma unroll
    for (int j = 0; j < E; ++j) any |= fin[j];
    if (!any) return false;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        T K2, r2;
        draw_model_error<T>(seed, base + (uint64_t)j, counter, stream, K_mean, r_mean, sigma_p, K2, r2);
        KK[j] = fin[j] ? K2 : KK[j];
        rr[j] = fin[j] ? r2 : rr[j];
        obs[j] = fin[j] ? reset_obs<T, MODEL>(x0, KK[j]) : obs[j];
        t[j] = fin[j] ? 0 : t[j];
    }
    return true;
}

// ---------------------------------------------------------------- 4-wide access helpers
// 16-byte aligned at most: that is what the ABI guarantees for every buffer (a Vec4<double> is moved as two
// 16-byte accesses either way)
template <typename T, int E>
struct alignas((E * sizeof(T) >= 16) ? 16 : E * sizeof(T)) VecE {
    T v[E];
};
template <typename T>
using Vec4 = VecE<T, 4>;

template <typename T>
__device__ __forceinline__ void load4(const T* p, int64_t base, int64_t n, bool full, T (&out)[4], T fill) {
    if (full) {
        const Vec4<T> q = *reinterpret_cast<const Vec4<T>*>(p + base);
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = q.v[j];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = (base + j < n) ? p[base + j] : fill;
    }
}

// NT = 1: nontemporal (streaming) store -- an experiment knob (scripts/tune_variants.py);
// measured no faster than plain stores on gfx950 for this kernel, so the default is 0.
template <typename T, int NT = 0>
__device__ __forceinline__ void store4(T* p, int64_t base, int64_t n, bool full, const T (&in)[4]) {
    if (full) {
        if (NT) {
            typedef T VecT __attribute__((ext_vector_type(4)));
            VecT q;
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] = in[j];
            __builtin_nontemporal_store(q, reinterpret_cast<VecT*>(p + base));
        } else {
            Vec4<T> q;
#pragma unroll
            for (int j = 0; j < 4; ++j) q.v[j] = in[j];
            *reinterpret_cast<Vec4<T>*>(p + base) = q;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (base + j < n) p[base + j] = in[j];
    }
}

// years_passed stream: int32 (SURVEY layout) or, under FISHING_FLAG_T_U8, one byte per env (4 envs =
// one dword per lane).  Counters saturate at 255 in the byte form (Tmax <= 254 is enforced on the
// host, so `t' > Tmax` never needs a larger value).
__device__ __forceinline__ void load_t4(const int32_t* tp, bool u8, int64_t base, int64_t n, bool full,
                                        int32_t (&t)[4]) {
    if (!u8) {
        load4<int32_t>(tp, base, n, full, t, 0);
        return;
    }
    const uint8_t* p = reinterpret_cast<const uint8_t*>(tp);
    if (full) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(p + base);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (int32_t)((w >> (8 * j)) & 255u);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (base + j < n) ? (int32_t)p[base + j] : 0;
    }
}
template <int E>
__device__ __forceinline__ uint32_t pack_t4(const int32_t (&t)[E]) {
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < E; ++j) w |= (uint32_t)(t[j] > 255 ? 255 : t[j]) << (8 * j);
    return w;
}
__device__ __forceinline__ void store_t4(int32_t* tp, bool u8, int64_t base, int64_t n, bool full,
                                         const int32_t (&t)[4]) {
    if (!u8) {
        store4<int32_t>(tp, base, n, full, t);
        return;
    }
    uint8_t* p = reinterpret_cast<uint8_t*>(tp);
    if (full) {
        *reinterpret_cast<uint32_t*>(p + base) = pack_t4(t);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (base + j < n) p[base + j] = (uint8_t)(t[j] > 255 ? 255 : t[j]);
    }
}

// Wave-ballot done mask in the natural layout (bit i%64 of word i/64 = done[i]).
// Lane l holds the flags of envs 4l..4l+3 of its wave's 256-env tile as a nibble; word k of
// the tile collects lanes 16k..16k+15.  Lane L fetches the nibble of lane 16k + L/4 with a
// ds_bpermute (cross-lane, no LDS memory) and the 64 lanes vote their bit with one ballot.
// Returns, in lanes 0..3, words 0..3 of the tile.
// (E envs per lane: the wave's 64 * E flags make E words; word k collects lanes (64 / E) * k .. + 64 / E - 1.)
template <int E = 4>
__device__ __forceinline__ uint64_t ballot_tile_words(uint32_t nibble, int lane) {
    static_assert(E == 4 || E == 2, "4 or 2 envs per lane");
    uint64_t mine = 0;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const uint32_t nb = (uint32_t)__shfl((int)nibble, (kWave / E) * k + lane / E, kWave);
        const uint64_t word = __ballot((nb >> (lane % E)) & 1u);
        mine = (lane == k) ? word : mine;
    }
    return mine;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// Episodic-return record of one thread's 4-env tile.  The tile's finished returns and their squares
// are summed in T (four values) and widened once, episode count and length are integer adds: four
// double operations per tile instead of sixteen, which is 7 % of the fp32 step kernel's launch time
// on the random-policy workload where every wave finishes an env every step.  Every kernel uses
// this routine, so step-wise, fused-rollout and general-kernel records agree to double rounding.
template <typename T, int E>
__device__ __forceinline__ void record_tile(const bool (&fin)[E], const T (&er)[E], const int32_t (&len)[E],
                                            double (&acc)[4]) {
    T s1 = (T)0, s2 = (T)0;
    int32_t cnt = 0, tot = 0;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        s1 += fin[j] ? er[j] : (T)0;
        s2 += fin[j] ? er[j] * er[j] : (T)0;
        cnt += fin[j] ? 1 : 0;
        tot += fin[j] ? len[j] : 0;
    }
    acc[0] += (double)s1;
    acc[1] += (double)s2;
    acc[2] += (double)cnt;
    acc[3] += (double)tot;
}

// One DPP lane permutation of a double (two 32-bit v_mov_dpp, no LDS traffic).  Every lane of the
// wave must be active.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

constexpr int kDppQuadXor1 = 0xB1;   // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;   // quad_perm:[2,3,0,1]
constexpr int kDppRowRor4 = 0x124;   // row_ror:4
constexpr int kDppRowRor8 = 0x128;   // row_ror:8

// Sum the four record fields over each 16-lane DPP row.  A packed butterfly: the xor-2 exchange
// leaves every lane with two fields, the xor-1 exchange with one (field = lane & 3), and two row
// rotations add the row's four lanes of that field -- 7 double adds and 14 DPP moves per lane where
// four independent 64-lane shuffle trees cost 24 adds and 48 ds_bpermute.  On return lane l holds
// field (l & 3) summed over its row; the order of the additions is fixed.
__device__ __forceinline__ double row_sum_fields(const double (&acc)[kPartialFields], int lane) {
    const bool up2 = (lane & 2) != 0;
    double k0 = up2 ? acc[2] : acc[0], k1 = up2 ? acc[3] : acc[1];
    const double s0 = up2 ? acc[0] : acc[2], s1 = up2 ? acc[1] : acc[3];
    k0 += dpp_f64<kDppQuadXor2>(s0);
    k1 += dpp_f64<kDppQuadXor2>(s1);
    const bool up1 = (lane & 1) != 0;
    double k = up1 ? k1 : k0;
    const double s = up1 ? k0 : k1;
    k += dpp_f64<kDppQuadXor1>(s);
    k += dpp_f64<kDppRowRor4>(k);
    k += dpp_f64<kDppRowRor8>(k);
    return k;
}

// Episodic-return record: add this workgroup's partial {sum R, sum R^2, n, sum length} to its own
// slot of `partials`.  DPP butterfly inside each 16-lane row -> one LDS hop across the rows of all
// waves -> thread k adds field k.  One owner thread per slot and a fixed addition order, so the sums
// are bitwise reproducible for a fixed launch shape.  Must be reached by every thread of the
// workgroup with all lanes active (it contains a barrier and whole-wave DPP moves).
template <int MAX_WAVES, int STATIC_WAVES = 0>
__device__ __forceinline__ void add_block_partials(const double (&acc)[kPartialFields], double* partials) {
    constexpr int kRows = kWave / 16;
    __shared__ double red[MAX_WAVES * kRows][kPartialFields];
    const int lane = threadIdx.x & (kWave - 1);
    const double s = row_sum_fields(acc, lane);
    if ((lane & 15) < kPartialFields) red[threadIdx.x >> 4][lane & 3] = s;
    __syncthreads();
    if (threadIdx.x < kPartialFields) {
        double tot = 0.0;
        const int nrows = STATIC_WAVES ? STATIC_WAVES * kRows : (int)(blockDim.x >> 4);
#pragma unroll
        for (int w = 0; w < nrows; ++w) tot += red[w][threadIdx.x];
        // The slot belongs to this workgroup alone, so a hardware no-return atomic add gives the same
        // bits as a read-modify-write (one add per slot per launch, launches are stream-ordered) without
        // the dependent load -> add -> store round trip at the very end of the kernel.
        if (tot != 0.0) unsafeAtomicAdd(&partials[(int64_t)blockIdx.x * kPartialFields + threadIdx.x], tot);
    }
}

}  // namespace fishing
