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
// (fishing-v11's model draw: model_block below.)
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

// Block of the per-env parameter draw (mirrored in oracle/fishing_oracle.py: param_words):
//   c0  = env[31:0]
//   c1  = counter[30:0] | (stream == kStreamReset ? 1 << 31 : 0)
//   key = seed[31:0] ^ seed[63:32] * 0x85EBCA6B ^ counter[62:31] * 0x9E3779B1 ^ env[63:32] * 0xC2B2AE35
// While env < 2^32 and counter < 2^31 (2^31 steps are 15 hours at 25 us per step) the block is an injective function of
// (env, counter, stream): the reset() draws and the auto-reset draws can never meet, whatever the two counters.  There the
// key is wave-uniform and its ten round keys stay in SGPRs, like the noise block's; beyond, the high parts perturb the key per
// lane.  Philox2x32 takes ONE 32-bit key, so the 64-bit seed is folded -- two seeds share their parameter stream with
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

// One growth function's parameter set (FishingGrowthParams) plus per-launch constants evaluated once on the host in
// double (libm; fishing_host.h: make_growth).  Every field is a double whatever the layout T (the float32 kernels narrow
// what they read): wave-uniform, so they live in SGPR pairs and only the fields a kernel's growth function reads are ever
// loaded.
template <typename T>
struct GrowthT {
    double r, K, sigma, C, M, theta, q, b, a;
    double bq;      // May:            b ** q                       (growth_models.py:238)
    double logA;    // B-H / Myers:    log(clip(r, 0, inf) + 1) / log(r + 1)   (:222,:225 / :248,:251)
    double B;       // Beverton-Holt:  clip(K, 0, inf) / clip(r, 0, inf)       (:224)
    // the algebraic form (both layouts):
    double A;       // B-H: clip(r, 0, inf) + 1;  Myers: r + 1
    double invK;    // 1 / K        (Allen, Ricker;  B-H under drift: 1 / clip(K, 0, inf))
    double invM;    // 1 / M        (May, Myers)
    double invB;    // B-H: clip(r, 0, inf) / clip(K, 0, inf) = 1 / B
    double gc;      // Allen: r (1 - C) / K, the coefficient of (1 - x / K) in mu - log x
    int32_t ipow;   // May's q / Myers' theta when it is one of 1, 2, 3, 4 (x ** e by multiplication), else 0
};

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
    int32_t* stamp;
};

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
// get_action, discrete (base_fishing_env.py:149-152): round(quota * n_actions / K) -- Python's round() of a NumPy float is
// rint (half to even).  Rounded in T: a float32 value is exactly a double, so rintf(v) == rint((double)v).
template <typename T>
__device__ __forceinline__ int32_t action_int_from_quota(T quota, int32_t n_actions, T K, const DivK& dk) {
    const T v = div_K<T>(quota * (T)n_actions, K, dk);
    if constexpr (sizeof(T) == 4) return (int32_t)__builtin_rintf(v);
    else return (int32_t)__builtin_rint(v);
}

// ---------------------------------------------------------------- the zoo (fishing-v5..v11): how a growth function is evaluated
// The growth functions of growth_models.py:208-261 are exp(mu(x) + sigma z) with mu = log(x) + ... : a round trip through log and
// exp.  Both layouts evaluate the algebraically equal form WITHOUT the round trip,
//     x' = max(0, pre(x) * exp(g(x, z))):   pre = x,                             g = r (1 - x/K) (1 - C)/K + sigma z    Allen
//                                           pre = x,                             g = r (1 - x/K) + sigma z              Ricker
//                                           pre = A x / (1 + x/B),               g = sigma z                            Beverton-Holt
//                                           pre = A x^theta / (1 + x^theta/M),   g = sigma z                            Myers
//                                           pre = exp_mu(x)  (NaN when < 0),     g = sigma z                            May
// whose special values agree with the reference's (x = 0 -> 0, x = inf -> NaN, exp_mu < 0 -> NaN, NaN -> NaN; K, M, B of
// 0 or inf as IEEE division has them).  One exp and at most one division per env, no logarithm:
//   float32 layout: fma chains, one <= 1-ulp division (div_f32), exp on the hardware (v_exp_f32): <= 4.7e-7 of the reference's
//       float64 numbers on a dense sweep of the state space (the literal round trip in float32: 1.06e-6);
//   float64 parity layout: the same form on the < 1-ulp exp_f64 / div_f64 below: <= 2e-14 of the reference's numbers, closer to
//       the exact value than the reference's own round trip (whose log x + ... loses |mu| ulp before the exp).
// The alternatives that were measured and dropped (hardware / libm round trips, a float64 and a hybrid evaluation for the float32
// layout, the library's log / exp) and their error tables: profiles/NOTES_r01_r05.md, profiles/r04_zoo_f32_error.json,
// profiles/r05_zoo_f64_error.json.

// ---------------------------------------------------------------- float64 log / exp of the parity layout's zoo
// The float64 zoo kernels are VALU-bound on their transcendentals: the device library's log is 98 VALU instructions (a
// double-double evaluation), its exp 42; fishing-v11 inlines eight logs and seven exps.  These are the classic < 1 ulp
// forms of FreeBSD msun / fdlibm (e_log.c, e_exp.c: the published argument reductions and minimax coefficients, restated
// here) with the two divisions done as a v_rcp_f64 seed + Newton steps on a denominator that is known to sit in
// [1.6, 2.5] -- no scaling, no fix-up: 38 and 33 instructions.  Same special values as the library (log: 0 -> -inf,
// negative -> NaN, inf -> inf; exp: -inf -> 0, overflow -> inf; NaN -> NaN; subnormal arguments and results handled by
// v_frexp / v_ldexp).  Held to the zoo's float64 tolerance (2e-14 of the population against the reference's numbers;
// the two routines themselves measure <= 1 ulp against libm: tests/test_gpu_zoo.py::test_zoo_f64_log_exp_are_within_one_ulp).
// n / d for d in [1.5, 2.6]: rcp seed (~2^-26), two Newton steps on the reciprocal, one correction of the quotient
__device__ __forceinline__ double div_safe_range(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    const double q = n * r;
    return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}
__device__ __forceinline__ double log_f64(double v) {
    // v = 2^k (1 + f), sqrt(2)/2 <= 1 + f < sqrt(2);  s = f / (2 + f);  log(1 + f) = f - (f^2/2 - s (f^2/2 + R(s^2)))
    double m = __builtin_amdgcn_frexp_mant(v);                 // [0.5, 1) (v itself for 0, inf, NaN)
    int k = __builtin_amdgcn_frexp_exp(v);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    k = low ? k - 1 : k;
    const double f = m - 1.0;
    const double s = div_safe_range(f, 2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                                         2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    const double r = __builtin_fma(dk, 6.93147180369123816490e-01,
                                   f - (hfsq - __builtin_fma(s, hfsq + R, dk * 1.90821492927058770002e-10)));
    const double inf = __builtin_huge_val();
    // (v == inf: frexp hands inf back, f = inf, the quotient NaN)
    return (v > 0.0) ? ((v == inf) ? inf : r) : ((v == 0.0) ? -inf : __builtin_nan(""));
}
__device__ __forceinline__ double exp_f64(double y) {
    // y = k ln2 + r, |r| <= ln2 / 2;  exp(r) = 1 + 2 r / (R(r) - r),  R(r) = 2 - c + ...:  1 - ((lo - r c / (2 - c)) - hi)
    const double yc = __builtin_fmin(__builtin_fmax(y, -746.0), 710.0);
    const double kf = __builtin_rint(yc * 1.44269504088896338700e+00);
    const double hi = __builtin_fma(-kf, 6.93147180369123816490e-01, yc);
    const double lo = kf * 1.90821492927058770002e-10;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, 4.13813679705723846039e-08,
                                           -1.65339022054652515390e-06), 6.61375632143793436117e-05), -2.77777777770155933842e-03),
                                           1.66666666666666019037e-01);
    const double e = 1.0 - ((lo - div_safe_range(r * c, 2.0 - c)) - hi);
    const double res = __builtin_amdgcn_ldexp(e, (int)kf);
    return (y != y) ? y : res;
}

template <typename W>
__device__ __forceinline__ W int_pow(W x, int e) {        // e in 1 .. 4
    const W x2 = x * x;
    return e == 1 ? x : e == 2 ? x2 : e == 3 ? x2 * x : x2 * x2;
}

// The five growth functions as the reference writes them (growth_models.py:208-261): each ends in
// np.maximum(0, np.random.lognormal(mu, sigma)) = max(0, exp(mu + sigma z)), a round trip through log and exp (also at
// sigma = 0).  The float64 layout's cold path (zoo_draw_f64 below: far stocks / far results).  x ** e with e one of 1 .. 4
// (make_growth: ipow; the reference's defaults theta = q = 3) by multiplication: <= 1 ulp from the correctly rounded np.power.
// RECOMPUTE: P.r changed on the device (fishing-v10 drift) -> logA / B are evaluated here
struct MathF64 {
    typedef double W;
    static constexpr bool kIntPow = true;
    static __device__ __forceinline__ double log(double v) { return log_f64(v); }
    static __device__ __forceinline__ double exp(double v) { return exp_f64(v); }
    static __device__ __forceinline__ double pow(double v, double e) { return exp(e * log(v)); }
};
template <int KIND, bool RECOMPUTE, typename T>
__device__ __forceinline__ T zoo_draw_round_trip(int kind_rt, T x_in, T z_in, const GrowthT<T>& P) {
    static_assert(sizeof(T) == 8, "the float64 parity layout's");
    typedef MathF64 M;
    typedef double W;
    const W inf = (W)__builtin_huge_val();
    const W x = (W)x_in, z = (W)z_in;
    const int kind = (KIND >= 0) ? KIND : kind_rt;      // compile-time kind folds the switch away
    W mu;
    switch (kind) {
        case FISHING_KIND_ALLEN:          // :208-217
            mu = M::log(x) + (W)P.r * ((W)1 - x / (W)P.K) * ((W)1 - (W)P.C) / (W)P.K;
            break;
        case FISHING_KIND_MYERS: {        // :247-255   (log(A), A = r + 1, comes from the host)
            const W lx = M::log(x);
            const W xt = (M::kIntPow && P.ipow) ? int_pow<W>(x, P.ipow) : M::exp((W)P.theta * lx);   // x**theta (wave-uniform choice)
            mu = (W)P.logA + (W)P.theta * lx - M::log((W)1 + xt / (W)P.M);
            break;
        }
        case FISHING_KIND_MAY: {          // :229-242   (b**q comes from the host)
            const W xq = (M::kIntPow && P.ipow) ? int_pow<W>(x, P.ipow) : M::pow(x, (W)P.q);
            const W exp_mu = x + x * (W)P.r * ((W)1 - x / (W)P.M) - (W)P.a * xq / (xq + (W)P.bq);
            mu = M::log(exp_mu);
            break;
        }
        case FISHING_KIND_RICKER:         // :258-261
            mu = M::log(x) + (W)P.r * ((W)1 - x / (W)P.K);
            break;
        default: {                        // Beverton-Holt :220-226 (np.clip(., 0, inf): NaN passes)
            const W xc = (x < (W)0) ? (W)0 : ((x > inf) ? inf : x);
            W logA = (W)P.logA, B = (W)P.B;
            if (RECOMPUTE) {
                const W rc = ((W)P.r < (W)0) ? (W)0 : (W)P.r;
                const W Kc = ((W)P.K < (W)0) ? (W)0 : (W)P.K;
                logA = M::log(rc + (W)1);
                B = Kc / rc;
            }
            mu = logA + M::log(xc) - M::log((W)1 + xc / B);
            break;
        }
    }
    const W g = M::exp(mu + (W)P.sigma * z);
    return (T)((g > (W)0) ? g : ((g != g) ? g : (W)0));    // np.maximum(0, g)
}

// ---- the float64 parity layout: the algebraic form on the < 1-ulp exp_f64
// Parity is tolerance-based (2e-14 of the population: the reference's libm log / exp are not reproducible bit for bit on the
// device anyway).  Same special values as the round trip (tests/test_gpu_zoo.py::test_zoo_special_values_follow_the_reference
// holds both layouts to them).
// FISHING_ZOO_F64_FAR: where the REFERENCE's round trip itself loses more than the parity bar, follow it (zoo_draw_f64).  A
// per-translation-unit setting: 1 in fishing_aux.hip (population_draw hands x' out itself), 0 in fishing_step.hip /
// fishing_rollout.hip (their outputs cannot carry the difference: fishing_step.hip, top).  The inline device templates below
// therefore have different bodies per translation unit -- sound only because every unit is its own device link (build.py:
// -fno-gpu-rdc; no device symbol crosses units).
#ifndef FISHING_ZOO_F64_FAR
#define FISHING_ZOO_F64_FAR 1
#endif
// n / d to <= 1 ulp: v_rcp_f64 seed (~2^-26), two Newton steps on the reciprocal, one correction of the quotient with an
// exact residual (8 instructions; the IEEE division's v_div_scale / v_div_fmas / v_div_fixup frame is ~14 and issues no
// faster).  Denominators outside the comfortable range (zeros, infinities, NaN included) take the IEEE division.
__device__ __forceinline__ double div_f64(double n, double d) {
    const double ad = __builtin_fabs(d);
    if (!(ad > 0x1p-500 && ad < 0x1p500)) return n / d;
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    const double q = n * r;
    return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}
__device__ __forceinline__ double pow_f64(double x, double e, int ipow) {
    if (ipow == 0) return exp_f64(e * log_f64(x));      // wave-uniform; pow(0, e > 0) = exp(-inf) = 0
    return int_pow<double>(x, ipow);
}
// x / M as the REFERENCE rounds it (an IEEE division) from the host's correctly rounded reciprocal: one Newton step on the
// quotient -- q0 = x * (1 / M) is within an ulp, its residual x - q0 M is exact in an fma, and the corrected quotient is the
// correctly rounded one (Markstein).  May's exp_mu = x + x r (1 - x / M) - a x^q / (x^q + b^q) is a difference of terms up to 60 times
// its own size when the stock stands above M: there the ulp a bare reciprocal multiply may be off came back as 2.5e-13 of the
// population (found by the random parameter sweep against the oracle, tests/test_gpu_zoo.py).  Non-finite cases (x = inf, M = 0
// or inf) fall back to the plain product, which has the division's value there.
__device__ __forceinline__ double div_by_reciprocal_f64(const double x, const double M, const double invM) {
    const double q0 = x * invM;
    const double q = __builtin_fma(__builtin_fma(-q0, M, x), invM, q0);
    return (q == q) ? q : q0;
}
template <int KIND, bool RECOMPUTE, typename T>
__device__ __forceinline__ void zoo_pre_g_f64(double x, double sz, const GrowthT<T>& P, double& pre, double& g) {
    static_assert(KIND >= 0 && KIND < FISHING_N_KINDS, "a compile-time kind");
    if constexpr (KIND == FISHING_KIND_ALLEN) {                 // :208-217: exp(log x + r (1 - x/K)(1 - C)/K + sz)
        pre = x;
        g = __builtin_fma(P.gc, __builtin_fma(-x, P.invK, 1.0), sz);
    } else if constexpr (KIND == FISHING_KIND_RICKER) {         // :258-261
        pre = x;
        g = __builtin_fma(P.r, __builtin_fma(-x, P.invK, 1.0), sz);
    } else if constexpr (KIND == FISHING_KIND_MYERS) {          // :247-255: A x^theta / (1 + x^theta / M)
        const double xt = pow_f64(x, P.theta, P.ipow);
        pre = div_f64(P.A * xt, __builtin_fma(xt, P.invM, 1.0));
        g = sz;
    } else if constexpr (KIND == FISHING_KIND_MAY) {            // :229-242: exp(log(exp_mu)); the log of a negative number is NaN
        const double xq = pow_f64(x, P.q, P.ipow);
        const double exp_mu = (x + x * P.r * (1.0 - div_by_reciprocal_f64(x, P.M, P.invM))) - div_f64(P.a * xq, xq + P.bq);
        pre = (exp_mu < 0.0) ? __builtin_nan("") : exp_mu;
        g = sz;
    } else {                                                    // Beverton-Holt :220-226: A x / (1 + x / B)
        const double xc = (x < 0.0) ? 0.0 : x;
        double A = P.A, invB = P.invB;
        if (RECOMPUTE) {              // fishing-v10: r drifts per env; invK = 1 / clip(K, 0, inf) from the host
            const double rc = (P.r < 0.0) ? 0.0 : P.r;
            A = rc + 1.0;
            invB = rc * P.invK;
        }
        pre = div_f64(A * xc, __builtin_fma(xc, invB, 1.0));
        g = sz;
    }
}
template <int KIND, bool RECOMPUTE, typename T>
__device__ __forceinline__ double zoo_draw_f64(int kind_rt, double x, double z, const GrowthT<T>& P) {
    const double sz = P.sigma * z;
    double pre = 0.0, g = 0.0;
    if constexpr (KIND >= 0) {
        zoo_pre_g_f64<KIND, RECOMPUTE, T>(x, sz, P, pre, g);
    } else {            // run-time kind (general kernel, population_draw sweeps): wave-uniform switch
        switch (kind_rt) {
            case FISHING_KIND_ALLEN: zoo_pre_g_f64<FISHING_KIND_ALLEN, false, T>(x, sz, P, pre, g); break;
            case FISHING_KIND_MYERS: zoo_pre_g_f64<FISHING_KIND_MYERS, false, T>(x, sz, P, pre, g); break;
            case FISHING_KIND_MAY: zoo_pre_g_f64<FISHING_KIND_MAY, false, T>(x, sz, P, pre, g); break;
            case FISHING_KIND_RICKER: zoo_pre_g_f64<FISHING_KIND_RICKER, false, T>(x, sz, P, pre, g); break;
            default: zoo_pre_g_f64<FISHING_KIND_BEVERTON_HOLT, RECOMPUTE, T>(x, sz, P, pre, g); break;
        }
    }
    const double res = pre * exp_f64(g);
    // Where the REFERENCE's round trip itself loses more than the parity bar, follow it.  exp(mu) carries the rounding of its
    // argument: half an ulp of |mu| -- 7e-15 of the result at |mu| = 64, 2.8e-14 at 207 (Myers at x = 1e-30: 2 e-90 exp(sigma z)
    // comes back 2.0e-14 off the exact value in NumPy) -- while this form is within a few ulp of exact.  A stock outside
    // [2^-30, 2^30] or a result outside [2^-92, 2^92] (terms of mu beyond ~21 theta / |mu| beyond ~64) is therefore evaluated
    // the reference's way; zeros, infinities and NaNs are the same in both forms and stay here.  Compiled into population_draw
    // (fishing_aux.hip: BMSY sweeps, the module-level growth functions, the special-value tests), which hands x' out itself; the
    // step / rollout translation units leave it out -- obs = x' / K - 1 cannot carry the difference (fishing_step.hip, top).
    const double inf = __builtin_huge_val();
    [[maybe_unused]] const bool far = (x > 0.0 && x < 0x1p-30) || (x > 0x1p30 && x < inf) || (res > 0.0 && res < 0x1p-92) || (res > 0x1p92 && res < inf);
    if constexpr (FISHING_ZOO_F64_FAR != 0) {    // (0 in the step / rollout translation units, 1 in population_draw's)
        if (__builtin_expect(far, 0)) return zoo_draw_round_trip<KIND, RECOMPUTE, T>(kind_rt, x, z, P);
    }
    return (res > 0.0) ? res : ((res != res) ? res : 0.0);      // np.maximum(0, .)
}

// ---- the float32 layout: the same algebraic form entirely in float32 on the hardware exp (v_exp_f32)
// n / d to <= 1 ulp: v_rcp_f32 + one correction of the quotient (4 instructions; the IEEE float32 division is ~10).
// Denominators outside the comfortable range (zeros, infinities, NaN included) take the IEEE division.
__device__ __forceinline__ float div_f32(float n, float d) {
    const float ad = __builtin_fabsf(d);
    if (!(ad > 0x1p-60f && ad < 0x1p60f)) return n / d;
    const float r = __builtin_amdgcn_rcpf(d);
    const float q = n * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
}
__device__ __forceinline__ float pow_f32(float x, float e, int ipow) {
    if (ipow == 0) return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(x));      // wave-uniform
    const float x2 = x * x;
    return ipow == 1 ? x : ipow == 2 ? x2 : ipow == 3 ? x2 * x : x2 * x2;
}
template <int KIND, bool RECOMPUTE, typename T>
__device__ __forceinline__ void zoo_pre_g_f32(float x, float sz, const GrowthT<T>& P, float& pre, float& g) {
    static_assert(KIND >= 0 && KIND < FISHING_N_KINDS, "a compile-time kind");
    if constexpr (KIND == FISHING_KIND_ALLEN) {
        pre = x;
        g = __builtin_fmaf((float)P.gc, __builtin_fmaf(-x, (float)P.invK, 1.0f), sz);
    } else if constexpr (KIND == FISHING_KIND_RICKER) {
        pre = x;
        g = __builtin_fmaf((float)P.r, __builtin_fmaf(-x, (float)P.invK, 1.0f), sz);
    } else if constexpr (KIND == FISHING_KIND_MYERS) {
        const float xt = pow_f32(x, (float)P.theta, P.ipow);
        pre = div_f32((float)P.A * xt, __builtin_fmaf(xt, (float)P.invM, 1.0f));
        g = sz;
    } else if constexpr (KIND == FISHING_KIND_MAY) {
        const float xq = pow_f32(x, (float)P.q, P.ipow);
        const float exp_mu = __builtin_fmaf(x * (float)P.r, __builtin_fmaf(-x, (float)P.invM, 1.0f), x) -
                             div_f32((float)P.a * xq, xq + (float)P.bq);
        pre = (exp_mu < 0.0f) ? __builtin_nanf("") : exp_mu;
        g = sz;
    } else {
        const float xc = (x < 0.0f) ? 0.0f : x;
        float A = (float)P.A, invB = (float)P.invB;
        if (RECOMPUTE) {
            const float rc = ((float)P.r < 0.0f) ? 0.0f : (float)P.r;
            A = rc + 1.0f;
            invB = rc * (float)P.invK;
        }
        pre = div_f32(A * xc, __builtin_fmaf(xc, invB, 1.0f));
        g = sz;
    }
}
__device__ __forceinline__ float zoo_finish_f32(float pre, float g) {
    const float res = pre * __builtin_amdgcn_exp2f(g * 1.44269504088896341f);
    return (res > 0.0f) ? res : ((res != res) ? res : 0.0f);
}
template <int KIND, bool RECOMPUTE, typename T>
__device__ __forceinline__ float zoo_draw_f32(int kind_rt, float x, float z, const GrowthT<T>& P) {
    const float sz = (float)P.sigma * z;
    float pre = 0.0f, g = 0.0f;
    if constexpr (KIND >= 0) {
        zoo_pre_g_f32<KIND, RECOMPUTE, T>(x, sz, P, pre, g);
    } else {
        switch (kind_rt) {
            case FISHING_KIND_ALLEN: zoo_pre_g_f32<FISHING_KIND_ALLEN, false, T>(x, sz, P, pre, g); break;
            case FISHING_KIND_MYERS: zoo_pre_g_f32<FISHING_KIND_MYERS, false, T>(x, sz, P, pre, g); break;
            case FISHING_KIND_MAY: zoo_pre_g_f32<FISHING_KIND_MAY, false, T>(x, sz, P, pre, g); break;
            case FISHING_KIND_RICKER: zoo_pre_g_f32<FISHING_KIND_RICKER, false, T>(x, sz, P, pre, g); break;
            default: zoo_pre_g_f32<FISHING_KIND_BEVERTON_HOLT, RECOMPUTE, T>(x, sz, P, pre, g); break;
        }
    }
    return zoo_finish_f32(pre, g);
}

template <typename T, int KIND = -1, bool RECOMPUTE = false>
__device__ __forceinline__ T zoo_population_draw(int kind_rt, T x, T z, const GrowthT<T>& P) {
    if constexpr (sizeof(T) == 4) return (T)zoo_draw_f32<KIND, RECOMPUTE, T>(kind_rt, (float)x, (float)z, P);
    else return (T)zoo_draw_f64<KIND, RECOMPUTE, T>(kind_rt, (double)x, (double)z, P);
}

// ---------------------------------------------------------------- fishing-v11: one division and one exp per env, whatever its kind
// Every growth function is x' = max(0, pre_k(x) * exp(g_k(x, z))), and what differs by kind is cheap: g is sigma_k z plus, for
// Allen and Ricker, c_k (1 - x / K_k); pre is x itself (Allen, Ricker), one quotient n_k / d_k (Beverton-Holt, Myers) or a cubic
// minus that quotient (May).  So a lane evaluates ONE division and ONE exp per env at full lane occupancy -- no ballots, no
// divergent passes -- once it has its env's coefficients: from a 5 x 8 table in LDS (zoo_draw_lut_tile, the step / rollout
// kernels) or, on the per-env-sigma path, by selects over the wave-uniform parameter sets (zoo_draw_select_one).  Both run the
// operations of zoo_pre_g_f32 / zoo_pre_g_f64 of the env's kind on the same operands: the same bits.  (An earlier
// regroup-by-kind form and the measurements of all three: profiles/NOTES_r01_r05.md, profiles/r05_v11_forms.jsonl.)
template <typename W>
struct ZooSelectMath;
template <>
struct ZooSelectMath<float> {
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static __device__ __forceinline__ float div(float n, float d) { return div_f32(n, d); }
    static __device__ __forceinline__ float pow(float x, float e, int ipow) { return pow_f32(x, e, ipow); }
    static __device__ __forceinline__ float nan() { return __builtin_nanf(""); }
};
template <>
struct ZooSelectMath<double> {
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static __device__ __forceinline__ double div(double n, double d) { return div_f64(n, d); }
    static __device__ __forceinline__ double pow(double x, double e, int ipow) { return pow_f64(x, e, ipow); }
    static __device__ __forceinline__ double nan() { return __builtin_nan(""); }
};
// `kind` in [0, FISHING_N_KINDS) (the callers map anything else to Beverton-Holt)
template <typename T>
__device__ __forceinline__ T zoo_draw_select_one(const int kind, const T x_in, const T z_in, const GrowthT<T> (&zoo)[FISHING_N_KINDS],
                                                  bool& far, const bool own_sigma = false, const T sigma_env = (T)0) {
    typedef T W;
    typedef ZooSelectMath<W> M;
    const GrowthT<T>& PA = zoo[FISHING_KIND_ALLEN];
    const GrowthT<T>& PB = zoo[FISHING_KIND_BEVERTON_HOLT];
    const GrowthT<T>& PM = zoo[FISHING_KIND_MYERS];
    const GrowthT<T>& PY = zoo[FISHING_KIND_MAY];
    const GrowthT<T>& PR = zoo[FISHING_KIND_RICKER];
    const bool isA = kind == FISHING_KIND_ALLEN, isM = kind == FISHING_KIND_MYERS, isY = kind == FISHING_KIND_MAY,
               isR = kind == FISHING_KIND_RICKER;
    const W x = (W)x_in;
    // sigma z of the env's own growth function
    // (`own_sigma`, compile-time at every call site: the env's own noise scale from the caller's sigma array instead)
    const W sg = own_sigma ? (W)sigma_env : (isA ? (W)PA.sigma : isM ? (W)PM.sigma : isY ? (W)PY.sigma : isR ? (W)PR.sigma : (W)PB.sigma);
    const W sz = sg * (W)z_in;
    // the exponent: Allen / Ricker carry their density dependence there, the others only the noise
    const W gA = M::fma((W)PA.gc, M::fma(-x, (W)PA.invK, (W)1), sz);
    const W gR = M::fma((W)PR.r, M::fma(-x, (W)PR.invK, (W)1), sz);
    const W g = isA ? gA : (isR ? gR : sz);
    // the quotient: Beverton-Holt A x / (1 + x / B), Myers A x^theta / (1 + x^theta / M), May's a x^q / (x^q + b^q)
    const W xc = (x < (W)0) ? (W)0 : x;
    const W xt = M::pow(x, (W)PM.theta, PM.ipow);
    const W xq = M::pow(x, (W)PY.q, PY.ipow);
    const W nB = (W)PB.A * xc, dB = M::fma(xc, (W)PB.invB, (W)1);
    const W nM = (W)PM.A * xt, dM = M::fma(xt, (W)PM.invM, (W)1);
    const W nY = (W)PY.a * xq, dY = xq + (W)PY.bq;
    const W n = isM ? nM : (isY ? nY : nB);
    const W d = isM ? dM : (isY ? dY : dB);
    const W q = M::div(n, d);
    // May: exp(log(exp_mu)), the log of a negative number being NaN
    W emu;
    if constexpr (sizeof(T) == 4) emu = M::fma(x * (W)PY.r, M::fma(-x, (W)PY.invM, (W)1), x) - q;
    else emu = (x + x * (W)PY.r * ((W)1 - (W)div_by_reciprocal_f64((double)x, PY.M, PY.invM))) - q;      // (x / M to the reference's rounding)
    const W preY = (emu < (W)0) ? M::nan() : emu;
    const W pre = (isA || isR) ? x : (isY ? preY : q);
    if constexpr (sizeof(T) == 4) {
        return (T)zoo_finish_f32(pre, g);
    } else {
        const double res = pre * exp_f64(g);
        const double inf = __builtin_huge_val();
        // (zoo_draw_f64: where the reference's own round trip loses more than the parity bar, the caller follows it)
        far = (x > 0.0 && x < 0x1p-30) || (x > 0x1p30 && x < inf) || (res > 0.0 && res < 0x1p-92) || (res > 0x1p92 && res < inf);
        return (res > 0.0) ? res : ((res != res) ? res : 0.0);
    }
}

// ---------------------------------------------------------------- fishing-v11: the growth function's coefficients from an LDS table
// The five kinds' coefficients sit in a 5 x 8 table in LDS, written once per workgroup (by its
// first wave, while the tile's loads are in flight; once per launch in the rollout kernels), and an env fetches ITS row with two
// 16-byte LDS reads (four in float64) -- no selects of constants, no ballots, no passes:
//     row k = { sigma, cg, cK, c1, c2, c3, cr, M (May, float64 layout) }
//     g   = cg (1 - x cK) + sigma z          Allen: cg = r (1 - C) / K, cK = 1 / K;  Ricker: cg = r, cK = 1 / K;  others cg = 0
//     q   = c1 w / (c2 + c3 w)               w = clip(x) (Beverton-Holt: c1 = A, c2 = 1, c3 = 1 / B) or x ** p (Myers: A, 1, 1 / M;
//                                            May: a, b ** q, 1);  Allen / Ricker: c1 = c3 = 0, c2 = 1 -> q = 0, unused
//     pre = x (Allen, Ricker) | q (Beverton-Holt, Myers) | x + x cr (1 - x cK) - q, NaN below zero (May: cr = r, cK = 1 / M)
//     x'  = max(0, pre exp(g))
// -- the operations of zoo_pre_g_f32 / zoo_pre_g_f64 of the env's kind on the same operands (the zero coefficients contribute
// exact zeros; fma(w, 1, b ** q) IS w + b ** q), hence the same bits.
constexpr int kZooLutRow = 8;
constexpr int kZooLutSize = FISHING_N_KINDS * kZooLutRow;
// by the lanes 0 of whichever waves call it (uniform values: every caller writes the same table); the caller synchronises
// (host and device: a launch may bring the table ready-made in its arguments -- the same conversions either way)
template <typename T>
__host__ __device__ __forceinline__ void zoo_lut_rows(const GrowthT<T> (&zoo)[FISHING_N_KINDS], T (&out)[kZooLutSize]) {
    const GrowthT<T>& PA = zoo[FISHING_KIND_ALLEN];
    const GrowthT<T>& PB = zoo[FISHING_KIND_BEVERTON_HOLT];
    const GrowthT<T>& PM = zoo[FISHING_KIND_MYERS];
    const GrowthT<T>& PY = zoo[FISHING_KIND_MAY];
    const GrowthT<T>& PR = zoo[FISHING_KIND_RICKER];
    const T rows[FISHING_N_KINDS][kZooLutRow] = {
        {(T)PA.sigma, (T)PA.gc, (T)PA.invK, (T)0, (T)1, (T)0, (T)0, (T)0},           // FISHING_KIND_ALLEN
        {(T)PB.sigma, (T)0, (T)0, (T)PB.A, (T)1, (T)PB.invB, (T)0, (T)0},            // FISHING_KIND_BEVERTON_HOLT
        {(T)PM.sigma, (T)0, (T)0, (T)PM.A, (T)1, (T)PM.invM, (T)0, (T)0},            // FISHING_KIND_MYERS
        {(T)PY.sigma, (T)0, (T)PY.invM, (T)PY.a, (T)PY.bq, (T)1, (T)PY.r, (T)PY.M},  // FISHING_KIND_MAY (M: the float64 layout's x / M)
        {(T)PR.sigma, (T)PR.r, (T)PR.invK, (T)0, (T)1, (T)0, (T)0, (T)0}};           // FISHING_KIND_RICKER
#pragma unroll
    for (int k = 0; k < FISHING_N_KINDS; ++k)
#pragma unroll
        for (int f = 0; f < kZooLutRow; ++f) out[k * kZooLutRow + f] = rows[k][f];
}
template <typename T>
__device__ __forceinline__ void zoo_lut_fill(T* __restrict__ lut, const GrowthT<T> (&zoo)[FISHING_N_KINDS]) {
    if ((threadIdx.x & (kWave - 1)) == 0) {
        T rows[kZooLutSize];
        zoo_lut_rows<T>(zoo, rows);
#pragma unroll
        for (int k = 0; k < kZooLutSize; ++k) lut[k] = rows[k];
    }
}
template <typename T>
__device__ __forceinline__ T zoo_draw_lut_one(const int kind, const T x, const T z_in, const T* __restrict__ lut, const int ipowM,
                                              const int ipowY, const T thetaM, const T qY, bool& far) {
    typedef ZooSelectMath<T> M;
    struct alignas(16) Quad { T v[4]; };     // (the tables are declared alignas(16); float64: two 16-byte reads per Quad)
    const Quad lo = *reinterpret_cast<const Quad*>(lut + kind * kZooLutRow);
    const Quad hi = *reinterpret_cast<const Quad*>(lut + kind * kZooLutRow + 4);
    const T sg = lo.v[0], cg = lo.v[1], cK = lo.v[2], c1 = lo.v[3], c2 = hi.v[0], c3 = hi.v[1], cr = hi.v[2];
    const bool isB = kind == FISHING_KIND_BEVERTON_HOLT, isY = kind == FISHING_KIND_MAY;
    const bool isAR = kind == FISHING_KIND_ALLEN || kind == FISHING_KIND_RICKER;
    const T sz = sg * z_in;
    const T lin = M::fma(-x, cK, (T)1);
    const T g = M::fma(cg, lin, sz);
    const T xc = (x < (T)0) ? (T)0 : x;
    // x ** theta (Myers) / x ** q (May): wave-uniform choices (equal small integer powers -- the defaults -- share the product)
    const T u = (ipowM == ipowY && ipowM != 0) ? int_pow<T>(x, ipowM) : (isY ? M::pow(x, qY, ipowY) : M::pow(x, thetaM, ipowM));
    const T w = isB ? xc : u;
    const T q = M::div(c1 * w, M::fma(w, c3, c2));
    T emu;
    if constexpr (sizeof(T) == 4) emu = M::fma(x * cr, lin, x) - q;
    else emu = (x + x * cr * ((T)1 - (T)div_by_reciprocal_f64((double)x, (double)hi.v[3], (double)cK))) - q;      // (zoo_pre_g_f64's May: separately
                                                                                                            // rounded operations, x / M included)
    const T preY = (emu < (T)0) ? M::nan() : emu;
    const T pre = isAR ? x : (isY ? preY : q);
    if constexpr (sizeof(T) == 4) {
        return (T)zoo_finish_f32(pre, g);
    } else {
        const double res = pre * exp_f64(g);
        const double inf = __builtin_huge_val();
        far = (x > 0.0 && x < 0x1p-30) || (x > 0x1p30 && x < inf) || (res > 0.0 && res < 0x1p-92) || (res > 0x1p92 && res < inf);
        return (res > 0.0) ? res : ((res != res) ? res : 0.0);
    }
}
// the N envs of a lane (4; 2 in the float64 layout's two-envs-per-thread forms)
template <typename T, int N>
__device__ __forceinline__ void zoo_draw_lut_tile(const int (&kind)[N], const T (&x)[N], const T (&z)[N],
                                                  const GrowthT<T> (&zoo)[FISHING_N_KINDS], const T* __restrict__ lut, T (&out)[N]) {
    const int ipowM = zoo[FISHING_KIND_MYERS].ipow, ipowY = zoo[FISHING_KIND_MAY].ipow;
    bool far[N];
    bool any_far = false;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        bool f = false;
        const T v = zoo_draw_lut_one<T>(kind[j] >= 0 ? kind[j] : FISHING_KIND_BEVERTON_HOLT, x[j], z[j], lut, ipowM, ipowY,
                                        (T)zoo[FISHING_KIND_MYERS].theta, (T)zoo[FISHING_KIND_MAY].q, f);
        out[j] = (kind[j] >= 0) ? v : out[j];
        far[j] = f && kind[j] >= 0;
        any_far |= far[j];
        // float64: one env after the other -- interleaved, the evaluations' temporaries (register pairs) push the fused kernels
        // into scratch; float32 keeps the interleaving's ILP
        if constexpr (sizeof(T) == 8) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (sizeof(T) == 8 && FISHING_ZOO_F64_FAR) {
        if (__builtin_expect(any_far, 0)) {         // (zoo_draw_f64: follow the reference's round trip there; ONE copy, real loops)
            for (int j = 0; j < N; ++j) {
                asm volatile("" : "+s"(j));
                bool fj = far[0];
                double xj = x[0], zj = z[0];
                int kj = kind[0];
#pragma unroll
                for (int k = 1; k < N; ++k) {
                    fj = (j == k) ? far[k] : fj;
                    xj = (j == k) ? (double)x[k] : xj;
                    zj = (j == k) ? (double)z[k] : zj;
                    kj = (j == k) ? kind[k] : kj;
                }
                if (!fj) continue;
                // (one compile-time kind per case: the run-time-kind instantiation needs 30 more registers than any of these)
                double r;
                switch (kj) {
                    case FISHING_KIND_ALLEN: r = zoo_draw_round_trip<FISHING_KIND_ALLEN, false, T>(kj, xj, zj, zoo[FISHING_KIND_ALLEN]); break;
                    case FISHING_KIND_MYERS: r = zoo_draw_round_trip<FISHING_KIND_MYERS, false, T>(kj, xj, zj, zoo[FISHING_KIND_MYERS]); break;
                    case FISHING_KIND_MAY: r = zoo_draw_round_trip<FISHING_KIND_MAY, false, T>(kj, xj, zj, zoo[FISHING_KIND_MAY]); break;
                    case FISHING_KIND_RICKER: r = zoo_draw_round_trip<FISHING_KIND_RICKER, false, T>(kj, xj, zj, zoo[FISHING_KIND_RICKER]); break;
                    default: r = zoo_draw_round_trip<FISHING_KIND_BEVERTON_HOLT, false, T>(kj, xj, zj, zoo[FISHING_KIND_BEVERTON_HOLT]); break;
                }
#pragma unroll
                for (int k = 0; k < N; ++k) out[k] = (j == k) ? (T)r : out[k];
            }
        }
    }
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
template <typename T>
__device__ __forceinline__ T div_K(T x, T K, const DivK& d);
// (a run-time `pow2` is wave-uniform: a real branch -- the empty asm keeps the compiler from evaluating the 10 / 25-instruction
// division speculatively and selecting afterwards; a compile-time `pow2` folds the branch away)
template <>
__device__ __forceinline__ float div_K<float>(float x, float K, const DivK& d) {
    if (d.pow2) return x * d.inv_f;
    asm volatile("");
    return x / K;
}
template <>
__device__ __forceinline__ double div_K<double>(double x, double K, const DivK& d) {
    if (d.pow2) return x * d.inv_d;
    asm volatile("");
    return x / K;
}

// harvest_draw (base_fishing_env.py:112-119): h = min(x, quota), then x = max(x - h, 0.0).  That max is the identity on
// every input: h is x itself or a quota BELOW x, so d = x - h is +0 (x - x), positive (x > quota: the difference of two
// distinct values never rounds below +0) or NaN (inf - inf, a NaN operand) -- and Python's max(d, 0.0) returns d in all
// three cases (`0.0 > d` is false).  It is therefore not evaluated: one compare / select pair per env-step less in the
// VALU-bound fused kernels (tests/test_gpu_parity.py holds the kernels to the oracle's literal form, NaN / inf actions
// and states included).
template <typename T>
__device__ __forceinline__ T stock_after_harvest(T x, T h) { return x - h; }

// step() with a zoo growth function: quota / obs maps use the env's K (K_obs), the growth its
// own parameter set (self.params in the reference).  `dk`: K_obs is a power of two (div_K below).
template <typename T, int KIND = -1, bool RECOMPUTE = false>
__device__ __forceinline__ void env_step_zoo(T obs, int32_t t, T quota, T z, int kind, const GrowthT<T>& P,
                                             T K_obs, int32_t Tmax, T& obs_next, T& reward, bool& done,
                                             int32_t& t_next, const DivK& dk = DivK{false, 0.0f, 0.0}) {
    T x = (obs + (T)1) * K_obs;
    const T h = (quota < x) ? quota : x;
    x = stock_after_harvest<T>(x, h);
    x = zoo_population_draw<T, KIND, RECOMPUTE>(kind, x, z, P);
    obs_next = div_K<T>(x, K_obs, dk) - (T)1;
    reward = ((T)0 > h) ? (T)0 : h;
    t_next = t + 1;
    done = (t_next > Tmax) || (x <= (T)0);
}

// fishing-v11 with a per-env noise scale (the caller's sigma array): one env, its growth function chosen per lane.  The select
// form with sigma_env in place of the kinds' own sigma: same bits as env_step_zoo<T, -1> on a copy of zoo[kind] with that sigma,
// without the per-lane index into the kernel-argument array (760 bytes of scratch in the float64 fused kernel).
template <typename T>
__device__ __forceinline__ void env_step_zoo_mixed(T obs, int32_t t, T quota, T z, int kind, const GrowthT<T> (&zoo)[FISHING_N_KINDS],
                                                   T sigma_env, T K_obs, int32_t Tmax, T& obs_next, T& reward, bool& done,
                                                   int32_t& t_next, const DivK& dk = DivK{false, 0.0f, 0.0}) {
    const int kk = (kind >= 0 && kind < FISHING_N_KINDS) ? kind : FISHING_KIND_BEVERTON_HOLT;
    {
        T x = (obs + (T)1) * K_obs;
        const T h = (quota < x) ? quota : x;
        x = stock_after_harvest<T>(x, h);
        bool far = false;
        T xn = zoo_draw_select_one<T>(kk, x, z, zoo, far, true, sigma_env);
        if constexpr (sizeof(T) == 8 && FISHING_ZOO_F64_FAR) {
            if (__builtin_expect(far, 0)) {         // (zoo_draw_select_tile: the reference's own round trip, one copy, a real loop)
                for (int k = 0; k < FISHING_N_KINDS; ++k) {
                    asm volatile("" : "+s"(k));
                    if (kk == k) {
                        GrowthT<T> P = zoo[k];
                        P.sigma = sigma_env;
                        xn = zoo_draw_round_trip<-1, false, T>(k, x, z, P);
                    }
                }
            }
        }
        obs_next = div_K<T>(xn, K_obs, dk) - (T)1;
        reward = ((T)0 > h) ? (T)0 : h;
        t_next = t + 1;
        done = (t_next > Tmax) || (xn <= (T)0);
    }
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
    x = stock_after_harvest<T>(x, h);         // max(x - h, 0.0)      :118 (the max is the identity: above)
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
// fishing-v11 (growth_models.py:187,200: np.random.choice(models)): a new growth function for the finished envs of one thread's
// 4-env tile.  ONE Philox2x32-10 block per env QUAD on the reset streams -- half the multiplies of a Philox4x32 block, on a
// workload where nearly every quad holds a finished env every step (random policy: mean episode length 1.47) -- and four
// 16-bit draws from its two words: env 4q + j takes half j (w0 low, w0 high, w1 low, w1 high),
// index = (half * n_models) >> 16.  Non-uniformity of np.random.choice: the 65536 halves split over n_models buckets whose
// sizes differ by at most one -> every model's probability is within 2^-16 = 1.5e-5 (absolute) of 1 / n_models (n = 5:
// 13108 / 13107 x 4).  Block: c0 = quad[31:0], c1 = counter[30:0] | reset-stream bit 31 (^ quad[63:32] * 0xC2B2AE35), key =
// fishing-v4's param_key ^ a tag ^ counter[62:31] * 0x9E3779B1 -- injective in (quad, counter, stream) while quad < 2^32
// (2^34 envs) and counter < 2^31, like param_block; mirrored in oracle/fishing_oracle.py: model_words / model_draw.
constexpr uint32_t kModelKeyTag = 0x4D4F444Cu;
__device__ __forceinline__ void model_block(uint64_t seed, uint64_t quad, uint64_t counter, bool reset_stream, uint32_t& w0, uint32_t& w1) {
    // (the key -- and with it the ten round keys -- stays wave-uniform, in SGPRs: the quad's high part, zero below 2^34 envs,
    // perturbs the counter word instead of the key)
    const uint32_t c1 = (((uint32_t)counter & ~kParamResetBit) | (reset_stream ? kParamResetBit : 0u)) ^ ((uint32_t)(quad >> 32) * 0xC2B2AE35u);
    const uint32_t key = param_key(seed) ^ kModelKeyTag ^ ((uint32_t)(counter >> 31) * 0x9E3779B1u);
    philox2x32_10((uint32_t)quad, c1, key, w0, w1);
}
__device__ __forceinline__ int32_t model_index_from_half(uint32_t half16, int32_t n_models) {
    return (int32_t)((half16 * (uint32_t)n_models) >> 16);
}
// Returns whether any kind was redrawn.  N = 4: the thread holds the quad; N = 2 (the float64 two-envs-per-thread forms): the thread
// holds the quad's lower or upper pair (base & 2) and takes that pair's halves of the quad's block.
template <int N>
__device__ __forceinline__ bool redraw_kinds(uint64_t seed, uint64_t base, uint64_t counter, uint32_t stream,
                                             const int32_t (&kinds)[FISHING_N_KINDS], int32_t n_models,
                                             const bool (&fin)[N], int32_t (&kind)[N]) {
    static_assert(N == 4 || N == 2, "a quad or a pair");
    bool any = false;
#pragma unroll
    for (int j = 0; j < N; ++j) any |= fin[j];
    if (!any) return false;
    uint32_t w0, w1;
    model_block(seed, base >> 2, counter, stream == kStreamReset, w0, w1);
    if constexpr (N == 4) {
        const uint32_t hh[4] = {w0 & 0xFFFFu, w0 >> 16, w1 & 0xFFFFu, w1 >> 16};
#pragma unroll
        for (int j = 0; j < 4; ++j) kind[j] = fin[j] ? kinds[model_index_from_half(hh[j], n_models)] : kind[j];
    } else {
        const uint32_t w = (base & 2u) ? w1 : w0;
        kind[0] = fin[0] ? kinds[model_index_from_half(w & 0xFFFFu, n_models)] : kind[0];
        kind[1] = fin[1] ? kinds[model_index_from_half(w >> 16, n_models)] : kind[1];
    }
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
// `stamp` (FishingBuffers.v4_stamp; 0 = none): the env was reset on its own by the masked reset() with reset counter
// stamp - 1 and has not finished since -- its episode runs on THAT reset's draw, whatever its year counter says.
template <typename T, bool NARROW = false>
__device__ __forceinline__ void derive_model_error(uint64_t seed, uint64_t env, uint64_t step_counter, int32_t t,
                                                   uint64_t origin_step, uint64_t origin_counter, T K_mean,
                                                   T r_mean, T sigma_p, T& K, T& r, int32_t stamp = 0) {
    if constexpr (NARROW) {
        // every env index of the wave fits 32 bits and every counter 31 (derive_fits_32): the same block with half the
        // integer work -- param_block's key terms of the high parts are zero, the key is wave-uniform
        const uint32_t since = (uint32_t)step_counter - (uint32_t)t;
        const bool from_reset = since == (uint32_t)origin_step;
        uint32_t c1 = from_reset ? ((uint32_t)origin_counter | kParamResetBit) : since - 1u;
        c1 = stamp ? (((uint32_t)stamp - 1u) | kParamResetBit) : c1;
        uint32_t w0, w1;
        philox2x32_10((uint32_t)env, c1, param_key(seed), w0, w1);
        float zK, zr;
        box_muller(w0, w1, zK, zr);
        K = clip_param<T>(K_mean + sigma_p * (T)zK);
        r = clip_param<T>(r_mean + sigma_p * (T)zr);
    } else {
        const uint64_t since = step_counter - (uint64_t)(int64_t)t;
        const bool from_reset = (since == origin_step) || stamp != 0;
        const uint64_t counter = stamp ? (uint64_t)(uint32_t)(stamp - 1) : (from_reset ? origin_counter : since - 1);
        draw_model_error_block<T>(seed, env, counter, from_reset, K_mean, r_mean, sigma_p, K, r);
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

// The same redraw with the wave's finished envs COMPACTED first (the fused kernels, where a wave steps its 256 envs many times
// and the redraw is VALU time): redraw_tile runs four draws per lane whenever one of the lane's envs finished -- on a workload
// that finishes a quarter of the envs per step every wave pays all four passes for ~64 draws.  Here every finished env takes a
// rank in a wave-private LDS window (ballot + prefix), pass p draws for ranks 64 p .. 64 p + 63 -- one lane per finished env,
// keyed by THAT env's global index, so the values are redraw_tile's bit for bit -- and the owners read their pair back:
// ceil(finished / 64) passes instead of four.  `win`: kRedrawSlots entries of the wave.
constexpr int kRedrawSlots = 256;
template <typename T>
struct alignas(2 * sizeof(T)) RedrawSlot {
    T K, r;
};
template <typename T, int MODEL>
__device__ __forceinline__ bool redraw_tile_compact(uint64_t seed, uint64_t base, uint64_t counter, uint32_t stream,
                                                    T K_mean, T r_mean, T sigma_p, T x0, const bool (&fin)[4],
                                                    T (&KK)[4], T (&rr)[4], T (&obs)[4], int32_t (&t)[4],
                                                    RedrawSlot<T>* __restrict__ win, int lane) {
    int slot[4];
    int total = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint64_t bal = __ballot(fin[j]);
        slot[j] = total + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        total += __popcll(bal);                   // wave-uniform
    }
    if (total == 0) return false;
    // the id of a finished env inside the wave's 256 (lane * 4 + j) travels in the K field, as bits
    typedef std::conditional_t<sizeof(T) == 4, uint32_t, uint64_t> bitsT;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (fin[j]) *reinterpret_cast<bitsT*>(&win[slot[j]].K) = (bitsT)(lane * 4 + j);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint64_t wave_base = base - (uint64_t)(lane * 4);     // global index of the wave's first env
    for (int c = 0; c < total; c += kWave) {                    // wave-uniform trip count
        if (c + lane < total) {
            const uint64_t id = (uint64_t)*reinterpret_cast<const bitsT*>(&win[c + lane].K);
            T K2, r2;
            draw_model_error<T>(seed, wave_base + id, counter, stream, K_mean, r_mean, sigma_p, K2, r2);
            win[c + lane] = RedrawSlot<T>{K2, r2};
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (fin[j]) {
            const RedrawSlot<T> v = win[slot[j]];
            KK[j] = v.K;
            rr[j] = v.r;
            obs[j] = reset_obs<T, MODEL>(x0, KK[j]);
            t[j] = 0;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();              // the next step reuses the window
    return fin[0] | fin[1] | fin[2] | fin[3];     // (this lane's own: what it has to write back)
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

// NT = 1: nontemporal (streaming) store (the fused kernel's per-step reward rows, which nobody re-reads inside the launch).
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
__device__ __forceinline__ void add_block_partials(const double (&acc)[kPartialFields], double* partials, const int64_t slot = -1) {
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
        // (`slot`: the tile's own slot where a workgroup steps more than one tile; default = the workgroup's)
        if (tot != 0.0) unsafeAtomicAdd(&partials[(slot >= 0 ? slot : (int64_t)blockIdx.x) * kPartialFields + threadIdx.x], tot);
    }
}

}  // namespace fishing
