// fishing_step.hip -- the vectorised step() / reset() kernels and their C ABI (gfx950).
//
// Data layout in HBM: structure of arrays, one contiguous stream per field
// (obs, action, reward, done, t [, r, K, sigma, z_ext, terminal_obs, ep_return]); env i of
// the shard sits at element i of every stream.  A thread owns 4 consecutive envs, so each
// f32 / i32 stream is read or written with one 16-byte access per lane (1 KiB per wave
// instruction) and the done bytes with one dword per lane.  No env ever reads another env's
// state: there is no LDS staging of the streams (nothing is reused) -- LDS only carries the
// per-workgroup reduction of the episodic-return record and fishing-v11's regroup-by-kind windows.
//
// Roofline: HBM.  Algorithmic bytes per env-step: f32 layout 25 B
// (R obs 4 + action 4 + t 4; W obs 4 + reward 4 + done 1 + t 4), v4 +12 B (r, K, sigma
// arrays), f64 parity layout 37 B; +4/8 B per optional stream.
#include "fishing_common.h"

namespace fishing {

#ifndef FISHING_STEP_ATTRS
#define FISHING_STEP_ATTRS
#endif
#ifndef FISHING_NT_STORE
#define FISHING_NT_STORE 0
#endif
#ifndef FISHING_GENERAL_BATCH_ARGS
#define FISHING_GENERAL_BATCH_ARGS 0
#endif
#ifndef FISHING_LEAN_FENCE
#define FISHING_LEAN_FENCE 1     // bit 0: scheduling fence after the tile's loads, bit 1: after the Philox block.
                                 // With one Philox block per tile (quad noise) the compiler otherwise sinks
                                 // loads behind the generator: 16.6 -> 16.3 us bare, 21.6 -> 21.5 us with
                                 // returns (bits 1, 2, 3 measure the same; profiles/r01f_lean_fence_ab.txt)
#endif
#ifndef FISHING_LEAN_BATCH_ARGS
#define FISHING_LEAN_BATCH_ARGS 1
#endif
#ifndef FISHING_STEP_MAXTHREADS
#define FISHING_STEP_MAXTHREADS 256      // experiment knob: 512 / 1024-thread workgroups
#endif

template <typename T, int MODEL, int NOISE>
__global__ void __launch_bounds__(FISHING_STEP_MAXTHREADS) FISHING_STEP_ATTRS
step_kernel(const ParamsT<T> p, const BuffersT<T> b, const int64_t n, const uint64_t env_offset,
            const uint64_t seed, const uint64_t step_counter_arg) {
#if FISHING_GENERAL_BATCH_ARGS
    // same one-batch kernel-argument load as the lean kernel (the always-used arguments only)
    asm volatile("" ::"s"(b.obs), "s"(b.action), "s"(b.reward), "s"(b.done), "s"(b.t), "s"(b.counter), "s"(b.sigma),
                 "s"(b.ep_return), "s"(b.partials), "s"(b.done_bits), "s"(b.terminal_obs), "s"(p.r), "s"(p.K), "s"(p.sigma),
                 "s"(p.C), "s"(p.x0), "s"(p.Tmax), "s"(p.flags), "s"(n), "s"(env_offset), "s"(seed), "s"(step_counter_arg));
#endif
    // graph-replay safety: with a device-resident counter the launch arguments can stay frozen
    // in a captured hipGraph while the noise key still advances (wave-uniform scalar load)
    const uint64_t step_counter = b.counter ? (*b.counter + step_counter_arg) : step_counter_arg;
    constexpr bool kPerEnv = (MODEL == FISHING_MODEL_V4);
    constexpr bool kZoo = is_zoo_tag(MODEL);
    constexpr int kZooKind = (kZoo && MODEL != kModelZooMixed) ? (MODEL - kModelZoo) : -1;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t tile_envs = (int64_t)blockDim.x * kEnvsPerThread;
    const int64_t ntiles = (n + tile_envs - 1) / tile_envs;
    const bool auto_reset = (p.flags & FISHING_FLAG_AUTO_RESET) != 0;
    double acc[kPartialFields] = {0.0, 0.0, 0.0, 0.0};
    // zoo (fishing-v5..v11): wave-uniform facts about the family
    const bool zoo_drift = kZoo && p.model == FISHING_MODEL_V10;       // r += alpha every draw
    constexpr bool zoo_mixed = (MODEL == kModelZooMixed);              // growth kind per env
    constexpr int zoo_kind = (kZooKind >= 0) ? kZooKind : FISHING_KIND_BEVERTON_HOLT;
    const GrowthT<T> zoo_base = p.growth;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = (tile * blockDim.x + threadIdx.x) * kEnvsPerThread;
        const bool active = base < n;
        const bool full = base + kEnvsPerThread <= n;

        T obs[4], rr[4], KK[4], sg[4], z[4];
        int32_t t[4];
        float a_f[4];
        int32_t a_i[4];
        int32_t kind[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            kind[j] = zoo_kind;
            obs[j] = (T)0;
            t[j] = 0;
            rr[j] = p.r;
            KK[j] = p.K;
            sg[j] = p.sigma;
            z[j] = (T)0;
            a_f[j] = -1.0f;
            a_i[j] = 0;
        }
        if (active) {
            load4<T>(b.obs, base, n, full, obs, (T)0);
            load_t4(b.t, (p.flags & FISHING_FLAG_T_U8) != 0, base, n, full, t);
            if (MODEL == FISHING_MODEL_V0)
                load4<int32_t>((const int32_t*)b.action, base, n, full, a_i, 0);
            else
                load4<float>((const float*)b.action, base, n, full, a_f, -1.0f);
            if (kPerEnv) {
                load4<T>(b.r, base, n, full, rr, p.r);
                load4<T>(b.K, base, n, full, KK, p.K);
            }
            if (zoo_drift) load4<T>(b.r, base, n, full, rr, p.r);
            if (zoo_mixed) load4<int32_t>(b.model_idx, base, n, full, kind, FISHING_KIND_BEVERTON_HOLT);
            if (b.sigma) load4<T>(b.sigma, base, n, full, sg, p.sigma);
            if (NOISE == kNoiseExt) load4<T>(b.z_ext, base, n, full, z, (T)0);
        }
        if (NOISE == kNoisePhilox) {
            float zq[4];
            noise_quad(seed, (env_offset + (uint64_t)base) >> 2, step_counter, zq);
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = (T)zq[j];
        }

        T obs_next[4], rew[4];
        int32_t t_next[4];
        bool dn[4];
        bool stepped = false;
        if constexpr (zoo_mixed) {
            if (!b.sigma) {     // wave-uniform.  fishing-v11: regroup the wave's envs by growth function
                __shared__ T win[(FISHING_STEP_MAXTHREADS / kWave) * 512];
                T xh[4], hv[4], xn[4];
                int kk[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const T quota = quota_cts<T>((T)a_f[j], KK[j]);
                    const T x = (obs[j] + (T)1) * KK[j];
                    hv[j] = (quota < x) ? quota : x;
                    const T d = x - hv[j];
                    xh[j] = ((T)0 > d) ? (T)0 : d;
                    xn[j] = (T)0;
                    kk[j] = (kind[j] >= 0 && kind[j] < FISHING_N_KINDS) ? kind[j] : FISHING_KIND_BEVERTON_HOLT;
                }
                zoo_draw_regrouped<T>(kk, xh, z, p.zoo, xn, win + (threadIdx.x >> 6) * 512, lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    obs_next[j] = xn[j] / KK[j] - (T)1;
                    rew[j] = ((T)0 > hv[j]) ? (T)0 : hv[j];
                    t_next[j] = t[j] + 1;
                    dn[j] = ((t_next[j] > p.Tmax) || (xn[j] <= (T)0)) && (base + j < n);
                }
                stepped = true;
            }
        }
        if (!stepped) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const T quota = (MODEL == FISHING_MODEL_V0) ? quota_int<T>(a_i[j], p.n_actions, KK[j])
                                                        : quota_cts<T>((T)a_f[j], KK[j]);
            if constexpr (kZoo) {
                GrowthT<T> P = zoo_base;
                if (zoo_mixed) {
                    const int kk = (kind[j] >= 0 && kind[j] < FISHING_N_KINDS) ? kind[j] : FISHING_KIND_BEVERTON_HOLT;
                    P = p.zoo[kk];
                }
                if (zoo_drift) {                     // growth_models.py:151: drift first, then draw
                    rr[j] = rr[j] + p.alpha;
                    P.r = rr[j];
                }
                if (b.sigma) P.sigma = sg[j];
                if (zoo_drift)
                    env_step_zoo<T, kZooKind, true>(obs[j], t[j], quota, z[j], kind[j], P, KK[j], p.Tmax, obs_next[j],
                                                    rew[j], dn[j], t_next[j]);
                else
                    env_step_zoo<T, kZooKind, false>(obs[j], t[j], quota, z[j], kind[j], P, KK[j], p.Tmax, obs_next[j],
                                                     rew[j], dn[j], t_next[j]);
            } else {
                env_step<T, MODEL>(obs[j], t[j], quota, z[j], rr[j], KK[j], sg[j], p.C, p.Tmax,
                                   obs_next[j], rew[j], dn[j], t_next[j]);
            }
            dn[j] = dn[j] && (base + j < n);
        }
        }
        const bool lane_done = dn[0] | dn[1] | dn[2] | dn[3];
        // wave-ballot termination mask: a wave with no finished env skips everything below
        const bool wave_done = __any(lane_done);

        if (active) {
            if (b.reward) store4<T, FISHING_NT_STORE>(b.reward, base, n, full, rew);
            if (b.terminal_obs) store4<T, FISHING_NT_STORE>(b.terminal_obs, base, n, full, obs_next);
            if (b.done) {
                if (full) {
                    const uint32_t packed = (uint32_t)dn[0] | ((uint32_t)dn[1] << 8) |
                                            ((uint32_t)dn[2] << 16) | ((uint32_t)dn[3] << 24);
                    if (FISHING_NT_STORE) __builtin_nontemporal_store(packed, reinterpret_cast<uint32_t*>(b.done + base));
                    else *reinterpret_cast<uint32_t*>(b.done + base) = packed;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (base + j < n) b.done[base + j] = (uint8_t)dn[j];
                }
            }
        }
        if (b.done_bits) {
            const uint32_t nibble = (uint32_t)dn[0] | ((uint32_t)dn[1] << 1) | ((uint32_t)dn[2] << 2) |
                                    ((uint32_t)dn[3] << 3);
            const uint64_t word = ballot_tile_words(nibble, lane);
            // first env of this wave's 256-env tile
            const int64_t wave_env0 = (tile * blockDim.x + (threadIdx.x & ~(kWave - 1))) * kEnvsPerThread;
            const int64_t widx = (wave_env0 >> 6) + lane;
            if (lane < 4 && (widx << 6) < n) b.done_bits[widx] = word;
        }

        if (b.ep_return) {
            T er[4] = {(T)0, (T)0, (T)0, (T)0};
            if (active) load4<T>(b.ep_return, base, n, full, er, (T)0);
#pragma unroll
            for (int j = 0; j < 4; ++j) er[j] = er[j] + rew[j];
            if (wave_done) {
                record_tile<T>(dn, er, t_next, acc);
#pragma unroll
                for (int j = 0; j < 4; ++j) er[j] = (dn[j] && auto_reset) ? (T)0 : er[j];
            }
            if (active) store4<T>(b.ep_return, base, n, full, er);
        }

        if (zoo_drift && active) store4<T>(b.r, base, n, full, rr);
        if (auto_reset && wave_done) {
            bool redrawn = false;
            if (zoo_mixed) {      // growth_models.py:200: a new model for the next episode
                if (redraw_kinds(seed, env_offset + (uint64_t)base, step_counter, kStreamAutoReset, p.kinds, p.n_models, dn,
                                 kind))
                    store4<int32_t>(b.model_idx, base, n, full, kind);
            }
            if (kPerEnv) {
                redrawn = redraw_tile<T, MODEL>(seed, env_offset + (uint64_t)base, step_counter, kStreamAutoReset,
                                                p.K_mean, p.r_mean, p.sigma_p, p.x0, dn, KK, rr, obs_next, t_next);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (dn[j]) {
                        obs_next[j] = reset_obs<T, MODEL>(p.x0, KK[j]);
                        t_next[j] = 0;
                    }
                }
            }
            if (kPerEnv && redrawn) {
                store4<T>(b.K, base, n, full, KK);
                store4<T>(b.r, base, n, full, rr);
            }
        }
        if (active) {
            store4<T>(b.obs, base, n, full, obs_next);
            store_t4(b.t, (p.flags & FISHING_FLAG_T_U8) != 0, base, n, full, t_next);
        }
    }

    if (b.partials) add_block_partials<FISHING_STEP_MAXTHREADS / kWave>(acc, b.partials);
}

// ---------------------------------------------------------------- lean fast path
// The same step as step_kernel for the common case -- fp32 layout, fishing-v0/v1/v2/v4, no
// optional stream except the episodic-return accumulator, whole 1024-env tiles -- with
// everything the general kernel decides at run time decided at compile time: unconditional
// 16-byte accesses (the ragged tail goes to a second, general launch), select-based
// auto-reset, and a compact argument block (72 SGPRs instead of 106 -> 8 waves per SIMD).
// Measured at N = 2^22 against a pure copy with the same stream shape (scripts/exp/):
// copy 16.0 us, this 16.2 us, general kernel 17.7 us.  Same results bit for bit
// (tests/test_gpu_parity.py::test_lean_and_general_kernels_agree).
template <typename T>
struct LeanArgs {
    T* obs;
    const void* action;
    T* reward;
    uint8_t* done;
    int32_t* t;
    T* r;
    T* K;
    T* ep_return;
    double* partials;
    const uint64_t* counter;
    const T* sigma_arr;      // per-env noise scale (fishing-v4 with parameter arrays, SIGARR)
    T* terminal_obs;         // TERM: the observation before the fused auto-reset (SB3's terminal_observation)
    uint64_t* done_bits;     // BITS: wave-ballot termination mask, bit i % 64 of word i / 64 = done[i]
    T pr, pK, sigma, C, x0, r_mean, K_mean, sigma_p;
    int32_t Tmax, n_actions;
    uint32_t auto_reset;
    GrowthT<T> growth;       // fishing-v5..v10: the growth function's parameter set (unused, hence never
                             // loaded, by the v0/v1/v2/v4 instantiations)
    T alpha;                 // fishing-v10: per-draw drift of the per-env r (DRIFT)
};

#ifndef FISHING_LEAN_ATTRS
#define FISHING_LEAN_ATTRS
#endif
template <typename T, int MODEL, int NOISE, bool RET, bool SIGARR = false, bool T8 = false, bool DRIFT = false,
          bool TERM = false, bool BITS = false, bool ZZ = false>
__global__ void __launch_bounds__(256) FISHING_LEAN_ATTRS
step_kernel_lean(const LeanArgs<T> a, const int64_t ntiles, const uint64_t env_offset, const uint64_t seed,
                 const uint64_t step_counter_arg) {
    constexpr bool kPerEnv = (MODEL == FISHING_MODEL_V4);
    constexpr bool kZoo = is_zoo_tag(MODEL);              // one growth function of the zoo (never the mixed tag)
    constexpr int kZooKind = kZoo ? (MODEL - kModelZoo) : -1;
    static_assert(MODEL != kModelZooMixed, "fishing-v11 runs on the general kernel");
    static_assert(!DRIFT || kZoo, "DRIFT is fishing-v10 (NonStationary Beverton-Holt)");
    // Pull the kernel arguments into SGPRs in ONE batch of scalar loads.  Left alone, the compiler
    // loads arguments next to their first use, which strings five dependent s_load / s_waitcnt round
    // trips in front of the first global load of every wave.  Measured (scripts/exp/ab_lean_variants.py,
    // N = 2^22): bare step 16.9 -> 16.5 us.  With the return accumulator the batch used to cost more than
    // it saved (23.0 -> 23.15 us) while the record ran sixteen double operations per tile; with the cheaper
    // record it pays there too, 21.67 -> 21.45 us, as long as the ep_return / partials pointers stay out of
    // the batch (21.50 with them): profiles/r01f_lean_fence_ab.txt.
    if constexpr (FISHING_LEAN_BATCH_ARGS != 0) {
        asm volatile("" ::"s"(a.obs), "s"(a.action), "s"(a.reward), "s"(a.done), "s"(a.t), "s"(a.counter), "s"(a.pr),
                     "s"(a.pK), "s"(a.sigma), "s"(a.C), "s"(a.x0), "s"(a.Tmax), "s"(a.n_actions), "s"(a.auto_reset),
                     "s"(ntiles), "s"(env_offset), "s"(seed), "s"(step_counter_arg));
        if constexpr (kPerEnv) asm volatile("" ::"s"(a.r), "s"(a.K), "s"(a.r_mean), "s"(a.K_mean), "s"(a.sigma_p));
        if constexpr (SIGARR) asm volatile("" ::"s"(a.sigma_arr));
        if constexpr (TERM) asm volatile("" ::"s"(a.terminal_obs));
        if constexpr (BITS) asm volatile("" ::"s"(a.done_bits));
        if constexpr (kZoo)
            asm volatile("" ::"s"(a.growth.r), "s"(a.growth.K), "s"(a.growth.sigma), "s"(a.growth.C), "s"(a.growth.M),
                         "s"(a.growth.theta), "s"(a.growth.q), "s"(a.growth.b), "s"(a.growth.a), "s"(a.growth.bq),
                         "s"(a.growth.logA), "s"(a.growth.B));
        if constexpr (DRIFT) asm volatile("" ::"s"(a.r), "s"(a.alpha));
    }
    const uint64_t step_counter = a.counter ? (*a.counter + step_counter_arg) : step_counter_arg;
    const bool auto_reset = a.auto_reset != 0;
    double acc[kPartialFields] = {0.0, 0.0, 0.0, 0.0};
    const T robs_scalar = reset_obs<T, MODEL>(a.x0, a.pK);

    for (int64_t it = blockIdx.x; it < ntiles; it += gridDim.x) {
        // ZZ (launched for N >= 2^25, far outside the 256 MiB Infinity Cache): odd steps walk the tiles
        // backwards, so what the previous step touched last is still cached when this one starts there
        // (N = 2^26: 331 -> 297 us).  Inside the cache the forward walk is the faster one (2^22: 16.1 vs
        // 16.5 us), and even a run-time switch costs the returns variant 2.5 % there: own instantiations.
        const int64_t tile = (ZZ && (step_counter & 1)) ? (ntiles - 1 - it) : it;
        const int64_t base = (tile * 256 + threadIdx.x) * kEnvsPerThread;
        T obs[4], rr[4], KK[4], z[4], er[4], sg[4];
        int32_t t[4], a_i[4];
        float a_f[4];
        {
#pragma unroll
            for (int j = 0; j < 4; ++j) sg[j] = a.sigma;
            if (SIGARR) {
                const Vec4<T> qs = *reinterpret_cast<const Vec4<T>*>(a.sigma_arr + base);
#pragma unroll
                for (int j = 0; j < 4; ++j) sg[j] = qs.v[j];
            }
            const Vec4<T> q = *reinterpret_cast<const Vec4<T>*>(a.obs + base);
            Vec4<int32_t> qt;
            if (T8) {
                const uint32_t w = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(a.t) + base);
#pragma unroll
                for (int j = 0; j < 4; ++j) qt.v[j] = (int32_t)((w >> (8 * j)) & 255u);
            } else {
                qt = *reinterpret_cast<const Vec4<int32_t>*>(a.t + base);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                obs[j] = q.v[j];
                t[j] = qt.v[j];
                rr[j] = a.pr;
                KK[j] = a.pK;
                z[j] = (T)0;
                a_i[j] = 0;
                a_f[j] = 0.0f;
            }
            if (MODEL == FISHING_MODEL_V0) {
                const Vec4<int32_t> qa = *reinterpret_cast<const Vec4<int32_t>*>((const int32_t*)a.action + base);
#pragma unroll
                for (int j = 0; j < 4; ++j) a_i[j] = qa.v[j];
            } else {
                const Vec4<float> qa = *reinterpret_cast<const Vec4<float>*>((const float*)a.action + base);
#pragma unroll
                for (int j = 0; j < 4; ++j) a_f[j] = qa.v[j];
            }
            if (kPerEnv) {
                const Vec4<T> qr = *reinterpret_cast<const Vec4<T>*>(a.r + base);
                const Vec4<T> qk = *reinterpret_cast<const Vec4<T>*>(a.K + base);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    rr[j] = qr.v[j];
                    KK[j] = qk.v[j];
                }
            }
            if (DRIFT) {
                const Vec4<T> qr = *reinterpret_cast<const Vec4<T>*>(a.r + base);
#pragma unroll
                for (int j = 0; j < 4; ++j) rr[j] = qr.v[j];
            }
            if (RET) {
                const Vec4<T> qe = *reinterpret_cast<const Vec4<T>*>(a.ep_return + base);
#pragma unroll
                for (int j = 0; j < 4; ++j) er[j] = qe.v[j];
            }
        }
        // the loads above must be in flight BEFORE the ~100-instruction Philox block starts: without
        // this fence the scheduler hoists the (independent) generator above them in some variants
        if (FISHING_LEAN_FENCE & 1) __builtin_amdgcn_sched_barrier(0);
        if (NOISE == kNoisePhilox) {
            float zq[4];
            noise_quad(seed, (env_offset + (uint64_t)base) >> 2, step_counter, zq);
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = (T)zq[j];
            // ... and the generator (which needs none of the loaded data) runs under their latency:
            // the first s_waitcnt vmcnt lands after it, at the first use of a loaded register
            if (FISHING_LEAN_FENCE & 2) __builtin_amdgcn_sched_barrier(0);
        }
        T obs_next[4], rew[4];
        int32_t t_next[4];
        bool dn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const T quota = (MODEL == FISHING_MODEL_V0) ? quota_int<T>(a_i[j], a.n_actions, KK[j])
                                                        : quota_cts<T>((T)a_f[j], KK[j]);
            if constexpr (DRIFT) {                   // growth_models.py:151: drift first, then draw
                GrowthT<T> P = a.growth;
                rr[j] = rr[j] + a.alpha;
                P.r = rr[j];
                env_step_zoo<T, kZooKind, true>(obs[j], t[j], quota, z[j], kZooKind, P, KK[j], a.Tmax, obs_next[j], rew[j],
                                                dn[j], t_next[j]);
            } else if constexpr (kZoo)
                env_step_zoo<T, kZooKind, false>(obs[j], t[j], quota, z[j], kZooKind, a.growth, KK[j], a.Tmax, obs_next[j],
                                                 rew[j], dn[j], t_next[j]);
            else
                env_step<T, MODEL>(obs[j], t[j], quota, z[j], rr[j], KK[j], sg[j], a.C, a.Tmax, obs_next[j], rew[j],
                                   dn[j], t_next[j]);
        }
        {
            // reward and done are write-only streams nobody re-reads inside the step loop: nontemporal
            // stores (0.5-0.7 % at N = 2^22, 1.5 % at 2^24 / 2^26; profiles/r01g_lean_nt_stores.txt)
            typedef T nt4 __attribute__((ext_vector_type(4)));
            const nt4 qv = {rew[0], rew[1], rew[2], rew[3]};
            __builtin_nontemporal_store(qv, reinterpret_cast<nt4*>(a.reward + base));
            __builtin_nontemporal_store((uint32_t)dn[0] | ((uint32_t)dn[1] << 8) | ((uint32_t)dn[2] << 16) | ((uint32_t)dn[3] << 24),
                                        reinterpret_cast<uint32_t*>(a.done + base));
            if constexpr (TERM) {       // obs_next is still the pre-reset observation here
                const nt4 qt = {obs_next[0], obs_next[1], obs_next[2], obs_next[3]};
                __builtin_nontemporal_store(qt, reinterpret_cast<nt4*>(a.terminal_obs + base));
            }
            if constexpr (BITS) {       // the wave's 256 flags as four 64-bit words (ballots, no LDS)
                const int lane = threadIdx.x & (kWave - 1);
                const uint32_t nibble = (uint32_t)dn[0] | ((uint32_t)dn[1] << 1) | ((uint32_t)dn[2] << 2) | ((uint32_t)dn[3] << 3);
                const uint64_t word = ballot_tile_words(nibble, lane);
                const int64_t wave_env0 = (tile * 256 + (threadIdx.x & ~(kWave - 1))) * kEnvsPerThread;
                if (lane < 4) a.done_bits[(wave_env0 >> 6) + lane] = word;
            }
        }
        const bool lane_done = dn[0] | dn[1] | dn[2] | dn[3];
        if (RET) {
#pragma unroll
            for (int j = 0; j < 4; ++j) er[j] = er[j] + rew[j];
            if (__any(lane_done)) {          // wave-ballot: only waves with a finished env record
                record_tile<T>(dn, er, t_next, acc);
#pragma unroll
                for (int j = 0; j < 4; ++j) er[j] = (dn[j] && auto_reset) ? (T)0 : er[j];
            }
            Vec4<T> qe;
#pragma unroll
            for (int j = 0; j < 4; ++j) qe.v[j] = er[j];
            *reinterpret_cast<Vec4<T>*>(a.ep_return + base) = qe;
        }
        if (kPerEnv) {
            if (auto_reset && __any(lane_done)) {
                const bool redrawn =
                    redraw_tile<T, MODEL>(seed, env_offset + (uint64_t)base, step_counter, kStreamAutoReset, a.K_mean,
                                          a.r_mean, a.sigma_p, a.x0, dn, KK, rr, obs_next, t_next);
                if (redrawn) {
                    Vec4<T> qk, qr;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        qk.v[j] = KK[j];
                        qr.v[j] = rr[j];
                    }
                    *reinterpret_cast<Vec4<T>*>(a.K + base) = qk;
                    *reinterpret_cast<Vec4<T>*>(a.r + base) = qr;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool rs = dn[j] && auto_reset;
                obs_next[j] = rs ? robs_scalar : obs_next[j];
                t_next[j] = rs ? 0 : t_next[j];
            }
        }
        {
            Vec4<T> qo;
            Vec4<int32_t> qt;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                qo.v[j] = obs_next[j];
                qt.v[j] = t_next[j];
            }
            *reinterpret_cast<Vec4<T>*>(a.obs + base) = qo;
            if (DRIFT) {
                Vec4<T> qr;
#pragma unroll
                for (int j = 0; j < 4; ++j) qr.v[j] = rr[j];
                *reinterpret_cast<Vec4<T>*>(a.r + base) = qr;
            }
            if (T8) *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(a.t) + base) = pack_t4(t_next);
            else *reinterpret_cast<Vec4<int32_t>*>(a.t + base) = qt;
        }
    }

    if (RET) {
        if (a.partials) add_block_partials<4>(acc, a.partials);
    }
}

// reset(): one env per thread (not on the hot path; runs once per rollout).
template <typename T, int MODEL>
__global__ void __launch_bounds__(256)
reset_kernel(const ParamsT<T> p, const BuffersT<T> b, const int64_t n, const uint64_t env_offset,
             const uint8_t* __restrict__ mask, const uint64_t seed, const uint64_t reset_counter) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        if (mask && !mask[i]) continue;
        T K = p.K;
        if (is_zoo_tag(MODEL) && p.model == FISHING_MODEL_V11) {
            const uint64_t env = env_offset + (uint64_t)i;      // quad scheme of redraw_kinds
            const Words4 w = philox_block(seed, env >> 2, reset_counter, kStreamReset);
            const uint32_t leg = (uint32_t)(env & 3);
            const uint32_t word = leg == 0 ? w.w0 : leg == 1 ? w.w1 : leg == 2 ? w.w2 : w.w3;
            b.model_idx[i] = p.kinds[action_int_from_word(word, p.n_models)];
        }
        if (MODEL == FISHING_MODEL_V4) {
            T r;
            draw_model_error<T>(seed, env_offset + (uint64_t)i, reset_counter, kStreamReset, p.K_mean,
                                p.r_mean, p.sigma_p, K, r);
            b.K[i] = K;
            b.r[i] = r;
        }
        b.obs[i] = reset_obs<T, MODEL>(p.x0, K);
        if (p.flags & FISHING_FLAG_T_U8) reinterpret_cast<uint8_t*>(b.t)[i] = 0;
        else b.t[i] = 0;
        if (b.ep_return) b.ep_return[i] = (T)0;
    }
}

__global__ void __launch_bounds__(256)
reduce_returns_kernel(const double* __restrict__ partials, double* __restrict__ out4) {
    // one workgroup of 4 waves: wave f sums field f.  Lane l adds slots l, l+64, ... in slot
    // order, then a fixed shuffle tree combines the 64 lanes: same bits on every run.
    const int field = threadIdx.x >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    // all 64 loads of a lane are issued before the first add (one latency instead of 64 in a row:
    // 20 us -> a few us), the adds stay in slot order
    constexpr int kPerLane = kMaxBlocks / kWave;
    double v[kPerLane];
#pragma unroll
    for (int k = 0; k < kPerLane; ++k) v[k] = partials[(lane + k * kWave) * kPartialFields + field];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < kPerLane; ++k) s += v[k];
    s = wave_sum(s);
    if (lane == 0) out4[field] = s;
}

// population_draw() over an array of populations, as BMSY() drives it (models/policies.py:59-63)
template <typename T, int MODEL>
__global__ void __launch_bounds__(256)
population_draw_kernel(const ParamsT<T> p, const int kind, const int64_t n, const T* __restrict__ x_in,
                       const T* __restrict__ z, T* __restrict__ x_out) {
    const GrowthT<T> P = p.growth;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        if constexpr (is_zoo_tag(MODEL))
            x_out[i] = zoo_population_draw<T>(kind, x_in[i], z ? z[i] : (T)0, P);
        else
            x_out[i] = population_draw<T, MODEL>(x_in[i], z ? z[i] : (T)0, p.r, p.K, p.sigma, p.C);
    }
}

__global__ void counter_add_kernel(uint64_t* counter, uint64_t delta) { *counter += delta; }

__global__ void __launch_bounds__(256)
noise_kernel(const int64_t n, const uint64_t env_offset, const uint64_t seed, const uint64_t counter,
             const uint32_t stream_tag, uint32_t* __restrict__ words, float* __restrict__ z0,
             float* __restrict__ z1) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const Words4 w = philox_block(seed, env_offset + (uint64_t)i, counter, stream_tag);
        float zc, zs;
        box_muller(w.w0, w.w1, zc, zs);
        if (words) {
            words[4 * i + 0] = w.w0;
            words[4 * i + 1] = w.w1;
            words[4 * i + 2] = w.w2;
            words[4 * i + 3] = w.w3;
        }
        if (z0) z0[i] = zc;
        if (z1) z1[i] = zs;
    }
}

// test hook: the per-env process noise the step / rollout kernels draw (quad scheme of noise_quad)
__global__ void __launch_bounds__(256)
step_normals_kernel(const int64_t n, const uint64_t env_offset, const uint64_t seed, const uint64_t counter,
                    float* __restrict__ z) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t env = env_offset + (uint64_t)i;
        float zq[4];
        noise_quad(seed, env >> 2, counter, zq);
        const int leg = (int)(env & 3);
        z[i] = leg == 0 ? zq[0] : leg == 1 ? zq[1] : leg == 2 ? zq[2] : zq[3];
    }
}

// test hook: the (zK, zr) normals of the fishing-v4 redraw, pair scheme of draw_model_error_pair
__global__ void __launch_bounds__(256)
reset_normals_kernel(const int64_t n, const uint64_t env_offset, const uint64_t seed, const uint64_t counter,
                     const uint32_t stream_tag, float* __restrict__ zK, float* __restrict__ zr) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t env = env_offset + (uint64_t)i;
        const Words4 w = philox_block(seed, env >> 1, counter, stream_tag);
        float a, c;
        box_muller((env & 1) ? w.w2 : w.w0, (env & 1) ? w.w3 : w.w1, a, c);
        if (zK) zK[i] = a;
        if (zr) zr[i] = c;
    }
}

// ---------------------------------------------------------------- host side
static inline bool misaligned(const void* p) { return p && (((uintptr_t)p) & 15u); }

int check_common(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b) {
    if (!p || !b) return FISHING_ERR_NULL;
    if (!is_core_model(p->model) && !is_zoo_model(p->model)) return FISHING_ERR_MODEL;
    if (p->model == FISHING_MODEL_V10 && !b->r) return FISHING_ERR_NULL;
    if (p->model == FISHING_MODEL_V11) {
        if (!b->model_idx) return FISHING_ERR_NULL;
        if (p->n_models < 1 || p->n_models > FISHING_N_KINDS) return FISHING_ERR_SIZE;
        for (int k = 0; k < p->n_models; ++k)
            if (p->kinds[k] < 0 || p->kinds[k] >= FISHING_N_KINDS) return FISHING_ERR_SIZE;
    }
    if (n < 0 || env_offset < 0 || (env_offset & 3)) return FISHING_ERR_SIZE;
    if (p->model == FISHING_MODEL_V0 && p->n_actions <= 0) return FISHING_ERR_SIZE;
    if (p->launch_threads != 0 &&
        (p->launch_threads < 64 || p->launch_threads > FISHING_STEP_MAXTHREADS || (p->launch_threads & 63)))
        return FISHING_ERR_SIZE;
    if (p->launch_blocks < 0) return FISHING_ERR_SIZE;
    if ((p->flags & FISHING_FLAG_T_U8) && (p->Tmax < 0 || p->Tmax > 254)) return FISHING_ERR_SIZE;
    if (!b->obs || !b->t) return FISHING_ERR_NULL;
    if (p->model == FISHING_MODEL_V4 && (!b->r || !b->K)) return FISHING_ERR_NULL;
    if (b->return_partials && !b->ep_return) return FISHING_ERR_NULL;
    if (b->counter && (((uintptr_t)b->counter) & 7u)) return FISHING_ERR_ALIGN;
    const void* ptrs[] = {b->obs,  b->action, b->reward, b->done,         b->done_bits, b->t,           b->r,
                          b->K,    b->sigma,  b->z_ext,  b->terminal_obs, b->ep_return, b->return_partials,
                          b->model_idx};
    for (const void* q : ptrs)
        if (misaligned(q)) return FISHING_ERR_ALIGN;
    return FISHING_OK;
}

void launch_shape(const FishingParams* p, int64_t n, int& blocks, int& threads) {
    threads = p->launch_threads ? p->launch_threads : 256;
    const int64_t tile = (int64_t)threads * kEnvsPerThread;
    const int64_t ntiles = (n + tile - 1) / tile;
    // default cap: one 1024-env tile per workgroup up to 4096 workgroups, grid-stride beyond.
    // Measured on MI355X (profiles/r01b_sweep_blocks.txt): caps from 1024 to 4096 stay within
    // +-4 % of each other at N = 2^21..2^24 with no consistent winner; 4096 is best or near-best
    // in most rows.
    int cap = p->launch_blocks ? p->launch_blocks : kMaxBlocks;
    if (cap > kMaxBlocks) cap = kMaxBlocks;
    blocks = (int)(ntiles < cap ? ntiles : cap);
    if (blocks < 1) blocks = 1;
}

template <typename T, int MODEL>
int launch_step_noise(const ParamsT<T>& pt, const BuffersT<T>& bt, int noise, int64_t n, uint64_t env_offset,
                      uint64_t seed, uint64_t step_counter, int blocks, int threads, hipStream_t s) {
    switch (noise) {
        case kNoiseNone:
            step_kernel<T, MODEL, kNoiseNone><<<blocks, threads, 0, s>>>(pt, bt, n, env_offset, seed, step_counter);
            break;
        case kNoiseExt:
            step_kernel<T, MODEL, kNoiseExt><<<blocks, threads, 0, s>>>(pt, bt, n, env_offset, seed, step_counter);
            break;
        default:
            step_kernel<T, MODEL, kNoisePhilox><<<blocks, threads, 0, s>>>(pt, bt, n, env_offset, seed, step_counter);
            break;
    }
    return (int)hipGetLastError();
}

// ---- lean-path dispatch
template <typename T>
BuffersT<T> offset_buffers(const BuffersT<T>& b, int64_t off, bool t_u8) {
    BuffersT<T> q = b;
    q.obs = b.obs + off;
    q.action = b.action ? (const void*)((const char*)b.action + 4 * off) : nullptr;
    q.reward = b.reward ? b.reward + off : nullptr;
    q.done = b.done ? b.done + off : nullptr;
    q.done_bits = b.done_bits ? b.done_bits + (off >> 6) : nullptr;      // off is a multiple of 1024
    q.t = t_u8 ? reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(b.t) + off) : b.t + off;
    q.r = b.r ? b.r + off : nullptr;
    q.K = b.K ? b.K + off : nullptr;
    q.sigma = b.sigma ? b.sigma + off : nullptr;
    q.z_ext = b.z_ext ? b.z_ext + off : nullptr;
    q.terminal_obs = b.terminal_obs ? b.terminal_obs + off : nullptr;
    q.ep_return = b.ep_return ? b.ep_return + off : nullptr;
    q.model_idx = b.model_idx ? b.model_idx + off : nullptr;
    return q;
}

template <typename T, int MODEL>
int launch_lean(const LeanArgs<T>& a, int noise, bool ret, bool t8, bool drift, int64_t ntiles, uint64_t env_offset,
                uint64_t seed, uint64_t step_counter, int blocks, hipStream_t s) {
    if constexpr (MODEL == kModelZoo + FISHING_KIND_BEVERTON_HOLT) {
        if (drift) {        // fishing-v10: per-env r, read and written every step
#define FISHING_LEAND(NZ, RT) \
    step_kernel_lean<T, MODEL, NZ, RT, false, false, true><<<blocks, 256, 0, s>>>(a, ntiles, env_offset, seed, step_counter)
            if (noise == kNoiseNone) {
                if (ret) FISHING_LEAND(kNoiseNone, true); else FISHING_LEAND(kNoiseNone, false);
            } else {
                if (ret) FISHING_LEAND(kNoisePhilox, true); else FISHING_LEAND(kNoisePhilox, false);
            }
#undef FISHING_LEAND
            return (int)hipGetLastError();
        }
    }
    if constexpr (!is_zoo_tag(MODEL)) if (t8) {       // compact layout: one-byte year counters
#define FISHING_LEAN8(NZ, RT) \
    step_kernel_lean<T, MODEL, NZ, RT, false, true><<<blocks, 256, 0, s>>>(a, ntiles, env_offset, seed, step_counter)
        if (noise == kNoiseNone) {
            if (ret) FISHING_LEAN8(kNoiseNone, true); else FISHING_LEAN8(kNoiseNone, false);
        } else {
            if (ret) FISHING_LEAN8(kNoisePhilox, true); else FISHING_LEAN8(kNoisePhilox, false);
        }
#undef FISHING_LEAN8
        return (int)hipGetLastError();
    }
    if constexpr (MODEL == FISHING_MODEL_V4) {
        if (a.sigma_arr) {      // BASELINE config 5: per-env (r, K, sigma) arrays; always noisy
            if (ret) step_kernel_lean<T, MODEL, kNoisePhilox, true, true><<<blocks, 256, 0, s>>>(a, ntiles, env_offset, seed, step_counter);
            else step_kernel_lean<T, MODEL, kNoisePhilox, false, true><<<blocks, 256, 0, s>>>(a, ntiles, env_offset, seed, step_counter);
            return (int)hipGetLastError();
        }
    }
    if constexpr (sizeof(T) == 4 && (MODEL == FISHING_MODEL_V0 || MODEL == FISHING_MODEL_V1 || MODEL == FISHING_MODEL_V2 ||
                                     MODEL == FISHING_MODEL_V4)) {
        if (ntiles >= (1 << 15) && !a.done_bits && !a.terminal_obs && noise == kNoisePhilox && !a.sigma_arr) {
            // N >= 2^25: the zig-zag walk
            if (ret) step_kernel_lean<T, MODEL, kNoisePhilox, true, false, false, false, false, false, true><<<blocks, 256, 0, s>>>(a, ntiles, env_offset, seed, step_counter);
            else step_kernel_lean<T, MODEL, kNoisePhilox, false, false, false, false, false, false, true><<<blocks, 256, 0, s>>>(a, ntiles, env_offset, seed, step_counter);
            return (int)hipGetLastError();
        }
        if (a.done_bits) {          // ballot bitmask of the finished envs next to the byte flags
            if (ret) step_kernel_lean<T, MODEL, kNoisePhilox, true, false, false, false, false, true><<<blocks, 256, 0, s>>>(a, ntiles, env_offset, seed, step_counter);
            else step_kernel_lean<T, MODEL, kNoisePhilox, false, false, false, false, false, true><<<blocks, 256, 0, s>>>(a, ntiles, env_offset, seed, step_counter);
            return (int)hipGetLastError();
        }
        if (a.terminal_obs) {       // SB3 semantics: record the pre-reset observation (float32, in-kernel noise)
            if (ret) step_kernel_lean<T, MODEL, kNoisePhilox, true, false, false, false, true><<<blocks, 256, 0, s>>>(a, ntiles, env_offset, seed, step_counter);
            else step_kernel_lean<T, MODEL, kNoisePhilox, false, false, false, false, true><<<blocks, 256, 0, s>>>(a, ntiles, env_offset, seed, step_counter);
            return (int)hipGetLastError();
        }
    }
#define FISHING_LEAN(NZ, RT) step_kernel_lean<T, MODEL, NZ, RT><<<blocks, 256, 0, s>>>(a, ntiles, env_offset, seed, step_counter)
    if (noise == kNoiseNone) {
        if (ret) FISHING_LEAN(kNoiseNone, true); else FISHING_LEAN(kNoiseNone, false);
    } else {
        if (ret) FISHING_LEAN(kNoisePhilox, true); else FISHING_LEAN(kNoisePhilox, false);
    }
#undef FISHING_LEAN
    return (int)hipGetLastError();
}

template <typename T>
int step_impl(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b, uint64_t seed,
              uint64_t step_counter, fishing_stream_t stream) {
    const int rc = check_common(p, n, env_offset, b);
    if (rc != FISHING_OK) return rc;
    if (!b->action) return FISHING_ERR_NULL;
    if (n == 0) return FISHING_OK;
    const ParamsT<T> pt = narrow_params<T>(*p);
    const BuffersT<T> bt = typed_buffers<T>(*b);
    // the reference draws a normal even at sigma == 0 (quirk B3) but multiplies it by 0:
    // skipping the generator there changes no result bit.
    bool quiet = (p->sigma == 0.0 && !b->sigma);
    if (p->model == FISHING_MODEL_V11) {
        quiet = !b->sigma;
        for (int k = 0; k < FISHING_N_KINDS; ++k) quiet = quiet && p->zoo[k].sigma == 0.0;
    }
    const int noise = b->z_ext ? kNoiseExt : (quiet ? kNoiseNone : kNoisePhilox);
    int blocks, threads;
    launch_shape(p, n, blocks, threads);
    hipStream_t s = (hipStream_t)stream;

    // lean fast path (v0/v1/v2/v4, no optional stream but the return accumulator)
    {
        // fishing-v5..v10 (one growth function; v10 adds the drifting per-env r stream) share the lean
        // kernel; v11 (growth function per env) needs the general kernel
        const bool zoo_lean = sizeof(T) == 4 && is_zoo_model(p->model) && p->model != FISHING_MODEL_V11 &&
                              !(p->flags & FISHING_FLAG_T_U8) && !b->sigma;
        const bool core = is_core_model(p->model) || zoo_lean;
        const int64_t tile = 256 * kEnvsPerThread;
        // (fp64 fishing-v4 stays on the general kernel: measured 50.1 vs 51.6 us at N = 2^22)
        if (core && (sizeof(T) == 4 || p->model != FISHING_MODEL_V4) && noise != kNoiseExt && !(p->flags & FISHING_FLAG_GENERAL_KERNEL) && b->reward && b->done &&
            (!(b->done_bits || b->terminal_obs) ||
             (!(b->done_bits && b->terminal_obs) && sizeof(T) == 4 && is_core_model(p->model) && noise == kNoisePhilox &&
              !b->sigma && !(p->flags & FISHING_FLAG_T_U8))) &&
            (!b->sigma || (p->model == FISHING_MODEL_V4 && noise == kNoisePhilox && !(p->flags & FISHING_FLAG_T_U8))) &&
            (p->launch_threads == 0 || p->launch_threads == 256) && n >= tile) {
            const int64_t ntiles = n / tile;
            const int64_t n_full = ntiles * tile;
            LeanArgs<T> a{bt.obs,      bt.action,  bt.reward,  bt.done,     bt.t,        bt.r,
                          bt.K,        bt.ep_return, bt.partials, bt.counter, bt.sigma,  bt.terminal_obs, bt.done_bits, pt.r, pt.K,
                          pt.sigma,    pt.C,       pt.x0,      pt.r_mean,   pt.K_mean,   pt.sigma_p,
                          pt.Tmax,     pt.n_actions, (uint32_t)(p->flags & FISHING_FLAG_AUTO_RESET), pt.growth, pt.alpha};
            // up to 4096 workgroups = one tile each at N = 2^22: 21.39 -> 21.13 us with returns against a cap of
            // 2048, equal for the bare step (profiles/r01j_lean_block_cap.jsonl)
            int cap = p->launch_blocks ? p->launch_blocks : kMaxBlocks;
            if (cap > kMaxBlocks) cap = kMaxBlocks;
            const int lb = (int)(ntiles < cap ? ntiles : cap);
            const bool ret = b->ep_return != nullptr;
            const bool t8 = (p->flags & FISHING_FLAG_T_U8) != 0;
            const int rc2 = with_model_tag(p->model, [&](auto tag) {
                constexpr int kTag = decltype(tag)::value;
                if constexpr (kTag == FISHING_MODEL_V0 || kTag == FISHING_MODEL_V1 || kTag == FISHING_MODEL_V2 ||
                              kTag == FISHING_MODEL_V4 || (sizeof(T) == 4 && is_zoo_tag(kTag) && kTag != kModelZooMixed))
                    return launch_lean<T, kTag>(a, noise, ret, t8, p->model == FISHING_MODEL_V10, ntiles, env_offset, seed,
                                                step_counter, lb, s);
                else
                    return (int)FISHING_ERR_MODEL;
            });
            if (rc2 != 0 || n_full == n) return rc2;
            // ragged tail (< 1024 envs): one workgroup of the general kernel
            const BuffersT<T> tb = offset_buffers<T>(bt, n_full, (p->flags & FISHING_FLAG_T_U8) != 0);
            return with_model_tag(p->model, [&](auto tag) {
                return launch_step_noise<T, decltype(tag)::value>(pt, tb, noise, n - n_full, env_offset + n_full, seed,
                                                                  step_counter, 1, 256, s);
            });
        }
    }
    return with_model_tag(p->model, [&](auto tag) {
        return launch_step_noise<T, decltype(tag)::value>(pt, bt, noise, n, env_offset, seed, step_counter, blocks,
                                                          threads, s);
    });
}

template <typename T>
int step_many_impl(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                   int64_t action_stride, int32_t ring_len, int32_t n_steps, uint64_t seed,
                   uint64_t step_counter, fishing_stream_t stream) {
    if (!b || !p) return FISHING_ERR_NULL;
    if (ring_len <= 0 || n_steps < 0 || action_stride < 0) return FISHING_ERR_SIZE;
    if (ring_len > 1 && (action_stride & 3)) return FISHING_ERR_ALIGN;
    FishingBuffers bb = *b;
    for (int32_t k = 0; k < n_steps; ++k) {
        // action elements are 4 bytes wide in every layout (f32 / i32)
        bb.action = (const char*)b->action + (size_t)(k % ring_len) * (size_t)action_stride * 4u;
        const int rc = step_impl<T>(p, n, env_offset, &bb, seed, step_counter + (uint64_t)k, stream);
        if (rc != FISHING_OK) return rc;
    }
    return FISHING_OK;
}

template <typename T>
int reset_impl(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
               const uint8_t* mask, uint64_t seed, uint64_t reset_counter, fishing_stream_t stream) {
    const int rc = check_common(p, n, env_offset, b);
    if (rc != FISHING_OK) return rc;
    if (n == 0) return FISHING_OK;
    const ParamsT<T> pt = narrow_params<T>(*p);
    const BuffersT<T> bt = typed_buffers<T>(*b);
    const int threads = 256;
    int64_t nb = (n + threads - 1) / threads;
    const int blocks = (int)(nb < 2048 ? nb : 2048);
    hipStream_t s = (hipStream_t)stream;
    with_model_tag(p->model, [&](auto tag) {
        // reset only distinguishes v4 (parameter redraw, un-normalised obs) and v11 (model draw)
        constexpr int kTag = decltype(tag)::value;
        constexpr int kResetTag = (kTag == FISHING_MODEL_V4) ? FISHING_MODEL_V4
                                  : is_zoo_tag(kTag)         ? kModelZooMixed
                                                             : FISHING_MODEL_V1;
        reset_kernel<T, kResetTag><<<blocks, threads, 0, s>>>(pt, bt, n, env_offset, mask, seed, reset_counter);
        return 0;
    });
    return (int)hipGetLastError();
}

template <typename T>
int population_draw_impl(const FishingParams* p, int64_t n, const void* x_in, const void* z, void* x_out,
                         fishing_stream_t stream) {
    if (!p || !x_in || !x_out) return FISHING_ERR_NULL;
    if (n < 0) return FISHING_ERR_SIZE;
    if (n == 0) return FISHING_OK;
    const ParamsT<T> pt = narrow_params<T>(*p);
    int64_t nb = (n + 255) / 256;
    const int blocks = (int)(nb < 2048 ? nb : 2048);
    hipStream_t s = (hipStream_t)stream;
    if (is_zoo_model(p->model) && p->model != FISHING_MODEL_V11) {
        population_draw_kernel<T, kModelZoo><<<blocks, 256, 0, s>>>(pt, kind_of_model(p->model), n, (const T*)x_in,
                                                                    (const T*)z, (T*)x_out);
        return (int)hipGetLastError();
    }
    if (p->model == FISHING_MODEL_V2)
        population_draw_kernel<T, FISHING_MODEL_V2><<<blocks, 256, 0, s>>>(pt, 0, n, (const T*)x_in, (const T*)z, (T*)x_out);
    else if (p->model == FISHING_MODEL_V0 || p->model == FISHING_MODEL_V1 || p->model == FISHING_MODEL_V4)
        population_draw_kernel<T, FISHING_MODEL_V1><<<blocks, 256, 0, s>>>(pt, 0, n, (const T*)x_in, (const T*)z, (T*)x_out);
    else
        return FISHING_ERR_MODEL;
    return (int)hipGetLastError();
}

}  // namespace fishing

extern "C" {

int fishing_population_draw_f32(const FishingParams* p, int64_t n, const void* x_in, const void* z, void* x_out,
                                fishing_stream_t stream) {
    return fishing::population_draw_impl<float>(p, n, x_in, z, x_out, stream);
}
int fishing_population_draw_f64(const FishingParams* p, int64_t n, const void* x_in, const void* z, void* x_out,
                                fishing_stream_t stream) {
    return fishing::population_draw_impl<double>(p, n, x_in, z, x_out, stream);
}

int fishing_abi_version(void) { return FISHING_ABI_VERSION; }

const char* fishing_error_string(int code) {
    switch (code) {
        case FISHING_OK: return "ok";
        case FISHING_ERR_NULL: return "a required pointer is NULL";
        case FISHING_ERR_MODEL: return "unknown model id";
        case FISHING_ERR_ALIGN: return "buffer not 16-byte aligned";
        case FISHING_ERR_SIZE: return "bad size / count / offset argument";
        case FISHING_ERR_POLICY: return "unknown in-kernel policy";
        case FISHING_ERR_NO_DEVICE: return "no usable HIP device";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

int64_t fishing_partials_len(void) { return (int64_t)fishing::kMaxBlocks * fishing::kPartialFields; }

int fishing_step_f32(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                     uint64_t seed, uint64_t step_counter, fishing_stream_t stream) {
    return fishing::step_impl<float>(p, n, env_offset, b, seed, step_counter, stream);
}
int fishing_step_f64(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                     uint64_t seed, uint64_t step_counter, fishing_stream_t stream) {
    return fishing::step_impl<double>(p, n, env_offset, b, seed, step_counter, stream);
}
int fishing_step_many_f32(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                          int64_t action_stride, int32_t ring_len, int32_t n_steps, uint64_t seed,
                          uint64_t step_counter, fishing_stream_t stream) {
    return fishing::step_many_impl<float>(p, n, env_offset, b, action_stride, ring_len, n_steps, seed,
                                          step_counter, stream);
}
int fishing_step_many_f64(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                          int64_t action_stride, int32_t ring_len, int32_t n_steps, uint64_t seed,
                          uint64_t step_counter, fishing_stream_t stream) {
    return fishing::step_many_impl<double>(p, n, env_offset, b, action_stride, ring_len, n_steps, seed,
                                           step_counter, stream);
}
int fishing_reset_f32(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                      const uint8_t* mask, uint64_t seed, uint64_t reset_counter, fishing_stream_t stream) {
    return fishing::reset_impl<float>(p, n, env_offset, b, mask, seed, reset_counter, stream);
}
int fishing_reset_f64(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                      const uint8_t* mask, uint64_t seed, uint64_t reset_counter, fishing_stream_t stream) {
    return fishing::reset_impl<double>(p, n, env_offset, b, mask, seed, reset_counter, stream);
}

int fishing_counter_add(uint64_t* counter, uint64_t delta, fishing_stream_t stream) {
    if (!counter) return FISHING_ERR_NULL;
    if (((uintptr_t)counter) & 7u) return FISHING_ERR_ALIGN;
    fishing::counter_add_kernel<<<1, 1, 0, (hipStream_t)stream>>>(counter, delta);
    return (int)hipGetLastError();
}

int fishing_stream_synchronize(fishing_stream_t stream) { return (int)hipStreamSynchronize((hipStream_t)stream); }

int fishing_reduce_returns(const double* return_partials, double* out4, fishing_stream_t stream) {
    if (!return_partials || !out4) return FISHING_ERR_NULL;
    fishing::reduce_returns_kernel<<<1, 256, 0, (hipStream_t)stream>>>(return_partials, out4);
    return (int)hipGetLastError();
}

int fishing_noise_f32(int64_t n, int64_t env_offset, uint64_t seed, uint64_t counter, int32_t stream_tag,
                      uint32_t* words, float* z0, float* z1, fishing_stream_t stream) {
    if (n < 0 || env_offset < 0 || stream_tag < 0 || stream_tag > 255) return FISHING_ERR_SIZE;
    if (n == 0) return FISHING_OK;
    int64_t nb = (n + 255) / 256;
    const int blocks = (int)(nb < 2048 ? nb : 2048);
    fishing::noise_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(n, (uint64_t)env_offset, seed, counter,
                                                                 (uint32_t)stream_tag, words, z0, z1);
    return (int)hipGetLastError();
}

int fishing_step_normals_f32(int64_t n, int64_t env_offset, uint64_t seed, uint64_t counter, float* z,
                             fishing_stream_t stream) {
    if (n < 0 || env_offset < 0) return FISHING_ERR_SIZE;
    if (!z) return FISHING_ERR_NULL;
    if (n == 0) return FISHING_OK;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, fishing::kMaxBlocks);
    fishing::step_normals_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(n, (uint64_t)env_offset, seed, counter, z);
    return (int)hipGetLastError();
}

int fishing_reset_normals_f32(int64_t n, int64_t env_offset, uint64_t seed, uint64_t counter, int32_t stream_tag,
                              float* zK, float* zr, fishing_stream_t stream) {
    if (n < 0 || env_offset < 0) return FISHING_ERR_SIZE;
    if (n == 0) return FISHING_OK;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, fishing::kMaxBlocks);
    fishing::reset_normals_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(n, (uint64_t)env_offset, seed, counter,
                                                                         (uint32_t)stream_tag, zK, zr);
    return (int)hipGetLastError();
}

}  // extern "C"
