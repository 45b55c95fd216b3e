// fishing_rollout.hip -- fused T-step rollout with an in-kernel policy (gfx950).
//
// The callers of step() in the reference are Python loops (shared_env.py:29-54 simulate_mdp,
// examples/const_escapement.py:19-26, SB3's DummyVecEnv).  Here the loop over time runs inside
// the kernel: a thread keeps its 4 envs' (obs, t, r, K, sigma, ep_return) in registers for
// T steps, draws the noise from the same Philox blocks a step-by-step run would use (and the
// random policy's actions from the policy stream), and touches HBM once at entry and once at exit (plus the optional
// trajectory record).  Results equal T fishing_step_* calls fed the policy's actions.
//
// Bound: VALU (Philox + transcendental issue), not HBM -- reported as env-steps/s only.
//
// Wave-ballot termination: without FISHING_FLAG_AUTO_RESET a finished env is frozen (the
// reference's simulate loop breaks on done, shared_env.py:51-52) and a wave whose 256 envs
// have all finished leaves the time loop (__all over the per-lane masks).
// (the float64 zoo's hand-over to the reference's round trip for far stocks / far results is compiled out of the step, fused-step and
// rollout kernels: fishing_step.hip says why)
#define FISHING_ZOO_F64_FAR 0
#include "fishing_common.h"
#include "fishing_host.h"

namespace fishing {

// PP: one policy parameter PER ENV (fishing_rollout_params_*: `pparams`, real[n]) instead of the scalar -- N fishing-v4 / v11
// envs each escaping to the S its own BMSY() found (models/policies.py:22-31 per env).  Own instantiations (run-time policy,
// auto-reset), so the scalar kernels keep their registers.
template <typename T, int MODEL, int POLICY, bool AUTO, bool KP2C = false, bool PP = false>
__global__ void __launch_bounds__(256)
rollout_kernel(const ParamsT<T> p, const BuffersT<T> b, const int64_t n, const uint64_t env_offset,
               const T policy_param, const int32_t Tsteps, T* __restrict__ traj, const uint64_t seed_arg,
               const uint64_t step_counter_arg, const int noise_on, const int policy_rt, const DivK dk_arg,
               const T* __restrict__ pparams = nullptr) {
    const uint64_t step_counter0 = b.counter ? (*b.counter + step_counter_arg) : step_counter_arg;
    constexpr bool kPerEnv = (MODEL == FISHING_MODEL_V4);
    const bool derived = kPerEnv && (p.flags & FISHING_FLAG_V4_DERIVED) != 0;   // no r / K arrays (derive_model_error)
    uint64_t origin_step = p.origin_step, origin_counter = p.origin_counter;
    if (derived) device_origin(b.counter, origin_step, origin_counter);
    // per-env K keeps the true division.  The power-of-two flag is a run-time one (a wave-uniform branch in front of each
    // division: div_K) -- except in the KP2C twins, which the dispatch picks for the float32 auto-resetting rollouts of
    // fishing-v0/v1/v2 with a power-of-two K (the reference's default K = 1): there it is a compile-time fact, the divisions and
    // their branches are gone from the step loop (+4 % random policy, +8 % escapement at N = 2^22; 12 instantiations more).
    // A compile-time flag for ALL 88 instantiations would double them; unswitching the
    // tile loop on the flag inside the kernel (two copies behind one run-time test) gave 2 % / 7 % -- and cost fishing-v4's and
    // fishing-v11's rollout kernels a wave of occupancy through nothing but the changed register allocation of the wrapped loop
    // (fishing-v11 2.57 -> 2.12e11): profiles/r04_rollout_unswitch.jsonl.
    DivK dk = kPerEnv ? DivK{false, 0.0f, 0.0} : dk_arg;
    if constexpr (KP2C) dk.pow2 = true;
    // POLICY >= 0: compile-time policy (v0/v1/v2/v4 with auto-reset); POLICY < 0: wave-uniform run-time policy (the zoo, to keep
    // the number of instantiations of the transcendental-heavy bodies small; the frozen-episode form -- no auto-reset: the
    // simulate tables -- and the per-env-parameter form of every model: one instantiation each instead of four)
    const int policy = (POLICY >= 0) ? POLICY : policy_rt;
    const bool kNeedWords = (policy == FISHING_POLICY_RANDOM);
    constexpr bool kZoo = is_zoo_tag(MODEL);
    constexpr int kZooKind = (kZoo && MODEL != kModelZooMixed) ? (MODEL - kModelZoo) : -1;
    constexpr bool zoo_mixed = (MODEL == kModelZooMixed);
    // (fishing-v10 is Beverton-Holt: only that tag -- and the general kernel's run-time-kind tag -- can see a drifting r)
    constexpr bool kMayDrift = MODEL == kModelZoo + FISHING_KIND_BEVERTON_HOLT || MODEL == kModelZooRT;
    const bool zoo_drift = kMayDrift && p.model == FISHING_MODEL_V10;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t tile_envs = (int64_t)blockDim.x * kEnvsPerThread;
    const int64_t ntiles = (n + tile_envs - 1) / tile_envs;
    double acc[kPartialFields] = {0.0, 0.0, 0.0, 0.0};
    // fishing-v0's index -> quota map (fishing_env.py:7-24, base_fishing_env.py:140: (a / n_actions) * K) as a table in LDS for the
    // in-kernel random policy, whose index is in [0, n_actions) by construction: entry i holds quota_int(i) itself -- the IEEE
    // division and the multiply, evaluated once per workgroup instead of once per env-step (~10 of a step's ~45 VALU instructions
    // per env).  Up to kQuotaLut actions; beyond that the arithmetic stays in the loop.
    constexpr bool kLutForm = (MODEL == FISHING_MODEL_V0) && (POLICY >= 0);      // (the compile-time-policy, auto-resetting forms)
    constexpr int kQuotaLut = 1024;
    __shared__ T quota_lut[kLutForm ? kQuotaLut : 1];
    const bool use_lut = kLutForm && p.n_actions > 0 && p.n_actions <= kQuotaLut;
    if constexpr (kLutForm) {
        if (use_lut) {
            for (int i = threadIdx.x; i < p.n_actions; i += blockDim.x) quota_lut[i] = quota_int<T>(i, p.n_actions, p.K);
            __syncthreads();
        }
    }

    // fishing-v11: the growth functions' coefficients as a table in LDS (fishing_common.h: zoo_lut_fill), written once per launch
    __shared__ alignas(16) T zoo_lut[zoo_mixed ? kZooLutSize : 4];
    if constexpr (zoo_mixed) {
        if (threadIdx.x < kWave) zoo_lut_fill<T>(zoo_lut, p.zoo);
        __syncthreads();
    }

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = (tile * blockDim.x + threadIdx.x) * kEnvsPerThread;
        const bool active = base < n;
        const bool full = base + kEnvsPerThread <= n;

        T obs[4], rr[4], KK[4], sg[4], er[4], rew[4];
        int32_t t[4];
        bool dn[4], live[4];    // live: a real env whose episode is still running
        int32_t kind[4];
        int32_t st[4] = {0, 0, 0, 0};       // fishing-v4 derived: per-env origin stamps (FishingBuffers.v4_stamp)
        const bool stamped = derived && b.stamp != nullptr;
        bool stamp_dirty = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            kind[j] = (kZooKind >= 0) ? kZooKind : FISHING_KIND_BEVERTON_HOLT;
            obs[j] = (T)0;
            t[j] = 0;
            rr[j] = p.r;
            KK[j] = p.K;
            sg[j] = p.sigma;
            er[j] = (T)0;
            rew[j] = (T)0;
            dn[j] = false;
            live[j] = base + j < n;
        }
        if (active) {
            load4<T>(b.obs, base, n, full, obs, (T)0);
            load_t4(b.t, (p.flags & FISHING_FLAG_T_U8) != 0, base, n, full, t);
            if (kPerEnv && !derived) {
                load4<T>(b.r, base, n, full, rr, p.r);
                load4<T>(b.K, base, n, full, KK, p.K);
            }
            if (zoo_drift) load4<T>(b.r, base, n, full, rr, p.r);
            if (zoo_mixed) load4<int32_t>(b.model_idx, base, n, full, kind, FISHING_KIND_BEVERTON_HOLT);
            if (b.sigma) load4<T>(b.sigma, base, n, full, sg, p.sigma);
            if (b.ep_return) load4<T>(b.ep_return, base, n, full, er, (T)0);
            if (stamped) load4<int32_t>(b.stamp, base, n, full, st, 0);
        }
        if (derived) {      // once per launch; the redraws below keep (K, r) current from then on
#pragma unroll
            for (int j = 0; j < 4; ++j)
                derive_model_error<T>(seed_arg, env_offset + (uint64_t)base + j, step_counter0, t[j], origin_step,
                                      origin_counter, p.K_mean, p.r_mean, p.sigma_p, KK[j], rr[j], st[j]);
        }
        T pp[4] = {policy_param, policy_param, policy_param, policy_param};
        if constexpr (PP) {
            if (active) load4<T>(pparams, base, n, full, pp, policy_param);
        }
        bool kind_dirty = false;
        const uint64_t quad = (env_offset + (uint64_t)base) >> 2;
        const T robs_scalar = reset_obs<T, MODEL>(p.x0, p.K);   // loop-invariant unless per-env K
        bool kr_dirty = false;

        for (int32_t s = 0; s < Tsteps; ++s) {
            const uint64_t step_counter = step_counter0 + (uint64_t)s;
            T z[4] = {(T)0, (T)0, (T)0, (T)0};
            uint32_t aw[4] = {0u, 0u, 0u, 0u};
            uint64_t seed = seed_arg;       // Philox key schedule next to its rounds, not in long-lived SGPRs (see the
            asm volatile("" : "+s"(seed));  // fused step kernel below)
            if (noise_on) {         // one block for the tile's four normals
                float zq[4];
                noise_quad(seed, quad, step_counter, zq);
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] = (T)zq[j];
            }
            if (kNeedWords) {       // random policy: one more block, a word per env
                const Words4 w = philox_block(seed, quad, step_counter, kStreamPolicy);
                aw[0] = w.w0;
                aw[1] = w.w1;
                aw[2] = w.w2;
                aw[3] = w.w3;
            }
            // ---- all four envs advance branch-free (independent chains, full ILP); a finished
            // env of a no-auto-reset rollout is frozen by selects, not by control flow
            T obs_in[4], act_rec[4], quota[4];
            bool fin[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                obs_in[j] = obs[j];
                T a_c = (T)-1;
                int32_t a_d = 0;
                if (policy == FISHING_POLICY_RANDOM) {
                    if (MODEL == FISHING_MODEL_V0) a_d = action_int_from_word(aw[j], p.n_actions);
                    else a_c = (T)action_cts_from_word(aw[j]);
                } else if (policy == FISHING_POLICY_CONSTANT) {
                    if (MODEL == FISHING_MODEL_V0) a_d = (int32_t)pp[j];
                    else a_c = (T)(float)pp[j];
                } else {
                    T q;
                    if (policy == FISHING_POLICY_ESCAPEMENT) {   // policies.py:27-31
                        const T x = (obs[j] + (T)1) * KK[j];
                        const T dq = x - pp[j];
                        q = ((T)0 > dq) ? (T)0 : dq;             // max(x - S, 0.0)
                    } else {                                     // MSY, policies.py:16-19
                        q = pp[j];
                    }
                    if (MODEL == FISHING_MODEL_V0) a_d = action_int_from_quota<T>(q, p.n_actions, KK[j], dk);
                    else a_c = (T)action_cts_from_quota<T>(q, KK[j], dk);
                }
                if constexpr (POLICY == FISHING_POLICY_RANDOM && MODEL != FISHING_MODEL_V0) {
                    // action_cts_from_word maps a 32-bit word into [-1, 1] exactly (word * 2^-31 - 1; 2^32 - 1 rounds up to
                    // 2^32 -> 1.0): get_quota's clip is the identity there -- two compare / select pairs per env-step less
                    quota[j] = (a_c + (T)1) * KK[j];
                } else if constexpr (KP2C && POLICY == FISHING_POLICY_ESCAPEMENT && MODEL != FISHING_MODEL_V0) {
                    // q = max(x - S, 0.0) is >= 0 (or NaN) and K a positive power of two here: a = q / K - 1 >= -1, so only the
                    // upper bound of get_quota's clip can bind
                    const T av = (a_c > (T)1) ? (T)1 : a_c;
                    quota[j] = (av + (T)1) * KK[j];
                } else if constexpr (kLutForm) {
                    // the random policy's index is in [0, n_actions) by construction; the quota-driven policies' -- round(q n / K)
                    // -- need not be (a stock above K escaping to a low level): a wave holding one outside evaluates the arithmetic
                    bool in_range = true;
                    if constexpr (POLICY != FISHING_POLICY_RANDOM) in_range = !__any((uint32_t)a_d >= (uint32_t)p.n_actions);
                    if (use_lut && in_range) {
                        quota[j] = quota_lut[a_d];
                    } else {
                        asm volatile("");           // (a real branch: no speculative division)
                        quota[j] = quota_int<T>(a_d, p.n_actions, KK[j]);
                    }
                } else {
                    quota[j] = (MODEL == FISHING_MODEL_V0) ? quota_int<T>(a_d, p.n_actions, KK[j]) : quota_cts<T>(a_c, KK[j]);
                }
                act_rec[j] = (MODEL == FISHING_MODEL_V0) ? (T)a_d : a_c;
            }
            T o2[4], r2[4];
            bool d2[4];
            int32_t t2[4];
            bool stepped = false;
            if constexpr (zoo_mixed) {
                if (!b.sigma) {     // wave-uniform.  fishing-v11: every env's coefficients from the LDS table
                    T xh[4], hv[4], xn[4];
                    int kk[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const T x = (obs[j] + (T)1) * KK[j];
                        hv[j] = (quota[j] < x) ? quota[j] : x;
                        const T d = x - hv[j];
                        xh[j] = d;       // (max(d, 0.0) is the identity here: stock_after_harvest)
                        xn[j] = (T)0;
                        kk[j] = (kind[j] >= 0 && kind[j] < FISHING_N_KINDS) ? kind[j] : FISHING_KIND_BEVERTON_HOLT;
                    }
                    zoo_draw_lut_tile<T, 4>(kk, xh, z, p.zoo, zoo_lut, xn);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o2[j] = xn[j] / KK[j] - (T)1;
                        r2[j] = ((T)0 > hv[j]) ? (T)0 : hv[j];
                        t2[j] = t[j] + 1;
                        d2[j] = (t2[j] > p.Tmax) || (xn[j] <= (T)0);
                    }
                    stepped = true;
                }
            }
            if (!stepped) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (kZoo) {
                        GrowthT<T> P = p.growth;
                        if (zoo_mixed) {
                            const int kk = (kind[j] >= 0 && kind[j] < FISHING_N_KINDS) ? kind[j] : FISHING_KIND_BEVERTON_HOLT;
                            P = p.zoo[kk];
                        }
                        if (b.sigma) P.sigma = sg[j];
                        if (zoo_drift) {                 // growth_models.py:151: r += alpha before every draw;
                            const T r_new = rr[j] + p.alpha;   // a frozen env makes no draw, so its r stays
                            P.r = r_new;
                            rr[j] = (AUTO || live[j]) ? r_new : rr[j];
                            env_step_zoo<T, kZooKind, true>(obs[j], t[j], quota[j], z[j], kind[j], P, KK[j], p.Tmax, o2[j], r2[j],
                                                            d2[j], t2[j]);
                        } else {
                            env_step_zoo<T, kZooKind, false>(obs[j], t[j], quota[j], z[j], kind[j], P, KK[j], p.Tmax, o2[j],
                                                             r2[j], d2[j], t2[j]);
                        }
                    } else {
                        env_step<T, MODEL>(obs[j], t[j], quota[j], z[j], rr[j], KK[j], sg[j], p.C, p.Tmax, o2[j], r2[j], d2[j],
                                           t2[j], dk);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (AUTO) {
                    obs[j] = o2[j];
                    rew[j] = r2[j];
                    dn[j] = d2[j];
                    t[j] = t2[j];
                    er[j] = er[j] + r2[j];
                    fin[j] = d2[j] && live[j];
                } else {
                    const bool lv = live[j];
                    obs[j] = lv ? o2[j] : obs[j];
                    rew[j] = lv ? r2[j] : (T)0;
                    dn[j] = lv ? d2[j] : dn[j];
                    t[j] = lv ? t2[j] : t[j];
                    er[j] = lv ? er[j] + r2[j] : er[j];
                    fin[j] = d2[j] && lv;
                }
            }
            if (traj && active) {
                T* row = traj + (int64_t)s * 4 * n;
                T dn_rec[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) dn_rec[j] = fin[j] ? (T)1 : (T)0;
                store4<T>(row, base, n, full, obs_in);
                store4<T>(row + n, base, n, full, act_rec);
                store4<T>(row + 2 * n, base, n, full, rew);
                store4<T>(row + 3 * n, base, n, full, dn_rec);
            }
            // ---- wave-ballot: only waves holding a finished env do the record / reset work
            if (__any(fin[0] | fin[1] | fin[2] | fin[3])) {
                record_tile<T>(fin, er, t, acc);
                if (AUTO && zoo_mixed) {      // growth_models.py:200: a new model for the next episode
                    if (redraw_kinds(seed, env_offset + (uint64_t)base, step_counter, kStreamAutoReset, p.kinds, p.n_models,
                                     fin, kind))
                        kind_dirty = true;
                }
                if constexpr (AUTO && kPerEnv) {
                    __shared__ RedrawSlot<T> rwin[4 * kRedrawSlots];       // (fishing-v4 instantiations only: 8 / 16 KB)
                    if (redraw_tile_compact<T, MODEL>(seed, env_offset + (uint64_t)base, step_counter, kStreamAutoReset,
                                                      p.K_mean, p.r_mean, p.sigma_p, p.x0, fin, KK, rr, obs, t,
                                                      rwin + (threadIdx.x >> 6) * kRedrawSlots, lane))
                        kr_dirty = true;
                    if (stamped) {          // an auto-reset env is dated by its year counter again
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            stamp_dirty |= fin[j] && st[j] != 0;
                            st[j] = fin[j] ? 0 : st[j];
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool f = fin[j];
                    if (AUTO) {
                        const T ro = kPerEnv ? reset_obs<T, MODEL>(p.x0, KK[j]) : robs_scalar;
                        er[j] = f ? (T)0 : er[j];
                        obs[j] = f ? ro : obs[j];
                        t[j] = f ? 0 : t[j];
                        // dn[j] keeps this step's flag for the final done output
                    } else {
                        live[j] = live[j] && !f;
                    }
                }
                if (!AUTO && __all(!(live[0] | live[1] | live[2] | live[3]))) break;
            }
        }

        if (active) {
            store4<T>(b.obs, base, n, full, obs);
            store_t4(b.t, (p.flags & FISHING_FLAG_T_U8) != 0, base, n, full, t);
            if (b.ep_return) store4<T>(b.ep_return, base, n, full, er);
            if (b.reward) store4<T>(b.reward, base, n, full, rew);
            if (kPerEnv && kr_dirty && !derived) {
                store4<T>(b.K, base, n, full, KK);
                store4<T>(b.r, base, n, full, rr);
            }
            if (stamp_dirty) store4<int32_t>(b.stamp, base, n, full, st);
            if (zoo_drift) store4<T>(b.r, base, n, full, rr);
            if (zoo_mixed && kind_dirty) store4<int32_t>(b.model_idx, base, n, full, kind);
            if (b.done) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (base + j < n) b.done[base + j] = (uint8_t)dn[j];
            }
        }
        if (b.done_bits) {
            const uint32_t nibble = (uint32_t)(dn[0] && base + 0 < n) | ((uint32_t)(dn[1] && base + 1 < n) << 1) |
                                    ((uint32_t)(dn[2] && base + 2 < n) << 2) |
                                    ((uint32_t)(dn[3] && base + 3 < n) << 3);
            const uint64_t word = ballot_tile_words(nibble, lane);
            const int64_t wave_env0 = (tile * blockDim.x + (threadIdx.x & ~(kWave - 1))) * kEnvsPerThread;
            const int64_t widx = (wave_env0 >> 6) + lane;
            if (lane < 4 && (widx << 6) < n) b.done_bits[widx] = word;
        }
    }

    if (b.partials) add_block_partials<4>(acc, b.partials);
}

template <typename T, int MODEL>
int launch_rollout_policy(int policy, const ParamsT<T>& pt, const BuffersT<T>& bt, int64_t n, uint64_t env_offset,
                          T policy_param, int32_t Tsteps, T* traj, uint64_t seed, uint64_t step_counter,
                          int noise_on, int blocks, int threads, hipStream_t s, const T* pparams = nullptr) {
    const DivK dk = make_divk((double)pt.K);
    if (pparams)        // one parameter per env: the run-time-policy, auto-resetting instantiation of this model (rollout_params_impl checks)
        return launch_kernel(rollout_kernel<T, MODEL, -1, true, false, true>, blocks, threads, s, pt, bt, n, env_offset,
                             policy_param, Tsteps, traj, seed, step_counter, noise_on, policy, dk, pparams);
#define FISHING_LAUNCH_ROLLOUT(POL)                                                                              \
    ((pt.flags & FISHING_FLAG_AUTO_RESET)                                                                        \
         ? launch_kernel(rollout_kernel<T, MODEL, POL, true>, blocks, threads, s, pt, bt, n, env_offset,         \
                         policy_param, Tsteps, traj, seed, step_counter, noise_on, policy, dk, (const T*)nullptr) \
         : launch_kernel(rollout_kernel<T, MODEL, -1, false>, blocks, threads, s, pt, bt, n, env_offset,         \
                         policy_param, Tsteps, traj, seed, step_counter, noise_on, policy, dk, (const T*)nullptr))
    if constexpr (is_zoo_tag(MODEL)) {
        return FISHING_LAUNCH_ROLLOUT(-1);          // run-time policy switch
    } else {
        if constexpr (sizeof(T) == 4 && MODEL != FISHING_MODEL_V4) {
            // the common request -- float32, auto-reset, K a power of two -- on the twins without divisions (KP2C)
            if ((pt.flags & FISHING_FLAG_AUTO_RESET) && dk.pow2) {
#define FISHING_LAUNCH_ROLLOUT_KP2(POL)                                                                          \
    launch_kernel(rollout_kernel<T, MODEL, POL, true, true>, blocks, threads, s, pt, bt, n, env_offset, policy_param, \
                  Tsteps, traj, seed, step_counter, noise_on, policy, dk, (const T*)nullptr)
                switch (policy) {
                    case FISHING_POLICY_RANDOM: return FISHING_LAUNCH_ROLLOUT_KP2(FISHING_POLICY_RANDOM);
                    case FISHING_POLICY_CONSTANT: return FISHING_LAUNCH_ROLLOUT_KP2(FISHING_POLICY_CONSTANT);
                    case FISHING_POLICY_ESCAPEMENT: return FISHING_LAUNCH_ROLLOUT_KP2(FISHING_POLICY_ESCAPEMENT);
                    default: return FISHING_LAUNCH_ROLLOUT_KP2(FISHING_POLICY_MSY);
                }
#undef FISHING_LAUNCH_ROLLOUT_KP2
            }
        }
        switch (policy) {
            case FISHING_POLICY_RANDOM: return FISHING_LAUNCH_ROLLOUT(FISHING_POLICY_RANDOM);
            case FISHING_POLICY_CONSTANT: return FISHING_LAUNCH_ROLLOUT(FISHING_POLICY_CONSTANT);
            case FISHING_POLICY_ESCAPEMENT: return FISHING_LAUNCH_ROLLOUT(FISHING_POLICY_ESCAPEMENT);
            default: return FISHING_LAUNCH_ROLLOUT(FISHING_POLICY_MSY);
        }
    }
#undef FISHING_LAUNCH_ROLLOUT
}

template <typename T>
int rollout_impl(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b, int32_t policy,
                 double policy_param, int32_t Tsteps, void* traj, uint64_t seed, uint64_t step_counter,
                 fishing_stream_t stream, const void* policy_params = nullptr, bool per_env = false) {
    const int rc = check_common(p, n, env_offset, b);
    if (rc != FISHING_OK) return rc;
    if (policy < FISHING_POLICY_RANDOM || policy > FISHING_POLICY_MSY) return FISHING_ERR_POLICY;
    if (Tsteps < 0) return FISHING_ERR_SIZE;
    if (per_env) {      // fishing_rollout_params_*
        if (!policy_params) return FISHING_ERR_NULL;
        if (((uintptr_t)policy_params) & 15u) return FISHING_ERR_ALIGN;
        if (policy == FISHING_POLICY_RANDOM) return FISHING_ERR_POLICY;             // (takes no parameter)
        if (!(p->flags & FISHING_FLAG_AUTO_RESET)) return FISHING_ERR_UNSUPPORTED;  // (the frozen-episode form: scalar parameter only)
    }
    // Without auto-reset the rollout FREEZES a finished env -- its year counter stops while the step counter runs on,
    // and with it the rule that dates the env's episode (derive_model_error): such a rollout needs the r / K arrays.
    if (p->model == FISHING_MODEL_V4 && (p->flags & FISHING_FLAG_V4_DERIVED) && !(p->flags & FISHING_FLAG_AUTO_RESET))
        return FISHING_ERR_UNSUPPORTED;
    if (traj && (((uintptr_t)traj) & 15u)) return FISHING_ERR_ALIGN;
    if (traj && (n & 3)) return FISHING_ERR_ALIGN;  // rows of the record must stay 16-byte aligned
    if (n == 0 || Tsteps == 0) return FISHING_OK;
    const ParamsT<T> pt = narrow_params<T>(*p);
    const BuffersT<T> bt = typed_buffers<T>(*b);
    bool quiet = (p->sigma == 0.0 && !b->sigma);
    if (p->model == FISHING_MODEL_V11) {
        quiet = !b->sigma;
        for (int k = 0; k < FISHING_N_KINDS; ++k) quiet = quiet && p->zoo[k].sigma == 0.0;
    }
    const int noise_on = !quiet;
    int blocks, threads;
    launch_shape(p, n, blocks, threads);
    hipStream_t s = (hipStream_t)stream;
    const T pp = (T)policy_param;
    return with_model_tag(p->model, [&](auto tag) {
        return launch_rollout_policy<T, decltype(tag)::value>(policy, pt, bt, n, env_offset, pp, Tsteps, (T*)traj, seed,
                                                              step_counter, noise_on, blocks, threads, s,
                                                              (const T*)policy_params);
    });
}


// ---------------------------------------------------------------- fused step_many: K caller-driven steps per launch
// fishing_step_fused_*: what n_steps fishing_step_* launches do, in one launch -- a thread keeps its four envs'
// (obs, t, r, K, ep_return) in registers, step s reads the caller's action row (s % ring_len) of the [R, n] ring
// and, optionally, writes that step's reward / done row.  HBM traffic per env-step drops from 25 B to 9 B
// (action 4 R, reward 4 + done 1 W) or 4 B (no per-step outputs); what it removes above all is the ~2.7 us a
// dependent launch costs whatever it moves, which bounds the per-step path below N ~ 2^20.
// The action rows do not depend on the state, so they are prefetched kPrefetch steps ahead: at N = 2^19 only two
// waves share a SIMD and nothing else would hide the load latency.
// step() semantics, not the rollout's: without FISHING_FLAG_AUTO_RESET a finished env keeps being stepped.
// Same Philox counters, same arithmetic, same record as the per-step kernels: identical bits.
template <typename T>
struct FusedArgs {
    T* obs;
    const void* action;      // row 0 of the ring
    T* reward;               // last step's values (FishingBuffers.reward / .done), nullable
    uint8_t* done;
    int32_t* t;
    T* r;
    T* K;
    T* ep_return;
    double* partials;
    const uint64_t* counter;
    const T* sigma_arr;
    T* reward_steps;         // [n_steps][out_stride], nullable
    uint8_t* done_steps;     // [n_steps][out_stride], nullable
    int32_t* stamp;          // fishing-v4 derived: per-env origin stamps (FishingBuffers.v4_stamp), nullable
    int64_t action_stride, out_stride;
    int32_t ring_len, n_steps;
    T pr, pK, sigma, C, x0, r_mean, K_mean, sigma_p;
    int32_t Tmax, n_actions;
    uint32_t auto_reset, t8, derived, drift;
    int32_t noise;           // kNoiseNone / kNoisePhilox
    uint64_t origin_step, origin_counter;
    GrowthT<T> growth;
    T alpha;
    DivK dk;                 // exact x / K as a multiply when the scalar K is a power of two (never for per-env K)
};

// fishing-v11 only: the growth kind in force per env, the model list it is redrawn from and the per-kind parameters.
// Every other model passes the empty struct (one byte of kernel arguments).
template <typename T>
struct FusedMixedArgs {
    int32_t* model_idx;
    int32_t n_models;
    int32_t kinds[FISHING_N_KINDS];
    GrowthT<T> zoo[FISHING_N_KINDS];
};
struct FusedNoExtra {};
template <typename T, int MODEL>
using FusedExtra = std::conditional_t<MODEL == kModelZooMixed, FusedMixedArgs<T>, FusedNoExtra>;

constexpr int kPrefetch = 4;

// RAGGED = false: n is a whole number of 1024-env tiles -- every access an unconditional 16-byte one.  That is not
// only shorter: with the per-thread `full ? vector : element-wise` choice in the code, the compiler merges the two
// paths' results right behind each prefetch load and has to wait for it there (`s_waitcnt vmcnt(0)` after every
// global_load: the prefetch hid nothing, waves sat 36 % of their cycles on s_waitcnt).  RAGGED = true is the same
// body for the < 1024-env tail, one workgroup.
// KP2 = true: the scalar K is a power of two, x / K is the exact multiply x * (1 / K) -- as a compile-time fact, so that
// the four envs' arithmetic is one basic block the scheduler can interleave (a run-time flag puts a uniform branch
// around every division and the envs' chains execute one after the other: it is the latency-bound small batches,
// two waves per SIMD at N = 2^19, that pay for that).
template <typename T, int MODEL, bool RAGGED, bool KP2 = false>
__global__ void __launch_bounds__(256)
step_fused_kernel(const FusedArgs<T> a, const FusedExtra<T, MODEL> ex, const int64_t n, const uint64_t env_offset,
                  const uint64_t seed, const uint64_t step_counter_arg) {
    constexpr bool kPerEnv = (MODEL == FISHING_MODEL_V4);
    constexpr bool kZoo = is_zoo_tag(MODEL);
    constexpr bool zoo_mixed = (MODEL == kModelZooMixed);              // fishing-v11: growth kind per env
    constexpr int kZooKind = (kZoo && !zoo_mixed) ? (MODEL - kModelZoo) : -1;
    static_assert(MODEL != kModelZooRT, "the run-time-kind tag belongs to the general step kernel");
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t step_counter0 = a.counter ? (*a.counter + step_counter_arg) : step_counter_arg;
    const bool auto_reset = a.auto_reset != 0;
    const bool derived = kPerEnv && a.derived != 0;
    uint64_t origin_step = a.origin_step, origin_counter = a.origin_counter;
    if (derived) device_origin(a.counter, origin_step, origin_counter);
    constexpr bool kMayDrift = MODEL == kModelZoo + FISHING_KIND_BEVERTON_HOLT || MODEL == kModelZooRT;
    const bool drift = kMayDrift && a.drift != 0;
    const bool t8 = a.t8 != 0;
    const DivK dk = RAGGED ? a.dk : (KP2 ? DivK{true, a.dk.inv_f, a.dk.inv_d} : DivK{false, 0.0f, 0.0});
    const int64_t tile_envs = (int64_t)blockDim.x * kEnvsPerThread;
    const int64_t ntiles = (n + tile_envs - 1) / tile_envs;
    double acc[kPartialFields] = {0.0, 0.0, 0.0, 0.0};
    const T robs_scalar = reset_obs<T, MODEL>(a.x0, a.pK);
    // fishing-v0: the index -> quota map as an LDS table (see rollout_kernel).  The caller's indices are not validated
    // (quirk B11: 100, 150, -3 are legal and map through the same arithmetic): a wave with an index outside [0, n_actions) --
    // or a batch with more than kQuotaLut actions -- evaluates the arithmetic for that step instead.
    constexpr bool kLutForm = (MODEL == FISHING_MODEL_V0) && !RAGGED;
    constexpr int kQuotaLut = 1024;
    __shared__ T quota_lut[kLutForm ? kQuotaLut : 1];
    const bool use_lut = kLutForm && a.n_actions > 0 && a.n_actions <= kQuotaLut;
    if constexpr (kLutForm) {
        if (use_lut) {
            for (int i = threadIdx.x; i < a.n_actions; i += blockDim.x) quota_lut[i] = quota_int<T>(i, a.n_actions, a.pK);
            __syncthreads();
        }
    }

    // (fishing-v11: see rollout_kernel)
    __shared__ alignas(16) T zoo_lut[zoo_mixed ? kZooLutSize : 4];
    if constexpr (zoo_mixed) {
        if (threadIdx.x < kWave) zoo_lut_fill<T>(zoo_lut, ex.zoo);
        __syncthreads();
    }

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = (tile * blockDim.x + threadIdx.x) * kEnvsPerThread;
        const bool active = RAGGED ? base < n : true;
        const bool full = RAGGED ? base + kEnvsPerThread <= n : true;
        T obs[4], rr[4], KK[4], sg[4], er[4], rew[4];
        int32_t t[4];
        int32_t kind[4];
        bool dn[4];
        int32_t st[4] = {0, 0, 0, 0};       // fishing-v4 derived: per-env origin stamps (FishingBuffers.v4_stamp)
        const bool stamped = derived && a.stamp != nullptr;
        bool stamp_dirty = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            obs[j] = (T)0;
            t[j] = 0;
            kind[j] = FISHING_KIND_BEVERTON_HOLT;
            rr[j] = a.pr;
            KK[j] = a.pK;
            sg[j] = a.sigma;
            er[j] = (T)0;
            rew[j] = (T)0;
            dn[j] = false;
        }
        if (active) {
            load4<T>(a.obs, base, n, full, obs, (T)0);
            load_t4(a.t, t8, base, n, full, t);
            if (kPerEnv && !derived) {
                load4<T>(a.r, base, n, full, rr, a.pr);
                load4<T>(a.K, base, n, full, KK, a.pK);
            }
            if (drift) load4<T>(a.r, base, n, full, rr, a.pr);
            if constexpr (zoo_mixed) load4<int32_t>(ex.model_idx, base, n, full, kind, FISHING_KIND_BEVERTON_HOLT);
            if (a.sigma_arr) load4<T>(a.sigma_arr, base, n, full, sg, a.sigma);
            if (a.ep_return) load4<T>(a.ep_return, base, n, full, er, (T)0);
            if (stamped) load4<int32_t>(a.stamp, base, n, full, st, 0);
        }
        // the action rows: kPrefetch steps in flight
        float pf_f[kPrefetch][4];
        int32_t pf_i[kPrefetch][4];
        auto load_action = [&](int32_t s, float (&af)[4], int32_t (&ai)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                af[j] = -1.0f;
                ai[j] = 0;
            }
            if (!active) return;
            const int64_t row = (int64_t)(s % a.ring_len) * a.action_stride;
            if (MODEL == FISHING_MODEL_V0) load4<int32_t>((const int32_t*)a.action + row, base, n, full, ai, 0);
            else load4<float>((const float*)a.action + row, base, n, full, af, -1.0f);
        };
#pragma unroll
        for (int u = 0; u < kPrefetch; ++u)
            if (u < a.n_steps) load_action(u, pf_f[u], pf_i[u]);
        if (derived) {      // once per launch; the redraws below keep (K, r) current from then on
#pragma unroll
            for (int j = 0; j < 4; ++j)
                derive_model_error<T>(seed, env_offset + (uint64_t)base + j, step_counter0, t[j], origin_step,
                                      origin_counter, a.K_mean, a.r_mean, a.sigma_p, KK[j], rr[j], st[j]);
        }
        const uint64_t quad = (env_offset + (uint64_t)base) >> 2;
        bool kr_dirty = false;
        bool kind_dirty = false;

        for (int32_t s0 = 0; s0 < a.n_steps; s0 += kPrefetch) {
#pragma unroll
            for (int u = 0; u < kPrefetch; ++u) {
                const int32_t s = s0 + u;
                if (s >= a.n_steps) break;          // wave-uniform
                float a_f[4];
                int32_t a_i[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a_f[j] = pf_f[u][j];
                    a_i[j] = pf_i[u][j];
                }
                if (s + kPrefetch < a.n_steps) load_action(s + kPrefetch, pf_f[u], pf_i[u]);
                const uint64_t step_counter = step_counter0 + (uint64_t)s;
                T z[4] = {(T)0, (T)0, (T)0, (T)0};
                uint64_t seed_s = seed;     // keep the Philox key schedule next to its rounds instead of in 20 SGPRs
                asm volatile("" : "+s"(seed_s));    // held across the whole step loop (the kernel is short of them)
                if (a.noise == kNoisePhilox) {
                    float zq[4];
                    noise_quad(seed_s, quad, step_counter, zq);
#pragma unroll
                    for (int j = 0; j < 4; ++j) z[j] = (T)zq[j];
                }
                T o2[4];
                int32_t t2[4];
                bool fresh[4];
                bool stepped = false;
                if constexpr (zoo_mixed) {
                    if (!a.sigma_arr) {     // wave-uniform: every env's coefficients from the LDS table, as the per-step kernel
                        T xh[4], hv[4], xn[4];
                        int kk[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const T quota = quota_cts<T>((T)a_f[j], KK[j]);
                            fresh[j] = auto_reset || !((t[j] > a.Tmax) || ((obs[j] + (T)1) * KK[j] <= (T)0));
                            const T x = (obs[j] + (T)1) * KK[j];
                            hv[j] = (quota < x) ? quota : x;
                            const T d = x - hv[j];
                            xh[j] = d;       // (max(d, 0.0) is the identity here: stock_after_harvest)
                            xn[j] = (T)0;
                            kk[j] = (kind[j] >= 0 && kind[j] < FISHING_N_KINDS) ? kind[j] : FISHING_KIND_BEVERTON_HOLT;
                        }
                        zoo_draw_lut_tile<T, 4>(kk, xh, z, ex.zoo, zoo_lut, xn);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            o2[j] = xn[j] / KK[j] - (T)1;
                            rew[j] = ((T)0 > hv[j]) ? (T)0 : hv[j];
                            t2[j] = t[j] + 1;
                            dn[j] = (t2[j] > a.Tmax) || (xn[j] <= (T)0);
                            if (RAGGED) dn[j] = dn[j] && (base + j < n);
                            fresh[j] = fresh[j] && dn[j];
                            er[j] = er[j] + rew[j];
                            obs[j] = o2[j];
                            t[j] = t2[j];
                        }
                        stepped = true;
                    }
                }
                if (!stepped) {
                T q_lut[4] = {(T)0, (T)0, (T)0, (T)0};
                bool lut_ok = false;
                if constexpr (kLutForm) {
                    const uint32_t lim = (uint32_t)a.n_actions;
                    lut_ok = use_lut && !__any(((uint32_t)a_i[0] >= lim) | ((uint32_t)a_i[1] >= lim) | ((uint32_t)a_i[2] >= lim) |
                                               ((uint32_t)a_i[3] >= lim));          // wave-uniform
                    if (lut_ok) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) q_lut[j] = quota_lut[a_i[j]];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    T quota;
                    if constexpr (kLutForm) {
                        if (lut_ok) {
                            quota = q_lut[j];
                        } else {
                            asm volatile("");       // (a real branch: the division must not be evaluated speculatively and selected)
                            quota = quota_int<T>(a_i[j], a.n_actions, KK[j]);
                        }
                    } else {
                        quota = (MODEL == FISHING_MODEL_V0) ? quota_int<T>(a_i[j], a.n_actions, KK[j]) : quota_cts<T>((T)a_f[j], KK[j]);
                    }
                    fresh[j] = auto_reset || !((t[j] > a.Tmax) || ((obs[j] + (T)1) * KK[j] <= (T)0));
                    if constexpr (zoo_mixed) {          // per-env sigma: a straight per-lane switch over the growth functions
                        env_step_zoo_mixed<T>(obs[j], t[j], quota, z[j], kind[j], ex.zoo, sg[j], KK[j], a.Tmax, o2[j], rew[j], dn[j],
                                              t2[j]);
                    } else if constexpr (kZoo) {
                        GrowthT<T> P = a.growth;
                        if (a.sigma_arr) P.sigma = sg[j];
                        if (drift) {                     // growth_models.py:151: drift first, then draw
                            rr[j] = rr[j] + a.alpha;
                            P.r = rr[j];
                            env_step_zoo<T, kZooKind, true>(obs[j], t[j], quota, z[j], kZooKind, P, KK[j], a.Tmax, o2[j],
                                                            rew[j], dn[j], t2[j]);
                        } else {
                            env_step_zoo<T, kZooKind, false>(obs[j], t[j], quota, z[j], kZooKind, P, KK[j], a.Tmax, o2[j],
                                                             rew[j], dn[j], t2[j]);
                        }
                    } else {
                        env_step<T, MODEL>(obs[j], t[j], quota, z[j], rr[j], KK[j], sg[j], a.C, a.Tmax, o2[j], rew[j], dn[j],
                                           t2[j], dk);
                    }
                    if (RAGGED) dn[j] = dn[j] && (base + j < n);
                    fresh[j] = fresh[j] && dn[j];
                    er[j] = er[j] + rew[j];
                    obs[j] = o2[j];
                    t[j] = t2[j];
                }
                }
                if (active) {
                    if (a.reward_steps) store4<T, 1>(a.reward_steps + (int64_t)s * a.out_stride, base, n, full, rew);
                    if (a.done_steps) {
                        uint8_t* row = a.done_steps + (int64_t)s * a.out_stride;
                        if (full) {
                            __builtin_nontemporal_store((uint32_t)dn[0] | ((uint32_t)dn[1] << 8) | ((uint32_t)dn[2] << 16) |
                                                            ((uint32_t)dn[3] << 24),
                                                        reinterpret_cast<uint32_t*>(row + base));
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (base + j < n) row[base + j] = (uint8_t)dn[j];
                        }
                    }
                }
                if (__any(dn[0] | dn[1] | dn[2] | dn[3])) {
                    if (a.ep_return) {
                        record_tile<T>(fresh, er, t, acc);
#pragma unroll
                        for (int j = 0; j < 4; ++j) er[j] = (dn[j] && auto_reset) ? (T)0 : er[j];
                    }
                    if (auto_reset) {
                        if constexpr (zoo_mixed) {      // growth_models.py:200: a new model for the next episode
                            if (redraw_kinds(seed_s, env_offset + (uint64_t)base, step_counter, kStreamAutoReset, ex.kinds,
                                             ex.n_models, dn, kind))
                                kind_dirty = true;
                        }
                        if constexpr (kPerEnv) {      // the next episode's (K, r): the draw a later derivation would re-make
                            __shared__ RedrawSlot<T> rwin[4 * kRedrawSlots];
                            if (redraw_tile_compact<T, MODEL>(seed_s, env_offset + (uint64_t)base, step_counter, kStreamAutoReset,
                                                              a.K_mean, a.r_mean, a.sigma_p, a.x0, dn, KK, rr, obs, t,
                                                              rwin + (threadIdx.x >> 6) * kRedrawSlots, lane))
                                kr_dirty = true;
                            if (stamped) {          // an auto-reset env is dated by its year counter again
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    stamp_dirty |= dn[j] && st[j] != 0;
                                    st[j] = dn[j] ? 0 : st[j];
                                }
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                obs[j] = dn[j] ? robs_scalar : obs[j];
                                t[j] = dn[j] ? 0 : t[j];
                            }
                        }
                    }
                }
            }
        }

        if (active) {
            store4<T>(a.obs, base, n, full, obs);
            store_t4(a.t, t8, base, n, full, t);
            if (a.ep_return) store4<T>(a.ep_return, base, n, full, er);
            if (a.reward) store4<T>(a.reward, base, n, full, rew);
            if (kPerEnv && kr_dirty && !derived) {
                store4<T>(a.K, base, n, full, KK);
                store4<T>(a.r, base, n, full, rr);
            }
            if (stamp_dirty) store4<int32_t>(a.stamp, base, n, full, st);
            if (drift) store4<T>(a.r, base, n, full, rr);
            if constexpr (zoo_mixed) {
                if (kind_dirty) store4<int32_t>(ex.model_idx, base, n, full, kind);
            }
            if (a.done) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (base + j < n) a.done[base + j] = (uint8_t)dn[j];
            }
        }
    }
    if (a.partials) add_block_partials<4>(acc, a.partials);
}

template <typename T>
int step_fused_impl(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b, int64_t action_stride,
                    int32_t ring_len, int32_t n_steps, void* reward_steps, uint8_t* done_steps, int64_t out_stride,
                    uint64_t seed, uint64_t step_counter, fishing_stream_t stream) {
    const int rc = check_common(p, n, env_offset, b);
    if (rc != FISHING_OK) return rc;
    if (!b->action) return FISHING_ERR_NULL;
    if (ring_len <= 0 || n_steps < 0 || action_stride < 0 || out_stride < 0) return FISHING_ERR_SIZE;
    if (ring_len > 1 && (action_stride & 3)) return FISHING_ERR_ALIGN;
    if ((reward_steps || done_steps) && (out_stride < n || (out_stride & 15))) return FISHING_ERR_ALIGN;
    if (misaligned(reward_steps) || misaligned(done_steps)) return FISHING_ERR_ALIGN;
    // the streams only the per-step kernels produce
    if (b->z_ext || b->terminal_obs || b->done_bits) return FISHING_ERR_UNSUPPORTED;
    if (p->model == FISHING_MODEL_V11 && !b->model_idx) return FISHING_ERR_NULL;
    if (p->launch_threads != 0 && p->launch_threads != 256) return FISHING_ERR_UNSUPPORTED;
    if (n == 0 || n_steps == 0) return FISHING_OK;
    const ParamsT<T> pt = narrow_params<T>(*p);
    const BuffersT<T> bt = typed_buffers<T>(*b);
    const int noise = noise_mode(p, b);
    const bool per_env = p->model == FISHING_MODEL_V4;
    const FusedArgs<T> a{bt.obs, bt.action, bt.reward, bt.done, bt.t, bt.r, bt.K, bt.ep_return, bt.partials, bt.counter,
                         bt.sigma, (T*)reward_steps, done_steps, bt.stamp, action_stride, out_stride, ring_len, n_steps, pt.r, pt.K,
                         pt.sigma, pt.C, pt.x0, pt.r_mean, pt.K_mean, pt.sigma_p, pt.Tmax, pt.n_actions,
                         (uint32_t)(p->flags & FISHING_FLAG_AUTO_RESET), (uint32_t)((p->flags & FISHING_FLAG_T_U8) != 0),
                         (uint32_t)(per_env && (p->flags & FISHING_FLAG_V4_DERIVED)), (uint32_t)(p->model == FISHING_MODEL_V10),
                         noise, pt.origin_step, pt.origin_counter, pt.growth, pt.alpha,
                         // x / K as an exact multiply changes no bit, so the per-step kernels' true division agrees
                         per_env ? DivK{false, 0.0f, 0.0} : make_divk((double)pt.K)};
    // whole tiles through the unconditional instantiation, the < 1024-env tail through one workgroup of the ragged one
    const int64_t tile = 256 * kEnvsPerThread;
    const int64_t n_full = (n / tile) * tile;
    const bool t8 = (p->flags & FISHING_FLAG_T_U8) != 0;
    return with_model_tag(p->model, [&](auto tag) {
        constexpr int kTag = decltype(tag)::value;
        FusedExtra<T, kTag> ex{};
        if constexpr (kTag == kModelZooMixed) {
            ex.model_idx = bt.model_idx;
            ex.n_models = pt.n_models;
            for (int k = 0; k < FISHING_N_KINDS; ++k) {
                ex.kinds[k] = pt.kinds[k];
                ex.zoo[k] = pt.zoo[k];
            }
        }
        if constexpr (sizeof(T) == 8) {
            // the float64 parity layout is not the fast path: one ragged-capable instantiation over the whole batch
            int blocks, threads;
            launch_shape(p, n, blocks, threads);
            return launch_kernel(step_fused_kernel<T, kTag, true>, blocks, 256, (hipStream_t)stream, a, ex, n,
                                 (uint64_t)env_offset, seed, step_counter);
        } else {
            if (n_full > 0) {
                int blocks, threads;
                launch_shape(p, n_full, blocks, threads);
                int rc2;
                if constexpr (kTag != FISHING_MODEL_V4 && !is_zoo_tag(kTag)) {
                    rc2 = a.dk.pow2 ? launch_kernel(step_fused_kernel<T, kTag, false, true>, blocks, 256, (hipStream_t)stream, a,
                                                    ex, n_full, (uint64_t)env_offset, seed, step_counter)
                                    : launch_kernel(step_fused_kernel<T, kTag, false, false>, blocks, 256, (hipStream_t)stream, a,
                                                    ex, n_full, (uint64_t)env_offset, seed, step_counter);
                } else {    // per-env K (fishing-v4) and the zoo (x / K only in the obs map: its growth functions divide by their own K)
                    rc2 = launch_kernel(step_fused_kernel<T, kTag, false, false>, blocks, 256, (hipStream_t)stream, a, ex,
                                        n_full, (uint64_t)env_offset, seed, step_counter);
                }
                if (rc2 != 0 || n_full == n) return rc2;
            }
            FusedArgs<T> tl = a;
            const int64_t o = n_full;
            tl.obs = a.obs + o;
            tl.action = (const char*)a.action + 4 * o;
            tl.reward = a.reward ? a.reward + o : nullptr;
            tl.done = a.done ? a.done + o : nullptr;
            tl.t = t8 ? reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(a.t) + o) : a.t + o;
            tl.r = a.r ? a.r + o : nullptr;
            tl.K = a.K ? a.K + o : nullptr;
            tl.ep_return = a.ep_return ? a.ep_return + o : nullptr;
            tl.sigma_arr = a.sigma_arr ? a.sigma_arr + o : nullptr;
            tl.reward_steps = a.reward_steps ? a.reward_steps + o : nullptr;
            tl.done_steps = a.done_steps ? a.done_steps + o : nullptr;
            tl.stamp = a.stamp ? a.stamp + o : nullptr;
            if constexpr (kTag == kModelZooMixed) ex.model_idx = ex.model_idx + o;
            // (the tail adds its record to workgroup slot 0, as the per-step path.s tail launch does)
            return launch_kernel(step_fused_kernel<T, kTag, true>, 1, 256, (hipStream_t)stream, tl, ex, n - n_full,
                                 (uint64_t)(env_offset + n_full), seed, step_counter);
        }
    });
}

}  // namespace fishing

extern "C" {

int fishing_rollout_f32(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                        int32_t policy, double policy_param, int32_t T, void* traj, uint64_t seed,
                        uint64_t step_counter, fishing_stream_t stream) {
    return fishing::rollout_impl<float>(p, n, env_offset, b, policy, policy_param, T, traj, seed, step_counter, stream);
}
int fishing_rollout_f64(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                        int32_t policy, double policy_param, int32_t T, void* traj, uint64_t seed,
                        uint64_t step_counter, fishing_stream_t stream) {
    return fishing::rollout_impl<double>(p, n, env_offset, b, policy, policy_param, T, traj, seed, step_counter, stream);
}
int fishing_rollout_params_f32(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                               int32_t policy, const void* policy_params, int32_t T, void* traj, uint64_t seed,
                               uint64_t step_counter, fishing_stream_t stream) {
    return fishing::rollout_impl<float>(p, n, env_offset, b, policy, 0.0, T, traj, seed, step_counter, stream, policy_params, true);
}
int fishing_rollout_params_f64(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                               int32_t policy, const void* policy_params, int32_t T, void* traj, uint64_t seed,
                               uint64_t step_counter, fishing_stream_t stream) {
    return fishing::rollout_impl<double>(p, n, env_offset, b, policy, 0.0, T, traj, seed, step_counter, stream, policy_params, true);
}

int fishing_step_fused_f32(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                           int64_t action_stride, int32_t ring_len, int32_t n_steps, void* reward_steps,
                           uint8_t* done_steps, int64_t out_stride, uint64_t seed, uint64_t step_counter,
                           fishing_stream_t stream) {
    return fishing::step_fused_impl<float>(p, n, env_offset, b, action_stride, ring_len, n_steps, reward_steps, done_steps,
                                           out_stride, seed, step_counter, stream);
}
int fishing_step_fused_f64(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                           int64_t action_stride, int32_t ring_len, int32_t n_steps, void* reward_steps,
                           uint8_t* done_steps, int64_t out_stride, uint64_t seed, uint64_t step_counter,
                           fishing_stream_t stream) {
    return fishing::step_fused_impl<double>(p, n, env_offset, b, action_stride, ring_len, n_steps, reward_steps, done_steps,
                                            out_stride, seed, step_counter, stream);
}

}  // extern "C"
