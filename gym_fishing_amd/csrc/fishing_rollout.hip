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
#include "fishing_common.h"

namespace fishing {

template <typename T, int MODEL, int POLICY, bool AUTO>
__global__ void __launch_bounds__(256)
rollout_kernel(const ParamsT<T> p, const BuffersT<T> b, const int64_t n, const uint64_t env_offset,
               const T policy_param, const int32_t Tsteps, T* __restrict__ traj, const uint64_t seed,
               const uint64_t step_counter_arg, const int noise_on, const int policy_rt, const DivK dk_arg) {
    const uint64_t step_counter0 = b.counter ? (*b.counter + step_counter_arg) : step_counter_arg;
    constexpr bool kPerEnv = (MODEL == FISHING_MODEL_V4);
    const DivK dk = kPerEnv ? DivK{false, 0.0f, 0.0} : dk_arg;      // per-env K keeps the true division
    // POLICY >= 0: compile-time policy (v0/v1/v2/v4); POLICY < 0: wave-uniform run-time policy (zoo,
    // to keep the number of instantiations of the transcendental-heavy bodies small)
    const int policy = (POLICY >= 0) ? POLICY : policy_rt;
    const bool kNeedWords = (policy == FISHING_POLICY_RANDOM);
    constexpr bool kZoo = is_zoo_tag(MODEL);
    constexpr int kZooKind = (kZoo && MODEL != kModelZooMixed) ? (MODEL - kModelZoo) : -1;
    constexpr bool zoo_mixed = (MODEL == kModelZooMixed);
    const bool zoo_drift = kZoo && p.model == FISHING_MODEL_V10;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t tile_envs = (int64_t)blockDim.x * kEnvsPerThread;
    const int64_t ntiles = (n + tile_envs - 1) / tile_envs;
    double acc[kPartialFields] = {0.0, 0.0, 0.0, 0.0};

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = (tile * blockDim.x + threadIdx.x) * kEnvsPerThread;
        const bool active = base < n;
        const bool full = base + kEnvsPerThread <= n;

        T obs[4], rr[4], KK[4], sg[4], er[4], rew[4];
        int32_t t[4];
        bool dn[4], live[4];    // live: a real env whose episode is still running
        int32_t kind[4];
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
            if (kPerEnv) {
                load4<T>(b.r, base, n, full, rr, p.r);
                load4<T>(b.K, base, n, full, KK, p.K);
            }
            if (zoo_drift) load4<T>(b.r, base, n, full, rr, p.r);
            if (zoo_mixed) load4<int32_t>(b.model_idx, base, n, full, kind, FISHING_KIND_BEVERTON_HOLT);
            if (b.sigma) load4<T>(b.sigma, base, n, full, sg, p.sigma);
            if (b.ep_return) load4<T>(b.ep_return, base, n, full, er, (T)0);
        }
        bool kind_dirty = false;
        const uint64_t quad = (env_offset + (uint64_t)base) >> 2;
        const T robs_scalar = reset_obs<T, MODEL>(p.x0, p.K);   // loop-invariant unless per-env K
        bool kr_dirty = false;

        for (int32_t s = 0; s < Tsteps; ++s) {
            const uint64_t step_counter = step_counter0 + (uint64_t)s;
            T z[4] = {(T)0, (T)0, (T)0, (T)0};
            uint32_t aw[4] = {0u, 0u, 0u, 0u};
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
                    if (MODEL == FISHING_MODEL_V0) a_d = (int32_t)policy_param;
                    else a_c = (T)(float)policy_param;
                } else {
                    T q;
                    if (policy == FISHING_POLICY_ESCAPEMENT) {   // policies.py:27-31
                        const T x = (obs[j] + (T)1) * KK[j];
                        const T dq = x - policy_param;
                        q = ((T)0 > dq) ? (T)0 : dq;             // max(x - S, 0.0)
                    } else {                                     // MSY, policies.py:16-19
                        q = policy_param;
                    }
                    if (MODEL == FISHING_MODEL_V0) a_d = action_int_from_quota<T>(q, p.n_actions, KK[j]);
                    else a_c = (T)action_cts_from_quota<T>(q, KK[j], dk);
                }
                quota[j] = (MODEL == FISHING_MODEL_V0) ? quota_int<T>(a_d, p.n_actions, KK[j]) : quota_cts<T>(a_c, KK[j]);
                act_rec[j] = (MODEL == FISHING_MODEL_V0) ? (T)a_d : a_c;
            }
            T o2[4], r2[4];
            bool d2[4];
            int32_t t2[4];
            bool stepped = false;
            if constexpr (zoo_mixed) {
                if (!b.sigma) {     // wave-uniform.  fishing-v11: regroup the wave's envs by growth function
                    __shared__ T win[4 * 512];
                    T xh[4], hv[4], xn[4];
                    int kk[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const T x = (obs[j] + (T)1) * KK[j];
                        hv[j] = (quota[j] < x) ? quota[j] : x;
                        const T d = x - hv[j];
                        xh[j] = ((T)0 > d) ? (T)0 : d;
                        xn[j] = (T)0;
                        kk[j] = (kind[j] >= 0 && kind[j] < FISHING_N_KINDS) ? kind[j] : FISHING_KIND_BEVERTON_HOLT;
                    }
                    zoo_draw_regrouped<T>(kk, xh, z, p.zoo, xn, win + (threadIdx.x >> 6) * 512, lane);
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
                if (AUTO && kPerEnv) {
                    if (redraw_tile<T, MODEL>(seed, env_offset + (uint64_t)base, step_counter, kStreamAutoReset,
                                              p.K_mean, p.r_mean, p.sigma_p, p.x0, fin, KK, rr, obs, t))
                        kr_dirty = true;
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
            if (kPerEnv && kr_dirty) {
                store4<T>(b.K, base, n, full, KK);
                store4<T>(b.r, base, n, full, rr);
            }
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

int check_common(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b);
void launch_shape(const FishingParams* p, int64_t n, int& blocks, int& threads);

template <typename T, int MODEL>
int launch_rollout_policy(int policy, const ParamsT<T>& pt, const BuffersT<T>& bt, int64_t n, uint64_t env_offset,
                          T policy_param, int32_t Tsteps, T* traj, uint64_t seed, uint64_t step_counter,
                          int noise_on, int blocks, int threads, hipStream_t s) {
    const DivK dk = make_divk((double)pt.K);
#define FISHING_LAUNCH_ROLLOUT(POL)                                                                                \
    do {                                                                                                           \
        if (pt.flags & FISHING_FLAG_AUTO_RESET)                                                                    \
            rollout_kernel<T, MODEL, POL, true><<<blocks, threads, 0, s>>>(pt, bt, n, env_offset, policy_param,  \
                                                                           Tsteps, traj, seed, step_counter,      \
                                                                           noise_on, policy, dk);                 \
        else                                                                                                       \
            rollout_kernel<T, MODEL, POL, false><<<blocks, threads, 0, s>>>(pt, bt, n, env_offset, policy_param, \
                                                                            Tsteps, traj, seed, step_counter,     \
                                                                            noise_on, policy, dk);                \
    } while (0)
    if constexpr (is_zoo_tag(MODEL)) {
        FISHING_LAUNCH_ROLLOUT(-1);          // run-time policy switch
    } else {
        switch (policy) {
            case FISHING_POLICY_RANDOM: FISHING_LAUNCH_ROLLOUT(FISHING_POLICY_RANDOM); break;
            case FISHING_POLICY_CONSTANT: FISHING_LAUNCH_ROLLOUT(FISHING_POLICY_CONSTANT); break;
            case FISHING_POLICY_ESCAPEMENT: FISHING_LAUNCH_ROLLOUT(FISHING_POLICY_ESCAPEMENT); break;
            default: FISHING_LAUNCH_ROLLOUT(FISHING_POLICY_MSY); break;
        }
    }
#undef FISHING_LAUNCH_ROLLOUT
    return (int)hipGetLastError();
}

template <typename T>
int rollout_impl(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b, int32_t policy,
                 double policy_param, int32_t Tsteps, void* traj, uint64_t seed, uint64_t step_counter,
                 fishing_stream_t stream) {
    const int rc = check_common(p, n, env_offset, b);
    if (rc != FISHING_OK) return rc;
    if (policy < FISHING_POLICY_RANDOM || policy > FISHING_POLICY_MSY) return FISHING_ERR_POLICY;
    if (Tsteps < 0) return FISHING_ERR_SIZE;
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
                                                              step_counter, noise_on, blocks, threads, s);
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

}  // extern "C"
