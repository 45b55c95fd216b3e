// fishing_step.hip -- the vectorised step() kernels and their dispatch (gfx950).
//
// Data layout in HBM: structure of arrays, one contiguous stream per field
// (obs, action, reward, done, t [, r, K, sigma, z_ext, terminal_obs, ep_return]); env i of
// the shard sits at element i of every stream.  A thread owns 4 consecutive envs, so each
// f32 / i32 stream is read or written with one 16-byte access per lane (1 KiB per wave
// instruction) and the done bytes with one dword per lane.  No env ever reads another env's
// state: there is no LDS staging of the streams (nothing is reused) -- LDS only carries the
// per-workgroup reduction of the episodic-return record and fishing-v11's coefficient table.
//
// Roofline: HBM.  Algorithmic bytes per env-step: f32 layout 25 B
// (R obs 4 + action 4 + t 4; W obs 4 + reward 4 + done 1 + t 4); fishing-v4 +4 B (sigma array) with
// derived parameters, +12 B with stored r / K arrays; f64 parity layout 37 B; +4/8 B per optional stream.
//
// Three kernels:
//   step_kernel_lean<T, MODEL, F, E>  whole 1024-env tiles, one tile per workgroup, one body; the optional streams are
//                                     selected by the feature mask F (namespace feat).  Only the masks a request can
//                                     reach are instantiated (lean_dispatch below).
//   step_kernel<T, MODEL>             the general kernel: ragged tails, batches below one tile, explicit launch
//                                     shapes.  Everything optional is a run-time decision.
//   step_floor_kernel<MODE>           diagnostic: the empty / copy floor of a lean launch (bench.py).
//
// FISHING_ZOO_F64_FAR 0: the float64 zoo's growth functions run on the algebraic form (fishing_common.h: zoo_draw_f64);
// the hand-over to the reference's own log / exp round trip for far stocks / far results is compiled out of THIS
// translation unit, because a step's outputs cannot carry the difference: the state leaves a step as obs = x' / K - 1,
// whose spacing near -1 is 1.1e-16 -- a population below 2^-53 K is obs = -1 in the reference and here alike, and below
// 2^-30 K the two evaluations differ by < 1e-13 of a result that obs resolves to 1e-7 of itself; reward is the harvest,
// done tests x' <= 0, zeros are zeros in both forms; obs beyond 1e9 lies nine orders of magnitude outside anything the
// dynamics reach.  population_draw (fishing_aux.hip), which hands x' out itself, keeps the hand-over.  The cold branch
// cost every kernel 16-32 VGPRs (fishing-v11 float64 41.9 -> 39.8 us at N = 2^22: profiles/r05_zoo_f64_far_path.jsonl);
// tests/test_gpu_zoo.py::test_zoo_f64_step_outputs_do_not_depend_on_how_a_far_stock_is_evaluated holds the argument.
#define FISHING_ZOO_F64_FAR 0
#include "fishing_common.h"
#include "fishing_host.h"

#include <cstdio>
#include <cstdlib>
#include <algorithm>

namespace fishing {

// Byte thresholds of the launch's walk (bytes ONE step streams); the three knobs a deployment on another cache hierarchy may
// want to move.
#ifndef FISHING_XZZ_MIN_BYTES
#define FISHING_XZZ_MIN_BYTES (100ll << 20)         // from here every lean kernel walks its tiles zig-zag
#endif
#ifndef FISHING_NTA_MIN_BYTES
#define FISHING_NTA_MIN_BYTES (200ll << 20)         // ... and loads the caller's actions nontemporal
#endif
#ifndef FISHING_F64_E2_MAX_BYTES
#define FISHING_F64_E2_MAX_BYTES (512ll << 20)      // float64: two envs per thread below this many bytes per step
#endif
constexpr int kStepMaxThreads = 256;                // workgroup size cap of the general kernel
constexpr int kTileEnvsLean = 1024;                 // envs per tile (= per workgroup) of the lean kernels

// An env that was already finished BEFORE this step (stepped on without a reset: years_passed beyond Tmax, or
// no fish left) must not enter the episodic-return record a second time.
template <typename T>
__device__ __forceinline__ bool was_done(T obs, int32_t t, T K, int32_t Tmax) {
    return (t > Tmax) || ((obs + (T)1) * K <= (T)0);
}

// ---------------------------------------------------------------- general kernel
template <typename T, int MODEL>
__global__ void __launch_bounds__(kStepMaxThreads)
step_kernel(const ParamsT<T> p, const BuffersT<T> b, const int64_t n, const uint64_t env_offset,
            const uint64_t seed_arg, const uint64_t step_counter_arg, const int noise) {
    // graph-replay safety: with a device-resident counter the launch arguments can stay frozen
    // in a captured hipGraph while the noise key still advances (wave-uniform scalar load)
    const uint64_t step_counter = b.counter ? (*b.counter + step_counter_arg) : step_counter_arg;
    constexpr bool kPerEnv = (MODEL == FISHING_MODEL_V4);
    constexpr bool kZoo = is_zoo_tag(MODEL);
    constexpr bool zoo_mixed = (MODEL == kModelZooMixed);              // growth kind per env
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t tile_envs = (int64_t)blockDim.x * kEnvsPerThread;
    const int64_t ntiles = (n + tile_envs - 1) / tile_envs;
    const bool auto_reset = (p.flags & FISHING_FLAG_AUTO_RESET) != 0;
    const bool derived = kPerEnv && (p.flags & FISHING_FLAG_V4_DERIVED) != 0;
    uint64_t origin_step = p.origin_step, origin_counter = p.origin_counter;
    if (derived) device_origin(b.counter, origin_step, origin_counter);
    double acc[kPartialFields] = {0.0, 0.0, 0.0, 0.0};
    // zoo (fishing-v5..v11): wave-uniform facts about the family
    // (fishing-v10 is Beverton-Holt: only the run-time-kind tag of this kernel can see a drifting r -- not fishing-v11's)
    const bool zoo_drift = kZoo && MODEL != kModelZooMixed && p.model == FISHING_MODEL_V10;       // r += alpha every draw
    const int zoo_kind = kZoo ? p.kind : FISHING_KIND_BEVERTON_HOLT;
    const GrowthT<T> zoo_base = p.growth;

    // (fishing-v11: the growth functions' coefficients as a table in LDS, fishing_common.h: zoo_lut_fill)
    __shared__ alignas(16) T zoo_lut[zoo_mixed ? kZooLutSize : 4];
    if constexpr (zoo_mixed) {
        if (threadIdx.x < kWave) zoo_lut_fill<T>(zoo_lut, p.zoo);
        __syncthreads();
    }

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = (tile * blockDim.x + threadIdx.x) * kEnvsPerThread;
        const bool active = base < n;
        const bool full = base + kEnvsPerThread <= n;
        // (keeps the Philox key schedule next to its rounds instead of in 20-30 long-lived SGPRs: this kernel, with
        // every argument a run-time decision, is the one shortest of them)
        uint64_t seed_it = seed_arg;
        asm volatile("" : "+s"(seed_it));
        const uint64_t seed = seed_it;

        T obs[4], rr[4], KK[4], sg[4], z[4];
        int32_t t[4];
        float a_f[4];
        int32_t a_i[4];
        int32_t kind[4];
        int32_t st[4] = {0, 0, 0, 0};       // fishing-v4 derived: per-env origin stamps (FishingBuffers.v4_stamp)
        const bool stamped = derived && b.stamp != nullptr;
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
            if (kPerEnv && !derived) {
                load4<T>(b.r, base, n, full, rr, p.r);
                load4<T>(b.K, base, n, full, KK, p.K);
            }
            if (zoo_drift) load4<T>(b.r, base, n, full, rr, p.r);
            if (zoo_mixed) load4<int32_t>(b.model_idx, base, n, full, kind, FISHING_KIND_BEVERTON_HOLT);
            if (b.sigma) load4<T>(b.sigma, base, n, full, sg, p.sigma);
            if (noise == kNoiseExt) load4<T>(b.z_ext, base, n, full, z, (T)0);
            if (stamped) load4<int32_t>(b.stamp, base, n, full, st, 0);
        }
        if (noise == kNoisePhilox) {
            float zq[4];
            noise_quad(seed, (env_offset + (uint64_t)base) >> 2, step_counter, zq);
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = (T)zq[j];
        }
        if (derived) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                derive_model_error<T>(seed, env_offset + (uint64_t)base + j, step_counter, t[j], origin_step,
                                      origin_counter, p.K_mean, p.r_mean, p.sigma_p, KK[j], rr[j], st[j]);
        }

        T obs_next[4], rew[4];
        int32_t t_next[4];
        bool dn[4];
        bool stepped = false;
        if constexpr (zoo_mixed) {
            if (!b.sigma) {     // wave-uniform.  fishing-v11: every env's coefficients from the LDS table
                T xh[4], hv[4], xn[4];
                int kk[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const T quota = quota_cts<T>((T)a_f[j], KK[j]);
                    const T x = (obs[j] + (T)1) * KK[j];
                    hv[j] = (quota < x) ? quota : x;
                    const T d = x - hv[j];
                    xh[j] = d;       // (max(d, 0.0) is the identity here: stock_after_harvest)
                    xn[j] = (T)0;
                    kk[j] = (kind[j] >= 0 && kind[j] < FISHING_N_KINDS) ? kind[j] : FISHING_KIND_BEVERTON_HOLT;
                }
                zoo_draw_lut_tile<T, 4>(kk, xh, z, p.zoo, zoo_lut, xn);
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
                        env_step_zoo<T, -1, true>(obs[j], t[j], quota, z[j], kind[j], P, KK[j], p.Tmax, obs_next[j], rew[j],
                                                  dn[j], t_next[j]);
                    else
                        env_step_zoo<T, -1, false>(obs[j], t[j], quota, z[j], kind[j], P, KK[j], p.Tmax, obs_next[j], rew[j],
                                                   dn[j], t_next[j]);
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
            if (b.reward) store4<T>(b.reward, base, n, full, rew);
            if (b.terminal_obs) store4<T>(b.terminal_obs, base, n, full, obs_next);
            if (b.done) {
                if (full) {
                    const uint32_t packed = (uint32_t)dn[0] | ((uint32_t)dn[1] << 8) |
                                            ((uint32_t)dn[2] << 16) | ((uint32_t)dn[3] << 24);
                    *reinterpret_cast<uint32_t*>(b.done + base) = packed;
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
                bool fresh[4];      // the episode ended on THIS step
#pragma unroll
                for (int j = 0; j < 4; ++j) fresh[j] = dn[j] && (auto_reset || !was_done<T>(obs[j], t[j], KK[j], p.Tmax));
                record_tile<T>(fresh, er, t_next, acc);
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
            if (kPerEnv && !derived) {
                redrawn = redraw_tile<T, MODEL>(seed, env_offset + (uint64_t)base, step_counter, kStreamAutoReset,
                                                p.K_mean, p.r_mean, p.sigma_p, p.x0, dn, KK, rr, obs_next, t_next);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (dn[j]) {
                        obs_next[j] = reset_obs<T, MODEL>(p.x0, KK[j]);    // fishing-v4: x0, whatever the new K
                        t_next[j] = 0;
                        st[j] = 0;          // ... and from here on the year counter dates the episode again
                    }
                }
                if (stamped && active) store4<int32_t>(b.stamp, base, n, full, st);
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

    if (b.partials) add_block_partials<kStepMaxThreads / kWave>(acc, b.partials);
}

// ---------------------------------------------------------------- lean fast path
// The same step as step_kernel for whole 1024-env tiles, with everything the general kernel decides per thread
// decided per kernel: unconditional 16-byte accesses (the ragged tail goes to a second, general launch),
// select-based auto-reset, and a compact argument block (72-97 SGPRs instead of 106 -> 8 waves per SIMD).
// Measured at N = 2^22 against a pure copy with the same stream shape (scripts/exp/): copy 16.0 us, this
// 16.2 us, general kernel 17.7 us.  Same results bit for bit (tests/test_gpu_parity.py::
// test_lean_and_general_kernels_agree).
//
// ONE body; the feature mask F says which optional streams exist.  Without feat::OPT a set bit means "this
// stream IS there" and a clear bit "is NOT": the exact instantiations of the hot requests carry no run-time
// test at all.  With feat::OPT a set bit means "MAY be there" -- the stream's pointer (or run-time flag)
// decides, wave-uniformly -- so one catch-all instantiation per (T, MODEL) serves every other combination.
namespace feat {
constexpr int kNoiseMask = 3;      // bits 0-1: kNoiseNone / kNoiseExt / kNoisePhilox, or
constexpr int kNoiseRT = 3;        //           3 = decided at run time (LeanArgs::noise_rt)
constexpr int RET = 1 << 2;        // ep_return accumulator + episodic-return record
constexpr int SIGARR = 1 << 3;     // per-env noise scale
constexpr int T8 = 1 << 4;         // compact layout: one-byte year counters
constexpr int TERM = 1 << 5;       // terminal_obs: the observation before the fused auto-reset (SB3)
constexpr int BITS = 1 << 6;       // done_bits: wave-ballot termination mask
// (1 << 7: unused -- the zig-zag walk is a run-time property of the launch, LeanArgs::zz_rt)
constexpr int DERIVED = 1 << 8;    // fishing-v4: (K, r) re-derived from the Philox streams, no r / K arrays
constexpr int DRIFT = 1 << 9;      // fishing-v10: per-env r, drifting by alpha every draw
constexpr int OPT = 1 << 10;
constexpr int LATCH = 1 << 11;     // RET without auto-reset: a finished env that is stepped on must not enter the record
                                   // again.  Catch-all only -- the exact RET instantiations are the auto-reset ones.
constexpr int ONE = 1 << 13;       // one tile per workgroup (grid == ntiles; every lean launch): no tile loop, and with RET the
                                   // return record -- workgroup reduction + its atomic -- is issued BEFORE the tile's stores, so the
                                   // atomic's round trip runs under theirs.  Exact instantiations and catch-alls alike.  Per step at N = 2^19 / 2^20 /
                                   // 2^21, back to back: 4.75 -> 4.09, 6.15 -> 5.71, 9.78 -> 8.15 us (profiles/r03_small_n/).
constexpr int STAMP = 1 << 14;     // fishing-v4 derived: per-env origin stamps (FishingBuffers.v4_stamp: R 4 + W 4).  Catch-all only --
                                   // masked resets are the rare path; the exact DERIVED instantiations stay stamp-free.
constexpr int KP2 = 1 << 12;       // the scalar K is a power of two (K = 1 included): x / K is the exact multiply x * (1 / K),
                                   // same bits, a third of the instructions.  Exact instantiations of fishing-v0/v1/v2 only;
                                   // any other K takes the catch-all's correctly rounded division.
}  // namespace feat

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
    const T* sigma_arr;      // SIGARR
    T* terminal_obs;         // TERM
    uint64_t* done_bits;     // BITS: bit i % 64 of word i / 64 = done[i]
    const T* z_ext;          // kNoiseExt
    int32_t* stamp;          // STAMP
    T pr, pK, sigma, C, x0, r_mean, K_mean, sigma_p;
    int32_t Tmax, n_actions;
    uint32_t auto_reset;
    int32_t noise_rt;        // feat::kNoiseRT: the noise mode of this launch
    uint32_t t8_rt, derived_rt, drift_rt;    // feat::OPT: run-time values of T8 / DERIVED / DRIFT
    uint32_t zz_rt;          // zig-zag tile walk for this launch
    uint32_t nta_rt;         // zig-zag forms: nontemporal action loads (a step streams >= FISHING_NTA_MIN_BYTES)
    int64_t n_live;          // FISHING_FLAG_PADDED_TILES: the number of envs that exist (a multiple of 4); INT64_MAX otherwise
    uint64_t origin_step, origin_counter;    // DERIVED (derive_model_error)
    GrowthT<T> growth;       // fishing-v5..v10: the growth function's parameter set (unused, hence never
                             // loaded, by the v0/v1/v2/v4 instantiations)
    T alpha;                 // DRIFT
    DivK dk;                 // KP2: the scalar K's exact reciprocal
    T robs;                  // the reset observation x0 / K - 1 of the scalar-K models, divided on the host (IEEE, same bits):
                             // wave-uniform, but the device has no scalar float division -- 13 VALU instructions per
                             // thread, ahead of the tile's stores
};

// fishing-v11 only (MODEL == kModelZooMixed): the growth kind in force per env, the model list it is redrawn from and
// the per-kind parameters.  Every other model passes the empty struct (one byte of kernel arguments).
template <typename T>
struct LeanMixedArgs {
    int32_t* model_idx;
    int32_t n_models;
    int32_t kinds[FISHING_N_KINDS];
    GrowthT<T> zoo[FISHING_N_KINDS];
    T lut[kZooLutSize];      // zoo_lut_rows(zoo): the coefficient table, converted on the host
};
struct LeanNoExtra {};
template <typename T, int MODEL>
using LeanExtra = std::conditional_t<MODEL == kModelZooMixed, LeanMixedArgs<T>, LeanNoExtra>;

// E = envs per thread: 4 everywhere (16-byte accesses on the 4-byte streams) except the float64 parity layout at
// cache-resident sizes, which runs E = 2 -- 16 bytes per lane on ITS streams instead of 32 (two 16-byte accesses, half
// of each 64-byte line per instruction): a copy over the same streams takes 23.5 instead of 27.2 us at N = 2^22
// (profiles/r02_f64_access_shape.jsonl).  A workgroup is 1024 / E threads on one 1024-env tile; the lane pair that shares
// an env quad computes the quad's Philox block twice and keeps one Box-Muller pair each.
// (fishing-v11's float32 kernel with returns needs 65 VGPRs, one above what eight waves per SIMD allow: compiled for eight --
// amdgpu_waves_per_eu(8, 8) -- it spills 24 bytes per lane; left at seven.)
// Shapes and orders that were measured and not adopted (128- / 512-thread workgroups, two tiles per workgroup, two envs per thread
// for fishing-v4, load orders, staggered starts, occupancy caps): profiles/NOTES_r01_r05.md.
// n_live_p, the fifth leading argument:  bits 0-39 the envs that exist (all ones: no bound), bits 40-57 how many tiles
// THIS launch walks backwards (0, or ntiles & ~7 on a zig-zag launch's odd steps when the host holds the step counter),
// bit 58: a zig-zag launch whose step counter lives in device memory (the kernel finds the parity itself), bit 59:
// nontemporal action loads.
constexpr int64_t kLiveMask = (1ll << 40) - 1;
constexpr int kWalkShift = 40;
constexpr int64_t kWalkMask = (1ll << 18) - 1;
constexpr int64_t kWalkDeviceParityBit = 1ll << 58;
constexpr int64_t kWalkNtaBit = 1ll << 59;
template <typename T, int MODEL, int F, int E = 4>
__global__ void __launch_bounds__(kTileEnvsLean / E)
step_kernel_lean(T* const obs_p, const void* const action_p, int32_t* const t_p, T* const ep_return_p, const int64_t n_live_p,
                 const LeanArgs<T> a, const LeanExtra<T, MODEL> ex, const int64_t ntiles, const uint64_t env_offset,
                 const uint64_t seed, const uint64_t step_counter_arg) {
    // The four streams every tile reads first, the padded-tile bound the action address needs and the launch's walk (n_live_p:
    // see kLiveMask) are LEADING SCALAR arguments: this translation unit is built with -mllvm
    // -amdgpu-kernarg-preload-count=10 (build.py), which has the command processor place the first ten kernarg dwords in
    // SGPRs at wave launch -- a wave finds its tile and its four addresses without a single s_load (a by-value struct is
    // not preloaded), and fetches the rest of its arguments in one batch.  Per step, back to back, against the same kernel
    // without preload: N = 2^20 6.04 -> 5.75 us, 2^22 21.41 -> 20.87, 2^19 unchanged
    // (profiles/r03_small_n/s15_kernarg_preload_product.jsonl; harness sweep of 4 .. 14 dwords: s14_*).  Firmware without
    // the feature runs the kernel's own s_load prologue.  The launch's walk rides in the same preloaded word (n_live_p): as
    // run-time fields of the struct its flags string FOUR dependent s_load round trips through every wave (the walk's flag,
    // the step's parity, the batch, the counter); in the word there are two, one of them hidden behind the tile's loads:
    // N = 2^19 4.67 -> 4.34 us, 2^20 6.0 -> 5.8, 2^22 18.94 -> 18.76 (profiles/r05_walk_word.jsonl).
    constexpr bool kPerEnv = (MODEL == FISHING_MODEL_V4);
    constexpr bool kMixed = (MODEL == kModelZooMixed);    // fishing-v11: growth function per env
    constexpr bool kZoo = is_zoo_tag(MODEL) && !kMixed;   // one growth function of the zoo, compile-time kind
    constexpr int kZooKind = kZoo ? (MODEL - kModelZoo) : -1;
    constexpr bool kOpt = (F & feat::OPT) != 0;
    constexpr bool kExact = !kOpt;
    static_assert(MODEL != kModelZooRT, "the run-time-kind tag belongs to the general kernel");
    static_assert(!(F & feat::DRIFT) || MODEL == kModelZoo + FISHING_KIND_BEVERTON_HOLT, "DRIFT is fishing-v10");
    static_assert(!(F & feat::DERIVED) || kPerEnv, "DERIVED is fishing-v4");
    static_assert(!(F & feat::LATCH) || kOpt, "LATCH lives in the catch-alls");
    static_assert(!(F & feat::STAMP) || (kOpt && kPerEnv && (F & feat::DERIVED)), "STAMP: fishing-v4's derived catch-all");
    static_assert(!(F & feat::KP2) || (kExact && !kPerEnv && !kZoo && !kMixed), "KP2: exact fishing-v0/v1/v2 instantiations");
    static_assert((F & feat::ONE) != 0, "every lean form is a one-tile form: a workgroup of 1024 / E threads per 1024-env tile");
    static_assert(E == 4 || (E == 2 && sizeof(T) == 8), "E = 2: the float64 layout");
    constexpr int kThreads = kTileEnvsLean / E;
    constexpr int kTileEnvs = kTileEnvsLean;
    // fishing-v4's derived-parameter exact kernels issue the year counters' load first (see the tile's loads)
    constexpr bool kTFirst = kExact && kPerEnv && (F & feat::DERIVED) != 0;
    // Without OPT these fold to compile-time constants; with OPT they are wave-uniform scalars.
    const bool RET = (F & feat::RET) && (kExact || ep_return_p != nullptr);
    const bool SIGARR = (F & feat::SIGARR) && (kExact || a.sigma_arr != nullptr);
    const bool T8 = (F & feat::T8) && (kExact || a.t8_rt != 0);
    const bool TERM = (F & feat::TERM) && (kExact || a.terminal_obs != nullptr);
    const bool BITS = (F & feat::BITS) && (kExact || a.done_bits != nullptr);
    const bool DERIVED = (F & feat::DERIVED) && (kExact || a.derived_rt != 0);
    const bool STAMP = (F & feat::STAMP) && DERIVED && a.stamp != nullptr;
    const bool DRIFT = (F & feat::DRIFT) && (kExact || a.drift_rt != 0);
    // The walk is a run-time property of the launch, decided by the HOST, in the preloaded n_live argument: as fields of the by-value struct, the walk's flags put one
    // s_load round trip in front of every wave's first global load -- what the kernarg preload had taken away.
    const int64_t n_live = n_live_p & kLiveMask;
    const bool ZZ = (n_live_p & kWalkDeviceParityBit) != 0;     // (the walks that need the step counter BEFORE the tile's loads)
    int64_t walk_back = (n_live_p >> kWalkShift) & kWalkMask;
    // The caller's action stream is read once per step and never again: from ~200 MB per step (N >= 2^23) the zig-zag forms
    // load it nontemporal, so that it does not evict the state lines the reversed walk is about to re-hit.  N = 2^26:
    // 283 -> 262 us bare, 398 -> 380 with returns; 2^25: 128.5 -> 124, 181.5 -> 172.6; 2^23 bare 32.1 -> 31.0.  Not at 2^22:
    // the action ring itself is cache-resident there, the hint costs 6-11 % (profiles/r03_nt_action_loads.jsonl,
    // r03_xcd_zigzag.jsonl).  (Streaming the state STORES of all but the walk's last 128-224 MB as well: < 1 %.)
    const bool NTA = (n_live_p & kWalkNtaBit) != 0;
    const int noise = ((F & feat::kNoiseMask) == feat::kNoiseRT) ? a.noise_rt : (F & feat::kNoiseMask);
    // Pull the kernel arguments into SGPRs in ONE batch of scalar loads.  Left alone, the compiler
    // loads arguments next to their first use, which strings five dependent s_load / s_waitcnt round
    // trips in front of the first global load of every wave.  Measured at N = 2^22: bare step 16.9 -> 16.5 us; with the return accumulator 21.67 -> 21.45 us, as long as the
    // ep_return / partials pointers stay out of the batch (21.50 with them): profiles/r01f_lean_fence_ab.txt.
    auto batch_args = [&]() {
    if constexpr (kExact) {     // (the catch-alls are short of SGPRs as it is)
        asm volatile("" ::"s"(obs_p), "s"(action_p), "s"(a.reward), "s"(a.done), "s"(t_p), "s"(a.counter), "s"(a.pr),
                     "s"(a.pK), "s"(a.sigma), "s"(a.C), "s"(a.x0), "s"(a.Tmax), "s"(a.n_actions), "s"(a.auto_reset),
                     "s"(ntiles), "s"(env_offset), "s"(seed), "s"(step_counter_arg));
        if constexpr (kPerEnv && !(F & feat::DERIVED)) asm volatile("" ::"s"(a.r), "s"(a.K));
        if constexpr (kPerEnv) asm volatile("" ::"s"(a.r_mean), "s"(a.K_mean), "s"(a.sigma_p));
        if constexpr ((F & feat::DERIVED) && kExact) asm volatile("" ::"s"(a.origin_step), "s"(a.origin_counter));
        if constexpr ((F & feat::SIGARR) != 0) asm volatile("" ::"s"(a.sigma_arr));
        if constexpr ((F & feat::TERM) && kExact) asm volatile("" ::"s"(a.terminal_obs));
        if constexpr ((F & feat::BITS) && kExact) asm volatile("" ::"s"(a.done_bits));
        if constexpr (kZoo) {
            asm volatile("" ::"s"(a.growth.r), "s"(a.growth.K), "s"(a.growth.sigma), "s"(a.growth.C), "s"(a.growth.M),
                         "s"(a.growth.theta), "s"(a.growth.q), "s"(a.growth.b), "s"(a.growth.a), "s"(a.growth.bq));
            asm volatile("" ::"s"(a.growth.A), "s"(a.growth.invK), "s"(a.growth.invM), "s"(a.growth.invB), "s"(a.growth.gc));
        }
        if constexpr ((F & feat::DRIFT) != 0) asm volatile("" ::"s"(a.r), "s"(a.alpha));
    }
    };
    // (called behind the tile's loads, which need only the preloaded arguments)
    // graph-replay mode keeps the step counter in device memory (wave-uniform: one scalar load), read AFTER the tile's
    // loads are issued, which do not depend on it
    auto read_counter = [&]() -> uint64_t {
        if (!a.counter) return step_counter_arg;
        uint64_t c;
        asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c) : "s"(a.counter) : "memory");
        return c + step_counter_arg;
    };
    // (... unless the launch walks zig-zag: the tile index needs the step's parity)
    uint64_t step_counter = 0;
    if (ZZ) {       // (graph replay at the zig-zag sizes)
        asm volatile("");       // a real branch: flattened, its tests wait for the struct's s_loads in every launch
        step_counter = read_counter();
        constexpr int64_t G = 8;
        walk_back = (step_counter & 1) ? (ntiles & ~(G - 1)) : 0;
    }
    uint64_t origin_step = a.origin_step, origin_counter = a.origin_counter;
    // an exact RET instantiation is only ever launched with auto-reset on (the dispatch sends RET without it to the
    // catch-all, which carries the LATCH): the flag is a compile-time fact there
    const bool auto_reset = (kExact && (F & feat::RET)) ? true : a.auto_reset != 0;
    const bool LATCH = (F & feat::LATCH) && !auto_reset;
    double acc[kPartialFields] = {0.0, 0.0, 0.0, 0.0};
    const T robs_scalar = (MODEL == FISHING_MODEL_V4) ? a.x0 : a.robs;       // (reset_obs<T, MODEL>(a.x0, a.pK))

    __shared__ alignas(16) T zoo_lut[kMixed ? kZooLutSize : 4];     // (fishing-v11 only; unused -- and not allocated -- elsewhere)
    auto do_tile = [&](const int64_t it) {
        // ZZ: odd steps walk the tiles backwards, so what the previous step touched LAST is what this one reads FIRST -- while
        // it is still cached.  Backwards IN GROUPS OF EIGHT: workgroups are dealt round-robin over the 8 XCDs, each with its
        // own 4 MiB L2 that keeps its lines across launches; tile % 8 == workgroup % 8 in both directions keeps every tile on
        // the XCD whose L2 may still hold it (a plain reversal moves every tile to another XCD each step: the one-tile form
        // at N = 2^20 then takes 8.8 instead of 4.1 us).  Two caches profit: the L2s from ~100 MB per step on, where an
        // XCD's share no longer fits its L2 (N = 2^22: 21.2 -> 18.9 us with returns, 16.0 -> 14.1 bare in the harness:
        // profiles/r03_xcd_zigzag.jsonl) -- below that everything is L2-resident and the forward walk is as good or
        // better --, and the 256 MiB Infinity Cache at the HBM-resident sizes (N = 2^26 with returns, a workgroup per tile:
        // 406 us forward, 347 zig-zag: profiles/r03_zz_nta_one_tile.jsonl).
        int64_t tile = it;
        // fishing-v11's coefficient table comes ready-made in the launch's arguments (LeanMixedArgs::lut): its one vector load goes
        // out FIRST, so that the wave's wait for it does not wait for the tile's loads as well (loads return in order)
        T lut_word = (T)0;
        if constexpr (kMixed) {
            if (threadIdx.x < kZooLutSize) lut_word = ex.lut[threadIdx.x];
        }
        {
            constexpr int64_t G = 8;
            const int64_t whole = walk_back;          // (ntiles & ~7 on a backwards step: a last partial group keeps its place)
            if (it < whole) tile = (whole - G - (it & ~(G - 1))) + (it & (G - 1));
        }
        const int64_t base = (tile * kThreads + threadIdx.x) * E;
        // FISHING_FLAG_PADDED_TILES: the state buffers have room for whole tiles, so a batch that is not a multiple of 1024
        // envs still runs in this ONE launch (no second, one-workgroup launch for the tail: 3.7-4.2 us per step).  The
        // envs behind the last one are scratch: stepped like any other, but they never finish (neither recorded nor
        // redrawn), and the streams the CALLER owns -- actions, external noise -- are read at the last quad that exists
        // instead of past their end.  `live` is true everywhere otherwise.
        // (the tile's share of n_live is a scalar; per lane one 32-bit compare and one 32-bit select)
        const int64_t tile_left = n_live - tile * kTileEnvs;                     // wave-uniform
        const uint32_t left32 = tile_left >= kTileEnvs ? (uint32_t)kTileEnvs : (tile_left > 0 ? (uint32_t)tile_left : 0u);
        const uint32_t lane_env = threadIdx.x * (uint32_t)E;
        const bool live = lane_env < left32;
        // (a padded tile holds at least one live quad, so the last live thread's envs are tile-relative -- one 32-bit select)
        const int64_t cbase = tile * kTileEnvs + (int64_t)(live ? lane_env : left32 - (uint32_t)E);
        // The Philox round keys (seed + i * Weyl) are wave-uniform; hoisted out of this loop they sit in 20-30 SGPRs
        // for the whole kernel, which pushes the fishing-v4 variants (two generators) past 100 SGPRs = 7 instead of
        // 8 waves per SIMD.  Laundering the seed per tile keeps the key schedule next to its rounds (a scalar add
        // each): fishing-v4 derived 10.7 -> 10.2 us at N = 2^21, 78.2 -> 74.6 us at 2^24, stored arrays 16.5 -> 16.1;
        // the single-generator kernels prefer the hoisted keys (fishing-v1 bare 16.16 vs 16.5 us), so only
        // fishing-v4 launders (profiles/r02_ab_variants.jsonl).
        uint64_t seed_it = seed;
        // (... and the float32 catch-alls, which sit at the 106-SGPR ceiling: 26 -> 2 SGPR-to-VGPR-lane spills)
        if constexpr (kPerEnv || (kOpt && sizeof(T) == 4)) asm volatile("" : "+s"(seed_it));
        T obs[E], rr[E], KK[E], z[E], er[E], sg[E];
        int32_t t[E], a_i[E];
        int32_t kind[E], st[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            kind[j] = FISHING_KIND_BEVERTON_HOLT;
            st[j] = 0;
        }
        float a_f[E];
        {
            // the year counters: fishing-v4's derivation of (K, r) is the one piece of arithmetic that needs loaded data before it can
            // start (the reset origin of each env), and loads return in the order they were issued -- in the derived-parameter exact
            // kernels this load goes out FIRST (not third, behind the sigma array's and the observations'):
            // config 5's shard 12.3 -> 11.85 us, N = 2^22 22.4 -> 22.1 (profiles/r05_v4_t_first.jsonl).  (LLVM's wait in front of the
            // derivation covers the observations' load too, vmcnt(3) of five -- they return right behind the counters; an explicit
            // s_waitcnt vmcnt(4) in front of it changes nothing: the compiler's own follows.)
            auto load_t = [&]() {
                VecE<int32_t, E> qt;
                if (T8) {
                    typedef std::conditional_t<E == 4, uint32_t, uint16_t> bytesE;
                    const uint32_t w = *reinterpret_cast<const bytesE*>(reinterpret_cast<const uint8_t*>(t_p) + base);
#pragma unroll
                    for (int j = 0; j < E; ++j) qt.v[j] = (int32_t)((w >> (8 * j)) & 255u);
                } else {
                    qt = *reinterpret_cast<const VecE<int32_t, E>*>(t_p + base);
                }
                return qt;
            };
            VecE<int32_t, E> qt;
            if constexpr (kTFirst) {
                qt = load_t();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (STAMP) {
                const VecE<int32_t, E> qs = *reinterpret_cast<const VecE<int32_t, E>*>(a.stamp + base);
#pragma unroll
                for (int j = 0; j < E; ++j) st[j] = qs.v[j];
            }
#pragma unroll
            for (int j = 0; j < E; ++j) {
                sg[j] = a.sigma;
                er[j] = (T)0;
            }
            // (fishing-v11's indices: their pointer is a field of the struct -- in the exact kernels the load goes out behind the ones
            // that need nothing but preloaded arguments)
            auto load_kinds = [&]() {
                if constexpr (kMixed) {
                    const VecE<int32_t, E> qk = *reinterpret_cast<const VecE<int32_t, E>*>(ex.model_idx + base);
#pragma unroll
                    for (int j = 0; j < E; ++j) kind[j] = qk.v[j];
                    // (an index outside the zoo steps as Beverton-Holt in every kernel: clamped HERE, once -- a second, clamped copy next to
                    // the loaded one, kept for the quad's write-back after a redraw, cost the float32 kernel its eighth wave per SIMD)
#pragma unroll
                    for (int j = 0; j < E; ++j)
                        kind[j] = (kind[j] >= 0 && kind[j] < FISHING_N_KINDS) ? kind[j] : FISHING_KIND_BEVERTON_HOLT;
                }
            };
            constexpr bool kKindsLate = kExact && kMixed;
            if constexpr (!kKindsLate) load_kinds();
            auto load_sigma = [&]() {
                if (SIGARR) {
                    const VecE<T, E> qs = *reinterpret_cast<const VecE<T, E>*>(a.sigma_arr + base);
#pragma unroll
                    for (int j = 0; j < E; ++j) sg[j] = qs.v[j];
                }
            };
            if constexpr (!kTFirst) load_sigma();       // (kTFirst: last -- its pointer is a field of the struct, and nothing needs it early)
            const VecE<T, E> q = *reinterpret_cast<const VecE<T, E>*>(obs_p + base);
            // In every other exact kernel the year counters' load goes out BEHIND the actions' (not in front): the
            // arithmetic starts on observations and actions -- first wait vmcnt(2) of four loads instead of vmcnt(1) --, the counters
            // are needed at its end.  Builds alternating (profiles/r05_t_late.jsonl): the metric 18.92 -> 18.67 us, fishing-v0 at 2^22
            // 18.93 -> 18.73, fishing-v11 float32 24.7 -> 24.3; N = 2^21 (one exact round of waves, forward walk) 8.33 -> 8.53: the one
            // size that loses; below 2^21 the better of the two within a noisy box.  (The actions' load in FRONT of the observations' --
            // the one stream that is never in the L2s first -- loses: its address takes 20 instructions more, 18.57 -> 18.80 us.)
            constexpr bool kTLate = kExact && !kTFirst;
            if constexpr (!kTFirst && !kTLate) qt = load_t();
#pragma unroll
            for (int j = 0; j < E; ++j) {
                obs[j] = q.v[j];
                rr[j] = a.pr;
                KK[j] = a.pK;
                z[j] = (T)0;
                a_i[j] = 0;
                a_f[j] = 0.0f;
            }
            if (MODEL == FISHING_MODEL_V0) {
                VecE<int32_t, E> qa;
                if (E == 4 && NTA) {
                    typedef int32_t i4 __attribute__((ext_vector_type(4)));
                    const i4 w = __builtin_nontemporal_load(reinterpret_cast<const i4*>((const int32_t*)action_p + cbase));
#pragma unroll
                    for (int j = 0; j < 4; ++j) qa.v[j] = w[j];
                } else {
                    qa = *reinterpret_cast<const VecE<int32_t, E>*>((const int32_t*)action_p + cbase);
                }
#pragma unroll
                for (int j = 0; j < E; ++j) a_i[j] = qa.v[j];
            } else {
                VecE<float, E> qa;
                if (E == 4 && NTA) {
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    const f4 w = __builtin_nontemporal_load(reinterpret_cast<const f4*>((const float*)action_p + cbase));
#pragma unroll
                    for (int j = 0; j < 4; ++j) qa.v[j] = w[j];
                } else {
                    qa = *reinterpret_cast<const VecE<float, E>*>((const float*)action_p + cbase);
                }
#pragma unroll
                for (int j = 0; j < E; ++j) a_f[j] = qa.v[j];
            }
            if constexpr (kTLate) qt = load_t();
            if constexpr (kKindsLate) load_kinds();
#pragma unroll
            for (int j = 0; j < E; ++j) t[j] = qt.v[j];
            if constexpr (kTFirst) load_sigma();
            if (kPerEnv && !DERIVED) {
                const VecE<T, E> qr = *reinterpret_cast<const VecE<T, E>*>(a.r + base);
                const VecE<T, E> qk = *reinterpret_cast<const VecE<T, E>*>(a.K + base);
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    rr[j] = qr.v[j];
                    KK[j] = qk.v[j];
                }
            }
            if (DRIFT) {
                const VecE<T, E> qr = *reinterpret_cast<const VecE<T, E>*>(a.r + base);
#pragma unroll
                for (int j = 0; j < E; ++j) rr[j] = qr.v[j];
            }
            if (RET) {
                const VecE<T, E> qe = *reinterpret_cast<const VecE<T, E>*>(ep_return_p + base);
#pragma unroll
                for (int j = 0; j < E; ++j) er[j] = qe.v[j];
            }
        }
        // the loads above must be in flight BEFORE the ~100-instruction Philox block starts: without
        // this fence the scheduler hoists the (independent) generator above them in some variants
        __builtin_amdgcn_sched_barrier(0);
        batch_args();
        // (The zoo's kernels branch per env, and in every such block LLVM re-loads the wave-uniform arguments it needs from the
        // kernarg segment -- ten s_load + s_waitcnt lgkmcnt(0) round trips through fishing-v11's arithmetic.  Pinning them in SGPRs
        // through an empty asm removes the loads and gains nothing: eight waves per SIMD hide them; the single-function float32
        // kernels lose 1 %: profiles/r05_pin_zoo_args.jsonl.)
        // (graph replay: the origin of the last reset() sits next to the step counter and is read behind the tile's loads like it,
        // in the same batch -- SCALAR loads: as a vector load it sat in the queue the tile's loads return through)
        if (DERIVED && a.counter) {
            uint64_t c = 0;
            asm volatile("s_load_dwordx2 %0, %3, 0x0\n\ts_load_dwordx2 %1, %3, 0x8\n\ts_load_dwordx2 %2, %3, 0x10\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(c), "=&s"(origin_step), "=&s"(origin_counter) : "s"(a.counter) : "memory");
            if (!ZZ) step_counter = c + step_counter_arg;
        } else if (!ZZ) {
            step_counter = read_counter();
        }
        if (noise == kNoisePhilox) {
            float zq[E];
            if constexpr (E == 4) {
                noise_quad(seed_it, (env_offset + (uint64_t)base) >> 2, step_counter, zq);
            } else {        // this lane's half of the quad's block (noise_quad: (w0, w1) -> envs 4q, 4q+1; (w2, w3) -> 4q+2, 4q+3)
                const Words4 w = philox_block(seed_it, (env_offset + (uint64_t)base) >> 2, step_counter, kStreamNoise);
                const bool upper = ((env_offset + (uint64_t)base) & 2u) != 0;
                box_muller(upper ? w.w2 : w.w0, upper ? w.w3 : w.w1, zq[0], zq[1]);
            }
            // ... and the generator (which needs none of the loaded data) runs under their latency: the first s_waitcnt vmcnt
            // lands after it, at the first use of a loaded register -- IF the normals are computed here.  Where control flow
            // follows (fishing-v4's (K, r) derivation, the power / quotient branches of Beverton-Holt, Myers and May), LLVM sinks
            // the whole generator into the block that first reads z, BEHIND the wait for the tile's loads (an IR-level move:
            // no scheduling fence binds it).  An empty asm that reads the normals pins them here.
            // Only where it is needed (the kernels without such control flow keep their order by themselves, and pay 0.5 % for the
            // constraint): 13.03 -> 12.43 us for fishing-v4's config-5 shard, 1-2 % for fishing-v6 / v7 / v8 / v10 in both layouts
            // (profiles/r05_noise_pin.jsonl).  fishing-v11: the generator stays in front of the table's barrier.
            constexpr bool kPinNoise = (kPerEnv && (F & feat::DERIVED) != 0) || kMixed ||
                                       (kZoo && (kZooKind == FISHING_KIND_BEVERTON_HOLT || kZooKind == FISHING_KIND_MYERS ||
                                                 kZooKind == FISHING_KIND_MAY));
            if constexpr (kPinNoise) {
#pragma unroll
                for (int j = 0; j < E; ++j) asm volatile("" : "+v"(zq[j]));
            }
#pragma unroll
            for (int j = 0; j < E; ++j) z[j] = (T)zq[j];
        }
        // The caller's normals are loaded HERE, in the generator's else: issued with the tile's other loads, their destination
        // registers are the generator's too, and the catch-alls (noise mode at run time) then wait for EVERY load of the tile
        // -- s_waitcnt vmcnt(0) -- before the generator's first write to them, on the path that never issued that load.
        else if (noise == kNoiseExt) {
            const VecE<T, E> qz = *reinterpret_cast<const VecE<T, E>*>(a.z_ext + cbase);
#pragma unroll
            for (int j = 0; j < E; ++j) z[j] = qz.v[j];
        }
        if (DERIVED) {      // needs the year counters: after the noise block, which hid their latency
            // (tile-uniform: the last env of this workgroup's tile and the counters all below 2^32 -> 32-bit integer work)
            if (derive_fits_32(env_offset + (uint64_t)(tile + 1) * (uint64_t)kTileEnvs - 1u, step_counter, origin_step, origin_counter)) {
#pragma unroll
                for (int j = 0; j < E; ++j)
                    derive_model_error<T, true>(seed_it, env_offset + (uint64_t)base + j, step_counter, t[j], origin_step,
                                                origin_counter, a.K_mean, a.r_mean, a.sigma_p, KK[j], rr[j], st[j]);
            } else {
#pragma unroll
                for (int j = 0; j < E; ++j)
                    derive_model_error<T>(seed_it, env_offset + (uint64_t)base + j, step_counter, t[j], origin_step,
                                          origin_counter, a.K_mean, a.r_mean, a.sigma_p, KK[j], rr[j], st[j]);
            }
        }
        // LATCH: envs that were finished before this step (only possible without auto-reset) must not be recorded
        // again.  The test reuses the population env_step computes anyway.
        bool stale[E];
#pragma unroll
        for (int j = 0; j < E; ++j) stale[j] = false;
        T obs_next[E], rew[E];
        int32_t t_next[E];
        bool dn[E];
        bool stepped = false;
        if constexpr (kMixed) {
            if (!SIGARR) {      // wave-uniform: every env's coefficients from the LDS table (fishing_common.h: zoo_draw_lut_tile)
                // (one LDS store of the word loaded at the tile's start, behind the noise block that hid its latency -- filled
                // from ex.zoo, the first wave strung nine dependent s_load batches and 35 conversions in front of this barrier,
                // at which its three sister waves wait)
                if (threadIdx.x < kZooLutSize) zoo_lut[threadIdx.x] = lut_word;
                __syncthreads();      // (the table)
                T xh[E], hv[E], xn[E];
                int kk[E];
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const T quota = quota_cts<T>((T)a_f[j], KK[j]);
                    if (RET && LATCH) stale[j] = was_done<T>(obs[j], t[j], KK[j], a.Tmax);
                    const T x = (obs[j] + (T)1) * KK[j];
                    hv[j] = (quota < x) ? quota : x;
                    const T d = x - hv[j];
                    xh[j] = d;       // (max(d, 0.0) is the identity here: stock_after_harvest)
                    xn[j] = (T)0;
                    kk[j] = kind[j];        // (clamped where it was loaded)
                }
                zoo_draw_lut_tile<T, E>(kk, xh, z, ex.zoo, zoo_lut, xn);
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    obs_next[j] = div_K<T>(xn[j], KK[j], a.dk) - (T)1;       // (the env's scalar K: a multiply when it is a power of two)
                    rew[j] = ((T)0 > hv[j]) ? (T)0 : hv[j];
                    t_next[j] = t[j] + 1;
                    dn[j] = (t_next[j] > a.Tmax) || (xn[j] <= (T)0);
                }
                stepped = true;
            }
        }
        if (!stepped) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const T quota = (MODEL == FISHING_MODEL_V0) ? quota_int<T>(a_i[j], a.n_actions, KK[j])
                                                        : quota_cts<T>((T)a_f[j], KK[j]);
            if (RET && LATCH) stale[j] = was_done<T>(obs[j], t[j], KK[j], a.Tmax);
            if constexpr (kMixed) {         // per-env sigma: a straight per-lane switch over the growth functions
                env_step_zoo_mixed<T>(obs[j], t[j], quota, z[j], kind[j], ex.zoo, sg[j], KK[j], a.Tmax, obs_next[j], rew[j], dn[j],
                                      t_next[j], a.dk);
            } else if constexpr (kZoo) {
                GrowthT<T> P = a.growth;
                if (SIGARR) P.sigma = sg[j];
                if (DRIFT) {                             // growth_models.py:151: drift first, then draw
                    rr[j] = rr[j] + a.alpha;
                    P.r = rr[j];
                    env_step_zoo<T, kZooKind, true>(obs[j], t[j], quota, z[j], kZooKind, P, KK[j], a.Tmax, obs_next[j],
                                                    rew[j], dn[j], t_next[j], a.dk);
                } else {
                    env_step_zoo<T, kZooKind, false>(obs[j], t[j], quota, z[j], kZooKind, P, KK[j], a.Tmax, obs_next[j],
                                                     rew[j], dn[j], t_next[j], a.dk);
                }
            } else {
                env_step<T, MODEL>(obs[j], t[j], quota, z[j], rr[j], KK[j], sg[j], a.C, a.Tmax, obs_next[j], rew[j],
                                   dn[j], t_next[j], (F & feat::KP2) ? DivK{true, a.dk.inv_f, a.dk.inv_d} : DivK{false, 0.0f, 0.0});
            }
        }
        }
#pragma unroll
        for (int j = 0; j < E; ++j) {       // scratch envs never finish, and their year counter stays put (no overflow, ever)
            dn[j] = dn[j] && live;
            t_next[j] = live ? t_next[j] : 0;
        }
        auto store_outputs = [&]() {
            // reward and done are write-only streams nobody re-reads inside the step loop: nontemporal
            // stores (0.5-0.7 % at N = 2^22, 1.5 % at 2^24 / 2^26; profiles/r01g_lean_nt_stores.txt)
            typedef T ntE __attribute__((ext_vector_type(E)));
            typedef std::conditional_t<E == 4, uint32_t, uint16_t> bytesE;
            ntE qv, qt;
            uint32_t packed = 0, nibble = 0;
#pragma unroll
            for (int j = 0; j < E; ++j) {
                qv[j] = rew[j];
                qt[j] = obs_next[j];
                packed |= (uint32_t)dn[j] << (8 * j);
                nibble |= (uint32_t)dn[j] << j;
            }
            __builtin_nontemporal_store(qv, reinterpret_cast<ntE*>(a.reward + base));
            __builtin_nontemporal_store((bytesE)packed, reinterpret_cast<bytesE*>(a.done + base));
            if (TERM) __builtin_nontemporal_store(qt, reinterpret_cast<ntE*>(a.terminal_obs + base));   // (still the pre-reset observation)
            if (BITS) {         // the wave's 64 * E flags as E 64-bit words (ballots, no LDS)
                const int lane = threadIdx.x & (kWave - 1);
                const uint64_t word = ballot_tile_words<E>(nibble, lane);
                const int64_t wave_env0 = (tile * kThreads + (threadIdx.x & ~(kWave - 1))) * E;
                if (lane < E) a.done_bits[(wave_env0 >> 6) + lane] = word;
            }
        };
        if (!RET) store_outputs();
        bool lane_done = false;
#pragma unroll
        for (int j = 0; j < E; ++j) lane_done |= dn[j];
        if (RET) {
#pragma unroll
            for (int j = 0; j < E; ++j) er[j] = er[j] + rew[j];
            if (__any(lane_done)) {          // wave-ballot: only waves with a finished env record
                bool fresh[E];               // the episode ended on THIS step (not: stepped on after its end)
#pragma unroll
                for (int j = 0; j < E; ++j) fresh[j] = dn[j] && !stale[j];
                record_tile<T>(fresh, er, t_next, acc);
#pragma unroll
                for (int j = 0; j < E; ++j) er[j] = (dn[j] && auto_reset) ? (T)0 : er[j];
            }
            // the record's atomic first, the tile's stores behind it (the other order measures the same: 18.66 / 18.63 us at 2^22)
            if (a.partials) add_block_partials<kThreads / kWave, kThreads / kWave>(acc, a.partials);
            store_outputs();
            VecE<T, E> qe;
#pragma unroll
            for (int j = 0; j < E; ++j) qe.v[j] = er[j];
            *reinterpret_cast<VecE<T, E>*>(ep_return_p + base) = qe;
        }
        if constexpr (kMixed) {     // growth_models.py:200: a new model for the next episode
            if (auto_reset && __any(lane_done)) {
                if (redraw_kinds<E>(seed_it, env_offset + (uint64_t)base, step_counter, kStreamAutoReset, ex.kinds, ex.n_models, dn,
                                    kind)) {
                    VecE<int32_t, E> qk;
#pragma unroll
                    for (int j = 0; j < E; ++j) qk.v[j] = kind[j];
                    *reinterpret_cast<VecE<int32_t, E>*>(ex.model_idx + base) = qk;
                }
            }
        }
        if (kPerEnv && !DERIVED) {
            if (auto_reset && __any(lane_done)) {
                const bool redrawn =
                    redraw_tile<T, MODEL>(seed_it, env_offset + (uint64_t)base, step_counter, kStreamAutoReset, a.K_mean,
                                          a.r_mean, a.sigma_p, a.x0, dn, KK, rr, obs_next, t_next);
                if (redrawn) {
                    VecE<T, E> qk, qr;
#pragma unroll
                    for (int j = 0; j < E; ++j) {
                        qk.v[j] = KK[j];
                        qr.v[j] = rr[j];
                    }
                    *reinterpret_cast<VecE<T, E>*>(a.K + base) = qk;
                    *reinterpret_cast<VecE<T, E>*>(a.r + base) = qr;
                }
            }
        } else {        // (fishing-v4 with derived parameters restarts at x0 whatever the next episode's K)
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const bool rs = dn[j] && auto_reset;
                obs_next[j] = rs ? robs_scalar : obs_next[j];
                t_next[j] = rs ? 0 : t_next[j];
                st[j] = rs ? 0 : st[j];         // (from here on the year counter dates the episode again)
            }
            if (STAMP && auto_reset && __any(lane_done)) {
                VecE<int32_t, E> qs;
#pragma unroll
                for (int j = 0; j < E; ++j) qs.v[j] = st[j];
                *reinterpret_cast<VecE<int32_t, E>*>(a.stamp + base) = qs;
            }
        }
        {
            VecE<T, E> qo;
            VecE<int32_t, E> qt;
#pragma unroll
            for (int j = 0; j < E; ++j) {
                qo.v[j] = obs_next[j];
                qt.v[j] = t_next[j];
            }
            *reinterpret_cast<VecE<T, E>*>(obs_p + base) = qo;
            if (DRIFT) {
                VecE<T, E> qr;
#pragma unroll
                for (int j = 0; j < E; ++j) qr.v[j] = rr[j];
                *reinterpret_cast<VecE<T, E>*>(a.r + base) = qr;
            }
            if (T8) {
                typedef std::conditional_t<E == 4, uint32_t, uint16_t> bytesE;
                *reinterpret_cast<bytesE*>(reinterpret_cast<uint8_t*>(t_p) + base) = (bytesE)pack_t4(t_next);
            }
            else *reinterpret_cast<VecE<int32_t, E>*>(t_p + base) = qt;
        }
    };

    do_tile(blockIdx.x);
}

// ---------------------------------------------------------------- the floor a lean launch stands on (diagnostic)
// The grid, workgroup size, argument list (hence kernarg size and preload) of step_kernel_lean<float, MODEL, F, 4>, and
//   MODE 0: an empty body -- what launching that grid costs, whatever it moves;
//   MODE 1: a copy over the step's streams with the step's access shape -- 16-byte loads of obs, action, t, ep_return, the
//           stores of obs, reward, done (one dword), t, ep_return, and nothing in between but an add.
// bench.py times both beside the step kernel at the launch-bound shard sizes (N = 2^19 .. 2^21), where "fraction of the HBM
// spec" says nothing about how close the kernel is to ITS floor.
template <int MODE>
__global__ void __launch_bounds__(kTileEnvsLean / 4)
step_floor_kernel(float* const obs_p, const void* const action_p, int32_t* const t_p, float* const ep_return_p,
                  const int64_t n_live_p, const LeanArgs<float> a, const LeanNoExtra ex, const int64_t ntiles,
                  const uint64_t env_offset, const uint64_t seed, const uint64_t step_counter_arg) {
    if constexpr (MODE == 1) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef int32_t i4 __attribute__((ext_vector_type(4)));
        const int64_t base = ((int64_t)blockIdx.x * (kTileEnvsLean / 4) + threadIdx.x) * 4;
        const f4 o = *reinterpret_cast<const f4*>(obs_p + base);
        const f4 ac = *reinterpret_cast<const f4*>((const float*)action_p + base);
        const i4 t = *reinterpret_cast<const i4*>(t_p + base);
        const f4 er = *reinterpret_cast<const f4*>(ep_return_p + base);
        const f4 rew = o + ac;
        __builtin_nontemporal_store(rew, reinterpret_cast<f4*>(a.reward + base));
        const uint32_t flags = (uint32_t)(t[0] & 1) | ((uint32_t)(t[1] & 1) << 8) | ((uint32_t)(t[2] & 1) << 16) | ((uint32_t)(t[3] & 1) << 24);
        __builtin_nontemporal_store(flags, reinterpret_cast<uint32_t*>(a.done + base));
        *reinterpret_cast<f4*>(ep_return_p + base) = er + rew;
        *reinterpret_cast<f4*>(obs_p + base) = o;
        *reinterpret_cast<i4*>(t_p + base) = t;
    }
}

// ---------------------------------------------------------------- host side
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
        (p->launch_threads < 64 || p->launch_threads > kStepMaxThreads || (p->launch_threads & 63)))
        return FISHING_ERR_SIZE;
    if (p->launch_blocks < 0) return FISHING_ERR_SIZE;
    if ((p->flags & FISHING_FLAG_T_U8) && (p->Tmax < 0 || p->Tmax > 254)) return FISHING_ERR_SIZE;
    if (!b->obs || !b->t) return FISHING_ERR_NULL;
    const bool derived = p->model == FISHING_MODEL_V4 && (p->flags & FISHING_FLAG_V4_DERIVED);
    if (derived && (p->flags & FISHING_FLAG_T_U8)) return FISHING_ERR_UNSUPPORTED;     // a saturating counter cannot date an episode
    if (b->v4_stamp && !derived) return FISHING_ERR_UNSUPPORTED;                          // origin stamps belong to the derived mode
    if (p->model == FISHING_MODEL_V4 && !derived && (!b->r || !b->K)) return FISHING_ERR_NULL;
    // (clip_param relies on it; the reference turns a non-finite mean into NaN populations)
    if (p->model == FISHING_MODEL_V4 && !(std::isfinite(p->K_mean) && std::isfinite(p->r_mean) && std::isfinite(p->sigma_p)))
        return FISHING_ERR_VALUE;
    if (b->return_partials && !b->ep_return) return FISHING_ERR_NULL;
    if (b->counter && (((uintptr_t)b->counter) & 7u)) return FISHING_ERR_ALIGN;
    const void* ptrs[] = {b->obs,  b->action, b->reward, b->done,         b->done_bits, b->t,           b->r,
                          b->K,    b->sigma,  b->z_ext,  b->terminal_obs, b->ep_return, b->return_partials,
                          b->model_idx, b->v4_stamp};
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

int noise_mode(const FishingParams* p, const FishingBuffers* b) {
    // the reference draws a normal even at sigma == 0 (quirk B3) but multiplies it by 0:
    // skipping the generator there changes no result bit.
    bool quiet = (p->sigma == 0.0 && !b->sigma);
    if (p->model == FISHING_MODEL_V11) {
        quiet = !b->sigma;
        for (int k = 0; k < FISHING_N_KINDS; ++k) quiet = quiet && p->zoo[k].sigma == 0.0;
    }
    return b->z_ext ? kNoiseExt : (quiet ? kNoiseNone : kNoisePhilox);
}

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
    q.stamp = b.stamp ? b.stamp + off : nullptr;
    return q;
}

template <typename T>
int launch_general(const ParamsT<T>& pt, const BuffersT<T>& bt, int noise, int64_t n, uint64_t env_offset, uint64_t seed,
                   uint64_t step_counter, int blocks, int threads, hipStream_t s, std::string* name) {
    return with_general_tag(pt.model, [&](auto tag) {
        constexpr int kTag = decltype(tag)::value;
        if (name) {
            char buf[96];
            std::snprintf(buf, sizeof buf, "fishing::step_kernel<%s, %d>", sizeof(T) == 4 ? "float" : "double", kTag);
            *name = buf;
            return (int)FISHING_OK;
        }
        return launch_kernel(step_kernel<T, kTag>, blocks, threads, s, pt, bt, n, env_offset, seed, step_counter, noise);
    });
}

// ---- lean dispatch: the instantiated feature masks
// One launch site: LeanCall carries what every instantiation needs; `name` set = report the kernel's name
// instead of launching it (fishing_step_kernel_name_*, the same decisions as the launch).
template <typename T>
struct LeanCall {
    const LeanArgs<T>& a;
    int64_t ntiles;
    uint64_t env_offset, seed, step_counter;
    hipStream_t s;
    std::string* name;
    const void* extra;       // LeanMixedArgs<T> for fishing-v11, unused otherwise
    bool two_per_thread;     // float64: E = 2 (512-thread workgroups)
};

// the fifth leading argument of a launch (see kLiveMask)
template <typename T>
int64_t walk_word(const LeanCall<T>& c, const int64_t ntiles) {
    int64_t w = c.a.n_live & kLiveMask;
    if (c.a.zz_rt) {
        constexpr int64_t G = 8;
        if (c.a.counter) w |= kWalkDeviceParityBit;
        else if ((c.step_counter & 1) && ntiles <= kWalkMask) w |= (ntiles & ~(G - 1)) << kWalkShift;
        if (c.a.nta_rt) w |= kWalkNtaBit;
    }
    return w;
}

template <typename T, int MODEL, int F, int E = 4>
int lean_launch(const LeanCall<T>& c) {
    if (c.name) {
        char buf[96];
        std::snprintf(buf, sizeof buf, "fishing::step_kernel_lean<%s, %d, %d, %d>", sizeof(T) == 4 ? "float" : "double", MODEL, F, E);
        *c.name = buf;
        return FISHING_OK;
    }
    if (c.a.n_live != INT64_MAX && c.a.n_live > kLiveMask) return FISHING_ERR_SIZE;     // (a launch steps at most 2^26 envs)
    LeanExtra<T, MODEL> ex{};
    if constexpr (MODEL == kModelZooMixed) ex = *static_cast<const LeanMixedArgs<T>*>(c.extra);
    // a workgroup of 1024 / E threads per 1024-env tile (feat::ONE): 256 x 4 envs, or 512 x 2 for the float64 layout at
    // cache-resident sizes; one return_partials slot per tile either way
    static_assert((F & feat::ONE) != 0, "every lean form is a one-tile form");
    const int64_t nt = c.ntiles;
    return launch_kernel(step_kernel_lean<T, MODEL, F, E>, (int)nt, kTileEnvsLean / E, c.s, c.a.obs, c.a.action, c.a.t,
                         c.a.ep_return, walk_word(c, nt), c.a, ex, nt, c.env_offset, c.seed, c.step_counter);
}

// the catch-all mask of a (T, MODEL): every optional stream "may be there", noise mode at run time
template <int MODEL>
constexpr int catch_all_mask() {
    int f = feat::kNoiseRT | feat::RET | feat::SIGARR | feat::T8 | feat::TERM | feat::BITS | feat::OPT | feat::LATCH;
    if (MODEL == FISHING_MODEL_V4) f |= feat::DERIVED | feat::STAMP;
    if (MODEL == kModelZoo + FISHING_KIND_BEVERTON_HOLT) f |= feat::DRIFT;
    return f;
}

// `req` = the exact mask of the request.  The hot requests (what bench.py and a training loop issue) have
// their own instantiation; everything else takes the catch-all of its (T, MODEL).
template <typename T, int MODEL>
int lean_dispatch(int req, const LeanCall<T>& c) {
    using namespace feat;
    constexpr int P = kNoisePhilox;
    // Every four-envs-per-thread instantiation -- exact or catch-all -- is a one-tile-per-workgroup form (feat::ONE): a
    // lean launch always covers its tiles one to one (step_dispatch_range).  Only the float64 two-envs-per-thread forms
    // keep the tile loop.  (Catch-all as a one-tile form, float32 with terminal observations + returns: N = 2^21 12.8 ->
    // 10.35 us, 2^22 23.6 -> 21.35; 62 instead of 106 SGPRs, 52 instead of 67 VGPRs: profiles/r03_catch_all_one_tile.jsonl.)
#define FISHING_LEAN_CASE(MASK) \
    case (MASK): return lean_launch<T, MODEL, (MASK) | ONE>(c)
    if constexpr (sizeof(T) == 4 && !is_zoo_tag(MODEL) && MODEL != FISHING_MODEL_V4) {
        // fishing-v0/v1/v2, float32, in-kernel noise: bare / with the return record.  K a power of two (KP2) skips the division;
        // any other K keeps the correctly rounded division (17.5 -> 16.2 us bare, 22.6 -> 21.5 us with returns
        // against the catch-all at N = 2^22).
        switch (req) {
            FISHING_LEAN_CASE(P | KP2);
            FISHING_LEAN_CASE(P | KP2 | RET);
            FISHING_LEAN_CASE(P);
            FISHING_LEAN_CASE(P | RET);
            default: break;
        }
        if constexpr (MODEL == FISHING_MODEL_V1) {
            switch (req) {      // the compact layout bench.py --compact measures
                FISHING_LEAN_CASE(P | KP2 | T8);
                FISHING_LEAN_CASE(P | KP2 | T8 | RET);
                default: break;
            }
        }
    }
    if constexpr (sizeof(T) == 4 && MODEL == FISHING_MODEL_V4) {
        // fishing-v4 (per-env K: always the true division): stored or derived (K, r), sigma scalar or array (BASELINE
        // config 5), bare / with the return record
        switch (req) {
            FISHING_LEAN_CASE(P);
            FISHING_LEAN_CASE(P | RET);
            FISHING_LEAN_CASE(P | SIGARR);
            FISHING_LEAN_CASE(P | SIGARR | RET);
            FISHING_LEAN_CASE(P | DERIVED);
            FISHING_LEAN_CASE(P | DERIVED | RET);
            FISHING_LEAN_CASE(P | DERIVED | SIGARR);
            FISHING_LEAN_CASE(P | DERIVED | SIGARR | RET);
            default: break;
        }
    }
    // float32 zoo (one growth function each; fishing-v11: growth function per env, coefficients from the LDS table):
    // bare / with the return record.  Measured against the catch-all at N = 2^22: fishing-v9 16.05 vs 17.7 us.  (The
    // float64 parity layout gains under 1 % from exact instantiations -- 26.65 vs 26.87 us, it is bound by its
    // 32-byte-per-lane access shape -- and runs on its catch-alls: profiles/r02_ab_variants.jsonl.)
    if constexpr (sizeof(T) == 4 && is_zoo_tag(MODEL)) {
        switch (req) {
            FISHING_LEAN_CASE(P);
            FISHING_LEAN_CASE(P | RET);
            default: break;
        }
    }
    // fishing-v10 = Beverton-Holt with the per-env drifting r stream
    if constexpr (sizeof(T) == 4 && MODEL == kModelZoo + FISHING_KIND_BEVERTON_HOLT) {
        switch (req) {
            FISHING_LEAN_CASE(P | DRIFT);
            FISHING_LEAN_CASE(P | DRIFT | RET);
            default: break;
        }
    }
#undef FISHING_LEAN_CASE
    // float64: two envs per thread up to ~512 MB per step (see the kernel and step_dispatch_range).  Relieved of the
    // 32-byte access shape the layout feels its arithmetic -- two IEEE float64 divisions per env, ~25 instructions each --
    // so fishing-v0/v1/v2 with K a power of two have exact instantiations there (no division, no option tests)
    // fishing-v11 in float64: the growth functions' log / exp make it VALU-bound, and its catch-all -- every option a
    // run-time test, plus the straight per-lane switch of the per-env-sigma path: five inlined growth functions for each of
    // a thread's four envs -- spilled (36 B of scratch per lane, 16 SGPRs) at 105 VGPRs; the two hot requests get exact forms
    if constexpr (sizeof(T) == 8 && MODEL == kModelZooMixed) {
        if (c.two_per_thread) {
            switch (req) {
                case (P): return lean_launch<T, MODEL, (P | ONE), 2>(c);
                case (P | RET): return lean_launch<T, MODEL, (P | RET | ONE), 2>(c);
                default: return lean_launch<T, MODEL, catch_all_mask<MODEL>() | feat::ONE, 2>(c);
            }
        }
        switch (req) {
            case (P): return lean_launch<T, MODEL, (P | ONE)>(c);
            case (P | RET): return lean_launch<T, MODEL, (P | RET | ONE)>(c);
            default: break;
        }
    }
    if constexpr (sizeof(T) == 8 && MODEL != kModelZooMixed) {
        if (c.two_per_thread) {
            // the float64 zoo's hot requests as exact two-envs-per-thread forms: 370 instead of 484 VALU instructions per thread,
            // worth 1-2 % (fishing-v5 33.5 -> 32.8 us = 0.85 of the spec, v9 33.7 -> 33.1, v8 and v7 unchanged:
            // profiles/r05_zoo_f64_exact.jsonl) -- the layout sits 6 % behind fishing-v1's 31.1 us whatever it executes
            if constexpr (is_zoo_tag(MODEL)) {
                switch (req) {
                    case (P): return lean_launch<T, MODEL, (P | ONE), 2>(c);
                    case (P | RET): return lean_launch<T, MODEL, (P | RET | ONE), 2>(c);
                    default: break;
                }
                if constexpr (MODEL == kModelZoo + FISHING_KIND_BEVERTON_HOLT) {
                    switch (req) {
                        case (P | DRIFT): return lean_launch<T, MODEL, (P | DRIFT | ONE), 2>(c);
                        case (P | DRIFT | RET): return lean_launch<T, MODEL, (P | DRIFT | RET | ONE), 2>(c);
                        default: break;
                    }
                }
            }
            if constexpr (!is_zoo_tag(MODEL) && MODEL != FISHING_MODEL_V4) {
                switch (req) {
                    case (P | KP2): return lean_launch<T, MODEL, (P | KP2 | ONE), 2>(c);
                    case (P | KP2 | RET): return lean_launch<T, MODEL, (P | KP2 | RET | ONE), 2>(c);
                    default: break;
                }
            }
            return lean_launch<T, MODEL, catch_all_mask<MODEL>() | feat::ONE, 2>(c);
        }
    }
    return lean_launch<T, MODEL, catch_all_mask<MODEL>() | feat::ONE>(c);
}

// Where a step() request runs.  lean: whole 1024-env tiles on step_kernel_lean (+ the ragged tail on one
// workgroup of the general kernel); otherwise everything on the general kernel.
// (`b`: which streams are there; `bt`: where this range of envs begins in them)
template <typename T>
int step_dispatch_range(const FishingParams* p, const ParamsT<T>& pt, int64_t n, int64_t env_offset, const FishingBuffers* b,
                        const BuffersT<T>& bt, uint64_t seed, uint64_t step_counter, hipStream_t s, std::string* name) {
    const int noise = noise_mode(p, b);
    const bool t8 = (p->flags & FISHING_FLAG_T_U8) != 0;
    const bool derived = p->model == FISHING_MODEL_V4 && (p->flags & FISHING_FLAG_V4_DERIVED);
    const int64_t tile = kTileEnvsLean;       // (256 * kEnvsPerThread)
    // (under FISHING_FLAG_PADDED_TILES also batches below one tile: 3.1 instead of 5.3 us per step at N = 1000)
    const bool pad_ok = (p->flags & FISHING_FLAG_PADDED_TILES) != 0 && (n % kEnvsPerThread) == 0;
    // The lean kernel runs a workgroup per tile (step_dispatch hands it at most kPartialSlots tiles); an explicit launch
    // shape that does not cover the tiles one to one (experiments, tests) gets the general kernel in that shape.
    const int64_t tiles_up = (n + tile - 1) / tile;
    const bool covers = tiles_up <= kPartialSlots && (!p->launch_blocks || p->launch_blocks >= tiles_up);
    const bool lean = !(p->flags & FISHING_FLAG_DIAG_GENERAL_KERNEL) && b->reward && b->done &&
                      (p->launch_threads == 0 || p->launch_threads == 256) && (n >= tile || pad_ok) && covers;
    if (!lean) {
        int blocks, threads;
        launch_shape(p, n, blocks, threads);
        return launch_general<T>(pt, bt, noise, n, env_offset, seed, step_counter, blocks, threads, s, name);
    }
    // FISHING_FLAG_PADDED_TILES: the caller's state buffers have room for whole tiles -> the last, partial tile runs in
    // the same launch (its scratch envs are stepped but never finish) instead of a second launch of the general kernel.
    // (Whole quads only: the caller-owned action stream is read 16 bytes at a time.)
    const bool padded = pad_ok && (n % tile) != 0;
    const int64_t ntiles = padded ? (n + tile - 1) / tile : n / tile;
    const int64_t n_full = padded ? n : ntiles * tile;
    const bool drift = p->model == FISHING_MODEL_V10;
    LeanArgs<T> a{bt.obs,   bt.action, bt.reward, bt.done,  bt.t,    bt.r,     bt.K,     bt.ep_return, bt.partials,
                  bt.counter, bt.sigma, bt.terminal_obs, bt.done_bits, bt.z_ext, bt.stamp, pt.r, pt.K, pt.sigma, pt.C, pt.x0,
                  pt.r_mean, pt.K_mean, pt.sigma_p, pt.Tmax, pt.n_actions, (uint32_t)(p->flags & FISHING_FLAG_AUTO_RESET),
                  noise, (uint32_t)t8, (uint32_t)derived, (uint32_t)drift, 0u, 0u, padded ? n : INT64_MAX, pt.origin_step,
                  pt.origin_counter, pt.growth, pt.alpha, make_divk((double)pt.K), (T)(pt.x0 / pt.K - (T)1)};
    // Launch geometry: a workgroup per tile at every size -- a grid of one-tile workgroups keeps the memory system full
    // through the dispatcher where a capped, looping grid has one tile's loads in flight per workgroup (with returns 2^24
    // 83 -> 80 us, 2^26 376 -> 345; float64 bare 2^24 108.6 -> 96.7: profiles/r03_one_tile_large_n.jsonl).  Batches beyond
    // the 65536 slots are stepped in ranges of 2^26 envs (step_dispatch below).
    int req = noise;
    if (b->ep_return) req |= feat::RET;
    if (b->ep_return && !(p->flags & FISHING_FLAG_AUTO_RESET)) req |= feat::LATCH;
    if (b->sigma) req |= feat::SIGARR;
    if (t8) req |= feat::T8;
    if (b->terminal_obs) req |= feat::TERM;
    if (b->done_bits) req |= feat::BITS;
    if (derived) req |= feat::DERIVED;
    if (derived && b->v4_stamp) req |= feat::STAMP;
    if (drift) req |= feat::DRIFT;
    if (a.dk.pow2 && is_core_model(p->model) && p->model != FISHING_MODEL_V4) req |= feat::KP2;
    // The zig-zag walk (see the kernel) from ~100 MB per step, nontemporal action loads from ~200 MB
    const int64_t step_bytes = n_full * (int64_t)((sizeof(T) == 4 ? 25 + (b->ep_return ? 8 : 0) + (b->sigma ? 4 : 0)
                                                                  : 37 + (b->ep_return ? 16 : 0) + (b->sigma ? 8 : 0)) + (b->v4_stamp ? 8 : 0));
    a.zz_rt = (step_bytes >= FISHING_XZZ_MIN_BYTES) ? 1u : 0u;
    a.nta_rt = (a.zz_rt && step_bytes >= FISHING_NTA_MIN_BYTES) ? 1u : 0u;
    LeanMixedArgs<T> mixed{};
    if (p->model == FISHING_MODEL_V11) {
        mixed.model_idx = bt.model_idx;
        mixed.n_models = pt.n_models;
        for (int k = 0; k < FISHING_N_KINDS; ++k) {
            mixed.kinds[k] = pt.kinds[k];
            mixed.zoo[k] = pt.zoo[k];
        }
        zoo_lut_rows<T>(mixed.zoo, mixed.lut);
    }
    // float64 with two envs per thread up to ~512 MB per step (N <= 2^23); four per thread beyond, where the access shape
    // stops mattering.  As 512-thread one-tile workgroups the two-per-thread forms win from the smallest batch on -- exact
    // and catch-all alike (the tile-loop form's catch-all only paid from ~105 MB per step) -- and at 2^23 too: bare 47-50 ->
    // 42.6 us, with returns 67-69 -> 66.3; 2^24 equal, 2^25 with returns 4 % behind (profiles/r03_f64_one_tile_512_threads.jsonl)
    const bool two = sizeof(T) == 8 && step_bytes < FISHING_F64_E2_MAX_BYTES && !p->launch_blocks;
    const LeanCall<T> call{a, ntiles, (uint64_t)env_offset, seed, step_counter, s, name, &mixed, two};
    const int rc = with_model_tag(p->model, [&](auto tag) {
        constexpr int kTag = decltype(tag)::value;
        return lean_dispatch<T, kTag>(req, call);
    });
    if (rc != 0 || n_full == n || name) return rc;
    // ragged tail (< 1024 envs): one workgroup of the general kernel
    const BuffersT<T> tb = offset_buffers<T>(bt, n_full, t8);
    return launch_general<T>(pt, tb, noise, n - n_full, env_offset + n_full, seed, step_counter, 1, 256, s, nullptr);
}

// A launch covers at most kPartialSlots tiles = 2^26 envs (a workgroup per tile, one return_partials slot each); larger
// batches run as consecutive ranges on the same stream -- same slots again, added in stream order, so the record keeps
// its bits.  Odd steps take the ranges last to first, like the tiles inside them (what the previous step touched last
// is what the Infinity Cache still holds), when the host knows the step's parity (no device-resident counter).
template <typename T>
int step_dispatch(const FishingParams* p, const ParamsT<T>& pt, int64_t n, int64_t env_offset, const FishingBuffers* b,
                  uint64_t seed, uint64_t step_counter, hipStream_t s, std::string* name) {
    const BuffersT<T> bt = typed_buffers<T>(*b);
    constexpr int64_t kRange = (int64_t)kPartialSlots * 256 * kEnvsPerThread;
    if (n <= kRange || p->launch_blocks) return step_dispatch_range<T>(p, pt, n, env_offset, b, bt, seed, step_counter, s, name);
    const bool t8 = (p->flags & FISHING_FLAG_T_U8) != 0;
    const int64_t ranges = (n + kRange - 1) / kRange;
    const bool backwards = !b->counter && (step_counter & 1);
    for (int64_t k = 0; k < ranges; ++k) {
        const int64_t off = (backwards ? ranges - 1 - k : k) * kRange;
        const int64_t len = n - off < kRange ? n - off : kRange;
        const int rc = step_dispatch_range<T>(p, pt, len, env_offset + off, b, offset_buffers<T>(bt, off, t8), seed, step_counter, s, name);
        if (rc != 0 || name) return rc;     // (name: the kernel of the first whole range)
    }
    return FISHING_OK;
}

template <typename T>
int step_impl(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b, uint64_t seed,
              uint64_t step_counter, fishing_stream_t stream) {
    const int rc = check_common(p, n, env_offset, b);
    if (rc != FISHING_OK) return rc;
    if (!b->action) return FISHING_ERR_NULL;
    if (n == 0) return FISHING_OK;
    const ParamsT<T> pt = narrow_params<T>(*p);
    return step_dispatch<T>(p, pt, n, env_offset, b, seed, step_counter, (hipStream_t)stream, nullptr);
}

template <typename T>
int step_many_impl(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                   int64_t action_stride, int32_t ring_len, int32_t n_steps, uint64_t seed,
                   uint64_t step_counter, fishing_stream_t stream) {
    if (!b || !p) return FISHING_ERR_NULL;
    if (ring_len <= 0 || n_steps < 0 || action_stride < 0) return FISHING_ERR_SIZE;
    if (ring_len > 1 && (action_stride & 3)) return FISHING_ERR_ALIGN;
    const int rc = check_common(p, n, env_offset, b);
    if (rc != FISHING_OK) return rc;
    if (!b->action) return FISHING_ERR_NULL;
    if (n == 0) return FISHING_OK;
    const ParamsT<T> pt = narrow_params<T>(*p);         // once, not per step (pow / log for the zoo)
    FishingBuffers bb = *b;
    for (int32_t k = 0; k < n_steps; ++k) {
        // action elements are 4 bytes wide in every layout (f32 / i32)
        bb.action = (const char*)b->action + (size_t)(k % ring_len) * (size_t)action_stride * 4u;
        const int rc2 = step_dispatch<T>(p, pt, n, env_offset, &bb, seed, step_counter + (uint64_t)k, (hipStream_t)stream,
                                         nullptr);
        if (rc2 != FISHING_OK) return rc2;
    }
    return FISHING_OK;
}

template <typename T>
int kernel_name_impl(const FishingParams* p, int64_t n, const FishingBuffers* b, char* out, int64_t len) {
    if (!out || len <= 0) return FISHING_ERR_NULL;
    const int rc = check_common(p, n, 0, b);
    if (rc != FISHING_OK) return rc;
    const ParamsT<T> pt = narrow_params<T>(*p);
    std::string name;
    const int rc2 = step_dispatch<T>(p, pt, n > 0 ? n : 1, 0, b, 0, 0, nullptr, &name);
    if (rc2 != FISHING_OK) return rc2;
    std::snprintf(out, (size_t)len, "%s", name.c_str());
    return FISHING_OK;
}

int step_floor_impl(int32_t mode, int64_t n, const FishingBuffers* b, int32_t n_launches, fishing_stream_t stream) {
    if (!b) return FISHING_ERR_NULL;
    if (mode < 0 || mode > 1 || n <= 0 || (n % kTileEnvsLean) || n / kTileEnvsLean > kPartialSlots || n_launches < 0)
        return FISHING_ERR_SIZE;
    if (mode == 1 && (!b->obs || !b->action || !b->reward || !b->done || !b->t || !b->ep_return)) return FISHING_ERR_NULL;
    const void* ptrs[] = {b->obs, b->action, b->reward, b->done, b->t, b->ep_return};
    for (const void* q : ptrs)
        if (misaligned(q)) return FISHING_ERR_ALIGN;
    LeanArgs<float> a{};
    a.obs = (float*)b->obs;
    a.action = b->action;
    a.reward = (float*)b->reward;
    a.done = b->done;
    a.t = b->t;
    a.ep_return = (float*)b->ep_return;
    const int64_t ntiles = n / kTileEnvsLean;
    for (int32_t k = 0; k < n_launches; ++k) {
        const int rc = mode == 0 ? launch_kernel(step_floor_kernel<0>, (int)ntiles, kTileEnvsLean / 4, (hipStream_t)stream, a.obs,
                                                 a.action, a.t, a.ep_return, (int64_t)kLiveMask, a, LeanNoExtra{}, ntiles,
                                                 (uint64_t)0, (uint64_t)0, (uint64_t)k)
                                 : launch_kernel(step_floor_kernel<1>, (int)ntiles, kTileEnvsLean / 4, (hipStream_t)stream, a.obs,
                                                 a.action, a.t, a.ep_return, (int64_t)kLiveMask, a, LeanNoExtra{}, ntiles,
                                                 (uint64_t)0, (uint64_t)0, (uint64_t)k);
        if (rc != FISHING_OK) return rc;
    }
    return FISHING_OK;
}

}  // namespace fishing

extern "C" {

int fishing_step_floor_f32(int32_t mode, int64_t n, const FishingBuffers* b, int32_t n_launches, fishing_stream_t stream) {
    return fishing::step_floor_impl(mode, n, b, n_launches, stream);
}

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
int fishing_step_kernel_name_f32(const FishingParams* p, int64_t n, const FishingBuffers* b, char* out, int64_t len) {
    return fishing::kernel_name_impl<float>(p, n, b, out, len);
}
int fishing_step_kernel_name_f64(const FishingParams* p, int64_t n, const FishingBuffers* b, char* out, int64_t len) {
    return fishing::kernel_name_impl<double>(p, n, b, out, len);
}

}  // extern "C"
