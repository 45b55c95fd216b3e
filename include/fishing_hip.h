/*
 * fishing_hip.h -- C ABI of libfishing_hip.so: the MI355X (gfx950) hot path of the
 * vectorised fisheries gym.  N independent 1-D environments advance in lockstep.
 *
 * The reference (boettiger-lab/gym_fishing, pure Python) has no FFI of its own; the
 * boundary it exposes for this path is the gym.Env protocol.  Each entry point below
 * names the reference method it replaces (paths relative to the reference root).
 * The host-side mirror of that protocol lives in gym_fishing_amd/envs.py and calls
 * these functions through ctypes; INTEGRATION.md shows the binding a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  Every pointer is a DEVICE-accessible pointer
 *     owned by the caller (device memory such as torch tensors, or pinned device-mapped host
 *     memory), 16-byte aligned, contiguous.
 *   - every function returns 0 on success or a negative FISHING_ERR_* code (argument
 *     errors, nothing launched) or a positive hipError_t (launch failure).
 *   - the library keeps no global state: re-entrant, one call = kernel launches
 *     enqueued on `stream` (a hipStream_t), no host synchronisation (fishing_stream_synchronize
 *     is the one call that waits), graph-capturable.
 *   - `_f32` = fp32 fast layout (obs/reward/params float, 25 B per env-step);
 *     `_f64` = fp64 parity layout (obs/reward/params double; bit-exact to the
 *     reference's NumPy arithmetic for v0/v1/v4 given the same noise).
 *     action is float32 (int32 for fishing-v0), done uint8, t int32 in both.
 */
#ifndef FISHING_HIP_H
#define FISHING_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 9: FISHING_FLAG_RESET_COUNTER_ON_DEVICE (the reset counter as a fourth word of FishingBuffers.counter, read and bumped by
 * fishing_reset_* itself: captured resets draw fresh parameters at every replay); fishing_math_f64 keeps fn 0-1 only.
 * 8: same entry points and structs as 7; fishing-v11's per-episode model choice is drawn from one Philox2x32-10 block per env
 * quad (four 16-bit draws) instead of a Philox4x32-10 block -- a run of a fishing-v11 batch continues with other model draws
 * than under ABI 7; the float64 zoo evaluates the growth functions' algebraic form (same 2e-14 bar). */
#define FISHING_ABI_VERSION 9

typedef void* fishing_stream_t; /* hipStream_t */

/* model ids = the registered env ids (gym_fishing/envs/__init__.py:17-35) */
#define FISHING_MODEL_V0 0 /* fishing-v0: logistic, Discrete(n_actions)   envs/fishing_env.py:6-24        */
#define FISHING_MODEL_V1 1 /* fishing-v1: logistic, continuous            envs/fishing_cts_env.py:4-12    */
#define FISHING_MODEL_V2 2 /* fishing-v2: tipping point, continuous       envs/fishing_tipping_env.py:6-35*/
#define FISHING_MODEL_V4 4 /* fishing-v4: per-episode (K, r) uncertainty  envs/fishing_model_error.py:6-48*/
/* growth-model zoo (envs/growth_models.py): lognormal noise, x' = max(0, exp(mu(x) + sigma z)) */
#define FISHING_MODEL_V5 5   /* fishing-v5  Allen            growth_models.py:6-25,   allen()  :208-217        */
#define FISHING_MODEL_V6 6   /* fishing-v6  Beverton-Holt    growth_models.py:28-40,  beverton_holt() :220-226 */
#define FISHING_MODEL_V7 7   /* fishing-v7  May              growth_models.py:75-108, may()    :229-242        */
#define FISHING_MODEL_V8 8   /* fishing-v8  Myers            growth_models.py:43-70,  myers()  :247-255        */
#define FISHING_MODEL_V9 9   /* fishing-v9  Ricker           growth_models.py:111-123, ricker() :258-261       */
#define FISHING_MODEL_V10 10 /* fishing-v10 NonStationary    growth_models.py:126-154 (r += alpha every draw)  */
#define FISHING_MODEL_V11 11 /* fishing-v11 ModelUncertainty growth_models.py:157-204 (one of five per episode)*/

/* growth-function kinds = positions in the reference's default model list (growth_models.py:160) */
#define FISHING_KIND_ALLEN 0
#define FISHING_KIND_BEVERTON_HOLT 1
#define FISHING_KIND_MYERS 2
#define FISHING_KIND_MAY 3
#define FISHING_KIND_RICKER 4
#define FISHING_N_KINDS 5

/* FishingParams.flags */
#define FISHING_FLAG_AUTO_RESET 1u /* SB3-VecEnv semantics: a finished env is reset inside step() */
#define FISHING_FLAG_T_U8 4u /* compact layout: FishingBuffers.t is uint8_t[n] instead of int32_t[n]
                                (needs Tmax <= 254; a counter that would pass 255 stays at 255).
                                Saves 6 of the 25 bytes per env-step of the fp32 layout.          */
#define FISHING_FLAG_V4_DERIVED 8u /* fishing-v4 on the in-kernel streams: FishingBuffers.r / .K are not
                                read or written (may be NULL); every kernel re-derives an env's (K, r)
                                from (seed, env index, the step or reset() that began its episode), which
                                it reads off years_passed -- see v4_origin_step below -- or, for envs reset
                                one by one, off FishingBuffers.v4_stamp.  Same values as the
                                stored-array mode bit for bit.  Not with FISHING_FLAG_T_U8 (a saturating
                                counter cannot date an episode) or user-supplied parameters (the host then
                                calls fishing_v4_params_* once and continues with arrays).              */
#define FISHING_FLAG_PADDED_TILES 16u /* every in/out STATE buffer of FishingBuffers (obs, t, reward, done, done_bits,
                                r, K, sigma, terminal_obs, ep_return, model_idx, v4_stamp -- not action, not z_ext) has room for
                                ceil(n / 1024) * 1024 elements; the elements behind the n-th are scratch the library may
                                overwrite.  fishing_step_* then runs a batch that is not a multiple of 1024 envs in ONE
                                launch instead of two (3.7-4.2 us per step less), and a batch below one tile on the lean
                                kernel.  Same results for the n envs.
                                Honoured when n is a multiple of 4 (else ignored: two launches as without it).    */
#define FISHING_FLAG_RESET_COUNTER_ON_DEVICE 32u /* (ABI 9) FishingBuffers.counter is u64[4] = {step counter, v4_origin_step,
                                v4_origin_counter, reset counter}.  fishing_reset_* then draws with reset counter
                                counter[3] + reset_counter (its argument) and, behind the reset on the same stream, bumps
                                counter[3] by one; a reset of every env (mask == NULL) also moves the episode origin:
                                counter[1] = counter[0], counter[2] = the reset counter it drew with.  All of it on the
                                device: a hipGraph that captured [reset(), K steps] draws fresh fishing-v4 (K, r) /
                                fishing-v11 models at every replay, as it draws fresh step noise.  Needs counter.   */
/* diagnostic (tests, A/B timing): route step() to the general kernel even where a lean instantiation applies */
#define FISHING_FLAG_DIAG_GENERAL_KERNEL 0x80000000u

/* error codes */
#define FISHING_OK 0
#define FISHING_ERR_NULL -1      /* a required pointer is NULL                    */
#define FISHING_ERR_MODEL -2     /* unknown model id                              */
#define FISHING_ERR_ALIGN -3     /* a buffer is not 16-byte aligned               */
#define FISHING_ERR_SIZE -4      /* n < 0, T < 0, n_actions <= 0, ...             */
#define FISHING_ERR_POLICY -5    /* unknown in-kernel policy                      */
#define FISHING_ERR_NO_DEVICE -6 /* no HIP device / wrong architecture            */
#define FISHING_ERR_UNSUPPORTED -7 /* this entry point does not serve this combination of flags / streams
                                      (e.g. fishing_step_fused_* with z_ext, derived fishing-v4 parameters with
                                      a one-byte year counter or a rollout without auto-reset)          */
#define FISHING_ERR_VALUE -8     /* a parameter value outside its domain (non-finite fishing-v4 mean) */

/* Scalar parameters of one env family: the constructor kwargs of the reference
 * (envs/fishing_env.py:7-16, fishing_cts_env.py:5-7, fishing_tipping_env.py:7-16,
 * fishing_model_error.py:9-19). */
/* parameters of one growth function (the per-model dicts of growth_models.py:161-186) */
typedef struct FishingGrowthParams {
    double r, K, sigma, C, M, theta, q, b, a;
} FishingGrowthParams;

typedef struct FishingParams {
    int32_t model;     /* FISHING_MODEL_*                                               */
    int32_t n_actions; /* fishing-v0 only (default 100)                                 */
    int32_t Tmax;      /* done when years_passed > Tmax (base_fishing_env.py:76)        */
    uint32_t flags;    /* FISHING_FLAG_*                                                */
    double r, K, sigma; /* scalars; ignored where the per-env array in FishingBuffers is set */
    double C;          /* fishing-v2 tipping point                                      */
    double x0;         /* init_state                                                    */
    double r_mean, K_mean, sigma_p; /* fishing-v4 redraw at reset; must be finite (else
                                       FISHING_ERR_VALUE: the reference would produce NaN stocks) */
    int32_t launch_blocks;  /* 0 = auto; else cap on workgroups (tuning knob)           */
    int32_t launch_threads; /* 0 = auto (256); 64..256, multiple of 64                  */
    /* zoo extras (fishing-v5..v11) */
    double M, theta, q, b, a; /* May / Myers shape parameters (growth_models.py:43-108) */
    double alpha;             /* fishing-v10: r += alpha before every draw (:151)       */
    int32_t n_models;         /* fishing-v11: length of the model list (1..5)           */
    int32_t kinds[FISHING_N_KINDS]; /* fishing-v11: FISHING_KIND_* of each list entry   */
    FishingGrowthParams zoo[FISHING_N_KINDS]; /* fishing-v11: parameters per KIND       */
    /* FISHING_FLAG_V4_DERIVED: the last reset() of ALL envs happened when `v4_origin_step` step() calls had
     * been made and drew with reset counter `v4_origin_counter`; an env whose years_passed t satisfies
     * step_counter - t == v4_origin_step still runs on those parameters, any other env was auto-reset by
     * step (step_counter - t - 1) and runs on that step's redraw (fishing_model_error.py:42-43). */
    uint64_t v4_origin_step, v4_origin_counter;
} FishingParams;

/* Device buffers, all of length n unless noted.  "real" = float (_f32) or double (_f64). */
typedef struct FishingBuffers {
    void* obs;           /* real  in/out  normalised state x/K - 1 (base_fishing_env.py:162-164) */
    const void* action;  /* f32 (i32 for v0)  in; unused by fishing_rollout_*                    */
    void* reward;        /* real  out     max(harvest, 0) (base_fishing_env.py:74); nullable      */
    uint8_t* done;       /* u8    out     nullable                                                */
    uint64_t* done_bits; /* u64[ceil(n/64)] out, bit (i%64) of word i/64 = done[i]; nullable      */
    int32_t* t;          /* i32   in/out  years_passed (base_fishing_env.py:75); uint8_t[n] under
                                          FISHING_FLAG_T_U8                                        */
    void* r;             /* real  in/out  per-env growth rate; required for v4 (unless FISHING_FLAG_V4_DERIVED)
                                          and v10 (drift)                                          */
    void* K;             /* real  in/out  per-env carrying capacity; required for v4 (unless ..._V4_DERIVED) */
    const void* sigma;   /* real  in      per-env noise scale; nullable => FishingParams.sigma    */
    const void* z_ext;   /* real  in      externally supplied standard normals; nullable =>
                                          in-kernel Philox4x32-10 + Box-Muller                    */
    void* terminal_obs;  /* real  out     obs before the auto-reset (SB3 terminal_observation); nullable */
    void* ep_return;     /* real  in/out  running episodic return; nullable                       */
    double* return_partials; /* f64[fishing_partials_len()], or f64[4 * fishing_partials_slots(n)] for batches of at
                                most n envs; in/out: per-workgroup partial sums of
                                {sum R, sum R^2, n_episodes, sum length} over finished episodes;
                                needs ep_return; nullable                                         */
    int32_t* model_idx;  /* i32   in/out  fishing-v11: FISHING_KIND_* in force per env (redrawn
                                          at reset, growth_models.py:187,200); else nullable.  An
                                          index outside [0, FISHING_N_KINDS) steps as Beverton-Holt
                                          and may be written back as FISHING_KIND_BEVERTON_HOLT     */
    const uint64_t* counter; /* u64[1] in  device-resident step counter, nullable.  When set, the
                                noise of fishing_step_* / fishing_rollout_* is keyed by
                                *counter + step_counter instead of step_counter alone, so a
                                hipGraph that captured the launch draws fresh noise on every
                                replay; advance it with fishing_counter_add (also capturable).
                                With FISHING_FLAG_V4_DERIVED the buffer is u64[3] = {step counter,
                                v4_origin_step, v4_origin_counter}: the kernels take the episode origin
                                from counter[1..2] and ignore FishingParams.v4_origin_* -- a captured
                                launch (frozen arguments) keeps deriving the right (K, r) after a
                                reset() of all envs, which only rewrites those two words (ABI 4).
                                With FISHING_FLAG_RESET_COUNTER_ON_DEVICE: u64[4], see the flag (fishing_reset_*
                                writes words 1-3 there: the buffer is in/out for that call).  */
    int32_t* v4_stamp;   /* i32   in/out  FISHING_FLAG_V4_DERIVED only, nullable (ABI 6).  Per-env episode origin for envs
                                that were reset ONE BY ONE: stamp[i] != 0 says the episode running in env i began with the
                                masked fishing_reset_* whose reset_counter was stamp[i] - 1, and its (K, r) are that
                                reset's draw for env i (fishing_model_error.py:41-43); stamp[i] == 0: the episode is dated by
                                years_passed as described at v4_origin_step.  fishing_reset_* with a mask writes the
                                stamps of the masked envs (and needs the buffer: FISHING_ERR_UNSUPPORTED without), a
                                reset of every env clears them all; step() / rollout() clear an env's stamp when they
                                auto-reset it.  R 4 + W 4 bytes per env-step on top of the derived mode's 37 -- 45 against
                                the 53 of stored r / K arrays.  Any other model or mode: must be NULL.  A state stream like the
                                others: under FISHING_FLAG_PADDED_TILES it needs ceil(n / 1024) * 1024 elements (the lean
                                kernels read and write it in whole 1024-env tiles).  */
} FishingBuffers;

/* In-kernel policies for the fused rollout (callers of step(): shared_env.py:29-54,
 * models/policies.py:4-31, examples/const_escapement.py:19-26). */
#define FISHING_POLICY_RANDOM 0     /* a ~ U[-1,1] (v0: uniform int in [0, n_actions)) from Philox word 2 */
#define FISHING_POLICY_CONSTANT 1   /* a = policy_param (v0: (int)policy_param)                           */
#define FISHING_POLICY_ESCAPEMENT 2 /* quota = max(x - policy_param, 0), action = get_action(quota)      */
#define FISHING_POLICY_MSY 3        /* quota = policy_param (constant), action = get_action(quota)       */

int fishing_abi_version(void);
const char* fishing_error_string(int code);
/* number of doubles a return_partials buffer must hold (zero-initialised by the caller): 4 per slot, one slot per
 * workgroup of the widest launch -- 65536 slots = 2 MiB since ABI 5 (a workgroup per 1024-env tile up to N = 2^26) */
int64_t fishing_partials_len(void);
/* how many of those slots launches over a batch of n_envs can touch (a multiple of 4096, at most
 * fishing_partials_len() / 4): what fishing_reduce_returns_slots needs to read, and -- for a caller that only ever
 * passes batches of at most n_envs -- all a return_partials buffer needs to hold (4 doubles per slot) (ABI 5) */
int64_t fishing_partials_slots(int64_t n_envs);

/* BaseFishingEnv.step (envs/base_fishing_env.py:60-81) for envs [0, n) of this shard;
 * env i is global env (env_offset + i) for the noise stream.  step_counter = number of
 * step() calls made so far on this vec-env (keys the noise with the env index). */
int fishing_step_f32(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                     uint64_t seed, uint64_t step_counter, fishing_stream_t stream);
int fishing_step_f64(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                     uint64_t seed, uint64_t step_counter, fishing_stream_t stream);

/* n_steps consecutive step() calls enqueued from C; step k reads its actions at
 * action + (k % ring_len) * action_stride elements.  Same results as n_steps calls. */
int fishing_step_many_f32(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                          int64_t action_stride, int32_t ring_len, int32_t n_steps, uint64_t seed,
                          uint64_t step_counter, fishing_stream_t stream);
int fishing_step_many_f64(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                          int64_t action_stride, int32_t ring_len, int32_t n_steps, uint64_t seed,
                          uint64_t step_counter, fishing_stream_t stream);

/* BaseFishingEnv.reset (envs/base_fishing_env.py:83-91) / FishingModelError.reset
 * (envs/fishing_model_error.py:41-48).  mask: u8[n], nullable => reset every env.
 * Writes obs, t = 0, ep_return = 0 (and K, r for v4; under FISHING_FLAG_V4_DERIVED nothing more -- a mask then stamps
 * the masked envs' episode origin into FishingBuffers.v4_stamp and needs that buffer: FISHING_ERR_UNSUPPORTED without). */
int fishing_reset_f32(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                      const uint8_t* mask, uint64_t seed, uint64_t reset_counter, fishing_stream_t stream);
int fishing_reset_f64(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                      const uint8_t* mask, uint64_t seed, uint64_t reset_counter, fishing_stream_t stream);

/* Fused T-step rollout with an in-kernel policy (state stays in registers; no action
 * traffic).  Equivalent to T fishing_step_* calls with auto-reset and the policy's
 * actions.  Updates obs, t, (r, K), ep_return, return_partials; writes the LAST step's
 * reward/done if those pointers are set.  Without FISHING_FLAG_AUTO_RESET a finished env is
 * frozen (its episode is over, as in simulate_mdp's `break`, shared_env.py:51-52) -- its year counter stops, so this
 * form is not available under FISHING_FLAG_V4_DERIVED (FISHING_ERR_UNSUPPORTED: call fishing_v4_params_* and pass arrays).
 * traj (nullable): real[T][4][n] recording per step {obs before acting, action, reward,
 * done} -- the raw material of the simulate_mdp table (shared_env.py:37-49); needs n % 4 == 0. */
int fishing_rollout_f32(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                        int32_t policy, double policy_param, int32_t T, void* traj, uint64_t seed,
                        uint64_t step_counter, fishing_stream_t stream);
int fishing_rollout_f64(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                        int32_t policy, double policy_param, int32_t T, void* traj, uint64_t seed,
                        uint64_t step_counter, fishing_stream_t stream);

/* The same rollout with one policy parameter PER ENV (ABI 7): policy_params real[n], 16-byte aligned -- env i escapes to
 * S = policy_params[i] (FISHING_POLICY_ESCAPEMENT), fishes the quota policy_params[i] (FISHING_POLICY_MSY) or takes the
 * action policy_params[i] (FISHING_POLICY_CONSTANT).  What N reference envs do when each builds its own escapement(env) /
 * msy(env): BMSY() (models/policies.py:51-67) runs under the (K, r) that env drew (fishing-v4) or the growth function in
 * force there (fishing-v11), so S differs per env.  FISHING_ERR_POLICY for FISHING_POLICY_RANDOM (no parameter),
 * FISHING_ERR_UNSUPPORTED without FISHING_FLAG_AUTO_RESET.  Same results as fishing_rollout_* where all parameters are equal. */
int fishing_rollout_params_f32(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                               int32_t policy, const void* policy_params, int32_t T, void* traj, uint64_t seed,
                               uint64_t step_counter, fishing_stream_t stream);
int fishing_rollout_params_f64(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                               int32_t policy, const void* policy_params, int32_t T, void* traj, uint64_t seed,
                               uint64_t step_counter, fishing_stream_t stream);

/* n_steps consecutive step() calls in ONE launch: obs, t, (r, K,) ep_return stay in registers, step k
 * reads its actions at action + (k % ring_len) * action_stride elements (the caller's [R, n] ring) and
 * -- optionally -- writes its reward / done rows at reward_steps + k * out_stride (real[n_steps][out_stride])
 * and done_steps + k * out_stride (u8).  The launch-bound regime's fast path (n <= 2^20: a dependent
 * launch costs ~2.7 us whatever it moves).  Bit-identical to n_steps fishing_step_* calls with the same
 * step_counter: same Philox counters, same arithmetic; FishingBuffers.reward / .done receive the last
 * step's values, return_partials the same record.  Every model id (fishing-v11 needs model_idx, as step()).
 * Not for z_ext / terminal_obs / done_bits (FISHING_ERR_UNSUPPORTED: use step()). */
int fishing_step_fused_f32(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                           int64_t action_stride, int32_t ring_len, int32_t n_steps, void* reward_steps,
                           uint8_t* done_steps, int64_t out_stride, uint64_t seed, uint64_t step_counter,
                           fishing_stream_t stream);
int fishing_step_fused_f64(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                           int64_t action_stride, int32_t ring_len, int32_t n_steps, void* reward_steps,
                           uint8_t* done_steps, int64_t out_stride, uint64_t seed, uint64_t step_counter,
                           fishing_stream_t stream);

/* fishing-v4 under FISHING_FLAG_V4_DERIVED: materialise the (K, r) in force for envs [0, n) given their
 * years_passed `t` (and, where masked resets happened, their origin stamps `stamp`: i32[n], nullable -- see
 * FishingBuffers.v4_stamp; ABI 6) at step count `step_counter` (what env.K / env.r show, and what the host stores when it
 * leaves the derived mode).  K_out / r_out: real[n], either may be NULL. */
int fishing_v4_params_f32(const FishingParams* p, int64_t n, int64_t env_offset, const int32_t* t, const int32_t* stamp,
                          void* K_out, void* r_out, uint64_t seed, uint64_t step_counter, fishing_stream_t stream);
int fishing_v4_params_f64(const FishingParams* p, int64_t n, int64_t env_offset, const int32_t* t, const int32_t* stamp,
                          void* K_out, void* r_out, uint64_t seed, uint64_t step_counter, fishing_stream_t stream);

/* Diagnostic: the demangled name (as rocprofv3 prints it, without the argument list) of the kernel that
 * fishing_step_f32/_f64 would launch for the whole tiles of this request -- the same dispatch code decides,
 * nothing is launched.  Writes at most len - 1 characters + NUL; returns 0 or a FISHING_ERR_*. */
int fishing_step_kernel_name_f32(const FishingParams* p, int64_t n, const FishingBuffers* b, char* out, int64_t len);
int fishing_step_kernel_name_f64(const FishingParams* p, int64_t n, const FishingBuffers* b, char* out, int64_t len);

/* Diagnostic (ABI 9): the floor a fishing_step_f32 launch over n envs stands on -- n_launches back-to-back launches of a kernel
 * with that launch's grid (n / 1024 workgroups of 256 threads), argument list and kernarg preload, and
 *   mode 0: an empty body (what launching the grid costs, whatever it moves);
 *   mode 1: a copy over the step's streams in the step's access shape (R obs, action, t, ep_return; W obs, reward, done, t,
 *           ep_return: 33 B per env, no arithmetic but an add) -- CLOBBERS reward, done and ep_return: scratch buffers only.
 * n a positive multiple of 1024 (at most 2^26); mode 1 needs obs, action, reward, done, t, ep_return.  bench.py times both
 * beside the step kernel at the launch-bound shard sizes (configs.*.latency_floor_us / copy_floor_us). */
int fishing_step_floor_f32(int32_t mode, int64_t n, const FishingBuffers* b, int32_t n_launches, fishing_stream_t stream);

/* *counter += delta on `stream` (one thread).  Pair with FishingBuffers.counter to make a
 * captured step() / rollout() replayable: capture {step, counter_add(1)} once, replay K times. */
int fishing_counter_add(uint64_t* counter, uint64_t delta, fishing_stream_t stream);

/* Sum the per-workgroup partials in slot order (deterministic) into out4 =
 * {sum R, sum R^2, n_episodes, sum length}.  out4 is then all-reduced across GPUs by
 * the host (RCCL). */
int fishing_reduce_returns(const double* return_partials, double* out4, fishing_stream_t stream);
/* The same over the first `slots` slots only (slots = fishing_partials_slots(n_envs) of the batch that filled them:
 * one 4096-slot pass, ~3 us, for every batch up to N = 2^22 instead of sixteen).  FISHING_ERR_SIZE unless slots is a
 * multiple of 4096 in [4096, fishing_partials_len() / 4].  Same bits as the full reduction while the slots beyond
 * are zero. (ABI 5) */
int fishing_reduce_returns_slots(const double* return_partials, int64_t slots, double* out4, fishing_stream_t stream);

/* population_draw() (envs/base_fishing_env.py:121-133; v2: envs/fishing_tipping_env.py:24-35; the zoo's:
 * envs/growth_models.py:208-261) over an array of populations with the scalar r, K, sigma, C, ... of `p`:
 * x_out[i] = growth of x_in[i] under noise z[i] (z nullable => 0).  This is how the reference's BMSY() sweeps the
 * growth curve (models/policies.py:59-63).
 * model_idx (ABI 6): i32[n], fishing-v11 only and required there (FISHING_ERR_NULL without, FISHING_ERR_UNSUPPORTED
 * with any other model): element i grows under growth function model_idx[i] (FISHING_KIND_*) with that function's
 * parameter set p->zoo[model_idx[i]] -- ModelUncertainty.population_draw (envs/growth_models.py:190-194: "the model
 * in force, with ITS params") for N envs, or for one sweep per growth function in a single launch.
 * r, K (ABI 6): real[n], nullable, fishing-v0/v1/v2/v4 only (FISHING_ERR_UNSUPPORTED with the zoo): element i grows under
 * r[i] / K[i] instead of p->r / p->K -- N fishing-v4 envs, each under the pair it drew (what models/policies.py:7-13
 * evaluates per env for msy). */
int fishing_population_draw_f32(const FishingParams* p, int64_t n, const void* x_in, const void* z, const int32_t* model_idx,
                                const void* r, const void* K, void* x_out, fishing_stream_t stream);
int fishing_population_draw_f64(const FishingParams* p, int64_t n, const void* x_in, const void* z, const int32_t* model_idx,
                                const void* r, const void* K, void* x_out, fishing_stream_t stream);

/* BMSY() (models/policies.py:51-67) for n_envs envs that each carry their own (K, r) -- what N reference fishing-v4 envs
 * return, one call each: env i sweeps the n_states observations `states` (the observation Box's linspace) through one
 * noise-free population_draw under K[i], r[i] (nullable => p->K / p->r) and S_out[i] = the population
 * (states[j] + 1) * K[i] with the largest growth population_draw(x0) - x0 (np.argmax: the first maximum; NaN counts as
 * one).  fishing-v0/v1/v2/v4 (FISHING_ERR_MODEL otherwise).  (ABI 6) */
int fishing_bmsy_sweep_f32(const FishingParams* p, int64_t n_envs, const void* K, const void* r, const void* states,
                           int64_t n_states, void* S_out, fishing_stream_t stream);
int fishing_bmsy_sweep_f64(const FishingParams* p, int64_t n_envs, const void* K, const void* r, const void* states,
                           int64_t n_states, void* S_out, fishing_stream_t stream);

/* hipStreamSynchronize(stream): lets a host binding without a HIP runtime binding of its own (ctypes)
 * wait for the launches it enqueued -- the scalar gym.Env protocol reads its one env's results from
 * pinned, device-mapped memory right after.  Returns 0 or the hipError_t. */
int fishing_stream_synchronize(fishing_stream_t stream);

/* Test/diagnostic: the generator itself.  For Philox index (env_offset + i): words[4*i..4*i+3] =
 * the Philox4x32-10 block on stream `stream_tag`, z0 / z1 = the cos / sin legs of the Box-Muller
 * pair of words (0, 1).  What the index means per stream: tags 0 (step noise) and 3 (random-policy
 * actions of the fused rollout) index by env QUAD (global env >> 2), see fishing_step_normals_f32;
 * the reset streams (tags 1 / 2) draw from Philox2x32-10 blocks instead: fishing-v4's (K, r) per env, see
 * fishing_reset_normals_f32; fishing-v11's model choice per env quad (four 16-bit draws from one block).
 * Any output pointer may be NULL. */
int fishing_noise_f32(int64_t n, int64_t env_offset, uint64_t seed, uint64_t counter, int32_t stream_tag,
                      uint32_t* words, float* z0, float* z1, fishing_stream_t stream);

/* Test/diagnostic: the process noise z the step and rollout kernels draw for envs env_offset ..
 * env_offset + n - 1 at step `counter` (np.random.normal(0, 1) of base_fishing_env.py:130).  One
 * Philox block per env quad on stream 0: Box-Muller of words (0, 1) -> z of envs 4q, 4q + 1, of
 * words (2, 3) -> z of envs 4q + 2, 4q + 3. */
int fishing_step_normals_f32(int64_t n, int64_t env_offset, uint64_t seed, uint64_t counter, float* z,
                             fishing_stream_t stream);

/* Test/diagnostic: the standard normals behind fishing-v4's (K, r) redraw (fishing_model_error.py:42-43)
 * for envs env_offset .. env_offset + n - 1 on reset stream `stream_tag` (1 = auto-reset inside step,
 * counter = that step's counter; 2 = reset(), counter = the reset counter).  One Philox2x32-10 block per
 * env, counter words {env[31:0], counter[31:0]}, key folded from seed / stream / the high halves:
 * Box-Muller of (w0, w1) -> (zK, zr). */
int fishing_reset_normals_f32(int64_t n, int64_t env_offset, uint64_t seed, uint64_t counter, int32_t stream_tag,
                              float* zK, float* zr, fishing_stream_t stream);

/* Test/diagnostic: the elementary functions the float64 zoo's growth functions are built on, applied elementwise --
 * out[i] = fn(in[i]): the < 1-ulp log / exp of csrc/fishing_common.h (msun forms), held to <= 1 ulp of libm by
 * tests/test_gpu_zoo.py.  Any other fn: FISHING_ERR_SIZE.  (ABI 6-8 also exposed the polynomial forms of builds that no
 * longer exist, fn 2-4.) */
#define FISHING_MATH_LOG_F64 0
#define FISHING_MATH_EXP_F64 1
int fishing_math_f64(int64_t n, int32_t fn, const double* in, double* out, fishing_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FISHING_HIP_H */
