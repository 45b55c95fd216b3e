/*
 * fishing_oracle.c -- plain-C CPU restatement of the gym_fishing hot path.
 * TEST INFRASTRUCTURE ONLY: used by tests/ (cross-checked against the NumPy oracle and the
 * golden vectors) and by bench.py's cpu_baseline leg (the "what can host cores do with
 * compiled code" figure next to the reference-equivalent Python port).  The product
 * (gym_fishing_amd) never links or loads it.
 *
 * Follows, operation for operation (paths relative to /root/reference/gym_fishing/envs/):
 *   base_fishing_env.py:60-81 step, :112-119 harvest_draw, :121-133 population_draw,
 *   :135-147 get_quota, :158-164 get_fish_population / get_state, :83-91 reset;
 *   fishing_tipping_env.py:24-35 (v2 growth); fishing_model_error.py:37-48 (v4 draws / reset).
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: every operation separately rounded).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MODEL_V0 0
#define MODEL_V1 1
#define MODEL_V2 2
#define MODEL_V4 4

/* ---- Philox4x32-10 (Salmon et al., SC'11) and the maps shared with the kernels ---- */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int rnd = 0; rnd < 10; ++rnd) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1;
        c[3] = (uint32_t)p0;
        c[0] = n0;
        c[2] = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

void oracle_philox_block(uint64_t seed, uint64_t index, uint64_t counter, uint32_t stream, uint32_t out[4]) {
    out[0] = (uint32_t)index;
    out[1] = (stream << 24) | ((uint32_t)(index >> 32) & 0xFFFFFFu);
    out[2] = (uint32_t)counter;
    out[3] = (uint32_t)(counter >> 32);
    philox4x32_10(out, (uint32_t)seed, (uint32_t)(seed >> 32));
}

/* Box-Muller in double, rounded to float: the value the device's hardware ops approximate */
static void box_muller(uint32_t w0, uint32_t w1, float* zc, float* zs) {
    const float u1 = (float)w0 * 0x1p-32f + 0x1p-33f;
    const float u2 = (float)w1 * 0x1p-32f;
    const double rad = sqrt(-2.0 * log((double)u1));
    const double ang = 6.283185307179586476925286766559 * (double)u2;
    *zc = (float)(rad * cos(ang));
    *zs = (float)(rad * sin(ang));
}

/* ---- one env, one step; T = double or float via macro expansion ---- */
#define DEFINE_STEP(NAME, T, EXPFN)                                                                      \
    static inline void NAME(int model, T obs, int32_t t, T quota, T z, T r, T K, T sigma, T C,          \
                            int32_t Tmax, T* obs_next, T* reward, uint8_t* done, int32_t* t_next) {    \
        T x = (obs + (T)1) * K;                        /* :159 */                                       \
        const T h = (quota < x) ? quota : x;           /* :117 min(x, quota) */                         \
        const T d = x - h;                                                                               \
        x = ((T)0 > d) ? (T)0 : d;                     /* :118 max(x - h, 0.0) */                       \
        T g;                                                                                             \
        if (model == MODEL_V2) {                                                                         \
            const T e = ((r * ((T)1 - (x / K))) * (x - C)) + ((x * sigma) * z);                         \
            g = x * EXPFN(e);                                                                            \
        } else {                                                                                         \
            g = (x + ((r * x) * ((T)1 - (x / K)))) + ((x * sigma) * z);                                 \
        }                                                                                                \
        x = (g > (T)0) ? g : ((g != g) ? g : (T)0);    /* np.maximum(g, 0.0) */                         \
        *obs_next = x / K - (T)1;                      /* :163 */                                       \
        *reward = ((T)0 > h) ? (T)0 : h;               /* :74 */                                        \
        *t_next = t + 1;                               /* :75 */                                        \
        *done = (uint8_t)((*t_next > Tmax) || (x <= (T)0)); /* :76-79 */                                \
    }

DEFINE_STEP(step_one_f64, double, exp)
DEFINE_STEP(step_one_f32, float, expf)

#define DEFINE_QUOTA(NAME, T)                                                          \
    static inline T NAME(int model, const void* action, int64_t i, int32_t n_actions, T K) { \
        if (model == MODEL_V0) return ((T)((const int32_t*)action)[i] / (T)n_actions) * K; \
        T a = (T)((const float*)action)[i];                                            \
        a = (a < (T)-1) ? (T)-1 : a;                                                   \
        a = (a > (T)1) ? (T)1 : a;                                                     \
        return (a + (T)1) * K;                                                         \
    }
DEFINE_QUOTA(quota_f64, double)
DEFINE_QUOTA(quota_f32, float)

/* Vector step over n envs.  r/K/sigma arrays may be NULL => the scalar is used. */
#define DEFINE_VSTEP(NAME, T, STEP1, QUOTA)                                                              \
    void NAME(int model, int64_t n, const T* obs, const int32_t* t, const void* action, const T* z,     \
              const T* r_arr, const T* K_arr, const T* sigma_arr, T r, T K, T sigma, T C, int32_t Tmax, \
              int32_t n_actions, T* obs_out, T* reward, uint8_t* done, int32_t* t_out) {                \
        for (int64_t i = 0; i < n; ++i) {                                                                \
            const T Ki = K_arr ? K_arr[i] : K;                                                           \
            STEP1(model, obs[i], t[i], QUOTA(model, action, i, n_actions, Ki), z ? z[i] : (T)0,          \
                  r_arr ? r_arr[i] : r, Ki, sigma_arr ? sigma_arr[i] : sigma, C, Tmax, &obs_out[i],      \
                  &reward[i], &done[i], &t_out[i]);                                                      \
        }                                                                                                \
    }
DEFINE_VSTEP(oracle_step_f64, double, step_one_f64, quota_f64)
DEFINE_VSTEP(oracle_step_f32, float, step_one_f32, quota_f32)

/* Noise / random-policy action of global env `env` at step `counter` (quad scheme of the kernels:
 * one block per 4 envs on stream 0 for the normals, one on stream 3 for the actions). */
void oracle_noise_f32(int64_t n, uint64_t env_offset, uint64_t seed, uint64_t counter, float* z, float* action_cts) {
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t env = env_offset + (uint64_t)i;
        const int leg = (int)(env & 3u);
        uint32_t w[4];
        if (z) {
            oracle_philox_block(seed, env >> 2, counter, 0u, w);
            float zc, zs;
            box_muller(w[leg & 2], w[(leg & 2) + 1], &zc, &zs);
            z[i] = (leg & 1) ? zs : zc;
        }
        if (action_cts) {
            oracle_philox_block(seed, env >> 2, counter, 3u, w);
            action_cts[i] = (float)w[leg] * 0x1p-31f - 1.0f;
        }
    }
}

/* CPU baseline workload: fishing-v1 (or v0/v2), float32, random policy, auto-reset, T steps over
 * n envs, OpenMP over envs.  State stays in the caller's arrays (obs, t); returns the sum of
 * rewards so the work cannot be optimised away.  Same Philox noise as the kernels. */
double oracle_rollout_random_f32(int model, int64_t n, uint64_t env_offset, int32_t T, float* obs, int32_t* t,
                                 float r, float K, float sigma, float C, float x0, int32_t Tmax,
                                 int32_t n_actions, uint64_t seed, uint64_t step_counter0, int32_t threads) {
    double total = 0.0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int64_t quad = 0; quad < (n + 3) / 4; ++quad) {
        for (int32_t s = 0; s < T; ++s) {
            uint32_t w[4], wn[4];
            oracle_philox_block(seed, (env_offset >> 2) + (uint64_t)quad, step_counter0 + (uint64_t)s, 0u, wn);
            oracle_philox_block(seed, (env_offset >> 2) + (uint64_t)quad, step_counter0 + (uint64_t)s, 3u, w);
            float zz[4];
            box_muller(wn[0], wn[1], &zz[0], &zz[1]);
            box_muller(wn[2], wn[3], &zz[2], &zz[3]);
            for (int leg = 0; leg < 4; ++leg) {
                const int64_t i = 4 * quad + leg;
                if (i >= n) break;
                float quota;
                if (model == MODEL_V0) {
                    const int32_t a = (int32_t)(((uint64_t)w[leg] * (uint64_t)(uint32_t)n_actions) >> 32);
                    quota = ((float)a / (float)n_actions) * K;
                } else {
                    float a = (float)w[leg] * 0x1p-31f - 1.0f;
                    a = (a < -1.0f) ? -1.0f : a;
                    a = (a > 1.0f) ? 1.0f : a;
                    quota = (a + 1.0f) * K;
                }
                float o2, rew;
                uint8_t dn;
                int32_t t2;
                step_one_f32(model, obs[i], t[i], quota, zz[leg], r, K, sigma, C, Tmax, &o2, &rew, &dn, &t2);
                total += (double)rew;
                if (dn) {
                    o2 = x0 / K - 1.0f;
                    t2 = 0;
                }
                obs[i] = o2;
                t[i] = t2;
            }
        }
    }
    return total;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
