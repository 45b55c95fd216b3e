// fishing_aux.hip -- everything of libfishing_hip.so that is not on the per-step hot path (gfx950):
// reset(), the return-record reduction, population_draw() sweeps, fishing-v4's parameter materialisation,
// the graph-replay counter and the generator's test hooks.
#include "fishing_common.h"
#include "fishing_host.h"

#include <algorithm>

namespace fishing {

// reset(): one env per thread (not on the hot path; runs once per rollout).
template <typename T, int MODEL>
__global__ void __launch_bounds__(256)
reset_kernel(const ParamsT<T> p, const BuffersT<T> b, const int64_t n, const uint64_t env_offset,
             const uint8_t* __restrict__ mask, const uint64_t seed, const uint64_t reset_counter_arg) {
    // FISHING_FLAG_RESET_COUNTER_ON_DEVICE: the reset counter lives in counter[3] (wave-uniform scalar load), so that a launch
    // captured in a hipGraph draws with a fresh counter at every replay; reset_counters_kernel bumps it behind this kernel
    const uint64_t reset_counter =
        (p.flags & FISHING_FLAG_RESET_COUNTER_ON_DEVICE) ? b.counter[3] + reset_counter_arg : reset_counter_arg;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        if (mask && !mask[i]) continue;
        T K = p.K;
        if (is_zoo_tag(MODEL) && p.model == FISHING_MODEL_V11) {
            const uint64_t env = env_offset + (uint64_t)i;      // quad scheme of redraw_kinds: half (env & 3) of the quad's block
            uint32_t w0, w1;
            model_block(seed, env >> 2, reset_counter, true, w0, w1);
            const uint32_t leg = (uint32_t)(env & 3);
            const uint32_t half = leg == 0 ? (w0 & 0xFFFFu) : leg == 1 ? (w0 >> 16) : leg == 2 ? (w1 & 0xFFFFu) : (w1 >> 16);
            b.model_idx[i] = p.kinds[model_index_from_half(half, p.n_models)];
        }
        if (MODEL == FISHING_MODEL_V4 && b.K && b.r) {      // derived mode keeps no arrays: nothing to draw here
            T r;
            draw_model_error<T>(seed, env_offset + (uint64_t)i, reset_counter, kStreamReset, p.K_mean,
                                p.r_mean, p.sigma_p, K, r);
            b.K[i] = K;
            b.r[i] = r;
        }
        // ... but an env reset on its own is stamped with this reset's counter (its episode's origin); a reset of every
        // env clears the stamps: the origin words date every episode again
        if (MODEL == FISHING_MODEL_V4 && b.stamp) b.stamp[i] = mask ? (int32_t)(uint32_t)(reset_counter + 1) : 0;
        b.obs[i] = reset_obs<T, MODEL>(p.x0, K);
        if (p.flags & FISHING_FLAG_T_U8) reinterpret_cast<uint8_t*>(b.t)[i] = 0;
        else b.t[i] = 0;
        if (b.ep_return) b.ep_return[i] = (T)0;
    }
}

// FISHING_FLAG_RESET_COUNTER_ON_DEVICE, behind reset_kernel on the same stream (one thread): a reset of every env moves the
// episode origin to (step counter, the reset counter it drew with); every reset bumps the reset counter.
__global__ void reset_counters_kernel(uint64_t* counter, const uint64_t reset_counter_arg, const int full) {
    const uint64_t drew_with = counter[3] + reset_counter_arg;
    if (full) {
        counter[1] = counter[0];
        counter[2] = drew_with;
    }
    counter[3] += 1;
}

// fishing-v4, derived parameters: the (K, r) in force for each env, from its year counter
template <typename T>
__global__ void __launch_bounds__(256)
v4_params_kernel(const ParamsT<T> p, const int64_t n, const uint64_t env_offset, const int32_t* __restrict__ t,
                 const int32_t* __restrict__ stamp, T* __restrict__ K_out, T* __restrict__ r_out, const uint64_t seed,
                 const uint64_t step_counter) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        T K, r;
        derive_model_error<T>(seed, env_offset + (uint64_t)i, step_counter, t[i], p.origin_step, p.origin_counter,
                              p.K_mean, p.r_mean, p.sigma_p, K, r, stamp ? stamp[i] : 0);
        if (K_out) K_out[i] = K;
        if (r_out) r_out[i] = r;
    }
}

constexpr int kReduceThreads = 1024;
constexpr int kReduceSlots = 4 * kReduceThreads;     // slots per pass
__global__ void __launch_bounds__(kReduceThreads)
reduce_returns_kernel(const double* __restrict__ partials, const int passes, double* __restrict__ out4) {
    // One workgroup of 16 waves, one pass per kReduceSlots = 4096 slots.  In a pass thread i owns slots i, i + 1024,
    // i + 2048, i + 3072: four 32-byte reads (a slot's four fields are contiguous), all issued before the first add;
    // then a fixed tree -- slot order inside the thread, a shuffle tree inside the wave, wave order across the
    // workgroup: same bits on every run.  (256 threads with 64 strided 8-byte reads each took 8 us
    // of the bench's 20-step region; this one ~3 for one pass = every batch up to N = 2^22.)
    static_assert(kPartialSlots % kReduceSlots == 0, "whole passes");
    typedef double d2 __attribute__((ext_vector_type(2)));
    double s[kPartialFields] = {0.0, 0.0, 0.0, 0.0};
    for (int pass = 0; pass < passes; ++pass) {
        constexpr int kPerThread = 4;
        d2 lo[kPerThread], hi[kPerThread];
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            const d2* q = reinterpret_cast<const d2*>(partials + (int64_t)(threadIdx.x + (pass * kPerThread + k) * kReduceThreads) * kPartialFields);
            lo[k] = q[0];
            hi[k] = q[1];
        }
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            s[0] += lo[k][0];
            s[1] += lo[k][1];
            s[2] += hi[k][0];
            s[3] += hi[k][1];
        }
    }
    __shared__ double red[kReduceThreads / kWave][kPartialFields];
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int f = 0; f < kPartialFields; ++f) {
        const double w = wave_sum(s[f]);
        if (lane == 0) red[threadIdx.x >> 6][f] = w;
    }
    __syncthreads();
    if (threadIdx.x < kPartialFields) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < kReduceThreads / kWave; ++w) tot += red[w][threadIdx.x];
        out4[threadIdx.x] = tot;
    }
}

// A population handed in from OUTSIDE may be negative (the step kernels' never is: harvest_draw leaves max(x - h, 0)).  The
// reference's allen / myers / ricker take log(x) of it -> NaN (growth_models.py:208-261; May's log(exp_mu) and
// Beverton-Holt's clip are handled where they are computed); the float32 layout's algebraic forms carry x itself as the
// prefactor and would return 0 after the max(0, .): restore the NaN here, outside the step kernels' instruction stream.
template <typename T>
__device__ __forceinline__ T zoo_draw_outside(const int kind, const T x, const T out) {
    const bool takes_log_x = kind == FISHING_KIND_ALLEN || kind == FISHING_KIND_MYERS || kind == FISHING_KIND_RICKER;
    return (takes_log_x && x < (T)0) ? (T)__builtin_nan("") : out;
}

// population_draw() over an array of populations, as BMSY() drives it (models/policies.py:59-63)
template <typename T, int MODEL>
__global__ void __launch_bounds__(256)
population_draw_kernel(const ParamsT<T> p, const int kind, const int64_t n, const T* __restrict__ x_in,
                       const T* __restrict__ z, const T* __restrict__ r_arr, const T* __restrict__ K_arr, T* __restrict__ x_out) {
    const GrowthT<T> P = p.growth;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        if constexpr (is_zoo_tag(MODEL))
            x_out[i] = zoo_draw_outside<T>(kind, x_in[i], zoo_population_draw<T>(kind, x_in[i], z ? z[i] : (T)0, P));
        else        // (r_arr / K_arr: element i under ITS parameters -- N fishing-v4 envs, each with the pair it drew)
            x_out[i] = population_draw<T, MODEL>(x_in[i], z ? z[i] : (T)0, r_arr ? r_arr[i] : p.r, K_arr ? K_arr[i] : p.K, p.sigma, p.C);
    }
}

// BMSY() for N envs that each carry their own (K, r) (fishing-v4; models/policies.py:51-67 run once per env): env i sweeps the
// observation grid `states` through one noise-free population_draw under ITS parameters and keeps the population with the
// largest growth -- x0 = (state + 1) * K_i (get_fish_population :158-160), growth = population_draw(x0) - x0, S_i = x0 at
// np.argmax (the first maximum; a NaN counts as the maximum, as in NumPy).  One env per thread, the grid point is
// wave-uniform: n_envs * n_states growth evaluations, ~10 ms for 2^21 envs x 10001 states.
template <typename T, int MODEL>
__global__ void __launch_bounds__(256)
bmsy_sweep_kernel(const ParamsT<T> p, const int64_t n_envs, const T* __restrict__ K_arr, const T* __restrict__ r_arr,
                  const T* __restrict__ states, const int64_t n_states, T* __restrict__ S_out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_envs;
         i += (int64_t)gridDim.x * blockDim.x) {
        const T K = K_arr ? K_arr[i] : p.K, r = r_arr ? r_arr[i] : p.r;
        T best = (T)0, S = (T)0;
        bool have = false, best_nan = false;
        for (int64_t j = 0; j < n_states; ++j) {
            const T x0 = (states[j] + (T)1) * K;
            const T g = population_draw<T, MODEL>(x0, (T)0, r, K, (T)0, p.C) - x0;
            const bool g_nan = g != g;
            const bool take = !have || (!best_nan && (g_nan || g > best));
            best = take ? g : best;
            S = take ? x0 : S;
            best_nan = take ? g_nan : best_nan;
            have = true;
        }
        S_out[i] = S;
    }
}

// ... and under fishing-v11 (ModelUncertainty.population_draw, growth_models.py:190-194): element i grows under the growth
// function model_idx[i] with THAT function's parameter set.  Five wave-uniform passes, each over the lanes of its kind
// (the parameter sets stay scalar operands); a kind outside [0, 5) counts as Beverton-Holt, as in the step kernels.
template <typename T>
__global__ void __launch_bounds__(256)
population_draw_mixed_kernel(const ParamsT<T> p, const int64_t n, const T* __restrict__ x_in, const T* __restrict__ z,
                             const int32_t* __restrict__ model_idx, T* __restrict__ x_out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int kind = model_idx[i];
        kind = (kind >= 0 && kind < FISHING_N_KINDS) ? kind : FISHING_KIND_BEVERTON_HOLT;
        const T x = x_in[i], zi = z ? z[i] : (T)0;
        T out = (T)0;
        if (kind == FISHING_KIND_ALLEN) out = zoo_population_draw<T, FISHING_KIND_ALLEN>(kind, x, zi, p.zoo[FISHING_KIND_ALLEN]);
        if (kind == FISHING_KIND_BEVERTON_HOLT) out = zoo_population_draw<T, FISHING_KIND_BEVERTON_HOLT>(kind, x, zi, p.zoo[FISHING_KIND_BEVERTON_HOLT]);
        if (kind == FISHING_KIND_MYERS) out = zoo_population_draw<T, FISHING_KIND_MYERS>(kind, x, zi, p.zoo[FISHING_KIND_MYERS]);
        if (kind == FISHING_KIND_MAY) out = zoo_population_draw<T, FISHING_KIND_MAY>(kind, x, zi, p.zoo[FISHING_KIND_MAY]);
        if (kind == FISHING_KIND_RICKER) out = zoo_population_draw<T, FISHING_KIND_RICKER>(kind, x, zi, p.zoo[FISHING_KIND_RICKER]);
        x_out[i] = zoo_draw_outside<T>(kind, x, out);
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

// test hook: the (zK, zr) normals of the fishing-v4 draw (draw_model_error: one Philox2x32-10 block per env)
__global__ void __launch_bounds__(256)
reset_normals_kernel(const int64_t n, const uint64_t env_offset, const uint64_t seed, const uint64_t counter,
                     const uint32_t stream_tag, float* __restrict__ zK, float* __restrict__ zr) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t env = env_offset + (uint64_t)i;
        uint32_t w0, w1;
        param_block(seed, env, counter, stream_tag == kStreamReset, w0, w1);
        float a, c;
        box_muller(w0, w1, a, c);
        if (zK) zK[i] = a;
        if (zr) zr[i] = c;
    }
}

// test hook: the library's own float64 / float32 elementary functions, one value per thread
__global__ void __launch_bounds__(256)
math_kernel(const int64_t n, const int fn, const double* __restrict__ in, double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const double v = in[i];
        const double r = (fn == FISHING_MATH_LOG_F64) ? log_f64(v) : exp_f64(v);          // wave-uniform
        out[i] = r;
    }
}

static inline int grid_for(int64_t n, int cap) {
    const int64_t nb = (n + 255) / 256;
    return (int)std::max<int64_t>(1, std::min<int64_t>(nb, cap));
}

template <typename T>
int reset_impl(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
               const uint8_t* mask, uint64_t seed, uint64_t reset_counter, fishing_stream_t stream) {
    const int rc = check_common(p, n, env_offset, b);
    if (rc != FISHING_OK) return rc;
    if (n == 0) return FISHING_OK;
    const bool device_counter = (p->flags & FISHING_FLAG_RESET_COUNTER_ON_DEVICE) != 0;
    if (device_counter && !b->counter) return FISHING_ERR_NULL;
    const ParamsT<T> pt = narrow_params<T>(*p);
    BuffersT<T> bt = typed_buffers<T>(*b);
    if (p->model == FISHING_MODEL_V4 && (p->flags & FISHING_FLAG_V4_DERIVED)) {
        // envs reset one by one no longer share the origin the derivation dates episodes from: each gets its own, in
        // FishingBuffers.v4_stamp (a caller without that buffer materialises the parameters -- fishing_v4_params_* -- and
        // continues with arrays)
        if (mask && !b->v4_stamp) return FISHING_ERR_UNSUPPORTED;
        if (mask && reset_counter >= 0x7FFFFFFFull) return FISHING_ERR_SIZE;       // (the stamp is 31 bits of reset counter + 1)
        bt.K = bt.r = nullptr;
    }
    const int blocks = grid_for(n, 2048);
    hipStream_t s = (hipStream_t)stream;
    const int rc2 = with_model_tag(p->model, [&](auto tag) {
        // reset only distinguishes v4 (parameter redraw, un-normalised obs) and v11 (model draw)
        constexpr int kTag = decltype(tag)::value;
        constexpr int kResetTag = (kTag == FISHING_MODEL_V4) ? FISHING_MODEL_V4
                                  : is_zoo_tag(kTag)         ? kModelZooMixed
                                                             : FISHING_MODEL_V1;
        return launch_kernel(reset_kernel<T, kResetTag>, blocks, 256, s, pt, bt, n, (uint64_t)env_offset, mask, seed,
                             reset_counter);
    });
    if (rc2 != FISHING_OK || !device_counter) return rc2;
    return launch_kernel(reset_counters_kernel, 1, 1, s, const_cast<uint64_t*>(b->counter), reset_counter, mask ? 0 : 1);
}

template <typename T>
int v4_params_impl(const FishingParams* p, int64_t n, int64_t env_offset, const int32_t* t, const int32_t* stamp, void* K_out,
                   void* r_out, uint64_t seed, uint64_t step_counter, fishing_stream_t stream) {
    if (!p || !t) return FISHING_ERR_NULL;
    if (p->model != FISHING_MODEL_V4) return FISHING_ERR_MODEL;
    if (n < 0 || env_offset < 0) return FISHING_ERR_SIZE;
    if (!(std::isfinite(p->K_mean) && std::isfinite(p->r_mean) && std::isfinite(p->sigma_p))) return FISHING_ERR_VALUE;
    if (n == 0) return FISHING_OK;
    const ParamsT<T> pt = narrow_params<T>(*p);
    return launch_kernel(v4_params_kernel<T>, grid_for(n, 2048), 256, (hipStream_t)stream, pt, n, (uint64_t)env_offset, t,
                         stamp, (T*)K_out, (T*)r_out, seed, step_counter);
}

template <typename T>
int population_draw_impl(const FishingParams* p, int64_t n, const void* x_in, const void* z, const int32_t* model_idx,
                         const void* r_arr, const void* K_arr, void* x_out, fishing_stream_t stream) {
    if (!p || !x_in || !x_out) return FISHING_ERR_NULL;
    if (n < 0) return FISHING_ERR_SIZE;
    if (!is_core_model(p->model) && !is_zoo_model(p->model)) return FISHING_ERR_MODEL;
    // per-element (r, K) are the logistic / tipping models' (the zoo's functions read their own parameter sets)
    if ((r_arr || K_arr) && !is_core_model(p->model)) return FISHING_ERR_UNSUPPORTED;
    // the growth function per element is fishing-v11's: there it is required, anywhere else there is nothing to select
    if (p->model == FISHING_MODEL_V11 && !model_idx) return FISHING_ERR_NULL;
    if (p->model != FISHING_MODEL_V11 && model_idx) return FISHING_ERR_UNSUPPORTED;
    if (n == 0) return FISHING_OK;
    const ParamsT<T> pt = narrow_params<T>(*p);
    const int blocks = grid_for(n, 2048);
    hipStream_t s = (hipStream_t)stream;
    if (p->model == FISHING_MODEL_V11)
        return launch_kernel(population_draw_mixed_kernel<T>, blocks, 256, s, pt, n, (const T*)x_in, (const T*)z, model_idx,
                             (T*)x_out);
    if (is_zoo_model(p->model) && p->model != FISHING_MODEL_V11)
        return launch_kernel(population_draw_kernel<T, kModelZoo>, blocks, 256, s, pt, kind_of_model(p->model), n,
                             (const T*)x_in, (const T*)z, (const T*)nullptr, (const T*)nullptr, (T*)x_out);
    if (p->model == FISHING_MODEL_V2)
        return launch_kernel(population_draw_kernel<T, FISHING_MODEL_V2>, blocks, 256, s, pt, 0, n, (const T*)x_in,
                             (const T*)z, (const T*)r_arr, (const T*)K_arr, (T*)x_out);
    if (p->model == FISHING_MODEL_V0 || p->model == FISHING_MODEL_V1 || p->model == FISHING_MODEL_V4)
        return launch_kernel(population_draw_kernel<T, FISHING_MODEL_V1>, blocks, 256, s, pt, 0, n, (const T*)x_in,
                             (const T*)z, (const T*)r_arr, (const T*)K_arr, (T*)x_out);
    return FISHING_ERR_MODEL;
}

template <typename T>
int bmsy_sweep_impl(const FishingParams* p, int64_t n_envs, const void* K_arr, const void* r_arr, const void* states,
                    int64_t n_states, void* S_out, fishing_stream_t stream) {
    if (!p || !states || !S_out) return FISHING_ERR_NULL;
    if (n_envs < 0 || n_states < 1) return FISHING_ERR_SIZE;
    if (!is_core_model(p->model)) return FISHING_ERR_MODEL;
    if (n_envs == 0) return FISHING_OK;
    const ParamsT<T> pt = narrow_params<T>(*p);
    const int blocks = grid_for(n_envs, 1 << 20);
    hipStream_t s = (hipStream_t)stream;
    if (p->model == FISHING_MODEL_V2)
        return launch_kernel(bmsy_sweep_kernel<T, FISHING_MODEL_V2>, blocks, 256, s, pt, n_envs, (const T*)K_arr, (const T*)r_arr,
                             (const T*)states, n_states, (T*)S_out);
    return launch_kernel(bmsy_sweep_kernel<T, FISHING_MODEL_V1>, blocks, 256, s, pt, n_envs, (const T*)K_arr, (const T*)r_arr,
                         (const T*)states, n_states, (T*)S_out);
}

}  // namespace fishing

extern "C" {

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
        case FISHING_ERR_UNSUPPORTED: return "this entry point does not serve this combination of flags / streams";
        case FISHING_ERR_VALUE: return "a parameter value outside its domain";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

int64_t fishing_partials_len(void) { return (int64_t)fishing::kPartialSlots * fishing::kPartialFields; }

int fishing_reset_f32(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                      const uint8_t* mask, uint64_t seed, uint64_t reset_counter, fishing_stream_t stream) {
    return fishing::reset_impl<float>(p, n, env_offset, b, mask, seed, reset_counter, stream);
}
int fishing_reset_f64(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b,
                      const uint8_t* mask, uint64_t seed, uint64_t reset_counter, fishing_stream_t stream) {
    return fishing::reset_impl<double>(p, n, env_offset, b, mask, seed, reset_counter, stream);
}

int fishing_v4_params_f32(const FishingParams* p, int64_t n, int64_t env_offset, const int32_t* t, const int32_t* stamp,
                          void* K_out, void* r_out, uint64_t seed, uint64_t step_counter, fishing_stream_t stream) {
    return fishing::v4_params_impl<float>(p, n, env_offset, t, stamp, K_out, r_out, seed, step_counter, stream);
}
int fishing_v4_params_f64(const FishingParams* p, int64_t n, int64_t env_offset, const int32_t* t, const int32_t* stamp,
                          void* K_out, void* r_out, uint64_t seed, uint64_t step_counter, fishing_stream_t stream) {
    return fishing::v4_params_impl<double>(p, n, env_offset, t, stamp, K_out, r_out, seed, step_counter, stream);
}

int fishing_population_draw_f32(const FishingParams* p, int64_t n, const void* x_in, const void* z, const int32_t* model_idx,
                                const void* r, const void* K, void* x_out, fishing_stream_t stream) {
    return fishing::population_draw_impl<float>(p, n, x_in, z, model_idx, r, K, x_out, stream);
}
int fishing_population_draw_f64(const FishingParams* p, int64_t n, const void* x_in, const void* z, const int32_t* model_idx,
                                const void* r, const void* K, void* x_out, fishing_stream_t stream) {
    return fishing::population_draw_impl<double>(p, n, x_in, z, model_idx, r, K, x_out, stream);
}
int fishing_bmsy_sweep_f32(const FishingParams* p, int64_t n_envs, const void* K, const void* r, const void* states,
                           int64_t n_states, void* S_out, fishing_stream_t stream) {
    return fishing::bmsy_sweep_impl<float>(p, n_envs, K, r, states, n_states, S_out, stream);
}
int fishing_bmsy_sweep_f64(const FishingParams* p, int64_t n_envs, const void* K, const void* r, const void* states,
                           int64_t n_states, void* S_out, fishing_stream_t stream) {
    return fishing::bmsy_sweep_impl<double>(p, n_envs, K, r, states, n_states, S_out, stream);
}

int fishing_counter_add(uint64_t* counter, uint64_t delta, fishing_stream_t stream) {
    if (!counter) return FISHING_ERR_NULL;
    if (((uintptr_t)counter) & 7u) return FISHING_ERR_ALIGN;
    return fishing::launch_kernel(fishing::counter_add_kernel, 1, 1, (hipStream_t)stream, counter, delta);
}

int fishing_stream_synchronize(fishing_stream_t stream) { return (int)hipStreamSynchronize((hipStream_t)stream); }

int64_t fishing_partials_slots(int64_t n_envs) {
    // the one-tile forms of step() give every 1024-env tile its own slot; every other kernel stays within kMaxBlocks
    const int64_t tiles = n_envs <= 0 ? 0 : (n_envs + 256 * fishing::kEnvsPerThread - 1) / (256 * fishing::kEnvsPerThread);
    const int64_t want = (tiles + fishing::kReduceSlots - 1) / fishing::kReduceSlots * fishing::kReduceSlots;
    return want < fishing::kReduceSlots ? fishing::kReduceSlots : want > fishing::kPartialSlots ? fishing::kPartialSlots : want;
}

int fishing_reduce_returns_slots(const double* return_partials, int64_t slots, double* out4, fishing_stream_t stream) {
    if (!return_partials || !out4) return FISHING_ERR_NULL;
    if (slots < fishing::kReduceSlots || slots > fishing::kPartialSlots || slots % fishing::kReduceSlots) return FISHING_ERR_SIZE;
    return fishing::launch_kernel(fishing::reduce_returns_kernel, 1, fishing::kReduceThreads, (hipStream_t)stream, return_partials,
                                  (int)(slots / fishing::kReduceSlots), out4);
}

int fishing_reduce_returns(const double* return_partials, double* out4, fishing_stream_t stream) {
    return fishing_reduce_returns_slots(return_partials, fishing::kPartialSlots, out4, stream);
}

int fishing_noise_f32(int64_t n, int64_t env_offset, uint64_t seed, uint64_t counter, int32_t stream_tag,
                      uint32_t* words, float* z0, float* z1, fishing_stream_t stream) {
    if (n < 0 || env_offset < 0 || stream_tag < 0 || stream_tag > 255) return FISHING_ERR_SIZE;
    if (n == 0) return FISHING_OK;
    return fishing::launch_kernel(fishing::noise_kernel, fishing::grid_for(n, 2048), 256, (hipStream_t)stream, n,
                                  (uint64_t)env_offset, seed, counter, (uint32_t)stream_tag, words, z0, z1);
}

int fishing_step_normals_f32(int64_t n, int64_t env_offset, uint64_t seed, uint64_t counter, float* z,
                             fishing_stream_t stream) {
    if (n < 0 || env_offset < 0) return FISHING_ERR_SIZE;
    if (!z) return FISHING_ERR_NULL;
    if (n == 0) return FISHING_OK;
    return fishing::launch_kernel(fishing::step_normals_kernel, fishing::grid_for(n, fishing::kMaxBlocks), 256,
                                  (hipStream_t)stream, n, (uint64_t)env_offset, seed, counter, z);
}

int fishing_math_f64(int64_t n, int32_t fn, const double* in, double* out, fishing_stream_t stream) {
    if (n < 0 || fn < FISHING_MATH_LOG_F64 || fn > FISHING_MATH_EXP_F64) return FISHING_ERR_SIZE;
    if (!in || !out) return FISHING_ERR_NULL;
    if (n == 0) return FISHING_OK;
    return fishing::launch_kernel(fishing::math_kernel, fishing::grid_for(n, 2048), 256, (hipStream_t)stream, n, (int)fn, in, out);
}

int fishing_reset_normals_f32(int64_t n, int64_t env_offset, uint64_t seed, uint64_t counter, int32_t stream_tag,
                              float* zK, float* zr, fishing_stream_t stream) {
    if (n < 0 || env_offset < 0 || stream_tag < 0 || stream_tag > 255) return FISHING_ERR_SIZE;
    if (n == 0) return FISHING_OK;
    return fishing::launch_kernel(fishing::reset_normals_kernel, fishing::grid_for(n, fishing::kMaxBlocks), 256,
                                  (hipStream_t)stream, n, (uint64_t)env_offset, seed, counter, (uint32_t)stream_tag, zK,
                                  zr);
}

}  // extern "C"
