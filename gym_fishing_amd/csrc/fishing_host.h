// fishing_host.h -- host side of libfishing_hip.so shared by its translation units: packing the C-ABI structs into kernel
// arguments, the run-time -> compile-time model dispatch, the launch helper and the argument checks.
#pragma once
#include <stdint.h>

#include <cmath>
#include <string>
#include <tuple>

#include "fishing_common.h"

namespace fishing {

static inline bool misaligned(const void* p) { return p && (((uintptr_t)p) & 15u); }

// argument checks common to step / reset / rollout (fishing_step.hip)
int check_common(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b);
// grid of the general step / rollout kernels
void launch_shape(const FishingParams* p, int64_t n, int& blocks, int& threads);
// kNoiseNone / kNoiseExt / kNoisePhilox for this request
int noise_mode(const FishingParams* p, const FishingBuffers* b);

template <typename T>
inline GrowthT<T> make_growth(double r, double K, double sigma, double C, double M, double theta, double q,
                              double b, double a, int kind) {
    GrowthT<T> g{r, K, sigma, C, M, theta, q, b, a, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0};
    g.gc = r * (1.0 - C) / K;
    g.bq = std::pow(b, q);
    g.invK = 1.0 / K;
    g.invM = 1.0 / M;
    if (kind == FISHING_KIND_BEVERTON_HOLT) {
        const double rc = r < 0 ? 0 : r, Kc = K < 0 ? 0 : K;
        g.logA = std::log(rc + 1.0);
        g.A = rc + 1.0;
        g.B = Kc / rc;
        g.invB = rc / Kc;
        g.invK = 1.0 / Kc;
    } else {
        g.logA = std::log(r + 1.0);
        // (Myers: log(r + 1) of a negative number is NaN in the reference, and so is its population: the algebraic forms,
        // which carry A itself, must not turn that into max(0, negative) = an extinct stock)
        g.A = (r + 1.0 < 0.0) ? std::nan("") : r + 1.0;
    }
    const double e = kind == FISHING_KIND_MAY ? q : theta;
    g.ipow = (e == 1.0 || e == 2.0 || e == 3.0 || e == 4.0) ? (int32_t)e : 0;
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
        q.growth = make_growth<T>(p.r, p.K, p.sigma, p.C, p.M, p.theta, p.q, p.b, p.a, kind_of_model(p.model));
    for (int k = 0; k < FISHING_N_KINDS; ++k) {
        q.kinds[k] = p.kinds[k];
        q.zoo[k] = GrowthT<T>{};
        if (p.model == FISHING_MODEL_V11) {
            const FishingGrowthParams& g = p.zoo[k];
            q.zoo[k] = make_growth<T>(g.r, g.K, g.sigma, g.C, g.M, g.theta, g.q, g.b, g.a, k);
        }
    }
    return q;
}


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
    q.stamp = b.v4_stamp;
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
template <typename... P, typename... A>
inline int launch_kernel(void (*kernel)(P...), int blocks, int threads, hipStream_t stream, A&&... args) {
    std::tuple<P...> packed{static_cast<P>(args)...};
    return std::apply(
        [&](auto&... a) {
            void* argv[] = {(void*)&a...};
            return (int)hipLaunchKernel((const void*)kernel, dim3((unsigned)blocks), dim3((unsigned)threads), argv, 0, stream);
        },
        packed);
}



inline DivK make_divk(double K) {
    int e = 0;
    const bool p2 = K > 0 && std::isfinite(K) && std::frexp(K, &e) == 0.5 && e > -120 && e < 120;
    return DivK{p2, p2 ? (float)(1.0 / K) : 0.0f, p2 ? 1.0 / K : 0.0};
}

}  // namespace fishing
