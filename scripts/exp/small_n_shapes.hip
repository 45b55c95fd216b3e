// Experiment (not product), round 3: what bounds the per-step launch at N = 2^17 .. 2^21?
//
// A C-enqueued harness (no Python between launches) over the product's stream shape -- obs / action / t in,
// obs / reward / t / done out, optionally the ep_return accumulator -- with three kernel bodies:
//   empty   nothing (the dispatch floor of that grid)
//   copy    the streams moved with the product's access widths, no arithmetic (the ceiling of that shape)
//   step    the fishing-v1 step (Philox quad block, Box-Muller, auto-reset), stripped of options
// each at THREADS in {64, 128, 256, 512} x envs per thread in {1, 2, 4, 8}, one tile per workgroup.
// `product` rows call libfishing_hip.so's fishing_step_f32 on the same buffers.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 scripts/exp/small_n_shapes.hip \
//         -Lgym_fishing_amd/_lib -lfishing_hip -Wl,-rpath,'$ORIGIN/../../gym_fishing_amd/_lib' -o scripts/exp/_build/small_n_shapes
//   small_n_shapes [reps]            -> one JSON line per (N, body, shape): us per launch, back to back (HIP events)
//   rocprofv3 --kernel-trace ... -- small_n_shapes   -> per-dispatch durations; scripts/exp/small_n_trace.py groups them
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#ifdef HARNESS_EMBED_PRODUCT      // build-variant A/B: the product's step translation unit compiled into this binary
#include "../../gym_fishing_amd/csrc/fishing_step.hip"
#else
#include "../../gym_fishing_amd/csrc/fishing_common.h"
#endif
using namespace fishing;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            std::exit(2);                                                                  \
        }                                                                                  \
    } while (0)

struct Streams {
    float* obs;
    const float* action;
    float* reward;
    uint8_t* done;
    int32_t* t;
    float* ep_return;
    double* partials;
    unsigned long long* stamps;     // REC >= 20: 8 s_memtime stamps per workgroup (wave 0, lane 0)
};

template <int E>
struct VecN;
template <>
struct VecN<1> {
    typedef float F;
    typedef int32_t I;
    typedef uint8_t D;
};
template <>
struct VecN<2> {
    typedef float F __attribute__((ext_vector_type(2)));
    typedef int32_t I __attribute__((ext_vector_type(2)));
    typedef uint16_t D;
};
template <>
struct VecN<4> {
    typedef float F __attribute__((ext_vector_type(4)));
    typedef int32_t I __attribute__((ext_vector_type(4)));
    typedef uint32_t D;
};

template <int E>
__device__ __forceinline__ void ld(const float* p, float (&o)[E]) {
    if constexpr (E == 8) {
        ld<4>(p, reinterpret_cast<float(&)[4]>(o[0]));
        ld<4>(p + 4, reinterpret_cast<float(&)[4]>(o[4]));
    } else {
        const typename VecN<E>::F q = *reinterpret_cast<const typename VecN<E>::F*>(p);
        if constexpr (E == 1) o[0] = q;
        else
#pragma unroll
            for (int j = 0; j < E; ++j) o[j] = q[j];
    }
}
template <int E>
__device__ __forceinline__ void ldi(const int32_t* p, int32_t (&o)[E]) {
    if constexpr (E == 8) {
        ldi<4>(p, reinterpret_cast<int32_t(&)[4]>(o[0]));
        ldi<4>(p + 4, reinterpret_cast<int32_t(&)[4]>(o[4]));
    } else {
        const typename VecN<E>::I q = *reinterpret_cast<const typename VecN<E>::I*>(p);
        if constexpr (E == 1) o[0] = q;
        else
#pragma unroll
            for (int j = 0; j < E; ++j) o[j] = q[j];
    }
}
template <int E, bool NT = false>
__device__ __forceinline__ void st(float* p, const float (&v)[E]) {
    if constexpr (E == 8) {
        st<4, NT>(p, reinterpret_cast<const float(&)[4]>(v[0]));
        st<4, NT>(p + 4, reinterpret_cast<const float(&)[4]>(v[4]));
    } else {
        typename VecN<E>::F q;
        if constexpr (E == 1) q = v[0];
        else
#pragma unroll
            for (int j = 0; j < E; ++j) q[j] = v[j];
        if (NT) __builtin_nontemporal_store(q, reinterpret_cast<typename VecN<E>::F*>(p));
        else *reinterpret_cast<typename VecN<E>::F*>(p) = q;
    }
}
template <int E>
__device__ __forceinline__ void sti(int32_t* p, const int32_t (&v)[E]) {
    if constexpr (E == 8) {
        sti<4>(p, reinterpret_cast<const int32_t(&)[4]>(v[0]));
        sti<4>(p + 4, reinterpret_cast<const int32_t(&)[4]>(v[4]));
    } else {
        typename VecN<E>::I q;
        if constexpr (E == 1) q = v[0];
        else
#pragma unroll
            for (int j = 0; j < E; ++j) q[j] = v[j];
        *reinterpret_cast<typename VecN<E>::I*>(p) = q;
    }
}
template <int E>
__device__ __forceinline__ void std_(uint8_t* p, const bool (&d)[E]) {
    if constexpr (E == 8) {
        std_<4>(p, reinterpret_cast<const bool(&)[4]>(d[0]));
        std_<4>(p + 4, reinterpret_cast<const bool(&)[4]>(d[4]));
    } else {
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < E; ++j) w |= (uint32_t)d[j] << (8 * j);
        __builtin_nontemporal_store((typename VecN<E>::D)w, reinterpret_cast<typename VecN<E>::D*>(p));
    }
}

enum Body { kEmpty = 0, kCopy = 1, kStep = 2 };

// z of env (base + j): the product's quad block -- Box-Muller legs (w0, w1) -> envs 4q, 4q+1; (w2, w3) -> 4q+2, 4q+3
template <int E>
__device__ __forceinline__ void noise(uint64_t seed, uint64_t base, uint64_t counter, float (&z)[E]) {
    if constexpr (E >= 4) {
#pragma unroll
        for (int q = 0; q < E / 4; ++q) {
            float zq[4];
            noise_quad(seed, (base >> 2) + q, counter, zq);
#pragma unroll
            for (int j = 0; j < 4; ++j) z[4 * q + j] = zq[j];
        }
    } else {
        // every lane of a quad's lane group recomputes the quad's block and keeps its own legs
        const Words4 w = philox_block(seed, base >> 2, counter, kStreamNoise);
        const int sub = (int)(base & 3);
        if constexpr (E == 2) {
            const bool hi = sub != 0;
            box_muller(hi ? w.w2 : w.w0, hi ? w.w3 : w.w1, z[0], z[1]);
        } else {
            const bool hi = (sub & 2) != 0;
            float zc, zs;
            box_muller(hi ? w.w2 : w.w0, hi ? w.w3 : w.w1, zc, zs);
            z[0] = (sub & 1) ? zs : zc;
        }
    }
}

// the four record fields summed over the whole wave (every lane ends with field lane & 3): DPP butterfly inside the
// rows, then two gfx950 half / row exchanges (v_permlane16_swap, v_permlane32_swap) -- no LDS, no barrier
__device__ __forceinline__ double swap_sum16(double k) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(k), (unsigned)__double2loint(k), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(k), (unsigned)__double2hiint(k), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double swap_sum32(double k) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(k), (unsigned)__double2loint(k), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(k), (unsigned)__double2hiint(k), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
template <int REC>
__device__ __forceinline__ void wave_partials(const double (&acc)[4], double* partials, int64_t wave_slot) {
    const int lane = threadIdx.x & 63;
    double k = row_sum_fields(acc, lane);
    if constexpr (REC == 2) {
        k += __shfl_xor(k, 16, 64);
        k += __shfl_xor(k, 32, 64);
    } else {
        k = swap_sum32(swap_sum16(k));
    }
    if (lane < 4 && k != 0.0) unsafeAtomicAdd(&partials[wave_slot * 4 + lane], k);
}

// REC: 0 none; 1 the product's (LDS + barrier, 4 atomics per workgroup); 2 per wave, ds_bpermute; 3 per wave, permlane
// swaps; +10 = the record (and its atomic) goes BEFORE the streaming stores of the tile
#ifdef HARNESS_SCALAR_ARGS
// the stream pointers as leading scalar arguments: what -mllvm -amdgpu-kernarg-preload-count=N can hand the wave in
// SGPRs at launch (no s_load round trip in front of the tile's loads); a by-value struct is not preloaded
template <int BODY, int THREADS, int E, bool RET, int REC = 0>
__global__ void __launch_bounds__(THREADS) shape_kernel(float* obs_, const float* action_, int32_t* t_, float* ep_return_,
                                                        float* reward_, uint8_t* done_, double* partials_, const uint64_t seed,
                                                        const uint64_t counter, const float r, const float K, const float sigma,
                                                        const float x0, const int32_t Tmax, const int stagger) {
    const Streams s{obs_, action_, reward_, done_, t_, ep_return_, partials_, nullptr};
#else
template <int BODY, int THREADS, int E, bool RET, int REC = 0>
__global__ void __launch_bounds__(THREADS) shape_kernel(const Streams s, const uint64_t seed, const uint64_t counter, const float r,
                                                        const float K, const float sigma, const float x0, const int32_t Tmax,
                                                        const int stagger) {
#endif
    if constexpr (BODY == kEmpty) return;
    constexpr bool kStamp = REC >= 20;
    constexpr int RECV = kStamp ? REC - 10 : REC;       // the record variant proper
    unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (kStamp) stamp[0] = __builtin_amdgcn_s_memtime();
    // HARNESS_STAGGER = d [, HARNESS_STAGGER_MODE]: half of the workgroups start d x 64 cycles late, so that in a one-round
    // grid their read phase meets the other half's compute / write phase (mode 0: odd workgroups, 1: every other group of
    // 8 = one per XCD, 2: the upper half of the grid)
    if (stagger > 0) {
        const int mode = stagger >> 16, d = stagger & 0xffff;
        const bool late = mode == 0 ? (blockIdx.x & 1) : mode == 1 ? ((blockIdx.x >> 3) & 1) : (blockIdx.x >= gridDim.x / 2);
        if (late && mode < 7)
            for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(1);
    }
    // HARNESS_XZZ (stagger mode 7): odd steps walk the tiles in reverse order IN GROUPS OF 8 -- tile % 8 == blockIdx % 8 stays,
    // i.e. every tile stays on its XCD (blocks are dealt round-robin over the 8 XCDs), so what an XCD's L2 still holds from
    // the end of the previous launch is what this launch's first workgroups read; mode 8: plain reversal (changes the XCD)
    int64_t tile_x = blockIdx.x;
    if ((stagger >> 16) == 7 && (counter & 1)) tile_x = (int64_t)(gridDim.x - 8 - (blockIdx.x & ~7u)) + (blockIdx.x & 7u);
    if ((stagger >> 16) == 8 && (counter & 1)) tile_x = (int64_t)gridDim.x - 1 - blockIdx.x;
    const int64_t base = (tile_x * THREADS + threadIdx.x) * E;
    float o[E], a[E], er[E], on[E], rw[E], erf_[E];
    int32_t t[E], tn[E], tl_[E];
    bool dn[E];
    double slot_old = 0.0;
    if constexpr (RECV % 10 == 4 || RECV % 10 == 5) {
        if (threadIdx.x < 4) slot_old = __builtin_nontemporal_load(&s.partials[(int64_t)blockIdx.x * 4 + threadIdx.x]);
    }
    ld<E>(s.obs + base, o);
    ldi<E>(s.t + base, t);
    ld<E>(s.action + base, a);
    if constexpr (RET) ld<E>(s.ep_return + base, er);
    if constexpr (kStamp) {
        __builtin_amdgcn_sched_barrier(0);
        stamp[1] = __builtin_amdgcn_s_memtime();        // loads issued
    }
    if constexpr (BODY == kCopy) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            on[j] = a[j];
            rw[j] = o[j];
            tn[j] = t[j] + 1;
            dn[j] = t[j] > Tmax;
            if constexpr (RET) er[j] = er[j] + o[j];
            erf_[j] = er[j];
            tl_[j] = tn[j];
        }
    } else {
        __builtin_amdgcn_sched_barrier(0);
        float z[E];
        noise<E>(seed, (uint64_t)base, counter, z);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kStamp) {
            stamp[2] = __builtin_amdgcn_s_memtime();    // Philox + Box-Muller done
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp[3] = __builtin_amdgcn_s_memtime();    // loads landed
            __builtin_amdgcn_sched_barrier(0);
        }
        const float ro = x0 / K - 1.0f;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            env_step<float, FISHING_MODEL_V1>(o[j], t[j], quota_cts<float>(a[j], K), z[j], r, K, sigma, 0.5f, Tmax, on[j], rw[j],
                                              dn[j], tn[j], DivK{true, 1.0f, 1.0});
            if constexpr (RET) {
                erf_[j] = er[j] + rw[j];
                tl_[j] = tn[j];
                er[j] = dn[j] ? 0.0f : erf_[j];
            }
            on[j] = dn[j] ? ro : on[j];
            tn[j] = dn[j] ? 0 : tn[j];
        }
    }
    auto record = [&]() {
        if constexpr (RECV % 10 != 0) {
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            bool any = false;
#pragma unroll
            for (int j = 0; j < E; ++j) any |= dn[j];
            if (__any(any)) {
                float s1 = 0.0f, s2 = 0.0f;
                int32_t cnt = 0, tot = 0;
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    s1 += dn[j] ? erf_[j] : 0.0f;
                    s2 += dn[j] ? erf_[j] * erf_[j] : 0.0f;
                    cnt += dn[j] ? 1 : 0;
                    tot += dn[j] ? tl_[j] : 0;
                }
                acc[0] = (double)s1;
                acc[1] = (double)s2;
                acc[2] = (double)cnt;
                acc[3] = (double)tot;
            }
            if constexpr (RECV % 10 == 1) add_block_partials<THREADS / 64>(acc, s.partials);
            else if constexpr (RECV % 10 == 4 || RECV % 10 == 5) {
                constexpr int kRows = THREADS / 16;
                __shared__ double red[kRows][4];
                const int lane = threadIdx.x & 63;
                const double sfield = row_sum_fields(acc, lane);
                if ((lane & 15) < 4) red[threadIdx.x >> 4][lane & 3] = sfield;
                __syncthreads();
                if (threadIdx.x < 4) {
                    double tot = 0.0;
#pragma unroll
                    for (int w = 0; w < kRows; ++w) tot += red[w][threadIdx.x];
                    if (RECV % 10 == 5) unsafeAtomicAdd(&s.partials[(int64_t)blockIdx.x * 4 + threadIdx.x], tot);   // (static rows, still atomic)
                    else s.partials[(int64_t)blockIdx.x * 4 + threadIdx.x] = slot_old + tot;
                }
            } else wave_partials<RECV % 10>(acc, s.partials, (int64_t)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6));
        }
    };
    if constexpr (kStamp) {
        asm volatile("" ::"v"(on[0]), "v"(rw[0]), "v"(tn[0]));
        __builtin_amdgcn_sched_barrier(0);
        stamp[4] = __builtin_amdgcn_s_memtime();        // arithmetic done
    }
    if constexpr (RECV >= 10) record();
    if constexpr (kStamp) {
        __builtin_amdgcn_sched_barrier(0);
        stamp[5] = __builtin_amdgcn_s_memtime();        // record (reduction + atomic issued)
    }
    st<E, true>(s.reward + base, rw);
    std_<E>(s.done + base, dn);
    if constexpr (RET) st<E>(s.ep_return + base, er);
    st<E>(s.obs + base, on);
    sti<E>(s.t + base, tn);
    if constexpr (RECV > 0 && RECV < 10) record();
    if constexpr (kStamp) {
        __builtin_amdgcn_sched_barrier(0);
        stamp[6] = __builtin_amdgcn_s_memtime();        // stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[7] = __builtin_amdgcn_s_memtime();        // stores + atomic acknowledged
        if (threadIdx.x == 0 && s.stamps) {
#pragma unroll
            for (int k = 0; k < 8; ++k) s.stamps[(int64_t)blockIdx.x * 8 + k] = stamp[k];
        }
    }
}

struct Case {
    const char* body;
    int threads, ept;
    bool ret;
    int rec;
    void (*launch)(int64_t n, const Streams&, uint64_t counter, hipStream_t);
};

static int g_stagger = 0;
template <int BODY, int THREADS, int E, bool RET, int REC = 0>
void launch_case(int64_t n, const Streams& s, uint64_t counter, hipStream_t st_) {
    const int64_t tile = (int64_t)THREADS * E;
#ifdef HARNESS_SCALAR_ARGS
    shape_kernel<BODY, THREADS, E, RET, REC><<<dim3((unsigned)(n / tile)), dim3(THREADS), 0, st_>>>(
        s.obs, s.action, s.t, s.ep_return, s.reward, s.done, s.partials, 1234u, counter, 0.3f, 1.0f, 0.1f, 0.75f, 100, g_stagger);
#else
    shape_kernel<BODY, THREADS, E, RET, REC><<<dim3((unsigned)(n / tile)), dim3(THREADS), 0, st_>>>(s, 1234u, counter, 0.3f, 1.0f, 0.1f,
                                                                                                 0.75f, 100, g_stagger);
#endif
}

static FishingParams g_params;
static bool g_prod_ret = false;
static double* g_partials = nullptr;
static unsigned long long* g_stamps = nullptr;
void launch_product(int64_t n, const Streams& s, uint64_t counter, hipStream_t st_) {
    FishingBuffers b;
    std::memset(&b, 0, sizeof b);
    b.obs = s.obs;
    b.action = s.action;
    b.reward = s.reward;
    b.done = s.done;
    b.t = s.t;
    if (g_prod_ret) {
        b.ep_return = s.ep_return;
        b.return_partials = g_partials;
    }
    const int rc = fishing_step_f32(&g_params, n, 0, &b, 1234u, counter, st_);
    if (rc != 0) {
        std::fprintf(stderr, "fishing_step_f32 rc %d\n", rc);
        std::exit(3);
    }
}
// the same step as HARNESS_SPLIT launches over contiguous parts of the batch (env_offset keys the noise: same results)
static int g_split = 1;
void launch_product_split(int64_t n, const Streams& s, uint64_t counter, hipStream_t st_) {
    const int64_t part = n / g_split;
    for (int k = 0; k < g_split; ++k) {
        FishingBuffers b;
        std::memset(&b, 0, sizeof b);
        b.obs = s.obs + k * part;
        b.action = s.action + k * part;
        b.reward = s.reward + k * part;
        b.done = s.done + k * part;
        b.t = s.t + k * part;
        if (g_prod_ret) {
            b.ep_return = s.ep_return + k * part;
            b.return_partials = g_partials;     // (shared slots: fine for timing; a product version would give each part its own)
        }
        const int rc = fishing_step_f32(&g_params, part, k * part, &b, 1234u, counter, st_);
        if (rc != 0) {
            std::fprintf(stderr, "fishing_step_f32 rc %d\n", rc);
            std::exit(3);
        }
    }
}
void launch_split_bare(int64_t n, const Streams& s, uint64_t c, hipStream_t st_) {
    g_prod_ret = false;
    launch_product_split(n, s, c, st_);
}
void launch_split_ret(int64_t n, const Streams& s, uint64_t c, hipStream_t st_) {
    g_prod_ret = true;
    launch_product_split(n, s, c, st_);
}
void launch_product_bare(int64_t n, const Streams& s, uint64_t c, hipStream_t st_) {
    g_prod_ret = false;
    launch_product(n, s, c, st_);
}
void launch_product_ret(int64_t n, const Streams& s, uint64_t c, hipStream_t st_) {
    g_prod_ret = true;
    launch_product(n, s, c, st_);
}

#define SHAPES(BODY, NAME, RET)                                                                                          \
    {NAME, 64, 1, RET, 0, launch_case<BODY, 64, 1, RET>}, {NAME, 64, 2, RET, 0, launch_case<BODY, 64, 2, RET>},              \
        {NAME, 64, 4, RET, 0, launch_case<BODY, 64, 4, RET>}, {NAME, 128, 1, RET, 0, launch_case<BODY, 128, 1, RET>},        \
        {NAME, 128, 2, RET, 0, launch_case<BODY, 128, 2, RET>}, {NAME, 128, 4, RET, 0, launch_case<BODY, 128, 4, RET>},      \
        {NAME, 256, 1, RET, 0, launch_case<BODY, 256, 1, RET>}, {NAME, 256, 2, RET, 0, launch_case<BODY, 256, 2, RET>},      \
        {NAME, 256, 4, RET, 0, launch_case<BODY, 256, 4, RET>}, {NAME, 256, 8, RET, 0, launch_case<BODY, 256, 8, RET>},      \
        {NAME, 512, 2, RET, 0, launch_case<BODY, 512, 2, RET>}, {NAME, 512, 4, RET, 0, launch_case<BODY, 512, 4, RET>},      \
        {NAME, 1024, 4, RET, 0, launch_case<BODY, 1024, 4, RET>}

#define RECS(TH, E)                                                                                                   \
    {"steprec", TH, E, true, 1, launch_case<kStep, TH, E, true, 1>}, {"steprec", TH, E, true, 2, launch_case<kStep, TH, E, true, 2>},   \
        {"steprec", TH, E, true, 3, launch_case<kStep, TH, E, true, 3>}, {"steprec", TH, E, true, 11, launch_case<kStep, TH, E, true, 11>}, \
        {"steprec", TH, E, true, 12, launch_case<kStep, TH, E, true, 12>}, {"steprec", TH, E, true, 13, launch_case<kStep, TH, E, true, 13>}, \
        {"steprec", TH, E, true, 4, launch_case<kStep, TH, E, true, 4>}, {"steprec", TH, E, true, 14, launch_case<kStep, TH, E, true, 14>}, \
        {"steprec", TH, E, true, 5, launch_case<kStep, TH, E, true, 5>}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? std::atoi(argv[1]) : 400;
    if (std::getenv("HARNESS_SPLIT")) g_split = std::atoi(std::getenv("HARNESS_SPLIT"));
    if (std::getenv("HARNESS_STAGGER"))
        g_stagger = std::atoi(std::getenv("HARNESS_STAGGER")) | ((std::getenv("HARNESS_STAGGER_MODE") ? std::atoi(std::getenv("HARNESS_STAGGER_MODE")) : 0) << 16);
    const int lo = argc > 2 ? std::atoi(argv[2]) : 17, hi = argc > 3 ? std::atoi(argv[3]) : 21;
    const char* only = argc > 4 ? argv[4] : nullptr;      // body filter
    std::vector<Case> cases = {
        {"empty", 256, 4, false, 0, launch_case<kEmpty, 256, 4, false>},
        {"empty", 64, 4, false, 0, launch_case<kEmpty, 64, 4, false>},
        {"empty", 64, 1, false, 0, launch_case<kEmpty, 64, 1, false>},
        SHAPES(kCopy, "copy", false),
        SHAPES(kCopy, "copy", true),
        SHAPES(kStep, "step", false),
        SHAPES(kStep, "step", true),
        RECS(256, 4), RECS(128, 4), RECS(256, 2), RECS(512, 4),
        {"stamps", 256, 4, true, 21, launch_case<kStep, 256, 4, true, 21>},
        {"split", 256, 4, false, 0, launch_split_bare},
        {"split", 256, 4, true, 1, launch_split_ret},
        {"product", 256, 4, false, 0, launch_product_bare},
        {"product", 256, 4, true, 1, launch_product_ret},
    };
    std::memset(&g_params, 0, sizeof g_params);
    g_params.model = FISHING_MODEL_V1;
    g_params.Tmax = 100;
    g_params.flags = FISHING_FLAG_AUTO_RESET;
    g_params.r = 0.3;
    g_params.K = 1.0;
    g_params.sigma = 0.1;
    g_params.C = 0.5;
    g_params.x0 = 0.75;
    g_params.n_actions = 100;

    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipMalloc(&g_partials, 65536 * 4 * sizeof(double)));
    CK(hipMalloc(&g_stamps, 65536 * 8 * sizeof(unsigned long long)));
    for (int ln = lo; ln <= hi; ++ln) {
        const int64_t n = 1ll << ln;
        // one arena, the product's placement: streams staggered by 12 KiB; 8 action batches
        const size_t gap = 12288, stride = (size_t)n * 4 + gap;
        char* arena;
        CK(hipMalloc(&arena, stride * 13 + (size_t)n));
        float* acts = (float*)(arena + 5 * stride);
        std::vector<float> ha((size_t)n);
        for (int k = 0; k < 8; ++k) {
            for (int64_t i = 0; i < n; ++i) ha[i] = -1.0f + 0.2f * (float)((i * 2654435761u + k * 40503u) & 0xffff) / 65536.0f;
            CK(hipMemcpy((char*)acts + k * stride, ha.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        }
        Streams s{(float*)arena, acts, (float*)(arena + stride), (uint8_t*)(arena + 4 * stride), (int32_t*)(arena + 2 * stride),
                  (float*)(arena + 3 * stride), g_partials, g_stamps};
        // HARNESS_RR=<burst>: round-robin over the selected cases, <burst> launches of each in turn, `reps` times --
        // per-kernel durations then come from a rocprofv3 trace and see the same clock / power state on average
        const int rr = std::getenv("HARNESS_RR") ? std::atoi(std::getenv("HARNESS_RR")) : 0;
        std::vector<const Case*> sel;
        for (const Case& c : cases) {
            if (only) {       // comma-separated body names
                const char* hit = std::strstr(only, c.body);
                const size_t len = std::strlen(c.body);
                if (!hit || (hit != only && hit[-1] != ',') || (hit[len] != 0 && hit[len] != ',')) continue;
            }
            if (const char* shp = std::getenv("HARNESS_SHAPE")) {       // e.g. 256x4
                char buf[32];
                std::snprintf(buf, sizeof buf, "%dx%d", c.threads, c.ept);
                if (std::strcmp(buf, shp) != 0) continue;
            }
            if ((int64_t)c.threads * c.ept > n) continue;
            if (rr > 0) {
                sel.push_back(&c);
                continue;
            }
            // fresh state: obs = -0.25, t = 0, ep_return = 0
            for (int64_t i = 0; i < n; ++i) ha[i] = -0.25f;
            CK(hipMemcpy(s.obs, ha.data(), (size_t)n * 4, hipMemcpyHostToDevice));
            CK(hipMemset(s.t, 0, (size_t)n * 4));
            CK(hipMemset(s.ep_return, 0, (size_t)n * 4));
            CK(hipMemset(g_partials, 0, 65536 * 4 * sizeof(double)));
            uint64_t counter = 0;
            auto run = [&](int k) {
                for (int i = 0; i < k; ++i, ++counter) {
                    Streams q = s;
                    q.action = (const float*)((const char*)acts + (counter % 8) * stride);
                    c.launch(n, q, counter, stream);
                }
            };
            run(64);
            CK(hipStreamSynchronize(stream));
            double best = 1e30, sum = 0;
            const int rounds = 5;
            for (int rd = 0; rd < rounds; ++rd) {
                run(16);
                CK(hipEventRecord(e0, stream));
                run(reps);
                CK(hipEventRecord(e1, stream));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / reps;
                best = us < best ? us : best;
                sum += us;
            }
            CK(hipGetLastError());
            std::vector<double> hp(65536 * 4);
            CK(hipMemcpy(hp.data(), g_partials, hp.size() * 8, hipMemcpyDeviceToHost));
            double fields[4] = {0, 0, 0, 0};
            for (size_t i = 0; i < hp.size(); ++i) fields[i & 3] += hp[i];
            if (std::strcmp(c.body, "stamps") == 0) {       // phase durations in shader cycles: median over the workgroups of the LAST launch
                const int64_t nwg = n / 1024;
                std::vector<unsigned long long> hs((size_t)nwg * 8);
                CK(hipMemcpy(hs.data(), g_stamps, hs.size() * 8, hipMemcpyDeviceToHost));
                const char* names[7] = {"issue_loads", "philox", "wait_loads", "arithmetic", "record", "issue_stores", "wait_acks"};
                unsigned long long first = ~0ull, last = 0;
                for (int64_t w = 0; w < nwg; ++w) {
                    first = hs[w * 8] < first ? hs[w * 8] : first;
                    last = hs[w * 8 + 7] > last ? hs[w * 8 + 7] : last;
                }
                std::printf("{\"log2_n\": %d, \"stamps\": true, \"first_start_to_last_end_cycles\": %llu", ln, last - first);
                for (int k = 0; k < 7; ++k) {
                    std::vector<long long> d((size_t)nwg);
                    for (int64_t w = 0; w < nwg; ++w) d[w] = (long long)(hs[w * 8 + k + 1] - hs[w * 8 + k]);
                    std::sort(d.begin(), d.end());
                    std::printf(", \"%s\": [%lld, %lld, %lld]", names[k], d[nwg / 10], d[nwg / 2], d[nwg * 9 / 10]);
                }
                std::vector<long long> st0((size_t)nwg);
                for (int64_t w = 0; w < nwg; ++w) st0[w] = (long long)(hs[w * 8] - first);
                std::sort(st0.begin(), st0.end());
                std::printf(", \"start_offset\": [%lld, %lld, %lld]}\n", st0[nwg / 10], st0[nwg / 2], st0[nwg * 9 / 10]);
            }
            std::printf("{\"log2_n\": %d, \"body\": \"%s\", \"threads\": %d, \"ept\": %d, \"ret\": %s, \"grid\": %lld, "
                        "\"rec\": %d, \"record\": [%.9g, %.9g, %.0f, %.0f], \"us_back_to_back_mean\": %.3f, \"us_back_to_back_min\": %.3f}\n",
                        ln, c.body, c.threads, c.ept, c.ret ? "true" : "false", (long long)(n / ((int64_t)c.threads * c.ept)),
                        c.rec, fields[0], fields[1], fields[2], fields[3], sum / rounds, best);
            std::fflush(stdout);
        }
        if (rr > 0) {
            for (int64_t i = 0; i < n; ++i) ha[i] = -0.25f;
            CK(hipMemcpy(s.obs, ha.data(), (size_t)n * 4, hipMemcpyHostToDevice));
            CK(hipMemset(s.t, 0, (size_t)n * 4));
            CK(hipMemset(s.ep_return, 0, (size_t)n * 4));
            CK(hipMemset(g_partials, 0, 65536 * 4 * sizeof(double)));
            uint64_t counter = 0;
            for (int rep = 0; rep < reps; ++rep) {
                for (const Case* c : sel)
                    for (int b = 0; b < rr; ++b, ++counter) {
                        Streams q = s;
                        q.action = (const float*)((const char*)acts + (counter % 8) * stride);
                        c->launch(n, q, counter, stream);
                    }
                if ((rep & 63) == 63) CK(hipStreamSynchronize(stream));
            }
            CK(hipStreamSynchronize(stream));
            std::printf("{\"log2_n\": %d, \"round_robin\": %d, \"cases\": %zu, \"reps\": %d}\n", ln, rr, sel.size(), reps);
            std::fflush(stdout);
        }
        CK(hipFree(arena));
    }
    return 0;
}
