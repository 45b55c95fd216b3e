// Experiment (not product): fishing-v1 f32 step with EPT envs per thread, to see whether wider
// per-lane accesses move the 2^22-env step closer to the copy rate.
#include "../../gym_fishing_amd/csrc/fishing_common.h"
using namespace fishing;

template <int EPT, int MAP = 0>
__global__ void __launch_bounds__(256)
exp_step_kernel(int64_t n, float* __restrict__ obs, const float* __restrict__ action, float* __restrict__ reward,
                uint8_t* __restrict__ done, int32_t* __restrict__ t, uint64_t seed, uint64_t step_counter,
                float r, float K, float sigma, float x0, int32_t Tmax) {
    const int64_t tile_envs = (int64_t)blockDim.x * EPT;
    const int64_t ntiles = n / tile_envs;
    const int64_t per = (ntiles + gridDim.x - 1) / gridDim.x;
    for (int64_t it = 0; it < per; ++it) {
        // MAP 0: tile = block + it * grid (strided);  MAP 1: contiguous run of tiles per block;
        // MAP 2: XCD-contiguous: blocks that share an XCD (b % 8) walk one contiguous eighth
        int64_t tile;
        if (MAP == 0) tile = blockIdx.x + it * gridDim.x;
        else if (MAP == 1) tile = (int64_t)blockIdx.x * per + it;
        else tile = (int64_t)(blockIdx.x & 7) * (ntiles / 8) + (int64_t)(blockIdx.x >> 3) * per + it;
        if (tile >= ntiles) break;
        const int64_t base = (tile * blockDim.x + threadIdx.x) * EPT;
        float o[EPT], a[EPT], z[EPT], on[EPT], rw[EPT];
        int32_t tt[EPT], tn[EPT];
        bool dn[EPT];
#pragma unroll
        for (int v = 0; v < EPT / 4; ++v) {
            const Vec4<float> q = *reinterpret_cast<const Vec4<float>*>(obs + base + 4 * v);
            const Vec4<float> qa = *reinterpret_cast<const Vec4<float>*>(action + base + 4 * v);
            const Vec4<int32_t> qt = *reinterpret_cast<const Vec4<int32_t>*>(t + base + 4 * v);
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[4 * v + j] = q.v[j]; a[4 * v + j] = qa.v[j]; tt[4 * v + j] = qt.v[j]; }
        }
        const uint64_t pair = (uint64_t)base >> 1;
#pragma unroll
        for (int q = 0; q < EPT / 2; ++q) {
            const Words4 w = philox_block(seed, pair + q, step_counter, kStreamNoise);
            box_muller(w.w0, w.w1, z[2 * q], z[2 * q + 1]);
        }
        const float ro = x0 / K - 1.0f;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            env_step<float, FISHING_MODEL_V1>(o[j], tt[j], quota_cts<float>(a[j], K), z[j], r, K, sigma, 0.5f, Tmax, on[j],
                                              rw[j], dn[j], tn[j]);
            on[j] = dn[j] ? ro : on[j];
            tn[j] = dn[j] ? 0 : tn[j];
        }
#pragma unroll
        for (int v = 0; v < EPT / 4; ++v) {
            Vec4<float> q, qr;
            Vec4<int32_t> qt;
            uint32_t packed = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                q.v[j] = on[4 * v + j]; qr.v[j] = rw[4 * v + j]; qt.v[j] = tn[4 * v + j];
                packed |= (uint32_t)dn[4 * v + j] << (8 * j);
            }
            *reinterpret_cast<Vec4<float>*>(obs + base + 4 * v) = q;
            *reinterpret_cast<Vec4<float>*>(reward + base + 4 * v) = qr;
            *reinterpret_cast<Vec4<int32_t>*>(t + base + 4 * v) = qt;
            *reinterpret_cast<uint32_t*>(done + base + 4 * v) = packed;
        }
    }
}

// software-pipelined variant: the next tile's loads are issued before the current tile is computed
__global__ void __launch_bounds__(256)
exp_step_pipelined(int64_t n, float* __restrict__ obs, const float* __restrict__ action, float* __restrict__ reward,
                   uint8_t* __restrict__ done, int32_t* __restrict__ t, uint64_t seed, uint64_t step_counter,
                   float r, float K, float sigma, float x0, int32_t Tmax) {
    const int64_t ntiles = n / 1024;
    int64_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    int64_t base = (tile * 256 + threadIdx.x) * 4;
    Vec4<float> qo = *reinterpret_cast<const Vec4<float>*>(obs + base);
    Vec4<float> qa = *reinterpret_cast<const Vec4<float>*>(action + base);
    Vec4<int32_t> qt = *reinterpret_cast<const Vec4<int32_t>*>(t + base);
    const float ro = x0 / K - 1.0f;
    while (true) {
        const int64_t next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        const int64_t nbase = (next * 256 + threadIdx.x) * 4;
        Vec4<float> no, na;
        Vec4<int32_t> nt;
        if (has_next) {
            no = *reinterpret_cast<const Vec4<float>*>(obs + nbase);
            na = *reinterpret_cast<const Vec4<float>*>(action + nbase);
            nt = *reinterpret_cast<const Vec4<int32_t>*>(t + nbase);
        }
        float z[4];
        const uint64_t pair = (uint64_t)base >> 1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const Words4 w = philox_block(seed, pair + q, step_counter, kStreamNoise);
            box_muller(w.w0, w.w1, z[2 * q], z[2 * q + 1]);
        }
        Vec4<float> wo, wr;
        Vec4<int32_t> wt;
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float on, rw;
            bool dn;
            int32_t tn;
            env_step<float, FISHING_MODEL_V1>(qo.v[j], qt.v[j], quota_cts<float>(qa.v[j], K), z[j], r, K, sigma, 0.5f, Tmax, on,
                                              rw, dn, tn);
            wo.v[j] = dn ? ro : on;
            wt.v[j] = dn ? 0 : tn;
            wr.v[j] = rw;
            packed |= (uint32_t)dn << (8 * j);
        }
        *reinterpret_cast<Vec4<float>*>(obs + base) = wo;
        *reinterpret_cast<Vec4<float>*>(reward + base) = wr;
        *reinterpret_cast<Vec4<int32_t>*>(t + base) = wt;
        *reinterpret_cast<uint32_t*>(done + base) = packed;
        if (!has_next) break;
        tile = next;
        base = nbase;
        qo = no;
        qa = na;
        qt = nt;
    }
}

// stateful-generator comparison (north star: "LDS-staged Philox/xoshiro RNG state per wavefront"):
// xoshiro128++ with its 16-byte state per env in HBM, read and written every step.  One draw gives
// 32 bits; a normal needs two -> two state advances per env-step.  The state stream is already
// 16 B/lane coalesced, so staging it through LDS would add a round trip and buy nothing.
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
__device__ __forceinline__ uint32_t xoshiro_next(uint4& s) {
    const uint32_t result = rotl32(s.x + s.w, 7) + s.x;
    const uint32_t tt = s.y << 9;
    s.z ^= s.x;
    s.w ^= s.y;
    s.y ^= s.z;
    s.x ^= s.w;
    s.z ^= tt;
    s.w = rotl32(s.w, 11);
    return result;
}
__global__ void __launch_bounds__(256)
exp_step_xoshiro(int64_t n, float* __restrict__ obs, const float* __restrict__ action, float* __restrict__ reward,
                 uint8_t* __restrict__ done, int32_t* __restrict__ t, uint4* __restrict__ state, float r, float K,
                 float sigma, float x0, int32_t Tmax) {
    const int64_t ntiles = n / 1024;
    const float ro = x0 / K - 1.0f;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = (tile * 256 + threadIdx.x) * 4;
        const Vec4<float> qo = *reinterpret_cast<const Vec4<float>*>(obs + base);
        const Vec4<float> qa = *reinterpret_cast<const Vec4<float>*>(action + base);
        const Vec4<int32_t> qt = *reinterpret_cast<const Vec4<int32_t>*>(t + base);
        uint4 st[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) st[j] = state[base + j];
        Vec4<float> wo, wr;
        Vec4<int32_t> wt;
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float zc, zs, on, rw;
            bool dn;
            int32_t tn;
            const uint32_t w0 = xoshiro_next(st[j]);
            const uint32_t w1 = xoshiro_next(st[j]);
            box_muller(w0, w1, zc, zs);
            env_step<float, FISHING_MODEL_V1>(qo.v[j], qt.v[j], quota_cts<float>(qa.v[j], K), zc, r, K, sigma, 0.5f, Tmax, on, rw,
                                              dn, tn);
            wo.v[j] = dn ? ro : on;
            wt.v[j] = dn ? 0 : tn;
            wr.v[j] = rw;
            packed |= (uint32_t)dn << (8 * j);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) state[base + j] = st[j];
        *reinterpret_cast<Vec4<float>*>(obs + base) = wo;
        *reinterpret_cast<Vec4<float>*>(reward + base) = wr;
        *reinterpret_cast<Vec4<int32_t>*>(t + base) = wt;
        *reinterpret_cast<uint32_t*>(done + base) = packed;
    }
}

extern "C" int exp_step_xo(int blocks, int64_t n, float* obs, const float* action, float* reward, uint8_t* done, int32_t* t,
                           void* state, void* stream) {
    exp_step_xoshiro<<<blocks, 256, 0, (hipStream_t)stream>>>(n, obs, action, reward, done, t, (uint4*)state, 0.3f, 1.0f, 0.1f,
                                                             0.75f, 100);
    return (int)hipGetLastError();
}

// copy with the same stream shape: 3 x 4-byte inputs -> 3 x 4-byte outputs + 1 byte
template <int EPT>
__global__ void __launch_bounds__(256)
exp_copy_kernel(int64_t n, float* __restrict__ obs, const float* __restrict__ action, float* __restrict__ reward,
                uint8_t* __restrict__ done, int32_t* __restrict__ t) {
    const int64_t tile_envs = (int64_t)blockDim.x * EPT;
    const int64_t ntiles = n / tile_envs;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = (tile * blockDim.x + threadIdx.x) * EPT;
#pragma unroll
        for (int v = 0; v < EPT / 4; ++v) {
            Vec4<float> q = *reinterpret_cast<const Vec4<float>*>(obs + base + 4 * v);
            const Vec4<float> qa = *reinterpret_cast<const Vec4<float>*>(action + base + 4 * v);
            Vec4<int32_t> qt = *reinterpret_cast<const Vec4<int32_t>*>(t + base + 4 * v);
            uint32_t packed = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { q.v[j] += qa.v[j]; qt.v[j] += 1; packed |= (uint32_t)(qt.v[j] & 1) << (8 * j); }
            *reinterpret_cast<Vec4<float>*>(obs + base + 4 * v) = q;
            *reinterpret_cast<Vec4<float>*>(reward + base + 4 * v) = qa;
            *reinterpret_cast<Vec4<int32_t>*>(t + base + 4 * v) = qt;
            *reinterpret_cast<uint32_t*>(done + base + 4 * v) = packed;
        }
    }
}

extern "C" int exp_step(int ept, int copy, int blocks, int64_t n, float* obs, const float* action, float* reward,
                        uint8_t* done, int32_t* t, uint64_t seed, uint64_t counter, void* stream) {
    hipStream_t s = (hipStream_t)stream;
#define RUN(E)                                                                                                  \
    if (copy) exp_copy_kernel<E><<<blocks, 256, 0, s>>>(n, obs, action, reward, done, t);                       \
    else exp_step_kernel<E><<<blocks, 256, 0, s>>>(n, obs, action, reward, done, t, seed, counter, 0.3f, 1.0f,  \
                                                   0.1f, 0.75f, 100)
    if (ept == 4) { RUN(4); } else if (ept == 8) { RUN(8); } else if (ept == 16) { RUN(16); }
    else if (ept == 41) exp_step_kernel<4, 1><<<blocks, 256, 0, s>>>(n, obs, action, reward, done, t, seed, counter, 0.3f, 1.0f, 0.1f, 0.75f, 100);
    else if (ept == 43) exp_step_pipelined<<<blocks, 256, 0, s>>>(n, obs, action, reward, done, t, seed, counter, 0.3f, 1.0f, 0.1f, 0.75f, 100);
    else if (ept == 42) exp_step_kernel<4, 2><<<blocks, 256, 0, s>>>(n, obs, action, reward, done, t, seed, counter, 0.3f, 1.0f, 0.1f, 0.75f, 100);
    return (int)hipGetLastError();
}
