// Experiment (not product): the fp64 parity layout's access shape.  A copy-shaped kernel over the same streams as
// fishing_step_f64 (obs f64 R+W, action f32 R, t i32 R+W, reward f64 W, done u8 W = 37 B/env) with EPT envs per
// thread: EPT = 4 is the product's shape (32 B of obs per lane = two 16-byte accesses 32 B apart), EPT = 2 gives one
// contiguous 16-byte access per lane per stream.  A few flops per env stand in for the arithmetic.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int EPT>
__global__ void __launch_bounds__(256)
shape_kernel(int64_t n, double* __restrict__ obs, const float* __restrict__ action, double* __restrict__ reward,
             uint8_t* __restrict__ done, int32_t* __restrict__ t) {
    const int64_t tile_envs = (int64_t)blockDim.x * EPT;
    const int64_t ntiles = n / tile_envs;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = (tile * blockDim.x + threadIdx.x) * EPT;
        double o[EPT], rw[EPT];
        float a[EPT];
        int32_t tt[EPT];
        uint8_t d[EPT];
#pragma unroll
        for (int v = 0; v < EPT / 2; ++v) {
            const double2 q = *reinterpret_cast<const double2*>(obs + base + 2 * v);
            o[2 * v] = q.x;
            o[2 * v + 1] = q.y;
        }
        if (EPT == 4) {
            const float4 qa = *reinterpret_cast<const float4*>(action + base);
            const int4 qt = *reinterpret_cast<const int4*>(t + base);
            a[0] = qa.x; a[1] = qa.y; a[EPT - 2] = qa.z; a[EPT - 1] = qa.w;
            tt[0] = qt.x; tt[1] = qt.y; tt[EPT - 2] = qt.z; tt[EPT - 1] = qt.w;
        } else {
            const float2 qa = *reinterpret_cast<const float2*>(action + base);
            const int2 qt = *reinterpret_cast<const int2*>(t + base);
            a[0] = qa.x; a[1] = qa.y;
            tt[0] = qt.x; tt[1] = qt.y;
        }
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const double x = o[j] + 1.0;
            const double h = fmin(x, (double)a[j] + 1.0);
            const double g = (x - h) * 1.3 - (x - h) * (x - h) * 0.3;
            o[j] = g - 1.0;
            rw[j] = h;
            tt[j] += 1;
            d[j] = (tt[j] > 100) || (g <= 0.0);
            if (d[j]) { o[j] = -0.25; tt[j] = 0; }
        }
#pragma unroll
        for (int v = 0; v < EPT / 2; ++v) {
            *reinterpret_cast<double2*>(obs + base + 2 * v) = double2{o[2 * v], o[2 * v + 1]};
            __builtin_nontemporal_store(rw[2 * v], reward + base + 2 * v);
            __builtin_nontemporal_store(rw[2 * v + 1], reward + base + 2 * v + 1);
        }
        if (EPT == 4) {
            *reinterpret_cast<int4*>(t + base) = int4{tt[0], tt[1], tt[EPT - 2], tt[EPT - 1]};
            *reinterpret_cast<uint32_t*>(done + base) = (uint32_t)d[0] | ((uint32_t)d[1] << 8) | ((uint32_t)d[EPT - 2] << 16) | ((uint32_t)d[EPT - 1] << 24);
        } else {
            *reinterpret_cast<int2*>(t + base) = int2{tt[0], tt[1]};
            *reinterpret_cast<uint16_t*>(done + base) = (uint16_t)((uint16_t)d[0] | ((uint16_t)d[1] << 8));
        }
    }
}

extern "C" int exp_shape(int ept, int blocks, int threads, int64_t n, void* obs, const void* action, void* reward, void* done,
                         void* t, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (ept == 4)
        hipLaunchKernelGGL(shape_kernel<4>, dim3(blocks), dim3(threads), 0, s, n, (double*)obs, (const float*)action, (double*)reward, (uint8_t*)done, (int32_t*)t);
    else
        hipLaunchKernelGGL(shape_kernel<2>, dim3(blocks), dim3(threads), 0, s, n, (double*)obs, (const float*)action, (double*)reward, (uint8_t*)done, (int32_t*)t);
    return (int)hipGetLastError();
}
