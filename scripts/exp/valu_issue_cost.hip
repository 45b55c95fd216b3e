// valu_issue_cost.hip -- what one VALU instruction costs a SIMD of gfx950, by kind: the constants behind "is this kernel
// VALU-bound?" (DESIGN section 5).  Every kernel runs ITER x 16 independent instructions of one kind per wave (16 accumulator
// chains, so latency is hidden inside a wave), on a grid that gives every SIMD of the chip W waves; cycles per instruction
// per SIMD = elapsed shader cycles (s_memtime inside the kernel, median over waves) * W_resident / (ITER * 16 * W_resident).
//   hipcc -O3 --offload-arch=gfx950 scripts/exp/valu_issue_cost.hip -o valu_issue_cost && ./valu_issue_cost
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 512;
enum Op { FMA_F32, ADD_F32, MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, BITOP3, XOR_B32, FMA_F64, MUL_F64, ADD_F64, EXP_F32, LOG_F32, RCP_F32, SQRT_F32,
          SIN_F32, CVT_F64_F32, CVT_F32_F64, LDEXP_F64, RCP_F64, CNDMASK, PK_FMA_F32, N_OPS };
const char* kNames[N_OPS] = {"v_fma_f32", "v_add_f32", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_bitop3_b32", "v_xor_b32", "v_fma_f64",
                             "v_mul_f64", "v_add_f64", "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_sqrt_f32", "v_sin_f32", "v_cvt_f64_f32",
                             "v_cvt_f32_f64", "v_ldexp_f64", "v_rcp_f64", "v_cndmask_b32", "v_pk_fma_f32"};

template <int OP>
__global__ void __launch_bounds__(256) k(uint64_t* __restrict__ cycles, float* __restrict__ sink, float seed) {
    float a[16];
    double d[16];
    uint32_t u[16];
    uint64_t w[16];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = seed + (float)(threadIdx.x + i);
        d[i] = (double)a[i];
        u[i] = (uint32_t)(threadIdx.x * 2654435761u + i);
        w[i] = u[i];
        p[i] = f2{a[i], a[i] + 1.0f};
    }
    const float c1 = seed * 0.5f + 0.999f, c2 = seed + 1e-3f;
    const double e1 = (double)c1, e2 = (double)c2;
    __builtin_amdgcn_s_barrier();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (OP == FMA_F32) a[i] = __builtin_fmaf(a[i], c1, c2);
            else if constexpr (OP == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
            else if constexpr (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(u[i]), "v"(0xD2511F53u) : "vcc");
            else if constexpr (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(0xD2511F53u));
            else if constexpr (OP == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(0xD2511F53u));
            else if constexpr (OP == BITOP3) u[i] = __builtin_amdgcn_bitop3_b32(u[i], (uint32_t)it, 0x9E3779B9u, 0x96);
            else if constexpr (OP == XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(0x9E3779B9u));
            else if constexpr (OP == FMA_F64) d[i] = __builtin_fma(d[i], e1, e2);
            else if constexpr (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e1));
            else if constexpr (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e2));
            else if constexpr (OP == EXP_F32) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            else if constexpr (OP == LOG_F32) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
            else if constexpr (OP == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            else if constexpr (OP == SQRT_F32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            else if constexpr (OP == SIN_F32) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
            else if constexpr (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
            else if constexpr (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
            else if constexpr (OP == LDEXP_F64) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(1));
            else if constexpr (OP == RCP_F64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
            else if constexpr (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(0x9E3779B9u) : "vcc");
            else if constexpr (OP == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(f2{c1, c1}), "v"(f2{c2, c2}));
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + (float)d[i] + (float)u[i] + (float)w[i] + p[i][0] + p[i][1];
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
int run_one(int waves_per_simd, uint64_t* dcyc, float* dsink, std::vector<uint64_t>& h) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * waves_per_simd;          // 256 threads = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, dcyc, dsink, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, dcyc, dsink, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int nw = blocks * 4;
    CHECK(hipMemcpy(h.data(), dcyc, nw * sizeof(uint64_t), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.begin() + nw);
    const double med = (double)h[nw / 2];
    // s_memtime ticks at the constant 100 MHz reference on this part: report wall-clock ns per instruction per SIMD too
    const double instr_per_simd = (double)ITER * 16 * waves_per_simd;
    printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"memtime_ticks_per_wave\": %.0f, \"kernel_us\": %.2f, \"ns_per_instr_per_simd\": %.3f, "
           "\"cycles_at_2p4GHz_per_instr_per_simd\": %.2f}\n",
           kNames[OP], waves_per_simd, med, ms * 1e3, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    return 0;
}

template <int OP>
int run_all(uint64_t* dcyc, float* dsink, std::vector<uint64_t>& h) {
    for (int w : {1, 2, 8})
        if (run_one<OP>(w, dcyc, dsink, h)) return 1;
    if constexpr (OP + 1 < N_OPS) return run_all<OP + 1>(dcyc, dsink, h);
    return 0;
}

int main() {
    uint64_t* dcyc;
    float* dsink;
    CHECK(hipMalloc(&dcyc, 1 << 20));
    CHECK(hipMalloc(&dsink, 64));
    std::vector<uint64_t> h(1 << 17);
    return run_all<0>(dcyc, dsink, h);
}
