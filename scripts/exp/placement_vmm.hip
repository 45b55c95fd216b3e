// Experiment (not product), round 3: does WHERE / HOW the arena is allocated decide the N = 2^26 step time?
//
// Round 2 saw the same arena re-allocated run 256 <-> 287 us (bare) and 391 <-> 420 us (with returns) and stopped at
// "which physical pages".  This harness allocates the product's streams three ways, several times each with a
// perturbing allocation in between, and times fishing_step_f32 (fishing-v1, returns) on each:
//   malloc    one hipMalloc for the whole arena (what torch's caching allocator hands the env)
//   separate  one hipMalloc per stream
//   vmm       hipMemAddressReserve + hipMemCreate + hipMemMap at the RECOMMENDED granularity (the VMM API)
//   vmm_min   the same at the minimum granularity
// One JSON line per trial.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 scripts/exp/placement_vmm.hip -Lgym_fishing_amd/_lib -lfishing_hip
//         -Wl,-rpath,'$ORIGIN/../../../gym_fishing_amd/_lib' -o scripts/exp/_build/placement_vmm
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fishing_hip.h"

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            std::exit(2);                                                                  \
        }                                                                                  \
    } while (0)

struct Arena {
    std::string mode;
    char* base = nullptr;
    size_t bytes = 0;
    std::vector<void*> parts;                       // separate
    hipMemGenericAllocationHandle_t handle{};       // vmm
    size_t mapped = 0;
};

static size_t g_gran_min = 0, g_gran_rec = 0;

static hipMemAllocationProp vmm_prop() {
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    return prop;
}

static void arena_alloc(Arena& a, const std::string& mode, size_t bytes) {
    a.mode = mode;
    a.bytes = bytes;
    if (mode == "malloc" || mode == "separate") {       // (separate: the caller carves streams out of parts instead)
        CK(hipMalloc((void**)&a.base, bytes));
        return;
    }
    const size_t gran = mode == "vmm" ? g_gran_rec : g_gran_min;
    a.mapped = (bytes + gran - 1) / gran * gran;
    hipMemAllocationProp prop = vmm_prop();
    CK(hipMemAddressReserve((void**)&a.base, a.mapped, gran, nullptr, 0));
    CK(hipMemCreate(&a.handle, a.mapped, &prop, 0));
    CK(hipMemMap(a.base, a.mapped, 0, a.handle, 0));
    hipMemAccessDesc acc;
    std::memset(&acc, 0, sizeof acc);
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(a.base, a.mapped, &acc, 1));
}

static void arena_free(Arena& a) {
    if (a.mode == "malloc" || a.mode == "separate") {
        CK(hipFree(a.base));
    } else {
        CK(hipMemUnmap(a.base, a.mapped));
        CK(hipMemRelease(a.handle));
        CK(hipMemAddressFree(a.base, a.mapped));
    }
    for (void* p : a.parts) CK(hipFree(p));
    a.parts.clear();
    a.base = nullptr;
}

int main(int argc, char** argv) {
    const int ln = argc > 1 ? std::atoi(argv[1]) : 26;
    const int trials = argc > 2 ? std::atoi(argv[2]) : 4;
    const bool ret = argc > 3 ? std::atoi(argv[3]) != 0 : true;
    const int64_t n = 1ll << ln;
    {
        hipMemAllocationProp prop = vmm_prop();
        CK(hipMemGetAllocationGranularity(&g_gran_min, &prop, hipMemAllocationGranularityMinimum));
        CK(hipMemGetAllocationGranularity(&g_gran_rec, &prop, hipMemAllocationGranularityRecommended));
        std::printf("{\"granularity_min\": %zu, \"granularity_recommended\": %zu}\n", g_gran_min, g_gran_rec);
    }
    FishingParams p;
    std::memset(&p, 0, sizeof p);
    p.model = FISHING_MODEL_V1;
    p.Tmax = 100;
    p.flags = FISHING_FLAG_AUTO_RESET;
    p.r = 0.3;
    p.K = 1.0;
    p.sigma = 0.1;
    p.C = 0.5;
    p.x0 = 0.75;
    p.n_actions = 100;
    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double* partials;
    CK(hipMalloc((void**)&partials, fishing_partials_len() * sizeof(double)));
    CK(hipMemset(partials, 0, fishing_partials_len() * sizeof(double)));
    const int rows = 4;
    const size_t gap = 12288, stride = (size_t)n * 4 + gap;
    // streams: obs, reward, t, ep_return (4 B each), done (1 B), action rows
    const size_t total = stride * (4 + rows) + (size_t)n + gap;
    std::vector<float> ha((size_t)n);
    std::vector<void*> dummies;
    unsigned lcg = 12345u;
    const char* modes[] = {"malloc", "vmm", "separate", "vmm_min"};
    for (int t = 0; t < trials; ++t) {
        for (const char* mode : modes) {
            // perturb the address space / physical placement between trials
            lcg = lcg * 1664525u + 1013904223u;
            void* d = nullptr;
            CK(hipMalloc(&d, (size_t)((lcg >> 8) % 96 + 1) << 20));
            dummies.push_back(d);
            Arena a;
            char *obs, *rew, *tt, *er, *done, *acts;
            if (std::string(mode) == "separate") {
                a.mode = mode;
                void* q[6];
                const size_t sz[6] = {(size_t)n * 4, (size_t)n * 4, (size_t)n * 4, (size_t)n * 4, (size_t)n, stride * rows};
                for (int k = 0; k < 6; ++k) {
                    CK(hipMalloc(&q[k], sz[k]));
                    a.parts.push_back(q[k]);
                }
                CK(hipMalloc((void**)&a.base, 256));
                obs = (char*)q[0], rew = (char*)q[1], tt = (char*)q[2], er = (char*)q[3], done = (char*)q[4], acts = (char*)q[5];
            } else {
                arena_alloc(a, mode, total);
                obs = a.base, rew = a.base + stride, tt = a.base + 2 * stride, er = a.base + 3 * stride;
                done = a.base + 4 * stride, acts = a.base + 4 * stride + (size_t)n + gap;
            }
            for (int64_t i = 0; i < n; ++i) ha[i] = -0.25f;
            CK(hipMemcpy(obs, ha.data(), (size_t)n * 4, hipMemcpyHostToDevice));
            CK(hipMemset(tt, 0, (size_t)n * 4));
            CK(hipMemset(er, 0, (size_t)n * 4));
            for (int k = 0; k < rows; ++k) {
                for (int64_t i = 0; i < n; ++i) ha[i] = -1.0f + 2.0f * (float)((i * 2654435761u + k * 40503u) & 0xffff) / 65536.0f;
                CK(hipMemcpy(acts + k * stride, ha.data(), (size_t)n * 4, hipMemcpyHostToDevice));
            }
            FishingBuffers b;
            std::memset(&b, 0, sizeof b);
            b.obs = obs;
            b.reward = rew;
            b.done = (uint8_t*)done;
            b.t = (int32_t*)tt;
            if (ret) {
                b.ep_return = er;
                b.return_partials = partials;
            }
            uint64_t counter = 0;
            auto run = [&](int k) {
                for (int i = 0; i < k; ++i, ++counter) {
                    b.action = acts + (counter % rows) * stride;
                    const int rc = fishing_step_f32(&p, n, 0, &b, 1234u, counter, stream);
                    if (rc) {
                        std::fprintf(stderr, "fishing_step_f32 rc %d\n", rc);
                        std::exit(3);
                    }
                }
            };
            run(24);
            CK(hipStreamSynchronize(stream));
            double us[3];
            for (int rd = 0; rd < 3; ++rd) {
                CK(hipEventRecord(e0, stream));
                run(60);
                CK(hipEventRecord(e1, stream));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                us[rd] = ms * 1e3 / 60;
            }
            std::printf("{\"log2_n\": %d, \"trial\": %d, \"mode\": \"%s\", \"ret\": %s, \"us\": [%.1f, %.1f, %.1f], \"base\": \"%p\", "
                        "\"GBps\": %.0f}\n",
                        ln, t, mode, ret ? "true" : "false", us[0], us[1], us[2], (void*)obs, n * (ret ? 33.0 : 25.0) / us[2] / 1e3);
            std::fflush(stdout);
            arena_free(a);
        }
    }
    for (void* d : dummies) CK(hipFree(d));
    return 0;
}
