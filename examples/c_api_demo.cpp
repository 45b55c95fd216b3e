// Using libfishing_hip.so from plain C/C++ -- no Python, no torch: hipMalloc'd buffers, one
// stream, fishing-v1 with 1<<20 envs, 101 steps of a constant action at sigma = 0, then the
// episodic-return record.  Prints the known answer of SURVEY.md A.4 (return 6.3125 in 101 steps).
//
//   hipcc -O2 --offload-arch=gfx950 -Iinclude examples/c_api_demo.cpp \
//         -Lgym_fishing_amd/_lib -lfishing_hip -Wl,-rpath,$PWD/gym_fishing_amd/_lib -o /tmp/c_api_demo
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "fishing_hip.h"

#define CHECK(x)                                                        \
    do {                                                                \
        int rc_ = (x);                                                  \
        if (rc_ != 0) {                                                 \
            std::printf("%s failed: %s\n", #x, fishing_error_string(rc_)); \
            return 1;                                                   \
        }                                                               \
    } while (0)

int main() {
    const int64_t n = 1 << 20;
    FishingParams p;
    std::memset(&p, 0, sizeof p);
    p.model = FISHING_MODEL_V1;
    p.Tmax = 100;
    p.flags = FISHING_FLAG_AUTO_RESET;
    p.r = 0.3;
    p.K = 1.0;
    p.sigma = 0.0;
    p.C = 0.5;
    p.x0 = 0.75;

    float *obs, *action, *reward, *ep_return;
    uint8_t* done;
    int32_t* t;
    double *partials, *record;
    hipMalloc(&obs, n * 4);
    hipMalloc(&action, n * 4);
    hipMalloc(&reward, n * 4);
    hipMalloc(&ep_return, n * 4);
    hipMalloc(&done, n);
    hipMalloc(&t, n * 4);
    const int64_t slots = fishing_partials_slots(n);      // what a batch of n envs can touch (4 doubles per slot)
    hipMalloc(&partials, slots * 4 * sizeof(double));
    hipMalloc(&record, 4 * sizeof(double));
    hipMemset(partials, 0, slots * 4 * sizeof(double));
    std::vector<float> a(n, -0.9375f);
    hipMemcpy(action, a.data(), n * 4, hipMemcpyHostToDevice);

    FishingBuffers b;
    std::memset(&b, 0, sizeof b);
    b.obs = obs;
    b.action = action;
    b.reward = reward;
    b.done = done;
    b.t = t;
    b.ep_return = ep_return;
    b.return_partials = partials;

    hipStream_t stream;
    hipStreamCreate(&stream);
    CHECK(fishing_reset_f32(&p, n, 0, &b, nullptr, /*seed*/ 0, /*reset_counter*/ 0, stream));
    for (uint64_t s = 0; s < 101; ++s) CHECK(fishing_step_f32(&p, n, 0, &b, 0, s, stream));
    CHECK(fishing_reduce_returns_slots(partials, slots, record, stream));
    hipStreamSynchronize(stream);
    double rec[4];
    hipMemcpy(rec, record, sizeof rec, hipMemcpyDeviceToHost);
    std::printf("episodes %.0f  mean return %.6f  mean length %.1f\n", rec[2], rec[0] / rec[2], rec[3] / rec[2]);
    return (rec[2] == (double)n && rec[0] / rec[2] == 6.3125 && rec[3] / rec[2] == 101.0) ? 0 : 2;
}
