// fishing_host.h -- host-side helpers shared by the translation units of libfishing_hip.so.
#pragma once
#include <stdint.h>

#include <string>

#include "../../include/fishing_hip.h"

namespace fishing {

static inline bool misaligned(const void* p) { return p && (((uintptr_t)p) & 15u); }

// argument checks common to step / reset / rollout (fishing_step.hip)
int check_common(const FishingParams* p, int64_t n, int64_t env_offset, const FishingBuffers* b);
// grid of the general step / rollout kernels
void launch_shape(const FishingParams* p, int64_t n, int& blocks, int& threads);
// kNoiseNone / kNoiseExt / kNoisePhilox for this request
int noise_mode(const FishingParams* p, const FishingBuffers* b);

}  // namespace fishing
