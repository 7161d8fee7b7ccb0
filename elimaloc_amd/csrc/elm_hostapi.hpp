// elm_hostapi.hpp -- internal entry points of elm_api.cpp used by the node callback in elm_glue.cpp (not part of the C ABI).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/elimaloc_hip.h"

namespace elm_host {
constexpr size_t kCbTableRows = 2000;                            // capacity of the IMU deskew tables (pcm.cpp:533-585 caps the queue likewise)
constexpr size_t kCbTableBytes = 65536;                          // 4 tables x 2000 doubles = 64 000 bytes, rounded up
void* callback_staging(elm_ctx* ctx, size_t bytes);
int callback_register(elm_ctx* ctx, const elm_map* map, const void* stage, const float* rel_time, size_t n, const elm_deskew_tables* tab,
                      double voxel_size, const double T0[16], const elm_reg_config* cfg, elm_reg_result* result, uint64_t* n_source,
                      int* unpackable);
} // namespace elm_host
