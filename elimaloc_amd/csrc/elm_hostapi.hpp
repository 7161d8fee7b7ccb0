// elm_hostapi.hpp -- internal entry points shared by the host translation units (not part of the C ABI): the node callback in
// elm_glue.cpp, the device groups in elm_multi.cpp.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/elimaloc_hip.h"

struct elm_group; // a lead context's group of per-device contexts inside one process (elm_multi.cpp)

namespace elm_host {
constexpr size_t kCbTableRows = 2000;                            // capacity of the IMU deskew tables (pcm.cpp:533-585 caps the queue likewise)
constexpr size_t kCbTableBytes = 65536;                          // 4 tables x 2000 doubles = 64 000 bytes, rounded up
void* callback_staging(elm_ctx* ctx, size_t bytes);
int callback_register(elm_ctx* ctx, const elm_map* map, const void* stage, const float* rel_time, size_t n, const elm_deskew_tables* tab,
                      double voxel_size, const double T0[16], const elm_reg_config* cfg, elm_reg_result* result, uint64_t* n_source,
                      int* unpackable);
// members of the structs elm_api.cpp defines, for elm_multi.cpp
elm_group*& ctx_group(elm_ctx* ctx);
int ctx_device(const elm_ctx* ctx);
void ctx_set_error(elm_ctx* ctx, const std::string& text);
std::vector<elm_map*>& map_replicas(elm_map* m);
elm_ctx* map_ctx(const elm_map* m);
std::vector<elm_scan*>& scan_shards(elm_scan* s);
elm_ctx* scan_ctx(const elm_scan* s);
void scan_set_total(elm_scan* s, size_t n_total);
} // namespace elm_host

// Device groups: N per-device contexts inside ONE process behind one lead context (elm_ctx_create_multi; SURVEY 8(b): the reference node is
// one process calling RunRegister).  The public entry points hand a call on the lead context to these; every rank's share runs on that
// rank's worker thread through the same public entry points on its own plain context.
namespace elm_multi {
bool in_worker(); // this thread is a group's worker: the entry points take their single-device path
void destroy(elm_group* g);
int set_work_counters(elm_ctx* lead, int enable);
int map_build(elm_ctx* lead, const float* xyz, size_t n, double voxel_size, int max_points_per_voxel, elm_map** out);
int map_call(elm_map* lead_map, int which, double arg); // 0 CalVoxelCovAll, 1 CalPointCovAll(arg), 2 the search index
int scan_upload(elm_ctx* lead, const float* xyz, size_t n, elm_scan** out);
int reg(elm_ctx* lead, const elm_map* map, const float* scan_xyz, size_t n, const double T0[16], const elm_reg_config* cfg, double T_out[16],
        int* is_success, double* fitness_score, double local_cov[36], elm_reg_result* result, elm_iter_trace* trace);
int reg_batch(elm_ctx* lead, const elm_map* map, elm_scan* const* scans, int count, const double* T0, const elm_reg_config* cfg, int slots,
              elm_reg_result* results, elm_iter_trace* trace); // slots = 0: the lockstep batch
} // namespace elm_multi
