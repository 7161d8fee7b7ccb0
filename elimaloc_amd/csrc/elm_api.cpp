// elm_api.cpp -- C ABI (include/elimaloc_hip.h) over the HIP kernels: context, device-resident voxel map, scans,
// the per-iteration launch sequence of RunRegister, deskew, and the RCCL exchange.  Host-side C++17.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "elm_hostapi.hpp"
#include "elm_internal.hpp"
#include "elm_la.hpp"

using namespace elm;

// ------------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct elm_scan {
    elm_ctx* ctx = nullptr;
    uint64_t ctx_id = 0; // the owning context's unique id (a context allocated later at the same address is not the owner)
    Pt3* d_pts = nullptr;
    size_t cap_bytes = 0;
    uint32_t n = 0, n_total = 0;
    std::vector<elm_scan*> shards; // a scan uploaded through a device GROUP (elm_ctx_create_multi): this handle is rank 0's shard, these
                                   // are the shards of ranks 1 .. N-1 (elm_multi.cpp)
};

typedef int (*nccl_get_unique_id_t)(void*);
typedef int (*nccl_comm_init_rank_t)(void**, int, struct elm_nccl_id, int);
typedef int (*nccl_all_reduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_comm_destroy_t)(void*);
typedef const char* (*nccl_get_error_string_t)(int);
typedef int (*nccl_comm_query_t)(const void*, int*);
struct elm_nccl_id {
    char internal[ELM_COMM_ID_BYTES];
};

struct RcclApi {
    void* handle = nullptr;
    nccl_get_unique_id_t get_unique_id = nullptr;
    nccl_comm_init_rank_t comm_init_rank = nullptr;
    nccl_all_reduce_t all_reduce = nullptr;
    nccl_comm_destroy_t comm_destroy = nullptr;
    nccl_get_error_string_t get_error_string = nullptr;
    nccl_comm_query_t comm_count = nullptr, comm_user_rank = nullptr;
};
static RcclApi g_rccl;

static bool load_rccl(std::string* err) {
    if (g_rccl.handle) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        if (err) *err = std::string("cannot dlopen librccl: ") + dlerror();
        return false;
    }
    g_rccl.get_unique_id = (nccl_get_unique_id_t)dlsym(h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (nccl_comm_init_rank_t)dlsym(h, "ncclCommInitRank");
    g_rccl.all_reduce = (nccl_all_reduce_t)dlsym(h, "ncclAllReduce");
    g_rccl.comm_destroy = (nccl_comm_destroy_t)dlsym(h, "ncclCommDestroy");
    g_rccl.get_error_string = (nccl_get_error_string_t)dlsym(h, "ncclGetErrorString");
    g_rccl.comm_count = (nccl_comm_query_t)dlsym(h, "ncclCommCount");
    g_rccl.comm_user_rank = (nccl_comm_query_t)dlsym(h, "ncclCommUserRank");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_reduce || !g_rccl.comm_destroy) {
        if (err) *err = "librccl lacks ncclGetUniqueId/ncclCommInitRank/ncclAllReduce/ncclCommDestroy";
        return false;
    }
    g_rccl.handle = h;
    return true;
}

struct elm_ctx {
    int device = 0;
    uint64_t id = 0; // unique for the life of the process (children compare ids, not pointers)
    hipStream_t stream = nullptr;
    // host-fed streams / ordered uploads: uploads (DMA) and the scan-ordering kernel run beside the iterations on the compute stream
    hipStream_t copy_stream = nullptr, order_stream = nullptr, poll_stream = nullptr;
    hipEvent_t ev_iter[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> ev_groups; // host-fed streams: per upload group its "copied" and "ordered" events
    uint16_t* d_hilbert = nullptr; // Hilbert index of every cell of the ordering grid (kOrderCells^2 entries)
    DevBuf d_order_jobs, d_order_tmp, d_arena, d_raw, d_flagged, d_asym;
    DevBuf d_q0, d_q1, d_q2, d_q3, d_q4, d_q5; // scratch of elm_map_get_correspondences / elm_align_clouds_local (kept between calls)
    bool work_counters = false; // elm_ctx_set_work_counters: the accumulate launches also sum the work counters of
                                // elm_reg_result (n_cand_total, n_occ_total, n_tested_total, fallback_blocks); off: those fields read 0
    void* h_jobs = nullptr; // pinned: ordering job descriptors
    size_t h_jobs_cap = 0;
    std::string last_error;
    // scratch (grow-only; no allocation on the per-scan path after warm-up)
    DevBuf d_scans, d_state, d_partials, d_sums, d_T0, d_trace, d_stage_pts, d_active, d_queue, d_ds;
    int stream_hint_count = 0, stream_hint_slots = 0, stream_hint_iters = 0; // iterations the last elm_register_stream call of that shape needed
    int* h_active = nullptr; // pinned: number of scans still iterating, read back at the early-stop checks
    int iter_hint = 0;       // iterations the longest of the last eight batches needed (0 = unknown): first early-stop check happens there
    int iter_ring[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned iter_ring_pos = 0;
    const void* iter_key_map = nullptr; // the history belongs to ONE kind of call: (map, method, batch size); another kind starts afresh
    int iter_key_method = -1, iter_key_batch = 0;
    void* h_state = nullptr; // pinned
    size_t h_state_cap = 0;
    void* h_trace = nullptr; // pinned
    size_t h_trace_cap = 0;
    void* h_stage = nullptr; // pinned staging for synchronous uploads (scan points, deskew tables)
    size_t h_stage_cap = 0;
    std::vector<uint32_t> h_key, h_ord, h_tmp; // scratch of the scan ordering (grow-only)
    std::vector<struct elm_scan*> scan_free;   // recycled scan handles
    std::vector<std::pair<void*, size_t>> scan_pool; // device buffers of destroyed scans, reused by the next upload (a scan per
                                                     // LiDAR message: no hipMalloc / hipFree on the per-scan path after warm-up)
    void* h_desc = nullptr; // pinned staging of the batch descriptors (read by an async copy)
    size_t h_desc_cap = 0;
    // in-flight batch
    int batch = 0;
    bool want_trace = false;
    bool in_flight = false;
    void* ds_clean_ptr = nullptr;     // the downsample hash table at this address ...
    unsigned ds_clean_cap_log2 = 0;   // ... of this size is all ones (every pass cleans up the slots it touched)
    bool results_ready = false; // the last early-stop check of the in-flight batch found every scan finished: h_state holds the final states
    RegParams rp{};
    int scan_order = 1;  // 1: order uploaded scans along a Hilbert curve (elm_scan_upload), 0: keep the caller's order (ELM_SCAN_ORDER=none)
    int kernel_mode = 4; // accumulate kernels: 4 = dense cell grid, or cell-indexed neighbourhood lists when the grid does not fit
                         // (P2P/GICP), voxel-mean lists (VGICP/AVGICP): the default; 3 = neighbourhood lists forced (ELM_KERNEL=lists);
                         // 2 = the plain 27-probe walk of k_accumulate_direct (ELM_KERNEL=direct: in-kernel reference for tests)
    // optional hipEvent timing
    bool profiling = false;
    bool keep_iter_ms = false;         // prof_collect also keeps every accumulate span of the call (b_debug_print: the reference's
    std::vector<double> iter_acc_ms;   // per-iteration "Total Correspondence Time for" lines)
    std::vector<hipEvent_t> events;
    int events_used = 0;
    elm_profile prof{};
    // exchange
    void* comm = nullptr;
    int rank = 0, nranks = 1;
    elm_allreduce_fn hook = nullptr;
    void* hook_user = nullptr;
    int path = 0; // ELM_PATH_* of the registration call in progress (elm_reg_result.path)
    elm_group* group = nullptr; // elm_ctx_create_multi: this context LEADS a group of per-device contexts inside this process (elm_multi.cpp);
                                // maps built and scans uploaded through it are replicated / sharded over the group, registrations run on all
};

// Contexts that are alive.  Maps and scans hold a pointer to their context; destroying the context first is legal (e.g. Python
// object finalisation order): a child destroyed later finds its context gone and only releases its own device memory.
// A child remembers the context's unique id: a NEW context that happens to be allocated at a destroyed context's address (possibly on
// another device) is not its owner.
static std::mutex g_live_mu;
static std::map<const elm_ctx*, uint64_t> g_live_ctx;
static uint64_t g_next_ctx_id = 1;
static bool ctx_alive(const elm_ctx* ctx, uint64_t id) {
    std::lock_guard<std::mutex> lk(g_live_mu);
    auto it = g_live_ctx.find(ctx);
    return it != g_live_ctx.end() && it->second == id;
}

// A group's lead context seen from the caller's thread: the public entry points hand the call to elm_multi (which runs it on every
// rank's own context from that rank's worker thread -- where the same entry points take their plain single-device path).
static inline bool group_call(const elm_ctx* ctx) { return ctx && ctx->group && !elm_multi::in_worker(); }

#define HIPCHK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_);                     \
            return ELM_ERR_DEVICE;                                                                     \
        }                                                                                              \
    } while (0)

static int dev_reserve(elm_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return ELM_OK;
    if (b.p) HIPCHK(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t cap = std::max<size_t>(bytes, 256);
    HIPCHK(ctx, hipMalloc(&b.p, cap));
    b.cap = cap;
    return ELM_OK;
}
// Host memory a large temporary may take: MemAvailable of /proc/meminfo (free + reclaimable page cache -- after a big PCD has been read
// most of the RAM is page cache, and MemFree alone would refuse the grid depending on the cache state); unknown: no limit (the
// allocations themselves fail with bad_alloc, which the builders handle).
// The library's run-time switches are FIVE environment variables (include/elimaloc_hip.h lists them): ELM_KERNEL, ELM_GRID, ELM_CHECK,
// ELM_SCAN_ORDER, ELM_GROUP_EXCHANGE.  ELM_GRID and ELM_CHECK hold comma-separated tokens, `name` or `name=value`.
static bool env_token(const char* var, const char* name, const char** value = nullptr) {
    const char* e = getenv(var);
    if (!e) return false;
    const size_t ln = strlen(name);
    for (const char* p = e; *p;) {
        const char* q = strchr(p, ',');
        const size_t len = q ? (size_t)(q - p) : strlen(p);
        if (len >= ln && strncmp(p, name, ln) == 0 && (len == ln || p[ln] == '=')) {
            if (value) *value = (len > ln) ? p + ln + 1 : "";
            return true;
        }
        if (!q) break;
        p = q + 1;
    }
    return false;
}
static bool check_mode(const char* name) { return env_token("ELM_CHECK", name); } // the in-product checkers of the fast forms (tests)

// The block array of the cell grid: 4 slots per block must number below 2^31 (the kernels carry slot numbers as ints); below
// grid_narrow_limit() bytes stage 1 uses 32-bit byte offsets, beyond it 16-byte units (ELM_GRID=max_block_bytes=N lowers the limit: tests
// force the wide form onto small maps).
constexpr uint64_t kGridMaxBlocks = 0x1FFFFFF0ull;
static uint64_t grid_narrow_limit() {
    const char* v = nullptr;
    if (env_token("ELM_GRID", "max_block_bytes", &v)) return std::min<uint64_t>(strtoull(v, nullptr, 10), 0xFFFFFF00ull);
    return 0xFFFFFF00ull;
}
static uint64_t host_available_bytes() {
    FILE* f = fopen("/proc/meminfo", "r");
    if (!f) return ~0ull;
    char line[256];
    uint64_t kb = 0;
    bool found = false;
    while (fgets(line, sizeof(line), f)) {
        unsigned long long v = 0;
        if (sscanf(line, "MemAvailable: %llu kB", &v) == 1) { kb = v; found = true; break; }
    }
    fclose(f);
    return found ? kb * 1024ull : ~0ull;
}
static int pinned_reserve(elm_ctx* ctx, void** p, size_t* cap, size_t bytes) {
    if (bytes <= *cap) return ELM_OK;
    if (*p) HIPCHK(ctx, hipHostFree(*p));
    *p = nullptr;
    *cap = 0;
    HIPCHK(ctx, hipHostMalloc(p, bytes, hipHostMallocDefault));
    *cap = bytes;
    return ELM_OK;
}

extern "C" const char* elm_strerror(int status) {
    switch (status) {
    case ELM_OK: return "ok";
    case ELM_ERR_INVALID: return "invalid argument";
    case ELM_ERR_DEVICE: return "HIP runtime error";
    case ELM_ERR_NO_DEVICE: return "no usable gfx950 device";
    case ELM_ERR_COMM: return "RCCL error";
    case ELM_ERR_UNSUPPORTED: return "unsupported configuration";
    case ELM_ERR_IO: return "file missing or unreadable";
    case ELM_ERR_ALLOC: return "host allocation failed";
    default: return "unknown status";
    }
}

extern "C" void elm_reg_config_default(elm_reg_config* c) {
    // config/localization.ini:80-105
    memset(c, 0, sizeof(*c));
    c->i_max_thread = 10;
    c->icp_method = ELM_GICP;
    c->voxel_search_method = 2;
    c->use_radar_cov = 0;
    c->max_iteration = 10;
    c->b_debug_print = 0;
    c->gicp_cov_search_dist = 0.4;
    c->max_search_dist = 5.0;
    c->lm_lambda = 0.5;
    c->icp_termination_threshold_m = 0.02;
    c->min_overlap_ratio = 0.4;
    c->max_fitness_score = 0.5;
    c->doppler_trans_lambda = 0.5;
    c->range_variance_m = 1.0;
    c->azimuth_variance_deg = 0.4;
    c->elevation_variance_deg = 0.4;
    for (int i = 0; i < 3; ++i) c->ego_to_lidar_rot[i * 4] = c->ego_to_imu_rot[i * 4] = 1.0;
}

extern "C" int elm_ctx_create(int device_id, elm_ctx** out) {
    if (!out) return ELM_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ELM_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= count) return ELM_ERR_INVALID;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return ELM_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ELM_ERR_NO_DEVICE; // kernels are built for gfx950 only
    elm_ctx* ctx = new elm_ctx();
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return ELM_ERR_DEVICE;
    }
    if (const char* k = getenv("ELM_KERNEL")) ctx->kernel_mode = (strcmp(k, "direct") == 0) ? 2 : (strcmp(k, "lists") == 0) ? 3 : 4;
    if (const char* o = getenv("ELM_SCAN_ORDER")) ctx->scan_order = (strcmp(o, "none") == 0) ? 0 : 1;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        ctx->id = g_next_ctx_id++;
        g_live_ctx[ctx] = ctx->id;
    }
    *out = ctx;
    return ELM_OK;
}

extern "C" void elm_ctx_destroy(elm_ctx* ctx) {
    if (!ctx) return;
    if (ctx->group && !elm_multi::in_worker()) { // the group first: its workers, communicators and the other ranks' contexts
        elm_group* g = ctx->group;
        elm_multi::destroy(g); // (clears ctx->group)
    }
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        if (!g_live_ctx.erase(ctx)) return; // not a live context (double destroy)
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm && g_rccl.comm_destroy) g_rccl.comm_destroy(ctx->comm);
    for (hipStream_t st : {ctx->copy_stream, ctx->order_stream, ctx->poll_stream})
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (hipEvent_t e : {ctx->ev_iter[0], ctx->ev_iter[1], ctx->ev_iter[2], ctx->ev_iter[3]})
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_groups) (void)hipEventDestroy(e);
    if (ctx->d_hilbert) (void)hipFree(ctx->d_hilbert);
    if (ctx->h_jobs) (void)hipHostFree(ctx->h_jobs);
    DevBuf* bufs[] = {&ctx->d_scans, &ctx->d_state, &ctx->d_partials, &ctx->d_sums, &ctx->d_T0, &ctx->d_trace, &ctx->d_stage_pts, &ctx->d_active, &ctx->d_queue, &ctx->d_ds,
                      &ctx->d_order_jobs, &ctx->d_order_tmp, &ctx->d_arena, &ctx->d_raw, &ctx->d_flagged, &ctx->d_asym, &ctx->d_q0, &ctx->d_q1, &ctx->d_q2, &ctx->d_q3, &ctx->d_q4, &ctx->d_q5};
    if (ctx->h_active) (void)hipHostFree(ctx->h_active);
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (auto& b : ctx->scan_pool) (void)hipFree(b.first);
    for (elm_scan* f : ctx->scan_free) delete f;
    if (ctx->h_state) (void)hipHostFree(ctx->h_state);
    if (ctx->h_trace) (void)hipHostFree(ctx->h_trace);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->h_desc) (void)hipHostFree(ctx->h_desc);
    for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" const char* elm_last_error(const elm_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }
extern "C" void* elm_ctx_stream(elm_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
extern "C" int elm_ctx_set_work_counters(elm_ctx* ctx, int enable) {
    if (!ctx) return ELM_ERR_INVALID;
    if (ctx->in_flight) return ELM_ERR_INVALID;
    if (group_call(ctx)) return elm_multi::set_work_counters(ctx, enable); // every rank: the counters ride in the exchanged record
    ctx->work_counters = enable != 0;
    return ELM_OK;
}

extern "C" int elm_ctx_set_profiling(elm_ctx* ctx, int enable) {
    if (!ctx) return ELM_ERR_INVALID;
    ctx->profiling = enable != 0;
    return ELM_OK;
}
extern "C" int elm_ctx_get_profile(elm_ctx* ctx, elm_profile* out, int reset) {
    if (!ctx || !out) return ELM_ERR_INVALID;
    *out = ctx->prof;
    if (reset) ctx->prof = elm_profile{};
    return ELM_OK;
}
static int prof_mark(elm_ctx* ctx) { // records the next pooled event on the context stream
    if (!ctx->profiling) return ELM_OK;
    if (ctx->events_used == (int)ctx->events.size()) {
        hipEvent_t e;
        HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableSystemFence)); // timing only: no system-scope fence between launches
        ctx->events.push_back(e);
    }
    HIPCHK(ctx, hipEventRecord(ctx->events[ctx->events_used++], ctx->stream));
    return ELM_OK;
}
extern "C" int elm_ctx_synchronize(elm_ctx* ctx) {
    if (!ctx) return ELM_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ELM_OK;
}

// ------------------------------------------------------------------------------------------------------
// map: host-side AddPoints (vhm.cpp:270-285) + upload
// ------------------------------------------------------------------------------------------------------
struct elm_map {
    elm_ctx* ctx = nullptr;
    uint64_t ctx_id = 0;
    std::vector<elm_map*> replicas; // built through a device group: the same map on the devices of ranks 1 .. N-1 (this one is rank 0's)
    elm_map_info info{};
    DevMap dm{};
    HashSlot* d_slots = nullptr;
    float4* d_pts = nullptr;
    uint2* d_ranges = nullptr;
    int32_t* d_keys = nullptr; // [n_vox][3] stored keys (for downloads)
    double *d_vox_mean = nullptr, *d_vox_cov = nullptr, *d_vox_cinv = nullptr, *d_vox_nk = nullptr;
    unsigned* d_bad = nullptr;     // covariances outside the compact form I + k n n^T (counted by the covariance kernels)
    double* d_grid_gicp8 = nullptr; // the compact GICP records in grid slot order
    double* d_pt_gicp = nullptr; // [n_pts][16]: mean, inverse covariance, fitness normal (DevMap::pt_gicp)
    double* d_pt_cov = nullptr;  // [n_pts][9]: the covariances themselves (Pointcloud() read-back only)
    HashSlot* d_qslots = nullptr;
    Pt3* d_nbr_pts = nullptr;
    uint32_t* d_nbr_idx = nullptr;
    uint16_t* d_nbr_cell_off = nullptr;
    HashSlot* d_vqslots = nullptr;
    uint32_t* d_vq_dense = nullptr;
    uint32_t* d_vqf_dense = nullptr;
    VoxRec* d_vox_rec = nullptr;   // [n_vox] (DevMap::vox_rec)
    VoxBlk* d_vnbr_blk = nullptr;
    VoxRec* d_vface = nullptr;
    GridBlk* d_grid_blk = nullptr;    // dense cell grid (DevMap::grid_*), the default P2P / GICP search index
    uint32_t* d_grid_idx = nullptr;
    uint32_t* d_grid_start = nullptr;
    uint2* d_grid_tiles = nullptr; // two-level grid only
    uint32_t* d_vox_stat = nullptr;
    double* d_grid_gicp = nullptr; // the GICP records in grid slot order (built with the grid / refreshed by CalPointCovAll)
    size_t grid_slots = 0;
    bool has_grid = false;
    bool want_gicp_compact = false; // the grid gets 64-byte GICP records (points outside the compact form are flagged and read pt_gicp)
    unsigned n_bad_pts = 0, n_bad_vox = 0; // covariances outside the compact form (diagnostics)
    unsigned n_asym_pts = 0, n_asym_vox = 0; // ... of which the stored inverse is not symmetric: such a map carries side records (choose_path)
    bool grid_refused = false; // the bounding box needs more cells than the budget: neighbourhood lists instead
    bool warned_slow_path = false; // choose_path has said that this map's covariance methods run the per-pair kernels
    bool has_vnbr = false; // voxel-mean lists (VGICP / AVGICP)
    bool has_vface = false, vface_refused = false; // AVGICP's face sublists (built at the first AVGICP call; refused: no dense table / too many records)
    bool has_cells = false; // lists sorted by half-voxel cell + offset tables (every list <= 1024 entries)
    bool has_nbr = false;
    std::vector<int32_t> h_keys;
    std::vector<uint2> h_ranges;
};

namespace {

struct HostTable { // open addressing: key -> voxel id, first-seen order
    std::vector<int32_t> kx, ky, kz;
    std::vector<int32_t> vid;
    uint32_t mask = 0;
    uint32_t used = 0;
    void init(uint32_t cap) {
        kx.assign(cap, 0); ky.assign(cap, 0); kz.assign(cap, 0);
        vid.assign(cap, -1);
        mask = cap - 1;
        used = 0;
    }
    void grow() {
        HostTable n;
        n.init((mask + 1) * 2);
        for (uint32_t i = 0; i <= mask; ++i)
            if (vid[i] >= 0) n.insert_known(kx[i], ky[i], kz[i], vid[i]);
        *this = std::move(n);
    }
    void insert_known(int32_t x, int32_t y, int32_t z, int32_t v) {
        uint32_t h = hash3(x, y, z) & mask;
        while (vid[h] >= 0) h = (h + 1) & mask;
        kx[h] = x; ky[h] = y; kz[h] = z; vid[h] = v;
        ++used;
    }
    // returns the voxel id, creating next_id when absent
    int32_t find_or_add(int32_t x, int32_t y, int32_t z, int32_t next_id, bool* added) {
        uint32_t h = hash3(x, y, z) & mask;
        while (vid[h] >= 0) {
            if (kx[h] == x && ky[h] == y && kz[h] == z) {
                *added = false;
                return vid[h];
            }
            h = (h + 1) & mask;
        }
        kx[h] = x; ky[h] = y; kz[h] = z; vid[h] = next_id;
        ++used;
        *added = true;
        return next_id;
    }
};

static uint32_t next_pow2(uint64_t v) {
    uint64_t p = 16;
    while (p < v) p <<= 1;
    return (uint32_t)p;
}

struct HostBuild {
    std::vector<int32_t> keys; // 3 per voxel (stored / trunc keys)
    std::vector<uint2> ranges; // per voxel (start, cnt) into pts
    std::vector<float4> pts;   // retained points, bucket order
};

// AddPoints (vhm.cpp:270-285) + VoxelBlock::AddPointWithSpacing (vhm.hpp:106-113).  Insertion into one voxel never
// depends on another voxel, so the serial order only matters inside a voxel: points are grouped per voxel in input
// order (stable counting sort) and every voxel replays its own insertions sequentially.
static void build_host(const float* xyz, size_t n, double voxel_size, int max_points, HostBuild& hb) {
    const double map_resolution = sqrt(voxel_size * voxel_size / max_points);
    std::vector<uint32_t> pvid(n);
    HostTable tab;
    tab.init(next_pow2(std::max<uint64_t>(1024, n / 4)));
    int32_t n_vox = 0;
    for (size_t i = 0; i < n; ++i) {
        // Voxel((point.pose / voxel_size_).cast<int>()): truncation toward zero (vhm.cpp:275)
        const int32_t kx = (int32_t)((double)xyz[3 * i] / voxel_size);
        const int32_t ky = (int32_t)((double)xyz[3 * i + 1] / voxel_size);
        const int32_t kz = (int32_t)((double)xyz[3 * i + 2] / voxel_size);
        if (tab.used * 2 >= tab.mask) tab.grow();
        bool added;
        const int32_t v = tab.find_or_add(kx, ky, kz, n_vox, &added);
        if (added) {
            hb.keys.push_back(kx); hb.keys.push_back(ky); hb.keys.push_back(kz);
            ++n_vox;
        }
        pvid[i] = (uint32_t)v;
    }
    // stable grouping of the input indices per voxel
    std::vector<uint64_t> off((size_t)n_vox + 1, 0);
    for (size_t i = 0; i < n; ++i) off[pvid[i] + 1]++;
    for (int32_t v = 0; v < n_vox; ++v) off[v + 1] += off[v];
    std::vector<uint32_t> order(n);
    {
        std::vector<uint64_t> cur(off.begin(), off.end() - 1);
        for (size_t i = 0; i < n; ++i) order[cur[pvid[i]]++] = (uint32_t)i;
    }
    std::vector<uint32_t>().swap(pvid);
    // per-voxel replay of the insertions
    std::vector<uint32_t> kept_cnt((size_t)n_vox, 0);
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    unsigned T = (n_vox > 4096) ? std::min(hw, 32u) : 1u;
    auto work = [&](int32_t vb, int32_t ve) {
        std::vector<uint32_t> kept;
        for (int32_t v = vb; v < ve; ++v) {
            kept.clear();
            for (uint64_t o = off[v]; o < off[v + 1]; ++o) {
                const uint32_t i = order[o];
                if (kept.empty()) { // map_.insert({voxel, VoxelBlock{{point}, ...}}): the first point is always kept
                    kept.push_back(i);
                    continue;
                }
                if (kept.size() >= (size_t)max_points) continue; // points.size() < num_points
                const double px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
                bool near = false;
                for (uint32_t k : kept) {
                    const double dx = (double)xyz[3 * k] - px, dy = (double)xyz[3 * k + 1] - py, dz = (double)xyz[3 * k + 2] - pz;
                    if (sqrt((dx * dx + dy * dy) + dz * dz) < map_resolution) { // (voxel_point.pose - point.pose).norm() < map_resolution
                        near = true;
                        break;
                    }
                }
                if (!near) kept.push_back(i);
            }
            kept_cnt[v] = (uint32_t)kept.size();
            for (size_t k = 0; k < kept.size(); ++k) order[off[v] + k] = kept[k]; // compact in place
        }
    };
    if (T <= 1) {
        work(0, n_vox);
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < T; ++t) {
            int32_t vb = (int32_t)((int64_t)n_vox * t / T), ve = (int32_t)((int64_t)n_vox * (t + 1) / T);
            pool.emplace_back(work, vb, ve);
        }
        for (auto& th : pool) th.join();
    }
    hb.ranges.resize(n_vox);
    uint64_t total = 0;
    for (int32_t v = 0; v < n_vox; ++v) {
        hb.ranges[v] = make_uint2((uint32_t)total, kept_cnt[v]);
        total += kept_cnt[v];
    }
    hb.pts.resize(total);
    for (int32_t v = 0; v < n_vox; ++v)
        for (uint32_t k = 0; k < kept_cnt[v]; ++k) {
            const uint32_t i = order[off[v] + k];
            hb.pts[hb.ranges[v].x + k] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 0.f);
        }
}

} // namespace

static void map_free(elm_map* m) {
    if (!m) return;
    for (elm_map* r : m->replicas) map_free(r);
    m->replicas.clear();
    if (ctx_alive(m->ctx, m->ctx_id)) (void)hipSetDevice(m->ctx->device); // a context destroyed first: just release the device memory
    void* ptrs[] = {m->d_grid_tiles, m->d_vox_nk, m->d_bad, m->d_grid_gicp8, m->d_slots, m->d_pts, m->d_ranges, m->d_keys, m->d_vox_mean, m->d_vox_cov, m->d_vox_cinv, m->d_pt_gicp, m->d_pt_cov, m->d_qslots, m->d_nbr_pts, m->d_nbr_idx, m->d_nbr_cell_off, m->d_vqslots, m->d_vq_dense, m->d_vqf_dense, m->d_vface, m->d_vox_rec, m->d_vnbr_blk,
                    m->d_grid_blk, m->d_grid_idx, m->d_grid_start, m->d_vox_stat, m->d_grid_gicp};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete m;
}

extern "C" int elm_map_build(elm_ctx* ctx, const float* xyz, size_t n, double voxel_size, int max_points_per_voxel,
                             elm_map** out) {
    if (!ctx || !out || (!xyz && n) || !(voxel_size > 0.0) || max_points_per_voxel <= 0 || n > 0xFFFFFFF0ull) return ELM_ERR_INVALID;
    *out = nullptr;
    if (group_call(ctx)) return elm_multi::map_build(ctx, xyz, n, voxel_size, max_points_per_voxel, out);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HostBuild hb;
    if (n) build_host(xyz, n, voxel_size, max_points_per_voxel, hb);
    elm_map* m = new elm_map();
    m->ctx = ctx;
    m->ctx_id = ctx->id;
    const uint32_t n_vox = (uint32_t)hb.ranges.size();
    const uint32_t n_pts = (uint32_t)hb.pts.size();
    // load factor <= 0.25: a probe for an absent key (most of a workgroup's box cells are empty) ends after ~1.4 slots
    const uint32_t cap = next_pow2((uint64_t)n_vox * 4);
    std::vector<HashSlot> slots(cap);
    for (auto& s : slots) {
        s.kx = s.ky = s.kz = 0;
        s.vid = -1;
        s.start = s.cnt = s.pad0 = s.pad1 = 0;
    }
    for (uint32_t v = 0; v < n_vox; ++v) {
        uint32_t h = hash3(hb.keys[3 * v], hb.keys[3 * v + 1], hb.keys[3 * v + 2]) & (cap - 1);
        while (slots[h].vid >= 0) h = (h + 1) & (cap - 1);
        slots[h].kx = hb.keys[3 * v]; slots[h].ky = hb.keys[3 * v + 1]; slots[h].kz = hb.keys[3 * v + 2];
        slots[h].vid = (int32_t)v;
        slots[h].start = hb.ranges[v].x;
        slots[h].cnt = hb.ranges[v].y;
    }
    size_t bytes = 0;
#define MAP_ALLOC(ptr, count, T)                                                        \
    do {                                                                                \
        size_t b_ = std::max<size_t>((size_t)(count) * sizeof(T), 256);                 \
        hipError_t e_ = hipMalloc((void**)&(ptr), b_);                                  \
        if (e_ != hipSuccess) {                                                         \
            ctx->last_error = std::string("hipMalloc(map): ") + hipGetErrorString(e_);  \
            map_free(m);                                                                \
            return ELM_ERR_DEVICE;                                                      \
        }                                                                               \
        bytes += b_;                                                                    \
    } while (0)
    MAP_ALLOC(m->d_slots, cap, HashSlot);
    MAP_ALLOC(m->d_pts, n_pts, float4);
    MAP_ALLOC(m->d_ranges, n_vox, uint2);
    MAP_ALLOC(m->d_keys, (size_t)n_vox * 3, int32_t);
    hipError_t e = hipMemcpy(m->d_slots, slots.data(), (size_t)cap * sizeof(HashSlot), hipMemcpyHostToDevice);
    if (e == hipSuccess && n_pts) e = hipMemcpy(m->d_pts, hb.pts.data(), (size_t)n_pts * sizeof(float4), hipMemcpyHostToDevice);
    if (e == hipSuccess && n_vox) e = hipMemcpy(m->d_ranges, hb.ranges.data(), (size_t)n_vox * sizeof(uint2), hipMemcpyHostToDevice);
    if (e == hipSuccess && n_vox) e = hipMemcpy(m->d_keys, hb.keys.data(), (size_t)n_vox * 3 * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        ctx->last_error = std::string("hipMemcpy(map): ") + hipGetErrorString(e);
        map_free(m);
        return ELM_ERR_DEVICE;
    }
    m->h_keys = std::move(hb.keys);
    m->h_ranges = std::move(hb.ranges);
    m->dm.slots = m->d_slots;
    m->dm.mask = cap - 1;
    m->dm.n_vox = n_vox;
    m->dm.n_pts = n_pts;
    m->dm.pts = m->d_pts;
    m->dm.voxel_size = voxel_size;
    {
        int e2 = 0;
        m->dm.inv_vs_exact = (frexp(voxel_size, &e2) == 0.5) ? 1.0 / voxel_size : 0.0;
    }
    m->info.n_input_points = n;
    m->info.n_points = n_pts;
    m->info.n_voxels = n_vox;
    m->info.hash_capacity = cap;
    m->info.voxel_size = voxel_size;
    m->info.max_points_per_voxel = max_points_per_voxel;
    m->info.device_bytes = bytes;
    *out = m;
    return ELM_OK;
}

extern "C" void elm_map_destroy(elm_map* m) { map_free(m); }

static bool full_records_forced() {
    return check_mode("full_records");
}
extern "C" int elm_map_cal_voxel_cov_all(elm_map* m) {
    if (!m) return ELM_ERR_INVALID;
    elm_ctx* ctx = m->ctx;
    if (!m->replicas.empty() && group_call(ctx)) return elm_multi::map_call(m, 0, 0.0);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!m->d_vox_mean) {
        HIPCHK(ctx, hipMalloc((void**)&m->d_vox_mean, std::max<size_t>((size_t)m->dm.n_vox * 3 * sizeof(double), 256)));
        HIPCHK(ctx, hipMalloc((void**)&m->d_vox_cov, std::max<size_t>((size_t)m->dm.n_vox * 9 * sizeof(double), 256)));
        HIPCHK(ctx, hipMalloc((void**)&m->d_vox_cinv, std::max<size_t>((size_t)m->dm.n_vox * 9 * sizeof(double), 256)));
        HIPCHK(ctx, hipMalloc((void**)&m->d_vox_nk, std::max<size_t>((size_t)m->dm.n_vox * 4 * sizeof(double), 256)));
        m->info.device_bytes += (size_t)m->dm.n_vox * 25 * sizeof(double);
    }
    if (!m->d_bad) HIPCHK(ctx, hipMalloc((void**)&m->d_bad, 256));
    unsigned bad2[2] = {0, 0}; // [0] flagged, [1] flagged with an asymmetric stored inverse
    unsigned& bad = bad2[0];
    if (m->dm.n_vox) {
        (void)hipGetLastError(); // drop stale errors of other libraries (RCCL probes peer devices)
        HIPCHK(ctx, hipMemsetAsync(m->d_bad, 0, 2 * sizeof(unsigned), ctx->stream));
        launch_voxel_cov(ctx->stream, m->dm, m->d_ranges, m->d_vox_mean, m->d_vox_cov, m->d_vox_cinv, m->d_vox_nk, m->d_bad);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(bad2, m->d_bad, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    m->dm.vox_mean = m->d_vox_mean;
    m->dm.vox_cov = m->d_vox_cov;
    m->dm.vox_cinv = m->d_vox_cinv;
    m->dm.vox_nk = m->d_vox_nk;
    // an inverse covariance of the form I + k n n^T (to 1e-10) is rebuilt by the pairs from the 64-byte list record; the `bad` voxels
    // outside that form (rank-deficient neighbourhood, U != V in its SVD) carry k = NaN and their pairs read the stored 3x3 inverse.
    // ELM_CHECK=full_records: every pair reads the stored inverses.
    m->n_bad_vox = bad;
    m->n_asym_vox = bad2[1];
    m->info.layout_flags = (m->info.layout_flags & ~256) | (bad2[1] ? 256 : 0);
    m->dm.vox_compact = full_records_forced() ? 0 : ((bad == 0 && !check_mode("pair_nine")) ? 2 : 1); // 2: no flagged voxel at all
    m->info.has_voxel_cov = 1;
    m->info.layout_flags = (m->info.layout_flags & ~(2 | 16)) | (m->dm.vox_compact ? 2 : 0) | (m->dm.vox_compact == 2 ? 16 : 0);
    return ELM_OK;
}

static int refresh_grid_gicp(elm_map* m);
extern "C" int elm_map_cal_point_cov_all(elm_map* m, double d_search_dist) {
    if (!m) return ELM_ERR_INVALID;
    elm_ctx* ctx = m->ctx;
    if (!m->replicas.empty() && group_call(ctx)) return elm_multi::map_call(m, 1, d_search_dist);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!m->d_pt_gicp) {
        HIPCHK(ctx, hipMalloc((void**)&m->d_pt_gicp, std::max<size_t>((size_t)m->dm.n_pts * 16 * sizeof(double), 256)));
        HIPCHK(ctx, hipMalloc((void**)&m->d_pt_cov, std::max<size_t>((size_t)m->dm.n_pts * 9 * sizeof(double), 256)));
        m->info.device_bytes += (size_t)m->dm.n_pts * 25 * sizeof(double);
    }
    if (!m->d_bad) HIPCHK(ctx, hipMalloc((void**)&m->d_bad, 256));
    unsigned bad2[2] = {0, 0}; // [0] flagged, [1] flagged with an asymmetric stored inverse
    unsigned& bad = bad2[0];
    if (m->dm.n_pts) {
        (void)hipGetLastError();
        HIPCHK(ctx, hipMemsetAsync(m->d_bad, 0, 2 * sizeof(unsigned), ctx->stream));
        launch_point_cov(ctx->stream, m->dm, d_search_dist * d_search_dist, m->d_pt_gicp, m->d_pt_cov, m->d_bad);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(bad2, m->d_bad, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    m->dm.pt_gicp = m->d_pt_gicp;
    m->dm.pt_cov = m->d_pt_cov;
    m->n_bad_pts = bad;
    m->n_asym_pts = bad2[1];
    m->info.layout_flags = (m->info.layout_flags & ~128) | (bad2[1] ? 128 : 0);
    m->want_gicp_compact = !full_records_forced(); // see elm_map_cal_voxel_cov_all: non-conforming points (k = NaN) read their full record
    m->info.has_point_cov = 1;
    return refresh_grid_gicp(m); // a grid built before the covariances (or a new search radius): regather
}

// query voxels = every floor key within +-1 of a stored (trunc) key that holds points, first-seen order
static void enumerate_query_keys(const elm_map* m, std::vector<int32_t>& qkeys) {
    const uint32_t n_vox = m->dm.n_vox;
    HostTable tab;
    tab.init(next_pow2(std::max<uint64_t>(1024, (uint64_t)n_vox * 8)));
    int32_t nq = 0;
    for (uint32_t v = 0; v < n_vox; ++v) {
        if (m->h_ranges[v].y == 0) continue;
        const int32_t kx = m->h_keys[3 * v], ky = m->h_keys[3 * v + 1], kz = m->h_keys[3 * v + 2];
        for (int dx = -1; dx <= 1; ++dx)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dz = -1; dz <= 1; ++dz) {
                    if (tab.used * 2 >= tab.mask) tab.grow();
                    bool added;
                    tab.find_or_add(kx + dx, ky + dy, kz + dz, nq, &added);
                    if (added) {
                        qkeys.push_back(kx + dx); qkeys.push_back(ky + dy); qkeys.push_back(kz + dz);
                        ++nq;
                    }
                }
    }
}

static uint64_t grid_max_cells();
// Voxel-mean lists for VGICP / AVGICP (DevMap::vnbr_blk, vox_rec): built lazily at the first VGICP / AVGICP registration, after
// CalVoxelCovAll.  The lists hold 16 bytes per slot (float32 mean + voxel id | position code); a voxel's float64 record is stored ONCE in
// vox_rec (round 6: rounds 2-5 copied the 64-byte record into every list, 27 x).  AVGICP's face sublists (whole records, <= 7 per query
// voxel) are built only when an AVGICP call asks for them (want_faces): a VGICP map does not pay their bytes.
static int build_voxel_neighbourhoods(elm_map* m, bool want_faces) {
    const bool need_lists = !m->has_vnbr;
    // faces exist only beside the dense floor-key table; a map whose lists were built without one never gets them
    const bool need_faces = want_faces && !m->has_vface && !m->vface_refused && (need_lists || m->d_vq_dense != nullptr);
    if (!need_lists && !need_faces) return ELM_OK;
    elm_ctx* ctx = m->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (m->dm.n_vox > kVidMask) { ctx->last_error = "voxel-mean lists: more than 2^26 voxels"; return ELM_ERR_UNSUPPORTED; }
    std::vector<int32_t> qkeys;
    enumerate_query_keys(m, qkeys);
    const uint32_t n_q = (uint32_t)(qkeys.size() / 3);
    int32_t* d_qkeys = nullptr;
    uint32_t *d_counts = nullptr, *d_nocc = nullptr, *d_off = nullptr, *d_fcnt = nullptr, *d_foff = nullptr;
    auto cleanup = [&]() {
        for (void* p : {(void*)d_qkeys, (void*)d_counts, (void*)d_nocc, (void*)d_off, (void*)d_fcnt, (void*)d_foff})
            if (p) (void)hipFree(p);
    };
#define VN_CHK(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            ctx->last_error = std::string(#call) + ": " + hipGetErrorString(e_);              \
            cleanup();                                                                        \
            return ELM_ERR_DEVICE;                                                            \
        }                                                                                     \
    } while (0)
    const size_t nq_alloc = std::max<size_t>(n_q, 1);
    VN_CHK(hipMalloc((void**)&d_qkeys, nq_alloc * 3 * sizeof(int32_t)));
    VN_CHK(hipMalloc((void**)&d_counts, nq_alloc * sizeof(uint32_t)));
    VN_CHK(hipMalloc((void**)&d_nocc, nq_alloc * sizeof(uint32_t)));
    VN_CHK(hipMalloc((void**)&d_off, nq_alloc * sizeof(uint32_t)));
    std::vector<uint32_t> nocc(n_q), offs(n_q);
    uint64_t total = 0;
    if (n_q) { // (deterministic: a later face build finds the offsets the lists were written with)
        VN_CHK(hipMemcpy(d_qkeys, qkeys.data(), (size_t)n_q * 3 * sizeof(int32_t), hipMemcpyHostToDevice));
        (void)hipGetLastError();
        launch_nbr_count(ctx->stream, m->dm, d_qkeys, n_q, d_counts, d_nocc);
        VN_CHK(hipGetLastError());
        VN_CHK(hipStreamSynchronize(ctx->stream));
        VN_CHK(hipMemcpy(nocc.data(), d_nocc, (size_t)n_q * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint32_t q = 0; q < n_q; ++q) { offs[q] = (uint32_t)total; total += (nocc[q] + 3u) & ~3u; } // list starts: multiples of four slots
        VN_CHK(hipMemcpy(d_off, offs.data(), (size_t)n_q * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (total >= (1ull << 32)) { ctx->last_error = "voxel-mean lists exceed 2^32 slots"; cleanup(); return ELM_ERR_UNSUPPORTED; }
    // the dense box of floor keys (when it fits the cell budget and the packed word holds the offsets): no hash probe in the kernel --
    // one 4-byte load at a computed, spatially coherent address
    int32_t klo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, khi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    for (uint32_t q = 0; q < n_q; ++q)
        for (int a = 0; a < 3; ++a) {
            klo[a] = std::min(klo[a], qkeys[3 * q + a]);
            khi[a] = std::max(khi[a], qkeys[3 * q + a]);
        }
    const int64_t vd[3] = {n_q ? (int64_t)khi[0] - klo[0] + 1 : 0, n_q ? (int64_t)khi[1] - klo[1] + 1 : 0, n_q ? (int64_t)khi[2] - klo[2] + 1 : 0};
    const uint64_t vcells = (uint64_t)vd[0] * (uint64_t)vd[1] * (uint64_t)vd[2];
    auto dense_index = [&](uint32_t q) {
        return ((uint64_t)(qkeys[3 * q] - klo[0]) * (uint64_t)vd[1] + (uint64_t)(qkeys[3 * q + 1] - klo[1])) * (uint64_t)vd[2] + (uint64_t)(qkeys[3 * q + 2] - klo[2]);
    };
    const bool dense_ok = n_q && total / 4 < (1ull << 27) && vcells <= grid_max_cells() && m->ctx->kernel_mode == 4;
    if (need_lists) {
        const size_t nb = (size_t)(total / 4) + 1; // + one block of padding slots at the end: the target of the filter's loads past a list's last block
        VN_CHK(hipMalloc((void**)&m->d_vnbr_blk, nb * sizeof(VoxBlk)));
        {
            VoxBlk padblk;
            for (int u = 0; u < 4; ++u) { padblk.g.x[u] = padblk.g.y[u] = padblk.g.z[u] = 1e18f; padblk.vc[u] = -1; }
            VN_CHK(hipMemcpy(m->d_vnbr_blk + (nb - 1), &padblk, sizeof(VoxBlk), hipMemcpyHostToDevice));
            m->dm.vnbr_pad_blk = (uint32_t)(nb - 1);
        }
        VN_CHK(hipMalloc((void**)&m->d_vox_rec, std::max<size_t>((size_t)m->dm.n_vox * sizeof(VoxRec), 256)));
        (void)hipGetLastError();
        launch_vox_rec_fill(ctx->stream, m->dm, m->d_vox_rec);
        if (n_q) launch_vnbr_fill(ctx->stream, m->dm, d_qkeys, n_q, d_off, m->d_vnbr_blk);
        VN_CHK(hipGetLastError());
        VN_CHK(hipStreamSynchronize(ctx->stream));
        const uint32_t qcap = next_pow2((uint64_t)n_q * 2);
        {
            std::vector<HashSlot> qs(qcap);
            for (auto& e : qs) { e.kx = e.ky = e.kz = 0; e.vid = -1; e.start = e.cnt = e.pad0 = e.pad1 = 0; }
            for (uint32_t q = 0; q < n_q; ++q) {
                uint32_t h = hash3(qkeys[3 * q], qkeys[3 * q + 1], qkeys[3 * q + 2]) & (qcap - 1);
                while (qs[h].vid >= 0) h = (h + 1) & (qcap - 1);
                qs[h].kx = qkeys[3 * q]; qs[h].ky = qkeys[3 * q + 1]; qs[h].kz = qkeys[3 * q + 2];
                qs[h].vid = (int32_t)q;
                qs[h].start = offs[q]; qs[h].cnt = nocc[q];
            }
            VN_CHK(hipMalloc((void**)&m->d_vqslots, (size_t)qcap * sizeof(HashSlot)));
            VN_CHK(hipMemcpy(m->d_vqslots, qs.data(), (size_t)qcap * sizeof(HashSlot), hipMemcpyHostToDevice));
        }
        size_t bytes = nb * sizeof(VoxBlk) + (size_t)m->dm.n_vox * sizeof(VoxRec);
        if (dense_ok) {
            std::vector<uint32_t> dense(vcells, 0u);
            for (uint32_t q = 0; q < n_q; ++q) dense[dense_index(q)] = ((offs[q] >> 2) << 5) | nocc[q]; // first block of four slots, nocc <= 27
            VN_CHK(hipMalloc((void**)&m->d_vq_dense, vcells * sizeof(uint32_t)));
            VN_CHK(hipMemcpy(m->d_vq_dense, dense.data(), vcells * sizeof(uint32_t), hipMemcpyHostToDevice));
            m->dm.vq_dense = m->d_vq_dense;
            m->dm.vq_x0 = klo[0]; m->dm.vq_y0 = klo[1]; m->dm.vq_z0 = klo[2];
            m->dm.vq_nx = (int32_t)vd[0]; m->dm.vq_ny = (int32_t)vd[1]; m->dm.vq_nz = (int32_t)vd[2];
            bytes += vcells * sizeof(uint32_t);
        } else {
            bytes += (size_t)qcap * sizeof(HashSlot); // (the probe table is read only without the dense one)
        }
        m->dm.vqslots = m->d_vqslots;
        m->dm.vqmask = qcap - 1;
        m->dm.vox_rec = m->d_vox_rec;
        m->dm.vnbr_blk = m->d_vnbr_blk;
        m->has_vnbr = true;
        m->info.device_bytes += bytes + (dense_ok ? (size_t)qcap * sizeof(HashSlot) : 0);
        m->info.index_bytes += bytes + (size_t)m->n_bad_vox * 9 * sizeof(double); // + the stored inverses flagged voxels read
        m->info.index_part_bytes[1] += bytes + (size_t)m->n_bad_vox * 9 * sizeof(double);
        m->info.n_list_voxels = n_q;
    }
    if (need_faces && m->d_vq_dense && n_q) {
        // AVGICP pairs with the face neighbours only (<= 7 of a list's <= 27 slots): their sublists as whole records, addressed like vq_dense
        VN_CHK(hipMalloc((void**)&d_fcnt, nq_alloc * sizeof(uint32_t)));
        VN_CHK(hipMalloc((void**)&d_foff, nq_alloc * sizeof(uint32_t)));
        (void)hipGetLastError();
        // the fused AVGICP walk's record format; flagged voxels (NaN normals) are left to its fix-up launch (ELM_CHECK=avg_inline: such maps
        // keep the nine-entry walk with its in-line fallback)
        // The fix-up launch pays while few workgroups meet a flagged record (round 4: +14 % with 0.3 % of the voxels flagged); when
        // flagged voxels are common -- sparse clutter: two or three points per voxel, rank-deficient -- nearly every workgroup is marked,
        // the second launch repeats the whole walk, and the in-line fallback is the cheaper form: by default the map decides at 1 % of
        // its voxels (ELM_CHECK=avg_fixup / avg_inline force either form).
        const bool fx_inline = check_mode("avg_inline"), fx_fixup = check_mode("avg_fixup"), fx_skip = check_mode("avg_skip");
        const bool fixup_ok = fx_inline ? false : ((fx_fixup || fx_skip) ? true : (uint64_t)m->n_bad_vox * 100ull <= (uint64_t)m->dm.n_vox);
        const int plain = (m->dm.vox_compact && (m->n_bad_vox == 0 || fixup_ok) && !check_mode("avg_nine") && !check_mode("pair_nine")) ? 1 : 0;
        launch_vface(ctx->stream, m->dm, d_off, d_nocc, n_q, d_fcnt, nullptr, nullptr, plain);
        VN_CHK(hipGetLastError());
        VN_CHK(hipStreamSynchronize(ctx->stream));
        std::vector<uint32_t> fcnt(n_q), foff(n_q);
        VN_CHK(hipMemcpy(fcnt.data(), d_fcnt, (size_t)n_q * sizeof(uint32_t), hipMemcpyDeviceToHost));
        uint64_t ftotal = 0;
        for (uint32_t q = 0; q < n_q; ++q) { foff[q] = (uint32_t)ftotal; ftotal += fcnt[q]; }
        if (ftotal < (1ull << 29)) {
            VN_CHK(hipMemcpy(d_foff, foff.data(), (size_t)n_q * sizeof(uint32_t), hipMemcpyHostToDevice));
            VN_CHK(hipMalloc((void**)&m->d_vface, std::max<size_t>((size_t)ftotal * sizeof(VoxRec), 256)));
            launch_vface(ctx->stream, m->dm, d_off, d_nocc, n_q, d_fcnt, d_foff, m->d_vface, plain);
            VN_CHK(hipGetLastError());
            VN_CHK(hipStreamSynchronize(ctx->stream));
            std::vector<uint32_t> dense(vcells, 0u);
            for (uint32_t q = 0; q < n_q; ++q) dense[dense_index(q)] = (foff[q] << 3) | fcnt[q]; // fcnt <= 7
            VN_CHK(hipMalloc((void**)&m->d_vqf_dense, vcells * sizeof(uint32_t)));
            VN_CHK(hipMemcpy(m->d_vqf_dense, dense.data(), vcells * sizeof(uint32_t), hipMemcpyHostToDevice));
            m->dm.vface = m->d_vface;
            m->dm.vqf_dense = m->d_vqf_dense;
            m->dm.vface_plain = plain;
            m->dm.vface_flagged = (plain && m->n_bad_vox != 0) ? (fx_skip ? 2 : 1) : 0; // (2: tests only -- no fix-up launch, the flagged pairs are dropped)
            m->info.layout_flags = (m->info.layout_flags & ~(32 | 64)) | (plain ? 32 : 0) | (m->dm.vface_flagged ? 64 : 0);
            m->info.device_bytes += vcells * sizeof(uint32_t) + (size_t)ftotal * sizeof(VoxRec);
            m->info.index_bytes += vcells * sizeof(uint32_t) + (size_t)ftotal * sizeof(VoxRec);
            m->info.index_part_bytes[2] += vcells * sizeof(uint32_t) + (size_t)ftotal * sizeof(VoxRec);
            m->has_vface = true;
        } else {
            m->vface_refused = true;
        }
    } else if (need_faces) {
        m->vface_refused = true;
    }
#undef VN_CHK
    cleanup();
    return ELM_OK;
}

// The GICP payload in grid slot order (DevMap::grid_gicp): needs the grid and the point covariances, whichever comes last.
static int refresh_grid_gicp(elm_map* m) {
    if (!m->has_grid || !m->info.has_point_cov || !m->grid_slots) return ELM_OK;
    elm_ctx* ctx = m->ctx;
    const bool compact = m->want_gicp_compact;
    double** dst = compact ? &m->d_grid_gicp8 : &m->d_grid_gicp;
    const size_t rec_bytes = (compact ? 8 : 16) * sizeof(double);
    if (!*dst) {
        HIPCHK(ctx, hipMalloc((void**)dst, m->grid_slots * rec_bytes));
        m->info.device_bytes += m->grid_slots * rec_bytes;
    }
    (void)hipGetLastError();
    launch_gather_gicp(ctx->stream, m->dm, m->grid_slots, *dst, compact ? 1 : 0);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    m->dm.grid_gicp = m->d_grid_gicp;
    m->dm.grid_gicp8 = m->d_grid_gicp8;
    // 2: no point outside the compact form -- the kernel has no full-record fallback and gathers the pair fused (ELM_CHECK=pair_nine: the
    // nine-entry form with its fallback, as for maps with flagged points; fusing the compact lanes of THOSE kernels too and reading a
    // flagged point's stored inverse row by row measured 15 % slower in round 5: profiles/r05_kernel_ab.txt)
    m->dm.gicp_compact = compact ? ((m->n_bad_pts == 0 && !check_mode("pair_nine")) ? 2 : 1) : 0;
    m->info.layout_flags = (m->info.layout_flags & ~(1 | 8)) | (compact ? 1 : 0) | (m->dm.gicp_compact == 2 ? 8 : 0);
    return ELM_OK;
}

// Dense half-voxel cell grid (see DevMap::grid_*): the map points once, counting-sorted by cell on the host (init time), the
// offsets of every cell of the bounding box (+ 2 cells of margin), and the dense voxel box of walk statistics.
// ELM_ERR_UNSUPPORTED (and grid_refused) when the grid is not affordable -- the box needs more than max_cells cells, its tables
// do not fit the host / device memory that is free right now, or an allocation fails half-way -- the caller then builds the
// neighbourhood lists instead; nothing is left allocated and the map is unchanged.
static int build_cell_grid_impl(elm_map* m, uint64_t max_cells);
static int build_cell_grid(elm_map* m, uint64_t max_cells) {
    if (m->has_grid) return ELM_OK;
    try {
        return build_cell_grid_impl(m, max_cells);
    } catch (const std::bad_alloc&) { // a host table did not fit after all: same answer as the byte budget below
        m->ctx->last_error = "cell grid: host allocation failed";
        m->grid_refused = true;
        return ELM_ERR_UNSUPPORTED;
    }
}
static int build_cell_grid_impl(elm_map* m, uint64_t max_cells) {
    elm_ctx* ctx = m->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = m->dm.n_pts;
    const double vs = m->dm.voxel_size;
    if (n == 0 || m->dm.n_vox == 0) return ELM_ERR_UNSUPPORTED;
    std::vector<float4> pts(n);
    HIPCHK(ctx, hipMemcpy(pts.data(), m->d_pts, n * sizeof(float4), hipMemcpyDeviceToHost));
    int32_t lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    for (size_t i = 0; i < n; ++i) {
        const float c3[3] = {pts[i].x, pts[i].y, pts[i].z};
        for (int a = 0; a < 3; ++a) {
            const int32_t c = grid_cell_of((double)c3[a], vs);
            lo[a] = std::min(lo[a], c);
            hi[a] = std::max(hi[a], c);
        }
    }
    int64_t dim[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] -= 2; hi[a] += 2; // a block or ball that reaches just past the outermost points stays inside the grid
        dim[a] = (int64_t)hi[a] - lo[a] + 1;
    }
    const uint64_t cells = (uint64_t)dim[0] * (uint64_t)dim[1] * (uint64_t)dim[2];
    if (dim[0] > 0x7FFFFFF0ll || dim[1] > 0x7FFFFFF0ll || dim[2] > 0x7FFFFFF0ll || cells > max_cells || cells > 0xFFFFFFF0ull) {
        m->grid_refused = true;
        return ELM_ERR_UNSUPPORTED;
    }
    // dense voxel box of floor keys: a query with floor key f walks the stored keys f-1 .. f+1
    int32_t klo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, khi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    for (uint32_t v = 0; v < m->dm.n_vox; ++v)
        for (int a = 0; a < 3; ++a) {
            klo[a] = std::min(klo[a], m->h_keys[3 * v + a]);
            khi[a] = std::max(khi[a], m->h_keys[3 * v + a]);
        }
    const int64_t vd[3] = {(int64_t)khi[0] - klo[0] + 3, (int64_t)khi[1] - klo[1] + 3, (int64_t)khi[2] - klo[2] + 3};
    const uint64_t vcells = (uint64_t)vd[0] * (uint64_t)vd[1] * (uint64_t)vd[2];
    if (vcells > max_cells) {
        m->grid_refused = true;
        return ELM_ERR_UNSUPPORTED;
    }
    // Byte budget (the cell count alone says nothing about a sparse or elongated map just under it): host transients = the
    // offsets + the sorted copy (blocks of four: at most one block per point) + two index arrays; device = offsets + statistics +
    // blocks + slot indices (+ the GICP records in slot order).  Not affordable right now -> the lists.
    {
        const uint64_t worst_blk = std::min<uint64_t>(n, cells) + n / 4 + 2;
        const uint64_t host_need = (cells + 4) * 4 + worst_blk * (sizeof(GridBlk) + 16) + n * 8;
        const uint64_t dev_need = (cells + 4) * 4 + vcells * 4 + worst_blk * (sizeof(GridBlk) + 16) + (m->info.has_point_cov ? worst_blk * 4 * (m->want_gicp_compact ? 64 : 128) : 0);
        size_t dev_free = 0, dev_total = 0;
        HIPCHK(ctx, hipMemGetInfo(&dev_free, &dev_total));
        const uint64_t host_free = host_available_bytes();
        if (host_need > host_free / 10 * 7 || dev_need > (uint64_t)dev_free / 10 * 8) {
            ctx->last_error = "cell grid: tables exceed the memory that is free (host " + std::to_string(host_need >> 20) + " MB, device " +
                              std::to_string(dev_need >> 20) + " MB needed)";
            m->grid_refused = true;
            return ELM_ERR_UNSUPPORTED;
        }
    }
    // points per cell -> stable order of the points by cell (perm) -> blocks of four per cell.  ONE table of `cells` entries: it
    // holds the counts, then the running cursors of the scatter, then -- rewritten cell by cell while the blocks are laid out --
    // the first block of every cell (start[c + 1] = its end)
    std::vector<uint32_t> start(cells + 4, 0u), lin(n);
    for (size_t i = 0; i < n; ++i) {
        const uint64_t l = ((uint64_t)(grid_cell_of((double)pts[i].x, vs) - lo[0]) * (uint64_t)dim[1] + (uint64_t)(grid_cell_of((double)pts[i].y, vs) - lo[1])) * (uint64_t)dim[2] +
                           (uint64_t)(grid_cell_of((double)pts[i].z, vs) - lo[2]);
        lin[i] = (uint32_t)l;
        start[l + 1]++;
    }
    uint64_t n_blk = 1; // block 0: four padding slots, read by masked-off loads
    for (uint64_t c = 0; c < cells; ++c) n_blk += (start[c + 1] + 3) / 4;
    // stage 1 addresses blocks by 32-bit BYTE offsets (one scalar base + a lane offset) while the array stays below 4 GB (~275 M map
    // points), by 32-bit offsets in 16-byte units beyond (template flag WIDE: one 64-bit shift-add per block); slot numbers are ints
    if (n_blk > kGridMaxBlocks) { m->grid_refused = true; return ELM_ERR_UNSUPPORTED; }
    const bool wide_blocks = n_blk * sizeof(GridBlk) > grid_narrow_limit();
    for (uint64_t c = 0; c < cells; ++c) start[c + 1] += start[c]; // start[c] = first position of cell c in the (unpadded) sorted order
    std::vector<uint32_t> perm(n);
    for (size_t i = 0; i < n; ++i) perm[start[lin[i]]++] = (uint32_t)i; // bucket order in, so a cell keeps its points in bucket (= insertion) order
    std::vector<uint32_t>().swap(lin);
    // the cursors now hold every cell's END: start[c] = end(c) = begin(c + 1)
    std::vector<GridBlk> gb(std::max<uint64_t>(n_blk, 1));
    std::vector<uint32_t> gi(std::max<uint64_t>(4 * n_blk, 4), 0xFFFFFFFFu);
    for (auto& b : gb)
        for (int u = 0; u < 4; ++u) b.x[u] = b.y[u] = b.z[u] = 1e18f; // padding: never the nearest, never within range
    {
        uint64_t blk = 1;
        uint32_t begin = 0;
        for (uint64_t c = 0; c < cells; ++c) {
            const uint32_t end = start[c];
            start[c] = (uint32_t)blk;
            for (uint32_t k = begin; k < end; ++k) {
                const uint32_t i = perm[k];
                const uint64_t pos = blk * 4 + (k - begin);
                gb[pos >> 2].x[pos & 3] = pts[i].x; gb[pos >> 2].y[pos & 3] = pts[i].y; gb[pos >> 2].z[pos & 3] = pts[i].z;
                gi[pos] = i;
            }
            blk += (end - begin + 3) / 4;
            begin = end;
        }
        for (uint64_t c = cells; c < cells + 4; ++c) start[c] = (uint32_t)blk;
    }
    std::vector<uint32_t>().swap(perm);
    std::vector<float4>().swap(pts);
    // device side: into locals, committed to the map only when every step has succeeded
    GridBlk* d_blk = nullptr;
    uint32_t *d_idx = nullptr, *d_start = nullptr, *d_stat = nullptr;
    const DevMap dm_before = m->dm;
    auto fail = [&](const std::string& what, hipError_t e) {
        ctx->last_error = what + ": " + hipGetErrorString(e);
        for (void* q : {(void*)d_blk, (void*)d_idx, (void*)d_start, (void*)d_stat})
            if (q) (void)hipFree(q);
        m->dm = dm_before;
        (void)hipGetLastError();
        m->grid_refused = true; // do not retry a multi-GB build at every registration
        return e == hipErrorOutOfMemory ? ELM_ERR_UNSUPPORTED : ELM_ERR_DEVICE;
    };
#define GRID_CHK(call)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) return fail(#call, e_);                                         \
    } while (0)
    GRID_CHK(hipMalloc((void**)&d_blk, gb.size() * sizeof(GridBlk)));
    GRID_CHK(hipMalloc((void**)&d_idx, gi.size() * sizeof(uint32_t)));
    GRID_CHK(hipMalloc((void**)&d_start, (cells + 4) * sizeof(uint32_t)));
    GRID_CHK(hipMalloc((void**)&d_stat, std::max<size_t>(vcells * sizeof(uint32_t), 256)));
    GRID_CHK(hipMemcpy(d_blk, gb.data(), gb.size() * sizeof(GridBlk), hipMemcpyHostToDevice));
    GRID_CHK(hipMemcpy(d_idx, gi.data(), gi.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    GRID_CHK(hipMemcpy(d_start, start.data(), (cells + 4) * sizeof(uint32_t), hipMemcpyHostToDevice));
    m->dm.grid_blk = d_blk;
    m->dm.grid_idx = d_idx;
    m->dm.grid_start = d_start;
    m->dm.grid_wide = wide_blocks ? 1 : 0;
    m->dm.grid_nslots = (uint32_t)(4 * n_blk);
    m->dm.gx0 = lo[0]; m->dm.gy0 = lo[1]; m->dm.gz0 = lo[2];
    m->dm.gnx = (int32_t)dim[0]; m->dm.gny = (int32_t)dim[1]; m->dm.gnz = (int32_t)dim[2];
    m->dm.vox_stat = d_stat;
    m->dm.vx0 = klo[0] - 1; m->dm.vy0 = klo[1] - 1; m->dm.vz0 = klo[2] - 1;
    m->dm.vnx = (int32_t)vd[0]; m->dm.vny = (int32_t)vd[1]; m->dm.vnz = (int32_t)vd[2];
    (void)hipGetLastError();
    launch_vox_stat(ctx->stream, m->dm, d_stat);
    GRID_CHK(hipGetLastError());
    GRID_CHK(hipStreamSynchronize(ctx->stream));
#undef GRID_CHK
    m->d_grid_blk = d_blk; m->d_grid_idx = d_idx; m->d_grid_start = d_start; m->d_vox_stat = d_stat;
    m->has_grid = true;
    m->grid_slots = gi.size();
    {
        const int rc2 = refresh_grid_gicp(m);
        if (rc2 != ELM_OK) { // the payload copy did not fit: give the grid back, the lists read pt_gicp through their index
            (void)hipFree(d_blk); (void)hipFree(d_idx); (void)hipFree(d_start); (void)hipFree(d_stat);
            m->d_grid_blk = nullptr; m->d_grid_idx = nullptr; m->d_grid_start = nullptr; m->d_vox_stat = nullptr;
            m->dm = dm_before;
            m->has_grid = false;
            m->grid_slots = 0;
            m->grid_refused = true;
            (void)hipGetLastError();
            return ELM_ERR_UNSUPPORTED;
        }
    }
    m->info.device_bytes += gb.size() * sizeof(GridBlk) + gi.size() * sizeof(uint32_t) + (cells + 4) * sizeof(uint32_t) + vcells * sizeof(uint32_t);
    m->info.n_query_voxels = vcells;
    m->info.nbr_entries = n;
    m->info.index_bytes += gb.size() * sizeof(GridBlk) + (cells + 4) * sizeof(uint32_t) + vcells * sizeof(uint32_t);
    m->info.index_part_bytes[0] += gb.size() * sizeof(GridBlk) + (cells + 4) * sizeof(uint32_t); // (the statistics box is read by the instrumented kernels only)
    return ELM_OK;
}

// The two-level form of the cell grid (DevMap::grid_tiles): for maps whose bounding box is too large or too sparse for one dense
// offset table (a city-scale map at half-metre cells: billions of cells, a few hundred thousand of them occupied).  The same
// blocks, sorted (tile, column, z); offsets only inside occupied tiles and only over each tile's own z range.  Host-side build
// like the dense one; nothing is left allocated on failure.
static int build_tiled_grid_impl(elm_map* m);
static int build_tiled_grid(elm_map* m) {
    if (m->has_grid) return ELM_OK;
    try {
        return build_tiled_grid_impl(m);
    } catch (const std::bad_alloc&) {
        m->ctx->last_error = "tiled cell grid: host allocation failed";
        return ELM_ERR_UNSUPPORTED;
    }
}
static int build_tiled_grid_impl(elm_map* m) {
    elm_ctx* ctx = m->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = m->dm.n_pts;
    const double vs = m->dm.voxel_size;
    if (n == 0 || m->dm.n_vox == 0) return ELM_ERR_UNSUPPORTED;
    std::vector<float4> pts(n);
    HIPCHK(ctx, hipMemcpy(pts.data(), m->d_pts, n * sizeof(float4), hipMemcpyDeviceToHost));
    int32_t lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    for (size_t i = 0; i < n; ++i) {
        const float c3[3] = {pts[i].x, pts[i].y, pts[i].z};
        for (int a = 0; a < 3; ++a) {
            const int32_t c = grid_cell_of((double)c3[a], vs);
            lo[a] = std::min(lo[a], c);
            hi[a] = std::max(hi[a], c);
        }
    }
    int64_t dim[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] -= 2; hi[a] += 2;
        dim[a] = (int64_t)hi[a] - lo[a] + 1;
    }
    const int64_t tnx = (dim[0] + kTile - 1) / kTile, tny = (dim[1] + kTile - 1) / kTile;
    // 64 M tiles = 512 MB of tile table (a 32 km x 32 km box at half-metre cells); the z range of a tile is a 16-bit pair
    if (tnx * tny > ((int64_t)64 << 20) || dim[2] > 65000 || tnx * kTile > 0x7FFFFFF0ll || tny * kTile > 0x7FFFFFF0ll) return ELM_ERR_UNSUPPORTED;
    const size_t n_tiles = (size_t)(tnx * tny);
    auto cell3 = [&](size_t i, int32_t& cx, int32_t& cy, int32_t& cz) {
        cx = grid_cell_of((double)pts[i].x, vs) - lo[0];
        cy = grid_cell_of((double)pts[i].y, vs) - lo[1];
        cz = grid_cell_of((double)pts[i].z, vs) - lo[2];
    };
    std::vector<uint16_t> zmin(n_tiles, 0xFFFFu), zmax(n_tiles, 0u);
    for (size_t i = 0; i < n; ++i) {
        int32_t cx, cy, cz;
        cell3(i, cx, cy, cz);
        const size_t t = (size_t)(cx >> kTileShift) * (size_t)tny + (size_t)(cy >> kTileShift);
        zmin[t] = std::min<uint16_t>(zmin[t], (uint16_t)cz);
        zmax[t] = std::max<uint16_t>(zmax[t], (uint16_t)cz);
    }
    std::vector<uint2> tiles(n_tiles);
    uint64_t entries = (uint64_t)kTile * kTile; // entries 0 .. 63: the shared zero run of every empty tile (nz = 0: one entry per column)
    for (size_t t = 0; t < n_tiles; ++t) {
        if (zmin[t] == 0xFFFFu) { tiles[t] = make_uint2(0u, 0u); continue; }
        const uint32_t nz = (uint32_t)zmax[t] - zmin[t] + 1;
        tiles[t] = make_uint2((uint32_t)entries, (uint32_t)zmin[t] | (nz << 16));
        entries += (uint64_t)kTile * kTile * (nz + 1);
        if (entries > 0xFFFFFFF0ull) return ELM_ERR_UNSUPPORTED;
    }
    auto entry_of = [&](int32_t cx, int32_t cy, int32_t cz) -> uint32_t {
        const uint2 te = tiles[(size_t)(cx >> kTileShift) * (size_t)tny + (size_t)(cy >> kTileShift)];
        const uint32_t z0 = te.y & 0xFFFFu, nz = te.y >> 16;
        return te.x + (uint32_t)(((cx & (kTile - 1)) << kTileShift) | (cy & (kTile - 1))) * (nz + 1) + ((uint32_t)cz - z0);
    };
    // byte budget, as for the dense grid
    {
        const uint64_t worst_blk = n + 2;
        const uint64_t host_need = (entries + 4) * 8 + worst_blk * (sizeof(GridBlk) + 16) + n * 8 + n_tiles * 12;
        const uint64_t dev_need = (entries + 4) * 4 + n_tiles * 8 + worst_blk * (sizeof(GridBlk) + 16) + (m->info.has_point_cov ? worst_blk * 4 * (m->want_gicp_compact ? 64 : 128) : 0);
        size_t dev_free = 0, dev_total = 0;
        HIPCHK(ctx, hipMemGetInfo(&dev_free, &dev_total));
        const uint64_t host_free = host_available_bytes();
        if (host_need > host_free / 10 * 7 || dev_need > (uint64_t)dev_free / 10 * 8) {
            ctx->last_error = "tiled cell grid: tables exceed the memory that is free";
            return ELM_ERR_UNSUPPORTED;
        }
    }
    std::vector<uint32_t> start(entries + 4, 0u), cursor(entries + 4, 0u), ent(n);
    for (size_t i = 0; i < n; ++i) {
        int32_t cx, cy, cz;
        cell3(i, cx, cy, cz);
        ent[i] = entry_of(cx, cy, cz);
        start[ent[i]]++; // counts first
    }
    // entry order = (tile, column, z): block starts; the extra entry of every column (index nz) = the column's end
    uint64_t n_blk = 1, pos = 0;
    for (uint64_t e = 0; e < entries; ++e) {
        const uint32_t cnt = start[e];
        start[e] = (uint32_t)n_blk;
        cursor[e] = (uint32_t)pos; // position in the (unpadded) sorted order
        n_blk += (cnt + 3) / 4;
        pos += cnt;
    }
    for (uint64_t e = 0; e < (uint64_t)kTile * kTile; ++e) start[e] = 0; // empty tiles: block 0 .. block 0 (nothing)
    for (uint64_t e = entries; e < entries + 4; ++e) start[e] = (uint32_t)n_blk;
    if (n_blk > kGridMaxBlocks) return ELM_ERR_UNSUPPORTED;
    const bool wide_blocks = n_blk * sizeof(GridBlk) > grid_narrow_limit();
    std::vector<GridBlk> gb(std::max<uint64_t>(n_blk, 1));
    std::vector<uint32_t> gi(std::max<uint64_t>(4 * n_blk, 4), 0xFFFFFFFFu);
    for (auto& b : gb)
        for (int u = 0; u < 4; ++u) b.x[u] = b.y[u] = b.z[u] = 1e18f;
    {
        std::vector<uint32_t> first(cursor); // where each entry's run begins in the sorted order
        for (size_t i = 0; i < n; ++i) { // bucket order in: a cell keeps its points in bucket (= insertion) order
            const uint32_t e = ent[i];
            const uint64_t slot = (uint64_t)start[e] * 4 + (cursor[e] - first[e]);
            cursor[e]++;
            gb[slot >> 2].x[slot & 3] = pts[i].x; gb[slot >> 2].y[slot & 3] = pts[i].y; gb[slot >> 2].z[slot & 3] = pts[i].z;
            gi[slot] = (uint32_t)i;
        }
    }
    std::vector<uint32_t>().swap(cursor);
    std::vector<uint32_t>().swap(ent);
    std::vector<float4>().swap(pts);
    GridBlk* d_blk = nullptr;
    uint32_t *d_idx = nullptr, *d_start = nullptr;
    uint2* d_tiles = nullptr;
    auto fail = [&](const std::string& what, hipError_t e) {
        ctx->last_error = what + ": " + hipGetErrorString(e);
        for (void* q : {(void*)d_blk, (void*)d_idx, (void*)d_start, (void*)d_tiles})
            if (q) (void)hipFree(q);
        (void)hipGetLastError();
        return e == hipErrorOutOfMemory ? ELM_ERR_UNSUPPORTED : ELM_ERR_DEVICE;
    };
#define TG_CHK(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) return fail(#call, e_);                                         \
    } while (0)
    TG_CHK(hipMalloc((void**)&d_blk, gb.size() * sizeof(GridBlk)));
    TG_CHK(hipMalloc((void**)&d_idx, gi.size() * sizeof(uint32_t)));
    TG_CHK(hipMalloc((void**)&d_start, (entries + 4) * sizeof(uint32_t)));
    TG_CHK(hipMalloc((void**)&d_tiles, n_tiles * sizeof(uint2)));
    TG_CHK(hipMemcpy(d_blk, gb.data(), gb.size() * sizeof(GridBlk), hipMemcpyHostToDevice));
    TG_CHK(hipMemcpy(d_idx, gi.data(), gi.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    TG_CHK(hipMemcpy(d_start, start.data(), (entries + 4) * sizeof(uint32_t), hipMemcpyHostToDevice));
    TG_CHK(hipMemcpy(d_tiles, tiles.data(), n_tiles * sizeof(uint2), hipMemcpyHostToDevice));
#undef TG_CHK
    const DevMap dm_before = m->dm;
    m->dm.grid_blk = d_blk; m->dm.grid_idx = d_idx; m->dm.grid_start = d_start; m->dm.grid_tiles = d_tiles;
    m->dm.grid_tiled = 1;
    m->dm.grid_wide = wide_blocks ? 1 : 0;
    m->dm.grid_nslots = (uint32_t)(4 * n_blk);
    m->dm.gtny = (int32_t)tny;
    m->dm.gx0 = lo[0]; m->dm.gy0 = lo[1]; m->dm.gz0 = lo[2];
    m->dm.gnx = (int32_t)(tnx * kTile); m->dm.gny = (int32_t)(tny * kTile); m->dm.gnz = (int32_t)dim[2];
    m->dm.vox_stat = nullptr;
    m->d_grid_blk = d_blk; m->d_grid_idx = d_idx; m->d_grid_start = d_start; m->d_grid_tiles = d_tiles;
    m->has_grid = true;
    m->grid_slots = gi.size();
    if (refresh_grid_gicp(m) != ELM_OK) {
        (void)hipFree(d_blk); (void)hipFree(d_idx); (void)hipFree(d_start); (void)hipFree(d_tiles);
        m->d_grid_blk = nullptr; m->d_grid_idx = nullptr; m->d_grid_start = nullptr; m->d_grid_tiles = nullptr;
        m->dm = dm_before;
        m->has_grid = false;
        m->grid_slots = 0;
        (void)hipGetLastError();
        return ELM_ERR_UNSUPPORTED;
    }
    const size_t idx_bytes = gb.size() * sizeof(GridBlk) + (entries + 4) * sizeof(uint32_t) + n_tiles * sizeof(uint2);
    m->info.device_bytes += idx_bytes + gi.size() * sizeof(uint32_t);
    m->info.n_query_voxels = n_tiles;
    m->info.layout_flags |= 4;
    m->info.nbr_entries = n;
    m->info.index_bytes += idx_bytes;
    m->info.index_part_bytes[0] += idx_bytes;
    return ELM_OK;
}

// Neighbourhood lists (see DevMap): query voxels = every floor key within +-1 of a stored (trunc) key.
static int build_neighbourhood_lists(elm_map* m) {
    if (!m) return ELM_ERR_INVALID;
    if (m->has_nbr) return ELM_OK;
    elm_ctx* ctx = m->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<int32_t> qkeys;
    enumerate_query_keys(m, qkeys);
    const uint32_t n_q = (uint32_t)(qkeys.size() / 3);
    int32_t* d_qkeys = nullptr;
    uint32_t *d_counts = nullptr, *d_nocc = nullptr, *d_off = nullptr;
    auto cleanup = [&]() {
        if (d_qkeys) (void)hipFree(d_qkeys);
        if (d_counts) (void)hipFree(d_counts);
        if (d_nocc) (void)hipFree(d_nocc);
        if (d_off) (void)hipFree(d_off);
    };
#define NBR_CHK(call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            ctx->last_error = std::string(#call) + ": " + hipGetErrorString(e_);              \
            cleanup();                                                                        \
            return ELM_ERR_DEVICE;                                                            \
        }                                                                                     \
    } while (0)
    const size_t nq_alloc = std::max<size_t>(n_q, 1);
    NBR_CHK(hipMalloc((void**)&d_qkeys, nq_alloc * 3 * sizeof(int32_t)));
    NBR_CHK(hipMalloc((void**)&d_counts, nq_alloc * sizeof(uint32_t)));
    NBR_CHK(hipMalloc((void**)&d_nocc, nq_alloc * sizeof(uint32_t)));
    NBR_CHK(hipMalloc((void**)&d_off, nq_alloc * sizeof(uint32_t)));
    std::vector<uint32_t> counts(n_q), nocc(n_q), offs(n_q);
    uint64_t total = 0;
    if (n_q) {
        NBR_CHK(hipMemcpy(d_qkeys, qkeys.data(), (size_t)n_q * 3 * sizeof(int32_t), hipMemcpyHostToDevice));
        (void)hipGetLastError();
        launch_nbr_count(ctx->stream, m->dm, d_qkeys, n_q, d_counts, d_nocc);
        NBR_CHK(hipGetLastError());
        NBR_CHK(hipStreamSynchronize(ctx->stream));
        NBR_CHK(hipMemcpy(counts.data(), d_counts, (size_t)n_q * sizeof(uint32_t), hipMemcpyDeviceToHost));
        NBR_CHK(hipMemcpy(nocc.data(), d_nocc, (size_t)n_q * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint32_t q = 0; q < n_q; ++q) {
            offs[q] = (uint32_t)total;
            total += counts[q];
        }
        if (total > 0xFFFFFFF0ull) {
            ctx->last_error = "neighbourhood lists exceed 2^32 entries";
            cleanup();
            return ELM_ERR_UNSUPPORTED;
        }
        NBR_CHK(hipMemcpy(d_off, offs.data(), (size_t)n_q * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    NBR_CHK(hipMalloc((void**)&m->d_nbr_pts, std::max<size_t>((size_t)total * sizeof(Pt3), 256) + 64)); // + pad: blocks of 4 are read whole
    NBR_CHK(hipMalloc((void**)&m->d_nbr_idx, std::max<size_t>((size_t)total * sizeof(uint32_t), 256)));
    if (n_q) {
        (void)hipGetLastError();
        launch_nbr_fill(ctx->stream, m->dm, d_qkeys, n_q, d_off, m->d_nbr_pts, m->d_nbr_idx);
        NBR_CHK(hipGetLastError());
        NBR_CHK(hipStreamSynchronize(ctx->stream));
    }
    // sort every list by half-voxel cell and write the per-list offset tables (cell-indexed kernel)
    uint32_t max_count = 0;
    for (uint32_t q = 0; q < n_q; ++q) max_count = std::max(max_count, counts[q]);
    const size_t cell_bytes = std::max<size_t>((size_t)n_q * nbr_cell_stride() * sizeof(uint16_t), 256);
    NBR_CHK(hipMalloc((void**)&m->d_nbr_cell_off, cell_bytes));
    NBR_CHK(hipMemsetAsync(m->d_nbr_cell_off, 0, cell_bytes, ctx->stream));
    m->has_cells = max_count <= 1024;
    if (n_q && m->has_cells) {
        (void)hipGetLastError();
        launch_nbr_cellsort(ctx->stream, m->dm, d_qkeys, n_q, d_off, d_counts, m->d_nbr_pts, m->d_nbr_idx, m->d_nbr_cell_off);
        NBR_CHK(hipGetLastError());
    }
    NBR_CHK(hipStreamSynchronize(ctx->stream));
    const uint32_t qcap = next_pow2((uint64_t)n_q * 2);
    {
        std::vector<HashSlot> qs(qcap);
        for (auto& e : qs) { e.kx = e.ky = e.kz = 0; e.vid = -1; e.start = e.cnt = e.pad0 = e.pad1 = 0; }
        for (uint32_t q = 0; q < n_q; ++q) {
            uint32_t h = hash3(qkeys[3 * q], qkeys[3 * q + 1], qkeys[3 * q + 2]) & (qcap - 1);
            while (qs[h].vid >= 0) h = (h + 1) & (qcap - 1);
            qs[h].kx = qkeys[3 * q]; qs[h].ky = qkeys[3 * q + 1]; qs[h].kz = qkeys[3 * q + 2];
            qs[h].vid = (int32_t)q;
            qs[h].start = offs[q]; qs[h].cnt = counts[q]; qs[h].pad0 = nocc[q];
        }
        NBR_CHK(hipMalloc((void**)&m->d_qslots, (size_t)qcap * sizeof(HashSlot)));
        NBR_CHK(hipMemcpy(m->d_qslots, qs.data(), (size_t)qcap * sizeof(HashSlot), hipMemcpyHostToDevice));
    }
#undef NBR_CHK
    cleanup();
    m->dm.qslots = m->d_qslots;
    m->dm.qmask = qcap - 1;
    m->dm.n_q = n_q;
    m->dm.nbr_pts = m->d_nbr_pts;
    m->dm.nbr_idx = m->d_nbr_idx;
    m->dm.nbr_cell_off = m->d_nbr_cell_off;
    m->has_nbr = true;
    m->info.device_bytes += (size_t)total * (sizeof(Pt3) + sizeof(uint32_t)) + (size_t)qcap * sizeof(HashSlot) + cell_bytes;
    m->info.n_query_voxels = n_q;
    m->info.nbr_entries = total;
    m->info.index_bytes += (size_t)total * sizeof(Pt3) + (size_t)qcap * sizeof(HashSlot) + cell_bytes;
    m->info.index_part_bytes[3] += (size_t)total * sizeof(Pt3) + (size_t)qcap * sizeof(HashSlot) + cell_bytes;
    return ELM_OK;
}

// The P2P / GICP search index of a map: the dense cell grid when its bounding box fits the cell budget (ELM_GRID=max_cells=N,
// default 1.5e9 = 6 GB of offsets), the per-query-voxel neighbourhood lists otherwise (or with ELM_KERNEL=lists).
static uint64_t grid_max_cells() {
    const char* v = nullptr;
    if (env_token("ELM_GRID", "max_cells", &v)) return strtoull(v, nullptr, 10);
    return 1500000000ull;
}
// The P2P / GICP search index of a map: the dense cell grid when its bounding box fits the cell and byte budgets, the two-level
// (tiled) grid when it does not -- large or sparse maps keep the grid kernel -- and the per-query-voxel neighbourhood lists only
// when neither can be built (or with ELM_KERNEL=lists; ELM_GRID=tiled forces the two-level form, ELM_GRID=dense forbids it).
static int build_search_index(elm_map* m, bool* use_grid) {
    *use_grid = false;
    elm_ctx* ctx = m->ctx;
    if (ctx->kernel_mode == 4 && !m->grid_refused && m->dm.n_pts) {
        const bool force_tiled = env_token("ELM_GRID", "tiled"), no_tiled = env_token("ELM_GRID", "dense");
        int rc = force_tiled ? ELM_ERR_UNSUPPORTED : build_cell_grid(m, grid_max_cells());
        if (rc == ELM_ERR_UNSUPPORTED && !no_tiled) {
            rc = build_tiled_grid(m);
            if (rc == ELM_ERR_UNSUPPORTED) m->grid_refused = true; // neither form: do not retry at every registration
            else if (rc == ELM_OK) m->grid_refused = false;
        }
        if (rc == ELM_OK) {
            *use_grid = true;
            return ELM_OK;
        }
        if (rc != ELM_ERR_UNSUPPORTED) return rc;
    }
    return build_neighbourhood_lists(m);
}
extern "C" int elm_map_build_neighbourhoods(elm_map* m) {
    if (!m) return ELM_ERR_INVALID;
    if (!m->replicas.empty() && group_call(m->ctx)) return elm_multi::map_call(m, 2, 0.0);
    bool g;
    return build_search_index(m, &g);
}

extern "C" int elm_map_get_info(const elm_map* m, elm_map_info* info) {
    if (!m || !info) return ELM_ERR_INVALID;
    *info = m->info;
    return ELM_OK;
}
extern "C" int elm_map_empty(const elm_map* m) { return (!m || m->dm.n_vox == 0) ? 1 : 0; }

static void rowmajor_to_colmajor3(const double* src, double* dst) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) dst[c * 3 + r] = src[r * 3 + c];
}

extern "C" int elm_map_download_points(const elm_map* m, double* xyz, double* cov9, double* mean3, size_t cap) {
    if (!m) return ELM_ERR_INVALID;
    elm_ctx* ctx = m->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = std::min<size_t>(cap, m->dm.n_pts);
    if (xyz && n) {
        std::vector<float4> tmp(n);
        HIPCHK(ctx, hipMemcpy(tmp.data(), m->d_pts, n * sizeof(float4), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) {
            xyz[3 * i] = tmp[i].x; xyz[3 * i + 1] = tmp[i].y; xyz[3 * i + 2] = tmp[i].z;
        }
    }
    if ((cov9 || mean3) && !m->info.has_point_cov) {
        // default CovStruct of every PointStruct: (I, 0) (vhm.hpp:45)
        for (size_t i = 0; i < n; ++i) {
            if (cov9) for (int k = 0; k < 9; ++k) cov9[9 * i + k] = (k % 4 == 0) ? 1.0 : 0.0;
            if (mean3) mean3[3 * i] = mean3[3 * i + 1] = mean3[3 * i + 2] = 0.0;
        }
        return ELM_OK;
    }
    if ((cov9 || mean3) && n) {
        std::vector<double> tmp(n * 16);
        HIPCHK(ctx, hipMemcpy(tmp.data(), m->d_pt_gicp, n * 16 * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i)
            if (mean3) memcpy(&mean3[3 * i], &tmp[16 * i], 3 * sizeof(double));
        if (cov9) {
            std::vector<double>().swap(tmp);
            tmp.resize(n * 9);
            HIPCHK(ctx, hipMemcpy(tmp.data(), m->d_pt_cov, n * 9 * sizeof(double), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < n; ++i) rowmajor_to_colmajor3(&tmp[9 * i], &cov9[9 * i]);
        }
    }
    return ELM_OK;
}

extern "C" int elm_map_download_voxels(const elm_map* m, int32_t* key3, int32_t* npts, double* cov9, double* mean3, size_t cap) {
    if (!m) return ELM_ERR_INVALID;
    elm_ctx* ctx = m->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = std::min<size_t>(cap, m->dm.n_vox);
    if (key3) memcpy(key3, m->h_keys.data(), n * 3 * sizeof(int32_t));
    if (npts) for (size_t v = 0; v < n; ++v) npts[v] = (int32_t)m->h_ranges[v].y;
    if ((cov9 || mean3) && !m->info.has_voxel_cov) {
        for (size_t i = 0; i < n; ++i) {
            if (cov9) for (int k = 0; k < 9; ++k) cov9[9 * i + k] = (k % 4 == 0) ? 1.0 : 0.0;
            if (mean3) mean3[3 * i] = mean3[3 * i + 1] = mean3[3 * i + 2] = 0.0;
        }
        return ELM_OK;
    }
    if (cov9 && n) {
        std::vector<double> tmp(n * 9);
        HIPCHK(ctx, hipMemcpy(tmp.data(), m->d_vox_cov, n * 9 * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) rowmajor_to_colmajor3(&tmp[9 * i], &cov9[9 * i]);
    }
    if (mean3 && n) HIPCHK(ctx, hipMemcpy(mean3, m->d_vox_mean, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return ELM_OK;
}

extern "C" int elm_map_find_ground_height(const elm_map* m, double x, double y, double* ground_z, int* found) {
    // vhm.hpp:285-322: mean z of the (up to) 5 lowest map points within 5 m in xy; needs more than 3 points
    if (!m || !ground_z || !found) return ELM_ERR_INVALID;
    elm_ctx* ctx = m->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    *found = 0;
    const size_t n = m->dm.n_pts;
    std::vector<float4> tmp(n);
    if (n) HIPCHK(ctx, hipMemcpy(tmp.data(), m->d_pts, n * sizeof(float4), hipMemcpyDeviceToHost));
    std::vector<double> zs;
    for (size_t i = 0; i < n; ++i) {
        const double dx = (double)tmp[i].x - x, dy = (double)tmp[i].y - y;
        if (dx * dx + dy * dy <= 25.0) zs.push_back((double)tmp[i].z);
    }
    if (zs.size() <= 3) return ELM_OK;
    const size_t N = std::min<size_t>(5, zs.size());
    std::partial_sort(zs.begin(), zs.begin() + N, zs.end());
    double s = 0.0;
    for (size_t i = 0; i < N; ++i) s += zs[i];
    *ground_z = s / (double)N;
    *found = 1;
    return ELM_OK;
}

// ------------------------------------------------------------------------------------------------------
// scans
// ------------------------------------------------------------------------------------------------------

// index of cell (x, y) along a Hilbert curve over a 2^order x 2^order grid
static inline uint32_t hilbert_xy2d(uint32_t order, uint32_t x, uint32_t y) {
    uint32_t d = 0;
    for (uint32_t s = 1u << (order - 1); s > 0; s >>= 1) {
        const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d += s * s * ((3u * rx) ^ ry);
        if (ry == 0) { // rotate the quadrant
            if (rx == 1) { x = s - 1 - x; y = s - 1 - y; }
            const uint32_t t = x; x = y; y = t;
        }
    }
    return d;
}

// The ordering grid's Hilbert table (kOrderCells^2 16-bit entries), uploaded once per context.
static int ensure_hilbert(elm_ctx* ctx) {
    if (ctx->d_hilbert) return ELM_OK;
    static_assert(kOrderCells == 64, "hilbert_xy2d order below");
    std::vector<uint16_t> lut((size_t)kOrderCells * kOrderCells);
    for (int y = 0; y < kOrderCells; ++y)
        for (int x = 0; x < kOrderCells; ++x) lut[(size_t)y * kOrderCells + x] = (uint16_t)hilbert_xy2d(6, (uint32_t)x, (uint32_t)y);
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_hilbert, lut.size() * sizeof(uint16_t)));
    HIPCHK(ctx, hipMemcpy(ctx->d_hilbert, lut.data(), lut.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return ELM_OK;
}

// A scan handle + a device buffer of at least n points, both from the context's pools when possible (a scan per LiDAR message: no
// hipMalloc / hipFree on the per-scan path after warm-up).
static int scan_alloc(elm_ctx* ctx, size_t n, elm_scan** out) {
    elm_scan* s;
    if (!ctx->scan_free.empty()) {
        s = ctx->scan_free.back();
        ctx->scan_free.pop_back();
        *s = elm_scan();
    } else {
        s = new (std::nothrow) elm_scan();
        if (!s) return ELM_ERR_ALLOC;
    }
    s->ctx = ctx;
    s->ctx_id = ctx->id;
    const size_t need = std::max<size_t>(n * sizeof(Pt3), 256);
    int best = -1; // smallest pooled buffer that fits
    for (int i = 0; i < (int)ctx->scan_pool.size(); ++i)
        if (ctx->scan_pool[i].second >= need && (best < 0 || ctx->scan_pool[i].second < ctx->scan_pool[best].second)) best = i;
    if (best >= 0) {
        s->d_pts = (Pt3*)ctx->scan_pool[best].first;
        s->cap_bytes = ctx->scan_pool[best].second;
        ctx->scan_pool.erase(ctx->scan_pool.begin() + best);
    } else {
        s->cap_bytes = (need + 65535) & ~(size_t)65535;
        const hipError_t e = hipMalloc((void**)&s->d_pts, s->cap_bytes);
        if (e != hipSuccess) {
            ctx->last_error = std::string("scan buffer allocation: ") + hipGetErrorString(e);
            delete s;
            return ELM_ERR_DEVICE;
        }
    }
    *out = s;
    return ELM_OK;
}

// Uploads one scan: the caller's packed float32 xyz (12 bytes per point) goes to HBM as it is -- no repack, no host-side sort.
// order = true: the points are then ordered ON THE DEVICE along a Hilbert curve over 2 m x 2 m sensor-frame cells (k_scan_order,
// ~0.1 ms per 131 072-point scan; the host radix sort of round 2 took 1.3 ms) -- consecutive points, hence every 256-point workgroup
// and, through the XCD-aware block mapping, every XCD's L2, touch a few adjacent map cells: +11..13 % registrations/s on resident
// scans.  order = false (elm_register): the caller's order; ordering one scan costs more than it saves on ONE registration.
// No allocation after warm-up: pinned staging, scan handles and device buffers are pooled in the context.
// Copy into page-locked staging memory that the CPU will not read again: non-temporal 32-byte stores (no read-for-ownership of the
// destination lines, no cache pollution) where the CPU has AVX2, memcpy otherwise.  dst must be 32-byte aligned (staging buffers are).
#include <immintrin.h>
__attribute__((target("avx2"))) static void stage_copy_avx2(void* dst, const void* src, size_t len) {
    char* d = (char*)dst;
    const char* s = (const char*)src;
    size_t i = 0;
    for (; i + 128 <= len; i += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i*)(s + i)), b = _mm256_loadu_si256((const __m256i*)(s + i + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i*)(s + i + 64)), e = _mm256_loadu_si256((const __m256i*)(s + i + 96));
        _mm256_stream_si256((__m256i*)(d + i), a);
        _mm256_stream_si256((__m256i*)(d + i + 32), b);
        _mm256_stream_si256((__m256i*)(d + i + 64), c);
        _mm256_stream_si256((__m256i*)(d + i + 96), e);
    }
    _mm_sfence();
    if (i < len) memcpy(d + i, s + i, len - i);
}
static void stage_copy(void* dst, const void* src, size_t len) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && (((uintptr_t)dst) & 31u) == 0) stage_copy_avx2(dst, src, len);
    else memcpy(dst, src, len);
}

static int scan_upload_impl(elm_ctx* ctx, const float* xyz, size_t n, size_t n_total, bool order, bool sync, elm_scan** out) {
    if (!ctx || !out || (!xyz && n) || n > 0x7FFFFFFFull || n_total > 0x7FFFFFFFull || n_total < n) return ELM_ERR_INVALID;
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t bytes = n * sizeof(Pt3);
    int rc = pinned_reserve(ctx, &ctx->h_stage, &ctx->h_stage_cap, std::max<size_t>(bytes + sizeof(OrderJob), 4096));
    if (rc != ELM_OK) return rc;
    elm_scan* s = nullptr;
    if ((rc = scan_alloc(ctx, n, &s)) != ELM_OK) return rc;
    s->n = (uint32_t)n;
    s->n_total = (uint32_t)n_total;
    const bool do_order = order && ctx->scan_order && n > 1;
    hipError_t e = hipSuccess;
    // A page-locked source (elm_host_alloc, hipHostMalloc, hipHostRegister) is read by the DMA engine where it lies.  A pageable one
    // goes through the context's pinned staging buffer in pieces, so that the copy of piece k + 1 overlaps the DMA of piece k
    // (the runtime's own staging of pageable memory is several times slower).
    bool pinned_src = false;
    if (n) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, xyz) == hipSuccess && attr.type == hipMemoryTypeHost) pinned_src = true;
        else (void)hipGetLastError();
    }
    auto upload = [&](void* dst) -> hipError_t {
        if (!n) return hipSuccess;
        if (pinned_src) return hipMemcpyAsync(dst, xyz, bytes, hipMemcpyHostToDevice, ctx->stream);
        const size_t piece = (size_t)384 << 10;
        hipError_t ee = hipSuccess;
        for (size_t o = 0; o < bytes && ee == hipSuccess; o += piece) {
            const size_t len = std::min(piece, bytes - o);
            stage_copy((char*)ctx->h_stage + o, (const char*)xyz + o, len); // (pieces are multiples of 32 bytes)
            ee = hipMemcpyAsync((char*)dst + o, (const char*)ctx->h_stage + o, len, hipMemcpyHostToDevice, ctx->stream);
        }
        return ee;
    };
    if (!do_order) {
        e = upload(s->d_pts);
    } else {
        rc = ensure_hilbert(ctx);
        if (rc == ELM_OK) rc = dev_reserve(ctx, ctx->d_raw, bytes);
        const bool wide = n >= 16384; // one scan on its own: many workgroups (same bytes, ~10x sooner)
        const size_t tmp_bytes = ((n * sizeof(uint32_t) + 255) / 256) * 256;
        if (rc == ELM_OK) rc = dev_reserve(ctx, ctx->d_order_tmp, tmp_bytes + (wide ? order_wide_scratch_bytes((unsigned)n) : 0));
        if (rc == ELM_OK) rc = dev_reserve(ctx, ctx->d_order_jobs, sizeof(OrderJob));
        if (rc != ELM_OK) { elm_scan_destroy(s); return rc; }
        OrderJob* hj = (OrderJob*)((char*)ctx->h_stage + ((bytes + 15) & ~(size_t)15));
        if ((char*)(hj + 1) > (char*)ctx->h_stage + ctx->h_stage_cap) { // keep the descriptor inside the staging buffer
            rc = pinned_reserve(ctx, &ctx->h_jobs, &ctx->h_jobs_cap, 4096);
            if (rc != ELM_OK) { elm_scan_destroy(s); return rc; }
            hj = (OrderJob*)ctx->h_jobs;
        }
        hj->src = (const Pt3*)ctx->d_raw.p; hj->dst = s->d_pts; hj->tmp = (uint32_t*)ctx->d_order_tmp.p; hj->n = (uint32_t)n; hj->_pad = 0;
        e = upload(ctx->d_raw.p);
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_order_jobs.p, hj, sizeof(OrderJob), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            (void)hipGetLastError();
            if (wide) launch_scan_order_wide(ctx->stream, (const OrderJob*)ctx->d_order_jobs.p, (unsigned)n, ctx->d_hilbert, (char*)ctx->d_order_tmp.p + tmp_bytes);
            else launch_scan_order(ctx->stream, (const OrderJob*)ctx->d_order_jobs.p, 1, ctx->d_hilbert);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess && sync) e = hipStreamSynchronize(ctx->stream); // the pinned staging buffer is reused by the next upload
    if (e != hipSuccess) {
        ctx->last_error = std::string("scan upload: ") + hipGetErrorString(e);
        elm_scan_destroy(s);
        return ELM_ERR_DEVICE;
    }
    *out = s;
    return ELM_OK;
}

extern "C" int elm_scan_upload(elm_ctx* ctx, const float* xyz, size_t n, size_t n_total, elm_scan** out) {
    if (group_call(ctx)) { // sharded over the group (the whole scan is handed over: a shard of a shard is not a thing)
        if (n_total != n) return ELM_ERR_INVALID;
        return elm_multi::scan_upload(ctx, xyz, n, out);
    }
    return scan_upload_impl(ctx, xyz, n, n_total, true, true, out);
}

extern "C" void elm_scan_destroy(elm_scan* s) {
    if (!s || !s->ctx) return; // a handle that is already back in the pool (double destroy)
    if (!s->shards.empty()) { // a group scan: the other ranks' shards first (each returns to its own context's pool)
        std::vector<elm_scan*> sh;
        sh.swap(s->shards);
        for (elm_scan* q : sh) elm_scan_destroy(q);
    }
    if (!ctx_alive(s->ctx, s->ctx_id)) { // the context went first (or its address was reused): release the device buffer, nothing to pool
        if (s->d_pts) (void)hipFree(s->d_pts);
        delete s;
        return;
    }
    elm_ctx* ctx = s->ctx;
    (void)hipSetDevice(ctx->device);
    if (s->d_pts) {
        if (ctx->scan_pool.size() < 64) {
            // the stream is in order: a later upload into this buffer is queued behind every kernel that still reads it
            ctx->scan_pool.emplace_back((void*)s->d_pts, s->cap_bytes);
        } else {
            (void)hipFree(s->d_pts);
        }
    }
    s->d_pts = nullptr;
    s->ctx = nullptr;
    if (ctx->scan_free.size() < 64) ctx->scan_free.push_back(s);
    else delete s;
}
extern "C" size_t elm_scan_size(const elm_scan* s) { return !s ? 0 : (s->shards.empty() ? s->n : s->n_total); }
extern "C" int elm_scan_download(const elm_scan* s, float* xyz, size_t cap) {
    if (!s || !s->ctx || (!xyz && cap)) return ELM_ERR_INVALID;
    if (!s->shards.empty() && group_call(s->ctx)) { // a group scan: rank 0's shard, then the others', each in its device order
        size_t done = std::min<size_t>(cap, s->n);
        elm_scan first = *s; // (a view without the shard list: the plain path below)
        first.shards.clear();
        int rc = elm_scan_download(&first, xyz, done);
        for (const elm_scan* q : s->shards) {
            if (rc != ELM_OK || done >= cap) break;
            const size_t k = std::min<size_t>(cap - done, q->n);
            rc = elm_scan_download(q, xyz + 3 * done, k);
            done += k;
        }
        return rc;
    }
    elm_ctx* ctx = s->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = std::min<size_t>(cap, s->n);
    if (!n) return ELM_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // the kernels that fill the scan run on the context stream
    HIPCHK(ctx, hipMemcpy(xyz, s->d_pts, n * sizeof(Pt3), hipMemcpyDeviceToHost)); // packed xyz on both sides
    return ELM_OK;
}

// ------------------------------------------------------------------------------------------------------
// registration
// ------------------------------------------------------------------------------------------------------
static int exchange(elm_ctx* ctx, double* d_sums, size_t count, hipStream_t stream = nullptr) {
    if (!stream) stream = ctx->stream;
    if (ctx->hook) {
        int rc = ctx->hook(d_sums, count, (void*)stream, ctx->hook_user);
        if (rc != 0) {
            ctx->last_error = "allreduce hook failed";
            return ELM_ERR_COMM;
        }
        return ELM_OK;
    }
    if (ctx->comm) {
        // ONE sum all-reduce of the packed normal equations of the whole batch per ICP iteration
        int rc = g_rccl.all_reduce(d_sums, d_sums, count, /*ncclDouble*/ 8, /*ncclSum*/ 0, ctx->comm, stream);
        if (rc != 0) {
            ctx->last_error = std::string("ncclAllReduce: ") + (g_rccl.get_error_string ? g_rccl.get_error_string(rc) : "error");
            return ELM_ERR_COMM;
        }
    }
    return ELM_OK;
}

// use_radar_cov (reg.hpp:186-217) changes the arithmetic of the covariance methods only: AlignCloudsLocal (P2P) never reads a covariance.
// ELM_CHECK=strict_pairs: the covariance-weighted methods run the reference's own per-pair arithmetic -- (R^-1 C R^-T)^-1 by 3x3 products and an
// inverse per pair, all 36 entries of J^T M J, LDLT on the lower triangle: the radar kernels with a zero source term -- instead of the
// world-frame / fused forms: the in-product checker of the fast forms (a plain walk, no streams: 12-27 times slower at 131 072-point
// scans, profiles/r04k_strict_rate.txt).
// Unset (the default): every map runs the fast kernels; a map with a flagged covariance of the method's own kind whose stored inverse is
// not symmetric (layout_flags bits 7 / 8, counted by k_point_cov / k_voxel_cov) additionally carries the antisymmetric part of J^T M J in
// side records (choose_path, asym_side_store) -- exact like the per-pair arithmetic.
static int strict_pairs() { // 1: per-pair kernels always, -1: fast kernels, side records by map
    return check_mode("strict_pairs") ? 1 : -1; // (read per call: a registration call, not a launch)
}
static int build_search_index(elm_map* m, bool* use_grid);
static int build_voxel_neighbourhoods(elm_map* m, bool want_faces);
// Which kernels one registration call runs.  `radar`: the per-pair kernels (use_radar_cov, or ELM_CHECK=strict_pairs).  Otherwise the search
// index (built on first use) -- dense / two-level cell grid, neighbourhood lists, voxel-mean lists, or the plain walk -- and `asym`: the
// map holds a flagged covariance of the method's kind whose stored inverse is not symmetric, and the fast kernels carry the antisymmetric
// part of J^T M J in side records (RegParams::asym; grid and voxel-list kernels, unfused reduction).  Such a map on one of the fall-back
// indices (lists / plain walk: maps the grid cannot hold, ELM_KERNEL=...) or under ELM_FUSED_REDUCE still takes the per-pair kernels.
// pcm.cpp:92-100: the covariance methods need their covariances computed (the index builders read them)
static int check_covariances(elm_ctx* ctx, const elm_map* map, int method) {
    if (!map || map->dm.n_vox == 0) return ELM_OK;
    if ((method == ELM_VGICP || method == ELM_AVGICP) && !map->info.has_voxel_cov) {
        ctx->last_error = "VGICP/AVGICP need elm_map_cal_voxel_cov_all() (pcm.cpp:92-95)";
        return ELM_ERR_INVALID;
    }
    if (method == ELM_GICP && !map->info.has_point_cov) {
        ctx->last_error = "GICP needs elm_map_cal_point_cov_all() (pcm.cpp:97-100)";
        return ELM_ERR_INVALID;
    }
    return ELM_OK;
}
struct PathChoice {
    bool radar = false, asym = false, use_grid = false, use_cells = false, use_vnbr = false;
};
static int choose_path(elm_ctx* ctx, const elm_map* map, const elm_reg_config* cfg, PathChoice* pc) {
    *pc = PathChoice();
    const int method = cfg->icp_method;
    if (method < ELM_P2P || method > ELM_AVGICP) return ELM_ERR_INVALID;
    if (!map || map->dm.n_vox == 0) return ELM_OK;
    int rc0;
    if ((rc0 = check_covariances(ctx, map, method)) != ELM_OK) return rc0;
    const int mode = strict_pairs();
    if (method != ELM_P2P && (cfg->use_radar_cov != 0 || mode == 1)) { pc->radar = true; return ELM_OK; }
    int rc;
    const bool use_nbr = ctx->kernel_mode != 2 && (method == ELM_P2P || method == ELM_GICP);
    pc->use_grid = use_nbr && ctx->kernel_mode == 4 && map->has_grid;
    if (use_nbr && !pc->use_grid && !map->has_nbr)
        if ((rc = build_search_index(const_cast<elm_map*>(map), &pc->use_grid)) != ELM_OK) return rc;
    pc->use_cells = use_nbr && !pc->use_grid && map->has_cells;
    pc->use_vnbr = ctx->kernel_mode != 2 && (method == ELM_VGICP || method == ELM_AVGICP);
    if (pc->use_vnbr)
        if ((rc = build_voxel_neighbourhoods(const_cast<elm_map*>(map), method == ELM_AVGICP)) != ELM_OK) return rc;
    const bool asym_map = mode < 0 && method != ELM_P2P && (method == ELM_GICP ? map->n_asym_pts != 0 : map->n_asym_vox != 0);
    if (asym_map) {
        if (pc->use_grid || pc->use_vnbr || pc->use_cells) pc->asym = true; // (use_cells: P2P / GICP on the neighbourhood lists -- GICP's kernel
                                                                            // there writes the side records too)
        else {
            // The one slow corner left: asymmetric covariances on a map whose search runs the PLAIN WALK (ELM_KERNEL=direct, or lists that
            // could not be cell-sorted) -- that kernel carries no side records, so the covariance methods run the per-pair kernels (exact,
            // 12-27 times slower).  Said ONCE per map, and visible in elm_reg_result.path.
            pc->radar = true; pc->use_grid = pc->use_cells = pc->use_vnbr = false;
            elm_map* mm = const_cast<elm_map*>(map);
            if (!mm->warned_slow_path) {
                mm->warned_slow_path = true;
                fprintf(stderr, "[elimaloc] map %p: %u point / %u voxel covariances are asymmetric (rank-deficient neighbourhoods) and its search index is a "
                                "fall-back form (%s): method %d runs the per-pair kernels, 12-27 times slower than the grid kernels (elm_reg_result.path = %d)\n",
                        (const void*)map, map->n_asym_pts, map->n_asym_vox, ctx->kernel_mode == 2 ? "ELM_KERNEL=direct" : "lists without cell order", method, ELM_PATH_PAIRS);
            }
        }
    }
    return ELM_OK;
}
static int path_code(const PathChoice& pc) {
    return pc.radar ? ELM_PATH_PAIRS : ((pc.use_grid ? ELM_PATH_GRID : pc.use_cells ? ELM_PATH_LISTS : pc.use_vnbr ? ELM_PATH_VOXEL_LISTS : ELM_PATH_WALK) | (pc.asym ? ELM_PATH_SIDE_RECORDS : 0));
}
// VoxelHashMap::GetCorrespondencePoints / GetCorrespondencesCov / GetCorrespondencesAllCov (vhm.cpp:31-206) as calls of their own: the pairs
// themselves, in input order, as (source index, target index).  The search is the production one -- the QUERY instantiations of the grid /
// voxel-list kernels (the code the fused accumulate kernels run, minus the sums) -- or, without such an index (ELM_KERNEL=direct / lists, a
// refused grid) and with ELM_CHECK=query_direct, the plain walk.
static int get_correspondences_impl(elm_ctx* ctx, const elm_map* map, int what, const double* xyz, size_t n, double max_dist,
                                    uint32_t* src_index, int32_t* tgt_index, size_t cap, size_t* n_pairs);
// (the host staging of these two calls is sized by the caller's n: an allocation failure is a status, never an exception through the C ABI)
extern "C" int elm_map_get_correspondences(elm_ctx* ctx, const elm_map* map, int what, const double* xyz, size_t n, double max_dist,
                                           uint32_t* src_index, int32_t* tgt_index, size_t cap, size_t* n_pairs) {
    try {
        return get_correspondences_impl(ctx, map, what, xyz, n, max_dist, src_index, tgt_index, cap, n_pairs);
    } catch (const std::bad_alloc&) {
        if (ctx) ctx->last_error = "elm_map_get_correspondences: host allocation failed";
        return ELM_ERR_ALLOC;
    }
}
static int get_correspondences_impl(elm_ctx* ctx, const elm_map* map, int what, const double* xyz, size_t n, double max_dist,
                                    uint32_t* src_index, int32_t* tgt_index, size_t cap, size_t* n_pairs) {
    if (!ctx || !map || what < 0 || what > 2 || (n && !xyz) || !n_pairs || map->ctx != ctx) return ELM_ERR_INVALID;
    *n_pairs = 0;
    if (n == 0) return ELM_OK;
    if (n > 0x7FFFFF00ull) return ELM_ERR_INVALID;
    if (ctx->in_flight) return ELM_ERR_INVALID; // the context's stream and query scratch belong to the batch in flight
    HIPCHK(ctx, hipSetDevice(ctx->device));
    elm_reg_config cfg;
    elm_reg_config_default(&cfg);
    cfg.icp_method = what == 0 ? ELM_P2P : what == 1 ? ELM_VGICP : ELM_AVGICP;
    cfg.use_radar_cov = 0;
    bool production = false;
    const bool force_direct = check_mode("query_direct");
    if (map->dm.n_vox != 0) {
        int rc;
        if (what != 0 && (rc = check_covariances(ctx, map, cfg.icp_method)) != ELM_OK) return rc;
        if (!force_direct && ctx->kernel_mode != 2) {
            if (what == 0) {
                bool g = ctx->kernel_mode == 4 && map->has_grid;
                if (!g && ctx->kernel_mode == 4 && !map->has_nbr)
                    if ((rc = build_search_index(const_cast<elm_map*>(map), &g)) != ELM_OK) return rc;
                production = g;
            } else {
                if ((rc = build_voxel_neighbourhoods(const_cast<elm_map*>(map), what == 2)) != ELM_OK) return rc;
                production = map->has_vnbr;
            }
        }
    }
    const size_t per = what == 2 ? 8 : 1;
    std::vector<int32_t> out(n * per);
    int rc;
    if ((rc = dev_reserve(ctx, ctx->d_q0, n * 3 * sizeof(double))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_q1, n * per * sizeof(int32_t))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_q2, sizeof(ScanDesc))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_q3, sizeof(ScanState))) != ELM_OK) return rc;
    double* d_q = (double*)ctx->d_q0.p;
    int32_t* d_out = (int32_t*)ctx->d_q1.p;
    ScanDesc* d_desc = (ScanDesc*)ctx->d_q2.p;
    ScanState* d_state = (ScanState*)ctx->d_q3.p;
    HIPCHK(ctx, hipMemcpyAsync(d_q, xyz, n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(d_out, 0xFF, n * per * sizeof(int32_t), ctx->stream)); // (AllCov: < 0 = no pair with that neighbour)
    if (production) {
        ScanDesc hd{};
        hd.pts = nullptr; hd.n = (uint32_t)n; hd.n_total = (uint32_t)n; hd.blk_begin = 0; hd.blk_end = (uint32_t)((n + kBlock - 1) / kBlock);
        HIPCHK(ctx, hipMemcpyAsync(d_desc, &hd, sizeof(ScanDesc), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(d_state, 0, sizeof(ScanState), ctx->stream)); // done = 0; the pose is not read by a query
        RegParams rp{};
        rp.th = max_dist; rp.th2 = max_dist * max_dist;
        rp.method = cfg.icp_method; rp.max_iter = 1;
        rp.query = d_q; rp.q_out = d_out;
        if (what == 0) launch_accumulate_grid(ctx->stream, map->dm, d_desc, 1, (int)hd.blk_end, d_state, nullptr, rp);
        else launch_accumulate_vnbr(ctx->stream, map->dm, d_desc, 1, (int)hd.blk_end, d_state, nullptr, rp);
    } else {
        launch_query_direct(ctx->stream, map->dm, what, d_q, n, max_dist * max_dist, d_out);
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out.data(), d_out, n * per * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // (also: the stack copy of the descriptor has been read)
    // marshalling: the reference's result vectors hold the pairs in input order (tbb::parallel_reduce joins its ranges in order)
    size_t k = 0;
    for (size_t i = 0; i < n; ++i) {
        if (what == 2) {
            for (int r = 0; r < 7; ++r) {
                const int32_t v = out[8 * i + r];
                if (v < 0) continue;
                if (k < cap) { if (src_index) src_index[k] = (uint32_t)i; if (tgt_index) tgt_index[k] = v; }
                ++k;
            }
        } else {
            const int32_t v = out[i];
            if (v < -1) continue;
            if (k < cap) { if (src_index) src_index[k] = (uint32_t)i; if (tgt_index) tgt_index[k] = v; }
            ++k;
        }
    }
    *n_pairs = k; // (the count even when cap was too small: call again with more room)
    return ELM_OK;
}

// Registration::AlignCloudsLocal (method ELM_P2P; reg.cpp:15-66), ::AlignCloudsLocalPointCov (ELM_GICP; :68-152) and
// ::AlignCloudsLocalVoxelCov (ELM_VGICP / ELM_AVGICP; :154-225) on pairs the caller holds: accumulate on the device with the reference's
// per-pair arithmetic, solve, return the step.
static int align_clouds_local_impl(elm_ctx* ctx, int method, const double* src_local, const double* tgt_xyz, const double* tgt_cov9,
                                   const double* src_cov9, size_t n, const double last_icp_pose[16], double trans_th,
                                   const elm_reg_config* cfg, double T_out[16], double local_cov[36], double* fitness_score);
extern "C" int elm_align_clouds_local(elm_ctx* ctx, int method, const double* src_local, const double* tgt_xyz, const double* tgt_cov9,
                                      const double* src_cov9, size_t n, const double last_icp_pose[16], double trans_th,
                                      const elm_reg_config* cfg, double T_out[16], double local_cov[36], double* fitness_score) {
    try {
        return align_clouds_local_impl(ctx, method, src_local, tgt_xyz, tgt_cov9, src_cov9, n, last_icp_pose, trans_th, cfg, T_out, local_cov, fitness_score);
    } catch (const std::bad_alloc&) {
        if (ctx) ctx->last_error = "elm_align_clouds_local: host allocation failed";
        return ELM_ERR_ALLOC;
    }
}
static int align_clouds_local_impl(elm_ctx* ctx, int method, const double* src_local, const double* tgt_xyz, const double* tgt_cov9,
                                   const double* src_cov9, size_t n, const double last_icp_pose[16], double trans_th,
                                   const elm_reg_config* cfg, double T_out[16], double local_cov[36], double* fitness_score) {
    if (!ctx || !last_icp_pose || !T_out || method < ELM_P2P || method > ELM_AVGICP || (n && (!src_local || !tgt_xyz))) return ELM_ERR_INVALID;
    if (n && method != ELM_P2P && !tgt_cov9) return ELM_ERR_INVALID;
    if (n > 0x7FFFFF00ull) return ELM_ERR_INVALID; // (the same bound as elm_map_get_correspondences: the pairs of one scan)
    if (ctx->in_flight) return ELM_ERR_INVALID;    // the context's stream and scratch belong to the batch in flight
    elm_reg_config dflt;
    if (!cfg) { elm_reg_config_default(&dflt); cfg = &dflt; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    AlignArgs a{};
    {
        double R[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R[r * 3 + c] = last_icp_pose[c * 4 + r];
        elm::inv3(R, a.Rinv); // (the 3x3 cofactor inverse of the rotation block, as the registration's own state keeps it)
        for (int r = 0; r < 3; ++r)
            a.tinv[r] = -((a.Rinv[r * 3] * last_icp_pose[12] + a.Rinv[r * 3 + 1] * last_icp_pose[13]) + a.Rinv[r * 3 + 2] * last_icp_pose[14]);
    }
    a.th = trans_th; a.th2 = trans_th * trans_th;
    a.lm_lambda = cfg->lm_lambda;
    a.method = method == ELM_AVGICP ? ELM_VGICP : method;
    a.use_src_cov = (cfg->use_radar_cov != 0 && src_cov9 && method != ELM_P2P) ? 1 : 0;
    // staging: positions as they are, covariances column-major -> row-major
    const bool cov = method != ELM_P2P;
    std::vector<double> stage_t, stage_s; // one per upload: both are read by asynchronous copies until the synchronisation below
    auto transposed = [&](const double* c9, std::vector<double>& stage) {
        stage.resize(n * 9);
        for (size_t i = 0; i < n; ++i)
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) stage[9 * i + r * 3 + c] = c9[9 * i + c * 3 + r];
        return stage.data();
    };
    int rc;
    const size_t nb = std::max<size_t>(n, 1);
    if ((rc = dev_reserve(ctx, ctx->d_q0, nb * 3 * sizeof(double))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_q1, nb * 3 * sizeof(double))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_q2, (size_t)1024 * kRadarRecord * sizeof(double))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_q3, kAlignOut * sizeof(double))) != ELM_OK) return rc;
    double *d_src = (double*)ctx->d_q0.p, *d_tgt = (double*)ctx->d_q1.p, *d_part = (double*)ctx->d_q2.p, *d_out = (double*)ctx->d_q3.p;
    double *d_cov = nullptr, *d_scov = nullptr;
    // uploads on the context's stream, like the launch behind them (the stream is non-blocking: nothing orders it after the null stream)
    if (n) HIPCHK(ctx, hipMemcpyAsync(d_src, src_local, n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (n) HIPCHK(ctx, hipMemcpyAsync(d_tgt, tgt_xyz, n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (cov && n) {
        if ((rc = dev_reserve(ctx, ctx->d_q4, n * 9 * sizeof(double))) != ELM_OK) return rc;
        d_cov = (double*)ctx->d_q4.p;
        HIPCHK(ctx, hipMemcpyAsync(d_cov, transposed(tgt_cov9, stage_t), n * 9 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        if (a.use_src_cov) {
            if ((rc = dev_reserve(ctx, ctx->d_q5, n * 9 * sizeof(double))) != ELM_OK) return rc;
            d_scov = (double*)ctx->d_q5.p;
            HIPCHK(ctx, hipMemcpyAsync(d_scov, transposed(src_cov9, stage_s), n * 9 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        }
    }
    double out[kAlignOut];
    launch_align_pairs(ctx->stream, d_src, d_tgt, d_cov, d_scov, n, a, d_part, d_out);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, d_out, sizeof(out), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(T_out, out, 16 * sizeof(double));
    if (local_cov && method == ELM_GICP) memcpy(local_cov, out + 16, 36 * sizeof(double)); // (the reference writes local_cov in AlignCloudsLocalPointCov only)
    if (fitness_score) *fitness_score = out[52];
    return ELM_OK;
}

// the side records of a launch of `blocks` workgroups over `n_scans` scans / slots; the scans' reduced side sums sit right behind the
// packed sums in d_sums (one exchange carries both)
static int reserve_asym(elm_ctx* ctx, RegParams& rp, uint32_t blocks, int n_scans) {
    int rc;
    if ((rc = dev_reserve(ctx, ctx->d_asym, (size_t)std::max<uint32_t>(blocks, 1) * kAsymRecord * sizeof(double))) != ELM_OK) return rc;
    rp.asym = (double*)ctx->d_asym.p;
    rp.asym_sums = (double*)ctx->d_sums.p + (size_t)n_scans * kSums;
    return ELM_OK;
}

// One ICP iteration's correspondence + accumulation launch for `n_scans` scans / slots, bracketed by two profiling marks (the
// solve span starts at the second).
static int enqueue_accumulate(elm_ctx* ctx, const elm_map* map, const ScanDesc* dsc, int n_scans, uint32_t blocks, ScanState* st,
                              const RegParams& rp, bool use_grid, bool use_cells, bool use_vnbr, double* partials = nullptr) {
    int rc;
    if ((rc = prof_mark(ctx)) != ELM_OK) return rc;
    if (!partials) partials = (double*)ctx->d_partials.p;
    if (blocks) {
        if (rp.radar) launch_accumulate_radar(ctx->stream, map->dm, dsc, n_scans, (int)blocks, st, partials, rp);
        else if (use_grid) launch_accumulate_grid(ctx->stream, map->dm, dsc, n_scans, (int)blocks, st, partials, rp);
        else if (use_cells) launch_accumulate_cell(ctx->stream, map->dm, dsc, n_scans, (int)blocks, st, partials, rp);
        else if (use_vnbr && rp.method == ELM_AVGICP && map->dm.vface_flagged) {
            // the fused walk on a map with flagged voxels: one flag per workgroup (all zero between launches: the fix-up launch clears what
            // the walk sets)
            RegParams rq = rp;
            const size_t need = (size_t)blocks * sizeof(uint32_t);
            if (need > ctx->d_flagged.cap) {
                if ((rc = dev_reserve(ctx, ctx->d_flagged, need + need / 2)) != ELM_OK) return rc;
                HIPCHK(ctx, hipMemsetAsync(ctx->d_flagged.p, 0, ctx->d_flagged.cap, ctx->stream));
            }
            rq.flagged = (need <= ctx->d_flagged.cap) ? (uint32_t*)ctx->d_flagged.p : nullptr;
            launch_accumulate_vnbr(ctx->stream, map->dm, dsc, n_scans, (int)blocks, st, partials, rq);
        }
        else if (use_vnbr) launch_accumulate_vnbr(ctx->stream, map->dm, dsc, n_scans, (int)blocks, st, partials, rp);
        else launch_accumulate_direct(ctx->stream, map->dm, dsc, n_scans, (int)blocks, st, partials, rp);
    }
    return prof_mark(ctx);
}
// folds the recorded events [acc_0, solve_0, acc_1, solve_1, ..., end] into the profile totals
static int prof_collect(elm_ctx* ctx) {
    if (ctx->profiling && ctx->events_used >= 3) {
        for (int k = 0; k + 1 < ctx->events_used; ++k) {
            float ms = 0.f;
            HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->events[k], ctx->events[k + 1]));
            if ((k & 1) == 0) { ctx->prof.accumulate_ms += ms; ctx->prof.accumulate_launches++; if (ctx->keep_iter_ms) ctx->iter_acc_ms.push_back(ms); }
            else { ctx->prof.solve_ms += ms; ctx->prof.solve_steps++; }
        }
    }
    ctx->events_used = 0;
    return ELM_OK;
}

// n_dev != nullptr (batch == 1): the scan's size lives in device memory (produced by the downsample kernels on this stream);
// scans[0]->n is its upper bound (grid size).  The result's point counts then come from the device state.
static int batch_enqueue_impl(elm_ctx* ctx, const elm_map* map, elm_scan* const* scans, int batch, const double* T0, const elm_reg_config* cfg,
                              int want_trace, const unsigned* n_dev) {
    if (!ctx || !map || !scans || batch <= 0 || !T0 || !cfg) return ELM_ERR_INVALID;
    if (map->ctx != ctx) return ELM_ERR_INVALID;
    if (cfg->icp_method < ELM_P2P || cfg->icp_method > ELM_AVGICP) return ELM_ERR_INVALID;
    if (ctx->in_flight) return ELM_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int method = cfg->icp_method;
    const bool map_empty = map->dm.n_vox == 0;
    if (ctx->iter_key_map != (const void*)map || ctx->iter_key_method != method || ctx->iter_key_batch != batch) {
        // (a registration that ran to max_iteration on another map / method / batch shape must not push THIS call's first early-stop
        // check to its count)
        ctx->iter_key_map = map; ctx->iter_key_method = method; ctx->iter_key_batch = batch;
        for (int k = 0; k < 8; ++k) ctx->iter_ring[k] = 0;
        ctx->iter_hint = 0;
    }
    // (use_radar_cov on several ranks: the all-reduce carries the radar kernel's 64 sums per scan instead of the 32 of the packed layout)
    if (!map_empty) {
        if ((method == ELM_VGICP || method == ELM_AVGICP) && !map->info.has_voxel_cov) {
            ctx->last_error = "VGICP/AVGICP need elm_map_cal_voxel_cov_all() (pcm.cpp:92-95)";
            return ELM_ERR_INVALID;
        }
        if (method == ELM_GICP && !map->info.has_point_cov) {
            ctx->last_error = "GICP needs elm_map_cal_point_cov_all() (pcm.cpp:97-100)";
            return ELM_ERR_INVALID;
        }
    }
    int rc;
    PathChoice pc;
    if ((rc = choose_path(ctx, map, cfg, &pc)) != ELM_OK) return rc;
    ctx->path = (map && map->dm.n_vox) ? path_code(pc) : 0;
    const bool radar = pc.radar;
    // batch descriptors
    const size_t stage_bytes = (size_t)batch * (sizeof(ScanDesc) + 16 * sizeof(double));
    if ((rc = pinned_reserve(ctx, &ctx->h_desc, &ctx->h_desc_cap, std::max<size_t>(stage_bytes, 4096))) != ELM_OK) return rc;
    ScanDesc* hd = (ScanDesc*)ctx->h_desc;
    double* hT = (double*)((char*)ctx->h_desc + (size_t)batch * sizeof(ScanDesc));
    uint32_t blocks = 0;
    uint32_t uniform_blocks = batch > 0 && scans[0] ? (scans[0]->n + kBlock - 1) / kBlock : 0;
    for (int b = 0; b < batch; ++b) {
        if (!scans[b] || scans[b]->ctx != ctx) return ELM_ERR_INVALID;
        const uint32_t nb = (scans[b]->n + kBlock - 1) / kBlock;
        if (nb != uniform_blocks) uniform_blocks = 0;
        hd[b].pts = scans[b]->d_pts;
        hd[b].n = scans[b]->n;
        hd[b].n_total = scans[b]->n_total;
        hd[b].blk_begin = blocks;
        blocks += nb;
        hd[b].blk_end = blocks;
    }
    memcpy(hT, T0, (size_t)batch * 16 * sizeof(double));
    // the counter of still-iterating scans sits right behind the states: an early-stop check reads both with ONE copy, and when the
    // counter is zero the final states are already on the host
    const size_t st_bytes = (size_t)batch * sizeof(ScanState);
    if ((rc = dev_reserve(ctx, ctx->d_scans, (size_t)batch * sizeof(ScanDesc))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_T0, (size_t)batch * 16 * sizeof(double))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_state, st_bytes + 64)) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_partials, (size_t)std::max<uint32_t>(blocks, 1) * (radar ? kRadarRecord : kSums) * sizeof(double))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_sums, (size_t)batch * (radar ? kRadarRecord : kSums + kAsymRecord) * sizeof(double))) != ELM_OK) return rc;
    if ((rc = pinned_reserve(ctx, &ctx->h_state, &ctx->h_state_cap, st_bytes + 64)) != ELM_OK) return rc;
    ctx->results_ready = false;
    elm_iter_trace* d_trace = nullptr;
    if (want_trace) {
        const size_t tb = (size_t)batch * ELM_MAX_ITER_TRACE * sizeof(elm_iter_trace);
        if ((rc = dev_reserve(ctx, ctx->d_trace, tb)) != ELM_OK) return rc;
        if ((rc = pinned_reserve(ctx, &ctx->h_trace, &ctx->h_trace_cap, tb)) != ELM_OK) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_trace.p, 0, tb, ctx->stream));
        d_trace = (elm_iter_trace*)ctx->d_trace.p;
    }
    const bool packed_init = batch <= kInitPack;
    if (!packed_init) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_scans.p, hd, (size_t)batch * sizeof(ScanDesc), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_T0.p, hT, (size_t)batch * 16 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }

    RegParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.th = cfg->max_search_dist;
    rp.th2 = cfg->max_search_dist * cfg->max_search_dist;
    rp.lm_lambda = cfg->lm_lambda;
    rp.term_thr = cfg->icp_termination_threshold_m;
    rp.min_overlap = cfg->min_overlap_ratio;
    rp.max_fitness = cfg->max_fitness_score;
    rp.method = method;
    rp.max_iter = cfg->max_iteration;
    rp.uniform_blocks = uniform_blocks;
    rp.radar = radar ? (cfg->use_radar_cov != 0 ? 1 : 2) : 0; // 2: the radar kernels without a source covariance (ELM_CHECK=strict_pairs)
    rp.stats = ctx->work_counters ? 1 : 0;
    rp.radar_var[0] = cfg->range_variance_m;
    rp.radar_var[1] = cfg->azimuth_variance_deg;
    rp.radar_var[2] = cfg->elevation_variance_deg;
    if (pc.asym && (rc = reserve_asym(ctx, rp, blocks, batch)) != ELM_OK) return rc;
    rp.rank_check = ((ctx->comm || ctx->hook) && !radar && !rp.stats) ? 1 : 0;
    ctx->rp = rp;

    // the search index (choose_path built it on first use): dense / two-level cell grid, else the cell-indexed neighbourhood lists, else
    // the plain walk (maps whose lists cannot be cell-sorted); use_radar_cov: k_accumulate_radar walks the hash map itself
    const bool use_grid = pc.use_grid, use_cells = pc.use_cells, use_vnbr = pc.use_vnbr;
    ScanState* st = (ScanState*)ctx->d_state.p;
    const ScanDesc* dsc = (const ScanDesc*)ctx->d_scans.p;
    int* d_active = (int*)((char*)ctx->d_state.p + st_bytes);
    (void)hipGetLastError();
    if (packed_init) { // descriptors and guesses as kernel arguments: no H2D copies, no memset
        InitPack pack;
        memset(&pack, 0, sizeof(pack));
        for (int b = 0; b < batch; ++b) {
            pack.d[b] = hd[b];
            memcpy(pack.T0[b], T0 + (size_t)b * 16, 16 * sizeof(double));
        }
        launch_init_pack(ctx->stream, (ScanDesc*)ctx->d_scans.p, st, pack, batch, map_empty ? 1 : 0, d_active, n_dev);
    } else {
        HIPCHK(ctx, hipMemsetAsync(d_active, 0, 2 * sizeof(int), ctx->stream));
        launch_init_state(ctx->stream, st, (const double*)ctx->d_T0.p, batch, map_empty ? 1 : 0, d_active);
    }
    const bool distributed = (ctx->comm != nullptr) || (ctx->hook != nullptr);
    ctx->events_used = 0;
    if (!map_empty) {
        for (int it = 0; it < cfg->max_iteration; ++it) {
            if ((rc = enqueue_accumulate(ctx, map, dsc, batch, blocks, st, rp, use_grid, use_cells, use_vnbr)) != ELM_OK) return rc;
            if (distributed) {
                launch_solve(ctx->stream, dsc, batch, st, (const double*)ctx->d_partials.p, (double*)ctx->d_sums.p, rp, d_trace, 1, d_active); // reduce only
                if ((rc = exchange(ctx, (double*)ctx->d_sums.p, (size_t)batch * (radar ? kRadarRecord : (rp.asym ? kSums + kAsymRecord : kSums)))) != ELM_OK) return rc;
                launch_solve(ctx->stream, dsc, batch, st, (const double*)ctx->d_partials.p, (double*)ctx->d_sums.p, rp, d_trace, 2, d_active);
            } else {
                launch_solve(ctx->stream, dsc, batch, st, (const double*)ctx->d_partials.p, (double*)ctx->d_sums.p, rp, d_trace, 0, d_active);
            }
            // Early stop: every scan of the batch has met the reference's termination rule (or a gate).  The counter is
            // derived from the (all-reduced) sums, so every rank reads the same value and stops at the same iteration.
            // The first check is placed where the previous batch finished; an unknown history never checks.
            const int done_iters = it + 1;
            if (ctx->iter_hint > 0 && done_iters >= ctx->iter_hint && done_iters < cfg->max_iteration &&
                ((done_iters - ctx->iter_hint) % 2) == 0) {
                HIPCHK(ctx, hipMemcpyAsync(ctx->h_state, st, st_bytes + 64, hipMemcpyDeviceToHost, ctx->stream)); // states + counter (+ the downsample totals)
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                if (*(const int*)((const char*)ctx->h_state + st_bytes) == 0) {
                    ctx->results_ready = true; // every scan has finished: these ARE the final states
                    break;
                }
            }
        }
        if ((rc = prof_mark(ctx)) != ELM_OK) return rc;
    }
    HIPCHK(ctx, hipGetLastError());
    if (!ctx->results_ready) HIPCHK(ctx, hipMemcpyAsync(ctx->h_state, st, st_bytes + 64, hipMemcpyDeviceToHost, ctx->stream));
    if (want_trace)
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_trace, ctx->d_trace.p, (size_t)batch * ELM_MAX_ITER_TRACE * sizeof(elm_iter_trace),
                                   hipMemcpyDeviceToHost, ctx->stream));
    ctx->batch = batch;
    ctx->want_trace = want_trace != 0;
    ctx->in_flight = true;
    return ELM_OK;
}
extern "C" int elm_register_batch_enqueue(elm_ctx* ctx, const elm_map* map, elm_scan* const* scans, int batch,
                                          const double* T0, const elm_reg_config* cfg, int want_trace) {
    if (group_call(ctx)) { ctx->last_error = "elm_register_batch_enqueue: not available on a device group (use elm_register_batch / _stream)"; return ELM_ERR_UNSUPPORTED; }
    return batch_enqueue_impl(ctx, map, scans, batch, T0, cfg, want_trace, nullptr);
}

static void state_to_result(const ScanState& h, const RegParams& rp, elm_reg_result& r, int path = 0) {
    memset(&r, 0, sizeof(r));
    r.path = path;
    memcpy(r.T, h.T, sizeof(r.T));
    memcpy(r.local_cov, h.local_cov, sizeof(r.local_cov));
    r.d_fitness = h.fitness;
    r.is_success = h.success;
    r.fitness_score = h.success ? h.fitness : 0.0; // written only on success (reg.cpp:415)
    r.iterations = h.iters;
    r.gate = h.gate;
    r.n_corr_last = h.n_corr_last;
    r.point_iterations = h.pt_iters;
    r.n_cand_total = h.cand_total;
    r.n_occ_total = h.occ_total;
    r.fallback_blocks = h.fallback_blocks;
    r.n_tested_total = h.tested_total;
    if (rp.max_iter <= 0 && h.gate == 0) r.is_success = 1; // no iteration ran: fitness gate on the initial 0.0 passes
}

extern "C" int elm_register_batch_finish(elm_ctx* ctx, elm_reg_result* results, elm_iter_trace* trace) {
    if (!ctx || !ctx->in_flight) return ELM_ERR_INVALID;
    ctx->in_flight = false;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->results_ready || ctx->want_trace || ctx->profiling) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    {
        int prc = prof_collect(ctx);
        if (prc != ELM_OK) return prc;
    }
    const ScanState* hs = (const ScanState*)ctx->h_state;
    if (ctx->rp.rank_check && ((const int*)((const char*)ctx->h_state + (size_t)ctx->batch * sizeof(ScanState)))[1] != 0) {
        ctx->last_error = "the ranks iterated different registrations in one slot (rank-agreement check of the exchanged sums)";
        return ELM_ERR_COMM;
    }
    for (int b = 0; results && b < ctx->batch; ++b) state_to_result(hs[b], ctx->rp, results[b], ctx->path);
    {
        int mx = 0;
        for (int b = 0; b < ctx->batch; ++b) mx = std::max(mx, (int)hs[b].iters);
        // where the next batch's first early-stop check goes: the LONGEST of the last eight batches.  (The previous batch's own count
        // mispredicts a stream whose registrations alternate between 2 and 3 iterations every other call: a check one iteration early
        // costs a host round trip in the middle of the registration, ~25 us, a check one iteration late one pair of launches that
        // return at once, ~10 us -- profiles/r04_single_timeline.txt.)
        ctx->iter_ring[ctx->iter_ring_pos++ & 7] = mx;
        int h = 0;
        for (int k = 0; k < 8; ++k) h = std::max(h, ctx->iter_ring[k]);
        ctx->iter_hint = h;
    }
    if (trace && ctx->want_trace)
        memcpy(trace, ctx->h_trace, (size_t)ctx->batch * ELM_MAX_ITER_TRACE * sizeof(elm_iter_trace));
    return ELM_OK;
}

extern "C" int elm_register_batch(elm_ctx* ctx, const elm_map* map, elm_scan* const* scans, int batch, const double* T0,
                                  const elm_reg_config* cfg, elm_reg_result* results, elm_iter_trace* trace) {
    if (group_call(ctx)) return elm_multi::reg_batch(ctx, map, scans, batch, T0, cfg, 0, results, trace);
    int rc = elm_register_batch_enqueue(ctx, map, scans, batch, T0, cfg, trace != nullptr);
    if (rc != ELM_OK) return rc;
    return elm_register_batch_finish(ctx, results, trace);
}

// Continuous batching: `count` registrations through `slots` slots.  Every ICP iteration is one accumulate launch over the
// slots, the solve, and a refill kernel that hands finished slots the next pending registration on the device -- so the
// launches stay full until the queue is empty (a lockstep batch ends in launches with a handful of live scans, which are
// latency bound).  Per-registration arithmetic is unchanged: results are bit-identical to elm_register_batch.
extern "C" int elm_register_stream(elm_ctx* ctx, const elm_map* map, elm_scan* const* scans, int count, const double* T0,
                                   const elm_reg_config* cfg, int slots, elm_reg_result* results, elm_iter_trace* trace) {
    if (!ctx || !map || !scans || count <= 0 || !T0 || !cfg || slots <= 0) return ELM_ERR_INVALID;
    if (group_call(ctx)) return elm_multi::reg_batch(ctx, map, scans, count, T0, cfg, slots, results, trace);
    if (map->dm.n_vox == 0 || cfg->max_iteration <= 0) // nothing iterates: the lockstep path handles the degenerate cases
        return elm_register_batch(ctx, map, scans, count, T0, cfg, results, trace);
    if (map->ctx != ctx) return ELM_ERR_INVALID;
    if (cfg->icp_method < ELM_P2P || cfg->icp_method > ELM_AVGICP) return ELM_ERR_INVALID;
    if (ctx->in_flight) return ELM_ERR_INVALID;
    int rc;
    PathChoice pc;
    if ((rc = choose_path(ctx, map, cfg, &pc)) != ELM_OK) return rc;
    ctx->path = (map && map->dm.n_vox) ? path_code(pc) : 0;
    if (pc.radar) {
        // use_radar_cov: lockstep batches of `slots` registrations (k_accumulate_radar is not a slot kernel; a radar scan is a few hundred
        // returns).  Per-registration arithmetic is that of elm_register_batch.
        for (int b0 = 0; b0 < count; b0 += slots) {
            const int n = std::min(slots, count - b0);
            const int rc = elm_register_batch(ctx, map, scans + b0, n, T0 + (size_t)b0 * 16, cfg, results ? results + b0 : nullptr,
                                              trace ? trace + (size_t)b0 * ELM_MAX_ITER_TRACE : nullptr);
            if (rc != ELM_OK) return rc;
        }
        return ELM_OK;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int method = cfg->icp_method;
    if ((method == ELM_VGICP || method == ELM_AVGICP) && !map->info.has_voxel_cov) {
        ctx->last_error = "VGICP/AVGICP need elm_map_cal_voxel_cov_all() (pcm.cpp:92-95)";
        return ELM_ERR_INVALID;
    }
    if (method == ELM_GICP && !map->info.has_point_cov) {
        ctx->last_error = "GICP needs elm_map_cal_point_cov_all() (pcm.cpp:97-100)";
        return ELM_ERR_INVALID;
    }
    const int S = std::min(std::min(slots, count), stream_max_slots());
    // queue (host staging): items, initial guesses; slot descriptors with fixed block ranges sized for the largest scan
    uint32_t max_n = 0;
    for (int b = 0; b < count; ++b) {
        if (!scans[b] || scans[b]->ctx != ctx) return ELM_ERR_INVALID;
        max_n = std::max(max_n, scans[b]->n);
    }
    const uint32_t cap_blocks = (max_n + kBlock - 1) / kBlock;
    const uint32_t blocks = cap_blocks * (uint32_t)S;
    const size_t q_bytes = (size_t)count * sizeof(QueueItem), t_bytes = (size_t)count * 16 * sizeof(double), d_bytes = (size_t)S * sizeof(ScanDesc);
    const size_t stage_bytes = q_bytes + t_bytes + d_bytes + sizeof(StreamCtrl);
    if ((rc = pinned_reserve(ctx, &ctx->h_desc, &ctx->h_desc_cap, std::max<size_t>(stage_bytes, 4096))) != ELM_OK) return rc;
    QueueItem* hq = (QueueItem*)ctx->h_desc;
    double* hT = (double*)((char*)ctx->h_desc + q_bytes);
    ScanDesc* hd = (ScanDesc*)((char*)ctx->h_desc + q_bytes + t_bytes);
    StreamCtrl* hc = (StreamCtrl*)((char*)ctx->h_desc + q_bytes + t_bytes + d_bytes);
    for (int b = 0; b < count; ++b) { hq[b].pts = scans[b]->d_pts; hq[b].n = scans[b]->n; hq[b].n_total = scans[b]->n_total; }
    memcpy(hT, T0, t_bytes);
    for (int s = 0; s < S; ++s) {
        hd[s].pts = nullptr; hd[s].n = 0; hd[s].n_total = 0;
        hd[s].blk_begin = cap_blocks * (uint32_t)s; hd[s].blk_end = cap_blocks * (uint32_t)(s + 1);
    }
    hc->next = 0; hc->completed = 0; hc->total = count; hc->ready = count; hc->done_iter = -1;
    if ((rc = dev_reserve(ctx, ctx->d_scans, d_bytes)) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_state, (size_t)S * sizeof(ScanState))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_partials, (size_t)std::max<uint32_t>(blocks, 1) * kSums * sizeof(double))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_sums, (size_t)S * (kSums + kAsymRecord) * sizeof(double))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_queue, q_bytes + t_bytes + sizeof(StreamCtrl) + (size_t)count * sizeof(ScanState) + 64)) != ELM_OK) return rc;
    if ((rc = pinned_reserve(ctx, &ctx->h_state, &ctx->h_state_cap, (size_t)count * sizeof(ScanState))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_active, 256)) != ELM_OK) return rc;
    if (!ctx->h_active) HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_active, 64, hipHostMallocDefault));
    char* qb = (char*)ctx->d_queue.p;
    QueueItem* d_q = (QueueItem*)qb;
    double* d_qT0 = (double*)(qb + q_bytes);
    StreamCtrl* d_ctrl = (StreamCtrl*)(qb + q_bytes + t_bytes);
    ScanState* d_out = (ScanState*)(qb + q_bytes + t_bytes + ((sizeof(StreamCtrl) + 63) / 64) * 64);
    elm_iter_trace* d_trace = nullptr;
    if (trace) {
        const size_t tb = (size_t)count * ELM_MAX_ITER_TRACE * sizeof(elm_iter_trace);
        if ((rc = dev_reserve(ctx, ctx->d_trace, tb)) != ELM_OK) return rc;
        if ((rc = pinned_reserve(ctx, &ctx->h_trace, &ctx->h_trace_cap, tb)) != ELM_OK) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_trace.p, 0, tb, ctx->stream));
        d_trace = (elm_iter_trace*)ctx->d_trace.p;
    }
    HIPCHK(ctx, hipMemcpyAsync(d_q, hq, q_bytes + t_bytes, hipMemcpyHostToDevice, ctx->stream)); // items + guesses are contiguous
    HIPCHK(ctx, hipMemcpyAsync(d_ctrl, hc, sizeof(StreamCtrl), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_scans.p, hd, d_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_active.p, 0, 2 * sizeof(int), ctx->stream));

    RegParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.th = cfg->max_search_dist;
    rp.th2 = cfg->max_search_dist * cfg->max_search_dist;
    rp.lm_lambda = cfg->lm_lambda;
    rp.term_thr = cfg->icp_termination_threshold_m;
    rp.min_overlap = cfg->min_overlap_ratio;
    rp.max_fitness = cfg->max_fitness_score;
    rp.method = method;
    rp.max_iter = cfg->max_iteration;
    rp.uniform_blocks = cap_blocks; // every slot owns cap_blocks workgroups
    rp.radar = 0;
    rp.stats = ctx->work_counters ? 1 : 0;
    rp.radar_var[0] = rp.radar_var[1] = rp.radar_var[2] = 0.0;
    if (pc.asym && (rc = reserve_asym(ctx, rp, blocks, S)) != ELM_OK) return rc;
    rp.rank_check = ((ctx->comm || ctx->hook) && !rp.stats) ? 1 : 0;
    ctx->rp = rp;
    const bool use_grid = pc.use_grid, use_cells = pc.use_cells, use_vnbr = pc.use_vnbr;

    ScanState* st = (ScanState*)ctx->d_state.p;
    ScanDesc* dsc = (ScanDesc*)ctx->d_scans.p;
    int* d_active = (int*)ctx->d_active.p;
    const bool distributed = (ctx->comm != nullptr) || (ctx->hook != nullptr);
    (void)hipGetLastError();
    launch_stream_refill(ctx->stream, dsc, st, S, d_q, d_qT0, d_out, d_ctrl, 1);
    ctx->events_used = 0;
    // Iterations needed: unknown in advance (it depends on when each registration converges).  The previous call with the
    // same shape is the prediction: enqueue that many without looking, then read the completed counter after every
    // further iteration.  All ranks read the same counter (it derives from all-reduced sums) and stop together.
    const int hard_limit = ((count + S - 1) / S + 1) * cfg->max_iteration;
    const bool same_shape = ctx->stream_hint_count == count && ctx->stream_hint_slots == S;
    const int predicted = same_shape ? ctx->stream_hint_iters : 0;
    double* const partials = (double*)ctx->d_partials.p;
    double* const sums = (double*)ctx->d_sums.p;
    int it = 0;
    for (; it < hard_limit; ++it) {
        if ((rc = enqueue_accumulate(ctx, map, dsc, S, blocks, st, rp, use_grid, use_cells, use_vnbr, partials)) != ELM_OK) return rc;
        if (distributed) {
            // reduce -> all-reduce -> solve -> refill: four launches + one collective per iteration.  (Side sums of a map with an asymmetric
            // covariance sit right behind the packed sums: one exchange carries both.)  The refill launch walks the slots in slot order and
            // hands the finished ones the next pending registrations -- first come, first served like the single-rank stream, yet identical
            // on every rank (the finished flags derive from the all-reduced sums), so no slot idles while the queue has work.
            launch_solve(ctx->stream, dsc, S, st, partials, sums, rp, d_trace, 1, d_active);
            if ((rc = exchange(ctx, sums, (size_t)S * (rp.asym ? kSums + kAsymRecord : kSums))) != ELM_OK) return rc;
            const StreamArgs sv = {dsc, d_q, d_qT0, d_out, d_ctrl, 0, /*save_only*/ 1, 0, it}; // the solve saves + counts, the refill assigns
            launch_solve(ctx->stream, dsc, S, st, partials, sums, rp, d_trace, 2, d_active, &sv);
            launch_stream_refill(ctx->stream, dsc, st, S, d_q, d_qT0, d_out, d_ctrl, 0, /*save*/ 0);
        } else {
            // single rank: the solve hands finished slots their next registration itself (no refill launch)
            const StreamArgs sa = {dsc, d_q, d_qT0, d_out, d_ctrl, 0, 0, 0, it};
            launch_solve(ctx->stream, dsc, S, st, partials, sums, rp, d_trace, 0, d_active, &sa);
        }
        const int done_iters = it + 1;
        const bool look = predicted > 0 ? done_iters >= predicted : (done_iters >= (count + S - 1) / S && (done_iters % 2) == 0);
        if (look) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_active, &d_ctrl->completed, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            if (*ctx->h_active == count) { ++it; break; }
        }
    }
    if ((rc = prof_mark(ctx)) != ELM_OK) return rc;
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_state, d_out, (size_t)count * sizeof(ScanState), hipMemcpyDeviceToHost, ctx->stream));
    if (trace)
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_trace, ctx->d_trace.p, (size_t)count * ELM_MAX_ITER_TRACE * sizeof(elm_iter_trace),
                                   hipMemcpyDeviceToHost, ctx->stream));
    ctx->h_active[2] = 0;
    if (rp.rank_check) HIPCHK(ctx, hipMemcpyAsync(ctx->h_active + 2, d_active + 1, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ctx->h_active[3] = -1;
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_active + 3, &d_ctrl->done_iter, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = prof_collect(ctx)) != ELM_OK) return rc;
    if (ctx->h_active[2] != 0) {
        ctx->last_error = "the ranks iterated different registrations in one slot (rank-agreement check of the exchanged sums)";
        return ELM_ERR_COMM;
    }
    ctx->stream_hint_count = count;
    ctx->stream_hint_slots = S;
    // what the next call of this shape enqueues before it first looks: the iterations this one NEEDED (the device notes the iteration
    // whose solve finished the last registration) -- not the count at which the host happened to look, which can only grow: a call
    // with poor initial guesses would leave every later call of the shape enqueueing its iteration count in empty launches
    ctx->stream_hint_iters = (ctx->h_active[3] >= 0) ? std::min(it, ctx->h_active[3] + 1) : it;
    const ScanState* hs = (const ScanState*)ctx->h_state;
    for (int b = 0; results && b < count; ++b) state_to_result(hs[b], rp, results[b], ctx->path);
    if (trace) memcpy(trace, ctx->h_trace, (size_t)count * ELM_MAX_ITER_TRACE * sizeof(elm_iter_trace));
    return ELM_OK;
}

// ---- host-fed continuous batching ---------------------------------------------------------------------------------------------
extern "C" void* elm_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, std::max<size_t>(bytes, 64), hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
extern "C" void elm_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

// Diagnostic for the benches: what one plain hipMemcpyAsync of `bytes` from `host` achieves on this box (median of reps), i.e.
// the PCIe rate a host-fed stream sits under.
extern "C" int elm_ctx_measure_h2d(elm_ctx* ctx, const void* host, size_t bytes, int reps, double* gb_per_s) {
    if (!ctx || !host || !bytes || reps <= 0 || !gb_per_s) return ELM_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = dev_reserve(ctx, ctx->d_arena, bytes);
    if (rc != ELM_OK) return rc;
    hipEvent_t a, b;
    HIPCHK(ctx, hipEventCreate(&a));
    HIPCHK(ctx, hipEventCreate(&b));
    std::vector<float> ms;
    hipError_t e = hipSuccess;
    for (int r = 0; r < reps + 1 && e == hipSuccess; ++r) { // the first copy warms the path up
        e = hipEventRecord(a, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_arena.p, host, bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(b, ctx->stream);
        if (e == hipSuccess) e = hipEventSynchronize(b);
        float t = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&t, a, b);
        if (r) ms.push_back(t);
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (e != hipSuccess) {
        ctx->last_error = std::string("h2d probe: ") + hipGetErrorString(e);
        return ELM_ERR_DEVICE;
    }
    std::sort(ms.begin(), ms.end());
    *gb_per_s = (double)bytes / ((double)ms[ms.size() / 2] * 1e-3) / 1e9;
    return ELM_OK;
}

constexpr int kStageSets = 3; // staging sets of a host-fed stream: the DMA of group g + 2 may run while group g is still being ordered
static int ensure_side_streams(elm_ctx* ctx) {
    if (!ctx->copy_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (!ctx->order_stream) {
        // the ordering kernel runs beside the accumulate launches of the compute stream: highest priority, so that its few workgroups
        // are dispatched as soon as a CU has room instead of behind the tail of a 65 536-workgroup launch
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; (void)hipGetLastError(); }
        HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->order_stream, hipStreamNonBlocking, hi));
    }
    if (!ctx->poll_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->poll_stream, hipStreamNonBlocking));
    for (hipEvent_t* e : {&ctx->ev_iter[0], &ctx->ev_iter[1], &ctx->ev_iter[2], &ctx->ev_iter[3]})
        if (!*e) HIPCHK(ctx, hipEventCreateWithFlags(e, hipEventDisableTiming));
    return ELM_OK;
}
static void sync_all_streams(elm_ctx* ctx) {
    for (hipStream_t st : {ctx->copy_stream, ctx->order_stream, ctx->poll_stream, ctx->stream})
        if (st) (void)hipStreamSynchronize(st);
}

// elm_register_stream with the scans still in HOST memory when the call starts (the per-call contract of RunRegister, reg.cpp:274-290:
// the caller hands over a point vector, not a device handle).  Three things overlap:
//   copy stream    H2D of the packed xyz of the next group of scans into one of kStageSets (3) staging sets (page-locked sources: one DMA per
//                  group when the group is contiguous in host memory)
//   order stream   k_scan_order over the group (staging -> the registration's own place in the arena, Hilbert order), then the
//                  group's arrival is published in ctrl->ready
//   compute stream the ICP iterations of the registrations that have arrived: a slot that finishes (or idles) claims the next
//                  arrived registration in the solve kernel, exactly as in elm_register_stream
// Results are bit-identical to elm_register_stream on the same scans uploaded with elm_scan_upload (same ordering kernel).
extern "C" int elm_register_stream_host(elm_ctx* ctx, const elm_map* map, const float* const* scan_xyz, const uint32_t* n_pts, int count,
                                        const double* T0, const elm_reg_config* cfg, int slots, elm_reg_result* results, elm_iter_trace* trace) {
    if (!ctx || !map || !scan_xyz || !n_pts || count <= 0 || !T0 || !cfg || slots <= 0) return ELM_ERR_INVALID;
    if (map->ctx != ctx) return ELM_ERR_INVALID;
    if (cfg->icp_method < ELM_P2P || cfg->icp_method > ELM_AVGICP) return ELM_ERR_INVALID;
    if (ctx->in_flight) return ELM_ERR_INVALID;
    if (ctx->comm || ctx->hook) {
        ctx->last_error = "host-fed streams run on one rank (slot assignment follows scan arrival, which differs between ranks)";
        return ELM_ERR_UNSUPPORTED;
    }
    uint32_t max_n = 0;
    for (int b = 0; b < count; ++b) {
        if ((!scan_xyz[b] && n_pts[b]) || n_pts[b] > 0x7FFFFFFFu) return ELM_ERR_INVALID;
        max_n = std::max(max_n, n_pts[b]);
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    PathChoice pc;
    if ((rc = choose_path(ctx, map, cfg, &pc)) != ELM_OK) return rc;
    ctx->path = (map && map->dm.n_vox) ? path_code(pc) : 0;
    if (map->dm.n_vox == 0 || cfg->max_iteration <= 0 || max_n == 0 || pc.radar) {
        // nothing iterates (or use_radar_cov: see elm_register_stream): upload and let the lockstep path handle the degenerate cases
        std::vector<elm_scan*> sc((size_t)count, nullptr);
        rc = ELM_OK;
        for (int b = 0; b < count && rc == ELM_OK; ++b) rc = elm_scan_upload(ctx, scan_xyz[b], n_pts[b], n_pts[b], &sc[b]);
        if (rc == ELM_OK) rc = elm_register_batch(ctx, map, sc.data(), count, T0, cfg, results, trace);
        for (elm_scan* x : sc) elm_scan_destroy(x);
        return rc;
    }
    const int method = cfg->icp_method;
    if ((method == ELM_VGICP || method == ELM_AVGICP) && !map->info.has_voxel_cov) {
        ctx->last_error = "VGICP/AVGICP need elm_map_cal_voxel_cov_all() (pcm.cpp:92-95)";
        return ELM_ERR_INVALID;
    }
    if (method == ELM_GICP && !map->info.has_point_cov) {
        ctx->last_error = "GICP needs elm_map_cal_point_cov_all() (pcm.cpp:97-100)";
        return ELM_ERR_INVALID;
    }
    if ((rc = ensure_side_streams(ctx)) != ELM_OK) return rc;
    if ((rc = ensure_hilbert(ctx)) != ELM_OK) return rc;
    const int S = std::min(std::min(slots, count), stream_max_slots());
    const uint32_t cap_blocks = (max_n + kBlock - 1) / kBlock;
    const uint32_t blocks = cap_blocks * (uint32_t)S;
    // upload groups: ~32 MB of points each, at most 64 scans (one ordering workgroup per scan)
    const size_t scan_stride = (((size_t)max_n * sizeof(Pt3)) + 255) & ~(size_t)255;
    const int G = (int)std::min<size_t>(std::min<size_t>(64, (size_t)count), std::max<size_t>(1, ((size_t)32 << 20) / scan_stride));
    const int n_groups = (count + G - 1) / G;
    // the arena: every registration's ordered scan has its own place for the whole call
    std::vector<size_t> off((size_t)count + 1, 0);
    for (int b = 0; b < count; ++b) off[b + 1] = off[b] + ((((size_t)n_pts[b] * sizeof(Pt3)) + 255) & ~(size_t)255);
    if ((rc = dev_reserve(ctx, ctx->d_arena, std::max<size_t>(off[count], 256))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_raw, (size_t)kStageSets * G * scan_stride)) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_order_tmp, (size_t)kStageSets * G * max_n * sizeof(uint32_t))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_order_jobs, (size_t)count * sizeof(OrderJob))) != ELM_OK) return rc;
    if ((rc = pinned_reserve(ctx, &ctx->h_jobs, &ctx->h_jobs_cap, std::max<size_t>((size_t)count * sizeof(OrderJob), 4096))) != ELM_OK) return rc;
    OrderJob* hj = (OrderJob*)ctx->h_jobs;
    for (int b = 0; b < count; ++b) {
        const int set = (b / G) % kStageSets, j = b % G;
        hj[b].src = (const Pt3*)((char*)ctx->d_raw.p + ((size_t)set * G + j) * scan_stride);
        hj[b].dst = (Pt3*)((char*)ctx->d_arena.p + off[b]);
        hj[b].tmp = (uint32_t*)ctx->d_order_tmp.p + ((size_t)set * G + j) * max_n;
        hj[b].n = n_pts[b];
        hj[b]._pad = 0;
    }
    // queue, guesses, control block, result states (the layout of elm_register_stream)
    const size_t q_bytes = (size_t)count * sizeof(QueueItem), t_bytes = (size_t)count * 16 * sizeof(double), d_bytes = (size_t)S * sizeof(ScanDesc);
    const size_t stage_bytes = q_bytes + t_bytes + sizeof(StreamCtrl);
    if ((rc = pinned_reserve(ctx, &ctx->h_desc, &ctx->h_desc_cap, std::max<size_t>(stage_bytes, 4096))) != ELM_OK) return rc;
    QueueItem* hq = (QueueItem*)ctx->h_desc;
    double* hT = (double*)((char*)ctx->h_desc + q_bytes);
    StreamCtrl* hc = (StreamCtrl*)((char*)ctx->h_desc + q_bytes + t_bytes);
    for (int b = 0; b < count; ++b) { hq[b].pts = hj[b].dst; hq[b].n = n_pts[b]; hq[b].n_total = n_pts[b]; }
    memcpy(hT, T0, t_bytes);
    hc->next = 0; hc->completed = 0; hc->total = count; hc->ready = 0; hc->done_iter = -1;
    if ((rc = dev_reserve(ctx, ctx->d_scans, d_bytes)) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_state, (size_t)S * sizeof(ScanState))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_partials, (size_t)std::max<uint32_t>(blocks, 1) * kSums * sizeof(double))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_sums, (size_t)S * (kSums + kAsymRecord) * sizeof(double))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_queue, q_bytes + t_bytes + sizeof(StreamCtrl) + (size_t)count * sizeof(ScanState) + 64)) != ELM_OK) return rc;
    if ((rc = pinned_reserve(ctx, &ctx->h_state, &ctx->h_state_cap, (size_t)count * sizeof(ScanState))) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_active, 256)) != ELM_OK) return rc;
    if (!ctx->h_active) HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_active, 64, hipHostMallocDefault));
    char* qb = (char*)ctx->d_queue.p;
    QueueItem* d_q = (QueueItem*)qb;
    double* d_qT0 = (double*)(qb + q_bytes);
    StreamCtrl* d_ctrl = (StreamCtrl*)(qb + q_bytes + t_bytes);
    ScanState* d_out = (ScanState*)(qb + q_bytes + t_bytes + ((sizeof(StreamCtrl) + 63) / 64) * 64);
    elm_iter_trace* d_trace = nullptr;
    if (trace) {
        const size_t tb = (size_t)count * ELM_MAX_ITER_TRACE * sizeof(elm_iter_trace);
        if ((rc = dev_reserve(ctx, ctx->d_trace, tb)) != ELM_OK) return rc;
        if ((rc = pinned_reserve(ctx, &ctx->h_trace, &ctx->h_trace_cap, tb)) != ELM_OK) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_trace.p, 0, tb, ctx->stream));
        d_trace = (elm_iter_trace*)ctx->d_trace.p;
    }
    RegParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.th = cfg->max_search_dist;
    rp.th2 = cfg->max_search_dist * cfg->max_search_dist;
    rp.lm_lambda = cfg->lm_lambda;
    rp.term_thr = cfg->icp_termination_threshold_m;
    rp.min_overlap = cfg->min_overlap_ratio;
    rp.max_fitness = cfg->max_fitness_score;
    rp.method = method;
    rp.max_iter = cfg->max_iteration;
    rp.uniform_blocks = cap_blocks;
    rp.radar = 0;
    rp.stats = ctx->work_counters ? 1 : 0;
    rp.radar_var[0] = rp.radar_var[1] = rp.radar_var[2] = 0.0;
    if (pc.asym && (rc = reserve_asym(ctx, rp, blocks, S)) != ELM_OK) return rc;
    ctx->rp = rp;
    const bool use_grid = pc.use_grid, use_cells = pc.use_cells, use_vnbr = pc.use_vnbr;

#define HF_CHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            ctx->last_error = std::string(#call) + ": " + hipGetErrorString(e_);                       \
            sync_all_streams(ctx);                                                                     \
            return ELM_ERR_DEVICE;                                                                     \
        }                                                                                              \
    } while (0)
    HF_CHK(hipMemcpyAsync(d_q, hq, q_bytes + t_bytes, hipMemcpyHostToDevice, ctx->stream));
    HF_CHK(hipMemcpyAsync(d_ctrl, hc, sizeof(StreamCtrl), hipMemcpyHostToDevice, ctx->stream));
    HF_CHK(hipMemcpyAsync(ctx->d_order_jobs.p, hj, (size_t)count * sizeof(OrderJob), hipMemcpyHostToDevice, ctx->stream));
    HF_CHK(hipMemsetAsync(ctx->d_active.p, 0, 2 * sizeof(int), ctx->stream));
    ScanState* st = (ScanState*)ctx->d_state.p;
    ScanDesc* dsc = (ScanDesc*)ctx->d_scans.p;
    int* d_active = (int*)ctx->d_active.p;
    (void)hipGetLastError();
    launch_slots_idle(ctx->stream, dsc, st, S, cap_blocks);
    // the side streams start behind the control block's initialisation (and behind whatever the compute stream still reads from
    // the arena / staging of an earlier call)
    HF_CHK(hipEventRecord(ctx->ev_iter[0], ctx->stream));
    HF_CHK(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_iter[0], 0));
    HF_CHK(hipStreamWaitEvent(ctx->order_stream, ctx->ev_iter[0], 0));
    ctx->events_used = 0;
    const OrderJob* d_jobs = (const OrderJob*)ctx->d_order_jobs.p;
    int g_enq = 0;
    // one pair of events per upload group (a pool that only grows): group g's "copied" and "ordered"
    while ((int)ctx->ev_groups.size() < 2 * n_groups) {
        hipEvent_t e;
        HF_CHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->ev_groups.push_back(e);
    }
    hipStream_t copy_stream = ctx->copy_stream, order_stream = ctx->order_stream;
    auto enqueue_group = [&](int g) -> hipError_t {
        const int set = g % kStageSets, r0 = g * G, r1 = std::min(count, r0 + G);
        hipEvent_t ev_copied = ctx->ev_groups[2 * g], ev_ordered = ctx->ev_groups[2 * g + 1];
        hipError_t e = hipSuccess;
        if (g >= kStageSets) e = hipStreamWaitEvent(copy_stream, ctx->ev_groups[2 * (g - kStageSets) + 1], 0); // the set's previous ordering kernel has read it
        char* base = (char*)ctx->d_raw.p + (size_t)set * G * scan_stride;
        // one DMA for the whole group when it is contiguous in host memory and in the staging set
        bool contiguous = (size_t)max_n * sizeof(Pt3) == scan_stride;
        for (int r = r0; r < r1 && contiguous; ++r)
            contiguous = n_pts[r] == max_n && (r == r0 || scan_xyz[r] == scan_xyz[r - 1] + 3 * (size_t)max_n);
        if (contiguous) {
            if (e == hipSuccess) e = hipMemcpyAsync(base, scan_xyz[r0], (size_t)(r1 - r0) * scan_stride, hipMemcpyHostToDevice, copy_stream);
        } else {
            for (int r = r0; r < r1 && e == hipSuccess; ++r)
                if (n_pts[r]) e = hipMemcpyAsync(base + (size_t)(r - r0) * scan_stride, scan_xyz[r], (size_t)n_pts[r] * sizeof(Pt3), hipMemcpyHostToDevice, copy_stream);
        }
        if (e == hipSuccess) e = hipEventRecord(ev_copied, copy_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(order_stream, ev_copied, 0);
        if (e == hipSuccess) {
            launch_scan_order(order_stream, d_jobs + r0, r1 - r0, ctx->d_hilbert);
            launch_publish_ready(order_stream, d_ctrl, r1);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(ev_ordered, order_stream);
        return e;
    };
    // Iterations are enqueued a few ahead of the device (an event per iteration throttles the host); the number of finished
    // registrations is read on a stream of its own, so that looking never waits for the iterations in flight.
    const StreamArgs sa = {dsc, d_q, d_qT0, d_out, d_ctrl, 1, 0, 0, 0};
    int it = 0, done_seen = 0, idle_turns = 0, g_done = 0;
    const int idle_limit = 4000000; // ~ minutes of polling without a single registration finishing: a lost upload, give up
    const int groups_ahead = 2 * kStageSets; // uploads enqueued but not yet ordered: enough to keep the DMA engine fed; a long backlog
                                             // of queued copies and cross-stream waits slows the runtime's submission path down
    for (;;) {
        while (g_done < g_enq && hipEventQuery(ctx->ev_groups[2 * g_done + 1]) == hipSuccess) ++g_done;
        (void)hipGetLastError(); // hipErrorNotReady of the query
        for (int k = 0; k < 2 && g_enq < n_groups && g_enq - g_done < groups_ahead; ++k, ++g_enq) HF_CHK(enqueue_group(g_enq));
        if (it >= 4) HF_CHK(hipEventSynchronize(ctx->ev_iter[it & 3]));
        if ((rc = enqueue_accumulate(ctx, map, dsc, S, blocks, st, rp, use_grid, use_cells, use_vnbr)) != ELM_OK) { sync_all_streams(ctx); return rc; }
        launch_solve(ctx->stream, dsc, S, st, (const double*)ctx->d_partials.p, (double*)ctx->d_sums.p, rp, d_trace, 0, d_active, &sa);
        HF_CHK(hipEventRecord(ctx->ev_iter[it & 3], ctx->stream));
        ++it;
        HF_CHK(hipMemcpyAsync(ctx->h_active, &d_ctrl->completed, sizeof(int), hipMemcpyDeviceToHost, ctx->poll_stream));
        HF_CHK(hipStreamSynchronize(ctx->poll_stream));
        const int c = *ctx->h_active;
        if (c >= count) break;
        idle_turns = (c == done_seen) ? idle_turns + 1 : 0;
        done_seen = c;
        if (idle_turns > idle_limit) {
            ctx->last_error = "host-fed stream made no progress";
            sync_all_streams(ctx);
            return ELM_ERR_DEVICE;
        }
    }
    if ((rc = prof_mark(ctx)) != ELM_OK) { sync_all_streams(ctx); return rc; }
    HF_CHK(hipGetLastError());
    HF_CHK(hipMemcpyAsync(ctx->h_state, d_out, (size_t)count * sizeof(ScanState), hipMemcpyDeviceToHost, ctx->stream));
    if (trace)
        HF_CHK(hipMemcpyAsync(ctx->h_trace, ctx->d_trace.p, (size_t)count * ELM_MAX_ITER_TRACE * sizeof(elm_iter_trace),
                              hipMemcpyDeviceToHost, ctx->stream));
    HF_CHK(hipStreamSynchronize(ctx->stream));
#undef HF_CHK
    if ((rc = prof_collect(ctx)) != ELM_OK) return rc;
    const ScanState* hs = (const ScanState*)ctx->h_state;
    for (int b = 0; results && b < count; ++b) state_to_result(hs[b], rp, results[b], ctx->path);
    if (trace) memcpy(trace, ctx->h_trace, (size_t)count * ELM_MAX_ITER_TRACE * sizeof(elm_iter_trace));
    return ELM_OK;
}

// RunRegister on host buffers; n_total > n: this context holds a SHARD of an n_total-point scan (a rank of a process-per-GPU job, or a
// rank of a device group: elm_multi.cpp) -- the sums are exchanged, the overlap gate is taken against n_total.  quiet: no log text (only one
// rank of a job prints what RunRegister prints).
static int register_impl(elm_ctx* ctx, const elm_map* map, const float* scan_xyz, size_t n, size_t n_total, const double T0[16],
                         const elm_reg_config* cfg, double T_out[16], int* is_success, double* fitness_score,
                         double local_cov[36], elm_reg_result* result, elm_iter_trace* trace, bool quiet) {
    if (!ctx || !map || !T0 || !cfg || n_total < n) return ELM_ERR_INVALID;
    elm_scan* s = nullptr;
    // the caller's point order (see scan_upload_impl); no wait between the upload and the iterations: the pinned staging buffer is
    // not touched again before elm_register_batch has synchronised the stream
    int rc = scan_upload_impl(ctx, scan_xyz, n, n_total, false, false, &s);
    if (rc != ELM_OK) return rc;
    elm_reg_result res;
    // b_debug_print (reg.cpp:343-347, 396-403): the reference's per-iteration and total timing lines need the iteration trace and one
    // hipEvent pair per accumulate launch; the default (0) records neither -- and prints nothing unless a gate fails
    const bool dbg = cfg->b_debug_print != 0;
    std::vector<elm_iter_trace> own_trace;
    elm_iter_trace* tr = trace;
    const bool was_profiling = ctx->profiling;
    const elm_profile saved_prof = ctx->prof;
    std::chrono::steady_clock::time_point t_begin;
    if (dbg) {
        if (!tr) { own_trace.resize(ELM_MAX_ITER_TRACE); tr = own_trace.data(); }
        ctx->profiling = true;
        ctx->keep_iter_ms = true;
        ctx->iter_acc_ms.clear();
        t_begin = std::chrono::steady_clock::now();
    }
    rc = elm_register_batch(ctx, map, &s, 1, T0, cfg, &res, tr);
    if (dbg) {
        ctx->profiling = was_profiling;
        ctx->keep_iter_ms = false;
        if (!was_profiling) ctx->prof = saved_prof; // the caller did not ask for a profile: its totals are untouched
    }
    if (rc != ELM_OK) (void)hipStreamSynchronize(ctx->stream); // the upload may still be reading the pinned staging buffer
    elm_scan_destroy(s);
    if (rc != ELM_OK) return rc;
    if ((dbg || res.gate != 0) && !quiet) {
        // what RunRegister writes to stdout (elm_format_register_log): warnings always, the timing lines with b_debug_print
        const double total_ms = dbg ? std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_begin).count() / 1000.0 : 0.0;
        std::vector<double> corr(ctx->iter_acc_ms);
        corr.resize((size_t)std::max(res.iterations, 0), 0.0);
        // sized from the text itself (up to ELM_MAX_ITER_TRACE per-iteration lines of ~95 bytes: a fixed 4 KB buffer cut the totals off)
        const size_t need = elm_format_register_log(cfg, &res, n_total, dbg ? tr : nullptr, dbg ? corr.data() : nullptr, total_ms, nullptr, 0);
        std::vector<char> text(need + 1);
        elm_format_register_log(cfg, &res, n_total, dbg ? tr : nullptr, dbg ? corr.data() : nullptr, total_ms, text.data(), text.size());
        fputs(text.data(), stdout);
        fflush(stdout);
    }
    if (T_out) memcpy(T_out, res.T, sizeof(res.T));
    if (is_success) *is_success = res.is_success;
    if (fitness_score && res.is_success) *fitness_score = res.fitness_score; // untouched on failure, like the reference
    if (local_cov) memcpy(local_cov, res.local_cov, sizeof(res.local_cov));
    if (result) *result = res;
    return ELM_OK;
}
extern "C" int elm_register(elm_ctx* ctx, const elm_map* map, const float* scan_xyz, size_t n, const double T0[16],
                            const elm_reg_config* cfg, double T_out[16], int* is_success, double* fitness_score,
                            double local_cov[36], elm_reg_result* result, elm_iter_trace* trace) {
    if (group_call(ctx)) return elm_multi::reg(ctx, map, scan_xyz, n, T0, cfg, T_out, is_success, fitness_score, local_cov, result, trace);
    return register_impl(ctx, map, scan_xyz, n, n, T0, cfg, T_out, is_success, fitness_score, local_cov, result, trace, false);
}
extern "C" int elm_register_shard(elm_ctx* ctx, const elm_map* map, const float* shard_xyz, size_t n, size_t n_total, const double T0[16],
                                  const elm_reg_config* cfg, elm_reg_result* result, elm_iter_trace* trace, int quiet) {
    if (!result) return ELM_ERR_INVALID;
    if (group_call(ctx)) return ELM_ERR_INVALID; // a group shards for itself
    return register_impl(ctx, map, shard_xyz, n, n_total, T0, cfg, nullptr, nullptr, nullptr, nullptr, result, trace, quiet != 0);
}

// accessors for elm_multi.cpp (the structs live in this file)
namespace elm_host {
elm_group*& ctx_group(elm_ctx* ctx) { return ctx->group; }
int ctx_device(const elm_ctx* ctx) { return ctx->device; }
void ctx_set_error(elm_ctx* ctx, const std::string& text) { if (ctx) ctx->last_error = text; }
std::vector<elm_map*>& map_replicas(elm_map* m) { return m->replicas; }
elm_ctx* map_ctx(const elm_map* m) { return m->ctx; }
std::vector<elm_scan*>& scan_shards(elm_scan* s) { return s->shards; }
elm_ctx* scan_ctx(const elm_scan* s) { return s->ctx; }
void scan_set_total(elm_scan* s, size_t n_total) { s->n_total = (uint32_t)n_total; }
} // namespace elm_host

// ------------------------------------------------------------------------------------------------------
// deskew
// ------------------------------------------------------------------------------------------------------
// uploads the raw points + tables and runs the deskew kernel; the undistorted points (3 floats each) stay on the device
static int deskew_enqueue(elm_ctx* ctx, const float* xyz, const float* rel_time, size_t n, const elm_deskew_tables* tab, float** d_out_p) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t k = (size_t)tab->i_imu_pointer_cur + 1;
    const size_t in_bytes = n * 4 * sizeof(float);
    const size_t tab_bytes = k * 4 * sizeof(double);
    int rc;
    if ((rc = dev_reserve(ctx, ctx->d_stage_pts, in_bytes + n * 3 * sizeof(float) + tab_bytes + 64)) != ELM_OK) return rc;
    if ((rc = pinned_reserve(ctx, &ctx->h_stage, &ctx->h_stage_cap, std::max<size_t>(tab_bytes, 4096))) != ELM_OK) return rc;
    char* base = (char*)ctx->d_stage_pts.p;
    double* d_tab = (double*)base;
    float* d_xyz = (float*)(base + ((tab_bytes + 63) / 64) * 64);
    float* d_time = d_xyz + 3 * n;
    float* d_out = d_time + n;
    HIPCHK(ctx, hipMemcpyAsync(d_xyz, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (!tab->b_run_deskew) { // pcm.cpp:513-525: plain copy
        *d_out_p = d_xyz;
        return ELM_OK;
    }
    double* ht = (double*)ctx->h_stage;
    memcpy(ht, tab->vec_d_imu_time, k * sizeof(double));
    memcpy(ht + k, tab->vec_d_imu_rot_x, k * sizeof(double));
    memcpy(ht + 2 * k, tab->vec_d_imu_rot_y, k * sizeof(double));
    memcpy(ht + 3 * k, tab->vec_d_imu_rot_z, k * sizeof(double));
    HIPCHK(ctx, hipMemcpyAsync(d_tab, ht, tab_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_time, rel_time, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    DeskewDev d;
    d.time_scan_cur = tab->d_time_scan_cur;
    d.time_scan_end = tab->d_time_scan_end;
    d.imu_pointer_cur = tab->i_imu_pointer_cur;
    d.odom_available = tab->b_is_odom_available;
    d.incre_x = tab->f_odom_incre_x; d.incre_y = tab->f_odom_incre_y; d.incre_z = tab->f_odom_incre_z;
    d._pad = 0.f;
    d.imu_time = d_tab; d.rot_x = d_tab + k; d.rot_y = d_tab + 2 * k; d.rot_z = d_tab + 3 * k;
    (void)hipGetLastError();
    launch_deskew(ctx->stream, d_xyz, d_time, (uint32_t)n, d, d_out);
    HIPCHK(ctx, hipGetLastError());
    *d_out_p = d_out;
    return ELM_OK;
}

extern "C" int elm_deskew(elm_ctx* ctx, const float* xyz, const float* rel_time, size_t n, const elm_deskew_tables* tab,
                          float* xyz_out, int* ok) {
    if (!ctx || !tab || !ok || (n && (!xyz || !rel_time || !xyz_out)) || n > 0x7FFFFFFFull) return ELM_ERR_INVALID;
    *ok = 0;
    if (!tab->b_is_imu_available || !tab->b_is_odom_available) return ELM_OK; // pcm.cpp:494-496
    *ok = 1;
    if (!tab->b_run_deskew) { // pcm.cpp:513-525: plain copy
        memcpy(xyz_out, xyz, n * 3 * sizeof(float));
        return ELM_OK;
    }
    if (n == 0) return ELM_OK;
    float* d_out = nullptr;
    int rc = deskew_enqueue(ctx, xyz, rel_time, n, tab, &d_out);
    if (rc != ELM_OK) return rc;
    HIPCHK(ctx, hipMemcpyAsync(xyz_out, d_out, n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ELM_OK;
}

// VoxelHashMap::VoxelDownsample (vhm.hpp:260-283) of `d_und` (n undistorted points in HBM) into a fresh scan, on the context
// stream, no host wait.  {kept points, "a key does not pack" flag} land in device memory right behind the state of a one-scan
// batch, so the registration's own read-back brings them to the host.  The hash table is left clean by the pass itself.
static int downsample_enqueue(elm_ctx* ctx, const float* d_und, size_t n, double voxel_size, elm_scan** sc_out, unsigned** d_total_out) {
    int rc;
    unsigned cap_log2 = 6;
    while (((size_t)1 << cap_log2) < 2 * std::max<size_t>(n, 1)) ++cap_log2;
    const size_t cap = (size_t)1 << cap_log2, nb = (n + 1023) / 1024;
    const size_t bytes = cap * 12 + std::max<size_t>(n, 1) * 4 + (nb + 1) * 4 + 64;
    if (bytes > ctx->d_ds.cap) ctx->ds_clean_ptr = nullptr; // dev_reserve reallocates: fresh memory (even at the same address) is not "all ones"
    if ((rc = dev_reserve(ctx, ctx->d_ds, bytes)) != ELM_OK) return rc;
    if ((rc = dev_reserve(ctx, ctx->d_state, sizeof(ScanState) + 64)) != ELM_OK) return rc;
    char* base = (char*)ctx->d_ds.p;
    unsigned long long* d_table = (unsigned long long*)base;
    unsigned* d_first = (unsigned*)(base + cap * 8);
    unsigned* d_slot = d_first + cap;
    unsigned* d_bcount = d_slot + std::max<size_t>(n, 1);
    unsigned* d_total = (unsigned*)((char*)ctx->d_state.p + sizeof(ScanState) + 16); // [0] kept points, [1] overflow flag
    elm_scan* sc = nullptr;
    if ((rc = scan_alloc(ctx, n, &sc)) != ELM_OK) return rc;
    hipError_t e = hipSuccess;
    const bool clean = ctx->ds_clean_ptr == ctx->d_ds.p && ctx->ds_clean_cap_log2 == cap_log2;
    ctx->ds_clean_ptr = nullptr; // dirty until this pass has cleaned up after itself
    if (!clean) e = hipMemsetAsync(d_table, 0xFF, cap * 12, ctx->stream); // keys and first indices: all ones
    if (e == hipSuccess) e = hipMemsetAsync(d_total, 0, 8, ctx->stream);
    if (e == hipSuccess && n) {
        (void)hipGetLastError();
        launch_voxel_downsample(ctx->stream, d_und, (uint32_t)n, voxel_size, d_table, d_first, cap_log2, d_slot, d_bcount, d_total,
                                (int*)(d_total + 1), sc->d_pts);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        ctx->last_error = std::string("downsample: ") + hipGetErrorString(e);
        elm_scan_destroy(sc);
        return ELM_ERR_DEVICE;
    }
    ctx->ds_clean_ptr = ctx->d_ds.p;
    ctx->ds_clean_cap_log2 = cap_log2;
    sc->n = (uint32_t)n; // upper bound until the host has read the kept count
    sc->n_total = (uint32_t)n;
    *sc_out = sc;
    *d_total_out = d_total;
    return ELM_OK;
}

// DeskewPointCloud's per-point loop + VoxelHashMap::VoxelDownsample fused on the device: the undistorted cloud never leaves
// HBM, what comes out is a resident scan (the kept points in input order) ready for elm_register_batch.
extern "C" int elm_deskew_downsample(elm_ctx* ctx, const float* xyz, const float* rel_time, size_t n, const elm_deskew_tables* tab,
                                     double voxel_size, elm_scan** scan_out, int* ok) {
    if (!ctx || !tab || !ok || !scan_out || !(voxel_size > 0.0) || (n && (!xyz || !rel_time)) || n > 0x7FFFFFFFull) return ELM_ERR_INVALID;
    *ok = 0;
    *scan_out = nullptr;
    if (!tab->b_is_imu_available || !tab->b_is_odom_available) return ELM_OK; // pcm.cpp:494-496
    *ok = 1;
    int rc;
    float* d_und = nullptr;
    if (n) {
        if ((rc = deskew_enqueue(ctx, xyz, rel_time, n, tab, &d_und)) != ELM_OK) return rc;
    }
    elm_scan* sc = nullptr;
    unsigned* d_total = nullptr;
    if ((rc = downsample_enqueue(ctx, d_und, n, voxel_size, &sc, &d_total)) != ELM_OK) return rc;
    unsigned h_total[2] = {0, 0};
    hipError_t e = hipMemcpyAsync(h_total, d_total, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess || h_total[1]) {
        ctx->last_error = e != hipSuccess ? std::string("deskew + downsample: ") + hipGetErrorString(e)
                                          : "a voxel key does not fit the packed device table (|coordinate / voxel size| >= 2^20)";
        elm_scan_destroy(sc);
        return e != hipSuccess ? ELM_ERR_DEVICE : ELM_ERR_UNSUPPORTED;
    }
    sc->n = h_total[0];
    sc->n_total = h_total[0];
    *scan_out = sc;
    return ELM_OK;
}

// ---- the device pass of the node callback (elm_glue.cpp) ---------------------------------------------------------------------
namespace elm_host {
// page-locked scratch of the context: the callback filters the raw cloud straight into it, laid out as the device wants it
// ([kCbTableBytes of deskew tables][xyz]), so that one DMA moves tables and points
void* callback_staging(elm_ctx* ctx, size_t bytes) {
    if (!ctx || hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    if (pinned_reserve(ctx, &ctx->h_stage, &ctx->h_stage_cap, std::max<size_t>(bytes, 4096)) != ELM_OK) return nullptr;
    return ctx->h_stage;
}
// DeskewPoint for every point -> VoxelDownsample -> RunRegister in ONE device pass: DMA of [tables | xyz] and of the per-point
// times, k_deskew, the k_ds_* kernels, k_init_pack (the scan's size stays on the device) and the ICP iterations are enqueued
// back to back; the host waits once, for the result.  stage = callback_staging(): tables at 0 (time, rot_x, rot_y, rot_z, 2000
// doubles each), xyz at kCbTableBytes.  *unpackable = 1: a voxel key does not fit the device table -> the caller's host path.
int callback_register(elm_ctx* ctx, const elm_map* map, const void* stage, const float* rel_time, size_t n, const elm_deskew_tables* tab,
                      double voxel_size, const double T0[16], const elm_reg_config* cfg, elm_reg_result* result, uint64_t* n_source,
                      int* unpackable) {
    if (!ctx || !map || !stage || stage != ctx->h_stage || !rel_time || !tab || !T0 || !cfg || !result || !n_source || !unpackable || n == 0 ||
        n > 0x7FFFFFFFull)
        return ELM_ERR_INVALID;
    if (group_call(ctx)) return ELM_ERR_UNSUPPORTED; // (a device group: the callback takes its stage-by-stage path -- deskew and downsample on the
                                                     // lead device, the registration sharded over the ranks; elm_glue.cpp)
    *unpackable = 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    const size_t pts_bytes = n * 3 * sizeof(float);
    if ((rc = dev_reserve(ctx, ctx->d_stage_pts, kCbTableBytes + n * 7 * sizeof(float) + 64)) != ELM_OK) return rc;
    char* base = (char*)ctx->d_stage_pts.p;
    double* d_tab = (double*)base;
    float* d_xyz = (float*)(base + kCbTableBytes);
    float* d_time = d_xyz + 3 * n;
    float* d_out = d_time + n;
    HIPCHK(ctx, hipMemcpyAsync(base, stage, kCbTableBytes + pts_bytes, hipMemcpyHostToDevice, ctx->stream));
    const float* d_und = d_xyz; // run_deskew = 0: the cloud as it is (pcm.cpp:513-525)
    if (tab->b_run_deskew) {
        HIPCHK(ctx, hipMemcpyAsync(d_time, rel_time, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        DeskewDev d;
        d.time_scan_cur = tab->d_time_scan_cur;
        d.time_scan_end = tab->d_time_scan_end;
        d.imu_pointer_cur = tab->i_imu_pointer_cur;
        d.odom_available = tab->b_is_odom_available;
        d.incre_x = tab->f_odom_incre_x; d.incre_y = tab->f_odom_incre_y; d.incre_z = tab->f_odom_incre_z;
        d._pad = 0.f;
        d.imu_time = d_tab; d.rot_x = d_tab + kCbTableRows; d.rot_y = d_tab + 2 * kCbTableRows; d.rot_z = d_tab + 3 * kCbTableRows;
        (void)hipGetLastError();
        launch_deskew(ctx->stream, d_xyz, d_time, (uint32_t)n, d, d_out);
        HIPCHK(ctx, hipGetLastError());
        d_und = d_out;
    }
    elm_scan* sc = nullptr;
    unsigned* d_total = nullptr;
    if ((rc = downsample_enqueue(ctx, d_und, n, voxel_size, &sc, &d_total)) != ELM_OK) return rc;
    rc = batch_enqueue_impl(ctx, map, &sc, 1, T0, cfg, 0, d_total);
    if (rc == ELM_OK) rc = elm_register_batch_finish(ctx, result, nullptr);
    else (void)hipStreamSynchronize(ctx->stream);
    elm_scan_destroy(sc);
    if (rc != ELM_OK) return rc;
    const unsigned* ht = (const unsigned*)((const char*)ctx->h_state + sizeof(ScanState) + 16);
    *n_source = ht[0];
    *unpackable = ht[1] ? 1 : 0;
    return ELM_OK;
}
} // namespace elm_host

namespace {
// pcl::getTransformation (float) -- rotation Rz(yaw) Ry(pitch) Rx(roll) and translation
struct Aff3f { float m[3][4]; };
static Aff3f get_transformation_f(float x, float y, float z, float roll, float pitch, float yaw) {
    const float A = cosf(yaw), B = sinf(yaw), C = cosf(pitch), D = sinf(pitch), E = cosf(roll), F = sinf(roll);
    const float DE = D * E, DF = D * F;
    Aff3f t;
    t.m[0][0] = A * C; t.m[0][1] = A * DF - B * E; t.m[0][2] = B * F + A * DE; t.m[0][3] = x;
    t.m[1][0] = B * C; t.m[1][1] = A * E + B * DF; t.m[1][2] = B * DE - A * F; t.m[1][3] = y;
    t.m[2][0] = -D;    t.m[2][1] = C * F;          t.m[2][2] = C * E;          t.m[2][3] = z;
    return t;
}
// tf::Matrix3x3(q).getRPY()
static void quat_to_rpy(const double q[4], double* roll, double* pitch, double* yaw) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double s = 2.0 / (x * x + y * y + z * z + w * w);
    const double xs = x * s, ys = y * s, zs = z * s;
    const double wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
    const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
    if (fabs(m20) >= 1.0) {
        *yaw = 0.0;
        *roll = atan2(m21, m22);
        *pitch = (m20 < 0) ? M_PI / 2.0 : -M_PI / 2.0;
    } else {
        *pitch = -asin(m20);
        const double cp = cos(*pitch);
        *roll = atan2(m21 / cp, m22 / cp);
        *yaw = atan2(m10 / cp, m00 / cp);
    }
}
} // namespace

extern "C" int elm_deskew_prepare(const double* imu4, size_t n_imu, const double* odom14, size_t n_odom, double stamp,
                                  float front_time, float back_time, int lidar_scan_time_end, int run_deskew,
                                  double* tab_time, double* tab_rx, double* tab_ry, double* tab_rz, size_t tab_cap,
                                  elm_deskew_tables* out) {
    if (!out || !tab_time || !tab_rx || !tab_ry || !tab_rz || tab_cap < 2) return ELM_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    // DeskewPointCloud head (pcm.cpp:473-486)
    double scan_cur = stamp, scan_end = stamp + (double)back_time;
    if (lidar_scan_time_end) {
        scan_end = stamp;
        scan_cur = scan_end + (double)front_time;
    }
    out->d_time_scan_cur = scan_cur;
    out->d_time_scan_end = scan_end;
    out->b_run_deskew = run_deskew;
    out->vec_d_imu_time = tab_time; out->vec_d_imu_rot_x = tab_rx; out->vec_d_imu_rot_y = tab_ry; out->vec_d_imu_rot_z = tab_rz;
    // ImuDeskewInfo (pcm.cpp:533-585)
    {
        size_t first = 0;
        while (first < n_imu && imu4[4 * first] < scan_cur - 0.01) ++first;
        int cur = 0;
        if (first < n_imu) {
            for (size_t i = first; i < n_imu; ++i) {
                const double t = imu4[4 * i];
                if (t > scan_end + 0.01) break;
                if ((size_t)cur >= tab_cap) break;
                if (cur == 0) {
                    tab_rx[0] = tab_ry[0] = tab_rz[0] = 0.0;
                    tab_time[0] = t;
                    ++cur;
                    continue;
                }
                const double dt = t - tab_time[cur - 1];
                tab_rx[cur] = tab_rx[cur - 1] + imu4[4 * i + 1] * dt;
                tab_ry[cur] = tab_ry[cur - 1] + imu4[4 * i + 2] * dt;
                tab_rz[cur] = tab_rz[cur - 1] + imu4[4 * i + 3] * dt;
                tab_time[cur] = t;
                ++cur;
            }
            --cur;
            out->i_imu_pointer_cur = cur < 0 ? 0 : cur;
            out->b_is_imu_available = cur > 0 ? 1 : 0;
        }
    }
    // OdomDeskewInfo (pcm.cpp:587-729)
    {
        size_t first = 0;
        while (first < n_odom && odom14[14 * first] < scan_cur - 0.1) ++first;
        if (first < n_odom && !(odom14[14 * first] > scan_cur)) {
            size_t si = first;
            for (size_t i = first; i < n_odom; ++i) {
                si = i;
                if (odom14[14 * i] < scan_cur) continue;
                break;
            }
            const double* so = odom14 + 14 * si;
            double roll, pitch, yaw;
            quat_to_rpy(so + 4, &roll, &pitch, &yaw);
            const Aff3f begin = get_transformation_f((float)so[1], (float)so[2], (float)so[3], (float)roll, (float)pitch, (float)yaw);
            const double* lo = odom14 + 14 * (n_odom - 1);
            double end_stamp, ep[3], eq[4];
            if (lo[0] > scan_end) {
                size_t ei = first;
                for (size_t i = first; i < n_odom; ++i) {
                    ei = i;
                    if (odom14[14 * i] < scan_end) continue;
                    break;
                }
                const double* eo = odom14 + 14 * ei;
                end_stamp = eo[0];
                for (int k = 0; k < 3; ++k) ep[k] = eo[1 + k];
                for (int k = 0; k < 4; ++k) eq[k] = eo[4 + k];
            } else { // extrapolate with the last twist (pcm.cpp:648-708)
                const double dt = scan_end - lo[0];
                end_stamp = scan_end;
                double r2, p2, y2;
                quat_to_rpy(lo + 4, &r2, &p2, &y2);
                const double cy = cos(y2), sy = sin(y2), cp = cos(p2), sp = sin(p2), cr = cos(r2), sr = sin(r2);
                const double R[3][3] = {{cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr},
                                        {sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr},
                                        {-sp, cp * sr, cp * cr}};
                for (int k = 0; k < 3; ++k) ep[k] = lo[1 + k] + (R[k][0] * lo[8] + R[k][1] * lo[9] + R[k][2] * lo[10]) * dt;
                r2 += lo[11] * dt; p2 += lo[12] * dt; y2 += lo[13] * dt;
                const double cY = cos(y2 * 0.5), sY = sin(y2 * 0.5), cP = cos(p2 * 0.5), sP = sin(p2 * 0.5), cR = cos(r2 * 0.5), sR = sin(r2 * 0.5);
                eq[0] = sR * cP * cY - cR * sP * sY;
                eq[1] = cR * sP * cY + sR * cP * sY;
                eq[2] = cR * cP * sY - sR * sP * cY;
                eq[3] = cR * cP * cY + sR * sP * sY;
            }
            quat_to_rpy(eq, &roll, &pitch, &yaw);
            const Aff3f end = get_transformation_f((float)ep[0], (float)ep[1], (float)ep[2], (float)roll, (float)pitch, (float)yaw);
            // (begin^-1 * end).translation(), float32 (Eigen Affine3f: linear().inverse(), -linv * t)
            float L[9], Li[9];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) L[i * 3 + j] = begin.m[i][j];
            {
                const float c00 = L[4] * L[8] - L[5] * L[7], c01 = L[5] * L[6] - L[3] * L[8], c02 = L[3] * L[7] - L[4] * L[6];
                const float det = (c00 * L[0] + c01 * L[1]) + c02 * L[2];
                const float id = 1.0f / det;
                Li[0] = c00 * id; Li[3] = c01 * id; Li[6] = c02 * id;
                Li[1] = (L[2] * L[7] - L[1] * L[8]) * id; Li[4] = (L[0] * L[8] - L[2] * L[6]) * id; Li[7] = (L[1] * L[6] - L[0] * L[7]) * id;
                Li[2] = (L[1] * L[5] - L[2] * L[4]) * id; Li[5] = (L[2] * L[3] - L[0] * L[5]) * id; Li[8] = (L[0] * L[4] - L[1] * L[3]) * id;
            }
            float bt[3];
            for (int i = 0; i < 3; ++i) {
                const float ti = -((Li[i * 3] * begin.m[0][3] + Li[i * 3 + 1] * begin.m[1][3]) + Li[i * 3 + 2] * begin.m[2][3]);
                bt[i] = ((Li[i * 3] * end.m[0][3] + Li[i * 3 + 1] * end.m[1][3]) + Li[i * 3 + 2] * end.m[2][3]) + ti * 1.0f;
            }
            // InterpolateTfWithTime (lf.hpp:219-241): only the translation reaches the per-point deskew
            const double dt_scan = scan_end - scan_cur, dt_trans = end_stamp - so[0];
            if (dt_trans == 0.0) {
                out->f_odom_incre_x = out->f_odom_incre_y = out->f_odom_incre_z = 0.f;
            } else {
                const float fr = (float)(dt_scan / dt_trans);
                out->f_odom_incre_x = bt[0] * fr;
                out->f_odom_incre_y = bt[1] * fr;
                out->f_odom_incre_z = bt[2] * fr;
            }
            out->b_is_odom_available = 1;
        }
    }
    return ELM_OK;
}

// ------------------------------------------------------------------------------------------------------
// multi-GPU
// ------------------------------------------------------------------------------------------------------
extern "C" int elm_comm_get_unique_id(void* id_bytes) {
    if (!id_bytes) return ELM_ERR_INVALID;
    std::string err;
    if (!load_rccl(&err)) return ELM_ERR_COMM;
    return g_rccl.get_unique_id(id_bytes) == 0 ? ELM_OK : ELM_ERR_COMM;
}

extern "C" int elm_comm_init(elm_ctx* ctx, int rank, int nranks, const void* id_bytes) {
    if (!ctx || !id_bytes || nranks < 1 || rank < 0 || rank >= nranks) return ELM_ERR_INVALID;
    if (group_call(ctx)) { ctx->last_error = "elm_comm_init: a device group exchanges among its own ranks"; return ELM_ERR_INVALID; }
    if (!load_rccl(&ctx->last_error)) return ELM_ERR_COMM;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    elm_nccl_id id;
    memcpy(id.internal, id_bytes, ELM_COMM_ID_BYTES);
    void* comm = nullptr;
    int rc = g_rccl.comm_init_rank(&comm, nranks, id, rank);
    if (rc != 0) {
        ctx->last_error = std::string("ncclCommInitRank: ") + (g_rccl.get_error_string ? g_rccl.get_error_string(rc) : "error");
        return ELM_ERR_COMM;
    }
    ctx->comm = comm;
    ctx->rank = rank;
    ctx->nranks = nranks;
    return ELM_OK;
}

extern "C" int elm_comm_destroy(elm_ctx* ctx) {
    if (!ctx) return ELM_ERR_INVALID;
    if (ctx->comm && g_rccl.comm_destroy) g_rccl.comm_destroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->nranks = 1;
    ctx->rank = 0;
    return ELM_OK;
}

// What the COMMUNICATOR says about itself (ncclCommCount / ncclCommUserRank), not what the caller passed to elm_comm_init: the benches
// print it so that a run labelled N GPUs is known to have formed an N-rank RCCL communicator.  No communicator: 0 ranks.
extern "C" int elm_comm_info(elm_ctx* ctx, int* rank, int* nranks) {
    if (!ctx || !rank || !nranks) return ELM_ERR_INVALID;
    *rank = 0;
    *nranks = 0;
    if (!ctx->comm) return ELM_OK;
    if (!g_rccl.comm_count || !g_rccl.comm_user_rank) {
        ctx->last_error = "librccl lacks ncclCommCount / ncclCommUserRank";
        return ELM_ERR_COMM;
    }
    int rc = g_rccl.comm_count(ctx->comm, nranks);
    if (rc == 0) rc = g_rccl.comm_user_rank(ctx->comm, rank);
    if (rc != 0) {
        ctx->last_error = std::string("ncclCommCount: ") + (g_rccl.get_error_string ? g_rccl.get_error_string(rc) : "error");
        return ELM_ERR_COMM;
    }
    return ELM_OK;
}

extern "C" int elm_comm_set_hook(elm_ctx* ctx, elm_allreduce_fn fn, void* user) {
    if (!ctx) return ELM_ERR_INVALID;
    ctx->hook = fn;
    ctx->hook_user = user;
    return ELM_OK;
}
