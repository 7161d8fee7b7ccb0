// elm_multi.cpp -- device groups: the multi-GPU registration path inside ONE process (elm_ctx_create(device_ids[], n, &ctx);
// the reference's pcm_matching node is one process that calls Registration::RunRegister, pcm.cpp:280-282).
//
// elm_ctx_create_multi(device_ids, n) creates one plain context per entry and returns the first as the group's LEAD.  Calls on the lead
// (elm_map_build, elm_map_cal_*_cov_all, elm_scan_upload, elm_register, elm_register_batch / _stream) are spread over the group: the map is
// replicated on every device, a scan's points are sharded (contiguous parts of its Hilbert order: dist.py / DESIGN.md section 6), and every
// ICP iteration sums the ranks' packed normal equations -- ONE ncclAllReduce(double, sum) of 32 (+16) doubles per slot over the communicators
// the ranks form among themselves (RCCL over xGMI; each rank is driven by its own host thread, so every collective is entered by all ranks
// at once without ncclGroupStart / End), each rank then solves the same sums redundantly.  The per-rank code is the process-per-GPU code:
// the workers call the same public entry points on their own contexts.
// A device id may repeat (device_ids = {0, 0}: two ranks on one GPU).  RCCL refuses two ranks on one device; such a group exchanges through
// page-locked host memory instead (sum in rank order, identical on every rank) -- the form the one-GPU test box runs.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "elm_hostapi.hpp"

namespace {
thread_local bool t_in_worker = false;

// Exchange through host memory (ranks that share a device): every rank copies its packed sums out, all wait, every rank adds the N
// buffers in rank order (the same operand order everywhere: bit-identical sums on all ranks), all wait, every rank copies its sum back.
struct HostExchange {
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    bool broken = false;
    std::vector<double*> out; // per rank, page-locked
    std::vector<double*> sum;
    std::vector<size_t> cap, count;
};
} // namespace

struct elm_group {
    int n = 0;
    std::vector<int> devices;
    std::vector<elm_ctx*> ctx; // ctx[0] = the lead
    int exchange = 0;          // 0: one rank, 1: RCCL, 2: host memory
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    uint64_t seq = 0;
    const std::function<int(int)>* job = nullptr;
    int pending = 0;
    std::vector<int> rc;
    bool quit = false;
    HostExchange hx;
    struct HookUser { elm_group* g; int r; };
    std::vector<HookUser> hook_user;
    std::mutex call_mu; // one group call at a time (the contexts are thread-compatible, not thread-safe)
};

namespace {
bool hx_wait(elm_group* g) { // a barrier that a failing rank can break
    HostExchange& x = g->hx;
    std::unique_lock<std::mutex> lk(x.mu);
    if (x.broken) return false;
    const uint64_t my = x.gen;
    if (++x.arrived == g->n) {
        x.arrived = 0;
        ++x.gen;
        x.cv.notify_all();
        return true;
    }
    const bool ok = x.cv.wait_for(lk, std::chrono::seconds(120), [&] { return x.gen != my || x.broken; });
    if (!ok) { x.broken = true; x.cv.notify_all(); }
    return !x.broken;
}
void hx_break(elm_group* g) {
    std::lock_guard<std::mutex> lk(g->hx.mu);
    g->hx.broken = true;
    g->hx.cv.notify_all();
}

int host_exchange(void* dev_ptr, size_t n, void* hip_stream, void* user) {
    auto* hu = static_cast<elm_group::HookUser*>(user);
    elm_group* g = hu->g;
    const int r = hu->r;
    HostExchange& x = g->hx;
    if (hipStreamSynchronize((hipStream_t)hip_stream) != hipSuccess) { hx_break(g); return 1; }
    if (x.cap[r] < n) { // (a rank's own buffers: the others read them only between the two barriers below)
        if (x.out[r]) (void)hipHostFree(x.out[r]);
        if (x.sum[r]) (void)hipHostFree(x.sum[r]);
        x.out[r] = x.sum[r] = nullptr;
        x.cap[r] = 0;
        const size_t want = std::max<size_t>(2 * n, 4096);
        if (hipHostMalloc((void**)&x.out[r], want * sizeof(double)) != hipSuccess || hipHostMalloc((void**)&x.sum[r], want * sizeof(double)) != hipSuccess) {
            hx_break(g);
            return 1;
        }
        x.cap[r] = want;
    }
    if (hipMemcpy(x.out[r], dev_ptr, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { hx_break(g); return 1; }
    x.count[r] = n;
    if (!hx_wait(g)) return 1;
    for (int q = 0; q < g->n; ++q)
        if (x.count[q] != n) { hx_break(g); return 1; } // the ranks are not in the same iteration of the same call
    double* s = x.sum[r];
    memcpy(s, x.out[0], n * sizeof(double));
    for (int q = 1; q < g->n; ++q) {
        const double* o = x.out[q];
        for (size_t k = 0; k < n; ++k) s[k] += o[k];
    }
    if (!hx_wait(g)) return 1;
    if (hipMemcpy(dev_ptr, s, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { hx_break(g); return 1; }
    return 0;
}

void worker_main(elm_group* g, int r) {
    t_in_worker = true;
    (void)hipSetDevice(g->devices[r]);
    uint64_t seen = 0;
    for (;;) {
        const std::function<int(int)>* job = nullptr;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv_job.wait(lk, [&] { return g->quit || g->seq != seen; });
            if (g->quit) return;
            seen = g->seq;
            job = g->job;
        }
        int rc;
        try {
            rc = (*job)(r);
        } catch (const std::bad_alloc&) {
            rc = ELM_ERR_ALLOC;
        } catch (...) {
            rc = ELM_ERR_DEVICE;
        }
        if (rc != ELM_OK && g->exchange == 2) hx_break(g); // the others may be waiting for this rank in the exchange
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->rc[r] = rc;
            if (--g->pending == 0) g->cv_done.notify_all();
        }
    }
}

// fn(rank) on every rank's worker thread; the first failure is the call's status, its context's message the lead's
int run_all(elm_group* g, const std::function<int(int)>& fn) {
    {
        std::lock_guard<std::mutex> lk(g->hx.mu); // every worker is idle here
        g->hx.broken = false;
        g->hx.arrived = 0;
    }
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->job = &fn;
        g->pending = g->n;
        std::fill(g->rc.begin(), g->rc.end(), ELM_OK);
        ++g->seq;
        g->cv_job.notify_all();
        g->cv_done.wait(lk, [&] { return g->pending == 0; });
        g->job = nullptr;
    }
    for (int r = 0; r < g->n; ++r)
        if (g->rc[r] != ELM_OK) {
            if (r != 0) elm_host::ctx_set_error(g->ctx[0], "rank " + std::to_string(r) + " (device " + std::to_string(g->devices[r]) + "): " + elm_last_error(g->ctx[r]));
            return g->rc[r];
        }
    return ELM_OK;
}

elm_group* group_of(elm_ctx* lead) { return lead ? elm_host::ctx_group(lead) : nullptr; }

// the Hilbert curve over the 64 x 64 ordering cells (2 m, sensor frame) of the device's ordering kernel (elm_api.cpp: hilbert_xy2d)
uint32_t hilbert_xy2d(uint32_t order, uint32_t x, uint32_t y) {
    uint32_t d = 0;
    for (uint32_t s = 1u << (order - 1); s > 0; s >>= 1) {
        const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d += s * s * ((3u * rx) ^ ry);
        if (ry == 0) {
            if (rx == 1) { x = s - 1 - x; y = s - 1 - y; }
            std::swap(x, y);
        }
    }
    return d;
}
// The scan sorted (stable counting sort) along that curve: what the ranks take contiguous shards of (locality-aware sharding).
void spatial_order(const float* xyz, size_t n, std::vector<float>& out) {
    static const std::vector<uint16_t> lut = [] {
        std::vector<uint16_t> t(64 * 64);
        for (uint32_t y = 0; y < 64; ++y)
            for (uint32_t x = 0; x < 64; ++x) t[y * 64 + x] = (uint16_t)hilbert_xy2d(6, x, y);
        return t;
    }();
    std::vector<uint16_t> key(n);
    std::vector<uint32_t> start(4097, 0);
    for (size_t i = 0; i < n; ++i) {
        const float fx = floorf(xyz[3 * i] * 0.5f), fy = floorf(xyz[3 * i + 1] * 0.5f);
        // (NaN / huge coordinates: the comparisons below are false for NaN -> cell 0 / 63 like the device's clamp)
        const int cx = fx >= 31.f ? 63 : (fx > -32.f ? (int)fx + 32 : 0), cy = fy >= 31.f ? 63 : (fy > -32.f ? (int)fy + 32 : 0);
        key[i] = lut[(size_t)cy * 64 + cx];
        ++start[key[i] + 1];
    }
    for (int k = 0; k < 4096; ++k) start[k + 1] += start[k];
    out.resize(3 * n);
    for (size_t i = 0; i < n; ++i) {
        const size_t o = start[key[i]]++;
        out[3 * o] = xyz[3 * i]; out[3 * o + 1] = xyz[3 * i + 1]; out[3 * o + 2] = xyz[3 * i + 2];
    }
}
} // namespace

namespace elm_multi {
bool in_worker() { return t_in_worker; }

void destroy(elm_group* g) {
    if (!g) return;
    elm_ctx* lead = g->ctx.empty() ? nullptr : g->ctx[0];
    if (!g->workers.empty()) {
        // communicators are destroyed by the rank that owns them, all at once (ncclCommDestroy may wait for the peers)
        run_all(g, [&](int r) {
            elm_comm_set_hook(g->ctx[r], nullptr, nullptr);
            return elm_comm_destroy(g->ctx[r]);
        });
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->quit = true;
            g->cv_job.notify_all();
        }
        for (std::thread& t : g->workers) t.join();
    }
    if (lead) elm_host::ctx_group(lead) = nullptr;
    for (size_t r = 1; r < g->ctx.size(); ++r) elm_ctx_destroy(g->ctx[r]);
    for (double* p : g->hx.out) if (p) (void)hipHostFree(p);
    for (double* p : g->hx.sum) if (p) (void)hipHostFree(p);
    delete g;
}

int set_work_counters(elm_ctx* lead, int enable) {
    elm_group* g = group_of(lead);
    std::lock_guard<std::mutex> call(g->call_mu);
    return run_all(g, [&](int r) { return elm_ctx_set_work_counters(g->ctx[r], enable); });
}

int map_build(elm_ctx* lead, const float* xyz, size_t n, double voxel_size, int max_points_per_voxel, elm_map** out) {
    elm_group* g = group_of(lead);
    std::lock_guard<std::mutex> call(g->call_mu);
    std::vector<elm_map*> maps(g->n, nullptr);
    // every rank replays AddPoints for itself (host work, in parallel) and uploads its replica
    const int rc = run_all(g, [&](int r) { return elm_map_build(g->ctx[r], xyz, n, voxel_size, max_points_per_voxel, &maps[r]); });
    if (rc != ELM_OK) {
        for (elm_map* m : maps) if (m) elm_map_destroy(m);
        return rc;
    }
    elm_host::map_replicas(maps[0]).assign(maps.begin() + 1, maps.end());
    *out = maps[0];
    return ELM_OK;
}

static const elm_map* rank_map(const elm_map* lead_map, int r) { return r == 0 ? lead_map : elm_host::map_replicas(const_cast<elm_map*>(lead_map))[(size_t)r - 1]; }

int map_call(elm_map* lead_map, int which, double arg) {
    elm_group* g = group_of(elm_host::map_ctx(lead_map));
    if ((int)elm_host::map_replicas(lead_map).size() != g->n - 1) return ELM_ERR_INVALID;
    std::lock_guard<std::mutex> call(g->call_mu);
    return run_all(g, [&](int r) {
        elm_map* m = const_cast<elm_map*>(rank_map(lead_map, r));
        return which == 0 ? elm_map_cal_voxel_cov_all(m) : which == 1 ? elm_map_cal_point_cov_all(m, arg) : elm_map_build_neighbourhoods(m);
    });
}

int scan_upload(elm_ctx* lead, const float* xyz, size_t n, elm_scan** out) {
    elm_group* g = group_of(lead);
    if (!out || (!xyz && n)) return ELM_ERR_INVALID;
    *out = nullptr;
    std::lock_guard<std::mutex> call(g->call_mu);
    std::vector<float> ordered;
    spatial_order(xyz, n, ordered);
    std::vector<elm_scan*> sh(g->n, nullptr);
    const int rc = run_all(g, [&](int r) {
        const size_t lo = n * (size_t)r / (size_t)g->n, hi = n * ((size_t)r + 1) / (size_t)g->n;
        return elm_scan_upload(g->ctx[r], ordered.data() + 3 * lo, hi - lo, n, &sh[r]);
    });
    if (rc != ELM_OK) {
        for (elm_scan* s : sh) if (s) elm_scan_destroy(s);
        return rc;
    }
    elm_host::scan_shards(sh[0]).assign(sh.begin() + 1, sh.end());
    *out = sh[0];
    return ELM_OK;
}

// the ranks solved the same all-reduced sums: their results are the same bits (anything else is a broken exchange); the counts of a
// result (point_iterations, n_corr_last, the work counters) are whole-scan figures on every rank
static bool same_result(const elm_reg_result& a, const elm_reg_result& b) {
    return memcmp(a.T, b.T, sizeof(a.T)) == 0 && a.iterations == b.iterations && a.is_success == b.is_success && a.gate == b.gate;
}

static std::string describe_difference(const elm_reg_result& a, const elm_reg_result& b) {
    double d = 0.0;
    for (int k = 0; k < 16; ++k) d = std::max(d, fabs(a.T[k] - b.T[k]));
    char t[256];
    snprintf(t, sizeof(t), "iterations %d / %d, is_success %d / %d, gate %d / %d, n_corr_last %.0f / %.0f, max |dT| %.3g", a.iterations, b.iterations, a.is_success,
             b.is_success, a.gate, b.gate, a.n_corr_last, b.n_corr_last, d);
    return t;
}

int reg(elm_ctx* lead, const elm_map* map, const float* scan_xyz, size_t n, const double T0[16], const elm_reg_config* cfg, double T_out[16],
        int* is_success, double* fitness_score, double local_cov[36], elm_reg_result* result, elm_iter_trace* trace) {
    elm_group* g = group_of(lead);
    if (!map || !T0 || !cfg || (!scan_xyz && n)) return ELM_ERR_INVALID;
    if (elm_host::map_ctx(map) != lead || (int)elm_host::map_replicas(const_cast<elm_map*>(map)).size() != g->n - 1) return ELM_ERR_INVALID;
    std::lock_guard<std::mutex> call(g->call_mu);
    // RunRegister's per-call contract: the caller's point order, cut into contiguous shards (a LiDAR driver's order is azimuthal: the
    // shards are sectors; elm_scan_upload orders a resident scan spatially first)
    std::vector<elm_reg_result> res(g->n);
    const int rc = run_all(g, [&](int r) {
        const size_t lo = n * (size_t)r / (size_t)g->n, hi = n * ((size_t)r + 1) / (size_t)g->n;
        return elm_register_shard(g->ctx[r], rank_map(map, r), scan_xyz ? scan_xyz + 3 * lo : nullptr, hi - lo, n, T0, cfg, &res[r], r == 0 ? trace : nullptr, r != 0);
    });
    if (rc != ELM_OK) return rc;
    for (int r = 1; r < g->n; ++r) {
        if (!same_result(res[0], res[r])) {
            elm_host::ctx_set_error(lead, "device group: rank " + std::to_string(r) + " returned another pose than rank 0 (broken exchange): " + describe_difference(res[0], res[r]));
            return ELM_ERR_COMM;
        }
    }
    if (T_out) memcpy(T_out, res[0].T, sizeof(res[0].T));
    if (is_success) *is_success = res[0].is_success;
    if (fitness_score && res[0].is_success) *fitness_score = res[0].fitness_score; // untouched on failure, like the reference
    if (local_cov) memcpy(local_cov, res[0].local_cov, sizeof(res[0].local_cov));
    if (result) *result = res[0];
    return ELM_OK;
}

int reg_batch(elm_ctx* lead, const elm_map* map, elm_scan* const* scans, int count, const double* T0, const elm_reg_config* cfg, int slots,
              elm_reg_result* results, elm_iter_trace* trace) {
    elm_group* g = group_of(lead);
    if (!map || !scans || count <= 0 || !T0 || !cfg || !results || slots < 0) return ELM_ERR_INVALID;
    if (elm_host::map_ctx(map) != lead || (int)elm_host::map_replicas(const_cast<elm_map*>(map)).size() != g->n - 1) return ELM_ERR_INVALID;
    for (int b = 0; b < count; ++b)
        if (!scans[b] || elm_host::scan_ctx(scans[b]) != lead || (int)elm_host::scan_shards(scans[b]).size() != g->n - 1) {
            elm_host::ctx_set_error(lead, "device group: a scan was not uploaded through the group's lead context (elm_scan_upload)");
            return ELM_ERR_INVALID;
        }
    std::lock_guard<std::mutex> call(g->call_mu);
    std::vector<std::vector<elm_reg_result>> res(g->n);
    const int rc = run_all(g, [&](int r) {
        std::vector<elm_scan*> mine((size_t)count);
        for (int b = 0; b < count; ++b) mine[(size_t)b] = r == 0 ? scans[b] : elm_host::scan_shards(scans[b])[(size_t)r - 1];
        elm_reg_result* out = results;
        if (r != 0) { res[(size_t)r].resize((size_t)count); out = res[(size_t)r].data(); }
        return slots > 0 ? elm_register_stream(g->ctx[r], rank_map(map, r), mine.data(), count, T0, cfg, slots, out, r == 0 ? trace : nullptr)
                         : elm_register_batch(g->ctx[r], rank_map(map, r), mine.data(), count, T0, cfg, out, r == 0 ? trace : nullptr);
    });
    if (rc != ELM_OK) return rc;
    for (int r = 1; r < g->n; ++r)
        for (int b = 0; b < count; ++b) {
            const elm_reg_result& q = res[(size_t)r][(size_t)b];
            if (!same_result(results[b], q)) {
                elm_host::ctx_set_error(lead, "device group: rank " + std::to_string(r) + " returned another pose than rank 0 for registration " + std::to_string(b) +
                                                  " (broken exchange): " + describe_difference(results[b], q));
                return ELM_ERR_COMM;
            }
        }
    return ELM_OK;
}
} // namespace elm_multi

// ---- C ABI ----------------------------------------------------------------------------------------------------------------------------
extern "C" int elm_ctx_create_multi(const int* device_ids, int n, elm_ctx** out) {
    if (!out || !device_ids || n < 1 || n > 64) return ELM_ERR_INVALID;
    *out = nullptr;
    elm_group* g = new elm_group();
    g->n = n;
    g->devices.assign(device_ids, device_ids + n);
    g->rc.assign((size_t)n, ELM_OK);
    for (int r = 0; r < n; ++r) {
        elm_ctx* c = nullptr;
        const int rc = elm_ctx_create(device_ids[r], &c);
        if (rc != ELM_OK) {
            for (elm_ctx* q : g->ctx) elm_ctx_destroy(q);
            delete g;
            return rc;
        }
        g->ctx.push_back(c);
    }
    elm_ctx* lead = g->ctx[0];
    const char* force = getenv("ELM_GROUP_EXCHANGE"); // rccl | host (default: RCCL when every rank has a device of its own)
    if (n == 1 && !(force && strcmp(force, "rccl") == 0)) { // a group of one is a plain context
        delete g;                                           // (forced RCCL: a one-rank GROUP -- worker thread, communicator and all-reduce of the
        *out = lead;                                        //  in-process path on a single GPU: what a one-GPU box can run of it)
        return ELM_OK;
    }
    const bool distinct = std::set<int>(g->devices.begin(), g->devices.end()).size() == (size_t)n;
    g->exchange = (force && strcmp(force, "host") == 0) ? 2 : (force && strcmp(force, "rccl") == 0) ? 1 : (distinct ? 1 : 2);
    g->hx.out.assign((size_t)n, nullptr);
    g->hx.sum.assign((size_t)n, nullptr);
    g->hx.cap.assign((size_t)n, 0);
    g->hx.count.assign((size_t)n, 0);
    g->hook_user.resize((size_t)n);
    for (int r = 0; r < n; ++r) g->hook_user[(size_t)r] = {g, r};
    for (int r = 0; r < n; ++r) g->workers.emplace_back(worker_main, g, r);
    int rc = ELM_OK;
    if (g->exchange == 1) {
        unsigned char id[ELM_COMM_ID_BYTES];
        rc = elm_comm_get_unique_id(id);
        // ncclCommInitRank from the N worker threads at once: the call returns when all ranks have joined
        if (rc == ELM_OK) rc = run_all(g, [&](int r) { return elm_comm_init(g->ctx[r], r, n, id); });
        if (rc == ELM_OK)
            rc = run_all(g, [&](int r) { // what the communicators say about themselves
                int cr = -1, cn = 0;
                const int q = elm_comm_info(g->ctx[r], &cr, &cn);
                return q != ELM_OK ? q : ((cr == r && cn == n) ? ELM_OK : ELM_ERR_COMM);
            });
    } else {
        rc = run_all(g, [&](int r) { return elm_comm_set_hook(g->ctx[r], host_exchange, &g->hook_user[(size_t)r]); });
    }
    if (rc != ELM_OK) {
        elm_multi::destroy(g); // (the lead has no group pointer yet: it is destroyed like the others below)
        elm_ctx_destroy(lead);
        return rc;
    }
    elm_host::ctx_group(lead) = g;
    *out = lead;
    return ELM_OK;
}

extern "C" int elm_ctx_group_info(elm_ctx* ctx, int* n_ranks, int* exchange, int* device_ids, int cap) {
    if (!ctx || !n_ranks) return ELM_ERR_INVALID;
    elm_group* g = elm_host::ctx_group(ctx);
    *n_ranks = g ? g->n : 1;
    if (exchange) *exchange = g ? g->exchange : 0;
    if (device_ids)
        for (int r = 0; r < std::min(cap, g ? g->n : 1); ++r) device_ids[r] = g ? g->devices[(size_t)r] : elm_host::ctx_device(ctx);
    return ELM_OK;
}
