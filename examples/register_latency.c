/* register_latency.c -- what ONE Registration::RunRegister-equivalent call costs a plain-C caller of the C ABI (no Python, no C++):
 * elm_register on a pageable buffer (a ROS message's data), on a page-locked buffer (elm_host_alloc), and elm_register_batch on a scan
 * that is already resident -- median / p10 / p90 wall time over ELM_LAT_CALLS calls cycling through eight scans.
 *   gcc -O2 -std=c11 -Iinclude examples/register_latency.c -Lelimaloc_amd -lelimaloc_hip -lm -Wl,-rpath,$PWD/elimaloc_amd -o register_latency
 * Sizes from the environment (defaults: a 9 M-point map -- 3000 x 3000 ground lattice + a wall --, 131 072-point scans, P2P):
 * ELM_HARNESS_GRID, ELM_HARNESS_SCAN, ELM_LAT_CALLS, ELM_LAT_METHOD.  Needs an MI355X to run. */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "elimaloc_hip.h"

#define CHECK(call)                                                                    \
    do {                                                                               \
        int rc_ = (call);                                                              \
        if (rc_ != ELM_OK) {                                                           \
            fprintf(stderr, "%s -> %s (%s)\n", #call, elm_strerror(rc_), ctx ? elm_last_error(ctx) : ""); \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static int env_int(const char* name, int def) {
    const char* v = getenv(name);
    return (v && atoi(v) >= 0 && v[0]) ? atoi(v) : def;
}
static int cmp_double(const void* a, const void* b) { return (*(const double*)a > *(const double*)b) - (*(const double*)a < *(const double*)b); }
static float frand(unsigned* s) { /* xorshift, uniform in [-1, 1) */
    *s ^= *s << 13; *s ^= *s >> 17; *s ^= *s << 5;
    return (float)((*s >> 8) * (1.0 / 8388608.0) - 1.0);
}
static void report(const char* what, double* t, int n, double iters) {
    qsort(t, (size_t)n, sizeof(double), cmp_double);
    printf("%-46s median %.4f ms  p10 %.4f  p90 %.4f  (%d calls, %.2f iterations on average)\n", what, t[n / 2], t[n / 10], t[(9 * n) / 10], n, iters);
}

int main(void) {
    elm_ctx* ctx = NULL;
    CHECK(elm_ctx_create(0, &ctx));
    elm_reg_config reg;
    elm_reg_config_default(&reg);
    reg.icp_method = env_int("ELM_LAT_METHOD", ELM_P2P);
    const int G = env_int("ELM_HARNESS_GRID", 3000), NS = env_int("ELM_HARNESS_SCAN", 131072), calls = env_int("ELM_LAT_CALLS", 200);
    /* map: a jittered ground lattice (0.2 m pitch) + one wall, float32 */
    unsigned seed = 12345u;
    const size_t n_ground = (size_t)G * G, n_wall = (size_t)G * 30, n_map = n_ground + n_wall;
    float* map = (float*)malloc(n_map * 3 * sizeof(float));
    if (!map) return 3;
    size_t k = 0;
    for (int i = 0; i < G; ++i)
        for (int j = 0; j < G; ++j, ++k) {
            map[3 * k] = (i - G / 2 + 0.5f) * 0.2f + 0.004f * frand(&seed);
            map[3 * k + 1] = (j - G / 2 + 0.5f) * 0.2f + 0.004f * frand(&seed);
            map[3 * k + 2] = 0.3f + 0.004f * frand(&seed);
        }
    for (int i = 0; i < G; ++i)
        for (int h = 0; h < 30; ++h, ++k) {
            map[3 * k] = (i - G / 2 + 0.5f) * 0.2f + 0.004f * frand(&seed);
            map[3 * k + 1] = 10.5f + 0.004f * frand(&seed);
            map[3 * k + 2] = 1.1f + 0.2f * h + 0.004f * frand(&seed);
        }
    elm_map* m = NULL;
    CHECK(elm_map_build(ctx, map, n_map, 1.0, 30, &m));
    if (reg.icp_method == ELM_GICP) CHECK(elm_map_cal_point_cov_all(m, reg.gicp_cov_search_dist));
    if (reg.icp_method >= ELM_VGICP) CHECK(elm_map_cal_voxel_cov_all(m));
    CHECK(elm_map_build_neighbourhoods(m));
    /* eight scans: NS ground points within 60 m of a sensor 1.8 m above the ground, seen from that sensor; the initial guess is off by
     * a few centimetres */
    enum { NSCAN = 8 };
    float* pageable[NSCAN];
    float* pinned[NSCAN];
    elm_scan* resident[NSCAN];
    double T0[NSCAN][16];
    for (int s = 0; s < NSCAN; ++s) {
        const float sx = 7.0f * s - 20.0f, sy = -3.0f * s + 5.0f, sz = 0.3f + 1.8f;
        pageable[s] = (float*)malloc((size_t)NS * 3 * sizeof(float));
        pinned[s] = (float*)elm_host_alloc((size_t)NS * 3 * sizeof(float));
        if (!pageable[s] || !pinned[s]) return 3;
        for (int p = 0; p < NS; ++p) {
            const float r = 60.0f * sqrtf(0.5f * (frand(&seed) + 1.0f)), a = 3.14159265f * frand(&seed);
            const float gx = sx + r * cosf(a), gy = sy + r * sinf(a);
            /* the nearest lattice point's coordinates, as the map holds them up to its jitter */
            const int i = (int)floorf(gx / 0.2f + G / 2), j = (int)floorf(gy / 0.2f + G / 2);
            const size_t q = (size_t)(i < 0 ? 0 : i >= G ? G - 1 : i) * G + (size_t)(j < 0 ? 0 : j >= G ? G - 1 : j);
            pageable[s][3 * p] = map[3 * q] - sx + 0.01f * frand(&seed);
            pageable[s][3 * p + 1] = map[3 * q + 1] - sy + 0.01f * frand(&seed);
            pageable[s][3 * p + 2] = map[3 * q + 2] - sz + 0.01f * frand(&seed);
        }
        memcpy(pinned[s], pageable[s], (size_t)NS * 3 * sizeof(float));
        CHECK(elm_scan_upload(ctx, pageable[s], (size_t)NS, (size_t)NS, &resident[s]));
        memset(T0[s], 0, sizeof(T0[s]));
        T0[s][0] = T0[s][5] = T0[s][10] = T0[s][15] = 1.0;
        T0[s][12] = sx + 0.06; T0[s][13] = sy - 0.04; T0[s][14] = sz + 0.02;
    }
    double* t = (double*)malloc((size_t)calls * sizeof(double));
    double T[16], cov[36], fit = 0.0;
    int ok = 0;
    elm_reg_result res;
    for (int mode = 0; mode < 3; ++mode) {
        double iters = 0.0;
        for (int c = -16; c < calls; ++c) { /* 16 warm-up calls */
            const int s = (c + 16) % NSCAN;
            const double t0 = now_ms();
            if (mode == 0) CHECK(elm_register(ctx, m, pageable[s], (size_t)NS, T0[s], &reg, T, &ok, &fit, cov, &res, NULL));
            else if (mode == 1) CHECK(elm_register(ctx, m, pinned[s], (size_t)NS, T0[s], &reg, T, &ok, &fit, cov, &res, NULL));
            else CHECK(elm_register_batch(ctx, m, &resident[s], 1, T0[s], &reg, &res, NULL));
            if (c >= 0) { t[c] = now_ms() - t0; iters += res.iterations; }
            if (!res.is_success || fabs(res.T[12] - (7.0 * s - 20.0)) > 0.05) { fprintf(stderr, "registration %d did not converge onto its sensor\n", s); return 4; }
        }
        report(mode == 0 ? "elm_register, pageable source" : mode == 1 ? "elm_register, page-locked source (elm_host_alloc)" : "elm_register_batch, resident scan", t, calls, iters / calls);
    }
    for (int s = 0; s < NSCAN; ++s) { elm_scan_destroy(resident[s]); elm_host_free(pinned[s]); free(pageable[s]); }
    elm_map_destroy(m);
    elm_ctx_destroy(ctx);
    free(map); free(t);
    return 0;
}
