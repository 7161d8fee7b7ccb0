/* stream_harness.c -- the two nodes of launch/ELiMaLoc.launch as a plain-C caller of the C ABI (no ROS, no C++):
 * pcm_matching's CallbackPointCloud (one call: filter, deskew, pose sync, downsample, VGICP on the GPU) feeding the CPU EKF's
 * PCM update, whose IMU-rate odometry deskews and seeds the next scan.  A parked vehicle: 2 s of 200 Hz IMU, 10 Hz LiDAR.
 *   gcc -std=c11 -Iinclude examples/stream_harness.c -Lelimaloc_amd -lelimaloc_hip -lm -Wl,-rpath,$PWD/elimaloc_amd -o stream_harness
 * Optional arguments: localization.ini calibration.ini (the reference's own files).  Needs an MI355X to run.
 * Sizes from the environment (defaults: a 90 000-point map, 20 000-point scans, 19 scans): ELM_HARNESS_GRID (ground lattice edge:
 * 3000 -> a 9 M-point map), ELM_HARNESS_SCAN (raw points per LiDAR message), ELM_HARNESS_SCANS.  Prints the wall time of the
 * per-scan work (callback + EKF update) -- the config-5 latency as a C caller sees it. */
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
    return (v && atoi(v) > 0) ? atoi(v) : def;
}
static int cmp_double(const void* a, const void* b) { return (*(const double*)a > *(const double*)b) - (*(const double*)a < *(const double*)b); }

static float frand(unsigned* s) { /* xorshift, uniform in [-1, 1) */
    *s ^= *s << 13; *s ^= *s >> 17; *s ^= *s << 5;
    return (float)((*s >> 8) * (1.0 / 8388608.0) - 1.0);
}

int main(int argc, char** argv) {
    elm_ctx* ctx = NULL;
    CHECK(elm_ctx_create(0, &ctx));
    elm_pcm_node_config node;
    elm_reg_config reg;
    elm_ekf_config ekf_cfg;
    elm_pcm_node_config_default(&node);
    elm_reg_config_default(&reg);
    elm_ekf_config_default(&ekf_cfg);
    reg.icp_method = ELM_VGICP;
    if (argc >= 3) { /* ProcessINI of both nodes on the reference's own files */
        CHECK(elm_load_pcm_config(argv[1], argv[2], &node, &reg));
        CHECK(elm_load_ekf_config(argv[1], &ekf_cfg));
    }
    /* map: a jittered ground lattice + one wall (float32, file order) */
    unsigned seed = 12345u;
    const int G = env_int("ELM_HARNESS_GRID", 300);
    size_t n_map = 0;
    float* map_xyz = (float*)malloc(sizeof(float) * 3 * ((size_t)G * G + (size_t)G * 30));
    for (int i = 0; i < G; ++i)
        for (int j = 0; j < G; ++j) {
            map_xyz[3 * n_map] = (i - G / 2 + 0.5f) * 0.2f + 0.004f * frand(&seed);
            map_xyz[3 * n_map + 1] = (j - G / 2 + 0.5f) * 0.2f + 0.004f * frand(&seed);
            map_xyz[3 * n_map + 2] = 0.3f + 0.004f * frand(&seed);
            ++n_map;
        }
    for (int i = 0; i < G; ++i)
        for (int k = 0; k < 30; ++k) {
            map_xyz[3 * n_map] = (i - G / 2 + 0.5f) * 0.2f + 0.004f * frand(&seed);
            map_xyz[3 * n_map + 1] = 12.1f + 0.004f * frand(&seed);
            map_xyz[3 * n_map + 2] = 0.4f + 0.2f * k + 0.004f * frand(&seed);
            ++n_map;
        }
    elm_map* map = NULL;
    CHECK(elm_map_build(ctx, map_xyz, n_map, node.pcm_voxel_size, node.pcm_voxel_max_point, &map));
    if (reg.icp_method == ELM_VGICP || reg.icp_method == ELM_AVGICP) CHECK(elm_map_cal_voxel_cov_all(map));
    if (reg.icp_method == ELM_GICP) CHECK(elm_map_cal_point_cov_all(map, reg.gicp_cov_search_dist));

    elm_ekf* ekf = NULL;
    CHECK(elm_ekf_create(&ekf_cfg, &ekf));
    /* the vehicle is parked at (1.5, -2.0, 0.3), yaw 0.3 rad; lidar = ego * tf_ego_to_lidar */
    const double yaw = 0.3, ego[3] = {1.5, -2.0, 0.3};
    const double q0[4] = {0.0, 0.0, sin(yaw / 2), cos(yaw / 2)};
    const double cr = cos(yaw), sr = sin(yaw);
    enum { MAX_ODOM = 4096 };
    const int N_SCAN = env_int("ELM_HARNESS_SCAN", 20000), N_MSG = env_int("ELM_HARNESS_SCANS", 19);
    static double imu4[MAX_ODOM * 4], odom14[MAX_ODOM * 14];
    double* scan_ms = (double*)malloc(sizeof(double) * (size_t)(N_MSG + 1));
    size_t n_imu = 0, n_odom = 0;
    float* scan = (float*)malloc(sizeof(float) * 3 * N_SCAN);
    float* ptime = (float*)malloc(sizeof(float) * N_SCAN);
    int updated = 0, predicted = 0, published = 0, n_pub = 0;
    const double t0 = 100.0, gyro[3] = {0, 0, 0}, acc[3] = {0, 0, 9.81};
    elm_pcm_scan_output out;
    for (int k = 0; k <= 20 * (N_MSG + 1); ++k) {
        const double t = t0 + k / 200.0;
        if (k == 2) CHECK(elm_ekf_update_pcm_odom(ekf, t, ego, q0, (double[36]){0}, ELM_GNSS_PCM_INIT, &updated)); /* init pose */
        CHECK(elm_ekf_predict_imu(ekf, t, gyro, acc, &predicted));
        elm_ego_state es;
        CHECK(elm_ekf_publish(ekf, &es)); /* EkfLocalization::PublishInThread -> odometry message -> pcm_matching's queues */
        if (n_imu < MAX_ODOM) { imu4[4 * n_imu] = t; memcpy(&imu4[4 * n_imu + 1], gyro, sizeof gyro); ++n_imu; }
        if (n_odom < MAX_ODOM && (fabs(es.x_m) > 1e-9 && fabs(es.y_m) > 1e-9)) {
            double* o = &odom14[14 * n_odom++];
            const double cy = cos(es.yaw_rad / 2), sy = sin(es.yaw_rad / 2), cp = cos(es.pitch_rad / 2), sp = sin(es.pitch_rad / 2),
                         cR = cos(es.roll_rad / 2), sR = sin(es.roll_rad / 2);
            o[0] = es.timestamp; o[1] = es.x_m; o[2] = es.y_m; o[3] = es.z_m;
            o[4] = sR * cp * cy - cR * sp * sy; o[5] = cR * sp * cy + sR * cp * sy; o[6] = cR * cp * sy - sR * sp * cy; o[7] = cR * cp * cy + sR * sp * sy;
            o[8] = es.vx; o[9] = es.vy; o[10] = es.vz; o[11] = es.roll_vel; o[12] = es.pitch_vel; o[13] = es.yaw_vel;
        }
        if (k > 10 && k % 20 == 0) { /* a LiDAR message: map points seen from the parked lidar + noise, time ramp -0.1 .. 0 s */
            const double* tf = node.tf_ego_to_lidar; /* column-major */
            for (int i = 0; i < N_SCAN; ++i) {
                const size_t m = ((size_t)i * 7919u + (size_t)k * 104729u) % n_map;
                const double wx = map_xyz[3 * m] - ego[0], wy = map_xyz[3 * m + 1] - ego[1], wz = map_xyz[3 * m + 2] - ego[2];
                const double ex = cr * wx + sr * wy - tf[12], ey = -sr * wx + cr * wy - tf[13], ez = wz - tf[14]; /* ego frame - lidar offset */
                scan[3 * i] = (float)(tf[0] * ex + tf[1] * ey + tf[2] * ez) + 0.01f * frand(&seed);       /* R_lidar^T */
                scan[3 * i + 1] = (float)(tf[4] * ex + tf[5] * ey + tf[6] * ez) + 0.01f * frand(&seed);
                scan[3 * i + 2] = (float)(tf[8] * ex + tf[9] * ey + tf[10] * ez) + 0.01f * frand(&seed);
                ptime[i] = (float)(-0.1 + 0.1 * (i + 0.5) / N_SCAN);
            }
            ptime[N_SCAN - 1] = 0.f;
            const double t_begin = now_ms();
            CHECK(elm_pcm_callback_point_cloud(ctx, map, &node, &reg, scan, ptime, N_SCAN, t - 0.005 + node.lidar_time_delay, imu4, n_imu,
                                               odom14, n_odom, &out, &published));
            if (published) {
                ++n_pub;
                /* nav_msgs/Odometry: position, orientation (from the ego pose's rotation: yaw only here), covariance */
                const double pyaw = atan2(out.pose_ego[1], out.pose_ego[0]);
                const double pos[3] = {out.pose_ego[12], out.pose_ego[13], out.pose_ego[14]}, q[4] = {0, 0, sin(pyaw / 2), cos(pyaw / 2)};
                CHECK(elm_ekf_update_pcm_odom(ekf, out.time_scan_end, pos, q, out.covariance, ELM_GNSS_PCM, &updated));
                if (n_pub <= N_MSG) scan_ms[n_pub - 1] = now_ms() - t_begin;
            }
        }
    }
    elm_ekf_state st;
    CHECK(elm_ekf_get_state(ekf, &st));
    const double err = sqrt((st.x[0] - ego[0]) * (st.x[0] - ego[0]) + (st.x[1] - ego[1]) * (st.x[1] - ego[1]));
    printf("scans published %d/%d  ekf xy error %.4f m  iterations(last) %d  fitness %.4f\n", n_pub, N_MSG, err, out.result.iterations, out.fitness_score);
    if (n_pub > 4) { /* the first scans carry one-time allocations: steady state = everything after the third */
        const int n = n_pub - 3;
        double sum = 0.0;
        for (int i = 0; i < n; ++i) sum += scan_ms[3 + i];
        qsort(scan_ms + 3, (size_t)n, sizeof(double), cmp_double);
        printf("per scan (callback + EKF update, %zu-point map, %d raw points, %zu after the node's filters): mean %.3f ms  median %.3f ms  max %.3f ms  (n = %d)\n",
               n_map, N_SCAN, (size_t)out.n_source, sum / n, scan_ms[3 + n / 2], scan_ms[3 + n - 1], n);
    }
    elm_ekf_destroy(ekf);
    elm_map_destroy(map);
    elm_ctx_destroy(ctx);
    free(map_xyz); free(scan); free(ptime); free(scan_ms);
    return (n_pub >= N_MSG - 4 && err < 0.05) ? 0 : 1;
}
