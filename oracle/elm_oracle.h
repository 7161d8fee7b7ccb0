/*
 * elm_oracle.h -- C interface of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  This library is a CPU restatement of the
 * reference's pcm_matching registration hot path (ELiMaLoc @ 2025-02-27).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / reported CPU baseline.  Nothing under elimaloc_amd/
 * may include, link or dlopen it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
 * this path and cannot be compiled here (Eigen3 / oneTBB / PCL / ROS absent), so
 * the oracle is pinned only by analytic known-answer tests and an independent
 * numpy float64 re-derivation (tests/np_ref.py), not by reference outputs.
 *
 * Matrices cross this interface column-major (Eigen's default storage).
 */
#ifndef ELM_ORACLE_H
#define ELM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_map orc_map; /* VoxelHashMap restatement (vhm.hpp:89-335) */

/* reg.hpp:60 */
enum { ORC_P2P = 0, ORC_GICP = 1, ORC_VGICP = 2, ORC_AVGICP = 3 };

/* POD mirror of the RegistrationConfig fields the path reads (reg.hpp:62-85). */
typedef struct orc_config {
    int32_t icp_method;
    int32_t max_iteration;
    int32_t max_thread; /* threads for the correspondence search (TBB arena size in the reference) */
    int32_t use_radar_cov; /* reg.hpp:186-217, reg.cpp:109-111, 188-190, 302-305 (off in the shipped localization.ini:105) */
    double max_search_dist;
    double lm_lambda;
    double icp_termination_threshold_m;
    double min_overlap_ratio;
    double max_fitness_score;
    double gicp_cov_search_dist;
    double range_variance_m;       /* reg.hpp:80-82: only read with use_radar_cov */
    double azimuth_variance_deg;
    double elevation_variance_deg;
} orc_config;

#define ORC_MAX_ITER_TRACE 64

/* Per-iteration trace of RunRegister (one entry per executed iteration). */
typedef struct orc_iter_trace {
    int64_t n_corr;      /* correspondences returned by the search (reg.cpp:349) */
    double JTJ[36];      /* column-major, before damping */
    double JTr[6];
    double residual_sum; /* numerator of d_fitness_score_ */
    double x[6];         /* LM step */
    double step_norm;    /* angle + |t| (reg.cpp:384) */
    double T[16];        /* pose after the update (reg.cpp:378), column-major */
    int64_t n_cand;      /* candidate map points / voxel means distance-tested this iteration */
    int64_t n_occ;       /* occupied neighbour voxels visited this iteration */
} orc_iter_trace;

typedef struct orc_result {
    double T[16];
    int32_t is_success;
    int32_t iterations;  /* executed iterations (i_iteration) */
    int32_t gate;        /* 0 none, 1 empty map, 2 overlap ratio, 3 fitness */
    int32_t _pad;
    double fitness;      /* d_fitness_score_ (written even on failure, unlike fitness_score) */
    double local_cov[36];
    double elapsed_ms;         /* span of reg.cpp:307-394 */
    double correspondence_ms;  /* sum over iterations of reg.cpp:316-341 */
    orc_iter_trace iters[ORC_MAX_ITER_TRACE];
} orc_result;

orc_map* orc_map_create(double voxel_size, int max_points_per_voxel);
void orc_map_destroy(orc_map*);
/* Pcl2PointStruct (pcm.hpp:205-220) + AddPoints (vhm.cpp:270-285): xyz float32 -> double */
void orc_map_add_points(orc_map*, const float* xyz, size_t n);
void orc_map_cal_voxel_cov_all(orc_map*, int threads);              /* vhm.hpp:183-193 */
void orc_map_cal_point_cov_all(orc_map*, double dist, int threads); /* vhm.hpp:252-257 */
size_t orc_map_num_points(const orc_map*);
size_t orc_map_num_voxels(const orc_map*);
int orc_map_empty(const orc_map*);
/* Pointcloud() (vhm.cpp:245-255) in container order: pose xyz (3 doubles), cov (9, col-major), mean (3) per point */
size_t orc_map_pointcloud(const orc_map*, double* xyz, double* cov9, double* mean3, size_t cap);
/* every voxel: key (3 ints as stored), n points, cov (9, col-major), mean (3) */
size_t orc_map_voxels(const orc_map*, int32_t* key3, int32_t* npts, double* cov9, double* mean3, size_t cap);
/* FindGroundHeight (vhm.hpp:285-322) */
int orc_map_find_ground_height(const orc_map*, double px, double py, double* ground_z);

/* GetCorrespondencePoints for world-frame query points (vhm.cpp:31-88):
 * accepted[i] in {0,1}; tgt_xyz = matched map point (or the (0,0,0) default). */
void orc_nearest_points(const orc_map*, const double* q_xyz, size_t n, double max_dist, int threads,
                        uint8_t* accepted, double* tgt_xyz, double* d2);
/* GetCorrespondencesCov (vhm.cpp:90-151): nearest voxel mean */
void orc_nearest_voxel(const orc_map*, const double* q_xyz, size_t n, double max_dist, int threads,
                       uint8_t* accepted, double* mean_xyz, double* cov9);

/* RunRegister (reg.cpp:274-418). scan_xyz: sensor-frame float32 points (pose == local). */
void orc_register(const orc_map*, const float* scan_xyz, size_t n, const double T0[16],
                  const orc_config* cfg, orc_result* out);

/* VoxelDownsample (vhm.hpp:260-283): writes the kept input indices sorted ascending
 * (the reference's output ORDER is unordered_map iteration order; only the set is contractual). */
size_t orc_voxel_downsample(const float* xyz, size_t n, double voxel_size, int64_t* keep_idx);

/* ---- deskew (pcm.cpp:467-824) ---- */
typedef struct orc_deskew_tables {
    double time_scan_cur;  /* d_time_scan_cur_ */
    double time_scan_end;  /* d_time_scan_end_ */
    int32_t imu_pointer_cur; /* index of the last valid IMU table entry */
    int32_t run_deskew;
    float odom_incre_x, odom_incre_y, odom_incre_z;
    int32_t odom_available;
    const double* imu_time;  /* [imu_pointer_cur+1] */
    const double* imu_rot_x;
    const double* imu_rot_y;
    const double* imu_rot_z;
} orc_deskew_tables;

/* DeskewPoint loop (pcm.cpp:498-525, 780-824): xyz_in[3n], rel_time[n] (already rebased) -> xyz_out[3n] */
void orc_deskew_points(const float* xyz_in, const float* rel_time, size_t n, const orc_deskew_tables* tab,
                       float* xyz_out);

/* ImuDeskewInfo (pcm.cpp:533-585): imu samples (t, wx, wy, wz) already in the ego frame and already
 * trimmed/ordered as the deque would be; fills tables, returns b_is_imu_available_. cap >= 2000. */
int orc_imu_deskew_info(const double* imu_t, const double* imu_w_xyz, size_t n_imu, double scan_cur,
                        double scan_end, double* tab_time, double* tab_rx, double* tab_ry, double* tab_rz,
                        int32_t* imu_pointer_cur);

/* OdomDeskewInfo (pcm.cpp:587-729). odom rows: t, px,py,pz, qx,qy,qz,qw, vx,vy,vz, wx,wy,wz (14 doubles).
 * Returns b_is_odom_available_; fills the float increments. */
int orc_odom_deskew_info(const double* odom14, size_t n_odom, double scan_cur, double scan_end,
                         float* incre_xyz);

/* caller glue around RunRegister */
size_t orc_filter_points_by_distance(const float* xyz, size_t n, double max_dist, int64_t* keep_idx); /* pcm.cpp:451-465 */
int orc_get_interpolated_pose(const double* odom14, size_t n_odom, double d_cur_time, float T_out[16]); /* pcm.cpp:933-1045 */
void orc_shape_odom_covariance(const double local_cov[36], const double pose[16], double icp_pose_std_m,
                               double cov_out[36]);                                                  /* pcm.cpp:1082-1098 */

/* small exported helpers so the tests can pin the linear-algebra kit */
void orc_ldlt_solve6(const double A[36], const double b[6], double x[6]);
void orc_inverse6(const double A[36], double Ainv[36]);
void orc_jacobi_svd3(const double A[9], double U[9], double S[3], double V[9]);
/* eigenvector of the smallest eigenvalue as reg.cpp:89-91 takes it from SelfAdjointEigenSolver (col(0)); C col-major */
void orc_smallest_eigenvector(const double C[9], double n[3]);
void orc_angle_axis_to_matrix(const double rotvec[3], double R[9]);
double orc_matrix_to_angle(const double R[9]);

#ifdef __cplusplus
}
#endif
#endif
