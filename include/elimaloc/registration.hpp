// registration.hpp -- drop-in shim for pcm_matching/include/registration.hpp:60-230 over the C ABI.
#pragma once
#include <array>
#include <cstring>
#include <vector>

#include "voxel_hash_map.hpp"

typedef enum { P2P, GICP, VGICP, AVGICP } IcpMethod; // reg.hpp:60

// reg.hpp:62-85, same field names; Eigen members become plain arrays unless Eigen is present
struct RegistrationConfig {
    int i_max_thread = 10;
    IcpMethod icp_method = GICP;
    int voxel_search_method = 2;
    double gicp_cov_search_dist = 0.4;
    bool use_radar_cov = false;
    int max_iteration = 10;
    double max_search_dist = 5.0;
    double lm_lambda = 0.5;
    double icp_termination_threshold_m = 0.02;
    double min_overlap_ratio = 0.4;
    double max_fitness_score = 0.5;
    double doppler_trans_lambda = 0.5;
    double range_variance_m = 1.0;
    double azimuth_variance_deg = 0.4;
    double elevation_variance_deg = 0.4;
    std::array<double, 3> ego_to_lidar_trans{{0, 0, 0}};
    std::array<double, 9> ego_to_lidar_rot{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    std::array<double, 9> ego_to_imu_rot{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    bool b_debug_print = false;

    elm_reg_config c_struct() const {
        elm_reg_config c;
        elm_reg_config_default(&c);
        c.i_max_thread = i_max_thread; c.icp_method = (int)icp_method; c.voxel_search_method = voxel_search_method;
        c.use_radar_cov = use_radar_cov ? 1 : 0; c.max_iteration = max_iteration; c.b_debug_print = b_debug_print ? 1 : 0;
        c.gicp_cov_search_dist = gicp_cov_search_dist; c.max_search_dist = max_search_dist; c.lm_lambda = lm_lambda;
        c.icp_termination_threshold_m = icp_termination_threshold_m; c.min_overlap_ratio = min_overlap_ratio;
        c.max_fitness_score = max_fitness_score; c.doppler_trans_lambda = doppler_trans_lambda;
        c.range_variance_m = range_variance_m; c.azimuth_variance_deg = azimuth_variance_deg;
        c.elevation_variance_deg = elevation_variance_deg;
        std::memcpy(c.ego_to_lidar_trans, ego_to_lidar_trans.data(), sizeof(c.ego_to_lidar_trans));
        std::memcpy(c.ego_to_lidar_rot, ego_to_lidar_rot.data(), sizeof(c.ego_to_lidar_rot));
        std::memcpy(c.ego_to_imu_rot, ego_to_imu_rot.data(), sizeof(c.ego_to_imu_rot));
        return c;
    }
};

using Matrix4dArr = std::array<double, 16>; // column-major, like Eigen::Matrix4d::data()
using Matrix6dArr = std::array<double, 36>;

struct Registration {
    Registration() {}
    explicit Registration(RegistrationConfig config) { config_ = config; }
    void Init(RegistrationConfig config) { config_ = config; } // reg.hpp:104

    // reg.hpp:122-124 / reg.cpp:274-418.  Same out-parameter behaviour: fitness_score is written only on success, the
    // returned pose is the current estimate on failure (initial_guess for an empty map), local_cov = I unless GICP.
    Matrix4dArr RunRegister(const std::vector<PointStruct>& source_local, VoxelHashMap& voxel_map,
                            const Matrix4dArr& initial_guess, RegistrationConfig m_config, bool& is_success,
                            double& fitness_score, Matrix6dArr& local_cov) {
        std::vector<float> xyz(3 * source_local.size());
        for (size_t i = 0; i < source_local.size(); ++i)
            for (int k = 0; k < 3; ++k) xyz[3 * i + k] = (float)source_local[i].pose[k]; // TransformPoints reads .pose (reg.hpp:142)
        const elm_reg_config c = m_config.c_struct();
        Matrix4dArr T;
        int ok = 0;
        elm_reg_result res;
        elimaloc::check(elm_register(VoxelHashMap::ctx(), voxel_map.handle(), xyz.data(), source_local.size(), initial_guess.data(), &c,
                                     T.data(), &ok, &fitness_score, local_cov.data(), &res, nullptr),
                        VoxelHashMap::ctx(), "RunRegister");
        is_success = ok != 0;
        d_fitness_score_ = res.d_fitness;
        return T;
    }
#ifdef ELM_HAVE_EIGEN
    Eigen::Matrix4d RunRegister(const std::vector<PointStruct>& source_local, VoxelHashMap& voxel_map,
                                const Eigen::Matrix4d& initial_guess, RegistrationConfig m_config, bool& is_success,
                                double& fitness_score, Eigen::Matrix<double, 6, 6>& local_cov) {
        Matrix4dArr T0;
        Matrix6dArr cov;
        std::memcpy(T0.data(), initial_guess.data(), sizeof(double) * 16);
        Matrix4dArr T = RunRegister(source_local, voxel_map, T0, m_config, is_success, fitness_score, cov);
        std::memcpy(local_cov.data(), cov.data(), sizeof(double) * 36);
        Eigen::Matrix4d out;
        std::memcpy(out.data(), T.data(), sizeof(double) * 16);
        return out;
    }
#endif
    // reg.hpp:136-148 (host convenience for the debug clouds, pcm.cpp:308-313)
    static void TransformPoints(const Matrix4dArr& T, const std::vector<PointStruct>& points, std::vector<PointStruct>& o_points) {
        o_points = points;
        for (size_t i = 0; i < points.size(); ++i)
            for (int r = 0; r < 3; ++r)
                o_points[i].pose[r] = ((T[r] * points[i].pose[0] + T[4 + r] * points[i].pose[1]) + T[8 + r] * points[i].pose[2]) + T[12 + r];
    }

    RegistrationConfig config_;
    double d_fitness_score_ = 0.0;
};
