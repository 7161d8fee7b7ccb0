// registration.hpp -- drop-in shim for pcm_matching/include/registration.hpp:44-230 over the C ABI: the reference's
// type names, member names and signatures (Eigen-typed when Eigen is present, see linalg_types.hpp).
#pragma once
#include <algorithm>
#include <cstring>
#include <utility>
#include <stdexcept>
#include <iostream>
#include <vector>

#include "voxel_hash_map.hpp"

using Correspondences = std::vector<std::pair<PointStruct, PointStruct>>; // reg.hpp:58

typedef enum { P2P, GICP, VGICP, AVGICP } IcpMethod; // reg.hpp:60

// reg.hpp:62-85, same field names and types; the initialisers are the shipped defaults of config/localization.ini:80-105
// (the reference leaves the members uninitialised and fills them in ProcessINI, pcm.cpp:121-196)
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
    elimaloc::Vector3d ego_to_lidar_trans = elimaloc::Vector3d::Zero();
    elimaloc::Matrix3d ego_to_lidar_rot = elimaloc::Matrix3d::Identity();
    elimaloc::Matrix3d ego_to_imu_rot = elimaloc::Matrix3d::Identity();

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
        std::memcpy(c.ego_to_lidar_rot, ego_to_lidar_rot.data(), sizeof(c.ego_to_lidar_rot)); // column-major on both sides
        std::memcpy(c.ego_to_imu_rot, ego_to_imu_rot.data(), sizeof(c.ego_to_imu_rot));
        return c;
    }
};

struct Registration {
    Registration() {}
    Registration(RegistrationConfig config) { config_ = config; }
    void Init(RegistrationConfig config) { config_ = config; } // reg.hpp:104

    // reg.hpp:122-124 / reg.cpp:274-418.  Same out-parameter behaviour: fitness_score is written only on success, the
    // returned pose is the current estimate on failure (initial_guess for an empty map), local_cov = I unless GICP.
    // Like the reference's (d_fitness_score_ is a member) the object is not re-entrant.
    elimaloc::Matrix4d RunRegister(const std::vector<PointStruct>& source_local, const VoxelHashMap& voxel_map,
                                   const elimaloc::Matrix4d& initial_guess, RegistrationConfig m_config, bool& is_success,
                                   double& fitness_score, elimaloc::Matrix6d& local_cov) {
        scratch_xyz_.resize(3 * source_local.size()); // grows once; no allocation per scan after warm-up
        // The reference transforms .pose (reg.hpp:142) but builds J and the residual from .local (reg.cpp:34,41); its only producer
        // sets both to the same float32 point (pcm_matching.hpp:205-220).  The C ABI carries ONE float32 triple per point, so a
        // caller whose .local differs from .pose is refused instead of being answered with other sums than the reference's
        // (coordinates that are not float32 values are rounded to float32: INTEGRATION.md, interface differences).
        bool same = true;
        for (size_t i = 0; i < source_local.size(); ++i)
            for (int k = 0; k < 3; ++k) {
                const double a = source_local[i].local(k), b = source_local[i].pose(k);
                scratch_xyz_[3 * i + k] = (float)b;
                // a NaN return (the node does not remove them: FilterPointsByDistance keeps a point whose range is NaN) is "the same"
                // in both fields; the reference pairs such a point with nothing (every comparison with its distance is false)
                same = same && (a == b || (a != a && b != b));
            }
        if (!same) {
            // the reference reports nothing but is_success (no exceptions, no error codes): refuse the same way
            std::cerr << "[elimaloc] RunRegister: PointStruct.local must equal PointStruct.pose (pcm_matching.hpp:205-220; INTEGRATION.md, "
                         "interface differences) -- registration refused" << std::endl;
            is_success = false;
            local_cov = elimaloc::Matrix6d::Identity();
            return initial_guess;
        }
        const elm_reg_config c = m_config.c_struct();
        elimaloc::Matrix4d T;
        int ok = 0;
        elm_reg_result res;
        elimaloc::check(elm_register(VoxelHashMap::ctx(), voxel_map.handle(), scratch_xyz_.data(), source_local.size(), initial_guess.data(), &c,
                                     T.data(), &ok, &fitness_score, local_cov.data(), &res, nullptr),
                        VoxelHashMap::ctx(), "RunRegister");
        is_success = ok != 0;
        d_fitness_score_ = res.d_fitness;
        return T;
    }

    // reg.cpp:15-66 / 68-152 / 154-225: the step functions of the reference's public interface on pairs the CALLER holds (RunRegister
    // pairs and accumulates in one kernel and never calls them): source PointStruct::local against the targets' pose (P2P) /
    // covariance.mean + covariance.cov, around last_icp_pose; the pairs are accumulated on the device (elm_align_clouds_local) with
    // the reference's per-pair arithmetic.  d_fitness_score_ is set as the reference sets it.
    elimaloc::Matrix4d AlignCloudsLocal(std::vector<PointStruct>& source_global, const std::vector<PointStruct>& target_global,
                                        elimaloc::Matrix4d& last_icp_pose, double trans_th, RegistrationConfig m_config) {
        return Align(ELM_P2P, source_global, &target_global, nullptr, nullptr, last_icp_pose, trans_th, m_config);
    }
    elimaloc::Matrix4d AlignCloudsLocalPointCov(std::vector<PointStruct>& source_global, const std::vector<PointStruct>& target_global,
                                                elimaloc::Matrix6d& local_cov, elimaloc::Matrix4d& last_icp_pose, double trans_th,
                                                RegistrationConfig m_config) {
        return Align(ELM_GICP, source_global, &target_global, nullptr, &local_cov, last_icp_pose, trans_th, m_config);
    }
    elimaloc::Matrix4d AlignCloudsLocalVoxelCov(std::vector<PointStruct>& source_global, const std::vector<CovStruct>& target_cov_global,
                                                elimaloc::Matrix4d& last_icp_pose, double trans_th, RegistrationConfig m_config) {
        return Align(ELM_VGICP, source_global, nullptr, &target_cov_global, nullptr, last_icp_pose, trans_th, m_config);
    }

    // reg.hpp:186-217: the covariance term R S the reference attaches to the source points under use_radar_cov (called on the points in
    // the map frame under the initial guess, reg.cpp:302-305); RunRegister evaluates the same function inside its radar kernel
    inline PointStruct CalPointCov(const PointStruct point, double range_var_m, double azim_var_deg, double ele_var_deg) {
        PointStruct cov_point = point;
        const double xyz[3] = {point.pose(0), point.pose(1), point.pose(2)};
        double c9[9];
        elimaloc::check(elm_cal_frame_point_cov(xyz, 1, range_var_m, azim_var_deg, ele_var_deg, c9), VoxelHashMap::ctx(), "CalPointCov");
        for (int k = 0; k < 9; ++k) cov_point.covariance.cov.data()[k] = c9[k]; // both column-major
        return cov_point;
    }
    inline void CalFramePointCov(std::vector<PointStruct>& points, double range_var_m, double azim_var_deg, double ele_var_deg) {
        std::vector<double> xyz(3 * points.size()), c9(9 * points.size());
        for (size_t i = 0; i < points.size(); ++i)
            for (int k = 0; k < 3; ++k) xyz[3 * i + k] = points[i].pose(k);
        elimaloc::check(elm_cal_frame_point_cov(xyz.data(), points.size(), range_var_m, azim_var_deg, ele_var_deg, c9.data()), VoxelHashMap::ctx(),
                        "CalFramePointCov");
        for (size_t i = 0; i < points.size(); ++i)
            for (int k = 0; k < 9; ++k) points[i].covariance.cov.data()[k] = c9[9 * i + k];
    }
    inline double square(double x) { return x * x; } // reg.hpp:219
    inline elimaloc::Matrix3d vectorToSkewSymmetricMatrix(const elimaloc::Vector3d& vec) { // reg.hpp:221-225
        elimaloc::Matrix3d skew_symmetric = elimaloc::Matrix3d::Identity();
        skew_symmetric(0, 0) = 0.0; skew_symmetric(0, 1) = -vec(2); skew_symmetric(0, 2) = vec(1);
        skew_symmetric(1, 0) = vec(2); skew_symmetric(1, 1) = 0.0; skew_symmetric(1, 2) = -vec(0);
        skew_symmetric(2, 0) = -vec(1); skew_symmetric(2, 1) = vec(0); skew_symmetric(2, 2) = 0.0;
        return skew_symmetric;
    }

    // reg.hpp:126-134: in place (the node's debug clouds, pcm.cpp:308-313); every other field is kept
    inline void TransformPoints(const elimaloc::Matrix4d& T, std::vector<PointStruct>& points) {
        for (auto& point : points) Apply(T, point.pose);
    }
    // reg.hpp:136-148
    inline void TransformPoints(const elimaloc::Matrix4d& T, const std::vector<PointStruct>& points, std::vector<PointStruct>& o_points) {
        o_points.resize(points.size());
        for (size_t i = 0; i < points.size(); ++i) {
            o_points[i] = points[i]; // copy all properties
            Apply(T, o_points[i].pose);
        }
    }

    RegistrationConfig config_;
    double d_fitness_score_ = 0.0;

private:
    // pose <- (T * [pose, 1]).head<3>() in the scalar association of Eigen's 4x4 * 4x1 product
    static inline void Apply(const elimaloc::Matrix4d& T, elimaloc::Vector3d& p) {
        const double x = p(0), y = p(1), z = p(2);
        for (int r = 0; r < 3; ++r) p(r) = ((T(r, 0) * x + T(r, 1) * y) + T(r, 2) * z) + T(r, 3) * 1.0;
    }
    elimaloc::Matrix4d Align(int method, const std::vector<PointStruct>& source, const std::vector<PointStruct>* target_points,
                             const std::vector<CovStruct>* target_covs, elimaloc::Matrix6d* local_cov, const elimaloc::Matrix4d& last_icp_pose,
                             double trans_th, const RegistrationConfig& m_config) {
        const size_t n = source.size();
        std::vector<double> src(3 * n), tgt(3 * n), cov(method == ELM_P2P ? 0 : 9 * n), scov(m_config.use_radar_cov && method != ELM_P2P ? 9 * n : 0);
        for (size_t i = 0; i < n; ++i) {
            const CovStruct* tc = target_covs ? &(*target_covs)[i] : &(*target_points)[i].covariance;
            for (int k = 0; k < 3; ++k) {
                src[3 * i + k] = source[i].local(k);
                tgt[3 * i + k] = (method == ELM_P2P) ? (*target_points)[i].pose(k) : tc->mean(k);
            }
            if (method != ELM_P2P)
                for (int k = 0; k < 9; ++k) cov[9 * i + k] = tc->cov.data()[k]; // both column-major
            if (!scov.empty())
                for (int k = 0; k < 9; ++k) scov[9 * i + k] = source[i].covariance.cov.data()[k];
        }
        const elm_reg_config c = m_config.c_struct();
        elimaloc::Matrix4d T = elimaloc::Matrix4d::Identity();
        double lc[36], fitness = 0.0;
        elimaloc::check(elm_align_clouds_local(VoxelHashMap::ctx(), method, src.data(), tgt.data(), cov.empty() ? nullptr : cov.data(),
                                               scov.empty() ? nullptr : scov.data(), n, last_icp_pose.data(), trans_th, &c, T.data(), lc, &fitness),
                        VoxelHashMap::ctx(), "AlignCloudsLocal");
        if (local_cov && method == ELM_GICP)
            for (int r = 0; r < 6; ++r)
                for (int q = 0; q < 6; ++q) (*local_cov)(r, q) = lc[r * 6 + q];
        d_fitness_score_ = fitness;
        return T;
    }
    std::vector<float> scratch_xyz_;
};
