// voxel_hash_map.hpp -- drop-in shim with the reference's class name and method names
// (pcm_matching/include/voxel_hash_map.hpp:41-335) over the C ABI in elimaloc_hip.h.
// pcm_matching.cpp compiles against this header instead of the reference's; Eigen-typed members are provided when
// <Eigen/Core> is available (it is not in the build image, so the plain-array forms are what the tests exercise).
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../elimaloc_hip.h"

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define ELM_HAVE_EIGEN 1
#endif
#endif

namespace elimaloc {
inline elm_ctx* default_context() {
    static elm_ctx* ctx = [] {
        elm_ctx* c = nullptr;
        int rc = elm_ctx_create(0, &c);
        if (rc != ELM_OK) throw std::runtime_error(std::string("elm_ctx_create: ") + elm_strerror(rc));
        return c;
    }();
    return ctx;
}
inline void check(int rc, elm_ctx* ctx, const char* what) {
    if (rc != ELM_OK) throw std::runtime_error(std::string(what) + ": " + elm_strerror(rc) + " " + elm_last_error(ctx));
}
} // namespace elimaloc

// vhm.hpp:41-53
struct CovStruct {
    std::array<double, 9> cov{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; // column-major 3x3
    std::array<double, 3> mean{{0, 0, 0}};
};

// vhm.hpp:55-87 -- only the fields the path reads; pose/local are float32-exact in the reference (pcm.hpp:205-215)
struct PointStruct {
    std::array<double, 3> pose{{0, 0, 0}};
    std::array<double, 3> local{{0, 0, 0}};
    CovStruct covariance;
    float vel = 0, azi_angle = 0, ele_angle = 0;
    double intensity = 0;
};

struct VoxelHashMap {
    VoxelHashMap() = default;
    VoxelHashMap(double voxel_size, int max_points_per_voxel) { Init(voxel_size, max_points_per_voxel); }
    VoxelHashMap(const VoxelHashMap&) = delete;
    VoxelHashMap& operator=(const VoxelHashMap&) = delete;
    ~VoxelHashMap() { Clear(); }

    void Init(double voxel_size, int max_points_per_voxel) { // vhm.cpp:26-29
        voxel_size_ = voxel_size;
        max_points_per_voxel_ = max_points_per_voxel;
    }
    // vhm.cpp:270-285.  Repeated calls append (the device map is rebuilt from the concatenation, which is what
    // sequential AddPoints calls produce in the reference).
    void AddPoints(const std::vector<PointStruct>& points) {
        for (const auto& p : points) {
            xyz_.push_back((float)p.pose[0]);
            xyz_.push_back((float)p.pose[1]);
            xyz_.push_back((float)p.pose[2]);
        }
        Release();
    }
    void AddPoints(const float* xyz, size_t n) {
        xyz_.insert(xyz_.end(), xyz, xyz + 3 * n);
        Release();
    }
    void Update(const std::vector<PointStruct>& points, const std::array<double, 3>&) { AddPoints(points); } // vhm.cpp:268
    void CalVoxelCovAll() { // vhm.hpp:183-193
        want_voxel_cov_ = true;
        elimaloc::check(elm_map_cal_voxel_cov_all(handle()), ctx(), "CalVoxelCovAll");
    }
    void CalPointCovAll(double d_search_dist) { // vhm.hpp:252-257
        want_point_cov_ = d_search_dist;
        elimaloc::check(elm_map_cal_point_cov_all(handle(), d_search_dist), ctx(), "CalPointCovAll");
    }
    bool Empty() const { return elm_map_empty(const_cast<VoxelHashMap*>(this)->handle()) != 0; } // vhm.hpp:325
    void Clear() { // vhm.hpp:324
        Release();
        xyz_.clear();
    }
    std::vector<PointStruct> Pointcloud() const { // vhm.cpp:245-255
        auto* self = const_cast<VoxelHashMap*>(this);
        elm_map_info mi;
        elimaloc::check(elm_map_get_info(self->handle(), &mi), ctx(), "elm_map_get_info");
        std::vector<double> xyz(3 * mi.n_points), cov(9 * mi.n_points), mean(3 * mi.n_points);
        elimaloc::check(elm_map_download_points(self->handle(), xyz.data(), cov.data(), mean.data(), mi.n_points), ctx(), "Pointcloud");
        std::vector<PointStruct> out(mi.n_points);
        for (size_t i = 0; i < out.size(); ++i) {
            for (int k = 0; k < 3; ++k) { out[i].pose[k] = out[i].local[k] = xyz[3 * i + k]; out[i].covariance.mean[k] = mean[3 * i + k]; }
            for (int k = 0; k < 9; ++k) out[i].covariance.cov[k] = cov[9 * i + k];
        }
        return out;
    }
    std::vector<CovStruct> Covariances() const { // vhm.cpp:257-265: voxels holding more than 2 points
        auto* self = const_cast<VoxelHashMap*>(this);
        elm_map_info mi;
        elimaloc::check(elm_map_get_info(self->handle(), &mi), ctx(), "elm_map_get_info");
        std::vector<int32_t> npts(mi.n_voxels);
        std::vector<double> cov(9 * mi.n_voxels), mean(3 * mi.n_voxels);
        elimaloc::check(elm_map_download_voxels(self->handle(), nullptr, npts.data(), cov.data(), mean.data(), mi.n_voxels), ctx(), "Covariances");
        std::vector<CovStruct> out;
        for (size_t v = 0; v < mi.n_voxels; ++v)
            if (npts[v] > 2) {
                CovStruct c;
                for (int k = 0; k < 9; ++k) c.cov[k] = cov[9 * v + k];
                for (int k = 0; k < 3; ++k) c.mean[k] = mean[3 * v + k];
                out.push_back(c);
            }
        return out;
    }
    bool FindGroundHeight(double x, double y, double& ground_z) const { // vhm.hpp:285-322
        int found = 0;
        elimaloc::check(elm_map_find_ground_height(const_cast<VoxelHashMap*>(this)->handle(), x, y, &ground_z, &found), ctx(), "FindGroundHeight");
        return found != 0;
    }
#ifdef ELM_HAVE_EIGEN
    bool FindGroundHeight(const Eigen::Vector2d& position, double& ground_z) const { return FindGroundHeight(position.x(), position.y(), ground_z); }
#endif

    elm_map* handle() {
        if (!map_) {
            elimaloc::check(elm_map_build(ctx(), xyz_.data(), xyz_.size() / 3, voxel_size_, max_points_per_voxel_, &map_), ctx(), "elm_map_build");
            if (want_voxel_cov_) elimaloc::check(elm_map_cal_voxel_cov_all(map_), ctx(), "CalVoxelCovAll");
            if (want_point_cov_ > 0) elimaloc::check(elm_map_cal_point_cov_all(map_, want_point_cov_), ctx(), "CalPointCovAll");
        }
        return map_;
    }
    static elm_ctx* ctx() { return elimaloc::default_context(); }

    double voxel_size_ = 1.0;
    int max_points_per_voxel_ = 30;

private:
    void Release() {
        if (map_) elm_map_destroy(map_);
        map_ = nullptr;
    }
    std::vector<float> xyz_;
    elm_map* map_ = nullptr;
    bool want_voxel_cov_ = false;
    double want_point_cov_ = -1.0;
};
