// voxel_hash_map.hpp -- drop-in shim with the reference's class name, member names and signatures
// (pcm_matching/include/voxel_hash_map.hpp:41-335) over the C ABI in elimaloc_hip.h.
// pcm_matching.cpp / pcm_matching.hpp compile against this header instead of the reference's: with <Eigen/Core> present
// every Eigen-typed member of the reference (PointStruct::pose/local, CovStruct::cov/mean, Voxel) has its Eigen type
// (linalg_types.hpp); without Eigen (this repository's build image) the same members are minimal fixed-size stand-ins.
// tests/test_shim_compile.py compiles the reference's literal call lines against this header with a test-only Eigen stub.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../elimaloc_hip.h"
#include "linalg_types.hpp"

namespace elimaloc {
// The process-wide context of the shims.  ELM_DEVICES="0,1,2,3" (e.g. exported by the launch file): a device GROUP -- the map is replicated
// on those GPUs, every RunRegister shards its scan over them and all-reduces the 6x6 normal equations per ICP iteration (RCCL over xGMI);
// pcm_matching.cpp's call sites (pcm.cpp:82-105, 280-282, 412-414) do not change.  Unset: device 0.
inline elm_ctx* default_context() {
    static elm_ctx* ctx = [] {
        std::vector<int> ids;
        if (const char* e = std::getenv("ELM_DEVICES")) {
            for (const char* p = e; *p;) {
                char* end = nullptr;
                const long v = std::strtol(p, &end, 10);
                if (end == p) break;
                ids.push_back((int)v);
                p = end;
                while (*p == ',' || *p == ' ') ++p;
            }
        }
        if (ids.empty()) ids.push_back(0);
        elm_ctx* c = nullptr;
        int rc = elm_ctx_create_multi(ids.data(), (int)ids.size(), &c);
        if (rc != ELM_OK) throw std::runtime_error(std::string("elm_ctx_create_multi: ") + elm_strerror(rc));
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
    elimaloc::Matrix3d cov;  // 3x3 covariance matrix
    elimaloc::Vector3d mean; // 3D mean vector
    CovStruct() : cov(elimaloc::Matrix3d::Identity()), mean(elimaloc::Vector3d::Zero()) {}
    CovStruct(const elimaloc::Matrix3d& c, const elimaloc::Vector3d& m) : cov(c), mean(m) {}
    void reset() {
        cov = elimaloc::Matrix3d::Identity();
        mean = elimaloc::Vector3d::Zero();
    }
};

// vhm.hpp:55-87; pose/local are float32-exact in the reference (filled from float32 PCL points, pcm.hpp:205-215)
struct PointStruct {
    elimaloc::Vector3d pose;
    elimaloc::Vector3d local;
    CovStruct covariance;
    float vel;       // mps
    float azi_angle; // deg
    float ele_angle; // deg
    double intensity;
    PointStruct()
        : pose(elimaloc::Vector3d::Zero()), local(elimaloc::Vector3d::Zero()), covariance(CovStruct()), vel(0.0), azi_angle(0.0),
          ele_angle(0.0), intensity(0.0) {}
    void reset() {
        pose.setZero();
        local.setZero();
        covariance.reset();
        vel = 0.0;
        azi_angle = 0.0;
        ele_angle = 0.0;
        intensity = 0.0;
    }
};

// the reference's layout (vhm.hpp:55-87: 24 + 24 + 96 + 3 floats + pad + 8; Vector3d / Matrix3d of doubles are not "fixed-size
// vectorizable" in Eigen, so neither struct carries an alignment requirement beyond 8 bytes or needs EIGEN_MAKE_ALIGNED_OPERATOR_NEW, and
// std::vector<PointStruct> uses the default allocator there as here)
static_assert(sizeof(CovStruct) == 96 && sizeof(PointStruct) == 168 && alignof(PointStruct) == 8, "PointStruct must keep the reference's layout");

struct VoxelHashMap {
    using RadarPointVector = std::vector<PointStruct>;
    using RadarPointVectorTuple = std::tuple<RadarPointVector, RadarPointVector>;
    using Voxel = elimaloc::Vector3i;

    VoxelHashMap() {}
    VoxelHashMap(double voxel_size, int max_points_per_voxel) { Init(voxel_size, max_points_per_voxel); }
    VoxelHashMap(const VoxelHashMap&) = delete; // owns device memory (the reference's node never copies its map either)
    VoxelHashMap& operator=(const VoxelHashMap&) = delete;
    ~VoxelHashMap() { Release(); }

    void Init(double voxel_size, int max_points_per_voxel) { // vhm.cpp:26-29
        voxel_size_ = voxel_size;
        max_points_per_voxel_ = max_points_per_voxel;
    }
    // vhm.cpp:270-285.  Repeated calls append (the device map is rebuilt from the concatenation, which is what
    // sequential AddPoints calls produce in the reference).
    void AddPoints(const RadarPointVector& points) {
        xyz_.reserve(xyz_.size() + 3 * points.size());
        for (const auto& p : points) {
            xyz_.push_back((float)p.pose(0));
            xyz_.push_back((float)p.pose(1));
            xyz_.push_back((float)p.pose(2));
        }
        Release();
    }
    void AddPoints(const float* xyz, size_t n) { // float32 fast path for callers that hold the PCD's own arrays
        xyz_.insert(xyz_.end(), xyz, xyz + 3 * n);
        Release();
    }
    void Update(const RadarPointVector& points, const elimaloc::Vector3d&) { AddPoints(points); } // vhm.cpp:268
    void CalVoxelCovAll() { // vhm.hpp:183-193
        want_voxel_cov_ = true;
        elimaloc::check(elm_map_cal_voxel_cov_all(handle()), ctx(), "CalVoxelCovAll");
    }
    void CalPointCovAll(double d_search_dist) { // vhm.hpp:252-257
        want_point_cov_ = d_search_dist;
        elimaloc::check(elm_map_cal_point_cov_all(handle(), d_search_dist), ctx(), "CalPointCovAll");
    }
    inline Voxel PointToVoxel(const elimaloc::Vector3d& point, const double voxel_size) const { // vhm.hpp:176-180
        return Voxel(static_cast<int>(std::floor(point.x() / voxel_size)), static_cast<int>(std::floor(point.y() / voxel_size)),
                     static_cast<int>(std::floor(point.z() / voxel_size)));
    }
    // vhm.hpp:260-283: the first point of every floor-keyed voxel.  The reference emits unordered_map iteration order (only
    // the set is contractual); here the kept points come back in input order.
    inline std::vector<PointStruct> VoxelDownsample(const std::vector<PointStruct>& points, const double voxel_size) const {
        std::vector<float> xyz(3 * points.size());
        for (size_t i = 0; i < points.size(); ++i)
            for (int k = 0; k < 3; ++k) xyz[3 * i + k] = (float)points[i].pose(k);
        std::vector<int64_t> keep(points.size());
        size_t n_keep = 0;
        elimaloc::check(elm_voxel_downsample(xyz.data(), points.size(), voxel_size, keep.data(), &n_keep), ctx(), "VoxelDownsample");
        std::vector<PointStruct> points_downsampled;
        points_downsampled.reserve(n_keep);
        for (size_t k = 0; k < n_keep; ++k) points_downsampled.emplace_back(points[(size_t)keep[k]]);
        return points_downsampled;
    }
    inline bool Empty() const { return elm_map_empty(handle()) != 0; } // vhm.hpp:325
    inline void Clear() { // vhm.hpp:324
        Release();
        xyz_.clear();
    }
    std::vector<PointStruct> Pointcloud() const { // vhm.cpp:245-255
        elm_map_info mi;
        elimaloc::check(elm_map_get_info(handle(), &mi), ctx(), "elm_map_get_info");
        std::vector<double> xyz(3 * mi.n_points), cov(9 * mi.n_points), mean(3 * mi.n_points);
        elimaloc::check(elm_map_download_points(handle(), xyz.data(), cov.data(), mean.data(), mi.n_points), ctx(), "Pointcloud");
        std::vector<PointStruct> out(mi.n_points);
        for (size_t i = 0; i < out.size(); ++i) {
            for (int k = 0; k < 3; ++k) {
                out[i].pose(k) = out[i].local(k) = xyz[3 * i + k];
                out[i].covariance.mean(k) = mean[3 * i + k];
            }
            for (int k = 0; k < 9; ++k) out[i].covariance.cov.data()[k] = cov[9 * i + k]; // both column-major
        }
        return out;
    }
    std::vector<CovStruct> Covariances() const { // vhm.cpp:257-265: voxels holding more than 2 points
        elm_map_info mi;
        elimaloc::check(elm_map_get_info(handle(), &mi), ctx(), "elm_map_get_info");
        std::vector<int32_t> npts(mi.n_voxels);
        std::vector<double> cov(9 * mi.n_voxels), mean(3 * mi.n_voxels);
        elimaloc::check(elm_map_download_voxels(handle(), nullptr, npts.data(), cov.data(), mean.data(), mi.n_voxels), ctx(), "Covariances");
        std::vector<CovStruct> out;
        for (size_t v = 0; v < mi.n_voxels; ++v)
            if (npts[v] > 2) {
                CovStruct c;
                for (int k = 0; k < 9; ++k) c.cov.data()[k] = cov[9 * v + k];
                for (int k = 0; k < 3; ++k) c.mean(k) = mean[3 * v + k];
                out.push_back(c);
            }
        return out;
    }
    // vhm.cpp:31-88 / 90-151 / 153-206: the correspondence calls of the reference's public interface.  RunRegister never needs them
    // here (its iterations search and accumulate in one kernel); a caller that wants the pairs themselves gets them from the same
    // search (elm_map_get_correspondences), in input order like the reference's vectors.  Source points are copied whole (every
    // PointStruct member), targets carry pose / local / covariance of the matched map point (or voxel); a point without any
    // neighbour bucket pairs with the default-constructed target when the origin is within range (vhm.cpp:37 / :105).
    std::tuple<std::vector<PointStruct>, std::vector<PointStruct>> GetCorrespondencePoints(const RadarPointVector& vec_points,
                                                                                           double d_max_correspondence_dist) const {
        std::vector<uint32_t> src;
        std::vector<int32_t> tgt;
        Pairs(0, vec_points, d_max_correspondence_dist, src, tgt);
        const std::vector<PointStruct> cloud = tgt.empty() ? std::vector<PointStruct>() : Pointcloud();
        std::vector<PointStruct> vec_source, vec_target;
        vec_source.reserve(src.size());
        vec_target.reserve(src.size());
        for (size_t k = 0; k < src.size(); ++k) {
            vec_source.emplace_back(vec_points[src[k]]);
            vec_target.emplace_back(tgt[k] >= 0 ? cloud[(size_t)tgt[k]] : PointStruct());
        }
        return std::make_tuple(std::move(vec_source), std::move(vec_target));
    }
    std::tuple<std::vector<PointStruct>, std::vector<CovStruct>> GetCorrespondencesCov(const RadarPointVector& vec_points,
                                                                                      double d_max_correspondence_dist) const {
        return CovPairs(1, vec_points, d_max_correspondence_dist);
    }
    std::tuple<std::vector<PointStruct>, std::vector<CovStruct>> GetCorrespondencesAllCov(const RadarPointVector& vec_points,
                                                                                         double d_max_correspondence_dist) const {
        return CovPairs(2, vec_points, d_max_correspondence_dist);
    }
    // vhm.cpp:208-243: key arithmetic only (whether or not such voxels exist): range 0 the voxel itself, 1 the seven
    // (0, +x, -x, +y, -y, +z, -z), anything else the 27 of the 3 x 3 x 3 block, x slowest
    std::vector<Voxel> GetAdjacentVoxels(const PointStruct& point, int range) const {
        const Voxel voxel = PointToVoxel(point.pose, voxel_size_);
        const int vx = voxel(0), vy = voxel(1), vz = voxel(2);
        if (range == 0) return std::vector<Voxel>{voxel};
        if (range == 1)
            return std::vector<Voxel>{Voxel(vx, vy, vz), Voxel(vx + 1, vy, vz), Voxel(vx - 1, vy, vz), Voxel(vx, vy + 1, vz),
                                      Voxel(vx, vy - 1, vz), Voxel(vx, vy, vz + 1), Voxel(vx, vy, vz - 1)};
        std::vector<Voxel> voxels;
        voxels.reserve(27);
        for (int i = vx - 1; i < vx + 2; ++i)
            for (int j = vy - 1; j < vy + 2; ++j)
                for (int k = vz - 1; k < vz + 2; ++k) voxels.emplace_back(i, j, k);
        return voxels;
    }
    inline bool FindGroundHeight(const elimaloc::Vector2d& position, double& ground_z) const { // vhm.hpp:285-322
        int found = 0;
        elimaloc::check(elm_map_find_ground_height(handle(), position(0), position(1), &ground_z, &found), ctx(), "FindGroundHeight");
        return found != 0;
    }

    void Pairs(int what, const RadarPointVector& vec_points, double max_dist, std::vector<uint32_t>& src, std::vector<int32_t>& tgt) const {
        std::vector<double> xyz(3 * vec_points.size());
        for (size_t i = 0; i < vec_points.size(); ++i)
            for (int k = 0; k < 3; ++k) xyz[3 * i + k] = vec_points[i].pose(k);
        const size_t cap = vec_points.size() * (what == 2 ? 7 : 1);
        src.resize(cap);
        tgt.resize(cap);
        size_t n_pairs = 0;
        elimaloc::check(elm_map_get_correspondences(ctx(), handle(), what, xyz.data(), vec_points.size(), max_dist, src.data(), tgt.data(), cap, &n_pairs),
                        ctx(), "elm_map_get_correspondences");
        src.resize(n_pairs);
        tgt.resize(n_pairs);
    }
    std::tuple<std::vector<PointStruct>, std::vector<CovStruct>> CovPairs(int what, const RadarPointVector& vec_points, double max_dist) const {
        std::vector<uint32_t> src;
        std::vector<int32_t> tgt;
        Pairs(what, vec_points, max_dist, src, tgt);
        elm_map_info mi;
        elimaloc::check(elm_map_get_info(handle(), &mi), ctx(), "elm_map_get_info");
        std::vector<double> cov(9 * mi.n_voxels), mean(3 * mi.n_voxels);
        if (!tgt.empty()) elimaloc::check(elm_map_download_voxels(handle(), nullptr, nullptr, cov.data(), mean.data(), mi.n_voxels), ctx(), "elm_map_download_voxels");
        std::vector<PointStruct> vec_source;
        std::vector<CovStruct> vec_target;
        vec_source.reserve(src.size());
        vec_target.reserve(src.size());
        for (size_t k = 0; k < src.size(); ++k) {
            vec_source.emplace_back(vec_points[src[k]]);
            CovStruct c; // (the default: identity covariance at the origin)
            if (tgt[k] >= 0) {
                for (int q = 0; q < 9; ++q) c.cov.data()[q] = cov[9 * (size_t)tgt[k] + q]; // both column-major
                for (int q = 0; q < 3; ++q) c.mean(q) = mean[3 * (size_t)tgt[k] + q];
            }
            vec_target.push_back(c);
        }
        return std::make_tuple(std::move(vec_source), std::move(vec_target));
    }
    // the device-resident map (built lazily from the accumulated points on first use; const like the reference's read paths)
    elm_map* handle() const {
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
    mutable elm_map* map_ = nullptr;
    bool want_voxel_cov_ = false;
    double want_point_cov_ = -1.0;
};
