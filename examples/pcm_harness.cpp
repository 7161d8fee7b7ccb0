// pcm_harness.cpp -- ROS-free replay of PcmMatching::Init + CallbackPointCloud's call sequence (pcm.cpp:81-101,
// 238-299) through the drop-in shims.  Build: g++ -std=c++17 -Iinclude examples/pcm_harness.cpp
//        -Lelimaloc_amd -lelimaloc_hip -Wl,-rpath,$PWD/elimaloc_amd -o pcm_harness ; needs an MI355X to run.
#include <cmath>
#include <cstdio>
#include <random>

#include "elimaloc/registration.hpp"

int main() {
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> jit(-0.004f, 0.004f);
    std::vector<float> map_xyz;
    for (int i = -200; i < 200; ++i)
        for (int j = -200; j < 200; ++j) {
            map_xyz.push_back((i + 0.5f) * 0.2f + jit(rng));
            map_xyz.push_back((j + 0.5f) * 0.2f + jit(rng));
            map_xyz.push_back(0.3f + jit(rng));
        }
    RegistrationConfig cfg;            // localization.ini defaults
    cfg.icp_method = VGICP;
    Registration registration_;
    registration_.Init(cfg);
    VoxelHashMap local_map_;
    local_map_.Init(1.0, 30);          // pcm.cpp:87
    local_map_.AddPoints(map_xyz.data(), map_xyz.size() / 3);
    local_map_.CalVoxelCovAll();       // pcm.cpp:92-95
    std::vector<PointStruct> scan;
    for (size_t i = 0; i < map_xyz.size() / 3; i += 17) {
        PointStruct p;
        for (int k = 0; k < 3; ++k) p.pose(k) = p.local(k) = map_xyz[3 * i + k] - (k == 2 ? 1.8 : 0.0);
        scan.push_back(p);
    }
    std::vector<PointStruct> ds = local_map_.VoxelDownsample(scan, 0.5); // pcm.cpp:257-258
    elimaloc::Matrix4d T0 = elimaloc::Matrix4d::Identity();               // Eigen::Matrix4d on a machine with Eigen
    T0(0, 3) = 0.08; T0(1, 3) = -0.05; T0(2, 3) = 1.8 + 0.03;
    bool ok = false;
    double fitness = 0.0;
    elimaloc::Matrix6d cov;
    elimaloc::Matrix4d T = registration_.RunRegister(ds, local_map_, T0, cfg, ok, fitness, cov); // pcm.cpp:280-282
    registration_.TransformPoints(T, ds);                                                        // pcm.cpp:308
    std::printf("ok=%d fitness=%.4f t=(%.4f %.4f %.4f) n=%zu\n", ok, fitness, T(0, 3), T(1, 3), T(2, 3), ds.size());
    // one more iteration from the reference's public pieces (reg.cpp:317-372): the pairs of the registered scan, then the step around T
    std::vector<PointStruct> src;
    std::vector<CovStruct> tgt;
    std::tie(src, tgt) = local_map_.GetCorrespondencesCov(ds, cfg.max_search_dist);             // reg.cpp:329
    const elimaloc::Matrix4d step = registration_.AlignCloudsLocalVoxelCov(src, tgt, T, cfg.max_search_dist, cfg); // reg.cpp:368
    std::printf("pairs=%zu step t=(%.4f %.4f %.4f) adjacent=%zu\n", src.size(), step(0, 3), step(1, 3), step(2, 3), local_map_.GetAdjacentVoxels(ds[0], 2).size());
    const bool small = std::fabs(step(0, 3)) + std::fabs(step(1, 3)) + std::fabs(step(2, 3)) < 0.05;
    return ok && std::fabs(T(0, 3)) < 0.02 && std::fabs(T(1, 3)) < 0.02 && src.size() == ds.size() && small ? 0 : 1;
}
