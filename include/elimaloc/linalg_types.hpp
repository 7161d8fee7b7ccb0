// linalg_types.hpp -- the fixed-size vector / matrix types the drop-in shims expose.
//
// With <Eigen/Core> on the include path (a ROS machine) they ARE the Eigen types the reference's headers use
// (voxel_hash_map.hpp:41-87, registration.hpp:44-85), so pcm_matching.cpp / pcm_matching.hpp compile against the shims
// unchanged.  Without Eigen (this repository's build image) they are minimal column-major stand-ins offering the same
// accessors the shims and the ROS-free examples need: data(), operator()(r, c), operator()(i), x() y() z(),
// Identity(), Zero(), setZero(), setIdentity().
#pragma once
#include <cstddef>

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define ELM_HAVE_EIGEN 1
#endif
#endif

#ifdef ELM_HAVE_EIGEN
namespace Eigen { // registration.hpp:44-57 declares these aliases inside namespace Eigen; pcm_matching.hpp uses Eigen::Matrix6d
using Matrix6d = Eigen::Matrix<double, 6, 6>;
using Matrix3_6d = Eigen::Matrix<double, 3, 6>;
using Vector6d = Eigen::Matrix<double, 6, 1>;
} // namespace Eigen
namespace elimaloc {
using Vector2d = Eigen::Vector2d;
using Vector3d = Eigen::Vector3d;
using Vector3i = Eigen::Vector3i;
using Matrix3d = Eigen::Matrix3d;
using Matrix4d = Eigen::Matrix4d;
using Matrix6d = Eigen::Matrix<double, 6, 6>;
} // namespace elimaloc
#else
namespace elimaloc {
template <typename T, int R, int C>
struct Fixed { // column-major, like Eigen's default
    T v[R * C];
    Fixed() {
        for (int i = 0; i < R * C; ++i) v[i] = T(0);
    }
    Fixed(T a, T b) : Fixed() { static_assert(R * C == 2, "two-coefficient constructor"); v[0] = a; v[1] = b; }
    Fixed(T a, T b, T c) : Fixed() { static_assert(R * C == 3, "three-coefficient constructor"); v[0] = a; v[1] = b; v[2] = c; }
    static Fixed Zero() { return Fixed(); }
    static Fixed Identity() {
        Fixed m;
        for (int i = 0; i < (R < C ? R : C); ++i) m.v[i * R + i] = T(1);
        return m;
    }
    void setZero() { *this = Zero(); }
    void setIdentity() { *this = Identity(); }
    T* data() { return v; }
    const T* data() const { return v; }
    T& operator()(int r, int c) { return v[c * R + r]; }
    const T& operator()(int r, int c) const { return v[c * R + r]; }
    T& operator()(int i) { return v[i]; }
    const T& operator()(int i) const { return v[i]; }
    T& operator[](int i) { return v[i]; }
    const T& operator[](int i) const { return v[i]; }
    T& x() { return v[0]; }
    T& y() { return v[1]; }
    T& z() { return v[2]; }
    const T& x() const { return v[0]; }
    const T& y() const { return v[1]; }
    const T& z() const { return v[2]; }
    static constexpr int size() { return R * C; }
};
using Vector2d = Fixed<double, 2, 1>;
using Vector3d = Fixed<double, 3, 1>;
using Vector3i = Fixed<int, 3, 1>;
using Matrix3d = Fixed<double, 3, 3>;
using Matrix4d = Fixed<double, 4, 4>;
using Matrix6d = Fixed<double, 6, 6>;
} // namespace elimaloc
#endif
