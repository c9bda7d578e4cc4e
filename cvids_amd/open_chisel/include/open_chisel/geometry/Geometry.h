// open_chisel/geometry/Geometry.h -- typedefs of the reference (geometry/Geometry.h:33-51) for the chisel_hip facade.
// With Eigen on the include path (the caller's build has it: chisel_ros/catkin.cmake:6-8) these are the reference's
// own Eigen types.  Without it (this repository's GPU-less build container) a minimal stand-in with the members the
// facade itself touches keeps the headers compilable; it is NOT an Eigen replacement for the reference's sources.
#ifndef CHISEL_HIP_FACADE_GEOMETRY_H_
#define CHISEL_HIP_FACADE_GEOMETRY_H_
#include <memory>
#include <vector>

#if defined(__has_include)
#if __has_include(<Eigen/Core>) && __has_include(<Eigen/Geometry>)
#define CHISEL_HIP_HAVE_EIGEN 1
#endif
#endif

#ifdef CHISEL_HIP_HAVE_EIGEN
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <Eigen/StdVector>
#else
namespace Eigen {
template <class T>
struct Vec3T {
    T v[3];
    Vec3T() : v{T(0), T(0), T(0)} {}
    Vec3T(T a, T b, T c) : v{a, b, c} {}
    T &operator()(int i) { return v[i]; }
    const T &operator()(int i) const { return v[i]; }
    T &x() { return v[0]; }
    T &y() { return v[1]; }
    T &z() { return v[2]; }
    const T &x() const { return v[0]; }
    const T &y() const { return v[1]; }
    const T &z() const { return v[2]; }
    bool operator==(const Vec3T &o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2]; }
    Vec3T operator+(const Vec3T &o) const { return Vec3T(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
};
typedef Vec3T<int> Vector3i;
typedef Vec3T<float> Vector3f;
struct Matrix3f {
    float m[9];  // row-major
    float &operator()(int r, int c) { return m[3 * r + c]; }
    const float &operator()(int r, int c) const { return m[3 * r + c]; }
};
struct Affine3f {  // camera -> world rigid transform
    Matrix3f R;
    Vector3f t;
    Affine3f() : R{{1, 0, 0, 0, 1, 0, 0, 0, 1}} {}
    Matrix3f &linear() { return R; }
    const Matrix3f &linear() const { return R; }
    Vector3f &translation() { return t; }
    const Vector3f &translation() const { return t; }
};
}  // namespace Eigen
#endif

namespace chisel {
typedef Eigen::Vector3i Point3;
typedef Eigen::Vector3f Vec3;
typedef Eigen::Matrix3f Mat3x3;
typedef Eigen::Affine3f Transform;
#ifdef CHISEL_HIP_HAVE_EIGEN
typedef std::vector<Vec3, Eigen::aligned_allocator<Vec3>> Vec3List;
typedef std::vector<Point3, Eigen::aligned_allocator<Point3>> Point3List;
#else
typedef std::vector<Vec3> Vec3List;
typedef std::vector<Point3> Point3List;
#endif
}  // namespace chisel
#endif
