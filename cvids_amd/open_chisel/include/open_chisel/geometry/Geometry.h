// open_chisel/geometry/Geometry.h -- the typedefs of the reference (geometry/Geometry.h:33-51) for the chisel_hip facade.
// With Eigen on the include path (the caller's build has it: chisel_ros/catkin.cmake:6-8) these are the reference's own Eigen
// types and nothing below the #else is compiled.  Without it (this repository's build container has no Eigen) a small stand-in
// with the members the facade and chisel_ros' call sites touch keeps the headers and the caller compile check
// (cvids_amd/open_chisel/tests/caller_check.cpp) buildable; it is NOT an Eigen replacement for the reference's sources and it
// takes no part in any parity claim (poses reach the library as 12 floats either way).
#ifndef CHISEL_HIP_FACADE_GEOMETRY_H_
#define CHISEL_HIP_FACADE_GEOMETRY_H_
#include <cmath>
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
#ifndef EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#endif
#else
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
namespace Eigen {
template <class T, int N>
struct VecT {
    T v[N];
    VecT() {
        for (int i = 0; i < N; i++) v[i] = T(0);
    }
    VecT(T a, T b) : v{a, b} { static_assert(N == 2, "two components"); }
    VecT(T a, T b, T c) : v{a, b, c} { static_assert(N == 3, "three components"); }
    VecT(T a, T b, T c, T d) : v{a, b, c, d} { static_assert(N == 4, "four components"); }
    T &operator()(int i) { return v[i]; }
    const T &operator()(int i) const { return v[i]; }
    T &operator[](int i) { return v[i]; }
    const T &operator[](int i) const { return v[i]; }
    T &x() { return v[0]; }
    T &y() { return v[1]; }
    T &z() { return v[2]; }
    T &w() { return v[3]; }
    const T &x() const { return v[0]; }
    const T &y() const { return v[1]; }
    const T &z() const { return v[2]; }
    const T &w() const { return v[3]; }
    bool operator==(const VecT &o) const {
        for (int i = 0; i < N; i++)
            if (!(v[i] == o.v[i])) return false;
        return true;
    }
    bool operator!=(const VecT &o) const { return !(*this == o); }
    VecT operator+(const VecT &o) const {
        VecT r;
        for (int i = 0; i < N; i++) r.v[i] = v[i] + o.v[i];
        return r;
    }
    VecT operator-(const VecT &o) const {
        VecT r;
        for (int i = 0; i < N; i++) r.v[i] = v[i] - o.v[i];
        return r;
    }
    VecT operator*(T s) const {
        VecT r;
        for (int i = 0; i < N; i++) r.v[i] = v[i] * s;
        return r;
    }
    friend VecT operator*(T s, const VecT &a) { return a * s; }
    T dot(const VecT &o) const {
        T s = T(0);
        for (int i = 0; i < N; i++) s += v[i] * o.v[i];
        return s;
    }
    VecT cross(const VecT &o) const {
        static_assert(N == 3, "cross product of three-vectors");
        return VecT(v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]);
    }
    T norm() const { return (T)std::sqrt((double)dot(*this)); }
    void normalize() {
        const T n = norm();
        if (n > T(0))
            for (int i = 0; i < N; i++) v[i] /= n;
    }
    VecT normalized() const {
        VecT r = *this;
        r.normalize();
        return r;
    }
    template <class U>
    VecT<U, N> cast() const {
        VecT<U, N> r;
        for (int i = 0; i < N; i++) r.v[i] = (U)v[i];
        return r;
    }
    static VecT Zero() { return VecT(); }
};
typedef VecT<int, 2> Vector2i;
typedef VecT<int, 3> Vector3i;
typedef VecT<float, 2> Vector2f;
typedef VecT<float, 3> Vector3f;
typedef VecT<float, 4> Vector4f;
template <int N>
struct MatT {
    float m[N * N];  // row-major
    MatT() {
        for (int i = 0; i < N * N; i++) m[i] = (i % (N + 1) == 0) ? 1.0f : 0.0f;
    }
    float &operator()(int r, int c) { return m[N * r + c]; }
    const float &operator()(int r, int c) const { return m[N * r + c]; }
    static MatT Identity() { return MatT(); }
    static MatT Zero() {
        MatT z;
        for (int i = 0; i < N * N; i++) z.m[i] = 0.0f;
        return z;
    }
};
struct Matrix3f : MatT<3> {
    Matrix3f inverse() const {  // cofactors over the determinant
        const Matrix3f &a = *this;
        Matrix3f r;
        const float det = a(0, 0) * (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) - a(0, 1) * (a(1, 0) * a(2, 2) - a(1, 2) * a(2, 0)) +
                          a(0, 2) * (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0));
        const float id = 1.0f / det;
        r(0, 0) = (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) * id;
        r(0, 1) = (a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2)) * id;
        r(0, 2) = (a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1)) * id;
        r(1, 0) = (a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2)) * id;
        r(1, 1) = (a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0)) * id;
        r(1, 2) = (a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2)) * id;
        r(2, 0) = (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0)) * id;
        r(2, 1) = (a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1)) * id;
        r(2, 2) = (a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0)) * id;
        return r;
    }
    Vector3f operator*(const Vector3f &p) const {
        const Matrix3f &a = *this;
        return Vector3f(a(0, 0) * p(0) + a(0, 1) * p(1) + a(0, 2) * p(2), a(1, 0) * p(0) + a(1, 1) * p(1) + a(1, 2) * p(2),
                        a(2, 0) * p(0) + a(2, 1) * p(1) + a(2, 2) * p(2));
    }
};
typedef MatT<4> Matrix4f;
struct Quaternionf {
    float q[4];  // x, y, z, w
    Quaternionf() : q{0, 0, 0, 1} {}
    Quaternionf(float w_, float x_, float y_, float z_) : q{x_, y_, z_, w_} {}
    explicit Quaternionf(const Matrix3f &m) {  // Shepperd's method
        const float tr = m(0, 0) + m(1, 1) + m(2, 2);
        if (tr > 0.0f) {
            const float s = std::sqrt(tr + 1.0f) * 2.0f;
            q[3] = 0.25f * s;
            q[0] = (m(2, 1) - m(1, 2)) / s;
            q[1] = (m(0, 2) - m(2, 0)) / s;
            q[2] = (m(1, 0) - m(0, 1)) / s;
        } else {
            int i = 0;
            if (m(1, 1) > m(0, 0)) i = 1;
            if (m(2, 2) > m(i, i)) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            const float s = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0f) * 2.0f;
            q[i] = 0.25f * s;
            q[3] = (m(k, j) - m(j, k)) / s;
            q[j] = (m(j, i) + m(i, j)) / s;
            q[k] = (m(k, i) + m(i, k)) / s;
        }
    }
    float &x() { return q[0]; }
    float &y() { return q[1]; }
    float &z() { return q[2]; }
    float &w() { return q[3]; }
    const float &x() const { return q[0]; }
    const float &y() const { return q[1]; }
    const float &z() const { return q[2]; }
    const float &w() const { return q[3]; }
    Matrix3f toRotationMatrix() const {
        const float x = q[0], y = q[1], z = q[2], w = q[3];
        Matrix3f m;
        m(0, 0) = 1 - 2 * (y * y + z * z); m(0, 1) = 2 * (x * y - z * w); m(0, 2) = 2 * (x * z + y * w);
        m(1, 0) = 2 * (x * y + z * w); m(1, 1) = 1 - 2 * (x * x + z * z); m(1, 2) = 2 * (y * z - x * w);
        m(2, 0) = 2 * (x * z - y * w); m(2, 1) = 2 * (y * z + x * w); m(2, 2) = 1 - 2 * (x * x + y * y);
        return m;
    }
};
struct Affine3f {  // camera -> world rigid transform
    Matrix3f R;
    Vector3f t;
    Matrix3f &linear() { return R; }
    const Matrix3f &linear() const { return R; }
    Matrix3f rotation() const { return R; }
    Vector3f &translation() { return t; }
    const Vector3f &translation() const { return t; }
    Affine3f inverse() const {
        Affine3f r;
        r.R = R.inverse();
        const Vector3f p = r.R * t;
        r.t = Vector3f(-p(0), -p(1), -p(2));
        return r;
    }
    Vector3f operator*(const Vector3f &p) const { return R * p + t; }
    static Affine3f Identity() { return Affine3f(); }
};
// Matrix<float, R, C> as marching_cubes/MarchingCubes.h uses it (3 x 8 corner coordinates, 8 x 1 distances, 3 x 12 edge vertices):
// column-major storage like Eigen's default, (r, c) / (i) access, col(c) as a three-vector
template <class T, int R, int C>
struct Matrix {
    T m[R * C];
    Matrix() {
        for (int i = 0; i < R * C; i++) m[i] = T(0);
    }
    T &operator()(int r, int c) { return m[c * R + r]; }
    const T &operator()(int r, int c) const { return m[c * R + r]; }
    T &operator()(int i) { return m[i]; }
    const T &operator()(int i) const { return m[i]; }
    struct Col {
        T *p;
        operator VecT<T, 3>() const { return VecT<T, 3>(p[0], p[1], p[2]); }
        Col &operator=(const VecT<T, 3> &v) {
            p[0] = v(0); p[1] = v(1); p[2] = v(2);
            return *this;
        }
    };
    struct ConstCol {
        const T *p;
        operator VecT<T, 3>() const { return VecT<T, 3>(p[0], p[1], p[2]); }
    };
    Col col(int c) {
        static_assert(R == 3, "columns of three-row matrices");
        return Col{m + c * R};
    }
    ConstCol col(int c) const {
        static_assert(R == 3, "columns of three-row matrices");
        return ConstCol{m + c * R};
    }
    const T *data() const { return m; }
    T *data() { return m; }
};
template <class T>
using aligned_allocator = std::allocator<T>;
}  // namespace Eigen
#endif

namespace chisel {
typedef Eigen::Vector2i Point2;
typedef Eigen::Vector3i Point3;
typedef Eigen::Vector2f Vec2;
typedef Eigen::Vector3f Vec3;
typedef Eigen::Vector4f Vec4;
typedef Eigen::Matrix3f Mat3x3;
typedef Eigen::Matrix4f Mat4x4;
typedef Eigen::Affine3f Transform;
typedef Eigen::Quaternionf Quaternion;

typedef std::vector<Point2, Eigen::aligned_allocator<Point2>> Point2List;
typedef std::vector<Point3, Eigen::aligned_allocator<Point3>> Point3List;
typedef std::vector<Vec2, Eigen::aligned_allocator<Vec2>> Vec2List;
typedef std::vector<Vec3, Eigen::aligned_allocator<Vec3>> Vec3List;
typedef std::vector<Vec4, Eigen::aligned_allocator<Vec4>> Vec4List;
typedef std::vector<Mat3x3, Eigen::aligned_allocator<Mat3x3>> Mat3x3List;
typedef std::vector<Mat4x4, Eigen::aligned_allocator<Mat4x4>> Mat4List;
typedef std::vector<Transform, Eigen::aligned_allocator<Transform>> TransformList;
typedef std::vector<Quaternion, Eigen::aligned_allocator<Quaternion>> QuaternionList;
}  // namespace chisel
#endif
