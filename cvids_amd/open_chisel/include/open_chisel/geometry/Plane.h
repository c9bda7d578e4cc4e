// geometry/Plane.h of the reference (facade edition): a plane as Plane(p1, p2, p3) leaves it -- normalised normal, offset
// computed from the UN-normalised cross product (src/geometry/Plane.cpp:44-52); the values come from chisel_hip_frustum().
#ifndef CHISEL_HIP_FACADE_PLANE_H_
#define CHISEL_HIP_FACADE_PLANE_H_
#include "Geometry.h"
namespace chisel {
class Plane {
  public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    enum class IntersectionType { Inside, Outside, Intersects };  // Plane.h:36-41
    Plane() : distance(0.0f) {}
    Plane(const Vec4 &params) : normal(Vec3(params(0), params(1), params(2))), distance(params(3)) {}  // Plane.cpp:32-36
    Plane(const Vec3 &n, float /*d*/) : normal(n), distance() {}  // Plane.cpp:38-42: the reference value-initialises `distance` (sic): 0, whatever is passed
    Plane(const Vec3 &a, const Vec3 &b, const Vec3 &c) {          // Plane.cpp:44-52: normalised normal, offset from the un-normalised cross product
        const Vec3 cross = (b - a).cross(c - a);
        normal = cross.normalized();
        distance = -(cross.dot(a));
    }
    Plane(float a, float b, float c, float d) : normal(a, b, c), distance(d) {}  // Plane.cpp:54-58
    virtual ~Plane() {}
    float GetSignedDistance(const Vec3 &point) const { return point.dot(normal) + distance; }  // Plane.cpp:60-63
    IntersectionType ClassifyPoint(const Vec3 &point) const {                                   // Plane.h:52-65
        const float d = GetSignedDistance(point);
        if (d < 0) return IntersectionType::Inside;
        if (d > 0) return IntersectionType::Outside;
        return IntersectionType::Intersects;
    }
    Vec3 normal;
    float distance;
};
}  // namespace chisel
#endif
