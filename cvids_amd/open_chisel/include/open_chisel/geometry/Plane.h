// geometry/Plane.h of the reference (facade edition): a plane as Plane(p1, p2, p3) leaves it -- normalised normal, offset
// computed from the UN-normalised cross product (src/geometry/Plane.cpp:44-52); the values come from chisel_hip_frustum().
#ifndef CHISEL_HIP_FACADE_PLANE_H_
#define CHISEL_HIP_FACADE_PLANE_H_
#include "Geometry.h"
namespace chisel {
class Plane {
  public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    Plane() : distance(0.0f) {}
    Plane(const Vec3 &n, float d) : normal(n), distance(d) {}
    virtual ~Plane() {}
    float GetSignedDistance(const Vec3 &point) const { return point.dot(normal) + distance; }  // Plane.h:43-46
    Vec3 normal;
    float distance;
};
}  // namespace chisel
#endif
