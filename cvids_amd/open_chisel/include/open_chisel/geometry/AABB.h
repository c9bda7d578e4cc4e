// geometry/AABB.h:33-72 of the reference (facade edition: the members chisel_ros and Chunk::ComputeBoundingBox use)
#ifndef CHISEL_HIP_FACADE_AABB_H_
#define CHISEL_HIP_FACADE_AABB_H_
#include "Geometry.h"
#include "Plane.h"
namespace chisel {
class AABB {
  public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    AABB() {}
    AABB(const Vec3 &min_, const Vec3 &max_) : min(min_), max(max_) {}
    virtual ~AABB() {}
    bool Contains(const Vec3 &pos) const {  // AABB.h:43-47
        return pos(0) >= min(0) && pos(1) >= min(1) && pos(2) >= min(2) && pos(0) <= max(0) && pos(1) <= max(1) && pos(2) <= max(2);
    }
    bool Intersects(const AABB &other) const {  // AABB.h:49-58
        if (min.x() > other.max.x()) return false;
        if (min.y() > other.max.y()) return false;
        if (min.z() > other.max.z()) return false;
        if (max.x() < other.min.x()) return false;
        if (max.y() < other.min.y()) return false;
        if (max.z() < other.min.z()) return false;
        return true;
    }
    // AABB.cpp:37-70: the signed distances of eight points, taken in the reference's order -- max, min, then min moved by the (half!)
    // extents along x, y, z, xz, xy, yz -- and "Intersects" as soon as two consecutive ones lie on different sides
    Plane::IntersectionType Intersects(const Plane &plane) const {
        static const unsigned char moved[8] = {0, 0, 1, 2, 4, 5, 3, 6};  // bit a: the point is min + extents along axis a (entry 0 stands for max)
        const Vec3 half = GetExtents();
        float previous = 0.0f;
        for (int i = 0; i < 8; i++) {
            Vec3 p = i == 0 ? max : min;
            for (int a = 0; a < 3; a++)
                if (i > 0 && ((moved[i] >> a) & 1)) p(a) = min(a) + half(a);
            const float d = plane.normal.dot(p) + plane.distance;
            if (i > 0 && ((d <= 0.0f && previous > 0.0f) || (d >= 0.0f && previous < 0.0f))) return Plane::IntersectionType::Intersects;
            previous = d;
        }
        return previous > 0.0f ? Plane::IntersectionType::Outside : Plane::IntersectionType::Inside;
    }
    Vec3 GetCenter() const { return (max + min) * 0.5f; }  // AABB.h:60-63
    Vec3 GetExtents() const { return (max - min) * 0.5f; }
    Vec3 min;
    Vec3 max;
};
}  // namespace chisel
#endif
