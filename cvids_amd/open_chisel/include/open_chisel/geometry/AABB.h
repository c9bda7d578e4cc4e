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
    Plane::IntersectionType Intersects(const Plane &plane) const {  // AABB.cpp:37-70
        const Vec3 ext = GetExtents();
        Vec3 corners[8];
        corners[0] = max;
        corners[1] = min;
        corners[2] = min + Vec3(ext(0), 0, 0);
        corners[3] = min + Vec3(0, ext(1), 0);
        corners[4] = min + Vec3(0, 0, ext(2));
        corners[5] = min + Vec3(ext(0), 0, ext(2));
        corners[6] = min + Vec3(ext(0), ext(1), 0);
        corners[7] = min + Vec3(0, ext(1), ext(2));
        float lastdistance = plane.normal.dot(corners[0]) + plane.distance;
        for (int i = 1; i < 8; i++) {
            const float distance = plane.normal.dot(corners[i]) + plane.distance;
            if ((distance <= 0.0f && lastdistance > 0.0f) || (distance >= 0.0f && lastdistance < 0.0f)) return Plane::IntersectionType::Intersects;
            lastdistance = distance;
        }
        if (lastdistance > 0.0f) return Plane::IntersectionType::Outside;
        return Plane::IntersectionType::Inside;
    }
    Vec3 GetCenter() const { return (max + min) * 0.5f; }  // AABB.h:60-63
    Vec3 GetExtents() const { return (max - min) * 0.5f; }
    Vec3 min;
    Vec3 max;
};
}  // namespace chisel
#endif
