// geometry/Frustum.h:32-70 of the reference, facade edition: the view frustum as the library builds it for the candidate
// enumeration (chisel_hip_frustum: Frustum::SetFromParams / SetFromVectors with the reference's quirks and fp32 order),
// read-only for the caller (chisel_ros draws GetLines(), ChiselServer.cpp:97-134).
#ifndef CHISEL_HIP_FACADE_FRUSTUM_H_
#define CHISEL_HIP_FACADE_FRUSTUM_H_
#include <chisel_hip.h>

#include <cmath>
#include <stdexcept>
#include <string>

#include "AABB.h"
#include "Geometry.h"
#include "Plane.h"
namespace chisel {
class Frustum {
  public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    Frustum() {}
    virtual ~Frustum() {}
    // Frustum.cpp:143-153: fx and cx are accepted and ignored, as in the reference (SetupFrustum passes fy twice anyway)
    void SetFromParams(const Transform &view, float nearDist, float farDist, float /*fx*/, float fy, float /*cx*/, float cy, float imgWidth,
                       float imgHeight) {
        float pose[12], c[24], l[72], p[24];
        for (int r = 0; r < 3; r++) {
            for (int k = 0; k < 3; k++) pose[4 * r + k] = view.linear()(r, k);
            pose[4 * r + 3] = view.translation()(r);
        }
        if (chisel_hip_frustum(pose, fy, cy, (int)imgWidth, (int)imgHeight, nearDist, farDist, c, l, p) != CHISEL_HIP_OK)
            throw std::runtime_error(std::string("chisel_hip: ") + chisel_hip_last_error());
        for (int i = 0; i < 8; i++) corners[i] = Vec3(c[3 * i], c[3 * i + 1], c[3 * i + 2]);
        for (int i = 0; i < 24; i++) lines[i] = Vec3(l[3 * i], l[3 * i + 1], l[3 * i + 2]);
        Plane *dst[6] = {&far, &near, &top, &bottom, &left, &right};
        for (int i = 0; i < 6; i++) *dst[i] = Plane(p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]);
    }
    // Frustum.cpp:155-219
    void SetFromVectors(const Vec3 &forward, const Vec3 &pos, const Vec3 &rightVec, const Vec3 &up, float nearDist, float farDist, float fov, float aspect) {
        const float f[3] = {forward(0), forward(1), forward(2)}, p0[3] = {pos(0), pos(1), pos(2)}, r[3] = {rightVec(0), rightVec(1), rightVec(2)},
                    u[3] = {up(0), up(1), up(2)};
        float c[24], l[72], p[24];
        if (chisel_hip_frustum_from_vectors(f, p0, r, u, nearDist, farDist, fov, aspect, c, l, p) != CHISEL_HIP_OK)
            throw std::runtime_error(std::string("chisel_hip: ") + chisel_hip_last_error());
        for (int i = 0; i < 8; i++) corners[i] = Vec3(c[3 * i], c[3 * i + 1], c[3 * i + 2]);
        for (int i = 0; i < 24; i++) lines[i] = Vec3(l[3 * i], l[3 * i + 1], l[3 * i + 2]);
        Plane *dst[6] = {&far, &near, &top, &bottom, &left, &right};
        for (int i = 0; i < 6; i++) *dst[i] = Plane(p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]);
    }
    // Frustum.cpp:124-141: right / up / -forward are the rows of the view matrix's rotation, the position its fourth column (sic: not the
    // eye point of a world-to-camera matrix), near and far come out of the projection's third row
    void SetFromOpenGLViewProjection(const Mat4x4 &view, const Mat4x4 &proj) {
        const Vec3 right(view(0, 0), view(0, 1), view(0, 2)), up(view(1, 0), view(1, 1), view(1, 2)), d(-view(2, 0), -view(2, 1), -view(2, 2)),
            p(view(0, 3), view(1, 3), view(2, 3));
        const float aa = proj(0, 0), bb = proj(1, 1), cc = proj(2, 2), dd = proj(2, 3);
        const float aspect = bb / aa;
        const float fov = (float)(2.0f * ::atan((double)(1.0f / bb)));
        const float kk = (cc - 1.0f) / (cc + 1.0f);
        const float n = (dd * (1.0f - kk)) / (2.0f * kk);
        const float f = kk * n;
        SetFromVectors(d, p, right, up, n, f, fov, aspect);
    }
    // Frustum.cpp:41-79: true as soon as ONE plane has the box's far vertex (along the plane's normal) on its positive side
    bool Intersects(const AABB &box) const {
        const Plane *planes[] = {&far, &near, &top, &bottom, &left, &right};
        for (const Plane *plane : planes) {
            Vec3 axisVert;
            const Vec3 &normal = plane->normal;
            axisVert(0) = normal(0) < 0.0f ? box.min(0) : box.max(0);
            axisVert(1) = normal(1) < 0.0f ? box.min(1) : box.max(1);
            axisVert(2) = normal(2) < 0.0f ? box.min(2) : box.max(2);
            if (axisVert.dot(normal) + plane->distance > 0.0f) return true;
        }
        return false;
    }
    bool Contains(const Vec3 &point) const {  // Frustum.cpp:81-99
        const Plane *planes[] = {&far, &near, &top, &bottom, &left, &right};
        for (const Plane *plane : planes)
            if (plane->ClassifyPoint(point) == Plane::IntersectionType::Outside) return false;
        return true;
    }
    void ComputeBoundingBox(AABB *box) const {  // Frustum.cpp:101-122
        Vec3 mn(corners[0]), mx(corners[0]);
        for (int i = 1; i < 8; i++)
            for (int k = 0; k < 3; k++) {
                mn(k) = corners[i](k) < mn(k) ? corners[i](k) : mn(k);
                mx(k) = corners[i](k) > mx(k) ? corners[i](k) : mx(k);
            }
        *box = AABB(mn, mx);
    }
    const Plane &GetBottomPlane() const { return bottom; }
    const Plane &GetTopPlane() const { return top; }
    const Plane &GetLeftPlane() const { return left; }
    const Plane &GetRightPlane() const { return right; }
    const Plane &GetNearPlane() const { return near; }
    const Plane &GetFarPlane() const { return far; }
    const Vec3 *GetLines() const { return lines; }
    const Vec3 *GetCorners() const { return corners; }
  protected:
    Vec3 corners[8];
    Vec3 lines[24];
    Plane top, left, right, bottom, near, far;
};
}  // namespace chisel
#endif
