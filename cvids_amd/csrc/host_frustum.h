// host_frustum.h -- per-frame host arithmetic of the product: view frustum -> candidate chunk-id range.
//
// The GPU enumerates candidates itself (kernels_cull.h) but must never touch a chunk the reference
// would not have enumerated (e.g. beyond the far plane), so the enumeration RANGE and the six plane
// equations are computed here exactly as the reference does, in fp32 with its operation order:
//   PinholeCamera::SetupFrustum   camera/PinholeCamera.cpp:55-59  (passes fy for both focal lengths)
//   Frustum::SetFromParams        geometry/Frustum.cpp:143-153    (cx unused; atan2 in double)
//   Frustum::SetFromVectors       geometry/Frustum.cpp:155-219
//   Plane::Plane(a, b, c)         geometry/Plane.cpp:44-52        (normalised normal, un-normalised offset)
//   Frustum::ComputeBoundingBox   geometry/Frustum.cpp:101-122
//   ChunkManager::GetIDAt         ChunkManager.h:136-145
//   range of GetChunkIDsIntersecting  ChunkManager.cpp:189-199    ([minID-1, maxID+1], maxID = GetIDAt(max)+1)
// 3-term reductions use Eigen's a0 + (a1 + a2) order (see DESIGN.md "fp32 operation order").
#pragma once
#include <cmath>

namespace chisel_hip {
namespace hostmath {

struct f3 {
    float v[3];
};
inline f3 mk(float a, float b, float c) { return f3{{a, b, c}}; }
inline f3 add(const f3 &a, const f3 &b) { return mk(a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]); }
inline f3 sub(const f3 &a, const f3 &b) { return mk(a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]); }
inline f3 scale(const f3 &a, float s) { return mk(a.v[0] * s, a.v[1] * s, a.v[2] * s); }
inline float red3(float a, float b, float c) { return a + (b + c); }
inline float dot3(const f3 &a, const f3 &b) { return red3(a.v[0] * b.v[0], a.v[1] * b.v[1], a.v[2] * b.v[2]); }
inline f3 cross3(const f3 &a, const f3 &b) {
    return mk(a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2], a.v[0] * b.v[1] - a.v[1] * b.v[0]);
}

struct PlaneEq {
    f3 n;
    float d;
};
inline PlaneEq plane_from_points(const f3 &a, const f3 &b, const f3 &c) {
    f3 cr = cross3(sub(b, a), sub(c, a));
    float z = red3(cr.v[0] * cr.v[0], cr.v[1] * cr.v[1], cr.v[2] * cr.v[2]);
    PlaneEq p;
    if (z > 0.0f) {
        float s = std::sqrt(z);
        p.n = mk(cr.v[0] / s, cr.v[1] / s, cr.v[2] / s);
    } else {
        p.n = cr;
    }
    p.d = -(dot3(cr, a));
    return p;
}

struct FrustumRange {
    int range_min[3];
    int range_dim[3];
    float planes[24];  // far, near, top, bottom, left, right (the order Frustum::Intersects tests them)
    float corners[24];
};

// Frustum::SetFromVectors (Frustum.cpp:155-219): corners (Frustum::GetCorners' order) and the six planes (far, near, top, bottom, left, right)
inline void frustum_from_vectors(const f3 &forward, const f3 &pos, const f3 &rightVec, const f3 &up, float nearDist, float farDist, float fov, float aspect,
                                 float planes[24], float corners_out[24]) {
    const float angleTangent = (float)::tan((double)(fov / 2));
    const float heightFar = angleTangent * farDist;
    const float widthFar = heightFar * aspect;
    const float heightNear = angleTangent * nearDist;
    const float widthNear = heightNear * aspect;
    const f3 farCenter = add(pos, scale(forward, farDist));
    const f3 farTopLeft = sub(add(farCenter, scale(up, heightFar)), scale(rightVec, widthFar));
    const f3 farTopRight = add(add(farCenter, scale(up, heightFar)), scale(rightVec, widthFar));
    const f3 farBotLeft = sub(sub(farCenter, scale(up, heightFar)), scale(rightVec, widthFar));
    const f3 farBotRight = add(sub(farCenter, scale(up, heightFar)), scale(rightVec, widthFar));
    const f3 nearCenter = add(pos, scale(forward, nearDist));
    const f3 nearTopLeft = sub(add(nearCenter, scale(up, heightNear)), scale(rightVec, widthNear));
    const f3 nearTopRight = add(add(nearCenter, scale(up, heightNear)), scale(rightVec, widthNear));
    const f3 nearBotLeft = sub(sub(nearCenter, scale(up, heightNear)), scale(rightVec, widthNear));
    const f3 nearBotRight = add(sub(nearCenter, scale(up, heightNear)), scale(rightVec, widthNear));

    const PlaneEq pl[6] = {
        plane_from_points(farTopRight, farTopLeft, farBotRight),      // far
        plane_from_points(nearBotLeft, nearTopLeft, nearBotRight),    // near
        plane_from_points(nearTopLeft, farTopLeft, nearTopRight),     // top
        plane_from_points(nearBotRight, farBotLeft, nearBotLeft),     // bottom
        plane_from_points(farTopLeft, nearTopLeft, farBotLeft),       // left
        plane_from_points(nearTopRight, farTopRight, nearBotRight)};  // right
    for (int i = 0; i < 6; i++) {
        planes[4 * i] = pl[i].n.v[0];
        planes[4 * i + 1] = pl[i].n.v[1];
        planes[4 * i + 2] = pl[i].n.v[2];
        planes[4 * i + 3] = pl[i].d;
    }
    const f3 corners[8] = {farTopLeft, farTopRight, farBotLeft, farBotRight, nearBotRight, nearTopLeft, nearTopRight, nearBotLeft};
    for (int i = 0; i < 8; i++)
        for (int k = 0; k < 3; k++) corners_out[3 * i + k] = corners[i].v[k];
}

// pose: row-major 3x4 camera->world.  Frustum::SetFromParams (Frustum.cpp:143-153: cx unused; PinholeCamera::SetupFrustum passes fy twice)
inline FrustumRange frustum_range(const float *pose, float nearDist, float farDist, float fy, float cy, int W, int H,
                                  int chunk_n, float res) {
    const f3 rightVec = mk(pose[0], pose[4], pose[8]);
    const f3 up = mk(-pose[1], -pose[5], -pose[9]);
    const f3 forward = mk(pose[2], pose[6], pose[10]);
    const f3 pos = mk(pose[3], pose[7], pose[11]);
    const float imgWidth = (float)W, imgHeight = (float)H;
    const float aspect = (fy * imgWidth) / (fy * imgHeight);
    const float fov = (float)(::atan2((double)cy, (double)fy) + ::atan2((double)(imgHeight - cy), (double)fy));
    FrustumRange out;
    frustum_from_vectors(forward, pos, rightVec, up, nearDist, farDist, fov, aspect, out.planes, out.corners);
    float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    float mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (int i = 0; i < 8; i++)
        for (int k = 0; k < 3; k++) {
            const float c = out.corners[3 * i + k];
            mn[k] = c < mn[k] ? c : mn[k];
            mx[k] = c > mx[k] ? c : mx[k];
        }
    const float roundingFactor = 1.0f / ((float)chunk_n * res);
    for (int k = 0; k < 3; k++) {
        const int minID = (int)std::floor(mn[k] * roundingFactor);
        const int maxID = (int)std::floor(mx[k] * roundingFactor) + 1;
        out.range_min[k] = minID - 1;
        out.range_dim[k] = (maxID + 1) - (minID - 1) + 1;
    }
    return out;
}

}  // namespace hostmath
}  // namespace chisel_hip
