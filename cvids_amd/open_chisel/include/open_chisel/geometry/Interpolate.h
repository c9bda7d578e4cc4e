// geometry/Interpolate.h:26-35 of the reference for the chisel_hip facade.  Off the live path there (its only caller,
// DepthImage::BilinearInterpolateDepth, is called from nowhere: ProjectionIntegrator.h:72,131 use the truncating lookup); pinned bit for
// bit against the reference's own header by tests/golden/ref_kat.json.
#ifndef CHISEL_HIP_FACADE_INTERPOLATE_H_
#define CHISEL_HIP_FACADE_INTERPOLATE_H_
namespace chisel {
inline float LinearInterpolate(float s, float e, float t) { return s + (e - s) * t; }
inline float BilinearInterpolate(float c00, float c10, float c01, float c11, float tx, float ty) {
    return LinearInterpolate(LinearInterpolate(c00, c10, tx), LinearInterpolate(c01, c11, tx), ty);
}
}  // namespace chisel
#endif
