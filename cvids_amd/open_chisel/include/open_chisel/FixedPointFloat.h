// FixedPointFloat.h:30-66 of the reference for the chisel_hip facade: the 16-bit fixed-point encodings of [-1000, 1000] and [0, 1000]
// (used by no live code path of the reference: DistVoxel stores plain floats, DistVoxel.h:78-79).
#ifndef CHISEL_HIP_FACADE_FIXEDPOINTFLOAT_H_
#define CHISEL_HIP_FACADE_FIXEDPOINTFLOAT_H_
#include <stdint.h>

#include <algorithm>
#include <limits>

namespace chisel {
const float MinFloatValue = -1000;
const float MaxFloatValue = 1000;
const float MaxUFloat = 1000;
typedef uint16_t FixedFloat16;
typedef uint16_t UFixedFloat16;

inline float ClampFloat(const float &input) { return std::max(std::min(input, MaxFloatValue), MinFloatValue); }
inline float ClampUnsignedFloat(const float input) { return std::max(std::min(input, MaxUFloat), 0.0f); }
inline FixedFloat16 FloatToFixedFloat16(const float &input) {
    const float unit = (ClampFloat(input) - MinFloatValue) / (MaxFloatValue - MinFloatValue);
    return static_cast<FixedFloat16>(unit * std::numeric_limits<FixedFloat16>::max());
}
inline float FixedFloat16ToFloat(const FixedFloat16 &input) {
    const float unit = static_cast<float>(input) / std::numeric_limits<FixedFloat16>::max();
    return MinFloatValue + unit * (MaxFloatValue - MinFloatValue);
}
inline UFixedFloat16 FloatToUFixedFloat16(const float &input) {
    return static_cast<UFixedFloat16>((ClampUnsignedFloat(input) / MaxUFloat) * std::numeric_limits<UFixedFloat16>::max());
}
inline float UFixedFloat16ToFloat(const UFixedFloat16 &input) {
    const float unit = static_cast<float>(input) / std::numeric_limits<UFixedFloat16>::max();
    return unit * (MaxUFloat);
}
}  // namespace chisel
#endif
