// geometry/Raycast.h:6-9 of the reference (Raycast.cpp:4-128) for the chisel_hip facade.  Raycast() -- the Amanatides-Woo walk through
// the integer grid, its cells clipped to [min, max) -- runs on the device (chisel_hip_raycast, the walk of the point-cloud fusion
// kernels); the three scalar helpers it is built from are one-liners and stay inline, in the reference's own arithmetic
// (`mod` through the double fmod: Raycast.cpp includes only <cmath>'s overloads).
#ifndef CHISEL_HIP_FACADE_RAYCAST_H_
#define CHISEL_HIP_FACADE_RAYCAST_H_
#include <chisel_hip.h>

#include <cassert>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <vector>

#include "Geometry.h"

inline float signum(int x) { return x == 0 ? 0 : x < 0 ? -1 : 1; }
inline float mod(float value, float modulus) { return (float)std::fmod(std::fmod((double)value, (double)modulus) + (double)modulus, (double)modulus); }
inline float intbound(float s, int ds) {  // the smallest positive t such that s + t * ds is an integer
    if (ds == 0) return (float)std::numeric_limits<double>::max();
    if (ds < 0) return intbound(-s, -ds);
    s = mod(s, 1.0f);
    return (1 - s) / ds;
}
inline void Raycast(const chisel::Vec3 &start, const chisel::Vec3 &end, const chisel::Point3 &min, const chisel::Point3 &max, chisel::Point3List *output) {
    assert(!!output);
    const float a[3] = {start(0), start(1), start(2)}, b[3] = {end(0), end(1), end(2)};
    const int lo[3] = {min(0), min(1), min(2)}, hi[3] = {max(0), max(1), max(2)};
    std::vector<int> cells(3 * 256);
    int64_t n = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        if (chisel_hip_raycast(a, b, lo, hi, cells.data(), (int64_t)(cells.size() / 3), &n) != CHISEL_HIP_OK) throw std::runtime_error(chisel_hip_last_error());
        if ((size_t)n <= cells.size() / 3) break;
        cells.resize(3 * (size_t)n);
    }
    for (int64_t i = 0; i < n; i++) output->push_back(chisel::Point3(cells[3 * i], cells[3 * i + 1], cells[3 * i + 2]));
}
#endif
