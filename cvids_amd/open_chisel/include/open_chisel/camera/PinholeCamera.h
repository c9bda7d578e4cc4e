// camera/{Intrinsics,PinholeCamera,DepthImage,ColorImage}.h of the reference, facade edition (containers and setters;
// projection and frustum construction happen inside libchisel_hip.so, SetupFrustum asks it for the result).
#ifndef CHISEL_HIP_FACADE_CAMERA_H_
#define CHISEL_HIP_FACADE_CAMERA_H_
#include <chisel_hip.h>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <new>
#include "../geometry/Frustum.h"
#include "../geometry/Geometry.h"
#include "../geometry/Interpolate.h"  // (its one caller is DepthImage::BilinearInterpolateDepth below)
namespace chisel {
class Intrinsics {  // Intrinsics.h:31-53: the 3x3 matrix K; fx, fy, cx, cy are entries of it
  public:
    Intrinsics() {
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) matrix(r, c) = 0.0f;  // (the reference leaves K uninitialised)
    }
    float GetFx() const { return matrix(0, 0); }
    float GetFy() const { return matrix(1, 1); }
    float GetCx() const { return matrix(0, 2); }
    float GetCy() const { return matrix(1, 2); }
    void SetFx(float v) { matrix(0, 0) = v; }
    void SetFy(float v) { matrix(1, 1) = v; }
    void SetCx(float v) { matrix(0, 2) = v; }
    void SetCy(float v) { matrix(1, 2) = v; }
    const Mat3x3 &GetMatrix() const { return matrix; }
    Mat3x3 &GetMutableMatrix() { return matrix; }
    void SetMatrix(const Mat3x3 &m) { matrix = m; }
  private:
    Mat3x3 matrix;
};
class PinholeCamera {  // PinholeCamera.h:35-69
  public:
    const Intrinsics &GetIntrinsics() const { return intrinsics; }
    Intrinsics &GetMutableIntrinsics() { return intrinsics; }
    void SetIntrinsics(const Intrinsics &v) { intrinsics = v; }
    int GetWidth() const { return width; }
    int GetHeight() const { return height; }
    void SetWidth(int v) { width = v; }
    void SetHeight(int v) { height = v; }
    float GetNearPlane() const { return nearPlane; }
    float GetFarPlane() const { return farPlane; }
    void SetNearPlane(float v) { nearPlane = v; }
    void SetFarPlane(float v) { farPlane = v; }
    // PinholeCamera.cpp:55-59: SetFromParams(view, near, far, fy, fy, cx, cy, width, height) -- fy twice, as the reference
    void SetupFrustum(const Transform &view, Frustum *frustum) const {
        frustum->SetFromParams(view, nearPlane, farPlane, intrinsics.GetFy(), intrinsics.GetFy(), intrinsics.GetCx(), intrinsics.GetCy(),
                               (float)width, (float)height);
    }
    // PinholeCamera.cpp:38-64 for a caller's own points (the kernels inline the same arithmetic: chisel_device.h)
    Vec3 ProjectPoint(const Vec3 &point) const {
        const float invZ = 1.0f / point(2);
        return Vec3(intrinsics.GetFx() * point(0) * invZ + intrinsics.GetCx(), intrinsics.GetFy() * point(1) * invZ + intrinsics.GetCy(), point(2));
    }
    Vec3 UnprojectPoint(const Vec3 &point) const {
        const float z = point(2);
        return Vec3(z * ((point(0) - intrinsics.GetCx()) / intrinsics.GetFx()), z * ((point(1) - intrinsics.GetCy()) / intrinsics.GetFy()), z);
    }
    bool IsPointOnImage(const Vec3 &point) const { return point(0) >= 0 && point(1) >= 0 && point(0) < width && point(1) < height; }
  private:
    Intrinsics intrinsics;
    int width = 640, height = 480;
    float nearPlane = 0.05f, farPlane = 5.0f;
};
// The pixel buffer of an image: the reference's `new DataType[n]` (DepthImage.h:42-52, ColorImage.h:44-58), here page-locked host memory
// from the library (chisel_hip_host_alloc).  chisel_ros allocates its images once and refills them every frame
// (ChiselServer.cpp:268-273,287-292): an integrate call then reads the frame without the runtime's staged copy of pageable memory.
template <class T>
class ImageBuffer {
  public:
    ImageBuffer() : p(nullptr), n(0) {}
    explicit ImageBuffer(size_t count) : p(nullptr), n(0) { resize(count); }
    ImageBuffer(const ImageBuffer &o) : p(nullptr), n(0) { assign(o.p, o.p + o.n); }
    ImageBuffer &operator=(const ImageBuffer &o) {
        if (this != &o) assign(o.p, o.p + o.n);
        return *this;
    }
    ~ImageBuffer() { chisel_hip_host_free(p); }
    void resize(size_t count) {  // (value-initialised, like std::vector)
        if (count != n) {
            chisel_hip_host_free(p);
            p = count ? static_cast<T *>(chisel_hip_host_alloc(count * sizeof(T))) : nullptr;
            if (count && !p) throw std::bad_alloc();
            n = count;
        }
        for (size_t i = 0; i < n; i++) p[i] = T();
    }
    void assign(const T *first, const T *last) {
        const size_t count = (size_t)(last - first);
        if (count != n) {
            chisel_hip_host_free(p);
            p = count ? static_cast<T *>(chisel_hip_host_alloc(count * sizeof(T))) : nullptr;
            if (count && !p) throw std::bad_alloc();
            n = count;
        }
        if (count) std::memcpy(p, first, count * sizeof(T));
    }
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
  private:
    T *p;
    size_t n;
};
template <class DataType>
class DepthImage {  // DepthImage.h:33-103 (row-major, Index = col + row * width)
  public:
    DepthImage() : width(-1), height(-1) {}
    DepthImage(int w, int h) : data((size_t)w * h), width(w), height(h) {}
    int Index(int row, int col) const { return col + row * width; }
    void SetDataAt(int row, int col, DataType d) { data[Index(row, col)] = d; }
    float DepthAt(int row, int col) const { return static_cast<float>(data[Index(row, col)]); }
    const DataType &At(int row, int col) const { return data[Index(row, col)]; }
    DataType &AtMutable(int row, int col) { return data[Index(row, col)]; }
    bool IsInside(int row, int col) const { return row >= 0 && row < width && col >= 0 && col < height; }  // DepthImage.h:86-89 (rows against the width: sic)
    float BilinearInterpolateDepth(float x, float y) const {  // DepthImage.h:59-68 (no caller on the live path: ProjectionIntegrator.h:72,131 are commented out)
        const int gxi = static_cast<int>(x), gyi = static_cast<int>(y);
        return BilinearInterpolate(DepthAt(gyi, gxi), DepthAt(gyi, gxi + 1), DepthAt(gyi + 1, gxi), DepthAt(gyi + 1, gxi + 1), x - gxi, y - gyi);
    }
    DataType *GetMutableData() { return data.data(); }
    const DataType *GetData() const { return data.data(); }
    void SetData(const DataType *d) { data.assign(d, d + (size_t)width * height); }
    int GetWidth() const { return width; }
    int GetHeight() const { return height; }
  protected:
    ImageBuffer<DataType> data;
    int width, height;
};
template <class DataType = uint8_t>
struct Color {  // ColorImage.h:30-36
    DataType red, green, blue, alpha;
};
template <class DataType>
class ColorImage {  // ColorImage.h:38-134 (1 = mono, 3 = BGR, 4 = BGRA)
  public:
    ColorImage() : width(-1), height(-1), numChannels(0) {}
    ColorImage(int w, int h, int c) : data((size_t)w * h * c), width(w), height(h), numChannels(c) {}
    int Index(int row, int col, int channel) const { return (col + row * width) * numChannels + channel; }
    // ColorImage.h:66-101: channel order by channel count -- what color_at() of the integration kernel decodes (chisel_device.h)
    void At(int row, int col, Color<DataType> *out) const {
        const DataType *p = &data[(size_t)Index(row, col, 0)];
        if (numChannels == 1) { out->red = out->green = out->blue = out->alpha = p[0]; }
        else if (numChannels == 2) { out->red = p[0]; out->green = out->blue = out->alpha = p[1]; }
        else if (numChannels == 3) { out->red = p[2]; out->green = p[1]; out->blue = p[0]; out->alpha = p[2]; }
        else if (numChannels == 4) { out->red = p[2]; out->green = p[1]; out->blue = p[0]; out->alpha = p[3]; }
    }
    const DataType &At(int row, int col, int channel) const { return data[(size_t)Index(row, col, channel)]; }
    DataType &AtMutable(int row, int col, int channel) { return data[(size_t)Index(row, col, channel)]; }
    bool IsInside(int row, int col) const { return row >= 0 && row < width && col >= 0 && col < height; }  // ColorImage.h:115-118 (sic)
    DataType *GetMutableData() { return data.data(); }
    const DataType *GetData() const { return data.data(); }
    int GetWidth() const { return width; }
    int GetHeight() const { return height; }
    int GetNumChannels() const { return numChannels; }
  protected:
    ImageBuffer<DataType> data;
    int width, height, numChannels;
};
}  // namespace chisel
#endif
