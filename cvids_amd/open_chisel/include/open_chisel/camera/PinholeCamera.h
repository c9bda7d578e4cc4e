// camera/{Intrinsics,PinholeCamera,DepthImage,ColorImage}.h of the reference, facade edition (containers and setters;
// projection and frustum construction happen inside libchisel_hip.so, SetupFrustum asks it for the result).
#ifndef CHISEL_HIP_FACADE_CAMERA_H_
#define CHISEL_HIP_FACADE_CAMERA_H_
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>
#include "../geometry/Frustum.h"
#include "../geometry/Geometry.h"
namespace chisel {
class Intrinsics {  // Intrinsics.h:31-53
  public:
    float GetFx() const { return fx; }
    float GetFy() const { return fy; }
    float GetCx() const { return cx; }
    float GetCy() const { return cy; }
    void SetFx(float v) { fx = v; }
    void SetFy(float v) { fy = v; }
    void SetCx(float v) { cx = v; }
    void SetCy(float v) { cy = v; }
  private:
    float fx = 0, fy = 0, cx = 0, cy = 0;
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
  private:
    Intrinsics intrinsics;
    int width = 640, height = 480;
    float nearPlane = 0.05f, farPlane = 5.0f;
};
template <class DataType>
class DepthImage {  // DepthImage.h:33-103 (row-major, Index = col + row * width)
  public:
    DepthImage() : width(-1), height(-1) {}
    DepthImage(int w, int h) : data((size_t)w * h), width(w), height(h) {}
    int Index(int row, int col) const { return col + row * width; }
    void SetDataAt(int row, int col, DataType d) { data[Index(row, col)] = d; }
    const DataType &DepthAt(int row, int col) const { return data[Index(row, col)]; }
    DataType *GetMutableData() { return data.data(); }
    const DataType *GetData() const { return data.data(); }
    void SetData(const DataType *d) { data.assign(d, d + (size_t)width * height); }
    int GetWidth() const { return width; }
    int GetHeight() const { return height; }
  protected:
    std::vector<DataType> data;
    int width, height;
};
template <class DataType>
class ColorImage {  // ColorImage.h:38-134 (1 = mono, 3 = BGR, 4 = BGRA)
  public:
    ColorImage() : width(-1), height(-1), numChannels(0) {}
    ColorImage(int w, int h, int c) : data((size_t)w * h * c), width(w), height(h), numChannels(c) {}
    int Index(int row, int col, int channel) const { return (col + row * width) * numChannels + channel; }
    DataType *GetMutableData() { return data.data(); }
    const DataType *GetData() const { return data.data(); }
    int GetWidth() const { return width; }
    int GetHeight() const { return height; }
    int GetNumChannels() const { return numChannels; }
  protected:
    std::vector<DataType> data;
    int width, height, numChannels;
};
}  // namespace chisel
#endif
