// truncation/*.h of the reference, facade edition: same class names and constructors; the arithmetic itself runs on
// the GPU (chisel_device.h truncation_distance), the host copy below is the same restatement for callers that ask.
#ifndef CHISEL_HIP_FACADE_TRUNCATOR_H_
#define CHISEL_HIP_FACADE_TRUNCATOR_H_
#include <cmath>
#include <memory>
#include <chisel_hip.h>  // repository include/ directory on the include path
namespace chisel {
class Truncator {  // Truncator.h:28-38
  public:
    virtual ~Truncator() {}
    virtual float GetTruncationDistance(float reading) const = 0;
    virtual int HipKind() const = 0;      // chisel_hip_truncator_kind
    virtual float HipParam() const = 0;
};
typedef std::shared_ptr<const Truncator> TruncatorPtr;
class ConstantTruncator : public Truncator {  // ConstantTruncator.h:31-56
  public:
    ConstantTruncator() = default;
    explicit ConstantTruncator(float value) : truncationDistance(value) {}
    float GetTruncationDistance(float) const override { return truncationDistance; }
    void SetTruncationDistance(float value) { truncationDistance = value; }  // ConstantTruncator.h:46
    int HipKind() const override { return CHISEL_HIP_TRUNC_CONSTANT; }
    float HipParam() const override { return truncationDistance; }
  protected:
    float truncationDistance = 0.0f;
};
class InverseTruncator : public Truncator {  // InverseTruncator.h:31-61
  public:
    InverseTruncator() = default;
    explicit InverseTruncator(float scale) : scalingFactor(scale) {}
    float GetTruncationDistance(float reading) const override {
        const float baseLine = 0.10, focal = 471.27, depSample = 1.0f / (baseLine * focal);
        const float inv = 1.0 / reading;
        return (depSample / (inv * inv)) * scalingFactor;
    }
    int HipKind() const override { return CHISEL_HIP_TRUNC_INVERSE; }
    float HipParam() const override { return scalingFactor; }
  protected:
    float scalingFactor = 1.0f;
};
class QuadraticTruncator : public Truncator {  // QuadraticTruncator.h:31-73
  public:
    QuadraticTruncator() = default;
    explicit QuadraticTruncator(float scale) : scalingFactor(scale) {}
    float GetTruncationDistance(float reading) const override {
        return std::abs(GetQuadraticTerm() * std::pow(reading, 2) + GetLinearTerm() * reading + GetConstantTerm()) * scalingFactor;
    }
    float GetQuadraticTerm() const { return 0.0019 * 10; }    // QuadraticTruncator.h:47-62 (the terms are constants of the class there too)
    float GetLinearTerm() const { return 0.00152 * 10; }
    float GetConstantTerm() const { return 0.001504 * 10; }
    float GetScalingFactor() const { return scalingFactor; }
    int HipKind() const override { return CHISEL_HIP_TRUNC_QUADRATIC; }
    float HipParam() const override { return scalingFactor; }
  protected:
    float scalingFactor = 1.0f;
};
}  // namespace chisel
#endif
