// weighting/Weighter.h + ConstantWeighter.h of the reference, facade edition
#ifndef CHISEL_HIP_FACADE_WEIGHTER_H_
#define CHISEL_HIP_FACADE_WEIGHTER_H_
#include <memory>
namespace chisel {
class Weighter {  // Weighter.h:28-38
  public:
    virtual ~Weighter() {}
    virtual float GetWeight(float surfaceDist, float truncationDist) const = 0;
    virtual float HipWeight() const = 0;
};
typedef std::shared_ptr<const Weighter> WeighterPtr;
class ConstantWeighter : public Weighter {  // ConstantWeighter.h:31-51
  public:
    ConstantWeighter() = default;
    explicit ConstantWeighter(float w) : weight(w) {}
    float GetWeight(float, float truncationDist) const override { return weight / (5 * truncationDist); }
    float HipWeight() const override { return weight; }
  protected:
    float weight = 1.0f;
};
}  // namespace chisel
#endif
