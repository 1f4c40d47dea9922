// CTransform.h -- parameter transforms of the optimiser space (reference CTransform.h:18-367, CTransform.cpp:20-120).
// exp ("defaultPositive") for every parameter of the in-scope kernels, sigmoid ("defaultZeroOne") for the rbfard
// input scales; the +-36 clamp and the EPS saturation are the reference's.
#ifndef GPC_AMD_CTRANSFORM_H
#define GPC_AMD_CTRANSFORM_H
#include <cmath>
#include <string>
#include <vector>
#include "CMatrix.h"

class CTransform {
 public:
  enum Kind { EXP = 0, SIGMOID = 1 };
  static double limVal() { return 36.0; }                       // CTransform.h:18
  static double eps() { return 2.220446049250313e-16; }         // ndlutil::EPS
  explicit CTransform(Kind k) : kind(k) {}
  static CTransform* defaultPositive() { return new CTransform(EXP); }
  static CTransform* defaultZeroOne() { return new CTransform(SIGMOID); }
  double atox(double a) const
  {
    if(kind == EXP) {                                           // CTransform.cpp:31-43
      if(a < -limVal()) return std::exp(-limVal());
      if(a < limVal()) return std::exp(a);
      return std::exp(limVal());
    }
    if(a < -limVal()) return eps();                             // CTransform.cpp:97-105
    if(a < limVal()) return 1.0 / (1.0 + std::exp(-a));
    return 1.0 - eps();
  }
  double xtoa(double x) const { return kind == EXP ? std::log(x) : std::log(x / (1.0 - x)); }
  double gradfact(double x) const { return kind == EXP ? x : x * (1.0 - x); }
  std::string getType() const { return kind == EXP ? "exp" : "sigmoid"; }
  Kind kind;
};

// Mix-in: a list of (transform, parameter index) pairs, CTransformable in the reference (CTransform.h:221-367).
class CTransformable {
 public:
  virtual ~CTransformable()
  {
    for(size_t i = 0; i < transforms.size(); i++) delete transforms[i];
  }
  virtual unsigned int getNumParams() const = 0;
  virtual double getParam(unsigned int index) const = 0;
  virtual void setParam(double val, unsigned int index) = 0;
  void addTransform(CTransform* t, unsigned int index)
  {
    transforms.push_back(t);
    transIndex.push_back(index);
  }
  void clearTransforms()
  {
    for(size_t i = 0; i < transforms.size(); i++) delete transforms[i];
    transforms.clear();
    transIndex.clear();
  }
  unsigned int getNumTransforms() const { return (unsigned int)transforms.size(); }
  unsigned int getTransformIndex(unsigned int i) const { return transIndex[i]; }
  const CTransform* getTransform(unsigned int i) const { return transforms[i]; }
  double getTransformGradFact(double val, unsigned int i) const { return transforms[i]->gradfact(val); }
  virtual void getTransParams(CMatrix& a) const
  {
    for(unsigned int i = 0; i < getNumParams(); i++) a.setVal(getParam(i), i);
    for(size_t t = 0; t < transforms.size(); t++) a.setVal(transforms[t]->xtoa(getParam(transIndex[t])), transIndex[t]);
  }
  virtual void setTransParams(const CMatrix& a)
  {
    for(unsigned int i = 0; i < getNumParams(); i++) setParam(a.getVal(i), i);
    for(size_t t = 0; t < transforms.size(); t++) setParam(transforms[t]->atox(a.getVal(transIndex[t])), transIndex[t]);
  }

 protected:
  std::vector<CTransform*> transforms;
  std::vector<unsigned int> transIndex;
};
#endif
