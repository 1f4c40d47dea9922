// CKern.h -- GPc's kernel class surface (reference CKern.h:30-1229) for the kernels the accelerated FTC path covers:
// rbf, rbfard, white, bias, lin and the compound (sum) kernel.  Same constructors, parameter order, names, transforms
// and ownership rules (CCmpndKern::addKern clones, CKern.h:384) as the reference.  Whole-matrix operations --
// compute(K,X), compute(K,X,X2), diagCompute, getGradParams(g,X,covGrad) -- are ONE call into libgpc_hip.so on the flat
// kernel spec (gpc_kspec) the kernel describes itself with; the scalar computeElement/diagComputeElement members keep
// the reference's per-element formulas for callers that want a single value.
#ifndef GPC_AMD_CKERN_H
#define GPC_AMD_CKERN_H
#include <iostream>
#include <string>
#include <vector>
#include "CMatrix.h"
#include "CTransform.h"
#include "gpc_hip.h"

class CKern : public CTransformable {
 public:
  CKern() : nParams(0), inputDim(0), stationary(false) {}
  virtual ~CKern() {}
  virtual CKern* clone() const = 0;
  virtual void setInitParam() = 0;
  virtual double diagComputeElement(const CMatrix& X, unsigned int index) const = 0;
  virtual double computeElement(const CMatrix& X1, unsigned int index1, const CMatrix& X2, unsigned int index2) const = 0;
  virtual double getVariance() const = 0;
  virtual double getWhite() const { return 0.0; }
  virtual unsigned int addKern(const CKern*)
  {
    std::cerr << "You cannot add a kernel to this kernel." << std::endl;
    return 0;
  }
  // append this kernel's term(s) to a flat spec (include/gpc_hip.h)
  virtual void appendKspec(gpc_kspec& ks) const = 0;
  void toKspec(gpc_kspec& ks) const;

  // whole-matrix operations: libgpc_hip.so
  virtual void compute(CMatrix& K, const CMatrix& X) const;                       // CKern.h:128-144
  virtual void compute(CMatrix& K, const CMatrix& X, const CMatrix& X2) const;    // CKern.h:146-157
  // selected rows / columns and a single column of the Gram matrix (CKern.h:94-126, 159-167): the rows are gathered on the
  // host and go through the same device kernels
  virtual void compute(CMatrix& K, const CMatrix& X1, const std::vector<unsigned int> indices1, const CMatrix& X2,
                       const std::vector<unsigned int> indices2) const;
  virtual void compute(CMatrix& K, const CMatrix& X, const std::vector<unsigned int> indices) const;
  virtual void compute(CMatrix& K, const CMatrix& X, const CMatrix& X2, unsigned int row) const;
  virtual void diagCompute(CMatrix& d, const CMatrix& X) const;                   // CKern.h:49-55
  virtual void getGradParams(CMatrix& g, const CMatrix& X, const CMatrix& covGrad, bool regularise = true) const;
  void getGradTransParams(CMatrix& g, const CMatrix& X, const CMatrix& covGrad, bool regularise = true) const;

  unsigned int getNumParams() const { return nParams; }
  unsigned int getInputDim() const { return inputDim; }
  void setInputDim(unsigned int d)
  {
    inputDim = d;
    setInitParam();
  }
  void getParams(CMatrix& p) const
  {
    for(unsigned int i = 0; i < nParams; i++) p.setVal(getParam(i), i);
  }
  void setParams(const CMatrix& p)
  {
    for(unsigned int i = 0; i < nParams; i++) setParam(p.getVal(i), i);
  }
  std::string getType() const { return type; }
  std::string getName() const { return kernName; }
  std::string getBaseType() const { return "kern"; }
  virtual std::string getParamName(unsigned int i) const { return paramNames.at(i); }
  bool isStationary() const { return stationary; }
  double priorLogProb() const { return 0.0; }   // no priors in scope (CDist is out of scope; CGp.cpp:1011 adds 0)
  virtual void writeParamsToStream(std::ostream& out) const;
  virtual void readParamsFromStream(std::istream& in);   // CKern.cpp:100-112 (after baseType/type were consumed)
  void toStream(std::ostream& out) const
  {
    out << "version=0.200000" << std::endl;
    writeParamsToStream(out);
  }
  virtual std::ostream& display(std::ostream& os) const;

 protected:
  void setType(const std::string& t) { type = t; }
  void setName(const std::string& n) { kernName = n; }
  void setParamName(const std::string& n, unsigned int i)
  {
    if(paramNames.size() <= i) paramNames.resize(i + 1, "no name");
    paramNames[i] = n;
  }
  unsigned int nParams;
  unsigned int inputDim;
  bool stationary;
  std::string type, kernName;
  std::vector<std::string> paramNames;
};

// k = variance * exp(-0.5 * inverseWidth * |x-x'|^2)   (CKern.cpp:1027-1250; params: inverseWidth, variance)
class CRbfKern : public CKern {
 public:
  CRbfKern() { _init(); }
  explicit CRbfKern(unsigned int inDim) { _init(); setInputDim(inDim); }
  explicit CRbfKern(const CMatrix& X) { _init(); setInputDim(X.getCols()); }
  CRbfKern(const CRbfKern& k) : CKern() { _init(); setInputDim(k.getInputDim()); variance = k.variance; inverseWidth = k.inverseWidth; }
  CRbfKern* clone() const { return new CRbfKern(*this); }
  void setInitParam() { inverseWidth = 1.0; variance = 1.0; }
  double diagComputeElement(const CMatrix&, unsigned int) const { return variance; }
  double computeElement(const CMatrix& X1, unsigned int i1, const CMatrix& X2, unsigned int i2) const;
  double getVariance() const { return variance; }
  void setParam(double val, unsigned int i);
  double getParam(unsigned int i) const;
  void appendKspec(gpc_kspec& ks) const;

 private:
  void _init();
  double variance, inverseWidth;
};

// k = variance * exp(-0.5 * inverseWidth * sum_k scale_k (x_k - x'_k)^2)   (CKern.cpp:3150-3410)
class CRbfardKern : public CKern {
 public:
  CRbfardKern() { _init(); }
  explicit CRbfardKern(unsigned int inDim) { _init(); setInputDim(inDim); }
  explicit CRbfardKern(const CMatrix& X) { _init(); setInputDim(X.getCols()); }
  CRbfardKern(const CRbfardKern& k) : CKern() { _init(); setInputDim(k.getInputDim()); variance = k.variance; inverseWidth = k.inverseWidth; scales = k.scales; }
  CRbfardKern* clone() const { return new CRbfardKern(*this); }
  void setInitParam();
  double diagComputeElement(const CMatrix&, unsigned int) const { return variance; }
  double computeElement(const CMatrix& X1, unsigned int i1, const CMatrix& X2, unsigned int i2) const;
  double getVariance() const { return variance; }
  void setParam(double val, unsigned int i);
  double getParam(unsigned int i) const;
  void appendKspec(gpc_kspec& ks) const;

 private:
  void _init();
  double variance, inverseWidth;
  std::vector<double> scales;
};

class CWhiteKern : public CKern {   // CKern.cpp:600-740
 public:
  CWhiteKern() { _init(); }
  explicit CWhiteKern(unsigned int inDim) { _init(); setInputDim(inDim); }
  explicit CWhiteKern(const CMatrix& X) { _init(); setInputDim(X.getCols()); }
  CWhiteKern(const CWhiteKern& k) : CKern() { _init(); setInputDim(k.getInputDim()); variance = k.variance; }
  CWhiteKern* clone() const { return new CWhiteKern(*this); }
  void setInitParam();
  double diagComputeElement(const CMatrix&, unsigned int) const { return variance; }
  double computeElement(const CMatrix&, unsigned int, const CMatrix&, unsigned int) const { return 0.0; }
  double getVariance() const { return variance; }
  double getWhite() const { return variance; }
  void setParam(double val, unsigned int i);
  double getParam(unsigned int i) const;
  void appendKspec(gpc_kspec& ks) const;

 private:
  void _init();
  double variance;
};

class CBiasKern : public CKern {   // CKern.cpp:890-1025
 public:
  CBiasKern() { _init(); }
  explicit CBiasKern(unsigned int inDim) { _init(); setInputDim(inDim); }
  explicit CBiasKern(const CMatrix& X) { _init(); setInputDim(X.getCols()); }
  CBiasKern(const CBiasKern& k) : CKern() { _init(); setInputDim(k.getInputDim()); variance = k.variance; }
  CBiasKern* clone() const { return new CBiasKern(*this); }
  void setInitParam();
  double diagComputeElement(const CMatrix&, unsigned int) const { return variance; }
  double computeElement(const CMatrix&, unsigned int, const CMatrix&, unsigned int) const { return variance; }
  double getVariance() const { return variance; }
  void setParam(double val, unsigned int i);
  double getParam(unsigned int i) const;
  void appendKspec(gpc_kspec& ks) const;

 private:
  void _init();
  double variance;
};

class CLinKern : public CKern {   // CKern.cpp:2220-2383
 public:
  CLinKern() { _init(); }
  explicit CLinKern(unsigned int inDim) { _init(); setInputDim(inDim); }
  explicit CLinKern(const CMatrix& X) { _init(); setInputDim(X.getCols()); }
  CLinKern(const CLinKern& k) : CKern() { _init(); setInputDim(k.getInputDim()); variance = k.variance; }
  CLinKern* clone() const { return new CLinKern(*this); }
  void setInitParam() { variance = 1.0; }
  double diagComputeElement(const CMatrix& X, unsigned int i) const { return variance * X.norm2Row(i); }
  double computeElement(const CMatrix& X1, unsigned int i1, const CMatrix& X2, unsigned int i2) const
  {
    return variance * X1.dotRowRow(i1, X2, i2);
  }
  double getVariance() const { return variance; }
  void setParam(double val, unsigned int i);
  double getParam(unsigned int i) const;
  void appendKspec(gpc_kspec& ks) const;

 private:
  void _init();
  double variance;
};

// Sum of component kernels (CKern.h:360-560, CKern.cpp:120-300).  addKern CLONES its argument and owns the clone.
class CCmpndKern : public CKern {
 public:
  CCmpndKern() { _init(); }
  explicit CCmpndKern(unsigned int inDim) { _init(); setInputDim(inDim); }
  explicit CCmpndKern(const CMatrix& X) { _init(); setInputDim(X.getCols()); }
  CCmpndKern(const CCmpndKern& k);
  ~CCmpndKern();
  CCmpndKern* clone() const { return new CCmpndKern(*this); }
  void setInitParam() {}
  unsigned int addKern(const CKern* kern);
  unsigned int getNumKerns() const { return (unsigned int)components.size(); }
  const CKern* getKern(unsigned int i) const { return components[i]; }
  double diagComputeElement(const CMatrix& X, unsigned int index) const;
  double computeElement(const CMatrix& X1, unsigned int i1, const CMatrix& X2, unsigned int i2) const;
  double getVariance() const;
  double getWhite() const;
  void setParam(double val, unsigned int paramNo);
  double getParam(unsigned int paramNo) const;
  std::string getParamName(unsigned int paramNo) const;
  void appendKspec(gpc_kspec& ks) const;
  void writeParamsToStream(std::ostream& out) const;
  void readParamsFromStream(std::istream& in);            // CComponentKern, CKern.cpp:125-140
  std::ostream& display(std::ostream& os) const;

 private:
  void _init();
  std::vector<CKern*> components;
};
// readKernFromStream (CKern.cpp:4192-4260) for the kernel types of the accelerated set; the caller owns the result.
CKern* readKernFromStream(std::istream& in);
#endif
