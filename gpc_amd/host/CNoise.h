// CNoise.h -- the Gaussian noise model of the FTC path (reference CNoise.h / CNoise.cpp:330-500): it only supplies
// the output bias and sigma2 that CGp::out adds to the predictive mean / variance.
#ifndef GPC_AMD_CNOISE_H
#define GPC_AMD_CNOISE_H
#include <cmath>
#include <iostream>
#include <string>
#include "CMatrix.h"

class CNoise {
 public:
  CNoise() : py(0) {}
  virtual ~CNoise() {}
  CMatrix* py;   // target data, borrowed (public in the reference too: CGp takes it from here, CGp.cpp:72)
  unsigned int getNumData() const { return py ? py->getRows() : 0; }
  virtual std::string getType() const = 0;
  virtual unsigned int getOutputDim() const = 0;
  virtual void out(CMatrix& yPred, const CMatrix& mu, const CMatrix& varSigma) const = 0;
  virtual void out(CMatrix& yPred, CMatrix& errorBar, const CMatrix& mu, const CMatrix& varSigma) const = 0;
  virtual void writeParamsToStream(std::ostream& out) const = 0;
  virtual std::ostream& display(std::ostream& os) const = 0;
  std::string getBaseType() const { return "noise"; }
};

class CGaussianNoise : public CNoise {
 public:
  explicit CGaussianNoise(CMatrix* pyin) : sigma2(1e-6), bias(1, pyin->getCols(), 0.0) { py = pyin; }   // CNoise.cpp:340-346
  std::string getType() const { return "gaussian"; }
  unsigned int getOutputDim() const { return py->getCols(); }
  void setBias(double v) { bias.setVals(v); }
  void setBias(const CMatrix& b) { bias.deepCopy(b); }
  double getBiasVal(unsigned int j) const { return bias.getVal(0, j); }
  double getSigma2() const { return sigma2; }
  void setSigma2(double v) { sigma2 = v; }
  // CGaussianNoise::out, CNoise.cpp:475-490
  void out(CMatrix& yPred, const CMatrix& mu, const CMatrix&) const
  {
    for(unsigned int i = 0; i < yPred.getRows(); i++)
      for(unsigned int j = 0; j < yPred.getCols(); j++) yPred.setVal(mu.getVal(i, j) + bias.getVal(0, j), i, j);
  }
  void out(CMatrix& yPred, CMatrix& errorBar, const CMatrix& mu, const CMatrix& varSigma) const
  {
    out(yPred, mu, varSigma);
    for(unsigned int i = 0; i < yPred.getRows(); i++)
      for(unsigned int j = 0; j < yPred.getCols(); j++) errorBar.setVal(std::sqrt(varSigma.getVal(i, j) + sigma2), i, j);
  }
  void writeParamsToStream(std::ostream& out) const
  {
    out << "baseType=noise" << std::endl << "type=gaussian" << std::endl;
    out << "outputDim=" << getOutputDim() << std::endl << "numParams=" << getOutputDim() + 1 << std::endl;
    CMatrix par(1, getOutputDim() + 1);
    for(unsigned int j = 0; j < getOutputDim(); j++) par.setVal(bias.getVal(0, j), 0, j);
    par.setVal(sigma2, 0, getOutputDim());
    out << "version=0.200000" << std::endl;
    par.writeParamsToStream(out);
  }
  std::ostream& display(std::ostream& os) const
  {
    os << "Gaussian Noise: " << std::endl;
    for(unsigned int j = 0; j < bias.getCols(); j++) os << "Bias on process " << j << ": " << bias.getVal(0, j) << std::endl;
    os << "Variance: " << sigma2 << std::endl;
    return os;
  }

 private:
  double sigma2;
  CMatrix bias;
};
#endif
