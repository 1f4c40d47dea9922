// COptimisable.h -- optimiser interface of GPc (reference COptimisable.h:20-258).  Only scaled conjugate gradients
// (the `gp learn` default) is provided; it reproduces the reference's iteration structure exactly, including its
// quirks, because the number of Gram builds + factorisations per run is part of the observable behaviour
// (SURVEY.md section 3.1).
#ifndef GPC_AMD_COPTIMISABLE_H
#define GPC_AMD_COPTIMISABLE_H
#include <string>
#include "CMatrix.h"

class COptimisable {
 public:
  enum { CG, SCG, GD, BFGS, LBFGS };
  COptimisable() : iter(0), verbosity(2), defaultOptimiser(SCG), objectiveTol(1e-6), parameterTol(1e-6), maxIters(1000),
                   funcEvals(0), gradEvals(0) {}
  virtual ~COptimisable() {}
  virtual unsigned int getOptNumParams() const = 0;
  virtual void getOptParams(CMatrix& param) const = 0;
  virtual void setOptParams(const CMatrix& param) = 0;
  virtual double computeObjectiveGradParams(CMatrix& g) const = 0;
  virtual double computeObjectiveVal() const = 0;

  void setVerbosity(int v) { verbosity = v; }
  int getVerbosity() const { return verbosity; }
  void setMaxIters(unsigned int v) { maxIters = v; }
  unsigned int getMaxIters() const { return maxIters; }
  void setObjectiveTol(double v) { objectiveTol = v; }
  double getObjectiveTol() const { return objectiveTol; }
  void setParamTol(double v) { parameterTol = v; }
  double getParamTol() const { return parameterTol; }
  void setDefaultOptimiser(int v) { defaultOptimiser = v; }
  int getDefaultOptimiser() const { return defaultOptimiser; }
  std::string getDefaultOptimiserStr() const { return defaultOptimiser == SCG ? "scg" : "other"; }
  void runDefaultOptimiser();
  void scgOptimise();
  void checkGradients();
  unsigned int getIterations() const { return iter; }
  // evaluation counters (one objective evaluation = one Gram build + Cholesky unless the cache is valid)
  mutable unsigned int funcEvals, gradEvals;

 protected:
  unsigned int iter;

 private:
  int verbosity;
  int defaultOptimiser;
  double objectiveTol, parameterTol;
  unsigned int maxIters;
};

class CProbabilisticOptimisable : public COptimisable {
 public:
  virtual double logLikelihood() const = 0;
  virtual double logLikelihoodGradient(CMatrix& g) const = 0;
  double computeObjectiveGradParams(CMatrix& g) const   // COptimisable.h:247-252
  {
    gradEvals++;
    const double L = logLikelihoodGradient(g);
    g.negate();
    return -L;
  }
  double computeObjectiveVal() const   // COptimisable.h:253-256
  {
    funcEvals++;
    return -logLikelihood();
  }
};
#endif
