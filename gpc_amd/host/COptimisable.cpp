// COptimisable.cpp -- scaled conjugate gradients (Moller 1993) as `gp learn` runs it.  The arithmetic and the order of
// model evaluations follow the reference's optimiser (COptimisable.cpp:246-396) because the trajectory is observable
// behaviour: the number of Gram builds + factorisations of a run and where it stops (tests: the 91-iteration sinc run).
// Three quirks of the reference are part of that behaviour and are kept, each marked where it happens:
//   Q1  the trust-region shift adds (lambda - lambdaBar) * |d| to the curvature, not * |d|^2   (COptimisable.cpp:320-322)
//   Q2  the parameter test of the stopping rule uses CMatrix::max(), i.e. the first and last entry of d only  (:385)
//   Q3  ... and its objective test compares the new value with itself (the old one has been overwritten)       (:385)
// The organisation below is this repository's own: the optimiser's state is one object, each phase of an iteration a
// method, the vectors are CMatrix rows only because their BLAS-1 helpers fix the order of the floating-point sums.
#include "COptimisable.h"
#include <cmath>
#include <iostream>

void COptimisable::runDefaultOptimiser()
{
  if(defaultOptimiser != SCG)
    throw ndlexceptions::NotImplementedError("only the scaled conjugate gradient optimiser is provided");
  scgOptimise();
}

namespace {

class ScgRun {
 public:
  ScgRun(COptimisable& model_, unsigned int dim_)
      : model(model_), dim(dim_), here(1, dim_), probe(1, dim_), downhill(1, dim_), dir(1, dim_), downhillNext(1, dim_),
        hessDir(1, dim_), trust(1.0), trustCarry(0.0), curvature(0.0), slope(0.0), stepLength(0.0), gain(0.0), fHere(0.0),
        fTrial(0.0), accepted(true)
  {
    model.getOptParams(here);
    fHere = model.computeObjectiveGradParams(downhill);
    downhill.negate();              // steepest descent to start with
    dir.deepCopy(downhill);
  }

  // one iteration; `restart` = drop the conjugacy and take the negative gradient as the next direction
  void iterate(bool restart)
  {
    const double len = dir.normRow(0), len2 = len * len;
    if(accepted) curvature = finiteDifferenceCurvature(len);
    shiftByTrustRegion(len, len2);
    slope = dir.dotRowRow(0, downhill, 0);
    stepLength = slope / curvature;
    moveTo(stepLength);
    fTrial = model.computeObjectiveVal();
    gain = 2.0 * curvature * (fHere - fTrial) / (slope * slope);   // actual over predicted decrease
    if(gain >= 0.0)
      acceptStep(restart);
    else
      rejectStep();
    if(gain < 0.25) trust *= 4.0;   // a poor quadratic model: lean towards gradient descent
  }

  // the reference's stopping rule, quirks Q2 and Q3 included
  bool converged() const
  {
    return accepted && std::fabs(dir.max() * stepLength) < model.getParamTol() &&
           std::fabs(fTrial - fHere) < model.getObjectiveTol();
  }
  double objective() const { return fHere; }
  double scale() const { return trust; }
  double lastObjectiveChange() const { return std::fabs(fTrial - fHere); }

 private:
  // d' H d from a one-sided difference of the gradient along d (step 1e-4 / |d|)
  double finiteDifferenceCurvature(double len)
  {
    const double h = 1.0e-4 / len, hInv = 1.0 / h;
    moveTo(h);
    model.computeObjectiveGradParams(hessDir);
    hessDir.scale(hInv);            // (g(x + h d) - g(x)) / h with -g(x) = downhill
    hessDir.axpy(downhill, hInv);
    return hessDir.dotRowRow(0, dir, 0);
  }

  void moveTo(double t)
  {
    probe.deepCopy(here);
    probe.axpy(dir, t);
    model.setOptParams(probe);
  }

  // Levenberg-Marquardt style regularisation of the curvature; makes it positive if it is not
  void shiftByTrustRegion(double len, double len2)
  {
    const double shift = trust - trustCarry;
    hessDir.axpy(dir, shift);
    curvature += shift * len;       // Q1: the reference multiplies by |d| here
    if(curvature <= 0.0) {
      const double perLen2 = curvature / len2;
      hessDir.axpy(dir, trust - 2.0 * perLen2);
      trustCarry = 2.0 * (trust - perLen2);
      curvature = trust * len2 - curvature;
      trust = trustCarry;
    }
  }

  void acceptStep(bool restart)
  {
    here.deepCopy(probe);
    fHere = fTrial;                 // (this is what turns the objective test of converged() into Q3)
    model.computeObjectiveGradParams(downhillNext);   // same parameters as the trial evaluation: served from the model's cache
    downhillNext.negate();
    trustCarry = 0.0;
    accepted = true;
    if(restart) {
      dir.deepCopy(downhillNext);
    } else {
      // Polak-Ribiere-like update in Moller's form
      const double nextNorm2 = downhillNext.norm2Row(0);
      const double overlap = downhill.dotRowRow(0, downhillNext, 0);
      dir.scale((nextNorm2 - overlap) / slope);
      dir.axpy(downhillNext, 1.0);
    }
    downhill.deepCopy(downhillNext);
    if(gain >= 0.75) trust *= 0.5;  // a good quadratic model: trust it more
    if(trust < 1e-15) trust = 1e-15;
  }

  void rejectStep()
  {
    model.setOptParams(here);
    trustCarry = trust;
    accepted = false;
  }

  COptimisable& model;
  unsigned int dim;
  CMatrix here, probe, downhill, dir, downhillNext, hessDir;   // position, trial point, -gradient, search direction, ...
  double trust, trustCarry;      // Moller's lambda and lambda-bar
  double curvature, slope, stepLength, gain, fHere, fTrial;
  bool accepted;
};

}  // namespace

void COptimisable::scgOptimise()
{
  if(getVerbosity() > 2) std::cout << "Scaled Conjugate Gradient Optimisation." << std::endl;
  const unsigned int dim = getOptNumParams();
  ScgRun run(*this, dim);
  for(iter = 1; iter <= getMaxIters(); iter++) {
    run.iterate(iter % dim == 0);
    if(getVerbosity() > 2)
      std::cout << "Iteration: " << iter << " Error: " << run.objective() << " Scale: " << run.scale() << std::endl;
    if(run.converged()) {
      if(getVerbosity() > 2) {
        std::cout << "Convergence criterion for parameters and objective met" << std::endl;
        std::cout << "Largest tolerance " << run.lastObjectiveChange() << std::endl;
      }
      return;
    }
  }
  std::cout << "Warning: Maximum number of iterations has been exceeded" << std::endl;
}

using std::cout;
using std::endl;

void COptimisable::checkGradients()
{
  // central differences against the analytic gradient (the reference runs this at verbosity 3, CGp.cpp:1544-1545)
  const unsigned int n = getOptNumParams();
  CMatrix params(1, n), g(1, n), origParams(1, n);
  getOptParams(origParams);
  computeObjectiveGradParams(g);
  const double change = 1e-6;
  for(unsigned int j = 0; j < n; j++) {
    params.deepCopy(origParams);
    params.setVal(origParams.getVal(j) + change, j);
    setOptParams(params);
    const double Lplus = computeObjectiveVal();
    params.setVal(origParams.getVal(j) - change, j);
    setOptParams(params);
    const double Lminus = computeObjectiveVal();
    const double diff = (Lplus - Lminus) / (2.0 * change);
    cout << "Param " << j << ": analytic " << g.getVal(j) << " numeric " << diff << " difference " << diff - g.getVal(j)
         << endl;
  }
  setOptParams(origParams);
}
