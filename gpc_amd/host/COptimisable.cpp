// COptimisable.cpp -- Moller's scaled conjugate gradients with the control flow of the reference
// (COptimisable.cpp:246-396), restated.  Quirks kept on purpose (they decide when the run stops and how many
// factorisations it costs):
//   * step 3 adds lambdaDiff * |p| (not |p|^2) to delta                               (COptimisable.cpp:320-322)
//   * the convergence test uses CMatrix::max(), which looks only at the first and the last element of p, and compares
//     newObj with oldObj AFTER oldObj has been overwritten by newObj                  (COptimisable.cpp:385)
#include "COptimisable.h"
#include <cmath>
#include <iostream>

using std::cout;
using std::endl;

void COptimisable::runDefaultOptimiser()
{
  if(defaultOptimiser != SCG)
    throw ndlexceptions::NotImplementedError("only the scaled conjugate gradient optimiser is provided");
  scgOptimise();
}

void COptimisable::scgOptimise()
{
  if(getVerbosity() > 2) cout << "Scaled Conjugate Gradient Optimisation." << endl;
  const unsigned int nParams = getOptNumParams();
  CMatrix w(1, nParams), wPlus(1, nParams), r(1, nParams), p(1, nParams), rp(1, nParams), s(1, nParams);
  getOptParams(w);
  const double m_step = 1.0e-4, m_reg = 1.0;
  bool success = true;
  double lambda = m_reg, lambdaBar = 0.0, sigma = 0.0, delta = 0.0, alpha = 0.0, mu = 0.0, newObj = 0.0, Delta = 0.0;

  double oldObj = computeObjectiveGradParams(r);
  r.negate();
  p.deepCopy(r);

  for(iter = 1; iter <= getMaxIters(); iter++) {
    const double normp = p.normRow(0);
    const double normp2 = normp * normp;
    if(success) {   // 2: second-order information from a finite difference of the gradient along p
      sigma = m_step / normp;
      wPlus.deepCopy(w);
      wPlus.axpy(p, sigma);
      setOptParams(wPlus);
      computeObjectiveGradParams(s);
      const double sigmaInv = 1.0 / sigma;
      s.scale(sigmaInv);
      s.axpy(r, sigmaInv);
      delta = s.dotRowRow(0, p, 0);
    }
    // 3: scale
    const double lambdaDiff = lambda - lambdaBar;
    s.axpy(p, lambdaDiff);
    delta += lambdaDiff * normp;
    // 4: make the Hessian estimate positive definite
    if(delta <= 0.0) {
      const double deltaOverNormp2 = delta / normp2;
      s.axpy(p, (lambda - 2.0 * deltaOverNormp2));
      lambdaBar = 2.0 * (lambda - deltaOverNormp2);
      delta = lambda * normp2 - delta;
      lambda = lambdaBar;
    }
    // 5: step size
    mu = p.dotRowRow(0, r, 0);
    alpha = mu / delta;
    // 6: comparison parameter
    wPlus.deepCopy(w);
    wPlus.axpy(p, alpha);
    setOptParams(wPlus);
    newObj = computeObjectiveVal();
    Delta = 2.0 * delta * (oldObj - newObj) / (mu * mu);
    // 7: accept / reject
    if(Delta >= 0.0) {
      w.deepCopy(wPlus);
      oldObj = newObj;
      computeObjectiveGradParams(rp);   // parameters unchanged since step 6: the model's cache makes this cheap
      rp.negate();
      lambdaBar = 0;
      success = true;
      if(iter % nParams == 0) {
        p.deepCopy(rp);   // 7.a restart
      } else {
        const double rpnorm2 = rp.norm2Row(0);
        const double rrp = r.dotRowRow(0, rp, 0);
        const double beta = (rpnorm2 - rrp) / mu;
        p.scale(beta);
        p.axpy(rp, 1.0);
      }
      r.deepCopy(rp);
      if(Delta >= 0.75) lambda *= 0.5;   // 7.b
      if(lambda < 1e-15) lambda = 1e-15;
    } else {
      setOptParams(w);
      lambdaBar = lambda;
      success = false;
    }
    if(Delta < 0.25) lambda *= 4.0;   // 8
    if(getVerbosity() > 2) cout << "Iteration: " << iter << " Error: " << oldObj << " Scale: " << lambda << endl;
    // 9: convergence (see the header comment for what this really tests)
    if(success && std::fabs(p.max() * alpha) < getParamTol() && std::fabs(newObj - oldObj) < getObjectiveTol()) {
      if(getVerbosity() > 2) {
        cout << "Convergence criterion for parameters and objective met" << endl;
        cout << "Largest tolerance " << std::fabs(newObj - oldObj) << endl;
      }
      return;
    }
  }
  cout << "Warning: Maximum number of iterations has been exceeded" << endl;
}

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
