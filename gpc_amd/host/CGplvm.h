// CGplvm.h -- GPc's Gaussian-process latent variable model (reference CGplvm.h:20-326, CGplvm.cpp) for the plain
// configuration `gplvm learn` builds by default: kernel on the latent points + CScaleNoise, PCA initialisation,
// latent regulariser; optimised over [transformed kernel parameters, X(:)] by SCG.
//
// Every objective evaluation runs on the GPU through the C-ABI (include/gpc_hip.h): Gram of the latent points,
// Cholesky (no jitter: CGplvm::_updateInvK calls plain chol(), CGplvm.cpp:435-446), log-det, inverse, A = invK m,
// G = sum_j covGrad_j (gpc_covgrad_multi_f64), kernel-parameter gradient (gpc_kern_grad_f64) and dL/dX
// (gpc_kern_gradx_f64).  The reference keeps N matrices of N x q for dK/dX and loops over the d outputs; here nothing
// larger than N x N exists and covGrad is traversed twice per evaluation instead of 2 d times.
//
// Not provided (ndlexceptions::NotImplementedError): dynamics kernels, back constraints, learnt output scales,
// the sparse approximations; posteriorMeanVar / out (only `learn` is on the hot path).
#ifndef GPC_AMD_CGPLVM_H
#define GPC_AMD_CGPLVM_H
#include <iostream>
#include <string>
#include <vector>
#include "CKern.h"
#include "CMatrix.h"
#include "CNoise.h"
#include "COptimisable.h"

class CGplvm : public CProbabilisticOptimisable {
 public:
  CGplvm(CKern* kernel, CScaleNoise* nois, int latDim = 2, int verbos = 2);   // CGplvm.cpp:17-36 (runs initXpca)
  CGplvm();   // empty model for readGplvmFromStream (CGplvm.cpp:905-910): owns the kernel, noise and Y it reads
  ~CGplvm();

  void initXpca();                              // CGplvm.cpp:157-192
  void updateX() { KupToDate = false; }         // CGplvm.cpp:224-245
  unsigned int getOptNumParams() const { return pkern->getNumParams() + getNumData() * getLatentDim(); }
  void getOptParams(CMatrix& param) const;      // CGplvm.cpp:257-290: kernel (transformed), then X column by column
  void setOptParams(const CMatrix& param);      // CGplvm.cpp:292-330
  double logLikelihood() const;                 // CGplvm.cpp:493-553
  double logLikelihoodGradient(CMatrix& g) const;   // CGplvm.cpp:555-716
  void optimise(const int iters = 1000);        // CGplvm.cpp:722-738
  void display(std::ostream& os) const;         // CGplvm.cpp:745-759

  void setLatentRegularised(bool v) { regulariseLatent = v; KupToDate = false; }
  bool isLatentRegularised() const { return regulariseLatent; }
  void setInputScaleLearnt(bool v)
  {
    if(v) throw ndlexceptions::NotImplementedError("learnt output scales are outside the accelerated GP-LVM path");
  }
  bool isInputScaleLearnt() const { return false; }
  bool isDynamicModelLearnt() const { return false; }
  bool isBackConstrained() const { return false; }
  void setLabels(const std::vector<int>& l) { labels = l; }
  bool isLabels() const { return !labels.empty(); }
  unsigned int getNumData() const { return numData; }
  unsigned int getLatentDim() const { return latentDim; }
  unsigned int getNumProcesses() const { return dataDim; }
  double getLogDetK() const { updateK(); return logDetK; }

  // text model file (CGplvm.cpp:761-800)
  void writeParamsToStream(std::ostream& out) const;
  void toStream(std::ostream& out) const;
  void readParamsFromStream(std::istream& in);   // CGplvm.cpp:802-899
  void fromStream(std::istream& in);

  CMatrix* pX;    // latent points (owned; public in the reference as well, CGplvm.h:260)
  CMatrix m;      // centred / scaled targets (CGplvm.h:263)

 private:
  void updateK() const;       // CGplvm.cpp:402-446: Gram, chol, logDet, pdinv -- and A = invK m, the quadratic forms
  void releaseDevice();
  CKern* pkern;
  CScaleNoise* pnoise;
  CMatrix* pYown;   // non-null when the model was read from a stream: Y, the kernel and the noise model are its own
  unsigned int latentDim, dataDim, numData;
  bool regulariseLatent;
  std::vector<int> labels;
  mutable bool KupToDate;
  mutable double* dX;      // N x q
  mutable double* dM;      // N x d
  mutable double* dK;      // N x N: invK (full symmetric)
  mutable double* dL;      // N x N: K, then its lower factor
  mutable double* dA;      // N x d: invK * m
  mutable double* dG;      // N x N: summed covGrad
  mutable double* dGX;     // N x q
  mutable std::vector<double> quad;   // m_j' invK m_j
  mutable double logDetK;
};

void writeGplvmToStream(const CGplvm& model, std::ostream& out);
void writeGplvmToFile(const CGplvm& model, const std::string modelFileName, const std::string comment = "");
CGplvm* readGplvmFromStream(std::istream& in);
CGplvm* readGplvmFromFile(const std::string modelFileName, int verbosity = 2);
#endif
