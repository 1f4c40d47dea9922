// CGp.cpp -- see CGp.h.  Every O(N^2) / O(N^3) step is a call into libgpc_hip.so; the host keeps the dirty flags,
// the parameter vector and the O(N d) / O(N* d) results.
#include "CGp.h"
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <vector>
#include "gpc_hip.h"
#include "ndlstream.h"

static const double HALFLOGTWOPI = 0.91893853320467274178;   // ndlutil::HALFLOGTWOPI

namespace {
void devFree(double*& p)
{
  if(p) (void)gpc_free(p);
  p = 0;
}
double* devAlloc(size_t n)
{
  void* d = 0;
  gpcCheck(gpc_malloc(&d, sizeof(double) * (n ? n : 1)));
  return static_cast<double*>(d);
}
}  // namespace

CGp::CGp(CKern* kernel, CNoise* nois, CMatrix* Xin, int approxType, unsigned int actSetSize, int verbos)
    : pX(Xin), py(nois->py), pkern(kernel), pnoise(nois), ownsKernNoise(false), fileNumData(0), fileInputDim(0),
      numActive(actSetSize), scale(1, nois->getOutputDim(), 1.0),
      bias(1, nois->getOutputDim(), 0.0), refTransRounding(true), MupToDate(false), KupToDate(false),
      AlphaUpToDate(false), invKupToDate(false), LcholRounded(false), dX(0), dM(0), dL(0), dInvKm(0), dAlpha(0), dInvK(0),
      dCovGrad(0), logDetK(0.0), lastJitter(0.0), needInverse(false)
{
  if(Xin->getRows() != nois->getNumData())
    throw ndlexceptions::MatrixError("CGp: X and the targets disagree on the number of data");   // CGp.cpp:60
  if(approxType != FTC)
    throw ndlexceptions::NotImplementedError("sparse approximations (DTC/FITC/PITC/DTCVAR) are outside the accelerated FTC path");
  setVerbosity(verbos);
  const char* e = std::getenv("GPC_EXACT_TRANS");
  if(e && e[0] == '1') refTransRounding = false;
}
CGp::CGp()
    : pX(0), py(0), pkern(0), pnoise(0), ownsKernNoise(true), fileNumData(0), fileInputDim(0), numActive(0), scale(1, 1, 1.0),
      bias(1, 1, 0.0), refTransRounding(true), MupToDate(false), KupToDate(false), AlphaUpToDate(false),
      invKupToDate(false), LcholRounded(false), dX(0), dM(0), dL(0), dInvKm(0), dAlpha(0), dInvK(0), dCovGrad(0), logDetK(0.0), lastJitter(0.0),
      needInverse(false)
{
  const char* e = std::getenv("GPC_EXACT_TRANS");
  if(e && e[0] == '1') refTransRounding = false;
}
void CGp::setData(CMatrix* Xin, CMatrix* yin)
{
  if(Xin->getRows() != yin->getRows()) throw ndlexceptions::MatrixError("CGp: X and the targets disagree on the number of data");
  if(yin->getCols() != scale.getCols()) throw ndlexceptions::MatrixError("CGp: targets do not have the model's output dimension");
  pX = Xin;
  py = yin;
  pnoise->py = yin;
  devFree(dX);   // re-staged from the new X on the next updateK
  devFree(dM);
  devFree(dL);
  devFree(dInvKm);
  devFree(dAlpha);
  devFree(dInvK);
  devFree(dCovGrad);
  MupToDate = KupToDate = AlphaUpToDate = invKupToDate = false;
}
CGp::~CGp()
{
  if(ownsKernNoise) {
    delete pkern;
    delete pnoise;
  }
  devFree(dX);
  devFree(dM);
  devFree(dL);
  devFree(dInvKm);
  devFree(dAlpha);
  devFree(dInvK);
  devFree(dCovGrad);
}

void CGp::updateM() const
{
  if(MupToDate) return;
  const unsigned int N = getNumData(), d = getOutputDim();
  m.resize(N, d);
  for(unsigned int j = 0; j < d; j++)
    for(unsigned int i = 0; i < N; i++) m.setVal((py->getVal(i, j) - bias.getVal(j)) * (1 / scale.getVal(j)), i, j);
  if(!dM) dM = devAlloc((size_t)N * d);
  gpcCheck(gpc_memcpy_h2d(dM, m.getVals(), sizeof(double) * (size_t)N * d, 0));
  MupToDate = true;
  KupToDate = false;   // quad / invKm depend on m
  AlphaUpToDate = false;
}
void CGp::ensureDeviceInputs() const
{
  if(!dX) {
    const size_t n = (size_t)pX->getRows() * pX->getCols();
    dX = devAlloc(n);
    gpcCheck(gpc_memcpy_h2d(dX, pX->getVals(), sizeof(double) * n, 0));
  }
  if(!MupToDate) updateM();
}

void CGp::updateK() const
{
  if(KupToDate && (invKupToDate || !needInverse)) return;
  ensureDeviceInputs();
  const int64_t N = getNumData(), D = getInputDim(), d = getOutputDim();
  if(KupToDate && needInverse && !LcholRounded) {
    // The factor of the current parameters is still exact in dL (SCG asks for the gradient at the point whose
    // objective it has just evaluated): only the inverse is missing -- no second Gram build + factorisation.
    if(!dInvK) dInvK = devAlloc((size_t)N * N);
    gpcCheck(gpc_memcpy_d2d(dInvK, dL, sizeof(double) * (size_t)N * N, 0));
    gpcCheck(gpc_potri_f64('L', N, dInvK, N, 0));
    invKupToDate = true;
    return;
  }
  if(!dL) dL = devAlloc((size_t)N * N);
  if(!dInvKm) dInvKm = devAlloc((size_t)N * d);
  gpc_kspec ks;
  pkern->toKspec(ks);
  double jit = 0.0;
  int info = 0;
  // _updateK + jitChol + logDet (CGp.cpp:698-712, 881-887) in one call: Gram, in-place lower Cholesky, log|K|
  gpcCheck(gpc_gp_update_k_f64(&ks, dX, N, D, N, dL, N, &logDetK, &jit, &info, 0));
  lastJitter = jit;
  if(info != 0) throw ndlexceptions::MatrixNonPosDef();
  if(jit > 1e-2 && getVerbosity() > 2)
    std::cout << "Warning: jitter of " << jit << " added to K in _updateInvK()." << std::endl;
  // invK * m without forming invK (the reference uses dsymv on the explicit inverse, CGp.cpp:928)
  gpcCheck(gpc_gp_alpha_f64(N, d, dL, N, dM, N, dInvKm, N, 0));
  quad.assign((size_t)d, 0.0);
  gpcCheck(gpc_coldot_f64(N, d, dM, N, dInvKm, N, &quad[0], 0));
  if(needInverse) {
    // invK.pdinv(LcholK) (CGp.cpp:889): only the gradient needs the explicit inverse
    if(!dInvK) dInvK = devAlloc((size_t)N * N);
    gpcCheck(gpc_memcpy_d2d(dInvK, dL, sizeof(double) * (size_t)N * N, 0));
    gpcCheck(gpc_potri_f64('L', N, dInvK, N, 0));
    invKupToDate = true;
  } else {
    invKupToDate = false;
  }
  // LcholK.trans() (CGp.cpp:890) of a reference built from ndlfortran.f leaves the strictly-lower part of LcholK in
  // single precision.  Only Alpha and the predictions read LcholK, so that rounding is applied lazily, by
  // updateAlpha(): during optimisation (likelihood and gradient only) the factor stays exact and reusable.
  LcholRounded = false;
  KupToDate = true;
  AlphaUpToDate = false;
}

void CGp::updateAlpha() const
{
  if(AlphaUpToDate && KupToDate) return;
  updateM();
  updateK();
  const int64_t N = getNumData(), d = getOutputDim();
  if(!dAlpha) dAlpha = devAlloc((size_t)N * d);
  if(refTransRounding && !LcholRounded) {
    gpcCheck(gpc_ref_trans_rounding_f64(N, dL, N, 0));
    LcholRounded = true;
  }
  if(refTransRounding)
    gpcCheck(gpc_gp_alpha_f64(N, d, dL, N, dM, N, dAlpha, N, 0));   // Alpha.trsm(LcholK ...) twice, CGp.cpp:481-483
  else
    gpcCheck(gpc_memcpy_d2d(dAlpha, dInvKm, sizeof(double) * (size_t)N * d, 0));
  AlphaUpToDate = true;
}

double CGp::logLikelihood() const
{
  updateM();
  needInverse = false;   // the likelihood alone never needs invK; a gradient at the same point adds it from the factor
  updateK();
  double L = 0.0;
  for(unsigned int j = 0; j < getOutputDim(); j++) {   // CGp.cpp:923-932
    L += quad[j];
    L += logDetK;
  }
  L *= -0.5;
  L += pkern->priorLogProb();
  L -= (double)getOutputDim() * (double)getNumData() * HALFLOGTWOPI;
  return L;
}

double CGp::logLikelihoodGradient(CMatrix& g) const
{
  // updateG (CGp.cpp:1080-1117): for every output, covGrad = -0.5 (invK - invKm invKm') and the kernel's
  // getGradTransParams against it, accumulated; then the log-likelihood itself.
  if(!MupToDate) updateM();
  needInverse = true;
  updateK();
  const int64_t N = getNumData(), D = getInputDim();
  const unsigned int np = pkern->getNumParams();
  if(g.getRows() != 1 || g.getCols() != np) throw ndlexceptions::MatrixError("logLikelihoodGradient: g must be 1 x nParams");
  if(!dCovGrad) dCovGrad = devAlloc((size_t)N * N);
  gpc_kspec ks;
  pkern->toKspec(ks);
  std::vector<double> acc(np, 0.0), tmp(np > 0 ? np : 1, 0.0);
  for(unsigned int j = 0; j < getOutputDim(); j++) {
    gpcCheck(gpc_covgrad_f64(N, dInvK, N, dInvKm + (size_t)j * N, dCovGrad, N, 0));   // updateCovGradient, CGp.cpp:666-679
    gpcCheck(gpc_kern_grad_f64(&ks, dX, N, D, N, dCovGrad, N, &tmp[0], 0));
    for(unsigned int i = 0; i < np; i++) acc[i] += tmp[i];
  }
  // chain rule into the optimiser space (CKern::getGradTransParams, CKern.cpp:50-63); linear in g, so once at the end
  for(unsigned int t = 0; t < pkern->getNumTransforms(); t++) {
    const unsigned int idx = pkern->getTransformIndex(t);
    acc[idx] *= pkern->getTransformGradFact(pkern->getParam(idx), t);
  }
  for(unsigned int i = 0; i < np; i++) g.setVal(acc[i], 0, i);
  return logLikelihood();
}

void CGp::getOptParams(CMatrix& param) const
{
  CMatrix tp(1, pkern->getNumParams());
  pkern->getTransParams(tp);
  for(unsigned int i = 0; i < pkern->getNumParams(); i++) param.setVal(tp.getVal(i), i);
}
void CGp::setOptParams(const CMatrix& param)
{
  KupToDate = false;   // CGp.cpp:389
  invKupToDate = false;
  AlphaUpToDate = false;
  CMatrix tp(1, pkern->getNumParams());
  for(unsigned int i = 0; i < pkern->getNumParams(); i++) tp.setVal(param.getVal(i), i);
  pkern->setTransParams(tp);
}

void CGp::posteriorMeanVar(CMatrix& mu, CMatrix& varSigma, const CMatrix& Xin) const
{
  const int64_t N = getNumData(), D = getInputDim(), d = getOutputDim(), Ns = Xin.getRows();
  if(mu.getCols() != d || varSigma.getCols() != d || mu.getRows() != Ns || varSigma.getRows() != Ns)
    throw ndlexceptions::MatrixError("posteriorMeanVar: output dimensions");   // CGp.cpp:644-647
  if(Xin.getCols() != D) throw ndlexceptions::MatrixError("posteriorMeanVar: input dimension");
  updateAlpha();
  gpc_kspec ks;
  pkern->toKspec(ks);
  double* dXs = devAlloc((size_t)Ns * D);
  double* dKx = devAlloc((size_t)N * Ns);
  double* dMu = devAlloc((size_t)Ns * d);
  double* dVar = devAlloc((size_t)Ns);
  std::vector<double> hmu((size_t)Ns * d), hvar((size_t)Ns);
  try {
    gpcCheck(gpc_memcpy_h2d(dXs, Xin.getVals(), sizeof(double) * (size_t)Ns * D, 0));
    gpcCheck(gpc_gp_posterior_f64(&ks, dX, N, D, N, dL, N, dAlpha, N, d, dXs, Ns, Ns, dKx, N, dMu, Ns, dVar, 0));
    gpcCheck(gpc_memcpy_d2h(&hmu[0], dMu, sizeof(double) * hmu.size(), 0));
    gpcCheck(gpc_memcpy_d2h(&hvar[0], dVar, sizeof(double) * hvar.size(), 0));
  } catch(...) {
    devFree(dXs); devFree(dKx); devFree(dMu); devFree(dVar);
    throw;
  }
  devFree(dXs); devFree(dKx); devFree(dMu); devFree(dVar);
  for(int64_t i = 0; i < Ns; i++) {
    if(!(hvar[i] >= 0.0)) throw ndlexceptions::Error("posterior variance is negative");   // CHECKZEROORPOSITIVE, CGp.cpp:607
    for(int64_t j = 0; j < d; j++) {
      double muv = hmu[i + j * Ns], vs = hvar[i];
      const double sc = scale.getVal((unsigned int)j), bi = bias.getVal((unsigned int)j);
      if(sc != 1.0) { muv *= sc; vs *= sc * sc; }   // CGp.cpp:561-573, 614-624
      if(bi != 0.0) muv += bi;
      mu.setVal(muv, (unsigned int)i, (unsigned int)j);
      varSigma.setVal(vs, (unsigned int)i, (unsigned int)j);
    }
  }
}
void CGp::out(CMatrix& yPred, const CMatrix& Xin) const
{
  CMatrix muTest(yPred.getRows(), yPred.getCols()), varSigmaTest(yPred.getRows(), yPred.getCols());
  posteriorMeanVar(muTest, varSigmaTest, Xin);
  pnoise->out(yPred, muTest, varSigmaTest);
}
void CGp::out(CMatrix& yPred, CMatrix& probPred, const CMatrix& Xin) const
{
  CMatrix muTest(yPred.getRows(), yPred.getCols()), varSigmaTest(yPred.getRows(), yPred.getCols());
  posteriorMeanVar(muTest, varSigmaTest, Xin);
  pnoise->out(yPred, probPred, muTest, varSigmaTest);
}

void CGp::optimise(unsigned int iters)
{
  if(getVerbosity() > 2) {
    std::cout << "Initial model:" << std::endl;
    display(std::cout);
  }
  if(getVerbosity() > 2 && getOptNumParams() < 40) checkGradients();
  setMaxIters(iters);
  runDefaultOptimiser();
  if(getVerbosity() > 1) std::cout << "... done. " << std::endl;
  if(getVerbosity() > 0) display(std::cout);
}
void CGp::display(std::ostream& os) const
{
  os << "Standard GP Model: " << std::endl;
  os << "Optimiser: " << getDefaultOptimiserStr() << std::endl;
  os << "Data Set Size: " << getNumData() << std::endl;
  os << "Kernel Type: " << std::endl;
  os << "Scales learnt: " << isOutputScaleLearnt() << std::endl;
  os << "X learnt: " << isOptimiseX() << std::endl;
  os << "Bias: " << bias << std::endl;
  os << "Scale: " << scale << std::endl;
  pnoise->display(os);
  pkern->display(os);
  if(py && pX) os << "Log likelihood: " << logLikelihood() << std::endl;   // a model read from a file has no data yet
}

void CGp::writeParamsToStream(std::ostream& out) const
{
  out << "baseType=dataModel" << std::endl << "type=gp" << std::endl;
  out << "numData=" << getNumData() << std::endl << "outputDim=" << getOutputDim() << std::endl;
  out << "inputDim=" << getInputDim() << std::endl;
  out << "sparseApproximation=" << getApproximationType() << std::endl;
  out << "numActive=" << 4294967295u << std::endl;   // (unsigned)-1 for FTC, as the reference writes it (gp.cpp:352)
  out << "learnScale=0" << std::endl << "learnBias=0" << std::endl;
  out << "version=0.200000" << std::endl;
  scale.writeParamsToStream(out);
  out << "version=0.200000" << std::endl;
  bias.writeParamsToStream(out);
  pkern->toStream(out);
  out << "version=0.200000" << std::endl;
  pnoise->writeParamsToStream(out);
}
void CGp::readParamsFromStream(std::istream& in)
{
  // CGp::readParamsFromStream, CGp.cpp:1606-1650 (FTC models only)
  const std::string base = ndlstream::readField(in, "baseType");
  if(base != "dataModel") throw ndlexceptions::StreamFormatError("baseType", "Error mismatch between saved base type, " + base + ", and Class base type, dataModel.");
  const std::string type = ndlstream::readField(in, "type");
  if(type != "gp") throw ndlexceptions::StreamFormatError("type", "Error mismatch between saved type, " + type + ", and Class type, gp.");
  fileNumData = (unsigned int)ndlstream::readInt(in, "numData");
  const unsigned int outDim = (unsigned int)ndlstream::readInt(in, "outputDim");
  fileInputDim = (unsigned int)ndlstream::readInt(in, "inputDim");
  const long approx = ndlstream::readInt(in, "sparseApproximation");
  if(approx != FTC) throw ndlexceptions::NotImplementedError("sparse approximations (DTC/FITC/PITC/DTCVAR) are outside the accelerated FTC path");
  numActive = (unsigned int)std::strtoul(ndlstream::readField(in, "numActive").c_str(), 0, 10);
  if(ndlstream::readBool(in, "learnScale")) throw ndlexceptions::NotImplementedError("learnt output scales are outside the accelerated FTC path");
  (void)ndlstream::readBool(in, "learnBias");
  scale.fromStream(in);
  bias.fromStream(in);
  if(scale.getCols() != outDim || bias.getCols() != outDim) throw ndlexceptions::StreamFormatError("outputDim", "scale / bias do not match the output dimension");
  if(ownsKernNoise) {
    delete pkern;
    delete pnoise;
  }
  pkern = 0;
  pnoise = 0;
  ownsKernNoise = true;
  pkern = readKernFromStream(in);
  pnoise = readNoiseFromStream(in);
  if(pkern->getInputDim() != fileInputDim) throw ndlexceptions::StreamFormatError("inputDim", "kernel input dimension does not match the model's");
  MupToDate = KupToDate = AlphaUpToDate = invKupToDate = false;
}
void CGp::fromStream(std::istream& in)
{
  ndlstream::readVersion(in);
  readParamsFromStream(in);
}
void CGp::toStream(std::ostream& out) const
{
  out << "version=0.200000" << std::endl;
  writeParamsToStream(out);
}
void CGp::toFile(const std::string fileName, const std::string comment) const
{
  std::ofstream out(fileName.c_str());
  if(!out) throw ndlexceptions::FileWriteError(fileName);
  if(comment.size() > 0) out << "# " << comment << std::endl;
  toStream(out);
}
void writeGpToStream(const CGp& model, std::ostream& out) { model.toStream(out); }
void writeGpToFile(const CGp& model, const std::string modelFileName, const std::string comment)
{
  model.toFile(modelFileName, comment);
}
CGp* readGpFromStream(std::istream& in)
{
  CGp* pmodel = new CGp();
  try {
    pmodel->fromStream(in);
  } catch(...) {
    delete pmodel;
    throw;
  }
  return pmodel;
}
CGp* readGpFromFile(const std::string modelFileName, int verbosity)
{
  if(verbosity > 0) std::cout << "Loading model file." << std::endl;
  std::ifstream in(modelFileName.c_str());
  if(!in.is_open()) throw ndlexceptions::FileReadError(modelFileName);
  CGp* pmodel = 0;
  try {
    pmodel = readGpFromStream(in);
  } catch(ndlexceptions::StreamFormatError& err) {
    throw ndlexceptions::FileFormatError(modelFileName, err.getMessage());
  }
  pmodel->setVerbosity(verbosity);
  if(verbosity > 0) std::cout << "... done." << std::endl;
  return pmodel;
}
