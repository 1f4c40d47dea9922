// CGplvm.cpp -- see CGplvm.h.  Host side: parameter vector, PCA start, dirty flag; everything O(N^2) and up is a call
// into libgpc_hip.so.
#include <sstream>
#include <fstream>
#include <cstdlib>
#include "CGplvm.h"
#include "ndlstream.h"
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <algorithm>
#include <vector>
#include <time.h>
#include "gpc_hip.h"

namespace {
// GPC_GPLVM_TIMING=1: host wall time per library call of an evaluation, printed when the process ends (measurement aid)
struct CallClock {
  enum { NSLOT = 12 };
  std::vector<double> samples[NSLOT];
  const char* name[NSLOT];
  bool on;
  CallClock() : on(false)
  {
    for(int i = 0; i < NSLOT; i++) name[i] = 0;
    const char* e = getenv("GPC_GPLVM_TIMING");
    on = e && atoi(e) != 0;
  }
  ~CallClock()
  {
    if(!on) return;
    for(int i = 0; i < NSLOT; i++)
      if(name[i] && !samples[i].empty()) {
        std::vector<double>& v = samples[i];
        const double first = v[0];
        std::sort(v.begin(), v.end());
        std::cerr << "  [gplvm timing] " << name[i] << ": " << v.size() << " calls, median " << 1e6 * v[v.size() / 2] << " us, min "
                  << 1e6 * v[0] << ", max " << 1e6 * v.back() << ", first " << 1e6 * first << std::endl;
      }
  }
};
CallClock g_clock;
struct Timed {
  int slot;
  struct timespec t0;
  Timed(int s, const char* n) : slot(s)
  {
    if(g_clock.on) { g_clock.name[s] = n; clock_gettime(CLOCK_MONOTONIC, &t0); }
  }
  ~Timed()
  {
    if(!g_clock.on) return;
    struct timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    g_clock.samples[slot].push_back((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec));
  }
};

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

// Cyclic Jacobi eigen-decomposition of a small symmetric matrix (d x d, d = number of outputs): the reference calls
// LAPACK dsyev here (CGplvm.cpp:170-172); the matrix is tiny and this stays on the host.
// a (column-major, destroyed) -> eigenvalues w (ascending), eigenvectors as columns of v.
void jacobiEig(std::vector<double>& a, int n, std::vector<double>& w, std::vector<double>& v)
{
  v.assign((size_t)n * n, 0.0);
  for(int i = 0; i < n; i++) v[i + (size_t)i * n] = 1.0;
  for(int sweep = 0; sweep < 100; sweep++) {
    double off = 0.0, diag = 0.0;
    for(int j = 0; j < n; j++)
      for(int i = 0; i < n; i++) (i == j ? diag : off) += a[i + (size_t)j * n] * a[i + (size_t)j * n];
    if(off <= 1e-32 * diag || off == 0.0) break;
    for(int p = 0; p < n - 1; p++)
      for(int q = p + 1; q < n; q++) {
        const double apq = a[p + (size_t)q * n];
        if(apq == 0.0) continue;
        const double theta = (a[q + (size_t)q * n] - a[p + (size_t)p * n]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for(int k = 0; k < n; k++) {   // A := A J
          const double akp = a[k + (size_t)p * n], akq = a[k + (size_t)q * n];
          a[k + (size_t)p * n] = c * akp - s * akq;
          a[k + (size_t)q * n] = s * akp + c * akq;
        }
        for(int k = 0; k < n; k++) {   // A := J' A
          const double apk = a[p + (size_t)k * n], aqk = a[q + (size_t)k * n];
          a[p + (size_t)k * n] = c * apk - s * aqk;
          a[q + (size_t)k * n] = s * apk + c * aqk;
        }
        for(int k = 0; k < n; k++) {
          const double vkp = v[k + (size_t)p * n], vkq = v[k + (size_t)q * n];
          v[k + (size_t)p * n] = c * vkp - s * vkq;
          v[k + (size_t)q * n] = s * vkp + c * vkq;
        }
      }
  }
  w.resize(n);
  std::vector<int> order(n);
  for(int i = 0; i < n; i++) order[i] = i;
  for(int i = 0; i < n; i++)   // selection sort, ascending like dsyev
    for(int j = i + 1; j < n; j++)
      if(a[order[j] + (size_t)order[j] * n] < a[order[i] + (size_t)order[i] * n]) std::swap(order[i], order[j]);
  std::vector<double> vs((size_t)n * n);
  for(int i = 0; i < n; i++) {
    w[i] = a[order[i] + (size_t)order[i] * n];
    for(int k = 0; k < n; k++) vs[k + (size_t)i * n] = v[k + (size_t)order[i] * n];
  }
  v.swap(vs);
}
}  // namespace

CGplvm::CGplvm(CKern* kernel, CScaleNoise* nois, int latDim, int verbos)
    : pX(new CMatrix()), pkern(kernel), pnoise(nois), pYown(0), latentDim((unsigned int)latDim), dataDim(nois->getOutputDim()),
      numData(nois->getNumData()), regulariseLatent(true), KupToDate(false), dX(0), dM(0), dK(0), dL(0), dA(0), dG(0), dGX(0),
      logDetK(0.0)
{
  setVerbosity(verbos);
  pX->resize(numData, latentDim);
  pnoise->computeM(m);   // initVals -> updateSites for every point, CGplvm.cpp:137-142
  initXpca();
}

CGplvm::CGplvm()
    : pX(new CMatrix()), pkern(0), pnoise(0), pYown(0), latentDim(0), dataDim(0), numData(0), regulariseLatent(true),
      KupToDate(false), dX(0), dM(0), dK(0), dL(0), dA(0), dG(0), dGX(0), logDetK(0.0)
{
}

CGplvm::~CGplvm()
{
  releaseDevice();
  delete pX;
  if(pYown) {
    delete pkern;
    delete pnoise;
    delete pYown;
  }
}

void CGplvm::releaseDevice()
{
  devFree(dX);
  devFree(dM);
  devFree(dK);
  devFree(dL);
  devFree(dA);
  devFree(dG);
  devFree(dGX);
}

void CGplvm::initXpca()
{
  // CGplvm.cpp:162-191: covm = m'm/N - ymean ymean' ; eigen-decomposition ; X = m U_q diag(lambda_q)^-1/2 ; centre X.
  const unsigned int N = numData, d = dataDim, q = latentDim;
  if(q > d) throw ndlexceptions::MatrixError("latent dimension exceeds data dimension");
  std::vector<double> mean(d, 0.0), cov((size_t)d * d, 0.0), w, U;
  for(unsigned int j = 0; j < d; j++) {
    double s = 0.0;
    for(unsigned int i = 0; i < N; i++) s += m.getVal(i, j);
    mean[j] = s / (double)N;
  }
  for(unsigned int a = 0; a < d; a++)
    for(unsigned int b = a; b < d; b++) {
      double s = 0.0;
      for(unsigned int i = 0; i < N; i++) s += m.getVal(i, a) * m.getVal(i, b);
      const double c = s / (double)N - mean[a] * mean[b];
      cov[a + (size_t)b * d] = c;
      cov[b + (size_t)a * d] = c;
    }
  jacobiEig(cov, (int)d, w, U);
  for(unsigned int c = 0; c < q; c++) {
    const unsigned int e = d - 1 - c;   // largest eigenvalues first
    // LAPACK leaves the sign of an eigenvector unspecified; pin it (largest-magnitude component positive).  The
    // objective is invariant to the sign of a latent column, so this only selects one of the mirror-image solutions.
    unsigned int big = 0;
    for(unsigned int k = 1; k < d; k++)
      if(std::fabs(U[k + (size_t)e * d]) > std::fabs(U[big + (size_t)e * d])) big = k;
    const double sgn = (U[big + (size_t)e * d] < 0.0 ? -1.0 : 1.0) / std::sqrt(w[e]);
    double colMean = 0.0;
    for(unsigned int i = 0; i < N; i++) {
      double s = 0.0;
      for(unsigned int k = 0; k < d; k++) s += m.getVal(i, k) * U[k + (size_t)e * d];
      pX->setVal(s * sgn, i, c);
      colMean += s * sgn;
    }
    colMean /= (double)N;
    for(unsigned int i = 0; i < N; i++) pX->setVal(pX->getVal(i, c) - colMean, i, c);
  }
  updateX();
}

void CGplvm::getOptParams(CMatrix& param) const
{
  CMatrix tp(1, pkern->getNumParams());
  pkern->getTransParams(tp);
  unsigned int counter = 0;
  for(unsigned int i = 0; i < pkern->getNumParams(); i++) param.setVal(tp.getVal(i), counter++);
  for(unsigned int j = 0; j < latentDim; j++)
    for(unsigned int i = 0; i < numData; i++) param.setVal(pX->getVal(i, j), counter++);
}

void CGplvm::setOptParams(const CMatrix& param)
{
  KupToDate = false;
  CMatrix tp(1, pkern->getNumParams());
  unsigned int counter = 0;
  for(unsigned int i = 0; i < pkern->getNumParams(); i++) tp.setVal(param.getVal(counter++), i);
  pkern->setTransParams(tp);
  for(unsigned int j = 0; j < latentDim; j++)
    for(unsigned int i = 0; i < numData; i++) pX->setVal(param.getVal(counter++), i, j);
  updateX();
}

namespace {
// gpc_defer(1) for the calls inside the scope (include/gpc_hip.h); exception-safe
struct DeferScope {
  DeferScope() { (void)gpc_defer(1); }
  ~DeferScope() { (void)gpc_defer(0); }
};
// Spans from a deferred call to the flush that delivers its outputs; declared AFTER the outputs' storage.  If the scope is
// left by an exception before done(), the postponed deliveries are dropped while their destinations still exist (they would
// otherwise be written -- stack and heap that are gone -- by this thread's next synchronising call).
struct PendingGuard {
  bool armed;
  PendingGuard() : armed(true) {}
  void done() { armed = false; }
  ~PendingGuard() { if(armed) (void)gpc_discard_pending(); }
};
}  // namespace

void CGplvm::updateK() const
{
  if(KupToDate) return;
  const int64_t N = numData, d = dataDim, q = latentDim;
  if(!dX) dX = devAlloc((size_t)N * q);
  if(!dM) {
    dM = devAlloc((size_t)N * d);
    gpcCheck(gpc_memcpy_h2d(dM, m.getVals(), sizeof(double) * (size_t)N * d, 0));
  }
  if(!dK) dK = devAlloc((size_t)N * N);
  if(!dA) dA = devAlloc((size_t)N * d);
  {
    Timed t(0, "h2d X");
    DeferScope defer;     // (no wait of its own: the bytes are staged at once, include/gpc_hip.h)
    gpcCheck(gpc_memcpy_h2d(dX, pX->getVals(), sizeof(double) * (size_t)N * q, 0));
  }
  gpc_kspec ks;
  pkern->toKspec(ks);
  if(!dL) dL = devAlloc((size_t)N * N);
  { Timed t(1, "gram"); gpcCheck(gpc_gram_sym_f64(&ks, dX, N, q, N, dL, N, 0)); }                // _updateK, CGplvm.cpp:418-432
  int info = 0;
  PendingGuard pending;      // (info and logDetK are written by the flush below)
  // LcholK.chol(), logDet(LcholK), invK.pdinv(LcholK) (CGplvm.cpp:441-444) in one pass: dL <- L, dK <- invK
  // (gpc_defer: the factorisation's info and log-determinant come back in the column dots' synchronisation two launches
  //  further on instead of one of their own -- the host issues the product and the dots while the device still factors)
  {
    Timed t(2, "chol_inverse");
    DeferScope defer;
    gpcCheck(gpc_chol_inverse_f64(N, dL, N, dK, N, &logDetK, &info, 0));
  }
  { Timed t(3, "gemm invK m"); gpcCheck(gpc_gemm_f64('N', 'N', N, d, N, 1.0, dK, N, dM, N, 0.0, dA, N, 0)); }  // invK * m, column by column in 503 / 374
  quad.assign((size_t)d, 0.0);
  { Timed t(4, "coldot"); gpcCheck(gpc_coldot_f64(N, d, dA, N, dM, N, &quad[0], 0)); }
  gpcCheck(gpc_sync_pending(0));     // (nothing left unless the dots had no rows)
  pending.done();
  if(info != 0) throw ndlexceptions::MatrixNonPosDef();
  KupToDate = true;
}

double CGplvm::logLikelihood() const
{
  updateK();
  double L = 0.0;
  for(unsigned int j = 0; j < dataDim; j++) {   // CGplvm.cpp:498-507
    L += quad[j];
    L += logDetK;
  }
  if(regulariseLatent)
    for(unsigned int j = 0; j < latentDim; j++) L += pX->norm2Col(j);   // CGplvm.cpp:533-540
  L *= -0.5;
  L += pkern->priorLogProb();
  return L;
}

double CGplvm::logLikelihoodGradient(CMatrix& g) const
{
  const int64_t N = numData, d = dataDim, q = latentDim;
  const unsigned int nk = pkern->getNumParams();
  if(g.getRows() != 1 || g.getCols() != getOptNumParams())
    throw ndlexceptions::MatrixError("logLikelihoodGradient: g must be 1 x nParams");
  updateK();
  if(!dG) dG = devAlloc((size_t)N * N);
  if(!dGX) dGX = devAlloc((size_t)N * q);
  gpc_kspec ks;
  pkern->toKspec(ks);
  // sum over the outputs of updateCovGradient (CGplvm.cpp:365-378); both passes below are linear in covGrad
  { Timed t(5, "covgrad_multi"); gpcCheck(gpc_covgrad_multi_f64(N, d, dK, N, dA, N, dG, N, 0)); }
  std::vector<double> gk(nk > 0 ? nk : 1, 0.0);
  std::vector<double> gx((size_t)N * q);
  PendingGuard pending;      // (gx is written by the flush below)
  // dL/dX first and its copy to the host postponed (gpc_defer), the parameter sums second: ONE wait for both instead of two
  { Timed t(7, "kern_gradx"); gpcCheck(gpc_kern_gradx_f64(&ks, dX, N, q, N, dG, N, dGX, N, 0)); }      // getGradX + dotColCol loop, 573-604
  {
    Timed t(8, "d2h gx");
    DeferScope defer;
    gpcCheck(gpc_memcpy_d2h(&gx[0], dGX, sizeof(double) * gx.size(), 0));
  }
  { Timed t(6, "kern_grad"); gpcCheck(gpc_kern_grad_f64(&ks, dX, N, q, N, dG, N, &gk[0], 0)); }      // getGradTransParams, CGplvm.cpp:589-596
  gpcCheck(gpc_sync_pending(0));     // (gx is there: kern_grad's wait brought it; this only covers a pass that did not wait)
  pending.done();
  for(unsigned int t = 0; t < pkern->getNumTransforms(); t++) {
    const unsigned int idx = pkern->getTransformIndex(t);
    gk[idx] *= pkern->getTransformGradFact(pkern->getParam(idx), t);
  }
  for(unsigned int i = 0; i < nk; i++) g.setVal(gk[i], 0, i);
  for(int64_t k = 0; k < q; k++)
    for(int64_t i = 0; i < N; i++) {
      double v = gx[i + k * N];
      if(regulariseLatent) v += -pX->getVal((unsigned int)i, (unsigned int)k);   // CGplvm.cpp:676-686
      g.setVal(v, 0, nk + (unsigned int)(i + N * k));
    }
  return logLikelihood();
}

void CGplvm::optimise(const int iters)
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

void CGplvm::display(std::ostream& os) const
{
  os << "GPLVM Model: " << std::endl;
  os << "Data Set Size: " << numData << std::endl;
  os << "Kernel Type: " << std::endl;
  os << "Latent space regularised: " << isLatentRegularised() << std::endl;
  os << "Dynamics learnt: " << isDynamicModelLearnt() << std::endl;
  os << "Scales learnt: " << isInputScaleLearnt() << std::endl;
  pnoise->display(os);
  pkern->display(os);
}

void CGplvm::writeParamsToStream(std::ostream& out) const
{
  out << "baseType=dataModel" << std::endl << "type=gplvm" << std::endl;
  out << "numData=" << getNumData() << std::endl;
  out << "outputDim=" << getNumProcesses() << std::endl;
  out << "inputDim=" << getLatentDim() << std::endl;
  out << "latentRegularised=" << isLatentRegularised() << std::endl;
  out << "backConstrained=0" << std::endl << "dynamicsLearnt=0" << std::endl;
  pkern->toStream(out);
  out << "version=0.200000" << std::endl;
  pnoise->writeParamsToStream(out);
  out << "Y:" << getNumProcesses() << ",X:" << getLatentDim();
  if(isLabels()) out << ",labels:1";
  out << std::endl;
  for(unsigned int i = 0; i < numData; i++) {
    for(unsigned int j = 0; j < dataDim; j++) out << pnoise->getTarget(i, j) << " ";
    for(unsigned int j = 0; j < latentDim; j++) out << pX->getVal(i, j) << " ";
    if(isLabels()) out << labels[i];
    out << std::endl;
  }
}
void CGplvm::toStream(std::ostream& out) const
{
  out << "version=0.200000" << std::endl;
  writeParamsToStream(out);
}
// CGplvm::readParamsFromStream (CGplvm.cpp:802-899): header fields, kernel, noise, then one row per data point
// "Y:<d>,X:<q>[,labels:1]" = d targets, q latent coordinates and the optional integer label.
void CGplvm::readParamsFromStream(std::istream& in)
{
  using namespace ndlstream;
  const std::string tbase = readField(in, "baseType");
  if(tbase != "dataModel")
    throw ndlexceptions::StreamFormatError("baseType", "Error mismatch between saved base type, " + tbase + ", and Class base type, dataModel.");
  const std::string ttype = readField(in, "type");
  if(ttype != "gplvm")
    throw ndlexceptions::StreamFormatError("type", "Error mismatch between saved type, " + ttype + ", and Class type, gplvm.");
  numData = (unsigned int)readInt(in, "numData");
  dataDim = (unsigned int)readInt(in, "outputDim");
  latentDim = (unsigned int)readInt(in, "inputDim");
  regulariseLatent = readBool(in, "latentRegularised");
  if(readBool(in, "backConstrained"))
    throw ndlexceptions::NotImplementedError("back-constrained GP-LVM models are outside the accelerated path");
  if(readBool(in, "dynamicsLearnt"))
    throw ndlexceptions::NotImplementedError("GP-LVM models with dynamics are outside the accelerated path");
  pkern = readKernFromStream(in);
  // the noise block: CScaleNoise, parameters [bias..., scale...]
  readVersion(in);
  if(readField(in, "baseType") != "noise") throw ndlexceptions::StreamFormatError("baseType", "noise block expected");
  const std::string ntype = readField(in, "type");
  if(ntype != "scale") throw ndlexceptions::StreamFormatError("type", "Noise type " + ntype + " is outside the GP-LVM path (scale)");
  const unsigned int outDim = (unsigned int)readInt(in, "outputDim"), numPar = (unsigned int)readInt(in, "numParams");
  if(outDim != dataDim || numPar != 2 * outDim)
    throw ndlexceptions::StreamFormatError("numParams", "Number of parameters in file does not match computed number.");
  CMatrix par(1, numPar);
  par.fromStream(in);
  // data rows
  std::string line;
  if(!ndlstream::getline(in, line)) throw ndlexceptions::StreamFormatError("Y", "data header missing");
  bool labelsPresent = false;
  {
    std::stringstream hs(line);
    std::string tok;
    while(std::getline(hs, tok, ',')) {
      const size_t c = tok.find(':');
      if(c == std::string::npos) throw ndlexceptions::StreamFormatError("Y", "bad data header: " + line);
      const std::string key = tok.substr(0, c);
      const long val = std::atol(tok.substr(c + 1).c_str());
      if(key == "Y" && (unsigned int)val != dataDim) throw ndlexceptions::StreamFormatError("Y", "data dimension mismatch");
      if(key == "X" && (unsigned int)val != latentDim) throw ndlexceptions::StreamFormatError("X", "latent dimension mismatch");
      if(key == "labels") labelsPresent = (val != 0);
    }
  }
  pYown = new CMatrix(numData, dataDim);
  pX->resize(numData, latentDim);
  labels.clear();
  for(unsigned int i = 0; i < numData; i++) {
    if(!ndlstream::getline(in, line)) throw ndlexceptions::StreamFormatError("Y", "Incorrect number of data rows.");
    std::istringstream ss(line);
    std::string tok;
    for(unsigned int j = 0; j < dataDim; j++) {
      if(!(ss >> tok)) throw ndlexceptions::StreamFormatError("Y", "short data row");
      pYown->setVal(std::strtod(tok.c_str(), 0), i, j);
    }
    for(unsigned int j = 0; j < latentDim; j++) {
      if(!(ss >> tok)) throw ndlexceptions::StreamFormatError("X", "short data row");
      pX->setVal(std::strtod(tok.c_str(), 0), i, j);
    }
    if(labelsPresent) {
      if(!(ss >> tok)) throw ndlexceptions::StreamFormatError("labels", "short data row");
      labels.push_back(std::atoi(tok.c_str()));
    }
  }
  pnoise = new CScaleNoise(pYown);
  pnoise->setParams(par);
  pnoise->computeM(m);
  releaseDevice();
  KupToDate = false;
}
void CGplvm::fromStream(std::istream& in)
{
  ndlstream::readVersion(in);
  readParamsFromStream(in);
}
CGplvm* readGplvmFromStream(std::istream& in)
{
  CGplvm* pmodel = new CGplvm();
  try {
    pmodel->fromStream(in);
  } catch(...) {
    delete pmodel;
    throw;
  }
  return pmodel;
}
CGplvm* readGplvmFromFile(const std::string modelFileName, int verbosity)
{
  if(verbosity > 0) std::cout << "Loading model file." << std::endl;
  std::ifstream in(modelFileName.c_str());
  if(!in.is_open()) throw ndlexceptions::FileReadError(modelFileName);
  CGplvm* pmodel = readGplvmFromStream(in);
  if(verbosity > 0) std::cout << "... done." << std::endl;
  return pmodel;
}

void writeGplvmToStream(const CGplvm& model, std::ostream& out) { model.toStream(out); }
void writeGplvmToFile(const CGplvm& model, const std::string modelFileName, const std::string comment)
{
  if(model.getVerbosity() > 0) std::cout << "Saving model file." << std::endl;
  std::ofstream out(modelFileName.c_str());
  if(!out) throw ndlexceptions::FileWriteError(modelFileName);
  out << std::setprecision(17);
  if(comment.size() > 0) out << "# " << comment << std::endl;
  writeGplvmToStream(model, out);
}
