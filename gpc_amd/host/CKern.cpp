// CKern.cpp -- see CKern.h.
#include "CKern.h"
#include "ndlstream.h"
#include <cmath>
#include <cstring>

namespace {
struct DevIn {   // read-only device view of a host or device matrix
  const double* p;
  double* owned;
  explicit DevIn(const CMatrix& M) : p(0), owned(0)
  {
    if(M.isOnDevice()) {
      p = M.devPtr();
    } else {
      void* d = 0;
      const size_t bytes = sizeof(double) * M.getNumElements();
      gpcCheck(gpc_malloc(&d, bytes ? bytes : 8));
      owned = static_cast<double*>(d);
      if(bytes) gpcCheck(gpc_memcpy_h2d(owned, M.getVals(), bytes, 0));
      p = owned;
    }
  }
  ~DevIn() { if(owned) (void)gpc_free(owned); }
};
struct DevOut {  // writable device view; copied back to a host matrix on destruction
  double* p;
  CMatrix* M;
  bool staged;
  explicit DevOut(CMatrix& m) : p(0), M(&m), staged(false)
  {
    if(m.isOnDevice()) {
      p = m.devPtr();
    } else {
      void* d = 0;
      const size_t bytes = sizeof(double) * m.getNumElements();
      gpcCheck(gpc_malloc(&d, bytes ? bytes : 8));
      p = static_cast<double*>(d);
      staged = true;
    }
  }
  ~DevOut()
  {
    if(staged) {
      const size_t bytes = sizeof(double) * M->getNumElements();
      if(bytes) (void)gpc_memcpy_d2h(M->getVals(), p, bytes, 0);
      (void)gpc_free(p);
    }
  }
};
inline int64_t ld(const CMatrix& M) { return M.getRows() > 0 ? (int64_t)M.getRows() : 1; }

void pushTerm(gpc_kspec& ks, int type, const double* params, int np)
{
  if(ks.n_terms >= GPC_MAX_TERMS || ks.offs[ks.n_terms] + np > GPC_MAX_PARAMS)
    throw ndlexceptions::Error("kernel spec exceeds libgpc_hip limits (terms or parameters)");
  const int off = ks.offs[ks.n_terms];
  ks.types[ks.n_terms] = type;
  for(int i = 0; i < np; i++) ks.params[off + i] = params[i];
  ks.n_terms++;
  ks.offs[ks.n_terms] = off + np;
}
}  // namespace

void CKern::toKspec(gpc_kspec& ks) const
{
  std::memset(&ks, 0, sizeof(ks));
  appendKspec(ks);
}

void CKern::compute(CMatrix& K, const CMatrix& X) const
{
  if(!K.rowsMatch(X) || !K.isSquare()) throw ndlexceptions::MatrixError("compute: K must be N x N");
  gpc_kspec ks;
  toKspec(ks);
  {
    DevIn x(X);
    DevOut k(K);
    gpcCheck(gpc_gram_sym_f64(&ks, x.p, X.getRows(), X.getCols(), ld(X), k.p, ld(K), 0));
    gpcCheck(gpc_stream_sync(0));
  }
  K.setSymmetric(true);
}
void CKern::compute(CMatrix& K, const CMatrix& X, const CMatrix& X2) const
{
  if(!K.rowsMatch(X) || K.getCols() != X2.getRows()) throw ndlexceptions::MatrixError("compute: K must be N x N2");
  gpc_kspec ks;
  toKspec(ks);
  DevIn x(X), x2(X2);
  DevOut k(K);
  gpcCheck(gpc_gram_cross_f64(&ks, x.p, X.getRows(), ld(X), x2.p, X2.getRows(), ld(X2), X.getCols(), k.p, ld(K), 0));
  gpcCheck(gpc_stream_sync(0));
}
namespace {
// the rows `idx` of X as a matrix of their own (BOUNDCHECK of the reference: an index past the data throws)
void gatherRows(CMatrix& out, const CMatrix& X, const std::vector<unsigned int>& idx)
{
  out.resize((unsigned int)idx.size(), X.getCols());
  for(unsigned int i = 0; i < idx.size(); i++) {
    if(idx[i] >= X.getRows()) throw ndlexceptions::MatrixError("compute: row index outside the data");
    out.copyRowRow(i, X, idx[i]);
  }
}
}  // namespace

void CKern::compute(CMatrix& K, const CMatrix& X1, const std::vector<unsigned int> indices1, const CMatrix& X2,
                    const std::vector<unsigned int> indices2) const
{
  if(K.getRows() != indices1.size() || K.getCols() != indices2.size())
    throw ndlexceptions::MatrixError("compute: K must be |indices1| x |indices2|");
  if(indices1.empty() || indices2.empty()) return;
  CMatrix A, B;
  gatherRows(A, X1, indices1);
  gatherRows(B, X2, indices2);
  compute(K, A, B);
}
void CKern::compute(CMatrix& K, const CMatrix& X, const std::vector<unsigned int> indices) const
{
  if(K.getRows() != indices.size() || !K.isSquare()) throw ndlexceptions::MatrixError("compute: K must be |indices| x |indices|");
  if(indices.empty()) return;
  const bool wasSymmetric = K.isSymmetric();   // the reference's overload leaves the flag alone (CKern.h:111-126)
  CMatrix A;
  gatherRows(A, X, indices);
  compute(K, A);
  K.setSymmetric(wasSymmetric);
}
void CKern::compute(CMatrix& K, const CMatrix& X, const CMatrix& X2, unsigned int row) const
{
  if(!K.rowsMatch(X) || K.getCols() != 1) throw ndlexceptions::MatrixError("compute: K must be N x 1");
  const std::vector<unsigned int> one(1, row);
  CMatrix B;
  gatherRows(B, X2, one);
  compute(K, X, B);
}
void CKern::diagCompute(CMatrix& d, const CMatrix& X) const
{
  if(!X.rowsMatch(d) || d.getCols() != 1) throw ndlexceptions::MatrixError("diagCompute: d must be N x 1");
  gpc_kspec ks;
  toKspec(ks);
  DevIn x(X);
  DevOut dd(d);
  gpcCheck(gpc_gram_diag_f64(&ks, x.p, X.getRows(), X.getCols(), ld(X), dd.p, 0));
  gpcCheck(gpc_stream_sync(0));
}
void CKern::getGradParams(CMatrix& g, const CMatrix& X, const CMatrix& covGrad, bool) const
{
  if(g.getRows() != 1 || g.getCols() != nParams) throw ndlexceptions::MatrixError("getGradParams: g must be 1 x nParams");
  if(!X.rowsMatch(covGrad) || !covGrad.isSquare()) throw ndlexceptions::MatrixError("getGradParams: covGrad must be N x N");
  gpc_kspec ks;
  toKspec(ks);
  std::vector<double> out(nParams > 0 ? nParams : 1);
  DevIn x(X), cg(covGrad);
  gpcCheck(gpc_kern_grad_f64(&ks, x.p, X.getRows(), X.getCols(), ld(X), cg.p, ld(covGrad), &out[0], 0));
  for(unsigned int i = 0; i < nParams; i++) g.setVal(out[i], 0, i);
}
void CKern::getGradTransParams(CMatrix& g, const CMatrix& X, const CMatrix& covGrad, bool regularise) const
{
  // CKern::getGradTransParams, CKern.cpp:50-63
  getGradParams(g, X, covGrad, regularise);
  for(unsigned int i = 0; i < getNumTransforms(); i++) {
    const unsigned int idx = getTransformIndex(i);
    g.setVal(g.getVal(idx) * getTransformGradFact(getParam(idx), i), idx);
  }
}
void CKern::writeParamsToStream(std::ostream& out) const
{
  // CKern::writeParamsToStream, CKern.cpp:15-26
  out << "baseType=" << getBaseType() << std::endl << "type=" << getType() << std::endl;
  out << "inputDim=" << getInputDim() << std::endl << "numParams=" << getNumParams() << std::endl;
  CMatrix par(1, getNumParams());
  getParams(par);
  out << "version=0.200000" << std::endl;
  par.writeParamsToStream(out);
  out << "numPriors=0" << std::endl;
}
std::ostream& CKern::display(std::ostream& os) const
{
  os << getName() << " kernel:" << std::endl;
  for(unsigned int i = 0; i < nParams; i++) os << getParamName(i) << ": " << getParam(i) << std::endl;
  return os;
}

// ---- rbf ----------------------------------------------------------------------------------------------------------------
void CRbfKern::_init()
{
  nParams = 2;
  setType("rbf");
  setName("RBF");
  setParamName("inverseWidth", 0);
  addTransform(CTransform::defaultPositive(), 0);
  setParamName("variance", 1);
  addTransform(CTransform::defaultPositive(), 1);
  stationary = true;
  inverseWidth = 1.0;
  variance = 1.0;
}
double CRbfKern::computeElement(const CMatrix& X1, unsigned int i1, const CMatrix& X2, unsigned int i2) const
{
  return variance * std::exp(-0.5 * inverseWidth * X1.dist2Row(i1, X2, i2));   // CKern.cpp:1147-1154
}
void CRbfKern::setParam(double val, unsigned int i)
{
  if(i == 0) inverseWidth = val;
  else if(i == 1) variance = val;
  else throw ndlexceptions::Error("Requested parameter doesn't exist.");
}
double CRbfKern::getParam(unsigned int i) const
{
  if(i == 0) return inverseWidth;
  if(i == 1) return variance;
  throw ndlexceptions::Error("Requested parameter doesn't exist.");
}
void CRbfKern::appendKspec(gpc_kspec& ks) const
{
  const double p[2] = {inverseWidth, variance};
  pushTerm(ks, GPC_KERN_RBF, p, 2);
}

// ---- rbfard ---------------------------------------------------------------------------------------------------------------
void CRbfardKern::_init()
{
  nParams = 2;
  setType("rbfard");
  setName("RBF ARD");
  setParamName("inverseWidth", 0);
  addTransform(CTransform::defaultPositive(), 0);
  setParamName("variance", 1);
  addTransform(CTransform::defaultPositive(), 1);
  stationary = true;
  inverseWidth = 1.0;
  variance = 1.0;
}
void CRbfardKern::setInitParam()
{
  // CRbfardKern::setInitParam, CKern.cpp:3199-3219: scales 0.5, sigmoid transforms on them
  nParams = 2 + getInputDim();
  inverseWidth = 1.0;
  variance = 1.0;
  scales.assign(getInputDim(), 0.5);
  while(getNumTransforms() > 2) {   // setInputDim may be called again: rebuild the scale transforms
    delete transforms.back();
    transforms.pop_back();
    transIndex.pop_back();
  }
  for(unsigned int i = 2; i < nParams; i++) {
    setParamName("inputScale", i);
    addTransform(CTransform::defaultZeroOne(), i);
  }
}
double CRbfardKern::computeElement(const CMatrix& X1, unsigned int i1, const CMatrix& X2, unsigned int i2) const
{
  double val = 0.0;   // CKern.cpp:3305-3316
  for(unsigned int k = 0; k < getInputDim(); k++) {
    const double x = X1.getVal(i1, k) - X2.getVal(i2, k);
    val += x * scales[k] * x;
  }
  return variance * std::exp(-val * inverseWidth * 0.5);
}
void CRbfardKern::setParam(double val, unsigned int i)
{
  if(i == 0) inverseWidth = val;
  else if(i == 1) variance = val;
  else if(i < nParams) scales[i - 2] = val;
  else throw ndlexceptions::Error("Requested parameter doesn't exist.");
}
double CRbfardKern::getParam(unsigned int i) const
{
  if(i == 0) return inverseWidth;
  if(i == 1) return variance;
  if(i < nParams) return scales[i - 2];
  throw ndlexceptions::Error("Requested parameter doesn't exist.");
}
void CRbfardKern::appendKspec(gpc_kspec& ks) const
{
  std::vector<double> p(2 + scales.size());
  p[0] = inverseWidth;
  p[1] = variance;
  for(size_t k = 0; k < scales.size(); k++) p[2 + k] = scales[k];
  pushTerm(ks, GPC_KERN_RBFARD, &p[0], (int)p.size());
}

// ---- white / bias / lin ---------------------------------------------------------------------------------------------------
void CWhiteKern::_init()
{
  nParams = 1;
  setType("white");
  setName("white noise");
  setParamName("variance", 0);
  addTransform(CTransform::defaultPositive(), 0);
  stationary = true;
  variance = std::exp(-2.0);
}
void CWhiteKern::setInitParam() { variance = std::exp(-2.0); }   // CKern.cpp:641-644
void CWhiteKern::setParam(double val, unsigned int i)
{
  if(i != 0) throw ndlexceptions::Error("Requested parameter doesn't exist.");
  variance = val;
}
double CWhiteKern::getParam(unsigned int i) const
{
  if(i != 0) throw ndlexceptions::Error("Requested parameter doesn't exist.");
  return variance;
}
void CWhiteKern::appendKspec(gpc_kspec& ks) const { pushTerm(ks, GPC_KERN_WHITE, &variance, 1); }

void CBiasKern::_init()
{
  nParams = 1;
  setType("bias");
  setName("bias");
  setParamName("variance", 0);
  addTransform(CTransform::defaultPositive(), 0);
  stationary = true;
  variance = std::exp(-2.0);
}
void CBiasKern::setInitParam() { variance = std::exp(-2.0); }   // CKern.cpp:928-931
void CBiasKern::setParam(double val, unsigned int i)
{
  if(i != 0) throw ndlexceptions::Error("Requested parameter doesn't exist.");
  variance = val;
}
double CBiasKern::getParam(unsigned int i) const
{
  if(i != 0) throw ndlexceptions::Error("Requested parameter doesn't exist.");
  return variance;
}
void CBiasKern::appendKspec(gpc_kspec& ks) const { pushTerm(ks, GPC_KERN_BIAS, &variance, 1); }

void CLinKern::_init()
{
  nParams = 1;
  setType("lin");
  setName("linear");
  setParamName("variance", 0);
  addTransform(CTransform::defaultPositive(), 0);
  stationary = false;
  variance = 1.0;
}
void CLinKern::setParam(double val, unsigned int i)
{
  if(i != 0) throw ndlexceptions::Error("Requested parameter doesn't exist.");
  variance = val;
}
double CLinKern::getParam(unsigned int i) const
{
  if(i != 0) throw ndlexceptions::Error("Requested parameter doesn't exist.");
  return variance;
}
void CLinKern::appendKspec(gpc_kspec& ks) const { pushTerm(ks, GPC_KERN_LIN, &variance, 1); }

// ---- compound ---------------------------------------------------------------------------------------------------------------
void CCmpndKern::_init()
{
  nParams = 0;
  setType("cmpnd");
  setName("compound");
  stationary = true;
}
CCmpndKern::CCmpndKern(const CCmpndKern& k) : CKern()
{
  _init();
  setInputDim(k.getInputDim());
  for(size_t i = 0; i < k.components.size(); i++) addKern(k.components[i]);
}
CCmpndKern::~CCmpndKern()
{
  for(size_t i = 0; i < components.size(); i++) delete components[i];   // CKern.cpp:150-154
}
unsigned int CCmpndKern::addKern(const CKern* kern)
{
  // CKern.h:382-392: clone, append, re-index the clone's transforms into the concatenated parameter vector
  CKern* c = kern->clone();
  components.push_back(c);
  const unsigned int oldN = nParams;
  nParams += c->getNumParams();
  for(unsigned int i = 0; i < c->getNumTransforms(); i++)
    addTransform(new CTransform(c->getTransform(i)->kind), c->getTransformIndex(i) + oldN);
  if(!c->isStationary()) stationary = false;
  return (unsigned int)components.size() - 1;
}
double CCmpndKern::diagComputeElement(const CMatrix& X, unsigned int index) const
{
  double v = 0.0;   // CKern.cpp:165-171
  for(size_t i = 0; i < components.size(); i++) v += components[i]->diagComputeElement(X, index);
  return v;
}
double CCmpndKern::computeElement(const CMatrix& X1, unsigned int i1, const CMatrix& X2, unsigned int i2) const
{
  double v = 0.0;   // CKern.cpp:219-226
  for(size_t i = 0; i < components.size(); i++) v += components[i]->computeElement(X1, i1, X2, i2);
  return v;
}
double CCmpndKern::getVariance() const
{
  double v = 0.0;
  for(size_t i = 0; i < components.size(); i++) v += components[i]->getVariance();
  return v;
}
double CCmpndKern::getWhite() const
{
  double v = 0.0;
  for(size_t i = 0; i < components.size(); i++) v += components[i]->getWhite();
  return v;
}
void CCmpndKern::setParam(double val, unsigned int paramNo)
{
  unsigned int start = 0;   // CKern.h:392-406
  for(size_t i = 0; i < components.size(); i++) {
    const unsigned int n = components[i]->getNumParams();
    if(paramNo < start + n) {
      components[i]->setParam(val, paramNo - start);
      return;
    }
    start += n;
  }
  throw ndlexceptions::Error("Requested parameter doesn't exist.");
}
double CCmpndKern::getParam(unsigned int paramNo) const
{
  unsigned int start = 0;
  for(size_t i = 0; i < components.size(); i++) {
    const unsigned int n = components[i]->getNumParams();
    if(paramNo < start + n) return components[i]->getParam(paramNo - start);
    start += n;
  }
  throw ndlexceptions::Error("Requested parameter doesn't exist.");
}
std::string CCmpndKern::getParamName(unsigned int paramNo) const
{
  unsigned int start = 0;
  for(size_t i = 0; i < components.size(); i++) {
    const unsigned int n = components[i]->getNumParams();
    if(paramNo < start + n) return components[i]->getType() + components[i]->getParamName(paramNo - start);
    start += n;
  }
  throw ndlexceptions::Error("Requested parameter doesn't exist.");
}
void CCmpndKern::appendKspec(gpc_kspec& ks) const
{
  for(size_t i = 0; i < components.size(); i++) components[i]->appendKspec(ks);
}
void CCmpndKern::writeParamsToStream(std::ostream& out) const
{
  // CComponentKern::writeParamsToStream, CKern.cpp:113-124
  out << "baseType=" << getBaseType() << std::endl << "type=" << getType() << std::endl;
  out << "inputDim=" << getInputDim() << std::endl << "numParams=" << getNumParams() << std::endl;
  out << "numKerns=" << components.size() << std::endl;
  for(size_t i = 0; i < components.size(); i++) components[i]->toStream(out);
}
std::ostream& CCmpndKern::display(std::ostream& os) const
{
  // the reference has no override: CKern::display lists the concatenated parameters under their compound names
  // ("rbfinverseWidth: ...", CKern.cpp:66-74, CKern.h:402-414)
  return CKern::display(os);
}

// ---- model-file reader (reference CKern.cpp:100-112, 125-140, 4192-4260) ----------------------------------------------
void CKern::readParamsFromStream(std::istream& in)
{
  setInputDim((unsigned int)ndlstream::readInt(in, "inputDim"));
  const unsigned int nPars = (unsigned int)ndlstream::readInt(in, "numParams");
  CMatrix par(1, nPars);
  par.fromStream(in);
  if(nPars != getNumParams())
    throw ndlexceptions::StreamFormatError("numParams", "Listed number of parameters does not match computed number of parameters.");
  setParams(par);
  const long numPriors = ndlstream::readInt(in, "numPriors");
  if(numPriors != 0) throw ndlexceptions::NotImplementedError("parameter priors (CDist) are outside the accelerated path");
}
void CCmpndKern::readParamsFromStream(std::istream& in)
{
  setInputDim((unsigned int)ndlstream::readInt(in, "inputDim"));
  (void)ndlstream::readInt(in, "numParams");
  const unsigned int numKerns = (unsigned int)ndlstream::readInt(in, "numKerns");
  for(unsigned int i = 0; i < numKerns; i++) {
    CKern* k = readKernFromStream(in);
    addKern(k);   // clones
    delete k;
  }
}
CKern* readKernFromStream(std::istream& in)
{
  ndlstream::readVersion(in);
  const std::string base = ndlstream::readField(in, "baseType");
  if(base != "kern")
    throw ndlexceptions::StreamFormatError("baseType", "Error mismatch between saved base type, " + base + ", and Class base type, kern.");
  const std::string type = ndlstream::readField(in, "type");
  CKern* k = 0;
  if(type == "white") k = new CWhiteKern(1u);
  else if(type == "bias") k = new CBiasKern(1u);
  else if(type == "rbf") k = new CRbfKern(1u);
  else if(type == "lin") k = new CLinKern(1u);
  else if(type == "rbfard") k = new CRbfardKern(1u);
  else if(type == "cmpnd") k = new CCmpndKern();
  else throw ndlexceptions::StreamFormatError("type", "Kernel type " + type + " is outside the accelerated set (rbf, rbfard, lin, bias, white, cmpnd)");
  try {
    k->readParamsFromStream(in);
  } catch(...) {
    delete k;
    throw;
  }
  return k;
}
