// CMatrix.cpp -- see CMatrix.h.  Host-side bookkeeping plus calls into libgpc_hip.so (include/gpc_hip.h); no LAPACK,
// no CPU fallback for the factorisation / solve members.
#include "CMatrix.h"
#include "ndlstream.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include "gpc_hip.h"

void gpcCheck(int rc)
{
  if(rc == GPC_OK) return;
  const char* msg = gpc_last_error();
  if(rc == GPC_EINVAL) throw ndlexceptions::MatrixError(std::string("libgpc_hip: ") + (msg ? msg : "invalid argument"));
  throw ndlexceptions::DeviceError(msg ? msg : "unknown failure");
}

namespace {
// RAII device view of a matrix: DEVICE matrices are used in place, HOST matrices are staged (and copied back if rw).
struct DevView {
  double* p;
  CMatrix* owner;
  bool staged, writeback;
  size_t bytes;
  DevView(const CMatrix& M, bool rw) : p(0), owner(const_cast<CMatrix*>(&M)), staged(false), writeback(rw), bytes(0)
  {
    bytes = sizeof(double) * M.getNumElements();
    if(M.isOnDevice()) {
      p = owner->devPtr();
    } else {
      void* d = 0;
      gpcCheck(gpc_malloc(&d, bytes ? bytes : 8));
      p = static_cast<double*>(d);
      staged = true;
      if(bytes) gpcCheck(gpc_memcpy_h2d(p, M.getVals(), bytes, 0));
    }
  }
  ~DevView()
  {
    if(staged) {
      if(writeback && bytes) (void)gpc_memcpy_d2h(owner->getVals(), p, bytes, 0);
      (void)gpc_free(p);
    }
  }
};
inline size_t ldOf(const CMatrix& M) { return M.getRows() > 0 ? M.getRows() : 1; }
}  // namespace

// ---- storage ---------------------------------------------------------------------------------------------------------
void CMatrix::alloc(size_t rows, size_t cols, Residence r)
{
  nrows = rows;
  ncols = cols;
  where = r;
  dev = 0;
  host.clear();
  if(r == HOST) {
    host.assign(rows * cols, 0.0);
  } else {
    void* d = 0;
    gpcCheck(gpc_malloc(&d, sizeof(double) * (rows * cols ? rows * cols : 1)));
    dev = static_cast<double*>(d);
  }
}
void CMatrix::release()
{
  if(dev) (void)gpc_free(dev);
  dev = 0;
  host.clear();
}
CMatrix::CMatrix() : nrows(0), ncols(0), where(HOST), dev(0), symmetric(false), triangular(false) { alloc(1, 1, HOST); }
CMatrix::CMatrix(double val) : dev(0), symmetric(false), triangular(false)
{
  alloc(1, 1, HOST);
  host[0] = val;
}
CMatrix::CMatrix(unsigned int r, unsigned int c) : dev(0), symmetric(false), triangular(false) { alloc(r, c, HOST); }
CMatrix::CMatrix(unsigned int r, unsigned int c, double val) : dev(0), symmetric(false), triangular(false)
{
  alloc(r, c, HOST);
  setVals(val);
}
CMatrix::CMatrix(unsigned int r, unsigned int c, const double* in) : dev(0), symmetric(false), triangular(false)
{
  alloc(r, c, HOST);
  if(in) std::memcpy(&host[0], in, sizeof(double) * host.size());
}
CMatrix::CMatrix(unsigned int r, unsigned int c, Residence w) : dev(0), symmetric(false), triangular(false)
{
  alloc(r, c, w);
}
CMatrix::CMatrix(const CMatrix& A) : dev(0), symmetric(false), triangular(false)
{
  alloc(A.nrows, A.ncols, A.where);
  deepCopy(A);
}
CMatrix& CMatrix::operator=(const CMatrix& A)
{
  if(this != &A) deepCopy(A);   // a real copy: the reference's default (shallow) operator= is a latent double free
  return *this;
}
CMatrix::~CMatrix() { release(); }

void CMatrix::resize(unsigned int rows, unsigned int cols)
{
  if(rows == nrows && cols == ncols) return;
  const Residence w = where;
  release();
  alloc(rows, cols, w);
}
void CMatrix::deepCopy(const CMatrix& A)
{
  if(nrows != A.nrows || ncols != A.ncols) {
    const Residence w = where;
    release();
    alloc(A.nrows, A.ncols, w);
  }
  const size_t bytes = sizeof(double) * nrows * ncols;
  if(bytes) {
    if(where == HOST && A.where == HOST) std::memcpy(&host[0], &A.host[0], bytes);
    else if(where == DEVICE && A.where == DEVICE) gpcCheck(gpc_memcpy_d2d(dev, A.dev, bytes, 0));
    else if(where == DEVICE) gpcCheck(gpc_memcpy_h2d(dev, &A.host[0], bytes, 0));
    else gpcCheck(gpc_memcpy_d2h(&host[0], A.dev, bytes, 0));
  }
  symmetric = A.symmetric;
  triangular = A.triangular;
}
void CMatrix::toDevice()
{
  if(where == DEVICE) return;
  void* d = 0;
  const size_t bytes = sizeof(double) * nrows * ncols;
  gpcCheck(gpc_malloc(&d, bytes ? bytes : 8));
  if(bytes) gpcCheck(gpc_memcpy_h2d(d, &host[0], bytes, 0));
  dev = static_cast<double*>(d);
  host.clear();
  where = DEVICE;
}
void CMatrix::toHost()
{
  if(where == HOST) return;
  host.assign(nrows * ncols, 0.0);
  if(!host.empty()) gpcCheck(gpc_memcpy_d2h(&host[0], dev, sizeof(double) * host.size(), 0));
  (void)gpc_free(dev);
  dev = 0;
  where = HOST;
}
double* CMatrix::getVals()
{
  if(where != HOST) throw ndlexceptions::MatrixError("getVals() on a device-resident matrix");
  return host.empty() ? 0 : &host[0];
}
const double* CMatrix::getVals() const
{
  if(where != HOST) throw ndlexceptions::MatrixError("getVals() on a device-resident matrix");
  return host.empty() ? 0 : &host[0];
}

// ---- element access (bounds-checked like the reference's BOUNDCHECK, CMatrix.h:255-269) -------------------------------
double CMatrix::getVal(unsigned int i, unsigned int j) const
{
  if(i >= nrows || j >= ncols) throw ndlexceptions::MatrixError("getVal: index out of bounds");
  if(where == HOST) return host[i + nrows * j];
  double v = 0.0;
  gpcCheck(gpc_memcpy_d2h(&v, dev + i + nrows * j, sizeof(double), 0));
  return v;
}
double CMatrix::getVal(unsigned int i) const
{
  if(i >= nrows * ncols) throw ndlexceptions::MatrixError("getVal: index out of bounds");
  if(where == HOST) return host[i];
  double v = 0.0;
  gpcCheck(gpc_memcpy_d2h(&v, dev + i, sizeof(double), 0));
  return v;
}
void CMatrix::setVal(double val, unsigned int i, unsigned int j)
{
  if(i >= nrows || j >= ncols) throw ndlexceptions::MatrixError("setVal: index out of bounds");
  if(where == HOST) host[i + nrows * j] = val;
  else gpcCheck(gpc_memcpy_h2d(dev + i + nrows * j, &val, sizeof(double), 0));
}
void CMatrix::setVal(double val, unsigned int i)
{
  if(i >= nrows * ncols) throw ndlexceptions::MatrixError("setVal: index out of bounds");
  if(where == HOST) host[i] = val;
  else gpcCheck(gpc_memcpy_h2d(dev + i, &val, sizeof(double), 0));
}
void CMatrix::setVals(double val)
{
  if(where == HOST) {
    for(size_t i = 0; i < host.size(); i++) host[i] = val;
  } else if(val == 0.0) {
    gpcCheck(gpc_memset(dev, 0, sizeof(double) * nrows * ncols, 0));
  } else {
    std::vector<double> tmp(nrows * ncols, val);
    if(!tmp.empty()) gpcCheck(gpc_memcpy_h2d(dev, &tmp[0], sizeof(double) * tmp.size(), 0));
  }
}

// ---- host helpers ------------------------------------------------------------------------------------------------------
#define HOSTONLY(name) \
  if(where != HOST) throw ndlexceptions::MatrixError(std::string(name) + ": matrix is device resident")

void CMatrix::scale(double alpha)
{
  HOSTONLY("scale");
  for(size_t i = 0; i < host.size(); i++) host[i] *= alpha;
}
void CMatrix::scaleCol(unsigned int j, double alpha)
{
  HOSTONLY("scaleCol");
  for(size_t i = 0; i < nrows; i++) host[i + nrows * j] *= alpha;
}
void CMatrix::axpy(const CMatrix& x, double alpha)
{
  HOSTONLY("axpy");
  if(!dimensionsMatch(x) || x.where != HOST) throw ndlexceptions::MatrixError("axpy: dimension mismatch");
  for(size_t i = 0; i < host.size(); i++) host[i] += alpha * x.host[i];
}
void CMatrix::addCol(unsigned int j, double c)
{
  HOSTONLY("addCol");
  for(size_t i = 0; i < nrows; i++) host[i + nrows * j] += c;
}
void CMatrix::addDiag(double c)
{
  if(!isSquare()) throw ndlexceptions::MatrixError("addDiag: matrix is not square");
  if(where == HOST) {
    for(size_t i = 0; i < nrows; i++) host[i + nrows * i] += c;
  } else {
    gpcCheck(gpc_add_diag_f64((int64_t)nrows, dev, (int64_t)ldOf(*this), c, 0));
  }
}
void CMatrix::copyRowRow(unsigned int i, const CMatrix& X, unsigned int k)
{
  HOSTONLY("copyRowRow");
  for(size_t j = 0; j < ncols; j++) host[i + nrows * j] = X.getVal(k, (unsigned int)j);
}
void CMatrix::copyColCol(unsigned int j, const CMatrix& X, unsigned int k)
{
  HOSTONLY("copyColCol");
  for(size_t i = 0; i < nrows; i++) host[i + nrows * j] = X.getVal((unsigned int)i, k);
}
double CMatrix::normRow(unsigned int i) const { return std::sqrt(norm2Row(i)); }
double CMatrix::norm2Row(unsigned int i) const
{
  HOSTONLY("norm2Row");
  double s = 0.0;
  for(size_t j = 0; j < ncols; j++) s += host[i + nrows * j] * host[i + nrows * j];
  return s;
}
double CMatrix::norm2Col(unsigned int j) const
{
  HOSTONLY("norm2Col");
  double s = 0.0;
  for(size_t i = 0; i < nrows; i++) s += host[i + nrows * j] * host[i + nrows * j];
  return s;
}
double CMatrix::dotRowRow(unsigned int i, const CMatrix& A, unsigned int k) const
{
  HOSTONLY("dotRowRow");
  double s = 0.0;
  for(size_t j = 0; j < ncols; j++) s += host[i + nrows * j] * A.getVal(k, (unsigned int)j);
  return s;
}
double CMatrix::dotColCol(unsigned int j, const CMatrix& A, unsigned int k) const
{
  HOSTONLY("dotColCol");
  double s = 0.0;
  for(size_t i = 0; i < nrows; i++) s += host[i + nrows * j] * A.getVal((unsigned int)i, k);
  return s;
}
double CMatrix::dist2Row(unsigned int i, const CMatrix& A, unsigned int k) const
{
  return norm2Row(i) + A.norm2Row(k) - 2.0 * dotRowRow(i, A, k);   // CMatrix.h:554-560
}
double CMatrix::sum() const
{
  HOSTONLY("sum");
  double s = 0.0;
  for(size_t i = 0; i < host.size(); i++) s += host[i];
  return s;
}
double CMatrix::trace() const
{
  if(!isSquare()) throw ndlexceptions::MatrixError("trace: matrix is not square");
  if(where == HOST) {
    double s = 0.0;
    for(size_t i = 0; i < nrows; i++) s += host[i + nrows * i];
    return s;
  }
  double t = 0.0;
  gpcCheck(gpc_trace_f64((int64_t)nrows, dev, (int64_t)ldOf(*this), &t, 0));
  return t;
}
double CMatrix::max() const
{
  // The reference's loop body is unbraced (CMatrix.cpp:568-577): only the LAST element is ever compared with the
  // first.  SCG's convergence test depends on it (COptimisable.cpp:385), so it is reproduced, not fixed.
  HOSTONLY("max");
  double mx = host[0];
  const double val = host.size() > 1 ? host[host.size() - 1] : 0.0;
  if(val > mx) mx = val;
  return mx;
}
double CMatrix::maxAbsDiff(const CMatrix& X) const
{
  HOSTONLY("maxAbsDiff");
  double m = 0.0;
  for(size_t i = 0; i < host.size(); i++) m = std::fmax(m, std::fabs(host[i] - X.getVal((unsigned int)i)));
  return m;
}
bool CMatrix::equals(const CMatrix& A, double tol) const
{
  if(!dimensionsMatch(A)) return false;
  return maxAbsDiff(A) <= tol;
}
void CMatrix::maxRow(CMatrix& m) const
{
  // NB: as in the reference (CMatrix.cpp:735-764) maxRow returns the column MINIMA and minRow the maxima.
  for(unsigned int j = 0; j < getCols(); j++) {
    m.setVal(getVal(0, j), j);
    for(unsigned int i = 1; i < getRows(); i++)
      if(getVal(i, j) < m.getVal(0, j)) m.setVal(getVal(i, j), 0, j);
  }
}
void CMatrix::minRow(CMatrix& m) const
{
  for(unsigned int j = 0; j < getCols(); j++) {
    m.setVal(getVal(0, j), j);
    for(unsigned int i = 1; i < getRows(); i++)
      if(getVal(i, j) > m.getVal(0, j)) m.setVal(getVal(i, j), 0, j);
  }
}
void CMatrix::getMatrix(CMatrix& out, unsigned int r0, unsigned int r1, unsigned int c0, unsigned int c1) const
{
  out.resize(r1 - r0 + 1, c1 - c0 + 1);
  for(unsigned int j = c0; j <= c1; j++)
    for(unsigned int i = r0; i <= r1; i++) out.setVal(getVal(i, j), i - r0, j - c0);
}
void CMatrix::setMatrix(unsigned int row, unsigned int col, const CMatrix& A)
{
  for(unsigned int j = 0; j < A.getCols(); j++)
    for(unsigned int i = 0; i < A.getRows(); i++) setVal(A.getVal(i, j), row + i, col + j);
}

// ---- LAPACK / BLAS-3 surface: libgpc_hip.so -------------------------------------------------------------------------------
void CMatrix::potrf(const char* type)
{
  if(!isSymmetric()) throw ndlexceptions::MatrixError("potrf: matrix is not flagged symmetric");   // CMatrix.cpp:373
  int info = 0;
  {
    DevView v(*this, true);
    gpcCheck(gpc_potrf_f64(type[0], (int64_t)nrows, v.p, (int64_t)ldOf(*this), &info, 0));
  }
  setSymmetric(false);
  setTriangular(true);
  if(info != 0) throw ndlexceptions::MatrixNonPosDef();
}
void CMatrix::chol(const char* type)
{
  if(!isSymmetric()) throw ndlexceptions::MatrixError("chol: matrix is not flagged symmetric");
  int info = 0;
  {
    DevView v(*this, true);
    gpcCheck(gpc_chol_f64(type[0], (int64_t)nrows, v.p, (int64_t)ldOf(*this), &info, 0));
  }
  setSymmetric(false);
  setTriangular(true);
  if(info != 0) throw ndlexceptions::MatrixNonPosDef();
}
double CMatrix::jitChol(CMatrix& A, unsigned int maxTries)
{
  // CMatrix.cpp:767-804, including the quirk that the value returned is the NEXT candidate jitter.
  if(!A.isSquare()) throw ndlexceptions::MatrixError("jitChol: matrix is not square");
  if(!A.isSymmetric()) throw ndlexceptions::MatrixError("jitChol: matrix is not flagged symmetric");
  double jitter = 1e-6 * A.trace() / (double)A.getRows();
  bool success = false;
  unsigned int tries = 0;
  while(!success && tries < maxTries) {
    try {
      deepCopy(A);
      chol();
      success = true;
    } catch(ndlexceptions::MatrixNonPosDef&) {
      A.addDiag(jitter);
      jitter *= 10;
      tries++;
      if(jitter > 10) throw ndlexceptions::MatrixNonPosDef();
    }
  }
  if(tries >= maxTries) {
    std::cout << "Adding jitter failed after " << tries << " tries." << std::endl;
    throw ndlexceptions::MatrixNonPosDef();
  }
  return jitter;
}
void CMatrix::potri(const char* type)
{
  if(!isSquare()) throw ndlexceptions::MatrixError("potri: matrix is not square");
  DevView v(*this, true);
  gpcCheck(gpc_potri_f64(type[0], (int64_t)nrows, v.p, (int64_t)ldOf(*this), 0));
}
void CMatrix::pdinv(const CMatrix& U)
{
  if(!U.isTriangular()) throw ndlexceptions::MatrixError("pdinv: U is not a Cholesky factor");   // CMatrix.cpp:423
  if(!isSymmetric()) throw ndlexceptions::MatrixError("pdinv: matrix is not flagged symmetric");
  deepCopy(U);
  potri("U");   // gpc_potri_f64 returns the full symmetric inverse: the reference's mirror loop is included
  setSymmetric(true);
  setTriangular(false);
}
void CMatrix::trans()
{
  if(isSquare()) {
    DevView v(*this, true);
    gpcCheck(gpc_transpose_inplace_f64((int64_t)nrows, v.p, (int64_t)ldOf(*this), 0));
    return;
  }
  HOSTONLY("trans (non-square)");
  std::vector<double> t(host.size());
  for(size_t j = 0; j < ncols; j++)
    for(size_t i = 0; i < nrows; i++) t[j + ncols * i] = host[i + nrows * j];
  host.swap(t);
  std::swap(nrows, ncols);
}
void CMatrix::trsm(const CMatrix& A, double alpha, const char* side, const char* type, const char* tr,
                   const char* diag)
{
  if(!A.isTriangular()) throw ndlexceptions::MatrixError("trsm: A is not flagged triangular");   // CMatrix.cpp:279
  const bool left = (side[0] == 'L' || side[0] == 'l');
  if((left ? nrows : ncols) != A.nrows || !A.isSquare()) throw ndlexceptions::MatrixError("trsm: dimension mismatch");
  DevView a(A, false), b(*this, true);
  gpcCheck(gpc_trsm_f64(side[0], type[0], tr[0], diag[0], (int64_t)nrows, (int64_t)ncols, alpha, a.p,
                        (int64_t)ldOf(A), b.p, (int64_t)ldOf(*this), 0));
}
void CMatrix::gemm(const CMatrix& A, const CMatrix& B, double alpha, double beta, const char* ta, const char* tb)
{
  const bool na = (ta[0] == 'n' || ta[0] == 'N'), nb = (tb[0] == 'n' || tb[0] == 'N');
  const size_t m = na ? A.nrows : A.ncols, k = na ? A.ncols : A.nrows;
  const size_t kb = nb ? B.nrows : B.ncols, n = nb ? B.ncols : B.nrows;
  if(m != nrows || n != ncols || k != kb) throw ndlexceptions::MatrixError("gemm: dimension mismatch");
  DevView a(A, false), b(B, false), c(*this, true);
  gpcCheck(gpc_gemm_f64(ta[0], tb[0], (int64_t)m, (int64_t)n, (int64_t)k, alpha, a.p, (int64_t)ldOf(A), b.p,
                        (int64_t)ldOf(B), beta, c.p, (int64_t)ldOf(*this), 0));
}
void CMatrix::syrk(const CMatrix& A, double alpha, double beta, const char* type, const char* tr)
{
  if(!(isSymmetric() || beta == 0.0)) throw ndlexceptions::MatrixError("syrk: matrix is not flagged symmetric");
  const bool nt = (tr[0] == 'n' || tr[0] == 'N');
  const size_t n = nt ? A.nrows : A.ncols, k = nt ? A.ncols : A.nrows;
  if(n != nrows || !isSquare()) throw ndlexceptions::MatrixError("syrk: dimension mismatch");
  DevView a(A, false), c(*this, true);
  gpcCheck(gpc_syrk_f64(type[0], tr[0], (int64_t)n, (int64_t)k, alpha, a.p, (int64_t)ldOf(A), beta, c.p,
                        (int64_t)ldOf(*this), 0));
  gpcCheck(gpc_symmetrize_f64(type[0], (int64_t)n, c.p, (int64_t)ldOf(*this), 0));   // copySymmetric(type)
  setSymmetric(true);
}
void CMatrix::copySymmetric(const char* type)
{
  DevView c(*this, true);
  gpcCheck(gpc_symmetrize_f64(type[0], (int64_t)nrows, c.p, (int64_t)ldOf(*this), 0));
}
void CMatrix::symv(const CMatrix& A, const CMatrix& x, double alpha, double beta, const char*)
{
  if(!A.isSymmetric()) throw ndlexceptions::MatrixError("symv: A is not flagged symmetric");
  if(A.nrows != nrows * ncols || x.nrows * x.ncols != A.nrows) throw ndlexceptions::MatrixError("symv: dimension mismatch");
  DevView a(A, false), xv(x, false), y(*this, true);
  gpcCheck(gpc_symv_f64((int64_t)A.nrows, alpha, a.p, (int64_t)ldOf(A), xv.p, beta, y.p, 0));
}
double logDet(const CMatrix& U)
{
  if(!U.isTriangular()) throw ndlexceptions::MatrixError("logDet: argument is not a Cholesky factor");
  DevView u(U, false);
  double out = 0.0;
  gpcCheck(gpc_logdet_chol_f64((int64_t)U.getRows(), u.p, (int64_t)ldOf(U), &out, 0));
  return out;
}

// ---- free functions / I/O ---------------------------------------------------------------------------------------------------
double trace(const CMatrix& A) { return A.trace(); }
double sum(const CMatrix& A) { return A.sum(); }
CMatrix sumCol(const CMatrix& A)
{
  CMatrix s(1, A.getCols());
  for(unsigned int j = 0; j < A.getCols(); j++) {
    double v = 0.0;
    for(unsigned int i = 0; i < A.getRows(); i++) v += A.getVal(i, j);
    s.setVal(v, 0, j);
  }
  return s;
}
CMatrix meanCol(const CMatrix& A)
{
  CMatrix m = sumCol(A);
  m.scale(1.0 / (double)A.getRows());
  return m;
}
CMatrix varCol(const CMatrix& A)
{
  // mean of squares minus square of the mean, as the reference's varCol does
  CMatrix mu = meanCol(A);
  CMatrix v(1, A.getCols());
  for(unsigned int j = 0; j < A.getCols(); j++) {
    double s2 = 0.0;
    for(unsigned int i = 0; i < A.getRows(); i++) s2 += A.getVal(i, j) * A.getVal(i, j);
    v.setVal(s2 / (double)A.getRows() - mu.getVal(0, j) * mu.getVal(0, j), 0, j);
  }
  return v;
}
CMatrix stdCol(const CMatrix& A)
{
  CMatrix v = varCol(A);
  for(unsigned int j = 0; j < v.getCols(); j++) v.setVal(std::sqrt(v.getVal(0, j)), 0, j);
  return v;
}
void CMatrix::toUnheadedStream(std::ostream& out) const
{
  // integers as integers like the reference (CMatrix.cpp:1158-1172); everything else with 17 significant digits in
  // plain decimal, which both the reference's atof-based reader and this one parse.
  char buf[64];
  for(unsigned int i = 0; i < getRows(); i++) {
    for(unsigned int j = 0; j < getCols(); j++) {
      const double val = getVal(i, j);
      if(std::fabs(val) < 2e9 && (val - (int)val) == 0.0) {
        out << (int)val << " ";
      } else {
        std::snprintf(buf, sizeof buf, "%.17g", val);
        std::string s(buf);
        if(s.find('.') == std::string::npos && s.find('e') == std::string::npos && s.find("inf") == std::string::npos &&
           s.find("nan") == std::string::npos)
          s += ".0";
        else if(s.find('.') == std::string::npos && s.find('e') != std::string::npos)
          s.insert(s.find('e'), ".0");   // the reference reader picks atof only when the token contains '.'
        out << s << " ";
      }
    }
    out << std::endl;
  }
}
void CMatrix::fromUnheadedStream(std::istream& in)
{
  std::vector<std::vector<double> > rows;
  std::string line;
  while(std::getline(in, line)) {
    if(!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
    if(line.empty() || line[0] == '#') continue;
    std::istringstream ss(line);
    std::vector<double> r;
    std::string tok;
    while(ss >> tok) r.push_back(std::strtod(tok.c_str(), 0));
    if(!rows.empty() && r.size() != rows[0].size()) throw ndlexceptions::StreamFormatError("matrix", "ragged rows");
    rows.push_back(r);
  }
  resize((unsigned int)rows.size(), rows.empty() ? 0 : (unsigned int)rows[0].size());
  for(size_t i = 0; i < rows.size(); i++)
    for(size_t j = 0; j < rows[i].size(); j++) setVal(rows[i][j], (unsigned int)i, (unsigned int)j);
}
void CMatrix::toUnheadedFile(const std::string fileName, const std::string comment) const
{
  std::ofstream out(fileName.c_str());
  if(!out) throw ndlexceptions::FileWriteError(fileName);
  if(comment.length() > 0) out << "#" << comment << std::endl;
  toUnheadedStream(out);
}
void CMatrix::fromUnheadedFile(const std::string fileName)
{
  std::ifstream in(fileName.c_str());
  if(!in) throw ndlexceptions::FileReadError(fileName);
  fromUnheadedStream(in);
}
using ndlstream::readField;
void CMatrix::writeParamsToStream(std::ostream& out) const
{
  out << "baseType=matrix" << std::endl << "type=doubleMatrix" << std::endl;
  out << "numRows=" << getRows() << std::endl << "numCols=" << getCols() << std::endl;
  toUnheadedStream(out);
}
void CMatrix::readParamsFromStream(std::istream& in)
{
  if(readField(in, "baseType") != "matrix") throw ndlexceptions::StreamFormatError("matrix", "Unexpected base type");
  if(readField(in, "type") != "doubleMatrix") throw ndlexceptions::StreamFormatError("matrix", "Unexpected matrix type");
  const int nr = std::atoi(readField(in, "numRows").c_str()), nc = std::atoi(readField(in, "numCols").c_str());
  resize(nr, nc);
  std::string line;
  for(int i = 0; i < nr; i++) {
    if(!ndlstream::getline(in, line)) throw ndlexceptions::StreamFormatError("matrix", "Incorrect number of rows in matrix.");
    std::istringstream ss(line);
    std::string tok;
    int j = 0;
    while(ss >> tok) {
      if(j >= nc) throw ndlexceptions::StreamFormatError("matrix", "Incorrect number of columns");
      setVal(std::strtod(tok.c_str(), 0), i, j++);   // strtod also reads the hexfloat tokens new libstdc++ builds emit
    }
    if(j != nc) throw ndlexceptions::StreamFormatError("matrix", "Incorrect number of columns");
  }
}
void CMatrix::fromStream(std::istream& in)   // CStreamInterface::fromStream: version line, then the parameters
{
  ndlstream::readVersion(in);
  readParamsFromStream(in);
}
std::ostream& operator<<(std::ostream& os, const CMatrix& A)
{
  A.toUnheadedStream(os);
  return os;
}
