// CMatrix.h -- GPc's dense column-major fp64 matrix surface (reference CMatrix.h:30-1274) on top of libgpc_hip.so.
//
// Same public names, argument meaning and error behaviour as the reference for everything gp.cpp, CKern and the FTC
// branches of CGp touch.  Differences that are deliberate:
//   * sizes are size_t internally (the reference's unsigned int nrows*ncols wraps at N = 65 536, CMatrix.h:1231-1232);
//     getRows()/getCols() keep the unsigned int return type;
//   * storage can live on the host (small things: parameters, X, y) or in HBM (the N x N objects): Residence;
//   * every LAPACK/BLAS-3 style member (potrf, chol, jitChol, pdinv, trsm, gemm, syrk, symv, trans, logDet) runs in
//     libgpc_hip.so -- operands on the host are staged to the device for the call.  There is no CPU LAPACK behind this
//     class; without a GPU these members throw ndlexceptions::DeviceError.
#ifndef GPC_AMD_CMATRIX_H
#define GPC_AMD_CMATRIX_H
#include <cstddef>
#include <iostream>
#include <string>
#include <vector>
#include "ndlexceptions.h"

class CMatrix {
 public:
  enum Residence { HOST = 0, DEVICE = 1 };

  CMatrix();
  explicit CMatrix(double val);
  CMatrix(unsigned int numRows, unsigned int numCols);
  CMatrix(unsigned int numRows, unsigned int numCols, double val);
  CMatrix(unsigned int numRows, unsigned int numCols, const double* inVals);
  CMatrix(unsigned int numRows, unsigned int numCols, Residence where);
  CMatrix(const CMatrix& A);
  CMatrix& operator=(const CMatrix& A);
  virtual ~CMatrix();

  void deepCopy(const CMatrix& A);                    // CMatrix.h:216-220
  void copy(const CMatrix& x) { deepCopy(x); }
  void resize(unsigned int rows, unsigned int cols);  // destroys contents (CMatrix.h:1205-1214)

  unsigned int getRows() const { return (unsigned int)nrows; }
  unsigned int getCols() const { return (unsigned int)ncols; }
  size_t getNumElements() const { return nrows * ncols; }
  bool isOnDevice() const { return where == DEVICE; }
  // host pointer (HOST matrices only) / device pointer (DEVICE matrices only)
  double* getVals();
  const double* getVals() const;
  double* devPtr() { return dev; }
  const double* devPtr() const { return dev; }
  void toDevice();   // move the storage to HBM
  void toHost();     // move it back

  double getVal(unsigned int i, unsigned int j) const;
  double getVal(unsigned int i) const;
  void setVal(double val, unsigned int i, unsigned int j);
  void setVal(double val, unsigned int i);
  void addVal(double val, unsigned int i, unsigned int j) { setVal(getVal(i, j) + val, i, j); }
  void addVal(double val, unsigned int i) { setVal(getVal(i) + val, i); }
  void setVals(double val);
  void zeros() { setVals(0.0); }
  void ones() { setVals(1.0); }
  void negate() { scale(-1.0); }

  bool isSquare() const { return nrows == ncols; }
  bool isTriangular() const { return triangular; }
  void setTriangular(bool v) { triangular = v; }
  bool isSymmetric() const { return symmetric; }
  void setSymmetric(bool v) { symmetric = v; }
  bool dimensionsMatch(const CMatrix& A) const { return nrows == A.nrows && ncols == A.ncols; }
  bool rowsMatch(const CMatrix& A) const { return nrows == A.nrows; }
  bool colsMatch(const CMatrix& A) const { return ncols == A.ncols; }

  // BLAS-1 style helpers on host matrices (CMatrix.h:408-640)
  void scale(double alpha);
  void scaleCol(unsigned int j, double alpha);
  void axpy(const CMatrix& x, double alpha);
  void add(const CMatrix& A) { axpy(A, 1.0); }
  void addCol(unsigned int j, double c);
  void addDiag(double c);
  void copyRowRow(unsigned int i, const CMatrix& X, unsigned int k);
  void copyColCol(unsigned int j, const CMatrix& X, unsigned int k);
  double normRow(unsigned int i) const;
  double norm2Row(unsigned int i) const;
  double norm2Col(unsigned int j) const;
  double dotRowRow(unsigned int i, const CMatrix& A, unsigned int k) const;
  double dotColCol(unsigned int j, const CMatrix& A, unsigned int k) const;
  double dist2Row(unsigned int i, const CMatrix& A, unsigned int k) const;
  double sum() const;
  double trace() const;
  double max() const;   // reproduces the reference's unbraced-loop bug (CMatrix.cpp:568-577): max(vals[0], vals[last])
  double maxAbsDiff(const CMatrix& X) const;
  bool equals(const CMatrix& A, double tol = 1e-10) const;
  void minRow(CMatrix& m) const;
  void maxRow(CMatrix& m) const;
  void getMatrix(CMatrix& out, unsigned int firstRow, unsigned int lastRow, unsigned int firstCol,
                 unsigned int lastCol) const;
  void setMatrix(unsigned int row, unsigned int col, const CMatrix& A);

  // LAPACK / BLAS-3 style members: all computed by libgpc_hip.so
  void potrf(const char* type);                                    // CMatrix.cpp:371-379
  void chol(const char* type);                                     // CMatrix.cpp:380-398
  void chol() { chol("U"); }
  double jitChol(CMatrix& A, unsigned int maxTries = 20);          // CMatrix.cpp:767-804
  void pdinv(const CMatrix& U);                                    // CMatrix.cpp:421-432
  void potri(const char* type);
  void trans();                                                    // CMatrix.h:789-801 (square: in place)
  void trsm(const CMatrix& A, double alpha, const char* side, const char* type, const char* trans,
            const char* diag);                                     // CMatrix.cpp:272-295 (this = B)
  void gemm(const CMatrix& A, const CMatrix& B, double alpha, double beta, const char* transa, const char* transb);
  void syrk(const CMatrix& A, double alpha, double beta, const char* type, const char* trans);  // + copySymmetric
  void symv(const CMatrix& A, const CMatrix& x, double alpha, double beta, const char* upperOrLower);
  void copySymmetric(const char* type);

  // text I/O in the reference's "unheaded" format (CMatrix.cpp:1057-1172)
  void toUnheadedStream(std::ostream& out) const;
  void fromUnheadedStream(std::istream& in);
  void toUnheadedFile(const std::string fileName, const std::string comment = "") const;
  void fromUnheadedFile(const std::string fileName);
  void writeParamsToStream(std::ostream& out) const;   // version/baseType/type/numRows/numCols + rows
  void readParamsFromStream(std::istream& in);
  void fromStream(std::istream& in);

 private:
  void alloc(size_t rows, size_t cols, Residence r);
  void release();
  size_t nrows, ncols;
  Residence where;
  std::vector<double> host;
  double* dev;
  bool symmetric, triangular;
};

double logDet(const CMatrix& U);          // CMatrix.cpp:404-412
double trace(const CMatrix& A);
double sum(const CMatrix& A);
CMatrix sumCol(const CMatrix& A);         // 1 x cols
CMatrix meanCol(const CMatrix& A);
CMatrix varCol(const CMatrix& A);
CMatrix stdCol(const CMatrix& A);
std::ostream& operator<<(std::ostream& os, const CMatrix& A);

// helper used by every class of the host layer: turn a libgpc_hip status into the matching exception
void gpcCheck(int rc);
#endif
