/* gpc_oracle.c -- plain-C CPU restatement of GPc's exact-GP (FTC) hot path.  TEST INFRASTRUCTURE ONLY; see
 * gpc_oracle.h for the scope, the parity status ("pinned") and the rules on who may call this. */
#include "gpc_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

#define A_(i, j) A[(i) + (size_t)(j) * lda]
#define HALFLOGTWOPI 0.91893853320467274178 /* ndlutil::HALFLOGTWOPI */
#define LIMVAL 36.0                         /* CTransform.h:18 */
#define EPS_ 2.220446049250313e-16          /* ndlutil::EPS */

/* ---- BLAS-1 pieces the reference reaches through CMatrix -------------------------------------------------------- */

/* dnrm2 (reference BLAS, scaled form), strided: CMatrix::normRow, CMatrix.h:580-584 */
static double nrm2(long n, const double* x, long inc)
{
  double scale = 0.0, ssq = 1.0;
  long i;
  if(n < 1) return 0.0;
  if(n == 1) return fabs(x[0]);
  for(i = 0; i < n; i++) {
    const double v = x[i * inc];
    if(v != 0.0) {
      const double a = fabs(v);
      if(scale < a) {
        ssq = 1.0 + ssq * (scale / a) * (scale / a);
        scale = a;
      } else {
        ssq += (a / scale) * (a / scale);
      }
    }
  }
  return scale * sqrt(ssq);
}

/* ddot: in the reference build the strided Fortran loop of ndlfortran.f:569 shadows the BLAS symbol */
static double dot(long n, const double* x, long incx, const double* y, long incy)
{
  double s = 0.0;
  long i;
  for(i = 0; i < n; i++) s += x[i * incx] * y[i * incy];
  return s;
}

/* CMatrix::dist2Row, CMatrix.h:554-560: norm2Row(i) + A.norm2Row(k) - 2*dotRowRow(i, A, k); norm2Row = dnrm2^2
 * (CMatrix.h:586-593). */
double orc_dist2_row(const double* X1, long ld1, long i, const double* X2, long ld2, long k, long D)
{
  const double n1 = nrm2(D, X1 + i, ld1), n2 = nrm2(D, X2 + k, ld2);
  return n1 * n1 + n2 * n2 - 2.0 * dot(D, X2 + k, ld2, X1 + i, ld1);
}

/* ---- LAPACK pieces ------------------------------------------------------------------------------------------------ */

/* dpotrf via the unblocked dpotf2 recurrences (reference LAPACK): CMatrix::potrf, CMatrix.cpp:371-379; the info
 * convention is lapack.h:59-65. */
int orc_potrf(char uplo, long N, double* A, long lda)
{
  long i, j, k;
  if(toupper(uplo) == 'U') {
    for(j = 0; j < N; j++) {
      double ajj = A_(j, j) - dot(j, &A_(0, j), 1, &A_(0, j), 1);
      if(!(ajj > 0.0)) {
        A_(j, j) = ajj;
        return (int)(j + 1);
      }
      ajj = sqrt(ajj);
      A_(j, j) = ajj;
      for(i = j + 1; i < N; i++) { /* row j right of the diagonal: dgemv('T') then dscal */
        double s = A_(j, i);
        for(k = 0; k < j; k++) s -= A_(k, j) * A_(k, i);
        A_(j, i) = s / ajj;
      }
    }
  } else {
    for(j = 0; j < N; j++) {
      double ajj = A_(j, j) - dot(j, &A_(j, 0), lda, &A_(j, 0), lda);
      if(!(ajj > 0.0)) {
        A_(j, j) = ajj;
        return (int)(j + 1);
      }
      ajj = sqrt(ajj);
      A_(j, j) = ajj;
      for(i = j + 1; i < N; i++) A_(i, j) -= dot(j, &A_(i, 0), lda, &A_(j, 0), lda);
      for(i = j + 1; i < N; i++) A_(i, j) /= ajj;
    }
  }
  return 0;
}

/* CMatrix::chol(type), CMatrix.cpp:380-398: potrf then zero the other triangle */
int orc_chol(char uplo, long N, double* A, long lda)
{
  long i, j;
  const int info = orc_potrf(uplo, N, A, lda);
  if(info != 0) return info;
  if(toupper(uplo) == 'L') {
    for(j = 0; j < N; j++)
      for(i = 0; i < j; i++) A_(i, j) = 0.0;
  } else {
    for(i = 0; i < N; i++)
      for(j = 0; j < i; j++) A_(i, j) = 0.0;
  }
  return 0;
}

/* CMatrix::jitChol, CMatrix.cpp:767-804.  A is modified by addDiag on every failure; returns the CURRENT jitter value
 * (the next candidate, even when none was added: SURVEY 0-7 iii).  *info != 0 mirrors the MatrixNonPosDef throws. */
double orc_jitchol(long N, double* A, double* U, int max_tries, int* info)
{
  long i;
  double tr = 0.0, jitter;
  int tries = 0, success = 0;
  for(i = 0; i < N; i++) tr += A[i + (size_t)i * N];
  jitter = 1e-6 * tr / (double)N;
  *info = 0;
  while(!success && tries < max_tries) {
    memcpy(U, A, sizeof(double) * (size_t)N * N); /* deepCopy(A) */
    if(orc_chol('U', N, U, N) == 0) {
      success = 1;
    } else {
      for(i = 0; i < N; i++) A[i + (size_t)i * N] += jitter; /* A.addDiag(jitter) */
      jitter *= 10.0;
      tries++;
      if(jitter > 10.0) {
        *info = 1;
        return jitter;
      }
    }
  }
  if(tries >= max_tries) *info = 1;
  return jitter;
}

/* logDet(U), CMatrix.cpp:404-412 */
double orc_logdet(long N, const double* U, long ldu)
{
  double s = 0.0;
  long i;
  for(i = 0; i < N; i++) s += log(U[i + (size_t)i * ldu]);
  return 2.0 * s;
}

/* CMatrix::pdinv(U), CMatrix.cpp:421-432: deepCopy(U); dpotri('U') = dtrtri('U','N') then dlauum('U') (reference
 * LAPACK, unblocked dtrti2 / dlauu2 recurrences); then mirror the upper triangle into the lower one. */
void orc_pdinv_upper(long N, const double* U, double* A)
{
  const long lda = N;
  long i, j, k;
  memcpy(A, U, sizeof(double) * (size_t)N * N);
  /* dtrti2, upper, non-unit: column j of inv(U) */
  for(j = 0; j < N; j++) {
    double ajj;
    A_(j, j) = 1.0 / A_(j, j);
    ajj = -A_(j, j);
    /* x := inv(U)(0:j,0:j) * U(0:j, j)  (dtrmv upper no-trans on the already inverted leading block), then scale */
    for(k = 0; k < j; k++) {
      const double t = A_(k, j);
      if(t != 0.0) {
        for(i = 0; i < k; i++) A_(i, j) += t * A_(i, k);
        A_(k, j) = t * A_(k, k);
      }
    }
    for(i = 0; i < j; i++) A_(i, j) *= ajj;
  }
  /* dlauu2, upper: A := V * V' with V = inv(U) upper triangular */
  for(i = 0; i < N; i++) {
    const double aii = A_(i, i);
    if(i < N - 1) {
      A_(i, i) = dot(N - i, &A_(i, i), lda, &A_(i, i), lda);
      /* A(0:i, i) := aii * A(0:i, i) + A(0:i, i+1:N) * A(i, i+1:N)' */
      for(k = 0; k < i; k++) {
        double s = aii * A_(k, i);
        for(j = i + 1; j < N; j++) s += A_(k, j) * A_(i, j);
        A_(k, i) = s;
      }
    } else {
      for(k = 0; k <= i; k++) A_(k, i) *= aii;
    }
  }
  for(i = 0; i < N; i++)
    for(j = 0; j < i; j++) A_(i, j) = A_(j, i);
}

/* CMatrix::trans for a square matrix (CMatrix.h:789-801) -> dtransr_, square branch, ndlfortran.f:2138-2157:
 *     DO I=1,N-1; DO J=I+1,N:  I1 = (I,J);  I2 = (J,I);  B = A(I1); A(I1) = A(I2); A(I2) = B
 * Only A is declared DOUBLE PRECISION there; the temporary B is implicitly REAL.  A reference built from the Fortran
 * source (make.linux uses gfortran; SURVEY 8c uses flang) therefore stores the element that lands BELOW the diagonal
 * rounded to single precision, the one that lands above it exactly.  (The f2c twin, ndlfortran.c:1232, declares
 * `doublereal b` and is exact.)  orc_exact_trans = 1 restates the f2c behaviour instead. */
int orc_exact_trans = 0;
void orc_trans(long N, double* A)
{
  long i, j;
  for(i = 0; i < N - 1; i++)
    for(j = i + 1; j < N; j++) {
      const size_t i1 = (size_t)i + (size_t)j * N; /* (I,J), upper */
      const size_t i2 = (size_t)j + (size_t)i * N; /* (J,I), lower */
      if(orc_exact_trans) {
        const double b = A[i1];
        A[i1] = A[i2];
        A[i2] = b;
      } else {
        const float b = (float)A[i1];
        A[i1] = A[i2];
        A[i2] = (double)b;
      }
    }
}

/* dtrsm (reference BLAS loops): CMatrix::trsm, CMatrix.cpp:272-295; lapack.h:208-218 */
void orc_trsm(char side, char uplo, char trans, char diag, long M, long N, double alpha, const double* A, long lda,
              double* B, long ldb)
{
#define B_(i, j) B[(i) + (size_t)(j) * ldb]
  const int lside = toupper(side) == 'L', upper = toupper(uplo) == 'U', notr = toupper(trans) == 'N',
            nounit = toupper(diag) == 'N';
  long i, j, k;
  if(M == 0 || N == 0) return;
  if(alpha == 0.0) {
    for(j = 0; j < N; j++)
      for(i = 0; i < M; i++) B_(i, j) = 0.0;
    return;
  }
  if(lside) {
    if(notr) { /* B := alpha * inv(A) * B */
      for(j = 0; j < N; j++) {
        if(alpha != 1.0)
          for(i = 0; i < M; i++) B_(i, j) *= alpha;
        if(upper) {
          for(k = M - 1; k >= 0; k--) {
            if(B_(k, j) != 0.0) {
              if(nounit) B_(k, j) /= A_(k, k);
              for(i = 0; i < k; i++) B_(i, j) -= B_(k, j) * A_(i, k);
            }
          }
        } else {
          for(k = 0; k < M; k++) {
            if(B_(k, j) != 0.0) {
              if(nounit) B_(k, j) /= A_(k, k);
              for(i = k + 1; i < M; i++) B_(i, j) -= B_(k, j) * A_(i, k);
            }
          }
        }
      }
    } else { /* B := alpha * inv(A') * B */
      for(j = 0; j < N; j++) {
        if(upper) {
          for(i = 0; i < M; i++) {
            double t = alpha * B_(i, j);
            for(k = 0; k < i; k++) t -= A_(k, i) * B_(k, j);
            if(nounit) t /= A_(i, i);
            B_(i, j) = t;
          }
        } else {
          for(i = M - 1; i >= 0; i--) {
            double t = alpha * B_(i, j);
            for(k = i + 1; k < M; k++) t -= A_(k, i) * B_(k, j);
            if(nounit) t /= A_(i, i);
            B_(i, j) = t;
          }
        }
      }
    }
  } else {
    if(notr) { /* B := alpha * B * inv(A) */
      if(upper) {
        for(j = 0; j < N; j++) {
          if(alpha != 1.0)
            for(i = 0; i < M; i++) B_(i, j) *= alpha;
          for(k = 0; k < j; k++)
            if(A_(k, j) != 0.0)
              for(i = 0; i < M; i++) B_(i, j) -= A_(k, j) * B_(i, k);
          if(nounit) {
            const double t = 1.0 / A_(j, j);
            for(i = 0; i < M; i++) B_(i, j) *= t;
          }
        }
      } else {
        for(j = N - 1; j >= 0; j--) {
          if(alpha != 1.0)
            for(i = 0; i < M; i++) B_(i, j) *= alpha;
          for(k = j + 1; k < N; k++)
            if(A_(k, j) != 0.0)
              for(i = 0; i < M; i++) B_(i, j) -= A_(k, j) * B_(i, k);
          if(nounit) {
            const double t = 1.0 / A_(j, j);
            for(i = 0; i < M; i++) B_(i, j) *= t;
          }
        }
      }
    } else { /* B := alpha * B * inv(A') */
      if(upper) {
        for(k = N - 1; k >= 0; k--) {
          if(nounit) {
            const double t = 1.0 / A_(k, k);
            for(i = 0; i < M; i++) B_(i, k) *= t;
          }
          for(j = 0; j < k; j++)
            if(A_(j, k) != 0.0) {
              const double t = A_(j, k);
              for(i = 0; i < M; i++) B_(i, j) -= t * B_(i, k);
            }
          if(alpha != 1.0)
            for(i = 0; i < M; i++) B_(i, k) *= alpha;
        }
      } else {
        for(k = 0; k < N; k++) {
          if(nounit) {
            const double t = 1.0 / A_(k, k);
            for(i = 0; i < M; i++) B_(i, k) *= t;
          }
          for(j = k + 1; j < N; j++)
            if(A_(j, k) != 0.0) {
              const double t = A_(j, k);
              for(i = 0; i < M; i++) B_(i, j) -= t * B_(i, k);
            }
          if(alpha != 1.0)
            for(i = 0; i < M; i++) B_(i, k) *= alpha;
        }
      }
    }
  }
#undef B_
}

/* dsymv('U'): CMatrix::symvColCol as used by CGp.cpp:675, 928 */
void orc_symv_upper(long N, const double* A, const double* x, double* y)
{
  const long lda = N;
  long i, j;
  for(i = 0; i < N; i++) y[i] = 0.0;
  for(j = 0; j < N; j++) {
    const double t1 = x[j];
    double t2 = 0.0;
    for(i = 0; i < j; i++) {
      y[i] += t1 * A_(i, j);
      t2 += A_(i, j) * x[i];
    }
    y[j] += t1 * A_(j, j) + t2;
  }
}

/* ---- kernels --------------------------------------------------------------------------------------------------------- */

/* CCmpndKern::computeElement, CKern.cpp:219-226: sum of the components' computeElement */
double orc_kern_element(const orc_kspec* ks, const double* X1, long ld1, long i, const double* X2, long ld2, long j,
                        long D)
{
  double k = 0.0;
  int t;
  long q;
  for(t = 0; t < ks->n_terms; t++) {
    const double* p = ks->params + ks->offs[t];
    switch(ks->types[t]) {
    case ORC_KERN_RBF: { /* CRbfKern::computeElement, CKern.cpp:1147-1154 */
      double v = orc_dist2_row(X1, ld1, i, X2, ld2, j, D);
      v = 0.5 * v * p[0];
      k += p[1] * exp(-v);
      break;
    }
    case ORC_KERN_RBFARD: { /* CRbfardKern::computeElement, CKern.cpp:3305-3316 */
      double val = 0.0;
      for(q = 0; q < D; q++) {
        double x = X1[i + q * ld1];
        x = x - X2[j + q * ld2];
        val += x * p[2 + q] * x;
      }
      k += p[1] * exp(-val * p[0] * 0.5);
      break;
    }
    case ORC_KERN_WHITE: /* CWhiteKern::computeElement returns 0, CKern.cpp:702-705 */ break;
    case ORC_KERN_BIAS: k += p[0]; break; /* CKern.cpp:989-993 */
    case ORC_KERN_LIN: k += p[0] * dot(D, X2 + j, ld2, X1 + i, ld1); break; /* CKern.cpp:2328-2332 */
    default: break;
    }
  }
  return k;
}

/* CCmpndKern::diagComputeElement, CKern.cpp:165-171 */
double orc_kern_diag_element(const orc_kspec* ks, const double* X, long ld, long i, long D)
{
  double k = 0.0;
  int t;
  for(t = 0; t < ks->n_terms; t++) {
    const double* p = ks->params + ks->offs[t];
    switch(ks->types[t]) {
    case ORC_KERN_RBF: k += p[1]; break;    /* CKern.cpp:1074-1077 */
    case ORC_KERN_RBFARD: k += p[1]; break; /* CKern.cpp:3294-3297 */
    case ORC_KERN_WHITE: k += p[0]; break;  /* CKern.cpp:646-649 */
    case ORC_KERN_BIAS: k += p[0]; break;   /* CKern.cpp:933-936 */
    case ORC_KERN_LIN: {                    /* CKern.cpp:2262-2265: variance * norm2Row */
      const double n = nrm2(D, X + i, ld);
      k += p[0] * n * n;
      break;
    }
    default: break;
    }
  }
  return k;
}

/* CKern::compute(K, X), CKern.h:128-144 == CGp::_updateK FTC, CGp.cpp:698-712 */
void orc_gram_sym(const orc_kspec* ks, const double* X, long N, long D, double* K)
{
  long i, j;
  for(i = 0; i < N; i++) {
    for(j = 0; j < i; j++) {
      const double k = orc_kern_element(ks, X, N, i, X, N, j, D);
      K[i + (size_t)j * N] = k;
      K[j + (size_t)i * N] = k;
    }
    K[i + (size_t)i * N] = orc_kern_diag_element(ks, X, N, i, D);
  }
}

/* CKern::compute(K, X, X2), CKern.h:146-157 */
void orc_gram_cross(const orc_kspec* ks, const double* X, long N, const double* X2, long N2, long D, double* K)
{
  long i, j;
  for(i = 0; i < N; i++)
    for(j = 0; j < N2; j++) K[i + (size_t)j * N] = orc_kern_element(ks, X, N, i, X2, N2, j, D);
}

/* CKern::diagCompute, CKern.h:49-55 */
void orc_gram_diag(const orc_kspec* ks, const double* X, long N, long D, double* d)
{
  long i;
  for(i = 0; i < N; i++) d[i] = orc_kern_diag_element(ks, X, N, i, D);
}

/* CCmpndKern::getGradParams(g, X, covGrad), CKern.cpp:284-298, over the components' own loops */
void orc_kern_grad_sym(const orc_kspec* ks, const double* X, long N, long D, const double* cg, double* g)
{
#define CG(i, j) cg[(i) + (size_t)(j) * N]
  int t;
  long i, j, q;
  for(t = 0; t < ks->n_terms; t++) {
    const double* p = ks->params + ks->offs[t];
    double* gt = g + ks->offs[t];
    switch(ks->types[t]) {
    case ORC_KERN_RBF: { /* CRbfKern::getGradParams, CKern.cpp:1204-1241 */
      double g1 = 0.0, g2 = 0.0;
      const double halfIw = 0.5 * p[0], halfVar = 0.5 * p[1];
      for(j = 0; j < N; j++) {
        g2 += CG(j, j);
        for(i = 0; i < j; i++) {
          const double dist2 = orc_dist2_row(X, N, i, X, N, j, D);
          const double k = exp(-dist2 * halfIw);
          const double kcg = k * CG(i, j);
          g1 -= 2.0 * halfVar * dist2 * kcg;
          g2 += 2.0 * kcg;
        }
      }
      gt[0] = g1;
      gt[1] = g2;
      break;
    }
    case ORC_KERN_RBFARD: { /* CRbfardKern::getGradParams, CKern.cpp:3359-3403 */
      double g1 = 0.0, g2 = 0.0;
      const double halfIw = 0.5 * p[0];
      for(q = 0; q < D; q++) gt[2 + q] = 0.0;
      for(i = 0; i < N; i++) {
        for(j = 0; j < i; j++) {
          double val = 0.0, kcg;
          for(q = 0; q < D; q++) {
            double x = X[i + q * N];
            x -= X[j + q * N];
            val += x * p[2 + q] * x;
          }
          kcg = exp(-halfIw * val) * CG(i, j);
          g1 -= 0.5 * val * kcg * p[1];
          g2 += kcg;
          for(q = 0; q < D; q++) {
            const double xi = X[i + q * N], xj = X[j + q * N];
            gt[2 + q] += p[0] * kcg * (xi * xj - .5 * xi * xi - .5 * xj * xj) * p[1];
          }
        }
      }
      g1 *= 2.0;
      g2 *= 2.0;
      for(i = 0; i < N; i++) g2 += CG(i, i);
      for(q = 0; q < D; q++) gt[2 + q] *= 2.0;
      gt[0] = g1;
      gt[1] = g2;
      break;
    }
    case ORC_KERN_WHITE: { /* trace(covGrad), CKern.cpp:735-739 */
      double s = 0.0;
      for(i = 0; i < N; i++) s += CG(i, i);
      gt[0] = s;
      break;
    }
    case ORC_KERN_BIAS: { /* covGrad.sum(), CKern.cpp:1020-1024 */
      double s = 0.0;
      for(j = 0; j < N; j++)
        for(i = 0; i < N; i++) s += CG(i, j);
      gt[0] = s;
      break;
    }
    case ORC_KERN_LIN: { /* CKern.cpp:2369-2383 */
      double s = 0.0;
      for(i = 0; i < N; i++)
        for(j = 0; j < N; j++) s += dot(D, X + j, N, X + i, N) * CG(i, j);
      gt[0] = s;
      break;
    }
    default: break;
    }
  }
#undef CG
}

/* CCmpndKern::getGradParams(g, X, X2, covGrad): rbf CKern.cpp:1175-1202, rbfard 3318-3357, white 730-734,
 * bias 1015-1019, lin 2354-2368 */
void orc_kern_grad_cross(const orc_kspec* ks, const double* X, long N, const double* X2, long N2, long D,
                         const double* cg, double* g)
{
#define CG(i, j) cg[(i) + (size_t)(j) * N]
  int t;
  long i, j, q;
  for(t = 0; t < ks->n_terms; t++) {
    const double* p = ks->params + ks->offs[t];
    double* gt = g + ks->offs[t];
    switch(ks->types[t]) {
    case ORC_KERN_RBF: {
      double g1 = 0.0, g2 = 0.0;
      for(j = 0; j < N; j++)
        for(i = 0; i < N2; i++) {
          const double dist2 = orc_dist2_row(X2, N2, i, X, N, j, D);
          const double kcg = exp(-dist2 * 0.5 * p[0]) * CG(j, i);
          g1 -= 0.5 * p[1] * dist2 * kcg;
          g2 += kcg;
        }
      gt[0] = g1;
      gt[1] = g2;
      break;
    }
    case ORC_KERN_RBFARD: {
      double g1 = 0.0, g2 = 0.0;
      for(q = 0; q < D; q++) gt[2 + q] = 0.0;
      for(i = 0; i < N; i++)
        for(j = 0; j < N2; j++) {
          double val = 0.0, kcg;
          for(q = 0; q < D; q++) {
            double x = X[i + q * N];
            x -= X2[j + q * N2];
            val += x * p[2 + q] * x;
          }
          kcg = exp(-0.5 * p[0] * val) * CG(i, j);
          g1 -= 0.5 * val * kcg * p[1];
          g2 += kcg;
          for(q = 0; q < D; q++) {
            const double xi = X[i + q * N], xj = X2[j + q * N2];
            gt[2 + q] += p[0] * kcg * (xi * xj - .5 * xi * xi - .5 * xj * xj) * p[1];
          }
        }
      gt[0] = g1;
      gt[1] = g2;
      break;
    }
    case ORC_KERN_WHITE: gt[0] = 0.0; break;
    case ORC_KERN_BIAS: {
      double s = 0.0;
      for(j = 0; j < N2; j++)
        for(i = 0; i < N; i++) s += CG(i, j);
      gt[0] = s;
      break;
    }
    case ORC_KERN_LIN: {
      double s = 0.0;
      for(i = 0; i < N; i++)
        for(j = 0; j < N2; j++) s += dot(D, X2 + j, N2, X + i, N) * CG(i, j);
      gt[0] = s;
      break;
    }
    default: break;
    }
  }
#undef CG
}

/* transform kind of parameter q of term type: rbfard input scales are sigmoid (CKern.cpp:3214-3217), all other
 * parameters of the in-scope kernels are exp ("defaultPositive"). */
static int is_sigmoid(int type, int q) { return type == ORC_KERN_RBFARD && q >= 2; }

/* CKern::getGradTransParams, CKern.cpp:50-63, with CExpTransform::gradfact = x and CSigmoidTransform::gradfact =
 * x(1-x) (CTransform.cpp:50-53, 109-112) */
void orc_grad_to_trans(const orc_kspec* ks, long D, double* g)
{
  int t, q;
  (void)D;
  for(t = 0; t < ks->n_terms; t++)
    for(q = 0; q < ks->offs[t + 1] - ks->offs[t]; q++) {
      const double x = ks->params[ks->offs[t] + q];
      g[ks->offs[t] + q] *= is_sigmoid(ks->types[t], q) ? x * (1.0 - x) : x;
    }
}

/* CTransformable::getTransParams via xtoa: log(x) (CTransform.cpp:44-49) / invSigmoid (CTransform.cpp:106-108) */
void orc_trans_params(const orc_kspec* ks, long D, double* a)
{
  int t, q;
  (void)D;
  for(t = 0; t < ks->n_terms; t++)
    for(q = 0; q < ks->offs[t + 1] - ks->offs[t]; q++) {
      const double x = ks->params[ks->offs[t] + q];
      a[ks->offs[t] + q] = is_sigmoid(ks->types[t], q) ? log(x / (1.0 - x)) : log(x);
    }
}

/* ---- CGp (FTC) --------------------------------------------------------------------------------------------------------- */

orc_gp* orc_gp_create(const orc_kspec* ks, const double* X, long N, long D, const double* y, long d,
                      const double* scale, const double* bias)
{
  long i, j;
  orc_gp* gp = (orc_gp*)calloc(1, sizeof(orc_gp));
  gp->N = N;
  gp->D = D;
  gp->d = d;
  gp->ks = *ks;
  gp->X = X;
  gp->m = (double*)malloc(sizeof(double) * (size_t)N * d);
  gp->scale = (double*)malloc(sizeof(double) * d);
  gp->bias = (double*)malloc(sizeof(double) * d);
  gp->K = (double*)malloc(sizeof(double) * (size_t)N * N);       /* CGp::initStoreage, CGp.cpp:171-175 */
  gp->L = (double*)malloc(sizeof(double) * (size_t)N * N);
  gp->invK = (double*)malloc(sizeof(double) * (size_t)N * N);
  gp->covGrad = (double*)malloc(sizeof(double) * (size_t)N * N);
  gp->Alpha = (double*)malloc(sizeof(double) * (size_t)N * d);
  for(j = 0; j < d; j++) {
    gp->scale[j] = scale ? scale[j] : 1.0;
    if(bias) {
      gp->bias[j] = bias[j];
    } else { /* gp.cpp:384-385: bias = meanCol(y) */
      double s = 0.0;
      for(i = 0; i < N; i++) s += y[i + (size_t)j * N];
      gp->bias[j] = s / (double)N;
    }
    /* CGp::updateM, CGp.cpp:248-260 */
    for(i = 0; i < N; i++) gp->m[i + (size_t)j * N] = (y[i + (size_t)j * N] - gp->bias[j]) / gp->scale[j];
  }
  return gp;
}

void orc_gp_free(orc_gp* gp)
{
  if(!gp) return;
  free(gp->m);
  free(gp->scale);
  free(gp->bias);
  free(gp->K);
  free(gp->L);
  free(gp->invK);
  free(gp->covGrad);
  free(gp->Alpha);
  free(gp);
}

/* CGp::updateK, CGp.cpp:682-691: _updateK (698-712) then _updateInvK FTC (881-891):
 * jitChol -> logDet -> pdinv -> trans */
int orc_gp_update_k(orc_gp* gp)
{
  const long N = gp->N;
  orc_gram_sym(&gp->ks, gp->X, N, gp->D, gp->K);
  gp->jitter = orc_jitchol(N, gp->K, gp->L, 20, &gp->info);
  if(gp->info != 0) return gp->info;
  gp->logDetK = orc_logdet(N, gp->L, N);
  orc_pdinv_upper(N, gp->L, gp->invK);
  orc_trans(N, gp->L); /* LcholK.trans(): now lower */
  return 0;
}

/* CGp::updateAlpha FTC, CGp.cpp:469-489 */
void orc_gp_update_alpha(orc_gp* gp)
{
  memcpy(gp->Alpha, gp->m, sizeof(double) * (size_t)gp->N * gp->d);
  orc_trsm('l', 'l', 'n', 'n', gp->N, gp->d, 1.0, gp->L, gp->N, gp->Alpha, gp->N);
  orc_trsm('l', 'l', 't', 'n', gp->N, gp->d, 1.0, gp->L, gp->N, gp->Alpha, gp->N);
}

/* CGp::logLikelihood FTC, CGp.cpp:913-938 and 1002-1013 (K must be up to date) */
double orc_gp_loglik(orc_gp* gp)
{
  const long N = gp->N;
  long j;
  double L = 0.0;
  double* invKm = (double*)malloc(sizeof(double) * N);
  for(j = 0; j < gp->d; j++) {
    orc_symv_upper(N, gp->invK, gp->m + (size_t)j * N, invKm);
    L += dot(N, invKm, 1, gp->m + (size_t)j * N, 1);
    L += gp->logDetK;
  }
  free(invKm);
  L *= -0.5;
  L -= (double)gp->d * (double)N * HALFLOGTWOPI;
  return L;
}

/* CGp::logLikelihoodGradient / updateG FTC, CGp.cpp:1016-1079, 1080-1117; updateCovGradient 666-679 */
double orc_gp_loglik_grad(orc_gp* gp, double* g)
{
  const long N = gp->N;
  const int np = gp->ks.offs[gp->ks.n_terms];
  long i, j, c;
  int q;
  double* invKm = (double*)malloc(sizeof(double) * N);
  double* tmp = (double*)malloc(sizeof(double) * (np > 0 ? np : 1));
  for(q = 0; q < np; q++) g[q] = 0.0;
  for(j = 0; j < gp->d; j++) {
    orc_symv_upper(N, gp->invK, gp->m + (size_t)j * N, invKm);
    /* covGrad = invK; syr(invKm, -1, "u") + copySymmetric (CMatrix.h:526-533); scale(-0.5) */
    for(c = 0; c < N; c++)
      for(i = 0; i <= c; i++) {
        const double v = -0.5 * (gp->invK[i + (size_t)c * N] - invKm[i] * invKm[c]);
        gp->covGrad[i + (size_t)c * N] = v;
        gp->covGrad[c + (size_t)i * N] = v;
      }
    orc_kern_grad_sym(&gp->ks, gp->X, N, gp->D, gp->covGrad, tmp);
    orc_grad_to_trans(&gp->ks, gp->D, tmp);
    for(q = 0; q < np; q++) g[q] += tmp[q];
  }
  free(invKm);
  free(tmp);
  return orc_gp_loglik(gp);
}

/* CGp::posteriorMeanVar FTC, CGp.cpp:642-663 -> _testComputeKx 540-547, _posteriorMean 548-574,
 * _posteriorVar 575-625 */
void orc_gp_posterior(orc_gp* gp, const double* Xs, long Ns, double* mu, double* var)
{
  const long N = gp->N;
  long i, j;
  double* kX = (double*)malloc(sizeof(double) * (size_t)N * Ns);
  orc_gram_cross(&gp->ks, gp->X, N, Xs, Ns, gp->D, kX);
  for(i = 0; i < Ns; i++)
    for(j = 0; j < gp->d; j++) {
      double v = dot(N, gp->Alpha + (size_t)j * N, 1, kX + (size_t)i * N, 1);
      if(gp->scale[j] != 1.0) v *= gp->scale[j];
      if(gp->bias[j] != 0.0) v += gp->bias[j];
      mu[i + (size_t)j * Ns] = v;
    }
  orc_trsm('L', 'L', 'N', 'N', N, Ns, 1.0, gp->L, N, kX, N);
  for(i = 0; i < Ns; i++) {
    const double n = nrm2(N, kX + (size_t)i * N, 1); /* norm2Col = dnrm2^2, CMatrix.h:601-608 */
    const double vs = orc_kern_diag_element(&gp->ks, Xs, Ns, i, gp->D) - n * n;
    for(j = 0; j < gp->d; j++) {
      double v = vs;
      if(gp->scale[j] != 1.0) v *= gp->scale[j] * gp->scale[j];
      var[i + (size_t)j * Ns] = v;
    }
  }
  free(kX);
}

/* ---- GP-LVM (no dynamics, no back constraints, scales not learnt) ------------------- */

/* CCmpndKern::getGradX(gX, X, row, X2 = X), CKern.cpp:184-193: gX(k,j) = sum over components of
 * d k(x_row, x_k) / d x_row,j.  gX is N x D. */
void orc_kern_gradx_row(const orc_kspec* ks, const double* X, long N, long D, long row, double* gX)
{
  int t;
  long k, j;
  for(k = 0; k < N * D; k++) gX[k] = 0.0;
  for(t = 0; t < ks->n_terms; t++) {
    const double* p = ks->params + ks->offs[t];
    switch(ks->types[t]) {
    case ORC_KERN_RBF: { /* CRbfKern::getGradX, CKern.cpp:1115-1135 */
      const double wi2 = 0.5 * p[0], pf = p[1] * p[0];
      for(k = 0; k < N; k++) {
        const double n2 = orc_dist2_row(X, N, row, X, N, k, D);
        for(j = 0; j < D; j++) gX[k + j * N] += pf * (X[k + j * N] - X[row + j * N]) * exp(-n2 * wi2);
      }
      break;
    }
    case ORC_KERN_RBFARD: { /* CRbfardKern::getGradX, CKern.cpp:3268-3293 */
      const double wi2 = 0.5 * p[0], pf = p[1] * p[0];
      for(k = 0; k < N; k++) {
        double n2 = 0.0;
        for(j = 0; j < D; j++) {
          double x = X[row + j * N];
          x = x - X[k + j * N];
          n2 += x * p[2 + j] * x;
        }
        for(j = 0; j < D; j++)
          gX[k + j * N] += pf * (X[k + j * N] - X[row + j * N]) * exp(-n2 * wi2) * p[2 + j];
      }
      break;
    }
    case ORC_KERN_LIN: /* CLinKern::getGradX, CKern.cpp:2291-2308 */
      for(k = 0; k < N; k++)
        for(j = 0; j < D; j++) gX[k + j * N] += p[0] * X[k + j * N];
      break;
    default: break; /* white (CKern.cpp:681-690) and bias (968-977) contribute nothing */
    }
  }
}

/* CCmpndKern::getDiagGradX, CKern.cpp:194-203: only the linear kernel has one (2 variance X, CKern.cpp:2310-2322) */
void orc_kern_diag_gradx(const orc_kspec* ks, const double* X, long N, long D, double* gD)
{
  int t;
  long k;
  for(k = 0; k < N * D; k++) gD[k] = 0.0;
  for(t = 0; t < ks->n_terms; t++)
    if(ks->types[t] == ORC_KERN_LIN)
      for(k = 0; k < N * D; k++) gD[k] += 2.0 * ks->params[ks->offs[t]] * X[k];
}

/* CGplvm::updateK + logLikelihood + logLikelihoodGradient, CGplvm.cpp:402-446, 493-553, 555-716, for the plain model
 * gplvm.cpp builds by default (dynamicsLearnt = backConstrained = inputScaleLearnt = false).
 *   m: N x d centred data (CScaleNoise::updateSites, CNoise.cpp:710-721), X: N x q latent points.
 *   g (may be NULL): nk transformed-kernel-parameter gradients, then dL/dX column by column (CGplvm.cpp:257-290).
 * Returns the log-likelihood (which, unlike CGp's, carries no -dN/2 log 2 pi term); *info = chol's LAPACK info. */
double orc_gplvm_loglik_grad(const orc_kspec* ks, const double* m, long N, long d, const double* X, long q,
                             int regularise, double* g, double* logdet_out, int* info)
{
  const int nk = ks->offs[ks->n_terms];
  const size_t NN = (size_t)N * N;
  double* K = (double*)malloc(sizeof(double) * NN);
  double* U = (double*)malloc(sizeof(double) * NN);
  double* invK = (double*)malloc(sizeof(double) * NN);
  double* invKm = (double*)malloc(sizeof(double) * N);
  double L = 0.0, logDetK;
  long i, j, c, k;
  orc_gram_sym(ks, X, N, q, K);                        /* _updateK, CGplvm.cpp:418-432 */
  memcpy(U, K, sizeof(double) * NN);                   /* _updateInvK, 435-446: chol (no jitter), logDet, pdinv */
  *info = orc_chol('U', N, U, N);
  if(*info != 0) {
    free(K); free(U); free(invK); free(invKm);
    return 0.0;
  }
  logDetK = orc_logdet(N, U, N);
  if(logdet_out) *logdet_out = logDetK;
  orc_pdinv_upper(N, U, invK);
  for(j = 0; j < d; j++) {                             /* logLikelihood, 493-553 */
    orc_symv_upper(N, invK, m + (size_t)j * N, invKm);
    L += dot(N, invKm, 1, m + (size_t)j * N, 1);
    L += logDetK;
  }
  if(regularise)
    for(j = 0; j < q; j++) {
      const double n = nrm2(N, X + (size_t)j * N, 1);  /* norm2Col = dnrm2^2 */
      L += n * n;
    }
  L *= -0.5;
  if(g) {                                              /* logLikelihoodGradient, 555-716 */
    double* covGrad = (double*)malloc(sizeof(double) * NN);
    double* gXi = (double*)malloc(sizeof(double) * (size_t)N * q);
    double* gD = (double*)malloc(sizeof(double) * (size_t)N * q);
    double* tmp = (double*)malloc(sizeof(double) * (nk > 0 ? nk : 1));
    for(k = 0; k < nk + N * q; k++) g[k] = 0.0;
    orc_kern_diag_gradx(ks, X, N, q, gD);
    for(j = 0; j < d; j++) {
      orc_symv_upper(N, invK, m + (size_t)j * N, invKm);   /* updateCovGradient, 365-378 */
      for(c = 0; c < N; c++)
        for(i = 0; i <= c; i++) {
          const double v = -0.5 * (invK[i + (size_t)c * N] - invKm[i] * invKm[c]);
          covGrad[i + (size_t)c * N] = v;
          covGrad[c + (size_t)i * N] = v;
        }
      orc_kern_grad_sym(ks, X, N, q, covGrad, tmp);
      orc_grad_to_trans(ks, q, tmp);
      for(k = 0; k < nk; k++) g[k] += tmp[k];
      for(i = 0; i < N; i++) {
        orc_kern_gradx_row(ks, X, N, q, i, gXi);
        for(k = 0; k < N * q; k++) gXi[k] *= 2.0;         /* "accounts for symmetric covariance", 577 */
        for(k = 0; k < q; k++) gXi[i + k * N] = gD[i + k * N];
        for(k = 0; k < q; k++) g[nk + i + N * k] += dot(N, gXi + (size_t)k * N, 1, covGrad + (size_t)i * N, 1);
      }
    }
    if(regularise)
      for(i = 0; i < N; i++)
        for(k = 0; k < q; k++) g[nk + i + N * k] += -X[i + N * k];
    free(covGrad); free(gXi); free(gD); free(tmp);
  }
  free(K); free(U); free(invK); free(invKm);
  return L;
}

/* ---- sparse approximation DTC (spherical noise, inducing points optimised) --------------- */

/* CCmpndKern::getGradX(gX, X, row, X2): gX(k,j) = d k(x_row, x2_k) / d x_row,j, N2 x D (CKern.cpp:184-193; components
 * as in orc_kern_gradx_row) */
void orc_kern_gradx_row2(const orc_kspec* ks, const double* X, long ldx, long row, const double* X2, long N2, long D,
                         double* gX)
{
  int t;
  long k, j;
  for(k = 0; k < N2 * D; k++) gX[k] = 0.0;
  for(t = 0; t < ks->n_terms; t++) {
    const double* p = ks->params + ks->offs[t];
    switch(ks->types[t]) {
    case ORC_KERN_RBF: {
      const double wi2 = 0.5 * p[0], pf = p[1] * p[0];
      for(k = 0; k < N2; k++) {
        const double n2 = orc_dist2_row(X, ldx, row, X2, N2, k, D);
        for(j = 0; j < D; j++) gX[k + j * N2] += pf * (X2[k + j * N2] - X[row + j * ldx]) * exp(-n2 * wi2);
      }
      break;
    }
    case ORC_KERN_RBFARD: {
      const double wi2 = 0.5 * p[0], pf = p[1] * p[0];
      for(k = 0; k < N2; k++) {
        double n2 = 0.0;
        for(j = 0; j < D; j++) {
          double x = X[row + j * ldx];
          x = x - X2[k + j * N2];
          n2 += x * p[2 + j] * x;
        }
        for(j = 0; j < D; j++)
          gX[k + j * N2] += pf * (X2[k + j * N2] - X[row + j * ldx]) * exp(-n2 * wi2) * p[2 + j];
      }
      break;
    }
    case ORC_KERN_LIN:
      for(k = 0; k < N2; k++)
        for(j = 0; j < D; j++) gX[k + j * N2] += p[0] * X2[k + j * N2];
      break;
    default: break;
    }
  }
}

static void mm(int ta, int tb, long M, long N, long K, double alpha, const double* A, long lda, const double* B, long ldb,
               double beta, double* C, long ldc)   /* dgemm */
{
  long i, j, k;
  for(j = 0; j < N; j++)
    for(i = 0; i < M; i++) {
      double s = 0.0;
      for(k = 0; k < K; k++) s += (ta ? A[k + i * lda] : A[i + k * lda]) * (tb ? B[j + k * ldb] : B[k + j * ldb]);
      C[i + j * ldc] = alpha * s + (beta != 0.0 ? beta * C[i + j * ldc] : 0.0);
    }
}

/* per-point diagonal kernel gradient against an N-vector gLambda: CKern::getDiagGradParams (CKern.h:198-213) =
 * getGradParams on every single point with a 1 x 1 covGrad, summed; natural parameters */
static void diag_grad_params(const orc_kspec* ks, const double* X, long N, long D, const double* gl, double* out)
{
  const int nk = ks->offs[ks->n_terms];
  double* t4 = (double*)malloc(sizeof(double) * (nk > 0 ? nk : 1));
  double* xi = (double*)malloc(sizeof(double) * (D > 0 ? D : 1));
  long i, j;
  int k;
  for(k = 0; k < nk; k++) out[k] = 0.0;
  for(i = 0; i < N; i++) {
    for(j = 0; j < D; j++) xi[j] = X[i + j * N];
    orc_kern_grad_sym(ks, xi, 1, D, gl + i, t4);
    for(k = 0; k < nk; k++) out[k] += t4[k];
  }
  free(t4);
  free(xi);
}

/* CGp with approximationType FITC (CGp.cpp:798-856 updateAD, 963-990 logLikelihood, 500-512 updateAlpha, 1320-1399
 * gpCovGrads, 1146-1218 updateG); arguments and gradient layout as orc_gp_dtc.  LcholK (the factor of K_uu) and Lm are
 * used AFTER their trans(), so a reference built from ndlfortran.f feeds single-precision lower triangles into these
 * solves (orc_trans reproduces that unless orc_exact_trans is set). */
double orc_gp_fitc(const orc_kspec* ks, const double* X, long N, long D, const double* m, long d, const double* Xu, long M,
                   double beta, double* g, double* alpha, const double* Xs, long Ns, double* mu, double* var, int* info)
{
  const int nk = ks->offs[ks->n_terms];
  const size_t MM = (size_t)M * M, MN = (size_t)M * N;
  double* Kuu = (double*)malloc(sizeof(double) * MM);
  double* Kuf = (double*)malloc(sizeof(double) * MN);
  double* Luu = (double*)malloc(sizeof(double) * MM);
  double* invKuu = (double*)malloc(sizeof(double) * MM);
  double* A = (double*)malloc(sizeof(double) * MM);
  double* LA = (double*)malloc(sizeof(double) * MM);
  double* Ainv = (double*)malloc(sizeof(double) * MM);
  double* Am = (double*)malloc(sizeof(double) * MM);
  double* Lm = (double*)malloc(sizeof(double) * MM);
  double* V = (double*)malloc(sizeof(double) * MN);
  double* V2 = (double*)malloc(sizeof(double) * MN);
  double* iKK = (double*)malloc(sizeof(double) * MN);
  double* diagD = (double*)malloc(sizeof(double) * N);
  double* sM = (double*)malloc(sizeof(double) * N * d);
  double* bet = (double*)malloc(sizeof(double) * M * d);
  double logDetKuu, logDetA, L = 0.0, tmp;
  long i, j, k, n;
  *info = 0;
  orc_gram_sym(ks, Xu, M, D, Kuu);
  for(n = 0; n < N; n++)
    for(i = 0; i < M; i++) Kuf[i + (size_t)n * M] = orc_kern_element(ks, Xu, M, i, X, N, n, D);
  orc_jitchol(M, Kuu, Luu, 20, info);                                 /* _updateInvK */
  if(*info != 0) goto done;
  logDetKuu = orc_logdet(M, Luu, M);
  (void)logDetKuu;
  orc_pdinv_upper(M, Luu, invKuu);
  orc_trans(M, Luu);                                                  /* LcholK.trans(): lower */
  /* updateAD, FITC */
  mm(0, 0, M, N, M, 1.0, invKuu, M, Kuf, M, 0.0, iKK, M);
  for(n = 0; n < N; n++) {
    double cs = 0.0, dd;
    for(i = 0; i < M; i++) cs += iKK[i + (size_t)n * M] * Kuf[i + (size_t)n * M];
    dd = -orc_kern_diag_element(ks, X, N, n, D);                      /* diagD = diagK; negate; += colsum; *= beta; negate; += 1 */
    dd = cs + dd;
    dd *= beta;
    dd = -dd;
    diagD[n] = dd + 1.0;
  }
  for(n = 0; n < N; n++) {
    const double di = 1 / diagD[n];
    for(i = 0; i < M; i++) V[i + (size_t)n * M] = Kuf[i + (size_t)n * M] * di;
    for(j = 0; j < d; j++) sM[n + (size_t)j * N] = m[n + (size_t)j * N] * sqrt(di);
  }
  memcpy(A, Kuu, sizeof(double) * MM);
  mm(0, 1, M, M, N, 1.0, Kuf, M, V, M, 1.0 / beta, A, M);
  orc_jitchol(M, A, LA, 20, info);
  if(*info != 0) goto done;
  logDetA = orc_logdet(M, LA, M);
  (void)logDetA;
  orc_pdinv_upper(M, LA, Ainv);
  orc_trans(M, LA);
  memcpy(V2, Kuf, sizeof(double) * MN);
  orc_trsm('l', 'l', 'n', 'n', M, N, 1.0, Luu, M, V2, M);
  for(n = 0; n < N; n++) {
    const double sc = 1 / sqrt(diagD[n]);
    for(i = 0; i < M; i++) V2[i + (size_t)n * M] *= sc;
  }
  for(k = 0; k < (long)MM; k++) Am[k] = 0.0;
  for(i = 0; i < M; i++) Am[i + i * M] = 1 / beta;
  mm(0, 1, M, M, N, 1.0, V2, M, V2, M, 1.0, Am, M);
  orc_jitchol(M, Am, Lm, 20, info);
  if(*info != 0) goto done;
  orc_trans(M, Lm);
  orc_trsm('l', 'l', 'n', 'n', M, N, 1.0, Lm, M, V2, M);              /* invLmV */
  mm(0, 0, M, d, N, 1.0, V2, M, sM, N, 0.0, bet, M);
  /* logLikelihood, CGp.cpp:963-990 */
  L += ((double)M - (double)N) * log(beta) + (double)N * 1.8378770664093454836;   /* ndlutil::LOGTWOPI */
  for(n = 0; n < N; n++) L += log(diagD[n]);
  tmp = 0.0;
  for(i = 0; i < M; i++) tmp += log(Lm[i + i * M]);
  L += tmp * 2.0;
  L *= (double)d;
  for(j = 0; j < d; j++)
    L += beta * (dot(N, sM + (size_t)j * N, 1, sM + (size_t)j * N, 1) - dot(M, bet + (size_t)j * M, 1, bet + (size_t)j * M, 1));
  L *= -0.5;
  L -= (double)d * (double)N * HALFLOGTWOPI;
  if(alpha) {                                                         /* updateAlpha, CGp.cpp:500-512 */
    double* s2 = (double*)malloc(sizeof(double) * N * d);
    for(n = 0; n < N; n++)
      for(j = 0; j < d; j++) s2[n + (size_t)j * N] = m[n + (size_t)j * N] * (1 / diagD[n]);
    mm(0, 0, M, d, N, 1.0, Kuf, M, s2, N, 0.0, alpha, M);
    orc_trsm('l', 'l', 'n', 'n', M, d, 1.0, LA, M, alpha, M);
    orc_trsm('l', 'l', 't', 'n', M, d, 1.0, LA, M, alpha, M);
    free(s2);
  }
  if(Xs && Ns > 0) {                                                  /* posterior: the sparse branch of _posteriorVar */
    double* kX = (double*)malloc(sizeof(double) * M * Ns);
    double* W = (double*)malloc(sizeof(double) * MM);
    double* st = (double*)malloc(sizeof(double) * M * Ns);
    for(n = 0; n < Ns; n++)
      for(i = 0; i < M; i++) kX[i + n * M] = orc_kern_element(ks, Xu, M, i, Xs, Ns, n, D);
    for(k = 0; k < (long)MM; k++) W[k] = invKuu[k] - Ainv[k] / beta;
    mm(0, 0, M, Ns, M, 1.0, W, M, kX, M, 0.0, st, M);
    for(n = 0; n < Ns; n++) {
      if(var) var[n] = orc_kern_diag_element(ks, Xs, Ns, n, D) - dot(M, kX + n * M, 1, st + n * M, 1) + 1.0 / beta;
      if(mu && alpha)
        for(j = 0; j < d; j++) mu[n + j * Ns] = dot(M, alpha + (size_t)j * M, 1, kX + n * M, 1);
    }
    free(kX); free(W); free(st);
  }
  if(g) {                                                             /* gpCovGrads FITC, CGp.cpp:1320-1399 */
    double* E = (double*)malloc(sizeof(double) * M * d);
    double* AinvE = (double*)malloc(sizeof(double) * M * d);
    double* EMT = (double*)malloc(sizeof(double) * MN);
    double* AinvEMT = (double*)malloc(sizeof(double) * MN);
    double* AEA = (double*)malloc(sizeof(double) * MM);
    double* Am2 = (double*)malloc(sizeof(double) * MM);
    double* V3 = (double*)malloc(sizeof(double) * MN);
    double* iKKD = (double*)malloc(sizeof(double) * MN);
    double* iKKDQ = (double*)malloc(sizeof(double) * MN);
    double* gKuu = (double*)malloc(sizeof(double) * MM);
    double* gKuf = (double*)malloc(sizeof(double) * MN);
    double* diagQ = (double*)malloc(sizeof(double) * N);
    double* gLambda = (double*)malloc(sizeof(double) * N);
    double* t1 = (double*)malloc(sizeof(double) * (nk > 0 ? nk : 1));
    double* t2 = (double*)malloc(sizeof(double) * (nk > 0 ? nk : 1));
    double* t3 = (double*)malloc(sizeof(double) * (nk > 0 ? nk : 1));
    double* gKX = (double*)malloc(sizeof(double) * M * D);
    double* gKXuf = (double*)malloc(sizeof(double) * N * D);
    double* dg = (double*)malloc(sizeof(double) * M * D);
    double gb;
    mm(0, 0, M, d, N, 1.0, V, M, m, N, 0.0, E, M);                    /* V = K_uf D^-1 (still from updateAD) */
    mm(0, 0, M, d, M, 1.0, Ainv, M, E, M, 0.0, AinvE, M);
    mm(0, 1, M, N, d, 1.0, E, M, m, N, 0.0, EMT, M);
    mm(0, 0, M, N, M, 1.0, Ainv, M, EMT, M, 0.0, AinvEMT, M);
    mm(0, 1, M, M, d, 1.0, AinvE, M, AinvE, M, 0.0, AEA, M);
    for(k = 0; k < (long)MM; k++) Am2[k] = (double)d * Ainv[k] + beta * AEA[k];
    mm(0, 0, M, N, M, 1.0, Am2, M, Kuf, M, 0.0, V3, M);
    for(n = 0; n < N; n++) {
      double kae = 0.0, kda = 0.0, mmt = 0.0;
      for(i = 0; i < M; i++) {
        kae += AinvEMT[i + (size_t)n * M] * Kuf[i + (size_t)n * M];
        kda += V3[i + (size_t)n * M] * Kuf[i + (size_t)n * M];
      }
      for(j = 0; j < d; j++) mmt += m[n + (size_t)j * N] * m[n + (size_t)j * N];
      diagQ[n] = kda - (double)d * diagD[n] + beta * mmt - 2.0 * beta * kae;
    }
    for(n = 0; n < N; n++)
      for(i = 0; i < M; i++) {
        iKKD[i + (size_t)n * M] = iKK[i + (size_t)n * M] * (1 / diagD[n]);
        iKKDQ[i + (size_t)n * M] = iKKD[i + (size_t)n * M] * diagQ[n];
      }
    for(k = 0; k < (long)MM; k++) gKuu[k] = (double)d * (invKuu[k] - Ainv[k] / beta) - AEA[k];
    mm(0, 1, M, M, N, beta, iKKDQ, M, iKKD, M, 1.0, gKuu, M);
    for(k = 0; k < (long)MM; k++) gKuu[k] *= 0.5;
    memcpy(gKuf, iKKDQ, sizeof(double) * MN);
    mm(0, 0, M, N, M, -(double)d, Ainv, M, Kuf, M, -beta, gKuf, M);
    mm(0, 0, M, N, M, -beta, AEA, M, Kuf, M, 1.0, gKuf, M);
    for(k = 0; k < (long)MN; k++) gKuf[k] += beta * AinvEMT[k];
    for(n = 0; n < N; n++)
      for(i = 0; i < M; i++) gKuf[i + (size_t)n * M] *= 1 / diagD[n];
    gb = 0.0;
    for(n = 0; n < N; n++) {
      gLambda[n] = ((diagQ[n] / diagD[n]) * (0.5 * beta)) / diagD[n];
      gb += gLambda[n];
    }
    gb = -gb / (beta * beta);
    orc_kern_grad_sym(ks, Xu, M, D, gKuu, t1);
    orc_grad_to_trans(ks, D, t1);
    orc_kern_grad_cross(ks, Xu, M, X, N, D, gKuf, t2);
    orc_grad_to_trans(ks, D, t2);
    diag_grad_params(ks, X, N, D, gLambda, t3);
    orc_grad_to_trans(ks, D, t3);
    orc_kern_diag_gradx(ks, Xu, M, D, dg);
    for(i = 0; i < M; i++) {
      orc_kern_gradx_row2(ks, Xu, M, i, X, N, D, gKXuf);
      orc_kern_gradx_row2(ks, Xu, M, i, Xu, M, D, gKX);
      for(k = 0; k < M * D; k++) gKX[k] *= 2.0;
      for(j = 0; j < D; j++) gKX[i + j * M] = dg[i + j * M];
      for(j = 0; j < D; j++) {
        double s = dot(M, gKX + (size_t)j * M, 1, gKuu + (size_t)i * M, 1);
        s += dot(N, gKXuf + (size_t)j * N, 1, gKuf + i, M);
        g[i + j * M] = s;
      }
    }
    for(k = 0; k < nk; k++) g[M * D + k] = t1[k] + t2[k] + t3[k];
    g[M * D + nk] = gb * beta;
    free(E); free(AinvE); free(EMT); free(AinvEMT); free(AEA); free(Am2); free(V3); free(iKKD); free(iKKDQ); free(gKuu);
    free(gKuf); free(diagQ); free(gLambda); free(t1); free(t2); free(t3); free(gKX); free(gKXuf); free(dg);
  }
done:
  free(Kuu); free(Kuf); free(Luu); free(invKuu); free(A); free(LA); free(Ainv); free(Am); free(Lm); free(V); free(V2);
  free(iKK); free(diagD); free(sM); free(bet);
  return L;
}

/* CGp with approximationType DTC: updateK (CGp.cpp:713-735), _updateInvK (896-909), updateAD (751-776), logLikelihood
 * (939-961, 1002-1013), gpCovGrads (1252-1316), updateG (1146-1190), logLikelihoodGradient (1016-1079), updateAlpha
 * (490-497), _posteriorMean / _posteriorVar (548-599).
 *   X N x D, m N x d (centred / scaled targets), Xu M x D inducing inputs, beta = noise precision.
 *   g (may be NULL): [d/dX_u column by column (M*D)] [kernel, transformed (nk)] [d/d log beta]  (CGp.cpp:330-385)
 *   alpha (may be NULL) M x d;  Xs (may be NULL) Ns x D -> mu Ns x d, var Ns.
 * Returns the log-likelihood; *info != 0 if a Cholesky failed even with jitter. */
double orc_gp_dtc(const orc_kspec* ks, const double* X, long N, long D, const double* m, long d, const double* Xu, long M,
                  double beta, int dtcvar, double* g, double* alpha, const double* Xs, long Ns, double* mu, double* var,
                  int* info)
{
  const int nk = ks->offs[ks->n_terms];
  const size_t MM = (size_t)M * M, MN = (size_t)M * N;
  double* Kuu = (double*)malloc(sizeof(double) * MM);
  double* Kuf = (double*)malloc(sizeof(double) * MN);
  double* U = (double*)malloc(sizeof(double) * MM);
  double* invKuu = (double*)malloc(sizeof(double) * MM);
  double* A = (double*)malloc(sizeof(double) * MM);
  double* LA = (double*)malloc(sizeof(double) * MM);
  double* Ainv = (double*)malloc(sizeof(double) * MM);
  double* e = (double*)malloc(sizeof(double) * M);
  double* invAe = (double*)malloc(sizeof(double) * M);
  double logDetKuu, logDetA, L = 0.0, sumDiagD = 0.0;
  double* iKK = NULL;   /* invK_uu K_uf (DTCVAR) */
  long i, j, k, n;
  *info = 0;
  orc_gram_sym(ks, Xu, M, D, Kuu);                                   /* K_uu */
  for(n = 0; n < N; n++)                                             /* K_uf(i,n) = k(xu_i, x_n) */
    for(i = 0; i < M; i++) Kuf[i + (size_t)n * M] = orc_kern_element(ks, Xu, M, i, X, N, n, D);
  orc_jitchol(M, Kuu, U, 20, info);                                  /* _updateInvK */
  if(*info != 0) goto done;
  logDetKuu = orc_logdet(M, U, M);
  orc_pdinv_upper(M, U, invKuu);
  memcpy(A, Kuu, sizeof(double) * MM);                               /* updateAD: A = K_uf K_uf' + K_uu / beta */
  mm(0, 1, M, M, N, 1.0, Kuf, M, Kuf, M, 1.0 / beta, A, M);
  orc_jitchol(M, A, LA, 20, info);
  if(*info != 0) goto done;
  logDetA = orc_logdet(M, LA, M);
  orc_pdinv_upper(M, LA, Ainv);
  orc_trans(M, LA);                                                  /* LcholA.trans(): lower (fp32 quirk included) */
  if(dtcvar) {
    /* DTCVAR, CGp.cpp:766-774: V = (invK_uu K_uf) .* K_uf; diagD = beta (diagK - column sums of V) */
    iKK = (double*)malloc(sizeof(double) * MN);
    mm(0, 0, M, N, M, 1.0, invKuu, M, Kuf, M, 0.0, iKK, M);
    for(n = 0; n < N; n++) {
      double cs = 0.0;
      for(i = 0; i < M; i++) cs += iKK[i + (size_t)n * M] * Kuf[i + (size_t)n * M];
      sumDiagD += beta * (orc_kern_diag_element(ks, X, N, n, D) - cs);
    }
  }
  L += (double)d * (((double)M - (double)N) * log(beta) - logDetKuu + logDetA);
  for(j = 0; j < d; j++) {
    mm(0, 0, M, 1, N, 1.0, Kuf, M, m + (size_t)j * N, N, 0.0, e, M);
    orc_symv_upper(M, Ainv, e, invAe);                               /* Ainv is fully symmetric */
    L -= beta * (dot(M, invAe, 1, e, 1) - dot(N, m + (size_t)j * N, 1, m + (size_t)j * N, 1));
  }
  if(dtcvar) L += (double)d * sumDiagD;                               /* CGp.cpp:955-956 */
  L *= -0.5;
  L -= (double)d * (double)N * HALFLOGTWOPI;
  if(alpha) {                                                        /* updateAlpha */
    mm(0, 0, M, d, N, 1.0, Kuf, M, m, N, 0.0, alpha, M);
    orc_trsm('l', 'l', 'n', 'n', M, d, 1.0, LA, M, alpha, M);
    orc_trsm('l', 'l', 't', 'n', M, d, 1.0, LA, M, alpha, M);
  }
  if(Xs && Ns > 0) {                                                 /* posteriorMeanVar */
    double* kX = (double*)malloc(sizeof(double) * M * Ns);
    double* W = (double*)malloc(sizeof(double) * MM);
    double* st = (double*)malloc(sizeof(double) * M * Ns);
    for(n = 0; n < Ns; n++)
      for(i = 0; i < M; i++) kX[i + n * M] = orc_kern_element(ks, Xu, M, i, Xs, Ns, n, D);
    for(k = 0; k < (long)MM; k++) W[k] = invKuu[k] - Ainv[k] / beta;
    mm(0, 0, M, Ns, M, 1.0, W, M, kX, M, 0.0, st, M);
    for(n = 0; n < Ns; n++) {
      if(var) var[n] = orc_kern_diag_element(ks, Xs, Ns, n, D) - dot(M, kX + n * M, 1, st + n * M, 1) + 1.0 / beta;
      if(mu && alpha)
        for(j = 0; j < d; j++) mu[n + j * Ns] = dot(M, alpha + (size_t)j * M, 1, kX + n * M, 1);
    }
    free(kX); free(W); free(st);
  }
  if(g) {                                                            /* gpCovGrads + updateG */
    double* E = (double*)malloc(sizeof(double) * M * d);
    double* EET = (double*)malloc(sizeof(double) * MM);
    double* AinvEET = (double*)malloc(sizeof(double) * MM);
    double* AEA = (double*)malloc(sizeof(double) * MM);
    double* gKuu = (double*)malloc(sizeof(double) * MM);
    double* AinvKuf = (double*)malloc(sizeof(double) * MN);
    double* EMT = (double*)malloc(sizeof(double) * MN);
    double* AinvEMT = (double*)malloc(sizeof(double) * MN);
    double* gKuf = (double*)malloc(sizeof(double) * MN);
    double* t1 = (double*)malloc(sizeof(double) * (nk > 0 ? nk : 1));
    double* t2 = (double*)malloc(sizeof(double) * (nk > 0 ? nk : 1));
    double* gKX = (double*)malloc(sizeof(double) * M * D);
    double* gKXuf = (double*)malloc(sizeof(double) * N * D);
    double* dg = (double*)malloc(sizeof(double) * M * D);
    double gb, tmp;
    mm(0, 0, M, d, N, 1.0, Kuf, M, m, N, 0.0, E, M);
    mm(0, 1, M, M, d, 1.0, E, M, E, M, 0.0, EET, M);
    mm(0, 0, M, M, M, 1.0, Ainv, M, EET, M, 0.0, AinvEET, M);
    mm(0, 0, M, M, M, 1.0, AinvEET, M, Ainv, M, 0.0, AEA, M);
    for(k = 0; k < (long)MM; k++) gKuu[k] = (double)d * (invKuu[k] - Ainv[k] / beta) - AEA[k];
    if(dtcvar) mm(0, 1, M, M, N, -beta * (double)d, iKK, M, iKK, M, 1.0, gKuu, M);   /* syrk, CGp.cpp:1275-1279 */
    for(k = 0; k < (long)MM; k++) gKuu[k] *= 0.5;
    mm(0, 0, M, N, M, 1.0, Ainv, M, Kuf, M, 0.0, AinvKuf, M);
    mm(0, 1, M, N, d, 1.0, E, M, m, N, 0.0, EMT, M);
    mm(0, 0, M, N, M, 1.0, Ainv, M, EMT, M, 0.0, AinvEMT, M);
    mm(0, 0, M, N, M, 1.0, AinvEET, M, AinvKuf, M, 0.0, gKuf, M);
    for(k = 0; k < (long)MN; k++) gKuf[k] = -(beta * (gKuf[k] - AinvEMT[k])) - (double)d * AinvKuf[k];
    if(dtcvar)
      for(k = 0; k < (long)MN; k++) gKuf[k] += beta * (double)d * iKK[k];                /* CGp.cpp:1292-1295 */
    gb = (double)(N - M) / beta;
    tmp = 0.0;
    for(k = 0; k < (long)MM; k++) tmp += Ainv[k] * Kuu[k];
    gb += tmp / (beta * beta);
    gb *= (double)d;
    tmp = 0.0;
    for(k = 0; k < (long)MM; k++) tmp += AEA[k] * Kuu[k];
    gb += tmp / beta;
    for(j = 0; j < d; j++) gb -= dot(N, m + (size_t)j * N, 1, m + (size_t)j * N, 1);
    for(i = 0; i < M; i++) gb += AinvEET[i + i * M];
    if(dtcvar) gb -= (double)d * sumDiagD / beta;                                          /* CGp.cpp:1309-1312 */
    gb *= 0.5;
    /* kernel parameters: symmetric pass on X_u against gK_uu + cross pass (X_u, X) against gK_uf, each transformed */
    orc_kern_grad_sym(ks, Xu, M, D, gKuu, t1);
    orc_grad_to_trans(ks, D, t1);
    orc_kern_grad_cross(ks, Xu, M, X, N, D, gKuf, t2);
    orc_grad_to_trans(ks, D, t2);
    if(dtcvar) {
      /* the diagonal term's effect on the kernel parameters: gLambda = -0.5 d beta for every point (CGp.cpp:1314-1317),
       * CKern::getDiagGradParams (CKern.h:198-213) = getGradParams on each single point, summed; then transformed */
      double* t3 = (double*)malloc(sizeof(double) * (nk > 0 ? nk : 1));
      double* t4 = (double*)malloc(sizeof(double) * (nk > 0 ? nk : 1));
      double* xi = (double*)malloc(sizeof(double) * (D > 0 ? D : 1));
      const double gl = -0.5 * (double)d * beta;
      for(k = 0; k < nk; k++) t3[k] = 0.0;
      for(i = 0; i < N; i++) {
        for(j = 0; j < D; j++) xi[j] = X[i + j * N];
        orc_kern_grad_sym(ks, xi, 1, D, &gl, t4);
        for(k = 0; k < nk; k++) t3[k] += t4[k];
      }
      orc_grad_to_trans(ks, D, t3);
      for(k = 0; k < nk; k++) t2[k] += t3[k];
      free(t3); free(t4); free(xi);
    }
    /* d/dX_u (CGp.cpp:1160-1182) */
    orc_kern_diag_gradx(ks, Xu, M, D, dg);
    for(i = 0; i < M; i++) {
      orc_kern_gradx_row2(ks, Xu, M, i, X, N, D, gKXuf);
      orc_kern_gradx_row2(ks, Xu, M, i, Xu, M, D, gKX);
      for(k = 0; k < M * D; k++) gKX[k] *= 2.0;
      for(j = 0; j < D; j++) gKX[i + j * M] = dg[i + j * M];
      for(j = 0; j < D; j++) {
        double s = dot(M, gKX + (size_t)j * M, 1, gKuu + (size_t)i * M, 1);
        s += dot(N, gKXuf + (size_t)j * N, 1, gKuf + i, M);   /* dotColRow(j, gK_uf, i) */
        g[i + j * M] = s;
      }
    }
    for(k = 0; k < nk; k++) g[M * D + k] = t1[k] + t2[k];
    g[M * D + nk] = gb * beta;                                       /* exp transform: gradfact(beta) = beta */
    free(E); free(EET); free(AinvEET); free(AEA); free(gKuu); free(AinvKuf); free(EMT); free(AinvEMT); free(gKuf);
    free(t1); free(t2); free(gKX); free(gKXuf); free(dg);
  }
done:
  free(iKK);
  free(Kuu); free(Kuf); free(U); free(invKuu); free(A); free(LA); free(Ainv); free(e); free(invAe);
  return L;
}
