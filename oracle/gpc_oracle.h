/* gpc_oracle.h -- plain-C CPU restatement of GPc's exact-GP (FTC) hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity checker of the build (tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it); nothing under gpc_amd/ includes, links, loads or executes it, and
 * the product path has no CPU fallback.
 *
 * Parity status: PINNED -- checked (tests/test_oracle_golden.py) against the reference's own fixtures
 * (rbf/rbfard/white/bias/lin KernTest.mat, choleskyMatrixTest.mat, trsmMatrixTest.mat, testGpftc.mat) and against
 * outputs of the unmodified reference compiled here (oracle/_ref, golden vectors under tests/golden/); the GP-LVM and
 * DTC restatements against the compiled reference's CGplvm / CGp(DTC) (tests/test_gplvm.py, tests/test_dtc.py).
 *
 * Every function cites the reference code it follows (file:line under /root/reference).  The arithmetic below
 * lapack.h lives in a third-party, un-vendored, unpinned BLAS/LAPACK (make.linux:8 `-llapack -lblas`; MKL 2021.4
 * from /opt/conda in the authoring container); its routines are restated here from the published reference-LAPACK
 * algorithms (dpotf2/dpotrf, dtrtri+dlauum = dpotri, dtrsm, dsymv, dsyr, dnrm2).
 * All matrices are column-major double, like CMatrix (CMatrix.h:30, 255-269).
 */
#ifndef GPC_ORACLE_H
#define GPC_ORACLE_H
#include <stddef.h>

#define ORC_KERN_RBF 1
#define ORC_KERN_RBFARD 2
#define ORC_KERN_WHITE 3
#define ORC_KERN_BIAS 4
#define ORC_KERN_LIN 5
#define ORC_MAX_TERMS 16
#define ORC_MAX_PARAMS 160

/* same layout as struct gpc_kspec (include/gpc_hip.h) */
typedef struct orc_kspec {
  int n_terms;
  int types[ORC_MAX_TERMS];
  int offs[ORC_MAX_TERMS + 1];
  double params[ORC_MAX_PARAMS];
} orc_kspec;

/* ---- CMatrix pieces ------------------------------------------------------------------------------------------- */
double orc_dist2_row(const double* X1, long ld1, long i, const double* X2, long ld2, long k, long D);
int orc_potrf(char uplo, long N, double* A, long lda);              /* returns LAPACK info */
int orc_chol(char uplo, long N, double* A, long lda);               /* potrf + zero the other triangle */
double orc_jitchol(long N, double* A, double* U, int max_tries, int* info);   /* CMatrix::jitChol */
double orc_logdet(long N, const double* U, long ldu);
void orc_pdinv_upper(long N, const double* U, double* invA);        /* CMatrix::pdinv(U) */
void orc_trans(long N, double* A);                                  /* CMatrix::trans (square) */
extern int orc_exact_trans;   /* 0 (default): single-precision swap temporary of ndlfortran.f; 1: exact (f2c twin) */
void orc_trsm(char side, char uplo, char trans, char diag, long M, long N, double alpha, const double* A,
              long lda, double* B, long ldb);
void orc_symv_upper(long N, const double* A, const double* x, double* y);   /* y = A x, A symmetric (upper read) */

/* ---- CKern pieces --------------------------------------------------------------------------------------------- */
double orc_kern_element(const orc_kspec* ks, const double* X1, long ld1, long i, const double* X2, long ld2,
                        long j, long D);
double orc_kern_diag_element(const orc_kspec* ks, const double* X, long ld, long i, long D);
void orc_gram_sym(const orc_kspec* ks, const double* X, long N, long D, double* K);
void orc_gram_cross(const orc_kspec* ks, const double* X, long N, const double* X2, long N2, long D, double* K);
void orc_gram_diag(const orc_kspec* ks, const double* X, long N, long D, double* d);
/* natural-space gradients, then the transform chain rule (CKern::getGradTransParams, CKern.cpp:50-63) */
void orc_kern_grad_sym(const orc_kspec* ks, const double* X, long N, long D, const double* covGrad, double* g);
void orc_kern_grad_cross(const orc_kspec* ks, const double* X, long N, const double* X2, long N2, long D,
                         const double* covGrad, double* g);
void orc_grad_to_trans(const orc_kspec* ks, long D, double* g);
void orc_trans_params(const orc_kspec* ks, long D, double* a);

/* ---- CGp (FTC) ------------------------------------------------------------------------------------------------- */
typedef struct orc_gp {
  long N, D, d;
  orc_kspec ks;
  const double* X;   /* N x D */
  double* m;         /* N x d : (y - bias)/scale */
  double* scale;     /* d */
  double* bias;      /* d */
  double* K;         /* N x N */
  double* L;         /* N x N lower Cholesky factor (after trans) */
  double* invK;      /* N x N */
  double* Alpha;     /* N x d */
  double* covGrad;   /* N x N */
  double logDetK;
  double jitter;
  int info;
} orc_gp;

orc_gp* orc_gp_create(const orc_kspec* ks, const double* X, long N, long D, const double* y, long d,
                      const double* scale, const double* bias);
void orc_gp_free(orc_gp* gp);
int orc_gp_update_k(orc_gp* gp);                      /* CGp::updateK: _updateK + _updateInvK */
void orc_gp_update_alpha(orc_gp* gp);                 /* CGp::updateAlpha */
double orc_gp_loglik(orc_gp* gp);                     /* CGp::logLikelihood */
double orc_gp_loglik_grad(orc_gp* gp, double* g);     /* CGp::logLikelihoodGradient (transformed kernel params) */
void orc_gp_posterior(orc_gp* gp, const double* Xs, long Ns, double* mu, double* var); /* posteriorMeanVar */

/* ---- CGplvm (plain model: no dynamics / back constraints / learnt scales) ----------------------------------------- */
void orc_kern_gradx_row(const orc_kspec* ks, const double* X, long N, long D, long row, double* gX);
void orc_kern_diag_gradx(const orc_kspec* ks, const double* X, long N, long D, double* gD);
double orc_gplvm_loglik_grad(const orc_kspec* ks, const double* m, long N, long d, const double* X, long q,
                             int regularise, double* g, double* logdet_out, int* info);

/* ---- CGp, approximation type DTC -------------------------------------------------------------------------------- */
void orc_kern_gradx_row2(const orc_kspec* ks, const double* X, long ldx, long row, const double* X2, long N2, long D,
                         double* gX);
double orc_gp_dtc(const orc_kspec* ks, const double* X, long N, long D, const double* m, long d, const double* Xu, long M,
                  double beta, int dtcvar, double* g, double* alpha, const double* Xs, long Ns, double* mu, double* var,
                  int* info);   /* dtcvar != 0: the DTCVAR variant (extra diagonal terms, CGp.cpp:766-774, 955, 1275-1317) */
double orc_gp_fitc(const orc_kspec* ks, const double* X, long N, long D, const double* m, long d, const double* Xu, long M,
                   double beta, double* g, double* alpha, const double* Xs, long Ns, double* mu, double* var, int* info);
#endif
