/* oracle_driver.c -- command-line front end of the plain-C restatement (gpc_oracle.c), speaking the same GPCB1
 * container and the same modes as oracle/ref_driver.cpp so that tests can run both side by side.
 *
 * TEST INFRASTRUCTURE ONLY.  Usage: oracle_driver <kern|gp|time|chol|trsm|gplvm|dtc> <in.gpcb> <out.gpcb>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include "gpc_oracle.h"
#include "gpcb_io.h"

static double now_s(void)
{
  struct timeval tv;
  gettimeofday(&tv, 0);
  return tv.tv_sec + 1e-6 * tv.tv_usec;
}

static void build_kspec(const gpcb_file* in, long D, orc_kspec* ks)
{
  const gpcb_array* types = gpcb_need(in, "kern_types");
  const gpcb_array* params = gpcb_need(in, "kern_params");
  long t, off = 0, q;
  memset(ks, 0, sizeof(*ks));
  ks->n_terms = (int)(types->rows * types->cols);
  for(t = 0; t < ks->n_terms; t++) {
    long np;
    ks->types[t] = (int)types->data[t];
    np = ks->types[t] == ORC_KERN_RBF ? 2 : ks->types[t] == ORC_KERN_RBFARD ? 2 + D : 1;
    ks->offs[t] = (int)off;
    for(q = 0; q < np; q++) ks->params[off + q] = params->data[off + q];
    off += np;
  }
  ks->offs[ks->n_terms] = (int)off;
  if(off != params->rows * params->cols) {
    fprintf(stderr, "oracle_driver: expected %ld kernel params\n", off);
    exit(2);
  }
}

static int run_kern(const gpcb_file* in, const char* out)
{
  const gpcb_array *X = gpcb_need(in, "X"), *X2 = gpcb_need(in, "X2"), *cg = gpcb_need(in, "covGrad"),
                   *cg2 = gpcb_need(in, "covGrad2");
  const long N = X->rows, D = X->cols, N2 = X2->rows;
  orc_kspec ks;
  int np;
  double *K2, *K4, *k2, *g2, *g4, *tp;
  FILE* fp;
  build_kspec(in, D, &ks);
  np = ks.offs[ks.n_terms];
  K2 = malloc(sizeof(double) * N * N);
  K4 = malloc(sizeof(double) * N * N2);
  k2 = malloc(sizeof(double) * N);
  g2 = malloc(sizeof(double) * np);
  g4 = malloc(sizeof(double) * np);
  tp = malloc(sizeof(double) * np);
  orc_gram_sym(&ks, X->data, N, D, K2);
  orc_gram_cross(&ks, X->data, N, X2->data, N2, D, K4);
  orc_gram_diag(&ks, X->data, N, D, k2);
  orc_kern_grad_sym(&ks, X->data, N, D, cg->data, g2);
  orc_grad_to_trans(&ks, D, g2);
  orc_kern_grad_cross(&ks, X->data, N, X2->data, N2, D, cg2->data, g4);
  orc_grad_to_trans(&ks, D, g4);
  orc_trans_params(&ks, D, tp);
  fp = gpcb_open_write(out);
  gpcb_write(fp, "K2", N, N, K2);
  gpcb_write(fp, "K4", N, N2, K4);
  gpcb_write(fp, "k2", N, 1, k2);
  gpcb_write(fp, "g2", 1, np, g2);
  gpcb_write(fp, "g4", 1, np, g4);
  gpcb_write(fp, "trans_params", 1, np, tp);
  gpcb_write(fp, "nat_params", 1, np, ks.params);
  fclose(fp);
  return 0;
}

static int run_gp(const gpcb_file* in, const char* out)
{
  const gpcb_array *X = gpcb_need(in, "X"), *y = gpcb_need(in, "y");
  const gpcb_array *xs = gpcb_find(in, "Xstar"), *sc = gpcb_find(in, "scale"), *bi = gpcb_find(in, "bias"),
                   *dump = gpcb_find(in, "dump_matrices"), *ex = gpcb_find(in, "exact_trans");
  const long N = X->rows, D = X->cols, d = y->cols;
  orc_kspec ks;
  orc_gp* gp;
  int np;
  double *g, *tp, ll, t0, t1;
  FILE* fp;
  build_kspec(in, D, &ks);
  np = ks.offs[ks.n_terms];
  orc_exact_trans = (ex && ex->data[0] != 0.0) ? 1 : 0;
  gp = orc_gp_create(&ks, X->data, N, D, y->data, d, sc ? sc->data : NULL, bi ? bi->data : NULL);
  t0 = now_s();
  if(orc_gp_update_k(gp) != 0) {
    fprintf(stderr, "oracle_driver: MatrixNonPosDef\n");
    return 3;
  }
  g = malloc(sizeof(double) * (np > 0 ? np : 1));
  tp = malloc(sizeof(double) * (np > 0 ? np : 1));
  ll = orc_gp_loglik_grad(gp, g);
  t1 = now_s();
  orc_gp_update_alpha(gp);
  orc_trans_params(&ks, D, tp);
  fp = gpcb_open_write(out);
  gpcb_write_scalar(fp, "ll", ll);
  gpcb_write_scalar(fp, "logdet", gp->logDetK);
  gpcb_write_scalar(fp, "jitter", gp->jitter);
  gpcb_write_scalar(fp, "t_llgrad", t1 - t0);
  gpcb_write(fp, "grads", 1, np, g);
  gpcb_write(fp, "opt_params", 1, np, tp);
  gpcb_write(fp, "alpha", N, d, gp->Alpha);
  gpcb_write(fp, "m", N, d, gp->m);
  if(dump && dump->data[0] != 0.0) {
    orc_gram_sym(&ks, X->data, N, D, gp->K);
    gpcb_write(fp, "K", N, N, gp->K);
    gpcb_write(fp, "L", N, N, gp->L);
    gpcb_write(fp, "invK", N, N, gp->invK);
  }
  if(xs) {
    const long Ns = xs->rows;
    double* mu = malloc(sizeof(double) * Ns * d);
    double* var = malloc(sizeof(double) * Ns * d);
    orc_gp_posterior(gp, xs->data, Ns, mu, var);
    gpcb_write(fp, "mu", Ns, d, mu);
    gpcb_write(fp, "var", Ns, d, var);
  }
  fclose(fp);
  orc_gp_free(gp);
  return 0;
}

/* one CGp::updateK()'s two phases (Gram build, jitChol + logDet), single thread, minimum over reps */
static int run_time(const gpcb_file* in, const char* out)
{
  const gpcb_array* X = gpcb_need(in, "X");
  const gpcb_array* ra = gpcb_find(in, "reps");
  const long N = X->rows, D = X->cols;
  const int reps = ra ? (int)ra->data[0] : 1;
  orc_kspec ks;
  double *K, *U, tg = 1e300, tc = 1e300, logdet = 0.0, jit = 0.0;
  int r, info = 0;
  FILE* fp;
  build_kspec(in, D, &ks);
  K = malloc(sizeof(double) * N * N);
  U = malloc(sizeof(double) * N * N);
  for(r = 0; r < reps; r++) {
    const double t0 = now_s();
    double t1, t2;
    orc_gram_sym(&ks, X->data, N, D, K);
    t1 = now_s();
    jit = orc_jitchol(N, K, U, 20, &info);
    logdet = orc_logdet(N, U, N);
    t2 = now_s();
    if(t1 - t0 < tg) tg = t1 - t0;
    if(t2 - t1 < tc) tc = t2 - t1;
  }
  fp = gpcb_open_write(out);
  gpcb_write_scalar(fp, "t_gram", tg);
  gpcb_write_scalar(fp, "t_chol", tc);
  gpcb_write_scalar(fp, "logdet", logdet);
  gpcb_write_scalar(fp, "jitter", jit);
  gpcb_write_scalar(fp, "info", (double)info);
  fclose(fp);
  return 0;
}

/* chol: C, uplo (1 = upper, 0 = lower) -> F ; trsm: A, B, flags (side L=1, upper=1, trans=1, unit=1), alpha -> X */
static int run_chol(const gpcb_file* in, const char* out)
{
  const gpcb_array* C = gpcb_need(in, "C");
  const int upper = gpcb_need(in, "upper")->data[0] != 0.0;
  const long N = C->rows;
  double* F = malloc(sizeof(double) * N * N);
  int info;
  FILE* fp;
  memcpy(F, C->data, sizeof(double) * N * N);
  info = orc_chol(upper ? 'U' : 'L', N, F, N);
  fp = gpcb_open_write(out);
  gpcb_write(fp, "F", N, N, F);
  gpcb_write_scalar(fp, "info", (double)info);
  fclose(fp);
  return 0;
}

static int run_trsm(const gpcb_file* in, const char* out)
{
  const gpcb_array *A = gpcb_need(in, "A"), *B = gpcb_need(in, "B"), *fl = gpcb_need(in, "flags");
  const double alpha = gpcb_need(in, "alpha")->data[0];
  double* X = malloc(sizeof(double) * B->rows * B->cols);
  FILE* fp;
  memcpy(X, B->data, sizeof(double) * B->rows * B->cols);
  orc_trsm(fl->data[0] != 0.0 ? 'L' : 'R', fl->data[1] != 0.0 ? 'U' : 'L', fl->data[2] != 0.0 ? 'T' : 'N',
           fl->data[3] != 0.0 ? 'U' : 'N', B->rows, B->cols, alpha, A->data, A->rows, X, B->rows);
  fp = gpcb_open_write(out);
  gpcb_write(fp, "X", B->rows, B->cols, X);
  fclose(fp);
  return 0;
}

/* same inputs as ref_driver's gplvm mode: Y (N x d, centred here as CScaleNoise does with bias = meanCol(Y), scale 1),
 * X (N x q), kernel, optional regularise flag -> ll, g, logdet, m */
static int run_gplvm(const gpcb_file* in, const char* out)
{
  const gpcb_array *Y = gpcb_need(in, "Y"), *X = gpcb_need(in, "X"), *reg = gpcb_find(in, "regularise");
  const long N = Y->rows, d = Y->cols, q = X->cols;
  orc_kspec ks;
  int nk, info = 0;
  long i, j;
  double *m, *g, ll, logdet = 0.0, infod;
  FILE* fp;
  build_kspec(in, q, &ks);
  nk = ks.offs[ks.n_terms];
  m = malloc(sizeof(double) * N * d);
  g = malloc(sizeof(double) * (nk + N * q));
  for(j = 0; j < d; j++) {
    double mean = 0.0;   /* meanCol: sum / nrows (CMatrix.cpp sumCol then scale) */
    for(i = 0; i < N; i++) mean += Y->data[i + j * N];
    mean /= (double)N;
    for(i = 0; i < N; i++) m[i + j * N] = (Y->data[i + j * N] - mean) / 1.0;
  }
  ll = orc_gplvm_loglik_grad(&ks, m, N, d, X->data, q, reg ? reg->data[0] != 0.0 : 1, g, &logdet, &info);
  infod = (double)info;
  fp = gpcb_open_write(out);
  gpcb_write(fp, "ll", 1, 1, &ll);
  gpcb_write(fp, "g", 1, nk + N * q, g);
  gpcb_write(fp, "logdet", 1, 1, &logdet);
  gpcb_write(fp, "info", 1, 1, &infod);
  gpcb_write(fp, "m", N, d, m);
  fclose(fp);
  free(m);
  free(g);
  return 0;
}

/* same inputs as ref_driver's dtc mode: X, y (centred here with bias = meanCol(y), scale 1), X_u, beta, optional Xstar */
static int run_dtc(const gpcb_file* in, const char* out)
{
  const gpcb_array *X = gpcb_need(in, "X"), *y = gpcb_need(in, "y"), *Xu = gpcb_need(in, "X_u"),
                   *xs = gpcb_find(in, "Xstar"), *ex = gpcb_find(in, "exact_trans"), *ap = gpcb_find(in, "approx");
  const double beta = gpcb_need(in, "beta")->data[0];
  const long N = X->rows, D = X->cols, d = y->cols, M = Xu->rows, Ns = xs ? xs->rows : 0;
  orc_kspec ks;
  int nk, info = 0;
  long i, j;
  double *m, *g, *alpha, *mu = 0, *var = 0, *means, ll, infod;
  FILE* fp;
  build_kspec(in, D, &ks);
  nk = ks.offs[ks.n_terms];
  orc_exact_trans = (ex && ex->data[0] != 0.0) ? 1 : 0;
  m = malloc(sizeof(double) * N * d);
  means = malloc(sizeof(double) * d);
  for(j = 0; j < d; j++) {
    double mean = 0.0;
    for(i = 0; i < N; i++) mean += y->data[i + j * N];
    mean /= (double)N;
    means[j] = mean;
    for(i = 0; i < N; i++) m[i + j * N] = (y->data[i + j * N] - mean) * (1 / 1.0);
  }
  g = malloc(sizeof(double) * (M * D + nk + 1));
  alpha = malloc(sizeof(double) * M * d);
  if(Ns) {
    mu = malloc(sizeof(double) * Ns * d);
    var = malloc(sizeof(double) * Ns);
  }
  if(ap && ap->data[0] == 2.0)   /* CGp::FITC */
    ll = orc_gp_fitc(&ks, X->data, N, D, m, d, Xu->data, M, beta, g, alpha, xs ? xs->data : 0, Ns, mu, var, &info);
  else
    ll = orc_gp_dtc(&ks, X->data, N, D, m, d, Xu->data, M, beta, (ap && ap->data[0] == 4.0) ? 1 : 0, g, alpha,
                    xs ? xs->data : 0, Ns, mu, var, &info);
  for(j = 0; j < d && Ns; j++)   /* _posteriorMean adds the output bias (CGp.cpp:566-573) */
    for(i = 0; i < Ns; i++) mu[i + j * Ns] += means[j];
  infod = (double)info;
  fp = gpcb_open_write(out);
  gpcb_write(fp, "ll", 1, 1, &ll);
  gpcb_write(fp, "grads", 1, M * D + nk + 1, g);
  gpcb_write(fp, "alpha", M, d, alpha);
  gpcb_write(fp, "info", 1, 1, &infod);
  if(Ns) {
    gpcb_write(fp, "mu", Ns, d, mu);
    gpcb_write(fp, "var", Ns, 1, var);
  }
  fclose(fp);
  return 0;
}

int main(int argc, char** argv)
{
  gpcb_file in;
  if(argc != 4) {
    fprintf(stderr, "usage: oracle_driver <kern|gp|time|chol|trsm|gplvm|dtc> <in.gpcb> <out.gpcb>\n");
    return 2;
  }
  if(gpcb_read(argv[2], &in) != 0) {
    fprintf(stderr, "oracle_driver: cannot read %s\n", argv[2]);
    return 2;
  }
  if(strcmp(argv[1], "kern") == 0) return run_kern(&in, argv[3]);
  if(strcmp(argv[1], "gp") == 0) return run_gp(&in, argv[3]);
  if(strcmp(argv[1], "time") == 0) return run_time(&in, argv[3]);
  if(strcmp(argv[1], "chol") == 0) return run_chol(&in, argv[3]);
  if(strcmp(argv[1], "trsm") == 0) return run_trsm(&in, argv[3]);
  if(strcmp(argv[1], "gplvm") == 0) return run_gplvm(&in, argv[3]);
  if(strcmp(argv[1], "dtc") == 0) return run_dtc(&in, argv[3]);
  fprintf(stderr, "oracle_driver: unknown mode %s\n", argv[1]);
  return 2;
}
