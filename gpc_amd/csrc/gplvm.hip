// gplvm.hip -- the two extra kernels the GP-LVM objective needs on top of the exact-GP path.
//
//  gpc_covgrad_multi_f64   G = sum_j covGrad_j = -0.5 * (d * invK - A A'),  A = invK * m  (N x d)
//      CGplvm::updateCovGradient (CGplvm.cpp:365-378) is called once per output dimension j and followed, each time,
//      by a kernel-parameter pass and a dL/dX pass (CGplvm.cpp:585-604).  Both passes are linear in covGrad, so the
//      d matrices are summed first: one N^2 pass instead of d (HBM: read invK, write G).
//  gpc_kern_gradx_f64      gX(i,q) = sum_n 2 G(n,i) dk(x_i,x_n)/dx_iq   (n != i)   +   G(i,i) dk(x_i,x_i)/dx_iq
//      Replaces CCmpndKern::getGradX (CKern.cpp:184-193; N matrices of N x q, components CRbfKern 1115-1135,
//      CRbfardKern 3268-3293, CLinKern 2291-2308, white/bias: nothing), the x2 / diagonal fix-up of
//      CGplvm.cpp:573-584 and the N q d dotColCol calls of CGplvm.cpp:597-603 by ONE pass over G (HBM-read bound,
//      8 N^2 bytes; nothing of size N^2 q is ever formed).  The only kernel with a diagonal derivative is the linear one
//      (CLinKern::getDiagGradX, CKern.cpp:2310-2322: 2 variance x_i), which is what the n == i term of the sum gives
//      anyway, so the diagonal needs no special case.
//
// Thread layout of the dX kernel: a workgroup owns 64 rows i (lane = row) and a slice of the columns n; its 4 waves
// take every 4th column of the slice, so x_n is wave-uniform (broadcast loads) and G(i,n) = G(n,i) is read coalesced
// along i.  Column slices write partial sums that a second tiny kernel adds in a fixed order (deterministic).
#include "gpc_common.hpp"
#include <vector>

namespace gpc {

namespace {

__global__ void __launch_bounds__(256) covgrad_multi_kernel(const double* __restrict__ invK, int64_t ldi,
                                                            const double* __restrict__ A, int64_t lda, int d,
                                                            double* __restrict__ cg, int64_t ldc, int64_t N, int64_t j0)
{
  const int64_t j = j0 + blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if(i >= N) return;
  double aa = 0.0;
  for(int k = 0; k < d; k++) aa += A[i + k * lda] * A[j + k * lda];
  cg[i + j * ldc] = -0.5 * ((double)d * invK[i + j * ldi] - aa);
}

// The same for d <= 16 outputs with a row's A(i, :) held in registers across CJ columns (the one-column form above issues 2 d
// cached loads per element beside its 16 bytes of HBM traffic and is bound by them: 1.5 TB/s at d = 12).
template <int DM, int CJ>
__global__ void __launch_bounds__(256) covgrad_multi_regs_kernel(const double* __restrict__ invK, int64_t ldi,
                                                                 const double* __restrict__ A, int64_t lda, int d,
                                                                 double* __restrict__ cg, int64_t ldc, int64_t N, int64_t j0)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t ic = i < N ? i : N - 1;
  double ai[DM];
#pragma unroll
  for(int k = 0; k < DM; k++) ai[k] = k < d ? A[ic + (int64_t)k * lda] : 0.0;
  const int64_t jb = j0 + (int64_t)blockIdx.y * CJ;
#pragma unroll
  for(int c = 0; c < CJ; c++) {
    const int64_t j = jb + c;
    if(j >= N) break;
    double aa = 0.0;
#pragma unroll
    for(int k = 0; k < DM; k++)
      if(k < d) aa += ai[k] * A[j + (int64_t)k * lda];      // wave-uniform address: scalar loads
    if(i < N) cg[i + j * ldc] = -0.5 * ((double)d * invK[i + j * ldi] - aa);
  }
}

struct GradXArgs {
  const double* X;    // rows i (the points the derivative is taken at), N x D
  const double* X2;   // columns n, N2 x D (== X for the symmetric pass)
  const double* G;    // N x N2
  double* part;       // [nsplit][D][N]
  int64_t ldx, ldx2, ldg, N, N2;
  int D, nsplit;
  int64_t cols_per_split;
  double pair_factor;  // 2 for the symmetric pass (CGplvm.cpp:577, CGp.cpp:1168), 1 for a cross Gram
};

template <int DMAX>
__global__ void __launch_bounds__(256) kern_gradx_kernel(const KSpecDev ks, const GradXArgs g)
{
  __shared__ double red[3][64][DMAX + 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const int64_t ic = (i < g.N) ? i : g.N - 1;   // clamped: loads stay unconditional, the store is masked
  const int64_t nbeg = (int64_t)blockIdx.y * g.cols_per_split;
  int64_t nend = nbeg + g.cols_per_split;
  if(nend > g.N2) nend = g.N2;

  double xi[DMAX], acc[DMAX];
#pragma unroll
  for(int q = 0; q < DMAX; q++) {
    xi[q] = (q < g.D) ? g.X[ic + (int64_t)q * g.ldx] : 0.0;
    acc[q] = 0.0;
  }
  const bool has_rbf = ks.n_rbf > 0, has_ard = ks.n_ard > 0;
  const double lin2 = g.pair_factor * ks.lin_var;

  for(int64_t n = nbeg + w; n < nend; n += 4) {
    const double gv = g.G[ic + n * g.ldg];   // G(i,n) == G(n,i)
    double dx[DMAX], xn[DMAX];
    double d2 = 0.0, d2a = 0.0;
#pragma unroll
    for(int q = 0; q < DMAX; q++) {
      xn[q] = (q < g.D) ? g.X2[n + (int64_t)q * g.ldx2] : 0.0;   // wave-uniform address
      dx[q] = xn[q] - xi[q];
      d2 += dx[q] * dx[q];
      if(has_ard) d2a += ks.ard_scale[0][q < GPC_MAX_ARD_DIM ? q : 0] * dx[q] * dx[q];
    }
    double crbf = 0.0;
    if(has_rbf) {
      for(int t = 0; t < ks.n_rbf; t++) crbf += 2.0 * ks.rbf_hiw[t] * ks.rbf_var[t] * exp(-ks.rbf_hiw[t] * d2);
    }
    double card = 0.0;
    if(has_ard) card = 2.0 * ks.ard_hiw[0] * ks.ard_var[0] * exp(-ks.ard_hiw[0] * d2a);
    const double g2 = g.pair_factor * gv;
    const double a = g2 * crbf, b = g2 * card, c = gv * lin2;
#pragma unroll
    for(int q = 0; q < DMAX; q++) {
      double v = a * dx[q] + c * xn[q];
      if(has_ard) v += b * ks.ard_scale[0][q < GPC_MAX_ARD_DIM ? q : 0] * dx[q];
      acc[q] += v;
    }
  }
  // add the four waves' partial sums in a fixed order
  if(w > 0) {
#pragma unroll
    for(int q = 0; q < DMAX; q++) red[w - 1][lane][q] = acc[q];
  }
  __syncthreads();
  if(w == 0 && i < g.N) {
#pragma unroll
    for(int q = 0; q < DMAX; q++) {
      if(q < g.D) {
        const double v = ((acc[q] + red[0][lane][q]) + red[1][lane][q]) + red[2][lane][q];
        g.part[((int64_t)blockIdx.y * g.D + q) * g.N + i] = v;
      }
    }
  }
}

// Kernel-parameter sums of a CROSS Gram in one pass over covGrad (N x N2):  CCmpndKern::getGradParams(g, X, X2, covGrad)
// (rbf CKern.cpp:1175-1202, rbfard 3318-3357, bias 1015-1019, lin 2354-2368, white 730-734 = 0), needed by the sparse
// approximations for K_uf (CGp.cpp:1153).  Per workgroup NPC partial sums; the host adds them in a fixed order.
//   [2t], [2t+1]  rbf term t: sum cg k~ d2, sum cg k~      [8], [9] rbfard: the same with the scaled distance
//   [10] sum cg (bias)   [11] sum cg x_i.x2_n (lin)   [12 + q] rbfard: sum cg k~ (x_iq - x2_nq)^2
constexpr int NPC = 12 + GPC_MAX_ARD_DIM;
template <int DMAX, bool ARD = true>
__global__ void __launch_bounds__(256) kern_grad_cross_kernel(const KSpecDev ks, const GradXArgs g, double* __restrict__ partial)
{
  __shared__ double red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const bool row_ok = i < g.N;
  const int64_t ic = row_ok ? i : g.N - 1;
  const int64_t nbeg = (int64_t)blockIdx.y * g.cols_per_split;
  int64_t nend = nbeg + g.cols_per_split;
  if(nend > g.N2) nend = g.N2;
  double xi[DMAX], sdim[ARD ? DMAX : 1], srbf[8], sard1 = 0.0, sard2 = 0.0, sbias = 0.0, slin = 0.0;
#pragma unroll
  for(int q = 0; q < DMAX; q++) {
    xi[q] = (q < g.D) ? g.X[ic + (int64_t)q * g.ldx] : 0.0;
    if(ARD) sdim[q] = 0.0;
  }
#pragma unroll
  for(int q = 0; q < 8; q++) srbf[q] = 0.0;
  const bool has_ard = ARD && ks.n_ard > 0;
  for(int64_t n = nbeg + w; n < nend; n += 4) {
    const double cg = row_ok ? g.G[ic + n * g.ldg] : 0.0;
    double d2 = 0.0, d2a = 0.0, dot = 0.0, dq[ARD ? DMAX : 1];
#pragma unroll
    for(int q = 0; q < DMAX; q++) {
      const double xn = (q < g.D) ? g.X2[n + (int64_t)q * g.ldx2] : 0.0;   // wave-uniform address
      const double dx = xi[q] - xn;
      const double dxx = dx * dx;
      if(ARD) dq[q] = dxx;
      d2 += dxx;
      dot += xi[q] * xn;
      if(has_ard) d2a += ks.ard_scale[0][q < GPC_MAX_ARD_DIM ? q : 0] * dxx;
    }
    for(int t = 0; t < ks.n_rbf; t++) {
      const double kcg = exp(-ks.rbf_hiw[t] * d2) * cg;
      srbf[2 * t] += kcg * d2;
      srbf[2 * t + 1] += kcg;
    }
    if(has_ard) {
      const double kcg = exp(-ks.ard_hiw[0] * d2a) * cg;
      sard1 += kcg * d2a;
      sard2 += kcg;
#pragma unroll
      for(int q = 0; q < DMAX; q++)
        if(ARD) sdim[q] += kcg * dq[q];
    }
    sbias += cg;
    slin += cg * dot;
  }
  // block sums, one value at a time (fixed order: lanes by shuffle, then the four waves)
  double* out = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * NPC;
  auto block_store = [&](double v, int slot) {
    for(int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if(lane == 0) red[w] = v;
    __syncthreads();
    if(threadIdx.x == 0) out[slot] = ((red[0] + red[1]) + red[2]) + red[3];
  };
#pragma unroll
  for(int q = 0; q < 8; q++) block_store(srbf[q], q);
  block_store(sard1, 8);
  block_store(sard2, 9);
  block_store(sbias, 10);
  block_store(slin, 11);
#pragma unroll
  for(int q = 0; q < (ARD ? DMAX : 1); q++) block_store(ARD ? sdim[q] : 0.0, 12 + q);
}

__global__ void __launch_bounds__(256) gradx_reduce_kernel(const double* __restrict__ part, int nsplit, int D, int64_t N,
                                                           double* __restrict__ out, int64_t ldo)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int q = blockIdx.y;
  if(i >= N) return;
  double v = 0.0;
#pragma unroll 8
  for(int s = 0; s < nsplit; s++) v += part[((int64_t)s * D + q) * N + i];      // (the loads of eight slices in flight; same order of addition)
  out[i + (int64_t)q * ldo] = v;
}

}  // namespace

}  // namespace gpc

using namespace gpc;

extern "C" int gpc_covgrad_multi_f64(int64_t N, int64_t d, const double* invK, int64_t ldi, const double* A, int64_t lda,
                                     double* covGrad, int64_t ldc, void* stream)
{
  GPC_CHECK(ensure_device());
  GPC_REQUIRE(N >= 0 && d >= 0 && d <= 4096 && ldi >= (N > 1 ? N : 1) && ldc >= (N > 1 ? N : 1) &&
                  lda >= (N > 1 ? N : 1),
              "covgrad_multi dims");
  if(N == 0) return GPC_OK;
  hipStream_t s = as_stream(stream);
  if(d <= 16) {
    constexpr int CJ = 8;
    for(int64_t j0 = 0; j0 < N; j0 += 32768 * CJ) {
      const int64_t nc = (N - j0 < 32768 * CJ) ? (N - j0) : 32768 * CJ;
      const dim3 grid((unsigned)((N + 255) / 256), (unsigned)((nc + CJ - 1) / CJ));
      if(d <= 4)
        hipLaunchKernelGGL((covgrad_multi_regs_kernel<4, CJ>), grid, dim3(256), 0, s, invK, ldi, A, lda, (int)d, covGrad, ldc, N, j0);
      else
        hipLaunchKernelGGL((covgrad_multi_regs_kernel<16, CJ>), grid, dim3(256), 0, s, invK, ldi, A, lda, (int)d, covGrad, ldc, N, j0);
    }
    GPC_HIP_CHECK(hipGetLastError());
    return GPC_OK;
  }
  for(int64_t j0 = 0; j0 < N; j0 += 32768) {
    const int64_t nc = (N - j0 < 32768) ? (N - j0) : 32768;
    hipLaunchKernelGGL(covgrad_multi_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)nc), dim3(256), 0, s, invK, ldi,
                       A, lda, (int)d, covGrad, ldc, N, j0);
  }
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

static int launch_gradx_pass(const gpc_kspec* ksp, const double* X, int64_t N, int64_t ldx, const double* X2, int64_t N2,
                             int64_t ldx2, int64_t D, const double* covGrad, int64_t ldc, double* gX, int64_t ldg,
                             double pair_factor, hipStream_t s);

// dL/dX is a sum over the compound's terms (CCmpndKern::getGradX, CKern.cpp:184-193): a compound with more rbf / rbfard terms
// than one pass holds is taken in several, the later ones added to the first one's result
static int launch_gradx(const gpc_kspec* ksp, const double* X, int64_t N, int64_t ldx, const double* X2, int64_t N2,
                        int64_t ldx2, int64_t D, const double* covGrad, int64_t ldc, double* gX, int64_t ldg,
                        double pair_factor, hipStream_t s)
{
  std::vector<gpc_kspec> chunks;
  GPC_CHECK(split_kspec(ksp, 4, 1, &chunks, nullptr));
  GPC_CHECK(launch_gradx_pass(&chunks[0], X, N, ldx, X2, N2, ldx2, D, covGrad, ldc, gX, ldg, pair_factor, s));
  if(chunks.size() == 1 || N == 0 || D == 0) return GPC_OK;
  void* wt = nullptr;
  GPC_CHECK(workspace(WS_TRSM_TMP, sizeof(double) * (size_t)N * (size_t)D, &wt));
  double* tmp = static_cast<double*>(wt);
  for(size_t c = 1; c < chunks.size(); c++) {
    GPC_CHECK(launch_gradx_pass(&chunks[c], X, N, ldx, X2, N2, ldx2, D, covGrad, ldc, tmp, N, pair_factor, s));
    GPC_CHECK(gpc_axpby_f64(N, D, 1.0, tmp, N, 1.0, gX, ldg, s));
  }
  return GPC_OK;
}

static int launch_gradx_pass(const gpc_kspec* ksp, const double* X, int64_t N, int64_t ldx, const double* X2, int64_t N2,
                             int64_t ldx2, int64_t D, const double* covGrad, int64_t ldc, double* gX, int64_t ldg,
                             double pair_factor, hipStream_t s)
{
  if(D > GPC_MAX_ARD_DIM) {
    set_error("kern_gradx: input dimension %lld > %d is outside the accelerated set", (long long)D, GPC_MAX_ARD_DIM);
    return GPC_EUNSUPPORTED;
  }
  KSpecDev ks;
  GPC_CHECK(collapse_kspec(ksp, D, &ks));
  if(ks.n_ard > 1) {
    set_error("kern_gradx: at most one rbfard term");
    return GPC_EUNSUPPORTED;
  }
  if(N == 0 || D == 0) return GPC_OK;
  // large enough, D <= 32: the MFMA walk of pair_walk.hip (round 4; 0.65-0.86 -> TB/s figures in DESIGN.md section 3)
  {
    const int rc = pair_walk_gradx(ksp, X, N, ldx, X2, N2, ldx2, D, covGrad, ldc, gX, ldg, pair_factor, s);
    if(rc != GPC_EUNSUPPORTED) return rc;
  }
  const int64_t rb = (N + 63) / 64;
  int64_t nsplit = (1024 + rb - 1) / rb;               // >= ~1024 workgroups when the sizes allow it
  const int64_t maxsplit = (N2 + 15) / 16;             // at least 16 columns (4 per wave) per slice
  if(nsplit > maxsplit) nsplit = maxsplit;
  if(nsplit < 1) nsplit = 1;
  if(nsplit > 65535) nsplit = 65535;
  GradXArgs g;
  g.X = X;
  g.X2 = X2;
  g.G = covGrad;
  g.ldx = ldx;
  g.ldx2 = ldx2;
  g.ldg = ldc;
  g.N = N;
  g.N2 = N2;
  g.D = (int)D;
  g.pair_factor = pair_factor;
  g.cols_per_split = (N2 + nsplit - 1) / nsplit;
  if(g.cols_per_split < 1) g.cols_per_split = 1;
  nsplit = (N2 + g.cols_per_split - 1) / g.cols_per_split;
  if(nsplit < 1) nsplit = 1;
  g.nsplit = (int)nsplit;
  void* ws = nullptr;
  GPC_CHECK(workspace(WS_KERN, sizeof(double) * (size_t)nsplit * (size_t)D * (size_t)N, &ws));
  g.part = static_cast<double*>(ws);
  const dim3 grid((unsigned)rb, (unsigned)nsplit);
  // (D > 16: the same kernel with 32 / 64 dimensions per lane -- a row's whole x, x_n, difference and sums live in
  // registers, so these instances run at one wave per SIMD; latent spaces and inducing inputs of that dimension are rare, what
  // counts is that the reference's getGradX has no limit, CKern.cpp:1115-1135, 3268-3293)
  if(D <= 4)
    hipLaunchKernelGGL(kern_gradx_kernel<4>, grid, dim3(256), 0, s, ks, g);
  else if(D <= 16)
    hipLaunchKernelGGL(kern_gradx_kernel<16>, grid, dim3(256), 0, s, ks, g);
  else if(D <= 32)
    hipLaunchKernelGGL(kern_gradx_kernel<32>, grid, dim3(256), 0, s, ks, g);
  else
    hipLaunchKernelGGL(kern_gradx_kernel<64>, grid, dim3(256), 0, s, ks, g);
  GPC_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(gradx_reduce_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)D), dim3(256), 0, s, g.part,
                     (int)nsplit, (int)D, N, gX, ldg);
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

extern "C" int gpc_kern_gradx_f64(const gpc_kspec* ksp, const double* X, int64_t N, int64_t D, int64_t ldx,
                                  const double* covGrad, int64_t ldc, double* gX, int64_t ldg, void* stream)
{
  GPC_CHECK(ensure_device());
  GPC_REQUIRE(ksp && N >= 0 && D >= 0 && ldx >= (N > 1 ? N : 1) && ldc >= (N > 1 ? N : 1) && ldg >= (N > 1 ? N : 1),
              "kern_gradx args");
  return launch_gradx(ksp, X, N, ldx, X, N, ldx, D, covGrad, ldc, gX, ldg, 2.0, as_stream(stream));
}

extern "C" int gpc_kern_gradx_cross_f64(const gpc_kspec* ksp, const double* X, int64_t N, int64_t ldx, const double* X2,
                                        int64_t N2, int64_t ldx2, int64_t D, const double* covGrad, int64_t ldc, double* gX,
                                        int64_t ldg, void* stream)
{
  GPC_CHECK(ensure_device());
  GPC_REQUIRE(ksp && N >= 0 && N2 >= 0 && D >= 0 && ldx >= (N > 1 ? N : 1) && ldx2 >= (N2 > 1 ? N2 : 1) &&
                  ldc >= (N > 1 ? N : 1) && ldg >= (N > 1 ? N : 1),
              "kern_gradx_cross args");
  if(N2 == 0) {   // empty sum
    if(N > 0 && D > 0) GPC_HIP_CHECK(hipMemset2DAsync(gX, sizeof(double) * (size_t)ldg, 0, sizeof(double) * (size_t)N, (size_t)D, as_stream(stream)));
    return GPC_OK;
  }
  return launch_gradx(ksp, X, N, ldx, X2, N2, ldx2, D, covGrad, ldc, gX, ldg, 1.0, as_stream(stream));
}

static int kern_grad_cross_pass(const gpc_kspec* ksp, const double* X, int64_t N, int64_t ldx, const double* X2, int64_t N2,
                                int64_t ldx2, int64_t D, const double* covGrad, int64_t ldc, double* gout, hipStream_t s);

extern "C" int gpc_kern_grad_cross_f64(const gpc_kspec* ksp, const double* X, int64_t N, int64_t ldx, const double* X2,
                                       int64_t N2, int64_t ldx2, int64_t D, const double* covGrad, int64_t ldc,
                                       double* gout, void* stream)
{
  GPC_CHECK(ensure_device());
  GPC_REQUIRE(ksp && gout && N >= 0 && N2 >= 0 && D >= 0 && ldx >= (N > 1 ? N : 1) && ldx2 >= (N2 > 1 ? N2 : 1) &&
                  ldc >= (N > 1 ? N : 1),
              "kern_grad_cross args");
  hipStream_t s = as_stream(stream);
  // a term's parameter sums involve that term alone: compounds beyond one pass (4 rbf, 1 rbfard) go chunk by chunk
  std::vector<gpc_kspec> chunks;
  std::vector<std::vector<int>> where;
  GPC_CHECK(split_kspec(ksp, 4, 1, &chunks, &where));
  if(chunks.size() == 1) return kern_grad_cross_pass(ksp, X, N, ldx, X2, N2, ldx2, D, covGrad, ldc, gout, s);
  for(size_t c = 0; c < chunks.size(); c++) {
    double gsub[GPC_MAX_PARAMS];
    GPC_CHECK(kern_grad_cross_pass(&chunks[c], X, N, ldx, X2, N2, ldx2, D, covGrad, ldc, gsub, s));
    for(int i = 0; i < chunks[c].n_terms; i++) {
      const int t = where[c][(size_t)i];
      for(int q = 0; q < chunks[c].offs[i + 1] - chunks[c].offs[i]; q++) gout[ksp->offs[t] + q] = gsub[chunks[c].offs[i] + q];
    }
  }
  return GPC_OK;
}

static int kern_grad_cross_pass(const gpc_kspec* ksp, const double* X, int64_t N, int64_t ldx, const double* X2, int64_t N2,
                                int64_t ldx2, int64_t D, const double* covGrad, int64_t ldc, double* gout, hipStream_t s)
{
  KSpecDev ks;
  GPC_CHECK(collapse_kspec(ksp, D, &ks));
  if(D > GPC_MAX_ARD_DIM) {
    set_error("kern_grad_cross: input dimension %lld is outside the accelerated set (%d)", (long long)D, GPC_MAX_ARD_DIM);
    return GPC_EUNSUPPORTED;
  }
  const int nparams = ksp->offs[ksp->n_terms];
  for(int p = 0; p < nparams; p++) gout[p] = 0.0;
  if(N == 0 || N2 == 0) return GPC_OK;
  double S[NPC];
  bool have_sums = false;
  {
    const int rc = pair_walk_grad_cross(ks, X, N, ldx, X2, N2, ldx2, D, covGrad, ldc, S, s);   // large enough, D <= 32: the MFMA walk
    if(rc == GPC_OK) have_sums = true;
    else if(rc != GPC_EUNSUPPORTED) return rc;
  }
  if(!have_sums) {
  const int64_t rb = (N + 63) / 64;
  int64_t nsplit = (512 + rb - 1) / rb;
  const int64_t maxsplit = (N2 + 15) / 16;
  if(nsplit > maxsplit) nsplit = maxsplit;
  if(nsplit < 1) nsplit = 1;
  if(nsplit > 65535) nsplit = 65535;
  GradXArgs g;
  g.X = X;
  g.X2 = X2;
  g.G = covGrad;
  g.ldx = ldx;
  g.ldx2 = ldx2;
  g.ldg = ldc;
  g.N = N;
  g.N2 = N2;
  g.D = (int)D;
  g.pair_factor = 1.0;
  g.part = nullptr;
  g.cols_per_split = (N2 + nsplit - 1) / nsplit;
  nsplit = (N2 + g.cols_per_split - 1) / g.cols_per_split;
  g.nsplit = (int)nsplit;
  const int64_t nblk = rb * nsplit;
  void* ws = nullptr;
  GPC_CHECK(workspace(WS_KERN, sizeof(double) * (size_t)nblk * NPC, &ws));
  double* partial = static_cast<double*>(ws);
  const dim3 grid((unsigned)rb, (unsigned)nsplit);
  if(D <= 4)
    hipLaunchKernelGGL(kern_grad_cross_kernel<4>, grid, dim3(256), 0, s, ks, g, partial);
  else if(D <= 16)
    hipLaunchKernelGGL(kern_grad_cross_kernel<16>, grid, dim3(256), 0, s, ks, g, partial);
  else if(D <= 32 && ks.n_ard == 0)
    hipLaunchKernelGGL((kern_grad_cross_kernel<32, false>), grid, dim3(256), 0, s, ks, g, partial);
  else if(D <= 32)
    hipLaunchKernelGGL((kern_grad_cross_kernel<32, true>), grid, dim3(256), 0, s, ks, g, partial);
  else if(ks.n_ard == 0)
    hipLaunchKernelGGL((kern_grad_cross_kernel<64, false>), grid, dim3(256), 0, s, ks, g, partial);
  else
    hipLaunchKernelGGL((kern_grad_cross_kernel<64, true>), grid, dim3(256), 0, s, ks, g, partial);
  GPC_HIP_CHECK(hipGetLastError());
  std::vector<double> h((size_t)nblk * NPC);
  HostFetch f;
  GPC_CHECK(f.add(h.data(), partial, sizeof(double) * h.size(), s));
  GPC_CHECK(f.finish(s));
  for(int q = 0; q < NPC; q++) {
    double acc = 0.0;
    for(int64_t b = 0; b < nblk; b++) acc += h[(size_t)b * NPC + q];
    S[q] = acc;
  }
  }
  int irbf = 0;
  for(int t = 0; t < ksp->n_terms; t++) {
    double* gt = gout + ksp->offs[t];
    const double* p = ksp->params + ksp->offs[t];
    switch(ksp->types[t]) {
    case GPC_KERN_RBF:
      gt[0] = -0.5 * p[1] * S[2 * irbf];
      gt[1] = S[2 * irbf + 1];
      irbf++;
      break;
    case GPC_KERN_RBFARD:
      gt[0] = -0.5 * p[1] * S[8];
      gt[1] = S[9];
      for(int64_t q = 0; q < D; q++) gt[2 + q] = -0.5 * p[0] * p[1] * S[12 + q];
      break;
    case GPC_KERN_WHITE: gt[0] = 0.0; break;
    case GPC_KERN_BIAS: gt[0] = S[10]; break;
    case GPC_KERN_LIN: gt[0] = S[11]; break;
    default: break;
    }
  }
  return GPC_OK;
}
