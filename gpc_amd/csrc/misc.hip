// misc.hip -- the O(N^2) / O(N) helpers around the Cholesky pipeline (HBM-bound; coalesced, no MFMA).
//
// Each routine names the reference code it stands in for:
//   transpose_inplace  CMatrix::trans -> dtransr_ (CMatrix.h:789-801, ndlfortran.f:2064)
//   symmetrize         CMatrix::copySymmetric / the mirror step of pdinv (CMatrix.cpp:425-430)
//   zero_triangle      the zero-fill of CMatrix::chol (CMatrix.cpp:384-403)
//   add_diag           CMatrix::addDiag (CMatrix.h:841) -- jitChol's jitter
//   diag_reduce        trace() and logDet() (CMatrix.cpp:404-412)
//   coldot/colnorm2    ddot_/dnrm2_ per column (CGp.cpp:553-559, 606, 928-930)
//   symv               dsymv_ (lapack.h:130-140)
//   covgrad            CGp::updateCovGradient (CGp.cpp:666-679)
#include "gpc_common.hpp"
#include "gpc_exp.hpp"

namespace gpc {

namespace {

constexpr int TT = 32;  // transpose tile

// In-place transpose: one workgroup per tile pair (bi >= bj); both tiles staged through LDS, written swapped.
__global__ void __launch_bounds__(256) transpose_inplace_kernel(double* __restrict__ A, int64_t lda, int64_t N,
                                                                int ntiles)
{
  __shared__ double Ta[TT][TT + 1];
  __shared__ double Tb[TT][TT + 1];
  // triangular block index -> (bi, bj), bj <= bi
  const unsigned s = blockIdx.x;
  int r = (int)((sqrt(8.0 * (double)s + 1.0) - 1.0) * 0.5);
  while((unsigned)(r + 1) * (unsigned)(r + 2) / 2 <= s) r++;
  while((unsigned)r * (unsigned)(r + 1) / 2 > s) r--;
  const int bi = r, bj = (int)(s - (unsigned)r * (unsigned)(r + 1) / 2);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // ty in 0..7
  const int64_t i0 = (int64_t)bi * TT, j0 = (int64_t)bj * TT;
  for(int c = ty; c < TT; c += 8) {
    const int64_t i = i0 + tx, j = j0 + c;
    Ta[c][tx] = (i < N && j < N) ? A[i + j * lda] : 0.0;  // tile (bi,bj): rows i0.., cols j0..
    const int64_t i2 = j0 + tx, j2 = i0 + c;
    Tb[c][tx] = (i2 < N && j2 < N) ? A[i2 + j2 * lda] : 0.0;  // tile (bj,bi)
  }
  __syncthreads();
  for(int c = ty; c < TT; c += 8) {
    // new tile (bi,bj)(i,j) = old A(j,i) = tile (bj,bi) element (row j-j0 = c, col i-i0 = tx) -> Tb[tx][c]
    const int64_t i = i0 + tx, j = j0 + c;
    if(i < N && j < N) A[i + j * lda] = Tb[tx][c];
    if(bi != bj) {
      const int64_t i2 = j0 + tx, j2 = i0 + c;
      if(i2 < N && j2 < N) A[i2 + j2 * lda] = Ta[tx][c];
    }
  }
}

// dst triangle := transpose of src triangle (from_lower: upper(i<j) := lower(j,i)).  Tile pairs through LDS so both
// the read and the write are coalesced.
__global__ void __launch_bounds__(256) symmetrize_kernel(double* __restrict__ A, int64_t lda, int64_t N,
                                                         int from_lower)
{
  __shared__ double Ta[TT][TT + 1];
  const unsigned s = blockIdx.x;
  int r = (int)((sqrt(8.0 * (double)s + 1.0) - 1.0) * 0.5);
  while((unsigned)(r + 1) * (unsigned)(r + 2) / 2 <= s) r++;
  while((unsigned)r * (unsigned)(r + 1) / 2 > s) r--;
  const int bi = r, bj = (int)(s - (unsigned)r * (unsigned)(r + 1) / 2);  // bj <= bi
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  // source tile: lower tile (bi,bj) if from_lower else upper tile (bj,bi)
  const int64_t si0 = (int64_t)(from_lower ? bi : bj) * TT, sj0 = (int64_t)(from_lower ? bj : bi) * TT;
  for(int c = ty; c < TT; c += 8) {
    const int64_t i = si0 + tx, j = sj0 + c;
    Ta[c][tx] = (i < N && j < N) ? A[i + j * lda] : 0.0;
  }
  __syncthreads();
  // destination tile is the mirror: rows sj0.., cols si0..; dest(i,j) = src(j,i) = Ta[col = i-sj0... ]
  for(int c = ty; c < TT; c += 8) {
    const int64_t i = sj0 + tx, j = si0 + c;  // dest element
    if(i < N && j < N) {
      const bool strictly = from_lower ? (i < j) : (i > j);  // only the strictly-other triangle is overwritten
      if(strictly) A[i + j * lda] = Ta[tx][c];               // src(row = j - si0 = c, col = i - sj0 = tx)
    }
  }
}

__global__ void __launch_bounds__(256) zero_triangle_kernel(double* __restrict__ A, int64_t lda, int64_t N,
                                                            int zero_lower, int64_t j0)
{
  const int64_t j = j0 + blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if(i < N && j < N) {
    if(zero_lower ? (i > j) : (i < j)) A[i + j * lda] = 0.0;
  }
}

// A(i,j) := (double)(float)A(i,j) for i > j: what the reference's in-place transpose does to LcholK (see capi).
__global__ void __launch_bounds__(256) round_lower_f32_kernel(double* __restrict__ A, int64_t lda, int64_t N,
                                                              int64_t j0)
{
  const int64_t j = j0 + blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if(i < N && j < N && i > j) A[i + j * lda] = (double)(float)A[i + j * lda];
}

__global__ void __launch_bounds__(256) add_diag_kernel(double* __restrict__ A, int64_t lda, int64_t N, double c)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if(i < N) A[i + i * lda] += c;
}

__device__ __forceinline__ double block_sum_256(double v, double* sh)
{
  for(int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if(lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  if(threadIdx.x == 0) r = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return r;  // valid on thread 0
}

// partial[b] = sum over this block's diagonal entries of f(A(i,i))
__global__ void __launch_bounds__(256) diag_reduce_kernel(const double* __restrict__ A, int64_t lda, int64_t N,
                                                          int what, double* __restrict__ partial)
{
  __shared__ double sh[4];
  double v = 0.0;
  for(int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
    const double a = A[i + i * lda];
    v += what == 1 ? log(a) : a;
  }
  const double r = block_sum_256(v, sh);
  if(threadIdx.x == 0) partial[blockIdx.x] = r;
}

// out[j*gridDim.y + b] = partial sum over rows of A(i,j)*B(i,j); columns along grid x (no 65 535 limit: the sparse
// approximations take column sums over N data points), row blocks along grid y
__global__ void __launch_bounds__(256) coldot_kernel(const double* __restrict__ A, int64_t lda,
                                                     const double* __restrict__ B, int64_t ldb, int64_t M,
                                                     double* __restrict__ partial)
{
  __shared__ double sh[4];
  const int64_t j = blockIdx.x;
  double v = 0.0;
  for(int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x; i < M; i += (int64_t)gridDim.y * 256)
    v += A[i + j * lda] * B[i + j * ldb];
  const double r = block_sum_256(v, sh);
  if(threadIdx.x == 0) partial[j * gridDim.y + blockIdx.y] = r;
}

// out[j] = sum_i A(i,j)^2 : one workgroup per column (columns are independent test points)
__global__ void __launch_bounds__(256) colnorm2_kernel(const double* __restrict__ A, int64_t lda, int64_t M,
                                                       double* __restrict__ out)
{
  __shared__ double sh[4];
  const int64_t j = blockIdx.x;
  double v = 0.0;
  for(int64_t i = threadIdx.x; i < M; i += 256) {
    const double a = A[i + j * lda];
    v += a * a;
  }
  const double r = block_sum_256(v, sh);
  if(threadIdx.x == 0) out[j] = r;
}

// C (M x n, n <= 16) = A (M x K) B (K x n): thread = row of A, B staged through LDS 64 k-rows at a time, K split over
// grid.y with one partial per chunk (added in chunk order by skinny_combine_kernel: deterministic).
constexpr int SK_N = 16, SK_KB = 64;
__global__ void __launch_bounds__(256) gemm_skinny_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ B,
                                                          int64_t ldb, int64_t M, int n, int64_t K, int64_t kchunk,
                                                          double* __restrict__ partial)
{
  __shared__ double Bs[SK_KB * SK_N];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t kbeg = (int64_t)blockIdx.y * kchunk;
  const int64_t kend = (kbeg + kchunk < K) ? (kbeg + kchunk) : K;
  double acc[SK_N];
#pragma unroll
  for(int c = 0; c < SK_N; c++) acc[c] = 0.0;
  for(int64_t k0 = kbeg; k0 < kend; k0 += SK_KB) {
    const int lim = (int)((kend - k0 < SK_KB) ? (kend - k0) : SK_KB);
    __syncthreads();
    for(int idx = threadIdx.x; idx < SK_KB * SK_N; idx += 256) {
      const int kk = idx % SK_KB, c = idx / SK_KB;
      Bs[kk * SK_N + c] = (kk < lim && c < n) ? B[k0 + kk + (int64_t)c * ldb] : 0.0;
    }
    __syncthreads();
    if(i < M) {
      for(int kk = 0; kk < lim; kk++) {
        const double a = A[i + (k0 + kk) * lda];
#pragma unroll
        for(int c = 0; c < SK_N; c++) acc[c] = fma(a, Bs[kk * SK_N + c], acc[c]);
      }
    }
  }
  if(i < M)
    for(int c = 0; c < n; c++) partial[((int64_t)blockIdx.y * n + c) * M + i] = acc[c];
}

__global__ void __launch_bounds__(256) skinny_combine_kernel(const double* __restrict__ partial, int nchunks, int64_t M, int n,
                                                             double alpha, double beta, double* __restrict__ C, int64_t ldc)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if(i >= M) return;
  double acc = 0.0;
  for(int q = 0; q < nchunks; q++) acc += partial[((int64_t)q * n + c) * M + i];
  C[i + (int64_t)c * ldc] = alpha * acc + (beta != 0.0 ? beta * C[i + (int64_t)c * ldc] : 0.0);
}

// y := alpha*A*x + beta*y, A full symmetric storage: use columns (coalesced along i): y_i = sum_j A(i,j) x_j.
// One thread per row i, loop over j in chunks staged through LDS for x.
__global__ void __launch_bounds__(256) symv_kernel(const double* __restrict__ A, int64_t lda, int64_t N,
                                                   double alpha, const double* __restrict__ x, double beta,
                                                   double* __restrict__ y, int64_t jchunk, double* __restrict__ partial)
{
  __shared__ double xs[256];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t jbeg = (int64_t)blockIdx.y * jchunk;
  const int64_t jend = (jbeg + jchunk < N) ? (jbeg + jchunk) : N;
  double acc = 0.0;
  for(int64_t j0 = jbeg; j0 < jend; j0 += 256) {
    const int64_t jj = j0 + threadIdx.x;
    xs[threadIdx.x] = (jj < jend) ? x[jj] : 0.0;
    __syncthreads();
    const int lim = (int)((jend - j0 < 256) ? (jend - j0) : 256);
    if(i < N)
      for(int c = 0; c < lim; c++) acc += A[i + (j0 + c) * lda] * xs[c];
    __syncthreads();
  }
  if(i < N) {
    // several column chunks: each leaves its partial sum, symv_combine_kernel adds them in chunk order -- the same bits
    // on every run (an fp64 atomicAdd into y would let the hardware pick the order)
    if(gridDim.y == 1)
      y[i] = alpha * acc + (beta != 0.0 ? beta * y[i] : 0.0);
    else
      partial[(int64_t)blockIdx.y * N + i] = acc;
  }
}

__global__ void __launch_bounds__(256) symv_combine_kernel(const double* __restrict__ partial, int nchunks, int64_t N,
                                                           double alpha, double beta, double* __restrict__ y)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if(i >= N) return;
  double acc = 0.0;
  for(int c = 0; c < nchunks; c++) acc += partial[(int64_t)c * N + i];
  y[i] = alpha * acc + (beta != 0.0 ? beta * y[i] : 0.0);
}

__global__ void __launch_bounds__(256) scale_vec_kernel(double* __restrict__ y, int64_t N, double beta)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if(i < N) y[i] = (beta != 0.0) ? beta * y[i] : 0.0;
}

// covGrad(i,j) = -0.5*(invK(i,j) - a_i a_j)
__global__ void __launch_bounds__(256) covgrad_kernel(const double* __restrict__ invK, int64_t ldi,
                                                      const double* __restrict__ a, double* __restrict__ cg,
                                                      int64_t ldc, int64_t N, int64_t j0)
{
  const int64_t j = j0 + blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if(i < N) cg[i + j * ldc] = -0.5 * (invK[i + j * ldi] - a[i] * a[j]);
}

// Y := alpha X + beta Y, elementwise over an M x N block (daxpy_ / dscal_ / dcopy_ of lapack.h:78-111 in one kernel)
__global__ void __launch_bounds__(256) axpby_kernel(int64_t M, double alpha, const double* __restrict__ X, int64_t ldx,
                                                    double beta, double* __restrict__ Y, int64_t ldy, int64_t j0)
{
  const int64_t j = j0 + blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if(i >= M) return;
  const double x = (alpha == 0.0) ? 0.0 : alpha * X[i + j * ldx];
  Y[i + j * ldy] = (beta == 0.0) ? x : fma(beta, Y[i + j * ldy], x);
}

// A(i,j) *= v[j] (by_rows == 0: CMatrix::scaleCol for every column) or A(i,j) *= v[i] (by_rows != 0: scaleRow)
__global__ void __launch_bounds__(256) scale_vec_kernel(int64_t M, double* __restrict__ A, int64_t lda,
                                                        const double* __restrict__ v, int by_rows, int64_t j0)
{
  const int64_t j = j0 + blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if(i < M) A[i + j * lda] *= by_rows ? v[i] : v[j];
}

int reduce_partials_to_host(const double* d_partial, int64_t ncols, int64_t nb, double* out_host, hipStream_t s,
                            int* extra_dst = nullptr, const int* extra_src = nullptr)
{
  // small: copy partials back and finish on the host in a fixed order (deterministic).  extra_src: one device int that
  // travels in the same synchronisation (LAPACK's info, the solve-fault word)
  const size_t n = (size_t)(ncols * nb);
  double* h = (double*)malloc(sizeof(double) * (n ? n : 1));
  if(!h) return GPC_ENOMEM;
  HostFetch f;
  int rc = f.add(h, d_partial, sizeof(double) * n, s);
  if(rc == GPC_OK && extra_src) rc = f.add(extra_dst, extra_src, sizeof(int), s);
  if(rc == GPC_OK) rc = f.finish(s);
  if(rc != GPC_OK) {
    free(h);
    return rc;
  }
  for(int64_t j = 0; j < ncols; j++) {
    double acc = 0.0;
    for(int64_t b = 0; b < nb; b++) acc += h[j * nb + b];
    out_host[j] = acc;
  }
  free(h);
  return GPC_OK;
}

// partial[by * M + j] = sum over the columns of chunk by of A(j,i)^2: lane = row (coalesced), the four waves stride
// the chunk's columns
__global__ void __launch_bounds__(256) rownorm2_partial_kernel(const double* __restrict__ A, int64_t lda, int64_t M, int64_t N,
                                                               int64_t chunk, double* __restrict__ partial)
{
  __shared__ double red[3][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * 64 + lane;
  const int64_t jc = (j < M) ? j : (M - 1);
  const int64_t c0 = (int64_t)blockIdx.y * chunk;
  const int64_t c1 = (c0 + chunk < N) ? (c0 + chunk) : N;
  double acc = 0.0;
  for(int64_t c = c0 + w; c < c1; c += 4) {
    const double a = A[jc + c * lda];
    acc = fma(a, a, acc);
  }
  if(w > 0) red[w - 1][lane] = acc;
  __syncthreads();
  if(w == 0 && j < M) partial[(int64_t)blockIdx.y * M + j] = ((acc + red[0][lane]) + red[1][lane]) + red[2][lane];
}

__global__ void __launch_bounds__(256) rownorm2_final_kernel(const double* __restrict__ partial, int64_t M, int nch,
                                                             const double* __restrict__ base, double* __restrict__ out)
{
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if(j >= M) return;
  double acc = 0.0;
  for(int c = 0; c < nch; c++) acc += partial[(int64_t)c * M + j];
  out[j] = base[j] - acc;
}

inline unsigned tri_count(int64_t nt) { return (unsigned)(nt * (nt + 1) / 2); }

}  // namespace

int transpose_inplace(int64_t N, double* A, int64_t lda, hipStream_t s)
{
  if(N <= 1) return GPC_OK;
  const int64_t nt = (N + TT - 1) / TT;
  hipLaunchKernelGGL(transpose_inplace_kernel, dim3(tri_count(nt)), dim3(256), 0, s, A, lda, N, (int)nt);
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

int symmetrize(bool from_lower, int64_t N, double* A, int64_t lda, hipStream_t s)
{
  if(N <= 1) return GPC_OK;
  const int64_t nt = (N + TT - 1) / TT;
  hipLaunchKernelGGL(symmetrize_kernel, dim3(tri_count(nt)), dim3(256), 0, s, A, lda, N, from_lower ? 1 : 0);
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

int zero_triangle(bool zero_lower, int64_t N, double* A, int64_t lda, hipStream_t s)
{
  if(N <= 1) return GPC_OK;
  for(int64_t j0 = 0; j0 < N; j0 += 32768) {  // grid.y limit
    const int64_t nc = (N - j0 < 32768) ? (N - j0) : 32768;
    hipLaunchKernelGGL(zero_triangle_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)nc), dim3(256), 0, s, A,
                       lda, N, zero_lower ? 1 : 0, j0);
  }
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

int add_diag(int64_t N, double* A, int64_t lda, double c, hipStream_t s)
{
  if(N <= 0) return GPC_OK;
  hipLaunchKernelGGL(add_diag_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, A, lda, N, c);
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

int gemm_skinny(int64_t M, int64_t n, int64_t K, double alpha, const double* A, int64_t lda, const double* B, int64_t ldb,
                double beta, double* C, int64_t ldc, hipStream_t s)
{
  if(M <= 0 || n <= 0) return GPC_OK;
  if(n > SK_N) return GPC_EINVAL;
  const unsigned gx = (unsigned)((M + 255) / 256);
  // enough column chunks for ~512 workgroups, each at least one staged block deep
  int64_t ny = (512 + gx - 1) / gx;
  const int64_t maxy = (K + SK_KB - 1) / SK_KB;
  if(ny > maxy) ny = maxy;
  if(ny < 1) ny = 1;
  int64_t kchunk = ((K + ny - 1) / ny + SK_KB - 1) / SK_KB * SK_KB;
  if(kchunk < SK_KB) kchunk = SK_KB;
  ny = K > 0 ? (K + kchunk - 1) / kchunk : 1;
  void* ws = nullptr;
  GPC_CHECK(workspace(WS_REDUCE, sizeof(double) * (size_t)(ny * n * M), &ws));
  double* partial = static_cast<double*>(ws);
  hipLaunchKernelGGL(gemm_skinny_kernel, dim3(gx, (unsigned)ny), dim3(256), 0, s, A, lda, B, ldb, M, (int)n, K, kchunk, partial);
  hipLaunchKernelGGL(skinny_combine_kernel, dim3(gx, (unsigned)n), dim3(256), 0, s, partial, (int)ny, M, (int)n, alpha, beta, C,
                     ldc);
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

__global__ void __launch_bounds__(256) build_augmented_kernel(int64_t N, int64_t Np, const double* __restrict__ K, int64_t ldk,
                                                              double* __restrict__ W, int64_t ldw)
{
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  if(r >= Np + N) return;
  double v;
  if(r < N && c < N) v = K[r + c * ldk];
  else if(r < Np) v = (r == c) ? 1.0 : 0.0;
  else v = (r - Np == c) ? 1.0 : 0.0;
  W[r + c * ldw] = v;
}

// The two halves of diag_reduce for callers that want to put more work on the stream before they wait for the scalar:
// diag_reduce_launch leaves the partial sums in the WS_REDUCE slot, diag_reduce_fetch brings them back (synchronises).
int diag_reduce_launch(int what, int64_t N, const double* A, int64_t lda, double** partial, int64_t* nparts, hipStream_t s)
{
  *partial = nullptr;
  *nparts = 0;
  if(N <= 0) return GPC_OK;
  int64_t nb = (N + 255) / 256;
  if(nb > 1024) nb = 1024;
  void* ws = nullptr;
  GPC_CHECK(workspace(WS_REDUCE, sizeof(double) * (size_t)nb, &ws));
  hipLaunchKernelGGL(diag_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, s, A, lda, N, what, static_cast<double*>(ws));
  GPC_HIP_CHECK(hipGetLastError());
  *partial = static_cast<double*>(ws);
  *nparts = nb;
  return GPC_OK;
}

int diag_reduce_fetch(const double* partial, int64_t nparts, double* out_host, hipStream_t s, int* extra_dst, const int* extra_src)
{
  *out_host = 0.0;
  if(nparts <= 0) {
    if(extra_src) {
      HostFetch f;
      GPC_CHECK(f.add(extra_dst, extra_src, sizeof(int), s));
      return f.finish(s);
    }
    return GPC_OK;
  }
  return reduce_partials_to_host(partial, 1, nparts, out_host, s, extra_dst, extra_src);
}

// W = [K 0; 0 I; I 0] for gpc_chol_inverse_f64 (capi.hip): K is N x N, W is (Np + N) x Np
int build_augmented(int64_t N, int64_t Np, const double* K, int64_t ldk, double* W, int64_t ldw, hipStream_t s)
{
  const int64_t rows = Np + N;
  hipLaunchKernelGGL(build_augmented_kernel, dim3((unsigned)((rows + 255) / 256), (unsigned)Np), dim3(256), 0, s, N, Np, K, ldk, W, ldw);
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

int diag_reduce(int what, int64_t N, const double* A, int64_t lda, double* out_host, hipStream_t s)
{
  *out_host = 0.0;
  if(N <= 0) return GPC_OK;
  int64_t nb = (N + 255) / 256;
  if(nb > 1024) nb = 1024;
  void* ws = nullptr;
  GPC_CHECK(workspace(WS_REDUCE, sizeof(double) * (size_t)nb, &ws));
  hipLaunchKernelGGL(diag_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, s, A, lda, N, what,
                     static_cast<double*>(ws));
  GPC_HIP_CHECK(hipGetLastError());
  return reduce_partials_to_host(static_cast<double*>(ws), 1, nb, out_host, s);
}

int rownorm2_sub(int64_t M, int64_t N, const double* A, int64_t lda, const double* base, double* out, hipStream_t s)
{
  if(M <= 0) return GPC_OK;
  int64_t nch = (N + 255) / 256;
  if(nch > 64) nch = 64;
  if(nch < 1) nch = 1;
  const int64_t chunk = (N + nch - 1) / nch;
  void* ws = nullptr;
  GPC_CHECK(workspace(WS_REDUCE, sizeof(double) * (size_t)(nch * M), &ws));
  double* partial = static_cast<double*>(ws);
  hipLaunchKernelGGL(rownorm2_partial_kernel, dim3((unsigned)((M + 63) / 64), (unsigned)nch), dim3(256), 0, s, A, lda, M, N, chunk,
                     partial);
  hipLaunchKernelGGL(rownorm2_final_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, partial, M, (int)nch, base, out);
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

}  // namespace gpc

using namespace gpc;

extern "C" int gpc_ref_trans_rounding_f64(int64_t N, double* A, int64_t lda, void* stream)
{
  GPC_CHECK(ensure_device());
  GPC_REQUIRE(N >= 0 && lda >= (N > 1 ? N : 1), "ref_trans_rounding dims");
  hipStream_t s = as_stream(stream);
  for(int64_t j0 = 0; j0 < N; j0 += 32768) {
    const int64_t nc = (N - j0 < 32768) ? (N - j0) : 32768;
    hipLaunchKernelGGL(round_lower_f32_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)nc), dim3(256), 0, s, A,
                       lda, N, j0);
  }
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

extern "C" int gpc_coldot_f64(int64_t M, int64_t ncols, const double* A, int64_t lda, const double* B, int64_t ldb,
                              double* out, void* stream)
{
  GPC_CHECK(ensure_device());
  GPC_REQUIRE(M >= 0 && ncols >= 0 && ncols <= 0x7fffffffLL, "coldot dims");
  if(ncols == 0) return GPC_OK;
  hipStream_t s = as_stream(stream);
  int64_t nb = (M + 255) / 256;
  if(nb > 256) nb = 256;
  if(nb < 1) nb = 1;
  void* ws = nullptr;
  GPC_CHECK(workspace(WS_REDUCE, sizeof(double) * (size_t)(nb * ncols), &ws));
  hipLaunchKernelGGL(coldot_kernel, dim3((unsigned)ncols, (unsigned)nb), dim3(256), 0, s, A, lda, B, ldb, M,
                     static_cast<double*>(ws));
  GPC_HIP_CHECK(hipGetLastError());
  // the column dots of CGp (m' K^-1 m, CGp.cpp:928-930) follow the dataflow solves: this is where a solve that gave up
  // (poisoned with NaN) is reported instead of being handed back as GPC_OK; the sticky word travels with the partial sums
  int fault = 0;
  void* wi = nullptr;
  GPC_CHECK(workspace(WS_INFO, 64, &wi));
  int* sticky = static_cast<int*>(wi) + SOLVE_FAULT_WORD;
  GPC_CHECK(reduce_partials_to_host(static_cast<double*>(ws), ncols, nb, out, s, &fault, sticky));
  if(fault) {
    GPC_HIP_CHECK(hipMemsetAsync(sticky, 0, sizeof(int), s));
    set_error("a dataflow triangular solve timed out (device shared or pre-empted?); its result is NaN -- repeat the call, or "
              "set GPC_TRSV_FLOW=0 for the stepped kernels");
    return GPC_EHIP;
  }
  return GPC_OK;
}

extern "C" int gpc_colnorm2_f64(int64_t M, int64_t ncols, const double* A, int64_t lda, double* out_dev,
                                void* stream)
{
  GPC_CHECK(ensure_device());
  GPC_REQUIRE(M >= 0 && ncols >= 0, "colnorm2 dims");
  if(ncols == 0) return GPC_OK;
  hipLaunchKernelGGL(colnorm2_kernel, dim3((unsigned)ncols), dim3(256), 0, as_stream(stream), A, lda, M, out_dev);
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

extern "C" int gpc_symv_f64(int64_t N, double alpha, const double* A, int64_t lda, const double* x, double beta,
                            double* y, void* stream)
{
  GPC_CHECK(ensure_device());
  GPC_REQUIRE(N >= 0 && lda >= N, "symv dims");
  if(N == 0) return GPC_OK;
  hipStream_t s = as_stream(stream);
  const unsigned gx = (unsigned)((N + 255) / 256);
  // split the j range so that at least ~1024 workgroups are in flight on large N
  unsigned gy = 1;
  if(gx < 1024) {
    gy = (1024 + gx - 1) / gx;
    const unsigned maxy = (unsigned)((N + 1023) / 1024);
    if(gy > maxy) gy = maxy;
    if(gy < 1) gy = 1;
  }
  int64_t jchunk = (N + gy - 1) / gy;
  jchunk = ((jchunk + 255) / 256) * 256;
  gy = (unsigned)((N + jchunk - 1) / jchunk);
  double* partial = nullptr;
  if(gy > 1) {
    void* ws = nullptr;
    GPC_CHECK(workspace(WS_REDUCE, sizeof(double) * (size_t)gy * (size_t)N, &ws));
    partial = static_cast<double*>(ws);
  }
  hipLaunchKernelGGL(symv_kernel, dim3(gx, gy), dim3(256), 0, s, A, lda, N, alpha, x, beta, y, jchunk, partial);
  if(gy > 1) hipLaunchKernelGGL(symv_combine_kernel, dim3(gx), dim3(256), 0, s, partial, (int)gy, N, alpha, beta, y);
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

extern "C" int gpc_covgrad_f64(int64_t N, const double* invK, int64_t ldi, const double* a, double* covGrad,
                               int64_t ldc, void* stream)
{
  GPC_CHECK(ensure_device());
  GPC_REQUIRE(N >= 0 && ldi >= N && ldc >= N, "covgrad dims");
  if(N == 0) return GPC_OK;
  hipStream_t s = as_stream(stream);
  for(int64_t j0 = 0; j0 < N; j0 += 32768) {
    const int64_t nc = (N - j0 < 32768) ? (N - j0) : 32768;
    hipLaunchKernelGGL(covgrad_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)nc), dim3(256), 0, s,
                       invK, ldi, a, covGrad, ldc, N, j0);
  }
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

extern "C" int gpc_axpby_f64(int64_t M, int64_t N, double alpha, const double* X, int64_t ldx, double beta, double* Y,
                             int64_t ldy, void* stream)
{
  GPC_CHECK(ensure_device());
  GPC_REQUIRE(M >= 0 && N >= 0 && ldx >= (M > 1 ? M : 1) && ldy >= (M > 1 ? M : 1), "axpby dims");
  if(M == 0 || N == 0) return GPC_OK;
  hipStream_t s = as_stream(stream);
  for(int64_t j0 = 0; j0 < N; j0 += 32768) {
    const int64_t nc = (N - j0 < 32768) ? (N - j0) : 32768;
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((M + 255) / 256), (unsigned)nc), dim3(256), 0, s, M, alpha, X, ldx,
                       beta, Y, ldy, j0);
  }
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

extern "C" int gpc_scale_vec_f64(int64_t M, int64_t N, double* A, int64_t lda, const double* v_dev, int by_rows, void* stream)
{
  GPC_CHECK(ensure_device());
  GPC_REQUIRE(M >= 0 && N >= 0 && lda >= (M > 1 ? M : 1) && v_dev != nullptr, "scale_vec args");
  if(M == 0 || N == 0) return GPC_OK;
  hipStream_t s = as_stream(stream);
  for(int64_t j0 = 0; j0 < N; j0 += 32768) {
    const int64_t nc = (N - j0 < 32768) ? (N - j0) : 32768;
    hipLaunchKernelGGL(scale_vec_kernel, dim3((unsigned)((M + 255) / 256), (unsigned)nc), dim3(256), 0, s, M, A, lda, v_dev,
                       by_rows, j0);
  }
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

namespace gpc {
namespace {
__global__ void __launch_bounds__(256) debug_exp_kernel(const double* __restrict__ x, double* __restrict__ y, int64_t n)
{
  __shared__ double tab[64];
  gpc_exp_tab_fill(tab);
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if(i < n) y[i] = gpc_exp_tab(x[i], tab);
}
}  // namespace
}  // namespace gpc

extern "C" int gpc_debug_exp_f64(const double* x, double* y, int64_t n, void* stream)
{
  GPC_CHECK(gpc::ensure_device());
  if(n <= 0) return GPC_OK;
  hipLaunchKernelGGL(gpc::debug_exp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}
