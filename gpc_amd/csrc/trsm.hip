// trsm.hip -- dtrsm for all 16 side/uplo/trans/diag variants (lapack.h:208-218; CMatrix::trsm CMatrix.cpp:272-295),
// and the triangular-inverse / LAUUM-style product behind dpotri (lapack.h:67-73; CMatrix::pdinv CMatrix.cpp:421-432).
//
// Blocked substitution over 64-wide diagonal blocks; each step is
//     X_b := op(A_bb)^-1 B_b            true forward/backward substitution in LDS (trsm_diag_kernel): one lane per
//                                       right-hand-side vector, the triangle broadcast from LDS.  Substitution, not
//                                       multiplication by an explicit inverse, keeps the componentwise accuracy of the
//                                       reference BLAS (the reference's trsm fixture holds ill-conditioned triangles).
//     B_rest -= op(A)[rest,b] * X_b     fp64 MFMA GEMM -- the O(n^2 * nrhs) part
// The FTC paths use side 'L', lower: (N) then (T) for alpha = K^-1 m (CGp.cpp:481-483) and (N) with nrhs = N* for the
// predictive variance (CGp.cpp:603).
#include "gpc_common.hpp"
#include <ctype.h>

namespace gpc {

namespace {

constexpr int JB = 64;

__global__ void __launch_bounds__(256) scale_matrix_kernel(double* __restrict__ B, int64_t ldb, int64_t M,
                                                           int64_t j0, double alpha)
{
  const int64_t j = j0 + blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if(i < M) B[i + j * ldb] = (alpha == 0.0) ? 0.0 : alpha * B[i + j * ldb];
}

__global__ void __launch_bounds__(256) set_identity_kernel(double* __restrict__ B, int64_t ldb, int64_t N,
                                                           int64_t j0)
{
  const int64_t j = j0 + blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if(i < N) B[i + j * ldb] = (i == j) ? 1.0 : 0.0;
}

// Solve S x = b for 64 vectors per workgroup (one wave), S = the effective nb x nb triangle of this step:
//   left  (vectors = columns of B_b): S = op(A_bb);   right (vectors = rows of B_b): S = op(A_bb)'.
// s_lower: S is lower triangular (forward substitution) else upper (backward).
// a_trans: S(i,k) = A_bb(k,i) instead of A_bb(i,k).  vec_is_col: vector v, element k lives at B[k + v*ldb] (left)
// else at B[v + k*ldb] (right).
__global__ void __launch_bounds__(64) trsm_diag_kernel(const double* __restrict__ Abb, int64_t lda, int nb,
                                                       int s_lower, int a_trans, int unit,
                                                       double* __restrict__ B, int64_t ldb, int64_t nvec,
                                                       int vec_is_col)
{
  __shared__ double S[JB * (JB + 1)];
  __shared__ double V[JB * (JB + 1)];
  const int t = threadIdx.x;
  const int64_t v0 = (int64_t)blockIdx.x * JB;
  // triangle: S[i*(JB+1) + k] = S(i,k); rows are read by all lanes at the same address (broadcast)
  for(int idx = t; idx < nb * nb; idx += 64) {
    const int i = idx % nb, k = idx / nb;           // coalesced along the stored column
    const double a = Abb[i + (int64_t)k * lda];     // A_bb(i,k)
    if(a_trans) S[k * (JB + 1) + i] = a;            // S(k,i) = A_bb(i,k)
    else S[i * (JB + 1) + k] = a;
  }
  // vectors: V[k*(JB+1) + v]
  if(vec_is_col) {
    for(int v = 0; v < JB; v++) {
      if(v0 + v < nvec && t < nb) V[t * (JB + 1) + v] = B[t + (v0 + v) * ldb];
    }
  } else {
    for(int k = 0; k < nb; k++) {
      if(v0 + t < nvec) V[k * (JB + 1) + t] = B[(v0 + t) + (int64_t)k * ldb];
    }
  }
  __syncthreads();
  if(v0 + t < nvec) {
    if(s_lower) {
      for(int i = 0; i < nb; i++) {
        double sum = V[i * (JB + 1) + t];
        for(int k = 0; k < i; k++) sum -= S[i * (JB + 1) + k] * V[k * (JB + 1) + t];
        V[i * (JB + 1) + t] = unit ? sum : sum / S[i * (JB + 1) + i];
      }
    } else {
      for(int i = nb - 1; i >= 0; i--) {
        double sum = V[i * (JB + 1) + t];
        for(int k = i + 1; k < nb; k++) sum -= S[i * (JB + 1) + k] * V[k * (JB + 1) + t];
        V[i * (JB + 1) + t] = unit ? sum : sum / S[i * (JB + 1) + i];
      }
    }
  }
  __syncthreads();
  if(vec_is_col) {
    for(int v = 0; v < JB; v++) {
      if(v0 + v < nvec && t < nb) B[t + (v0 + v) * ldb] = V[t * (JB + 1) + v];
    }
  } else {
    for(int k = 0; k < nb; k++) {
      if(v0 + t < nvec) B[(v0 + t) + (int64_t)k * ldb] = V[k * (JB + 1) + t];
    }
  }
}

int scale_matrix(int64_t M, int64_t N, double alpha, double* B, int64_t ldb, hipStream_t s)
{
  if(M <= 0 || N <= 0 || alpha == 1.0) return GPC_OK;
  for(int64_t j0 = 0; j0 < N; j0 += 32768) {
    const int64_t nc = (N - j0 < 32768) ? (N - j0) : 32768;
    hipLaunchKernelGGL(scale_matrix_kernel, dim3((unsigned)((M + 255) / 256), (unsigned)nc), dim3(256), 0, s, B,
                       ldb, M, j0, alpha);
  }
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

// tri_rhs: the right-hand side is itself lower triangular with the same blocking (B = I for trtri); only the
// columns that can be non-zero are touched.  Only meaningful for side L / effective-lower / forward.
int trsm_impl(bool left, bool lower, bool tr, bool unit, int64_t M, int64_t Nrhs, double alpha, const double* A,
              int64_t lda, double* B, int64_t ldb, bool tri_rhs, hipStream_t s)
{
  if(M <= 0 || Nrhs <= 0) return GPC_OK;
  const int64_t nt = left ? M : Nrhs;
  const int64_t nblk = (nt + JB - 1) / JB;
  GPC_CHECK(scale_matrix(M, Nrhs, alpha, B, ldb, s));

  const bool eff_lower = (lower != tr);  // is op(A) lower triangular?
  // left : op(A) X = B.  eff_lower -> forward over row blocks, else backward.
  // right: X op(A) = B.  eff_lower -> backward over column blocks, else forward.
  const bool forward = left ? eff_lower : !eff_lower;
  for(int64_t step = 0; step < nblk; step++) {
    const int64_t b = forward ? step : (nblk - 1 - step);
    const int64_t b0 = b * JB;
    const int64_t nb = (nt - b0 < JB) ? (nt - b0) : JB;
    const double* Abb = A + b0 + b0 * lda;
    // "rest" = the blocks still to be solved
    const int64_t r0 = forward ? (b0 + nb) : 0;
    const int64_t nrest = forward ? (nt - (b0 + nb)) : b0;
    if(left) {
      const int64_t ncols = tri_rhs ? ((b0 + nb < Nrhs) ? (b0 + nb) : Nrhs) : Nrhs;
      double* Bb = B + b0;
      // S = op(A_bb): lower iff eff_lower
      hipLaunchKernelGGL(trsm_diag_kernel, dim3((unsigned)((ncols + JB - 1) / JB)), dim3(64), 0, s, Abb, lda,
                         (int)nb, eff_lower ? 1 : 0, tr ? 1 : 0, unit ? 1 : 0, Bb, ldb, ncols, 1);
      if(nrest > 0) {
        // op(A)[rest, b]: not transposed -> A(rest rows, b cols); transposed -> A(b rows, rest cols)'
        const double* Arb = tr ? (A + b0 + r0 * lda) : (A + r0 + b0 * lda);
        GPC_CHECK(gemm(tr, false, nrest, ncols, nb, -1.0, Arb, lda, Bb, ldb, 1.0, B + r0, ldb, 0, s));
      }
    } else {
      double* Bb = B + b0 * ldb;
      // rows of X_b solve x' op(A_bb) = b'  <=>  op(A_bb)' x = b: S = op(A_bb)', lower iff op(A_bb) is upper
      hipLaunchKernelGGL(trsm_diag_kernel, dim3((unsigned)((M + JB - 1) / JB)), dim3(64), 0, s, Abb, lda, (int)nb,
                         eff_lower ? 0 : 1, tr ? 0 : 1, unit ? 1 : 0, Bb, ldb, M, 0);
      if(nrest > 0) {
        // op(A)[b, rest]: not transposed -> A(b rows, rest cols); transposed -> A(rest rows, b cols)'
        const double* Abr = tr ? (A + r0 + b0 * lda) : (A + b0 + r0 * lda);
        GPC_CHECK(gemm(false, tr, M, nrest, nb, -1.0, Bb, ldb, Abr, lda, 1.0, B + r0 * ldb, ldb, 0, s));
      }
    }
  }
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

}  // namespace

int trsm(char side, char uplo, char trans, char diag, int64_t M, int64_t Nrhs, double alpha, const double* A,
         int64_t lda, double* B, int64_t ldb, hipStream_t s)
{
  const char sd = (char)toupper(side), ul = (char)toupper(uplo), tc = (char)toupper(trans),
             dg = (char)toupper(diag);
  if(!(sd == 'L' || sd == 'R') || !(ul == 'L' || ul == 'U') || !(tc == 'N' || tc == 'T' || tc == 'C') ||
     !(dg == 'N' || dg == 'U')) {
    set_error("trsm: bad side/uplo/trans/diag '%c%c%c%c'", side, uplo, trans, diag);
    return GPC_EINVAL;
  }
  const int64_t nt = (sd == 'L') ? M : Nrhs;
  if(M < 0 || Nrhs < 0 || lda < (nt > 1 ? nt : 1) || ldb < (M > 1 ? M : 1)) {
    set_error("trsm: bad dimensions");
    return GPC_EINVAL;
  }
  return trsm_impl(sd == 'L', ul == 'L', tc != 'N', dg == 'U', M, Nrhs, alpha, A, lda, B, ldb, false, s);
}

// A (factor in triangle uplo) -> full symmetric inverse of the factored matrix, in place.
//   lower: K^-1 = W' W with W = L^-1;  upper (K = U'U): K^-1 = V V' with V = U^-1 = (L^-1)' for L = U'.
// The upper case is handled by transposing in place, so only the lower algorithm exists.
int potri_full(bool lower, int64_t N, double* A, int64_t lda, hipStream_t s)
{
  if(N <= 0) return GPC_OK;
  if(!lower) GPC_CHECK(transpose_inplace(N, A, lda, s));
  void* ws = nullptr;
  GPC_CHECK(workspace(WS_POTRI, sizeof(double) * (size_t)N * (size_t)N, &ws));
  double* W = static_cast<double*>(ws);
  for(int64_t j0 = 0; j0 < N; j0 += 32768) {
    const int64_t nc = (N - j0 < 32768) ? (N - j0) : 32768;
    hipLaunchKernelGGL(set_identity_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)nc), dim3(256), 0, s, W, N,
                       N, j0);
  }
  GPC_HIP_CHECK(hipGetLastError());
  // W := L^-1 (lower triangular; the strictly upper part of W stays exactly zero)
  GPC_CHECK(trsm_impl(true, true, false, false, N, N, 1.0, A, lda, W, N, true, s));
  // lower(A) := W' W, then mirror
  GPC_CHECK(gemm(true, false, N, N, N, 1.0, W, N, W, N, 0.0, A, lda, 1, s));
  GPC_CHECK(symmetrize(true, N, A, lda, s));
  return GPC_OK;
}

}  // namespace gpc
