// potrf.hip -- right-looking blocked Cholesky (lower, column-major, in place) for gfx950.
//
// Replaces dpotrf_ as called by CMatrix::potrf / chol / jitChol (CMatrix.cpp:371-403, 767-804; lapack.h:59-65).
//
// Structure:
//   outer panels of NB columns (default 512): after a panel is final, ONE SYRK  A22 -= L21 * L21'  of depth NB runs on
//   the fp64 MFMA tiles of gemm_f64.hip -- this is where N^3/3 of the flops go, and a deep k keeps the read+write of
//   A22 (8 N^2 bytes per panel) at NB/8 flop per byte;
//   inside a panel, JB = 64 columns at a time:
//     1. potf2_inv_kernel: one workgroup holds the 64 x 64 diagonal block in LDS, factors it column by column
//        (fp64 VALU, one barrier per column) and inverts the triangular factor by 16 x 16 blocks;
//     2. the panel solve  L21 := A21 * L11^-T  is then a plain GEMM against the inverted block (MFMA, in place,
//        every workgroup owns whole rows);
//     3. the still-to-be-factored columns of the panel are updated with that 64-deep block column (lower trapezoid).
//   A non-positive pivot writes the LAPACK `info` (1-based order of the failing minor) to a device word; later
//   diagonal kernels turn into no-ops and the host reads the word once at the end.
#include "gpc_common.hpp"
#include <stdlib.h>

namespace gpc {

namespace {

constexpr int JB = 64;
constexpr int LDP = 65;  // padded leading dimension of the LDS images

int64_t g_nb_outer = 0;

// W := inverse of the lower-triangular 64 x 64 matrix L (both LDS, leading dimension LDP); T is a 16 x 48 scratch.
// All 256 threads of the workgroup call this.  Padding rows/cols of L must be identity.
__device__ void tri_inverse_64(const double* __restrict__ L, double* __restrict__ W, double* __restrict__ T)
{
  const int t = threadIdx.x;
  for(int idx = t; idx < JB * LDP; idx += 256) W[idx] = 0.0;
  __syncthreads();
  // phase A: the four 16 x 16 diagonal blocks, one column per thread, forward substitution
  if(t < 64) {
    const int b = t >> 4, c = t & 15;
    const int o = b * 16;
    for(int i = c; i < 16; i++) {
      double sum = (i == c) ? 1.0 : 0.0;
      for(int k = c; k < i; k++) sum -= L[(o + i) + (o + k) * LDP] * W[(o + k) + (o + c) * LDP];
      W[(o + i) + (o + c) * LDP] = sum / L[(o + i) + (o + i) * LDP];
    }
  }
  __syncthreads();
  // phase B: block rows 1..3:  W(bi,bj) = -W(bi,bi) * sum_{bk=bj}^{bi-1} L(bi,bk) * W(bk,bj)
  const int r = t & 15, c = t >> 4;  // element inside a 16 x 16 block
  for(int bi = 1; bi < 4; bi++) {
    for(int bj = 0; bj < bi; bj++) {
      double sum = 0.0;
      for(int k = bj * 16; k < bi * 16; k++) sum += L[(bi * 16 + r) + k * LDP] * W[k + (bj * 16 + c) * LDP];
      T[r + (bj * 16 + c) * 16] = sum;
    }
    __syncthreads();
    for(int bj = 0; bj < bi; bj++) {
      double sum = 0.0;
      for(int k = 0; k <= r; k++) sum += W[(bi * 16 + r) + (bi * 16 + k) * LDP] * T[k + (bj * 16 + c) * 16];
      W[(bi * 16 + r) + (bj * 16 + c) * LDP] = -sum;
    }
    __syncthreads();
  }
}

// Factor the n x n (n <= 64) diagonal block at A (lower, in place) and write inv(L) (64 x 64, ld 64, zero-padded
// identity) to `inv`.  col0 = global index of the block's first column, for `info`.
__global__ void __launch_bounds__(256) potf2_inv_kernel(double* __restrict__ A, int64_t lda, int n,
                                                        double* __restrict__ inv, int* __restrict__ info,
                                                        int64_t col0)
{
  __shared__ double S[JB * LDP];
  __shared__ double W[JB * LDP];
  __shared__ double T[16 * 48];
  const int t = threadIdx.x;
  if(*info != 0) return;  // an earlier block already failed: uniform exit

  for(int idx = t; idx < JB * JB; idx += 256) {
    const int i = idx & 63, j = idx >> 6;
    double v = 0.0;
    if(i < n && j < n) {
      if(i >= j) v = A[i + (int64_t)j * lda];
    } else if(i == j) {
      v = 1.0;
    }
    S[i + j * LDP] = v;
  }
  __syncthreads();

  const int il = t & 63, cg = t >> 6;
  for(int j = 0; j < n; j++) {
    const double ajj = S[j + j * LDP];
    if(!(ajj > 0.0)) {  // also catches NaN; uniform across the workgroup
      if(t == 0) atomicCAS(info, 0, (int)(col0 + j + 1));
      return;
    }
    const double rinv = 1.0 / ajj;
    if(il > j && il < n) {
      const double lij = S[il + j * LDP] * rinv;
      for(int c = j + 1 + cg; c <= il; c += 4) S[il + c * LDP] -= lij * S[c + j * LDP];
    }
    __syncthreads();
  }
  // scale the columns: L(j,j) = sqrt(pivot), L(i,j) = S(i,j)/sqrt(pivot)
  for(int idx = t; idx < JB * JB; idx += 256) {
    const int i = idx & 63, j = idx >> 6;
    if(i < n && j < n && i >= j) {
      const double d = sqrt(S[j + j * LDP]);
      const double v = (i == j) ? d : S[i + j * LDP] / d;
      W[i + j * LDP] = v;  // stash; S(j,j) is still needed by other threads
    }
  }
  __syncthreads();
  for(int idx = t; idx < JB * JB; idx += 256) {
    const int i = idx & 63, j = idx >> 6;
    if(i < n && j < n && i >= j) {
      const double v = W[i + j * LDP];
      S[i + j * LDP] = v;
      A[i + (int64_t)j * lda] = v;
    }
  }
  __syncthreads();
  tri_inverse_64(S, W, T);
  for(int idx = t; idx < JB * JB; idx += 256) {
    const int i = idx & 63, j = idx >> 6;
    inv[i + j * JB] = W[i + j * LDP];
  }
}

// Invert diagonal blocks of a triangular matrix (general trsm / potri support).  One workgroup per block.
__global__ void __launch_bounds__(256) tri_inv_blocks_kernel(const double* __restrict__ A, int64_t lda, int64_t N,
                                                             int lower, int unit, double* __restrict__ inv)
{
  __shared__ double S[JB * LDP];
  __shared__ double W[JB * LDP];
  __shared__ double T[16 * 48];
  const int t = threadIdx.x;
  const int64_t o = (int64_t)blockIdx.x * JB;
  const int n = (int)((N - o) < JB ? (N - o) : JB);
  const double* Ab = A + o + o * lda;
  // Load as a LOWER triangular matrix: an upper block is loaded transposed (inv(U) = inv(U')').
  for(int idx = t; idx < JB * JB; idx += 256) {
    const int i = idx & 63, j = idx >> 6;
    double v = 0.0;
    if(i < n && j < n) {
      if(i > j) v = lower ? Ab[i + (int64_t)j * lda] : Ab[j + (int64_t)i * lda];
      else if(i == j) v = unit ? 1.0 : Ab[i + (int64_t)i * lda];
    } else if(i == j) {
      v = 1.0;
    }
    S[i + j * LDP] = v;
  }
  __syncthreads();
  tri_inverse_64(S, W, T);
  double* out = inv + (int64_t)blockIdx.x * JB * JB;
  for(int idx = t; idx < JB * JB; idx += 256) {
    const int i = idx & 63, j = idx >> 6;
    out[i + j * JB] = lower ? W[i + j * LDP] : W[j + i * LDP];
  }
}

}  // namespace

int invert_diag_blocks(bool lower, bool unit, int64_t N, int64_t jb, const double* A, int64_t lda, double* inv,
                       hipStream_t s)
{
  if(jb != JB) {
    set_error("invert_diag_blocks: block size must be 64");
    return GPC_EINVAL;
  }
  if(N <= 0) return GPC_OK;
  const unsigned nblk = (unsigned)((N + JB - 1) / JB);
  hipLaunchKernelGGL(tri_inv_blocks_kernel, dim3(nblk), dim3(256), 0, s, A, lda, N, lower ? 1 : 0, unit ? 1 : 0,
                     inv);
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

static int64_t outer_nb()
{
  if(g_nb_outer <= 0) {
    const char* e = getenv("GPC_NB");
    int64_t v = e ? atoll(e) : 0;
    if(v < JB) v = 512;
    g_nb_outer = (v / JB) * JB;
  }
  return g_nb_outer;
}

int potrf_lower(int64_t N, double* A, int64_t lda, int* d_info, hipStream_t s)
{
  if(N <= 0) return GPC_OK;
  const int64_t NB = outer_nb();
  void* ws = nullptr;
  GPC_CHECK(workspace(WS_POTRF_INV, sizeof(double) * JB * JB * (size_t)((N + JB - 1) / JB), &ws));
  double* invs = static_cast<double*>(ws);

  for(int64_t k0 = 0; k0 < N; k0 += NB) {
    const int64_t nbk = (N - k0 < NB) ? (N - k0) : NB;
    const int64_t kend = k0 + nbk;
    for(int64_t j0 = k0; j0 < kend; j0 += JB) {
      const int64_t jb = (kend - j0 < JB) ? (kend - j0) : JB;
      double* Ajj = A + j0 + j0 * lda;
      double* inv = invs + (j0 / JB) * JB * JB;
      hipLaunchKernelGGL(potf2_inv_kernel, dim3(1), dim3(256), 0, s, Ajj, lda, (int)jb, inv, d_info, j0);
      GPC_HIP_CHECK(hipGetLastError());
      const int64_t below = N - (j0 + jb);
      if(below <= 0) continue;
      double* A21 = A + (j0 + jb) + j0 * lda;
      // L21 := A21 * inv(L11)'   (in place: C tile spans all jb <= 128 columns, rows are private to a workgroup)
      GPC_CHECK(gemm(false, true, below, jb, jb, 1.0, A21, lda, inv, JB, 0.0, A21, lda, 0, s));
      // update the not-yet-factored columns of this panel: lower trapezoid below the diagonal
      const int64_t nc = kend - (j0 + jb);
      if(nc > 0) {
        double* A22 = A + (j0 + jb) + (j0 + jb) * lda;
        GPC_CHECK(gemm(false, true, below, nc, jb, -1.0, A21, lda, A21, lda, 1.0, A22, lda, 3, s));
      }
    }
    const int64_t mt = N - kend;
    if(mt > 0) {
      const double* L21 = A + kend + k0 * lda;
      double* A22 = A + kend + kend * lda;
      prof_begin(PROF_SYRK, (double)mt * (double)(mt + 1) * (double)nbk, s);  // lower-triangle SYRK flops
      GPC_CHECK(gemm(false, true, mt, mt, nbk, -1.0, L21, lda, L21, lda, 1.0, A22, lda, 1, s));
      prof_end(PROF_SYRK, s);
    }
  }
  return GPC_OK;
}

}  // namespace gpc

extern "C" int gpc_set_potrf_blocking(int64_t nb_outer, int64_t jb_inner)
{
  if(jb_inner != 0 && jb_inner != gpc::JB) {
    gpc::set_error("inner block is fixed at 64 in this build");
    return GPC_EINVAL;
  }
  if(nb_outer < gpc::JB || nb_outer % gpc::JB != 0) {
    gpc::set_error("outer block must be a positive multiple of 64");
    return GPC_EINVAL;
  }
  gpc::g_nb_outer = nb_outer;
  return GPC_OK;
}
