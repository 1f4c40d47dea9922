// kern_grad.hip -- fused kernel-parameter gradient  g_p = sum_ij covGrad(i,j) * dK(i,j)/dtheta_p  for every natural
// parameter of a compound kernel, in one pass over covGrad (HBM-read bound: 8 N^2 bytes in).
//
// Replaces the scalar O(N^2 D) loops of CCmpndKern::getGradParams (CKern.cpp:284-298) over its components:
//   CRbfKern::getGradParams    (CKern.cpp:1204-1241): g_iw = -var * sum_{i<j} d2 k~ cg,  g_var = tr(cg) + 2 sum_{i<j} k~ cg
//   CRbfardKern::getGradParams (CKern.cpp:3359-3403): same two plus g_s_k = 2 iw var sum_{i>j} k~ cg (xi xj - xi^2/2 - xj^2/2)
//   CWhiteKern (735-739) tr(cg);  CBiasKern (1020-1024) sum(cg);  CLinKern (2369-2383) sum_ij cg x_i.x_j
// written here over ALL ordered pairs (covGrad is symmetric, so 2*sum_{i<j} == sum_{i!=j}).
// The chain rule to the transformed (optimiser) space, CKern::getGradTransParams (CKern.cpp:50-63), stays on the host.
//
// Tiling is the Gram kernel's (128 x 32 patch per tile, thread = 2 rows x 8 columns, covGrad read in contiguous
// 1 KiB runs).  A fixed-size grid strides over the tiles and writes per-workgroup partial sums that the host adds in
// a fixed order, so the result is deterministic.
#include "gpc_common.hpp"
#include "gpc_exp.hpp"
#include <vector>
#include <type_traits>
#include <string.h>
#include <string.h>
#include <stdlib.h>

namespace gpc {

namespace {

constexpr int TI = 128;
constexpr int TJ = 32;
constexpr int DC = 16;
constexpr int NP_MAIN = 16;  // rbf: 4 x (d2k, k); ard: (vk, k); lin; all; trace; 3 spare
constexpr int ARD_PASS = 32;

struct GradArgs {
  const double* X;
  const double* n1;
  const double* cg;
  int64_t ldx, ldc, N, D;
  int64_t tiles_i, tiles_j;
  // fused covGrad (kern_grad_sym_kernel only): cg points at invK and the kernel forms
  //   covGrad(i,j) = -0.5 * (nd * invK(i,j) - sum_o A(i,o) A(j,o))      (CGp::updateCovGradient, CGp.cpp:666-679, summed
  // over the nd outputs) on the fly, so the N x N covGrad is never written or read
  const double* A;
  int64_t lda;
  int nd;
  int all_general;   // symmetric kernels: 1 = ONE launch, every tile in the general form (small N: a second launch costs more than the
                     // predicate-free form saves); 0 = two launches, MODE 1 + MODE 2
};

template <int NW>
__device__ __forceinline__ double block_sum_n(double v, double* sh)
{
  for(int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if(lane == 0) sh[w] = v;
  __syncthreads();
  double r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  if(NW == 8) r += (sh[NW - 4] + sh[NW - 3]) + (sh[NW - 2] + sh[NW - 1]);
  return r;
}

__device__ __forceinline__ double block_sum(double v, double* sh)
{
  for(int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if(lane == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// Accumulate x_i.x_j (DOT) and sum_k s_k (x_ik - x_jk)^2 (ARD) for the thread's 2 x 8 patch of tile (i0, j0).
template <bool DOT, bool ARD>
__device__ __forceinline__ void tile_products(const KSpecDev& ks, const GradArgs& g, int64_t i0, int64_t j0,
                                              double* Xi, double* Xj, double* Ai, double* Aj, double (&dot)[2][8],
                                              double (&ard)[2][8])
{
  const int t = threadIdx.x;
  const int il = 2 * (t & 63), jl = 8 * (t >> 6);
#pragma unroll
  for(int a = 0; a < 2; a++)
#pragma unroll
    for(int b = 0; b < 8; b++) {
      dot[a][b] = 0.0;
      ard[a][b] = 0.0;
    }
  for(int64_t d0 = 0; d0 < g.D; d0 += DC) {
    const int dc = (int)((g.D - d0 < DC) ? (g.D - d0) : DC);
    __syncthreads();
    for(int idx = t; idx < DC * TI; idx += 256) {
      const int d = idx / TI, i = idx % TI;
      double v = 0.0;
      if(d < dc && i0 + i < g.N) v = g.X[(i0 + i) + (d0 + d) * g.ldx];
      if(DOT) Xi[idx] = v;
      if(ARD) Ai[idx] = v * sqrt(ks.ard_scale[0][(d0 + d) < GPC_MAX_ARD_DIM ? (d0 + d) : 0]);
    }
    for(int idx = t; idx < DC * TJ; idx += 256) {
      const int d = idx / TJ, j = idx % TJ;
      double v = 0.0;
      if(d < dc && j0 + j < g.N) v = g.X[(j0 + j) + (d0 + d) * g.ldx];
      if(DOT) Xj[idx] = v;
      if(ARD) Aj[idx] = v * sqrt(ks.ard_scale[0][(d0 + d) < GPC_MAX_ARD_DIM ? (d0 + d) : 0]);
    }
    __syncthreads();
    for(int d = 0; d < dc; d++) {
      if(DOT) {
        const double2_t xi = *reinterpret_cast<const double2_t*>(&Xi[d * TI + il]);
#pragma unroll
        for(int b = 0; b < 8; b++) {
          const double xj = Xj[d * TJ + jl + b];
          dot[0][b] = fma(xi.x, xj, dot[0][b]);
          dot[1][b] = fma(xi.y, xj, dot[1][b]);
        }
      }
      if(ARD) {
        const double2_t xi = *reinterpret_cast<const double2_t*>(&Ai[d * TI + il]);
#pragma unroll
        for(int b = 0; b < 8; b++) {
          const double xj = Aj[d * TJ + jl + b];
          const double e0 = xi.x - xj, e1 = xi.y - xj;
          ard[0][b] = fma(e0, e0, ard[0][b]);
          ard[1][b] = fma(e1, e1, ard[1][b]);
        }
      }
    }
  }
}

template <bool DOT, bool ARD>
__global__ void __launch_bounds__(256) kern_grad_kernel(const KSpecDev ks, const GradArgs g,
                                                        double* __restrict__ partial)
{
  __shared__ __attribute__((aligned(16))) double Xi[DC * TI];
  __shared__ __attribute__((aligned(16))) double Xj[DC * TJ];
  __shared__ __attribute__((aligned(16))) double Ai[ARD ? DC * TI : 2];
  __shared__ __attribute__((aligned(16))) double Aj[ARD ? DC * TJ : 2];
  __shared__ double sh[4];
  const int t = threadIdx.x;
  const int il = 2 * (t & 63), jl = 8 * (t >> 6);

  double acc[NP_MAIN];
#pragma unroll
  for(int p = 0; p < NP_MAIN; p++) acc[p] = 0.0;

  const int64_t total = g.tiles_i * g.tiles_j;
  for(int64_t tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int64_t i0 = (tile % g.tiles_i) * TI, j0 = (tile / g.tiles_i) * TJ;
    double dot[2][8], ard[2][8];
    tile_products<DOT, ARD>(ks, g, i0, j0, Xi, Xj, Ai, Aj, dot, ard);
    const int64_t gi = i0 + il;
    double ni[2] = {0.0, 0.0};
    if(DOT) {
      if(gi < g.N) ni[0] = g.n1[gi];
      if(gi + 1 < g.N) ni[1] = g.n1[gi + 1];
    }
#pragma unroll
    for(int b = 0; b < 8; b++) {
      const int64_t gj = j0 + jl + b;
      if(gj >= g.N) continue;
      const double nj = DOT ? g.n1[gj] : 0.0;
#pragma unroll
      for(int a = 0; a < 2; a++) {
        if(gi + a >= g.N) continue;
        const double c = g.cg[(gi + a) + gj * g.ldc];
        acc[12] += c;  // bias: sum of all
        if(gi + a == gj) {
          acc[13] += c;  // trace
          if(DOT) acc[11] += c * ni[a];  // lin: x_i . x_i
        } else {
          if(DOT) {
            const double d2 = ni[a] + nj - 2.0 * dot[a][b];
#pragma unroll
            for(int r = 0; r < 4; r++) {
              if(r < ks.n_rbf) {
                const double e = c * exp(-(ks.rbf_hiw[r] * d2));
                acc[2 * r] += d2 * e;
                acc[2 * r + 1] += e;
              }
            }
            acc[11] += c * dot[a][b];
          }
          if(ARD) {
            const double e = c * exp(-(ard[a][b] * ks.ard_hiw[0]));
            acc[8] += ard[a][b] * e;
            acc[9] += e;
          }
        }
      }
    }
  }
#pragma unroll
  for(int p = 0; p < NP_MAIN; p++) {
    const double r = block_sum(acc[p], sh);
    if(t == 0) partial[(int64_t)blockIdx.x * NP_MAIN + p] = r;
  }
}

// ---- symmetric fast path: kernels without an ARD term, D <= 32 -----------------------------------------------------------
// covGrad is symmetric (CGp::updateCovGradient mirrors it, CGp.cpp:666-679), so only the tiles left of and inside a row
// block's diagonal block are visited: half the covGrad bytes, half the dot products and exponentials, tiles strictly
// left of the diagonal block counting twice.  The structure is gram_sym_kernel's (gram.hip): the row block's MFMA
// operand fragments stay in registers for the whole walk, the 64-column tile is double-buffered in LDS with the next
// one's rows in flight, x_i.x_j comes out of v_mfma_f64_16x16x4 -- and where that kernel stores K(i,j), this one loads
// covGrad(i,j) (same access pattern: 128-byte runs along i) and accumulates the per-parameter sums.
// HBM-read bound: 4 N^2 bytes.  Partial sums per workgroup in the NP_MAIN layout of kern_grad_kernel, added on the host
// in a fixed order (deterministic).
constexpr int GMI = 128, GMJ = 64, GSJ = 80, GMDC = 32, GSI = 144;
// From this many k-steps (D > 12) the kernel does not request the next half-tile's covGrad values while it works on the current
// one: holding them costs 32 registers, which at NK >= 4 meant scratch; the CU's other workgroup covers the latency instead
// (N = 65 536: D = 16 6.4 -> 5.8 ms, D = 32 7.4 -> 6.9 ms; below that the early request still wins by a few per cent).
#ifndef GPC_KG_TABLE_EXP
#define GPC_KG_TABLE_EXP 1
#endif
#ifndef GPC_KG_LEAN_NK
#define GPC_KG_LEAN_NK 4
#endif
#ifndef GPC_KG_TABLE_MAXNK      // the table exponential up to this many k-steps (D <= 8), ocml's above (A/B builds move it)
#define GPC_KG_TABLE_MAXNK 2
#endif
#ifndef GPC_KG_XPREF_MINNK      // the next tile's first half of covGrad requested a half ahead from this many k-steps on
#define GPC_KG_XPREF_MINNK 3
#endif
typedef double gdouble4 __attribute__((ext_vector_type(4)));
typedef double gdouble2 __attribute__((ext_vector_type(2)));
// Which row of a wave's patch lane l holds in its 16-row MFMA tile tm.  The matrix instruction does not care (any permutation of
// the patch's rows, applied to the operand fragments, the row norms and covGrad alike); with tiles 2t, 2t + 1 interleaved -- rows
// 2 (l & 15) and 2 (l & 15) + 1 of a 32-row group -- the pair's operand fragment is ONE ds_read_b128 and the pair's two covGrad loads share
// their cache lines (N = 65 536, same box, against tile-major rows: rbf D = 4 / 8 / 16 / 32 4.33 / 4.51 / 5.54 / 6.64 -> 4.17 / 4.24 /
// 5.35 / 6.45 ms, rbfard 5.43 / 5.66 / 7.29 / 9.66 -> 5.33 / 5.50 / 7.15 / 9.36).  The same pairing for covGrad itself -- one global_load_dwordx4 for a lane's two
// values instead of two dwordx2 -- was built and LOSES heavily (rbf D = 8 / 32: 4.99 / 6.84 against 4.22 / 6.33 ms; rbfard 8.83 /
// 10.66 against 5.36 / 9.04): it stays two 8-byte loads.  -DGPC_KG_OLDROW restores tile-major rows (A/B builds).
#ifdef GPC_KG_OLDROW
#define KG_ROW(tm) ((tm) * 16 + (lane & 15))
#else
// (MODE 2 -- the diagonal blocks' general tiles -- keeps tile-major rows: interleaved, its instances spill 15-40 registers more, which
//  shows where those tiles are most of the walk: N = 8192, D = 8 0.165 -> 0.208 ms.  The two modes are separate launches with separate
//  partial sums, so each may deal its rows as it likes.)
#define KG_ROW(tm) (MODE == 1 ? (32 * ((tm) >> 1) + 2 * (lane & 15) + ((tm) & 1)) : ((tm) * 16 + (lane & 15)))
#endif

// MODE (round 4): the walk is two launches.  MODE 1 takes the full tiles strictly left of a row block's diagonal block -- all but
// two tiles of its walk --, which need no edge masks, no diagonal test and carry weight 2: ONE predicate-free form of the tile
// body.  MODE 2 takes the rest (the diagonal block's two tiles; every tile of a ragged last row block) in the general form.  As two
// copies of the body inside one kernel the D > 8 instances spilled 200+ registers; as two kernels each sits in its own budget.
template <int NRBF, int NK, int ND = 0, int MODE = 1>
__global__ void __launch_bounds__(256, 2) kern_grad_sym_kernel(const KSpecDev ks, const GradArgs g, int jt_per_block,
                                                               double* __restrict__ partial)
{
  __shared__ double Xj[2][GMDC * GSJ];
  __shared__ double Nj[2][GMJ];
  __shared__ __attribute__((aligned(16))) double Xi[NK > 2 ? GMDC * GSI : 1];
  __shared__ double sh[4];
  __shared__ double Etab[64];
  gpc_exp_tab_fill(Etab);      // the table of gpc_exp.hpp; published by the first tile's barrier
  const int t = threadIdx.x;
  const int lane = t & 63, w = t >> 6;
  const int wm = w & 1, wn = w >> 1;
  const int64_t i0 = (int64_t)blockIdx.x * GMI;
  double* mypartial = partial + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * NP_MAIN;
  int64_t tiles_j = (g.N + GMJ - 1) / GMJ;
  if(tiles_j > 2 * ((int64_t)blockIdx.x + 1)) tiles_j = 2 * ((int64_t)blockIdx.x + 1);
  int64_t split = (i0 + GMI <= g.N && !g.all_general) ? 2 * (int64_t)blockIdx.x : 0;   // tiles [0, split): full and strictly left of the diagonal block
  if(split > tiles_j) split = tiles_j;
  const int64_t lo = (MODE == 1) ? 0 : split, hi = (MODE == 1) ? split : tiles_j;
  const int64_t jt0 = lo + (int64_t)blockIdx.y * jt_per_block;
  int64_t jt1 = jt0 + jt_per_block;
  if(jt1 > hi) jt1 = hi;
  if(jt0 >= jt1) {
    if(t < NP_MAIN) mypartial[t] = 0.0;
    return;
  }
  const int dc = (int)g.D;   // <= 4 NK

  // the row block's MFMA operand fragments: in registers for D <= 8; beyond that they would push the kernel into scratch
  // (32 doubles per lane), so there they live in LDS (Xi[k * GSI + row], the Gram kernel's conflict-free stride)
  constexpr bool AF_LDS = (NK > 2);
  double af[AF_LDS ? 1 : NK][4];
  if(AF_LDS) {
#pragma unroll
    for(int u = 0; u < (GMDC * GMI) / 256; u++) {
      const int idx = t + 256 * u;
      const int kr = idx >> 7, row = idx & 127;
      int64_t gi = i0 + row;
      if(gi > g.N - 1) gi = g.N - 1;
      Xi[kr * GSI + row] = (kr < dc) ? g.X[gi + (int64_t)kr * g.ldx] : 0.0;
    }
  } else {
#pragma unroll
    for(int kk = 0; kk < (AF_LDS ? 1 : NK); kk++)
#pragma unroll
      for(int tm = 0; tm < 4; tm++) {
        int64_t gi = i0 + wm * 64 + KG_ROW(tm);
        if(gi > g.N - 1) gi = g.N - 1;
        int kr = kk * 4 + (lane >> 4);
        if(kr > dc - 1) kr = dc - 1;
        af[kk][tm] = g.X[gi + (int64_t)kr * g.ldx];
      }
  }
  double ni[4];
  double ai[ND > 0 ? ND : 1][4];   // fused covGrad: A(i, o) of this lane's four rows
#pragma unroll
  for(int tm = 0; tm < 4; tm++) {
    int64_t gi = i0 + wm * 64 + KG_ROW(tm);
    if(gi > g.N - 1) gi = g.N - 1;
    ni[tm] = g.n1[gi];
#pragma unroll
    for(int o = 0; o < ND; o++) ai[o][tm] = g.A[gi + (int64_t)o * g.lda];
  }

  const int ws = __builtin_amdgcn_readfirstlane(w);
  double vj[8], vn;
  auto prefetch = [&](int64_t jt) {
    int64_t gj = jt * GMJ + lane;
    if(gj > g.N - 1) gj = g.N - 1;
#pragma unroll
    for(int u = 0; u < 8; u++) {
      const int d = ws + 4 * u;
      vj[u] = 0.0;
      if(d < dc) vj[u] = (g.X + (int64_t)d * g.ldx)[gj];
    }
    vn = g.n1[gj];
  };
  prefetch(jt0);

  double s_d2e[NRBF > 0 ? NRBF : 1], s_e[NRBF > 0 ? NRBF : 1], s_lin = 0.0, s_all = 0.0, s_tr = 0.0;
#pragma unroll
  for(int q = 0; q < (NRBF > 0 ? NRBF : 1); q++) s_d2e[q] = s_e[q] = 0.0;

  // covGrad values of a half-tile (16 per lane), requested a half ahead.  MODE 1 keeps that distance ACROSS tiles as well -- the
  // next tile's first half is requested during this tile's second (round 3 asked for it at the top of the tile, i.e. ~0.5 us
  // before its use: every tile began with an exposed HBM round trip, which is what held these kernels at 1.7-2.5 TB/s)
  constexpr bool KFAST = (MODE == 1);
    constexpr bool LEAN = !KFAST && (NK >= GPC_KG_LEAN_NK);   // MODE 2, D > 12: no covGrad held for the next half
    double c[LEAN ? 1 : 2][4][4];
    auto load_cg = [&](int tn, const int64_t j0) {
      const bool full = KFAST || ((i0 + GMI <= g.N) && (j0 + GMJ <= g.N));
#pragma unroll
      for(int r = 0; r < 4; r++) {
        const int64_t gj = j0 + wn * 32 + tn * 16 + (lane >> 4) + 4 * r;
        const int64_t gjc = (KFAST || gj < g.N) ? gj : (g.N - 1);
        double aj[ND > 0 ? ND : 1];
#pragma unroll
        for(int o = 0; o < ND; o++) aj[o] = g.A[gjc + (int64_t)o * g.lda];
#pragma unroll
        for(int tm = 0; tm < 4; tm++) {
          const int64_t gi = i0 + wm * 64 + KG_ROW(tm);
          const int64_t gic = (KFAST || gi < g.N) ? gi : (g.N - 1);
#ifdef GPC_KG_ABL_NOLOAD
          double v = (double)(gic & 7);
#else
          double v = g.cg[gic + gjc * g.ldc];
#endif
          if(ND > 0) {   // covGrad from invK: same operations as covgrad_kernel / covgrad_multi_kernel, element by element
            double aa = 0.0;
#pragma unroll
            for(int o = 0; o < ND; o++) aa = fma(ai[o][tm], aj[o], aa);
            v = -0.5 * ((double)ND * v - aa);
          }
          c[LEAN ? 0 : tn][r][tm] = (KFAST || full || (gi < g.N && gj < g.N)) ? v : 0.0;
        }
      }
    };
  // (D <= 8: with the table exponential the two live halves across the loop edge spill 200 registers -- 4.3 -> 22 ms --, so there
  //  the first half of a tile is requested at its top as before)
  constexpr bool XPREF = KFAST && !LEAN && NK >= GPC_KG_XPREF_MINNK;
  if(XPREF) load_cg(0, jt0 * GMJ);

  for(int64_t jt = jt0; jt < jt1; jt++) {
    double* Xjb = Xj[(jt - jt0) & 1];
    double* Njb = Nj[(jt - jt0) & 1];
    const int64_t j0 = jt * GMJ;
#pragma unroll
    for(int u = 0; u < 8; u++) {
      const int idx = t + 256 * u;
      Xjb[(idx >> 6) * GSJ + (idx & 63)] = vj[u];
    }
    if(t < GMJ) Njb[t] = vn;
    __syncthreads();
    if(jt + 1 < jt1) prefetch(jt + 1);

    // One 16-column half of the wave's patch at a time (products, then sums), so that only 16 accumulators and 16 + 16
    // covGrad values are live: the full 64 x 32 patch at once needs more registers than a wave has at D = 32.  The
    // covGrad values of a half are requested one half ahead: their latency hides behind the other half's work.
    const bool mirror = (j0 + GMJ <= i0);   // strictly left of the diagonal block: every element stands for two
    // FAST (round 4): a full tile strictly left of the diagonal block -- all but two tiles of a walk -- has no edge masks and no
    // diagonal elements: no selects around the loads, no 64-bit compares in the sums, weight 2 folded into the final sums
    // MODE 1 = the fast form (KG_FAST) with the table-driven exponential of gpc_exp.hpp; MODE 2 the general form with ocml's
#ifdef GPC_KG_ABL_NOEXP      // (timing-only builds, tools/kgrad_abl.sh: -DGPC_KG_ABL_NOEXP / _NOLOAD / _NOMMA take one ingredient out)
#define KG_EXP(x) (x)
#else
#define KG_EXP(x) ((MODE == 1 && NK <= GPC_KG_TABLE_MAXNK && GPC_KG_TABLE_EXP) ? gpc_exp_tab((x), Etab) : exp(x))
#endif
#define KG_FAST (MODE == 1)
#include "kern_grad_sym_tile.inc"
#undef KG_FAST
#undef KG_EXP
  }
  double out[NP_MAIN];
#pragma unroll
  for(int p = 0; p < NP_MAIN; p++) out[p] = 0.0;
#pragma unroll
  for(int q = 0; q < NRBF; q++) {
    out[2 * q] = s_d2e[q];
    out[2 * q + 1] = s_e[q];
  }
  out[11] = s_lin;
  out[12] = s_all;
  out[13] = s_tr;
#pragma unroll
  for(int p = 0; p < NP_MAIN; p++) {
    if(p < 2 * NRBF || (p >= 11 && p <= 13)) {
      const double rsum = block_sum(out[p], sh);
      if(t == 0) mypartial[p] = rsum;
    } else if(t == 0) {
      mypartial[p] = 0.0;
    }
  }
}

// MODE 2's launch: per row block the two tiles of its diagonal block -- and, when N is not a multiple of 128, every tile of the ragged
// last row block, which therefore gets a workgroup per two tiles (one workgroup walking all of them was 80 us of a 0.1 ms pass at
// N = 1000).  Returns the slices per row block.
constexpr int KG_MODE2_PER = 2;
static inline int64_t kg_mode2_ny(int64_t N)
{
  const int64_t tiles_all = (N + GMJ - 1) / GMJ;
  return (N % GMI != 0) ? (tiles_all + KG_MODE2_PER - 1) / KG_MODE2_PER : 1;
}

// two launches: the fast tiles on `grid` (partials at partial[0 .. grid.x * grid.y)), the diagonal blocks / ragged row block on
// grid.x x kg_mode2_ny workgroups (partials behind them)
template <int NRBF, int ND>
int launch_grad_sym_nd(const KSpecDev& ks, const GradArgs& g, int per, dim3 grid, double* partial, hipStream_t s)
{
  double* p2 = partial + (size_t)grid.x * grid.y * NP_MAIN;
  const dim3 grid2(grid.x, (unsigned)kg_mode2_ny(g.N));
  const int per2 = KG_MODE2_PER;
#define GPC_SYM_LAUNCH2(NKV)                                                                                                  \
  do {                                                                                                                        \
    if(g.all_general) {                                                                                                       \
      hipLaunchKernelGGL((kern_grad_sym_kernel<NRBF, NKV, ND, 2>), grid, dim3(256), 0, s, ks, g, per, partial);               \
    } else {                                                                                                                  \
      hipLaunchKernelGGL((kern_grad_sym_kernel<NRBF, NKV, ND, 1>), grid, dim3(256), 0, s, ks, g, per, partial);               \
      hipLaunchKernelGGL((kern_grad_sym_kernel<NRBF, NKV, ND, 2>), grid2, dim3(256), 0, s, ks, g, per2, p2);                  \
    }                                                                                                                         \
  } while(0)
  if(g.D <= 4) GPC_SYM_LAUNCH2(1);
  else if(g.D <= 8) GPC_SYM_LAUNCH2(2);
  else if(g.D <= 16) GPC_SYM_LAUNCH2(4);
  else GPC_SYM_LAUNCH2(8);
#undef GPC_SYM_LAUNCH2
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

template <int NRBF>
int launch_grad_sym(const KSpecDev& ks, const GradArgs& g, int per, dim3 grid, double* partial, hipStream_t s)
{
  if(g.nd == 0) return launch_grad_sym_nd<NRBF, 0>(ks, g, per, grid, partial, s);
  if(g.nd == 1) return launch_grad_sym_nd<NRBF, 1>(ks, g, per, grid, partial, s);
  return launch_grad_sym_nd<NRBF, 2>(ks, g, per, grid, partial, s);
}

// ---- symmetric fast path for ONE rbfard term (+ white / bias), D <= 32 ------------------------------------------------------
// CRbfardKern::getGradParams (CKern.cpp:3359-3403) needs, besides the two sums of the rbf kernel, one sum per input
// dimension:  S_q = sum_{i != j} W(i,j) (x_iq - x_jq)^2,  W = covGrad o k~.  On the symmetric walk above these are
//     S_q = sum_i x_iq^2 rho_i  +  sum_i Y2(q,i)  -  2 sum_i x_iq Y(q,i),
//     rho_i = sum_j W(i,j),   Y(q,i) = sum_j W(i,j) x_jq,   Y2(q,i) = sum_j W(i,j) x_jq^2      (j over the walked tiles,
// W carrying the tile's weight 2 / 1), and Y, Y2 are matrix products whose W operand needs no data movement at all:
// register r of a 16 x 16 accumulator tile holds W(i = lane & 15, j = 4 r + (lane >> 4)), which IS the MFMA operand
// layout for a k-step over j = 4 r .. 4 r + 3.  So the weights go from the exp epilogue straight back into
// v_mfma_f64_16x16x4 against rows of X^T (16 q per operand, loaded from a transposed copy, squares formed in registers);
// the mirrored contribution is never needed because only the scalars x_q' W x_q are.  The inputs are centred (differences
// do not change) and scaled by sqrt(s_q) first, so that x.x' from the same MFMA walk gives the ARD distance and the three
// terms of S_q do not cancel more than the data's spread demands; S_q comes out in scaled coordinates and the host divides
// by s_q.  Partial sums per workgroup: the NP_MAIN layout (slots 8, 9, 12, 13) followed by 32 S_q.
constexpr int NP_ARDSYM = NP_MAIN + 32;

__global__ void __launch_bounds__(256) ard_mean_kernel(const double* __restrict__ X, int64_t ldx, int64_t N, double* __restrict__ mean)
{
  __shared__ double sh[4];
  const int64_t q = blockIdx.x;
  double a = 0.0;
  for(int64_t i = threadIdx.x; i < N; i += 256) a += X[i + q * ldx];
  const double tot = block_sum(a, sh);
  if(threadIdx.x == 0) mean[q] = tot / (double)N;
}

// Xs = (X - mean) sqrt(s) column-major (ld N), XT the same transposed (row i at XT + i dp, zero-padded to dp), n1 = |xs_i|^2
__global__ void __launch_bounds__(256) ard_prep_kernel(const KSpecDev ks, const double* __restrict__ X, int64_t ldx, int64_t N, int D,
                                                       const double* __restrict__ mean, double* __restrict__ Xs,
                                                       double* __restrict__ XT, int dp, double* __restrict__ n1)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if(i >= N) return;
  double acc = 0.0;
  for(int q = 0; q < dp; q++) {
    double v = 0.0;
    if(q < D) {
      v = (X[i + (int64_t)q * ldx] - mean[q]) * sqrt(ks.ard_scale[0][q]);
      Xs[i + (int64_t)q * N] = v;
      acc = fma(v, v, acc);
    }
    XT[i * dp + q] = v;
  }
  n1[i] = acc;
}

// NW waves per workgroup: 4 (each a 64 x 32 patch of the 128 x 64 tile) or 8 (32 x 32 each: half the per-lane state -- the
// accumulators Y, the covGrad values, the dot products -- so that at D > 16, where the 4-wave form fits one wave per SIMD
// only, two waves share a SIMD without spilling and cover each other's memory latency).
template <int NK, int ND, int OCC, int NW = 4, int MODE = 1>   // MODE: as kern_grad_sym_kernel
__global__ void __launch_bounds__(64 * NW, OCC) kern_grad_ard_sym_kernel(const KSpecDev ks, const GradArgs g, const double* __restrict__ XT,
                                                                   int jt_per_block, double* __restrict__ partial)
{
  constexpr int QX = (NK > 4) ? 2 : 1;          // 16-wide groups of input dimensions
  constexpr int DP = 16 * QX;
  __shared__ double Xj[2][GMDC * GSJ];
  // The column tile a second time, transposed ([column][dimension]): the operand of the Y product, x_jq for lane (j, q).  Round 3
  // fetched it from a row-major copy of X in global memory (eight L2 round trips per half-tile and lane: 1 ms of 9.9 at D = 32);
  // the values are in this workgroup's LDS anyway.  Row stride 17 / 49 doubles: conflict-free for the staging writes (a lane per
  // column) and, to one collision per half-wave, for the fragment reads.
  constexpr int XTS = (DP == 16) ? 17 : 49;
  __shared__ double XjT[2][GMJ * XTS];
  __shared__ double Nj[2][GMJ];
  __shared__ __attribute__((aligned(16))) double Xi[NK > 2 ? GMDC * GSI : 1];
  constexpr int NT = 64 * NW;           // threads
  constexpr int RW = 256 / NW;          // rows of a wave's patch: 64 or 32
  constexpr int TM = RW / 16;           // its 16-row MFMA tiles: 4 or 2
  static_assert(NW == 4 || NW == 8, "waves per workgroup");
  __shared__ double sh[NW];
  __shared__ double red[NW][32];
  __shared__ double Etab[64];
  gpc_exp_tab_fill(Etab);      // the table of gpc_exp.hpp; published by the first tile's barrier
  const int t = threadIdx.x;
  const int lane = t & 63, w = t >> 6;
  const int wm = w & (NW / 2 - 1), wn = w / (NW / 2);
  const int64_t i0 = (int64_t)blockIdx.x * GMI;
  double* mypartial = partial + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * NP_ARDSYM;
  int64_t tiles_j = (g.N + GMJ - 1) / GMJ;
  if(tiles_j > 2 * ((int64_t)blockIdx.x + 1)) tiles_j = 2 * ((int64_t)blockIdx.x + 1);
  int64_t split = (i0 + GMI <= g.N && !g.all_general) ? 2 * (int64_t)blockIdx.x : 0;   // tiles [0, split): full and strictly left of the diagonal block
  if(split > tiles_j) split = tiles_j;
  const int64_t lo = (MODE == 1) ? 0 : split, hi = (MODE == 1) ? split : tiles_j;
  const int64_t jt0 = lo + (int64_t)blockIdx.y * jt_per_block;
  int64_t jt1 = jt0 + jt_per_block;
  if(jt1 > hi) jt1 = hi;
  if(jt0 >= jt1) {
    if(t < NP_ARDSYM) mypartial[t] = 0.0;
    return;
  }
  const int dc = (int)g.D;   // <= 4 NK

  constexpr bool AF_LDS = (NK > 2);
  double af[AF_LDS ? 1 : NK][TM];
  if(AF_LDS) {
#pragma unroll
    for(int u = 0; u < (GMDC * GMI) / NT; u++) {
      const int idx = t + NT * u;
      const int kr = idx >> 7, row = idx & 127;
      int64_t gi = i0 + row;
      if(gi > g.N - 1) gi = g.N - 1;
      Xi[kr * GSI + row] = (kr < dc) ? g.X[gi + (int64_t)kr * g.ldx] : 0.0;
    }
  } else {
#pragma unroll
    for(int kk = 0; kk < (AF_LDS ? 1 : NK); kk++)
#pragma unroll
      for(int tm = 0; tm < TM; tm++) {
        int64_t gi = i0 + wm * RW + KG_ROW(tm);
        if(gi > g.N - 1) gi = g.N - 1;
        int kr = kk * 4 + (lane >> 4);
        if(kr > dc - 1) kr = dc - 1;
        af[kk][tm] = (kk * 4 + (lane >> 4) < dc) ? g.X[gi + (int64_t)kr * g.ldx] : 0.0;
      }
  }
  double ni[TM];
  double ai[ND > 0 ? ND : 1][TM];
#pragma unroll
  for(int tm = 0; tm < TM; tm++) {
    int64_t gi = i0 + wm * RW + KG_ROW(tm);
    if(gi > g.N - 1) gi = g.N - 1;
    ni[tm] = g.n1[gi];
#pragma unroll
    for(int o = 0; o < ND; o++) ai[o][tm] = g.A[gi + (int64_t)o * g.lda];
  }

  const int ws = __builtin_amdgcn_readfirstlane(w);
  constexpr int VJ = (GMDC * GMJ) / NT;   // staged values of the column tile per thread: 8 or 4
  double vj[VJ], vn;
  auto prefetch = [&](int64_t jt) {
    int64_t gj = jt * GMJ + lane;
    if(gj > g.N - 1) gj = g.N - 1;
#pragma unroll
    for(int u = 0; u < VJ; u++) {
      const int d = ws + NW * u;
      vj[u] = 0.0;
      if(d < dc) vj[u] = (g.X + (int64_t)d * g.ldx)[gj];
    }
    vn = g.n1[gj];
  };
  prefetch(jt0);

  double s_d2e = 0.0, s_e = 0.0, s_all = 0.0, s_tr = 0.0;
  double rho[TM];
#pragma unroll
  for(int tm = 0; tm < TM; tm++) rho[tm] = 0.0;
  gdouble4 Y1[TM][QX];
  double Bq[QX];             // sum_j kappa_j x_jq^2 for q = (lane & 15) + 16 qx, over this lane's columns j = 4 r + (lane >> 4)
#pragma unroll
  for(int qx = 0; qx < QX; qx++) Bq[qx] = 0.0;
#pragma unroll
  for(int tm = 0; tm < TM; tm++)
#pragma unroll
    for(int qx = 0; qx < QX; qx++) Y1[tm][qx] = (gdouble4){0.0, 0.0, 0.0, 0.0};
  const double hiw = ks.ard_hiw[0];

  // covGrad values of a half-tile, requested a half ahead -- in MODE 1 across tiles as well (see kern_grad_sym_kernel)
  constexpr bool KFAST = (MODE == 1);
    constexpr bool LEAN = (OCC == 2);   // two workgroups per CU: no covGrad held for the next half (the other workgroup's waves cover the latency)
    double c[LEAN ? 1 : 2][4][TM];
    auto load_cg = [&](int tn, const int64_t j0) {
      const bool full = KFAST || ((i0 + GMI <= g.N) && (j0 + GMJ <= g.N));
#pragma unroll
      for(int r = 0; r < 4; r++) {
        const int64_t gj = j0 + wn * 32 + tn * 16 + (lane >> 4) + 4 * r;
        const int64_t gjc = (KFAST || gj < g.N) ? gj : (g.N - 1);
        double aj[ND > 0 ? ND : 1];
#pragma unroll
        for(int o = 0; o < ND; o++) aj[o] = g.A[gjc + (int64_t)o * g.lda];
#pragma unroll
        for(int tm = 0; tm < TM; tm++) {
          const int64_t gi = i0 + wm * RW + KG_ROW(tm);
          const int64_t gic = (KFAST || gi < g.N) ? gi : (g.N - 1);
#ifdef GPC_KG_ABL_NOLOAD
          double v = (double)(gic & 7);
#else
          double v = g.cg[gic + gjc * g.ldc];
#endif
          if(ND > 0) {
            double aa = 0.0;
#pragma unroll
            for(int o = 0; o < ND; o++) aa = fma(ai[o][tm], aj[o], aa);
            v = -0.5 * ((double)ND * v - aa);
          }
          c[LEAN ? 0 : tn][r][tm] = (KFAST || full || (gi < g.N && gj < g.N)) ? v : 0.0;
        }
      }
    };
  constexpr bool XPREF = KFAST && !LEAN;
  if(XPREF) load_cg(0, jt0 * GMJ);

  for(int64_t jt = jt0; jt < jt1; jt++) {
    double* Xjb = Xj[(jt - jt0) & 1];
    double* Njb = Nj[(jt - jt0) & 1];
    const int64_t j0 = jt * GMJ;
    double* XjTb = XjT[(jt - jt0) & 1];
#pragma unroll
    for(int u = 0; u < VJ; u++) {
      const int idx = t + NT * u;
      Xjb[(idx >> 6) * GSJ + (idx & 63)] = vj[u];
      if((idx >> 6) < DP) XjTb[(idx & 63) * XTS + (idx >> 6)] = vj[u];
    }
    if(t < GMJ) Njb[t] = vn;
    __syncthreads();
    if(jt + 1 < jt1) prefetch(jt + 1);

    const bool mirror = (j0 + GMJ <= i0);
    // FAST (round 4): full tiles strictly left of the diagonal block -- no edge masks, no diagonal elements, weight 2
#ifdef GPC_KG_ABL_NOEXP
#define KG_EXP(x) (x)
#else
#define KG_EXP(x) (MODE == 1 ? gpc_exp_tab((x), Etab) : exp(x))
#endif
#define KG_FAST (MODE == 1)
#include "kern_grad_ard_tile.inc"
#undef KG_FAST
#undef KG_EXP
  }

  // ---- the walk is over: scalars as in kern_grad_sym_kernel, then S_q -----------------------------------------------------
  double out[NP_MAIN];
#pragma unroll
  for(int p = 0; p < NP_MAIN; p++) out[p] = 0.0;
  out[8] = s_d2e;
  out[9] = s_e;
  out[12] = s_all;
  out[13] = s_tr;
#pragma unroll
  for(int p = 0; p < NP_MAIN; p++) {
    if(p == 8 || p == 9 || p == 12 || p == 13) {
      const double rsum = block_sum_n<NW>(out[p], sh);
      if(t == 0) mypartial[p] = rsum;
    } else if(t == 0) {
      mypartial[p] = 0.0;
    }
  }
  // rho_i over the four column groups of the wave (lanes with the same lane & 15)
#pragma unroll
  for(int tm = 0; tm < TM; tm++) {
    rho[tm] += __shfl_xor(rho[tm], 16, 64);
    rho[tm] += __shfl_xor(rho[tm], 32, 64);
  }
  // sum_i (rho_i x_iq^2 - 2 x_iq Y(q,i)): this lane's dimensions are q = (lane >> 4) + 4 r + 16 qx, its rows lane & 15 of each of
  // the wave's four 16-row tiles
  if(t < 32 * NW) red[t >> 5][t & 31] = 0.0;
  __syncthreads();
#pragma unroll
  for(int qx = 0; qx < QX; qx++)
#pragma unroll
    for(int r = 0; r < 4; r++) {
      const int q = (lane >> 4) + 4 * r + 16 * qx;
      double v = 0.0;
#pragma unroll
      for(int tm = 0; tm < TM; tm++) {
        int64_t gi = i0 + wm * RW + KG_ROW(tm);
        if(gi > g.N - 1) gi = g.N - 1;                              // (rows past the end carry zero weights)
        const double x = (q < dc) ? g.X[gi + (int64_t)q * g.ldx] : 0.0;
        v += x * fma(rho[tm], x, -2.0 * Y1[tm][qx][r]);
      }
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      if((lane & 15) == 0) red[w][q] = v;
      asm volatile("" ::: "memory");   // one group of loads at a time: hoisted together they would set the kernel's register count
    }
  __syncthreads();
  // ... + sum_j kappa_j x_jq^2: lane (lane >> 4, q = lane & 15) holds its columns' share; the four column groups of the wave are
  // added in a fixed order by the lane of group 0
#pragma unroll
  for(int qx = 0; qx < QX; qx++) {
    double v = Bq[qx];
    const double v1 = __shfl(v, (lane & 15) + 16, 64), v2 = __shfl(v, (lane & 15) + 32, 64), v3 = __shfl(v, (lane & 15) + 48, 64);
    if(lane < 16) red[w][lane + 16 * qx] += (v + v1) + (v2 + v3);
  }
  __syncthreads();
  if(t < 32) {
    double v = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    if(NW == 8) v += (red[NW - 4][t] + red[NW - 3][t]) + (red[NW - 2][t] + red[NW - 1][t]);
    mypartial[NP_MAIN + t] = v;
  }
}

template <int ND>
int launch_grad_ard_sym(const KSpecDev& ks, const GradArgs& g, const double* XT, int per, dim3 grid, double* partial, hipStream_t s)
{
  // two launches (MODE 1 / 2, see kern_grad_sym_kernel): the second one's partials follow the first one's
  double* p2 = partial + (size_t)grid.x * grid.y * NP_ARDSYM;
  const dim3 grid2(grid.x, (unsigned)kg_mode2_ny(g.N));
  const int per2 = KG_MODE2_PER;
  // workgroups per CU the kernel is compiled for.  At two per CU (256 registers a wave) the variant keeps no covGrad values for
  // the next half in registers (the other workgroup's waves cover that latency) and still spills a little; that wins up to
  // D = 16 (N = 65 536: D = 4 6.3 ms, D = 8 6.9 ms = 2.5 TB/s of the 4 N^2 bytes, D = 16 8.6 ms, against 18.2 / 8.4 / 8.9 ms at
  // one per CU) and loses at D = 32, where the second group of per-dimension accumulators pushes the spills past 150 registers
  // (16.9 ms against 11.5).  GPC_KGRAD_ARD_OCC = 1 / 2 forces one or the other.
  static int occ_env = -1;
  if(occ_env < 0) {
    const char* e = getenv("GPC_KGRAD_ARD_OCC");
    occ_env = e ? atoi(e) : 0;
  }
  const int occ8 = occ_env ? occ_env : (g.D <= 16 ? 2 : 1);
  // D > 8: eight waves of 32 x 32 patches, one workgroup per CU = two waves per SIMD without spills (246 registers; the four-wave
  // form needs 452 at D = 32, i.e. one wave per SIMD).  N = 65 536: D = 32 11.7 -> 10.2 ms, D = 16 8.7 -> 8.0 ms; at D <= 8 the
  // four-wave form at two workgroups per CU stays ahead (6.8 against 7.2 ms).  PMC (profiles/r03_pmc_kgrad_ard.txt, D = 32): the
  // matrix pipe is busy 36 % -> 43 % of the time, 1.9e9 vector instructions beside 5.4e8 MFMA ops: the exponentials and the
  // weights are as much work as the two products.  GPC_KGRAD_ARD_NW=4 / 8 forces a form.
  static int nw_env = -1;
  if(nw_env < 0) {
    const char* e = getenv("GPC_KGRAD_ARD_NW");
    nw_env = e ? atoi(e) : 0;
  }
  if((g.D > 8 && nw_env != 4) || nw_env == 8 || nw_env == 82) {
    const int occ = (nw_env == 82) ? 2 : 1;     // (8 / 82: eight waves at every D, one / two workgroups per CU; A/B runs)
#define GPC_ARD_LAUNCH8(NKV)                                                                                                    \
  do {                                                                                                                            \
    if(g.all_general) hipLaunchKernelGGL((kern_grad_ard_sym_kernel<NKV, ND, 1, 8, 2>), grid, dim3(512), 0, s, ks, g, XT, per, partial);  \
    else {                                                                                                                            \
    if(occ == 1) hipLaunchKernelGGL((kern_grad_ard_sym_kernel<NKV, ND, 1, 8, 1>), grid, dim3(512), 0, s, ks, g, XT, per, partial);   \
    else hipLaunchKernelGGL((kern_grad_ard_sym_kernel<NKV, ND, 2, 8, 1>), grid, dim3(512), 0, s, ks, g, XT, per, partial);           \
    hipLaunchKernelGGL((kern_grad_ard_sym_kernel<NKV, ND, 1, 8, 2>), grid2, dim3(512), 0, s, ks, g, XT, per2, p2);                    \
    }                                                                                                                                 \
  } while(0)
    if(g.D <= 4) GPC_ARD_LAUNCH8(1);
    else if(g.D <= 8) GPC_ARD_LAUNCH8(2);
    else if(g.D <= 16) GPC_ARD_LAUNCH8(4);
    else GPC_ARD_LAUNCH8(8);
#undef GPC_ARD_LAUNCH8
    GPC_HIP_CHECK(hipGetLastError());
    return GPC_OK;
  }
#define GPC_ARD_LAUNCH(NKV)                                                                                              \
  do {                                                                                                                     \
    if(g.all_general) {                                                                                                    \
      hipLaunchKernelGGL((kern_grad_ard_sym_kernel<NKV, ND, 2, 4, 2>), grid, dim3(256), 0, s, ks, g, XT, per, partial);    \
    } else {                                                                                                               \
    if(occ8 == 1)                                                                                                          \
      hipLaunchKernelGGL((kern_grad_ard_sym_kernel<NKV, ND, 1, 4, 1>), grid, dim3(256), 0, s, ks, g, XT, per, partial);    \
    else                                                                                                                   \
      hipLaunchKernelGGL((kern_grad_ard_sym_kernel<NKV, ND, 2, 4, 1>), grid, dim3(256), 0, s, ks, g, XT, per, partial);    \
    hipLaunchKernelGGL((kern_grad_ard_sym_kernel<NKV, ND, 1, 8, 2>), grid2, dim3(512), 0, s, ks, g, XT, per2, p2);          \
    }                                                                                                                      \
  } while(0)
  if(g.D <= 4) GPC_ARD_LAUNCH(1);
  else if(g.D <= 8) GPC_ARD_LAUNCH(2);
  else if(g.D <= 16) GPC_ARD_LAUNCH(4);
  else GPC_ARD_LAUNCH(8);
#undef GPC_ARD_LAUNCH
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

// per-dimension ARD sums: S_k = sum_{i != j} cg(i,j) k~(i,j) (x_ik - x_jk)^2 for k in [dim0, dim0 + 32)
__global__ void __launch_bounds__(256) ard_dim_grad_kernel(const KSpecDev ks, const GradArgs g, int64_t dim0,
                                                           double* __restrict__ partial)
{
  __shared__ __attribute__((aligned(16))) double Xi[DC * TI];
  __shared__ __attribute__((aligned(16))) double Xj[DC * TJ];
  __shared__ __attribute__((aligned(16))) double Ai[DC * TI];
  __shared__ __attribute__((aligned(16))) double Aj[DC * TJ];
  __shared__ double sh[4];
  const int t = threadIdx.x;
  const int il = 2 * (t & 63), jl = 8 * (t >> 6);
  double acc[ARD_PASS];
#pragma unroll
  for(int p = 0; p < ARD_PASS; p++) acc[p] = 0.0;

  const int64_t total = g.tiles_i * g.tiles_j;
  for(int64_t tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int64_t i0 = (tile % g.tiles_i) * TI, j0 = (tile / g.tiles_i) * TJ;
    double dot[2][8], ard[2][8];
    tile_products<false, true>(ks, g, i0, j0, Xi, Xj, Ai, Aj, dot, ard);
    const int64_t gi = i0 + il;
    double W[2][8];
#pragma unroll
    for(int b = 0; b < 8; b++) {
      const int64_t gj = j0 + jl + b;
#pragma unroll
      for(int a = 0; a < 2; a++) {
        double w = 0.0;
        if(gj < g.N && gi + a < g.N && gi + a != gj)
          w = g.cg[(gi + a) + gj * g.ldc] * exp(-(ard[a][b] * ks.ard_hiw[0]));
        W[a][b] = w;
      }
    }
#pragma unroll
    for(int c = 0; c < ARD_PASS / DC; c++) {
      const int64_t d0 = dim0 + c * DC;
      if(d0 >= g.D) break;
      const int dc = (int)((g.D - d0 < DC) ? (g.D - d0) : DC);
      __syncthreads();
      for(int idx = t; idx < DC * TI; idx += 256) {
        const int d = idx / TI, i = idx % TI;
        Xi[idx] = (d < dc && i0 + i < g.N) ? g.X[(i0 + i) + (d0 + d) * g.ldx] : 0.0;
      }
      for(int idx = t; idx < DC * TJ; idx += 256) {
        const int d = idx / TJ, j = idx % TJ;
        Xj[idx] = (d < dc && j0 + j < g.N) ? g.X[(j0 + j) + (d0 + d) * g.ldx] : 0.0;
      }
      __syncthreads();
#pragma unroll
      for(int d = 0; d < DC; d++) {
        if(d < dc) {
          const double2_t xi = *reinterpret_cast<const double2_t*>(&Xi[d * TI + il]);
          double s = 0.0;
#pragma unroll
          for(int b = 0; b < 8; b++) {
            const double xj = Xj[d * TJ + jl + b];
            const double e0 = xi.x - xj, e1 = xi.y - xj;
            s = fma(W[0][b] * e0, e0, s);
            s = fma(W[1][b] * e1, e1, s);
          }
          acc[c * DC + d] += s;
        }
      }
    }
  }
#pragma unroll
  for(int p = 0; p < ARD_PASS; p++) {
    const double r = block_sum(acc[p], sh);
    if(t == 0) partial[(int64_t)blockIdx.x * ARD_PASS + p] = r;
  }
}

__global__ void __launch_bounds__(256) row_norms_kernel2(const double* __restrict__ X, int64_t ldx, int64_t N,
                                                         int64_t D, double* __restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if(i >= N) return;
  double acc = 0.0;
  for(int64_t d = 0; d < D; d++) {
    const double x = X[i + d * ldx];
    acc = fma(x, x, acc);
  }
  out[i] = acc;
}

int fetch_partials(const double* d_p, int64_t nblk, int np, double* sums, hipStream_t s)
{
  const size_t n = (size_t)nblk * np;
  double* h = (double*)malloc(sizeof(double) * n);
  if(!h) return GPC_ENOMEM;
  HostFetch f;
  int rc = f.add(h, d_p, sizeof(double) * n, s);
  if(rc == GPC_OK) rc = f.finish(s);
  if(rc != GPC_OK) {
    free(h);
    return rc;
  }
  for(int p = 0; p < np; p++) {
    double a = 0.0;
    for(int64_t b = 0; b < nblk; b++) a += h[b * np + p];
    sums[p] = a;
  }
  free(h);
  return GPC_OK;
}

}  // namespace
}  // namespace gpc

using namespace gpc;

static int kern_grad_pass(const gpc_kspec* ksp, const double* X, int64_t N, int64_t D, int64_t ldx, const double* covGrad,
                          int64_t ldc, const double* A, int64_t lda, int nd, double* gout, hipStream_t s, bool* took_fused);

// The parameter gradient is a sum over covGrad that separates by term: dK/dtheta of a term's parameter involves that term
// alone (CCmpndKern::getGradParams, CKern.cpp:284-298, simply concatenates its components' results).  So a compound outside the
// domain of ONE of the fast symmetric kernels -- an rbfard term beside rbf / lin terms, more than two rbf terms, several
// rbfard terms -- is evaluated in several passes over (half of) covGrad, each on the fast kernel of its terms, instead of one
// pass of the scalar kernel over all of it (0.7-1.5 TB/s): the distance terms two rbf at a time together with every lin /
// bias / white term, each rbfard term alone (the first one takes bias / white along when there is no distance term).
static int kern_grad_impl(const gpc_kspec* ksp, const double* X, int64_t N, int64_t D, int64_t ldx, const double* covGrad,
                          int64_t ldc, const double* A, int64_t lda, int nd, double* gout, hipStream_t s, bool* took_fused)
{
  if(!ksp || ksp->n_terms < 0 || ksp->n_terms > GPC_MAX_TERMS) {
    set_error("kernel spec: bad term count");
    return GPC_EINVAL;
  }
  int n_rbf = 0, n_ard = 0, n_lin = 0;
  for(int t = 0; t < ksp->n_terms; t++) {
    n_rbf += ksp->types[t] == GPC_KERN_RBF;
    n_ard += ksp->types[t] == GPC_KERN_RBFARD;
    n_lin += ksp->types[t] == GPC_KERN_LIN;
  }
  const bool has_dot = n_rbf + n_lin > 0;
  static int split = -1;
  if(split < 0) {
    const char* e = getenv("GPC_KGRAD_SPLIT");     // 0: mixed compounds on the one-pass scalar kernel, as before (A/B runs)
    split = e ? atoi(e) : 1;
  }
  const bool one_pass = (n_ard == 0 && n_rbf <= 2) || (n_ard == 1 && !has_dot) || (!split && n_ard <= 1 && n_rbf <= 4);
  if(one_pass) return kern_grad_pass(ksp, X, N, D, ldx, covGrad, ldc, A, lda, nd, gout, s, took_fused);
  // chunk 0: all lin / bias / white terms with the first two rbf terms (or, without distance terms, with the first rbfard
  // term); then two rbf terms per chunk; then one rbfard term per chunk
  std::vector<std::vector<int>> chunks(1);
  int rbf_in0 = 0;
  bool ard_in0 = false;
  std::vector<int> later_rbf, later_ard;
  for(int t = 0; t < ksp->n_terms; t++) {
    const int ty = ksp->types[t];
    if(ty == GPC_KERN_RBF) {
      if(rbf_in0 < 2) { chunks[0].push_back(t); rbf_in0++; }
      else later_rbf.push_back(t);
    } else if(ty == GPC_KERN_RBFARD) {
      if(!has_dot && !ard_in0) { chunks[0].push_back(t); ard_in0 = true; }
      else later_ard.push_back(t);
    } else {
      chunks[0].push_back(t);
    }
  }
  for(size_t i = 0; i < later_rbf.size(); i += 2) {
    chunks.push_back(std::vector<int>(later_rbf.begin() + (long)i, later_rbf.begin() + (long)(i + 2 < later_rbf.size() ? i + 2 : later_rbf.size())));
  }
  for(int t : later_ard) chunks.push_back(std::vector<int>(1, t));
  const int nparams = ksp->offs[ksp->n_terms];
  for(int p = 0; p < nparams; p++) gout[p] = 0.0;
  bool all_fused = true;
  for(const std::vector<int>& ch : chunks) {
    if(ch.empty()) continue;
    gpc_kspec sub;
    memset(&sub, 0, sizeof(sub));
    for(int t : ch) {
      const int off = ksp->offs[t], np = ksp->offs[t + 1] - off;
      if(off < 0 || np < 0 || off + np > GPC_MAX_PARAMS) {
        set_error("kernel spec: bad parameter offsets");
        return GPC_EINVAL;
      }
      const int o = sub.offs[sub.n_terms];
      sub.types[sub.n_terms] = ksp->types[t];
      for(int q = 0; q < np; q++) sub.params[o + q] = ksp->params[off + q];
      sub.offs[sub.n_terms + 1] = o + np;
      sub.n_terms++;
    }
    double gsub[GPC_MAX_PARAMS];
    bool took = false;
    GPC_CHECK(kern_grad_pass(&sub, X, N, D, ldx, covGrad, ldc, A, lda, nd, gsub, s, &took));
    all_fused = all_fused && took;
    if(nd > 0 && !took) break;     // a fused request one of the passes cannot take: the caller materialises covGrad
    for(int i = 0; i < sub.n_terms; i++) {
      const int t = ch[(size_t)i];
      for(int q = 0; q < sub.offs[i + 1] - sub.offs[i]; q++) gout[ksp->offs[t] + q] = gsub[sub.offs[i] + q];
    }
  }
  if(took_fused) *took_fused = all_fused;
  return GPC_OK;
}

extern "C" int gpc_kern_grad_f64(const gpc_kspec* ksp, const double* X, int64_t N, int64_t D, int64_t ldx,
                                 const double* covGrad, int64_t ldc, double* gout, void* stream)
{
  GPC_CHECK(ensure_device());
  GPC_REQUIRE(ksp && gout && N >= 0 && D >= 0 && ldx >= (N > 1 ? N : 1) && ldc >= (N > 1 ? N : 1), "kern_grad args");
  return kern_grad_impl(ksp, X, N, D, ldx, covGrad, ldc, nullptr, 0, 0, gout, as_stream(stream), nullptr);
}

// CGp::updateG without the covGrad matrix (CGp.cpp:666-679 + 1096-1117 in one pass): the kernel reads invK and the
// N x d matrix A = invK * m and forms covGrad(i,j) = -0.5 (d invK(i,j) - sum_o A(i,o) A(j,o)) in registers.  Taken when the
// kernel is on one of the symmetric MFMA walks (no rbfard term and at most two rbf terms, or exactly one rbfard term with
// bias / white only), D <= 32 and d <= 2; otherwise GPC_EUNSUPPORTED (the caller materialises covGrad as before).
extern "C" int gpc_kern_grad_fused_f64(const gpc_kspec* ksp, const double* X, int64_t N, int64_t D, int64_t ldx,
                                       const double* invK, int64_t ldi, const double* A, int64_t lda, int64_t d, double* gout,
                                       void* stream)
{
  GPC_CHECK(ensure_device());
  GPC_REQUIRE(ksp && gout && invK && A && N >= 0 && D >= 0 && d >= 1 && ldx >= (N > 1 ? N : 1) && ldi >= (N > 1 ? N : 1) &&
                  lda >= (N > 1 ? N : 1),
              "kern_grad_fused args");
  if(d > 2) return GPC_EUNSUPPORTED;
  bool took = false;
  GPC_CHECK(kern_grad_impl(ksp, X, N, D, ldx, invK, ldi, A, lda, (int)d, gout, as_stream(stream), &took));
  return took ? GPC_OK : GPC_EUNSUPPORTED;
}

static int kern_grad_pass(const gpc_kspec* ksp, const double* X, int64_t N, int64_t D, int64_t ldx, const double* covGrad,
                          int64_t ldc, const double* A, int64_t lda, int nd, double* gout, hipStream_t s, bool* took_fused)
{
  KSpecDev ks;
  GPC_CHECK(collapse_kspec(ksp, D, &ks));
  const int nparams = ksp->offs[ksp->n_terms];
  for(int p = 0; p < nparams; p++) gout[p] = 0.0;
  if(N == 0) return GPC_OK;

  GradArgs g;
  g.X = X;
  g.cg = covGrad;
  g.ldx = ldx;
  g.ldc = ldc;
  g.N = N;
  g.D = D;
  g.tiles_i = (N + TI - 1) / TI;
  g.tiles_j = (N + TJ - 1) / TJ;
  g.A = A;
  g.lda = lda;
  g.nd = nd;
  const int64_t total = g.tiles_i * g.tiles_j;
  const int64_t nblk = total < 2048 ? total : 2048;

  void* ws = nullptr;
  // column tiles per workgroup: enough workgroups to fill the chip on a small matrix, long walks (operand fragments and
  // the final reductions amortised) on a large one
  const int64_t sym_nrb = (N + GMI - 1) / GMI;
  int64_t sym_per = sym_nrb * (sym_nrb + 1) / 768;
  // (N = 1000, the GP-LVM's size: 16 row blocks; at four tiles per workgroup 72 workgroups had work and the rbfard pass took
  //  52 us of a 670 us evaluation; one tile each fills the chip)
  const int64_t per_min = (N < 2048) ? 1 : 4;
  if(sym_per < per_min) sym_per = per_min;
  if(sym_per > 48) sym_per = 48;
  const int64_t sym_ny = (2 * sym_nrb + sym_per - 1) / sym_per;
  g.all_general = (N < 4096) ? 1 : 0;
  const int64_t sym_nwg = g.all_general ? sym_nrb * sym_ny : sym_nrb * sym_ny + sym_nrb * kg_mode2_ny(N);   // the fast tiles' workgroups + the second launch's (diagonal blocks, ragged row block)
  size_t pbytes = sizeof(double) * (size_t)nblk * (NP_MAIN > ARD_PASS ? NP_MAIN : ARD_PASS);
  if(pbytes < sizeof(double) * (size_t)sym_nwg * NP_MAIN) pbytes = sizeof(double) * (size_t)sym_nwg * NP_MAIN;
  GPC_CHECK(workspace(WS_KERN, sizeof(double) * (size_t)N + pbytes, &ws));
  double* nrm = static_cast<double*>(ws);
  double* partial = nrm + N;
  g.n1 = nrm;
  if(ks.need_dot) {
    hipLaunchKernelGGL(row_norms_kernel2, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, X, ldx, N, D, nrm);
  }
  const bool dot = ks.need_dot != 0, ard = ks.n_ard > 0;
  static int use_sym = -1;
  if(use_sym < 0) {
    const char* e = getenv("GPC_KGRAD_SYM");
    use_sym = e ? atoi(e) : 1;
  }
  if(use_sym && !ard && D >= 1 && D <= GMDC && ks.n_rbf <= 2) {
    const int64_t nrb = sym_nrb, per = sym_per, ny = sym_ny, nwg = sym_nwg;
    const dim3 grid((unsigned)nrb, (unsigned)ny);
    if(ks.n_rbf == 0) GPC_CHECK(launch_grad_sym<0>(ks, g, (int)per, grid, partial, s));
    else if(ks.n_rbf == 1) GPC_CHECK(launch_grad_sym<1>(ks, g, (int)per, grid, partial, s));
    else GPC_CHECK(launch_grad_sym<2>(ks, g, (int)per, grid, partial, s));
    double S2[NP_MAIN];
    GPC_CHECK(fetch_partials(partial, nwg, NP_MAIN, S2, s));
    if(took_fused) *took_fused = true;
    int ir = 0;
    for(int t = 0; t < ksp->n_terms; t++) {
      double* gt = gout + ksp->offs[t];
      const double* p = ksp->params + ksp->offs[t];
      switch(ksp->types[t]) {
      case GPC_KERN_RBF:
        gt[0] = -0.5 * p[1] * S2[2 * ir];
        gt[1] = S2[13] + S2[2 * ir + 1];
        ir++;
        break;
      case GPC_KERN_WHITE: gt[0] = S2[13]; break;
      case GPC_KERN_BIAS: gt[0] = S2[12]; break;
      case GPC_KERN_LIN: gt[0] = S2[11]; break;
      default: break;
      }
    }
    return GPC_OK;
  }
  // one rbfard term (+ white / bias), D <= 32: the symmetric walk with the per-dimension sums as MFMA products
  static int use_ard_sym = -1;
  if(use_ard_sym < 0) {
    const char* e = getenv("GPC_KGRAD_ARD_SYM");
    use_ard_sym = e ? atoi(e) : 1;
  }
  bool scales_ok = ard && ks.n_ard == 1;
  for(int64_t q = 0; scales_ok && q < D; q++) scales_ok = ks.ard_scale[0][q] > 1e-150;   // (S_q is divided by s_q below)
  if(use_sym && use_ard_sym && scales_ok && !dot && D >= 1 && D <= GMDC && nd <= 2) {
    const int dp = (D > 16) ? 32 : 16;
    const int64_t nrb = sym_nrb, per = sym_per, ny = sym_ny, nwg = sym_nwg;
    void* wx = nullptr;
    GPC_CHECK(workspace(WS_XSCALED, sizeof(double) * ((size_t)N * (size_t)(D + dp) + (size_t)nwg * NP_ARDSYM + 64), &wx));
    double* Xs = static_cast<double*>(wx);
    double* XT = Xs + (size_t)N * D;
    double* mean = XT + (size_t)N * dp;
    double* part2 = mean + 64;
    hipLaunchKernelGGL(ard_mean_kernel, dim3((unsigned)D), dim3(256), 0, s, X, ldx, N, mean);
    hipLaunchKernelGGL(ard_prep_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, ks, X, ldx, N, (int)D, mean, Xs, XT, dp,
                       nrm);
    GPC_HIP_CHECK(hipGetLastError());
    GradArgs ga = g;
    ga.X = Xs;
    ga.ldx = N;
    const dim3 grid((unsigned)nrb, (unsigned)ny);
    if(nd == 0) GPC_CHECK(launch_grad_ard_sym<0>(ks, ga, XT, (int)per, grid, part2, s));
    else if(nd == 1) GPC_CHECK(launch_grad_ard_sym<1>(ks, ga, XT, (int)per, grid, part2, s));
    else GPC_CHECK(launch_grad_ard_sym<2>(ks, ga, XT, (int)per, grid, part2, s));
    double S3[NP_ARDSYM];
    GPC_CHECK(fetch_partials(part2, nwg, NP_ARDSYM, S3, s));
    if(took_fused) *took_fused = true;
    for(int t = 0; t < ksp->n_terms; t++) {
      double* gt = gout + ksp->offs[t];
      const double* p = ksp->params + ksp->offs[t];
      switch(ksp->types[t]) {
      case GPC_KERN_RBFARD:
        gt[0] = -0.5 * p[1] * S3[8];
        gt[1] = S3[13] + S3[9];
        for(int64_t q = 0; q < D; q++) gt[2 + q] = -0.5 * p[0] * p[1] * (S3[NP_MAIN + q] / ks.ard_scale[0][q]);
        break;
      case GPC_KERN_WHITE: gt[0] = S3[13]; break;
      case GPC_KERN_BIAS: gt[0] = S3[12]; break;
      default: break;
      }
    }
    return GPC_OK;
  }
  if(nd > 0) return GPC_OK;   // fused request outside the symmetric kernels' domain: nothing launched, *took_fused stays false
  if(dot && ard)
    hipLaunchKernelGGL((kern_grad_kernel<true, true>), dim3((unsigned)nblk), dim3(256), 0, s, ks, g, partial);
  else if(dot)
    hipLaunchKernelGGL((kern_grad_kernel<true, false>), dim3((unsigned)nblk), dim3(256), 0, s, ks, g, partial);
  else if(ard)
    hipLaunchKernelGGL((kern_grad_kernel<false, true>), dim3((unsigned)nblk), dim3(256), 0, s, ks, g, partial);
  else
    hipLaunchKernelGGL((kern_grad_kernel<false, false>), dim3((unsigned)nblk), dim3(256), 0, s, ks, g, partial);
  GPC_HIP_CHECK(hipGetLastError());
  double S[NP_MAIN];
  GPC_CHECK(fetch_partials(partial, nblk, NP_MAIN, S, s));

  double Sdim[GPC_MAX_ARD_DIM];
  if(ard) {
    for(int64_t dim0 = 0; dim0 < D; dim0 += ARD_PASS) {
      hipLaunchKernelGGL(ard_dim_grad_kernel, dim3((unsigned)nblk), dim3(256), 0, s, ks, g, dim0, partial);
      GPC_HIP_CHECK(hipGetLastError());
      double tmp[ARD_PASS];
      GPC_CHECK(fetch_partials(partial, nblk, ARD_PASS, tmp, s));
      for(int q = 0; q < ARD_PASS && dim0 + q < D; q++) Sdim[dim0 + q] = tmp[q];
    }
  }

  // assemble in spec order
  int irbf = 0;
  for(int t = 0; t < ksp->n_terms; t++) {
    double* gt = gout + ksp->offs[t];
    const double* p = ksp->params + ksp->offs[t];
    switch(ksp->types[t]) {
    case GPC_KERN_RBF:
      gt[0] = -0.5 * p[1] * S[2 * irbf];
      gt[1] = S[13] + S[2 * irbf + 1];
      irbf++;
      break;
    case GPC_KERN_RBFARD:
      gt[0] = -0.5 * p[1] * S[8];
      gt[1] = S[13] + S[9];
      for(int64_t q = 0; q < D; q++) gt[2 + q] = -0.5 * p[0] * p[1] * Sdim[q];
      break;
    case GPC_KERN_WHITE: gt[0] = S[13]; break;
    case GPC_KERN_BIAS: gt[0] = S[12]; break;
    case GPC_KERN_LIN: gt[0] = S[11]; break;
    default: break;
    }
  }
  return GPC_OK;
}
