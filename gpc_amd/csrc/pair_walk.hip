// pair_walk.hip -- one MFMA walk over ALL pairs (i, n) of two point sets for the passes that read a FULL N x N2 weight matrix
// once (HBM-read bound: 8 N N2 bytes):
//
//   gpc_kern_gradx_f64 / gpc_kern_gradx_cross_f64   gX(i,q) = pf sum_n G(i,n) dk(x_i, x2_n)/dx_iq
//       CCmpndKern::getGradX + the dotColCol / dotColRow loops of the callers (CKern.cpp:184-193, rbf 1115-1135, rbfard
//       3268-3293, lin 2291-2308; CGplvm.cpp:573-604; CGp.cpp:1163-1176)
//   gpc_kern_grad_cross_f64                          g_p = sum_{i,n} G(i,n) dk(x_i, x2_n)/dtheta_p
//       CCmpndKern::getGradParams(g, X, X2, covGrad) (rbf CKern.cpp:1175-1202, rbfard 3318-3357, bias 1015-1019, lin 2354-2368)
//
// Until round 4 these ran on scalar kernels (gplvm.hip: a lane per row, one column at a time, x_n through scalar loads:
// 0.65-0.86 TB/s of G at N = 32 768).  Here they take the structure of the symmetric parameter-gradient kernels
// (kern_grad.hip) without the symmetry: a workgroup owns 128 rows, walks 64-column tiles of its slice of the columns, x_i.x_n
// comes out of v_mfma_f64_16x16x4, the exponentials run on the accumulator tile, and everything that is "weights times
// coordinates" is a second MFMA product whose weight operand is the accumulator tile itself (register r of a 16 x 16 tile IS
// the operand layout of a k-step over the columns 4 r .. 4 r + 3):
//       rho_i = sum_n W(i,n),     Y(q,i) = sum_n W(i,n) x_nq,     W = G o (sum_t c_t exp(-h_t d2))
//   dL/dX:       distance terms  dk/dx_iq = -2 h k (x_iq - x_nq)   =>   gX(i,q) = pf s_q^1/2 (Y(q,i) - x_iq rho_i)   (c_t = 2 h_t var_t)
//                linear term     dk/dx_iq = var x_nq                =>   joins Y with weight var G (no rho term)
//   parameters:  sum G k~, sum G k~ d2, sum G, sum G x.x2 as scalars; the rbfard per-dimension sums
//                S_q = sum W (x_iq - x_nq)^2 = sum_i rho_i x_iq^2 + sum_n kappa_n x_nq^2 - 2 sum_i x_iq Y(q,i)
// Differences of large sums: the passes that form Y - x rho or S_q run on inputs CENTRED by the column set's mean (differences
// do not change) and, for an rbfard term, scaled by sqrt(s_q) so that x.x' from the same MFMA walk is the ARD distance.
// Column slices write partial results that are added in a fixed order (deterministic, like everything else here).
// D <= 32; wider inputs, and problems too small to fill the chip, stay on gplvm.hip's kernels.
#include "gpc_common.hpp"
#include <vector>
#include <stdlib.h>
#include <string.h>

namespace gpc {

namespace {

constexpr int PW_MI = 128, PW_MJ = 64, PW_SJ = 80, PW_DC = 32, PW_SI = 144;
constexpr int PW_NP = 8;               // scalar partials per workgroup: d2e[0], e[0], d2e[1], e[1], all, lin, 2 spare
constexpr int PW_NPQ = PW_NP + 32;     // ... followed by 32 S_q (SQ instances)
typedef double pdouble4 __attribute__((ext_vector_type(4)));

struct WalkArgs {
  const double* XA;    // rows: N x D column-major (as the pass wants them: raw, or centred / scaled)
  const double* nA;    // |xa_i|^2
  const double* XB;    // columns: N2 x D column-major
  const double* nB;
  const double* XTB;   // columns again, row-major N2 x DP (zero-padded): the operand of the Y product (null when no Y)
  const double* G;     // N x N2
  int64_t ldxa, ldxb, ldg, N, N2;
  int D, jt_per_block;
  double hiw[2], coef[2];   // W = G sum_t coef[t] exp(-hiw[t] d2)
  double lin;               // ROWS: G * lin joins the Y weights (its row sums in rho_l); parameter passes: unused
  double pf;                // ROWS: factor of the result (2: symmetric pass, CGplvm.cpp:577; 1: cross Gram)
  double* part;             // ROWS: [slice][D][N]
  double* partial;          // parameter passes: [workgroup][PW_NP or PW_NPQ]
  double scale[32];         // ROWS: sqrt(s_q) of an rbfard pass (1 otherwise)
  double mlin[32];          // ROWS: the mean the inputs were centred by (the linear term multiplies the ACTUAL x_n)
};

template <int NW>
__device__ __forceinline__ double pw_block_sum(double v, double* sh)
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

// NK: k-steps of the dot product (D <= 4 NK).  NW waves: 4 (64 x 32 patches) or 8 (32 x 32).  OCC: workgroups per CU the
// kernel is compiled for (2: no G values held for the next half-tile).  NEXP: exponentials per pair.
// ROWS: per-row output (dL/dX).  SQ: the rbfard per-dimension parameter sums.  Neither: scalar parameter sums only.
template <int NK, int NW, int OCC, int NEXP, bool ROWS, bool SQ>
__global__ void __launch_bounds__(64 * NW, OCC) pair_walk_kernel(const WalkArgs g)
{
  constexpr bool WANT_Y = ROWS || SQ;
  constexpr int QX = (NK > 4) ? 2 : 1;          // 16-wide groups of input dimensions
  constexpr int DP = 16 * QX;
  __shared__ double Xj[2][PW_DC * PW_SJ];
  __shared__ double Nj[2][PW_MJ];
  __shared__ double Xi[NK > 2 ? PW_DC * PW_SI : 1];
  constexpr int NT = 64 * NW;           // threads
  constexpr int RW = 256 / NW;          // rows of a wave's patch: 64 or 32
  constexpr int TM = RW / 16;           // its 16-row MFMA tiles: 4 or 2
  static_assert(NW == 4 || NW == 8, "waves per workgroup");
  static_assert(!(ROWS && SQ), "one kind of output per instance");
  __shared__ double sh[NW];
  __shared__ double red[NW][32];
  const int t = threadIdx.x;
  const int lane = t & 63, w = t >> 6;
  const int wm = w & (NW / 2 - 1), wn = w / (NW / 2);
  const int64_t i0 = (int64_t)blockIdx.x * PW_MI;
  const int64_t tiles_j = (g.N2 + PW_MJ - 1) / PW_MJ;
  const int64_t jt0 = (int64_t)blockIdx.y * g.jt_per_block;
  int64_t jt1 = jt0 + g.jt_per_block;
  if(jt1 > tiles_j) jt1 = tiles_j;        // (the host sizes the grid so that jt0 < jt1 for every slice)
  const int dc = g.D;   // <= 4 NK

  constexpr bool AF_LDS = (NK > 2);
  double af[AF_LDS ? 1 : NK][TM];
  if(AF_LDS) {
#pragma unroll
    for(int u = 0; u < (PW_DC * PW_MI) / NT; u++) {
      const int idx = t + NT * u;
      const int kr = idx >> 7, row = idx & 127;
      int64_t gi = i0 + row;
      if(gi > g.N - 1) gi = g.N - 1;
      Xi[kr * PW_SI + row] = (kr < dc) ? g.XA[gi + (int64_t)kr * g.ldxa] : 0.0;
    }
  } else {
#pragma unroll
    for(int kk = 0; kk < (AF_LDS ? 1 : NK); kk++)
#pragma unroll
      for(int tm = 0; tm < TM; tm++) {
        int64_t gi = i0 + wm * RW + tm * 16 + (lane & 15);
        if(gi > g.N - 1) gi = g.N - 1;
        int kr = kk * 4 + (lane >> 4);
        if(kr > dc - 1) kr = dc - 1;
        af[kk][tm] = (kk * 4 + (lane >> 4) < dc) ? g.XA[gi + (int64_t)kr * g.ldxa] : 0.0;
      }
  }
  double ni[TM];
#pragma unroll
  for(int tm = 0; tm < TM; tm++) {
    int64_t gi = i0 + wm * RW + tm * 16 + (lane & 15);
    if(gi > g.N - 1) gi = g.N - 1;
    ni[tm] = g.nA[gi];
  }

  const int ws = __builtin_amdgcn_readfirstlane(w);
  constexpr int VJ = (PW_DC * PW_MJ) / NT;   // staged values of the column tile per thread: 8 or 4
  double vj[VJ], vn;
  auto prefetch = [&](int64_t jt) {
    int64_t gj = jt * PW_MJ + lane;
    if(gj > g.N2 - 1) gj = g.N2 - 1;
#pragma unroll
    for(int u = 0; u < VJ; u++) {
      const int d = ws + NW * u;
      vj[u] = 0.0;
      if(d < dc) vj[u] = (g.XB + (int64_t)d * g.ldxb)[gj];
    }
    vn = g.nB[gj];
  };
  prefetch(jt0);

  double s_d2e[NEXP], s_e[NEXP], s_all = 0.0, s_lin = 0.0;
#pragma unroll
  for(int q = 0; q < NEXP; q++) s_d2e[q] = s_e[q] = 0.0;
  double rho[TM], rhol[TM];
#pragma unroll
  for(int tm = 0; tm < TM; tm++) rho[tm] = rhol[tm] = 0.0;
  pdouble4 Y1[TM][QX];
  double Bq[QX];             // SQ: sum_n kappa_n x_nq^2 for q = (lane & 15) + 16 qx, over this lane's columns n = 4 r + (lane >> 4)
#pragma unroll
  for(int qx = 0; qx < QX; qx++) Bq[qx] = 0.0;
#pragma unroll
  for(int tm = 0; tm < TM; tm++)
#pragma unroll
    for(int qx = 0; qx < QX; qx++) Y1[tm][qx] = (pdouble4){0.0, 0.0, 0.0, 0.0};
  const double lin = g.lin;
  const bool has_lin = ROWS && (lin != 0.0);

  for(int64_t jt = jt0; jt < jt1; jt++) {
    double* Xjb = Xj[(jt - jt0) & 1];
    double* Njb = Nj[(jt - jt0) & 1];
    const int64_t j0 = jt * PW_MJ;
#pragma unroll
    for(int u = 0; u < VJ; u++) {
      const int idx = t + NT * u;
      Xjb[(idx >> 6) * PW_SJ + (idx & 63)] = vj[u];
    }
    if(t < PW_MJ) Njb[t] = vn;
    __syncthreads();
    if(jt + 1 < jt1) prefetch(jt + 1);

    const bool full = (i0 + PW_MI <= g.N) && (j0 + PW_MJ <= g.N2);
    constexpr bool LEAN = (OCC == 2);   // two workgroups per CU: no G values held for the next half (the other workgroup's waves cover the latency)
    double c[LEAN ? 1 : 2][4][TM];
    auto load_g = [&](int tn) {
#pragma unroll
      for(int r = 0; r < 4; r++) {
        const int64_t gj = j0 + wn * 32 + tn * 16 + (lane >> 4) + 4 * r;
        const int64_t gjc = (gj < g.N2) ? gj : (g.N2 - 1);
#pragma unroll
        for(int tm = 0; tm < TM; tm++) {
          const int64_t gi = i0 + wm * RW + tm * 16 + (lane & 15);
          const int64_t gic = (gi < g.N) ? gi : (g.N - 1);
          const double v = g.G[gic + gjc * g.ldg];
          c[LEAN ? 0 : tn][r][tm] = (full || (gi < g.N && gj < g.N2)) ? v : 0.0;
        }
      }
    };
    if(!LEAN) load_g(0);
#pragma unroll
    for(int tn = 0; tn < 2; tn++) {
      if(LEAN) load_g(tn);
      // rows of X2^T for this half's 16 columns: r -> n = 4 r + (lane >> 4), QX groups of 16 dimensions; asked for now, used after
      // the dot products and the exponentials
      double xc[4][QX];
      if(WANT_Y) {
#pragma unroll
        for(int r = 0; r < 4; r++) {
          int64_t jj = j0 + wn * 32 + tn * 16 + 4 * r + (lane >> 4);
          if(jj > g.N2 - 1) jj = g.N2 - 1;
#pragma unroll
          for(int qx = 0; qx < QX; qx++) xc[r][qx] = g.XTB[jj * DP + qx * 16 + (lane & 15)];
        }
      }
      pdouble4 acc[TM];
#pragma unroll
      for(int a = 0; a < TM; a++) acc[a] = (pdouble4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for(int kk = 0; kk < NK; kk++) {
        const int kr = kk * 4 + (lane >> 4);
        const double b = Xjb[kr * PW_SJ + wn * 32 + tn * 16 + (lane & 15)];
#pragma unroll
        for(int tm = 0; tm < TM; tm++) {
          const double a = AF_LDS ? Xi[kr * PW_SI + wm * RW + tm * 16 + (lane & 15)] : af[AF_LDS ? 0 : kk][tm];
          acc[tm] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc[tm], 0, 0, 0);
        }
      }
      if(!LEAN && tn == 0) load_g(1);
      // the weights replace the dot products in acc, register for register
#pragma unroll
      for(int r = 0; r < 4; r++) {
        const int jl = wn * 32 + tn * 16 + (lane >> 4) + 4 * r;
        const double nj = Njb[jl];
#pragma unroll
        for(int th = 0; th < TM; th += 2) {
#pragma unroll
          for(int u = 0; u < 2; u++) {
            const int tm = th + u;
            const double cw = c[LEAN ? 0 : tn][r][tm];
            const double dot = acc[tm][r];
            const double d2 = fma(-2.0, dot, ni[tm] + nj);
            double wd = 0.0;
#pragma unroll
            for(int q = 0; q < NEXP; q++) {
              const double e = cw * exp(-(g.hiw[q] * d2));   // (the table form of gpc_exp.hpp is slower here: D = 8 3.2 -> 4.9 ms; two chains per wave leave its LDS read exposed)
              if(!ROWS) {
                s_d2e[q] = fma(d2, e, s_d2e[q]);
                s_e[q] += e;
              }
              wd = fma(g.coef[q], e, wd);
            }
            if(!ROWS) {
              s_all += cw;
              s_lin = fma(cw, dot, s_lin);
            }
            rho[tm] += wd;
            if(has_lin) {
              const double wl = cw * lin;
              rhol[tm] += wl;
              wd += wl;
            }
            acc[tm][r] = wd;
          }
        }
      }
      if(WANT_Y) {
        // Y(q, i) += sum_n x_nq W(i, n): k-step r covers n = 4 r .. 4 r + 3 of this half.  SQ: the x_nq^2 term needs only the
        // column sums kappa_n = sum_i W(i, n): a butterfly over the 16 lanes that share n, after which lane (n, q) holds both
#pragma unroll
        for(int r = 0; r < 4; r++) {
          double kap = 0.0;
          if(SQ) {
            kap = acc[0][r] + acc[1][r];
            if(TM == 4) kap += acc[TM - 2][r] + acc[TM - 1][r];
            kap += __shfl_xor(kap, 1, 64);
            kap += __shfl_xor(kap, 2, 64);
            kap += __shfl_xor(kap, 4, 64);
            kap += __shfl_xor(kap, 8, 64);
          }
#pragma unroll
          for(int qx = 0; qx < QX; qx++) {
            const double x1 = xc[r][qx];
            if(SQ) Bq[qx] = fma(kap * x1, x1, Bq[qx]);
#pragma unroll
            for(int tm = 0; tm < TM; tm++) Y1[tm][qx] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, acc[tm][r], Y1[tm][qx], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- the walk is over ------------------------------------------------------------------------------------------------------
  // rho_i over the four column groups of the wave (lanes with the same lane & 15)
#pragma unroll
  for(int tm = 0; tm < TM; tm++) {
    rho[tm] += __shfl_xor(rho[tm], 16, 64);
    rho[tm] += __shfl_xor(rho[tm], 32, 64);
    if(ROWS) {
      rhol[tm] += __shfl_xor(rhol[tm], 16, 64);
      rhol[tm] += __shfl_xor(rhol[tm], 32, 64);
    }
  }
  if(ROWS) {
    // the waves with wn = 1 hand their share of Y and rho (the other 32 columns of every tile) to the wave with the same rows
    // and wn = 0 through LDS (the column-tile buffers are free now); that wave adds -- one fixed order -- and stores
    //   part[slice][q][i] = pf scale_q (Y(q,i) - x_iq rho_i + mean_q rho_l,i)
    constexpr int RS = DP + 2;
    double* R = &Xj[0][0];     // 2 * 32 * 80 doubles >= 128 * (32 + 2)
    static_assert(2 * PW_DC * PW_SJ >= PW_MI * (32 + 2), "row reduction buffer");
    __syncthreads();
    if(wn == 1) {
#pragma unroll
      for(int tm = 0; tm < TM; tm++) {
        const int row = wm * RW + tm * 16 + (lane & 15);
#pragma unroll
        for(int qx = 0; qx < QX; qx++)
#pragma unroll
          for(int r = 0; r < 4; r++) R[row * RS + (lane >> 4) + 4 * r + 16 * qx] = Y1[tm][qx][r];
        if((lane >> 4) == 0) {
          R[row * RS + DP] = rho[tm];
          R[row * RS + DP + 1] = rhol[tm];
        }
      }
    }
    __syncthreads();
    if(wn == 0) {
#pragma unroll
      for(int tm = 0; tm < TM; tm++) {
        const int row = wm * RW + tm * 16 + (lane & 15);
        const int64_t gi = i0 + row;
        const int64_t gic = (gi < g.N) ? gi : (g.N - 1);
        const double rd = rho[tm] + R[row * RS + DP], rl = rhol[tm] + R[row * RS + DP + 1];
#pragma unroll
        for(int qx = 0; qx < QX; qx++)
#pragma unroll
          for(int r = 0; r < 4; r++) {
            const int q = (lane >> 4) + 4 * r + 16 * qx;
            if(q < dc) {
              const double x = g.XA[gic + (int64_t)q * g.ldxa];
              const double y = Y1[tm][qx][r] + R[row * RS + q];
              const double v = fma(g.mlin[q], rl, fma(-x, rd, y));
              if(gi < g.N) g.part[((int64_t)blockIdx.y * dc + q) * g.N + gi] = g.pf * g.scale[q] * v;
            }
          }
      }
    }
    return;
  }

  constexpr int NPW = SQ ? PW_NPQ : PW_NP;
  double* mypartial = g.partial + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * NPW;
  {
    double out[PW_NP];
#pragma unroll
    for(int p = 0; p < PW_NP; p++) out[p] = 0.0;
#pragma unroll
    for(int q = 0; q < NEXP; q++) {
      out[2 * q] = s_d2e[q];
      out[2 * q + 1] = s_e[q];
    }
    out[4] = s_all;
    out[5] = s_lin;
#pragma unroll
    for(int p = 0; p < PW_NP; p++) {
      if(p < 2 * NEXP || p == 4 || p == 5) {
        const double rsum = pw_block_sum<NW>(out[p], sh);
        if(t == 0) mypartial[p] = rsum;
      } else if(t == 0) {
        mypartial[p] = 0.0;
      }
    }
  }
  if(SQ) {
    // sum_i (rho_i x_iq^2 - 2 x_iq Y(q,i)): this lane's dimensions are q = (lane >> 4) + 4 r + 16 qx, its rows lane & 15 of each of
    // the wave's 16-row tiles
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
          int64_t gi = i0 + wm * RW + tm * 16 + (lane & 15);
          if(gi > g.N - 1) gi = g.N - 1;                              // (rows past the end carry zero weights)
          const double x = (q < dc) ? g.XA[gi + (int64_t)q * g.ldxa] : 0.0;
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
    // ... + sum_n kappa_n x_nq^2: lane (lane >> 4, q = lane & 15) holds its columns' share; the four column groups of the wave are
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
      mypartial[PW_NP + t] = v;
    }
  }
}

// column means of X (N x D) -> mean[D]
__global__ void __launch_bounds__(256) pw_mean_kernel(const double* __restrict__ X, int64_t ldx, int64_t N, double* __restrict__ mean)
{
  __shared__ double sh[4];
  const int64_t q = blockIdx.x;
  double a = 0.0;
  for(int64_t i = threadIdx.x; i < N; i += 256) a += X[i + q * ldx];
  const double tot = pw_block_sum<4>(a, sh);
  if(threadIdx.x == 0) mean[q] = tot / (double)N;
}

// Xs = (X - mean) sc column-major (ld N), XT the same transposed (row i at XT + i dp, zero-padded to dp; may be null),
// n1 = |xs_i|^2.  mean / sc may be null (no centring / no scaling).
__global__ void __launch_bounds__(256) pw_prep_kernel(const double* __restrict__ X, int64_t ldx, int64_t N, int D, const double* __restrict__ mean,
                                                      const double* __restrict__ sc, double* __restrict__ Xs, double* __restrict__ XT, int dp,
                                                      double* __restrict__ n1)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if(i >= N) return;
  double acc = 0.0;
  for(int q = 0; q < dp; q++) {
    double v = 0.0;
    if(q < D) {
      v = X[i + (int64_t)q * ldx];
      if(mean) v -= mean[q];
      if(sc) v *= sc[q];
      if(Xs) Xs[i + (int64_t)q * N] = v;
      acc = fma(v, v, acc);
    }
    if(XT) XT[i * dp + q] = v;
  }
  n1[i] = acc;
}

__global__ void __launch_bounds__(256) pw_rows_reduce_kernel(const double* __restrict__ part, int nsplit, int D, int64_t N,
                                                             double* __restrict__ out, int64_t ldo)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int q = blockIdx.y;
  if(i >= N) return;
  double v = 0.0;
  for(int s = 0; s < nsplit; s++) v += part[((int64_t)s * D + q) * N + i];
  out[i + (int64_t)q * ldo] = v;
}

template <int NEXP, bool ROWS, bool SQ>
int launch_walk(const WalkArgs& g, dim3 grid, hipStream_t s)
{
  // waves / occupancy by input dimension: four 64 x 32 patches at two workgroups per CU up to D = 16, eight 32 x 32 patches at one
  // per CU beyond (where the four-wave form's per-lane state no longer fits 256 registers)
  // D = 9 .. 16: four waves at two workgroups per CU (N = 32 768, D = 16: 3.0 ms = 2.9 TB/s against 4.0 ms with eight waves at
  // one per CU -- the other way round from the rbfard parameter kernel, whose per-dimension sums cost it more registers);
  // GPC_PAIR_WALK_FORM=0 keeps the eight-wave form there (A/B)
  const char* e = getenv("GPC_PAIR_WALK_FORM");
  const int form = e ? atoi(e) : 1;
  if(g.D <= 4) hipLaunchKernelGGL((pair_walk_kernel<1, 4, 2, NEXP, ROWS, SQ>), grid, dim3(256), 0, s, g);
  else if(g.D <= 8) hipLaunchKernelGGL((pair_walk_kernel<2, 4, 2, NEXP, ROWS, SQ>), grid, dim3(256), 0, s, g);
  else if(g.D <= 16 && form == 1) hipLaunchKernelGGL((pair_walk_kernel<4, 4, 2, NEXP, ROWS, SQ>), grid, dim3(256), 0, s, g);
  else if(g.D <= 16) hipLaunchKernelGGL((pair_walk_kernel<4, 8, 1, NEXP, ROWS, SQ>), grid, dim3(512), 0, s, g);
  else hipLaunchKernelGGL((pair_walk_kernel<8, 8, 1, NEXP, ROWS, SQ>), grid, dim3(512), 0, s, g);
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

// slices of the column tiles: enough workgroups to fill the chip, long walks when the problem is large
void walk_grid(int64_t N, int64_t N2, int* per, dim3* grid)
{
  const int64_t rb = (N + PW_MI - 1) / PW_MI, tj = (N2 + PW_MJ - 1) / PW_MJ;
  int64_t nsplit = (768 + rb - 1) / rb;
  if(nsplit > tj) nsplit = tj;
  if(nsplit < 1) nsplit = 1;
  int64_t p = (tj + nsplit - 1) / nsplit;
  if(p < 1) p = 1;
  nsplit = (tj + p - 1) / p;
  if(nsplit > 65535) {
    nsplit = 65535;
    p = (tj + nsplit - 1) / nsplit;
    nsplit = (tj + p - 1) / p;
  }
  *per = (int)p;
  *grid = dim3((unsigned)rb, (unsigned)nsplit);
}

struct Prepared {   // one point set as a pass wants it
  const double* Xs;
  int64_t ld;
  const double* XT;
  const double* n;
};

}  // namespace

bool pair_walk_applies(int64_t N, int64_t N2, int64_t D)
{
  // (read per call: tests and A/B tools switch them; two getenv beside an O(N N2) pass)
  const char* e = getenv("GPC_PAIR_WALK");
  const int on = e ? atoi(e) : 1;
  // below ~2^21 pairs a launch of the scalar kernels is over before the preparation passes of this one are
  e = getenv("GPC_PAIR_WALK_MINPAIRS");
  const int64_t minpairs = e ? atoll(e) : (int64_t)1 << 21;
  return on && D >= 1 && D <= PW_DC && N >= 1 && N2 >= 1 && N * N2 >= minpairs;
}

// dL/dX of one chunk of a compound (<= 4 rbf, <= 1 rbfard, any lin / bias / white): the sub-passes -- rbf terms two at a time
// (the first with the linear term), the rbfard term alone on scaled inputs -- write their slices side by side and ONE reduction
// adds everything in a fixed order.  GPC_EUNSUPPORTED outside the domain (the caller keeps the scalar kernel).
int pair_walk_gradx(const gpc_kspec* ksp, const double* X, int64_t N, int64_t ldx, const double* X2, int64_t N2, int64_t ldx2,
                    int64_t D, const double* G, int64_t ldg, double* gX, int64_t ldgx, double pair_factor, hipStream_t s)
{
  if(!pair_walk_applies(N, N2, D)) return GPC_EUNSUPPORTED;
  KSpecDev ks;
  GPC_CHECK(collapse_kspec(ksp, D, &ks));
  if(ks.n_ard > 1 || ks.n_rbf > 4) return GPC_EUNSUPPORTED;
  for(int64_t q = 0; ks.n_ard == 1 && q < D; q++)
    if(!(ks.ard_scale[0][q] >= 0.0)) return GPC_EUNSUPPORTED;
  const bool same = (X == X2 && ldx == ldx2 && N == N2);
  const int dp = D > 16 ? 32 : 16;
  const int n_rbf_pass = (ks.n_rbf + 1) / 2;
  int npass = n_rbf_pass + (ks.n_ard ? 1 : 0);
  if(ks.lin_var != 0.0 && n_rbf_pass == 0) npass++;   // a linear term without rbf company: a pass of its own (weights G lin)
  if(npass == 0) {   // bias / white only: no dependence on X
    GPC_HIP_CHECK(hipMemset2DAsync(gX, sizeof(double) * (size_t)ldgx, 0, sizeof(double) * (size_t)N, (size_t)D, s));
    return GPC_OK;
  }
  int per = 1;
  dim3 grid;
  walk_grid(N, N2, &per, &grid);
  const int64_t nsplit = grid.y;
  // scratch: mean[32] sc[32] | per point set: Xs (n D), XT (n dp), norms (n) -- twice (unscaled / scaled) | slices
  const bool need_u = n_rbf_pass > 0 || ks.lin_var != 0.0, need_a = ks.n_ard > 0;
  const size_t per_set_A = (size_t)N * (size_t)(D + 1), per_set_B = (size_t)N2 * (size_t)(D + dp + 1);
  const size_t sets = (size_t)(need_u ? 1 : 0) + (size_t)(need_a ? 1 : 0);
  const size_t nprep = 64 + sets * (per_set_B + (same ? 0 : per_set_A));
  void* wx = nullptr;
  GPC_CHECK(workspace(WS_XSCALED, sizeof(double) * nprep, &wx));
  void* wp = nullptr;
  GPC_CHECK(workspace(WS_KERN, sizeof(double) * (size_t)npass * (size_t)nsplit * (size_t)D * (size_t)N, &wp));
  double* part = static_cast<double*>(wp);
  double* mean = static_cast<double*>(wx);
  double* scd = mean + 32;
  double* cur = scd + 32;
  double hmean[32], hsc[32];
  hipLaunchKernelGGL(pw_mean_kernel, dim3((unsigned)D), dim3(256), 0, s, X2, ldx2, N2, mean);
  for(int q = 0; q < 32; q++) hsc[q] = (q < D && ks.n_ard) ? sqrt(ks.ard_scale[0][q]) : 1.0;
  GPC_HIP_CHECK(hipMemcpyAsync(scd, hsc, sizeof(hsc), hipMemcpyHostToDevice, s));
  if(ks.lin_var != 0.0) {
    HostFetch f;   // the linear term multiplies the actual x_n = centred x_n + mean: the host needs the mean as a kernel argument
    GPC_CHECK(f.add(hmean, mean, sizeof(double) * (size_t)D, s));
    GPC_CHECK(f.finish(s));
  }
  auto prep = [&](bool scaled, Prepared* A, Prepared* B) {
    double* XsB = cur;  cur += (size_t)N2 * D;
    double* XTB = cur;  cur += (size_t)N2 * dp;
    double* nB = cur;   cur += N2;
    hipLaunchKernelGGL(pw_prep_kernel, dim3((unsigned)((N2 + 255) / 256)), dim3(256), 0, s, X2, ldx2, N2, (int)D, mean,
                       scaled ? scd : (const double*)nullptr, XsB, XTB, dp, nB);
    *B = Prepared{XsB, N2, XTB, nB};
    if(same) {
      *A = *B;
    } else {
      double* XsA = cur;  cur += (size_t)N * D;
      double* nA = cur;   cur += N;
      hipLaunchKernelGGL(pw_prep_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, X, ldx, N, (int)D, mean,
                         scaled ? scd : (const double*)nullptr, XsA, (double*)nullptr, dp, nA);
      *A = Prepared{XsA, N, nullptr, nA};
    }
  };
  Prepared Au{}, Bu{}, Aa{}, Ba{};
  if(need_u) prep(false, &Au, &Bu);
  if(need_a) prep(true, &Aa, &Ba);
  GPC_HIP_CHECK(hipGetLastError());
  WalkArgs g;
  memset(&g, 0, sizeof(g));
  g.G = G;
  g.ldg = ldg;
  g.N = N;
  g.N2 = N2;
  g.D = (int)D;
  g.jt_per_block = per;
  g.pf = pair_factor;
  int pass = 0;
  auto run = [&](const Prepared& A, const Prepared& B, int nexp) -> int {
    g.XA = A.Xs;  g.ldxa = A.ld;  g.nA = A.n;
    g.XB = B.Xs;  g.ldxb = B.ld;  g.nB = B.n;  g.XTB = B.XT;
    g.part = part + (size_t)pass * (size_t)nsplit * (size_t)D * (size_t)N;
    pass++;
    return nexp == 2 ? launch_walk<2, true, false>(g, grid, s) : launch_walk<1, true, false>(g, grid, s);
  };
  for(int q = 0; q < 32; q++) { g.scale[q] = 1.0; g.mlin[q] = 0.0; }
  bool lin_done = false;
  for(int p = 0; p < n_rbf_pass || (p == 0 && ks.lin_var != 0.0 && !lin_done); p++) {
    const int t0 = 2 * p, ne = (ks.n_rbf - t0 >= 2) ? 2 : (ks.n_rbf - t0 == 1 ? 1 : 0);
    for(int q = 0; q < 2; q++) {
      const bool on = q < ne;
      g.hiw[q] = on ? ks.rbf_hiw[t0 + q] : 0.0;
      g.coef[q] = on ? 2.0 * ks.rbf_hiw[t0 + q] * ks.rbf_var[t0 + q] : 0.0;
    }
    g.lin = (!lin_done) ? ks.lin_var : 0.0;
    for(int q = 0; q < 32; q++) g.mlin[q] = (g.lin != 0.0 && q < D) ? hmean[q] : 0.0;
    lin_done = true;
    GPC_CHECK(run(Au, Bu, ne == 2 ? 2 : 1));
  }
  if(ks.n_ard) {
    g.hiw[0] = ks.ard_hiw[0];
    g.coef[0] = 2.0 * ks.ard_hiw[0] * ks.ard_var[0];
    g.hiw[1] = g.coef[1] = 0.0;
    g.lin = 0.0;
    for(int q = 0; q < 32; q++) { g.scale[q] = hsc[q]; g.mlin[q] = 0.0; }
    GPC_CHECK(run(Aa, Ba, 1));
  }
  hipLaunchKernelGGL(pw_rows_reduce_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)D), dim3(256), 0, s, part, (int)((int64_t)pass * nsplit),
                     (int)D, N, gX, ldgx);
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

// The parameter sums of a CROSS Gram for one chunk of a compound (<= 4 rbf, <= 1 rbfard): S in the layout of gplvm.hip's
// kern_grad_cross_pass -- [2t], [2t+1] rbf term t: sum G k~ d2, sum G k~; [8], [9] the rbfard term's; [10] sum G; [11] sum G x.x2;
// [12 + q] rbfard: sum G k~ (x_iq - x2_nq)^2 (unscaled coordinates).  GPC_EUNSUPPORTED outside the domain.
int pair_walk_grad_cross(const KSpecDev& ks, const double* X, int64_t N, int64_t ldx, const double* X2, int64_t N2, int64_t ldx2,
                         int64_t D, const double* G, int64_t ldg, double* S, hipStream_t s)
{
  if(!pair_walk_applies(N, N2, D)) return GPC_EUNSUPPORTED;
  if(ks.n_ard > 1 || ks.n_rbf > 4) return GPC_EUNSUPPORTED;
  for(int64_t q = 0; ks.n_ard == 1 && q < D; q++)
    if(!(ks.ard_scale[0][q] > 1e-150)) return GPC_EUNSUPPORTED;   // (S_q is divided by s_q below)
  const int dp = D > 16 ? 32 : 16;
  int per = 1;
  dim3 grid;
  walk_grid(N, N2, &per, &grid);
  const int64_t nwg = (int64_t)grid.x * grid.y;
  // Scalar passes: the rbf terms two at a time on inputs CENTRED by the mean of X2 -- d2 = |xi|^2 + |xj|^2 - 2 xi.xj loses
  // eps |x|^2 / d2 of its accuracy on inputs far from the origin, which the scalar kernels this walk replaces did not (they formed
  // differences), and pair_walk_gradx centres for the same reason --; the first of them also gives sum G.  sum G x.x2 (the linear
  // term's) needs the ACTUAL dot products: a pass on the raw inputs -- the only one when there is no rbf term.
  const int n_plain = ks.n_rbf > 2 ? 2 : (ks.n_rbf > 0 ? 1 : 0);
  const bool centred = ks.n_rbf > 0;
  const size_t nprep = 64 + (size_t)N + (size_t)N2 + (centred ? ((size_t)N + (size_t)N2) * (size_t)(D + 1) : 0) +
                       (ks.n_ard ? (size_t)N * (size_t)(D + 1) + (size_t)N2 * (size_t)(D + dp + 1) : 0);
  void* wx = nullptr;
  GPC_CHECK(workspace(WS_XSCALED, sizeof(double) * nprep, &wx));
  void* wp = nullptr;
  GPC_CHECK(workspace(WS_KERN, sizeof(double) * (size_t)nwg * PW_NPQ, &wp));
  double* partial = static_cast<double*>(wp);
  double* mean = static_cast<double*>(wx);
  double* scd = mean + 32;
  double* nA = scd + 32;
  double* nB = nA + N;
  double* cur = nB + N2;
  for(int q = 0; q < 12 + GPC_MAX_ARD_DIM; q++) S[q] = 0.0;
  WalkArgs g;
  memset(&g, 0, sizeof(g));
  g.G = G;
  g.ldg = ldg;
  g.N = N;
  g.N2 = N2;
  g.D = (int)D;
  g.jt_per_block = per;
  g.partial = partial;
  auto fetch = [&](int np, std::vector<double>* sums) -> int {
    std::vector<double> h((size_t)nwg * (size_t)np);
    HostFetch f;
    GPC_CHECK(f.add(h.data(), partial, sizeof(double) * h.size(), s));
    GPC_CHECK(f.finish(s));
    sums->assign((size_t)np, 0.0);
    for(int q = 0; q < np; q++) {
      double acc = 0.0;
      for(int64_t b = 0; b < nwg; b++) acc += h[(size_t)b * (size_t)np + (size_t)q];
      (*sums)[(size_t)q] = acc;
    }
    return GPC_OK;
  };
  const bool raw_needed = ks.lin_var != 0.0 || (!centred && (ks.bias_var != 0.0 || ks.n_ard == 0));
  const bool plain_needed = centred || raw_needed;
  if(centred || ks.n_ard) hipLaunchKernelGGL(pw_mean_kernel, dim3((unsigned)D), dim3(256), 0, s, X2, ldx2, N2, mean);
  if(centred) {
    double* XcA = cur;  cur += (size_t)N * D;
    double* ncA = cur;  cur += N;
    double* XcB = cur;  cur += (size_t)N2 * D;
    double* ncB = cur;  cur += N2;
    hipLaunchKernelGGL(pw_prep_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, X, ldx, N, (int)D, mean,
                       (const double*)nullptr, XcA, (double*)nullptr, (int)D, ncA);
    hipLaunchKernelGGL(pw_prep_kernel, dim3((unsigned)((N2 + 255) / 256)), dim3(256), 0, s, X2, ldx2, N2, (int)D, mean,
                       (const double*)nullptr, XcB, (double*)nullptr, (int)D, ncB);
    GPC_HIP_CHECK(hipGetLastError());
    g.XA = XcA; g.ldxa = N;  g.nA = ncA;
    g.XB = XcB; g.ldxb = N2; g.nB = ncB;
    g.XTB = nullptr;
    for(int p = 0; p < n_plain; p++) {
      const int t0 = 2 * p, ne = (ks.n_rbf - t0 >= 2) ? 2 : 1;
      for(int q = 0; q < 2; q++) {
        g.hiw[q] = q < ne ? ks.rbf_hiw[t0 + q] : 0.0;
        g.coef[q] = 0.0;
      }
      if(ne == 2) GPC_CHECK((launch_walk<2, false, false>(g, grid, s)));
      else GPC_CHECK((launch_walk<1, false, false>(g, grid, s)));
      std::vector<double> sums;
      GPC_CHECK(fetch(PW_NP, &sums));
      for(int q = 0; q < ne; q++) {
        S[2 * (t0 + q)] = sums[(size_t)(2 * q)];
        S[2 * (t0 + q) + 1] = sums[(size_t)(2 * q + 1)];
      }
      if(p == 0) S[10] = sums[4];
    }
  }
  if(raw_needed) {
    hipLaunchKernelGGL(pw_prep_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, X, ldx, N, (int)D, (const double*)nullptr,
                       (const double*)nullptr, (double*)nullptr, (double*)nullptr, (int)D, nA);
    hipLaunchKernelGGL(pw_prep_kernel, dim3((unsigned)((N2 + 255) / 256)), dim3(256), 0, s, X2, ldx2, N2, (int)D, (const double*)nullptr,
                       (const double*)nullptr, (double*)nullptr, (double*)nullptr, (int)D, nB);
    GPC_HIP_CHECK(hipGetLastError());
    g.XA = X;  g.ldxa = ldx;  g.nA = nA;
    g.XB = X2; g.ldxb = ldx2; g.nB = nB;
    g.XTB = nullptr;
    g.hiw[0] = g.hiw[1] = g.coef[0] = g.coef[1] = 0.0;
    GPC_CHECK((launch_walk<1, false, false>(g, grid, s)));
    std::vector<double> sums;
    GPC_CHECK(fetch(PW_NP, &sums));
    S[10] = sums[4];
    S[11] = sums[5];
  }
  if(ks.n_ard) {
    double hsc[32];
    for(int q = 0; q < 32; q++) hsc[q] = q < D ? sqrt(ks.ard_scale[0][q]) : 1.0;
    GPC_HIP_CHECK(hipMemcpyAsync(scd, hsc, sizeof(hsc), hipMemcpyHostToDevice, s));
    double* XsA = cur;  cur += (size_t)N * D;
    double* nAs = cur;  cur += N;
    double* XsB = cur;  cur += (size_t)N2 * D;
    double* XTB = cur;  cur += (size_t)N2 * dp;
    double* nBs = cur;  cur += N2;
    hipLaunchKernelGGL(pw_prep_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, X, ldx, N, (int)D, mean, scd, XsA, (double*)nullptr, dp, nAs);
    hipLaunchKernelGGL(pw_prep_kernel, dim3((unsigned)((N2 + 255) / 256)), dim3(256), 0, s, X2, ldx2, N2, (int)D, mean, scd, XsB, XTB, dp, nBs);
    GPC_HIP_CHECK(hipGetLastError());
    g.XA = XsA; g.ldxa = N;  g.nA = nAs;
    g.XB = XsB; g.ldxb = N2; g.nB = nBs;
    g.XTB = XTB;
    g.hiw[0] = ks.ard_hiw[0];
    g.coef[0] = 1.0;
    g.hiw[1] = g.coef[1] = 0.0;
    GPC_CHECK((launch_walk<1, false, true>(g, grid, s)));
    std::vector<double> sums;
    GPC_CHECK(fetch(PW_NPQ, &sums));
    S[8] = sums[0];
    S[9] = sums[1];
    if(!plain_needed) S[10] = sums[4];
    for(int64_t q = 0; q < D; q++) S[12 + q] = sums[(size_t)(PW_NP + q)] / ks.ard_scale[0][q];
  }
  return GPC_OK;
}

}  // namespace gpc
