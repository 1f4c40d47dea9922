// panel_flow.hip -- one Cholesky panel (dpotrf of the diagonal block + dtrsm of the rows below; CMatrix.cpp:371-403 via
// lapack.h:59-65) as ONE dataflow launch instead of the five dependent launches per 128 columns of potrf.hip's chain.
//
// One workgroup per 64 x 64 block (b, c) of the panel's lower trapezoid, ids from a ticket counter in column-major order, so
// that a workgroup only ever waits for blocks whose workgroups are already running or done: no deadlock whatever the
// residency.  Block (b, c):
//     C = A(b,c) - sum_{t<c} L(b,t) L(c,t)'       MFMA products, one 64-deep chunk per finished block pair, consumed AS THEY
//                                                 APPEAR (so only the last chunk is ever on the critical path)
//     b == c :  C = chol(C)                       8 columns at a time: one wave eliminates, the other three update (MFMA) and
//                                                 publish the previous group meanwhile
//     b >  c :  C = C L(c,c)^-T                   consuming L(c,c) 16 columns at a time, the rows in the MFMA accumulator
//                                                 layout throughout (every step an MFMA, no LDS staging of the solution)
// Finished blocks are PUBLISHED into an exchange buffer that starts as a sentinel NaN payload arithmetic never produces;
// consumers read it with device-scope atomic loads and poll the VALUES until they stop being the sentinel.  No flags and
// no fences: the XCDs' L2s are not coherent with each other, so an agent-scope release / acquire fence costs an L2
// write-back / invalidate per use (the first version of this kernel had two per column block and lost to the launch chain).
// A poll that is not answered after ~10 s, or a non-positive pivot anywhere, raises ctl[1]; everybody then leaves (the host
// sees LAPACK's info, or an error).  The critical path per 64 columns is chol(c,c) -> [the solve of (c+1,c) runs 16 columns
// behind it] -> last product chunk of (c+1,c+1) -> chol(c+1,c+1): 14.4 us (round 3: 18; a hand-over between two workgroups
// through the exchange buffer is 0.35 - 0.4 us whichever XCDs they run on -- tools/probes/hop_probe.hip).
#include "gpc_common.hpp"

namespace gpc {

namespace {

constexpr int PF_OS = 80;   // LDS stride of an operand stage [k][row] in doubles (= 16 mod 32: conflict-free fragment reads)
constexpr int PF_SS = 65;   // LDS column stride of the working block S[c * PF_SS + r]
constexpr int PF_LS = 66;   // row stride of the L image for the solve
constexpr int PF_TS = 18;   // row stride of potf2's multiplier table
#ifndef PF_LEAN_NS
#define PF_LEAN_NS 1
#endif
#ifndef PF_LEAN_MINROWS
#define PF_LEAN_MINROWS 5120   // launches of at least this many rows take the two-per-CU form (pf_lean)
#endif

struct PanelFlowArgs {
  double* P;          // the panel: M rows x nbk columns, leading dimension lda
  int64_t lda, M;
  int nbk, ncb;       // columns, column blocks of 64 (the last one may be narrower)
  int nrb;            // row blocks of 64
  int zb0, zshift;    // rows from block zb0 on hold an identity whose row block i is zero left of column block i - zshift (zb0 < 0: none)
  int64_t col0;       // global index of the panel's first column (for info)
  int* info;          // LAPACK info word (device)
  int* ctl;           // [0] ticket counter, [1] abort flag (1 = pivot failure, 2 = time-out)
  int trace;          // measurement aid: stamp pf_trace
  int max_polls;      // polls before a wait gives up (2^23: ~10 s; GPC_PANEL_FLOW_POLLS shortens it to provoke the time-out path)
  double* X;          // exchange buffer: every finished block, (64 nrb) x (64 ncb), leading dimension ldx
  int64_t ldx;
  // "given" mode (panel_flow_given: the triangular inversion of dpotri): the row blocks b < zb0 are NOT computed -- they are
  // the blocks of a finished factor G (leading dimension ldg, g_rows x g_cols real entries, continued as an identity beyond
  // them) and are only published; P holds the rows from block zb0 on (the identity that turns into L^-T), row 0 of P = row
  // 64 zb0 of the panel.
  const double* G;
  int64_t ldg, g_rows, g_cols;
  int store_cols;     // columns of P that exist (given mode: P need not have whole 64-column blocks)
  double* Xinv;       // given mode: the INVERSES of the factor's diagonal blocks, transposed ([c][k][n] = Linv_cc(n, k)), published
                      // by the diagonal blocks' workgroups next to the blocks themselves (nullptr: the rows substitute instead)
};

constexpr unsigned long long PF_SENT = 0xFFF8C0DEFACE0002ull;

__device__ __forceinline__ double pf_lane(double v, int lane)
{
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}

// one round of NV loads, `step` apart; true when none of this wave's values is the sentinel any more
template <int NV>
__device__ __forceinline__ bool pf_try(const double* p, int64_t step, double (&v)[NV])
{
  bool all = true;
#pragma unroll
  for(int i = 0; i < NV; i++) {
    const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p + i * step), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    all = all && (u != PF_SENT);
    v[i] = __longlong_as_double((long long)u);
  }
  return __builtin_amdgcn_read_exec() == __builtin_amdgcn_ballot_w64(all);
}

// ... repeated until they are all there.  false = somebody raised the abort flag (or nobody answered).
template <int NV>
__device__ __forceinline__ bool pf_fetch(const PanelFlowArgs& g, const double* p, int64_t step, double (&v)[NV])
{
  // (the poll limit is a constant: with the run-time limit as the loop bound the kernel's critical path grew from 17 to 26 us
  // per 64 columns -- cfg 2 5.5 -> 7.3 ms; the run-time limit, a testing aid, is looked at every 64th poll with the abort flag)
  for(int it = 0; it < (1 << 23); it++) {
    if(pf_try<NV>(p, step, v)) return true;
    if((it & 63) == 63) {
      if(__hip_atomic_load(&g.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
      if(it >= g.max_polls) break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  atomicCAS(&g.ctl[1], 0, 2);
  atomicExch(g.info, PANEL_FLOW_TIMEOUT);   // the host turns this into an error (read_info)
  return false;
}

__device__ __forceinline__ void pf_put(double* p, double x)
{
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void panel_flow_giveup_kernel(int* ctl, int* info)
{
  if(threadIdx.x == 0) {
    ctl[1] = 2;
    *info = PANEL_FLOW_TIMEOUT;
  }
}

__global__ void __launch_bounds__(256) panel_flow_init_kernel(int* ctl, unsigned long long* X, int64_t n)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if(i < 2) ctl[i] = 0;
  if(i < n) X[i] = PF_SENT;
}

__device__ long long pf_trace[64 * 64 * 4];   // measurement aid (GPC_PANEL_FLOW_TRACE): per block (b < 64, c < 64) four stamps

// NS: operand sets of the products in registers (NS - 1 requests in flight).  3 is the fast form (416 registers: a SIMD to
// itself); 2 fits in 256 registers, i.e. BESIDE one workgroup of the trailing-update GEMM on the same CU, which is what a
// look-ahead panel needs in order to start before the update has drained (see S below for the LDS side of the same story).
template <int NS>
__global__ void __launch_bounds__(256, (NS < 3 ? 2 : 1)) panel_flow_kernel(const PanelFlowArgs g)
{
  // one LDS arena, carved per phase:
  //   products : As = arena[0 .. 64*OS), Bs = arena[64*OS .. 128*OS)
  //   chol     : Pc, Wl, Tl = arena[0 .. 512), [512 .. 1024), [1024 .. 1536)   (careful loop: Pb = [0 .. 1024), T = next 64*TS)
  //   solve    : Ls = arena[0 .. 64*LS), Xs = next 16*OS
  //   S (the working block of the chol / solve phases) lives BEHIND the solve's carve-up, inside the part only the products
  //   use: the products are over before S is first written.  That keeps the workgroup at 82 KB of LDS instead of 115 --
  //   and 86 KB is what one retiring workgroup of the trailing-update GEMM leaves free on a CU (2 x 73.5 of 160 KB): at
  //   115 KB a panel launched beside a running update could not start before the update's LAST workgroups had gone
  //   (tools/overlap_probe.py: a 0.32 ms tile factorisation took 8.3 ms beside a 9.3 ms product)
  __shared__ __attribute__((aligned(16))) double arena[2 * 64 * PF_OS];
  constexpr int S_OFF = 64 * PF_LS + 16 * PF_OS;
  static_assert(S_OFF + 64 * PF_SS <= 2 * 64 * PF_OS && S_OFF >= 1024 + 64 * PF_TS, "S must fit behind the solve's / careful loop's carve-up");
  double* const S = arena + S_OFF;
  // two words in the padding of the arena's last operand row (columns 64 .. 79 of a stage row are never written or read by
  // the products, and no other phase reaches this far): the workgroup stays at exactly 80 KB, two to a CU
  int& tk_s = reinterpret_cast<int*>(arena + 2 * 64 * PF_OS - 2)[0];
  int& giveup = reinterpret_cast<int*>(arena + 2 * 64 * PF_OS - 2)[1];
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wv & 1, wn = wv >> 1;
  if(t == 0) {
    tk_s = atomicAdd(&g.ctl[0], 1);
    giveup = 0;
  }
  __syncthreads();
  // ticket -> block, column-major over the trapezoid: column c holds row blocks c .. nrb-1
  int c = 0, left = tk_s;
  while(left >= g.nrb - c) {
    left -= g.nrb - c;
    c++;
  }
  const int b = c + left;
  const int64_t r0 = (int64_t)b * 64;
  const int nr = (int)((g.M - r0 < 64) ? (g.M - r0) : 64);        // real rows
  const int ncol = (c == g.ncb - 1) ? (g.nbk - 64 * c) : 64;      // real columns
  const bool diag = (b == c);
  // rows of an identity riding below the matrix (chol_inverse's [K; I]): block (b, c) left of the identity's own diagonal is
  // zero and stays zero -- nothing to compute, nothing to publish, and nobody asks for it (products of such a row start at
  // its first non-zero column block)
  const int tstart = (g.zb0 >= 0 && b >= g.zb0 && b - g.zb0 - g.zshift > 0) ? (b - g.zb0 - g.zshift) : 0;
  if(c < tstart) return;
  if(g.G != nullptr && b < g.zb0) {
    // a block of the given factor: nothing to compute, its consumers (the products and solves of the rows below) find it in
    // the exchange buffer like a block that was just factored
#pragma unroll
    for(int i = 0; i < 16; i++) {
      const int m = lane, n = wv + 4 * i;
      const int64_t gm = r0 + m, gn = (int64_t)c * 64 + n;
      double v = (gm == gn) ? 1.0 : 0.0;
      if(gm < g.g_rows && gn < g.g_cols) v = (gm >= gn) ? g.G[gm + gn * g.ldg] : 0.0;
      pf_put(&g.X[r0 + m + gn * g.ldx], v);
    }
    if(diag && g.Xinv != nullptr) {
      // ... and the inverse of a diagonal block: nobody has to wait for it (the factor is given, this workgroup has nothing
      // else to do), and with it the step every row block walks through per column block -- from its last product to its own
      // solution -- is one 64-deep MFMA product instead of four dependent 16-column substitutions (13 -> 2 us; trsv_flow_kernel
      // of trsm.hip does the same for vectors).  Wave 0, lane = column j of L^-1: x_i = (delta_ij - sum_{k<i} L(i,k) x_k) / L(i,i),
      // the x_k in registers, L(i,k) a wave-uniform LDS operand.
      double* Lb = arena;                 // Lb[k * 65 + i] = L(i, k)
      double* Dv = arena + 64 * 65;       // 1 / L(i, i)
#pragma unroll
      for(int i = 0; i < 16; i++) {
        const int m = lane, n = wv + 4 * i;
        const int64_t gm = r0 + m, gn = (int64_t)c * 64 + n;
        double v = (gm == gn) ? 1.0 : 0.0;
        if(gm < g.g_rows && gn < g.g_cols) v = (gm >= gn) ? g.G[gm + gn * g.ldg] : 0.0;
        Lb[n * 65 + m] = v;
        if(m == n) Dv[m] = 1.0 / v;
      }
      __syncthreads();
      if(wv == 0) {
        double xi[64];
#pragma unroll
        for(int i = 0; i < 64; i++) {
          double sacc = (i == lane) ? 1.0 : 0.0;
#pragma unroll
          for(int k = 0; k < i; k++) sacc -= Lb[k * 65 + i] * xi[k];
          xi[i] = sacc * Dv[i];
        }
        double* out = g.Xinv + (int64_t)c * 4096 + lane * 64;     // [k = lane][n = i] = Linv(i, lane)
#pragma unroll
        for(int i = 0; i < 64; i++) pf_put(out + i, xi[i]);
      }
    }
    return;
  }
  const int64_t rp = (g.G != nullptr) ? r0 - (int64_t)g.zb0 * 64 : r0;   // row of this block in P
  const bool tr = g.trace && b < 64 && t == 0;
  if(tr) pf_trace[(b * 64 + c) * 4 + 0] = wall_clock64();

  // my block of the input, in the accumulator layout of the products below: lane l, register r of tile (tm, tn) holds
  // row m = wm*32 + tm*16 + (l & 15), column n = wn*32 + tn*16 + (l >> 4) + 4 r
  double4_t a0[2][2];
#pragma unroll
  for(int tn = 0; tn < 2; tn++)
#pragma unroll
    for(int tm = 0; tm < 2; tm++)
#pragma unroll
      for(int r = 0; r < 4; r++) {
        const int m = wm * 32 + tm * 16 + (lane & 15), n = wn * 32 + tn * 16 + (lane >> 4) + 4 * r;
        a0[tm][tn][r] = (m < nr && n < ncol && c * 64 + n < g.store_cols) ? g.P[rp + m + ((int64_t)c * 64 + n) * g.lda] : 0.0;
      }

  // ---- acc = sum_{t<c} L(b,t) L(c,t)' ------------------------------------------------------------------------------------------
  double4_t acc[2][2];
#pragma unroll
  for(int i = 0; i < 2; i++)
#pragma unroll
    for(int j = 0; j < 2; j++) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  if(c > tstart) {
    double* As = arena;
    double* Bs = arena + 64 * PF_OS;
    const double* pa = g.X + r0 + lane + (int64_t)wv * g.ldx;                 // row `lane` of row block b, column wv + 4 i
    const double* pb = g.X + (int64_t)c * 64 + lane + (int64_t)wv * g.ldx;    // ... of row block c
    // Full chunks t = 0 .. c-2, their operands requested TWO chunks ahead (one set of registers being staged and multiplied,
    // two in flight): a device-scope atomic load takes 2-3 us to come back, a 64-deep product 1.7 us, and one workgroup has
    // the CU to itself, so the second outstanding chunk is what keeps the matrix cores busy.  A request is one round of
    // loads; whoever finds a sentinel among them when the values are needed polls (pf_fetch).
    double va[NS][16], vb[NS][16];
    bool have[NS];
#pragma unroll
    for(int i = 0; i < NS; i++) have[i] = false;
    bool lost = false;
    auto request = [&](int slot, int tt) {
      bool ok = pf_try<16>(pa + (int64_t)tt * 64 * g.ldx, 4 * g.ldx, va[slot]);
      if(!diag) ok = pf_try<16>(pb + (int64_t)tt * 64 * g.ldx, 4 * g.ldx, vb[slot]) && ok;
      have[slot] = ok;
    };
    auto complete = [&](int slot, int tt) {
      if(!have[slot]) {
        if(!pf_fetch<16>(g, pa + (int64_t)tt * 64 * g.ldx, 4 * g.ldx, va[slot])) lost = true;
        if(!diag && !pf_fetch<16>(g, pb + (int64_t)tt * 64 * g.ldx, 4 * g.ldx, vb[slot])) lost = true;
      }
    };
    auto multiply = [&](int slot) {
      __syncthreads();                                                   // the previous chunk's fragment reads are done
#pragma unroll
      for(int i = 0; i < 16; i++) {
        As[(wv + 4 * i) * PF_OS + lane] = va[slot][i];
        Bs[(wv + 4 * i) * PF_OS + lane] = diag ? va[slot][i] : vb[slot][i];
      }
      __syncthreads();
    };
    auto products = [&]() {
#pragma unroll
      for(int kk = 0; kk < 16; kk++) {
        double a[2], bb[2];
        const int kr = kk * 4 + (lane >> 4);
#pragma unroll
        for(int s = 0; s < 2; s++) {
          a[s] = As[kr * PF_OS + wm * 32 + s * 16 + (lane & 15)];
          bb[s] = Bs[kr * PF_OS + wn * 32 + s * 16 + (lane & 15)];
        }
#pragma unroll
        for(int tn = 0; tn < 2; tn++)
#pragma unroll
          for(int tm = 0; tm < 2; tm++) acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(bb[tn], a[tm], acc[tm][tn], 0, 0, 0);
      }
    };
    const int nfull = c - 1 - tstart;                     // chunks t = tstart .. c-2
    constexpr int AHEAD = (NS == 1) ? 1 : NS - 1;         // chunks requested ahead of the one being multiplied
#pragma unroll
    for(int i = 0; i < AHEAD; i++)
      if(i < nfull) request(i % NS, tstart + i);
    // the slots rotate 0 .. NS-1; unrolled by NS so that every register index is static
    for(int t0 = 0; t0 < nfull; t0 += NS) {
#pragma unroll
      for(int u = 0; u < NS; u++) {
        const int tt = t0 + u;
        if(tt < nfull) {
          complete(u, tstart + tt);
          multiply(u);
          if(tt + AHEAD < nfull) request((u + AHEAD) % NS, tstart + tt + AHEAD);
          products();
        }
      }
    }
    // the last pair of blocks, (b, c-1) and (c, c-1), is the one still being solved when this block sits on the critical
    // path: taken 16 columns at a time, as the solves publish them, so that only a 16-deep product follows the last piece
    {
      const double* qa = pa + (int64_t)(c - 1) * 64 * g.ldx;
      const double* qb = pb + (int64_t)(c - 1) * 64 * g.ldx;
#pragma unroll 1
      for(int sub = 0; sub < 4; sub++) {
        double ua[4], ub[4];
        if(!pf_fetch<4>(g, qa + (int64_t)sub * 16 * g.ldx, 4 * g.ldx, ua)) lost = true;
        if(diag) {
#pragma unroll
          for(int i = 0; i < 4; i++) ub[i] = ua[i];
        } else if(!pf_fetch<4>(g, qb + (int64_t)sub * 16 * g.ldx, 4 * g.ldx, ub)) {
          lost = true;
        }
        __syncthreads();
#pragma unroll
        for(int i = 0; i < 4; i++) {
          As[(wv + 4 * i) * PF_OS + lane] = ua[i];
          Bs[(wv + 4 * i) * PF_OS + lane] = ub[i];
        }
        __syncthreads();
#pragma unroll
        for(int kk = 0; kk < 4; kk++) {
          double a[2], bb[2];
          const int kr = kk * 4 + (lane >> 4);
#pragma unroll
          for(int s = 0; s < 2; s++) {
            a[s] = As[kr * PF_OS + wm * 32 + s * 16 + (lane & 15)];
            bb[s] = Bs[kr * PF_OS + wn * 32 + s * 16 + (lane & 15)];
          }
#pragma unroll
          for(int tn = 0; tn < 2; tn++)
#pragma unroll
            for(int tm = 0; tm < 2; tm++) acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(bb[tn], a[tm], acc[tm][tn], 0, 0, 0);
        }
      }
    }
    if(lost) giveup = 1;
  }
  if(tr) pf_trace[(b * 64 + c) * 4 + 1] = wall_clock64();
  // ---- C = A(b,c) - acc -------------------------------------------------------------------------------------------------
  __syncthreads();   // the arena is free, giveup is final
  if(giveup) return;
  if(!diag) {
    // a block below the diagonal: into LDS, a column per register of a lane = row (the substitution's layout)
#pragma unroll
    for(int tn = 0; tn < 2; tn++)
#pragma unroll
      for(int tm = 0; tm < 2; tm++)
#pragma unroll
        for(int r = 0; r < 4; r++) {
          const int m = wm * 32 + tm * 16 + (lane & 15), n = wn * 32 + tn * 16 + (lane >> 4) + 4 * r;
          S[n * PF_SS + m] = a0[tm][tn][r] - acc[tm][tn][r];
        }
    __syncthreads();
    if(g.G != nullptr && g.Xinv != nullptr) {
      // X = C L(c,c)^-T as ONE product with the published inverse: X(m, n) = sum_k C(m, k) Linv(n, k).  C is in S ([k][m], row
      // stride PF_SS), the inverse goes into the first operand stage ([k][n]); the result comes out in the accumulator layout.
      double* Bi = arena;
      double li[16];
      if(!pf_fetch<16>(g, g.Xinv + (int64_t)c * 4096 + lane + wv * 64, 4 * 64, li)) giveup = 1;
#pragma unroll
      for(int i = 0; i < 16; i++) Bi[(wv + 4 * i) * PF_OS + lane] = li[i];
      __syncthreads();
      if(giveup) return;
      double4_t xacc[2][2];
#pragma unroll
      for(int i = 0; i < 2; i++)
#pragma unroll
        for(int j = 0; j < 2; j++) xacc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for(int kk = 0; kk < 16; kk++) {
        double a[2], bb[2];
        const int kr = kk * 4 + (lane >> 4);
#pragma unroll
        for(int q = 0; q < 2; q++) {
          a[q] = S[kr * PF_SS + wm * 32 + q * 16 + (lane & 15)];
          bb[q] = Bi[kr * PF_OS + wn * 32 + q * 16 + (lane & 15)];
        }
#pragma unroll
        for(int tn = 0; tn < 2; tn++)
#pragma unroll
          for(int tm = 0; tm < 2; tm++) xacc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(bb[tn], a[tm], xacc[tm][tn], 0, 0, 0);
      }
      const bool wanted = (c + 1 < g.ncb);
#pragma unroll
      for(int tn = 0; tn < 2; tn++)
#pragma unroll
        for(int tm = 0; tm < 2; tm++)
#pragma unroll
          for(int r = 0; r < 4; r++) {
            const int m = wm * 32 + tm * 16 + (lane & 15), n = wn * 32 + tn * 16 + (lane >> 4) + 4 * r;
            const double v = xacc[tm][tn][r];
            if(wanted) pf_put(&g.X[r0 + m + ((int64_t)c * 64 + n) * g.ldx], v);
            if(m < nr && c * 64 + n < g.store_cols) g.P[rp + m + ((int64_t)c * 64 + n) * g.lda] = v;
          }
      return;
    }
  }
  if(tr) pf_trace[(b * 64 + c) * 4 + 2] = wall_clock64();

  if(diag) {
    // ---- chol of the ncol x ncol diagonal block, 8 columns at a time (potf2_blk_kernel of potrf.hip); each group is
    //      normalised, stored and published as soon as it is final.  The block stays in the accumulator layout it was
    //      computed in (wave (wm, wn) holds the 32 x 32 quadrant): a group's 8 columns go through LDS to the wave that
    //      eliminates them (lane = row), and the rank-8 update of the rest is two MFMA steps per 16 x 16 tile (the same
    //      flops as the vector units but no broadcast LDS reads, which bound the vector form).  The strict upper triangle
    //      is never referenced: zero at the start, its tiles are skipped or hold harmless values of dead rows.
    double* Pc = arena;            // [8][64] the group's columns
    const int rr = lane, gq = wv;
    const int n = ncol;
    double4_t cur[2][2];
#pragma unroll
    for(int tn = 0; tn < 2; tn++)
#pragma unroll
      for(int tm = 0; tm < 2; tm++)
#pragma unroll
        for(int r = 0; r < 4; r++) {
          const int m = wm * 32 + tm * 16 + (lane & 15), nn = wn * 32 + tn * 16 + (lane >> 4) + 4 * r;
          double v = 0.0;
          if(m < n && nn < n) {
            if(m >= nn) v = a0[tm][tn][r] - acc[tm][tn][r];
          } else if(m == nn) {
            v = 1.0;            // identity padding of a ragged last block
          }
          cur[tm][tn][r] = v;
        }
    int done = 0;
    bool safe = true;
    // (round 4) one wave eliminates, the others update and publish.  The round-3 loop ran every group's elimination in all four
    // waves (lane = row, redundantly) and then its update and its publishing in all four: 1.3 - 1.5 us per group, all of it
    // on the critical path.  The quadrant (rows 0..31, columns 32..63) is strictly above the diagonal, so its wave has no
    // accumulators to update: it is the PIVOT wave.  Per group it reads the staged columns, eliminates them and writes all
    // eight multiplier / pivot columns to LDS; the other three meanwhile finish the PREVIOUS group (the update of the tiles
    // the next group does not need, the square roots, the stores, the publishing) on their own SIMDs.  After the barrier
    // they apply the new group to the one tile column the next group lives in and stage it.  On the critical path per group:
    // staged columns -> elimination -> barrier -> two MFMA -> stage -> barrier.
    double* WT = arena + 512;                           // two buffers of [Wl 8 x 64 | Tl 8 x 64]: -w_j (row operand), p_j (column operand)
    int& okflag = reinterpret_cast<int*>(arena + 2560)[0];
    const bool pw = (wv == 2);                          // wm = 0, wn = 1
    const int np = (wv == 3) ? 2 : wv;                  // the other three: publisher 0, 1, 2 takes the group's columns np, np + 3, np + 6
    if(wn == 0) {
#pragma unroll
      for(int tm = 0; tm < 2; tm++)
#pragma unroll
        for(int r2 = 0; r2 < 2; r2++) Pc[((lane >> 4) + 4 * r2) * 64 + wm * 32 + tm * 16 + (lane & 15)] = cur[tm][0][r2];
    }
    if(t == 0) okflag = 0;
    __syncthreads();
    // square roots, stores and publication of group pb's columns np, np + 3, np + 6 from the LDS copies (three chains interleaved)
    auto publish = [&](int pb) {
      const double* Wp = WT + (pb & 1) * 1024;
      const double* Tp = Wp + 512;
      double wj[3], xd[3];
#pragma unroll
      for(int i = 0; i < 3; i++) {
        const int j = (np + 3 * i < 8) ? np + 3 * i : np;      // (wave-uniform; a third column only for publishers 0 and 1)
        wj[i] = -Wp[j * 64 + rr];
        xd[i] = pf_lane(Tp[j * 64 + rr], 8 * pb + j);
      }
      double yy[3], gg[3], hh[3], rq[3], eq[3];
#pragma unroll
      for(int i = 0; i < 3; i++) yy[i] = __builtin_amdgcn_rsq(xd[i]);
#pragma unroll
      for(int i = 0; i < 3; i++) {
        gg[i] = xd[i] * yy[i];
        hh[i] = 0.5 * yy[i];
      }
#pragma unroll
      for(int i = 0; i < 3; i++) rq[i] = fma(-hh[i], gg[i], 0.5);
#pragma unroll
      for(int i = 0; i < 3; i++) {
        gg[i] = fma(gg[i], rq[i], gg[i]);
        hh[i] = fma(hh[i], rq[i], hh[i]);
      }
#pragma unroll
      for(int i = 0; i < 3; i++) eq[i] = fma(-gg[i], gg[i], xd[i]);
#pragma unroll
      for(int i = 0; i < 3; i++) gg[i] = fma(eq[i], hh[i], gg[i]);
#pragma unroll
      for(int i = 0; i < 3; i++) eq[i] = fma(-gg[i], gg[i], xd[i]);
#pragma unroll
      for(int i = 0; i < 3; i++) {
        if(np + 3 * i < 8) {
          const int cj = 8 * pb + np + 3 * i;
          const double d = fma(eq[i], hh[i], gg[i]);
          const double lv = (rr == cj) ? d : ((rr > cj) ? wj[i] * d : 0.0);      // (identity in the padding: p_j(j) = 1 there)
          pf_put(&g.X[r0 + rr + ((int64_t)c * 64 + cj) * g.ldx], lv);             // (first: the solves below are polling)
          if(rr < nr && cj < n && rr >= cj) g.P[r0 + rr + ((int64_t)c * 64 + cj) * g.lda] = lv;
        }
      }
    };
#pragma unroll
    for(int blk = 0; blk < 8; blk++) {
      if(safe) {
        double* Wl = WT + (blk & 1) * 1024;
        double* Tl = Wl + 512;
        const bool fine = (g.trace == 2) && b == 0 && t == 128;
        const bool fine0 = (g.trace == 2) && b == 0 && t == 0;
        if(fine) pf_trace[63 * 256 + blk * 8 + 0] = wall_clock64();
        if(fine0) pf_trace[61 * 256 + blk * 8 + 0] = wall_clock64();
        if(pw) {
          double p[8], w[8];
#pragma unroll
          for(int j = 0; j < 8; j++) p[j] = Pc[j * 64 + rr];
          if(fine) pf_trace[63 * 256 + blk * 8 + 1] = wall_clock64();
          bool ok = true;
#pragma unroll
          for(int j = 0; j < 8; j++) {
            const double pj = pf_lane(p[j], 8 * blk + j);
            // pj in [2^-930, 2^930) (1.1e-280 .. 9.1e279; zero, negative, Inf and NaN are outside): sign 0 and biased exponent in
            // [93, 1953), one unsigned compare of the high word -- the pivot is wave-uniform, so this runs on the scalar unit
            ok = ok && (((unsigned)(__double_as_longlong(pj) >> 32) - (93u << 20)) < ((1953u - 93u) << 20));
            double xx = __builtin_amdgcn_rcp(pj);
            double e = fma(-pj, xx, 1.0);
            xx = fma(xx, e, xx);
            e = fma(-pj, xx, 1.0);
            const double rp = fma(xx, e, xx);
            w[j] = p[j] * rp;
#pragma unroll
            for(int cc = j + 1; cc < 8; cc++) p[cc] -= w[j] * pf_lane(p[j], 8 * blk + cc);
          }
          if(fine) pf_trace[63 * 256 + blk * 8 + 2] = wall_clock64();
#pragma unroll
          for(int j = 0; j < 8; j++) {
            Wl[j * 64 + rr] = -w[j];
            Tl[j * 64 + rr] = p[j];
          }
          if(!ok && lane == 0) okflag = 1;
          if(fine) pf_trace[63 * 256 + blk * 8 + 3] = wall_clock64();
        } else if(blk > 0) {
          // the previous group: the tiles its successor did not need, then its publication
          const int pb = blk - 1;
          const double* Wp = WT + (pb & 1) * 1024;
          const double* Tp = Wp + 512;
          if(wm >= wn) {
#pragma unroll
            for(int tn = 0; tn < 2; tn++) {
              if(wn * 32 + tn * 16 + 8 > 8 * pb && wn * 2 + tn != blk / 2) {
#pragma unroll
                for(int tm = 0; tm < 2; tm++) {
                  if(wm > wn || tm >= tn) {
#pragma unroll
                    for(int kk = 0; kk < 2; kk++) {
                      const double fa = Wp[(kk * 4 + (lane >> 4)) * 64 + wm * 32 + tm * 16 + (lane & 15)];
                      const double fb = Tp[(kk * 4 + (lane >> 4)) * 64 + wn * 32 + tn * 16 + (lane & 15)];
                      cur[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb, fa, cur[tm][tn], 0, 0, 0);
                    }
                  }
                }
              }
            }
          }
          if(fine0) pf_trace[61 * 256 + blk * 8 + 1] = wall_clock64();
          publish(pb);
          if(fine0) pf_trace[61 * 256 + blk * 8 + 2] = wall_clock64();
        }
        __syncthreads();
        if(fine) pf_trace[63 * 256 + blk * 8 + 4] = wall_clock64();
        if(fine0) pf_trace[61 * 256 + blk * 8 + 3] = wall_clock64();
        if(okflag) {
          safe = false;      // (the same in every wave: the careful loop takes over from this group)
        } else {
          if(blk + 1 < 8 && wn == (blk + 1) / 4) {
            // the tile column the next group lives in: this group's update, then its columns to the pivot wave
            const int tn = ((blk + 1) / 2) & 1;
            if(wm >= wn) {
#pragma unroll
              for(int tm = 0; tm < 2; tm++) {
                if(wm > wn || tm >= tn) {
#pragma unroll
                  for(int kk = 0; kk < 2; kk++) {
                    const double fa = Wl[(kk * 4 + (lane >> 4)) * 64 + wm * 32 + tm * 16 + (lane & 15)];
                    const double fb = Tl[(kk * 4 + (lane >> 4)) * 64 + wn * 32 + tn * 16 + (lane & 15)];
                    cur[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb, fa, cur[tm][tn], 0, 0, 0);
                  }
                }
              }
            }
#pragma unroll
            for(int tm = 0; tm < 2; tm++)
#pragma unroll
              for(int r2 = 0; r2 < 2; r2++)
                Pc[((lane >> 4) + 4 * r2) * 64 + wm * 32 + tm * 16 + (lane & 15)] = cur[tm][tn][2 * ((blk + 1) & 1) + r2];
          }
          if(fine0) pf_trace[61 * 256 + blk * 8 + 4] = wall_clock64();
          done = blk + 1;
          if(blk + 1 < 8) __syncthreads();
          if(fine0) pf_trace[61 * 256 + blk * 8 + 5] = wall_clock64();
        }
      }
    }
    if(safe && !pw) publish(7);
    if(!safe) {
      // the careful loop (exact division, LAPACK's info on a non-positive pivot), from the first group the fast path refused:
      // the rest of the block goes through LDS into a column per register of a lane = row, column 8 blk + 4 q + gq in a[q]
      __syncthreads();
#pragma unroll
      for(int tn = 0; tn < 2; tn++)
#pragma unroll
        for(int tm = 0; tm < 2; tm++)
#pragma unroll
          for(int r = 0; r < 4; r++) {
            const int m = wm * 32 + tm * 16 + (lane & 15), nn = wn * 32 + tn * 16 + (lane >> 4) + 4 * r;
            S[nn * PF_SS + m] = (m >= nn) ? cur[tm][tn][r] : 0.0;
          }
      __syncthreads();
      double* Pb = arena;
      double* T = arena + 1024;
      double a[16];
#pragma unroll
      for(int q = 0; q < 16; q++) {
        const int cc = 8 * done + 4 * q + gq;
        a[q] = (cc < 64) ? S[cc * PF_SS + rr] : 0.0;
      }
      auto publish = [&](int blk, const double (&p)[8], const double (&w)[8]) {
#pragma unroll
        for(int j = 0; j < 8; j++) {
          if((j >> 1) == gq) {
            const int cj = 8 * blk + j;
            const double d = sqrt(pf_lane(p[j], cj));
            const double lv = (rr == cj) ? d : ((rr > cj) ? w[j] * d : 0.0);
            if(rr < nr && cj < n && rr >= cj) g.P[r0 + rr + ((int64_t)c * 64 + cj) * g.lda] = lv;
            pf_put(&g.X[r0 + rr + ((int64_t)c * 64 + cj) * g.ldx], lv);
          }
        }
      };
      bool failed = false;
#pragma unroll 1
      for(int blk = done; blk < 8; blk++) {
        double* Pc2 = Pb + (blk & 1) * 512;
#pragma unroll
        for(int i = 0; i < 2; i++) Pc2[(4 * i + gq) * 64 + rr] = a[i];
        __syncthreads();
        double p[8], w[8];
#pragma unroll
        for(int j = 0; j < 8; j++) p[j] = Pc2[j * 64 + rr];
#pragma unroll
        for(int j = 0; j < 8; j++) {
          const int cj = 8 * blk + j;
          const double pj = pf_lane(p[j], cj);
          if(!(pj > 0.0)) {
            if(t == 0) {
              atomicCAS(g.info, 0, (int)(g.col0 + 64 * c + cj + 1));
              atomicCAS(&g.ctl[1], 0, 1);
            }
            failed = true;
            break;
          }
          w[j] = p[j] * (1.0 / pj);
#pragma unroll
          for(int cc = j + 1; cc < 8; cc++) p[cc] -= w[j] * pf_lane(p[j], 8 * blk + cc);
        }
        if(failed) break;   // the pivot is wave-uniform and the same in all four waves: the whole workgroup leaves
        publish(blk, p, w);
        if(blk + 1 < 8) {
#pragma unroll
          for(int j = 0; j < 8; j++) T[rr * PF_TS + j] = p[j];
#pragma unroll
          for(int q = 2; q < 16; q++) {
            const int cc = 8 * blk + 4 * q + gq;
            if(cc < 64) {
              const double* tc = &T[cc * PF_TS];
              double sacc = a[q];
#pragma unroll
              for(int j = 0; j < 8; j++) sacc -= w[j] * tc[j];
              a[q] = sacc;
            }
          }
#pragma unroll
          for(int q = 0; q < 14; q++) a[q] = a[q + 2];
        }
      }
    }
  } else {
    // ---- C = C L(c,c)^-T, 16 columns at a time as L(c,c)'s workgroup publishes them (round 4 form).  Wave wv owns the rows
    //      16 wv .. 16 wv + 15 and keeps them in the MFMA accumulator layout for the whole solve: register r of 16-column tile tn
    //      at lane l is column 16 tn + 4 r + (l >> 4) of row l & 15.  In that layout register kk of a solved tile IS the row
    //      operand of the k-step 4 kk .. 4 kk + 3, so the update of the tiles to the right is four MFMA per tile with no staging,
    //      and the 16 x 16 triangle is four MFMA steps too (below).  The round-3 form solved all 64 rows redundantly in the
    //      four waves, lane = row, every multiplier an LDS broadcast read, and went through LDS twice per group: 3.2 us per
    //      group, which is why the block under the diagonal finished 4.5 us after it (now 2.6).
    double* Ls = arena;                // Ls[row * PF_LS + col]: the image of L(c,c)
    double* Bt = arena + 64 * PF_LS;   // [4][64]: the column operands of a group's four triangle steps
    const int q = lane >> 4, lr = lane & 15;
    const double* Lx = g.X + (int64_t)c * 64 + lane + ((int64_t)c * 64 + wv * 4) * g.ldx;   // L(c,c)(lane, 16 blk + 4 wv + u)
    const bool wanted = (c + 1 < g.ncb);                                                     // somebody's product operand
    double lv[4];
    bool have = pf_try<4>(Lx, g.ldx, lv);
    double4_t y[4];
#pragma unroll
    for(int tn = 0; tn < 4; tn++)
#pragma unroll
      for(int r = 0; r < 4; r++) y[tn][r] = S[(16 * tn + 4 * r + q) * PF_SS + 16 * wv + lr];
#pragma unroll
    for(int blk = 0; blk < 4; blk++) {
      const int o = blk * 16;
      const bool fine = (g.trace == 2) && b == 1 && c == 0 && t == 0;
      if(fine) pf_trace[62 * 256 + blk * 8 + 0] = wall_clock64();
      if(!have && !pf_fetch<4>(g, Lx + (int64_t)o * g.ldx, g.ldx, lv)) giveup = 1;
      if(fine) pf_trace[62 * 256 + blk * 8 + 1] = wall_clock64();
#pragma unroll
      for(int u = 0; u < 4; u++) Ls[lane * PF_LS + o + wv * 4 + u] = lv[u];
      __syncthreads();
      if(giveup) return;
      if(blk + 1 < 4) have = pf_try<4>(Lx + (int64_t)(o + 16) * g.ldx, g.ldx, lv);          // in flight during this block's work
      if(fine) pf_trace[62 * 256 + blk * 8 + 2] = wall_clock64();
      // The 16 x 16 triangle as FOUR MFMA steps, one per block of four columns K = 4 rk .. 4 rk + 3 (register rk of the tile).
      // With V = inv(L_KK) the step is  x_K = y_K V',  y_n -= sum_k x_k L(n, k) for the columns n right of K -- both linear
      // in y_K, so ONE product with the column operand
      //     B(n, j) = V(n - 4 rk, j)               n in K
      //             = - sum_k L(n, 4 rk + k) V(k, j)   n right of K          (0 left of K)
      // whose row operand is the register rk itself and whose accumulator input is the tile with that register cleared.
      // The operands depend on L(c,c) only, not on the rows: wave rk prepares block rk's (lane (n, j): column j of the 4 x 4
      // inverse by substitution, then its own entry) and hands it to the others through LDS.  No cross-lane traffic, no
      // dependent vector chain per column.  (The lane-swap substitution this replaces: 0.9 us per group, issue-bound.)
      {
        const int d0 = o + 4 * wv;
        auto recip = [](double dk) {       // (square roots of pivots: rcp + two Newton steps)
          double xx = __builtin_amdgcn_rcp(dk);
          double e = fma(-dk, xx, 1.0);
          xx = fma(xx, e, xx);
          e = fma(-dk, xx, 1.0);
          return fma(xx, e, xx);
        };
        const double* Ld = Ls + d0 * PF_LS + d0;          // the 4 x 4 diagonal block (the same for every lane)
        const double r0i = recip(Ld[0]), r1i = recip(Ld[PF_LS + 1]), r2i = recip(Ld[2 * PF_LS + 2]), r3i = recip(Ld[3 * PF_LS + 3]);
        const double l10 = Ld[PF_LS], l20 = Ld[2 * PF_LS], l21 = Ld[2 * PF_LS + 1];
        const double l30 = Ld[3 * PF_LS], l31 = Ld[3 * PF_LS + 1], l32 = Ld[3 * PF_LS + 2];
        const double v0 = (q == 0) ? r0i : 0.0;
        const double v1 = (((q == 1) ? 1.0 : 0.0) - l10 * v0) * r1i;
        const double v2 = ((((q == 2) ? 1.0 : 0.0) - l20 * v0) - l21 * v1) * r2i;
        const double v3 = (((((q == 3) ? 1.0 : 0.0) - l30 * v0) - l31 * v1) - l32 * v2) * r3i;
        const int i = lr - 4 * wv;
        const double* lrow = Ls + (o + lr) * PF_LS + d0;  // L(o + lr, d0 ..): used by the lanes right of the block
        const double below = -(((lrow[0] * v0 + lrow[1] * v1) + lrow[2] * v2) + lrow[3] * v3);
        const double inside = (i == 0) ? v0 : (i == 1) ? v1 : (i == 2) ? v2 : v3;
        Bt[wv * 64 + lane] = (i < 0) ? 0.0 : ((i < 4) ? inside : below);
      }
      __syncthreads();
      double bv[4];
#pragma unroll
      for(int rk = 0; rk < 4; rk++) bv[rk] = Bt[rk * 64 + lane];
#pragma unroll
      for(int rk = 0; rk < 4; rk++) {
        double4_t cin = y[blk];
        const double yk = cin[rk];
        cin[rk] = 0.0;
        y[blk] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[rk], yk, cin, 0, 0, 0);
      }
      if(fine) pf_trace[62 * 256 + blk * 8 + 3] = wall_clock64();
      double x[4], nx[4];
#pragma unroll
      for(int r = 0; r < 4; r++) {
        x[r] = y[blk][r];
        nx[r] = -x[r];
      }
      // the tiles to the right: y(:, 16 g2 ..) -= X L(16 g2 .., o ..)'
#pragma unroll
      for(int g2 = blk + 1; g2 < 4; g2++)
#pragma unroll
        for(int kk = 0; kk < 4; kk++) {
          const double lb = Ls[(16 * g2 + lr) * PF_LS + o + kk * 4 + q];
          y[g2] = __builtin_amdgcn_mfma_f64_16x16x4f64(lb, nx[kk], y[g2], 0, 0, 0);
        }
      if(fine) pf_trace[62 * 256 + blk * 8 + 4] = wall_clock64();
      // these 16 columns are final: publish and store them
#pragma unroll
      for(int r = 0; r < 4; r++) {
        const int col = o + 4 * r + q, row = 16 * wv + lr;
        if(wanted) pf_put(&g.X[r0 + row + ((int64_t)c * 64 + col) * g.ldx], x[r]);           // (first: somebody may be polling)
        if(row < nr && c * 64 + col < g.store_cols) g.P[rp + row + ((int64_t)c * 64 + col) * g.lda] = x[r];
      }
      if(fine) pf_trace[62 * 256 + blk * 8 + 5] = wall_clock64();
    }
  }
  if(tr) pf_trace[(b * 64 + c) * 4 + 3] = wall_clock64();
}

}  // namespace

// which form of the kernel a launch of M rows takes (see panel_flow)
static bool pf_lean(int64_t M)
{
  static const int lean_env = [] { const char* e = getenv("GPC_PANEL_FLOW_LEAN"); return e ? atoi(e) : -1; }();
  static const int64_t minrows = [] { const char* e = getenv("GPC_PANEL_FLOW_LEAN_MINROWS"); return e ? atoll(e) : (int64_t)PF_LEAN_MINROWS; }();
  return lean_env >= 0 ? (lean_env > 0) : (M >= minrows);
}

// Factor the nbk-column panel whose diagonal block starts at P (M rows, M >= nbk): one launch.  Returns GPC_EUNSUPPORTED
// when the shape is outside what the kernel takes (the caller then runs the launch chain).  zero_row0 >= 0: the rows from
// zero_row0 on (a multiple of 64, relative to the panel) are an identity block whose 64-row block i is still zero left of
// column block i - zero_shift of this panel; those blocks are skipped.
int panel_flow(int64_t M, int64_t nbk, double* P, int64_t lda, int* d_info, int64_t col0, hipStream_t s, int64_t zero_row0,
               int64_t zero_shift)
{
  if(M < nbk || nbk <= 0 || nbk > 4096) return GPC_EUNSUPPORTED;
  const int64_t nrb = (M + 63) / 64;
  const int ncb = (int)((nbk + 63) / 64);
  // the last diagonal block may be narrower than 64, but then it has to be the last ROW block too (a panel that ends the matrix)
  if(nbk % 64 != 0 && M != nbk) return GPC_EUNSUPPORTED;
  const int64_t ldx = 64 * nrb, nx = ldx * 64 * ncb;
  void* ws = nullptr;
  GPC_CHECK(workspace(WS_FLOW, sizeof(double) * (size_t)nx + 64, &ws));
  double* X = static_cast<double*>(ws);
  int* ctl = reinterpret_cast<int*>(X + nx);
  hipLaunchKernelGGL(panel_flow_init_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, s, ctl,
                     reinterpret_cast<unsigned long long*>(X), nx);
  PanelFlowArgs g;
  g.P = P;
  g.lda = lda;
  g.M = M;
  g.nbk = (int)nbk;
  g.ncb = ncb;
  g.nrb = (int)nrb;
  g.zb0 = (zero_row0 >= 0 && zero_row0 % 64 == 0) ? (int)(zero_row0 / 64) : -1;
  g.zshift = (int)zero_shift;
  g.col0 = col0;
  g.info = d_info;
  g.ctl = ctl;
  g.X = X;
  g.ldx = ldx;
  g.G = nullptr;
  g.ldg = g.g_rows = g.g_cols = 0;
  g.store_cols = ncb * 64;
  g.Xinv = nullptr;
  static const int trace = [] { const char* e = getenv("GPC_PANEL_FLOW_TRACE"); return e ? atoi(e) : 0; }();
  g.trace = trace;
  static const int polls = [] { const char* e = getenv("GPC_PANEL_FLOW_POLLS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : (1 << 23); }();
  g.max_polls = polls;
  // GPC_PANEL_FLOW_POLLS < 64 (tests): the give-up state is raised before the launch, so that the host's handling of a time-out
  // (read_info -> the launch-chain retry of gpc_gp_update_k_f64 / the grid) is exercised every time, not when a wait happens
  // to outlast the limit
  if(polls < 64) hipLaunchKernelGGL(panel_flow_giveup_kernel, dim3(1), dim3(64), 0, s, ctl, d_info);
  const int64_t nblocks = (int64_t)ncb * nrb - (int64_t)ncb * (ncb - 1) / 2;
  // The two-per-CU form (one operand set, 256 registers, 124 bytes of scratch) against the one-per-CU form (three operand sets,
  // 416 registers).  Where the critical path rules the two are level or the fat form wins (N = 1000: 0.30 against 0.32 ms, N =
  // 4096 as one launch: 1.21 against 1.19); a TALL panel is bound by its blocks' products, and two workgroups per CU cover each
  // other's polls and LDS round trips: N = 7168 3.65 -> 3.47 ms, 8192 4.90 -> 4.71, 10 240 8.45 -> 8.06, 12 288 13.35 -> 12.64, 16 384
  // 28.3 -> 27.6 (tools/factor_sweep.py under GPC_PANEL_FLOW_LEAN_MINROWS; 5120 ... 6144 rows: level).  So the form is
  // chosen per launch by the panel's height (pf_lean(); GPC_PANEL_FLOW_LEAN = 0 / 1 forces one form, GPC_PANEL_FLOW_LEAN_MINROWS
  // moves the height).  The lean form also starts beside a running trailing update (a tile factorisation launched 1 ms into a
  // 9.2 ms update: done after 2.0 ms instead of 8.1), but look-ahead with it buys nothing on one GPU (cfg 3: 1393 against 1389
  // ms, N = 8192: 5.28 against 4.72 -- the panels' work is conserved, not hidden).
  if(pf_lean(M))
    hipLaunchKernelGGL(panel_flow_kernel<PF_LEAN_NS>, dim3((unsigned)nblocks), dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL(panel_flow_kernel<3>, dim3((unsigned)nblocks), dim3(256), 0, s, g);
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

// The columns [0, nbk) of X L'^T = E for the rows of E that are not zero there, as ONE dataflow launch: G is the finished
// factor's block that starts at the panel's diagonal (g_rows x g_cols real entries, leading dimension ldg; continued as an
// identity, so ragged sizes need no padded copy), E (id_rows x 64 ncb, leading dimension lde) holds the rows -- a right-hand
// side whose 64-row block i is zero left of column block i - zero_shift of this panel (an identity further along its own
// diagonal) -- and is overwritten by the solution.  The same blocks, products and substitution as the factorisation's
// launch; the factor's blocks are published instead of computed (dpotri's V = L^-T: potrf.hip, trtri_flow).
int panel_flow_given(int64_t id_rows, int64_t nbk, double* E, int64_t lde, const double* G, int64_t ldg, int64_t g_rows, int64_t g_cols,
                     int64_t zero_shift, int64_t store_cols, int* d_info, hipStream_t s)
{
  if(id_rows <= 0 || nbk <= 0 || nbk > 4096) return GPC_EUNSUPPORTED;
  const int ncb = (int)((nbk + 63) / 64);
  const int64_t M = (int64_t)ncb * 64 + id_rows, nrb = (M + 63) / 64;
  // The rows multiply with the published inverse of L(c,c) instead of substituting through it where the launch is bound by its
  // chain of column blocks -- everything in one launch, i.e. more than 1024 columns (dpotri at N = 2048 0.64 -> 0.57 ms, 4096
  // 2.56 -> 2.43); the 1024-column panels of larger problems are bound by their products and lose 2-3 % to the extra LDS
  // round trip (8192: 9.24 -> 9.54 ms).  GPC_FLOW_GIVEN_INV=0 / 1 forces it off / on.
  static const int inv_env = [] { const char* e = getenv("GPC_FLOW_GIVEN_INV"); return e ? atoi(e) : -1; }();
  const int use_inv = inv_env >= 0 ? inv_env : (nbk > 1024 ? 1 : 0);
  const int64_t ldx = 64 * nrb, nx = ldx * 64 * ncb, ninv = use_inv ? (int64_t)ncb * 4096 : 0;
  void* ws = nullptr;
  GPC_CHECK(workspace(WS_FLOW, sizeof(double) * (size_t)(nx + ninv) + 64, &ws));
  double* X = static_cast<double*>(ws);
  int* ctl = reinterpret_cast<int*>(X + nx + ninv);
  hipLaunchKernelGGL(panel_flow_init_kernel, dim3((unsigned)((nx + ninv + 255) / 256)), dim3(256), 0, s, ctl,
                     reinterpret_cast<unsigned long long*>(X), nx + ninv);
  PanelFlowArgs g;
  g.Xinv = use_inv ? X + nx : nullptr;
  g.P = E;
  g.lda = lde;
  g.M = M;
  g.nbk = ncb * 64;      // whole column blocks (G continues as an identity; E's missing columns read as zero and are not stored)
  g.ncb = ncb;
  g.nrb = (int)nrb;
  g.zb0 = ncb;
  g.zshift = (int)zero_shift;
  g.col0 = 0;
  g.info = d_info;
  g.ctl = ctl;
  g.X = X;
  g.ldx = ldx;
  g.G = G;
  g.ldg = ldg;
  g.g_rows = g_rows;
  g.g_cols = g_cols;
  g.store_cols = (int)store_cols;
  g.trace = 0;
  static const int polls = [] { const char* e = getenv("GPC_PANEL_FLOW_POLLS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : (1 << 23); }();
  g.max_polls = polls;
  if(polls < 64) hipLaunchKernelGGL(panel_flow_giveup_kernel, dim3(1), dim3(64), 0, s, ctl, d_info);
  const int64_t nblocks = (int64_t)ncb * nrb - (int64_t)ncb * (ncb - 1) / 2;
  // the two-per-CU form from the same height as the factorisation's launches (dpotri at N = 3072 1.20 -> 1.10 ms, 4096 2.07 ->
  // 1.87, 8192 8.13 -> 7.9, 12 288 23.8 -> 22.8, 16 384 51.6 -> 50.5; N = 2048, a launch of 4096 rows, loses: 0.56 -> 0.61)
  if(pf_lean(M))
    hipLaunchKernelGGL(panel_flow_kernel<PF_LEAN_NS>, dim3((unsigned)nblocks), dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL(panel_flow_kernel<3>, dim3((unsigned)nblocks), dim3(256), 0, s, g);
  GPC_HIP_CHECK(hipGetLastError());
  return GPC_OK;
}

}  // namespace gpc

// measurement aid: the stamps of the last traced panel_flow launch (wall_clock64 ticks, 100 MHz), [row block][column block][4]
extern "C" int gpc_debug_panel_flow_trace(long long* out, int64_t n)
{
  if(n > 64 * 64 * 4) n = 64 * 64 * 4;
  GPC_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(gpc::pf_trace), sizeof(long long) * (size_t)n));
  return GPC_OK;
}
