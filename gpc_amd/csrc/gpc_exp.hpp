// gpc_exp.hpp -- exp(x) for the element epilogues of the Gram / gradient kernels (CRbfKern::computeElement,
// CKern.cpp:1147-1154: variance * exp(-0.5 * inverseWidth * dist2)), table-driven:
//     x = (64 m + j) ln2/64 + f,  |f| <= ln2/128:   exp(x) = 2^m * 2^(j/64) * (1 + f + f^2/2 + ... + f^5/120)
// 64 correctly rounded values of 2^(j/64) in LDS (one ds_read_b64 per call), a two-constant Cody-Waite reduction, a
// degree-5 polynomial (the first neglected term is f^6/720 < 3.5e-17), v_ldexp_f64.  Maximum relative error 2.3e-16 over
// [-700, 0] (tests: test_exp_primitive; libm: 1.3e-16) at ~20 vector instructions instead of the ~40 of ocml's exp (whose 11 Horner
// steps each compile to a v_mov_b64 of the coefficient + v_fmac_f64): the exponentials were 1.7 of the 8.0 ms of the N = 65 536
// Gram build and a third of the gradient passes' vector work.  x < -745.2 (and -inf) gives 0; NaN stays NaN.
#pragma once
#include <hip/hip_runtime.h>

namespace gpc {

__device__ static const double gpc_exp2_tab[64] = {
  0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
  0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
  0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
  0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
  0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
  0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
  0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
  0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
  0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
  0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
  0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
  0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
  0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
  0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
  0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
  0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0,
};

// fill a workgroup's LDS copy of the table (64 doubles); the caller's next barrier publishes it
__device__ __forceinline__ void gpc_exp_tab_fill(double* tab)
{
  if(threadIdx.x < 64) tab[threadIdx.x] = gpc_exp2_tab[threadIdx.x];
}

__device__ __forceinline__ double gpc_exp_tab(double x, const double* __restrict__ tab)
{
  const double n = __builtin_rint(x * 0x1.71547652b82fep+6);        // x * 64 / ln 2
  double f = fma(n, -0x1.62e42fefa0000p-7, x);                      // x - n (ln2/64): high part (35 bits, n * hi is exact) ...
  f = fma(n, -0x1.cf79abc9e3b3ap-46, f);                            // ... and low part
  const int ni = (int)n;
  const double T = tab[ni & 63];
  // p(f) - 1 = f + f^2 (1/2 + f/6 + f^2 (1 + f/5) / 24): Estrin's form -- four dependent steps instead of six, and every
  // instruction has at most one 64-bit literal (the others are inline constants), so none needs a coefficient copied first
  const double f2 = f * f;
  const double a = fma(f, 0x1.5555555555555p-3, 0.5);
  const double u = fma(f, 0x1.999999999999ap-3, 1.0);
  const double w = f2 * 0x1.5555555555555p-5;
  const double pm1 = fma(f2, fma(w, u, a), f);
  const double r = ldexp(fma(T, pm1, T), ni >> 6);
  // The reduction is exact only while n * hi is (|x| < ~2800); below the underflow threshold the answer is 0 whatever the
  // arithmetic above made of it (also for -inf).  A compare + select, so that NaN stays NaN.
  return (x < -745.2) ? 0.0 : r;
}

}  // namespace gpc
