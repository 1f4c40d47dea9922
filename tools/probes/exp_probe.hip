// Cost and accuracy of a leaner fp64 exp against ocml's (run on the GPU box).  Throughput form: 8 independent chains per
// lane, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
__device__ __forceinline__ double lean_exp(double x)
{
  // exp(x) = 2^k exp(r), k = rint(x log2 e), r = x - k ln2 (two-part ln2), |r| <= 0.3466; degree-13 Taylor in Horner form
  const double k = __builtin_rint(x * 1.4426950408889634);
  double r = fma(-k, 6.93147180369123816490e-01, x);
  r = fma(-k, 1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;          // 1/13!
  p = fma(p, r, 2.08767569878681e-09);        // 1/12!
  p = fma(p, r, 2.505210838544172e-08);       // 1/11!
  p = fma(p, r, 2.755731922398589e-07);       // 1/10!
  p = fma(p, r, 2.7557319223985893e-06);      // 1/9!
  p = fma(p, r, 2.48015873015873e-05);        // 1/8!
  p = fma(p, r, 1.984126984126984e-04);       // 1/7!
  p = fma(p, r, 1.388888888888889e-03);       // 1/6!
  p = fma(p, r, 8.333333333333333e-03);       // 1/5!
  p = fma(p, r, 4.1666666666666664e-02);      // 1/4!
  p = fma(p, r, 1.6666666666666666e-01);      // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  double y = __builtin_amdgcn_ldexp(p, (int)k);
  y = (x < -745.2) ? 0.0 : y;
  return y;
}
template <int V>
__global__ void k(const double* in, double* out, int n, int reps)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double a[8];
  for(int u = 0; u < 8; u++) a[u] = in[(i * 8 + u) % n];
  double s = 0.0;
  for(int r = 0; r < reps; r++) {
#pragma unroll
    for(int u = 0; u < 8; u++) {
      const double e = V ? lean_exp(a[u]) : exp(a[u]);
      s += e;
      a[u] = a[u] * 0.9999 - 1e-3 * e;
    }
  }
  out[i] = s;
}
__global__ void acc(const double* in, double* o0, double* o1, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n) { o0[i] = exp(in[i]); o1[i] = lean_exp(in[i]); }
}
int main()
{
  const int n = 1 << 20;
  std::vector<double> h(n);
  for(int i = 0; i < n; i++) h[i] = -50.0 * (double)i / n - ((i % 17) == 0 ? 700.0 * (double)i / n : 0.0);
  double *din, *d0, *d1; hipMalloc(&din, 8 * n); hipMalloc(&d0, 8 * n); hipMalloc(&d1, 8 * n);
  hipMemcpy(din, h.data(), 8 * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(acc, dim3(n / 256), dim3(256), 0, 0, din, d0, d1, n);
  std::vector<double> r0(n), r1(n);
  hipMemcpy(r0.data(), d0, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, 8 * n, hipMemcpyDeviceToHost);
  double worst_ocml = 0, worst_lean = 0;
  for(int i = 0; i < n; i++) {
    const long double t = expl((long double)h[i]);
    if(t > 1e-300L) {
      worst_ocml = fmax(worst_ocml, (double)fabsl((r0[i] - t) / t));
      worst_lean = fmax(worst_lean, (double)fabsl((r1[i] - t) / t));
    }
  }
  printf("max relative error vs long double: ocml %.3e  lean %.3e  (eps = 1.1e-16)\n", worst_ocml, worst_lean);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 4 * 4, reps = 200;
  for(int v = 0; v < 2; v++) {
    for(int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if(v == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, din, d0, n, reps);
      else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, din, d0, n, reps);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if(rep) printf("%s: %.3f ms for %.3g exps -> %.1f Gexp/s\n", v ? "lean" : "ocml", ms, (double)blocks * 256 * 8 * reps, (double)blocks * 256 * 8 * reps / ms * 1e-6);
    }
  }
  return 0;
}
