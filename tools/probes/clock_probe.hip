// What does one cycle cost?  s_memtime (clock64) against s_memrealtime (wall_clock64, 100 MHz) around chains of dependent
// fp64 FMAs / v_rcp_f64 / v_readlane, one wave and four waves.  (run on the GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(double* out, long long* st, int n)
{
  double x = out[threadIdx.x], y = 1.0000001;
  long long c0 = clock64(), w0 = wall_clock64();
  for(int i = 0; i < n; i++) {
#pragma unroll
    for(int u = 0; u < 64; u++) x = fma(x, y, 1e-9);
  }
  long long c1 = clock64(), w1 = wall_clock64();
  double z = x;
  for(int i = 0; i < n; i++) {
#pragma unroll
    for(int u = 0; u < 16; u++) z = __builtin_amdgcn_rcp(z) + 1.5;
  }
  long long c2 = clock64(), w2 = wall_clock64();
  double q = z;
  for(int i = 0; i < n; i++) {
#pragma unroll
    for(int u = 0; u < 32; u++) {
      union { double d; int i[2]; } v; v.d = q;
      v.i[0] = __builtin_amdgcn_readlane(v.i[0], u); v.i[1] = __builtin_amdgcn_readlane(v.i[1], u);
      q = fma(q, v.d, 1e-9);
    }
  }
  long long c3 = clock64(), w3 = wall_clock64();
  out[threadIdx.x] = x + z + q;
  if(threadIdx.x == 0) { st[0] = c1 - c0; st[1] = w1 - w0; st[2] = c2 - c1; st[3] = w2 - w1; st[4] = c3 - c2; st[5] = w3 - w2; }
}
int main()
{
  double* d; long long* st; hipMalloc(&d, 8 * 256); hipMemset(d, 0, 8 * 256); hipMalloc(&st, 64);
  for(int threads = 64; threads <= 256; threads *= 4)
    for(int rep = 0; rep < 3; rep++) {
      const int n = 200;
      hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d, st, n);
      long long h[6]; hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
      printf("threads %3d: fma chain: %.1f memtime ticks / %.2f ns per op | rcp+add: %.1f ticks / %.2f ns | readlane x2 + fma: %.1f ticks / %.2f ns\n",
             threads, (double)h[0] / (n * 64), h[1] * 10.0 / (n * 64), (double)h[2] / (n * 16), h[3] * 10.0 / (n * 16),
             (double)h[4] / (n * 32), h[5] * 10.0 / (n * 32));
    }
  return 0;
}
