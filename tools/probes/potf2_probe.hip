// Ablation of the potf2 column loop: which part costs ~1 us per column?  (run on the GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int V>
__global__ void __launch_bounds__(256) k(double* A, double* out)
{
  __shared__ double col[2][64];
  __shared__ double Lout[64 * 64];
  const int t = threadIdx.x, r = t & 63, g = t >> 6;
  double a[16];
  for(int q = 0; q < 16; q++) a[q] = A[r + (4 * q + g) * 64];
#pragma unroll 1
  for(int jq = 0; jq < 16; jq++) {
#pragma unroll
    for(int jj = 0; jj < 4; jj++) {
      const int j = 4 * jq + jj;
      if(V >= 1) { if(g == jj) col[jj & 1][r] = a[0]; }
      __syncthreads();
      if(V >= 1) {
        const double* cj = col[jj & 1];
        const double pj = cj[j];
        const double cr = cj[r];
        double lrj = cr;
        if(V >= 2) {
          if(g == jj) { const double d = sqrt(pj); Lout[j * 64 + r] = (r == j) ? d : cr / d; }
          lrj = cr * (1.0 / pj);
        }
        if(V == 3) {
#pragma unroll
          for(int q = 0; q < 16; q++) {
            const int c = 4 * (q + jq) + g;
            if(q + jq < 16 && c > j && c <= r) a[q] -= lrj * cj[c];
          }
        } else if(V == 4) {
          double cv[16];
#pragma unroll
          for(int q = 0; q < 16; q++) cv[q] = cj[(4 * (q + jq) + g) & 63];
#pragma unroll
          for(int q = 0; q < 16; q++) {
            const int c = 4 * (q + jq) + g;
            const bool on = (q + jq < 16) && (c > j) && (c <= r);
            a[q] -= on ? lrj * cv[q] : 0.0;
          }
        } else a[1] += lrj;
      }
    }
#pragma unroll
    for(int q = 0; q < 15; q++) a[q] = a[q + 1];
  }
  __syncthreads();
  double s = 0; for(int q = 0; q < 16; q++) s += a[q];
  out[t] = s + Lout[t];
}
template <int V> void run(double* A, double* out)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(1), dim3(256), 0, 0, A, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for(int i = 0; i < 200; i++) hipLaunchKernelGGL(k<V>, dim3(1), dim3(256), 0, 0, A, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("variant %d: %.2f us per launch (back-to-back)\n", V, ms / 200 * 1e3);
}
int main()
{
  double *A, *out; hipMalloc(&A, 64 * 64 * 8); hipMalloc(&out, 4096 * 8);
  double h[4096]; for(int i = 0; i < 4096; i++) h[i] = (i % 65 == 0) ? 70.0 : 1.0;
  hipMemcpy(A, h, sizeof(h), hipMemcpyHostToDevice);
  run<0>(A, out); run<1>(A, out); run<2>(A, out); run<3>(A, out); run<4>(A, out);
  // busy GPU in the background? run a second pass after a long spin kernel to see clock effects
  run<3>(A, out);
  return 0;
}
