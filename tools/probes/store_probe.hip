// HBM write-pattern probe: column-major N x N fp64 matrix written in tiles, no compute.  (run on the GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef double double2_t __attribute__((ext_vector_type(2)));
// pattern A: block tile R x C, each wave-store = 64 lanes x 16 B contiguous rows (1 KiB run of one column)
template <int R, int C>
__global__ void __launch_bounds__(256) tileA(double* K, int64_t ld, int64_t tiles_i)
{
  const int64_t bi = blockIdx.x % tiles_i, bj = blockIdx.x / tiles_i;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const double2_t v = {1.0 + t, 2.0};
  for(int c = w; c < C; c += 4)
    for(int r = 2 * lane; r < R; r += 128)
      *reinterpret_cast<double2_t*>(K + bi * R + r + (bj * C + c) * ld) = v;
}
// pattern B: the MFMA epilogue: per store 16 lanes x 8 B contiguous (128 B) x 4 columns
template <int R, int C>
__global__ void __launch_bounds__(256) tileB(double* K, int64_t ld, int64_t tiles_i)
{
  const int64_t bi = blockIdx.x % tiles_i, bj = blockIdx.x / tiles_i;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for(int c0 = 4 * w; c0 < C; c0 += 16)
    for(int r0 = 0; r0 < R; r0 += 16)
      K[bi * R + r0 + (lane & 15) + (bj * C + c0 + (lane >> 4)) * ld] = 1.0 + t;
}
template <typename F> float timeit(F f)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 2;
}
#define RUN(KERN, R, C) { int64_t ti = N / R, tj = N / C; float ms = timeit([&]{ hipLaunchKernelGGL((KERN<R, C>), dim3((unsigned)(ti * tj)), dim3(256), 0, 0, K, N, ti); }); printf("%-6s R=%5d C=%3d : %7.1f GB/s\n", #KERN, R, C, bytes / ms * 1e-6); }
int main()
{
  const int64_t N = 32768; const double bytes = 8.0 * N * N;
  double* K; hipMalloc(&K, (size_t)bytes);
  float ms = timeit([&]{ hipMemsetAsync(K, 0, (size_t)bytes, 0); });
  printf("hipMemset        : %7.1f GB/s\n", bytes / ms * 1e-6);
  RUN(tileA, 128, 64) RUN(tileA, 128, 32) RUN(tileA, 256, 32) RUN(tileA, 512, 16) RUN(tileA, 1024, 8) RUN(tileA, 4096, 4) RUN(tileA, 32768, 1)
  RUN(tileB, 128, 64) RUN(tileB, 128, 32) RUN(tileB, 256, 16) RUN(tileB, 1024, 16)
  return 0;
}
