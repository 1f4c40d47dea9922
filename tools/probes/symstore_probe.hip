// Store-pattern probe for the symmetric Gram build (gram.hip gram_sym_kernel): row block I (128 rows) walks the 64-column tiles
// left of its diagonal and writes every tile twice -- directly and mirrored -- with no compute.  Which run lengths / orders does
// HBM take best?  (run on the GPU box: hipcc --offload-arch=gfx950 -O3 symstore_probe.hip -o symstore_probe && ./symstore_probe)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// MODE 0: the kernel's pattern: per store instruction 16 lanes x 8 B (128 B) x 4 columns, direct and mirrored
// MODE 1: 512-byte runs: per instruction one column x 64 rows (direct), one mirrored column x 64 j
// MODE 2: MODE 0 with non-temporal stores
// MODE 3: MODE 1 with non-temporal stores
// MODE 4: direct only (MODE 0 pattern), MODE 5: mirror only (MODE 0 pattern)
template <int MODE>
__global__ void __launch_bounds__(256, 2) symwalk(double* K, int64_t ld, int64_t N, int per)
{
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w & 1, wn = w >> 1;
  const int64_t i0 = (int64_t)blockIdx.x * 128;
  int64_t tiles = 2 * ((int64_t)blockIdx.x + 1) - 2;          // full tiles strictly left of the diagonal block
  const int64_t jt0 = (int64_t)blockIdx.y * per;
  int64_t jt1 = jt0 + per;
  if(jt1 > tiles) jt1 = tiles;
  const double v = 1.0 + t;
  for(int64_t jt = jt0; jt < jt1; jt++) {
    const int64_t j0 = jt * 64;
    if(MODE == 0 || MODE == 2 || MODE == 4 || MODE == 5) {
      if(MODE != 5) {
        double* Kd = K + (i0 + wm * 64 + (lane & 15)) + (j0 + wn * 32 + (lane >> 4)) * ld;
#pragma unroll
        for(int tn = 0; tn < 2; tn++)
#pragma unroll
          for(int r = 0; r < 4; r++)
#pragma unroll
            for(int tm = 0; tm < 4; tm++) {
              double* p = Kd + (tn * 16 + 4 * r) * ld + tm * 16;
              if(MODE == 2) __builtin_nontemporal_store(v, p); else *p = v;
            }
      }
      if(MODE != 4) {
#pragma unroll
        for(int tn = 0; tn < 2; tn++) {
          double* Kc = K + (j0 + wn * 32 + tn * 16 + (lane & 15)) + (i0 + wm * 64 + (lane >> 4)) * ld;
#pragma unroll
          for(int u = 0; u < 16; u++) {
            if(MODE == 2) __builtin_nontemporal_store(v, Kc); else *Kc = v;
            Kc += 4 * ld;
          }
        }
      }
    } else {
      // direct: wave w owns columns 16 w .. 16 w + 15 of the tile, two instructions of 64 rows each per column
      double* Kd = K + (i0 + lane) + (j0 + 16 * w) * ld;
#pragma unroll
      for(int c = 0; c < 16; c++)
#pragma unroll
        for(int h = 0; h < 2; h++) {
          double* p = Kd + c * ld + 64 * h;
          if(MODE == 3) __builtin_nontemporal_store(v, p); else *p = v;
        }
      // mirror: K(j0 + lane, i): wave w owns mirrored columns i = i0 + 32 w .. + 31
      double* Km = K + (j0 + lane) + (i0 + 32 * w) * ld;
#pragma unroll
      for(int c = 0; c < 32; c++) {
        double* p = Km + c * ld;
        if(MODE == 3) __builtin_nontemporal_store(v, p); else *p = v;
      }
    }
  }
}
template <typename F> float timeit(F f)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); f(); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3;
}
int main(int argc, char** argv)
{
  const int64_t N = argc > 1 ? atoll(argv[1]) : 65536;
  const int64_t ld = N + (argc > 2 ? atoll(argv[2]) : 0);
  const int per = argc > 3 ? atoi(argv[3]) : 48;
  double* K; hipMalloc(&K, (size_t)(8 * ld * N));
  const int64_t nrb = N / 128;
  const dim3 grid((unsigned)nrb, (unsigned)((2 * nrb + per - 1) / per));
  const double bytes = 8.0 * (double)N * (double)N;    // both triangles (the diagonal blocks are skipped: 0.4 %)
  float ms = timeit([&]{ hipMemsetAsync(K, 0, (size_t)(8 * ld * N), 0); });
  printf("N=%lld ld=%lld per=%d   memset %.2f ms %.0f GB/s\n", (long long)N, (long long)ld, per, ms, 8.0 * ld * N / ms * 1e-6);
#define RUN(M, what) { float t = timeit([&]{ hipLaunchKernelGGL(symwalk<M>, grid, dim3(256), 0, 0, K, ld, N, per); }); printf("mode %d %-44s %7.3f ms  %6.0f GB/s\n", M, what, t, bytes * ((M == 4 || M == 5) ? 0.5 : 1.0) / t * 1e-6); }
  RUN(0, "kernel's pattern (128 B x 4 columns)")
  RUN(1, "512 B runs")
  RUN(2, "kernel's pattern, non-temporal")
  RUN(3, "512 B runs, non-temporal")
  RUN(4, "direct half only")
  RUN(5, "mirrored half only")
  return 0;
}
