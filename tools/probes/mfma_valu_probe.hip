// mfma_valu_probe.hip -- do v_mfma_f64_16x16x4 and fp64 vector instructions overlap on gfx950?  One workgroup per CU; per SIMD either
// one wave or two; a wave runs NM matrix instructions and NV vector FMAs per loop iteration (independent chains).  Measurement aid
// for DESIGN section 3.5 (not part of the library).
//   build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_probe mfma_valu_probe.hip ; run: ./mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)
typedef double d4 __attribute__((ext_vector_type(4)));

// ROLE 0: every wave issues NM MFMA + NV FMA per iteration.  ROLE 1: waves 0..3 only MFMA, waves 4..7 only FMA (two waves per SIMD)
// F32: the vector work is fp32 FMAs instead of fp64
template <int NM, int NV, int ROLE, bool F32>
__global__ void __launch_bounds__(512) probe(double* out, int iters)
{
  const int w = threadIdx.x >> 6;
  d4 acc[4];
  for(int i = 0; i < 4; i++) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
  double v[8];
  float vf[8];
  for(int i = 0; i < 8; i++) { v[i] = 1.0 + i + threadIdx.x; vf[i] = 1.0f + i; }
  const bool do_m = (ROLE == 0) || (w < 4), do_v = (ROLE == 0) || (w >= 4);
  for(int it = 0; it < iters; it++) {
    if(do_m) {
#pragma unroll
      for(int i = 0; i < NM; i++) acc[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 3], 0, 0, 0);
    }
    if(do_v) {
#pragma unroll
      for(int i = 0; i < NV; i++) {
        if(F32) vf[i & 7] = __builtin_fmaf(vf[i & 7], 1.0000001f, 1e-9f);
        else v[i & 7] = __builtin_fma(v[i & 7], 1.0000000001, 1e-9);
      }
    }
  }
  double s = 0.0;
  for(int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for(int i = 0; i < 8; i++) s += v[i] + vf[i];
  if(s == 12345.678) out[0] = s;
}

template <int NM, int NV, int ROLE, bool F32>
static double run(int nthreads, const char* what)
{
  double* d;
  CHECK(hipMalloc(&d, 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 20000;
  probe<NM, NV, ROLE, F32><<<256, nthreads>>>(d, 100);
  CHECK(hipEventRecord(e0));
  probe<NM, NV, ROLE, F32><<<256, nthreads>>>(d, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double per_it_ns = ms * 1e6 / iters;
  printf("%-72s %8.3f ms  %7.1f ns per iteration\n", what, ms, per_it_ns);
  CHECK(hipFree(d));
  return per_it_ns;
}

int main()
{
  // one wave per SIMD (256 threads)
  run<16, 0, 0, false>(256, "1 wave/SIMD: 16 MFMA");
  run<0, 64, 0, false>(256, "1 wave/SIMD: 64 fp64 FMA");
  run<16, 64, 0, false>(256, "1 wave/SIMD: 16 MFMA + 64 fp64 FMA (same wave, independent)");
  run<0, 64, 0, true>(256, "1 wave/SIMD: 64 fp32 FMA");
  run<16, 64, 0, true>(256, "1 wave/SIMD: 16 MFMA + 64 fp32 FMA (same wave, independent)");
  // two waves per SIMD (512 threads)
  run<16, 0, 0, false>(512, "2 waves/SIMD: each 16 MFMA");
  run<0, 64, 0, false>(512, "2 waves/SIMD: each 64 fp64 FMA");
  run<16, 64, 0, false>(512, "2 waves/SIMD: each 16 MFMA + 64 fp64 FMA");
  run<16, 64, 1, false>(512, "2 waves/SIMD: one 16 MFMA, the other 64 fp64 FMA");
  run<16, 64, 1, true>(512, "2 waves/SIMD: one 16 MFMA, the other 64 fp32 FMA");
  return 0;
}
