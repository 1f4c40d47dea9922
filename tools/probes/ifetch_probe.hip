// ifetch_probe.hip -- does straight-line code that runs once (cold in the instruction cache, warm in L2) issue slower than the same
// instructions in a loop?  Measurement aid for panel_flow.hip, whose unrolled block body is ~160 KB of code (I-cache: 64 KB per CU pair).
//   build: hipcc --offload-arch=gfx950 -O3 -o ifetch_probe ifetch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

// 16 independent-enough double FMAs (four chains): 128 bytes of code
#define BODY16                                                                                                 \
  "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n" \
  "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n" \
  "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n" \
  "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"

// LINEAR: 16 * 768 = 12288 instructions straight (96 KB); LOOP: 16 * 16 instructions (2 KB) x 48 trips
template <bool LINEAR>
__global__ void __launch_bounds__(256) k(double* out, long long* ticks, int warm)
{
  double a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
  const double m = 0.999, s = 1e-3;
  for(int rep = 0; rep <= warm; rep++) {
    __syncthreads();
    const long long t0 = clock64();
    if(LINEAR) {
      asm volatile(".rept 768\n" BODY16 ".endr\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(s));
    } else {
      for(int i = 0; i < 48; i++) asm volatile(".rept 16\n" BODY16 ".endr\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(s));
    }
    const long long t1 = clock64();
    if(threadIdx.x == 0) ticks[blockIdx.x * 2 + (rep == warm && warm ? 1 : 0)] = t1 - t0;
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}

int main()
{
  const int nb = 256;
  double* out;
  long long* ticks;
  hipMalloc(&out, nb * 256 * 8);
  hipMalloc(&ticks, nb * 2 * 8);
  for(int lin = 0; lin < 2; lin++) {
    for(int pass = 0; pass < 2; pass++) {
      hipMemset(ticks, 0, nb * 2 * 8);
      if(lin) hipLaunchKernelGGL(k<true>, dim3(nb), dim3(256), 0, 0, out, ticks, 1);
      else hipLaunchKernelGGL(k<false>, dim3(nb), dim3(256), 0, 0, out, ticks, 1);
      hipDeviceSynchronize();
      std::vector<long long> h(nb * 2);
      hipMemcpy(h.data(), ticks, nb * 2 * 8, hipMemcpyDeviceToHost);
      std::vector<long long> first, second;
      for(int i = 0; i < nb; i++) { first.push_back(h[2 * i]); second.push_back(h[2 * i + 1]); }
      std::sort(first.begin(), first.end());
      std::sort(second.begin(), second.end());
      printf("%s launch %d: 12288 v_fma_f64 per wave, 4 waves per CU: first run median %.2f cycles/instr (min %.2f max %.2f), second run in the same workgroup median %.2f\n",
             lin ? "straight-line 96 KB " : "loop of 2 KB        ", pass, first[nb / 2] / 12288.0, first[0] / 12288.0, first[nb - 1] / 12288.0, second[nb / 2] / 12288.0);
    }
  }
  return 0;
}
