// hop_probe.hip -- latency of one producer -> consumer hand-over through global memory between two workgroups, by placement (same XCD /
// different XCDs) and by the instructions used.  Measurement aid for panel_flow.hip's exchange buffer (not part of the library).
//   build: hipcc --offload-arch=gfx950 -O3 -o hop_probe hop_probe.hip ; run: ./hop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)

// mode 0: device-scope relaxed atomic store / load (sc1): what panel_flow.hip does
// mode 1: device-scope store, polled by an atomic OR 0 at workgroup scope (performed in this XCD's L2)
// mode 2: workgroup-scope store, polled by an atomic OR 0 at workgroup scope
// mode 3: device-scope store, polled by buffer_inv sc0 + a workgroup-scope load (L1 dropped, L2 may answer)
template <int MODE>
__device__ __forceinline__ void put(unsigned long long* p, unsigned long long v)
{
  if(MODE == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE>
__device__ __forceinline__ unsigned long long get(unsigned long long* p)
{
  if(MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if(MODE == 1 || MODE == 2) {   // (written as an instruction: the compiler turns an atomic OR 0 into a load)
    unsigned long long r, z = 0;
    asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p), "v"(z) : "memory");
    return r;
  }
  asm volatile("buffer_inv sc0" ::: "memory");
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int MODE>
__global__ void hop_kernel(unsigned long long* slots, int partner, int iters, long long* out, int* xcc)
{
  const int b = blockIdx.x;
  if(threadIdx.x == 0) xcc[b] = __builtin_amdgcn_s_getreg(6164) & 15;   // HW_REG_XCC_ID, bits 3:0
  if(b != 0 && b != partner) return;
  if(threadIdx.x != 0) return;
  unsigned long long* mine = slots + (b == 0 ? 0 : 32);     // 256 bytes apart
  unsigned long long* theirs = slots + (b == 0 ? 32 : 0);
  long long t0 = wall_clock64();
  int bad = 0;
  for(int i = 1; i <= iters && !bad; i++) {
    if(b == 0) put<MODE>(mine, (unsigned long long)i);
    int polls = 0;
    while(get<MODE>(theirs) != (unsigned long long)i) {
      if(++polls > (1 << 22)) { bad = 1; break; }
    }
    if(b != 0) put<MODE>(mine, (unsigned long long)i);
  }
  long long t1 = wall_clock64();
  if(bad) {   // let the partner out
    for(int i = 1; i <= iters; i++) __hip_atomic_store(mine, (unsigned long long)iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if(b == 0) { out[0] = t1 - t0; out[1] = bad; }
}

template <int MODE>
void run(unsigned long long* slots, long long* out, int* xcc, int partner, const char* what)
{
  const int iters = 2000;
  CHECK(hipMemset(slots, 0, 1024));
  hipLaunchKernelGGL(hop_kernel<MODE>, dim3(64), dim3(64), 0, 0, slots, partner, iters, out, xcc);
  CHECK(hipDeviceSynchronize());
  long long h[2];
  int hx[64];
  CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost));
  printf("mode %d partner %2d (xcc %d / %d) %-34s: %s  %.3f us per hop\n", MODE, partner, hx[0], hx[partner], what, h[1] ? "FAILED (stale)" : "ok",
         (double)h[0] / 100.0 / iters / 2.0);
}

int main()
{
  unsigned long long* slots;
  long long* out;
  int* xcc;
  CHECK(hipMalloc(&slots, 1024));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMalloc(&xcc, 64 * sizeof(int)));
  for(int rep = 0; rep < 2; rep++) {
    for(int partner : {8, 16, 1, 3}) {
      run<0>(slots, out, xcc, partner, "sc1 store, sc1 load");
      run<3>(slots, out, xcc, partner, "sc1 store, buffer_inv sc0 + sc0 load");
      if(partner % 8 == 0) {
        run<1>(slots, out, xcc, partner, "sc1 store, L2 atomic poll");
        run<2>(slots, out, xcc, partner, "sc0 store, L2 atomic poll");
      }
    }
  }
  int hx[64];
  CHECK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost));
  printf("xcc of workgroups 0..63:");
  for(int i = 0; i < 64; i++) printf(" %d", hx[i]);
  printf("\n");
  return 0;
}
