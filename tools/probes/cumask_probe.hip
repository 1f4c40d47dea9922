// Which CUs / XCDs does a stream created with hipExtStreamCreateWithCUMask run on?  (measurement aid for a panel stream beside a
// trailing-update stream.)  build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/cumask_probe.hip -o /tmp/cumask && /tmp/cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <string>

__global__ void where_kernel(unsigned* out, int spin)
{
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // keep the workgroup alive a little so that the grid spreads over every CU the queue may use
  long long t0 = clock64();
  while(clock64() - t0 < spin) {}
  if(threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw;
  }
}

static void run(const char* name, const std::vector<uint32_t>& mask)
{
  hipStream_t st;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
  if(e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return; }
  const int nblk = 4096;
  unsigned* d;
  hipMalloc(&d, sizeof(unsigned) * 2 * nblk);
  hipLaunchKernelGGL(where_kernel, dim3(nblk), dim3(256), 65536, st, d, 20000);
  hipStreamSynchronize(st);
  std::vector<unsigned> h(2 * nblk);
  hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * nblk, hipMemcpyDeviceToHost);
  std::map<unsigned, std::map<unsigned, int>> per;   // xcc -> (se, cu) -> blocks
  for(int b = 0; b < nblk; b++) {
    const unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    per[xcc][(se << 8) | (sh << 4) | cu]++;
  }
  int total = 0;
  printf("%s:", name);
  for(auto& x : per) { printf(" xcc%u:%zu", x.first, x.second.size()); total += (int)x.second.size(); }
  printf("  = %d distinct CUs\n", total);
  hipFree(d);
  hipStreamDestroy(st);
}

int main()
{
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
  printf("CUs %d, mask words %d\n", ncu, words);
  std::vector<uint32_t> all(words, 0xffffffffu);
  run("all", all);
  std::vector<uint32_t> m(words, 0);
  for(int i = 0; i < 32; i++) m[i / 32] |= 1u << (i % 32);
  run("bits 0..31", m);
  std::fill(m.begin(), m.end(), 0);
  for(int i = 0; i < ncu; i += 8) m[i / 32] |= 1u << (i % 32);
  run("every 8th bit", m);
  std::fill(m.begin(), m.end(), 0);
  for(int i = 0; i < ncu; i++) if(i % 8 != 0) m[i / 32] |= 1u << (i % 32);
  run("all but every 8th", m);
  std::fill(m.begin(), m.end(), 0);
  for(int i = 0; i < ncu; i++) if((i / 8) % 8 == 7) m[i / 32] |= 1u << (i % 32);
  run("bits 56..63 of every 64", m);
  std::fill(m.begin(), m.end(), 0);
  for(int i = ncu - 32; i < ncu; i++) m[i / 32] |= 1u << (i % 32);
  run("last 32 bits", m);
  return 0;
}
