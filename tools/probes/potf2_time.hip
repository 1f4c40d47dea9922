// Phase timing of the diagonal-block kernels (run on the GPU box): hipcc --offload-arch=gfx950 -O3 -std=c++17
//   -I include -I gpc_amd/csrc tools/probes/potf2_time.hip -L gpc_amd/lib -lgpc_hip -o tools/probes/potf2_time
#include "../../gpc_amd/csrc/potrf.hip"
#include <stdio.h>
using namespace gpc;
int main()
{
  const int R = 64;
  std::vector<double> h(64 * 64);
  for(int j = 0; j < 64; j++)
    for(int i = 0; i < 64; i++) h[i + j * 64] = (i == j ? 65.0 : 1.0) + 0.01 * ((i * 7 + j * 3) % 5);
  for(int j = 0; j < 64; j++)
    for(int i = 0; i < j; i++) h[i + j * 64] = h[j + i * 64];
  double* d; int* info; long long* dbg;
  hipMalloc(&d, sizeof(double) * 64 * 64 * R); hipMalloc(&info, 4); hipMemset(info, 0, 4); hipMalloc(&dbg, 8 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int variant = 0; variant < 3; variant++) {
    for(int rep = 0; rep < 2; rep++) {
      for(int i = 0; i < R; i++) hipMemcpy(d + (size_t)i * 4096, h.data(), sizeof(double) * 4096, hipMemcpyHostToDevice);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for(int i = 0; i < R; i++) {
        if(variant == 0) hipLaunchKernelGGL(potf2_kernel, dim3(1), dim3(256), 0, 0, d + (size_t)i * 4096, (int64_t)64, 64, info, (int64_t)0);
        else if(variant == 1) hipLaunchKernelGGL((potf2_blk_kernel<8, 0>), dim3(1), dim3(256), 0, 0, d + (size_t)i * 4096, (int64_t)64, 64, info, (int64_t)0, (long long*)nullptr, 0, (double*)nullptr);
        else hipLaunchKernelGGL((potf2_blk_kernel<4, 0>), dim3(1), dim3(256), 0, 0, d + (size_t)i * 4096, (int64_t)64, 64, info, (int64_t)0, (long long*)nullptr, 0, (double*)nullptr);
      }
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if(rep) printf("variant %d: %.2f us per launch (back to back)\n", variant, ms * 1e3 / R);
    }
  }
  hipMemcpy(d, h.data(), sizeof(double) * 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((potf2_blk_kernel<8, 1>), dim3(1), dim3(256), 0, 0, d, (int64_t)64, 64, info, (int64_t)0, dbg, 0, (double*)nullptr);
  hipMemcpy(d, h.data(), sizeof(double) * 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((potf2_blk_kernel<8, 1>), dim3(1), dim3(256), 0, 0, d, (int64_t)64, 64, info, (int64_t)0, dbg, 0, (double*)nullptr);
  long long st[16]; hipMemcpy(st, dbg, sizeof(st), hipMemcpyDeviceToHost);
  printf("phases (10 ns ticks since entry): loads %lld | blocks %lld %lld %lld %lld | store %lld\n", st[1] - st[0], st[2] - st[0],
         st[3] - st[0], st[4] - st[0], st[5] - st[0], st[6] - st[0]);
  // the fused step kernel on a 8192-row panel
  {
    const int64_t M = 8192, ld = 8192 + 64;
    double *P, *ref; hipMalloc(&P, sizeof(double) * ld * 128); hipMalloc(&ref, sizeof(double) * 4096);
    std::vector<double> hp((size_t)ld * 128);
    for(size_t i = 0; i < hp.size(); i++) hp[i] = 0.001 * (double)((i * 7) % 13);
    for(int j = 0; j < 64; j++) for(int i = 0; i < 64; i++) hp[i + (size_t)j * ld] = (i >= j) ? h[i + j * 64] / 65.0 + (i == j) : 0.0;
    hipMemcpy(P, hp.data(), sizeof(double) * hp.size(), hipMemcpyHostToDevice);
    hipMemset(ref, 0, sizeof(double) * 4096);
    hipFuncSetAttribute(reinterpret_cast<const void*>(panel_step_kernel<false, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)STEP_LDS_1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(panel_step_kernel<true, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)STEP_LDS_2);
    for(int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL((panel_step_kernel<false, true, 1>), dim3(128), dim3(256), STEP_LDS_1, 0, P, ld, 64, P + 64, ld, M, (double*)nullptr, (int64_t)0, 0, (const double*)nullptr, (int64_t)0, dbg);
      hipMemcpy(st, dbg, sizeof(st), hipMemcpyDeviceToHost);
      printf("step<false> (10 ns ticks): loads %lld | blocks %lld %lld %lld %lld | end %lld\n", st[1] - st[0], st[2] - st[0], st[3] - st[0], st[4] - st[0], st[5] - st[0], st[6] - st[0]);
      hipLaunchKernelGGL((panel_step_kernel<true, true, 1>), dim3(128), dim3(512), STEP_LDS_2, 0, P, ld, 64, P + 64, ld, M, P + 64 + 64 * ld, ld, 64, ref, (int64_t)64, dbg);
      hipMemcpy(st, dbg, sizeof(st), hipMemcpyDeviceToHost);
      printf("step<true>  (10 ns ticks): loads %lld | blocks %lld %lld %lld %lld | end %lld\n", st[1] - st[0], st[2] - st[0], st[3] - st[0], st[4] - st[0], st[5] - st[0], st[6] - st[0]);
    }
  }
  int hi; hipMemcpy(&hi, info, 4, hipMemcpyDeviceToHost); printf("info %d\n", hi);
  return 0;
}
