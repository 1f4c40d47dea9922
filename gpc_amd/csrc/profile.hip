// profile.hip -- measurement hooks: HIP-event bracketing of the dominant launches and an fp64 MFMA issue-rate probe.
#include "gpc_common.hpp"
#include <vector>

namespace gpc {

namespace {
struct Rec {
  hipEvent_t a, b;
  double work;
};
bool g_on = false;
std::vector<Rec> g_recs[PROF_NKINDS];
std::vector<Rec> g_pool;
bool g_open[PROF_NKINDS] = {false, false};

Rec take()
{
  Rec r;
  if(!g_pool.empty()) {
    r = g_pool.back();
    g_pool.pop_back();
  } else {
    (void)hipEventCreate(&r.a);
    (void)hipEventCreate(&r.b);
  }
  r.work = 0.0;
  return r;
}

// 4 independent accumulator chains per wave, 8 waves per CU-slot: pure issue-rate measurement
__global__ void __launch_bounds__(256) mfma_f64_probe_kernel(double* out, int iters)
{
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for(int i = 0; i < iters; i++) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  const double4_t s = c0 + c1 + c2 + c3;
  if(s.x == 123.456) out[0] = s.x + s.y + s.z + s.w;  // keep the chain alive
}
}  // namespace

void prof_begin(int kind, double work, hipStream_t s)
{
  if(!g_on || kind < 0 || kind >= PROF_NKINDS) return;
  Rec r = take();
  r.work = work;
  (void)hipEventRecord(r.a, s);
  g_recs[kind].push_back(r);
  g_open[kind] = true;
}

void prof_end(int kind, hipStream_t s)
{
  if(!g_on || kind < 0 || kind >= PROF_NKINDS || !g_open[kind]) return;
  (void)hipEventRecord(g_recs[kind].back().b, s);
  g_open[kind] = false;
}

}  // namespace gpc

using namespace gpc;

extern "C" int gpc_profile_enable(int on)
{
  GPC_CHECK(ensure_device());
  g_on = on != 0;
  return GPC_OK;
}

extern "C" int gpc_profile_read(int kind, int64_t* launches, double* total_ms, double* work, int reset)
{
  GPC_CHECK(ensure_device());
  GPC_REQUIRE(kind >= 0 && kind < PROF_NKINDS, "profile kind");
  GPC_HIP_CHECK(hipDeviceSynchronize());
  double ms = 0.0, w = 0.0;
  int64_t n = 0;
  for(Rec& r : g_recs[kind]) {
    float t = 0.f;
    if(hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
      ms += t;
      w += r.work;
      n++;
    }
  }
  (void)hipGetLastError();
  if(launches) *launches = n;
  if(total_ms) *total_ms = ms;
  if(work) *work = w;
  if(reset) {
    for(Rec& r : g_recs[kind]) g_pool.push_back(r);
    g_recs[kind].clear();
  }
  return GPC_OK;
}

extern "C" int gpc_probe_mfma_f64(double* tflops, void* stream)
{
  GPC_CHECK(ensure_device());
  GPC_REQUIRE(tflops != nullptr, "null output");
  hipStream_t s = as_stream(stream);
  int dev = 0;
  GPC_HIP_CHECK(hipGetDevice(&dev));
  hipDeviceProp_t p;
  GPC_HIP_CHECK(hipGetDeviceProperties(&p, dev));
  void* ws = nullptr;
  GPC_CHECK(workspace(WS_REDUCE, 256, &ws));
  const int blocks = p.multiProcessorCount * 2, iters = 20000;
  hipEvent_t a, b;
  GPC_HIP_CHECK(hipEventCreate(&a));
  GPC_HIP_CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(mfma_f64_probe_kernel, dim3(blocks), dim3(256), 0, s, static_cast<double*>(ws), 200);
  GPC_HIP_CHECK(hipEventRecord(a, s));
  hipLaunchKernelGGL(mfma_f64_probe_kernel, dim3(blocks), dim3(256), 0, s, static_cast<double*>(ws), iters);
  GPC_HIP_CHECK(hipEventRecord(b, s));
  GPC_HIP_CHECK(hipEventSynchronize(b));
  float ms = 0.f;
  GPC_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  const double flops = (double)blocks * 4.0 /*waves*/ * (double)iters * 4.0 /*mfma*/ * 2048.0;
  *tflops = flops / ((double)ms * 1e-3) * 1e-12;
  return GPC_OK;
}
