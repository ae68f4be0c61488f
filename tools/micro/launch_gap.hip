// Dependent-launch overhead on one stream for kernels with different resource footprints (LDS size, scratch, wave count).
// hipcc --offload-arch=gfx950 -O3 tools/micro/launch_gap.hip -o /tmp/launch_gap && /tmp/launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_plain(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void k_lds(float* p) { extern __shared__ float sm[]; sm[threadIdx.x] = p[0]; __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = sm[1] + 1.f; }
__global__ void k_scratch(float* p, int n) {
  float a[64];
  for (int i = 0; i < 64; ++i) a[i] = p[0] + i;
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += a[(i * 7 + threadIdx.x) & 63];
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = s * 0.f + p[0] + 1.f;
}
// kernels that stay busy for a set time (wall_clock64 = 100 MHz), with the footprints of the rollout's three launches
__global__ void k_spin(float* p, long long ticks) {
  extern __shared__ float sm[];
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += sm[0] * 0.f + 1.f;
}
// copy kernel with the stack kernel's traffic (n floats in, n floats out), grid-stride
__global__ void k_copy(const float* __restrict__ a, float* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n / 4; i += (size_t)gridDim.x * blockDim.x)
    reinterpret_cast<float4*>(b)[i] = reinterpret_cast<const float4*>(a)[i];
}
// busy kernel whose arguments are a large by-value struct (the env-step kernel takes ~800 B of pointers that way)
struct BigArgs { float* p[100]; long long ticks; };
__global__ void k_spin_big(BigArgs a) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < a.ticks) { __builtin_amdgcn_s_sleep(8); }
  if (threadIdx.x == 0 && blockIdx.x == 0) a.p[blockIdx.x % 100][0] += 1.f;
}
template <class F> static float run(const char* name, int iters, hipStream_t st, F launch) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch();
  (void)hipStreamSynchronize(st);
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) launch();
  (void)hipEventRecord(e1, st);
  (void)hipStreamSynchronize(st);
  float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %7.2f us per launch\n", name, 1e3f * ms / iters);
  return ms;
}
int main() {
  float* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipStream_t st2; CK(hipStreamCreateWithPriority(&st2, hipStreamNonBlocking, 0));
  CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  const int N = 3000;
  run("plain 256x256", N, st, [&] { hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st, d); });
  run("plain 512x64", N, st, [&] { hipLaunchKernelGGL(k_plain, dim3(512), dim3(64), 0, st, d); });
  run("lds 37 KB dynamic, 512x64", N, st, [&] { hipLaunchKernelGGL(k_lds, dim3(512), dim3(64), 37 * 1024, st, d); });
  run("lds 128 KB dynamic, 256x512", N, st, [&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 128 * 1024, st, d); });
  run("scratch 256 B/lane, 512x64", N, st, [&] { hipLaunchKernelGGL(k_scratch, dim3(512), dim3(64), 0, st, d, 64); });
  run("alternate: lds128 -> scratch -> plain(4096x256)", N, st, [&] {
    hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 128 * 1024, st, d);
    hipLaunchKernelGGL(k_scratch, dim3(512), dim3(64), 0, st, d, 64);
    hipLaunchKernelGGL(k_plain, dim3(4096), dim3(256), 0, st, d); });
  run("alternate: lds128 -> lds37 -> plain(4096x256)", N, st, [&] {
    hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 128 * 1024, st, d);
    hipLaunchKernelGGL(k_lds, dim3(512), dim3(64), 37 * 1024, st, d);
    hipLaunchKernelGGL(k_plain, dim3(4096), dim3(256), 0, st, d); });
  run("alternate: plain -> plain -> plain", N, st, [&] {
    hipLaunchKernelGGL(k_plain, dim3(256), dim3(512), 0, st, d);
    hipLaunchKernelGGL(k_plain, dim3(512), dim3(64), 0, st, d);
    hipLaunchKernelGGL(k_plain, dim3(4096), dim3(256), 0, st, d); });
  CK(hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  // 100 MHz ticks: 47 us, 188 us, 16 us = the undisturbed durations of actor, env step, stack kernel (sum 251 us)
  run("busy chain: actor-like 47us -> env-like 188us -> stack-like 16us  (per 3 launches)", 300, st, [&] {
    hipLaunchKernelGGL(k_spin, dim3(256), dim3(512), 128 * 1024, st, d, 4700LL);
    hipLaunchKernelGGL(k_spin, dim3(512), dim3(64), 37 * 1024, st, d, 18800LL);
    hipLaunchKernelGGL(k_spin, dim3(4096), dim3(256), 0, st, d, 1600LL); });
  run("busy chain, all three as 256x256 without LDS                        (per 3 launches)", 300, st, [&] {
    hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, st, d, 4700LL);
    hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, st, d, 18800LL);
    hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, st, d, 1600LL); });
  run("one busy kernel 251us 256x256                                       (per launch)", 300, st, [&] {
    hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, st, d, 25100LL); });
  // the same busy chain with realistic grids (stack-like kernel as 1024 x 256), stream launches vs one captured graph of 20 steps
  auto chain = [&](hipStream_t q) {
    hipLaunchKernelGGL(k_spin, dim3(256), dim3(512), 100 * 1024, q, d, 4700LL);
    hipLaunchKernelGGL(k_spin, dim3(512), dim3(64), 37 * 1024, q, d, 18800LL);
    hipLaunchKernelGGL(k_spin, dim3(1024), dim3(256), 0, q, d, 1600LL); };
  run("busy chain 47 + 188 + 16 us, stream launches                  (per 3 launches)", 300, st, [&] { chain(st); });
  {
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 20; ++i) chain(st);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    const float ms = run("busy chain x 20 as ONE hipGraph launch                       (per graph)", 15, st, [&] { (void)hipGraphLaunch(exec, st); });
    printf("   -> %.2f us per 3-kernel step inside the graph\n", 1e3f * ms / 15 / 20);
  }
  {
    // does the data a kernel writes cost the NEXT launch boundary?  (8 XCDs with private L2s: written lines are written back
    // at the end of a kernel and the next kernel starts on cold L2s)
    const size_t n = (size_t)4096 * (616 + 1052);          // the stack kernel's rows: 27 MB
    float *a, *b; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMemset(a, 0, n * 4));
    const float t_copy = run("copy 27 MB -> 27 MB alone, 4096 x 256                        (per launch)", 300, st, [&] { hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, st, a, b, n); });
    const float t_chain = run("busy 47 us -> busy 188 us -> that copy                      (per 3 launches)", 300, st, [&] {
      hipLaunchKernelGGL(k_spin, dim3(256), dim3(512), 100 * 1024, st, d, 4700LL);
      hipLaunchKernelGGL(k_spin, dim3(512), dim3(64), 37 * 1024, st, d, 18800LL);
      hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, st, a, b, n); });
    printf("   -> chain minus (235 us busy + copy alone) = %.2f us\n", 1e3f * (t_chain / 300) - 235.f - 1e3f * (t_copy / 300));
  }
  {
    BigArgs ba; for (int i = 0; i < 100; ++i) ba.p[i] = d; ba.ticks = 4700LL;
    BigArgs bb = ba; bb.ticks = 18800LL; BigArgs bc = ba; bc.ticks = 1600LL;
    run("busy chain 47 + 188 + 16 us, 808-byte by-value arguments      (per 3 launches)", 300, st, [&] {
      hipLaunchKernelGGL(k_spin_big, dim3(256), dim3(512), 0, st, ba);
      hipLaunchKernelGGL(k_spin_big, dim3(512), dim3(64), 0, st, bb);
      hipLaunchKernelGGL(k_spin_big, dim3(1024), dim3(256), 0, st, bc); });
  }
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  run("plain + event record (no timing) per launch", N, st, [&] { hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st, d); (void)hipEventRecord(ev, st); });
  hipEvent_t evt; CK(hipEventCreate(&evt));
  run("plain + event record (timing) per launch", N, st, [&] { hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st, d); (void)hipEventRecord(evt, st); });
  run("plain + record + other stream waits on it", N, st, [&] { hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st, d); (void)hipEventRecord(ev, st); (void)hipStreamWaitEvent(st2, ev, 0); });
  CK(hipStreamSynchronize(st2));
  return 0;
}
