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
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  run("plain + event record (no timing) per launch", N, st, [&] { hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st, d); (void)hipEventRecord(ev, st); });
  hipEvent_t evt; CK(hipEventCreate(&evt));
  run("plain + event record (timing) per launch", N, st, [&] { hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st, d); (void)hipEventRecord(evt, st); });
  run("plain + record + other stream waits on it", N, st, [&] { hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st, d); (void)hipEventRecord(ev, st); (void)hipStreamWaitEvent(st2, ev, 0); });
  CK(hipStreamSynchronize(st2));
  return 0;
}
