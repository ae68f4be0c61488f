// hx_common.h -- error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <string>

void hx_set_error(const std::string& s);

#define HX_CHECK(expr)                                                                         \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      hx_set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " at " + __FILE__ + \
                   ":" + std::to_string(__LINE__));                                            \
      return -100 - (int)_e;                                                                   \
    }                                                                                          \
  } while (0)

// ---- device side of the deferred frame stacking (include/hx_sim.h hx_pending_step): shared by the env-step launch
// (privileged rows, hx_sim.hip) and the fused rollout actor (observation rows + bookkeeping, hx_ppo.hip), and equal to what
// hx_stack_kernel does for the same step.
#if defined(__HIPCC__)
#include "../../include/hx_sim.h"
__device__ __forceinline__ float hx_row_stack_value(const hx_row_stack& a, int e, bool rst, int k) {
  const int keep = (a.stack - 1) * a.f;
  if (k < keep) return rst ? 0.f : a.src[(size_t)e * a.ld + k + a.f];
  if (k < keep + a.f) return fminf(fmaxf(a.frame[(size_t)(k - keep) * a.n + e], -a.clip), a.clip);
  return 0.f;
}
// The same value without divergent control flow around the load (the address is selected, the load is unconditional), so
// that a caller can keep a batch of them in flight: hipcc drains the VM counter at every guarded load otherwise.
__device__ __forceinline__ const float* hx_row_stack_addr(const hx_row_stack& a, int e, int k) {
  const int keep = (a.stack - 1) * a.f;
  const int kf = min(max(k - keep, 0), a.f - 1);
  return (k < keep) ? a.src + (size_t)e * a.ld + k + a.f : a.frame + (size_t)kf * a.n + e;
}
__device__ __forceinline__ float hx_row_stack_finish(const hx_row_stack& a, bool rst, int k, float loaded) {
  const int keep = (a.stack - 1) * a.f;
  const float hist = rst ? 0.f : loaded;
  const float fresh = fminf(fmaxf(loaded, -a.clip), a.clip);
  return (k < keep) ? hist : ((k < keep + a.f) ? fresh : 0.f);
}
// per-env part: extras["time_outs"] is rebound only inside reset_idx, i.e. when at least one env reset this step
// (legged_robot.py:172-173,208-209); reward / done / time-out go to the learner's slot
__device__ __forceinline__ void hx_step_book_row(const hx_step_book& b, int e) {
  unsigned char tv = b.timeout_visible[e];
  if (*b.num_reset > 0) { tv = b.timeout[e]; b.timeout_visible[e] = tv; }
  if (b.rew_out) { b.rew_out[e] = b.rew[e]; b.done_out[e] = b.reset[e] ? 1 : 0; b.timeout_out[e] = tv; }
}
// once per step: recycle the other reset counter, fold the step's episode statistics (on_policy_runner.py:141-142)
__device__ __forceinline__ void hx_step_book_global(const hx_step_book& b) {
  *b.num_reset_next = 0;
  const int nr = *b.num_reset;
  if (nr > 0) {
    for (int r = 0; r < HX_NUM_REWARDS; ++r) { b.stat_last[r] = b.stat_sum[r] / (float)nr; b.stat_sum[r] = 0.f; }
    b.stat_steps[1] = 1;
  }
  if (b.stat_steps[1]) {
    for (int r = 0; r < HX_NUM_REWARDS; ++r) b.stat_acc[r] += b.stat_last[r];
    b.stat_steps[0] += 1;
  }
}
#endif
