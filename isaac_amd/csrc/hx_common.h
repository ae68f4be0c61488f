// hx_common.h -- error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <string>

void hx_set_error(const std::string& s);

// Experiment knobs are environment variables read when a simulator / learner is created.  None is needed to use
// the library; all of them change scheduling only, never results.  They are VALIDATED: hx_knobs_check() fails creation on an
// HX_* variable this build does not know (a typo would otherwise silently run the default) and hx_knob_int() on a value that is
// not an integer inside the knob's range.
int hx_knobs_check(void);
int hx_knob_int(const char* name, int dflt, int lo, int hi, int* out);
int hx_knob_hex32(const char* name, bool* present, unsigned* out);

#define HX_CHECK(expr)                                                                         \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      hx_set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " at " + __FILE__ + \
                   ":" + std::to_string(__LINE__));                                            \
      return -100 - (int)_e;                                                                   \
    }                                                                                          \
  } while (0)

// ---- per-step bookkeeping that depends on the env-step launch's TOTAL reset count and therefore runs in the launch that
// follows it: the stacking kernel (row API), or the consumer of a frame-mode step (include/hx_sim.h hx_sim_take_book) -- the
// fused rollout actor while it stages its rows, or hx_book_kernel.
#if defined(__HIPCC__)
#include "../../include/hx_sim.h"
// per-env part: extras["time_outs"] is rebound only inside reset_idx, i.e. when at least one env reset this step
// (legged_robot.py:172-173,208-209); reward / done / time-out go to the learner's slot
__device__ __forceinline__ void hx_step_book_row(const hx_step_book& b, int e) {
  unsigned char tv = b.timeout_visible[e];
  if (*b.num_reset > 0) { tv = b.timeout[e]; b.timeout_visible[e] = tv; }
  if (b.rew_out) { b.rew_out[e] = b.rew[e]; b.done_out[e] = b.reset[e] ? 1 : 0; b.timeout_out[e] = tv; }
}
// once per step: recycle the other reset counter, fold the step's episode statistics (on_policy_runner.py:141-142).
// Called by the lanes 0 .. HX_NUM_REWARDS-1 (at least) of ONE wave, lane = `j`: every lane folds one reward term, so the
// chain of dependent memory operations is a few deep instead of 4 x 22 (the one-thread form was most of the stacking
// kernel's duration).  Lock-step execution of a wave orders the flag reads of all lanes before lane 0's writes.
__device__ __forceinline__ void hx_step_book_global(const hx_step_book& b, int j) {
  const int nr = *b.num_reset;
  const int had = b.stat_steps[1];
  const bool have = (nr > 0) || (had != 0);
  if (j < HX_NUM_REWARDS) {
    float last = b.stat_last[j];
    if (nr > 0) { last = b.stat_sum[j] / (float)nr; b.stat_last[j] = last; b.stat_sum[j] = 0.f; }
    if (have) b.stat_acc[j] += last;
  }
  if (j == 0) {
    *b.num_reset_next = 0;
    if (nr > 0) b.stat_steps[1] = 1;
    if (have) b.stat_steps[0] += 1;
  }
}
#endif
