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
