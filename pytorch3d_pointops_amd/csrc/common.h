// common.h -- shared host/device helpers for libpointops_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pointops_amd.h"

// Parity rule: every distance is an UNFUSED fp32
// multiply followed by an add.  hipcc's default -ffp-contract=fast would fuse
// them into v_fma/v_fmac and change the last bit, hence this pragma in every
// translation unit (the build also passes -ffp-contract=off).
#pragma clang fp contract(off)

namespace pointops {

constexpr int kWave = 64;  // CDNA wavefront width

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return POINTOPS_ELAUNCH;
  }
  return POINTOPS_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace pointops

#define POINTOPS_REQUIRE(cond, ...)        \
  do {                                     \
    if (!(cond)) {                         \
      pointops::set_error(__VA_ARGS__);    \
      return POINTOPS_EINVAL;              \
    }                                      \
  } while (0)
