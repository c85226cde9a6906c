// Shared host-side helpers for the C ABI (include/mdfnet_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "mdfnet_hip.h"

namespace mdf {

void set_error(const char* fmt, ...);

inline int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  set_error("%s", buf);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MDF_EHIP, "%s: %s", what, hipGetErrorString(e));
  return MDF_OK;
}

// Bijective XCD-aware remap (cdna_hip_programming.md T1): blocks b and b+8 share an XCD, so give
// every XCD a contiguous chunk of the tile range (neighbouring tiles share source footprints / halos).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, x = bid & 7u;
  const unsigned base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

}  // namespace mdf

#define MDF_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) return mdf::fail(MDF_EARG, __VA_ARGS__); \
  } while (0)
