// Shared host-side helpers for the C ABI (include/mdfnet_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "mdfnet_hip.h"

namespace mdf {

void set_error(const char* fmt, ...);
void note_launch(const char* kernel);     // abi.cpp: the kernel the calling thread enqueued last (mdf_last_launch)

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
  note_launch(what);
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

// ATen's CPU float sum (SumKernel.cpp multi_row_sum / cascade_sum): elements are added sequentially into
// level 0; after every 16 elements level 0 is folded into level 1 (after every 256 into level 2, ...); the
// result is ((l0 + l1) + l2) + l3.  Mirroring it makes `torch.sum(a*b, dim)`-style reductions bit-identical
// to the reference's CPU result for the same inputs (verified against the goldens for D = 8, 24, 48).
struct CascadeSum {
  float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
  unsigned n = 0;
  __device__ __forceinline__ void add(float v) {
    l0 += v;
    ++n;
    if ((n & 15u) == 0) {
      l1 += l0; l0 = 0.f;
      if ((n & 255u) == 0) {
        l2 += l1; l1 = 0.f;
        if ((n & 4095u) == 0) { l3 += l2; l2 = 0.f; }
      }
    }
  }
  __device__ __forceinline__ float result() const { return ((l0 + l1) + l2) + l3; }
};

// Epilogue sums of the training path (conv3d.hip / conv_lds.hip; all null / 0 for inference): see ConvParams::stat_mode.
struct ConvStat {
  int mode;
  const float* y;
  const float* aux;
  double* out;
  int group_imgs;      // 2-D: images per BatchNorm group (the tensor is [groups][imgs][H][W][C]); 0 = one group
  int nslices;         // the sums are spread over this many copies ("slices") of [groups][2C]; the consumers add them up
};

// Sending a block's sums to memory costs 2C fp64 atomics on the SAME 2C addresses for every block, and same-address atomics
// retire one per ~11 ns (rocprof, r03: a 2592-block launch paid 28 us for them -- more than its MFMA work).  A two-level
// scheme with per-slice counters (last arriver collapses the slice) was built and is far worse: its release fence writes the
// L2 back per block (24 -> 120 us).  So: block b adds into slice b % R of out[R][groups][2C] (contention / R, no ordering
// needed) and the BatchNorm kernels that consume the sums add the R slices up (16 x 32 B per thread, L2 hits).
inline int conv_stat_slices(long long blocks_per_group, int nslices) {
  long long r = (blocks_per_group + 47) / 48;          // <= ~48 blocks per address
  if (r > nslices) r = nslices;
  return (int)(r < 1 ? 1 : r);
}
// tab: the block's LDS table [2][64] (sum, sum2).  out: slice 0 of this block's group; consecutive slices are slice_stride apart.
template <int COUT>
__device__ __forceinline__ void conv_stat_send(const double* tab, double* out, long long slice_stride, int R, unsigned local_blk) {
  const int tid = threadIdx.x;
  if (tid < 2 * COUT) {
    const int c = tid % COUT, which = tid / COUT;
    const double v = tab[which * 64 + c];
    if (v != 0.0) atomicAdd(&out[(size_t)(local_blk % (unsigned)R) * slice_stride + which * COUT + c], v);
  }
}

}  // namespace mdf

#define MDF_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) return mdf::fail(MDF_EARG, __VA_ARGS__); \
  } while (0)
