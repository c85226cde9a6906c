// Adjoint of the FPN's bilinear x2 upsampling (net/unit/backbone.py:60,62: F.interpolate(scale_factor=2, mode="bilinear",
// align_corners=False)), NHWC: the backward of `up(t)` in the top-down path, d t[j] = sum_i d fine[i] * w(i, j).
// Forward weights (per axis): fine[2j] = 0.25*c[j-1] + 0.75*c[j], fine[2j+1] = 0.75*c[j] + 0.25*c[j+1], indices clamped at
// the border -- so coarse sample j collects 0.25, 0.75, 0.75, 0.25 from fine 2j-1 .. 2j+2, and the clamped border taps fold
// back onto j = 0 and j = n-1.  One thread per (coarse pixel, 4 channels); an HBM stream (reads the fine map once).
#include "common.h"

namespace {

__device__ __forceinline__ void axis_taps(int j, int n, int* idx, float* wt) {
  // fine indices 2j-1, 2j, 2j+1, 2j+2 with weights 0.25, 0.75, 0.75, 0.25; outside [0, 2n): dropped, except that the
  // border-clamped forward taps land on j itself: fine[0] uses c[0] with weight 1 (= 0.75 + 0.25), fine[2n-1] likewise
  idx[0] = 2 * j - 1; idx[1] = 2 * j; idx[2] = 2 * j + 1; idx[3] = 2 * j + 2;
  wt[0] = 0.25f; wt[1] = 0.75f; wt[2] = 0.75f; wt[3] = 0.25f;
  if (j == 0) { wt[0] = 0.0f; idx[0] = 0; wt[1] = 1.0f; }
  if (j == n - 1) { wt[3] = 0.0f; idx[3] = 2 * n - 1; wt[2] = 1.0f; }
}

__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const float* __restrict__ dfine, float* __restrict__ dcoarse, int B, int h,
                                                            int w, int C4, int accumulate) {
  const long long n = (long long)B * h * w * C4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    long long r = i / C4;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h);
    const int b = (int)(r / h);
    int iy[4], ix[4];
    float wy[4], wx[4];
    axis_taps(y, h, iy, wy);
    axis_taps(x, w, ix, wx);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* src = reinterpret_cast<const float4*>(dfine) + (long long)b * (2 * h) * (2 * w) * C4 + c4;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (wy[a] == 0.0f) continue;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (wx[k] == 0.0f) continue;
        const float4 v = src[((long long)iy[a] * (2 * w) + ix[k]) * C4];
        const float ww = wy[a] * wx[k];
        acc.x = fmaf(ww, v.x, acc.x); acc.y = fmaf(ww, v.y, acc.y); acc.z = fmaf(ww, v.z, acc.z); acc.w = fmaf(ww, v.w, acc.w);
      }
    }
    float4* dst = reinterpret_cast<float4*>(dcoarse) + i;
    if (accumulate) {
      const float4 o = *dst;
      acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    *dst = acc;
  }
}

}  // namespace

extern "C" int mdf_upsample2_bilinear_bwd(const float* dfine, float* dcoarse, int B, int h, int w, int C, int accumulate, void* stream) {
  MDF_REQUIRE(dfine && dcoarse, "null pointer argument");
  MDF_REQUIRE(B > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0, "bad shape (C must be a multiple of 4)");
  const long long n = (long long)B * h * w * (C / 4);
  long long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(upsample2_bwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dfine, dcoarse, B, h, w, C / 4, accumulate);
  return mdf::check_launch("upsample2_bwd_kernel");
}
