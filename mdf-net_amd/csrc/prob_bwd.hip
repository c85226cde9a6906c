// Backward of the regulariser's `prob` head: Conv3d(C->1,k3,p1,no bias) -> softmax over D -> soft-argmin
// (net/unit/regular.py:43,69 / :110,133, net/unit/regress.py:5-7).  Both kernels are HBM-bound streams.
//
//   softmax_regress_bwd   dlogit[d] = prob[d] * (g[d] - sum_k prob[k]*g[k]),  g[d] = ddepth*hypos[d] (+ dprob[d])
//   prob_conv_dgrad       dx[o][c] = sum_tap dlogit[o - tap + 1] * W[c][tap]     (1 -> C channels, NDHWC out)
// The weight gradient of the head is the generic correlation of wgrad.hip with A = 1.
#include "common.h"

namespace {

// S threads share a pixel (plane d belongs to thread d % S): at 1/8 resolution a thread per pixel is 6912 threads walking 48 planes
// twice -- 27 blocks on 256 CUs.  tid = s * (256 / S) + local pixel, so a wave's lanes still read consecutive pixels of a plane.
template <int S>
__global__ __launch_bounds__(256) void softmax_regress_bwd_kernel(const float* __restrict__ prob, const float* __restrict__ hypos,
                                                                  int per_pixel, const float* __restrict__ ddepth,
                                                                  const float* __restrict__ dprob, float* __restrict__ dlogit, int B,
                                                                  int D, int hw) {
  constexpr int PPB = 256 / S;
  __shared__ float part[S][PPB];
  const int s = threadIdx.x / PPB, pl = threadIdx.x % PPB;
  const long long i = (long long)blockIdx.x * PPB + pl;
  const bool live = i < (long long)B * hw;
  const long long ii = live ? i : 0;
  const int b = (int)(ii / hw), pix = (int)(ii % hw);
  const float dd = ddepth ? ddepth[ii] : 0.0f;
  const float* pr = prob + (long long)b * D * hw + pix;
  const float* hy = per_pixel ? hypos + (long long)b * D * hw + pix : hypos + (long long)b * D;
  const float* dp = dprob ? dprob + (long long)b * D * hw + pix : nullptr;
  float dot = 0.0f;
  for (int d = s; d < D; d += S) {
    const float g = dd * (per_pixel ? hy[(long long)d * hw] : hy[d]) + (dp ? dp[(long long)d * hw] : 0.0f);
    dot = fmaf(pr[(long long)d * hw], g, dot);
  }
  if constexpr (S > 1) {
    part[s][pl] = dot;
    __syncthreads();
    dot = 0.0f;
#pragma unroll
    for (int k = 0; k < S; ++k) dot += part[k][pl];      // the same order in every thread of the pixel
  }
  if (!live) return;
  float* o = dlogit + (long long)b * D * hw + pix;
  for (int d = s; d < D; d += S) {
    const float g = dd * (per_pixel ? hy[(long long)d * hw] : hy[d]) + (dp ? dp[(long long)d * hw] : 0.0f);
    o[(long long)d * hw] = pr[(long long)d * hw] * (g - dot);
  }
}

template <int C>
__global__ __launch_bounds__(256) void prob_conv_dgrad_kernel(const float* __restrict__ dlogit, const float* __restrict__ w,
                                                              float* __restrict__ dx, int B, int D, int H, int W) {
  __shared__ float wt[27 * C];   // [tap][c], tap order of the FORWARD kernel (kd,kh,kw)
  for (int i = threadIdx.x; i < 27 * C; i += 256) {
    const int c = i % C, tap = i / C;
    wt[i] = w[c * 27 + tap];     // torch [1,C,3,3,3]
  }
  __syncthreads();
  const int n = B * D * H * W;            // (< 2^31: checked by the entry point; 64-bit divisions cost more than the taps)
  for (int v = blockIdx.x * 256 + threadIdx.x; v < n; v += gridDim.x * 256) {
    const int x = v % W;
    int r = v / W;
    const int y = r % H; r /= H;
    const int z = r % D;
    const int b = r / D;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0f;
    // the (kd, kh) loops stay rolled: fully unrolled, the compiler keeps all 27 x C weights in registers (246 VGPRs for C = 8, 256 + 220
    // AGPRs for C = 16: one or two waves per SIMD, and the kernel ran at 0.6 TB/s); three taps' weights at a time come from LDS as broadcasts
#pragma unroll 1
    for (int kd = 0; kd < 3; ++kd) {
      const int zz = z + 1 - kd;
      if (zz < 0 || zz >= D) continue;
#pragma unroll 1
      for (int kh = 0; kh < 3; ++kh) {
        const int yy = y + 1 - kh;
        if (yy < 0 || yy >= H) continue;
        const float* row = dlogit + ((b * D + zz) * H + yy) * W;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int xx = x + 1 - kw;
          if (xx < 0 || xx >= W) continue;
          const float g = row[xx];
          const float* wp = wt + ((kd * 3 + kh) * 3 + kw) * C;
#pragma unroll
          for (int c = 0; c < C; ++c) acc[c] = fmaf(g, wp[c], acc[c]);
        }
      }
    }
    float4* o = reinterpret_cast<float4*>(dx + (size_t)v * C);
#pragma unroll
    for (int c = 0; c < C; c += 4) o[c / 4] = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
  }
}

}  // namespace

extern "C" int mdf_prob_softmax_regress_bwd(const float* prob, const float* hypos, int hypos_per_pixel, const float* ddepth,
                                            const float* dprob, float* dlogit, int B, int D, int h, int w, void* stream) {
  MDF_REQUIRE(prob && dlogit && (ddepth || dprob), "null pointer argument");
  MDF_REQUIRE(!ddepth || hypos, "ddepth needs the hypotheses");
  MDF_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "bad shape");
  const long long n = (long long)B * h * w;
  // threads per pixel: enough blocks for the chip at the coarse stages (pixels x S >= ~256k threads), never more than the planes
  int S = 1;
  while (S < 8 && n * S < 262144 && 2 * S <= D) S *= 2;
#define MDF_SRB(SS) hipLaunchKernelGGL(softmax_regress_bwd_kernel<SS>, dim3((unsigned)((n + 256 / SS - 1) / (256 / SS))), dim3(256), 0, (hipStream_t)stream, \
                                       prob, hypos, hypos_per_pixel, ddepth, dprob, dlogit, B, D, h * w)
  if (S == 8) MDF_SRB(8); else if (S == 4) MDF_SRB(4); else if (S == 2) MDF_SRB(2); else MDF_SRB(1);
#undef MDF_SRB
  return mdf::check_launch("softmax_regress_bwd_kernel");
}

extern "C" int mdf_prob_conv_dgrad(const float* dlogit, const float* w, float* dx, int B, int D, int h, int wd, int C, void* stream) {
  MDF_REQUIRE(dlogit && w && dx, "null pointer argument");
  MDF_REQUIRE(B > 0 && D > 0 && h > 0 && wd > 0, "bad shape");
  const long long n = (long long)B * D * h * wd;
  MDF_REQUIRE(n < (1ll << 31), "volume must have fewer than 2^31 voxels");
  long long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  if (C == 8) hipLaunchKernelGGL(prob_conv_dgrad_kernel<8>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dlogit, w, dx, B, D, h, wd);
  else if (C == 16) hipLaunchKernelGGL(prob_conv_dgrad_kernel<16>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dlogit, w, dx, B, D, h, wd);
  else return mdf::fail(MDF_EUNSUPPORTED, "prob head backward is built for C in {8,16}, got %d", C);
  return mdf::check_launch("prob_conv_dgrad_kernel");
}
