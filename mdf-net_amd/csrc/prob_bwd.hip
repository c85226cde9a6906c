// Backward of the regulariser's `prob` head: Conv3d(C->1,k3,p1,no bias) -> softmax over D -> soft-argmin
// (net/unit/regular.py:43,69 / :110,133, net/unit/regress.py:5-7).  Both kernels are HBM-bound streams.
//
//   softmax_regress_bwd   dlogit[d] = prob[d] * (g[d] - sum_k prob[k]*g[k]),  g[d] = ddepth*hypos[d] (+ dprob[d])
//   prob_conv_dgrad       dx[o][c] = sum_tap dlogit[o - tap + 1] * W[c][tap]     (1 -> C channels, NDHWC out)
// The weight gradient of the head is the generic correlation of wgrad.hip with A = 1.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void softmax_regress_bwd_kernel(const float* __restrict__ prob, const float* __restrict__ hypos,
                                                                  int per_pixel, const float* __restrict__ ddepth,
                                                                  const float* __restrict__ dprob, float* __restrict__ dlogit, int B,
                                                                  int D, int hw) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * hw) return;
  const int b = (int)(i / hw), pix = (int)(i % hw);
  const float dd = ddepth ? ddepth[i] : 0.0f;
  const float* pr = prob + (long long)b * D * hw + pix;
  const float* hy = per_pixel ? hypos + (long long)b * D * hw + pix : hypos + (long long)b * D;
  const float* dp = dprob ? dprob + (long long)b * D * hw + pix : nullptr;
  float dot = 0.0f;
  for (int d = 0; d < D; ++d) {
    const float g = dd * (per_pixel ? hy[(long long)d * hw] : hy[d]) + (dp ? dp[(long long)d * hw] : 0.0f);
    dot = fmaf(pr[(long long)d * hw], g, dot);
  }
  float* o = dlogit + (long long)b * D * hw + pix;
  for (int d = 0; d < D; ++d) {
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
  const long long n = (long long)B * D * H * W;
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < n; v += (long long)gridDim.x * 256) {
    const int x = (int)(v % W);
    long long r = v / W;
    const int y = (int)(r % H); r /= H;
    const int z = (int)(r % D);
    const int b = (int)(r / D);
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0f;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int zz = z + 1 - kd;
      if (zz < 0 || zz >= D) continue;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int yy = y + 1 - kh;
        if (yy < 0 || yy >= H) continue;
        const float* row = dlogit + (((long long)b * D + zz) * H + yy) * W;
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
    float4* o = reinterpret_cast<float4*>(dx + v * C);
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
  hipLaunchKernelGGL(softmax_regress_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, prob, hypos,
                     hypos_per_pixel, ddepth, dprob, dlogit, B, D, h * w);
  return mdf::check_launch("softmax_regress_bwd_kernel");
}

extern "C" int mdf_prob_conv_dgrad(const float* dlogit, const float* w, float* dx, int B, int D, int h, int wd, int C, void* stream) {
  MDF_REQUIRE(dlogit && w && dx, "null pointer argument");
  MDF_REQUIRE(B > 0 && D > 0 && h > 0 && wd > 0, "bad shape");
  const long long n = (long long)B * D * h * wd;
  long long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  if (C == 8) hipLaunchKernelGGL(prob_conv_dgrad_kernel<8>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dlogit, w, dx, B, D, h, wd);
  else if (C == 16) hipLaunchKernelGGL(prob_conv_dgrad_kernel<16>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dlogit, w, dx, B, D, h, wd);
  else return mdf::fail(MDF_EUNSUPPORTED, "prob head backward is built for C in {8,16}, got %d", C);
  return mdf::check_launch("prob_conv_dgrad_kernel");
}
