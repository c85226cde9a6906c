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
  // dlogit[d] = p[d] * (dd * (h[d] - M) + (dp[d] - sum_k p[k] dp[k])),  M = sum_k p[k] h[k], with the bracket (h[d] - M) formed
  // WITHOUT cancelling two products of the size of the depth: m = fl(sum p h) first (good to eps * depth), then the correction
  // c = sum_k p[k] (h[k] - m), whose terms are small wherever p is large, and h[d] - M = (h[d] - m) - c.  The straight form
  // `p * (dd * h[d] - dot)` of autograd's softmax backward carries eps * depth / |h[d] - M| of relative error at the peak of a peaked
  // volume, which the float64 yardstick of tests/test_train_gpu.py then finds in every gradient upstream: with this form the
  // regulariser's parameter gradients are 4-16 x closer to float64 than torch's fp32 CPU autograd (r04: median ratio 0.06-0.26).
  auto hyp = [&](int d) { return per_pixel ? hy[(long long)d * hw] : hy[d]; };
  auto reduce = [&](float v) {
    if constexpr (S > 1) {
      __syncthreads();                                     // (the previous round's reads of `part` are over)
      part[s][pl] = v;
      __syncthreads();
      v = 0.0f;
#pragma unroll
      for (int k = 0; k < S; ++k) v += part[k][pl];        // the same order in every thread of the pixel
    }
    return v;
  };
  float m = 0.0f, mp = 0.0f;
  for (int d = s; d < D; d += S) {
    const float p = pr[(long long)d * hw];
    m = fmaf(p, hyp(d), m);
    if (dp) mp = fmaf(p, dp[(long long)d * hw], mp);
  }
  m = reduce(m);
  if (dp) mp = reduce(mp);                                 // (dp is uniform over the block: every thread takes the same path)
  float c = 0.0f;
  for (int d = s; d < D; d += S) c = fmaf(pr[(long long)d * hw], hyp(d) - m, c);
  c = reduce(c);
  if (!live) return;
  float* o = dlogit + (long long)b * D * hw + pix;
  for (int d = s; d < D; d += S) {
    const float g = dd * ((hyp(d) - m) - c) + (dp ? dp[(long long)d * hw] - mp : 0.0f);
    o[(long long)d * hw] = pr[(long long)d * hw] * g;
  }
}

template <int CTRL>
__device__ __forceinline__ float dpp_row_shr(float v) {     // lane l reads lane l - n of its 16-lane row (0 where there is none)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// STAT: dx is the COMPLETE gradient of the regulariser's last layer (nothing else consumes its output), so the sums its BatchNorm
// backward needs -- sum dr, sum dr * xhat with dr = dx where the layer's ReLU was open -- are taken here, while dx is in registers,
// instead of by a pass of their own over dx and the layer's raw output (mdf_bn_relu_bwd_reduce; same formulas, bn_train.hip).
// stat_y = that raw output [n][C], stat_aux = (a, b, mean, invstd) [4C], red = [nslices][2C] doubles, zeroed by the caller.
template <int C, bool STAT>
__global__ __launch_bounds__(256) void prob_conv_dgrad_kernel(const float* __restrict__ dlogit, const float* __restrict__ w,
                                                              float* __restrict__ dx, int B, int D, int H, int W,
                                                              const float* __restrict__ stat_y, const float* __restrict__ stat_aux,
                                                              double* __restrict__ red, int nslices) {
  __shared__ float wt[27 * C];   // [tap][c], tap order of the FORWARD kernel (kd,kh,kw)
  __shared__ float ax[STAT ? 4 * C : 1];
  __shared__ double rsum[STAT ? 16 : 1][2 * C];
  for (int i = threadIdx.x; i < 27 * C; i += 256) {
    const int c = i % C, tap = i / C;
    wt[i] = w[c * 27 + tap];     // torch [1,C,3,3,3]
  }
  if constexpr (STAT) {
    for (int i = threadIdx.x; i < 4 * C; i += 256) ax[i] = stat_aux[i];
  }
  __syncthreads();
  float ps[STAT ? C : 1], pq[STAT ? C : 1];
  if constexpr (STAT) {
#pragma unroll
    for (int c = 0; c < C; ++c) { ps[c] = 0.0f; pq[c] = 0.0f; }
  }
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
    if constexpr (STAT) {
      const float4* yp = reinterpret_cast<const float4*>(stat_y + (size_t)v * C);
#pragma unroll
      for (int c = 0; c < C; c += 4) {
        const float4 yv4 = yp[c / 4];
        const float yv[4] = {yv4.x, yv4.y, yv4.z, yv4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float dr = (fmaf(yv[k], ax[c + k], ax[C + c + k]) > 0.0f) ? acc[c + k] : 0.0f;
          ps[c + k] += dr;
          pq[c + k] = fmaf(dr, (yv[k] - ax[2 * C + c + k]) * ax[3 * C + c + k], pq[c + k]);
        }
      }
    }
  }
  if constexpr (STAT) {
    // the 16 lanes of a row add up with DPP row shifts (an inclusive scan: the row's last lane ends up with the total), the 16 rows
    // of the block through LDS, and the block sends 2C doubles to its copy of the sums
    const int tid = threadIdx.x;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float x = ps[c], y = pq[c];
      x += dpp_row_shr<0x111>(x); y += dpp_row_shr<0x111>(y);
      x += dpp_row_shr<0x112>(x); y += dpp_row_shr<0x112>(y);
      x += dpp_row_shr<0x114>(x); y += dpp_row_shr<0x114>(y);
      x += dpp_row_shr<0x118>(x); y += dpp_row_shr<0x118>(y);
      if ((tid & 15) == 15) { rsum[tid >> 4][c] = (double)x; rsum[tid >> 4][C + c] = (double)y; }
    }
    __syncthreads();
    if (tid < 2 * C) {
      double t = 0.0;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += rsum[r][tid];
      atomicAdd(&red[(size_t)(blockIdx.x % nslices) * 2 * C + tid], t);
    }
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

static int prob_conv_dgrad_impl(const float* dlogit, const float* w, float* dx, int B, int D, int h, int wd, int C, const float* stat_y,
                                const float* stat_aux, double* red, int nslices, void* stream) {
  MDF_REQUIRE(dlogit && w && dx, "null pointer argument");
  MDF_REQUIRE(B > 0 && D > 0 && h > 0 && wd > 0, "bad shape");
  const long long n = (long long)B * D * h * wd;
  MDF_REQUIRE(n < (1ll << 31), "volume must have fewer than 2^31 voxels");
  long long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  const bool stat = red != nullptr;
#define MDF_PCD(CC, ST) hipLaunchKernelGGL((prob_conv_dgrad_kernel<CC, ST>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dlogit, w, dx, B, D, h, wd, \
                                           stat_y, stat_aux, red, nslices)
  if (C == 8) { if (stat) MDF_PCD(8, true); else MDF_PCD(8, false); }
  else if (C == 16) { if (stat) MDF_PCD(16, true); else MDF_PCD(16, false); }
  else return mdf::fail(MDF_EUNSUPPORTED, "prob head backward is built for C in {8,16}, got %d", C);
#undef MDF_PCD
  return mdf::check_launch("prob_conv_dgrad_kernel");
}

extern "C" int mdf_prob_conv_dgrad(const float* dlogit, const float* w, float* dx, int B, int D, int h, int wd, int C, void* stream) {
  return prob_conv_dgrad_impl(dlogit, w, dx, B, D, h, wd, C, nullptr, nullptr, nullptr, 1, stream);
}

extern "C" int mdf_prob_conv_dgrad_stat(const float* dlogit, const float* w, float* dx, int B, int D, int h, int wd, int C, const float* stat_y,
                                        const float* stat_aux, double* red, int nslices, void* stream) {
  MDF_REQUIRE(stat_y && stat_aux && red, "null pointer argument");
  MDF_REQUIRE(nslices >= 1, "nslices must be >= 1");
  return prob_conv_dgrad_impl(dlogit, w, dx, B, D, h, wd, C, stat_y, stat_aux, red, nslices, stream);
}
