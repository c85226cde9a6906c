// `prob` head of the regularisers: Conv3d(Cin -> 1, k3, p1, no bias) + softmax over the depth axis
// (net/unit/regular.py:43,69 and :110,133) with the soft-argmin (net/unit/regress.py:5-7) fused in.
// Cout = 1 makes this layer HBM/L2-bound (12 flop/B), not a matrix-core contraction: one thread per
// pixel walks the D axis; the 27*Cin weights are wave-uniform (scalar loads); adjacent lanes are
// adjacent pixels, so the Cin-contiguous (NDHWC) taps of a wavefront form one contiguous segment.
// The logits are parked in the output buffer, then normalised in place (each thread re-reads only its
// own writes), so no [D] register array is needed and D is a runtime value.
#include "common.h"

namespace {

template <int CIN>
__global__ __launch_bounds__(256) void prob_head_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                        const float* __restrict__ hypos, int per_pixel,
                                                        float* __restrict__ prob, float* __restrict__ depth, int B, int D,
                                                        int h, int w) {
  __shared__ float ws[27 * CIN];  // [tap][cin]
  for (int i = threadIdx.x; i < 27 * CIN; i += blockDim.x) ws[i] = wt[(i % CIN) * 27 + (i / CIN)];
  __syncthreads();
  const size_t hw = (size_t)h * w, n = (size_t)B * hw;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = (int)(i / hw);
  const int pix = (int)(i % hw);
  const int y = pix / w, xx = pix - y * w;
  float* pr = prob + (size_t)b * D * hw + pix;
  float mx = -INFINITY;
  for (int d = 0; d < D; ++d) {
    float acc = 0.f;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int dz = d + kd - 1;
      if (dz < 0 || dz >= D) continue;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int yy = y + kh - 1;
        if (yy < 0 || yy >= h) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int xc = xx + kw - 1;
          if (xc < 0 || xc >= w) continue;
          const float* px = x + ((((size_t)b * D + dz) * h + yy) * w + xc) * CIN;
          const float* wk = ws + ((kd * 3 + kh) * 3 + kw) * CIN;
#pragma unroll
          for (int c = 0; c < CIN; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(px + c);
            acc += v.x * wk[c] + v.y * wk[c + 1] + v.z * wk[c + 2] + v.w * wk[c + 3];
          }
        }
      }
    }
    pr[(size_t)d * hw] = acc;
    mx = fmaxf(mx, acc);
  }
  float sum = 0.f;
  for (int d = 0; d < D; ++d) {
    const float e = expf(pr[(size_t)d * hw] - mx);
    pr[(size_t)d * hw] = e;
    sum += e;
  }
  mdf::CascadeSum dep;  // regress.py:5-7 with ATen's summation order
  for (int d = 0; d < D; ++d) {
    const float pv = pr[(size_t)d * hw] / sum;
    pr[(size_t)d * hw] = pv;
    if (depth) dep.add(pv * (per_pixel ? hypos[((size_t)b * D + d) * hw + pix] : hypos[(size_t)b * D + d]));
  }
  if (depth) depth[i] = dep.result();
}

}  // namespace

extern "C" int mdf_prob_softmax_regress_fwd(const float* x, const float* w, const float* hypos, int hypos_per_pixel,
                                            float* prob, float* depth, int B, int D, int h, int wd, int Cin,
                                            void* stream) {
  MDF_REQUIRE(x && w && prob, "null pointer argument");
  MDF_REQUIRE(depth == nullptr || hypos != nullptr, "depth output needs hypos");
  MDF_REQUIRE(B > 0 && D > 0 && h > 0 && wd > 0, "bad shape");
  const size_t n = (size_t)B * h * wd;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  switch (Cin) {
    case 8: hipLaunchKernelGGL((prob_head_kernel<8>), grid, block, 0, (hipStream_t)stream, x, w, hypos, hypos_per_pixel, prob, depth, B, D, h, wd); break;
    case 16: hipLaunchKernelGGL((prob_head_kernel<16>), grid, block, 0, (hipStream_t)stream, x, w, hypos, hypos_per_pixel, prob, depth, B, D, h, wd); break;
    case 32: hipLaunchKernelGGL((prob_head_kernel<32>), grid, block, 0, (hipStream_t)stream, x, w, hypos, hypos_per_pixel, prob, depth, B, D, h, wd); break;
    default: return mdf::fail(MDF_EUNSUPPORTED, "prob head is built for Cin in {8,16,32}, got %d", Cin);
  }
  return mdf::check_launch("prob_head_kernel");
}
