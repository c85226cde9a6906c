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

// v2: one block = a 16x16 pixel tile walking the depth axis.  Each input plane tile (18x18xCIN, halo zero-filled) is
// staged in LDS once ([cin/4][voxel] float4 layout: neighbouring pixels read neighbouring 16-B slots, conflict-free)
// and every thread forms THREE partial sums from its 3x3 neighbourhood -- the kd = 0,1,2 slices of the kernel --
// so logit[d] = P0[d-1] + P1[d] + P2[d+1] needs each input voxel once instead of 27 times.  Double-buffered planes:
// the next plane's global loads are issued before the current plane's FMAs, one barrier per plane.
template <int CIN>
__global__ __launch_bounds__(256) void prob_head_tiled_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                              const float* __restrict__ hypos, int per_pixel,
                                                              float* __restrict__ prob, float* __restrict__ depth, int B, int D,
                                                              int h, int w, int tiles_w, int tiles_h) {
  constexpr int T = 16, P = T + 2, NV = P * P, C4 = CIN / 4;
  constexpr int NF = (C4 * NV + 255) / 256;
  __shared__ __attribute__((aligned(16))) float4 plane[2][C4 * NV];
  __shared__ float ws[27 * CIN];  // [kd][kh][kw][cin]
  const int tid = threadIdx.x;
  for (int i = tid; i < 27 * CIN; i += 256) ws[i] = wt[(i % CIN) * 27 + (i / CIN)];
  int tile = blockIdx.x;
  const int tw = tile % tiles_w; tile /= tiles_w;
  const int th = tile % tiles_h;
  const int b = tile / tiles_h;
  const int ty = tid / T, tx = tid % T;
  const int y = th * T + ty, xx = tw * T + tx;
  const bool live = (y < h) && (xx < w);
  const size_t hw = (size_t)h * w;
  const int pix = live ? y * w + xx : 0;
  float* pr = prob + (size_t)b * D * hw + pix;

  auto fetch = [&](int idx, int dz) -> float4 {
    const int c4 = idx / NV, v = idx - c4 * NV;
    const int row = v / P, col = v - row * P;
    const int iy = th * T - 1 + row, ix = tw * T - 1 + col;
    if (idx >= C4 * NV || dz >= D || iy < 0 || iy >= h || ix < 0 || ix >= w) return make_float4(0.f, 0.f, 0.f, 0.f);
    return *reinterpret_cast<const float4*>(x + ((((size_t)b * D + dz) * h + iy) * w + ix) * CIN + c4 * 4);
  };
#pragma unroll
  for (int k = 0; k < NF; ++k) {
    const int idx = tid + k * 256;
    const float4 v = fetch(idx, 0);
    if (idx < C4 * NV) plane[0][idx] = v;
  }
  __syncthreads();

  float acc_prev = 0.f, acc_cur = 0.f;  // logit[dz-1] (awaiting the kd=2 term), logit[dz] (awaiting kd=1,2 terms)
  float mx = -INFINITY;
  for (int dz = 0; dz < D; ++dz) {
    const int slot = dz & 1;
    float4 pf[NF];
    if (dz + 1 < D) {
#pragma unroll
      for (int k = 0; k < NF; ++k) pf[k] = fetch(tid + k * 256, dz + 1);
    }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int v = (ty + kh) * P + tx + kw;
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4) {
          const float4 xv = plane[slot][c4 * NV + v];
          const float* w0 = ws + ((0 * 3 + kh) * 3 + kw) * CIN + c4 * 4;
          const float* w1 = ws + ((1 * 3 + kh) * 3 + kw) * CIN + c4 * 4;
          const float* w2 = ws + ((2 * 3 + kh) * 3 + kw) * CIN + c4 * 4;
          a0 += xv.x * w0[0] + xv.y * w0[1] + xv.z * w0[2] + xv.w * w0[3];
          a1 += xv.x * w1[0] + xv.y * w1[1] + xv.z * w1[2] + xv.w * w1[3];
          a2 += xv.x * w2[0] + xv.y * w2[1] + xv.z * w2[2] + xv.w * w2[3];
        }
      }
    // input plane dz feeds logit[dz+1] (kd=0), logit[dz] (kd=1), logit[dz-1] (kd=2)
    const float done = acc_prev + a2;  // logit[dz-1] complete
    if (live && dz >= 1) { pr[(size_t)(dz - 1) * hw] = done; mx = fmaxf(mx, done); }
    acc_prev = acc_cur + a1;
    acc_cur = a0;
    if (dz + 1 < D) {
#pragma unroll
      for (int k = 0; k < NF; ++k)
        if (tid + k * 256 < C4 * NV) plane[slot ^ 1][tid + k * 256] = pf[k];
    }
    __syncthreads();
  }
  if (!live) return;
  pr[(size_t)(D - 1) * hw] = acc_prev;
  mx = fmaxf(mx, acc_prev);
  float sum = 0.f;
  for (int d = 0; d < D; ++d) {
    const float e = expf(pr[(size_t)d * hw] - mx);
    pr[(size_t)d * hw] = e;
    sum += e;
  }
  mdf::CascadeSum dep;
  for (int d = 0; d < D; ++d) {
    const float pv = pr[(size_t)d * hw] / sum;
    pr[(size_t)d * hw] = pv;
    if (depth) dep.add(pv * (per_pixel ? hypos[((size_t)b * D + d) * hw + pix] : hypos[(size_t)b * D + d]));
  }
  if (depth) depth[(size_t)b * hw + pix] = dep.result();
}

// ---- partial-sum form -----------------------------------------------------------------------------------------
// logit[d] = P0[d-1] + P1[d] + P2[d+1] where P_kd[z] is the 2-D (kh,kw,cin) contraction of input plane z with the kd-th
// slice of the 3x3x3 kernel.  The three slices are the output channels of an ordinary 2-D MFMA conv over the B*D planes
// (conv_lds_kernel<Cin,Cin,4,1,3,1,4>, 4th channel zero), so every input voxel is read once and the 27*Cin-deep
// contraction runs on the matrix cores; this kernel is the light rest: combine, softmax over D, soft-argmin.
// One thread per pixel; the D logits stay in registers (DMAX = 8/24/48), each partial is read once, prob written once.
// FULL: D == DMAX is known to the compiler -- no `d < D` guard between the plane loads, so they are issued together (regress.hip:
// hypos_fit_kernel has the measurement).
template <int DMAX, bool FULL = false>
__global__ __launch_bounds__(256) void prob_from_partials_kernel(const float4* __restrict__ part, const float* __restrict__ hypos,
                                                                 int per_pixel, float* __restrict__ prob, float* __restrict__ depth,
                                                                 int B, int D_rt, int h, int w) {
  const int D = FULL ? DMAX : D_rt;
  const size_t hw = (size_t)h * w, n = (size_t)B * hw;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = (int)(i / hw);
  const size_t pix = i % hw;
  const float4* pp = part + (size_t)b * D * hw + pix;
  float lg[DMAX];
  float carry1 = 0.f, carry0 = 0.f;   // P0[d-1] + P1[d] (awaiting P2[d+1]);  P0[d] (feeds logit[d+1])
  float mx = -INFINITY;
#pragma unroll
  for (int d = 0; d < DMAX; ++d) {
    if (d < D) {
      const float4 v = pp[(size_t)d * hw];
      if (d >= 1) { lg[d - 1] = carry1 + v.z; mx = fmaxf(mx, lg[d - 1]); }
      carry1 = carry0 + v.y;
      carry0 = v.x;
      if (d == D - 1) { lg[d] = carry1; mx = fmaxf(mx, lg[d]); }
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int d = 0; d < DMAX; ++d)
    if (d < D) { lg[d] = expf(lg[d] - mx); sum += lg[d]; }
  mdf::CascadeSum dep;
  float* pr = prob + (size_t)b * D * hw + pix;
#pragma unroll
  for (int d = 0; d < DMAX; ++d)
    if (d < D) {
      const float pv = lg[d] / sum;
      pr[(size_t)d * hw] = pv;
      if (depth) dep.add(pv * (per_pixel ? hypos[((size_t)b * D + d) * hw + pix] : hypos[(size_t)b * D + d]));
    }
  if (depth) depth[(size_t)b * hw + pix] = dep.result();
}

}  // namespace

extern "C" int mdf_prob_from_partials_fwd(const float* partials, const float* hypos, int hypos_per_pixel, float* prob, float* depth,
                                          int B, int D, int h, int wd, void* stream) {
  MDF_REQUIRE(partials && prob, "null pointer argument");
  MDF_REQUIRE(depth == nullptr || hypos != nullptr, "depth output needs hypos");
  MDF_REQUIRE(B > 0 && D > 0 && h > 0 && wd > 0, "bad shape");
  MDF_REQUIRE((reinterpret_cast<uintptr_t>(partials) & 15) == 0, "partials must be 16-byte aligned");
  const size_t n = (size_t)B * h * wd;
  const unsigned bt = n < (size_t)256 * 1024 ? 64u : 256u;     // few pixels: one-wave blocks reach every CU (regress.hip:block_for)
  dim3 grid((unsigned)((n + bt - 1) / bt)), block(bt);
  const float4* pp = reinterpret_cast<const float4*>(partials);
  if (D == 48) hipLaunchKernelGGL((prob_from_partials_kernel<48, true>), grid, block, 0, (hipStream_t)stream, pp, hypos, hypos_per_pixel, prob, depth, B, D, h, wd);
  else if (D == 24) hipLaunchKernelGGL((prob_from_partials_kernel<24, true>), grid, block, 0, (hipStream_t)stream, pp, hypos, hypos_per_pixel, prob, depth, B, D, h, wd);
  else if (D == 8) hipLaunchKernelGGL((prob_from_partials_kernel<8, true>), grid, block, 0, (hipStream_t)stream, pp, hypos, hypos_per_pixel, prob, depth, B, D, h, wd);
  else if (D <= 8) hipLaunchKernelGGL((prob_from_partials_kernel<8>), grid, block, 0, (hipStream_t)stream, pp, hypos, hypos_per_pixel, prob, depth, B, D, h, wd);
  else if (D <= 24) hipLaunchKernelGGL((prob_from_partials_kernel<24>), grid, block, 0, (hipStream_t)stream, pp, hypos, hypos_per_pixel, prob, depth, B, D, h, wd);
  else if (D <= 48) hipLaunchKernelGGL((prob_from_partials_kernel<48>), grid, block, 0, (hipStream_t)stream, pp, hypos, hypos_per_pixel, prob, depth, B, D, h, wd);
  else if (D <= 96) hipLaunchKernelGGL((prob_from_partials_kernel<96>), grid, block, 0, (hipStream_t)stream, pp, hypos, hypos_per_pixel, prob, depth, B, D, h, wd);
  else return mdf::fail(MDF_EUNSUPPORTED, "partial-sum prob head is built for D <= 96, got %d (use mdf_prob_softmax_regress_fwd)", D);
  return mdf::check_launch("prob_from_partials_kernel");
}

extern "C" int mdf_prob_softmax_regress_fwd(const float* x, const float* w, const float* hypos, int hypos_per_pixel,
                                            float* prob, float* depth, int B, int D, int h, int wd, int Cin,
                                            void* stream) {
  MDF_REQUIRE(x && w && prob, "null pointer argument");
  MDF_REQUIRE(depth == nullptr || hypos != nullptr, "depth output needs hypos");
  MDF_REQUIRE(B > 0 && D > 0 && h > 0 && wd > 0, "bad shape");
  {
    const int tiles_w = (wd + 15) / 16, tiles_h = (h + 15) / 16;
    dim3 grid((unsigned)((size_t)B * tiles_w * tiles_h)), block(256);
    switch (Cin) {
      case 8: hipLaunchKernelGGL((prob_head_tiled_kernel<8>), grid, block, 0, (hipStream_t)stream, x, w, hypos, hypos_per_pixel, prob, depth, B, D, h, wd, tiles_w, tiles_h); return mdf::check_launch("prob_head_tiled_kernel");
      case 16: hipLaunchKernelGGL((prob_head_tiled_kernel<16>), grid, block, 0, (hipStream_t)stream, x, w, hypos, hypos_per_pixel, prob, depth, B, D, h, wd, tiles_w, tiles_h); return mdf::check_launch("prob_head_tiled_kernel");
      default: break;
    }
  }
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
