// Per-pixel heads between the stages: soft-argmin depth regression (net/unit/regress.py:5-7),
// photometric confidence (regress.py:9-25) and the curve-fit hypothesis generator
// (net/unit/depthhypos.py:27-125,169-215).  All are HBM-bound streaming kernels: one thread per
// pixel, the D axis walked with stride h*w so every load of a wavefront is one coalesced 256-B row.
// The reference builds [B,h,w,D,3] temporaries and a batched 3x3 inverse for the gauss fit; here the
// fit is a dot product with a host-prepared row (hypotheses are shared by all pixels at that stage).
#include "common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float hyp_at(const float* __restrict__ hypos, int per_pixel, size_t b, int D, int d, size_t hw,
                                        size_t pix) {
  return per_pixel ? hypos[(b * D + d) * hw + pix] : hypos[b * D + d];
}

__global__ void depth_regress_kernel(const float* __restrict__ prob, const float* __restrict__ hypos, int per_pixel,
                                     float* __restrict__ depth, int B, int D, int hw) {
  const size_t n = (size_t)B * hw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / hw, pix = i % hw;
    mdf::CascadeSum acc;  // torch.sum(prob * hypos, 1): product tensor, then ATen's cascade sum
    for (int d = 0; d < D; ++d) acc.add(prob[(b * D + d) * hw + pix] * hyp_at(hypos, per_pixel, b, D, d, hw, pix));
    depth[i] = acc.result();
  }
}

// conf = sum prob[idx-1 .. idx+2], idx = trunc(sum_d prob_d * d)   (n=4, pad=(1,2))
__global__ void confidence_kernel(const float* __restrict__ prob, float* __restrict__ conf, int64_t* __restrict__ idx_out,
                                  int B, int D, int hw) {
  const size_t n = (size_t)B * hw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / hw, pix = i % hw;
    const float* p = prob + b * D * hw + pix;
    mdf::CascadeSum ex;
    for (int d = 0; d < D; ++d) ex.add(p[(size_t)d * hw] * (float)d);
    long long idx = (long long)ex.result();  // .long() truncates toward zero
    if (idx_out) idx_out[i] = idx;
    idx = idx < 0 ? 0 : (idx > D - 1 ? D - 1 : idx);  // torch.gather would raise outside [0,D-1]; prob sums to 1 so never hit
    float s = 0.0f;
    for (int k = (int)idx - 1; k <= (int)idx + 2; ++k) s += (k >= 0 && k < D) ? p[(size_t)k * hw] : 0.0f;
    conf[i] = s;
  }
}

// the same with the nearest-neighbour x2 upsampling of core.py:76 (F.interpolate(conf, scale_factor=2, mode="nearest")) folded in:
// every value goes to its 2x2 output pixels
template <int DT>   // D at compile time (8: the model's last stage) or 0: the plane loads leave together and the window sum reads registers
__global__ void confidence_up2_kernel(const float* __restrict__ prob, float* __restrict__ conf2, int B, int D_rt, int h, int w) {
  const int D = DT ? DT : D_rt;
  const int hw = h * w;
  const size_t n = (size_t)B * hw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / hw, pix = i % hw;
    const float* p = prob + b * D * hw + pix;
    mdf::CascadeSum ex;
    float s = 0.0f;
    if constexpr (DT != 0) {
      float pv[DT];
#pragma unroll
      for (int d = 0; d < DT; ++d) pv[d] = p[(size_t)d * hw];
#pragma unroll
      for (int d = 0; d < DT; ++d) ex.add(pv[d] * (float)d);
      long long idx = (long long)ex.result();
      idx = idx < 0 ? 0 : (idx > D - 1 ? D - 1 : idx);
      // the same four terms in the same order (k = idx-1 .. idx+2), selected from the registers
      for (int k = (int)idx - 1; k <= (int)idx + 2; ++k) {
        float v = 0.0f;
#pragma unroll
        for (int d = 0; d < DT; ++d) v = (d == k) ? pv[d] : v;
        s += v;
      }
    } else {
      for (int d = 0; d < D; ++d) ex.add(p[(size_t)d * hw] * (float)d);
      long long idx = (long long)ex.result();
      idx = idx < 0 ? 0 : (idx > D - 1 ? D - 1 : idx);
      for (int k = (int)idx - 1; k <= (int)idx + 2; ++k) s += (k >= 0 && k < D) ? p[(size_t)k * hw] : 0.0f;
    }
    const int y = (int)(pix / w), x = (int)(pix % w);
    float* o = conf2 + (b * 2 * h + 2 * y) * (size_t)(2 * w) + 2 * x;
    *reinterpret_cast<float2*>(o) = make_float2(s, s);
    *reinterpret_cast<float2*>(o + 2 * w) = make_float2(s, s);
  }
}

// Range mapping around the refinement net (refine.py:29,44), one launch each instead of two or three ATen ones; the separate
// roundings of torch's op sequence are kept: mode 0: y = (x - lo[b]) / span[b];  mode 1: y = lo[b] + x * span[b]
__global__ void range_affine_kernel(const float* __restrict__ x, const float* __restrict__ lo, const float* __restrict__ span, int mode,
                                    float* __restrict__ y, int B, size_t n) {
  const size_t total = (size_t)B * n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / n;
    const float l = lo[b], sp = span[b], v = x[i];
    y[i] = (mode == 0) ? __fdiv_rn(__fsub_rn(v, l), sp) : __fadd_rn(l, __fmul_rn(v, sp));
  }
}

// mode 1: s = |-1 / sum_d row[b,d] * ln(max(p,1e-40))|          depthhypos.py:194-212
// mode 2: s = 1 / |sum(x*y)/sum(x*x)|, x = |hyp - depth|          depthhypos.py:116-123
// DT = D at compile time (48 / 24 / 8: the model's stages) or 0: with a runtime trip count the compiler issues the plane loads a few at
// a time and a thread waits out D / 4 memory round trips in a launch that has one wave per SIMD; unrolled, they leave together.  The
// arithmetic and its order are the same.
template <int DT>
__global__ void hypos_fit_kernel(int mode, const float* __restrict__ prob, const float* __restrict__ depth,
                                 const float* __restrict__ hypos, int per_pixel, const float* __restrict__ row,
                                 float* __restrict__ s_out, int B, int D_rt, int hw) {
  const int D = DT ? DT : D_rt;
  const size_t n = (size_t)B * hw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / hw, pix = i % hw;
    const float* p = prob + b * D * hw + pix;
    if (mode == 1) {
      // b0 = row . ln(p): torch.matmul([.,3,D] @ [.,D,1]) runs ATen's naive bmm kernel (contraction*rows*cols
      // < 400): sequential k loop, separate multiply and add in fp32.  The sum cancels heavily, so mirroring
      // the order (verified bit-exact on the goldens given torch's log) matters more than accuracy here.
      float acc = 0.0f;
      if constexpr (DT != 0) {
        float pv[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) pv[d] = p[(size_t)d * hw];
#pragma unroll
        for (int d = 0; d < DT; ++d) acc = acc + row[b * D + d] * logf(fmaxf(pv[d], 1e-40f));
      } else {
        for (int d = 0; d < D; ++d) acc = acc + row[b * D + d] * logf(fmaxf(p[(size_t)d * hw], 1e-40f));
      }
      s_out[i] = fabsf(-1.0f / acc);
    } else {
      const float dep = depth[i];
      mdf::CascadeSum sxy, sxx;  // torch.sum(x*y, -1), torch.sum(x*x, -1)
      if constexpr (DT != 0) {
        float pv[DT], hv[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) { pv[d] = p[(size_t)d * hw]; hv[d] = hyp_at(hypos, per_pixel, b, D, d, hw, pix); }
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const float x = fabsf(hv[d] - dep);
          const float y = logf(fmaxf(pv[d], 1e-40f));
          sxy.add(x * y);
          sxx.add(x * x);
        }
      } else {
        for (int d = 0; d < D; ++d) {
          const float x = fabsf(hyp_at(hypos, per_pixel, b, D, d, hw, pix) - dep);
          const float y = logf(fmaxf(p[(size_t)d * hw], 1e-40f));
          sxy.add(x * y);
          sxx.add(x * x);
        }
      }
      s_out[i] = 1.0f / fabsf(sxy.result() / sxx.result());
    }
  }
}

// ATen upsample_bilinear2d, align_corners=False, scale 2: src = (o+0.5)*0.5-0.5 clamped at 0.
__device__ __forceinline__ void up2_coord(int o, int n_in, int& i0, int& i1, float& l0, float& l1) {
  float src = ((float)o + 0.5f) * 0.5f - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  i0 = (int)src;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.0f - l1;
}

__device__ __forceinline__ float up2_sample(const float* __restrict__ m, int w, int y0, int y1, int x0, int x1, float ly0,
                                            float ly1, float lx0, float lx1) {
  // rounding order of ATen's CPU upsample_bilinear2d (found by exhaustive search, bit-exact on the goldens):
  //   w_ij = ly_i*lx_j ;  out = fma(w11,v11, fma(w10,v10, fma(w00,v00, w01*v01)))
  const float v00 = m[(size_t)y0 * w + x0], v01 = m[(size_t)y0 * w + x1];
  const float v10 = m[(size_t)y1 * w + x0], v11 = m[(size_t)y1 * w + x1];
  const float w00 = ly0 * lx0, w01 = ly0 * lx1, w10 = ly1 * lx0, w11 = ly1 * lx1;
  return __fmaf_rn(w11, v11, __fmaf_rn(w10, v10, __fmaf_rn(w00, v00, w01 * v01)));
}

// depthhypos.py:49-76.  One thread per OUTPUT pixel; writes D_out hypotheses (stride Ho*Wo -> coalesced).
__global__ void hypos_from_fit_kernel(int mode, const float* __restrict__ s, const float* __restrict__ depth,
                                      const float* __restrict__ range, float log_thresh, float* __restrict__ out, int B,
                                      int D_out, int h, int w, int upsample) {
  const int Ho = upsample ? 2 * h : h, Wo = upsample ? 2 * w : w;
  const size_t ohw = (size_t)Ho * Wo, n = (size_t)B * ohw;
  float gmin = range[0], gmax = range[1];
  for (int b = 1; b < B; ++b) {
    gmin = fminf(gmin, range[2 * b]);
    gmax = fmaxf(gmax, range[2 * b + 1]);
  }
  const float cap_all = (gmax - gmin) / 2.0f;  // depthhypos.py:58
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / ohw;
    const int oy = (int)((i % ohw) / Wo), ox = (int)(i % Wo);
    const float lo = range[2 * b], hi = range[2 * b + 1];
    float sv, dv;
    if (upsample) {
      int y0, y1, x0, x1;
      float ly0, ly1, lx0, lx1;
      up2_coord(oy, h, y0, y1, ly0, ly1);
      up2_coord(ox, w, x0, x1, lx0, lx1);
      sv = up2_sample(s + b * h * w, w, y0, y1, x0, x1, ly0, ly1, lx0, lx1);
      dv = up2_sample(depth + b * h * w, w, y0, y1, x0, x1, ly0, ly1, lx0, lx1);
    } else {
      sv = s[i];
      dv = depth[i];
    }
    float res = (mode == 1) ? sqrtf((-1.0f * sv) * log_thresh) : fabsf(sv * log_thresh);  // :54-57
    res = res < 1e-6f ? 1e-6f : res;                                                      // :58 clamp keeps NaN
    res = res > cap_all ? cap_all : res;
    const float cap_b = (hi - lo) * 0.2f;                                                 // :60-61
    res = res > cap_b ? cap_b : res;
    const float step = res / (float)(D_out - 1);                                          // :64
    const float base = dv - 0.5f * res;                                                   // :65
    for (int k = 0; k < D_out; ++k) {
      float hk = base + step * (float)k;                                                  // :66-67
      float t = hk - lo;                                                                  // :71-72
      hk = lo + (t < 0.0f ? 0.0f : t);
      t = hk - hi;                                                                        // :73-74
      hk = hi + (t > 0.0f ? 0.0f : t);
      out[(b * D_out + k) * ohw + (i % ohw)] = hk;
    }
  }
}

// One thread per pixel: at 1/8 resolution (148 x 200) that is 29,600 threads -- 116 blocks of 256 on 256 CUs, i.e. 140 CUs idle and
// four waves on each of the others.  Small problems therefore go out as one-wave blocks, which the dispatcher spreads over all CUs
// (the kernels index with blockDim.x; nothing in them is block-cooperative).
inline int block_for(size_t n) { return n < (size_t)256 * 1024 ? 64 : kThreads; }
inline int grid_for(size_t n) {
  const size_t t = (size_t)block_for(n);
  size_t g = (n + t - 1) / t;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int mdf_depth_regress_fwd(const float* prob, const float* hypos, int hypos_per_pixel, float* depth, int B,
                                     int D, int h, int w, void* stream) {
  MDF_REQUIRE(prob && hypos && depth, "null pointer argument");
  MDF_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "bad shape");
  hipLaunchKernelGGL(depth_regress_kernel, dim3(grid_for((size_t)B * h * w)), dim3(block_for((size_t)B * h * w)), 0, (hipStream_t)stream,
                     prob, hypos, hypos_per_pixel, depth, B, D, h * w);
  return mdf::check_launch("depth_regress_kernel");
}

extern "C" int mdf_confidence_fwd(const float* prob, float* conf, int64_t* idx_out, int B, int D, int h, int w,
                                  void* stream) {
  MDF_REQUIRE(prob && conf, "null pointer argument");
  MDF_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "bad shape");
  hipLaunchKernelGGL(confidence_kernel, dim3(grid_for((size_t)B * h * w)), dim3(block_for((size_t)B * h * w)), 0, (hipStream_t)stream, prob,
                     conf, idx_out, B, D, h * w);
  return mdf::check_launch("confidence_kernel");
}

extern "C" int mdf_confidence_up2_fwd(const float* prob, float* conf2, int B, int D, int h, int w, void* stream) {
  MDF_REQUIRE(prob && conf2, "null pointer argument");
  MDF_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "bad shape");
  if (D == 8) hipLaunchKernelGGL(confidence_up2_kernel<8>, dim3(grid_for((size_t)B * h * w)), dim3(block_for((size_t)B * h * w)), 0, (hipStream_t)stream, prob, conf2, B, D, h, w);
  else hipLaunchKernelGGL(confidence_up2_kernel<0>, dim3(grid_for((size_t)B * h * w)), dim3(block_for((size_t)B * h * w)), 0, (hipStream_t)stream, prob, conf2, B, D, h, w);
  return mdf::check_launch("confidence_up2_kernel");
}

extern "C" int mdf_range_affine_fwd(const float* x, const float* lo, const float* span, int mode, float* y, int B, long long n, void* stream) {
  MDF_REQUIRE(x && lo && span && y, "null pointer argument");
  MDF_REQUIRE(B > 0 && n > 0 && (mode == 0 || mode == 1), "bad shape / mode");
  hipLaunchKernelGGL(range_affine_kernel, dim3(grid_for((size_t)B * n)), dim3(block_for((size_t)B * n)), 0, (hipStream_t)stream, x, lo, span, mode, y, B, (size_t)n);
  return mdf::check_launch("range_affine_kernel");
}

extern "C" int mdf_hypos_fit_fwd(int mode, const float* prob, const float* depth, const float* hypos,
                                 int hypos_per_pixel, const float* fit_row, float* s_out, int B, int D, int h, int w,
                                 void* stream) {
  MDF_REQUIRE(prob && s_out, "null pointer argument");
  MDF_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "bad shape");
  if (mode == 1) {
    MDF_REQUIRE(fit_row, "gauss1 fit needs fit_row (row 0 of (X^T X)^-1 X^T, [B,D])");
    if (hypos_per_pixel)
      return mdf::fail(MDF_EUNSUPPORTED, "gauss1 fit with per-pixel hypotheses is not built (config.py:199-201 uses it "
                                         "only after the uniform stage)");
  } else if (mode == 2) {
    MDF_REQUIRE(depth && hypos, "laplace fit needs depth and hypos");
  } else {
    return mdf::fail(MDF_EARG, "mode must be 1 (gauss1) or 2 (laplace), got %d", mode);
  }
#define MDF_HF(DT) hipLaunchKernelGGL(hypos_fit_kernel<DT>, dim3(grid_for((size_t)B * h * w)), dim3(block_for((size_t)B * h * w)), 0, (hipStream_t)stream, mode, \
                                      prob, depth, hypos, hypos_per_pixel, fit_row, s_out, B, D, h * w)
  if (D == 48) MDF_HF(48); else if (D == 24) MDF_HF(24); else if (D == 8) MDF_HF(8); else MDF_HF(0);
#undef MDF_HF
  return mdf::check_launch("hypos_fit_kernel");
}

extern "C" int mdf_hypos_from_fit_fwd(int mode, const float* s, const float* depth, const float* range, float log_thresh,
                                      float* hypos_out, int B, int D_out, int h, int w, int upsample, void* stream) {
  MDF_REQUIRE(s && depth && range && hypos_out, "null pointer argument");
  MDF_REQUIRE(B > 0 && D_out > 1 && h > 0 && w > 0, "bad shape");
  MDF_REQUIRE(mode == 1 || mode == 2, "mode must be 1 (gauss1) or 2 (laplace), got %d", mode);
  const size_t n = (size_t)B * h * w * (upsample ? 4 : 1);
  hipLaunchKernelGGL(hypos_from_fit_kernel, dim3(grid_for(n)), dim3(block_for(n)), 0, (hipStream_t)stream, mode, s, depth,
                     range, log_thresh, hypos_out, B, D_out, h, w, upsample);
  return mdf::check_launch("hypos_from_fit_kernel");
}
