// Device helpers shared by the eval kernels (warp_aggregate.hip) and the training kernels (warp_aggregate_train.hip):
// sample-position arithmetic (bit-exact w.r.t. torch's CPU path), the bilinear tap table entry, DPP reductions over the
// lanes of a pixel, pairwise softmax.  Compile with -ffp-contract=off: every fused multiply-add below is explicit.
#pragma once
#include "common.h"

namespace {

constexpr int kThreads = 256;

// Pixels of a block's tile.  TH = 0: a run of PPB consecutive pixels in row-major order (wraps over row ends); TH > 0: a
// (PPB / TH) x TH rectangle -- the taps of vertically neighbouring pixels share source texels inside ONE block's L1 working set
// instead of across blocks, and a tile's footprint in a source map is a compact box instead of a slanted line.
#ifndef MDF_WARP_TILE_H
#define MDF_WARP_TILE_H 0
#endif
template <int PPB, int TH = MDF_WARP_TILE_H>
struct PixTile {
  static constexpr int TW = (TH > 0) ? PPB / TH : PPB;
  static_assert(TH == 0 || TW * TH == PPB, "tile shape");
  int x0, y0, lin0;
  __device__ __forceinline__ PixTile(int tile, int W) {
    if constexpr (TH > 0) {
      const int tx = (W + TW - 1) / TW;
      y0 = (tile / tx) * TH; x0 = (tile % tx) * TW; lin0 = 0;
    } else {
      x0 = y0 = 0; lin0 = tile * PPB;
    }
  }
  // -> linear pixel index (clamped into the map), live = the tile slot is a pixel of the map
  __device__ __forceinline__ int pix(int pl, int W, int H, bool& live) const {
    if constexpr (TH > 0) {
      const int x = x0 + pl % TW, y = y0 + pl / TW;
      live = (x < W) && (y < H);
      return min(y, H - 1) * W + min(x, W - 1);
    } else {
      live = (lin0 + pl) < W * H;
      return min(lin0 + pl, W * H - 1);
    }
  }
  static int blocks(int W, int H) {
    if constexpr (TH > 0) return ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
    else return (W * H + PPB - 1) / PPB;
  }
};

struct Geom {
  float half_w, half_h;  // f32((w-1)/2), f32((h-1)/2)       base.py:117-118
  float sw, sh;          // f32(w/2), f32(h/2)               ATen unnormalise scale
  int w, h;
};

// Sample position in source pixel units.  Rounding order == torch 2.10 CPU (verified bitwise by
// tests/test_oracle_golden.py against the real reference):
//   rot_xyz_i = fma(r_i2, 1, fma(r_i1, y, r_i0*x))      base.py:110 (MKL sgemm, K = 3)
//   P = rot_xyz*depth ; P += t ; px = Px/Pz ; py = Py/Pz  base.py:112-115 (no z>0 test)
//   xn = px / f32((w-1)/2) - 1                             base.py:117 (true divide)
//   ix = fma(xn + 1, w/2, -0.5)                            grid_sample(align_corners=False), FMA-contracted
// The two halves of warp_position: the part that depends on (pixel, view) only -- rot_xyz and the translation --, and the part per
// depth hypothesis.  A kernel whose threads keep their (pixel, view) pair over the planes computes the first half once.
struct PixelRay {
  float q0, q1, q2, t0, t1, t2;
};
__device__ __forceinline__ PixelRay warp_ray(const float* __restrict__ m, float x, float y) {
  PixelRay r;
  r.q0 = __fadd_rn(m[2], __fmaf_rn(m[1], y, __fmul_rn(m[0], x)));
  r.q1 = __fadd_rn(m[6], __fmaf_rn(m[5], y, __fmul_rn(m[4], x)));
  r.q2 = __fadd_rn(m[10], __fmaf_rn(m[9], y, __fmul_rn(m[8], x)));
  r.t0 = m[3]; r.t1 = m[7]; r.t2 = m[11];
  return r;
}
__device__ __forceinline__ void warp_position_ray(const PixelRay& r, float dep, const Geom& g, float& ix, float& iy) {
  const float X = __fadd_rn(__fmul_rn(r.q0, dep), r.t0);
  const float Y = __fadd_rn(__fmul_rn(r.q1, dep), r.t1);
  const float Z = __fadd_rn(__fmul_rn(r.q2, dep), r.t2);
  const float px = __fdiv_rn(X, Z);
  const float py = __fdiv_rn(Y, Z);
  const float xn = __fsub_rn(__fdiv_rn(px, g.half_w), 1.0f);
  const float yn = __fsub_rn(__fdiv_rn(py, g.half_h), 1.0f);
  ix = __fmaf_rn(__fadd_rn(xn, 1.0f), g.sw, -0.5f);
  iy = __fmaf_rn(__fadd_rn(yn, 1.0f), g.sh, -0.5f);
}

__device__ __forceinline__ void warp_position(const float* __restrict__ m, float x, float y, float dep,
                                              const Geom& g, float& ix, float& iy) {
  const float q0 = __fadd_rn(m[2], __fmaf_rn(m[1], y, __fmul_rn(m[0], x)));
  const float q1 = __fadd_rn(m[6], __fmaf_rn(m[5], y, __fmul_rn(m[4], x)));
  const float q2 = __fadd_rn(m[10], __fmaf_rn(m[9], y, __fmul_rn(m[8], x)));
  const float X = __fadd_rn(__fmul_rn(q0, dep), m[3]);
  const float Y = __fadd_rn(__fmul_rn(q1, dep), m[7]);
  const float Z = __fadd_rn(__fmul_rn(q2, dep), m[11]);
  const float px = __fdiv_rn(X, Z);
  const float py = __fdiv_rn(Y, Z);
  const float xn = __fsub_rn(__fdiv_rn(px, g.half_w), 1.0f);
  const float yn = __fsub_rn(__fdiv_rn(py, g.half_h), 1.0f);
  ix = __fmaf_rn(__fadd_rn(xn, 1.0f), g.sw, -0.5f);
  iy = __fmaf_rn(__fadd_rn(yn, 1.0f), g.sh, -0.5f);
}

struct __attribute__((aligned(16))) TapEntry {
  int off[4];   // float offsets of the nw, ne, sw, se taps inside one [h,w,C] map (clamped in range)
  float wt[4];  // bilinear weights; 0 for out-of-bounds taps, NaN for non-finite positions
};

// Weights (bit-exact w.r.t. ATen's grid_sampler) and clamped integer corners of one bilinear sample.
__device__ __forceinline__ void tap_weights_corners(float ix, float iy, const Geom& g, float* wt, int& xa, int& xb, int& ya, int& yb) {
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float fw = __fsub_rn(ix, x0f), fe = __fsub_rn(1.0f, fw);
  const float fn = __fsub_rn(iy, y0f), fs = __fsub_rn(1.0f, fn);
  const float wnw = __fmul_rn(fs, fe), wne = __fmul_rn(fs, fw), wsw = __fmul_rn(fn, fe), wse = __fmul_rn(fn, fw);
  const float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
  const float mw = (float)(g.w - 1), mh = (float)(g.h - 1);
  const bool bx0 = (x0f >= 0.0f) && (x0f <= mw), bx1 = (x1f >= 0.0f) && (x1f <= mw);
  const bool by0 = (y0f >= 0.0f) && (y0f <= mh), by1 = (y1f >= 0.0f) && (y1f <= mh);
  // out-of-bounds taps read 0 in the reference; w*0 keeps NaN/inf weights NaN (z == 0 planes -> NaN).
  wt[0] = (bx0 && by0) ? wnw : __fmul_rn(wnw, 0.0f);
  wt[1] = (bx1 && by0) ? wne : __fmul_rn(wne, 0.0f);
  wt[2] = (bx0 && by1) ? wsw : __fmul_rn(wsw, 0.0f);
  wt[3] = (bx1 && by1) ? wse : __fmul_rn(wse, 0.0f);
  const int xi = (int)fminf(fmaxf(x0f, -2.0f), (float)g.w);  // NaN -> -2
  const int yi = (int)fminf(fmaxf(y0f, -2.0f), (float)g.h);
  xa = min(max(xi, 0), g.w - 1); xb = min(max(xi + 1, 0), g.w - 1);
  ya = min(max(yi, 0), g.h - 1); yb = min(max(yi + 1, 0), g.h - 1);
}

__device__ __forceinline__ void make_taps(float ix, float iy, const Geom& g, int C, TapEntry& t) {
  int xa, xb, ya, yb;
  tap_weights_corners(ix, iy, g, t.wt, xa, xb, ya, yb);
  t.off[0] = (ya * g.w + xa) * C;
  t.off[1] = (ya * g.w + xb) * C;
  t.off[2] = (yb * g.w + xa) * C;
  t.off[3] = (yb * g.w + xb) * C;
}

// The same sample with its corners kept as coordinates (kernels that address an LDS window of the source map).
struct __attribute__((aligned(16))) TapXY {
  int xa, xb, ya, yb;   // clamped in range
  float wt[4];
};

// Reductions over the LPP (4/8/16) lanes of one pixel with DPP row operations (full-rate VALU, no LDS-pipe
// permutes): xor-1 and xor-2 inside the quad, then row_half_mirror / row_mirror -- valid because after the quad steps
// all 4 lanes of a quad already hold the same partial result.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int LPP>
__device__ __forceinline__ float pixel_sum(float v) {
  v += dpp_mov<0xB1>(v);                  // quad_perm [1,0,3,2]
  if (LPP >= 4) v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  if (LPP >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror
  if (LPP >= 16) v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}
template <int LPP>
__device__ __forceinline__ float pixel_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  if (LPP >= 8) v = fmaxf(v, dpp_mov<0x141>(v));
  if (LPP >= 16) v = fmaxf(v, dpp_mov<0x140>(v));
  return v;
}

constexpr float kLog2e = 1.4426950408889634f;

// softmax over a pair (C/G = 2): p0 = e^a/(e^a+e^b) = 1/(1 + e^(b-a)), p1 = 1 - p0.  One v_exp + one v_rcp, no selects;
// b-a -> +inf gives p0 = 0, NaN propagates.  (|error| ~1e-7 vs ATen's exp(x-max)/sum; tolerance of the cost is 2e-6.)
__device__ __forceinline__ float softmax2_p0(float a, float b) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((b - a) * kLog2e));
}
__device__ __forceinline__ void softmax2(float a, float b, float& p0, float& p1) {
  p0 = softmax2_p0(a, b);
  p1 = 1.0f - p0;
}

Geom make_geom(int h, int w) {
  Geom g;
  g.w = w;
  g.h = h;
  g.half_w = (float)((double)(w - 1) / 2.0);
  g.half_h = (float)((double)(h - 1) / 2.0);
  g.sw = (float)((double)w / 2.0);
  g.sh = (float)((double)h / 2.0);
  return g;
}

}  // namespace
