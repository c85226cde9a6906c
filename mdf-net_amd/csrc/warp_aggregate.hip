// Fused plane-sweep warp + cost-volume aggregation for gfx950 (MI355X).
//
// Replaces, per stage, the reference's chain  homo_warping (net/unit/base.py:85-126) ->
// softmax -> mul -> sum -> depth_weight -> accumulate (net/unit/homoaggregate.py:25-46), which
// materialises a [B,C,D,h,w] warped volume per source view and makes ~10 passes over it.
// Here one kernel reads the V feature maps and writes only the [B,G,D,h,w] cost volume.
//
// Data layout: features are NHWC so one bilinear tap of a pixel is C contiguous floats; a lane
// owns 4 consecutive channels (= 2 groups of C/G = 2) and fetches each tap with one 16-byte load,
// so the C/4 lanes of a pixel read a tap as one contiguous C*4-byte segment.
//
// Per block (256 threads, PPB = 256/(C/4) pixels) and per chunk of depth planes:
//   phase A  every thread computes whole sample positions (one per (pixel, plane, view)): the
//            homography, 4 IEEE divides, floor, bilinear weights, bounds -> an LDS table entry
//            {4 tap offsets, 4 weights}.  This arithmetic is done ONCE per sample instead of once
//            per lane, and follows the rounding order of torch's CPU path exactly (see
//            warp_position), so the warped values are bit-identical to the reference's.
//   phase B  the C/4 lanes of a pixel read the entry (LDS broadcast), gather 4x16 B per view,
//            blend, form the group similarities in registers, reduce the view-weight dot product
//            across the pixel's lanes with wavefront shuffles, and accumulate over views.
//
// Compile with -ffp-contract=off: every fused multiply-add below is explicit.
#include <cstdlib>
#include "warp_common.h"

namespace {

enum Mode { kWarp = 0, kVec = 1, kVar = 2 };

struct Params {
  const float* ref;                      // [B,h,w,C] (unused for kWarp)
  const float* src[MDF_MAX_SRC_VIEWS];   // each [B,h,w,C]
  const float* proj;                     // [n_src,B,12]
  const float* hypos;                    // [B,D] or [B,D,h,w]
  const float* wpar;                     // [G+4] (kVec)
  float* out;
  Geom g;
  int B, D, n_src, hypos_per_pixel, out_ndhwc, dchunk, nblk_x;
};

template <int C, int MODE>
__global__ __launch_bounds__(kThreads) void warp_kernel(const Params p) {
  constexpr int LPP = C / 4;           // lanes per pixel
  constexpr int PPB = kThreads / LPP;  // pixels per block
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TapEntry* tab = reinterpret_cast<TapEntry*>(smem);

  const int hw = p.g.h * p.g.w;
  const int b = blockIdx.y;
  const PixTile<PPB> pt((int)mdf::xcd_remap(blockIdx.x, p.nblk_x), p.g.w);
  const int tid = threadIdx.x;
  const int pl = tid / LPP;   // pixel inside the block tile
  const int sub = tid % LPP;  // which 4-channel slice of the pixel
  bool live;
  const int pix = pt.pix(pl, p.g.w, p.g.h, live);
  const int G = C / 2;

  // per-lane constants
  float r[4] = {0.f, 0.f, 0.f, 0.f};   // kVec: ref group softmax (p0,p1 | p0,p1); kVar: raw ref
  float cw0 = 0.f, cw1 = 0.f, alpha = 0.f, beta = 0.f, w2 = 0.f, b2 = 0.f;
  if (MODE != kWarp) {
    const float4 rv = *reinterpret_cast<const float4*>(p.ref + ((size_t)b * hw + pix) * C + 4 * sub);
    if (MODE == kVec) {
      softmax2(rv.x, rv.y, r[0], r[1]);
      softmax2(rv.z, rv.w, r[2], r[3]);
      r[0] -= r[1];  // sim = p0*r0 + (1-p0)*r1 = r1 + p0*(r0 - r1)
      r[2] -= r[3];
      cw0 = p.wpar[2 * sub];
      cw1 = p.wpar[2 * sub + 1];
      alpha = p.wpar[G];
      beta = p.wpar[G + 1];
      w2 = p.wpar[G + 2];
      b2 = p.wpar[G + 3];
    } else {
      r[0] = rv.x; r[1] = rv.y; r[2] = rv.z; r[3] = rv.w;
    }
  }
  const size_t map_stride = (size_t)hw * C;
  const unsigned lane_b = 16u * (unsigned)sub;      // byte offset of this lane's 4 channels inside a texel

  // phase A bookkeeping: when the (pixel, view) pairs of the tile divide the block (5 views: 4 x PPB = 256 / 128 / 64), a thread keeps
  // ITS pair over all planes and chunks, so the pair's index arithmetic, its three 16-byte loads of the projection and rot_xyz are done
  // once per kernel instead of once per sample (they were a third of the per-sample instructions; the kernel is VALU-bound)
  const int npair = PPB * p.n_src;
  const bool fixed_pair = (kThreads % npair) == 0;
  const int pa_pair = tid % npair, pa_grp = tid / npair, pa_ngrp = kThreads / npair;
  const int pa_pl = pa_pair % PPB, pa_v = pa_pair / PPB;
  bool pa_live;
  const int pa_pix = pt.pix(pa_pl, p.g.w, p.g.h, pa_live);
  PixelRay ray{};
  if (fixed_pair) {
    const int yy = pa_pix / p.g.w, xx = pa_pix - yy * p.g.w;
    ray = warp_ray(p.proj + ((size_t)pa_v * p.B + b) * 12, (float)xx, (float)yy);
  }

  for (int d0 = 0; d0 < p.D; d0 += p.dchunk) {
    const int nd = min(p.dchunk, p.D - d0);
    // ---------------- phase A: one thread per (plane, view, pixel) sample
    if (fixed_pair) {
      for (int ed = pa_grp; ed < nd; ed += pa_ngrp) {
        const int d = d0 + ed;
        const float dep = p.hypos_per_pixel ? p.hypos[((size_t)b * p.D + d) * hw + pa_pix] : p.hypos[(size_t)b * p.D + d];
        float ix, iy;
        warp_position_ray(ray, dep, p.g, ix, iy);
        TapEntry t;
        make_taps(ix, iy, p.g, C, t);
        tab[(ed * p.n_src + pa_v) * PPB + pa_pl] = t;
      }
    } else {
      const int nent = nd * p.n_src * PPB;
      for (int e = tid; e < nent; e += kThreads) {
        const int epl = e % PPB;
        const int ev = (e / PPB) % p.n_src;
        const int ed = e / (PPB * p.n_src);
        bool elive;
        const int epix = pt.pix(epl, p.g.w, p.g.h, elive);
        const int yy = epix / p.g.w, xx = epix - yy * p.g.w;
        const float* m = p.proj + ((size_t)ev * p.B + b) * 12;
        const int d = d0 + ed;
        const float dep = p.hypos_per_pixel ? p.hypos[((size_t)b * p.D + d) * hw + epix] : p.hypos[(size_t)b * p.D + d];
        float ix, iy;
        warp_position(m, (float)xx, (float)yy, dep, p.g, ix, iy);
        TapEntry t;
        make_taps(ix, iy, p.g, C, t);
        tab[e] = t;
      }
    }
    __syncthreads();
    // ---------------- phase B
    for (int dd = 0; dd < nd; ++dd) {
      const int d = d0 + dd;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      float acc2[4] = {0.f, 0.f, 0.f, 0.f};
      float wsum = 0.f;
      if (MODE == kVar) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[k] = r[k]; acc2[k] = r[k] * r[k]; }
      }
      for (int v = 0; v < p.n_src; ++v) {
        const TapEntry t = tab[(dd * p.n_src + v) * PPB + pl];
        // uniform base (SGPR pair) + 32-bit byte offset per lane: `global_load_dwordx4 v, v_off, s[base]`.  With per-lane 64-bit
        // pointers the four gathers cost 13 VALU instructions of address arithmetic per (plane, view) -- a fifth of this loop,
        // which is VALU-bound (r03, ISA of warp_kernel<32,kVec>)
        const char* sb = reinterpret_cast<const char*>(p.src[v] + (size_t)b * map_stride);
        const float4 nw = *reinterpret_cast<const float4*>(sb + ((unsigned)t.off[0] * 4u + lane_b));
        const float4 ne = *reinterpret_cast<const float4*>(sb + ((unsigned)t.off[1] * 4u + lane_b));
        const float4 sw = *reinterpret_cast<const float4*>(sb + ((unsigned)t.off[2] * 4u + lane_b));
        const float4 se = *reinterpret_cast<const float4*>(sb + ((unsigned)t.off[3] * 4u + lane_b));
        float val[4];
        // ATen tap order: nw*w + ne*w + sw*w + se*w, each step one fma
        val[0] = __fmaf_rn(se.x, t.wt[3], __fmaf_rn(sw.x, t.wt[2], __fmaf_rn(ne.x, t.wt[1], __fmul_rn(nw.x, t.wt[0]))));
        val[1] = __fmaf_rn(se.y, t.wt[3], __fmaf_rn(sw.y, t.wt[2], __fmaf_rn(ne.y, t.wt[1], __fmul_rn(nw.y, t.wt[0]))));
        val[2] = __fmaf_rn(se.z, t.wt[3], __fmaf_rn(sw.z, t.wt[2], __fmaf_rn(ne.z, t.wt[1], __fmul_rn(nw.z, t.wt[0]))));
        val[3] = __fmaf_rn(se.w, t.wt[3], __fmaf_rn(sw.w, t.wt[2], __fmaf_rn(ne.w, t.wt[1], __fmul_rn(nw.w, t.wt[0]))));
        if (MODE == kWarp) {
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] = val[k];
        } else if (MODE == kVec) {
          const float sim0 = __fmaf_rn(softmax2_p0(val[0], val[1]), r[0], r[1]);   // homoaggregate.py:38-39
          const float sim1 = __fmaf_rn(softmax2_p0(val[2], val[3]), r[2], r[3]);
          const float z = pixel_sum<LPP>(__fmaf_rn(cw0, sim0, cw1 * sim1));        // Conv3d(G->1, 1x1x1)
          const float u = __fmaf_rn(fmaxf(__fmaf_rn(z, alpha, beta), 0.0f), w2, b2);  // BN(eval) -> ReLU -> Conv3d(1->1)
          const float wv = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-u * kLog2e));  // Sigmoid
          wsum += wv;
          acc[0] += wv * sim0;
          acc[1] += wv * sim1;
        } else {  // kVar: softmax over all C channels of this pixel (homoaggregate.py:60)
          const float mx = pixel_max<LPP>(fmaxf(fmaxf(val[0], val[1]), fmaxf(val[2], val[3])));
          float e[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) e[k] = expf(val[k] - mx);
          const float den = pixel_sum<LPP>((e[0] + e[1]) + (e[2] + e[3]));
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float pr = e[k] / den;
            acc[k] += pr;
            acc2[k] += pr * pr;
          }
        }
      }
      if (!live) continue;
      const size_t vox = ((size_t)b * p.D + d) * hw + pix;
      if (MODE == kVec) {
        const float o0 = acc[0] / wsum, o1 = acc[1] / wsum;  // homoaggregate.py:46
        if (p.out_ndhwc) {
          { typedef float f2v __attribute__((ext_vector_type(2))); f2v ov = {o0, o1}; __builtin_nontemporal_store(ov, reinterpret_cast<f2v*>(p.out + vox * G + 2 * sub)); }   // (non-temporal: see warp_vec8_kernel)
        } else {
          const size_t cs = (size_t)p.D * hw;
          float* o = p.out + ((size_t)b * G * p.D + d) * hw + pix;
          o[(size_t)(2 * sub) * cs] = o0;
          o[(size_t)(2 * sub + 1) * cs] = o1;
        }
      } else {
        float o[4];
        if (MODE == kVar) {
          const float n = (float)(p.n_src + 1);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float mean = acc[k] / n;
            o[k] = acc2[k] / n - mean * mean;  // homoaggregate.py:66
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = acc[k];
        }
        if (p.out_ndhwc) {
          *reinterpret_cast<float4*>(p.out + vox * C + 4 * sub) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
          const size_t cs = (size_t)p.D * hw;
          float* op = p.out + ((size_t)b * C * p.D + d) * hw + pix;
#pragma unroll
          for (int k = 0; k < 4; ++k) op[(size_t)(4 * sub + k) * cs] = o[k];
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// kVec with EIGHT channels (four groups) per lane.  The loop of warp_kernel<C,kVec> is VALU-bound (PMC: the vector ALU is ~72 %
// busy), and a third of its ~45 instructions per (lane, plane, view) is per-PIXEL work every lane of the pixel repeats: the
// reduction of the view-weight dot product, BN -> ReLU -> 1x1 conv -> sigmoid (two transcendentals), the weighted accumulation.
// With 8 channels per lane a pixel has half the lanes, so that part runs half as often, the reduction tree is one step shorter,
// and the two 16-byte gathers of a corner are one contiguous 32 bytes per lane.  Lane s takes the channels of lanes 2s and 2s+1
// of the 4-channel kernel and adds their two partial dot products first -- the first step of that kernel's reduction tree --, so
// every sum is formed from the same operands: bit-identical cost volume.
#ifndef MDF_VEC8_MIN_BLOCKS
#define MDF_VEC8_MIN_BLOCKS 1      // dev: blocks per CU the register allocation must allow
#endif
#ifndef MDF_VEC8_TAB
#define MDF_VEC8_TAB 1024          // dev: tap-table entries (32 B each) per block
#endif
template <int C>
__global__ __launch_bounds__(kThreads, MDF_VEC8_MIN_BLOCKS) void warp_vec8_kernel(const Params p) {
  constexpr int LPP = C / 8;           // lanes per pixel
  constexpr int PPB = kThreads / LPP;  // pixels per block
  constexpr int G = C / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TapEntry* tab = reinterpret_cast<TapEntry*>(smem);
  const int hw = p.g.h * p.g.w;
  const int b = blockIdx.y;
  const PixTile<PPB> pt((int)mdf::xcd_remap(blockIdx.x, p.nblk_x), p.g.w);
  const int tid = threadIdx.x;
  const int pl = tid / LPP, sub = tid % LPP;
  bool live;
  const int pix = pt.pix(pl, p.g.w, p.g.h, live);

  float rd[2][2], r1[2][2], cw[2][2];          // [half j = 4-channel slice 2*sub + j][group]: r0 - r1, r1 of the reference softmax; conv weights
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float4 rv = *reinterpret_cast<const float4*>(p.ref + ((size_t)b * hw + pix) * C + 8 * sub + 4 * j);
    float a0, a1, b0, b1;
    softmax2(rv.x, rv.y, a0, a1);
    softmax2(rv.z, rv.w, b0, b1);
    rd[j][0] = a0 - a1; r1[j][0] = a1;
    rd[j][1] = b0 - b1; r1[j][1] = b1;
    cw[j][0] = p.wpar[2 * (2 * sub + j)];
    cw[j][1] = p.wpar[2 * (2 * sub + j) + 1];
  }
  const float alpha = p.wpar[G], beta = p.wpar[G + 1], w2 = p.wpar[G + 2], b2 = p.wpar[G + 3];
  const size_t map_stride = (size_t)hw * C;
  const unsigned lane_b = 32u * (unsigned)sub;      // byte offset of this lane's 8 channels inside a texel

  // tap table: a thread keeps its (pixel, view) pair(s) over the planes (warp_kernel, phase A); here a tile has up to 2 pairs per thread
  const int npair = PPB * p.n_src;
  const bool few = (npair <= kThreads) && (kThreads % npair) == 0;          // >= 1 thread group per pair set
  const bool two = (npair == 2 * kThreads);                                 // exactly two pairs per thread
  const int pa_ngrp = few ? kThreads / npair : 1, pa_grp = few ? tid / npair : 0;
  PixelRay ray[2];
  int pa_pl[2], pa_v[2], pa_pix[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int pair = (few ? tid % npair : tid) + k * kThreads;
    pa_pl[k] = pair % PPB; pa_v[k] = min(pair / PPB, p.n_src - 1);
    bool pa_live;
    pa_pix[k] = pt.pix(pa_pl[k], p.g.w, p.g.h, pa_live);
    const int yy = pa_pix[k] / p.g.w, xx = pa_pix[k] - yy * p.g.w;
    ray[k] = warp_ray(p.proj + ((size_t)pa_v[k] * p.B + b) * 12, (float)xx, (float)yy);
  }

  for (int d0 = 0; d0 < p.D; d0 += p.dchunk) {
    const int nd = min(p.dchunk, p.D - d0);
    if (few || two) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (k == 1 && !two) break;
        for (int ed = pa_grp; ed < nd; ed += pa_ngrp) {
          const int d = d0 + ed;
          const float dep = p.hypos_per_pixel ? p.hypos[((size_t)b * p.D + d) * hw + pa_pix[k]] : p.hypos[(size_t)b * p.D + d];
          float ix, iy;
          warp_position_ray(ray[k], dep, p.g, ix, iy);
          TapEntry t;
          make_taps(ix, iy, p.g, C, t);
          tab[(ed * p.n_src + pa_v[k]) * PPB + pa_pl[k]] = t;
        }
      }
    } else {
      const int nent = nd * p.n_src * PPB;
      for (int e = tid; e < nent; e += kThreads) {
        const int epl = e % PPB;
        const int ev = (e / PPB) % p.n_src;
        const int ed = e / (PPB * p.n_src);
        bool elive;
        const int epix = pt.pix(epl, p.g.w, p.g.h, elive);
        const int yy = epix / p.g.w, xx = epix - yy * p.g.w;
        const float* m = p.proj + ((size_t)ev * p.B + b) * 12;
        const int d = d0 + ed;
        const float dep = p.hypos_per_pixel ? p.hypos[((size_t)b * p.D + d) * hw + epix] : p.hypos[(size_t)b * p.D + d];
        float ix, iy;
        warp_position(m, (float)xx, (float)yy, dep, p.g, ix, iy);
        TapEntry t;
        make_taps(ix, iy, p.g, C, t);
        tab[e] = t;
      }
    }
    __syncthreads();
    for (int dd = 0; dd < nd; ++dd) {
      float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
      float wsum = 0.f;
      for (int v = 0; v < p.n_src; ++v) {
        const TapEntry t = tab[(dd * p.n_src + v) * PPB + pl];
        const char* sb = reinterpret_cast<const char*>(p.src[v] + (size_t)b * map_stride);
        float sim[2][2], part[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const unsigned lb = lane_b + 16u * j;
          const float4 nw = *reinterpret_cast<const float4*>(sb + ((unsigned)t.off[0] * 4u + lb));
          const float4 ne = *reinterpret_cast<const float4*>(sb + ((unsigned)t.off[1] * 4u + lb));
          const float4 sw = *reinterpret_cast<const float4*>(sb + ((unsigned)t.off[2] * 4u + lb));
          const float4 se = *reinterpret_cast<const float4*>(sb + ((unsigned)t.off[3] * 4u + lb));
          const float v0 = __fmaf_rn(se.x, t.wt[3], __fmaf_rn(sw.x, t.wt[2], __fmaf_rn(ne.x, t.wt[1], __fmul_rn(nw.x, t.wt[0]))));
          const float v1 = __fmaf_rn(se.y, t.wt[3], __fmaf_rn(sw.y, t.wt[2], __fmaf_rn(ne.y, t.wt[1], __fmul_rn(nw.y, t.wt[0]))));
          const float v2 = __fmaf_rn(se.z, t.wt[3], __fmaf_rn(sw.z, t.wt[2], __fmaf_rn(ne.z, t.wt[1], __fmul_rn(nw.z, t.wt[0]))));
          const float v3 = __fmaf_rn(se.w, t.wt[3], __fmaf_rn(sw.w, t.wt[2], __fmaf_rn(ne.w, t.wt[1], __fmul_rn(nw.w, t.wt[0]))));
          sim[j][0] = __fmaf_rn(softmax2_p0(v0, v1), rd[j][0], r1[j][0]);   // homoaggregate.py:38-39
          sim[j][1] = __fmaf_rn(softmax2_p0(v2, v3), rd[j][1], r1[j][1]);
          part[j] = __fmaf_rn(cw[j][0], sim[j][0], cw[j][1] * sim[j][1]);
        }
        const float z = pixel_sum<LPP>(part[0] + part[1]);                     // Conv3d(G->1, 1x1x1); first tree step in the lane
        const float u = __fmaf_rn(fmaxf(__fmaf_rn(z, alpha, beta), 0.0f), w2, b2);  // BN(eval) -> ReLU -> Conv3d(1->1)
        const float wv = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-u * kLog2e));  // Sigmoid
        wsum += wv;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[j][0] += wv * sim[j][0];
          acc[j][1] += wv * sim[j][1];
        }
      }
      if (!live) continue;
      const size_t vox = ((size_t)b * p.D + d0 + dd) * hw + pix;
      const float o0 = acc[0][0] / wsum, o1 = acc[0][1] / wsum, o2 = acc[1][0] / wsum, o3 = acc[1][1] / wsum;   // homoaggregate.py:46
      if (p.out_ndhwc) {
        // written once, read by another launch: a non-temporal store keeps the 121-182 MB of a cost volume from sweeping the source
        // texels out of this XCD's 4-MB L2 (r04: -1 % per launch, bit-identical)
        { typedef float f4v __attribute__((ext_vector_type(4))); f4v ov = {o0, o1, o2, o3}; __builtin_nontemporal_store(ov, reinterpret_cast<f4v*>(p.out + vox * G + 4 * sub)); }
      } else {
        const size_t cs = (size_t)p.D * hw;
        float* o = p.out + ((size_t)b * G * p.D + d0 + dd) * hw + pix;
        o[(size_t)(4 * sub) * cs] = o0;
        o[(size_t)(4 * sub + 1) * cs] = o1;
        o[(size_t)(4 * sub + 2) * cs] = o2;
        o[(size_t)(4 * sub + 3) * cs] = o3;
      }
    }
    __syncthreads();
  }
}

template <int C>
int launch_vec8(Params& p, hipStream_t st) {
  constexpr int ppb = kThreads / (C / 8);
  p.nblk_x = PixTile<ppb>::blocks(p.g.w, p.g.h);
  int dch = MDF_VEC8_TAB / (p.n_src * ppb);  // ~32 KiB of tap table per block (twice the pixels of warp_kernel's tile, the same planes)
  if (dch < 1) dch = 1;
  if (dch > p.D) dch = p.D;
  p.dchunk = dch;
  const size_t lds = (size_t)dch * p.n_src * ppb * sizeof(TapEntry);
  if (lds > 64 * 1024) return MDF_EUNSUPPORTED;
  hipLaunchKernelGGL((warp_vec8_kernel<C>), dim3(p.nblk_x, p.B), dim3(kThreads), lds, st, p);
  return mdf::check_launch("warp_vec8_kernel");
}

// ------------------------------------------------------------------------------------------------------------------
// kVec with LDS-staged source-feature tiles.  Per depth chunk the block finds, for every source view, the bounding box
// of its taps in that view's feature map (min/max over the tap table), loads the boxes ONCE with coalesced row loads
// (ww*C contiguous floats per window row in NHWC) into a pool of LDS windows, and takes the 4 bilinear taps of every
// sample from LDS: a texel crosses the L1/TA path once per chunk instead of once per tap that touches it (4 taps x
// planes x neighbouring pixels).  The loop order, the accumulation order over views and all arithmetic are those of
// warp_kernel<C,kVec>, so the cost volume is bit-identical.  A view whose box does not fit what is left of the pool
// (strong rotation, wide depth range) gathers from memory as before -- a block-uniform, per-view decision.
struct WinDesc {
  int off;          // float offset of the window in the pool, -1: gather from memory
  int xmin, ymin, ww;
};

template <int C>
__global__ __launch_bounds__(kThreads) void warp_vec_win_kernel(const Params p, int pool_floats) {
  constexpr int LPP = C / 4;
  constexpr int PPB = kThreads / LPP;
  constexpr int G = C / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TapXY* tab = reinterpret_cast<TapXY*>(smem);
  float* pool = reinterpret_cast<float*>(smem + (size_t)p.dchunk * p.n_src * PPB * sizeof(TapXY));
  __shared__ int bb[MDF_MAX_SRC_VIEWS][4];
  __shared__ WinDesc wd[MDF_MAX_SRC_VIEWS];

  const int hw = p.g.h * p.g.w;
  const int W = p.g.w;
  const int b = blockIdx.y;
  const int tile = (int)mdf::xcd_remap(blockIdx.x, p.nblk_x);
  const int pix0 = tile * PPB;
  const int tid = threadIdx.x;
  const int pl = tid / LPP, sub = tid % LPP;
  const int pix = min(pix0 + pl, hw - 1);
  const bool live = (pix0 + pl) < hw;

  float r[4];
  {
    const float4 rv = *reinterpret_cast<const float4*>(p.ref + ((size_t)b * hw + pix) * C + 4 * sub);
    softmax2(rv.x, rv.y, r[0], r[1]);
    softmax2(rv.z, rv.w, r[2], r[3]);
    r[0] -= r[1];
    r[2] -= r[3];
  }
  const float cw0 = p.wpar[2 * sub], cw1 = p.wpar[2 * sub + 1];
  const float alpha = p.wpar[G], beta = p.wpar[G + 1], w2 = p.wpar[G + 2], b2 = p.wpar[G + 3];
  const size_t map_stride = (size_t)hw * C;

  for (int d0 = 0; d0 < p.D; d0 += p.dchunk) {
    const int nd = min(p.dchunk, p.D - d0);
    if (tid < 4 * p.n_src) bb[tid >> 2][tid & 3] = (tid & 1) ? INT32_MIN : INT32_MAX;
    __syncthreads();
    const int nent = nd * p.n_src * PPB;
    for (int e = tid; e < nent; e += kThreads) {
      const int epl = e % PPB;
      const int ev = (e / PPB) % p.n_src;
      const int ed = e / (PPB * p.n_src);
      const int epix = min(pix0 + epl, hw - 1);
      const int yy = epix / W, xx = epix - yy * W;
      const float* m = p.proj + ((size_t)ev * p.B + b) * 12;
      const int d = d0 + ed;
      const float dep = p.hypos_per_pixel ? p.hypos[((size_t)b * p.D + d) * hw + epix] : p.hypos[(size_t)b * p.D + d];
      float ix, iy;
      warp_position(m, (float)xx, (float)yy, dep, p.g, ix, iy);
      TapXY t;
      tap_weights_corners(ix, iy, p.g, t.wt, t.xa, t.xb, t.ya, t.yb);
      tab[e] = t;
      // every tap is READ (a zero weight still multiplies a finite value), so the box covers all four clamped corners
      atomicMin(&bb[ev][0], t.xa);
      atomicMax(&bb[ev][1], t.xb);
      atomicMin(&bb[ev][2], t.ya);
      atomicMax(&bb[ev][3], t.yb);
    }
    __syncthreads();
    if (tid == 0) {   // hand out pool space view by view
      int used = 0;
      for (int v = 0; v < p.n_src; ++v) {
        const int ww = bb[v][1] - bb[v][0] + 1, wh = bb[v][3] - bb[v][2] + 1;
        const long long need = (long long)ww * wh * C;
        WinDesc w_;
        w_.xmin = bb[v][0]; w_.ymin = bb[v][2]; w_.ww = ww;
        if (need <= pool_floats - used) { w_.off = used; used += (int)need; } else { w_.off = -1; }
        wd[v] = w_;
      }
    }
    __syncthreads();
    for (int v = 0; v < p.n_src; ++v) {
      const WinDesc w_ = wd[v];
      if (w_.off < 0) continue;
      const int wh = bb[v][3] - bb[v][2] + 1;
      const int row4 = w_.ww * (C / 4);                 // float4s per window row, contiguous in memory
      const float* sv = p.src[v] + (size_t)b * map_stride;
      for (int wy = 0; wy < wh; ++wy) {
        const float4* grow = reinterpret_cast<const float4*>(sv + ((size_t)(w_.ymin + wy) * W + w_.xmin) * C);
        float4* lrow = reinterpret_cast<float4*>(pool + w_.off) + wy * row4;
        for (int j = tid; j < row4; j += kThreads) lrow[j] = grow[j];
      }
    }
    __syncthreads();

    for (int dd = 0; dd < nd; ++dd) {
      float acc0 = 0.f, acc1 = 0.f, wsum = 0.f;
      for (int v = 0; v < p.n_src; ++v) {
        const TapXY t = tab[(dd * p.n_src + v) * PPB + pl];
        const WinDesc w_ = wd[v];
        float4 nw, ne, sw, se;
        if (w_.off >= 0) {
          const float* lp = pool + w_.off + 4 * sub;
          const int ra = (t.ya - w_.ymin) * w_.ww, rb = (t.yb - w_.ymin) * w_.ww, ca = t.xa - w_.xmin, cb = t.xb - w_.xmin;
          nw = *reinterpret_cast<const float4*>(lp + (ra + ca) * C);
          ne = *reinterpret_cast<const float4*>(lp + (ra + cb) * C);
          sw = *reinterpret_cast<const float4*>(lp + (rb + ca) * C);
          se = *reinterpret_cast<const float4*>(lp + (rb + cb) * C);
        } else {
          const float* sp = p.src[v] + (size_t)b * map_stride + 4 * sub;
          nw = *reinterpret_cast<const float4*>(sp + (size_t)(t.ya * W + t.xa) * C);
          ne = *reinterpret_cast<const float4*>(sp + (size_t)(t.ya * W + t.xb) * C);
          sw = *reinterpret_cast<const float4*>(sp + (size_t)(t.yb * W + t.xa) * C);
          se = *reinterpret_cast<const float4*>(sp + (size_t)(t.yb * W + t.xb) * C);
        }
        const float v0 = __fmaf_rn(se.x, t.wt[3], __fmaf_rn(sw.x, t.wt[2], __fmaf_rn(ne.x, t.wt[1], __fmul_rn(nw.x, t.wt[0]))));
        const float v1 = __fmaf_rn(se.y, t.wt[3], __fmaf_rn(sw.y, t.wt[2], __fmaf_rn(ne.y, t.wt[1], __fmul_rn(nw.y, t.wt[0]))));
        const float v2 = __fmaf_rn(se.z, t.wt[3], __fmaf_rn(sw.z, t.wt[2], __fmaf_rn(ne.z, t.wt[1], __fmul_rn(nw.z, t.wt[0]))));
        const float v3 = __fmaf_rn(se.w, t.wt[3], __fmaf_rn(sw.w, t.wt[2], __fmaf_rn(ne.w, t.wt[1], __fmul_rn(nw.w, t.wt[0]))));
        const float sim0 = __fmaf_rn(softmax2_p0(v0, v1), r[0], r[1]);
        const float sim1 = __fmaf_rn(softmax2_p0(v2, v3), r[2], r[3]);
        const float z = pixel_sum<LPP>(__fmaf_rn(cw0, sim0, cw1 * sim1));
        const float u = __fmaf_rn(fmaxf(__fmaf_rn(z, alpha, beta), 0.0f), w2, b2);
        const float wv = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-u * kLog2e));
        wsum += wv;
        acc0 += wv * sim0;
        acc1 += wv * sim1;
      }
      if (!live) continue;
      const size_t vox = ((size_t)b * p.D + d0 + dd) * hw + pix;
      const float o0 = acc0 / wsum, o1 = acc1 / wsum;
      if (p.out_ndhwc) {
        *reinterpret_cast<float2*>(p.out + vox * G + 2 * sub) = make_float2(o0, o1);
      } else {
        const size_t cs = (size_t)p.D * hw;
        float* o = p.out + ((size_t)b * G * p.D + d0 + dd) * hw + pix;
        o[(size_t)(2 * sub) * cs] = o0;
        o[(size_t)(2 * sub + 1) * cs] = o1;
      }
    }
    __syncthreads();   // the next chunk overwrites tab, bb, wd and the pool
  }
}

template <int C>
int launch_win(Params& p, hipStream_t st) {
  constexpr int ppb = kThreads / (C / 4);
  const int hw = p.g.h * p.g.w;
  p.nblk_x = (hw + ppb - 1) / ppb;
  int dch = 512 / (p.n_src * ppb);
  if (const char* e = getenv("MDF_WARP_DCHUNK")) { if (atoi(e) > 0) dch = atoi(e); }   // dev A/B
  if (dch < 1) dch = 1;
  if (dch > p.D) dch = p.D;
  p.dchunk = dch;
  int pool_kb = 64;
  if (const char* e = getenv("MDF_WARP_POOL_KB")) { if (atoi(e) > 0) pool_kb = atoi(e); }   // dev A/B
  const size_t tab_bytes = (size_t)dch * p.n_src * ppb * sizeof(TapXY);
  const size_t lds = tab_bytes + (size_t)pool_kb * 1024;
  if (lds > 150 * 1024) return MDF_EUNSUPPORTED;
  static bool attr_done[64] = {};
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  if (dev_id < 0 || dev_id >= 64 || !attr_done[dev_id]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&warp_vec_win_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return mdf::fail(MDF_EHIP, "hipFuncSetAttribute(dynamic LDS): %s", hipGetErrorString(e));
    if (dev_id >= 0 && dev_id < 64) attr_done[dev_id] = true;
  }
  hipLaunchKernelGGL((warp_vec_win_kernel<C>), dim3(p.nblk_x, p.B), dim3(kThreads), lds, st, p, pool_kb * 256);
  return mdf::check_launch("warp_vec_win_kernel");
}

__global__ void corner_index_kernel(const float* __restrict__ proj, const float* __restrict__ hypos, int per_pixel,
                                    int32_t* __restrict__ out, Geom g, int B, int D) {
  const int hw = g.h * g.w;
  const size_t n = (size_t)B * D * hw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int pix = (int)(i % hw);
    const int d = (int)((i / hw) % D);
    const int b = (int)(i / ((size_t)hw * D));
    const int yy = pix / g.w, xx = pix - yy * g.w;
    const float dep = per_pixel ? hypos[i] : hypos[(size_t)b * D + d];
    float ix, iy;
    warp_position(proj + (size_t)b * 12, (float)xx, (float)yy, dep, g, ix, iy);
    const float lim = 1073741824.0f;
    const bool fin = (fabsf(ix) <= 3.0e38f) && (fabsf(iy) <= 3.0e38f);  // false for NaN/inf
    int32_t x0 = INT32_MIN, y0 = INT32_MIN;
    if (fin) {
      x0 = (int32_t)fminf(fmaxf(floorf(ix), -lim), lim);
      y0 = (int32_t)fminf(fmaxf(floorf(iy), -lim), lim);
    }
    out[2 * i] = x0;
    out[2 * i + 1] = y0;
  }
}

template <int MODE>
int launch(Params& p, int C, hipStream_t st) {
  const int lpp = C / 4, ppb = kThreads / lpp;
  p.nblk_x = (C == 64) ? PixTile<16>::blocks(p.g.w, p.g.h) : (C == 32) ? PixTile<32>::blocks(p.g.w, p.g.h) : PixTile<64>::blocks(p.g.w, p.g.h);
  int dch = 512 / (p.n_src * ppb);  // ~16 KiB of tap table per block
  if (dch < 1) dch = 1;
  if (dch > p.D) dch = p.D;
  p.dchunk = dch;
  const size_t lds = (size_t)dch * p.n_src * ppb * sizeof(TapEntry);
  dim3 grid(p.nblk_x, p.B), block(kThreads);
  switch (C) {
    case 64: hipLaunchKernelGGL((warp_kernel<64, MODE>), grid, block, lds, st, p); break;
    case 32: hipLaunchKernelGGL((warp_kernel<32, MODE>), grid, block, lds, st, p); break;
    case 16: hipLaunchKernelGGL((warp_kernel<16, MODE>), grid, block, lds, st, p); break;
    default: return mdf::fail(MDF_EUNSUPPORTED, "warp kernels are built for C in {16,32,64}, got %d", C);
  }
  return mdf::check_launch("warp_kernel");
}

int check_common(const void* a, const void* b, const void* c, const void* d, int fea_layout, int B, int C, int D, int h,
                 int w) {
  MDF_REQUIRE(a && b && c && d, "null pointer argument");
  MDF_REQUIRE(B > 0 && D > 0 && h > 1 && w > 1, "bad shape B=%d D=%d h=%d w=%d", B, D, h, w);
  MDF_REQUIRE((long long)h * w * C < (1ll << 30), "feature map too large for 32-bit byte offsets");
  if (fea_layout != MDF_FEA_NHWC)
    return mdf::fail(MDF_EUNSUPPORTED, "feature layout %d not supported (kernels gather NHWC taps)", fea_layout);
  if (C != 16 && C != 32 && C != 64)
    return mdf::fail(MDF_EUNSUPPORTED, "C=%d not supported (built for 16, 32, 64)", C);
  return MDF_OK;
}

}  // namespace

extern "C" int mdf_homo_warp_fwd(const float* src_fea, int fea_layout, const float* proj, const float* hypos,
                                 int hypos_per_pixel, float* out, int out_layout, int B, int C, int D, int h, int w,
                                 void* stream) {
  if (int rc = check_common(src_fea, proj, hypos, out, fea_layout, B, C, D, h, w)) return rc;
  Params p{};
  p.src[0] = src_fea;
  p.proj = proj;
  p.hypos = hypos;
  p.out = out;
  p.g = make_geom(h, w);
  p.B = B; p.D = D; p.n_src = 1; p.hypos_per_pixel = hypos_per_pixel; p.out_ndhwc = (out_layout == MDF_VOL_NDHWC);
  return launch<kWarp>(p, C, (hipStream_t)stream);
}

extern "C" int mdf_warp_corner_indices(const float* proj, const float* hypos, int hypos_per_pixel, int32_t* x0y0,
                                       int B, int D, int h, int w, void* stream) {
  MDF_REQUIRE(proj && hypos && x0y0, "null pointer argument");
  MDF_REQUIRE(B > 0 && D > 0 && h > 1 && w > 1, "bad shape");
  const size_t n = (size_t)B * D * h * w;
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(corner_index_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, proj, hypos, hypos_per_pixel,
                     x0y0, make_geom(h, w), B, D);
  return mdf::check_launch("corner_index_kernel");
}

extern "C" int mdf_warp_aggregate_vec_fwd(const float* ref_fea, const float* const* src_feas, int fea_layout,
                                          const float* proj, const float* hypos, int hypos_per_pixel,
                                          const float* w_params, float* cost, int cost_layout, int B, int C, int G,
                                          int D, int h, int w, int n_src, void* stream) {
  if (int rc = check_common(ref_fea, proj, hypos, cost, fea_layout, B, C, D, h, w)) return rc;
  MDF_REQUIRE(src_feas && w_params, "null pointer argument");
  MDF_REQUIRE(n_src >= 1 && n_src <= MDF_MAX_SRC_VIEWS, "n_src=%d out of range [1,%d]", n_src, MDF_MAX_SRC_VIEWS);
  if (G * 2 != C) return mdf::fail(MDF_EUNSUPPORTED, "only C/G == 2 is built (C=%d, G=%d)", C, G);
  Params p{};
  p.ref = ref_fea;
  for (int v = 0; v < n_src; ++v) {
    MDF_REQUIRE(src_feas[v], "src_feas[%d] is null", v);
    p.src[v] = src_feas[v];
  }
  p.proj = proj; p.hypos = hypos; p.wpar = w_params; p.out = cost;
  p.g = make_geom(h, w);
  p.B = B; p.D = D; p.n_src = n_src; p.hypos_per_pixel = hypos_per_pixel; p.out_ndhwc = (cost_layout == MDF_VOL_NDHWC);
  // LDS-staged source windows: bit-identical, but measured SLOWER than the L1 gather on MI355X (r02: 1.37 ms vs 0.80 ms per
  // cfg2 view at a 32-KiB pool, 3.0 ms at 64 KiB -- the kernel lives on occupancy), so it is opt-in
  const char* we = getenv("MDF_WARP_WINDOW");
  if (we && atoi(we) != 0) {
    int rc = MDF_EUNSUPPORTED;
    if (C == 64) rc = launch_win<64>(p, (hipStream_t)stream);
    else if (C == 32) rc = launch_win<32>(p, (hipStream_t)stream);
    else if (C == 16) rc = launch_win<16>(p, (hipStream_t)stream);
    if (rc != MDF_EUNSUPPORTED) return rc;
  }
  {
    const char* e8 = getenv("MDF_WARP_VEC8");      // dev A/B (read per call): 8 channels per lane
    if (!e8 || atoi(e8) != 0) {
      int rc = MDF_EUNSUPPORTED;
      // (in a cfg2 forward: C 64 243 -> 233 us, C 32 254 -> 250 us; C 16 -- two lanes per pixel, a 128-pixel tile -- 204 -> 221 us: not used)
      if (C == 64) rc = launch_vec8<64>(p, (hipStream_t)stream);
      else if (C == 32) rc = launch_vec8<32>(p, (hipStream_t)stream);
      else if (C == 16 && e8 && atoi(e8) == 2) rc = launch_vec8<16>(p, (hipStream_t)stream);
      if (rc != MDF_EUNSUPPORTED) return rc;
    }
  }
  return launch<kVec>(p, C, (hipStream_t)stream);
}

extern "C" int mdf_warp_aggregate_var_fwd(const float* ref_fea, const float* const* src_feas, int fea_layout,
                                          const float* proj, const float* hypos, int hypos_per_pixel, float* cost,
                                          int cost_layout, int B, int C, int D, int h, int w, int n_src, void* stream) {
  if (int rc = check_common(ref_fea, proj, hypos, cost, fea_layout, B, C, D, h, w)) return rc;
  MDF_REQUIRE(src_feas, "null pointer argument");
  MDF_REQUIRE(n_src >= 1 && n_src <= MDF_MAX_SRC_VIEWS, "n_src=%d out of range [1,%d]", n_src, MDF_MAX_SRC_VIEWS);
  Params p{};
  p.ref = ref_fea;
  for (int v = 0; v < n_src; ++v) {
    MDF_REQUIRE(src_feas[v], "src_feas[%d] is null", v);
    p.src[v] = src_feas[v];
  }
  p.proj = proj; p.hypos = hypos; p.out = cost;
  p.g = make_geom(h, w);
  p.B = B; p.D = D; p.n_src = n_src; p.hypos_per_pixel = hypos_per_pixel; p.out_ndhwc = (cost_layout == MDF_VOL_NDHWC);
  return launch<kVar>(p, C, (hipStream_t)stream);
}
