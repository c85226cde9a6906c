// Training-mode VectorAggregate fused with the plane-sweep warp (net/unit/homoaggregate.py:16-20,25-46 with
// depth_weight's BatchNorm3d(1) in batch-statistics mode, net/unit/base.py:85-126 whose sampling grid is built under
// no_grad, base.py:97 -- gradient flows to the features only).
//
// The 1-channel BatchNorm sits between "similarity" and "view weight" and needs the statistics of the WHOLE volume of
// one source view before any weight can be formed, so every direction is two passes over the (cheap to recompute)
// warp + similarity instead of storing the n_src similarity volumes:
//   kStats      t_v = conv(G->1)(sim_v):  sum t, sum t^2 per source view                      (fp64 atomics)
//   kFwd        cost = sum_v w_v sim_v / sum_v w_v with per-view (alpha_v, beta_v); also writes wsum = sum_v w_v
//   kBwdReduce  per view S1 = sum dz, S2 = sum dz*xhat (BatchNorm backward), plus d w2, d b2   (fp64 atomics)
//   kBwd        dt -> dsim -> d ref (direct store: a block owns its pixels over all planes), d src (scatter through the
//               4 bilinear taps -- the transpose of the gather -- staged in an LDS window, see warp_bwd_kernel), d conv weight
// Same thread mapping as the eval kernel (warp_aggregate.hip): a lane owns 4 channels = 2 groups, the C/4 lanes of a
// pixel reduce with DPP row operations, sample positions are computed once per (pixel, plane, view) into an LDS table.
//
// par (float):  [0,G) conv weight | G: w2, G+1: b2, G+2: gamma, G+3: 1/N | G+4+4v..: alpha_v, beta_v, mean_v, invstd_v
#include <cstdlib>
#include "warp_common.h"

namespace {

enum Pass { kStats = 0, kFwd = 1, kBwdReduce = 2, kBwd = 3 };

struct TrainParams {
  const float* ref;
  const float* src[MDF_MAX_SRC_VIEWS];
  const float* proj;
  const float* hypos;
  const float* par;
  const double* red_in;   // kBwd: [2*n_src] S1_v, S2_v
  const float* dcost;     // [B,D,h,w,G]
  float* cost;            // [B,D,h,w,G]  (written by kFwd, read by the backward passes)
  float* wsum;            // [B,D,h,w]    (same)
  double* red_out;        // kStats: [2*n_src]; kBwdReduce: [2*n_src + 2]
  float* dref;            // [B,h,w,C]
  float* dsrc[MDF_MAX_SRC_VIEWS];   // [B,h,w,G]: gradient of the EVEN channel of every group (odd = -even); zero-initialised by the caller
  float* dcw;             // [G], zero-initialised
  float4* aux;            // [n_src][B*D*h*w] (w_v, dz_v, t_v, -): written by kBwdReduce, read by kBwd
  Geom g;
  int B, D, n_src, hypos_per_pixel, dchunk, nblk_x;
  int dslice;             // planes per blockIdx.z (the depth range is cut into gridDim.z slices: more blocks for the small cfg3 maps)
  int all_atomic;         // debug (MDF_WARP_BWD_ATOMIC=1): every window update is an LDS atomic -- the checker of the claim-based plain adds
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int C, int PASS>
__global__ __launch_bounds__(kThreads) void warp_train_kernel(const TrainParams p) {
  constexpr int LPP = C / 4;
  constexpr int PPB = kThreads / LPP;
  constexpr int G = C / 2;
  constexpr int KMAX = 2 * MDF_MAX_SRC_VIEWS + 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TapEntry* tab = reinterpret_cast<TapEntry*>(smem);
  __shared__ float part[4][KMAX];     // per-wave partial sums of the current depth chunk

  const int hw = p.g.h * p.g.w;
  const int b = blockIdx.y;
  const int tile = (int)mdf::xcd_remap(blockIdx.x, p.nblk_x);
  const int pix0 = tile * PPB;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int pl = tid / LPP, sub = tid % LPP;
  const int pix = min(pix0 + pl, hw - 1);
  const bool live = (pix0 + pl) < hw;
  const bool owner = live && (sub == 0);   // one lane per pixel contributes pixel-level scalars to the reductions

  // reference features: group softmax
  float r[4];
  {
    const float4 rv = *reinterpret_cast<const float4*>(p.ref + ((size_t)b * hw + pix) * C + 4 * sub);
    softmax2(rv.x, rv.y, r[0], r[1]);
    softmax2(rv.z, rv.w, r[2], r[3]);
    r[0] -= r[1];   // sim = r1 + q0*(r0 - r1)
    r[2] -= r[3];
  }
  const float cw0 = p.par[2 * sub], cw1 = p.par[2 * sub + 1];
  const float w2 = p.par[G], b2 = p.par[G + 1];
  const float* vpar = p.par + G + 4;
  const size_t map_stride = (size_t)hw * C;
  const unsigned lane_b = 16u * (unsigned)sub;      // byte offset of this lane's 4 channels inside a texel
  const int nred = (PASS == kStats) ? 2 * p.n_src : 2 * p.n_src + 2;
  double total = 0.0;                 // thread k < nred: block total of reduction slot k

  // tap table: a thread keeps its (pixel, view) pair over the planes when the pairs divide the block (warp_aggregate.hip, phase A)
  const int npair = PPB * p.n_src;
  const bool fixed_pair = (kThreads % npair) == 0;
  const int pa_pair = tid % npair, pa_grp = tid / npair, pa_ngrp = kThreads / npair;
  const int pa_pl = pa_pair % PPB, pa_v = pa_pair / PPB;
  const int pa_pix = min(pix0 + pa_pl, hw - 1);
  PixelRay ray{};
  if (fixed_pair) {
    const int yy = pa_pix / p.g.w, xx = pa_pix - yy * p.g.w;
    ray = warp_ray(p.proj + ((size_t)pa_v * p.B + b) * 12, (float)xx, (float)yy);
  }

  const int dlo = blockIdx.z * p.dslice, dhi = min(p.D, dlo + p.dslice);
  for (int d0 = dlo; d0 < dhi; d0 += p.dchunk) {
    const int nd = min(p.dchunk, dhi - d0);
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
        const int epix = min(pix0 + epl, hw - 1);
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

    if (PASS == kStats || PASS == kBwdReduce) {
      // views outermost: the sums of a view stay in registers over the chunk's planes, one wave reduction per view
      float a3 = 0.f, a4 = 0.f;
      for (int v = 0; v < p.n_src; ++v) {
        float s1 = 0.f, s2 = 0.f;
        const char* sb = reinterpret_cast<const char*>(p.src[v] + (size_t)b * map_stride);     // uniform base + 32-bit lane offsets
        for (int dd = 0; dd < nd; ++dd) {
          const TapEntry t = tab[(dd * p.n_src + v) * PPB + pl];
          const float4 nw = *reinterpret_cast<const float4*>(sb + ((unsigned)t.off[0] * 4u + lane_b));
          const float4 ne = *reinterpret_cast<const float4*>(sb + ((unsigned)t.off[1] * 4u + lane_b));
          const float4 sw = *reinterpret_cast<const float4*>(sb + ((unsigned)t.off[2] * 4u + lane_b));
          const float4 se = *reinterpret_cast<const float4*>(sb + ((unsigned)t.off[3] * 4u + lane_b));
          const float v0 = __fmaf_rn(se.x, t.wt[3], __fmaf_rn(sw.x, t.wt[2], __fmaf_rn(ne.x, t.wt[1], __fmul_rn(nw.x, t.wt[0]))));
          const float v1 = __fmaf_rn(se.y, t.wt[3], __fmaf_rn(sw.y, t.wt[2], __fmaf_rn(ne.y, t.wt[1], __fmul_rn(nw.y, t.wt[0]))));
          const float v2 = __fmaf_rn(se.z, t.wt[3], __fmaf_rn(sw.z, t.wt[2], __fmaf_rn(ne.z, t.wt[1], __fmul_rn(nw.z, t.wt[0]))));
          const float v3 = __fmaf_rn(se.w, t.wt[3], __fmaf_rn(sw.w, t.wt[2], __fmaf_rn(ne.w, t.wt[1], __fmul_rn(nw.w, t.wt[0]))));
          const float sim0 = __fmaf_rn(softmax2_p0(v0, v1), r[0], r[1]);
          const float sim1 = __fmaf_rn(softmax2_p0(v2, v3), r[2], r[3]);
          const float tt = pixel_sum<LPP>(__fmaf_rn(cw0, sim0, cw1 * sim1));
          if (PASS == kStats) {
            if (owner) { s1 += tt; s2 = fmaf(tt, tt, s2); }
          } else {
            const size_t vox = ((size_t)b * p.D + d0 + dd) * hw + pix;
            const float2 dc = *reinterpret_cast<const float2*>(p.dcost + vox * G + 2 * sub);
            const float2 co = *reinterpret_cast<const float2*>(p.cost + vox * G + 2 * sub);
            const float dn = p.wsum[vox];
            const float dDn = -pixel_sum<LPP>(__fmaf_rn(dc.x, co.x, dc.y * co.y)) / dn;
            const float z = __fmaf_rn(tt, vpar[4 * v], vpar[4 * v + 1]);
            const float rl = fmaxf(z, 0.0f);
            const float u = __fmaf_rn(rl, w2, b2);
            const float wv = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-u * kLog2e));
            const float dwv = pixel_sum<LPP>(__fmaf_rn(dc.x / dn, sim0, (dc.y / dn) * sim1)) + dDn;
            const float du = dwv * wv * (1.0f - wv);
            const float dz = (z > 0.0f) ? du * w2 : 0.0f;
            const float xh = (tt - vpar[4 * v + 2]) * vpar[4 * v + 3];
            if (owner) {
              s1 += dz; s2 = fmaf(dz, xh, s2); a3 = fmaf(du, rl, a3); a4 += du;
              // the per-sample scalars the scatter pass needs: with them it has no reduction over a pixel's channels left
              p.aux[(size_t)v * ((size_t)p.B * p.D * hw) + vox] = make_float4(wv, dz, tt, 0.f);
            }
          }
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) { part[wave][2 * v] = s1; part[wave][2 * v + 1] = s2; }
      }
      if (PASS == kBwdReduce) {
        a3 = wave_sum(a3);
        a4 = wave_sum(a4);
        if (lane == 0) { part[wave][2 * p.n_src] = a3; part[wave][2 * p.n_src + 1] = a4; }
      }
    } else {
      for (int dd = 0; dd < nd; ++dd) {
        const int d = d0 + dd;
        const size_t vox = ((size_t)b * p.D + d) * hw + pix;
        float acc0 = 0.f, acc1 = 0.f, wsum = 0.f;
        for (int v = 0; v < p.n_src; ++v) {
          const TapEntry t = tab[(dd * p.n_src + v) * PPB + pl];
          const char* sb = reinterpret_cast<const char*>(p.src[v] + (size_t)b * map_stride);
          const float4 nw = *reinterpret_cast<const float4*>(sb + ((unsigned)t.off[0] * 4u + lane_b));
          const float4 ne = *reinterpret_cast<const float4*>(sb + ((unsigned)t.off[1] * 4u + lane_b));
          const float4 sw = *reinterpret_cast<const float4*>(sb + ((unsigned)t.off[2] * 4u + lane_b));
          const float4 se = *reinterpret_cast<const float4*>(sb + ((unsigned)t.off[3] * 4u + lane_b));
          const float v0 = __fmaf_rn(se.x, t.wt[3], __fmaf_rn(sw.x, t.wt[2], __fmaf_rn(ne.x, t.wt[1], __fmul_rn(nw.x, t.wt[0]))));
          const float v1 = __fmaf_rn(se.y, t.wt[3], __fmaf_rn(sw.y, t.wt[2], __fmaf_rn(ne.y, t.wt[1], __fmul_rn(nw.y, t.wt[0]))));
          const float v2 = __fmaf_rn(se.z, t.wt[3], __fmaf_rn(sw.z, t.wt[2], __fmaf_rn(ne.z, t.wt[1], __fmul_rn(nw.z, t.wt[0]))));
          const float v3 = __fmaf_rn(se.w, t.wt[3], __fmaf_rn(sw.w, t.wt[2], __fmaf_rn(ne.w, t.wt[1], __fmul_rn(nw.w, t.wt[0]))));
          const float q0 = softmax2_p0(v0, v1), q1 = softmax2_p0(v2, v3);
          const float sim0 = __fmaf_rn(q0, r[0], r[1]);
          const float sim1 = __fmaf_rn(q1, r[2], r[3]);
          const float tt = pixel_sum<LPP>(__fmaf_rn(cw0, sim0, cw1 * sim1));          // Conv3d(G->1, 1x1x1)
          const float z = __fmaf_rn(tt, vpar[4 * v], vpar[4 * v + 1]);                 // BatchNorm3d(1), batch statistics
          const float rl = fmaxf(z, 0.0f);
          const float u = __fmaf_rn(rl, w2, b2);
          const float wv = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-u * kLog2e));
          if (PASS == kFwd) {
            wsum += wv;
            acc0 += wv * sim0;
            acc1 += wv * sim1;
          }
        }
        if (PASS == kFwd && live) {
          *reinterpret_cast<float2*>(p.cost + vox * G + 2 * sub) = make_float2(acc0 / wsum, acc1 / wsum);
          if (sub == 0) p.wsum[vox] = wsum;
        }
      }
    }
    __syncthreads();
    if ((PASS == kStats || PASS == kBwdReduce) && tid < nred)   // (the next chunk's writes come after its phase-A barrier)
      total += (double)part[0][tid] + (double)part[1][tid] + (double)part[2][tid] + (double)part[3][tid];
  }

  if ((PASS == kStats || PASS == kBwdReduce) && tid < nred) atomicAdd(&p.red_out[tid], total);
}

// ------------------------------------------------------------------------------------------------------------------
// kBwd: the scatter.  The source-feature gradient of a softmax PAIR is antisymmetric (d v1 = -d v0), so only the even
// channel of every group is accumulated ([B,h,w,G] buffers; the caller expands to (g, -g)).  Per (depth chunk, view)
// the block finds the bounding box of its live taps in the source map; when it fits the LDS window the taps are
// accumulated there with LDS atomics and the window is flushed once with DENSE global atomics (whole rows of the
// window are contiguous in the NHWC gradient map) -- each texel is sent to memory once per chunk instead of once per
// tap; a block whose footprint does not fit (strong rotation, very wide depth range) scatters to memory directly.
#ifndef MDF_BWD_WIN_FLOATS
#define MDF_BWD_WIN_FLOATS 4096
#endif
constexpr int kWinFloats = MDF_BWD_WIN_FLOATS;   // 16 KiB: with the tap table (16-32 KiB) four blocks per CU

// LDS float add through an address-space-3 pointer: `ds_add_f32` (with a generic pointer next to the global fallback the
// compiler merges both branches into one `flat_atomic_add_f32` on a selected 64-bit address).
__device__ __forceinline__ void lds_add(float* p, float v) {
  (void)__hip_atomic_fetch_add((__attribute__((address_space(3))) float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Thread mapping of the scatter: the four waves of a block split the CHANNELS of the tile's pixels (wave w owns channels
// [w*C/4, (w+1)*C/4) of all PPB pixels), so two waves never touch the same (texel, channel) of the window, and a lane needs
// nothing from the other channels of its pixel: the per-sample scalars (view weight, dz, t) come from pass 2 (`aux`).
// That leaves one hazard for updating the window with plain read-add-write instead of LDS float atomics -- which retire about
// one LANE per clock per CU (ablation at cfg3: 0.69 of this kernel's 1.27 ms per step were those adds; the same updates as plain
// ds_read/ds_write cost 0.11 ms): two pixels of ONE wave instruction on the same texel.  For unclamped corners (xb = xa+1,
// yb = ya+1) the texel of tap k is (xa,ya) + offset_k, so two pixels collide on some tap iff their (xa,ya) agree.  Every
// flushing pixel writes its id to the per-wave claim byte of its (xa,ya) window texel and reads it back (LDS executes a wave's
// instructions in order): the pixel whose id survived adds non-atomically; the others on that texel and every pixel with
// clamped corners use atomic adds, in a separate basic block.
template <int C>
struct BwdTile {      // pixels of a scatter tile: 16 x 4 (C = 16), 8 x 4 (C = 32), 4 x 4 (C = 64)
  static constexpr int PPB = kThreads / (C / 4);
  static constexpr int TH = 4, TW = PPB / TH;
};

template <int C>
__global__ __launch_bounds__(kThreads) void warp_bwd_kernel(const TrainParams p) {
  constexpr int LPP = C / 4;            // lanes per pixel over the whole block
  constexpr int PPB = kThreads / LPP;   // pixels per tile = pixels per wave (every wave sees all of them)
  constexpr int LW = LPP / 4;           // lanes per pixel inside one wave (4 channels each)
  constexpr int G = C / 2;
  constexpr int TW = BwdTile<C>::TW, TH = BwdTile<C>::TH;
  static_assert(PPB * LW == 64, "a wave holds every pixel of the tile");
  static_assert(TW * TH == PPB, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TapXY* tab = reinterpret_cast<TapXY*>(smem);
  float* win = reinterpret_cast<float*>(smem + (size_t)p.dchunk * p.n_src * PPB * sizeof(TapXY));
  __shared__ int bb[MDF_MAX_SRC_VIEWS][4];   // xmin, xmax, ymin, ymax of the live taps of one view in this chunk
  __shared__ unsigned char claim[4][kWinFloats / 8];    // one id byte per window texel (G >= 8) and wave
  __shared__ float dcw_sm[kThreads][2];

  const int hw = p.g.h * p.g.w;
  const int W = p.g.w;
  const int b = blockIdx.y;
  const int tile = (int)mdf::xcd_remap(blockIdx.x, p.nblk_x);
  // a tile is BwdTile<C>::TW x TH pixels, not a run of a row: its taps' bounding box in a source map is (TW+1) x (TH+1) texels plus the
  // depth sweep instead of a slanted (PPB+1)-texel line's box -- half the window texels to zero and flush, and a window that fits
  const int tiles_x = (W + TW - 1) / TW;
  const int tile_y0 = (tile / tiles_x) * TH, tile_x0 = (tile % tiles_x) * TW;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int pl = lane / LW, sub = wave * LW + lane % LW;    // pixel of the tile, 4-channel slot of the pixel
  const bool live = (tile_x0 + pl % TW) < W && (tile_y0 + pl / TW) < p.g.h;
  const int pix = min(tile_y0 + pl / TW, p.g.h - 1) * W + min(tile_x0 + pl % TW, W - 1);
  volatile unsigned char* my_claim = claim[wave];

  float r[4];
  {
    const float4 rv = *reinterpret_cast<const float4*>(p.ref + ((size_t)b * hw + pix) * C + 4 * sub);
    softmax2(rv.x, rv.y, r[0], r[1]);
    softmax2(rv.z, rv.w, r[2], r[3]);
    r[0] -= r[1];
    r[2] -= r[3];
  }
  const float cw0 = p.par[2 * sub], cw1 = p.par[2 * sub + 1];
  const float gamma = p.par[G + 2], inv_n = p.par[G + 3];
  const float* vpar = p.par + G + 4;
  const size_t map_stride = (size_t)hw * C;
  const size_t gmap_stride = (size_t)hw * G;
  const size_t nvox = (size_t)p.B * p.D * hw;
  float gref0 = 0.f, gref1 = 0.f;     // d sim/d p0 accumulated over planes and views
  float dcw0 = 0.f, dcw1 = 0.f;       // d conv weight of this lane's two groups

  const int dlo = blockIdx.z * p.dslice, dhi = min(p.D, dlo + p.dslice);
  // Per-pixel hypotheses (stages 1, 2) keep a pixel's planes within a fraction of a source texel of each other: there the tap table
  // is built per VIEW over all the planes it can hold (the same LDS: dchunk * n_src planes), so a pixel's tap sums stay in registers
  // across them and the window is zeroed and flushed once per view instead of once per (chunk, view).  Uniform hypotheses (stage 0)
  // sweep the whole depth range -- a long epipolar segment -- and keep the short chunks whose footprint fits the window.
  const int nv = p.hypos_per_pixel ? 1 : p.n_src;                 // views per tap table
  const int dstep = p.hypos_per_pixel ? p.dchunk * p.n_src : p.dchunk;
  for (int v_lo = 0; v_lo < p.n_src; v_lo += nv)
  for (int d0 = dlo; d0 < dhi; d0 += dstep) {
    const int nd = min(dstep, dhi - d0);
    if (tid < 4 * p.n_src) bb[tid >> 2][tid & 3] = (tid & 1) ? INT32_MIN : INT32_MAX;
    __syncthreads();
    const int nent = nd * nv * PPB;
    for (int e = tid; e < nent; e += kThreads) {
      const int epl = e % PPB;
      const int ev = v_lo + (e / PPB) % nv;
      const int ed = e / (PPB * nv);
      const bool elive = (tile_x0 + epl % TW) < W && (tile_y0 + epl / TW) < p.g.h;
      const int yy = min(tile_y0 + epl / TW, p.g.h - 1), xx = min(tile_x0 + epl % TW, W - 1);
      const int epix = yy * W + xx;
      const float* m = p.proj + ((size_t)ev * p.B + b) * 12;
      const int d = d0 + ed;
      const float dep = p.hypos_per_pixel ? p.hypos[((size_t)b * p.D + d) * hw + epix] : p.hypos[(size_t)b * p.D + d];
      float ix, iy;
      warp_position(m, (float)xx, (float)yy, dep, p.g, ix, iy);
      TapXY t;
      tap_weights_corners(ix, iy, p.g, t.wt, t.xa, t.xb, t.ya, t.yb);
      tab[e] = t;
      if (elive) {
        const bool a = (t.wt[0] != 0.0f) || (t.wt[2] != 0.0f), bq = (t.wt[1] != 0.0f) || (t.wt[3] != 0.0f);   // column xa / xb live
        const bool cq = (t.wt[0] != 0.0f) || (t.wt[1] != 0.0f), dq = (t.wt[2] != 0.0f) || (t.wt[3] != 0.0f);  // row ya / yb live
        if (a || bq) {
          atomicMin(&bb[ev][0], a ? t.xa : t.xb);
          atomicMax(&bb[ev][1], bq ? t.xb : t.xa);
          atomicMin(&bb[ev][2], cq ? t.ya : t.yb);
          atomicMax(&bb[ev][3], dq ? t.yb : t.ya);
        }
      }
    }
    __syncthreads();

    for (int v = v_lo; v < v_lo + nv; ++v) {
      const int xmin = bb[v][0], xmax = bb[v][1], ymin = bb[v][2], ymax = bb[v][3];
      const int ww = xmax - xmin + 1, wh = ymax - ymin + 1;
      const bool any = (xmax >= xmin) && (ymax >= ymin);
      const bool use_win = any && ((long long)ww * wh * G <= kWinFloats);      // block-uniform
      if (use_win) {
        for (int i = tid; i < ww * wh * G; i += kThreads) win[i] = 0.0f;
        __syncthreads();
      }
      const char* sb = reinterpret_cast<const char*>(p.src[v] + (size_t)b * map_stride);     // uniform base + 32-bit lane offsets
      const unsigned lane_b = 16u * (unsigned)sub;
      float* gp = p.dsrc[v] + (size_t)b * gmap_stride + 2 * sub;
      const float4* ax = p.aux + (size_t)v * nvox;
      const float s1n = (float)p.red_in[2 * v] * inv_n, s2n = (float)p.red_in[2 * v + 1] * inv_n;
      const float mu = vpar[4 * v + 2], is = vpar[4 * v + 3];
      float pend0[4] = {0.f, 0.f, 0.f, 0.f}, pend1[4] = {0.f, 0.f, 0.f, 0.f};   // pending tap sums of the current corner set
      int cxa = -1, cxb = -1, cya = -1, cyb = -1;
      // Tap liveness comes from the WEIGHTS (a tap whose weight was non-zero for some pending plane), never from the pending
      // value: an out-of-bounds tap has weight 0 and lies outside the bounding box `bb` (built from the same weight test), but
      // 0 * (non-finite gradient) = NaN would pass a value test and index the window out of range.  grid_sample's backward
      // likewise adds nothing for out-of-bounds taps and propagates NaN through the in-bounds ones.
      unsigned lm = 0;
      auto flush_taps = [&](int xa, int xb, int ya, int yb) {
        bool plain = false;
        const unsigned m = lm;
        lm = 0;
        if (use_win && !(p.all_atomic & 1)) {
          const int slot = min(max((ya - ymin) * ww + (xa - xmin), 0), kWinFloats / 8 - 1);   // the window texel of (xa,ya): no false sharing
          my_claim[slot] = (unsigned char)pl;
          plain = (my_claim[slot] == (unsigned char)pl) && (xb == xa + 1) && (yb == ya + 1);
        }
        float a0[4], a1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { a0[k] = pend0[k]; a1[k] = pend1[k]; pend0[k] = 0.f; pend1[k] = 0.f; }
        if (plain) {          // this pixel owns its texels among the pixels flushing in this instruction
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (m & (1u << k)) {      // zero-weight (out-of-bounds) taps never leave the registers
              const int tx = (k & 1) ? xb : xa, ty = (k & 2) ? yb : ya;
              float2* o = reinterpret_cast<float2*>(win + ((ty - ymin) * ww + (tx - xmin)) * G + 2 * sub);
              float2 cur = *o;
              cur.x += a0[k]; cur.y += a1[k];
              *o = cur;
            }
            asm volatile("" ::: "memory");   // tap k's write stays ahead of tap k+1's read (a neighbour's tap k+1 may be this texel)
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (m & (1u << k)) {
              const int tx = (k & 1) ? xb : xa, ty = (k & 2) ? yb : ya;
              if (use_win) {
                float* o = win + ((ty - ymin) * ww + (tx - xmin)) * G + 2 * sub;
                lds_add(o, a0[k]);
                lds_add(o + 1, a1[k]);
              } else {
                float* o = gp + (size_t)(ty * W + tx) * G;
                unsafeAtomicAdd(o, a0[k]);
                unsafeAtomicAdd(o + 1, a1[k]);
              }
            }
          }
        }
      };
      for (int dd = 0; dd < nd; ++dd) {
        const TapXY t = tab[(dd * nv + (v - v_lo)) * PPB + pl];
        const int o0 = (t.ya * W + t.xa), o1 = (t.ya * W + t.xb), o2 = (t.yb * W + t.xa), o3 = (t.yb * W + t.xb);
        const float4 nw = *reinterpret_cast<const float4*>(sb + ((unsigned)o0 * (4u * C) + lane_b));
        const float4 ne = *reinterpret_cast<const float4*>(sb + ((unsigned)o1 * (4u * C) + lane_b));
        const float4 sw = *reinterpret_cast<const float4*>(sb + ((unsigned)o2 * (4u * C) + lane_b));
        const float4 se = *reinterpret_cast<const float4*>(sb + ((unsigned)o3 * (4u * C) + lane_b));
        const size_t vox = ((size_t)b * p.D + d0 + dd) * hw + pix;
        const float4 sc = ax[vox];                                   // (w_v, dz_v, t_v, -)
        const float2 dc = *reinterpret_cast<const float2*>(p.dcost + vox * G + 2 * sub);
        const float dn = p.wsum[vox];
        const float v0 = __fmaf_rn(se.x, t.wt[3], __fmaf_rn(sw.x, t.wt[2], __fmaf_rn(ne.x, t.wt[1], __fmul_rn(nw.x, t.wt[0]))));
        const float v1 = __fmaf_rn(se.y, t.wt[3], __fmaf_rn(sw.y, t.wt[2], __fmaf_rn(ne.y, t.wt[1], __fmul_rn(nw.y, t.wt[0]))));
        const float v2 = __fmaf_rn(se.z, t.wt[3], __fmaf_rn(sw.z, t.wt[2], __fmaf_rn(ne.z, t.wt[1], __fmul_rn(nw.z, t.wt[0]))));
        const float v3 = __fmaf_rn(se.w, t.wt[3], __fmaf_rn(sw.w, t.wt[2], __fmaf_rn(ne.w, t.wt[1], __fmul_rn(nw.w, t.wt[0]))));
        const float q0 = softmax2_p0(v0, v1), q1 = softmax2_p0(v2, v3);
        const float sim0 = __fmaf_rn(q0, r[0], r[1]);
        const float sim1 = __fmaf_rn(q1, r[2], r[3]);
        const float wv = sc.x, dz = sc.y, tt = sc.z;
        const float dN0 = dc.x / dn, dN1 = dc.y / dn;
        const float xh = (tt - mu) * is;
        const float dt = gamma * is * (dz - s1n - xh * s2n);
        const float ds0 = __fmaf_rn(dN0, wv, dt * cw0);
        const float ds1 = __fmaf_rn(dN1, wv, dt * cw1);
        if (live) {
          dcw0 = fmaf(dt, sim0, dcw0);
          dcw1 = fmaf(dt, sim1, dcw1);
          gref0 = fmaf(ds0, 2.0f * q0 - 1.0f, gref0);       // d sim / d p0 = 2 q0 - 1
          gref1 = fmaf(ds1, 2.0f * q1 - 1.0f, gref1);
          const float g0 = ds0 * r[0] * q0 * (1.0f - q0);   // d sim / d q0 = p0 - p1; softmax pair: d v1 = -d v0
          const float g1 = ds1 * r[2] * q1 * (1.0f - q1);
          // neighbouring planes of a pixel mostly hit the SAME four texels (per-pixel hypotheses span a fraction of a
          // source pixel per plane): keep the tap sums in registers while the corners do not move, send them on when they do
          if (t.xa != cxa || t.ya != cya || t.xb != cxb || t.yb != cyb) {
            flush_taps(cxa, cxb, cya, cyb);
            cxa = t.xa; cxb = t.xb; cya = t.ya; cyb = t.yb;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            pend0[k] = fmaf(t.wt[k], g0, pend0[k]);
            pend1[k] = fmaf(t.wt[k], g1, pend1[k]);
            lm |= (t.wt[k] != 0.0f) ? (1u << k) : 0u;      // NaN weights (z == 0 planes) count as live, as in `bb`
          }
        }
      }
      flush_taps(cxa, cxb, cya, cyb);
      if (use_win) {
        __syncthreads();
        float* gv = p.dsrc[v] + (size_t)b * gmap_stride;
        for (int wy = 0; wy < wh; ++wy) {
          float* grow = gv + ((size_t)(ymin + wy) * W + xmin) * G;     // ww*G contiguous floats
          const float* wrow = win + wy * ww * G;
          for (int j = tid; j < ww * G; j += kThreads) {
            const float val = wrow[j];
            if (val != 0.0f && !(p.all_atomic & 2)) unsafeAtomicAdd(grow + j, val);     // (bit 1: timing experiment, flush dropped)
          }
        }
        __syncthreads();
      }
    }
    __syncthreads();   // the next chunk overwrites tab and bb
  }

  // d ref: sim = p1 + q0*(p0 - p1) with (p0,p1) = softmax(a0,a1): d a0 = gref * p0*p1, d a1 = -d a0
  if (live) {
    const float p1a = r[1], p0a = 1.0f - r[1], p1b = r[3], p0b = 1.0f - r[3];
    const float ga = gref0 * p0a * p1a, gb = gref1 * p0b * p1b;
    float* o = p.dref + ((size_t)b * hw + pix) * C + 4 * sub;      // zero-initialised: the depth slices of a pixel meet here
    if (gridDim.z == 1) {
      *reinterpret_cast<float4*>(o) = make_float4(ga, -ga, gb, -gb);
    } else {
      unsafeAtomicAdd(o, ga); unsafeAtomicAdd(o + 1, -ga); unsafeAtomicAdd(o + 2, gb); unsafeAtomicAdd(o + 3, -gb);
    }
  }
  dcw_sm[tid][0] = dcw0;
  dcw_sm[tid][1] = dcw1;
  __syncthreads();
  if (tid < G) {   // group tid lives in 4-channel slot tid/2, i.e. wave (tid/2)/LW, lanes pl*LW + (tid/2)%LW
    const int sl = tid >> 1, wv_ = sl / LW, ln = sl % LW;
    float sacc = 0.f;
    for (int qd = 0; qd < PPB; ++qd) sacc += dcw_sm[wv_ * 64 + qd * LW + ln][tid & 1];
    unsafeAtomicAdd(&p.dcw[tid], sacc);
  }
}

// A block walks its pixels over the planes one (plane, view) at a time, each step a dependent gather: with the few blocks of a
// cfg3-sized map (432 at 72x96x64ch) the chip holds < 2 waves per SIMD and the kernel is latency-bound.  Cut the depth range
// into slices (gridDim.z) until there are a few thousand blocks; per-pixel results that span the planes (d ref) meet through atomics.
int depth_slices(TrainParams& p, int& dch, int target) {
  const long long blocks = (long long)p.nblk_x * p.B;
  int nz = (int)((target + blocks - 1) / blocks);
  if (nz > p.D / 4) nz = p.D / 4;          // >= 4 planes per slice: a slice re-reads the reference features and the tap setup
  if (nz < 1) nz = 1;
  p.dslice = (p.D + nz - 1) / nz;
  nz = (p.D + p.dslice - 1) / p.dslice;
  if (dch > p.dslice) dch = p.dslice;
  const int nch = (p.dslice + dch - 1) / dch;   // equal chunks inside a slice
  dch = (p.dslice + nch - 1) / nch;
  return nz;
}

int launch_bwd(TrainParams& p, int C, hipStream_t st) {
  const int lpp = C / 4, ppb = kThreads / lpp;
  const int th = 4, tw = ppb / th;            // BwdTile<C>
  p.nblk_x = ((p.g.w + tw - 1) / tw) * ((p.g.h + th - 1) / th);
  // tap-table entries per block: uniform hypotheses (stage 0) sweep an epipolar segment and amortise a view's window over more
  // planes; per-pixel hypotheses take a whole view's planes anyway
  static const int tab_env = [] { const char* e = getenv("MDF_WARP_BWD_TAB"); return (e && atoi(e) > 0) ? atoi(e) : 0; }();   // dev A/B
  const int tab_entries = tab_env ? tab_env : (p.hypos_per_pixel ? 512 : 1024);
  int dch = tab_entries / (p.n_src * ppb);
  if (dch < 1) dch = 1;
  // depth slices of the scatter: with per-pixel hypotheses a pixel's planes hit the same few texels, and their tap sums stay in
  // registers across ALL the planes a block walks -- slicing the depth range costs more than the extra blocks bring (cfg3 stage 2:
  // 178 us unsliced, 284 in two slices); uniform hypotheses (stage 0, 432 blocks at cfg3) take three slices (162 us; 192 in two)
  static const int target_env = [] { const char* e = getenv("MDF_WARP_BWD_BLOCKS"); return (e && atoi(e) > 0) ? atoi(e) : 0; }();   // dev A/B
  const int nz = depth_slices(p, dch, target_env ? target_env : (p.hypos_per_pixel ? 768 : 1024));
  p.dchunk = dch;
  const char* dbg = getenv("MDF_WARP_BWD_ATOMIC");      // read per call: tests flip it inside one process
  p.all_atomic = (dbg && atoi(dbg) > 0) ? atoi(dbg) : 0;     // bit 0: all-atomic window updates; bit 1: timing experiment without the flush
  const size_t lds = (size_t)dch * p.n_src * ppb * sizeof(TapXY) + (size_t)kWinFloats * sizeof(float);
  dim3 grid(p.nblk_x, p.B, nz), block(kThreads);
  switch (C) {
    case 64: hipLaunchKernelGGL((warp_bwd_kernel<64>), grid, block, lds, st, p); break;
    case 32: hipLaunchKernelGGL((warp_bwd_kernel<32>), grid, block, lds, st, p); break;
    case 16: hipLaunchKernelGGL((warp_bwd_kernel<16>), grid, block, lds, st, p); break;
    default: return mdf::fail(MDF_EUNSUPPORTED, "warp kernels are built for C in {16,32,64}, got %d", C);
  }
  return mdf::check_launch("warp_bwd_kernel");
}

template <int PASS>
int launch_train(TrainParams& p, int C, hipStream_t st) {
  const int lpp = C / 4, ppb = kThreads / lpp;
  const int hw = p.g.h * p.g.w;
  p.nblk_x = (hw + ppb - 1) / ppb;
  static const int tab_env = [] { const char* e = getenv("MDF_WARP_TRAIN_TAB"); return (e && atoi(e) > 0) ? atoi(e) : 512; }();   // dev A/B
  int dch = tab_env / (p.n_src * ppb);
  if (dch < 1) dch = 1;
  static const int target_env = [] { const char* e = getenv("MDF_WARP_TRAIN_BLOCKS"); return (e && atoi(e) > 0) ? atoi(e) : 0; }();   // dev A/B
  const int nz = depth_slices(p, dch, target_env ? target_env : (p.hypos_per_pixel ? 1024 : 2048));
  p.dchunk = dch;
  const size_t lds = (size_t)dch * p.n_src * ppb * sizeof(TapEntry);
  dim3 grid(p.nblk_x, p.B, nz), block(kThreads);
  switch (C) {
    case 64: hipLaunchKernelGGL((warp_train_kernel<64, PASS>), grid, block, lds, st, p); break;
    case 32: hipLaunchKernelGGL((warp_train_kernel<32, PASS>), grid, block, lds, st, p); break;
    case 16: hipLaunchKernelGGL((warp_train_kernel<16, PASS>), grid, block, lds, st, p); break;
    default: return mdf::fail(MDF_EUNSUPPORTED, "warp kernels are built for C in {16,32,64}, got %d", C);
  }
  return mdf::check_launch("warp_train_kernel");
}

// ---- control plane of the training aggregate: the handful of scalars between the passes, computed where they live (as
// torch expressions they were ~35 tiny launches per stage and direction, on a step that is bound by its host's issue rate)
__global__ void agg_prepare_kernel(const float* cw, const float* w2, const float* b2, const float* gamma, float inv_n, int G, int n_src,
                                   float* par, double* red, int nred) {
  const int i = threadIdx.x;
  if (i < G) par[i] = cw[i];
  if (i == 0) { par[G] = w2[0]; par[G + 1] = b2[0]; par[G + 2] = gamma[0]; par[G + 3] = inv_n; }
  if (i < 4 * n_src) par[G + 4 + i] = 0.f;
  if (i < nred) red[i] = 0.0;
}

// batch statistics of the 1-channel BatchNorm per source view -> (alpha, shift, mean, invstd), and the running statistics
// advanced as by n_src successive module calls (homoaggregate.py:35-40 calls depth_weight once per source view)
__global__ void agg_finalize_kernel(const double* red, const float* gamma, const float* beta, double eps, double momentum, double n, int G,
                                    int n_src, float* par, float* rmean, float* rvar, long long* nbt) {
  if (threadIdx.x != 0) return;
  double rm = rmean ? (double)rmean[0] : 0.0, rv = rvar ? (double)rvar[0] : 0.0;
  for (int v = 0; v < n_src; ++v) {
    const double mean = red[2 * v] / n;
    double var = red[2 * v + 1] / n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const double invstd = 1.0 / sqrt(var + eps);
    const double alpha = (double)gamma[0] * invstd;
    float* o = par + G + 4 + 4 * v;
    o[0] = (float)alpha; o[1] = (float)((double)beta[0] - mean * alpha); o[2] = (float)mean; o[3] = (float)invstd;
    rm = (1.0 - momentum) * rm + momentum * mean;
    rv = (1.0 - momentum) * rv + momentum * var * (n / (n > 1.0 ? n - 1.0 : 1.0));
  }
  if (rmean) rmean[0] = (float)rm;
  if (rvar) rvar[0] = (float)rv;
  if (nbt) nbt[0] += n_src;
}

// backward epilogue: the scatter accumulated the gradient of the EVEN channel of every softmax pair; the odd channel gets
// its negative (d softmax pair: dv1 = -dv0).  dpar = (d gamma, d beta, d w2, d b2) from the fp64 reductions.
__global__ void agg_bwd_finalize_kernel(const float* __restrict__ dhalf, const double* __restrict__ red, int n_src, long long n_half,
                                        float* __restrict__ dfull, float* __restrict__ dpar, const float* __restrict__ dref_acc,
                                        float* __restrict__ dref_out, long long n_ref4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_half) {
    const float v = dhalf[i];
    reinterpret_cast<float2*>(dfull)[i] = make_float2(v, -v);
  }
  if (dref_acc && i < n_ref4) reinterpret_cast<float4*>(dref_out)[i] = reinterpret_cast<const float4*>(dref_acc)[i];
  if (i == 0) {
    double dg = 0.0, db = 0.0;
    for (int v = 0; v < n_src; ++v) { db += red[2 * v]; dg += red[2 * v + 1]; }
    dpar[0] = (float)dg; dpar[1] = (float)db; dpar[2] = (float)red[2 * n_src]; dpar[3] = (float)red[2 * n_src + 1];
  }
}

}  // namespace

extern "C" int mdf_warp_aggregate_vec_train(int pass, const float* ref_fea, const float* const* src_feas, const float* proj,
                                            const float* hypos, int hypos_per_pixel, const float* par, const double* red_in,
                                            const float* dcost, float* cost, float* wsum, double* red_out, float* dref,
                                            float* const* dsrc, float* dcw, float* aux, int B, int C, int G, int D, int h, int w,
                                            int n_src, void* stream) {
  MDF_REQUIRE(ref_fea && src_feas && proj && hypos && par, "null pointer argument");
  MDF_REQUIRE(pass >= 0 && pass <= 3, "pass=%d not in 0..3", pass);
  MDF_REQUIRE(B > 0 && D > 0 && h > 1 && w > 1, "bad shape B=%d D=%d h=%d w=%d", B, D, h, w);
  MDF_REQUIRE((long long)h * w * C < (1ll << 30), "feature map too large for 32-bit byte offsets");
  MDF_REQUIRE(n_src >= 1 && n_src <= MDF_MAX_SRC_VIEWS, "n_src=%d out of range [1,%d]", n_src, MDF_MAX_SRC_VIEWS);
  if (G * 2 != C) return mdf::fail(MDF_EUNSUPPORTED, "only C/G == 2 is built (C=%d, G=%d)", C, G);
  TrainParams p{};
  p.ref = ref_fea;
  for (int v = 0; v < n_src; ++v) {
    MDF_REQUIRE(src_feas[v], "src_feas[%d] is null", v);
    p.src[v] = src_feas[v];
  }
  p.proj = proj; p.hypos = hypos; p.par = par; p.red_in = red_in; p.dcost = dcost; p.cost = cost; p.wsum = wsum;
  p.red_out = red_out; p.dref = dref; p.dcw = dcw; p.aux = reinterpret_cast<float4*>(aux);
  p.g = make_geom(h, w);
  p.B = B; p.D = D; p.n_src = n_src; p.hypos_per_pixel = hypos_per_pixel;
  hipStream_t st = (hipStream_t)stream;
  switch (pass) {
    case kStats:
      MDF_REQUIRE(red_out, "stats pass needs red_out");
      return launch_train<kStats>(p, C, st);
    case kFwd:
      MDF_REQUIRE(cost && wsum, "forward pass needs cost and wsum");
      return launch_train<kFwd>(p, C, st);
    case kBwdReduce:
      MDF_REQUIRE(dcost && cost && wsum && red_out && aux, "backward-reduce pass needs dcost, cost, wsum, red_out, aux");
      return launch_train<kBwdReduce>(p, C, st);
    default:
      MDF_REQUIRE(dcost && wsum && red_in && dref && dsrc && dcw && aux, "backward pass needs dcost, wsum, red_in, dref, dsrc, dcw, aux");
      for (int v = 0; v < n_src; ++v) {
        MDF_REQUIRE(dsrc[v], "dsrc[%d] is null", v);
        p.dsrc[v] = dsrc[v];
      }
      return launch_bwd(p, C, st);
  }
}

extern "C" int mdf_aggregate_train_prepare(const float* cw, const float* w2, const float* b2, const float* gamma, long long n, int G,
                                           int n_src, float* par, double* red, int nred, void* stream) {
  MDF_REQUIRE(cw && w2 && b2 && gamma && par && red, "null pointer argument");
  MDF_REQUIRE(n > 0 && G >= 1 && G <= 32 && n_src >= 1 && n_src <= MDF_MAX_SRC_VIEWS && nred >= 0 && nred <= 64, "bad sizes");
  hipLaunchKernelGGL(agg_prepare_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cw, w2, b2, gamma, (float)(1.0 / (double)n), G, n_src, par,
                     red, nred);
  return mdf::check_launch("agg_prepare_kernel");
}

extern "C" int mdf_aggregate_train_finalize(const double* red, const float* gamma, const float* beta, float eps, float momentum, long long n,
                                            int G, int n_src, float* par, float* running_mean, float* running_var,
                                            long long* num_batches_tracked, void* stream) {
  MDF_REQUIRE(red && gamma && beta && par, "null pointer argument");
  MDF_REQUIRE(n > 0 && G >= 1 && n_src >= 1 && n_src <= MDF_MAX_SRC_VIEWS, "bad sizes");
  hipLaunchKernelGGL(agg_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, red, gamma, beta, (double)eps, (double)momentum, (double)n,
                     G, n_src, par, running_mean, running_var, num_batches_tracked);
  return mdf::check_launch("agg_finalize_kernel");
}

extern "C" int mdf_aggregate_train_bwd_finalize(const float* dhalf, const double* red, int n_src, long long n_half, float* dsrc,
                                                float* dpar, const float* dref_acc, float* dref, long long n_ref, void* stream) {
  MDF_REQUIRE(dhalf && red && dsrc && dpar, "null pointer argument");
  MDF_REQUIRE(n_half > 0 && n_src >= 1 && n_src <= MDF_MAX_SRC_VIEWS, "bad sizes");
  MDF_REQUIRE((dref_acc == nullptr) == (dref == nullptr) && (dref_acc == nullptr || (n_ref > 0 && n_ref % 4 == 0)), "dref_acc / dref / n_ref mismatch");
  const long long n_ref4 = dref_acc ? n_ref / 4 : 0;
  const long long blocks = ((n_half > n_ref4 ? n_half : n_ref4) + 255) / 256;
  MDF_REQUIRE(blocks < (1ll << 31), "too many elements");
  hipLaunchKernelGGL(agg_bwd_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dhalf, red, n_src, n_half, dsrc, dpar,
                     dref_acc, dref, n_ref4);
  return mdf::check_launch("agg_bwd_finalize_kernel");
}
