// Depth-map geometric/photometric consistency filter + fusion: replaces the per-source-view chain of
// ~40 tensor passes in tools/filter/dynamic_filter_gpu.py (`reproject_with_depth` :194-238, `check_geometric_consistency`
// :166-191, fusion in `filter` :63-103) by ONE kernel: a thread owns a reference pixel, walks the source views, and
// keeps the nine dynamic-threshold counters, the valid-view count and the depth sum in registers.  HBM-bound:
// reads (1 + n_src) depth maps + confidence (the source gathers hit L2), writes depth + 3 mask bytes.
//
// Arithmetic order == torch CPU (verified bit-exact against goldens produced by the reference's functions):
// every [3x3]/[4x4] x [k,N] matmul is the forward chain acc = m0*x0; acc = fma(m_j, x_j, acc); pixel indices are
// converted to float before the multiply by depth; divisions are true divides; bilinear_sampler is
// 2*x/(W-1)-1 -> grid_sample(align_corners=True, zeros): ix = (g+1)*((W-1)/2), taps nw,ne,sw,se as one fma chain.
#include "common.h"

namespace {

struct ViewMats {   // per source view, row-major
  float kr_inv[9];  // inverse(K_ref)
  float t_rs[16];   // E_src @ inverse(E_ref)
  float k_s[9];     // K_src
  float ks_inv[9];  // inverse(K_src)
  float t_sr[16];   // E_ref @ inverse(E_src)
  float k_r[9];     // K_ref
};
static_assert(sizeof(ViewMats) == 68 * sizeof(float), "ViewMats layout");

template <typename P>   // P: pointer to the nine floats (constant address space in the kernel: scalar loads)
__device__ __forceinline__ void mm3(P m, float x0, float x1, float x2, float& o0, float& o1, float& o2) {
  o0 = __fmaf_rn(m[2], x2, __fmaf_rn(m[1], x1, __fmul_rn(m[0], x0)));
  o1 = __fmaf_rn(m[5], x2, __fmaf_rn(m[4], x1, __fmul_rn(m[3], x0)));
  o2 = __fmaf_rn(m[8], x2, __fmaf_rn(m[7], x1, __fmul_rn(m[6], x0)));
}
// rows 0..2 of a 4x4 times [x0,x1,x2,1]
template <typename P>
__device__ __forceinline__ void mm4(P m, float x0, float x1, float x2, float& o0, float& o1, float& o2) {
  o0 = __fmaf_rn(m[3], 1.0f, __fmaf_rn(m[2], x2, __fmaf_rn(m[1], x1, __fmul_rn(m[0], x0))));
  o1 = __fmaf_rn(m[7], 1.0f, __fmaf_rn(m[6], x2, __fmaf_rn(m[5], x1, __fmul_rn(m[4], x0))));
  o2 = __fmaf_rn(m[11], 1.0f, __fmaf_rn(m[10], x2, __fmaf_rn(m[9], x1, __fmul_rn(m[8], x0))));
}

struct FuseParams {
  const float* depth_ref;
  const float* conf;
  const float* src[MDF_MAX_SRC_VIEWS];
  const ViewMats* mats;
  float* depth_avg;
  unsigned char* masks;       // [3][h][w]: photo, geo, final
  unsigned short* view_masks; // optional [n_src][h][w], bit i <-> threshold index i (i = 0..8 for 2..10)
  float* rep_out;             // optional [n_src][h][w]: depth_reprojected (zeroed outside the last mask)
  float thr_dist[9], thr_rel[9];
  float photo_threshold, half_w, half_h;
  int n_src, h, w, nconditions;
};

// Uniform (per-view) data is read through the CONSTANT address space: the matrices live in a device buffer the kernel never
// writes, but behind a plain pointer the compiler cannot know that and fetched all 68 floats per view with 14 vector loads per
// lane -- 14 KB through the texture path per wave and view, which (not HBM, not the ALUs) was what the kernel waited for.  As
// addrspace(4) loads with a uniform address they are s_load_dwordx16 into SGPRs.
typedef const __attribute__((address_space(4))) float* cfloat_p;

struct Taps {                 // stage A of one source view: where the reference pixel lands + the four source depths around it
  float xs, ys;               // source pixel coordinates (:214)
  float w_nw, w_ne, w_sw, w_se;
  float l_nw, l_ne, l_sw, l_se;
  bool in_nw, in_ne, in_sw, in_se;   // tap inside the map (carried over the loop edge as lane masks in SGPRs, not as data)
};

__device__ __forceinline__ Taps project(cfloat_p M, const float* __restrict__ ds, float xf, float yf, float d, float mw, float mh,
                                        float half_w, float half_h, int w) {
  Taps t;
  // reference pixel -> reference camera -> source camera -> source pixel      (:205-214)
  float c0, c1, c2, s0, s1, s2, q0, q1, q2;
  mm3(M + 0, __fmul_rn(xf, d), __fmul_rn(yf, d), d, c0, c1, c2);
  mm4(M + 9, c0, c1, c2, s0, s1, s2);
  mm3(M + 25, s0, s1, s2, q0, q1, q2);
  t.xs = __fdiv_rn(q0, q2);
  t.ys = __fdiv_rn(q1, q2);
  // bilinear_sampler (data_io.py:117-131): pixel coords -> [-1,1] -> grid_sample(align_corners=True, zeros)
  const float gx = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, t.xs), mw), 1.0f);
  const float gy = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, t.ys), mh), 1.0f);
  const float ix = __fmul_rn(__fadd_rn(gx, 1.0f), half_w);
  const float iy = __fmul_rn(__fadd_rn(gy, 1.0f), half_h);
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float fw = __fsub_rn(ix, x0f), fe = __fsub_rn(1.0f, fw), fn = __fsub_rn(iy, y0f), fs = __fsub_rn(1.0f, fn);
  const float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
  const bool bx0 = (x0f >= 0.f) && (x0f <= mw), bx1 = (x1f >= 0.f) && (x1f <= mw);
  const bool by0 = (y0f >= 0.f) && (y0f <= mh), by1 = (y1f >= 0.f) && (y1f <= mh);
  t.in_nw = bx0 && by0; t.in_ne = bx1 && by0; t.in_sw = bx0 && by1; t.in_se = bx1 && by1;
  const int xa = (int)fminf(fmaxf(x0f, 0.f), mw), xb = (int)fminf(fmaxf(x1f, 0.f), mw);
  const int ya = (int)fminf(fmaxf(y0f, 0.f), mh), yb = (int)fminf(fmaxf(y1f, 0.f), mh);
  t.w_nw = __fmul_rn(fs, fe); t.w_ne = __fmul_rn(fs, fw); t.w_sw = __fmul_rn(fn, fe); t.w_se = __fmul_rn(fn, fw);
  // the four taps are loaded unconditionally (the corners are clamped into the map) and out-of-bounds ones replaced by 0 by the
  // consumer; 32-bit offsets from the map's (uniform) base
  const unsigned ra = (unsigned)(ya * w), rb = (unsigned)(yb * w);      // (clamped: 0 <= index < h*w < 2^31)
  t.l_nw = ds[ra + (unsigned)xa]; t.l_ne = ds[ra + (unsigned)xb]; t.l_sw = ds[rb + (unsigned)xa]; t.l_se = ds[rb + (unsigned)xb];
  return t;
}

// One thread owns a reference pixel and walks the source views in a two-stage software pipeline: the projection of view v + 1 and
// its four gathers are issued BEFORE the back-projection / divide chain of view v, so a gather's latency is covered by ~150
// vector instructions of the same wave instead of by other waves only.
__global__ __launch_bounds__(256) void consistency_fuse_kernel(const FuseParams p) {
  const int hw = p.h * p.w;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= hw) return;
  const int yi = pix / p.w, xi = pix - yi * p.w;
  const float xf = (float)xi, yf = (float)yi;
  const float d = p.depth_ref[pix];
  // The nine masks of a view are nested: thr_dist[i] and thr_rel[i] grow with i, so mask i implies mask i + 1 and a view is
  // described by k = the first index whose mask holds (9 = none; a NaN compares false everywhere, as in torch).  hist packs the
  // ten possible k into 5-bit bins (n_src <= 16): cnt[i] of the reference code = views with k <= i.
  unsigned long long hist = 0ull;
  int nvalid = 0;
  float acc = 0.f;
  const float mw = (float)(p.w - 1), mh = (float)(p.h - 1);
  const cfloat_p mats = (cfloat_p)(const float*)p.mats;
  Taps cur = project(mats, p.src[0], xf, yf, d, mw, mh, p.half_w, p.half_h, p.w);
  for (int v = 0; v < p.n_src; ++v) {
    const cfloat_p M = mats + 68 * v;
    Taps nxt = cur;
    if (v + 1 < p.n_src) nxt = project(M + 68, p.src[v + 1], xf, yf, d, mw, mh, p.half_w, p.half_h, p.w);
    // (the empty asm pins the loads: with a bare select the compiler sinks each load under its condition, i.e. four
    //  branch + wait pairs per view in a dependent chain)
    asm volatile("" : "+v"(cur.l_nw), "+v"(cur.l_ne), "+v"(cur.l_sw), "+v"(cur.l_se));
    const float t_nw = cur.in_nw ? cur.l_nw : 0.f, t_ne = cur.in_ne ? cur.l_ne : 0.f;
    const float t_sw = cur.in_sw ? cur.l_sw : 0.f, t_se = cur.in_se ? cur.l_se : 0.f;
    const float samp = __fmaf_rn(t_se, cur.w_se, __fmaf_rn(t_sw, cur.w_sw, __fmaf_rn(t_ne, cur.w_ne, __fmul_rn(t_nw, cur.w_nw))));
    // back to the reference view with the SAMPLED source depth                  (:226-236)
    float b0, b1, b2, r0, r1, r2, u0, u1, u2;
    mm3(M + 34, __fmul_rn(cur.xs, samp), __fmul_rn(cur.ys, samp), samp, b0, b1, b2);
    mm4(M + 43, b0, b1, b2, r0, r1, r2);
    mm3(M + 59, r0, r1, r2, u0, u1, u2);
    const float xr = __fdiv_rn(u0, u2), yr = __fdiv_rn(u1, u2);
    // :179-183
    const float dx = __fsub_rn(xr, xf), dy = __fsub_rn(yr, yf);
    const float dist = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    const float rel = __fdiv_rn(fabsf(__fsub_rn(r2, d)), d);
    int kd = 0, kr = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      kd += (dist < p.thr_dist[i]) ? 0 : 1;                             // NaN compares false, as in torch
      kr += (rel < p.thr_rel[i]) ? 0 : 1;
    }
    const int k = kd > kr ? kd : kr;                                    // masks k .. 8 hold
    hist += 1ull << (5 * k);
    const bool last = k <= 8;
    const float rep = last ? r2 : 0.f;                                  // depth_reprojected[~mask] = 0   (:189)
    nvalid += last ? 1 : 0;
    acc = __fadd_rn(acc, rep);
    if (p.view_masks) p.view_masks[(size_t)v * hw + pix] = (unsigned short)((0x1FFu << k) & 0x1FFu);
    if (p.rep_out) p.rep_out[(size_t)v * hw + pix] = rep;
    cur = nxt;
  }
  int geo = 0, run = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    run += (int)((hist >> (5 * i)) & 31ull);                            // = cnt[i]
    geo += (run >= i + 2) ? 1 : 0;                                      // :93-94
  }
  p.depth_avg[pix] = __fdiv_rn(__fadd_rn(acc, d), (float)(nvalid + 1));        // :98
  const bool geo_m = geo >= p.nconditions, photo_m = p.conf[pix] > p.photo_threshold;
  p.masks[pix] = photo_m;
  p.masks[hw + pix] = geo_m;
  p.masks[2 * hw + pix] = photo_m && geo_m;
}

}  // namespace

extern "C" int mdf_consistency_fuse_fwd(const float* depth_ref, const float* conf, const float* const* src_depths,
                                        const float* mats, int n_src, int h, int w, float photo_threshold, int nconditions,
                                        float thre1, float thre2, float* depth_avg, unsigned char* masks,
                                        unsigned short* view_masks, float* rep_out, void* stream) {
  MDF_REQUIRE(depth_ref && conf && src_depths && mats && depth_avg && masks, "null pointer argument");
  MDF_REQUIRE(n_src >= 1 && n_src <= MDF_MAX_SRC_VIEWS, "n_src=%d out of range [1,%d]", n_src, MDF_MAX_SRC_VIEWS);
  MDF_REQUIRE(h > 1 && w > 1 && thre1 > 0 && thre2 > 0, "bad shape / thresholds");
  FuseParams p{};
  p.depth_ref = depth_ref; p.conf = conf; p.mats = reinterpret_cast<const ViewMats*>(mats);
  for (int v = 0; v < n_src; ++v) {
    MDF_REQUIRE(src_depths[v], "src_depths[%d] is null", v);
    p.src[v] = src_depths[v];
  }
  p.depth_avg = depth_avg; p.masks = masks; p.view_masks = view_masks; p.rep_out = rep_out;
  for (int i = 0; i < 9; ++i) {            // python: i / thre  (double), compared against float32 tensors
    p.thr_dist[i] = (float)((double)(i + 2) / (double)thre1);
    p.thr_rel[i] = (float)((double)(i + 2) / (double)thre2);
  }
  p.photo_threshold = photo_threshold;
  p.half_w = (float)((double)(w - 1) / 2.0);
  p.half_h = (float)((double)(h - 1) / 2.0);
  p.n_src = n_src; p.h = h; p.w = w; p.nconditions = nconditions;
  hipLaunchKernelGGL(consistency_fuse_kernel, dim3((h * w + 255) / 256), dim3(256), 0, (hipStream_t)stream, p);
  return mdf::check_launch("consistency_fuse_kernel");
}
