// 3-D convolution layers of the cost regularisers (net/unit/regular.py:9-133, base.py:50-68) as an
// implicit GEMM on the fp32 matrix cores of gfx950 (v_mfma_f32_16x16x4_f32: exact f32 fma chain, so
// results differ from the reference only by summation order).
//
// Data layout: activations NDHWC (the aggregation kernel writes the cost volume in this layout), so the
// Cin values of one voxel are contiguous and one MFMA B-fragment (16 voxels x 16 cin) is a single
// coalesced 1-KiB wavefront load.  GEMM orientation is  D[cout][voxel] = W[cout][k] * X[k][voxel]:
//   A operand (weights)     lane l holds W[cout = l&15][k = l>>4]
//   B operand (activations) lane l holds X[k = l>>4][voxel = l&15]
//   C/D                     lane l holds couts 4*(l>>4)..+3 of voxel l&15  -> one 16-B store per lane,
//                           the 4 lanes of a voxel write 64 contiguous bytes (NDHWC output).
// K is walked tap by tap; inside a tap the k order is permuted so that lane group q = l>>4 owns cin
// {KPL*q .. KPL*q+KPL-1} of a chunk: a lane fetches its KPL consecutive cin with ONE 8/16-byte load and
// feeds them to KPL successive MFMAs.  Weights are pre-packed on the device into exactly that fragment
// order (mdf_conv3d_pack_weights), so an A fragment is also one coalesced load.
//
// Epilogue (fused): y = [res +] [relu]( acc * alpha + beta )   (BatchNorm3d folded as ATen does in eval).
//
// Modes: stride 1, stride 2, and ConvTranspose3d(k3,s2,p1,op1).  The transposed conv is computed per
// output parity class (blockIdx.y = 0..7): inside a class every output voxel uses the same 1/2/4/8 taps,
// so no MFMA is spent on structurally-zero taps.
#include <cstdlib>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum ConvMode { kS1 = 0, kS2 = 1, kTr = 2 };

struct ConvParams {
  const float* x;
  const float* wpack;
  const float* alpha;
  const float* beta;
  const float* res;
  float* y;
  int B, Di, Hi, Wi, Do, Ho, Wo;
  int relu;
  long long m_total;  // voxels in the M index space (per parity class for kTr)
  unsigned nblk;
  // training (ST kernels): per-channel sums of the OUTPUT ride in the epilogue -- one pass over the activations less per BatchNorm
  //   stat_mode 1: out[c] += sum y, out[C+c] += sum y^2 of the raw conv output (the layer's own batch statistics)
  //   stat_mode 2: the output is dz of the PRODUCING layer (input gradient + skip gradient): out[c] += sum dr,
  //                out[C+c] += sum dr*xhat with dr = dz*[y*a+b > 0], xhat = (y-mean)*invstd; stat_y = that layer's raw conv
  //                output (same shape as the output), stat_aux = its (a, b, mean, invstd)[C]
  int stat_mode;
  const float* stat_y;
  const float* stat_aux;
  double* stat_out;
  int stat_slices;        // slices of stat_out this launch spreads its blocks over (common.h: conv_stat_send)
  int kd_skip;            // shallow volumes (<= 3 input planes): skip the depth taps no voxel of a wave has (conv3d_kernel: kd_any)
};

// fp64 LDS add (ds_add_f64) / the DPP sum over the 16 lanes of an MFMA column group (lanes q*16 .. q*16+15 hold the 16
// voxels of one m-tile for the same 4 output channels)
__device__ __forceinline__ void lds_add_f64(double* p, double v) { atomicAdd(p, v); }
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov_f<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov_f<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov_f<0x141>(v);   // row_half_mirror
  v += dpp_mov_f<0x140>(v);   // row_mirror
  return v;
}

template <int KPL>
struct Frag;
template <>
struct Frag<4> {
  float v[4];
  __device__ __forceinline__ void load(const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ __forceinline__ void zero() { v[0] = v[1] = v[2] = v[3] = 0.f; }
  __device__ __forceinline__ void scale(float m) { v[0] *= m; v[1] *= m; v[2] *= m; v[3] *= m; }
};
template <>
struct Frag<2> {
  float v[2];
  __device__ __forceinline__ void load(const float* p) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  }
  __device__ __forceinline__ void zero() { v[0] = v[1] = 0.f; }
  __device__ __forceinline__ void scale(float m) { v[0] *= m; v[1] *= m; }
};

// SPLITK = 4: the four waves of a block share ONE wave tile and each takes every 4th tap; partial accumulators are summed
// through LDS.  For tiny volumes (a few thousand voxels, 27*64 deep K) this turns ~90 serial-latency-bound blocks into 4x
// as many quarter-length ones.
template <int CIN, int COUT, int MODE, int MT, int SPLITK, int ST = 0>
__global__ __launch_bounds__(256) void conv3d_kernel(const ConvParams p) {
  constexpr int KPL = (CIN >= 16) ? 4 : 2;   // k values per lane per chunk
  constexpr int CK = 4 * KPL;                // cin per chunk
  constexpr int NCH = CIN / CK;
  // transposed: GEMM rows are [pw][cout] (both output w-parities of an input voxel in one wave -> one contiguous
  // 2*COUT store per input voxel, and no padded rows for COUT = 8)
  constexpr int ROWS = (MODE == kTr) ? 2 * COUT : COUT;
  constexpr int NT = (ROWS + 15) / 16;
  static_assert(CIN % CK == 0, "CIN must be a multiple of the chunk");

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int q = lane >> 4;    // k sub-slot / cout quad
  const int n16 = lane & 15;  // voxel inside the m-tile (B operand, C/D column)
  // ST: block-level fp64 table of the epilogue sums [2][64] + the producing layer's (a, b, mean, invstd) [4][64].  Sending the
  // sums to memory costs 2*COUT same-address fp64 atomics per BLOCK (~10 ns each, serialised at the memory side): a one-tile
  // block per 16-64 voxels made that the most expensive part of the small layers (rocprof: 32->32 split-K 24 -> 52 us).  The ST
  // form therefore (a) is persistent above the number of blocks the chip holds at once (these kernels live on residency: the
  // grid is capped at that, not lower) and (b) spreads the blocks over slices of stat_out (common.h: conv_stat_send).
  __shared__ double st_tab[ST ? 128 : 1];
  __shared__ float st_aux[ST ? 256 : 1];
  if constexpr (ST != 0) {
    if (threadIdx.x < 128) st_tab[threadIdx.x] = 0.0;
    if (p.stat_mode == 2) {
      const int c = threadIdx.x & 63;
      st_aux[threadIdx.x] = (c < COUT) ? p.stat_aux[(threadIdx.x >> 6) * COUT + c] : 0.f;
    }
    __syncthreads();
  }
  for (unsigned vblk = blockIdx.x; vblk < (ST ? p.nblk : blockIdx.x + 1); vblk += gridDim.x) {
  const unsigned tile_blk = mdf::xcd_remap(vblk, p.nblk);
  const long long m0 = (SPLITK > 1 ? (long long)tile_blk : (long long)tile_blk * 4 + wave) * (MT * 16);

  // transposed: parity class of this block
  const int pc = (MODE == kTr) ? blockIdx.y : 0;   // 4 classes: (pd, ph); both pw live in the wave
  const int pd = (pc >> 1) & 1, ph = pc & 1;

  // per-lane voxel bookkeeping for each m-tile
  int in_off[MT];       // float offset of the base input voxel (channel 0)
  long long out_vox[MT];
  unsigned vmask[MT];   // bits 0-2: kd valid, 3-5: kh valid, 6-8: kw valid
  bool live[MT];
  // index space dims (output dims for S1/S2, input dims for Tr)
  const int Md = (MODE == kTr) ? p.Di : p.Do, Mh = (MODE == kTr) ? p.Hi : p.Ho, Mw = (MODE == kTr) ? p.Wi : p.Wo;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    long long m = m0 + t * 16 + n16;
    live[t] = m < p.m_total;
    if (!live[t]) m = p.m_total - 1;
    // (m < 2^31: the entry bounds the volume to 32-bit offsets.  32-bit divisions: the 64-bit ones were ~240 vector instructions per
    //  m-tile in front of the 100-400 MFMAs of a block -- a quarter of the run time of the small-volume launches, r05)
    const unsigned mu = (unsigned)m, r1 = mu / (unsigned)Mw, r2 = r1 / (unsigned)Mh;
    const int mw = (int)(mu - r1 * (unsigned)Mw), mh = (int)(r1 - r2 * (unsigned)Mh);
    const int b = (int)(r2 / (unsigned)Md), md = (int)(r2 - (unsigned)b * (unsigned)Md);
    int bd, bh, bw;  // base input coordinate
    unsigned vm = 0;
    if (MODE == kS1) { bd = md; bh = mh; bw = mw; }
    else if (MODE == kS2) { bd = 2 * md; bh = 2 * mh; bw = 2 * mw; }
    else { bd = md; bh = mh; bw = mw; }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int od, oh, ow;
      if (MODE == kTr) { od = oh = ow = (k == 0) ? 1 : 0; }
      else { od = oh = ow = k - 1; }
      if (bd + od >= 0 && bd + od < p.Di) vm |= 1u << k;
      if (bh + oh >= 0 && bh + oh < p.Hi) vm |= 8u << k;
      if (bw + ow >= 0 && bw + ow < p.Wi) vm |= 64u << k;
    }
    vmask[t] = live[t] ? vm : 0u;
    in_off[t] = (int)((((long long)b * p.Di + bd) * p.Hi + bh) * p.Wi + bw) * CIN;
    if (MODE == kTr)
      out_vox[t] = (((long long)b * p.Do + (2 * md + pd)) * p.Ho + (2 * mh + ph)) * p.Wo + 2 * mw;   // pw added in the epilogue
    else
      out_vox[t] = m;
  }

  // Depth taps that NO voxel of this wave's tiles has (the innermost U-Net levels are 1-3 planes deep: at D = 1 only kd = 1 exists,
  // at D = 2 two of three) are skipped altogether -- they contributed exact zeros (operand scaled by 0) for a third to two thirds of
  // the launch's MFMAs and fragment fetches.  Wave-uniform: a wave's tiles are runs of a row and share d except across a plane edge.
  unsigned kd_any = 7u;
  if (MODE != kTr && p.kd_skip) {
    unsigned m_or = 0u;
#pragma unroll
    for (int t = 0; t < MT; ++t) m_or |= vmask[t];
    kd_any = (__any((int)(m_or & 1u)) ? 1u : 0u) | (__any((int)(m_or & 2u)) ? 2u : 0u) | (__any((int)(m_or & 4u)) ? 4u : 0u);
    kd_any = (unsigned)__builtin_amdgcn_readfirstlane((int)kd_any);
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Epilogue operands (folded BN, residual) are fetched NOW so their L2 round trip overlaps the tap loop: a block lives for
  // one wave tile (10-20 us), an exposed ~1 us at its end is 5-10 % of it.
  // (the residual only for the transposed layers -- the ones that carry a skip connection in these networks; holding it for
  // the others costs the big-tile kernels an occupancy step)
  constexpr bool kPrefetchRes = (MODE == kTr);
  float4 ep_al[NT], ep_be[NT], ep_res[kPrefetchRes ? MT : 1][NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int r0 = nt * 16 + 4 * q;
    const int c0 = (MODE == kTr) ? r0 % COUT : r0;
    ep_al[nt] = make_float4(1.f, 1.f, 1.f, 1.f);
    ep_be[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 < ROWS && p.alpha) {
      ep_al[nt] = *reinterpret_cast<const float4*>(p.alpha + c0);
      ep_be[nt] = *reinterpret_cast<const float4*>(p.beta + c0);
    }
    if constexpr (kPrefetchRes) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        ep_res[t][nt] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 < ROWS && p.res && live[t]) {
          const int pw_out = r0 / COUT;
          ep_res[t][nt] = *reinterpret_cast<const float4*>(p.res + (size_t)(out_vox[t] + pw_out) * COUT + c0);
        }
      }
    }
  }

  const float* xq = p.x + KPL * q;                       // this lane's cin slot inside a chunk
  const float* wl = p.wpack + (size_t)lane * KPL;        // this lane's slot inside a packed 64-lane fragment

  // transposed: output parity p uses kernel taps with (k + p) odd: p=0 -> k=1 (input i') ; p=1 -> k=0 (input i'+1),
  // k=2 (input i').  Only the class's valid (kd, kh) pairs are enumerated (1, 2 or 4 of them) x 2 input w-offsets, so a
  // split-K stride hands every wave real work.  Along w: offset 0 feeds pw=0 (kw=1) and pw=1 (kw=2); offset +1 feeds pw=1 (kw=0).
  const int nkh = 1 + ph;
  const int ntaps = (MODE == kTr) ? (1 + pd) * nkh * 2 : 27;
  // Software pipeline over the taps: the operands of tap j+1 (all cin chunks: NCH*(MT+NT) 16-byte fragments) are requested before
  // the MFMAs of tap j.  A block lives for ONE wave tile, so without it every tap exposes an L2 round trip: the small layers
  // (1/8-resolution volumes, the transposed up-sampling layers) were pure latency -- 64->32 T at 1x74x100: 46 us for 0.8 GFLOP.
  struct TapRegs {
    Frag<KPL> bf[NCH][MT], af[NCH][NT];
    int ow;
  };
  auto load_tap = [&](int j, TapRegs& r) {
    int kd, kh, kw, od, oh, ow, tap;
    if (MODE == kTr) {
      ow = j & 1;
      const int jh = (j >> 1) % nkh, jd = (j >> 1) / nkh;
      kd = pd ? 2 * jd : 1;
      kh = ph ? 2 * jh : 1;
      od = (kd == 0); oh = (kh == 0);
      kw = ow ? 0 : 1;                  // validity-mask slot of the w offset: bit 0 <-> offset +1, bit 1 <-> offset 0
      tap = (kd * 3 + kh) * 2 + ow;     // slot in the packed weights
    } else {
      tap = j;
      kd = tap / 9; kh = (tap / 3) % 3; kw = tap % 3;
      od = kd - 1; oh = kh - 1; ow = kw - 1;
    }
    r.ow = ow;
    const int tapoff = ((od * p.Hi + oh) * p.Wi + ow) * CIN;
    const unsigned need = (1u << kd) | (8u << kh) | (64u << kw);
    const float* wt = wl + (size_t)tap * (NCH * NT * 64 * KPL);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        // branch-free: an out-of-range tap reads the voxel's own (valid) position and is zeroed by a multiply.  A predicated
        // load compiles to a branch and a wait per load (s_and_saveexec / s_cbranch_execz), which serialises the fetches of
        // a tap step; a select instead of the multiply lets the compiler sink the load back under the condition.
        const bool ok = (vmask[t] & need) == need;
        r.bf[ch][t].load(xq + in_off[t] + (ok ? tapoff : 0) + ch * CK);
        r.bf[ch][t].scale(ok ? 1.0f : 0.0f);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) r.af[ch][nt].load(wt + (size_t)(ch * NT + nt) * (64 * KPL));
    }
  };
  auto mul_tap = [&](const TapRegs& r) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int s = 0; s < KPL; ++s)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            // transposed, offset +1: rows of parity pw = 0 are structurally zero -> skip n-tiles made only of such rows
            if (MODE == kTr && r.ow == 1 && (nt + 1) * 16 <= COUT) continue;
            acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.af[ch][nt].v[s], r.bf[ch][t].v[s], acc[t][nt], 0, 0, 0);
          }
  };
  // (measured per variant: the single-tile transposed kernels gain 1.1-1.9x -- 64->32 T 77 / 44 / 46 -> 53 / 23 / 25 us; the big-tile
  // variants lose the occupancy their HBM-bound layers live on -- 32->16 T 92 -> 125 us, 16->8 T 88 -> 104 --, and the split-K stride-1
  // kernels lose too -- 64->64 65 -> 85: those keep one tap in flight)
  constexpr bool kPipeTaps = (MODE == kTr && MT == 1);
  {
    const int js = (SPLITK > 1) ? SPLITK : 1;
    int j = (SPLITK > 1) ? wave : 0;
    if constexpr (MODE == kTr) {
      // shallow volumes: a class with output parity pd = 1 has the depth taps kd = 0 (input plane d + 1) and kd = 2 (plane d); on the last
      // plane -- the only one at Di = 1 -- the kd = 0 half multiplies zeros.  kd is the slowest tap index: skipping it is a later start.
      if (p.kd_skip && pd && !__any((int)(vmask[0] & 1u))) {
        bool none = true;
#pragma unroll
        for (int t = 1; t < MT; ++t) none = none && !__any((int)(vmask[t] & 1u));
        if (none) j = __builtin_amdgcn_readfirstlane(nkh * 2);
      }
    }
    if constexpr (kPipeTaps) {
      TapRegs ra, rb;
      if (j < ntaps) load_tap(j, ra);
      while (j < ntaps) {
        if (j + js < ntaps) load_tap(j + js, rb);
        __builtin_amdgcn_sched_barrier(0);       // (the requests stay ahead of the MFMAs; hipcc otherwise sinks them to their first use)
        mul_tap(ra);
        j += js;
        if (j >= ntaps) break;
        if (j + js < ntaps) load_tap(j + js, ra);
        __builtin_amdgcn_sched_barrier(0);
        mul_tap(rb);
        j += js;
      }
    } else {
      if (kd_any == 7u) {                 // (the common case keeps its own loop: a test per tap costs the unrolled form 10-25 %, measured)
        for (; j < ntaps; j += js) {
          TapRegs r;
          load_tap(j, r);
          mul_tap(r);
        }
      } else {
        for (; j < ntaps; j += js) {
          if (!((kd_any >> (j / 9)) & 1u)) continue;     // (see kd_any)
          TapRegs r;
          load_tap(j, r);
          mul_tap(r);
        }
      }
    }
  }

  if (SPLITK > 1) {
    // partial sums -> LDS [wave][t*NT+nt][lane]; pair p is finished (summed + epilogue) by wave p % 4
    __shared__ f32x4 part[4][MT * NT][64];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) part[wave][t * NT + nt][lane] = acc[t][nt];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (((t * NT + nt) & 3) == wave) {
          const f32x4 a0 = part[0][t * NT + nt][lane], a1 = part[1][t * NT + nt][lane];
          const f32x4 a2 = part[2][t * NT + nt][lane], a3 = part[3][t * NT + nt][lane];
          acc[t][nt] = (a0 + a1) + (a2 + a3);
        }
      }
  }

  // epilogue: lane owns couts nt*16 + 4q .. +3 of voxel out_vox[t]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int r0 = nt * 16 + 4 * q;   // first GEMM row of this lane
    if (r0 >= ROWS) continue;
    const int pw_out = (MODE == kTr) ? r0 / COUT : 0;
    const int c0 = (MODE == kTr) ? r0 % COUT : r0;
    const float4 al = ep_al[nt], be = ep_be[nt];
    float ps[4] = {0.f, 0.f, 0.f, 0.f}, pq[4] = {0.f, 0.f, 0.f, 0.f};   // ST: this lane's sums over its m-tiles
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      if (!live[t]) continue;
      if (SPLITK > 1 && ((t * NT + nt) & 3) != wave) continue;  // another wave finishes this pair
      float4 o;
      o.x = acc[t][nt][0] * al.x + be.x;
      o.y = acc[t][nt][1] * al.y + be.y;
      o.z = acc[t][nt][2] * al.z + be.z;
      o.w = acc[t][nt][3] * al.w + be.w;
      if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      const size_t oi = (size_t)(out_vox[t] + pw_out) * COUT + c0;
      if (p.res) {
        const float4 rr = kPrefetchRes ? ep_res[kPrefetchRes ? t : 0][nt] : *reinterpret_cast<const float4*>(p.res + oi);
        o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
      }
      *reinterpret_cast<float4*>(p.y + oi) = o;
      if constexpr (ST != 0) {
        const float ov[4] = {o.x, o.y, o.z, o.w};
        if (p.stat_mode == 1) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { ps[k] += ov[k]; pq[k] = fmaf(ov[k], ov[k], pq[k]); }
        } else {
          const float4 yv4 = *reinterpret_cast<const float4*>(p.stat_y + oi);
          const float yv[4] = {yv4.x, yv4.y, yv4.z, yv4.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float dr = (fmaf(yv[k], st_aux[c0 + k], st_aux[64 + c0 + k]) > 0.0f) ? ov[k] : 0.0f;
            ps[k] += dr;
            pq[k] = fmaf(dr, (yv[k] - st_aux[128 + c0 + k]) * st_aux[192 + c0 + k], pq[k]);
          }
        }
      }
    }
    if constexpr (ST != 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a1 = row16_sum(ps[k]), a2 = row16_sum(pq[k]);
        if (n16 == 0) { lds_add_f64(&st_tab[c0 + k], (double)a1); lds_add_f64(&st_tab[64 + c0 + k], (double)a2); }
      }
    }
  }
  if constexpr (ST != 0 && SPLITK > 1) __syncthreads();   // the next tile's partial sums overwrite `part`
  }  // tile loop (one tile unless ST)
  if constexpr (ST != 0) {
    __syncthreads();
    mdf::conv_stat_send<COUT>(st_tab, p.stat_out, 2 * COUT, p.stat_slices, blockIdx.y * gridDim.x + blockIdx.x);
  }
}

// ---- ConvTranspose3d(k3,s2,p1,op1) on large volumes: all four (pd, ph) output parity classes from one pass over the input -----
// conv3d_kernel<kTr> runs one block per class: every class walks its own 2-8 taps, and the four together fetch the tile's input
// fragments 18 times (once per class tap).  The classes use only 8 distinct input positions -- (d, h, w) + (od, oh, ow), offsets
// 0/1 -- so here a wave visits the 8 positions once and feeds each fragment to every class that has a tap there (4 + 2 + 2 + 1
// classes per w offset = the same 18 MFMA groups): 8 input fetches instead of 18, four times the MFMAs behind every fetch, and
// the 2 x 2 (d, h) output neighbourhood of a tile written by one block.  Positions run (od, oh) = (1,1), (1,0), (0,1), (0,0), so
// every class accumulates its taps in the order of conv3d_kernel<kTr>: bit-identical results.
// NS > 1 (small volumes: fewer m-tiles than the chip has SIMDs): the n-tiles of an m-tile are dealt out to NS waves (blockIdx.y = first
// n-tile, stride NS), each fetching the input fragments itself -- NS times the waves at 1/NS of the MFMAs and weight bytes each.  With
// NS = 2 and four n-tiles a wave holds one tile of each output w-parity, so the two halves carry the same work.
// WL (r05; = waves per block): the packed weight set (18 tap slots; 72 KB at 32 -> 16, 18 KB at 16 -> 8) in LDS, one 12- or 16-wave block per CU walking the m-tiles
// (XCD-chunked, as conv3d_wlds_kernel): the per-class kernel streams 27 KB (16 -> 8) / 55 KB (32 -> 16) of weight fragments through the
// vector L1 for every 16 input voxels -- 3-6 times the bytes of the skip + output stream those voxels cause in HBM.
template <int CIN, int COUT, int MT, int NS, int WL>
__global__ __launch_bounds__(WL ? WL * 64 : 256) void convtr_all_kernel(const ConvParams p) {
  constexpr int KPL = (CIN >= 16) ? 4 : 2, CK = 4 * KPL, NCH = CIN / CK;
  constexpr int ROWS = 2 * COUT, NTA = (ROWS + 15) / 16, NT = NTA / NS;     // NTA n-tiles in all, NT of them in this wave
  static_assert(NTA % NS == 0, "the n-tiles must divide over the NS waves");
  const int nt0 = (NS > 1) ? (int)blockIdx.y : 0;                          // this wave's n-tiles: nt0 + i * NS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, n16 = lane & 15;
  extern __shared__ float wsm[];
  float4 ep_al[NT], ep_be[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int r0 = (nt0 + nt * NS) * 16 + 4 * q, c0 = r0 % COUT;
    ep_al[nt] = make_float4(1.f, 1.f, 1.f, 1.f);
    ep_be[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 < ROWS && p.alpha) {
      ep_al[nt] = *reinterpret_cast<const float4*>(p.alpha + c0);
      ep_be[nt] = *reinterpret_cast<const float4*>(p.beta + c0);
    }
  }
  const float* xq = p.x + KPL * q;
  const float* wl = (WL ? wsm : p.wpack) + (size_t)lane * KPL;
  unsigned l_first = 0u, l_end = 1u, l_step = 1u, c_start = 0u;
  if constexpr (WL != 0) {
    static_assert(NS == 1, "the LDS-weights form keeps an m-tile's n-tiles in one wave");
    const float4* src = reinterpret_cast<const float4*>(p.wpack);
    float4* dst = reinterpret_cast<float4*>(wsm);
    for (int i = threadIdx.x; i < 18 * NCH * NTA * 64 * KPL / 4; i += WL * 64) dst[i] = src[i];
    __syncthreads();
    const unsigned xcd = blockIdx.x & 7u, bx = blockIdx.x >> 3, nbx = gridDim.x >> 3;      // (p.nblk = wave tiles here)
    const unsigned cq = p.nblk >> 3, cr = p.nblk & 7u;
    c_start = (xcd < cr) ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    l_first = (unsigned)wave * nbx + bx; l_end = cq + (xcd < cr ? 1u : 0u); l_step = (unsigned)WL * nbx;
  }
  for (unsigned l = l_first; l < l_end; l += l_step) {
  const long long m0 = WL ? (long long)(c_start + l) * (MT * 16) : ((long long)mdf::xcd_remap(blockIdx.x, p.nblk) * 4 + wave) * (MT * 16);

  int in_off[MT];
  long long out_vox[MT];      // output voxel (2d, 2h, 2w) of the lane's input voxel
  unsigned vmask[MT];         // bit 0: d+1 inside, bit 1: h+1 inside, bit 2: w+1 inside, bit 3: live
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    long long m = m0 + t * 16 + n16;
    const bool live = m < p.m_total;
    if (!live) m = p.m_total - 1;
    const unsigned mu = (unsigned)m, r1 = mu / (unsigned)p.Wi, r2 = r1 / (unsigned)p.Hi;      // (32-bit: see conv3d_kernel)
    const int mw = (int)(mu - r1 * (unsigned)p.Wi), mh = (int)(r1 - r2 * (unsigned)p.Hi);
    const int b = (int)(r2 / (unsigned)p.Di), md = (int)(r2 - (unsigned)b * (unsigned)p.Di);
    vmask[t] = live ? (8u | (md + 1 < p.Di ? 1u : 0u) | (mh + 1 < p.Hi ? 2u : 0u) | (mw + 1 < p.Wi ? 4u : 0u)) : 0u;
    in_off[t] = (int)((((long long)b * p.Di + md) * p.Hi + mh) * p.Wi + mw) * CIN;
    out_vox[t] = (((long long)b * p.Do + 2 * md) * p.Ho + 2 * mh) * p.Wo + 2 * mw;
  }
  f32x4 acc[4][MT][NT];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[c][t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // WL: the skip tensor's values are requested now, ahead of the tile's MFMAs (a persistent wave has no neighbour block to hide the
  // epilogue's round trip behind)
  constexpr bool kPreRes = (WL != 0);
  float4 ep_res[kPreRes ? 4 : 1][kPreRes ? MT : 1][kPreRes ? NT : 1];
  if constexpr (kPreRes) {
    if (p.res) {
#pragma unroll
      for (int cls = 0; cls < 4; ++cls)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int r0 = (nt0 + nt * NS) * 16 + 4 * q;
            const size_t oi = (size_t)(out_vox[t] + ((long long)(cls >> 1) * p.Ho + (cls & 1)) * p.Wo + r0 / COUT) * COUT + r0 % COUT;
            ep_res[cls][t][nt] = (r0 < ROWS && (vmask[t] & 8u)) ? *reinterpret_cast<const float4*>(p.res + oi) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
    }
  }
  bool no_next_plane = false;
  if (p.kd_skip) {
    unsigned m_or = 0u;
#pragma unroll
    for (int t = 0; t < MT; ++t) m_or |= vmask[t];
    no_next_plane = !__any((int)(m_or & 1u));
  }
  // The fragments of PG positions are requested together and zeroed where they are used (one round trip per group instead of one per
  // position: a persistent wave's tiles run back to back, nobody else hides them)
#ifndef MDF_CONVTR_PG64
#define MDF_CONVTR_PG64 0      // dev A/B (build_variant.sh): 1 = the one-tile 64 -> 32 kernels fetch four positions at a time too (measured: @12x37x50 49.9 -> 60.0 us, worse)
#endif
  constexpr int PG = (WL != 0 || (CIN >= 64 && MDF_CONVTR_PG64)) ? ((64 / (NCH * MT * KPL) >= 8) ? 8 : (64 / (NCH * MT * KPL) >= 4) ? 4 : 1) : 1;
  Frag<KPL> bfa[PG][NCH][MT];
#pragma unroll
  for (int pos = 0; pos < 8; ++pos) {
    const int od = 1 - (pos >> 2), oh = 1 - ((pos >> 1) & 1), ow = pos & 1;
    const unsigned need = 8u | (od ? 1u : 0u) | (oh ? 2u : 0u) | (ow ? 4u : 0u);
    if constexpr (PG > 1) {
      if (pos % PG == 0) {
#pragma unroll
        for (int p2 = pos; p2 < pos + PG; ++p2) {
          const int od2 = 1 - (p2 >> 2), oh2 = 1 - ((p2 >> 1) & 1), ow2 = p2 & 1;
          const unsigned need2 = 8u | (od2 ? 1u : 0u) | (oh2 ? 2u : 0u) | (ow2 ? 4u : 0u);
          const int tapoff2 = ((od2 * p.Hi + oh2) * p.Wi + ow2) * CIN;
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int t = 0; t < MT; ++t) bfa[p2 - pos][ch][t].load(xq + in_off[t] + (((vmask[t] & need2) == need2) ? tapoff2 : 0) + ch * CK);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const int tapoff = ((od * p.Hi + oh) * p.Wi + ow) * CIN;
    if (NS == NTA && NTA > 1 && ow == 1 && (nt0 + 1) * 16 <= COUT) continue;   // one n-tile per wave, of parity pw = 0: nothing to do at offset +1
    if (od == 1 && no_next_plane) continue;             // shallow volumes: no voxel of the wave has plane d + 1 (see conv3d_kernel: kd_skip)
    Frag<KPL> bf[NCH][MT];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int t = 0; t < MT; ++t) {   // (branch-free, as in conv3d_kernel: an absent neighbour reads the voxel itself and is zeroed)
        const bool ok = (vmask[t] & need) == need;
        if constexpr (PG > 1) bf[ch][t] = bfa[pos % PG][ch][t];
        else bf[ch][t].load(xq + in_off[t] + (ok ? tapoff : 0) + ch * CK);
        bf[ch][t].scale(ok ? 1.0f : 0.0f);
      }
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
      const int pd = cls >> 1, ph = cls & 1;
      if (od > pd || oh > ph) continue;                   // parity 0 has its one tap at offset 0
      const int kd = pd ? (od ? 0 : 2) : 1, kh = ph ? (oh ? 0 : 2) : 1;
      const float* wt = wl + (size_t)((kd * 3 + kh) * 2 + ow) * (NCH * NTA * 64 * KPL);
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        Frag<KPL> af[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) af[nt].load(wt + (size_t)(ch * NTA + nt0 + nt * NS) * (64 * KPL));
#pragma unroll
        for (int s = 0; s < KPL; ++s)
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              // offset +1 feeds parity pw = 1 only: the n-tiles of pw = 0 are structurally zero (NS = 2 of 4 tiles: the wave's first; else wave-uniform)
              const bool pw0_tile = (NS == 2 && NTA == 4) ? (nt == 0) : ((nt0 + nt * NS + 1) * 16 <= COUT);
              if (ow == 1 && pw0_tile) continue;
              acc[cls][t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[nt].v[s], bf[ch][t].v[s], acc[cls][t][nt], 0, 0, 0);
            }
      }
    }
  }
#pragma unroll
  for (int cls = 0; cls < 4; ++cls) {
    const int pd = cls >> 1, ph = cls & 1;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int r0 = (nt0 + nt * NS) * 16 + 4 * q;
      if (r0 >= ROWS) continue;
      const int pw_out = r0 / COUT, c0 = r0 % COUT;
      const float4 al = ep_al[nt], be = ep_be[nt];
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        if (!(vmask[t] & 8u)) continue;
        float4 o;
        o.x = acc[cls][t][nt][0] * al.x + be.x;
        o.y = acc[cls][t][nt][1] * al.y + be.y;
        o.z = acc[cls][t][nt][2] * al.z + be.z;
        o.w = acc[cls][t][nt][3] * al.w + be.w;
        if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        const size_t oi = (size_t)(out_vox[t] + ((long long)pd * p.Ho + ph) * p.Wo + pw_out) * COUT + c0;
        if (p.res) {
          float4 rr;
          if constexpr (kPreRes) rr = ep_res[cls][t][nt];
          else rr = *reinterpret_cast<const float4*>(p.res + oi);
          o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
        }
        *reinterpret_cast<float4*>(p.y + oi) = o;
      }
    }
  }
  }  // tile loop (one tile unless WL)
}

// ---- 64 -> 32 transposed layers: one parity class per persistent block, the class's tap slots in LDS ---------------------------------
// The 64 -> 32 weight set (18 slots x 16 KB) does not fit LDS, and the all-classes kernel streams it through the vector L1 once per
// 16 input voxels (0.6 GB from the L2s per launch at 12x37x50: that stream, not the MFMAs, is its time).  The slots of ONE (pd, ph) class
// do fit (2 / 4 / 4 / 8 slots = 32-128 KB): the 256 blocks are shared out 32 : 56 : 56 : 112 over the four classes (their MFMA counts
// are 1 : 2 : 2 : 4; multiples of 8, so every class has blocks on all XCDs), a block copies its class's slots to LDS once and its waves
// walk ALL m-tiles for that class (XCD-chunked, as conv3d_wlds_kernel): A fragments by ds_read_b128, the B fragments of up to four
// input positions and the skip values requested together.  Per accumulator the taps run in conv3d_kernel<kTr>'s order: bit-identical.
template <int CIN, int COUT, int NWV, int PGMAX, int PD, int PH>
__device__ __forceinline__ void convtr_cls_body(const ConvParams& p, float* wsm, unsigned cls_first, unsigned cls_blocks) {
  constexpr int KPL = 4, CK = 16, NCH = CIN / CK, ROWS = 2 * COUT, NT = (ROWS + 15) / 16;
  constexpr int SLOTF = NCH * NT * 64 * KPL;             // floats per tap slot
  constexpr int NKH = 1 + PH, NCOMB = (1 + PD) * NKH, NPOS = 2 * NCOMB;
  constexpr int PG = (NPOS < PGMAX) ? NPOS : PGMAX;      // positions fetched together (16 registers each at 64 input channels)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, n16 = lane & 15;
  {   // the class's slots, in tap order: LDS slot (c, ow) = packed slot (kd*3 + kh)*2 + ow
#pragma unroll
    for (int c = 0; c < NCOMB; ++c) {
      const int jd = c / NKH, jh = c % NKH;
      const int kd = PD ? 2 * jd : 1, kh = PH ? 2 * jh : 1;
      const float4* src = reinterpret_cast<const float4*>(p.wpack + (size_t)((kd * 3 + kh) * 2) * SLOTF);
      float4* dst = reinterpret_cast<float4*>(wsm + (size_t)(c * 2) * SLOTF);
      for (int i = threadIdx.x; i < 2 * SLOTF / 4; i += NWV * 64) dst[i] = src[i];
    }
  }
  __syncthreads();
  const float* xq = p.x + KPL * q;
  const float* wl = wsm + lane * KPL;
  const unsigned xcd = blockIdx.x & 7u, bx = (blockIdx.x - cls_first) >> 3, nbx = cls_blocks >> 3;
  const unsigned cq = p.nblk >> 3, cr = p.nblk & 7u;                     // (p.nblk = m-tiles)
  const unsigned c_start = (xcd < cr) ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq, c_len = cq + (xcd < cr ? 1u : 0u);
  for (unsigned l = (unsigned)wave * nbx + bx; l < c_len; l += (unsigned)NWV * nbx) {
    long long m = (long long)(c_start + l) * 16 + n16;
    const bool live = m < p.m_total;
    if (!live) m = p.m_total - 1;
    const unsigned mu = (unsigned)m, r1 = mu / (unsigned)p.Wi, r2 = r1 / (unsigned)p.Hi;
    const int mw = (int)(mu - r1 * (unsigned)p.Wi), mh = (int)(r1 - r2 * (unsigned)p.Hi);
    const int b = (int)(r2 / (unsigned)p.Di), md = (int)(r2 - (unsigned)b * (unsigned)p.Di);
    const unsigned vmask = live ? (8u | (md + 1 < p.Di ? 1u : 0u) | (mh + 1 < p.Hi ? 2u : 0u) | (mw + 1 < p.Wi ? 4u : 0u)) : 0u;
    const int in_off = (int)((((long long)b * p.Di + md) * p.Hi + mh) * p.Wi + mw) * CIN;
    const long long out_vox = (((long long)b * p.Do + 2 * md + PD) * p.Ho + 2 * mh + PH) * p.Wo + 2 * mw;
    const bool no_next_plane = p.kd_skip && !__any((int)(vmask & 1u));
    float4 ep_res[NT];
    if (p.res) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int r0 = nt * 16 + 4 * q;
        ep_res[nt] = (r0 < ROWS && live) ? *reinterpret_cast<const float4*>(p.res + (size_t)(out_vox + r0 / COUT) * COUT + r0 % COUT)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Frag<KPL> bfa[PG][NCH];
#pragma unroll
    for (int pos = 0; pos < NPOS; ++pos) {     // pos = c*2 + ow: conv3d_kernel<kTr>'s tap order
      if (pos % PG == 0) {
#pragma unroll
        for (int p2 = pos; p2 < pos + PG && p2 < NPOS; ++p2) {
          const int c2 = p2 >> 1, ow2 = p2 & 1, jd2 = c2 / NKH, jh2 = c2 % NKH;
          const int od2 = (PD && jd2 == 0) ? 1 : 0, oh2 = (PH && jh2 == 0) ? 1 : 0;
          const unsigned need2 = 8u | (od2 ? 1u : 0u) | (oh2 ? 2u : 0u) | (ow2 ? 4u : 0u);
          const int tapoff2 = ((od2 * p.Hi + oh2) * p.Wi + ow2) * CIN;
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch) bfa[p2 - pos][ch].load(xq + in_off + (((vmask & need2) == need2) ? tapoff2 : 0) + ch * CK);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const int c = pos >> 1, ow = pos & 1, jd = c / NKH, jh = c % NKH;
      const int od = (PD && jd == 0) ? 1 : 0, oh = (PH && jh == 0) ? 1 : 0;
      const unsigned need = 8u | (od ? 1u : 0u) | (oh ? 2u : 0u) | (ow ? 4u : 0u);
      if (od == 1 && no_next_plane) continue;       // shallow volumes: no voxel of the wave has plane d + 1 (conv3d_kernel: kd_skip)
      const float okf = ((vmask & need) == need) ? 1.0f : 0.0f;
      const float* wt = wl + pos * SLOTF;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        Frag<KPL> bf = bfa[pos % PG][ch];
        bf.scale(okf);
        Frag<KPL> af[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) af[nt].load(wt + (ch * NT + nt) * (64 * KPL));
#pragma unroll
        for (int s = 0; s < KPL; ++s)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if (ow == 1 && (nt + 1) * 16 <= COUT) continue;   // offset +1 feeds parity pw = 1 only
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[nt].v[s], bf.v[s], acc[nt], 0, 0, 0);
          }
      }
    }
    if (!live) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int r0 = nt * 16 + 4 * q;
      if (r0 >= ROWS) continue;
      const int pw_out = r0 / COUT, c0 = r0 % COUT;
      float4 al = make_float4(1.f, 1.f, 1.f, 1.f), be = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.alpha) { al = *reinterpret_cast<const float4*>(p.alpha + c0); be = *reinterpret_cast<const float4*>(p.beta + c0); }
      float4 o;
      o.x = acc[nt][0] * al.x + be.x;
      o.y = acc[nt][1] * al.y + be.y;
      o.z = acc[nt][2] * al.z + be.z;
      o.w = acc[nt][3] * al.w + be.w;
      if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      if (p.res) { const float4 rr = ep_res[nt]; o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w; }
      *reinterpret_cast<float4*>(p.y + (size_t)(out_vox + pw_out) * COUT + c0) = o;
    }
  }
}

template <int CIN, int COUT, int NWV, int PGMAX>
__global__ __launch_bounds__(NWV * 64) void convtr_cls_kernel(const ConvParams p) {
  extern __shared__ float wsm[];
  const unsigned bid = blockIdx.x;       // grid = 256 blocks
  if (bid < 32u) convtr_cls_body<CIN, COUT, NWV, PGMAX, 0, 0>(p, wsm, 0u, 32u);
  else if (bid < 88u) convtr_cls_body<CIN, COUT, NWV, PGMAX, 0, 1>(p, wsm, 32u, 56u);
  else if (bid < 144u) convtr_cls_body<CIN, COUT, NWV, PGMAX, 1, 0>(p, wsm, 88u, 56u);
  else convtr_cls_body<CIN, COUT, NWV, PGMAX, 1, 1>(p, wsm, 144u, 112u);
}

template <int CIN, int COUT, int NWV, int PGMAX>
int launch_convtr_cls(ConvParams& p, hipStream_t st) {
  constexpr int NCH = CIN / 16, NT = (2 * COUT + 15) / 16;
  constexpr size_t kLds = (size_t)8 * NCH * NT * 64 * 4 * sizeof(float);      // the (1, 1) class: 8 slots
  static_assert(kLds <= 160 * 1024, "a class's tap slots must fit LDS");
  auto kern = &convtr_cls_kernel<CIN, COUT, NWV, PGMAX>;
  static bool attr_done_dev[64] = {};     // (per-device function attribute: see conv_lds.hip)
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  bool& attr_done = attr_done_dev[(dev_id >= 0 && dev_id < 64) ? dev_id : 0];
  if (!attr_done || dev_id >= 64) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
    if (e != hipSuccess) return mdf::fail(MDF_EHIP, "hipFuncSetAttribute(dynamic LDS %zu): %s", kLds, hipGetErrorString(e));
    attr_done = true;
  }
  p.nblk = (unsigned)((p.m_total + 15) / 16);      // m-tiles
  hipLaunchKernelGGL(kern, dim3(256), dim3(NWV * 64), kLds, st, p);
  return mdf::check_launch("convtr_cls_kernel");
}

template <int CIN, int COUT, int MT, int NWV>
int launch_convtr_all_wl(ConvParams& p, hipStream_t st) {
  constexpr int KPL = (CIN >= 16) ? 4 : 2, NCH = CIN / (4 * KPL), NTA = (2 * COUT + 15) / 16;
  constexpr size_t kLds = (size_t)18 * NCH * NTA * 64 * KPL * sizeof(float);
  static_assert(kLds <= 160 * 1024, "the packed weight set must fit LDS");
  auto kern = &convtr_all_kernel<CIN, COUT, MT, 1, NWV>;
  static bool attr_done_dev[64] = {};     // (per-device function attribute: see conv_lds.hip)
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  bool& attr_done = attr_done_dev[(dev_id >= 0 && dev_id < 64) ? dev_id : 0];
  if (!attr_done || dev_id >= 64) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
    if (e != hipSuccess) return mdf::fail(MDF_EHIP, "hipFuncSetAttribute(dynamic LDS %zu): %s", kLds, hipGetErrorString(e));
    attr_done = true;
  }
  p.nblk = (unsigned)((p.m_total + MT * 16 - 1) / (MT * 16));      // wave tiles
  hipLaunchKernelGGL(kern, dim3(256), dim3(NWV * 64), kLds, st, p);     // one block per CU
  return mdf::check_launch("convtr_all_kernel");
}

template <int CIN, int COUT, int MT, int NS = 1>
int launch_convtr_all(ConvParams& p, hipStream_t st) {
  p.nblk = (unsigned)((p.m_total + 4LL * MT * 16 - 1) / (4LL * MT * 16));
  hipLaunchKernelGGL((convtr_all_kernel<CIN, COUT, MT, NS, 0>), dim3(p.nblk, NS), dim3(256), 0, st, p);
  return mdf::check_launch("convtr_all_kernel");
}

// ---- weight packing ------------------------------------------------------------------------------------------------
// Every packing reads the LOGICAL conv weight W[o][i][t] (o < Co output rows, i < Ci input channels in memory, t < T taps)
// through WSrc, which maps it onto the parameter tensor as it sits in the module.  Mode 0 is the parameter itself; the
// other modes are the re-indexings the training step needs every time the optimizer has changed the weights (the input-
// gradient convs of train_ops.py), done here instead of as flip / transpose / pad / index_select launches of their own.
enum { kSrcDirect = 0, kSrcSwapFlip = 1, kSrcK5S2Dgrad = 2, kSrcProbSlices = 3, kSrcShuffle2 = 4, kSrcSwap = 5, kSrcK5S2Phases = 6 };
struct WSrc {
  const float* w;
  int mode, Co, Ci, T, a0, a1;
  __device__ __forceinline__ float at(int o, int i, int t) const {
    switch (mode) {
      case kSrcSwapFlip:    // input gradient of a stride-1 conv: parameter [Ci][Co][T], taps mirrored
        return w[((size_t)i * Co + o) * T + (T - 1 - t)];
      case kSrcSwap:        // parameter laid out [Ci][Co][T] (ConvTranspose3d, or a conv weight read as one)
        return w[((size_t)i * Co + o) * T + t];
      case kSrcK5S2Dgrad: { // input gradient of Conv2d(k5,s2,p2) [Ci][a0][5][5] as a 3x3 conv over dy: row (py*2+px)*a0 + ci is dx[ci]
        const int of = o + a1, cls = of / a0, ci = of - cls * a0;   // at the pixels of parity (py,px); dx[2j+p] = sum_t dy[j+t-1] w[k(p,t)]
        const int ty = t / 3, tx = t - ty * 3;
        const int ky = (cls >> 1) ? (ty == 0 ? -1 : 5 - 2 * ty) : 4 - 2 * ty;    // p=0: (4,2,0); p=1: (-,3,1)
        const int kx = (cls & 1) ? (tx == 0 ? -1 : 5 - 2 * tx) : 4 - 2 * tx;
        return (ky < 0 || kx < 0) ? 0.f : w[(((size_t)i * a0 + ci) * 5 + ky) * 5 + kx];
      }
      case kSrcK5S2Phases: { // Conv2d(k5,s2,p2) [Co][a0][5][5] as a stride-1 3x3 conv over the four parity images of its input (internal:
        // the Winograd segment of a k5-s2 weight set): logical input channel i = (py*2+px)*a0 + ci, tap (ty,tx) <- (2(ty-1)+py+2, 2(tx-1)+px+2)
        const int ph = i / a0, ci = i - ph * a0;
        const int ky = 2 * (t / 3) + (ph >> 1), kx = 2 * (t % 3) + (ph & 1);
        return (ky > 4 || kx > 4) ? 0.f : w[(((size_t)o * a0 + ci) * 5 + ky) * 5 + kx];
      }
      case kSrcProbSlices:  // `prob` conv [1][Ci][3][3][3] as a 2-D conv with one output row per depth tap (+ a zero row)
        return o < 3 ? w[((size_t)i * 3 + o) * 9 + t] : 0.f;
      case kSrcShuffle2: {  // conv feeding PixelShuffle(2): row sub*Cq + oc <- channel oc*4 + sub
        const int cq = Co >> 2, sub = o / cq, oc = o - sub * cq;
        return w[((size_t)(oc * 4 + sub) * Ci + i) * T + t];
      }
      default:
        return w[((size_t)o * Ci + i) * T + t];
    }
  }
};

// plain: wpack[tap][chunk][nt][q][n][s] = W(cout = nt*16+n, cin = chunk*CK + KPL*q + s, tap)
__device__ __forceinline__ int pack_plain_total(int Cin, int Cout, int ntaps) {
  const int KPL = (Cin >= 16) ? 4 : (Cin == 8 ? 2 : 1), CK = 4 * KPL, NCH = Cin / CK, NT = (Cout + 15) / 16;
  return ntaps * NCH * NT * 64 * KPL;
}
__device__ __forceinline__ float pack_plain_elem(const WSrc& src, int i, int Cin, int Cin_mem, int Cout) {
  const int KPL = (Cin >= 16) ? 4 : (Cin == 8 ? 2 : 1), CK = 4 * KPL, NCH = Cin / CK, NT = (Cout + 15) / 16;
  int r = i;
  const int s = r % KPL; r /= KPL;
  const int n = r % 16; r /= 16;
  const int qq = r % 4; r /= 4;
  const int nt = r % NT; r /= NT;
  const int ch = r % NCH; r /= NCH;
  const int tap = r;
  const int cout = nt * 16 + n, cin = ch * CK + KPL * qq + s;
  return (cout < Cout && cin < Cin_mem) ? src.at(cout, cin, tap) : 0.f;
}

// w-phase packing for Cout < 16 (conv_lds.hip, Cfg::RW): the conv rewritten with GEMM row r*Cout + c = channel c of output
// phase r (RW phases along w) and KW' = KHW + RW - 1 taps along w; tap kw' of phase r is the original tap kw' - r.
// wp[tap' = kdh*KW' + kw'][chunk][q][n][s], one n-tile.
__device__ __forceinline__ int pack_rw_total(int Cin, int nkdh, int KHW, int RW) {
  const int KPL = (Cin >= 16) ? 4 : (Cin == 8 ? 2 : 1), CK = 4 * KPL, NCH = Cin / CK, KW = KHW + RW - 1;
  return nkdh * KW * NCH * 64 * KPL;
}
__device__ __forceinline__ float pack_rw_elem(const WSrc& src, int i, int Cin, int Cin_mem, int Cout, int KHW, int RW) {
  const int KPL = (Cin >= 16) ? 4 : (Cin == 8 ? 2 : 1), CK = 4 * KPL, NCH = Cin / CK, KW = KHW + RW - 1;
  int r = i;
  const int s = r % KPL; r /= KPL;
  const int n = r % 16; r /= 16;
  const int qq = r % 4; r /= 4;
  const int ch = r % NCH; r /= NCH;
  const int kwp = r % KW, kdh = r / KW;
  const int phase = n / Cout, cout = n % Cout, cin = ch * CK + KPL * qq + s;
  const int kw = kwp - phase;
  return (phase < RW && cin < Cin_mem && kw >= 0 && kw < KHW) ? src.at(cout, cin, kdh * KHW + kw) : 0.f;
}

// Winograd F(2x2,3x3) weights for conv_lds.hip step_wino: U[kd][a][b] = G g[kd] G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1];
// fragments in the order the kernel walks them: [kd][chunk][ab = a*4+b][nt][lane = q*16+m][s], cout = nt*16+m, cin = chunk*16+4q+s.
__device__ __forceinline__ int pack_wino_total(int Cin, int Cout, int nkd) { return nkd * (Cin / 16) * 16 * ((Cout + 15) / 16) * 64 * 4; }
__device__ __forceinline__ float pack_wino_elem(const WSrc& src, int i, int Cin, int Cout, int nkd) {
  const int NCH = Cin / 16, NT = (Cout + 15) / 16;
  const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
  int r = i;
  const int s = r % 4; r /= 4;
  const int m = r % 16; r /= 16;
  const int qq = r % 4; r /= 4;
  const int nt = r % NT; r /= NT;
  const int ab = r % 16; r /= 16;
  const int ch = r % NCH; r /= NCH;
  const int kd = r;
  const int a = ab >> 2, b = ab & 3;
  const int cout = nt * 16 + m, cin = ch * 16 + 4 * qq + s;
  float v = 0.f;
  if (cout < Cout) {
    for (int y = 0; y < 3; ++y)
      for (int x = 0; x < 3; ++x) v += G[a][y] * G[b][x] * src.at(cout, cin, kd * 9 + y * 3 + x);
  }
  return v;
}

// Depth-pair Winograd weights (conv_lds.hip, Cfg::RD = 2; 3-D layers with 8 output channels): GEMM row m = r*8 + c is channel c of
// output plane d + r (r = 0, 1), so the 16 MFMA rows are all live; input plane j = 0..3 of a step (depth d - 1 + j) meets tap kd = j - r.
// [j][chunk][ab][lane = q*16+m][s], cin = chunk*CK + KPL*q + s.
__device__ __forceinline__ float pack_wd_elem(const WSrc& src, int i, int Cin) {
  const int KPL = (Cin >= 16) ? 4 : 2, CK = 4 * KPL, NCH = Cin / CK;
  const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
  int r = i;
  const int s = r % KPL; r /= KPL;
  const int m = r % 16; r /= 16;
  const int qq = r % 4; r /= 4;
  const int ab = r % 16; r /= 16;
  const int ch = r % NCH; r /= NCH;
  const int j = r;
  const int a = ab >> 2, b = ab & 3;
  const int cout = m & 7, kd = j - (m >> 3), cin = ch * CK + KPL * qq + s;
  float v = 0.f;
  if (kd >= 0 && kd < 3) {
    for (int y = 0; y < 3; ++y)
      for (int x = 0; x < 3; ++x) v += G[a][y] * G[b][x] * src.at(cout, cin, kd * 9 + y * 3 + x);
  }
  return v;
}

// ConvTranspose3d weights [Cin][Cout][3][3][3] -> wpack[tap' = (kd*3+kh)*2+ow][chunk][nt][q][n][s] with GEMM row
// r = nt*16+n = pw*Cout + cout and kernel tap kw(pw, ow): (0,0)->1, (1,0)->2, (1,1)->0, (0,1)-> structurally zero.
__device__ __forceinline__ int pack_tr_total(int Cin, int Cout) {
  const int KPL = (Cin >= 16) ? 4 : 2, CK = 4 * KPL, NCH = Cin / CK, NT = (2 * Cout + 15) / 16;
  return 18 * NCH * NT * 64 * KPL;
}
__device__ __forceinline__ float pack_tr_elem(const WSrc& src, int i, int Cin, int Cout) {
  const int KPL = (Cin >= 16) ? 4 : 2, CK = 4 * KPL, NCH = Cin / CK, NT = (2 * Cout + 15) / 16;
  int r = i;
  const int s = r % KPL; r /= KPL;
  const int n = r % 16; r /= 16;
  const int qq = r % 4; r /= 4;
  const int nt = r % NT; r /= NT;
  const int ch = r % NCH; r /= NCH;
  const int tap = r;
  const int kd = tap / 6, kh = (tap / 2) % 3, ow = tap & 1;
  const int row = nt * 16 + n, cin = ch * CK + KPL * qq + s;
  if (row >= 2 * Cout) return 0.f;
  const int pw = row / Cout, cout = row % Cout;
  const int kw = (pw == 0) ? (ow == 0 ? 1 : -1) : (ow == 0 ? 2 : 0);
  return kw >= 0 ? src.at(cout, cin, (kd * 3 + kh) * 3 + kw) : 0.f;
}

__host__ __device__ inline int rw_of(int Cout) { return Cout == 8 ? 2 : (Cout == 4 ? 4 : 0); }   // w-phase factor of a stride-1 k3 layer (0 = none)
__host__ __device__ inline bool wino_built(int Cin, int Cout) {   // 3-D ((16, 32): the input-gradient conv of the stage-0 regulariser's first layer, training)
  return ((Cout == 16 || Cout == 32) && (Cin == 16 || Cin == 32)) || (Cin == 16 && Cout == 8);
}
__host__ __device__ inline bool wino2d_built(int Cin, int Cout) {   // ((16, 32), (32, 64): input gradients of the k5-s2 layers as 3x3 convs over the parity classes, training)
  return (Cin == 16 && Cout == 16) || (Cin == 32 && Cout == 32) || (Cin == 64 && Cout == 64) || (Cin == 16 && Cout == 32) || (Cin == 32 && Cout == 64);
}
__host__ __device__ inline bool wd_built(int Cin, int Cout) { return Cout == 8 && (Cin == 8 || Cin == 16); }   // 3-D, depth-pair Winograd
// 2-D k5 s2 layers that also run as a Winograd 3x3 conv over the four parity images of their input (conv_lds.hip, LdsConvParams::s2d)
__host__ __device__ inline bool k5w_built(int Cin, int Cout) { return (Cin == 16 && Cout == 32) || (Cin == 8 && Cout == 16); }
__host__ __device__ inline int padded_cin(int c) { return c <= 4 ? 4 : c; }

// One complete packed weight set, as mdf_conv3d_pack_weights / mdf_conv_pack_weights lay it out: the plain fragments, then
// (Cout 8 / 4, 3x3 taps) the w-phase fragments, then (where a Winograd kernel exists) the transform-domain fragments, then (3-D,
// Cout 8) the depth-pair Winograd fragments; or the transposed-conv fragments alone.
struct PackJob {
  const float* src;
  float* dst;
  int mode, transposed, is3d, Cin_mem, Cout, ntaps, a0, a1;
  int blk0, nblk;
};
__host__ __device__ inline void pack_segments(int is3d, int transposed, int Cin_mem, int Cout, int ntaps, long long* plain, long long* rw, long long* wino,
                                              long long* wd) {
  const int Cin = padded_cin(Cin_mem);
  const int KPL = (Cin >= 16) ? 4 : (Cin == 8 ? 2 : 1), NCH = Cin / (4 * KPL);
  *rw = 0; *wino = 0; *wd = 0;
  if (transposed) {
    const int K2 = (Cin >= 16) ? 4 : 2;
    *plain = 18ll * (Cin / (4 * K2)) * ((2 * Cout + 15) / 16) * 64 * K2;
    return;
  }
  *plain = (long long)ntaps * NCH * ((Cout + 15) / 16) * 64 * KPL;
  const bool k3 = is3d ? (ntaps == 27) : (ntaps == 9);
  if (k3 && rw_of(Cout)) *rw = (long long)(is3d ? 9 : 3) * (3 + rw_of(Cout) - 1) * NCH * 64 * KPL;
  if (k3 && (is3d ? wino_built(Cin_mem, Cout) : wino2d_built(Cin_mem, Cout))) *wino = (long long)(is3d ? 3 : 1) * (Cin / 16) * 16 * ((Cout + 15) / 16) * 64 * 4;
  if (k3 && is3d && wd_built(Cin_mem, Cout)) *wd = 4ll * NCH * 16 * 64 * KPL;
  if (!is3d && ntaps == 25 && k5w_built(Cin_mem, Cout)) *wino = (long long)(4 * Cin / 16) * 16 * ((Cout + 15) / 16) * 64 * 4;   // (its only extra segment)
}
__device__ __forceinline__ void pack_job_elem(const PackJob& j, long long i) {
  const int Cin = padded_cin(j.Cin_mem);
  WSrc src{j.src, j.transposed ? kSrcSwap : j.mode, j.Cout, j.Cin_mem, j.ntaps, j.a0, j.a1};
  long long plain, rw, wino, wd;
  pack_segments(j.is3d, j.transposed, j.Cin_mem, j.Cout, j.ntaps, &plain, &rw, &wino, &wd);
  if (i >= plain + rw + wino + wd) return;
  float v;
  if (j.transposed) v = pack_tr_elem(src, (int)i, Cin, j.Cout);
  else if (i < plain) v = pack_plain_elem(src, (int)i, Cin, j.Cin_mem, j.Cout);
  else if (i < plain + rw) v = pack_rw_elem(src, (int)(i - plain), Cin, j.Cin_mem, j.Cout, 3, rw_of(j.Cout));
  else if (i < plain + rw + wino) {
    if (!j.is3d && j.ntaps == 25) {   // k5 s2 as 3x3 over the parity images: 4*Cin logical input channels
      const WSrc ph{j.src, kSrcK5S2Phases, j.Cout, 4 * j.Cin_mem, 9, j.Cin_mem, 0};
      v = (j.mode == kSrcDirect) ? pack_wino_elem(ph, (int)(i - plain - rw), 4 * Cin, j.Cout, 1) : 0.f;
    } else {
      v = pack_wino_elem(src, (int)(i - plain - rw), Cin, j.Cout, j.is3d ? 3 : 1);
    }
  }
  else v = pack_wd_elem(src, (int)(i - plain - rw - wino), Cin);
  j.dst[i] = v;
}

__global__ void pack_weights_kernel(PackJob j) {
  long long plain, rw, wino, wd;
  pack_segments(j.is3d, j.transposed, j.Cin_mem, j.Cout, j.ntaps, &plain, &rw, &wino, &wd);
  const long long total = plain + rw + wino + wd;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) pack_job_elem(j, i);
}

// every weight set of a training step in ONE launch: block -> job through a per-block table (built once, mdf_pack_plan_*)
constexpr int kPackPerBlock = 1024;
__global__ void pack_batch_kernel(const PackJob* __restrict__ jobs, const int* __restrict__ block_job) {
  const PackJob j = jobs[block_job[blockIdx.x]];
  const long long base = (long long)(blockIdx.x - j.blk0) * kPackPerBlock;
#pragma unroll
  for (int k = 0; k < kPackPerBlock / 256; ++k) pack_job_elem(j, base + k * 256 + threadIdx.x);
}

// ---- small and mid-size stride-1 / stride-2 layers whose whole packed weight set fits LDS (16 -> 32: 55 KB, 32 -> 32: 110 KB) ---------
// conv3d_kernel streams every weight fragment through the vector L1 for ONE 16-voxel tile (a fragment is used by 1-2 MFMAs per m-tile
// and n-tile: 512 B per MFMA at 32 -> 32, the whole L1 rate), and its one-tile blocks come in a ragged 1.3-2.7 rounds.  Here a block is
// the 16 waves one CU holds: the weight set is copied to LDS once per block, every wave walks m-tiles (stride = the grid's waves) and
// reads its A fragments with ds_read_b128; the B fragments of the next tap are requested before the MFMAs of the current one.
// Per accumulator the MFMA order is conv3d_kernel's without split-K (tap, cin chunk, k).  Eval only (no epilogue sums).
template <int CIN, int COUT, int MODE, int MT, int ST>
__global__ __launch_bounds__(1024) void conv3d_wlds_kernel(const ConvParams p) {
  constexpr int KPL = (CIN >= 16) ? 4 : 2, CK = 4 * KPL, NCH = CIN / CK, NT = (COUT + 15) / 16;
  constexpr int TAPF = NCH * NT * 64 * KPL;             // floats per tap in the packed set
  static_assert(MODE == kS1 || MODE == kS2, "stride-1 / stride-2 layers");
  extern __shared__ float wsm[];
  {
    const float4* src = reinterpret_cast<const float4*>(p.wpack);
    float4* dst = reinterpret_cast<float4*>(wsm);
    for (int i = threadIdx.x; i < 27 * TAPF / 4; i += 1024) dst[i] = src[i];
  }
  // ST (training): the block's fp64 table of the epilogue sums and the producing layer's (a, b, mean, invstd), as in conv3d_kernel
  __shared__ double st_tab[ST ? 128 : 1];
  __shared__ float st_aux[ST ? 256 : 1];
  if constexpr (ST != 0) {
    if (threadIdx.x < 128) st_tab[threadIdx.x] = 0.0;
    if (p.stat_mode == 2 && threadIdx.x < 256) {
      const int c = threadIdx.x & 63;
      st_aux[threadIdx.x] = (c < COUT) ? p.stat_aux[(threadIdx.x >> 6) * COUT + c] : 0.f;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, n16 = lane & 15;
  float4 ep_al[NT], ep_be[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int r0 = nt * 16 + 4 * q;
    ep_al[nt] = make_float4(1.f, 1.f, 1.f, 1.f);
    ep_be[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 < COUT && p.alpha) {
      ep_al[nt] = *reinterpret_cast<const float4*>(p.alpha + r0);
      ep_be[nt] = *reinterpret_cast<const float4*>(p.beta + r0);
    }
  }
  const float* xq = p.x + KPL * q;
  const float* wl = wsm + lane * KPL;
  const int sd = (MODE == kS2) ? 2 : 1;
  // m-tiles: XCD x (= blockIdx.x & 7) owns a contiguous eighth of them (neighbouring tiles' halos meet in one L2); inside it consecutive
  // tiles go to consecutive blocks, so that every CU is busy whatever the tile count (nblk = number of m-tile groups of MT tiles)
  const unsigned xcd = blockIdx.x & 7u, bx = blockIdx.x >> 3, nbx = gridDim.x >> 3;
  const unsigned cq = p.nblk >> 3, cr = p.nblk & 7u;
  const unsigned c_start = (xcd < cr) ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq, c_len = cq + (xcd < cr ? 1u : 0u);
  for (unsigned l = (unsigned)wave * nbx + bx; l < c_len; l += 16u * nbx) {
    const long long m0 = (long long)(c_start + l) * (MT * 16);
    int in_off[MT];
    unsigned vmask[MT];                     // bits 0-2: kd valid, 3-5: kh valid, 6-8: kw valid
    long long out_vox[MT];
    bool live[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      long long m = m0 + t * 16 + n16;
      live[t] = m < p.m_total;
      if (!live[t]) m = p.m_total - 1;
      const unsigned mu = (unsigned)m, r1 = mu / (unsigned)p.Wo, r2 = r1 / (unsigned)p.Ho;
      const int mw = (int)(mu - r1 * (unsigned)p.Wo), mh = (int)(r1 - r2 * (unsigned)p.Ho);
      const int b = (int)(r2 / (unsigned)p.Do), md = (int)(r2 - (unsigned)b * (unsigned)p.Do);
      const int bd = sd * md, bh = sd * mh, bw = sd * mw;
      unsigned vm = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (bd + k - 1 >= 0 && bd + k - 1 < p.Di) vm |= 1u << k;
        if (bh + k - 1 >= 0 && bh + k - 1 < p.Hi) vm |= 8u << k;
        if (bw + k - 1 >= 0 && bw + k - 1 < p.Wi) vm |= 64u << k;
      }
      vmask[t] = live[t] ? vm : 0u;
      in_off[t] = (int)((((long long)b * p.Di + bd) * p.Hi + bh) * p.Wi + bw) * CIN;
      out_vox[t] = m;
    }
    // depth taps no voxel of this wave has (volumes 1-3 planes deep): skipped, as in conv3d_kernel (kd_any)
    unsigned kd_any = 7u;
    if (p.kd_skip) {
      unsigned m_or = 0u;
#pragma unroll
      for (int t = 0; t < MT; ++t) m_or |= vmask[t];
      kd_any = (__any((int)(m_or & 1u)) ? 1u : 0u) | (__any((int)(m_or & 2u)) ? 2u : 0u) | (__any((int)(m_or & 4u)) ? 4u : 0u);
      kd_any = (unsigned)__builtin_amdgcn_readfirstlane((int)kd_any);
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_b = [&](int kd, int kh, int kw, Frag<KPL> (&bf)[NCH][MT]) {
      const int tapoff = (((kd - 1) * p.Hi + (kh - 1)) * p.Wi + (kw - 1)) * CIN;
      const unsigned need = (1u << kd) | (8u << kh) | (64u << kw);
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int t = 0; t < MT; ++t) {      // (branch-free: see conv3d_kernel)
          const bool ok = (vmask[t] & need) == need;
          bf[ch][t].load(xq + in_off[t] + (ok ? tapoff : 0) + ch * CK);      // (zeroed in mul_tap: a multiply here would wait for the load)
        }
    };
    auto mul_tap = [&](int tap, int kd, int kh, int kw, Frag<KPL> (&bf)[NCH][MT]) {
      const float* wt = wl + tap * TAPF;
      const unsigned need = (1u << kd) | (8u << kh) | (64u << kw);
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const float okf = ((vmask[t] & need) == need) ? 1.0f : 0.0f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) bf[ch][t].scale(okf);
      }
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        Frag<KPL> af[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) af[nt].load(wt + (ch * NT + nt) * (64 * KPL));
#pragma unroll
        for (int s = 0; s < KPL; ++s)
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[nt].v[s], bf[ch][t].v[s], acc[t][nt], 0, 0, 0);
      }
    };
    // groups of RB rows (kd, kh) of three taps; the next group's B fragments are requested before the current group's MFMAs (two register
    // sets).  (8 input channels, 2 registers per fragment: a whole kd plane of 9 taps per group was measured -- 8->16 s2 @24x296x400
    // 44.7 -> 50.0 us, @8x592x800 58.7 -> 64.8: worse; dev: -DMDF_WLDS_RB8=3)
#ifndef MDF_WLDS_RB8
#define MDF_WLDS_RB8 1
#endif
    constexpr int RB = (CIN == 8) ? MDF_WLDS_RB8 : 1;
    struct RowRegs { Frag<KPL> b[3 * RB][NCH][MT]; };
    auto load_row = [&](int r, RowRegs& rr) {
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int kd = (r + i) / 3, kh = (r + i) - 3 * kd;
        load_b(kd, kh, 0, rr.b[3 * i]);
        load_b(kd, kh, 1, rr.b[3 * i + 1]);
        load_b(kd, kh, 2, rr.b[3 * i + 2]);
      }
    };
    auto mul_row = [&](int r, RowRegs& rr) {
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int kd = (r + i) / 3, kh = (r + i) - 3 * kd;
        mul_tap(3 * (r + i), kd, kh, 0, rr.b[3 * i]);
        mul_tap(3 * (r + i) + 1, kd, kh, 1, rr.b[3 * i + 1]);
        mul_tap(3 * (r + i) + 2, kd, kh, 2, rr.b[3 * i + 2]);
      }
    };
    auto next_row = [&](int r) {
      r += RB;
      while (r < 9 && !((kd_any >> (r / 3)) & 1u)) r += RB;
      return r;
    };
    {
      RowRegs ra, rb;
      int r = next_row(-RB);
      if (r < 9) load_row(r, ra);
      while (r < 9) {
        int rn = next_row(r);
        if (rn < 9) load_row(rn, rb);
        __builtin_amdgcn_sched_barrier(0);       // (the requests stay ahead of the MFMAs)
        mul_row(r, ra);
        r = rn;
        if (r >= 9) break;
        rn = next_row(r);
        if (rn < 9) load_row(rn, ra);
        __builtin_amdgcn_sched_barrier(0);
        mul_row(r, rb);
        r = rn;
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int r0 = nt * 16 + 4 * q;
      if (r0 >= COUT) continue;
      const float4 al = ep_al[nt], be = ep_be[nt];
      float ps[4] = {0.f, 0.f, 0.f, 0.f}, pq[4] = {0.f, 0.f, 0.f, 0.f};   // ST: this lane's sums over its m-tiles
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        if (!live[t]) continue;
        float4 o;
        o.x = acc[t][nt][0] * al.x + be.x;
        o.y = acc[t][nt][1] * al.y + be.y;
        o.z = acc[t][nt][2] * al.z + be.z;
        o.w = acc[t][nt][3] * al.w + be.w;
        if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        const size_t oi = (size_t)out_vox[t] * COUT + r0;
        if (p.res) {
          const float4 rr = *reinterpret_cast<const float4*>(p.res + oi);
          o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
        }
        *reinterpret_cast<float4*>(p.y + oi) = o;
        if constexpr (ST != 0) {
          const float ov[4] = {o.x, o.y, o.z, o.w};
          if (p.stat_mode == 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { ps[c] += ov[c]; pq[c] = fmaf(ov[c], ov[c], pq[c]); }
          } else {
            const float4 yv4 = *reinterpret_cast<const float4*>(p.stat_y + oi);
            const float yv[4] = {yv4.x, yv4.y, yv4.z, yv4.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float dr = (fmaf(yv[c], st_aux[r0 + c], st_aux[64 + r0 + c]) > 0.0f) ? ov[c] : 0.0f;
              ps[c] += dr;
              pq[c] = fmaf(dr, (yv[c] - st_aux[128 + r0 + c]) * st_aux[192 + r0 + c], pq[c]);
            }
          }
        }
      }
      if constexpr (ST != 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float a1 = row16_sum(ps[c]), a2 = row16_sum(pq[c]);
          if (n16 == 0) { lds_add_f64(&st_tab[r0 + c], (double)a1); lds_add_f64(&st_tab[64 + r0 + c], (double)a2); }
        }
      }
    }
  }
  if constexpr (ST != 0) {
    __syncthreads();
    mdf::conv_stat_send<COUT>(st_tab, p.stat_out, 2 * COUT, p.stat_slices, blockIdx.x);
  }
}

template <int CIN, int COUT, int MODE, int MT, int ST = 0>
int launch_conv_wlds(ConvParams& p, hipStream_t st) {
  constexpr int KPL = (CIN >= 16) ? 4 : 2, NCH = CIN / (4 * KPL), NT = (COUT + 15) / 16;
  constexpr size_t kLds = (size_t)27 * NCH * NT * 64 * KPL * sizeof(float);
  static_assert(kLds <= 160 * 1024, "the packed weight set must fit LDS");
  auto kern = &conv3d_wlds_kernel<CIN, COUT, MODE, MT, ST>;
  static bool attr_done_dev[64] = {};     // (per-device function attribute: see conv_lds.hip)
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  bool& attr_done = attr_done_dev[(dev_id >= 0 && dev_id < 64) ? dev_id : 0];
  if (!attr_done || dev_id >= 64) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
    if (e != hipSuccess) return mdf::fail(MDF_EHIP, "hipFuncSetAttribute(dynamic LDS %zu): %s", kLds, hipGetErrorString(e));
    attr_done = true;
  }
  p.nblk = (unsigned)((p.m_total + MT * 16 - 1) / (MT * 16));      // wave tiles
  if (ST) p.stat_slices = mdf::conv_stat_slices(256, p.stat_slices);
  hipLaunchKernelGGL(kern, dim3(256), dim3(1024), kLds, st, p);     // one block (16 waves) per CU
  return mdf::check_launch("conv3d_wlds_kernel");
}

template <int CIN, int COUT, int MODE, int MT, int SPLITK, int ST>
int launch_conv(ConvParams& p, hipStream_t st) {
  const long long per_blk = (SPLITK > 1 ? 1LL : 4LL) * MT * 16;
  p.nblk = (unsigned)((p.m_total + per_blk - 1) / per_blk);
  unsigned gx = p.nblk;
  if (ST) {   // persistent above what the chip holds at once (8 blocks per CU)
    const unsigned cap = (MODE == kTr) ? 512u : 2048u;
    if (gx > cap) gx = cap;
    p.stat_slices = mdf::conv_stat_slices((long long)gx * (MODE == kTr ? 4 : 1), p.stat_slices);
  }
  dim3 grid(gx, MODE == kTr ? 4 : 1), block(256);
  hipLaunchKernelGGL((conv3d_kernel<CIN, COUT, MODE, MT, SPLITK, ST>), grid, block, 0, st, p);
  return mdf::check_launch("conv3d_kernel");
}

template <int CIN, int COUT, int MODE, int ST>
int launch_conv_mt(ConvParams& p, hipStream_t st) {
  constexpr int ROWS = (MODE == kTr) ? 2 * COUT : COUT;
  constexpr int MTMAX = (ROWS > 32) ? 2 : 4;
  // small volumes: shrink the wave tile so the grid still covers the 256 CUs a few times over
  const long long tiles_big = p.m_total / (64LL * MTMAX) * (MODE == kTr ? 4 : 1);
  // (8 -> 16 stride 2 at 24x296x400: 51.5 us with 4 m-tiles per wave, 47.8 with 2, 56.1 with 1; the other stride-2 / transposed layers
  //  are best at their MTMAX or indifferent)
  if (MODE == kS2 && CIN == 8 && ST == 0 && tiles_big >= 1024) return launch_conv<CIN, COUT, MODE, 2, 1, ST>(p, st);
  {   // dev A/B: two m-tiles per wave for the mid-size one-tile layers (half the weight bytes through L1 per voxel)
    // (r05: 16 -> 32 stride 2 at 48x148x200 -- 693 four-tile blocks' worth -- 68.6 -> 62.6 us; below ~500 the grid is too small)
    const long long mt2_min = [] { const char* e = getenv("MDF_CONV3D_MT2_MIN_TILES"); return e ? atoll(e) : 512LL; }();
    if (ST == 0 && MODE != kTr && mt2_min >= 0 && tiles_big < 1024 && tiles_big >= mt2_min && !(CIN >= 32 && p.m_total / 16 < 4096))
      return launch_conv<CIN, COUT, MODE, 2, 1, ST>(p, st);
  }
  const long long mt_min = [] { const char* e = getenv("MDF_CONV3D_MT_MIN_TILES"); return e ? atoll(e) : 1024LL; }();   // dev A/B (read per call)
  if (tiles_big >= mt_min) return launch_conv<CIN, COUT, MODE, MTMAX, 1, ST>(p, st);
  // few tiles and a deep K (27*CIN >= 864): split the taps over the block's waves
  const long long tiles_1 = p.m_total / 16 * (MODE == kTr ? 4 : 1);
  if (MODE != kTr && CIN >= 32 && tiles_1 < 4096) {   // (transposed classes have only 2-8 taps)
    // m-tiles per split-K block: with ONE 16-voxel tile a block streams the layer's whole weight set (442 KB at 64 -> 64) through its
    // L1 for 16 voxels -- 1388 blocks x 442 KB = 0.6 GB from the L2s per launch at 12x37x50; two tiles halve that
    // (r05, 64 output channels at ~1400 tiles: 64 -> 64 @12x37x50 62 -> 57 us, 32 -> 64 s2 @24x74x100 35 -> 31; the 32 -> 32 layers and
    //  the volumes under ~500 tiles are indifferent or lose: they keep one tile)
    const int sk_mt = [] { const char* e = getenv("MDF_CONV3D_SK_MT"); return e ? atoi(e) : 0; }();   // dev A/B (read per call): 0 = the rule
    if (ST == 0 && (sk_mt == 2 || (sk_mt == 0 && COUT >= 64)) && tiles_1 >= 1024) return launch_conv<CIN, COUT, MODE, 2, 4, ST>(p, st);
    return launch_conv<CIN, COUT, MODE, 1, 4, ST>(p, st);
  }
  return launch_conv<CIN, COUT, MODE, 1, 1, ST>(p, st);
}

}  // namespace

static int64_t pack_total(int is3d, int transposed, int Cin_mem, int Cout, int ntaps) {
  long long plain, rw, wino, wd;
  pack_segments(is3d, transposed, Cin_mem, Cout, ntaps, &plain, &rw, &wino, &wd);
  return plain + rw + wino + wd;
}

extern "C" int64_t mdf_conv3d_packed_size(int Cin, int Cout) {
  if (Cin < 8 || Cout < 1) return 0;
  const int64_t plain = pack_total(1, 0, Cin, Cout, 27), transposed = pack_total(1, 1, Cin, Cout, 27);
  return plain > transposed ? plain : transposed;   // one size serves both packings
}

static int launch_pack(const PackJob& j, void* stream) {
  hipLaunchKernelGGL(pack_weights_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, j);
  return mdf::check_launch("pack_weights_kernel");
}

extern "C" int mdf_conv3d_pack_weights(const float* w, float* wpack, int Cin, int Cout, int transposed, void* stream) {
  MDF_REQUIRE(w && wpack, "null pointer argument");
  MDF_REQUIRE(Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64, "Cin=%d not in {8,16,32,64}", Cin);
  MDF_REQUIRE(Cout >= 1 && Cout <= 64, "Cout=%d out of range", Cout);
  return launch_pack(PackJob{w, wpack, kSrcDirect, transposed ? 1 : 0, 1, Cin, Cout, 27, 0, 0, 0, 0}, stream);
}

extern "C" int64_t mdf_conv_packed_size(int Cin_mem, int Cout, int ntaps) {
  if (Cin_mem < 1 || Cout < 1 || ntaps < 1) return 0;
  return pack_total(0, 0, Cin_mem, Cout, ntaps);
}

extern "C" int mdf_conv_pack_weights(const float* w, float* wpack, int Cin_mem, int Cout, int ntaps, void* stream) {
  MDF_REQUIRE(w && wpack, "null pointer argument");
  const int Cin = padded_cin(Cin_mem);
  MDF_REQUIRE(Cin == 4 || Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64, "Cin=%d not supported", Cin_mem);
  MDF_REQUIRE(Cout >= 1 && Cout <= 64 && ntaps >= 1 && ntaps <= 27, "Cout=%d ntaps=%d out of range", Cout, ntaps);
  return launch_pack(PackJob{w, wpack, kSrcDirect, 0, 0, Cin_mem, Cout, ntaps, 0, 0, 0, 0}, stream);
}

// ---- batched packing (training: every weight set the step's kernels read, re-packed after each optimizer update) ------
extern "C" int64_t mdf_pack_job_bytes(void) { return (int64_t)sizeof(PackJob); }

extern "C" int64_t mdf_pack_job_fill(void* jobs_host, int index, const float* src, float* dst, int is3d, int transposed, int mode,
                                     int Cin_mem, int Cout, int ntaps, int aux0, int aux1, int first_block) {
  if (!jobs_host || index < 0 || !src || !dst) { mdf::set_error("mdf_pack_job_fill: null pointer or negative index"); return MDF_EARG; }
  const int Cin = padded_cin(Cin_mem);
  if (!(Cin == 4 || Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64) || Cout < 1 || Cout > 64 || ntaps < 1 || ntaps > 27 ||
      (is3d && (ntaps != 27 || Cin_mem < 8)) || (transposed && !is3d) || mode < kSrcDirect || mode > kSrcSwap ||
      (mode == kSrcK5S2Dgrad && (ntaps != 9 || aux0 < 1 || aux1 < 0)) || (mode == kSrcProbSlices && (ntaps != 9 || Cout != 4)) ||
      (mode == kSrcShuffle2 && Cout % 4 != 0)) {
    mdf::set_error("mdf_pack_job_fill: unsupported job (3d=%d tr=%d mode=%d Cin=%d Cout=%d taps=%d)", is3d, transposed, mode, Cin_mem, Cout, ntaps);
    return MDF_EARG;
  }
  const int64_t total = pack_total(is3d, transposed, Cin_mem, Cout, ntaps);
  const int nblk = (int)((total + kPackPerBlock - 1) / kPackPerBlock);
  static_cast<PackJob*>(jobs_host)[index] = PackJob{src, dst, mode, transposed ? 1 : 0, is3d ? 1 : 0, Cin_mem, Cout, ntaps, aux0, aux1, first_block, nblk};
  return nblk;
}

extern "C" int mdf_pack_batch(const void* jobs_dev, const int* block_job_dev, int nblocks, void* stream) {
  MDF_REQUIRE(jobs_dev && block_job_dev, "null pointer argument");
  MDF_REQUIRE(nblocks > 0, "no blocks");
  hipLaunchKernelGGL(pack_batch_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, static_cast<const PackJob*>(jobs_dev), block_job_dev);
  return mdf::check_launch("pack_batch_kernel");
}

using mdf::ConvStat;

int mdf_conv_lds_dispatch(const float* x, const float* wpack, const float* alpha, const float* beta, const float* res,
                          float res_scale, const float* res_up, float* y, int B, int D, int H, int W, int Cin, int Cin_mem, int Cout, int KD,
                          int KHW, int stride, int relu, void* stream, int planar_in, int shuffle2, const ConvStat* stat);

int mdf_conv1x1_dispatch(const float* x, const float* wpack, const float* alpha, const float* beta, const float* res, float res_scale,
                         const float* res_up, float* y, int B, int H, int W, int Cin, int Cout, int relu, void* stream);

static int conv2d_entry(const float* x, const float* wpack, const float* alpha, const float* beta, const float* res,
                        float res_scale, const float* res_up, float* y, int B, int H, int W, int Cin_mem, int Cout, int ksize, int stride,
                        int relu, int planar_in, int pixel_shuffle2, void* stream, const ConvStat* stat) {
  MDF_REQUIRE(x && wpack && y, "null pointer argument");
  MDF_REQUIRE(B > 0 && H > 0 && W > 0, "bad shape");
  MDF_REQUIRE((long long)B * H * W * Cin_mem < (1ll << 31), "input too large for 32-bit offsets");
  MDF_REQUIRE(res_up == nullptr || (stride == 1 && H % 2 == 0 && W % 2 == 0 && Cout % 4 == 0),
              "res_up needs stride 1, even H and W, Cout %% 4 == 0");
  MDF_REQUIRE(!planar_in || Cin_mem < 4, "planar (NCHW) input is supported for Cin < 4 only (the image layer)");
  MDF_REQUIRE(!pixel_shuffle2 || ((Cout == 32 || (Cout == 64 && Cin_mem == 32 && ksize == 3)) && stride == 1 && !res && !res_up && !stat), "pixel_shuffle2 output is built for Cout = 32 (and 32 -> 64 k3), stride 1, no residual, no epilogue sums");
  if (ksize == 1 && stride == 1 && !stat && !planar_in && !pixel_shuffle2) {   // 1x1 layers: streaming kernel (conv1x1.hip)
    const int rc1 = mdf_conv1x1_dispatch(x, wpack, alpha, beta, res, res_scale, res_up, y, B, H, W, Cin_mem, Cout, relu, stream);
    if (rc1 != MDF_EUNSUPPORTED) return rc1;
  }
  const int rc = mdf_conv_lds_dispatch(x, wpack, alpha, beta, res, res_scale, res_up, y, B, 1, H, W, padded_cin(Cin_mem), Cin_mem, Cout, 1,
                                       ksize, stride, relu, stream, planar_in, pixel_shuffle2, stat);
  if (rc == MDF_EUNSUPPORTED)
    return mdf::fail(MDF_EUNSUPPORTED, "conv2d Cin=%d Cout=%d k=%d stride=%d%s is not built", Cin_mem, Cout, ksize, stride, stat ? " (with epilogue sums)" : "");
  return rc;
}

extern "C" int mdf_conv2d_fwd(const float* x, const float* wpack, const float* alpha, const float* beta, const float* res,
                              float res_scale, const float* res_up, float* y, int B, int H, int W, int Cin_mem, int Cout, int ksize, int stride,
                              int relu, int planar_in, int pixel_shuffle2, void* stream) {
  return conv2d_entry(x, wpack, alpha, beta, res, res_scale, res_up, y, B, H, W, Cin_mem, Cout, ksize, stride, relu, planar_in, pixel_shuffle2,
                      stream, nullptr);
}

static int check_stat(int stat_mode, const float* stat_y, const float* stat_aux, const double* stat_out, int nslices, int Cout) {
  MDF_REQUIRE(stat_mode == 1 || stat_mode == 2, "stat_mode=%d not in {1 (sum y, sum y^2), 2 (BatchNorm-backward sums)}", stat_mode);
  MDF_REQUIRE(stat_out, "stat_out is null");
  MDF_REQUIRE(nslices >= 1 && nslices <= 64, "nslices=%d out of range [1,64]", nslices);
  MDF_REQUIRE(stat_mode == 1 || (stat_y && stat_aux), "stat_mode 2 needs stat_y and stat_aux");
  MDF_REQUIRE(Cout % 4 == 0 && Cout >= 8 && Cout <= 64, "epilogue sums are built for Cout in {8,..,64}, Cout %% 4 == 0 (got %d)", Cout);
  return MDF_OK;
}

// Training: the RAW 2-D conv (no scale / shift / ReLU) whose epilogue also accumulates per-channel sums of its output, per
// BatchNorm group (ngroups consecutive sets of B/ngroups images): see ConvParams::stat_mode.  stat_out [nslices][ngroups][2*Cout]
// fp64, zero-initialised by the caller (the sums are the totals over the slices); stat_aux [ngroups][4*Cout].
extern "C" int mdf_conv2d_train_fwd(const float* x, const float* wpack, float* y, int B, int H, int W, int Cin_mem, int Cout, int ksize,
                                    int stride, int planar_in, int stat_mode, const float* stat_y, const float* stat_aux, double* stat_out,
                                    int nslices, int ngroups, void* stream) {
  if (int rc = check_stat(stat_mode, stat_y, stat_aux, stat_out, nslices, Cout)) return rc;
  MDF_REQUIRE(ngroups >= 1 && B % ngroups == 0, "B=%d is not a multiple of ngroups=%d", B, ngroups);
  const ConvStat st{stat_mode, stat_y, stat_aux, stat_out, B / ngroups, nslices};
  return conv2d_entry(x, wpack, nullptr, nullptr, nullptr, 1.0f, nullptr, y, B, H, W, Cin_mem, Cout, ksize, stride, 0, planar_in, 0, stream, &st);
}

#define MDF_CONV_CASE(ci, co, mode)                          \
  if (Cin == ci && Cout == co && m == mode)                  \
    return stat ? launch_conv_mt<ci, co, mode, 1>(p, (hipStream_t)stream) : launch_conv_mt<ci, co, mode, 0>(p, (hipStream_t)stream);

static int conv3d_entry(const float* x, const float* wpack, const float* alpha, const float* beta,
                        const float* res, float* y, int B, int Di, int Hi, int Wi, int Cin, int Cout, int stride,
                        int transposed, int relu, void* stream, const ConvStat* stat) {
  MDF_REQUIRE(x && wpack && y, "null pointer argument");
  MDF_REQUIRE((alpha == nullptr) == (beta == nullptr), "alpha and beta must both be given or both be NULL");
  MDF_REQUIRE(B > 0 && Di > 0 && Hi > 0 && Wi > 0, "bad shape");
  MDF_REQUIRE(stride == 1 || stride == 2, "stride must be 1 or 2");
  MDF_REQUIRE(!transposed || stride == 2, "transposed conv is built for stride 2 only");
  MDF_REQUIRE((long long)B * Di * Hi * Wi * Cin < (1ll << 31), "input volume too large for 32-bit offsets");
  ConvParams p{};
  p.x = x; p.wpack = wpack; p.alpha = alpha; p.beta = beta; p.res = res; p.y = y;
  p.B = B; p.Di = Di; p.Hi = Hi; p.Wi = Wi; p.relu = relu;
  if (stat) { p.stat_mode = stat->mode; p.stat_y = stat->y; p.stat_aux = stat->aux; p.stat_out = stat->out; p.stat_slices = stat->nslices; }
  const int m = transposed ? kTr : (stride == 2 ? kS2 : kS1);
  p.kd_skip = [] { const char* e = getenv("MDF_CONV3D_KDSKIP"); return e ? atoi(e) : 1; }() && Di <= 3;   // dev A/B (read per call)
  if (m == kS1) { p.Do = Di; p.Ho = Hi; p.Wo = Wi; }
  else if (m == kS2) { p.Do = (Di - 1) / 2 + 1; p.Ho = (Hi - 1) / 2 + 1; p.Wo = (Wi - 1) / 2 + 1; }
  else { p.Do = 2 * Di; p.Ho = 2 * Hi; p.Wo = 2 * Wi; }
  p.m_total = (m == kTr) ? (long long)B * Di * Hi * Wi : (long long)B * p.Do * p.Ho * p.Wo;
  const long long lds_min = [] {  // test hook (read per call): MDF_CONV_LDS_MIN_VOXELS=0 forces the LDS kernels at any size
    const char* e = getenv("MDF_CONV_LDS_MIN_VOXELS");
    return e ? atoll(e) : 150000LL;
  }();
  if (m == kS1 && p.m_total >= lds_min) {  // large stride-1 layers: LDS-staged planes (conv_lds.hip)
    const int rc = mdf_conv_lds_dispatch(x, wpack, alpha, beta, res, 1.0f, nullptr, y, B, Di, Hi, Wi, Cin, Cin, Cout, 3, 3, 1, relu, stream, 0, 0, stat);
    if (rc != MDF_EUNSUPPORTED) return rc;
  }
  if (m == kTr && !stat) {   // large transposed layers: all four parity classes per tile (convtr_all_kernel)
    const long long tr_min = [] { const char* e = getenv("MDF_CONVTR_ALL_MIN_VOXELS"); return e ? atoll(e) : 40000LL; }();   // dev A/B (read per call); -1 = never  (r05: 100000 -> 40000, 32->16 @2x148x200 36.7 -> 30.2 us, @6x74x100 29.2 -> 24.7)
    const int ns = [] { const char* e = getenv("MDF_CONVTR_NS"); return e ? atoi(e) : 0; }();   // dev A/B and the equality test (read per call): 0 = the rule, 1 = off
    // one class per persistent block with the class's slots in LDS (convtr_cls_kernel; r05: @12x37x50 42.6 -> 32.9 us, @3x37x50 19.0 -> 14.4;
    // a single-plane volume halves the work of the pd = 1 classes and with it the block shares: @1x74x100 15.3 -> 16.6, so that one stays)
    const int tcl = [] { const char* e = getenv("MDF_CONVTR_CLS"); return e ? atoi(e) : -1; }();   // dev A/B and the equality test (read per call): 0 = off
    if (tr_min >= 0 && Cin == 64 && Cout == 32 && ns == 0) {
      if (tcl == 1) return launch_convtr_cls<64, 32, 12, 4>(p, (hipStream_t)stream);
      if (tcl == 2 || (tcl < 0 && Di >= 2)) return launch_convtr_cls<64, 32, 16, 2>(p, (hipStream_t)stream);
    }
    if (tr_min >= 0 && p.m_total >= tr_min && !(Cin == 64 && Cout == 32 && ns > 1)) {
      // One 16-voxel m-tile per wave (r05; two until then): these layers' time is MFMA time PLUS streaming time (skip + output), and the
      // smaller accumulator set lets 4-5 waves per SIMD instead of 3-4 overlap one block's streaming with another's MFMAs:
      // 32->16 @24x74x100 89.2 -> 69.3 us, 16->8 @12x148x200 51.2 -> 47.4, @4x296x400 79.4 -> 74.7 (four tiles: 88 / 67 / 97)
      const int mt = [] { const char* e = getenv("MDF_CONVTR_MT"); return e ? atoi(e) : 1; }();   // dev A/B (read per call)
      // LDS-weights persistent form (r05): 32->16 @24x74x100 69.5 -> 60.8 us, @2x148x200 29.9 -> 26.1, @6x74x100 24.2 -> 21.3; 16->8 @12x148x200 47.6 -> 44.9, @4x296x400 71.0 -> 59.3
      const int twl = [] { const char* e = getenv("MDF_CONVTR_WLDS"); return e ? atoi(e) : 1; }();   // dev A/B and the equality test (read per call): 0 = off, 2 = fewer waves / two m-tiles
      if (Cin == 16 && Cout == 8 && twl == 1) return launch_convtr_all_wl<16, 8, 1, 16>(p, (hipStream_t)stream);
      if (Cin == 16 && Cout == 8 && twl == 2) return launch_convtr_all_wl<16, 8, 2, 12>(p, (hipStream_t)stream);
      if (Cin == 32 && Cout == 16 && twl == 1) return launch_convtr_all_wl<32, 16, 1, 12>(p, (hipStream_t)stream);
      if (Cin == 32 && Cout == 16 && twl == 2) return launch_convtr_all_wl<32, 16, 1, 8>(p, (hipStream_t)stream);
      if (Cin == 16 && Cout == 8 && mt == 2) return launch_convtr_all<16, 8, 2>(p, (hipStream_t)stream);
      if (Cin == 32 && Cout == 16 && mt == 2) return launch_convtr_all<32, 16, 2>(p, (hipStream_t)stream);
      if (Cin == 16 && Cout == 8) return launch_convtr_all<16, 8, 1>(p, (hipStream_t)stream);
      if (Cin == 32 && Cout == 16) return launch_convtr_all<32, 16, 1>(p, (hipStream_t)stream);
      if (Cin == 64 && Cout == 32) return launch_convtr_all<64, 32, 1>(p, (hipStream_t)stream);
    }
    // The innermost 64 -> 32 layers (1388 / 347 / 463 m-tiles at cfg2: fewer waves than the chip has SIMDs, each with 864 MFMAs behind 27
    // weight taps): the four n-tiles of an m-tile over two or four waves
    // (r05, against conv3d_kernel<kTr>: @12x37x50 53.9 -> 40.5 us over two waves, 38.7 over four; @3x37x50 21.4 -> 16.8 over four)
    if (tr_min >= 0 && Cin == 64 && Cout == 32 && ns != 1 && (p.m_total < tr_min || ns > 1)) {
      if (ns == 2) return launch_convtr_all<64, 32, 1, 2>(p, (hipStream_t)stream);
      return launch_convtr_all<64, 32, 1, 4>(p, (hipStream_t)stream);
    }
  }
  // weights-in-LDS form for the small and mid-size layers whose packed set fits (eval)
  if (stat) {    // training: the same form with the epilogue sums
    const int wlt = [] { const char* e = getenv("MDF_CONV3D_WLDS_TRAIN"); return e ? atoi(e) : 1; }();   // dev A/B (read per call): 0 = off
    if (wlt == 1) {
      if (Cin == 32 && Cout == 32 && m == kS1) return launch_conv_wlds<32, 32, kS1, 1, 1>(p, (hipStream_t)stream);
      if (Cin == 16 && Cout == 16 && m == kS1) return launch_conv_wlds<16, 16, kS1, 1, 1>(p, (hipStream_t)stream);
      if (Cin == 16 && Cout == 32 && m == kS2) return launch_conv_wlds<16, 32, kS2, 1, 1>(p, (hipStream_t)stream);
      if (Cin == 8 && Cout == 16 && m == kS2) return launch_conv_wlds<8, 16, kS2, 2, 1>(p, (hipStream_t)stream);
    }
  }
  if (!stat) {
    const int wl = [] { const char* e = getenv("MDF_CONV3D_WLDS"); return e ? atoi(e) : 1; }();   // dev A/B (read per call): 0 = off
    if (wl == 1) {     // (two m-tiles per wave were measured and lose: 32 -> 32 @6x74x100 35 -> 64 us, 16 -> 32 s2 @48x148x200 61 -> 76)
      if (Cin == 32 && Cout == 32 && m == kS1) return launch_conv_wlds<32, 32, kS1, 1>(p, (hipStream_t)stream);
      if (Cin == 16 && Cout == 32 && m == kS2) return launch_conv_wlds<16, 32, kS2, 1>(p, (hipStream_t)stream);
    }
    const int wl8 = [] { const char* e = getenv("MDF_CONV3D_WLDS8"); return e ? atoi(e) : 2; }();   // dev A/B (read per call): m-tiles per wave, 0 = off
    // (8 -> 16 stride 2, the gather-heavy layers: @24x296x400 50.4 us -> 46.5 / 43.5 / 49.7 with 1 / 2 / 4 m-tiles per wave, @8x592x800 66.0 -> 60.1 / 55.0 / 64.8)
    if (Cin == 8 && Cout == 16 && m == kS2) {
      if (wl8 == 1) return launch_conv_wlds<8, 16, kS2, 1>(p, (hipStream_t)stream);
      if (wl8 == 2) return launch_conv_wlds<8, 16, kS2, 2>(p, (hipStream_t)stream);
      if (wl8 == 4) return launch_conv_wlds<8, 16, kS2, 4>(p, (hipStream_t)stream);
    }
  }
  // stride 1 (every Cin x Cout the nets use)
  MDF_CONV_CASE(32, 16, kS1) MDF_CONV_CASE(16, 16, kS1) MDF_CONV_CASE(32, 32, kS1) MDF_CONV_CASE(64, 64, kS1)
  MDF_CONV_CASE(16, 8, kS1) MDF_CONV_CASE(8, 8, kS1) MDF_CONV_CASE(8, 16, kS1) MDF_CONV_CASE(16, 32, kS1)
  // stride 2
  MDF_CONV_CASE(16, 32, kS2) MDF_CONV_CASE(32, 64, kS2) MDF_CONV_CASE(8, 16, kS2)
  // transposed
  MDF_CONV_CASE(64, 32, kTr) MDF_CONV_CASE(32, 16, kTr) MDF_CONV_CASE(16, 8, kTr)
  return mdf::fail(MDF_EUNSUPPORTED, "conv3d Cin=%d Cout=%d stride=%d transposed=%d is not built", Cin, Cout, stride,
                   transposed);
}

extern "C" int mdf_conv3d_fwd(const float* x, const float* wpack, const float* alpha, const float* beta,
                              const float* res, float* y, int B, int Di, int Hi, int Wi, int Cin, int Cout, int stride,
                              int transposed, int relu, void* stream) {
  return conv3d_entry(x, wpack, alpha, beta, res, y, B, Di, Hi, Wi, Cin, Cout, stride, transposed, relu, stream, nullptr);
}

// Training: the RAW 3-D conv / transposed conv ([res +] conv(x), no scale / shift / ReLU) whose epilogue also accumulates
// per-channel sums of its output (ConvParams::stat_mode) into stat_out [nslices][2*Cout] fp64, zero-initialised by the caller.
// stat_mode 1 serves the forward pass (the layer's batch statistics); stat_mode 2 the backward pass, where this launch is the
// input-gradient conv of the NEXT layer and its output (+ res, the skip gradient) is dz of the layer that produced stat_y.
extern "C" int mdf_conv3d_train_fwd(const float* x, const float* wpack, const float* res, float* y, int B, int Di, int Hi, int Wi, int Cin,
                                    int Cout, int stride, int transposed, int stat_mode, const float* stat_y, const float* stat_aux,
                                    double* stat_out, int nslices, void* stream) {
  if (int rc = check_stat(stat_mode, stat_y, stat_aux, stat_out, nslices, Cout)) return rc;
  const ConvStat st{stat_mode, stat_y, stat_aux, stat_out, 0, nslices};
  return conv3d_entry(x, wpack, nullptr, nullptr, res, y, B, Di, Hi, Wi, Cin, Cout, stride, transposed, 0, stream, &st);
}
