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
};

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
template <int CIN, int COUT, int MODE, int MT, int SPLITK>
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
  const unsigned tile_blk = mdf::xcd_remap(blockIdx.x, p.nblk);
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
    const int mw = (int)(m % Mw);
    long long r = m / Mw;
    const int mh = (int)(r % Mh);
    r /= Mh;
    const int md = (int)(r % Md);
    const int b = (int)(r / Md);
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
  for (int j = (SPLITK > 1 ? wave : 0); j < ntaps; j += (SPLITK > 1 ? SPLITK : 1)) {
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
    const int tapoff = ((od * p.Hi + oh) * p.Wi + ow) * CIN;
    const unsigned need = (1u << kd) | (8u << kh) | (64u << kw);
    const float* wt = wl + (size_t)tap * (NCH * NT * 64 * KPL);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      Frag<KPL> bf[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        // branch-free: an out-of-range tap reads the voxel's own (valid) position and is zeroed by a multiply.  A predicated
        // load compiles to a branch and a wait per load (s_and_saveexec / s_cbranch_execz), which serialises the fetches of
        // a tap step; a select instead of the multiply lets the compiler sink the load back under the condition.
        const bool ok = (vmask[t] & need) == need;
        bf[t].load(xq + in_off[t] + (ok ? tapoff : 0) + ch * CK);
        bf[t].scale(ok ? 1.0f : 0.0f);
      }
      Frag<KPL> af[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) af[nt].load(wt + (size_t)(ch * NT + nt) * (64 * KPL));
#pragma unroll
      for (int s = 0; s < KPL; ++s)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            // transposed, offset +1: rows of parity pw = 0 are structurally zero -> skip n-tiles made only of such rows
            if (MODE == kTr && ow == 1 && (nt + 1) * 16 <= COUT) continue;
            acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[nt].v[s], bf[t].v[s], acc[t][nt], 0, 0, 0);
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
    }
  }
}

// torch weights -> fragment order  wpack[tap][chunk][nt][q][n][s] = W(cout = nt*16+n, cin = chunk*CK + KPL*q + s, tap)
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cin_mem, int Cout,
                                    int ntaps, int transposed) {
  const int KPL = (Cin >= 16) ? 4 : (Cin == 8 ? 2 : 1), CK = 4 * KPL, NCH = Cin / CK, NT = (Cout + 15) / 16;
  const int total = ntaps * NCH * NT * 64 * KPL;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int r = i;
    const int s = r % KPL; r /= KPL;
    const int n = r % 16; r /= 16;
    const int qq = r % 4; r /= 4;
    const int nt = r % NT; r /= NT;
    const int ch = r % NCH; r /= NCH;
    const int tap = r;
    const int cout = nt * 16 + n, cin = ch * CK + KPL * qq + s;
    float v = 0.f;
    if (cout < Cout && cin < Cin_mem)
      v = transposed ? w[((size_t)cin * Cout + cout) * ntaps + tap] : w[((size_t)cout * Cin_mem + cin) * ntaps + tap];
    wp[i] = v;
  }
}

// w-phase packing for Cout < 16 (conv_lds.hip, Cfg::RW): the conv rewritten with GEMM row r*Cout + c = channel c of output
// phase r (RW phases along w) and KW' = KHW + RW - 1 taps along w; tap kw' of phase r is the original tap kw' - r.
// wp[tap' = kdh*KW' + kw'][chunk][q][n][s], one n-tile.
__global__ void pack_weights_rw_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cin_mem, int Cout, int nkdh,
                                       int KHW, int RW) {
  const int KPL = (Cin >= 16) ? 4 : (Cin == 8 ? 2 : 1), CK = 4 * KPL, NCH = Cin / CK, KW = KHW + RW - 1;
  const int total = nkdh * KW * NCH * 64 * KPL;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int r = i;
    const int s = r % KPL; r /= KPL;
    const int n = r % 16; r /= 16;
    const int qq = r % 4; r /= 4;
    const int ch = r % NCH; r /= NCH;
    const int kwp = r % KW, kdh = r / KW;
    const int phase = n / Cout, cout = n % Cout, cin = ch * CK + KPL * qq + s;
    const int kw = kwp - phase;
    float v = 0.f;
    if (phase < RW && cin < Cin_mem && kw >= 0 && kw < KHW) v = w[((size_t)cout * Cin_mem + cin) * (nkdh * KHW) + kdh * KHW + kw];
    wp[i] = v;
  }
}

// Winograd F(2x2,3x3) weights for conv_lds.hip step_wino: U[kd][a][b] = G g[kd] G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1];
// fragments in the order the kernel walks them: [kd][chunk][ab = a*4+b][nt][lane = q*16+m][s], cout = nt*16+m, cin = chunk*16+4q+s.
__global__ void pack_weights_wino_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int nkd) {
  const int NCH = Cin / 16, NT = (Cout + 15) / 16;
  const int total = nkd * NCH * 16 * NT * 64 * 4;
  const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
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
      const float* g = w + (((size_t)cout * Cin + cin) * nkd + kd) * 9;
      for (int y = 0; y < 3; ++y)
        for (int x = 0; x < 3; ++x) v += G[a][y] * G[b][x] * g[y * 3 + x];
    }
    wp[i] = v;
  }
}
static bool wino_built(int Cin, int Cout) { return ((Cout == 16 || Cout == 32) && (Cin == 16 || Cin == 32) && Cout <= Cin) || (Cin == 16 && Cout == 8); }   // 3-D
static bool wino2d_built(int Cin, int Cout) { return (Cin == 16 && Cout == 16) || (Cin == 32 && Cout == 32) || (Cin == 64 && Cout == 64); }

// ConvTranspose3d weights [Cin][Cout][3][3][3] -> wpack[tap' = (kd*3+kh)*2+ow][chunk][nt][q][n][s] with GEMM row
// r = nt*16+n = pw*Cout + cout and kernel tap kw(pw, ow): (0,0)->1, (1,0)->2, (1,1)->0, (0,1)-> structurally zero.
__global__ void pack_weights_tr_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout) {
  const int KPL = (Cin >= 16) ? 4 : 2, CK = 4 * KPL, NCH = Cin / CK, NT = (2 * Cout + 15) / 16;
  const int total = 18 * NCH * NT * 64 * KPL;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int r = i;
    const int s = r % KPL; r /= KPL;
    const int n = r % 16; r /= 16;
    const int qq = r % 4; r /= 4;
    const int nt = r % NT; r /= NT;
    const int ch = r % NCH; r /= NCH;
    const int tap = r;
    const int kd = tap / 6, kh = (tap / 2) % 3, ow = tap & 1;
    const int row = nt * 16 + n, cin = ch * CK + KPL * qq + s;
    float v = 0.f;
    if (row < 2 * Cout) {
      const int pw = row / Cout, cout = row % Cout;
      const int kw = (pw == 0) ? (ow == 0 ? 1 : -1) : (ow == 0 ? 2 : 0);
      if (kw >= 0) v = w[((size_t)cin * Cout + cout) * 27 + (kd * 3 + kh) * 3 + kw];
    }
    wp[i] = v;
  }
}

template <int CIN, int COUT, int MODE, int MT, int SPLITK>
int launch_conv(ConvParams& p, hipStream_t st) {
  const long long per_blk = (SPLITK > 1 ? 1LL : 4LL) * MT * 16;
  p.nblk = (unsigned)((p.m_total + per_blk - 1) / per_blk);
  dim3 grid(p.nblk, MODE == kTr ? 4 : 1), block(256);
  hipLaunchKernelGGL((conv3d_kernel<CIN, COUT, MODE, MT, SPLITK>), grid, block, 0, st, p);
  return mdf::check_launch("conv3d_kernel");
}

template <int CIN, int COUT, int MODE>
int launch_conv_mt(ConvParams& p, hipStream_t st) {
  constexpr int ROWS = (MODE == kTr) ? 2 * COUT : COUT;
  constexpr int MTMAX = (ROWS > 32) ? 2 : 4;
  // small volumes: shrink the wave tile so the grid still covers the 256 CUs a few times over
  const long long tiles_big = p.m_total / (64LL * MTMAX) * (MODE == kTr ? 4 : 1);
  if (tiles_big >= 1024) return launch_conv<CIN, COUT, MODE, MTMAX, 1>(p, st);
  // few tiles and a deep K (27*CIN >= 864): split the taps over the block's waves
  const long long tiles_1 = p.m_total / 16 * (MODE == kTr ? 4 : 1);
  if (MODE != kTr && CIN >= 32 && tiles_1 < 4096) return launch_conv<CIN, COUT, MODE, 1, 4>(p, st);  // (transposed classes have only 2-8 taps)
  return launch_conv<CIN, COUT, MODE, 1, 1>(p, st);
}

}  // namespace

// w-phase factor of a stride-1 k3 layer (0 = none): Cout 8 -> 2 outputs per MFMA column, Cout 4 -> 4
static int rw_of(int Cout) { return Cout == 8 ? 2 : (Cout == 4 ? 4 : 0); }

extern "C" int64_t mdf_conv3d_packed_size(int Cin, int Cout) {
  if (Cin < 8 || Cout < 1) return 0;
  int64_t plain = (int64_t)27 * Cin * (((Cout + 15) / 16) * 16);
  if (rw_of(Cout)) plain += (int64_t)9 * (3 + rw_of(Cout) - 1) * Cin * 16;   // + the w-phase packing behind the plain one
  if (wino_built(Cin, Cout)) plain += (int64_t)48 * Cin * (((Cout + 15) / 16) * 16);   // + the Winograd-domain weights
  const int64_t transposed = (int64_t)18 * Cin * (((2 * Cout + 15) / 16) * 16);
  return plain > transposed ? plain : transposed;   // one size serves both packings
}

extern "C" int mdf_conv3d_pack_weights(const float* w, float* wpack, int Cin, int Cout, int transposed, void* stream) {
  MDF_REQUIRE(w && wpack, "null pointer argument");
  MDF_REQUIRE(Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64, "Cin=%d not in {8,16,32,64}", Cin);
  MDF_REQUIRE(Cout >= 1 && Cout <= 64, "Cout=%d out of range", Cout);
  if (transposed) {
    hipLaunchKernelGGL(pack_weights_tr_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, w, wpack, Cin, Cout);
  } else {
    hipLaunchKernelGGL(pack_weights_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, w, wpack, Cin, Cin, Cout, 27, 0);
    if (rw_of(Cout))
      hipLaunchKernelGGL(pack_weights_rw_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, w, wpack + (size_t)27 * Cin * 16, Cin, Cin, Cout, 9, 3,
                         rw_of(Cout));
    if (wino_built(Cin, Cout))
      hipLaunchKernelGGL(pack_weights_wino_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, w,
                         wpack + (size_t)27 * Cin * (((Cout + 15) / 16) * 16) + (rw_of(Cout) ? (size_t)36 * Cin * 16 : 0), Cin, Cout, 3);
  }
  return mdf::check_launch("pack_weights_kernel");
}

static int padded_cin(int c) { return c <= 4 ? 4 : c; }

extern "C" int64_t mdf_conv_packed_size(int Cin_mem, int Cout, int ntaps) {
  if (Cin_mem < 1 || Cout < 1 || ntaps < 1) return 0;
  int64_t n = (int64_t)ntaps * padded_cin(Cin_mem) * (((Cout + 15) / 16) * 16);
  if (ntaps == 9 && rw_of(Cout)) n += (int64_t)3 * (3 + rw_of(Cout) - 1) * padded_cin(Cin_mem) * 16;   // + w-phase packing (3x3 layers)
  if (ntaps == 9 && wino2d_built(Cin_mem, Cout)) n += (int64_t)16 * Cin_mem * (((Cout + 15) / 16) * 16);   // + Winograd-domain weights
  return n;
}

extern "C" int mdf_conv_pack_weights(const float* w, float* wpack, int Cin_mem, int Cout, int ntaps, void* stream) {
  MDF_REQUIRE(w && wpack, "null pointer argument");
  const int Cin = padded_cin(Cin_mem);
  MDF_REQUIRE(Cin == 4 || Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64, "Cin=%d not supported", Cin_mem);
  MDF_REQUIRE(Cout >= 1 && Cout <= 64 && ntaps >= 1 && ntaps <= 27, "Cout=%d ntaps=%d out of range", Cout, ntaps);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, w, wpack, Cin, Cin_mem, Cout, ntaps, 0);
  if (ntaps == 9 && rw_of(Cout))
    hipLaunchKernelGGL(pack_weights_rw_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, w, wpack + (size_t)9 * Cin * 16, Cin, Cin_mem, Cout, 3, 3,
                       rw_of(Cout));
  if (ntaps == 9 && wino2d_built(Cin_mem, Cout))
    hipLaunchKernelGGL(pack_weights_wino_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, w, wpack + (size_t)9 * Cin * (((Cout + 15) / 16) * 16),
                       Cin, Cout, 1);
  return mdf::check_launch("pack_weights_kernel");
}

int mdf_conv_lds_dispatch(const float* x, const float* wpack, const float* alpha, const float* beta, const float* res,
                          float res_scale, const float* res_up, float* y, int B, int D, int H, int W, int Cin, int Cin_mem, int Cout, int KD,
                          int KHW, int stride, int relu, void* stream, int planar_in, int shuffle2);

extern "C" int mdf_conv2d_fwd(const float* x, const float* wpack, const float* alpha, const float* beta, const float* res,
                              float res_scale, const float* res_up, float* y, int B, int H, int W, int Cin_mem, int Cout, int ksize, int stride,
                              int relu, int planar_in, int pixel_shuffle2, void* stream) {
  MDF_REQUIRE(x && wpack && y, "null pointer argument");
  MDF_REQUIRE(B > 0 && H > 0 && W > 0, "bad shape");
  MDF_REQUIRE((long long)B * H * W * Cin_mem < (1ll << 31), "input too large for 32-bit offsets");
  MDF_REQUIRE(res_up == nullptr || (stride == 1 && H % 2 == 0 && W % 2 == 0 && Cout % 4 == 0),
              "res_up needs stride 1, even H and W, Cout %% 4 == 0");
  MDF_REQUIRE(!planar_in || Cin_mem < 4, "planar (NCHW) input is supported for Cin < 4 only (the image layer)");
  MDF_REQUIRE(!pixel_shuffle2 || (Cout == 32 && stride == 1 && !res && !res_up), "pixel_shuffle2 output is built for Cout = 32, stride 1, no residual");
  const int rc = mdf_conv_lds_dispatch(x, wpack, alpha, beta, res, res_scale, res_up, y, B, 1, H, W, padded_cin(Cin_mem), Cin_mem, Cout, 1,
                                       ksize, stride, relu, stream, planar_in, pixel_shuffle2);
  if (rc == MDF_EUNSUPPORTED)
    return mdf::fail(MDF_EUNSUPPORTED, "conv2d Cin=%d Cout=%d k=%d stride=%d is not built", Cin_mem, Cout, ksize, stride);
  return rc;
}

#define MDF_CONV_CASE(ci, co, mode)                          \
  if (Cin == ci && Cout == co && m == mode) return launch_conv_mt<ci, co, mode>(p, (hipStream_t)stream);

extern "C" int mdf_conv3d_fwd(const float* x, const float* wpack, const float* alpha, const float* beta,
                              const float* res, float* y, int B, int Di, int Hi, int Wi, int Cin, int Cout, int stride,
                              int transposed, int relu, void* stream) {
  MDF_REQUIRE(x && wpack && y, "null pointer argument");
  MDF_REQUIRE((alpha == nullptr) == (beta == nullptr), "alpha and beta must both be given or both be NULL");
  MDF_REQUIRE(B > 0 && Di > 0 && Hi > 0 && Wi > 0, "bad shape");
  MDF_REQUIRE(stride == 1 || stride == 2, "stride must be 1 or 2");
  MDF_REQUIRE(!transposed || stride == 2, "transposed conv is built for stride 2 only");
  MDF_REQUIRE((long long)B * Di * Hi * Wi * Cin < (1ll << 31), "input volume too large for 32-bit offsets");
  ConvParams p{};
  p.x = x; p.wpack = wpack; p.alpha = alpha; p.beta = beta; p.res = res; p.y = y;
  p.B = B; p.Di = Di; p.Hi = Hi; p.Wi = Wi; p.relu = relu;
  const int m = transposed ? kTr : (stride == 2 ? kS2 : kS1);
  if (m == kS1) { p.Do = Di; p.Ho = Hi; p.Wo = Wi; }
  else if (m == kS2) { p.Do = (Di - 1) / 2 + 1; p.Ho = (Hi - 1) / 2 + 1; p.Wo = (Wi - 1) / 2 + 1; }
  else { p.Do = 2 * Di; p.Ho = 2 * Hi; p.Wo = 2 * Wi; }
  p.m_total = (m == kTr) ? (long long)B * Di * Hi * Wi : (long long)B * p.Do * p.Ho * p.Wo;
  static const long long lds_min = [] {  // test hook: MDF_CONV_LDS_MIN_VOXELS=0 forces the LDS kernel at any size
    const char* e = getenv("MDF_CONV_LDS_MIN_VOXELS");
    return e ? atoll(e) : 150000LL;
  }();
  if (m == kS1 && p.m_total >= lds_min) {  // large stride-1 layers: LDS-staged planes (conv_lds.hip)
    const int rc = mdf_conv_lds_dispatch(x, wpack, alpha, beta, res, 1.0f, nullptr, y, B, Di, Hi, Wi, Cin, Cin, Cout, 3, 3, 1, relu, stream, 0, 0);
    if (rc != MDF_EUNSUPPORTED) return rc;
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
