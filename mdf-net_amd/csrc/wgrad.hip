// Weight gradient of the regulariser's 3x3x3 conv layers (net/unit/regular.py:17-43,82-110; backward of nn.Conv3d /
// nn.ConvTranspose3d) on the fp32 matrix cores of gfx950.
//
// One generic correlation covers every layer kind:
//     G[a][b][tap] = sum over voxels o of  small[o][a] * big[s*o + tap - 1][b]        (zero outside `big`)
//   Conv3d  (stride s):      small = dy [.., Cout], big = x  [.., Cin]  -> dW[Cout][Cin][27]   (torch layout)
//   ConvTranspose3d (k3,s2,p1,op1): small = x [.., Cin], big = dy [.., Cout] (twice the size), s = 2
//                                                                          -> dW[Cin][Cout][27]   (torch layout)
// Activations are NDHWC, so the 16 channels of an MFMA row/column are 64 contiguous bytes.
//
// GEMM view: M = a (16 per tile), N = b (16 per tile), K = voxels -- a long K, so the kernel is a split-K GEMM: a
// wave keeps the 9 (kh,kw) tap tiles of ONE (a-tile, b-tile, kd) triple in registers (36 accumulator VGPRs) while it walks its
// share of the voxels in chunks of 16 along w (4 MFMA k-steps of v_mfma_f32_16x16x4_f32; exact fp32 fma chain); per
// chunk the `small` fragment is loaded once and reused by the 27 taps.  Blocks write their partial tiles to a slab,
// a second kernel sums the slab in a few dozen partial sums per element (no storm of float atomics on the 27*A*B hot addresses).
//   A operand: lane l holds small[voxel 4j + (l>>4)][a = l&15]      C/D: lane l holds rows 4*(l>>4)..+3, column l&15
//   B operand: lane l holds big  [voxel 4j + (l>>4)][b = l&15]
#include <cstdlib>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef unsigned u32;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);   // raw buffer, dword format (gfx9)
}
// One operand element: buffer load at {lane-constant byte offset} + {scalar byte offset}.  No per-load address arithmetic on
// the VALU (the first version spent 284 VALU instructions per 12 MFMAs on 64-bit addresses, divisions and predicated loads:
// PMC SQ_INSTS_VALU / SQ_INSTS_MFMA = 23.6); out-of-range offsets (also "negative" ones, which wrap) return 0.
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, u32 lane_off, u32 s_off) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_off, (int)s_off, 0));
}
// The same with a spatial mask (edge chunks only).  The empty asm pins the load: a bare select lets the compiler sink the
// load under the select's condition, i.e. a branch and a wait per load.
__device__ __forceinline__ float bload_masked(__amdgpu_buffer_rsrc_t r, u32 lane_off, u32 s_off, bool ok) {
  float v = bload(r, lane_off, s_off);
  asm volatile("" : "+v"(v));
  return ok ? v : 0.0f;
}

struct WgradParams {
  const float* small_;   // [B,Ds,Hs,Ws,A]
  const float* big;      // [B,Db,Hb,Wb,Bc]
  float* slab;           // [gridDim.x][A][Bc][27]
  int B, Ds, Hs, Ws, Db, Hb, Wb, A, Bc, stride;
  int chunks_per_row;    // ceil(Ws/16)
  long long n_items;     // B*Ds*Hs*chunks_per_row
  int NB;                // b-tiles
  int split;             // waves of a block that share one pair and split the chunks (1, 2 or 4)
  float* zero_out;       // when set: zero_n floats cleared by this launch (the dw the slab is summed into next)
  int zero_n;
};

// the launch also clears the gradient tensor its partial tiles are summed into afterwards (slab_sum_kernel adds with
// atomics): no separate memset launch per layer
__device__ __forceinline__ void zero_slice(float* out, int n) {
  if (!out) return;
  const int nb = gridDim.x * gridDim.y * gridDim.z;
  const int bid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  for (int i = bid * 256 + threadIdx.x; i < n; i += nb * 256) out[i] = 0.f;
}


__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  zero_slice(p.zero_out, p.zero_n);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: the item bookkeeping below runs on the scalar unit
  const int q = lane >> 4, c16 = lane & 15;
  // which (a-tile, b-tile) pair this wave owns, and which slice of the block's chunk stream
  const int pairs_per_block = 4 / p.split;
  const int pair = blockIdx.y * pairs_per_block + wave / p.split;
  const int part = wave % p.split;
  const int na = pair / p.NB, nb = pair % p.NB;
  const int a = na * 16 + c16, bcol = nb * 16 + c16;
  const bool b_ok = bcol < p.Bc;
  // lanes past the channel count read a clamped channel: their tile rows / columns are never stored, no masking needed
  const int ac = min(a, p.A - 1), bc = min(bcol, p.Bc - 1);
  const int s = p.stride;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.small_, (unsigned)((long long)p.B * p.Ds * p.Hs * p.Ws * p.A * 4));
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.big, (unsigned)((long long)p.B * p.Db * p.Hb * p.Wb * p.Bc * 4));
  const u32 la = (u32)((q * p.A + ac) * 4), lb = (u32)((q * s * p.Bc + bc) * 4);

  const int kd = blockIdx.z;          // one depth tap per block: 9 accumulator tiles per wave, 3x the independent work
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Software pipeline over the wave's items: the 40 operand loads of item i+1 are in flight while the 36 MFMAs of item i run
  // (the first version loaded, waited and multiplied one kernel row at a time: three exposed memory latencies per item).
  auto issue = [&](long long item, float (&af)[4], float (&bf)[9][4]) -> int {
    const unsigned iu = (unsigned)item, r1 = iu / (unsigned)p.chunks_per_row, r2 = r1 / (unsigned)p.Hs;      // (< 2^31 items: 32-bit divisions)
    const int ch = (int)(iu - r1 * (unsigned)p.chunks_per_row), oh = (int)(r1 - r2 * (unsigned)p.Hs);
    const int n = (int)(r2 / (unsigned)p.Ds), od = (int)(r2 - (unsigned)n * (unsigned)p.Ds);
    const int id = od * s + kd - 1;
    if (id < 0 || id >= p.Db) return 0;            // wave-uniform
    const int ow0 = ch * 16;
    const u32 srow = (u32)(((((long long)n * p.Ds + od) * p.Hs + oh) * p.Ws + ow0) * p.A * 4);
    const bool edge = (ow0 == 0) || ((ow0 + 16) * s + 1 > p.Wb) || (ow0 + 16 > p.Ws);   // wave-uniform
    if (!edge) {
#pragma unroll
      for (int j = 0; j < 4; ++j) af[j] = bload(rs, la, srow + (u32)(4 * j * p.A * 4));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) af[j] = bload_masked(rs, la, srow + (u32)(4 * j * p.A * 4), ow0 + 4 * j + q < p.Ws);
    }
    int mask = 0;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh * s + kh - 1;
      if (ih < 0 || ih >= p.Hb) continue;        // wave-uniform
      mask |= 1 << kh;
      const long long brow = ((((long long)n * p.Db + id) * p.Hb + ih) * p.Wb + (long long)ow0 * s - 1) * p.Bc * 4;   // tap kw = 0 of voxel ow0
      if (!edge) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[kh * 3 + kw][j] = bload(rb, lb, (u32)(brow + (long long)(4 * j * s + kw) * p.Bc * 4));
      } else {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // (the hardware adds lane and scalar offsets WITHOUT wrapping, so a "negative" scalar part is out of range for every
            //  lane: at the row start the offset is formed per lane from the clamped coordinate instead)
            const int iw = (ow0 + 4 * j + q) * s + kw - 1;
            const int base = max(ow0 * s - 1, 0);             // first voxel any lane of this chunk can need
            bf[kh * 3 + kw][j] = bload_masked(rb, (u32)(((max(iw, 0) - base) * p.Bc + bc) * 4),
                                              (u32)(brow + (long long)(base - (ow0 * s - 1)) * p.Bc * 4), iw >= 0 && iw < p.Wb);
          }
      }
    }
    return mask;
  };
  auto compute = [&](const float (&af)[4], const float (&bf)[9][4], int mask) {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      if (!((mask >> kh) & 1)) continue;         // wave-uniform
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[kh * 3 + kw][j], acc[kh * 3 + kw], 0, 0, 0);
    }
  };
  {
    const long long step = (long long)gridDim.x * p.split;
    long long it = (long long)blockIdx.x * p.split + part;
    float a0[4], b0[9][4], a1[4], b1[9][4];
    int m0 = (it < p.n_items) ? issue(it, a0, b0) : 0, m1 = 0;
    while (it < p.n_items) {
      const long long nx = it + step;
      m1 = (nx < p.n_items) ? issue(nx, a1, b1) : 0;
      compute(a0, b0, m0);
      if (nx >= p.n_items) break;
      it = nx + step;
      m0 = (it < p.n_items) ? issue(it, a0, b0) : 0;
      compute(a1, b1, m1);
    }
  }

  // partial tiles of the `split` waves that share a pair: summed through LDS (one 27-KB buffer per pair, the sharing
  // waves take turns), then one slab write per pair.  Every wave runs the same barrier sequence.
  __shared__ float red[2][9 * 4 * 64];
  float* mine = red[(wave / p.split) & 1];
  for (int turn = 1; turn < p.split; ++turn) {
    if (part == turn) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) mine[(t * 4 + i) * 64 + lane] = acc[t][i];
    }
    __syncthreads();
    if (part == 0) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] += mine[(t * 4 + i) * 64 + lane];
    }
    __syncthreads();
  }
  if (part == 0 && b_ok) {
    float* out = p.slab + (long long)blockIdx.x * p.A * p.Bc * 27;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = na * 16 + 4 * q + i;
      if (row < p.A) {
        float* o = out + ((long long)row * p.Bc + bcol) * 27 + kd * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) o[t] = acc[t][i];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// A = 1 (the `prob` conv, regular.py:43/110): a 1-row `small` would use 1/16 of every MFMA tile.  Re-index the sum by the
// voxel of `big` instead:  dw[0][b][kd][kh][kw] = sum_o' small[o' - (0,kh-1,kw-1)] * big[o' + (kd-1,0,0)][b], and put the
// nine (kh,kw) shifts of `small` in the ROWS of the tile: one MFMA step yields all nine taps of a depth tap (27 -> 3 MFMA
// steps per chunk, `big` read once per kd instead of nine times).  blockIdx.z = kd; the 4 waves of a block split the chunks.
__global__ __launch_bounds__(256) void wgrad_a1_kernel(const WgradParams p) {
  zero_slice(p.zero_out, p.zero_n);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, c16 = lane & 15;
  const int kd = blockIdx.z;
  const int kh = c16 / 3, kw = c16 % 3;            // row r = c16 < 9
  const bool r_ok = c16 < 9, b_ok = c16 < p.Bc;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (long long item = (long long)blockIdx.x * 4 + wave; item < p.n_items; item += (long long)gridDim.x * 4) {
    const int ch = (int)(item % p.chunks_per_row);
    long long r = item / p.chunks_per_row;
    const int h = (int)(r % p.Hs); r /= p.Hs;
    const int d = (int)(r % p.Ds);
    const int n = (int)(r / p.Ds);
    const int id = d + kd - 1;
    if (id < 0 || id >= p.Ds) continue;            // wave-uniform
    const int hs = h - (kh - 1);                   // row of `small` this lane's tile row reads
    const bool hs_ok = r_ok && hs >= 0 && hs < p.Hs;
    const float* srow = p.small_ + (((long long)n * p.Ds + d) * p.Hs + (hs_ok ? hs : 0)) * (long long)p.Ws;
    const float* brow = p.big + (((long long)n * p.Ds + id) * p.Hs + h) * (long long)p.Ws * p.Bc;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int w = ch * 16 + 4 * j + q;
      const int wsm = w - (kw - 1);
      const float av = srow[min(max(wsm, 0), p.Ws - 1)];
      const float bl = brow[min(w, p.Ws - 1) * p.Bc + min(c16, p.Bc - 1)];
      const float a = av * ((hs_ok && w < p.Ws && wsm >= 0 && wsm < p.Ws) ? 1.0f : 0.0f);
      const float bv = bl * ((b_ok && w < p.Ws) ? 1.0f : 0.0f);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc, 0, 0, 0);
    }
  }
  __shared__ float red[4][4 * 64];
#pragma unroll
  for (int i = 0; i < 4; ++i) red[wave][i * 64 + lane] = acc[i];
  __syncthreads();
  if (wave == 0 && b_ok) {
    float* out = p.slab + (long long)blockIdx.x * p.Bc * 27;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * q + i;
      if (row < 9) out[(long long)c16 * 27 + kd * 9 + row] = red[0][i * 64 + lane] + red[1][i * 64 + lane] + red[2][i * 64 + lane] + red[3][i * 64 + lane];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// A = 1 with 8 or 16 `big` channels -- the `prob` conv as the model has it -- on the vector ALUs.  The sum is 0.4-0.9 GFLOP behind
// 32-60 MB of operands: an HBM stream, not a GEMM.  The MFMA form above feeds every 16x16x4 step with two dword loads per lane (75 /
// 63 / 36 us at cfg3: 0.4 TB/s).  Here a thread owns a voxel v of `big` and four of its channels: ONE 16-byte load of big[v], the 27
// neighbours small[v - tap] (dword loads shared by the lanes of a voxel and contiguous along w: cache hits), 27 x 4 fused
// multiply-adds into 108 accumulators, over a grid-stride loop; the lanes of a 16-lane row that own the same channel quad add up with
// DPP row shifts (as ds_bpermute shuffles the reduction cost more than the loop: 39 -> 25 us), the 16 rows through LDS, and the block
// writes one slab.  cfg3: 75 / 64 / 35 us -> 25 / 23 / 23 us.
template <int CTRL>
__device__ __forceinline__ float dpp_row_shr(float v) {     // lane l reads lane l - n of its 16-lane row (0 where there is none)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

template <int BC>
__global__ __launch_bounds__(256) void wgrad_a1_valu_kernel(const WgradParams p) {
  zero_slice(p.zero_out, p.zero_n);
  constexpr int NQ = BC / 4;               // channel quads per voxel = lanes per voxel
  constexpr int VPB = 256 / NQ;            // voxels per block and iteration
  const int tid = threadIdx.x;
  const int cq = tid % NQ;
  const int nvox = p.B * p.Ds * p.Hs * p.Ws;            // (the entry point has checked that the operands fit 32-bit offsets)
  float acc[27][4];
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = 0.0f;
  const int plane = p.Hs * p.Ws;
  for (int v = blockIdx.x * VPB + tid / NQ; v < nvox; v += gridDim.x * VPB) {
    const int w = v % p.Ws;
    int r = v / p.Ws;
    const int h = r % p.Hs; r /= p.Hs;
    const int d = r % p.Ds;
    const float4 bv = *reinterpret_cast<const float4*>(p.big + (size_t)v * BC + 4 * cq);
    const float* sv0 = p.small_ + v;                     // small[v]; the taps are small[v - (kd-1)*plane - (kh-1)*Ws - (kw-1)]
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int od = d - (kd - 1);         // dw[kd][kh][kw] += small[o] * big[o + (kd-1, kh-1, kw-1)]  with  o = v - tap
      const bool dok = od >= 0 && od < p.Ds;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int oh = h - (kh - 1);
        const bool hok = dok && oh >= 0 && oh < p.Hs;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ow = w - (kw - 1);
          const bool ok = hok && ow >= 0 && ow < p.Ws;
          const int delta = (kd - 1) * plane + (kh - 1) * p.Ws + (kw - 1);
          const float sv = sv0[ok ? -delta : 0];
          const float a = ok ? sv : 0.0f;
          const int t = kd * 9 + kh * 3 + kw;
          acc[t][0] = fmaf(a, bv.x, acc[t][0]);
          acc[t][1] = fmaf(a, bv.y, acc[t][1]);
          acc[t][2] = fmaf(a, bv.z, acc[t][2]);
          acc[t][3] = fmaf(a, bv.w, acc[t][3]);
        }
      }
    }
  }
  // lanes of a 16-lane row that own the same channel quad: shifted adds on the vector ALU (DPP row_shr with zero fill -- an inclusive
  // scan with steps NQ, 2NQ, .. whose last NQ lanes end up with the row's totals); the 16 rows of the block meet in LDS
  __shared__ float red[16][NQ][108];
  const int row = tid >> 4;
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float x = acc[t][i];
      if constexpr (NQ == 2) x += dpp_row_shr<0x112>(x);
      x += dpp_row_shr<0x114>(x);
      x += dpp_row_shr<0x118>(x);
      if ((tid & 15) >= 16 - NQ) red[row][(tid & 15) - (16 - NQ)][t * 4 + i] = x;
    }
  __syncthreads();
  float* out = p.slab + (long long)blockIdx.x * BC * 27;                      // [A = 1][Bc][27]
  for (int e = tid; e < BC * 27; e += 256) {
    const int c = e / 27, t = e - c * 27;
    const int k = t * 4 + (c & 3);
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += red[r][c >> 2][k];
    out[e] = s;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// 2-D layers of the feature pyramid (net/unit/backbone.py:17-45: Conv2d k3 s1 / k5 s2, pad (k-1)/2), NHWC:
//     dw[a][b][kh][kw] = sum over pixels o of  small[o][a] * big[s*o + (kh,kw) - pad][b]
// Same split-K scheme; blockIdx.z = kernel row kh, a wave keeps the KS tap tiles of that row in registers.
struct Wgrad2dParams {
  const float* small_;   // [B,Hs,Ws,A]
  const float* big;      // [B,Hb,Wb,Bc]
  float* slab;           // [gridDim.x][A][Bc][KS*KS]
  int B, Hs, Ws, Hb, Wb, A, Bc, stride;
  int chunks_per_row;
  long long n_items;     // B*Hs*chunks_per_row
  int NB, split;
  float* zero_out;
  int zero_n;
};

template <int KS>
__global__ __launch_bounds__(256) void wgrad2d_kernel(const Wgrad2dParams p) {
  zero_slice(p.zero_out, p.zero_n);
  constexpr int PAD = (KS - 1) / 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, c16 = lane & 15;
  const int pairs_per_block = 4 / p.split;
  const int pair = blockIdx.y * pairs_per_block + wave / p.split;
  const int part = wave % p.split;
  const int na = pair / p.NB, nb = pair % p.NB;
  const int a = na * 16 + c16, bcol = nb * 16 + c16;
  const bool b_ok = bcol < p.Bc;
  const int ac = min(a, p.A - 1), bc = min(bcol, p.Bc - 1);
  const int kh = blockIdx.z;
  const int s = p.stride;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.small_, (unsigned)((long long)p.B * p.Hs * p.Ws * p.A * 4));
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.big, (unsigned)((long long)p.B * p.Hb * p.Wb * p.Bc * 4));
  const u32 la = (u32)((q * p.A + ac) * 4), lb = (u32)((q * s * p.Bc + bc) * 4);
  f32x4 acc[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto issue = [&](long long item, float (&af)[4], float (&bf)[KS][4]) -> int {
    const unsigned iu = (unsigned)item, r1 = iu / (unsigned)p.chunks_per_row;      // (< 2^31 items: 32-bit divisions)
    const int ch = (int)(iu - r1 * (unsigned)p.chunks_per_row), n = (int)(r1 / (unsigned)p.Hs), oh = (int)(r1 - (unsigned)n * (unsigned)p.Hs);
    const int ih = oh * s + kh - PAD;
    if (ih < 0 || ih >= p.Hb) return 0;            // wave-uniform
    const int ow0 = ch * 16;
    const u32 srow = (u32)((((long long)n * p.Hs + oh) * p.Ws + ow0) * p.A * 4);
    const long long brow = (((long long)n * p.Hb + ih) * p.Wb + (long long)ow0 * s - PAD) * p.Bc * 4;      // tap kw = 0 of voxel ow0
    const bool edge = (ow0 * s < PAD) || ((ow0 + 16) * s + PAD > p.Wb) || (ow0 + 16 > p.Ws);              // wave-uniform
    if (!edge) {
#pragma unroll
      for (int j = 0; j < 4; ++j) af[j] = bload(rs, la, srow + (u32)(4 * j * p.A * 4));
#pragma unroll
      for (int kw = 0; kw < KS; ++kw)
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[kw][j] = bload(rb, lb, (u32)(brow + (long long)(4 * j * s + kw) * p.Bc * 4));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) af[j] = bload_masked(rs, la, srow + (u32)(4 * j * p.A * 4), ow0 + 4 * j + q < p.Ws);
#pragma unroll
      for (int kw = 0; kw < KS; ++kw)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int iw = (ow0 + 4 * j + q) * s + kw - PAD;      // (offsets do not wrap: formed per lane from the clamped coordinate)
          const int base = max(ow0 * s - PAD, 0);             // first pixel any lane of this chunk can need
          bf[kw][j] = bload_masked(rb, (u32)(((max(iw, 0) - base) * p.Bc + bc) * 4), (u32)(brow + (long long)(base - (ow0 * s - PAD)) * p.Bc * 4),
                                   iw >= 0 && iw < p.Wb);
        }
    }
    return 1;
  };
  auto compute = [&](const float (&af)[4], const float (&bf)[KS][4], int mask) {
    if (!mask) return;                             // wave-uniform
#pragma unroll
    for (int kw = 0; kw < KS; ++kw)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[kw] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[kw][j], acc[kw], 0, 0, 0);
  };
  {   // software pipeline: item i+1's loads in flight under item i's MFMAs
    const long long step = (long long)gridDim.x * p.split;
    long long it = (long long)blockIdx.x * p.split + part;
    float a0[4], b0[KS][4], a1[4], b1[KS][4];
    int m0 = (it < p.n_items) ? issue(it, a0, b0) : 0, m1 = 0;
    while (it < p.n_items) {
      const long long nx = it + step;
      m1 = (nx < p.n_items) ? issue(nx, a1, b1) : 0;
      compute(a0, b0, m0);
      if (nx >= p.n_items) break;
      it = nx + step;
      m0 = (it < p.n_items) ? issue(it, a0, b0) : 0;
      compute(a1, b1, m1);
    }
  }
  __shared__ float red[2][KS * 4 * 64];
  float* mine = red[(wave / p.split) & 1];
  for (int turn = 1; turn < p.split; ++turn) {
    if (part == turn) {
#pragma unroll
      for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) mine[(t * 4 + i) * 64 + lane] = acc[t][i];
    }
    __syncthreads();
    if (part == 0) {
#pragma unroll
      for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] += mine[(t * 4 + i) * 64 + lane];
    }
    __syncthreads();
  }
  if (part == 0 && b_ok) {
    float* out = p.slab + (long long)blockIdx.x * p.A * p.Bc * (KS * KS);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = na * 16 + 4 * q + i;
      if (row < p.A) {
        float* o = out + ((long long)row * p.Bc + bcol) * (KS * KS) + kh * KS;
#pragma unroll
        for (int t = 0; t < KS; ++t) o[t] = acc[t][i];
      }
    }
  }
}

// out[i] (+)= sum over slabs: blockIdx.y takes every gridDim.y-th slab, partial sums meet in `out` through fp32 atomics
// (a few thousand per launch, spread over n addresses)
__global__ void slab_sum_kernel(const float* __restrict__ slab, int nslab, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s0 = 0.f, s1 = 0.f;
  int k = blockIdx.y;
  for (; k + (int)gridDim.y < nslab; k += 2 * gridDim.y) {
    s0 += slab[(long long)k * n + i];
    s1 += slab[(long long)(k + gridDim.y) * n + i];
  }
  if (k < nslab) s0 += slab[(long long)k * n + i];
  unsafeAtomicAdd(&out[i], s0 + s1);
}

// The same for MANY layers in one launch: the partial tiles of a training step's ~60 weight gradients are not needed before
// the optimizer, so their sums wait for the end of the backward pass (59 launches of ~5 us -> 1).  Jobs travel BY VALUE in the
// kernel arguments (the slabs and gradient tensors are fresh allocations every step: no device-side table to refresh).
constexpr int kSumJobs = 96;
struct SumJobs {
  const float* slab[kSumJobs];
  float* out[kSumJobs];
  int nslab[kSumJobs], n[kSumJobs], gys[kSumJobs];
  int blk0[kSumJobs + 1];       // first block of job j; block = (chunk of 256 elements) * gys + slab slice
  int njobs;
};
__global__ void slab_sum_batch_kernel(const SumJobs J) {
  __shared__ int job_s;
  if (threadIdx.x == 0) {
    int lo = 0, hi = J.njobs - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (J.blk0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    job_s = lo;
  }
  __syncthreads();
  const int j = job_s;
  const int local = blockIdx.x - J.blk0[j];
  const int gys = J.gys[j], ys = local % gys, chunk = local / gys;
  const int n = J.n[j], nslab = J.nslab[j];
  const int i = chunk * 256 + threadIdx.x;
  if (i >= n) return;
  const float* slab = J.slab[j];
  float s0 = 0.f, s1 = 0.f;
  int k = ys;
  for (; k + gys < nslab; k += 2 * gys) {
    s0 += slab[(long long)k * n + i];
    s1 += slab[(long long)(k + gys) * n + i];
  }
  if (k < nslab) s0 += slab[(long long)k * n + i];
  unsafeAtomicAdd(&J.out[j][i], s0 + s1);
}

}  // namespace

extern "C" int mdf_wgrad_sum_batch(const float* const* slabs, float* const* outs, const int* nslabs, const int* ns, int njobs, void* stream) {
  MDF_REQUIRE(slabs && outs && nslabs && ns, "null pointer argument");
  MDF_REQUIRE(njobs >= 1, "no jobs");
  for (int first = 0; first < njobs; first += kSumJobs) {
    SumJobs J{};
    J.njobs = (njobs - first < kSumJobs) ? njobs - first : kSumJobs;
    int blk = 0;
    for (int j = 0; j < J.njobs; ++j) {
      const int g = first + j;
      MDF_REQUIRE(slabs[g] && outs[g] && nslabs[g] >= 1 && ns[g] >= 1, "job %d: bad slab / out / counts", g);
      J.slab[j] = slabs[g]; J.out[j] = outs[g]; J.nslab[j] = nslabs[g]; J.n[j] = ns[g];
      int gys = nslabs[g] / 8;                  // >= 8 slabs per partial sum
      if (gys < 1) gys = 1;
      if (gys > 32) gys = 32;
      J.gys[j] = gys;
      J.blk0[j] = blk;
      blk += ((ns[g] + 255) / 256) * gys;
    }
    J.blk0[J.njobs] = blk;
    hipLaunchKernelGGL(slab_sum_batch_kernel, dim3(blk), dim3(256), 0, (hipStream_t)stream, J);
    if (int rc = mdf::check_launch("slab_sum_batch_kernel")) return rc;
  }
  return MDF_OK;
}

int mdf_wgrad_lds_dispatch(const float* small_, const float* big, float* workspace, int* gx_io, int B, int Ds, int Hs, int Ws, int A, int Bc,
                           int stride, int ksize, int is3d, float* zero_out, int zero_n, void* stream);
// blocks per launch: every block writes one partial tile set to the slab (PMC: 2 GB written + 2 GB re-read per cfg3 step at
// 2048 blocks), so no more blocks than it takes to fill the chip a few times over
static int wgrad_target_blocks() {
  static const int n = [] { const char* e = getenv("MDF_WGRAD_BLOCKS"); return (e && atoi(e) > 0) ? atoi(e) : 1024; }();
  return n;
}
static long long wgrad_slab_cap_bytes() {     // partial-tile bytes per launch (written by the blocks, read again by the sum)
  static const long long n = [] { const char* e = getenv("MDF_WGRAD_SLAB_MIB"); return (long long)((e && atoi(e) > 0) ? atoi(e) : 16) << 20; }();   // dev A/B
  return n;
}
static bool wgrad_use_lds() {
  static const bool on = [] { const char* e = getenv("MDF_WGRAD_LDS"); return e ? atoi(e) != 0 : true; }();   // dev A/B
  return on;
}

extern "C" int64_t mdf_conv3d_wgrad_workspace(int B, int Ds, int Hs, int Ws, int A, int Bc) {
  if (B < 1 || Ds < 1 || Hs < 1 || Ws < 1 || A < 1 || Bc < 1) return 0;
  const long long items = (long long)B * Ds * Hs * ((Ws + 15) / 16);
  const int pairs = ((A + 15) / 16) * ((Bc + 15) / 16);
  const int split = pairs >= 4 ? 1 : (pairs == 2 ? 2 : 4);
  const int gy = (pairs * split + 3) / 4;
  // enough blocks to fill the chip several times over (3 depth taps x gy pair groups x g), at least `split` chunks each
  long long g = wgrad_target_blocks() / (3 * gy);
  if (g > items / split) g = items / split;
  // every block writes a partial tile set (A*Bc*27 floats: 442 KB for 64 x 64 channels) that is read again by the sum: <= 16 MiB of
  // them per launch (r03 sweep, scripts/diag_wgrad_sweep.sh: 64 x 64 @12x18x24 45 -> 37 us, 32 x 32 @24x36x48 50 -> 44 us with
  // half the blocks; the 16 x 16 layers are not touched by the cap)
  const long long slab_cap = wgrad_slab_cap_bytes() / ((long long)A * Bc * 27 * 4);
  if (g > slab_cap && slab_cap >= 16) g = slab_cap;
  if (g < 1) g = 1;
  if (A == 1 && (Bc == 8 || Bc == 16)) {      // wgrad_a1_valu_kernel: one block per slab, a slab is 216 / 432 floats
    const long long vpb = 256 / (Bc / 4);
    long long gv = ((long long)B * Ds * Hs * Ws + vpb - 1) / vpb;
    if (gv > 512) gv = 512;
    if (gv > g) g = gv;
  }
  return g * A * Bc * 27;   // floats
}

static int conv3d_wgrad_impl(const float* small_, const float* big, float* dw, float* workspace, int B, int Ds, int Hs, int Ws,
                             int A, int Bc, int stride, int accumulate, int* nslab_out, void* stream) {
  MDF_REQUIRE(small_ && big && dw && workspace, "null pointer argument");
  MDF_REQUIRE(B > 0 && Ds > 0 && Hs > 0 && Ws > 0, "bad shape");
  MDF_REQUIRE(stride == 1 || stride == 2, "stride must be 1 or 2");
  MDF_REQUIRE(A >= 1 && A <= 64 && Bc >= 1 && Bc <= 64, "channel counts out of range (A=%d, B=%d)", A, Bc);
  MDF_REQUIRE((long long)B * Ds * Hs * Ws * A * 4 < (1ll << 32) && (long long)B * Ds * Hs * Ws * stride * stride * stride * Bc * 4 < (1ll << 32),
              "operand volumes must fit 32-bit byte offsets");
  WgradParams p{};
  p.small_ = small_; p.big = big; p.slab = workspace;
  p.B = B; p.Ds = Ds; p.Hs = Hs; p.Ws = Ws; p.Db = Ds * stride; p.Hb = Hs * stride; p.Wb = Ws * stride; p.A = A; p.Bc = Bc; p.stride = stride;
  p.chunks_per_row = (Ws + 15) / 16;
  p.n_items = (long long)B * Ds * Hs * p.chunks_per_row;
  const int NA = (A + 15) / 16;
  p.NB = (Bc + 15) / 16;
  const int pairs = NA * p.NB;
  p.split = pairs >= 4 ? 1 : (pairs == 2 ? 2 : 4);
  const int gy = (pairs * p.split + 3) / 4;
  const int gx = (int)(mdf_conv3d_wgrad_workspace(B, Ds, Hs, Ws, A, Bc) / ((long long)A * Bc * 27));
  const int n = A * Bc * 27;
  p.zero_out = accumulate ? nullptr : dw;
  p.zero_n = n;
  int gx_used = gx;
  int rc_lds = MDF_EUNSUPPORTED;
  static const bool a1_valu = [] { const char* e = getenv("MDF_WGRAD_A1_VALU"); return e ? atoi(e) != 0 : true; }();   // dev A/B
  if (A == 1 && (Bc == 8 || Bc == 16) && stride == 1 && a1_valu) {
    const long long vpb = 256 / (Bc / 4);
    long long g = ((long long)B * Ds * Hs * Ws + vpb - 1) / vpb;     // at most one voxel group per block and iteration ...
    if (g > 512) g = 512;                                            // ... and two blocks per CU (r03 sweep: 256 / 512 / 1024 / 2048 blocks -> 48 / 25 / 30 / 36 us @8x288x384: the per-block reduction)
    if (g > gx) g = gx;                                              // (the workspace holds gx slabs)
    gx_used = (int)g;
    if (Bc == 8) hipLaunchKernelGGL(wgrad_a1_valu_kernel<8>, dim3(gx_used), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(wgrad_a1_valu_kernel<16>, dim3(gx_used), dim3(256), 0, (hipStream_t)stream, p);
  } else if (A == 1 && Bc <= 16 && stride == 1) {
    hipLaunchKernelGGL(wgrad_a1_kernel, dim3(gx, 1, 3), dim3(256), 0, (hipStream_t)stream, p);
  } else {
    if (wgrad_use_lds()) {
      gx_used = gx;
      rc_lds = mdf_wgrad_lds_dispatch(small_, big, workspace, &gx_used, B, Ds, Hs, Ws, A, Bc, stride, 3, 1, p.zero_out, n, stream);
      if (rc_lds != MDF_OK && rc_lds != MDF_EUNSUPPORTED) return rc_lds;
    }
    if (rc_lds == MDF_EUNSUPPORTED) {
      gx_used = gx;
      hipLaunchKernelGGL(wgrad_kernel, dim3(gx, gy, 3), dim3(256), 0, (hipStream_t)stream, p);
    }
  }
  if (int rc = mdf::check_launch("wgrad_kernel")) return rc;
  if (nslab_out) { *nslab_out = gx_used; return MDF_OK; }      // the caller sums the partial tiles later (mdf_wgrad_sum_batch)
  int gys = gx_used / 8;                  // >= 8 slabs per partial sum
  if (gys < 1) gys = 1;
  if (gys > 32) gys = 32;
  hipLaunchKernelGGL(slab_sum_kernel, dim3((n + 255) / 256, gys), dim3(256), 0, (hipStream_t)stream, workspace, gx_used, n, dw);
  return mdf::check_launch("slab_sum_kernel");
}

extern "C" int mdf_conv3d_wgrad(const float* small_, const float* big, float* dw, float* workspace, int B, int Ds, int Hs, int Ws,
                                int A, int Bc, int stride, int accumulate, void* stream) {
  return conv3d_wgrad_impl(small_, big, dw, workspace, B, Ds, Hs, Ws, A, Bc, stride, accumulate, nullptr, stream);
}

// The same without the final sum: dw is cleared, the partial tiles stay in `workspace` as *nslab_out slabs of A*Bc*27 floats
// ([A][Bc][27] each) for a later mdf_wgrad_sum_batch (one launch for all the layers of a backward pass).
extern "C" int mdf_conv3d_wgrad_partial(const float* small_, const float* big, float* dw, float* workspace, int B, int Ds, int Hs, int Ws,
                                        int A, int Bc, int stride, int* nslab_out, void* stream) {
  MDF_REQUIRE(nslab_out, "nslab_out is null");
  return conv3d_wgrad_impl(small_, big, dw, workspace, B, Ds, Hs, Ws, A, Bc, stride, 0, nslab_out, stream);
}

static long long wgrad2d_grid(int B, int Hs, int Ws, int A, int Bc, int ksize, int* split_out, int* gy_out) {
  const long long items = (long long)B * Hs * ((Ws + 15) / 16);
  const int pairs = ((A + 15) / 16) * ((Bc + 15) / 16);
  const int split = pairs >= 4 ? 1 : (pairs == 2 ? 2 : 4);
  const int gy = (pairs * split + 3) / 4;
  long long g = wgrad_target_blocks() / (ksize * gy);
  if (g > items / split) g = items / split;
  const long long slab_cap = wgrad_slab_cap_bytes() / ((long long)A * Bc * ksize * ksize * 4);
  if (g > slab_cap && slab_cap >= 16) g = slab_cap;
  if (g < 1) g = 1;
  if (split_out) *split_out = split;
  if (gy_out) *gy_out = gy;
  return g;
}

extern "C" int64_t mdf_conv2d_wgrad_workspace(int B, int Hs, int Ws, int A, int Bc, int ksize) {
  if (B < 1 || Hs < 1 || Ws < 1 || A < 1 || Bc < 1 || ksize < 1) return 0;
  return wgrad2d_grid(B, Hs, Ws, A, Bc, ksize, nullptr, nullptr) * A * Bc * ksize * ksize;
}

static int conv2d_wgrad_impl(const float* small_, const float* big, float* dw, float* workspace, int B, int Hs, int Ws, int A, int Bc,
                             int ksize, int stride, int accumulate, int* nslab_out, void* stream) {
  MDF_REQUIRE(small_ && big && dw && workspace, "null pointer argument");
  MDF_REQUIRE(B > 0 && Hs > 0 && Ws > 0, "bad shape");
  MDF_REQUIRE(stride == 1 || stride == 2, "stride must be 1 or 2");
  MDF_REQUIRE(ksize == 1 || ksize == 3 || ksize == 5, "ksize=%d not in {1,3,5}", ksize);
  MDF_REQUIRE(A >= 1 && A <= 64 && Bc >= 1 && Bc <= 64, "channel counts out of range (A=%d, B=%d)", A, Bc);
  MDF_REQUIRE((long long)B * Hs * Ws * A * 4 < (1ll << 32) && (long long)B * Hs * Ws * stride * stride * Bc * 4 < (1ll << 32),
              "operand maps must fit 32-bit byte offsets");
  Wgrad2dParams p{};
  p.small_ = small_; p.big = big; p.slab = workspace;
  p.B = B; p.Hs = Hs; p.Ws = Ws; p.Hb = Hs * stride; p.Wb = Ws * stride; p.A = A; p.Bc = Bc; p.stride = stride;
  p.chunks_per_row = (Ws + 15) / 16;
  p.n_items = (long long)B * Hs * p.chunks_per_row;
  p.NB = (Bc + 15) / 16;
  int gy = 1;
  const int gx = (int)wgrad2d_grid(B, Hs, Ws, A, Bc, ksize, &p.split, &gy);
  const dim3 grid(gx, gy, ksize);
  hipStream_t st = (hipStream_t)stream;
  const int n = A * Bc * ksize * ksize;
  p.zero_out = accumulate ? nullptr : dw;
  p.zero_n = n;
  int gx_used = gx, rc_lds = MDF_EUNSUPPORTED;
  if (wgrad_use_lds()) {
    gx_used = gx;
    rc_lds = mdf_wgrad_lds_dispatch(small_, big, workspace, &gx_used, B, 1, Hs, Ws, A, Bc, stride, ksize, 0, p.zero_out, n, stream);
    if (rc_lds != MDF_OK && rc_lds != MDF_EUNSUPPORTED) return rc_lds;
  }
  if (rc_lds == MDF_EUNSUPPORTED) {
    gx_used = gx;
    if (ksize == 1) hipLaunchKernelGGL(wgrad2d_kernel<1>, grid, dim3(256), 0, st, p);
    else if (ksize == 3) hipLaunchKernelGGL(wgrad2d_kernel<3>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(wgrad2d_kernel<5>, grid, dim3(256), 0, st, p);
    if (int rc = mdf::check_launch("wgrad2d_kernel")) return rc;
  }
  if (nslab_out) { *nslab_out = gx_used; return MDF_OK; }
  int gys = gx_used / 8;
  if (gys < 1) gys = 1;
  if (gys > 32) gys = 32;
  hipLaunchKernelGGL(slab_sum_kernel, dim3((n + 255) / 256, gys), dim3(256), 0, st, workspace, gx_used, n, dw);
  return mdf::check_launch("slab_sum_kernel");
}

extern "C" int mdf_conv2d_wgrad(const float* small_, const float* big, float* dw, float* workspace, int B, int Hs, int Ws, int A, int Bc,
                                int ksize, int stride, int accumulate, void* stream) {
  return conv2d_wgrad_impl(small_, big, dw, workspace, B, Hs, Ws, A, Bc, ksize, stride, accumulate, nullptr, stream);
}

extern "C" int mdf_conv2d_wgrad_partial(const float* small_, const float* big, float* dw, float* workspace, int B, int Hs, int Ws, int A, int Bc,
                                        int ksize, int stride, int* nslab_out, void* stream) {
  MDF_REQUIRE(nslab_out, "nslab_out is null");
  return conv2d_wgrad_impl(small_, big, dw, workspace, B, Hs, Ws, A, Bc, ksize, stride, 0, nslab_out, stream);
}

