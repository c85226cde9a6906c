// The two full-resolution layers of the feature pyramid as ONE kernel (net/unit/backbone.py:28: conv01 = ConvBNReLU(3, 8, k3)
// -> ConvBNReLU(8, 8, k3), eval mode, BatchNorm folded): the 8-channel full-resolution map between them (303 MB per 5-view
// item at 1600x1184, written by one launch and read back by the next) never leaves the CU.
//
// A block owns a strip of 62 output pixels x a segment of rows of one image and walks down the rows 8 at a time with two
// rolling windows in LDS: the input rows (planar RGB as the loader hands it over, + halo) and the layer-1 output rows.
// Per step: the 8 new input rows are committed, every wave computes 2 rows of layer 1 into the intermediate window (ReLU and
// the zero padding of layer 2 applied), barrier, every wave computes 2 output rows of layer 2 and stores them (NHWC).
// Both layers use the w-phase form of conv_lds.hip (Cfg::RW = 2): an MFMA column is 2 neighbouring output pixels, GEMM row
// = phase * 8 + cout, 4 taps along w -- the same packed weights, tap order and epilogue arithmetic as the two single-layer
// launches, so the result is BIT-IDENTICAL to them (tests/test_conv2d_gpu.py).
// A strip is 62 pixels because its intermediate row is then 64 = 2 MFMA tiles (a 64-pixel strip would need 66: a third,
// almost empty tile per row of layer 1).
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kOW = 62;          // output pixels per strip
constexpr int kIW = 66;          // input pixels staged per row: [w0 - 2, w0 + 64)
constexpr int kRing = 16;        // rows per rolling window (>= 10 live rows; power of two)
constexpr int kRowsStep = 8;     // rows per step (2 per wave)
constexpr int kS1 = 67;          // floats per (cin, slot) row of the input window: odd, so the q groups of a ds_read_b32 fall on disjoint banks
constexpr int kS2 = 66;          // float2 per (cin pair, slot) row of the intermediate window: 64 computed + 2 columns only the discarded MFMA column reads
constexpr int kP1 = kRing * kS1 + 1;   // cin-plane stride of the input window: odd, so lane groups q and q+1 (one cin apart) hit odd / even banks
constexpr int kInFloats = 4 * kP1;
constexpr int kMidFloats = 4 * kRing * kS2 * 2;

struct PairParams {
  const float* x;     // [N,3,H,W] planar
  const float* w1;    // w-phase packing of layer 1 (cin 4(3), cout 8): [12 taps][64 lanes]
  const float* a1;    // [8] folded BN scale / shift
  const float* b1;
  const float* w2;    // w-phase packing of layer 2 (cin 8, cout 8): [12 taps][64 lanes][2]
  const float* a2;
  const float* b2;
  float* y;           // [N,H,W,8]
  int N, H, W, strips, segs, seg_rows;
};

__global__ __launch_bounds__(256) void conv_pair_kernel(const PairParams p) {
  // (+ 6 KB of padding: 56 KB per block keeps TWO blocks per CU -- with three the kernel is 12 % slower, measured twice)
  __shared__ __attribute__((aligned(16))) float in_img[kInFloats + 1536];
  __shared__ __attribute__((aligned(16))) float mid_img[kMidFloats];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, n16 = lane & 15;
  int bid = blockIdx.x;
  const int sx = bid % p.strips; bid /= p.strips;
  const int seg = bid % p.segs;
  const int n = bid / p.segs;
  const int w0 = sx * kOW;
  const int r0 = seg * p.seg_rows, r1 = min(p.H, r0 + p.seg_rows);
  if (r0 >= r1) return;

  // weights of both layers and the epilogue constants of this lane's 4 output channels stay in registers
  float w1r[12], w2r[12][2];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    w1r[i] = p.w1[i * 64 + lane];
    const float2 t = *reinterpret_cast<const float2*>(p.w2 + (i * 64 + lane) * 2);
    w2r[i][0] = t.x; w2r[i][1] = t.y;
  }
  const int c0 = 4 * (q & 1), phase = q >> 1;
  float a1r[4], b1r[4], a2r[4], b2r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { a1r[k] = p.a1[c0 + k]; b1r[k] = p.b1[c0 + k]; a2r[k] = p.a2[c0 + k]; b2r[k] = p.b2[c0 + k]; }

  // the zero channel of the 3 -> 4 padded input (every slot, once)
  for (int i = tid; i < kRing * kS1; i += 256) in_img[3 * kP1 + i] = 0.f;

  const float* xin = p.x + (size_t)n * 3 * p.H * p.W;
  auto load_elem = [&](int e, int first_row, int nrows) -> float {     // element e of rows [first_row, first_row + nrows): (c, rr, px), px fastest
    const int c = e / (nrows * kIW), rem = e - c * (nrows * kIW);
    const int rr = rem / kIW, px = rem - rr * kIW;
    const int gy = first_row + rr, gx = w0 - 2 + px;
    return (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? xin[((size_t)c * p.H + gy) * p.W + gx] : 0.f;
  };
  auto store_elem = [&](int e, int first_row, int nrows, float v) {
    const int c = e / (nrows * kIW), rem = e - c * (nrows * kIW);
    const int rr = rem / kIW, px = rem - rr * kIW;
    in_img[c * kP1 + ((first_row + rr) & (kRing - 1)) * kS1 + px] = v;
  };

  // prologue: rows r0-2 .. r0+1 straight into the window; rows r0+2 .. r0+9 take the pipelined route below
  for (int e = tid; e < 4 * 3 * kIW; e += 256) store_elem(e, r0 - 2, 4, load_elem(e, r0 - 2, 4));
  // The 8-row transfers of the steps: element e = tid + 256 k of a step is always the same (channel, row-in-step, pixel), so its
  // decomposition, global offset and window offset are computed ONCE (as per-step index arithmetic -- two divisions and the
  // bounds logic per element -- they were ~350 VALU instructions per step next to 144 MFMAs per wave: (3 us MFMA-bound -> 4.3 us)
  constexpr int NPF = (kRowsStep * 3 * kIW + 255) / 256;      // 7
  int pf_g[NPF], pf_l[NPF], pf_rr[NPF];       // global offset at first_row = 0 (or -1: never loaded), window offset without the row slot, row in step
#pragma unroll
  for (int k = 0; k < NPF; ++k) {
    const int e = tid + 256 * k;
    const int c = e / (kRowsStep * kIW), rem = e - c * (kRowsStep * kIW);
    const int rr = rem / kIW, px = rem - rr * kIW;
    const int gx = w0 - 2 + px;
    const bool ok = e < kRowsStep * 3 * kIW && gx >= 0 && gx < p.W;
    pf_rr[k] = (e < kRowsStep * 3 * kIW) ? rr : -1;
    pf_g[k] = ok ? (c * p.H + rr) * p.W + gx : -1;
    pf_l[k] = c * kP1 + px;
  }
  auto fetch = [&](int first_row, float (&pf)[NPF]) {
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int gy = first_row + pf_rr[k];
      pf[k] = (pf_g[k] >= 0 && gy >= 0 && gy < p.H) ? xin[pf_g[k] + first_row * p.W] : 0.f;
    }
  };
  auto commit = [&](int first_row, const float (&pf)[NPF]) {
#pragma unroll
    for (int k = 0; k < NPF; ++k)
      if (pf_rr[k] >= 0) in_img[pf_l[k] + ((first_row + pf_rr[k]) & (kRing - 1)) * kS1] = pf[k];
  };
  float pf[NPF];
  fetch(r0 + 2, pf);

  // layer 1: intermediate row m, both MFMA tiles -> mid window (zero outside the image: layer 2's padding)
  auto layer1_row = [&](int m) {
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const float* row = in_img + q * kP1 + ((m + kh - 1) & (kRing - 1)) * kS1 + n16 * 2;
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1r[kh * 4 + kw], row[t * 32 + kw], acc[t], 0, 0, 0);
      }
    }
    const bool row_in = (m >= 0 && m < p.H);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ix = (t * 16 + n16) * 2 + phase;           // pixel w0 - 1 + ix
      const int gx = w0 - 1 + ix;
      const bool ok = row_in && gx >= 0 && gx < p.W;
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float o = acc[t][k] * a1r[k] + b1r[k];
        o = fmaxf(o, 0.f);
        v[k] = ok ? o : 0.f;
      }
      float* dst = mid_img + (((c0 >> 1) * kRing + (m & (kRing - 1))) * kS2 + ix) * 2;      // cin pair c0/2, then c0/2 + 1
      *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
      *reinterpret_cast<float2*>(dst + kRing * kS2 * 2) = make_float2(v[2], v[3]);
    }
  };
  // layer 2: output row o
  auto layer2_row = [&](int o) {
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const float* row = mid_img + ((q * kRing + ((o + kh - 1) & (kRing - 1))) * kS2 + n16 * 2) * 2;
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        float2 bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bv[t] = *reinterpret_cast<const float2*>(row + (t * 32 + kw) * 2);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2r[kh * 4 + kw][0], bv[t].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2r[kh * 4 + kw][1], bv[t].y, acc[t], 0, 0, 0);
      }
    }
    if (o >= r1) return;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ox = (t * 16 + n16) * 2 + phase;
      const int gx = w0 + ox;
      if (ox >= kOW || gx >= p.W) continue;
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float r = acc[t][k] * a2r[k] + b2r[k];
        v[k] = fmaxf(r, 0.f);
      }
      *reinterpret_cast<float4*>(p.y + (((size_t)n * p.H + o) * p.W + gx) * 8 + c0) = make_float4(v[0], v[1], v[2], v[3]);
    }
  };

  for (int R = r0; R < r1; R += kRowsStep) {
    commit(R + 2, pf);            // input rows R+2 .. R+9
    __syncthreads();
    if (R + kRowsStep < r1) fetch(R + 2 + kRowsStep, pf);     // next step's rows in flight during the MFMAs below
    if (R == r0 && wave < 2) layer1_row(r0 - 1 + wave);       // the two rows above the first step's own eight
    layer1_row(R + 1 + 2 * wave);
    layer1_row(R + 2 + 2 * wave);
    __syncthreads();
    layer2_row(R + 2 * wave);
    layer2_row(R + 2 * wave + 1);
  }
}


// ------------------------------------------------------------------------------------------------------------------
// The same pair on the VECTOR ALUs (r05).  With 3 / 8 input and 8 output channels the MFMA form above executes 2304 multiply-adds
// per pixel for the 1584 the two layers need (a quarter of the w-phase taps and of layer 1's K are structural zeros), and on gfx950
// packed fp32 FMAs run at the fp32 MFMA rate (256 flop / clk / CU either way, profiles/r05_mfma_issue.md): here every multiply-add
// is a useful one.  A WAVE owns a strip of 64 columns (62 outputs + the two halo columns of the intermediate map) and a segment of
// rows; a lane owns one column and walks down the rows:
//   * the 3 x 3 x 3 input window and the 3 x 3 x 8 window of layer-1 outputs live in registers (rows rotate by unrolling the row
//     loop three times; the left / right neighbours of a new layer-1 row arrive through DPP wave shifts), nothing goes through LDS;
//   * weights are read with scalar loads from the plain fragments of the two packed sets ([tap][cin][cout 16] and
//     [tap][cin pair][cout 16][2]) and enter v_pk_fma_f32 as SGPR pairs: layer 1 pairs two output channels per instruction, layer 2
//     two input channels (its accumulators are (even-cin, odd-cin) partial sums, added at the end);
//   * the next input row's 9 loads (3 channels x 3 columns, coalesced rows of the planar image) are in flight one row ahead.
// Sums are formed in a different order than in the MFMA form: equal to ~1e-6 relative, not bit for bit (tests/test_conv2d_gpu.py).
// Measured at cfg2 (5 x 1184 x 1600): 253 -> 230 us.  ~600 vector instructions per row of 62 pixels (396 packed FMAs) would be ~165 us at
// the full issue rate; the vector pipe is ~70 % busy at the 3 waves per SIMD that 144 registers allow (128 registers / 4 waves spills
// and is slower: 302 us), each wave stopping at 27 scalar-load waits per row.
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f8s __attribute__((ext_vector_type(8)));      // 8 / 16 scalar registers of weights
typedef float f16s __attribute__((ext_vector_type(16)));

struct PairVParams {
  const float* x;      // [N,3,H,W] planar
  const float* w1;     // plain fragments of layer 1: [9 taps][4 cin][16 cout]
  const float* a1;
  const float* b1;
  const float* w2;     // plain fragments of layer 2: [9 taps][4 cin pairs][16 cout][2]
  const float* a2;
  const float* b2;
  float* y;            // [N,H,W,8]
  int N, H, W, strips, segs, seg_rows;
};

__device__ __forceinline__ float dpp_from_left(float v) {    // lane i <- lane i - 1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_right(float v) {   // lane i <- lane i + 1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

template <int I, int N, typename F>
__device__ __forceinline__ void pv_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    pv_static_for<I + 1, N>(f);
  }
}

#ifndef MDF_PAIRV_BLOCKS
#define MDF_PAIRV_BLOCKS 1
#endif
// (the weight pointers are kernel arguments of their own, const + __restrict__: that is what lets the compiler read them with scalar loads)
__global__ __launch_bounds__(256, MDF_PAIRV_BLOCKS) void conv_pair_valu_kernel(const PairVParams p, const float* __restrict__ w1_, const float* __restrict__ w2_,
                                                             const float* __restrict__ a1, const float* __restrict__ b1, const float* __restrict__ a2,
                                                             const float* __restrict__ b2) {
  // epilogue constants: [a1 | b1 | a2 | b2] in LDS, read back by broadcast where they are used (as scalar registers they crowd out the
  // weight stream: 490 spilled SGPRs; as vector registers they cost an occupancy step)
  __shared__ __attribute__((aligned(16))) float epi[32];
  if (threadIdx.x < 32) epi[threadIdx.x] = (threadIdx.x < 8 ? a1 : threadIdx.x < 16 ? b1 : threadIdx.x < 24 ? a2 : b2)[threadIdx.x & 7];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int n_waves = p.N * p.strips * p.segs;
  if (wid >= n_waves) return;
  const int sx = wid % p.strips, seg = (wid / p.strips) % p.segs, n = wid / (p.strips * p.segs);
  const int w0 = sx * kOW;
  const int r0 = seg * p.seg_rows, r1 = min(p.H, r0 + p.seg_rows);
  if (r0 >= r1) return;
  const int xm = w0 - 1 + lane;                  // this lane's column of the intermediate map (and, lanes 1..62, of the output)
  const bool col_in = xm >= 0 && xm < p.W;
  const bool writes = lane >= 1 && lane <= kOW && xm < p.W;

  // input: one raw buffer over image n's three planes; a lane's three column offsets are constants, the row / channel part is scalar
  const float* img = p.x + (size_t)n * 3 * p.H * p.W;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, (int)((unsigned)(3 * p.H * p.W) * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t xr0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, 0, 0x00020000);      // zero records: a row outside the image reads 0
  int coff[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int gx = xm + dx - 1;
    coff[dx] = (gx >= 0 && gx < p.W) ? gx * 4 : (int)0x80000000u;       // out of range: reads 0
  }
  float in[3][3][3];      // [row slot][column dx][channel]
  auto fetch_row = [&](int gy, float (&dst)[3][3]) {       // straight into the slot of the row that just left the window
    const bool ok = gy >= 0 && gy < p.H;        // uniform
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int so = ok ? ((c * p.H + gy) * p.W) * 4 : 0;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) dst[dx][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ok ? xr : xr0, coff[dx], so, 0));
    }
  };
  f32x2v mid[3][3][4];    // [row slot][column dx][cin pair]: layer-1 outputs after BN + ReLU, zero outside the image

  const int m0 = r0 - 1;           // first intermediate row
  // iteration i: input row m0 + i + 1 arrives, intermediate row m0 + i is formed, output row r0 + i - 2 leaves (i >= 2)
  auto step = [&](auto pc, int i) {
    constexpr int P = decltype(pc)::value;                       // i % 3
    [[maybe_unused]] constexpr int S0 = P % 3, S1 = (P + 1) % 3, S2 = (P + 2) % 3;   // input rows m-1, m, m+1 / intermediate rows m-2, m-1, m live in slots S0.. (see below)
    const int m = m0 + i;
    // ---- layer 1, row m: rows m-1, m, m+1 are in slots S0, S1, S2
    f32x2v acc1[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    {
      // weights of tap t: [cin 0..2][cout 0..7] = three 8-dword scalar loads; the loads of tap t + 1 fly during the 12 FMAs of tap t
      f8s wa[3], wb[3];
      asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx8 %1, %3, 0x40\n\ts_load_dwordx8 %2, %3, 0x80"
                   : "=&s"(wa[0]), "=&s"(wa[1]), "=&s"(wa[2]) : "s"(w1_));
      pv_static_for<0, 9>([&](auto tc) {
        constexpr int tap = decltype(tc)::value, kh = tap / 3, kw = tap % 3;
        constexpr int SL = (P + kh) % 3;
        f8s (&cur)[3] = (tap & 1) ? wb : wa;
        f8s (&nxt)[3] = (tap & 1) ? wa : wb;
        if constexpr (tap < 8) {
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_load_dwordx8 %0, %10, %11\n\ts_load_dwordx8 %1, %10, %12\n\ts_load_dwordx8 %2, %10, %13"
                       : "=&s"(nxt[0]), "=&s"(nxt[1]), "=&s"(nxt[2]), "+s"(cur[0]), "+s"(cur[1]), "+s"(cur[2]), "+v"(acc1[0]), "+v"(acc1[1]), "+v"(acc1[2]), "+v"(acc1[3])
                       : "s"(w1_), "n"((tap + 1) * 256), "n"((tap + 1) * 256 + 64), "n"((tap + 1) * 256 + 128));      // (the accumulators: this statement stays behind the previous tap's FMAs)
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(cur[0]), "+s"(cur[1]), "+s"(cur[2]), "+v"(acc1[0]), "+v"(acc1[1]), "+v"(acc1[2]), "+v"(acc1[3]));
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const f32x2v xv = {in[SL][kw][c], in[SL][kw][c]};
#pragma unroll
          for (int j = 0; j < 4; ++j) acc1[j] = __builtin_elementwise_fma(xv, (f32x2v){cur[c][2 * j], cur[c][2 * j + 1]}, acc1[j]);
        }
      });
    }
    fetch_row(m + 2, in[S0]);               // row m - 1 has left the window: its slot takes row m + 2 (in flight during layer 2)
    const bool mid_ok = col_in && m >= 0 && m < p.H;
    f32x2v mv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float o0 = fmaxf(acc1[j][0] * epi[2 * j] + epi[8 + 2 * j], 0.f), o1 = fmaxf(acc1[j][1] * epi[2 * j + 1] + epi[8 + 2 * j + 1], 0.f);
      mv[j] = mid_ok ? (f32x2v){o0, o1} : (f32x2v){0.f, 0.f};
    }
    // intermediate rows m-2, m-1, m live in slots S1, S2, S0 of `mid` (slot of row m0 + i is i % 3 = S0)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mid[S0][1][j] = mv[j];
      mid[S0][0][j] = (f32x2v){dpp_from_left(mv[j][0]), dpp_from_left(mv[j][1])};
      mid[S0][2][j] = (f32x2v){dpp_from_right(mv[j][0]), dpp_from_right(mv[j][1])};
    }
    if (i < 2) return;
    // ---- layer 2, output row o = m - 1: intermediate rows o-1, o, o+1 = m-2, m-1, m in slots S1, S2, S0
    f32x2v acc2[8];
#pragma unroll
    for (int co = 0; co < 8; ++co) acc2[co] = (f32x2v){0.f, 0.f};
    {
      // weights of half a tap: cin pairs q = 2h, 2h + 1 x cout 0..7 x (even, odd cin) = two 16-dword scalar loads, in flight during the
      // previous half's 16 FMAs (scalar loads return out of order: the only wait is lgkmcnt(0), so a deeper queue means wider groups, and
      // 16-register tuples are aligned -- the file has six).  Every load statement carries the accumulators, which keeps it behind the
      // previous group's FMAs: left free, the scheduler runs the loads ahead and parks ~2000 scalar registers in vector lanes.
      f16s wa0, wa1, wb0, wb1;
      asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x80" : "=&s"(wa0), "=&s"(wa1) : "s"(w2_));
      pv_static_for<0, 18>([&](auto gc) {
        constexpr int g = decltype(gc)::value, tap = g / 2, h = g % 2, kh = tap / 3, kw = tap % 3;
        constexpr int SL = (P + 1 + kh) % 3;
        f16s& c0 = (g & 1) ? wb0 : wa0;
        f16s& c1 = (g & 1) ? wb1 : wa1;
        f16s& n0 = (g & 1) ? wa0 : wb0;
        f16s& n1 = (g & 1) ? wa1 : wb1;
        if constexpr (g < 17) {
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 %0, %12, %13\n\ts_load_dwordx16 %1, %12, %14"
                       : "=&s"(n0), "=&s"(n1), "+s"(c0), "+s"(c1), "+v"(acc2[0]), "+v"(acc2[1]), "+v"(acc2[2]), "+v"(acc2[3]), "+v"(acc2[4]), "+v"(acc2[5]), "+v"(acc2[6]), "+v"(acc2[7]) : "s"(w2_), "n"((g + 1) * 256), "n"((g + 1) * 256 + 128));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(c0), "+s"(c1), "+v"(acc2[0]), "+v"(acc2[1]), "+v"(acc2[2]), "+v"(acc2[3]), "+v"(acc2[4]), "+v"(acc2[5]), "+v"(acc2[6]), "+v"(acc2[7]));
        }
#pragma unroll
        for (int co = 0; co < 8; ++co) acc2[co] = __builtin_elementwise_fma(mid[SL][kw][2 * h], (f32x2v){c0[2 * co], c0[2 * co + 1]}, acc2[co]);
#pragma unroll
        for (int co = 0; co < 8; ++co) acc2[co] = __builtin_elementwise_fma(mid[SL][kw][2 * h + 1], (f32x2v){c1[2 * co], c1[2 * co + 1]}, acc2[co]);
      });
    }
    const int o = m - 1;
    if (writes && o < r1) {
      float v[8];
#pragma unroll
      for (int co = 0; co < 8; ++co) v[co] = fmaxf((acc2[co][0] + acc2[co][1]) * epi[16 + co] + epi[24 + co], 0.f);
      float* dst = p.y + (((size_t)n * p.H + o) * p.W + xm) * 8;
      *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  };

  // prologue: input rows m0 - 1, m0, m0 + 1 into slots 0, 1, 2
  fetch_row(m0 - 1, in[0]);
  fetch_row(m0, in[1]);
  fetch_row(m0 + 1, in[2]);
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int j = 0; j < 4; ++j) mid[s][dx][j] = (f32x2v){0.f, 0.f};
  const int i_last = r1 - r0 + 1;
  for (int i = 0;; i += 3) {
    step(std::integral_constant<int, 0>{}, i);
    if (i + 1 > i_last) break;
    step(std::integral_constant<int, 1>{}, i + 1);
    if (i + 2 > i_last) break;
    step(std::integral_constant<int, 2>{}, i + 2);
    if (i + 3 > i_last) break;
  }
}

}  // namespace

// weights: the packings mdf_conv_pack_weights produces for (Cin_mem 3, Cout 8, 9 taps) and (8, 8, 9 taps); the kernel reads
// their w-phase segments (behind the plain fragments, as conv_lds.hip's LDS_CASE_RW does)
extern "C" int mdf_conv2d_pair_fwd(const float* x, const float* w1pack, const float* alpha1, const float* beta1, const float* w2pack,
                                   const float* alpha2, const float* beta2, float* y, int N, int H, int W, void* stream) {
  MDF_REQUIRE(x && w1pack && alpha1 && beta1 && w2pack && alpha2 && beta2 && y, "null pointer argument");
  MDF_REQUIRE(N > 0 && H > 0 && W > 0, "bad shape");
  MDF_REQUIRE((long long)N * H * W * 8 < (1ll << 31), "output too large for 32-bit offsets");
  {
    const bool valu = [] { const char* e = getenv("MDF_CONV_PAIR_VALU"); return e ? atoi(e) != 0 : true; }();      // dev A/B and the equality test (read per call)
    if (valu) {
      PairVParams q{};
      q.x = x; q.w1 = w1pack; q.a1 = alpha1; q.b1 = beta1; q.w2 = w2pack; q.a2 = alpha2; q.b2 = beta2; q.y = y;
      q.N = N; q.H = H; q.W = W;
      q.strips = (W + kOW - 1) / kOW;
      // ~4096 waves (4 per SIMD): a segment costs two extra intermediate rows and a three-row prologue
      long long segs = 4096 / ((long long)N * q.strips);
      if (segs > H / 8) segs = H / 8;
      if (segs < 1) segs = 1;
      q.seg_rows = (int)((H + segs - 1) / segs);
      q.segs = (H + q.seg_rows - 1) / q.seg_rows;
      const long long waves = (long long)N * q.strips * q.segs;
      MDF_REQUIRE(waves < (1ll << 31) && (long long)3 * H * W * 4 < (1ll << 31), "image too large");
      hipLaunchKernelGGL(conv_pair_valu_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, q, w1pack, w2pack, alpha1, beta1, alpha2, beta2);
      return mdf::check_launch("conv_pair_valu_kernel");
    }
  }
  PairParams p{};
  p.x = x; p.a1 = alpha1; p.b1 = beta1; p.a2 = alpha2; p.b2 = beta2; p.y = y;
  p.w1 = w1pack + 9 * 4 * 16;       // plain fragments of (cin 4, cout 8): 9 taps x 64 lanes x 1
  p.w2 = w2pack + 9 * 8 * 16;       // (cin 8, cout 8): 9 taps x 64 lanes x 2
  p.N = N; p.H = H; p.W = W;
  p.strips = (W + kOW - 1) / kOW;
  // many short blocks: two are resident per CU (512 in all), so with ~1000 long ones the last few ran alone for a third of the
  // kernel's time (1040 blocks = 2.03 rounds); a segment costs 2 extra intermediate rows and a prologue, so not too many either
  static const long long target = [] { const char* e = getenv("MDF_PAIR_BLOCKS"); return (e && atoll(e) > 0) ? atoll(e) : 2048ll; }();   // dev A/B (sweep at cfg2: 1024 265 us, 2048 252, 4096 265, 8192 286)
  long long segs = target / ((long long)N * p.strips);
  if (segs > H / 16) segs = H / 16;
  if (segs < 1) segs = 1;
  p.seg_rows = (int)(((H + segs - 1) / segs + kRowsStep - 1) / kRowsStep * kRowsStep);
  p.segs = (H + p.seg_rows - 1) / p.seg_rows;
  const long long blocks = (long long)N * p.strips * p.segs;
  MDF_REQUIRE(blocks < (1ll << 31), "too many blocks");
  hipLaunchKernelGGL(conv_pair_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  return mdf::check_launch("conv_pair_kernel");
}
