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

}  // namespace

// weights: the packings mdf_conv_pack_weights produces for (Cin_mem 3, Cout 8, 9 taps) and (8, 8, 9 taps); the kernel reads
// their w-phase segments (behind the plain fragments, as conv_lds.hip's LDS_CASE_RW does)
extern "C" int mdf_conv2d_pair_fwd(const float* x, const float* w1pack, const float* alpha1, const float* beta1, const float* w2pack,
                                   const float* alpha2, const float* beta2, float* y, int N, int H, int W, void* stream) {
  MDF_REQUIRE(x && w1pack && alpha1 && beta1 && w2pack && alpha2 && beta2 && y, "null pointer argument");
  MDF_REQUIRE(N > 0 && H > 0 && W > 0, "bad shape");
  MDF_REQUIRE((long long)N * H * W * 8 < (1ll << 31), "output too large for 32-bit offsets");
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
