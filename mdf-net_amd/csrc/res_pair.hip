// A residual block of the refinement net as ONE kernel (net/unit/base.py:39-47 `Res`, used by net/unit/refine.py:29,40):
//   y = x + s * conv_b(relu(conv_a(x)))      conv_a, conv_b = Conv2d(8, 8, k3, p1, no bias), s = 0.1, eval mode, NHWC
// Launched one after the other the two convs of a block take 18.6 us each at 592x800 for 0.55 GFLOP and 30 MB -- they are bound by
// their ramp (a launch per 4 us of MFMA work), and the 8-channel intermediate map is written and read back for nothing.
//
// Structure = conv_pair.hip (the backbone's two full-resolution layers): a block owns a strip of 62 output pixels x a segment of
// rows of one image and walks down the rows 8 at a time with two rolling windows in LDS -- the INPUT rows (NHWC, + halo; also the
// source of the residual) and the conv_a output rows (ReLU and conv_b's zero padding applied).  Both layers use the w-phase form of
// conv_lds.hip (Cfg::RW = 2: an MFMA column is 2 neighbouring output pixels, GEMM row = phase * 8 + cout, 4 taps along w) with the
// same packed weights, tap order and epilogue arithmetic as the single-layer launches (conv_lds_kernel<8,8,8,1,3,1,2,2>), so the
// result is BIT-IDENTICAL to them (tests/test_conv2d_gpu.py).
#include <cstdlib>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kOW = 62;          // output pixels per strip
constexpr int kIW = 66;          // input pixels staged per row: [w0 - 2, w0 + 64)
constexpr int kRing = 16;        // rows per rolling window (>= 10 live rows; power of two)
constexpr int kRowsStep = 8;     // rows per step (2 per wave)
constexpr int kS = 66;           // float2 per (cin pair, slot) row of either window: 64 computed + 2 columns only the discarded MFMA column reads
constexpr int kWinFloats = 4 * kRing * kS * 2;

struct ResPairParams {
  const float* x;     // [N,H,W,8]
  const float* wa;    // w-phase packing of (cin 8, cout 8): [12 taps][64 lanes][2]
  const float* wb;
  float* y;           // [N,H,W,8]
  float scale;
  int N, H, W, strips, segs, seg_rows;
};

__global__ __launch_bounds__(256) void res_pair_kernel(const ResPairParams p) {
  __shared__ __attribute__((aligned(16))) float in_img[kWinFloats];     // [cin pair][ring row][px][2]
  __shared__ __attribute__((aligned(16))) float mid_img[kWinFloats];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, n16 = lane & 15;
  int bid = blockIdx.x;
  const int sx = bid % p.strips; bid /= p.strips;
  const int seg = bid % p.segs;
  const int n = bid / p.segs;
  const int w0 = sx * kOW;
  const int r0 = seg * p.seg_rows, r1 = min(p.H, r0 + p.seg_rows);
  if (r0 >= r1) return;

  // weights of both layers stay in registers
  float war[12][2], wbr[12][2];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const float2 a = *reinterpret_cast<const float2*>(p.wa + (i * 64 + lane) * 2);
    const float2 b = *reinterpret_cast<const float2*>(p.wb + (i * 64 + lane) * 2);
    war[i][0] = a.x; war[i][1] = a.y; wbr[i][0] = b.x; wbr[i][1] = b.y;
  }
  const int c0 = 4 * (q & 1), phase = q >> 1;

  // input rows: element e of a group of `nrows` rows = (rr, px, half): one 16-B load = channels 4*half .. +3 of pixel (first_row + rr, w0 - 2 + px)
  const float* xin = p.x + (size_t)n * p.H * p.W * 8;
  auto fetch_elem = [&](int e, int first_row, int nrows) -> float4 {
    const int half = e & 1, r = e >> 1;
    const int rr = r / kIW, px = r - rr * kIW;
    const int gy = first_row + rr, gx = w0 - 2 + px;
    if (e >= nrows * kIW * 2 || gy < 0 || gy >= p.H || gx < 0 || gx >= p.W) return make_float4(0.f, 0.f, 0.f, 0.f);
    return *reinterpret_cast<const float4*>(xin + ((size_t)gy * p.W + gx) * 8 + 4 * half);
  };
  auto commit_elem = [&](int e, int first_row, int nrows, const float4& v) {
    if (e >= nrows * kIW * 2) return;
    const int half = e & 1, r = e >> 1;
    const int rr = r / kIW, px = r - rr * kIW;
    float* dst = in_img + (((2 * half) * kRing + ((first_row + rr) & (kRing - 1))) * kS + px) * 2;     // cin pair 2*half, then 2*half + 1
    *reinterpret_cast<float2*>(dst) = make_float2(v.x, v.y);
    *reinterpret_cast<float2*>(dst + kRing * kS * 2) = make_float2(v.z, v.w);
  };
  // prologue: rows r0-2 .. r0+1 straight into the window; rows r0+2 .. r0+9 take the pipelined route below
  for (int e = tid; e < 4 * kIW * 2; e += 256) commit_elem(e, r0 - 2, 4, fetch_elem(e, r0 - 2, 4));
  constexpr int NPF = (kRowsStep * kIW * 2 + 255) / 256;      // 5 float4 per thread per step
  float4 pf[NPF];
  auto fetch = [&](int first_row) {
#pragma unroll
    for (int k = 0; k < NPF; ++k) pf[k] = fetch_elem(tid + 256 * k, first_row, kRowsStep);
  };
  auto commit = [&](int first_row) {
#pragma unroll
    for (int k = 0; k < NPF; ++k) commit_elem(tid + 256 * k, first_row, kRowsStep, pf[k]);
  };
  fetch(r0 + 2);

  // one output row of a layer from a window: both MFMA tiles of the 64-pixel row, taps (kh, kw) in conv_lds's order
  auto conv_row = [&](const float* win, const float (&wr)[12][2], int m, f32x4 (&acc)[2]) {
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const float* row = win + ((q * kRing + ((m + kh - 1) & (kRing - 1))) * kS + n16 * 2) * 2;
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        float2 bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bv[t] = *reinterpret_cast<const float2*>(row + (t * 32 + kw) * 2);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[kh * 4 + kw][0], bv[t].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[kh * 4 + kw][1], bv[t].y, acc[t], 0, 0, 0);
      }
    }
  };
  // conv_a: intermediate row m (pixels w0 - 1 .. w0 + 62) -> mid window, ReLU, zero outside the image (conv_b's padding)
  auto layer_a_row = [&](int m) {
    f32x4 acc[2];
    conv_row(in_img, war, m, acc);
    const bool row_in = (m >= 0 && m < p.H);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ix = (t * 16 + n16) * 2 + phase;           // intermediate pixel w0 - 1 + ix
      const int gx = w0 - 1 + ix;
      const bool ok = row_in && gx >= 0 && gx < p.W;
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float o = fmaxf(acc[t][k] * 1.0f + 0.0f, 0.f);     // (the single-layer kernel's epilogue with alpha = 1, beta = 0, ReLU)
        v[k] = ok ? o : 0.f;
      }
      float* dst = mid_img + (((c0 >> 1) * kRing + (m & (kRing - 1))) * kS + ix) * 2;
      *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
      *reinterpret_cast<float2*>(dst + kRing * kS * 2) = make_float2(v[2], v[3]);
    }
  };
  // the residual of output row o (this lane's 4 channels of its 2 x 2 pixels), taken from the input window BEFORE the barrier that
  // ends the conv_a phase: the next step's commit re-uses the ring slots of rows R, R+1 while slower waves are still in conv_b
  auto residual_row = [&](int o, float4 (&res)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ox = (t * 16 + n16) * 2 + phase;
      const float* src = in_img + (((c0 >> 1) * kRing + (o & (kRing - 1))) * kS + ox + 2) * 2;        // input pixel w0 + ox = window index ox + 2
      const float2 ra = *reinterpret_cast<const float2*>(src), rb = *reinterpret_cast<const float2*>(src + kRing * kS * 2);
      res[t] = make_float4(ra.x, ra.y, rb.x, rb.y);
    }
  };
  // conv_b: output row o, y = x + scale * conv
  auto layer_b_row = [&](int o, const float4 (&res)[2]) {
    f32x4 acc[2];
    conv_row(mid_img, wbr, o, acc);
    if (o >= r1) return;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ox = (t * 16 + n16) * 2 + phase;
      const int gx = w0 + ox;
      if (ox >= kOW || gx >= p.W) continue;
      float4 v;
      v.x = res[t].x + (acc[t][0] * 1.0f + 0.0f) * p.scale;         // (the single-layer kernel's epilogue: alpha = 1, beta = 0, then res + o * s)
      v.y = res[t].y + (acc[t][1] * 1.0f + 0.0f) * p.scale;
      v.z = res[t].z + (acc[t][2] * 1.0f + 0.0f) * p.scale;
      v.w = res[t].w + (acc[t][3] * 1.0f + 0.0f) * p.scale;
      *reinterpret_cast<float4*>(p.y + (((size_t)n * p.H + o) * p.W + gx) * 8 + c0) = v;
    }
  };

  for (int R = r0; R < r1; R += kRowsStep) {
    commit(R + 2);                // input rows R+2 .. R+9
    __syncthreads();
    if (R + kRowsStep < r1) fetch(R + 2 + kRowsStep);         // next step's rows in flight during the MFMAs below
    if (R == r0 && wave < 2) layer_a_row(r0 - 1 + wave);      // the two rows above the first step's own eight
    layer_a_row(R + 1 + 2 * wave);
    layer_a_row(R + 2 + 2 * wave);
    float4 res0[2], res1[2];
    residual_row(R + 2 * wave, res0);
    residual_row(R + 2 * wave + 1, res1);
    __syncthreads();
    layer_b_row(R + 2 * wave, res0);
    layer_b_row(R + 2 * wave + 1, res1);
  }
}

// ---- head of the refinement net: (depth - lo) / span -> Conv2d(1, 8, k3, p1, no bias) as one launch (net/unit/refine.py:29,36) -------
// On the matrix cores the 1 -> 8 layer uses one live k in four and 8 rows in 16 (17.5 us at 592x800 for 0.07 GFLOP) behind a range
// mapping launch of its own (6.4 us).  Here a thread owns an output pixel: it normalises its 3x3 neighbourhood with torch's roundings
// (sub, then a true divide; 0 outside the image = the conv's zero padding) and runs the 9-tap fma chain of each of the 8 output channels in
// the MFMA's order (kh, kw ascending, acc = fma(w, x, acc) from 0), weights as scalar operands: bit-identical to the two launches.
typedef const __attribute__((address_space(4))) float* cweights_p;

__global__ __launch_bounds__(256) void refine_head_kernel(const float* __restrict__ depth, const float* __restrict__ lo, const float* __restrict__ span,
                                                          const float* w, float* __restrict__ y, int B, int H, int W) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * H * W) return;
  const int x = (int)(i % W), yy = (int)((i / W) % H), b = (int)(i / ((long long)W * H));
  const float l = lo ? lo[b] : 0.f, sp = lo ? span[b] : 1.f;
  const float* d = depth + (size_t)b * H * W;
  float v[9];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int gy = yy + kh - 1, gx = x + kw - 1;
      const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
      const float t = d[in ? (size_t)gy * W + gx : (size_t)yy * W + x];
      const float nv = lo ? __fdiv_rn(__fsub_rn(t, l), sp) : t;
      v[kh * 3 + kw] = in ? nv : 0.f;
    }
  const cweights_p cw = (cweights_p)w;            // [8][1][3][3]: uniform addresses -> scalar loads
  float o[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc = __fmaf_rn(cw[c * 9 + t], v[t], acc);
    o[c] = acc * 1.0f + 0.0f;                      // (the conv kernel's epilogue with alpha = 1, beta = 0)
  }
  float4* dst = reinterpret_cast<float4*>(y + (size_t)i * 8);
  dst[0] = make_float4(o[0], o[1], o[2], o[3]);
  dst[1] = make_float4(o[4], o[5], o[6], o[7]);
}

}  // namespace

extern "C" int mdf_refine_head_fwd(const float* depth, const float* lo, const float* span, const float* weight, float* y, int B, int H, int W,
                                   void* stream) {
  MDF_REQUIRE(depth && weight && y, "null pointer argument");
  MDF_REQUIRE((lo == nullptr) == (span == nullptr), "lo and span must both be given or both be NULL");
  MDF_REQUIRE(B > 0 && H > 0 && W > 0, "bad shape");
  const long long n = (long long)B * H * W;
  MDF_REQUIRE(n * 8 < (1ll << 31), "map too large for 32-bit offsets");
  hipLaunchKernelGGL(refine_head_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, depth, lo, span, weight, y, B, H, W);
  return mdf::check_launch("refine_head_kernel");
}

// weights: the packings mdf_conv_pack_weights produces for (Cin_mem 8, Cout 8, 9 taps); the kernel reads their w-phase segments
// (behind the plain fragments, as conv_lds.hip's LDS_CASE_RW and conv_pair.hip do)
extern "C" int mdf_conv2d_res_pair_fwd(const float* x, const float* wa_pack, const float* wb_pack, float scale, float* y, int N, int H,
                                       int W, void* stream) {
  MDF_REQUIRE(x && wa_pack && wb_pack && y, "null pointer argument");
  MDF_REQUIRE(x != y, "in-place operation is not supported (a strip reads its neighbours' input pixels)");
  MDF_REQUIRE(N > 0 && H > 0 && W > 0, "bad shape");
  MDF_REQUIRE((long long)N * H * W * 8 < (1ll << 31), "map too large for 32-bit offsets");
  ResPairParams p{};
  p.x = x; p.y = y; p.scale = scale;
  p.wa = wa_pack + 9 * 8 * 16;       // behind the plain fragments of (cin 8, cout 8): 9 taps x 64 lanes x 2
  p.wb = wb_pack + 9 * 8 * 16;
  p.N = N; p.H = H; p.W = W;
  p.strips = (W + kOW - 1) / kOW;
  // enough blocks to fill the chip a few times over (two are resident per CU: 68 KB of LDS); a segment costs 2 extra intermediate rows
  static const long long target = [] { const char* e = getenv("MDF_RES_PAIR_BLOCKS"); return (e && atoll(e) > 0) ? atoll(e) : 1024ll; }();   // dev A/B
  long long segs = target / ((long long)N * p.strips);
  if (segs > H / 16) segs = H / 16;
  if (segs < 1) segs = 1;
  p.seg_rows = (int)(((H + segs - 1) / segs + kRowsStep - 1) / kRowsStep * kRowsStep);
  p.segs = (H + p.seg_rows - 1) / p.seg_rows;
  const long long blocks = (long long)N * p.strips * p.segs;
  MDF_REQUIRE(blocks < (1ll << 31), "too many blocks");
  hipLaunchKernelGGL(res_pair_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  return mdf::check_launch("res_pair_kernel");
}
