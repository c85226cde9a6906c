// The tail of the refinement net as ONE launch (mdf_refine_tail_fwd); see the block comment below.  Built on conv_lds_common.h.
#include "conv_lds_common.h"

// ---- tail of the refinement net as one launch (refine.py:18-20,42-44) ---------------------------------------------------------
//   Conv2d(8, 32, k3) -> PixelShuffle(2) -> Conv2d(8, 1, k3) [-> lo + y * span]
// As two launches the 8-channel map at twice the resolution (60 MB at 1184 x 1600) is written and read back, and the 8 -> 1
// layer runs on the matrix cores with ONE live GEMM row in sixteen.  Here a block owns 8 x 30 low-resolution pixels: it stages
// their 12 x 34 input neighbourhood in LDS, computes the 8 -> 32 layer on 10 x 32 pixels with the MFMA step of the single-layer
// kernel (same packed weights, PixelShuffle as the row order), writes the 18 x 64 x 8 shuffled block (zero outside the image:
// the second layer's padding) to LDS, and finishes with the 8 -> 1 layer on the vector ALUs (72 FMAs per output, weights as
// scalar operands, a column per lane so every LDS read is conflict-free and every store is a contiguous row).
namespace {

struct TailParams {
  const float* x;      // [B,H,W,8]
  const float* w1;     // pack_conv2d_weight(shuffle2_rows(conv[0].weight)): plain fragments [9][nt 2][64][2]
  const float* w2;     // conv[2].weight [1,8,3,3]
  const float* lo;     // [B] or null: y = lo + y * span (torch's roundings)
  const float* span;
  float* y;            // [B,2H,2W]
  int B, H, W, tiles_h, tiles_w;
};

struct TailCfg : Cfg<8, 8, 32, 1, 3, 1, 2> {
  typedef Cfg<8, 8, 32, 1, 3, 1, 2> Base;
  static constexpr int LR_H = 8, LR_W = 30;          // low-resolution pixels a block owns (10 or 12 rows: 58 us against 54 at 592 x 800, two blocks per CU instead of three)
  static constexpr int CR = LR_H + 2;                // rows of the first layer it computes (one halo row each side); 32 columns
  static constexpr int PH = CR + 2;                  // input rows staged; Base::PW = 34 columns
  static constexpr int S = round_s(PH * Base::PW, Base::KPL, Base::SW);
  static constexpr int PLANE = 8 * S;
  static constexpr int NFILL = (Base::NG * PH * Base::PW + 255) / 256;
  static constexpr int MID_ROWS = 2 * LR_H + 2, MID_COLS = 64;   // shuffled block: rows 2*r0-1 .., columns 2*c0-2 ..; [half][row][col] float4
  static constexpr int MID_FLOATS = 2 * MID_ROWS * MID_COLS * 4;
  static_assert(Base::WREG, "the first layer's weights are expected to fit the registers");
};

__global__ __launch_bounds__(256, 3) void refine_tail_kernel(const TailParams p) {
  typedef TailCfg C;
  constexpr int KPL = C::KPL, NG = C::NG, S = C::S, PW = C::PW, PH = C::PH;
  typedef typename VecT<KPL>::type vec_t;
  __shared__ __attribute__((aligned(16))) float in_img[C::PLANE];
  __shared__ __attribute__((aligned(16))) float mid[C::MID_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, n16 = lane & 15;
  int bid = blockIdx.x;
  const int twi = bid % p.tiles_w; bid /= p.tiles_w;
  const int th = bid % p.tiles_h;
  const int b = bid / p.tiles_h;
  const int r0 = th * C::LR_H, c0 = twi * C::LR_W;

  // input neighbourhood rows r0-2 .., columns c0-2 ..  (cin group fastest: coalesced 32-B pixels)
#pragma unroll
  for (int k = 0; k < C::NFILL; ++k) {
    const int idx = tid + k * 256;
    if (idx < NG * PH * PW) {
      const int g = idx % NG, v = idx / NG;
      const int row = v / PW, col = v - row * PW;
      const int ih = r0 - 2 + row, iw = c0 - 2 + col;
      vec_t val = vec_zero<KPL>();
      if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) val = *reinterpret_cast<const vec_t*>(p.x + (((size_t)b * p.H + ih) * p.W + iw) * 8 + g * KPL);
      *reinterpret_cast<vec_t*>(in_img + (g * S + v) * KPL) = val;
    }
  }
  const __amdgpu_buffer_rsrc_t wres = make_rsrc(p.w1, (unsigned)(C::NSTEP * C::NT * 64 * KPL * 4));
  const int wvoff = lane * KPL * 4;
  float wr[C::WN][C::NT][KPL], wfirst[2][C::NT][KPL];
#pragma unroll
  for (int i = 0; i < C::NSTEP; ++i)
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) buf_load_to<KPL>(wres, wvoff, (i * C::NT + nt) * (64 * KPL * 4), wr[i][nt]);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
      for (int k = 0; k < KPL; ++k) wfirst[i][nt][k] = 0.f;
  __syncthreads();

  // layer 1 on rows r0-1 .. r0+8, columns c0-1 .. c0+30; PixelShuffle(2) into `mid`
  for (int rr = wave; rr < C::CR; rr += 4) {
    const float* planes[1] = {in_img + (q * S + rr * PW + n16) * KPL};
    f32x4 acc[2][C::NT];
    step_mfma<C, 1, 3, 2>(planes, wres, wvoff, wr, wfirst, acc);
    const int lr = r0 - 1 + rr;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt) {
        const int row0 = nt * 16 + 4 * q;                 // GEMM row = sub*8 + oc, sub = dy*2 + dx
        const int sub = row0 >> 3, oc0 = row0 & 7;
        const int lc = c0 - 1 + t * 16 + n16;
        const int hy = 2 * lr + (sub >> 1), hx = 2 * lc + (sub & 1);
        const int my = hy - (2 * r0 - 1), mxx = hx - (2 * c0 - 2);
        if (my < 0 || my >= C::MID_ROWS) continue;        // (the outermost computed rows have one sub-row nobody reads)
        const bool in = hy >= 0 && hy < 2 * p.H && hx >= 0 && hx < 2 * p.W;
        const float4 v = in ? make_float4(acc[t][nt][0], acc[t][nt][1], acc[t][nt][2], acc[t][nt][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(mid + (((oc0 >> 2) * C::MID_ROWS + my) * C::MID_COLS + mxx) * 4) = v;
      }
  }
  __syncthreads();

  // layer 2: lane = output column 2*c0 + tx, rows 2*r0 + RPT*ty .. +RPT-1; out row o reads mid rows o..o+2, out column tx reads mid columns tx+1..tx+3
  constexpr int RPT = (2 * C::LR_H + 3) / 4;
  const int tx = tid & 63, ty = tid >> 6;
  if (tx >= 2 * C::LR_W) return;
  const int hx = 2 * c0 + tx;
  if (hx >= 2 * p.W) return;
  const float l = p.lo ? p.lo[b] : 0.f, sp = p.lo ? p.span[b] : 1.f;
#pragma unroll 1      // (unrolled, hipcc hoists all 36 LDS reads of the four rows: 224 registers, two blocks per CU instead of three)
  for (int j = 0; j < RPT; ++j) {
    const int orow = RPT * ty + j, hy = 2 * r0 + orow;
    if (orow >= 2 * C::LR_H || hy >= 2 * p.H) break;
    float o = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const float* row = mid + ((hf * C::MID_ROWS + orow + kh) * C::MID_COLS + tx + 1) * 4;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float4 v = *reinterpret_cast<const float4*>(row + kw * 4);
          const float* w = p.w2 + (hf * 4) * 9 + kh * 3 + kw;     // weight [c][kh][kw], c = 4*hf + i
          o = fmaf(v.x, w[0], o);
          o = fmaf(v.y, w[9], o);
          o = fmaf(v.z, w[18], o);
          o = fmaf(v.w, w[27], o);
        }
      }
    p.y[((size_t)b * 2 * p.H + hy) * (2 * p.W) + hx] = p.lo ? __fadd_rn(l, __fmul_rn(o, sp)) : o;
  }
}

}  // namespace

extern "C" int mdf_refine_tail_fwd(const float* x, const float* w1pack, const float* w2, const float* lo, const float* span, float* y,
                                   int B, int H, int W, void* stream) {
  MDF_REQUIRE(x && w1pack && w2 && y, "null pointer argument");
  MDF_REQUIRE((lo == nullptr) == (span == nullptr), "lo and span must both be given or both be NULL");
  MDF_REQUIRE(B > 0 && H > 0 && W > 0, "bad shape");
  MDF_REQUIRE((long long)B * H * W * 8 < (1ll << 31), "input too large for 32-bit offsets");
  TailParams p{};
  p.x = x; p.w1 = w1pack; p.w2 = w2; p.lo = lo; p.span = span; p.y = y; p.B = B; p.H = H; p.W = W;
  p.tiles_h = (H + TailCfg::LR_H - 1) / TailCfg::LR_H;
  p.tiles_w = (W + TailCfg::LR_W - 1) / TailCfg::LR_W;
  const long long blocks = (long long)B * p.tiles_h * p.tiles_w;
  MDF_REQUIRE(blocks < (1ll << 31), "too many blocks");
  hipLaunchKernelGGL(refine_tail_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  return mdf::check_launch("refine_tail_kernel");
}
