// 1x1 convolutions (the feature pyramid's lateral / output heads, reference net/unit/backbone.py:40-47,58-66): pure streaming.
//
// A 1x1 layer reads every input pixel once and has no halo, so staging tiles through LDS (conv_lds.hip, where these layers ran
// at ~3.3 TB/s of algorithmic traffic, occupancy-bound) buys nothing: with NHWC activations the B fragment of an MFMA column IS
// a coalesced load -- lane (q, n) of a wave reads channels 4q..4q+3 (+16 per chunk) of pixel n, 64 lanes = 16 pixels x 64 B
// contiguous -- and the D fragment is a coalesced float4 store.  A wave keeps the whole (packed) weight matrix in registers,
// owns runs of U consecutive 16-pixel tiles and issues all U x Cin/16 loads of a run before its first MFMA; 8-16 waves per CU
// keep >= 32 KB in flight per CU, which is what HBM latency x bandwidth asks for.  Arithmetic and epilogue are those of the
// LDS kernels (same MFMA, same k order inside a chunk, same scale/shift/ReLU/upsample-add/residual sequence): results are
// bit-identical to that path.
#include <cstdlib>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct P1 {
  const float* x;       // [npx][CIN]
  const float* wpack;   // plain packing, one tap: [chunk][nt][lane][4]
  const float* alpha;   // [COUT] or null
  const float* beta;    // [COUT] or null
  const float* res;     // [npx][COUT] or null
  const float* res_up;  // [B][Ho/2][Wo/2][COUT] or null
  float* y;             // [npx][COUT]
  float res_scale;
  int relu;
  int Ho, Wo;
  long long npx;
  int n_runs;           // runs of U tiles
};

// (image, row, column) of a run's pixels for the upsample-add: two 32-bit divisions per RUN (its first pixel) and a carry per tile, instead
// of two 64-bit divisions per tile -- ~250 vector instructions beside the 4-12 MFMAs of a tile, which kept these HBM-bound layers at
// 3.0-3.6 TB/s (r05; the launchers bound the pixel count to 31 bits)
struct PixBase { int img, oh, ow; };
__device__ __forceinline__ PixBase pix_base(unsigned first_px, int Ho, int Wo) {
  const unsigned rr = first_px / (unsigned)Wo, img = rr / (unsigned)Ho;
  return PixBase{(int)img, (int)(rr - img * (unsigned)Ho), (int)(first_px - rr * (unsigned)Wo)};
}
__device__ __forceinline__ void pix_at(const PixBase& b, int delta, int Ho, int Wo, int& img, int& oh, int& ow) {
  img = b.img; oh = b.oh; ow = b.ow + delta;
  while (ow >= Wo) {
    ow -= Wo;
    if (++oh == Ho) { oh = 0; ++img; }
  }
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv1x1_kernel(const P1 p) {
  constexpr int NCH = CIN / 16, NT = COUT / 16;
  constexpr int U = (8 / NCH) < 1 ? 1 : 8 / NCH;   // tiles per run: ~8 float4 loads in flight per lane
  const int lane = threadIdx.x & 63, q = lane >> 4, n16 = lane & 15;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;

  float wr[NCH][NT][4];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float4 v = *reinterpret_cast<const float4*>(p.wpack + ((size_t)(ch * NT + nt) * 64 + lane) * 4);
      wr[ch][nt][0] = v.x; wr[ch][nt][1] = v.y; wr[ch][nt][2] = v.z; wr[ch][nt][3] = v.w;
    }
  float al[NT][4], be[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = nt * 16 + 4 * q + k;
      al[nt][k] = p.alpha ? p.alpha[c] : 1.f;
      be[nt][k] = p.beta ? p.beta[c] : 0.f;
    }

  for (int run = wave_g; run < p.n_runs; run += n_waves) {
    const long long px0 = (long long)run * (U * 16) + n16;
    float4 xb[U][NCH];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long px = px0 + u * 16;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch)
        xb[u][ch] = (px < p.npx) ? *reinterpret_cast<const float4*>(p.x + px * CIN + ch * 16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    PixBase pb{0, 0, 0};
    if (p.res_up) pb = pix_base((unsigned)run * (U * 16), p.Ho, p.Wo);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long px = px0 + u * 16;
      f32x4 acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const float b4[4] = {xb[u][ch].x, xb[u][ch].y, xb[u][ch].z, xb[u][ch].w};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[ch][nt][s], b4[s], acc[nt], 0, 0, 0);
      }
      if (px >= p.npx) continue;
      // (upsample-add: pixel -> (image, row, column) of the output grid)
      int oh = 0, ow = 0, img = 0;
      if (p.res_up) pix_at(pb, u * 16 + n16, p.Ho, p.Wo, img, oh, ow);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c0 = nt * 16 + 4 * q;
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          o[k] = acc[nt][k] * al[nt][k] + be[nt][k];
          if (p.relu) o[k] = fmaxf(o[k], 0.f);
        }
        if (p.res_up) {  // + F.interpolate(top, scale_factor=2, bilinear, align_corners=False)[row, ow]   (backbone.py:60,62)
          const int Hh = p.Ho >> 1, Wh = p.Wo >> 1;
          float sy = ((float)oh + 0.5f) * 0.5f - 0.5f; sy = sy < 0.f ? 0.f : sy;
          float sx = ((float)ow + 0.5f) * 0.5f - 0.5f; sx = sx < 0.f ? 0.f : sx;
          const int y0 = (int)sy, x0 = (int)sx;
          const int y1 = y0 + (y0 < Hh - 1), x1 = x0 + (x0 < Wh - 1);
          const float ly1 = sy - (float)y0, ly0 = 1.f - ly1, lx1 = sx - (float)x0, lx0 = 1.f - lx1;
          const float* base = p.res_up + (size_t)img * Hh * Wh * COUT + c0;
          const float4 v00 = *reinterpret_cast<const float4*>(base + ((size_t)y0 * Wh + x0) * COUT);
          const float4 v01 = *reinterpret_cast<const float4*>(base + ((size_t)y0 * Wh + x1) * COUT);
          const float4 v10 = *reinterpret_cast<const float4*>(base + ((size_t)y1 * Wh + x0) * COUT);
          const float4 v11 = *reinterpret_cast<const float4*>(base + ((size_t)y1 * Wh + x1) * COUT);
          const float w00 = ly0 * lx0, w01 = ly0 * lx1, w10 = ly1 * lx0, w11 = ly1 * lx1;
          // torch order: interpolate(...) + lat(x)  ->  up + o
          o[0] = __fmaf_rn(w11, v11.x, __fmaf_rn(w10, v10.x, __fmaf_rn(w00, v00.x, w01 * v01.x))) + o[0];
          o[1] = __fmaf_rn(w11, v11.y, __fmaf_rn(w10, v10.y, __fmaf_rn(w00, v00.y, w01 * v01.y))) + o[1];
          o[2] = __fmaf_rn(w11, v11.z, __fmaf_rn(w10, v10.z, __fmaf_rn(w00, v00.z, w01 * v01.z))) + o[2];
          o[3] = __fmaf_rn(w11, v11.w, __fmaf_rn(w10, v10.w, __fmaf_rn(w00, v00.w, w01 * v01.w))) + o[3];
        }
        const size_t oi = (size_t)px * COUT + c0;
        if (p.res) {
          const float4 rv = *reinterpret_cast<const float4*>(p.res + oi);
          o[0] = rv.x + o[0] * p.res_scale; o[1] = rv.y + o[1] * p.res_scale;
          o[2] = rv.z + o[2] * p.res_scale; o[3] = rv.w + o[3] * p.res_scale;
        }
        *reinterpret_cast<float4*>(p.y + oi) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
}

// ---- several 1x1 heads on ONE input (the feature pyramid's composed heads, net/unit/backbone.py:_composed_heads) ------------------
// out4 / the level-3 and level-2 contributions of t4 are three 1x1 convs over the same 64-channel map, and the two heads of t3 two over
// the same 32-channel map: as separate launches the input is streamed once per head.  Here a wave loads its run of input tiles once
// and multiplies it with the concatenated weight matrix (COUT0 + COUT1 + COUT2 rows); every head keeps its own output tensor, bias and
// upsample-add source, and its n-tiles go through the single-head kernel's epilogue, so each output is bit-identical to the
// single-head launch.
struct P1M {
  const float* x;          // [npx][CIN]
  const float* wpack[3];   // per head: plain packing, one tap: [chunk][nt][lane][4]
  const float* beta[3];    // [COUT_h] or null
  const float* res_up[3];  // [B][Ho/2][Wo/2][COUT_h] or null
  float* y[3];             // [npx][COUT_h]
  int Ho, Wo;
  long long npx;
  int n_runs;
};

template <int CIN, int C0, int C1, int C2, int U>      // U: tiles per run (the weights take NCH * NT * 4 registers)
__global__ __launch_bounds__(256) void conv1x1_heads_kernel(const P1M p) {
  constexpr int NCH = CIN / 16;
  constexpr int N0 = C0 / 16, N1 = C1 / 16, N2 = C2 / 16, NT = N0 + N1 + N2;
  const int lane = threadIdx.x & 63, q = lane >> 4, n16 = lane & 15;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
  auto head_of = [](int nt) { return nt < N0 ? 0 : (nt < N0 + N1 ? 1 : 2); };
  auto first_of = [](int h) { return h == 0 ? 0 : (h == 1 ? N0 : N0 + N1); };
  auto cout_of = [](int h) { return h == 0 ? C0 : (h == 1 ? C1 : C2); };

  // weights: in registers (NCH * NT * 4 of them) -- or, WLDS (the 64-channel input: 112 registers, one wave per SIMD), in LDS as the
  // same [chunk][nt][lane][4] fragments, read back with one ds_read_b128 per (chunk, n-tile) and tile
  constexpr bool WLDS = (NCH * NT * 4 > 64);
  __shared__ __attribute__((aligned(16))) float wsm[WLDS ? NCH * NT * 64 * 4 : 4];
  float wr[WLDS ? 1 : NCH][WLDS ? 1 : NT][4];
  float be[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int h = head_of(nt), lt = nt - first_of(h), nth = cout_of(h) / 16;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      if constexpr (WLDS) {
        if (threadIdx.x < 64)
          *reinterpret_cast<float4*>(wsm + ((ch * NT + nt) * 64 + lane) * 4) = *reinterpret_cast<const float4*>(p.wpack[h] + ((size_t)(ch * nth + lt) * 64 + lane) * 4);
      } else {
        const float4 v = *reinterpret_cast<const float4*>(p.wpack[h] + ((size_t)(ch * nth + lt) * 64 + lane) * 4);
        wr[ch][nt][0] = v.x; wr[ch][nt][1] = v.y; wr[ch][nt][2] = v.z; wr[ch][nt][3] = v.w;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) be[nt][k] = p.beta[h] ? p.beta[h][lt * 16 + 4 * q + k] : 0.f;
  }
  if constexpr (WLDS) __syncthreads();

  for (int run = wave_g; run < p.n_runs; run += n_waves) {
    const long long px0 = (long long)run * (U * 16) + n16;
    float4 xb[U][NCH];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long px = px0 + u * 16;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch)
        xb[u][ch] = (px < p.npx) ? *reinterpret_cast<const float4*>(p.x + px * CIN + ch * 16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const PixBase pb = pix_base((unsigned)run * (U * 16), p.Ho, p.Wo);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long px = px0 + u * 16;
      f32x4 acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const float b4[4] = {xb[u][ch].x, xb[u][ch].y, xb[u][ch].z, xb[u][ch].w};
        if constexpr (WLDS) {
          asm volatile("" ::: "memory");      // (the fragments are re-read per tile: hoisted out of the loop they are 112 registers again)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float4 wv = *reinterpret_cast<const float4*>(wsm + ((ch * NT + nt) * 64 + lane) * 4);
            const float w4[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[s], b4[s], acc[nt], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[ch][nt][s], b4[s], acc[nt], 0, 0, 0);
        }
      }
      if (px >= p.npx) continue;
      int img, oh, ow;
      pix_at(pb, u * 16 + n16, p.Ho, p.Wo, img, oh, ow);
      // bilinear x2 taps of this pixel (F.interpolate(scale_factor=2, bilinear, align_corners=False), backbone.py:60,62)
      const int Hh = p.Ho >> 1, Wh = p.Wo >> 1;
      float sy = ((float)oh + 0.5f) * 0.5f - 0.5f; sy = sy < 0.f ? 0.f : sy;
      float sx = ((float)ow + 0.5f) * 0.5f - 0.5f; sx = sx < 0.f ? 0.f : sx;
      const int y0 = (int)sy, x0 = (int)sx;
      const int y1 = y0 + (y0 < Hh - 1), x1 = x0 + (x0 < Wh - 1);
      const float ly1 = sy - (float)y0, ly0 = 1.f - ly1, lx1 = sx - (float)x0, lx0 = 1.f - lx1;
      const float w00 = ly0 * lx0, w01 = ly0 * lx1, w10 = ly1 * lx0, w11 = ly1 * lx1;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int h = head_of(nt), co = cout_of(h), c0 = (nt - first_of(h)) * 16 + 4 * q;
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = acc[nt][k] * 1.0f + be[nt][k];          // (the single-head epilogue with alpha = 1)
        if (p.res_up[h]) {
          const float* base = p.res_up[h] + (size_t)img * Hh * Wh * co + c0;
          const float4 v00 = *reinterpret_cast<const float4*>(base + ((size_t)y0 * Wh + x0) * co);
          const float4 v01 = *reinterpret_cast<const float4*>(base + ((size_t)y0 * Wh + x1) * co);
          const float4 v10 = *reinterpret_cast<const float4*>(base + ((size_t)y1 * Wh + x0) * co);
          const float4 v11 = *reinterpret_cast<const float4*>(base + ((size_t)y1 * Wh + x1) * co);
          // torch order: interpolate(...) + lat(x)  ->  up + o
          o[0] = __fmaf_rn(w11, v11.x, __fmaf_rn(w10, v10.x, __fmaf_rn(w00, v00.x, w01 * v01.x))) + o[0];
          o[1] = __fmaf_rn(w11, v11.y, __fmaf_rn(w10, v10.y, __fmaf_rn(w00, v00.y, w01 * v01.y))) + o[1];
          o[2] = __fmaf_rn(w11, v11.z, __fmaf_rn(w10, v10.z, __fmaf_rn(w00, v00.z, w01 * v01.z))) + o[2];
          o[3] = __fmaf_rn(w11, v11.w, __fmaf_rn(w10, v10.w, __fmaf_rn(w00, v00.w, w01 * v01.w))) + o[3];
        }
        *reinterpret_cast<float4*>(p.y[h] + (size_t)px * co + c0) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
}

template <int CIN, int C0, int C1, int C2, int U>
int launch_heads(P1M& p, hipStream_t st) {
  const long long runs = (p.npx + U * 16 - 1) / (U * 16);
  if (runs > 0x7fffffff) return mdf::fail(MDF_EARG, "conv1x1 heads: too many pixels");
  p.n_runs = (int)runs;
  long long blocks = (runs + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL((conv1x1_heads_kernel<CIN, C0, C1, C2, U>), dim3((unsigned)blocks), dim3(256), 0, st, p);
  return mdf::check_launch("conv1x1_heads_kernel");
}

template <int CIN, int COUT>
int launch_1x1(P1& p, hipStream_t st) {
  constexpr int NCH = CIN / 16;
  constexpr int U = (8 / NCH) < 1 ? 1 : 8 / NCH;
  const long long runs = (p.npx + U * 16 - 1) / (U * 16);
  if (runs > 0x7fffffff) return mdf::fail(MDF_EARG, "conv1x1: too many pixels");
  p.n_runs = (int)runs;
  long long blocks = (runs + 3) / 4;
  int cap = 256 * 8;
  if (const char* e = getenv("MDF_CONV1X1_BLOCKS")) { if (atoi(e) > 0) cap = atoi(e); }   // dev A/B
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL((conv1x1_kernel<CIN, COUT>), dim3((unsigned)blocks), dim3(256), 0, st, p);
  return mdf::check_launch("conv1x1_kernel");
}

}  // namespace

// Internal entry used by mdf_conv2d_fwd for ksize 1, stride 1, NHWC input.  MDF_EUNSUPPORTED: no instantiation (the caller
// falls back to the LDS kernels).
#define C1_CASE(ci, co) \
  if (Cin == ci && Cout == co) return launch_1x1<ci, co>(p, (hipStream_t)stream);

int mdf_conv1x1_dispatch(const float* x, const float* wpack, const float* alpha, const float* beta, const float* res, float res_scale,
                         const float* res_up, float* y, int B, int H, int W, int Cin, int Cout, int relu, void* stream) {
  if (const char* e = getenv("MDF_CONV1X1")) { if (atoi(e) == 0) return MDF_EUNSUPPORTED; }   // A/B and parity-test switch (read per call)
  P1 p{};
  p.x = x; p.wpack = wpack; p.alpha = alpha; p.beta = beta; p.res = res; p.res_up = res_up; p.y = y;
  p.res_scale = res_scale; p.relu = relu; p.Ho = H; p.Wo = W; p.npx = (long long)B * H * W;
  if (p.npx >= (1ll << 31) - 256) return MDF_EUNSUPPORTED;      // (pixel coordinates are 32-bit in the kernel)
  // (64 input channels stay on the LDS kernels: 2 tiles per run is all the registers allow, and that measured 1-2 us slower;
  //  non-temporal loads / stores measured slower too: 32->32 @296x400x5 50 -> 60 us inside a forward)
  C1_CASE(16, 16) C1_CASE(16, 32) C1_CASE(16, 64) C1_CASE(32, 16) C1_CASE(32, 32) C1_CASE(32, 64)
  return MDF_EUNSUPPORTED;
}

// Up to three bias-only 1x1 heads over one NHWC input in one launch; head h: y[h] [B,H,W,couts[h]] = [up2(res_ups[h]) +] W_h x + bias_h.
extern "C" int mdf_conv1x1_heads_fwd(const float* x, int n_heads, const float* const* wpacks, const float* const* biases,
                                     const float* const* res_ups, float* const* ys, const int* couts, int B, int H, int W, int Cin,
                                     void* stream) {
  MDF_REQUIRE(x && wpacks && biases && res_ups && ys && couts, "null pointer argument");
  MDF_REQUIRE(n_heads == 2 || n_heads == 3, "n_heads=%d not in {2,3}", n_heads);
  MDF_REQUIRE(B > 0 && H > 0 && W > 0, "bad shape");
  P1M p{};
  p.x = x; p.Ho = H; p.Wo = W; p.npx = (long long)B * H * W;
  bool any_up = false;
  for (int h = 0; h < n_heads; ++h) {
    MDF_REQUIRE(wpacks[h] && ys[h], "head %d: null weights or output", h);
    p.wpack[h] = wpacks[h]; p.beta[h] = biases[h]; p.res_up[h] = res_ups[h]; p.y[h] = ys[h];
    any_up = any_up || res_ups[h];
    MDF_REQUIRE((long long)p.npx * couts[h] < (1ll << 31) && (long long)p.npx * Cin < (1ll << 31), "map too large for 32-bit offsets");
  }
  MDF_REQUIRE(!any_up || (H % 2 == 0 && W % 2 == 0), "upsample-add needs even H and W");
  hipStream_t st = (hipStream_t)stream;
  // one tile per run for both (r04 sweep: 64 -> 64/32/16 with its weights in LDS 41.3 us at U = 1, 48.0 at 2, 42.5 with 112 weight
  // registers and one wave per SIMD; 32 -> 32/16 62.9 / 66.5 / 65.8 us at U = 1 / 2 / 4)
  if (Cin == 64 && n_heads == 3 && couts[0] == 64 && couts[1] == 32 && couts[2] == 16) return launch_heads<64, 64, 32, 16, 1>(p, st);
  if (Cin == 32 && n_heads == 2 && couts[0] == 32 && couts[1] == 16) return launch_heads<32, 32, 16, 0, 1>(p, st);
  return mdf::fail(MDF_EUNSUPPORTED, "conv1x1 heads: Cin=%d with %d heads is not built (64 -> 64/32/16, 32 -> 32/16)", Cin, n_heads);
}
