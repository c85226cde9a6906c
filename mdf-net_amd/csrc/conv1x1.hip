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
      int oh = 0, ow = 0;
      long long img = 0;
      if (p.res_up) {
        const long long rr = px / p.Wo;
        ow = (int)(px - rr * p.Wo);
        oh = (int)(rr % p.Ho);
        img = rr / p.Ho;
      }
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
  // (64 input channels stay on the LDS kernels: 2 tiles per run is all the registers allow, and that measured 1-2 us slower;
  //  non-temporal loads / stores measured slower too: 32->32 @296x400x5 50 -> 60 us inside a forward)
  C1_CASE(16, 16) C1_CASE(16, 32) C1_CASE(16, 64) C1_CASE(32, 16) C1_CASE(32, 32) C1_CASE(32, 64)
  return MDF_EUNSUPPORTED;
}
