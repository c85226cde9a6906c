// The `prob` head of a regulariser as ONE launch (mdf_prob_fused_fwd); see the block comment below.  Built on conv_lds_common.h.
#include "conv_lds_common.h"

namespace {

// ---- `prob` head in one launch (regular.py:66-69,129-133 + regress.py:5-7) -------------------------------------------------
// logit[d] = P0[d-1] + P1[d] + P2[d+1], P_kd[z] = the 2-D (kh, kw, cin) contraction of input plane z with the kd-th slice of
// the 3x3x3 kernel (prob_head.hip).  The two-launch route runs the partials as a 4-channel 2-D conv over the B*D planes
// (Cfg<Cin,Cin,4,1,3,1,1,4>: 4 pixels along w per MFMA column, GEMM row = phase*4 + kd) and writes / re-reads them: 16 B per
// voxel each way next to the 4*Cin B the layer has to read.  Here a block owns ONE 4 x 64 pixel tile and walks the depth axis
// with that same MFMA step: a lane's accumulator is (P0, P1, P2, 0) of its pixel, the two-plane delay line and the running
// maximum live in registers, the D logits of a pixel wait in LDS (own column: no barrier; parked in global memory the two
// normalisation passes were a chain of D dependent L2 round trips per thread) and `prob` is written once.  Same partial sums, same combine / softmax / soft-argmin arithmetic and order as conv + prob_from_partials:
// bit-identical results, 2/3 of the traffic.  (Parallelism is spatial only: tiles = B*ceil(h/4)*ceil(w/64) -- the caller keeps
// the two-launch route for maps too small to fill the chip.)
struct ProbParams {
  const float* x;       // [B*D][H][W][CIN]
  const float* wpack;   // w-phase segment of pack_conv2d_weight([kd (+ zero), cin, 3, 3])
  const float* hypos;   // [B,D] or [B,D,H,W] or null
  int per_pixel;
  float* prob;          // [B,D,H,W]
  float* depth;         // [B,H,W] or null
  int B, D, H, W, tiles_h, tiles_w;
};

// the w-phase configuration of the partial-sum conv with a tile of NW rows (one per wave).  NW = 8 re-reads less halo
// ((NW + 2) / NW rows) but halves the blocks: 68 vs 69 us at 8x592x800, 108 vs 72 us at 24x296x400 -- 4 it is
template <int CIN, int NW>
struct ProbCfg : Cfg<CIN, CIN, 4, 1, 3, 1, 1, 4> {
  typedef Cfg<CIN, CIN, 4, 1, 3, 1, 1, 4> Base;
  static constexpr int TH = NW, PH = NW + 2;
  static constexpr int S = round_s(PH * Base::PW, Base::KPL, Base::SW);
  static constexpr int PLANE = CIN * S;
  static constexpr int NTHR = 64 * NW;
  static constexpr int NFILL = (Base::NG * PH * Base::PW + NTHR - 1) / NTHR;
};

// DT = D at compile time (24 / 8: the stages this kernel serves) or 0.  With DT the pixel's hypotheses are requested BEFORE the walk over
// the planes and wait in registers; read inside the last loop (runtime trip count, a division and a cascade sum per plane between two
// loads) they were D memory round trips at the end of every tile.
template <int CIN, int NW, int DT>
__global__ __launch_bounds__(64 * NW) void prob_fused_kernel(const ProbParams p) {
  typedef ProbCfg<CIN, NW> C;
  constexpr int KPL = C::KPL, NG = C::NG, S = C::S, PW = C::PW, PH = C::PH, NTHR = C::NTHR;
  typedef typename VecT<KPL>::type vec_t;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, n16 = lane & 15;
  const __amdgpu_buffer_rsrc_t wres = make_rsrc(p.wpack, (unsigned)(C::NSTEP * C::NT * 64 * KPL * 4));
  const int wvoff = lane * KPL * 4;
  const int lane_lds = (q * S + wave * PW + n16 * C::SW) * KPL;
  float wfirst[2][C::NT][KPL], wr[C::WN][C::NT][KPL];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if constexpr (!C::WREG) buf_load_to<KPL>(wres, wvoff, i * (64 * KPL * 4), wfirst[i][0]);
    else {
#pragma unroll
      for (int k = 0; k < KPL; ++k) wfirst[i][0][k] = 0.f;
    }
  }
  if constexpr (C::WREG) {
#pragma unroll
    for (int i = 0; i < C::NSTEP; ++i) buf_load_to<KPL>(wres, wvoff, i * (64 * KPL * 4), wr[i][0]);
  } else {
#pragma unroll
    for (int k = 0; k < KPL; ++k) wr[0][0][k] = 0.f;
  }
  int sp = blockIdx.x;
  const int twi = sp % p.tiles_w; sp /= p.tiles_w;
  const int th = sp % p.tiles_h;
  const int b = sp / p.tiles_h;
  const int th0 = th * C::TH, tw0 = twi * C::TWO;

  // plane tile -> LDS: the mapping of conv_lds_kernel's 2-D path (cin group fastest: coalesced 16-B pieces)
  constexpr int GF2 = (NG < 4) ? NG : 4;
  auto split2 = [&](int idx, int& v, int& g) {
    const int glo = idx % GF2, r = idx / GF2;
    v = r % (PH * PW);
    g = (r / (PH * PW)) * GF2 + glo;
  };
  auto load2 = [&](int idx, int d) -> vec_t {
    int g, v;
    split2(idx, v, g);
    const int row = v / PW, col = v - row * PW;
    const int ih = th0 - 1 + row, iw = tw0 - 1 + col;
    if (idx >= NG * PH * PW || ih < 0 || ih >= p.H || iw < 0 || iw >= p.W) return vec_zero<KPL>();
    return *reinterpret_cast<const vec_t*>(p.x + ((((size_t)b * p.D + d) * p.H + ih) * p.W + iw) * CIN + g * KPL);
  };
  auto store2 = [&](int idx, int slot, const vec_t& val) {
    if (idx < NG * PH * PW) {
      int g, v;
      split2(idx, v, g);
      *reinterpret_cast<vec_t*>(lds + slot * C::PLANE + (g * S + v) * KPL) = val;
    }
  };
#pragma unroll
  for (int k = 0; k < C::NFILL; ++k) store2(tid + k * NTHR, 0, load2(tid + k * NTHR, 0));
  __syncthreads();

  // this lane's pixel: row th0 + wave, column tw0 + 4*n16 + q (GEMM rows 4q..4q+3 = phase q, channels P0 P1 P2 0)
  const int oh = th0 + wave, ow = tw0 + 4 * n16 + q;
  const bool row_live = oh < p.H;
  const bool live = row_live && ow < p.W;
  const size_t hw = (size_t)p.H * p.W;
  const size_t pix = live ? (size_t)oh * p.W + ow : 0;
  float* pr = p.prob + (size_t)b * p.D * hw + pix;
  float* lg = lds + 2 * C::PLANE + tid;   // [D][NTHR]: this thread's logits
  float hy[DT ? DT : 1];
  if constexpr (DT != 0) {
    if (p.depth && live) {
#pragma unroll
      for (int d = 0; d < DT; ++d) hy[d] = p.per_pixel ? p.hypos[((size_t)b * DT + d) * hw + pix] : p.hypos[(size_t)b * DT + d];
    }
  }
  float carry1 = 0.f, carry0 = 0.f;   // P0[d-1] + P1[d] (awaiting P2[d+1]);  P0[d] (feeds logit[d+1])
  float mx = -INFINITY;
  for (int d = 0; d < p.D; ++d) {
    const int slot = d & 1;
    const bool more = d + 1 < p.D;
    vec_t pf[C::NFILL];
    if (more) {
#pragma unroll
      for (int k = 0; k < C::NFILL; ++k) pf[k] = load2(tid + k * NTHR, d + 1);
    }
    if (row_live) {
      const float* planes[1] = {lds + slot * C::PLANE + lane_lds};
      f32x4 acc[1][C::NT];
      step_mfma<C, 1, 3, 1>(planes, wres, wvoff, wr, wfirst, acc);
      const float done = carry1 + acc[0][0][2];      // logit[d-1] complete
      if (d >= 1) { lg[(d - 1) * NTHR] = done; mx = fmaxf(mx, done); }
      carry1 = carry0 + acc[0][0][1];
      carry0 = acc[0][0][0];
    }
    if (more) {
#pragma unroll
      for (int k = 0; k < C::NFILL; ++k) store2(tid + k * NTHR, slot ^ 1, pf[k]);
      __syncthreads();
    }
  }
  if (!live) return;
  lg[(p.D - 1) * NTHR] = carry1;
  mx = fmaxf(mx, carry1);
  float sum = 0.f;
  for (int d = 0; d < p.D; ++d) {
    const float e = expf(lg[d * NTHR] - mx);
    lg[d * NTHR] = e;
    sum += e;
  }
  mdf::CascadeSum dep;   // regress.py:5-7 with ATen's summation order
  if constexpr (DT != 0) {
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const float pv = lg[d * NTHR] / sum;
      pr[(size_t)d * hw] = pv;
      if (p.depth) dep.add(pv * hy[d]);
    }
  } else {
    for (int d = 0; d < p.D; ++d) {
      const float pv = lg[d * NTHR] / sum;
      pr[(size_t)d * hw] = pv;
      if (p.depth) dep.add(pv * (p.per_pixel ? p.hypos[((size_t)b * p.D + d) * hw + pix] : p.hypos[(size_t)b * p.D + d]));
    }
  }
  if (p.depth) p.depth[(size_t)b * hw + pix] = dep.result();
}

template <int CIN, int NW, int DT>
int launch_prob_fused_d(ProbParams& p, hipStream_t st) {
  typedef ProbCfg<CIN, NW> C;
  const size_t kLds = ((size_t)2 * C::PLANE + (size_t)p.D * C::NTHR) * sizeof(float);
  if (kLds > 160 * 1024) return mdf::fail(MDF_EUNSUPPORTED, "fused prob head: D=%d does not fit the LDS (use the two-launch route)", p.D);
  p.tiles_h = (p.H + C::TH - 1) / C::TH;
  p.tiles_w = (p.W + C::TWO - 1) / C::TWO;
  const long long tiles = (long long)p.B * p.tiles_h * p.tiles_w;
  if (tiles > 0x7fffffff) return mdf::fail(MDF_EARG, "prob head: too many tiles");
  static bool attr_done_dev[64] = {};   // (per instantiation; the attribute is set to the device maximum once: the size depends on D)
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  bool& attr_done = attr_done_dev[(dev_id >= 0 && dev_id < 64) ? dev_id : 0];
  if (!attr_done || dev_id >= 64) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&prob_fused_kernel<CIN, NW, DT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return mdf::fail(MDF_EHIP, "hipFuncSetAttribute(dynamic LDS): %s", hipGetErrorString(e));
    attr_done = true;
  }
  hipLaunchKernelGGL((prob_fused_kernel<CIN, NW, DT>), dim3((unsigned)tiles), dim3(C::NTHR), kLds, st, p);
  return mdf::check_launch("prob_fused_kernel");
}

template <int CIN, int NW>
int launch_prob_fused(ProbParams& p, hipStream_t st) {
  if (p.D == 24) return launch_prob_fused_d<CIN, NW, 24>(p, st);
  if (p.D == 8) return launch_prob_fused_d<CIN, NW, 8>(p, st);
  return launch_prob_fused_d<CIN, NW, 0>(p, st);
}

}  // namespace

extern "C" int mdf_prob_fused_fwd(const float* x, const float* wpack, const float* hypos, int hypos_per_pixel, float* prob, float* depth,
                                  int B, int D, int h, int wd, int Cin, void* stream) {
  MDF_REQUIRE(x && wpack && prob, "null pointer argument");
  MDF_REQUIRE(depth == nullptr || hypos != nullptr, "depth output needs hypos");
  MDF_REQUIRE(B > 0 && D > 0 && h > 0 && wd > 0, "bad shape");
  MDF_REQUIRE((long long)B * D * h * wd * Cin < (1ll << 31), "input too large for 32-bit offsets");
  ProbParams p{};
  p.x = x; p.hypos = hypos; p.per_pixel = hypos_per_pixel; p.prob = prob; p.depth = depth; p.B = B; p.D = D; p.H = h; p.W = wd;
  p.wpack = wpack + (size_t)9 * Cin * 16;      // behind the plain fragments (conv3d.hip pack_segments), as LDS_CASE_RW reads them
  if (Cin == 8) return launch_prob_fused<8, 4>(p, (hipStream_t)stream);
  if (Cin == 16) return launch_prob_fused<16, 4>(p, (hipStream_t)stream);
  return mdf::fail(MDF_EUNSUPPORTED, "fused prob head is built for Cin in {8,16}, got %d", Cin);
}
