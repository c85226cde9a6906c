// Device-side building blocks shared by the LDS-staged conv kernels (conv_lds.hip) and the fused kernels built on its MFMA
// step (prob_fused.hip, refine_tail.hip): fragment loads, the LDS plane layout (Cfg, round_s) and step_mfma.  Everything here
// has internal linkage: each translation unit gets its own copy.
#pragma once
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {


typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LdsConvParams {
  const float* x;      // [B,D,H,W,CIN_MEM]
  const float* wpack;  // conv3d.hip packing (taps = KD*KHW*KHW)
  const float* alpha;  // [COUT] or null
  const float* beta;   // [COUT] or null (bias when alpha is null)
  const float* res;    // [B,D,Ho,Wo,COUT] or null
  const float* res_up; // [B,D,Ho/2,Wo/2,COUT] or null: bilinear x2 (align_corners=False) of it is added (FPN top-down)
  float* y;            // [B,D,Ho,Wo,COUT]
  float res_scale;     // y = res + res_scale * act(...)  (Res block: x + 0.1*conv)
  int B, D, H, W, Ho, Wo;
  int relu;
  int tiles_h, tiles_w, dchunks, dch;  // item space (3-D: tile x depth chunk)
  int n_items;
  int n_tiles, tiles_per_item;         // 2-D: an item is a run of consecutive tiles
  int planar_in;                       // 2-D, CIN_MEM != CIN: input is planar [B,CIN_MEM,H,W] (e.g. the RGB images as they arrive)
  int sched_slot;                      // which pair of g_sched words this launch uses (one per stream)
  int shuffle2;                        // 2-D, Cout = 32: write PixelShuffle(2) of the result ([B,2Ho,2Wo,8]); rows packed sub-pixel-major
  int prefetch_early;                  // 3-D: issue the next plane's loads before (1) or after (0) the MFMA block
  int s2d;                             // 2-D: x is [B,H,W,CIN/4] at TWICE the resolution; the kernel's input is its four parity images as
                                       // channels (py*2+px)*CIN/4 + c (space-to-depth done by the tile fill): a k5 s2 layer as a 3x3 conv
  // training (ST kernels): per-channel sums of the output in the epilogue, see conv3d.hip ConvParams::stat_mode
  int stat_mode;
  const float* stat_y;
  const float* stat_aux;               // [groups][4*COUT]
  double* stat_out;                    // [groups][2*COUT]
  int stat_groups;                     // 2-D: BatchNorm groups (consecutive sets of B / groups images); the ST grid is groups x blocks-per-group
  int stat_slices;                     // slices of stat_out [slices][groups][2C] the blocks are spread over (common.h: conv_stat_send)
};

// fp64 LDS add and the DPP sum over the 16 lanes that hold the 16 MFMA columns of one 4-channel row group
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

template <int N> struct VecT;
template <> struct VecT<4> { typedef float4 type; };
template <> struct VecT<2> { typedef float2 type; };
template <> struct VecT<1> { typedef float type; };

template <int KPL> __device__ __forceinline__ void vec_to(const typename VecT<KPL>::type& v, float* o);
template <> __device__ __forceinline__ void vec_to<4>(const float4& v, float* o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <> __device__ __forceinline__ void vec_to<2>(const float2& v, float* o) { o[0] = v.x; o[1] = v.y; }
template <> __device__ __forceinline__ void vec_to<1>(const float& v, float* o) { o[0] = v; }

// Weight fragments are fetched with BUFFER loads: the address is {SGPR resource, one lane-offset VGPR that never changes,
// scalar/immediate fragment offset}.  With flat global loads every fragment needed its own 64-bit VGPR address (held in
// registers or re-added on the VALU): 150-300 address pairs per kernel, the reason the pipelined tap loop sat at its
// register cap and spilled (scripts/isa_stats.py).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);   // raw buffer, dword format (gfx9)
}
template <int KPL> __device__ __forceinline__ void buf_load_to(__amdgpu_buffer_rsrc_t r, int voff, int soff, float* o);
template <> __device__ __forceinline__ void buf_load_to<4>(__amdgpu_buffer_rsrc_t r, int voff, int soff, float* o) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
}
template <> __device__ __forceinline__ void buf_load_to<2>(__amdgpu_buffer_rsrc_t r, int voff, int soff, float* o) {
  const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y);
}
template <> __device__ __forceinline__ void buf_load_to<1>(__amdgpu_buffer_rsrc_t r, int voff, int soff, float* o) {
  o[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// B-fragment read from LDS.  KPL = 2: hipcc merges two neighbouring 8-byte reads (taps kw, kw+1) into ONE ds_read2_b64, which
// the LDS serves at half the rate of ds_read_b64 (8 cycles per wave instead of 2 x 2) and banks modulo 32 instead of 64 -- the
// layouts chosen for ds_read_b64 (round_s) then conflict: SQ_LDS_BANK_CONFLICT was 90 % of the LDS cycles of the 8-channel
// kernels.  A volatile access is not merged.
typedef float lds_f2 __attribute__((ext_vector_type(2)));
template <int KPL> __device__ __forceinline__ void lds_frag(const float* p, float* o) {
  if constexpr (KPL == 2) {
    typedef const volatile __attribute__((address_space(3))) lds_f2* lds_ptr;   // (explicitly LDS: a volatile generic access becomes a flat load)
    const lds_f2 v = *(lds_ptr)(p);
    o[0] = v.x; o[1] = v.y;
  } else {
    vec_to<KPL>(*reinterpret_cast<const typename VecT<KPL>::type*>(p), o);
  }
}

template <int KPL> __device__ __forceinline__ typename VecT<KPL>::type vec_zero();
template <> __device__ __forceinline__ float4 vec_zero<4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ __forceinline__ float2 vec_zero<2>() { return make_float2(0.f, 0.f); }
template <> __device__ __forceinline__ float vec_zero<1>() { return 0.f; }

// Plane stride S (in KPL-float vectors per cin group) such that the B-fragment reads of a lane group fall on distinct banks
// (MI355X LDS: ds_read_b128 is served in 16-lane groups that mix two q rows, ds_read_b64/b32 in 32-lane halves = two q rows;
// bank = dword address mod 64, mod 32 for b32 and all stores).  A lane reads vector q*S + n16*SW (+ tap shift):
//  SW = 1: the two q rows of a group must land 16 vectors (KPL 4) / 32 dwords (KPL 2) apart: S % 32 == 16;
//  SW = 2 (stride-2 layers, w-phase RW = 2, Winograd): each row covers the even slots, so the neighbour row needs an odd S --
//          with S % 32 == 16 every one of those reads was a 2-way conflict.  Among the odd residues the fill stores (cin group
//          fastest over 4 groups: vectors g*S + v) are conflict-free for S % 8 == 3 (KPL 4) and nearly so for S % 16 == 5 (KPL 2);
//  SW = 4 (w-phase RW = 4): a row alone is 2-way (n16 and n16 + 4/8 share banks); S % 4 == 2 keeps the two rows apart (was 4-way).
constexpr int round_up_mod(int n, int m, int r) { return n + ((r - n % m) + m) % m; }
constexpr int round_s(int n, int kpl, int sw) {
  if (sw == 1) return round_up_mod(n, 32, 16);
  if (sw % 4 == 0) return round_up_mod(n, 4, 2);
  return kpl == 4 ? round_up_mod(n, 8, 3) : (kpl == 2 ? round_up_mod(n, 16, 5) : round_up_mod(n, 2, 1));
}

#ifndef EARLY2_MAX_REGS
#define EARLY2_MAX_REGS 32
#endif
// RW > 1 ("w-phase" form, for Cout < 16): an MFMA output tile has 16 rows, so a Cout = 8 (4) layer would waste half
// (three quarters) of every MFMA.  Instead RW = 2 (4) neighbouring output voxels along w share one MFMA column: GEMM row
// r*Cout + c is channel c of output voxel RW*m + r, which is the same conv written with stride RW, Cout' = RW*Cout and a
// kernel of KHW + RW - 1 taps along w (tap kw' of phase r is the original tap kw' - r, zero outside) -- KW' taps serve RW
// outputs instead of RW*KHW: 4 vs 6 (RW 2), 6 vs 12 (RW 4) MFMAs.  The expanded weights are packed by conv3d.hip.
//
// WG = 1 (Winograd form, 3-D stride-1 3x3x3 layers): F(2x2, 3x3) in (h, w), direct in d.  An MFMA column is a 2x2 output
// tile; per input plane and 16-cin chunk a lane reads its tile's 4x4 patch from LDS, transforms it (B^T d B, 32 adds per
// channel), and feeds 16 transform-domain GEMMs (one accumulator each); the output transform A^T M A runs in the epilogue.
// 16 x 3 instead of 9 x 4 x 3 MFMAs per 4 outputs: 2.25x fewer.  fp32 Winograd F(2,3) costs no accuracy here: single-layer
// error vs fp64 1.1e-6 (direct 1.8e-6), end-to-end depth deviation 3.9-4.4e-4 mm = the floor of any fp32 re-ordering
// (scripts/study_winograd.py).  Geometry: RW = 2 supplies the w bookkeeping (4-wide patch, stride 2); a wave owns 2 rows.
//
// WG = 2 (depth-pair Winograd, 3-D layers with 8 output channels): with Cout = 8 the Winograd GEMMs would fill only half of
// the 16 MFMA rows.  Here a step produces TWO output planes d, d+1: GEMM row r*8 + c is channel c of plane d + r, and the four
// input planes d-1 .. d+2 of the step (ring of 4) each meet tap kd = j - r (zero rows where that is outside 0..2).  Every
// transformed patch feeds both planes: 4 x 16 instead of 2 x 3 x 16 MFMA groups (and 4 instead of 6 patch transforms) per
// pair of planes, 2.25x fewer MFMAs than the w-phase form.  Also for Cin = 8 (two k-steps per group).
template <int CIN, int CIN_MEM, int COUT, int KD, int KHW, int SHW, int MT, int RW = 1, int WG = 0>
struct Cfg {
  static_assert(RW == 1 || WG != 0 || (SHW == 1 && COUT * RW <= 16 && COUT % 4 == 0), "w-phase form: stride 1, RW*Cout <= 16, Cout % 4 == 0");
  static_assert(WG == 0 || ((KD == 3 || KD == 1) && KHW == 3 && SHW == 1 && RW == 2 && MT == 1 && CIN % (WG >= 2 ? 8 : 16) == 0 && COUT % 4 == 0), "Winograd form: 3x3(x3) stride 1, RW = 2, MT = 1");
  static_assert(WG < 2 || (KD == 3 && COUT == 8), "depth-pair Winograd form: 3-D, Cout = 8");
  static constexpr bool WINO = (WG != 0);
  static constexpr int RD = (WG >= 2) ? 2 : 1;    // output planes per depth step
  // WG = 3: the depth-pair form on a ring of THREE planes: the pair's first three input planes are multiplied, the fourth takes the
  // oldest plane's slot and is multiplied next (conv_lds_kernel, STREAM path) -- a 16-channel plane tile is 22 KB, and three of them
  // leave room for a second block per CU where four do not
  static constexpr bool STREAM = (WG == 3);
  static constexpr int NPL = KD + RD - 1;         // input planes a depth step reads
  static constexpr int NTP = ((COUT + 15) / 16 > 2) ? 2 : (COUT + 15) / 16;   // Winograd: n-tiles per pass over K (16 accumulators each)
  static constexpr int CIN_ = CIN;
  static constexpr int RWF = RW;
  static constexpr int KW = KHW + RW - 1;       // taps along w
  static constexpr int SW = SHW * RW;           // input step along w between neighbouring MFMA columns
  static constexpr int ROWS = WINO ? COUT * RD : COUT * RW;   // GEMM rows
  static constexpr int KPL = (CIN >= 16) ? 4 : (CIN == 8 ? 2 : 1);
  static constexpr int CK = 4 * KPL;
  static constexpr int NCH = CIN / CK;
  static constexpr int NG = CIN / KPL;  // k-groups per voxel
  static constexpr int NT = (ROWS + 15) / 16;
  static constexpr int WROWS = WINO ? 2 : 1;    // output rows per wave
  static constexpr int TH = 4 * WROWS, TW = 16 * MT;   // tile: TH rows x TW MFMA columns = TW*RW output voxels along w
  static constexpr int TWO = TW * RW;
  static constexpr int PAD = (KHW - 1) / 2, PD = (KD - 1) / 2;
  static constexpr int PH = WINO ? TH + 2 : (TH - 1) * SHW + KHW, PW = (TW - 1) * SW + KW;
  static constexpr int S = round_s(PH * PW, KPL, SW);
  static constexpr int PLANE = CIN * S;  // floats
  static constexpr int NFILL = (NG * PH * PW + 255) / 256;
  // 3-D: rolling window of KD planes; 2-D: double-buffered tiles -- except the 64-channel Winograd form, whose 10x34 tile
  // (87 KB) fits once: single buffer, the next tile's loads wait in registers during the (long) compute
  static constexpr int RING = (KD > 1) ? (STREAM ? KD : NPL) : ((WINO && CIN >= 64) ? 1 : 2);
  static constexpr int NSTEP = WINO ? NPL * 16 * NCH : KD * KHW * KW * NCH;   // MFMA pipeline steps per output row-tile (tap x cin chunk)
  // small layers keep ALL their weight fragments in registers for the whole kernel (<= 40 VGPRs; beyond that occupancy drops and it is a loss, measured) instead of re-fetching
  // them from L1 for every tile: with 8-16 MFMAs per step there is nothing to hide that round trip behind
  static constexpr bool WREG = (KD == 1) && !WINO && (NSTEP * NT * KPL <= 40);
  // epilogue scale/shift hoisted out of the tile loop where registers allow (the 3-D and 4-n-tile kernels sit at their caps)
  static constexpr bool EPI_REG = (KD == 1) && (NT <= 2);
  static constexpr int WN = WREG ? NSTEP : 1;
  // 2-D: issue the next tile's global loads before this tile's MFMAs when the staging registers are cheap
  static constexpr bool EARLY2 = (KD == 1) && ((!WINO && (NFILL * KPL <= EARLY2_MAX_REGS)) || (WINO && CIN >= 64));
  // + the broadcast slot of the item id (16 B) + the epilogue table: alpha[64], beta[64] (read per step from LDS instead of
  // from global memory: the per-call L1/L2 round trip was ~1000 exposed cycles per depth step, in-kernel stamps)
  static constexpr int EPI_OFF = RING * PLANE + 4;   // floats
  static constexpr size_t LDS_BYTES = (size_t)RING * PLANE * sizeof(float) + 16 + 128 * sizeof(float);
  // ST kernels: + fp64 sums [2][64] + the producing layer's (a, b, mean, invstd) [4][64]
  static constexpr int STAT_OFF = EPI_OFF + 128, SAUX_OFF = STAT_OFF + 256;   // floats (STAT_OFF*4 is a multiple of 8)
  static constexpr size_t LDS_BYTES_ST = LDS_BYTES + 512 * sizeof(float);
};

// ST epilogue: the lane's 4 output values o[] of channels c0.. at output index oi -> its running sums (ps, pq)
template <typename C, int ST>
__device__ __forceinline__ void stat_accum(const LdsConvParams& p, const float* lds_base, size_t oi, int c0, const float (&o)[4],
                                           float (&ps)[4], float (&pq)[4]) {
  if constexpr (ST == 1) {     // (the two modes are separate instantiations: together they cost the 2-D kernels an occupancy step)
#pragma unroll
    for (int k = 0; k < 4; ++k) { ps[k] += o[k]; pq[k] = fmaf(o[k], o[k], pq[k]); }
  } else {
    const float* ax = lds_base + C::SAUX_OFF;
    const float4 yv4 = *reinterpret_cast<const float4*>(p.stat_y + oi);
    const float yv[4] = {yv4.x, yv4.y, yv4.z, yv4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float dr = (fmaf(yv[k], ax[c0 + k], ax[64 + c0 + k]) > 0.0f) ? o[k] : 0.0f;
      ps[k] += dr;
      pq[k] = fmaf(dr, (yv[k] - ax[128 + c0 + k]) * ax[192 + c0 + k], pq[k]);
    }
  }
}
template <typename C>
__device__ __forceinline__ void stat_commit(float* lds_base, int c0, int n16, const float (&ps)[4], const float (&pq)[4]) {
  double* tab = reinterpret_cast<double*>(lds_base + C::STAT_OFF);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float a1 = row16_sum(ps[k]), a2 = row16_sum(pq[k]);
    if (n16 == 0) { lds_add_f64(&tab[c0 + k], (double)a1); lds_add_f64(&tab[64 + c0 + k], (double)a2); }
  }
}

// The MFMA part of one output row-tile of one depth plane: MTL live m-tiles (16 voxels each) x all GEMM rows, accumulated into
// acc (zeroed here).  Fully unrolled over the taps: LDS offsets are immediates, no bounds logic (halos are zero-filled in LDS).
template <typename C, int KD, int KHW, int MTL>
__device__ __forceinline__ void step_mfma(const float* const (&planes)[KD], __amdgpu_buffer_rsrc_t wres, int wvoff,
                                          const float (&wr)[C::WN][C::NT][C::KPL], const float (&wfirst)[2][C::NT][C::KPL],
                                          f32x4 (&acc)[MTL][C::NT]) {
  constexpr int KPL = C::KPL, NCH = C::NCH, NT = C::NT, S = C::S, PW = C::PW;
#pragma unroll
  for (int t = 0; t < MTL; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Software pipeline over the flattened (kd,kh,kw,chunk) steps.  hipcc otherwise places every operand load right
  // before its MFMAs (prefetch distance <= 1), leaving the matrix pipe idle for an L1/L2 round trip per step.
  // Weight fragments (global, L1-resident) run AHEAD_A steps ahead, LDS activation fragments one step ahead; the
  // order is pinned with sched_barrier, the counted s_waitcnt is left to the compiler.
  constexpr int NSTEP = C::NSTEP;
  constexpr int AHEAD_A = (NSTEP >= 3) ? 2 : (NSTEP - 1 > 0 ? NSTEP - 1 : 0);
  constexpr int NA = AHEAD_A + 1;
  float af[NA][NT][KPL], bf[2][MTL][KPL];
  auto load_a = [&](int i, int buf) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      buf_load_to<KPL>(wres, wvoff, (i * NT + nt) * (64 * KPL * 4), af[buf][nt]);
  };
  auto load_b = [&](int i, int buf) {
    const int ch = i % NCH, tap = i / NCH;
    const int kw = tap % C::KW, kh = (tap / C::KW) % KHW, kd = tap / (KHW * C::KW);
#pragma unroll
    for (int t = 0; t < MTL; ++t)
      lds_frag<KPL>(planes[kd] + ((ch * 4) * S + kh * PW + kw + t * 16 * C::SW) * KPL, bf[buf][t]);
  };
  if constexpr (!C::WREG) {
    // the first fragments of every call are the same: they stay in registers for the whole kernel (wfirst), so a step
    // does not begin with an exposed L1/L2 round trip
#pragma unroll
    for (int i = 0; i < AHEAD_A; ++i) {
      if (i < 2) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int k = 0; k < KPL; ++k) af[i % NA][nt][k] = wfirst[i][nt][k];
      } else {
        load_a(i, i % NA);
      }
    }
  }
  load_b(0, 0);
#pragma unroll
  for (int i = 0; i < NSTEP; ++i) {
    if constexpr (!C::WREG) {
      if (i + AHEAD_A < NSTEP) load_a(i + AHEAD_A, (i + AHEAD_A) % NA);
    }
    if (i + 1 < NSTEP) load_b(i + 1, (i + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KPL; ++s)
#pragma unroll
      for (int t = 0; t < MTL; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float a = C::WREG ? wr[C::WREG ? i : 0][nt][s] : af[i % NA][nt][s];
          acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bf[i & 1][t][s], acc[t][nt], 0, 0, 0);
        }
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace
