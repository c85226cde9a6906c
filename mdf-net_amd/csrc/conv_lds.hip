// conv v2: implicit-GEMM convolution with LDS-staged activation planes (fp32 MFMA 16x16x4), for the layers
// that carry most of the work: stride-1 3x3x3 layers of the regularisers on large volumes, and (KD = 1) the
// 2-D 3x3 / 5x5(stride 2) / 1x1 layers of the feature pyramid and refinement net.
//
// Why: conv v1 (conv3d.hip) fetches every B fragment (16 voxels x 16 cin) from L1/L2 once per tap, i.e. each
// input voxel 27 times per output tile.  Here a block owns a TH x TW output tile and walks the depth axis with a
// rolling window of KD input planes in LDS (halo included, zero-filled outside the volume, so the MFMA loop has
// no bounds masks): every input voxel is fetched from global memory once per tile, B fragments come from LDS by
// conflict-free ds_read_b128/b64 with immediate offsets (all taps unrolled), the next plane is prefetched into
// registers while the current one is multiplied, and only the small packed-weight fragments stream through L1.
//
// LDS image of one plane:  [cin / KPL][S] vectors of KPL floats, S = plane-tile voxels rounded up to the residue (round_s)
// that puts the lane groups of ds_read_b128 (and the two halves of ds_read_b64) on disjoint banks for every tap shift.
// GEMM orientation and weight packing are those of conv3d.hip:  D[cout][voxel] = W[cout][k] * X[k][voxel].
#include <cstdlib>
#include <mutex>
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

// Dynamic work distribution: per stream slot [0] next item, [1] finished blocks.  Module-level device words (nothing is
// allocated); the last block of a launch resets both, so every launch starts from zero.  Launches on DIFFERENT streams
// may overlap (two items in flight): each stream gets its own slot (host table below, kSchedSlots streams per process).
constexpr int kSchedSlots = 128;   // > torch's per-device stream pool (2 x 32 + default)
__device__ unsigned g_sched[2 * kSchedSlots];
#ifdef MDF_STAMPS
// diagnostic build only (scripts/diag_conv_stamps.sh): cycles spent by wave 0 of every block in each phase
__device__ unsigned long long g_stamps[8];   // sched, prologue, compute, refill, total, blocks, items, dsteps
#define STAMP() __builtin_readcyclecounter()
#endif

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
  static_assert(WG == 0 || ((KD == 3 || KD == 1) && KHW == 3 && SHW == 1 && RW == 2 && MT == 1 && CIN % (WG == 2 ? 8 : 16) == 0 && COUT % 4 == 0), "Winograd form: 3x3(x3) stride 1, RW = 2, MT = 1");
  static_assert(WG != 2 || (KD == 3 && COUT == 8), "depth-pair Winograd form: 3-D, Cout = 8");
  static constexpr bool WINO = (WG != 0);
  static constexpr int RD = (WG == 2) ? 2 : 1;    // output planes per depth step
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
  static constexpr int RING = (KD > 1) ? NPL : ((WINO && CIN >= 64) ? 1 : 2);
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

// One output row-tile of one depth plane: the MFMAs above, then the epilogue.
template <typename C, int KD, int KHW, int SHW, int COUT, int MTL, int ST = 0>
__device__ __forceinline__ void step(const float* const (&planes)[KD], __amdgpu_buffer_rsrc_t wres, int wvoff, const LdsConvParams& p,
                                     size_t row_vox, int w0, int q, int n16, const float (&wr)[C::WN][C::NT][C::KPL],
                                     const float (&al)[C::NT][4], const float (&be)[C::NT][4], const float (&wfirst)[2][C::NT][C::KPL]) {
  constexpr int NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) float lds_base_[];
  const float* epi_tab = lds_base_ + C::EPI_OFF;
  static_assert(ST == 0 || COUT % 4 == 0, "epilogue sums need Cout % 4 == 0");
  f32x4 acc[MTL][NT];
  step_mfma<C, KD, KHW, MTL>(planes, wres, wvoff, wr, wfirst, acc);
  // epilogue: lane owns GEMM rows nt*16 + 4q .. +3 of MFMA column t*16 + n16, i.e. couts c0..c0+3 of output voxel
  // (row, w0 + (t*16 + n16)*RW + phase)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int row0 = nt * 16 + 4 * q;
    if (row0 >= C::ROWS) continue;
    const int phase = (C::RWF > 1) ? row0 / COUT : 0;
    const int c0 = (C::RWF > 1) ? row0 % COUT : row0;
    float al_l[4], be_l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (C::EPI_REG) {
        al_l[k] = al[nt][k]; be_l[k] = be[nt][k];
      } else {
        al_l[k] = epi_tab[c0 + k];
        be_l[k] = epi_tab[64 + c0 + k];
      }
    }
    float ps[4] = {0.f, 0.f, 0.f, 0.f}, pq[4] = {0.f, 0.f, 0.f, 0.f};   // ST: this lane's sums over its m-tiles
#pragma unroll
    for (int t = 0; t < MTL; ++t) {
      const int ow = w0 + (t * 16 + n16) * C::RWF + phase;
      if (ow >= p.Wo) continue;
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[k] = acc[t][nt][k] * al_l[k] + be_l[k];
        if (p.relu) o[k] = fmaxf(o[k], 0.f);
      }
      size_t oi = (row_vox + ow) * COUT + c0;
      if constexpr (COUT == 32 && KD == 1 && C::RWF == 1) {
        if (p.shuffle2) {   // GEMM row = sub*8 + oc (sub = dy*2+dx): the lane's 4 rows are 4 channels of output pixel (2h+dy, 2w+dx)
          const int sub = c0 >> 3, oc0 = c0 & 7;
          oi = ((2 * (row_vox / p.Wo) + (sub >> 1)) * (size_t)(2 * p.Wo) + 2 * ow + (sub & 1)) * 8 + oc0;
        }
      }
      if (COUT % 4 == 0) {
        if (p.res_up) {  // + F.interpolate(top, scale_factor=2, bilinear, align_corners=False)[row, ow]   (backbone.py:60,62)
          const int Hh = p.Ho >> 1, Wh = p.Wo >> 1;
          const size_t rr = row_vox / p.Wo;             // (b*D + d)*Ho + oh
          const int oh = (int)(rr % p.Ho);
          const size_t img = rr / p.Ho;
          float sy = ((float)oh + 0.5f) * 0.5f - 0.5f; sy = sy < 0.f ? 0.f : sy;
          float sx = ((float)ow + 0.5f) * 0.5f - 0.5f; sx = sx < 0.f ? 0.f : sx;
          const int y0 = (int)sy, x0 = (int)sx;
          const int y1 = y0 + (y0 < Hh - 1), x1 = x0 + (x0 < Wh - 1);
          const float ly1 = sy - (float)y0, ly0 = 1.f - ly1, lx1 = sx - (float)x0, lx0 = 1.f - lx1;
          const float* base = p.res_up + img * Hh * Wh * COUT + c0;
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
        if (p.res) {
          const float4 rr = *reinterpret_cast<const float4*>(p.res + oi);
          o[0] = rr.x + o[0] * p.res_scale; o[1] = rr.y + o[1] * p.res_scale;
          o[2] = rr.z + o[2] * p.res_scale; o[3] = rr.w + o[3] * p.res_scale;
        }
        *reinterpret_cast<float4*>(p.y + oi) = make_float4(o[0], o[1], o[2], o[3]);
        if constexpr (ST != 0) stat_accum<C, ST>(p, lds_base_, oi, c0, o, ps, pq);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (c0 + k < COUT) p.y[oi + k] = p.res ? p.res[oi + k] + o[k] * p.res_scale : o[k];
        }
      }
    }
    if constexpr (ST != 0) stat_commit<C>(lds_base_, c0, n16, ps, pq);
  }
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// ---- Winograd form (Cfg::WINO) ---------------------------------------------------------------------------------
// Weight fragments (pack_weights_wino_kernel, conv3d.hip): [kd][chunk][ab = a*4+b][nt][lane][4], U = G g_kd G^T.
template <typename C, int COUT, int NKD, int ST = 0>
__device__ __forceinline__ void step_wino(const float* const (&planes)[NKD], __amdgpu_buffer_rsrc_t wres, int wvoff, const LdsConvParams& p,
                                          int b, int d, int d_lim, int h, int w0, int q, int n16, const float (&wfirst)[2][C::NT][C::KPL]) {
  constexpr int NCH = C::NCH, NTALL = C::NT, NT = C::NTP, S = C::S, PW = C::PW, KPL = C::KPL;
  extern __shared__ __attribute__((aligned(16))) float lds_base_[];
  const float* epi_tab = lds_base_ + C::EPI_OFF;
  // Cout > 32: two passes over K with 2 n-tiles each (16 accumulators x 4 n-tiles would be the whole register file); the
  // patch is re-read and re-transformed per pass, which costs ~15 % of a pass
#pragma unroll 1
  for (int pass = 0; pass < NTALL / NT; ++pass) {
  f32x4 acc[16][NT];
#pragma unroll
  for (int ab = 0; ab < 16; ++ab)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[ab][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NF = NKD * NCH * 16;        // (kd, chunk, ab) steps
#ifndef MDF_WG_AHEAD
#define MDF_WG_AHEAD 2
#endif
  constexpr int AHEAD = MDF_WG_AHEAD, NA = AHEAD + 1;
  float af[NA][NT][KPL];
  auto load_a = [&](int i, int buf) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) buf_load_to<KPL>(wres, wvoff, (i * NTALL + pass * NT + nt) * (64 * KPL * 4), af[buf][nt]);
  };
#pragma unroll
  for (int i = 0; i < AHEAD; ++i) {
    if (i < 2 && pass == 0 && C::CIN_ < 64) {   // (pass 0 starts with the kernel-resident fragments; not kept for 64 channels: registers)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int k = 0; k < KPL; ++k) af[i % NA][nt][k] = wfirst[i][nt][k];
    } else {
      load_a(i, i % NA);
    }
  }
  constexpr int NG_ = NKD * NCH;
  // one block per CU (32+ channels: the LDS image allows no second one) means one wave per SIMD and nobody to hide this
  // wave's LDS latency and transform arithmetic: software-pipeline them into the MFMA loop instead -- while group g
  // multiplies, the patch of group g+1 is read (one element per ab step), its row pass runs column by column as the reads
  // land, and its column pass runs row by row right before the four MFMA steps that need that row.
  // (64 channels: two passes x 128 accumulator registers leave no room for the second patch buffer; the 16-channel depth-pair
  // form keeps a ring of 4 planes = 94 KB, one block per CU as well)
  constexpr bool PIPE = (C::CIN_ >= 32 && C::CIN_ < 64) || (C::RD == 2 && C::CIN_ >= 16);
  constexpr int NV = KPL / 2;                                              // packed pairs of input channels per lane
  auto read_elem = [&](f32x2_t (&v)[16][NV], int kd, int ch, int e) {     // e = j*4 + i: column-major so a column completes every 4 reads
    const int i = e & 3, j = e >> 2;
    float t[KPL];
    lds_frag<KPL>(planes[kd] + ((ch * 4) * S + i * PW + j) * KPL, t);
#pragma unroll
    for (int c = 0; c < NV; ++c) v[i * 4 + j][c] = (f32x2_t){t[2 * c], t[2 * c + 1]};
  };
  auto row_pass = [&](f32x2_t (&v)[16][NV], int j) {                      // B^T d : over the patch rows, column j
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const f32x2_t d0 = v[j][c], d1 = v[4 + j][c], d2 = v[8 + j][c], d3 = v[12 + j][c];
      v[j][c] = d0 - d2; v[4 + j][c] = d1 + d2; v[8 + j][c] = d2 - d1; v[12 + j][c] = d1 - d3;
    }
  };
  auto col_pass = [&](f32x2_t (&v)[16][NV], int a) {                      // (B^T d) B : over the columns, row a
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const f32x2_t e0 = v[a * 4][c], e1 = v[a * 4 + 1][c], e2 = v[a * 4 + 2][c], e3 = v[a * 4 + 3][c];
      v[a * 4][c] = e0 - e2; v[a * 4 + 1][c] = e1 + e2; v[a * 4 + 2][c] = e2 - e1; v[a * 4 + 3][c] = e1 - e3;
    }
  };
  if constexpr (!PIPE) {
    static_for<0, NG_>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int kd = g / NCH, ch = g % NCH;
      f32x2_t v[16][NV];
#pragma unroll
      for (int e = 0; e < 16; ++e) read_elem(v, kd, ch, e);
#pragma unroll
      for (int j = 0; j < 4; ++j) row_pass(v, j);
#pragma unroll
      for (int a4 = 0; a4 < 4; ++a4) col_pass(v, a4);
      static_for<0, 16>([&](auto abc) {
        constexpr int ab = decltype(abc)::value;
        constexpr int i = g * 16 + ab;
        if constexpr (i + AHEAD < NF) load_a(i + AHEAD, (i + AHEAD) % NA);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sidx = 0; sidx < KPL; ++sidx)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[ab][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i % NA][nt][sidx], v[ab][sidx >> 1][sidx & 1], acc[ab][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  } else {
    f32x2_t vbuf[2][16][NV];
#pragma unroll
    for (int e = 0; e < 16; ++e) read_elem(vbuf[0], 0, 0, e);
#pragma unroll
    for (int j = 0; j < 4; ++j) row_pass(vbuf[0], j);
    static_for<0, NG_>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int cur = g & 1, nxt = cur ^ 1;
      constexpr bool more = (g + 1 < NG_);
      constexpr int nkd = more ? (g + 1) / NCH : 0, nch = more ? (g + 1) % NCH : 0;
      static_for<0, 16>([&](auto abc) {
        constexpr int ab = decltype(abc)::value;
        constexpr int i = g * 16 + ab;
        if constexpr (i + AHEAD < NF) load_a(i + AHEAD, (i + AHEAD) % NA);
        if constexpr (more) read_elem(vbuf[nxt], nkd, nch, ab);                       // next group's patch, one element per step
        if constexpr (more && ab >= 5 && (ab & 3) == 1) row_pass(vbuf[nxt], (ab >> 2) - 1);   // column j complete since step 4j+3
        if constexpr ((ab & 3) == 0) col_pass(vbuf[cur], ab >> 2);                    // row a of the current group, just in time
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sidx = 0; sidx < KPL; ++sidx)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[ab][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i % NA][nt][sidx], vbuf[cur][ab][sidx >> 1][sidx & 1], acc[ab][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (more) row_pass(vbuf[nxt], 3);                                     // last column (its reads were issued at steps 12..15)
    });
  }
  // epilogue: Y = A^T M A per cout, A^T = [1 1 1 0; 0 1 -1 -1]; outputs (h + pr, w0 + 2*n16 + r)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int row0 = (pass * NT + nt) * 16 + 4 * q;
    const int c0 = (C::RD == 2) ? (row0 & 7) : row0;        // depth-pair form: rows 8..15 are the 8 channels of plane d + 1
    const int dd = (C::RD == 2) ? d + (row0 >> 3) : d;
    if (C::RD == 1 && c0 >= COUT) continue;
    float y[2][2][4];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {          // two couts per packed instruction
      f32x2_t srow[2][4];
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const f32x2_t m0 = {acc[bb][nt][2 * k2], acc[bb][nt][2 * k2 + 1]}, m1 = {acc[4 + bb][nt][2 * k2], acc[4 + bb][nt][2 * k2 + 1]};
        const f32x2_t m2 = {acc[8 + bb][nt][2 * k2], acc[8 + bb][nt][2 * k2 + 1]}, m3 = {acc[12 + bb][nt][2 * k2], acc[12 + bb][nt][2 * k2 + 1]};
        srow[0][bb] = m0 + m1 + m2;
        srow[1][bb] = m1 - m2 - m3;
      }
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const f32x2_t y0 = srow[pr][0] + srow[pr][1] + srow[pr][2];
        const f32x2_t y1 = srow[pr][1] - srow[pr][2] - srow[pr][3];
        y[pr][0][2 * k2] = y0[0]; y[pr][0][2 * k2 + 1] = y0[1];
        y[pr][1][2 * k2] = y1[0]; y[pr][1][2 * k2 + 1] = y1[1];
      }
    }
    float al_l[4], be_l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      al_l[k] = epi_tab[c0 + k];
      be_l[k] = epi_tab[64 + c0 + k];
    }
    float ps[4] = {0.f, 0.f, 0.f, 0.f}, pq[4] = {0.f, 0.f, 0.f, 0.f};   // ST: this lane's sums over its 2x2 outputs
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      if (h + pr >= p.Ho || dd >= d_lim) continue;
      const size_t row_vox = (((size_t)b * p.D + dd) * p.Ho + (h + pr)) * p.Wo;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int ow = w0 + 2 * n16 + r;
        if (ow >= p.Wo) continue;
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          o[k] = y[pr][r][k] * al_l[k] + be_l[k];
          if (p.relu) o[k] = fmaxf(o[k], 0.f);
        }
        const size_t oi = (row_vox + ow) * COUT + c0;
        if (p.res) {
          const float4 rr = *reinterpret_cast<const float4*>(p.res + oi);
          o[0] = rr.x + o[0] * p.res_scale; o[1] = rr.y + o[1] * p.res_scale;
          o[2] = rr.z + o[2] * p.res_scale; o[3] = rr.w + o[3] * p.res_scale;
        }
        *reinterpret_cast<float4*>(p.y + oi) = make_float4(o[0], o[1], o[2], o[3]);
        if constexpr (ST != 0) stat_accum<C, ST>(p, lds_base_, oi, c0, o, ps, pq);
      }
    }
    if constexpr (ST != 0) stat_commit<C>(lds_base_, c0, n16, ps, pq);
  }
  }  // pass
}

template <int CIN, int CIN_MEM, int COUT, int KD, int KHW, int SHW, int MT, int RW = 1, int WG = 0, int ST = 0>
// Register budget: the unrolled, pipelined tap loop wants ~280 registers (one 64-bit address pair per weight tap), which
// leaves ONE wave per SIMD.  Capping at 256 (two resident blocks per CU) is worth 6-10 % for the single-n-tile kernels
// (A/B in one process, scripts/bench_conv3d.py); with 2+ n-tiles the cap makes hipcc spill, so those stay uncapped.
// (Tighter caps for the 2-D kernels were tried: they spill the MFMA-heavy ones and do not help the latency-bound ones.)
// (Winograd form with 32+ input channels: its 141 KB of LDS allow one block per CU anyway, so it may use the whole register file)
// (ST variants of the HBM-bound 2-D layers: the ~10 registers of the epilogue sums would cost an occupancy step -- 16->16 Winograd 26 -> 43 us,
// 3->8 40 -> 58 us by rocprof -- so those keep their residency and spill a few registers instead)
#ifndef MDF_WD8_BLOCKS
#define MDF_WD8_BLOCKS 2
#endif
__global__ __launch_bounds__(256, (WG == 2 && CIN == 8) ? MDF_WD8_BLOCKS : (ST != 0 && KD == 1 && WG == 1 && CIN == 16) ? 3 : (ST != 0 && KD == 1 && CIN == 4) ? 4 :
                                  ((COUT <= 16 && !(WG == 1 && CIN >= 32) && !(WG == 2 && CIN >= 16)) ? 2 : 1)) void conv_lds_kernel(const LdsConvParams p) {
  typedef Cfg<CIN, CIN_MEM, COUT, KD, KHW, SHW, MT, RW, WG> C;
  constexpr int KPL = C::KPL, NG = C::NG, S = C::S, PW = C::PW, PH = C::PH;
  typedef typename VecT<KPL>::type vec_t;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, n16 = lane & 15;
  const __amdgpu_buffer_rsrc_t wres = make_rsrc(p.wpack, (unsigned)(C::NSTEP * C::NT * 64 * KPL * 4));
  const int wvoff = lane * KPL * 4;   // bytes
  // lane-constant part of the B-fragment LDS address (floats)
  const int lane_lds = (q * S + wave * (C::WINO ? 2 : SHW) * PW + n16 * C::SW) * KPL;

  int* item_slot = reinterpret_cast<int*>(lds + C::RING * C::PLANE);
  if (tid < 128) {   // epilogue table (1 / 0 beyond COUT or without BN)
    const int c = tid & 63;
    lds[C::EPI_OFF + tid] = (tid < 64) ? ((c < COUT && p.alpha) ? p.alpha[c] : 1.f) : ((c < COUT && p.beta) ? p.beta[c] : 0.f);
  }
  // ST: the block's fp64 sums [2][64] (sent to memory with one atomic per channel when the block -- or, 2-D, its run of tiles
  // inside one BatchNorm group -- is done) and, mode 2, the producing layer's (a, b, mean, invstd) of the current group
  [[maybe_unused]] double* st_tab = reinterpret_cast<double*>(lds + C::STAT_OFF);
  // 2-D ST: the grid is groups x blocks-per-group and a block's run of tiles stays inside its BatchNorm group
  [[maybe_unused]] const int st_bpg = (ST != 0 && KD == 1) ? (int)gridDim.x / (p.stat_groups > 0 ? p.stat_groups : 1) : (int)gridDim.x;
  [[maybe_unused]] const int st_group = (ST != 0 && KD == 1) ? (int)blockIdx.x / st_bpg : 0;
  [[maybe_unused]] const int st_local = (ST != 0 && KD == 1) ? (int)blockIdx.x % st_bpg : (int)blockIdx.x;
  if constexpr (ST != 0) {
    if (tid < 128) st_tab[tid] = 0.0;
    if constexpr (ST == 2) {
      const int c = tid & 63;
      lds[C::SAUX_OFF + tid] = (c < COUT) ? p.stat_aux[(size_t)st_group * 4 * COUT + (tid >> 6) * COUT + c] : 0.f;
    }
  }
  __syncthreads();
  // per-lane constants for the whole kernel: epilogue scale/shift of the lane's 4 couts, and (small layers) all weights
  float al[C::NT][4], be[C::NT][4];
#pragma unroll
  for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = nt * 16 + 4 * q + k;
      const int c = (RW > 1) ? row % COUT : row;
      al[nt][k] = (C::EPI_REG && row < C::ROWS && p.alpha) ? p.alpha[c] : 1.f;
      be[nt][k] = (C::EPI_REG && row < C::ROWS && p.beta) ? p.beta[c] : 0.f;
    }
  float wfirst[2][C::NT][KPL];   // fragments 0 and 1 of the packed weights: what every step call starts with
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) {
      if constexpr (!C::WREG && !(C::WINO && CIN >= 64)) buf_load_to<KPL>(wres, wvoff, (i * C::NT + nt) * (64 * KPL * 4), wfirst[i][nt]);   // (past the end reads 0)
      else {
#pragma unroll
        for (int k = 0; k < KPL; ++k) wfirst[i][nt][k] = 0.f;
      }
    }
  float wr[C::WN][C::NT][KPL];
  if constexpr (C::WREG) {
#pragma unroll
    for (int i = 0; i < C::NSTEP; ++i)
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt)
        buf_load_to<KPL>(wres, wvoff, (i * C::NT + nt) * (64 * KPL * 4), wr[i][nt]);
  } else {
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
      for (int k = 0; k < KPL; ++k) wr[0][nt][k] = 0.f;
  }
#ifdef MDF_STAMPS
  unsigned long long t_sched = 0, t_pro = 0, t_comp = 0, t_fill = 0, n_items_done = 0, n_dsteps = 0;
  const unsigned long long t_begin_all = STAMP();
#endif
  for (int pass = 0;; ++pass) {
#ifdef MDF_STAMPS
    const unsigned long long ts0 = STAMP();
#endif
    int item = 0;
    if constexpr (KD == 1) {
      // 2-D tiles all cost the same: static contiguous partition, no scheduling atomics (one global counter serialises at
      // ~90 dequeues/us and dominated the small layers: 54 -> 40 us on the refine-size convs)
      if (pass > 0) break;
    } else {
      __syncthreads();  // previous item's readers are done with the ring (and with item_slot)
      if (tid == 0) *item_slot = (int)atomicAdd(&g_sched[2 * p.sched_slot], 1u);
      __syncthreads();
      item = *item_slot;
      if (item >= p.n_items) break;
    }
#ifdef MDF_STAMPS
    const unsigned long long ts1 = STAMP();
    t_sched += ts1 - ts0; ++n_items_done;
#endif
    if constexpr (KD == 1) {
      // ------------------------------------------------------------------ 2-D: one run of consecutive tiles, double-buffered
      int t_begin = (int)((long long)blockIdx.x * p.n_tiles / gridDim.x);
      int t_end = (int)((long long)(blockIdx.x + 1) * p.n_tiles / gridDim.x);
      if constexpr (ST != 0) {   // per-group partition (n_tiles is a multiple of the group count)
        const int tg = p.n_tiles / (p.stat_groups > 0 ? p.stat_groups : 1);
        t_begin = st_group * tg + (int)((long long)st_local * tg / st_bpg);
        t_end = st_group * tg + (int)((long long)(st_local + 1) * tg / st_bpg);
      }
      if (t_begin >= t_end) break;
      auto tile_origin = [&](int tl, int& tb, int& th0, int& tw0) {
        const int twi = tl % p.tiles_w;
        const int rest = tl / p.tiles_w;
        tb = rest / p.tiles_h;
        th0 = (rest % p.tiles_h) * C::TH;
        tw0 = twi * C::TWO;
      };
      // fill mapping as in the 3-D path: cin-group fastest, so consecutive lanes read consecutive 16-B pieces (coalesced)
      constexpr int GF2 = (NG < 4) ? NG : 4;
      auto split2 = [&](int idx, int& v, int& g) {
        if (CIN_MEM != CIN && p.planar_in) {   // NCHW image planes: pixel fastest (consecutive lanes = consecutive floats of a plane)
          g = idx / (PH * PW);
          v = idx - g * (PH * PW);
          return;
        }
        const int glo = idx % GF2, r = idx / GF2;
        v = r % (PH * PW);
        g = (r / (PH * PW)) * GF2 + glo;
      };
      auto load2 = [&](int idx, int tb, int th0, int tw0) -> vec_t {
        int g, v;
        split2(idx, v, g);
        const int row = v / PW, col = v - row * PW;
        const int ih = th0 * SHW - C::PAD + row, iw = tw0 * SHW - C::PAD + col;
        if (idx >= NG * PH * PW || ih < 0 || ih >= p.H || iw < 0 || iw >= p.W) return vec_zero<KPL>();
        const float* src = p.x + (((size_t)tb * p.H + ih) * p.W + iw) * CIN_MEM + g * KPL;
        if (CIN_MEM == CIN) return *reinterpret_cast<const vec_t*>(src);
        float o[KPL];
        if (p.planar_in) {   // NCHW input: channel planes; consecutive lanes (pixels) read consecutive floats
#pragma unroll
          for (int k = 0; k < KPL; ++k)
            o[k] = (g * KPL + k < CIN_MEM) ? p.x[(((size_t)tb * CIN_MEM + g * KPL + k) * p.H + ih) * p.W + iw] : 0.f;
        } else {
#pragma unroll
          for (int k = 0; k < KPL; ++k) o[k] = (g * KPL + k < CIN_MEM) ? src[k] : 0.f;
        }
        return *reinterpret_cast<vec_t*>(o);
      };
      auto store2 = [&](int idx, int slot, const vec_t& val) {
        if (idx < NG * PH * PW) {
          int g, v;
          split2(idx, v, g);
          *reinterpret_cast<vec_t*>(lds + slot * C::PLANE + (g * S + v) * KPL) = val;
        }
      };
      int tb, th0, tw0;
      tile_origin(t_begin, tb, th0, tw0);
#pragma unroll
      for (int k = 0; k < C::NFILL; ++k) store2(tid + k * 256, 0, load2(tid + k * 256, tb, th0, tw0));
      __syncthreads();
#ifdef MDF_STAMPS
      t_pro += STAMP() - ts1;
#endif
      for (int tl = t_begin; tl < t_end; ++tl) {
        const int slot = (C::RING == 1) ? 0 : ((tl - t_begin) & 1);
        const bool row_live2 = (th0 + wave * C::WROWS) < p.Ho;
        const int cols2 = (min(p.Wo - tw0, C::TWO) + RW - 1) / RW;   // live MFMA columns
        const int mt_live2 = row_live2 ? (cols2 + 15) / 16 : 0;
        const int cb = tb, ch0 = th0, cw0 = tw0;           // this tile's origin; (tb,th0,tw0) move on to the next one
        const bool more2 = tl + 1 < t_end;
        if (more2) tile_origin(tl + 1, tb, th0, tw0);
#ifdef MDF_STAMPS
        const unsigned long long tc0 = STAMP();
#endif
        // next tile's global loads in flight during this tile's MFMAs (once the weights live in registers the tile's
        // compute no longer hides anything else, and this wait was 45 % of the small-channel layers' time)
        vec_t pf[C::NFILL];
        if constexpr (C::EARLY2) {
          if (more2) {
#pragma unroll
            for (int k = 0; k < C::NFILL; ++k) pf[k] = load2(tid + k * 256, tb, th0, tw0);
          }
        }
        if (mt_live2 > 0) {
          const float* planes[1] = {lds + slot * C::PLANE + lane_lds};
          const size_t row_vox = ((size_t)cb * p.Ho + (ch0 + wave)) * p.Wo;
          if constexpr (C::WINO) {
            step_wino<C, COUT, 1, ST>(planes, wres, wvoff, p, cb, 0, 1, ch0 + 2 * wave, cw0, q, n16, wfirst);
          } else {
            switch (mt_live2) {
              case 1: step<C, KD, KHW, SHW, COUT, 1, ST>(planes, wres, wvoff, p, row_vox, cw0, q, n16, wr, al, be, wfirst); break;
              case 2: if (MT >= 2) step<C, KD, KHW, SHW, COUT, (MT >= 2 ? 2 : 1), ST>(planes, wres, wvoff, p, row_vox, cw0, q, n16, wr, al, be, wfirst); break;
              case 3: if (MT >= 3) step<C, KD, KHW, SHW, COUT, (MT >= 3 ? 3 : 1), ST>(planes, wres, wvoff, p, row_vox, cw0, q, n16, wr, al, be, wfirst); break;
              default: step<C, KD, KHW, SHW, COUT, MT, ST>(planes, wres, wvoff, p, row_vox, cw0, q, n16, wr, al, be, wfirst); break;
            }
          }
        }
#ifdef MDF_STAMPS
        const unsigned long long tc1 = STAMP();
        t_comp += tc1 - tc0; ++n_dsteps;
#endif
        if (more2) {
          // next tile -> the other slot.  Every wave passed the barrier before this tile's compute, so nobody still
          // reads that slot; one barrier (after the writes) per tile.
          if constexpr (!C::EARLY2) {
#pragma unroll
            for (int k = 0; k < C::NFILL; ++k) pf[k] = load2(tid + k * 256, tb, th0, tw0);
          }
          if constexpr (C::RING == 1) __syncthreads();   // single buffer: everybody is done reading this tile
#pragma unroll
          for (int k = 0; k < C::NFILL; ++k) store2(tid + k * 256, (C::RING == 1) ? 0 : (slot ^ 1), pf[k]);
          __syncthreads();
        }
#ifdef MDF_STAMPS
        t_fill += STAMP() - tc1;
#endif
      }
      continue;
    } else {
    int r = item;
    const int dc = r % p.dchunks; r /= p.dchunks;
    const int tw = r % p.tiles_w; r /= p.tiles_w;
    const int th = r % p.tiles_h;
    const int b = r / p.tiles_h;
    const int h0 = th * C::TH, w0 = tw * C::TWO;          // output tile origin
    const int ih0 = h0 * SHW - C::PAD, iw0 = w0 * SHW - C::PAD;  // input tile origin
    const int d0 = dc * p.dch, d1 = min(d0 + p.dch, p.D);
    const bool row_live = (h0 + wave * C::WROWS) < p.Ho;
    const int cols = (min(p.Wo - w0, C::TWO) + RW - 1) / RW;   // live MFMA columns
    const int mt_live = row_live ? (cols + 15) / 16 : 0;  // wave-uniform

    // ---- plane fill helpers -------------------------------------------------------------------------
    // fill mapping: cin-group fastest, so consecutive lanes read consecutive 16-B pieces of a voxel / of neighbouring
    // voxels (coalesced global loads; the in-kernel stamps showed 10K cycles per plane with the voxel-fastest mapping whose
    // lanes were 128 B apart).  The price is bank conflicts on the few LDS stores, which is far cheaper.
    constexpr int GF = (NG < 4) ? NG : 4;   // groups walked fastest: 64 contiguous bytes per voxel, <= 4-way store conflicts
    auto split = [](int idx, int& v, int& g) {
      const int glo = idx % GF, r = idx / GF;
      v = r % (PH * PW);
      g = (r / (PH * PW)) * GF + glo;
    };
    auto load_elem = [&](int idx, int dz) -> vec_t {
      int v, g;
      split(idx, v, g);
      const int row = v / PW, col = v - row * PW;
      const int ih = ih0 + row, iw = iw0 + col;
      if (idx >= NG * PH * PW || dz < 0 || dz >= p.D || ih < 0 || ih >= p.H || iw < 0 || iw >= p.W) return vec_zero<KPL>();
      const float* src = p.x + ((((size_t)b * p.D + dz) * p.H + ih) * p.W + iw) * CIN_MEM + g * KPL;
      if (CIN_MEM == CIN) return *reinterpret_cast<const vec_t*>(src);
      float o[KPL];
#pragma unroll
      for (int k = 0; k < KPL; ++k) o[k] = (g * KPL + k < CIN_MEM) ? src[k] : 0.f;
      return *reinterpret_cast<vec_t*>(o);
    };
    auto store_elem = [&](int idx, int slot, const vec_t& val) {
      if (idx < NG * PH * PW) {
        int v, g;
        split(idx, v, g);
        *reinterpret_cast<vec_t*>(lds + slot * C::PLANE + (g * S + v) * KPL) = val;
      }
    };
    constexpr int NPL = C::NPL, RD = C::RD;
    auto slot_of = [](int dz) { return (KD == 1) ? 0 : ((dz % C::RING) + C::RING) % C::RING; };

    // prologue: the planes of the first depth step, d0-PD .. d0-PD+NPL-1
    if constexpr (CIN <= 16 || (C::WINO && CIN >= 32)) {   // (the one-block-per-CU Winograd kernels have the registers for it)
      vec_t pro[NPL][C::NFILL];   // all planes in flight at once: one memory latency instead of NPL
#pragma unroll
      for (int j = 0; j < NPL; ++j)
#pragma unroll
        for (int k = 0; k < C::NFILL; ++k) pro[j][k] = load_elem(tid + k * 256, d0 - C::PD + j);
#pragma unroll
      for (int j = 0; j < NPL; ++j)
#pragma unroll
        for (int k = 0; k < C::NFILL; ++k) store_elem(tid + k * 256, slot_of(d0 - C::PD + j), pro[j][k]);
    } else {                     // CIN = 32: 84 more live registers make the kernel spill (measured), so plane by plane
#pragma unroll 1
      for (int dz = d0 - C::PD; dz < d0 - C::PD + NPL; ++dz) {
#pragma unroll
        for (int k = 0; k < C::NFILL; ++k) store_elem(tid + k * 256, slot_of(dz), load_elem(tid + k * 256, dz));
      }
    }
    __syncthreads();
#ifdef MDF_STAMPS
    t_pro += STAMP() - ts1;
#endif

    for (int d = d0; d < d1; d += RD) {
#ifdef MDF_STAMPS
      const unsigned long long tc0 = STAMP();
#endif
      const bool more = (KD > 1) && (d + RD < d1);
      // fetch the RD planes the next step needs (from d-PD+NPL on) into registers; consumed after this step's MFMAs
      vec_t pf[RD][C::NFILL];
      if (more && p.prefetch_early) {
#pragma unroll
        for (int j = 0; j < RD; ++j)
#pragma unroll
          for (int k = 0; k < C::NFILL; ++k) pf[j][k] = load_elem(tid + k * 256, d + NPL - C::PD + j);
      }
      if (mt_live > 0) {
        const float* planes[NPL];
#pragma unroll
        for (int kd = 0; kd < NPL; ++kd) planes[kd] = lds + slot_of(d + kd - C::PD) * C::PLANE + lane_lds;
        const size_t row_vox = (((size_t)b * p.D + d) * p.Ho + (h0 + wave)) * p.Wo;
        if constexpr (C::WINO) {
          step_wino<C, COUT, NPL, ST>(planes, wres, wvoff, p, b, d, d1, h0 + 2 * wave, w0, q, n16, wfirst);
        } else {
          switch (mt_live) {
            case 1: step<C, KD, KHW, SHW, COUT, 1, ST>(planes, wres, wvoff, p, row_vox, w0, q, n16, wr, al, be, wfirst); break;
            case 2: if (MT >= 2) step<C, KD, KHW, SHW, COUT, (MT >= 2 ? 2 : 1), ST>(planes, wres, wvoff, p, row_vox, w0, q, n16, wr, al, be, wfirst); break;
            case 3: if (MT >= 3) step<C, KD, KHW, SHW, COUT, (MT >= 3 ? 3 : 1), ST>(planes, wres, wvoff, p, row_vox, w0, q, n16, wr, al, be, wfirst); break;
            default: step<C, KD, KHW, SHW, COUT, MT, ST>(planes, wres, wvoff, p, row_vox, w0, q, n16, wr, al, be, wfirst); break;
          }
        }
      }
#ifdef MDF_STAMPS
      const unsigned long long tc1 = STAMP();
      t_comp += tc1 - tc0; ++n_dsteps;
#endif
      if (more && !p.prefetch_early) {
#pragma unroll
        for (int j = 0; j < RD; ++j)
#pragma unroll
          for (int k = 0; k < C::NFILL; ++k) pf[j][k] = load_elem(tid + k * 256, d + NPL - C::PD + j);
      }
      if (more) {
        __syncthreads();  // all waves finished reading planes d-PD .. d-PD+RD-1: their slots are free
#pragma unroll
        for (int j = 0; j < RD; ++j)
#pragma unroll
          for (int k = 0; k < C::NFILL; ++k) store_elem(tid + k * 256, slot_of(d + NPL - C::PD + j), pf[j][k]);
        __syncthreads();
      }
#ifdef MDF_STAMPS
      t_fill += STAMP() - tc1;
#endif
    }
    }  // 3-D path
  }
#ifdef MDF_STAMPS
  if (tid == 0) {
    atomicAdd(&g_stamps[0], t_sched); atomicAdd(&g_stamps[1], t_pro); atomicAdd(&g_stamps[2], t_comp);
    atomicAdd(&g_stamps[3], t_fill); atomicAdd(&g_stamps[4], STAMP() - t_begin_all); atomicAdd(&g_stamps[5], 1ull);
    atomicAdd(&g_stamps[6], n_items_done); atomicAdd(&g_stamps[7], n_dsteps);
  }
#endif
  if constexpr (ST != 0) {
    __syncthreads();
    mdf::conv_stat_send<COUT>(st_tab, p.stat_out + (size_t)st_group * 2 * COUT, (long long)(p.stat_groups > 0 ? p.stat_groups : 1) * 2 * COUT,
                              p.stat_slices, (unsigned)st_local);
  }
  if (KD > 1 && tid == 0) {
    const unsigned done = atomicAdd(&g_sched[2 * p.sched_slot + 1], 1u);
    if (done == gridDim.x - 1) {  // last block out: re-arm the counters for the next launch
      g_sched[2 * p.sched_slot] = 0u;
      g_sched[2 * p.sched_slot + 1] = 0u;
      __threadfence();
    }
  }
}

template <int CIN, int CIN_MEM, int COUT, int KD, int KHW, int SHW, int MT, int RW = 1, int WG = 0, int ST = 0>
int launch_lds(LdsConvParams& p, hipStream_t st) {
  typedef Cfg<CIN, CIN_MEM, COUT, KD, KHW, SHW, MT, RW, WG> C;
  constexpr size_t kLds = ST ? C::LDS_BYTES_ST : C::LDS_BYTES;
  p.tiles_h = (p.Ho + C::TH - 1) / C::TH;
  p.tiles_w = (p.Wo + C::TWO - 1) / C::TWO;
  const long long tiles = (long long)p.B * p.tiles_h * p.tiles_w;
  const int blocks_per_cu = (int)(160 * 1024 / kLds) < 1 ? 1 : (int)(160 * 1024 / kLds);
  const int max_grid = 256 * (blocks_per_cu > 4 ? 4 : blocks_per_cu);
  // 3-D: items = tile x depth chunk, handed out dynamically; aim for >= ~6 items per resident block (smooths the cheaper
  // partially-filled tile columns) while keeping >= 3 planes per chunk (prologue = KD-1 extra planes).
  // 2-D: one run of consecutive tiles per block (double-buffered inside the run), grid = min(tiles/2, resident blocks).
  if (KD > 1) {
    // Depth-chunk length: ~6 items per resident block (the dynamic queue then evens out the cheaper, partially filled tile
    // columns), at least 3 planes per chunk (an item's prologue fetches KD-1 extra planes), chunks of equal length -- a
    // 1-plane tail chunk (D = 4 -> 3 + 1) costs a whole prologue for a third of the work: 93 -> 77 us on 16->16 @4x296x400.
    // (A makespan model "rounds x (chunk + prologue)" was tried and is wrong here: two resident blocks share one MFMA pipe,
    // so fewer, longer items do not finish sooner.)
    long long ipb = (C::RD == 2 && blocks_per_cu == 1) ? 3 : 6;   // (one resident block and long pair steps: 179 -> 172 us on 16->8 @24x296x400)
    if (const char* e = getenv("MDF_CONV_ITEMS_PER_BLOCK")) { if (atoi(e) > 0) ipb = atoi(e); }   // dev A/B
    long long want = (ipb * max_grid + tiles - 1) / tiles;
    if (want < 1) want = 1;
    if (want > p.D / 3) want = p.D / 3;
    if (want < 1) want = 1;
    int dch = (int)((p.D + want - 1) / want);
    if (C::RD == 2 && (dch & 1)) ++dch;   // depth-pair form: whole pairs per chunk (an odd tail pair computes a plane it does not store)
    p.dch = dch;
    p.dchunks = (p.D + dch - 1) / dch;
    const long long items = tiles * p.dchunks;
    if (items > 0x7fffffff) return mdf::fail(MDF_EARG, "conv_lds: too many work items");
    p.n_items = (int)items;
  } else {
    if (p.D != 1) return mdf::fail(MDF_EARG, "2-D conv path needs D == 1");
    if (tiles > 0x7fffffff) return mdf::fail(MDF_EARG, "conv_lds: too many tiles");
    p.n_tiles = (int)tiles;
    p.tiles_per_item = 0;
    long long tpb = 2;   // small layers: residency (4 blocks/CU) beats long runs -- 30 -> 20 us on the refine-size convs (A/B)
    if (const char* e = getenv("MDF_CONV2D_TILES_PER_BLOCK")) { if (atoi(e) > 0) tpb = atoi(e); }   // dev A/B
    long long g = tiles / tpb;                   // >= tpb tiles per block
    if (g < 256) g = 256;
    if (g > max_grid) g = max_grid;
    if (g > tiles) g = tiles;
    p.n_items = (int)g;                          // = grid size (static partition)
    p.dch = 1; p.dchunks = 1;
  }
  // the dynamic-LDS attribute is a per-DEVICE property of the function: one flag per device (several GPUs in one
  // process, DataParallel-style).  Benign race: the call is idempotent.
  static bool attr_done_dev[64] = {};     // (one per template instantiation, ST included)
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  bool& attr_done = attr_done_dev[(dev_id >= 0 && dev_id < 64) ? dev_id : 0];
  if (!attr_done || dev_id >= 64) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_lds_kernel<CIN, CIN_MEM, COUT, KD, KHW, SHW, MT, RW, WG, ST>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
    if (e != hipSuccess) return mdf::fail(MDF_EHIP, "hipFuncSetAttribute(dynamic LDS %zu): %s", kLds, hipGetErrorString(e));
    attr_done = true;
  }
  int grid = max_grid;
  if (const char* g = getenv("MDF_CONV_GRID")) { if (atoi(g) > 0) grid = atoi(g); }   // dev: residency experiments
  if (grid > p.n_items) grid = p.n_items;
  if (ST) {
    if (KD == 1) {   // groups x blocks-per-group
      const int G = p.stat_groups > 0 ? p.stat_groups : 1;
      int bpg = grid / G;
      if (bpg < 1) bpg = 1;
      grid = G * bpg;
      p.stat_slices = mdf::conv_stat_slices(bpg, p.stat_slices);
    } else {
      p.stat_slices = mdf::conv_stat_slices(grid, p.stat_slices);
    }
  }
  if (getenv("MDF_CONV_DEBUG")) {
    int nb = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_lds_kernel<CIN, CIN_MEM, COUT, KD, KHW, SHW, MT, RW, WG, ST>, 256, kLds);
    hipFuncAttributes fa{};
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&conv_lds_kernel<CIN, CIN_MEM, COUT, KD, KHW, SHW, MT, RW, WG, ST>));
    fprintf(stderr, "[conv_lds<%d,%d,%d,%d,%d,%d,%d,rw%d>] LDS %zu B dyn + %zu static, regs %d, occupancy API: %d blocks/CU (%s), grid %d, items %d\n",
            CIN, CIN_MEM, COUT, KD, KHW, SHW, MT, RW, kLds, fa.sharedSizeBytes, fa.numRegs, nb, hipGetErrorString(e), grid, p.n_items);
  }
  hipLaunchKernelGGL((conv_lds_kernel<CIN, CIN_MEM, COUT, KD, KHW, SHW, MT, RW, WG, ST>), dim3(grid), dim3(256), kLds, st, p);
  return mdf::check_launch("conv_lds_kernel");
}

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

template <int CIN, int NW>
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
  for (int d = 0; d < p.D; ++d) {
    const float pv = lg[d * NTHR] / sum;
    pr[(size_t)d * hw] = pv;
    if (p.depth) dep.add(pv * (p.per_pixel ? p.hypos[((size_t)b * p.D + d) * hw + pix] : p.hypos[(size_t)b * p.D + d]));
  }
  if (p.depth) p.depth[(size_t)b * hw + pix] = dep.result();
}

template <int CIN, int NW>
int launch_prob_fused(ProbParams& p, hipStream_t st) {
  typedef ProbCfg<CIN, NW> C;
  const size_t kLds = ((size_t)2 * C::PLANE + (size_t)p.D * C::NTHR) * sizeof(float);
  if (kLds > 160 * 1024) return mdf::fail(MDF_EUNSUPPORTED, "fused prob head: D=%d does not fit the LDS (use the two-launch route)", p.D);
  p.tiles_h = (p.H + C::TH - 1) / C::TH;
  p.tiles_w = (p.W + C::TWO - 1) / C::TWO;
  const long long tiles = (long long)p.B * p.tiles_h * p.tiles_w;
  if (tiles > 0x7fffffff) return mdf::fail(MDF_EARG, "prob head: too many tiles");
  static bool attr_done_dev[64] = {};   // (the attribute is set to the device maximum once: the size depends on D)
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  bool& attr_done = attr_done_dev[(dev_id >= 0 && dev_id < 64) ? dev_id : 0];
  if (!attr_done || dev_id >= 64) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&prob_fused_kernel<CIN, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return mdf::fail(MDF_EHIP, "hipFuncSetAttribute(dynamic LDS): %s", hipGetErrorString(e));
    attr_done = true;
  }
  hipLaunchKernelGGL((prob_fused_kernel<CIN, NW>), dim3((unsigned)tiles), dim3(C::NTHR), kLds, st, p);
  return mdf::check_launch("prob_fused_kernel");
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

#ifdef MDF_STAMPS
extern "C" int mdf_debug_read_stamps(unsigned long long* out8, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8);
  if (e != hipSuccess) return -3;
  if (reset) {
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z);
  }
  return 0;
}
#endif

// Internal entry used by mdf_conv3d_fwd (stride-1 3x3x3) and mdf_conv2d_fwd.  Returns MDF_EUNSUPPORTED when the
// configuration has no LDS-kernel instantiation (caller falls back to conv v1 / reports the error).
#define LDS_CASE(ci, cim, co, kd, k, s, mt)                                                      \
  if (!stat && Cin == ci && Cin_mem == cim && Cout == co && KD == kd && KHW == k && stride == s) \
    return launch_lds<ci, cim, co, kd, k, s, mt>(p, (hipStream_t)stream);
// layers the training step runs raw with epilogue sums (forward statistics / backward reductions): both variants
#define LDS_CASE_T(ci, cim, co, kd, k, s, mt)                                                    \
  if (Cin == ci && Cin_mem == cim && Cout == co && KD == kd && KHW == k && stride == s)          \
    return !stat ? launch_lds<ci, cim, co, kd, k, s, mt>(p, (hipStream_t)stream) : (stat->mode == 1 ? launch_lds<ci, cim, co, kd, k, s, mt, 1, 0, 1>(p, (hipStream_t)stream) : launch_lds<ci, cim, co, kd, k, s, mt, 1, 0, 2>(p, (hipStream_t)stream));

// stream -> scheduler slot (launches on one stream are ordered, so they can share a slot; different streams must not).
// A slot stays bound to its stream handle until mdf_release_stream(stream) gives it back (long-lived processes that
// create and destroy streams); the table holds kSchedSlots live streams.
namespace {
std::mutex g_slot_mu;
void* g_slot_stream[kSchedSlots];
bool g_slot_used[kSchedSlots];
}  // namespace

static int sched_slot_of(void* stream) {
  std::lock_guard<std::mutex> lock(g_slot_mu);
  int free_slot = -1;
  for (int i = 0; i < kSchedSlots; ++i) {
    if (g_slot_used[i] && g_slot_stream[i] == stream) return i;
    if (!g_slot_used[i] && free_slot < 0) free_slot = i;
  }
  if (free_slot >= 0) { g_slot_used[free_slot] = true; g_slot_stream[free_slot] = stream; }
  return free_slot;
}

extern "C" int mdf_release_stream(void* stream) {
  std::lock_guard<std::mutex> lock(g_slot_mu);
  for (int i = 0; i < kSchedSlots; ++i)
    if (g_slot_used[i] && g_slot_stream[i] == stream) { g_slot_used[i] = false; g_slot_stream[i] = nullptr; return MDF_OK; }
  return MDF_OK;   // never used by a conv launch: nothing to release
}

// w-phase variants read the expanded packing that conv3d.hip appends after the plain one (mdf_conv_rw_of / pack functions)
#define LDS_CASE_RW(ci, cim, co, kd, k, s, mt, rw)                                               \
  if (!stat && use_rw && Cin == ci && Cin_mem == cim && Cout == co && KD == kd && KHW == k && stride == s && !res_up) { \
    p.wpack = wpack + (size_t)kd * k * k * ci * 16;                                              \
    return launch_lds<ci, cim, co, kd, k, s, mt, rw>(p, (hipStream_t)stream);                    \
  }
#define LDS_CASE_RW_T(ci, cim, co, kd, k, s, mt, rw)                                             \
  if (use_rw && Cin == ci && Cin_mem == cim && Cout == co && KD == kd && KHW == k && stride == s && !res_up) { \
    p.wpack = wpack + (size_t)kd * k * k * ci * 16;                                              \
    return !stat ? launch_lds<ci, cim, co, kd, k, s, mt, rw>(p, (hipStream_t)stream) : (stat->mode == 1 ? launch_lds<ci, cim, co, kd, k, s, mt, rw, 0, 1>(p, (hipStream_t)stream) : launch_lds<ci, cim, co, kd, k, s, mt, rw, 0, 2>(p, (hipStream_t)stream)); \
  }

// Winograd variants read the transform-domain weights appended behind the plain (and w-phase) packing
#define LDS_CASE_WG(ci, co)                                                                      \
  if (use_wg && KD == 3 && KHW == 3 && stride == 1 && Cin == ci && Cin_mem == ci && Cout == co && !res_up) { \
    p.wpack = wpack + (size_t)27 * ci * (((co + 15) / 16) * 16) + (co == 8 ? (size_t)36 * ci * 16 : 0); \
    return !stat ? launch_lds<ci, ci, co, 3, 3, 1, 1, 2, 1>(p, (hipStream_t)stream) : (stat->mode == 1 ? launch_lds<ci, ci, co, 3, 3, 1, 1, 2, 1, 1>(p, (hipStream_t)stream) : launch_lds<ci, ci, co, 3, 3, 1, 1, 2, 1, 2>(p, (hipStream_t)stream)); \
  }
#define LDS_CASE_WG2(ci, co)                                                                     \
  if (use_wg && KD == 1 && KHW == 3 && stride == 1 && Cin == ci && Cin_mem == ci && Cout == co && !res_up && !shuffle2) { \
    p.wpack = wpack + (size_t)9 * ci * (((co + 15) / 16) * 16);                                  \
    return !stat ? launch_lds<ci, ci, co, 1, 3, 1, 1, 2, 1>(p, (hipStream_t)stream) : (stat->mode == 1 ? launch_lds<ci, ci, co, 1, 3, 1, 1, 2, 1, 1>(p, (hipStream_t)stream) : launch_lds<ci, ci, co, 1, 3, 1, 1, 2, 1, 2>(p, (hipStream_t)stream)); \
  }

// depth-pair Winograd (3-D, Cout 8): its fragments follow the plain, w-phase and (Cin 16) Winograd ones
#define LDS_CASE_WD(ci)                                                                          \
  if (use_wd && KD == 3 && KHW == 3 && stride == 1 && Cin == ci && Cin_mem == ci && Cout == 8 && !res_up && D >= 2) { \
    p.wpack = wpack + (size_t)27 * ci * 16 + (size_t)36 * ci * 16 + (ci == 16 ? (size_t)3 * 16 * 64 * 4 : 0); \
    return !stat ? launch_lds<ci, ci, 8, 3, 3, 1, 1, 2, 2>(p, (hipStream_t)stream) : (stat->mode == 1 ? launch_lds<ci, ci, 8, 3, 3, 1, 1, 2, 2, 1>(p, (hipStream_t)stream) : launch_lds<ci, ci, 8, 3, 3, 1, 1, 2, 2, 2>(p, (hipStream_t)stream)); \
  }

int mdf_conv_lds_dispatch(const float* x, const float* wpack, const float* alpha, const float* beta, const float* res,
                          float res_scale, const float* res_up, float* y, int B, int D, int H, int W, int Cin, int Cin_mem, int Cout, int KD,
                          int KHW, int stride, int relu, void* stream, int planar_in, int shuffle2, const mdf::ConvStat* stat) {
  static const int wd_mask = [] { const char* e = getenv("MDF_CONV_WD"); return e ? atoi(e) : 3; }();   // dev A/B: bit 0 Cin 8, bit 1 Cin 16
  static const bool use_wg = [] { const char* e = getenv("MDF_CONV_WINOGRAD"); return e ? atoi(e) != 0 : true; }();   // dev A/B
  static const bool use_rw = [] { const char* e = getenv("MDF_CONV_RW"); return e ? atoi(e) != 0 : true; }();   // dev A/B
  LdsConvParams p{};
  p.x = x; p.wpack = wpack; p.alpha = alpha; p.beta = beta; p.res = res; p.res_scale = res_scale; p.res_up = res_up; p.y = y;
  p.B = B; p.D = D; p.H = H; p.W = W; p.relu = relu; p.planar_in = planar_in; p.shuffle2 = shuffle2;
  if (stat) {
    if (Cout % 4 != 0 || shuffle2 || res_up) return mdf::fail(MDF_EUNSUPPORTED, "epilogue sums: Cout %% 4 == 0, no pixel-shuffle / upsample-add");
    p.stat_mode = stat->mode; p.stat_y = stat->y; p.stat_aux = stat->aux; p.stat_out = stat->out; p.stat_slices = stat->nslices;
    p.stat_groups = (KD == 1 && stat->group_imgs > 0) ? B / stat->group_imgs : 1;
  }
  p.sched_slot = sched_slot_of(stream);
  if (p.sched_slot < 0) return mdf::fail(MDF_EUNSUPPORTED, "conv kernels support up to %d distinct HIP streams per process", kSchedSlots);
  {
    const char* e = getenv("MDF_CONV_PREFETCH_EARLY");   // A/B switch (dev): default = after the MFMA block
    p.prefetch_early = e ? atoi(e) : 0;
  }
  const int pad = (KHW - 1) / 2;
  p.Ho = (H + 2 * pad - KHW) / stride + 1;
  p.Wo = (W + 2 * pad - KHW) / stride + 1;
  // 3-D stride-1 layers with 8 output channels: depth-pair Winograd
  { const bool use_wd = use_wg && (wd_mask & 1); LDS_CASE_WD(8) }
  { const bool use_wd = use_wg && (wd_mask & 2); LDS_CASE_WD(16) }
  // 3-D stride-1 layers with 16 output channels: Winograd F(2x2,3x3) in (h,w)
  LDS_CASE_WG(16, 16) LDS_CASE_WG(32, 16) LDS_CASE_WG(32, 32) LDS_CASE_WG(16, 8)
  LDS_CASE_WG2(16, 16) LDS_CASE_WG2(32, 32) LDS_CASE_WG2(64, 64)
  // Cout < 16: w-phase form (RW output voxels per MFMA column)
  LDS_CASE_RW_T(16, 16, 8, 3, 3, 1, 2, 2) LDS_CASE_RW_T(8, 8, 8, 3, 3, 1, 2, 2)
  LDS_CASE_RW(16, 16, 4, 1, 3, 1, 1, 4) LDS_CASE_RW(8, 8, 4, 1, 3, 1, 1, 4)
  LDS_CASE_RW_T(8, 8, 8, 1, 3, 1, 2, 2) LDS_CASE_RW_T(4, 3, 8, 1, 3, 1, 2, 2) LDS_CASE_RW(4, 1, 8, 1, 3, 1, 2, 2)   // also the HBM-bound ones: 64-B stores, half the LDS reads
  // 3-D regulariser layers (stride 1)
  LDS_CASE(32, 32, 16, 3, 3, 1, 2) LDS_CASE(16, 16, 16, 3, 3, 1, 4) LDS_CASE(16, 16, 8, 3, 3, 1, 4) LDS_CASE(8, 8, 8, 3, 3, 1, 4)
  LDS_CASE(32, 32, 32, 3, 3, 1, 2)
  LDS_CASE(16, 16, 32, 3, 3, 1, 2) LDS_CASE(8, 8, 16, 3, 3, 1, 4)      // input gradients of the regularisers' first layers (training)
  // 2-D layers of the feature pyramid / refinement (KD = 1)
  LDS_CASE(4, 3, 8, 1, 3, 1, 4) LDS_CASE(8, 8, 8, 1, 3, 1, 4) LDS_CASE(16, 16, 16, 1, 3, 1, 4) LDS_CASE(32, 32, 32, 1, 3, 1, 2)
  LDS_CASE(64, 64, 64, 1, 3, 1, 1)
  LDS_CASE_T(8, 8, 16, 1, 5, 2, 2) LDS_CASE_T(16, 16, 32, 1, 5, 2, 2) LDS_CASE_T(32, 32, 64, 1, 5, 2, 1)
  LDS_CASE(16, 16, 32, 1, 1, 1, 4)      // composed FPN heads, training backward (B3^T)
  LDS_CASE(16, 16, 64, 1, 1, 1, 4) LDS_CASE(32, 32, 64, 1, 1, 1, 2) LDS_CASE(64, 64, 16, 1, 1, 1, 1) LDS_CASE(64, 64, 32, 1, 1, 1, 1)
  LDS_CASE(64, 64, 64, 1, 1, 1, 1)
  LDS_CASE(32, 32, 32, 1, 1, 1, 2) LDS_CASE(32, 32, 16, 1, 1, 1, 2) LDS_CASE(16, 16, 16, 1, 1, 1, 4)   // composed FPN heads
  LDS_CASE(4, 1, 8, 1, 3, 1, 4) LDS_CASE(8, 8, 32, 1, 3, 1, 4) LDS_CASE(8, 8, 1, 1, 3, 1, 4)
  LDS_CASE(32, 32, 8, 1, 3, 1, 2)                                       // input gradient of the refinement net's 8 -> 32 conv (training)
  LDS_CASE(16, 16, 32, 1, 3, 1, 2) LDS_CASE(32, 32, 64, 1, 3, 1, 1)   // input gradients of the k5-s2 layers (training): 3x3 over dy, 4 parity classes as channels
  LDS_CASE(16, 16, 4, 1, 3, 1, 4) LDS_CASE(8, 8, 4, 1, 3, 1, 4)   // prob head as per-plane partial sums (prob_head.hip)
  return MDF_EUNSUPPORTED;
}
