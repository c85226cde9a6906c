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

#include "conv_lds_common.h"

namespace {

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
#ifndef MDF_WG_AHEAD
#define MDF_WG_AHEAD 2
#endif
// The transform-domain MFMAs of the planes J0 .. J0+NKD-1 of a step's window (planes[] holds those), added to acc.
template <typename C, int J0, int NKD>
__device__ __forceinline__ void wino_accumulate(const float* const (&planes)[NKD], __amdgpu_buffer_rsrc_t wres, int wvoff, int pass,
                                                f32x4 (&acc)[16][C::NTP], const float (&wfirst)[2][C::NT][C::KPL]) {
  constexpr int NCH = C::NCH, NTALL = C::NT, NT = C::NTP, S = C::S, PW = C::PW, KPL = C::KPL;
  constexpr int NF = NKD * NCH * 16;        // (kd, chunk, ab) steps of this call; fragment J0*NCH*16 + i of the packed weights
  constexpr int AHEAD = MDF_WG_AHEAD, NA = AHEAD + 1;
  float af[NA][NT][KPL];
  auto load_a = [&](int i, int buf) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) buf_load_to<KPL>(wres, wvoff, ((J0 * NCH * 16 + i) * NTALL + pass * NT + nt) * (64 * KPL * 4), af[buf][nt]);
  };
#pragma unroll
  for (int i = 0; i < AHEAD; ++i) {
    if (J0 == 0 && i < 2 && pass == 0 && C::CIN_ < 64) {   // (pass 0 starts with the kernel-resident fragments; not kept for 64 channels: registers)
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
  constexpr bool PIPE = (C::CIN_ >= 32 && C::CIN_ < 64) || (C::RD == 2 && C::CIN_ >= 16 && !C::STREAM);
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
}

// Output transform and epilogue of the accumulators: Y = A^T M A per cout, A^T = [1 1 1 0; 0 1 -1 -1]; outputs (h + pr, w0 + 2*n16 + r)
template <typename C, int COUT, int ST>
__device__ __forceinline__ void wino_epilogue(const f32x4 (&acc)[16][C::NTP], int pass, const LdsConvParams& p, int b, int d, int d_lim, int h,
                                              int w0, int q, int n16) {
  constexpr int NT = C::NTP;
  extern __shared__ __attribute__((aligned(16))) float lds_base_[];
  const float* epi_tab = lds_base_ + C::EPI_OFF;
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
        size_t oi = (row_vox + ow) * COUT + c0;
        if constexpr (C::NPL == 1 && ((C::CIN_ == 16 && COUT == 32) || (C::CIN_ == 32 && COUT == 64))) {
          if (p.shuffle2) {   // input gradient of a k5-s2 layer (training): GEMM row = parity class * Cq + channel, stored at pixel (2h + py, 2w + px)
            constexpr int CQ = COUT / 4;
            const int sub = c0 / CQ, oc0 = c0 % CQ;
            oi = ((2 * (row_vox / p.Wo) + (sub >> 1)) * (size_t)(2 * p.Wo) + 2 * ow + (sub & 1)) * CQ + oc0;
          }
        }
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
}

// Weight fragments (pack_weights_wino_kernel, conv3d.hip): [kd][chunk][ab = a*4+b][nt][lane][4], U = G g_kd G^T.
template <typename C, int COUT, int NKD, int ST = 0>
__device__ __forceinline__ void step_wino(const float* const (&planes)[NKD], __amdgpu_buffer_rsrc_t wres, int wvoff, const LdsConvParams& p,
                                          int b, int d, int d_lim, int h, int w0, int q, int n16, const float (&wfirst)[2][C::NT][C::KPL]) {
  // Cout > 32: two passes over K with 2 n-tiles each (16 accumulators x 4 n-tiles would be the whole register file); the
  // patch is re-read and re-transformed per pass, which costs ~15 % of a pass
#pragma unroll 1
  for (int pass = 0; pass < C::NT / C::NTP; ++pass) {
    f32x4 acc[16][C::NTP];
#pragma unroll
    for (int ab = 0; ab < 16; ++ab)
#pragma unroll
      for (int nt = 0; nt < C::NTP; ++nt) acc[ab][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    wino_accumulate<C, 0, NKD>(planes, wres, wvoff, pass, acc, wfirst);
    wino_epilogue<C, COUT, ST>(acc, pass, p, b, d, d_lim, h, w0, q, n16);
  }
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
        if constexpr (CIN_MEM == CIN && NG % 4 == 0 && KPL == 4) {
          if (p.s2d) {   // parity images of a map at twice the resolution (p.H x p.W is that map): group g = parity * (NG/4) + 4-channel slice
            constexpr int GPP = NG / 4;
            const int par = g / GPP, c4 = g - par * GPP;
            const int ihf = 2 * ih + (par >> 1), iwf = 2 * iw + (par & 1);
            if (idx >= NG * PH * PW || ihf < 0 || ihf >= p.H || iwf < 0 || iwf >= p.W) return vec_zero<KPL>();
            return *reinterpret_cast<const vec_t*>(p.x + (((size_t)tb * p.H + ihf) * p.W + iwf) * (CIN / 4) + c4 * KPL);
          }
        }
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

    if constexpr (C::STREAM) {
      // ---- depth-pair form on a ring of three planes (Cfg::STREAM): per pair of output planes d, d+1
      //   planes d-1, d, d+1 resident -> their MFMAs (plane d+2 in flight) | barrier, d+2 -> slot of d-1, barrier |
      //   MFMAs of plane d+2, output transform, stores (plane d+3 in flight) | barrier, d+3 -> slot of d, barrier
      vec_t pro[KD][C::NFILL];
#pragma unroll
      for (int j = 0; j < KD; ++j)
#pragma unroll
        for (int k = 0; k < C::NFILL; ++k) pro[j][k] = load_elem(tid + k * 256, d0 - 1 + j);
#pragma unroll
      for (int j = 0; j < KD; ++j)
#pragma unroll
        for (int k = 0; k < C::NFILL; ++k) store_elem(tid + k * 256, slot_of(d0 - 1 + j), pro[j][k]);
      __syncthreads();
      for (int d = d0; d < d1; d += 2) {
        vec_t pf[C::NFILL];
#pragma unroll
        for (int k = 0; k < C::NFILL; ++k) pf[k] = load_elem(tid + k * 256, d + 2);
        f32x4 acc[16][C::NTP];
#pragma unroll
        for (int ab = 0; ab < 16; ++ab) acc[ab][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (mt_live > 0) {
          const float* pa[3];
#pragma unroll
          for (int j = 0; j < 3; ++j) pa[j] = lds + slot_of(d - 1 + j) * C::PLANE + lane_lds;
          wino_accumulate<C, 0, 3>(pa, wres, wvoff, 0, acc, wfirst);
        }
        __syncthreads();   // every wave is past plane d-1
#pragma unroll
        for (int k = 0; k < C::NFILL; ++k) store_elem(tid + k * 256, slot_of(d + 2), pf[k]);
        __syncthreads();
        const bool more = d + 2 < d1;
        if (more) {
#pragma unroll
          for (int k = 0; k < C::NFILL; ++k) pf[k] = load_elem(tid + k * 256, d + 3);
        }
        if (mt_live > 0) {
          const float* pb[1] = {lds + slot_of(d + 2) * C::PLANE + lane_lds};
          wino_accumulate<C, 3, 1>(pb, wres, wvoff, 0, acc, wfirst);
          wino_epilogue<C, COUT, ST>(acc, 0, p, b, d, d1, h0 + 2 * wave, w0, q, n16);
        }
        if (more) {
          __syncthreads();   // every wave is past plane d (read in the first half only, but it shares the barrier)
#pragma unroll
          for (int k = 0; k < C::NFILL; ++k) store_elem(tid + k * 256, slot_of(d + 3), pf[k]);
          __syncthreads();
        }
      }
      continue;
    }
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
    // (one resident block and long pair steps: 179 -> 172 us on 16->8 @24x296x400 with 3; the three-plane ring form of that layer,
    //  two resident blocks: 163 us with 6, 158 with 4, 151 with 2)
    long long ipb = C::STREAM ? 2 : ((C::RD == 2 && blocks_per_cu == 1) ? 3 : 6);
    if (const char* e = getenv("MDF_CONV_ITEMS_PER_BLOCK")) { if (atoi(e) > 0) ipb = atoi(e); }   // dev A/B
    long long want = (ipb * max_grid + tiles - 1) / tiles;
    if (want < 1) want = 1;
    if (want > p.D / 3) want = p.D / 3;
    if (want < 1) want = 1;
    int dch = (int)((p.D + want - 1) / want);
    if (C::RD == 2 && (dch & 1)) ++dch;   // depth-pair form: whole pairs per chunk (an odd tail pair computes a plane it does not store)
    // Between one and two rounds of resident blocks the second round runs mostly empty (32->32 @24x74x100: 40 tiles x 8 chunks = 320
    // items on 256 blocks took 98 us; x 6 chunks = 240 items: 74 us; 16->16 @12x148x200: 532 items on 512 blocks 50 us, 399 items 45 us):
    // lengthen the chunks until one round holds everything, if that round is shorter than the two it replaces.
    {
      const long long items0 = tiles * ((p.D + dch - 1) / dch);
      // (also from the third round: 16->8 @24x296x400, 481 tiles x 3 chunks on 512 blocks 152 us, one chunk 144 us; not beyond -- with four
      //  or more rounds the dynamic queue fills the tail, and long chunks lose: 16->16 @48x148x200 130 -> 154 us)
      if (items0 > max_grid && items0 < 3 * (long long)max_grid) {
        const long long rounds0 = (items0 + max_grid - 1) / max_grid;
        for (int d2 = dch + C::RD; d2 <= p.D + C::RD - 1; d2 += C::RD) {
          if (tiles * ((p.D + d2 - 1) / d2) <= max_grid) {
            if (10 * (d2 + C::NPL - 1) < 9 * rounds0 * (dch + C::NPL - 1)) dch = d2;
            break;
          }
        }
      }
    }
    if (const char* e = getenv("MDF_CONV_DCH")) { if (atoi(e) > 0) dch = atoi(e); }   // dev A/B
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

}  // namespace

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
  if (use_wg && KD == 1 && KHW == 3 && stride == 1 && Cin == ci && Cin_mem == ci && Cout == co && !res_up && (!shuffle2 || (ci == 16 && co == 32) || (ci == 32 && co == 64))) { \
    p.wpack = wpack + (size_t)9 * ci * (((co + 15) / 16) * 16);                                  \
    return !stat ? launch_lds<ci, ci, co, 1, 3, 1, 1, 2, 1>(p, (hipStream_t)stream) : (stat->mode == 1 ? launch_lds<ci, ci, co, 1, 3, 1, 1, 2, 1, 1>(p, (hipStream_t)stream) : launch_lds<ci, ci, co, 1, 3, 1, 1, 2, 1, 2>(p, (hipStream_t)stream)); \
  }

// depth-pair Winograd (3-D, Cout 8): its fragments follow the plain, w-phase and (Cin 16) Winograd ones
#define LDS_CASE_WD(ci)                                                                          \
  if (use_wd && KD == 3 && KHW == 3 && stride == 1 && Cin == ci && Cin_mem == ci && Cout == 8 && !res_up && D >= 2) { \
    p.wpack = wpack + (size_t)27 * ci * 16 + (size_t)36 * ci * 16 + (ci == 16 ? (size_t)3 * 16 * 64 * 4 : 0); \
    if (ci == 16 && wd_stream && !stat) return launch_lds<ci, ci, 8, 3, 3, 1, 1, 2, 3>(p, (hipStream_t)stream); \
    return !stat ? launch_lds<ci, ci, 8, 3, 3, 1, 1, 2, 2>(p, (hipStream_t)stream) : (stat->mode == 1 ? launch_lds<ci, ci, 8, 3, 3, 1, 1, 2, 2, 1>(p, (hipStream_t)stream) : launch_lds<ci, ci, 8, 3, 3, 1, 1, 2, 2, 2>(p, (hipStream_t)stream)); \
  }

int mdf_wino3d_dispatch(const float* x, const float* wpack_wino, const float* alpha, const float* beta, const float* res, float res_scale,
                        float* y, int B, int D, int H, int W, int Cin, int Cout, int relu, void* stream);   // wino3d.hip
int mdf_wino2d_dispatch(const float* x, const float* wpack_wino, const float* alpha, const float* beta, const float* res, float res_scale,
                        float* y, int B, int H, int W, int Cin, int Cout, int relu, void* stream);
int mdf_wino2d_s2d_dispatch(const float* x, const float* wpack_k5w, const float* alpha, const float* beta, float* y, int B, int Ho, int Wo,
                            int Cin_mem, int Cout, int relu, void* stream);          // wino2d.hip

int mdf_conv_lds_dispatch(const float* x, const float* wpack, const float* alpha, const float* beta, const float* res,
                          float res_scale, const float* res_up, float* y, int B, int D, int H, int W, int Cin, int Cin_mem, int Cout, int KD,
                          int KHW, int stride, int relu, void* stream, int planar_in, int shuffle2, const mdf::ConvStat* stat) {
  static const int wd_mask = [] { const char* e = getenv("MDF_CONV_WD"); return e ? atoi(e) : 3; }();   // dev A/B: bit 0 Cin 8, bit 1 Cin 16
  const bool wd_stream = [] { const char* e = getenv("MDF_CONV_WD_STREAM"); return e ? atoi(e) != 0 : true; }();   // dev A/B (read per call): ring of 3 planes
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
  // The 2-D k5 s2 layer 16 -> 32 as a Winograd 3x3 conv over the four parity images of its input (64 -> 32: 16 instead of 25 MFMA groups per
  // input channel; 163 -> 141 us); the transform-domain fragments follow the 25 plain taps (conv3d.hip: k5w_built).  (8 -> 16 the same way,
  // 32 -> 16 over the parity images, measured 200 against 203 us -- that layer waits on HBM, and one block per CU does not help it; 32 -> 64
  // would need 128 input channels in LDS.)
  {
    const bool k5w = [] { const char* e = getenv("MDF_CONV_K5_WINOGRAD"); return e ? atoi(e) != 0 : true; }();   // dev A/B (read per call)
    if (use_wg && k5w && KD == 1 && KHW == 5 && stride == 2 && !stat && !res_up && !shuffle2 && !planar_in && H % 2 == 0 && W % 2 == 0) {
      // (wino2d.hip's S2D form first: pinned accumulators, one transform cluster per chunk -- 16 -> 32 and, there, 8 -> 16 too)
      const bool use_w2 = [] { const char* e = getenv("MDF_CONV_WINO2D"); return e ? atoi(e) != 0 : true; }();
      if (use_w2 && !res && Cin == Cin_mem && ((Cin == 16 && Cout == 32) || (Cin == 8 && Cout == 16))) {
        const size_t plain = (size_t)25 * 64 * (Cin == 16 ? 2 * 4 : 1 * 2);      // 25 taps x NCH 1 x NT x 64 lanes x KPL floats
        const int rc = mdf_wino2d_s2d_dispatch(x, wpack + plain, alpha, beta, y, B, p.Ho, p.Wo, Cin, Cout, relu, stream);
        if (rc != MDF_EUNSUPPORTED) return rc;
      }
      if (Cin == 16 && Cin_mem == 16 && Cout == 32) {
        p.s2d = 1; p.wpack = wpack + (size_t)25 * 1 * 2 * 64 * 4;
        return launch_lds<64, 64, 32, 1, 3, 1, 1, 2, 1>(p, (hipStream_t)stream);
      }
    }
  }
  if (shuffle2 && Cout == 64 && !use_wg) return mdf::fail(MDF_EUNSUPPORTED, "pixel-shuffle store for Cout = 64 is built into the Winograd form only (MDF_CONV_WINOGRAD=0 is set)");
  // 3-D stride-1 layers with 8 output channels: depth-pair Winograd
  { const bool use_wd = use_wg && (wd_mask & 1); LDS_CASE_WD(8) }
  { const bool use_wd = use_wg && (wd_mask & 2); LDS_CASE_WD(16) }
  // 3-D stride-1 layers with 16 output channels, eval: input-stationary Winograd (wino3d.hip); same fragments as the form below
  {
    const bool use_w3 = [] { const char* e = getenv("MDF_CONV_WINO3D"); return e ? atoi(e) != 0 : true; }();   // dev A/B and the equality test (read per call)
    if (use_wg && use_w3 && !stat && KD == 3 && KHW == 3 && stride == 1 && Cin == Cin_mem && !res_up && !shuffle2 && Cout % 16 == 0) {
      const int rc = mdf_wino3d_dispatch(x, wpack + (size_t)27 * Cin * Cout, alpha, beta, res, res_scale, y, B, D, H, W, Cin, Cout, relu, stream);
      if (rc != MDF_EUNSUPPORTED) return rc;
    }
  }
  // 2-D 3x3 stride-1 layers of the feature pyramid, eval: wino2d.hip (same fragments as LDS_CASE_WG2 below)
  {
    const bool use_w2 = [] { const char* e = getenv("MDF_CONV_WINO2D"); return e ? atoi(e) != 0 : true; }();   // dev A/B and the equality test (read per call)
    if (use_wg && use_w2 && !stat && KD == 1 && KHW == 3 && stride == 1 && Cin == Cin_mem && !res_up && !shuffle2 && !planar_in && D == 1) {
      const int rc = mdf_wino2d_dispatch(x, wpack + (size_t)9 * Cin * (((Cout + 15) / 16) * 16), alpha, beta, res, res_scale, y, B, H, W, Cin, Cout, relu, stream);
      if (rc != MDF_EUNSUPPORTED) return rc;
    }
  }
  // 3-D stride-1 layers with 16 output channels: Winograd F(2x2,3x3) in (h,w)
  LDS_CASE_WG(16, 16) LDS_CASE_WG(32, 16) LDS_CASE_WG(32, 32) LDS_CASE_WG(16, 8) LDS_CASE_WG(16, 32)
  LDS_CASE_WG2(16, 16) LDS_CASE_WG2(32, 32) LDS_CASE_WG2(64, 64) LDS_CASE_WG2(16, 32) LDS_CASE_WG2(32, 64)
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
