// Input-stationary Winograd F(2x2,3x3) for the large stride-1 3x3x3 layers of the regularisers (Cout 16 / 32; regular.py:17-43,80-110).
//
// What bounds an fp32-MFMA kernel on gfx950 (scripts/micro/gen_mfma_issue.py, profiles/r05_mfma_issue.md): v_mfma_f32_16x16x4_f32
// runs at the rate of the vector ALUs and does NOT overlap with them -- every vector instruction placed between two MFMAs costs
// 4-12 cycles of matrix-pipe time, a second wave on the SIMD recovers at most half of that, and LDS reads / buffer loads are free.
// So the lever is the number of vector instructions per MFMA.  conv_lds.hip's Winograd form (output-stationary: per output plane
// the three input planes d-1, d, d+1 are read from LDS and transformed) spends one packed add per MFMA on input transforms, plus
// an item prologue per 4-plane depth chunk.  Here a block walks the depth axis of its tile ONCE:
//   * an input plane is read and transformed once and multiplied with all three kd slices of the weights into the accumulators of
//     the output planes z+1, z, z-1 (three sets of 16 transform-domain accumulators, rotating): 1/3 of the transform work, one
//     LDS plane read per plane instead of three; each transform is ONE cluster of 64 packed adds at a chunk boundary;
//   * output plane z-1 is complete after input plane z: output transform + epilogue, then its set starts over (first MFMA with C = 0);
//   * the work is cut stream-K style: block b owns steps [b T / G, (b+1) T / G) of the T = tiles x D (tile, plane) steps in
//     depth-fastest order -- one round, equal length, at most two tile changes per block, no scheduling atomics;
//   * plane fill: the thread's global offsets and LDS addresses are lane constants of a tile (buffer loads with an out-of-range
//     offset for halo voxels, zero planes through num_records = 0): no address arithmetic per plane.
// Per accumulator the MFMA order is that of conv_lds.hip's form (kd, cin chunk, k), so the two kernels agree bit for bit.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "conv_lds_common.h"
#include "wino3d_acc.h"

namespace {

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

struct Wino3dParams {
  const float* x;      // [B,D,H,W,CIN]
  const float* wpack;  // transform-domain fragments [kd][chunk][ab][nt][lane][4] (conv3d.hip: pack_wino_elem)
  const float* alpha;  // [COUT] or null
  const float* beta;   // [COUT] or null
  const float* res;    // [B,D,H,W,COUT] or null
  float* y;            // [B,D,H,W,COUT]
  float res_scale;
  int B, D, H, W, relu;
  int tiles_h, tiles_w;
  long long total;     // B * tiles_h * tiles_w * D steps (< 2^31)
};

template <int CIN, int COUT>
struct W3 {
  static_assert(CIN % 16 == 0 && COUT % 16 == 0 && COUT <= 64, "Winograd 3-D form: Cin, Cout multiples of 16");
  static constexpr int NCH = CIN / 16, NT = COUT / 16, NG = CIN / 4;
  static constexpr int WM = 4 / NT;                 // waves along the tile's row pairs (the others along the cout tiles)
  static constexpr int TH = 2 * WM, TWO = 32;       // output tile
  static constexpr int PH = TH + 2, PW = TWO + 2;   // input patch
  static constexpr int NPP = PH * PW;
  static constexpr int S = round_s(NPP, 4, 2);      // vectors per cin group (conv_lds_common.h: bank-conflict-free for the stride-2 patch reads)
  static constexpr int PLANE = CIN * S;             // floats
  static constexpr int RING = 3;
  static constexpr int NPOS = NG * NPP;
  static constexpr int NFILL = (NPOS + 255) / 256;
  static constexpr int EPI_OFF = RING * PLANE;      // alpha[64], beta[64]
  // the weight fragments of kd slices 0 .. WL-1 stay in LDS for the whole kernel where they fit beside the planes (16 input channels:
  // 48 KB): as buffer loads the fragments are 88 % of the vector-L1 accesses of the kernel (52 M lines per launch of the 32 -> 16
  // layer, TCP_TOTAL_CACHE_ACCESSES) and keep that path ~half busy, which costs 9-12 % of the run time (ablation, profiles/r05_wino3d.md)
  static constexpr int FRAG = 64 * 4;               // floats per fragment
  static constexpr int WL = ((size_t)(RING * PLANE + 128 + 3 * NCH * 16 * NT * FRAG) * sizeof(float) <= 160 * 1024) ? 3 : 0;
  static constexpr int W_OFF = EPI_OFF + 128;
  static constexpr size_t LDS_BYTES = (size_t)(RING * PLANE + 128 + WL * NCH * 16 * NT * FRAG) * sizeof(float);
  static_assert(S > NPP, "a pad vector per group takes the fill's surplus lanes");
};

#ifndef MDF_W3_DIAG
#define MDF_W3_DIAG 0        // dev ablations (wrong results): 1 no plane traffic, 2 no output stores, 4 no epilogue, 8 no transforms in the steps, 16 no fragment loads in the steps
#endif
#ifndef MDF_W3_NOMASK
#define MDF_W3_NOMASK 0      // dev: 1 = every kd slice of every step is multiplied (no masked step bodies)
#endif
#ifndef MDF_W3_PF_AT_START
#define MDF_W3_PF_AT_START 0  // dev: 1 = the next plane is requested at the step's start (first version)
#endif
#ifndef MDF_W3_NA
#define MDF_W3_NA 4          // weight-fragment ring: the fragments live in a[0:47] (the "a" operands of wino3d_acc.h) and run NA - 1 ab-steps ahead
#endif
#ifndef MDF_W3_WRITE_AB
#define MDF_W3_WRITE_AB 2
#endif
template <int CIN, int COUT, int NA, int WRITE_AB>
__global__ __launch_bounds__(256, 1) void wino3d_kernel(const Wino3dParams p) {
  typedef W3<CIN, COUT> C;
  constexpr int NCH = C::NCH, NT = C::NT, S = C::S, PW = C::PW, NPP = C::NPP, NFILL = C::NFILL;
  constexpr int AH = NA - 1;                         // weight fragments run AH ab-steps ahead of their MFMAs
  constexpr int NSTEP = NCH * 16;                    // ab-steps per plane
  static_assert(NSTEP % NA == 0, "the fragment ring must close over a plane");
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, n16 = lane & 15;
  const int wm = wave / NT, wn = wave % NT;
  const __amdgpu_buffer_rsrc_t wres = make_rsrc(p.wpack, (unsigned)(3 * NCH * 16 * NT * 64 * 4 * 4));
  const int wvoff = lane * 16;
  // lane-constant part of the patch-read address (floats): cin group q of the chunk, rows 2 wm .., columns 2 n16 ..
  const int lane_lds = (q * S + wm * 2 * PW + n16 * 2) * 4;

  if (tid < 128) {
    const int c = tid & 63;
    lds[C::EPI_OFF + tid] = (tid < 64) ? ((c < COUT && p.alpha) ? p.alpha[c] : 1.f) : ((c < COUT && p.beta) ? p.beta[c] : 0.f);
  }

  if constexpr (C::WL > 0) {      // resident weight fragments: [kd < WL][ab-step][nt][lane][4], as in memory
    constexpr int NV = C::WL * NSTEP * NT * 64;   // float4s
    for (int v = tid; v < NV; v += 256)
      *reinterpret_cast<float4*>(lds + C::W_OFF + v * 4) = *reinterpret_cast<const float4*>(p.wpack + (size_t)v * 4);
  }

  // ---- fill constants of the thread: element k is vector v (row, col of the patch) of cin group g; 4 groups fastest so that
  // consecutive lanes read 64 contiguous bytes of a voxel (coalescing; conv_lds.hip).  Surplus lanes write zeros to group 0's pad vector.
  int f_lds[NFILL];
#pragma unroll
  for (int k = 0; k < NFILL; ++k) {
    const int idx = tid + k * 256;
    const int glo = idx & 3, r = idx >> 2;
    const int v = r % NPP, ghi = r / NPP;
    f_lds[k] = (idx < C::NPOS) ? ((ghi * 4 + glo) * S + v) * 4 : NPP * 4;   // floats
  }

  const size_t plane_elems = (size_t)p.H * p.W * CIN;
  const unsigned plane_bytes = (unsigned)(plane_elems * 4);

  // ---- stream-K range of this block (depth fastest)
  // (32-bit step counts: the launcher refuses volumes with 2^31 steps; every segment scalar goes through readfirstlane so that
  //  the buffer descriptors built from them are provably wave-uniform -- otherwise hipcc wraps each buffer load in a waterfall loop)
  const unsigned lb = mdf::xcd_remap(blockIdx.x, gridDim.x);
  unsigned s_cur = (unsigned)((unsigned long long)lb * (unsigned)p.total / gridDim.x);
  const unsigned s_end = (unsigned)((unsigned long long)(lb + 1) * (unsigned)p.total / gridDim.x);
  // the 3 x 16 accumulator tiles are pinned to a[64:255] (wino3d_acc.h); the compiler never sees them

  while (s_cur < s_end) {
    const int tile_l = __builtin_amdgcn_readfirstlane((int)(s_cur / (unsigned)p.D));
    const int da = __builtin_amdgcn_readfirstlane((int)(s_cur - (unsigned)tile_l * (unsigned)p.D));
    const int db = __builtin_amdgcn_readfirstlane((int)min((unsigned)p.D, (unsigned)da + (s_end - s_cur)));
    s_cur += db - da;
    const int tw = __builtin_amdgcn_readfirstlane(tile_l % p.tiles_w);
    const int th = __builtin_amdgcn_readfirstlane((tile_l / p.tiles_w) % p.tiles_h);
    const int b = __builtin_amdgcn_readfirstlane(tile_l / (p.tiles_w * p.tiles_h));
    const int h0 = th * C::TH, w0 = tw * C::TWO;
    const bool wave_live = (h0 + 2 * wm) < p.H;      // (wave-uniform) rows of this wave inside the volume

    // per-tile global offsets (bytes inside a plane), 0x80000000 = reads as zero
    unsigned f_off[NFILL];
    {
      int tid_l = tid;
      asm volatile("" : "+v"(tid_l));               // (recomputed per tile: hoisted out of the loop the rows / columns / groups would hold 3 NFILL registers)
#pragma unroll
      for (int k = 0; k < NFILL; ++k) {
        const int idx = tid_l + k * 256;
        const int glo = idx & 3, r2 = idx >> 2;
        const int v = r2 % NPP, g = (r2 / NPP) * 4 + glo;
        const int ih = h0 - 1 + v / PW, iw = w0 - 1 + v % PW;
        const bool ok = idx < C::NPOS && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
        f_off[k] = ok ? (unsigned)(((ih * p.W + iw) * CIN + g * 4) * 4) : 0x80000000u;
      }
    }
    const float* xb = p.x + (size_t)b * p.D * plane_elems;
    float4 pf[NFILL];
    auto issue_plane = [&](int z) {                 // global loads of input plane z -> pf (zeros outside the volume)
      const bool in = z >= 0 && z < p.D;
      const float* pz = xb + (size_t)(in ? z : 0) * plane_elems;
      const unsigned long long pa = (unsigned long long)pz;
      const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pa), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(pa >> 32));
      const unsigned nrec = (MDF_W3_DIAG & 1) ? 0u : (unsigned)__builtin_amdgcn_readfirstlane((int)(in ? plane_bytes : 0u));
      const __amdgpu_buffer_rsrc_t xr = make_rsrc((const void*)(((unsigned long long)hi << 32) | lo), nrec);
#pragma unroll
      for (int k = 0; k < NFILL; ++k) {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)f_off[k], 0, 0);
        pf[k] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
    };
    auto write_plane = [&](int slot) {
#pragma unroll
      for (int k = 0; k < NFILL; ++k) *reinterpret_cast<float4*>(lds + slot * C::PLANE + f_lds[k]) = pf[k];
    };

    f32x2_t V[16][2];     // transformed patch of the chunk being multiplied: V[ab][cin pair]
    f32x2_t Dn[16][2];    // the next chunk's patch as it arrives from LDS (element e = i*4 + j)
    auto read_elem = [&](int slot, int ch, int e) {
      const int i = e >> 2, j = e & 3;
      float t[4];
      lds_frag<4>(lds + slot * C::PLANE + lane_lds + ((ch * 4) * S + i * PW + j) * 4, t);
      Dn[e][0] = (f32x2_t){t[0], t[1]};
      Dn[e][1] = (f32x2_t){t[2], t[3]};
    };
    auto transform = [&]() {                        // V = B^T Dn B, the row pass then the column pass of conv_lds.hip (same operations, same order)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const f32x2_t d0 = Dn[j][c], d1 = Dn[4 + j][c], d2 = Dn[8 + j][c], d3 = Dn[12 + j][c];
          Dn[j][c] = d0 - d2; Dn[4 + j][c] = d1 + d2; Dn[8 + j][c] = d2 - d1; Dn[12 + j][c] = d1 - d3;
        }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const f32x2_t e0 = Dn[a * 4][c], e1 = Dn[a * 4 + 1][c], e2 = Dn[a * 4 + 2][c], e3 = Dn[a * 4 + 3][c];
          V[a * 4][c] = e0 - e2; V[a * 4 + 1][c] = e1 + e2; V[a * 4 + 2][c] = e2 - e1; V[a * 4 + 3][c] = e1 - e3;
        }
      // the whole transform is ONE cluster here: hipcc otherwise sinks each packed add next to the MFMA that reads its result, and a
      // vector instruction alone between two MFMAs costs three times what it costs inside a cluster (profiles/r05_mfma_issue.md)
#pragma unroll
      for (int e = 0; e < 16; ++e)
#pragma unroll
        for (int c = 0; c < 2; ++c) asm volatile("" : "+v"(V[e][c]));
    };

    float af[NA][3][4];
    auto load_a = [&](int i, int buf) {             // fragments (kd 0..2, chunk, ab) of ab-step i = chunk*16 + ab
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
        if (kd < C::WL) lds_frag<4>(lds + C::W_OFF + ((kd * NSTEP + i) * NT + wn) * C::FRAG + lane * 4, af[buf][kd]);
        else buf_load_to<4>(wres, wvoff, ((kd * NSTEP + i) * NT + wn) * (64 * 4 * 4), af[buf][kd]);
      }
    };

    // output transform + epilogue of set SET for output plane o (conv_lds.hip: wino_epilogue, NT = 1 per wave)
    auto epilogue = [&](auto setc, int o) {
      constexpr int SET = decltype(setc)::value;
      if (!wave_live) return;
      const float* epi_tab = lds + C::EPI_OFF;
      const int c0 = wn * 16 + 4 * q;
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last MFMAs' results are read by the vector ALU below (nothing pads inline assembly)
      float yv[2][2][4];
      static_for<0, 2>([&](auto k2c) {        // two output channels at a time: 32 values out of the accumulator file, not 64
        constexpr int k2 = decltype(k2c)::value;
        float m[16][2];
        static_for<0, 16>([&](auto abc) { AccTile<SET * 16 + decltype(abc)::value>::template read2<k2>(m[decltype(abc)::value]); });
        f32x2_t srow[2][4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
          const f32x2_t m0 = {m[bb][0], m[bb][1]}, m1 = {m[4 + bb][0], m[4 + bb][1]};
          const f32x2_t m2 = {m[8 + bb][0], m[8 + bb][1]}, m3 = {m[12 + bb][0], m[12 + bb][1]};
          srow[0][bb] = m0 + m1 + m2;
          srow[1][bb] = m1 - m2 - m3;
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const f32x2_t y0 = srow[pr][0] + srow[pr][1] + srow[pr][2];
          const f32x2_t y1 = srow[pr][1] - srow[pr][2] - srow[pr][3];
          yv[pr][0][2 * k2] = y0[0]; yv[pr][0][2 * k2 + 1] = y0[1];
          yv[pr][1][2 * k2] = y1[0]; yv[pr][1][2 * k2 + 1] = y1[1];
        }
      });
      float al_l[4], be_l[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { al_l[k] = epi_tab[c0 + k]; be_l[k] = epi_tab[64 + c0 + k]; }
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int h = h0 + 2 * wm + pr;
        if (h >= p.H) continue;
        const size_t row_vox = (((size_t)b * p.D + o) * p.H + h) * p.W;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int ow = w0 + 2 * n16 + rr;
          if (ow >= p.W) continue;
          float ov[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            ov[k] = yv[pr][rr][k] * al_l[k] + be_l[k];
            if (p.relu) ov[k] = fmaxf(ov[k], 0.f);
          }
          const size_t oi = (row_vox + ow) * COUT + c0;
          if (p.res) {
            const float4 rv = *reinterpret_cast<const float4*>(p.res + oi);
            ov[0] = rv.x + ov[0] * p.res_scale; ov[1] = rv.y + ov[1] * p.res_scale;
            ov[2] = rv.z + ov[2] * p.res_scale; ov[3] = rv.w + ov[3] * p.res_scale;
          }
          if (!(MDF_W3_DIAG & 2) || ov[0] == 1234.5f) *reinterpret_cast<float4*>(p.y + oi) = make_float4(ov[0], ov[1], ov[2], ov[3]);
        }
      }
    };

    // One input plane zi in ring slot R (rotation R: its kd slice goes to set (R + 1 - kd) mod 3).  On entry: the plane's first
    // transformed chunk is in V, the first AH fragments are in flight, pf holds plane zi + 1 (requested before the previous
    // step's epilogue).  On exit the same for plane zi + 1.  Everything that defines V / Dn / af / pf is unconditional (reads of
    // a slot that was not refilled, a transform of it, fragment loads past the last plane, a zero-length plane request are
    // harmless): a conditional definition would put a 64-register merge behind every step.
    // kd slice kd of plane zi feeds output plane zi + 1 - kd: multiplied only if this segment owns that plane (k0, k1, k2:
    // wave-uniform); fresh1: zi is plane 0 of the volume, so kd 1 (not kd 0 of plane -1) starts output plane 0's set.
    auto step = [&](auto rc, int zi, int z_next2, bool has_next, bool k0, bool k1, bool k2, bool fresh1) {
      constexpr int R = decltype(rc)::value;
      const bool all3 = MDF_W3_NOMASK || (k0 && k1 && k2 && !fresh1);      // the segment owns all three output planes this plane feeds: one not-taken branch per ab-step
      if (MDF_W3_PF_AT_START) issue_plane(has_next ? zi + 1 : -1);
      constexpr int SLOT_N = (R + 1) % 3;
      static_for<0, NSTEP>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int ch = i / 16, ab = i % 16;
        constexpr bool last_chunk = (ch == NCH - 1);
        if (!(MDF_W3_DIAG & 16)) load_a((i + AH) % NSTEP, (i + AH) % NA);     // weight fragments AH steps ahead; past the plane's end the next plane's first ones
        if constexpr (ch == 0 && ab == WRITE_AB) {
          if (has_next) {                 // plane zi + 1 -> its slot; everybody is past plane zi - 2
            write_plane(SLOT_N);
            __syncthreads();
          }
        }
        // the next chunk's patch, two elements per step behind the write point (the staging registers of the fill are free again):
        // the same plane's next chunk, or the next plane's first one
        if constexpr (ab > WRITE_AB && ab <= WRITE_AB + 8) {
          constexpr int e0 = 2 * (ab - WRITE_AB - 1);
          if constexpr (!last_chunk) { read_elem(R, ch + 1, e0); read_elem(R, ch + 1, e0 + 1); }
          else { read_elem(SLOT_N, 0, e0); read_elem(SLOT_N, 0, e0 + 1); }
        }
        __builtin_amdgcn_sched_barrier(0);
        // (rows beyond the volume are zero halos in LDS: a wave with no live row multiplies them instead of idling at the barrier)
        // kd-major: a dependent chain of four MFMAs per accumulator issues at the full rate (profiles/r05_mfma_issue.md)
        constexpr int S0 = ((R + 1) % 3) * 16 + ab, S1 = R * 16 + ab, S2 = ((R + 2) % 3) * 16 + ab;
        if (__builtin_expect(all3, 1)) {
          if constexpr (ch == 0) AccTile<S0>::mfma_fresh(af[i % NA][0][0], V[ab][0][0]);      // kd 0 starts output plane zi + 1
          else AccTile<S0>::mfma(af[i % NA][0][0], V[ab][0][0]);
          AccTile<S0>::mfma(af[i % NA][0][1], V[ab][0][1]);
          AccTile<S0>::mfma(af[i % NA][0][2], V[ab][1][0]);
          AccTile<S0>::mfma(af[i % NA][0][3], V[ab][1][1]);
          AccTile<S1>::mfma(af[i % NA][1][0], V[ab][0][0]);
          AccTile<S1>::mfma(af[i % NA][1][1], V[ab][0][1]);
          AccTile<S1>::mfma(af[i % NA][1][2], V[ab][1][0]);
          AccTile<S1>::mfma(af[i % NA][1][3], V[ab][1][1]);
          AccTile<S2>::mfma(af[i % NA][2][0], V[ab][0][0]);
          AccTile<S2>::mfma(af[i % NA][2][1], V[ab][0][1]);
          AccTile<S2>::mfma(af[i % NA][2][2], V[ab][1][0]);
          AccTile<S2>::mfma(af[i % NA][2][3], V[ab][1][1]);
        } else {       // a segment's first two and last two planes
          if (k0) {
            if constexpr (ch == 0) AccTile<S0>::mfma_fresh(af[i % NA][0][0], V[ab][0][0]);
            else AccTile<S0>::mfma(af[i % NA][0][0], V[ab][0][0]);
            AccTile<S0>::mfma(af[i % NA][0][1], V[ab][0][1]);
            AccTile<S0>::mfma(af[i % NA][0][2], V[ab][1][0]);
            AccTile<S0>::mfma(af[i % NA][0][3], V[ab][1][1]);
          }
          if (k1) {
            if constexpr (ch == 0) {
              if (fresh1) AccTile<S1>::mfma_fresh(af[i % NA][1][0], V[ab][0][0]);
              else AccTile<S1>::mfma(af[i % NA][1][0], V[ab][0][0]);
            } else {
              AccTile<S1>::mfma(af[i % NA][1][0], V[ab][0][0]);
            }
            AccTile<S1>::mfma(af[i % NA][1][1], V[ab][0][1]);
            AccTile<S1>::mfma(af[i % NA][1][2], V[ab][1][0]);
            AccTile<S1>::mfma(af[i % NA][1][3], V[ab][1][1]);
          }
          if (k2) {
            AccTile<S2>::mfma(af[i % NA][2][0], V[ab][0][0]);
            AccTile<S2>::mfma(af[i % NA][2][1], V[ab][0][1]);
            AccTile<S2>::mfma(af[i % NA][2][2], V[ab][1][0]);
            AccTile<S2>::mfma(af[i % NA][2][3], V[ab][1][1]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ab == 15 && !last_chunk) { if (!(MDF_W3_DIAG & 8)) transform(); }        // chunk boundary: the next patch is complete -> one cluster of 64 packed adds
      });
      // plane zi + 2 is requested HERE: vmcnt retires in order, so every weight-fragment wait behind a plane request waits for the
      // plane (HBM latency) too.  The next step's first AH fragments are already in flight; the transform, the epilogue and AH
      // ab-steps pass before a wait can see these loads (requested at the step's start they stalled its first ab-step: 18 % of the wave time)
      if (!MDF_W3_PF_AT_START) issue_plane(z_next2);
      if (!(MDF_W3_DIAG & 8)) transform();
      if (k2 && !(MDF_W3_DIAG & 4)) epilogue(std::integral_constant<int, (R + 2) % 3>{}, zi - 1);    // kd 2 completed output plane zi - 1
    };

    // ---- segment prologue: planes da - 1 .. db feed the output planes da .. db - 1 (planes -1 and D are zero: skipped)
    int zi = (da == 0) ? 0 : da - 1;
    const int z_last = (db == p.D) ? p.D - 1 : db;
    __syncthreads();                                    // the previous segment's readers are done with the ring; epi table written
    issue_plane(zi);
    write_plane(0);
    __syncthreads();
    if (!MDF_W3_PF_AT_START) issue_plane(zi < z_last ? zi + 1 : -1);
#pragma unroll
    for (int e = 0; e < 16; ++e) read_elem(0, 0, e);
    transform();
#pragma unroll
    for (int i = 0; i < AH; ++i) load_a(i, i % NA);

    // three steps per trip, one per rotation, straight-line (no merge of the register arrays between them)
    int r_last = 0;
#define W3_STEP(RR)                                                                                          \
    {                                                                                                        \
      const bool has_next = zi < z_last;                                                                     \
      step(std::integral_constant<int, RR>{}, zi, (zi + 2 <= z_last) ? zi + 2 : -1, has_next,                \
           zi + 1 >= da && zi + 1 < db, zi >= da && zi < db, zi - 1 >= da && zi - 1 < db, zi == 0);          \
      if (!has_next) { r_last = RR; break; }                                                                 \
      ++zi;                                                                                                  \
    }
    for (;;) {
      W3_STEP(0)
      W3_STEP(1)
      W3_STEP(2)
    }
#undef W3_STEP
    if (db == p.D) {       // output plane D - 1 got its last term (kd = 1) from plane D - 1 = the last step: set (r_last + 1 - 1) mod 3
      switch (r_last) {
        case 0: epilogue(std::integral_constant<int, 0>{}, p.D - 1); break;
        case 1: epilogue(std::integral_constant<int, 1>{}, p.D - 1); break;
        default: epilogue(std::integral_constant<int, 2>{}, p.D - 1); break;
      }
    }
  }
}

template <int CIN, int COUT>
int launch_wino3d(Wino3dParams& p, hipStream_t st) {
  typedef W3<CIN, COUT> C;
  constexpr int NA = MDF_W3_NA;
  constexpr int WRITE_AB = MDF_W3_WRITE_AB;   // ab-step of a plane's first chunk at which the NEXT plane goes to LDS (its loads left at the step's start)
  p.tiles_h = (p.H + C::TH - 1) / C::TH;
  p.tiles_w = (p.W + C::TWO - 1) / C::TWO;
  p.total = (long long)p.B * p.tiles_h * p.tiles_w * p.D;
  if (p.total >= (1ll << 31)) return MDF_EUNSUPPORTED;
  auto kern = &wino3d_kernel<CIN, COUT, NA, WRITE_AB>;
  static bool attr_done_dev[64] = {};
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  bool& attr_done = attr_done_dev[(dev_id >= 0 && dev_id < 64) ? dev_id : 0];
  if (!attr_done || dev_id >= 64) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
    if (e != hipSuccess) return mdf::fail(MDF_EHIP, "hipFuncSetAttribute(dynamic LDS %zu): %s", (size_t)C::LDS_BYTES, hipGetErrorString(e));
    attr_done = true;
  }
  // one block per CU (192 accumulator registers per lane); every block at least ~4 planes so that a segment's two halo planes stay a fraction
  long long grid = 256;
  if (const char* g = getenv("MDF_WINO3D_GRID")) { if (atoi(g) > 0) grid = atoi(g); }   // dev
  if (grid > p.total / 4) grid = p.total / 4;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), C::LDS_BYTES, st, p);
  return mdf::check_launch("wino3d_kernel");
}

}  // namespace

// Internal entry used by mdf_conv_lds_dispatch (conv_lds.hip) for the eval-mode stride-1 3x3x3 layers; wpack points at the layer's
// transform-domain fragments.  MDF_EUNSUPPORTED: no instantiation for this channel pair.
int mdf_wino3d_dispatch(const float* x, const float* wpack_wino, const float* alpha, const float* beta, const float* res, float res_scale,
                        float* y, int B, int D, int H, int W, int Cin, int Cout, int relu, void* stream) {
  Wino3dParams p{};
  p.x = x; p.wpack = wpack_wino; p.alpha = alpha; p.beta = beta; p.res = res; p.res_scale = res_scale; p.y = y;
  p.B = B; p.D = D; p.H = H; p.W = W; p.relu = relu;
  if ((long long)H * W * Cin * 4 >= (1ll << 31)) return MDF_EUNSUPPORTED;   // plane offsets are 31-bit
  if (Cin == 32 && Cout == 16) return launch_wino3d<32, 16>(p, (hipStream_t)stream);
  if (Cin == 16 && Cout == 16) return launch_wino3d<16, 16>(p, (hipStream_t)stream);
  // (32 -> 32 @24x74x100 was built and measured: 81 us against 69 us in conv_lds.hip's form -- 76 four-row tiles x 24 planes leave a block
  //  ~7 planes per segment, two of them halo work, and a quarter of the 128-column tiles is empty; bit-identical, not instantiated)
  return MDF_EUNSUPPORTED;
}
