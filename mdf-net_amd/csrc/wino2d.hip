// Winograd F(2x2,3x3) for the 3x3 stride-1 layers of the feature pyramid (Cin = Cout = 32 / 64; backbone.py:17-30), eval mode: the
// 2-D sibling of wino3d.hip, built on the same findings (profiles/r05_mfma_issue.md) -- on gfx950 a vector instruction between two
// fp32 MFMAs costs matrix-pipe time, so the kernel is organised around the number of vector instructions per tile:
//   * conv_lds.hip's 2-D Winograd form spends ~955 vector instructions per 8 x 32 tile of the 32 -> 32 layer next to 256 MFMAs per
//     wave (3.7 per MFMA): per-tile index arithmetic of the fill (divisions, 64-bit addresses), accumulator reads repeated per
//     output branch, transforms interleaved one by one with the MFMAs.  Here: the fill's patch coordinates and LDS addresses are
//     lane constants (two registers per element), each transform is ONE cluster, every accumulator is read once;
//   * the accumulator tiles are pinned (wino3d_acc.h: a[64:..]), weight fragments and the fill's staging registers live in a[0:63]
//     and the other free accumulator registers, so the next tile's loads are in flight a whole tile ahead at no vector-register cost;
//   * tiles are handed out as one contiguous run per block (no atomics), three LDS slots rotate (one barrier per tile).
// Per accumulator the MFMA order is conv_lds.hip's (cin chunk, k): the two kernels agree bit for bit.
//
// S2D: the 5x5 stride-2 layers of the pyramid (8 -> 16, 16 -> 32; backbone.py:20-27) as a 3x3 stride-1 conv over the FOUR PARITY IMAGES of
// their input (conv_lds.hip, LdsConvParams::s2d; weights: conv3d.hip kSrcK5S2Phases): 16 instead of 25 MFMA groups per input channel.
// x is then [B, 2H, 2W, CIN/4]; cin group g = parity (py*2+px) * (CIN/16) + 4-channel slice, and only the fill's addresses change.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "conv_lds_common.h"
#define W3_PINNED_TILES_32      // 2 cout tiles x 16 transform-domain elements per wave: a[64:191] pinned, a[192:255] stay with the compiler
#include "wino3d_acc.h"

namespace {

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

struct Wino2dParams {
  const float* x;      // [B,H,W,CIN]; S2D: [B,2H,2W,CIN/4]
  const float* wpack;  // transform-domain fragments [chunk][ab][nt][lane][4] (conv3d.hip: pack_wino_elem, nkd = 1)
  const float* alpha;  // [COUT] or null
  const float* beta;   // [COUT] or null
  const float* res;    // [B,H,W,COUT] or null
  float* y;            // [B,H,W,COUT]
  float res_scale;
  int B, H, W, relu;
  int tiles_h, tiles_w, n_tiles;
};

template <int CIN, int COUT>
struct W2 {
  static_assert(CIN % 16 == 0 && COUT % 16 == 0 && COUT <= 64, "2-D Winograd form: Cin, Cout multiples of 16, Cout <= 64");
  static constexpr int NCH = CIN / 16, NT = COUT / 16, NG = CIN / 4;
  // a wave owns two output rows and NTW <= 2 cout tiles (32 pinned accumulator tiles); with 64 input channels a 10-row patch does not
  // fit three times (87 KB): the four waves are 2 row pairs x 2 cout halves (both halves transform the same patch: 0.5 packed adds per
  // MFMA with 64 output channels, 1 with 32), the tile is 4 rows high
  static constexpr int WN = (CIN >= 64) ? 2 : 1, NTW = NT / WN, WM = 4 / WN;
  static_assert(NT % WN == 0 && NTW >= 1 && NTW <= 2, "cout tiles per wave");
  static constexpr int TH = 2 * WM, TWO = 32;
  static constexpr int PH = TH + 2, PW = TWO + 2;
  static constexpr int NPP = PH * PW;
  static constexpr int S = round_s(NPP, 4, 2);
  static constexpr int PLANE = CIN * S;             // floats
  static constexpr int RING = 3;
  static constexpr int NPOS = NG * NPP;
  static constexpr int NFILL = (NPOS + 255) / 256;
  static constexpr int EPI_OFF = RING * PLANE;      // alpha[64], beta[64]
  static constexpr int FRAG = 64 * 4;
  static constexpr int NSTEP = NCH * 16;
  // weight fragments resident in LDS where they fit beside the three tiles (16 -> 16: 16 KB)
  static constexpr bool WL = ((size_t)(RING * PLANE + 128 + NSTEP * NT * FRAG) * sizeof(float) <= 160 * 1024);
  static constexpr int W_OFF = EPI_OFF + 128;
  static constexpr size_t LDS_BYTES = (size_t)(RING * PLANE + 128 + (WL ? NSTEP * NT * FRAG : 0)) * sizeof(float);
  static_assert(S > NPP, "a pad vector per group takes the fill's surplus lanes");
};

#ifndef MDF_W2_DIAG
#define MDF_W2_DIAG 0        // dev ablations (wrong results): 1 no input traffic, 2 no output stores, 4 no epilogue, 8 no transforms in the steps, 16 no fragment loads in the steps, 32 no patch reads
#endif
#ifndef MDF_W2_HOLD_REL
#define MDF_W2_HOLD_REL 0
#endif
#ifndef MDF_W2_NA
#define MDF_W2_NA 4
#endif
#ifndef MDF_W2_WRITE_AB
#define MDF_W2_WRITE_AB 2
#endif

template <int CIN, int COUT, int NA, int WRITE_AB, bool S2D>
__global__ __launch_bounds__(256, 1) void wino2d_kernel(const Wino2dParams p) {
  typedef W2<CIN, COUT> C;
  constexpr int NCH = C::NCH, NT = C::NT, S = C::S, PW = C::PW, NPP = C::NPP, NFILL = C::NFILL, NSTEP = C::NSTEP;
  constexpr int AH = NA - 1;
  static_assert(NSTEP % NA == 0, "the fragment ring must close over a tile");
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, n16 = lane & 15;
  const __amdgpu_buffer_rsrc_t wres = make_rsrc(p.wpack, (unsigned)(NSTEP * NT * 64 * 4 * 4));
  const int wvoff = lane * 16;
  const int wm = wave / C::WN, wn = wave % C::WN;
  constexpr int NTW = C::NTW;
  const int lane_lds = (q * S + wm * 2 * PW + n16 * 2) * 4;

  if (tid < 128) {
    const int c = tid & 63;
    lds[C::EPI_OFF + tid] = (tid < 64) ? ((c < COUT && p.alpha) ? p.alpha[c] : 1.f) : ((c < COUT && p.beta) ? p.beta[c] : 0.f);
  }
  if constexpr (C::WL) {
    constexpr int NV = NSTEP * NT * 64;
    for (int v = tid; v < NV; v += 256)
      *reinterpret_cast<float4*>(lds + C::W_OFF + v * 4) = *reinterpret_cast<const float4*>(p.wpack + (size_t)v * 4);
  }

  // ---- fill constants of the thread (element k = vector v of cin group g, 4 groups fastest: wino3d.hip): its LDS address, its patch
  // row / column (packed) and its offset inside the image relative to the patch origin
  // (64 channels: 13 elements -- the offset is rebuilt from the packed coordinates per tile instead of held: registers)
  constexpr bool HOLD_REL = (NFILL <= 11) || (NTW == 1) || MDF_W2_HOLD_REL;
  constexpr int CM = S2D ? CIN / 4 : CIN;            // channels of a pixel in memory
  [[maybe_unused]] constexpr int GPP = CM / 4;       // S2D: 4-channel slices per parity image
  auto rel_of = [&](int row, int col, int g) -> int {      // byte offset of element (patch row, patch column, cin group) from the patch origin
    if constexpr (S2D) {
      const int par = g / GPP, c4 = g - par * GPP;
      return (((2 * row + (par >> 1)) * (2 * p.W) + 2 * col + (par & 1)) * CM + c4 * 4) * 4;
    } else {
      return ((row * p.W + col) * CIN + g * 4) * 4;
    }
  };
  int f_lds[NFILL], f_rc[NFILL], f_rel[HOLD_REL ? NFILL : 1];
#pragma unroll
  for (int k = 0; k < NFILL; ++k) {
    const int idx = tid + k * 256;
    const int glo = idx & 3, r = idx >> 2;
    const int v = r % NPP, g = (r / NPP) * 4 + glo;
    const bool live = idx < C::NPOS;
    const int row = live ? v / PW : 0x40, col = v % PW;           // (a dead lane's row is outside every tile's valid range)
    f_lds[k] = live ? (g * S + v) * 4 : NPP * 4;
    f_rc[k] = (row << 16) | (col << 8) | g;
    if constexpr (HOLD_REL) f_rel[k] = live ? rel_of(row, col, g) : (int)0x80000000u;     // (a dead lane reads nothing)
  }

  // ---- the block's run of tiles
  const unsigned lb = mdf::xcd_remap(blockIdx.x, gridDim.x);
  const int t_begin = __builtin_amdgcn_readfirstlane((int)((unsigned long long)lb * (unsigned)p.n_tiles / gridDim.x));
  const int t_end = __builtin_amdgcn_readfirstlane((int)((unsigned long long)(lb + 1) * (unsigned)p.n_tiles / gridDim.x));
  if (t_begin >= t_end) return;

  float4 pf[NFILL];
  // tile coordinates live in scalar registers and advance tile by tile: two integer divisions per tile (issue + epilogue) were ~100 vector
  // instructions of the ~250 a tile spends outside its MFMAs and transforms (ablation: profiles/r05_wino2d.md)
  struct TileAt { int b, th, tw; };
  auto tile_at = [&](int t) -> TileAt {
    const int tw = t % p.tiles_w, rest = t / p.tiles_w;
    return TileAt{__builtin_amdgcn_readfirstlane(rest / p.tiles_h), __builtin_amdgcn_readfirstlane(rest % p.tiles_h), __builtin_amdgcn_readfirstlane(tw)};
  };
  auto tile_advance = [&](TileAt& c) {
    if (++c.tw == p.tiles_w) {
      c.tw = 0;
      if (++c.th == p.tiles_h) { c.th = 0; ++c.b; }
    }
  };
  auto issue_tile = [&](const TileAt& at, bool valid) {      // global loads of the tile's patch -> pf (zeros outside the image; !valid: nothing)
    const int b = valid ? at.b : 0, h0 = (valid ? at.th : 0) * C::TH, w0 = (valid ? at.tw : 0) * C::TWO;
    // base = the patch origin (h0 - 1, w0 - 1) of image b: may lie before the image, every VALID element's address does not
    const long long org = S2D ? ((long long)b * (2 * p.H) + 2 * (h0 - 1)) * (2 * p.W) + 2 * (w0 - 1)      // (pixels of the full-resolution map)
                              : ((long long)b * p.H + (h0 - 1)) * p.W + (w0 - 1);
    const float* pz = p.x + org * CM;
    const unsigned long long pa = (unsigned long long)pz;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pa), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(pa >> 32));
    const unsigned nrec = (!valid || (MDF_W2_DIAG & 1)) ? 0u : 0x7fffffffu;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc((const void*)(((unsigned long long)hi << 32) | lo), nrec);
    const int r_lo = 1 - h0, r_hi = p.H + 1 - h0, c_lo = 1 - w0, c_hi = p.W + 1 - w0;     // valid patch rows / columns of this tile
    if constexpr (HOLD_REL) {
      if (r_lo <= 0 && r_hi >= C::PH && c_lo <= 0 && c_hi >= C::PW) {      // the whole patch lies inside the image (most tiles): no bounds arithmetic
#pragma unroll
        for (int k = 0; k < NFILL; ++k) {
          const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(xr, f_rel[k], 0, 0);
          pf[k] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
        return;
      }
    }
#pragma unroll
    for (int k = 0; k < NFILL; ++k) {
      const int row = f_rc[k] >> 16, col = (f_rc[k] >> 8) & 0xff;
      const bool ok = row >= r_lo && row < r_hi && col >= c_lo && col < c_hi;
      int rel;
      if constexpr (HOLD_REL) rel = f_rel[k];
      else rel = rel_of(row, col, f_rc[k] & 0xff);
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? rel : (int)0x80000000u, 0, 0);
      pf[k] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  };
  auto write_tile = [&](int slot) {
#pragma unroll
    for (int k = 0; k < NFILL; ++k) *reinterpret_cast<float4*>(lds + slot * C::PLANE + f_lds[k]) = pf[k];
  };

  f32x2_t V[16][2], Dn[16][2];
  auto read_elem = [&](int slot, int ch, int e) {
    const int i = e >> 2, j = e & 3;
    float t[4];
    lds_frag<4>(lds + slot * C::PLANE + lane_lds + ((ch * 4) * S + i * PW + j) * 4, t);
    Dn[e][0] = (f32x2_t){t[0], t[1]};
    Dn[e][1] = (f32x2_t){t[2], t[3]};
  };
  auto transform = [&]() {                        // V = B^T Dn B: conv_lds.hip's row pass then column pass, as ONE cluster (wino3d.hip)
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
#pragma unroll
    for (int e = 0; e < 16; ++e)
#pragma unroll
      for (int c = 0; c < 2; ++c) asm volatile("" : "+v"(V[e][c]));
  };

  float af[NA][NTW][4];
  auto load_a = [&](int i, int buf) {             // this wave's fragments (chunk, ab, nt = wn*NTW + ntl) of ab-step i = chunk*16 + ab
#pragma unroll
    for (int ntl = 0; ntl < NTW; ++ntl) {
      if constexpr (C::WL) lds_frag<4>(lds + C::W_OFF + (i * NT + wn * NTW + ntl) * C::FRAG + lane * 4, af[buf][ntl]);
      else buf_load_to<4>(wres, wvoff, (i * NT + wn * NTW + ntl) * (64 * 4 * 4), af[buf][ntl]);
    }
  };

  // output transform + epilogue of tile t (conv_lds.hip: wino_epilogue); accumulator tile nt*16 + ab
  auto epilogue = [&](const TileAt& at) {
    const int b = at.b, h0 = at.th * C::TH, w0 = at.tw * C::TWO;
    if (h0 + 2 * wm >= p.H) return;
    const float* epi_tab = lds + C::EPI_OFF;
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    static_for<0, NTW>([&](auto ntc) {
      constexpr int nt = decltype(ntc)::value;         // local tile; channels of cout tile wn*NTW + nt
      const int c0 = (wn * NTW + nt) * 16 + 4 * q;
      float yv[2][2][4];
      static_for<0, 2>([&](auto k2c) {
        constexpr int k2 = decltype(k2c)::value;
        float m[16][2];
        static_for<0, 16>([&](auto abc) { AccTile<nt * 16 + decltype(abc)::value>::template read2<k2>(m[decltype(abc)::value]); });
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
        const size_t row_vox = ((size_t)b * p.H + h) * p.W;
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
          if (!(MDF_W2_DIAG & 2) || ov[0] == 1234.5f) *reinterpret_cast<float4*>(p.y + oi) = make_float4(ov[0], ov[1], ov[2], ov[3]);
        }
      }
    });
  };

  // One tile in ring slot R.  On entry: its first transformed chunk is in V, the first AH fragments are in flight, pf holds tile
  // t + 1 (requested before the previous tile's epilogue).  On exit the same for tile t + 1 (unconditional definitions: wino3d.hip).
  auto step = [&](auto rc, const TileAt& at, const TileAt& at_next2, bool valid_next2, bool has_next) {
    constexpr int R = decltype(rc)::value;
    constexpr int SLOT_N = (R + 1) % 3;
    static_for<0, NSTEP>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int ch = i / 16, ab = i % 16;
      constexpr bool last_chunk = (ch == NCH - 1);
      if (!(MDF_W2_DIAG & 16)) load_a((i + AH) % NSTEP, (i + AH) % NA);
      if constexpr (ch == 0 && ab == WRITE_AB) {
        if (has_next) {                 // tile t + 1 -> its slot; everybody is past tile t - 2
          write_tile(SLOT_N);
          __syncthreads();
        }
      }
      if constexpr (ab > WRITE_AB && ab <= WRITE_AB + 8 && !(MDF_W2_DIAG & 32)) {
        constexpr int e0 = 2 * (ab - WRITE_AB - 1);
        if constexpr (!last_chunk) { read_elem(R, ch + 1, e0); read_elem(R, ch + 1, e0 + 1); }
        else { read_elem(SLOT_N, 0, e0); read_elem(SLOT_N, 0, e0 + 1); }
      }
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NTW>([&](auto ntc) {
        constexpr int nt = decltype(ntc)::value;
        constexpr int T = nt * 16 + ab;
        if constexpr (ch == 0) AccTile<T>::mfma_fresh(af[i % NA][nt][0], V[ab][0][0]);
        else AccTile<T>::mfma(af[i % NA][nt][0], V[ab][0][0]);
        AccTile<T>::mfma(af[i % NA][nt][1], V[ab][0][1]);
        AccTile<T>::mfma(af[i % NA][nt][2], V[ab][1][0]);
        AccTile<T>::mfma(af[i % NA][nt][3], V[ab][1][1]);
      });
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ab == 15 && !last_chunk) { if (!(MDF_W2_DIAG & 8)) transform(); }
    });
    issue_tile(at_next2, valid_next2);          // behind the tile's MFMAs, ahead of the epilogue (vmcnt retires in order: wino3d.hip)
    if (!(MDF_W2_DIAG & 8)) transform();
    if (!(MDF_W2_DIAG & 4)) epilogue(at);
  };

  // ---- prologue
  int t = t_begin;
  const int t_last = t_end - 1;
  TileAt at = tile_at(t), at2 = at;
  issue_tile(at, true);
  write_tile(0);
  __syncthreads();
  tile_advance(at2);
  issue_tile(at2, t < t_last);
  tile_advance(at2);              // tile t + 2
#pragma unroll
  for (int e = 0; e < 16; ++e) read_elem(0, 0, e);
  transform();
#pragma unroll
  for (int i = 0; i < AH; ++i) load_a(i, i % NA);

#define W2_STEP(RR)                                                                        \
  {                                                                                        \
    const bool has_next = t < t_last;                                                      \
    step(std::integral_constant<int, RR>{}, at, at2, t + 2 <= t_last, has_next);           \
    if (!has_next) break;                                                                  \
    ++t; tile_advance(at); tile_advance(at2);                                              \
  }
  for (;;) {
    W2_STEP(0)
    W2_STEP(1)
    W2_STEP(2)
  }
#undef W2_STEP
}

template <int CIN, int COUT, bool S2D = false>
int launch_wino2d(Wino2dParams& p, hipStream_t st) {
  typedef W2<CIN, COUT> C;
  constexpr int NA = MDF_W2_NA, WRITE_AB = MDF_W2_WRITE_AB;
  p.tiles_h = (p.H + C::TH - 1) / C::TH;
  p.tiles_w = (p.W + C::TWO - 1) / C::TWO;
  const long long tiles = (long long)p.B * p.tiles_h * p.tiles_w;
  if (tiles >= (1ll << 31)) return MDF_EUNSUPPORTED;
  p.n_tiles = (int)tiles;
  auto kern = &wino2d_kernel<CIN, COUT, NA, WRITE_AB, S2D>;
  static bool attr_done_dev[64] = {};
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  bool& attr_done = attr_done_dev[(dev_id >= 0 && dev_id < 64) ? dev_id : 0];
  if (!attr_done || dev_id >= 64) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
    if (e != hipSuccess) return mdf::fail(MDF_EHIP, "hipFuncSetAttribute(dynamic LDS %zu): %s", (size_t)C::LDS_BYTES, hipGetErrorString(e));
    attr_done = true;
  }
  long long grid = 256;      // one block per CU
  if (const char* g = getenv("MDF_WINO2D_GRID")) { if (atoi(g) > 0) grid = atoi(g); }   // dev
  if (grid > tiles / 2) grid = tiles / 2;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), C::LDS_BYTES, st, p);
  return mdf::check_launch("wino2d_kernel");
}

}  // namespace

// Internal entry used by mdf_conv_lds_dispatch (conv_lds.hip) for the eval-mode 3x3 stride-1 2-D layers; wpack_wino points at the
// layer's transform-domain fragments.  MDF_EUNSUPPORTED: no instantiation for this channel pair.
int mdf_wino2d_dispatch(const float* x, const float* wpack_wino, const float* alpha, const float* beta, const float* res, float res_scale,
                        float* y, int B, int H, int W, int Cin, int Cout, int relu, void* stream) {
  Wino2dParams p{};
  p.x = x; p.wpack = wpack_wino; p.alpha = alpha; p.beta = beta; p.res = res; p.res_scale = res_scale; p.y = y;
  p.B = B; p.H = H; p.W = W; p.relu = relu;
  if ((long long)B * H * W * Cin * 4 >= (1ll << 31)) return MDF_EUNSUPPORTED;     // byte offsets inside a patch are 31-bit
  // (16 -> 16 @592x800x5 is built and bit-identical too, but that layer moves 350 MB in ~100 us in either kernel (3.4 TB/s): it is
  //  memory-bound, and conv_lds.hip's form with two or three blocks per CU is 1-3 % faster; dev: MDF_WINO2D_16=1)
  if (Cin == 16 && Cout == 16 && getenv("MDF_WINO2D_16")) return launch_wino2d<16, 16>(p, (hipStream_t)stream);
  if (Cin == 32 && Cout == 32) return launch_wino2d<32, 32>(p, (hipStream_t)stream);
  if (Cin == 64 && Cout == 64) return launch_wino2d<64, 64>(p, (hipStream_t)stream);
  return MDF_EUNSUPPORTED;
}

// The 5x5 stride-2 layers over the parity images of their input: x [B,2*Ho,2*Wo,Cin_mem], y [B,Ho,Wo,Cout]; wpack_k5w = the layer's
// transform-domain fragments over 4*Cin_mem logical input channels (conv3d.hip: k5w_built).
int mdf_wino2d_s2d_dispatch(const float* x, const float* wpack_k5w, const float* alpha, const float* beta, float* y, int B, int Ho, int Wo,
                            int Cin_mem, int Cout, int relu, void* stream) {
  Wino2dParams p{};
  p.x = x; p.wpack = wpack_k5w; p.alpha = alpha; p.beta = beta; p.res = nullptr; p.res_scale = 0.f; p.y = y;
  p.B = B; p.H = Ho; p.W = Wo; p.relu = relu;
  if ((long long)B * Ho * Wo * 4 * Cin_mem * 4 >= (1ll << 31)) return MDF_EUNSUPPORTED;     // byte offsets inside a patch are 31-bit
  if (Cin_mem == 16 && Cout == 32) return launch_wino2d<64, 32, true>(p, (hipStream_t)stream);
  if (Cin_mem == 8 && Cout == 16) return launch_wino2d<32, 16, true>(p, (hipStream_t)stream);
  return MDF_EUNSUPPORTED;
}
