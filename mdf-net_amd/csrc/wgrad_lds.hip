// LDS-staged weight-gradient kernels (the fast path of mdf_conv3d_wgrad / mdf_conv2d_wgrad, wgrad.hip).
//
// Why: the first kernels fed every MFMA operand with one `global_load_dword` per lane.  The texture-address unit retires
// 4 lanes per clock whatever the width, so a 64-lane dword load costs the CU 16 clocks -- and 1.1-1.3 such loads are needed
// per MFMA (8 CU-clocks): PMC showed the kernels TA-issue-bound at 8-40 TFLOP/s, with every `big` element fetched once per
// tap (9-27 times).  Here a block copies a tv-voxel row segment of `small` and the (KH rows x (tv*s + KW - 1) voxels) patch
// of `big` into LDS with 16-byte loads (each element once per tile, out-of-range voxels zero-filled, next tile's loads in
// flight in registers during the MFMAs), and the waves take their operands from LDS with `ds_read_b32` (2 LDS clocks per
// 64-lane read).  Voxel strides in LDS are padded so that the two 16-lane groups of a 32-lane half hit disjoint banks.
//
//     dw[a][b][z][kh][kw] (+)= sum_o small[o][a] * big[s*o + tap - pad][b]       z = blockIdx.z (kd in 3-D, kh in 2-D)
#include <cstdlib>
#include <vector>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WgradLdsParams {
  const float* small_;   // rows of Ws voxels x A channels
  const float* big;      // rows of Wb voxels x Bc channels
  float* slab;           // [gridDim.x][A][Bc][ntaps_total]
  int B, Ds, Hs, Ws, Db, Hb, Wb, A, Bc, stride;
  int AS, BS;            // padded voxel strides in LDS (floats)
  int wtiles;            // ceil(Ws/tv)
  int htiles;            // ceil(Hs/TH): a tile is TH consecutive rows of one (n, od) plane x tv voxels
  long long n_tiles;     // B * Ds * htiles * wtiles
  int NB, split;
  int ntaps_total;       // 27 (3-D) or KS*KS (2-D)
  int pad;               // spatial padding of the taps inside a row (and of the z tap)
  int tv;                // voxels of `small` per tile along w: 64, or 256 for few-channel layers (more MFMAs per barrier)
  float* zero_out;       // when set: zero_n floats cleared by this launch (the dw the slab is summed into next)
  int zero_n;
  int pa, pb;            // tap packing (R > 0): `small` shifts in the tile rows (A <= 8), `big` shifts in the tile columns (Bc <= 8)
};

// KH kernel rows x KW taps per block; MODE3D: z = kd and the rows are the 3 kh rows of depth plane od*s+kd-1;
// otherwise (2-D): z = kh, one row.
//
// R > 0 (stride 1, KW = 3, A <= 8 and/or Bc <= 8): TAP PACKING.  An 8 x 8 channel pair fills a quarter of a 16 x 16 MFMA tile.
// Tile row (sa, a) instead holds `small` shifted by sa voxels and tile column (sb, b) holds `big` shifted by sb:
//     D[(sa,a)][(sb,b)] = sum_v small[v - sa][a] * big[v + t + sb][b] = dw[a][b][kw = t + sa + sb]
// so ONE chain of MFMAs yields the taps t .. t + R, R = (pa-1) + (pb-1): all three kw taps for 8 x 8 (R = 2; the (1,0) block
// duplicates (0,1) and is not stored), two steps t = 0, 2 for 8 x 16 / 16 x 8 (R = 1).  The row sum of a shifted block runs over
// v - sa, so `small` is staged with pa-1 voxels of left halo and the tiles cover Ws + pa - 1 voxels (zero beyond the row).
//
// TH > 1: a tile is TH consecutive output rows.  The volumes of a cfg3-sized step have SHORT rows (96, 48, 24 voxels): one row
// is 54-108 MFMAs per wave between two barriers and a commit, and its 6 chunks split unevenly over 4 waves.  TH rows share their
// staged `big` rows (3-D, stride 1: TH + 2 rows of a depth plane serve the 3 kernel rows of TH output rows instead of 3 TH) and
// amortise the per-tile fixed cost TH times.
// VB: the block's coordinates in its launch's (gx, gy, gz) grid -- the hardware's for a one-layer launch, computed from the job table
// for a batched launch (wgrad_lds_batch_kernel)
struct VBlock { int bx, by, bz, gx, gy, gz; };

template <int KH, int KW, bool MODE3D, int R, int TH>
__device__ __forceinline__ void wgrad_lds_body(const WgradLdsParams& p, const VBlock vb) {
  constexpr int NT = (R == 0) ? KW : (R == 1 ? 2 : 1);     // MFMA chains (accumulators) per kernel row
  constexpr int NBR = MODE3D ? TH + KH - 1 : TH;           // staged rows of `big` (3-D with TH > 1: stride 1 only, host-checked)
  constexpr int TSTEP = (R == 0) ? 1 : R + 1;              // first tap of chain ts: ts * TSTEP
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (p.zero_out) {
    const int nblk = vb.gx * vb.gy * vb.gz;
    const int bid = (vb.bz * vb.gy + vb.by) * vb.gx + vb.bx;
    for (int i = bid * 256 + threadIdx.x; i < p.zero_n; i += nblk * 256) p.zero_out[i] = 0.f;
  }
  const int s = p.stride;
  const int WB = p.tv * s + KW - 1;              // voxels of `big` per staged row
  const int hal = (R > 0) ? p.pa - 1 : 0;        // left halo of the staged `small` segment
  float* sm_small = lds;                              // [TH][hal + tv][AS]
  float* sm_big = lds + TH * (p.tv + hal) * p.AS;     // [NBR][WB][BS] (+ one voxel of slack: the discarded tap t + sa + sb = 3 reads it)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, c16 = lane & 15;
  const int pairs_per_block = 4 / p.split;
  const int pair = vb.by * pairs_per_block + wave / p.split;
  const int part = wave % p.split;
  const int na = pair / p.NB, nb = pair % p.NB;
  const int a = na * 16 + c16, bcol = nb * 16 + c16;
  const bool b_ok = (R > 0 && p.pb > 1) ? (c16 < p.Bc * p.pb) : (bcol < p.Bc);
  // operand lanes: (shift, channel) when packed; clamped lanes feed tile rows / columns that are never stored
  const int sa = (R > 0 && p.pa > 1) ? min(c16 / p.A, p.pa - 1) : 0, sb = (R > 0 && p.pb > 1) ? min(c16 / p.Bc, p.pb - 1) : 0;
  const int ac = (R > 0 && p.pa > 1) ? c16 % p.A : min(a, p.A - 1), bc = (R > 0 && p.pb > 1) ? c16 % p.Bc : min(bcol, p.Bc - 1);
  const int z = vb.bz;

  f32x4 acc[KH * NT];
#pragma unroll
  for (int t = 0; t < KH * NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Register staging of the next tile in 16-byte pieces.  A row segment is CONTIGUOUS in memory (NHWC / NDHWC), so piece i of a
  // row is `row4[first + i]`: no per-piece index arithmetic beyond a bound check; its LDS slot (voxel i >> log2(C/4), channel
  // quad i & (C/4 - 1), padded voxel stride) does not depend on the tile and is computed once.
  constexpr int NS = (TH == 1) ? 4 : 2;                                  // small: pieces per staged row and thread ((tv + hal) * A/4 <= 256 * NS, host-checked)
  constexpr int NPR = (TH == 1) ? (MODE3D ? 4 : 5) : (TH == 2 ? 4 : 2);  // big: likewise (WB * Bc/4 <= 256 * NPR)
  float4 st_small[TH][NS], st_big[NBR][NPR];
  const int c4a = p.A >> 2, c4b = p.Bc >> 2;
  const int la2 = 31 - __clz(c4a), lb2 = 31 - __clz(c4b);
  const int small_pieces = (p.tv + hal) * c4a, row_pieces = WB * c4b;
  int lds_small[NS], lds_big[NPR];
#pragma unroll
  for (int k = 0; k < NS; ++k) { const int i = threadIdx.x + 256 * k; lds_small[k] = (i >> la2) * p.AS + 4 * (i & (c4a - 1)); }
#pragma unroll
  for (int k = 0; k < NPR; ++k) { const int i = threadIdx.x + 256 * k; lds_big[k] = (i >> lb2) * p.BS + 4 * (i & (c4b - 1)); }

  auto decode = [&](long long tile, int& n, int& od, int& oh, int& ow0) {      // oh = first of the tile's TH rows
    // (32-bit: a launch has < 2^31 tiles -- the host checks; three 64-bit divisions were ~250 vector instructions per decode, r05)
    const unsigned tu = (unsigned)tile, r1 = tu / (unsigned)p.wtiles, r2 = r1 / (unsigned)p.htiles;
    const int wt = (int)(tu - r1 * (unsigned)p.wtiles);
    oh = (int)(r1 - r2 * (unsigned)p.htiles) * TH;
    n = (int)(r2 / (unsigned)p.Ds);
    od = (int)(r2 - (unsigned)n * (unsigned)p.Ds);
    ow0 = wt * p.tv;
  };
  auto stage = [&](long long tile) {          // global -> registers (zero for out-of-range voxels / rows)
    int n, od, oh, ow0;
    decode(tile, n, od, oh, ow0);             // wave-uniform (scalar unit)
    const int s_lo = max(0, hal - ow0) * c4a, s_hi = min(p.tv + hal, p.Ws - ow0 + hal) * c4a;     // staged slot 0 = voxel ow0 - hal
    const int id_s = MODE3D ? od * s + z - p.pad : 0;
    const bool plane_ok = !MODE3D || (id_s >= 0 && id_s < p.Db);     // (3-D: this kd's plane of `big` exists; else the tile adds nothing and is not multiplied, below)
#pragma unroll
    for (int r = 0; r < TH; ++r) {
      const bool srow_ok = plane_ok && (oh + r) < p.Hs;                                          // wave-uniform
      const float4* srow = reinterpret_cast<const float4*>(p.small_) + ((((long long)n * p.Ds + od) * p.Hs + (oh + r)) * p.Ws + (ow0 - hal)) * c4a;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const int i = threadIdx.x + 256 * k;
        st_small[r][k] = (srow_ok && i >= s_lo && i < s_hi) ? srow[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const int id = MODE3D ? od * s + z - p.pad : 0;
    const int v_first = ow0 * s - p.pad;                                  // big voxel of staged slot 0
    const int b_lo = max(0, -v_first) * c4b, b_hi = min(WB, p.Wb - v_first) * c4b;
#pragma unroll
    for (int br = 0; br < NBR; ++br) {
      // 3-D: staged row br holds big row oh*s + br - pad (kernel row kh of output row r reads staged row r*s + kh; TH > 1 has s = 1)
      // 2-D: staged row br = output row br, big row (oh + br)*s + z - pad
      const int ih = MODE3D ? oh * s + br - p.pad : (oh + br) * s + z - p.pad;
      const bool row_ok = (!MODE3D || (id >= 0 && id < p.Db)) && ih >= 0 && ih < p.Hb;      // wave-uniform
      const long long row = MODE3D ? (((long long)n * p.Db + id) * p.Hb + ih) : ((long long)n * p.Hb + ih);
      const float4* brow = reinterpret_cast<const float4*>(p.big) + (row * p.Wb + v_first) * c4b;
#pragma unroll
      for (int k = 0; k < NPR; ++k) {
        const int i = threadIdx.x + 256 * k;
        st_big[br][k] = (row_ok && i >= b_lo && i < b_hi) ? brow[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto commit = [&]() {                        // registers -> LDS
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int k = 0; k < NS; ++k)
        if (threadIdx.x + 256 * k < small_pieces) *reinterpret_cast<float4*>(sm_small + r * (p.tv + hal) * p.AS + lds_small[k]) = st_small[r][k];
#pragma unroll
    for (int br = 0; br < NBR; ++br)
#pragma unroll
      for (int k = 0; k < NPR; ++k)
        if (threadIdx.x + 256 * k < row_pieces) *reinterpret_cast<float4*>(sm_big + br * WB * p.BS + lds_big[k]) = st_big[br][k];
  };

  // chunks of the tile this wave multiplies: the `split` waves that share a pair take every split-th chunk
  const int cpr = p.tv / 16;                                           // chunks per row
  const int nchunk = (TH * cpr - part + p.split - 1) / p.split;        // tv is any multiple of 16: the first waves may take one chunk more
  const float* la = sm_small + (q + hal - sa) * p.AS + ac;
  const float* lb = sm_big + (q * s + sb) * p.BS + bc;

  // 3-D, shallow volumes (the innermost levels are 1-2 planes deep): the tiles whose kd plane lies outside the volume -- two of three
  // at Ds = 1 -- staged zeros and multiplied them; they are passed over (the barriers stay: the pipeline's shape does not change)
  auto tile_live = [&](long long t) {
    if constexpr (!MODE3D) return true;
    int n, od, oh, ow0;
    decode(t, n, od, oh, ow0);
    const int id = od * s + z - p.pad;
    return id >= 0 && id < p.Db;
  };
  long long tile = vb.bx;
  if (tile < p.n_tiles) stage(tile);
  while (tile < p.n_tiles) {
    commit();
    __syncthreads();
    const long long next = tile + vb.gx;
    if (next < p.n_tiles) stage(next);         // in flight during the MFMAs below
    const int nchunk_t = tile_live(tile) ? nchunk : 0;
    for (int cq = 0; cq < nchunk_t; ++cq) {
      const int c = part + cq * p.split;
      const int r = (TH == 1) ? 0 : c / cpr;        // row of the tile
      const int v0 = (c - r * cpr) * 16;            // first voxel of the chunk inside the row
      const float* lar = la + r * (p.tv + hal) * p.AS;
      const float* lbr = lb + (MODE3D ? r * s : r) * WB * p.BS;
      float af[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) af[j] = lar[(v0 + 4 * j) * p.AS];
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
        for (int ts = 0; ts < NT; ++ts) {
          float bf[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[j] = lbr[(kh * WB + (v0 + 4 * j) * s + ts * TSTEP) * p.BS];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[kh * NT + ts] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], acc[kh * NT + ts], 0, 0, 0);
        }
      }
    }
    __syncthreads();                            // everybody is done with this tile's LDS image
    tile = next;
  }

  // partial tiles of the `split` waves that share a pair: summed through LDS (the staging area is free now)
  float* mine = lds + (wave / p.split) * (KH * NT * 4 * 64);
  for (int turn = 1; turn < p.split; ++turn) {
    if (part == turn) {
#pragma unroll
      for (int t = 0; t < KH * NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) mine[(t * 4 + i) * 64 + lane] = acc[t][i];
    }
    __syncthreads();
    if (part == 0) {
#pragma unroll
      for (int t = 0; t < KH * NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] += mine[(t * 4 + i) * 64 + lane];
    }
    __syncthreads();
  }
  if (part == 0 && b_ok) {
    float* out = p.slab + (long long)vb.bx * p.A * p.Bc * p.ntaps_total;
    if constexpr (R == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = na * 16 + 4 * q + i;
        if (row < p.A) {
          float* o = out + ((long long)row * p.Bc + bcol) * p.ntaps_total + z * (KH * KW);
#pragma unroll
          for (int t = 0; t < KH * KW; ++t) o[t] = acc[t][i];
        }
      }
    } else {
      // tile (row, column) = ((sa, a), (sb, b)); chain ts holds tap kw = ts*TSTEP + sa + sb.  Every (a, b, kw) is stored once: by the
      // block with the largest row shift that reaches it (sa = min(m, pa-1), m = kw - ts*TSTEP).
      const int sbc = (p.pb > 1) ? c16 / p.Bc : 0, bch = (p.pb > 1) ? c16 % p.Bc : bcol;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = na * 16 + 4 * q + i;
        const int sar = (p.pa > 1) ? row / p.A : 0, ach = (p.pa > 1) ? row % p.A : row;
        if ((p.pa > 1) ? (row >= p.A * p.pa) : (row >= p.A)) continue;
        const int m = sar + sbc;
        if (sar != min(m, p.pa - 1)) continue;
        float* o = out + ((long long)ach * p.Bc + bch) * p.ntaps_total + z * (KH * KW);
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
          for (int ts = 0; ts < NT; ++ts) {
            const int kw = ts * TSTEP + m;
            if (kw < KW) o[kh * KW + kw] = acc[kh * NT + ts][i];
          }
      }
    }
  }
}

template <int KH, int KW, bool MODE3D, int R, int TH>
__global__ __launch_bounds__(256) void wgrad_lds_kernel(const WgradLdsParams p) {
  wgrad_lds_body<KH, KW, MODE3D, R, TH>(p, VBlock{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y, (int)gridDim.z});
}

// The weight gradients of SEVERAL layers in one launch (the ~20 small layers of a cfg3 step paid 12-15 us of
// fixed cost each as launches of their own).  The job table travels by value (kernel arguments, <= 4 KB): block b belongs to the
// job j with first[j] <= b < first[j+1] and takes the place (bx, by, bz) of that job's own grid.
constexpr int kWgradBatchMax = 22;
struct WgradBatch {
  int njobs;
  int zfast;     // block order inside a job: 1 = the gz blocks of one (bx, by) next to each other and on ONE XCD (below), 0 = z slowest
  int first[kWgradBatchMax + 1];
  int gx[kWgradBatchMax], gy[kWgradBatchMax];
  WgradLdsParams job[kWgradBatchMax];
};
static_assert(sizeof(WgradBatch) <= 4096, "the job table is a kernel argument");

template <int KH, int KW, bool MODE3D, int R, int TH>
__global__ __launch_bounds__(256) void wgrad_lds_batch_kernel(const WgradBatch tb) {
  const int b = blockIdx.x;
  int j = 0;
  while (j + 1 < tb.njobs && b >= tb.first[j + 1]) ++j;           // (wave-uniform: scalar unit)
  const int local = b - tb.first[j];
  const int gx = tb.gx[j], gy = tb.gy[j];
  const int G = gx * gy;
  const int gz = (tb.first[j + 1] - tb.first[j]) / G;
  if (!tb.zfast) {
    wgrad_lds_body<KH, KW, MODE3D, R, TH>(tb.job[j], VBlock{local % gx, (local / gx) % gy, local / G, gx, gy, gz});
    return;
  }
  // The gz blocks of one (bx, by) -- the kd (3-D) / kh (2-D) slices of the same tiles -- stage the SAME `small` segments and
  // overlapping `big` patches.  With z slowest they ran a third of the launch apart and every re-read came from the fabric (PMC:
  // 6.3 GB per cfg3 step for 2.3 algorithmic); here they are issued 8 block ids apart -- workgroups go round-robin over the 8 XCDs,
  // so on the SAME XCD and at the same time: the second and third read hit that XCD's L2.  Groups of 8 (bx, by) x gz; the tail
  // (G % 8 blocks) keeps z fastest without the XCD alignment.  A pure renumbering: every block computes what it computed before.
  const int full = G & ~7;
  int bxy, bz;
  if (local < full * gz) {
    const int t = local / (8 * gz), w = local - t * 8 * gz;
    bz = w >> 3;
    bxy = t * 8 + (w & 7);
  } else {
    const int v = local - full * gz, m = G - full;
    bz = v / m;
    bxy = full + v - bz * m;
  }
  wgrad_lds_body<KH, KW, MODE3D, R, TH>(tb.job[j], VBlock{bxy % gx, bxy / gx, bz, gx, gy, gz});
}

// voxel stride (floats) such that q and q+1 (voxels `step` apart) land 16 banks apart: stride*step = 16 (mod 32)
int padded_stride(int C, int step) {
  if (C <= 8) return C;                         // 8 (or 4) valid lanes per voxel: q groups cannot collide within 32 banks
  for (int S = C; S < C + 32; S += 4)
    if (((S * step) & 31) == 16) return S;
  return C + 4;
}

// ---- deferred launches (mdf_wgrad_batch_begin / _flush): the calling thread's dispatches are recorded instead of launched
struct PendingJob { int key; WgradLdsParams p; int gx, gy, gz; size_t lds; };
thread_local bool g_batching = false;
thread_local std::vector<PendingJob>* g_pending = nullptr;

template <int KH, int KW, bool MODE3D, int R, int TH>
int launch_batch(const WgradBatch& tb, int blocks, size_t lds, hipStream_t st) {
  static bool attr_done[64] = {};
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  if (dev_id < 0 || dev_id >= 64 || !attr_done[dev_id]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_lds_batch_kernel<KH, KW, MODE3D, R, TH>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return mdf::fail(MDF_EHIP, "hipFuncSetAttribute(dynamic LDS): %s", hipGetErrorString(e));
    if (dev_id >= 0 && dev_id < 64) attr_done[dev_id] = true;
  }
  hipLaunchKernelGGL((wgrad_lds_batch_kernel<KH, KW, MODE3D, R, TH>), dim3(blocks), dim3(256), lds, st, tb);
  return mdf::check_launch("wgrad_lds_batch_kernel");
}

// key = ((KH * 8 + KW) * 2 + MODE3D) * 4 + R) * 8 + TH
constexpr int wg_key(int kh, int kw, bool m3, int r, int th) { return ((((kh * 8 + kw) * 2 + (m3 ? 1 : 0)) * 4 + r) * 8) + th; }

int launch_batch_key(int key, const WgradBatch& tb, int blocks, size_t lds, hipStream_t st) {
#define WGB(KHv, KWv, M3, Rv, THv) if (key == wg_key(KHv, KWv, M3, Rv, THv)) return launch_batch<KHv, KWv, M3, Rv, THv>(tb, blocks, lds, st);
#define WGB3(KHv, KWv, M3, Rv) WGB(KHv, KWv, M3, Rv, 1) WGB(KHv, KWv, M3, Rv, 2) WGB(KHv, KWv, M3, Rv, 4)
  WGB3(3, 3, true, 2) WGB3(3, 3, true, 1) WGB3(3, 3, true, 0) WGB3(1, 3, false, 2) WGB3(1, 3, false, 1) WGB3(1, 3, false, 0) WGB3(1, 5, false, 0) WGB3(1, 1, false, 0)
#undef WGB3
#undef WGB
  return mdf::fail(MDF_EUNSUPPORTED, "wgrad batch: no kernel for key %d", key);
}

}  // namespace

extern "C" int mdf_wgrad_batch_begin(void) {
  if (!g_pending) g_pending = new std::vector<PendingJob>();
  g_pending->clear();
  g_batching = true;
  return MDF_OK;
}

extern "C" int mdf_wgrad_batch_flush(void* stream) {
  g_batching = false;
  if (!g_pending || g_pending->empty()) return MDF_OK;
  std::vector<PendingJob> jobs;
  jobs.swap(*g_pending);
  std::vector<char> done(jobs.size(), 0);
  for (size_t i = 0; i < jobs.size(); ++i) {
    if (done[i]) continue;
    WgradBatch tb{};
    tb.zfast = [] { const char* e = getenv("MDF_WGRAD_ZFAST"); return e ? atoi(e) : 1; }();   // dev A/B (read per flush)
    int blocks = 0;
    size_t lds = 0;
    for (size_t k = i; k < jobs.size() && tb.njobs < kWgradBatchMax; ++k) {
      if (done[k] || jobs[k].key != jobs[i].key) continue;
      const PendingJob& jb = jobs[k];
      tb.first[tb.njobs] = blocks;
      tb.gx[tb.njobs] = jb.gx; tb.gy[tb.njobs] = jb.gy;
      tb.job[tb.njobs] = jb.p;
      blocks += jb.gx * jb.gy * jb.gz;
      if (jb.lds > lds) lds = jb.lds;
      ++tb.njobs;
      done[k] = 1;
    }
    tb.first[tb.njobs] = blocks;
    if (int rc = launch_batch_key(jobs[i].key, tb, blocks, lds, (hipStream_t)stream)) return rc;
  }
  return MDF_OK;
}

// returns MDF_EUNSUPPORTED when the shape has no LDS instantiation (the caller falls back to the direct kernels)
int mdf_wgrad_lds_dispatch(const float* small_, const float* big, float* workspace, int* gx_io, int B, int Ds, int Hs, int Ws, int A, int Bc,
                           int stride, int ksize, int is3d, float* zero_out, int zero_n, void* stream) {
  if ((A & 3) || (Bc & 3)) return MDF_EUNSUPPORTED;
  WgradLdsParams p{};
  p.small_ = small_; p.big = big; p.slab = workspace; p.zero_out = zero_out; p.zero_n = zero_n;
  p.B = B; p.Ds = Ds; p.Hs = Hs; p.Ws = Ws; p.Db = Ds * stride; p.Hb = Hs * stride; p.Wb = Ws * stride; p.A = A; p.Bc = Bc; p.stride = stride;
  if (!is3d) { p.Ds = 1; p.Db = 1; }
  p.pad = (ksize - 1) / 2;
  p.AS = padded_stride(A, 1);
  p.BS = padded_stride(Bc, stride);
  const int NA = (A + 15) / 16;
  p.NB = (Bc + 15) / 16;
  const int pairs = NA * p.NB;
  p.split = pairs >= 4 ? 1 : (pairs == 2 ? 2 : 4);
  const int gy = (pairs * p.split + 3) / 4;
  const int KH = is3d ? 3 : 1, KW = ksize;
  p.ntaps_total = is3d ? 27 : ksize * ksize;
  // tap packing for the few-channel layers (see the kernel): shifts of `small` in the rows, of `big` in the columns
  static const bool pack_on = [] { const char* e = getenv("MDF_WGRAD_PACK"); return e ? atoi(e) != 0 : true; }();   // dev A/B
  p.pa = p.pb = 1;
  if (pack_on && stride == 1 && ksize == 3) {
    if (A <= 8) p.pa = 2;
    if (Bc <= 8) p.pb = 2;
  }
  const int R = (p.pa - 1) + (p.pb - 1), hal = p.pa - 1;
  // tile = TH rows x tv voxels.  tv: a multiple of 16 voxels, as wide as the register staging and ~72 KiB of LDS allow (more MFMAs
  // per barrier pair), chosen to waste the fewest voxels of the row's last tile; TH in {1, 2, 4} rows where rows are short
  // (3-D with TH > 1: stride 1 only -- the staged `big` rows of neighbouring output rows must be neighbours).
  static const int th_max = [] { const char* e = getenv("MDF_WGRAD_TH"); return (e && atoi(e) > 0) ? atoi(e) : 2; }();   // dev A/B (4 rows measured slower than 2 on every cfg3 shape: fewer, fatter tiles starve the chip)
  int best_tv = 0, best_th = 1;
  long long best_cost = 0;
  size_t best_lds = 0;
  for (int th = 1; th <= th_max; th *= 2) {
    if (th > 1 && (!is3d || stride != 1)) break;     // (2-D: two rows share no staged data, and measured slower on every trunk shape)
    if (th > 1 && th > Hs) break;
    const int ns = (th == 1) ? 4 : 2, npr = (th == 1) ? (is3d ? 4 : 5) : (th == 2 ? 4 : 2);
    const int nbr = is3d ? th + KH - 1 : th;
    // (unpacked layers: multiples of 16*split up to 256, every wave the same number of chunks -- the A/B-tuned choice; packed layers
    // need Ws + 1 voxels covered, which would cost a whole extra tile at those widths: any multiple of 16 up to 512)
    const int step = (R > 0 || th > 1) ? 16 : 16 * p.split;
    for (int tv = step; tv <= (R > 0 ? 512 : 256); tv += step) {
      const int WB = tv * stride + KW - 1;
      if ((tv + hal) * (A / 4) > ns * 256 || WB * (Bc / 4) > npr * 256) break;    // register staging capacity
      const size_t lds = (size_t)(th * (tv + hal) * p.AS + (nbr * WB + 1) * p.BS) * sizeof(float);
      static const int lds_kb = [] { const char* e = getenv("MDF_WGRAD_LDS_KB"); return (e && atoi(e) > 0) ? atoi(e) : 0; }();   // dev A/B
      if (lds > (size_t)(lds_kb ? lds_kb : (th == 1 ? 72 : 80)) * 1024) break;   // two blocks per CU
      const int rounds = (th * (tv / 16) + p.split - 1) / p.split;                 // chunks of the busiest wave
      const long long tiles = (long long)((Hs + th - 1) / th) * ((Ws + hal + tv - 1) / tv);
      const long long cost = tiles * (rounds * p.split * 16 + 24);                 // + ~24 voxel-times of fixed cost per tile
      if (best_tv == 0 || cost < best_cost || (cost == best_cost && th == best_th)) { best_tv = tv; best_th = th; best_cost = cost; best_lds = lds; }
    }
  }
  if (best_tv == 0) return MDF_EUNSUPPORTED;
  p.tv = best_tv;
  const int TH = best_th;
  p.wtiles = (Ws + hal + p.tv - 1) / p.tv;      // (packed rows sum over v - sa: the tiles reach pa-1 voxels past the row)
  p.htiles = (Hs + TH - 1) / TH;
  p.n_tiles = (long long)B * p.Ds * p.htiles * p.wtiles;
  size_t lds = best_lds;
  const size_t red = (size_t)2 * KH * KW * 4 * 64 * sizeof(float);
  if (lds < red) lds = red;
  if (lds > 150 * 1024) return MDF_EUNSUPPORTED;
  int gx = *gx_io;                               // in: slabs the workspace holds; out: blocks launched (= slabs written)
  if (gx > p.n_tiles) gx = (int)p.n_tiles;
  if (gx > p.n_tiles / 2 && p.n_tiles / 2 >= 64) gx = (int)(p.n_tiles / 2);     // >= 2 tiles per block: the second tile's loads fly during the first one's MFMAs
  if (gx < 1) gx = 1;
  *gx_io = gx;
  const dim3 grid(gx, gy, is3d ? 3 : ksize);
  hipStream_t st = (hipStream_t)stream;
#define WG_LAUNCH_TH(KHv, KWv, M3, Rv, THv)                                                                               \
  {                                                                                                                        \
    if (g_batching) {       /* recorded: launched by mdf_wgrad_batch_flush together with the other layers of this instantiation */ \
      g_pending->push_back(PendingJob{wg_key(KHv, KWv, M3, Rv, THv), p, (int)grid.x, (int)grid.y, (int)grid.z, lds});       \
      return MDF_OK;                                                                                                       \
    }                                                                                                                      \
    static bool attr_done[64] = {};                                                                                        \
    int dev_id = 0;                                                                                                        \
    (void)hipGetDevice(&dev_id);                                                                                           \
    if (dev_id < 0 || dev_id >= 64 || !attr_done[dev_id]) {                                                                \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_lds_kernel<KHv, KWv, M3, Rv, THv>),          \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);                          \
      if (e != hipSuccess) return mdf::fail(MDF_EHIP, "hipFuncSetAttribute(dynamic LDS): %s", hipGetErrorString(e));       \
      if (dev_id >= 0 && dev_id < 64) attr_done[dev_id] = true;                                                            \
    }                                                                                                                      \
    hipLaunchKernelGGL((wgrad_lds_kernel<KHv, KWv, M3, Rv, THv>), grid, dim3(256), lds, st, p);                             \
    return mdf::check_launch("wgrad_lds_kernel");                                                                          \
  }
#define WG_LAUNCH(KHv, KWv, M3, Rv)                  \
  {                                                  \
    if (TH == 4) WG_LAUNCH_TH(KHv, KWv, M3, Rv, 4)   \
    if (TH == 2) WG_LAUNCH_TH(KHv, KWv, M3, Rv, 2)   \
    WG_LAUNCH_TH(KHv, KWv, M3, Rv, 1)                \
  }
  if (is3d && ksize == 3 && R == 2) WG_LAUNCH(3, 3, true, 2)
  if (is3d && ksize == 3 && R == 1) WG_LAUNCH(3, 3, true, 1)
  if (is3d && ksize == 3) WG_LAUNCH(3, 3, true, 0)
  if (!is3d && ksize == 3 && R == 2) WG_LAUNCH(1, 3, false, 2)
  if (!is3d && ksize == 3 && R == 1) WG_LAUNCH(1, 3, false, 1)
  if (!is3d && ksize == 3) WG_LAUNCH(1, 3, false, 0)
  if (!is3d && ksize == 5) WG_LAUNCH(1, 5, false, 0)
  if (!is3d && ksize == 1) WG_LAUNCH(1, 1, false, 0)
#undef WG_LAUNCH_TH
#undef WG_LAUNCH
  return MDF_EUNSUPPORTED;
}
