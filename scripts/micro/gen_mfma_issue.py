"""Generates scripts/micro/mfma_issue_gen.hip: loop bodies in explicit gfx950 assembly (fixed registers, nothing left to the
compiler's scheduler) that answer what one or two waves per SIMD can issue beside v_mfma_f32_16x16x4_f32 -- the measurements
behind the Winograd conv kernels' structure.
   python scripts/micro/gen_mfma_issue.py && hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_issue_gen.hip -o /tmp/mfma_issue && /tmp/mfma_issue
A body = 64 MFMAs in 16 groups of 4; `order` says which accumulators a group's MFMAs use, `extras` what follows each group."""
import os

ACC0, A0, B0, PK0, DS0, BUF0 = 100, 170, 174, 180, 200, 232


def mfma(acc, a, b):
    r = ACC0 + 4 * acc
    return f"v_mfma_f32_16x16x4_f32 v[{r}:{r+3}], v{A0 + a}, v{B0 + b}, v[{r}:{r+3}]"


def body(order, pk, add, ds, buf, spread=False):
    """pk/add/ds/buf: instructions per group of 4 MFMAs.  spread: place the extras between the group's MFMAs instead of behind them."""
    out = []
    n_pk = n_add = n_ds = n_buf = 0
    for g in range(16):
        if order == "dep":
            ms = [mfma(g, k, k) for k in range(4)]
        elif order == "pair":
            g0, h = g & ~1, (g & 1) * 2
            ms = [mfma(g0, h, h), mfma(g0 + 1, h, h), mfma(g0, h + 1, h + 1), mfma(g0 + 1, h + 1, h + 1)]
        elif order == "rr3":       # kd = 0,1,2 of one transform-domain element share the B operand
            j = (4 * g) // 3
            ms = [mfma((4 * g + i) % 3 + 3 * (((4 * g + i) // 3) % 5), (4 * g + i) % 3, ((4 * g + i) // 3) % 4) for i in range(4)]
        elif order == "rr4":
            ms = [mfma((g % 4) * 4 + i, i, g % 4) for i in range(4)]
        elif order == "is12":      # input-stationary ab-step: 4 k-steps x 3 kd on three accumulators, B operand shared by the 3 kd
            ms = [mfma(3 * (g % 5) + kd, (kd + s_) % 4, s_) for s_ in range(4) for kd in range(3)]
        ex = []
        for _ in range(pk):
            r = PK0 + 2 * (n_pk % 4); n_pk += 1
            ex.append(f"v_pk_add_f32 v[{r}:{r+1}], v[{r}:{r+1}], v[{PK0+8}:{PK0+9}]")
        for _ in range(add):
            r = PK0 + (n_add % 8); n_add += 1
            ex.append(f"v_add_f32 v{r}, v{r}, v{PK0+8}")
        for _ in range(ds):
            r = DS0 + 4 * (n_ds % 8); off = (n_ds % 8) * 1024; n_ds += 1
            ex.append(f"ds_read_b128 v[{r}:{r+3}], %0 offset:{off}")
        for _ in range(buf):
            r = BUF0 + 4 * (n_buf % 4); off = (n_buf % 4) * 1024; n_buf += 1
            ex.append(f"buffer_load_dwordx4 v[{r}:{r+3}], %1, %2, 0 offen offset:{off}")
        if spread:
            k = 0
            n = len(ms)
            for i, m in enumerate(ms):
                out.append(m)
                take = (len(ex) - k + (n - 1 - i)) // (n - i)
                out += ex[k:k + take]; k += take
        else:
            out += ms + ex
    out.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return out, sum(1 for l in out if l.startswith("v_mfma"))


VARIANTS = [
    ("dep, bare (4 MFMAs in a row on one accumulator: r04 order)", dict(order="dep", pk=0, add=0, ds=0, buf=0)),
    ("pair, bare (two accumulators alternate)", dict(order="pair", pk=0, add=0, ds=0, buf=0)),
    ("rr3, bare (three accumulators alternate)", dict(order="rr3", pk=0, add=0, ds=0, buf=0)),
    ("rr4, bare (four accumulators alternate)", dict(order="rr4", pk=0, add=0, ds=0, buf=0)),
    ("dep  + 4 pk_add per group", dict(order="dep", pk=4, add=0, ds=0, buf=0)),
    ("pair + 4 pk_add per group", dict(order="pair", pk=4, add=0, ds=0, buf=0)),
    ("pair + 4 pk_add per group, spread", dict(order="pair", pk=4, add=0, ds=0, buf=0, spread=True)),
    ("pair + 8 pk_add per group", dict(order="pair", pk=8, add=0, ds=0, buf=0)),
    ("pair + 8 pk_add per group, spread", dict(order="pair", pk=8, add=0, ds=0, buf=0, spread=True)),
    ("pair + 12 pk_add per group, spread", dict(order="pair", pk=12, add=0, ds=0, buf=0, spread=True)),
    ("pair + 8 v_add per group, spread", dict(order="pair", pk=0, add=8, ds=0, buf=0, spread=True)),
    ("pair + 16 v_add per group, spread", dict(order="pair", pk=0, add=16, ds=0, buf=0, spread=True)),
    ("pair + 24 v_add per group, spread", dict(order="pair", pk=0, add=24, ds=0, buf=0, spread=True)),
    ("dep  + 4 pk_add + 1 ds_read_b128 + 1 buffer_load per group (r04 mix)", dict(order="dep", pk=4, add=0, ds=1, buf=1)),
    ("pair + 4 pk_add + 1 ds_read_b128 + 1 buffer_load per group", dict(order="pair", pk=4, add=0, ds=1, buf=1)),
    ("pair + 4 pk_add + 1 ds_read_b128 + 1 buffer_load per group, spread", dict(order="pair", pk=4, add=0, ds=1, buf=1, spread=True)),
    ("rr3  + 2 pk_add + 1 buffer_load per group (input-stationary mix)", dict(order="rr3", pk=2, add=0, ds=0, buf=1, spread=True)),
    ("rr3  + 2 pk_add + 1 ds_read + 1 buffer_load per group", dict(order="rr3", pk=2, add=0, ds=1, buf=1, spread=True)),
    ("is12: 12 MFMAs + 4 pk_add + 1 ds_read + 3 buffer_load per group, spread", dict(order="is12", pk=4, add=0, ds=1, buf=3, spread=True)),
    ("is12: 12 MFMAs + 4 pk_add + 1 ds_read + 3 buffer_load per group, behind", dict(order="is12", pk=4, add=0, ds=1, buf=3)),
    ("is12: 12 MFMAs + 8 v_add + 1 ds_read + 3 buffer_load per group, spread", dict(order="is12", pk=0, add=8, ds=1, buf=3, spread=True)),
    ("is12: 12 MFMAs + 1 ds_read + 3 buffer_load per group, spread", dict(order="is12", pk=0, add=0, ds=1, buf=3, spread=True)),
    ("is12: 12 MFMAs bare", dict(order="is12", pk=0, add=0, ds=0, buf=0)),
    ("pair + 2 ds_read_b128 per group", dict(order="pair", pk=0, add=0, ds=2, buf=0, spread=True)),
    ("pair + 4 ds_read_b128 per group", dict(order="pair", pk=0, add=0, ds=4, buf=0, spread=True)),
    ("pair + 1 buffer_load per group", dict(order="pair", pk=0, add=0, ds=0, buf=1, spread=True)),
    ("pair + 2 buffer_load per group", dict(order="pair", pk=0, add=0, ds=0, buf=2, spread=True)),
]

NM = []
clob = ", ".join(f'"v{r}"' for r in range(ACC0, 248))
src = ['// GENERATED by scripts/micro/gen_mfma_issue.py -- do not edit', '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <vector>', '#include <algorithm>',
       'typedef float f32x4 __attribute__((ext_vector_type(4)));', '']
for i, (name, kw) in enumerate(VARIANTS):
    lines, nm = body(**kw)
    NM.append(nm)
    asm = "\n".join(f'      "{l}\\n"' for l in lines)
    src.append(f'''__global__ __launch_bounds__(512) void k{i}(const float* __restrict__ w, float* out, long long* cyc, int iters) {{
  __shared__ __attribute__((aligned(16))) float lds[8192 + 2048];
  for (int j = threadIdx.x; j < 8192 + 2048; j += blockDim.x) lds[j] = (float)((j * 37) % 101) * 0.01f - 0.5f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, 1 << 20, 0x00020000);
  const unsigned lp = (unsigned)(size_t)(lds + lane * 4), vo = lane * 16;
  float init[8];
  for (int j = 0; j < 8; ++j) init[j] = w[lane + 64 * j];
  asm volatile("v_mov_b32 v170, %0\\nv_mov_b32 v171, %1\\nv_mov_b32 v172, %2\\nv_mov_b32 v173, %3\\nv_mov_b32 v174, %4\\nv_mov_b32 v175, %5\\nv_mov_b32 v176, %6\\nv_mov_b32 v177, %7\\n"
               :: "v"(init[0]), "v"(init[1]), "v"(init[2]), "v"(init[3]), "v"(init[4]), "v"(init[5]), "v"(init[6]), "v"(init[7]) : {clob});
  for (int r = {ACC0}; r < {ACC0} + 64; ++r) {{}}
  asm volatile(
''' + "\n".join(f'      "v_mov_b32 v{r}, 0\\n"' for r in list(range(ACC0, ACC0 + 64)) + list(range(PK0, PK0 + 10))) + f'''
      ::: {clob});
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {{
    asm volatile(
{asm}
      :: "v"(lp), "v"(vo), "s"(rs) : "memory", {clob});
  }}
  const long long t1 = __builtin_readcyclecounter();
  float r0;
  asm volatile("v_add_f32 %0, v{ACC0}, v{ACC0 + 5}\\nv_add_f32 %0, %0, v{PK0}\\nv_add_f32 %0, %0, v{DS0}\\nv_add_f32 %0, %0, v{BUF0}\\n" : "=v"(r0) :: {clob});
  if (r0 == 12345.678f) out[0] = r0;
  if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}}
''')
src.append('''typedef void (*kern_t)(const float*, float*, long long*, int);
static void run(const char* name, kern_t k, int nm, const float* w, float* out, long long* cyc, int threads) {
  const int iters = 2000 * 64 / nm, blocks = 256, waves = threads / 64;
  (void)hipMemset(cyc, 0, blocks * 8 * sizeof(long long));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, w, out, cyc, 300);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, w, out, cyc, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * 8);
  (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  std::vector<long long> v;
  for (int b = 0; b < blocks; ++b) for (int q = 0; q < waves; ++q) v.push_back(h[b * 8 + q]);
  std::sort(v.begin(), v.end());
  const double med = (double)v[v.size() / 2];
  const double per_wave = med / (iters * (double)nm), per_simd = per_wave / (waves / 4);
  printf("%-72s %d w/SIMD: %6.1f cyc/MFMA per wave, %5.1f per SIMD (pipe %3.0f%% busy), %6.1f TFLOP/s, %.2f GHz\\n", name, waves / 4, per_wave, per_simd,
         3200.0 / per_simd, 256.0 * waves * iters * nm * 2048.0 / (ms * 1e-3) / 1e12, med / (ms * 1e-3) / 1e9);
}
int main() {
  float* w; float* out; long long* cyc;
  (void)hipMalloc(&w, 1 << 20); (void)hipMalloc(&out, 64); (void)hipMalloc(&cyc, 256 * 8 * sizeof(long long));
  std::vector<float> hw((1 << 20) / 4);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  (void)hipMemcpy(w, hw.data(), 1 << 20, hipMemcpyHostToDevice);
  for (int threads : {256, 512}) {''')
for i, (name, _) in enumerate(VARIANTS):
    src.append(f'    run("{name}", k{i}, {NM[i]}, w, out, cyc, threads);')
src.append('  }\n  return 0;\n}')
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mfma_issue_gen.hip")
open(path, "w").write("\n".join(src) + "\n")
print("wrote", path)
