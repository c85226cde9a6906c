// Training-mode BatchNorm3d + ReLU (+ residual) around the regulariser's conv layers (net/unit/base.py:50-68 with
// nn.BatchNorm3d in batch-statistics mode, net/unit/regular.py:17-41,82-108), channels-last activations [N, C]
// (N = B*D*H*W voxels, C in {8,16,32,64}).  All kernels are HBM-bound streams:
//
//   bn_stats          sum, sum of squares per channel                                  reads y once
//   bn_finalize       mean/var -> (a, b, mean, invstd), running-stat update            C threads
//   bn_relu_apply     z = [res +] relu(y*a + b)                                        reads y (+res), writes z
//   bn_relu_bwd_reduce  S1 = sum dr, S2 = sum dr*xhat, dr = dz*[y*a+b > 0]             reads dz, y
//   bn_relu_bwd       dy = gamma*invstd*(dr - S1/N - xhat*S2/N); dgamma = S2, dbeta = S1   reads dz, y, writes dy
// Every entry takes `ngroups`: the tensor is [ngroups][N][C] and each group is normalised on its own (the feature pyramid
// is called once per VIEW in training, net/core.py:42, so a batched pass over all views has one group per view; blockIdx.y).
//
// A block strides over the tensor in steps that are multiples of C floats, so a thread owns the same 4 channels for
// the whole kernel: per-thread fp32 partial sums (a few dozen elements each), combined in fp64 through LDS and one
// fp64 atomic per channel per block.
#include "common.h"

namespace {

constexpr int kT = 256;

__device__ __forceinline__ void lds_add(double* p, double v) { atomicAdd(p, v); }

// The sums may arrive spread over `nslices` copies [slice][group][2C] (the conv epilogues of conv3d.hip / conv_lds.hip send
// them that way to keep same-address atomics few): total of entry i of group g.
__device__ __forceinline__ double sliced(const double* __restrict__ sums, int g, int i, int C, int ngroups, int nslices) {
  double t = 0.0;
  for (int s = 0; s < nslices; ++s) t += sums[((long long)s * ngroups + g) * 2 * C + i];
  return t;
}

// The same for a whole block: totals of group g's 2C entries into LDS `tot` (2C <= 128), every thread issuing its share of the
// nslices*2C loads at once (one L2 round trip instead of nslices dependent ones per value).  Ends with a barrier.
__device__ __forceinline__ void sliced_block(const double* __restrict__ sums, int g, int C, int ngroups, int nslices, double* tot) {
  const int tid = threadIdx.x, n2 = 2 * C;
  __syncthreads();             // (a previous use of `tot` by the caller's loop is over)
  if (tid < n2) tot[tid] = 0.0;
  __syncthreads();
  if (nslices == 1) {
    if (tid < n2) tot[tid] = sums[(long long)g * n2 + tid];
  } else {
    for (int i = tid; i < nslices * n2; i += kT) {
      const int s = i / n2, e = i - s * n2;
      const double v = sums[((long long)s * ngroups + g) * n2 + e];
      if (v != 0.0) lds_add(&tot[e], v);
    }
  }
  __syncthreads();
}

template <bool BWD>
__global__ __launch_bounds__(kT) void bn_reduce_kernel(const float* __restrict__ y, const float* __restrict__ dz,
                                                       const float* __restrict__ aux, long long n4, int C,
                                                       double* __restrict__ out) {
  // FWD: out[c] += sum y, out[C+c] += sum y^2.   BWD: out[c] += sum dr, out[C+c] += sum dr*xhat.   blockIdx.y = group
  __shared__ double sm[128];
  const int tid = threadIdx.x;
  y += (long long)blockIdx.y * n4 * 4;
  if (BWD) { dz += (long long)blockIdx.y * n4 * 4; aux += (long long)blockIdx.y * 4 * C; }
  out += (long long)blockIdx.y * 2 * C;
  if (tid < 2 * C) sm[tid] = 0.0;
  __syncthreads();
  const int c0 = (4 * tid) % C;
  float a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0}, mu[4] = {0, 0, 0, 0}, is[4] = {0, 0, 0, 0};
  if (BWD) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] = aux[c0 + k]; b[k] = aux[C + c0 + k]; mu[k] = aux[2 * C + c0 + k]; is[k] = aux[3 * C + c0 + k]; }
  }
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  const long long stride = (long long)gridDim.x * kT;
  for (long long i = (long long)blockIdx.x * kT + tid; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(y)[i];
    const float yv[4] = {v.x, v.y, v.z, v.w};
    if (!BWD) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { s1[k] += yv[k]; s2[k] = fmaf(yv[k], yv[k], s2[k]); }
    } else {
      const float4 g = reinterpret_cast<const float4*>(dz)[i];
      const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dr = (fmaf(yv[k], a[k], b[k]) > 0.0f) ? gv[k] : 0.0f;
        s1[k] += dr;
        s2[k] = fmaf(dr, (yv[k] - mu[k]) * is[k], s2[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { lds_add(&sm[c0 + k], (double)s1[k]); lds_add(&sm[C + c0 + k], (double)s2[k]); }
  __syncthreads();
  if (tid < 2 * C) atomicAdd(&out[tid], sm[tid]);
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float eps, float momentum, double n, int C, int ngroups, float* __restrict__ aux,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, long long* __restrict__ nbt, int nslices) {
  const int c = threadIdx.x;
  if (c < C) {
    for (int g = 0; g < ngroups; ++g) {     // groups = successive calls of the module: the running statistics see them in order
      float* a4 = aux + (long long)g * 4 * C;
      const double mean = sliced(sums, g, c, C, ngroups, nslices) / n;
      double var = sliced(sums, g, C + c, C, ngroups, nslices) / n - mean * mean;
      if (var < 0.0) var = 0.0;
      const float invstd = (float)(1.0 / sqrt(var + (double)eps));
      const float a = gamma[c] * invstd;
      a4[c] = a;
      a4[C + c] = beta[c] - (float)mean * a;
      a4[2 * C + c] = (float)mean;
      a4[3 * C + c] = invstd;
      if (running_mean) {   // nn.BatchNorm: running = (1-m)*running + m*batch, variance unbiased
        const double unb = (n > 1.0) ? var * n / (n - 1.0) : var;
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unb;
      }
    }
  }
  if (c == 0 && nbt) *nbt += ngroups;
}

__global__ __launch_bounds__(kT) void bn_relu_apply_kernel(const float* __restrict__ y, const float* __restrict__ aux,
                                                           const float* __restrict__ res, float* __restrict__ z, long long n4, int C) {
  const int tid = threadIdx.x;
  const int c0 = (4 * tid) % C;
  y += (long long)blockIdx.y * n4 * 4;
  z += (long long)blockIdx.y * n4 * 4;
  if (res) res += (long long)blockIdx.y * n4 * 4;
  aux += (long long)blockIdx.y * 4 * C;
  float a[4], b[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { a[k] = aux[c0 + k]; b[k] = aux[C + c0 + k]; }
  const long long stride = (long long)gridDim.x * kT;
  for (long long i = (long long)blockIdx.x * kT + tid; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(y)[i];
    float4 o;
    o.x = fmaxf(fmaf(v.x, a[0], b[0]), 0.0f);
    o.y = fmaxf(fmaf(v.y, a[1], b[1]), 0.0f);
    o.z = fmaxf(fmaf(v.z, a[2], b[2]), 0.0f);
    o.w = fmaxf(fmaf(v.w, a[3], b[3]), 0.0f);
    if (res) {
      const float4 r = reinterpret_cast<const float4*>(res)[i];
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    reinterpret_cast<float4*>(z)[i] = o;
  }
}

// finalize + apply in one launch: every block derives (a, b) of its 4 channels from the statistics itself (a few fp64
// operations), block 0 of each group also publishes aux for the backward pass, block (0,0) updates the running statistics
// of all groups in call order.
__global__ __launch_bounds__(kT) void bn_finalize_apply_kernel(const float* __restrict__ y, const double* __restrict__ sums,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                               float momentum, double n, const float* __restrict__ res,
                                                               float* __restrict__ z, float* __restrict__ aux,
                                                               float* __restrict__ running_mean, float* __restrict__ running_var,
                                                               long long* __restrict__ nbt, long long n4, int C, int ngroups, int nslices) {
  const int tid = threadIdx.x, g = blockIdx.y;
  const int c0 = (4 * tid) % C;
  __shared__ double tot[128];
  sliced_block(sums, g, C, ngroups, nslices, tot);
  float a[4], b[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double mean = tot[c0 + k] / n;
    double var = tot[C + c0 + k] / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    a[k] = gamma[c0 + k] * invstd;
    b[k] = beta[c0 + k] - (float)mean * a[k];
    if (blockIdx.x == 0 && 4 * tid < C) {      // threads 0..C/4-1 cover every channel once
      float* a4 = aux + (long long)g * 4 * C;
      a4[c0 + k] = a[k]; a4[C + c0 + k] = b[k]; a4[2 * C + c0 + k] = (float)mean; a4[3 * C + c0 + k] = invstd;
    }
  }
  if (blockIdx.x == 0 && g == 0 && running_mean) {      // (block-uniform)
    __shared__ double tg[128];
    for (int gg = 0; gg < ngroups; ++gg) {      // groups = successive calls of the module
      sliced_block(sums, gg, C, ngroups, nslices, tg);
      if (tid < C) {
        const double mean = tg[tid] / n;
        double var = tg[C + tid] / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const double unb = (n > 1.0) ? var * n / (n - 1.0) : var;
        running_mean[tid] = (1.0f - momentum) * running_mean[tid] + momentum * (float)mean;
        running_var[tid] = (1.0f - momentum) * running_var[tid] + momentum * (float)unb;
      }
    }
    if (tid == 0 && nbt) *nbt += ngroups;
  }
  y += (long long)g * n4 * 4;
  z += (long long)g * n4 * 4;
  if (res) res += (long long)g * n4 * 4;
  const long long stride = (long long)gridDim.x * kT;
  for (long long i = (long long)blockIdx.x * kT + tid; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(y)[i];
    float4 o;
    o.x = fmaxf(fmaf(v.x, a[0], b[0]), 0.0f);
    o.y = fmaxf(fmaf(v.y, a[1], b[1]), 0.0f);
    o.z = fmaxf(fmaf(v.z, a[2], b[2]), 0.0f);
    o.w = fmaxf(fmaf(v.w, a[3], b[3]), 0.0f);
    if (res) {
      const float4 r = reinterpret_cast<const float4*>(res)[i];
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    reinterpret_cast<float4*>(z)[i] = o;
  }
}

__global__ __launch_bounds__(kT) void bn_relu_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                         const float* __restrict__ aux, const double* __restrict__ red,
                                                         const float* __restrict__ gamma, double inv_n, float* __restrict__ dy,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, long long n4, int C, int nslices) {
  const int tid = threadIdx.x;
  const int c0 = (4 * tid) % C;
  const int ngroups = (int)gridDim.y;
  dz += (long long)blockIdx.y * n4 * 4;
  y += (long long)blockIdx.y * n4 * 4;
  dy += (long long)blockIdx.y * n4 * 4;
  aux += (long long)blockIdx.y * 4 * C;
  if (blockIdx.x == 0 && blockIdx.y == 0) {   // parameter gradients: the groups are calls of ONE module, their gradients add (block-uniform)
    __shared__ double tg[128];
    double sb = 0.0, sg = 0.0;
    for (int g = 0; g < ngroups; ++g) {
      sliced_block(red, g, C, ngroups, nslices, tg);
      if (tid < C) { sb += tg[tid]; sg += tg[C + tid]; }
    }
    if (tid < C) {
      dbeta[tid] = (float)sb;
      dgamma[tid] = (float)sg;
    }
  }
  __shared__ double tot[128];
  sliced_block(red, (int)blockIdx.y, C, ngroups, nslices, tot);
  float a[4], b[4], mu[4], is[4], m1[4], m2[4], gi[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = aux[c0 + k]; b[k] = aux[C + c0 + k]; mu[k] = aux[2 * C + c0 + k]; is[k] = aux[3 * C + c0 + k];
    m1[k] = (float)(tot[c0 + k] * inv_n);
    m2[k] = (float)(tot[C + c0 + k] * inv_n);
    gi[k] = gamma[c0 + k] * is[k];
  }
  const long long stride = (long long)gridDim.x * kT;
  for (long long i = (long long)blockIdx.x * kT + tid; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(y)[i];
    const float4 g = reinterpret_cast<const float4*>(dz)[i];
    const float yv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {g.x, g.y, g.z, g.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float dr = (fmaf(yv[k], a[k], b[k]) > 0.0f) ? gv[k] : 0.0f;
      const float xh = (yv[k] - mu[k]) * is[k];
      o[k] = gi[k] * (dr - m1[k] - xh * m2[k]);
    }
    reinterpret_cast<float4*>(dy)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// reductions: every block ends with 2C fp64 atomics on the same 2C addresses -- few, fat blocks
int grid_for_reduce(long long n4, int ngroups) {
  long long g = (n4 + 4 * kT - 1) / (4 * kT);
  const long long cap = 512 / ngroups > 32 ? 512 / ngroups : 32;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

int grid_for(long long n4, int ngroups) {
  long long g = (n4 + kT - 1) / kT;
  const long long cap = 2048 / ngroups > 64 ? 2048 / ngroups : 64;   // 8 blocks per CU: enough bytes in flight for an HBM stream
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

int check_bn(const void* y, long long N, int C) {
  MDF_REQUIRE(y, "null pointer argument");
  MDF_REQUIRE(N > 0, "bad voxel count");
  MDF_REQUIRE(C == 8 || C == 16 || C == 32 || C == 64, "C=%d not in {8,16,32,64}", C);
  return MDF_OK;
}

}  // namespace

extern "C" int mdf_bn_stats_fwd(const float* y, long long N, int C, int ngroups, double* sums, void* stream) {
  if (int rc = check_bn(y, N, C)) return rc;
  MDF_REQUIRE(sums && ngroups >= 1 && ngroups <= 65535, "bad argument");
  const long long n4 = N * C / 4;
  hipLaunchKernelGGL(bn_reduce_kernel<false>, dim3(grid_for_reduce(n4, ngroups), ngroups), dim3(kT), 0, (hipStream_t)stream, y, nullptr, nullptr, n4, C, sums);
  return mdf::check_launch("bn_stats_kernel");
}

extern "C" int mdf_bn_finalize_fwd(const double* sums, const float* gamma, const float* beta, float eps, float momentum,
                                   long long N, int C, int ngroups, float* aux, float* running_mean, float* running_var,
                                   long long* num_batches_tracked, int nslices, void* stream) {
  MDF_REQUIRE(sums && gamma && beta && aux, "null pointer argument");
  MDF_REQUIRE(C >= 1 && C <= 64 && N > 0 && ngroups >= 1 && nslices >= 1, "bad shape");
  MDF_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running_mean and running_var go together");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, gamma, beta, eps, momentum, (double)N, C, ngroups, aux,
                     running_mean, running_var, num_batches_tracked, nslices);
  return mdf::check_launch("bn_finalize_kernel");
}

extern "C" int mdf_bn_relu_apply_fwd(const float* y, const float* aux, const float* res, float* z, long long N, int C, int ngroups,
                                     void* stream) {
  if (int rc = check_bn(y, N, C)) return rc;
  MDF_REQUIRE(aux && z && ngroups >= 1 && ngroups <= 65535, "bad argument");
  const long long n4 = N * C / 4;
  hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(grid_for(n4, ngroups), ngroups), dim3(kT), 0, (hipStream_t)stream, y, aux, res, z, n4, C);
  return mdf::check_launch("bn_relu_apply_kernel");
}

extern "C" int mdf_bn_relu_bwd_reduce(const float* dz, const float* y, const float* aux, long long N, int C, int ngroups, double* red,
                                      void* stream) {
  if (int rc = check_bn(y, N, C)) return rc;
  MDF_REQUIRE(dz && aux && red && ngroups >= 1 && ngroups <= 65535, "bad argument");
  const long long n4 = N * C / 4;
  hipLaunchKernelGGL(bn_reduce_kernel<true>, dim3(grid_for_reduce(n4, ngroups), ngroups), dim3(kT), 0, (hipStream_t)stream, y, dz, aux, n4, C, red);
  return mdf::check_launch("bn_relu_bwd_reduce_kernel");
}

extern "C" int mdf_bn_relu_bwd(const float* dz, const float* y, const float* aux, const double* red, const float* gamma, long long N,
                               int C, int ngroups, float* dy, float* dgamma, float* dbeta, int nslices, void* stream) {
  if (int rc = check_bn(y, N, C)) return rc;
  MDF_REQUIRE(dz && aux && red && gamma && dy && dgamma && dbeta && ngroups >= 1 && ngroups <= 65535 && nslices >= 1, "bad argument");
  const long long n4 = N * C / 4;
  hipLaunchKernelGGL(bn_relu_bwd_kernel, dim3(grid_for(n4, ngroups), ngroups), dim3(kT), 0, (hipStream_t)stream, dz, y, aux, red, gamma, 1.0 / (double)N, dy,
                     dgamma, dbeta, n4, C, nslices);
  return mdf::check_launch("bn_relu_bwd_kernel");
}

extern "C" int mdf_bn_finalize_apply_fwd(const float* y, const double* sums, const float* gamma, const float* beta, float eps, float momentum,
                                         const float* res, float* z, float* aux, float* running_mean, float* running_var,
                                         long long* num_batches_tracked, long long N, int C, int ngroups, int nslices, void* stream) {
  if (int rc = check_bn(y, N, C)) return rc;
  MDF_REQUIRE(sums && gamma && beta && z && aux && ngroups >= 1 && ngroups <= 65535 && nslices >= 1, "bad argument");
  MDF_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running_mean and running_var go together");
  const long long n4 = N * C / 4;
  hipLaunchKernelGGL(bn_finalize_apply_kernel, dim3(grid_for(n4, ngroups), ngroups), dim3(kT), 0, (hipStream_t)stream, y, sums, gamma, beta, eps,
                     momentum, (double)N, res, z, aux, running_mean, running_var, num_batches_tracked, n4, C, ngroups, nslices);
  return mdf::check_launch("bn_finalize_apply_kernel");
}
