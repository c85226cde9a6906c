// Training loss on the device (net/loss.py:10-27): sum over the output scales of smooth-L1 (beta = 1, mean) over the pixels
// with gt > depth_min.  As torch expressions this is ~55 tiny launches per step (mask, boolean-free masked mean, the same
// backwards) on a step that is bound by launch count; here: one reduction launch per scale, one finalize, one backward
// launch per scale.
#include <cmath>
#include "common.h"

namespace {

constexpr int kT = 256;

// depth_min arrives as the loader made it (float64, load/dtutrain.py) or as float32; `gt > depth_min` promotes to the wider type
__device__ __forceinline__ double load_floor(const void* p, int is_f64, long long i) {
  return is_f64 ? static_cast<const double*>(p)[i] : (double)static_cast<const float*>(p)[i];
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// acc[0] += sum of smooth_l1(e - gt) over valid pixels, acc[1] += number of valid pixels
__global__ __launch_bounds__(kT) void masked_smooth_l1_reduce_kernel(const float* __restrict__ e, const float* __restrict__ gt,
                                                                     const void* __restrict__ floor_, int floor_f64, int floor_stride,
                                                                     long long per_batch, double* __restrict__ acc) {
  __shared__ double sh[4];
  const int b = blockIdx.y;
  const double fl = load_floor(floor_, floor_f64, (long long)b * floor_stride);
  const float* eb = e + (long long)b * per_batch;
  const float* gb = gt + (long long)b * per_batch;
  double s = 0.0, c = 0.0;
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < per_batch; i += (long long)gridDim.x * kT) {
    const float g = gb[i];
    if ((double)g > fl) {
      const float d = eb[i] - g, a = fabsf(d);
      s += (double)(a < 1.0f ? 0.5f * d * d : a - 0.5f);
      c += 1.0;
    }
  }
  s = block_sum(s, sh);
  c = block_sum(c, sh);
  if (threadIdx.x == 0) {
    atomicAdd(&acc[0], s);
    atomicAdd(&acc[1], c);
  }
}

// loss = sum_s acc[2s] / acc[2s+1];  inv_count[s] = 1 / acc[2s+1] (for the backward launches)
__global__ void masked_smooth_l1_finalize_kernel(const double* __restrict__ acc, int nscales, float* __restrict__ loss, float* __restrict__ inv_count) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float total = 0.f;
  for (int s = 0; s < nscales; ++s) {
    total += (float)(acc[2 * s] / acc[2 * s + 1]);      // each scale's mean rounded to fp32, then added: torch's order (loss.py:19-25)
    inv_count[s] = (float)(1.0 / acc[2 * s + 1]);
  }
  loss[0] = total;
}

// de = dloss * inv_count * clamp(e - gt, -1, 1) on valid pixels, 0 elsewhere
__global__ __launch_bounds__(kT) void masked_smooth_l1_bwd_kernel(const float* __restrict__ e, const float* __restrict__ gt,
                                                                  const void* __restrict__ floor_, int floor_f64, int floor_stride,
                                                                  long long per_batch, const float* __restrict__ dloss,
                                                                  const float* __restrict__ inv_count, float* __restrict__ de) {
  const int b = blockIdx.y;
  const double fl = load_floor(floor_, floor_f64, (long long)b * floor_stride);
  const float scale = dloss[0] * inv_count[0];
  const long long off = (long long)b * per_batch;
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < per_batch; i += (long long)gridDim.x * kT) {
    const float g = gt[off + i];
    const float d = e[off + i] - g;
    de[off + i] = ((double)g > fl) ? scale * fminf(fmaxf(d, -1.0f), 1.0f) : 0.0f;
  }
}

// ---- all scales in one launch: the scales travel by value (<= 8), blockIdx.z is the scale ------------------------------------
constexpr int kMaxScales = 8;
struct L1Scales {
  const float* e[kMaxScales];
  const float* gt[kMaxScales];
  float* de[kMaxScales];          // backward only (NULL: this scale needs no gradient)
  long long per_batch[kMaxScales];
  int n;
};

__global__ __launch_bounds__(kT) void masked_smooth_l1_reduce_multi_kernel(L1Scales sc, const void* __restrict__ floor_, int floor_f64, int floor_stride,
                                                                           double* __restrict__ acc) {
  __shared__ double sh[4];
  const int s_ = blockIdx.z, b = blockIdx.y;
  const long long per_batch = sc.per_batch[s_];
  if ((long long)blockIdx.x * kT >= per_batch) return;             // (the grid is sized for the largest scale; whole blocks leave)
  const double fl = load_floor(floor_, floor_f64, (long long)b * floor_stride);
  const float* eb = sc.e[s_] + (long long)b * per_batch;
  const float* gb = sc.gt[s_] + (long long)b * per_batch;
  double s = 0.0, c = 0.0;
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < per_batch; i += (long long)gridDim.x * kT) {
    const float g = gb[i];
    if ((double)g > fl) {
      const float d = eb[i] - g, a = fabsf(d);
      s += (double)(a < 1.0f ? 0.5f * d * d : a - 0.5f);
      c += 1.0;
    }
  }
  s = block_sum(s, sh);
  c = block_sum(c, sh);
  if (threadIdx.x == 0) {
    atomicAdd(&acc[2 * s_], s);
    atomicAdd(&acc[2 * s_ + 1], c);
  }
}

__global__ __launch_bounds__(kT) void masked_smooth_l1_bwd_multi_kernel(L1Scales sc, const void* __restrict__ floor_, int floor_f64, int floor_stride,
                                                                        const float* __restrict__ dloss, const float* __restrict__ inv_count) {
  const int s_ = blockIdx.z, b = blockIdx.y;
  float* de = sc.de[s_];
  const long long per_batch = sc.per_batch[s_];
  if (!de || (long long)blockIdx.x * kT >= per_batch) return;
  const double fl = load_floor(floor_, floor_f64, (long long)b * floor_stride);
  const float scale = dloss[0] * inv_count[s_];
  const long long off = (long long)b * per_batch;
  const float* e = sc.e[s_];
  const float* gt = sc.gt[s_];
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < per_batch; i += (long long)gridDim.x * kT) {
    const float g = gt[off + i];
    const float d = e[off + i] - g;
    de[off + i] = ((double)g > fl) ? scale * fminf(fmaxf(d, -1.0f), 1.0f) : 0.0f;
  }
}

int grid_x(long long per_batch) {
  long long g = (per_batch + 4 * kT - 1) / (4 * kT);
  return (int)(g < 1 ? 1 : (g > 256 ? 256 : g));
}

}  // namespace

extern "C" int mdf_masked_smooth_l1_reduce(const float* est, const float* gt, const void* floor_, int floor_f64, int floor_stride, int B,
                                           long long per_batch, double* acc, void* stream) {
  MDF_REQUIRE(est && gt && floor_ && acc, "null pointer argument");
  MDF_REQUIRE(B > 0 && B <= 65535 && per_batch > 0 && floor_stride >= 0, "bad shape");
  hipLaunchKernelGGL(masked_smooth_l1_reduce_kernel, dim3(grid_x(per_batch), B), dim3(kT), 0, (hipStream_t)stream, est, gt, floor_, floor_f64,
                     floor_stride, per_batch, acc);
  return mdf::check_launch("masked_smooth_l1_reduce_kernel");
}

extern "C" int mdf_masked_smooth_l1_finalize(const double* acc, int nscales, float* loss, float* inv_count, void* stream) {
  MDF_REQUIRE(acc && loss && inv_count && nscales >= 1 && nscales <= 64, "bad argument");
  hipLaunchKernelGGL(masked_smooth_l1_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, acc, nscales, loss, inv_count);
  return mdf::check_launch("masked_smooth_l1_finalize_kernel");
}

extern "C" int mdf_masked_smooth_l1_bwd(const float* est, const float* gt, const void* floor_, int floor_f64, int floor_stride, int B,
                                        long long per_batch, const float* dloss, const float* inv_count, float* dest, void* stream) {
  MDF_REQUIRE(est && gt && floor_ && dloss && inv_count && dest, "null pointer argument");
  MDF_REQUIRE(B > 0 && B <= 65535 && per_batch > 0 && floor_stride >= 0, "bad shape");
  hipLaunchKernelGGL(masked_smooth_l1_bwd_kernel, dim3(grid_x(per_batch), B), dim3(kT), 0, (hipStream_t)stream, est, gt, floor_, floor_f64,
                     floor_stride, per_batch, dloss, inv_count, dest);
  return mdf::check_launch("masked_smooth_l1_bwd_kernel");
}

extern "C" int mdf_masked_smooth_l1_reduce_multi(const float* const* est, const float* const* gt, const long long* per_batch, int nscales,
                                                 const void* floor_, int floor_f64, int floor_stride, int B, double* acc, void* stream) {
  MDF_REQUIRE(est && gt && per_batch && floor_ && acc, "null pointer argument");
  MDF_REQUIRE(nscales >= 1 && nscales <= kMaxScales && B > 0 && B <= 65535 && floor_stride >= 0, "bad argument (at most %d scales)", kMaxScales);
  L1Scales sc{};
  sc.n = nscales;
  long long big = 0;
  for (int i = 0; i < nscales; ++i) {
    MDF_REQUIRE(est[i] && gt[i] && per_batch[i] > 0, "null pointer / empty scale %d", i);
    sc.e[i] = est[i]; sc.gt[i] = gt[i]; sc.per_batch[i] = per_batch[i];
    big = per_batch[i] > big ? per_batch[i] : big;
  }
  hipLaunchKernelGGL(masked_smooth_l1_reduce_multi_kernel, dim3(grid_x(big), B, nscales), dim3(kT), 0, (hipStream_t)stream, sc, floor_, floor_f64,
                     floor_stride, acc);
  return mdf::check_launch("masked_smooth_l1_reduce_multi_kernel");
}

extern "C" int mdf_masked_smooth_l1_bwd_multi(const float* const* est, const float* const* gt, const long long* per_batch, int nscales,
                                              const void* floor_, int floor_f64, int floor_stride, int B, const float* dloss,
                                              const float* inv_count, float* const* dest, void* stream) {
  MDF_REQUIRE(est && gt && per_batch && floor_ && dloss && inv_count && dest, "null pointer argument");
  MDF_REQUIRE(nscales >= 1 && nscales <= kMaxScales && B > 0 && B <= 65535 && floor_stride >= 0, "bad argument (at most %d scales)", kMaxScales);
  L1Scales sc{};
  sc.n = nscales;
  long long big = 0;
  for (int i = 0; i < nscales; ++i) {
    MDF_REQUIRE(est[i] && gt[i] && per_batch[i] > 0, "null pointer / empty scale %d", i);
    sc.e[i] = est[i]; sc.gt[i] = gt[i]; sc.de[i] = dest[i]; sc.per_batch[i] = per_batch[i];
    if (dest[i]) big = per_batch[i] > big ? per_batch[i] : big;
  }
  if (big == 0) return MDF_OK;       // no scale wants a gradient
  hipLaunchKernelGGL(masked_smooth_l1_bwd_multi_kernel, dim3(grid_x(big), B, nscales), dim3(kT), 0, (hipStream_t)stream, sc, floor_, floor_f64,
                     floor_stride, dloss, inv_count);
  return mdf::check_launch("masked_smooth_l1_bwd_multi_kernel");
}

// ---- Adam over all parameters in one launch (train.py:14: torch.optim.Adam(lr=1e-3), defaults otherwise) -------------------
// The parameters stay the module's own tensors (state_dict surface untouched); the gradients and both moments are flat buffers
// in parameter order (ddp.FlatBucket).  One launch over all 1.2 M elements instead of the ~20 multi-tensor launches of torch's
// foreach implementation over 158 tensors; the arithmetic follows torch/optim/adam.py:_multi_tensor_adam operation by operation.
namespace {
struct AdamJob {
  float* param;        // the tensor
  long long offset;    // its first element in the flat buffers
  int n;               // elements
  int blk0;            // first block of the tensor in the launch
};
constexpr int kAdamPerBlock = 1024;

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamJob* __restrict__ jobs, const int* __restrict__ block_job,
                                                        const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, float lr,
                                                        float beta1, float beta2, float eps, float weight_decay, float bc1, float bc2_sqrt,
                                                        const float* __restrict__ hyper) {
  const AdamJob j = jobs[block_job[blockIdx.x]];
  if (hyper) {   // a step replayed as a hipGraph: the per-step scalars come from memory, not from the (recorded) arguments
    lr = hyper[0]; bc1 = hyper[1]; bc2_sqrt = hyper[2];
  }
  const float step_size = lr / bc1;
  const int base = (blockIdx.x - j.blk0) * kAdamPerBlock;
#pragma unroll
  for (int k = 0; k < kAdamPerBlock / 256; ++k) {
    const int i = base + k * 256 + threadIdx.x;
    if (i >= j.n) break;
    const long long f = j.offset + i;
    float gi = g[f];
    const float pi = j.param[i];
    if (weight_decay != 0.0f) gi = fmaf(pi, weight_decay, gi);
    const float mi = fmaf(gi - m[f], 1.0f - beta1, m[f]);              // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = fmaf(gi * gi, 1.0f - beta2, v[f] * beta2);        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    m[f] = mi;
    v[f] = vi;
    j.param[i] = pi - step_size * (mi / denom);                        // param.addcdiv_(exp_avg, denom, value=-step_size)
  }
}
}  // namespace

extern "C" int64_t mdf_adam_job_bytes(void) { return (int64_t)sizeof(AdamJob); }

extern "C" int64_t mdf_adam_job_fill(void* jobs_host, int index, float* param, long long offset, long long n, int first_block) {
  if (!jobs_host || index < 0 || !param || offset < 0 || n < 1 || n > 0x7fffffffll) {
    mdf::set_error("mdf_adam_job_fill: bad argument");
    return MDF_EARG;
  }
  const int nblk = (int)((n + kAdamPerBlock - 1) / kAdamPerBlock);
  static_cast<AdamJob*>(jobs_host)[index] = AdamJob{param, offset, (int)n, first_block};
  return nblk;
}

extern "C" int mdf_adam_step(const void* jobs_dev, const int* block_job_dev, int nblocks, const float* grads, float* exp_avg,
                             float* exp_avg_sq, float lr, float beta1, float beta2, float eps, float weight_decay, long long step,
                             void* stream) {
  MDF_REQUIRE(jobs_dev && block_job_dev && grads && exp_avg && exp_avg_sq, "null pointer argument");
  MDF_REQUIRE(nblocks > 0 && step >= 1 && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "bad argument");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_step_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, static_cast<const AdamJob*>(jobs_dev), block_job_dev, grads,
                     exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), (const float*)nullptr);
  return mdf::check_launch("adam_step_kernel");
}

extern "C" int mdf_adam_step_hyper(const void* jobs_dev, const int* block_job_dev, int nblocks, const float* grads, float* exp_avg,
                                   float* exp_avg_sq, const float* hyper_dev, float beta1, float beta2, float eps, float weight_decay,
                                   void* stream) {
  MDF_REQUIRE(jobs_dev && block_job_dev && grads && exp_avg && exp_avg_sq && hyper_dev, "null pointer argument");
  MDF_REQUIRE(nblocks > 0 && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "bad argument");
  hipLaunchKernelGGL(adam_step_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, static_cast<const AdamJob*>(jobs_dev), block_job_dev, grads,
                     exp_avg, exp_avg_sq, 0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f, hyper_dev);
  return mdf::check_launch("adam_step_kernel");
}
