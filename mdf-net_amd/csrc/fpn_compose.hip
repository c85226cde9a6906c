// The algebra of the feature pyramid's heads in TRAINING mode (net/unit/backbone.py:59-63: lat2 / lat3 / out2 / out3 / out4, five
// 1x1 convs around two bilinear up-samplings).  1x1 convs and bilinear up-sampling are linear and commute, so the heads are
//     y4 = O4 t4,   y3 = up(O3 t4) + (O3 L3) t3 + O3 b3,   y2 = up(up(O2 t4) + (O2 L3) t3 + O2 b3) + (O2 L2) t2 + O2 b2
// and the 64-channel 1/2- and 1/4-resolution tensors are never formed (train_ops.py:FPNHeadsComposedFn).  What is left of the
// five convs besides the large-map kernels is a dozen products of matrices with 16..64 rows: as torch ops they were ~45 launches
// per training step (3 mm + 3 mv forward; zeros, mm, mv, outer, add backward) of 3.5-5 us each by the rocprof trace, whatever little they
// compute.  Here: ONE launch forward (the composed matrices and bias vectors, which the
// step's batched weight pack then reads like any parameter) and ONE launch backward (the gradients of the composed matrices
// mapped back onto the seven parameters).  Matrices are row-major [out][in] as the Conv2d weights are; sums run in index order.
#include "common.h"

namespace {

struct ComposeFwd {
  const float *O2, *O3, *L2, *b2, *L3, *b3;   // O2 [c2][cm], O3 [c3][cm], L2 [cm][c2], b2 [cm], L3 [cm][c3], b3 [cm]
  float* comp;                                // A2 [c2][c2] | B3 [c2][c3] | A3 [c3][c3] | e2 [c2] | e3 [c2] | f3 [c3]
  int c2, c3, cm;
};

// Both kernels first copy the six parameter arrays (6272 floats for 16 / 32 / 64 channels) into LDS with coalesced loads: a thread's
// 64-term dot product read straight from memory is a chain of dependent, strided loads (20 us per launch, all latency).
__device__ __forceinline__ void stage(float* dst, const float* src, int n) {
  for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void fpn_compose_fwd_kernel(ComposeFwd p) {
  extern __shared__ float lds[];
  const int c2 = p.c2, c3 = p.c3, cm = p.cm;
  float* O2 = lds;              float* O3 = O2 + c2 * cm;
  float* L2 = O3 + c3 * cm;     float* L3 = L2 + cm * c2;
  float* b2 = L3 + cm * c3;     float* b3 = b2 + cm;
  stage(O2, p.O2, c2 * cm); stage(O3, p.O3, c3 * cm); stage(L2, p.L2, cm * c2); stage(L3, p.L3, cm * c3); stage(b2, p.b2, cm); stage(b3, p.b3, cm);
  __syncthreads();
  const int nA2 = c2 * c2, nB3 = c2 * c3, nA3 = c3 * c3;
  const int total = nA2 + nB3 + nA3 + c2 + c2 + c3;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const float* row;      // a row of O2 / O3
    const float* col;      // a column of L2 / L3 (stride ld) or a bias vector (stride 1)
    int ld;
    int r = e;
    if (r < nA2) { row = O2 + (r / c2) * cm; col = L2 + (r % c2); ld = c2; }
    else if ((r -= nA2) < nB3) { row = O2 + (r / c3) * cm; col = L3 + (r % c3); ld = c3; }
    else if ((r -= nB3) < nA3) { row = O3 + (r / c3) * cm; col = L3 + (r % c3); ld = c3; }
    else if ((r -= nA3) < c2) { row = O2 + r * cm; col = b2; ld = 1; }
    else if ((r -= c2) < c2) { row = O2 + r * cm; col = b3; ld = 1; }
    else { r -= c2; row = O3 + r * cm; col = b3; ld = 1; }
    double acc = 0.0;     // the composed matrices stand in for two fp32 convs in a row: form them exactly, round once
    for (int k = 0; k < cm; ++k) acc = fma((double)row[k], (double)col[k * ld], acc);
    p.comp[e] = (float)acc;
  }
}

struct ComposeBwd {
  const float *O2, *O3, *L2, *b2, *L3, *b3;
  const float *dA2, *dB3, *dA3;        // gradients of the composed matrices: [c2][c2], [c2][c3], [c3][c3]
  const float *W2, *W3;                // sum_pixels gc4 (x) t4 [c2][cm], sum_pixels ga4 (x) t4 [c3][cm]  (the O2 t4 / O3 t4 terms)
  const double *s2, *sc3, *s3;         // per-channel sums of g2 [c2], up^T g2 [c2], g3 [c3]  (the bias terms)
  float *dO2, *dO3, *dL2, *dL3, *db2, *db3;
  int c2, c3, cm;
};

__global__ __launch_bounds__(256) void fpn_compose_bwd_kernel(ComposeBwd p) {
  extern __shared__ float lds[];
  const int c2 = p.c2, c3 = p.c3, cm = p.cm;
  float* O2 = lds;              float* O3 = O2 + c2 * cm;
  float* L2 = O3 + c3 * cm;     float* L3 = L2 + cm * c2;
  float* b2 = L3 + cm * c3;     float* b3 = b2 + cm;
  float* dA2 = b3 + cm;         float* dB3 = dA2 + c2 * c2;   float* dA3 = dB3 + c2 * c3;
  float* s2 = dA3 + c3 * c3;    float* sc3 = s2 + c2;         float* s3 = sc3 + c2;
  stage(O2, p.O2, c2 * cm); stage(O3, p.O3, c3 * cm); stage(L2, p.L2, cm * c2); stage(L3, p.L3, cm * c3); stage(b2, p.b2, cm); stage(b3, p.b3, cm);
  stage(dA2, p.dA2, c2 * c2); stage(dB3, p.dB3, c2 * c3); stage(dA3, p.dA3, c3 * c3);
  for (int i = threadIdx.x; i < c2; i += 256) { s2[i] = (float)p.s2[i]; sc3[i] = (float)p.sc3[i]; }
  for (int i = threadIdx.x; i < c3; i += 256) s3[i] = (float)p.s3[i];
  __syncthreads();
  const int nO2 = c2 * cm, nO3 = c3 * cm, nL2 = cm * c2, nL3 = cm * c3;
  const int total = nO2 + nO3 + nL2 + nL3 + cm + cm;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    int r = e;
    if (r < nO2) {                      // dO2[i][k] = W2 + dA2 L2^T + s2 (x) b2 + dB3 L3^T + sc3 (x) b3
      const int i = r / cm, k = r % cm;
      float a = 0.0f, b = 0.0f;
      for (int j = 0; j < c2; ++j) a = fmaf(dA2[i * c2 + j], L2[k * c2 + j], a);
      for (int j = 0; j < c3; ++j) b = fmaf(dB3[i * c3 + j], L3[k * c3 + j], b);
      p.dO2[r] = p.W2[r] + a + s2[i] * b2[k] + b + sc3[i] * b3[k];
    } else if ((r -= nO2) < nO3) {      // dO3[i][k] = W3 + dA3 L3^T + s3 (x) b3
      const int i = r / cm, k = r % cm;
      float a = 0.0f;
      for (int j = 0; j < c3; ++j) a = fmaf(dA3[i * c3 + j], L3[k * c3 + j], a);
      p.dO3[r] = p.W3[r] + a + s3[i] * b3[k];
    } else if ((r -= nO3) < nL2) {      // dL2[k][j] = sum_i O2[i][k] dA2[i][j]
      const int k = r / c2, j = r % c2;
      float a = 0.0f;
      for (int i = 0; i < c2; ++i) a = fmaf(O2[i * cm + k], dA2[i * c2 + j], a);
      p.dL2[r] = a;
    } else if ((r -= nL2) < nL3) {      // dL3[k][j] = sum_i O2[i][k] dB3[i][j] + sum_i O3[i][k] dA3[i][j]
      const int k = r / c3, j = r % c3;
      float a = 0.0f, b = 0.0f;
      for (int i = 0; i < c2; ++i) a = fmaf(O2[i * cm + k], dB3[i * c3 + j], a);
      for (int i = 0; i < c3; ++i) b = fmaf(O3[i * cm + k], dA3[i * c3 + j], b);
      p.dL3[r] = a + b;
    } else if ((r -= nL3) < cm) {       // db2[k] = sum_i O2[i][k] s2[i]
      float a = 0.0f;
      for (int i = 0; i < c2; ++i) a = fmaf(O2[i * cm + r], s2[i], a);
      p.db2[r] = a;
    } else {                            // db3[k] = sum_i O2[i][k] sc3[i] + sum_i O3[i][k] s3[i]
      r -= cm;
      float a = 0.0f, b = 0.0f;
      for (int i = 0; i < c2; ++i) a = fmaf(O2[i * cm + r], sc3[i], a);
      for (int i = 0; i < c3; ++i) b = fmaf(O3[i * cm + r], s3[i], b);
      p.db3[r] = a + b;
    }
  }
}

size_t lds_floats(int c2, int c3, int cm, bool bwd) {
  return (size_t)2 * cm * (c2 + c3) + 2 * cm + (bwd ? (size_t)c2 * c2 + c2 * c3 + c3 * c3 + 2 * c2 + c3 : 0);
}
bool dims_ok(int c2, int c3, int cm) {   // the staged operands must fit 64 KB of LDS
  return c2 >= 1 && c3 >= 1 && cm >= 1 && c2 <= 256 && c3 <= 256 && cm <= 1024 && lds_floats(c2, c3, cm, true) * sizeof(float) <= 64 * 1024;
}

}  // namespace

extern "C" int mdf_fpn_compose_fwd(const float* O2, const float* O3, const float* L2, const float* b2, const float* L3, const float* b3,
                                   int c2, int c3, int cm, float* comp, void* stream) {
  MDF_REQUIRE(O2 && O3 && L2 && b2 && L3 && b3 && comp, "null pointer argument");
  MDF_REQUIRE(dims_ok(c2, c3, cm), "bad shape");
  const int total = c2 * c2 + c2 * c3 + c3 * c3 + 2 * c2 + c3;
  hipLaunchKernelGGL(fpn_compose_fwd_kernel, dim3((total + 255) / 256), dim3(256), lds_floats(c2, c3, cm, false) * sizeof(float), (hipStream_t)stream,
                     ComposeFwd{O2, O3, L2, b2, L3, b3, comp, c2, c3, cm});
  return mdf::check_launch("fpn_compose_fwd_kernel");
}

extern "C" int mdf_fpn_compose_bwd(const float* O2, const float* O3, const float* L2, const float* b2, const float* L3, const float* b3,
                                   const float* dA2, const float* dB3, const float* dA3, const float* W2, const float* W3,
                                   const double* s2, const double* sc3, const double* s3, int c2, int c3, int cm, float* dO2, float* dO3,
                                   float* dL2, float* dL3, float* db2, float* db3, void* stream) {
  MDF_REQUIRE(O2 && O3 && L2 && b2 && L3 && b3 && dA2 && dB3 && dA3 && W2 && W3 && s2 && sc3 && s3, "null pointer argument");
  MDF_REQUIRE(dO2 && dO3 && dL2 && dL3 && db2 && db3, "null pointer argument");
  MDF_REQUIRE(dims_ok(c2, c3, cm), "bad shape");
  const int total = c2 * cm + c3 * cm + cm * c2 + cm * c3 + 2 * cm;
  hipLaunchKernelGGL(fpn_compose_bwd_kernel, dim3((total + 255) / 256), dim3(256), lds_floats(c2, c3, cm, true) * sizeof(float), (hipStream_t)stream,
                     ComposeBwd{O2, O3, L2, b2, L3, b3, dA2, dB3, dA3, W2, W3, s2, sc3, s3, dO2, dO3, dL2, dL3, db2, db3, c2, c3, cm});
  return mdf::check_launch("fpn_compose_bwd_kernel");
}
