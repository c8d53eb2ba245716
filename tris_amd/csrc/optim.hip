// Fused AdamW over one flat parameter arena (gfx950).  Parameters, gradients and both moments of a parameter group
// live in four contiguous fp32 buffers, so one launch updates the whole group at HBM streaming rate
// (16 B/lane loads, 4 streams in, 3 out).  Semantics = torch.optim.AdamW (train_stage1.py:135-139, 370).
#include "common.h"
#include "tris_hip.h"

namespace {
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2s) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  const long n4 = n >> 2;
  const float decay = 1.0f - lr * wd, step = lr / bc1;
  for (long j = i; j < n4; j += stride) {
    float4 P = *reinterpret_cast<float4*>(p + j * 4), G = *reinterpret_cast<const float4*>(g + j * 4);
    float4 M = *reinterpret_cast<float4*>(m + j * 4), V = *reinterpret_cast<float4*>(v + j * 4);
#define TRIS_ADAM1(e)                       \
  P.e *= decay;                             \
  M.e = b1 * M.e + (1.0f - b1) * G.e;       \
  V.e = b2 * V.e + (1.0f - b2) * G.e * G.e; \
  P.e -= step * M.e / (sqrtf(V.e) / bc2s + eps);
    TRIS_ADAM1(x) TRIS_ADAM1(y) TRIS_ADAM1(z) TRIS_ADAM1(w)
#undef TRIS_ADAM1
    *reinterpret_cast<float4*>(p + j * 4) = P;
    *reinterpret_cast<float4*>(m + j * 4) = M;
    *reinterpret_cast<float4*>(v + j * 4) = V;
  }
  for (long j = n4 * 4 + i; j < n; j += stride) {
    float P = p[j] * decay, G = g[j];
    float M = b1 * m[j] + (1.0f - b1) * G, V = b2 * v[j] + (1.0f - b2) * G * G;
    p[j] = P - step * M / (sqrtf(V) / bc2s + eps);
    m[j] = M;
    v[j] = V;
  }
}
// Same update with the step-dependent scalars read from DEVICE memory: hyper = {lr, 1 - beta1^t, sqrt(1 - beta2^t)}.
// A captured (hipGraph) training step replays this launch unchanged while the host refreshes the three floats per step.
__global__ __launch_bounds__(256) void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long n,
                                                        const float* __restrict__ hyper, float b1, float b2, float eps,
                                                        float wd) {
  const float lr = hyper[0], bc1 = hyper[1], bc2s = hyper[2];
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  const long n4 = n >> 2;
  const float decay = 1.0f - lr * wd, step = lr / bc1;
  for (long j = i; j < n4; j += stride) {
    float4 P = *reinterpret_cast<float4*>(p + j * 4), G = *reinterpret_cast<const float4*>(g + j * 4);
    float4 M = *reinterpret_cast<float4*>(m + j * 4), V = *reinterpret_cast<float4*>(v + j * 4);
#define TRIS_ADAM1(e)                       \
  P.e *= decay;                             \
  M.e = b1 * M.e + (1.0f - b1) * G.e;       \
  V.e = b2 * V.e + (1.0f - b2) * G.e * G.e; \
  P.e -= step * M.e / (sqrtf(V.e) / bc2s + eps);
    TRIS_ADAM1(x) TRIS_ADAM1(y) TRIS_ADAM1(z) TRIS_ADAM1(w)
#undef TRIS_ADAM1
    *reinterpret_cast<float4*>(p + j * 4) = P;
    *reinterpret_cast<float4*>(m + j * 4) = M;
    *reinterpret_cast<float4*>(v + j * 4) = V;
  }
  for (long j = n4 * 4 + i; j < n; j += stride) {
    float P = p[j] * decay, G = g[j];
    float M = b1 * m[j] + (1.0f - b1) * G, V = b2 * v[j] + (1.0f - b2) * G * G;
    p[j] = P - step * M / (sqrtf(V) / bc2s + eps);
    m[j] = M;
    v[j] = V;
  }
}
}  // namespace

// step_count is the 1-based step index t; bias corrections are computed on the host in double.
extern "C" int tris_adamw_f32(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int step_count, void* stream) {
  double bc1 = 1.0 - pow((double)beta1, (double)step_count);
  double bc2 = 1.0 - pow((double)beta2, (double)step_count);
  long g4 = (n / 4 + 255) / 256;
  if (g4 > 8192) g4 = 8192;
  if (g4 < 1) g4 = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)g4), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                     beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2));
  TRIS_LAUNCH_CHECK();
  return 0;
}

// hyper: device pointer to {lr, 1 - beta1^t, sqrt(1 - beta2^t)} (computed by the host in double, as above)
extern "C" int tris_adamw_dev_f32(float* p, const float* g, float* m, float* v, long n, const float* hyper, float beta1,
                                  float beta2, float eps, float weight_decay, void* stream) {
  long g4 = (n / 4 + 255) / 256;
  if (g4 > 8192) g4 = 8192;
  if (g4 < 1) g4 = 1;
  hipLaunchKernelGGL(adamw_dev_kernel, dim3((unsigned)g4), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, hyper, beta1,
                     beta2, eps, weight_decay);
  TRIS_LAUNCH_CHECK();
  return 0;
}
