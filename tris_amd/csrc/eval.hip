// Evaluation post-processing for one (image, sentence) response map (validate.py:180-190, utils/util.py:9-15):
// bilinear resize (align_corners=True) to the annotation size, max-normalise, threshold, intersection / union counts
// and the arg-max pixel for the pointing-game hit test.  Integer counts are exact (64-bit atomics).
#include "common.h"
#include "tris_hip.h"

namespace {
__device__ __forceinline__ float sample_ac(const float* __restrict__ m, int S, int oH, int oW, int oy, int ox) {
  float sy = oH > 1 ? (float)(S - 1) / (float)(oH - 1) : 0.f, sx = oW > 1 ? (float)(S - 1) / (float)(oW - 1) : 0.f;
  float fy = sy * (float)oy, fx = sx * (float)ox;
  int y0 = min((int)fy, S - 1), x0 = min((int)fx, S - 1);
  int y1 = y0 + (y0 < S - 1 ? 1 : 0), x1 = x0 + (x0 < S - 1 ? 1 : 0);
  float wy = fy - (float)y0, wx = fx - (float)x0;
  float v00 = m[y0 * S + x0], v01 = m[y0 * S + x1], v10 = m[y1 * S + x0], v11 = m[y1 * S + x1];
  float top = v00 + wx * (v01 - v00), bot = v10 + wx * (v11 - v10);
  return top + wy * (bot - top);
}

// pass 1: per-block (max, first arg-max) partials -> ws[2*blk], ws[2*blk+1] (index stored as float bits)
__global__ __launch_bounds__(256) void eval_max_kernel(const float* __restrict__ m, int S, int oH, int oW,
                                                       float* __restrict__ ws) {
  __shared__ float sv[256];
  __shared__ int si[256];
  long n = (long)oH * oW;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v = sample_ac(m, S, oH, oW, (int)(i / oW), (int)(i % oW));
    if (v > best) { best = v; bi = (int)i; }
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      float ov = sv[threadIdx.x + o];
      int oi = si[threadIdx.x + o];
      if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { ws[2 * blockIdx.x] = sv[0]; ws[2 * blockIdx.x + 1] = __int_as_float(si[0]); }
}
// pass 2: final max / arg-max; zero the counters
__global__ void eval_max_final_kernel(float* __restrict__ ws, int nb, long* __restrict__ out_iu) {
  if (threadIdx.x != 0) return;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int b = 0; b < nb; ++b) {
    float v = ws[2 * b];
    int i = __float_as_int(ws[2 * b + 1]);
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  }
  ws[2 * nb] = best;
  out_iu[0] = 0;
  out_iu[1] = 0;
  out_iu[2] = bi;
}
// pass 3: normalise, threshold, count
__global__ __launch_bounds__(256) void eval_count_kernel(const float* __restrict__ m, int S,
                                                         const unsigned char* __restrict__ target, int oH, int oW,
                                                         float* __restrict__ cam, const float* __restrict__ ws, int nb,
                                                         long* __restrict__ out_iu) {
  __shared__ int ri[4], ru[4];
  const float denom = ws[2 * nb] + 1e-5f;
  long n = (long)oH * oW;
  int I = 0, U = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v = sample_ac(m, S, oH, oW, (int)(i / oW), (int)(i % oW)) / denom;
    if (cam) cam[i] = v;
    bool p = v > 1e-9f, t = target[i] != 0;
    I += (p && t) ? 1 : 0;
    U += (p || t) ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) { I += __shfl_xor(I, o, 64); U += __shfl_xor(U, o, 64); }
  if ((threadIdx.x & 63) == 0) { ri[threadIdx.x >> 6] = I; ru[threadIdx.x >> 6] = U; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd((unsigned long long*)&out_iu[0], (unsigned long long)(ri[0] + ri[1] + ri[2] + ri[3]));
    atomicAdd((unsigned long long*)&out_iu[1], (unsigned long long)(ru[0] + ru[1] + ru[2] + ru[3]));
  }
}
}  // namespace

extern "C" int tris_eval_post_f32(const float* relu_map, int S, const unsigned char* target, int oH, int oW, float* cam,
                                  long* out_iu, float* workspace, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  long n = (long)oH * oW;
  int nb = (int)((n + 255) / 256);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(eval_max_kernel, dim3(nb), dim3(256), 0, st, relu_map, S, oH, oW, workspace);
  TRIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(eval_max_final_kernel, dim3(1), dim3(64), 0, st, workspace, nb, out_iu);
  TRIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(eval_count_kernel, dim3(nb), dim3(256), 0, st, relu_map, S, target, oH, oW, cam, workspace, nb,
                     out_iu);
  TRIS_LAUNCH_CHECK();
  return 0;
}
