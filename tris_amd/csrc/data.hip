// Input-pipeline kernels: the whole pre-decoded dataset lives in HBM as uint8; a training batch is a
// gather + normalise away from it.  All integer / table work here is bit-exact against Pillow + torchvision semantics:
//   * resample_*: Pillow's two-pass antialiased resampling of 8-bit images (fixed point, 22 fraction bits, intermediate
//     image rounded to uint8 between the passes) with host-built coefficient tables  -> dataset/transform.py:29 (F.resize)
//   * gather2d:   Pillow's NEAREST resize (host-built source index tables)            -> dataset/transform.py:32
//   * u8_gather_normalize: to_tensor (/255) + normalize ((x-mean)/std) through a 3x256 table built by the host with the
//     reference's own float ops, written channels-last (or planar)                    -> dataset/transform.py:41-53
//   * gather_rows: token-id rows of the sampled sentences                               -> dataset/ReferDataset.py:172-229
#include "common.h"
#include "tris_hip.h"

namespace {

// one pass of the separable resampler.  AXIS 0: along x (in [H][Win][C] -> out [H][Wout][C]);  AXIS 1: along y
// (in [Hin][W][C] -> out [Hout][W][C]).  bounds[o] = {first source index, tap count}; kk[o*ksize + t] fixed-point taps.
template <int AXIS>
__global__ __launch_bounds__(256) void resample_pass_kernel(const unsigned char* __restrict__ in,
                                                            unsigned char* __restrict__ out, int H, int Win, int Wout,
                                                            int C, const int* __restrict__ bounds,
                                                            const int* __restrict__ kk, int ksize) {
  // AXIS 0: H rows, Wout outputs per row;  AXIS 1: H = Hout output rows, Win = Wout = row width
  const long n = (long)H * Wout * C;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const long p = i / C;
  const int xo = (int)(p % Wout);
  const int yo = (int)(p / Wout);
  const int o = AXIS == 0 ? xo : yo;
  const int first = bounds[2 * o], cnt = bounds[2 * o + 1];
  const int* k = kk + (long)o * ksize;
  int ss = 1 << 21;  // 0.5 in 22-bit fixed point
  if (AXIS == 0) {
    const unsigned char* src = in + ((long)yo * Win + first) * C + c;
    for (int t = 0; t < cnt; ++t) ss += (int)src[(long)t * C] * k[t];
  } else {
    const unsigned char* src = in + ((long)first * Win + xo) * C + c;
    for (int t = 0; t < cnt; ++t) ss += (int)src[(long)t * Win * C] * k[t];
  }
  ss >>= 22;
  out[i] = (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
}

__global__ __launch_bounds__(256) void gather2d_kernel(const unsigned char* __restrict__ in, int Win, int C,
                                                       const int* __restrict__ yidx, const int* __restrict__ xidx,
                                                       int Hout, int Wout, unsigned char* __restrict__ out) {
  const long n = (long)Hout * Wout * C;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const long p = i / C;
  const int xo = (int)(p % Wout), yo = (int)(p / Wout);
  const int ys = yidx[yo], xs = xidx[xo];
  out[i] = (ys < 0 || xs < 0) ? (unsigned char)0 : in[((long)ys * Win + xs) * C + c];
}

// cache [N][HW][3] uint8 -> out fp32, one thread per 4 pixels (12 bytes in as three dwords, 48 bytes out).
// LAYOUT 0: [B][HW][3] (channels-last, what the stem convolution consumes); 1: [B][3][HW] (the reference's NCHW).
template <int LAYOUT>
__global__ __launch_bounds__(256) void u8_gather_normalize_kernel(const unsigned char* __restrict__ cache,
                                                                  const long* __restrict__ index, long HW,
                                                                  const float* __restrict__ lut,
                                                                  float* __restrict__ out) {
  __shared__ float tab[768];
  for (int t = threadIdx.x; t < 768; t += 256) tab[t] = lut[t];
  __syncthreads();
  const int b = blockIdx.y;
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;  // group of 4 pixels
  if (q * 4 >= HW) return;
  const unsigned char* src = cache + index[b] * HW * 3 + q * 12;
  const uint3 w = *reinterpret_cast<const uint3*>(src);  // HW % 4 == 0 and 16-byte aligned images (host-checked)
  unsigned char v[12];
  *reinterpret_cast<unsigned*>(v) = w.x;
  *reinterpret_cast<unsigned*>(v + 4) = w.y;
  *reinterpret_cast<unsigned*>(v + 8) = w.z;
  if (LAYOUT == 0) {
    float r[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) r[t] = tab[(t % 3) * 256 + v[t]];
    float4* dst = reinterpret_cast<float4*>(out + ((long)b * HW + q * 4) * 3);
    dst[0] = make_float4(r[0], r[1], r[2], r[3]);
    dst[1] = make_float4(r[4], r[5], r[6], r[7]);
    dst[2] = make_float4(r[8], r[9], r[10], r[11]);
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float4 o = make_float4(tab[c * 256 + v[c]], tab[c * 256 + v[3 + c]], tab[c * 256 + v[6 + c]],
                             tab[c * 256 + v[9 + c]]);
      *reinterpret_cast<float4*>(out + ((long)b * 3 + c) * HW + q * 4) = o;
    }
  }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const unsigned* __restrict__ table,
                                                          const long* __restrict__ index, long n, int row_words,
                                                          unsigned* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * row_words) return;
  const long r = i / row_words;
  const int w = (int)(i - r * row_words);
  out[i] = table[index[r] * row_words + w];
}

}  // namespace

extern "C" int tris_resample_u8(const unsigned char* in, int Hin, int Win, int C, const int* bounds_h, const int* kk_h,
                                int ksize_h, const int* bounds_v, const int* kk_v, int ksize_v, int Hout, int Wout,
                                unsigned char* tmp, unsigned char* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || C <= 0) return (int)hipErrorInvalidValue;
  const unsigned char* src = in;
  int W = Win;
  if (bounds_h != nullptr) {  // horizontal pass: [Hin][Win][C] -> [Hin][Wout][C]
    unsigned char* dst = bounds_v != nullptr ? tmp : out;
    const long n = (long)Hin * Wout * C;
    hipLaunchKernelGGL(resample_pass_kernel<0>, dim3(cdiv(n, 256)), dim3(256), 0, st, src, dst, Hin, Win, Wout, C,
                       bounds_h, kk_h, ksize_h);
    TRIS_LAUNCH_CHECK();
    src = dst;
    W = Wout;
  }
  if (bounds_v != nullptr) {  // vertical pass: [Hin][W][C] -> [Hout][W][C]
    const long n = (long)Hout * W * C;
    hipLaunchKernelGGL(resample_pass_kernel<1>, dim3(cdiv(n, 256)), dim3(256), 0, st, src, out, Hout, W, W, C, bounds_v,
                       kk_v, ksize_v);
    TRIS_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int tris_gather2d_u8(const unsigned char* in, int Hin, int Win, int C, const int* yidx, const int* xidx,
                                int Hout, int Wout, unsigned char* out, void* stream) {
  (void)Hin;
  const long n = (long)Hout * Wout * C;
  if (n <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(gather2d_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in, Win, C, yidx, xidx,
                     Hout, Wout, out);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_u8_gather_normalize_f32(const unsigned char* cache, const long* index, int B, long HW,
                                            const float* lut, float* out, int planar, void* stream) {
  if (B <= 0 || HW <= 0 || HW % 4 || (((uintptr_t)cache) & 3)) return (int)hipErrorInvalidValue;
  dim3 grid(cdiv(HW / 4, 256), B);
  if (planar)
    hipLaunchKernelGGL(u8_gather_normalize_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, cache, index, HW, lut, out);
  else
    hipLaunchKernelGGL(u8_gather_normalize_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, cache, index, HW, lut, out);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_gather_rows(const void* table, const long* index, long rows, long row_bytes, void* out,
                                void* stream) {
  if (rows <= 0 || row_bytes <= 0 || row_bytes % 4) return (int)hipErrorInvalidValue;
  const long n = rows * (row_bytes / 4);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned*)table,
                     index, rows, (int)(row_bytes / 4), (unsigned*)out);
  TRIS_LAUNCH_CHECK();
  return 0;
}
