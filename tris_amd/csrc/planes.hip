// fp16 operand planes ("P8", planes.h) for the h2 arithmetic: conversions and the element-wise passes of the RN50 trunk that WRITE
// their output as planes (BatchNorm apply / backward apply, pooling), so that the products consuming those tensors take their
// operands as they lie in memory (gemm_fast.h PREC 4) instead of splitting them again in every tile that reads them.
// The scale of a plane tensor comes from an amax word that is final BEFORE the writing pass starts: the tensor's true amax where a
// pass over it exists anyway (weights: tris_amax_segments_f32), otherwise an upper bound (BatchNorm outputs: Samuelson's bound from
// the affine parameters, tris_bn_out_bound_f32; BatchNorm input gradients: tris_bn_bwd_bound_f32).
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "tris_hip.h"

namespace {
#include "x3_split.h"
#include "planes.h"

__device__ __forceinline__ float4 ldf4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void stf4(float* p, const float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---- conversions -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void to_planes_kernel(const float* __restrict__ x, float* __restrict__ out, long n8,
                                                        const unsigned* __restrict__ word) {
  const float s = pl_scale(word);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < n8; g += stride)
    pl8_store(out, g, pl8_split(ldf4(x + g * 8), ldf4(x + g * 8 + 4), s));
}
// many tensors of one flat buffer in ONE launch (the convolution weights of an optimiser arena): grid (chunks, segments); segment i
// is sizes[i] floats at base + offs[i] with amax word slots + 2048 i; its planes go to out_base + offs[i]
__global__ __launch_bounds__(256) void to_planes_segments_kernel(const float* __restrict__ base, const long* __restrict__ offs,
                                                                 const long* __restrict__ sizes, const long* __restrict__ slot_index,
                                                                 const unsigned* __restrict__ slots, float* __restrict__ out_base) {
  const long n8 = sizes[blockIdx.y] >> 3;
  const float* x = base + offs[blockIdx.y];
  float* out = out_base + offs[blockIdx.y];
  const unsigned* word = slots + slot_index[blockIdx.y] * 2048;
  const float s = pl_scale(word);
  const long stride = (long)gridDim.x * blockDim.x;
  unsigned below = 0u;
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < n8; g += stride) {
    const float4 a = ldf4(x + g * 8), b = ldf4(x + g * 8 + 4);
    below += pl_below_floor(a, b, s);
    pl8_store(out, g, pl8_split(a, b, s));
  }
  pl_floor_commit(below, word);
}
// TRANSPOSED planes of 1x1-convolution weights W [rows = Cout][cols = Cin] -> planes of W^T [Cin][Cout] (P8 along Cout), so that the
// data gradient dX = dY . W is the same row-major A x B^T product as the forward (B^T = W^T, k = Cout contiguous).  grid (tiles, segments):
// a block turns 64 x 64 tiles around through LDS (coalesced reads along Cin, 32-byte group stores along Cout).
__global__ __launch_bounds__(256) void to_planes_t_segments_kernel(const float* __restrict__ base, const long* __restrict__ offs,
                                                                   const long* __restrict__ rows, const long* __restrict__ cols,
                                                                   const long* __restrict__ slot_index,
                                                                   const unsigned* __restrict__ slots, float* __restrict__ out_base) {
  __shared__ float t[64][65];
  const int R = (int)rows[blockIdx.y], C = (int)cols[blockIdx.y];
  const float* x = base + offs[blockIdx.y];
  float* out = out_base + offs[blockIdx.y];
  const float s = pl_scale(slots + slot_index[blockIdx.y] * 2048);
  const int tr = R / 64, tc = C / 64;
  for (int tile = blockIdx.x; tile < tr * tc; tile += gridDim.x) {
    const int r0 = (tile / tc) * 64, c0 = (tile % tc) * 64;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
      const int r = i >> 4, c4 = (i & 15) * 4;
      const float4 v = ldf4(x + (long)(r0 + r) * C + c0 + c4);
      t[r][c4] = v.x; t[r][c4 + 1] = v.y; t[r][c4 + 2] = v.z; t[r][c4 + 3] = v.w;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {   // group (column c of W = row of W^T, 8 consecutive rows of W)
      const int c = i >> 3, g = i & 7;
      const float4 a = make_float4(t[g * 8][c], t[g * 8 + 1][c], t[g * 8 + 2][c], t[g * 8 + 3][c]);
      const float4 b = make_float4(t[g * 8 + 4][c], t[g * 8 + 5][c], t[g * 8 + 6][c], t[g * 8 + 7][c]);
      pl8_store(out, ((long)(c0 + c) * R + r0 + g * 8) >> 3, pl8_split(a, b, s));
    }
  }
}
__global__ __launch_bounds__(256) void from_planes_kernel(const float* __restrict__ pl, float* __restrict__ out, long n8,
                                                          const unsigned* __restrict__ word) {
  const float inv = 1.0f / pl_scale(word);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < n8; g += stride) {
    uint4 hi, lo;
    pl8_load(pl, g, hi, lo);
    float4 a, b;
    pl8_join(hi, lo, inv, a, b);
    stf4(out + g * 8, a);
    stf4(out + g * 8 + 4, b);
  }
}
inline int grid_for(long n8) {
  long g = (n8 + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}
inline bool al16p(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int tris_h2_planes_f32(const float* x, float* out, long n, const unsigned* word, void* stream) {
  if (n < 8 || (n & 7) || !al16p(x) || !al16p(out) || word == nullptr) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(to_planes_kernel, dim3(grid_for(n >> 3)), dim3(256), 0, (hipStream_t)stream, x, out, n >> 3, word);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_h2_planes_segments_f32(const float* base, const long* offs, const long* sizes, const long* slot_index, int nseg,
                                           const unsigned* slots, float* out_base, void* stream) {
  if (nseg < 1 || !al16p(base) || !al16p(out_base)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(to_planes_segments_kernel, dim3(64, (unsigned)nseg), dim3(256), 0, (hipStream_t)stream, base, offs, sizes,
                     slot_index, slots, out_base);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_h2_planes_t_segments_f32(const float* base, const long* offs, const long* rows, const long* cols,
                                             const long* slot_index, int nseg, const unsigned* slots, float* out_base, void* stream) {
  if (nseg < 1 || !al16p(base) || !al16p(out_base)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(to_planes_t_segments_kernel, dim3(64, (unsigned)nseg), dim3(256), 0, (hipStream_t)stream, base, offs, rows, cols,
                     slot_index, slots, out_base);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_h2_unplanes_f32(const float* planes, float* out, long n, const unsigned* word, void* stream) {
  if (n < 8 || (n & 7) || !al16p(planes) || !al16p(out) || word == nullptr) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(from_planes_kernel, dim3(grid_for(n >> 3)), dim3(256), 0, (hipStream_t)stream, planes, out, n >> 3, word);
  TRIS_LAUNCH_CHECK();
  return 0;
}

// ---- BatchNorm passes that write planes -------------------------------------------------------------------------------------
namespace {
typedef float pf32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned pu32x4_t __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 ldx(const float* p) {
  if constexpr (NT) { const pf32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const pf32x4_t*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
  else return ldf4(p);
}
template <bool NT> __device__ __forceinline__ void stx(float* p, const float4 v) {
  if constexpr (NT) { const pf32x4_t w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<pf32x4_t*>(p)); }
  else stf4(p, v);
}
template <bool NT> __device__ __forceinline__ uint4 ldu(const float* p) {
  if constexpr (NT) { const pu32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const pu32x4_t*>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
  else return *reinterpret_cast<const uint4*>(p);
}
template <bool NT> __device__ __forceinline__ void stu(float* p, const uint4 v) {
  if constexpr (NT) { const pu32x4_t w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<pu32x4_t*>(p)); }
  else *reinterpret_cast<uint4*>(p) = v;
}
// the streaming form of norm.hip ("BIG": one contiguous piece per block, nontemporal, PL_U groups of 8 elements per thread and stream)
struct Ch8 { float4 a, b; };
__device__ __forceinline__ Ch8 ld8(const float* p) { Ch8 v; v.a = ldf4(p); v.b = ldf4(p + 4); return v; }
__device__ __forceinline__ float4 mul4(const float4 x, const float4 y) { return make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w); }
__device__ __forceinline__ float4 bn4(const float4 x, const float4 mu, const float4 sc, const float4 b) {   // bn_apply_kernel's expression
  return make_float4((x.x - mu.x) * sc.x + b.x, (x.y - mu.y) * sc.y + b.y, (x.z - mu.z) * sc.z + b.z, (x.w - mu.w) * sc.w + b.w);
}
__device__ __forceinline__ float4 add4(const float4 x, const float4 y) { return make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w); }
__device__ __forceinline__ float4 relu4(const float4 x) { return make_float4(fmaxf(x.x, 0.f), fmaxf(x.y, 0.f), fmaxf(x.z, 0.f), fmaxf(x.w, 0.f)); }

// ---- access pattern of the streaming passes below ----------------------------------------------------------------------------
// A group of 8 elements is 32 bytes; a thread that loaded / stored its own group with two 16-byte instructions would touch memory at
// a 32-byte lane stride -- every instruction half-covers its cache lines (measured: the first version of these passes ran at half
// the rate of their fp32 forms).  Instead a wave walks UNITS of 512 elements = two 1 KB chunks; an instruction is lane-linear (lane l
// <-> bytes 16 l of the chunk, fully coalesced), and the two lanes of a pair (2p, 2p + 1) trade one register quad through DPP so
// that the even lane ends up with the whole group p of chunk 0 and the odd lane with group p of chunk 1 -- for fp32 data (elements
// 0-3 | 4-7 of a group sit in adjacent lanes) and for plane data (hi piece | lo piece sit in adjacent lanes) alike, and the same
// trade turns results back into the chunks' lane-linear order (pair_xchg is its own inverse).
__device__ __forceinline__ float dpp_nb(const float v) {   // the value of lane ^ 1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ void pair_xchg(float4& a, float4& b, const bool odd) {
  const float4 send = odd ? a : b;
  const float4 recv = make_float4(dpp_nb(send.x), dpp_nb(send.y), dpp_nb(send.z), dpp_nb(send.w));
  if (odd) a = recv; else b = recv;
}
__device__ __forceinline__ void pair_xchg(uint4& a, uint4& b, const bool odd) {
  float4 fa = __builtin_bit_cast(float4, a), fb = __builtin_bit_cast(float4, b);
  pair_xchg(fa, fb, odd);
  a = __builtin_bit_cast(uint4, fa);
  b = __builtin_bit_cast(uint4, fb);
}
constexpr int PL_UPB = 8;   // units per block of the streaming ("BIG") form: two per wave, both in flight (16 KB per block and stream)
// the units a wave walks: BIG: block b owns units [b PL_UPB, (b + 1) PL_UPB), wave w takes b PL_UPB + w and + 4; otherwise a grid-stride
// walk.  Either way a thread's units are a multiple of 2048 elements apart, so it sees ONE group of 8 channels (C divides 2048).
template <bool BIG> struct UnitWalk {
  long u, step, end;
  __device__ UnitWalk(long units, int wave) {
    if (BIG) { u = (long)blockIdx.x * PL_UPB + wave; step = 4; end = min(units, (long)(blockIdx.x + 1) * PL_UPB); }
    else { u = (long)blockIdx.x * 4 + wave; step = (long)gridDim.x * 4; end = units; }
  }
};

// Y (planes, scale from out_word) = [relu]((X - mean) * invstd * gamma + beta [+ resid]); resid: fp32 (rk 1) or planes with the scale
// of resid_word (rk 2).
template <bool BIG>
__global__ __launch_bounds__(256) void bn_apply_pl_kernel(const float* __restrict__ X, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ resid, int rk,
                                                          const unsigned* __restrict__ resid_word, float* __restrict__ Y,
                                                          const unsigned* __restrict__ out_word, long n8, int C, int relu,
                                                          unsigned char* __restrict__ mask_out = nullptr) {
  const float s = pl_scale(out_word);
  const float rinv = rk == 2 ? 1.0f / pl_scale(resid_word) : 1.0f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool odd = lane & 1;
  const long n = n8 * 8;
  UnitWalk<BIG> w((n + 511) / 512, wave);
  const int c = (int)((w.u * 512 + (odd ? 256 : 0) + 8 * (lane >> 1)) % C);
  const Ch8 mu = ld8(mean + c), is = ld8(invstd + c), g = ld8(gamma + c), b = ld8(beta + c);
  const float4 sca = mul4(is.a, g.a), scb = mul4(is.b, g.b);
  const float4 z4 = make_float4(0, 0, 0, 0);
  unsigned below = 0u;     // non-zero outputs under the 2^-27 floor of the scale (planes.h range tell-tale)
  auto finish = [&](long e0, float4 x0, float4 x1, float4 r0, float4 r1) {
    const long e1 = e0 + 256;
    pair_xchg(x0, x1, odd);
    float4 y0 = bn4(x0, mu.a, sca, b.a), y1 = bn4(x1, mu.b, scb, b.b);
    if (rk) {
      pair_xchg(r0, r1, odd);      // fp32: the group's elements 0-3 | 4-7; planes: its hi | lo piece
      if (rk == 2) {
        float4 ra, rb;
        pl8_join(__builtin_bit_cast(uint4, r0), __builtin_bit_cast(uint4, r1), rinv, ra, rb);
        r0 = ra;
        r1 = rb;
      }
      y0 = add4(y0, r0);
      y1 = add4(y1, r1);
    }
    if (relu) { y0 = relu4(y0); y1 = relu4(y1); }
    below += pl_below_floor(y0, y1, s);
    Pl8 o = pl8_split(y0, y1, s);
    if (mask_out != nullptr) {
      // the ReLU mask of this group of 8 channels as ONE byte (bit t <=> element t has a non-zero piece: exactly what a reader of the
      // planes would decide, pl8_positive): the backward reads 1 bit per element instead of the 4-byte plane element (tris_bn_mask_next)
      const long eg = (e0 - lane * 4) + (odd ? 256 : 0) + 8 * (lane >> 1);    // first element of this thread's group
      if (eg < n) mask_out[eg >> 3] = (unsigned char)pl8_positive(o.hi, o.lo);
    }
    pair_xchg(o.hi, o.lo, odd);
    if (e0 < n) stu<BIG>(Y + e0, o.hi);
    if (e1 < n) stu<BIG>(Y + e1, o.lo);
  };
  if (BIG) {
    // (a block takes pieces b, b + gridDim.x, ...: the launch may hold fewer blocks than pieces -- pl_big_blocks)
    const long units = (n + 511) / 512;
    for (long pb = blockIdx.x; pb * PL_UPB < units; pb += gridDim.x) {
      const long u0 = pb * PL_UPB + wave, uend = min(units, (pb + 1) * PL_UPB);
      float4 x0[2], x1[2], r0[2], r1[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const long e0 = (u0 + 4 * q) * 512 + lane * 4, e1 = e0 + 256;
        const bool ok0 = (u0 + 4 * q) < uend && e0 < n, ok1 = (u0 + 4 * q) < uend && e1 < n;
        x0[q] = ok0 ? ldx<true>(X + e0) : z4;
        x1[q] = ok1 ? ldx<true>(X + e1) : z4;
        r0[q] = (ok0 && rk) ? ldx<true>(resid + e0) : z4;
        r1[q] = (ok1 && rk) ? ldx<true>(resid + e1) : z4;
      }
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if ((u0 + 4 * q) < uend) finish((u0 + 4 * q) * 512 + lane * 4, x0[q], x1[q], r0[q], r1[q]);   // (wave-uniform)
    }
  } else {
    for (; w.u < w.end; w.u += w.step) {
      const long e0 = w.u * 512 + lane * 4, e1 = e0 + 256;
      finish(e0, e0 < n ? ldf4(X + e0) : z4, e1 < n ? ldf4(X + e1) : z4, (rk && e0 < n) ? ldf4(resid + e0) : z4,
             (rk && e1 < n) ? ldf4(resid + e1) : z4);
    }
  }
  pl_floor_commit(below, out_word);
}

// Yp (planes) = avgpool2(relu(bn(X)))  (norm.hip bn_apply_pool_kernel's expression on the four normalised pixels)
__global__ __launch_bounds__(256) void bn_apply_pool_pl_kernel(const float* __restrict__ X, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ Yp,
                                                               const unsigned* __restrict__ out_word, int B, int H, int W, int C) {
  const float s = pl_scale(out_word);
  const int Ho = H >> 1, Wo = W >> 1, C8 = C >> 3;
  const long n = (long)B * Ho * Wo * C8;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int c = (int)(i % C8) * 8;
    long t = i / C8;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const Ch8 mu = ld8(mean + c), is = ld8(invstd + c), g = ld8(gamma + c), be = ld8(beta + c);
    const float4 sca = mul4(is.a, g.a), scb = mul4(is.b, g.b);
    const float* p = X + (((long)b * H + oy * 2) * W + ox * 2) * C + c;
    auto pix = [&](const float* q, float4& ya, float4& yb) {
      ya = relu4(bn4(ldf4(q), mu.a, sca, be.a));
      yb = relu4(bn4(ldf4(q + 4), mu.b, scb, be.b));
    };
    float4 a0, a1, b0, b1, c0, c1, d0, d1;
    pix(p, a0, a1);
    pix(p + C, b0, b1);
    pix(p + (long)W * C, c0, c1);
    pix(p + (long)W * C + C, d0, d1);
    const float4 o0 = make_float4(0.25f * (a0.x + b0.x + c0.x + d0.x), 0.25f * (a0.y + b0.y + c0.y + d0.y),
                                  0.25f * (a0.z + b0.z + c0.z + d0.z), 0.25f * (a0.w + b0.w + c0.w + d0.w));
    const float4 o1 = make_float4(0.25f * (a1.x + b1.x + c1.x + d1.x), 0.25f * (a1.y + b1.y + c1.y + d1.y),
                                  0.25f * (a1.z + b1.z + c1.z + d1.z), 0.25f * (a1.w + b1.w + c1.w + d1.w));
    pl8_store(Yp, i, pl8_split(o0, o1, s));
  }
}

// Y[b, oy, ox, :] (planes, SAME scale word as X) = mean of the 2 x 2 block of X (planes): norm.hip avgpool2_fwd_kernel on rebuilt values
__global__ __launch_bounds__(256) void avgpool2_pl_kernel(const float* __restrict__ X, float* __restrict__ Y,
                                                          const unsigned* __restrict__ word, int B, int H, int W, int C) {
  const float s = pl_scale(word), inv = 1.0f / s;
  const int Ho = H >> 1, Wo = W >> 1, C8 = C >> 3;
  const long n = (long)B * Ho * Wo * C8;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int c = (int)(i % C8) * 8;
    long t = i / C8;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const float* p = X + (((long)b * H + oy * 2) * W + ox * 2) * C + c;
    auto pix = [&](const float* q, float4& ya, float4& yb) {
      pl8_join(*reinterpret_cast<const uint4*>(q), *reinterpret_cast<const uint4*>(q + 4), inv, ya, yb);
    };
    float4 a0, a1, b0, b1, c0, c1, d0, d1;
    pix(p, a0, a1);
    pix(p + C, b0, b1);
    pix(p + (long)W * C, c0, c1);
    pix(p + (long)W * C + C, d0, d1);
    const float4 o0 = make_float4(0.25f * (a0.x + b0.x + c0.x + d0.x), 0.25f * (a0.y + b0.y + c0.y + d0.y),
                                  0.25f * (a0.z + b0.z + c0.z + d0.z), 0.25f * (a0.w + b0.w + c0.w + d0.w));
    const float4 o1 = make_float4(0.25f * (a1.x + b1.x + c1.x + d1.x), 0.25f * (a1.y + b1.y + c1.y + d1.y),
                                  0.25f * (a1.z + b1.z + c1.z + d1.z), 0.25f * (a1.w + b1.w + c1.w + d1.w));
    pl8_store(Y, i, pl8_split(o0, o1, s));
  }
}

__device__ __forceinline__ long pooled_row_pl(long row, int H, int W) {   // (norm.hip pooled_row)
  const int x = (int)(row % W);
  const long t = row / W;
  const int y = (int)(t % H);
  const long b = t / H;
  return (b * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1);
}
// dX (planes, scale from out_word) = gamma * invstd * (dz - sum_dz / cnt - xhat * sum_dzxhat / cnt), dz = dY masked by the ReLU
// (Y given as planes: y > 0 <=> a piece is non-zero; beta_mask: recomputed from X; neither: no ReLU); optional dZ (fp32) <- dz.
// norm.hip bn_bwd_apply_kernel's arithmetic on groups of 8 channels, in the unit walk above.  POOL: dY is the gradient of avgpool2 of
// the output (read per group, a quarter of the traffic).
template <bool POOL, bool BIG>
__global__ __launch_bounds__(256) void bn_bwd_apply_pl_kernel(const float* __restrict__ dY, const float* __restrict__ Ypl,
                                                              const float* __restrict__ X, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ sum_dz, const float* __restrict__ sum_dzx,
                                                              float inv_cnt, float* __restrict__ dX, const unsigned* __restrict__ out_word,
                                                              float* __restrict__ dZ, long n8, int C, const float* __restrict__ beta_mask,
                                                              int pool_h, int pool_w) {
  const float s = pl_scale(out_word);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool odd = lane & 1;
  const long n = n8 * 8;
  UnitWalk<BIG> w((n + 511) / 512, wave);
  const int c = (int)((w.u * 512 + (odd ? 256 : 0) + 8 * (lane >> 1)) % C);
  const Ch8 mu = ld8(mean + c), is = ld8(invstd + c), ga = ld8(gamma + c), sa = ld8(sum_dz + c), sb = ld8(sum_dzx + c);
  Ch8 be;
  be.a = be.b = make_float4(0, 0, 0, 0);
  if (beta_mask) be = ld8(beta_mask + c);
  const float4 k1a = mul4(ga.a, is.a), k1b = mul4(ga.b, is.b);
  const float4 k2a = make_float4(sa.a.x * inv_cnt, sa.a.y * inv_cnt, sa.a.z * inv_cnt, sa.a.w * inv_cnt);
  const float4 k2b = make_float4(sa.b.x * inv_cnt, sa.b.y * inv_cnt, sa.b.z * inv_cnt, sa.b.w * inv_cnt);
  const float4 k3a = make_float4(is.a.x * sb.a.x * inv_cnt, is.a.y * sb.a.y * inv_cnt, is.a.z * sb.a.z * inv_cnt, is.a.w * sb.a.w * inv_cnt);
  const float4 k3b = make_float4(is.b.x * sb.b.x * inv_cnt, is.b.y * sb.b.y * inv_cnt, is.b.z * sb.b.z * inv_cnt, is.b.w * sb.b.w * inv_cnt);
  const bool use_y = !beta_mask && Ypl != nullptr;
  const float4 z4 = make_float4(0, 0, 0, 0);
  unsigned below = 0u;
  auto half = [&](float4 g, const float4 x, const float4 mu4, const float4 k1, const float4 k2, const float4 k3, const float4 be4,
                  unsigned pos, float4& gm) {
    if (beta_mask) {
      if (!((x.x - mu4.x) * k1.x + be4.x > 0.f)) g.x = 0.f;
      if (!((x.y - mu4.y) * k1.y + be4.y > 0.f)) g.y = 0.f;
      if (!((x.z - mu4.z) * k1.z + be4.z > 0.f)) g.z = 0.f;
      if (!((x.w - mu4.w) * k1.w + be4.w > 0.f)) g.w = 0.f;
    } else if (use_y) {
      if (!(pos & 1u)) g.x = 0.f;
      if (!(pos & 2u)) g.y = 0.f;
      if (!(pos & 4u)) g.z = 0.f;
      if (!(pos & 8u)) g.w = 0.f;
    }
    gm = g;
    return make_float4(k1.x * (g.x - k2.x - (x.x - mu4.x) * k3.x), k1.y * (g.y - k2.y - (x.y - mu4.y) * k3.y),
                       k1.z * (g.z - k2.z - (x.z - mu4.z) * k3.z), k1.w * (g.w - k2.w - (x.w - mu4.w) * k3.w));
  };
  // (chunk-ordered registers of one unit -> this thread's group -> chunk-ordered results)
  auto finish = [&](long e0, float4 g0, float4 g1, float4 x0, float4 x1, float4 y0, float4 y1) {
    const long e1 = e0 + 256;
    pair_xchg(x0, x1, odd);
    if (!POOL) pair_xchg(g0, g1, odd);
    unsigned pos = 0u;
    if (use_y) {
      pair_xchg(y0, y1, odd);
      pos = pl8_positive(__builtin_bit_cast(uint4, y0), __builtin_bit_cast(uint4, y1));
    }
    float4 m0, m1;
    const float4 o0 = half(g0, x0, mu.a, k1a, k2a, k3a, be.a, pos & 15u, m0);
    const float4 o1 = half(g1, x1, mu.b, k1b, k2b, k3b, be.b, pos >> 4, m1);
    if (dZ) {
      pair_xchg(m0, m1, odd);
      if (e0 < n) stx<BIG>(dZ + e0, m0);
      if (e1 < n) stx<BIG>(dZ + e1, m1);
    }
    below += pl_below_floor(o0, o1, s);
    Pl8 o = pl8_split(o0, o1, s);
    pair_xchg(o.hi, o.lo, odd);
    if (e0 < n) stu<BIG>(dX + e0, o.hi);
    if (e1 < n) stu<BIG>(dX + e1, o.lo);
  };
  // POOL: the pooled gradient of THIS thread's group (its own 8 channels of the pooled pixel), already a quarter
  auto pooled_g = [&](long u, float4& g0, float4& g1) {
    const long eg = u * 512 + (odd ? 256 : 0) + 8 * (lane >> 1);
    g0 = g1 = z4;
    if (eg < n) {
      const float* q = dY + pooled_row_pl(eg / C, pool_h, pool_w) * C + c;
      g0 = ldf4(q);
      g1 = ldf4(q + 4);
      g0 = make_float4(g0.x * 0.25f, g0.y * 0.25f, g0.z * 0.25f, g0.w * 0.25f);
      g1 = make_float4(g1.x * 0.25f, g1.y * 0.25f, g1.z * 0.25f, g1.w * 0.25f);
    }
  };
  if (BIG) {
    const long units = (n + 511) / 512;
    for (long pb = blockIdx.x; pb * PL_UPB < units; pb += gridDim.x) {     // (pieces b, b + gridDim.x, ...: pl_big_blocks)
      const long u0 = pb * PL_UPB + wave, uend = min(units, (pb + 1) * PL_UPB);
      float4 g0[2], g1[2], x0[2], x1[2], y0[2], y1[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const long u = u0 + 4 * q;
        const long e0 = u * 512 + lane * 4, e1 = e0 + 256;
        const bool ok0 = u < uend && e0 < n, ok1 = u < uend && e1 < n;
        if (POOL) {
          if (u < uend) pooled_g(u, g0[q], g1[q]); else g0[q] = g1[q] = z4;
        } else {
          g0[q] = ok0 ? ldx<true>(dY + e0) : z4;
          g1[q] = ok1 ? ldx<true>(dY + e1) : z4;
        }
        x0[q] = ok0 ? ldx<true>(X + e0) : z4;
        x1[q] = ok1 ? ldx<true>(X + e1) : z4;
        y0[q] = (ok0 && use_y) ? ldx<true>(Ypl + e0) : z4;
        y1[q] = (ok1 && use_y) ? ldx<true>(Ypl + e1) : z4;
      }
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if ((u0 + 4 * q) < uend) finish((u0 + 4 * q) * 512 + lane * 4, g0[q], g1[q], x0[q], x1[q], y0[q], y1[q]);
    }
  } else {
    for (; w.u < w.end; w.u += w.step) {
      const long e0 = w.u * 512 + lane * 4, e1 = e0 + 256;
      float4 g0, g1;
      if (POOL) pooled_g(w.u, g0, g1);
      else { g0 = e0 < n ? ldf4(dY + e0) : z4; g1 = e1 < n ? ldf4(dY + e1) : z4; }
      finish(e0, g0, g1, e0 < n ? ldf4(X + e0) : z4, e1 < n ? ldf4(X + e1) : z4, (use_y && e0 < n) ? ldf4(Ypl + e0) : z4,
             (use_y && e1 < n) ? ldf4(Ypl + e1) : z4);
    }
  }
  pl_floor_commit(below, out_word);
}

// amax words that are bounds (one block; only line 0 of the word is written, the others stay zero -- the caller zeroes the word)
__global__ __launch_bounds__(256) void bn_out_bound2_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, int C,
                                                            float xhat_max, const unsigned* __restrict__ add_word,
                                                            unsigned* __restrict__ out) {
  __shared__ float red[4];
  float m = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, fabsf(gamma[c]) * xhat_max + fabsf(beta[c]));
  m = block_max_256(m, red);
  if (add_word != nullptr) m += __builtin_bit_cast(float, h2_amax_of(add_word, threadIdx.x & 63));
  if (threadIdx.x == 0) out[0] = __builtin_bit_cast(unsigned, m);
}
__global__ __launch_bounds__(256) void bn_bwd_bound_kernel(const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                           const float* __restrict__ sum_dz, const float* __restrict__ sum_dzx, int C,
                                                           float inv_cnt, float xhat_max, const unsigned* __restrict__ dz_word,
                                                           unsigned* __restrict__ out) {
  __shared__ float red[4];
  const float adz = __builtin_bit_cast(float, h2_amax_of(dz_word, threadIdx.x & 63));
  float m = 0.f;
  for (int c = threadIdx.x; c < C; c += 256)
    m = fmaxf(m, fabsf(gamma[c] * invstd[c]) * (adz + fabsf(sum_dz[c]) * inv_cnt + xhat_max * fabsf(sum_dzx[c]) * inv_cnt));
  m = block_max_256(m, red);
  if (threadIdx.x == 0) out[0] = __builtin_bit_cast(unsigned, m);
}

inline bool pl_big(long n8, int C, int streams) {   // (norm.hip big_form: the launch's streams exceed the 256 MB memory-side cache)
  return n8 * 32 * streams > (256L << 20) && n8 >= 64L * 16 * PL_UPB;
}
inline int pl_big_grid(long n8) { return (int)((n8 / 64 + 1 + PL_UPB - 1) / PL_UPB); }   // units = ceil(n8 / 64)
// blocks of a streaming-form launch: at most 1024 (TRIS_PL_BIG_GRID overrides; 0 = every 4096-element piece its own block, the form of
// rounds 3-5), each walking pieces gridDim.x apart (a thread then still sees one channel group).  Same reason as pl_grid_cap:
// 34.54 -> 34.32 ms, means of three A/B rounds (profiles/r6_plane_pass_grid_ab.txt)
static int pl_big_cap() {
  static const int cap = [] { const char* e = getenv("TRIS_PL_BIG_GRID"); return e ? std::max(0, atoi(e)) : 1024; }();
  return cap;
}
inline int pl_big_blocks(long n8) {
  const int g = pl_big_grid(n8), cap = pl_big_cap();
  return cap > 0 && g > cap ? cap : g;
}
// Blocks of the grid-stride form.  Rounds 4-5 launched up to 4096 (one 512-element unit per wave: the fastest on an idle device for
// the large tensors).  Inside the step these passes run beside the weight-gradient stream's products, and 1024 blocks that each walk
// several units -- fewer workgroups to place among another stream's -- take 0.4 ms off the step (36.33 -> 35.87 ms, means of three
// A/B rounds; 768 and 2048 the same, 512 and 256 slower: profiles/r6_plane_pass_grid_ab.txt).  TRIS_PL_GRID overrides (read once).
static int pl_grid_cap() {
  static const int cap = [] { const char* e = getenv("TRIS_PL_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
  return cap;
}
inline int pl_grid(long n8, int C) {   // one unit (64 groups) per wave and trip, four waves per block
  long g = (n8 + 255) / 256;
  const int cap = pl_grid_cap();
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}
}  // namespace

// one-shot: the next tris_bn_apply_pl_f32 of the calling thread also writes the ReLU mask of its output, one byte per 8 channels
static thread_local unsigned char* g_bn_mask_next = nullptr;
extern "C" int tris_bn_mask_next(unsigned char* mask) {
  g_bn_mask_next = mask;
  return 0;
}

extern "C" int tris_bn_apply_pl_f32(const float* X, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                    const float* resid, int resid_kind, const unsigned* resid_word, float* Ypl, const unsigned* out_word,
                                    long M, int C, int relu, void* stream) {
  if (C % 8 || 2048 % C || out_word == nullptr || (resid_kind != 0 && resid == nullptr) || (resid_kind == 2 && resid_word == nullptr) ||
      resid_kind < 0 || resid_kind > 2 || !al16p(X) || !al16p(Ypl) || !al16p(resid))
    return (int)hipErrorInvalidValue;
  unsigned char* const mask = g_bn_mask_next;
  g_bn_mask_next = nullptr;
  const long n8 = M * C / 8;
  if (pl_big(n8, C, resid_kind ? 3 : 2))
    hipLaunchKernelGGL(bn_apply_pl_kernel<true>, dim3(pl_big_blocks(n8)), dim3(256), 0, (hipStream_t)stream, X, mean, invstd, gamma, beta,
                       resid, resid_kind, resid_word, Ypl, out_word, n8, C, relu, mask);
  else
    hipLaunchKernelGGL(bn_apply_pl_kernel<false>, dim3(pl_grid(n8, C)), dim3(256), 0, (hipStream_t)stream, X, mean, invstd, gamma, beta,
                       resid, resid_kind, resid_word, Ypl, out_word, n8, C, relu, mask);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_bn_apply_pool_pl_f32(const float* X, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                         float* Ypl, const unsigned* out_word, int B, int H, int W, int C, void* stream) {
  if (C % 8 || (H & 1) || (W & 1) || out_word == nullptr || !al16p(X) || !al16p(Ypl)) return (int)hipErrorInvalidValue;
  const long n = (long)B * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(bn_apply_pool_pl_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, X, mean, invstd, gamma, beta, Ypl,
                     out_word, B, H, W, C);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_avgpool2_fwd_pl_f32(const float* Xpl, float* Ypl, const unsigned* word, int B, int H, int W, int C, void* stream) {
  if (C % 8 || (H & 1) || (W & 1) || word == nullptr || !al16p(Xpl) || !al16p(Ypl)) return (int)hipErrorInvalidValue;
  const long n = (long)B * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(avgpool2_pl_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, Xpl, Ypl, word, B, H, W, C);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_bn_bwd_apply_pl_f32(const float* dY, const float* Ypl, const float* X, const float* mean, const float* invstd,
                                        const float* gamma, const float* sum_dz, const float* sum_dzx, float inv_count, float* dXpl,
                                        const unsigned* out_word, float* dZ, long M, int C, const float* beta_mask, void* stream) {
  if (C % 8 || 2048 % C || out_word == nullptr || !al16p(dY) || !al16p(X) || !al16p(dXpl) || !al16p(Ypl) || !al16p(dZ)) return (int)hipErrorInvalidValue;
  const long n8 = M * C / 8;
  const int streams = 3 + ((Ypl && !beta_mask) ? 1 : 0) + (dZ ? 1 : 0);
  if (pl_big(n8, C, streams))
    hipLaunchKernelGGL((bn_bwd_apply_pl_kernel<false, true>), dim3(pl_big_blocks(n8)), dim3(256), 0, (hipStream_t)stream, dY, Ypl, X, mean,
                       invstd, gamma, sum_dz, sum_dzx, inv_count, dXpl, out_word, dZ, n8, C, beta_mask, 0, 0);
  else
    hipLaunchKernelGGL((bn_bwd_apply_pl_kernel<false, false>), dim3(pl_grid(n8, C)), dim3(256), 0, (hipStream_t)stream, dY, Ypl, X, mean,
                       invstd, gamma, sum_dz, sum_dzx, inv_count, dXpl, out_word, dZ, n8, C, beta_mask, 0, 0);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_bn_bwd_apply_pool_pl_f32(const float* dYp, const float* X, const float* mean, const float* invstd, const float* gamma,
                                             const float* beta, const float* sum_dz, const float* sum_dzx, float inv_count, float* dXpl,
                                             const unsigned* out_word, int B, int H, int W, int C, void* stream) {
  if (C % 8 || 2048 % C || (H & 1) || (W & 1) || out_word == nullptr || beta == nullptr || !al16p(dYp) || !al16p(X) || !al16p(dXpl))
    return (int)hipErrorInvalidValue;
  const long n8 = (long)B * H * W * C / 8;
  hipLaunchKernelGGL((bn_bwd_apply_pl_kernel<true, false>), dim3(pl_grid(n8, C)), dim3(256), 0, (hipStream_t)stream, dYp,
                     (const float*)nullptr, X, mean, invstd, gamma, sum_dz, sum_dzx, inv_count, dXpl, out_word, (float*)nullptr, n8, C,
                     beta, H, W);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_bn_out_bound2_f32(const float* gamma, const float* beta, int C, float xhat_max, const unsigned* add_word,
                                      unsigned* out, void* stream) {
  if (C <= 0 || out == nullptr) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(bn_out_bound2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, gamma, beta, C, xhat_max, add_word, out);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_bn_bwd_bound_f32(const float* gamma, const float* invstd, const float* sum_dz, const float* sum_dzx, int C,
                                     float inv_count, float xhat_max, const unsigned* dz_word, unsigned* out, void* stream) {
  if (C <= 0 || out == nullptr || dz_word == nullptr) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(bn_bwd_bound_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, gamma, invstd, sum_dz, sum_dzx, C, inv_count,
                     xhat_max, dz_word, out);
  TRIS_LAUNCH_CHECK();
  return 0;
}
