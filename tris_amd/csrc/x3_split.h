// split-bf16 ("x3") helpers shared by the GEMM core (gemm_fast.h) and the fused cross attention (xattn.hip).
// An fp32 value is EXACTLY the sum of three bf16 pieces (round-to-nearest residual splitting: 8 + 8 + 8 significand bits).
#pragma once
#include <hip/hip_runtime.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 h = __builtin_convertvector(v, bf16x2);  // v_cvt_pk_bf16_f32 (RNE)
  return __builtin_bit_cast(unsigned, h);
}
struct Split8 { bf16x8 hi, mid, lo; };
__device__ __forceinline__ Split8 split8(const float4 u, const float4 w) {
  const float x[8] = {u.x, u.y, u.z, u.w, w.x, w.y, w.z, w.w};
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float x0 = x[2 * t], x1 = x[2 * t + 1];
    h[t] = pk_bf16(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h[t] << 16), r1 = x1 - __builtin_bit_cast(float, h[t] & 0xffff0000u);
    m[t] = pk_bf16(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, m[t] << 16), s1 = r1 - __builtin_bit_cast(float, m[t] & 0xffff0000u);
    l[t] = pk_bf16(s0, s1);
  }
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  Split8 o;
  o.hi = __builtin_bit_cast(bf16x8, (u32x4){h[0], h[1], h[2], h[3]});
  o.mid = __builtin_bit_cast(bf16x8, (u32x4){m[0], m[1], m[2], m[3]});
  o.lo = __builtin_bit_cast(bf16x8, (u32x4){l[0], l[1], l[2], l[3]});
  return o;
}

struct Split4 { uint2 hi, mid, lo; };
__device__ __forceinline__ Split4 split4(const float4 u) {
  Split4 o;
  o.hi.x = pk_bf16(u.x, u.y);
  o.hi.y = pk_bf16(u.z, u.w);
  const float r0 = u.x - __builtin_bit_cast(float, o.hi.x << 16), r1 = u.y - __builtin_bit_cast(float, o.hi.x & 0xffff0000u);
  const float r2 = u.z - __builtin_bit_cast(float, o.hi.y << 16), r3 = u.w - __builtin_bit_cast(float, o.hi.y & 0xffff0000u);
  o.mid.x = pk_bf16(r0, r1);
  o.mid.y = pk_bf16(r2, r3);
  o.lo.x = pk_bf16(r0 - __builtin_bit_cast(float, o.mid.x << 16), r1 - __builtin_bit_cast(float, o.mid.x & 0xffff0000u));
  o.lo.y = pk_bf16(r2 - __builtin_bit_cast(float, o.mid.y << 16), r3 - __builtin_bit_cast(float, o.mid.y & 0xffff0000u));
  return o;
}


// ---- two-piece fp16 split ("h2"): x * scale = hi + lo' * 2^-11 with hi, lo' fp16 (11 + 11 significand bits + the sign of the
// residual), both round-to-nearest.  The caller's scale is a power of two that brings the tensor's largest magnitude -- or any
// upper bound of it -- to [2^13, 2^14), so hi cannot overflow.  The residual is stored PRE-SCALED by 2^11 (lo' = (x - hi) * 2^11,
// |lo'| <= |hi|): it stays a normal fp16 number for as long as hi does, i.e. an element keeps all 22 bits down to 2^-27 of the
// tensor's maximum (an unscaled residual would go subnormal 2^-16 below it).  Below that the representation error is ABSOLUTE,
// <= 2^-36 in scaled units = 2^-49 of the maximum -- the whole scheme is fp32-class relative accuracy plus a fixed-point floor 49
// bits under the tensor's largest magnitude.  The products hi x lo' and lo' x hi are accumulated apart from hi x hi and join it
// with the factor 2^-11 in the epilogue (gemm_fast.h).  The pieces travel in the same 16-bit planes as the bf16 ones.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_f16(float a, float b) {
  f32x2 v = {a, b};
  f16x2 h = __builtin_convertvector(v, f16x2);  // v_cvt_f16_f32 (RNE) x 2
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ f32x2 unpk_f16(unsigned u) {
  return __builtin_convertvector(__builtin_bit_cast(f16x2, u), f32x2);
}
__device__ __forceinline__ Split4 split4h(const float4 u, const float s) {
  Split4 o;
  const float x0 = u.x * s, x1 = u.y * s, x2 = u.z * s, x3 = u.w * s;
  o.hi.x = pk_f16(x0, x1);
  o.hi.y = pk_f16(x2, x3);
  const f32x2 h0 = unpk_f16(o.hi.x), h1 = unpk_f16(o.hi.y);
  o.mid.x = pk_f16((x0 - h0[0]) * 2048.f, (x1 - h0[1]) * 2048.f);   // (the differences are exact in fp32)
  o.mid.y = pk_f16((x2 - h1[0]) * 2048.f, (x3 - h1[1]) * 2048.f);
  o.lo = o.mid;
  return o;
}
// an amax "word" is 128 cache lines (2048 words) with one used word each (csrc/norm.hip amax_raise)
__device__ __forceinline__ unsigned h2_amax_of(const unsigned* __restrict__ slots, int lane) {
  unsigned m = max(slots[lane * 16], slots[(lane + 64) * 16]);
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, sft, 64));
  return m;
}
// scale for a tensor whose largest magnitude has the bit pattern `amax_bits` (0: empty or all-zero tensor -> 1)
__device__ __forceinline__ float h2_scale_from_bits(unsigned amax_bits) {
  const int e = (int)((amax_bits >> 23) & 0xffu) - 127;        // floor(log2(amax)) for normal numbers
  if (amax_bits == 0u || e < -100) return 1.0f;
  int se = 13 - e;
  se = se > 120 ? 120 : (se < -120 ? -120 : se);
  return __builtin_bit_cast(float, (unsigned)(se + 127) << 23);
}
