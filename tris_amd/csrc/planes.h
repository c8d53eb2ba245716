// "P8" operand planes of the h2 arithmetic (x3_split.h): a tensor whose pass-through consumers are dense products is WRITTEN as
// its two fp16 pieces by the pass that produces it, so that no product has to split it again (gemm_fast.h PREC 4).  Byte geometry
// of the fp32 tensor (4 bytes per element, same row stride): every 8 consecutive elements of the contiguous dimension are 16 bytes
// of hi pieces followed by 16 bytes of pre-scaled lo pieces,
//     x[8 g + e] * s = hi[e] + lo'[e] * 2^-11,   hi = fp16(x s), lo' = fp16((x s - hi) * 2^11),
// with ONE power-of-two scale s per tensor, formed from the tensor's amax word -- its largest magnitude or any upper bound of it --
// by h2_scale_from_bits, exactly as the products form it.  Element-wise readers rebuild x with one FMA per element (pl8_join): the
// pieces hold 22 significand bits + the sign of the residual, i.e. a plane tensor IS its fp32 original rounded to the h2 operand.
// Included INSIDE a translation unit's anonymous namespace, after x3_split.h.
#pragma once

struct Pl8 { uint4 hi, lo; };
// eight consecutive elements (a = 0..3, b = 4..7), scaled by s -> the two 16-byte pieces
__device__ __forceinline__ Pl8 pl8_split(const float4 a, const float4 b, const float s) {
  const Split4 p = split4h(a, s), q = split4h(b, s);
  Pl8 o;
  o.hi = make_uint4(p.hi.x, p.hi.y, q.hi.x, q.hi.y);
  o.lo = make_uint4(p.mid.x, p.mid.y, q.mid.x, q.mid.y);
  return o;
}
// the two pieces -> eight elements; inv_s = 1 / s (a power of two: exact)
__device__ __forceinline__ void pl8_join(const uint4 hi, const uint4 lo, const float inv_s, float4& a, float4& b) {
  const f32x2 h0 = unpk_f16(hi.x), h1 = unpk_f16(hi.y), h2 = unpk_f16(hi.z), h3 = unpk_f16(hi.w);
  const f32x2 l0 = unpk_f16(lo.x), l1 = unpk_f16(lo.y), l2 = unpk_f16(lo.z), l3 = unpk_f16(lo.w);
  const float w = 1.0f / 2048.0f;
  a = make_float4(fmaf(l0[0], w, h0[0]) * inv_s, fmaf(l0[1], w, h0[1]) * inv_s, fmaf(l1[0], w, h1[0]) * inv_s, fmaf(l1[1], w, h1[1]) * inv_s);
  b = make_float4(fmaf(l2[0], w, h2[0]) * inv_s, fmaf(l2[1], w, h2[1]) * inv_s, fmaf(l3[0], w, h3[0]) * inv_s, fmaf(l3[1], w, h3[1]) * inv_s);
}
// group g (8 elements) of a plane tensor: pieces at p + 8 g floats
__device__ __forceinline__ void pl8_load(const float* __restrict__ p, long g, uint4& hi, uint4& lo) {
  const uint4* q = reinterpret_cast<const uint4*>(p + g * 8);
  hi = q[0];
  lo = q[1];
}
__device__ __forceinline__ void pl8_store(float* __restrict__ p, long g, const Pl8& v) {
  uint4* q = reinterpret_cast<uint4*>(p + g * 8);
  q[0] = v.hi;
  q[1] = v.lo;
}
// y = relu(..) held as planes is positive iff one of its pieces is non-zero: bit e of the result <-> element e
__device__ __forceinline__ unsigned pl8_positive(const uint4 hi, const uint4 lo) {
  const unsigned m[4] = {(hi.x | lo.x) & 0x7fff7fffu, (hi.y | lo.y) & 0x7fff7fffu, (hi.z | lo.z) & 0x7fff7fffu, (hi.w | lo.w) & 0x7fff7fffu};
  unsigned r = 0u;
#pragma unroll
  for (int t = 0; t < 4; ++t) r |= ((m[t] & 0xffffu) ? 1u : 0u) << (2 * t) | ((m[t] >> 16) ? 1u : 0u) << (2 * t + 1);
  return r;
}
// scale of a plane tensor from its amax word; every lane of the wave takes part (shuffles)
__device__ __forceinline__ float pl_scale(const unsigned* __restrict__ word) {
  return h2_scale_from_bits(h2_amax_of(word, threadIdx.x & 63));
}

// ---- range tell-tale ----------------------------------------------------------------------------------
// An element keeps its 22 bits only while its hi piece is a NORMAL fp16 number, i.e. |x s| >= 2^-14 = 2^-27 of the scaled bound; below
// that the representation error is absolute.  The passes that write plane tensors count the non-zero elements that fall below that
// floor and add the count into the SECOND unsigned of a line of the tensor's amax word (the consumers of the word read the first
// unsigned of each line only): per-XCD lines and workgroup-scope atomics, as for the amax itself (amax.h).  ops.h2_range_report()
// turns the counts of a step into "how many operands have more than 1 % of their elements in the absolute-accuracy regime".
__device__ __forceinline__ unsigned pl_below_floor(const float4 a, const float4 b, const float s) {
  const float f = 6.103515625e-05f;   // 2^-14
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned n = 0u;
#pragma unroll
  for (int e = 0; e < 8; ++e) n += (v[e] != 0.f && fabsf(v[e]) * s < f) ? 1u : 0u;
  return n;
}
__device__ __forceinline__ void pl_floor_commit(unsigned n, const unsigned* __restrict__ word) {   // every lane of the wave calls it
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) n += (unsigned)__shfl_xor((int)n, sft, 64);
  if ((threadIdx.x & 63) == 0 && n != 0u) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;
    unsigned* w = const_cast<unsigned*>(word) + (xcc * 16 + ((blockIdx.x >> 3) & 15)) * 16 + 1;
    __hip_atomic_fetch_add(w, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
