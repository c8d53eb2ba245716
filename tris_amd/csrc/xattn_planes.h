// Sentence operands of the fused cross attention (xattn_fused.hip, xattn_px.hip) pre-split into piece planes in MFMA fragment
// order: one small launch in front of the persistent kernel, shared by both of its forms.  H2 = false: three bf16 pieces (x3);
// H2 = true: two fp16 pieces hi | lo' of x * s (x3_split.h "h2"), s from the tensor's amax word (am[0..2] = Qt, Kt, Vt).
#pragma once
#include "common.h"
#include "x3_split.h"

namespace {

__device__ __forceinline__ float4 xp_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// eight values -> h2 pieces carried in a Split8 (hi = fp16 hi pieces, mid = lo = fp16 pre-scaled residuals)
__device__ __forceinline__ Split8 split8h(const float4 u, const float4 w, const float s) {
  const Split4 p = split4h(u, s), q = split4h(w, s);
  typedef unsigned xu32x4 __attribute__((ext_vector_type(4)));
  Split8 o;
  o.hi = __builtin_bit_cast(bf16x8, (xu32x4){p.hi.x, p.hi.y, q.hi.x, q.hi.y});
  o.mid = __builtin_bit_cast(bf16x8, (xu32x4){p.mid.x, p.mid.y, q.mid.x, q.mid.y});
  o.lo = o.mid;
  return o;
}
struct XpAmax { const unsigned* w[6]; };   // amax words of Qv, Kv, Vv, Qt, Kt, Vt (h2 form)

// ---- sentence operands -> bf16 piece planes in MFMA fragment order ------------------------------------------------------------
// QtF / KtF (A operand of phase A: rows = sentences, k = channels):  [NT][C/32][3 planes][64 lanes] x 16 B;
//     lane l of fragment (j, s): sentence n = 16 j + (l & 15), channels c = 32 s + 8 (l >> 4) .. + 7      (n >= N -> zeros)
// VtF (B operand of phase B1: k = sentences, columns = channels):    [C/16][KS2][3 planes][64 lanes] x 16 B;
//     lane l of fragment (ct, ks): channel c = 16 ct + (l & 15), sentences n = 32 ks + 8 (l >> 4) .. + 7   (n >= N -> zeros)
template <bool H2 = false>
__global__ __launch_bounds__(256) void xattn_text_planes_kernel(const float* __restrict__ Qt, const float* __restrict__ Kt,
                                                                const float* __restrict__ Vt, uint4* __restrict__ QtF,
                                                                uint4* __restrict__ KtF, uint4* __restrict__ VtF, int N, int C,
                                                                int NT, int KS2, XpAmax am = XpAmax(),
                                                                float* __restrict__ scl_out = nullptr) {
  constexpr int NP = H2 ? 2 : 3;
  const int KST0 = C / 32, nA0 = NT * KST0 * 64;     // (the tensor a thread works on is wave-uniform: the slot counts are multiples of 64)
  const int g0 = blockIdx.x * 256 + threadIdx.x;
  float sc = 1.f;
  if constexpr (H2) {
    // one amax word per wave (the tensor it splits); the six scales of the persistent launch come from six waves of blocks 0 and 1
    const int which0 = g0 < nA0 ? 0 : (g0 < 2 * nA0 ? 1 : 2);
    sc = h2_scale_from_bits(h2_amax_of(am.w[3 + which0], threadIdx.x & 63));
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t < 6) {
      const float v = h2_scale_from_bits(h2_amax_of(am.w[t], threadIdx.x & 63));
      if ((threadIdx.x & 63) == 0) scl_out[t] = v;
    }
  }
  const int KST = C / 32;
  const int nA = NT * KST * 64;           // lane slots of one A-operand tensor
  const int nB = (C / 16) * KS2 * 64;     // lane slots of VtF
  const int g = blockIdx.x * 256 + threadIdx.x;
  float x[8];
  uint4* dst;
  if (g < 2 * nA) {
    const int t = g / nA, e = g - t * nA;
    const int l = e & 63, fs = e >> 6;    // fs = j * KST + s
    const int j = fs / KST, s = fs - j * KST;
    const int n = j * 16 + (l & 15), c = s * 32 + (l >> 4) * 8;
    const float* src = (t == 0 ? Qt : Kt) + (long)min(n, N - 1) * C + c;
    const float4 u = xp_ld4(src), w = xp_ld4(src + 4);
    const bool ok = n < N;
    x[0] = ok ? u.x : 0.f; x[1] = ok ? u.y : 0.f; x[2] = ok ? u.z : 0.f; x[3] = ok ? u.w : 0.f;
    x[4] = ok ? w.x : 0.f; x[5] = ok ? w.y : 0.f; x[6] = ok ? w.z : 0.f; x[7] = ok ? w.w : 0.f;
    dst = (t == 0 ? QtF : KtF) + ((long)fs * NP) * 64 + l;
  } else if (g < 2 * nA + nB) {
    const int e = g - 2 * nA;
    const int l = e & 63, fs = e >> 6;    // fs = ct * KS2 + ks
    const int ct = fs / KS2, ks = fs - ct * KS2;
    const int c = ct * 16 + (l & 15), n0 = ks * 32 + (l >> 4) * 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = (n0 + q < N) ? Vt[(long)(n0 + q) * C + c] : 0.f;
    dst = VtF + ((long)fs * NP) * 64 + l;
  } else {
    return;
  }
  if constexpr (H2) {
    const Split8 sp = split8h(make_float4(x[0], x[1], x[2], x[3]), make_float4(x[4], x[5], x[6], x[7]), sc);
    dst[0] = __builtin_bit_cast(uint4, sp.hi);
    dst[64] = __builtin_bit_cast(uint4, sp.mid);
  } else {
    const Split8 sp = split8(make_float4(x[0], x[1], x[2], x[3]), make_float4(x[4], x[5], x[6], x[7]));
    dst[0] = __builtin_bit_cast(uint4, sp.hi);
    dst[64] = __builtin_bit_cast(uint4, sp.mid);
    dst[128] = __builtin_bit_cast(uint4, sp.lo);
  }
}

}  // namespace
