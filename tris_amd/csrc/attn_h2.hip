// Multi-head self-attention in the h2 arithmetic on the 16-bit MFMA (head dim 64, any sequence length): the flash-style kernels of
// attn_mfma.hip (reference CLIP/clip/model.py:400-448 through nn.MultiheadAttention) with every product on two fp16 pieces per
// operand -- x s = hi + lo' 2^-11 (x3_split.h), three v_mfma_f32_16x16x32_f16 per product, two accumulator sets -- instead of
// v_mfma_f32_16x16x4_f32: a 16 x 16 x 64 product is 6 MFMAs of 16 cycles instead of 16 of 32.  One power-of-two scale per tensor from
// its amax word (the packed qkv, which the producing product leaves behind; dO likewise), the fixed scale 2^13 for probabilities
// (<= 1), and for dS -- whose size is not known in advance -- the largest magnitude of the wave's current 16 x 32 block: a product
// needs ONE scale per operand, and a block is a product of its own (its result joins the fp32 running sum with that scale's inverse).
//
// Streamed blocks of 64 rows (keys for the forward / dQ, queries for dK / dV) are split while they are staged and live in LDS as two
// fp16 planes [64 rows][64 d] with a 144-byte row stride, which serves both ways a block is consumed:
//   * as the A operand of a "rows x resident rows" product (S^T = K Q^T, dP^T = V dO^T, S = Q K^T, dP = dO V^T): lane (r, kg) reads
//     16 bytes of row row(r) at d = 32 ks + 8 kg (ds_read_b128, conflict-free);
//   * as the B operand of a "probabilities x rows" product (P V, dS K, P^T dO, dS^T Q): k = the block's rows, gathered with
//     ds_read_b64_tr_b16 (lane (c, kg): rows 8 kg .. 8 kg + 7 of column c of a 16-column tile).
// The accumulator of a 16 x 16 output gives lane (c, kg) the rows 4 kg + t; the A operand of the following product wants k = 8 kg + i.
// The two agree if the rows of the two 16-row tiles of a 32-row step are CHOSEN as row(r) = 8 (r >> 2) + 4 j + (r & 3), j = 0, 1: the
// lane then holds rows 8 kg .. 8 kg + 7 of its column, which IS the A layout -- the probabilities never leave their registers.
#include "common.h"
#include "tris_hip.h"
#include "x3_split.h"

namespace {

#include "amax.h"

constexpr int HD = 64;
constexpr int RS = 144;            // bytes per row of a plane (64 fp16 + 16: b128 row reads and b64 transpose reads conflict-free)
constexpr int PLB = 64 * RS;       // one plane of a 64-row block
constexpr float PS = 8192.f;       // scale of the probabilities

typedef float f4v __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

struct H8 { f16x8 hi, lo; };

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ H8 split8(const float4 u, const float4 w, const float s) {
  const Split4 p = split4h(u, s), q = split4h(w, s);
  H8 o;
  o.hi = __builtin_bit_cast(f16x8, (u32x4v){p.hi.x, p.hi.y, q.hi.x, q.hi.y});
  o.lo = __builtin_bit_cast(f16x8, (u32x4v){p.mid.x, p.mid.y, q.mid.x, q.mid.y});
  return o;
}
// c += a_hi b_hi, cx += a_lo b_hi + a_hi b_lo   (value = (c + cx 2^-11) / (s_a s_b))
__device__ __forceinline__ void mm3(const H8& a, const H8& b, f4v& c, f4v& cx) {
  cx = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo, b.hi, cx, 0, 0, 0);
  cx = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.lo, cx, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c, 0, 0, 0);
}
__device__ __forceinline__ f4v join(const f4v c, const f4v cx) { return c + cx * (1.0f / 2048.0f); }

// stage 64 rows (first row `row0`, clamped to L - 1) x 64 channels of an fp32 tensor as two fp16 planes; 256 threads
__device__ __forceinline__ void stage64(char* planes, const float* base, long row_stride, int row0, int L, int tid, float s) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = q * 256 + tid, row = idx >> 4, c4 = (idx & 15) * 4;
    const Split4 sp = split4h(ldg4(base + (long)min(row0 + row, L - 1) * row_stride + c4), s);
    *reinterpret_cast<uint2*>(planes + row * RS + c4 * 2) = sp.hi;
    *reinterpret_cast<uint2*>(planes + PLB + row * RS + c4 * 2) = sp.mid;
  }
}
// A operand: 8 channels d = 32 ks + 8 kg .. of block row `row`
__device__ __forceinline__ H8 afrag(const char* planes, int row, int ks, int kg) {
  H8 o;
  o.hi = *reinterpret_cast<const f16x8*>(planes + row * RS + (32 * ks + 8 * kg) * 2);
  o.lo = *reinterpret_cast<const f16x8*>(planes + PLB + row * RS + (32 * ks + 8 * kg) * 2);
  return o;
}
// B operand by the transpose read: k = block rows rb + 8 kg .. + 7, column 16 ct + (lane & 15)
__device__ __forceinline__ f16x8 trf1(const char* plane, int rb, int ct, int r16, int kg) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const char* a = plane + (rb + kg * 8 + (r16 >> 2)) * RS + ct * 32 + 8 * (r16 & 3);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * RS));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(f16x8, v);
}
__device__ __forceinline__ H8 bfrag(const char* planes, int rb, int ct, int r16, int kg) {
  H8 o;
  o.hi = trf1(planes, rb, ct, r16, kg);
  o.lo = trf1(planes + PLB, rb, ct, r16, kg);
  return o;
}
// the wave's resident rows as B operand fragments (two 32-channel steps), from global memory
__device__ __forceinline__ void resident(H8 (&f)[2], const float* row_kg, float s) {   // row_kg = &row[8 kg]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) f[ks] = split8(ldg4(row_kg + 32 * ks), ldg4(row_kg + 32 * ks + 4), s);
}
// tile row of lane r in the j-th 16-row tile of a 32-row step (see the header)
__device__ __forceinline__ int trow(int r, int j) { return 8 * (r >> 2) + 4 * j + (r & 3); }
__device__ __forceinline__ float wave_scale_of(float m) {   // power-of-two h2 scale for magnitudes <= the wave's largest m
  unsigned u = __builtin_bit_cast(unsigned, m);
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, sft, 64));
  return h2_scale_from_bits(u);
}

// ------------------------------------------------------------------------------------------------------------- forward
// grid (ceil(L/64), heads, N), 4 waves; a wave owns 16 queries
__global__ __launch_bounds__(256) void mha_h2_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse,
                                                         int L, int W, int causal, float scale, const unsigned* __restrict__ am_qkv,
                                                         unsigned* __restrict__ amax) {
  __shared__ __attribute__((aligned(16))) char Kp[2 * PLB];
  __shared__ __attribute__((aligned(16))) char Vp[2 * PLB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kg = lane >> 4;
  const int h = blockIdx.y, n = blockIdx.z, H = gridDim.y;
  const int qb = blockIdx.x * 64, q0 = qb + wave * 16;
  const float s = h2_scale_from_bits(h2_amax_of(am_qkv, lane));
  const float inv_s = scale / (s * s), inv_o = 1.0f / (PS * s);
  const float* base = qkv + (long)n * L * 3 * W + h * HD;
  const int qrow = min(q0 + r, L - 1);
  H8 qf[2];
  resident(qf, base + (long)qrow * 3 * W + 8 * kg, s);
  float m = -INFINITY, l = 0.f;
  f4v oc[4], ox[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) oc[j] = ox[j] = (f4v){0.f, 0.f, 0.f, 0.f};
  const int kend = causal ? min(L, qb + 64) : L;
  for (int kb = 0; kb < kend; kb += 64) {
    __syncthreads();
    stage64(Kp, base + W, 3L * W, kb, L, tid, s);
    stage64(Vp, base + 2 * W, 3L * W, kb, L, tid, s);
    __syncthreads();
    if (q0 < L) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int key0 = kb + 32 * pr;
        if (key0 < kend && !(causal && key0 > q0 + 15)) {  // wave-uniform
          f4v sc[2], sx[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            sc[j] = sx[j] = (f4v){0.f, 0.f, 0.f, 0.f};
            const int row = 32 * pr + trow(r, j);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) mm3(afrag(Kp, row, ks, kg), qf[ks], sc[j], sx[j]);   // S^T[key 8kg + 4j + t][query r]
          }
          float sv[8], mx = -INFINITY;
          const bool full = key0 + 32 <= L && !(causal && key0 + 31 > q0);   // (wave-uniform) every key of the step seen by every query
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int key = key0 + 8 * kg + i;
            const float v = (sc[i >> 2][i & 3] + sx[i >> 2][i & 3] * (1.0f / 2048.0f)) * inv_s;
            sv[i] = (full || (key < L && !(causal && key > q0 + r))) ? v : -INFINITY;
            mx = fmaxf(mx, sv[i]);
          }
          mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          const float mn = fmaxf(m, mx);
          float p[8], rs = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) { p[i] = mn == -INFINITY ? 0.f : __expf(sv[i] - mn); rs += p[i]; }
          rs += __shfl_xor(rs, 16, 64);
          rs += __shfl_xor(rs, 32, 64);
          const float alpha = mn == -INFINITY ? 1.f : __expf(m - mn);
          l = l * alpha + rs;
          m = mn;
          if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {   // (no running maximum moved: nothing to rescale)
#pragma unroll
            for (int t = 0; t < 4; ++t) {  // the accumulator rows are queries 4kg+t: fetch their rescale factors
              const float ar = __shfl(alpha, 4 * kg + t, 64);
#pragma unroll
              for (int j = 0; j < 4; ++j) { oc[j][t] *= ar; ox[j][t] *= ar; }
            }
          }
          const H8 pf = split8(make_float4(p[0], p[1], p[2], p[3]), make_float4(p[4], p[5], p[6], p[7]), PS);
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) mm3(pf, bfrag(Vp, 32 * pr, ct, r, kg), oc[ct], ox[ct]);   // O[query][d] += P V
        }
      }
    }
  }
  unsigned am = 0u;
  if (q0 < L) {
    const float linv = 1.f / l;
    float* ob = out + (long)n * L * W + h * HD;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float li = __shfl(linv, 4 * kg + t, 64) * inv_o;
      const int q = q0 + 4 * kg + t;
      if (q < L) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float o = (oc[j][t] + ox[j][t] * (1.0f / 2048.0f)) * li;
          ob[(long)q * W + 16 * j + r] = o;
          am = max(am, __builtin_bit_cast(unsigned, o) & 0x7fffffffu);
        }
      }
    }
    if (kg == 0 && q0 + r < L) lse[((long)n * H + h) * L + q0 + r] = m + __logf(l);
  }
  if (amax != nullptr) amax_commit(am, amax);
}

// ---------------------------------------------------------------------------------------------------------- backward dQ
__global__ __launch_bounds__(256) void mha_h2_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                        const float* __restrict__ dout, const float* __restrict__ lse,
                                                        float* __restrict__ delta, float* __restrict__ dqkv, int L, int W, int causal,
                                                        float scale, const unsigned* __restrict__ am_qkv,
                                                        const unsigned* __restrict__ am_do, unsigned* __restrict__ amax) {
  __shared__ __attribute__((aligned(16))) char Kp[2 * PLB];
  __shared__ __attribute__((aligned(16))) char Vp[2 * PLB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kg = lane >> 4;
  const int h = blockIdx.y, n = blockIdx.z, H = gridDim.y;
  const int qb = blockIdx.x * 64, q0 = qb + wave * 16;
  const float s = h2_scale_from_bits(h2_amax_of(am_qkv, lane)), sg = h2_scale_from_bits(h2_amax_of(am_do, lane));
  const float inv_s = scale / (s * s), inv_p = 1.0f / (s * sg);
  const float* base = qkv + (long)n * L * 3 * W + h * HD;
  const int qrow = min(q0 + r, L - 1);
  H8 qf[2], gf[2];
  resident(qf, base + (long)qrow * 3 * W + 8 * kg, s);
  float dl = 0.f;
  {
    const float* gp = dout + ((long)n * L + qrow) * W + h * HD + 8 * kg;
    const float* op = out + ((long)n * L + qrow) * W + h * HD + 8 * kg;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float4 g0 = ldg4(gp + 32 * ks), g1 = ldg4(gp + 32 * ks + 4), o0 = ldg4(op + 32 * ks), o1 = ldg4(op + 32 * ks + 4);
      gf[ks] = split8(g0, g1, sg);
      dl += g0.x * o0.x + g0.y * o0.y + g0.z * o0.z + g0.w * o0.w + g1.x * o1.x + g1.y * o1.y + g1.z * o1.z + g1.w * o1.w;
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    if (kg == 0 && q0 + r < L) delta[((long)n * H + h) * L + q0 + r] = dl;
  }
  const float ls = lse[((long)n * H + h) * L + qrow];
  f4v acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (f4v){0.f, 0.f, 0.f, 0.f};
  const int kend = causal ? min(L, qb + 64) : L;
  for (int kb = 0; kb < kend; kb += 64) {
    __syncthreads();
    stage64(Kp, base + W, 3L * W, kb, L, tid, s);
    stage64(Vp, base + 2 * W, 3L * W, kb, L, tid, s);
    __syncthreads();
    if (q0 < L) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int key0 = kb + 32 * pr;
        if (key0 < kend && !(causal && key0 > q0 + 15)) {
          f4v sc[2], sx[2], pc[2], px[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            sc[j] = sx[j] = pc[j] = px[j] = (f4v){0.f, 0.f, 0.f, 0.f};
            const int row = 32 * pr + trow(r, j);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              mm3(afrag(Kp, row, ks, kg), qf[ks], sc[j], sx[j]);   // S^T[key][query r]
              mm3(afrag(Vp, row, ks, kg), gf[ks], pc[j], px[j]);   // dP^T[key][query r]
            }
          }
          float ds[8], mx = 0.f;
          const bool full = key0 + 32 <= L && !(causal && key0 + 31 > q0);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int key = key0 + 8 * kg + i;
            const bool ok = full || (key < L && !(causal && key > q0 + r));
            const float sv = (sc[i >> 2][i & 3] + sx[i >> 2][i & 3] * (1.0f / 2048.0f)) * inv_s;
            const float dp = (pc[i >> 2][i & 3] + px[i >> 2][i & 3] * (1.0f / 2048.0f)) * inv_p;
            ds[i] = ok ? __expf(sv - ls) * (dp - dl) * scale : 0.f;
            mx = fmaxf(mx, fabsf(ds[i]));
          }
          const float sd = wave_scale_of(mx);
          const H8 df = split8(make_float4(ds[0], ds[1], ds[2], ds[3]), make_float4(ds[4], ds[5], ds[6], ds[7]), sd);
          const float inv_d = 1.0f / (sd * s);
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {   // dQ[query][d] += dS[query][key] K[key][d]
            f4v tc = {0.f, 0.f, 0.f, 0.f}, tx = {0.f, 0.f, 0.f, 0.f};
            mm3(df, bfrag(Kp, 32 * pr, ct, r, kg), tc, tx);
            acc[ct] += join(tc, tx) * inv_d;
          }
        }
      }
    }
  }
  unsigned am = 0u;
  if (q0 < L) {
    float* ob = dqkv + (long)n * L * 3 * W + h * HD;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int q = q0 + 4 * kg + t;
      if (q < L) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ob[(long)q * 3 * W + 16 * j + r] = acc[j][t];
          am = max(am, __builtin_bit_cast(unsigned, acc[j][t]) & 0x7fffffffu);
        }
      }
    }
  }
  if (amax != nullptr) amax_commit(am, amax);
}

// ------------------------------------------------------------------------------------------------------ backward dK, dV
__global__ __launch_bounds__(256) void mha_h2_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                         const float* __restrict__ lse, const float* __restrict__ delta,
                                                         float* __restrict__ dqkv, int L, int W, int causal, float scale,
                                                         const unsigned* __restrict__ am_qkv, const unsigned* __restrict__ am_do,
                                                         unsigned* __restrict__ amax) {
  __shared__ __attribute__((aligned(16))) char Qp[2 * PLB];
  __shared__ __attribute__((aligned(16))) char Gp[2 * PLB];
  __shared__ float Ls[64], Ds[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kg = lane >> 4;
  const int h = blockIdx.y, n = blockIdx.z, H = gridDim.y;
  const int kb0 = blockIdx.x * 64, k0 = kb0 + wave * 16;
  const float s = h2_scale_from_bits(h2_amax_of(am_qkv, lane)), sg = h2_scale_from_bits(h2_amax_of(am_do, lane));
  const float inv_s = scale / (s * s), inv_p = 1.0f / (s * sg), inv_v = 1.0f / (PS * sg);
  const float* base = qkv + (long)n * L * 3 * W + h * HD;
  const int krow = min(k0 + r, L - 1);
  H8 kf[2], vf[2];
  resident(kf, base + W + (long)krow * 3 * W + 8 * kg, s);
  resident(vf, base + 2 * W + (long)krow * 3 * W + 8 * kg, s);
  f4v dk[4], vc[4], vx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) dk[j] = vc[j] = vx[j] = (f4v){0.f, 0.f, 0.f, 0.f};
  const int qstart = causal ? kb0 : 0;
  for (int qb = qstart; qb < L; qb += 64) {
    __syncthreads();
    stage64(Qp, base, 3L * W, qb, L, tid, s);
    stage64(Gp, dout + (long)n * L * W + h * HD, (long)W, qb, L, tid, sg);
    if (tid < 64) {
      const int q = min(qb + tid, L - 1);
      Ls[tid] = lse[((long)n * H + h) * L + q];
      Ds[tid] = delta[((long)n * H + h) * L + q];
    }
    __syncthreads();
    if (k0 < L) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int qs0 = qb + 32 * pr;
        if (qs0 < L && !(causal && qs0 + 31 < k0)) {  // wave-uniform: some query of the step sees some key of the wave
          f4v sc[2], sx[2], pc[2], px[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            sc[j] = sx[j] = pc[j] = px[j] = (f4v){0.f, 0.f, 0.f, 0.f};
            const int row = 32 * pr + trow(r, j);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              mm3(afrag(Qp, row, ks, kg), kf[ks], sc[j], sx[j]);   // S[query 8kg + 4j + t][key r]
              mm3(afrag(Gp, row, ks, kg), vf[ks], pc[j], px[j]);   // dP[query][key r]
            }
          }
          float p[8], ds[8], mx = 0.f;
          const bool full = qs0 + 32 <= L && !(causal && k0 + 15 > qs0);   // every query of the step sees every key of the wave
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int ql = 32 * pr + 8 * kg + i, q = qb + ql;
            const bool ok = full || (q < L && !(causal && k0 + r > q));
            const float sv = (sc[i >> 2][i & 3] + sx[i >> 2][i & 3] * (1.0f / 2048.0f)) * inv_s;
            const float dp = (pc[i >> 2][i & 3] + px[i >> 2][i & 3] * (1.0f / 2048.0f)) * inv_p;
            p[i] = ok ? __expf(sv - Ls[ql]) : 0.f;
            ds[i] = p[i] * (dp - Ds[ql]) * scale;
            mx = fmaxf(mx, fabsf(ds[i]));
          }
          const float sd = wave_scale_of(mx);
          const H8 pf = split8(make_float4(p[0], p[1], p[2], p[3]), make_float4(p[4], p[5], p[6], p[7]), PS);
          const H8 df = split8(make_float4(ds[0], ds[1], ds[2], ds[3]), make_float4(ds[4], ds[5], ds[6], ds[7]), sd);
          const float inv_d = 1.0f / (sd * s);
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            mm3(pf, bfrag(Gp, 32 * pr, ct, r, kg), vc[ct], vx[ct]);   // dV[key][d] += P[query][key] dO[query][d]
            f4v tc = {0.f, 0.f, 0.f, 0.f}, tx = {0.f, 0.f, 0.f, 0.f};
            mm3(df, bfrag(Qp, 32 * pr, ct, r, kg), tc, tx);           // dK[key][d] += dS[query][key] Q[query][d]
            dk[ct] += join(tc, tx) * inv_d;
          }
        }
      }
    }
  }
  unsigned am = 0u;
  if (k0 < L) {
    float* ob = dqkv + (long)n * L * 3 * W + h * HD;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = k0 + 4 * kg + t;
      if (k < L) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float dv = (vc[j][t] + vx[j][t] * (1.0f / 2048.0f)) * inv_v;
          ob[(long)k * 3 * W + W + 16 * j + r] = dk[j][t];
          ob[(long)k * 3 * W + 2 * W + 16 * j + r] = dv;
          am = max(am, max(__builtin_bit_cast(unsigned, dk[j][t]) & 0x7fffffffu, __builtin_bit_cast(unsigned, dv) & 0x7fffffffu));
        }
      }
    }
  }
  if (amax != nullptr) amax_commit(am, amax);
}

}  // namespace

// (one-shot arming of an amax by-product, csrc/norm.hip tris_amax_next)
extern "C" __attribute__((visibility("hidden"))) unsigned* tris_internal_take_amax_next();

extern "C" int tris_mha_h2_fwd_f32(const float* qkv, float* out, float* lse, const unsigned* amax_qkv, int N, int L, int W, int heads,
                                   int causal, void* stream) {
  if (W != heads * HD || L < 1 || N < 1 || (((uintptr_t)qkv) & 15) || amax_qkv == nullptr) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mha_h2_fwd_kernel, dim3(cdiv(L, 64), heads, N), dim3(256), 0, (hipStream_t)stream, qkv, out, lse, L, W, causal,
                     1.0f / sqrtf((float)HD), amax_qkv, tris_internal_take_amax_next());
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_mha_h2_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* delta, float* dqkv,
                                   const unsigned* amax_qkv, const unsigned* amax_dout, int N, int L, int W, int heads, int causal,
                                   void* stream) {
  if (W != heads * HD || L < 1 || N < 1 || amax_qkv == nullptr || amax_dout == nullptr || (((uintptr_t)qkv) & 15) ||
      (((uintptr_t)dout) & 15) || (((uintptr_t)out) & 15))
    return (int)hipErrorInvalidValue;
  const float scale = 1.0f / sqrtf((float)HD);
  const dim3 grid(cdiv(L, 64), heads, N);
  unsigned* amax = tris_internal_take_amax_next();   // (dqkv's amax word: both launches max into it)
  hipLaunchKernelGGL(mha_h2_dq_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, out, dout, lse, delta, dqkv, L, W, causal, scale,
                     amax_qkv, amax_dout, amax);
  TRIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(mha_h2_dkv_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, dout, lse, delta, dqkv, L, W, causal, scale,
                     amax_qkv, amax_dout, amax);
  TRIS_LAUNCH_CHECK();
  return 0;
}
