// Multi-head self-attention on the f32 MFMA for any sequence length (head dim 64): the CLIP text encoder (L = 20, causal),
// the aux ViT-B/32 (L = 50) and a ViT-B/16 trunk at 320 px (L = 401).  Flash-style: no L x L matrix ever leaves the chip.
//
//   forward   grid (ceil(L/64), heads, N), 4 waves; a wave owns 16 queries.  K/V blocks of 64 keys are staged in LDS with
//             coalesced 16-byte loads; per 16-key sub-tile the wave forms S^T = K.Q^T (16 x v_mfma_f32_16x16x4_f32), so
//             that the accumulator layout (lane (r, kg) holds key 4kg+t of query r) IS the A-operand layout of the
//             following P.V product -- the probabilities never go through LDS.  Online softmax: running max / sum per
//             query, reduced across the four key groups of a query with two wavefront shuffles.
//   backward  two launches, both deterministic (no atomics):
//             dQ     same walk as the forward (queries resident, keys streamed): S^T, dP^T = V.dO^T, dS, dQ += dS.K
//             dK,dV  keys resident, queries streamed: S = Q.K^T, dP = dO.V^T, dV += P^T.dO, dK += dS^T.Q
//             P is recomputed from the saved log-sum-exp; delta = rowsum(dO * O) is produced by the dQ launch.
//
// k-slot convention: a lane's 16 consecutive channels d = 16*kg + i feed MFMA number i (slot kg <-> channel 16kg+i); both
// operands of a product use the same map, so the contraction is unchanged and every fragment is 4 x ds_read_b128.
#include "common.h"
#include "tris_hip.h"

// (one-shot arming of an amax by-product, csrc/norm.hip tris_amax_next)
extern "C" __attribute__((visibility("hidden"))) unsigned* tris_internal_take_amax_next();

namespace {

#include "amax.h"

constexpr int HD = 64;
constexpr int LDT = HD + 4;  // LDS row stride (floats): 16-byte aligned, conflict-free for 16-row b128 fragment reads

typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

struct Frag16 { float v[16]; };

__device__ __forceinline__ Frag16 lds_frag(const float* row_kg) {  // 16 consecutive floats
  Frag16 f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 u = *reinterpret_cast<const float4*>(row_kg + 4 * q);
    f.v[4 * q] = u.x; f.v[4 * q + 1] = u.y; f.v[4 * q + 2] = u.z; f.v[4 * q + 3] = u.w;
  }
  return f;
}
__device__ __forceinline__ Frag16 glb_frag(const float* p, float s) {
  Frag16 f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 u = ldg4(p + 4 * q);
    f.v[4 * q] = u.x * s; f.v[4 * q + 1] = u.y * s; f.v[4 * q + 2] = u.z * s; f.v[4 * q + 3] = u.w * s;
  }
  return f;
}
// D[i][j] = sum_d A[i][d] B[j][d] over the 64 channels: a = rows of A (this lane: row lane&15), b = rows of B
__device__ __forceinline__ f4v dot64(const Frag16& a, const Frag16& b) {
  f4v c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 16; ++i) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[i], b.v[i], c, 0, 0, 0);
  return c;
}
// acc[j] (16 x 16 tile of columns 16j..16j+15) += sum over the 16 rows k of P[.][k] * T[k][16j + r]; p[t] is this lane's
// P[row r][k = 4kg + t]; T rows live in LDS (tile row stride LDT) starting at `trow` = &T[first k][0]
__device__ __forceinline__ void accum_pv(f4v acc[4], const float p[4], const float* trow, int r, int kg) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float* vr = trow + (4 * kg + t) * LDT + r;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[t], vr[16 * j], acc[j], 0, 0, 0);
  }
}
// stage 64 rows (first row `row0`, clamped to L-1) of one of q/k/v (column offset `col`) or of a [N,L,W] tensor into LDS
__device__ __forceinline__ void stage64(float* dst, const float* base, long row_stride, int row0, int L, int tid) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = q * 256 + tid, row = idx >> 4, c4 = (idx & 15) * 4;
    *reinterpret_cast<float4*>(dst + row * LDT + c4) = ldg4(base + (long)min(row0 + row, L - 1) * row_stride + c4);
  }
}

// ------------------------------------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(256) void mha_mfma_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                           float* __restrict__ lse, int L, int W, int causal,
                                                           float scale, unsigned* __restrict__ amax) {
  __shared__ __attribute__((aligned(16))) float Ks[64 * LDT];
  __shared__ __attribute__((aligned(16))) float Vs[64 * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kg = lane >> 4;
  const int h = blockIdx.y, n = blockIdx.z, H = gridDim.y;
  const int qb = blockIdx.x * 64, q0 = qb + wave * 16;
  const float* base = qkv + (long)n * L * 3 * W + h * HD;
  const int qrow = min(q0 + r, L - 1);
  const Frag16 qf = glb_frag(base + (long)qrow * 3 * W + 16 * kg, scale);
  float m = -INFINITY, l = 0.f;
  f4v acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (f4v){0.f, 0.f, 0.f, 0.f};
  const int kend = causal ? min(L, qb + 64) : L;  // keys any query of this workgroup can see
  for (int kb = 0; kb < kend; kb += 64) {
    __syncthreads();
    stage64(Ks, base + W, 3L * W, kb, L, tid);
    stage64(Vs, base + 2 * W, 3L * W, kb, L, tid);
    __syncthreads();
    if (q0 < L) {
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const int key0 = kb + 16 * sub;
        if (key0 < kend && !(causal && key0 > q0 + 15)) {  // wave-uniform
          const f4v st = dot64(lds_frag(Ks + (16 * sub + r) * LDT + 16 * kg), qf);  // S^T[key 4kg+t][query r]
          float s[4], mx = -INFINITY;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int key = key0 + 4 * kg + t;
            s[t] = (key < L && !(causal && key > q0 + r)) ? st[t] : -INFINITY;
            mx = fmaxf(mx, s[t]);
          }
          mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          const float mn = fmaxf(m, mx);
          float p[4], rs = 0.f;
#pragma unroll
          for (int t = 0; t < 4; ++t) { p[t] = mn == -INFINITY ? 0.f : __expf(s[t] - mn); rs += p[t]; }
          rs += __shfl_xor(rs, 16, 64);
          rs += __shfl_xor(rs, 32, 64);
          const float alpha = mn == -INFINITY ? 1.f : __expf(m - mn);
          l = l * alpha + rs;
          m = mn;
#pragma unroll
          for (int t = 0; t < 4; ++t) {  // the accumulator rows are queries 4kg+t: fetch their rescale factors
            const float ar = __shfl(alpha, 4 * kg + t, 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][t] *= ar;
          }
          accum_pv(acc, p, Vs + 16 * sub * LDT, r, kg);
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
      const float li = __shfl(linv, 4 * kg + t, 64);
      const int q = q0 + 4 * kg + t;
      if (q < L) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float o = acc[j][t] * li;
          ob[(long)q * W + 16 * j + r] = o;
          am = max(am, __builtin_bit_cast(unsigned, o) & 0x7fffffffu);
        }
      }
    }
    if (kg == 0 && q0 + r < L) lse[((long)n * H + h) * L + q0 + r] = m + __logf(l);
  }
  if (amax != nullptr) amax_commit(am, amax);   // (per wave: q0 < L is wave-uniform; the amax word of `out`)
}

// ---------------------------------------------------------------------------------------------------------- backward dQ
__global__ __launch_bounds__(256) void mha_mfma_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                          const float* __restrict__ dout, const float* __restrict__ lse,
                                                          float* __restrict__ delta, float* __restrict__ dqkv, int L,
                                                          int W, int causal, float scale, unsigned* __restrict__ amax) {
  __shared__ __attribute__((aligned(16))) float Ks[64 * LDT];
  __shared__ __attribute__((aligned(16))) float Vs[64 * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kg = lane >> 4;
  const int h = blockIdx.y, n = blockIdx.z, H = gridDim.y;
  const int qb = blockIdx.x * 64, q0 = qb + wave * 16;
  const float* base = qkv + (long)n * L * 3 * W + h * HD;
  const int qrow = min(q0 + r, L - 1);
  const Frag16 qf = glb_frag(base + (long)qrow * 3 * W + 16 * kg, scale);
  const Frag16 gf = glb_frag(dout + ((long)n * L + qrow) * W + h * HD + 16 * kg, 1.f);
  float dl;
  {
    const Frag16 of = glb_frag(out + ((long)n * L + qrow) * W + h * HD + 16 * kg, 1.f);
    dl = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) dl += gf.v[i] * of.v[i];
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
    stage64(Ks, base + W, 3L * W, kb, L, tid);
    stage64(Vs, base + 2 * W, 3L * W, kb, L, tid);
    __syncthreads();
    if (q0 < L) {
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const int key0 = kb + 16 * sub;
        if (key0 < kend && !(causal && key0 > q0 + 15)) {
          const f4v st = dot64(lds_frag(Ks + (16 * sub + r) * LDT + 16 * kg), qf);  // S^T[key][query r]
          const f4v dp = dot64(lds_frag(Vs + (16 * sub + r) * LDT + 16 * kg), gf);  // dP^T[key][query r]
          float ds[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int key = key0 + 4 * kg + t;
            const bool ok = key < L && !(causal && key > q0 + r);
            ds[t] = ok ? __expf(st[t] - ls) * (dp[t] - dl) * scale : 0.f;
          }
          accum_pv(acc, ds, Ks + 16 * sub * LDT, r, kg);  // dQ[query][d] += dS[query][key] K[key][d]
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
  if (amax != nullptr) amax_commit(am, amax);   // (dqkv's amax word: this launch's dQ part, the dK / dV launch adds its own)
}

// ------------------------------------------------------------------------------------------------------ backward dK, dV
__global__ __launch_bounds__(256) void mha_mfma_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           float* __restrict__ dqkv, int L, int W, int causal,
                                                           float scale, unsigned* __restrict__ amax) {
  __shared__ __attribute__((aligned(16))) float Qs[64 * LDT];
  __shared__ __attribute__((aligned(16))) float Gs[64 * LDT];
  __shared__ float Ls[64], Ds[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kg = lane >> 4;
  const int h = blockIdx.y, n = blockIdx.z, H = gridDim.y;
  const int kb0 = blockIdx.x * 64, k0 = kb0 + wave * 16;
  const float* base = qkv + (long)n * L * 3 * W + h * HD;
  const int krow = min(k0 + r, L - 1);
  const Frag16 kf = glb_frag(base + W + (long)krow * 3 * W + 16 * kg, scale);   // scale folded into K: S = Q.(scale K)^T
  const Frag16 vf = glb_frag(base + 2 * W + (long)krow * 3 * W + 16 * kg, 1.f);
  f4v dk[4], dv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { dk[j] = (f4v){0.f, 0.f, 0.f, 0.f}; dv[j] = dk[j]; }
  const int qstart = causal ? kb0 : 0;  // queries before the first key of this workgroup see none of its keys
  for (int qb = qstart; qb < L; qb += 64) {
    __syncthreads();
    stage64(Qs, base, 3L * W, qb, L, tid);
    stage64(Gs, dout + (long)n * L * W + h * HD, (long)W, qb, L, tid);
    if (tid < 64) {
      const int q = min(qb + tid, L - 1);
      Ls[tid] = lse[((long)n * H + h) * L + q];
      Ds[tid] = delta[((long)n * H + h) * L + q];
    }
    __syncthreads();
    if (k0 < L) {
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const int qs0 = qb + 16 * sub;
        if (qs0 < L && !(causal && qs0 + 15 < k0)) {  // wave-uniform: some query of the sub-tile sees some key of the wave
          const f4v s = dot64(lds_frag(Qs + (16 * sub + r) * LDT + 16 * kg), kf);   // S[query 4kg+t][key r]
          const f4v dp = dot64(lds_frag(Gs + (16 * sub + r) * LDT + 16 * kg), vf);  // dP[query 4kg+t][key r]
          float p[4], ds[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int ql = 16 * sub + 4 * kg + t, q = qb + ql;
            const bool ok = q < L && !(causal && k0 + r > q);
            p[t] = ok ? __expf(s[t] - Ls[ql]) : 0.f;
            ds[t] = p[t] * (dp[t] - Ds[ql]) * scale;
          }
          accum_pv(dv, p, Gs + 16 * sub * LDT, r, kg);   // dV[key][d] += P[query][key] dO[query][d]
          accum_pv(dk, ds, Qs + 16 * sub * LDT, r, kg);  // dK[key][d] += dS[query][key] Q[query][d]
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
          ob[(long)k * 3 * W + W + 16 * j + r] = dk[j][t];
          ob[(long)k * 3 * W + 2 * W + 16 * j + r] = dv[j][t];
          am = max(am, max(__builtin_bit_cast(unsigned, dk[j][t]) & 0x7fffffffu, __builtin_bit_cast(unsigned, dv[j][t]) & 0x7fffffffu));
        }
      }
    }
  }
  if (amax != nullptr) amax_commit(am, amax);
}

}  // namespace

extern "C" int tris_mha_mfma_fwd_f32(const float* qkv, float* out, float* lse, int N, int L, int W, int heads, int causal,
                                     void* stream) {
  if (W != heads * HD || L < 1 || N < 1 || (((uintptr_t)qkv) & 15)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mha_mfma_fwd_kernel, dim3(cdiv(L, 64), heads, N), dim3(256), 0, (hipStream_t)stream, qkv, out, lse, L,
                     W, causal, 1.0f / sqrtf((float)HD), tris_internal_take_amax_next());
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_mha_mfma_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse,
                                     float* delta, float* dqkv, int N, int L, int W, int heads, int causal,
                                     void* stream) {
  if (W != heads * HD || L < 1 || N < 1) return (int)hipErrorInvalidValue;
  const float scale = 1.0f / sqrtf((float)HD);
  const dim3 grid(cdiv(L, 64), heads, N);
  unsigned* amax = tris_internal_take_amax_next();   // (dqkv's amax word: both launches max into it)
  hipLaunchKernelGGL(mha_mfma_dq_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, out, dout, lse, delta, dqkv, L, W,
                     causal, scale, amax);
  TRIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(mha_mfma_dkv_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, dout, lse, delta, dqkv, L, W,
                     causal, scale, amax);
  TRIS_LAUNCH_CHECK();
  return 0;
}
