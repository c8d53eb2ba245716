// Transformer-side kernels: fused multi-head self-attention for short sequences (L <= 64, head dim 64),
// token embedding, EOT-row gather.  (CLIP text encoder L=20 causal; aux ViT-B/32 L=50 unmasked.)
//
// One workgroup per (head, sequence): Q/K/V tiles of the packed QKV activation are staged in LDS (row pad 1 ->
// conflict-free column walks), scores and probabilities live in LDS, the row softmax is a 64-lane wavefront
// shuffle reduction.  The FLOPs here are ~1% of the step, so this is VALU f32, not MFMA.
#include <algorithm>

#include "common.h"
#include "tris_hip.h"

// (one-shot arming of an amax by-product, csrc/norm.hip tris_amax_next)
extern "C" __attribute__((visibility("hidden"))) unsigned* tris_internal_take_amax_next();

namespace {

#include "amax.h"

constexpr int HD = 64;       // head dim
constexpr int HP = HD + 4;   // padded LDS row: 68 floats = 17 sixteen-byte slots -> ds_read_b128 of 16 consecutive rows is
                             // conflict-free (17 j mod 16 is a permutation) and rows stay 16-byte aligned

__device__ __forceinline__ float4 lds4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ void fma4(float4& acc, float s, const float4 v) {
  acc.x += s * v.x; acc.y += s * v.y; acc.z += s * v.z; acc.w += s * v.w;
}

// qkv: [N, L, 3W] packed (q | k | v), out: [N, L, W].  All LDS traffic is 16-byte wide: scores read q/k rows as float4,
// the output phase gives each thread 4 consecutive channels of one row.
__global__ __launch_bounds__(256) void mha_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, int L,
                                                      int W, int causal, float scale, unsigned* __restrict__ amax) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* q = sm;
  float* k = q + L * HP;
  float* v = k + L * HP;
  float* s = v + L * HP;  // [L][L+1]
  const int LP = L + 1;
  const int h = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const float* base = qkv + (long)n * L * 3 * W + h * HD;
  for (int idx = tid; idx < L * (HD / 4); idx += 256) {
    const int l = idx >> 4, d = (idx & 15) * 4;
    const float* r = base + (long)l * 3 * W + d;
    *reinterpret_cast<float4*>(&q[l * HP + d]) = *reinterpret_cast<const float4*>(r);
    *reinterpret_cast<float4*>(&k[l * HP + d]) = *reinterpret_cast<const float4*>(r + W);
    *reinterpret_cast<float4*>(&v[l * HP + d]) = *reinterpret_cast<const float4*>(r + 2 * W);
  }
  __syncthreads();
  for (int idx = tid; idx < L * L; idx += 256) {
    const int i = idx / L, j = idx - i * L;
    float acc = -INFINITY;
    if (!causal || j <= i) {
      acc = 0.f;
#pragma unroll
      for (int d = 0; d < HD; d += 4) acc += dot4(lds4(&q[i * HP + d]), lds4(&k[j * HP + d]));
      acc *= scale;
    }
    s[i * LP + j] = acc;
  }
  __syncthreads();
  const int lane = tid & 63, wv = tid >> 6;
  for (int i = wv; i < L; i += 4) {
    float x = lane < L ? s[i * LP + lane] : -INFINITY;
    float m = wave_max(x);
    float e = lane < L ? expf(x - m) : 0.f;
    float sum = wave_sum(e);
    if (lane < L) s[i * LP + lane] = e / sum;
  }
  __syncthreads();
  float* ob = out + (long)n * L * W + h * HD;
  unsigned am = 0u;
  for (int idx = tid; idx < L * (HD / 4); idx += 256) {
    const int i = idx >> 4, d = (idx & 15) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < L; ++j) fma4(acc, s[i * LP + j], lds4(&v[j * HP + d]));
    *reinterpret_cast<float4*>(&ob[(long)i * W + d]) = acc;
    am = max(am, abits4(acc));
  }
  if (amax != nullptr) amax_commit_block(am, amax);   // (the amax word of `out`: operand scale of the h2 product that consumes it)
}

// dqkv: [N, L, 3W]; recomputes P from q,k.
__global__ __launch_bounds__(256) void mha_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                      float* __restrict__ dqkv, int L, int W, int causal, float scale,
                                                      unsigned* __restrict__ amax) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* q = sm;
  float* k = q + L * HP;
  float* v = k + L * HP;
  float* g = v + L * HP;   // dO
  float* s = g + L * HP;   // P  [L][L+1]
  float* ds = s + L * (L + 1);  // dP -> dS
  const int LP = L + 1;
  const int h = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const float* base = qkv + (long)n * L * 3 * W + h * HD;
  const float* gb = dout + (long)n * L * W + h * HD;
  for (int idx = tid; idx < L * (HD / 4); idx += 256) {
    const int l = idx >> 4, d = (idx & 15) * 4;
    const float* r = base + (long)l * 3 * W + d;
    *reinterpret_cast<float4*>(&q[l * HP + d]) = *reinterpret_cast<const float4*>(r);
    *reinterpret_cast<float4*>(&k[l * HP + d]) = *reinterpret_cast<const float4*>(r + W);
    *reinterpret_cast<float4*>(&v[l * HP + d]) = *reinterpret_cast<const float4*>(r + 2 * W);
    *reinterpret_cast<float4*>(&g[l * HP + d]) = *reinterpret_cast<const float4*>(gb + (long)l * W + d);
  }
  __syncthreads();
  for (int idx = tid; idx < L * L; idx += 256) {
    const int i = idx / L, j = idx - i * L;
    float acc = -INFINITY, dp = 0.f;
    if (!causal || j <= i) {
      acc = 0.f;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        acc += dot4(lds4(&q[i * HP + d]), lds4(&k[j * HP + d]));
        dp += dot4(lds4(&g[i * HP + d]), lds4(&v[j * HP + d]));
      }
      acc *= scale;
    }
    s[i * LP + j] = acc;
    ds[i * LP + j] = dp;
  }
  __syncthreads();
  const int lane = tid & 63, wv = tid >> 6;
  for (int i = wv; i < L; i += 4) {
    float x = lane < L ? s[i * LP + lane] : -INFINITY;
    float m = wave_max(x);
    float e = lane < L ? expf(x - m) : 0.f;
    float sum = wave_sum(e);
    float p = e / sum;
    float dp = lane < L ? ds[i * LP + lane] : 0.f;
    float delta = wave_sum(p * dp);
    if (lane < L) {
      s[i * LP + lane] = p;
      ds[i * LP + lane] = p * (dp - delta) * scale;  // dS, pre-multiplied by the 1/sqrt(d) of the logits
    }
  }
  __syncthreads();
  float* ob = dqkv + (long)n * L * 3 * W + h * HD;
  unsigned am = 0u;
  for (int idx = tid; idx < L * (HD / 4); idx += 256) {
    const int i = idx >> 4, d = (idx & 15) * 4;
    float4 dq = make_float4(0.f, 0.f, 0.f, 0.f), dk = dq, dv = dq;
    for (int j = 0; j < L; ++j) {
      fma4(dq, ds[i * LP + j], lds4(&k[j * HP + d]));
      fma4(dk, ds[j * LP + i], lds4(&q[j * HP + d]));
      fma4(dv, s[j * LP + i], lds4(&g[j * HP + d]));
    }
    float* r = ob + (long)i * 3 * W + d;
    *reinterpret_cast<float4*>(r) = dq;
    *reinterpret_cast<float4*>(r + W) = dk;
    *reinterpret_cast<float4*>(r + 2 * W) = dv;
    am = max(am, max(abits4(dq), max(abits4(dk), abits4(dv))));
  }
  if (amax != nullptr) amax_commit_block(am, amax);
}

__global__ void embed_fwd_kernel(const long* __restrict__ ids, const float* __restrict__ tok,
                                 const float* __restrict__ pos, float* __restrict__ out, long NL, int L, int W) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int W4 = W >> 2;
  if (i >= NL * W4) return;
  long t = i / W4;
  int c = (int)(i - t * W4) * 4;
  int l = (int)(t % L);
  float4 a = *reinterpret_cast<const float4*>(tok + ids[t] * W + c);
  float4 b = *reinterpret_cast<const float4*>(pos + (long)l * W + c);
  *reinterpret_cast<float4*>(out + t * W + c) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// dTok must be zero-filled by the caller; dPos[l] = sum_n dOut[n,l] (deterministic, rows >= L untouched).
__global__ void embed_bwd_kernel(const long* __restrict__ ids, const float* __restrict__ dout, float* __restrict__ dtok,
                                 float* __restrict__ dpos, int N, int L, int W) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long NLW = (long)N * L * W;
  if (dtok != nullptr && i < NLW) {   // (dtok == NULL: the token part is done by embed_rows_bwd_kernel)
    long t = i / W;
    int c = (int)(i - t * W);
    atomicAdd(dtok + ids[t] * W + c, dout[i]);
  }
  if (dpos != nullptr && i < (long)L * W) {
    int l = (int)(i / W), c = (int)(i - (long)l * W);
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dout[((long)n * L + l) * W + c];
    dpos[(long)l * W + c] = s;
  }
}

// Token-embedding gradient from a ROW LIST, deterministic (no atomics): dtok[ids[t]] = scale * sum over the list positions t'
// with ids[t'] == ids[t], added in list order.  One workgroup per list position t; only the FIRST occurrence of an id does the
// work (it scans the rest of the list for repeats -- SOT / EOT / padding ids repeat in every sentence).  The row list may be
// this rank's own R = N*L positions or the concatenation of every rank's (sparse data-parallel exchange: identical input in
// identical order on every rank -> bit-identical token-embedding gradients on every rank).  dtok must be zero-filled.
__global__ __launch_bounds__(128) void embed_rows_bwd_kernel(const long* __restrict__ ids, const float* __restrict__ rows,
                                                             float* __restrict__ dtok, int R, int W, float scale) {
  // The list is walked in chunks of CH positions (an LDS bitmap of the repeats per chunk), so R is not bounded by the bitmap:
  // 16 ranks x 64 sentences x 20 tokens = 20480 positions must work.  W <= 4 * 128 * MAXC.
  constexpr int MAXW = 512, CH = MAXW * 32, MAXC = 4;
  __shared__ unsigned bm[MAXW];        // bit u - base: ids[u] == ids[t]   (u > t)
  __shared__ int s_first, s_any;
  const int t = blockIdx.x;
  const long id = ids[t];
  float4 acc[MAXC];
#pragma unroll
  for (int ci = 0; ci < MAXC; ++ci) {
    const int c = (threadIdx.x + ci * blockDim.x) * 4;
    acc[ci] = c < W ? *reinterpret_cast<const float4*>(rows + (long)t * W + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (threadIdx.x == 0) s_first = 1;
  for (int base = 0; base < R; base += CH) {
    const int n = min(CH, R - base), nw = (n + 31) >> 5;
    __syncthreads();                   // (the previous chunk's bitmap has been consumed)
    for (int w = threadIdx.x; w < nw; w += blockDim.x) bm[w] = 0u;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    for (int u = base + threadIdx.x; u < base + n; u += blockDim.x) {
      if (u != t && ids[u] == id) {
        if (u < t) s_first = 0;        // (benign race: every writer stores the same value)
        else { atomicOr(&bm[(u - base) >> 5], 1u << ((u - base) & 31)); s_any = 1; }
      }
    }
    __syncthreads();
    if (!s_first) return;              // a later position of an id already summed by its first occurrence (uniform)
    if (s_any) {
#pragma unroll
      for (int ci = 0; ci < MAXC; ++ci) {
        const int c = (threadIdx.x + ci * blockDim.x) * 4;
        if (c >= W) continue;
        for (int w = max(0, (t - base) >> 5); w < nw; ++w) {   // ascending list order: the same sum on every rank and every run
          unsigned m = bm[w];
          while (m) {
            const int u = base + (w << 5) + __builtin_ctz(m);
            m &= m - 1;
            const float4 v = *reinterpret_cast<const float4*>(rows + (long)u * W + c);
            acc[ci].x += v.x; acc[ci].y += v.y; acc[ci].z += v.z; acc[ci].w += v.w;
          }
        }
      }
    }
  }
#pragma unroll
  for (int ci = 0; ci < MAXC; ++ci) {
    const int c = (threadIdx.x + ci * blockDim.x) * 4;
    if (c < W) {
      float4 a = acc[ci];
      a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale;
      *reinterpret_cast<float4*>(dtok + id * W + c) = a;
    }
  }
}

// out[n] = x[n, argmax_l ids[n,l]]  (first maximal index, as torch.argmax)
__global__ void eot_gather_kernel(const long* __restrict__ ids, const float* __restrict__ x, float* __restrict__ out,
                                  float* __restrict__ dx, const float* __restrict__ dout, int L, int W) {
  const int n = blockIdx.x;
  int best = 0;
  if (ids != nullptr) {   // (ids == NULL: row 0 of every sequence -- the ViT class token)
    long bv = ids[(long)n * L];
    for (int l = 1; l < L; ++l) {
      long vv = ids[(long)n * L + l];
      if (vv > bv) { bv = vv; best = l; }
    }
  }
  if (out) {
    for (int c = threadIdx.x; c < W; c += blockDim.x) out[(long)n * W + c] = x[((long)n * L + best) * W + c];
  } else {  // backward: dx is [N,L,W], all rows zero except the gathered one
    for (int i = threadIdx.x; i < L * W; i += blockDim.x) {
      int l = i / W, c = i - l * W;
      dx[(long)n * L * W + i] = (l == best) ? dout[(long)n * W + c] : 0.f;
    }
  }
}

// ---- packed text rows (round 6; reference CLIP/clip/model.py:537-564) -------------------------------------------------------------
// A causal text tower is read at the EOT row of every sentence only, and under the causal mask nothing behind that row can reach it
// (tests/test_oracle_golden.py::test_tokens_behind_eot_cannot_reach_hidden): the rows behind EOT -- 42 % of a RefCOCOg-shaped batch --
// need not exist.  The packed pass keeps the rows 0 .. eot_n of sentence n back to back.  All extents live in a device array `plan`
// so that a captured pass stays valid for any token ids:
//   plan[0] = P   rows in use            plan[1] = P rounded up to 256 = the ROW LIMIT of the row-wise launches (tris_rows_limit_thread)
//   plan[2 + n], n = 0 .. N: first packed row of sentence n (plan[2 + N] = P)
//   plan[3 + N + r], r = 0 .. R - 1 (R = N L rounded up to 256 = the row count of every packed buffer): the source token n L + l of
//   packed row r, -1 for r >= P
// Rows P .. plan[1] - 1 are kept ZERO at the input (and by the attention kernel): row-wise launches compute them like any row --
// finite values of ordinary size, never read by a sentence -- so tiles and amax by-products need no masks; rows >= plan[1] are never
// touched.
__global__ __launch_bounds__(256) void text_pack_plan_kernel(const long* __restrict__ ids, int N, int L, int* __restrict__ plan) {
  extern __shared__ int len[];     // [N]
  for (int n = threadIdx.x; n < N; n += 256) {
    int best = 0;
    long bv = ids[(long)n * L];
    for (int l = 1; l < L; ++l) {
      const long vv = ids[(long)n * L + l];
      if (vv > bv) { bv = vv; best = l; }
    }
    len[n] = best + 1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int n = 0; n < N; ++n) { plan[2 + n] = acc; acc += len[n]; }
    plan[2 + N] = acc;
    plan[0] = acc;
    plan[1] = ((acc + 255) / 256) * 256;
  }
  __syncthreads();
  int* map = plan + 3 + N;
  const int P = plan[0];
  const int R = ((N * L + 255) / 256) * 256;      // the packed buffers have R rows: plan[1] <= R
  for (int r = P + threadIdx.x; r < R; r += 256) map[r] = -1;
  for (int n = threadIdx.x; n < N; n += 256) {
    const int o = plan[2 + n];
    for (int l = 0; l < len[n]; ++l) map[o + l] = n * L + l;
  }
}

__global__ void embed_packed_fwd_kernel(const long* __restrict__ ids, const float* __restrict__ tok, const float* __restrict__ pos,
                                        const int* __restrict__ plan, float* __restrict__ out, long NL, int N, int L, int W) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int W4 = W >> 2;
  if (i >= NL * W4) return;
  const long r = i / W4;
  if (r >= plan[1]) return;
  const int c = (int)(i - r * W4) * 4;
  const int t = plan[3 + N + r];
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t >= 0) {
    const float4 a = *reinterpret_cast<const float4*>(tok + ids[t] * W + c);
    const float4 b = *reinterpret_cast<const float4*>(pos + (long)(t % L) * W + c);
    o = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
  *reinterpret_cast<float4*>(out + r * W + c) = o;
}

// mha_fwd_kernel over packed rows: workgroup (head, n) owns the plan[2 + n + 1] - plan[2 + n] rows of sentence n; the extra workgroups
// n == N keep the rows P .. plan[1] - 1 of `out` zero (the next product reads them).  LDS sized for Lmax rows.
__global__ __launch_bounds__(256) void mha_packed_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                             const int* __restrict__ plan, int N, int Lmax, int W, int causal,
                                                             float scale, unsigned* __restrict__ amax) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int h = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  if (n == N) {
    const int P = plan[0], P2 = plan[1];
    for (int idx = tid; idx < (P2 - P) * (HD / 4); idx += 256) {
      const int i = P + (idx >> 4), d = (idx & 15) * 4;
      *reinterpret_cast<float4*>(out + (long)i * W + h * HD + d) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (amax != nullptr) amax_commit_block(0u, amax);
    return;
  }
  const int r0 = plan[2 + n], L = plan[2 + n + 1] - r0;
  float* q = sm;
  float* k = q + Lmax * HP;
  float* v = k + Lmax * HP;
  float* s = v + Lmax * HP;  // [L][L+1]
  const int LP = L + 1;
  const float* base = qkv + (long)r0 * 3 * W + h * HD;
  for (int idx = tid; idx < L * (HD / 4); idx += 256) {
    const int l = idx >> 4, d = (idx & 15) * 4;
    const float* r = base + (long)l * 3 * W + d;
    *reinterpret_cast<float4*>(&q[l * HP + d]) = *reinterpret_cast<const float4*>(r);
    *reinterpret_cast<float4*>(&k[l * HP + d]) = *reinterpret_cast<const float4*>(r + W);
    *reinterpret_cast<float4*>(&v[l * HP + d]) = *reinterpret_cast<const float4*>(r + 2 * W);
  }
  __syncthreads();
  for (int idx = tid; idx < L * L; idx += 256) {
    const int i = idx / L, j = idx - i * L;
    float acc = -INFINITY;
    if (!causal || j <= i) {
      acc = 0.f;
#pragma unroll
      for (int d = 0; d < HD; d += 4) acc += dot4(lds4(&q[i * HP + d]), lds4(&k[j * HP + d]));
      acc *= scale;
    }
    s[i * LP + j] = acc;
  }
  __syncthreads();
  const int lane = tid & 63, wv = tid >> 6;
  for (int i = wv; i < L; i += 4) {
    float x = lane < L ? s[i * LP + lane] : -INFINITY;
    float m = wave_max(x);
    float e = lane < L ? expf(x - m) : 0.f;
    float sum = wave_sum(e);
    if (lane < L) s[i * LP + lane] = e / sum;
  }
  __syncthreads();
  float* ob = out + (long)r0 * W + h * HD;
  unsigned am = 0u;
  for (int idx = tid; idx < L * (HD / 4); idx += 256) {
    const int i = idx >> 4, d = (idx & 15) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < L; ++j) fma4(acc, s[i * LP + j], lds4(&v[j * HP + d]));
    *reinterpret_cast<float4*>(&ob[(long)i * W + d]) = acc;
    am = max(am, abits4(acc));
  }
  if (amax != nullptr) amax_commit_block(am, amax);
}

// out[n] = x[last packed row of sentence n]
__global__ void eot_gather_packed_kernel(const float* __restrict__ x, const int* __restrict__ plan, float* __restrict__ out, int W) {
  const int n = blockIdx.x;
  const long r = plan[2 + n + 1] - 1;
  for (int c = threadIdx.x; c < W; c += blockDim.x) out[(long)n * W + c] = x[r * W + c];
}

// both kernels may need more than the default 64 KB of dynamic LDS (L = 50..64)
int mha_init() {
  static int state = -1;
  if (state < 0) {
    hipError_t a = hipFuncSetAttribute((const void*)mha_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipError_t b = hipFuncSetAttribute((const void*)mha_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    state = (a != hipSuccess) ? (int)a : (int)b;
  }
  return state;
}

}  // namespace

extern "C" int tris_mha_fwd_f32(const float* qkv, float* out, int N, int L, int W, int heads, int causal, void* stream) {
  if (W != heads * HD || L > 64 || L < 1) return (int)hipErrorInvalidValue;
  size_t lds = (size_t)(3 * L * HP + L * (L + 1)) * sizeof(float);
  if (int e = mha_init()) return e;
  hipLaunchKernelGGL(mha_fwd_kernel, dim3(heads, N), dim3(256), lds, (hipStream_t)stream, qkv, out, L, W, causal,
                     0.125f, tris_internal_take_amax_next());
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_mha_bwd_f32(const float* qkv, const float* dout, float* dqkv, int N, int L, int W, int heads,
                                int causal, void* stream) {
  if (W != heads * HD || L > 64 || L < 1) return (int)hipErrorInvalidValue;
  size_t lds = (size_t)(4 * L * HP + 2 * L * (L + 1)) * sizeof(float);
  if (int e = mha_init()) return e;
  hipLaunchKernelGGL(mha_bwd_kernel, dim3(heads, N), dim3(256), lds, (hipStream_t)stream, qkv, dout, dqkv, L, W, causal,
                     0.125f, tris_internal_take_amax_next());
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_text_pack_plan_i64(const long* ids, int N, int L, int* plan, void* stream) {
  if (N < 1 || L < 1 || N > 8192) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(text_pack_plan_kernel, dim3(1), dim3(256), (size_t)N * sizeof(int), (hipStream_t)stream, ids, N, L, plan);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_embed_packed_fwd_f32(const long* ids, const float* tok, const float* pos, const int* plan, float* out, int N, int L,
                                         int W, void* stream) {
  if (W % 4) return (int)hipErrorInvalidValue;
  const long R = (((long)N * L + 255) / 256) * 256;      // rows of `out`
  const long n = R * (W / 4);
  hipLaunchKernelGGL(embed_packed_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, ids, tok, pos, plan, out,
                     R, N, L, W);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_mha_packed_fwd_f32(const float* qkv, float* out, const int* plan, int N, int Lmax, int W, int heads, int causal,
                                       void* stream) {
  if (W != heads * HD || Lmax > 64 || Lmax < 1) return (int)hipErrorInvalidValue;
  static int st = -1;
  if (st < 0) st = (int)hipFuncSetAttribute((const void*)mha_packed_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  if (st) return st;
  const size_t lds = (size_t)(3 * Lmax * HP + Lmax * (Lmax + 1)) * sizeof(float);
  hipLaunchKernelGGL(mha_packed_fwd_kernel, dim3(heads, N + 1), dim3(256), lds, (hipStream_t)stream, qkv, out, plan, N, Lmax, W, causal,
                     0.125f, tris_internal_take_amax_next());
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_eot_gather_packed_f32(const float* x, const int* plan, float* out, int N, int W, void* stream) {
  hipLaunchKernelGGL(eot_gather_packed_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, x, plan, out, W);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_embed_fwd_f32(const long* ids, const float* tok, const float* pos, float* out, int N, int L, int W,
                                  void* stream) {
  if (W % 4) return (int)hipErrorInvalidValue;
  long n = (long)N * L * (W / 4);
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, ids, tok, pos, out,
                     (long)N * L, L, W);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_embed_bwd_f32(const long* ids, const float* dout, float* dtok, float* dpos, int N, int L, int W,
                                  void* stream) {
  long n = (long)N * L * W;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, ids, dout, dtok, dpos, N,
                     L, W);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_embed_rows_bwd_f32(const long* ids, const float* rows, float* dtok, int R, int W, float scale,
                                       void* stream) {
  if (R < 1 || W % 4 != 0 || W > 2048) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(embed_rows_bwd_kernel, dim3(R), dim3(128), 0, (hipStream_t)stream, ids, rows, dtok, R, W, scale);
  TRIS_LAUNCH_CHECK();
  return 0;
}

// out = [a; b]  (int64 token rows: the positive and the negative queries of a step in one list, train_stage1.py:342-347: batched here)
namespace {
__global__ void concat_i64_kernel(const long* __restrict__ a, long na, const long* __restrict__ b, long nb, long* __restrict__ out) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += stride) out[i] = i < na ? a[i] : b[i - na];
}
}  // namespace
extern "C" int tris_concat_i64(const long* a, long na, const long* b, long nb, long* out, void* stream) {
  const long n = na + nb;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(concat_i64_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 1024)), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, out);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_eot_gather_fwd_f32(const long* ids, const float* x, float* out, int N, int L, int W, void* stream) {
  hipLaunchKernelGGL(eot_gather_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, ids, x, out, (float*)nullptr,
                     (const float*)nullptr, L, W);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_eot_gather_bwd_f32(const long* ids, const float* dout, float* dx, int N, int L, int W,
                                       void* stream) {
  hipLaunchKernelGGL(eot_gather_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, ids, (const float*)nullptr,
                     (float*)nullptr, dx, dout, L, W);
  TRIS_LAUNCH_CHECK();
  return 0;
}
