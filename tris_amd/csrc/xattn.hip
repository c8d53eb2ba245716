// Fused image<->text cross attention of `bilateral_prompt` (reference model/attn.py:117-128) for ALL images of the batch
// in three launches (gfx950).  This is the kernel pair the north star prices against the HBM roofline:
// algorithmic traffic per image = Qv,Kv,Vv reads (3 x P x C x 4 B) + new_vis, new_lan writes = 1.84 MB at P=100, N=48, C=1024.
//
//   pixels   Qv,Kv,Vv [B,P,C]   (v_proj1..3 outputs, channels-last)
//   sentences Qt,Kt,Vt [N,C]    (t_proj1..3 outputs; one sentence set shared by every image, model_stage1.py:66)
//   Av  = softmax_n(Qv Kt^T / sqrt(C))   [B,P,N]      new_vis = Av  Vt      [B,P,C]
//   AtT = softmax_p(Kv Qt^T / sqrt(C))   [B,P,N]      new_lan = AtT^T Vv    [B,N,C]
//
// Launch 1 (scores): one workgroup per (16-pixel tile, {Qv|Kv}, image).  Operand rows stream straight from HBM/L2 into
//   MFMA registers (no LDS staging): a lane loads 16 bytes of its row and the 4 floats feed 4 consecutive
//   v_mfma_f32_16x16x4_f32 (logical k order permuted identically for both operands).  The 4 waves split C, partial 16xN
//   tiles are reduced through LDS; the Qv half finishes its row softmax in place (wave shuffles), the Kv half stores
//   scaled logits (its softmax runs over pixels, i.e. across workgroups).
// Launch 2 (outputs): one workgroup per (32-channel tile, image): probabilities (2 x P x N), a Vt tile and a Vv tile
//   are staged in LDS (Vv/Vt with coalesced 16-byte loads), the pixel softmax is finished, then both products run on
//   the f32 MFMA.  ~65 KB LDS -> 2 workgroups per CU, 32*B workgroups >> 256 CUs.
#include "common.h"
#include "tris_hip.h"

namespace {

constexpr int MAXNF = 4;  // N <= 64 sentences (16 per MFMA fragment)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// grid (tiles of 16 pixels, 2, B); block 256
template <int NF>
__global__ __launch_bounds__(256) void xattn_scores_kernel(const float* __restrict__ Qv, const float* __restrict__ Kv,
                                                           const float* __restrict__ Qt, const float* __restrict__ Kt,
                                                           float* __restrict__ probs, int P, int N, int C, float scale) {
  __shared__ float part[4][16][NF * 16 + 1];
  const int tile = blockIdx.x, which = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const float* __restrict__ X = (which == 0 ? Qv : Kv) + (long)b * P * C;  // pixel rows
  const float* __restrict__ T = (which == 0 ? Kt : Qt);                    // sentence rows
  const int row = min(tile * 16 + li, P - 1);
  const float* xr = X + (long)row * C + kq * 4;
  const float* tr[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) tr[f] = T + (long)min(f * 16 + li, N - 1) * C + kq * 4;
  f32x4 acc[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int kspan = C / 4;  // channels per wave
  const int kbeg = wave * kspan;
  // 64 channels per trip: all 4*(1+NF) 16-byte loads are issued before the first MFMA consumes one (the loop is
  // latency-bound otherwise: every MFMA group would wait a full HBM/L2 round trip)
  int k = kbeg;
  for (; k + 64 <= kbeg + kspan; k += 64) {
    float4 a[4], t[NF][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = ld4(xr + k + 16 * u);
#pragma unroll
      for (int f = 0; f < NF; ++f) t[f][u] = ld4(tr[f] + k + 16 * u);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, t[f][u].x, acc[f], 0, 0, 0);
        acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, t[f][u].y, acc[f], 0, 0, 0);
        acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, t[f][u].z, acc[f], 0, 0, 0);
        acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, t[f][u].w, acc[f], 0, 0, 0);
      }
  }
  for (; k < kbeg + kspan; k += 16) {  // remainder when C/4 is not a multiple of 64
    const float4 a = ld4(xr + k);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const float4 t = ld4(tr[f] + k);
      acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, t.x, acc[f], 0, 0, 0);
      acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, t.y, acc[f], 0, 0, 0);
      acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, t.z, acc[f], 0, 0, 0);
      acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, t.w, acc[f], 0, 0, 0);
    }
  }
  // C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][kq * 4 + r][f * 16 + li] = acc[f][r];
  __syncthreads();
  // 4 waves x 4 rows each: sum the partials, then (Qv half) softmax over the N sentences of a row
  float* out = probs + (((long)b * 3 + which) * P) * N;  // plane 0: Av, plane 1: Kv.Qt^T logits, plane 2: AtT
  for (int r = wave * 4; r < wave * 4 + 4; ++r) {
    const int p = tile * 16 + r;
    float v = -INFINITY;
    if (lane < N) v = (part[0][r][lane] + part[1][r][lane] + part[2][r][lane] + part[3][r][lane]) * scale;
    if (which == 0) {
      const float m = wave_max(v);
      const float e = lane < N ? expf(v - m) : 0.f;
      const float s = wave_sum(e);
      v = e / s;
    }
    if (p < P && lane < N) out[(long)p * N + lane] = v;
  }
}

// Pixel softmax of the Kv.Qt^T logits (plane 1 -> AtT probabilities in plane 2); one workgroup per image, so the
// exponentials are evaluated once per image instead of once per channel tile of launch 3.
__global__ __launch_bounds__(256) void xattn_colsoftmax_kernel(float* __restrict__ probs, int P, int N) {
  __shared__ float cm[128];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* lg = probs + ((long)b * 3 + 1) * P * N;
  float* out = probs + ((long)b * 3 + 2) * P * N;
  const int n = tid >> 2, q = tid & 3;
  float m = -INFINITY;
  if (n < N) for (int p = q; p < P; p += 4) m = fmaxf(m, lg[(long)p * N + n]);
  m = fmaxf(m, __shfl_xor(m, 1, 64));
  m = fmaxf(m, __shfl_xor(m, 2, 64));
  float s = 0.f;
  if (n < N) for (int p = q; p < P; p += 4) s += expf(lg[(long)p * N + n] - m);
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  if (q == 0) { cm[n] = m; cm[64 + n] = s; }
  __syncthreads();
  for (int i = tid; i < P * N; i += 256) {
    const int nn = i % N;
    out[i] = expf(lg[i] - cm[nn]) / cm[64 + nn];
  }
}

// grid (C/32, B); block 256; dynamic LDS
template <int NF>
__global__ __launch_bounds__(256) void xattn_out_kernel(const float* __restrict__ Vv, const float* __restrict__ Vt,
                                                        float* __restrict__ probs, float* __restrict__ new_vis,
                                                        float* __restrict__ new_lan, int P, int N, int C) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int NP = NF * 16 + 4;  // LDS row stride of the probability tiles
  constexpr int CT = 32, CP = CT + 4;
  const int MT = (P + 15) / 16, PR = MT * 16;
  float* pA = sm;                 // Av  [PR][NP]   (rows >= P zero)
  float* pT = pA + PR * NP;       // AtT [PR][NP]
  float* vt = pT + PR * NP;       // Vt tile [NF*16][CP] (rows >= N zero)
  float* vv = vt + NF * 16 * CP;  // Vv tile [PR][CP]     (rows >= P zero)
  const int c0 = blockIdx.x * CT, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* pb = probs + (long)b * 3 * P * N;
  // Fills are written as "issue U independent loads, then store" so the global latency is paid once per batch.
  {
    const int total = P * N;  // compact [P][N] planes in HBM -> padded [PR][NP] LDS rows; pads pre-set below
    for (int i = tid; i < PR * NP; i += 256) { pA[i] = 0.f; pT[i] = 0.f; }
    __syncthreads();
    constexpr int U = 8;
    for (int i0 = tid; i0 < total; i0 += 256 * U) {
      float va[U], vl[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = min(i0 + u * 256, total - 1);
        va[u] = pb[i];
        vl[u] = pb[(long)2 * total + i];  // plane 2: AtT probabilities (xattn_colsoftmax_kernel)
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * 256;
        if (i < total) {
          const int p = i / N, n = i - p * N;
          pA[p * NP + n] = va[u];
          pT[p * NP + n] = vl[u];
        }
      }
    }
  }
  {
    constexpr int F4 = CT / 4;
    for (int i = tid; i < NF * 16 * F4; i += 256) {
      const int n = i / F4, c4 = (i - n * F4) * 4;
      float4 v = ld4(Vt + (long)min(n, N - 1) * C + c0 + c4);
      if (n >= N) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(&vt[n * CP + c4]) = v;
    }
    constexpr int U = 4;
    for (int i0 = tid; i0 < PR * F4; i0 += 256 * U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = min(i0 + u * 256, PR * F4 - 1);
        const int p = i / F4, c4 = (i - p * F4) * 4;
        v[u] = ld4(Vv + ((long)b * P + min(p, P - 1)) * C + c0 + c4);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * 256;
        if (i < PR * F4) {
          const int p = i / F4, c4 = (i - p * F4) * 4;
          *reinterpret_cast<float4*>(&vv[p * CP + c4]) = p < P ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
  }
  __syncthreads();
  const int li = lane & 15, kq = lane >> 4;
  const int nf = wave & 1, mh = wave >> 1;  // wave -> (16-channel fragment, half of the row fragments)
  const int ccol = nf * 16 + li;
  // ---- new_vis tile [P x 32] = Av [P x N] . Vt tile [N x 32] ------------------------------------------------------------
  for (int mf = mh; mf < MT; mf += 2) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < NF * 16; k += 4)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pA[(mf * 16 + li) * NP + k + kq], vt[(k + kq) * CP + ccol], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = mf * 16 + kq * 4 + r;
      if (p < P) new_vis[((long)b * P + p) * C + c0 + ccol] = acc[r];
    }
  }
  // ---- new_lan tile [N x 32] = AtT^T [N x P] . Vv tile [P x 32] ---------------------------------------------------------
  for (int mf = mh; mf < NF; mf += 2) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < PR; k += 4)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pT[(k + kq) * NP + mf * 16 + li], vv[(k + kq) * CP + ccol], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = mf * 16 + kq * 4 + r;
      if (n < N) new_lan[((long)b * N + n) * C + c0 + ccol] = acc[r];
    }
  }
}

// dX[b][p][n] = scale * Y * (dY - sum_p' Y dY)   (softmax over the P axis of [B,P,N]); one workgroup per image
__global__ __launch_bounds__(256) void softmax_col_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                              float* __restrict__ dX, int P, int N, float scale) {
  __shared__ float dot[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const long base = (long)b * P * N;
  const int n = tid >> 2, q = tid & 3;
  float s = 0.f;
  if (n < N) for (int p = q; p < P; p += 4) s += Y[base + (long)p * N + n] * dY[base + (long)p * N + n];
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  if (n < 64 && q == 0) dot[n] = s;
  __syncthreads();
  for (int i = tid; i < P * N; i += 256) {
    const int nn = i % N;
    dX[base + i] = scale * Y[base + i] * (dY[base + i] - dot[nn]);
  }
}

size_t out_lds_bytes(int P, int NF) {
  const int PR = (P + 15) / 16 * 16, NP = NF * 16 + 4, CP = 36;
  return (size_t)(2 * PR * NP + NF * 16 * CP + PR * CP + 128) * sizeof(float);
}

template <int NF>
int launch_fwd(const float* Qv, const float* Kv, const float* Vv, const float* Qt, const float* Kt, const float* Vt,
               float* new_vis, float* new_lan, float* probs, int B, int P, int N, int C, hipStream_t st) {
  const float scale = 1.0f / sqrtf((float)C);
  hipLaunchKernelGGL((xattn_scores_kernel<NF>), dim3((P + 15) / 16, 2, B), dim3(256), 0, st, Qv, Kv, Qt, Kt, probs, P, N,
                     C, scale);
  TRIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(xattn_colsoftmax_kernel, dim3(B), dim3(256), 0, st, probs, P, N);
  TRIS_LAUNCH_CHECK();
  const size_t lds = out_lds_bytes(P, NF);
  static bool attr_done[MAXNF + 1] = {false, false, false, false, false};
  if (!attr_done[NF]) {
    hipError_t e = hipFuncSetAttribute((const void*)xattn_out_kernel<NF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       150 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done[NF] = true;
  }
  hipLaunchKernelGGL((xattn_out_kernel<NF>), dim3(C / 32, B), dim3(256), lds, st, Vv, Vt, probs, new_vis, new_lan, P, N,
                     C);
  TRIS_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int tris_xattn_fwd_f32(const float* Qv, const float* Kv, const float* Vv, const float* Qt, const float* Kt,
                                  const float* Vt, float* new_vis, float* new_lan, float* probs, int B, int P, int N,
                                  int C, void* stream) {
  if (N < 1 || N > 16 * MAXNF || P < 1 || C % 64 != 0 || out_lds_bytes(P, (N + 15) / 16) > 150 * 1024)
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  switch ((N + 15) / 16) {
    case 1: return launch_fwd<1>(Qv, Kv, Vv, Qt, Kt, Vt, new_vis, new_lan, probs, B, P, N, C, st);
    case 2: return launch_fwd<2>(Qv, Kv, Vv, Qt, Kt, Vt, new_vis, new_lan, probs, B, P, N, C, st);
    case 3: return launch_fwd<3>(Qv, Kv, Vv, Qt, Kt, Vt, new_vis, new_lan, probs, B, P, N, C, st);
    default: return launch_fwd<4>(Qv, Kv, Vv, Qt, Kt, Vt, new_vis, new_lan, probs, B, P, N, C, st);
  }
}

extern "C" int tris_softmax_col_bwd_f32(const float* dY, const float* Y, float* dX, int B, int P, int N, float scale,
                                        void* stream) {
  if (N > 64) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(softmax_col_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dY, Y, dX, P, N, scale);
  TRIS_LAUNCH_CHECK();
  return 0;
}
