// Fused image<->text cross attention of `bilateral_prompt` (reference model/attn.py:117-128) for ALL images of the batch
// in two (split-bf16 arithmetic, default) or three (f32 arithmetic) launches (gfx950).  This is the kernel pair the north
// star prices against the HBM roofline:
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
#include "x3_split.h"

namespace {

constexpr int MAXNF = 4;  // N <= 64 sentences (16 per MFMA fragment)
constexpr int NPL = 4;    // planes of the probability scratch: 0 Av, 1 Kv.Qt^T logits, 2 AtT, 3 Qv.Kt^T logits (x3 path)

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
  float* out = probs + (((long)b * NPL + which) * P) * N;  // plane 0: Av, plane 1: Kv.Qt^T logits, plane 2: AtT
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
  const float* lg = probs + ((long)b * NPL + 1) * P * N;
  float* out = probs + ((long)b * NPL + 2) * P * N;
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
  float* pb = probs + (long)b * NPL * P * N;
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

// ---- split-bf16 ("x3") path: two launches, HBM-streaming --------------------------------------------------------------------
// The f32-input MFMA makes this pair MFMA-co-bound (21 FLOP/B sits on the f32 ridge).  With every fp32 operand split into
// three exact bf16 pieces (x3_split.h) the six significant piece products run on v_mfma_f32_16x16x32_bf16 at 2.5x the rate
// and the kernels become HBM streams:
//   scores (xattn_scores_x3_kernel): one wave = 16 pixel rows x a quarter of C.  The rows go HBM -> registers (32 B per
//     lane per step, a full 128-B line per row), the sentence tile comes from L2 the same way, both are split in
//     registers; no LDS, no barrier in the loop; the 4 waves' partial 16 x N tiles meet in LDS once.  Pixel rows are
//     taken from the flat [B*P, C] matrix (the sentence set is shared), so there is no per-image padding.
//   outputs (xattn_out_x3_kernel): one workgroup per (64-channel tile, image).  Both logit planes of the image -> LDS,
//     row softmax (over sentences) and column softmax (over pixels) in place, then new_vis = Av.Vt and new_lan = At.Vv
//     on the MFMA with the probabilities read from LDS and the Vt / Vv fragments loaded straight from global in MFMA
//     operand order (issued before the softmax so their latency hides behind it).
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4v mfma6(const Split8& a, const Split8& b, f32x4v c) {  // smallest terms first
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.mid, b.mid, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.mid, b.hi, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.mid, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, c, 0, 0, 0);
  return c;
}

// Wave-local LDS transposition: global loads are issued row-contiguous (8 rows x 128 B per instruction, so every 16-lane
// group reads whole lines -- fragment-shaped loads cost 4-8x the address/tag work for the same bytes), parked in a
// per-wave LDS tile [rows][32 + 4 floats] and read back as MFMA fragments (8 consecutive k per lane, two conflict-free
// ds_read_b128).  Only the owning wave touches its tile: program order + a wave-scope fence, no workgroup barrier.
constexpr int TLD = 36;
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// grid (ceil(B*P / 32), 2); block 256.  y = 0: Qv.Kt^T -> plane 3, y = 1: Kv.Qt^T -> plane 1 (both scaled logits).
// Wave w reduces over channels [w*C/4, (w+1)*C/4): its whole share of the pixel rows (16 rows x C/4 floats <= 16 KB) is
// requested from HBM up front, the sentence tile (L2) is fetched one 32-channel step ahead.  Requires C <= 1024, C % 128 == 0.
template <int NT, int KS>
__global__ __launch_bounds__(256) void xattn_scores_x3_kernel(const float* __restrict__ Qv, const float* __restrict__ Kv,
                                                              const float* __restrict__ Qt, const float* __restrict__ Kt,
                                                              float* __restrict__ probs, long R, int P, int N, int C,
                                                              float scale) {
  // 32 pixel rows per workgroup (two MFMA row tiles share every sentence fragment); wave w reduces over channels
  // [w*C/4, (w+1)*C/4).  Pixel rows are requested 4 steps (4 x 128 B per row) ahead, the sentence tile (L2) one step.
  __shared__ __attribute__((aligned(16))) float At[4][32 * TLD];
  __shared__ __attribute__((aligned(16))) float Bt[4][NT * 16 * TLD];
  f32x4v(*red)[2 * NT][64] = reinterpret_cast<f32x4v(*)[2 * NT][64]>(&Bt[0][0]);  // [3][2 NT][64], reuses the tiles after the loop
  static_assert(sizeof(float) * 4 * NT * 16 * TLD >= sizeof(f32x4v) * 3 * 2 * NT * 64, "reduction scratch must fit");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, kg = lane >> 4;       // fragment coordinates
  const int lr = lane >> 3, lc = (lane & 7) * 4;  // loader coordinates: row within an 8-row pass, float offset in the 128-B segment
  const int m = blockIdx.y;
  const float* __restrict__ A = m == 0 ? Qv : Kv;
  const float* __restrict__ T = m == 0 ? Kt : Qt;
  const long row0 = (long)blockIdx.x * 32;
  const int kq = C >> 2;  // = 32 * KS (compile-time step count: no control flow around the prefetches, so the waits are exact)
  const float* ap[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) ap[q] = A + min(row0 + q * 8 + lr, R - 1) * C + wave * kq + lc;
  // Issue order matters: the vector-memory counter retires in order, so what is needed soonest is requested first (step 0's
  // rows and sentence tile), the deep row prefetch last -- a wait for the sentence tile then leaves the prefetch in flight.
  float4 a[4][4];  // [step & 3][8-row pass]
#pragma unroll
  for (int q = 0; q < 4; ++q) a[0][q] = ld4(ap[q]);
  const float* tp[2 * NT];
#pragma unroll
  for (int q = 0; q < 2 * NT; ++q) tp[q] = T + (long)min(q * 8 + lr, N - 1) * C + wave * kq + lc;
  float4 bq[2 * NT];
#pragma unroll
  for (int q = 0; q < 2 * NT; ++q) bq[q] = ld4(tp[q]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 1; s < 4; ++s)
    if (s < KS) {
#pragma unroll
      for (int q = 0; q < 4; ++q) a[s][q] = ld4(ap[q] + s * 32);
    }
  f32x4v acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4v){0.f, 0.f, 0.f, 0.f};
  float* at = At[wave];
  float* bt = Bt[wave];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (s < KS) {
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(at + (q * 8 + lr) * TLD + lc) = a[s & 3][q];
#pragma unroll
      for (int q = 0; q < 2 * NT; ++q) *reinterpret_cast<float4*>(bt + (q * 8 + lr) * TLD + lc) = bq[q];
      if (s + 1 < KS) {
#pragma unroll
        for (int q = 0; q < 2 * NT; ++q) bq[q] = ld4(tp[q] + (s + 1) * 32);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the request order (see above)
      if (s + 4 < KS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[s & 3][q] = ld4(ap[q] + (s + 4) * 32);
      }
      __builtin_amdgcn_sched_barrier(0);
      wave_lds_fence();
      Split8 sa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        sa[i] = split8(*reinterpret_cast<const float4*>(at + (i * 16 + r) * TLD + kg * 8),
                       *reinterpret_cast<const float4*>(at + (i * 16 + r) * TLD + kg * 8 + 4));
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float* bs = bt + (j * 16 + r) * TLD + kg * 8;
        const Split8 sb = split8(*reinterpret_cast<const float4*>(bs), *reinterpret_cast<const float4*>(bs + 4));
        acc[0][j] = mfma6(sa[0], sb, acc[0][j]);
        acc[1][j] = mfma6(sa[1], sb, acc[1][j]);
      }
      wave_lds_fence();
    }
  }
  __syncthreads();  // every wave is done with its tiles
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) red[wave - 1][i * NT + j][lane] = acc[i][j];
  }
  __syncthreads();
  if (wave == 0) {
    const int plane = m == 0 ? 3 : 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const f32x4v v = acc[i][j] + red[0][i * NT + j][lane] + red[1][i * NT + j][lane] + red[2][i * NT + j][lane];
        const int n = j * 16 + r;
#pragma unroll
        for (int t = 0; t < 4; ++t) {  // D[4*kg + t][r]
          const long row = row0 + i * 16 + 4 * kg + t;
          if (row < R && n < N) {
            const long b = row / P;
            const int p = (int)(row - b * P);
            probs[((b * NPL + plane) * P + p) * N + n] = v[t] * scale;
          }
        }
      }
  }
}

constexpr int XLD = 68;    // row stride (floats) of the Av plane [p][n]: 16-byte aligned, conflict-free b128 fragments
constexpr int XLT = 132;   // row stride (floats) of the At plane [n][p]

// grid (C / 128, B); block 256 (wave w owns channels c0 + 32w .. +31: two MFMA column tiles share every probability
// fragment); dynamic LDS (round16(P) * XLD + 64 * XLT) floats
template <int ABL = 0>   // ABL: developer ablation bits (tools/probes/xattn_out_probe.hip); 0 in the product
__global__ __launch_bounds__(256) void xattn_out_x3_kernel(const float* __restrict__ Vv, const float* __restrict__ Vt,
                                                           float* __restrict__ probs, float* __restrict__ new_vis,
                                                           float* __restrict__ new_lan, int P, int N, int C) {
  extern __shared__ __attribute__((aligned(16))) float xlds[];
  const int PR = (P + 15) / 16 * 16;
  float* Sa = xlds;             // S1 logits -> Av  [p][n]   (softmax over n)
  float* St = xlds + PR * XLD;  // S2 logits -> At  [n][p]   (softmax over p)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kg = lane >> 4;
  const int b = blockIdx.y;
  const int c = blockIdx.x * 128 + wave * 32 + r;  // first of this lane's two channels (the other is c + 16)
  const int PT = PR / 16, NT = (N + 15) / 16, NK = (N + 31) / 32, PK = (P + 31) / 32;

  // operand fragments straight from global, in MFMA order (8 consecutive k per lane): issued first, consumed last
  float vv[2][4][8], vt[2][2][8];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = ks * 32 + kg * 8 + i;
        vv[ct][ks][i] = (ABL & 1) ? 0.5f : ((ks < PK && p < P) ? Vv[((long)b * P + p) * C + c + ct * 16] : 0.f);
      }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int n = ks * 32 + kg * 8 + i;
        vt[ct][ks][i] = (ks < NK && n < N) ? Vt[(long)n * C + c + ct * 16] : 0.f;
      }

  // both logit planes of the image -> LDS (all requests in flight before the first LDS write)
  const float* s1 = probs + ((long)b * NPL + 3) * P * N;
  const float* s2 = probs + ((long)b * NPL + 1) * P * N;
  const int tot = P * N;
  if ((N & 3) == 0 && tot <= 8 * 1024) {
    float4 u[8], w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = (q * 256 + tid) * 4;
      if (i < tot) { u[q] = ld4(s1 + i); w[q] = ld4(s2 + i); }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = (q * 256 + tid) * 4;
      if (i < tot) {
        const int p = i / N, n = i - p * N;
        *reinterpret_cast<float4*>(Sa + p * XLD + n) = u[q];
        St[(n + 0) * XLT + p] = w[q].x;
        St[(n + 1) * XLT + p] = w[q].y;
        St[(n + 2) * XLT + p] = w[q].z;
        St[(n + 3) * XLT + p] = w[q].w;
      }
    }
  } else {
    for (int i = tid; i < tot; i += 256) {
      const int p = i / N, n = i - p * N;
      Sa[p * XLD + n] = s1[i];
      St[n * XLT + p] = s2[i];
    }
  }
  __syncthreads();
  if ((ABL & 2) == 0 && tid < PR) {  // Av: softmax over the sentences of pixel `tid`; k padding (n >= N) and pad rows become exact zeros
    float* row = Sa + tid * XLD;
    float4 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = *reinterpret_cast<const float4*>(row + q * 4);
    float mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float* e = reinterpret_cast<float*>(&v[q]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (q * 4 + t >= N || tid >= P) e[t] = -INFINITY;
        mx = fmaxf(mx, e[t]);
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float* e = reinterpret_cast<float*>(&v[q]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        e[t] = (q * 4 + t < N && tid < P) ? __expf(e[t] - mx) : 0.f;
        sum += e[t];
      }
    }
    const float inv = tid < P ? 1.f / sum : 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      v[q].x *= inv; v[q].y *= inv; v[q].z *= inv; v[q].w *= inv;
      *reinterpret_cast<float4*>(row + q * 4) = v[q];
    }
  }
  if ((ABL & 2) == 0) {  // At: softmax over the pixels of sentence n -- 4 threads per sentence, 32 pixels each (p = q4 + 4 i)
    const int n = tid >> 2, q4 = tid & 3;
    float* row = St + n * XLT;
    float e[32];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int p = q4 + 4 * i;
      e[i] = (n < N && p < P) ? row[p] : -INFINITY;
      mx = fmaxf(mx, e[i]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int p = q4 + 4 * i;
      e[i] = (n < N && p < P) ? __expf(e[i] - mx) : 0.f;
      sum += e[i];
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    const float inv = n < N ? 1.f / sum : 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) row[q4 + 4 * i] = e[i] * inv;  // pixels >= P and sentences >= N become exact zeros
  }
  __syncthreads();
  if (blockIdx.x == 0) {  // probabilities for backward: plane 0 = Av [p][n], plane 2 = AtT [p][n]
    float* o0 = probs + ((long)b * NPL + 0) * P * N;
    float* o2 = probs + ((long)b * NPL + 2) * P * N;
    for (int i = tid; i < tot; i += 256) {
      const int p = i / N, n = i - p * N;
      o0[i] = Sa[p * XLD + n];
      o2[i] = St[n * XLT + p];
    }
  }

  // new_vis[b][p][c] = sum_n Av[p][n] Vt[n][c]
  {
    f32x4v acc[2][8];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int rt = 0; rt < 8; ++rt) acc[ct][rt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks < NK) {
        Split8 sb[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
          sb[ct] = split8(make_float4(vt[ct][ks][0], vt[ct][ks][1], vt[ct][ks][2], vt[ct][ks][3]),
                          make_float4(vt[ct][ks][4], vt[ct][ks][5], vt[ct][ks][6], vt[ct][ks][7]));
#pragma unroll
        for (int rt = 0; rt < 8; ++rt) {
          if (rt < PT) {
            const float* s = Sa + (rt * 16 + r) * XLD + ks * 32 + kg * 8;
            Split8 sa;
            if (ABL & 8) { const float4 u = ld4(s), w = ld4(s + 4); sa.hi = __builtin_bit_cast(bf16x8, make_uint4(__builtin_bit_cast(unsigned, u.x), __builtin_bit_cast(unsigned, u.y), __builtin_bit_cast(unsigned, u.z), __builtin_bit_cast(unsigned, u.w))); sa.mid = __builtin_bit_cast(bf16x8, make_uint4(__builtin_bit_cast(unsigned, w.x), __builtin_bit_cast(unsigned, w.y), __builtin_bit_cast(unsigned, w.z), __builtin_bit_cast(unsigned, w.w))); sa.lo = sa.hi; }
            else sa = split8(ld4(s), ld4(s + 4));
            acc[0][rt] = mfma6(sa, sb[0], acc[0][rt]);
            acc[1][rt] = mfma6(sa, sb[1], acc[1][rt]);
          }
        }
      }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int rt = 0; rt < 8; ++rt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int p = rt * 16 + 4 * kg + t;
          if ((ABL & 4) == 0 ? (rt < PT && p < P) : (acc[ct][rt][t] == 12345.f)) new_vis[((long)b * P + p) * C + c + ct * 16] = acc[ct][rt][t];
        }
  }
  // new_lan[b][n][c] = sum_p At[n][p] Vv[b][p][c]
  {
    f32x4v acc[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[ct][nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < PK) {
        Split8 sb[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
          sb[ct] = split8(make_float4(vv[ct][ks][0], vv[ct][ks][1], vv[ct][ks][2], vv[ct][ks][3]),
                          make_float4(vv[ct][ks][4], vv[ct][ks][5], vv[ct][ks][6], vv[ct][ks][7]));
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          if (nt < NT) {
            const float* s = St + (nt * 16 + r) * XLT + ks * 32 + kg * 8;
            Split8 sa;
            if (ABL & 8) { const float4 u = ld4(s), w = ld4(s + 4); sa.hi = __builtin_bit_cast(bf16x8, make_uint4(__builtin_bit_cast(unsigned, u.x), __builtin_bit_cast(unsigned, u.y), __builtin_bit_cast(unsigned, u.z), __builtin_bit_cast(unsigned, u.w))); sa.mid = __builtin_bit_cast(bf16x8, make_uint4(__builtin_bit_cast(unsigned, w.x), __builtin_bit_cast(unsigned, w.y), __builtin_bit_cast(unsigned, w.z), __builtin_bit_cast(unsigned, w.w))); sa.lo = sa.hi; }
            else sa = split8(ld4(s), ld4(s + 4));
            acc[0][nt] = mfma6(sa, sb[0], acc[0][nt]);
            acc[1][nt] = mfma6(sa, sb[1], acc[1][nt]);
          }
        }
      }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int n = nt * 16 + 4 * kg + t;
          if ((ABL & 4) == 0 ? (nt < NT && n < N) : (acc[ct][nt][t] == 12345.f)) new_lan[((long)b * N + n) * C + c + ct * 16] = acc[ct][nt][t];
        }
  }
}

template <int NT, int KS>
int launch_fwd_x3(const float* Qv, const float* Kv, const float* Vv, const float* Qt, const float* Kt, const float* Vt,
                  float* new_vis, float* new_lan, float* probs, int B, int P, int N, int C, hipStream_t st) {
  const float scale = 1.0f / sqrtf((float)C);
  const long R = (long)B * P;
  hipLaunchKernelGGL((xattn_scores_x3_kernel<NT, KS>), dim3(cdiv(R, 32), 2), dim3(256), 0, st, Qv, Kv, Qt, Kt, probs, R, P, N,
                     C, scale);
  TRIS_LAUNCH_CHECK();
  const size_t lds = (size_t)(((P + 15) / 16 * 16) * XLD + 64 * XLT) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)xattn_out_x3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       96 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  hipLaunchKernelGGL(xattn_out_x3_kernel<0>, dim3(C / 128, B), dim3(256), lds, st, Vv, Vt, probs, new_vis, new_lan, P, N, C);
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
  if (tris_get_gemm_mode() >= 1 && P <= 128 && (C == 1024 || C == 512 || C == 256)) {  // split-bf16: two-launch streaming path
#define TRIS_X3(NT_)                                                                                                \
  (C == 1024 ? launch_fwd_x3<NT_, 8>(Qv, Kv, Vv, Qt, Kt, Vt, new_vis, new_lan, probs, B, P, N, C, st)               \
             : C == 512 ? launch_fwd_x3<NT_, 4>(Qv, Kv, Vv, Qt, Kt, Vt, new_vis, new_lan, probs, B, P, N, C, st)     \
                        : launch_fwd_x3<NT_, 2>(Qv, Kv, Vv, Qt, Kt, Vt, new_vis, new_lan, probs, B, P, N, C, st))
    switch ((N + 15) / 16) {
      case 1: return TRIS_X3(1);
      case 2: return TRIS_X3(2);
      case 3: return TRIS_X3(3);
      default: return TRIS_X3(4);
    }
#undef TRIS_X3
  }
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
