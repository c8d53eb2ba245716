// Core of the MFMA GEMM / implicit-GEMM convolution family: the generic kernel (any shape / alignment, f32-input MFMA), the
// split-K reduce, the fast kernels (gemm_fast.h) and run_cfg<AK, BKIND>() -- the launch of ONE (tile, split-K, loop) configuration
// in one arithmetic.  Instantiated once per operand-kind pair by gemm_inst.hip (seven translation units that compile side by
// side: the family is ~350 kernels); gemm_conv.hip keeps the shape logic (heuristic, autotuner, entry points) and calls in
// through tris_internal_run_cfg.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "tris_hip.h"

namespace {

#include "gemm_params.h"
#include "amax.h"

constexpr int BK = 16;
constexpr int PAD = 4;

template <int BM, int BN, int AK, int BKIND, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int FM = WM / 32, FN = WN / 32;
  constexpr int PA = BM / 64, PB = BN / 64;  // float4 per thread per tile
  __shared__ float As[BK][BM + PAD];
  __shared__ float Bs[BK][BN + PAD];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = blockIdx.x;
  const int m0 = (tile / p.tiles_n) * BM;
  const int n0 = (tile % p.tiles_n) * BN;
  if (p.m_limit != nullptr && m0 >= *p.m_limit) return;   // (packed text rows: uniform exit before the first barrier)
  const int zb = blockIdx.z / p.splitk;  // batch index
  const int zs = blockIdx.z % p.splitk;  // k slice
  const int kbeg = zs * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);

  const float* __restrict__ A = p.A + (long)zb * p.sA;
  const float* __restrict__ Bp = p.B + (long)zb * p.sB;

  // ---- per-thread loader state ------------------------------------------------------------------------
  // A, k-contiguous kinds (ROWK / IM2COL): thread -> (row = tid>>2 + 64*pass, kofs = (tid&3)*4)
  // A COLK and B KN*: thread -> (k = tid / F4 + pass*RPP, col4 = (tid % F4)*4), F4 = tile_width/4
  int a_b[PA], a_iy0[PA], a_ix0[PA];
  bool a_ok[PA];
  if (AK == A_IM2COL) {
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      int m = m0 + (tid >> 2) + q * 64;
      a_ok[q] = m < p.M;
      int mm = a_ok[q] ? m : 0;
      int hw = p.gHo * p.gWo;
      a_b[q] = mm / hw;
      int r = mm - a_b[q] * hw;
      int oy = r / p.gWo, ox = r - oy * p.gWo;
      a_iy0[q] = oy * p.gStride - 1;
      a_ix0[q] = ox * p.gStride - 1;
    }
  }
  // B_KN_IM2COL: column j = (tap, ci) is fixed per thread
  int bj_tap = 0, bj_ci = 0;
  if (BKIND == B_KN_IM2COL) {
    constexpr int F4 = BN / 4;
    int j = n0 + (tid % F4) * 4;
    bj_tap = j / p.gC;
    bj_ci = j - bj_tap * p.gC;
  }

  float4 ra[PA], rb[PB];

  auto load_A = [&](int k0) {
    if (AK == A_ROWK) {
      const int kk = k0 + (tid & 3) * 4;
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        int m = m0 + (tid >> 2) + q * 64;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < p.M) {
          const float* src = A + (long)m * p.lda + kk;
          if (p.vecA && kk + 3 < kend) {
            v = ld4(src);
          } else {
            if (kk + 0 < kend) v.x = src[0];
            if (kk + 1 < kend) v.y = src[1];
            if (kk + 2 < kend) v.z = src[2];
            if (kk + 3 < kend) v.w = src[3];
          }
        }
        ra[q] = v;
      }
    } else if (AK == A_COLK) {
      constexpr int F4 = BM / 4, RPP = 256 / F4;
      const int mc = m0 + (tid % F4) * 4;
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        int kk = k0 + tid / F4 + q * RPP;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kk < kend) {
          const float* src = A + (long)kk * p.lda + mc;
          if (p.vecA && mc + 3 < p.M) {
            v = ld4(src);
          } else {
            if (mc + 0 < p.M) v.x = src[0];
            if (mc + 1 < p.M) v.y = src[1];
            if (mc + 2 < p.M) v.z = src[2];
            if (mc + 3 < p.M) v.w = src[3];
          }
        }
        ra[q] = v;
      }
    } else {  // A_IM2COL: k = tap*gC + ci
      const int kk = k0 + (tid & 3) * 4;
      if (p.vecA) {  // gC % 16 == 0: the whole 16-wide k tile sits inside one tap
        const int tap = kk / p.gC, ci = kk - tap * p.gC;
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int q = 0; q < PA; ++q) {
          int iy = a_iy0[q] + ky, ix = a_ix0[q] + kx;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (a_ok[q] && kk < kend && (unsigned)iy < (unsigned)p.gH && (unsigned)ix < (unsigned)p.gW)
            v = ld4(A + ((long)(a_b[q] * p.gH + iy) * p.gW + ix) * p.gC + ci);
          ra[q] = v;
        }
      } else {
#pragma unroll
        for (int q = 0; q < PA; ++q) {
          float t[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int k = kk + e;
            float v = 0.f;
            if (a_ok[q] && k < kend) {
              int tap = k / p.gC, ci = k - tap * p.gC;
              int ky = tap / 3, kx = tap - ky * 3;
              int iy = a_iy0[q] + ky, ix = a_ix0[q] + kx;
              if ((unsigned)iy < (unsigned)p.gH && (unsigned)ix < (unsigned)p.gW)
                v = A[((long)(a_b[q] * p.gH + iy) * p.gW + ix) * p.gC + ci];
            }
            t[e] = v;
          }
          ra[q] = make_float4(t[0], t[1], t[2], t[3]);
        }
      }
    }
  };

  auto load_B = [&](int k0) {
    if (BKIND == B_NK) {  // B[n*ldb + k]
      const int kk = k0 + (tid & 3) * 4;
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        int n = n0 + (tid >> 2) + q * 64;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < p.N) {
          const float* src = Bp + (long)n * p.ldb + kk;
          if (p.vecB && kk + 3 < kend) {
            v = ld4(src);
          } else {
            if (kk + 0 < kend) v.x = src[0];
            if (kk + 1 < kend) v.y = src[1];
            if (kk + 2 < kend) v.z = src[2];
            if (kk + 3 < kend) v.w = src[3];
          }
        }
        rb[q] = v;
      }
    } else {
      constexpr int F4 = BN / 4, RPP = 256 / F4;
      const int nc = n0 + (tid % F4) * 4;
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        int kk = k0 + tid / F4 + q * RPP;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kk < kend) {
          if (BKIND == B_KN || BKIND == B_KN_DGRAD) {
            const float* src;
            if (BKIND == B_KN) {
              src = Bp + (long)kk * p.ldb + nc;
            } else {  // k = tap'*Cout + co ; B[k][ci] = W[co][8 - tap'][ci]
              int tapp = kk / p.wCout, co = kk - tapp * p.wCout;
              src = Bp + ((long)co * 9 + (8 - tapp)) * p.wCin + nc;
            }
            if (p.vecB && nc + 3 < p.N) {
              v = ld4(src);
            } else {
              if (nc + 0 < p.N) v.x = src[0];
              if (nc + 1 < p.N) v.y = src[1];
              if (nc + 2 < p.N) v.z = src[2];
              if (nc + 3 < p.N) v.w = src[3];
            }
          } else {  // B_KN_IM2COL: k = output pixel, column = (tap, ci) of the gathered input
            int hw = p.gHo * p.gWo;
            int b = kk / hw;
            int r = kk - b * hw;
            int oy = r / p.gWo, ox = r - oy * p.gWo;
            if (p.vecB) {  // gC % 4 == 0: 4 consecutive columns share the tap
              if (nc < p.N) {
                int ky = bj_tap / 3, kx = bj_tap - ky * 3;
                int iy = oy * p.gStride - 1 + ky, ix = ox * p.gStride - 1 + kx;
                if ((unsigned)iy < (unsigned)p.gH && (unsigned)ix < (unsigned)p.gW)
                  v = ld4(Bp + ((long)(b * p.gH + iy) * p.gW + ix) * p.gC + bj_ci);
              }
            } else {
              float t[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                int j = nc + e;
                float x = 0.f;
                if (j < p.N) {
                  int tap = j / p.gC, ci = j - tap * p.gC;
                  int ky = tap / 3, kx = tap - ky * 3;
                  int iy = oy * p.gStride - 1 + ky, ix = ox * p.gStride - 1 + kx;
                  if ((unsigned)iy < (unsigned)p.gH && (unsigned)ix < (unsigned)p.gW)
                    x = Bp[((long)(b * p.gH + iy) * p.gW + ix) * p.gC + ci];
                }
                t[e] = x;
              }
              v = make_float4(t[0], t[1], t[2], t[3]);
            }
          }
        }
        rb[q] = v;
      }
    }
  };

  auto store_lds = [&]() {
    if (AK == A_ROWK || AK == A_IM2COL) {
      const int c = (tid & 3) * 4;
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        int r = (tid >> 2) + q * 64;
        As[c + 0][r] = ra[q].x;
        As[c + 1][r] = ra[q].y;
        As[c + 2][r] = ra[q].z;
        As[c + 3][r] = ra[q].w;
      }
    } else {
      constexpr int F4 = BM / 4, RPP = 256 / F4;
#pragma unroll
      for (int q = 0; q < PA; ++q)
        *reinterpret_cast<float4*>(&As[tid / F4 + q * RPP][(tid % F4) * 4]) = ra[q];
    }
    if (BKIND == B_NK) {
      const int c = (tid & 3) * 4;
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        int r = (tid >> 2) + q * 64;
        Bs[c + 0][r] = rb[q].x;
        Bs[c + 1][r] = rb[q].y;
        Bs[c + 2][r] = rb[q].z;
        Bs[c + 3][r] = rb[q].w;
      }
    } else {
      constexpr int F4 = BN / 4, RPP = 256 / F4;
#pragma unroll
      for (int q = 0; q < PB; ++q)
        *reinterpret_cast<float4*>(&Bs[tid / F4 + q * RPP][(tid % F4) * 4]) = rb[q];
    }
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (kbeg < kend) {
    load_A(kbeg);
    load_B(kbeg);
    store_lds();
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      const bool more = (k0 + BK) < kend;
      if (more) {
        load_A(k0 + BK);
        load_B(k0 + BK);
      }
      const int kh = lane >> 5, li = lane & 31;
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        float a[FM], b[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) a[i] = As[kk + kh][wm * WM + i * 32 + li];
#pragma unroll
        for (int j = 0; j < FN; ++j) b[j] = Bs[kk + kh][wn * WN + j * 32 + li];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
      if (more) {
        store_lds();
        __syncthreads();
      }
    }
  }

  // ---- epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) --------------
  const int li = lane & 31, kh = lane >> 5;
  unsigned g_am = 0u;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = n0 + wn * WN + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < p.M && col < p.N) {
          float v = acc[i][j][r];
          if (EPI == EPI_SLAB) {
            p.C[((long)blockIdx.z * p.M + row) * p.N + col] = v;
          } else {
            v *= p.alpha;
            if (p.bias_mode == 1) v += p.bias[col];
            else if (p.bias_mode == 2) v += p.bias[row];
            if (p.act == 1) v = fmaxf(v, 0.f);
            else if (p.act == 2) v = v / (1.0f + expf(-1.702f * v));
            if (p.resid) v += p.resid[(long)zb * p.sR + (long)row * p.ldr + col];
            p.C[(long)zb * p.sC + (long)row * p.ldc + col] = v;
            g_am = max(g_am, __builtin_bit_cast(unsigned, v) & 0x7fffffffu);
          }
        }
      }
    }
  if (EPI != EPI_SLAB && p.amax_out != nullptr) amax_commit(g_am, p.amax_out);
}

// Sum split-K slabs (ws[s][M][N]) and apply the standard epilogue.  VEC = 4: one thread owns 4 consecutive columns of a
// row (16-byte loads, 4 slabs in flight per trip) -- the reduce is a pure HBM/L2 stream and runs a few hundred times per step.
template <int VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, GemmParams p) {
  // grid-stride: the launch may hold fewer blocks than 1024-element pieces (option RED_GRID; a thread's sums are the same either way)
  const long total = (long)p.M * p.N;
  const long stride = (long)gridDim.x * blockDim.x * VEC;
  unsigned am = 0u;
  for (long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC; idx < total; idx += stride) {
    if (p.m_limit != nullptr && idx / p.N >= (long)*p.m_limit) break;   // (rows ascend with idx)
    float v[VEC];
#pragma unroll
    for (int t = 0; t < VEC; ++t) v[t] = 0.f;
    if (VEC == 4) {
      int s = 0;
      for (; s + 3 < S; s += 4) {  // fixed summation order s = 0, 1, 2, ... (deterministic)
        const float4 a = ld4(ws + (long)s * total + idx), b = ld4(ws + (long)(s + 1) * total + idx);
        const float4 c = ld4(ws + (long)(s + 2) * total + idx), d = ld4(ws + (long)(s + 3) * total + idx);
        v[0] = (((v[0] + a.x) + b.x) + c.x) + d.x;
        v[1] = (((v[1] + a.y) + b.y) + c.y) + d.y;
        v[2] = (((v[2] + a.z) + b.z) + c.z) + d.z;
        v[3] = (((v[3] + a.w) + b.w) + c.w) + d.w;
      }
      for (; s < S; ++s) {
        const float4 a = ld4(ws + (long)s * total + idx);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
      }
    } else {
      for (int s = 0; s < S; ++s) v[0] += ws[(long)s * total + idx];
    }
    const int row = (int)(idx / p.N), col0 = (int)(idx - (long)row * p.N);
#pragma unroll
    for (int t = 0; t < VEC; ++t) {
      const int col = col0 + t;
      float x = v[t] * p.alpha;
      if (p.bias_mode == 1) x += p.bias[col];
      else if (p.bias_mode == 2) x += p.bias[row];
      if (p.act == 1) x = fmaxf(x, 0.f);
      else if (p.act == 2) x = x / (1.0f + expf(-1.702f * x));
      if (p.resid) x += p.resid[(long)row * p.ldr + col];
      v[t] = x;
    }
    if (VEC == 4 && (p.ldc & 3) == 0)
      *reinterpret_cast<float4*>(p.C + (long)row * p.ldc + col0) = make_float4(v[0], v[1], v[2], v[3]);
    else
#pragma unroll
      for (int t = 0; t < VEC; ++t) p.C[(long)row * p.ldc + col0 + t] = v[t];
#pragma unroll
    for (int t = 0; t < VEC; ++t) am = max(am, __builtin_bit_cast(unsigned, v[t]) & 0x7fffffffu);
  }
  if (p.amax_out != nullptr) amax_commit(am, p.amax_out);   // (every lane arrives here)
}

extern "C" { extern __attribute__((visibility("hidden"))) int tris_internal_red_grid; }      // option RED_GRID (gemm_conv.hip)
extern "C" { extern __attribute__((visibility("hidden"))) int tris_internal_reduce_wide; }   // option REDUCE_WIDE (gemm_conv.hip)
extern "C" { extern __attribute__((visibility("hidden"))) int tris_internal_xcd_order; }     // option XCD_ORDER (gemm_conv.hip)
extern "C" { extern __attribute__((visibility("hidden"))) int tris_internal_fuse_splitk; }   // option FUSE_SPLITK (gemm_conv.hip)
extern "C" { extern __attribute__((visibility("hidden"))) long tris_internal_fused_count; }   // launches that took the fused finish
// The same sum for outputs too SMALL to fill the chip with one thread per four columns (a 64 x 64 weight gradient cut into 128
// slabs is 4 blocks of the kernel above, each thread walking 128 slabs four at a time: ~30 us of latency for 2 MB): a block of
// G waves owns 64 float4 columns, wave w takes the slabs s = w, w + G, ..., the partial sums meet in LDS in a fixed order
// (deterministic; the order differs from the kernel above -- ((s0 + sG + ..) + (s1 + s(G+1) + ..)) + ..).  G = 4 by default: a
// 1024-thread block (G = 16) is faster on an idle device but waits for sixteen free wave slots of one CU inside the step (47 us
// on average against ~5, measured); option REDUCE_WIDE = G.
template <int G>   // slab groups = waves per block
__global__ __launch_bounds__(64 * G) void splitk_reduce_wide_kernel(const float* __restrict__ ws, int S, GemmParams p) {
  __shared__ float4 part[G][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long total = (long)p.M * p.N;
  const long idx = ((long)blockIdx.x * 64 + c) * 4;
  const bool in = idx < total && !(p.m_limit != nullptr && idx / p.N >= (long)*p.m_limit);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (in) {
    int s = g;
    for (; s + 3 * G < S; s += 4 * G) {
      const float4 x0 = ld4(ws + (long)s * total + idx), x1 = ld4(ws + (long)(s + G) * total + idx);
      const float4 x2 = ld4(ws + (long)(s + 2 * G) * total + idx), x3 = ld4(ws + (long)(s + 3 * G) * total + idx);
      a.x = (((a.x + x0.x) + x1.x) + x2.x) + x3.x; a.y = (((a.y + x0.y) + x1.y) + x2.y) + x3.y;
      a.z = (((a.z + x0.z) + x1.z) + x2.z) + x3.z; a.w = (((a.w + x0.w) + x1.w) + x2.w) + x3.w;
    }
    for (; s < S; s += G) {
      const float4 x0 = ld4(ws + (long)s * total + idx);
      a.x += x0.x; a.y += x0.y; a.z += x0.z; a.w += x0.w;
    }
  }
  part[g][c] = a;
  __syncthreads();
  if (g != 0) return;
  float v[4] = {part[0][c].x, part[0][c].y, part[0][c].z, part[0][c].w};
#pragma unroll
  for (int w = 1; w < G; ++w) { v[0] += part[w][c].x; v[1] += part[w][c].y; v[2] += part[w][c].z; v[3] += part[w][c].w; }
  unsigned am = 0u;
  if (in) {
    const int row = (int)(idx / p.N), col0 = (int)(idx - (long)row * p.N);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int col = col0 + t;
      float x = v[t] * p.alpha;
      if (p.bias_mode == 1) x += p.bias[col];
      else if (p.bias_mode == 2) x += p.bias[row];
      if (p.act == 1) x = fmaxf(x, 0.f);
      else if (p.act == 2) x = x / (1.0f + expf(-1.702f * x));
      if (p.resid) x += p.resid[(long)row * p.ldr + col];
      v[t] = x;
      am = max(am, __builtin_bit_cast(unsigned, x) & 0x7fffffffu);
    }
    if ((p.ldc & 3) == 0)
      *reinterpret_cast<float4*>(p.C + (long)row * p.ldc + col0) = make_float4(v[0], v[1], v[2], v[3]);
    else
#pragma unroll
      for (int t = 0; t < 4; ++t) p.C[(long)row * p.ldc + col0 + t] = v[t];
  }
  if (p.amax_out != nullptr) amax_commit(am, p.amax_out);   // (all lanes of wave 0 take part in the shuffles)
}

#include "gemm_fast.h"

static inline long red_blocks(long want) { return tris_internal_red_grid > 0 && want > tris_internal_red_grid ? tris_internal_red_grid : want; }

// launch one configuration (+ split-K reduce).  mode: 0 = f32-input MFMA, 1 = split-bf16 x3, 3 = two-piece fp16 h2,
// 4 = h2 with both operands arriving as fp16 piece planes (gemm_fast.h PREC 4)
template <int AK, int BKIND>
int run_cfg(GemmParams p, int batch, float* ws, hipStream_t st, Cfg cfg, int mode) {
  int bm = cfg.bm, bn = cfg.bn, splitk = ws != nullptr ? cfg.splitk : 1;   // (a tuned split-K choice without a workspace: one slice)
  const bool fast = gemm_fast_ok(p);
  if (bn == 32 && !fast) bm = bn = 64;  // the 128x32 tile exists in the fast kernel only
  const bool xtra = p.pre_out != nullptr || p.dact_x != nullptr;   // (tris_gemm_epilogue_next: classic loop, one slice, fast kernel)
  if (mode == 4 && !fast) return (int)hipErrorInvalidValue;   // (operand planes exist for the fast kernel only: the caller checks)
  const bool pipe = cfg.pipe == 1 && (mode == 1 || mode == 3 || mode == 4) && fast && bn != 32 && !xtra;   // (the pipelined loop: x3 and h2)
  // cfg.pipe 2 | 3: the LDS-DMA loop of gemm_fast.h (operand planes, row-major A x B^T): 2 = eight waves per 128 x 128 tile (2 x 4),
  // 3 = four (2 x 2: 64 x 64 per wave); the other tiles have one form
  // 4 | 5: the same with the XCD-contiguous tile order (gemm_fast.h xcd_remap 2)
  const bool glds = cfg.pipe >= 2 && mode == 4 && fast && !xtra && bn != 32 && AK == A_ROWK && BKIND == B_NK;
  const int gform = glds ? (cfg.pipe >= 4 ? cfg.pipe - 2 : cfg.pipe) : 0;   // 2: eight waves, 3: four
  if (bm == 256 && !pipe && !glds) bm = 128;     // the 256-row tile exists in the pipelined / LDS-DMA forms only
  int tiles_m = cdiv(p.M, bm), tiles_n = cdiv(p.N, bn);
  p.tiles_n = tiles_n;
  const int kalign = fast ? 32 : BK;
  p.splitk = splitk;
  p.kchunk = cdiv(cdiv(p.K, splitk), kalign) * kalign;
  if (splitk > 1) splitk = cdiv(p.K, p.kchunk), p.splitk = splitk;
  dim3 grid((unsigned)(tiles_m * tiles_n), 1, (unsigned)(batch * splitk));
  // 3x3 weight gradients: the tap tiles of a k slice share their operands -> one XCD per slice
  p.xcd_remap = (fast && BKIND == B_KN_IM2COL && batch == 1 && splitk >= 8 && splitk % 8 == 0 && tiles_m * tiles_n > 1) ? 1 : 0;
  if (glds && cfg.pipe >= 4) p.xcd_remap = 2;
  // developer option XCD_ORDER = 1: the XCD-contiguous tile order for EVERY fast-kernel launch with at least two tiles per XCD that has
  // no placement of its own (consecutive tiles share their A rows: one L2 instead of eight fetches them); 0: never
  // (2: only the products that split their operands in the kernel -- the transformer towers' -- keep the tuner's choice for the plane products)
  if (tris_internal_xcd_order >= 0 && fast && p.xcd_remap != 1 && !(tris_internal_xcd_order == 2 && mode == 4))
    p.xcd_remap = (tris_internal_xcd_order && tiles_m * tiles_n >= 16 && tiles_n > 1) ? 2 : 0;
  float* Cfinal = p.C;
  // fused split-K finish (gemm_fast.h): the fast kernels, one batch, few slices, an armed ticket array with a slot per tile
  int* const tk = p.tickets;
  p.tickets = nullptr;
  // (not where the wide reduce kernel would run -- many slices of a small output: its summation order differs, and the fused finish
  //  is to be the separate launch's result bit for bit: static-tile runs and their tests see no change in arithmetic)
  const bool wide = (p.N & 3) == 0 && al16(ws) && al16(Cfinal) && splitk >= 8 && (long)p.M * p.N / 4 < 128 * 256 && tris_internal_reduce_wide;
  const bool fuse = splitk > 1 && splitk <= tris_internal_fuse_splitk && splitk <= TRIS_FUSE_SPLITK_MAX && tk != nullptr && fast &&
                    batch == 1 && tiles_m * tiles_n <= p.tickets_n && !wide;
  if (fuse) {
    __atomic_fetch_add(&tris_internal_fused_count, 1L, __ATOMIC_RELAXED);
    p.tickets = tk;
    p.Cfin = Cfinal;
    p.vecCfin = (p.N % 4 == 0) && al16(Cfinal) && (p.ldc % 4 == 0) && (!p.resid || (al16(p.resid) && p.ldr % 4 == 0)) &&
                (p.bias_mode != 1 || al16(p.bias));
  }
  if (splitk > 1)
    p.vecC = (p.N % 4 == 0) && al16(ws);
  else
    p.vecC = (p.N % 4 == 0) && al16(p.C) && (p.ldc % 4 == 0) && (p.sC % 4 == 0) &&
             (!p.resid || (al16(p.resid) && p.ldr % 4 == 0 && p.sR % 4 == 0)) && (p.bias_mode != 1 || al16(p.bias));
// waves per block: 128 x 128 in x3 / h2: eight (2 x 4); 128 x 32: two (2 x 1); everything else four (2 x 2)
#define TRIS_NW(BM_, BN_, PREC_) ((BM_) == 128 && (BN_) == 128 && (PREC_) != 0 ? 8 : (BN_) == 32 ? 2 : 4)
#define TRIS_FAST(BM_, BN_, EPI_, PREC_)                                                                       \
  hipLaunchKernelGGL((gemm_fast_kernel<BM_, BN_, AK, BKIND, EPI_, PREC_, TRIS_NW(BM_, BN_, PREC_)>), grid,    \
                     dim3(64 * TRIS_NW(BM_, BN_, PREC_)), 0, st, p)
#define TRIS_FAST_EPI(BM_, BN_, EPI_)                                       \
  do {                                                                      \
    if (mode == 1) TRIS_FAST(BM_, BN_, EPI_, 1);                            \
    else if (mode == 3) TRIS_FAST(BM_, BN_, EPI_, 3);                       \
    else if (mode == 4) TRIS_FAST(BM_, BN_, EPI_, 4);                       \
    else TRIS_FAST(BM_, BN_, EPI_, 0);                                      \
  } while (0)
#define TRIS_GO(BM_, BN_, GENERIC_OK_)                                                             \
  do {                                                                                             \
    if (fast) {                                                                                    \
      if (splitk > 1) {                                                                            \
        p.C = ws;                                                                                  \
        TRIS_FAST_EPI(BM_, BN_, EPI_SLAB);                                                         \
      } else if (xtra) {                                                                           \
        if constexpr (AK == A_ROWK && (BKIND == B_NK || BKIND == B_KN)) TRIS_FAST_EPI(BM_, BN_, EPI_XTRA); \
        else return (int)hipErrorInvalidValue;                                                     \
      } else {                                                                                     \
        TRIS_FAST_EPI(BM_, BN_, EPI_STD);                                                          \
      }                                                                                            \
    } else if (GENERIC_OK_) {                                                                      \
      if (splitk > 1) {                                                                            \
        p.C = ws;                                                                                  \
        hipLaunchKernelGGL((gemm_kernel<(GENERIC_OK_ ? BM_ : 64), (GENERIC_OK_ ? BN_ : 64), AK, BKIND, EPI_SLAB>), grid, dim3(256), 0, st, p); \
      } else {                                                                                     \
        hipLaunchKernelGGL((gemm_kernel<(GENERIC_OK_ ? BM_ : 64), (GENERIC_OK_ ? BN_ : 64), AK, BKIND, EPI_STD>), grid, dim3(256), 0, st, p);  \
      }                                                                                            \
    }                                                                                              \
  } while (0)
#define TRIS_PIPE_ONE(BM_, BN_, NW_, NWM_, EPI_)                                                                                   \
  do {                                                                                                                             \
    if (mode == 3)                                                                                                                 \
      hipLaunchKernelGGL((gemm_fast_kernel<BM_, BN_, AK, BKIND, EPI_, 3, NW_, 16, 2, NWM_>), grid, dim3(NW_ * 64), 0, st, p);      \
    else if (mode == 4)                                                                                                            \
      hipLaunchKernelGGL((gemm_fast_kernel<BM_, BN_, AK, BKIND, EPI_, 4, NW_, 16, 2, NWM_>), grid, dim3(NW_ * 64), 0, st, p);      \
    else                                                                                                                           \
      hipLaunchKernelGGL((gemm_fast_kernel<BM_, BN_, AK, BKIND, EPI_, 1, NW_, 16, 2, NWM_>), grid, dim3(NW_ * 64), 0, st, p);      \
  } while (0)
#define TRIS_PIPE_GO(BM_, BN_, NW_, NWM_)                        \
  do {                                                           \
    if (splitk > 1) {                                            \
      p.C = ws;                                                  \
      TRIS_PIPE_ONE(BM_, BN_, NW_, NWM_, EPI_SLAB);              \
    } else {                                                     \
      TRIS_PIPE_ONE(BM_, BN_, NW_, NWM_, EPI_STD);               \
    }                                                            \
  } while (0)
#define TRIS_GLDS_GO(BM_, BN_, NW_, NWM_)                                                                                          \
  do {                                                                                                                           \
    if (splitk > 1) {                                                                                                            \
      p.C = ws;                                                                                                                  \
      hipLaunchKernelGGL((gemm_fast_kernel<BM_, BN_, AK, BKIND, EPI_SLAB, 4, NW_, 32, 4, NWM_>), grid, dim3(NW_ * 64), 0, st, p); \
    } else {                                                                                                                     \
      hipLaunchKernelGGL((gemm_fast_kernel<BM_, BN_, AK, BKIND, EPI_STD, 4, NW_, 32, 4, NWM_>), grid, dim3(NW_ * 64), 0, st, p);  \
    }                                                                                                                            \
  } while (0)
  bool launched = false;
  if constexpr (AK == A_ROWK && BKIND == B_NK) {
    if (glds) {
      launched = true;
      if (bm == 256 && bn == 128) TRIS_GLDS_GO(256, 128, 8, 4);
      else if (bm == 128 && bn == 128 && gform == 3) TRIS_GLDS_GO(128, 128, 4, 2);
      else if (bm == 128 && bn == 128) TRIS_GLDS_GO(128, 128, 8, 2);
      else if (bm == 128 && bn == 64) TRIS_GLDS_GO(128, 64, 4, 2);
      else TRIS_GLDS_GO(64, 64, 4, 2);
    }
  }
#undef TRIS_GLDS_GO
  if (launched) {
  } else
  if (pipe) {
    if (bm == 256 && bn == 128) TRIS_PIPE_GO(256, 128, 8, 4);
    else if (bm == 128 && bn == 128) TRIS_PIPE_GO(128, 128, 8, 2);
    else if (bm == 128 && bn == 64) TRIS_PIPE_GO(128, 64, 4, 2);
    else TRIS_PIPE_GO(64, 64, 4, 2);
  } else if (bm == 128 && bn == 128) TRIS_GO(128, 128, true);
  else if (bm == 128 && bn == 64) TRIS_GO(128, 64, true);
  else if (bm == 128 && bn == 32) TRIS_GO(128, 32, false);   // (fast kernel only: run_cfg turned it into 64 x 64 otherwise)
  else TRIS_GO(64, 64, true);
#undef TRIS_GO
#undef TRIS_PIPE_GO
#undef TRIS_PIPE_ONE
#undef TRIS_FAST_EPI
#undef TRIS_FAST
#undef TRIS_NW
  TRIS_LAUNCH_CHECK();
  if (splitk > 1 && !fuse) {
    p.C = Cfinal;
    long total = (long)p.M * p.N;
    if ((p.N & 3) == 0 && al16(ws) && al16(p.C) && splitk >= 8 && total / 4 < 128 * 256 && tris_internal_reduce_wide)
    {
      if (tris_internal_reduce_wide >= 16)
        hipLaunchKernelGGL(splitk_reduce_wide_kernel<16>, dim3(cdiv(total / 4, 64)), dim3(1024), 0, st, ws, splitk, p);
      else if (tris_internal_reduce_wide >= 8)
        hipLaunchKernelGGL(splitk_reduce_wide_kernel<8>, dim3(cdiv(total / 4, 64)), dim3(512), 0, st, ws, splitk, p);
      else
        hipLaunchKernelGGL(splitk_reduce_wide_kernel<4>, dim3(cdiv(total / 4, 64)), dim3(256), 0, st, ws, splitk, p);
    }
    else if ((p.N & 3) == 0 && al16(ws) && al16(p.C))
      hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3(red_blocks(cdiv(total / 4, 256))), dim3(256), 0, st, ws, splitk, p);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(red_blocks(cdiv(total, 256))), dim3(256), 0, st, ws, splitk, p);
    TRIS_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace
