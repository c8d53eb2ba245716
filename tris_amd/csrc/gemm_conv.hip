// MFMA GEMM / implicit-GEMM convolution core for the TRIS Stage-1 hot path (gfx950).
//
// One templated kernel serves every dense product on the path:
//   * Linear / 1x1 conv (NHWC => plain GEMM), their dgrad (NN) and wgrad (TN, split-K)
//   * 3x3 conv forward, dgrad and wgrad as implicit GEMM (the im2col gather lives in the tile loaders)
//   * batched products of the cross-modal attention
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- f32 in / f32 accumulate, bit-equal to an fmaf chain.  That is what the
// 1e-3 fp32 parity bar of the north star needs; roofline = 157.3 TFLOP/s (MI355X_MICROARCH.md).
//
// Tile: BM x BN x 16, 256 threads = 4 waves in a 2x2 grid, each wave owns (BM/2)x(BN/2) as 32x32 MFMA fragments.
// LDS image is k-major (As[k][m], Bs[k][n], row pad 4) so a fragment read is 32 consecutive floats per half-wave
// (conflict-free ds_read_b32).  Global loads for tile t+1 are issued before the MFMA block of tile t
// (register staging), stored to LDS after it.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

#include "common.h"
#include "tris_hip.h"

namespace {

#include "gemm_params.h"

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
          }
        }
      }
    }
}

// Sum split-K slabs (ws[s][M][N]) and apply the standard epilogue.  VEC = 4: one thread owns 4 consecutive columns of a
// row (16-byte loads, 4 slabs in flight per trip) -- the reduce is a pure HBM/L2 stream and runs a few hundred times per step.
template <int VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, GemmParams p) {
  const long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  const long total = (long)p.M * p.N;
  if (idx >= total) return;
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
}

// arithmetic of the fast kernels: 0 = f32-input MFMA, 1 = split-bf16 x3 (6 bf16 MFMAs per product, fp32-class accuracy),
// 2 = split-bf16 x2 (3 bf16 MFMAs per product, 16-bit significands: between fp32 and TF32)
// Process-wide default + per-thread override, both read on the HOST when a product is launched (a launch is otherwise
// stateless).  The override exists for callers that run some products in another arithmetic from their own thread (the
// autograd engine thread running weight-gradient products in x2) without touching what other threads launch.
static int g_mode_default = 1;
static thread_local int g_mode_thread = -1;
#define g_gemm_mode (g_mode_thread >= 0 ? g_mode_thread : g_mode_default)
static int g_x3_waves = (getenv("TRIS_X3_WAVES") && atoi(getenv("TRIS_X3_WAVES")) == 4) ? 4 : 8;  // waves per 128x128 x3 block

#include "gemm_fast.h"

// "h2" for ONE product: tris_h2_next() arms the calling thread; the next dense product launched from it (tris_gemm_f32,
// tris_gemm_bnstat_f32, tris_gemm_bnbwd_f32) runs with two fp16 pieces per operand (PREC 3) and these operand scales, whatever
// the process-wide mode -- if the fast kernel serves its shape; otherwise it runs as usual.  One shot.
struct H2Next { const unsigned* a; const unsigned* b; float sa, sb; bool armed; };
static thread_local H2Next g_h2_next = {nullptr, nullptr, 0.f, 0.f, false};
// every entry point that honours the arming TAKES it first thing (h2_take), whether or not it then launches anything: an arming
// never survives the call it was made for
static H2Next h2_take() {
  H2Next n = g_h2_next;
  g_h2_next.armed = false;
  return n;
}
struct H2Guard {
  int saved;
  bool on;
  H2Guard(GemmParams& p, const H2Next& n) : saved(g_mode_thread), on(false) {
    if (!n.armed) return;
    if (!(p.fastA && p.fastB && p.K % 32 == 0 && p.M >= 4 && p.N >= 4)) return;
    p.h2_amaxA = n.a; p.h2_amaxB = n.b; p.h2_sA = n.sa; p.h2_sB = n.sb;
    g_mode_thread = 3;
    on = true;
  }
  ~H2Guard() { if (on) g_mode_thread = saved; }
};

// TRIS_FORCE_PIPE=0|1 (read per call: tests switch it at run time) overrides the loop structure of the x3 products
static int forced_pipe() {
  const char* e = getenv("TRIS_FORCE_PIPE");
  return !e ? -1 : (e[0] == '1' ? 1 : 0);
}
static bool pipe_ok(const GemmParams& p) {
  return g_gemm_mode == 1 && p.fastA && p.fastB && (p.K % 32 == 0) && p.M >= 4 && p.N >= 4;
}
// static choice of the loop structure (the autotuner times both)
static int default_pipe(const GemmParams& p, int bm, int bn, int splitk) {
  if (!pipe_ok(p) || bn == 32) return 0;
  const int f = forced_pipe();
  if (f >= 0) return f;
  // measured (tools/gemm_bench.py, autotuned tiles, classic | pipelined): the 3x3 convolutions from 128 channels up gain
  // 3-13 % (fwd 40x40x256: 165 -> 175, 20x20x512: 136 -> 154, wgrad 20x20x512: 154 -> 174 TFLOP/s); the short-K 1x1
  // products and the transformer GEMMs are on par or a few % slower -> static default by kind, the autotuner times both
  static const int dflt = getenv("TRIS_PIPE") ? (getenv("TRIS_PIPE")[0] == '0' ? 0 : 1) : -1;
  if (dflt >= 0) return dflt;
  return p.gC >= 128 ? 1 : 0;
}

// ---- configuration = (tile, split-K) ----------------------------------------------------------------------------------
struct Cfg { int bm, bn, splitk, nw, pipe; };  // nw: waves per 128x128 block of the split-bf16 kernels (8 = 2x4 wave grid, 4 = 2x2)
// pipe = 1: the pipelined loop of gemm_fast.h (x3 only: two 16-deep LDS stages, one barrier per K tile); tiles 256x128 exist
// in that form only

// Tile / split-K choice by a small cost model (cycles on the MFMA pipe); also the starting point of the autotuner.
//   per-wave cycles per 32-deep k step = (BM/64)*(BN/64)*c; a block owns a CU's 4 SIMDs; blocks beyond the 256 CUs queue.
//   Split-K adds a slab round trip + a reduce launch.  Smaller tiles pay extra LDS / L2 traffic.
static Cfg heuristic_cfg(const GemmParams& p, int batch, const float* ws, long ws_bytes) {
  const bool can_split = (batch == 1 && ws != nullptr && p.K >= 512);
  int bm = 64, bn = 64, splitk = 1;
  if (p.stat_part != nullptr) {  // fused BN statistics: fixed 128-row tiles (the caller sizes the partial buffer), no split-K
    bm = 128;
    bn = p.N <= 32 ? 32 : p.N <= 64 ? 64 : 128;  // (run_cfg turns 128x32 into 64x64 when the fast kernel does not apply)
  } else if (can_split && p.K >= 4096) {
    // long-K reductions (wgrad over pixels): 2 co-resident blocks per CU keep the MFMA pipe busy across the
    // barrier / staging phases, and ~512 blocks smooth the wave quantisation -> split K until there are ~512 blocks
    if (p.N <= 64) { bn = 64; bm = p.M >= 128 ? 128 : 64; }
    else if (p.M >= 128 && p.N >= 128) { bm = bn = 128; }
    const long tiles = (long)cdiv(p.M, bm) * cdiv(p.N, bn);
    splitk = (int)min((long)cdiv(512, tiles), (long)(p.K / 256));
    const long per = (long)p.M * p.N * (long)sizeof(float);
    if ((long)splitk * per > ws_bytes) splitk = (int)(ws_bytes / per);
    if (splitk < 2) splitk = 1;
  } else {
    static const int cand[4][2] = {{128, 128}, {128, 64}, {64, 64}, {128, 32}};
    static const double pen[4] = {1.0, 1.08, 1.2, 1.15};
    const bool fastk = p.fastA && p.fastB && (p.K % 32 == 0) && p.M >= 4 && p.N >= 4;
    double best = 1e30;
    for (int c = 0; c < 4; ++c) {
      const int cbm = cand[c][0], cbn = cand[c][1];
      if (c != 2 && p.M < 96) continue;          // 128-row tiles on a tiny M waste the MFMA
      if (c == 0 && p.N <= 64) continue;
      if (c == 3 && !(p.N <= 32 && fastk)) continue;  // 32-wide outputs (stem convolutions): half of a 64-wide tile would be padding
      const long tiles = (long)cdiv(p.M, cbm) * cdiv(p.N, cbn) * batch;
      // MFMA cycles of one 32x32 fragment pair per 32-deep step: 16 f32 MFMAs x 64, or 12 bf16 MFMAs x 32 (x3 mode)
      const double per_k32 = (cbm / 64) * (cbn / 64.0) * (g_gemm_mode == 1 ? 384.0 + 250.0 : (g_gemm_mode == 2 || g_gemm_mode == 3) ? 192.0 + 200.0 : 1024.0) * pen[c];
      const int smax = can_split ? (int)min((long)64, (long)(p.K / 256)) : 1;
      for (int sk = 1; sk <= smax; sk = (sk < 4 ? sk + 1 : sk + sk / 2)) {
        if (sk > 1 && (long)sk * p.M * p.N * (long)sizeof(float) > ws_bytes) break;
        const double ksteps = (double)cdiv(cdiv(p.K, sk), 32);
        const double blocks = (double)tiles * sk;
        double t = (ksteps * per_k32 + 3000.0) * (blocks <= 256.0 ? 1.0 : blocks / 256.0);
        if (sk > 1) t += 14000.0 + 2.0 * sk * (double)p.M * p.N * 4.0 / 2000.0;  // reduce launch + slab bytes @ ~2 kB/cycle
        if (t < best) { best = t; bm = cbm; bn = cbn; splitk = sk; }
      }
    }
  }
  {  // developer knob: TRIS_FORCE_TILE=128x128|128x64|64x64|128x32 overrides the tile choice (tools/x3_probe.py)
    const char* e = getenv("TRIS_FORCE_TILE");  // read per call: tests switch it at run time
    const int forced = !e ? 0 : (!strcmp(e, "128x128") ? 1 : !strcmp(e, "128x64") ? 2 : !strcmp(e, "64x64") ? 3 : !strcmp(e, "128x32") ? 4
                                 : !strcmp(e, "256x128") ? 5 : 0);
    if (forced == 1 && p.N > 64) { bm = 128; bn = 128; }
    if (forced == 2) { bm = 128; bn = 64; }
    if (forced == 3) { bm = 64; bn = 64; }
    if (forced == 4 && p.N <= 32) { bm = 128; bn = 32; }
    if (forced == 5 && p.N > 64 && p.M >= 256 && p.stat_part == nullptr) { bm = 256; bn = 128; }   // (pipelined x3 loop only: run_cfg falls back)
  }
  Cfg c = {bm, bn, splitk, g_x3_waves, default_pipe(p, bm, bn, splitk)};
  return c;
}

// launch one configuration (+ split-K reduce)
template <int AK, int BKIND>
int run_cfg(GemmParams p, int batch, float* ws, hipStream_t st, Cfg cfg) {
  int bm = cfg.bm, bn = cfg.bn, splitk = cfg.splitk;
  const int nw = cfg.nw;
  const bool fast = p.fastA && p.fastB && (p.K % 32 == 0) && p.M >= 4 && p.N >= 4;
  if (bn == 32 && !fast) bm = bn = 64;  // the 128x32 tile exists in the fast kernel only
  const bool pipe = cfg.pipe && pipe_ok(p) && BKIND != B_NK_PRE && bn != 32;
  if (bm == 256 && !pipe) bm = 128;     // the 256-row tile exists in the pipelined form only
  int tiles_m = cdiv(p.M, bm), tiles_n = cdiv(p.N, bn);
  p.tiles_n = tiles_n;
  const int kalign = fast ? 32 : BK;
  p.splitk = splitk;
  p.kchunk = cdiv(cdiv(p.K, splitk), kalign) * kalign;
  if (splitk > 1) splitk = cdiv(p.K, p.kchunk), p.splitk = splitk;
  dim3 grid((unsigned)(tiles_m * tiles_n), 1, (unsigned)(batch * splitk));
  {  // 3x3 weight gradients: the tap tiles of a k slice share their operands -> one XCD per slice (TRIS_XCD_REMAP=0: A/B knob)
    static const bool remap_ok = !(getenv("TRIS_XCD_REMAP") && getenv("TRIS_XCD_REMAP")[0] == '0');
    p.xcd_remap = (remap_ok && fast && BKIND == B_KN_IM2COL && batch == 1 && splitk >= 8 && splitk % 8 == 0 && tiles_m * tiles_n > 1) ? 1 : 0;
  }
  float* Cfinal = p.C;
  static const bool vec_epi_ok = !(getenv("TRIS_VEC_EPILOGUE") && getenv("TRIS_VEC_EPILOGUE")[0] == '0');  // developer A/B knob
  if (splitk > 1)
    p.vecC = vec_epi_ok && (p.N % 4 == 0) && al16(ws);
  else
    p.vecC = vec_epi_ok && (p.N % 4 == 0) && al16(p.C) && (p.ldc % 4 == 0) && (p.sC % 4 == 0) &&
             (!p.resid || (al16(p.resid) && p.ldr % 4 == 0 && p.sR % 4 == 0)) && (p.bias_mode != 1 || al16(p.bias));
#define TRIS_FAST(BM_, BN_, EPI_, PREC_)                                                                              \
  do {                                                                                                                 \
    if (BM_ == 128 && BN_ == 128 && nw == 8)                                                                            \
      hipLaunchKernelGGL((gemm_fast_kernel<BM_, BN_, AK, BKIND, EPI_, PREC_, (BM_ == 128 && BN_ == 128 ? 8 : BN_ == 32 ? 2 : 4)>),  \
                         grid, dim3(512), 0, st, p);                                                       \
    else  /* 128x32: two waves (2x1); everything else four (2x2) */                                                     \
      hipLaunchKernelGGL((gemm_fast_kernel<BM_, BN_, AK, BKIND, EPI_, PREC_, (BN_ == 32 ? 2 : 4)>), grid,               \
                         dim3(BN_ == 32 ? 128 : 256), 0, st, p);                                                       \
  } while (0)
#define TRIS_GO(BM_, BN_)                                                                          \
  do {                                                                                             \
    if (fast) {                                                                                    \
      if (splitk > 1) {                                                                            \
        p.C = ws;                                                                                  \
        if (g_gemm_mode == 1) TRIS_FAST(BM_, BN_, EPI_SLAB, 1);                                    \
        else if (g_gemm_mode == 2) TRIS_FAST(BM_, BN_, EPI_SLAB, 2);                               \
        else if (g_gemm_mode == 3) TRIS_FAST(BM_, BN_, EPI_SLAB, 3);                               \
        else hipLaunchKernelGGL((gemm_fast_kernel<BM_, BN_, AK, BKIND, EPI_SLAB, 0>), grid, dim3(256), 0, st, p); \
      } else {                                                                                     \
        if (g_gemm_mode == 1) TRIS_FAST(BM_, BN_, EPI_STD, 1);                                     \
        else if (g_gemm_mode == 2) TRIS_FAST(BM_, BN_, EPI_STD, 2);                                \
        else if (g_gemm_mode == 3) TRIS_FAST(BM_, BN_, EPI_STD, 3);                                \
        else hipLaunchKernelGGL((gemm_fast_kernel<BM_, BN_, AK, BKIND, EPI_STD, 0>), grid, dim3(256), 0, st, p);  \
      }                                                                                            \
    } else if (splitk > 1) {                                                                       \
      p.C = ws;                                                                                    \
      hipLaunchKernelGGL((gemm_kernel<BM_, BN_, AK, BKIND, EPI_SLAB>), grid, dim3(256), 0, st, p); \
    } else {                                                                                       \
      hipLaunchKernelGGL((gemm_kernel<BM_, BN_, AK, BKIND, EPI_STD>), grid, dim3(256), 0, st, p);  \
    }                                                                                              \
  } while (0)
#define TRIS_GO_FAST_ONLY(BM_, BN_)                                                                \
  do {                                                                                             \
    if (splitk > 1) {                                                                              \
      p.C = ws;                                                                                    \
      if (g_gemm_mode == 1) TRIS_FAST(BM_, BN_, EPI_SLAB, 1);                                      \
      else if (g_gemm_mode == 2) TRIS_FAST(BM_, BN_, EPI_SLAB, 2);                                 \
      else if (g_gemm_mode == 3) TRIS_FAST(BM_, BN_, EPI_SLAB, 3);                                 \
      else TRIS_FAST(BM_, BN_, EPI_SLAB, 0);                                                       \
    } else {                                                                                       \
      if (g_gemm_mode == 1) TRIS_FAST(BM_, BN_, EPI_STD, 1);                                       \
      else if (g_gemm_mode == 2) TRIS_FAST(BM_, BN_, EPI_STD, 2);                                  \
      else if (g_gemm_mode == 3) TRIS_FAST(BM_, BN_, EPI_STD, 3);                                  \
      else TRIS_FAST(BM_, BN_, EPI_STD, 0);                                                        \
    }                                                                                              \
  } while (0)
#define TRIS_PIPE_GO(BM_, BN_, NW_, NWM_)                                                                                  \
  do {                                                                                                                     \
    if (splitk > 1) {                                                                                                      \
      p.C = ws;                                                                                                            \
      hipLaunchKernelGGL((gemm_fast_kernel<BM_, BN_, AK, (BKIND == B_NK_PRE ? B_NK : BKIND), EPI_SLAB, 1, NW_, 16, 2, NWM_>), grid, \
                         dim3(NW_ * 64), 0, st, p);                                                                        \
    } else {                                                                                                               \
      hipLaunchKernelGGL((gemm_fast_kernel<BM_, BN_, AK, (BKIND == B_NK_PRE ? B_NK : BKIND), EPI_STD, 1, NW_, 16, 2, NWM_>), grid,  \
                         dim3(NW_ * 64), 0, st, p);                                                                        \
    }                                                                                                                      \
  } while (0)
  if (pipe) {
    if (bm == 256 && bn == 128) TRIS_PIPE_GO(256, 128, 8, 4);
    else if (bm == 128 && bn == 128) TRIS_PIPE_GO(128, 128, 8, 2);
    else if (bm == 128 && bn == 64) TRIS_PIPE_GO(128, 64, 4, 2);
    else TRIS_PIPE_GO(64, 64, 4, 2);
  } else
  if (bm == 128 && bn == 128) TRIS_GO(128, 128);
  else if (bm == 128 && bn == 64) TRIS_GO(128, 64);
  else if (bm == 128 && bn == 32) TRIS_GO_FAST_ONLY(128, 32);
  else TRIS_GO(64, 64);
#undef TRIS_GO
#undef TRIS_PIPE_GO
#undef TRIS_GO_FAST_ONLY
#undef TRIS_FAST
  TRIS_LAUNCH_CHECK();
  if (splitk > 1) {
    p.C = Cfinal;
    long total = (long)p.M * p.N;
    static const bool vec_ok = !(getenv("TRIS_REDUCE_VEC") && getenv("TRIS_REDUCE_VEC")[0] == '0');  // developer A/B knob
    if (vec_ok && (p.N & 3) == 0 && al16(ws) && al16(p.C))
      hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3(cdiv(total / 4, 256)), dim3(256), 0, st, ws, splitk, p);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(cdiv(total, 256)), dim3(256), 0, st, ws, splitk, p);
    TRIS_LAUNCH_CHECK();
  }
  return 0;
}

// ---- autotuner ------------------------------------------------------------------------------------------------------------
// The first time a (kind, M, N, K, batch, mode) product is seen, every admissible (tile, split-K) pair is launched on the
// caller's buffers (the kernels are idempotent), timed with HIP events, and the fastest is cached for the life of the
// process.  Host-synchronising, so only outside stream capture; TRIS_AUTOTUNE=0 keeps the cost model.
struct TuneKey {
  int ak, bk, M, N, K, batch, mode;
  bool operator<(const TuneKey& o) const {
    return std::tie(ak, bk, M, N, K, batch, mode) < std::tie(o.ak, o.bk, o.M, o.N, o.K, o.batch, o.mode);
  }
};
static std::map<TuneKey, Cfg> g_tuned;
static std::mutex g_tune_mu;

static int g_autotune = -1;  // -1: read TRIS_AUTOTUNE on first use
static bool autotune_enabled() {
  if (g_autotune < 0) { const char* e = getenv("TRIS_AUTOTUNE"); g_autotune = (e && e[0] == '0') ? 0 : 1; }
  return g_autotune == 1;
}

template <int AK, int BKIND>
int launch_cfg(GemmParams& p, int batch, float* ws, long ws_bytes, hipStream_t st) {
  if (BKIND == B_NK_PRE && !(g_gemm_mode == 1 && p.fastA && p.fastB && p.K % 32 == 0 && p.M >= 4 && p.N >= 4 && batch == 1))
    return TRIS_WP_UNSUPPORTED;  // pre-split operands exist only for the fast x3 kernel: the caller falls back to fp32 B
  Cfg h = heuristic_cfg(p, batch, ws, ws_bytes);
  if (!autotune_enabled() || getenv("TRIS_FORCE_TILE")) return run_cfg<AK, BKIND>(p, batch, ws, st, h);
  const bool stat = p.stat_part != nullptr;  // fused BN statistics: 128-row tiles and no split-K are fixed, the tile width is tuned
  const TuneKey key = {AK, BKIND + (stat ? 16 : 0) + (p.bnb_x != nullptr ? 32 : 0), p.M, p.N, p.K, batch, g_gemm_mode};
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tuned.find(key);
    if (it != g_tuned.end()) return run_cfg<AK, BKIND>(p, batch, ws, st, it->second);
  }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;   // tuning synchronises the device: never under stream capture
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
    return run_cfg<AK, BKIND>(p, batch, ws, st, h);          // (a shape first met during capture runs the cost model's choice)
  // candidates
  static const int tiles[5][2] = {{128, 128}, {128, 64}, {64, 64}, {128, 32}, {256, 128}};
  const bool fastk = p.fastA && p.fastB && (p.K % 32 == 0) && p.M >= 4 && p.N >= 4;
  static const int sks[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128};
  const bool can_split = (batch == 1 && ws != nullptr && p.K >= 512) && !stat;
  const bool can_pipe = pipe_ok(p) && BKIND != B_NK_PRE && forced_pipe() != 0;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return run_cfg<AK, BKIND>(p, batch, ws, st, h);
  (void)hipDeviceSynchronize();  // drain the other streams: candidates are timed on an otherwise idle device
  Cfg best = h;
  float best_ms = 1e30f;
  auto time_cfg = [&](Cfg c) -> float {
    float ms_min = 1e30f;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0, st);
      if (run_cfg<AK, BKIND>(p, batch, ws, st, c) != 0) return 1e30f;
      (void)hipEventRecord(e1, st);
      if (hipEventSynchronize(e1) != hipSuccess) return 1e30f;
      float ms = 1e30f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      ms_min = ms < ms_min ? ms : ms_min;
    }
    return ms_min;
  };
  for (int t = 0; t < 5; ++t) {
    const int cbm = tiles[t][0], cbn = tiles[t][1];
    if (t != 2 && p.M < 96) continue;
    if ((t == 0 || t == 4) && p.N <= 64) continue;
    if (t == 3 && !(p.N <= 32 && fastk)) continue;
    if (t == 4 && (!can_pipe || p.M < 256)) continue;   // the 256-row tile exists in the pipelined form only
    if (stat && cbm != 128) continue;
    const long ntiles = (long)cdiv(p.M, cbm) * cdiv(p.N, cbn) * batch;
    for (int sk : sks) {
      if (sk > 1 && (!can_split || sk > p.K / 256 || (long)sk * p.M * p.N * (long)sizeof(float) > ws_bytes)) break;
      if (sk > 1 && ntiles * sk > 1536) break;  // more than ~6 blocks per CU buys nothing
      if (sk * 8 < h.splitk && ntiles * sk < 256) continue;  // a handful of blocks walking a huge K serially: not worth timing
      for (int pipe = (can_pipe && t != 3) ? 1 : 0; pipe >= (t == 4 || forced_pipe() == 1 ? (can_pipe && t != 3 ? 1 : 0) : 0); --pipe) {
        const Cfg c = {cbm, cbn, sk, g_x3_waves, pipe};
        const float ms = time_cfg(c);
        if (ms < best_ms) { best_ms = ms; best = c; }
      }
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tuned[key] = best;
    if (const char* lg = getenv("TRIS_TUNE_LOG")) {  // developer knob: one line per tuned shape (idle-device time of the winner)
      if (FILE* f = fopen(lg, "a")) {
        fprintf(f, "ak=%d bkind=%d M=%d N=%d K=%d batch=%d mode=%d -> %dx%d sk=%d nw=%d pipe=%d  %.1f us  %.1f TFLOP/s\n", AK, key.bk,
                p.M, p.N, p.K, batch, g_gemm_mode, best.bm, best.bn, best.splitk, best.nw, best.pipe, best_ms * 1e3f,
                2.0 * p.M * p.N * p.K * batch / (best_ms * 1e-3) * 1e-12);
        fclose(f);
      }
    }
  }
  return run_cfg<AK, BKIND>(p, batch, ws, st, best);  // leave the outputs of the chosen configuration
}





// ---- direct 3x3 convolution: configurations of the A_HALO kernels (gemm_fast.h) and the choice between them and the
// implicit GEMM ---------------------------------------------------------------------------------------------------------------
static long g_direct_launches[2] = {0, 0};   // diagnostics: launches of the direct convolution / direct weight-gradient kernels
#include "conv_direct_cfg.h"

// launchers in conv_direct.hip (hidden symbols of the same shared object)
}  // namespace
extern "C" __attribute__((visibility("hidden"))) int tris_internal_run_halo(const void* params, int id, int dgrad, void* stream);
extern "C" __attribute__((visibility("hidden"))) int tris_internal_run_wgrad_direct(int id, const float* X, const float* dY, float* dW, int B,
    int H, int W, int Ci, int Co, float* ws, long ws_bytes, void* stream, const float* mean, const float* invstd, const float* gamma,
    const float* beta);
extern "C" __attribute__((visibility("hidden"))) int tris_internal_stem_conv1(const float* X, const float* Wt, float* Y, int B, int H, int W,
    int Cin, int Cout, int stride, double* stat_part, void* stream);
namespace {
template <int BKIND>
int run_halo(const GemmParams& p, int id, hipStream_t st) {
  const int rc = tris_internal_run_halo(&p, id, BKIND == B_KN_DGRAD ? 1 : 0, st);
  if (rc == 0) ++g_direct_launches[0];
  return rc;
}
static int run_wgrad_direct(int id, const float* X, const float* dY, float* dW, int B, int H, int W, int Ci, int Co, float* ws,
                            long ws_bytes, hipStream_t st, BnIn bn = BnIn{nullptr, nullptr, nullptr, nullptr}) {
  const int rc = tris_internal_run_wgrad_direct(id, X, dY, dW, B, H, W, Ci, Co, ws, ws_bytes, st, bn.mean, bn.invstd, bn.gamma, bn.beta);
  if (rc == 0) ++g_direct_launches[1];
  return rc;
}
static int run_stem_conv1(const float* X, const float* Wt, float* Y, int B, int H, int W, int Cin, int Cout, int stride,
                          double* stat_part, hipStream_t st) {
  return tris_internal_stem_conv1(X, Wt, Y, B, H, W, Cin, Cout, stride, stat_part, st);
}

static bool halo_shape_ok(const GemmParams& p) {
  return g_gemm_mode == 1 && p.gStride == 1 && p.gC % 16 == 0 && p.fastB && al16(p.A) && al16(p.B) && p.N % 4 == 0 &&
         p.gHo == p.gH && p.gWo == p.gW && (long)p.gB * p.gH * p.gW * p.gC < (1L << 31);
}
// static choice (no autotuning, or under stream capture), from the measured table in DESIGN.md: the 2-D patch kernels win
// wherever they apply (W, H multiples of 16: the stem and the 80 x 80 stages); the flattened-window kernels pay only for the
// data gradients of the 40 x 40 / 20 x 20 stages from 256 channels up.  0 = implicit GEMM.
static int halo_static_choice(const GemmParams& p, bool dgrad) {
  if (p.N <= 32 && halo_ok(p, 3)) return 3;
  if (p.N <= 64 && halo_ok(p, 2)) return 2;
  if (halo_ok(p, 6)) return 6;
  if (halo_ok(p, 1)) return 1;
  if (dgrad && p.N >= 256 && p.gW >= 20 && halo_ok(p, 4)) return 4;
  return 0;
}

// direct_only: the caller needs a direct kernel (p.in_mean: BatchNorm + ReLU folded into the window staging) and has checked
// with tris_conv3x3_bnin_ok that one applies: the implicit GEMM is neither timed nor chosen.
template <int BKIND>
int conv3_dispatch(GemmParams& p, hipStream_t st, int* stat_rows, bool direct_only = false) {
  const bool stat = p.stat_part != nullptr;
  const int first = halo_shape_ok(p) ? halo_static_choice(p, BKIND == B_KN_DGRAD) : 0;
  auto direct = [&](int id) {
    if (stat_rows) *stat_rows = stat ? halo_tiles_m(p, id) : 0;
    return run_halo<BKIND>(p, id, st);
  };
  auto im2col = [&]() {
    if (direct_only) return first ? direct(first) : (int)hipErrorInvalidValue;   // (stands in for "the default" below)
    if (stat_rows) *stat_rows = stat ? cdiv(p.M, 128) : 0;
    return launch_cfg<A_IM2COL, BKIND>(p, 1, nullptr, 0, st);
  };
  const char* e = getenv("TRIS_CONV_DIRECT");   // read per call: tests switch it at run time
  const int forced = e ? atoi(e) : -1;
  if (forced == 0 || !halo_shape_ok(p)) return im2col();
  if (forced > 0) return (forced < kHaloN && halo_ok(p, forced)) ? direct(forced) : im2col();
  bool any = false;
  for (int id = 1; id < kHaloN; ++id) any = any || halo_ok(p, id);
  if (!any) return im2col();
  const TuneKey key = {A_HALO, BKIND + (stat ? 16 : 0) + (direct_only ? 32 : 0) + (p.bnb_x != nullptr ? 64 : 0), p.M, p.N, p.K,
                       p.gH * 4096 + p.gW, g_gemm_mode};
  int cached = -1;
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);   // (released before the launch: the implicit GEMM looks up its own tile under it)
    auto it = g_tuned.find(key);
    if (it != g_tuned.end()) cached = it->second.bm;
  }
  if (cached >= 0) return cached ? direct(cached) : im2col();
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (!autotune_enabled() || hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
    return first ? direct(first) : im2col();
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return first ? direct(first) : im2col();
  int rc = im2col();   // (tunes the implicit GEMM's own tile on first sight)
  if (rc != 0) return rc;
  (void)hipDeviceSynchronize();
  auto timed = [&](int id) -> float {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0, st);
      if ((id ? direct(id) : im2col()) != 0) return 1e30f;
      (void)hipEventRecord(e1, st);
      if (hipEventSynchronize(e1) != hipSuccess) return 1e30f;
      float ms = 1e30f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    return best;
  };
  int best_id = 0;
  const float t_im2col = timed(0);
  float best_ms = t_im2col;
  for (int id = 1; id < kHaloN; ++id) {
    if (!halo_ok(p, id)) continue;
    const float ms = timed(id);
    if (ms < best_ms) { best_ms = ms; best_id = id; }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tuned[key] = Cfg{best_id, 0, 1, 0, 0};
    if (const char* lg = getenv("TRIS_TUNE_LOG")) {
      if (FILE* f = fopen(lg, "a")) {
        fprintf(f, "conv3x3 bkind=%d M=%d N=%d K=%d HxW=%dx%d -> %s %d  %.1f us  %.1f TFLOP/s  (implicit GEMM %.1f us)\n", key.bk, p.M, p.N,
                p.K, p.gH, p.gW, best_id ? "direct" : "implicit", best_id, best_ms * 1e3f,
                2.0 * p.M * p.N * p.K / (best_ms * 1e-3) * 1e-12, t_im2col * 1e3f);
        fclose(f);
      }
    }
  }
  return best_id ? direct(best_id) : im2col();   // leave the outputs (and *stat_rows) of the chosen kernel
}


}  // namespace

// may the fused-statistics epilogue of the fast kernel serve this product?
static bool stats_eligible(const GemmParams& p) {
  return p.fastA && p.fastB && (p.K % 32 == 0) && p.M >= 128 && p.N >= 4;
}

extern "C" int tris_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, long lda, long ldb,
                             long ldc, int transA, int transB, int batch, long sA, long sB, long sC,
                             const float* bias, int bias_mode, const float* resid, long ldr, long sR, int act,
                             float alpha, float* workspace, long ws_bytes, void* stream) {
  const H2Next h2n = h2_take();
  if (M <= 0 || N <= 0 || batch <= 0) return 0;
  if (K <= 0) return (int)hipErrorInvalidValue;
  GemmParams p = {};
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.sA = sA; p.sB = sB; p.sC = sC;
  p.bias = bias; p.bias_mode = bias ? bias_mode : 0;
  p.resid = resid; p.ldr = ldr; p.sR = sR; p.act = act; p.alpha = alpha;
  p.vecA = al16(A) && (lda % 4 == 0) && (sA % 4 == 0);
  p.vecB = al16(B) && (ldb % 4 == 0) && (sB % 4 == 0);
  p.fastA = p.vecA && (!transA || M % 4 == 0);
  p.fastB = p.vecB && (transB || N % 4 == 0);
  hipStream_t st = (hipStream_t)stream;
  H2Guard h2(p, h2n);
  if (!transA && transB) return launch_cfg<A_ROWK, B_NK>(p, batch, workspace, ws_bytes, st);
  if (!transA && !transB) return launch_cfg<A_ROWK, B_KN>(p, batch, workspace, ws_bytes, st);
  if (transA && !transB) return launch_cfg<A_COLK, B_KN>(p, batch, workspace, ws_bytes, st);
  return launch_cfg<A_COLK, B_NK>(p, batch, workspace, ws_bytes, st);
}

// Y[B,Ho,Wo,Cout] = conv3x3(X[B,H,W,Cin], Wt[Cout][3][3][Cin]), pad 1, stride 1|2, optional fused ReLU-less epilogue.
extern "C" int tris_conv3x3_fwd_f32(const float* X, const float* Wt, float* Y, int B, int H, int W, int Cin, int Cout,
                                    int stride, void* stream) {
  const H2Next h2n = h2_take();
  int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  GemmParams p = {};
  p.A = X; p.B = Wt; p.C = Y;
  p.M = B * Ho * Wo; p.N = Cout; p.K = 9 * Cin;
  p.ldb = 9L * Cin; p.ldc = Cout; p.alpha = 1.f;
  p.gH = H; p.gW = W; p.gC = Cin; p.gHo = Ho; p.gWo = Wo; p.gStride = stride; p.gB = B;
  p.vecA = al16(X) && (Cin % 16 == 0);
  p.vecB = al16(Wt) && ((9 * Cin) % 4 == 0);
  p.fastA = al16(X) && (Cin % 32 == 0);
  p.fastB = p.vecB;
  if (const int rows = run_stem_conv1(X, Wt, Y, B, H, W, Cin, Cout, stride, nullptr, (hipStream_t)stream))
    return rows > 0 ? 0 : (int)hipErrorLaunchFailure;
  H2Guard h2(p, h2n);   // (armed: the implicit GEMM in h2 -- the direct kernels exist in x3 only)
  return conv3_dispatch<B_NK>(p, (hipStream_t)stream, nullptr);
}

// dX[B,H,W,Cin] = conv3x3_transpose(dY[B,H,W,Cout], Wt), stride 1 only: a 3x3 conv of dY with the taps mirrored.
extern "C" int tris_conv3x3_dgrad_f32(const float* dY, const float* Wt, float* dX, int B, int H, int W, int Cin,
                                      int Cout, void* stream) {
  const H2Next h2n = h2_take();
  GemmParams p = {};
  p.A = dY; p.B = Wt; p.C = dX;
  p.M = B * H * W; p.N = Cin; p.K = 9 * Cout;
  p.ldc = Cin; p.alpha = 1.f;
  p.gH = H; p.gW = W; p.gC = Cout; p.gHo = H; p.gWo = W; p.gStride = 1; p.gB = B;
  p.wCin = Cin; p.wCout = Cout;
  p.vecA = al16(dY) && (Cout % 16 == 0);
  p.vecB = al16(Wt) && (Cin % 4 == 0);
  p.fastA = al16(dY) && (Cout % 32 == 0);
  p.fastB = p.vecB;
  if (Cout % 16 != 0) return (int)hipErrorInvalidValue;  // k tile must not straddle taps for the B loader
  H2Guard h2(p, h2n);
  return conv3_dispatch<B_KN_DGRAD>(p, (hipStream_t)stream, nullptr);
}

// Data gradient of a 3x3 convolution whose INPUT is the output of a train-mode BatchNorm + ReLU (no residual): the reduction
// pass of that BatchNorm's backward rides in the epilogue, as in tris_gemm_bnbwd_f32 (mask recomputed from bn_x / gamma / beta).
// dZ[B,H,W,Cin] <- masked gradient; part <- [*part_rows][2][Cin] fp64 partial; *part_rows = 0: not eligible, nothing launched.
extern "C" int tris_conv3x3_dgrad_bnbwd_f32(const float* dY, const float* Wt, float* dZ, int B, int H, int W, int Cin, int Cout,
                                            const float* bn_x, const float* mean, const float* invstd, const float* gamma,
                                            const float* beta, double* part, int* part_rows, void* stream) {
  const H2Next h2n = h2_take();
  *part_rows = 0;
  if (bn_x == nullptr || part == nullptr || gamma == nullptr || beta == nullptr) return (int)hipErrorInvalidValue;
  GemmParams p = {};
  p.A = dY; p.B = Wt; p.C = dZ;
  p.M = B * H * W; p.N = Cin; p.K = 9 * Cout;
  p.ldc = Cin; p.alpha = 1.f;
  p.gH = H; p.gW = W; p.gC = Cout; p.gHo = H; p.gWo = W; p.gStride = 1; p.gB = B;
  p.wCin = Cin; p.wCout = Cout;
  p.vecA = al16(dY) && (Cout % 16 == 0);
  p.vecB = al16(Wt) && (Cin % 4 == 0);
  p.fastA = al16(dY) && (Cout % 32 == 0);
  p.fastB = p.vecB;
  if (Cout % 16 != 0) return (int)hipErrorInvalidValue;
  const bool al = al16(dZ) && al16(bn_x) && al16(mean) && al16(invstd) && al16(gamma) && al16(beta) && Cin % 4 == 0;
  if (!stats_eligible(p) || !al) return 0;
  p.stat_part = part;
  p.bnb_x = bn_x; p.bnb_mean = mean; p.bnb_invstd = invstd; p.bnb_gamma = gamma; p.bnb_beta = beta;
  H2Guard h2(p, h2n);
  return conv3_dispatch<B_KN_DGRAD>(p, (hipStream_t)stream, part_rows);
}

// ---- weight gradient of the stem's first convolution (Cin = 3: the 27-wide "N" does not fit the tiled kernels) -----------
// dW[co][tap][ci] = sum over output pixels of dY[p][co] * X[in(p, tap)][ci] is ONE 32 x 32 output tile with a reduction
// over ~10^6 pixels.  Each wave walks a contiguous pixel range with v_mfma_f32_32x32x2_f32 (2 pixels per step: lane
// (m = co, kh) reads dY[p + kh][co] -- one 128-byte line per pixel --, lane (n = (tap, ci), kh) gathers its input value
// with incrementally maintained coordinates, no divisions in the loop); the block's four tiles meet in LDS, per-block
// partials are summed in fixed order by a second launch.  HBM-bound on the dY stream.
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                         float* __restrict__ part, int Bn, int H, int W, int Cin, int Cout,
                                                         int Ho, int Wo, int stride, long pix_per_wave) {
  __shared__ float red[3][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mn = lane & 31, kh = lane >> 5;
  const long total = (long)Bn * Ho * Wo;
  const long p_beg = ((long)blockIdx.x * 4 + wave) * pix_per_wave;
  const long p_end = min(total, p_beg + pix_per_wave);
  const int tap = mn / Cin, ci = mn - tap * Cin;
  const bool nv = mn < 9 * Cin;
  const int ky = tap / 3, kx = tap - ky * 3;
  const bool mv = mn < Cout;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // this lane's pixel (p_beg + kh) in (b, oy, ox) form; advanced by 2 per step
  long p = p_beg + kh;
  int b = (int)(p / ((long)Ho * Wo));
  int rem = (int)(p - (long)b * Ho * Wo);
  int oy = rem / Wo, ox = rem - oy * Wo;
  for (; p - kh < p_end; p += 16) {  // 8 MFMA steps (16 pixels) per trip: all 16 loads are in flight before the first MFMA
    float a[8], v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long pu = p + 2 * u;
      a[u] = 0.f;
      v[u] = 0.f;
      if (pu < p_end) {
        if (mv) a[u] = dY[pu * Cout + mn];
        const int iy = oy * stride - 1 + ky, ix = ox * stride - 1 + kx;
        if (nv && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v[u] = X[(((long)b * H + iy) * W + ix) * Cin + ci];
      }
      ox += 2;
      while (ox >= Wo) { ox -= Wo; if (++oy == Ho) { oy = 0; ++b; } }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], v[u], acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
    float* o = part + (long)blockIdx.x * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) {  // D[row = (r&3) + 8*(r>>2) + 4*kh][col = mn]
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
      o[row * 32 + mn] = acc[r] + red[0][r][lane] + red[1][r][lane] + red[2][r][lane];
    }
  }
}

// grid 32 blocks: block j sums the partials of outputs [32 j, 32 j + 32) -- 32 outputs x 8 slices of the partial list,
// fixed summation order (deterministic)
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ part, int nblk, float* __restrict__ dW,
                                                                int Cout, int N) {
  __shared__ float sh[8][32];
  const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + o;
  float s = 0.f;
  for (int bk = sl; bk < nblk; bk += 8) s += part[(long)bk * 1024 + idx];
  sh[sl][o] = s;
  __syncthreads();
  if (sl == 0) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += sh[q][o];
    const int row = idx >> 5, col = idx & 31;
    if (row < Cout && col < N) dW[row * N + col] = t;
  }
}

// dW[Cout][3][3][Cin] = sum over output pixels of dY (x) gathered X.  Split-K over pixels through `workspace`.
extern "C" int tris_conv3x3_wgrad_f32(const float* X, const float* dY, float* dW, int B, int H, int W, int Cin,
                                      int Cout, int stride, float* workspace, long ws_bytes, void* stream) {
  const H2Next h2n = h2_take();
  int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  if (9 * Cin <= 32 && Cout <= 32 && workspace != nullptr) {  // the stem's first conv: dedicated single-tile reduction
    const long total = (long)B * Ho * Wo;
    int nblk = (int)std::min<long>(512, std::max<long>(1, total / 512));
    long ppw = (cdiv(total, (long)nblk * 4) + 15) / 16 * 16;  // multiple of 16: whole 8-step trips, pixel pairs stay in one range
    nblk = cdiv(total, ppw * 4);
    if ((long)nblk * 1024 * (long)sizeof(float) <= ws_bytes) {
      hipStream_t st = (hipStream_t)stream;
      hipLaunchKernelGGL(stem_wgrad_kernel, dim3(nblk), dim3(256), 0, st, X, dY, workspace, B, H, W, Cin, Cout, Ho, Wo, stride,
                         ppw);
      TRIS_LAUNCH_CHECK();
      hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(32), dim3(256), 0, st, workspace, nblk, dW, Cout, 9 * Cin);
      TRIS_LAUNCH_CHECK();
      return 0;
    }
  }
  GemmParams p = {};
  p.A = dY; p.B = X; p.C = dW;
  p.M = Cout; p.N = 9 * Cin; p.K = B * Ho * Wo;
  p.lda = Cout; p.ldc = 9L * Cin; p.alpha = 1.f;
  p.gH = H; p.gW = W; p.gC = Cin; p.gHo = Ho; p.gWo = Wo; p.gStride = stride; p.gB = B;
  p.vecA = al16(dY) && (Cout % 4 == 0);
  p.vecB = al16(X) && (Cin % 4 == 0);
  p.fastA = p.vecA;
  p.fastB = p.vecB;
  hipStream_t st = (hipStream_t)stream;
  H2Guard h2(p, h2n);   // (armed: A = dY, B = X; the implicit GEMM in h2)
  auto gemm = [&]() { return launch_cfg<A_COLK, B_KN_IM2COL>(p, 1, workspace, ws_bytes, st); };
  // direct kernel (wgrad3x3_direct_kernel) or the implicit GEMM: timed once per shape.  TRIS_WGRAD_DIRECT=0 keeps the GEMM,
  // =1..5 forces a direct configuration where it applies (tests).
  const char* e = getenv("TRIS_WGRAD_DIRECT");
  const int forced = e ? atoi(e) : -1;
  const bool shape_ok = g_gemm_mode == 1 && stride == 1 && workspace != nullptr && al16(X) && al16(dY) && al16(dW) && al16(workspace) &&
                        (long)B * H * W * std::max(Cin, Cout) < (1L << 31);
  auto direct = [&](int id) { return run_wgrad_direct(id, X, dY, dW, B, H, W, Cin, Cout, workspace, ws_bytes, st); };
  auto usable = [&](int id) { return wg_ok(id, H, W, Cin, Cout) && wg_slices(id, B, H, W, Cin, Cout, ws_bytes) >= 1; };
  if (forced == 0 || !shape_ok) return gemm();
  if (forced > 0) return (forced < kWgN && usable(forced)) ? direct(forced) : gemm();
  int first = 0;   // static choice: the measured winners; the autotuner times every usable configuration
  if (usable(1)) first = 1;
  else if (usable(2)) first = 2;
  else if (usable(5)) first = 5;
  if (!first) return gemm();
  const TuneKey key = {A_HALO, 64 + B_KN_IM2COL, p.M, p.N, p.K, H * 4096 + W, g_gemm_mode};
  int cached = -1;
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tuned.find(key);
    if (it != g_tuned.end()) cached = it->second.bm;
  }
  if (cached >= 0) return cached ? direct(cached) : gemm();
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (!autotune_enabled() || hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return direct(first);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return direct(first);
  int rc = gemm();   // (tunes the GEMM's own tile / split-K on first sight)
  if (rc != 0) return rc;
  (void)hipDeviceSynchronize();
  auto timed = [&](int id) -> float {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0, st);
      if ((id ? direct(id) : gemm()) != 0) return 1e30f;
      (void)hipEventRecord(e1, st);
      if (hipEventSynchronize(e1) != hipSuccess) return 1e30f;
      float ms = 1e30f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    return best;
  };
  const float t_gemm = timed(0);
  float best_ms = t_gemm;
  int best_id = 0;
  for (int id = 1; id < kWgN; ++id) {
    if (!usable(id)) continue;
    const float ms = timed(id);
    if (ms < best_ms) { best_ms = ms; best_id = id; }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tuned[key] = Cfg{best_id, 0, 1, 0, 0};
    if (const char* lg = getenv("TRIS_TUNE_LOG")) {
      if (FILE* f = fopen(lg, "a")) {
        fprintf(f, "wgrad3x3 Cout=%d Cin=%d pixels=%d HxW=%dx%d -> %s %d  %.1f us  %.1f TFLOP/s  (implicit GEMM %.1f us)\n", Cout, Cin, p.K, H,
                W, best_id ? "direct" : "implicit", best_id, best_ms * 1e3f, 2.0 * p.M * p.N * p.K / (best_ms * 1e-3) * 1e-12,
                t_gemm * 1e3f);
        fclose(f);
      }
    }
  }
  return best_id ? direct(best_id) : gemm();
}

// Conv / 1x1-conv (GEMM) forward with the BatchNorm batch statistics of the OUTPUT fused into the epilogue.
// stat_part receives [stat_rows][2][N] fp64 partial (sum, sum of squares); *stat_rows (host) = number of partial rows,
// or 0 when the shape is not eligible for the fused path (then the caller runs tris_bn_stats_f32 as usual).

extern "C" int tris_gemm_bnstat_f32(const float* A, const float* B, float* C, int M, int N, int K, double* stat_part,
                                    int* stat_rows, void* stream) {
  const H2Next h2n = h2_take();
  GemmParams p = {};
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
  p.lda = K; p.ldb = K; p.ldc = N; p.alpha = 1.f;
  p.vecA = al16(A) && (K % 4 == 0);
  p.vecB = al16(B) && (K % 4 == 0);
  p.fastA = p.vecA;
  p.fastB = p.vecB;
  p.stat_part = stats_eligible(p) ? stat_part : nullptr;
  *stat_rows = p.stat_part ? cdiv(M, 128) : 0;
  H2Guard h2(p, h2n);
  return launch_cfg<A_ROWK, B_NK>(p, 1, nullptr, 0, (hipStream_t)stream);
}

// Data gradient of a 1x1 convolution / Linear, dZ[M,N] = mask(dY[M,K] . W[K,N] (+ resid)), with the reduction pass of the
// BatchNorm(+ReLU) BACKWARD that consumes it fused into the epilogue (GemmParams::bnb_*): bn_x [M,N] is that BatchNorm's raw
// input, bn_y its output (residual form: the mask is y > 0) or NULL (the mask is recomputed from bn_x, gamma, beta).  part receives
// [*part_rows][2][N] fp64 partial (sum dz, sum dz * xhat) -- finish with tris_part_finalize_f32; *part_rows = 0: the shape is not
// eligible (nothing was launched; the caller runs tris_gemm_f32 + tris_bn_bwd_reduce_f32 as usual).
extern "C" int tris_gemm_bnbwd_f32(const float* dY, const float* Wt, float* dZ, int M, int N, int K, const float* resid,
                                   long ldr, const float* bn_x, const float* bn_y, const float* mean, const float* invstd,
                                   const float* gamma, const float* beta, double* part, int* part_rows, void* stream) {
  const H2Next h2n = h2_take();
  *part_rows = 0;
  if (M <= 0 || N <= 0 || K <= 0 || bn_x == nullptr || part == nullptr) return (int)hipErrorInvalidValue;
  if (bn_y == nullptr && (gamma == nullptr || beta == nullptr)) return (int)hipErrorInvalidValue;
  GemmParams p = {};
  p.A = dY; p.B = Wt; p.C = dZ; p.M = M; p.N = N; p.K = K;
  p.lda = K; p.ldb = N; p.ldc = N; p.alpha = 1.f;
  p.resid = resid; p.ldr = ldr;
  p.vecA = al16(dY) && (K % 4 == 0);
  p.vecB = al16(Wt) && (N % 4 == 0);
  p.fastA = p.vecA;
  p.fastB = p.vecB;
  const bool al = al16(dZ) && al16(bn_x) && (bn_y == nullptr || al16(bn_y)) && (resid == nullptr || (al16(resid) && ldr % 4 == 0)) &&
                  al16(mean) && al16(invstd) && (gamma == nullptr || al16(gamma)) && (beta == nullptr || al16(beta));
  if (!stats_eligible(p) || !al || N % 4 != 0) return 0;
  p.stat_part = part;
  p.bnb_x = bn_x; p.bnb_y = bn_y; p.bnb_mean = mean; p.bnb_invstd = invstd; p.bnb_gamma = gamma; p.bnb_beta = beta;
  *part_rows = cdiv(M, 128);
  H2Guard h2(p, h2n);
  return launch_cfg<A_ROWK, B_KN>(p, 1, nullptr, 0, (hipStream_t)stream);
}

extern "C" int tris_conv3x3_fwd_bnstat_f32(const float* X, const float* Wt, float* Y, int B, int H, int W, int Cin,
                                           int Cout, int stride, double* stat_part, int* stat_rows, void* stream) {
  const H2Next h2n = h2_take();
  int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  GemmParams p = {};
  p.A = X; p.B = Wt; p.C = Y;
  p.M = B * Ho * Wo; p.N = Cout; p.K = 9 * Cin;
  p.ldb = 9L * Cin; p.ldc = Cout; p.alpha = 1.f;
  p.gH = H; p.gW = W; p.gC = Cin; p.gHo = Ho; p.gWo = Wo; p.gStride = stride; p.gB = B;
  p.vecA = al16(X) && (Cin % 16 == 0);
  p.vecB = al16(Wt) && ((9 * Cin) % 4 == 0);
  p.fastA = al16(X) && (Cin % 32 == 0);
  p.fastB = p.vecB;
  if (const int rows = run_stem_conv1(X, Wt, Y, B, H, W, Cin, Cout, stride, p.M >= 128 ? stat_part : nullptr, (hipStream_t)stream)) {
    *stat_rows = (rows > 0 && p.M >= 128 && stat_part != nullptr) ? rows : 0;
    return rows > 0 ? 0 : (int)hipErrorLaunchFailure;
  }
  p.stat_part = stats_eligible(p) ? stat_part : nullptr;
  H2Guard h2(p, h2n);
  return conv3_dispatch<B_NK>(p, (hipStream_t)stream, stat_rows);
}

// ---- BatchNorm + ReLU folded into the consuming 3x3 convolution ------------------------------------------------------------
// conv3x3(relu(bn(X))) without relu(bn(X)) ever existing in HBM: the direct kernels apply the normalisation while they stage
// their window (forward: gemm_fast.h A_HALO; weight gradient: wgrad3x3_direct_kernel).  Only where a direct kernel serves BOTH
// products -- tris_conv3x3_bnin_ok says so -- otherwise the caller materialises relu(bn(X)) with tris_bn_apply_f32 as before.
static bool bnin_enabled() {
  const char *a = getenv("TRIS_CONV_DIRECT"), *b = getenv("TRIS_WGRAD_DIRECT"), *c = getenv("TRIS_BN_FOLD");
  return g_gemm_mode == 1 && !(a && atoi(a) == 0) && !(b && atoi(b) == 0) && !(c && c[0] == '0');
}
static GemmParams conv3_fwd_params(const float* X, const float* Wt, float* Y, int B, int H, int W, int Cin, int Cout) {
  GemmParams p = {};
  p.A = X; p.B = Wt; p.C = Y;
  p.M = B * H * W; p.N = Cout; p.K = 9 * Cin;
  p.ldb = 9L * Cin; p.ldc = Cout; p.alpha = 1.f;
  p.gH = H; p.gW = W; p.gC = Cin; p.gHo = H; p.gWo = W; p.gStride = 1; p.gB = B;
  p.vecA = al16(X) && (Cin % 16 == 0);
  p.vecB = al16(Wt) && ((9 * Cin) % 4 == 0);
  p.fastA = al16(X) && (Cin % 32 == 0);
  p.fastB = p.vecB;
  return p;
}
static int wg_static_choice(int B, int H, int W, int Cin, int Cout, long ws_bytes) {
  for (int id : {1, 2, 5})
    if (wg_ok(id, H, W, Cin, Cout) && wg_slices(id, B, H, W, Cin, Cout, ws_bytes) >= 1) return id;
  return 0;
}

extern "C" int tris_conv3x3_bnin_ok(int B, int H, int W, int Cin, int Cout) {
  if (!bnin_enabled() || Cin % 16 != 0 || Cout % 4 != 0) return 0;
  GemmParams p = conv3_fwd_params(reinterpret_cast<const float*>(16), reinterpret_cast<const float*>(16),
                                  reinterpret_cast<float*>(16), B, H, W, Cin, Cout);   // (shape test only: aligned dummies)
  return halo_shape_ok(p) && halo_static_choice(p, false) > 0 && wg_static_choice(B, H, W, Cin, Cout, 64L << 20) > 0;
}

extern "C" int tris_conv3x3_fwd_bnin_f32(const float* X, const float* mean, const float* invstd, const float* gamma,
                                         const float* beta, const float* Wt, float* Y, int B, int H, int W, int Cin, int Cout,
                                         double* stat_part, int* stat_rows, void* stream) {
  GemmParams p = conv3_fwd_params(X, Wt, Y, B, H, W, Cin, Cout);
  if (!bnin_enabled() || !halo_shape_ok(p) || halo_static_choice(p, false) == 0 || !al16(mean) || !al16(invstd) || !al16(gamma) ||
      !al16(beta))
    return (int)hipErrorInvalidValue;   // ask tris_conv3x3_bnin_ok first
  p.in_mean = mean; p.in_invstd = invstd; p.in_gamma = gamma; p.in_beta = beta;
  p.stat_part = (stat_part != nullptr && p.fastA && p.M >= 128) ? stat_part : nullptr;
  if (stat_rows) *stat_rows = 0;
  return conv3_dispatch<B_NK>(p, (hipStream_t)stream, stat_rows, true);
}

extern "C" int tris_conv3x3_wgrad_bnin_f32(const float* X, const float* mean, const float* invstd, const float* gamma,
                                           const float* beta, const float* dY, float* dW, int B, int H, int W, int Cin, int Cout,
                                           float* workspace, long ws_bytes, void* stream) {
  if (!bnin_enabled() || workspace == nullptr || !al16(X) || !al16(dY) || !al16(dW) || !al16(workspace) || !al16(mean) ||
      !al16(invstd) || !al16(gamma) || !al16(beta))
    return (int)hipErrorInvalidValue;
  int id = wg_static_choice(B, H, W, Cin, Cout, ws_bytes);
  if (id == 0) return (int)hipErrorInvalidValue;
  {  // the configuration the plain weight gradient timed as fastest for this shape, if it is a direct one
    const TuneKey key = {A_HALO, 64 + B_KN_IM2COL, Cout, 9 * Cin, B * H * W, H * 4096 + W, g_gemm_mode};
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tuned.find(key);
    if (it != g_tuned.end() && it->second.bm > 0) id = it->second.bm;
  }
  return run_wgrad_direct(id, X, dY, dW, B, H, W, Cin, Cout, workspace, ws_bytes, (hipStream_t)stream,
                          BnIn{mean, invstd, gamma, beta});
}

// ---- pre-split weight operands ("weight planes") ---------------------------------------------------------------------------
// In x3 arithmetic every block re-splits the tile of B it stages -- for a weight matrix that is the same work repeated by
// every M tile (2400 times in layer1).  tris_weight_planes_f32 splits the weights ONCE per optimiser step into three bf16
// planes (and the transposed planes the data-gradient products need); the *_wp entry points stage those planes straight
// into LDS (16-byte loads, no VALU).

// table (device, int64[entries][10]): src fp32 ptr, src row stride, rows, cols, P ptr, P plane stride, PT ptr, PT row
// stride, PT plane stride, first tile index.  P[pl][r*src_ld + c] = piece pl of src[r*src_ld + c];
// PT[pl][c*pt_ld + r] = the same piece transposed.  rows % 4 == 0, cols % 4 == 0.
__global__ __launch_bounds__(256) void weight_planes_kernel(const long* __restrict__ table, int entries) {
  __shared__ unsigned short tile[3][32][34];
  int lo = 0, hi = entries - 1;
  const long blk = blockIdx.x;
  while (lo < hi) {  // last entry whose first tile <= blk
    const int mid = (lo + hi + 1) >> 1;
    if (table[(long)mid * 10 + 9] <= blk) lo = mid; else hi = mid - 1;
  }
  const long* e = table + (long)lo * 10;
  const float* src = reinterpret_cast<const float*>(e[0]);
  const long src_ld = e[1];
  const int rows = (int)e[2], cols = (int)e[3];
  unsigned short* P = reinterpret_cast<unsigned short*>(e[4]);
  const long ppl = e[5];
  unsigned short* PT = reinterpret_cast<unsigned short*>(e[6]);
  const long pt_ld = e[7], ptpl = e[8];
  const int t = (int)(blk - e[9]);
  const int tiles_c = (cols + 31) / 32;
  const int r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
  const int tid = threadIdx.x, tr = tid >> 3, tc = (tid & 7) * 4;
  if (r0 + tr < rows && c0 + tc < cols) {
    const float4 v = *reinterpret_cast<const float4*>(src + (long)(r0 + tr) * src_ld + c0 + tc);
    const Split4 sp = split4(v);
    const long o = (long)(r0 + tr) * src_ld + c0 + tc;
    *reinterpret_cast<uint2*>(P + o) = sp.hi;
    *reinterpret_cast<uint2*>(P + ppl + o) = sp.mid;
    *reinterpret_cast<uint2*>(P + 2 * ppl + o) = sp.lo;
    const uint2 pc[3] = {sp.hi, sp.mid, sp.lo};
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      tile[pl][tr][tc] = (unsigned short)(pc[pl].x & 0xffffu);
      tile[pl][tr][tc + 1] = (unsigned short)(pc[pl].x >> 16);
      tile[pl][tr][tc + 2] = (unsigned short)(pc[pl].y & 0xffffu);
      tile[pl][tr][tc + 3] = (unsigned short)(pc[pl].y >> 16);
    }
  }
  __syncthreads();
  if (PT != nullptr && c0 + tr < cols && r0 + tc < rows) {  // transposed: row = original column c0+tr, 4 original rows
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      uint2 w;
      w.x = (unsigned)tile[pl][tc][tr] | ((unsigned)tile[pl][tc + 1][tr] << 16);
      w.y = (unsigned)tile[pl][tc + 2][tr] | ((unsigned)tile[pl][tc + 3][tr] << 16);
      *reinterpret_cast<uint2*>(PT + pl * ptpl + (long)(c0 + tr) * pt_ld + r0 + tc) = w;
    }
  }
}

extern "C" int tris_weight_planes_f32(const long* table, int entries, long total_tiles, void* stream) {
  if (entries <= 0 || total_tiles <= 0) return 0;
  hipLaunchKernelGGL(weight_planes_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, table, entries);
  TRIS_LAUNCH_CHECK();
  return 0;
}

// C[M,N] = act(A[M,K] . B^T + bias[n]) + resid[M,N] with B given as bf16 planes [3][N][K] (plane stride bpl elements).
// Optional fused BN statistics of C (stat_part / stat_rows as in tris_gemm_bnstat_f32; pass NULL for none).
// Returns TRIS_WP_UNSUPPORTED when the planes cannot be used for this shape / arithmetic mode.
extern "C" int tris_gemm_wp_f32(const float* A, const void* Bplanes, long bpl, float* C, int M, int N, int K,
                                const float* bias, const float* resid, int act, float* workspace, long ws_bytes,
                                double* stat_part, int* stat_rows, void* stream) {
  GemmParams p = {};
  p.A = A; p.B = reinterpret_cast<const float*>(Bplanes); p.bpl = bpl; p.C = C; p.M = M; p.N = N; p.K = K;
  p.lda = K; p.ldb = K; p.ldc = N; p.alpha = 1.f;
  p.bias = bias; p.bias_mode = bias ? 1 : 0; p.resid = resid; p.ldr = N; p.act = act;
  p.vecA = al16(A) && (K % 4 == 0);
  p.vecB = al16(Bplanes) && (K % 8 == 0) && (bpl % 8 == 0);
  p.fastA = p.vecA;
  p.fastB = p.vecB;
  if (stat_part != nullptr) {
    const bool ok = stats_eligible(p) && bias == nullptr && resid == nullptr && act == 0;
    p.stat_part = ok ? stat_part : nullptr;
    *stat_rows = ok ? cdiv(M, 128) : 0;
    return launch_cfg<A_ROWK, B_NK_PRE>(p, 1, nullptr, 0, (hipStream_t)stream);
  }
  return launch_cfg<A_ROWK, B_NK_PRE>(p, 1, workspace, ws_bytes, (hipStream_t)stream);
}

// tris_conv3x3_fwd[_bnstat]_f32 with the weights given as planes [3][Cout][9*Cin].  With the transposed + tap-mirrored
// planes Wd[ci][tap'][co] = W[co][8-tap'][ci] and (Cin, Cout) swapped this is also the data gradient of a stride-1 conv.
extern "C" int tris_conv3x3_wp_fwd_f32(const float* X, const void* Wplanes, long bpl, float* Y, int B, int H, int W, int Cin,
                                       int Cout, int stride, double* stat_part, int* stat_rows, void* stream) {
  int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  GemmParams p = {};
  p.A = X; p.B = reinterpret_cast<const float*>(Wplanes); p.bpl = bpl; p.C = Y;
  p.M = B * Ho * Wo; p.N = Cout; p.K = 9 * Cin;
  p.ldb = 9L * Cin; p.ldc = Cout; p.alpha = 1.f;
  p.gH = H; p.gW = W; p.gC = Cin; p.gHo = Ho; p.gWo = Wo; p.gStride = stride; p.gB = B;
  p.vecA = al16(X) && (Cin % 16 == 0);
  p.vecB = al16(Wplanes) && ((9 * Cin) % 8 == 0) && (bpl % 8 == 0);
  p.fastA = al16(X) && (Cin % 32 == 0);
  p.fastB = p.vecB;
  if (stat_part != nullptr) {
    p.stat_part = stats_eligible(p) ? stat_part : nullptr;
    *stat_rows = p.stat_part ? cdiv(p.M, 128) : 0;
  }
  return launch_cfg<A_IM2COL, B_NK_PRE>(p, 1, nullptr, 0, (hipStream_t)stream);
}

extern "C" int tris_set_gemm_mode(int mode) {
  if (mode < 0 || mode > 3) return (int)hipErrorInvalidValue;
  g_mode_default = mode;
  return 0;
}
extern "C" int tris_set_gemm_mode_thread(int mode) {
  if (mode < -1 || mode > 3) return (int)hipErrorInvalidValue;
  g_mode_thread = mode;
  return 0;
}
extern "C" int tris_set_autotune(int on) {
  g_autotune = on ? 1 : 0;
  return 0;
}

extern "C" int tris_get_gemm_mode(void) { return g_gemm_mode; }

extern "C" int tris_h2_next(const unsigned* amaxA, const unsigned* amaxB, float scaleA, float scaleB) {
  g_h2_next = H2Next{amaxA, amaxB, scaleA, scaleB, true};
  return 0;
}

extern "C" long tris_direct_launches(int kind) { return (kind == 0 || kind == 1) ? g_direct_launches[kind] : -1; }
