// Direct 3x3 convolution kernels and their launchers ("direct 3x3 convolutions"): the A_HALO
// instantiations of gemm_fast_kernel (forward / data gradient), the direct weight gradient and the stem's first convolution.
// Translation units of their own (one per arithmetic) so that they compile next to gemm_conv.hip, which keeps the dispatch
// (conv3_dispatch, the entry points) and calls in through the hidden symbols at the bottom.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "tris_hip.h"

namespace {

#include "gemm_params.h"
#include "gemm_fast.h"
#include "conv_direct_cfg.h"

// ---- the stem's first convolution (Cin = 3, stride 2): dedicated forward ----------------------------------------------------
// [B,320,320,3] -> [B,160,160,32]: K = 27 does not fill an MFMA k tile of the tiled kernels (16 TFLOP/s as an implicit GEMM with
// 128 us for a product whose HBM floor is ~40 us).  Here a wave owns 32 consecutive output pixels x 32 channels per trip: the
// 27 (padded to 28) weights of its (channel, k half) live in registers for the whole kernel, the input values are gathered
// straight into the A operand of v_mfma_f32_32x32x2_f32 (exact fp32: 14 MFMAs per tile), the 32 x 32 result is stored as sixteen
// 128-byte pixel rows.  The BatchNorm statistics of the output come from the same registers (one fp64 partial row per block).
__global__ __launch_bounds__(256) void stem_conv1_kernel(const float* __restrict__ X, const float* __restrict__ Wt,
                                                         float* __restrict__ Y, double* __restrict__ stat_part, int Bn, int H,
                                                         int W, int Cin, int Cout, int Ho, int Wo, int stride, int tiles_per_wave) {
  __shared__ double red[4][2][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mn = lane & 31, kh = lane >> 5;
  const int K = 9 * Cin;
  const long total = (long)Bn * Ho * Wo;
  float wreg[14];
  int ky[14], kx[14], kc[14];
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int k = 2 * s + kh;
    const int tap = k / Cin;
    wreg[s] = (k < K && mn < Cout) ? Wt[mn * K + k] : 0.f;
    ky[s] = k < K ? tap / 3 : -100000;   // (an always-out-of-range row: the padded k contributes zero)
    kx[s] = tap - (tap / 3) * 3;
    kc[s] = k - tap * Cin;
  }
  double s1 = 0.0, s2 = 0.0;   // (fp64 like the separate statistics pass: the kernel is gather-bound, the adds are free)
  const long tile0 = ((long)blockIdx.x * 4 + wave) * tiles_per_wave;
  for (int t = 0; t < tiles_per_wave; ++t) {
    const long p0 = (tile0 + t) * 32;
    if (p0 >= total) break;   // (wave-uniform)
    const long p = min(p0 + mn, total - 1);
    const int b = (int)(p / ((long)Ho * Wo));
    const int r = (int)(p - (long)b * Ho * Wo);
    const int oy = r / Wo, ox = r - oy * Wo;
    const int iy0 = oy * stride - 1, ix0 = ox * stride - 1;
    float a[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      const int iy = iy0 + ky[s], ix = ix0 + kx[s];
      const bool inb = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      a[s] = inb ? X[(((long)b * H + iy) * W + ix) * Cin + kc[s]] : 0.f;
    }
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int s = 0; s < 14; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wreg[s], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {  // D[row = (q&3) + 8*(q>>2) + 4*kh][col = mn]
      const long pix = p0 + (q & 3) + 8 * (q >> 2) + 4 * kh;
      if (pix < total && mn < Cout) {
        const float v = acc[q];
        Y[pix * Cout + mn] = v;
        s1 += (double)v;
        s2 += (double)v * (double)v;
      }
    }
  }
  if (stat_part != nullptr) {
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (kh == 0) { red[wave][0][mn] = s1; red[wave][1][mn] = s2; }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int which = threadIdx.x >> 5, c = threadIdx.x & 31;
      if (c < Cout)
        stat_part[((long)blockIdx.x * 2 + which) * Cout + c] =
            red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
    }
  }
}
// rows of the fp64 partial buffer = blocks launched; 0 = shape not served
static int run_stem_conv1(const float* X, const float* Wt, float* Y, int B, int H, int W, int Cin, int Cout, int stride,
                          double* stat_part, hipStream_t st) {
  if (9 * Cin > 28 || Cout > 32 || (stride != 1 && stride != 2)) return 0;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const long tiles = cdiv((long)B * Ho * Wo, 32L);
  int nblk = (int)std::min<long>(512, cdiv(tiles, 4L));
  if (stat_part != nullptr && nblk > cdiv(B * Ho * Wo, 128)) nblk = std::max(1, cdiv(B * Ho * Wo, 128));   // partial-buffer capacity
  const int tpw = (int)cdiv(tiles, (long)nblk * 4);
  nblk = (int)cdiv(tiles, (long)tpw * 4);
  hipLaunchKernelGGL(stem_conv1_kernel, dim3(nblk), dim3(256), 0, st, X, Wt, Y, stat_part, B, H, W, Cin, Cout, Ho, Wo, stride, tpw);
  return hipGetLastError() == hipSuccess ? nblk : -1;
}

template <int BKIND, int PREC>
int run_halo(GemmParams p, int id, hipStream_t st) {
  const HaloCfg& h = kHalo[id];
  p.hmode = h.hmode;
  p.tiles_n = cdiv(p.N, h.bn);
  p.splitk = 1;
  p.kchunk = p.gC;
  p.xcd_remap = 0;
  p.vecC = (p.N % 4 == 0) && al16(p.C) && (p.ldc % 4 == 0) && (!p.resid || (al16(p.resid) && p.ldr % 4 == 0));
  dim3 grid((unsigned)(halo_tiles_m(p, id) * p.tiles_n), 1, 1);
#define TRIS_HALO_GO(BM_, BN_, NW_, NWM_, HS_)                                                                          \
  hipLaunchKernelGGL((gemm_fast_kernel<BM_, BN_, A_HALO, BKIND, EPI_STD, PREC, NW_, 16, 2, NWM_, HS_>), grid, dim3(NW_ * 64), 0, st, p)
  switch (id) {
    case 1: TRIS_HALO_GO(256, 128, 8, 4, 324); break;
    case 2: TRIS_HALO_GO(256, 64, 4, 4, 324); break;
    case 3: TRIS_HALO_GO(256, 32, 4, 4, 324); break;
    case 4: TRIS_HALO_GO(128, 128, 4, 2, 324); break;
    case 5: TRIS_HALO_GO(256, 128, 8, 4, 452); break;
    case 6: TRIS_HALO_GO(128, 128, 4, 2, 180); break;
    default: return (int)hipErrorInvalidValue;
  }
#undef TRIS_HALO_GO
  TRIS_LAUNCH_CHECK();
  return 0;
}

// 3x3 convolution, stride 1 (forward and data gradient): direct kernel or implicit GEMM, timed once per shape like the tile
// choice.  TRIS_CONV_DIRECT=0 keeps the implicit GEMM, =1..6 forces a direct configuration where it applies (tests).
// *stat_rows (when statistics are fused) = number of partial rows the chosen kernel writes.
// ---- direct 3x3 weight gradient ---------------------------------------------------------------------------------------------
// dW[co][tap][ci] = sum over pixels p of dY[p][co] * X[p + tap][ci].  As an implicit GEMM (A_COLK x B_KN_IM2COL) every one of
// the 9 Cin / BN column tiles re-stages -- and re-splits -- the same dY rows and its own shifted copy of the same X rows, and
// for the wide early stages (160 x 160 x 32, 80 x 80 x 64: one or two row tiles, K = 10^5..10^6 pixels) that staging is the
// whole kernel.  Here a block owns a (COT x CIT) tile of (co, ci) for ALL nine taps and walks R x 16 pixel windows: the
// window's dY rows and its (R+2) x 18 input rows are split ONCE into k-major LDS planes; the nine taps read the SAME input
// image at nine slot offsets (ds_read_b64_tr_b16 fragments, as in the GEMM's k-major kinds).  Accumulators: 9 taps x 32 x 32
// per wave (144 registers).  WK > 1: waves share a tile and take alternate window rows (narrow tiles: the stem).
// Blocks (and the WK waves of a block) write partial tiles to slabs [slice][Co][9 Cin], summed by splitk_reduce_kernel.
// PREC 3 ("h2", x3_split.h): two fp16 pieces per operand, scales from the amax words of dY and of the convolution's input (for the
// BatchNorm-folded form: an upper bound of it).  The nine taps' cross products (hi x lo', lo' x hi) do not get nine accumulator
// sets of their own -- 288 registers -- but ONE transient set per tap and window, folded into the tap's accumulator with its
// weight 2^-11 before the next tap starts (16 VALU operations per 3 (R x XW / 16 / WK) MFMAs).
template <int COT, int CIT, int WCO, int WCI, int WK, int R, int OCC = 2, int XW = 16, int PREC = 1>
__global__ __launch_bounds__(256, OCC) void wgrad3x3_direct_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                                 float* __restrict__ slab, int H, int W, int Ci, int Co,
                                                                 int n_ci_tiles, int units_per_block, int n_units,
                                                                 const float* __restrict__ in_mean, const float* __restrict__ in_invstd,
                                                                 const float* __restrict__ in_gamma, const float* __restrict__ in_beta,
                                                                 const unsigned* __restrict__ amax_dy, const unsigned* __restrict__ amax_x) {
  constexpr bool H2 = (PREC == 3 || PREC == 4);
  constexpr bool PLN = (PREC == 4);   // X and dY arrive as fp16 piece planes (gemm_fast.h PREC 4): pieces go to LDS as they are
  constexpr int NPL = H2 ? 2 : 3;   // 16-bit planes per operand
  // XW = window width in pixels: 16 (a 16-pixel k group = one window row) or 8 (= two window rows: W = 40)
  constexpr int NPX = R * XW, NSL = (R + 2) * (XW + 2), PITCH = XW + 2, NG = NPX / 16;
  static_assert(WCO * WCI * WK == 4 && COT == 32 * WCO && CIT == 32 * WCI && NG % WK == 0 && (XW == 16 || XW == 8), "wave layout");
  constexpr int KS_A = COT == 32 ? 64 : 2 * COT + 64, KS_X = CIT == 32 ? 64 : 2 * CIT + 64;  // bytes per k row: odd multiples of 64
  constexpr int PA = NPX * COT / 4 / 256, PX = (NSL * CIT / 4 + 255) / 256;
  static_assert(PA * 256 * 4 == NPX * COT, "dY window / thread count mismatch");
  __shared__ __attribute__((aligned(16))) char Ash[NPL * NPX * KS_A];
  __shared__ __attribute__((aligned(16))) char Xsh[NPL * NSL * KS_X];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float sc_a = 1.f, sc_x = 1.f;
  if constexpr (H2) {
    if (amax_dy != nullptr) sc_a = h2_scale_from_bits(h2_amax_of(amax_dy, lane));
    if (amax_x != nullptr) sc_x = h2_scale_from_bits(h2_amax_of(amax_x, lane));
  }
  auto split_a = [&](const float4& v) { if constexpr (H2) return split4h(v, sc_a); else return split4(v); };
  auto split_x = [&](const float4& v) { if constexpr (H2) return split4h(v, sc_x); else return split4(v); };
  const int wk = wave % WK, wci = (wave / WK) % WCI, wco = wave / (WK * WCI);
  const int kh = lane >> 5;
  const int co0 = (blockIdx.x / n_ci_tiles) * COT, ci0 = (blockIdx.x % n_ci_tiles) * CIT;
  const int u_beg = blockIdx.y * units_per_block, u_end = min(n_units, u_beg + units_per_block);
  const int xsn = W / XW, upi = xsn * (H / R);   // units per image

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  float4 ra[PA], rx[PX];
  // in_mean != NULL: X is the raw input of a BatchNorm + ReLU and the convolution's input relu(bn(X)) is formed here, once per
  // window slot (tris_bn_apply_f32's expression); a thread always stages the same four input channels
  unsigned rx_ok = 0;
  float4 x_mu = make_float4(0.f, 0.f, 0.f, 0.f), x_sc = x_mu, x_be = x_mu;
  if (in_mean != nullptr) {
    const int c = ci0 + (tid % (CIT / 4)) * 4;
    const float4 is = ld4(in_invstd + c), ga = ld4(in_gamma + c);
    x_mu = ld4(in_mean + c);
    x_be = ld4(in_beta + c);
    x_sc = make_float4(is.x * ga.x, is.y * ga.y, is.z * ga.z, is.w * ga.w);
  }
  auto load_unit = [&](int u) {
    const int b = u / upi, r0 = u - b * upi;
    const int y0 = (r0 / xsn) * R, x0 = (r0 - (r0 / xsn) * xsn) * XW;
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int j = tid + q * 256;
      const int px = j / (COT / 4), c4 = j - px * (COT / 4);
      ra[q] = ld4(dY + ((long)(b * H + y0 + px / XW) * W + x0 + px % XW) * Co + co0 + c4 * 4);
    }
#pragma unroll
    for (int q = 0; q < PX; ++q) {
      const int j = tid + q * 256;
      const int sl = j / (CIT / 4), c4 = j - sl * (CIT / 4);
      const int sy = sl / PITCH, sx = sl - sy * PITCH;
      const int iy = y0 + sy - 1, ix = x0 + sx - 1;
      const bool ok = sl < NSL && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const float4 v = ld4(X + (ok ? ((long)(b * H + iy) * W + ix) * Ci + ci0 + c4 * 4 : 0));
      rx[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      rx_ok = ok ? (rx_ok | (1u << q)) : (rx_ok & ~(1u << q));
    }
  };
  auto store_unit = [&]() {
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int j = tid + q * 256;
      const int px = j / (COT / 4), c4 = j - px * (COT / 4);
      if constexpr (PLN) {
        *reinterpret_cast<float4*>(Ash + (c4 & 1) * (NPX * KS_A) + px * KS_A + (c4 >> 1) * 16) = ra[q];
        continue;
      }
      const Split4 sp = split_a(ra[q]);
      char* d = Ash + px * KS_A + c4 * 8;
      *reinterpret_cast<uint2*>(d) = sp.hi;
      *reinterpret_cast<uint2*>(d + NPX * KS_A) = sp.mid;
      if constexpr (!H2) *reinterpret_cast<uint2*>(d + 2 * NPX * KS_A) = sp.lo;
    }
#pragma unroll
    for (int q = 0; q < PX; ++q) {
      const int j = tid + q * 256;
      const int sl = j / (CIT / 4), c4 = j - sl * (CIT / 4);
      if (PX * 256 * 4 == NSL * CIT || sl < NSL) {
        float4 v = rx[q];
        if constexpr (PLN) {
          *reinterpret_cast<float4*>(Xsh + (c4 & 1) * (NSL * KS_X) + sl * KS_X + (c4 >> 1) * 16) = v;
          continue;
        }
        if (in_mean != nullptr && ((rx_ok >> q) & 1u)) {
          v.x = fmaxf((v.x - x_mu.x) * x_sc.x + x_be.x, 0.f);
          v.y = fmaxf((v.y - x_mu.y) * x_sc.y + x_be.y, 0.f);
          v.z = fmaxf((v.z - x_mu.z) * x_sc.z + x_be.z, 0.f);
          v.w = fmaxf((v.w - x_mu.w) * x_sc.w + x_be.w, 0.f);
        }
        const Split4 sp = split_x(v);
        char* d = Xsh + sl * KS_X + c4 * 8;
        *reinterpret_cast<uint2*>(d) = sp.hi;
        *reinterpret_cast<uint2*>(d + NSL * KS_X) = sp.mid;
        if constexpr (!H2) *reinterpret_cast<uint2*>(d + 2 * NSL * KS_X) = sp.lo;
      }
    }
  };
  const int m16 = wco * 32 + ((lane >> 4) & 1) * 16, n16 = wci * 32 + ((lane >> 4) & 1) * 16;
  if (u_beg < u_end) load_unit(u_beg);
  for (int u = u_beg; u < u_end; ++u) {
    __syncthreads();   // the previous window has been consumed
    store_unit();
    __syncthreads();
    load_unit(min(u + 1, u_end - 1));   // (unconditional: a load under a branch stalls on itself, see gemm_fast.h)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (H2) {
      constexpr int NGW = NG / WK;
      f16x8 ah[NGW], al[NGW];
#pragma unroll
      for (int g = 0; g < NGW; ++g) {
        const int row = g * WK + wk;
        ah[g] = __builtin_bit_cast(f16x8, tr_frag8(Ash, KS_A, row * 16 + 8 * kh, m16, lane));
        al[g] = __builtin_bit_cast(f16x8, tr_frag8(Ash + NPX * KS_A, KS_A, row * 16 + 8 * kh, m16, lane));
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        f32x16 cx;
#pragma unroll
        for (int r = 0; r < 16; ++r) cx[r] = 0.f;
#pragma unroll
        for (int g = 0; g < NGW; ++g) {
          const int row = g * WK + wk;
          const int k0 = XW == 16 ? (row + t / 3) * 18 + (t % 3) + 8 * kh : (2 * row + kh + t / 3) * 10 + (t % 3);
          const f16x8 bh = __builtin_bit_cast(f16x8, tr_frag8(Xsh, KS_X, k0, n16, lane));
          const f16x8 bl = __builtin_bit_cast(f16x8, tr_frag8(Xsh + NSL * KS_X, KS_X, k0, n16, lane));
          cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[g], bh, cx, 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g], bh, acc[t], 0, 0, 0);
          cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g], bl, cx, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = fmaf(cx[r], 1.0f / 2048.0f, acc[t][r]);
      }
    } else
#pragma unroll
    for (int g = 0; g < NG / WK; ++g) {
      const int row = g * WK + wk;   // this wave's k group: 16 pixels = window row `row` (XW 16) or rows 2 row, 2 row + 1 (XW 8)
      Split8 a;
      a.hi = tr_frag8(Ash, KS_A, row * 16 + 8 * kh, m16, lane);
      a.mid = tr_frag8(Ash + NPX * KS_A, KS_A, row * 16 + 8 * kh, m16, lane);
      a.lo = tr_frag8(Ash + 2 * NPX * KS_A, KS_A, row * 16 + 8 * kh, m16, lane);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int k0 = XW == 16 ? (row + t / 3) * 18 + (t % 3) + 8 * kh : (2 * row + kh + t / 3) * 10 + (t % 3);
        Split8 b;
        b.hi = tr_frag8(Xsh, KS_X, k0, n16, lane);
        b.mid = tr_frag8(Xsh + NSL * KS_X, KS_X, k0, n16, lane);
        b.lo = tr_frag8(Xsh + 2 * NSL * KS_X, KS_X, k0, n16, lane);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.mid, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.hi, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.mid, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc[t], 0, 0, 0);
      }
    }
  }
  if constexpr (WK > 1) {
    // the WK waves of a (co, ci) quadrant hold partial sums over alternate window rows: add them up tap by tap through LDS
    // (fixed order wk = 1, 2, ..: deterministic); wave wk = 0 keeps the result
    static_assert((WK - 1) * WCO * WCI * 16 * 64 * 4 <= (int)sizeof(Xsh), "reduction scratch does not fit");
    float* red = reinterpret_cast<float*>(Xsh);
    const int quad = wco * WCI + wci;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      __syncthreads();
      if (wk > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(((wk - 1) * (WCO * WCI) + quad) * 16 + r) * 64 + lane] = acc[t][r];
      }
      __syncthreads();
      if (wk == 0) {
#pragma unroll
        for (int w = 1; w < WK; ++w)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] += red[(((w - 1) * (WCO * WCI) + quad) * 16 + r) * 64 + lane];
      }
    }
  }
  // partial tile of this block -> slab blockIdx.y: rows co, columns (tap, ci)
  if (wk == 0) {
    float* o = slab + (long)blockIdx.y * Co * 9 * Ci;
    const int ci = ci0 + wci * 32 + (lane & 31);
    const float inv = H2 ? 1.0f / (sc_a * sc_x) : 1.0f;   // (h2: the operand scales leave here -- powers of two, exact)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        o[((long)co * 9 + t) * Ci + ci] = H2 ? acc[t][r] * inv : acc[t][r];
      }
  }
}

// out[i] = sum over S slabs of ws[s][i] (i < total, total % 4 == 0).  The direct weight gradient leaves hundreds of slabs of a
// SMALL matrix (9 K..150 K elements): 64 column vectors x 4 slab groups per block, 4 loads in flight per thread, the groups
// meet in LDS in fixed order (deterministic).
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ ws, int S, long total, float* __restrict__ out) {
  __shared__ float4 sh[3][64];
  const int cv = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long idx = ((long)blockIdx.x * 64 + cv) * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (idx < total) {
    int s = g;
    for (; s + 12 < S; s += 16) {
      const float4 a = ld4(ws + (long)s * total + idx), b = ld4(ws + (long)(s + 4) * total + idx);
      const float4 c = ld4(ws + (long)(s + 8) * total + idx), d = ld4(ws + (long)(s + 12) * total + idx);
      v.x = (((v.x + a.x) + b.x) + c.x) + d.x;
      v.y = (((v.y + a.y) + b.y) + c.y) + d.y;
      v.z = (((v.z + a.z) + b.z) + c.z) + d.z;
      v.w = (((v.w + a.w) + b.w) + c.w) + d.w;
    }
    for (; s < S; s += 4) {
      const float4 a = ld4(ws + (long)s * total + idx);
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
  }
  if (g > 0) sh[g - 1][cv] = v;
  __syncthreads();
  if (g == 0 && idx < total) {
#pragma unroll
    for (int q = 0; q < 3; ++q) { const float4 a = sh[q][cv]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
    *reinterpret_cast<float4*>(out + idx) = v;
  }
}

template <int PREC>
static int run_wgrad_direct(int id, const float* X, const float* dY, float* dW, int B, int H, int W, int Ci, int Co, float* ws,
                            long ws_bytes, int blocks, hipStream_t st, BnIn bn, const unsigned* amax_dy, const unsigned* amax_x) {
  const WgCfg& c = kWg[id];
  const int S = wg_slices(id, B, H, W, Ci, Co, ws_bytes, blocks);
  if (S < 1) return (int)hipErrorInvalidValue;
  const int units = B * (H / c.r) * (W / kWgXW[id]);
  const int upb = cdiv(units, S);
  const int slices = cdiv(units, upb);
  dim3 grid((unsigned)((Co / c.cot) * (Ci / c.cit)), (unsigned)slices);
  const int nci = Ci / c.cit;
#define TRIS_WG_GO(...)                                                                                                          \
  hipLaunchKernelGGL((wgrad3x3_direct_kernel<__VA_ARGS__, PREC>), grid, dim3(256), 0, st, X, dY, ws, H, W, Ci, Co, nci, upb, units, \
                     bn.mean, bn.invstd, bn.gamma, bn.beta, amax_dy, amax_x)
  // blocks per CU the kernel is compiled for: two in x3; h2 holds the tap accumulators PLUS a transient set and the fragments of
  // all its k groups -- more than 256 registers -- and runs one block per CU
  constexpr int OCC = (PREC == 3 || PREC == 4) ? 1 : 2;
  switch (id) {
    case 1: TRIS_WG_GO(32, 32, 1, 1, 4, 4, OCC, 16); break;
    case 2: TRIS_WG_GO(64, 32, 2, 1, 2, 4, OCC, 16); break;
    case 3: TRIS_WG_GO(64, 64, 2, 2, 1, 2, OCC, 16); break;
    case 4: TRIS_WG_GO(64, 64, 2, 2, 1, 4, 1, 16); break;
    case 5: TRIS_WG_GO(64, 32, 2, 1, 2, 8, OCC, 8); break;
    default: return (int)hipErrorInvalidValue;
  }
#undef TRIS_WG_GO
  TRIS_LAUNCH_CHECK();
  const long total = (long)Co * 9 * Ci;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)cdiv(total / 4, 64)), dim3(256), 0, st, ws, slices, total, dW);
  TRIS_LAUNCH_CHECK();
  return 0;
}

}  // namespace

#define TRIS_HIDDEN extern "C" __attribute__((visibility("hidden")))
// One arithmetic per translation unit (build.sh: -DTRIS_DIRECT_PREC=1 -> x3, =3 -> h2, =4 -> h2 on operand planes); the stem's first convolution (exact f32
// MFMA) lives in the x3 unit.
#ifndef TRIS_DIRECT_PREC
#error "compile with -DTRIS_DIRECT_PREC=1, 3 or 4"
#endif
#define TRIS_CAT2(a, b) a##b
#define TRIS_DIRECT_NAME(base, prec) TRIS_CAT2(base, prec)
// params: a GemmParams (same layout in every unit: gemm_params.h); dgrad: mirrored-tap weight loader
TRIS_HIDDEN int TRIS_DIRECT_NAME(tris_internal_run_halo_p, TRIS_DIRECT_PREC)(const void* params, int id, int dgrad, void* stream) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  hipStream_t st = (hipStream_t)stream;
  return dgrad ? run_halo<B_KN_DGRAD, TRIS_DIRECT_PREC>(p, id, st) : run_halo<B_NK, TRIS_DIRECT_PREC>(p, id, st);
}
TRIS_HIDDEN int TRIS_DIRECT_NAME(tris_internal_run_wgrad_direct_p, TRIS_DIRECT_PREC)(
    int id, const float* X, const float* dY, float* dW, int B, int H, int W, int Ci, int Co, float* ws, long ws_bytes, int blocks,
    void* stream, const float* mean, const float* invstd, const float* gamma, const float* beta, const unsigned* amax_dy,
    const unsigned* amax_x) {
  return run_wgrad_direct<TRIS_DIRECT_PREC>(id, X, dY, dW, B, H, W, Ci, Co, ws, ws_bytes, blocks, (hipStream_t)stream,
                                            BnIn{mean, invstd, gamma, beta}, amax_dy, amax_x);
}
#if TRIS_DIRECT_PREC == 1
TRIS_HIDDEN int tris_internal_stem_conv1(const float* X, const float* Wt, float* Y, int B, int H, int W, int Cin, int Cout, int stride,
                                         double* stat_part, void* stream) {
  return run_stem_conv1(X, Wt, Y, B, H, W, Cin, Cout, stride, stat_part, (hipStream_t)stream);
}
#endif
