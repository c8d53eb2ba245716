// Fast path of the MFMA GEMM / implicit-GEMM core (included by gemm_conv.hip).
//
// BM x BN block (128x128 with 8 waves 2x4, 128x64 / 64x64 with 4 waves 2x2, 128x32 with 2 waves 2x1), 32x32 MFMA fragments:
//   * BK = 32 and *branch-free* tile loaders: rows/columns past the edge are clamped (their results are never
//     stored) and out-of-image im2col taps are loaded from a valid address and zeroed with a select, so the K loop has
//     no divergent control flow and the accumulators stay pinned in registers;
//   * PREC 0 (f32-input MFMA): k-contiguous operands (row-major A, B^T, im2col) keep a row-major LDS image [rows][32+4]:
//     16-byte global loads go to LDS as ds_write_b128, and a fragment read is ONE ds_read_b128 per lane = 4 k-values
//     feeding 4 MFMAs.  Row stride 36 floats puts the 16 lanes of every ds_read_b128 service group on 16 distinct
//     16-byte slots (9*i mod 16 is a permutation) -> conflict-free.  The logical k order inside an MFMA is permuted
//     (lane-half kh, MFMA j  <->  k = 8g + 4kh + j); both operands use the same permutation, the sum is unchanged;
//     m-contiguous operands (A^T for wgrad, B for dgrad / NN) keep the k-major image and read 4 scalars;
//   * PREC 1 / 2 (split-bf16 x3 / x2, below): operands are split when the tile is stored, the LDS image is bf16 planes --
//     row-major [plane][row][32+8] for the k-contiguous kinds, k-major [plane][k][m] read with ds_read_b64_tr_b16 for the
//     m-contiguous ones;
//   * epilogue: bias / activation / residual / BN statistics, stored as 16-byte rows after an in-LDS turn of each wave's
//     accumulator block (scalar fallback for unaligned or N % 4 != 0 outputs).
// Preconditions (checked on the host, otherwise the generic kernel runs): K % 32 == 0 per k-slice, 16-byte aligned
// operands, M % 4 == 0 / N % 4 == 0 for the m-/n-contiguous kinds, gathered channels % 32 == 0 for im2col.
#pragma once

// ---- split-bf16 ("x3") arithmetic -----------------------------------------------------------------------------------------
// An fp32 value is EXACTLY the sum of three bf16 pieces (round-to-nearest residual splitting: 8 + 8 + 8 significand
// bits).  a*b = sum of the 9 piece products; the 6 with combined weight >= 2^-24 are kept, each is exact in fp32
// (8 x 8 bits) and is accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Result: fp32-class accuracy at 6 bf16 MFMAs per
// 16-deep step (6 x 32 cycles) instead of 8 f32 MFMAs (8 x 64 cycles).  Pieces are produced in registers right after
// the LDS fragment read (the LDS image stays fp32 and is shared with the f32 path).
#include "x3_split.h"

constexpr bool KM_TR = true;  // k-major operands via the LDS transpose read (false: "row per thread" scalar staging)

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
// 8 consecutive k (k0 .. k0+7) of column m = m16 + (lane & 15) from a k-major bf16 plane (row stride `ks` bytes)
__device__ __forceinline__ bf16x8 tr_frag8(const char* plane, int ks, int k0, int m16, int lane) {
  const int i16 = lane & 15;
  const char* a = plane + (k0 + (i16 >> 2)) * ks + (m16 + 4 * (i16 & 3)) * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * ks));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// NW = waves per workgroup: 4 (2x2 wave grid) or 8 (2x4: smaller wave tiles, twice the resident waves per CU -- used by
// the x3 mode on 128x128 tiles, where the bf16 MFMA time per tile is short and the barrier / staging phases need hiding)
template <int BM, int BN, int AK, int BKIND, int EPI, int PREC, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 4 : 2) void gemm_fast_kernel(GemmParams p) {
  constexpr int NTHR = NW * 64, RPASS = NTHR / 8, NWN = NW / 2;
  constexpr int FBK = 32;
  constexpr int LDK = FBK + 4;
  constexpr bool A_RM = (AK != A_COLK);
  // B_NK_PRE: the B operand arrives already split -- three bf16 planes [plane][N][K] (weights, split once per step by
  // tris_weight_planes_f32): 16-byte loads go straight to the LDS planes, no VALU work (x3 only).  Measured ON PAR with the
  // in-kernel split at this kernel's operating point: three half-line (64 B) streams per row cost in address/tag
  // work what the split saves in VALU, and the 8-wave kernel has no registers left to fetch full-line 64-k windows.
  constexpr bool B_PRE = (BKIND == B_NK_PRE);
  constexpr bool B_RM = (BKIND == B_NK) || B_PRE;
  constexpr int WM = BM / 2, WN = BN / NWN;
  constexpr int FM = WM / 32, FN = WN / 32;
  constexpr int PA = BM / RPASS, PB = BN / RPASS;  // float4 per thread per tile
  // x3 mode: every operand is split into its three bf16 pieces ONCE, when the tile is stored: the LDS image is three
  // bf16 planes [plane][row][32 + 8 pad] (row stride 80 B = 5 sixteen-byte slots -> conflict-free ds_read_b128), i.e.
  // 60 floats' worth per row.
  // PREC 1 = x3 (three bf16 pieces, six products: fp32-class); PREC 2 = x2 (two pieces, three products hi.hi + hi.mid + mid.hi:
  // 16 significand bits per operand, relative product error <= 2^-15 -- between fp32 and TF32; planes shrink to 4 B/element)
  constexpr bool A_PL = (PREC >= 1), B_PL = (PREC >= 1);
  constexpr int NPLN = PREC == 2 ? 2 : 3;  // bf16 planes per operand
  // x3 staging of the m-/n-contiguous operands ("row per thread"): thread -> one row (m or n) and KPT consecutive k,
  // loaded with scalar loads (a wave covers 64 consecutive rows = 256 contiguous bytes per k), split once, and written as
  // 16-byte bf16 runs into the same [plane][row][k] image the k-contiguous operands use
  // KM_TR (default): the m-/n-contiguous operands are instead staged like everything else -- 16-byte loads along the
  // contiguous dimension, split, 8-byte LDS stores into k-major planes [plane][k][m] -- and the MFMA fragments (8 consecutive
  // k per lane) are gathered by the LDS transpose read ds_read_b64_tr_b16: per 16-lane group, lane i points at the 8-byte
  // piece [k0 + i/4][m0 + 4(i%4) ..+3] and receives [k0..k0+3][m0 + i] (semantics established with tools/probes/
  // tr_read_probe.hip).  Plane row stride = 2*BM + 64 bytes: the four k rows of a group and the two groups of a 32-lane
  // half fall on disjoint banks.
  constexpr bool A_TR = KM_TR && (PREC >= 1) && !A_RM, B_TR = KM_TR && (PREC >= 1) && !B_RM;
  constexpr bool A_KM = (PREC >= 1) && !A_RM && !A_TR, B_KM = (PREC >= 1) && !B_RM && !B_TR;
  constexpr int A_KS = 2 * BM + 64, B_KS = 2 * BN + 64;  // bytes per k row of a k-major plane
  constexpr int A_KG = NTHR / BM, A_KPT = 32 / A_KG, B_KG = NTHR / BN, B_KPT = 32 / B_KG;
  constexpr int PLB = 80;  // bytes per row of one bf16 plane
  constexpr int A_SZ = A_TR ? NPLN * 8 * A_KS : A_PL ? BM * 20 * NPLN : (A_RM ? BM * LDK : FBK * (BM + 4));
  constexpr int B_SZ = B_TR ? NPLN * 8 * B_KS : B_PL ? BN * 20 * NPLN : (B_RM ? BN * LDK : FBK * (BN + 4));
  __shared__ __attribute__((aligned(16))) float As[A_SZ];
  __shared__ __attribute__((aligned(16))) float Bs[B_SZ];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int tile = blockIdx.x;
  const int m0 = (tile / p.tiles_n) * BM;
  const int n0 = (tile % p.tiles_n) * BN;
  const int zb = blockIdx.z / p.splitk;
  const int zs = blockIdx.z % p.splitk;
  const int kbeg = zs * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const float* __restrict__ A = p.A + (long)zb * p.sA;
  const float* __restrict__ Bp = p.B + (long)zb * p.sB;

  // Row handled by this thread's 8-lane set in the row-major kinds.  The split pieces go to LDS with ds_write_b64, which is
  // serviced in groups of 16 CONTIGUOUS lanes against a 32-bank (128-byte) modulus: two 8-lane sets = two rows of 64 bytes.
  // With the 80-byte plane rows, rows r and r+1 overlap on four banks (2-way: every store group took two LDS cycles --
  // SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS 0.84 against 0.01 for the transpose-read kinds); rows r and r+4 are exactly
  // 16 banks apart.  So within every 8 rows the lane sets visit rows 0,4,1,5,2,6,3,7: same layout, same global
  // coalescing (one 128-byte row piece per 8 lanes), conflict-free stores.
  const int trow = ((tid >> 3) & ~7) | (((tid >> 3) & 1) << 2) | ((tid >> 4) & 3);
  // ---- loader state ---------------------------------------------------------------------------------------------
  // row-major kinds: thread -> (row = tid>>3 + 32*q, kofs = (tid&7)*4);  k-major kinds: (k = tid/F4 + q*RPP, col4)
  long a_off[PA];  // ROWK: row offset; IM2COL: unused
  int a_b[PA], a_iy0[PA], a_ix0[PA];
  if (AK == A_ROWK) {
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      int m = min(m0 + trow + q * RPASS, p.M - 1);
      a_off[q] = (long)m * p.lda + (tid & 7) * 4;
    }
  } else if (AK == A_IM2COL) {
    const int hw = p.gHo * p.gWo;
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      int m = min(m0 + trow + q * RPASS, p.M - 1);
      a_b[q] = m / hw;
      int r = m - a_b[q] * hw;
      int oy = r / p.gWo, ox = r - oy * p.gWo;
      a_iy0[q] = oy * p.gStride - 1;
      a_ix0[q] = ox * p.gStride - 1;
    }
  }
  constexpr int AF4 = BM / 4, ARPP = NTHR / AF4;
  const int a_mc = min(m0 + (tid % AF4) * 4, p.M - 4);  // COLK column (clamped)
  long b_off[PB];
  if (BKIND == B_NK) {
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      int n = min(n0 + trow + q * RPASS, p.N - 1);
      b_off[q] = (long)n * p.ldb + (tid & 7) * 4;
    }
  }
  constexpr int BF4 = BN / 4, BRPP = NTHR / BF4;
  const int b_nc = min(n0 + (tid % BF4) * 4, p.N - 4);  // KN* column (clamped)
  int bj_tap = 0, bj_ci = 0;
  if (BKIND == B_KN_IM2COL) {
    bj_tap = b_nc / p.gC;
    bj_ci = b_nc - bj_tap * p.gC;
  }
  const int bj_ky = bj_tap / 3, bj_kx = bj_tap - (bj_tap / 3) * 3;

  float4 ra[PA], rb[PB];
  int bw_b[PB], bw_oy[PB], bw_ox[PB];  // B_KN_IM2COL: running (image, row, column) of this thread's pixel rows
  constexpr int PBP = B_PRE ? (BN * 4 + NTHR - 1) / NTHR : 1;  // 16-byte pieces (8 bf16) per thread per plane per tile
  float4 rbp[3 * PBP];  // (a flat float4 array: the 2-D uint4 form was not promoted to registers)
  const unsigned short* bp_src[PBP];
  if (B_PRE) {
#pragma unroll
    for (int q = 0; q < PBP; ++q) {
      const int row = min(n0 + (tid >> 2) + q * (NTHR / 4), p.N - 1);
      bp_src[q] = reinterpret_cast<const unsigned short*>(p.B) + (long)row * p.ldb + (tid & 3) * 8;
    }
  }
  float rka[A_KM ? A_KPT : 1], rkb[B_KM ? B_KPT : 1];
  const int a_rm = min(m0 + tid % BM, p.M - 1), a_kg = tid / BM;  // x3 row-per-thread coordinates
  const int b_rn = min(n0 + tid % BN, p.N - 1), b_kg = tid / BN;
  int bx_tap = 0, bx_ci = 0;
  if (BKIND == B_KN_IM2COL) { bx_tap = b_rn / p.gC; bx_ci = b_rn - bx_tap * p.gC; }
  const int bx_ky = bx_tap / 3, bx_kx = bx_tap - (bx_tap / 3) * 3;

  auto load_A = [&](int k0) {
    if (AK == A_ROWK) {
#pragma unroll
      for (int q = 0; q < PA; ++q) ra[q] = ld4(A + a_off[q] + k0);
    } else if (AK == A_IM2COL) {
      const int tap = k0 / p.gC;  // whole 32-wide k tile lies inside one tap (gC % 32 == 0)
      const int ci = k0 - tap * p.gC + (tid & 7) * 4;
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        const int iy = a_iy0[q] + ky, ix = a_ix0[q] + kx;
        const bool inb = (unsigned)iy < (unsigned)p.gH && (unsigned)ix < (unsigned)p.gW;
        const long off = inb ? ((long)(a_b[q] * p.gH + iy) * p.gW + ix) * p.gC + ci : 0;
        float4 v = ld4(A + off);
        ra[q] = inb ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else if (A_KM) {  // A_COLK, x3: KPT consecutive k of row a_rm
      const float* src = A + (long)(k0 + a_kg * A_KPT) * p.lda + a_rm;
#pragma unroll
      for (int j = 0; j < A_KPT; ++j) rka[j] = src[(long)j * p.lda];
    } else {  // A_COLK: A[k*lda + m]
#pragma unroll
      for (int q = 0; q < PA; ++q) ra[q] = ld4(A + (long)(k0 + tid / AF4 + q * ARPP) * p.lda + a_mc);
    }
  };

  auto load_B = [&](int k0) {
    if (B_PRE) {
#pragma unroll
      for (int q = 0; q < PBP; ++q)
#pragma unroll
        for (int pl = 0; pl < NPLN; ++pl) rbp[pl * PBP + q] = ld4(reinterpret_cast<const float*>(bp_src[q] + pl * p.bpl + k0));
    } else if (BKIND == B_NK) {
#pragma unroll
      for (int q = 0; q < PB; ++q) rb[q] = ld4(Bp + b_off[q] + k0);
    } else if (B_KM) {
      const int kb = k0 + b_kg * B_KPT;
      if (BKIND == B_KN) {
        const float* src = Bp + (long)kb * p.ldb + b_rn;
#pragma unroll
        for (int j = 0; j < B_KPT; ++j) rkb[j] = src[(long)j * p.ldb];
      } else if (BKIND == B_KN_DGRAD) {  // k = tap'*Cout + co (the KPT-run stays inside one tap: Cout % 32 == 0)
        const int tapp = kb / p.wCout, co = kb - tapp * p.wCout;
        const float* src = Bp + ((long)co * 9 + (8 - tapp)) * p.wCin + b_rn;
#pragma unroll
        for (int j = 0; j < B_KPT; ++j) rkb[j] = src[(long)j * 9 * p.wCin];
      } else {  // B_KN_IM2COL: k = output pixel (KPT consecutive pixels, walked incrementally), column = (tap, ci)
        const int hw = p.gHo * p.gWo;
        int b = kb / hw;
        const int r = kb - b * hw;
        int oy = r / p.gWo, ox = r - oy * p.gWo;
#pragma unroll
        for (int j = 0; j < B_KPT; ++j) {
          const int iy = oy * p.gStride - 1 + bx_ky, ix = ox * p.gStride - 1 + bx_kx;
          const bool inb = (unsigned)iy < (unsigned)p.gH && (unsigned)ix < (unsigned)p.gW;
          const long off = inb ? ((long)(b * p.gH + iy) * p.gW + ix) * p.gC + bx_ci : 0;
          const float v = Bp[off];
          rkb[j] = inb ? v : 0.f;
          if (++ox == p.gWo) { ox = 0; if (++oy == p.gHo) { oy = 0; ++b; } }
        }
      }
    } else if (BKIND == B_KN) {
#pragma unroll
      for (int q = 0; q < PB; ++q) rb[q] = ld4(Bp + (long)(k0 + tid / BF4 + q * BRPP) * p.ldb + b_nc);
    } else if (BKIND == B_KN_DGRAD) {  // k = tap'*Cout + co ; B[k][ci] = W[co][8 - tap'][ci]
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        if (k0 == kbeg) {  // (tap', co) of this thread's rows, carried from tile to tile like the im2col coordinates below
          const int kk = k0 + tid / BF4 + q * BRPP;
          bw_b[q] = kk / p.wCout;
          bw_oy[q] = kk - bw_b[q] * p.wCout;
        }
        rb[q] = ld4(Bp + ((long)bw_oy[q] * 9 + (8 - bw_b[q])) * p.wCin + b_nc);
        bw_oy[q] += FBK;
        while (bw_oy[q] >= p.wCout) { bw_oy[q] -= p.wCout; ++bw_b[q]; }
      }
    } else {  // B_KN_IM2COL: k = output pixel, column = (tap, ci) of the gathered input
      // the pixel coordinates of this thread's PB rows are carried from tile to tile (tiles are requested in order, 32
      // pixels apart): two integer divisions per row once, then additions
      if (k0 == kbeg) {
        const int hw = p.gHo * p.gWo;
#pragma unroll
        for (int q = 0; q < PB; ++q) {
          const int kk = k0 + tid / BF4 + q * BRPP;
          bw_b[q] = kk / hw;
          const int r = kk - bw_b[q] * hw;
          bw_oy[q] = r / p.gWo;
          bw_ox[q] = r - bw_oy[q] * p.gWo;
        }
      }
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        const int iy = bw_oy[q] * p.gStride - 1 + bj_ky, ix = bw_ox[q] * p.gStride - 1 + bj_kx;
        const bool inb = (unsigned)iy < (unsigned)p.gH && (unsigned)ix < (unsigned)p.gW;
        const long off = inb ? ((long)(bw_b[q] * p.gH + iy) * p.gW + ix) * p.gC + bj_ci : 0;
        float4 v = ld4(Bp + off);
        rb[q] = inb ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        bw_ox[q] += FBK;  // next tile
        while (bw_ox[q] >= p.gWo) { bw_ox[q] -= p.gWo; if (++bw_oy[q] == p.gHo) { bw_oy[q] = 0; ++bw_b[q]; } }
      }
    }
  };

  auto store_lds = [&]() {
    if (A_TR) {
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        const Split4 sp = split4(ra[q]);
        char* d = reinterpret_cast<char*>(As) + (tid / AF4 + q * ARPP) * A_KS + (tid % AF4) * 8;
        *reinterpret_cast<uint2*>(d) = sp.hi;
        *reinterpret_cast<uint2*>(d + 32 * A_KS) = sp.mid;
        if (NPLN == 3) *reinterpret_cast<uint2*>(d + 64 * A_KS) = sp.lo;
      }
    } else if (A_KM) {
#pragma unroll
      for (int h = 0; h < A_KPT / 8; ++h) {
        const Split8 sp = split8(make_float4(rka[8 * h], rka[8 * h + 1], rka[8 * h + 2], rka[8 * h + 3]),
                                 make_float4(rka[8 * h + 4], rka[8 * h + 5], rka[8 * h + 6], rka[8 * h + 7]));
        char* d = reinterpret_cast<char*>(As) + (tid % BM) * PLB + (a_kg * A_KPT + 8 * h) * 2;
        *reinterpret_cast<bf16x8*>(d) = sp.hi;
        *reinterpret_cast<bf16x8*>(d + BM * PLB) = sp.mid;
        if (NPLN == 3) *reinterpret_cast<bf16x8*>(d + 2 * BM * PLB) = sp.lo;
      }
    } else if (A_PL) {
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        const Split4 sp = split4(ra[q]);
        char* d = reinterpret_cast<char*>(As) + (trow + q * RPASS) * PLB + (tid & 7) * 8;
        *reinterpret_cast<uint2*>(d) = sp.hi;
        *reinterpret_cast<uint2*>(d + BM * PLB) = sp.mid;
        if (NPLN == 3) *reinterpret_cast<uint2*>(d + 2 * BM * PLB) = sp.lo;
      }
    } else if (A_RM) {
#pragma unroll
      for (int q = 0; q < PA; ++q)
        *reinterpret_cast<float4*>(&As[(trow + q * RPASS) * LDK + (tid & 7) * 4]) = ra[q];
    } else {
#pragma unroll
      for (int q = 0; q < PA; ++q)
        *reinterpret_cast<float4*>(&As[(tid / AF4 + q * ARPP) * (BM + 4) + (tid % AF4) * 4]) = ra[q];
    }
    if (B_PRE) {
#pragma unroll
      for (int q = 0; q < PBP; ++q) {
        const int row = (tid >> 2) + q * (NTHR / 4);
        if (PBP * (NTHR / 4) == BN || row < BN) {
          char* d = reinterpret_cast<char*>(Bs) + row * PLB + (tid & 3) * 16;
#pragma unroll
          for (int pl = 0; pl < NPLN; ++pl) *reinterpret_cast<float4*>(d + pl * BN * PLB) = rbp[pl * PBP + q];
        }
      }
    } else if (B_TR) {
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        const Split4 sp = split4(rb[q]);
        char* d = reinterpret_cast<char*>(Bs) + (tid / BF4 + q * BRPP) * B_KS + (tid % BF4) * 8;
        *reinterpret_cast<uint2*>(d) = sp.hi;
        *reinterpret_cast<uint2*>(d + 32 * B_KS) = sp.mid;
        if (NPLN == 3) *reinterpret_cast<uint2*>(d + 64 * B_KS) = sp.lo;
      }
    } else if (B_KM) {
#pragma unroll
      for (int h = 0; h < B_KPT / 8; ++h) {
        const Split8 sp = split8(make_float4(rkb[8 * h], rkb[8 * h + 1], rkb[8 * h + 2], rkb[8 * h + 3]),
                                 make_float4(rkb[8 * h + 4], rkb[8 * h + 5], rkb[8 * h + 6], rkb[8 * h + 7]));
        char* d = reinterpret_cast<char*>(Bs) + (tid % BN) * PLB + (b_kg * B_KPT + 8 * h) * 2;
        *reinterpret_cast<bf16x8*>(d) = sp.hi;
        *reinterpret_cast<bf16x8*>(d + BN * PLB) = sp.mid;
        if (NPLN == 3) *reinterpret_cast<bf16x8*>(d + 2 * BN * PLB) = sp.lo;
      }
    } else if (B_PL) {
#pragma unroll
      for (int q = 0; q < PB; ++q) {
#ifdef TRIS_EXP_NOBSPLIT   // experiment: what a pre-split (weight) operand would save -- raw bits instead of the split
        Split4 sp;
        sp.hi = make_uint2(__builtin_bit_cast(unsigned, rb[q].x), __builtin_bit_cast(unsigned, rb[q].y));
        sp.mid = make_uint2(__builtin_bit_cast(unsigned, rb[q].z), __builtin_bit_cast(unsigned, rb[q].w));
        sp.lo = sp.hi;
#else
        const Split4 sp = split4(rb[q]);
#endif
        char* d = reinterpret_cast<char*>(Bs) + (trow + q * RPASS) * PLB + (tid & 7) * 8;
        *reinterpret_cast<uint2*>(d) = sp.hi;
        *reinterpret_cast<uint2*>(d + BN * PLB) = sp.mid;
        if (NPLN == 3) *reinterpret_cast<uint2*>(d + 2 * BN * PLB) = sp.lo;
      }
    } else if (B_RM) {
#pragma unroll
      for (int q = 0; q < PB; ++q)
        *reinterpret_cast<float4*>(&Bs[(trow + q * RPASS) * LDK + (tid & 7) * 4]) = rb[q];
    } else {
#pragma unroll
      for (int q = 0; q < PB; ++q)
        *reinterpret_cast<float4*>(&Bs[(tid / BF4 + q * BRPP) * (BN + 4) + (tid % BF4) * 4]) = rb[q];
    }
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int li = lane & 31, kh = lane >> 5;
  load_A(kbeg);
  load_B(kbeg);
  store_lds();
  __syncthreads();
  for (int k0 = kbeg; k0 < kend; k0 += FBK) {
    const bool more = (k0 + FBK) < kend;  // uniform
#ifndef TRIS_EXP_NOLOAD
    if (more) {
      load_A(k0 + FBK);
      load_B(k0 + FBK);
    }
#endif
    if constexpr (PREC == 0) {
#pragma unroll
    for (int g = 0; g < FBK; g += 8) {
      float4 a[FM], b[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm * WM + i * 32 + li;
        if (A_RM) {
          a[i] = *reinterpret_cast<const float4*>(&As[row * LDK + g + kh * 4]);
        } else {
          const float* s = &As[(g + kh * 4) * (BM + 4) + row];
          a[i] = make_float4(s[0], s[BM + 4], s[2 * (BM + 4)], s[3 * (BM + 4)]);
        }
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = wn * WN + j * 32 + li;
        if (B_RM) {
          b[j] = *reinterpret_cast<const float4*>(&Bs[col * LDK + g + kh * 4]);
        } else {
          const float* s = &Bs[(g + kh * 4) * (BN + 4) + col];
          b[j] = make_float4(s[0], s[BN + 4], s[2 * (BN + 4)], s[3 * (BN + 4)]);
        }
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
    } else {
#pragma unroll
      for (int g = 0; g < FBK; g += 16) {
        // lane (li, kh) owns MFMA k-slots 8*kh + j  <->  k = g + 8*kh + j, for A and B alike
        Split8 sa[FM], sb[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int row = wm * WM + i * 32 + li;
          if (A_TR) {
            const char* pl0 = reinterpret_cast<const char*>(As);
            const int m16 = wm * WM + i * 32 + ((lane >> 4) & 1) * 16;
            sa[i].hi = tr_frag8(pl0, A_KS, g + 8 * kh, m16, lane);
            sa[i].mid = tr_frag8(pl0 + 32 * A_KS, A_KS, g + 8 * kh, m16, lane);
            if (NPLN == 3) sa[i].lo = tr_frag8(pl0 + 64 * A_KS, A_KS, g + 8 * kh, m16, lane);
          } else if (A_PL) {
            const char* s0 = reinterpret_cast<const char*>(As) + row * PLB + g * 2 + kh * 16;
            sa[i].hi = *reinterpret_cast<const bf16x8*>(s0);
            sa[i].mid = *reinterpret_cast<const bf16x8*>(s0 + BM * PLB);
            if (NPLN == 3) sa[i].lo = *reinterpret_cast<const bf16x8*>(s0 + 2 * BM * PLB);
          } else {
            const float* s0 = &As[(g + kh * 8) * (BM + 4) + row];
            sa[i] = split8(make_float4(s0[0], s0[BM + 4], s0[2 * (BM + 4)], s0[3 * (BM + 4)]),
                           make_float4(s0[4 * (BM + 4)], s0[5 * (BM + 4)], s0[6 * (BM + 4)], s0[7 * (BM + 4)]));
          }
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int col = wn * WN + j * 32 + li;
          if (B_TR) {
            const char* pl0 = reinterpret_cast<const char*>(Bs);
            const int n16 = wn * WN + j * 32 + ((lane >> 4) & 1) * 16;
            sb[j].hi = tr_frag8(pl0, B_KS, g + 8 * kh, n16, lane);
            sb[j].mid = tr_frag8(pl0 + 32 * B_KS, B_KS, g + 8 * kh, n16, lane);
            if (NPLN == 3) sb[j].lo = tr_frag8(pl0 + 64 * B_KS, B_KS, g + 8 * kh, n16, lane);
          } else if (B_PL) {
            const char* s0 = reinterpret_cast<const char*>(Bs) + col * PLB + g * 2 + kh * 16;
            sb[j].hi = *reinterpret_cast<const bf16x8*>(s0);
            sb[j].mid = *reinterpret_cast<const bf16x8*>(s0 + BN * PLB);
            if (NPLN == 3) sb[j].lo = *reinterpret_cast<const bf16x8*>(s0 + 2 * BN * PLB);
          } else {
            const float* s0 = &Bs[(g + kh * 8) * (BN + 4) + col];
            sb[j] = split8(make_float4(s0[0], s0[BN + 4], s0[2 * (BN + 4)], s0[3 * (BN + 4)]),
                           make_float4(s0[4 * (BN + 4)], s0[5 * (BN + 4)], s0[6 * (BN + 4)], s0[7 * (BN + 4)]));
          }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) {  // smallest terms first
            if (NPLN == 3) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].lo, sb[j].hi, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].hi, sb[j].lo, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].mid, sb[j].mid, acc[i][j], 0, 0, 0);
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].mid, sb[j].hi, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].hi, sb[j].mid, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].hi, sb[j].hi, acc[i][j], 0, 0, 0);
          }
      }
    }
    __syncthreads();
#ifndef TRIS_EXP_NOSTORE
    if (more) {
      store_lds();
      __syncthreads();
    }
#endif
  }

  // ---- epilogue (C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) ---------------------
  float st_s[FN], st_q[FN];  // fused BatchNorm statistics: per-column sum / sum of squares of this block's rows
#pragma unroll
  for (int j = 0; j < FN; ++j) st_s[j] = st_q[j] = 0.f;
  // Vector epilogue: in the accumulator layout a lane owns single floats of 16 different rows, so direct stores are 16
  // scalar instructions per 32x32 block, each touching two 128-byte row pieces (and the residual comes in the same way).
  // For the short-K products (1x1 convolutions of layer1/2: K = 64..256 against a [M, 256..512] output) that is most of the
  // kernel.  Each wave instead turns its block around in the idle staging LDS (32 x 36 floats, wave-private: program order
  // + a wave fence) and handles rows: 8 lanes x 16 bytes per row, 8 rows per instruction -- 4 loads/stores per block.
  constexpr int ELD = 36;
  constexpr bool EPI_LDS = (NWN * 32 * ELD <= A_SZ) && (NWN * 32 * ELD <= B_SZ);
  float4 vs_s[FN], vs_q[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) vs_s[j] = vs_q[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool vec_epi = EPI_LDS && p.vecC;  // uniform
  if (vec_epi) {
    float* stg = (wm == 0 ? As : Bs) + wn * 32 * ELD;
    const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * kh) * ELD + li] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int col = n0 + wn * WN + j * 32 + ec;
        float4 bc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (EPI == EPI_STD && p.bias_mode == 1 && col < p.N) bc = ld4(p.bias + col);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = m0 + wm * WM + i * 32 + er + 8 * t;
          float4 v = *reinterpret_cast<const float4*>(stg + (er + 8 * t) * ELD + ec);
          if (row < p.M && col < p.N) {  // N % 4 == 0: a vector never straddles the edge
            if (EPI == EPI_SLAB) {
              *reinterpret_cast<float4*>(p.C + ((long)blockIdx.z * p.M + row) * p.N + col) = v;
            } else {
              v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha;
              if (p.bias_mode == 1) { v.x += bc.x; v.y += bc.y; v.z += bc.z; v.w += bc.w; }
              else if (p.bias_mode == 2) { const float bb = p.bias[row]; v.x += bb; v.y += bb; v.z += bb; v.w += bb; }
              if (p.act == 1) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
              } else if (p.act == 2) {
                v.x = v.x / (1.0f + expf(-1.702f * v.x)); v.y = v.y / (1.0f + expf(-1.702f * v.y));
                v.z = v.z / (1.0f + expf(-1.702f * v.z)); v.w = v.w / (1.0f + expf(-1.702f * v.w));
              }
              if (p.resid) {
                const float4 rr = ld4(p.resid + (long)zb * p.sR + (long)row * p.ldr + col);
                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
              }
              *reinterpret_cast<float4*>(p.C + (long)zb * p.sC + (long)row * p.ldc + col) = v;
              vs_s[j].x += v.x; vs_s[j].y += v.y; vs_s[j].z += v.z; vs_s[j].w += v.w;
              vs_q[j].x += v.x * v.x; vs_q[j].y += v.y * v.y; vs_q[j].z += v.z * v.z; vs_q[j].w += v.w * v.w;
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
  } else {
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
            st_s[j] += v;
            st_q[j] += v * v;
          }
        }
      }
    }
  }
  if (EPI == EPI_STD && p.stat_part != nullptr) {
    // rows of one column live in the 2 lane halves (kh) [vector epilogue: the 8 row groups er] and the 2 waves along M:
    // shuffle, then LDS, then one fp64 partial row per block: part[tile_m][2][N]  (finished by bn_finalize_kernel)
    float* red = As;  // the k loop ended on a barrier: the staging buffer is free
    if (vec_epi) {
      __syncthreads();  // the other waves' epilogue blocks live in As / Bs
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        float4 a = vs_s[j], b = vs_q[j];
#pragma unroll
        for (int sh = 8; sh < 64; sh <<= 1) {
          a.x += __shfl_xor(a.x, sh, 64); a.y += __shfl_xor(a.y, sh, 64);
          a.z += __shfl_xor(a.z, sh, 64); a.w += __shfl_xor(a.w, sh, 64);
          b.x += __shfl_xor(b.x, sh, 64); b.y += __shfl_xor(b.y, sh, 64);
          b.z += __shfl_xor(b.z, sh, 64); b.w += __shfl_xor(b.w, sh, 64);
        }
        if (lane < 8) {
          const int cl = wn * WN + j * 32 + lane * 4;
          float* o = red + (wm * BN + cl) * 2;
          o[0] = a.x; o[1] = b.x; o[2] = a.y; o[3] = b.y; o[4] = a.z; o[5] = b.z; o[6] = a.w; o[7] = b.w;
        }
      }
    } else {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      float a = st_s[j] + __shfl_xor(st_s[j], 32, 64);
      float b = st_q[j] + __shfl_xor(st_q[j], 32, 64);
      if (kh == 0) {
        const int cl = wn * WN + j * 32 + li;
        red[(wm * BN + cl) * 2 + 0] = a;
        red[(wm * BN + cl) * 2 + 1] = b;
      }
    }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      const int tm = tile / p.tiles_n;
      double* o = p.stat_part + (long)tm * 2 * p.N;
      o[n0 + tid] = (double)red[tid * 2] + (double)red[(BN + tid) * 2];
      o[p.N + n0 + tid] = (double)red[tid * 2 + 1] + (double)red[(BN + tid) * 2 + 1];
    }
  }
}
