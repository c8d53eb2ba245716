// Fast path of the MFMA GEMM / implicit-GEMM core (included by gemm_conv.hip).
//
// BM x BN block, NW waves in an NWM x (NW/NWM) grid, 32x32 MFMA fragments, FBK-deep K tiles:
//   * *branch-free* tile loaders: rows/columns past the edge are clamped (their results are never stored) and out-of-image
//     im2col taps are loaded from a valid address and zeroed with a select, so the K loop has no divergent control flow
//     and the accumulators stay pinned in registers;
//   * PREC 0 (f32-input MFMA, FBK = 32): k-contiguous operands (row-major A, B^T, im2col) keep a row-major LDS image
//     [rows][32+4]: 16-byte global loads go to LDS as ds_write_b128, and a fragment read is ONE ds_read_b128 per lane =
//     4 k-values feeding 4 MFMAs.  Row stride 36 floats puts the 16 lanes of every ds_read_b128 service group on 16 distinct
//     16-byte slots (9*i mod 16 is a permutation) -> conflict-free.  The logical k order inside an MFMA is permuted
//     (lane-half kh, MFMA j  <->  k = 8g + 4kh + j); both operands use the same permutation, the sum is unchanged;
//     m-contiguous operands (A^T for wgrad, B for dgrad / NN) keep the k-major image and read 4 scalars;
//   * PREC 1 (split-bf16 "x3") and PREC 3 (two-piece fp16 "h2", below): operands are split when the tile is stored, the LDS
//     image is 16-bit planes -- row-major [plane][row][FBK + 8] for the k-contiguous kinds, k-major [plane][k][m] read with
//     ds_read_b64_tr_b16 for the m-contiguous ones;
//   * two loop structures:
//       NSTG = 1 ("classic"): one LDS buffer, register-staged prefetch, two barriers per K tile (compute | store);
//       NSTG = 2 ("pipelined", PREC >= 1): two LDS stages held in SEPARATE __shared__ objects (so the compiler may interleave
//         the split + LDS stores of tile k+1 with the fragment reads + MFMAs of tile k), ONE barrier per K tile, global
//         loads two tiles ahead in two register sets, branch-free steady state (a load inside a conditional block makes
//         the compiler's wait-count pass wait for the loads it has just issued).  FBK = 16 keeps two such blocks per CU.
//       NSTG = 4 ("LDS-DMA", PREC 4, row-major A x B^T only): operand planes are streamed into LDS by global_load_lds_dwordx4 --
//         no staging registers, no ds_write, no VALU on the operands; see "LDS-DMA loop" in the kernel.
//   * epilogue: bias / activation / residual / BN statistics, stored as 16-byte rows after an in-LDS turn of each wave's
//     accumulator block (scalar fallback for unaligned or N % 4 != 0 outputs).
// Preconditions (checked on the host, otherwise the generic kernel runs): K % 32 == 0 per k-slice, 16-byte aligned
// operands, M % 4 == 0 / N % 4 == 0 for the m-/n-contiguous kinds, gathered channels % 32 == 0 for im2col.
#pragma once

// ---- split-bf16 ("x3") arithmetic -----------------------------------------------------------------------------------------
// An fp32 value is EXACTLY the sum of three bf16 pieces (round-to-nearest residual splitting: 8 + 8 + 8 significand
// bits).  a*b = sum of the 9 piece products; the 6 with combined weight >= 2^-24 are kept, each is exact in fp32
// (8 x 8 bits) and is accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Result: fp32-class accuracy at 6 bf16 MFMAs per
// 16-deep step (6 x 32 cycles) instead of 8 f32 MFMAs (8 x 64 cycles).
#include "x3_split.h"
#include "amax.h"

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

// LDS bytes of one operand stage (host + device): see the layout notes in the kernel
constexpr int gf_stage_floats(int BMN, int FBK, int PREC, bool row_major) {
  const int NPLN = (PREC == 3 || PREC == 4) ? 2 : 3;
  return PREC >= 1 ? (row_major ? BMN * (2 * FBK + 16) * NPLN / 4 : NPLN * FBK * (2 * BMN + 64) / 4)
                   : (row_major ? BMN * (FBK + 4) : FBK * (BMN + 4));
}
constexpr int gf_halo_floats(int HS, int PREC) { return HS * ((PREC == 3 || PREC == 4) ? 16 : 24); }   // HS pixel slots x 3 (h2: 2) planes x 16 channels, 16-bit (no padding)
constexpr int gf_min_waves_per_simd(int BM, int BN, int PREC, int NW, int FBK, int NSTG, bool a_rm, bool b_rm, int HS = 0) {
  // (the classic kernels: two co-resident 8-wave blocks per CU, as tuned in round 1; h2's 4-wave 128 x 64 / 64 x 64 kernels fit
  // 128 registers with both accumulator sets and keep four waves per SIMD)
  if (NSTG == 4) {   // LDS-DMA loop: two stages of (BM + BN) rows x 128 bytes; 32 accumulator registers per 32 x 32 block pair
    const int blocks = (2 * (BM + BN) * 128 * 2 <= 160 * 1024) ? 2 : 1;
    const int acc = (BM * BN / (NW * 1024)) * 32;
    const int w = blocks * NW / 4;
    return acc >= 256 ? 1 : (acc >= 128 && w > 2) ? 2 : (w < 1 ? 1 : w);
  }
  if (NSTG == 1) return (NW == 8 || ((PREC == 3 || PREC == 4) && NW == 4 && BM * BN <= 128 * 64)) ? 4 : 2;
  const int lds = HS > 0 ? 4 * (gf_halo_floats(HS, PREC) + NSTG * gf_stage_floats(BN, FBK, PREC, b_rm))
                         : 4 * NSTG * (gf_stage_floats(BM, FBK, PREC, a_rm) + gf_stage_floats(BN, FBK, PREC, b_rm));
  // (h2, 256-row tiles: 128 accumulator registers per wave -- one block per CU whatever the LDS would allow)
  const int blocks = (lds * 2 <= 160 * 1024 && !((PREC == 3 || PREC == 4) && BM == 256)) ? 2 : 1;
  const int w = blocks * NW / 4;
  if (HS > 0) {   // (direct 3x3: the window registers need more than 128 VGPRs; h2 with four 32 x 32 blocks per wave -- eight
    // accumulator blocks -- more than 256: one wave per SIMD)
    if ((PREC == 3 || PREC == 4) && NW == 4 && BM * BN >= 128 * 128) return 1;
    return w > 2 ? 2 : (w < 1 ? 1 : w);
  }
  return w < 1 ? 1 : w;
}

// NW = waves per workgroup; NWM x (NW / NWM) wave grid (classic kernels: 2 x NW/2)
// AK == A_HALO (3x3 convolution, stride 1, x3 arithmetic, FBK 16, NSTG 2, HS > 0): see "direct 3x3" below
template <int BM, int BN, int AK, int BKIND, int EPI, int PREC, int NW = 4, int FBK = 32, int NSTG = 1, int NWM = 2, int HS = 0>
__global__ __launch_bounds__(NW * 64, gf_min_waves_per_simd(BM, BN, PREC, NW, FBK, NSTG, AK != A_COLK,
                                                             BKIND == B_NK, HS))
void gemm_fast_kernel(GemmParams p) {
  constexpr int NTHR = NW * 64, NWN = NW / NWM;
  // ---- direct 3x3 convolution (A_HALO) ---------------------------------------------------------------------------------------
  // The implicit GEMM above (A_IM2COL) re-gathers and RE-SPLITS every input value nine times, once per tap, and the split
  // (5.5 VALU operations per element) + LDS store of the A tile is what bounds the x3 kernels (tools/probes/x3_pipe_probe).
  // Here the block's input window -- its BM output pixels plus the one-pixel border -- is split ONCE per 16-channel chunk into
  // an LDS image [plane][channel half][pixel slot][8 ch]; the nine taps of that chunk then read their A fragments from the same
  // image at a per-tap slot offset ((ty-1) * pitch + (tx-1)): the zero padding of the convolution is part of the image, so
  // the tap loop has no bounds logic at all.  The K order is (channel chunk, tap, channel); B (weights) is staged per
  // (chunk, tap) exactly as in the pipelined loop.  Two window shapes (p.hmode):
  //   2: 2-D patches of (BM/16) x 16 pixels (W % 16 == 0, H % (BM/16) == 0), slot pitch 18: 1.27x input re-read at BM 256;
  //   1: BM consecutive pixels of the flattened [B, H, W] grid, held in PADDED coordinates (pitch W + 2, one zero row between
  //      images): any H, W; the window is BM + ~2 W slots, so this is for the narrow late stages (W <= 40).
  constexpr bool HALO = (AK == A_HALO);
  static_assert(!HALO || ((PREC == 1 || PREC == 3 || PREC == 4) && FBK == 16 && NSTG == 2 && HS % 8 == 4), "A_HALO: x3 / h2, 16-channel chunks, pipelined B");
  constexpr int KL = FBK / 4;          // 16-byte pieces per row of a row-major tile
  constexpr int RPASS = NTHR / KL;     // rows of a row-major tile covered per pass
  constexpr int LDK = FBK + 4;
  constexpr bool A_RM = (AK != A_COLK);
  constexpr bool B_RM = (BKIND == B_NK);
  static_assert(PREC == 0 || PREC == 1 || PREC == 3 || PREC == 4, "arithmetic: 0 = f32 MFMA, 1 = x3, 3 = h2, 4 = h2 on pre-split planes");
  static_assert(PREC >= 1 || (FBK == 32 && NSTG == 1), "the f32-MFMA path exists as the classic 32-deep loop only");
  static_assert(NW % NWM == 0 && BM % (32 * NWM) == 0 && BN % (32 * NWN) == 0, "wave grid does not tile the block");
  constexpr int WM = BM / NWM, WN = BN / NWN;
  constexpr int FM = WM / 32, FN = WN / 32;
  // float4 per thread per tile; B_PART: a B tile smaller than one piece per thread (32-wide tiles of the direct convolution):
  // the first BN * FBK / 4 threads stage it
  constexpr bool B_PART = (BN * FBK < 4 * NTHR);
  constexpr int PA = BM * FBK / (4 * NTHR), PB = B_PART ? 1 : BN * FBK / (4 * NTHR);
  static_assert(PA >= 1 && PA * 4 * NTHR == BM * FBK && (B_PART ? HALO : PB * 4 * NTHR == BN * FBK), "tile / thread count mismatch");
  const bool b_act = !B_PART || threadIdx.x < BN * FBK / 4;
  // x3 / x2: every operand is split into its bf16 pieces ONCE, when the tile is stored.  Row-major kinds: three planes
  // [plane][row][FBK + 8 pad] (row stride PLB = 2 FBK + 16 bytes: 80 -> 5, 48 -> 3 sixteen-byte slots, both odd, so the 16
  // lanes of a ds_read_b128 service group fall on 16 distinct slots).
  constexpr bool A_PL = (PREC >= 1), B_PL = (PREC >= 1);
  constexpr bool H2 = (PREC == 3 || PREC == 4);
  // PREC 4: both operands ARRIVE as fp16 piece planes ("P8": per 8 consecutive elements of the contiguous dimension 16 bytes of hi
  // pieces, then 16 bytes of pre-scaled lo pieces -- the byte geometry of the fp32 tensor, include/tris_hip.h tris_h2_planes_f32),
  // written once by the pass that produced the tensor.  The loaders fetch the same 16-byte pieces at the same addresses as in
  // PREC 3; a piece is eight fp16 values of ONE plane and goes to LDS as it is (one ds_write_b128, no VALU).
  constexpr bool PLN = (PREC == 4);
  constexpr int NPLN = H2 ? 2 : 3;  // 16-bit planes per operand (h2: fp16 pieces)
  // m-/n-contiguous operands: 16-byte loads along the contiguous dimension, split, 8-byte LDS stores into k-major planes
  // [plane][k][m], and the MFMA fragments (8 consecutive k per lane) are gathered by the LDS transpose read
  // ds_read_b64_tr_b16: per 16-lane group, lane i points at the 8-byte piece [k0 + i/4][m0 + 4(i%4) ..+3] and receives
  // [k0..k0+3][m0 + i] (semantics established with tools/probes/tr_read_probe.hip).  Plane row stride = 2*BM + 64 bytes:
  // the four k rows of a group and the two groups of a 32-lane half fall on disjoint banks.
  constexpr bool A_TR = (PREC >= 1) && !A_RM, B_TR = (PREC >= 1) && !B_RM;
  constexpr int A_KS = 2 * BM + 64, B_KS = 2 * BN + 64;  // bytes per k row of a k-major plane
  constexpr int PLB = 2 * FBK + 16;                      // bytes per row of one row-major bf16 plane
  constexpr int A_SZ = HALO ? gf_halo_floats(HS, PREC) : gf_stage_floats(BM, FBK, PREC, A_RM);
  constexpr int B_SZ = gf_stage_floats(BN, FBK, PREC, B_RM);
  // LDS-DMA loop (NSTG 4): ALL of the kernel's LDS is ONE object (with a second one, however small, the compiler waits for
  // vmcnt(0) before the first fragment read of every k step): two stages of [A rows | B rows] x 128 bytes
  constexpr bool GLDS = (NSTG == 4);
  static_assert(!GLDS || (PLN && AK == A_ROWK && BKIND == B_NK && FBK == 32 && !HALO), "LDS-DMA loop: operand planes, row-major A x B^T");
  constexpr int G_ST = (BM + BN) * 128;                      // bytes per stage
  constexpr int AS_ALL = GLDS ? 2 * G_ST / 4 : A_SZ;         // floats the epilogue may use through `As`
  __shared__ __attribute__((aligned(1024))) float As[AS_ALL];
  __shared__ __attribute__((aligned(16))) float Bs_[GLDS ? 4 : B_SZ];
  __shared__ __attribute__((aligned(16))) float As1[(NSTG == 2 && !HALO) ? A_SZ : 4];   // second stage (pipelined loop): separate objects
  __shared__ __attribute__((aligned(16))) float Bs1_[NSTG == 2 ? B_SZ : 4];
  float* const Bs = Bs_;
  float* const Bs1 = Bs1_;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  // Workgroup -> (tile, z) mapping.  The dispatcher places consecutive workgroups on consecutive XCDs (8 private L2s).  For the
  // 3x3 weight gradient the N tiles of one k slice are the nine TAPS of the same pixel window: they gather the same input rows
  // (shifted by a pixel / a row) and read the same dY rows -- spread over 8 XCDs every L2 fetches its own copy (measured:
  // 1.6 GB of fabric traffic per layer1 launch for 157 MB of operands).  With xcd_remap (host: split-K slices % 8 == 0) all
  // tiles of a slice run on ONE XCD, so that L2 serves eight of the nine reads.  Pure placement: any mapping is correct.
  int bid_x = blockIdx.x, bid_z = blockIdx.z;
  if (p.xcd_remap == 2) {
    // every XCD takes a CONTIGUOUS range of tiles: the column tiles of one row block -- same A rows -- run on one XCD and share its
    // L2 instead of eight L2s fetching a copy each (the M-large / N-small products of the trunk: +7 ... +30 %, tools/probes/
    // h2_plane_probe.hip; square products lose -- the tuner times both).  Bijective for any grid size.
    const int T = gridDim.x, q = T >> 3, r = T & 7, x = bid_x & 7;
    bid_x = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid_x >> 3);
  } else
  if (p.xcd_remap) {
    const int T = gridDim.x;
    const int L = bid_x + T * bid_z;          // dispatch order (gridDim.y == 1)
    const int xcd = L & 7, j = L >> 3;        // j-th workgroup of its XCD
    bid_z = xcd + 8 * (j / T);                // bijective for gridDim.z % 8 == 0
    bid_x = j % T;
  }
  const int tile = bid_x;
  const int m0 = (tile / p.tiles_n) * BM;
  const int n0 = (tile % p.tiles_n) * BN;
  if (p.m_limit != nullptr && m0 >= *p.m_limit) return;   // (uniform: the whole workgroup leaves before its first barrier)
  const int zb = bid_z / p.splitk;
  const int zs = bid_z % p.splitk;
  const int kbeg = zs * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const float* __restrict__ A = p.A + (long)zb * p.sA;
  const float* __restrict__ Bp = p.B + (long)zb * p.sB;

  // Row handled by this thread's KL-lane set in the row-major kinds.  The split pieces go to LDS with ds_write_b64, which is
  // serviced in groups of 16 CONTIGUOUS lanes against a 32-bank (128-byte) modulus.
  //   FBK 32: two 8-lane sets = two rows of 64 bytes per group; with 80-byte plane rows, rows r and r+1 overlap on four banks
  //     (2-way), rows r and r+4 are exactly 16 banks apart -> within every 8 rows the sets visit rows 0,4,1,5,2,6,3,7;
  //   FBK 16: four 4-lane sets = four rows of 32 bytes per group; with 48-byte plane rows, rows {0,2,4,6} (and {1,3,5,7}) start
  //     at byte offsets {0,96,64,32} (+48) mod 128 -> four disjoint 32-byte windows -> a group takes the even or the odd rows.
  // Same layout, same global coalescing (one contiguous row piece per lane set), conflict-free stores.
  const int trow = KL == 8 ? (((tid >> 3) & ~7) | (((tid >> 3) & 1) << 2) | ((tid >> 4) & 3))
                           : (((tid >> 5) << 3) | (2 * ((tid >> 2) & 3) + ((tid >> 4) & 1)));
  const int kq4 = (tid % KL) * 4;   // first k of this thread's 16-byte piece (row-major kinds)
  // ---- loader state ---------------------------------------------------------------------------------------------
  long a_off[PA];  // ROWK: row offset; IM2COL: unused
  int a_b[PA], a_iy0[PA], a_ix0[PA];
  if (AK == A_ROWK) {
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      int m = min(m0 + trow + q * RPASS, p.M - 1);
      a_off[q] = (long)m * p.lda + kq4;
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
      b_off[q] = (long)n * p.ldb + kq4;
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

  int bw_b[PB], bw_oy[PB], bw_ox[PB];  // B_KN_IM2COL / B_KN_DGRAD: running coordinates of this thread's k rows
  // PREC 3 ("h2"): power-of-two operand scales (gemm_params.h): from the bit pattern of the tensor's largest magnitude -- or of an
  // UPPER BOUND of it -- in device memory, or from the host; the epilogue takes them out again (exact: powers of two)
  float h2a = 1.f, h2b = 1.f;
  if constexpr (H2) {
    h2a = p.h2_amaxA ? h2_scale_from_bits(h2_amax_of(p.h2_amaxA, lane)) : (p.h2_sA != 0.f ? p.h2_sA : 1.f);
    h2b = p.h2_amaxB ? h2_scale_from_bits(h2_amax_of(p.h2_amaxB, lane)) : (p.h2_sB != 0.f ? p.h2_sB : 1.f);
  }
  const float h2inv = 1.0f / (h2a * h2b);
  auto splitA = [&](const float4& v) { if constexpr (H2) return split4h(v, h2a); else return split4(v); };
  auto splitB = [&](const float4& v) { if constexpr (H2) return split4h(v, h2b); else return split4(v); };
  // ---- direct 3x3: window geometry (see the head of the kernel) -------------------------------------------------------------
  constexpr int HP = HALO ? (HS * 4 + NTHR - 1) / NTHR : 1;   // 16-byte pieces (4 channels of one slot) per thread per chunk
  int h_pitch = 0, h_b = 0, h_y0 = 0, h_x0 = 0;
  int h_rowslot[FM];   // slot of the CENTRE tap of this lane's row in fragment i
  int h_pix[HP];       // input pixel (b*H + y)*W + x held by this thread's q-th slot, -1 = padding
  if constexpr (HALO) {
    const int tm = tile / p.tiles_n;
    if (p.hmode == 2) {
      constexpr int TH = BM / 16;
      const int txn = p.gW / 16, tpi = txn * (p.gH / TH);
      h_b = tm / tpi;
      const int r = tm - h_b * tpi;
      h_y0 = (r / txn) * TH;
      h_x0 = (r - (r / txn) * txn) * 16;
      h_pitch = 18;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int rl = wm * WM + i * 32 + (lane & 31);
        h_rowslot[i] = ((rl >> 4) + 1) * 18 + (rl & 15) + 1;
      }
#pragma unroll
      for (int q = 0; q < HP; ++q) {
        const int sl = (tid + q * NTHR) >> 2;
        const int sy = sl / 18, sx = sl - sy * 18;
        const int iy = h_y0 + sy - 1, ix = h_x0 + sx - 1;
        const bool ok = sl < HS && (unsigned)iy < (unsigned)p.gH && (unsigned)ix < (unsigned)p.gW;
        h_pix[q] = ok ? (h_b * p.gH + iy) * p.gW + ix : -1;
      }
    } else {
      h_pitch = p.gW + 2;
      const int hw = p.gH * p.gW, pp = (p.gH + 2) * h_pitch;
      auto padded = [&](int m) {
        const int b = m / hw, r = m - b * hw;
        const int y = r / p.gW, x = r - y * p.gW;
        return (b * (p.gH + 2) + y + 1) * h_pitch + x + 1;
      };
      const int base = padded(m0) - h_pitch - 1;
#pragma unroll
      for (int i = 0; i < FM; ++i) h_rowslot[i] = padded(min(m0 + wm * WM + i * 32 + (lane & 31), p.M - 1)) - base;
#pragma unroll
      for (int q = 0; q < HP; ++q) {
        const int sl = (tid + q * NTHR) >> 2;
        const int qd = base + sl;
        const int b = qd / pp, r = qd - b * pp;
        const int y = r / h_pitch, x = r - y * h_pitch;
        const bool ok = sl < HS && b < p.gB && y >= 1 && y <= p.gH && x >= 1 && x <= p.gW;
        h_pix[q] = ok ? (b * p.gH + y - 1) * p.gW + x - 1 : -1;
      }
    }
  }
  // tile-local row -> row of the output matrix (rows >= p.M are not stored)
  auto grow = [&](int rl) -> int {
    if (HALO && p.hmode == 2) return (h_b * p.gH + h_y0 + (rl >> 4)) * p.gW + h_x0 + (rl & 15);
    return m0 + rl;
  };
  // p.in_mean != NULL: the gathered tensor is the RAW input of a BatchNorm + ReLU (x, not y = relu((x - mean) * invstd * gamma
  // + beta)): y is formed here, once per window slot, with tris_bn_apply_f32's own expression -- y never exists in HBM.  A thread
  // always stages the same 4-channel group of a chunk ((tid + q NTHR) & 3 == tid & 3), so its constants are three float4.
  float4 h_mu = make_float4(0.f, 0.f, 0.f, 0.f), h_sc = h_mu, h_be = h_mu;
  auto load_halo = [&](float4 (&hr)[HP], int c0) {
    if (p.in_mean != nullptr) {   // uniform
      const int c = c0 + (tid & 3) * 4;
      const float4 is = ld4(p.in_invstd + c), ga = ld4(p.in_gamma + c);
      h_mu = ld4(p.in_mean + c);
      h_be = ld4(p.in_beta + c);
      h_sc = make_float4(is.x * ga.x, is.y * ga.y, is.z * ga.z, is.w * ga.w);
    }
#pragma unroll
    for (int q = 0; q < HP; ++q) {
      const bool ok = h_pix[q] >= 0;
      const float4 v = ld4(A + (ok ? (long)h_pix[q] * p.gC + c0 + ((tid + q * NTHR) & 3) * 4 : 0));
      hr[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_halo = [&](float* Ad, const float4 (&hr)[HP]) {
#pragma unroll
    for (int q = 0; q < HP; ++q) {
      const int j = tid + q * NTHR;
      if (HP * NTHR == HS * 4 || j < HS * 4) {
        float4 v = hr[q];
        if constexpr (PLN) {
          // a slot's 16 channels are 64 bytes [hi 0-7 | lo 0-7 | hi 8-15 | lo 8-15]: piece j & 3 -> plane (j & 1), channel half (j >> 1) & 1
          char* d = reinterpret_cast<char*>(Ad) + (j >> 2) * 16 + ((j >> 1) & 1) * (HS * 16) + (j & 1) * (HS * 32);
          *reinterpret_cast<float4*>(d) = v;
          continue;
        }
        if (p.in_mean != nullptr && h_pix[q] >= 0) {   // (padding slots stay zero: the convolution pads y, not x)
          v.x = fmaxf((v.x - h_mu.x) * h_sc.x + h_be.x, 0.f);
          v.y = fmaxf((v.y - h_mu.y) * h_sc.y + h_be.y, 0.f);
          v.z = fmaxf((v.z - h_mu.z) * h_sc.z + h_be.z, 0.f);
          v.w = fmaxf((v.w - h_mu.w) * h_sc.w + h_be.w, 0.f);
        }
        const Split4 sp = splitA(v);
        // image [plane][channel half kh][slot][8 ch]: a fragment read is 16 lanes x 16 contiguous bytes (conflict-free without
        // padding); the four 8-byte pieces of a slot go to two 64-byte windows HS*16 bytes apart (HS % 8 == 4: disjoint banks)
        char* d = reinterpret_cast<char*>(Ad) + (j >> 2) * 16 + ((j >> 1) & 1) * (HS * 16) + (j & 1) * 8;
        *reinterpret_cast<uint2*>(d) = sp.hi;
        *reinterpret_cast<uint2*>(d + HS * 32) = sp.mid;
        if (NPLN == 3) *reinterpret_cast<uint2*>(d + 2 * HS * 32) = sp.lo;
      }
    }
  };
  // B tile of (tap, 16 channels from c0) -- B_NK: weights [n][tap][c]; B_KN_DGRAD: k = (tap, co) reads W[co][8 - tap][n]
  auto load_Bh = [&](float4 (&rb)[PB], int tap, int c0) {
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      if (BKIND == B_NK) rb[q] = ld4(Bp + b_off[q] + tap * p.gC + c0);
      else {
        const int co = min(c0 + tid / BF4 + q * BRPP, p.wCout - 1);
        rb[q] = ld4(Bp + ((long)co * 9 + (8 - tap)) * p.wCin + b_nc);
      }
    }
  };

  // Tiles must be requested in order, FBK apart, starting at kbeg (the im2col / dgrad coordinates are carried from tile to tile)
  auto load_A = [&](float4 (&ra)[PA], int k0) {
    if (AK == A_ROWK) {
#pragma unroll
      for (int q = 0; q < PA; ++q) ra[q] = ld4(A + a_off[q] + k0);
    } else if (AK == A_IM2COL) {
      const int tap = k0 / p.gC;  // a whole FBK-wide k tile lies inside one tap (gC % 32 == 0)
      const int ci = k0 - tap * p.gC + kq4;
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        const int iy = a_iy0[q] + ky, ix = a_ix0[q] + kx;
        const bool inb = (unsigned)iy < (unsigned)p.gH && (unsigned)ix < (unsigned)p.gW;
        const long off = inb ? ((long)(a_b[q] * p.gH + iy) * p.gW + ix) * p.gC + ci : 0;
        float4 v = ld4(A + off);
        ra[q] = inb ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {  // A_COLK: A[k*lda + m]
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        const int kk = k0 + tid / AF4 + q * ARPP;
        if (p.Kv > 0) {   // (uniform) K rounded up: rows past the last valid one read as zeros
          const float4 v = ld4(A + (long)min(kk, p.Kv - 1) * p.lda + a_mc);
          ra[q] = kk < p.Kv ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
          ra[q] = ld4(A + (long)kk * p.lda + a_mc);
        }
      }
    }
  };

  auto load_B = [&](float4 (&rb)[PB], int k0) {
    if (BKIND == B_NK) {
#pragma unroll
      for (int q = 0; q < PB; ++q) rb[q] = ld4(Bp + b_off[q] + k0);
    } else if (BKIND == B_KN) {
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        const int kk = k0 + tid / BF4 + q * BRPP;
        if (p.Kv > 0) {
          const float4 v = ld4(Bp + (long)min(kk, p.Kv - 1) * p.ldb + b_nc);
          rb[q] = kk < p.Kv ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
          rb[q] = ld4(Bp + (long)kk * p.ldb + b_nc);
        }
      }
    } else if (BKIND == B_KN_DGRAD) {  // k = tap'*Cout + co ; B[k][ci] = W[co][8 - tap'][ci]
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        if (k0 == kbeg) {  // (tap', co) of this thread's rows, carried from tile to tile like the im2col coordinates below
          const int kk = k0 + tid / BF4 + q * BRPP;
          bw_b[q] = kk / p.wCout;
          bw_oy[q] = kk - bw_b[q] * p.wCout;
        }
        rb[q] = ld4(Bp + ((long)bw_oy[q] * 9 + (8 - min(bw_b[q], 8))) * p.wCin + b_nc);   // (min: surplus prefetches past the last tap)
        bw_oy[q] += FBK;
        while (bw_oy[q] >= p.wCout) { bw_oy[q] -= p.wCout; ++bw_b[q]; }
      }
    } else {  // B_KN_IM2COL: k = output pixel, column = (tap, ci) of the gathered input
      // the pixel coordinates of this thread's PB rows are carried from tile to tile (tiles are requested in order, FBK
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
        const bool inb = (unsigned)iy < (unsigned)p.gH && (unsigned)ix < (unsigned)p.gW && bw_b[q] < p.gB;
        const long off = inb ? ((long)(bw_b[q] * p.gH + iy) * p.gW + ix) * p.gC + bj_ci : 0;
        float4 v = ld4(Bp + off);
        rb[q] = inb ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        bw_ox[q] += FBK;  // next tile
        while (bw_ox[q] >= p.gWo) { bw_ox[q] -= p.gWo; if (++bw_oy[q] == p.gHo) { bw_oy[q] = 0; ++bw_b[q]; } }
      }
    }
  };

  // ---- LDS stores: one 16-byte piece (chunk) of a tile at a time, so that the pipelined loop can spread them -----------
  auto store_A = [&](float* Ad, const float4& v, int q) {
    if constexpr (PLN) {   // piece j of a row: plane (j & 1), elements 8 (j >> 1) .. + 7 of the contiguous dimension
      if (A_TR) {
        const int j = tid % AF4;
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(Ad) + (j & 1) * (FBK * A_KS) + (tid / AF4 + q * ARPP) * A_KS + (j >> 1) * 16) = v;
      } else {
        const int j = tid % KL;
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(Ad) + (j & 1) * (BM * PLB) + (trow + q * RPASS) * PLB + (j >> 1) * 16) = v;
      }
    } else
    if (A_TR) {
      const Split4 sp = splitA(v);
      char* d = reinterpret_cast<char*>(Ad) + (tid / AF4 + q * ARPP) * A_KS + (tid % AF4) * 8;
      *reinterpret_cast<uint2*>(d) = sp.hi;
      *reinterpret_cast<uint2*>(d + FBK * A_KS) = sp.mid;
      if (NPLN == 3) *reinterpret_cast<uint2*>(d + 2 * FBK * A_KS) = sp.lo;
    } else if (A_PL) {
      const Split4 sp = splitA(v);
      char* d = reinterpret_cast<char*>(Ad) + (trow + q * RPASS) * PLB + (tid % KL) * 8;
      *reinterpret_cast<uint2*>(d) = sp.hi;
      *reinterpret_cast<uint2*>(d + BM * PLB) = sp.mid;
      if (NPLN == 3) *reinterpret_cast<uint2*>(d + 2 * BM * PLB) = sp.lo;
    } else if (A_RM) {
      *reinterpret_cast<float4*>(&Ad[(trow + q * RPASS) * LDK + kq4]) = v;
    } else {
      *reinterpret_cast<float4*>(&Ad[(tid / AF4 + q * ARPP) * (BM + 4) + (tid % AF4) * 4]) = v;
    }
  };
  auto store_B = [&](float* Bd, const float4& v, int q) {
    if constexpr (PLN) {
      if (B_TR) {
        const int j = tid % BF4;
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(Bd) + (j & 1) * (FBK * B_KS) + (tid / BF4 + q * BRPP) * B_KS + (j >> 1) * 16) = v;
      } else {
        const int j = tid % KL;
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(Bd) + (j & 1) * (BN * PLB) + (trow + q * RPASS) * PLB + (j >> 1) * 16) = v;
      }
    } else
    if (B_TR) {
      const Split4 sp = splitB(v);
      char* d = reinterpret_cast<char*>(Bd) + (tid / BF4 + q * BRPP) * B_KS + (tid % BF4) * 8;
      *reinterpret_cast<uint2*>(d) = sp.hi;
      *reinterpret_cast<uint2*>(d + FBK * B_KS) = sp.mid;
      if (NPLN == 3) *reinterpret_cast<uint2*>(d + 2 * FBK * B_KS) = sp.lo;
    } else if (B_PL) {
      const Split4 sp = splitB(v);
      char* d = reinterpret_cast<char*>(Bd) + (trow + q * RPASS) * PLB + (tid % KL) * 8;
      *reinterpret_cast<uint2*>(d) = sp.hi;
      *reinterpret_cast<uint2*>(d + BN * PLB) = sp.mid;
      if (NPLN == 3) *reinterpret_cast<uint2*>(d + 2 * BN * PLB) = sp.lo;
    } else if (B_RM) {
      *reinterpret_cast<float4*>(&Bd[(trow + q * RPASS) * LDK + kq4]) = v;
    } else {
      *reinterpret_cast<float4*>(&Bd[(tid / BF4 + q * BRPP) * (BN + 4) + (tid % BF4) * 4]) = v;
    }
  };
  f32x16 acc[FM][FN];
  // h2: a second accumulator set for the two cross products (hi x lo', lo' x hi), whose lo' pieces carry a factor 2^11 (x3_split.h)
  f32x16 acx[H2 ? FM : 1][H2 ? FN : 1];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[i][j][r] = 0.f;
        if constexpr (H2) acx[i][j][r] = 0.f;
      }

  const int li = lane & 31, kh = lane >> 5;

  // fragments of k group g (16 k values: lane (li, kh) owns MFMA k-slots 8*kh + j  <->  k = 16 g + 8*kh + j, for A and B alike)
  auto frag_A = [&](const float* Ac, int g, int i) {
    Split8 s;
    const int row = wm * WM + i * 32 + li;
    if (A_TR) {
      const char* pl0 = reinterpret_cast<const char*>(Ac);
      const int m16 = wm * WM + i * 32 + ((lane >> 4) & 1) * 16;
      s.hi = tr_frag8(pl0, A_KS, g * 16 + 8 * kh, m16, lane);
      s.mid = tr_frag8(pl0 + FBK * A_KS, A_KS, g * 16 + 8 * kh, m16, lane);
      if (NPLN == 3) s.lo = tr_frag8(pl0 + 2 * FBK * A_KS, A_KS, g * 16 + 8 * kh, m16, lane);
    } else {
      const char* s0 = reinterpret_cast<const char*>(Ac) + row * PLB + g * 32 + kh * 16;
      s.hi = *reinterpret_cast<const bf16x8*>(s0);
      s.mid = *reinterpret_cast<const bf16x8*>(s0 + BM * PLB);
      if (NPLN == 3) s.lo = *reinterpret_cast<const bf16x8*>(s0 + 2 * BM * PLB);
    }
    return s;
  };
  auto frag_B = [&](const float* Bc, int g, int j) {
    Split8 s;
    const int col = wn * WN + j * 32 + li;
    if (B_TR) {
      const char* pl0 = reinterpret_cast<const char*>(Bc);
      const int n16 = wn * WN + j * 32 + ((lane >> 4) & 1) * 16;
      s.hi = tr_frag8(pl0, B_KS, g * 16 + 8 * kh, n16, lane);
      s.mid = tr_frag8(pl0 + FBK * B_KS, B_KS, g * 16 + 8 * kh, n16, lane);
      if (NPLN == 3) s.lo = tr_frag8(pl0 + 2 * FBK * B_KS, B_KS, g * 16 + 8 * kh, n16, lane);
    } else {
      const char* s0 = reinterpret_cast<const char*>(Bc) + col * PLB + g * 32 + kh * 16;
      s.hi = *reinterpret_cast<const bf16x8*>(s0);
      s.mid = *reinterpret_cast<const bf16x8*>(s0 + BN * PLB);
      if (NPLN == 3) s.lo = *reinterpret_cast<const bf16x8*>(s0 + 2 * BN * PLB);
    }
    return s;
  };
  auto mfma_x = [&](const Split8& a, const Split8& b, int i, int j) {
    f32x16& c = acc[i][j];
    if constexpr (H2) {   // two fp16 pieces: hi x hi into acc, the two cross products (x 2^11) into acx -- independent chains
      f32x16& cx = acx[i][j];
      const f16x8 ah = __builtin_bit_cast(f16x8, a.hi), al = __builtin_bit_cast(f16x8, a.mid);
      const f16x8 bh = __builtin_bit_cast(f16x8, b.hi), bl = __builtin_bit_cast(f16x8, b.mid);
      cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, cx, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
      cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, cx, 0, 0, 0);
    } else {              // three bf16 pieces, smallest terms first
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.mid, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.hi, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.mid, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, c, 0, 0, 0);
    }
  };

  if constexpr (HALO) {
    // ---- direct 3x3 loop: per 16-channel chunk ONE window image, nine pipelined (tap) steps over it --------------------------
    const int cbeg = zs * p.kchunk;                       // channel range of this k slice (kchunk: channels, multiple of 16)
    const int nch = (min(p.gC, cbeg + p.kchunk) - cbeg) / 16;
    auto frag_H = [&](int i, int tapoff) {
      Split8 s;
      const char* s0 = reinterpret_cast<const char*>(As) + (h_rowslot[i] + tapoff) * 16 + kh * (HS * 16);
      s.hi = *reinterpret_cast<const bf16x8*>(s0);
      s.mid = *reinterpret_cast<const bf16x8*>(s0 + HS * 32);
      if (NPLN == 3) s.lo = *reinterpret_cast<const bf16x8*>(s0 + 2 * HS * 32);
      return s;
    };
    // (chunk, tap) past the end are clamped to the last tile: surplus tiles go to a stage that is never read
    auto tile_B = [&](float4 (&rb)[PB], int c, int tap) {
      if (tap >= 9) { tap -= 9; ++c; }
      if (c >= nch) { c = nch - 1; tap = 8; }
      load_Bh(rb, tap, cbeg + c * 16);
    };
    auto step_h = [&](const float* Bc, float* Bn, const float4 (&rb)[PB], int tap) {
      const int tapoff = (tap / 3 - 1) * h_pitch + (tap % 3 - 1);
      Split8 sa[FM], sb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) sa[i] = frag_H(i, tapoff);
#pragma unroll
      for (int j = 0; j < FN; ++j) sb[j] = frag_B(Bc, 0, j);
      int done = 0;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          mfma_x(sa[i], sb[j], i, j);
          const int want = ((i * FN + j + 1) * PB) / (FM * FN);
#pragma unroll
          for (int c = 0; c < PB; ++c)
            if (c >= done && c < want && b_act) store_B(Bn, rb[c], c);
          done = want;
        }
    };
    float4 hr[HP], rb0[PB], rb1[PB];
    // one chunk.  Entry: window image of chunk c in LDS, B tile (c, tap 0) in S0, registers r1 = tile (c, tap 1).
    // Exit (nine steps later, an odd number): tile (c+1, 0) in S1, registers r0 = tile (c+1, 1), window image of chunk c+1.
    auto chunk = [&](float* S0, float* S1, float4 (&r0)[PB], float4 (&r1)[PB], int c) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap == 5) {   // the next chunk's window: in flight over the last four steps only (register pressure)
          load_halo(hr, cbeg + min(c + 1, nch - 1) * 16);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (tap % 2 == 0) {
          tile_B(r0, c, tap + 2);
          __builtin_amdgcn_sched_barrier(0);
          step_h(S0, S1, r1, tap);
        } else {
          tile_B(r1, c, tap + 2);
          __builtin_amdgcn_sched_barrier(0);
          step_h(S1, S0, r0, tap);
        }
        __syncthreads();
      }
      store_halo(As, hr);
      __syncthreads();
    };
    load_halo(hr, cbeg);
    tile_B(rb0, 0, 0);
    store_halo(As, hr);
#pragma unroll
    for (int q = 0; q < PB; ++q)
      if (b_act) store_B(Bs, rb0[q], q);
    tile_B(rb1, 0, 1);
    __syncthreads();
    int c = 0;
    for (; c + 1 < nch; c += 2) {
      chunk(Bs, Bs1, rb0, rb1, c);
      chunk(Bs1, Bs, rb1, rb0, c + 1);
    }
    if (c < nch) chunk(Bs, Bs1, rb0, rb1, c);
  } else
  if constexpr (GLDS) {
    // ---- LDS-DMA loop (operand planes: the pieces arrive split; probe: tools/probes/h2_plane_probe.hip) ------------------------
    // A tile row is 32 k of both planes = 128 bytes = 8 pieces [hi 0-7 | lo 0-7 | hi 8-15 | ...]; one wave instruction moves 8 rows
    // (64 lanes x 16 bytes) into a LANE-LINEAR 1 KB of LDS.  Bank conflicts of the ds_read_b128 fragment reads (16-lane service
    // groups: 16 rows, same piece) are avoided by a swizzle applied to the SOURCE address: LDS slot s of row r holds piece
    // s ^ ((r >> 1) & 7), so the group's rows fall on 16 distinct 16-byte slots of the 256-byte bank row.  Two stages, raw
    // s_barrier and counted vmcnt: the loads of tile k + 1 are in flight while tile k is computed; the MFMA k order is the classic
    // loop's (lane half kh, step g <-> k = 16 g + 8 kh + j), so the sums are bit-identical to it.
    char* const gs = reinterpret_cast<char*>(As);
    constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW;     // LDS-DMA instructions per wave and stage
    static_assert(GA >= 1 && GB >= 1 && GA * 8 * NW == BM && GB * 8 * NW == BN, "LDS-DMA loop: tile rows / waves mismatch");
    const int lrow = lane >> 3, lslot = lane & 7;
    const char* a_src[GA];
    const char* b_src[GB];
#pragma unroll
    for (int q = 0; q < GA; ++q) {
      const int r = (wave * GA + q) * 8 + lrow;
      a_src[q] = reinterpret_cast<const char*>(A + (long)min(m0 + r, p.M - 1) * p.lda + kbeg) + ((lslot ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int q = 0; q < GB; ++q) {
      const int r = (wave * GB + q) * 8 + lrow;
      b_src[q] = reinterpret_cast<const char*>(Bp + (long)min(n0 + r, p.N - 1) * p.ldb + kbeg) + ((lslot ^ ((r >> 1) & 7)) << 4);
    }
    auto issue = [&](int stage, int kt) {
      char* sa = gs + stage * G_ST + (wave * GA) * 1024;
      char* sb = gs + stage * G_ST + BM * 128 + (wave * GB) * 1024;
#pragma unroll
      for (int q = 0; q < GA; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[q] + (long)kt * 128),
                                         (__attribute__((address_space(3))) void*)(sa + q * 1024), 16, 0, 0);
#pragma unroll
      for (int q = 0; q < GB; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[q] + (long)kt * 128),
                                         (__attribute__((address_space(3))) void*)(sb + q * 1024), 16, 0, 0);
    };
    int a_ro[FM], b_ro[FN], a_sw[FM], b_sw[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) { const int r = wm * WM + i * 32 + li; a_ro[i] = r * 128; a_sw[i] = (r >> 1) & 7; }
#pragma unroll
    for (int j = 0; j < FN; ++j) { const int r = wn * WN + j * 32 + li; b_ro[j] = BM * 128 + r * 128; b_sw[j] = (r >> 1) & 7; }
    auto compute = [&](int stage) {
      const char* st = gs + stage * G_ST;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        Split8 sa[FM], sb[FN];
        const int p0 = 2 * (2 * g + kh);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          sa[i].hi = *reinterpret_cast<const bf16x8*>(st + a_ro[i] + ((p0 ^ a_sw[i]) << 4));
          sa[i].mid = *reinterpret_cast<const bf16x8*>(st + a_ro[i] + (((p0 + 1) ^ a_sw[i]) << 4));
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          sb[j].hi = *reinterpret_cast<const bf16x8*>(st + b_ro[j] + ((p0 ^ b_sw[j]) << 4));
          sb[j].mid = *reinterpret_cast<const bf16x8*>(st + b_ro[j] + (((p0 + 1) ^ b_sw[j]) << 4));
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) mfma_x(sa[i], sb[j], i, j);
      }
    };
    const int nk = (kend - kbeg) / 32;
    constexpr int GPS = GA + GB;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      // (everyone has finished reading stage (kt + 1) & 1: the barrier that closed the previous step)
      if (kt + 1 < nk) {
        issue((kt + 1) & 1, kt + 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPS) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();   // every wave's pieces of tile kt have landed
      compute(kt & 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else
  if constexpr (NSTG == 2) {
    // ---- pipelined loop: one barrier per K tile -----------------------------------------------------------------------------
    constexpr int G = FBK / 16, NCH = PA + PB, NPR = FM * FN * G;
    float4 ra0[PA], rb0[PB], ra1[PA], rb1[PB];  // two register sets: tiles of even / odd index
    // one K step: MFMAs of the current stage, with the split + store of the next tile's chunks spread between the products
    auto step = [&](const float* Ac, const float* Bc, float* An, float* Bn, const float4 (&ra)[PA], const float4 (&rb)[PB],
                    bool do_store) {
      int done = 0;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        Split8 sa[FM], sb[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) sa[i] = frag_A(Ac, g, i);
#pragma unroll
        for (int j = 0; j < FN; ++j) sb[j] = frag_B(Bc, g, j);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            mfma_x(sa[i], sb[j], i, j);
            const int t = (g * FM + i) * FN + j + 1;  // products issued so far
            const int want = (t * NCH) / NPR;
            if (do_store) {
#pragma unroll
              for (int c = 0; c < NCH; ++c)
                if (c >= done && c < want) {
                  if (c < PA) store_A(An, ra[c < PA ? c : 0], c);
                  else store_B(Bn, rb[c >= PA ? c - PA : 0], c - PA);
                }
            }
            done = want;
          }
      }
    };
    // Branch-free steady state: every step issues its prefetch and its stores unconditionally; tile positions are clamped to
    // the last tile (loaders that carry coordinates simply run past the end: their values are stored into a stage that is
    // never read, and every gather address is formed from clamped / in-range indices).
    const int nk = (kend - kbeg) / FBK;
    const int klast = kbeg + (nk - 1) * FBK;
    load_A(ra0, kbeg);
    load_B(rb0, kbeg);
#pragma unroll
    for (int q = 0; q < PA; ++q) store_A(As, ra0[q], q);
#pragma unroll
    for (int q = 0; q < PB; ++q) store_B(Bs, rb0[q], q);
    load_A(ra1, min(kbeg + FBK, klast));
    load_B(rb1, min(kbeg + FBK, klast));
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      // even step: current = stage 0 (tile kt); tile kt+1 (register set 1) -> stage 1; prefetch tile kt+2 into set 0
      load_A(ra0, min(kbeg + (kt + 2) * FBK, klast));
      load_B(rb0, min(kbeg + (kt + 2) * FBK, klast));
      __builtin_amdgcn_sched_barrier(0);
      step(As, Bs, As1, Bs1, ra1, rb1, true);
      __syncthreads();
      // odd step: current = stage 1 (tile kt+1); tile kt+2 (register set 0) -> stage 0; prefetch tile kt+3 into set 1
      load_A(ra1, min(kbeg + (kt + 3) * FBK, klast));
      load_B(rb1, min(kbeg + (kt + 3) * FBK, klast));
      __builtin_amdgcn_sched_barrier(0);
      step(As1, Bs1, As, Bs, ra0, rb0, true);
      __syncthreads();
    }
    if (kt < nk) {  // odd tile count: the last tile sits in stage 0
      step(As, Bs, As1, Bs1, ra1, rb1, false);
      __syncthreads();
    }
  } else {
    // ---- classic loop: one LDS buffer, two barriers per K tile ----------------------------------------------------------------
    float4 ra[PA], rb[PB];
    auto store_lds = [&]() {
#pragma unroll
      for (int q = 0; q < PA; ++q) store_A(As, ra[q], q);
#pragma unroll
      for (int q = 0; q < PB; ++q) store_B(Bs, rb[q], q);
    };
    load_A(ra, kbeg);
    load_B(rb, kbeg);
    store_lds();
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += FBK) {
      const bool more = (k0 + FBK) < kend;  // uniform
      if (more) {
        load_A(ra, k0 + FBK);
        load_B(rb, k0 + FBK);
      }
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
        for (int g = 0; g < FBK / 16; ++g) {
          Split8 sa[FM], sb[FN];
#pragma unroll
          for (int i = 0; i < FM; ++i) sa[i] = frag_A(As, g, i);
#pragma unroll
          for (int j = 0; j < FN; ++j) sb[j] = frag_B(Bs, g, j);
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) mfma_x(sa[i], sb[j], i, j);
        }
      }
      __syncthreads();
      if (more) {
        store_lds();
        __syncthreads();
      }
    }
  }

  // ---- epilogue (C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) ---------------------
  // (both loops end on a barrier: the staging buffers are free)
  if constexpr (H2) {   // the cross products join at their weight 2^-11 (exact scaling, one rounding)
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = fmaf(acx[i][j][r], 1.0f / 2048.0f, acc[i][j][r]);
  }
  const float alpha_e = H2 ? p.alpha * h2inv : p.alpha;   // (h2: the operand scales leave with alpha; slabs likewise)
  float st_s[FN], st_q[FN];  // fused BatchNorm statistics: per-column sum / sum of squares of this block's rows
#pragma unroll
  for (int j = 0; j < FN; ++j) st_s[j] = st_q[j] = 0.f;
  // Vector epilogue: in the accumulator layout a lane owns single floats of 16 different rows, so direct stores are 16
  // scalar instructions per 32x32 block, each touching two 128-byte row pieces (and the residual comes in the same way).
  // For the short-K products (1x1 convolutions of layer1/2: K = 64..256 against a [M, 256..512] output) that is most of the
  // kernel.  Each wave instead turns its block around in the idle staging LDS (32 x 36 floats, wave-private: program order
  // + a wave fence) and handles rows: 8 lanes x 16 bytes per row, 8 rows per instruction -- 4 loads/stores per block.
  constexpr int ELD = 36;
  // waves whose 32 x 36 turn-around block fits an array
  constexpr int EW_A = AS_ALL / (32 * ELD), EW_B = GLDS ? 0 : B_SZ / (32 * ELD);
  constexpr int EW_A1 = (NSTG == 2 && !HALO) ? EW_A : 0, EW_B1 = NSTG == 2 ? EW_B : 0;
  constexpr bool EPI_LDS = EW_A + EW_B + EW_A1 + EW_B1 >= NW;
  float4 vs_s[FN], vs_q[FN];
  unsigned e_am = 0u;   // largest magnitude stored (bits): by-product for p.amax_out
#pragma unroll
  for (int j = 0; j < FN; ++j) vs_s[j] = vs_q[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool vec_epi = EPI_LDS && p.vecC;  // uniform
  const bool nt_e = EPI != EPI_SLAB && p.nt != 0;   // uniform: epilogue streams larger than the memory-side cache (gemm_conv.hip stream_nt)
  if (vec_epi) {
    float* stg;
    if constexpr (GLDS) {
      stg = As + wave * 32 * ELD;     // (the one LDS object of the LDS-DMA loop)
    } else {
      int w = wave;
      if (w < EW_A) stg = As + w * 32 * ELD;
      else if ((w -= EW_A) < EW_B) stg = Bs_ + w * 32 * ELD;
      else if ((w -= EW_B) < EW_A1) stg = As1 + w * 32 * ELD;
      else stg = Bs1_ + (w - EW_A1) * 32 * ELD;
    }
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
        if (EPI != EPI_SLAB && p.bias_mode == 1 && col < p.N) bc = ld4(p.bias + col);
        float4 nmu = bc, nis = bc, nsc = bc, nbe = bc;   // fused BatchNorm-backward reduction: the columns' constants
        if (EPI != EPI_SLAB && p.bnb_x != nullptr && col < p.N) {
          nmu = ld4(p.bnb_mean + col);
          nis = ld4(p.bnb_invstd + col);
          if (p.bnb_y == nullptr) {
            const float4 ga = ld4(p.bnb_gamma + col);
            nbe = ld4(p.bnb_beta + col);
            nsc = make_float4(nis.x * ga.x, nis.y * ga.y, nis.z * ga.z, nis.w * ga.w);
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = grow(wm * WM + i * 32 + er + 8 * t);
          float4 v = *reinterpret_cast<const float4*>(stg + (er + 8 * t) * ELD + ec);
          if (row < p.M && col < p.N) {  // N % 4 == 0: a vector never straddles the edge
            if (EPI == EPI_SLAB) {
              if (H2) { v.x *= h2inv; v.y *= h2inv; v.z *= h2inv; v.w *= h2inv; }
              *reinterpret_cast<float4*>(p.C + ((long)bid_z * p.M + row) * p.N + col) = v;
            } else {
              v.x *= alpha_e; v.y *= alpha_e; v.z *= alpha_e; v.w *= alpha_e;
              if (p.bias_mode == 1) { v.x += bc.x; v.y += bc.y; v.z += bc.z; v.w += bc.w; }
              else if (p.bias_mode == 2) { const float bb = p.bias[row]; v.x += bb; v.y += bb; v.z += bb; v.w += bb; }
              if (EPI == EPI_XTRA && p.pre_out != nullptr) *reinterpret_cast<float4*>(p.pre_out + (long)row * p.ldc + col) = v;
              if (p.act == 1) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
              } else if (p.act == 2) {
                v.x = v.x / (1.0f + expf(-1.702f * v.x)); v.y = v.y / (1.0f + expf(-1.702f * v.y));
                v.z = v.z / (1.0f + expf(-1.702f * v.z)); v.w = v.w / (1.0f + expf(-1.702f * v.w));
              }
              if (p.resid) {
                const float4 rr = ld4s(p.resid + (long)zb * p.sR + (long)row * p.ldr + col, nt_e);
                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
              }
              if (p.bnb_x != nullptr) {
                const long o = (long)row * p.ldc + col;
                const float4 xx = ld4s(p.bnb_x + o, nt_e);
                if (p.bnb_y != nullptr && p.bnb_y_pl == 2) {
                  // the ReLU mask as one BYTE per 8 columns, written by the forward pass beside the planes (tris_bn_mask_next): 1 bit
                  // per element instead of re-reading the 4-byte plane element
                  const unsigned mb = reinterpret_cast<const unsigned char*>(p.bnb_y)[((long)row * p.ldc + col) >> 3] >> (col & 4);
                  if (!(mb & 1u)) v.x = 0.f;
                  if (!(mb & 2u)) v.y = 0.f;
                  if (!(mb & 4u)) v.z = 0.f;
                  if (!(mb & 8u)) v.w = 0.f;
                } else if (p.bnb_y != nullptr && p.bnb_y_pl) {
                  // y = relu(..) as fp16 piece planes (8 columns = 16 bytes of hi pieces, then 16 of lo pieces): y > 0 <=> a piece is non-zero
                  const char* yb = reinterpret_cast<const char*>(p.bnb_y + (long)row * p.ldc + (col & ~7)) + (col & 4) * 2;
                  const uint2 yh = *reinterpret_cast<const uint2*>(yb), yl = *reinterpret_cast<const uint2*>(yb + 16);
                  const unsigned m0 = (yh.x | yl.x) & 0x7fff7fffu, m1 = (yh.y | yl.y) & 0x7fff7fffu;
                  if (!(m0 & 0xffffu)) v.x = 0.f;
                  if (!(m0 >> 16)) v.y = 0.f;
                  if (!(m1 & 0xffffu)) v.z = 0.f;
                  if (!(m1 >> 16)) v.w = 0.f;
                } else if (p.bnb_y != nullptr) {
                  const float4 yy = ld4s(p.bnb_y + o, nt_e);
                  if (!(yy.x > 0.f)) v.x = 0.f;
                  if (!(yy.y > 0.f)) v.y = 0.f;
                  if (!(yy.z > 0.f)) v.z = 0.f;
                  if (!(yy.w > 0.f)) v.w = 0.f;
                } else {
                  if (!((xx.x - nmu.x) * nsc.x + nbe.x > 0.f)) v.x = 0.f;
                  if (!((xx.y - nmu.y) * nsc.y + nbe.y > 0.f)) v.y = 0.f;
                  if (!((xx.z - nmu.z) * nsc.z + nbe.z > 0.f)) v.z = 0.f;
                  if (!((xx.w - nmu.w) * nsc.w + nbe.w > 0.f)) v.w = 0.f;
                }
                st4s(p.C + o, v, nt_e);
                e_am = max(e_am, abits4(v));   // (the masked gradient's amax: the bound of the BatchNorm's dx, tris_bn_bwd_bound_f32)
                vs_s[j].x += v.x; vs_s[j].y += v.y; vs_s[j].z += v.z; vs_s[j].w += v.w;
                vs_q[j].x += v.x * ((xx.x - nmu.x) * nis.x); vs_q[j].y += v.y * ((xx.y - nmu.y) * nis.y);
                vs_q[j].z += v.z * ((xx.z - nmu.z) * nis.z); vs_q[j].w += v.w * ((xx.w - nmu.w) * nis.w);
              } else {
              if (EPI == EPI_XTRA && p.dact_x != nullptr) {
                const float4 xg = ld4(p.dact_x + (long)row * p.ldc + col);
                v.x = v.x * qgelu_grad(xg.x); v.y = v.y * qgelu_grad(xg.y); v.z = v.z * qgelu_grad(xg.z); v.w = v.w * qgelu_grad(xg.w);
              }
              st4s(p.C + (long)zb * p.sC + (long)row * p.ldc + col, v, nt_e);
              e_am = max(e_am, abits4(v));
              vs_s[j].x += v.x; vs_s[j].y += v.y; vs_s[j].z += v.z; vs_s[j].w += v.w;
              vs_q[j].x += v.x * v.x; vs_q[j].y += v.y * v.y; vs_q[j].z += v.z * v.z; vs_q[j].w += v.w * v.w;
              }
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
        const int row = grow(wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh);
        if (row < p.M && col < p.N) {
          float v = acc[i][j][r];
          if (EPI == EPI_SLAB) {
            p.C[((long)bid_z * p.M + row) * p.N + col] = H2 ? v * h2inv : v;
          } else {
            v *= alpha_e;
            if (p.bias_mode == 1) v += p.bias[col];
            else if (p.bias_mode == 2) v += p.bias[row];
            if (EPI == EPI_XTRA && p.pre_out != nullptr) p.pre_out[(long)row * p.ldc + col] = v;
            if (p.act == 1) v = fmaxf(v, 0.f);
            else if (p.act == 2) v = v / (1.0f + expf(-1.702f * v));
            if (p.resid) v += p.resid[(long)zb * p.sR + (long)row * p.ldr + col];
            if (EPI == EPI_XTRA && p.dact_x != nullptr) v = v * qgelu_grad(p.dact_x[(long)row * p.ldc + col]);
            if (p.bnb_x != nullptr) {   // (fused BatchNorm-backward reduction, scalar form: see the vector epilogue)
              const long o = (long)row * p.ldc + col;
              const float xx = p.bnb_x[o], mu1 = p.bnb_mean[col], is1 = p.bnb_invstd[col];
              bool on;
              if (p.bnb_y != nullptr && p.bnb_y_pl == 2) {
                on = (reinterpret_cast<const unsigned char*>(p.bnb_y)[((long)row * p.ldc + col) >> 3] >> (col & 7)) & 1u;
              } else if (p.bnb_y != nullptr && p.bnb_y_pl) {
                const unsigned short* yb = reinterpret_cast<const unsigned short*>(p.bnb_y + (long)row * p.ldc + (col & ~7)) + (col & 7);
                on = ((yb[0] | yb[8]) & 0x7fffu) != 0;
              } else
                on = p.bnb_y != nullptr ? (p.bnb_y[o] > 0.f) : ((xx - mu1) * (is1 * p.bnb_gamma[col]) + p.bnb_beta[col] > 0.f);
              if (!on) v = 0.f;
              p.C[o] = v;
              e_am = max(e_am, __builtin_bit_cast(unsigned, v) & 0x7fffffffu);
              st_s[j] += v;
              st_q[j] += v * ((xx - mu1) * is1);
            } else {
            p.C[(long)zb * p.sC + (long)row * p.ldc + col] = v;
            e_am = max(e_am, __builtin_bit_cast(unsigned, v) & 0x7fffffffu);
            st_s[j] += v;
            st_q[j] += v * v;
            }
          }
        }
      }
    }
  }
  if (EPI != EPI_SLAB && p.amax_out != nullptr) amax_commit(e_am, p.amax_out);   // (per wave, uniform)
  if (EPI != EPI_SLAB && p.stat_part != nullptr) {
    // rows of one column live in the 2 lane halves (kh) [vector epilogue: the 8 row groups er] and the NWM waves along M:
    // shuffle, then LDS, then one fp64 partial row per block: part[tile_m][2][N]  (finished by bn_finalize_kernel)
    static_assert(NWM * BN * 4 <= AS_ALL, "statistics scratch does not fit the staging buffer");
    float* red = As;
    if (p.bnb_x != nullptr) {
      // BatchNorm-backward sums cancel (dx = k1 * (dz - mean(dz) - xhat * mean(dz * xhat))): a lane's <= 16 products are summed in
      // fp32, everything beyond that -- lanes, waves, tiles -- in fp64 (col_partial_kernel<1> accumulates in fp64 throughout)
      double* redd = reinterpret_cast<double*>(As);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if (vec_epi) {
          double a[4] = {vs_s[j].x, vs_s[j].y, vs_s[j].z, vs_s[j].w}, b[4] = {vs_q[j].x, vs_q[j].y, vs_q[j].z, vs_q[j].w};
#pragma unroll
          for (int sh = 8; sh < 64; sh <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] += __shfl_xor(a[e], sh, 64); b[e] += __shfl_xor(b[e], sh, 64); }
          if (lane < 8) {
            const int cl = wn * WN + j * 32 + lane * 4;
            double* o = redd + (wm * BN + cl) * 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[2 * e] = a[e]; o[2 * e + 1] = b[e]; }
          }
        } else {
          double a = st_s[j], b = st_q[j];
          a += __shfl_xor(a, 32, 64);
          b += __shfl_xor(b, 32, 64);
          if (kh == 0) {
            const int cl = wn * WN + j * 32 + li;
            redd[(wm * BN + cl) * 2 + 0] = a;
            redd[(wm * BN + cl) * 2 + 1] = b;
          }
        }
      }
      __syncthreads();
      if (tid < BN && n0 + tid < p.N) {
        const int tm = tile / p.tiles_n;
        double* o = p.stat_part + (long)tm * 2 * p.N;
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int w = 0; w < NWM; ++w) { s += redd[(w * BN + tid) * 2]; q += redd[(w * BN + tid) * 2 + 1]; }
        o[n0 + tid] = s;
        o[p.N + n0 + tid] = q;
      }
      return;
    }
    if (vec_epi) {
      __syncthreads();  // the other waves' epilogue blocks live in the staging buffers
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
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int w = 0; w < NWM; ++w) { s += (double)red[(w * BN + tid) * 2]; q += (double)red[(w * BN + tid) * 2 + 1]; }
      o[n0 + tid] = s;
      o[p.N + n0 + tid] = q;
    }
  }
  // ---- fused split-K finish (EPI_SLAB, p.tickets != NULL: run_cfg, S <= 8 slices) ------------------------------------------------
  // The slab of this block is in the workspace.  Every block of a tile takes a ticket; the LAST arriver sums the tile's S slabs in
  // the fixed order s = 0 .. S - 1 (the order of splitk_reduce_kernel: the result does not depend on who arrives last, and equals
  // the separate reduce launch bit for bit), applies the standard epilogue and stores C -- one launch instead of two for the ~190
  // small split-K products of a step.  Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): plain
  // stores -> every wave waits for its own stores -> barrier -> one lane: agent-scope release (writes the XCD's L2 back), wait,
  // relaxed agent-scope ticket; last arriver: one agent-scope acquire (drops this CU's L1) -> barrier -> plain loads.  The last
  // arriver leaves its ticket at zero: the array is zero again when the launch ends (the next launch on the stream, or the next
  // replay of a captured one, finds it so).  Placement-independent: nothing assumes which XCD ran which slice.
  if constexpr (EPI == EPI_SLAB) {
    if (p.tickets != nullptr) {   // (uniform)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int old = __hip_atomic_fetch_add(p.tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        As[0] = __builtin_bit_cast(float, old);
      }
      __syncthreads();
      const int old = __builtin_bit_cast(int, As[0]);
      if (old != p.splitk - 1) return;
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(p.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      const int S = p.splitk;
      const long total = (long)p.M * p.N;
      const float* __restrict__ slab = p.C;
      float* __restrict__ Cf = p.Cfin;
      unsigned am = 0u;
      if (p.vecC && p.vecCfin) {
        constexpr int VPR = BN / 4;
        for (int e = tid; e < BM * VPR; e += NTHR) {
          const int row = grow(e / VPR), col = n0 + (e % VPR) * 4;
          if (row >= p.M || col >= p.N) continue;
          const long idx = (long)row * p.N + col;
          float v[4] = {0.f, 0.f, 0.f, 0.f};
          int s2 = 0;
          for (; s2 + 3 < S; s2 += 4) {   // (four slabs in flight per trip; the additions in slice order)
            const float4 a = ld4(slab + (long)s2 * total + idx), b = ld4(slab + (long)(s2 + 1) * total + idx);
            const float4 c = ld4(slab + (long)(s2 + 2) * total + idx), d = ld4(slab + (long)(s2 + 3) * total + idx);
            v[0] = (((v[0] + a.x) + b.x) + c.x) + d.x;
            v[1] = (((v[1] + a.y) + b.y) + c.y) + d.y;
            v[2] = (((v[2] + a.z) + b.z) + c.z) + d.z;
            v[3] = (((v[3] + a.w) + b.w) + c.w) + d.w;
          }
          for (; s2 < S; ++s2) {
            const float4 a = ld4(slab + (long)s2 * total + idx);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
          }
          float4 bc = make_float4(0.f, 0.f, 0.f, 0.f);
          if (p.bias_mode == 1) bc = ld4(p.bias + col);
          else if (p.bias_mode == 2) { const float bb = p.bias[row]; bc = make_float4(bb, bb, bb, bb); }
          const float bcv[4] = {bc.x, bc.y, bc.z, bc.w};
          float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
          if (p.resid) rr = ld4(p.resid + (long)row * p.ldr + col);
          const float rrv[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            float x = v[t] * p.alpha;
            if (p.bias_mode) x += bcv[t];
            if (p.act == 1) x = fmaxf(x, 0.f);
            else if (p.act == 2) x = x / (1.0f + expf(-1.702f * x));
            if (p.resid) x += rrv[t];
            v[t] = x;
            am = max(am, __builtin_bit_cast(unsigned, x) & 0x7fffffffu);
          }
          *reinterpret_cast<float4*>(Cf + (long)row * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
        }
      } else {
        for (int e = tid; e < BM * BN; e += NTHR) {
          const int row = grow(e / BN), col = n0 + (e % BN);
          if (row >= p.M || col >= p.N) continue;
          const long idx = (long)row * p.N + col;
          float x = 0.f;
          for (int s2 = 0; s2 < S; ++s2) x += slab[(long)s2 * total + idx];
          x *= p.alpha;
          if (p.bias_mode == 1) x += p.bias[col];
          else if (p.bias_mode == 2) x += p.bias[row];
          if (p.act == 1) x = fmaxf(x, 0.f);
          else if (p.act == 2) x = x / (1.0f + expf(-1.702f * x));
          if (p.resid) x += p.resid[(long)row * p.ldr + col];
          Cf[(long)row * p.ldc + col] = x;
          am = max(am, __builtin_bit_cast(unsigned, x) & 0x7fffffffu);
        }
      }
      if (p.amax_out != nullptr) amax_commit(am, p.amax_out);   // (every lane of every wave reaches this point)
    }
  }
}
