// Dev probe: the h2 GEMM core fed by OPERAND PLANES through LDS-DMA -- C[M,N] = A[M,K] . B[N,K]^T, A and B given as P8 fp16 piece
// planes (tris_amd/csrc/planes.h), fp32 out.  The main loop has no VALU work on the operands at all: global_load_lds_dwordx4 streams
// 8 rows x 128 bytes (= 32 k of both planes) per wave instruction into a lane-linear LDS image whose 16-byte pieces are XOR-swizzled
// on the SOURCE side (piece ^= (row >> 1) & 7: the 16 lanes of every ds_read_b128 service group hit 16 distinct slots), fragments are
// single ds_read_b128, three v_mfma_f32_32x32x16_f16 per product into two accumulator sets.
// Variants (template): tile, wave grid, LDS stages (2: two barriers per k step; 3: one barrier, a tile in flight across it), and
// ablations of the loop (ABL 1: no MFMA; 2: no fragment reads; 4: no global loads after the prologue).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tris_amd/csrc tools/probes/h2_plane_probe.hip -o tools/probes/h2_plane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "x3_split.h"
#include "planes.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void to_planes(const float* __restrict__ x, float* __restrict__ out, long n8, float s) {
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < n8; g += (long)gridDim.x * blockDim.x) {
    const float4 a = *reinterpret_cast<const float4*>(x + g * 8), b = *reinterpret_cast<const float4*>(x + g * 8 + 4);
    pl8_store(out, g, pl8_split(a, b, s));
  }
}
__global__ void ref_rows(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int rows, int N, int K) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N || m >= rows) return;
  double s = 0.0;
  for (int k = 0; k < K; ++k) s += (double)A[(long)m * K + k] * (double)B[(long)n * K + k];
  C[(long)m * N + n] = (float)s;
}

template <int BM, int BN, int NWM, int NWN, int NST, int OCC, int ABL = 0>
__global__ __launch_bounds__(NWM* NWN * 64, (OCC * NWM * NWN + 3) / 4) void plane_kernel(const float* __restrict__ A,
                                                                                       const float* __restrict__ B,
                                                                                       float* __restrict__ C, int M, int N, int K,
                                                                                       float inv_scale) {
  constexpr int NW = NWM * NWN, NTHR = NW * 64;
  constexpr int WM = BM / NWM, WN = BN / NWN, FM = WM / 32, FN = WN / 32;
  constexpr int A_ST = BM * 128, B_ST = BN * 128, ST = A_ST + B_ST;   // bytes per stage (32 k x 2 planes = 128 bytes per row)
  constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW;                   // LDS-DMA instructions per wave per stage (8 rows each)
  static_assert(GA >= 1 && GB >= 1 && GA * 8 * NW == BM && GB * 8 * NW == BN, "tile / wave count mismatch");
  __shared__ __attribute__((aligned(1024))) char lds[NST * ST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int tiles_n = (N + BN - 1) / BN;
  int bid = blockIdx.x;
  if (ABL & 8) {   // XCD-aware order: consecutive workgroups go to consecutive XCDs -> give every XCD a CONTIGUOUS range of tiles, so that
    // the column tiles of one row block (same A rows) run on one XCD and share its L2 (bijective for any grid size)
    const int T = gridDim.x, q = T >> 3, r = T & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  // this lane's source rows / pieces for its LDS-DMA instructions: instruction q of the wave covers tile rows (wave * G + q) * 8 .. + 7
  const int lrow = lane >> 3, lslot = lane & 7;
  const char* a_src[GA];
  const char* b_src[GB];
#pragma unroll
  for (int q = 0; q < GA; ++q) {
    const int r = (wave * GA + q) * 8 + lrow;
    a_src[q] = reinterpret_cast<const char*>(A + (long)min(m0 + r, M - 1) * K) + ((lslot ^ ((r >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int q = 0; q < GB; ++q) {
    const int r = (wave * GB + q) * 8 + lrow;
    b_src[q] = reinterpret_cast<const char*>(B + (long)min(n0 + r, N - 1) * K) + ((lslot ^ ((r >> 1) & 7)) << 4);
  }
  auto issue = [&](int stage, int kt) {
    if ((ABL & 4) && kt >= NST - 1) return;
    char* sa = lds + stage * ST + (wave * GA) * 1024;
    char* sb = lds + stage * ST + A_ST + (wave * GB) * 1024;
#pragma unroll
    for (int q = 0; q < GA; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[q] + (long)kt * 128),
                                       (__attribute__((address_space(3))) void*)(sa + q * 1024), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < GB; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[q] + (long)kt * 128),
                                       (__attribute__((address_space(3))) void*)(sb + q * 1024), 16, 0, 0);
  };
  f32x16 acc[FM][FN], acx[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = acx[i][j][r] = 0.f;
  const int li = lane & 31, kh = lane >> 5;
  // fragment addresses inside a stage: row * 128 + ((2 (2 g + kh) + plane) ^ ((row >> 1) & 7)) * 16
  int a_row[FM], b_row[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) a_row[i] = wm * WM + i * 32 + li;
#pragma unroll
  for (int j = 0; j < FN; ++j) b_row[j] = wn * WN + j * 32 + li;
  auto compute = [&](int stage) {
    const char* sa = lds + stage * ST;
    const char* sb = sa + A_ST;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f16x8 ah[FM], al[FM], bh[FN], bl[FN];
      if (!(ABL & 2)) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int sw = (a_row[i] >> 1) & 7, p0 = 2 * (2 * g + kh);
          ah[i] = *reinterpret_cast<const f16x8*>(sa + a_row[i] * 128 + ((p0 ^ sw) << 4));
          al[i] = *reinterpret_cast<const f16x8*>(sa + a_row[i] * 128 + (((p0 + 1) ^ sw) << 4));
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int sw = (b_row[j] >> 1) & 7, p0 = 2 * (2 * g + kh);
          bh[j] = *reinterpret_cast<const f16x8*>(sb + b_row[j] * 128 + ((p0 ^ sw) << 4));
          bl[j] = *reinterpret_cast<const f16x8*>(sb + b_row[j] * 128 + (((p0 + 1) ^ sw) << 4));
        }
      } else {
#pragma unroll
        for (int i = 0; i < FM; ++i) { ah[i] = (f16x8)(_Float16)(1.0f + lane); al[i] = ah[i]; }
#pragma unroll
        for (int j = 0; j < FN; ++j) { bh[j] = (f16x8)(_Float16)(0.5f); bl[j] = bh[j]; }
      }
      if (!(ABL & 1)) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acx[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acx[i][j], 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j][0] += (float)ah[i][0] * (float)bh[j][0] + (float)al[i][1] * (float)bl[j][1];
      }
    }
  };
  const int nk = K / 32;
  constexpr int GPS = GA + GB;   // LDS-DMA instructions per wave per stage
  if constexpr (NST == 2) {
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      // (everyone has finished reading stage (kt + 1) & 1 -- the barrier that closed the previous step)
      if (kt + 1 < nk) {
        issue((kt + 1) & 1, kt + 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPS) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();   // every wave's pieces of stage kt have landed
      compute(kt & 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // stage kt landed everywhere AND everyone is done reading stage (kt - 1) % 3 = (kt + 2) % 3
      if (kt + 2 < nk) issue((kt + 2) % 3, kt + 2);
      compute(kt % 3);
    }
  }
  // epilogue (probe: direct stores of the accumulator layout)
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = n0 + wn * WN + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < M && col < N) C[(long)row * N + col] = fmaf(acx[i][j][r], 1.0f / 2048.0f, acc[i][j][r]) * inv_scale;
      }
    }
}


// ---- 8-wave PING-PONG: 256 x 128 tile, wave tile 64 x 64, three LDS stages; waves 0-3 (one per SIMD) and waves 4-7 run the same
// two-phase loop half a step apart: while one group issues its 24 MFMAs of a k tile from registers, the other reads its 16 fragments
// of the next tile from LDS (and group A issues the LDS-DMA of the tile two ahead).  Every phase ends with one s_barrier of all
// eight waves; group B enters one barrier late.  ABL as above.
template <int ABL = 0, int PRIO = 1>
__global__ __launch_bounds__(512, 2) void pingpong_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          float* __restrict__ C, int M, int N, int K, float inv_scale) {
  constexpr int BM = 256, BN = 128, NW = 8, NST = 3;
  constexpr int A_ST = BM * 128, B_ST = BN * 128, ST = A_ST + B_ST;
  constexpr int GA = BM / 8 / 4, GB = BN / 8 / 4;        // LDS-DMA instructions per ISSUING wave (group A: 4 waves) and stage
  __shared__ __attribute__((aligned(1024))) char lds[NST * ST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wq = wave & 3;
  const int wm = wave >> 1, wn = wave & 1;               // 4 x 2 wave grid, 64 x 64 each
  const int tiles_n = (N + BN - 1) / BN;
  int bid = blockIdx.x;
  if (ABL & 8) {
    const int T = gridDim.x, q = T >> 3, r = T & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  const int lrow = lane >> 3, lslot = lane & 7;
  const char* a_src[GA];
  const char* b_src[GB];
#pragma unroll
  for (int q = 0; q < GA; ++q) {
    const int r = (wq * GA + q) * 8 + lrow;
    a_src[q] = reinterpret_cast<const char*>(A + (long)min(m0 + r, M - 1) * K) + ((lslot ^ ((r >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int q = 0; q < GB; ++q) {
    const int r = (wq * GB + q) * 8 + lrow;
    b_src[q] = reinterpret_cast<const char*>(B + (long)min(n0 + r, N - 1) * K) + ((lslot ^ ((r >> 1) & 7)) << 4);
  }
  auto issue = [&](int stage, int kt) {
    if ((ABL & 4) && kt >= 2) return;
    char* sa = lds + stage * ST + (wq * GA) * 1024;
    char* sb = lds + stage * ST + A_ST + (wq * GB) * 1024;
#pragma unroll
    for (int q = 0; q < GA; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[q] + (long)kt * 128),
                                       (__attribute__((address_space(3))) void*)(sa + q * 1024), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < GB; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[q] + (long)kt * 128),
                                       (__attribute__((address_space(3))) void*)(sb + q * 1024), 16, 0, 0);
  };
  f32x16 acc[2][2], acx[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = acx[i][j][r] = 0.f;
  const int li = lane & 31, kh = lane >> 5;
  int a_off[2], b_off[2], a_sw[2], b_sw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int r = wm * 64 + i * 32 + li; a_off[i] = r * 128; a_sw[i] = (r >> 1) & 7; }
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int r = wn * 64 + j * 32 + li; b_off[j] = A_ST + r * 128; b_sw[j] = (r >> 1) & 7; }
  f16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];   // [g][i]
  auto read_frags = [&](int stage) {
    const char* st = lds + stage * ST;
    if (ABL & 2) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i) { ah[g][i] = (f16x8)(_Float16)(1.0f + lane); al[g][i] = ah[g][i]; bh[g][i] = (f16x8)(_Float16)(0.5f); bl[g][i] = bh[g][i]; }
      return;
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int p0 = 2 * (2 * g + kh);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[g][i] = *reinterpret_cast<const f16x8*>(st + a_off[i] + ((p0 ^ a_sw[i]) << 4));
        al[g][i] = *reinterpret_cast<const f16x8*>(st + a_off[i] + (((p0 + 1) ^ a_sw[i]) << 4));
        bh[g][i] = *reinterpret_cast<const f16x8*>(st + b_off[i] + ((p0 ^ b_sw[i]) << 4));
        bl[g][i] = *reinterpret_cast<const f16x8*>(st + b_off[i] + (((p0 + 1) ^ b_sw[i]) << 4));
      }
    }
  };
  auto mfmas = [&]() {
    if (ABL & 1) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j][0] += (float)ah[g][i][0] * (float)bh[g][j][0] + (float)al[g][i][1] * (float)bl[g][j][1];
      return;
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[g][i], bh[g][j], acx[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g][i], bh[g][j], acc[i][j], 0, 0, 0);
          acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g][i], bl[g][j], acx[i][j], 0, 0, 0);
        }
  };
  const int nk = K / 32;
  constexpr int GPS = GA + GB;
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  if (grp == 0) {
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();                                       // tile 0 is in LDS
    for (int kt = 0; kt < nk; ++kt) {
      read_frags(kt % 3);
      if (kt + 2 < nk) issue((kt + 2) % 3, kt + 2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      bar();
      if (PRIO) __builtin_amdgcn_s_setprio(1);
      mfmas();
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPS) : "memory");   // tile kt + 1 has landed
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      bar();
    }
    bar();
  } else {
    bar();
    bar();
    for (int kt = 0; kt < nk; ++kt) {
      read_frags(kt % 3);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      bar();
      if (PRIO) __builtin_amdgcn_s_setprio(1);
      mfmas();
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      bar();
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < M && col < N) C[(long)row * N + col] = fmaf(acx[i][j][r], 1.0f / 2048.0f, acc[i][j][r]) * inv_scale;
      }
    }
}


// ---- register-pipelined form: 256 x 128 tile, FOUR waves (2 x 2), wave tile 128 x 64 (FM = 4, FN = 2: 12 KB of fragments per 24 MFMAs instead
// of 8 KB per 12), three LDS stages, ONE barrier per k tile, fragments double-buffered in registers: the ds_reads of sub-step g + 1 are issued
// right before the MFMAs of sub-step g, so a wave's own matrix work covers its LDS latency (one wave per SIMD: nobody else would).
template <int ABL = 0>
__global__ __launch_bounds__(256, 1) void regpipe_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                         int M, int N, int K, float inv_scale) {
  constexpr int BM = 256, BN = 128, NW = 4, NST = 3, FM = 4, FN = 2;
  constexpr int A_ST = BM * 128, B_ST = BN * 128, ST = A_ST + B_ST;
  constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW;        // 8 + 4 LDS-DMA instructions per wave and stage
  __shared__ __attribute__((aligned(1024))) char lds[NST * ST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (N + BN - 1) / BN;
  int bid = blockIdx.x;
  if (ABL & 8) {
    const int T = gridDim.x, q = T >> 3, r = T & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  const int lrow = lane >> 3, lslot = lane & 7;
  const char* a_src[GA];
  const char* b_src[GB];
#pragma unroll
  for (int q = 0; q < GA; ++q) {
    const int r = (wave * GA + q) * 8 + lrow;
    a_src[q] = reinterpret_cast<const char*>(A + (long)min(m0 + r, M - 1) * K) + ((lslot ^ ((r >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int q = 0; q < GB; ++q) {
    const int r = (wave * GB + q) * 8 + lrow;
    b_src[q] = reinterpret_cast<const char*>(B + (long)min(n0 + r, N - 1) * K) + ((lslot ^ ((r >> 1) & 7)) << 4);
  }
  auto issue = [&](int stage, int kt) {
    char* sa = lds + stage * ST + (wave * GA) * 1024;
    char* sb = lds + stage * ST + A_ST + (wave * GB) * 1024;
#pragma unroll
    for (int q = 0; q < GA; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[q] + (long)kt * 128),
                                       (__attribute__((address_space(3))) void*)(sa + q * 1024), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < GB; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[q] + (long)kt * 128),
                                       (__attribute__((address_space(3))) void*)(sb + q * 1024), 16, 0, 0);
  };
  f32x16 acc[FM][FN], acx[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = acx[i][j][r] = 0.f;
  const int li = lane & 31, kh = lane >> 5;
  int a_off[FM], b_off[FN], a_sw[FM], b_sw[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) { const int r = wm * 128 + i * 32 + li; a_off[i] = r * 128; a_sw[i] = (r >> 1) & 7; }
#pragma unroll
  for (int j = 0; j < FN; ++j) { const int r = wn * 64 + j * 32 + li; b_off[j] = A_ST + r * 128; b_sw[j] = (r >> 1) & 7; }
  struct Fr { f16x8 ah[FM], al[FM], bh[FN], bl[FN]; };
  // ABL & 16: the fragment reads are inline asm (the compiler does not see LDS operations, so it inserts no waits of its own -- its
  // s_waitcnt lgkmcnt(0) in front of the first MFMA group drains the reads just issued for the NEXT sub-step) and the waits are counted
  // by hand: LDS returns in order, 12 reads per sub-step
  auto lds_rd = [&](const char* p) -> f16x8 {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 v;
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a));
    return __builtin_bit_cast(f16x8, v);
  };
  auto read_frags = [&](Fr& f, int stage, int g) {
    const char* st = lds + stage * ST;
    const int p0 = 2 * (2 * g + kh);
    if (ABL & 16) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        f.ah[i] = lds_rd(st + a_off[i] + ((p0 ^ a_sw[i]) << 4));
        f.al[i] = lds_rd(st + a_off[i] + (((p0 + 1) ^ a_sw[i]) << 4));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        f.bh[j] = lds_rd(st + b_off[j] + ((p0 ^ b_sw[j]) << 4));
        f.bl[j] = lds_rd(st + b_off[j] + (((p0 + 1) ^ b_sw[j]) << 4));
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      f.ah[i] = *reinterpret_cast<const f16x8*>(st + a_off[i] + ((p0 ^ a_sw[i]) << 4));
      f.al[i] = *reinterpret_cast<const f16x8*>(st + a_off[i] + (((p0 + 1) ^ a_sw[i]) << 4));
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      f.bh[j] = *reinterpret_cast<const f16x8*>(st + b_off[j] + ((p0 ^ b_sw[j]) << 4));
      f.bl[j] = *reinterpret_cast<const f16x8*>(st + b_off[j] + (((p0 + 1) ^ b_sw[j]) << 4));
    }
  };
  auto mfmas = [&](const Fr& f) {
    if (ABL & 1) return;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[j], acx[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
        acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[j], acx[i][j], 0, 0, 0);
      }
  };
  const int nk = K / 32;
  constexpr int GPS = GA + GB;
  Fr f0, f1;
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPS) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                 // tile 0 is in LDS
  read_frags(f0, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt + 1: my pieces landed -> barrier: everyone's landed, and everyone has issued all reads of tile kt - 1 (stage (kt + 2) % 3)
    if (kt + 1 < nk) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 2 < nk) issue((kt + 2) % 3, kt + 2);
    }
    read_frags(f1, kt % 3, 1);                  // sub-step 1 of this tile: in flight under the MFMAs of sub-step 0
    if (ABL & 16) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");   // f0 (the 12 reads before these 12) has arrived
    __builtin_amdgcn_sched_barrier(0);
    mfmas(f0);
    __builtin_amdgcn_sched_barrier(0);
    read_frags(f0, (kt + 1) % 3, 0);   // sub-step 0 of the next tile: under the MFMAs of sub-step 1 (unconditional: the last one reads a dead stage)
    if (ABL & 16) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");   // f1 has arrived
    __builtin_amdgcn_sched_barrier(0);
    mfmas(f1);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = n0 + wn * 64 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < M && col < N) C[(long)row * N + col] = fmaf(acx[i][j][r], 1.0f / 2048.0f, acc[i][j][r]) * inv_scale;
      }
    }
}


// ---- the same 4-wave form with SIX stages of 16 k each (24 KB per stage): four stages in flight while one is consumed -- the three-stage
// form above has ONE 48 KB tile in flight per CU, and at ~1.6 us of LDS-DMA round trip that is a quarter of what the matrix pipe eats.
// A stage row is 16 k of both planes = 64 bytes = 4 pieces [hi 0-7 | lo 0-7 | hi 8-15 | lo 8-15]; one DMA instruction moves 16 rows; swizzle
// piece ^= (row >> 2) & 3 on the source side (the 16 lanes of a b128 service group hit 16 distinct slots of the 256-byte bank row).
template <int ABL = 0, int NST = 6>
__global__ __launch_bounds__(256, 1) void regpipe16_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                           int M, int N, int K, float inv_scale) {
  constexpr int BM = 256, BN = 128, NW = 4, FM = 4, FN = 2;
  constexpr int A_ST = BM * 64, B_ST = BN * 64, ST = A_ST + B_ST;     // 24 KB
  constexpr int GA = BM / 16 / NW, GB = BN / 16 / NW;               // 4 + 2 LDS-DMA instructions per wave and stage
  __shared__ __attribute__((aligned(1024))) char lds[NST * ST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (N + BN - 1) / BN;
  int bid = blockIdx.x;
  if (ABL & 8) {
    const int T = gridDim.x, q = T >> 3, r = T & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  const int lrow = lane >> 2, lslot = lane & 3;
  const char* a_src[GA];
  const char* b_src[GB];
#pragma unroll
  for (int q = 0; q < GA; ++q) {
    const int r = (wave * GA + q) * 16 + lrow;
    a_src[q] = reinterpret_cast<const char*>(A + (long)min(m0 + r, M - 1) * K) + ((lslot ^ ((r >> 2) & 3)) << 4);
  }
#pragma unroll
  for (int q = 0; q < GB; ++q) {
    const int r = (wave * GB + q) * 16 + lrow;
    b_src[q] = reinterpret_cast<const char*>(B + (long)min(n0 + r, N - 1) * K) + ((lslot ^ ((r >> 2) & 3)) << 4);
  }
  auto issue = [&](int stage, int kt) {       // kt counts 16-k tiles: 64 bytes along a row of the plane tensor
    char* sa = lds + stage * ST + (wave * GA) * 1024;
    char* sb = lds + stage * ST + A_ST + (wave * GB) * 1024;
#pragma unroll
    for (int q = 0; q < GA; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[q] + (long)kt * 64),
                                       (__attribute__((address_space(3))) void*)(sa + q * 1024), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < GB; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[q] + (long)kt * 64),
                                       (__attribute__((address_space(3))) void*)(sb + q * 1024), 16, 0, 0);
  };
  f32x16 acc[FM][FN], acx[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = acx[i][j][r] = 0.f;
  const int li = lane & 31, kh = lane >> 5;
  unsigned a_ad[FM][2], b_ad[FN][2];       // LDS byte offsets inside a stage: [fragment][hi | lo]
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int r = wm * 128 + i * 32 + li, sw = (r >> 2) & 3;
    a_ad[i][0] = r * 64 + (((2 * kh) ^ sw) << 4);
    a_ad[i][1] = r * 64 + (((2 * kh + 1) ^ sw) << 4);
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int r = wn * 64 + j * 32 + li, sw = (r >> 2) & 3;
    b_ad[j][0] = A_ST + r * 64 + (((2 * kh) ^ sw) << 4);
    b_ad[j][1] = A_ST + r * 64 + (((2 * kh + 1) ^ sw) << 4);
  }
  struct Fr { f16x8 ah[FM], al[FM], bh[FN], bl[FN]; };
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)lds;
  auto lds_rd = [&](unsigned a) -> f16x8 {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a));
    return __builtin_bit_cast(f16x8, v);
  };
  auto read_frags = [&](Fr& f, int stage) {
    const unsigned st = lds0 + stage * ST;
#pragma unroll
    for (int i = 0; i < FM; ++i) { f.ah[i] = lds_rd(st + a_ad[i][0]); f.al[i] = lds_rd(st + a_ad[i][1]); }
#pragma unroll
    for (int j = 0; j < FN; ++j) { f.bh[j] = lds_rd(st + b_ad[j][0]); f.bl[j] = lds_rd(st + b_ad[j][1]); }
  };
  auto mfmas = [&](const Fr& f) {
    if (ABL & 1) return;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[j], acx[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
        acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[j], acx[i][j], 0, 0, 0);
      }
  };
  const int nk = K / 16;                      // 16-k tiles
  constexpr int GPS = GA + GB;                // 6 DMA instructions per wave and tile
  constexpr int AHEAD = NST - 2;              // tiles in flight behind the one being read and the one being consumed from registers
  // prologue: tiles 0 .. AHEAD
#pragma unroll
  for (int t = 0; t <= AHEAD; ++t) if (t < nk) issue(t, t);
  // wait for tile 0 (everything but the AHEAD newest tiles), barrier, read it
  if (nk > AHEAD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPS * AHEAD) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  Fr f0, f1;
  read_frags(f0, 0);
  // steady state, two tiles per trip (f0: even tiles, f1: odd tiles).  At the top of the half-trip for tile t: tiles t + 1 .. t + AHEAD are in
  // flight (or landed); wait for t + 1, barrier (everyone's t + 1 landed; everyone has issued its reads of tile t, i.e. is done with tile
  // t - 1's stage), issue tile t + AHEAD + 1 into that stage, read tile t + 1 under the MFMAs of tile t.
  auto half = [&](Fr& cur, Fr& nxt, int t) {
    if (t + 1 < nk) {
      if (t + AHEAD < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPS * (AHEAD - 1)) : "memory");   // t + 1 .. t + AHEAD are outstanding
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          // the tail: fewer are
      __builtin_amdgcn_s_barrier();
      if (t + AHEAD + 1 < nk) issue((t + AHEAD + 1) % NST, t + AHEAD + 1);
      else { /* keep the counter arithmetic uniform: nothing to issue, later waits are then stricter than needed */ }
    }
    read_frags(nxt, (t + 1) % NST);
    asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");       // cur's 12 reads (issued before these 12) have arrived
    __builtin_amdgcn_sched_barrier(0);
    mfmas(cur);
    __builtin_amdgcn_sched_barrier(0);
  };
  int t = 0;
  for (; t + 1 < nk; t += 2) {
    half(f0, f1, t);
    half(f1, f0, t + 1);
  }
  if (t < nk) half(f0, f1, t);
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = n0 + wn * 64 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < M && col < N) C[(long)row * N + col] = fmaf(acx[i][j][r], 1.0f / 2048.0f, acc[i][j][r]) * inv_scale;
      }
    }
}

static float host_scale(float amax) {
  int e;
  frexpf(amax, &e);          // amax = f * 2^e, f in [0.5, 1)  ->  floor(log2 amax) = e - 1
  return ldexpf(1.0f, 13 - (e - 1));
}

template <int BM, int BN, int NWM, int NWN, int NST, int OCC, int ABL>
static void run(const char* name, const float* Ap, const float* Bp, float* C, const float* Cref, int M, int N, int K, float inv,
                int ref_rows_n) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  auto launch = [&]() {
    hipLaunchKernelGGL((plane_kernel<BM, BN, NWM, NWN, NST, OCC, ABL>), dim3(tiles), dim3(NWM * NWN * 64), 0, 0, Ap, Bp, C, M, N, K, inv);
  };
  for (int i = 0; i < 3; ++i) launch();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int it = 10;
  for (int i = 0; i < it; ++i) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= it;
  double err = -1.0;
  if (ABL == 0 && Cref) {
    std::vector<float> h((size_t)ref_rows_n * N), r((size_t)ref_rows_n * N);
    hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), Cref, r.size() * 4, hipMemcpyDeviceToHost);
    double num = 0, den = 0;
    for (size_t i = 0; i < h.size(); ++i) { num = fmax(num, fabs((double)h[i] - r[i])); den = fmax(den, fabs((double)r[i])); }
    err = num / den;
  }
  printf("%-44s M%-7d N%-5d K%-5d %9.1f us %7.1f TF/s  err/max %.2e\n", name, M, N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) * 1e-12, err);
  fflush(stdout);
}

template <int ABL, int PRIO>
static void run_pp(const char* name, const float* Ap, const float* Bp, float* C, const float* Cref, int M, int N, int K, float inv,
                   int ref_rows_n) {
  const int tiles = ((M + 255) / 256) * ((N + 127) / 128);
  auto launch = [&]() { hipLaunchKernelGGL((pingpong_kernel<ABL, PRIO>), dim3(tiles), dim3(512), 0, 0, Ap, Bp, C, M, N, K, inv); };
  for (int i = 0; i < 3; ++i) launch();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int it = 10;
  for (int i = 0; i < it; ++i) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= it;
  double err = -1.0;
  if ((ABL & 7) == 0 && Cref) {
    std::vector<float> h((size_t)ref_rows_n * N), r((size_t)ref_rows_n * N);
    hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), Cref, r.size() * 4, hipMemcpyDeviceToHost);
    double num = 0, den = 0;
    for (size_t i = 0; i < h.size(); ++i) { num = fmax(num, fabs((double)h[i] - r[i])); den = fmax(den, fabs((double)r[i])); }
    err = num / den;
  }
  printf("%-44s M%-7d N%-5d K%-5d %9.1f us %7.1f TF/s  err/max %.2e\n", name, M, N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) * 1e-12, err);
  fflush(stdout);
}

template <int ABL>
static void run_rp(const char* name, const float* Ap, const float* Bp, float* C, const float* Cref, int M, int N, int K, float inv,
                   int ref_rows_n) {
  const int tiles = ((M + 255) / 256) * ((N + 127) / 128);
  auto launch = [&]() { hipLaunchKernelGGL((regpipe_kernel<ABL>), dim3(tiles), dim3(256), 0, 0, Ap, Bp, C, M, N, K, inv); };
  for (int i = 0; i < 3; ++i) launch();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int it = 10;
  for (int i = 0; i < it; ++i) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= it;
  double err = -1.0;
  if ((ABL & 7) == 0 && Cref) {
    std::vector<float> h((size_t)ref_rows_n * N), r((size_t)ref_rows_n * N);
    hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), Cref, r.size() * 4, hipMemcpyDeviceToHost);
    double num = 0, den = 0;
    for (size_t i = 0; i < h.size(); ++i) { num = fmax(num, fabs((double)h[i] - r[i])); den = fmax(den, fabs((double)r[i])); }
    err = num / den;
  }
  printf("%-44s M%-7d N%-5d K%-5d %9.1f us %7.1f TF/s  err/max %.2e\n", name, M, N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) * 1e-12, err);
  fflush(stdout);
}

template <int ABL>
static void run_rp16(const char* name, const float* Ap, const float* Bp, float* C, const float* Cref, int M, int N, int K, float inv,
                     int ref_rows_n) {
  const int tiles = ((M + 255) / 256) * ((N + 127) / 128);
  auto launch = [&]() { hipLaunchKernelGGL((regpipe16_kernel<ABL, 6>), dim3(tiles), dim3(256), 0, 0, Ap, Bp, C, M, N, K, inv); };
  for (int i = 0; i < 3; ++i) launch();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int it = 10;
  for (int i = 0; i < it; ++i) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= it;
  double err = -1.0;
  if ((ABL & 7) == 0 && Cref) {
    std::vector<float> h((size_t)ref_rows_n * N), r((size_t)ref_rows_n * N);
    hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), Cref, r.size() * 4, hipMemcpyDeviceToHost);
    double num = 0, den = 0;
    for (size_t i = 0; i < h.size(); ++i) { num = fmax(num, fabs((double)h[i] - r[i])); den = fmax(den, fabs((double)r[i])); }
    err = num / den;
  }
  printf("%-44s M%-7d N%-5d K%-5d %9.1f us %7.1f TF/s  err/max %.2e\n", name, M, N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) * 1e-12, err);
  fflush(stdout);
}

int main(int argc, char** argv) {
  struct Shape { int M, N, K; };
  std::vector<Shape> shapes = {{4096, 4096, 4096}, {76800, 256, 2304}, {19200, 512, 4608}, {19200, 1024, 1024}, {19200, 1024, 256}, {4800, 2048, 1024},
                               {76800, 128, 512}, {76800, 512, 256}, {307200, 256, 64}, {307200, 64, 256}};
  for (const Shape& s : shapes) {
    const int M = s.M, N = s.N, K = s.K;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
    float amA = 0, amB = 0;
    for (auto& v : hA) { v = rnd(); amA = fmaxf(amA, fabsf(v)); }
    for (auto& v : hB) { v = rnd(); amB = fmaxf(amB, fabsf(v)); }
    float *A, *B, *Ap, *Bp, *C, *Cref;
    hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&Ap, hA.size() * 4); hipMalloc(&Bp, hB.size() * 4);
    hipMalloc(&C, (size_t)M * N * 4);
    const int RR = 128;
    hipMalloc(&Cref, (size_t)RR * N * 4);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    const float sA = host_scale(amA), sB = host_scale(amB);
    hipLaunchKernelGGL(to_planes, dim3(2048), dim3(256), 0, 0, A, Ap, (long)M * K / 8, sA);
    hipLaunchKernelGGL(to_planes, dim3(2048), dim3(256), 0, 0, B, Bp, (long)N * K / 8, sB);
    hipLaunchKernelGGL(ref_rows, dim3((N + 255) / 256, RR), dim3(256), 0, 0, A, B, Cref, RR, N, K);
    hipDeviceSynchronize();
    const float inv = 1.0f / (sA * sB);
#define RUN(BM, BN, NWM, NWN, NST, OCC, ABL, NAME) run<BM, BN, NWM, NWN, NST, OCC, ABL>(NAME, Ap, Bp, C, Cref, M, N, K, inv, RR)
    run_rp16<0>("256x128 4w k16 x 6 stages, 4 in flight", Ap, Bp, C, Cref, M, N, K, inv, RR);
    run_rp16<8>("256x128 4w k16 x 6 stages, 4 in flight xcd", Ap, Bp, C, Cref, M, N, K, inv, RR);
    run_rp16<1>("256x128 4w k16 x 6 stages -mfma", Ap, Bp, C, Cref, M, N, K, inv, RR);
    run_rp<16>("256x128 4w reg-pipelined 3st asm-waits", Ap, Bp, C, Cref, M, N, K, inv, RR);
    run_rp<24>("256x128 4w reg-pipelined asm-waits xcd", Ap, Bp, C, Cref, M, N, K, inv, RR);
    run_rp<0>("256x128 4w reg-pipelined 3st", Ap, Bp, C, Cref, M, N, K, inv, RR);
    run_rp<8>("256x128 4w reg-pipelined 3st xcd", Ap, Bp, C, Cref, M, N, K, inv, RR);
    run_rp<1>("256x128 4w reg-pipelined -mfma", Ap, Bp, C, Cref, M, N, K, inv, RR);
    run_pp<0, 1>("256x128 8w ping-pong 3st prio", Ap, Bp, C, Cref, M, N, K, inv, RR);
    run_pp<0, 0>("256x128 8w ping-pong 3st", Ap, Bp, C, Cref, M, N, K, inv, RR);
    run_pp<8, 1>("256x128 8w ping-pong 3st prio xcd", Ap, Bp, C, Cref, M, N, K, inv, RR);
    run_pp<1, 1>("256x128 8w ping-pong -mfma", Ap, Bp, C, Cref, M, N, K, inv, RR);
    run_pp<2, 1>("256x128 8w ping-pong -fragreads", Ap, Bp, C, Cref, M, N, K, inv, RR);
    run_pp<4, 1>("256x128 8w ping-pong -loads", Ap, Bp, C, Cref, M, N, K, inv, RR);
    run_pp<6, 1>("256x128 8w ping-pong mfma+barriers", Ap, Bp, C, Cref, M, N, K, inv, RR);
    RUN(128, 128, 2, 4, 2, 2, 0, "128x128 8w(2x4) 2 stages occ2");
    RUN(128, 128, 2, 4, 2, 2, 8, "128x128 8w(2x4) 2 stages occ2 xcd");
    RUN(128, 128, 2, 2, 2, 2, 0, "128x128 4w(2x2) 2 stages occ2");
    RUN(128, 128, 2, 2, 2, 2, 8, "128x128 4w(2x2) 2 stages occ2 xcd");
    RUN(128, 256, 2, 4, 2, 1, 0, "128x256 8w(2x4) 2 stages occ1");
    RUN(128, 256, 2, 4, 2, 1, 8, "128x256 8w(2x4) 2 stages occ1 xcd");
    RUN(128, 256, 2, 4, 3, 1, 0, "128x256 8w(2x4) 3 stages occ1");
    RUN(128, 256, 2, 4, 3, 1, 8, "128x256 8w(2x4) 3 stages occ1 xcd");
    RUN(256, 128, 4, 2, 3, 1, 0, "256x128 8w(4x2) 3 stages occ1");
    RUN(256, 128, 4, 2, 3, 1, 8, "256x128 8w(4x2) 3 stages occ1 xcd");
    RUN(128, 64, 2, 2, 2, 2, 0, "128x64  4w(2x2) 2 stages occ2");
    RUN(128, 64, 2, 2, 2, 2, 8, "128x64  4w(2x2) 2 stages occ2 xcd");
    RUN(128, 128, 2, 4, 2, 2, 9, "128x128 8w 2st xcd -mfma");
    hipFree(A); hipFree(B); hipFree(Ap); hipFree(Bp); hipFree(C); hipFree(Cref);
  }
  return 0;
}
