// Dev probe: software-pipelined split-bf16 ("x3") GEMM core, C[M,N] = A[M,K] . B[N,K]^T (fp32 in / out).
// One barrier per K step, double-buffered LDS planes (separate __shared__ objects per stage so the compiler can interleave the
// split + LDS stores of tile k+1 with the fragment reads + MFMAs of tile k), global loads two tiles ahead in two register sets.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tris_amd/csrc tools/probes/x3_pipe_probe.hip -o tools/probes/x3_pipe_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "x3_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int BM, int BN, int FBK, int NWM, int NWN, int OCC, int ABL = 0>
__global__ __launch_bounds__(NWM* NWN * 64, (OCC * NWM * NWN + 3) / 4) void pipe_kernel(const float* __restrict__ A,
                                                                                     const float* __restrict__ B,
                                                                                     float* __restrict__ C, int M, int N, int K) {
  constexpr int NW = NWM * NWN, NTHR = NW * 64;
  constexpr int KL = FBK / 4, RPASS = NTHR / KL;
  constexpr int PA = BM / RPASS, PB = BN / RPASS;
  static_assert(PA >= 1 && PB >= 1 && BM % RPASS == 0 && BN % RPASS == 0, "tile / thread mismatch");
  constexpr int PLB = 2 * FBK + 16;
  constexpr int WM = BM / NWM, WN = BN / NWN, FM = WM / 32, FN = WN / 32, G = FBK / 16;
  constexpr int A_ST = 3 * BM * PLB, B_ST = 3 * BN * PLB;
  __shared__ __attribute__((aligned(16))) char As0[A_ST];
  __shared__ __attribute__((aligned(16))) char As1[A_ST];
  __shared__ __attribute__((aligned(16))) char Bs0[B_ST];
  __shared__ __attribute__((aligned(16))) char Bs1[B_ST];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  // ABL & 16: PERSISTENT form -- the grid is one wave of workgroups, each walks tiles blockIdx.x, + gridDim.x, ...; the first
  // global loads of the NEXT tile are issued before the epilogue of the current one (they fly under its stores)
  constexpr bool PERSIST = (ABL & 16) != 0;
  const int tiles_n = (N + BN - 1) / BN;
  const int ntiles = ((M + BM - 1) / BM) * tiles_n;
  int tile = blockIdx.x;
  int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  // rows visited so that the ds_write_b64 groups (16 contiguous lanes) hit disjoint banks (see gemm_fast.h / DESIGN.md)
  int trow;
  if (KL == 8) trow = ((tid >> 3) & ~7) | (((tid >> 3) & 1) << 2) | ((tid >> 4) & 3);
  else { const int u = tid & 31; trow = ((tid >> 5) << 3) | (2 * ((u >> 2) & 3) + (u >> 4)); }
  const int kq = (tid % KL) * 4;
  const float* a_src[PA];
  const float* b_src[PB];
  auto point = [&](int tm0, int tn0) {
#pragma unroll
    for (int q = 0; q < PA; ++q) a_src[q] = A + (long)min(tm0 + trow + q * RPASS, M - 1) * K + kq;
#pragma unroll
    for (int q = 0; q < PB; ++q) b_src[q] = B + (long)min(tn0 + trow + q * RPASS, N - 1) * K + kq;
  };
  point(m0, n0);

  f32x16 acc[FM][FN];
  const int li = lane & 31, kh = lane >> 5;

  float4 r0[PA + PB], r1[PA + PB];  // two register sets: tiles of even / odd index
  auto load = [&](float4 (&r)[PA + PB], int k0) {
    if ((ABL & 4) && k0 > FBK) return;  // no global loads after the first two tiles
#pragma unroll
    for (int q = 0; q < PA; ++q) r[q] = ld4(a_src[q] + k0);
#pragma unroll
    for (int q = 0; q < PB; ++q) r[PA + q] = ld4(b_src[q] + k0);
  };
  auto store_chunk = [&](const float4 (&r)[PA + PB], char* As, char* Bs, int c) {  // c: compile-time after unrolling
    Split4 sp;
    if (ABL & 1) {  // no split VALU: raw bits
      sp.hi = make_uint2(__builtin_bit_cast(unsigned, r[c].x), __builtin_bit_cast(unsigned, r[c].y));
      sp.mid = make_uint2(__builtin_bit_cast(unsigned, r[c].z), __builtin_bit_cast(unsigned, r[c].w));
      sp.lo = sp.hi;
    } else sp = split4(r[c]);
    char* d;
    int pstride;
    if (c < PA) { d = As + (trow + c * RPASS) * PLB + (tid % KL) * 8; pstride = BM * PLB; }
    else { d = Bs + (trow + (c - PA) * RPASS) * PLB + (tid % KL) * 8; pstride = BN * PLB; }
    if (ABL & 2) {  // no LDS stores: keep the values alive without storing
      asm volatile("" ::"v"(sp.hi.x), "v"(sp.hi.y), "v"(sp.mid.x), "v"(sp.mid.y), "v"(sp.lo.x), "v"(sp.lo.y));
      return;
    }
    *reinterpret_cast<uint2*>(d) = sp.hi;
    *reinterpret_cast<uint2*>(d + pstride) = sp.mid;
    *reinterpret_cast<uint2*>(d + 2 * pstride) = sp.lo;
  };
  constexpr int NCH = PA + PB, NPR = FM * FN * G;
  Split8 fixA, fixB;
  {
    const Split8 t = split8(make_float4(lane * 0.01f, 1.f, 2.f, 3.f), make_float4(0.5f, 0.25f, lane, 1.f));
    fixA = t; fixB = t;
  }
  // one K step: MFMAs of the current stage, with the split + store of the next tile's chunks spread between them
  auto step = [&](const char* Ac, const char* Bc, char* An, char* Bn, const float4 (&rs)[PA + PB], bool do_store) {
    int done = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      Split8 sa[FM], sb[FN];
      if (ABL & 8) {  // no fragment reads: operands = loop-invariant register values (MFMA + staging only)
#pragma unroll
        for (int i = 0; i < FM; ++i) { sa[i] = fixA; asm volatile("" : "+v"(sa[i].hi), "+v"(sa[i].mid), "+v"(sa[i].lo)); }
#pragma unroll
        for (int j = 0; j < FN; ++j) { sb[j] = fixB; asm volatile("" : "+v"(sb[j].hi), "+v"(sb[j].mid), "+v"(sb[j].lo)); }
      } else {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const char* s0 = Ac + (wm * WM + i * 32 + li) * PLB + g * 32 + kh * 16;
        sa[i].hi = *reinterpret_cast<const bf16x8*>(s0);
        sa[i].mid = *reinterpret_cast<const bf16x8*>(s0 + BM * PLB);
        sa[i].lo = *reinterpret_cast<const bf16x8*>(s0 + 2 * BM * PLB);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const char* s0 = Bc + (wn * WN + j * 32 + li) * PLB + g * 32 + kh * 16;
        sb[j].hi = *reinterpret_cast<const bf16x8*>(s0);
        sb[j].mid = *reinterpret_cast<const bf16x8*>(s0 + BN * PLB);
        sb[j].lo = *reinterpret_cast<const bf16x8*>(s0 + 2 * BN * PLB);
      }
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].lo, sb[j].hi, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].hi, sb[j].lo, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].mid, sb[j].mid, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].mid, sb[j].hi, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].hi, sb[j].mid, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].hi, sb[j].hi, acc[i][j], 0, 0, 0);
          const int t = (g * FM + i) * FN + j + 1;  // products issued so far
          const int want = (t * NCH) / NPR;
          if (do_store) {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
              if (c >= done && c < want) store_chunk(rs, An, Bn, c);
          }
          done = want;
        }
    }
  };

  // Branch-free main loop: every step issues its prefetch and its stores unconditionally (tile indices are clamped to the
  // last tile, the surplus work of the final steps lands in a stage nobody reads) -- with loads inside a conditional block
  // the compiler's wait-count pass must assume the shorter path and waits for the loads it has just issued.
  const int nk = K / FBK;
  const int klast = (nk - 1) * FBK;
  load(r0, 0);
  for (;;) {
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) store_chunk(r0, As0, Bs0, c);
  load(r1, min(FBK, klast));
  __syncthreads();
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    // even step: current = stage 0 (tile kt); tile kt+1 (in r1) -> stage 1; prefetch tile kt+2 into r0
    load(r0, min((kt + 2) * FBK, klast));
    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch at the top of the step (the scheduler sinks it otherwise)
    step(As0, Bs0, As1, Bs1, r1, true);
    __syncthreads();
    // odd step: current = stage 1 (tile kt+1); tile kt+2 (in r0) -> stage 0; prefetch tile kt+3 into r1
    load(r1, min((kt + 3) * FBK, klast));
    __builtin_amdgcn_sched_barrier(0);
    step(As1, Bs1, As0, Bs0, r0, true);
    __syncthreads();
  }
  if (kt < nk) step(As0, Bs0, As1, Bs1, r1, false);  // odd tile count: the last tile sits in stage 0
  const int cm0 = m0, cn0 = n0;
  const int next = tile + (int)gridDim.x;
  const bool more = PERSIST && next < ntiles;
  if (more) {   // the next tile's first loads fly under this tile's epilogue
    tile = next;
    m0 = (tile / tiles_n) * BM;
    n0 = (tile % tiles_n) * BN;
    point(m0, n0);
    load(r0, 0);
  }

  // plain epilogue (probe): accumulator layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = cn0 + wn * WN + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = cm0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < M && col < N) C[(long)row * N + col] = acc[i][j][r];
      }
    }
  if (!more) break;
  __syncthreads();   // every wave is done with the stages before the next tile's first store
  }
}

// ---- wave-specialised variant: NC = NWM*NWN consumer waves (fragment reads + MFMA only) and NP producer waves (global loads,
// bf16 split, LDS stores only) in one workgroup; two LDS stages, one barrier per K tile.  The consumers' instruction stream
// then carries nothing but ds_read + MFMA, the split VALU runs on the same SIMDs from OTHER waves.
template <int BM, int BN, int FBK, int NWM, int NWN, int NP, int OCC>
__global__ __launch_bounds__((NWM * NWN + NP) * 64, (OCC * (NWM * NWN + NP) + 3) / 4) void pipe_ws_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  constexpr int NC = NWM * NWN, NTP = NP * 64;
  constexpr int KL = FBK / 4, RPASS = NTP / KL;
  constexpr int PA = BM / RPASS, PB = BN / RPASS;
  static_assert(PA >= 1 && PB >= 1 && BM % RPASS == 0 && BN % RPASS == 0, "tile / producer mismatch");
  constexpr int PLB = 2 * FBK + 16;
  constexpr int WM = BM / NWM, WN = BN / NWN, FM = WM / 32, FN = WN / 32, G = FBK / 16;
  constexpr int A_ST = 3 * BM * PLB, B_ST = 3 * BN * PLB;
  __shared__ __attribute__((aligned(16))) char As0[A_ST];
  __shared__ __attribute__((aligned(16))) char As1[A_ST];
  __shared__ __attribute__((aligned(16))) char Bs0[B_ST];
  __shared__ __attribute__((aligned(16))) char Bs1[B_ST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = (N + BN - 1) / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int nk = K / FBK;
  const int klast = (nk - 1) * FBK;
  if (wave >= NC) {
    // ------------------------------------------------------------------ producers
    const int pt = tid - NC * 64;
    int trow;
    if (KL == 8) trow = ((pt >> 3) & ~7) | (((pt >> 3) & 1) << 2) | ((pt >> 4) & 3);
    else { const int u = pt & 31; trow = ((pt >> 5) << 3) | (2 * ((u >> 2) & 3) + (u >> 4)); }
    const int kq = (pt % KL) * 4;
    const float* a_src[PA];
    const float* b_src[PB];
#pragma unroll
    for (int q = 0; q < PA; ++q) a_src[q] = A + (long)min(m0 + trow + q * RPASS, M - 1) * K + kq;
#pragma unroll
    for (int q = 0; q < PB; ++q) b_src[q] = B + (long)min(n0 + trow + q * RPASS, N - 1) * K + kq;
    float4 r0[PA + PB], r1[PA + PB];
    auto load = [&](float4 (&r)[PA + PB], int k0) {
#pragma unroll
      for (int q = 0; q < PA; ++q) r[q] = ld4(a_src[q] + k0);
#pragma unroll
      for (int q = 0; q < PB; ++q) r[PA + q] = ld4(b_src[q] + k0);
    };
    auto store = [&](const float4 (&r)[PA + PB], char* As, char* Bs) {
#pragma unroll
      for (int c = 0; c < PA + PB; ++c) {
        const Split4 sp = split4(r[c]);
        char* d;
        int pstride;
        if (c < PA) { d = As + (trow + c * RPASS) * PLB + (pt % KL) * 8; pstride = BM * PLB; }
        else { d = Bs + (trow + (c - PA) * RPASS) * PLB + (pt % KL) * 8; pstride = BN * PLB; }
        *reinterpret_cast<uint2*>(d) = sp.hi;
        *reinterpret_cast<uint2*>(d + pstride) = sp.mid;
        *reinterpret_cast<uint2*>(d + 2 * pstride) = sp.lo;
      }
    };
    load(r0, 0);
    store(r0, As0, Bs0);
    load(r1, min(FBK, klast));
    load(r0, min(2 * FBK, klast));
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      store(r1, As1, Bs1);                         // tile kt+1 -> stage 1 (consumers read stage 0)
      load(r1, min((kt + 3) * FBK, klast));
      __syncthreads();
      store(r0, As0, Bs0);                         // tile kt+2 -> stage 0 (consumers read stage 1)
      load(r0, min((kt + 4) * FBK, klast));
      __syncthreads();
    }
    return;
  }
  // -------------------------------------------------------------------- consumers
  const int wm = wave / NWN, wn = wave % NWN;
  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int li = lane & 31, kh = lane >> 5;
  auto step = [&](const char* Ac, const char* Bc) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      Split8 sa[FM], sb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const char* s0 = Ac + (wm * WM + i * 32 + li) * PLB + g * 32 + kh * 16;
        sa[i].hi = *reinterpret_cast<const bf16x8*>(s0);
        sa[i].mid = *reinterpret_cast<const bf16x8*>(s0 + BM * PLB);
        sa[i].lo = *reinterpret_cast<const bf16x8*>(s0 + 2 * BM * PLB);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const char* s0 = Bc + (wn * WN + j * 32 + li) * PLB + g * 32 + kh * 16;
        sb[j].hi = *reinterpret_cast<const bf16x8*>(s0);
        sb[j].mid = *reinterpret_cast<const bf16x8*>(s0 + BN * PLB);
        sb[j].lo = *reinterpret_cast<const bf16x8*>(s0 + 2 * BN * PLB);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].lo, sb[j].hi, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].hi, sb[j].lo, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].mid, sb[j].mid, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].mid, sb[j].hi, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].hi, sb[j].mid, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].hi, sb[j].hi, acc[i][j], 0, 0, 0);
        }
    }
  };
  __syncthreads();
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    step(As0, Bs0);
    __syncthreads();
    step(As1, Bs1);
    __syncthreads();
  }
  if (kt < nk) step(As0, Bs0);
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = n0 + wn * WN + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < M && col < N) C[(long)row * N + col] = acc[i][j][r];
      }
    }
}

static void fill(std::vector<float>& v, unsigned seed) {
  unsigned s = seed;
  for (auto& x : v) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffffff) / 8388608.0f - 1.0f; }
}

template <int BM, int BN, int FBK, int NWM, int NWN, int OCC, int ABL = 0>
static void run(const char* name, int M, int N, int K, const float* dA, const float* dB, float* dC, const std::vector<float>& hA,
                const std::vector<float>& hB, bool check) {
  int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  if ((ABL & 16) && tiles > 256 * OCC) tiles = 256 * OCC;   // persistent: one wave of workgroups
  auto launch = [&]() { hipLaunchKernelGGL((pipe_kernel<BM, BN, FBK, NWM, NWN, OCC, ABL>), dim3(tiles), dim3(NWM * NWN * 64), 0, 0, dA, dB, dC, M, N, K); };
  for (int i = 0; i < 3; ++i) launch();
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(hipGetLastError())); return; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int it = 20;
  hipEventRecord(e0);
  for (int i = 0; i < it; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
  double err = 0, scale = 0;
  if (check) {
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    for (int s = 0; s < 48; ++s) {
      const int r = (int)(((long)s * 2654435761u) % M);
      for (int c = 0; c < N; c += 7) {
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * (double)hB[(size_t)c * K + k];
        err = fmax(err, fabs(ref - hC[(size_t)r * N + c]));
        scale = fmax(scale, fabs(ref));
      }
    }
  }
  printf("%-34s M%-6d N%-5d K%-5d %8.1f us %7.1f TF/s   err %.2e (scale %.1f)\n", name, M, N, K, ms * 1e3,
         2.0 * M * N * K / (ms * 1e-3) * 1e-12, err, scale);
  fflush(stdout);
}

template <int BM, int BN, int FBK, int NWM, int NWN, int NP, int OCC>
static void run_ws(const char* name, int M, int N, int K, const float* dA, const float* dB, float* dC, const std::vector<float>& hA,
                   const std::vector<float>& hB, bool check) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  auto launch = [&]() { hipLaunchKernelGGL((pipe_ws_kernel<BM, BN, FBK, NWM, NWN, NP, OCC>), dim3(tiles), dim3((NWM * NWN + NP) * 64), 0, 0, dA, dB, dC, M, N, K); };
  hipMemset(dC, 0, (size_t)M * N * 4);
  for (int i = 0; i < 3; ++i) launch();
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(hipGetLastError())); return; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int it = 20;
  hipEventRecord(e0);
  for (int i = 0; i < it; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
  double err = 0, scale = 0;
  if (check) {
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    for (int s = 0; s < 48; ++s) {
      const int r = (int)(((long)s * 2654435761u) % M);
      for (int c = 0; c < N; c += 7) {
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * (double)hB[(size_t)c * K + k];
        err = fmax(err, fabs(ref - hC[(size_t)r * N + c]));
        scale = fmax(scale, fabs(ref));
      }
    }
  }
  printf("%-34s M%-6d N%-5d K%-5d %8.1f us %7.1f TF/s   err %.2e (scale %.1f)\n", name, M, N, K, ms * 1e3,
         2.0 * M * N * K / (ms * 1e-3) * 1e-12, err, scale);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int shapes[][3] = {{4096, 4096, 4096}, {8192, 8192, 1024}, {19200, 1024, 256}, {19200, 256, 1024}, {4800, 2048, 512},
                           {76800, 512, 128}, {76800, 128, 512}, {307200, 256, 64}, {3840, 2048, 512}, {2400, 3072, 768}};
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    fill(hA, 1); fill(hB, 2);
    float *dA, *dB, *dC;
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    const bool chk = (long)M * N <= 70000000L;
    if (argc > 1 && argv[1][0] == 'w') {  // wave-specialised variants next to the plain pipelined kernel
      run<128, 128, 16, 2, 4, 2>("pipe 128x128 k16 8w occ2", M, N, K, dA, dB, dC, hA, hB, chk);
      run_ws<128, 128, 16, 2, 4, 4, 2>("ws 128x128 k16 8c+4p occ2", M, N, K, dA, dB, dC, hA, hB, chk);
      run_ws<128, 128, 16, 2, 4, 2, 2>("ws 128x128 k16 8c+2p occ2", M, N, K, dA, dB, dC, hA, hB, chk);
      run_ws<128, 128, 32, 2, 4, 4, 1>("ws 128x128 k32 8c+4p occ1", M, N, K, dA, dB, dC, hA, hB, chk);
      run_ws<128, 128, 32, 2, 4, 8, 1>("ws 128x128 k32 8c+8p occ1", M, N, K, dA, dB, dC, hA, hB, chk);
      run_ws<256, 128, 16, 4, 2, 4, 1>("ws 256x128 k16 8c+4p occ1", M, N, K, dA, dB, dC, hA, hB, chk);
      run_ws<256, 256, 16, 2, 4, 4, 1>("ws 256x256 k16 8c+4p occ1", M, N, K, dA, dB, dC, hA, hB, chk);
      run_ws<256, 256, 16, 2, 4, 8, 1>("ws 256x256 k16 8c+8p occ1", M, N, K, dA, dB, dC, hA, hB, chk);
      hipFree(dA); hipFree(dB); hipFree(dC);
      continue;
    }
    if (argc > 1 && argv[1][0] == 'p') {  // persistent form (one wave of workgroups walking the tiles) next to one tile per workgroup
      run<128, 128, 32, 2, 4, 1, 0>("128x128 k32 occ1", M, N, K, dA, dB, dC, hA, hB, chk);
      run<128, 128, 32, 2, 4, 1, 16>("128x128 k32 occ1 persistent", M, N, K, dA, dB, dC, hA, hB, chk);
      run<128, 128, 16, 2, 4, 2, 0>("128x128 k16 occ2", M, N, K, dA, dB, dC, hA, hB, chk);
      run<128, 128, 16, 2, 4, 2, 16>("128x128 k16 occ2 persistent", M, N, K, dA, dB, dC, hA, hB, chk);
      run<256, 128, 16, 4, 2, 1, 0>("256x128 k16 occ1", M, N, K, dA, dB, dC, hA, hB, chk);
      run<256, 128, 16, 4, 2, 1, 16>("256x128 k16 occ1 persistent", M, N, K, dA, dB, dC, hA, hB, chk);
      hipFree(dA); hipFree(dB); hipFree(dC);
      continue;
    }
    if (argc > 1 && argv[1][0] == 'a') {  // ablations on two configurations
      if (!(M == 4096 || (M == 19200 && N == 1024))) { hipFree(dA); hipFree(dB); hipFree(dC); continue; }
#define ABLS(BM_, BN_, FBK_, NWM_, NWN_, OCC_, nm)                                                              \
      run<BM_, BN_, FBK_, NWM_, NWN_, OCC_, 0>(nm " full", M, N, K, dA, dB, dC, hA, hB, false);                      \
      run<BM_, BN_, FBK_, NWM_, NWN_, OCC_, 1>(nm " -split", M, N, K, dA, dB, dC, hA, hB, false);                    \
      run<BM_, BN_, FBK_, NWM_, NWN_, OCC_, 3>(nm " -split-store", M, N, K, dA, dB, dC, hA, hB, false);              \
      run<BM_, BN_, FBK_, NWM_, NWN_, OCC_, 7>(nm " -split-store-load", M, N, K, dA, dB, dC, hA, hB, false);         \
      run<BM_, BN_, FBK_, NWM_, NWN_, OCC_, 15>(nm " mfma+barrier only", M, N, K, dA, dB, dC, hA, hB, false);        \
      run<BM_, BN_, FBK_, NWM_, NWN_, OCC_, 8>(nm " -fragreads", M, N, K, dA, dB, dC, hA, hB, false);                \
      run<BM_, BN_, FBK_, NWM_, NWN_, OCC_, 4>(nm " -load", M, N, K, dA, dB, dC, hA, hB, false);
      ABLS(128, 128, 32, 2, 4, 1, "128x128k32 8w")
      ABLS(128, 128, 16, 2, 4, 2, "128x128k16 8w occ2")
      ABLS(256, 256, 16, 2, 4, 1, "256x256k16 8w")
      hipFree(dA); hipFree(dB); hipFree(dC);
      continue;
    }
    run<128, 128, 32, 2, 4, 1>("128x128 k32 8w(2x4) occ1", M, N, K, dA, dB, dC, hA, hB, chk);
    run<128, 128, 32, 2, 2, 1>("128x128 k32 4w(2x2) occ1", M, N, K, dA, dB, dC, hA, hB, chk);
    run<128, 128, 16, 2, 4, 2>("128x128 k16 8w(2x4) occ2", M, N, K, dA, dB, dC, hA, hB, chk);
    run<128, 128, 16, 2, 2, 2>("128x128 k16 4w(2x2) occ2", M, N, K, dA, dB, dC, hA, hB, chk);
    run<256, 128, 16, 4, 2, 1>("256x128 k16 8w(4x2) occ1", M, N, K, dA, dB, dC, hA, hB, chk);
    run<256, 256, 16, 2, 4, 1>("256x256 k16 8w(2x4) occ1", M, N, K, dA, dB, dC, hA, hB, chk);
    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  return 0;
}
