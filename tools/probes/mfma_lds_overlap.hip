// Dev probe: do v_mfma_f32_32x32x16_f16 and ds_read_b128 / LDS-DMA overlap on gfx950 at all?  One workgroup per CU (4 waves, one per SIMD,
// 512 registers each), no barriers, no dependence between the matrix work and what is read:
//   A  24 MFMAs per trip (8 accumulators in AGPRs, operands in registers)                          -> the matrix pipe alone
//   B  12 ds_read_b128 per trip (inline asm, results consumed by a final xor so that nothing is dropped) -> LDS reads alone
//   C  both per trip, the reads issued in front of the MFMAs, one s_waitcnt lgkmcnt(12) per trip     -> do they overlap?
//   D  6 global_load_lds_dwordx4 per trip (a 24 KB ring in LDS, L2-resident source)               -> LDS-DMA alone
//   E  A + D, F  A + B + D
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_lds_overlap.hip -o tools/probes/mfma_lds_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int MODE, int NT>
__global__ __launch_bounds__(NT, NT / 256) void k(const float* __restrict__ src, float* __restrict__ out, int trips) {
  __shared__ __attribute__((aligned(1024))) char lds[(NT / 64) * 8 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < (NT / 64) * 8 * 1024 / 16; i += NT) reinterpret_cast<u4*>(lds)[i] = (u4){1u, 2u, 3u, (unsigned)i};
  __syncthreads();
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a[4], b[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = (f16x8)(_Float16)(0.001f * (lane + i));
#pragma unroll
  for (int j = 0; j < 2; ++j) b[j] = (f16x8)(_Float16)(0.5f + j);
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)lds + wave * (8 * 1024) + lane * 16;
  u4 f[12], x = (u4){0u, 0u, 0u, 0u};
#pragma unroll
  for (int q = 0; q < 12; ++q) f[q] = (u4){0u, 0u, 0u, 0u};
  const char* g = reinterpret_cast<const char*>(src) + ((size_t)blockIdx.x * NT + tid) * 16;
  if constexpr ((MODE & 8) != 0) {
    // FINE INTERLEAVE (round 6; VERDICT r5: the rows above issue all reads, wait, then all MFMAs -- a serial schedule): the same work per
    // trip with ONE ds_read_b128 in front of every PAIR of MFMAs and one LDS-DMA piece per FOUR, the order pinned by sched_barrier; a
    // fragment is consumed one trip after it was read (11 younger reads outstanding -> lgkmcnt(11)); DMA three trips deep (vmcnt(18)).
    for (int t = 0; t < trips; ++t) {
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        if (MODE & 2) {
          asm volatile("s_waitcnt lgkmcnt(11)" : "+v"(f[q]));
          x ^= f[q];
          asm volatile("ds_read_b128 %0, %1" : "=v"(f[q]) : "v"(base + (unsigned)(((t + q) & 7) * 1024)));
        }
        if ((MODE & 4) && (q & 1) == 0) {
          const int d = q >> 1;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)((t * 6 + d) & 63) * 65536 * 4),
                                           (__attribute__((address_space(3))) void*)(lds + wave * 8192 + ((t * 6 + d) % 8) * 1024), 16, 0, 0);
          if (d == 5) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 1) {
          const int i = q >> 2, j = (q >> 1) & 1, h = q & 1;    // 12 pairs = the 24 MFMAs of a trip, consecutive ones on different accumulators
          acc[(2 * q) & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + h) & 3], b[j], acc[(2 * q) & 7], 0, 0, 0);
          acc[(2 * q + 1) & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + 2) & 3], b[j ^ h], acc[(2 * q + 1) & 7], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else
  for (int t = 0; t < trips; ++t) {
    if (MODE & 4) {   // LDS-DMA: 6 x 1 KB per wave and trip into the wave's 24 KB ring
#pragma unroll
      for (int q = 0; q < 6; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)((t * 6 + q) & 63) * 65536 * 4),
                                         (__attribute__((address_space(3))) void*)(lds + wave * 8192 + ((t * 6 + q) % 8) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(18)" ::: "memory");   // three trips of DMA stay in flight
    }
    if (MODE & 2) {
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        x ^= f[q];                                          // consume what the previous trip read
        asm volatile("ds_read_b128 %0, %1" : "=v"(f[q]) : "v"(base + (unsigned)(((t + q) & 7) * 1024)));
      }
      if (MODE & 1) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE & 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
          acc[(i * 2 + j + 4) & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + 1) & 3], b[j], acc[(i * 2 + j + 4) & 7], 0, 0, 0);
          acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + 2) & 3], b[j ^ 1], acc[i * 2 + j], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
#pragma unroll
  for (int q = 0; q < 12; ++q) x ^= f[q];
  out[(size_t)blockIdx.x * NT + tid] = s + (float)(x[0] ^ x[1] ^ x[2] ^ x[3]) + (float)lds[tid];
}

template <int MODE, int NT> static void run(const char* name, const float* src, float* out) {
  const int trips = 20000, grid = 256;
  hipLaunchKernelGGL((k<MODE, NT>), dim3(grid), dim3(NT), 0, 0, src, out, 100);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<MODE, NT>), dim3(grid), dim3(NT), 0, 0, src, out, trips);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns_trip = ms * 1e6 / trips;
  printf("%d waves/SIMD  %-40s %8.1f ns per trip", NT / 256, name, ns_trip);
  if (MODE & 1) printf("   MFMA %6.1f TFLOP/s (f16 dense, whole chip)", 24.0 * 32768 * (NT / 64) * grid / ns_trip * 1e-3);
  if (MODE & 2) printf("   LDS reads %5.1f B/clk/CU at 2.4 GHz", 12.0 * 1024 * (NT / 64) / ns_trip / 2.4);
  if (MODE & 4) printf("   LDS-DMA %5.1f B/clk/CU", 6.0 * 1024 * (NT / 64) / ns_trip / 2.4);
  printf("\n");
}

int main() {
  float *src, *out;
  hipMalloc(&src, (size_t)64 * 65536 * 4 * 4 + (1 << 22));
  hipMemset(src, 0, (size_t)64 * 65536 * 4 * 4 + (1 << 22));
  hipMalloc(&out, 256 * 1024 * 4);
#define ALL(NT)                                                       \
  run<1, NT>("A  24 MFMA 32x32x16 f16 per trip", src, out);       \
  run<2, NT>("B  12 ds_read_b128 per trip", src, out);            \
  run<3, NT>("C  A + B (reads in front, lgkmcnt(12))", src, out); \
  run<4, NT>("D  6 LDS-DMA x 1 KB per trip", src, out);           \
  run<5, NT>("E  A + D", src, out);                               \
  run<6, NT>("   B + D", src, out);                               \
  run<7, NT>("F  A + B + D", src, out);                            \
  run<11, NT>("C' A + B, 1 read per 2 MFMAs (interleaved)", src, out);   \
  run<13, NT>("E' A + D, 1 DMA per 4 MFMAs (interleaved)", src, out);    \
  run<15, NT>("F' A + B + D interleaved", src, out);
  ALL(256) ALL(512)
  return 0;
}
