// Amax by-product of a producing pass (included INSIDE a translation unit's anonymous namespace): the largest magnitude of what a
// kernel WRITES, as a bit pattern, atomically maxed into a caller-zeroed amax word -- the operand scale of an "h2" product that
// consumes the tensor (include/tris_hip.h).  Armed per launch by tris_amax_next() on the calling thread; unarmed launches pass NULL.
#pragma once

__device__ __forceinline__ unsigned abits4(const float4 v) {
  return max(max(__builtin_bit_cast(unsigned, v.x) & 0x7fffffffu, __builtin_bit_cast(unsigned, v.y) & 0x7fffffffu),
             max(__builtin_bit_cast(unsigned, v.z) & 0x7fffffffu, __builtin_bit_cast(unsigned, v.w) & 0x7fffffffu));
}
// An amax "word" is 128 cache lines (8 KB): 8 XCDs x 16 lines, one unsigned used in each.  Device-scope atomics are served
// memory-side (the eight L2s are not coherent) at a few hundred per microsecond -- a launch has thousands of waves.  Instead a
// wave (or, where every thread reaches the end, a block) maxes into a line of ITS XCD with a workgroup-scope atomic, which the
// XCD's own L2 serves (all CUs of an XCD share it, so every line is exact for what its writers saw); sixteen lines per XCD keep
// the queue per address short.  The dirty lines reach memory at the end of the kernel like any other output, and the consumer
// takes the max over the 128 (x3_split.h h2_amax_of).  Writers that would not raise their line skip the atomic.
__device__ __forceinline__ void amax_raise(unsigned m, unsigned* __restrict__ out) {   // one lane
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;   // HW_REG_XCC_ID, bits [3:0]
  unsigned* w = out + (xcc * 16 + (((blockIdx.x + blockIdx.y) >> 3) & 15)) * 16;
  if (m > __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
    __hip_atomic_fetch_max(w, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void amax_commit(unsigned m, unsigned* __restrict__ out) {   // per wave; no barrier
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, sft, 64));
  if ((threadIdx.x & 63) == 0 && m != 0u) amax_raise(m, out);
}
__device__ __forceinline__ void amax_commit_block(unsigned m, unsigned* __restrict__ out) {   // ALL threads of a <= 256-thread block
  __shared__ unsigned sh_amax[4];
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, sft, 64));
  if ((threadIdx.x & 63) == 0) sh_amax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = max(m, sh_amax[w]);
    if (m != 0u) amax_raise(m, out);
  }
}
