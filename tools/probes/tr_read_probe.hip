// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read): LDS holds u16 value = its element index; every lane reads
// with a per-lane byte address given by `mode`, and reports the four 16-bit elements it received.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned short* out, int mode) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr;
  if (mode == 0) addr = 0;                                   // uniform address
  else if (mode == 1) addr = (unsigned)l * 8;                // lane-linear 8-byte pieces
  else if (mode == 2) addr = (unsigned)(l & 15) * 2 + (unsigned)(l >> 4) * 128;  // guide: column (l&15) of a [4][16] block per 16-lane group
  else addr = (unsigned)(l & 15) * 8 + (unsigned)(l >> 4) * 512;                 // rows of 4 elements, row stride 8 B
  addr += (unsigned)(size_t)lds;  // LDS base is 0 for the first __shared__ array; keep generic
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
  unsigned short* d;
  hipMalloc(&d, 64 * 4 * 2);
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    std::vector<unsigned short> h(256);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]); if (l == 19 && mode != 2) { l = 47; printf("  ...\n"); } }
  }
  return 0;
}
