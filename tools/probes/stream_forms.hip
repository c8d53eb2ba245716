// probe: how fast can a BatchNorm-apply-shaped stream (y = relu(bn(x) + r): 2 reads + 1 write, or 1 + 1) go on an idle MI355X, by loop form / grid / cache policy
// build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/probes/stream_forms.hip -o /tmp/sf && /tmp/sf   (-> profiles/r4_stream_forms.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4nt(const float* p) { f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st4nt(float* p, float4 v) { f4v w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<f4v*>(p)); }
__device__ __forceinline__ float4 f(float4 x, float4 r, float4 mu, float4 sc, float4 b) {
  float4 y;
  y.x = fmaxf((x.x - mu.x) * sc.x + b.x + r.x, 0.f); y.y = fmaxf((x.y - mu.y) * sc.y + b.y + r.y, 0.f);
  y.z = fmaxf((x.z - mu.z) * sc.z + b.z + r.z, 0.f); y.w = fmaxf((x.w - mu.w) * sc.w + b.w + r.w, 0.f);
  return y;
}
// FORM 0: grid-stride, U vectors per trip; FORM 1: block-contiguous chunks (each block a contiguous range, threads stride 256 inside)
template <int FORM, int U, bool NTL, bool NTS, bool RES>
__global__ __launch_bounds__(256) void k(const float* __restrict__ X, const float* __restrict__ R, float* __restrict__ Y, const float* __restrict__ cst,
                                         long n4, int C) {
  long stride, i, end;
  if (FORM == 0) { stride = (long)gridDim.x * 256; i = (long)blockIdx.x * 256 + threadIdx.x; end = n4; }
  else { long per = (n4 + gridDim.x - 1) / gridDim.x; per = (per + 255) / 256 * 256; i = blockIdx.x * per + threadIdx.x; end = min(n4, (blockIdx.x + 1) * per); stride = 256; }
  const int c = (int)((i * 4) % C);
  const float4 mu = ld4(cst + c), sc = ld4(cst + C + c), b = ld4(cst + 2 * C + c);
  const float4 z = make_float4(0, 0, 0, 0);
  for (; i + (U - 1) * stride < end; i += U * stride) {
    float4 x[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { x[u] = NTL ? ld4nt(X + (i + u * stride) * 4) : ld4(X + (i + u * stride) * 4); r[u] = RES ? (NTL ? ld4nt(R + (i + u * stride) * 4) : ld4(R + (i + u * stride) * 4)) : z; }
#pragma unroll
    for (int u = 0; u < U; ++u) { float4 y = f(x[u], r[u], mu, sc, b); if (NTS) st4nt(Y + (i + u * stride) * 4, y); else st4(Y + (i + u * stride) * 4, y); }
  }
  for (; i < end; i += stride) { float4 y = f(ld4(X + i * 4), RES ? ld4(R + i * 4) : z, mu, sc, b); st4(Y + i * 4, y); }
}
template <int FORM, int U, bool NTL, bool NTS, bool RES>
void run(const char* name, int grid, const float* X, const float* R, float* Y, const float* cst, long n4, int C) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<FORM, U, NTL, NTS, RES>), dim3(grid), dim3(256), 0, 0, X, R, Y, cst, n4, C);
  hipEventRecord(a);
  const int reps = 20;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<FORM, U, NTL, NTS, RES>), dim3(grid), dim3(256), 0, 0, X, R, Y, cst, n4, C);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double bytes = (double)n4 * 16 * (RES ? 3 : 2);
  printf("%-34s grid %5d  %7.1f us  %6.2f TB/s\n", name, grid, ms * 1e3 / reps, bytes / (ms / reps * 1e-3) / 1e12);
}
__global__ void fill(float* p, long n, unsigned seed) { long i = (long)blockIdx.x * 256 + threadIdx.x; long st = (long)gridDim.x * 256; for (; i < n; i += st) { unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; p[i] = (float)(int)(h & 0xffff) * (1.f / 4096.f) - 8.f; } }
int main() {
  struct S { long M; int C; } shapes[] = {{307200, 256}, {307200, 64}, {76800, 512}, {19200, 1024}};
  for (auto sh : shapes) {
    const long M = sh.M; const int C = sh.C; const long n4 = M * C / 4;
    float *X, *R, *Y, *cst;
    hipMalloc(&X, n4 * 16); hipMalloc(&R, n4 * 16); hipMalloc(&Y, n4 * 16); hipMalloc(&cst, 3 * C * 4);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, X, n4 * 4, 1u); hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, R, n4 * 4, 7u);
    hipLaunchKernelGGL(fill, dim3(4), dim3(256), 0, 0, cst, (long)3 * C, 3u);
    printf("--- M %ld C %d (%.0f MB per tensor), random data\n", M, C, n4 * 16 / 1e6);
    for (int grid : {2048, 4096, 8192, 16384, 32768}) {
      if ((long)grid * 256 > n4) continue;
      run<0, 2, false, false, true>("stride U2 (shipped form) +res", grid, X, R, Y, cst, n4, C);
      run<0, 4, true, true, true>("stride U4 nt +res", grid, X, R, Y, cst, n4, C);
      run<1, 2, true, true, true>("chunk U2 nt +res", grid, X, R, Y, cst, n4, C);
      run<1, 4, true, true, true>("chunk U4 nt +res", grid, X, R, Y, cst, n4, C);
      run<1, 8, true, true, true>("chunk U8 nt +res", grid, X, R, Y, cst, n4, C);
      run<1, 4, true, false, true>("chunk U4 nt-ld only +res", grid, X, R, Y, cst, n4, C);
      run<1, 4, false, true, true>("chunk U4 nt-st only +res", grid, X, R, Y, cst, n4, C);
      run<0, 2, false, false, false>("stride U2 (shipped) 1R+1W", grid, X, R, Y, cst, n4, C);
      run<1, 4, true, true, false>("chunk U4 nt 1R+1W", grid, X, R, Y, cst, n4, C);
    }
    hipFree(X); hipFree(R); hipFree(Y); hipFree(cst);
  }
  return 0;
}
