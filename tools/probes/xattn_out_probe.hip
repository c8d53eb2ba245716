// Dev probe: ablations of the cross-attention OUTPUTS kernel (xattn_out_x3_kernel<ABL>) at the Stage-1 shape.
// ABL bits: 1 no Vv global loads, 2 no soft-maxes, 4 no output stores, 8 no bf16 split of the probability fragments.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Itris_amd/csrc tools/probes/xattn_out_probe.hip -o tools/probes/xattn_out_probe
extern "C" int tris_get_gemm_mode(void) { return 1; }
#include "../../tris_amd/csrc/xattn.hip"
#include <cstdio>
#include <vector>
template <int ABL>
static void run(const char* name, const float* Vv, const float* Vt, float* probs, float* nv, float* nl, int B, int P, int N, int C) {
  const size_t lds = (size_t)(((P + 15) / 16 * 16) * XLD + 64 * XLT) * sizeof(float);
  hipFuncSetAttribute((const void*)xattn_out_x3_kernel<ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  auto launch = [&]() { hipLaunchKernelGGL(xattn_out_x3_kernel<ABL>, dim3(C / 128, B), dim3(256), lds, 0, Vv, Vt, probs, nv, nl, P, N, C); };
  for (int i = 0; i < 5; ++i) launch();
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s %7.1f us\n", name, ms / 50 * 1e3); fflush(stdout);
}
int main() {
  const int B = 48, P = 100, N = 48, C = 1024;
  std::vector<float> h((size_t)B * P * C);
  unsigned s = 1; for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffffff) / 8388608.0f - 1.0f; }
  float *Vv, *Vt, *probs, *nv, *nl;
  hipMalloc(&Vv, h.size() * 4); hipMalloc(&Vt, (size_t)N * C * 4); hipMalloc(&probs, (size_t)B * 4 * P * N * 4);
  hipMalloc(&nv, h.size() * 4); hipMalloc(&nl, (size_t)B * N * C * 4);
  hipMemcpy(Vv, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(Vt, h.data(), (size_t)N * C * 4, hipMemcpyHostToDevice);
  hipMemcpy(probs, h.data(), (size_t)B * 4 * P * N * 4, hipMemcpyHostToDevice);
  run<0>("full", Vv, Vt, probs, nv, nl, B, P, N, C);
  run<1>("- Vv loads", Vv, Vt, probs, nv, nl, B, P, N, C);
  run<2>("- softmaxes", Vv, Vt, probs, nv, nl, B, P, N, C);
  run<4>("- stores", Vv, Vt, probs, nv, nl, B, P, N, C);
  run<8>("- probability split", Vv, Vt, probs, nv, nl, B, P, N, C);
  run<3>("- Vv loads - softmaxes", Vv, Vt, probs, nv, nl, B, P, N, C);
  run<7>("- Vv loads - softmaxes - stores", Vv, Vt, probs, nv, nl, B, P, N, C);
  run<15>("- everything but logits fill + MFMA", Vv, Vt, probs, nv, nl, B, P, N, C);
  run<0>("full (again)", Vv, Vt, probs, nv, nl, B, P, N, C);
  return 0;
}
