// Shared by the GEMM / convolution translation units (gemm_conv.hip, gemm_inst.hip, conv_direct.hip): operand kinds, the launch
// parameter block, a launch configuration and the 16-byte load.  Included INSIDE each unit's anonymous namespace.
#pragma once

enum { A_ROWK = 0, A_COLK = 1, A_IM2COL = 2, A_HALO = 3 };  // A_HALO: direct 3x3 convolution (gemm_fast.h)
enum { B_NK = 0, B_KN = 1, B_KN_DGRAD = 2, B_KN_IM2COL = 3 };
enum { EPI_STD = 0, EPI_SLAB = 1, EPI_XTRA = 2 };   // EPI_XTRA: EPI_STD + the pre_out / dact_x extras (classic loop, row-major A)

struct GemmParams {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  int Kv;           // 0, or the number of VALID k rows when both operands are k-major (A_COLK x B_KN) and K was rounded up to a
                    // multiple of 32 for the fast kernel: its loaders read rows >= Kv as zeros (tris_gemm_f32: weight gradients
                    // over a token count that is not a multiple of 32, e.g. 48 x 401 ViT-B/16 tokens)
  long lda, ldb, ldc;
  long sA, sB, sC;  // batch strides (elements)
  const float* bias;
  int bias_mode;  // 0 none, 1 per column n, 2 per row m
  const float* resid;
  long ldr, sR;
  int act;  // 0 none, 1 relu, 2 quick-gelu
  float alpha;
  int vecA, vecB;  // 16-byte vector loads legal for the operand
  int fastA, fastB;  // operand satisfies the preconditions of gemm_fast_kernel
  int vecC;          // 16-byte epilogue legal: C / resid / bias aligned, N and the leading dimensions multiples of 4 (set by run_cfg)
  double* stat_part;  // optional fused BN statistics partials [tiles_m][2][N] (fast kernel, no split-K)
  int kchunk, splitk;
  int tiles_n;
  // gather geometry (conv): gathered tensor [gB, gH, gW, gC] NHWC, output grid [gB, gHo, gWo], pad 1
  int gH, gW, gC, gHo, gWo, gStride;
  int xcd_remap;    // fast kernel: place all tiles of one split-K slice on one XCD (see gemm_fast.h)
  int gB;           // images in the gathered tensor (B_KN_IM2COL: bounds the running pixel coordinates of surplus prefetches)
  int wCin, wCout;  // weight geometry for B_KN_DGRAD: W[co][tap][ci]
  int hmode;        // A_HALO: window shape, 1 = BM consecutive pixels in padded coordinates, 2 = (BM/16) x 16 patches
  // A_HALO, optional: the gathered tensor is the raw input x of a BatchNorm + ReLU; the kernel forms relu(bn(x)) in its window
  const float *in_mean, *in_invstd, *in_gamma, *in_beta;
  // Optional fused BatchNorm-BACKWARD reduction (EPI_STD, no split-K, with stat_part): C is the gradient of the OUTPUT of a
  // train-mode BatchNorm (+ReLU) whose raw input is bnb_x [M, N] (same leading dimension as C).  The epilogue masks the value
  // with that ReLU -- from the BatchNorm's output bnb_y where given (residual form: y = relu(bn(x) + identity)), else recomputed
  // from bnb_x with bn_apply_kernel's own expression (bnb_gamma / bnb_beta) -- stores the MASKED gradient dz, and accumulates
  // sum(dz), sum(dz * xhat) per column into stat_part [tiles_m][2][N] (fp64 across lanes / waves / tiles), i.e. exactly what
  // col_partial_kernel<1> computes in a pass of its own.
  const float *bnb_x, *bnb_y, *bnb_mean, *bnb_invstd, *bnb_gamma, *bnb_beta;
  int bnb_y_pl;   // 1: bnb_y is an fp16-plane tensor (PREC 4 callers): its ReLU mask is "either piece non-zero"; 2: bnb_y is that mask itself, one
                  // byte per 8 columns (tris_bn_mask_next: written by the forward pass beside the planes)
  // PREC 3 ("h2": two fp16 pieces per operand, three f16 MFMAs per product): per-operand power-of-two scales, either derived in
  // the kernel from the bit pattern of the tensor's largest magnitude -- or of an upper bound of it -- in device memory (h2_amaxA /
  // h2_amaxB: amax words, include/tris_hip.h) or, where that pointer is NULL, given by the host (h2_sA / h2_sB; 0 = 1.0)
  const unsigned *h2_amaxA, *h2_amaxB;
  float h2_sA, h2_sB;
  // optional by-product (tris_gemm_f32 after tris_amax_next): the largest magnitude of the values written to C, maxed into this
  // amax word -- the operand scale of an h2 product that consumes C directly (amax.h); split-K products leave it in the reduce
  unsigned* amax_out;
  // optional (tris_gemm_epilogue_next, EPI_STD, no split-K; same layout as C): pre_out receives the value BEFORE the activation (what
  // a fused QuickGELU's backward needs); dact_x: the stored value is multiplied by quickgelu'(dact_x[m, n]) last -- the data gradient
  // of the Linear behind a QuickGELU comes out as the gradient of its pre-activation
  float* pre_out;
  const float* dact_x;
  int nt;   // EPI_STD vector epilogue: C stores and the residual / bnb_x / bnb_y loads are nontemporal (set per launch: stream_nt)
  // fused split-K finish (EPI_SLAB, fast kernel, batch 1, at most TRIS_FUSE_SPLITK_MAX slices; armed by tris_splitk_tickets_next):
  // one int per output tile, zero between launches; the last-arriving block of a tile sums the slabs and writes Cfin
  // (gemm_fast.h "fused split-K finish"); NULL: the slabs are summed by splitk_reduce_kernel in a launch of its own
  // packed text rows (tris_rows_limit_thread): a device word; tiles whose first row is >= *m_limit leave at once, the split-K reduce
  // skips those rows (row-major A only; the limit is a multiple of every tile height, so a tile is either whole or absent)
  const int* m_limit;
  int* tickets;
  int tickets_n;    // (host: capacity of the armed array; run_cfg keeps `tickets` only where the fused form is launched)
  float* Cfin;
  int vecCfin;      // 16-byte stores / loads legal for Cfin, the residual and the bias
};
#define TRIS_FUSE_SPLITK_MAX 8
#define TRIS_FUSE_SPLITK_DEFAULT 8

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
typedef float gp_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4s(const float* p, bool nt) {   // (nt uniform)
  if (nt) { const gp_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const gp_f32x4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
  return ld4(p);
}
__device__ __forceinline__ void st4s(float* p, float4 v, bool nt) {
  if (nt) { const gp_f32x4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<gp_f32x4*>(p)); }
  else *reinterpret_cast<float4*>(p) = v;
}
inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// one launch configuration of the family: tile, split-K slices, pipe = 1: the pipelined loop of gemm_fast.h (x3 only: two 16-deep
// LDS stages, one barrier per K tile; 256 x 128 tiles exist in that form only).  conv3_dispatch / the direct weight gradient
// store the id of a direct kernel in bm.
// d/dx [x * sigmoid(1.702 x)] with norm.hip's TRIS_EW_QGELU_BWD expression (bit-identical to the unfused pass)
__device__ __forceinline__ float qgelu_grad(float b) {
  const float sg = 1.0f / (1.0f + expf(-1.702f * b));
  return sg + 1.702f * b * sg * (1.f - sg);
}
struct Cfg { int bm, bn, splitk, pipe; };
// do the operands meet the preconditions of gemm_fast_kernel?
inline bool gemm_fast_ok(const GemmParams& p) { return p.fastA && p.fastB && (p.K % 32 == 0) && p.M >= 4 && p.N >= 4; }
