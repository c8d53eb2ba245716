// Normalisation, pooling and elementwise kernels on channels-last activations [M = B*H*W, C] (gfx950).
// All of these are HBM-bound streaming kernels: 16-byte loads per lane along the channel axis, per-channel
// reductions done as deterministic two-stage (per-block partials -> fixed-order finalize), no atomics.
#include <cstdlib>

#include "common.h"
#include "tris_hip.h"

extern "C" {
// options (tris_set_option, gemm_conv.hip): 1 = element-wise passes over more than the memory-side cache take the streaming form
__attribute__((visibility("hidden"))) int tris_internal_stream_form = 256;   // MB; 0 = never
__attribute__((visibility("hidden"))) int tris_internal_col_blocks = 512;   // blocks a column reduction aims for
}

namespace {

#include "amax.h"

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// Streaming ("BIG") form of the element-wise passes.  Measured on an idle MI355X (tools/probes/stream_forms.hip ->
// profiles/r4_stream_forms.txt; y = relu(bn(x) + r), M = 307200, C = 256: three tensors of 315 MB): the grid-stride walk below
// with the default cache policy moves 4.7-5.0 TB/s; ONE contiguous piece of 1024 vectors per block with four nontemporal loads
// per stream in flight and nontemporal stores moves 6.9-7.2 TB/s (either change alone: 5.3-5.7).  Tensors that fit the 256 MB
// memory-side cache together (3 x 79 MB) run at 6.5-7.0 TB/s in the default form, are no faster in this one -- and their
// consumer finds them in that cache -- so the launchers pick this form only when the streams of a launch exceed 256 MB.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4nt(const float* p) {
  const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4nt(float* p, float4 v) {
  const f32x4_t w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<f32x4_t*>(p));
}
constexpr int BIG_U = 4;                       // vectors per thread and stream
constexpr long BIG_PIECE = 256L * BIG_U;       // vectors per block
// (a thread of the BIG form sees ONE channel vector for its BIG_U vectors iff 256 * 4 floats is a multiple of C)
inline bool big_form(long n4, int C, int streams) { return tris_internal_stream_form && C >= 4 && 1024 % C == 0 && n4 * 16 * streams > ((long)tris_internal_stream_form << 20) && n4 >= 4 * BIG_PIECE; }
inline int big_grid(long n4) { return (int)((n4 + BIG_PIECE - 1) / BIG_PIECE); }

// ------------------------------------------------------------------------------------------------------
// Column (per-channel) partial reductions.  Block = 256 threads covering CVB = min(C/4,256) channel-vectors
// x RS = 256/CVB rows at a time.  MODE 0: shifted sum / sum of squares of X (BN forward statistics).
// MODE 1: sum(dz), sum(dz * xhat) with dz = dY * (Y > 0 if Y given)  (BN backward).  MODE 2: plain column sum.
// part layout: [nblocks][2][C].
// ------------------------------------------------------------------------------------------------------
struct d4 { double x, y, z, w; };
__device__ __forceinline__ d4 d4zero() { d4 r; r.x = r.y = r.z = r.w = 0.0; return r; }
__device__ __forceinline__ void st4d(double* p, d4 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w; }

// Accumulators are fp64: these reductions feed train-mode BatchNorm, whose backward cancels large terms, and the
// CPU oracle (ATen) accumulates them in double as well.  fp64 VALU adds are free next to the HBM stream.
// POOL (MODE 1): dY is the gradient of avgpool2(relu(bn(X))) -- [B, H/2, W/2, C] against X [B, H, W, C] (pool_h = H, pool_w = W):
// the gradient of a row is a quarter of its pooled pixel's, read in place of a full-size tensor that is never written.
__device__ __forceinline__ long pooled_row(long r, int H, int W) {
  const int w = (int)(r % W);
  const long t = r / W;
  const int h = (int)(t % H);
  const long b = t / H;
  return (b * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1);
}

template <int MODE, bool POOL = false>
__global__ __launch_bounds__(256) void col_partial_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                          const float* __restrict__ Y, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, long M, int C, long ld,
                                                          long rows_per_block, double* __restrict__ part,
                                                          const float* __restrict__ gamma = nullptr,
                                                          const float* __restrict__ beta = nullptr,
                                                          float* __restrict__ dZ = nullptr, int pool_h = 0, int pool_w = 0,
                                                          int y_pl = 0, unsigned* __restrict__ amax = nullptr) {
  // y_pl (MODE 1): Y is an fp16-plane tensor (csrc/planes.h): y = relu(..) > 0 <=> one of its two pieces is non-zero
  // amax (MODE 1): by-product, the largest magnitude of the masked gradient dz (the bound of the BatchNorm's dx needs it)
  unsigned am = 0u;
  auto ldy = [&](long o) -> float4 {
    if (!y_pl) return ld4(Y + o);
    const char* yb = reinterpret_cast<const char*>(Y + (o & ~7L)) + (o & 4) * 2;
    const uint2 h = *reinterpret_cast<const uint2*>(yb), l = *reinterpret_cast<const uint2*>(yb + 16);
    const unsigned m0 = (h.x | l.x) & 0x7fff7fffu, m1 = (h.y | l.y) & 0x7fff7fffu;
    return make_float4((m0 & 0xffffu) ? 1.f : 0.f, (m0 >> 16) ? 1.f : 0.f, (m1 & 0xffffu) ? 1.f : 0.f, (m1 >> 16) ? 1.f : 0.f);
  };
  __shared__ d4 l0[256];
  __shared__ d4 l1[256];
  const int CV = C >> 2;
  const int CVB = CV < 256 ? CV : 256;
  const int RS = 256 / CVB;
  const int tid = threadIdx.x;
  const int cvl = tid % CVB, ro = tid / CVB;
  const long rbeg = (long)blockIdx.x * rows_per_block;
  const long rend = min(M, rbeg + rows_per_block);
  for (int cv0 = 0; cv0 < CV; cv0 += CVB) {
    const int c = (cv0 + cvl) * 4;
    d4 s0 = d4zero(), s1 = d4zero();
    float4 mu = make_float4(0, 0, 0, 0), is = mu, sc = mu, be = mu;
    const bool act = (ro < RS) && (cv0 + cvl < CV);
    // MODE 1, gamma != NULL: the ReLU mask is recomputed from X with bn_apply_kernel's own expression (same sign bit for
    // bit) instead of being read from Y -- one 4-byte stream less for every BatchNorm+ReLU without a residual input
    const bool mask_x = (MODE == 1) && (gamma != nullptr);
    if (act) {
      if (MODE == 1) { mu = ld4(mean + c); is = ld4(invstd + c); }
      if (mask_x) {
        const float4 ga = ld4(gamma + c);
        be = ld4(beta + c);
        sc = make_float4(is.x * ga.x, is.y * ga.y, is.z * ga.z, is.w * ga.w);
      }
      // MODE 1, dZ != NULL: the masked gradient dz is also WRITTEN (it is the gradient of the fused residual branch): the apply
      // pass then reads dz and X only -- one activation-sized stream less over the two passes than masking twice from Y
      auto accum = [&](const float4 x, float4 g, const float4 y, long o) {
        if (MODE == 0) {
          s0.x += x.x; s0.y += x.y; s0.z += x.z; s0.w += x.w;
          s1.x += (double)x.x * x.x; s1.y += (double)x.y * x.y; s1.z += (double)x.z * x.z; s1.w += (double)x.w * x.w;
        } else if (MODE == 1) {
          if (mask_x) {
            if (!((x.x - mu.x) * sc.x + be.x > 0.f)) g.x = 0.f;
            if (!((x.y - mu.y) * sc.y + be.y > 0.f)) g.y = 0.f;
            if (!((x.z - mu.z) * sc.z + be.z > 0.f)) g.z = 0.f;
            if (!((x.w - mu.w) * sc.w + be.w > 0.f)) g.w = 0.f;
          } else if (Y) {
            if (!(y.x > 0.f)) g.x = 0.f;
            if (!(y.y > 0.f)) g.y = 0.f;
            if (!(y.z > 0.f)) g.z = 0.f;
            if (!(y.w > 0.f)) g.w = 0.f;
          }
          if (dZ) st4(dZ + o, g);
          am = max(am, abits4(g));
          s0.x += g.x; s0.y += g.y; s0.z += g.z; s0.w += g.w;
          s1.x += (double)(g.x * ((x.x - mu.x) * is.x)); s1.y += (double)(g.y * ((x.y - mu.y) * is.y));
          s1.z += (double)(g.z * ((x.z - mu.z) * is.z)); s1.w += (double)(g.w * ((x.w - mu.w) * is.w));
        } else {
          s0.x += x.x; s0.y += x.y; s0.z += x.z; s0.w += x.w;
        }
      };
      const float4 z4 = make_float4(0, 0, 0, 0);
      long r = rbeg + ro;
      // 4 rows per trip: up to 12 independent 16-byte loads in flight per lane (these kernels are pure HBM streams)
      for (; r + 3 * RS < rend; r += 4 * RS) {
        float4 x[4], g[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long o = (r + (long)u * RS) * ld + c;
          x[u] = ld4(X + o);
          if (POOL) {
            g[u] = ld4(dY + pooled_row(r + (long)u * RS, pool_h, pool_w) * ld + c);
            g[u].x *= 0.25f; g[u].y *= 0.25f; g[u].z *= 0.25f; g[u].w *= 0.25f;
          } else {
            g[u] = (MODE == 1) ? ld4(dY + o) : z4;
          }
          y[u] = (MODE == 1 && Y) ? ldy(o) : z4;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) accum(x[u], g[u], y[u], (r + (long)u * RS) * ld + c);
      }
      for (; r < rend; r += RS) {
        const long o = r * ld + c;
        float4 g1 = z4;
        if (POOL) {
          g1 = ld4(dY + pooled_row(r, pool_h, pool_w) * ld + c);
          g1.x *= 0.25f; g1.y *= 0.25f; g1.z *= 0.25f; g1.w *= 0.25f;
        } else if (MODE == 1) {
          g1 = ld4(dY + o);
        }
        accum(ld4(X + o), g1, (MODE == 1 && Y) ? ldy(o) : z4, o);
      }
    }
    __syncthreads();
    l0[tid] = s0;
    l1[tid] = s1;
    __syncthreads();
    if (ro == 0 && cv0 + cvl < CV) {
      for (int q = 1; q < RS; ++q) {
        d4 a = l0[q * CVB + cvl], b = l1[q * CVB + cvl];
        s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
      }
      double* o = part + (long)blockIdx.x * 2 * C;
      st4d(o + c, s0);
      if (MODE != 2) st4d(o + C + c, s1);
    }
  }
  if (MODE == 1 && amax != nullptr) amax_commit(am, amax);   // (per wave; every thread arrives here)
}

// Partial-sum finalizers.  Block = 4 channels x (blockDim/4) slices of the nb partial rows, reduced through LDS in a
// fixed order (deterministic).  The partial rows lie 2*C doubles apart and were written by other XCDs, so every load is
// a long-latency miss: what matters is how many are in flight -- C/4 blocks (16..512), and each thread issues its loads
// in independent batches of 8 before the dependent fp64 adds.  nb <= 512 (col_partial plans): 64 slices, <= 8 loads per
// thread; conv-epilogue statistics (one row per 128-row GEMM tile, up to ~10^4 rows): 1024-thread blocks, 256 slices.
constexpr int FIN_CH = 4;
template <typename T>
__device__ __forceinline__ void reduce_parts(const T* __restrict__ part, int nb, int C, bool two, double& s, double& ss,
                                             int& c, bool& lead) {
  __shared__ double sh0[1024];
  __shared__ double sh1[1024];
  const int SL = blockDim.x / FIN_CH;
  const int cl = threadIdx.x % FIN_CH, j = threadIdx.x / FIN_CH;
  c = blockIdx.x * FIN_CH + cl;
  s = 0.0;
  ss = 0.0;
  if (c < C) {
    constexpr int U = 8;
    const T* p0 = part + c;
    for (int b0 = j; b0 < nb; b0 += SL * U) {
      T v0[U], v1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int b = b0 + u * SL;
        const bool ok = b < nb;
        v0[u] = ok ? p0[(long)b * 2 * C] : (T)0;
        v1[u] = (ok && two) ? p0[(long)b * 2 * C + C] : (T)0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { s += (double)v0[u]; ss += (double)v1[u]; }
    }
  }
  sh0[threadIdx.x] = s;
  sh1[threadIdx.x] = ss;
  __syncthreads();
  lead = (j == 0) && (c < C);
  if (lead) {
    s = 0.0;
    ss = 0.0;
    for (int q = 0; q < SL; ++q) { s += sh0[q * FIN_CH + cl]; ss += sh1[q * FIN_CH + cl]; }
  }
}
inline int fin_grid(int C, int nb) { (void)nb; return cdiv(C, FIN_CH); }
// (FIN_BLOCK option: threads of a finalizer block for more than 2048 partial rows.  Rounds 1-3 used 1024 there -- more loads in flight on
//  an idle device; but a block of sixteen waves has to find sixteen free wave slots on one CU, which takes time inside the step:
//  256 everywhere is 0.2 ms per step faster, 39.30 -> 39.10 ms, A/B x3 on one box)
extern "C" { __attribute__((visibility("hidden"))) int tris_internal_fin_block = 256; }
inline int fin_block(int nb) { return nb > 2048 ? tris_internal_fin_block : 256; }

// BN forward finalize: stats[0]=mean, stats[1]=invstd, stats[2]=biased var; optional running-stat update.
// bound_out != NULL (operand planes, csrc/planes.hip): the launch also leaves the Samuelson bound of the BatchNorm's OUTPUT, max_c
// |gamma_c| xhat_max + |beta_c| (+ the value of *add_word: the bound of the residual it is added to), in the amax word bound_out --
// tris_bn_out_bound2_f32 without a launch of its own (the bound depends on the parameters alone; it rides here because this launch
// exists anyway and precedes the pass that needs the word)
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const double* __restrict__ part, int nb, long M, int C,
                                                          float eps, float momentum, float* __restrict__ stats,
                                                          float* running_mean, float* running_var,
                                                          const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr,
                                                          float xhat_max = 0.f, const unsigned* __restrict__ add_word = nullptr,
                                                          unsigned* __restrict__ bound_out = nullptr) {
  __shared__ float sh_add;
  if (bound_out != nullptr && threadIdx.x < 64) {
    const unsigned v = add_word != nullptr ? max(add_word[threadIdx.x * 16], add_word[(threadIdx.x + 64) * 16]) : 0u;
    unsigned m = v;
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, sft, 64));
    if (threadIdx.x == 0) sh_add = __builtin_bit_cast(float, m);
  }
  double s, ss;
  int c;
  bool lead;
  reduce_parts(part, nb, C, true, s, ss, c, lead);
  if (!lead) return;
  if (bound_out != nullptr) amax_raise(__builtin_bit_cast(unsigned, fabsf(gamma[c]) * xhat_max + fabsf(beta[c]) + sh_add), bound_out);
  double mean = s / (double)M;
  double var = ss / (double)M - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[c] = (float)mean;
  stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
  stats[2 * C + c] = (float)var;
  if (running_mean) {
    double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

// Sum partials [nb][2][C] -> out0[C] (and out1[C] if given).
// bound_out != NULL (operand planes; the sums are sum(dz), sum(dz xhat) of a BatchNorm backward): the launch also leaves the bound of
// that BatchNorm's dx, max_c |gamma invstd| (amax_dz + |sum_dz| / n + xhat_max |sum_dzx| / n), in the amax word bound_out
// (tris_bn_bwd_bound_f32 without a launch of its own); *dz_word: the amax of the masked gradient, left by the pass that made the partials
template <typename T>
__global__ __launch_bounds__(1024) void part_finalize_kernel(const T* __restrict__ part, int nb, int C,
                                                            float* __restrict__ out0, float* __restrict__ out1,
                                                            const float* __restrict__ gamma = nullptr, const float* __restrict__ invstd = nullptr,
                                                            float inv_cnt = 0.f, float xhat_max = 0.f,
                                                            const unsigned* __restrict__ dz_word = nullptr,
                                                            unsigned* __restrict__ bound_out = nullptr) {
  __shared__ float sh_adz;
  if (bound_out != nullptr && threadIdx.x < 64) {
    unsigned m = max(dz_word[threadIdx.x * 16], dz_word[(threadIdx.x + 64) * 16]);
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, sft, 64));
    if (threadIdx.x == 0) sh_adz = __builtin_bit_cast(float, m);
  }
  double s, ss;
  int c;
  bool lead;
  reduce_parts(part, nb, C, out1 != nullptr, s, ss, c, lead);
  if (!lead) return;
  out0[c] = (float)s;
  if (out1) out1[c] = (float)ss;
  if (bound_out != nullptr)
    amax_raise(__builtin_bit_cast(unsigned, fabsf(gamma[c] * invstd[c]) * (sh_adz + fabsf((float)s) * inv_cnt + xhat_max * fabsf((float)ss) * inv_cnt)),
               bound_out);
}

// SyncBN: combine the per-rank statistics blocks [mean | invstd | biased var] (3*C floats each, exactly what
// bn_finalize_kernel writes, so the all_gather needs no repacking) -> global mean / invstd / var, running-stat update.
// Every rank contributes `count` rows (DistributedSampler gives equal per-rank batches).
__global__ void bn_sync_combine_kernel(const float* __restrict__ gathered, int Wn, int C, float count, float eps,
                                       float momentum, float* __restrict__ stats, float* running_mean,
                                       float* running_var) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double mean = 0.0;
  for (int w = 0; w < Wn; ++w) mean += (double)gathered[(long)w * 3 * C + c];
  mean /= (double)Wn;
  double m2 = 0.0;
  for (int w = 0; w < Wn; ++w) {
    const float* g = gathered + (long)w * 3 * C;
    double d = g[c] - mean;
    m2 += (double)g[2 * C + c] + d * d;
  }
  double var = m2 / (double)Wn;
  double n = (double)count * Wn;
  stats[c] = (float)mean;
  stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
  stats[2 * C + c] = (float)var;
  if (running_mean) {
    double unb = n > 1.0 ? var * (n / (n - 1.0)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

// (amax by-product of the passes below: amax.h)
// y = (x - mean) * invstd * gamma + beta (+ resid) (relu)
// The launch keeps gridDim*blockDim a multiple of C/4, so a thread sees ONE channel vector for its whole grid-stride walk:
// the per-channel constants are folded to (scale, shift) once and the loop is a pure 16-byte stream, two vectors per trip.
template <bool BIG>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ X, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ resid,
                                                       float* __restrict__ Y, long n4, int C, int relu,
                                                       unsigned* __restrict__ amax = nullptr) {
  const long stride = (long)gridDim.x * blockDim.x;
  long i = BIG ? (long)blockIdx.x * BIG_PIECE + threadIdx.x : (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = (int)((i * 4) % C);
  const float4 mu = ld4(mean + c), is = ld4(invstd + c), g = ld4(gamma + c), b = ld4(beta + c);
  const float4 sc = make_float4(is.x * g.x, is.y * g.y, is.z * g.z, is.w * g.w);
  unsigned am = 0u;
  auto one = [&](const float4 x, const float4 r) {
    float4 y;
    y.x = (x.x - mu.x) * sc.x + b.x + r.x;
    y.y = (x.y - mu.y) * sc.y + b.y + r.y;
    y.z = (x.z - mu.z) * sc.z + b.z + r.z;
    y.w = (x.w - mu.w) * sc.w + b.w + r.w;
    if (relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
    am = max(am, abits4(y));
    return y;
  };
  const float4 z4 = make_float4(0, 0, 0, 0);
  if (BIG) {
    float4 x[BIG_U], r[BIG_U];
#pragma unroll
    for (int u = 0; u < BIG_U; ++u) {
      const long j = i + u * 256;
      x[u] = j < n4 ? ld4nt(X + j * 4) : z4;
      r[u] = (resid && j < n4) ? ld4nt(resid + j * 4) : z4;
    }
#pragma unroll
    for (int u = 0; u < BIG_U; ++u) {
      const long j = i + u * 256;
      if (j < n4) st4nt(Y + j * 4, one(x[u], r[u]));
    }
  } else {
    for (; i + stride < n4; i += 2 * stride) {
      const float4 x0 = ld4(X + i * 4), x1 = ld4(X + (i + stride) * 4);
      const float4 r0 = resid ? ld4(resid + i * 4) : z4, r1 = resid ? ld4(resid + (i + stride) * 4) : z4;
      st4(Y + i * 4, one(x0, r0));
      st4(Y + (i + stride) * 4, one(x1, r1));
    }
    if (i < n4) st4(Y + i * 4, one(ld4(X + i * 4), resid ? ld4(resid + i * 4) : z4));
  }
  if (amax != nullptr) amax_commit_block(am, amax);
}

// Yp[b, oy, ox, :] = avgpool2(relu(bn(X)))  -- the stem's bn3 and the stride-2 Bottlenecks' bn2 feed an AvgPool2d(2) and nothing
// else: the full-size activation is never written (avgpool2_fwd_kernel's own expression on the four normalised pixels)
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(const float* __restrict__ X, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ Yp, int B, int H,
                                                            int W, int C, unsigned* __restrict__ amax = nullptr) {
  const int Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
  const long n = (long)B * Ho * Wo * C4;
  const long stride = (long)gridDim.x * blockDim.x;
  unsigned am = 0u;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int c = (int)(i % C4) * 4;
    long t = i / C4;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const float4 mu = ld4(mean + c), is = ld4(invstd + c), g = ld4(gamma + c), be = ld4(beta + c);
    const float4 sc = make_float4(is.x * g.x, is.y * g.y, is.z * g.z, is.w * g.w);
    auto act = [&](const float4 x) {
      return make_float4(fmaxf((x.x - mu.x) * sc.x + be.x, 0.f), fmaxf((x.y - mu.y) * sc.y + be.y, 0.f),
                         fmaxf((x.z - mu.z) * sc.z + be.z, 0.f), fmaxf((x.w - mu.w) * sc.w + be.w, 0.f));
    };
    const float* p = X + (((long)b * H + oy * 2) * W + ox * 2) * C + c;
    const float4 a = act(ld4(p)), b4 = act(ld4(p + C)), cc = act(ld4(p + (long)W * C)), d = act(ld4(p + (long)W * C + C));
    const float4 o = make_float4(0.25f * (a.x + b4.x + cc.x + d.x), 0.25f * (a.y + b4.y + cc.y + d.y),
                                 0.25f * (a.z + b4.z + cc.z + d.z), 0.25f * (a.w + b4.w + cc.w + d.w));
    st4(Yp + i * 4, o);
    am = max(am, abits4(o));
  }
  if (amax != nullptr) amax_commit_block(am, amax);
}

// dx = gamma * invstd * (dz - sum_dz/cnt - xhat * sum_dzxhat/cnt),  dz = dY * (Y>0 if Y); optional dZ <- dz
// POOL: dY is the pooled gradient (see col_partial_kernel), pool_h / pool_w the full-size map.
template <bool POOL, bool BIG = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                           const float* __restrict__ X, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ sum_dz,
                                                           const float* __restrict__ sum_dzx, float inv_cnt,
                                                           float* __restrict__ dX, float* __restrict__ dZ, long n4,
                                                           int C, const float* __restrict__ beta_mask, int pool_h = 0,
                                                           int pool_w = 0, unsigned* __restrict__ amax = nullptr) {
  unsigned am = 0u;
  const long stride = (long)gridDim.x * blockDim.x;
  long i = BIG ? (long)blockIdx.x * BIG_PIECE + threadIdx.x : (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = (int)((i * 4) % C);
  const float4 mu = ld4(mean + c), is = ld4(invstd + c), ga = ld4(gamma + c), a = ld4(sum_dz + c), b = ld4(sum_dzx + c);
  // beta_mask != NULL: ReLU mask recomputed from X (bn_apply_kernel's expression) instead of read from Y
  const float4 be = beta_mask ? ld4(beta_mask + c) : make_float4(0, 0, 0, 0);
  // dx = k1 * (g - k2 - (x - mu) * k3)   with k1 = gamma*invstd, k2 = sum_dz/cnt, k3 = invstd * sum_dzx/cnt
  const float4 k1 = make_float4(ga.x * is.x, ga.y * is.y, ga.z * is.z, ga.w * is.w);
  const float4 k2 = make_float4(a.x * inv_cnt, a.y * inv_cnt, a.z * inv_cnt, a.w * inv_cnt);
  const float4 k3 = make_float4(is.x * b.x * inv_cnt, is.y * b.y * inv_cnt, is.z * b.z * inv_cnt, is.w * b.w * inv_cnt);
  const bool use_y = !beta_mask && Y != nullptr;
  auto load_g = [&](long j) {
    float4 g;
    if (POOL) {
      g = ld4(dY + pooled_row(j * 4 / C, pool_h, pool_w) * C + c);
      g.x *= 0.25f; g.y *= 0.25f; g.z *= 0.25f; g.w *= 0.25f;
    } else {
      g = BIG ? ld4nt(dY + j * 4) : ld4(dY + j * 4);
    }
    return g;
  };
  // (x, g, y) of vector j -> masked g stored to dZ (if asked), dx stored
  auto finish = [&](long j, float4 g, const float4 x, const float4 y) {
    if (beta_mask) {
      if (!((x.x - mu.x) * k1.x + be.x > 0.f)) g.x = 0.f;   // k1 = invstd * gamma = the forward's scale
      if (!((x.y - mu.y) * k1.y + be.y > 0.f)) g.y = 0.f;
      if (!((x.z - mu.z) * k1.z + be.z > 0.f)) g.z = 0.f;
      if (!((x.w - mu.w) * k1.w + be.w > 0.f)) g.w = 0.f;
    } else if (use_y) {
      if (!(y.x > 0.f)) g.x = 0.f;
      if (!(y.y > 0.f)) g.y = 0.f;
      if (!(y.z > 0.f)) g.z = 0.f;
      if (!(y.w > 0.f)) g.w = 0.f;
    }
    if (dZ) { if (BIG) st4nt(dZ + j * 4, g); else st4(dZ + j * 4, g); }  // masked upstream gradient = gradient of the residual branch
    float4 o;
    o.x = k1.x * (g.x - k2.x - (x.x - mu.x) * k3.x);
    o.y = k1.y * (g.y - k2.y - (x.y - mu.y) * k3.y);
    o.z = k1.z * (g.z - k2.z - (x.z - mu.z) * k3.z);
    o.w = k1.w * (g.w - k2.w - (x.w - mu.w) * k3.w);
    if (BIG) st4nt(dX + j * 4, o); else st4(dX + j * 4, o);
    am = max(am, abits4(o));
  };
  const float4 z4 = make_float4(0, 0, 0, 0);
  if (BIG) {
    float4 g[BIG_U], x[BIG_U], y[BIG_U];
#pragma unroll
    for (int u = 0; u < BIG_U; ++u) {
      const long j = i + u * 256;
      const bool ok = j < n4;
      g[u] = ok ? load_g(j) : z4;
      x[u] = ok ? ld4nt(X + j * 4) : z4;
      y[u] = (ok && use_y) ? ld4nt(Y + j * 4) : z4;
    }
#pragma unroll
    for (int u = 0; u < BIG_U; ++u) {
      const long j = i + u * 256;
      if (j < n4) finish(j, g[u], x[u], y[u]);
    }
  } else {
    auto one = [&](long j) { finish(j, load_g(j), ld4(X + j * 4), use_y ? ld4(Y + j * 4) : z4); };
    for (; i + stride < n4; i += 2 * stride) {
      one(i);
      one(i + stride);
    }
    if (i < n4) one(i);
  }
  if (amax != nullptr) amax_commit_block(am, amax);
}

// ------------------------------------------------------------------------------------------------------
// InstanceNorm over P pixels of [B, P, C] (channels-last), affine, optional ReLU.  Block = (b, 64 channels):
// 256 threads = 64 channels x 4 pixel groups.  P is small (100) so two passes out of L1/L2 are cheap.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void instnorm_fwd_kernel(const float* __restrict__ X, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ Y,
                                                           float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                           int P, int C, float eps, int relu,
                                                           unsigned* __restrict__ amax = nullptr) {
  __shared__ float red[4][64];
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), pg = threadIdx.x >> 6;
  const bool ok = c < C;
  const float* x = X + (long)b * P * C + c;
  float s = 0.f;
  if (ok) for (int p = pg; p < P; p += 4) s += x[(long)p * C];
  red[pg][threadIdx.x & 63] = s;
  __syncthreads();
  const int cl = threadIdx.x & 63;
  float mean = (red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]) / (float)P;
  __syncthreads();
  float v = 0.f;
  if (ok) for (int p = pg; p < P; p += 4) { float d = x[(long)p * C] - mean; v += d * d; }
  red[pg][cl] = v;
  __syncthreads();
  float var = (red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]) / (float)P;
  float is = rsqrtf(var + eps);
  unsigned am = 0u;   // (h2: the amax of Y as a by-product, tris_amax_next; every lane reaches the commit)
  if (ok) {
    float g = gamma[c], be = beta[c];
    float* y = Y + (long)b * P * C + c;
    for (int p = pg; p < P; p += 4) {
      float o = (x[(long)p * C] - mean) * is * g + be;
      if (relu) o = fmaxf(o, 0.f);
      y[(long)p * C] = o;
      am = max(am, __builtin_bit_cast(unsigned, o) & 0x7fffffffu);
    }
    if (pg == 0) { mean_out[(long)b * C + c] = mean; invstd_out[(long)b * C + c] = is; }
  }
  if (amax != nullptr) amax_commit(am, amax);
}

__global__ __launch_bounds__(256) void instnorm_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                           const float* __restrict__ X, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean_in,
                                                           const float* __restrict__ invstd_in, float* __restrict__ dX,
                                                           float* __restrict__ dgamma_part, float* __restrict__ dbeta_part,
                                                           int P, int C, int relu, unsigned* __restrict__ amax = nullptr) {
  __shared__ float r0[4][64];
  __shared__ float r1[4][64];
  const int b = blockIdx.y, cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, pg = threadIdx.x >> 6;
  const bool ok = c < C;
  const long base = (long)b * P * C + c;
  float mean = ok ? mean_in[(long)b * C + c] : 0.f, is = ok ? invstd_in[(long)b * C + c] : 0.f;
  float s0 = 0.f, s1 = 0.f;
  if (ok)
    for (int p = pg; p < P; p += 4) {
      float g = dY[base + (long)p * C];
      if (relu && !(Y[base + (long)p * C] > 0.f)) g = 0.f;
      s0 += g;
      s1 += g * (X[base + (long)p * C] - mean) * is;
    }
  r0[pg][cl] = s0;
  r1[pg][cl] = s1;
  __syncthreads();
  s0 = r0[0][cl] + r0[1][cl] + r0[2][cl] + r0[3][cl];
  s1 = r1[0][cl] + r1[1][cl] + r1[2][cl] + r1[3][cl];
  unsigned am = 0u;   // (h2: the amax of dX as a by-product; every lane reaches the commit)
  if (ok) {
    float ga = gamma[c], ip = 1.0f / (float)P;
    for (int p = pg; p < P; p += 4) {
      float g = dY[base + (long)p * C];
      if (relu && !(Y[base + (long)p * C] > 0.f)) g = 0.f;
      float xh = (X[base + (long)p * C] - mean) * is;
      const float d = ga * is * (g - s0 * ip - xh * s1 * ip);
      dX[base + (long)p * C] = d;
      am = max(am, __builtin_bit_cast(unsigned, d) & 0x7fffffffu);
    }
    if (pg == 0) { dgamma_part[(long)b * C + c] = s1; dbeta_part[(long)b * C + c] = s0; }
  }
  if (amax != nullptr) amax_commit(am, amax);
}

// ------------------------------------------------------------------------------------------------------
// LayerNorm over the last dim W (<= 1024, W % 4 == 0): one wave per row, row held in registers.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ X, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ Y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            long rows, int W, float eps, unsigned* __restrict__ amax = nullptr,
                                                            const int* __restrict__ limit = nullptr) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (limit != nullptr && row >= (long)*limit) return;   // (packed text rows: nothing lives behind the limit; a whole wave leaves)
  unsigned am = 0u;
  const int W4 = W >> 2;
  float4 v[4];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int i = lane + q * 64;
    v[q] = make_float4(0, 0, 0, 0);
    if (i < W4) { v[q] = ld4(X + row * W + i * 4); s += v[q].x + v[q].y + v[q].z + v[q].w; }
  }
  float mean = wave_sum(s) / (float)W;
  float s2 = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int i = lane + q * 64;
    if (i < W4) {
      float a = v[q].x - mean, b = v[q].y - mean, c = v[q].z - mean, d = v[q].w - mean;
      s2 += a * a + b * b + c * c + d * d;
    }
  }
  float rstd = rsqrtf(wave_sum(s2) / (float)W + eps);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int i = lane + q * 64;
    if (i < W4) {
      float4 g = ld4(gamma + i * 4), b = ld4(beta + i * 4), o;
      o.x = (v[q].x - mean) * rstd * g.x + b.x;
      o.y = (v[q].y - mean) * rstd * g.y + b.y;
      o.z = (v[q].z - mean) * rstd * g.z + b.z;
      o.w = (v[q].w - mean) * rstd * g.w + b.w;
      st4(Y + row * W + i * 4, o);
      am = max(am, abits4(o));
    }
  }
  if (amax != nullptr) amax_commit(am, amax);
  if (lane == 0 && mean_out) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// dX per row; per-block partial dgamma/dbeta [nblocks][2][W] (block covers rows_per_block rows, 4 at a time).
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, float* __restrict__ dX,
                                                            float* __restrict__ part, long rows, int W,
                                                            long rows_per_block, const float* __restrict__ extra,
                                                            unsigned* __restrict__ amax = nullptr) {
  unsigned am = 0u;
  __shared__ float4 lg[4][256];
  __shared__ float4 lb[4][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int W4 = W >> 2;
  float4 ag[4], ab[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) ag[q] = ab[q] = make_float4(0, 0, 0, 0);
  const long rbeg = (long)blockIdx.x * rows_per_block, rend = min(rows, rbeg + rows_per_block);
  for (long row = rbeg + wv; row < rend; row += 4) {
    const float mean = mean_in[row], rstd = rstd_in[row];
    float4 xh[4], dh[4];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int i = lane + q * 64;
      xh[q] = dh[q] = make_float4(0, 0, 0, 0);
      if (i < W4) {
        float4 x = ld4(X + row * W + i * 4), dy = ld4(dY + row * W + i * 4), g = ld4(gamma + i * 4);
        xh[q] = make_float4((x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd);
        dh[q] = make_float4(dy.x * g.x, dy.y * g.y, dy.z * g.z, dy.w * g.w);
        c1 += dh[q].x + dh[q].y + dh[q].z + dh[q].w;
        c2 += dh[q].x * xh[q].x + dh[q].y * xh[q].y + dh[q].z * xh[q].z + dh[q].w * xh[q].w;
        ag[q].x += dy.x * xh[q].x; ag[q].y += dy.y * xh[q].y; ag[q].z += dy.z * xh[q].z; ag[q].w += dy.w * xh[q].w;
        ab[q].x += dy.x; ab[q].y += dy.y; ab[q].z += dy.z; ab[q].w += dy.w;
      }
    }
    c1 = wave_sum(c1) / (float)W;
    c2 = wave_sum(c2) / (float)W;
    if (dX) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int i = lane + q * 64;
        if (i < W4) {
          float4 o;
          o.x = rstd * (dh[q].x - c1 - xh[q].x * c2);
          o.y = rstd * (dh[q].y - c1 - xh[q].y * c2);
          o.z = rstd * (dh[q].z - c1 - xh[q].z * c2);
          o.w = rstd * (dh[q].w - c1 - xh[q].w * c2);
          if (extra) {  // gradient of the residual branch that by-passes this LayerNorm (ops.GradBox)
            const float4 e = ld4(extra + row * W + i * 4);
            o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
          }
          st4(dX + row * W + i * 4, o);
          am = max(am, abits4(o));
        }
      }
    }
  }
  if (amax != nullptr) amax_commit_block(am, amax);
  if (!part) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) { lg[wv][lane + q * 64] = ag[q]; lb[wv][lane + q * 64] = ab[q]; }
  __syncthreads();
  if (wv == 0) {
    float* o = part + (long)blockIdx.x * 2 * W;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int i = lane + q * 64;
      if (i < W4) {
        float4 g = lg[0][i], b = lb[0][i];
        for (int w = 1; w < 4; ++w) {
          float4 g2 = lg[w][i], b2 = lb[w][i];
          g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
          b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
        }
        st4(o + i * 4, g);
        st4(o + W + i * 4, b);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// 2x2 average pool on NHWC, forward and backward.
// ------------------------------------------------------------------------------------------------------
__global__ void avgpool2_fwd_kernel(const float* __restrict__ X, float* __restrict__ Y, int B, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
  long n = (long)B * Ho * Wo * C4;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int c4 = (int)(i % C4);
    long t = i / C4;
    int ox = (int)(t % Wo);
    t /= Wo;
    int oy = (int)(t % Ho);
    int b = (int)(t / Ho);
    const float* p = X + (((long)b * H + oy * 2) * W + ox * 2) * C + c4 * 4;
    float4 a = ld4(p), b4 = ld4(p + C), c = ld4(p + (long)W * C), d = ld4(p + (long)W * C + C);
    st4(Y + i * 4, make_float4(0.25f * (a.x + b4.x + c.x + d.x), 0.25f * (a.y + b4.y + c.y + d.y),
                               0.25f * (a.z + b4.z + c.z + d.z), 0.25f * (a.w + b4.w + c.w + d.w)));
  }
}

__global__ void avgpool2_bwd_kernel(const float* __restrict__ dY, float* __restrict__ dX, int B, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
  long n = (long)B * H * W * C4;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int c4 = (int)(i % C4);
    long t = i / C4;
    int x = (int)(t % W);
    t /= W;
    int y = (int)(t % H);
    int b = (int)(t / H);
    float4 g = make_float4(0, 0, 0, 0);
    if ((y >> 1) < Ho && (x >> 1) < Wo) {
      g = ld4(dY + (((long)b * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c4 * 4);
      g.x *= 0.25f; g.y *= 0.25f; g.z *= 0.25f; g.w *= 0.25f;
    }
    st4(dX + i * 4, g);
  }
}

// ------------------------------------------------------------------------------------------------------
// Elementwise (op codes in tris_hip.h).  n is a float count, n % 4 == 0 fast path else scalar tail.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ew1(int op, float a, float b, float s) {
  switch (op) {
    case TRIS_EW_ADD: return a + b;
    case TRIS_EW_AXPY: return s * a + b;
    case TRIS_EW_RELU_BWD: return b > 0.f ? a : 0.f;              // a = dY, b = Y
    case TRIS_EW_QGELU: return a / (1.0f + expf(-1.702f * a));
    case TRIS_EW_QGELU_BWD: {                                     // a = dY, b = pre-activation
      float sg = 1.0f / (1.0f + expf(-1.702f * b));
      return a * (sg + 1.702f * b * sg * (1.f - sg));
    }
    case TRIS_EW_MUL: return a * b;
    case TRIS_EW_SCALE: return s * a;
    case TRIS_EW_RELU: return fmaxf(a, 0.f);
    default: return a;
  }
}
// nb > 0: B has nb elements and is read modulo nb (a [rows, nb] operand against one broadcast row; n % 4 == nb % 4 == 0)
__global__ void ew_kernel(int op, const float* __restrict__ A, const float* __restrict__ Bp, float* __restrict__ O,
                          long n, float s, unsigned* __restrict__ amax = nullptr, long nb = 0) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  const long n4 = n >> 2;
  unsigned am = 0u;
  for (long j = i; j < n4; j += stride) {
    float4 a = ld4(A + j * 4), b = Bp ? ld4(Bp + (nb > 0 ? (j * 4) % nb : j * 4)) : make_float4(0, 0, 0, 0);
    const float4 o = make_float4(ew1(op, a.x, b.x, s), ew1(op, a.y, b.y, s), ew1(op, a.z, b.z, s), ew1(op, a.w, b.w, s));
    st4(O + j * 4, o);
    am = max(am, abits4(o));
  }
  for (long j = n4 * 4 + i; j < n; j += stride) {
    const float o = ew1(op, A[j], Bp ? Bp[j] : 0.f, s);
    O[j] = o;
    am = max(am, __builtin_bit_cast(unsigned, o) & 0x7fffffffu);
  }
  if (amax != nullptr) amax_commit_block(am, amax);
}

// NCHW [B,C,H,W] <-> NHWC [B,H,W,C] (C tiny: the 3-channel input image)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ X, float* __restrict__ Y, int B, int C, long HW) {
  long n = (long)B * HW * C;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int c = (int)(i % C);
    long t = i / C;
    long p = t % HW;
    long b = t / HW;
    Y[i] = X[(b * C + c) * HW + p];
  }
}

// generic column sum for widths that are not a multiple of 4 (rare, small): one thread per column
__global__ void colsum_scalar_kernel(const float* __restrict__ X, long M, int N, long ld, float* __restrict__ out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  float s = 0.f;
  for (long r = 0; r < M; ++r) s += X[r * ld + c];
  out[c] = s;
}

inline int grid_for(long n, int block = 256) {
  long g = (n + block - 1) / block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}
// grid for the channel-invariant BN apply kernels: gridDim*256 must be a multiple of C/4 (C/4 = 2^k or any divisor-friendly
// width); returns 0 when that cannot be arranged (caller reports an error: BN widths on this path are powers of two)
inline int bn_grid(long n4, int C) {
  const long cv = C / 4;
  long g = (n4 + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  // smallest multiple m of cv/gcd(cv,256) that is >= g
  long a = cv, b = 256;
  while (b) { long t = a % b; a = b; b = t; }
  const long unit = cv / a;
  g = (g + unit - 1) / unit * unit;
  return (int)g;
}

struct ColPlan { int nb; long rpb; };
inline ColPlan col_plan(long M, int C) {
  int CV = C / 4, CVB = CV < 256 ? CV : 256, RS = 256 / CVB;
  const int target = tris_internal_col_blocks;
  long rpb = (M + target - 1) / target;
  long minr = (long)RS * 8;
  if (rpb < minr) rpb = minr;
  rpb = (rpb + RS - 1) / RS * RS;
  ColPlan p;
  p.rpb = rpb;
  p.nb = (int)((M + rpb - 1) / rpb);
  return p;
}

}  // namespace

static thread_local unsigned* g_amax_next = nullptr;
static unsigned* take_amax_next() {
  unsigned* p = g_amax_next;
  g_amax_next = nullptr;
  return p;
}
// (the other translation units that can produce an amax -- attention kernels, the GEMM epilogue -- take the arming through this)
extern "C" __attribute__((visibility("hidden"))) unsigned* tris_internal_take_amax_next() { return take_amax_next(); }
// Row limit of the calling thread (tris_rows_limit_thread, include/tris_hip.h "packed text rows"): a device word; launches of
// tris_gemm_f32 (row-major A) and tris_layernorm_fwd_f32 skip rows >= *limit.  Persistent until cleared with NULL.
static thread_local const int* g_rows_limit = nullptr;
extern "C" __attribute__((visibility("hidden"))) const int* tris_internal_rows_limit() { return g_rows_limit; }
extern "C" int tris_rows_limit_thread(const int* limit) {
  g_rows_limit = limit;
  return 0;
}
extern "C" int tris_amax_next(unsigned* out) {
  g_amax_next = out;
  return 0;
}

extern "C" long tris_col_workspace_bytes(long M, int C) {
  ColPlan p = col_plan(M, C);
  return (long)p.nb * 2 * C * sizeof(double);
}

extern "C" int tris_bn_stats_f32(const float* X, long M, int C, float eps, float momentum, float* stats,
                                 float* running_mean, float* running_var, float* workspace, void* stream) {
  if (C % 4) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  ColPlan p = col_plan(M, C);
  hipLaunchKernelGGL(col_partial_kernel<0>, dim3(p.nb), dim3(256), 0, st, X, nullptr, nullptr, nullptr, nullptr, M, C,
                     (long)C, p.rpb, (double*)workspace);
  TRIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(fin_grid(C, p.nb)), dim3(fin_block(p.nb)), 0, st, (const double*)workspace, p.nb, M, C,
                     eps, momentum, stats, running_mean, running_var);
  TRIS_LAUNCH_CHECK();
  return 0;
}

// finish BN statistics from fp64 partials produced by a fused conv epilogue (tris_*_bnstat_f32)
extern "C" int tris_bn_finalize_f32(const double* part, int rows, long M, int C, float eps, float momentum, float* stats,
                                    float* running_mean, float* running_var, void* stream) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(fin_grid(C, rows)), dim3(fin_block(rows)), 0, (hipStream_t)stream, part, rows, M, C, eps,
                     momentum, stats, running_mean, running_var);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_bn_finalize_bound_f32(const double* part, int rows, long M, int C, float eps, float momentum, float* stats,
                                          float* running_mean, float* running_var, const float* gamma, const float* beta, float xhat_max,
                                          const unsigned* add_word, unsigned* bound_out, void* stream) {
  if (gamma == nullptr || beta == nullptr || bound_out == nullptr) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(fin_grid(C, rows)), dim3(fin_block(rows)), 0, (hipStream_t)stream, part, rows, M, C, eps,
                     momentum, stats, running_mean, running_var, gamma, beta, xhat_max, add_word, bound_out);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_bn_sync_combine_f32(const float* gathered, int world, int C, long count_per_rank, float eps,
                                        float momentum, float* stats, float* running_mean, float* running_var,
                                        void* stream) {
  hipLaunchKernelGGL(bn_sync_combine_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gathered, world, C,
                     (float)count_per_rank, eps, momentum, stats, running_mean, running_var);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_bn_apply_f32(const float* X, const float* mean, const float* invstd, const float* gamma,
                                 const float* beta, const float* resid, float* Y, long M, int C, int relu,
                                 void* stream) {
  if (C % 4) return (int)hipErrorInvalidValue;
  long n4 = M * C / 4;
  if (big_form(n4, C, resid ? 3 : 2))
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(big_grid(n4)), dim3(256), 0, (hipStream_t)stream, X, mean, invstd, gamma, beta,
                       resid, Y, n4, C, relu, take_amax_next());
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(bn_grid(n4, C)), dim3(256), 0, (hipStream_t)stream, X, mean, invstd, gamma,
                       beta, resid, Y, n4, C, relu, take_amax_next());
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_bn_bwd_reduce_f32(const float* dY, const float* Y, const float* X, const float* mean,
                                      const float* invstd, long M, int C, float* sum_dz, float* sum_dzx,
                                      float* workspace, const float* gamma_mask, const float* beta_mask, float* dz_out,
                                      void* stream) {
  if (C % 4) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  ColPlan p = col_plan(M, C);
  if ((gamma_mask == nullptr) != (beta_mask == nullptr)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(col_partial_kernel<1>, dim3(p.nb), dim3(256), 0, st, X, dY, gamma_mask ? nullptr : Y, mean, invstd, M, C,
                     (long)C, p.rpb, (double*)workspace, gamma_mask, beta_mask, dz_out, 0, 0, 0, take_amax_next());
  TRIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(part_finalize_kernel<double>, dim3(fin_grid(C, p.nb)), dim3(fin_block(p.nb)), 0, st, (const double*)workspace, p.nb, C,
                     sum_dz, sum_dzx);
  TRIS_LAUNCH_CHECK();
  return 0;
}

// the same with the BatchNorm's OUTPUT given as fp16 planes (csrc/planes.h): its ReLU mask is "a piece is non-zero"
extern "C" int tris_bn_bwd_reduce_pl_f32(const float* dY, const float* Ypl, const float* X, const float* mean, const float* invstd,
                                         long M, int C, float* sum_dz, float* sum_dzx, float* workspace, float* dz_out,
                                         void* stream) {
  if (C % 8 || Ypl == nullptr) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  ColPlan p = col_plan(M, C);
  hipLaunchKernelGGL(col_partial_kernel<1>, dim3(p.nb), dim3(256), 0, st, X, dY, Ypl, mean, invstd, M, C, (long)C, p.rpb,
                     (double*)workspace, (const float*)nullptr, (const float*)nullptr, dz_out, 0, 0, 1, take_amax_next());
  TRIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(part_finalize_kernel<double>, dim3(fin_grid(C, p.nb)), dim3(fin_block(p.nb)), 0, st, (const double*)workspace, p.nb, C,
                     sum_dz, sum_dzx);
  TRIS_LAUNCH_CHECK();
  return 0;
}

// BatchNorm + ReLU + AvgPool2d(2) as one op: forward, and the two backward passes reading the POOLED upstream gradient
extern "C" int tris_bn_apply_pool_f32(const float* X, const float* mean, const float* invstd, const float* gamma,
                                      const float* beta, float* Yp, int B, int H, int W, int C, void* stream) {
  if (C % 4 || (H & 1) || (W & 1)) return (int)hipErrorInvalidValue;
  const long n = (long)B * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(bn_apply_pool_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, X, mean, invstd, gamma, beta, Yp,
                     B, H, W, C, take_amax_next());
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_bn_bwd_reduce_pool_f32(const float* dYp, const float* X, const float* mean, const float* invstd, int B, int H,
                                           int W, int C, float* sum_dz, float* sum_dzx, float* workspace, const float* gamma,
                                           const float* beta, void* stream) {
  if (C % 4 || (H & 1) || (W & 1) || gamma == nullptr || beta == nullptr) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const long M = (long)B * H * W;
  ColPlan p = col_plan(M, C);
  hipLaunchKernelGGL((col_partial_kernel<1, true>), dim3(p.nb), dim3(256), 0, st, X, dYp, (const float*)nullptr, mean, invstd, M, C,
                     (long)C, p.rpb, (double*)workspace, gamma, beta, (float*)nullptr, H, W, 0, take_amax_next());
  TRIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(part_finalize_kernel<double>, dim3(fin_grid(C, p.nb)), dim3(fin_block(p.nb)), 0, st, (const double*)workspace,
                     p.nb, C, sum_dz, sum_dzx);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_bn_bwd_apply_pool_f32(const float* dYp, const float* X, const float* mean, const float* invstd,
                                          const float* gamma, const float* beta, const float* sum_dz, const float* sum_dzx,
                                          float inv_count, float* dX, int B, int H, int W, int C, void* stream) {
  if (C % 4 || (H & 1) || (W & 1)) return (int)hipErrorInvalidValue;
  const long n4 = (long)B * H * W * C / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(bn_grid(n4, C)), dim3(256), 0, (hipStream_t)stream, dYp, (const float*)nullptr, X,
                     mean, invstd, gamma, sum_dz, sum_dzx, inv_count, dX, (float*)nullptr, n4, C, beta, H, W, take_amax_next());
  TRIS_LAUNCH_CHECK();
  return 0;
}

// largest magnitude of a tensor as a bit pattern (positive floats order like unsigned integers): atomicMax into *out, which the
// caller zeroes beforehand -- the operand scale of an "h2" product is derived from it inside the GEMM kernel (x3_split.h)
// upper bound of |bn(x)| over the whole tensor from the affine parameters alone (include/tris_hip.h tris_bn_out_bound_f32): the
// operand scale of an h2 product whose input relu(bn(x)) is formed inside the consuming kernel and never exists in memory
__global__ __launch_bounds__(256) void bn_out_bound_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, int C,
                                                           float xhat_max, unsigned* __restrict__ out) {
  __shared__ float red[4];
  float m = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, fabsf(gamma[c]) * xhat_max + fabsf(beta[c]));
  m = block_max_256(m, red);
  if (threadIdx.x == 0) out[0] = __builtin_bit_cast(unsigned, m);   // (one of the word's 128 lines; the others stay zero)
}
extern "C" int tris_bn_out_bound_f32(const float* gamma, const float* beta, int C, float xhat_max, unsigned* out, void* stream) {
  if (C <= 0 || out == nullptr) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(bn_out_bound_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, gamma, beta, C, xhat_max, out);
  TRIS_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void amax_bits_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  unsigned m = 0u;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = ld4(x + i * 4);
    m = max(max(m, __builtin_bit_cast(unsigned, v.x) & 0x7fffffffu), __builtin_bit_cast(unsigned, v.y) & 0x7fffffffu);
    m = max(max(m, __builtin_bit_cast(unsigned, v.z) & 0x7fffffffu), __builtin_bit_cast(unsigned, v.w) & 0x7fffffffu);
  }
  for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    m = max(m, __builtin_bit_cast(unsigned, x[i]) & 0x7fffffffu);
  amax_commit_block(m, out);
}

// the same for many tensors of one flat buffer in ONE launch (the weights of an optimiser arena): grid (chunks, segments)
__global__ __launch_bounds__(256) void amax_segments_kernel(const float* __restrict__ base, const long* __restrict__ offs,
                                                            const long* __restrict__ sizes, unsigned* __restrict__ slots) {
  const long n = sizes[blockIdx.y];
  const float* x = base + offs[blockIdx.y];
  const long n4 = n >> 2;   // (arena slots are 256-byte aligned)
  const long stride = (long)gridDim.x * blockDim.x;
  unsigned m = 0u;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) m = max(m, abits4(ld4(x + i * 4)));
  for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    m = max(m, __builtin_bit_cast(unsigned, x[i]) & 0x7fffffffu);
  amax_commit_block(m, slots + (long)blockIdx.y * 2048);
}

extern "C" int tris_amax_segments_f32(const float* base, const long* offs, const long* sizes, int nseg, unsigned* slots,
                                      void* stream) {
  if (nseg < 1 || (((uintptr_t)base) & 15)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(amax_segments_kernel, dim3(128, (unsigned)nseg), dim3(256), 0, (hipStream_t)stream, base, offs, sizes, slots);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_amax_bits_f32(const float* x, long n, unsigned* out, void* stream) {
  if (n < 1 || (((uintptr_t)x) & 15)) return (int)hipErrorInvalidValue;
  long g = (n / 4 + 255) / 256;
  g = g > 2048 ? 2048 : (g < 1 ? 1 : g);
  hipLaunchKernelGGL(amax_bits_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, n, out);
  TRIS_LAUNCH_CHECK();
  return 0;
}

// finish fp64 partial rows [rows][2][C] (a fused producer's epilogue, e.g. tris_gemm_bnbwd_f32) -> out0[C], out1[C]
extern "C" int tris_part_finalize_f32(const double* part, int rows, int C, float* out0, float* out1, void* stream) {
  if (rows < 1 || C < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(part_finalize_kernel<double>, dim3(fin_grid(C, rows)), dim3(fin_block(rows)), 0, (hipStream_t)stream, part, rows,
                     C, out0, out1);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_part_finalize_bound_f32(const double* part, int rows, int C, float* out0, float* out1, const float* gamma,
                                            const float* invstd, float inv_count, float xhat_max, const unsigned* dz_word,
                                            unsigned* bound_out, void* stream) {
  if (rows < 1 || C < 1 || out1 == nullptr || gamma == nullptr || invstd == nullptr || dz_word == nullptr || bound_out == nullptr)
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(part_finalize_kernel<double>, dim3(fin_grid(C, rows)), dim3(fin_block(rows)), 0, (hipStream_t)stream, part, rows,
                     C, out0, out1, gamma, invstd, inv_count, xhat_max, dz_word, bound_out);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_bn_bwd_apply_f32(const float* dY, const float* Y, const float* X, const float* mean,
                                     const float* invstd, const float* gamma, const float* sum_dz, const float* sum_dzx,
                                     float inv_count, float* dX, float* dZ, long M, int C, const float* beta_mask,
                                     void* stream) {
  long n4 = M * C / 4;
  const int streams = 3 + ((Y && !beta_mask) ? 1 : 0) + (dZ ? 1 : 0);
  if (big_form(n4, C, streams))
    hipLaunchKernelGGL((bn_bwd_apply_kernel<false, true>), dim3(big_grid(n4)), dim3(256), 0, (hipStream_t)stream, dY, Y, X, mean,
                       invstd, gamma, sum_dz, sum_dzx, inv_count, dX, dZ, n4, C, beta_mask, 0, 0, take_amax_next());
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<false, false>), dim3(bn_grid(n4, C)), dim3(256), 0, (hipStream_t)stream, dY, Y, X, mean,
                       invstd, gamma, sum_dz, sum_dzx, inv_count, dX, dZ, n4, C, beta_mask, 0, 0, take_amax_next());
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_colsum_f32(const float* X, long M, int N, long ld, float* out, float* workspace, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (N % 4 || ld % 4 || (((uintptr_t)X) & 15)) {
    hipLaunchKernelGGL(colsum_scalar_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, X, M, N, ld, out);
    TRIS_LAUNCH_CHECK();
    return 0;
  }
  ColPlan p = col_plan(M, N);
  hipLaunchKernelGGL(col_partial_kernel<2>, dim3(p.nb), dim3(256), 0, st, X, nullptr, nullptr, nullptr, nullptr, M, N,
                     ld, p.rpb, (double*)workspace);
  TRIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(part_finalize_kernel<double>, dim3(fin_grid(N, p.nb)), dim3(fin_block(p.nb)), 0, st, (const double*)workspace, p.nb, N,
                     out, (float*)nullptr);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_instnorm_fwd_f32(const float* X, const float* gamma, const float* beta, float* Y, float* mean,
                                     float* invstd, int B, int P, int C, float eps, int relu, void* stream) {
  hipLaunchKernelGGL(instnorm_fwd_kernel, dim3(cdiv(C, 64), B), dim3(256), 0, (hipStream_t)stream, X, gamma, beta, Y,
                     mean, invstd, P, C, eps, relu, take_amax_next());
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_instnorm_bwd_f32(const float* dY, const float* Y, const float* X, const float* gamma,
                                     const float* mean, const float* invstd, float* dX, float* dgamma_part,
                                     float* dbeta_part, int B, int P, int C, int relu, void* stream) {
  hipLaunchKernelGGL(instnorm_bwd_kernel, dim3(cdiv(C, 64), B), dim3(256), 0, (hipStream_t)stream, dY, Y, X, gamma,
                     mean, invstd, dX, dgamma_part, dbeta_part, P, C, relu, take_amax_next());
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_layernorm_fwd_f32(const float* X, const float* gamma, const float* beta, float* Y, float* mean,
                                      float* rstd, long rows, int W, float eps, void* stream) {
  if (W % 4 || W > 1024) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, X, gamma, beta, Y,
                     mean, rstd, rows, W, eps, take_amax_next(), g_rows_limit);
  TRIS_LAUNCH_CHECK();
  return 0;
}

// rows per block of the LayerNorm backward: a wave walks its rows one after the other (two dependent wave reductions per row), so the
// launch is a chain of row latencies -- at most 512 blocks (partial rows for the finalizer), at least 4 rows (one per wave).
// LN_BWD_BLOCKS option: the block count aimed for (128 = rounds 1-3).
extern "C" { __attribute__((visibility("hidden"))) int tris_internal_ln_bwd_blocks = 512; }
static long ln_bwd_rpb(long rows) {
  const long target = tris_internal_ln_bwd_blocks;
  long rpb = (rows + target - 1) / target;
  const long floor_ = target > 128 ? 4 : 8;
  return rpb < floor_ ? floor_ : rpb;
}
extern "C" long tris_layernorm_bwd_workspace_bytes(long rows, int W) {
  const long rpb = ln_bwd_rpb(rows);
  long nb = (rows + rpb - 1) / rpb;
  if (nb < 512) nb = 512;   // (sized for either setting of the option)
  return nb * 2 * W * (long)sizeof(float);
}

extern "C" int tris_layernorm_bwd_f32(const float* dY, const float* X, const float* gamma, const float* mean,
                                      const float* rstd, float* dX, float* dgamma, float* dbeta, long rows, int W,
                                      float* workspace, const float* extra, void* stream) {
  if (W % 4 || W > 1024) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const long rpb = ln_bwd_rpb(rows);
  int nb = (int)((rows + rpb - 1) / rpb);
  float* part = dgamma ? workspace : nullptr;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nb), dim3(256), 0, st, dY, X, gamma, mean, rstd, dX, part, rows, W,
                     rpb, extra, take_amax_next());
  TRIS_LAUNCH_CHECK();
  if (dgamma) {
    hipLaunchKernelGGL(part_finalize_kernel<float>, dim3(fin_grid(W, nb)), dim3(fin_block(nb)), 0, st, (const float*)workspace, nb, W, dgamma, dbeta);
    TRIS_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int tris_avgpool2_fwd_f32(const float* X, float* Y, int B, int H, int W, int C, void* stream) {
  if (C % 4) return (int)hipErrorInvalidValue;
  long n = (long)B * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, X, Y, B, H, W, C);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_avgpool2_bwd_f32(const float* dY, float* dX, int B, int H, int W, int C, void* stream) {
  if (C % 4) return (int)hipErrorInvalidValue;
  long n = (long)B * H * W * (C / 4);
  hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dY, dX, B, H, W, C);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_elementwise_f32(int op, const float* A, const float* B, float* O, long n, float s, void* stream) {
  hipLaunchKernelGGL(ew_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, op, A, B, O, n, s,
                     take_amax_next());
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_elementwise_bcast_f32(int op, const float* A, const float* B, float* O, long n, long nb, float s,
                                          void* stream) {
  if (B == nullptr || nb < 4 || (nb & 3) || (n & 3) || n % nb) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(ew_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, op, A, B, O, n, s, take_amax_next(), nb);
  TRIS_LAUNCH_CHECK();
  return 0;
}

// O = X * exp(ls[0]); e_out[0] = exp(ls[0])   (model_stage1.py:77-78: score * logit_scale.exp())
namespace {
__global__ __launch_bounds__(256) void scale_exp_fwd_kernel(const float* __restrict__ X, const float* __restrict__ ls,
                                                            float* __restrict__ O, float* __restrict__ e_out, long n) {
  const float e = expf(ls[0]);
  if (blockIdx.x == 0 && threadIdx.x == 0) e_out[0] = e;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) O[i] = X[i] * e;
}
// dX = dO * e; part[block] = sum(dO * O) over the block's elements (fp64 from the block reduction on)
__global__ __launch_bounds__(256) void scale_exp_bwd_kernel(const float* __restrict__ dO, const float* __restrict__ O,
                                                            const float* __restrict__ ls, float* __restrict__ dX,
                                                            double* __restrict__ part, long n) {
  __shared__ double red[256];
  const float e = expf(ls[0]);
  const long stride = (long)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float g = dO[i];
    if (dX) dX[i] = g * e;
    acc += g * O[i];
  }
  red[threadIdx.x] = (double)acc;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void scale_exp_bwd_finish_kernel(const double* __restrict__ part, int nb, float* __restrict__ dls) {
  double s = 0.0;
  for (int i = 0; i < nb; ++i) s += part[i];   // fixed order
  dls[0] = (float)s;
}
constexpr int SE_BLOCKS = 128;
}  // namespace
extern "C" int tris_scale_exp_fwd_f32(const float* X, const float* ls, float* O, float* e_out, long n, void* stream) {
  hipLaunchKernelGGL(scale_exp_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, X, ls, O, e_out, n);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" long tris_scale_exp_workspace_bytes() { return (long)SE_BLOCKS * sizeof(double); }
// dls[0] = d/d ls = sum(dO * O) (written, not accumulated); dX may be NULL
extern "C" int tris_scale_exp_bwd_f32(const float* dO, const float* O, const float* ls, float* dX, float* dls,
                                      float* workspace, long n, void* stream) {
  double* part = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(scale_exp_bwd_kernel, dim3(SE_BLOCKS), dim3(256), 0, (hipStream_t)stream, dO, O, ls, dX, part, n);
  TRIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(scale_exp_bwd_finish_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, part, SE_BLOCKS, dls);
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_nchw_to_nhwc_f32(const float* X, float* Y, int B, int C, int H, int W, void* stream) {
  long n = (long)B * C * H * W;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, X, Y, B, C,
                     (long)H * W);
  TRIS_LAUNCH_CHECK();
  return 0;
}
