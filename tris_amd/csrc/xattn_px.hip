// Bilateral image<->text cross attention (reference model/attn.py:117-128) as ONE persistent launch, cut by PIXELS
// (gfx950; arithmetic x3 = three bf16 pieces / six MFMAs per product, or -- template H2, armed per call by tris_xattn_amax_next --
// h2 = two fp16 pieces of x * s / three MFMAs, two accumulator sets, the scales from the six operands' amax words; the
// probabilities, which are <= 1, take the fixed scale 2^13).
//
//   Av  = softmax_n(Qv Kt^T / sqrt(C))   [B,P,N]      new_vis = Av  Vt      [B,P,C]
//   AtT = softmax_p(Kv Qt^T / sqrt(C))   [B,P,N]      new_lan = AtT^T Vv    [B,N,C]
//
// Why this cut.  The pair is an HBM stream (1.84 MB per image at P = 100, N = 48, C = 1024) and one CU fetches ~10 B/clk of it,
// so an image has to be spread over several CUs -- the question is which axis is cut.  Channel slices (xattn_fused.hip) make
// BOTH soft-maxes wait for a reduction over workgroups: three dependent hand-offs of 43 + 5 KB per workgroup, and eight slices
// x 48 images do not fit 256 CUs once.  Pixel rows do: S workgroups per image (S = 5 at B = 48: 240 workgroups, one per CU), and
// workgroup s owns pixels [s P / S, (s + 1) P / S).  Then
//   * the pixel -> sentence direction never leaves the workgroup: Qv rows . Kt^T over ALL channels, the soft-max over the
//     sentences of a row, new_vis rows = Av . Vt;
//   * the sentence -> pixel direction needs ONE hand-off of N x (own pixels) floats (3.8 KB per workgroup, 19 KB per image):
//     the logits Kv rows . Qt^T are complete per pixel; the soft-max over the pixels of the image needs everyone's columns.  Each
//     workgroup publishes its columns as soon as they exist, runs the whole pixel -> sentence direction, and only then looks at
//     the flags: the hand-off's latency chain (~8 us on this part) hides behind the other direction's products;
//   * new_lan = At . Vv reduces over pixels, so it is cut by CHANNELS instead: workgroup s owns 32-channel units
//     [s U / S, (s + 1) U / S), U = C / 32, reads Vv[b, :, own channels] (requested before the hand-off) and every workgroup of
//     the image repeats the tiny soft-max over pixels (N x P values).
// Every operand byte is read by exactly one workgroup; the sentence operands (3 x N x C, shared by all images) come pre-split into
// bf16 planes in MFMA fragment order from L2 (xattn_planes.h).  HBM traffic = algorithmic + saved probabilities + 2 x 19 KB per
// image of exchange.
//
// Workgroup = 512 threads = two halves of four waves, one per direction, that share no data and therefore synchronise on LDS
// counters of their own (group_sync) instead of s_barrier; they meet once, before new_lan.
//   T (waves 0-3)  D_t = Kv rows . Qt^T, each wave over a quarter of the channels -> summed through LDS -> published (write-through,
//                  drain, flag) -> wait for the S flags -> gather everyone's columns (one round trip) -> soft-max over pixels ->
//                  At as bf16 piece planes in LDS.  A chain of latencies (~12 us) that runs under V's products.
//   V (waves 4-7)  D_v = Qv rows . Kt^T likewise -> soft-max over sentences -> Av as piece planes in LDS -> new_vis, wave w owns the
//                  channel tiles (w - 4) + 4 i (channels are the MFMA rows: a lane stores four consecutive channels of a pixel),
//                  Vt^T fragments four tiles ahead.
//   all            new_lan: wave w owns 32-channel unit w of the workgroup (<= 8): its Vv columns (requested early, split once) are
//                  staged k-major per 32-pixel step in a wave-private LDS buffer and gathered with ds_read_b64_tr_b16.
// Pixel rows are read row-contiguous (8 rows x 128 B per instruction) and turned into MFMA fragments through wave-private LDS
// tiles.  Workgroup barriers are s_waitcnt lgkmcnt(0) + s_barrier: __syncthreads() is a workgroup-scope release, i.e. vmcnt(0),
// and would wait for every prefetch in flight.  Measured per phase: profiles/r4_xattn_phase_table.txt (tools/xattn_px_trace.py).
//
// Inter-workgroup protocol as in xattn_fused.hip (MI355X_MICROARCH "workgroup dispatch / visibility", recipe R1): write-through
// (sc1) 16-byte stores, every storing wave drains vmcnt, barrier, one lane raises the (image, slot) flag relaxed at agent scope;
// every consuming wave polls relaxed itself, then reads with sc1 loads; no fences.  Epoch in device memory (sync[0], advanced by the last
// workgroup to finish): a captured launch replays.  Spins are bounded (sync[2] != 0 afterwards: outputs undefined).  All B * S
// workgroups must be co-resident (one per CU: 120-145 KB of LDS); the entry point declines otherwise.
#include "common.h"
#include "tris_hip.h"
#include "x3_split.h"
#include "xattn_planes.h"

// XATTN_PX_SLOTS option (tris_set_option): workgroups per image, 0 = as many as fit one per CU
extern "C" { __attribute__((visibility("hidden"))) int tris_internal_xattn_px_slots = 0; }

namespace {

#include "amax.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int XP_MAXS = 8;         // workgroups per image (and 32-channel units per workgroup) at most
constexpr int XP_SYNC_FLAGS = 16;  // sync[0] epoch, [1] finish ticket, [2] time-out flag, [16 + b * 8 + s] publish flags
constexpr long XP_SPIN = 4000000;  // polls before a wait gives up (~seconds)
constexpr int XP_TLD = 36;         // row stride (floats) of a wave-private 16 x 32 turn-around tile (conflict-free b128 fragment reads)
constexpr int XP_AVS = 68;         // row stride (floats) of the Av rows [32][64]
constexpr int XP_ATS = 132;        // row stride (floats) of the gathered At rows [NT * 16][128]
constexpr int XP_KS = 80;          // bytes per pixel row of a k-major Vv plane (32 channels x 2 B + 16)
constexpr int XP_PLANE = 32 * XP_KS;   // one plane of one 32-pixel step

#ifdef TRIS_XP_TRACE
#define XP_STAMP(i) do { if (tid == 0) xp_trace[(long)blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define XP_STAMP_V(i) do { if (tid == 256) xp_trace[(long)blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define XP_STAMP_W(i) do { if (lane == 0) xp_trace[(long)blockIdx.x * 32 + (i) + wave] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define XP_STAMP_W(i) do { } while (0)
#define XP_STAMP(i) do { } while (0)
#define XP_STAMP_V(i) do { } while (0)
#endif

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// workgroup barrier for LDS hand-offs that leaves global loads in flight: __syncthreads() is a workgroup-scope release, i.e.
// s_waitcnt vmcnt(0) as well -- every prefetch issued before it would be waited for (measured: 2.5 us per barrier behind the Vv request)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// (nontemporal loads of the pixel rows / stores of the outputs were measured: 30.7 vs 27.7 us per workgroup -- the default policy stays)
__device__ __forceinline__ float4 ld4_nt(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4_nt(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4_wt(__amdgpu_buffer_rsrc_t rs, long float_off, float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)(float_off * 4), 0, 16);
}
__device__ __forceinline__ float4 ld4_wt(__amdgpu_buffer_rsrc_t rs, long float_off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(float_off * 4), 0, 16));
}
__device__ __forceinline__ bf16x8 ldf(const uint4* p) { return __builtin_bit_cast(bf16x8, *p); }

// N independent accumulation chains issued piece by piece: consecutive MFMAs never depend on each other; smallest terms first
template <int NCH>
__device__ __forceinline__ void mfma6_n(const Split8 (&a)[NCH], const Split8 (&b)[NCH], f32x4v (&c)[NCH]) {
#define XP_PIECE(PA, PB)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < NCH; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].PA, b[i].PB, c[i], 0, 0, 0);
  XP_PIECE(lo, hi) XP_PIECE(hi, lo) XP_PIECE(mid, mid) XP_PIECE(mid, hi) XP_PIECE(hi, mid) XP_PIECE(hi, hi)
#undef XP_PIECE
}

// h2: the same chains on fp16 pieces (carried in Split8: hi | mid = lo'): lo' x hi and hi x lo' into cx, hi x hi into c
template <int NCH>
__device__ __forceinline__ void mfma3_n(const Split8 (&a)[NCH], const Split8 (&b)[NCH], f32x4v (&c)[NCH], f32x4v (&cx)[NCH]) {
#define XP_H(x) __builtin_bit_cast(f16x8, x)
#pragma unroll
  for (int i = 0; i < NCH; ++i) cx[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(XP_H(a[i].mid), XP_H(b[i].hi), cx[i], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < NCH; ++i) cx[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(XP_H(a[i].hi), XP_H(b[i].mid), cx[i], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < NCH; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(XP_H(a[i].hi), XP_H(b[i].hi), c[i], 0, 0, 0);
#undef XP_H
}
template <bool H2, int NCH>
__device__ __forceinline__ void xp_mfma_n(const Split8 (&a)[NCH], const Split8 (&b)[NCH], f32x4v (&c)[NCH], f32x4v (&cx)[NCH]) {
  if constexpr (H2) mfma3_n<NCH>(a, b, c, cx);
  else mfma6_n<NCH>(a, b, c);
}
template <bool H2> __device__ __forceinline__ Split8 xp_split8(const float4 u, const float4 w, const float s) {
  if constexpr (H2) return split8h(u, w, s);
  else return split8(u, w);
}
template <bool H2> __device__ __forceinline__ f32x4v xp_join(const f32x4v c, const f32x4v cx) {
  if constexpr (H2) return c + cx * (1.0f / 2048.0f);
  else return c;
}
constexpr float XP_PS = 8192.f;   // h2 scale of the probabilities (<= 1 -> hi <= 2^13)

constexpr int xp_max(int a, int b) { return a > b ? a : b; }
template <int NT, int NPT, int NP = 3> struct XpLds {
  static constexpr int turn = 8 * NPT * 16 * XP_TLD * 4;              // wave-private turn-around tiles (logits phase)
  static constexpr int red = 8 * NPT * NT * 1024;                    // partial logit blocks of the eight waves
  static constexpr int av = 32 * XP_AVS * 4;                         // Av rows, behind turn | red
  static constexpr int planes = 8 * NP * XP_PLANE;                   // wave-private Vv planes (new_lan phase), aliases turn | red
  static constexpr int atf = NT * 4 * NP * 1024;                     // At as piece planes in MFMA fragment order, behind the Vv planes
  static constexpr int arena = xp_max(turn + red + av, planes + atf);
  static constexpr int at = NT * 16 * XP_ATS * 4;                    // gathered logits / At rows (fp32)
  static constexpr int total = arena + at + 16;
};

// grid B * S, block 512.  KQ = 32-channel steps per wave of the logits phase (C = 128 KQ).
template <int NT, int NPT, int KQ, bool H2>
__global__ __launch_bounds__(512, 2) void xattn_px_kernel(const float* __restrict__ Qv, const float* __restrict__ Kv,
                                                          const float* __restrict__ Vv, const uint4* __restrict__ QtF,
                                                          const uint4* __restrict__ KtF, const uint4* __restrict__ VtF,
                                                          float* __restrict__ new_vis, float* __restrict__ new_lan,
                                                          float* __restrict__ probs, float* __restrict__ Sx, int sx_bytes,
                                                          unsigned* __restrict__ sync, int B, int P, int N, int S, float scale,
                                                          const float* __restrict__ scl) {
  constexpr int C = 128 * KQ;
  constexpr int NP = H2 ? 2 : 3;         // piece planes per operand
  // h2: scl[0..5] = the power-of-two scales of Qv, Kv, Vv, Qt, Kt, Vt (written by the plane preparation launch from the amax words)
  const float s_qv = H2 ? scl[0] : 1.f, s_kv = H2 ? scl[1] : 1.f, s_vv = H2 ? scl[2] : 1.f;
  const float s_qt = H2 ? scl[3] : 1.f, s_kt = H2 ? scl[4] : 1.f, s_vt = H2 ? scl[5] : 1.f;
  constexpr int KST = C / 32;
  constexpr int KS2 = (NT + 1) / 2;      // 32-sentence steps of the new_vis product
  constexpr int CT = C / 16;             // channel tiles
  constexpr int U = C / 32;              // 32-channel units
  using L = XpLds<NT, NPT, NP>;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* AvL = reinterpret_cast<float*>(lds + L::turn + L::red);
  float* AtL = reinterpret_cast<float*>(lds + L::arena);
  uint4* AtF = reinterpret_cast<uint4*>(lds + L::planes);
  unsigned* s_epoch_p = reinterpret_cast<unsigned*>(lds + L::arena + L::at);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  const int b = blockIdx.x / S, slot = blockIdx.x - b * S;
  const int p0 = (slot * P) / S, p1 = ((slot + 1) * P) / S, PW = p1 - p0;   // own pixels
  const int u0 = (slot * U) / S, NU = ((slot + 1) * U) / S - u0;            // own 32-channel units (new_lan)
  if (tid == 0) {
    *s_epoch_p = __hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    s_epoch_p[1] = 0u; s_epoch_p[2] = 0u;   // arrival counters of the two halves
  }
  const __amdgpu_buffer_rsrc_t sxr = __builtin_amdgcn_make_buffer_rsrc(Sx, 0, sx_bytes, 0x00020000);
#ifdef TRIS_XP_TRACE
  unsigned long long* xp_trace = reinterpret_cast<unsigned long long*>(Sx + (long)B * XP_MAXS * NT * 16 * 32);
#endif
  XP_STAMP(0);

  // ---- logits of the own pixels over all channels ---------------------------------------------------------------------------------
  // waves 0-3: D_t[p][n] = sum_c Kv[p][c] Qt[n][c];  waves 4-7: D_v[p][n] = sum_c Qv[p][c] Kt[n][c];  wave (g, q): channels
  // [q C / 4, (q + 1) C / 4).  The pixels are the MFMA rows: a lane ends up with four consecutive pixels of one sentence.
  {
    const int g = wave >> 2, q = wave & 3;
    const float* X = g ? Qv : Kv;
    const uint4* F = (g ? KtF : QtF) + ((long)(q * KQ) * NP) * 64 + lane;
    const float s_x = g ? s_qv : s_kv;                      // h2 scale of the pixel rows of this half
    const float inv_l = H2 ? 1.0f / (s_x * (g ? s_kt : s_qt)) : 1.0f;
    float* tile = reinterpret_cast<float*>(lds) + wave * (NPT * 16 * XP_TLD);
    const int lr = lane >> 3, lc = (lane & 7) * 4;   // loader coordinates: row within an 8-row half, float offset in the 128-B piece
    const float* gp[NPT][2];
#pragma unroll
    for (int t = 0; t < NPT; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h) gp[t][h] = X + ((long)b * P + min(p0 + t * 16 + lr + 8 * h, p1 - 1)) * C + q * (C / 4) + lc;
    f32x4v acc[NPT * NT], acx[NPT * NT];
#pragma unroll
    for (int i = 0; i < NPT * NT; ++i) acc[i] = acx[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    constexpr int AHEAD = KQ < 3 ? KQ : 3;           // k-steps of pixel rows in flight (4: no faster, spills)
    float4 v[KQ][NPT][2];
    Split8 fr[KQ][NT];
    auto load_px = [&](int i) {
#pragma unroll
      for (int t = 0; t < NPT; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) v[i][t][h] = ld4_nt(gp[t][h] + i * 32);
    };
    auto load_fr = [&](int i) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const uint4* f = F + ((long)(j * KST + i) * NP) * 64;
        fr[i][j].hi = ldf(f); fr[i][j].mid = ldf(f + 64);
        if constexpr (!H2) fr[i][j].lo = ldf(f + 128);
      }
    };
    constexpr int FAHEAD = (KQ < 2 || NT * NPT >= 8) ? 1 : 2;          // k-steps of sentence fragments in flight (throughput = bytes in flight / latency)
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) load_px(i);
#pragma unroll
    for (int i = 0; i < FAHEAD; ++i) load_fr(i);
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      if (i + FAHEAD < KQ) load_fr(i + FAHEAD);   // (L2 fragments before the HBM rows: the memory counter is in order)
      if (i + AHEAD < KQ) load_px(i + AHEAD);
#pragma unroll
      for (int t = 0; t < NPT; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) *reinterpret_cast<float4*>(tile + t * (16 * XP_TLD) + (lr + 8 * h) * XP_TLD + lc) = v[i][t][h];
      wave_lds_fence();
      Split8 sp[NPT];
#pragma unroll
      for (int t = 0; t < NPT; ++t) {
        const float* f = tile + t * (16 * XP_TLD) + r16 * XP_TLD + kg * 8;
        sp[t] = xp_split8<H2>(*reinterpret_cast<const float4*>(f), *reinterpret_cast<const float4*>(f + 4), s_x);
      }
      wave_lds_fence();
      Split8 ca[NPT * NT], cb[NPT * NT];
#pragma unroll
      for (int t = 0; t < NPT; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j) { ca[t * NT + j] = sp[t]; cb[t * NT + j] = fr[i][j]; }
      xp_mfma_n<H2, NPT * NT>(ca, cb, acc, acx);
    }
    XP_STAMP(1);
    XP_STAMP_W(16);
    // partial blocks -> LDS  (red[wave][t * NT + j][lane] x 16 B; behind the turn-around tiles of all waves)
    float* red = reinterpret_cast<float*>(lds + L::turn) + wave * (NPT * NT * 256);
#pragma unroll
    for (int i = 0; i < NPT * NT; ++i) {
      const f32x4v a = xp_join<H2>(acc[i], acx[i]) * inv_l;
      *reinterpret_cast<float4*>(red + i * 256 + lane * 4) = make_float4(a[0], a[1], a[2], a[3]);
    }
  }
  lds_barrier();
  const unsigned epoch = *s_epoch_p;
  constexpr int TILES = NPT * NT;
  constexpr int VR = 13;   // 8 pixel rows per instruction: P <= 104
  float4 vreg[VR];
  auto load_vv = [&]() {   // the Vv columns of the own unit (wave w = unit w): they do not depend on anything
    const int prow = lane >> 3, c4 = (lane & 7) * 4;
    const int unit = u0 + min(wave, NU - 1);
#pragma unroll
    for (int i = 0; i < VR; ++i) vreg[i] = ld4_nt(Vv + ((long)b * P + min(i * 8 + prow, P - 1)) * C + unit * 32 + c4);
  };
  // ---- the two directions run side by side, four waves each, and meet once before new_lan.  They share no data, so each half
  // synchronises on its own LDS counter (group_sync) instead of the workgroup barrier: the latency chain of the hand-off
  // (write-through drain ~2.5 us, flags ~2 us, gather ~3 us, soft-max over pixels) runs under the L2-bound new_vis product.
  const int grp = wave >> 2, tg = tid & 255;
  const int nks = (P + 31) >> 5;   // 32-pixel steps of the new_lan product, <= 4
  unsigned* gctr = s_epoch_p + 1 + grp;      // arrivals of the own half (zeroed before the barrier above)
  unsigned gtarget = 0;
  auto group_sync = [&]() {
    gtarget += 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(gctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(gctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < gtarget) __builtin_amdgcn_s_sleep(0);
    asm volatile("" ::: "memory");
  };
  if (grp == 1) {
    // ---- pixel -> sentence direction, all inside the workgroup (waves 4-7) ------------------------------------------------------------
    {
      const float* red = reinterpret_cast<const float*>(lds + L::turn) + 4 * TILES * 256;   // the D_v quarters
      for (int e = tg; e < TILES * 64; e += 256) {
        const int tj = e >> 6, l = e & 63, t = tj / NT, j = tj - t * NT;
        float4 a = *reinterpret_cast<const float4*>(red + tj * 256 + l * 4);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
          const float4 w = *reinterpret_cast<const float4*>(red + (q * TILES + tj) * 256 + l * 4);
          a.x += w.x; a.y += w.y; a.z += w.z; a.w += w.w;
        }
        const int pl = t * 16 + 4 * (l >> 4), n = j * 16 + (l & 15);
        AvL[(pl + 0) * XP_AVS + n] = a.x * scale; AvL[(pl + 1) * XP_AVS + n] = a.y * scale;
        AvL[(pl + 2) * XP_AVS + n] = a.z * scale; AvL[(pl + 3) * XP_AVS + n] = a.w * scale;
      }
    }
    group_sync();
    XP_STAMP_V(12);
    // soft-max over the sentences: 4 threads per row, <= 16 sentences each, one pass (rows >= PW and columns >= N become zeros:
    // they are k / column padding of the MFMA operands below)
    {
      const int px = tg >> 2, q = tg & 3;
      if (px < NPT * 16) {
        float* row = AvL + px * XP_AVS;
        float x[16];
        float m = -INFINITY;
#pragma unroll
        for (int u = 0; u < 16; ++u) { x[u] = (q + 4 * u < N) ? row[q + 4 * u] : -INFINITY; m = fmaxf(m, x[u]); }
        m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64));
        float sm = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) { x[u] = (q + 4 * u < N) ? __expf(x[u] - m) : 0.f; sm += x[u]; }
        sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64);
        const float inv = px < PW ? 1.f / sm : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const float y = x[u] * inv;
          row[q + 4 * u] = y;
          if (px < PW && q + 4 * u < N) probs[(((long)b * 4 + 0) * P + p0 + px) * N + q + 4 * u] = y;
        }
      }
    }
    group_sync();
    XP_STAMP_V(3);
    load_vv();
    // new_vis[b, own pixels, :] = Av . Vt: wave w owns the channel tiles (w - 4) + 4 i; channels are the MFMA rows.  Av goes to
    // LDS as bf16 piece planes in fragment order once (AvF, in the space of the turn-around tiles: the other half left its own
    // when it arrived at its first group barrier) and is re-read per channel tile: the registers hold Vt^T fragments in flight
    // instead -- this product runs at (bytes in flight) / (L2 latency)
    while (__hip_atomic_load(s_epoch_p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u) __builtin_amdgcn_s_sleep(0);
    asm volatile("" ::: "memory");
    uint4* AvF = reinterpret_cast<uint4*>(lds);
    for (int e = tg; e < NPT * KS2 * 64; e += 256) {
      const int fk = e >> 6, l = e & 63, t = fk / KS2, ks = fk - t * KS2;
      const float* a = AvL + (t * 16 + (l & 15)) * XP_AVS + ks * 32 + (l >> 4) * 8;
      const Split8 sp = xp_split8<H2>(*reinterpret_cast<const float4*>(a), *reinterpret_cast<const float4*>(a + 4), XP_PS);
      uint4* d = AvF + (fk * NP) * 64 + l;
      d[0] = __builtin_bit_cast(uint4, sp.hi); d[64] = __builtin_bit_cast(uint4, sp.mid);
      if constexpr (!H2) d[128] = __builtin_bit_cast(uint4, sp.lo);
    }
    group_sync();   // (third arrival: this half is done with the partial blocks and the Av rows -- the other half may reuse their LDS)
    // Vt^T fragments through a buffer resource: lanes whose eight sentences are all >= N ask for an out-of-range offset and get
    // zeros without a byte moved (N = 48: a quarter of the second 32-sentence step) -- and without a branch around the load
    const __amdgpu_buffer_rsrc_t vtr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(VtF), 0, CT * KS2 * NP * 1024, 0x00020000);
    const float inv_v = H2 ? 1.0f / (XP_PS * s_vt) : 1.0f;
    constexpr int NIT = CT / 4;
#ifndef XP_VA
#define XP_VA (NPT == 2 ? 4 : 5)
#endif
    constexpr int VA = XP_VA;              // channel tiles of Vt^T fragments in flight per wave
    Split8 vt[VA + 1][KS2];
    const int wv = wave - 4;
    auto load_vt = [&](int ct, Split8 (&d)[KS2]) {
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) {
        const int off = (ks * 32 + kg * 8 < N) ? (((ct * KS2 + ks) * NP) * 64 + lane) * 16 : 0x7fffffff;
        d[ks].hi = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(vtr, off, 0, 0));
        d[ks].mid = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(vtr, off, 1024, 0));
        if constexpr (!H2) d[ks].lo = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(vtr, off, 2048, 0));
      }
    };
#pragma unroll
    for (int it = 0; it < VA; ++it) load_vt(wv + 4 * it, vt[it]);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int ct = wv + 4 * it;
      if (it + VA < NIT) load_vt(ct + 4 * VA, vt[(it + VA) % (VA + 1)]);
      constexpr int NCH = NPT * KS2;
      Split8 ca[NCH], cb[NCH];
      f32x4v co[NCH], cox[NCH];
#pragma unroll
      for (int t = 0; t < NPT; ++t)
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
          const uint4* f = AvF + ((t * KS2 + ks) * NP) * 64 + lane;
          ca[t * KS2 + ks] = vt[it % (VA + 1)][ks];
          cb[t * KS2 + ks].hi = ldf(f); cb[t * KS2 + ks].mid = ldf(f + 64);
          if constexpr (!H2) cb[t * KS2 + ks].lo = ldf(f + 128);
          co[t * KS2 + ks] = cox[t * KS2 + ks] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        }
      xp_mfma_n<H2, NCH>(ca, cb, co, cox);        // D[c = 4 kg + r][p = r16]
#pragma unroll
      for (int t = 0; t < NPT; ++t) {
        f32x4v o = xp_join<H2>(co[t * KS2], cox[t * KS2]);
#pragma unroll
        for (int ks = 1; ks < KS2; ++ks) o += xp_join<H2>(co[t * KS2 + ks], cox[t * KS2 + ks]);
        o *= inv_v;
        if (t * 16 + r16 < PW)   // 4 consecutive channels of one pixel per lane: one 16-byte store
          st4_nt(new_vis + ((long)b * P + p0 + t * 16 + r16) * C + ct * 16 + 4 * kg, make_float4(o[0], o[1], o[2], o[3]));
      }
    }
    XP_STAMP_V(9);
  } else {
    // ---- sentence -> pixel direction up to At (waves 0-3): sum the quarters of D_t, publish the own columns, gather everyone's ----------
    {
      const float* red = reinterpret_cast<const float*>(lds + L::turn);   // the D_t quarters
      for (int e = tg; e < TILES * 64; e += 256) {
        const int tj = e >> 6, l = e & 63, t = tj / NT, j = tj - t * NT;
        float4 a = *reinterpret_cast<const float4*>(red + tj * 256 + l * 4);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
          const float4 w = *reinterpret_cast<const float4*>(red + (q * TILES + tj) * 256 + l * 4);
          a.x += w.x; a.y += w.y; a.z += w.z; a.w += w.w;
        }
        a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale;
        st4_wt(sxr, (((long)b * S + slot) * (NT * 16) + j * 16 + (l & 15)) * 32 + t * 16 + 4 * (l >> 4), a);   // Sx[b][slot][n][32 local pixels]
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave drains its write-through stores ...
    group_sync();
    if (tg == 0)                                        // ... then ONE lane raises the flag
      __hip_atomic_store(&sync[XP_SYNC_FLAGS + b * XP_MAXS + slot], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    XP_STAMP(2);
    load_vv();
    {   // every wave looks at the flags itself (no barrier between the wait and the gather)
      bool ok = true;
      if (lane < S) {
        const unsigned* f = &sync[XP_SYNC_FLAGS + b * XP_MAXS + lane];
        long spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > XP_SPIN) { ok = false; break; }
        }
      }
      if (!ok) atomicExch(&sync[2], epoch);
      __builtin_amdgcn_wave_barrier();   // (no acquire fence: the payload is read with sc1 loads, which do not look at this CU's L1)
    }
    XP_STAMP(5);
    // gather: rows (workgroup s2 of the image, sentence n) of PC 16-byte pieces; a thread keeps its (row-in-sweep, piece) and walks
    // the rows; all its loads in flight at once (one round trip); idle threads ask out of range
    {
      const int PC = ((P + S - 1) / S + 3) >> 2, RPI = 256 / PC;      // pieces per row, rows per sweep of the 256 threads
      const int lr = tg / PC, f4 = (tg - lr * PC) * 4;
      const int tab = (min(lane, S) * P) / S;                         // lane i: first pixel of workgroup i (i = S: P)
      const int rows = S * NT * 16;
      constexpr int GB = 8;
      for (int r0 = 0; r0 < rows; r0 += GB * RPI) {
        float4 w[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const int rw = r0 + u * RPI + lr;
          const bool in = lr < RPI && rw < rows;
          const int s2 = in ? rw / (NT * 16) : 0;
          const int pw2 = __shfl(tab, s2 + 1, 64) - __shfl(tab, s2, 64);
          const int off = (in && f4 < pw2) ? (int)((((long)b * S * (NT * 16) + rw) * 32 + f4) * 4) : 0x7fffffff;
          w[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(sxr, off, 0, 16));
        }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const int rw = r0 + u * RPI + lr;
          const bool in = lr < RPI && rw < rows;
          const int s2 = in ? rw / (NT * 16) : 0, n = rw - s2 * (NT * 16);
          const int q0 = __shfl(tab, s2, 64), pw2 = __shfl(tab, s2 + 1, 64) - q0;
          if (in && f4 < pw2) {
            float* d = AtL + n * XP_ATS + q0 + f4;
            d[0] = w[u].x;
            if (f4 + 1 < pw2) d[1] = w[u].y;
            if (f4 + 2 < pw2) d[2] = w[u].z;
            if (f4 + 3 < pw2) d[3] = w[u].w;
          }
        }
      }
    }
    group_sync();
    XP_STAMP(6);
    // soft-max over the pixels of each sentence row: 4 threads per row, 32 pixels each, one pass (columns >= P and rows >= N
    // become zeros)
    {
      const int n = tg >> 2, q = tg & 3;
      if (n < NT * 16) {
        float* row = AtL + n * XP_ATS;
        float x[32];
        float m = -INFINITY;
#pragma unroll
        for (int u = 0; u < 32; ++u) { x[u] = (q + 4 * u < P) ? row[q + 4 * u] : -INFINITY; m = fmaxf(m, x[u]); }
        m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64));
        float sm = 0.f;
#pragma unroll
        for (int u = 0; u < 32; ++u) { x[u] = (q + 4 * u < P) ? __expf(x[u] - m) : 0.f; sm += x[u]; }
        sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64);
        const float inv = n < N ? 1.f / sm : 0.f;
#pragma unroll
        for (int u = 0; u < 32; ++u) row[q + 4 * u] = x[u] * inv;
      }
    }
    group_sync();
    XP_STAMP(7);
    // At -> bf16 piece planes in MFMA fragment order (every new_lan wave reads all of them) -- their LDS was the other half's
    // partial blocks and Av rows: wait for its third arrival (long past)
    while (__hip_atomic_load(s_epoch_p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 12u) __builtin_amdgcn_s_sleep(0);
    asm volatile("" ::: "memory");
    for (int e = tg; e < NT * 4 * 64; e += 256) {
      const int fk = e >> 6, l = e & 63, j = fk >> 2, ks = fk & 3;
      if (ks < nks) {
        const float* a = AtL + (j * 16 + (l & 15)) * XP_ATS + ks * 32 + (l >> 4) * 8;
        const Split8 sp = xp_split8<H2>(*reinterpret_cast<const float4*>(a), *reinterpret_cast<const float4*>(a + 4), XP_PS);
        uint4* d = AtF + (fk * NP) * 64 + l;
        d[0] = __builtin_bit_cast(uint4, sp.hi); d[64] = __builtin_bit_cast(uint4, sp.mid);
        if constexpr (!H2) d[128] = __builtin_bit_cast(uint4, sp.lo);
      }
    }
    XP_STAMP(10);
  }
  // the Vv rows of the own unit as bf16 pieces (registers): only LDS traffic and MFMAs follow the barrier
  Split4 vsp[VR];
  {
    const int prow = lane >> 3;
#pragma unroll
    for (int i = 0; i < VR; ++i) {
      const float4 v = i * 8 + prow < P ? vreg[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (H2) vsp[i] = split4h(v, s_vv);
      else vsp[i] = split4(v);
    }
  }
  XP_STAMP_W(24);
  lds_barrier();   // both directions meet
  XP_STAMP(4);
  for (int e = tid; e < PW * N; e += 512) {   // the AtT plane [P][N] of the own pixels that the backward pass reads
    const int lp = e / N, n = e - lp * N;
    probs[(((long)b * 4 + 2) * P + p0 + lp) * N + n] = AtL[n * XP_ATS + p0 + lp];
  }
  // new_lan[b, :, unit] = At . Vv[b, :, unit]: wave w = unit w of the workgroup; per 32-pixel step the pieces of the Vv rows are
  // stored k-major in the wave's own plane buffer and gathered as MFMA fragments by the transpose read
  if (wave < NU) {
    char* pl = lds + wave * (NP * XP_PLANE);
    const float inv_n = H2 ? 1.0f / (XP_PS * s_vv) : 1.0f;
    const int prow = lane >> 3, c8 = (lane & 7) * 8;
    const int k0 = kg * 8 + (r16 >> 2);
    const int tro0 = k0 * XP_KS + 8 * (r16 & 3), tro1 = (k0 + 4) * XP_KS + 8 * (r16 & 3);
    auto trf = [&](const char* plane, int ct) {
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(plane + tro0 + ct * 32));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(plane + tro1 + ct * 32));
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(bf16x8, v);
    };
    f32x4v co[2 * NT], cox[2 * NT];
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) co[i] = cox[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < nks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int vi = ks * 4 + i;
          char* d = pl + (i * 8 + prow) * XP_KS + c8;
          const uint2 z = make_uint2(0u, 0u);
          *reinterpret_cast<uint2*>(d) = vi < VR ? vsp[vi < VR ? vi : 0].hi : z;
          *reinterpret_cast<uint2*>(d + XP_PLANE) = vi < VR ? vsp[vi < VR ? vi : 0].mid : z;
          if constexpr (!H2) *reinterpret_cast<uint2*>(d + 2 * XP_PLANE) = vi < VR ? vsp[vi < VR ? vi : 0].lo : z;
        }
        wave_lds_fence();
        Split8 ca[2 * NT], cb[2 * NT];
        Split8 va[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          va[ct].hi = trf(pl, ct); va[ct].mid = trf(pl + XP_PLANE, ct);
          if constexpr (!H2) va[ct].lo = trf(pl + 2 * XP_PLANE, ct);
          else va[ct].lo = va[ct].mid;
        }
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const uint4* f = AtF + ((j * 4 + ks) * NP) * 64 + lane;
          Split8 at;
          at.hi = ldf(f); at.mid = ldf(f + 64);
          if constexpr (!H2) at.lo = ldf(f + 128);
          else at.lo = at.mid;
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) { ca[ct * NT + j] = va[ct]; cb[ct * NT + j] = at; }
        }
        xp_mfma_n<H2, 2 * NT>(ca, cb, co, cox);      // D[c = 4 kg + r][n = r16]
      }
    }
    const int c0 = (u0 + wave) * 32;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        if (j * 16 + r16 < N) {
          const f32x4v o = xp_join<H2>(co[ct * NT + j], cox[ct * NT + j]) * inv_n;
          st4_nt(new_lan + ((long)b * N + j * 16 + r16) * C + c0 + ct * 16 + 4 * kg, make_float4(o[0], o[1], o[2], o[3]));
        }
  }
  XP_STAMP(8);
  // ---- the last workgroup to finish advances the epoch ----------------------------------------------------------------------------------
  if (tid == 0) {
    const unsigned t = atomicAdd(&sync[1], 1u);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sync[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ======================================================================================================================== backward
// The backward of the pair on the saved probabilities (ops.XAttnFn.backward), cut by pixel rows like the forward: workgroup s of image b
// owns pixels [s P / S, (s + 1) P / S) and computes, for its rows,
//   dAv  = d_vis . Vt^T            (all channels)          dS1 = Av  o (dAv  - rowsum_n(Av o dAv)) / sqrt(C)      dQv = dS1 . Kt
//   dAtT = Vv . d_lan[b]^T         (all channels)          dS2 = AtT o (dAtT - colsum_p(AtT o dAtT)) / sqrt(C)    dKv = dS2 . Qt
//   dVv  = AtT . d_lan[b]
// i.e. two products of the forward's "logits" form (pixel rows from HBM x sentence fragments from L2; waves 0-3 / 4-7, a quarter of the
// channels each) and three of its "new_vis" form (k = sentences).  Only the column sums of the sentence -> pixel soft-max cross the
// workgroup: N floats per workgroup through the forward's fence-free write-through protocol (same sync words, same epoch).  The
// three [N, C] gradients that sum over images and pixels (dVt = Av^T d_vis, dKt = dS1^T Qv, dQt = dS2^T Kv) stay split-K products of
// the GEMM core on dS1 / dS2 / Av, which this launch leaves in dS [3][B, P, N].  The sentence-side operands arrive as piece planes in
// fragment order from xattn_bwd_planes_kernel: Vt and d_lan[b] in the A layout (rows = sentences, k = channels), Kt, Qt and d_lan[b]
// in the B layout (k = sentences, columns = channels).  h2: dS1 / dS2 take a power-of-two scale from the largest magnitude of the
// workgroup's own rows (a product only needs ONE scale per operand, and the rows of a workgroup are a product of their own).
template <int NT, int NPT> struct XbLds {
  static constexpr int turn = 8 * NPT * 16 * XP_TLD * 4;     // phase 1: wave-private turn-around tiles; afterwards three sets of row planes
  static constexpr int red = 8 * NPT * NT * 1024;            // partial blocks of the eight waves
  static constexpr int rows = NPT * 16 * XP_AVS * 4;         // one [own pixels][64] fp32 block
  static constexpr int misc = turn + red + 3 * rows;         // column sums [64], epoch, counters, amax words
  static constexpr int total = misc + 64 * 4 + 32;
};

// out[b, own pixels, :] = R . T over k = sentences: R = the workgroup's rows as piece planes in LDS (AF, [NPT * KS2][NP][64] x 16 B),
// T = a sentence operand as B-layout planes in global memory (BF); the four waves of a half own the channel tiles wv + 4 i (the
// forward's new_vis loop)
template <int NT, int NPT, int KQ, bool H2>
__device__ __forceinline__ void xp_rows_product(const uint4* AF, const uint4* BF, float* __restrict__ out, const float inv, const int wv,
                                                const int lane, const int b, const int P, const int p0, const int PW, const int N,
                                                unsigned* __restrict__ am_out) {
  unsigned am = 0u;
  constexpr int C = 128 * KQ, NP = H2 ? 2 : 3, KS2 = (NT + 1) / 2, CT = C / 16, NIT = CT / 4;
  const int r16 = lane & 15, kg = lane >> 4;
  const __amdgpu_buffer_rsrc_t vtr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(BF), 0, CT * KS2 * NP * 1024, 0x00020000);
  constexpr int VA = XP_VA;
  Split8 vt[VA + 1][KS2];
  auto load_vt = [&](int ct, Split8 (&d)[KS2]) {
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
      const int off = (ks * 32 + kg * 8 < N) ? (((ct * KS2 + ks) * NP) * 64 + lane) * 16 : 0x7fffffff;
      d[ks].hi = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(vtr, off, 0, 0));
      d[ks].mid = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(vtr, off, 1024, 0));
      if constexpr (!H2) d[ks].lo = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(vtr, off, 2048, 0));
      else d[ks].lo = d[ks].mid;
    }
  };
#pragma unroll
  for (int it = 0; it < VA; ++it) load_vt(wv + 4 * it, vt[it]);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int ct = wv + 4 * it;
    if (it + VA < NIT) load_vt(ct + 4 * VA, vt[(it + VA) % (VA + 1)]);
    constexpr int NCH = NPT * KS2;
    Split8 ca[NCH], cb[NCH];
    f32x4v co[NCH], cox[NCH];
#pragma unroll
    for (int t = 0; t < NPT; ++t)
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) {
        const uint4* f = AF + ((t * KS2 + ks) * NP) * 64 + lane;
        ca[t * KS2 + ks] = vt[it % (VA + 1)][ks];
        cb[t * KS2 + ks].hi = ldf(f); cb[t * KS2 + ks].mid = ldf(f + 64);
        if constexpr (!H2) cb[t * KS2 + ks].lo = ldf(f + 128);
        else cb[t * KS2 + ks].lo = cb[t * KS2 + ks].mid;
        co[t * KS2 + ks] = cox[t * KS2 + ks] = (f32x4v){0.f, 0.f, 0.f, 0.f};
      }
    xp_mfma_n<H2, NCH>(ca, cb, co, cox);        // D[c = 4 kg + r][p = r16]
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
      f32x4v o = xp_join<H2>(co[t * KS2], cox[t * KS2]);
#pragma unroll
      for (int ks = 1; ks < KS2; ++ks) o += xp_join<H2>(co[t * KS2 + ks], cox[t * KS2 + ks]);
      o *= inv;
      if (t * 16 + r16 < PW) {
        const float4 o4 = make_float4(o[0], o[1], o[2], o[3]);
        st4_nt(out + ((long)b * P + p0 + t * 16 + r16) * C + ct * 16 + 4 * kg, o4);
        am = max(am, abits4(o4));
      }
    }
  }
  if (am_out != nullptr) amax_commit(am, am_out);   // (the amax word of `out`: its consumers are h2 products)
}

// rows [NPT * 16][XP_AVS] fp32 in LDS (columns >= N and rows >= PW hold zeros) -> piece planes in fragment order; 256 threads of a half
template <int NT, int NPT, bool H2>
__device__ __forceinline__ void xp_rows_to_planes(const float* rows, uint4* AF, const float s, const int tg) {
  constexpr int NP = H2 ? 2 : 3, KS2 = (NT + 1) / 2;
  for (int e = tg; e < NPT * KS2 * 64; e += 256) {
    const int fk = e >> 6, l = e & 63, t = fk / KS2, ks = fk - t * KS2;
    const float* a = rows + (t * 16 + (l & 15)) * XP_AVS + ks * 32 + (l >> 4) * 8;
    const Split8 sp = xp_split8<H2>(*reinterpret_cast<const float4*>(a), *reinterpret_cast<const float4*>(a + 4), s);
    uint4* d = AF + (fk * NP) * 64 + l;
    d[0] = __builtin_bit_cast(uint4, sp.hi); d[64] = __builtin_bit_cast(uint4, sp.mid);
    if constexpr (!H2) d[128] = __builtin_bit_cast(uint4, sp.lo);
  }
}

// grid B * S, block 512.  scl (h2): scales of d_vis, Vv, d_lan, Qt, Kt, Vt.
template <int NT, int NPT, int KQ, bool H2>
__global__ __launch_bounds__(512, 2) void xattn_px_bwd_kernel(const float* __restrict__ dvis, const float* __restrict__ Vv,
                                                              const uint4* __restrict__ VtA, const uint4* __restrict__ dlA,
                                                              const uint4* __restrict__ KtB, const uint4* __restrict__ QtB,
                                                              const uint4* __restrict__ dlB, long a_stride, long b_stride,
                                                              const float* __restrict__ probs, float* __restrict__ dQv,
                                                              float* __restrict__ dKv, float* __restrict__ dVv,
                                                              float* __restrict__ dS, float* __restrict__ Sx, int sx_bytes,
                                                              unsigned* __restrict__ sync, int B, int P, int N, int S, float scale,
                                                              const float* __restrict__ scl, unsigned* __restrict__ am_dqv,
                                                              unsigned* __restrict__ am_dkv, unsigned* __restrict__ am_dvv) {
  constexpr int C = 128 * KQ;
  constexpr int NP = H2 ? 2 : 3;
  constexpr int KST = C / 32;
  constexpr int KS2 = (NT + 1) / 2;
  constexpr int TILES = NPT * NT;
  constexpr int SET = NPT * KS2 * NP * 1024;      // bytes of one set of row planes
  static_assert(3 * SET <= XbLds<NT, NPT>::turn, "the three plane sets live in the turn-around tiles' space");
  using L = XbLds<NT, NPT>;
  const float s_dv = H2 ? scl[0] : 1.f, s_vv = H2 ? scl[1] : 1.f, s_dl = H2 ? scl[2] : 1.f;
  const float s_qt = H2 ? scl[3] : 1.f, s_kt = H2 ? scl[4] : 1.f, s_vt = H2 ? scl[5] : 1.f;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* rowsV = reinterpret_cast<float*>(lds + L::turn + L::red);
  float* rowsT = rowsV + L::rows / 4;
  float* rowsP = rowsT + L::rows / 4;
  float* csum = reinterpret_cast<float*>(lds + L::misc);
  unsigned* s_epoch_p = reinterpret_cast<unsigned*>(lds + L::misc + 64 * 4);   // [0] epoch, [1] [2] arrivals, [3] [4] amax of dS1 / dS2
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  const int b = blockIdx.x / S, slot = blockIdx.x - b * S;
  const int p0 = (slot * P) / S, p1 = ((slot + 1) * P) / S, PW = p1 - p0;
  if (tid == 0) {
    *s_epoch_p = __hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    s_epoch_p[1] = 0u; s_epoch_p[2] = 0u; s_epoch_p[3] = 0u; s_epoch_p[4] = 0u;
  }
  const __amdgpu_buffer_rsrc_t sxr = __builtin_amdgcn_make_buffer_rsrc(Sx, 0, sx_bytes, 0x00020000);
  // ---- the two all-channel products: waves 0-3 E_t = Vv rows . d_lan[b]^T, waves 4-7 E_v = d_vis rows . Vt^T -----------------------------
  {
    const int g = wave >> 2, q = wave & 3;
    const float* X = g ? dvis : Vv;
    const uint4* F = (g ? VtA : dlA + (long)b * a_stride) + ((long)(q * KQ) * NP) * 64 + lane;
    const float s_x = g ? s_dv : s_vv;
    const float inv_l = H2 ? 1.0f / (s_x * (g ? s_vt : s_dl)) : 1.0f;
    float* tile = reinterpret_cast<float*>(lds) + wave * (NPT * 16 * XP_TLD);
    const int lr = lane >> 3, lc = (lane & 7) * 4;
    const float* gp[NPT][2];
#pragma unroll
    for (int t = 0; t < NPT; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h) gp[t][h] = X + ((long)b * P + min(p0 + t * 16 + lr + 8 * h, p1 - 1)) * C + q * (C / 4) + lc;
    f32x4v acc[TILES], acx[TILES];
#pragma unroll
    for (int i = 0; i < TILES; ++i) acc[i] = acx[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    constexpr int AHEAD = KQ < 3 ? KQ : 3;
    float4 v[KQ][NPT][2];
    Split8 fr[KQ][NT];
    auto load_px = [&](int i) {
#pragma unroll
      for (int t = 0; t < NPT; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) v[i][t][h] = ld4_nt(gp[t][h] + i * 32);
    };
    auto load_fr = [&](int i) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const uint4* f = F + ((long)(j * KST + i) * NP) * 64;
        fr[i][j].hi = ldf(f); fr[i][j].mid = ldf(f + 64);
        if constexpr (!H2) fr[i][j].lo = ldf(f + 128);
        else fr[i][j].lo = fr[i][j].mid;
      }
    };
    constexpr int FAHEAD = (KQ < 2 || NT * NPT >= 8) ? 1 : 2;
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) load_px(i);
#pragma unroll
    for (int i = 0; i < FAHEAD; ++i) load_fr(i);
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      if (i + FAHEAD < KQ) load_fr(i + FAHEAD);
      if (i + AHEAD < KQ) load_px(i + AHEAD);
#pragma unroll
      for (int t = 0; t < NPT; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) *reinterpret_cast<float4*>(tile + t * (16 * XP_TLD) + (lr + 8 * h) * XP_TLD + lc) = v[i][t][h];
      wave_lds_fence();
      Split8 sp[NPT];
#pragma unroll
      for (int t = 0; t < NPT; ++t) {
        const float* f = tile + t * (16 * XP_TLD) + r16 * XP_TLD + kg * 8;
        sp[t] = xp_split8<H2>(*reinterpret_cast<const float4*>(f), *reinterpret_cast<const float4*>(f + 4), s_x);
      }
      wave_lds_fence();
      Split8 ca[TILES], cb[TILES];
#pragma unroll
      for (int t = 0; t < NPT; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j) { ca[t * NT + j] = sp[t]; cb[t * NT + j] = fr[i][j]; }
      xp_mfma_n<H2, TILES>(ca, cb, acc, acx);
    }
    float* red = reinterpret_cast<float*>(lds + L::turn) + wave * (TILES * 256);
#pragma unroll
    for (int i = 0; i < TILES; ++i) {
      const f32x4v a = xp_join<H2>(acc[i], acx[i]) * inv_l;
      *reinterpret_cast<float4*>(red + i * 256 + lane * 4) = make_float4(a[0], a[1], a[2], a[3]);
    }
  }
  lds_barrier();
  const unsigned epoch = *s_epoch_p;
  const int grp = wave >> 2, tg = tid & 255;
  unsigned* gctr = s_epoch_p + 1 + grp;
  unsigned gtarget = 0;
  auto group_sync = [&]() {
    gtarget += 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(gctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(gctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < gtarget) __builtin_amdgcn_s_sleep(0);
    asm volatile("" ::: "memory");
  };
  // the quarters of this half's product -> rows [own pixel][sentence] (pixels are the MFMA rows: a lane holds four pixels of one sentence)
  auto sum_quarters = [&](float* rows) {
    const float* red = reinterpret_cast<const float*>(lds + L::turn) + (grp ? 4 * TILES * 256 : 0);
    for (int e = tg; e < TILES * 64; e += 256) {
      const int tj = e >> 6, l = e & 63, t = tj / NT, j = tj - t * NT;
      float4 a = *reinterpret_cast<const float4*>(red + tj * 256 + l * 4);
#pragma unroll
      for (int q = 1; q < 4; ++q) {
        const float4 w = *reinterpret_cast<const float4*>(red + (q * TILES + tj) * 256 + l * 4);
        a.x += w.x; a.y += w.y; a.z += w.z; a.w += w.w;
      }
      const int pl = t * 16 + 4 * (l >> 4), n = j * 16 + (l & 15);
      rows[(pl + 0) * XP_AVS + n] = a.x; rows[(pl + 1) * XP_AVS + n] = a.y;
      rows[(pl + 2) * XP_AVS + n] = a.z; rows[(pl + 3) * XP_AVS + n] = a.w;
    }
  };
  // largest magnitude of this half's dS rows -> its h2 scale (integer max: order-independent)
  auto half_scale = [&](float m) -> float {
    if constexpr (!H2) return 1.f;
    unsigned u = __builtin_bit_cast(unsigned, m);
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, sft, 64));
    if (lane == 0 && u != 0u) __hip_atomic_fetch_max(s_epoch_p + 3 + grp, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    group_sync();
    return h2_scale_from_bits(__hip_atomic_load(s_epoch_p + 3 + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  };
  const int px = tg >> 2, q4 = tg & 3;                       // the row functions: four threads per pixel row, <= 16 sentences each
  const bool rowok = px < NPT * 16;
  const long prow = ((long)b * 4) * P + p0 + min(px, PW > 0 ? PW - 1 : 0);   // row of plane 0 in probs [B][4][P][N]
  const long BPN = (long)B * P * N;
  uint4* setV = reinterpret_cast<uint4*>(lds);               // dS1 planes
  uint4* setA = reinterpret_cast<uint4*>(lds + SET);         // AtT planes
  uint4* setT = reinterpret_cast<uint4*>(lds + 2 * SET);     // dS2 planes
  if (grp == 1) {
    // ---- pixel -> sentence direction: dS1, dQv; and dVv, which needs nothing but the saved AtT --------------------------------------------
    sum_quarters(rowsV);
    group_sync();
    float y[16], mx = 0.f;
    if (rowok) {
      float* row = rowsV + px * XP_AVS;
      float a[16], dot = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int n = q4 + 4 * u;
        a[u] = (px < PW && n < N) ? probs[prow * N + n] : 0.f;
        y[u] = n < NT * 16 ? row[n] : 0.f;     // (columns >= 16 NT of the row block were never written)
        dot += a[u] * y[u];
      }
      dot += __shfl_xor(dot, 1, 64); dot += __shfl_xor(dot, 2, 64);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int n = q4 + 4 * u;
        y[u] = scale * a[u] * (y[u] - dot);
        mx = fmaxf(mx, fabsf(y[u]));
        row[n] = y[u];
        if (px < PW && n < N) {
          const long o = ((long)b * P + p0 + px) * N + n;
          dS[o] = y[u];
          dS[2 * BPN + o] = a[u];
        }
      }
    }
    const float s1 = half_scale(mx);     // (a group barrier in h2)
    if constexpr (!H2) group_sync();
    xp_rows_to_planes<NT, NPT, H2>(rowsV, setV, s1, tg);
    // AtT rows of the own pixels straight from the saved plane -> planes (probabilities: fixed scale)
    for (int e = tg; e < NPT * KS2 * 64; e += 256) {
      const int fk = e >> 6, l = e & 63, t = fk / KS2, ks = fk - t * KS2;
      const int pl = t * 16 + (l & 15), n0 = ks * 32 + (l >> 4) * 8;
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = (pl < PW && n0 + i < N) ? probs[(((long)b * 4 + 2) * P + p0 + pl) * N + n0 + i] : 0.f;
      const Split8 sp = xp_split8<H2>(make_float4(x[0], x[1], x[2], x[3]), make_float4(x[4], x[5], x[6], x[7]), XP_PS);
      uint4* d = setA + (fk * NP) * 64 + l;
      d[0] = __builtin_bit_cast(uint4, sp.hi); d[64] = __builtin_bit_cast(uint4, sp.mid);
      if constexpr (!H2) d[128] = __builtin_bit_cast(uint4, sp.lo);
    }
    group_sync();
    xp_rows_product<NT, NPT, KQ, H2>(setV, KtB, dQv, H2 ? 1.0f / (s1 * s_kt) : 1.0f, wave - 4, lane, b, P, p0, PW, N, am_dqv);
    xp_rows_product<NT, NPT, KQ, H2>(setA, dlB + (long)b * b_stride, dVv, H2 ? 1.0f / (XP_PS * s_dl) : 1.0f, wave - 4, lane, b, P, p0, PW, N, am_dvv);
  } else {
    // ---- sentence -> pixel direction: column sums over ALL pixels of the image (one hand-off of N floats), dS2, dKv ------------------------
    sum_quarters(rowsT);
    group_sync();
    float a[16], x[16];
    if (rowok) {
      const float* row = rowsT + px * XP_AVS;
      float* prd = rowsP + px * XP_AVS;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int n = q4 + 4 * u;
        a[u] = (px < PW && n < N) ? probs[(prow + 2 * P) * N + n] : 0.f;
        x[u] = n < NT * 16 ? row[n] : 0.f;
        prd[n] = a[u] * x[u];
      }
    }
    group_sync();
    if (tg < NT * 4) {   // own partial column sums, four columns per thread, rows in order (deterministic) -> published write-through
      float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int r = 0; r < NPT * 16; ++r) {
        const float4 w = *reinterpret_cast<const float4*>(rowsP + r * XP_AVS + 4 * tg);
        cs.x += w.x; cs.y += w.y; cs.z += w.z; cs.w += w.w;
      }
      st4_wt(sxr, ((long)b * S + slot) * (NT * 16) + 4 * tg, cs);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    group_sync();
    if (tg == 0) __hip_atomic_store(&sync[XP_SYNC_FLAGS + b * XP_MAXS + slot], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
      bool ok = true;
      if (lane < S) {
        const unsigned* f = &sync[XP_SYNC_FLAGS + b * XP_MAXS + lane];
        long spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > XP_SPIN) { ok = false; break; }
        }
      }
      if (!ok) atomicExch(&sync[2], epoch);
      __builtin_amdgcn_wave_barrier();
    }
    if (tg < NT * 4) {   // everyone's partial sums, workgroups in order
      float4 w[XP_MAXS];
#pragma unroll
      for (int s2 = 0; s2 < XP_MAXS; ++s2)
        w[s2] = s2 < S ? ld4_wt(sxr, ((long)b * S + s2) * (NT * 16) + 4 * tg) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 cs = w[0];
#pragma unroll
      for (int s2 = 1; s2 < XP_MAXS; ++s2) { cs.x += w[s2].x; cs.y += w[s2].y; cs.z += w[s2].z; cs.w += w[s2].w; }
      *reinterpret_cast<float4*>(csum + 4 * tg) = cs;
    }
    group_sync();
    float mx = 0.f;
    if (rowok) {
      float* row = rowsT + px * XP_AVS;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int n = q4 + 4 * u;
        const float y = (n < NT * 16) ? scale * a[u] * (x[u] - csum[n]) : 0.f;   // (csum holds 16 NT columns)
        mx = fmaxf(mx, fabsf(y));
        row[n] = y;
        if (px < PW && n < N) dS[BPN + ((long)b * P + p0 + px) * N + n] = y;
      }
    }
    const float s2s = half_scale(mx);
    if constexpr (!H2) group_sync();
    xp_rows_to_planes<NT, NPT, H2>(rowsT, setT, s2s, tg);
    group_sync();
    xp_rows_product<NT, NPT, KQ, H2>(setT, QtB, dKv, H2 ? 1.0f / (s2s * s_qt) : 1.0f, wave, lane, b, P, p0, PW, N, am_dkv);
  }
  // ---- the last workgroup to finish advances the epoch ----------------------------------------------------------------------------------
  if (tid == 0) {
    const unsigned t = atomicAdd(&sync[1], 1u);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sync[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// sentence-side operands of the backward -> piece planes in fragment order.  grid (slots / 256, 3 + 2 B): job 0 Vt -> A layout; 1 Kt,
// 2 Qt -> B layout; 3 + b: d_lan[b] -> A layout; 3 + B + b: d_lan[b] -> B layout.  am.w = amax words of d_vis, Vv, d_lan, Qt, Kt, Vt.
template <bool H2>
__global__ __launch_bounds__(256) void xattn_bwd_planes_kernel(const float* __restrict__ Qt, const float* __restrict__ Kt,
                                                               const float* __restrict__ Vt, const float* __restrict__ dlan,
                                                               uint4* __restrict__ VtA, uint4* __restrict__ KtB,
                                                               uint4* __restrict__ QtB, uint4* __restrict__ dlA, uint4* __restrict__ dlB,
                                                               long a_stride, long b_stride, int B, int N, int C, int NT, int KS2,
                                                               XpAmax am, float* __restrict__ scl_out) {
  constexpr int NP = H2 ? 2 : 3;
  const int job = blockIdx.y;
  const int KST = C / 32;
  const int nA = NT * KST * 64, nB = (C / 16) * KS2 * 64;
  const bool alay = job == 0 || (job >= 3 && job < 3 + B);
  const float* src;
  uint4* dst;
  int widx;
  if (job == 0) { src = Vt; dst = VtA; widx = 5; }
  else if (job == 1) { src = Kt; dst = KtB; widx = 4; }
  else if (job == 2) { src = Qt; dst = QtB; widx = 3; }
  else if (job < 3 + B) { src = dlan + (long)(job - 3) * N * C; dst = dlA + (long)(job - 3) * a_stride; widx = 2; }
  else { src = dlan + (long)(job - 3 - B) * N * C; dst = dlB + (long)(job - 3 - B) * b_stride; widx = 2; }
  float sc = 1.f;
  if constexpr (H2) {
    sc = h2_scale_from_bits(h2_amax_of(am.w[widx], threadIdx.x & 63));
    if (job == 0) {
      const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
      if (t < 6) {
        const float v = h2_scale_from_bits(h2_amax_of(am.w[t], threadIdx.x & 63));
        if ((threadIdx.x & 63) == 0) scl_out[t] = v;
      }
    }
  }
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= (alay ? nA : nB)) return;
  const int l = g & 63, fs = g >> 6;
  float x[8];
  if (alay) {
    const int j = fs / KST, s = fs - j * KST;
    const int n = j * 16 + (l & 15), c = s * 32 + (l >> 4) * 8;
    const float* p = src + (long)min(n, N - 1) * C + c;
    const float4 u = xp_ld4(p), w = xp_ld4(p + 4);
    const bool ok = n < N;
    x[0] = ok ? u.x : 0.f; x[1] = ok ? u.y : 0.f; x[2] = ok ? u.z : 0.f; x[3] = ok ? u.w : 0.f;
    x[4] = ok ? w.x : 0.f; x[5] = ok ? w.y : 0.f; x[6] = ok ? w.z : 0.f; x[7] = ok ? w.w : 0.f;
  } else {
    const int ct = fs / KS2, ks = fs - ct * KS2;
    const int c = ct * 16 + (l & 15), n0 = ks * 32 + (l >> 4) * 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = (n0 + q < N) ? src[(long)(n0 + q) * C + c] : 0.f;
  }
  uint4* d = dst + ((long)fs * NP) * 64 + l;
  const Split8 sp = xp_split8<H2>(make_float4(x[0], x[1], x[2], x[3]), make_float4(x[4], x[5], x[6], x[7]), sc);
  d[0] = __builtin_bit_cast(uint4, sp.hi);
  d[64] = __builtin_bit_cast(uint4, sp.mid);
  if constexpr (!H2) d[128] = __builtin_bit_cast(uint4, sp.lo);
}

struct XbPlan { long vta, ktb, qtb, dla, dlb, sx, scl, total, a_stride, b_stride; int NT, KS2; };
inline XbPlan xb_plan(int B, int N, int C) {
  XbPlan p;
  p.NT = (N + 15) / 16;
  p.KS2 = (p.NT + 1) / 2;
  const long a = (long)p.NT * (C / 32) * 3 * 64 * 16, v = (long)(C / 16) * p.KS2 * 3 * 64 * 16;   // (sized for three pieces)
  p.a_stride = a / 16; p.b_stride = v / 16;
  p.vta = 0; p.ktb = a; p.qtb = a + v; p.dla = a + 2 * v; p.dlb = p.dla + (long)B * a;
  p.sx = p.dlb + (long)B * v;
  p.scl = p.sx + (long)B * XP_MAXS * p.NT * 16 * 4;
  p.total = p.scl + 64;
  return p;
}

struct XpPlan { long qtf, ktf, vtf, sx, scl, total; int NT, KS2; };
inline XpPlan xp_plan(int B, int N, int C) {
  XpPlan p;
  p.NT = (N + 15) / 16;
  p.KS2 = (p.NT + 1) / 2;
  const long a = (long)p.NT * (C / 32) * 3 * 64 * 16, v = (long)(C / 16) * p.KS2 * 3 * 64 * 16;
  p.qtf = 0; p.ktf = a; p.vtf = 2 * a; p.sx = 2 * a + v;
  p.total = p.sx + (long)B * XP_MAXS * p.NT * 16 * 32 * 4;
#ifdef TRIS_XP_TRACE
  p.total += (long)B * XP_MAXS * 32 * 8;
#endif
  p.scl = p.total;     // six h2 scales (the planes above are sized for three pieces; the h2 form uses two)
  p.total += 64;
  return p;
}

// workgroups per image: as many as fit one per CU (<= 8), at least what 32 own pixels / 8 own units per workgroup need
inline int xp_slots(int B, int P, int C, int cus) {
  const int U = C / 32;
  int smin = (P + 31) / 32;
  if ((U + XP_MAXS - 1) / XP_MAXS > smin) smin = (U + XP_MAXS - 1) / XP_MAXS;
  int smax = cus / B;
  if (smax > XP_MAXS) smax = XP_MAXS;
  if (smax > P) smax = P;
  if (tris_internal_xattn_px_slots > 0 && tris_internal_xattn_px_slots < smax) smax = tris_internal_xattn_px_slots;
  return smax >= smin ? smax : 0;
}

template <int NT, int NPT, int KQ, bool H2>
int launch_px(const float* Qv, const float* Kv, const float* Vv, const float* Qt, const float* Kt, const float* Vt,
              float* new_vis, float* new_lan, float* probs, int B, int P, int N, int C, int S, char* ws, unsigned* sync,
              hipStream_t st, const XpAmax& am) {
  const XpPlan pl = xp_plan(B, N, C);
  uint4* QtF = reinterpret_cast<uint4*>(ws + pl.qtf);
  uint4* KtF = reinterpret_cast<uint4*>(ws + pl.ktf);
  uint4* VtF = reinterpret_cast<uint4*>(ws + pl.vtf);
  float* Sx = reinterpret_cast<float*>(ws + pl.sx);
  float* scl = reinterpret_cast<float*>(ws + pl.scl);
  const int slots = 2 * NT * (C / 32) * 64 + (C / 16) * pl.KS2 * 64;
  hipLaunchKernelGGL(xattn_text_planes_kernel<H2>, dim3(cdiv(slots, 256)), dim3(256), 0, st, Qt, Kt, Vt, QtF, KtF, VtF, N, C, NT,
                     pl.KS2, am, scl);
  constexpr int lds = XpLds<NT, NPT, H2 ? 2 : 3>::total;
  static bool attr_done = false;   // (one instantiation = one function-local flag)
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_px_kernel<NT, NPT, KQ, H2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds > 65536 ? lds : 65536);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  hipLaunchKernelGGL((xattn_px_kernel<NT, NPT, KQ, H2>), dim3(B * S), dim3(512), (size_t)lds, st, Qv, Kv, Vv, QtF, KtF, VtF,
                     new_vis, new_lan, probs, Sx, (int)(pl.scl - pl.sx), sync, B, P, N, S, 1.0f / sqrtf((float)C), scl);
  TRIS_LAUNCH_CHECK();
  return 0;
}

int xp_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 1;
    cus = n;
  }
  return cus;
}

}  // namespace

// h2 for ONE call: arms the calling thread with the amax words of Qv, Kv, Vv, Qt, Kt, Vt (ops.XAttnFn passes the words of the
// producing products); the next tris_xattn_px_fwd_f32 of the thread runs the two-piece fp16 form and disarms.
static thread_local XpAmax g_xp_amax = {{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}};
static thread_local bool g_xp_amax_armed = false;
extern "C" int tris_xattn_amax_next(const unsigned* qv, const unsigned* kv, const unsigned* vv, const unsigned* qt,
                                    const unsigned* kt, const unsigned* vt) {
  if (!qv || !kv || !vv || !qt || !kt || !vt) { g_xp_amax_armed = false; return (int)hipErrorInvalidValue; }
  g_xp_amax = XpAmax{{qv, kv, vv, qt, kt, vt}};
  g_xp_amax_armed = true;
  return 0;
}

template <int NT, int NPT, int KQ, bool H2>
int launch_px_bwd(const float* dvis, const float* dlan, const float* Vv, const float* Qt, const float* Kt, const float* Vt,
                  const float* probs, float* dQv, float* dKv, float* dVv, float* dS, int B, int P, int N, int C, int S, char* ws,
                  unsigned* sync, hipStream_t st, const XpAmax& am, unsigned* const* am_out) {
  const XbPlan pl = xb_plan(B, N, C);
  uint4* VtA = reinterpret_cast<uint4*>(ws + pl.vta);
  uint4* KtB = reinterpret_cast<uint4*>(ws + pl.ktb);
  uint4* QtB = reinterpret_cast<uint4*>(ws + pl.qtb);
  uint4* dlA = reinterpret_cast<uint4*>(ws + pl.dla);
  uint4* dlB = reinterpret_cast<uint4*>(ws + pl.dlb);
  float* Sx = reinterpret_cast<float*>(ws + pl.sx);
  float* scl = reinterpret_cast<float*>(ws + pl.scl);
  const int nA = NT * (C / 32) * 64, nB = (C / 16) * pl.KS2 * 64;
  hipLaunchKernelGGL(xattn_bwd_planes_kernel<H2>, dim3(cdiv(nA > nB ? nA : nB, 256), 3 + 2 * B), dim3(256), 0, st, Qt, Kt, Vt, dlan, VtA,
                     KtB, QtB, dlA, dlB, pl.a_stride, pl.b_stride, B, N, C, NT, pl.KS2, am, scl);
  constexpr int lds = XbLds<NT, NPT>::total;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_px_bwd_kernel<NT, NPT, KQ, H2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds > 65536 ? lds : 65536);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  hipLaunchKernelGGL((xattn_px_bwd_kernel<NT, NPT, KQ, H2>), dim3(B * S), dim3(512), (size_t)lds, st, dvis, Vv, VtA, dlA, KtB, QtB, dlB,
                     pl.a_stride, pl.b_stride, probs, dQv, dKv, dVv, dS, Sx, (int)(pl.scl - pl.sx), sync, B, P, N, S,
                     1.0f / sqrtf((float)C), scl, am_out[0], am_out[1], am_out[2]);
  TRIS_LAUNCH_CHECK();
  return 0;
}

static int g_xp_last_form = 0;
// arithmetic of the last pixel-row launch of the process: 0 none yet, 1 split-bf16 (x3), 2 h2 (tests, tools/xattn_check.py)
extern "C" long tris_xattn_px_last_form(void) { return g_xp_last_form; }

extern "C" long tris_xattn_px_ws_bytes(int B, int N, int C) {
  if (B < 1 || N < 1 || N > 64 || !(C == 512 || C == 1024)) return 0;
  return xp_plan(B, N, C).total;
}

extern "C" long tris_xattn_px_sync_words(int B) { return XP_SYNC_FLAGS + (long)XP_MAXS * B; }

extern "C" long tris_xattn_px_slots(int B, int P, int C, int cus) {
  if (B < 1 || P < 1 || P > 104 || !(C == 512 || C == 1024) || cus < 1) return 0;
  return xp_slots(B, P, C, cus);
}

extern "C" int tris_xattn_px_fwd_f32(const float* Qv, const float* Kv, const float* Vv, const float* Qt, const float* Kt,
                                     const float* Vt, float* new_vis, float* new_lan, float* probs, int B, int P, int N,
                                     int C, float* ws, long ws_bytes, unsigned* sync, void* stream) {
  // supported: split-bf16 arithmetic (or h2 when armed), C = 512 | 1024, P <= 104, N <= 64, B * S workgroups co-resident one per CU
  const bool h2 = g_xp_amax_armed;
  const XpAmax am = g_xp_amax;
  g_xp_amax_armed = false;
  if (tris_get_gemm_mode() < 1 || !(C == 512 || C == 1024) || P < 1 || P > 104 || N < 1 || N > 64 || B < 1 || ws == nullptr ||
      sync == nullptr || ws_bytes < xp_plan(B, N, C).total)
    return TRIS_DECLINED;
  const int S = xp_slots(B, P, C, xp_cus());
  if (S == 0) return TRIS_DECLINED;
  g_xp_last_form = h2 ? 2 : 1;
  const int npt = ((P + S - 1) / S + 15) / 16;   // pixel tiles of the largest own range
  hipStream_t st = (hipStream_t)stream;
  char* w = reinterpret_cast<char*>(ws);
#define TRIS_XP4(NT_, NPT_, KQ_)                                                                                                \
  (h2 ? launch_px<NT_, NPT_, KQ_, true>(Qv, Kv, Vv, Qt, Kt, Vt, new_vis, new_lan, probs, B, P, N, C, S, w, sync, st, am)        \
      : launch_px<NT_, NPT_, KQ_, false>(Qv, Kv, Vv, Qt, Kt, Vt, new_vis, new_lan, probs, B, P, N, C, S, w, sync, st, am))
#define TRIS_XP3(NT_, NPT_) (C == 1024 ? TRIS_XP4(NT_, NPT_, 8) : TRIS_XP4(NT_, NPT_, 4))
#define TRIS_XP(NT_) (npt == 1 ? TRIS_XP3(NT_, 1) : TRIS_XP3(NT_, 2))
  switch ((N + 15) / 16) {
    case 1: return TRIS_XP(1);
    case 2: return TRIS_XP(2);
    case 3: return TRIS_XP(3);
    default: return TRIS_XP(4);
  }
#undef TRIS_XP
#undef TRIS_XP3
#undef TRIS_XP4
}

extern "C" long tris_xattn_px_bwd_ws_bytes(int B, int N, int C) {
  if (B < 1 || N < 1 || N > 64 || !(C == 512 || C == 1024)) return 0;
  return xb_plan(B, N, C).total;
}

// dQv, dKv, dVv [B, P, C] and dS [3][B, P, N] (dS1, dS2, a copy of Av) from d_vis [B, P, C], d_lan [B, N, C], the forward's operands and
// its saved probabilities probs [B][4][P][N] (planes 0 = Av, 2 = AtT): one preparation launch + ONE persistent launch; same domain,
// sync words and TRIS_DECLINED behaviour as tris_xattn_px_fwd_f32.  tris_xattn_amax_next(d_vis, Vv, d_lan, Qt, Kt, Vt) arms the h2 form.
extern "C" int tris_xattn_px_bwd_f32(const float* d_vis, const float* d_lan, const float* Vv, const float* Qt, const float* Kt,
                                     const float* Vt, const float* probs, float* dQv, float* dKv, float* dVv, float* dS, int B, int P,
                                     int N, int C, float* ws, long ws_bytes, unsigned* sync, unsigned* amax_dQv, unsigned* amax_dKv,
                                     unsigned* amax_dVv, void* stream) {
  unsigned* const am_out[3] = {amax_dQv, amax_dKv, amax_dVv};
  const bool h2 = g_xp_amax_armed;
  const XpAmax am = g_xp_amax;
  g_xp_amax_armed = false;
  if (tris_get_gemm_mode() < 1 || !(C == 512 || C == 1024) || P < 1 || P > 104 || N < 1 || N > 64 || B < 1 || ws == nullptr ||
      sync == nullptr || ws_bytes < xb_plan(B, N, C).total)
    return TRIS_DECLINED;
  const int S = xp_slots(B, P, C, xp_cus());
  if (S == 0) return TRIS_DECLINED;
  const int npt = ((P + S - 1) / S + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  char* w = reinterpret_cast<char*>(ws);
#define TRIS_XB4(NT_, NPT_, KQ_)                                                                                                       \
  (h2 ? launch_px_bwd<NT_, NPT_, KQ_, true>(d_vis, d_lan, Vv, Qt, Kt, Vt, probs, dQv, dKv, dVv, dS, B, P, N, C, S, w, sync, st, am, am_out) \
      : launch_px_bwd<NT_, NPT_, KQ_, false>(d_vis, d_lan, Vv, Qt, Kt, Vt, probs, dQv, dKv, dVv, dS, B, P, N, C, S, w, sync, st, am, am_out))
#define TRIS_XB3(NT_, NPT_) (C == 1024 ? TRIS_XB4(NT_, NPT_, 8) : TRIS_XB4(NT_, NPT_, 4))
#define TRIS_XB(NT_) (npt == 1 ? TRIS_XB3(NT_, 1) : TRIS_XB3(NT_, 2))
  switch ((N + 15) / 16) {
    case 1: return TRIS_XB(1);
    case 2: return TRIS_XB(2);
    case 3: return TRIS_XB(3);
    default: return TRIS_XB(4);
  }
#undef TRIS_XB
#undef TRIS_XB3
#undef TRIS_XB4
}
