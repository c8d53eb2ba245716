// Bilateral image<->text cross attention (reference model/attn.py:117-128) as ONE persistent launch (gfx950, split-bf16 x3).
//
//   Av  = softmax_n(Qv Kt^T / sqrt(C))   [B,P,N]      new_vis = Av  Vt      [B,P,C]
//   AtT = softmax_p(Kv Qt^T / sqrt(C))   [B,P,N]      new_lan = AtT^T Vv    [B,N,C]
//
// Why one launch and how the work is cut.  The pair is an HBM stream (1.84 MB per image at P = 100, N = 48, C = 1024, 21 FLOP/B)
// and a CU moves ~25 GB/s of it, so an image has to be spread over several CUs in EVERY phase -- but the pixel soft-max of the
// sentence->pixel direction couples all pixels of an image.  Eight workgroups per image (B * 8 >= 256 CUs at B = 48):
//   phase A  workgroup s owns pixels [s P/8, (s+1) P/8) (<= 16 rows): both logit blocks of those rows against ALL sentences,
//            S_t[n][p] = Qt[n].Kv[p] and S_v[n][p] = Kt[n].Qv[p], full reduction over C (the 4 waves split C, partial tiles are
//            summed through LDS).  The pixel rows stream from HBM straight into MFMA operand registers (16 rows x 128 B per
//            instruction); the sentence operands arrive PRE-SPLIT into three bf16 planes in MFMA fragment order (one coalesced
//            16-byte load per lane per fragment, no VALU, no LDS), written once per call by xattn_text_planes_kernel.
//   hand-off S_t block (N x 16 floats, 3 KB) -> global, agent-scope release, one flag per (image, slot).  It is issued BEFORE the
//            pixel->sentence half below, whose work hides the peers' latency.
//   phase B1 row soft-max of S_v (local), new_vis rows of the own pixels = Av . Vt  (Vt^T fragments pre-split likewise).
//   phase B2 after the 8 flags of the image: gather the 8 S_t blocks (24 KB), pixel soft-max (each workgroup redundantly, 48 x 100
//            exponentials), then new_lan[:, c-slice of C/8 channels] = At . Vv[:, slice]: the Vv slice is read row-wise
//            (512 B per pixel), split once, staged k-major in LDS and gathered by the transpose read ds_read_b64_tr_b16.
// So every byte of Qv, Kv, Vv is read once and new_vis / new_lan are written once: HBM traffic = algorithmic + the saved
// probabilities (Av, AtT: 2 %) + 1.2 MB of logit hand-off.  The sentence planes (0.96 MB) are re-read by every workgroup from L2.
//
// Inter-workgroup protocol (MI355X_MICROARCH "workgroup dispatch / visibility"): plain payload stores -> every storing wave
// drains vmcnt -> barrier -> one lane: agent-scope release fence + drained flag store; consumers: relaxed polls of the 8 flags by
// one wave, ONE agent-scope acquire, barrier, plain loads.  Placement-independent.  The epoch that tags the flags lives in device
// memory (sync[0]) and is advanced by the last workgroup to finish, so a captured launch replays correctly; spins are bounded
// (sync[2] != 0 afterwards = a peer never published; the outputs of that launch are then undefined).
#include "common.h"
#include "tris_hip.h"
#include "x3_split.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int XF_SLOTS = 8;       // workgroups per image
constexpr int XF_PP = 112;        // padded pixel count: 7 MFMA tiles of 16 (P <= 104 leaves a zero row in the planes)
constexpr int XF_PK = 104;        // pixel rows of a k-major Vv plane; rows >= P are zero, reads beyond are clamped to row XF_PK - 1
constexpr int XF_KS = 96;         // bytes per pixel row of a k-major Vv plane (32 channels x 2 B + 32: the 4 k rows of a tr read fall on disjoint banks)
constexpr int XF_SYNC_FLAGS = 16; // sync[0] epoch, [1] finish ticket, [2] time-out flag, [16 + (stage*B + b)*8 + s] publish flags of the two exchange stages
constexpr long XF_SPIN = 4000000; // polls before a wait gives up (~seconds)

// developer build (-DTRIS_XF_TRACE, tools/xattn_fused_trace.py): per-workgroup s_memtime stamps at the phase boundaries, written
// behind the hand-off scratch; compiled out of the product
#ifdef TRIS_XF_TRACE
#define XF_STAMP(i) do { if (tid == 0) xf_trace[(long)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define XF_STAMP(i) do { } while (0)
#endif

constexpr int TLD = 36;   // row stride (floats) of a wave-private 16 x 32 turn-around tile (conflict-free b128 fragment reads)
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ bf16x8 ldf(const uint4* p) { return __builtin_bit_cast(bf16x8, *p); }

__device__ __forceinline__ f32x4v mfma6(const Split8& a, const Split8& b, f32x4v c) {   // smallest terms first
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.mid, b.mid, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.mid, b.hi, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.mid, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, c, 0, 0, 0);
  return c;
}

// ---- sentence operands -> bf16 piece planes in MFMA fragment order ------------------------------------------------------------
// QtF / KtF (A operand of phase A: rows = sentences, k = channels):  [NT][C/32][3 planes][64 lanes] x 16 B;
//     lane l of fragment (j, s): sentence n = 16 j + (l & 15), channels c = 32 s + 8 (l >> 4) .. + 7      (n >= N -> zeros)
// VtF (B operand of phase B1: k = sentences, columns = channels):    [C/16][KS2][3 planes][64 lanes] x 16 B;
//     lane l of fragment (ct, ks): channel c = 16 ct + (l & 15), sentences n = 32 ks + 8 (l >> 4) .. + 7   (n >= N -> zeros)
__global__ __launch_bounds__(256) void xattn_text_planes_kernel(const float* __restrict__ Qt, const float* __restrict__ Kt,
                                                                const float* __restrict__ Vt, uint4* __restrict__ QtF,
                                                                uint4* __restrict__ KtF, uint4* __restrict__ VtF, int N, int C,
                                                                int NT, int KS2) {
  const int KST = C / 32;
  const int nA = NT * KST * 64;           // lane slots of one A-operand tensor
  const int nB = (C / 16) * KS2 * 64;     // lane slots of VtF
  const int g = blockIdx.x * 256 + threadIdx.x;
  float x[8];
  uint4* dst;
  if (g < 2 * nA) {
    const int t = g / nA, e = g - t * nA;
    const int l = e & 63, fs = e >> 6;    // fs = j * KST + s
    const int j = fs / KST, s = fs - j * KST;
    const int n = j * 16 + (l & 15), c = s * 32 + (l >> 4) * 8;
    const float* src = (t == 0 ? Qt : Kt) + (long)min(n, N - 1) * C + c;
    const float4 u = ld4(src), w = ld4(src + 4);
    const bool ok = n < N;
    x[0] = ok ? u.x : 0.f; x[1] = ok ? u.y : 0.f; x[2] = ok ? u.z : 0.f; x[3] = ok ? u.w : 0.f;
    x[4] = ok ? w.x : 0.f; x[5] = ok ? w.y : 0.f; x[6] = ok ? w.z : 0.f; x[7] = ok ? w.w : 0.f;
    dst = (t == 0 ? QtF : KtF) + ((long)fs * 3) * 64 + l;
  } else if (g < 2 * nA + nB) {
    const int e = g - 2 * nA;
    const int l = e & 63, fs = e >> 6;    // fs = ct * KS2 + ks
    const int ct = fs / KS2, ks = fs - ct * KS2;
    const int c = ct * 16 + (l & 15), n0 = ks * 32 + (l >> 4) * 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = (n0 + q < N) ? Vt[(long)(n0 + q) * C + c] : 0.f;
    dst = VtF + ((long)fs * 3) * 64 + l;
  } else {
    return;
  }
  const Split8 sp = split8(make_float4(x[0], x[1], x[2], x[3]), make_float4(x[4], x[5], x[6], x[7]));
  dst[0] = __builtin_bit_cast(uint4, sp.hi);
  dst[64] = __builtin_bit_cast(uint4, sp.mid);
  dst[128] = __builtin_bit_cast(uint4, sp.lo);
}

// 8 consecutive k (rows k0 .. k0+7, clamped to zrow) of column m16 + (lane & 15) from a k-major bf16 plane
__device__ __forceinline__ bf16x8 tr_frag8c(const char* plane, int k0, int m16, int lane, int zrow) {
  const int i16 = lane & 15;
  const int ra = min(k0 + (i16 >> 2), zrow), rb = min(k0 + 4 + (i16 >> 2), zrow);
  const int col = (m16 + 4 * (i16 & 3)) * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(plane + ra * XF_KS + col));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(plane + rb * XF_KS + col));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// grid B * 8, block 256, dynamic LDS xf_lds_bytes(P, NT).  KS = 32-channel steps of a workgroup's channel slice (C = 256 KS).
template <int NT, int KS>
__global__ __launch_bounds__(256, 2) void xattn_fused_kernel(const float* __restrict__ Qv, const float* __restrict__ Kv,
                                                             const float* __restrict__ Vv, const uint4* __restrict__ QtF,
                                                             const uint4* __restrict__ KtF, const uint4* __restrict__ VtF,
                                                             float* __restrict__ new_vis, float* __restrict__ new_lan,
                                                             float* __restrict__ probs, float* __restrict__ Sx,
                                                             unsigned* __restrict__ sync, int B, int P, int N, float scale) {
  constexpr int CS = KS * 32;            // channels owned by a workgroup
  constexpr int C = CS * XF_SLOTS;
  constexpr int KST = C / 32;            // 32-channel steps over all of C
  constexpr int KS2 = (NT + 1) / 2;      // 32-sentence steps of the new_vis product
  constexpr int SVS = NT <= 3 ? 52 : 64; // row stride (floats) of the Av plane [p][n] in LDS
  constexpr int PP = XF_PP;              // padded pixel count (7 tiles of 16)
  constexpr int NPASS = CS / 32;         // 32-channel passes of the new_lan product
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // LDS: SvL / Av [PP][SVS] floats | StF / At [NT*16][PP] floats | k-major Vv planes [3][XF_PK][XF_KS bytes] | epoch
  float* SvL = reinterpret_cast<float*>(lds);
  float* StF = SvL + PP * SVS;
  char* planes = reinterpret_cast<char*>(StF + NT * 16 * PP);
  unsigned* s_epoch_p = reinterpret_cast<unsigned*>(planes + 3 * XF_PK * XF_KS);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  const int b = blockIdx.x / XF_SLOTS, slot = blockIdx.x % XF_SLOTS;
  const int NPT = (P + 15) >> 4;
  if (tid == 0) *s_epoch_p = __hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
#ifdef TRIS_XF_TRACE
  unsigned long long* xf_trace = reinterpret_cast<unsigned long long*>(Sx + (long)B * XF_SLOTS * 2 * NT * 16 * PP + (long)B * (NT * 16 * PP + PP * 64));
#endif
  XF_STAMP(0);

  // ---- phase A: partial logits of ALL pixels of the image over the own channel slice ----------------------------------------------------
  // D_t[p][n] = sum_c Kv[p][c] Qt[n][c]   D_v[p][n] = sum_c Qv[p][c] Kt[n][c],  c in [slot CS, (slot + 1) CS).
  // wave w owns the pixel tiles w and w + 4; the sentence fragments of a 32-channel step are loaded once and used for both.
  // Pixel rows are read ROW-CONTIGUOUS (8 rows x 128 B per instruction: fragment-shaped loads -- 16 rows x 32 B -- cost the
  // address / tag path 2-4x as much for the same bytes, measured 18 us for this phase) and turned into MFMA fragments through a
  // wave-private LDS tile (program order + a wavefront fence; no workgroup barrier in the loop).  The pixels are the A operand,
  // so a lane ends up with FOUR CONSECUTIVE pixels of one sentence: the partial blocks leave as 16-byte stores.
  f32x4v acc[2][2][NT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[t][y][j] = (f32x4v){0.f, 0.f, 0.f, 0.f};
  {
    const bool two = wave + 4 < NPT;   // uniform
    float* tk = reinterpret_cast<float*>(lds) + wave * (2 * 16 * TLD);   // wave-private tiles [16][TLD]: Kv, then Qv
    float* tq = tk + 16 * TLD;
    const int lr = lane >> 3, lc = (lane & 7) * 4;   // loader coordinates: row within an 8-row half, float offset in the 128-B piece
    long g0[2], g1[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      g0[h] = ((long)b * P + min(wave * 16 + lr + 8 * h, P - 1)) * C + slot * CS + lc;
      g1[h] = ((long)b * P + min((wave + 4) * 16 + lr + 8 * h, P - 1)) * C + slot * CS + lc;
    }
    float4 vk[KS][2], vq[KS][2], wk[2], wq[2];   // tile 0: all KS steps in flight; tile 1: one step ahead
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int h = 0; h < 2; ++h) { vk[ks][h] = ld4(Kv + g0[h] + ks * 32); vq[ks][h] = ld4(Qv + g0[h] + ks * 32); }
#pragma unroll
    for (int h = 0; h < 2; ++h) { wk[h] = ld4(Kv + g1[h]); wq[h] = ld4(Qv + g1[h]); }
    const uint4* qf = QtF + ((long)(slot * KS) * 3) * 64 + lane;
    const uint4* kf = KtF + ((long)(slot * KS) * 3) * 64 + lane;
    auto turn = [&](const float4 (&k2)[2], const float4 (&q2)[2], Split8& sk, Split8& sq) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        *reinterpret_cast<float4*>(tk + (lr + 8 * h) * TLD + lc) = k2[h];
        *reinterpret_cast<float4*>(tq + (lr + 8 * h) * TLD + lc) = q2[h];
      }
      wave_lds_fence();
      const float* fk = tk + r16 * TLD + kg * 8;
      const float* fq = tq + r16 * TLD + kg * 8;
      sk = split8(*reinterpret_cast<const float4*>(fk), *reinterpret_cast<const float4*>(fk + 4));
      sq = split8(*reinterpret_cast<const float4*>(fq), *reinterpret_cast<const float4*>(fq + 4));
      wave_lds_fence();
    };
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      Split8 fq_[NT], fk_[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const uint4* f = qf + ((long)(j * KST + ks) * 3) * 64;
        const uint4* h = kf + ((long)(j * KST + ks) * 3) * 64;
        fq_[j].hi = ldf(f); fq_[j].mid = ldf(f + 64); fq_[j].lo = ldf(f + 128);
        fk_[j].hi = ldf(h); fk_[j].mid = ldf(h + 64); fk_[j].lo = ldf(h + 128);
      }
      {
        Split8 sk, sq;
        turn(vk[ks], vq[ks], sk, sq);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[0][0][j] = mfma6(sk, fq_[j], acc[0][0][j]);
          acc[0][1][j] = mfma6(sq, fk_[j], acc[0][1][j]);
        }
      }
      {   // (a wave without a second tile computes on its clamped rows and never publishes the result)
        Split8 sk, sq;
        turn(wk, wq, sk, sq);
        if (ks + 1 < KS) {
#pragma unroll
          for (int h = 0; h < 2; ++h) { wk[h] = ld4(Kv + g1[h] + (ks + 1) * 32); wq[h] = ld4(Qv + g1[h] + (ks + 1) * 32); }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[1][0][j] = mfma6(sk, fq_[j], acc[1][0][j]);
          acc[1][1][j] = mfma6(sq, fk_[j], acc[1][1][j]);
        }
      }
    }
    XF_STAMP(1);
    // publish the partial blocks: Sx[b][slot][type][n][p]   (D[p = 16 tile + 4 kg + r][n = 16 j + r16]: 4 consecutive p per lane)
    float* mine = Sx + ((long)b * XF_SLOTS + slot) * (2 * NT * 16 * PP);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t == 0 || two) {
        const int pc = (wave + 4 * t) * 16 + 4 * kg;
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            *reinterpret_cast<float4*>(mine + (long)(y * NT * 16 + j * 16 + r16) * PP + pc) =
                make_float4(acc[t][y][j][0], acc[t][y][j][1], acc[t][y][j][2], acc[t][y][j][3]);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned epoch = *s_epoch_p;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&sync[XF_SYNC_FLAGS + b * XF_SLOTS + slot], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  XF_STAMP(2);
  // ---- everything that does not depend on the exchange is requested now: Vt^T fragments of the own channel tiles (wave w: tiles
  // 2w, 2w+1 of the slice), the Vv slice of the image ----------------------------------------------------------------------------------------
  constexpr int CTW = CS / 64;   // channel tiles per wave in the new_vis product (2 at C = 1024)
  Split8 vt[CTW][KS2];
#pragma unroll
  for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
      const uint4* f = VtF + ((long)(((slot * (CS / 16) + wave * CTW + ct) * KS2 + ks) * 3)) * 64 + lane;
      vt[ct][ks].hi = ldf(f); vt[ct][ks].mid = ldf(f + 64); vt[ct][ks].lo = ldf(f + 128);
    }
  constexpr int VR = 4;   // 8 threads x 16 B per pixel row of a 32-channel pass, 32 rows per sweep: P <= 128
  float4 vreg[NPASS][VR];
  {
    const int prow = tid >> 3, c4 = (tid & 7) * 4;
#pragma unroll
    for (int h = 0; h < NPASS; ++h)
#pragma unroll
      for (int q = 0; q < VR; ++q)
        vreg[h][q] = ld4(Vv + ((long)b * P + min(q * 32 + prow, P - 1)) * C + slot * CS + h * 32 + c4);
  }
  // ---- exchange, stage 1 (reduce-scatter): this workgroup finishes RN sentence rows of S_t and the 16 pixels of tile `slot` of S_v ---
  auto wait_flags = [&](int base) {
    if (wave == 0) {
      bool ok = true;
      if (lane < XF_SLOTS) {
        const unsigned* f = &sync[base + b * XF_SLOTS + lane];
        long spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > XF_SPIN) { ok = false; break; }
        }
      }
      if (!ok) atomicExch(&sync[2], epoch);
      if (lane == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  };
  wait_flags(XF_SYNC_FLAGS);
  XF_STAMP(3);
  constexpr int RN = NT * 2;             // sentence rows finished by a workgroup (NT * 16 / 8)
  constexpr int P4 = PP / 4;
  float* Rt = reinterpret_cast<float*>(lds);            // [RN][PP]   logits -> At rows
  float* Rv = Rt + RN * PP;                             // [16][64]   logits -> Av rows of pixel tile `slot` ([p][n])
  float* Ax = Sx + (long)B * XF_SLOTS * (2 * NT * 16 * PP) + (long)b * (NT * 16 * PP + PP * 64);   // image b: At [NT*16][PP] | Av [PP][64]
  {
    const float* base = Sx + (long)b * XF_SLOTS * (2 * NT * 16 * PP);
    constexpr int NA = RN * P4, NB = NT * 16 * 4;
    for (int e = tid; e < NA + NB; e += 256) {
      long off;
      if (e < NA) { const int rr = e / P4; off = (long)(slot * RN + rr) * PP + (e - rr * P4) * 4; }
      else { const int n = (e - NA) >> 2; off = (long)(NT * 16 + n) * PP + slot * 16 + ((e - NA) & 3) * 4; }
      float4 v[XF_SLOTS];
#pragma unroll
      for (int s2 = 0; s2 < XF_SLOTS; ++s2) v[s2] = ld4(base + (long)s2 * (2 * NT * 16 * PP) + off);
      float4 a = v[0];
#pragma unroll
      for (int s2 = 1; s2 < XF_SLOTS; ++s2) { a.x += v[s2].x; a.y += v[s2].y; a.z += v[s2].z; a.w += v[s2].w; }
      a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale;
      if (e < NA) {
        *reinterpret_cast<float4*>(Rt + e * 4) = a;
      } else {
        const int n = (e - NA) >> 2, px = ((e - NA) & 3) * 4;
        Rv[(px + 0) * 64 + n] = a.x; Rv[(px + 1) * 64 + n] = a.y; Rv[(px + 2) * 64 + n] = a.z; Rv[(px + 3) * 64 + n] = a.w;
      }
    }
  }
  __syncthreads();
  if (tid < 64) {   // Av rows of the 16 pixels of tile `slot`: soft-max over the sentences, 4 threads per pixel
    const int px = tid >> 2, q = tid & 3;
    float* row = Rv + px * 64;
    float m = -INFINITY;
    for (int n = q; n < N; n += 4) m = fmaxf(m, row[n]);
    m = fmaxf(m, __shfl_xor(m, 1, 64));
    m = fmaxf(m, __shfl_xor(m, 2, 64));
    float sm = 0.f;
    for (int n = q; n < N; n += 4) { const float e = __expf(row[n] - m); row[n] = e; sm += e; }
    sm += __shfl_xor(sm, 1, 64);
    sm += __shfl_xor(sm, 2, 64);
    const float inv = 1.f / sm;
    for (int n = q; n < N; n += 4) row[n] *= inv;
    for (int n = N + q; n < 64; n += 4) row[n] = 0.f;   // k padding of the MFMA A operand: exact zeros
  } else if (tid < 64 + 4 * RN) {   // At rows: soft-max over the pixels, 4 threads per sentence; pad columns -> 0
    const int rr = (tid - 64) >> 2, q = tid & 3;
    float* row = Rt + rr * PP;
    float m = -INFINITY;
    for (int p = q; p < P; p += 4) m = fmaxf(m, row[p]);
    m = fmaxf(m, __shfl_xor(m, 1, 64));
    m = fmaxf(m, __shfl_xor(m, 2, 64));
    float sm = 0.f;
    for (int p = q; p < P; p += 4) { const float e = __expf(row[p] - m); row[p] = e; sm += e; }
    sm += __shfl_xor(sm, 1, 64);
    sm += __shfl_xor(sm, 2, 64);
    const float inv = 1.f / sm;
    for (int p = q; p < P; p += 4) row[p] *= inv;
    for (int p = P + q; p < PP; p += 4) row[p] = 0.f;
  }
  __syncthreads();
  // publish the finished rows (exchange stage 2) and save the probabilities the backward pass reads
  for (int e = tid; e < RN * P4; e += 256) {
    const int rr = e / P4;
    *reinterpret_cast<float4*>(Ax + (long)(slot * RN + rr) * PP + (e - rr * P4) * 4) = *reinterpret_cast<const float4*>(Rt + e * 4);
  }
  {
    const int px = tid >> 4, c4 = (tid & 15) * 4;   // 16 pixels x 16 float4
    if (slot * 16 + px < PP)
      *reinterpret_cast<float4*>(Ax + (long)NT * 16 * PP + (long)(slot * 16 + px) * 64 + c4) = *reinterpret_cast<const float4*>(Rv + px * 64 + c4);
  }
  for (int e = tid; e < 16 * N; e += 256) {       // Av plane of the saved probabilities: [P][N]
    const int px = e / N, n = e - px * N;
    if (slot * 16 + px < P) probs[(((long)b * 4 + 0) * P + slot * 16 + px) * N + n] = Rv[px * 64 + n];
  }
  for (int e = tid; e < RN * P; e += 256) {       // AtT plane: [P][N]
    const int rr = e / P, pp = e - rr * P;
    if (slot * RN + rr < N) probs[(((long)b * 4 + 2) * P + pp) * N + slot * RN + rr] = Rt[rr * PP + pp];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&sync[XF_SYNC_FLAGS + (B + b) * XF_SLOTS + slot], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  wait_flags(XF_SYNC_FLAGS + B * XF_SLOTS);
  // all-gather: At [NT*16][PP] -> StF, Av [PP][64] -> SvL [PP][SVS]
  for (int e = tid; e < NT * 16 * P4; e += 256)
    *reinterpret_cast<float4*>(StF + e * 4) = ld4(Ax + e * 4);
  for (int e = tid; e < PP * (SVS / 4); e += 256) {
    const int pp = e / (SVS / 4), c4 = (e - pp * (SVS / 4)) * 4;
    *reinterpret_cast<float4*>(SvL + pp * SVS + c4) = ld4(Ax + (long)NT * 16 * PP + (long)pp * 64 + c4);
  }
  __syncthreads();
  XF_STAMP(4);
  // ---- new_vis[b, :, own channel tiles] = Av . Vt -------------------------------------------------------------------------------------------------
  for (int pt = 0; pt < NPT; ++pt) {
    Split8 sa[KS2];
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
      const int k = ks * 32 + kg * 8;
      const float* a = SvL + (pt * 16 + r16) * SVS + k;
      const bool in = k + 8 <= SVS;
      sa[ks] = split8(in ? *reinterpret_cast<const float4*>(a) : make_float4(0.f, 0.f, 0.f, 0.f),
                      in ? *reinterpret_cast<const float4*>(a + 4) : make_float4(0.f, 0.f, 0.f, 0.f));
    }
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) {
      f32x4v o = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) o = mfma6(sa[ks], vt[ct][ks], o);
      float* dst = new_vis + ((long)b * P + pt * 16 + 4 * kg) * C + slot * CS + (wave * CTW + ct) * 16 + r16;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (pt * 16 + 4 * kg + r < P) dst[(long)r * C] = o[r];
    }
  }
  XF_STAMP(5);
  // ---- new_lan[b, :, slice] = At . Vv[b, :, slice], 32 channels per pass ---------------------------------------------------------------------------------
  Split8 at[4];   // A fragments of sentence tile `wave` (wave < NT): 4 steps of 32 pixels
  if (wave < NT) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int k = ks * 32 + kg * 8;
      const float* a = StF + (wave * 16 + r16) * PP + k;
      const bool in = k + 8 <= PP;
      at[ks] = split8(in ? *reinterpret_cast<const float4*>(a) : make_float4(0.f, 0.f, 0.f, 0.f),
                      in ? *reinterpret_cast<const float4*>(a + 4) : make_float4(0.f, 0.f, 0.f, 0.f));
    }
  }
  const int nks = (P + 31) / 32;   // <= 4
#pragma unroll
  for (int h = 0; h < NPASS; ++h) {
    if (h > 0) __syncthreads();   // the previous pass is done with the planes
    {
      const int prow = tid >> 3, c8 = (tid & 7) * 8;
#pragma unroll
      for (int q = 0; q < VR; ++q) {
        const int p = q * 32 + prow;
        if (p < XF_PK) {
          const Split4 sp = split4(p < P ? vreg[h][q] : make_float4(0.f, 0.f, 0.f, 0.f));
          char* d = planes + p * XF_KS + c8;
          *reinterpret_cast<uint2*>(d) = sp.hi;
          *reinterpret_cast<uint2*>(d + XF_PK * XF_KS) = sp.mid;
          *reinterpret_cast<uint2*>(d + 2 * XF_PK * XF_KS) = sp.lo;
        }
      }
    }
    __syncthreads();
    if (wave < NT) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f32x4v o = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks < nks) {
            Split8 sb;
            sb.hi = tr_frag8c(planes, ks * 32 + kg * 8, ct * 16, lane, XF_PK - 1);
            sb.mid = tr_frag8c(planes + XF_PK * XF_KS, ks * 32 + kg * 8, ct * 16, lane, XF_PK - 1);
            sb.lo = tr_frag8c(planes + 2 * XF_PK * XF_KS, ks * 32 + kg * 8, ct * 16, lane, XF_PK - 1);
            o = mfma6(at[ks], sb, o);
          }
        }
        float* dst = new_lan + ((long)b * N + wave * 16 + 4 * kg) * C + slot * CS + h * 32 + ct * 16 + r16;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (wave * 16 + 4 * kg + r < N) dst[(long)r * C] = o[r];
      }
    }
  }
  XF_STAMP(6);
  // ---- the last workgroup to finish advances the epoch (the next launch on this stream starts after all of them) -----------------------------------
  if (tid == 0) {
    const unsigned t = atomicAdd(&sync[1], 1u);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sync[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

inline long xf_lds_bytes(int P, int NT) {
  (void)P;
  const int SVS = NT <= 3 ? 52 : 64;
  return (long)XF_PP * SVS * 4 + (long)NT * 16 * XF_PP * 4 + 3L * XF_PK * XF_KS + 16;
}

struct XfPlan { long qtf, ktf, vtf, sx, total; int NT, KS2; };
inline XfPlan xf_plan(int B, int N, int C) {
  XfPlan p;
  p.NT = (N + 15) / 16;
  p.KS2 = (p.NT + 1) / 2;
  const long a = (long)p.NT * (C / 32) * 3 * 64 * 16, v = (long)(C / 16) * p.KS2 * 3 * 64 * 16;
  p.qtf = 0; p.ktf = a; p.vtf = 2 * a; p.sx = 2 * a + v;
  p.total = p.sx + (long)B * XF_SLOTS * 2 * p.NT * 16 * XF_PP * 4 + (long)B * (p.NT * 16 * XF_PP + XF_PP * 64) * 4;
#ifdef TRIS_XF_TRACE
  p.total += (long)B * XF_SLOTS * 8 * 8;
#endif
  return p;
}

template <int NT, int KS>
int launch_fused(const float* Qv, const float* Kv, const float* Vv, const float* Qt, const float* Kt, const float* Vt,
                 float* new_vis, float* new_lan, float* probs, int B, int P, int N, int C, char* ws, unsigned* sync,
                 hipStream_t st) {
  const XfPlan pl = xf_plan(B, N, C);
  uint4* QtF = reinterpret_cast<uint4*>(ws + pl.qtf);
  uint4* KtF = reinterpret_cast<uint4*>(ws + pl.ktf);
  uint4* VtF = reinterpret_cast<uint4*>(ws + pl.vtf);
  float* Sx = reinterpret_cast<float*>(ws + pl.sx);
  const int slots = 2 * NT * (C / 32) * 64 + (C / 16) * pl.KS2 * 64;
  hipLaunchKernelGGL(xattn_text_planes_kernel, dim3(cdiv(slots, 256)), dim3(256), 0, st, Qt, Kt, Vt, QtF, KtF, VtF, N, C, NT,
                     pl.KS2);
  const long lds = xf_lds_bytes(P, NT);
  static bool attr_done = false;   // (one instantiation = one function-local flag)
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<NT, KS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds > 65536 ? (int)lds : 65536);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  hipLaunchKernelGGL((xattn_fused_kernel<NT, KS>), dim3(B * XF_SLOTS), dim3(256), (size_t)lds, st, Qv, Kv, Vv, QtF, KtF, VtF,
                     new_vis, new_lan, probs, Sx, sync, B, P, N, 1.0f / sqrtf((float)C));
  TRIS_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" long tris_xattn_fused_ws_bytes(int B, int N, int C) {
  if (B < 1 || N < 1 || N > 64 || C % 512 != 0) return 0;
  return xf_plan(B, N, C).total;
}

extern "C" long tris_xattn_fused_sync_words(int B) { return XF_SYNC_FLAGS + 2L * B * XF_SLOTS; }

extern "C" int tris_xattn_fused_fwd_f32(const float* Qv, const float* Kv, const float* Vv, const float* Qt, const float* Kt,
                                        const float* Vt, float* new_vis, float* new_lan, float* probs, int B, int P, int N,
                                        int C, float* ws, long ws_bytes, unsigned* sync, void* stream) {
  // supported: split-bf16 arithmetic, C = 512 | 1024, 8 <= P <= 104 (<= 13 pixel rows per workgroup x 8), N <= 64
  if (tris_get_gemm_mode() < 1 || !(C == 512 || C == 1024) || P < XF_SLOTS || P > 104 || N < 1 || N > 64 || B < 1 ||
      ws == nullptr || sync == nullptr || ws_bytes < xf_plan(B, N, C).total || xf_lds_bytes(P, (N + 15) / 16) > 160 * 1024)
    return TRIS_WP_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  char* w = reinterpret_cast<char*>(ws);
#define TRIS_XF(NT_)                                                                                                            \
  (C == 1024 ? launch_fused<NT_, 4>(Qv, Kv, Vv, Qt, Kt, Vt, new_vis, new_lan, probs, B, P, N, C, w, sync, st)                   \
             : launch_fused<NT_, 2>(Qv, Kv, Vv, Qt, Kt, Vt, new_vis, new_lan, probs, B, P, N, C, w, sync, st))
  switch ((N + 15) / 16) {
    case 1: return TRIS_XF(1);
    case 2: return TRIS_XF(2);
    case 3: return TRIS_XF(3);
    default: return TRIS_XF(4);
  }
#undef TRIS_XF
}
