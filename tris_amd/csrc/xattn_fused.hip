// Bilateral image<->text cross attention (reference model/attn.py:117-128) as ONE persistent launch (gfx950, split-bf16 x3).
//
//   Av  = softmax_n(Qv Kt^T / sqrt(C))   [B,P,N]      new_vis = Av  Vt      [B,P,C]
//   AtT = softmax_p(Kv Qt^T / sqrt(C))   [B,P,N]      new_lan = AtT^T Vv    [B,N,C]
//
// Why one launch and how the work is cut.  The pair is an HBM stream (1.84 MB per image at P = 100, N = 48, C = 1024, 21 FLOP/B)
// and a CU moves ~25 GB/s of it, so an image has to be spread over several CUs in EVERY phase -- but both soft-maxes need the
// full reduction over C, and the pixel soft-max couples all pixels of an image.  Eight workgroups per image (B * 8 = 384 at
// B = 48), ALL of them sliced by CHANNEL (workgroup s owns channels [s C/8, (s+1) C/8)), so that every operand byte is read by
// exactly one workgroup and the sentence operands it needs are a 1/8 slice as well:
//   logits    partial D_t[p][n], D_v[p][n] of ALL pixels of the image over the own slice.  Pixel rows are read row-contiguous
//             (8 rows x 128 B per instruction) and turned into MFMA fragments through wave-private LDS tiles; the sentence
//             operands arrive PRE-SPLIT into three bf16 planes in MFMA fragment order (one coalesced 16-byte load per lane per
//             fragment, no VALU, no LDS), written once per call by xattn_text_planes_kernel (a 4 us launch in front).
//   exchange  reduce-scatter + all-gather instead of "everyone reads everything" (which cost 132 MB of reads, more than the
//             kernel's algorithmic bytes): stage 1 -- each workgroup publishes its partial blocks (43 KB), then sums, for 1/8 of
//             the sentence rows of S_t and for one 16-pixel tile of S_v, the eight partials (46 KB), takes the soft-max of those
//             complete rows and publishes the PROBABILITIES (5 KB); stage 2 -- everyone reads the image's At and Av (50 KB).
//   outputs   new_vis[:, slice] = Av . Vt[:, slice] (Vt^T fragments pre-split like the other sentence operands) and
//             new_lan[:, slice] = At . Vv[:, slice]: the Vv slice (requested before the exchange) is split once, staged k-major
//             in two LDS buffers and gathered by the transpose read ds_read_b64_tr_b16.  In all four products the CHANNELS are
//             the MFMA rows, so a lane holds four consecutive channels (pixels) of one output row: every store is 16 bytes.
// HBM traffic = algorithmic (88.7 MB) + saved probabilities (1.8 MB) + exchange (16.5 + 2.4 MB written, 18 + 19 MB read).
//
// Inter-workgroup protocol (MI355X_MICROARCH "workgroup dispatch / visibility", recipe R1): payloads are WRITE-THROUGH (sc1)
// 16-byte stores, every storing wave drains vmcnt, barrier, ONE lane raises the (image, slot) flag with a relaxed agent-scope
// store; consumers poll the eight flags relaxed from one wave, barrier, and read the payload with sc1 loads (no L1) -- no
// release / acquire fence anywhere (the first build used plain stores + fences: each release wrote back the XCD's dirty L2,
// 8 + 29 us of a 75 us launch).  Placement-independent.  The epoch that tags the flags lives in device memory (sync[0]) and is
// advanced by the last workgroup to finish, so a captured launch replays correctly; spins are bounded (sync[2] != 0 afterwards:
// a peer never published, the outputs of that launch are undefined).  All B * 8 workgroups must be co-resident: the entry point
// declines (TRIS_DECLINED) when B * 8 exceeds CUs x workgroups per CU.
//
// Measured at B = 48 (tools/xattn_fused_trace.py, s_memtime stamps, median workgroup, us): logits 9.3 | publish 1.7 | wait 2.4 |
// reduce 2.3 | soft-max 1.1 | publish 1.9 | wait 0.7 | gather 1.7 | new_vis 3.4 | new_lan 5.2 = 31 us per workgroup, 42 us per
// call with the preparation launch (two-launch pair: 57 us).  What bounds it: 384 workgroups on 256 CUs put two workgroups on
// half the CUs, and a CU fetches ~10 B/clk from HBM: the doubly loaded CUs need 2 x 102 KB / 24 GB/s = 8.5 us for the logits
// phase alone, ~19 us for their 470 KB in total; with 48 images there is no second image per workgroup to overlap the three
// dependent phases with.  0.60 of 8 TB/s (18.5 us) is not reachable at this batch size; see DESIGN.md section 3.
#include "common.h"
#include "tris_hip.h"
#include "x3_split.h"
#include "xattn_planes.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int XF_SLOTS = 8;       // workgroups per image
constexpr int XF_PP = 112;        // padded pixel count: 7 MFMA tiles of 16 (P <= 104 leaves a zero row in the planes)
constexpr int XF_PK = 104;        // pixel rows of a k-major Vv plane; rows >= P are zero, reads beyond are clamped to row XF_PK - 1
constexpr int XF_KS = 80;         // bytes per pixel row of a k-major Vv plane (32 channels x 2 B + 16): two plane buffers fit the LDS budget
constexpr int XF_SYNC_FLAGS = 16; // sync[0] epoch, [1] finish ticket, [2] time-out flag, [16 + (stage*B + b)*8 + s] publish flags of the two exchange stages
constexpr long XF_SPIN = 4000000; // polls before a wait gives up (~seconds)

// developer build (-DTRIS_XF_TRACE, tools/xattn_fused_trace.py): per-workgroup s_memtime stamps at the phase boundaries, written
// behind the hand-off scratch; compiled out of the product
#ifdef TRIS_XF_TRACE
#define XF_STAMP(i) do { if (tid == 0) xf_trace[(long)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define XF_STAMP(i) do { } while (0)
#endif

constexpr int TLD = 36;   // row stride (floats) of a wave-private 16 x 32 turn-around tile (conflict-free b128 fragment reads)
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// Exchange payloads travel WRITE-THROUGH (sc1 stores) and are read with sc1 loads (L1 bypass): no release / acquire fence at all.
// A release fence writes back every dirty line of the XCD's L2 (6.5 us with this kernel's stores in flight, measured: the two
// fenced hand-offs of the first build cost 8 + 29 us of the 75 us launch); MI355X_MICROARCH "publish-large": 8.2 vs 3.0 us.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_wt(__amdgpu_buffer_rsrc_t rs, long float_off, float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)(float_off * 4), 0, 16);
}
__device__ __forceinline__ float4 ld4_wt(__amdgpu_buffer_rsrc_t rs, long float_off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(float_off * 4), 0, 16));
}
__device__ __forceinline__ bf16x8 ldf(const uint4* p) { return __builtin_bit_cast(bf16x8, *p); }

// N independent accumulation chains issued piece by piece (chain 0, 1, .., N-1, then the next piece): consecutive MFMAs never
// depend on each other (a dependent v_mfma waits for its predecessor's result: a serial chain of 6 runs at less than half rate); smallest terms first
template <int NCH>
__device__ __forceinline__ void mfma6_n(const Split8 (&a)[NCH], const Split8 (&b)[NCH], f32x4v (&c)[NCH]) {
#define XF_PIECE(PA, PB)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < NCH; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].PA, b[i].PB, c[i], 0, 0, 0);
  XF_PIECE(lo, hi) XF_PIECE(hi, lo) XF_PIECE(mid, mid) XF_PIECE(mid, hi) XF_PIECE(hi, mid) XF_PIECE(hi, hi)
#undef XF_PIECE
}

// grid B * 8, block 256, dynamic LDS xf_lds_bytes(P, NT).  KS = 32-channel steps of a workgroup's channel slice (C = 256 KS).
template <int NT, int KS>
__global__ __launch_bounds__(256, 2) void xattn_fused_kernel(const float* __restrict__ Qv, const float* __restrict__ Kv,
                                                             const float* __restrict__ Vv, const uint4* __restrict__ QtF,
                                                             const uint4* __restrict__ KtF, const uint4* __restrict__ VtF,
                                                             float* __restrict__ new_vis, float* __restrict__ new_lan,
                                                             float* __restrict__ probs, float* __restrict__ Sx, int sx_bytes,
                                                             unsigned* __restrict__ sync, int B, int P, int N, float scale) {
  constexpr int CS = KS * 32;            // channels owned by a workgroup
  constexpr int C = CS * XF_SLOTS;
  constexpr int KST = C / 32;            // 32-channel steps over all of C
  constexpr int KS2 = (NT + 1) / 2;      // 32-sentence steps of the new_vis product
  constexpr int SVS = NT <= 3 ? 56 : 64; // row stride (floats) of the Av plane [p][n] in LDS (>= 3 XF_PK XF_KS bytes in all: second plane buffer)
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
  const __amdgpu_buffer_rsrc_t sxr = __builtin_amdgcn_make_buffer_rsrc(Sx, 0, sx_bytes, 0x00020000);   // the exchange scratch (Sx | Ax)
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
        Split8 ca[2 * NT], cb[2 * NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) { ca[j] = sk; cb[j] = fq_[j]; ca[NT + j] = sq; cb[NT + j] = fk_[j]; }
        mfma6_n<2 * NT>(ca, cb, reinterpret_cast<f32x4v(&)[2 * NT]>(acc[0]));   // 2 NT independent chains
      }
      {   // (a wave without a second tile computes on its clamped rows and never publishes the result)
        Split8 sk, sq;
        turn(wk, wq, sk, sq);
        if (ks + 1 < KS) {
#pragma unroll
          for (int h = 0; h < 2; ++h) { wk[h] = ld4(Kv + g1[h] + (ks + 1) * 32); wq[h] = ld4(Qv + g1[h] + (ks + 1) * 32); }
        }
        Split8 ca[2 * NT], cb[2 * NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) { ca[j] = sk; cb[j] = fq_[j]; ca[NT + j] = sq; cb[NT + j] = fk_[j]; }
        mfma6_n<2 * NT>(ca, cb, reinterpret_cast<f32x4v(&)[2 * NT]>(acc[1]));
      }
    }
    XF_STAMP(1);
    // publish the partial blocks: Sx[b][slot][type][n][p]   (D[p = 16 tile + 4 kg + r][n = 16 j + r16]: 4 consecutive p per lane)
    const long mine = ((long)b * XF_SLOTS + slot) * (2 * NT * 16 * PP);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t == 0 || two) {
        const int pc = (wave + 4 * t) * 16 + 4 * kg;
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            st4_wt(sxr, mine + (long)(y * NT * 16 + j * 16 + r16) * PP + pc,
                   make_float4(acc[t][y][j][0], acc[t][y][j][1], acc[t][y][j][2], acc[t][y][j][3]));
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave drains its write-through stores ...
  __syncthreads();
  const unsigned epoch = *s_epoch_p;
  if (tid == 0)                                       // ... then ONE lane raises the flag (no fence: nothing is left in a cache)
    __hip_atomic_store(&sync[XF_SYNC_FLAGS + b * XF_SLOTS + slot], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    }
    __syncthreads();   // (no acquire fence: the payload is read with sc1 loads, which do not look at this CU's L1)
  };
  wait_flags(XF_SYNC_FLAGS);
  XF_STAMP(3);
  constexpr int RN = NT * 2;             // sentence rows finished by a workgroup (NT * 16 / 8)
  constexpr int P4 = PP / 4;
  float* Rt = reinterpret_cast<float*>(lds);            // [RN][PP]   logits -> At rows
  float* Rv = Rt + RN * PP;                             // [16][64]   logits -> Av rows of pixel tile `slot` ([p][n])
  const long Ax = (long)B * XF_SLOTS * (2 * NT * 16 * PP) + (long)b * (NT * 16 * PP + PP * 64);   // image b: At [NT*16][PP] | Av [PP][64]
  {
    const long base = (long)b * XF_SLOTS * (2 * NT * 16 * PP);
    constexpr int NA = RN * P4, NB = NT * 16 * 4;
    for (int e = tid; e < NA + NB; e += 256) {
      long off;
      if (e < NA) { const int rr = e / P4; off = (long)(slot * RN + rr) * PP + (e - rr * P4) * 4; }
      else { const int n = (e - NA) >> 2; off = (long)(NT * 16 + n) * PP + slot * 16 + ((e - NA) & 3) * 4; }
      float4 v[XF_SLOTS];
#pragma unroll
      for (int s2 = 0; s2 < XF_SLOTS; ++s2) v[s2] = ld4_wt(sxr, base + (long)s2 * (2 * NT * 16 * PP) + off);
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
  XF_STAMP(10);
  if (tid < 64) {   // Av rows of the 16 pixels of tile `slot`: soft-max over the sentences, 4 threads per pixel, values in registers
    const int px = tid >> 2, q = tid & 3;
    float* row = Rv + px * 64;
    float x[16];
    float m = -INFINITY;
#pragma unroll
    for (int u = 0; u < 16; ++u) { x[u] = (q + 4 * u < N) ? row[q + 4 * u] : -INFINITY; m = fmaxf(m, x[u]); }
    m = fmaxf(m, __shfl_xor(m, 1, 64));
    m = fmaxf(m, __shfl_xor(m, 2, 64));
    float sm = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) { x[u] = (q + 4 * u < N) ? __expf(x[u] - m) : 0.f; sm += x[u]; }
    sm += __shfl_xor(sm, 1, 64);
    sm += __shfl_xor(sm, 2, 64);
    const float inv = 1.f / sm;
#pragma unroll
    for (int u = 0; u < 16; ++u) row[q + 4 * u] = x[u] * inv;   // (columns >= N: exact zeros = k padding of the MFMA operand)
  } else {   // At rows: soft-max over the pixels, 32 lanes per sentence row (<= 4 pixels per lane, all in registers); pad columns -> 0
    const int q = lane & 31;
    for (int rr = (tid - 64) >> 5; rr < RN; rr += 6) {
      float* row = Rt + rr * PP;
      float x[4];
      float m = -INFINITY;
#pragma unroll
      for (int u = 0; u < 4; ++u) { x[u] = (q + 32 * u < P) ? row[q + 32 * u] : -INFINITY; m = fmaxf(m, x[u]); }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      float sm = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) { x[u] = (q + 32 * u < P) ? __expf(x[u] - m) : 0.f; sm += x[u]; }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
      const float inv = 1.f / sm;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (q + 32 * u < PP) row[q + 32 * u] = x[u] * inv;   // (columns >= P: exact zeros)
    }
  }
  __syncthreads();
  XF_STAMP(11);
  // publish the finished rows (exchange stage 2) and save the probabilities the backward pass reads
  for (int e = tid; e < RN * P4; e += 256) {
    const int rr = e / P4;
    st4_wt(sxr, Ax + (long)(slot * RN + rr) * PP + (e - rr * P4) * 4, *reinterpret_cast<const float4*>(Rt + e * 4));
  }
  {
    const int px = tid >> 4, c4 = (tid & 15) * 4;   // 16 pixels x 16 float4
    if (slot * 16 + px < PP)
      st4_wt(sxr, Ax + (long)NT * 16 * PP + (long)(slot * 16 + px) * 64 + c4, *reinterpret_cast<const float4*>(Rv + px * 64 + c4));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0)
    __hip_atomic_store(&sync[XF_SYNC_FLAGS + (B + b) * XF_SLOTS + slot], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (not part of the exchange, so behind the flag: the probabilities the backward pass reads)
  for (int e = tid; e < 16 * N; e += 256) {       // Av plane: [P][N]
    const int px = e / N, n = e - px * N;
    if (slot * 16 + px < P) probs[(((long)b * 4 + 0) * P + slot * 16 + px) * N + n] = Rv[px * 64 + n];
  }
  for (int e = tid; e < RN * P; e += 256) {       // AtT plane: [P][N]
    const int rr = e / P, pp = e - rr * P;
    if (slot * RN + rr < N) probs[(((long)b * 4 + 2) * P + pp) * N + slot * RN + rr] = Rt[rr * PP + pp];
  }
  XF_STAMP(12);
  wait_flags(XF_SYNC_FLAGS + B * XF_SLOTS);
  XF_STAMP(13);
  // all-gather: At [NT*16][PP] -> StF, Av [PP][64] -> SvL [PP][SVS]   (all loads of a thread in flight before the first LDS store;
  // the stage-1 rows Rt / Rv alias SvL, but every thread passed the barrier inside wait_flags after its last read of them)
  {
    constexpr int NA = NT * 16 * P4, NV = PP * (SVS / 4), UA = (NA + 255) / 256, UV = (NV + 255) / 256;
    float4 ga[UA], gv[UV];
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const int e = tid + u * 256;
      if (e < NA) ga[u] = ld4_wt(sxr, Ax + e * 4);
    }
#pragma unroll
    for (int u = 0; u < UV; ++u) {
      const int e = tid + u * 256;
      const int pp = e / (SVS / 4), c4 = (e - pp * (SVS / 4)) * 4;
      if (e < NV) gv[u] = ld4_wt(sxr, Ax + (long)NT * 16 * PP + (long)pp * 64 + c4);
    }
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const int e = tid + u * 256;
      if (e < NA) *reinterpret_cast<float4*>(StF + e * 4) = ga[u];
    }
#pragma unroll
    for (int u = 0; u < UV; ++u) {
      const int e = tid + u * 256;
      const int pp = e / (SVS / 4), c4 = (e - pp * (SVS / 4)) * 4;
      if (e < NV) *reinterpret_cast<float4*>(SvL + pp * SVS + c4) = gv[u];
    }
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
    {
      constexpr int NCH = CTW * KS2;   // independent chains: (channel tile, 32-sentence step)
      Split8 ca[NCH], cb[NCH];
      f32x4v co[NCH];
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
          ca[ct * KS2 + ks] = vt[ct][ks];
          cb[ct * KS2 + ks] = sa[ks];
          co[ct * KS2 + ks] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        }
      mfma6_n<NCH>(ca, cb, co);        // channels as rows: D[c = 4 kg + r][p = r16]
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) {
        f32x4v o = co[ct * KS2];
#pragma unroll
        for (int ks = 1; ks < KS2; ++ks) o += co[ct * KS2 + ks];
        if (pt * 16 + r16 < P)   // 4 consecutive channels of one pixel per lane: one 16-byte store
          *reinterpret_cast<float4*>(new_vis + ((long)b * P + pt * 16 + r16) * C + slot * CS + (wave * CTW + ct) * 16 + 4 * kg) =
              make_float4(o[0], o[1], o[2], o[3]);
      }
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
  // two plane buffers: the planes region and the (now dead) Av region -- pass h + 1 is split and stored while pass h computes,
  // one barrier per pass
  auto stage = [&](char* dstp, const float4 (&v)[VR]) {
    const int prow = tid >> 3, c8 = (tid & 7) * 8;
#pragma unroll
    for (int q = 0; q < VR; ++q) {
      const int p = q * 32 + prow;
      if (p < XF_PK) {
        const Split4 sp = split4(p < P ? v[q] : make_float4(0.f, 0.f, 0.f, 0.f));
        char* d = dstp + p * XF_KS + c8;
        *reinterpret_cast<uint2*>(d) = sp.hi;
        *reinterpret_cast<uint2*>(d + XF_PK * XF_KS) = sp.mid;
        *reinterpret_cast<uint2*>(d + 2 * XF_PK * XF_KS) = sp.lo;
      }
    }
  };
  // byte offsets of this lane's two transpose reads of 32-pixel step ks inside a plane (rows clamped to the last plane row)
  int troff[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int k0 = ks * 32 + kg * 8 + (r16 >> 2);
    troff[ks][0] = min(k0, XF_PK - 1) * XF_KS + 8 * (r16 & 3);
    troff[ks][1] = min(k0 + 4, XF_PK - 1) * XF_KS + 8 * (r16 & 3);
  }
  auto trf = [&](const char* pl, int ks, int ct) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pl + troff[ks][0] + ct * 32));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pl + troff[ks][1] + ct * 32));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  char* const buf0 = planes;
  char* const buf1 = lds;   // (SvL: every wave is past new_vis once it has passed the barrier below)
  stage(buf0, vreg[0]);
  __syncthreads();
#pragma unroll
  for (int h = 0; h < NPASS; ++h) {
    const char* cur = (h & 1) ? buf1 : buf0;
    if (h + 1 < NPASS) stage((h & 1) ? buf0 : buf1, vreg[h + 1 < NPASS ? h + 1 : 0]);
    if (wave < NT) {
      f32x4v co[4] = {(f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kp = 0; kp < 2; ++kp) {   // two 32-pixel steps at a time: chains (ct 0 | 1) x (step 2 kp | 2 kp + 1)
        if (2 * kp < nks) {
          Split8 ca[4], cb[4];
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int ks = 2 * kp + u;   // (a step beyond nks multiplies clamped plane rows by the zero columns of At)
              ca[ct * 2 + u].hi = trf(cur, ks, ct);
              ca[ct * 2 + u].mid = trf(cur + XF_PK * XF_KS, ks, ct);
              ca[ct * 2 + u].lo = trf(cur + 2 * XF_PK * XF_KS, ks, ct);
              cb[ct * 2 + u] = at[ks];
            }
          mfma6_n<4>(ca, cb, co);          // channels as rows: D[c = 4 kg + r][n = r16]
        }
      }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const f32x4v o = co[ct * 2] + co[ct * 2 + 1];
        if (wave * 16 + r16 < N)
          *reinterpret_cast<float4*>(new_lan + ((long)b * N + wave * 16 + r16) * C + slot * CS + h * 32 + ct * 16 + 4 * kg) =
              make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    if (h + 1 < NPASS) __syncthreads();
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
  const int SVS = NT <= 3 ? 56 : 64;
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
  p.total += (long)B * XF_SLOTS * 16 * 8;
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
  hipLaunchKernelGGL(xattn_text_planes_kernel<false>, dim3(cdiv(slots, 256)), dim3(256), 0, st, Qt, Kt, Vt, QtF, KtF, VtF, N, C, NT,
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
                     new_vis, new_lan, probs, Sx, (int)(pl.total - pl.sx), sync, B, P, N, 1.0f / sqrtf((float)C));
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
    return TRIS_DECLINED;
  {   // every workgroup must be resident at once (they wait for each other)
    static int cus = 0;
    if (cus == 0) {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        n = 1;
      cus = n;
    }
    const long per_cu = (160L * 1024) / xf_lds_bytes(P, (N + 15) / 16);
    if ((long)B * XF_SLOTS > (long)cus * (per_cu > 2 ? 2 : per_cu)) return TRIS_DECLINED;
  }
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
