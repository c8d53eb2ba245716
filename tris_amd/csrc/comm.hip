// SyncBatchNorm statistics exchange over xGMI peer memory (gfx950, one process per GPU of ONE node).
//
// The data-parallel step has two exchange steps (reference: nn.SyncBatchNorm + DistributedDataParallel,
// train_stage1.py:69-70).  The gradient mean is a handful of large RCCL all-reduces that overlap backward.  The SyncBatchNorm
// statistics are the opposite: 55 all-gathers of [mean | invstd | var] in forward and 55 all-reduces of two sums in backward,
// each 1-24 KB, each ON the critical path.  As RCCL calls they cost ~32 us (torch.distributed) to ~55 us (a direct
// ncclAllGather) of host enqueue work apiece -- +3.5 ... +7 ms per step at ONE rank, before any wire time.
//
// Here every rank owns a MAILBOX in uncached device memory that its peers map through HIP IPC.  One exchange is ONE small
// kernel launch on the compute stream: each rank stores its block straight into every peer's mailbox over xGMI (16-byte
// stores), publishes a per-sender flag (system-scope release), then waits for the flags of its own mailbox (system-scope
// acquire, bounded spin) and gathers / sums what arrived.  No RCCL, no second stream, no host round trip.
//
//   mailbox (per rank, per parity b = seq & 1):   slot[b][sender][cap floats]   flag[b][sender] (= seq of the block it holds)
//   Two parities suffice: a rank can start exchange k+2 (which overwrites parity k) only after it has seen every peer's
//   flag k+1, and a peer posts flag k+1 only after its own exchange k has completed (stream order).
//
// The exchange counter `seq` lives in DEVICE memory (one word per mailbox owner, advanced by the kernel itself): nothing about an
// exchange depends on host state, so the launches can be captured into a hipGraph and replayed -- every rank runs the same
// sequence of exchanges on one stream, so the counters advance in lockstep.
//
// A flag that does not arrive within the spin bound sets *err (checked by the host on EVERY rank, tris_amd.comm.check_errors)
// and the kernel does NOT consume the stale slots: its outputs are filled with NaN and the running statistics are left
// untouched, so the losses of that step are NaN on the rank that gave up -- the step fails loudly instead of hanging the GPU or
// training on another layer's statistics.
#include <cstring>

#include "common.h"
#include "tris_hip.h"
#include "x3_split.h"

namespace {

struct MboxHeader {
  unsigned flag[2][TRIS_MBOX_MAX_WORLD];
};
constexpr long HDR_FLOATS = 64;  // 256 bytes reserved in front of the slots

__device__ __forceinline__ float* slot_of(void* box, int parity, int sender, int cap) {
  return reinterpret_cast<float*>(box) + HDR_FLOATS + ((long)parity * TRIS_MBOX_MAX_WORLD + sender) * cap;
}

// send my block [src0[n0] | src1[n1]] to every mailbox, publish, wait for the world's flags in my mailbox (one workgroup)
// returns false (uniformly) when a sender's flag did not arrive within the spin bound
// *seq_out = the number of this exchange (read from / advanced in *seq_dev: launches on one stream run in order)
__device__ __forceinline__ bool mbox_send_wait(const float* src0, int n0, const float* src1, int n1,
                                               void* const* __restrict__ boxes, int world, int rank, unsigned* seq_dev, int cap,
                                               long spin_limit, int* __restrict__ err, unsigned* seq_out) {
  __shared__ int s_fail;
  __shared__ unsigned s_seq;
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_fail = 0;
    s_seq = *seq_dev + 1u;
    *seq_dev = s_seq;
  }
  __syncthreads();
  const unsigned seq = s_seq;
  *seq_out = seq;
  const int par = seq & 1u;
  for (int w = 0; w < world; ++w) {
    float* dst = slot_of(boxes[w], par, rank, cap);
    if (((n0 | n1) & 3) == 0) {
      for (int i = tid; i < (n0 >> 2); i += 256) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src0)[i];
      for (int i = tid; i < (n1 >> 2); i += 256) reinterpret_cast<float4*>(dst + n0)[i] = reinterpret_cast<const float4*>(src1)[i];
    } else {
      for (int i = tid; i < n0; i += 256) dst[i] = src0[i];
      for (int i = tid; i < n1; i += 256) dst[n0 + i] = src1[i];
    }
  }
  __threadfence_system();
  __syncthreads();
  if (tid < world) {  // publish: one flag per receiver
    MboxHeader* h = reinterpret_cast<MboxHeader*>(boxes[tid]);
    __hip_atomic_store(&h->flag[par][rank], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (tid < world) {  // wait for every sender's flag in MY mailbox
    MboxHeader* h = reinterpret_cast<MboxHeader*>(boxes[rank]);
    long spins = 0;
    while (__hip_atomic_load(&h->flag[par][tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > spin_limit) { atomicExch(err, (int)seq); s_fail = 1; break; }
    }
  }
  __syncthreads();
  __threadfence_system();
  return s_fail == 0;
}

// mode 0: out[w][n] = gathered blocks, mode 1: out[n] = sum over ranks (fixed order 0..world-1: deterministic and identical
// on every rank); the block of a rank is [src0[n0] | src1[n1]]
// (src0 / out are NOT restrict: the in-place sum of tris_amd.comm.syncbn_all_reduce_sum passes the same buffer for both;
// every source element is stored to the mailboxes before the first element of `out` is written)
// Optional by-product of an exchange launch inside an h2 step with operand planes (csrc/planes.h): the amax WORD that bounds the
// plane tensor the following pass writes -- what tris_bn_out_bound2_f32 (forward: |gamma| xhat_max + |beta| [+ the residual's bound])
// and tris_bn_bwd_bound_f32 (backward: |gamma invstd| (amax dz + |sum dz| / n + xhat_max |sum dz xhat| / n)) compute in launches of
// their own.  The statistics are on the chip at the end of the exchange anyway: two launches less per SyncBatchNorm layer and pass.
struct MboxBound {
  const float* gamma;       // NULL: no bound wanted
  const float* other;       // forward: beta; backward: invstd
  const unsigned* word_in;  // forward: the residual's amax word or NULL; backward: the amax word of dz
  unsigned* word_out;
  float xhat_max, inv_cnt;
};
__device__ __forceinline__ void mbox_bound_commit(float m, const MboxBound& b, bool fwd) {   // every thread of the 256
  __shared__ float red[4];
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x < 64) {
    float v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (fwd && b.word_in != nullptr) v += __builtin_bit_cast(float, h2_amax_of(b.word_in, threadIdx.x));
    if (threadIdx.x == 0) b.word_out[0] = __builtin_bit_cast(unsigned, v);
  }
}

__global__ __launch_bounds__(256) void mbox_exchange_kernel(const float* src0, int n0, const float* src1,
                                                            int n1, float* out, void* const* __restrict__ boxes,
                                                            int world, int rank, unsigned* seq_dev, int cap, int mode, long spin_limit,
                                                            int* __restrict__ err, MboxBound bnd) {
  unsigned seq;
  const bool ok = mbox_send_wait(src0, n0, src1, n1, boxes, world, rank, seq_dev, cap, spin_limit, err, &seq);
  const int tid = threadIdx.x, par = seq & 1u, n = n0 + n1;
  if (!ok) {   // a peer never posted: do not read the stale slots
    const int tot = mode == 0 ? world * n : n;
    for (int i = tid; i < tot; i += 256) out[i] = __builtin_nanf("");
    return;
  }
  if (mode == 0) {
    for (int w = 0; w < world; ++w) {
      const float* s = slot_of(boxes[rank], par, w, cap);
      for (int i = tid; i < n; i += 256) out[(long)w * n + i] = __builtin_nontemporal_load(s + i);
    }
  } else {
    for (int i = tid; i < n; i += 256) {
      float acc = 0.f;
      for (int w = 0; w < world; ++w) acc += __builtin_nontemporal_load(slot_of(boxes[rank], par, w, cap) + i);
      out[i] = acc;
    }
    if (bnd.gamma != nullptr) {   // SyncBatchNorm backward, out = [sum dz | sum dz xhat] over all ranks: the bound of dx (n0 = n1 = C)
      __syncthreads();
      const float adz = __builtin_bit_cast(float, h2_amax_of(bnd.word_in, tid & 63));
      float m = 0.f;
      for (int c = tid; c < n0; c += 256)
        m = fmaxf(m, fabsf(bnd.gamma[c] * bnd.other[c]) * (adz + fabsf(out[c]) * bnd.inv_cnt + bnd.xhat_max * fabsf(out[n0 + c]) * bnd.inv_cnt));
      mbox_bound_commit(m, bnd, false);
    }
  }
}

// SyncBatchNorm forward in one launch: exchange the per-rank [mean | invstd | biased var] blocks (3 C floats, exactly what
// bn_finalize_kernel writes) and combine them -> global mean / invstd / var (+ running statistics), the arithmetic of
// bn_sync_combine_kernel (norm.hip).  Every rank contributes `count` rows.
__global__ __launch_bounds__(256) void mbox_bn_combine_kernel(const float* __restrict__ local, int C, float count, float eps,
                                                              float momentum, float* __restrict__ stats, float* running_mean,
                                                              float* running_var, void* const* __restrict__ boxes, int world,
                                                              int rank, unsigned* seq_dev, int cap, long spin_limit, int* __restrict__ err,
                                                              MboxBound bnd) {
  unsigned seq;
  const bool ok = mbox_send_wait(local, 3 * C, nullptr, 0, boxes, world, rank, seq_dev, cap, spin_limit, err, &seq);
  const int par = seq & 1u;
  if (bnd.gamma != nullptr) {   // the bound of the plane output: from the affine parameters alone (Samuelson), whatever arrived
    float m = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, fabsf(bnd.gamma[c]) * bnd.xhat_max + fabsf(bnd.other[c]));
    mbox_bound_commit(m, bnd, true);
  }
  if (!ok) {   // a peer never posted: NaN statistics (the step's losses turn NaN), running statistics untouched
    for (int c = threadIdx.x; c < 3 * C; c += 256) stats[c] = __builtin_nanf("");
    return;
  }
  for (int c = threadIdx.x; c < C; c += 256) {
    double mean = 0.0;
    for (int w = 0; w < world; ++w) mean += (double)__builtin_nontemporal_load(slot_of(boxes[rank], par, w, cap) + c);
    mean /= (double)world;
    double m2 = 0.0;
    for (int w = 0; w < world; ++w) {
      const float* g = slot_of(boxes[rank], par, w, cap);
      const double d = (double)__builtin_nontemporal_load(g + c) - mean;
      m2 += (double)__builtin_nontemporal_load(g + 2 * C + c) + d * d;
    }
    const double var = m2 / (double)world;
    const double n = (double)count * world;
    stats[c] = (float)mean;
    stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    stats[2 * C + c] = (float)var;
    if (running_mean) {
      const double unb = n > 1.0 ? var * (n / (n - 1.0)) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  }
}

}  // namespace

extern "C" long tris_mbox_bytes(int cap_floats) {
  return (long)sizeof(float) * (HDR_FLOATS + 2L * TRIS_MBOX_MAX_WORLD * cap_floats);
}

extern "C" int tris_mbox_alloc(void** ptr, int cap_floats) {
  if (cap_floats <= 0 || (cap_floats & 3)) return (int)hipErrorInvalidValue;
  const long bytes = tris_mbox_bytes(cap_floats);
  hipError_t e = hipExtMallocWithFlags(ptr, (size_t)bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) return (int)e;
  return (int)hipMemset(*ptr, 0, (size_t)bytes);
}

extern "C" int tris_mbox_free(void* ptr) { return (int)hipFree(ptr); }

extern "C" int tris_mbox_ipc_handle(void* ptr, void* handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  return (int)hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), ptr);
}

extern "C" int tris_mbox_ipc_open(const void* handle64, void** ptr) {
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  return (int)hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
}

extern "C" int tris_mbox_ipc_close(void* ptr) { return (int)hipIpcCloseMemHandle(ptr); }

extern "C" int tris_mbox_exchange_f32(const float* src0, int n0, const float* src1, int n1, float* out, void* const* boxes,
                                      int world, int rank, unsigned* seq, int cap_floats, int mode, long spin_limit, int* err,
                                      void* stream) {
  if (n0 <= 0 || n1 < 0 || n0 + n1 > cap_floats || world < 1 || world > TRIS_MBOX_MAX_WORLD || rank < 0 || rank >= world ||
      seq == nullptr)
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mbox_exchange_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, src0, n0, src1, n1, out, boxes, world, rank,
                     seq, cap_floats, mode, spin_limit, err, MboxBound{nullptr, nullptr, nullptr, nullptr, 0.f, 0.f});
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_mbox_bn_bwd_exchange_f32(const float* sum_dz, const float* sum_dzx, int C, float* out, void* const* boxes, int world,
                                             int rank, unsigned* seq, int cap_floats, long spin_limit, int* err, const float* gamma,
                                             const float* invstd, float inv_count, float xhat_max, const unsigned* dz_word,
                                             unsigned* bound_word, void* stream) {
  if (C <= 0 || 2 * C > cap_floats || world < 1 || world > TRIS_MBOX_MAX_WORLD || rank < 0 || rank >= world || seq == nullptr ||
      gamma == nullptr || invstd == nullptr || dz_word == nullptr || bound_word == nullptr)
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mbox_exchange_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sum_dz, C, sum_dzx, C, out, boxes, world, rank,
                     seq, cap_floats, 1, spin_limit, err, MboxBound{gamma, invstd, dz_word, bound_word, xhat_max, inv_count});
  TRIS_LAUNCH_CHECK();
  return 0;
}

extern "C" int tris_mbox_bn_combine_f32(const float* local_stats, int C, long count_per_rank, float eps, float momentum,
                                        float* stats, float* running_mean, float* running_var, void* const* boxes, int world,
                                        int rank, unsigned* seq, int cap_floats, long spin_limit, int* err, void* stream) {
  if (C <= 0 || 3 * C > cap_floats || world < 1 || world > TRIS_MBOX_MAX_WORLD || rank < 0 || rank >= world || seq == nullptr)
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mbox_bn_combine_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, local_stats, C, (float)count_per_rank, eps,
                     momentum, stats, running_mean, running_var, boxes, world, rank, seq, cap_floats, spin_limit, err,
                     MboxBound{nullptr, nullptr, nullptr, nullptr, 0.f, 0.f});
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_mbox_bn_combine_bound_f32(const float* local_stats, int C, long count_per_rank, float eps, float momentum,
                                              float* stats, float* running_mean, float* running_var, void* const* boxes, int world,
                                              int rank, unsigned* seq, int cap_floats, long spin_limit, int* err, const float* gamma,
                                              const float* beta, float xhat_max, const unsigned* resid_word, unsigned* bound_word,
                                              void* stream) {
  if (C <= 0 || 3 * C > cap_floats || world < 1 || world > TRIS_MBOX_MAX_WORLD || rank < 0 || rank >= world || seq == nullptr ||
      gamma == nullptr || beta == nullptr || bound_word == nullptr)
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mbox_bn_combine_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, local_stats, C, (float)count_per_rank, eps,
                     momentum, stats, running_mean, running_var, boxes, world, rank, seq, cap_floats, spin_limit, err,
                     MboxBound{gamma, beta, resid_word, bound_word, xhat_max, 0.f});
  TRIS_LAUNCH_CHECK();
  return 0;
}
