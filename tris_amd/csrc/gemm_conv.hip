// MFMA GEMM / implicit-GEMM convolution family for the TRIS Stage-1 hot path (gfx950): shape logic and entry points.
//
// One templated kernel family serves every dense product on the path:
//   * Linear / 1x1 conv (NHWC => plain GEMM), their dgrad (NN) and wgrad (TN, split-K)
//   * 3x3 conv forward, dgrad and wgrad as implicit GEMM (the im2col gather lives in the tile loaders) or as the direct
//     (window-image) kernels of conv_direct.hip
//   * batched products of the cross-modal attention
// The kernels live in gemm_core.h / gemm_fast.h and are instantiated per operand-kind pair by gemm_inst.hip (seven translation
// units); this file keeps what decides WHICH kernel runs: the cost model, the first-encounter autotuner, the choice between the
// direct and the implicit 3x3 kernels, the arithmetic mode, and the extern "C" entry points.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

#include "common.h"
#include "tris_hip.h"

extern "C" {
// options that live in norm.hip's translation unit (set through tris_set_option below)
extern __attribute__((visibility("hidden"))) int tris_internal_stream_form;
extern __attribute__((visibility("hidden"))) int tris_internal_col_blocks;
extern __attribute__((visibility("hidden"))) int tris_internal_ln_bwd_blocks;
extern __attribute__((visibility("hidden"))) int tris_internal_fin_block;
// REDUCE_WIDE=0: split-K slabs of small outputs are summed by the one-thread-per-four-columns kernel as well; 4 | 8 | 16: slab
// groups (waves) per block of the wide kernel (gemm_core.h)
__attribute__((visibility("hidden"))) int tris_internal_reduce_wide = 4;
// XCD_ORDER: -1 = the tuner's choice (LDS-DMA products only), 0 = never, 1 = every fast-kernel launch (gemm_core.h run_cfg)
__attribute__((visibility("hidden"))) int tris_internal_xcd_order = -1;
// RED_GRID: most blocks of a split-K reduce launch (grid-stride above that); 0 = one block per 1024 outputs
__attribute__((visibility("hidden"))) int tris_internal_red_grid = 0;
// FUSE_SPLITK: largest slice count whose slabs are summed by the last-arriving block of each tile inside the product's own launch
// (gemm_fast.h "fused split-K finish"; needs tris_splitk_tickets_next); 0 = always the separate reduce launch
__attribute__((visibility("hidden"))) int tris_internal_fuse_splitk = 8;
__attribute__((visibility("hidden"))) long tris_internal_fused_count = 0;
// option that lives in xattn_px.hip's translation unit
extern __attribute__((visibility("hidden"))) int tris_internal_xattn_px_slots;
}

namespace {

#include "gemm_params.h"

// Epilogue streams (C, residual, the BatchNorm-backward operands) of a launch that together exceed the 256 MB memory-side cache
// travel with the nontemporal policy -- norm.hip "BIG form" has the measurement; split-K slabs are re-read at once and stay cached.
static int stream_nt(const GemmParams& p, int batch) {
  if (!tris_internal_stream_form) return 0;
  const long streams = 1 + (p.resid ? 1 : 0) + (p.bnb_x ? 1 : 0) + (p.bnb_y ? 1 : 0);
  return (long)p.M * p.N * 4 * batch * streams > ((long)tris_internal_stream_form << 20) ? 1 : 0;
}

// ---- developer options ------------------------------------------------------------------------------------------------------
// Read from the environment ONCE, when the library is loaded; tests and tools change them through tris_set_option() (name =
// the environment variable without its TRIS_ prefix).  Nothing below reads the environment per call.
struct Options {
  int force_tile = 0;     // FORCE_TILE=128x128|128x64|64x64|128x32|256x128 -> 1..5
  int force_pipe = -1;    // FORCE_PIPE=0|1: loop structure of the x3 products
  int pipe_default = -1;  // PIPE=0|1: static default of the loop structure (the autotuner still times both)
  int conv_direct = -1;   // CONV_DIRECT=0 keeps the implicit GEMM, 1..6 forces a direct configuration where it applies
  int wgrad_direct = -1;  // WGRAD_DIRECT=0 keeps the implicit GEMM, 1..5 forces a direct configuration
  int bn_fold = 1;        // BN_FOLD=0: BatchNorm + ReLU never folded into the direct convolutions
  int stem_conv1 = 1;     // STEM_CONV1=0: the stem's first convolution through the generic kernels
  int wg_blocks = 256;    // WG_BLOCKS: blocks the direct weight gradient aims for (one per CU; 512 until round 6: on an idle device the same, inside the
                          // step -0.25 ms, means of two A/B sweeps -- profiles/r6_wg_blocks_ab.txt: fewer slabs to sum, fewer workgroups among the compute stream's)
  // (STREAM_FORM=0: element-wise passes never take the nontemporal one-piece-per-block form, n > 1: they do above n MB (1 = default = 256); COL_BLOCKS: blocks a column
  //  reduction aims for -- both live in norm.hip's translation unit: tris_internal_stream_form / tris_internal_col_blocks;
  //  XATTN_PX_SLOTS=n: workgroups per image of the pixel-row cross attention, 0 = as many as fit one per CU -- xattn_px.hip)
  char tune_log[256] = {0};  // TUNE_LOG=<file>: one line per tuned shape
};
static int parse_tile(const char* e) {
  return !e ? 0 : (!strcmp(e, "128x128") ? 1 : !strcmp(e, "128x64") ? 2 : !strcmp(e, "64x64") ? 3 : !strcmp(e, "128x32") ? 4
                   : !strcmp(e, "256x128") ? 5 : 0);
}
static bool set_option(Options& o, const char* name, const char* v) {
  const bool unset = (v == nullptr || v[0] == 0);
  if (!strcmp(name, "FORCE_TILE")) o.force_tile = parse_tile(v);
  else if (!strcmp(name, "FORCE_PIPE")) o.force_pipe = unset ? -1 : std::min(5, std::max(0, atoi(v)));   // (2 .. 5: the LDS-DMA loop where it applies)
  else if (!strcmp(name, "PIPE")) o.pipe_default = unset ? -1 : (v[0] == '0' ? 0 : 1);
  else if (!strcmp(name, "CONV_DIRECT")) o.conv_direct = unset ? -1 : atoi(v);
  else if (!strcmp(name, "WGRAD_DIRECT")) o.wgrad_direct = unset ? -1 : atoi(v);
  else if (!strcmp(name, "BN_FOLD")) o.bn_fold = unset ? 1 : (v[0] != '0');
  else if (!strcmp(name, "STEM_CONV1")) o.stem_conv1 = unset ? 1 : (v[0] != '0');
  else if (!strcmp(name, "WG_BLOCKS")) o.wg_blocks = unset ? 256 : std::max(1, atoi(v));
  else if (!strcmp(name, "STREAM_FORM")) tris_internal_stream_form = unset ? 256 : (atoi(v) == 1 ? 256 : std::max(0, atoi(v)));
  else if (!strcmp(name, "COL_BLOCKS")) tris_internal_col_blocks = unset ? 512 : std::max(1, atoi(v));
  else if (!strcmp(name, "LN_BWD_BLOCKS")) tris_internal_ln_bwd_blocks = unset ? 512 : std::min(512, std::max(1, atoi(v)));
  else if (!strcmp(name, "FIN_BLOCK")) tris_internal_fin_block = unset ? 256 : (atoi(v) >= 1024 ? 1024 : atoi(v) >= 512 ? 512 : 256);
  else if (!strcmp(name, "REDUCE_WIDE")) tris_internal_reduce_wide = unset ? 4 : std::max(0, atoi(v));
  else if (!strcmp(name, "RED_GRID")) tris_internal_red_grid = unset ? 0 : std::max(0, atoi(v));
  else if (!strcmp(name, "XCD_ORDER")) tris_internal_xcd_order = unset ? -1 : std::min(2, std::max(0, atoi(v)));
  else if (!strcmp(name, "FUSE_SPLITK")) tris_internal_fuse_splitk = unset ? TRIS_FUSE_SPLITK_DEFAULT : std::max(0, atoi(v));
  else if (!strcmp(name, "XATTN_PX_SLOTS")) tris_internal_xattn_px_slots = unset ? 0 : std::max(0, atoi(v));
  else if (!strcmp(name, "TUNE_LOG")) { strncpy(o.tune_log, unset ? "" : v, sizeof(o.tune_log) - 1); o.tune_log[sizeof(o.tune_log) - 1] = 0; }
  else return false;
  return true;
}
static Options init_options() {
  Options o;
  for (const char* n : {"FORCE_TILE", "FORCE_PIPE", "PIPE", "CONV_DIRECT", "WGRAD_DIRECT", "BN_FOLD", "STEM_CONV1", "WG_BLOCKS", "STREAM_FORM", "COL_BLOCKS", "XATTN_PX_SLOTS", "LN_BWD_BLOCKS", "FIN_BLOCK", "REDUCE_WIDE", "XCD_ORDER", "FUSE_SPLITK", "RED_GRID", "TUNE_LOG"}) {
    char env[64];
    snprintf(env, sizeof(env), "TRIS_%s", n);
    if (const char* v = getenv(env)) set_option(o, n, v);
  }
  return o;
}
static Options g_opt = init_options();

// arithmetic of the fast kernels: 0 = f32-input MFMA, 1 = split-bf16 x3 (6 bf16 MFMAs per product, fp32-class accuracy),
// 3 = h2 (two fp16 pieces per operand, 3 f16 MFMAs per product, power-of-two operand scales; x3_split.h)
// Process-wide default + per-thread override, both read on the HOST when a product is launched (a launch is otherwise
// stateless).  The override exists for callers that run some products in another arithmetic from their own thread without
// touching what other threads launch.
static int g_mode_default = 1;
static thread_local int g_mode_thread = -1;
#define g_gemm_mode (g_mode_thread >= 0 ? g_mode_thread : g_mode_default)
// 4 = h2 with both operands arriving as fp16 piece planes (gemm_fast.h PREC 4): never a process default, only ever set for ONE
// product by its arming (tris_h2_next_planes)
static inline bool mode_h2() { return g_gemm_mode == 3 || g_gemm_mode == 4; }

}  // namespace
// one configuration of one operand-kind pair: gemm_inst.hip (hidden symbols of the same shared object)
extern "C" __attribute__((visibility("hidden"))) unsigned* tris_internal_take_amax_next();   // (csrc/norm.hip)
#define TRIS_RUN_DECL(AK_, BK_)                                                                                                   \
  extern "C" __attribute__((visibility("hidden"))) int tris_internal_run_cfg_##AK_##BK_(const void* params, int batch, float* ws, \
                                                                                         void* stream, const void* cfg, int mode);
TRIS_RUN_DECL(0, 0) TRIS_RUN_DECL(0, 1) TRIS_RUN_DECL(1, 0) TRIS_RUN_DECL(1, 1) TRIS_RUN_DECL(2, 0) TRIS_RUN_DECL(2, 2) TRIS_RUN_DECL(1, 3)
#undef TRIS_RUN_DECL
namespace {
template <int AK, int BKIND>
int run_cfg(const GemmParams& p0, int batch, float* ws, hipStream_t st, const Cfg& cfg) {
  const int mode = g_gemm_mode;
  GemmParams p = p0;
  p.nt = stream_nt(p, batch);
  if (AK == A_ROWK && BKIND == B_NK) return tris_internal_run_cfg_00(&p, batch, ws, st, &cfg, mode);
  if (AK == A_ROWK && BKIND == B_KN) return tris_internal_run_cfg_01(&p, batch, ws, st, &cfg, mode);
  if (AK == A_COLK && BKIND == B_NK) return tris_internal_run_cfg_10(&p, batch, ws, st, &cfg, mode);
  if (AK == A_COLK && BKIND == B_KN) return tris_internal_run_cfg_11(&p, batch, ws, st, &cfg, mode);
  if (AK == A_IM2COL && BKIND == B_NK) return tris_internal_run_cfg_20(&p, batch, ws, st, &cfg, mode);
  if (AK == A_IM2COL && BKIND == B_KN_DGRAD) return tris_internal_run_cfg_22(&p, batch, ws, st, &cfg, mode);
  if (AK == A_COLK && BKIND == B_KN_IM2COL) return tris_internal_run_cfg_13(&p, batch, ws, st, &cfg, mode);
  return (int)hipErrorInvalidValue;
}

// "h2" for ONE product: tris_h2_next() arms the calling thread; the next dense product launched from it (tris_gemm_f32,
// tris_gemm_bnstat_f32, tris_gemm_bnbwd_f32) runs with two fp16 pieces per operand (PREC 3) and these operand scales, whatever
// the process-wide mode -- if the fast kernel serves its shape; otherwise it runs as usual.  One shot.
struct H2Next { const unsigned* a; const unsigned* b; float sa, sb; bool armed; bool planes; int flags; };
static thread_local H2Next g_h2_next = {nullptr, nullptr, 0.f, 0.f, false, false, 0};
// every entry point that honours the arming TAKES it first thing (h2_take), whether or not it then launches anything: an arming
// never survives the call it was made for
static H2Next h2_take() {
  H2Next n = g_h2_next;
  g_h2_next.armed = false;
  return n;
}
struct H2Guard {
  int saved;
  bool on;
  bool bad;   // operand planes handed to a product the fast kernel does not serve: the entry point fails (there is no fp32 operand to fall back on)
  H2Guard(GemmParams& p, const H2Next& n) : saved(g_mode_thread), on(false), bad(false) {
    if (!n.armed) return;
    if (!gemm_fast_ok(p)) { bad = n.planes; return; }
    p.h2_amaxA = n.a; p.h2_amaxB = n.b; p.h2_sA = n.sa; p.h2_sB = n.sb;
    g_mode_thread = n.planes ? 4 : 3;
    p.bnb_y_pl = (n.planes && (n.flags & 4)) ? 2 : (n.planes && (n.flags & 1)) ? 1 : 0;   // (4: bn_y is the BYTE mask of tris_bn_mask_next)
    on = true;
  }
  ~H2Guard() { if (on) g_mode_thread = saved; }
};

static bool pipe_ok(const GemmParams& p) { return (g_gemm_mode == 1 || mode_h2()) && gemm_fast_ok(p); }
// static choice of the loop structure (the autotuner times both)
static int default_pipe(const GemmParams& p, int bm, int bn, int splitk) {
  if (!pipe_ok(p) || bn == 32) return 0;
  if (g_opt.force_pipe >= 0) return g_opt.force_pipe;   // (2 | 3 where the LDS-DMA loop does not apply: run_cfg runs the classic loop)
  // measured (tools/gemm_bench.py, autotuned tiles, classic | pipelined): the 3x3 convolutions from 128 channels up gain
  // 3-13 % (fwd 40x40x256: 165 -> 175, 20x20x512: 136 -> 154, wgrad 20x20x512: 154 -> 174 TFLOP/s); the short-K 1x1
  // products and the transformer GEMMs are on par or a few % slower -> static default by kind, the autotuner times both
  if (g_opt.pipe_default >= 0) return g_opt.pipe_default;
  if (mode_h2()) return 0;   // (h2: the classic loop unless the tuner finds the pipelined one faster for the shape)
  return p.gC >= 128 ? 1 : 0;
}

// Tile / split-K choice by a small cost model (cycles on the MFMA pipe); also the starting point of the autotuner.
//   per-wave cycles per 32-deep k step = (BM/64)*(BN/64)*c; a block owns a CU's 4 SIMDs; blocks beyond the 256 CUs queue.
//   Split-K adds a slab round trip + a reduce launch.  Smaller tiles pay extra LDS / L2 traffic.
static Cfg heuristic_cfg(const GemmParams& p, int batch, const float* ws, long ws_bytes) {
  const bool can_split = (batch == 1 && ws != nullptr && p.K >= 512);
  int bm = 64, bn = 64, splitk = 1;
  if (p.stat_part != nullptr) {  // fused BN statistics: fixed 128-row tiles (the caller sizes the partial buffer), no split-K
    bm = 128;
    bn = p.N <= 32 ? 32 : p.N <= 64 ? 64 : 128;  // (run_cfg turns 128x32 into 64x64 when the fast kernel does not apply)
  } else if (can_split && p.K >= 4096) {
    // long-K reductions (wgrad over pixels): 2 co-resident blocks per CU keep the MFMA pipe busy across the
    // barrier / staging phases, and ~512 blocks smooth the wave quantisation -> split K until there are ~512 blocks
    if (p.N <= 64) { bn = 64; bm = p.M >= 128 ? 128 : 64; }
    else if (p.M >= 128 && p.N >= 128) { bm = bn = 128; }
    const long tiles = (long)cdiv(p.M, bm) * cdiv(p.N, bn);
    splitk = (int)min((long)cdiv(512, tiles), (long)(p.K / 256));
    const long per = (long)p.M * p.N * (long)sizeof(float);
    if ((long)splitk * per > ws_bytes) splitk = (int)(ws_bytes / per);
    if (splitk < 2) splitk = 1;
  } else {
    static const int cand[4][2] = {{128, 128}, {128, 64}, {64, 64}, {128, 32}};
    static const double pen[4] = {1.0, 1.08, 1.2, 1.15};
    const bool fastk = gemm_fast_ok(p);
    double best = 1e30;
    for (int c = 0; c < 4; ++c) {
      const int cbm = cand[c][0], cbn = cand[c][1];
      if (c != 2 && p.M < 96) continue;          // 128-row tiles on a tiny M waste the MFMA
      if (c == 0 && p.N <= 64) continue;
      if (c == 3 && !(p.N <= 32 && fastk)) continue;  // 32-wide outputs (stem convolutions): half of a 64-wide tile would be padding
      const long tiles = (long)cdiv(p.M, cbm) * cdiv(p.N, cbn) * batch;
      // MFMA cycles of one 32x32 fragment pair per 32-deep step: 16 f32 MFMAs x 64, or 12 bf16 MFMAs x 32 (x3 mode)
      const double per_k32 = (cbm / 64) * (cbn / 64.0) * (g_gemm_mode == 1 ? 384.0 + 250.0 : mode_h2() ? 192.0 + 200.0 : 1024.0) * pen[c];
      const int smax = can_split ? (int)min((long)64, (long)(p.K / 256)) : 1;
      for (int sk = 1; sk <= smax; sk = (sk < 4 ? sk + 1 : sk + sk / 2)) {
        if (sk > 1 && (long)sk * p.M * p.N * (long)sizeof(float) > ws_bytes) break;
        const double ksteps = (double)cdiv(cdiv(p.K, sk), 32);
        const double blocks = (double)tiles * sk;
        double t = (ksteps * per_k32 + 3000.0) * (blocks <= 256.0 ? 1.0 : blocks / 256.0);
        if (sk > 1) t += 14000.0 + 2.0 * sk * (double)p.M * p.N * 4.0 / 2000.0;  // reduce launch + slab bytes @ ~2 kB/cycle
        if (t < best) { best = t; bm = cbm; bn = cbn; splitk = sk; }
      }
    }
  }
  {  // developer option FORCE_TILE overrides the tile choice
    const int forced = g_opt.force_tile;
    if (forced == 1 && p.N > 64) { bm = 128; bn = 128; }
    if (forced == 2) { bm = 128; bn = 64; }
    if (forced == 3) { bm = 64; bn = 64; }
    if (forced == 4 && p.N <= 32) { bm = 128; bn = 32; }
    if (forced == 5 && p.N > 64 && p.M >= 256 && p.stat_part == nullptr) { bm = 256; bn = 128; }   // (pipelined x3 loop only: run_cfg falls back)
  }
  Cfg c = {bm, bn, splitk, default_pipe(p, bm, bn, splitk)};
  return c;
}

// ---- autotuner ------------------------------------------------------------------------------------------------------------
// The first time a (kind, M, N, K, batch, mode) product is seen, every admissible (tile, split-K) pair is launched on the
// caller's buffers (the kernels are idempotent), timed with HIP events, and the fastest is cached for the life of the
// process.  Host-synchronising, so only outside stream capture; TRIS_AUTOTUNE=0 keeps the cost model.
struct TuneKey {
  int ak, bk, M, N, K, batch, mode;
  bool operator<(const TuneKey& o) const {
    return std::tie(ak, bk, M, N, K, batch, mode) < std::tie(o.ak, o.bk, o.M, o.N, o.K, o.batch, o.mode);
  }
};
static std::map<TuneKey, Cfg> g_tuned;
static std::mutex g_tune_mu;

static int g_autotune = -1;  // -1: read TRIS_AUTOTUNE on first use
static bool autotune_enabled() {
  if (g_autotune < 0) { const char* e = getenv("TRIS_AUTOTUNE"); g_autotune = (e && e[0] == '0') ? 0 : 1; }
  return g_autotune == 1;
}

template <int AK, int BKIND>
int launch_cfg(GemmParams& p, int batch, float* ws, long ws_bytes, hipStream_t st) {
  Cfg h = heuristic_cfg(p, batch, ws, ws_bytes);
  if (!autotune_enabled() || g_opt.force_tile) return run_cfg<AK, BKIND>(p, batch, ws, st, h);
  const bool stat = p.stat_part != nullptr;  // fused BN statistics: 128-row tiles and no split-K are fixed, the tile width is tuned
  const TuneKey key = {AK, BKIND + (stat ? 16 : 0) + (p.bnb_x != nullptr ? 32 : 0), p.M, p.N, p.K, batch, g_gemm_mode};
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tuned.find(key);
    if (it != g_tuned.end()) return run_cfg<AK, BKIND>(p, batch, ws, st, it->second);
  }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;   // tuning synchronises the device: never under stream capture
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
    return run_cfg<AK, BKIND>(p, batch, ws, st, h);          // (a shape first met during capture runs the cost model's choice)
  // candidates
  static const int tiles[5][2] = {{128, 128}, {128, 64}, {64, 64}, {128, 32}, {256, 128}};
  const bool fastk = gemm_fast_ok(p);
  static const int sks[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128};
  const bool can_split = (batch == 1 && ws != nullptr && p.K >= 512) && !stat;
  const bool can_pipe = pipe_ok(p) && g_opt.force_pipe != 0;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return run_cfg<AK, BKIND>(p, batch, ws, st, h);
  (void)hipDeviceSynchronize();  // drain the other streams: candidates are timed on an otherwise idle device
  Cfg best = h;
  float best_ms = 1e30f;
  auto time_cfg = [&](Cfg c) -> float {
    float ms_min = 1e30f;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0, st);
      if (run_cfg<AK, BKIND>(p, batch, ws, st, c) != 0) return 1e30f;
      (void)hipEventRecord(e1, st);
      if (hipEventSynchronize(e1) != hipSuccess) return 1e30f;
      float ms = 1e30f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      ms_min = ms < ms_min ? ms : ms_min;
    }
    return ms_min;
  };
  for (int t = 0; t < 5; ++t) {
    const int cbm = tiles[t][0], cbn = tiles[t][1];
    if (t != 2 && p.M < 96) continue;
    if ((t == 0 || t == 4) && p.N <= 64) continue;
    if (t == 3 && !(p.N <= 32 && fastk)) continue;
    if (t == 4 && (!can_pipe || p.M < 256)) continue;   // the 256-row tile exists in the pipelined form only
    if (stat && cbm != 128) continue;
    const long ntiles = (long)cdiv(p.M, cbm) * cdiv(p.N, cbn) * batch;
    for (int sk : sks) {
      if (sk > 1 && (!can_split || sk > p.K / 256 || (long)sk * p.M * p.N * (long)sizeof(float) > ws_bytes)) break;
      if (sk > 1 && ntiles * sk > 1536) break;  // more than ~6 blocks per CU buys nothing
      if (sk * 8 < h.splitk && ntiles * sk < 256) continue;  // a handful of blocks walking a huge K serially: not worth timing
      for (int pipe = (can_pipe && t != 3) ? 1 : 0; pipe >= (t == 4 || g_opt.force_pipe == 1 ? (can_pipe && t != 3 ? 1 : 0) : 0); --pipe) {
        const Cfg c = {cbm, cbn, sk, pipe};
        const float ms = time_cfg(c);
        if (ms < best_ms) { best_ms = ms; best = c; }
      }
      if (g_gemm_mode == 4 && AK == A_ROWK && BKIND == B_NK && t != 3 && fastk && g_opt.force_pipe < 0) {
        // operand planes, row-major A x B^T: the LDS-DMA loop (gemm_fast.h NSTG 4), eight- and four-wave forms of the 128 x 128 tile
        // (+ 2: with the XCD-contiguous tile order)
        for (int gp : {2, 3, 4, 5}) {
          if ((gp == 3 || gp == 5) && t != 0) continue;
          const Cfg c = {cbm, cbn, sk, gp};
          const float ms = time_cfg(c);
          if (ms < best_ms) { best_ms = ms; best = c; }
        }
      }
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tuned[key] = best;
    if (g_opt.tune_log[0]) {  // developer option: one line per tuned shape (idle-device time of the winner)
      if (FILE* f = fopen(g_opt.tune_log, "a")) {
        fprintf(f, "ak=%d bkind=%d M=%d N=%d K=%d batch=%d mode=%d -> %dx%d sk=%d pipe=%d  %.1f us  %.1f TFLOP/s\n", AK, key.bk,
                p.M, p.N, p.K, batch, g_gemm_mode, best.bm, best.bn, best.splitk, best.pipe, best_ms * 1e3f,
                2.0 * p.M * p.N * p.K * batch / (best_ms * 1e-3) * 1e-12);
        fclose(f);
      }
    }
  }
  return run_cfg<AK, BKIND>(p, batch, ws, st, best);  // leave the outputs of the chosen configuration
}





// ---- direct 3x3 convolution: configurations of the A_HALO kernels (gemm_fast.h) and the choice between them and the
// implicit GEMM ---------------------------------------------------------------------------------------------------------------
static long g_direct_launches[2] = {0, 0};   // diagnostics: launches of the direct convolution / direct weight-gradient kernels
// per-thread override of Options::conv_direct (tris_set_conv_direct_thread): batch-invariant evaluation pins the implicit GEMM for
// the products IT launches without touching what other threads run
static thread_local int g_conv_direct_thread = -1;
#include "conv_direct_cfg.h"

// launchers in conv_direct.hip (hidden symbols of the same shared object)
}  // namespace
#define TRIS_DIRECT_DECL(P_)                                                                                                          \
  extern "C" __attribute__((visibility("hidden"))) int tris_internal_run_halo_p##P_(const void* params, int id, int dgrad, void* stream); \
  extern "C" __attribute__((visibility("hidden"))) int tris_internal_run_wgrad_direct_p##P_(                                            \
      int id, const float* X, const float* dY, float* dW, int B, int H, int W, int Ci, int Co, float* ws, long ws_bytes, int blocks,    \
      void* stream, const float* mean, const float* invstd, const float* gamma, const float* beta, const unsigned* amax_dy,            \
      const unsigned* amax_x);
TRIS_DIRECT_DECL(1) TRIS_DIRECT_DECL(3) TRIS_DIRECT_DECL(4)
#undef TRIS_DIRECT_DECL
extern "C" __attribute__((visibility("hidden"))) int tris_internal_stem_conv1(const float* X, const float* Wt, float* Y, int B, int H, int W,
    int Cin, int Cout, int stride, double* stat_part, void* stream);
namespace {
template <int BKIND>
int run_halo(const GemmParams& p0, int id, hipStream_t st) {
  const int dg = BKIND == B_KN_DGRAD ? 1 : 0;
  GemmParams p = p0;
  p.nt = stream_nt(p, 1);
  const int rc = g_gemm_mode == 4 ? tris_internal_run_halo_p4(&p, id, dg, st)
                 : g_gemm_mode == 3 ? tris_internal_run_halo_p3(&p, id, dg, st) : tris_internal_run_halo_p1(&p, id, dg, st);
  if (rc == 0) ++g_direct_launches[0];
  return rc;
}
// h2: amax words of dY and of the convolution's input (GemmParams::h2_amaxA / h2_amaxB of the armed product)
static int run_wgrad_direct(int id, const float* X, const float* dY, float* dW, int B, int H, int W, int Ci, int Co, float* ws,
                            long ws_bytes, hipStream_t st, BnIn bn = BnIn{nullptr, nullptr, nullptr, nullptr},
                            const unsigned* amax_dy = nullptr, const unsigned* amax_x = nullptr) {
  const int rc = g_gemm_mode == 4 ? tris_internal_run_wgrad_direct_p4(id, X, dY, dW, B, H, W, Ci, Co, ws, ws_bytes, g_opt.wg_blocks, st,
                                                                      nullptr, nullptr, nullptr, nullptr, amax_dy, amax_x)
                 : g_gemm_mode == 3 ? tris_internal_run_wgrad_direct_p3(id, X, dY, dW, B, H, W, Ci, Co, ws, ws_bytes, g_opt.wg_blocks, st,
                                                                      bn.mean, bn.invstd, bn.gamma, bn.beta, amax_dy, amax_x)
                                  : tris_internal_run_wgrad_direct_p1(id, X, dY, dW, B, H, W, Ci, Co, ws, ws_bytes, g_opt.wg_blocks, st,
                                                                      bn.mean, bn.invstd, bn.gamma, bn.beta, nullptr, nullptr);
  if (rc == 0) ++g_direct_launches[1];
  return rc;
}
static int run_stem_conv1(const float* X, const float* Wt, float* Y, int B, int H, int W, int Cin, int Cout, int stride,
                          double* stat_part, hipStream_t st) {
  if (!g_opt.stem_conv1) return 0;
  return tris_internal_stem_conv1(X, Wt, Y, B, H, W, Cin, Cout, stride, stat_part, st);
}

static bool halo_shape_ok(const GemmParams& p) {
  return (g_gemm_mode == 1 || mode_h2()) && p.gStride == 1 && p.gC % 16 == 0 && p.fastB && al16(p.A) && al16(p.B) && p.N % 4 == 0 &&
         p.gHo == p.gH && p.gWo == p.gW && (long)p.gB * p.gH * p.gW * p.gC < (1L << 31);
}
// static choice (no autotuning, or under stream capture), from the measured table in DESIGN.md: the 2-D patch kernels win
// wherever they apply (W, H multiples of 16: the stem and the 80 x 80 stages); the flattened-window kernels pay only for the
// data gradients of the 40 x 40 / 20 x 20 stages from 256 channels up.  0 = implicit GEMM.
static int halo_static_choice(const GemmParams& p, bool dgrad) {
  if (p.N <= 32 && halo_ok(p, 3)) return 3;
  if (p.N <= 64 && halo_ok(p, 2)) return 2;
  if (halo_ok(p, 6)) return 6;
  if (halo_ok(p, 1)) return 1;
  if (dgrad && p.N >= 256 && p.gW >= 20 && halo_ok(p, 4)) return 4;
  return 0;
}

// direct_only: the caller needs a direct kernel (p.in_mean: BatchNorm + ReLU folded into the window staging) and has checked
// with tris_conv3x3_bnin_ok that one applies: the implicit GEMM is neither timed nor chosen.
template <int BKIND>
int conv3_dispatch(GemmParams& p, hipStream_t st, int* stat_rows, bool direct_only = false) {
  const bool stat = p.stat_part != nullptr;
  const int first = halo_shape_ok(p) ? halo_static_choice(p, BKIND == B_KN_DGRAD) : 0;
  auto direct = [&](int id) {
    if (stat_rows) *stat_rows = stat ? halo_tiles_m(p, id) : 0;
    return run_halo<BKIND>(p, id, st);
  };
  auto im2col = [&]() {
    if (direct_only) return first ? direct(first) : (int)hipErrorInvalidValue;   // (stands in for "the default" below)
    if (stat_rows) *stat_rows = stat ? cdiv(p.M, 128) : 0;
    return launch_cfg<A_IM2COL, BKIND>(p, 1, nullptr, 0, st);
  };
  const int forced = g_conv_direct_thread >= 0 ? g_conv_direct_thread : g_opt.conv_direct;
  if (forced == 0 || !halo_shape_ok(p)) return im2col();
  if (forced > 0) return (forced < kHaloN && halo_ok(p, forced)) ? direct(forced) : im2col();
  bool any = false;
  for (int id = 1; id < kHaloN; ++id) any = any || halo_ok(p, id);
  if (!any) return im2col();
  const TuneKey key = {A_HALO, BKIND + (stat ? 16 : 0) + (direct_only ? 32 : 0) + (p.bnb_x != nullptr ? 64 : 0), p.M, p.N, p.K,
                       p.gH * 4096 + p.gW, g_gemm_mode};
  int cached = -1;
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);   // (released before the launch: the implicit GEMM looks up its own tile under it)
    auto it = g_tuned.find(key);
    if (it != g_tuned.end()) cached = it->second.bm;
  }
  if (cached >= 0) return cached ? direct(cached) : im2col();
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (!autotune_enabled() || hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
    return first ? direct(first) : im2col();
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return first ? direct(first) : im2col();
  int rc = im2col();   // (tunes the implicit GEMM's own tile on first sight)
  if (rc != 0) return rc;
  (void)hipDeviceSynchronize();
  auto timed = [&](int id) -> float {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0, st);
      if ((id ? direct(id) : im2col()) != 0) return 1e30f;
      (void)hipEventRecord(e1, st);
      if (hipEventSynchronize(e1) != hipSuccess) return 1e30f;
      float ms = 1e30f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    return best;
  };
  int best_id = 0;
  const float t_im2col = timed(0);
  float best_ms = t_im2col;
  for (int id = 1; id < kHaloN; ++id) {
    if (!halo_ok(p, id)) continue;
    const float ms = timed(id);
    if (ms < best_ms) { best_ms = ms; best_id = id; }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tuned[key] = Cfg{best_id, 0, 1, 0};
    if (g_opt.tune_log[0]) {
      if (FILE* f = fopen(g_opt.tune_log, "a")) {
        fprintf(f, "conv3x3 bkind=%d M=%d N=%d K=%d HxW=%dx%d -> %s %d  %.1f us  %.1f TFLOP/s  (implicit GEMM %.1f us) mode=%d\n", key.bk,
                p.M, p.N, p.K, p.gH, p.gW, best_id ? "direct" : "implicit", best_id, best_ms * 1e3f,
                2.0 * p.M * p.N * p.K / (best_ms * 1e-3) * 1e-12, t_im2col * 1e3f, g_gemm_mode);
        fclose(f);
      }
    }
  }
  return best_id ? direct(best_id) : im2col();   // leave the outputs (and *stat_rows) of the chosen kernel
}


}  // namespace

// may the fused-statistics epilogue of the fast kernel serve this product?
static bool stats_eligible(const GemmParams& p) { return gemm_fast_ok(p) && p.M >= 128; }

// One-shot arming of the NEXT tris_gemm_f32 of the calling thread with two epilogue extras (GemmParams::pre_out / dact_x)
struct EpiNext { float* pre; const float* dact; bool armed; };
static thread_local EpiNext g_epi_next = {nullptr, nullptr, false};
extern "C" __attribute__((visibility("hidden"))) const int* tris_internal_rows_limit();   // (norm.hip: tris_rows_limit_thread)
extern "C" int tris_gemm_epilogue_next(float* pre_out, const float* dact_x) {
  g_epi_next = {pre_out, dact_x, pre_out != nullptr || dact_x != nullptr};
  return 0;
}

// One-shot arming of the NEXT tris_gemm_f32 of the calling thread with a ticket array for the fused split-K finish (gemm_fast.h):
// `count` ints, ZERO when handed over and left zero by every launch; private to the stream the product is launched on (two
// products in flight at once must not share one).  Without it a split-K product sums its slabs in a second launch, as before.
struct TicketsNext { int* t; int n; };
static thread_local TicketsNext g_tickets_next = {nullptr, 0};
extern "C" long tris_splitk_fused_launches(void) { return __atomic_load_n(&tris_internal_fused_count, __ATOMIC_RELAXED); }
extern "C" int tris_splitk_tickets_next(int* tickets, int count) {
  g_tickets_next = {count > 0 ? tickets : nullptr, count > 0 ? count : 0};
  return 0;
}

extern "C" int tris_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, long lda, long ldb,
                             long ldc, int transA, int transB, int batch, long sA, long sB, long sC,
                             const float* bias, int bias_mode, const float* resid, long ldr, long sR, int act,
                             float alpha, float* workspace, long ws_bytes, void* stream) {
  const H2Next h2n = h2_take();
  unsigned* amax_out = tris_internal_take_amax_next();   // (one-shot by-product: the amax word of C, tris_amax_next)
  const EpiNext epi = g_epi_next;
  g_epi_next.armed = false;
  const TicketsNext tkn = g_tickets_next;
  g_tickets_next = {nullptr, 0};
  if (M <= 0 || N <= 0 || batch <= 0) return 0;
  if (K <= 0) return (int)hipErrorInvalidValue;
  GemmParams p = {};
  p.amax_out = amax_out;
  p.tickets = tkn.t;
  p.tickets_n = tkn.n;
  if (!transA && batch == 1) p.m_limit = tris_internal_rows_limit();
  if (epi.armed) { p.pre_out = epi.pre; p.dact_x = epi.dact; }
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.sA = sA; p.sB = sB; p.sC = sC;
  p.bias = bias; p.bias_mode = bias ? bias_mode : 0;
  p.resid = resid; p.ldr = ldr; p.sR = sR; p.act = act; p.alpha = alpha;
  p.vecA = al16(A) && (lda % 4 == 0) && (sA % 4 == 0);
  p.vecB = al16(B) && (ldb % 4 == 0) && (sB % 4 == 0);
  p.fastA = p.vecA && (!transA || M % 4 == 0);
  p.fastB = p.vecB && (transB || N % 4 == 0);
  if (transA && !transB && K % 32 != 0 && K > 32 && p.fastA && p.fastB && M >= 4 && N >= 4 && !epi.armed) {
    // both operands k-major (a weight gradient) over a k extent that is not a multiple of 32: the fast kernel runs over K rounded
    // up and its loaders zero the rows past the end (the generic kernel took 1.4 ms for the 768 x 768 x 19248 products of ViT-B/16)
    p.Kv = K;
    p.K = (K + 31) & ~31;
  }
  hipStream_t st = (hipStream_t)stream;
  if (epi.armed) {   // the extras live in the fast kernel's one-pass epilogue only: no split-K, no generic kernel
    if (batch != 1 || transA || !gemm_fast_ok(p)) return TRIS_DECLINED;
    workspace = nullptr;
    ws_bytes = 0;
  }
  H2Guard h2(p, h2n);
  if (h2.bad) return (int)hipErrorInvalidValue;
  if (!transA && transB) return launch_cfg<A_ROWK, B_NK>(p, batch, workspace, ws_bytes, st);
  if (!transA && !transB) return launch_cfg<A_ROWK, B_KN>(p, batch, workspace, ws_bytes, st);
  if (transA && !transB) return launch_cfg<A_COLK, B_KN>(p, batch, workspace, ws_bytes, st);
  return launch_cfg<A_COLK, B_NK>(p, batch, workspace, ws_bytes, st);
}

// Y[B,Ho,Wo,Cout] = conv3x3(X[B,H,W,Cin], Wt[Cout][3][3][Cin]), pad 1, stride 1|2, optional fused ReLU-less epilogue.
extern "C" int tris_conv3x3_fwd_f32(const float* X, const float* Wt, float* Y, int B, int H, int W, int Cin, int Cout,
                                    int stride, void* stream) {
  const H2Next h2n = h2_take();
  int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  GemmParams p = {};
  p.A = X; p.B = Wt; p.C = Y;
  p.M = B * Ho * Wo; p.N = Cout; p.K = 9 * Cin;
  p.ldb = 9L * Cin; p.ldc = Cout; p.alpha = 1.f;
  p.gH = H; p.gW = W; p.gC = Cin; p.gHo = Ho; p.gWo = Wo; p.gStride = stride; p.gB = B;
  p.vecA = al16(X) && (Cin % 16 == 0);
  p.vecB = al16(Wt) && ((9 * Cin) % 4 == 0);
  p.fastA = al16(X) && (Cin % 32 == 0);
  p.fastB = p.vecB;
  if (const int rows = run_stem_conv1(X, Wt, Y, B, H, W, Cin, Cout, stride, nullptr, (hipStream_t)stream))
    return rows > 0 ? 0 : (int)hipErrorLaunchFailure;
  H2Guard h2(p, h2n);   // (armed: the implicit GEMM in h2 -- the direct kernels exist in x3 only)
  if (h2.bad) return (int)hipErrorInvalidValue;
  return conv3_dispatch<B_NK>(p, (hipStream_t)stream, nullptr);
}

// dX[B,H,W,Cin] = conv3x3_transpose(dY[B,H,W,Cout], Wt), stride 1 only: a 3x3 conv of dY with the taps mirrored.
extern "C" int tris_conv3x3_dgrad_f32(const float* dY, const float* Wt, float* dX, int B, int H, int W, int Cin,
                                      int Cout, void* stream) {
  const H2Next h2n = h2_take();
  GemmParams p = {};
  p.A = dY; p.B = Wt; p.C = dX;
  p.M = B * H * W; p.N = Cin; p.K = 9 * Cout;
  p.ldc = Cin; p.alpha = 1.f;
  p.gH = H; p.gW = W; p.gC = Cout; p.gHo = H; p.gWo = W; p.gStride = 1; p.gB = B;
  p.wCin = Cin; p.wCout = Cout;
  p.vecA = al16(dY) && (Cout % 16 == 0);
  p.vecB = al16(Wt) && (Cin % 4 == 0);
  p.fastA = al16(dY) && (Cout % 32 == 0);
  p.fastB = p.vecB;
  if (Cout % 16 != 0) return (int)hipErrorInvalidValue;  // k tile must not straddle taps for the B loader
  H2Guard h2(p, h2n);
  if (h2.bad) return (int)hipErrorInvalidValue;
  return conv3_dispatch<B_KN_DGRAD>(p, (hipStream_t)stream, nullptr);
}

// Data gradient of a 3x3 convolution whose INPUT is the output of a train-mode BatchNorm + ReLU (no residual): the reduction
// pass of that BatchNorm's backward rides in the epilogue, as in tris_gemm_bnbwd_f32 (mask recomputed from bn_x / gamma / beta).
// dZ[B,H,W,Cin] <- masked gradient; part <- [*part_rows][2][Cin] fp64 partial; *part_rows = 0: not eligible, nothing launched.
extern "C" int tris_conv3x3_dgrad_bnbwd_f32(const float* dY, const float* Wt, float* dZ, int B, int H, int W, int Cin, int Cout,
                                            const float* bn_x, const float* mean, const float* invstd, const float* gamma,
                                            const float* beta, double* part, int* part_rows, void* stream) {
  const H2Next h2n = h2_take();
  unsigned* amax_out = tris_internal_take_amax_next();   // (one-shot by-product: the amax word of the masked gradient dZ)
  *part_rows = 0;
  if (bn_x == nullptr || part == nullptr || gamma == nullptr || beta == nullptr) return (int)hipErrorInvalidValue;
  GemmParams p = {};
  p.amax_out = amax_out;
  p.A = dY; p.B = Wt; p.C = dZ;
  p.M = B * H * W; p.N = Cin; p.K = 9 * Cout;
  p.ldc = Cin; p.alpha = 1.f;
  p.gH = H; p.gW = W; p.gC = Cout; p.gHo = H; p.gWo = W; p.gStride = 1; p.gB = B;
  p.wCin = Cin; p.wCout = Cout;
  p.vecA = al16(dY) && (Cout % 16 == 0);
  p.vecB = al16(Wt) && (Cin % 4 == 0);
  p.fastA = al16(dY) && (Cout % 32 == 0);
  p.fastB = p.vecB;
  if (Cout % 16 != 0) return (int)hipErrorInvalidValue;
  const bool al = al16(dZ) && al16(bn_x) && al16(mean) && al16(invstd) && al16(gamma) && al16(beta) && Cin % 4 == 0;
  if (!stats_eligible(p) || !al) return 0;
  p.stat_part = part;
  p.bnb_x = bn_x; p.bnb_mean = mean; p.bnb_invstd = invstd; p.bnb_gamma = gamma; p.bnb_beta = beta;
  H2Guard h2(p, h2n);
  if (h2.bad) return (int)hipErrorInvalidValue;
  return conv3_dispatch<B_KN_DGRAD>(p, (hipStream_t)stream, part_rows);
}

// ---- weight gradient of the stem's first convolution (Cin = 3: the 27-wide "N" does not fit the tiled kernels) -----------
// dW[co][tap][ci] = sum over output pixels of dY[p][co] * X[in(p, tap)][ci] is ONE 32 x 32 output tile with a reduction
// over ~10^6 pixels.  Each wave walks a contiguous pixel range with v_mfma_f32_32x32x2_f32 (2 pixels per step: lane
// (m = co, kh) reads dY[p + kh][co] -- one 128-byte line per pixel --, lane (n = (tap, ci), kh) gathers its input value
// with incrementally maintained coordinates, no divisions in the loop); the block's four tiles meet in LDS, per-block
// partials are summed in fixed order by a second launch.  HBM-bound on the dY stream.
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                         float* __restrict__ part, int Bn, int H, int W, int Cin, int Cout,
                                                         int Ho, int Wo, int stride, long pix_per_wave) {
  __shared__ float red[3][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mn = lane & 31, kh = lane >> 5;
  const long total = (long)Bn * Ho * Wo;
  const long p_beg = ((long)blockIdx.x * 4 + wave) * pix_per_wave;
  const long p_end = min(total, p_beg + pix_per_wave);
  const int tap = mn / Cin, ci = mn - tap * Cin;
  const bool nv = mn < 9 * Cin;
  const int ky = tap / 3, kx = tap - ky * 3;
  const bool mv = mn < Cout;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // this lane's pixel (p_beg + kh) in (b, oy, ox) form; advanced by 2 per step
  long p = p_beg + kh;
  int b = (int)(p / ((long)Ho * Wo));
  int rem = (int)(p - (long)b * Ho * Wo);
  int oy = rem / Wo, ox = rem - oy * Wo;
  for (; p - kh < p_end; p += 16) {  // 8 MFMA steps (16 pixels) per trip: all 16 loads are in flight before the first MFMA
    float a[8], v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long pu = p + 2 * u;
      a[u] = 0.f;
      v[u] = 0.f;
      if (pu < p_end) {
        if (mv) a[u] = dY[pu * Cout + mn];
        const int iy = oy * stride - 1 + ky, ix = ox * stride - 1 + kx;
        if (nv && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v[u] = X[(((long)b * H + iy) * W + ix) * Cin + ci];
      }
      ox += 2;
      while (ox >= Wo) { ox -= Wo; if (++oy == Ho) { oy = 0; ++b; } }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], v[u], acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
    float* o = part + (long)blockIdx.x * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) {  // D[row = (r&3) + 8*(r>>2) + 4*kh][col = mn]
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
      o[row * 32 + mn] = acc[r] + red[0][r][lane] + red[1][r][lane] + red[2][r][lane];
    }
  }
}

// grid 32 blocks: block j sums the partials of outputs [32 j, 32 j + 32) -- 32 outputs x 8 slices of the partial list,
// fixed summation order (deterministic)
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ part, int nblk, float* __restrict__ dW,
                                                                int Cout, int N) {
  __shared__ float sh[8][32];
  const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + o;
  float s = 0.f;
  for (int bk = sl; bk < nblk; bk += 8) s += part[(long)bk * 1024 + idx];
  sh[sl][o] = s;
  __syncthreads();
  if (sl == 0) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += sh[q][o];
    const int row = idx >> 5, col = idx & 31;
    if (row < Cout && col < N) dW[row * N + col] = t;
  }
}

// Direct weight-gradient kernel (wgrad3x3_direct_kernel, configuration ids 1..5) or the implicit GEMM: every usable candidate is
// timed once per (shape, arithmetic, plain | BatchNorm-folded) on first sight and the winner cached, like the tile choice.  Option
// WGRAD_DIRECT=0 keeps the GEMM, =1..5 forces a configuration where it applies (tests).  gemm: null for the BatchNorm-folded form
// (only the direct kernels can normalise their input while staging it).
namespace {
template <class Direct, class Gemm>
int wgrad_pick(int B, int H, int W, int Cin, int Cout, long ws_bytes, bool bnin, hipStream_t st, Direct direct, Gemm gemm) {
  const int forced = g_opt.wgrad_direct;
  auto usable = [&](int id) { return wg_ok(id, H, W, Cin, Cout) && wg_slices(id, B, H, W, Cin, Cout, ws_bytes, g_opt.wg_blocks) >= 1; };
  int first = 0;   // static choice: the measured winners; the autotuner times every usable configuration
  for (int id : {1, 2, 5})
    if (!first && usable(id)) first = id;
  if (mode_h2() && usable(4)) first = 4;   // (h2: the 64 x 64 single-wave-per-quadrant tile folds its cross products least often)
  if (forced == 0 && !bnin) return gemm();
  if (forced > 0) return (forced < kWgN && usable(forced)) ? direct(forced) : (bnin ? (first ? direct(first) : (int)hipErrorInvalidValue) : gemm());
  if (!first) return bnin ? (int)hipErrorInvalidValue : gemm();
  const TuneKey key = {A_HALO, 64 + B_KN_IM2COL + (bnin ? 128 : 0), Cout, 9 * Cin, B * H * W, H * 4096 + W, g_gemm_mode};
  int cached = -1;
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tuned.find(key);
    if (it != g_tuned.end()) cached = it->second.bm;
  }
  if (cached >= 0) return cached ? direct(cached) : gemm();
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (!autotune_enabled() || hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return direct(first);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return direct(first);
  if (!bnin) {
    const int rc = gemm();   // (tunes the GEMM's own tile / split-K on first sight)
    if (rc != 0) return rc;
  }
  (void)hipDeviceSynchronize();
  auto timed = [&](int id) -> float {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0, st);
      if ((id ? direct(id) : gemm()) != 0) return 1e30f;
      (void)hipEventRecord(e1, st);
      if (hipEventSynchronize(e1) != hipSuccess) return 1e30f;
      float ms = 1e30f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    return best;
  };
  const float t_gemm = bnin ? 1e30f : timed(0);
  float best_ms = t_gemm;
  int best_id = 0;
  for (int id = 1; id < kWgN; ++id) {
    if (!usable(id)) continue;
    const float ms = timed(id);
    if (ms < best_ms) { best_ms = ms; best_id = id; }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tuned[key] = Cfg{best_id, 0, 1, 0};
    if (g_opt.tune_log[0]) {
      if (FILE* f = fopen(g_opt.tune_log, "a")) {
        fprintf(f, "wgrad3x3 Cout=%d Cin=%d pixels=%d HxW=%dx%d -> %s %d  %.1f us  %.1f TFLOP/s  (implicit GEMM %.1f us)%s mode=%d\n", Cout,
                Cin, B * H * W, H, W, best_id ? "direct" : "implicit", best_id, best_ms * 1e3f,
                2.0 * Cout * 9.0 * Cin * B * H * W / (best_ms * 1e-3) * 1e-12, bnin ? 0.f : t_gemm * 1e3f, bnin ? " bn-folded" : "",
                g_gemm_mode);
        fclose(f);
      }
    }
  }
  return best_id ? direct(best_id) : gemm();
}
}  // namespace

// dW[Cout][3][3][Cin] = sum over output pixels of dY (x) gathered X.  Split-K over pixels through `workspace`.
extern "C" int tris_conv3x3_wgrad_f32(const float* X, const float* dY, float* dW, int B, int H, int W, int Cin,
                                      int Cout, int stride, float* workspace, long ws_bytes, void* stream) {
  const H2Next h2n = h2_take();
  int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  if (9 * Cin <= 32 && Cout <= 32 && workspace != nullptr) {  // the stem's first conv: dedicated single-tile reduction
    const long total = (long)B * Ho * Wo;
    int nblk = (int)std::min<long>(512, std::max<long>(1, total / 512));
    long ppw = (cdiv(total, (long)nblk * 4) + 15) / 16 * 16;  // multiple of 16: whole 8-step trips, pixel pairs stay in one range
    nblk = cdiv(total, ppw * 4);
    if ((long)nblk * 1024 * (long)sizeof(float) <= ws_bytes) {
      hipStream_t st = (hipStream_t)stream;
      hipLaunchKernelGGL(stem_wgrad_kernel, dim3(nblk), dim3(256), 0, st, X, dY, workspace, B, H, W, Cin, Cout, Ho, Wo, stride,
                         ppw);
      TRIS_LAUNCH_CHECK();
      hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(32), dim3(256), 0, st, workspace, nblk, dW, Cout, 9 * Cin);
      TRIS_LAUNCH_CHECK();
      return 0;
    }
  }
  GemmParams p = {};
  p.A = dY; p.B = X; p.C = dW;
  p.M = Cout; p.N = 9 * Cin; p.K = B * Ho * Wo;
  p.lda = Cout; p.ldc = 9L * Cin; p.alpha = 1.f;
  p.gH = H; p.gW = W; p.gC = Cin; p.gHo = Ho; p.gWo = Wo; p.gStride = stride; p.gB = B;
  p.vecA = al16(dY) && (Cout % 4 == 0);
  p.vecB = al16(X) && (Cin % 4 == 0);
  p.fastA = p.vecA;
  p.fastB = p.vecB;
  hipStream_t st = (hipStream_t)stream;
  H2Guard h2(p, h2n);   // (armed: A = dY, B = X)
  if (h2.bad) return (int)hipErrorInvalidValue;
  auto gemm = [&]() { return launch_cfg<A_COLK, B_KN_IM2COL>(p, 1, workspace, ws_bytes, st); };
  const bool shape_ok = (g_gemm_mode == 1 || mode_h2()) && stride == 1 && workspace != nullptr && al16(X) && al16(dY) && al16(dW) && al16(workspace) &&
                        (long)B * H * W * std::max(Cin, Cout) < (1L << 31);
  auto direct = [&](int id) {
    return run_wgrad_direct(id, X, dY, dW, B, H, W, Cin, Cout, workspace, ws_bytes, st, BnIn{nullptr, nullptr, nullptr, nullptr},
                            p.h2_amaxA, p.h2_amaxB);
  };
  if (!shape_ok) return gemm();
  return wgrad_pick(B, H, W, Cin, Cout, ws_bytes, false, st, direct, gemm);
}

// Conv / 1x1-conv (GEMM) forward with the BatchNorm batch statistics of the OUTPUT fused into the epilogue.
// stat_part receives [stat_rows][2][N] fp64 partial (sum, sum of squares); *stat_rows (host) = number of partial rows,
// or 0 when the shape is not eligible for the fused path (then the caller runs tris_bn_stats_f32 as usual).

extern "C" int tris_gemm_bnstat_f32(const float* A, const float* B, float* C, int M, int N, int K, double* stat_part,
                                    int* stat_rows, void* stream) {
  const H2Next h2n = h2_take();
  GemmParams p = {};
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
  p.lda = K; p.ldb = K; p.ldc = N; p.alpha = 1.f;
  p.vecA = al16(A) && (K % 4 == 0);
  p.vecB = al16(B) && (K % 4 == 0);
  p.fastA = p.vecA;
  p.fastB = p.vecB;
  p.stat_part = stats_eligible(p) ? stat_part : nullptr;
  *stat_rows = p.stat_part ? cdiv(M, 128) : 0;
  H2Guard h2(p, h2n);
  if (h2.bad) return (int)hipErrorInvalidValue;
  return launch_cfg<A_ROWK, B_NK>(p, 1, nullptr, 0, (hipStream_t)stream);
}

// Data gradient of a 1x1 convolution / Linear, dZ[M,N] = mask(dY[M,K] . W[K,N] (+ resid)), with the reduction pass of the
// BatchNorm(+ReLU) BACKWARD that consumes it fused into the epilogue (GemmParams::bnb_*): bn_x [M,N] is that BatchNorm's raw
// input, bn_y its output (residual form: the mask is y > 0) or NULL (the mask is recomputed from bn_x, gamma, beta).  part receives
// [*part_rows][2][N] fp64 partial (sum dz, sum dz * xhat) -- finish with tris_part_finalize_f32; *part_rows = 0: the shape is not
// eligible (nothing was launched; the caller runs tris_gemm_f32 + tris_bn_bwd_reduce_f32 as usual).
extern "C" int tris_gemm_bnbwd_f32(const float* dY, const float* Wt, float* dZ, int M, int N, int K, const float* resid,
                                   long ldr, const float* bn_x, const float* bn_y, const float* mean, const float* invstd,
                                   const float* gamma, const float* beta, double* part, int* part_rows, void* stream) {
  const H2Next h2n = h2_take();
  unsigned* amax_out = tris_internal_take_amax_next();   // (one-shot by-product: the amax word of the masked gradient dZ)
  *part_rows = 0;
  if (M <= 0 || N <= 0 || K <= 0 || bn_x == nullptr || part == nullptr) return (int)hipErrorInvalidValue;
  if (bn_y == nullptr && (gamma == nullptr || beta == nullptr)) return (int)hipErrorInvalidValue;
  GemmParams p = {};
  p.amax_out = amax_out;
  p.A = dY; p.B = Wt; p.C = dZ; p.M = M; p.N = N; p.K = K;
  p.lda = K; p.ldb = N; p.ldc = N; p.alpha = 1.f;
  p.resid = resid; p.ldr = ldr;
  p.vecA = al16(dY) && (K % 4 == 0);
  p.vecB = al16(Wt) && (N % 4 == 0);
  p.fastA = p.vecA;
  p.fastB = p.vecB;
  const bool al = al16(dZ) && al16(bn_x) && (bn_y == nullptr || al16(bn_y)) && (resid == nullptr || (al16(resid) && ldr % 4 == 0)) &&
                  al16(mean) && al16(invstd) && (gamma == nullptr || al16(gamma)) && (beta == nullptr || al16(beta));
  if (!stats_eligible(p) || !al || N % 4 != 0) return 0;
  p.stat_part = part;
  p.bnb_x = bn_x; p.bnb_y = bn_y; p.bnb_mean = mean; p.bnb_invstd = invstd; p.bnb_gamma = gamma; p.bnb_beta = beta;
  *part_rows = cdiv(M, 128);
  H2Guard h2(p, h2n);
  if (h2.bad) return (int)hipErrorInvalidValue;
  if (h2n.armed && h2n.planes && (h2n.flags & 2)) {   // the weight arrives as planes of W^T [N][K]: the forward's operand kinds
    p.ldb = K;
    p.vecB = al16(Wt) && (K % 4 == 0);
    p.fastB = p.vecB;
    return launch_cfg<A_ROWK, B_NK>(p, 1, nullptr, 0, (hipStream_t)stream);
  }
  return launch_cfg<A_ROWK, B_KN>(p, 1, nullptr, 0, (hipStream_t)stream);
}

extern "C" int tris_conv3x3_fwd_bnstat_f32(const float* X, const float* Wt, float* Y, int B, int H, int W, int Cin,
                                           int Cout, int stride, double* stat_part, int* stat_rows, void* stream) {
  const H2Next h2n = h2_take();
  int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  GemmParams p = {};
  p.A = X; p.B = Wt; p.C = Y;
  p.M = B * Ho * Wo; p.N = Cout; p.K = 9 * Cin;
  p.ldb = 9L * Cin; p.ldc = Cout; p.alpha = 1.f;
  p.gH = H; p.gW = W; p.gC = Cin; p.gHo = Ho; p.gWo = Wo; p.gStride = stride; p.gB = B;
  p.vecA = al16(X) && (Cin % 16 == 0);
  p.vecB = al16(Wt) && ((9 * Cin) % 4 == 0);
  p.fastA = al16(X) && (Cin % 32 == 0);
  p.fastB = p.vecB;
  if (const int rows = run_stem_conv1(X, Wt, Y, B, H, W, Cin, Cout, stride, p.M >= 128 ? stat_part : nullptr, (hipStream_t)stream)) {
    *stat_rows = (rows > 0 && p.M >= 128 && stat_part != nullptr) ? rows : 0;
    return rows > 0 ? 0 : (int)hipErrorLaunchFailure;
  }
  p.stat_part = stats_eligible(p) ? stat_part : nullptr;
  H2Guard h2(p, h2n);
  if (h2.bad) return (int)hipErrorInvalidValue;
  return conv3_dispatch<B_NK>(p, (hipStream_t)stream, stat_rows);
}

// ---- BatchNorm + ReLU folded into the consuming 3x3 convolution ------------------------------------------------------------
// conv3x3(relu(bn(X))) without relu(bn(X)) ever existing in HBM: the direct kernels apply the normalisation while they stage
// their window (forward: gemm_fast.h A_HALO; weight gradient: wgrad3x3_direct_kernel).  Only where a direct kernel serves BOTH
// products -- tris_conv3x3_bnin_ok says so -- otherwise the caller materialises relu(bn(X)) with tris_bn_apply_f32 as before.
static bool bnin_enabled() {
  const int cd = g_conv_direct_thread >= 0 ? g_conv_direct_thread : g_opt.conv_direct;
  return (g_gemm_mode == 1 || g_gemm_mode == 3) && cd != 0 && g_opt.wgrad_direct != 0 && g_opt.bn_fold;
}
static GemmParams conv3_fwd_params(const float* X, const float* Wt, float* Y, int B, int H, int W, int Cin, int Cout) {
  GemmParams p = {};
  p.A = X; p.B = Wt; p.C = Y;
  p.M = B * H * W; p.N = Cout; p.K = 9 * Cin;
  p.ldb = 9L * Cin; p.ldc = Cout; p.alpha = 1.f;
  p.gH = H; p.gW = W; p.gC = Cin; p.gHo = H; p.gWo = W; p.gStride = 1; p.gB = B;
  p.vecA = al16(X) && (Cin % 16 == 0);
  p.vecB = al16(Wt) && ((9 * Cin) % 4 == 0);
  p.fastA = al16(X) && (Cin % 32 == 0);
  p.fastB = p.vecB;
  return p;
}
static int wg_static_choice(int B, int H, int W, int Cin, int Cout, long ws_bytes) {
  for (int id : {1, 2, 5})
    if (wg_ok(id, H, W, Cin, Cout) && wg_slices(id, B, H, W, Cin, Cout, ws_bytes, g_opt.wg_blocks) >= 1) return id;
  return 0;
}

extern "C" int tris_conv3x3_bnin_ok(int B, int H, int W, int Cin, int Cout) {
  if (!bnin_enabled() || Cin % 16 != 0 || Cout % 4 != 0) return 0;
  GemmParams p = conv3_fwd_params(reinterpret_cast<const float*>(16), reinterpret_cast<const float*>(16),
                                  reinterpret_cast<float*>(16), B, H, W, Cin, Cout);   // (shape test only: aligned dummies)
  return halo_shape_ok(p) && halo_static_choice(p, false) > 0 && wg_static_choice(B, H, W, Cin, Cout, 64L << 20) > 0;
}

extern "C" int tris_conv3x3_fwd_bnin_f32(const float* X, const float* mean, const float* invstd, const float* gamma,
                                         const float* beta, const float* Wt, float* Y, int B, int H, int W, int Cin, int Cout,
                                         double* stat_part, int* stat_rows, void* stream) {
  // (armed for h2: operand A of the arming is an UPPER BOUND of |relu(bn(X))| -- tris_bn_out_bound_f32 -- since that tensor never exists)
  const H2Next h2n = h2_take();
  GemmParams p = conv3_fwd_params(X, Wt, Y, B, H, W, Cin, Cout);
  H2Guard h2(p, h2n);
  if (h2.bad || g_gemm_mode == 4) return (int)hipErrorInvalidValue;   // (raw fp32 input, normalised while it is staged: no planes)
  if (!bnin_enabled() || !halo_shape_ok(p) || halo_static_choice(p, false) == 0 || !al16(mean) || !al16(invstd) || !al16(gamma) ||
      !al16(beta))
    return (int)hipErrorInvalidValue;   // ask tris_conv3x3_bnin_ok first
  p.in_mean = mean; p.in_invstd = invstd; p.in_gamma = gamma; p.in_beta = beta;
  p.stat_part = (stat_part != nullptr && p.fastA && p.M >= 128) ? stat_part : nullptr;
  if (stat_rows) *stat_rows = 0;
  return conv3_dispatch<B_NK>(p, (hipStream_t)stream, stat_rows, true);
}

extern "C" int tris_conv3x3_wgrad_bnin_f32(const float* X, const float* mean, const float* invstd, const float* gamma,
                                           const float* beta, const float* dY, float* dW, int B, int H, int W, int Cin, int Cout,
                                           float* workspace, long ws_bytes, void* stream) {
  const H2Next h2n = h2_take();   // (armed for h2: A = dY, B = the bound of |relu(bn(X))|)
  GemmParams pq = {};
  pq.M = Cout; pq.N = 9 * Cin; pq.K = B * H * W;
  pq.fastA = al16(dY) && (Cout % 4 == 0);
  pq.fastB = al16(X) && (Cin % 4 == 0);
  H2Guard h2(pq, h2n);
  if (h2.bad || g_gemm_mode == 4) return (int)hipErrorInvalidValue;   // (the folded form normalises fp32 input while staging it: no planes)
  if (!bnin_enabled() || workspace == nullptr || !al16(X) || !al16(dY) || !al16(dW) || !al16(workspace) || !al16(mean) ||
      !al16(invstd) || !al16(gamma) || !al16(beta))
    return (int)hipErrorInvalidValue;
  auto direct = [&](int id) {
    return run_wgrad_direct(id, X, dY, dW, B, H, W, Cin, Cout, workspace, ws_bytes, (hipStream_t)stream, BnIn{mean, invstd, gamma, beta},
                            pq.h2_amaxA, pq.h2_amaxB);
  };
  return wgrad_pick(B, H, W, Cin, Cout, ws_bytes, true, (hipStream_t)stream, direct, []() { return (int)hipErrorInvalidValue; });
}

extern "C" int tris_set_gemm_mode(int mode) {
  if (mode != 0 && mode != 1 && mode != 3) return (int)hipErrorInvalidValue;
  g_mode_default = mode;
  return 0;
}
extern "C" int tris_set_gemm_mode_thread(int mode) {
  if (mode != -1 && mode != 0 && mode != 1 && mode != 3) return (int)hipErrorInvalidValue;
  g_mode_thread = mode;
  return 0;
}
extern "C" int tris_set_autotune(int on) {
  g_autotune = on ? 1 : 0;
  return 0;
}

extern "C" int tris_get_gemm_mode(void) { return g_gemm_mode; }

extern "C" int tris_h2_next(const unsigned* amaxA, const unsigned* amaxB, float scaleA, float scaleB) {
  g_h2_next = H2Next{amaxA, amaxB, scaleA, scaleB, true, false, 0};
  return 0;
}
// the next dense product of the calling thread takes BOTH operands as fp16 piece planes (tris_h2_planes_f32 layout) scaled by the
// powers of two that *amaxA / *amaxB imply; flags bit 0: the bn_y operand of tris_gemm_bnbwd_f32 is such a plane tensor too
extern "C" int tris_h2_next_planes(const unsigned* amaxA, const unsigned* amaxB, int flags) {
  if (amaxA == nullptr || amaxB == nullptr) return (int)hipErrorInvalidValue;
  g_h2_next = H2Next{amaxA, amaxB, 0.f, 0.f, true, true, flags};
  return 0;
}

extern "C" long tris_direct_launches(int kind) { return (kind == 0 || kind == 1) ? g_direct_launches[kind] : -1; }

extern "C" int tris_set_option(const char* name, const char* value) {
  if (name == nullptr) return (int)hipErrorInvalidValue;
  if (!strncmp(name, "TRIS_", 5)) name += 5;
  return set_option(g_opt, name, value) ? 0 : (int)hipErrorInvalidValue;
}
extern "C" int tris_set_conv_direct_thread(int v) {
  g_conv_direct_thread = v < 0 ? -1 : v;
  return 0;
}
