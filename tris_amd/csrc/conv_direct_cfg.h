// Configurations of the direct 3x3 convolution kernels, shared by the dispatcher (gemm_conv.hip) and the launchers
// (conv_direct.hip).  Included inside each unit's anonymous namespace, after gemm_params.h.
#pragma once

struct HaloCfg { int bm, bn, nw, nwm, hs, hmode; };
static const HaloCfg kHalo[] = {
    {0, 0, 0, 0, 0, 0},             // 0: the implicit GEMM (A_IM2COL)
    {256, 128, 8, 4, 324, 2},       // 1: 16 x 16 patches, wide outputs (W, H % 16 == 0: the 80 x 80 stages)
    {256, 64, 4, 4, 324, 2},        // 2:                  64 output channels (layer1, stem conv3)
    {256, 32, 4, 4, 324, 2},        // 3:                  32 output channels (stem conv2, stem data gradients)
    {128, 128, 4, 2, 324, 1},       // 4: 128 consecutive pixels, padded coordinates (W <= 40)
    {256, 128, 8, 4, 452, 1},       // 5: 256 consecutive pixels
    {128, 128, 4, 2, 180, 2},       // 6: 8 x 16 patches (W % 16 == 0, H % 8 == 0)
};
constexpr int kHaloN = 7;

static bool halo_ok(const GemmParams& p, int id) {
  const HaloCfg& h = kHalo[id];
  if (h.bn == 128 && p.N <= 64) return false;
  if (h.bn == 64 && (p.N <= 32 || p.N > 64)) return false;
  if (h.bn == 32 && p.N > 32) return false;
  if (p.M < h.bm) return false;
  if (h.hmode == 2) return p.gW % 16 == 0 && p.gH % (h.bm / 16) == 0;
  const int pitch = p.gW + 2;   // largest window of a tile: its pixels, the row / image padding they cross, one row either side
  const long slots = (h.bm - 1) + 2L * cdiv(h.bm - 1, p.gW) + 2L * pitch * cdiv(h.bm - 1, p.gH * p.gW) + 2L * pitch + 3;
  return slots <= h.hs;
}
static int halo_tiles_m(const GemmParams& p, int id) {
  const HaloCfg& h = kHalo[id];
  return h.hmode == 2 ? p.gB * (p.gH / (h.bm / 16)) * (p.gW / 16) : cdiv(p.M, h.bm);
}

struct WgCfg { int cot, cit, wk, r; };
static const WgCfg kWg[] = {{0, 0, 0, 0}, {32, 32, 4, 4}, {64, 32, 2, 4}, {64, 64, 1, 2}, {64, 64, 1, 4}, {64, 32, 2, 8}};
static const int kWgXW[] = {0, 16, 16, 16, 16, 8};
constexpr int kWgN = 6;
static bool wg_ok(int id, int H, int W, int Ci, int Co) {
  const WgCfg& c = kWg[id];
  return W % kWgXW[id] == 0 && H % c.r == 0 && Co % c.cot == 0 && Ci % c.cit == 0 && (id >= 2 || (Co == c.cot && Ci == c.cit));
}
// slices: enough blocks for ~two per CU (`target` blocks in all), bounded by the workspace
static int wg_slices(int id, int B, int H, int W, int Ci, int Co, long ws_bytes, int target = 512) {
  const WgCfg& c = kWg[id];
  const int tiles = (Co / c.cot) * (Ci / c.cit);
  const int units = B * (H / c.r) * (W / kWgXW[id]);
  long s = std::max(1, target / tiles);
  s = std::min<long>(s, std::max(1, units / 8));
  const long per = (long)Co * 9 * Ci * (long)sizeof(float);
  if (s * per > ws_bytes) s = ws_bytes / per;
  return (int)s;
}
struct BnIn { const float *mean, *invstd, *gamma, *beta; };
