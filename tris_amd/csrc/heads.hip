// Stage-1 head kernels (gfx950): L2 row normalise, row softmax, the training cls head (bg-augmented per-pixel
// softmax + mean/max pooling + focal term), diagonal response maps with the x32 bilinear upsample, generic bilinear
// resize (both align_corners conventions), CLIP-guided foreground patches, and the contrastive losses.
// All tiny, latency/HBM-bound: one workgroup per image / row group, wavefront shuffles for the row reductions.
#include "common.h"
#include "tris_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------------ l2norm
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ X, float* __restrict__ Y,
                                                         float* __restrict__ inv, long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) { float v = X[r * C + c]; s += v * v; }
  float iv = 1.0f / sqrtf(wave_sum(s));  // no eps, as the reference (model_stage1.py:68-69)
  for (int c = lane; c < C; c += 64) Y[r * C + c] = X[r * C + c] * iv;
  if (lane == 0) inv[r] = iv;
}
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                         const float* __restrict__ inv, float* __restrict__ dX,
                                                         long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += Y[r * C + c] * dY[r * C + c];
  s = wave_sum(s);
  float iv = inv[r];
  for (int c = lane; c < C; c += 64) dX[r * C + c] = (dY[r * C + c] - Y[r * C + c] * s) * iv;
}

// ------------------------------------------------------------------------------------------------------ softmax
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ X, float* __restrict__ Y, long rows,
                                                          int n, float scale) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float m = -INFINITY;
  for (int c = lane; c < n; c += 64) m = fmaxf(m, X[r * n + c] * scale);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < n; c += 64) s += expf(X[r * n + c] * scale - m);
  s = wave_sum(s);
  for (int c = lane; c < n; c += 64) Y[r * n + c] = expf(X[r * n + c] * scale - m) / s;
}
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                          float* __restrict__ dX, long rows, int n, float scale) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float s = 0.f;
  for (int c = lane; c < n; c += 64) s += Y[r * n + c] * dY[r * n + c];
  s = wave_sum(s);
  for (int c = lane; c < n; c += 64) dX[r * n + c] = scale * Y[r * n + c] * (dY[r * n + c] - s);
}

// ------------------------------------------------------------------------------------------------------ cls head
// score [B, P, N] (image, pixel, sentence).  One workgroup per image.  model_stage1.py:80-108 restated per image:
// channels = {bg == 1} U {N sentences}; softmax over channels per pixel; per sentence: mean+max of logits over
// pixels + focal(mean prob).  LDS: sc[P][N+1], pm[P] (row max), pz[P] (row partition sum), aux[P].
__global__ __launch_bounds__(256) void cls_head_fwd_kernel(const float* __restrict__ score, float* __restrict__ cls_out,
                                                           float* __restrict__ cls_fg, int P, int N, float focal_p,
                                                           float focal_c) {
  extern __shared__ float sm[];
  const int NP = N + 1;
  float* sc = sm;
  float* pm = sc + P * NP;
  float* pz = pm + P;
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* s = score + (long)i * P * N;
  for (int idx = tid; idx < P * N; idx += 256) { int p = idx / N, j = idx - p * N; sc[p * NP + j] = s[idx]; }
  __syncthreads();
  for (int p = wv; p < P; p += 4) {
    float m = 1.0f;
    for (int j = lane; j < N; j += 64) m = fmaxf(m, sc[p * NP + j]);
    m = wave_max(m);
    float z = 0.f;
    for (int j = lane; j < N; j += 64) z += expf(sc[p * NP + j] - m);
    z = wave_sum(z) + expf(1.0f - m);
    if (lane == 0) { pm[p] = m; pz[p] = z; }
  }
  __syncthreads();
  for (int j = wv; j < N; j += 4) {
    float sum = 0.f, mx = -INFINITY, pr = 0.f;
    for (int p = lane; p < P; p += 64) {
      float v = sc[p * NP + j];
      sum += v;
      mx = fmaxf(mx, v);
      pr += expf(v - pm[p]) / pz[p];
    }
    sum = wave_sum(sum);
    mx = wave_max(mx);
    pr = wave_sum(pr) / (float)P;
    if (lane == 0) {
      float c2 = powf(1.0f - pr, focal_p) * logf(focal_c + pr);
      cls_out[(long)i * N + j] = sum / (float)P + mx + c2;
      if (j == i && cls_fg) cls_fg[i] = pr;
    }
  }
}

// dscore (overwritten, all entries) from g = dL/dcls_out.
__global__ __launch_bounds__(256) void cls_head_bwd_kernel(const float* __restrict__ score, const float* __restrict__ g,
                                                           float* __restrict__ dscore, int P, int N, float focal_p,
                                                           float focal_c) {
  extern __shared__ float sm[];
  const int NP = N + 1;
  float* sc = sm;            // logits, later probabilities
  float* pm = sc + P * NP;
  float* pz = pm + P;
  float* tp = pz + P;        // t_p = sum_j a_j prob[p][j]
  float* aj = tp + P;        // a_j
  float* gj = aj + N;        // g_ij
  int* am = (int*)(gj + N);  // argmax pixel per sentence
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* s = score + (long)i * P * N;
  for (int idx = tid; idx < P * N; idx += 256) { int p = idx / N, j = idx - p * N; sc[p * NP + j] = s[idx]; }
  __syncthreads();
  for (int p = wv; p < P; p += 4) {
    float m = 1.0f;
    for (int j = lane; j < N; j += 64) m = fmaxf(m, sc[p * NP + j]);
    m = wave_max(m);
    float z = 0.f;
    for (int j = lane; j < N; j += 64) z += expf(sc[p * NP + j] - m);
    z = wave_sum(z) + expf(1.0f - m);
    if (lane == 0) { pm[p] = m; pz[p] = z; }
  }
  __syncthreads();
  for (int j = wv; j < N; j += 4) {
    float mx = -INFINITY, pr = 0.f;
    int arg = 0x7fffffff;
    for (int p = lane; p < P; p += 64) {
      float v = sc[p * NP + j];
      if (v > mx) { mx = v; arg = p; }
      pr += expf(v - pm[p]) / pz[p];
    }
    // first arg-max across lanes (ties -> smallest pixel index, as torch.max on CPU)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float om = __shfl_xor(mx, o, 64);
      int oa = __shfl_xor(arg, o, 64);
      if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
    }
    pr = wave_sum(pr) / (float)P;
    if (lane == 0) {
      float gg = g[(long)i * N + j];
      float om = 1.0f - pr;
      float d2 = -focal_p * powf(om, focal_p - 1.0f) * logf(focal_c + pr) + powf(om, focal_p) / (focal_c + pr);
      aj[j] = gg * d2 / (float)P;
      gj[j] = gg;
      am[j] = arg;
    }
  }
  __syncthreads();
  // logits -> probabilities in place
  for (int idx = tid; idx < P * N; idx += 256) {
    int p = idx / N, j = idx - p * N;
    sc[p * NP + j] = expf(sc[p * NP + j] - pm[p]) / pz[p];
  }
  __syncthreads();
  for (int p = wv; p < P; p += 4) {
    float t = 0.f;
    for (int j = lane; j < N; j += 64) t += aj[j] * sc[p * NP + j];
    t = wave_sum(t);
    if (lane == 0) tp[p] = t;
  }
  __syncthreads();
  float* d = dscore + (long)i * P * N;
  const float ip = 1.0f / (float)P;
  for (int idx = tid; idx < P * N; idx += 256) {
    int p = idx / N, j = idx - p * N;
    float v = gj[j] * ip + (am[j] == p ? gj[j] : 0.f) + sc[p * NP + j] * (aj[j] - tp[p]);
    d[idx] = v;
  }
}

// ------------------------------------------------------------------------------------------------------ bilinear
// PyTorch source-index conventions (upsample_bilinear2d): align_corners=False: src = (dst+0.5)*in/out-0.5 clamped
// at 0; True: src = dst*(in-1)/(out-1).
struct Lerp { int i0, i1; float w1; };
__device__ __forceinline__ Lerp lerp_of(int dst, int in, int out, int align) {
  float src;
  if (align) {
    float sc = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    src = sc * (float)dst;
  } else {
    float sc = (float)in / (float)out;
    src = sc * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
  }
  Lerp l;
  l.i0 = (int)src;
  if (l.i0 > in - 1) l.i0 = in - 1;
  l.i1 = l.i0 + (l.i0 < in - 1 ? 1 : 0);
  l.w1 = src - (float)l.i0;
  return l;
}
// weight with which input index `y` contributes to output index `dst`
__device__ __forceinline__ float contrib(int dst, int in, int out, int align, int y) {
  Lerp l = lerp_of(dst, in, out, align);
  float w = 0.f;
  if (l.i0 == y) w += 1.0f - l.w1;
  if (l.i1 == y) w += l.w1;
  return w;
}
// conservative output window [lo, hi] that can reference input index y
__device__ __forceinline__ void window(int y, int in, int out, int align, int& lo, int& hi) {
  if (in == 1) { lo = 0; hi = out - 1; return; }
  float a, b;
  if (align) {
    float inv = out > 1 ? (float)(out - 1) / (float)(in - 1) : 0.f;
    a = ((float)y - 1.f) * inv;
    b = ((float)y + 1.f) * inv;
  } else {
    float r = (float)out / (float)in;
    a = ((float)y - 0.5f) * r - 0.5f;
    b = ((float)y + 1.5f) * r - 0.5f;
  }
  lo = (int)floorf(a) - 1;
  hi = (int)ceilf(b) + 1;
  if (y == 0 || lo < 0) lo = 0;
  if (y == in - 1 || hi > out - 1) hi = out - 1;
}

__global__ void resize_fwd_kernel(const float* __restrict__ X, float* __restrict__ Y, int planes, int Hi, int Wi, int Ho,
                                  int Wo, int align) {
  long n = (long)planes * Ho * Wo;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int ox = (int)(i % Wo);
    long t = i / Wo;
    int oy = (int)(t % Ho);
    long pl = t / Ho;
    Lerp ly = lerp_of(oy, Hi, Ho, align), lx = lerp_of(ox, Wi, Wo, align);
    const float* p = X + pl * Hi * Wi;
    float v00 = p[ly.i0 * Wi + lx.i0], v01 = p[ly.i0 * Wi + lx.i1], v10 = p[ly.i1 * Wi + lx.i0],
          v11 = p[ly.i1 * Wi + lx.i1];
    float top = v00 + lx.w1 * (v01 - v00), bot = v10 + lx.w1 * (v11 - v10);
    Y[i] = top + ly.w1 * (bot - top);
  }
}
// gather form: one thread per INPUT pixel, loops the (small) window of outputs that reference it
__global__ void resize_bwd_kernel(const float* __restrict__ dY, float* __restrict__ dX, int planes, int Hi, int Wi,
                                  int Ho, int Wo, int align) {
  long n = (long)planes * Hi * Wi;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int x = (int)(i % Wi);
  long t = i / Wi;
  int y = (int)(t % Hi);
  long pl = t / Hi;
  int ylo, yhi, xlo, xhi;
  window(y, Hi, Ho, align, ylo, yhi);
  window(x, Wi, Wo, align, xlo, xhi);
  const float* g = dY + pl * Ho * Wo;
  float acc = 0.f;
  for (int oy = ylo; oy <= yhi; ++oy) {
    float wy = contrib(oy, Hi, Ho, align, y);
    if (wy == 0.f) continue;
    float row = 0.f;
    for (int ox = xlo; ox <= xhi; ++ox) row += contrib(ox, Wi, Wo, align, x) * g[(long)oy * Wo + ox];
    acc += wy * row;
  }
  dX[i] = acc;
}

// ------------------------------------------------------------------------------------------------------ response maps
// diag d[i][p] = score[i][p][i] (model_stage1.py:110-113) -> bilinear (align False) -> relu / sigmoid (:116-119).
__global__ __launch_bounds__(256) void maps_fwd_kernel(const float* __restrict__ score, float* __restrict__ relu_map,
                                                       float* __restrict__ sig_map, int h, int w, int N, int S) {
  __shared__ float d[1024];
  const int i = blockIdx.y, P = h * w;
  for (int p = threadIdx.x; p < P; p += 256) d[p] = score[((long)i * P + p) * N + i];
  __syncthreads();
  long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (o >= (long)S * S) return;
  int oy = (int)(o / S), ox = (int)(o - (long)oy * S);
  Lerp ly = lerp_of(oy, h, S, 0), lx = lerp_of(ox, w, S, 0);
  float v00 = d[ly.i0 * w + lx.i0], v01 = d[ly.i0 * w + lx.i1], v10 = d[ly.i1 * w + lx.i0], v11 = d[ly.i1 * w + lx.i1];
  float top = v00 + lx.w1 * (v01 - v00), bot = v10 + lx.w1 * (v11 - v10);
  float v = top + ly.w1 * (bot - top);
  long off = (long)i * S * S + o;
  relu_map[off] = fmaxf(v, 0.f);
  if (sig_map) sig_map[off] = 1.0f / (1.0f + expf(-v));
}
// one workgroup per (input pixel, image): dscore[i][p][i] += sum over the output window
__global__ __launch_bounds__(256) void maps_bwd_kernel(const float* __restrict__ score, const float* __restrict__ d_relu,
                                                       const float* __restrict__ d_sig, float* __restrict__ dscore,
                                                       int h, int w, int N, int S) {
  __shared__ float d[1024];
  __shared__ float red[4];
  const int i = blockIdx.y, P = h * w, p0 = blockIdx.x;
  for (int p = threadIdx.x; p < P; p += 256) d[p] = score[((long)i * P + p) * N + i];
  __syncthreads();
  const int y = p0 / w, x = p0 - y * w;
  int ylo, yhi, xlo, xhi;
  window(y, h, S, 0, ylo, yhi);
  window(x, w, S, 0, xlo, xhi);
  const int ww = xhi - xlo + 1, wh = yhi - ylo + 1;
  float acc = 0.f;
  for (int idx = threadIdx.x; idx < ww * wh; idx += 256) {
    int oy = ylo + idx / ww, ox = xlo + idx % ww;
    float wgt = contrib(oy, h, S, 0, y) * contrib(ox, w, S, 0, x);
    if (wgt == 0.f) continue;
    Lerp ly = lerp_of(oy, h, S, 0), lx = lerp_of(ox, w, S, 0);
    float v00 = d[ly.i0 * w + lx.i0], v01 = d[ly.i0 * w + lx.i1], v10 = d[ly.i1 * w + lx.i0],
          v11 = d[ly.i1 * w + lx.i1];
    float top = v00 + lx.w1 * (v01 - v00), bot = v10 + lx.w1 * (v11 - v10);
    float v = top + ly.w1 * (bot - top);
    long off = (long)i * S * S + (long)oy * S + ox;
    float gs = 0.f;
    if (d_relu && v > 0.f) gs += d_relu[off];
    if (d_sig) { float sg = 1.0f / (1.0f + expf(-v)); gs += d_sig[off] * sg * (1.0f - sg); }
    acc += wgt * gs;
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) dscore[((long)i * P + p0) * N + i] += acc;
}

// ------------------------------------------------------------------------------------------------------ fg patches
// patches[b][py*G+px][c*ps*ps + ky*ps + kx] = cam[b][y][x] * img[b][c][y][x]   (train_stage1.py:333-338 + the k32s32
// patch conv of the aux ViT, CLIP/clip/model.py:405, expressed as a GEMM operand)
__global__ void fg_patch_fwd_kernel(const float* __restrict__ cam, const float* __restrict__ img,
                                    float* __restrict__ patches, int B, int C, int R, int ps) {
  long n = (long)B * C * R * R;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  const int G = R / ps;
  for (; i < n; i += stride) {
    int x = (int)(i % R);
    long t = i / R;
    int y = (int)(t % R);
    t /= R;
    int c = (int)(t % C);
    int b = (int)(t / C);
    float v = cam[((long)b * R + y) * R + x] * img[i];
    int py = y / ps, ky = y - py * ps, px = x / ps, kx = x - px * ps;
    patches[(((long)b * G * G + py * G + px) * C + c) * ps * ps + ky * ps + kx] = v;
  }
}
__global__ void fg_patch_bwd_kernel(const float* __restrict__ dpatches, const float* __restrict__ img,
                                    float* __restrict__ dcam, int B, int C, int R, int ps) {
  long n = (long)B * R * R;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int G = R / ps;
  int x = (int)(i % R);
  long t = i / R;
  int y = (int)(t % R);
  int b = (int)(t / R);
  int py = y / ps, ky = y - py * ps, px = x / ps, kx = x - px * ps;
  float acc = 0.f;
  for (int c = 0; c < C; ++c)
    acc += dpatches[(((long)b * G * G + py * G + px) * C + c) * ps * ps + ky * ps + kx] *
           img[(((long)b * C + c) * R + y) * R + x];
  dcam[i] = acc;
}

// x[b][0] = cls + pos[0]; x[b][1+p] = emb[b][p] + pos[1+p]   (CLIP/clip/model.py:423-424)
__global__ void vit_assemble_kernel(const float* __restrict__ emb, const float* __restrict__ cls,
                                    const float* __restrict__ pos, float* __restrict__ x, int B, int T, int W) {
  long n = (long)B * T * W;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int c = (int)(i % W);
  long t = i / W;
  int tok = (int)(t % T);
  int b = (int)(t / T);
  float v = tok == 0 ? cls[c] : emb[((long)b * (T - 1) + tok - 1) * W + c];
  x[i] = v + pos[(long)tok * W + c];
}
// demb[b][p] = dx[b][1+p]
__global__ void vit_assemble_bwd_kernel(const float* __restrict__ dx, float* __restrict__ demb, int B, int T, int W) {
  long n = (long)B * (T - 1) * W;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int c = (int)(i % W);
  long t = i / W;
  int p = (int)(t % (T - 1));
  int b = (int)(t / (T - 1));
  demb[i] = dx[((long)b * T + p + 1) * W + c];
}

// ------------------------------------------------------------------------------------------------------ losses
// One workgroup per image.  out[b] = {cos_pos, fg term, cbs term}  (train_stage1.py:263-284, 340-353)
//   fg term  = -log(clamp(cos_pos, 1e-4, 0.9999));  cbs term = mean_k -log(1 - cos(fi, neg_k))
__global__ __launch_bounds__(256) void clip_loss_fwd_kernel(const float* __restrict__ fi, const float* __restrict__ ft,
                                                            const float* __restrict__ fneg, float* __restrict__ out,
                                                            int E, int K) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* a = fi + (long)b * E;
  const float* t = ft + (long)b * E;
  float saa = 0.f, stt = 0.f, sat = 0.f;
  for (int c = tid; c < E; c += 256) { float x = a[c], y = t[c]; saa += x * x; stt += y * y; sat += x * y; }
  saa = block_sum_256(saa, red);
  stt = block_sum_256(stt, red);
  sat = block_sum_256(sat, red);
  float na = sqrtf(saa);
  float cosp = sat / (na * sqrtf(stt));
  float cbs = 0.f;
  for (int k = 0; k < K; ++k) {
    const float* nk = fneg + ((long)b * K + k) * E;
    float snn = 0.f, san = 0.f;
    for (int c = tid; c < E; c += 256) { float y = nk[c]; snn += y * y; san += a[c] * y; }
    snn = block_sum_256(snn, red);
    san = block_sum_256(san, red);
    cbs += -logf(1.0f - san / (na * sqrtf(snn)));
  }
  if (tid == 0) {
    out[b * 3 + 0] = cosp;
    out[b * 3 + 1] = -logf(fminf(fmaxf(cosp, 0.0001f), 0.9999f));
    out[b * 3 + 2] = K > 0 ? cbs / (float)K : 0.f;
  }
}
// d fi given the upstream gradient g[4] of (total, l1, l4, l5) in device memory: dL/dl1 = g[0]*w1 + g[1], dL/dl5 = g[0]*w5 + g[3]
__global__ __launch_bounds__(256) void clip_loss_bwd_kernel(const float* __restrict__ fi, const float* __restrict__ ft,
                                                            const float* __restrict__ fneg, const float* __restrict__ g,
                                                            float w1, float w5, float* __restrict__ dfi, int B, int E, int K) {
  extern __shared__ float sm[];  // dhat[E]
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* a = fi + (long)b * E;
  const float* t = ft + (long)b * E;
  const float g1 = __fadd_rn(__fmul_rn(g[0], w1), g[1]), g5 = __fadd_rn(__fmul_rn(g[0], w5), g[3]);
  float saa = 0.f, stt = 0.f, sat = 0.f;
  for (int c = tid; c < E; c += 256) { float x = a[c], y = t[c]; saa += x * x; stt += y * y; sat += x * y; }
  saa = block_sum_256(saa, red);
  stt = block_sum_256(stt, red);
  sat = block_sum_256(sat, red);
  const float ia = 1.0f / sqrtf(saa), it = 1.0f / sqrtf(stt);
  const float cosp = sat * ia * it;
  const float c1 = (cosp > 0.0001f && cosp < 0.9999f) ? -g1 / ((float)B * cosp) : 0.f;  // clamp: zero grad outside
  for (int c = tid; c < E; c += 256) sm[c] = c1 * t[c] * it;
  for (int k = 0; k < K; ++k) {
    const float* nk = fneg + ((long)b * K + k) * E;
    float snn = 0.f, san = 0.f;
    for (int c = tid; c < E; c += 256) { float y = nk[c]; snn += y * y; san += a[c] * y; }
    snn = block_sum_256(snn, red);
    san = block_sum_256(san, red);
    const float in = 1.0f / sqrtf(snn);
    const float s = san * ia * in;
    const float ck = g5 / ((float)B * (float)K * (1.0f - s));
    for (int c = tid; c < E; c += 256) sm[c] += ck * nk[c] * in;
  }
  float dot = 0.f;
  for (int c = tid; c < E; c += 256) dot += sm[c] * a[c] * ia;
  dot = block_sum_256(dot, red);
  for (int c = tid; c < E; c += 256) dfi[(long)b * E + c] = (sm[c] - a[c] * ia * dot) * ia;
}

// multilabel_soft_margin_loss(x [B,N], eye) rows -> rowloss[b]; bwd elementwise  (train_stage1.py:354)
__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
__global__ void msm_fwd_kernel(const float* __restrict__ x, float* __restrict__ rowloss, int B, int N) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int j = 0; j < N; ++j) {
    float v = x[(long)b * N + j];
    s += (j == b) ? log_sigmoid(v) : log_sigmoid(-v);
  }
  rowloss[b] = -s / (float)N;
}
__global__ void msm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, float w4, float* __restrict__ dx,
                               int B, int N) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * N) return;
  int b = i / N, j = i - b * N;
  float sg = 1.0f / (1.0f + expf(-x[i]));
  const float g4 = __fadd_rn(__fmul_rn(g[0], w4), g[2]);   // dL/dl4
  dx[i] = g4 * (sg - (j == b ? 1.0f : 0.f)) / ((float)B * (float)N);
}
// losses[0..3] = total, l1, l4, l5 from per-image partials (fixed order => deterministic)
__global__ void loss_finalize_kernel(const float* __restrict__ per_img, const float* __restrict__ rowloss, int B,
                                     float w1, float w4, float w5, float* __restrict__ losses) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float l1 = 0.f, l4 = 0.f, l5 = 0.f;
  for (int b = 0; b < B; ++b) { l1 += per_img[b * 3 + 1]; l5 += per_img[b * 3 + 2]; l4 += rowloss[b]; }
  l1 /= (float)B; l4 /= (float)B; l5 /= (float)B;
  losses[0] = w1 * l1 + w4 * l4 + w5 * l5;
  losses[1] = l1;
  losses[2] = l4;
  losses[3] = l5;
}

inline int grid_for(long n, int block = 256) {
  long g = (n + block - 1) / block;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int tris_l2norm_fwd_f32(const float* X, float* Y, float* inv_norm, long rows, int C, void* stream) {
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, X, Y, inv_norm, rows, C);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_l2norm_bwd_f32(const float* dY, const float* Y, const float* inv_norm, float* dX, long rows, int C,
                                   void* stream) {
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, dY, Y, inv_norm, dX,
                     rows, C);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_softmax_fwd_f32(const float* X, float* Y, long rows, int n, float scale, void* stream) {
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, X, Y, rows, n, scale);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_softmax_bwd_f32(const float* dY, const float* Y, float* dX, long rows, int n, float scale,
                                    void* stream) {
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, dY, Y, dX, rows, n,
                     scale);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_cls_head_fwd_f32(const float* score, float* cls_out, float* cls_fg, int B, int P, int N,
                                     float focal_p, float focal_c, void* stream) {
  size_t lds = (size_t)(P * (N + 1) + 2 * P) * sizeof(float);
  if (lds > 150 * 1024) return (int)hipErrorInvalidValue;
  static bool big_ok = false;  // more than the default 64 KB of dynamic LDS (a 20 x 20 ViT grid with 48 sentences: 82 KB)
  if (lds > 60000 && !big_ok) {
    hipError_t e = hipFuncSetAttribute((const void*)cls_head_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return (int)e;
    big_ok = true;
  }
  hipLaunchKernelGGL(cls_head_fwd_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, score, cls_out, cls_fg, P, N,
                     focal_p, focal_c);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_cls_head_bwd_f32(const float* score, const float* g, float* dscore, int B, int P, int N,
                                     float focal_p, float focal_c, void* stream) {
  size_t lds = (size_t)(P * (N + 1) + 3 * P + 3 * N) * sizeof(float);
  if (lds > 150 * 1024) return (int)hipErrorInvalidValue;
  static bool big_ok = false;
  if (lds > 60000 && !big_ok) {
    hipError_t e = hipFuncSetAttribute((const void*)cls_head_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return (int)e;
    big_ok = true;
  }
  hipLaunchKernelGGL(cls_head_bwd_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, score, g, dscore, P, N, focal_p,
                     focal_c);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_maps_fwd_f32(const float* score, float* relu_map, float* sig_map, int B, int h, int w, int N, int S,
                                 void* stream) {
  if (h * w > 1024) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(maps_fwd_kernel, dim3(cdiv((long)S * S, 256), B), dim3(256), 0, (hipStream_t)stream, score,
                     relu_map, sig_map, h, w, N, S);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_maps_bwd_f32(const float* score, const float* d_relu, const float* d_sig, float* dscore, int B,
                                 int h, int w, int N, int S, void* stream) {
  if (h * w > 1024) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(maps_bwd_kernel, dim3(h * w, B), dim3(256), 0, (hipStream_t)stream, score, d_relu, d_sig, dscore, h,
                     w, N, S);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_resize_bilinear_fwd_f32(const float* X, float* Y, int planes, int Hi, int Wi, int Ho, int Wo,
                                            int align_corners, void* stream) {
  hipLaunchKernelGGL(resize_fwd_kernel, dim3(grid_for((long)planes * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, X, Y,
                     planes, Hi, Wi, Ho, Wo, align_corners);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_resize_bilinear_bwd_f32(const float* dY, float* dX, int planes, int Hi, int Wi, int Ho, int Wo,
                                            int align_corners, void* stream) {
  hipLaunchKernelGGL(resize_bwd_kernel, dim3(cdiv((long)planes * Hi * Wi, 256)), dim3(256), 0, (hipStream_t)stream, dY,
                     dX, planes, Hi, Wi, Ho, Wo, align_corners);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_fg_patch_fwd_f32(const float* cam, const float* img, float* patches, int B, int C, int R, int ps,
                                     void* stream) {
  hipLaunchKernelGGL(fg_patch_fwd_kernel, dim3(grid_for((long)B * C * R * R)), dim3(256), 0, (hipStream_t)stream, cam,
                     img, patches, B, C, R, ps);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_fg_patch_bwd_f32(const float* dpatches, const float* img, float* dcam, int B, int C, int R, int ps,
                                     void* stream) {
  hipLaunchKernelGGL(fg_patch_bwd_kernel, dim3(cdiv((long)B * R * R, 256)), dim3(256), 0, (hipStream_t)stream, dpatches,
                     img, dcam, B, C, R, ps);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_vit_assemble_fwd_f32(const float* emb, const float* cls, const float* pos, float* x, int B, int T,
                                         int W, void* stream) {
  hipLaunchKernelGGL(vit_assemble_kernel, dim3(cdiv((long)B * T * W, 256)), dim3(256), 0, (hipStream_t)stream, emb, cls,
                     pos, x, B, T, W);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_vit_assemble_bwd_f32(const float* dx, float* demb, int B, int T, int W, void* stream) {
  hipLaunchKernelGGL(vit_assemble_bwd_kernel, dim3(cdiv((long)B * (T - 1) * W, 256)), dim3(256), 0, (hipStream_t)stream,
                     dx, demb, B, T, W);
  TRIS_LAUNCH_CHECK();
  return 0;
}
extern "C" int tris_stage1_loss_fwd_f32(const float* cls, const float* fi, const float* ft, const float* fneg, int B,
                                        int N, int E, int K, float w1, float w4, float w5, float* per_img,
                                        float* rowloss, float* losses, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(clip_loss_fwd_kernel, dim3(B), dim3(256), 0, st, fi, ft, fneg, per_img, E, K);
  TRIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(msm_fwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, st, cls, rowloss, B, N);
  TRIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, per_img, rowloss, B, w1, w4, w5, losses);
  TRIS_LAUNCH_CHECK();
  return 0;
}
// g = device pointer to the upstream gradient of losses[4] = (total, l1, l4, l5); the loss weights fold it into dL/dl1, dL/dl4, dL/dl5
extern "C" int tris_stage1_loss_bwd_f32(const float* cls, const float* fi, const float* ft, const float* fneg,
                                        const float* g, float w1, float w4, float w5, int B, int N, int E, int K,
                                        float* dcls, float* dfi, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(clip_loss_bwd_kernel, dim3(B), dim3(256), (size_t)E * sizeof(float), st, fi, ft, fneg, g, w1, w5, dfi, B,
                     E, K);
  TRIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(msm_bwd_kernel, dim3(cdiv((long)B * N, 256)), dim3(256), 0, st, cls, g, w4, dcls, B, N);
  TRIS_LAUNCH_CHECK();
  return 0;
}
