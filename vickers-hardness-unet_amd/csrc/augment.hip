// Training-time augmentation on the device: the reference's VickersDataset.__getitem__ (train.py:173-200)
// runs cv2 + albumentations on the training thread (num_workers=0, train.py:586-589) — orders of magnitude slower than the
// 2,500 img/s training step.  Here the whole (183-image) dataset lives in HBM as letterboxed uint8 tensors and one fused kernel per
// batch produces the network input:
//
//   vk_letterbox_u8 / vk_letterbox_mask_u8 : once per image — LongestMaxSize + PadIfNeeded of train.py:70-75 (geometry chosen by the
//        host, convention "train"): uint8 BGR [h][w][3] -> uint8 RGB [S][S][3] (cv2 8-bit INTER_LINEAR) and mask [h][w] ->
//        {0,1} [S][S] (INTER_NEAREST of (m > 0), train.py:150-168)
//   vk_augment_batch : per step — for each of the n samples, with draws made on the host (vk_aug_params):
//        OneOf(HorizontalFlip, VerticalFlip, RandomRotate90)   train.py:81-85   exact pixel permutation, folded into the tap fetch
//        Rotate(limit=180, BORDER_CONSTANT)                    train.py:89      inverse-mapped bilinear (image) / nearest (mask) gather
//        OneOf(RandomBrightnessContrast, CLAHE, GaussianBlur)  train.py:96-100 LUT arithmetic / tile-histogram equalisation of L / binomial blur
//        GaussNoise                                            train.py:104     counter-based hash -> Irwin-Hall normal deviate
//        Normalize + ToTensorV2                                train.py:108-112 float32 [3][S][S], mask float32 [1][S][S]
//      in ONE pass: 64 x 4 pixel tiles; the geometric result of the tile (+ a 2-pixel apron when the sample is blurred) is
//      staged in LDS as uint8, everything after it is per-pixel.  HBM-bound byte work: ~3-12 source bytes read (L2-resident
//      neighbours) and 16 bytes written per pixel; no MFMA.
//      CLAHE is the one transform that needs the whole image before any pixel: samples that drew it get one extra pass first
//      (k_clahe_tiles: one workgroup per 8 x 8 tile — geometric stage, RGB -> L*a*b* in integer fixed point, 256-bin histogram
//      in LDS, OpenCV's clip / redistribute / cdf -> tile LUT), and k_augment interpolates the four neighbouring LUTs and goes
//      back to RGB.
// Arithmetic follows oracle/augment_oracle.py operation by operation (contraction off): outputs are bit-identical to it.
#include <stdlib.h>

#include "vk_common.h"

#pragma clang fp contract(off)

namespace vk {

struct AugParams {      // = vk_aug_params
  int d4, rotate;
  float cos_a, sin_a;
  int photo;
  float alpha, beta;
  int blur_ksize;
  float noise_scale;
  uint32_t noise_seed;
  int clahe_limit;
};
static_assert(sizeof(AugParams) == sizeof(vk_aug_params), "vk_aug_params layout");

// pixel (y, x) of the D4-transformed image = pixel (sy, sx) of the stored one
__device__ __forceinline__ void d4_src(int d4, int S, int y, int x, int& sy, int& sx) {
  switch (d4) {
    case 1: sy = y; sx = S - 1 - x; break;               // HorizontalFlip
    case 2: sy = S - 1 - y; sx = x; break;               // VerticalFlip
    case 4: sy = x; sx = S - 1 - y; break;               // np.rot90(k=1): out[i][j] = in[j][S-1-i]
    case 5: sy = S - 1 - y; sx = S - 1 - x; break;       // k = 2
    case 6: sy = S - 1 - x; sx = y; break;               // k = 3: out[i][j] = in[S-1-j][i]
    default: sy = y; sx = x; break;                      // 0 and 3 (k = 0)
  }
}

struct Px { int r, g, b, m; };

// geometric stage for output pixel (y, x): D4 + optional rotation; uint8 values
__device__ __forceinline__ Px aug_geom(const AugParams& p, int S, const uint8_t* __restrict__ img, const uint8_t* __restrict__ msk, int y, int x) {
#pragma clang fp contract(off)
  Px o;
  if (!p.rotate) {
    int sy, sx;
    d4_src(p.d4, S, y, x, sy, sx);
    const uint8_t* s = img + ((size_t)sy * S + sx) * 3;
    o.r = s[0]; o.g = s[1]; o.b = s[2];
    o.m = msk[(size_t)sy * S + sx];
    return o;
  }
  const float c = (float)S * 0.5f - 0.5f;
  const float dx = (float)x - c, dy = (float)y - c;
  const float fx_ = (p.cos_a * dx - p.sin_a * dy) + c;
  const float fy_ = (p.sin_a * dx + p.cos_a * dy) + c;
  const float x0f = floorf(fx_), y0f = floorf(fy_);
  const float wx = fx_ - x0f, wy = fy_ - y0f;
  const int x0 = (int)x0f, y0 = (int)y0f;
  float t[4][3];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int yy = y0 + (q >> 1), xx = x0 + (q & 1);
    if ((unsigned)yy < (unsigned)S && (unsigned)xx < (unsigned)S) {
      int sy, sx;
      d4_src(p.d4, S, yy, xx, sy, sx);
      const uint8_t* s = img + ((size_t)sy * S + sx) * 3;
      t[q][0] = (float)s[0]; t[q][1] = (float)s[1]; t[q][2] = (float)s[2];
    } else {
      t[q][0] = t[q][1] = t[q][2] = 0.f;                 // BORDER_CONSTANT, value 0
    }
  }
  const float w0x = 1.f - wx, w0y = 1.f - wy;
  int v[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float top = t[0][ch] * w0x + t[1][ch] * wx;
    const float bot = t[2][ch] * w0x + t[3][ch] * wx;
    const float val = top * w0y + bot * wy;
    v[ch] = min(max((int)rintf(val), 0), 255);
  }
  o.r = v[0]; o.g = v[1]; o.b = v[2];
  const int xi = (int)floorf(fx_ + 0.5f), yi = (int)floorf(fy_ + 0.5f);
  o.m = 0;
  if ((unsigned)yi < (unsigned)S && (unsigned)xi < (unsigned)S) {
    int sy, sx;
    d4_src(p.d4, S, yi, xi, sy, sx);
    o.m = msk[(size_t)sy * S + sx];
  }
  return o;
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ int reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

constexpr int AUG_TW = 64, AUG_TH = 4, AUG_R = 2;       // tile and blur apron

// ---- CLAHE (train.py:98): 8-bit RGB <-> L*a*b* in integer fixed point (oracle/augment_oracle.py: rgb_to_lab_u8 / lab_to_rgb_u8)
constexpr int TAB_LIN = 0, TAB_FT = 256, TAB_ENC = 256 + 4097;
static_assert(TAB_ENC + 4097 == VK_AUG_TABLE_INTS, "colour table layout");

__device__ __forceinline__ long long floordiv(long long a, long long b) {      // b > 0; Python's //
  long long q = a / b;
  return (a % b < 0) ? q - 1 : q;
}
__device__ __forceinline__ int clampi(long long v, int lo, int hi) { return (int)(v < lo ? lo : (v > hi ? hi : v)); }

__device__ __forceinline__ void rgb_to_lab(const int* __restrict__ tab, int r, int g, int b, int& L, int& A, int& B) {
  const int lr = tab[TAB_LIN + r], lg = tab[TAB_LIN + g], lb = tab[TAB_LIN + b];
  const int tx = clampi((1777 * lr + 1541 * lg + 778 * lb + 2048) >> 12, 0, 4096);
  const int ty = clampi((871 * lr + 2929 * lg + 296 * lb + 2048) >> 12, 0, 4096);
  const int tz = clampi((73 * lr + 448 * lg + 3575 * lb + 2048) >> 12, 0, 4096);
  const long long fx = tab[TAB_FT + tx], fy = tab[TAB_FT + ty], fz = tab[TAB_FT + tz];
  L = clampi(floordiv(116ll * fy * 255 - 16ll * 255 * 32768 + 50ll * 32768, 100ll * 32768), 0, 255);
  A = clampi(((500 * (fx - fy) + 16384) >> 15) + 128, 0, 255);
  B = clampi(((200 * (fy - fz) + 16384) >> 15) + 128, 0, 255);
}

__device__ __forceinline__ int lab_finv(long long f) {                           // t x 4096 from f x 32768
  const long long cube = (f * f * f + (1ll << 32)) >> 33;
  const long long low = floordiv((f * 116 - 16ll * 32768) * 27 * 4096 + (24389ll * 32768) / 2, 24389ll * 32768);
  return clampi(f > 6780 ? cube : low, 0, 8192);
}

__device__ __forceinline__ void lab_to_rgb(const int* __restrict__ tab, int L, int A, int B, int& r, int& g, int& b) {
  const long long fy = ((long long)(L * 100 * 32768 + 127) / 255 + 16 * 32768 + 58) / 116;
  const long long fx = fy + floordiv((long long)(A - 128) * 32768 + 250, 500);
  const long long fz = fy - floordiv((long long)(B - 128) * 32768 + 100, 200);
  const long long tx = lab_finv(fx), ty = lab_finv(fy), tz = lab_finv(fz);
  r = tab[TAB_ENC + clampi((12615 * tx - 6296 * ty - 2223 * tz + 2048) >> 12, 0, 4096)];
  g = tab[TAB_ENC + clampi((-3773 * tx + 7684 * ty + 185 * tz + 2048) >> 12, 0, 4096)];
  b = tab[TAB_ENC + clampi((217 * tx - 836 * ty + 4715 * tz + 2048) >> 12, 0, 4096)];
}

// One workgroup per (tile, CLAHE sample): geometric stage -> L, a, b, mask of the tile's pixels into the workspace, 256-bin
// histogram of L in LDS, then OpenCV's clahe.cpp per-tile LUT (clip at `limit`, excess / 256 to every bin + the residual to every
// (256 / residual)-th bin, lut = cvRound(cdf * 255 / tileArea) in float32).  grid (8, 8, n); samples without CLAHE leave at once.
__global__ __launch_bounds__(256) void k_clahe_tiles(int S, int n_items, const uint8_t* __restrict__ images, const uint8_t* __restrict__ masks,
                                                     const int* __restrict__ index, const AugParams* __restrict__ params,
                                                     const int* __restrict__ tab, uint8_t* __restrict__ ws) {
#pragma clang fp contract(off)
  const int n = blockIdx.z;
  const AugParams p = params[n];
  if (p.photo != 2) return;
  __shared__ int hist[256];
  __shared__ int scan[256];
  const int item = min(max(index[n], 0), n_items - 1);
  const uint8_t* img = images + (size_t)item * S * S * 3;
  const uint8_t* msk = masks + (size_t)item * S * S;
  const size_t per = (size_t)S * S * 4 + 64 * 256;
  uint32_t* lab = (uint32_t*)(ws + (size_t)n * per);
  uint8_t* lut = ws + (size_t)n * per + (size_t)S * S * 4 + (size_t)(blockIdx.y * 8 + blockIdx.x) * 256;
  const int ts = S >> 3, y0 = blockIdx.y * ts, x0 = blockIdx.x * ts;
  hist[threadIdx.x] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < ts * ts; i += 256) {
    const int ly = i / ts, lx = i - ly * ts;
    const Px q = aug_geom(p, S, img, msk, y0 + ly, x0 + lx);
    int L, A, B;
    rgb_to_lab(tab, q.r, q.g, q.b, L, A, B);
    lab[(size_t)(y0 + ly) * S + x0 + lx] = (uint32_t)L | ((uint32_t)A << 8) | ((uint32_t)B << 16) | ((uint32_t)q.m << 24);
    atomicAdd(&hist[L], 1);
  }
  __syncthreads();
  const int t = threadIdx.x;
  int h = hist[t];
  const int limit = p.clahe_limit;
  scan[t] = max(h - limit, 0);
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off) scan[t] += scan[t + off];
    __syncthreads();
  }
  const int clipped = scan[0];
  __syncthreads();
  h = min(h, limit) + (clipped >> 8);
  const int residual = clipped & 255;
  if (residual) {
    const int step = max(256 / residual, 1);
    if (t % step == 0 && t / step < residual) ++h;
  }
  scan[t] = h;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {                // inclusive scan (integers: order-free)
    const int v = t >= off ? scan[t - off] : 0;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  const float scale = 255.f / (float)(ts * ts);
  lut[t] = (uint8_t)min(max((int)rintf((float)scan[t] * scale), 0), 255);
}

// the four-LUT interpolation of clahe.cpp for pixel (y, x) with value v: tile coordinates x / tileW - 0.5, float32, fixed order
__device__ __forceinline__ int clahe_interp(const uint8_t* __restrict__ lut, int S, int y, int x, int v) {
#pragma clang fp contract(off)
  const float inv = 1.f / (float)(S >> 3);
  const float txf = (float)x * inv - 0.5f, tyf = (float)y * inv - 0.5f;
  const float tx1f = floorf(txf), ty1f = floorf(tyf);
  const float xa = txf - tx1f, ya = tyf - ty1f;
  const float xa1 = 1.f - xa, ya1 = 1.f - ya;
  int tx1 = (int)tx1f, ty1 = (int)ty1f;
  const int tx2 = min(tx1 + 1, 7), ty2 = min(ty1 + 1, 7);
  tx1 = max(tx1, 0); ty1 = max(ty1, 0);
  const float l11 = (float)lut[(ty1 * 8 + tx1) * 256 + v], l12 = (float)lut[(ty1 * 8 + tx2) * 256 + v];
  const float l21 = (float)lut[(ty2 * 8 + tx1) * 256 + v], l22 = (float)lut[(ty2 * 8 + tx2) * 256 + v];
  const float top = l11 * xa1 + l12 * xa;
  const float bot = l21 * xa1 + l22 * xa;
  const float res = top * ya1 + bot * ya;
  return min(max((int)rintf(res), 0), 255);
}

__global__ __launch_bounds__(256) void k_augment(int S, int n_items, const uint8_t* __restrict__ images, const uint8_t* __restrict__ masks,
                                                 const int* __restrict__ index, const AugParams* __restrict__ params,
                                                 const int* __restrict__ tab, const uint8_t* __restrict__ ws, float* __restrict__ xo,
                                                 float* __restrict__ yo) {
#pragma clang fp contract(off)
  const int n = blockIdx.z;
  const AugParams p = params[n];
  const int item = min(max(index[n], 0), n_items - 1);      // a bad index can never leave the dataset
  const uint8_t* img = images + (size_t)item * S * S * 3;
  const uint8_t* msk = masks + (size_t)item * S * S;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int x = blockIdx.x * AUG_TW + tx, y = blockIdx.y * AUG_TH + ty;
  const bool in = x < S && y < S;
  int r = 0, g = 0, b = 0, m = 0;
  if (p.photo == 3) {
    // blurred sample: geometric result of the tile + apron into LDS (uint8 x 3), then the binomial filter
    constexpr int LW = AUG_TW + 2 * AUG_R, LH = AUG_TH + 2 * AUG_R;
    __shared__ uint8_t tile[LH][LW][4];
    for (int i = threadIdx.x; i < LW * LH; i += 256) {
      const int ly = i / LW, lx = i - ly * LW;
      const int gy = reflect101(blockIdx.y * AUG_TH + ly - AUG_R, S), gx = reflect101(blockIdx.x * AUG_TW + lx - AUG_R, S);
      const bool ok = (unsigned)gy < (unsigned)S && (unsigned)gx < (unsigned)S;      // tiles hanging over the right / bottom edge
      Px q{0, 0, 0, 0};
      if (ok) q = aug_geom(p, S, img, msk, gy, gx);
      tile[ly][lx][0] = (uint8_t)q.r; tile[ly][lx][1] = (uint8_t)q.g; tile[ly][lx][2] = (uint8_t)q.b; tile[ly][lx][3] = (uint8_t)q.m;
    }
    __syncthreads();
    if (!in) return;
    const int k = p.blur_ksize, rad = k >> 1;
    // binomial weights without a runtime-indexed local array (that would live in scratch memory): [1 2 1] / [1 4 6 4 1]
    auto bw = [k](int i) { return k == 3 ? (i == 1 ? 2 : 1) : (i == 2 ? 6 : ((i == 1 || i == 3) ? 4 : 1)); };
    int acc[3] = {0, 0, 0};
    for (int i = 0; i < k; ++i) {
      const int wy = bw(i);
      for (int j = 0; j < k; ++j) {
        const int w = wy * bw(j);
        // the apron was filled through reflect101 of the GLOBAL coordinate, so plain LDS offsets are already border-correct —
        // except that a reflected coordinate must be reflected about the image edge, not about the tile: recompute when near it
        const int gy = reflect101(y + i - rad, S), gx = reflect101(x + j - rad, S);
        const int ly = gy - (int)(blockIdx.y * AUG_TH) + AUG_R, lx = gx - (int)(blockIdx.x * AUG_TW) + AUG_R;
        const uint8_t* t = tile[ly][lx];
        acc[0] += w * t[0]; acc[1] += w * t[1]; acc[2] += w * t[2];
      }
    }
    const int tot = k == 3 ? 16 : 256;
    r = (acc[0] + tot / 2) / tot; g = (acc[1] + tot / 2) / tot; b = (acc[2] + tot / 2) / tot;
    m = tile[ty + AUG_R][tx + AUG_R][3];
  } else if (p.photo == 2) {
    // CLAHE: k_clahe_tiles left L, a, b, mask of the geometric result and the 64 tile LUTs of this sample in the workspace
    if (!in) return;
    const size_t per = (size_t)S * S * 4 + 64 * 256;
    const uint32_t q = ((const uint32_t*)(ws + (size_t)n * per))[(size_t)y * S + x];
    const int L = clahe_interp(ws + (size_t)n * per + (size_t)S * S * 4, S, y, x, (int)(q & 255));
    lab_to_rgb(tab, L, (int)((q >> 8) & 255), (int)((q >> 16) & 255), r, g, b);
    m = (int)(q >> 24);
  } else {
    if (!in) return;
    const Px q = aug_geom(p, S, img, msk, y, x);
    r = q.r; g = q.g; b = q.b; m = q.m;
    if (p.photo == 1) {                                  // RandomBrightnessContrast: lut[v] = trunc(clip(v * alpha + beta * 255))
      const float off = p.beta * 255.f;
      r = (int)fminf(fmaxf((float)r * p.alpha + off, 0.f), 255.f);
      g = (int)fminf(fmaxf((float)g * p.alpha + off, 0.f), 255.f);
      b = (int)fminf(fmaxf((float)b * p.alpha + off, 0.f), 255.f);
    }
  }
  if (p.noise_scale > 0.f) {
    int v[3] = {r, g, b};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const unsigned long long cnt = ((unsigned long long)y * S + x) * 3ull + ch;
      const unsigned long long base = (((unsigned long long)p.noise_seed << 32) | cnt) * 3ull;
      int isum = 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const unsigned long long z = splitmix64(base + j);
        isum += (int)(z & 0xFFFF) + (int)((z >> 16) & 0xFFFF) + (int)((z >> 32) & 0xFFFF) + (int)(z >> 48);
      }
      const float gn = (float)(isum - 393210) * p.noise_scale;
      v[ch] = (int)fminf(fmaxf((float)v[ch] + gn, 0.f), 255.f);
    }
    r = v[0]; g = v[1]; b = v[2];
  }
  // Normalize (albumentations: (v - mean * 255) * (1 / (std * 255)), float32) + CHW
  const float mean255[3] = {0.485f * 255.f, 0.456f * 255.f, 0.406f * 255.f};
  const float inv[3] = {1.f / (0.229f * 255.f), 1.f / (0.224f * 255.f), 1.f / (0.225f * 255.f)};
  const size_t plane = (size_t)S * S, o = (size_t)y * S + x;
  float* xb = xo + (size_t)n * 3 * plane;
  xb[o] = ((float)r - mean255[0]) * inv[0];
  xb[plane + o] = ((float)g - mean255[1]) * inv[1];
  xb[2 * plane + o] = ((float)b - mean255[2]) * inv[2];
  yo[(size_t)n * plane + o] = (float)m;
}

// ---- one-off letterboxing of the dataset into uint8 (the cv2 8-bit INTER_LINEAR arithmetic of prepost.hip's k_letterbox_pre)
struct LbU8 {
  int h, w, stride, S, nh, nw, top, left, pad;
  double scale_x, scale_y;
};

__device__ __forceinline__ void lin_coord_u8(int d, double scale, int src, int& s, float& f) {
#pragma clang fp contract(off)
  const double t = ((double)d + 0.5) * scale;
  float fv = (float)(t - 0.5);
  int sv = (int)floorf(fv);
  fv -= (float)sv;
  if (sv < 0) { fv = 0.f; sv = 0; }
  if (sv >= src - 1) { fv = 0.f; sv = src - 1; }
  s = sv;
  f = fv;
}

__global__ __launch_bounds__(256) void k_letterbox_u8(const LbU8 p, const uint8_t* __restrict__ src, uint8_t* __restrict__ out) {
#pragma clang fp contract(off)
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= p.S || y >= p.S) return;
  const int dx = x - p.left, dy = y - p.top;
  int v[3] = {p.pad, p.pad, p.pad};
  if ((unsigned)dx < (unsigned)p.nw && (unsigned)dy < (unsigned)p.nh) {
    if (p.nh == p.h && p.nw == p.w) {
      const uint8_t* s = src + (size_t)dy * p.stride + dx * 3;
      v[0] = s[0]; v[1] = s[1]; v[2] = s[2];
    } else {
      int sx, sy;
      float fx, fy;
      lin_coord_u8(dx, p.scale_x, p.w, sx, fx);
      lin_coord_u8(dy, p.scale_y, p.h, sy, fy);
      const int a0 = __float2int_rn((1.f - fx) * 2048.f), a1 = __float2int_rn(fx * 2048.f);
      const int b0 = __float2int_rn((1.f - fy) * 2048.f), b1 = __float2int_rn(fy * 2048.f);
      const int sx1 = min(sx + 1, p.w - 1), sy1 = min(sy + 1, p.h - 1);
      const uint8_t* r0 = src + (size_t)sy * p.stride;
      const uint8_t* r1 = src + (size_t)sy1 * p.stride;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int h0 = (int)r0[sx * 3 + c] * a0 + (int)r0[sx1 * 3 + c] * a1;
        const int h1 = (int)r1[sx * 3 + c] * a0 + (int)r1[sx1 * 3 + c] * a1;
        const int o = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        v[c] = min(max(o, 0), 255);
      }
    }
  }
  uint8_t* o = out + ((size_t)y * p.S + x) * 3;
  o[0] = (uint8_t)v[2]; o[1] = (uint8_t)v[1]; o[2] = (uint8_t)v[0];      // BGR -> RGB (train.py:149)
}

__global__ __launch_bounds__(256) void k_letterbox_mask_u8(const LbU8 p, const uint8_t* __restrict__ src, uint8_t* __restrict__ out) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= p.S || y >= p.S) return;
  const int dx = x - p.left, dy = y - p.top;
  int v = 0;                                               // PadIfNeeded: mask border value 0
  if ((unsigned)dx < (unsigned)p.nw && (unsigned)dy < (unsigned)p.nh) {
    int sx = dx, sy = dy;
    if (!(p.nh == p.h && p.nw == p.w)) {                   // cv2 INTER_NEAREST: floor(dst * scale), clamped
      sx = min((int)floor((double)dx * p.scale_x), p.w - 1);
      sy = min((int)floor((double)dy * p.scale_y), p.h - 1);
    }
    v = src[(size_t)sy * p.stride + sx] > 0 ? 1 : 0;       // train.py:166  m = (m > 0)
  }
  out[(size_t)y * p.S + x] = (uint8_t)v;
}

static int fill_u8(const vk_letterbox_desc* d, LbU8& p, int min_stride, const char* who) {
  VK_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  VK_CHECK_ARG(d->h > 0 && d->w > 0 && d->size > 0 && d->nh > 0 && d->nw > 0, "%s: non-positive size", who);
  VK_CHECK_ARG(d->top >= 0 && d->left >= 0 && d->top + d->nh <= d->size && d->left + d->nw <= d->size,
               "%s: the %dx%d resized image at (%d,%d) does not fit the %d-pixel square", who, d->nh, d->nw, d->top, d->left, d->size);
  VK_CHECK_ARG(d->h <= 16384 && d->w <= 16384 && d->size <= 16384, "%s: image side above 16384", who);
  VK_CHECK_ARG(d->pad_value >= 0 && d->pad_value <= 255, "%s: pad_value outside 0..255", who);
  VK_CHECK_ARG(d->src_stride >= min_stride, "%s: src_stride %d below %d", who, d->src_stride, min_stride);
  p.h = d->h; p.w = d->w; p.stride = d->src_stride; p.S = d->size; p.nh = d->nh; p.nw = d->nw;
  p.top = d->top; p.left = d->left; p.pad = d->pad_value;
  p.scale_x = 1.0 / ((double)d->nw / (double)d->w);
  p.scale_y = 1.0 / ((double)d->nh / (double)d->h);
  return VK_OK;
}

}  // namespace vk

using namespace vk;

extern "C" int vk_letterbox_u8(const vk_letterbox_desc* d, const uint8_t* bgr, uint8_t* rgb_sq, void* stream) {
  LbU8 p;
  int rc = fill_u8(d, p, d ? 3 * d->w : 0, "vk_letterbox_u8");
  if (rc != VK_OK) return rc;
  VK_CHECK_ARG(bgr && rgb_sq, "vk_letterbox_u8: null buffer");
  hipLaunchKernelGGL(k_letterbox_u8, dim3((d->size + 63) / 64, (d->size + 3) / 4), dim3(256), 0, (hipStream_t)stream, p, bgr, rgb_sq);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_letterbox_mask_u8(const vk_letterbox_desc* d, const uint8_t* mask_hw, uint8_t* mask_sq, void* stream) {
  LbU8 p;
  int rc = fill_u8(d, p, d ? d->w : 0, "vk_letterbox_mask_u8");
  if (rc != VK_OK) return rc;
  VK_CHECK_ARG(mask_hw && mask_sq, "vk_letterbox_mask_u8: null buffer");
  hipLaunchKernelGGL(k_letterbox_mask_u8, dim3((d->size + 63) / 64, (d->size + 3) / 4), dim3(256), 0, (hipStream_t)stream, p, mask_hw, mask_sq);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" size_t vk_augment_workspace_bytes(int n, int size) {
  if (n < 1 || size < 8) return 0;
  return (size_t)n * ((size_t)size * size * 4 + 64 * 256);
}

extern "C" int vk_augment_batch(int n, int size, int n_items, const uint8_t* images_rgb, const uint8_t* masks, const int* index_dev,
                                const vk_aug_params* params_host, void* params_dev, const int* color_tables, void* workspace,
                                size_t workspace_bytes, float* x, float* y, void* stream) {
  VK_CHECK_ARG(n >= 1 && n <= 65535 && size >= 8 && size <= 16384 && n_items >= 1, "vk_augment_batch: bad batch / size / item count");
  VK_CHECK_ARG(images_rgb && masks && index_dev && params_host && params_dev && x && y, "vk_augment_batch: null buffer");
  int n_clahe = 0;
  for (int i = 0; i < n; ++i) {
    const vk_aug_params& p = params_host[i];
    VK_CHECK_ARG(p.d4 >= 0 && p.d4 <= 6, "vk_augment_batch: sample %d: d4 %d outside 0..6", i, p.d4);
    VK_CHECK_ARG(p.photo >= 0 && p.photo <= 3, "vk_augment_batch: sample %d: photo %d outside 0..3", i, p.photo);
    if (p.photo == 2) {
      ++n_clahe;
      VK_CHECK_ARG(p.clahe_limit >= 1, "vk_augment_batch: sample %d: clahe_limit %d must be >= 1", i, p.clahe_limit);
    }
    VK_CHECK_ARG(p.photo != 3 || p.blur_ksize == 3 || p.blur_ksize == 5, "vk_augment_batch: sample %d: blur_ksize %d must be 3 or 5", i, p.blur_ksize);
    VK_CHECK_ARG(!p.rotate || fabsf(p.cos_a * p.cos_a + p.sin_a * p.sin_a - 1.f) < 1e-3f, "vk_augment_batch: sample %d: (cos, sin) not a rotation", i);
    VK_CHECK_ARG(p.noise_scale >= 0.f && p.noise_scale < 1.f, "vk_augment_batch: sample %d: noise_scale %g", i, (double)p.noise_scale);
  }
  if (n_clahe) {
    VK_CHECK_ARG(size % 8 == 0, "vk_augment_batch: CLAHE needs size %% 8 == 0 (8 x 8 tiles), got %d", size);
    VK_CHECK_ARG(color_tables && workspace, "vk_augment_batch: %d sample(s) draw CLAHE but color_tables / workspace is null", n_clahe);
    VK_CHECK_ARG(workspace_bytes >= vk_augment_workspace_bytes(n, size), "vk_augment_batch: workspace of %zu bytes, need %zu", workspace_bytes,
                 vk_augment_workspace_bytes(n, size));
  }
  hipStream_t st = (hipStream_t)stream;
  VK_CHECK_HIP(hipMemcpyAsync(params_dev, params_host, (size_t)n * sizeof(vk_aug_params), hipMemcpyHostToDevice, st));
  vkh::ProfScope ps("augment", st, 0.0, (double)n * size * size * (4.0 + 16.0) + (double)n_clahe * size * size * (4.0 + 4.0 + 4.0));
  if (n_clahe)
    hipLaunchKernelGGL(k_clahe_tiles, dim3(8, 8, n), dim3(256), 0, st, size, n_items, images_rgb, masks, index_dev, (const AugParams*)params_dev,
                       color_tables, (uint8_t*)workspace);
  hipLaunchKernelGGL(k_augment, dim3((size + AUG_TW - 1) / AUG_TW, (size + AUG_TH - 1) / AUG_TH, n), dim3(256), 0, st, size, n_items, images_rgb, masks,
                     index_dev, (const AugParams*)params_dev, color_tables, (const uint8_t*)workspace, x, y);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}
