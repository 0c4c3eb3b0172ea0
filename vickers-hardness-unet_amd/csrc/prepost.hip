// The steps either side of model(x) in the reference's inference wrappers, one fused
// pass each, on the device:
//
//   k_letterbox_pre  : uint8 BGR [h][w][3] -> cv2.resize(INTER_LINEAR) -> constant border to S x S -> BGR->RGB ->
//                      /255 -> (x - mean) / std -> float32 NCHW [3][S][S]
//                      (infer_pth_gui.py:17-24, 46-49; ui_infer_quadrilateral.py:197-216, 662-678)
//   k_letterbox_mask / k_letterbox_post<true>  : logits [S][S] -> (sigmoid >= thresh) * 255 -> crop ->
//                      cv2.resize(INTER_NEAREST) -> uint8 [h][w]                 (infer_pth_gui.py:50-53, 26-29)
//   k_letterbox_prob / k_letterbox_post<false> : logits [S][S] -> sigmoid -> crop -> cv2.resize(INTER_LINEAR, float32)
//                      (a copy when the crop already has the original size) -> clip [0,1] -> float32 [h][w]
//                      (ui_infer_quadrilateral.py:705-711, 219-231)
//   (first name: one thread per pixel, small outputs; second: 256 x 16 pixel blocks with the source window in LDS, large outputs)
//
// All are byte / HBM-bound gathers with lanes along x (coalesced stores, source reads of neighbouring lanes fall into the
// same or adjacent lines).  The interpolation arithmetic restates OpenCV's
// resize.cpp (see oracle/prepost_oracle.py for the formulae); floating-point contraction is switched off in this file so
// that products and sums round exactly as the scalar CPU code does.
#include <stdlib.h>

#include "vk_common.h"

#pragma clang fp contract(off)

namespace vk {

struct LbParams {
  int h, w, stride, S, nh, nw, top, left, pad;
  int window;                   // post-processing: stage the transformed source window in LDS (large outputs)
  double scale_x, scale_y;      // 1 / (dst / src) in double, computed on the host as cv::resize does
};

// cv::resize linear coordinates: index of the left/top sample and the float weight of the right/bottom one
__device__ __forceinline__ void lin_coord(int d, double scale, int src, int& s, float& f) {
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

__device__ __forceinline__ int fix11(float w) { return __float2int_rn(w * 2048.f); }   // saturate_cast<short>(w * INTER_RESIZE_COEF_SCALE)

__device__ __forceinline__ float sigmoidf(float x) {
#pragma clang fp contract(off)
  return 1.f / (1.f + expf(-x));
}

__global__ __launch_bounds__(256) void k_letterbox_pre(const LbParams p, const uint8_t* __restrict__ src, float* __restrict__ out) {
#pragma clang fp contract(off)
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= p.S || y >= p.S) return;
  const int dx = x - p.left, dy = y - p.top;
  int v[3] = {p.pad, p.pad, p.pad};
  if ((unsigned)dx < (unsigned)p.nw && (unsigned)dy < (unsigned)p.nh) {
    if (p.nh == p.h && p.nw == p.w) {
      const uint8_t* s = src + (size_t)dy * p.stride + dx * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = s[c];
    } else {
      int sx, sy;
      float fx, fy;
      lin_coord(dx, p.scale_x, p.w, sx, fx);
      lin_coord(dy, p.scale_y, p.h, sy, fy);
      const int a0 = fix11(1.f - fx), a1 = fix11(fx), b0 = fix11(1.f - fy), b1 = fix11(fy);
      const int sx1 = min(sx + 1, p.w - 1), sy1 = min(sy + 1, p.h - 1);
      const uint8_t* r0 = src + (size_t)sy * p.stride;
      const uint8_t* r1 = src + (size_t)sy1 * p.stride;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int h0 = (int)r0[sx * 3 + c] * a0 + (int)r0[sx1 * 3 + c] * a1;      // HResizeLinear, scale 2048
        const int h1 = (int)r1[sx * 3 + c] * a0 + (int)r1[sx1 * 3 + c] * a1;
        const int o = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;   // VResizeLinear<uchar>
        v[c] = min(max(o, 0), 255);
      }
    }
  }
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  const size_t plane = (size_t)p.S * p.S, o = (size_t)y * p.S + x;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float f = (float)v[2 - k] / 255.f;      // RGB plane k = BGR channel 2-k
    out[k * plane + o] = (f - mean[k]) / stdv[k];
  }
}

// Post-processing of small outputs (launch/latency-bound): one thread per destination pixel, every tap evaluated in place.
__global__ __launch_bounds__(256) void k_letterbox_mask(const LbParams p, const float* __restrict__ logits, float thresh,
                                                        uint8_t* __restrict__ mask) {
#pragma clang fp contract(off)
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= p.w || y >= p.h) return;
  // resizeNN from the nh x nw crop to h x w: scale here is 1 / (dst / src) with dst = original, src = crop
  int sx = x, sy = y;
  if (!(p.nh == p.h && p.nw == p.w)) {
    sx = min((int)floor((double)x * p.scale_x), p.nw - 1);
    sy = min((int)floor((double)y * p.scale_y), p.nh - 1);
  }
  const float l = logits[(size_t)(p.top + sy) * p.S + p.left + sx];
  mask[(size_t)y * p.w + x] = sigmoidf(l) >= thresh ? 255 : 0;
}

__global__ __launch_bounds__(256) void k_letterbox_prob(const LbParams p, const float* __restrict__ logits, float* __restrict__ prob) {
#pragma clang fp contract(off)
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= p.w || y >= p.h) return;
  const float* base = logits + (size_t)p.top * p.S + p.left;
  float o;
  if (p.nh == p.h && p.nw == p.w) {
    o = sigmoidf(base[(size_t)y * p.S + x]);
  } else {
    int sx, sy;
    float fx, fy;
    lin_coord(x, p.scale_x, p.nw, sx, fx);
    lin_coord(y, p.scale_y, p.nh, sy, fy);
    const int sx1 = min(sx + 1, p.nw - 1), sy1 = min(sy + 1, p.nh - 1);
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const float p00 = sigmoidf(base[(size_t)sy * p.S + sx]), p01 = sigmoidf(base[(size_t)sy * p.S + sx1]);
    const float p10 = sigmoidf(base[(size_t)sy1 * p.S + sx]), p11 = sigmoidf(base[(size_t)sy1 * p.S + sx1]);
    const float h0 = p00 * a0 + p01 * a1;
    const float h1 = p10 * a0 + p11 * a1;
    o = h0 * b0 + h1 * b1;
  }
  prob[(size_t)y * p.w + x] = fminf(fmaxf(o, 0.f), 1.f);
}

// Post-processing of large outputs, both flavours: a workgroup owns a 256 x 16 block of the ORIGINAL-size output (thread = 4 consecutive
// pixels in 4 rows).  The window of the logit map that block reads is transformed ONCE into LDS (sigmoid, or the
// thresholded 0/255 value), so the expf + division run once per source sample instead of once per tap of every
// destination pixel; windows above WIN samples (strong reductions) fall back to per-tap evaluation.  Results are the
// same numbers either way.
constexpr int PP_BW = 256, PP_BH = 16, PP_WIN = 6144;

template <bool MASK>
__global__ __launch_bounds__(256) void k_letterbox_post(const LbParams p, const float* __restrict__ logits, float thresh, void* __restrict__ outv) {
#pragma clang fp contract(off)
  __shared__ float win[PP_WIN];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bx0 = blockIdx.x * PP_BW, by0 = blockIdx.y * PP_BH;
  const int bx1 = min(bx0 + PP_BW, p.w) - 1, by1 = min(by0 + PP_BH, p.h) - 1;      // last pixel of this block (inclusive)
  const float* base = logits + (size_t)p.top * p.S + p.left;

  // source window of the block (coordinates are monotone in the destination index)
  int wx0, wx1, wy0, wy1;
  if (MASK) {
    wx0 = min((int)floor((double)bx0 * p.scale_x), p.nw - 1);
    wx1 = min((int)floor((double)bx1 * p.scale_x), p.nw - 1);
    wy0 = min((int)floor((double)by0 * p.scale_y), p.nh - 1);
    wy1 = min((int)floor((double)by1 * p.scale_y), p.nh - 1);
  } else {
    float f;
    lin_coord(bx0, p.scale_x, p.nw, wx0, f);
    lin_coord(bx1, p.scale_x, p.nw, wx1, f);
    lin_coord(by0, p.scale_y, p.nh, wy0, f);
    lin_coord(by1, p.scale_y, p.nh, wy1, f);
    wx1 = min(wx1 + 1, p.nw - 1);
    wy1 = min(wy1 + 1, p.nh - 1);
  }
  const int nx = wx1 - wx0 + 1, ny = wy1 - wy0 + 1;
  const bool windowed = p.window && nx * ny <= PP_WIN;         // workgroup-uniform
  auto value = [&](float l) -> float {
    const float s = sigmoidf(l);
    return MASK ? (s >= thresh ? 255.f : 0.f) : s;
  };
  if (windowed) {
    // thread = window column, eight rows requested before the first sigmoid (the fill is one memory round trip)
    for (int c = threadIdx.x; c < nx; c += 256) {
      const float* col = base + (size_t)wy0 * p.S + wx0 + c;
      for (int r0 = 0; r0 < ny; r0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = r0 + u < ny ? col[(size_t)(r0 + u) * p.S] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (r0 + u < ny) win[(r0 + u) * nx + c] = value(v[u]);
      }
    }
  }
  __syncthreads();
  auto tap = [&](int sy, int sx) -> float {
    return windowed ? win[(sy - wy0) * nx + (sx - wx0)] : value(base[(size_t)sy * p.S + sx]);
  };

  const int x0 = bx0 + lane * 4;
  if (x0 >= p.w) return;
  int sx[4], sx1[4];
  float fx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int x = min(x0 + j, p.w - 1);
    if (MASK) {
      sx[j] = min((int)floor((double)x * p.scale_x), p.nw - 1);
    } else {
      lin_coord(x, p.scale_x, p.nw, sx[j], fx[j]);
      sx1[j] = min(sx[j] + 1, p.nw - 1);
    }
  }
  const bool vec = (p.w & 3) == 0 && (reinterpret_cast<uintptr_t>(outv) & 15) == 0;      // x0 + 3 < w, aligned vector stores
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = by0 + wave * 4 + i;
    if (y >= p.h) break;
    if (MASK) {
      const int sy = min((int)floor((double)y * p.scale_y), p.nh - 1);
      uint8_t* out = reinterpret_cast<uint8_t*>(outv) + (size_t)y * p.w + x0;
      uint32_t m[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) m[j] = tap(sy, sx[j]) != 0.f ? 255u : 0u;
      if (vec) {
        *reinterpret_cast<uint32_t*>(out) = m[0] | (m[1] << 8) | (m[2] << 16) | (m[3] << 24);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (x0 + j < p.w) out[j] = (uint8_t)m[j];
      }
    } else {
      int sy;
      float fy;
      lin_coord(y, p.scale_y, p.nh, sy, fy);
      const int sy1 = min(sy + 1, p.nh - 1);
      const float b0 = 1.f - fy, b1 = fy;
      float* out = reinterpret_cast<float*>(outv) + (size_t)y * p.w + x0;
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a0 = 1.f - fx[j], a1 = fx[j];
        const float h0 = tap(sy, sx[j]) * a0 + tap(sy, sx1[j]) * a1;
        const float h1 = tap(sy1, sx[j]) * a0 + tap(sy1, sx1[j]) * a1;
        o[j] = fminf(fmaxf(h0 * b0 + h1 * b1, 0.f), 1.f);
      }
      if (vec) {
        *reinterpret_cast<f32x4_t*>(out) = f32x4_t{o[0], o[1], o[2], o[3]};
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (x0 + j < p.w) out[j] = o[j];
      }
    }
  }
}

// VK_PP_WINDOW=0/1 forces one of the two post-processing kernels (tests run every shape through both)
static bool use_window(size_t pixels, size_t from) {
  const char* e = getenv("VK_PP_WINDOW");
  return e ? atoi(e) != 0 : pixels >= from;
}

static int fill_params(const vk_letterbox_desc* d, LbParams& p, bool forward, const char* who) {
  VK_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  VK_CHECK_ARG(d->h > 0 && d->w > 0 && d->size > 0 && d->nh > 0 && d->nw > 0, "%s: non-positive size", who);
  VK_CHECK_ARG(d->top >= 0 && d->left >= 0 && d->top + d->nh <= d->size && d->left + d->nw <= d->size,
               "%s: the %dx%d resized image at (%d,%d) does not fit the %d-pixel square", who, d->nh, d->nw, d->top, d->left, d->size);
  VK_CHECK_ARG(d->h <= 16384 && d->w <= 16384 && d->size <= 16384, "%s: image side above 16384", who);
  VK_CHECK_ARG(d->pad_value >= 0 && d->pad_value <= 255, "%s: pad_value outside 0..255", who);
  p.h = d->h; p.w = d->w; p.stride = d->src_stride; p.S = d->size; p.nh = d->nh; p.nw = d->nw;
  p.top = d->top; p.left = d->left; p.pad = d->pad_value;
  p.window = 1;
  if (forward) {   // original -> resized
    p.scale_x = 1.0 / ((double)d->nw / (double)d->w);
    p.scale_y = 1.0 / ((double)d->nh / (double)d->h);
  } else {         // resized crop -> original
    p.scale_x = 1.0 / ((double)d->w / (double)d->nw);
    p.scale_y = 1.0 / ((double)d->h / (double)d->nh);
  }
  return VK_OK;
}

}  // namespace vk

using namespace vk;

extern "C" int vk_letterbox_preprocess(const vk_letterbox_desc* d, const uint8_t* bgr, float* x_nchw, void* stream) {
  LbParams p;
  int rc = fill_params(d, p, true, "vk_letterbox_preprocess");
  if (rc != VK_OK) return rc;
  VK_CHECK_ARG(bgr && x_nchw, "vk_letterbox_preprocess: null buffer");
  VK_CHECK_ARG(d->src_stride >= 3 * d->w, "vk_letterbox_preprocess: src_stride %d below 3*w", d->src_stride);
  const bool same = d->nh == d->h && d->nw == d->w;
  const double src_touched = same ? 3.0 * d->h * d->w : fmin(3.0 * d->h * d->w, 12.0 * d->nh * d->nw);
  vkh::ProfScope ps("letterbox_pre", (hipStream_t)stream, 0.0, src_touched + 12.0 * d->size * d->size);
  hipLaunchKernelGGL(k_letterbox_pre, dim3((d->size + 63) / 64, (d->size + 3) / 4), dim3(256), 0, (hipStream_t)stream, p, bgr, x_nchw);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_letterbox_postprocess_mask(const vk_letterbox_desc* d, const float* logits, float thresh, uint8_t* mask_hw, void* stream) {
  LbParams p;
  int rc = fill_params(d, p, false, "vk_letterbox_postprocess_mask");
  if (rc != VK_OK) return rc;
  VK_CHECK_ARG(logits && mask_hw, "vk_letterbox_postprocess_mask: null buffer");
  vkh::ProfScope ps("letterbox_mask", (hipStream_t)stream, 0.0, fmin(4.0 * d->nh * d->nw, 4.0 * d->h * d->w) + 1.0 * d->h * d->w);
  // small outputs are launch/latency-bound: one pixel per thread beats the load -> barrier -> gather structure (measured:
  // 1200x1600 5.3 vs 6.2 us, 2048x2048 8.7 vs 6.0 us)
  if (use_window((size_t)d->h * d->w, 3u << 20))
    hipLaunchKernelGGL(k_letterbox_post<true>, dim3((d->w + PP_BW - 1) / PP_BW, (d->h + PP_BH - 1) / PP_BH), dim3(256), 0, (hipStream_t)stream, p, logits, thresh, (void*)mask_hw);
  else
    hipLaunchKernelGGL(k_letterbox_mask, dim3((d->w + 63) / 64, (d->h + 3) / 4), dim3(256), 0, (hipStream_t)stream, p, logits, thresh, mask_hw);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_letterbox_postprocess_prob(const vk_letterbox_desc* d, const float* logits, float* prob_hw, void* stream) {
  LbParams p;
  int rc = fill_params(d, p, false, "vk_letterbox_postprocess_prob");
  if (rc != VK_OK) return rc;
  VK_CHECK_ARG(logits && prob_hw, "vk_letterbox_postprocess_prob: null buffer");
  vkh::ProfScope ps("letterbox_prob", (hipStream_t)stream, 0.0, 4.0 * d->nh * d->nw + 4.0 * d->h * d->w);
  // measured: 512x512 2.5 vs 8.9 us, 1200x1600 9.4 vs 7.6 us, 2048x2048 18.9 vs 11.5 us (per-pixel vs windowed)
  if (use_window((size_t)d->h * d->w, 1u << 20))
    hipLaunchKernelGGL(k_letterbox_post<false>, dim3((d->w + PP_BW - 1) / PP_BW, (d->h + PP_BH - 1) / PP_BH), dim3(256), 0, (hipStream_t)stream, p, logits, 0.f, (void*)prob_hw);
  else
    hipLaunchKernelGGL(k_letterbox_prob, dim3((d->w + 63) / 64, (d->h + 3) / 4), dim3(256), 0, (hipStream_t)stream, p, logits, prob_hw);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}
