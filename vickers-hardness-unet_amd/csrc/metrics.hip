// Validation metrics of the reference on the device: thresholded Dice / IoU per image, batch mean
// (reference: train.py:230-255 `dice_coef`, :259-281 `iou_coef`, called from `validate` train.py:518-522).
// The reference forms a 0/1 prediction tensor, two products and five reductions with torch ops; here ONE pass reads the
// probabilities (or logits) and the target once and leaves three sums per image, a second tiny launch turns them into
// the two batch means.  HBM-bound: 8 B per pixel.
#include "vk_common.h"

namespace vk {

// sums[img][0] = sum pred*t, [1] = sum pred, [2] = sum t over the image; pred = (p > thr) as 0/1.
// For 0/1 targets every partial is an exact integer in fp64, so the atomics commute and the result is bit-reproducible.
template <bool FROM_LOGITS>
__global__ __launch_bounds__(256) void k_seg_counts(size_t per_image, const float* __restrict__ p, const float* __restrict__ t,
                                                    float thr, double* __restrict__ sums, int vec_ok) {
  const int img = blockIdx.y;
  const float* pi = p + (size_t)img * per_image;
  const float* ti = t + (size_t)img * per_image;
  float inter = 0.f, psum = 0.f, tsum = 0.f;      // exact below 2^24 per thread: a thread sees per_image / (grid * 256) elements
  auto one = [&](float pv, float tv) {
    if (FROM_LOGITS) pv = 1.f / (1.f + expf(-pv));
    const float hit = pv > thr ? 1.f : 0.f;
    inter += hit * tv;
    psum += hit;
    tsum += tv;
  };
  const size_t gtid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  const size_t n4 = vec_ok ? per_image / 4 : 0;
  for (size_t i = gtid; i < n4; i += stride) {
    const f32x4_t pv = *reinterpret_cast<const f32x4_t*>(pi + 4 * i), tv = *reinterpret_cast<const f32x4_t*>(ti + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) one(pv[e], tv[e]);
  }
  for (size_t i = n4 * 4 + gtid; i < per_image; i += stride) one(pi[i], ti[i]);
  __shared__ double red[4][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double a = wave_sum_d((double)inter), b = wave_sum_d((double)psum), c = wave_sum_d((double)tsum);
  if (lane == 0) { red[wave][0] = a; red[wave][1] = b; red[wave][2] = c; }
  __syncthreads();
  if (threadIdx.x < 3) atomicAdd(sums + (size_t)img * 3 + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// per image in fp32, with the reference's operation order: (2 I + eps) / ((P + T) + eps) and (I + eps) / (((P + T) - I) + eps);
// the batch mean is taken over doubles and rounded once.  out[0] = dice, out[1] = iou, out[2 + 2 i], out[3 + 2 i] = image i.
__global__ void k_seg_finalize(int n, const double* __restrict__ sums, float eps, float* __restrict__ out) {
  __shared__ double acc[2][64];
  double d = 0.0, u = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) {
    const float I = (float)sums[3 * i], P = (float)sums[3 * i + 1], T = (float)sums[3 * i + 2];
    const float card = P + T;
    const float dice = (2.f * I + eps) / (card + eps);
    const float iou = (I + eps) / ((card - I) + eps);
    out[2 + 2 * i] = dice;
    out[3 + 2 * i] = iou;
    d += (double)dice;
    u += (double)iou;
  }
  acc[0][threadIdx.x] = d;
  acc[1][threadIdx.x] = u;
  __syncthreads();
  if (threadIdx.x < 2) {
    double s = 0.0;
    for (int j = 0; j < 64; ++j) s += acc[threadIdx.x][j];      // fixed order
    out[threadIdx.x] = (float)(s / (double)n);
  }
}

}  // namespace vk

extern "C" size_t vk_seg_metrics_workspace_bytes(int n_images) { return n_images > 0 ? (size_t)n_images * 3 * sizeof(double) : 0; }

extern "C" int vk_seg_metrics(int n_images, size_t per_image, const float* pred, const float* target, int from_logits, float threshold,
                              float eps, void* workspace, size_t workspace_bytes, float* out, void* stream) {
  using namespace vk;
  VK_CHECK_ARG(n_images >= 1 && per_image >= 1 && pred && target && workspace && out, "vk_seg_metrics: null or empty argument");
  VK_CHECK_ARG(workspace_bytes >= vk_seg_metrics_workspace_bytes(n_images), "vk_seg_metrics: workspace too small (%zu < %zu)",
               workspace_bytes, vk_seg_metrics_workspace_bytes(n_images));
  VK_CHECK_ARG(((uintptr_t)workspace & 7) == 0, "vk_seg_metrics: workspace must be 8-byte aligned");
  VK_CHECK_ARG(n_images <= 65535, "vk_seg_metrics: at most 65535 images per call");
  hipStream_t st = (hipStream_t)stream;
  vkh::ProfScope ps_("seg_metrics", st, 0.0, 8.0 * (double)n_images * (double)per_image);
  double* sums = (double*)workspace;
  VK_CHECK_HIP(hipMemsetAsync(sums, 0, (size_t)n_images * 3 * sizeof(double), st));
  const int vec_ok = ((((uintptr_t)pred | (uintptr_t)target) & 15) == 0 && per_image % 4 == 0) ? 1 : 0;
  // enough workgroups to fill the chip across the batch, none with fewer than ~4 vectors per thread
  size_t per_wg = 256 * (vec_ok ? 16 : 4);
  unsigned bx = (unsigned)((per_image + per_wg - 1) / per_wg);
  const unsigned cap = (unsigned)((4096 + n_images - 1) / n_images);
  if (bx > cap) bx = cap;
  if (bx < 1) bx = 1;
  dim3 grid(bx, (unsigned)n_images), block(256);
  if (from_logits) hipLaunchKernelGGL(k_seg_counts<true>, grid, block, 0, st, per_image, pred, target, threshold, sums, vec_ok);
  else hipLaunchKernelGGL(k_seg_counts<false>, grid, block, 0, st, per_image, pred, target, threshold, sums, vec_ok);
  hipLaunchKernelGGL(k_seg_finalize, dim3(1), dim3(64), 0, st, n_images, (const double*)sums, eps, out);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}
