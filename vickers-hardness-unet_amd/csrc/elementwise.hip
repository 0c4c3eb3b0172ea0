// HBM-bound kernels of the U-Net path (everything that is not a convolution GEMM).  All activation
// tensors are NHWC and are moved as 16-byte vectors; per-channel reductions use wave shuffles + LDS
// and finish with two fp64 atomics per channel per workgroup.
// Reference operators replaced (all dispatched by reference train.py:436 model(x) / :438 loss /
// :443,:448 backward / :449 optimizer.step): batch_norm, relu, max_pool2d, add, interpolate(nearest),
// cat, BCEWithLogitsLoss, smp DiceLoss, AdamW.
#include <math.h>

#include "vk_common.h"

namespace vk {

static inline int grid_for(size_t work_items, int block = 256, int max_blocks = 256 * 16) {
  size_t b = (work_items + block - 1) / block;
  if (b > (size_t)max_blocks) b = max_blocks;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------------------------------------
// K0: NCHW fp32 -> NHWC4 T
template <typename T>
__global__ void k_input_transform(int N, int H, int W, const float* __restrict__ x, T* __restrict__ x4) {
  // grid.y = image; one 8- (16-bit types) or 16-byte store per pixel
  const size_t HW = (size_t)H * W;
  const float* xn = x + (size_t)blockIdx.y * 3 * HW;
  T* on = x4 + (size_t)blockIdx.y * HW * 4;
  for (size_t hw = blockIdx.x * (size_t)blockDim.x + threadIdx.x; hw < HW; hw += (size_t)gridDim.x * blockDim.x) {
    const float r = xn[hw], g = xn[HW + hw], b = xn[2 * HW + hw];
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<f32x4_t*>(on + hw * 4) = f32x4_t{r, g, b, 0.f};
    } else {
      float f[8] = {r, g, b, 0.f, 0.f, 0.f, 0.f, 0.f};
      const u32x4_t pk = Vec16<T>::pack(f);
      *reinterpret_cast<u32x2_t*>(on + hw * 4) = u32x2_t{pk[0], pk[1]};
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K5: BatchNorm statistics -> affine
__global__ __launch_bounds__(256) void k_bn_finalize(int C, int train, const double* __restrict__ stats, double count,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
                                                     float* running_var, float eps, float momentum, float* scale, float* shift,
                                                     float* save_mean, float* save_invstd) {
  // 8 channels per workgroup; the 32 lanes of a half-wave each fetch one replica of the partial sums
  static_assert(VK_STATS_REPLICAS == 32, "lane mapping assumes 32 replicas");
  const int c = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int r = threadIdx.x & 31;
  double s1 = 0.0, s2 = 0.0;
  if (train && c < C) {
    s1 = stats[(size_t)r * 2 * C + c];
    s2 = stats[(size_t)r * 2 * C + C + c];
  }
  // the finishing lane's other operands are requested NOW, beside the statistics, not one dependent round trip after the reduction
  // (this launch sits on the critical path between a convolution and its consumer 46 times per step)
  const bool fin = r == 0 && c < C;
  const float ga = fin ? gamma[c] : 0.f, be = fin ? beta[c] : 0.f;
  const float rm0 = fin && running_mean ? running_mean[c] : 0.f, rv0 = fin && running_var ? running_var[c] : 0.f;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if (!fin) return;
  double mean, var;
  if (train) {
    mean = s1 / count;
    var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    if (running_mean) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (float)((1.0 - momentum) * rm0 + momentum * mean);
      running_var[c] = (float)((1.0 - momentum) * rv0 + momentum * unb);
    }
  } else {
    mean = rm0;
    var = rv0;
  }
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = ga * invstd;
  scale[c] = sc;
  shift[c] = be - (float)mean * sc;
  if (save_mean) save_mean[c] = (float)mean;
  if (save_invstd) save_invstd[c] = invstd;
}

// ------------------------------------------------------------------------------------------------
// K2: BN-apply + ReLU + maxpool 3x3 s2 p1 (+ argmax for backward)
template <typename T>
__global__ void k_bn_relu_maxpool(int N, int H, int W, int C, const T* __restrict__ z, const float* __restrict__ scale,
                                  const float* __restrict__ shift, T* __restrict__ pooled, uint8_t* __restrict__ argmax) {
  constexpr int VE = ElemTraits<T>::kVec;
  const int Hp = H / 2, Wp = W / 2, CV = C / VE;
  // grid = (pooled row segment, pooled row, image): no 64-bit division per element (it was most of this kernel's instructions)
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (unsigned)(Wp * CV)) {
    const int pw = (int)(t / (unsigned)CV), cv = (int)(t - (unsigned)pw * CV);
    const int ph = blockIdx.y, n = blockIdx.z;
    const size_t i = (((size_t)n * Hp + ph) * Wp + pw) * CV + cv;
    float sc[VE], sh[VE], best[VE];
    int bi[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) {
      sc[j] = scale[cv * VE + j];
      sh[j] = shift[cv * VE + j];
      best[j] = -INFINITY;
      bi[j] = 0;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int h = 2 * ph - 1 + r;
      if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int w = 2 * pw - 1 + s;
        if ((unsigned)w >= (unsigned)W) continue;
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(z + (((size_t)n * H + h) * W + w) * C + cv * VE);
        float f[VE];
        Vec16<T>::unpack(v, f);
#pragma unroll
        for (int j = 0; j < VE; ++j) {
          const float a = fmaxf(fmaf(f[j], sc[j], sh[j]), 0.f);
          if (a > best[j]) { best[j] = a; bi[j] = r * 3 + s; }
        }
      }
    }
    *reinterpret_cast<u32x4_t*>(pooled + i * VE) = Vec16<T>::pack(best);
    uint8_t* ap = argmax + i * VE;
    if (VE == 8) {
      uint32_t lo = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
      uint32_t hi = bi[4 % VE] | (bi[5 % VE] << 8) | (bi[6 % VE] << 16) | (bi[7 % VE] << 24);
      *reinterpret_cast<u32x2_t*>(ap) = u32x2_t{lo, hi};
    } else {
      *reinterpret_cast<uint32_t*>(ap) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    }
  }
}

// maxpool backward, gather form: dy[n][h][w][c] += sum over the <=4 windows containing (h,w) whose argmax is (h,w)
template <typename T>
__global__ void k_maxpool_bwd(int N, int H, int W, int C, const T* __restrict__ dpool, const uint8_t* __restrict__ argmax,
                              T* __restrict__ dy) {
  constexpr int VE = ElemTraits<T>::kVec;
  const int Hp = H / 2, Wp = W / 2, CV = C / VE;
  // grid = (row segment, row, image): no 64-bit division per element
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (unsigned)(W * CV)) {
    const int w = (int)(t / (unsigned)CV), cv = (int)(t - (unsigned)w * CV);
    const int h = blockIdx.y, n = blockIdx.z;
    const size_t i = (((size_t)n * H + h) * W + w) * CV + cv;
    float acc[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) acc[j] = 0.f;
    const int ph0 = (h & 1) ? (h - 1) / 2 : h / 2, nph = (h & 1) ? 2 : 1;
    const int pw0 = (w & 1) ? (w - 1) / 2 : w / 2, npw = (w & 1) ? 2 : 1;
    bool any = false;
    for (int a = 0; a < nph; ++a) {
      const int ph = ph0 + a;
      if (ph >= Hp) continue;
      const int r = h - (2 * ph - 1);
      for (int b = 0; b < npw; ++b) {
        const int pw = pw0 + b;
        if (pw >= Wp) continue;
        const int s = w - (2 * pw - 1);
        const int code = r * 3 + s;
        const size_t pi = (((size_t)n * Hp + ph) * Wp + pw) * C + cv * VE;
        const uint8_t* ap = argmax + pi;
        uint8_t am[VE];
        if (VE == 8) {
          const u32x2_t q = *reinterpret_cast<const u32x2_t*>(ap);
#pragma unroll
          for (int j = 0; j < VE; ++j) am[j] = (uint8_t)((q[j >> 2] >> (8 * (j & 3))) & 0xff);
        } else {
          const uint32_t q = *reinterpret_cast<const uint32_t*>(ap);
#pragma unroll
          for (int j = 0; j < VE; ++j) am[j] = (uint8_t)((q >> (8 * j)) & 0xff);
        }
        float g[VE];
        Vec16<T>::unpack(*reinterpret_cast<const u32x4_t*>(dpool + pi), g);
#pragma unroll
        for (int j = 0; j < VE; ++j)
          if (am[j] == code) { acc[j] += g[j]; any = true; }
      }
    }
    if (any) {
      u32x4_t* dp = reinterpret_cast<u32x4_t*>(dy + i * VE);
      float o[VE];
      Vec16<T>::unpack(*dp, o);
#pragma unroll
      for (int j = 0; j < VE; ++j) o[j] += acc[j];
      *dp = Vec16<T>::pack(o);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K6: BasicBlock tail  out = relu(bn2(z) + shortcut)
template <typename T>
__global__ __launch_bounds__(256) void k_bn_add_relu(size_t pixels, int C, const T* __restrict__ z, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, const T* __restrict__ res, const float* __restrict__ rscale,
                                                     const float* __restrict__ rshift, T* __restrict__ out) {
  constexpr int VE = ElemTraits<T>::kVec;
  constexpr int U = 4;                                   // independent 16-byte loads in flight per operand
  const int CV = C / VE;
  const size_t total = pixels * CV;
  const size_t stride = (size_t)gridDim.x * blockDim.x;  // multiple of CV => the channel group of a thread is fixed
  const size_t t0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const int c0 = (int)(t0 % CV) * VE;
  // per-workgroup coefficient table in LDS, one thread per channel: 32 scalar global loads per thread in this prologue cost
  // more than the whole tensor pass on the small (layer3/4) blocks (35 us floor per launch, measured)
  __shared__ float cf[4][512];
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    cf[0][c] = scale[c];
    cf[1][c] = shift[c];
    cf[2][c] = rscale ? rscale[c] : 1.f;
    cf[3][c] = rscale ? rshift[c] : 0.f;
  }
  __syncthreads();
  float sc[VE], sh[VE], rsc[VE], rsh[VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) {
    sc[j] = cf[0][c0 + j];
    sh[j] = cf[1][c0 + j];
    rsc[j] = cf[2][c0 + j];
    rsh[j] = cf[3][c0 + j];
  }
  for (size_t i = t0; i < total; i += U * stride) {
    u32x4_t zv[U], rv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t k = i + u * stride;
      if (k < total) {
        zv[u] = *reinterpret_cast<const u32x4_t*>(z + k * VE);
        rv[u] = *reinterpret_cast<const u32x4_t*>(res + k * VE);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t k = i + u * stride;
      if (k >= total) continue;
      float f[VE], r[VE];
      Vec16<T>::unpack(zv[u], f);
      Vec16<T>::unpack(rv[u], r);
#pragma unroll
      for (int j = 0; j < VE; ++j) f[j] = fmaxf(fmaf(f[j], sc[j], sh[j]) + fmaf(r[j], rsc[j], rsh[j]), 0.f);
      *reinterpret_cast<u32x4_t*>(out + k * VE) = Vec16<T>::pack(f);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Stem tail backward in ONE pass over the 256x256x64 gradient (instead of maxpool backward's read-modify-write followed by the
// BN-backward reduce's read): g = (skip gradient + maxpool-backward gather) * [relu(bn(z)) > 0] is written back in place and
// sum(g), sum(g*z) go to the replicated fp64 slabs, so the second phase is the pre-masked apply (mask_mode 0).
// grid = (row segment, group of PB_ROWS rows, image); lanes along (w, channel vector).
constexpr int PB_ROWS = 16;
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool_bwd_bn_reduce(int N, int H, int W, int C, const T* __restrict__ dpool,
                                                               const uint8_t* __restrict__ argmax, const T* __restrict__ z,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               T* __restrict__ dy, double* sums) {
  constexpr int VE = ElemTraits<T>::kVec;
  const int Hp = H / 2, Wp = W / 2, CV = C / VE;
  const int tid = threadIdx.x;
  const unsigned t = blockIdx.x * blockDim.x + tid;
  const bool active = t < (unsigned)(W * CV);
  const int w = active ? (int)(t / (unsigned)CV) : 0, cv = active ? (int)(t - (unsigned)w * CV) : 0;
  const int n = blockIdx.z;
  float sc[VE], sh[VE], s1[VE], s2[VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) {
    sc[j] = scale[cv * VE + j];
    sh[j] = shift[cv * VE + j];
    s1[j] = 0.f;
    s2[j] = 0.f;
  }
  const int pw0 = (w & 1) ? (w - 1) / 2 : w / 2, npw = (w & 1) ? 2 : 1;
  if (active) {
    for (int hr = 0; hr < PB_ROWS; ++hr) {
      const int h = blockIdx.y * PB_ROWS + hr;
      if (h >= H) break;
      const size_t i = (((size_t)n * H + h) * W + w) * CV + cv;
      float g[VE], zf[VE];
      Vec16<T>::unpack(*reinterpret_cast<const u32x4_t*>(dy + i * VE), g);
      Vec16<T>::unpack(*reinterpret_cast<const u32x4_t*>(z + i * VE), zf);
      const int ph0 = (h & 1) ? (h - 1) / 2 : h / 2, nph = (h & 1) ? 2 : 1;
      float acc[VE];                       // pool contributions first, then one add: the summation order of k_maxpool_bwd
#pragma unroll
      for (int j = 0; j < VE; ++j) acc[j] = 0.f;
      for (int a = 0; a < nph; ++a) {
        const int ph = ph0 + a;
        if (ph >= Hp) continue;
        const int r = h - (2 * ph - 1);
        for (int b = 0; b < npw; ++b) {
          const int pw = pw0 + b;
          if (pw >= Wp) continue;
          const int code = r * 3 + (w - (2 * pw - 1));
          const size_t pi = (((size_t)n * Hp + ph) * Wp + pw) * C + cv * VE;
          uint32_t am[VE];
          if (VE == 8) {
            const u32x2_t q = *reinterpret_cast<const u32x2_t*>(argmax + pi);
#pragma unroll
            for (int j = 0; j < VE; ++j) am[j] = (q[j >> 2] >> (8 * (j & 3))) & 0xffu;
          } else {
            const uint32_t q = *reinterpret_cast<const uint32_t*>(argmax + pi);
#pragma unroll
            for (int j = 0; j < VE; ++j) am[j] = (q >> (8 * j)) & 0xffu;
          }
          float pg[VE];
          Vec16<T>::unpack(*reinterpret_cast<const u32x4_t*>(dpool + pi), pg);
#pragma unroll
          for (int j = 0; j < VE; ++j)
            if (am[j] == (uint32_t)code) acc[j] += pg[j];
        }
      }
#pragma unroll
      for (int j = 0; j < VE; ++j) g[j] += acc[j];
      // the unfused path rounds the sum to T when maxpool backward stores it; keep that rounding, then mask
      u32x4_t v = Vec16<T>::pack(g);
      Vec16<T>::unpack(v, g);
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        if (!(fmaf(zf[j], sc[j], sh[j]) > 0.f)) g[j] = 0.f;
        s1[j] += g[j];
        s2[j] += g[j] * zf[j];
      }
      *reinterpret_cast<u32x4_t*>(dy + i * VE) = Vec16<T>::pack(g);
    }
  }
  // block reduction: thread (px, cv) -> channel sums over the block's pixels (as k_bn_bwd_reduce)
  __shared__ float red[256 * 2 * 8];
#pragma unroll
  for (int j = 0; j < VE; ++j) {
    red[(tid * VE + j) * 2] = active ? s1[j] : 0.f;
    red[(tid * VE + j) * 2 + 1] = active ? s2[j] : 0.f;
  }
  __syncthreads();
  const int rows = 256 / CV;           // pixels per block (CV divides 256 for the channel counts on this path)
  for (int c = tid; c < C; c += 256) {
    const int ccv = c / VE, cj = c % VE;
    double a = 0.0, b = 0.0;
    for (int r = 0; r < rows; ++r) {
      const int tt = r * CV + ccv;
      a += red[(tt * VE + cj) * 2];
      b += red[(tt * VE + cj) * 2 + 1];
    }
    double* sp = sums + (size_t)((blockIdx.x + blockIdx.y * gridDim.x) % VK_STATS_REPLICAS) * 2 * C;
    atomicAdd(sp + c, a);
    atomicAdd(sp + C + c, b);
  }
}

// ------------------------------------------------------------------------------------------------
// K13: BatchNorm (+ReLU) backward
template <typename T, int MASK>
__device__ __forceinline__ void masked_grad(const T* dy, const T* z, const T* mask_src, size_t off, const float* sc,
                                            const float* sh, float* g, float* zf) {
  constexpr int VE = ElemTraits<T>::kVec;
  Vec16<T>::unpack(*reinterpret_cast<const u32x4_t*>(dy + off), g);
  Vec16<T>::unpack(*reinterpret_cast<const u32x4_t*>(z + off), zf);
  if (MASK == 1) {
#pragma unroll
    for (int j = 0; j < VE; ++j)
      if (!(fmaf(zf[j], sc[j], sh[j]) > 0.f)) g[j] = 0.f;
  } else if (MASK == 2) {
    float m[VE];
    Vec16<T>::unpack(*reinterpret_cast<const u32x4_t*>(mask_src + off), m);
#pragma unroll
    for (int j = 0; j < VE; ++j)
      if (!(m[j] > 0.f)) g[j] = 0.f;
  }
}

template <typename T, int MASK>
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(size_t pixels, int C, const T* __restrict__ dy, const T* __restrict__ z,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const T* __restrict__ mask_src, double* sums) {
  constexpr int VE = ElemTraits<T>::kVec;
  const int CV = C / VE;               // <= 128
  const int rows = 256 / CV;           // pixel rows handled per pass by this block (>= 2)
  const int tid = threadIdx.x;
  const int cv = tid % CV, row = tid / CV;
  const bool active = row < rows;
  float sc[VE], sh[VE], s1[VE], s2[VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) {
    sc[j] = (MASK == 1) ? scale[cv * VE + j] : 1.f;
    sh[j] = (MASK == 1) ? shift[cv * VE + j] : 0.f;
    s1[j] = 0.f;
    s2[j] = 0.f;
  }
  if (active) {
    // 4 independent row loads in flight per thread (latency-bound otherwise)
    const size_t stride = (size_t)gridDim.x * rows;
    size_t p = (size_t)blockIdx.x * rows + row;
    for (; p + 3 * stride < pixels; p += 4 * stride) {
      float g[4][VE], zf[4][VE];
#pragma unroll
      for (int u = 0; u < 4; ++u) masked_grad<T, MASK>(dy, z, mask_src, (p + u * stride) * C + cv * VE, sc, sh, g[u], zf[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < VE; ++j) { s1[j] += g[u][j]; s2[j] += g[u][j] * zf[u][j]; }
    }
    for (; p < pixels; p += stride) {
      float g[VE], zf[VE];
      masked_grad<T, MASK>(dy, z, mask_src, p * C + cv * VE, sc, sh, g, zf);
#pragma unroll
      for (int j = 0; j < VE; ++j) { s1[j] += g[j]; s2[j] += g[j] * zf[j]; }
    }
  }
  __shared__ float red[256 * 2 * 8];
#pragma unroll
  for (int j = 0; j < VE; ++j) {
    red[(tid * VE + j) * 2] = active ? s1[j] : 0.f;
    red[(tid * VE + j) * 2 + 1] = active ? s2[j] : 0.f;
  }
  __syncthreads();
  // thread c (< C) sums its channel over the `rows` row-threads
  for (int c = tid; c < C; c += 256) {
    const int ccv = c / VE, cj = c % VE;
    double a = 0.0, b = 0.0;
    for (int r = 0; r < rows; ++r) {
      const int t = r * CV + ccv;
      a += red[(t * VE + cj) * 2];
      b += red[(t * VE + cj) * 2 + 1];
    }
    double* sp = sums + (size_t)(blockIdx.x % VK_STATS_REPLICAS) * 2 * C;
    atomicAdd(sp + c, a);
    atomicAdd(sp + C + c, b);
  }
}

__global__ __launch_bounds__(256) void k_bn_bwd_coeffs(int C, const double* __restrict__ sums, double count, const float* __restrict__ gamma,
                                                       const float* __restrict__ save_mean, const float* __restrict__ save_invstd, float* dgamma,
                                                       float* dbeta, float* coef) {
  // 8 channels per workgroup; the 32 lanes of a half-wave each fetch one replica of the partial sums
  const int c = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int q = threadIdx.x & 31;
  double sg = 0.0, sgz = 0.0;
  if (c < C) {
    sg = sums[(size_t)q * 2 * C + c];
    sgz = sums[(size_t)q * 2 * C + C + c];
  }
  const bool fin = q == 0 && c < C;           // its operands travel beside the sums (see k_bn_finalize)
  const double mu = fin ? save_mean[c] : 0.f, r = fin ? save_invstd[c] : 0.f, ga = fin ? gamma[c] : 0.f;
  const float dg0 = fin ? dgamma[c] : 0.f, db0 = fin ? dbeta[c] : 0.f;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    sg += __shfl_xor(sg, o, 64);
    sgz += __shfl_xor(sgz, o, 64);
  }
  if (!fin) return;
  const double dg = r * (sgz - mu * sg);     // sum g * xhat
  const double db = sg;
  dgamma[c] = dg0 + (float)dg;
  dbeta[c] = db0 + (float)db;
  const double a = ga * r;
  const double b = -ga * r * r * dg / count;
  const double cc = -a * db / count - b * mu;
  coef[c] = (float)a;
  coef[C + c] = (float)b;
  coef[2 * C + c] = (float)cc;
}

// dz = a*g + b*z + c with (a, b, c) derived in-kernel from the reduction sums:
//   dgamma = r*(S_gz - mu*S_g), dbeta = S_g, a = gamma*r, b = -gamma*r^2*dgamma/M, c = -a*dbeta/M - b*mu
// Workgroup 0 also accumulates dgamma/dbeta into the flat gradient buffer (coef == nullptr: fused mode).
template <typename T, int MASK>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(size_t pixels, int C, const T* __restrict__ dy, const T* __restrict__ z,
                                                      const float* __restrict__ scale, const float* __restrict__ shift, const T* __restrict__ mask_src,
                                                      const float* __restrict__ coef, const double* __restrict__ sums, double count,
                                                      const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                      const float* __restrict__ save_invstd, float* dgamma, float* dbeta, T* __restrict__ dz, T* g_out,
                                                      int g_acc) {
  constexpr int VE = ElemTraits<T>::kVec;
  constexpr int U = 2;
  const int CV = C / VE;
  const size_t total = pixels * CV;
  const size_t stride = (size_t)gridDim.x * blockDim.x;   // multiple of CV
  const size_t t0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const int c0 = (int)(t0 % CV) * VE;
  // per-workgroup coefficient table in LDS (one thread per channel), so the per-thread prologue is 3 LDS reads / channel
  __shared__ float cf[3][512];
  if (coef) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      cf[0][c] = coef[c];
      cf[1][c] = coef[C + c];
      cf[2][c] = coef[2 * C + c];
    }
  } else {
    // 8 threads per channel add 4 of the 32 replicas each, then an 8-lane butterfly
    static_assert(VK_STATS_REPLICAS == 32, "replica mapping");
    for (int c8 = threadIdx.x; c8 < C * 8; c8 += blockDim.x) {
      const int c = c8 >> 3, q0 = (c8 & 7) * 4;
      double sg = 0.0, sgz = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        sg += sums[(size_t)(q0 + q) * 2 * C + c];
        sgz += sums[(size_t)(q0 + q) * 2 * C + C + c];
      }
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) {
        sg += __shfl_xor(sg, o, 64);
        sgz += __shfl_xor(sgz, o, 64);
      }
      if ((c8 & 7) == 0) {
        // only the cancelling difference needs fp64 (one DFMA per channel); the rest is fp32
        const float mu = save_mean[c], r = save_invstd[c], ga = gamma[c];
        const float d = (float)(sgz - (double)mu * sg);
        const float sgf = (float)sg, inv = (float)(1.0 / count);
        const float dg = r * d;
        const float a = ga * r, b = -a * r * dg * inv;
        cf[0][c] = a;
        cf[1][c] = b;
        cf[2][c] = -a * sgf * inv - b * mu;
        if (blockIdx.x == 0) {
          dgamma[c] += dg;
          dbeta[c] += sgf;
        }
      }
    }
  }
  __syncthreads();
  float sc[VE], sh[VE], ca[VE], cb[VE], cc[VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) {
    sc[j] = (MASK == 1) ? scale[c0 + j] : 1.f;
    sh[j] = (MASK == 1) ? shift[c0 + j] : 0.f;
    ca[j] = cf[0][c0 + j];
    cb[j] = cf[1][c0 + j];
    cc[j] = cf[2][c0 + j];
  }
  for (size_t i = t0; i < total; i += U * stride) {
    float g[U][VE], zf[U][VE];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * stride < total) masked_grad<T, MASK>(dy, z, mask_src, (i + u * stride) * VE, sc, sh, g[u], zf[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t k = i + u * stride;
      if (k >= total) continue;
      float o[VE];
#pragma unroll
      for (int j = 0; j < VE; ++j) o[j] = fmaf(ca[j], g[u][j], fmaf(cb[j], zf[u][j], cc[j]));
      *reinterpret_cast<u32x4_t*>(dz + k * VE) = Vec16<T>::pack(o);
      if (g_out) {
        u32x4_t* gp = reinterpret_cast<u32x4_t*>(g_out + k * VE);
        if (g_acc) {
          float old[VE];
          Vec16<T>::unpack(*gp, old);
#pragma unroll
          for (int j = 0; j < VE; ++j) g[u][j] += old[j];
        }
        *gp = Vec16<T>::pack(g[u]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K14: nearest-x2 upsample backward (2x2 sum)
template <typename T>
__global__ void k_upsample2x_bwd(int N, int H, int W, int C, const T* __restrict__ d_up, T* __restrict__ d_low, int acc) {
  constexpr int VE = ElemTraits<T>::kVec;
  const int Hl = H / 2, Wl = W / 2, CV = C / VE;
  const size_t total = (size_t)N * Hl * Wl * CV;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    size_t t = i / CV;
    const int wl = (int)(t % Wl);
    t /= Wl;
    const int hl = (int)(t % Hl);
    const int n = (int)(t / Hl);
    float s[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) s[j] = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float f[VE];
        Vec16<T>::unpack(*reinterpret_cast<const u32x4_t*>(d_up + (((size_t)n * H + 2 * hl + a) * W + 2 * wl + b) * C + cv * VE), f);
#pragma unroll
        for (int j = 0; j < VE; ++j) s[j] += f[j];
      }
    u32x4_t* op = reinterpret_cast<u32x4_t*>(d_low + i * VE);
    if (acc) {
      float o[VE];
      Vec16<T>::unpack(*op, o);
#pragma unroll
      for (int j = 0; j < VE; ++j) s[j] += o[j];
    }
    *op = Vec16<T>::pack(s);
  }
}

// ------------------------------------------------------------------------------------------------
// K9: segmentation head 3x3, 16 -> 1, bias, fp32 logits; input = relu(bn(z)) applied on load
template <typename T>
__device__ __forceinline__ void load_act16(const T* p, const float* sc, const float* sh, bool affine, bool relu, float* a) {
  constexpr int VE = ElemTraits<T>::kVec;
#pragma unroll
  for (int v = 0; v < 16 / VE; ++v) Vec16<T>::unpack(*reinterpret_cast<const u32x4_t*>(p + v * VE), a + v * VE);
  if (affine) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      a[j] = fmaf(a[j], sc[j], sh[j]);
      if (relu) a[j] = fmaxf(a[j], 0.f);
    }
  }
}

// ---- forward: 16x16 pixel tile per workgroup; the 18x18x16 activated halo (BN + ReLU applied ONCE per element) is
// staged in LDS as fp32 (80-byte pixel stride: conflict-free ds_read_b128); weights are read through a uniform
// pointer with compile-time indices, i.e. they live in scalar registers.
template <typename T>
__global__ __launch_bounds__(256) void k_head_fwd(int N, int H, int W, int tiles_x, int tiles_y, const T* __restrict__ z,
                                                  const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                  const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ logits) {
  constexpr int VE = ElemTraits<T>::kVec, VPP = 16 / VE;      // vectors per pixel
  constexpr int PS = 20;                                      // floats per LDS pixel (16 + 4 pad)
  __shared__ __attribute__((aligned(16))) float tile[18 * 18 * PS];
  const int tid = threadIdx.x;
  int bt = blockIdx.x;
  const int tx0 = bt % tiles_x;
  bt /= tiles_x;
  const int ty0 = bt % tiles_y;
  const int n = bt / tiles_y;
  const int y0 = ty0 * 16, x0 = tx0 * 16;
  const bool affine = scale != nullptr;
  // 256 % VPP == 0: a thread's vectors always cover the same channels -> scale/shift once, not 2 x 8 scalar loads per vector;
  // all of the thread's halo vectors are requested before the first one is consumed
  const int vec = tid % VPP;
  float sc[VE], sh[VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) { sc[j] = affine ? scale[vec * VE + j] : 1.f; sh[j] = affine ? shift[vec * VE + j] : 0.f; }
  constexpr int NHV = (324 * VPP + 255) / 256;
  u32x4_t raw[NHV];
  bool inb[NHV];
#pragma unroll
  for (int i = 0; i < NHV; ++i) {
    const int v = tid + i * 256;
    const int hp = v / VPP;
    const int hy = hp / 18, hx = hp - hy * 18;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    inb[i] = v < 324 * VPP && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    raw[i] = u32x4_t{0, 0, 0, 0};
    if (inb[i]) raw[i] = *reinterpret_cast<const u32x4_t*>(z + (((size_t)n * H + y) * W + x) * 16 + vec * VE);
  }
#pragma unroll
  for (int i = 0; i < NHV; ++i) {
    const int v = tid + i * 256;
    if (v >= 324 * VPP) continue;
    const int hp = v / VPP;
    float f[VE];
    Vec16<T>::unpack(raw[i], f);
    if (affine) {
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        f[j] = fmaf(f[j], sc[j], sh[j]);
        if (relu) f[j] = fmaxf(f[j], 0.f);
      }
    }
    if (!inb[i]) {
#pragma unroll
      for (int j = 0; j < VE; ++j) f[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < VE; j += 4) *reinterpret_cast<f32x4_t*>(&tile[hp * PS + vec * VE + j]) = f32x4_t{f[j], f[j + 1], f[j + 2], f[j + 3]};
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
  // two accumulator pairs (channels 4q, 4q+1 / 4q+2, 4q+3) so that the 144 products go through v_pk_fma_f32, two per lane and cycle,
  // in two independent chains — one scalar chain of 144 dependent FMAs made this kernel VALU-bound at 2.2x its HBM time (r03)
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  f32x2_t acc_a = f32x2_t{bias[0], 0.f}, acc_b = f32x2_t{0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const float* a = &tile[((ty + r) * 18 + tx + s) * PS];
      const float* wt = w + (r * 3 + s) * 16;
#pragma unroll
      for (int j = 0; j < 16; j += 4) {
        const f32x4_t av = *reinterpret_cast<const f32x4_t*>(a + j);
        acc_a = __builtin_elementwise_fma(f32x2_t{av[0], av[1]}, f32x2_t{wt[j], wt[j + 1]}, acc_a);
        acc_b = __builtin_elementwise_fma(f32x2_t{av[2], av[3]}, f32x2_t{wt[j + 2], wt[j + 3]}, acc_b);
      }
    }
  const float acc = (acc_a[0] + acc_a[1]) + (acc_b[0] + acc_b[1]);
  const int y = y0 + ty, x = x0 + tx;
  if (y < H && x < W) logits[((size_t)n * H + y) * W + x] = acc;
}

// ---- backward, data: dy[m][c] = sum_tap dl[m + (1-r, 1-s)] * w[tap][c]
template <typename T>
__global__ __launch_bounds__(256) void k_head_dgrad(int N, int H, int W, int tiles_x, int tiles_y, int ntiles, const float* __restrict__ w,
                                                    const float* __restrict__ dl, T* __restrict__ dy, const T* __restrict__ bnr_z,
                                                    const float* __restrict__ bnr_scale, const float* __restrict__ bnr_shift, double* bnr_sums) {
  constexpr int VE = ElemTraits<T>::kVec;
  __shared__ float dt[2][18 * 18];
  __shared__ float bred[4][32];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  float s1[16], s2[16], bsc[16], bsh[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    s1[j] = 0.f; s2[j] = 0.f;
    bsc[j] = bnr_z ? bnr_scale[j] : 1.f;
    bsh[j] = bnr_z ? bnr_shift[j] : 0.f;
  }
  int it = 0;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x, ++it) {
    int bt = t;
    const int tx0 = bt % tiles_x;
    bt /= tiles_x;
    const int ty0 = bt % tiles_y;
    const int n = bt / tiles_y;
    const int y0 = ty0 * 16, x0 = tx0 * 16;
    float* d = dt[it & 1];
    // the z vectors of the fused BN+ReLU backward are requested first: their latency runs under the dlogits staging and the FMAs
    const int y = y0 + ty, x = x0 + tx;
    const size_t off = (((size_t)n * H + (y < H ? y : 0)) * W + (x < W ? x : 0)) * 16;
    u32x4_t zraw[16 / VE];
#pragma unroll
    for (int v = 0; v < 16 / VE; ++v) zraw[v] = u32x4_t{0, 0, 0, 0};
    if (bnr_z && y < H && x < W) {
#pragma unroll
      for (int v = 0; v < 16 / VE; ++v) zraw[v] = *reinterpret_cast<const u32x4_t*>(bnr_z + off + v * VE);
    }
    for (int hp = tid; hp < 324; hp += 256) {
      const int hy = hp / 18, hx = hp - hy * 18;
      const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
      d[hp] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? dl[((size_t)n * H + yy) * W + xx] : 0.f;
    }
    __syncthreads();                 // (double-buffered tile: the other buffer was last read two iterations ago)
    float o[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j] = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const float dv = d[(ty + 2 - r) * 18 + tx + 2 - s];      // halo origin is (y0-1, x0-1): pixel + (1 - r, 1 - s)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] = fmaf(dv, w[(r * 3 + s) * 16 + j], o[j]);
      }
    if (y < H && x < W) {
#pragma unroll
      for (int v = 0; v < 16 / VE; ++v) {
        u32x4_t pk = Vec16<T>::pack(o + v * VE);
        if (bnr_z) {       // g = dy * [relu(bn(z)) > 0]; sums over the stored values
          float g[VE], zf[VE];
          Vec16<T>::unpack(pk, g);
          Vec16<T>::unpack(zraw[v], zf);
#pragma unroll
          for (int j = 0; j < VE; ++j) {
            if (!(fmaf(zf[j], bsc[v * VE + j], bsh[v * VE + j]) > 0.f)) g[j] = 0.f;
            s1[v * VE + j] += g[j];
            s2[v * VE + j] += g[j] * zf[j];
          }
          pk = Vec16<T>::pack(g);
        }
        *reinterpret_cast<u32x4_t*>(dy + off + v * VE) = pk;
      }
    }
  }
  if (bnr_z) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float a = wave_sum(s1[j]), b = wave_sum(s2[j]);
      if (lane == 0) { bred[wave][j] = a; bred[wave][16 + j] = b; }
    }
    __syncthreads();
    if (tid < 32) {
      const float v = bred[0][tid] + bred[1][tid] + bred[2][tid] + bred[3][tid];
      atomicAdd(bnr_sums + (size_t)(blockIdx.x % VK_STATS_REPLICAS) * 32 + tid, (double)v);   // [replica][2][16]
    }
  }
}

// ---- backward, weights: dw[tap][c] += sum_m dl[m + (1-r, 1-s)] * a[m][c];  dbias += sum_m dl[m]
// persistent workgroups walk 16x16 tiles; 144 + 1 accumulators per thread, one cross-lane reduction at the end.
template <typename T>
__global__ __launch_bounds__(256) void k_head_wgrad(int N, int H, int W, int tiles_x, int tiles_y, int ntiles, const T* __restrict__ z,
                                                    const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                    const float* __restrict__ dl, float* dw, float* dbias, float* __restrict__ part) {
  __shared__ float dt[2][18 * 18];
  __shared__ float red[4][145];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const bool affine = scale != nullptr;
  float sc[16], sh[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) { sc[j] = affine ? scale[j] : 1.f; sh[j] = affine ? shift[j] : 0.f; }
  float gw[144];
#pragma unroll
  for (int j = 0; j < 144; ++j) gw[j] = 0.f;
  float gb = 0.f;
  // software pipeline over this workgroup's tiles: the next tile's activations (raw vectors) and dlogits halo are loaded
  // into registers before the current tile's 144 FMAs per pixel, so the HBM latency is paid once, not once per tile
  constexpr int VE = ElemTraits<T>::kVec, VPP = 16 / VE;
  u32x4_t ar[VPP], ar_n[VPP];
  float dr[2], dr_n[2];
  bool ok = false, ok_n = false;
  auto fetch = [&](int t, u32x4_t (&av)[VPP], float (&dv)[2], bool& inb) {
    int bt = t;
    const int tx0 = bt % tiles_x;
    bt /= tiles_x;
    const int ty0 = bt % tiles_y;
    const int n = bt / tiles_y;
    const int y0 = ty0 * 16, x0 = tx0 * 16;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int hp = tid + k * 256;
      const int hy = hp / 18, hx = hp - hy * 18;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      dv[k] = (hp < 324 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? dl[((size_t)n * H + y) * W + x] : 0.f;
    }
    const int y = y0 + ty, x = x0 + tx;
    inb = y < H && x < W;
    const T* zp = z + (((size_t)n * H + (inb ? y : 0)) * W + (inb ? x : 0)) * 16;
#pragma unroll
    for (int v = 0; v < VPP; ++v) av[v] = *reinterpret_cast<const u32x4_t*>(zp + v * VE);
  };
  int t = blockIdx.x;
  if (t < ntiles) {
    fetch(t, ar, dr, ok);
    dt[0][tid] = dr[0];
    if (tid + 256 < 324) dt[0][tid + 256] = dr[1];
  }
  __syncthreads();
  for (int it = 0; t < ntiles; t += gridDim.x, ++it) {
    const bool has_next = t + (int)gridDim.x < ntiles;
    if (has_next) fetch(t + gridDim.x, ar_n, dr_n, ok_n);
    const float* d = dt[it & 1];
    float a[16];
#pragma unroll
    for (int v = 0; v < VPP; ++v) Vec16<T>::unpack(ar[v], a + v * VE);
    if (affine) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        a[j] = fmaf(a[j], sc[j], sh[j]);
        if (relu) a[j] = fmaxf(a[j], 0.f);
      }
    }
    if (!ok) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = 0.f;
    }
    gb += d[(ty + 1) * 18 + tx + 1];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const float dv = d[(ty + 2 - r) * 18 + tx + 2 - s];
#pragma unroll
        for (int j = 0; j < 16; ++j) gw[(r * 3 + s) * 16 + j] = fmaf(dv, a[j], gw[(r * 3 + s) * 16 + j]);
      }
    if (has_next) {                  // the other dlogits buffer was last read one iteration ago, before the barrier below
      float* dn = dt[(it + 1) & 1];
      dn[tid] = dr_n[0];
      if (tid + 256 < 324) dn[tid + 256] = dr_n[1];
#pragma unroll
      for (int v = 0; v < VPP; ++v) ar[v] = ar_n[v];
      ok = ok_n;
    }
    __syncthreads();
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int j = 0; j < 144; ++j) {
    const float v = wave_sum(gw[j]);
    if (lane == 0) red[wave][j] = v;
  }
  {
    const float v = wave_sum(gb);
    if (lane == 0) red[wave][144] = v;
  }
  __syncthreads();
  if (tid < 145) {
    const float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    if (part) part[(size_t)blockIdx.x * 148 + tid] = v;       // reproducible mode: k_head_wgrad_reduce adds the workgroups in order
    else if (tid < 144) atomicAdd(dw + tid, v);
    else atomicAdd(dbias, v);
  }
}

// ---- backward, data + weights in ONE pass on the matrix cores (16-bit types, r03).  The two kernels above are VALU-bound (144 FMAs
// per pixel each, the BN+ReLU transform of the same z twice) at 134 + 117 us for 570 + 300 MB; here per 16x16 tile
//   * every thread stages its pixel once: raw z, a = relu(bn(z)) (AffineRelu, as every other convolution stages its operand) and the
//     pixel's nine shifted dlogits as a 16-"channel" vector T[px][tap] (taps 9..15 zero), all as T in wave-private LDS tiles
//     (32 B per pixel; the wave that writes a tile is the only one that reads it: no workgroup barrier for them);
//   * data gradient: D[c][px] = sum_tap W[c][tap] T[px][tap] — one 16x16x32 MFMA per 16 pixels (A = the head filter as a constant
//     fragment, B = ds_read_b128 of T), lane = 4 channels of one pixel as in the convolution kernels: round, ReLU mask from z,
//     BN-backward sums, 8-byte store;
//   * weight gradient: D[tap][c] += sum_px T[px][tap] a[px][c] — the LDS-transposed fragment reads of wgrad_halo_kernel with T in the
//     role of dz: one MFMA per 32 pixels, one f32x4 accumulator per wave for the whole launch.
// dlogits and the filter are rounded to T here (what autocast does to them in the reference: the head convolution runs in the
// low-precision type, train.py:431-438); accumulation, the gradient of the filter and the sums stay fp32 / fp64.
template <typename T>
__global__ __launch_bounds__(256) void k_head_bwd_mfma(int N, int H, int W, int tiles_x, int tiles_y, int ntiles, const T* __restrict__ z,
                                                       const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                       const float* __restrict__ w, const float* __restrict__ dl, T* __restrict__ dy,
                                                       double* bnr_sums, float* dw, float* dbias, float* __restrict__ part) {
  static_assert(sizeof(T) == 2, "16-bit element types");
  constexpr int PXB = 32;                                   // bytes per pixel of the wave-private tiles (16 elements)
  __shared__ float dt[2][18 * 18];
  __shared__ __attribute__((aligned(16))) char wl[4][3][64 * PXB];
  __shared__ float red[4][148];
  typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ty = tid >> 4, tx = tid & 15;
  const int li = lane & 15, kg = lane >> 4;
  char* const zt = wl[wave][0];
  char* const at = wl[wave][1];
  char* const Tt = wl[wave][2];
  float sc[16], sh[16], bsc[4], bsh[4];
#pragma unroll
  for (int j = 0; j < 16; ++j) { sc[j] = scale[j]; sh[j] = shift[j]; }
#pragma unroll
  for (int e = 0; e < 4; ++e) { bsc[e] = scale[kg * 4 + e]; bsh[e] = shift[kg * 4 + e]; }
  // constant A fragment of the data gradient: W[c = li][tap = 8 kg + i], taps >= 9 are zero
  u32x4_t wfrag;
  {
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int tap = kg * 8 + i;
      f[i] = tap < 9 ? w[tap * 16 + li] : 0.f;
    }
    wfrag = Vec16<T>::pack(f);
  }
  f32x4_t dwacc = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float gb = 0.f, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  // transposed-read lane geometry of wgrad_halo_kernel at a 32-byte pixel stride
  const int lane_t = (4 * (lane >> 4) + ((lane & 15) >> 2)) * PXB + (4 * (lane & 3)) * 2;
  auto tr = [](const char* q) { return __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(q))); };

  u32x4_t ar[2], ar_n[2];
  float dr[2], dr_n[2];
  bool ok = false, ok_n = false;
  auto fetch = [&](int t, u32x4_t (&av)[2], float (&dv)[2], bool& inb) {
    int bt = t;
    const int tx0 = bt % tiles_x;
    bt /= tiles_x;
    const int ty0 = bt % tiles_y;
    const int n = bt / tiles_y;
    const int y0 = ty0 * 16, x0 = tx0 * 16;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int hp = tid + k * 256;
      const int hy = hp / 18, hx = hp - hy * 18;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      dv[k] = (hp < 324 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? dl[((size_t)n * H + y) * W + x] : 0.f;
    }
    const int y = y0 + ty, x = x0 + tx;
    inb = y < H && x < W;
    const T* zp = z + (((size_t)n * H + (inb ? y : 0)) * W + (inb ? x : 0)) * 16;
    av[0] = *reinterpret_cast<const u32x4_t*>(zp);
    av[1] = *reinterpret_cast<const u32x4_t*>(zp + 8);
  };
  int t = blockIdx.x;
  if (t < ntiles) {
    fetch(t, ar, dr, ok);
    dt[0][tid] = dr[0];
    if (tid + 256 < 324) dt[0][tid + 256] = dr[1];
  }
  __syncthreads();
  for (int it = 0; t < ntiles; t += gridDim.x, ++it) {
    const bool has_next = t + (int)gridDim.x < ntiles;
    if (has_next) fetch(t + gridDim.x, ar_n, dr_n, ok_n);
    int bt = t;
    const int tx0 = bt % tiles_x;
    bt /= tiles_x;
    const int ty0 = bt % tiles_y;
    const int n = bt / tiles_y;
    const int y0 = ty0 * 16, x0 = tx0 * 16;
    const float* d = dt[it & 1];
    // ---- thread = pixel: the three wave-private tiles
    {
      const u32x4_t z0 = ok ? ar[0] : u32x4_t{0, 0, 0, 0}, z1 = ok ? ar[1] : u32x4_t{0, 0, 0, 0};
      *reinterpret_cast<u32x4_t*>(zt + lane * PXB) = z0;
      *reinterpret_cast<u32x4_t*>(zt + lane * PXB + 16) = z1;
      u32x4_t a0 = AffineRelu<T>::run(z0, sc, sh, relu != 0), a1 = AffineRelu<T>::run(z1, sc + 8, sh + 8, relu != 0);
      if (!ok) { a0 = u32x4_t{0, 0, 0, 0}; a1 = u32x4_t{0, 0, 0, 0}; }          // pixels outside the map add nothing to dW
      *reinterpret_cast<u32x4_t*>(at + lane * PXB) = a0;
      *reinterpret_cast<u32x4_t*>(at + lane * PXB + 16) = a1;
      float tf[16];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) tf[r * 3 + s_] = d[(ty + 2 - r) * 18 + tx + 2 - s_];     // halo origin (y0-1, x0-1): pixel + (1-r, 1-s)
#pragma unroll
      for (int k = 9; k < 16; ++k) tf[k] = 0.f;
      *reinterpret_cast<u32x4_t*>(Tt + lane * PXB) = Vec16<T>::pack(tf);
      *reinterpret_cast<u32x4_t*>(Tt + lane * PXB + 16) = Vec16<T>::pack(tf + 8);
      gb += d[(ty + 1) * 18 + tx + 1];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- weight gradient: two 32-pixel steps
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const char* tq = Tt + lane_t + (32 * ks) * PXB;
      const char* aq = at + lane_t + (32 * ks) * PXB;
      const u32x2_t tl = tr(tq), th = tr(tq + 16 * PXB), al = tr(aq), ah = tr(aq + 16 * PXB);
      dwacc = Mma<T>::run(u32x4_t{tl[0], tl[1], th[0], th[1]}, u32x4_t{al[0], al[1], ah[0], ah[1]}, dwacc);
    }
    // ---- data gradient: tile rows 4 wave + j, lane = 4 channels (4 kg ..) of pixel li
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pl = j * 16 + li;
      u32x4_t X = *reinterpret_cast<const u32x4_t*>(Tt + pl * PXB + (kg & 1) * 16);
      if (kg >= 2) X = u32x4_t{0, 0, 0, 0};
      const f32x4_t o = Mma<T>::run(wfrag, X, f32x4_t{0.f, 0.f, 0.f, 0.f});
      const u32x2_t zr = *reinterpret_cast<const u32x2_t*>(zt + pl * PXB + kg * 8);
      const int y = y0 + 4 * wave + j, x = x0 + li;
      const bool okd = y < H && x < W;
      float f[8] = {o[0], o[1], o[2], o[3], 0.f, 0.f, 0.f, 0.f}, zf[8];
      u32x4_t pk = Vec16<T>::pack(f);
      Vec16<T>::unpack(pk, f);                        // sums are over the STORED (rounded) values
      Vec16<T>::unpack(u32x4_t{zr[0], zr[1], 0u, 0u}, zf);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (!okd || !(fmaf(zf[e], bsc[e], bsh[e]) > 0.f)) f[e] = 0.f;
        s1[e] += f[e];
        s2[e] += f[e] * zf[e];
      }
      pk = Vec16<T>::pack(f);
      if (okd) *reinterpret_cast<u32x2_t*>(dy + (((size_t)n * H + y) * W + x) * 16 + kg * 4) = u32x2_t{pk[0], pk[1]};
    }
    if (has_next) {                  // the other dlogits buffer was last read one iteration ago, before the barrier below
      float* dn = dt[(it + 1) & 1];
      dn[tid] = dr_n[0];
      if (tid + 256 < 324) dn[tid + 256] = dr_n[1];
      ar[0] = ar_n[0]; ar[1] = ar_n[1];
      ok = ok_n;
    }
    __syncthreads();
  }
  // ---- BN-backward sums: [replica][2][16]
  {
    float* bred = red[0];                     // [wave][32] inside red (148 floats per wave)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = row16_sum(s1[e]), b = row16_sum(s2[e]);
      if (li == 0) { red[wave][kg * 4 + e] = a; red[wave][16 + kg * 4 + e] = b; }
    }
    (void)bred;
    __syncthreads();
    if (tid < 32) {
      const float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
      atomicAdd(bnr_sums + (size_t)(blockIdx.x % VK_STATS_REPLICAS) * 32 + tid, (double)v);
    }
    __syncthreads();
  }
  // ---- weight gradient: lane holds D[tap = 4 kg + e][c = li]
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int tap = kg * 4 + e;
    if (tap < 9) red[wave][tap * 16 + li] = dwacc[e];
  }
  {
    const float v = wave_sum(gb);
    if (lane == 0) red[wave][144] = v;
  }
  __syncthreads();
  if (tid < 145) {
    const float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    if (part) part[(size_t)blockIdx.x * 148 + tid] = v;       // reproducible mode: k_head_wgrad_reduce adds the workgroups in order
    else if (tid < 144) atomicAdd(dw + tid, v);
    else atomicAdd(dbias, v);
  }
}

// dw[0..143] / dbias += sum over the nb per-workgroup partials in a fixed order (same bits on every run): one workgroup per
// 16 outputs, 16 lanes per output each adding every 16th partial, then the 16 lane sums in lane order
__global__ __launch_bounds__(256) void k_head_wgrad_reduce(int nb, const float* __restrict__ part, float* dw, float* dbias) {
  const int o = blockIdx.x * 16 + (threadIdx.x & 15), l = threadIdx.x >> 4;
  __shared__ float red[16][17];
  float s = 0.f;
  if (o < 145)
    for (int b = l; b < nb; b += 16) s += part[(size_t)b * 148 + o];
  red[l][threadIdx.x & 15] = s;
  __syncthreads();
  if (l == 0 && o < 145) {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) v += red[j][threadIdx.x & 15];
    if (o < 144) dw[o] += v;
    else *dbias += v;
  }
}

// ------------------------------------------------------------------------------------------------
// K10: BCE-with-logits (mean) + binary Dice (batch-global, smooth 0, eps 1e-7)
__global__ __launch_bounds__(256) void k_loss_reduce(size_t count, const float* __restrict__ x, const float* __restrict__ y,
                                                     double* sums, int vec_ok) {
  float bce = 0.f, py = 0.f, ps = 0.f, ys = 0.f;
  auto term = [&](float xv, float yv) {
    bce += fmaxf(xv, 0.f) - xv * yv + log1pf(expf(-fabsf(xv)));
    const float p = 1.f / (1.f + expf(-xv));
    py += p * yv;
    ps += p;
    ys += yv;
  };
  const size_t gtid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  const size_t n4 = vec_ok ? count / 4 : 0;          // 16-byte loads when both tensors are 16-byte aligned
  for (size_t i = gtid; i < n4; i += stride) {
    const f32x4_t xv = *reinterpret_cast<const f32x4_t*>(x + 4 * i), yv = *reinterpret_cast<const f32x4_t*>(y + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) term(xv[e], yv[e]);
  }
  for (size_t i = n4 * 4 + gtid; i < count; i += stride) term(x[i], y[i]);
  __shared__ double red[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double a = wave_sum_d((double)bce), b = wave_sum_d((double)py), c = wave_sum_d((double)ps), d = wave_sum_d((double)ys);
  if (lane == 0) { red[wave][0] = a; red[wave][1] = b; red[wave][2] = c; red[wave][3] = d; }
  __syncthreads();
  if (threadIdx.x < 4) atomicAdd(sums + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// sums[0..3] = {bce, p*y, p, y}; writes loss_out[0..2] and the two Dice gradient coefficients to sums[4..5]
__global__ void k_loss_finalize(double count, double* sums, float* loss_out, float w_bce, float w_dice) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double bce = sums[0] / count;
  const double I = sums[1], card = sums[2] + sums[3], ysum = sums[3];
  const double eps = 1e-7;
  const double denom = card > eps ? card : eps;
  const double mask = ysum > 0.0 ? 1.0 : 0.0;
  const double dice = (1.0 - 2.0 * I / denom) * mask;
  loss_out[0] = (float)(w_bce * bce + w_dice * dice);
  loss_out[1] = (float)bce;
  loss_out[2] = (float)dice;
  // d dice / d p_i = -2 (y_i * card - I) / card^2  (clamp inactive whenever mask == 1)
  sums[4] = card > eps ? -2.0 * w_dice * mask / card : 0.0;            // multiplies y_i
  sums[5] = card > eps ? 2.0 * w_dice * mask * I / (card * card) : 0.0;  // constant term
  sums[6] = w_bce / count;
}

__global__ void k_loss_bwd(size_t count, const float* __restrict__ x, const float* __restrict__ y, const double* __restrict__ sums,
                           float grad_scale, float* __restrict__ dl) {
  const float ky = (float)sums[4], k0 = (float)sums[5];
  const float invc = (float)sums[6];
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    const float xv = x[i], yv = y[i];
    const float p = 1.f / (1.f + expf(-xv));
    const float g = (p - yv) * invc + (ky * yv + k0) * p * (1.f - p);
    dl[i] = g * grad_scale;
  }
}

// ------------------------------------------------------------------------------------------------
// K15: AdamW over the flat fp32 buffers (same arithmetic order as torch/optim/adam.py single-tensor path)
template <typename LT>
__global__ void k_adamw(size_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                        float lr, float beta1, float beta2, float eps, float wd, float step_size, float bc2_sqrt,
                        float inv_scale, const int* found_inf, LT* lowp) {
  if (found_inf && *found_inf) return;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * inv_scale;
    float pi = p[i] * (1.f - lr * wd);
    float mi = m[i];
    mi = mi + (gi - mi) * (1.f - beta1);
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - step_size * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    if (lowp) st1(lowp + i, pi);
  }
}

__global__ void k_check_inf(size_t n, const float* __restrict__ g, int* found) {
  bool bad = false;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    bad |= !isfinite(g[i]);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(found, 1);
}

// K16, GradScaler form (torch._amp_foreach_non_finite_check_and_unscale_ over the ONE flat gradient buffer):
// g *= *inv_scale (skipped when the factor is exactly 1: GradScaler's check-only call passes a dummy 1.0), *found_inf = 1.0f if any
// element is inf / nan.  16-byte accesses; n4 = n / 4 (the flat buffer is padded to a multiple of 64 elements).
__global__ __launch_bounds__(256) void k_amp_unscale_check(size_t n4, float* __restrict__ g, const float* __restrict__ inv_scale,
                                                           float* __restrict__ found_inf) {
  const float inv = inv_scale ? *inv_scale : 1.f;
  const bool scale = inv != 1.f;
  bool bad = false;
  f32x4_t* g4 = reinterpret_cast<f32x4_t*>(g);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4_t v = g4[i];
    bad |= !(isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]) && isfinite(v[3]));
    if (scale) {
      v = v * inv;
      g4[i] = v;
    }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) *found_inf = 1.f;     // every writer stores the same value
}

// One thread: resolves the GradScaler protocol on the device for k_adamw_dev (no host round trip, train.py:443-445).
//   st[0] = skip flag (found_inf != 0), st[1] = lr / (1 - beta1^t), st[2] = sqrt(1 - beta2^t), st[3] = inv_scale / grad_scale;
//   the step counter t advances only when the step is taken (torch: optimizer.step() is not called on an overflow).
__global__ void k_adamw_prepare(int* __restrict__ step_count, const float* __restrict__ grad_scale, const float* __restrict__ found_inf,
                                float lr, float beta1, float beta2, float inv_scale, float* __restrict__ st) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const bool skip = found_inf && *found_inf != 0.f;
  int t = *step_count;
  if (!skip) {
    t += 1;
    *step_count = t;
  }
  const double bc1 = 1.0 - pow((double)beta1, (double)(t > 0 ? t : 1));
  const double bc2 = 1.0 - pow((double)beta2, (double)(t > 0 ? t : 1));
  st[0] = skip ? 1.f : 0.f;
  st[1] = (float)((double)lr / bc1);
  st[2] = (float)sqrt(bc2);
  st[3] = grad_scale ? (float)((double)inv_scale / (double)*grad_scale) : inv_scale;
}

template <typename LT>
__global__ void k_adamw_dev(size_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            float lr, float beta1, float beta2, float eps, float wd, const float* __restrict__ st, LT* lowp) {
  if (st[0] != 0.f) return;
  const float step_size = st[1], bc2_sqrt = st[2], inv_scale = st[3];
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * inv_scale;
    float pi = p[i] * (1.f - lr * wd);
    float mi = m[i];
    mi = mi + (gi - mi) * (1.f - beta1);
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - step_size * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    if (lowp) st1(lowp + i, pi);
  }
}

}  // namespace vk

// =================================================================================================
// C ABI wrappers
// =================================================================================================
using namespace vk;

#define DISPATCH_T(dt, CALL)                         \
  switch (dt) {                                      \
    case VK_F32: { using T = float; CALL; } break;   \
    case VK_BF16: { using T = bf16_t; CALL; } break; \
    case VK_F16: { using T = f16_t; CALL; } break;   \
    default: vkh::set_error("bad dtype %d", (int)dt); return VK_ERR_ARG; \
  }

extern "C" int vk_input_transform(vk_dtype dtype, int N, int H, int W, const float* x, void* x4, void* stream) {
  VK_CHECK_ARG(x && x4 && N > 0 && H > 0 && W > 0, "vk_input_transform: bad argument");
  hipStream_t st = (hipStream_t)stream;
  vkh::ProfScope ps_("input_transform", st, 0.0, (double)N * H * W * (12.0 + 4.0 * (dtype == VK_F32 ? 4.0 : 2.0)));
  VK_CHECK_ARG(N <= 65535, "vk_input_transform: N too large for the launch grid");
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_input_transform<T>, dim3(grid_for((size_t)H * W, 256, 1024), (unsigned)N), dim3(256), 0, st, N, H, W, x, (T*)x4));
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_bn_finalize(int C, int train, const double* stats, double count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float eps, float momentum, float* scale, float* shift,
                              float* save_mean, float* save_invstd, void* stream) {
  VK_CHECK_ARG(C > 0 && gamma && beta && scale && shift, "vk_bn_finalize: null argument");
  VK_CHECK_ARG(train ? (stats != nullptr && count > 0) : (running_mean && running_var), "vk_bn_finalize: missing statistics");
  vkh::ProfScope ps_("bn_finalize", (hipStream_t)stream, 0.0, (double)C * 40.0);
  hipLaunchKernelGGL(k_bn_finalize, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, C, train, stats, count, gamma, beta,
                     running_mean, running_var, eps, momentum, scale, shift, save_mean, save_invstd);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_bn_relu_maxpool(vk_dtype dtype, int N, int H, int W, int C, const void* z, const float* scale,
                                  const float* shift, void* pooled, uint8_t* argmax, void* stream) {
  VK_CHECK_ARG(z && scale && shift && pooled && argmax, "vk_bn_relu_maxpool: null argument");
  VK_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "vk_bn_relu_maxpool: H, W even and C %% 8 == 0 required");
  hipStream_t st = (hipStream_t)stream;
  vkh::ProfScope ps_("bn_relu_maxpool", st, 0.0, (double)N * H * W * C * (dtype == VK_F32 ? 4.0 : 2.0) * 1.25 + (double)N * H * W * C / 4.0);
  VK_CHECK_ARG(N <= 65535 && H / 2 <= 65535, "vk_bn_relu_maxpool: N or H too large for the launch grid");
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_bn_relu_maxpool<T>, dim3((unsigned)(((W / 2) * (C / ElemTraits<T>::kVec) + 255) / 256), (unsigned)(H / 2), (unsigned)N),
                                       dim3(256), 0, st, N, H, W, C, (const T*)z, scale, shift, (T*)pooled, argmax));
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_maxpool_bwd(vk_dtype dtype, int N, int H, int W, int C, const void* dpool, const uint8_t* argmax, void* dy,
                              void* stream) {
  VK_CHECK_ARG(dpool && argmax && dy && C % 8 == 0, "vk_maxpool_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  vkh::ProfScope ps_("maxpool_bwd", st, 0.0, (double)N * H * W * C * (dtype == VK_F32 ? 4.0 : 2.0) * 2.25 + (double)N * H * W * C / 4.0);
  VK_CHECK_ARG(N <= 65535 && H <= 65535 && H % 2 == 0 && W % 2 == 0, "vk_maxpool_bwd: N or H too large for the launch grid, or odd size");
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_maxpool_bwd<T>, dim3((unsigned)((W * (C / ElemTraits<T>::kVec) + 255) / 256), (unsigned)H, (unsigned)N), dim3(256), 0, st,
                                       N, H, W, C, (const T*)dpool, argmax, (T*)dy));
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_maxpool_bwd_bn_reduce(vk_dtype dtype, int N, int H, int W, int C, const void* dpool, const uint8_t* argmax, const void* z,
                                        const float* scale, const float* shift, void* dy, double* sums, void* stream) {
  VK_CHECK_ARG(dpool && argmax && z && scale && shift && dy && sums, "vk_maxpool_bwd_bn_reduce: null argument");
  VK_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "vk_maxpool_bwd_bn_reduce: H, W even and C %% 8 == 0 required");
  const int cv = C / (dtype == VK_F32 ? 4 : 8);
  VK_CHECK_ARG(cv <= 256 && 256 % cv == 0, "vk_maxpool_bwd_bn_reduce: C=%d unsupported", C);
  VK_CHECK_ARG(N <= 65535 && H <= 65535 * vk::PB_ROWS, "vk_maxpool_bwd_bn_reduce: N or H too large for the launch grid");
  hipStream_t st = (hipStream_t)stream;
  const double eb = dtype == VK_F32 ? 4.0 : 2.0;
  vkh::ProfScope ps_("maxpool_bwd_bn_reduce", st, 0.0, (double)N * H * W * C * eb * 3.25 + (double)N * H * W * C / 4.0);
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_maxpool_bwd_bn_reduce<T>, dim3((unsigned)((W * cv + 255) / 256), (unsigned)((H + PB_ROWS - 1) / PB_ROWS), (unsigned)N),
                                       dim3(256), 0, st, N, H, W, C, (const T*)dpool, argmax, (const T*)z, scale, shift, (T*)dy, sums));
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_bn_add_relu(vk_dtype dtype, size_t pixels, int C, const void* z, const float* scale, const float* shift,
                              const void* res, const float* rscale, const float* rshift, void* out, void* stream) {
  VK_CHECK_ARG(z && scale && shift && res && out && C % 8 == 0, "vk_bn_add_relu: bad argument");
  hipStream_t st = (hipStream_t)stream;
  vkh::ProfScope ps_("bn_add_relu", st, 0.0, (double)pixels * C * (dtype == VK_F32 ? 4.0 : 2.0) * 3.0);
  VK_CHECK_ARG(C <= 512, "vk_bn_add_relu: C=%d unsupported", C);
  // four vectors per thread (one unrolled pass) before the grid grows; the grid stays a multiple of C/VE threads
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_bn_add_relu<T>, dim3(grid_for((pixels * (C / ElemTraits<T>::kVec) + 3) / 4)), dim3(256), 0, st, pixels, C,
                                       (const T*)z, scale, shift, (const T*)res, rscale, rshift, (T*)out));
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

template <typename T>
static int launch_bn_bwd_reduce(size_t pixels, int C, const void* dy, const void* z, int mask_mode, const float* scale,
                                const float* shift, const void* mask_src, double* sums, hipStream_t st) {
  const int CV = C / ElemTraits<T>::kVec;
  const int rows = 256 / CV;
  // every block ends with 2*C fp64 atomics, so give each at least 16 row passes
  size_t nb = (pixels + (size_t)rows * 16 - 1) / ((size_t)rows * 16);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  dim3 grid((unsigned)nb), block(256);
  if (mask_mode == 0) hipLaunchKernelGGL((k_bn_bwd_reduce<T, 0>), grid, block, 0, st, pixels, C, (const T*)dy, (const T*)z, scale, shift, (const T*)mask_src, sums);
  else if (mask_mode == 1) hipLaunchKernelGGL((k_bn_bwd_reduce<T, 1>), grid, block, 0, st, pixels, C, (const T*)dy, (const T*)z, scale, shift, (const T*)mask_src, sums);
  else hipLaunchKernelGGL((k_bn_bwd_reduce<T, 2>), grid, block, 0, st, pixels, C, (const T*)dy, (const T*)z, scale, shift, (const T*)mask_src, sums);
  return VK_OK;
}

extern "C" int vk_bn_bwd_reduce(vk_dtype dtype, size_t pixels, int C, const void* dy, const void* z, int mask_mode,
                                const float* scale, const float* shift, const void* mask_src, double* sums, void* stream) {
  VK_CHECK_ARG(dy && z && sums, "vk_bn_bwd_reduce: null argument");
  VK_CHECK_ARG(mask_mode >= 0 && mask_mode <= 2, "vk_bn_bwd_reduce: mask_mode %d", mask_mode);
  VK_CHECK_ARG(mask_mode != 1 || (scale && shift), "vk_bn_bwd_reduce: mask_mode 1 needs scale/shift");
  VK_CHECK_ARG(mask_mode != 2 || mask_src, "vk_bn_bwd_reduce: mask_mode 2 needs mask_src");
  VK_CHECK_ARG(C % 8 == 0 && C <= 512, "vk_bn_bwd_reduce: C=%d unsupported", C);
  hipStream_t st = (hipStream_t)stream;
  vkh::ProfScope ps_("bn_bwd_reduce", st, 0.0, (double)pixels * C * (dtype == VK_F32 ? 4.0 : 2.0) * (mask_mode == 2 ? 3.0 : 2.0));
  DISPATCH_T(dtype, launch_bn_bwd_reduce<T>(pixels, C, dy, z, mask_mode, scale, shift, mask_src, sums, st));
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_bn_bwd_coeffs(int C, const double* sums, double count, const float* gamma, const float* save_mean,
                                const float* save_invstd, float* dgamma, float* dbeta, float* coef_abc, void* stream) {
  VK_CHECK_ARG(sums && gamma && save_mean && save_invstd && dgamma && dbeta && coef_abc, "vk_bn_bwd_coeffs: null argument");
  vkh::ProfScope ps_("bn_bwd_coeffs", (hipStream_t)stream, 0.0, (double)C * 48.0);
  hipLaunchKernelGGL(k_bn_bwd_coeffs, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, C, sums, count, gamma, save_mean,
                     save_invstd, dgamma, dbeta, coef_abc);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

struct BnApplyArgs {
  size_t pixels; int C; const void* dy; const void* z; int mask_mode; const float* scale; const float* shift; const void* mask_src;
  const float* coef; const double* sums; double count; const float* gamma; const float* mean; const float* invstd; float* dgamma; float* dbeta;
  void* dz; void* g_out; int g_acc;
};

template <typename T>
static int launch_bn_bwd_apply(const BnApplyArgs& a, hipStream_t st) {
  const int CV = a.C / ElemTraits<T>::kVec;
  // grid * 256 must be a multiple of CV (<= 128): any block count works; keep enough blocks to fill the chip
  dim3 grid(grid_for(a.pixels * CV / 2 + 1)), block(256);
#define VK_APPLY(M) hipLaunchKernelGGL((k_bn_bwd_apply<T, M>), grid, block, 0, st, a.pixels, a.C, (const T*)a.dy, (const T*)a.z, a.scale, a.shift, \
    (const T*)a.mask_src, a.coef, a.sums, a.count, a.gamma, a.mean, a.invstd, a.dgamma, a.dbeta, (T*)a.dz, (T*)a.g_out, a.g_acc)
  if (a.mask_mode == 0) VK_APPLY(0);
  else if (a.mask_mode == 1) VK_APPLY(1);
  else VK_APPLY(2);
#undef VK_APPLY
  return VK_OK;
}

static int bn_bwd_apply_common(vk_dtype dtype, const BnApplyArgs& a, void* stream) {
  VK_CHECK_ARG(a.dy && a.z && a.dz && (a.coef || (a.sums && a.gamma && a.mean && a.invstd && a.dgamma && a.dbeta)), "vk_bn_bwd_apply: null argument");
  VK_CHECK_ARG(a.mask_mode >= 0 && a.mask_mode <= 2, "vk_bn_bwd_apply: mask_mode %d", a.mask_mode);
  VK_CHECK_ARG(a.mask_mode != 1 || (a.scale && a.shift), "vk_bn_bwd_apply: mask_mode 1 needs scale/shift");
  VK_CHECK_ARG(a.mask_mode != 2 || a.mask_src, "vk_bn_bwd_apply: mask_mode 2 needs mask_src");
  VK_CHECK_ARG(a.C <= 512 && a.C % 8 == 0, "vk_bn_bwd_apply: C=%d unsupported", a.C);
  hipStream_t st = (hipStream_t)stream;
  vkh::ProfScope ps_("bn_bwd_apply", st, 0.0, (double)a.pixels * a.C * (dtype == VK_F32 ? 4.0 : 2.0) * ((a.mask_mode == 2 ? 4.0 : 3.0) + (a.g_out ? (a.g_acc ? 2.0 : 1.0) : 0.0)));
  DISPATCH_T(dtype, launch_bn_bwd_apply<T>(a, st));
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_bn_bwd_apply(vk_dtype dtype, size_t pixels, int C, const void* dy, const void* z, int mask_mode,
                               const float* scale, const float* shift, const void* mask_src, const float* coef_abc, void* dz,
                               void* g_out, int g_accumulate, void* stream) {
  VK_CHECK_ARG(coef_abc, "vk_bn_bwd_apply: null coefficients");
  BnApplyArgs a{pixels, C, dy, z, mask_mode, scale, shift, mask_src, coef_abc, nullptr, 1.0, nullptr, nullptr, nullptr, nullptr, nullptr, dz, g_out, g_accumulate};
  return bn_bwd_apply_common(dtype, a, stream);
}

extern "C" int vk_bn_bwd_apply_fused(vk_dtype dtype, size_t pixels, int C, const void* dy, const void* z, int mask_mode,
                                     const float* scale, const float* shift, const void* mask_src, const double* sums, double count,
                                     const float* gamma, const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                                     void* dz, void* g_out, int g_accumulate, void* stream) {
  BnApplyArgs a{pixels, C, dy, z, mask_mode, scale, shift, mask_src, nullptr, sums, count, gamma, save_mean, save_invstd, dgamma, dbeta, dz, g_out, g_accumulate};
  return bn_bwd_apply_common(dtype, a, stream);
}

extern "C" int vk_upsample2x_bwd(vk_dtype dtype, int N, int H, int W, int C, const void* d_up, void* d_low, int accumulate,
                                 void* stream) {
  VK_CHECK_ARG(d_up && d_low && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "vk_upsample2x_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  vkh::ProfScope ps_("upsample2x_bwd", st, 0.0, (double)N * H * W * C * (dtype == VK_F32 ? 4.0 : 2.0) * 1.25);
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_upsample2x_bwd<T>, dim3(grid_for((size_t)N * (H / 2) * (W / 2) * (C / ElemTraits<T>::kVec))),
                                       dim3(256), 0, st, N, H, W, C, (const T*)d_up, (T*)d_low, accumulate));
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_head_fwd(vk_dtype dtype, int N, int H, int W, const vk_src* src, const float* w9x16, const float* bias,
                           float* logits, void* stream) {
  VK_CHECK_ARG(src && src->ptr && w9x16 && bias && logits, "vk_head_fwd: null argument");
  VK_CHECK_ARG(src->C == 16 && !src->up, "vk_head_fwd: head input must have 16 channels, no upsample");
  hipStream_t st = (hipStream_t)stream;
  vkh::ProfScope ps_("head_fwd", st, 2.0 * 144.0 * N * H * W, (double)N * H * W * (16.0 * (dtype == VK_F32 ? 4.0 : 2.0) + 4.0));
  const int tx = (W + 15) / 16, ty = (H + 15) / 16;
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_head_fwd<T>, dim3((unsigned)(N * tx * ty)), dim3(256), 0, st, N, H, W, tx, ty, (const T*)src->ptr,
                                       src->scale, src->shift, src->relu, w9x16, bias, logits));
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

static int head_bwd_impl(vk_dtype dtype, int N, int H, int W, const vk_src* src, const float* w9x16, const float* dlogits,
                         void* dy, float* dw9x16, float* dbias, const vk_bnr* bnr, void* workspace, size_t workspace_bytes, void* stream);

extern "C" int vk_head_bwd(vk_dtype dtype, int N, int H, int W, const vk_src* src, const float* w9x16, const float* dlogits,
                           void* dy, float* dw9x16, float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
  return head_bwd_impl(dtype, N, H, W, src, w9x16, dlogits, dy, dw9x16, dbias, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int vk_head_bwd_fused(vk_dtype dtype, int N, int H, int W, const vk_src* src, const float* w9x16, const float* dlogits,
                                 void* dy, float* dw9x16, float* dbias, const vk_bnr* bnr, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  VK_CHECK_ARG(bnr && bnr->z && bnr->scale && bnr->shift && bnr->sums, "vk_head_bwd_fused: incomplete vk_bnr");
  return head_bwd_impl(dtype, N, H, W, src, w9x16, dlogits, dy, dw9x16, dbias, bnr, workspace, workspace_bytes, stream);
}

static int head_bwd_impl(vk_dtype dtype, int N, int H, int W, const vk_src* src, const float* w9x16, const float* dlogits,
                         void* dy, float* dw9x16, float* dbias, const vk_bnr* bnr, void* workspace, size_t workspace_bytes, void* stream) {
  VK_CHECK_ARG(src && src->ptr && w9x16 && dlogits && dy && dw9x16 && dbias, "vk_head_bwd: null argument");
  VK_CHECK_ARG(src->C == 16 && !src->up, "vk_head_bwd: head input must have 16 channels, no upsample");
  hipStream_t st = (hipStream_t)stream;
  const int tx = (W + 15) / 16, ty = (H + 15) / 16;
  const int ntiles = N * tx * ty;
  const double eb = dtype == VK_F32 ? 4.0 : 2.0;
  // 16-bit types with the fused reduce of the SAME tensor the head reads (what the training plan asks for): one pass on the matrix cores
  if (dtype != VK_F32 && bnr && bnr->z == src->ptr && bnr->scale == src->scale && bnr->shift == src->shift && src->scale && src->shift &&
      !getenv("VK_HEAD_NO_MFMA")) {
    vkh::ProfScope ps_("head_bwd_mfma", st, 4.0 * 144.0 * N * H * W, (double)N * H * W * (32.0 * eb + 4.0));
    int nb = ntiles < 1024 ? ntiles : 1024;
    float* part = nullptr;
    if (workspace && workspace_bytes >= 148 * sizeof(float)) {
      const size_t cap = workspace_bytes / (148 * sizeof(float));
      if ((size_t)nb > cap) nb = (int)cap;
      part = (float*)workspace;
    }
    if (dtype == VK_BF16)
      hipLaunchKernelGGL(k_head_bwd_mfma<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, N, H, W, tx, ty, ntiles, (const bf16_t*)src->ptr, src->scale,
                         src->shift, src->relu, w9x16, dlogits, (bf16_t*)dy, bnr->sums, dw9x16, dbias, part);
    else
      hipLaunchKernelGGL(k_head_bwd_mfma<f16_t>, dim3((unsigned)nb), dim3(256), 0, st, N, H, W, tx, ty, ntiles, (const f16_t*)src->ptr, src->scale,
                         src->shift, src->relu, w9x16, dlogits, (f16_t*)dy, bnr->sums, dw9x16, dbias, part);
    if (part) hipLaunchKernelGGL(k_head_wgrad_reduce, dim3(10), dim3(256), 0, st, nb, (const float*)part, dw9x16, dbias);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
  }
  {
    vkh::ProfScope ps_("head_dgrad", st, 2.0 * 144.0 * N * H * W, (double)N * H * W * (16.0 * eb + 4.0));
    const int nbd = ntiles < 2048 ? ntiles : 2048;
    DISPATCH_T(dtype, hipLaunchKernelGGL(k_head_dgrad<T>, dim3((unsigned)nbd), dim3(256), 0, st, N, H, W, tx, ty, ntiles, w9x16, dlogits, (T*)dy,
                                         (const T*)(bnr ? bnr->z : nullptr), bnr ? bnr->scale : nullptr, bnr ? bnr->shift : nullptr,
                                         bnr ? bnr->sums : nullptr));
  }
  {
    vkh::ProfScope ps_("head_wgrad", st, 2.0 * 144.0 * N * H * W, (double)N * H * W * (16.0 * eb + 4.0));
    int nb = ntiles < 1024 ? ntiles : 1024;
    float* part = nullptr;
    if (workspace && workspace_bytes >= 148 * sizeof(float)) {      // reproducible mode: per-workgroup partials + ordered reduce
      const size_t cap = workspace_bytes / (148 * sizeof(float));
      if ((size_t)nb > cap) nb = (int)cap;
      part = (float*)workspace;
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(k_head_wgrad<T>, dim3((unsigned)nb), dim3(256), 0, st, N, H, W, tx, ty, ntiles, (const T*)src->ptr,
                                         src->scale, src->shift, src->relu, dlogits, dw9x16, dbias, part));
    if (part) hipLaunchKernelGGL(k_head_wgrad_reduce, dim3(10), dim3(256), 0, st, nb, (const float*)part, dw9x16, dbias);
  }
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_bce_dice_loss(size_t count, const float* logits, const float* target, double* sums, float* loss_out,
                                float* dlogits, float grad_scale, float w_bce, float w_dice, void* stream) {
  VK_CHECK_ARG(count > 0 && logits && target && sums && loss_out, "vk_bce_dice_loss: null argument");
  hipStream_t st = (hipStream_t)stream;
  vkh::ProfScope ps_("bce_dice_loss", st, 0.0, (double)count * (dlogits ? 20.0 : 8.0));
  VK_CHECK_HIP(hipMemsetAsync(sums, 0, 8 * sizeof(double), st));
  const int vec_ok = (((uintptr_t)logits | (uintptr_t)target) & 15) == 0 ? 1 : 0;
  hipLaunchKernelGGL(k_loss_reduce, dim3(grid_for(vec_ok ? (count + 3) / 4 : count, 256, 2048)), dim3(256), 0, st, count, logits, target, sums, vec_ok);
  hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(64), 0, st, (double)count, sums, loss_out, w_bce, w_dice);
  if (dlogits) hipLaunchKernelGGL(k_loss_bwd, dim3(grid_for(count)), dim3(256), 0, st, count, logits, target, sums, grad_scale, dlogits);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_adamw_step(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, float inv_scale, const int* found_inf,
                             void* lowp_copy, vk_dtype lowp_dtype, void* stream) {
  VK_CHECK_ARG(n > 0 && param && grad && exp_avg && exp_avg_sq && step >= 1, "vk_adamw_step: bad argument");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  hipStream_t st = (hipStream_t)stream;
  vkh::ProfScope ps_("adamw", st, 0.0, (double)n * 28.0);
  dim3 grid(grid_for(n)), block(256);
  if (!lowp_copy || lowp_dtype == VK_F32) {
    hipLaunchKernelGGL(k_adamw<float>, grid, block, 0, st, n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay,
                       step_size, bc2_sqrt, inv_scale, found_inf, (float*)nullptr);
  } else if (lowp_dtype == VK_BF16) {
    hipLaunchKernelGGL(k_adamw<bf16_t>, grid, block, 0, st, n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay,
                       step_size, bc2_sqrt, inv_scale, found_inf, (bf16_t*)lowp_copy);
  } else {
    hipLaunchKernelGGL(k_adamw<f16_t>, grid, block, 0, st, n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay,
                       step_size, bc2_sqrt, inv_scale, found_inf, (f16_t*)lowp_copy);
  }
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_adamw_step_amp(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int* step_count, float inv_scale, const float* grad_scale,
                                 const float* found_inf, float* scratch4, void* lowp_copy, vk_dtype lowp_dtype, void* stream) {
  VK_CHECK_ARG(n > 0 && param && grad && exp_avg && exp_avg_sq && step_count && scratch4, "vk_adamw_step_amp: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_adamw_prepare, dim3(1), dim3(64), 0, st, step_count, grad_scale, found_inf, lr, beta1, beta2, inv_scale, scratch4);
  vkh::ProfScope ps_("adamw", st, 0.0, (double)n * 28.0);
  dim3 grid(grid_for(n)), block(256);
  if (!lowp_copy || lowp_dtype == VK_F32) {
    hipLaunchKernelGGL(k_adamw_dev<float>, grid, block, 0, st, n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay,
                       (const float*)scratch4, (float*)nullptr);
  } else if (lowp_dtype == VK_BF16) {
    hipLaunchKernelGGL(k_adamw_dev<bf16_t>, grid, block, 0, st, n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay,
                       (const float*)scratch4, (bf16_t*)lowp_copy);
  } else {
    hipLaunchKernelGGL(k_adamw_dev<f16_t>, grid, block, 0, st, n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay,
                       (const float*)scratch4, (f16_t*)lowp_copy);
  }
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_amp_unscale_check(size_t n, float* grad, const float* inv_scale, float* found_inf, void* stream) {
  VK_CHECK_ARG(n > 0 && grad && found_inf, "vk_amp_unscale_check: null argument");
  VK_CHECK_ARG((n & 3) == 0 && ((uintptr_t)grad & 15) == 0, "vk_amp_unscale_check: the gradient buffer must be 16-byte aligned with n %% 4 == 0");
  vkh::ProfScope ps_("amp_unscale_check", (hipStream_t)stream, 0.0, (double)n * 4.0);
  hipLaunchKernelGGL(k_amp_unscale_check, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, n / 4, grad, inv_scale, found_inf);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_amp_check_inf(size_t n, const float* grad, int* found_inf, void* stream) {
  VK_CHECK_ARG(n > 0 && grad && found_inf, "vk_amp_check_inf: null argument");
  hipLaunchKernelGGL(k_check_inf, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n, grad, found_inf);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}
