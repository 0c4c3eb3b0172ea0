// Weight gradient of the convolutions (replaces ATen conv backward-weight behind loss.backward(),
// reference train.py:443/:448):
//
//   dW[k][tap][c] += sum_{pixels m in this split}  dz[m][k] * V[m @ tap][c]
//
// A GEMM whose reduction runs over pixels.  Both operands are pixel-major in HBM (NHWC), i.e.
// "K-strided" for the MFMA; they are staged as they lie ([pixel][channel], 16-byte vectors, BN+ReLU /
// upsample / concat applied to V on the way) and the transposition happens in the LDS read:
//   16-bit : ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, two reads per fragment)
//   fp32   : ds_read_b32, lane = (channel, pixel-in-group) — already the 16x16x4 operand layout.
// Split-K over pixel ranges (grid.z): every split stores its partial tile into a slab of the caller's workspace and
// k_wgrad_slab_reduce (wgrad_halo.hip) adds the splits in a fixed order -> bit-reproducible gradients (fp32 atomics into
// the flat gradient buffer only when no workspace is given).
// Two wave layouts: big tiles split the output tile over the 4 waves; small-channel layers
// (K or C <= 32: decoder blocks 3/4, stem) give every wave the whole tile on its own 32-pixel slice.
#include <stdlib.h>

#include <string>

#include "vk_common.h"

namespace vk {

struct SrcDevW {
  const void* ptr;
  const float* scale;
  const float* shift;
  int C, up, relu;
  uint32_t bytes;
};

struct WgradParams {
  SrcDevW s0, s1;
  const void* dz;
  uint32_t dz_bytes;
  float* dw;
  float* slab;            // [splits][K * RS * C] partial results (nullptr: fp32 atomics into dw)
  int N, H, W, Ho, Wo, K, R, S, slog, pad;
  int C, M, RS, mps, ctiles, stem;
  FastDiv div_hw, div_w, div_ct, div_s;
};

template <typename T, int BMW, int BNW, bool WSPLIT>
struct WgradCfg {
  using Tr = ElemTraits<T>;
  static constexpr int VE = Tr::kVec;
  static constexpr int EB = Tr::kBytes;
  static constexpr int PX = WSPLIT ? 128 : 32;                 // pixels per chunk
  static constexpr int ZV = BMW / VE, VV = BNW / VE;            // vectors per pixel row
  static constexpr int ZPASS = (PX * ZV + 255) / 256, VPASS = (PX * VV + 255) / 256;
  // row pads chosen so that the transposed / b32 fragment reads are bank-conflict free
  static constexpr int zpad(int ch) { return EB == 2 ? (((ch * 2 / 32) % 2 == 0) ? 32 : 0) : ((ch % 32 == 0) ? 64 : 0); }
  static constexpr int ZSB = BMW * EB + zpad(BMW);
  static constexpr int VSB = BNW * EB + zpad(BNW);
  static constexpr int STAGE = PX * (ZSB + VSB);
  static constexpr int WGK = WSPLIT ? 1 : 2, WGC = WSPLIT ? 1 : 2;
  static constexpr int WK = BMW / WGK, WC = BNW / WGC;
  static constexpr int TK = WK / 16, TCc = WC / 16;
  static constexpr int SMEM = 2 * STAGE;
  static_assert(TK >= 1 && TCc >= 1, "tile too small");
};

template <typename T, int BMW, int BNW, bool WSPLIT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
  using Cfg = WgradCfg<T, BMW, BNW, WSPLIT>;
  constexpr int VE = Cfg::VE, EB = Cfg::EB, PX = Cfg::PX, ZV = Cfg::ZV, VV = Cfg::VV;
  constexpr int ZPASS = Cfg::ZPASS, VPASS = Cfg::VPASS, ZSB = Cfg::ZSB, VSB = Cfg::VSB, STAGE = Cfg::STAGE;
  constexpr int TK = Cfg::TK, TCc = Cfg::TCc;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = blockIdx.x * BMW;
  const int tap = (int)fdiv(blockIdx.y, p.div_ct);
  const int ct = blockIdx.y - tap * p.ctiles;
  const int c0 = ct * BNW;                       // channel in concat space
  const int r = (int)fdiv((uint32_t)tap, p.div_s), s = tap - r * p.S;
  const int mbeg = blockIdx.z * p.mps;
  const int mend = min(p.M, mbeg + p.mps);
  const int nch = (mend - mbeg + PX - 1) / PX;
  if (nch <= 0) return;

  const bool first = c0 < p.s0.C;
  const SrcDevW& sd = first ? p.s0 : p.s1;
  const int cl0 = first ? c0 : c0 - p.s0.C;
  const __amdgpu_buffer_rsrc_t rsv = make_rsrc(sd.ptr, sd.bytes);
  const __amdgpu_buffer_rsrc_t rsz = make_rsrc(p.dz, p.dz_bytes);
  const int up = sd.up;
  const int Hs = p.H >> up, Ws = p.W >> up;
  const int HoWo = p.Ho * p.Wo;
  const bool affine = sd.scale != nullptr;
  const bool relu = sd.relu != 0;

  // V-vector channel of this thread is fixed for the whole kernel -> scale/shift loaded once
  float sc[VPASS][VE], sh[VPASS][VE];
#pragma unroll
  for (int i = 0; i < VPASS; ++i) {
    const int v = tid + i * 256;
    const int vec = v % VV;
#pragma unroll
    for (int j = 0; j < VE; ++j) { sc[i][j] = 1.f; sh[i][j] = 0.f; }
    if (affine && v < PX * VV) {
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        sc[i][j] = sd.scale[cl0 + vec * VE + j];
        sh[i][j] = sd.shift[cl0 + vec * VE + j];
      }
    }
  }

  u32x4_t zreg[ZPASS], vreg[VPASS];
  uint32_t vmask = 0;

  auto load_chunk = [&](int ch) {
    const int mb = mbeg + ch * PX;
#pragma unroll
    for (int i = 0; i < ZPASS; ++i) {
      const int v = tid + i * 256;
      const int px = v / ZV, vec = v % ZV;
      const int m = mb + px;
      const bool ok = (v < PX * ZV) && (m < mend) && (k0 + vec * VE < p.K);
      const uint32_t off = (uint32_t)(m * p.K + k0 + vec * VE) * (uint32_t)EB;
      zreg[i] = buf_load16(rsz, ok ? off : kOOB);
    }
    vmask = 0;
#pragma unroll
    for (int i = 0; i < VPASS; ++i) {
      const int v = tid + i * 256;
      const int px = v / VV, vec = v % VV;
      const int m = mb + px;
      bool ok = (v < PX * VV) && (m < mend);
      const int n = (int)fdiv((uint32_t)m, p.div_hw);
      const int rem = m - n * HoWo;
      const int pp = (int)fdiv((uint32_t)rem, p.div_w);
      const int q = rem - pp * p.Wo;
      uint32_t off;
      if (!p.stem) {
        const int h = (pp << p.slog) - p.pad + r;
        const int w = (q << p.slog) - p.pad + s;
        ok = ok && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
        off = (uint32_t)(((n * Hs + (h >> up)) * Ws + (w >> up)) * sd.C + cl0 + vec * VE) * (uint32_t)EB;
        vreg[i] = buf_load16(rsv, ok ? off : kOOB);
      } else {
        // stem: "tap" = filter row, "channels" = 8 pixels x 4 channels along W
        constexpr int PPV = VE / 4;
        const int h = 2 * pp - 3 + tap;
        const int wb = 2 * q - 3 + vec * PPV;
        const bool okh = ok && (unsigned)h < (unsigned)p.H;
        const uint32_t rowoff = (uint32_t)((n * p.H + h) * p.W) * 4u * (uint32_t)EB;
        if (PPV == 1) {
          const bool o0 = okh && (unsigned)wb < (unsigned)p.W;
          vreg[i] = buf_load16(rsv, o0 ? rowoff + (uint32_t)wb * 4u * EB : kOOB);
        } else {
          const bool o0 = okh && (unsigned)wb < (unsigned)p.W;
          const bool o1 = okh && (unsigned)(wb + 1) < (unsigned)p.W;
          const u32x2_t lo = __builtin_amdgcn_raw_buffer_load_b64(rsv, o0 ? rowoff + (uint32_t)wb * 4u * EB : kOOB, 0, 0);
          const u32x2_t hi = __builtin_amdgcn_raw_buffer_load_b64(rsv, o1 ? rowoff + (uint32_t)(wb + 1) * 4u * EB : kOOB, 0, 0);
          vreg[i] = u32x4_t{lo[0], lo[1], hi[0], hi[1]};
        }
      }
      vmask |= (ok ? 1u : 0u) << i;
    }
  };

  auto store_chunk = [&](int stage) {
    char* Zs = smem + stage * STAGE;
    char* Vs = Zs + PX * ZSB;
#pragma unroll
    for (int i = 0; i < ZPASS; ++i) {
      const int v = tid + i * 256;
      if (v < PX * ZV) *reinterpret_cast<u32x4_t*>(Zs + (v / ZV) * ZSB + (v % ZV) * 16) = zreg[i];
    }
#pragma unroll
    for (int i = 0; i < VPASS; ++i) {
      const int v = tid + i * 256;
      u32x4_t x = vreg[i];
      if (affine) {
        float f[VE];
        Vec16<T>::unpack(x, f);
#pragma unroll
        for (int j = 0; j < VE; ++j) {
          f[j] = fmaf(f[j], sc[i][j], sh[i][j]);
          if (relu) f[j] = fmaxf(f[j], 0.f);
        }
        x = Vec16<T>::pack(f);
        if (!((vmask >> i) & 1u)) x = u32x4_t{0, 0, 0, 0};
      }
      if (v < PX * VV) *reinterpret_cast<u32x4_t*>(Vs + (v / VV) * VSB + (v % VV) * 16) = x;
    }
  };

  const int wk0 = WSPLIT ? 0 : (wave >> 1) * Cfg::WK;
  const int wc0 = WSPLIT ? 0 : (wave & 1) * Cfg::WC;
  const int wpx = WSPLIT ? wave * 32 : 0;        // this wave's pixel slice inside the chunk
  f32x4_t acc[TK][TCc];
#pragma unroll
  for (int a = 0; a < TK; ++a)
#pragma unroll
    for (int b = 0; b < TCc; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int stage) {
    const char* Zs = smem + stage * STAGE + wpx * ZSB;
    const char* Vs = smem + stage * STAGE + PX * ZSB + wpx * VSB;
    if (EB == 2) {
      // lane j of 16-lane group g supplies row (pixel) 4g + (j>>2) [+16 for the second read], columns 4*(j&3)..+3
      const int g = lane >> 4, j = lane & 15;
      const int prow = 4 * g + (j >> 2);
      const int cofs = 4 * (j & 3) * 2;
      u32x4_t zf[TK], vf[TCc];
      typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
#pragma unroll
      for (int a = 0; a < TK; ++a) {
        const char* b0 = Zs + prow * ZSB + (wk0 + a * 16) * 2 + cofs;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0 + 16 * ZSB));
        const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
        zf[a] = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
      }
#pragma unroll
      for (int b = 0; b < TCc; ++b) {
        const char* b0 = Vs + prow * VSB + (wc0 + b * 16) * 2 + cofs;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0 + 16 * VSB));
        const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
        vf[b] = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
      }
#pragma unroll
      for (int a = 0; a < TK; ++a)
#pragma unroll
        for (int b = 0; b < TCc; ++b) acc[a][b] = Mma<T>::run(zf[a], vf[b], acc[a][b]);
    } else {
      const int i = lane & 15, kg = lane >> 4;
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        float zf[TK], vf[TCc];
#pragma unroll
        for (int a = 0; a < TK; ++a)
          zf[a] = *reinterpret_cast<const float*>(Zs + (4 * st + kg) * ZSB + (wk0 + a * 16 + i) * 4);
#pragma unroll
        for (int b = 0; b < TCc; ++b)
          vf[b] = *reinterpret_cast<const float*>(Vs + (4 * st + kg) * VSB + (wc0 + b * 16 + i) * 4);
#pragma unroll
        for (int a = 0; a < TK; ++a)
#pragma unroll
          for (int b = 0; b < TCc; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(zf[a], vf[b], acc[a][b], 0, 0, 0);
      }
    }
  };

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    const bool more = ch + 1 < nch;
    if (more) load_chunk(ch + 1);
    compute(ch & 1);
    if (more) store_chunk((ch + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: dW[k][tap][c] (stem: [k][r][s][3], padding columns dropped): this split's slab, or fp32 atomics ----
  // (WSPLIT layouts give every wave the whole tile on its own pixel slice: their four partial tiles are added through LDS first)
  float* const out = p.slab ? p.slab + (size_t)blockIdx.z * ((size_t)p.K * (p.stem ? 147 : p.RS * p.C)) : p.dw;
  if (WSPLIT && p.slab) {
    float* red = reinterpret_cast<float*>(smem);           // [wave][TK][TCc][64 lanes][4]: at most 4 * 4 * 4 * 256 * 4 B = 64 KB <= 2 * STAGE
    static_assert(4 * TK * TCc * 1024 <= Cfg::SMEM || !WSPLIT, "wave-partial tiles must fit the staging buffers");
#pragma unroll
    for (int a = 0; a < TK; ++a)
#pragma unroll
      for (int b = 0; b < TCc; ++b) *reinterpret_cast<f32x4_t*>(red + (((wave * TK + a) * TCc + b) * 64 + lane) * 4) = acc[a][b];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int a = 0; a < TK; ++a)
#pragma unroll
        for (int b = 0; b < TCc; ++b) {
          f32x4_t t = acc[a][b];
#pragma unroll
          for (int w = 1; w < 4; ++w) t = t + *reinterpret_cast<const f32x4_t*>(red + (((w * TK + a) * TCc + b) * 64 + lane) * 4);
          acc[a][b] = t;
        }
    } else {
      return;
    }
  }
#pragma unroll
  for (int a = 0; a < TK; ++a)
#pragma unroll
    for (int b = 0; b < TCc; ++b) {
      const int cc = c0 + wc0 + b * 16 + (lane & 15);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = k0 + wk0 + a * 16 + (lane >> 4) * 4 + e;
        if (k >= p.K) continue;
        float* dst = nullptr;
        if (!p.stem) {
          if (cc < p.C) dst = out + ((size_t)k * p.RS + tap) * p.C + cc;
        } else {
          const int sx = cc >> 2, ci = cc & 3;
          if (sx < 7 && ci < 3) dst = out + (((size_t)k * 7 + tap) * 7 + sx) * 3 + ci;
        }
        if (!dst) continue;
        if (p.slab) *dst = acc[a][b][e];
        else atomicAdd(dst, acc[a][b][e]);
      }
    }
}

void launch_slab_reduce(size_t n4, int splits, const float* slab, float* dw, hipStream_t st);     // wgrad_halo.hip

// ------------------------------------------------------------------------------------------------ host
template <typename T, int BMW, int BNW, bool WSPLIT>
static int launch_w(WgradParams p, void* workspace, size_t workspace_bytes, hipStream_t st) {
  using Cfg = WgradCfg<T, BMW, BNW, WSPLIT>;
  p.ctiles = (p.C + BNW - 1) / BNW;
  p.div_ct = vkh::make_fastdiv((uint32_t)p.ctiles);
  const int ktiles = (p.K + BMW - 1) / BMW;
  const int out_tiles = ktiles * p.ctiles * p.RS;
  // split the pixel reduction so that ~4 workgroups per CU exist, each with at least 8 chunks
  int chunks = (p.M + Cfg::PX - 1) / Cfg::PX;
  int splits = (1024 + out_tiles - 1) / out_tiles;
  if (splits > chunks / 8) splits = chunks / 8;
  if (splits < 1) splits = 1;
  // reproducible mode: the splits' partial results go to a slab in the workspace ([splits][K * RS * C] floats) and are added in split order
  const size_t out_elems = (size_t)p.K * (p.stem ? 147 : (size_t)p.RS * p.C);
  const bool use_slab = workspace && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && out_elems % 4 == 0 && workspace_bytes >= out_elems * 4 &&
                        (reinterpret_cast<uintptr_t>(p.dw) & 15) == 0 && !getenv("VK_WGRAD_ATOMICS");
  if (use_slab) {
    const size_t cap = workspace_bytes / (out_elems * 4);
    if ((size_t)splits > cap) splits = (int)cap;
  }
  int cps = (chunks + splits - 1) / splits;
  p.mps = cps * Cfg::PX;
  splits = (p.M + p.mps - 1) / p.mps;
  p.slab = use_slab ? (float*)workspace : nullptr;
  dim3 grid(ktiles, p.ctiles * p.RS, splits);
  static bool attr_done = false;
  if (!attr_done && Cfg::SMEM > 64 * 1024) {
    VK_CHECK_HIP(hipFuncSetAttribute((const void*)conv_wgrad_kernel<T, BMW, BNW, WSPLIT>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
    attr_done = true;
  }
  {
    static const std::string tag_c = std::string("wgrad_") + (sizeof(T) == 4 ? "f32" : "16b") + "_" + std::to_string(BMW) + "x" + std::to_string(BNW);
    static const std::string tag_s = std::string("wgrad_stem_") + (sizeof(T) == 4 ? "f32" : "16b");
    const std::string& tag = p.stem ? tag_s : tag_c;
    const double rsc = p.stem ? 147.0 : (double)p.RS * p.C;
    const double eb = sizeof(T);
    const double bytes = ((double)p.N * p.H * p.W * (p.stem ? 3 : p.C) + (double)p.M * p.K) * eb + (double)p.K * rsc * 4.0;
    const std::string dtag = getenv("VK_PROF_DETAIL") ? tag + ":H" + std::to_string(p.H) + "_K" + std::to_string(p.K) + "_C" + std::to_string(p.C) + "_R" + std::to_string(p.R) + "_s" + std::to_string(splits) : tag;
    vkh::ProfScope ps(dtag.c_str(), st, 2.0 * (double)p.M * p.K * rsc, bytes);
    hipLaunchKernelGGL((conv_wgrad_kernel<T, BMW, BNW, WSPLIT>), grid, dim3(256), Cfg::SMEM, st, p);
  }
  if (use_slab) launch_slab_reduce(out_elems / 4, splits, p.slab, p.dw, st);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

template <typename T>
static int launch_w_shape(const WgradParams& p, int cmin, void* ws, size_t wsb, hipStream_t st) {
  // cmin = smallest per-source channel count (a c-tile must not straddle the concat boundary)
  if (p.K >= 128 && cmin % 128 == 0) return launch_w<T, 128, 128, false>(p, ws, wsb, st);
  if (p.K >= 64 && cmin % 64 == 0) return launch_w<T, 64, 64, false>(p, ws, wsb, st);
  if (p.K >= 64 && cmin % 32 == 0) return launch_w<T, 64, 32, true>(p, ws, wsb, st);
  if (p.K >= 32 && cmin % 64 == 0) return launch_w<T, 32, 64, true>(p, ws, wsb, st);
  if (p.K >= 32 && cmin % 32 == 0) return launch_w<T, 32, 32, true>(p, ws, wsb, st);
  if (cmin % 32 == 0) return launch_w<T, 16, 32, true>(p, ws, wsb, st);
  return launch_w<T, 16, 16, true>(p, ws, wsb, st);
}

static int dispatch_w(vk_dtype dt, const WgradParams& p, int cmin, void* ws, size_t wsb, hipStream_t st) {
  switch (dt) {
    case VK_F32: return launch_w_shape<float>(p, cmin, ws, wsb, st);
    case VK_BF16: return launch_w_shape<bf16_t>(p, cmin, ws, wsb, st);
    case VK_F16: return launch_w_shape<f16_t>(p, cmin, ws, wsb, st);
  }
  vkh::set_error("bad dtype %d", (int)dt);
  return VK_ERR_ARG;
}

static SrcDevW make_srcw(const vk_src& s, int N, int H, int W, int eb) {
  SrcDevW d;
  d.ptr = s.ptr; d.scale = s.scale; d.shift = s.shift; d.C = s.C; d.up = s.up; d.relu = s.relu;
  d.bytes = s.ptr ? (uint32_t)((size_t)N * (H >> s.up) * (W >> s.up) * s.C * eb) : 0u;
  return d;
}

int wgrad_halo_try(const vk_conv_desc* d, const void* dz, float* dw, void* workspace, size_t workspace_bytes, hipStream_t st);

int conv_wgrad_impl(const vk_conv_desc* d, const void* dz, float* dw, void* workspace, size_t workspace_bytes, hipStream_t st) {
  VK_CHECK_ARG(d && dz && dw && d->src0.ptr, "vk_conv_wgrad: null argument");
  VK_CHECK_ARG(!d->transposed, "vk_conv_wgrad: descriptor must describe the forward convolution");
  {
    const int rc = wgrad_halo_try(d, dz, dw, workspace, workspace_bytes, st);      // 3x3 stride-1 layers with enough pixels per output tile
    if (rc != VK_ERR_UNSUPPORTED) return rc;
  }
  const int eb = d->dtype == VK_F32 ? 4 : 2;
  const int C = d->src0.C + (d->src1.ptr ? d->src1.C : 0);
  VK_CHECK_ARG(d->K % 16 == 0 && C % 16 == 0, "vk_conv_wgrad: K=%d, C=%d must be multiples of 16", d->K, C);
  VK_CHECK_ARG(d->stride == 1 || d->stride == 2, "vk_conv_wgrad: stride %d unsupported", d->stride);
  const size_t in_bytes = (size_t)d->N * d->H * d->W * C * eb, dz_bytes = (size_t)d->N * d->Ho * d->Wo * d->K * eb;
  VK_CHECK_ARG(in_bytes < (1ull << 31) && dz_bytes < (1ull << 31), "vk_conv_wgrad: tensor too large for 32-bit offsets");
  WgradParams p;
  p.s0 = make_srcw(d->src0, d->N, d->H, d->W, eb);
  if (d->src1.ptr) p.s1 = make_srcw(d->src1, d->N, d->H, d->W, eb);
  else p.s1 = SrcDevW{nullptr, nullptr, nullptr, 0, 0, 0, 0u};
  p.dz = dz;
  p.dz_bytes = (uint32_t)dz_bytes;
  p.dw = dw;
  p.slab = nullptr;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo; p.K = d->K; p.R = d->R; p.S = d->S;
  p.slog = d->stride == 2 ? 1 : 0;
  p.pad = d->pad;
  p.C = C;
  p.M = d->N * d->Ho * d->Wo;
  p.RS = d->R * d->S;
  p.stem = 0;
  p.div_hw = vkh::make_fastdiv((uint32_t)(d->Ho * d->Wo));
  p.div_w = vkh::make_fastdiv((uint32_t)d->Wo);
  p.div_s = vkh::make_fastdiv((uint32_t)d->S);
  int cmin = d->src0.C;
  if (d->src1.ptr && d->src1.C < cmin) cmin = d->src1.C;
  if (d->src1.ptr) {
    // tiles must not straddle the concat boundary: use the gcd-like granularity of both sources
    int g = 128;
    while (g > 16 && (d->src0.C % g || d->src1.C % g)) g >>= 1;
    cmin = g;
  }
  return dispatch_w(d->dtype, p, cmin, workspace, workspace_bytes, st);
}

// ---- stem (7x7 stride 2, 3->64) weight gradient for the 16-bit types: one workgroup stages a 4 x 32 output-pixel tile of dz
// and the 13 x 72 input pixels under it ONCE and accumulates all seven filter rows from them (the tap-by-tap kernel above
// streams dz once per filter row: 7 passes over the 268 MB tensor).
//   dW[k][r][s][c] = sum_px dz[px][k] * x4[2y + r - 3][2x + s - 3][c]
// GEMM view per filter row r: A = dz^T (k x pixels), B[px][v] with v = (s, c) = the 64 contiguous bytes starting at input pixel
// (2y + r - 3, 2x - 3): overlapping windows of the staged rows, read by ds_read_b64_tr_b16 with per-lane addresses.
// Wave w owns output channels 16w..16w+15 and all 7 x 2 (s-half) tiles: 56 accumulator registers.  Persistent workgroups,
// double-buffered LDS, next tile's loads in flight (registers) under the MFMAs.  fp32 atomics at the end (as above).
struct StemWgParams {
  const void* x4;
  const void* dz;
  const void* z;          // BNA form: dz is the masked upstream gradient g, the operand is a*g + b*z + c (BatchNorm backward apply, folded in)
  const float* coef;      // [3][64]: a, b, c
  float* dw;
  float* slab;            // [gridDim.x][64 * 147] per-workgroup partial results (nullptr: fp32 atomics into dw)
  uint32_t x_bytes, dz_bytes;
  int N, H, W, Ho, Wo, tiles_x, tiles_y, ntiles;
};

// BNA (r04): the stem's BatchNorm-backward apply pass exists only to feed this kernel (the stem has no data gradient): instead of
// reading g and z (2 x 268 MB at bs 32), writing dz and reading it again here, the kernel reads g and z itself and forms
// dz = a*g + b*z + c (same fp32 expression and rounding as k_bn_bwd_apply) while staging — the 805 MB pass and its launch go
template <typename T, bool BNA>
__global__ __launch_bounds__(256) void k_stem_wgrad(const StemWgParams p) {
  static_assert(sizeof(T) == 2, "16-bit types only");
  constexpr int ZSB = 160;                   // dz pixel row: 64 k x 2 B + 32 pad (conflict-free transposed reads)
  constexpr int XW = 72, XROWS = 13;         // staged input: 13 rows x 72 pixels x 8 B
  constexpr int Z_BYTES = 128 * ZSB, X_BYTES = XROWS * XW * 8;
  constexpr int STAGE = (Z_BYTES + X_BYTES + 255) / 256 * 256;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const __amdgpu_buffer_rsrc_t rsx = make_rsrc(p.x4, p.x_bytes);
  const __amdgpu_buffer_rsrc_t rsz = make_rsrc(p.dz, p.dz_bytes);

  u32x4_t zr[4], zr_n[4];
  u32x4_t qr[BNA ? 4 : 1], qr_n[BNA ? 4 : 1];       // BNA: the z vectors beside the g vectors
  uint32_t okm = 0, okm_n = 0;                       // BNA: which of the four vectors lie inside the image (c must not leak into the padding)
  u32x2_t xr[4], xr_n[4];
  const __amdgpu_buffer_rsrc_t rsq = make_rsrc(BNA ? p.z : p.dz, p.dz_bytes);
  float ca[8], cb[8], cc[8];                         // this thread's 8 channels: vec = tid & 7 in every vector it stages
  if (BNA) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ca[j] = p.coef[(tid & 7) * 8 + j];
      cb[j] = p.coef[64 + (tid & 7) * 8 + j];
      cc[j] = p.coef[128 + (tid & 7) * 8 + j];
    }
  }
  auto fetch = [&](int t, u32x4_t (&zv)[4], u32x4_t (&qv)[BNA ? 4 : 1], uint32_t& okv, u32x2_t (&xv)[4]) {
    int bt = t;
    const int tx = bt % p.tiles_x;
    bt /= p.tiles_x;
    const int ty = bt % p.tiles_y;
    const int n = bt / p.tiles_y;
    const int y0 = ty * 4, x0 = tx * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      const int px = idx >> 3, vec = idx & 7;
      const int y = y0 + (px >> 5), x = x0 + (px & 31);
      const bool ok = y < p.Ho && x < p.Wo;
      const uint32_t off = ok ? (uint32_t)(((n * p.Ho + y) * p.Wo + x) * 64 + vec * 8) * 2u : kOOB;
      zv[i] = buf_load16(rsz, off);
      if (BNA) {
        qv[i] = buf_load16(rsq, off);
        if (i == 0) okv = 0;
        okv |= (ok ? 1u : 0u) << i;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      const int ry = idx / XW, rx = idx - ry * XW;
      const int iy = 2 * y0 - 3 + ry, ix = 2 * x0 - 3 + rx;
      const bool ok = idx < XROWS * XW && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      xv[i] = __builtin_amdgcn_raw_buffer_load_b64(rsx, ok ? (uint32_t)(((n * p.H + iy) * p.W + ix) * 4) * 2u : kOOB, 0, 0);
    }
  };
  auto stage = [&](int st, const u32x4_t (&zv)[4], const u32x4_t (&qv)[BNA ? 4 : 1], uint32_t okv, const u32x2_t (&xv)[4]) {
    char* Zs = smem + st * STAGE;
    char* Xs = Zs + Z_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      u32x4_t v = zv[i];
      if (BNA) {
        float g[8], zf[8], o[8];
        Vec16<T>::unpack(zv[i], g);
        Vec16<T>::unpack(qv[i], zf);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf(ca[j], g[j], fmaf(cb[j], zf[j], cc[j]));
        v = Vec16<T>::pack(o);
        if (!((okv >> i) & 1u)) v = u32x4_t{0, 0, 0, 0};
      }
      *reinterpret_cast<u32x4_t*>(Zs + (idx >> 3) * ZSB + (idx & 7) * 16) = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      if (idx < XROWS * XW) *reinterpret_cast<u32x2_t*>(Xs + idx * 8) = xv[i];
    }
  };

  f32x4_t acc[7][2];
#pragma unroll
  for (int r = 0; r < 7; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h) acc[r][h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // transposed-read lane geometry: lane j of 16-lane group g supplies pixel 4g + (j >> 2), elements 4 (j & 3) .. +3
  const int pl = 4 * (lane >> 4) + ((lane & 15) >> 2), q = lane & 3;
  const int lane_z = pl * ZSB + (wave * 16 + 4 * q) * 2;
  const int lane_x = (2 * pl + q) * 8;

  int t = blockIdx.x;
  if (t < p.ntiles) {
    fetch(t, zr, qr, okm, xr);
    stage(0, zr, qr, okm, xr);
  }
  __syncthreads();
  for (int it = 0; t < p.ntiles; t += gridDim.x, ++it) {
    const bool has_next = t + (int)gridDim.x < p.ntiles;
    if (has_next) fetch(t + gridDim.x, zr_n, qr_n, okm_n, xr_n);
    const char* Zs = smem + (it & 1) * STAGE;
    const char* Xs = Zs + Z_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                       // one output row = 32 pixels of reduction
      const char* zb = Zs + lane_z + (32 * ks) * ZSB;
      const u32x2_t zl = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(zb)));
      const u32x2_t zh = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(zb + 16 * ZSB)));
      const u32x4_t zf = u32x4_t{zl[0], zl[1], zh[0], zh[1]};
#pragma unroll
      for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const char* xb = Xs + lane_x + ((2 * ks + r) * XW + 4 * h) * 8;
          const u32x2_t xl = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(xb)));
          const u32x2_t xh = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(xb + 32 * 8)));
          acc[r][h] = Mma<T>::run(zf, u32x4_t{xl[0], xl[1], xh[0], xh[1]}, acc[r][h]);
        }
    }
    if (has_next) {
      stage((it + 1) & 1, zr_n, qr_n, okm_n, xr_n);        // that buffer was last read one iteration ago, before the barrier below
#pragma unroll
      for (int i = 0; i < 4; ++i) { zr[i] = zr_n[i]; xr[i] = xr_n[i]; }
    }
    __syncthreads();
  }
  // D[row = (lane >> 4) * 4 + e -> k][col = lane & 15 -> (s - 4h, c)]
#pragma unroll
  for (int r = 0; r < 7; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int sx = 4 * h + ((lane & 15) >> 2), ci = lane & 3;
      if (sx < 7 && ci < 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = wave * 16 + (lane >> 4) * 4 + e;
          const size_t off = (((size_t)k * 7 + r) * 7 + sx) * 3 + ci;
          if (p.slab) p.slab[(size_t)blockIdx.x * (64 * 147) + off] = acc[r][h][e];       // every workgroup writes all 9,408 elements
          else atomicAdd(p.dw + off, acc[r][h][e]);
        }
      }
    }
}

int stem_wgrad_impl(vk_dtype dt, int N, int H, int W, const void* x4, const void* dz, float* dw, void* workspace, size_t workspace_bytes,
                    hipStream_t st, const void* bn_z = nullptr, const float* bn_coef = nullptr) {
  VK_CHECK_ARG(x4 && dz && dw, "vk_stem_wgrad: null argument");
  const int eb = dt == VK_F32 ? 4 : 2;
  const bool tile = dt != VK_F32 && !getenv("VK_STEM_WGRAD_TAPS") && (size_t)N * (H / 2) * (W / 2) * 64 * 2 < (1ull << 31);
  if (bn_z && !tile) return VK_ERR_UNSUPPORTED;        // the folded BatchNorm apply lives in the 16-bit tile kernel only
  if (tile) {
    StemWgParams q;
    q.x4 = x4; q.dz = dz; q.dw = dw; q.z = bn_z; q.coef = bn_coef;
    q.N = N; q.H = H; q.W = W; q.Ho = H / 2; q.Wo = W / 2;
    q.x_bytes = (uint32_t)((size_t)N * H * W * 4 * 2);
    q.dz_bytes = (uint32_t)((size_t)N * q.Ho * q.Wo * 64 * 2);
    q.tiles_x = (q.Wo + 31) / 32; q.tiles_y = (q.Ho + 3) / 4;
    q.ntiles = N * q.tiles_y * q.tiles_x;
    int nb = q.ntiles < 512 ? q.ntiles : 512;
    const size_t out_elems = 64 * 147;
    q.slab = nullptr;
    if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0 &&
        workspace_bytes >= out_elems * 4 && !getenv("VK_WGRAD_ATOMICS")) {
      const size_t cap = workspace_bytes / (out_elems * 4);
      if ((size_t)nb > cap) nb = (int)cap;
      q.slab = (float*)workspace;
    }
    static const std::string tag = "wgrad_stem_16b", tag_bn = "wgrad_stem_16b_bn";
    vkh::ProfScope ps(bn_z ? tag_bn.c_str() : tag.c_str(), st, 2.0 * (double)N * q.Ho * q.Wo * 64.0 * 147.0,
                      ((double)N * H * W * 4 + (double)N * q.Ho * q.Wo * 64 * (bn_z ? 2 : 1)) * 2.0 + 64.0 * 147 * 4);
    if (bn_z) {
      if (dt == VK_BF16) hipLaunchKernelGGL((k_stem_wgrad<bf16_t, true>), dim3((unsigned)nb), dim3(256), 0, st, q);
      else hipLaunchKernelGGL((k_stem_wgrad<f16_t, true>), dim3((unsigned)nb), dim3(256), 0, st, q);
    } else {
      if (dt == VK_BF16) hipLaunchKernelGGL((k_stem_wgrad<bf16_t, false>), dim3((unsigned)nb), dim3(256), 0, st, q);
      else hipLaunchKernelGGL((k_stem_wgrad<f16_t, false>), dim3((unsigned)nb), dim3(256), 0, st, q);
    }
    if (q.slab) launch_slab_reduce(out_elems / 4, nb, q.slab, dw, st);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
  }
  WgradParams p;
  p.s0 = SrcDevW{x4, nullptr, nullptr, 4, 0, 0, (uint32_t)((size_t)N * H * W * 4 * eb)};
  p.s1 = SrcDevW{nullptr, nullptr, nullptr, 0, 0, 0, 0u};
  p.dz = dz;
  p.dz_bytes = (uint32_t)((size_t)N * (H / 2) * (W / 2) * 64 * eb);
  p.dw = dw;
  p.slab = nullptr;
  p.N = N; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2; p.K = 64; p.R = 7; p.S = 1;
  p.slog = 1; p.pad = 3;
  p.C = 32;
  p.M = N * p.Ho * p.Wo;
  p.RS = 7;
  p.stem = 1;
  p.div_hw = vkh::make_fastdiv((uint32_t)(p.Ho * p.Wo));
  p.div_w = vkh::make_fastdiv((uint32_t)p.Wo);
  p.div_s = vkh::make_fastdiv(1);
  switch (dt) {
    case VK_F32: return launch_w<float, 64, 32, true>(p, workspace, workspace_bytes, st);
    case VK_BF16: return launch_w<bf16_t, 64, 32, true>(p, workspace, workspace_bytes, st);
    case VK_F16: return launch_w<f16_t, 64, 32, true>(p, workspace, workspace_bytes, st);
  }
  return VK_ERR_ARG;
}

}  // namespace vk

extern "C" int vk_conv_wgrad(const vk_conv_desc* d, const void* dz, float* dw, void* workspace, size_t workspace_bytes, void* stream) {
  return vk::conv_wgrad_impl(d, dz, dw, workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" int vk_stem_wgrad(vk_dtype dtype, int N, int H, int W, const void* x4, const void* dz, float* dw_krsc3, void* workspace,
                             size_t workspace_bytes, void* stream) {
  return vk::stem_wgrad_impl(dtype, N, H, W, x4, dz, dw_krsc3, workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" int vk_stem_wgrad_bn(vk_dtype dtype, int N, int H, int W, const void* x4, const void* g, const void* z, const float* coef_abc,
                                float* dw_krsc3, void* workspace, size_t workspace_bytes, void* stream) {
  VK_CHECK_ARG(z && coef_abc, "vk_stem_wgrad_bn: null argument");
  return vk::stem_wgrad_impl(dtype, N, H, W, x4, g, dw_krsc3, workspace, workspace_bytes, (hipStream_t)stream, z, coef_abc);
}
