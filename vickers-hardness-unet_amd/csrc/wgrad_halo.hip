// Weight gradient of the 3x3 stride-1 pad-1 convolutions with LDS-staged tiles (reference train.py:443/:448,
// ATen conv backward-weight):
//
//   dW[k][r][s][c] += sum over pixel tiles  sum_{px in tile}  dz[px][k] * V[px + (r-1, s-1)][c]
//
// Per workgroup: a KT x CT x 9-tap output tile kept in registers (<= 36 MFMA tiles / wave) while the
// workgroup streams over 8x16 pixel tiles: per tile dz[128 px][KT] and the V halo [10x18 px][CT] (BN+ReLU /
// upsample / concat applied on load) are staged ONCE and feed all 9 taps (the tap-by-tap kernel re-reads
// both operands 9 times through L2).  For the shallow high-resolution layers (layer1/2, decoder) this makes
// the weight gradient read dz and V from HBM about once.
//   16-bit: ds_read_b64_tr_b16 transposed fragment reads;  fp32: ds_read_b32.
//   WS = false: 2x2 waves split the output tile (wave = (KT/2) x (CT/2) x 9);
//   WS = true : every wave owns the whole (small) output tile on its own 32-pixel slices; the four
//               partial results are combined with LDS atomics before the global fp32 atomics.
#include <stdlib.h>

#include <string>

#include "vk_common.h"

namespace vk {

struct WhSrc {
  const void* ptr;
  const float* scale;
  const float* shift;
  int C, up, relu;
  uint32_t bytes;
};

struct WhParams {
  WhSrc s0, s1;
  const void* dz;
  uint32_t dz_bytes;
  float* dw;
  int N, H, W, K, C;
  int tiles_x, tiles_y, ntiles, splits;
};

template <typename T, int KT, int CT, bool WS>
struct WhCfg {
  using Tr = ElemTraits<T>;
  static constexpr int EB = Tr::kBytes, VE = Tr::kVec;
  static constexpr int TH = 8, PX = 128, HPIX = 10 * 18;
  static constexpr int ZV = KT / VE, VV = CT / VE;
  static constexpr int ZPASS = (PX * ZV + 255) / 256, VPASS = (HPIX * VV + 255) / 256;
  static constexpr int zpad(int ch) { return EB == 2 ? (((ch * 2 / 32) % 2 == 0) ? 32 : 0) : ((ch % 32 == 0) ? 64 : 0); }
  static constexpr int ZSB = KT * EB + zpad(KT);
  static constexpr int VSB = CT * EB + zpad(CT);
  static constexpr int STAGE = PX * ZSB + HPIX * VSB;
  static constexpr int WK = WS ? KT : KT / 2, WC = WS ? CT : CT / 2;
  static constexpr int TK = WK / 16, TCc = WC / 16;
  static constexpr int RED = WS ? KT * CT * 9 * 4 : 0;
  static constexpr int SMEM = (2 * STAGE > RED) ? 2 * STAGE : RED;
  static_assert(TK >= 1 && TCc >= 1 && TK * TCc * 9 <= 36, "accumulator budget");
};

template <typename T, int KT, int CT, bool WS>
__global__ __launch_bounds__(256) void wgrad_halo_kernel(const WhParams p) {
  using Cfg = WhCfg<T, KT, CT, WS>;
  constexpr int EB = Cfg::EB, VE = Cfg::VE, PX = Cfg::PX, HPIX = Cfg::HPIX, ZV = Cfg::ZV, VV = Cfg::VV;
  constexpr int ZPASS = Cfg::ZPASS, VPASS = Cfg::VPASS, ZSB = Cfg::ZSB, VSB = Cfg::VSB, STAGE = Cfg::STAGE;
  constexpr int TK = Cfg::TK, TCc = Cfg::TCc;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = blockIdx.x * KT;
  const int c0 = blockIdx.y * CT;                 // channel in concat space
  const bool first = c0 < p.s0.C;
  const WhSrc& sd = first ? p.s0 : p.s1;
  const int cl0 = first ? c0 : c0 - p.s0.C;
  const __amdgpu_buffer_rsrc_t rsv = make_rsrc(sd.ptr, sd.bytes);
  const __amdgpu_buffer_rsrc_t rsz = make_rsrc(p.dz, p.dz_bytes);
  const int up = sd.up;
  const int Hs = p.H >> up, Ws = p.W >> up;
  const bool affine = sd.scale != nullptr;
  const bool relu = sd.relu != 0;

  // this thread's V vectors always cover the same channels -> scale/shift once
  float sc[VE], sh[VE];
  const int vvec = tid % VV;                       // VV divides 256
#pragma unroll
  for (int j = 0; j < VE; ++j) {
    sc[j] = affine ? sd.scale[cl0 + vvec * VE + j] : 1.f;
    sh[j] = affine ? sd.shift[cl0 + vvec * VE + j] : 0.f;
  }

  u32x4_t zreg[ZPASS], vreg[VPASS];
  uint32_t vmask = 0;

  auto load_tile = [&](int t) {
    int tt = t;
    const int tx = tt % p.tiles_x;
    tt /= p.tiles_x;
    const int ty = tt % p.tiles_y;
    const int n = tt / p.tiles_y;
    const int y0 = ty * 8, x0 = tx * 16;
#pragma unroll
    for (int i = 0; i < ZPASS; ++i) {
      const int v = tid + i * 256;
      const int px = v / ZV, vec = v % ZV;
      const int y = y0 + (px >> 4), x = x0 + (px & 15);
      const bool ok = (v < PX * ZV) && y < p.H && x < p.W && (k0 + vec * VE < p.K);
      const uint32_t off = (uint32_t)(((n * p.H + y) * p.W + x) * p.K + k0 + vec * VE) * (uint32_t)EB;
      zreg[i] = buf_load16(rsz, ok ? off : kOOB);
    }
    vmask = 0;
#pragma unroll
    for (int i = 0; i < VPASS; ++i) {
      const int v = tid + i * 256;
      const int hp = v / VV, vec = v % VV;
      const int hy = hp / 18, hx = hp - hy * 18;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = (v < HPIX * VV) && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      const uint32_t off = (uint32_t)(((n * Hs + (y >> up)) * Ws + (x >> up)) * sd.C + cl0 + vec * VE) * (uint32_t)EB;
      vreg[i] = buf_load16(rsv, ok ? off : kOOB);
      vmask |= (ok ? 1u : 0u) << i;
    }
  };

  auto store_tile = [&](int stage) {
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
          f[j] = fmaf(f[j], sc[j], sh[j]);
          if (relu) f[j] = fmaxf(f[j], 0.f);
        }
        x = Vec16<T>::pack(f);
        if (!((vmask >> i) & 1u)) x = u32x4_t{0, 0, 0, 0};
      }
      if (v < HPIX * VV) *reinterpret_cast<u32x4_t*>(Vs + (v / VV) * VSB + (v % VV) * 16) = x;
    }
  };

  const int wk0 = WS ? 0 : (wave >> 1) * Cfg::WK;
  const int wc0 = WS ? 0 : (wave & 1) * Cfg::WC;
  f32x4_t acc[9][TK][TCc];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int a = 0; a < TK; ++a)
#pragma unroll
      for (int b = 0; b < TCc; ++b) acc[t][a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

  // one 32-pixel reduction step = tile rows (2*ks, 2*ks+1)
  auto compute_step = [&](const char* Zs, const char* Vs, int ks) {
    if (EB == 2) {
      const int g = lane >> 4, j = lane & 15, q = j >> 2, pp = j & 3;
      u32x4_t zf[TK];
#pragma unroll
      for (int a = 0; a < TK; ++a) {
        const char* b0 = Zs + (32 * ks + 4 * g + q) * ZSB + (wk0 + a * 16 + 4 * pp) * 2;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0 + 16 * ZSB));
        const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
        zf[a] = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
      }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          u32x4_t vf[TCc];
#pragma unroll
          for (int b = 0; b < TCc; ++b) {
            const char* b0 = Vs + ((2 * ks + r) * 18 + 4 * g + q + s) * VSB + (wc0 + b * 16 + 4 * pp) * 2;
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0 + 18 * VSB));
            const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
            vf[b] = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
          }
#pragma unroll
          for (int a = 0; a < TK; ++a)
#pragma unroll
            for (int b = 0; b < TCc; ++b) acc[r * 3 + s][a][b] = Mma<T>::run(zf[a], vf[b], acc[r * 3 + s][a][b]);
        }
    } else {
      const int i = lane & 15, kg = lane >> 4;
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int px = 32 * ks + 4 * st + kg;          // tile pixel of this lane for this 4-pixel step
        const int row = px >> 4, x = px & 15;
        float zf[TK];
#pragma unroll
        for (int a = 0; a < TK; ++a) zf[a] = *reinterpret_cast<const float*>(Zs + px * ZSB + (wk0 + a * 16 + i) * 4);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            float vf[TCc];
#pragma unroll
            for (int b = 0; b < TCc; ++b)
              vf[b] = *reinterpret_cast<const float*>(Vs + ((row + r) * 18 + x + s) * VSB + (wc0 + b * 16 + i) * 4);
#pragma unroll
            for (int a = 0; a < TK; ++a)
#pragma unroll
              for (int b = 0; b < TCc; ++b)
                acc[r * 3 + s][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(zf[a], vf[b], acc[r * 3 + s][a][b], 0, 0, 0);
          }
      }
    }
  };

  // ---- stream over this workgroup's pixel tiles: t = blockIdx.z, + splits, ...
  int t = blockIdx.z;
  if (t < p.ntiles) {
    load_tile(t);
    store_tile(0);
  }
  __syncthreads();
  int it = 0;
  for (; t < p.ntiles; t += p.splits, ++it) {
    const bool more = t + p.splits < p.ntiles;
    if (more) load_tile(t + p.splits);
    const char* Zs = smem + (it & 1) * STAGE;
    const char* Vs = Zs + PX * ZSB;
    if (WS) {
      compute_step(Zs, Vs, wave);
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) compute_step(Zs, Vs, ks);
    }
    if (more) store_tile((it + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue
  if (WS) {
    float* red = reinterpret_cast<float*>(smem);
    for (int i = tid; i < KT * CT * 9; i += 256) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int a = 0; a < TK; ++a)
#pragma unroll
        for (int b = 0; b < TCc; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = a * 16 + (lane >> 4) * 4 + e, c = b * 16 + (lane & 15);
            atomicAdd(red + (k * 9 + tp) * CT + c, acc[tp][a][b][e]);
          }
    __syncthreads();
    for (int i = tid; i < KT * CT * 9; i += 256) {
      const int c = i % CT, kt = i / CT;
      const int k = kt / 9, tp = kt - k * 9;
      if (k0 + k < p.K) atomicAdd(p.dw + ((size_t)(k0 + k) * 9 + tp) * p.C + c0 + c, red[i]);
    }
  } else {
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int a = 0; a < TK; ++a)
#pragma unroll
        for (int b = 0; b < TCc; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = k0 + wk0 + a * 16 + (lane >> 4) * 4 + e, c = c0 + wc0 + b * 16 + (lane & 15);
            if (k < p.K) atomicAdd(p.dw + ((size_t)k * 9 + tp) * p.C + c, acc[tp][a][b][e]);
          }
  }
}

// ------------------------------------------------------------------------------------------------ host
// tuning knobs, read per call so that tests can force the kernel onto small problems
static int wh_max_combo() { const char* e = getenv("VK_WH_MAXCOMBO"); return e ? atoi(e) : 16; }
static int wh_min_blocks() { const char* e = getenv("VK_WH_MINBLOCKS"); return e ? atoi(e) : 256; }

template <typename T, int KT, int CT, bool WS>
static int launch_wh(WhParams p, hipStream_t st) {
  using Cfg = WhCfg<T, KT, CT, WS>;
  p.tiles_x = (p.W + 15) / 16;
  p.tiles_y = (p.H + 7) / 8;
  p.ntiles = p.N * p.tiles_y * p.tiles_x;
  const int kt = (p.K + KT - 1) / KT, ct = p.C / CT;
  if (kt * ct > wh_max_combo()) return VK_ERR_UNSUPPORTED;      // deep layers: output tile traffic would dominate
  // every workgroup ends with KT*CT*9 fp32 atomics: give it at least ~6 pixel tiles of work
  const char* e_blk = getenv("VK_WH_BLOCKS");
  const int target_blocks = e_blk ? atoi(e_blk) : (WS ? 1024 : 320);
  int splits = (target_blocks + kt * ct - 1) / (kt * ct);
  if (splits > p.ntiles / 6) splits = p.ntiles / 6;
  if (splits < 1) splits = 1;
  if ((long)splits * kt * ct < wh_min_blocks()) return VK_ERR_UNSUPPORTED;  // too few workgroups to fill the chip
  p.splits = splits;
  dim3 grid(kt, ct, splits);
  static bool attr_done = false;
  if (!attr_done && Cfg::SMEM > 64 * 1024) {
    VK_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_halo_kernel<T, KT, CT, WS>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
    attr_done = true;
  }
  {
    static const std::string tag = std::string("wgrad_halo_") + (sizeof(T) == 4 ? "f32" : "16b") + "_" + std::to_string(KT) + "x" + std::to_string(CT);
    const double bytes = (double)p.N * p.H * p.W * (p.C + p.K) * sizeof(T) + 9.0 * p.K * p.C * 4.0;
    const std::string dtag = getenv("VK_PROF_DETAIL") ? tag + ":H" + std::to_string(p.H) + "_K" + std::to_string(p.K) + "_C" + std::to_string(p.C) + "_s" + std::to_string(splits) : tag;
    vkh::ProfScope ps(dtag.c_str(), st, 2.0 * (double)p.N * p.H * p.W * p.K * 9.0 * p.C, bytes);
    hipLaunchKernelGGL((wgrad_halo_kernel<T, KT, CT, WS>), grid, dim3(256), Cfg::SMEM, st, p);
  }
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

template <typename T>
static int wh_select(const WhParams& p, int cgran, hipStream_t st) {
  // cgran: channel granularity that keeps a c-tile inside one concat source
  if constexpr (sizeof(T) == 2) {      // fp32 double-buffered 64x64 stages would need 197 KB of LDS
    if (p.K >= 64 && cgran % 64 == 0) return launch_wh<T, 64, 64, false>(p, st);
  }
  if (p.K >= 32 && cgran % 32 == 0) return launch_wh<T, 32, 32, true>(p, st);
  if (p.K >= 16 && cgran % 32 == 0) return launch_wh<T, 16, 32, true>(p, st);
  if (p.K >= 16 && cgran % 16 == 0) return launch_wh<T, 16, 16, true>(p, st);
  return VK_ERR_UNSUPPORTED;
}

// returns VK_ERR_UNSUPPORTED when the shape is not covered (caller falls back to the tap-by-tap kernel)
int wgrad_halo_try(const vk_conv_desc* d, const void* dz, float* dw, hipStream_t st) {
  if (d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->H != d->Ho || d->W != d->Wo) return VK_ERR_UNSUPPORTED;
  if (getenv("VK_NO_WGRAD_HALO")) return VK_ERR_UNSUPPORTED;
  const int eb = d->dtype == VK_F32 ? 4 : 2;
  const int C = d->src0.C + (d->src1.ptr ? d->src1.C : 0);
  if (d->K % 16 || d->src0.C % 16 || (d->src1.ptr && (d->src1.C % 16 || d->src1.up))) return VK_ERR_UNSUPPORTED;
  if ((size_t)d->N * d->H * d->W * C * eb >= (1ull << 31) || (size_t)d->N * d->H * d->W * d->K * eb >= (1ull << 31)) return VK_ERR_UNSUPPORTED;
  int cgran = 64;
  while (cgran > 16 && (d->src0.C % cgran || (d->src1.ptr && d->src1.C % cgran))) cgran >>= 1;
  WhParams p;
  auto mk = [&](const vk_src& s) {
    WhSrc h;
    h.ptr = s.ptr; h.scale = s.scale; h.shift = s.shift; h.C = s.C; h.up = s.up; h.relu = s.relu;
    h.bytes = s.ptr ? (uint32_t)((size_t)d->N * (d->H >> s.up) * (d->W >> s.up) * s.C * eb) : 0u;
    return h;
  };
  p.s0 = mk(d->src0);
  if (d->src1.ptr) p.s1 = mk(d->src1);
  else p.s1 = WhSrc{nullptr, nullptr, nullptr, 0, 0, 0, 0u};
  p.dz = dz;
  p.dz_bytes = (uint32_t)((size_t)d->N * d->H * d->W * d->K * eb);
  p.dw = dw;
  p.N = d->N; p.H = d->H; p.W = d->W; p.K = d->K; p.C = C;
  p.tiles_x = p.tiles_y = p.ntiles = p.splits = 0;
  switch (d->dtype) {
    case VK_F32: return wh_select<float>(p, cgran, st);
    case VK_BF16: return wh_select<bf16_t>(p, cgran, st);
    case VK_F16: return wh_select<f16_t>(p, cgran, st);
  }
  return VK_ERR_ARG;
}

}  // namespace vk
