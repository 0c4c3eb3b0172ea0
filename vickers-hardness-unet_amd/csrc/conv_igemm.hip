// Implicit-GEMM convolution for gfx950: forward conv and data-gradient (transposed gather) in one
// kernel family.  Replaces the ATen conv2d / conv backward-data the reference dispatches to for
// every conv of smp.Unet(resnet34) (reference train.py:436 model(x), :443/:448 loss.backward()).
//
//   D[ch][px] = sum_k  W[ch][k] * V[px][k]          k = (tap, c)   ("swapped" orientation)
//
// * NHWC activations: for a fixed tap the C channels of a pixel are contiguous, so the A-operand
//   gather is 16-byte buffer loads (bounds-checked by the SRD => zero padding costs no branch).
// * the previous layer's BatchNorm scale/shift + ReLU and the decoder's nearest-x2 upsample +
//   channel concat are applied while the operand is staged into LDS (no materialised tensors).
// * 16x16 MFMA tiles (v_mfma_f32_16x16x32_bf16/f16, v_mfma_f32_16x16x4_f32), 4 waves / workgroup,
//   LDS rows padded by 32 B => conflict-free ds_read_b128 fragment reads.
// * swapped orientation: each lane ends with 4 consecutive output channels of one pixel, the tile is
//   transposed through LDS once and stored as full 16-byte NHWC vectors; the same pass produces the
//   per-channel sum / sum-of-squares partials for train-mode BatchNorm (fp64 atomics, 2 per channel
//   per workgroup).
#include <stdlib.h>

#include <string>

#include "vk_common.h"

namespace vk {

struct SrcDev {
  const void* ptr;
  const float* scale;
  const float* shift;
  int C, up, relu;
  uint32_t bytes;
};

struct ConvParams {
  SrcDev s0, s1;
  const void* w;
  uint32_t w_bytes;
  void* y0;
  void* y1;
  int ld0, ld1, split;
  double* stats;
  int N, H, W, Ho, Wo, K, R, S, slog, pad, transposed, accumulate;
  int C, M, RS, RSC, nchunks, log2C;
  FastDiv div_hw, div_w, div_s;
  // stride-2 data gradient (3x3): output pixels are enumerated parity class by parity class — (h & 1, w & 1) selects which of
  // the 9 taps can contribute at all (1, 2, 2 or 4 of them) — so a tile of one class walks only those taps instead of
  // zero-filling 3/4 of its operand rows: `parity` = pixels per class and image (0: plain row-major enumeration)
  int parity;
  FastDiv div_q, div_w2;
};

constexpr int kBK = 32;

template <typename T, int BM, int BN>
struct IgemmCfg {
  using Tr = ElemTraits<T>;
  static constexpr int VE = Tr::kVec;
  static constexpr int VPR = kBK / VE;
  static constexpr int RPP = 256 / VPR;
  static constexpr int APASS = BM / RPP;
  static constexpr int BPASS = (BN + RPP - 1) / RPP;
  static constexpr int RSB = kBK * Tr::kBytes + 32;
  static constexpr int STAGE = (BM + BN) * RSB;
  static constexpr int WGN = (BN >= 64) ? 2 : 1;
  static constexpr int WGM = 4 / WGN;
  static constexpr int WPX = BM / WGM;
  static constexpr int WCH = BN / WGN;
  static constexpr int TP = WPX / 16;
  static constexpr int TC = WCH / 16;
  static constexpr int ESB = BN * Tr::kBytes + 16;
  static constexpr int EVPR = BN / VE;
  static constexpr int ERPP = 256 / EVPR;
  static constexpr int EPASS = BM / ERPP;
  static constexpr int RED_OFF = BM * ESB;
  static constexpr int ESLOTS = (EVPR > 16) ? 16 / (EVPR / 16) : 16;      // [wave x 16-lane row] sets of channel partial sums
  static constexpr int RED_BYTES = ESLOTS * BN * 2 * 4;
  static constexpr int SMEM = (2 * STAGE > RED_OFF + RED_BYTES) ? 2 * STAGE : RED_OFF + RED_BYTES;
  static_assert(APASS >= 1 && TP >= 1 && TC >= 1, "tile too small");
  static_assert(EPASS >= 1, "epilogue mapping");
};

// MODE 0: channel chunk lies inside one tap (C % 32 == 0), loop order (channel chunk, tap)
// MODE 1: C == 16 (chunk = two taps), per-thread tap decode
// MODE 2: stem 7x7 s2 on NHWC4 input: chunk = one filter row (8 pixels x 4 channels)
template <typename T, int BM, int BN, int MODE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
  using Cfg = IgemmCfg<T, BM, BN>;
  using Tr = ElemTraits<T>;
  constexpr int VE = Cfg::VE, VPR = Cfg::VPR, RPP = Cfg::RPP, APASS = Cfg::APASS, BPASS = Cfg::BPASS;
  constexpr int RSB = Cfg::RSB, STAGE = Cfg::STAGE, TP = Cfg::TP, TC = Cfg::TC;
  constexpr int EB = Tr::kBytes;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(p.s0.ptr, p.s0.bytes);
  const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(p.s1.ptr ? p.s1.ptr : p.s0.ptr, p.s1.ptr ? p.s1.bytes : 0u);
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.w, p.w_bytes);

  // ---- per-thread A rows (output pixels) ----
  const int a_row = tid / VPR, a_vec = tid % VPR;
  int a_n[APASS], a_h0[APASS], a_w0[APASS];
  const int HoWo = p.Ho * p.Wo;
  // m -> (image, row, column); in parity order: m = ((n * 4 + class) * Q + yy * (Wo / 2) + xx), pixel (2 yy + ph, 2 xx + pw)
  auto decode = [&](int m, int& n, int& pp, int& q) {
    n = (int)fdiv((uint32_t)m, p.div_hw);
    const int rem = m - n * HoWo;
    if (p.parity) {
      const int cls = (int)fdiv((uint32_t)rem, p.div_q);
      const int r2 = rem - cls * p.parity;
      const int yy = (int)fdiv((uint32_t)r2, p.div_w2);
      const int xx = r2 - yy * (p.Wo >> 1);
      pp = 2 * yy + (cls >> 1);
      q = 2 * xx + (cls & 1);
    } else {
      pp = (int)fdiv((uint32_t)rem, p.div_w);
      q = rem - pp * p.Wo;
    }
  };
  // taps this tile has to walk: all of them, or (parity order, tile inside one class) those with r = ph + pad, s = pw + pad mod 2
  int tap_vr = -1, tap_vs = -1, nch = p.nchunks;
  if (p.parity && p.parity % BM == 0) {
    const int cls = (m0 % HoWo) / p.parity;
    tap_vr = ((cls >> 1) + p.pad) & 1;
    tap_vs = ((cls & 1) + p.pad) & 1;
    const int nr = (p.R + 1 - tap_vr) / 2, ns = (p.S + 1 - tap_vs) / 2;      // taps 0..R-1 of that parity
    nch = nr * ns * (p.C / kBK);
  }
  auto tap_valid = [&](int r, int s) { return tap_vr < 0 || ((((r ^ tap_vr) | (s ^ tap_vs)) & 1) == 0); };
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int m = m0 + a_row + i * RPP;
    if (m < p.M) {
      int n, pp, q;
      decode(m, n, pp, q);
      a_n[i] = n;
      if (MODE == 2) {
        a_h0[i] = 2 * pp - 3;
        a_w0[i] = 2 * q - 3;
      } else if (p.transposed) {
        a_h0[i] = pp + p.pad;
        a_w0[i] = q + p.pad;
      } else {
        a_h0[i] = (pp << p.slog) - p.pad;
        a_w0[i] = (q << p.slog) - p.pad;
      }
    } else {
      a_n[i] = 0;
      a_h0[i] = -(1 << 20);
      a_w0[i] = -(1 << 20);
    }
  }
  const int b_row = tid / VPR, b_vec = tid % VPR;

  // ---- staging registers ----
  u32x4_t areg[APASS], breg[BPASS];
  uint32_t vmask = 0;
  float sc[VE], sh[VE];
  bool pend_affine = false, pend_relu = false;
  int sc_cc = -1;
#pragma unroll
  for (int j = 0; j < VE; ++j) { sc[j] = 1.f; sh[j] = 0.f; }

  auto load_affine = [&](const SrcDev& sd, int cl) {
    pend_affine = sd.scale != nullptr;
    pend_relu = sd.relu != 0;
    if (pend_affine) {
      const float* sp = sd.scale + cl + a_vec * VE;
      const float* hp = sd.shift + cl + a_vec * VE;
#pragma unroll
      for (int j = 0; j < VE; j += 4) {
        const f32x4_t s4 = *reinterpret_cast<const f32x4_t*>(sp + j);
        const f32x4_t h4 = *reinterpret_cast<const f32x4_t*>(hp + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[j + e] = s4[e]; sh[j + e] = h4[e]; }
      }
    }
  };

  // loads one source's A vectors for tap (r, s), local channel cl
  auto load_a_src = [&](const SrcDev& sd, __amdgpu_buffer_rsrc_t rs, int cl, int r, int s, bool tap_ok) {
    const int up = sd.up;
    const int Hs = p.H >> up, Ws = p.W >> up;
    vmask = 0;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      int h, w;
      bool ok;
      if (!p.transposed) {
        h = a_h0[i] + r;
        w = a_w0[i] + s;
        ok = (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
      } else {
        const int th = a_h0[i] - r, tw = a_w0[i] - s;
        const int smask = (1 << p.slog) - 1;
        ok = (th >= 0) && (tw >= 0) && (((th | tw) & smask) == 0);
        h = th >> p.slog;
        w = tw >> p.slog;
        ok = ok && (h < p.H) && (w < p.W);
      }
      ok = ok && tap_ok;
      const uint32_t off =
          (uint32_t)(((a_n[i] * Hs + (h >> up)) * Ws + (w >> up)) * sd.C + cl + a_vec * VE) * (uint32_t)EB;
      areg[i] = buf_load16(rs, ok ? off : kOOB);
      vmask |= (ok ? 1u : 0u) << i;
    }
  };

  auto load_b = [&](int koff) {
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const int rl = b_row + i * RPP;
      const int nrow = n0 + rl;
      const bool ok = (rl < BN) && (nrow < p.K);
      const uint32_t off = (uint32_t)(nrow * p.RSC + koff + b_vec * VE) * (uint32_t)EB;
      breg[i] = buf_load16(rsw, ok ? off : kOOB);
    }
  };

  // chunk iteration state (MODE 0): channel chunk cc (outer), tap (inner)
  int ld_cc = 0, ld_tap = 0, ld_r = 0, ld_s = 0, ld_kc = 0;

  auto load_chunk = [&]() {
    if (MODE == 0) {
      const int c = ld_cc * kBK;
      const bool first = c < p.s0.C;
      if (first) {
        if (sc_cc != ld_cc) { load_affine(p.s0, c); sc_cc = ld_cc; }
        load_a_src(p.s0, rs0, c, ld_r, ld_s, true);
      } else {
        if (sc_cc != ld_cc) { load_affine(p.s1, c - p.s0.C); sc_cc = ld_cc; }
        load_a_src(p.s1, rs1, c - p.s0.C, ld_r, ld_s, true);
      }
      load_b(ld_tap * p.C + c);
      // advance to the next tap this tile walks
      do {
        ++ld_s;
        ++ld_tap;
        if (ld_s == p.S) { ld_s = 0; ++ld_r; }
        if (ld_tap == p.RS) { ld_tap = 0; ld_r = 0; ld_s = 0; ++ld_cc; }
      } while (!tap_valid(ld_r, ld_s));
    } else if (MODE == 1) {
      const int kelem = ld_kc * kBK + a_vec * VE;
      const int tap = kelem >> p.log2C;
      const int cl = kelem & (p.C - 1);
      const int r = (int)fdiv((uint32_t)tap, p.div_s);
      const int s = tap - r * p.S;
      if (sc_cc < 0) { load_affine(p.s0, cl - a_vec * VE); sc_cc = 0; }   // load_affine adds a_vec*VE back
      load_a_src(p.s0, rs0, cl - a_vec * VE, r, s, tap < p.RS);   // load_a_src adds a_vec*VE back
      load_b(ld_kc * kBK);
      ++ld_kc;
    } else {
      // stem: filter row r = ld_kc; 8 pixels x 4 channels = 32 contiguous elements
      const int r = ld_kc;
      pend_affine = false;
      constexpr int PPV = VE / 4;   // pixels per 16-byte vector: 2 (16-bit) or 1 (fp32)
      vmask = 0;
#pragma unroll
      for (int i = 0; i < APASS; ++i) {
        const int h = a_h0[i] + r;
        const bool okh = (unsigned)h < (unsigned)p.H;
        const int wb = a_w0[i] + a_vec * PPV;
        const uint32_t rowoff = (uint32_t)((a_n[i] * p.H + h) * p.W) * 4u * (uint32_t)EB;
        if (PPV == 1) {
          const bool ok = okh && (unsigned)wb < (unsigned)p.W;
          areg[i] = buf_load16(rs0, ok ? rowoff + (uint32_t)wb * 4u * EB : kOOB);
        } else {
          const bool ok0 = okh && (unsigned)wb < (unsigned)p.W;
          const bool ok1 = okh && (unsigned)(wb + 1) < (unsigned)p.W;
          const u32x2_t lo = __builtin_amdgcn_raw_buffer_load_b64(rs0, ok0 ? rowoff + (uint32_t)wb * 4u * EB : kOOB, 0, 0);
          const u32x2_t hi = __builtin_amdgcn_raw_buffer_load_b64(rs0, ok1 ? rowoff + (uint32_t)(wb + 1) * 4u * EB : kOOB, 0, 0);
          areg[i] = u32x4_t{lo[0], lo[1], hi[0], hi[1]};
        }
      }
      load_b(ld_kc * kBK);
      ++ld_kc;
    }
  };

  auto store_chunk = [&](int stage) {
    char* As = smem + stage * STAGE;
    char* Bs = As + BM * RSB;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      u32x4_t v = areg[i];
      if (pend_affine) {
        v = AffineRelu<T>::run(v, sc, sh, pend_relu);
        if (!((vmask >> i) & 1u)) v = u32x4_t{0, 0, 0, 0};
      }
      *reinterpret_cast<u32x4_t*>(As + (a_row + i * RPP) * RSB + a_vec * 16) = v;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const int rl = b_row + i * RPP;
      if (rl < BN) *reinterpret_cast<u32x4_t*>(Bs + rl * RSB + b_vec * 16) = breg[i];
    }
  };

  // ---- accumulators ----
  const int wpx0 = (wave / Cfg::WGN) * Cfg::WPX;
  const int wch0 = (wave % Cfg::WGN) * Cfg::WCH;
  f32x4_t acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int stage) {
    const char* As = smem + stage * STAGE;
    const char* Bs = As + BM * RSB;
    constexpr int KSTEPS = (EB == 4) ? 2 : 1;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int koff = ks * 64 + (lane >> 4) * 16;
      u32x4_t wf[TC], xf[TP];
#pragma unroll
      for (int a = 0; a < TC; ++a)
        wf[a] = *reinterpret_cast<const u32x4_t*>(Bs + (wch0 + a * 16 + (lane & 15)) * RSB + koff);
#pragma unroll
      for (int b = 0; b < TP; ++b)
        xf[b] = *reinterpret_cast<const u32x4_t*>(As + (wpx0 + b * 16 + (lane & 15)) * RSB + koff);
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[a][b] = Mma<T>::run(wf[a], xf[b], acc[a][b]);
    }
  };

  // ---- main loop: register-staged double buffer, one barrier per 32-deep K chunk ----
  if (nch == 0 && p.accumulate) return;   // a parity class no tap reaches (1x1 stride 2: three of the four): nothing to add
  if (MODE == 0 && nch > 0) {
    while (!tap_valid(ld_r, ld_s)) {      // first tap of this tile's parity class
      ++ld_s;
      ++ld_tap;
      if (ld_s == p.S) { ld_s = 0; ++ld_r; }
    }
  }
  if (nch > 0) {
    load_chunk();
    store_chunk(0);
  }
  __syncthreads();
  for (int kc = 0; kc < nch; ++kc) {
    const bool more = kc + 1 < nch;
    if (more) load_chunk();
    compute(kc & 1);
    if (more) store_chunk((kc + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS [px][ch] (as T) -> 16-byte NHWC stores (+ BN partial sums) ----
  constexpr int ESB = Cfg::ESB;
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      const int ch = wch0 + a * 16 + (lane >> 4) * 4;
      const int px = wpx0 + b * 16 + (lane & 15);
      char* dst = smem + px * ESB + ch * EB;
      if (EB == 4) {
        *reinterpret_cast<f32x4_t*>(dst) = acc[a][b];
      } else {
        float f[8] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3], 0.f, 0.f, 0.f, 0.f};
        const u32x4_t pk = Vec16<T>::pack(f);
        *reinterpret_cast<u32x2_t*>(dst) = u32x2_t{pk[0], pk[1]};
      }
    }
  __syncthreads();

  constexpr int EVPR = Cfg::EVPR, ERPP = Cfg::ERPP, EPASS = Cfg::EPASS;
  const int e_row = tid / EVPR, e_vec = tid % EVPR;
  const int col0 = n0 + e_vec * VE;
  const bool col_ok = col0 < p.K;
  char* yb = (char*)p.y0;
  int ld = p.ld0, colx = col0;
  if (p.split > 0 && col0 >= p.split) { yb = (char*)p.y1; ld = p.ld1; colx = col0 - p.split; }
  float s1[VE], s2[VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
  for (int ps = 0; ps < EPASS; ++ps) {
    const int row = e_row + ps * ERPP;
    const int m = m0 + row;
    if (m < p.M && col_ok) {
      u32x4_t v = *reinterpret_cast<const u32x4_t*>(smem + row * ESB + e_vec * 16);
      size_t mm = (size_t)m;                    // linear NHWC pixel index of this row
      if (p.parity) {
        int n, pp, q;
        decode(m, n, pp, q);
        mm = ((size_t)n * p.Ho + pp) * p.Wo + q;
      }
      u32x4_t* gp = reinterpret_cast<u32x4_t*>(yb + (mm * ld + colx) * EB);
      float f[VE];
      Vec16<T>::unpack(v, f);
      if (p.accumulate) {
        float o[VE];
        Vec16<T>::unpack(*gp, o);
#pragma unroll
        for (int j = 0; j < VE; ++j) f[j] += o[j];
        v = Vec16<T>::pack(f);
        Vec16<T>::unpack(v, f);
      }
#pragma unroll
      for (int j = 0; j < VE; ++j) { s1[j] += f[j]; s2[j] += f[j] * f[j]; }
      *gp = v;
    }
  }
  if (p.stats) {
    // same reduction as the tile kernels' epilogue (conv_halo.hip): DPP inside a 16-lane row, rows through LDS slots
#pragma unroll
    for (int j = 0; j < VE; ++j) {
      if (EVPR <= 2) { s1[j] += dpp_f<0x4E>(s1[j]); s2[j] += dpp_f<0x4E>(s2[j]); }       // quad_perm [2,3,0,1]
      if (EVPR <= 4) { s1[j] += dpp_f<0x124>(s1[j]); s2[j] += dpp_f<0x124>(s2[j]); }     // row_ror:4
      if (EVPR <= 8) { s1[j] += dpp_f<0x128>(s1[j]); s2[j] += dpp_f<0x128>(s2[j]); }     // row_ror:8
    }
    float* red = reinterpret_cast<float*>(smem + Cfg::RED_OFF);
    constexpr int RPS = EVPR > 16 ? EVPR / 16 : 1, NSLOT = 16 / RPS;
    static_assert(EVPR >= 2 && EVPR <= 32, "partial-sum slots");
    if (EVPR > 16 || (lane & 15) < EVPR) {
      const int slot = (wave * 4 + (lane >> 4)) / RPS;
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        red[(slot * BN + (lane % EVPR) * VE + j) * 2 + 0] = s1[j];
        red[(slot * BN + (lane % EVPR) * VE + j) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.K) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < NSLOT; ++w) { a += red[(w * BN + tid) * 2]; b += red[(w * BN + tid) * 2 + 1]; }
      double* sp = p.stats + (size_t)(blockIdx.x % VK_STATS_REPLICAS) * 2 * p.K;
      atomicAdd(sp + n0 + tid, (double)a);
      atomicAdd(sp + p.K + n0 + tid, (double)b);
    }
  }
}

// ------------------------------------------------------------------------------------------------ host
static int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return ((1 << l) == v) ? l : -1;
}

static SrcDev make_src(const vk_src& s, int N, int H, int W, int ebytes) {
  SrcDev d;
  d.ptr = s.ptr;
  d.scale = s.scale;
  d.shift = s.shift;
  d.C = s.C;
  d.up = s.up;
  d.relu = s.relu;
  d.bytes = s.ptr ? (uint32_t)((size_t)N * (H >> s.up) * (W >> s.up) * s.C * ebytes) : 0u;
  return d;
}

template <typename T, int BM, int BN, int MODE>
static int launch_cfg(const ConvParams& p, hipStream_t st) {
  using Cfg = IgemmCfg<T, BM, BN>;
  dim3 grid((p.M + BM - 1) / BM, (p.K + BN - 1) / BN, 1);
  static bool attr_done = false;
  if (!attr_done && Cfg::SMEM > 64 * 1024) {
    VK_CHECK_HIP(hipFuncSetAttribute((const void*)conv_igemm_kernel<T, BM, BN, MODE>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
    attr_done = true;
  }
  {
    static const std::string tag = std::string(MODE == 2 ? "stem_" : "igemm_") + (sizeof(T) == 4 ? "f32" : "16b") + "_bn" + std::to_string(BN) +
                                   (MODE == 1 ? "_c16" : "");
    static const std::string tag_t = tag + "_dgrad";
    // algorithmic work: real conv MACs (the stride-2 transposed gather executes stride^2 x more) and one read of the
    // input + weights, one write of the output
    const double macs = (double)p.M * p.K * (MODE == 2 ? 147.0 : (double)p.RSC) / (p.transposed ? (double)(1 << (2 * p.slog)) : 1.0);
    const double eb = sizeof(T);
    const double in_px = p.transposed ? (double)p.N * p.H * p.W : (double)p.N * p.H * p.W;
    const double bytes = (in_px * (MODE == 2 ? 3 : p.C) + (double)p.M * p.K + (double)p.K * (MODE == 2 ? 147.0 : (double)p.RSC)) * eb;
    const std::string& btag = p.transposed ? tag_t : tag;
    const std::string dtag = getenv("VK_PROF_DETAIL") ? btag + ":H" + std::to_string(p.H) + "_K" + std::to_string(p.K) + "_C" + std::to_string(p.C) + "_R" + std::to_string(p.R) : btag;
    vkh::ProfScope ps(dtag.c_str(), st, 2.0 * macs, bytes);
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, MODE>), grid, dim3(256), Cfg::SMEM, st, p);
  }
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

template <typename T, int MODE>
static int launch_bn(const ConvParams& p, hipStream_t st) {
  if (p.K >= 128) return launch_cfg<T, 128, 128, MODE>(p, st);
  if (p.K >= 64) return launch_cfg<T, 128, 64, MODE>(p, st);
  if (p.K >= 32) return launch_cfg<T, 128, 32, MODE>(p, st);
  return launch_cfg<T, 128, 16, MODE>(p, st);
}

template <typename T>
static int launch_mode(const ConvParams& p, int mode, hipStream_t st) {
  if (mode == 0) return launch_bn<T, 0>(p, st);
  if (mode == 1) {
    if (p.K >= 32) return launch_cfg<T, 128, 32, 1>(p, st);
    return launch_cfg<T, 128, 16, 1>(p, st);
  }
  return launch_cfg<T, 128, 64, 2>(p, st);
}

static int dispatch(vk_dtype dt, const ConvParams& p, int mode, hipStream_t st) {
  switch (dt) {
    case VK_F32: return launch_mode<float>(p, mode, st);
    case VK_BF16: return launch_mode<bf16_t>(p, mode, st);
    case VK_F16: return launch_mode<f16_t>(p, mode, st);
  }
  vkh::set_error("bad dtype %d", (int)dt);
  return VK_ERR_ARG;
}

int conv3x3_halo_try(const vk_conv_desc* d, const void* w, int packed, void* y, void* y1, int split_k1, int accumulate, double* stats,
                     int pool2, const vk_bnr* bnr, hipStream_t st, void* workspace, size_t workspace_bytes);
int halo_pack_impl(vk_dtype dt, int rows, int red, const void* src, void* dst, hipStream_t st);

static bool is_c16(const vk_conv_desc* d) {
  return d->dtype != VK_F32 && d->src0.C == 16 && !d->src1.ptr && !d->src0.up;
}

static bool halo_enabled() {       // read per call (diagnostic switch): a plan bound under VK_NO_HALO holds no halo packs
  return getenv("VK_NO_HALO") == nullptr;
}

int conv_fwd_impl(const vk_conv_desc* d, const void* w, int packed, void* y, void* y1, int split_k1, int accumulate,
                  double* stats, int pool2, const vk_bnr* bnr, hipStream_t st, void* workspace = nullptr, size_t workspace_bytes = 0) {
  VK_CHECK_ARG(d && w && y, "vk_conv_fwd: null argument");
  const int eb = d->dtype == VK_F32 ? 4 : 2;
  const int ve = 16 / eb;
  const int C = d->src0.C + (d->src1.ptr ? d->src1.C : 0);
  VK_CHECK_ARG(d->src0.ptr, "vk_conv_fwd: src0.ptr is null");
  VK_CHECK_ARG(d->stride == 1 || d->stride == 2, "vk_conv_fwd: stride %d unsupported", d->stride);
  VK_CHECK_ARG(d->K % ve == 0 && d->K >= 16, "vk_conv_fwd: K=%d must be a multiple of %d and >= 16", d->K, ve);
  VK_CHECK_ARG(!split_k1 || (y1 && split_k1 % ve == 0 && split_k1 < d->K), "vk_conv_fwd: bad split %d", split_k1);
  int mode;
  if (C % kBK == 0 && d->src0.C % kBK == 0) {
    mode = 0;
  } else {
    VK_CHECK_ARG(C == 16 && !d->src1.ptr, "vk_conv_fwd: C=%d unsupported (need multiple of 32, or 16 without concat)", C);
    mode = 1;
  }
  VK_CHECK_ARG(!d->src0.up || (d->H % 2 == 0 && d->W % 2 == 0), "vk_conv_fwd: upsampled source needs even H, W");
  VK_CHECK_ARG(!(d->src1.ptr && d->src1.up), "vk_conv_fwd: only src0 may be upsampled");
  if (halo_enabled()) {
    // 3x3 stride-1 convolutions (and their data gradients) and the stride-2 forward go to the LDS-staged tile kernels
    const int rc = conv3x3_halo_try(d, w, packed, y, y1, split_k1, accumulate, stats, pool2, bnr, st, workspace, workspace_bytes);
    if (rc != VK_ERR_UNSUPPORTED) return rc;
  }
  if (packed) {
    vkh::set_error("vk_conv_fwd_packed: shape not covered by the 3x3 stride-1 tile kernels (use vk_conv_fwd with plain weights)");
    return VK_ERR_UNSUPPORTED;
  }
  if (pool2 || bnr) {
    vkh::set_error("vk_conv_dgrad_fused: pooled / BN-fused outputs are only available on the 3x3 stride-1 tile kernels");
    return VK_ERR_UNSUPPORTED;
  }
  ConvParams p;
  p.s0 = make_src(d->src0, d->N, d->H, d->W, eb);
  if (d->src1.ptr) {
    p.s1 = make_src(d->src1, d->N, d->H, d->W, eb);
  } else {
    p.s1 = SrcDev{nullptr, nullptr, nullptr, 0, 0, 0, 0u};
  }
  const size_t in_bytes = (size_t)d->N * d->H * d->W * C * eb, out_bytes = (size_t)d->N * d->Ho * d->Wo * d->K * eb;
  VK_CHECK_ARG(in_bytes < (1ull << 31) && out_bytes < (1ull << 32), "vk_conv_fwd: tensor too large for 32-bit offsets");
  p.w = w;
  p.RS = d->R * d->S;
  p.RSC = p.RS * C;
  p.w_bytes = (uint32_t)((size_t)d->K * p.RSC * eb);
  p.y0 = y;
  p.y1 = y1;
  p.split = split_k1;
  p.ld0 = split_k1 ? split_k1 : d->K;
  p.ld1 = split_k1 ? d->K - split_k1 : 0;
  p.stats = stats;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo; p.K = d->K; p.R = d->R; p.S = d->S;
  p.slog = d->stride == 2 ? 1 : 0;
  p.pad = d->pad;
  p.transposed = d->transposed;
  p.accumulate = accumulate;
  p.C = C;
  p.M = d->N * d->Ho * d->Wo;
  p.nchunks = (p.RSC + kBK - 1) / kBK;
  p.log2C = mode == 1 ? ilog2_exact(C) : 0;
  p.div_hw = vkh::make_fastdiv((uint32_t)(d->Ho * d->Wo));
  p.div_w = vkh::make_fastdiv((uint32_t)d->Wo);
  p.div_s = vkh::make_fastdiv((uint32_t)d->S);
  p.parity = 0;
  p.div_q = p.div_w2 = vkh::make_fastdiv(1);
  if (mode == 0 && d->transposed && d->stride == 2 && ((d->R == 3 && d->S == 3 && d->pad == 1) || (d->R == 1 && d->S == 1 && d->pad == 0)) &&
      d->Ho % 2 == 0 && d->Wo % 2 == 0 && C % kBK == 0 &&
      !getenv("VK_IGEMM_NO_PARITY")) {
    p.parity = (d->Ho / 2) * (d->Wo / 2);
    p.div_q = vkh::make_fastdiv((uint32_t)p.parity);
    p.div_w2 = vkh::make_fastdiv((uint32_t)(d->Wo / 2));
  }
  VK_CHECK_ARG((size_t)d->N * d->Ho * d->Wo < (1ull << 31), "vk_conv_fwd: too many output pixels");
  return dispatch(d->dtype, p, mode, st);
}

int stem_tile_launch(vk_dtype dt, int N, int H, int W, const void* x4, const void* wp, void* y, double* stats, hipStream_t st);      // conv_halo.hip

int stem_fwd_impl(vk_dtype dt, int N, int H, int W, const void* x4, const void* wp, void* y, double* stats,
                  hipStream_t st) {
  VK_CHECK_ARG(x4 && wp && y, "vk_stem_fwd: null argument");
  VK_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "vk_stem_fwd: H, W must be even");
  {
    // 16-bit types: the input window of a 16x16 output tile staged once (stem7x7_kernel); fp32 stays on the tap-by-tap kernel
    const int rc = stem_tile_launch(dt, N, H, W, x4, wp, y, stats, st);
    if (rc != VK_ERR_UNSUPPORTED) return rc;
  }
  const int eb = dt == VK_F32 ? 4 : 2;
  ConvParams p;
  p.s0 = SrcDev{x4, nullptr, nullptr, 4, 0, 0, (uint32_t)((size_t)N * H * W * 4 * eb)};
  p.s1 = SrcDev{nullptr, nullptr, nullptr, 0, 0, 0, 0u};
  p.w = wp;
  p.RS = 7;
  p.RSC = 7 * 32;
  p.w_bytes = (uint32_t)(64 * p.RSC * eb);
  p.y0 = y; p.y1 = nullptr; p.split = 0; p.ld0 = 64; p.ld1 = 0;
  p.stats = stats;
  p.N = N; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2; p.K = 64; p.R = 7; p.S = 1;
  p.slog = 1; p.pad = 3; p.transposed = 0; p.accumulate = 0;
  p.C = 32;
  p.M = N * p.Ho * p.Wo;
  p.nchunks = 7;
  p.log2C = 0;
  p.div_hw = vkh::make_fastdiv((uint32_t)(p.Ho * p.Wo));
  p.div_w = vkh::make_fastdiv((uint32_t)p.Wo);
  p.div_s = vkh::make_fastdiv(1);
  p.parity = 0;
  p.div_q = p.div_w2 = vkh::make_fastdiv(1);
  return dispatch(dt, p, 2, st);
}

}  // namespace vk

extern "C" int vk_conv_fwd(const vk_conv_desc* d, const void* w, void* y, void* y1, int split_k1, int accumulate,
                           double* stats, void* stream) {
  return vk::conv_fwd_impl(d, w, 0, y, y1, split_k1, accumulate, stats, 0, nullptr, (hipStream_t)stream);
}

extern "C" int vk_conv_dgrad_pool2(const vk_conv_desc* d, const void* w, void* y_half, void* y1, int split_k1, int accumulate,
                                   void* stream) {
  return vk::conv_fwd_impl(d, w, vk::is_c16(d) ? 0 : 1, y_half, y1, split_k1, accumulate, nullptr, 1, nullptr, (hipStream_t)stream);
}

extern "C" int vk_conv_dgrad_fused(const vk_conv_desc* d, const void* w, void* y, void* y1, int split_k1, int pool2, const vk_bnr* bnr,
                                   void* stream) {
  VK_CHECK_ARG(!bnr || (bnr->z && bnr->sums && (bnr->mask || (bnr->scale && bnr->shift))), "vk_conv_dgrad_fused: incomplete vk_bnr");
  VK_CHECK_ARG(!bnr || !(bnr->mask && pool2), "vk_conv_dgrad_fused: the external mask does not combine with pool2");
  return vk::conv_fwd_impl(d, w, vk::is_c16(d) ? 0 : 1, y, y1, split_k1, bnr ? bnr->accumulate : 0, nullptr, pool2, bnr, (hipStream_t)stream);
}

extern "C" int vk_conv_fwd_packed(const vk_conv_desc* d, const void* w_halo, void* y, void* y1, int split_k1, int accumulate, double* stats,
                                  void* stream) {
  return vk::conv_fwd_impl(d, w_halo, 1, y, y1, split_k1, accumulate, stats, 0, nullptr, (hipStream_t)stream);
}

extern "C" int vk_conv_fwd_splitk(const vk_conv_desc* d, const void* w_halo, void* y, void* workspace, size_t workspace_bytes, void* stream) {
  return vk::conv_fwd_impl(d, w_halo, 1, y, nullptr, 0, 0, nullptr, 0, nullptr, (hipStream_t)stream, workspace, workspace_bytes);
}

extern "C" int vk_halo_pack(vk_dtype dtype, int rows, int red, const void* src, void* dst, void* stream) {
  return vk::halo_pack_impl(dtype, rows, red, src, dst, (hipStream_t)stream);
}

extern "C" int vk_conv_uses_halo_pack(const vk_conv_desc* d) {
  if (!d || d->R != 3 || d->S != 3 || d->pad != 1) return 0;
  const bool s2 = d->stride == 2 && !d->transposed && d->H == 2 * d->Ho && d->W == 2 * d->Wo && d->K >= 128 && !d->src0.up && !d->src1.ptr &&
                  !getenv("VK_NO_S2_TILE");                 // the stride-2 forward tile kernel (conv3x3_halo_try)
  const bool s2d = d->stride == 2 && d->transposed && d->Ho == 2 * d->H && d->Wo == 2 * d->W && d->K >= 64 && d->K % 64 == 0 && !d->src0.up &&
                   !d->src1.ptr && !d->src0.scale && !getenv("VK_NO_S2_TILE");      // its data gradient (conv3x3_s2dg_kernel)
  if (s2d && (size_t)d->N * d->Ho * d->Wo * d->K * (d->dtype == VK_F32 ? 4 : 2) >= (1ull << 32)) return 0;
  if (!s2 && !s2d && (d->stride != 1 || d->H != d->Ho || d->W != d->Wo)) return 0;
  if (vk::is_c16(d)) return 0;
  const int ck = d->dtype == VK_F32 ? 16 : 32;
  if (d->src0.C % ck || (d->src1.ptr && (d->src1.C % ck || d->src1.up)) || d->K % 16) return 0;
  const size_t C = (size_t)d->src0.C + (d->src1.ptr ? d->src1.C : 0), px = (size_t)d->N * d->H * d->W;
  if (px * C * (d->dtype == VK_F32 ? 4 : 2) >= (1ull << 31) || px >= (1ull << 31)) return 0;
  return getenv("VK_NO_HALO") ? 0 : 1;
}

extern "C" int vk_stem_fwd(vk_dtype dtype, int N, int H, int W, const void* x4, const void* wp, void* y,
                           double* stats, void* stream) {
  return vk::stem_fwd_impl(dtype, N, H, W, x4, wp, y, stats, (hipStream_t)stream);
}
