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
#include <type_traits>

#include "vk_common.h"

// -DVK_WH_PIPE=1: the fragment reads of the 2x2-wave 16-bit tiles software-pipelined two (step, tap) units ahead (compute_tile_pipe) —
// built, bit-identical, 174 tests green, and measured 2-3 % SLOWER than the per-step form in same-box A/B runs through VK_LIB
// (profiles/r03/wgrad_pipe_experiment.log: 64x64 tap-split kernel 1.839 -> 1.902 ms per step): the partner wave already covers the
// LDS latency the compiler's schedule exposes; what bounds this kernel is the LDS pipe itself (1.2 transposed reads per MFMA + the
// staging writes: ~87 % of its cycles).  Default 0 = the per-step form.  Compile-time: both forms in one kernel cost 48 registers.
#ifndef VK_WH_PIPE
#define VK_WH_PIPE 0
#endif
// -DVK_WH_ZDMA=0: dz of the 64 x 64 tap-split configuration staged through registers + ds_write_b128 as in r03 (A/B builds through VK_LIB)
#ifndef VK_WH_XCD
#define VK_WH_XCD 1
#endif
#ifndef VK_WH_ZDMA
#define VK_WH_ZDMA 1
#endif

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
  int N, H, W, K, C;            // H, W: size of dz (= of the input for the stride-1 layers)
  int Hi, Wi;                   // size of the input V (stride 2: 2 H x 2 W)
  int tiles_x, tiles_y, ntiles, splits;
  int dbg_skip_epilogue;      // timing experiments only (VK_WH_DBG_NOEPI)
  float* slab;                // [splits][K][9][C] partial results (deterministic two-stage reduce) or nullptr (atomics)
};

// STR = 2 (the stride-2 3x3 openers of layers 2-4): the (2 * 8 + 1) x 33 input pixels under an 8 x 16 dz tile are staged with odd and
// even columns in separate planes of every LDS row (slot = row * 34 + (x & 1) * 17 + (x >> 1)), so the 16 pixels of a tile row of
// tap column s are 16 consecutive slots again; tile row b of filter row r reads halo row 2 b + r.
template <typename T, int KT, int CT, bool WS, bool TS = false, int STR = 1>
struct WhCfg {
  using Tr = ElemTraits<T>;
  static constexpr int EB = Tr::kBytes, VE = Tr::kVec;
  static constexpr int HWI = STR == 1 ? 18 : 33, HRS = STR == 1 ? 18 : 34, HHI = STR == 1 ? 10 : 17;      // staged columns, slots per row, rows
  static constexpr int TH = 8, PX = 128, HPIX = HHI * HWI;
  static constexpr int NT = TS ? 512 : 256;                    // TS: two wave groups share the staged tiles, taps 0-4 / 5-8
  static constexpr int NTAP = TS ? 5 : 9;                      // accumulated taps per wave
  static constexpr int ZV = KT / VE, VV = CT / VE;
  static constexpr int ZPASS = (PX * ZV + NT - 1) / NT, VPASS = (HPIX * VV + NT - 1) / NT;
  static constexpr int zpad(int ch) { return EB == 2 ? (((ch * 2 / 32) % 2 == 0) ? 32 : 0) : ((ch % 32 == 0) ? 64 : 0); }
  // ZDMA (r04, the 64 x 64 tap-split 16-bit configuration = the batched kernel; same-box A/B 1.522 -> 1.50 ms per step, after the
  // loop-carried vmcnt(0) drain described in the tile loop had been removed from both forms: 1.67 -> 1.52): dz needs no operand transform, so its 128 x 64 tile is
  // written straight into LDS by LDS-DMA (`buffer_load ... lds`: no staging registers, no ds_write, no VALU) as an UNPADDED image of
  // 128-byte pixel rows whose 32-byte slots are XOR-swizzled with (row >> 1) & 3 — the swizzle sits in the per-lane SOURCE address
  // (the DMA writes 1 KiB = 8 rows x 8 pieces linearly) and in the transposed fragment reads, which stay conflict-free: the eight
  // pixel rows one 32-lane group of a ds_read_b64_tr_b16 touches land in eight different 32-byte slots of the 256-byte bank row.
  static constexpr bool ZDMA = VK_WH_ZDMA && TS && !WS && EB == 2 && STR == 1 && KT == 64;
  static constexpr int ZSB = ZDMA ? KT * EB : KT * EB + zpad(KT);
  static constexpr int VSB = CT * EB + zpad(CT);
  static constexpr int STAGE = PX * ZSB + HHI * HRS * VSB;
  static constexpr int WK = WS ? KT : KT / 2, WC = WS ? CT : CT / 2;
  static constexpr int TK = WK / 16, TCc = WC / 16;
  static constexpr int RED = KT * CT * 9 * 4;      // fp32 [KT][9][CT] output tile staged for coalesced stores
  static constexpr int NSTAGE = 2;
  static constexpr int SMEM = (NSTAGE * STAGE > RED) ? NSTAGE * STAGE : RED;
  static_assert(TK >= 1 && TCc >= 1 && TK * TCc * NTAP <= 36, "accumulator budget");
  static_assert(!(TS && WS), "tap split is for the 2x2 wave layout");
  static_assert(STR == 1 || (STR == 2 && EB == 2 && !WS), "stride 2: 16-bit types, 2x2 wave layout");
  static_assert(NSTAGE * STAGE <= 160 * 1024 && RED <= 160 * 1024, "LDS image exceeds the 160 KiB of a CU");
};

// The body of the kernel for ONE segment of work: output tile (k0, c0) over the pixel tiles t0, t0 + t_step, ... < t_end; the result
// tile goes to dst[(k * 9 + tap) * ldc + c] (plain stores: a slab or a per-segment partial tile) or, with dst == nullptr, to p.dw by
// fp32 atomics.  wgrad_halo_kernel runs one segment per workgroup (strided tiles); wgrad_halo_batch_kernel runs the 1-3 segments of
// a contiguous range of (layer, output tile, pixel tile) units.
template <typename T, int KT, int CT, bool WS, bool TS, int STR = 1>
__device__ __forceinline__ void wh_segment(const WhParams& p, char* smem, const int k0, const int c0, const int t0, const int t_end,
                                           const int t_step, float* const dst, const int ldc) {
  using Cfg = WhCfg<T, KT, CT, WS, TS, STR>;
  constexpr int HWI = Cfg::HWI, HRS = Cfg::HRS;
  constexpr int NT = Cfg::NT, NTAP = Cfg::NTAP;
  constexpr int EB = Cfg::EB, VE = Cfg::VE, PX = Cfg::PX, HPIX = Cfg::HPIX, ZV = Cfg::ZV, VV = Cfg::VV;
  constexpr int ZPASS = Cfg::ZPASS, VPASS = Cfg::VPASS, ZSB = Cfg::ZSB, VSB = Cfg::VSB, STAGE = Cfg::STAGE;
  constexpr int TK = Cfg::TK, TCc = Cfg::TCc;

  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;
  const int half = __builtin_amdgcn_readfirstlane(tid >> 8);   // tap group (TS); wave-uniform by construction
  const bool first = c0 < p.s0.C;                 // c0: channel in concat space
  const WhSrc& sd = first ? p.s0 : p.s1;
  const int cl0 = first ? c0 : c0 - p.s0.C;
  const __amdgpu_buffer_rsrc_t rsv = make_rsrc(sd.ptr, sd.bytes);
  const __amdgpu_buffer_rsrc_t rsz = make_rsrc(p.dz, p.dz_bytes);
  const int up = STR == 1 ? sd.up : 0;
  const int Hs = p.Hi >> up, Ws = p.Wi >> up;
  const bool affine = sd.scale != nullptr;
  const bool relu = sd.relu != 0;
  constexpr bool ZDMA = Cfg::ZDMA;
  static_assert(!(ZDMA && VK_WH_PIPE), "the pipelined fragment reads do not know the swizzled dz image");
  typedef __attribute__((address_space(3))) void lds_void;

  // this thread's V vectors always cover the same channels -> scale/shift once
  float sc[VE], sh[VE];
  const int vvec = tid % VV;                       // VV divides NT
#pragma unroll
  for (int j = 0; j < VE; ++j) {
    sc[j] = affine ? sd.scale[cl0 + vvec * VE + j] : 1.f;
    sh[j] = affine ? sd.shift[cl0 + vvec * VE + j] : 0.f;
  }

  // staging registers: one set (tile t+1 in flight under the MFMAs of tile t) or, in the small-channel WS configurations whose
  // tiles are a few MFMAs long, two sets (tiles t+1 and t+2 in flight: an HBM round trip is longer than one tile's work)
  // (also the stride-2 configuration: 36 MFMAs per wave and tile against 52 KB of staging)
  // measured and rejected (r02): two sets for the 64x64 tap-split configuration as well (layer 2-4 / dec0 shapes unchanged within 1 %)
  constexpr int DEPTH = (WS || KT <= 32 || STR == 2) ? 2 : 1;
  // measured and rejected (r02): the next tile's LDS writes between the 32-pixel steps instead of after the last one — L1 / L2 shapes
  // 60 -> 66 us (the writes then wait for HBM loads that the remaining steps used to cover), others unchanged
  constexpr bool STORE_MID = false;
  u32x4_t zreg[DEPTH][ZPASS], vreg[DEPTH][VPASS];
  uint32_t vmask[DEPTH] = {};

  // ---- per-thread staging geometry relative to the tile origin, computed once
  int z_rel[ZPASS], z_yx[ZPASS];                 // element offset (py*W + px)*K + k, packed (py << 8 | px), -1 = unused slot
  int v_rel[VPASS], v_yx[VPASS];                 // element offset in the (possibly half-res) source, packed ((hy) << 8 | hx)
#pragma unroll
  for (int i = 0; i < ZPASS; ++i) {
    const int v = tid + i * NT;
    const int px = v / ZV, vec = v % ZV;
    const bool ok = (v < PX * ZV) && (k0 + vec * VE < p.K);
    z_rel[i] = ((px >> 4) * p.W + (px & 15)) * p.K + k0 + vec * VE;
    z_yx[i] = ok ? (((px >> 4) << 8) | (px & 15)) : -1;
  }
  // ZDMA: wave w issues the 1 KiB pieces 2 w and 2 w + 1 of the tile (8 pixel rows each); lane l supplies row (l >> 3), LDS piece (l & 7),
  // i.e. the 8 channels of piece (l & 7) ^ (((row >> 1) & 3) << 1) — (row >> 1) & 3 = (l >> 4) & 3 for every piece (8 rows per piece)
  const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const i32x4_t rsz_words = make_rsrc_words(p.dz, p.dz_bytes);
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  int zd_rel[2], zd_yx[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (2 * w8 + j) * 8 + (lane >> 3);
    const int cp = (lane & 7) ^ (((lane >> 4) & 3) << 1);
    zd_rel[j] = ((row >> 4) * p.W + (row & 15)) * p.K + k0 + cp * VE;
    zd_yx[j] = (k0 + cp * VE < p.K) ? (((row >> 4) << 8) | (row & 15)) : -1;
  }
  auto dma_z = [&](int t, int stage) {            // dz tile t -> the dz image of `stage` (every wave: 2 LDS-DMA instructions)
    int tt = t;
    const int tx = tt % p.tiles_x;
    tt /= p.tiles_x;
    const int ty = tt % p.tiles_y;
    const int n = tt / p.tiles_y;
    const int y0 = ty * 8, x0 = tx * 16;
    const int zbase = ((n * p.H + y0) * p.W + x0) * p.K;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = (zd_yx[j] >= 0) & (y0 + (zd_yx[j] >> 8) < p.H) & (x0 + (zd_yx[j] & 255) < p.W);
      const uint32_t dst = smem_base + (uint32_t)(stage * STAGE + (2 * w8 + j) * 1024);
      // hidden from hipcc (vk_common.h): with the builtin the compiler drains vmcnt(0) in front of the first transposed fragment read
      lds_dma16_hidden(rsz_words, ok ? (uint32_t)(zbase + zd_rel[j]) * (uint32_t)EB : kOOB, (uint32_t)__builtin_amdgcn_readfirstlane((int)dst));
    }
  };
#pragma unroll
  for (int i = 0; i < VPASS; ++i) {
    const int v = tid + i * NT;
    const int hp = v / VV, vec = v % VV;
    const int hy = hp / HWI, hx = hp - hy * HWI;
    const bool ok = v < HPIX * VV;
    v_rel[i] = (((hy - 1) >> up) * Ws + ((hx - 1) >> up)) * sd.C + cl0 + vec * VE;     // tile origins are even
    v_yx[i] = ok ? ((hy << 8) | hx) : -1;
  }

  auto load_tile = [&](int t, u32x4_t (&zreg)[ZPASS], u32x4_t (&vreg)[VPASS], uint32_t& vmask) {
    int tt = t;
    const int tx = tt % p.tiles_x;
    tt /= p.tiles_x;
    const int ty = tt % p.tiles_y;
    const int n = tt / p.tiles_y;
    const int y0 = ty * 8, x0 = tx * 16;
    const int zbase = ((n * p.H + y0) * p.W + x0) * p.K;                                // block-uniform
    const int vbase = ((n * Hs + ((STR * y0) >> up)) * Ws + ((STR * x0) >> up)) * sd.C;
    if constexpr (!ZDMA) {
#pragma unroll
      for (int i = 0; i < ZPASS; ++i) {
        const bool ok = (z_yx[i] >= 0) & (y0 + (z_yx[i] >> 8) < p.H) & (x0 + (z_yx[i] & 255) < p.W);      // & : no branches between the loads
        zreg[i] = buf_load16(rsz, ok ? (uint32_t)(zbase + z_rel[i]) * (uint32_t)EB : kOOB);
      }
    }
    vmask = 0;
#pragma unroll
    for (int i = 0; i < VPASS; ++i) {
      const int y = STR * y0 - 1 + (v_yx[i] >> 8), x = STR * x0 - 1 + (v_yx[i] & 255);
      const bool ok = (v_yx[i] >= 0) & ((unsigned)y < (unsigned)p.Hi) & ((unsigned)x < (unsigned)p.Wi);
      vreg[i] = buf_load16(rsv, ok ? (uint32_t)(vbase + v_rel[i]) * (uint32_t)EB : kOOB);
      vmask |= (ok ? 1u : 0u) << i;
    }
  };

  auto store_tile = [&](int stage, const u32x4_t (&zreg)[ZPASS], const u32x4_t (&vreg)[VPASS], uint32_t vmask) {
    char* Zs = smem + stage * STAGE;
    char* Vs = Zs + PX * ZSB;
    if constexpr (!ZDMA) {
#pragma unroll
      for (int i = 0; i < ZPASS; ++i) {
        const int v = tid + i * NT;
        if (v < PX * ZV) *reinterpret_cast<u32x4_t*>(Zs + (v / ZV) * ZSB + (v % ZV) * 16) = zreg[i];
      }
    }
#pragma unroll
    for (int i = 0; i < VPASS; ++i) {
      const int v = tid + i * NT;
      u32x4_t x = vreg[i];
      if (affine) {
        x = AffineRelu<T>::run(x, sc, sh, relu);
        if (!((vmask >> i) & 1u)) x = u32x4_t{0, 0, 0, 0};
      }
      int slot = v / VV;
      if (STR == 2) {                                      // odd / even input columns in separate planes of the row
        const int hy = slot / HWI, hx = slot - hy * HWI;
        slot = hy * HRS + (hx & 1) * 17 + (hx >> 1);
      }
      if (v < HPIX * VV) *reinterpret_cast<u32x4_t*>(Vs + slot * VSB + (v % VV) * 16) = x;
    }
  };

  const int wk0 = WS ? 0 : (wave >> 1) * Cfg::WK;
  const int wc0 = WS ? 0 : (wave & 1) * Cfg::WC;
  f32x4_t acc[NTAP][TK][TCc];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int a = 0; a < TK; ++a)
#pragma unroll
      for (int b = 0; b < TCc; ++b) acc[t][a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
  // transposed-read lane geometry: lane j of 16-lane group g supplies pixel 4g + (j>>2), channels 4*(j&3)..+3
  const int lane_z = ZDMA ? (4 * (lane >> 4) + ((lane & 15) >> 2)) * ZSB + 8 * (lane & 3)
                          : (4 * (lane >> 4) + ((lane & 15) >> 2)) * ZSB + (wk0 + 4 * (lane & 3)) * 2;
  // ZDMA: the 32-byte slot of channel tile (wk0 / 16 + a) in this lane's pixel rows — row = 16 h + 4 (lane >> 4) + ((lane & 15) >> 2)
  // (+ 32 ks), so (row >> 1) & 3 depends on the lane alone
  int zslot[TK];
#pragma unroll
  for (int a = 0; a < TK; ++a) zslot[a] = ((((wk0 >> 4) + a) ^ ((2 * (lane >> 4) + ((lane & 15) >> 3)) & 3)) << 5);
  const int lane_v = (4 * (lane >> 4) + ((lane & 15) >> 2)) * VSB + (wc0 + 4 * (lane & 3)) * 2;

  // one 32-pixel reduction step = tile rows (2*ks, 2*ks+1)
  // hsel: 0 = all nine taps (no tap split), 1 = taps 0-4, 2 = taps 5-8.  Compile-time, chosen ONCE outside the tile loop, so that
  // the four 32-pixel steps of a tile form one basic block and the fragment reads of step k+1 can be scheduled under the MFMAs
  // of step k (a wave-uniform branch inside every step cut the schedule into four blocks, each with its own read-latency bubble)
  auto compute_step = [&](const char* Zs, const char* Vs, int ks, auto hsel_c) {
    constexpr int HSEL = decltype(hsel_c)::value;
    if (EB == 2) {
      const char* const Zl = Zs + lane_z;      // fragment reads are lane base + immediate
      const char* const Vl = Vs + lane_v;
      u32x4_t zf[TK];
#pragma unroll
      for (int a = 0; a < TK; ++a) {
        const char* b0 = ZDMA ? Zl + zslot[a] + (32 * ks) * ZSB : Zl + (32 * ks) * ZSB + (a * 16) * 2;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0 + 16 * ZSB));
        const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
        zf[a] = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
      }
      auto taps16 = [&](auto tap0_c, auto ntap_c) {
        constexpr int TAP0 = decltype(tap0_c)::value, NTP = decltype(ntap_c)::value;
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
          constexpr int dummy = 0;
          (void)dummy;
          const int tap = TAP0 + t;
          const int r = tap / 3, s = tap - r * 3;
          u32x4_t vf[TCc];
#pragma unroll
          for (int b = 0; b < TCc; ++b) {
            const char* b0 = Vl + (STR == 1 ? (2 * ks + r) * 18 + s : (4 * ks + r) * HRS + (s & 1) * 17 + (s >> 1)) * VSB + (b * 16) * 2;
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0 + STR * HRS * VSB));
            const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
            vf[b] = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
          }
#pragma unroll
          for (int a = 0; a < TK; ++a)
#pragma unroll
            for (int b = 0; b < TCc; ++b) acc[t][a][b] = Mma<T>::run(zf[a], vf[b], acc[t][a][b]);
        }
      };
      if constexpr (HSEL == 0) taps16(std::integral_constant<int, 0>{}, std::integral_constant<int, 9>{});
      else if constexpr (HSEL == 1) taps16(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
      else taps16(std::integral_constant<int, 5>{}, std::integral_constant<int, 4>{});
    } else {
      const int i = lane & 15, kg = lane >> 4;
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int px = 32 * ks + 4 * st + kg;          // tile pixel of this lane for this 4-pixel step
        const int row = px >> 4, x = px & 15;
        float zf[TK];
#pragma unroll
        for (int a = 0; a < TK; ++a) zf[a] = *reinterpret_cast<const float*>(Zs + px * ZSB + (wk0 + a * 16 + i) * 4);
        auto taps32 = [&](auto tap0_c, auto ntap_c) {
          constexpr int TAP0 = decltype(tap0_c)::value, NTP = decltype(ntap_c)::value;
#pragma unroll
          for (int t = 0; t < NTP; ++t) {
            const int tap = TAP0 + t;
            const int r = tap / 3, s = tap - r * 3;
            float vf[TCc];
#pragma unroll
            for (int b = 0; b < TCc; ++b)
              vf[b] = *reinterpret_cast<const float*>(Vs + ((row + r) * 18 + x + s) * VSB + (wc0 + b * 16 + i) * 4);
#pragma unroll
            for (int a = 0; a < TK; ++a)
#pragma unroll
              for (int b = 0; b < TCc; ++b)
                acc[t][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(zf[a], vf[b], acc[t][a][b], 0, 0, 0);
          }
        };
        if constexpr (HSEL == 0) taps32(std::integral_constant<int, 0>{}, std::integral_constant<int, 9>{});
        else if constexpr (HSEL == 1) taps32(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
        else taps32(std::integral_constant<int, 5>{}, std::integral_constant<int, 4>{});
      }
    }
  };

  // ---- the same four 32-pixel steps as ONE software pipeline over (step, tap) units (16-bit, 2x2 wave layout, stride 1):
  // the transposed fragment reads of unit u + 2 are issued in front of the MFMAs of unit u, so every read has two units (8 MFMAs,
  // >= 128 matrix cycles) to return — the per-step form leaves the compiler to place them, and it waits lgkmcnt(1..3) one or two
  // MFMAs after issuing them (ISA of r02: the LDS latency shows at every tap).  Three V fragment sets, two dz fragment sets;
  // same accumulation order per accumulator (steps 0..3), so the results are bit-identical.
  auto compute_tile_pipe = [&](const char* Zs, const char* Vs, auto hsel_c) {
    constexpr int HSEL = decltype(hsel_c)::value;
    constexpr int TAP0 = HSEL == 2 ? 5 : 0;
    constexpr int NTP = HSEL == 0 ? 9 : (HSEL == 1 ? 5 : 4);
    constexpr int U = 4 * NTP;
    const char* const Zl = Zs + lane_z;
    const char* const Vl = Vs + lane_v;
    u32x4_t zf[2][TK], vf[3][TCc];
    auto tr = [](const char* q) {
      const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(q));
      return __builtin_bit_cast(u32x2_t, v);
    };
    auto load_unit = [&](auto u_c) {
      constexpr int u = decltype(u_c)::value;
      constexpr int ks = u / NTP, tap = TAP0 + u % NTP, r = tap / 3, sx = tap % 3;
      if constexpr (u % NTP == 0) {
#pragma unroll
        for (int a = 0; a < TK; ++a) {
          const char* b0 = Zl + (32 * ks) * ZSB + (a * 16) * 2;
          const u32x2_t lo = tr(b0), hi = tr(b0 + 16 * ZSB);
          zf[ks & 1][a] = u32x4_t{lo[0], lo[1], hi[0], hi[1]};
        }
      }
#pragma unroll
      for (int b = 0; b < TCc; ++b) {
        const char* b0 = Vl + ((2 * ks + r) * 18 + sx) * VSB + (b * 16) * 2;
        const u32x2_t lo = tr(b0), hi = tr(b0 + HRS * VSB);
        vf[u % 3][b] = u32x4_t{lo[0], lo[1], hi[0], hi[1]};
      }
    };
    auto mma_unit = [&](auto u_c) {
      constexpr int u = decltype(u_c)::value;
      constexpr int ks = u / NTP, t = u % NTP;
#pragma unroll
      for (int a = 0; a < TK; ++a)
#pragma unroll
        for (int b = 0; b < TCc; ++b) acc[t][a][b] = Mma<T>::run(zf[ks & 1][a], vf[u % 3][b], acc[t][a][b]);
    };
    auto unit = [&](auto u_c, auto&& self) {
      constexpr int u = decltype(u_c)::value;
      if constexpr (u < U) {
        if constexpr (u + 2 < U) {
          load_unit(std::integral_constant<int, u + 2>{});
          __builtin_amdgcn_sched_group_barrier(0x100, ((u + 2) % NTP == 0 ? 2 * TK : 0) + 2 * TCc, 0);
        }
        mma_unit(u_c);
        __builtin_amdgcn_sched_group_barrier(0x008, TK * TCc, 0);
        self(std::integral_constant<int, u + 1>{}, self);
      }
    };
    load_unit(std::integral_constant<int, 0>{});
    load_unit(std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * TK + 4 * TCc, 0);
    unit(std::integral_constant<int, 0>{}, unit);
  };

  // ---- stream over this workgroup's pixel tiles: t = blockIdx.z, + splits, ...
  const int dbg = p.dbg_skip_epilogue >> 1;     // timing experiments (VK_WH_DBG, results WRONG): 1 no tile loads, 2 no LDS tile stores, 4 no MFMA steps, 8 no barriers
  int t = t0;
  if (t < t_end) {
    load_tile(t, zreg[0], vreg[0], vmask[0]);
    if constexpr (ZDMA) dma_z(t, 0);
    store_tile(0, zreg[0], vreg[0], vmask[0]);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0), known to the compiler (see the tile loop)
  if (DEPTH == 2 && t + t_step < t_end) load_tile(t + t_step, zreg[0], vreg[0], vmask[0]);     // tile 1 -> set 0
  __syncthreads();
  auto tile_loop = [&](auto hsel_c) {
    auto iteration = [&](int it, u32x4_t (&zl)[ZPASS], u32x4_t (&vl)[VPASS], uint32_t& ml, const u32x4_t (&zs)[ZPASS],
                         const u32x4_t (&vs)[VPASS], const uint32_t& ms) {
      // loads go into set (zl, vl); the set (zs, vs) — requested one (DEPTH 2) or zero (DEPTH 1: same set) iterations ago — is stored
      const bool more = t + t_step < t_end;
      const int tl = t + DEPTH * t_step;
      if (tl < t_end && !(dbg & 1)) {
        load_tile(tl, zl, vl, ml);
        // ZDMA (DEPTH 1): dz of tile t + 1 goes straight into the other stage buffer, which every wave has finished reading (the
        // barrier that closed the previous iteration); lds_dma_wait_all() in front of the barrier closing this one waits for it
        if constexpr (ZDMA) dma_z(tl, (it + 1) & 1);
      }
      const char* Zs = smem + (it & 1) * STAGE;
      const char* Vs = Zs + PX * ZSB;
      bool stored = false;
      if (!(dbg & 4)) {
        if (WS) {
          compute_step(Zs, Vs, wave, hsel_c);
        } else {
          // the next tile's LDS writes (other stage buffer) go between the 32-pixel steps instead of after the last one: the
          // ds_write_b128 transfers then run beside the MFMAs of the remaining steps instead of in front of the barrier
          if constexpr (VK_WH_PIPE && EB == 2 && STR == 1) {
            compute_tile_pipe(Zs, Vs, hsel_c);
          } else {
            compute_step(Zs, Vs, 0, hsel_c);
            compute_step(Zs, Vs, 1, hsel_c);
            if (STORE_MID && more && !(dbg & 2)) { store_tile((it + 1) & 1, zs, vs, ms); stored = true; }
            compute_step(Zs, Vs, 2, hsel_c);
            compute_step(Zs, Vs, 3, hsel_c);
          }
        }
      }
      if (!stored && more && !(dbg & 2)) store_tile((it + 1) & 1, zs, vs, ms);
      // every vector-memory operation of this iteration has landed here: the V loads were consumed by store_tile, the hidden dz DMAs
      // of tile t + 1 were issued in front of this tile's MFMAs.  The wait is the BUILTIN (free at run time) so that hipcc knows it:
      // without it the loads of the previous iteration count as possibly pending at the loop head (the store is conditional), and the
      // first reuse of one of their destination registers drew an `s_waitcnt vmcnt(0)` in the middle of the NEXT tile's loads — a full
      // HBM round trip per tile for the waves of one tap group (r03 binary: ISA of the HSEL = 1 loop)
      __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0); expcnt / lgkmcnt untouched
      if (!(dbg & 8)) __syncthreads();
    };
    for (int it = 0; t < t_end; t += t_step, ++it) {
      if (DEPTH == 1) {
        iteration(it, zreg[0], vreg[0], vmask[0], zreg[0], vreg[0], vmask[0]);
      } else {
        // even iterations: tile t+2 -> set 1, tile t+1 (set 0) -> LDS; odd iterations the other way round
        iteration(it, zreg[DEPTH - 1], vreg[DEPTH - 1], vmask[DEPTH - 1], zreg[0], vreg[0], vmask[0]);
        t += t_step;
        ++it;
        if (t >= t_end) break;
        iteration(it, zreg[0], vreg[0], vmask[0], zreg[DEPTH - 1], vreg[DEPTH - 1], vmask[DEPTH - 1]);
      }
    }
  };
  if (!TS) tile_loop(std::integral_constant<int, 0>{});
  else if (half == 0) tile_loop(std::integral_constant<int, 1>{});
  else tile_loop(std::integral_constant<int, 2>{});

  // ---- epilogue
  if (p.dbg_skip_epilogue & 1) {
    if (acc[0][0][0][0] == 123.456f) p.dw[0] = 1.f;     // keep the accumulators alive
    return;
  }
  // stage the fp32 output tile [k][tap][c] in LDS (WS: the four waves add their partial tiles with LDS atomics),
  // then write whole 16-byte vectors: plain stores into this split's slab, or fp32 atomics when no slab was given
  float* red = reinterpret_cast<float*>(smem);
  // WS: the four waves hold partial sums of the SAME output tile (identical lane -> element mapping): they add
  // themselves into the LDS tile one after the other (ds_add_f32 atomics proved ~10x slower than this).
#pragma unroll 1
  for (int phase = 0; phase < (WS ? 4 : 1); ++phase) {
    if (!WS || wave == phase) {
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        const int tp = (TS && half) ? 5 + t : t;
        if (tp >= 9) continue;
#pragma unroll
        for (int a = 0; a < TK; ++a)
#pragma unroll
          for (int b = 0; b < TCc; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int k = wk0 + a * 16 + (lane >> 4) * 4 + e, c = wc0 + b * 16 + (lane & 15);
              float* dst = red + (k * 9 + tp) * CT + c;
              if (WS && phase > 0) *dst += acc[t][a][b][e];
              else *dst = acc[t][a][b][e];
            }
      }
    }
    __syncthreads();
  }
  constexpr int C4 = CT / 4;
  for (int i = tid; i < KT * 9 * C4; i += NT) {
    const int row = i / C4, c4 = i - row * C4;
    const int k = row / 9, tp = row - k * 9;
    if (k0 + k >= p.K) continue;
    const f32x4_t v = *reinterpret_cast<const f32x4_t*>(red + row * CT + c4 * 4);
    if (dst) {
      *reinterpret_cast<f32x4_t*>(dst + (size_t)row * ldc + c4 * 4) = v;
    } else {
      const size_t off = ((size_t)(k0 + k) * 9 + tp) * p.C + c0 + c4 * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(p.dw + off + e, v[e]);
    }
  }
}

template <typename T, int KT, int CT, bool WS, bool TS, int STR = 1>
__global__ __launch_bounds__(TS ? 512 : 256) void wgrad_halo_kernel(const WhParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k0 = blockIdx.x * KT, c0 = blockIdx.y * CT;
  // slab mode: this split's slab in the layout of dw ([K][9][C]); the tile starts at row k0 * 9, column c0
  float* const dst = p.slab ? p.slab + (size_t)blockIdx.z * p.K * 9 * p.C + (size_t)k0 * 9 * p.C + c0 : nullptr;
  wh_segment<T, KT, CT, WS, TS, STR>(p, smem, k0, c0, (int)blockIdx.z, p.ntiles, p.splits, dst, p.C);
}

// ---- one launch for SEVERAL layers of the 64 x 64 tap-split class (r03): the units (layer, output tile, 128-pixel tile) of all of them in
// one line, cut into equal ranges, one per workgroup; a range that crosses an output-tile boundary becomes two (three) segments, each
// leaving a partial tile [64][9][64]; k_wgrad_tile_reduce adds the partial tiles of every output tile in range order (reproducible).
// Why: a single layer's launch is 16 tiles of work per workgroup against ~11 us of fixed cost, a 7 us epilogue, 37.7 MB of slabs and a
// 9.5 us reduce launch (profiles/r03/wgrad_layout_experiments.log); batching the 2-7 layers of a backward stage pays those once.
struct WhSeg {
  int layer, k0, c0, t0, t1, slot;
};

template <typename T>
__global__ __launch_bounds__(512) void wgrad_halo_batch_kernel(const WhParams* __restrict__ layers, const WhSeg* __restrict__ segs,
                                                               const int* __restrict__ wg_first, float* slab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // workgroups go round-robin over the 8 XCDs (each with its own L2): XCD x takes the x-th eighth of the unit list, so that the
  // workgroups walking neighbouring output tiles of a layer — which read the same dz tiles (same K tile) or the same operand tiles
  // (same C tile) at about the same time — share one L2 (VK_WH_XCD=0: workgroup b takes range b, r03)
  int b = (int)blockIdx.x;
  if (VK_WH_XCD && (gridDim.x & 7) == 0) b = (b & 7) * (int)(gridDim.x >> 3) + (b >> 3);
  const int s0 = wg_first[b], s1 = wg_first[b + 1];
  for (int s_ = s0; s_ < s1; ++s_) {
    const WhSeg sg = segs[s_];
    const WhParams p = layers[sg.layer];
    wh_segment<T, 64, 64, false, true, 1>(p, smem, sg.k0, sg.c0, sg.t0, sg.t1, 1, slab + (size_t)sg.slot * (64 * 9 * 64), 64);
    __syncthreads();                                // the epilogue's LDS tile is read out before the next segment stages into it
  }
}

// dw tile (k0, c0) of one layer += its partial tiles first .. first + n - 1, in that order
struct WhTileRed {
  float* dw;            // dw + k0 * 9 * C + c0
  int ldc, first, n, krows;
};
__global__ __launch_bounds__(256) void k_wgrad_tile_reduce(const WhTileRed* __restrict__ tab, const float* __restrict__ slab) {
  const WhTileRed tr = tab[blockIdx.x];
  // 64 x 9 rows of 16 float4: blockIdx.y takes 64 rows; thread = (row, float4)
  const int row = (int)blockIdx.y * 16 + (threadIdx.x >> 4), c4 = threadIdx.x & 15;
#pragma unroll 1
  for (int rr = row; rr < 64 * 9; rr += 16 * (int)gridDim.y) {
    if (rr / 9 >= tr.krows) continue;
    const float* src = slab + (size_t)tr.first * (64 * 9 * 64) + (size_t)rr * 64 + c4 * 4;
    f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; i + 4 <= tr.n; i += 4) {                  // four loads in flight, added in order
      const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(src + (size_t)i * (64 * 9 * 64));
      const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(src + (size_t)(i + 1) * (64 * 9 * 64));
      const f32x4_t v2 = *reinterpret_cast<const f32x4_t*>(src + (size_t)(i + 2) * (64 * 9 * 64));
      const f32x4_t v3 = *reinterpret_cast<const f32x4_t*>(src + (size_t)(i + 3) * (64 * 9 * 64));
      a = a + v0; a = a + v1; a = a + v2; a = a + v3;
    }
    for (; i < tr.n; ++i) a = a + *reinterpret_cast<const f32x4_t*>(src + (size_t)i * (64 * 9 * 64));
    float* d = tr.dw + (size_t)rr * tr.ldc + c4 * 4;
    *reinterpret_cast<f32x4_t*>(d) = *reinterpret_cast<const f32x4_t*>(d) + a;
  }
}

// dw[e] += sum_s slab[s][e]  in a fixed order (reproducible gradients).  16 float4 elements x 16 split lanes per
// workgroup: lane ty adds splits ty, ty+16, ... (4 loads in flight), then the 16 partial sums are added in order.
__global__ __launch_bounds__(256) void k_wgrad_slab_reduce(size_t n4, int splits, const float* __restrict__ slab, float* __restrict__ dw) {
  __shared__ f32x4_t red[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const size_t e = (size_t)blockIdx.x * 16 + tx;
  f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (e < n4) {
    int s = ty;
    for (; s + 48 < splits; s += 64) {
      const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(slab + ((size_t)s * n4 + e) * 4);
      const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(slab + ((size_t)(s + 16) * n4 + e) * 4);
      const f32x4_t v2 = *reinterpret_cast<const f32x4_t*>(slab + ((size_t)(s + 32) * n4 + e) * 4);
      const f32x4_t v3 = *reinterpret_cast<const f32x4_t*>(slab + ((size_t)(s + 48) * n4 + e) * 4);
      a = a + v0; a = a + v1; a = a + v2; a = a + v3;
    }
    for (; s < splits; s += 16) a = a + *reinterpret_cast<const f32x4_t*>(slab + ((size_t)s * n4 + e) * 4);
  }
  red[ty][tx] = a;
  __syncthreads();
  if (ty == 0 && e < n4) {
    f32x4_t t = *reinterpret_cast<const f32x4_t*>(dw + 4 * e);
#pragma unroll
    for (int j = 0; j < 16; ++j) t = t + red[j][tx];
    *reinterpret_cast<f32x4_t*>(dw + 4 * e) = t;
  }
}

// ---- streaming form for the 32-channel inputs of decoder blocks 3 / 4 (r03): C = 32 (one source, optionally nearest-x2 upsampled),
// K = 16 or 32, 16-bit types.  These launches move 0.27-0.34 GB for 10-20 GFLOP (HBM-bound, floors 54 / 67 us) and the tile kernel
// runs them at 121 / 188 us: every 128-pixel tile is a load -> transform -> LDS -> barrier -> 18-36 MFMAs chain.  Here, as in
// conv3x3_stream_kernel, every WAVE is its own pipeline and the only workgroup barriers are those of the final reduction:
//   * a wave owns a strip of 16 columns x RS rows of one image and walks down it two rows (= one K = 32 MFMA step) at a time, the whole
//     K x 9 x 32 output in its accumulators (18 / 36 tiles);
//   * per step it needs two new dz rows and two new V rows (one SOURCE row when upsampled: each source pixel goes to its two halo
//     columns, each source row serves two steps' worth of taps), BN+ReLU-transformed once and written once to the wave's private LDS
//     ring; the rows of the next steps are in flight in registers (queue depth = ring period, indices compile-time);
//   * fragments are the LDS-transposed reads of wgrad_halo_kernel (pixels along the MFMA reduction), the dz fragments shared by the
//     nine taps: (2 TK + 36) reads per 18 TK MFMAs;
//   * at the end the four waves of a workgroup add their tiles in LDS in a fixed order and the workgroup writes ONE slab
//     (k_wgrad_slab_reduce adds the workgroups in order: reproducible).
template <typename T, int K, bool UP>
struct WsCfg {
  static constexpr int C = 32, TK = K / 16, TCc = 2;
  static constexpr int VSB = 96, ZSB = K == 16 ? 32 : 96;            // pixel strides of the transposed reads (WhCfg::zpad)
  static constexpr int VROW = 18 * VSB, ZROW = 16 * ZSB;
  static constexpr int PER = UP ? 4 : 3;                              // ring period in steps = look-ahead depth
  static constexpr int NVS = UP ? 4 : 6, NZS = UP ? 8 : 6;            // ring slots: V (source) rows, dz rows
  static constexpr int WAVE_LDS = NVS * VROW + NZS * ZROW;
  static constexpr int RED = K * 9 * C * 4;
  static constexpr int SMEM = 4 * WAVE_LDS > RED ? 4 * WAVE_LDS : RED;
  static constexpr int NVL = UP ? 1 : 3;                              // V vectors per lane and step (40 / 144 vectors)
  static constexpr int NZL = K / 16;                                  // dz vectors per lane and step (64 / 128 vectors)
};

// Work unit = (strip, 32-channel chunk of the source, K-wide slice of the output channels): one wave each; the four waves of a workgroup
// are four strips of the SAME (chunk, slice) = blockIdx.y, so their tiles add up.  p.K / p.C: channel counts of dz / of the whole
// (concatenated) input, p.s0: THIS launch's source (its own channel stride), c_dst0: where that source starts in the concatenation.
struct WsArgs {
  int RS, c_dst0, nks;
};

#ifndef VK_WS_MINWG
#define VK_WS_MINWG 2
#endif
template <typename T, int K, bool UP>
__global__ __launch_bounds__(256, (K == 16 ? (UP ? VK_WS_MINWG : 2) : 1)) void wgrad_stream_kernel(const WhParams p, const WsArgs wa) {
  using Cfg = WsCfg<T, K, UP>;
  const int RS = wa.RS;
  const int chunk = (int)blockIdx.y / wa.nks, kslice = (int)blockIdx.y - chunk * wa.nks;
  const int cl0 = chunk * 32, kbase = kslice * K, Cs = p.s0.C;
  constexpr int TK = Cfg::TK, TCc = Cfg::TCc, VSB = Cfg::VSB, ZSB = Cfg::ZSB, VROW = Cfg::VROW, ZROW = Cfg::ZROW;
  constexpr int PER = Cfg::PER, NVS = Cfg::NVS, NZS = Cfg::NZS, NVL = Cfg::NVL, NZL = Cfg::NZL, VE = 8, C = 32;
  static_assert(sizeof(T) == 2, "16-bit element types");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* const Vr = smem + wave * Cfg::WAVE_LDS;
  char* const Zr = Vr + NVS * VROW;
  const int strips_x = (p.W + 15) / 16, strips_y = (p.H + RS - 1) / RS;
  int sid = (int)blockIdx.x * 4 + wave;
  const bool active = sid < p.N * strips_y * strips_x;
  const int sx = sid % strips_x;
  sid /= strips_x;
  const int sy = sid % strips_y;
  const int n = sid / strips_y;
  const int x0 = sx * 16, ys = sy * RS, ye = min(p.H, ys + RS);
  const __amdgpu_buffer_rsrc_t rsv = make_rsrc(p.s0.ptr, p.s0.bytes);
  const __amdgpu_buffer_rsrc_t rsz = make_rsrc(p.dz, p.dz_bytes);
  const bool affine = p.s0.scale != nullptr, relu = p.s0.relu != 0;
  const int Hs = p.H >> (UP ? 1 : 0), Ws = p.W >> (UP ? 1 : 0);

  f32x4_t acc[9][TK][TCc];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int a = 0; a < TK; ++a)
#pragma unroll
      for (int b = 0; b < TCc; ++b) acc[t][a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if (active) {
    // ---- staging geometry (fixed for the strip)
    // V: vector v = lane + 64 q -> (row of the pair, halo / source pixel, 16-byte piece); every lane's vectors cover the same 8 channels
    float sc[VE], sh[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) {
      sc[j] = affine ? p.s0.scale[cl0 + (lane & 3) * VE + j] : 1.f;
      sh[j] = affine ? p.s0.shift[cl0 + (lane & 3) * VE + j] : 0.f;
    }
    int v_rowsel[NVL], v_col[NVL], v_goff[NVL];
    bool v_ok[NVL];
#pragma unroll
    for (int q = 0; q < NVL; ++q) {
      const int v = lane + 64 * q;
      if (UP) {
        const int j = v >> 2;                                // source pixel 0..9 <-> source column x0/2 - 1 + j
        const int cs = (x0 >> 1) - 1 + j;
        v_rowsel[q] = 0;
        v_col[q] = j;
        v_ok[q] = v < 40 && (unsigned)cs < (unsigned)Ws;
        v_goff[q] = cs * Cs + cl0 + (v & 3) * VE;
      } else {
        const int rs_ = v / 72, hp = (v - rs_ * 72) >> 2;    // 72 vectors per row
        const int xs = x0 - 1 + hp;
        v_rowsel[q] = rs_;
        v_col[q] = hp;
        v_ok[q] = v < 144 && (unsigned)xs < (unsigned)p.W;
        v_goff[q] = xs * Cs + cl0 + (v & 3) * VE;
      }
    }
    int z_rowsel[NZL], z_px[NZL], z_pc[NZL];
    bool z_ok[NZL];
#pragma unroll
    for (int q = 0; q < NZL; ++q) {
      const int v = lane + 64 * q;
      constexpr int VPR = 16 * (K / 8);                       // vectors per dz row
      z_rowsel[q] = v / VPR;
      z_px[q] = (v % VPR) / (K / 8);
      z_pc[q] = v % (K / 8);
      z_ok[q] = x0 + z_px[q] < p.W;
    }
    // requests: V row(s) and dz rows that step s adds (s >= 0; the prologue issues the rows of step 0 separately)
    auto issue_v = [&](int row, int rowsel_only, u32x4_t (&r)[NVL]) {     // non-UP: `row` = first of the two rows; UP: the source row
#pragma unroll
      for (int q = 0; q < NVL; ++q) {
        const int rr = row + v_rowsel[q];
        const bool ok = v_ok[q] && (unsigned)rr < (unsigned)Hs && (rowsel_only < 0 || v_rowsel[q] == rowsel_only);
        r[q] = buf_load16(rsv, ok ? (uint32_t)(((n * Hs + rr) * Ws) * Cs + v_goff[q]) * 2u : kOOB);
      }
    };
    auto issue_z = [&](int row, u32x4_t (&r)[NZL]) {                      // dz rows row, row + 1 (zeros beyond the strip's end)
#pragma unroll
      for (int q = 0; q < NZL; ++q) {
        const int rr = row + z_rowsel[q];
        const bool ok = z_ok[q] && rr < ye;
        r[q] = buf_load16(rsz, ok ? (uint32_t)(((n * p.H + rr) * p.W + x0 + z_px[q]) * p.K + kbase + z_pc[q] * VE) * 2u : kOOB);
      }
    };
    // ring writes.  V: transform, zero outside the map (the buffer load returned zeros there, the affine must not turn them into shift)
    auto write_v = [&](int row, int slot0, const u32x4_t (&r)[NVL]) {     // non-UP: rows row, row + 1 -> slots slot0, slot0 + 1 (mod NVS)
#pragma unroll
      for (int q = 0; q < NVL; ++q) {
        const int v = lane + 64 * q;
        if (v >= (UP ? 40 : 144)) continue;
        const int rr = row + v_rowsel[q];
        u32x4_t x = r[q];
        if (affine) x = AffineRelu<T>::run(x, sc, sh, relu);
        if (!(v_ok[q] && (unsigned)rr < (unsigned)Hs)) x = u32x4_t{0, 0, 0, 0};
        char* const dst = Vr + ((slot0 + v_rowsel[q]) % NVS) * VROW + (v & 3) * 16;
        if (UP) {
          const int j = v_col[q];                            // source pixel j -> halo columns 2j - 1, 2j
          if (j > 0) *reinterpret_cast<u32x4_t*>(dst + (2 * j - 1) * VSB) = x;
          if (j < 9) *reinterpret_cast<u32x4_t*>(dst + (2 * j) * VSB) = x;
        } else {
          *reinterpret_cast<u32x4_t*>(dst + v_col[q] * VSB) = x;
        }
      }
    };
    auto write_z = [&](int slot0, const u32x4_t (&r)[NZL]) {
#pragma unroll
      for (int q = 0; q < NZL; ++q)
        *reinterpret_cast<u32x4_t*>(Zr + ((slot0 + z_rowsel[q]) % NZS) * ZROW + z_px[q] * ZSB + z_pc[q] * 16) = r[q];
    };
    // transposed-read lane geometry (wgrad_halo_kernel): lane j of 16-lane group g supplies pixel 4 g + (j >> 2), channels 4 (j & 3) ..
    const int lane_z = (4 * (lane >> 4) + ((lane & 15) >> 2)) * ZSB + (4 * (lane & 3)) * 2;
    const int lane_v = (4 * (lane >> 4) + ((lane & 15) >> 2)) * VSB + (4 * (lane & 3)) * 2;
    auto tr = [](const char* q) { return __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(q))); };

    // ---- prologue.  Ring slots: non-UP  V row ys - 1 + i -> slot i % 6, dz row ys + j -> slot (j + 1) % 6;
    //                            UP      source row ys/2 - 1 + i -> slot i % 4, dz row ys + j -> slot j % 8
    u32x4_t vq[PER][NVL], zq[PER][NZL];
    const int ss = ys >> 1;
    if (UP) {
      u32x4_t v0[NVL], v1[NVL], v2[NVL], z0[NZL];
      issue_v(ss - 1, -1, v0);
      issue_v(ss, -1, v1);
      issue_v(ss + 1, -1, v2);
      issue_z(ys, z0);
#pragma unroll
      for (int s_ = 1; s_ < PER; ++s_) { issue_v(ss + 1 + s_, -1, vq[s_]); issue_z(ys + 2 * s_, zq[s_]); }     // step s_ adds source row ss + s_ + 1
      write_v(ss - 1, 0, v0);
      write_v(ss, 1, v1);
      write_v(ss + 1, 2, v2);
      write_z(0, z0);
    } else {
      u32x4_t v0[NVL], v1[NVL], z0[NZL];
      issue_v(ys - 1, -1, v0);                               // rows ys - 1, ys
      issue_v(ys + 1, -1, v1);                               // rows ys + 1, ys + 2
      issue_z(ys, z0);
#pragma unroll
      for (int s_ = 1; s_ < PER; ++s_) { issue_v(ys + 2 * s_ + 1, -1, vq[s_]); issue_z(ys + 2 * s_, zq[s_]); }   // step s_ adds V rows ys + 2 s_ + 1, + 2
      write_v(ys - 1, 0, v0);
      write_v(ys + 1, 2, v1);
      write_z(1, z0);
    }

    auto step = [&](int t, auto ph_c) {
      constexpr int PH = decltype(ph_c)::value;              // t mod PER
      const int y = ys + 2 * t;
      // rows of step t + PER into the queue slot whose content went into the ring one step ago
      if (UP) issue_v(ss + 1 + t + PER, -1, vq[PH]);
      else issue_v(y + 2 * PER + 1, -1, vq[PH]);
      issue_z(y + 2 * PER, zq[PH]);
      // ---- fragments and MFMAs
      const char* const Z0 = Zr + (UP ? (2 * PH) % NZS : (2 * PH + 1) % NZS) * ZROW + lane_z;
      const char* const Z1 = Zr + (UP ? (2 * PH + 1) % NZS : (2 * PH + 2) % NZS) * ZROW + lane_z;
      u32x4_t zf[TK];
#pragma unroll
      for (int a = 0; a < TK; ++a) {
        const u32x2_t lo = tr(Z0 + a * 32), hi = tr(Z1 + a * 32);
        zf[a] = u32x4_t{lo[0], lo[1], hi[0], hi[1]};
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        // output rows y, y + 1 see V rows y - 1 + r, y + r
        const int slo = UP ? (PH + (r + 1) / 2) % NVS : (2 * PH + r) % NVS;          // UP: source rows (y - 1 + r) >> 1, (y + r) >> 1, y even
        const int shi = UP ? (PH + (r + 2) / 2) % NVS : (2 * PH + r + 1) % NVS;
#pragma unroll
        for (int sx_ = 0; sx_ < 3; ++sx_) {
          u32x4_t vf[TCc];
#pragma unroll
          for (int b = 0; b < TCc; ++b) {
            const u32x2_t lo = tr(Vr + slo * VROW + sx_ * VSB + lane_v + b * 32), hi = tr(Vr + shi * VROW + sx_ * VSB + lane_v + b * 32);
            vf[b] = u32x4_t{lo[0], lo[1], hi[0], hi[1]};
          }
#pragma unroll
          for (int a = 0; a < TK; ++a)
#pragma unroll
            for (int b = 0; b < TCc; ++b) acc[r * 3 + sx_][a][b] = Mma<T>::run(zf[a], vf[b], acc[r * 3 + sx_][a][b]);
        }
      }
      // ---- the rows step t + 1 adds: into slots this step does not read
      if (UP) {
        write_v(ss + 2 + t, (PH + 3) % NVS, vq[(PH + 1) % PER]);
        write_z((2 * PH + 2) % NZS, zq[(PH + 1) % PER]);
      } else {
        write_v(y + 3, (2 * PH + 4) % NVS, vq[(PH + 1) % PER]);
        write_z((2 * PH + 3) % NZS, zq[(PH + 1) % PER]);
      }
    };
    const int nsteps = (ye - ys + 1) >> 1;
    for (int t = 0; t < nsteps; t += PER) {
      step(t, std::integral_constant<int, 0>{});
      if (t + 1 < nsteps) step(t + 1, std::integral_constant<int, 1>{});
      if (t + 2 < nsteps) step(t + 2, std::integral_constant<int, 2>{});
      if constexpr (PER == 4) {
        if (t + 3 < nsteps) step(t + 3, std::integral_constant<int, 3>{});
      }
    }
  }

  // ---- the four waves add their tiles in wave order; one slab per workgroup
  __syncthreads();                                             // every wave is done with its ring
  float* const red = reinterpret_cast<float*>(smem);
#pragma unroll 1
  for (int phase = 0; phase < 4; ++phase) {
    if (wave == phase) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < TK; ++a)
#pragma unroll
          for (int b = 0; b < TCc; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int k = a * 16 + (lane >> 4) * 4 + e, c = b * 16 + (lane & 15);
              float* dst = red + (k * 9 + t) * C + c;
              if (phase > 0) *dst += acc[t][a][b][e];
              else *dst = acc[t][a][b][e];
            }
    }
    __syncthreads();
  }
  float* const slab = p.slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * K * 9 * C;
  for (int i = tid; i < K * 9 * C / 4; i += 256)
    *reinterpret_cast<f32x4_t*>(slab + i * 4) = *reinterpret_cast<const f32x4_t*>(red + i * 4);
}

// ------------------------------------------------------------------------------------------------ host
// tuning knobs, read per call so that tests can force the kernel onto small problems
static int wh_max_combo() { const char* e = getenv("VK_WH_MAXCOMBO"); return e ? atoi(e) : 64; }
static int wh_min_blocks() { const char* e = getenv("VK_WH_MINBLOCKS"); return e ? atoi(e) : 160; }

template <typename T, int KT, int CT, bool WS, bool TS = false, int STR = 1>
static int launch_wh(WhParams p, size_t slab_bytes, hipStream_t st) {
  using Cfg = WhCfg<T, KT, CT, WS, TS, STR>;
  p.tiles_x = (p.W + 15) / 16;
  p.tiles_y = (p.H + 7) / 8;
  p.ntiles = p.N * p.tiles_y * p.tiles_x;
  const int kt = (p.K + KT - 1) / KT, ct = p.C / CT;
  if (kt * ct > wh_max_combo()) return VK_ERR_UNSUPPORTED;      // deep layers: output tile traffic would dominate
  // every workgroup ends with KT*CT*9 fp32 atomics: give it at least ~6 pixel tiles of work
  const char* e_blk = getenv("VK_WH_BLOCKS");
  // non-WS: LDS allows exactly one workgroup per CU; with compute units reserved for a concurrent collective (vk_set_reserved_cus)
  // the persistent grid is sized for the rest, instead of stranding its last workgroups behind a full round
  const int target_blocks = e_blk ? atoi(e_blk) : (WS ? (KT >= 32 ? 512 : 1024) : 256 - vkh::reserved_cus());
  int splits = target_blocks / (kt * ct);     // never exceed the target: a second round of workgroups costs a full round
  if (splits > p.ntiles / 6) splits = p.ntiles / 6;
  if (splits < 1) splits = 1;
  if ((long)splits * kt * ct < wh_min_blocks()) return VK_ERR_UNSUPPORTED;  // too few workgroups to fill the chip
  p.splits = splits;
  p.dbg_skip_epilogue = (getenv("VK_WH_DBG_NOEPI") ? 1 : 0) | ((getenv("VK_WH_DBG") ? atoi(getenv("VK_WH_DBG")) : 0) << 1);
  const size_t slab_need = (size_t)splits * p.K * 9 * p.C * sizeof(float);
  if (!p.slab || slab_need > slab_bytes || getenv("VK_WH_NO_SLAB")) p.slab = nullptr;
  dim3 grid(kt, ct, splits);
  static bool attr_done = false;
  if (!attr_done && Cfg::SMEM > 64 * 1024) {
    VK_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_halo_kernel<T, KT, CT, WS, TS, STR>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
    attr_done = true;
  }
  {
    static const std::string tag = std::string("wgrad_halo_") + (sizeof(T) == 4 ? "f32" : "16b") + "_" + std::to_string(KT) + "x" + std::to_string(CT) + (TS ? "ts" : "") +
                                   (STR == 2 ? "_s2" : "");
    const double bytes = ((double)p.N * p.Hi * p.Wi * p.C + (double)p.N * p.H * p.W * p.K) * sizeof(T) + 9.0 * p.K * p.C * 4.0;
    const std::string dtag = getenv("VK_PROF_DETAIL") ? tag + ":H" + std::to_string(p.H) + "_K" + std::to_string(p.K) + "_C" + std::to_string(p.C) + "_s" + std::to_string(splits) : tag;
    vkh::ProfScope ps(dtag.c_str(), st, 2.0 * (double)p.N * p.H * p.W * p.K * 9.0 * p.C, bytes);
    hipLaunchKernelGGL((wgrad_halo_kernel<T, KT, CT, WS, TS, STR>), grid, dim3(Cfg::NT), Cfg::SMEM, st, p);
  }
  if (p.slab) {
    const size_t n4 = (size_t)p.K * 9 * p.C / 4;
    vkh::ProfScope ps("wgrad_slab_reduce", st, 0.0, (double)(splits + 2) * n4 * 16.0);
    hipLaunchKernelGGL(k_wgrad_slab_reduce, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, n4, splits, p.slab, p.dw);
  }
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

// shared with conv_wgrad.hip (tap-by-tap and stem kernels in reproducible mode)
void launch_slab_reduce(size_t n4, int splits, const float* slab, float* dw, hipStream_t st) {
  vkh::ProfScope ps("wgrad_slab_reduce", st, 0.0, (double)(splits + 2) * n4 * 16.0);
  hipLaunchKernelGGL(k_wgrad_slab_reduce, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, n4, splits, slab, dw);
}

// dw[(kbase + k)][tap][c_dst + c] += sum over the nwg workgroup tiles [KW][9][32] of one (chunk, slice), in workgroup order (same
// lane split as k_wgrad_slab_reduce: 16 lanes per float4, each adding every 16th tile, then the 16 partial sums in lane order)
__global__ __launch_bounds__(256) void k_ws_slab_reduce(int KW, int Ctot, int c_dst0, int nks, int nwg, const float* __restrict__ slab, float* __restrict__ dw) {
  __shared__ f32x4_t red[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int n4 = KW * 9 * 8;
  const int e = (int)blockIdx.x * 16 + tx;
  const int chunk = (int)blockIdx.y / nks, kslice = (int)blockIdx.y - chunk * nks;
  const float* const base = slab + (size_t)blockIdx.y * nwg * n4 * 4;
  f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (e < n4) {
    int s_ = ty;
    for (; s_ + 48 < nwg; s_ += 64) {                      // four loads in flight per lane (the tiles come from HBM)
      const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(base + ((size_t)s_ * n4 + e) * 4);
      const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(base + ((size_t)(s_ + 16) * n4 + e) * 4);
      const f32x4_t v2 = *reinterpret_cast<const f32x4_t*>(base + ((size_t)(s_ + 32) * n4 + e) * 4);
      const f32x4_t v3 = *reinterpret_cast<const f32x4_t*>(base + ((size_t)(s_ + 48) * n4 + e) * 4);
      a = a + v0; a = a + v1; a = a + v2; a = a + v3;
    }
    for (; s_ < nwg; s_ += 16) a = a + *reinterpret_cast<const f32x4_t*>(base + ((size_t)s_ * n4 + e) * 4);
  }
  red[ty][tx] = a;
  __syncthreads();
  if (ty == 0 && e < n4) {
    const int row = e >> 3, c4 = e & 7;                   // row = k * 9 + tap inside the slice
    float* const dst = dw + ((size_t)(kslice * KW * 9 + row)) * Ctot + c_dst0 + chunk * 32 + c4 * 4;
    f32x4_t t = *reinterpret_cast<const f32x4_t*>(dst);
#pragma unroll
    for (int j = 0; j < 16; ++j) t = t + red[j][tx];
    *reinterpret_cast<f32x4_t*>(dst) = t;
  }
}

// the 32-channel-chunk layers of decoder blocks 3 / 4 on the streaming kernel: one launch per source (needs the slab workspace: it has
// no atomics epilogue).  Returns the slab bytes it used through *used.
template <typename T, int KW, bool UP>
static int launch_ws(WhParams p, int c_dst0, char* slab, size_t slab_bytes, size_t* used, hipStream_t st) {
  using Cfg = WsCfg<T, KW, UP>;
  const int nchunk = p.s0.C / 32, nks = p.K / KW, gy = nchunk * nks;
  // strips of 16 columns x RS rows, one per wave: enough waves for ~two rounds of the chip, RS even
  int RS = p.H;
  const long per_image = (long)((p.W + 15) / 16);
  // exactly ONE round of resident waves (measured: D3c2 73 / 93 / 134 us at 1 / 2 / 4 rounds, D4c1 132 / 105 / 118 us at 0.5 / 1 / 2):
  // 36 accumulator tiles run one wave per SIMD (1024 waves), 18 tiles two (2048)
  const long want = KW == 32 ? 1024 : 1024 * (UP ? VK_WS_MINWG : 2);
  const long want_e = getenv("VK_WS_WANT") ? atol(getenv("VK_WS_WANT")) : 0;
  while (RS > 32 && (long)p.N * per_image * ((p.H + RS - 1) / RS) * gy < (want_e ? want_e : want)) RS >>= 1;
  RS = (RS + 1) & ~1;
  const long strips = (long)p.N * per_image * ((p.H + RS - 1) / RS);
  const long nwg = (strips + 3) / 4;
  *used = (size_t)nwg * gy * KW * 9 * 32 * sizeof(float);
  if (*used > slab_bytes) return VK_ERR_UNSUPPORTED;
  static bool attr_done = false;
  if (!attr_done && Cfg::SMEM > 64 * 1024) {
    VK_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_stream_kernel<T, KW, UP>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
    attr_done = true;
  }
  p.slab = (float*)slab;
  WsArgs wa{RS, c_dst0, nks};
  {
    static const std::string tag = std::string("wgrad_stream_16b_c32") + (UP ? "up" : "") + "_k" + std::to_string(KW);
    const double bytes = ((double)p.N * (p.H >> (UP ? 1 : 0)) * (p.W >> (UP ? 1 : 0)) * p.s0.C + (double)p.N * p.H * p.W * p.K) * 2.0 + 9.0 * p.K * p.s0.C * 4.0;
    vkh::ProfScope ps(tag.c_str(), st, 2.0 * (double)p.N * p.H * p.W * p.K * 9.0 * p.s0.C, bytes);
    hipLaunchKernelGGL((wgrad_stream_kernel<T, KW, UP>), dim3((unsigned)nwg, (unsigned)gy), dim3(256), Cfg::SMEM, st, p, wa);
  }
  {
    vkh::ProfScope ps("wgrad_slab_reduce", st, 0.0, (double)*used + 2.0 * p.K * 9.0 * p.s0.C * 4.0);
    hipLaunchKernelGGL(k_ws_slab_reduce, dim3((unsigned)((KW * 9 * 8 + 15) / 16), (unsigned)gy), dim3(256), 0, st, KW, p.C, c_dst0, nks, (int)nwg,
                       (const float*)slab, p.dw);
  }
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

// Routed here: ONE source of exactly 32 channels (decoder block 3 conv2, block 4 conv1).  The kernel's work units also cover several
// 32-channel chunks and two sources (decoder block 3 conv1: 64 upsampled + 64 skip channels) — built and measured: every chunk
// re-reads dz and every K slice re-reads V, 294-466 us against the tile kernel's 230 us (profiles/r03/wgrad_stream_ab.log), so the
// host does not send those layers.  VK_NO_WSTREAM=1: the tile kernels instead (tests / A-B);  VK_WS_KW=16: K = 32 as two 16-wide
// slices (two waves per SIMD; measured slower than one 32-wide slice: 115 vs 100 us).
template <typename T>
static int ws_try(const WhParams& p, size_t slab_bytes, hipStream_t st) {
  if constexpr (sizeof(T) != 2) return VK_ERR_UNSUPPORTED;
  else {
    if (!p.slab || getenv("VK_NO_WSTREAM") || getenv("VK_WH_NO_SLAB")) return VK_ERR_UNSUPPORTED;
    if ((p.K != 16 && p.K != 32) || p.s1.ptr || p.s0.C != 32 || p.C != 32) return VK_ERR_UNSUPPORTED;
    if (p.s0.up && ((p.H | p.W) & 1)) return VK_ERR_UNSUPPORTED;
    const char* e = getenv("VK_WS_KW");
    const int kw = (p.K == 32 && e && atoi(e) == 16) ? 16 : p.K;
    size_t used = 0;
    if (kw == 32) return p.s0.up ? launch_ws<T, 32, true>(p, 0, (char*)p.slab, slab_bytes, &used, st) : launch_ws<T, 32, false>(p, 0, (char*)p.slab, slab_bytes, &used, st);
    return p.s0.up ? launch_ws<T, 16, true>(p, 0, (char*)p.slab, slab_bytes, &used, st) : launch_ws<T, 16, false>(p, 0, (char*)p.slab, slab_bytes, &used, st);
  }
}

template <typename T>
static int wh_select(const WhParams& p, int cgran, size_t slab_bytes, hipStream_t st) {
  {
    const int rc = ws_try<T>(p, slab_bytes, st);
    if (rc != VK_ERR_UNSUPPORTED) return rc;
  }
  // cgran: channel granularity that keeps a c-tile inside one concat source
  if constexpr (sizeof(T) == 2) {      // fp32 double-buffered 64x64 stages would need 197 KB of LDS
    if (p.K >= 64 && cgran % 64 == 0) {
      if (getenv("VK_WH_NO_TS")) return launch_wh<T, 64, 64, false>(p, slab_bytes, st);
      return launch_wh<T, 64, 64, false, true>(p, slab_bytes, st);
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (p.K >= 32 && cgran % 64 == 0) return launch_wh<T, 32, 64, false, true>(p, slab_bytes, st);   // decoder block 3 conv1
  }
  if (p.K >= 32 && cgran % 32 == 0) return launch_wh<T, 32, 32, true>(p, slab_bytes, st);
  if (p.K >= 16 && cgran % 32 == 0) return launch_wh<T, 16, 32, true>(p, slab_bytes, st);
  if (p.K >= 16 && cgran % 16 == 0) return launch_wh<T, 16, 16, true>(p, slab_bytes, st);
  return VK_ERR_UNSUPPORTED;
}

// descriptor -> kernel parameters (shared by the single-layer and the batched launch); *cgran: channel granularity that keeps a
// c-tile inside one concat source
static int make_wh_params(const vk_conv_desc* d, const void* dz, float* dw, WhParams* out, int* cgran_out) {
  const int eb = d->dtype == VK_F32 ? 4 : 2;
  const int C = d->src0.C + (d->src1.ptr ? d->src1.C : 0);
  if (d->K % 16 || d->src0.C % 16 || (d->src1.ptr && (d->src1.C % 16 || d->src1.up))) return VK_ERR_UNSUPPORTED;
  if ((size_t)d->N * d->H * d->W * C * eb >= (1ull << 31) || (size_t)d->N * d->Ho * d->Wo * d->K * eb >= (1ull << 31)) return VK_ERR_UNSUPPORTED;
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
  p.dz_bytes = (uint32_t)((size_t)d->N * d->Ho * d->Wo * d->K * eb);
  p.dw = dw;
  p.N = d->N; p.H = d->Ho; p.W = d->Wo; p.K = d->K; p.C = C;
  p.Hi = d->H; p.Wi = d->W;
  p.tiles_x = p.tiles_y = p.ntiles = p.splits = 0;
  p.dbg_skip_epilogue = 0;
  p.slab = nullptr;
  *out = p;
  *cgran_out = cgran;
  return VK_OK;
}

// returns VK_ERR_UNSUPPORTED when the shape is not covered (caller falls back to the tap-by-tap kernel)
int wgrad_halo_try(const vk_conv_desc* d, const void* dz, float* dw, void* workspace, size_t workspace_bytes, hipStream_t st) {
  // stride 2 (16-bit types): the openers of layers 2-4 — materialised single source, reduction chunks of 32 channels
  const bool s2 = d->stride == 2 && d->dtype != VK_F32 && d->H == 2 * d->Ho && d->W == 2 * d->Wo && !d->src1.ptr && !d->src0.up && d->src0.C % 32 == 0 &&
                  d->K >= 64 && !getenv("VK_NO_S2_TILE");
  if (d->R != 3 || d->S != 3 || d->pad != 1) return VK_ERR_UNSUPPORTED;
  if (!s2 && (d->stride != 1 || d->H != d->Ho || d->W != d->Wo)) return VK_ERR_UNSUPPORTED;
  if (getenv("VK_NO_WGRAD_HALO")) return VK_ERR_UNSUPPORTED;
  WhParams p;
  int cgran = 64;
  {
    const int rc = make_wh_params(d, dz, dw, &p, &cgran);
    if (rc != VK_OK) return rc;
  }
  p.slab = (float*)workspace;
  if (s2) {
    if (d->dtype == VK_BF16) return launch_wh<bf16_t, 64, 32, false, true, 2>(p, workspace_bytes, st);
    return launch_wh<f16_t, 64, 32, false, true, 2>(p, workspace_bytes, st);
  }
  switch (d->dtype) {
    case VK_F32: return wh_select<float>(p, cgran, workspace_bytes, st);
    case VK_BF16: return wh_select<bf16_t>(p, cgran, workspace_bytes, st);
    case VK_F16: return wh_select<f16_t>(p, cgran, workspace_bytes, st);
  }
  return VK_ERR_ARG;
}

// ------------------------------------------------------------------------------------------------ batched launch (see wgrad_halo_batch_kernel)
// would wgrad_halo_try run this layer on the 64 x 64 tap-split tile kernel (16-bit, 3x3 stride 1, K >= 64, every source a multiple of
// 64 channels)?  Those are the layers a batch may hold.
bool wgrad_batch_supports(const vk_conv_desc* d) {
  if (d->dtype == VK_F32 || d->R != 3 || d->S != 3 || d->pad != 1 || d->stride != 1 || d->H != d->Ho || d->W != d->Wo) return false;
  if (getenv("VK_NO_WGRAD_HALO") || getenv("VK_WH_NO_TS") || getenv("VK_WH_NO_SLAB") || getenv("VK_NO_WGRAD_BATCH")) return false;
  WhParams p;
  int cgran;
  if (make_wh_params(d, d->src0.ptr, nullptr, &p, &cgran) != VK_OK) return false;
  if (p.K < 64 || cgran % 64) return false;
  const int kt = (p.K + 63) / 64, ct = p.C / 64;
  return kt * ct <= wh_max_combo();
}


// Builds the tables of one batch on the host, copies them into `tables` (synchronous copy: called once per stage and plan) and
// returns the plan.  target_blocks: workgroups of the launch (one per CU that may be used).
int wgrad_batch_build(const vk_conv_desc* descs, const void* const* dz, float* const* dw, int n, int target_blocks, void* tables,
                      size_t tables_bytes, WgradBatchPlan* plan) {
  VK_CHECK_ARG(n >= 1 && n <= 40 && target_blocks >= 1, "wgrad batch: %d layers (1..40), %d workgroups", n, target_blocks);
  std::vector<WhParams> layers((size_t)n);
  long total_units = 0;
  std::vector<int> kt((size_t)n), ct((size_t)n);
  for (int l = 0; l < n; ++l) {
    int cgran;
    const int rc = make_wh_params(&descs[l], dz[l], dw[l], &layers[(size_t)l], &cgran);
    if (rc != VK_OK) return rc;
    WhParams& q = layers[(size_t)l];
    q.tiles_x = (q.W + 15) / 16;
    q.tiles_y = (q.H + 7) / 8;
    q.ntiles = q.N * q.tiles_y * q.tiles_x;
    q.splits = 1;
    kt[(size_t)l] = (q.K + 63) / 64;
    ct[(size_t)l] = q.C / 64;
    total_units += (long)kt[(size_t)l] * ct[(size_t)l] * q.ntiles;
    plan->flops += 2.0 * (double)q.N * q.H * q.W * q.K * 9.0 * q.C;
    plan->bytes += ((double)q.N * q.Hi * q.Wi * q.C + (double)q.N * q.H * q.W * q.K) * 2.0 + 9.0 * q.K * q.C * 4.0;
  }
  const int nwg = target_blocks;
  const long chunk = (total_units + nwg - 1) / nwg;
  constexpr long MINSEG = 4;                              // a segment pays a full epilogue (~3 units of work): no crumbs
  std::vector<WhSeg> segs;
  std::vector<int> wg_first;
  std::vector<WhTileRed> tred;
  int wg = 0;
  long room = chunk;
  wg_first.push_back(0);
  auto next_wg = [&]() {
    if (wg + 1 < nwg) {
      ++wg;
      wg_first.push_back((int)segs.size());
      room = chunk;
    } else {
      room = 1L << 40;                                   // the last workgroup takes whatever is left
    }
  };
  for (int l = 0; l < n; ++l) {
    const WhParams& q = layers[(size_t)l];
    for (int ki = 0; ki < kt[(size_t)l]; ++ki)
      for (int ci = 0; ci < ct[(size_t)l]; ++ci) {
        WhTileRed tr;
        tr.dw = q.dw + (size_t)ki * 64 * 9 * q.C + (size_t)ci * 64;
        tr.ldc = q.C;
        tr.first = (int)segs.size();
        tr.krows = q.K - ki * 64 < 64 ? q.K - ki * 64 : 64;
        int t = 0;
        while (t < q.ntiles) {
          long left = q.ntiles - t;
          if (room < MINSEG && wg_first.back() != (int)segs.size()) next_wg();       // do not start a crumb at the end of a range
          long m = left < room ? left : room;
          if (left - m < MINSEG) m = left;                                           // do not leave a crumb behind
          segs.push_back(WhSeg{l, ki * 64, ci * 64, t, (int)(t + m), (int)segs.size()});
          t += (int)m;
          room -= m;
          if (room <= 0) next_wg();
        }
        tr.n = (int)segs.size() - tr.first;
        tred.push_back(tr);
      }
  }
  while ((int)wg_first.size() < nwg + 1) wg_first.push_back((int)segs.size());      // workgroups without work (tiny problems)
  wg_first[(size_t)nwg] = (int)segs.size();
  plan->nwg = nwg; plan->nsegs = (int)segs.size(); plan->ntred = (int)tred.size(); plan->nlayers = n;
  size_t off = 0;
  auto take = [&](size_t b) { const size_t o = off; off = (off + b + 255) & ~(size_t)255; return o; };
  plan->off_layers = take(layers.size() * sizeof(WhParams));
  plan->off_segs = take(segs.size() * sizeof(WhSeg));
  plan->off_wgfirst = take(wg_first.size() * sizeof(int));
  plan->off_tred = take(tred.size() * sizeof(WhTileRed));
  VK_CHECK_ARG(off <= tables_bytes, "wgrad batch: tables need %zu bytes, %zu given", off, tables_bytes);
  plan->slab_need = segs.size() * (size_t)(64 * 9 * 64) * sizeof(float);
  char* tb = (char*)tables;
  VK_CHECK_HIP(hipMemcpy(tb + plan->off_layers, layers.data(), layers.size() * sizeof(WhParams), hipMemcpyHostToDevice));
  VK_CHECK_HIP(hipMemcpy(tb + plan->off_segs, segs.data(), segs.size() * sizeof(WhSeg), hipMemcpyHostToDevice));
  VK_CHECK_HIP(hipMemcpy(tb + plan->off_wgfirst, wg_first.data(), wg_first.size() * sizeof(int), hipMemcpyHostToDevice));
  VK_CHECK_HIP(hipMemcpy(tb + plan->off_tred, tred.data(), tred.size() * sizeof(WhTileRed), hipMemcpyHostToDevice));
  return VK_OK;
}

int wgrad_batch_launch(vk_dtype dt, const WgradBatchPlan& plan, const void* tables, void* slab, size_t slab_bytes, hipStream_t st) {
  VK_CHECK_ARG(dt != VK_F32 && plan.nwg > 0 && slab && plan.slab_need <= slab_bytes, "wgrad batch: slab of %zu bytes needed, %zu given",
               plan.slab_need, slab_bytes);
  using Cfg = WhCfg<bf16_t, 64, 64, false, true, 1>;
  static bool attr_done = false;
  if (!attr_done) {
    VK_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_halo_batch_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
    VK_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_halo_batch_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
    attr_done = true;
  }
  const char* tb = (const char*)tables;
  {
    vkh::ProfScope ps("wgrad_halo_16b_64x64ts_batch", st, plan.flops, plan.bytes);
    if (dt == VK_BF16)
      hipLaunchKernelGGL(wgrad_halo_batch_kernel<bf16_t>, dim3((unsigned)plan.nwg), dim3(512), Cfg::SMEM, st, (const WhParams*)(tb + plan.off_layers),
                         (const WhSeg*)(tb + plan.off_segs), (const int*)(tb + plan.off_wgfirst), (float*)slab);
    else
      hipLaunchKernelGGL(wgrad_halo_batch_kernel<f16_t>, dim3((unsigned)plan.nwg), dim3(512), Cfg::SMEM, st, (const WhParams*)(tb + plan.off_layers),
                         (const WhSeg*)(tb + plan.off_segs), (const int*)(tb + plan.off_wgfirst), (float*)slab);
  }
  {
    vkh::ProfScope ps("wgrad_slab_reduce", st, 0.0, (double)plan.slab_need + 2.0 * plan.ntred * 64.0 * 9.0 * 64.0 * 4.0);
    hipLaunchKernelGGL(k_wgrad_tile_reduce, dim3((unsigned)plan.ntred, 4), dim3(256), 0, st, (const WhTileRed*)(tb + plan.off_tred), (const float*)slab);
  }
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

}  // namespace vk

// C ABI: weight gradients of n layers (every one must satisfy vk_conv_wgrad_batch_supports) in one launch.  `tables`: device scratch of
// VK_WGRAD_BATCH_TABLE_BYTES that the call fills (synchronously) before it launches; `workspace`: the slab workspace of vk_conv_wgrad.
extern "C" int vk_conv_wgrad_batch_supports(const vk_conv_desc* d) { return d && vk::wgrad_batch_supports(d) ? 1 : 0; }
extern "C" int vk_conv_wgrad_batch(const vk_conv_desc* descs, const void* const* dz, float* const* dw, int n, int workgroups, void* tables,
                                   size_t tables_bytes, void* workspace, size_t workspace_bytes, void* stream) {
  VK_CHECK_ARG(descs && dz && dw && tables && workspace, "vk_conv_wgrad_batch: null argument");
  for (int l = 0; l < n; ++l) {
    VK_CHECK_ARG(vk::wgrad_batch_supports(&descs[l]), "vk_conv_wgrad_batch: layer %d is not of the batched class", l);
    VK_CHECK_ARG(descs[l].dtype == descs[0].dtype, "vk_conv_wgrad_batch: mixed element types");
  }
  // the tables are refilled with synchronous copies below: a previous call's kernels on `stream` may still be READING them (and the
  // slab) when the caller reuses the buffers, so wait for the stream first (the engine's own batches do the same in
  // flush_wgrads).  One host wait per call — this entry is the convenience form, the engine keeps a prebuilt plan.
  VK_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  vk::WgradBatchPlan plan;
  const int rc = vk::wgrad_batch_build(descs, dz, dw, n, workgroups > 0 ? workgroups : 256 - vkh::reserved_cus(), tables, tables_bytes, &plan);
  if (rc != VK_OK) return rc;
  return vk::wgrad_batch_launch(descs[0].dtype, plan, tables, workspace, workspace_bytes, (hipStream_t)stream);
}

namespace vk {
}  // namespace vk
