// 3x3 stride-1 pad-1 convolution with an LDS-staged input halo tile ("LDS-staged 3x3 input/weight tiles"
// of the north star).  Covers 27 encoder convs, all 10 decoder convs and the data-gradients of all of
// them (a 3x3 s1 dgrad is the same conv with flipped taps and [C][RS][K] weights) — >85 % of the conv
// FLOPs of the U-Net (reference train.py:436 forward, :443/:448 backward).
//
// Per workgroup: a TH x 16 pixel tile of one image x BN output channels.
//   * per 64-byte channel chunk (32 x 16-bit or 16 x fp32 channels) the (TH+2) x 18 halo is staged ONCE
//     (BatchNorm scale/shift + ReLU, nearest-x2 upsample and channel concat applied on the way) and feeds
//     all 9 taps: 9x less gather traffic / prologue VALU / LDS writes than tap-by-tap implicit GEMM.
//   * weights stream through LDS one filter row (3 taps) at a time, double buffered; the next chunk's
//     halo and the next row's weights are prefetched into registers underneath the MFMAs; one barrier per
//     filter row (48-96 MFMAs per wave between barriers).
//   * LDS image: 64-byte rows, 16-byte chunk j of row r stored at j ^ (((r >> 2) & 1) << 1): conflict-free
//     ds_read_b128 for 16 consecutive rows at ANY row offset (tap shifts), no padding.
//   * 16x16 MFMA tiles, "swapped" orientation, epilogue through LDS with BN partial sums — as conv_igemm.
#include <stdlib.h>
#include <string.h>

#include <string>
#include <type_traits>

#include "vk_common.h"

#ifdef VK_STAMP
#define VK_T(var) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); var = t_; }
#else
#define VK_T(var)
#endif

namespace vk {

struct HaloSrc {
  const void* ptr;
  const float* scale;
  const float* shift;
  int C, up, relu;
  uint32_t bytes;
};

// -DVK_COLQ_PRE=0: the pipelined K >= 128 kernel's epilogue requests the accumulate / BN-backward operands pass by pass (r03 form)
#ifndef VK_EPI_BATCH_READS
#define VK_EPI_BATCH_READS 1    // plain-store epilogue: the tile rows of up to 8 passes are read from LDS up front (0: one read -> wait -> store per pass, r03)
#endif
#ifndef VK_COLP_DEFER_STATS
#define VK_COLP_DEFER_STATS 1   // persistent K < 128 kernel: BN sums accumulate across a workgroup's tiles, one atomic pair at the end (0: per tile, r03)
#endif
#ifndef VK_COLQ_ILV
#define VK_COLQ_ILV 1           // weight DMAs of a stage issued between the MFMAs of filter row 2 (0: in front of them, r03)
#endif
#ifndef VK_COLQ_DIAG
#define VK_COLQ_DIAG 0          // knock-out timing builds of the pipelined K >= 128 loop (WRONG results): 1 one weight DMA per wave and stage, 2 no halo store, 4 no halo loads
#endif
#ifndef VK_COLQ_PRE
#define VK_COLQ_PRE 1
#endif
struct HaloParams {
  HaloSrc s0, s1;
  const void* w;
  uint32_t w_bytes;
  void* y0;
  void* y1;
  int ld0, ld1, split;
  double* stats;
  int N, H, W, K, C, flip, accumulate, pool2;      // H, W: OUTPUT size (= input size for the stride-1 kernels)
  int Hi, Wi;                                       // input size (stride-2 forward: 2 * H, 2 * W)
  int tiles_x, tiles_y, nchunks;
  // fused BatchNorm+ReLU backward reduce on the first output part: g = y * [z*scale+shift > 0] is stored instead of y and
  // sum(g), sum(g*z) go to bnr_sums [VK_STATS_REPLICAS][2][ld0]
  unsigned long long* stamps;   // diagnostic builds (-DVK_STAMP) only
  int kyn;                      // > 0: 1-D grid, the kyn channel tiles of one pixel tile are neighbours ON ONE XCD (see conv3x3_col_kernel)
  int ksplit;                   // > 1: blockIdx.z owns a slice of the channel chunks and writes an fp32 partial tile into `slab`
  size_t slab_bytes;
  float* slab;                  //      [ksplit][N*H*W][K]; k_splitk_reduce adds the slices up in a fixed order (small grids: batch-1 inference)
  int dbg;                      // timing experiments (VK_COL_DBG, results are WRONG by construction): 1 = weight descriptor with zero
                                // records, 2 = no LDS-DMA in the loop, 4 = no halo refresh in the loop, 8 = no stage barriers
  const void* bnr_z;
  const float* bnr_scale;
  const float* bnr_shift;
  double* bnr_sums;
  const void* bnr_mask;         // != nullptr: ReLU mask = [bnr_mask > 0] (a residual block's stored output) instead of the affine of bnr_z
};

template <typename T, int TH, int BN, int WGM, int WGN>
struct HaloCfg {
  using Tr = ElemTraits<T>;
  static constexpr int EB = Tr::kBytes;
  static constexpr int VE = Tr::kVec;
  static constexpr int CK = 64 / EB;                 // channels per chunk
  static constexpr int HW_ = 18, HH = TH + 2, HPIX = HH * HW_;
  static constexpr int APS = 96;                     // halo pixel stride: 64 B + 32 B pad => conflict-free at any shift, addresses = base + immediate
  static constexpr int A_BYTES = HPIX * APS;         // single buffer (next chunk's halo waits in registers)
  static constexpr int B_BYTES = 3 * BN * 64;        // one filter row of weights, XOR-swizzled 64-B rows, double buffered
  static constexpr int NPV = (HPIX * 4 + 255) / 256;   // halo vectors per thread
  static constexpr int NBV = (3 * BN * 4 + 255) / 256; // weight vectors per thread per filter row
  static constexpr int BM = TH * 16;
  static constexpr int WR = TH / WGM;                  // tile rows per wave (= MFMA pixel tiles)
  static constexpr int WCH = BN / WGN;
  static constexpr int TP = WR, TC = WCH / 16;
  static constexpr int ESB = BN * EB + 16;
  static constexpr int EVPR = BN / VE;
  static constexpr int ERPP = 256 / EVPR;
  static constexpr int EPASS = BM / ERPP;
  static constexpr int RED_OFF = BM * ESB;
  static constexpr int MAIN = A_BYTES + 2 * B_BYTES;
  static constexpr int ESLOTS = (BN / VE > 16) ? 4 * 4 / (BN / VE / 16) : 4 * 4;
  static constexpr int EPI = RED_OFF + ESLOTS * BN * 2 * 4;       // + [wave x 16-lane row][channel][2] partial sums
  static constexpr int SMEM = MAIN > EPI ? MAIN : EPI;
  static_assert(WGM * WGN == 4 && TH % WGM == 0 && BN % (16 * WGN) == 0, "wave layout");
  static_assert(EPASS >= 1, "epilogue mapping");
  static_assert(SMEM <= 160 * 1024, "main-loop / epilogue LDS image exceeds the 160 KiB of a CU");
};

__device__ __forceinline__ int swz(int row) { return ((row >> 2) & 1) << 1; }

// Which halo pixel (of the 64 / 128 a pass covers) thread tid stages; piece = tid & 3.  The 8 lanes that one ds_write_b128 lane group
// holds are two pixels x four 16-byte pieces: with NEIGHBOURING pixels (96 bytes apart) the second pixel's pieces 2, 3 fall on the
// banks of the first one's pieces 0, 1 (128-byte bank row of the stores): a 2-way conflict on every halo store, 50 % of their LDS
// cycles (tools/lds_bank_sim.py).  Pixels p and p + 2 (192 bytes apart) tile the bank row exactly: bits 0 and 1 of the group index
// are swapped.  A permutation inside aligned groups of four pixels: every pixel is still staged exactly once, by another thread.
__device__ __forceinline__ int halo_group(int tid) {
  const int g = tid >> 2;
  return (g & ~3) | ((g & 1) << 1) | ((g >> 1) & 1);
}


// ---- shared epilogue: accumulators -> LDS [tile pixel][channel] as T -> 16-byte NHWC stores
//   * optional BN partial sums (sum / sum of squares of the stored values, fp64 atomics into replicated slabs)
//   * optional channel split (concat gradient): channels >= split go to y1
//   * optional pool2: the FIRST output part is summed over 2x2 pixel groups and written at half resolution
//     (nearest-x2 upsample backward fused into the data gradient; the full-resolution tensor never exists)
//   * OS = 2 (stride-2 data gradient): tile pixel (row, col) is output pixel (2 (y0 + row) + py, 2 (x0 + col) + px)
template <typename T, int TH, int BN, int TP, int TC, int NT = 256, bool PRE = false, int OS = 1>
__device__ __forceinline__ void halo_epilogue(char* smem, f32x4_t (&acc)[TC][TP], const HaloParams& p, int n, int y0, int x0, int n0,
                                              int wrow0, int wch0, int py = 0, int px = 0, double* defer = nullptr) {
  using Tr = ElemTraits<T>;
  constexpr int EB = Tr::kBytes, VE = Tr::kVec;
  constexpr int BM = TH * 16;
  constexpr int ESB = BN * EB + 16;
  constexpr int EVPR = BN / VE, ERPP = NT / EVPR, EPASS = BM / ERPP, NW = NT / 64;
  constexpr int RED_OFF = BM * ESB;
  static_assert(BM % ERPP == 0 && EVPR <= 64, "epilogue mapping");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kg = lane >> 4;

  // thread -> (tile pixel, 16-byte vector of its channel row).  BN = 128, 16-bit (rows of 272 bytes, 16 vectors): the odd rows take
  // their vectors rotated by one — the 16-lane groups of a ds_read_b128 then see 16 different 16-byte slots of the 256-byte bank row
  // (in thread order lanes 12-15 of row r and 20-23 of row r + 1 shared slots: 2-way, tools/lds_bank_sim.py); a per-thread constant
  // (ERPP is even), so a thread still owns the same 8 channels in every pass
  const int e_row = tid / EVPR;
  const int e_vec = (EVPR == 16 && EB == 2) ? (((tid & 15) + ((e_row & 1) ? 15 : 0)) & 15) : tid % EVPR;
  const int col0 = n0 + e_vec * VE;
  const bool col_ok = col0 < p.K;
  char* yb = (char*)p.y0;
  int ld = p.ld0, colx = col0;
  bool first_part = true;
  if (p.split > 0 && col0 >= p.split) { yb = (char*)p.y1; ld = p.ld1; colx = col0 - p.split; first_part = false; }
  float s1[VE], s2[VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  const bool bnr = p.bnr_z != nullptr && first_part && col_ok;
  const bool ext_mask = p.bnr_mask != nullptr;               // block-tail form (vk_bnr.mask)
  float bsc[VE], bsh[VE];
  if (bnr && !ext_mask) {
#pragma unroll
    for (int j = 0; j < VE; ++j) { bsc[j] = p.bnr_scale[colx + j]; bsh[j] = p.bnr_shift[colx + j]; }
  }
  // final value of one output vector: optional accumulate, optional BN+ReLU-backward masking + sums, store
  // pre_old / pre_z: the destination's previous content (accumulate) and the matching vector of bnr_z when the caller has
  // requested them ahead (PRE below); otherwise they are loaded here
  auto finish = [&](float (&f)[VE], size_t elem_off, const u32x4_t* pre_old, const u32x4_t* pre_z, const u32x4_t* pre_m = nullptr) {
    u32x4_t* gp = reinterpret_cast<u32x4_t*>(yb + elem_off * EB);
    if (p.accumulate) {
      float o[VE];
      Vec16<T>::unpack(pre_old ? *pre_old : *gp, o);
#pragma unroll
      for (int j = 0; j < VE; ++j) f[j] += o[j];
    }
    u32x4_t v = Vec16<T>::pack(f);
    Vec16<T>::unpack(v, f);                      // statistics are over the STORED (rounded) values
    if (bnr) {
      float zf[VE];
      Vec16<T>::unpack(pre_z ? *pre_z : *reinterpret_cast<const u32x4_t*>((const char*)p.bnr_z + elem_off * EB), zf);
      if (ext_mask) {
        float mf[VE];
        Vec16<T>::unpack(pre_m ? *pre_m : *reinterpret_cast<const u32x4_t*>((const char*)p.bnr_mask + elem_off * EB), mf);
#pragma unroll
        for (int j = 0; j < VE; ++j)
          if (!(mf[j] > 0.f)) f[j] = 0.f;
      } else {
#pragma unroll
        for (int j = 0; j < VE; ++j)
          if (!(fmaf(zf[j], bsc[j], bsh[j]) > 0.f)) f[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        s1[j] += f[j];
        s2[j] += f[j] * zf[j];
      }
      v = Vec16<T>::pack(f);
    } else {
#pragma unroll
      for (int j = 0; j < VE; ++j) { s1[j] += f[j]; s2[j] += f[j] * f[j]; }
    }
    *gp = v;
  };
  // r04: these requests are issued BEFORE the accumulators go through LDS (they depend on nothing the tile computed): the HBM round
  // trip of up to 3 x EPASS vectors per thread runs under the transposition and its barrier instead of behind them
  // PRE: request the accumulate / bnr operands of all EPASS vectors before the first one is finished (one memory round
  // trip instead of EPASS; pays on the 64-channel tiles with 8 passes per thread: L1 data gradient + reduce 73 -> 70 us,
  // costs registers, hence occupancy, on the C = 16 kernels: off there)
  constexpr bool PRE_M = PRE && NT == 512;       // the external-mask vectors ride along only where registers allow (4-wave 64-channel tile: 250 of 256 in use)
  u32x4_t oldv[PRE ? EPASS : 1], zv[PRE ? EPASS : 1], mv[PRE_M ? EPASS : 1];
  // early only on the 8-wave tiles (256 registers per wave): beside the live accumulators the 3 x EPASS vectors would cost the 4-wave
  // tiles their second workgroup per CU
  constexpr bool EARLY = PRE && NT == 512;
  auto request_pre = [&]() {
    if (!(PRE && (p.accumulate || bnr) && !(p.pool2 && first_part))) return;
#pragma unroll
    for (int ps = 0; ps < (PRE ? EPASS : 0); ++ps) {
      const int row = e_row + ps * ERPP;
      const int y = OS * (y0 + (row >> 4)) + py, x = OS * (x0 + (row & 15)) + px;
      oldv[ps] = u32x4_t{0, 0, 0, 0};
      zv[ps] = u32x4_t{0, 0, 0, 0};
      if (PRE_M) mv[ps] = u32x4_t{0, 0, 0, 0};
      if (y < p.H && x < p.W && col_ok) {
        const size_t eoff = (((size_t)n * p.H + y) * p.W + x) * ld + colx;
        if (p.accumulate) oldv[ps] = *reinterpret_cast<const u32x4_t*>(yb + eoff * EB);
        if (bnr) zv[ps] = *reinterpret_cast<const u32x4_t*>((const char*)p.bnr_z + eoff * EB);
        if (PRE_M && bnr && ext_mask) mv[ps] = *reinterpret_cast<const u32x4_t*>((const char*)p.bnr_mask + eoff * EB);
      }
    }
  };
  if (EARLY) request_pre();
  // ---- accumulators -> LDS [tile pixel][channel] as T
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      const int ch = wch0 + a * 16 + kg * 4;
      const int px = (wrow0 + b) * 16 + li;
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
  if (!EARLY) request_pre();

  if (p.pool2 && first_part) {
    // (TH/2) x 8 pooled pixels; tile origins are even, H and W are even
    const int Hh = p.H >> 1, Wh = p.W >> 1;
    for (int pr = e_row; pr < (TH / 2) * 8; pr += ERPP) {
      const int py = pr >> 3, px = pr & 7;
      const int y = (y0 >> 1) + py, x = (x0 >> 1) + px;
      if (y < Hh && x < Wh && col_ok) {
        float f[VE];
#pragma unroll
        for (int j = 0; j < VE; ++j) f[j] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = (2 * py + (q >> 1)) * 16 + 2 * px + (q & 1);
          float t[VE];
          Vec16<T>::unpack(*reinterpret_cast<const u32x4_t*>(smem + row * ESB + e_vec * 16), t);
#pragma unroll
          for (int j = 0; j < VE; ++j) f[j] += t[j];
        }
        finish(f, (((size_t)n * Hh + y) * Wh + x) * ld + colx, nullptr, nullptr);
      }
    }
  } else if (p.accumulate || p.bnr_z != nullptr) {          // (launch-uniform)
#pragma unroll
    for (int ps = 0; ps < EPASS; ++ps) {
      const int row = e_row + ps * ERPP;          // tile pixel index
      const int y = OS * (y0 + (row >> 4)) + py, x = OS * (x0 + (row & 15)) + px;
      if (y < p.H && x < p.W && col_ok) {
        const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(smem + row * ESB + e_vec * 16);
        const size_t eoff = (((size_t)n * p.H + y) * p.W + x) * ld + colx;
        float f[VE];
        Vec16<T>::unpack(raw, f);
        if (PRE) finish(f, eoff, &oldv[ps], &zv[ps], PRE_M ? &mv[ps] : nullptr);
        else finish(f, eoff, nullptr, nullptr);
      }
    }
  } else {
    // plain store: the LDS tile already holds the rounded values, so they go out as they are (no pack / unpack round trip) and only
    // a launch that wants BN statistics pays for the sums.  r04: the tile rows of up to 8 passes are read from LDS up front (in
    // bounds for every thread) instead of one dependent ds_read -> wait -> store chain per pass behind the bounds branch
    // (VK_EPI_BATCH_READS=0: the r03 order); the operand vectors of the path above are dead here, so this costs no registers.
    // Measured null on layers 1-2 (profiles/r04/epi_batch_ab.log): the other workgroup of the CU hides the chain
    constexpr int RB = VK_EPI_BATCH_READS ? (EPASS < 8 ? EPASS : 8) : 1;
    u32x4_t raws[RB];
#pragma unroll
    for (int ps = 0; ps < EPASS; ++ps) {
      if (ps % RB == 0) {
#pragma unroll
        for (int q = 0; q < RB; ++q)
          if (ps + q < EPASS) raws[q] = *reinterpret_cast<const u32x4_t*>(smem + (e_row + (ps + q) * ERPP) * ESB + e_vec * 16);
      }
      const int row = e_row + ps * ERPP;
      const int y = OS * (y0 + (row >> 4)) + py, x = OS * (x0 + (row & 15)) + px;
      if (y < p.H && x < p.W && col_ok) {
        const u32x4_t raw = raws[ps % RB];
        const size_t eoff = (((size_t)n * p.H + y) * p.W + x) * ld + colx;
        if (p.stats) {
          float f[VE];
          Vec16<T>::unpack(raw, f);
#pragma unroll
          for (int j = 0; j < VE; ++j) { s1[j] += f[j]; s2[j] += f[j] * f[j]; }
        }
        *reinterpret_cast<u32x4_t*>(yb + eoff * EB) = raw;
      }
    }
  }
  if (p.stats || p.bnr_sums) {
    if (p.bnr_sums && !bnr) {       // threads outside the first output part contribute nothing to the BN sums
#pragma unroll
      for (int j = 0; j < VE; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    }
    // lanes holding the same channels sit EVPR apart.  Inside a 16-lane row they are added with DPP moves (VALU rate; the
    // former ds_bpermute butterfly was 16 * log2(64 / EVPR) LDS-pipe instructions per thread and cost the 512x512 decoder
    // layers a third of their time); the four rows of every wave go through LDS to the thread that owns the channel.
#pragma unroll
    for (int j = 0; j < VE; ++j) {
      if (EVPR <= 1) { s1[j] += dpp_f<0xB1>(s1[j]); s2[j] += dpp_f<0xB1>(s2[j]); }       // quad_perm [1,0,3,2]
      if (EVPR <= 2) { s1[j] += dpp_f<0x4E>(s1[j]); s2[j] += dpp_f<0x4E>(s2[j]); }       // quad_perm [2,3,0,1]
      if (EVPR <= 4) { s1[j] += dpp_f<0x124>(s1[j]); s2[j] += dpp_f<0x124>(s2[j]); }     // row_ror:4
      if (EVPR <= 8) { s1[j] += dpp_f<0x128>(s1[j]); s2[j] += dpp_f<0x128>(s2[j]); }     // row_ror:8
    }
    float* red = reinterpret_cast<float*>(smem + RED_OFF);
    // slot = one full set of BN channel sums: a 16-lane row when it holds every vector of a tile row (EVPR <= 16), otherwise
    // the EVPR / 16 neighbouring rows that hold them together (fp32, BN = 128)
    constexpr int RPS = EVPR > 16 ? EVPR / 16 : 1, NSLOT = NW * 4 / RPS;
    static_assert(EVPR <= 32, "partial-sum slots");
    if (EVPR > 16 || li < EVPR) {
      const int slot = (wave * 4 + kg) / RPS;
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        red[(slot * BN + e_vec * VE + j) * 2 + 0] = s1[j];
        red[(slot * BN + e_vec * VE + j) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.K) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < NSLOT; ++w) { a += red[(w * BN + tid) * 2]; b += red[(w * BN + tid) * 2 + 1]; }
      if (defer) {                     // persistent callers: the tile's sums join the workgroup's running fp64 pair (stats_flush)
        defer[0] += (double)a;
        defer[1] += (double)b;
      } else if (p.bnr_sums) {
        const int kc = p.ld0;          // channel count of the first output part
        if (n0 + tid < kc) {
          double* sp = p.bnr_sums + (size_t)(blockIdx.x % VK_STATS_REPLICAS) * 2 * kc;
          atomicAdd(sp + n0 + tid, (double)a);
          atomicAdd(sp + kc + n0 + tid, (double)b);
        }
      } else {
        double* sp = p.stats + (size_t)(blockIdx.x % VK_STATS_REPLICAS) * 2 * p.K;
        atomicAdd(sp + n0 + tid, (double)a);
        atomicAdd(sp + p.K + n0 + tid, (double)b);
      }
    }
  }
}

// the deferred form of the epilogue's statistics atomics: thread tid < BN adds the running sums of channel n0 + tid (all tiles a
// persistent workgroup has finished on this channel tile) to the workgroup's replica — one pair of fp64 atomics per workgroup and
// channel instead of one per tile (layer 1: 4 tiles per workgroup, the 512 x 512 decoder layers: 16-64)
template <int BN>
__device__ __forceinline__ void stats_flush(const HaloParams& p, int n0, double (&d)[2]) {
  const int tid = threadIdx.x;
  if (tid < BN && n0 + tid < p.K) {
    if (p.bnr_sums) {
      const int kc = p.ld0;
      if (n0 + tid < kc) {
        double* sp = p.bnr_sums + (size_t)(blockIdx.x % VK_STATS_REPLICAS) * 2 * kc;
        atomicAdd(sp + n0 + tid, d[0]);
        atomicAdd(sp + kc + n0 + tid, d[1]);
      }
    } else if (p.stats) {
      double* sp = p.stats + (size_t)(blockIdx.x % VK_STATS_REPLICAS) * 2 * p.K;
      atomicAdd(sp + n0 + tid, d[0]);
      atomicAdd(sp + p.K + n0 + tid, d[1]);
    }
  }
  d[0] = 0.0;
  d[1] = 0.0;
}

template <typename T, int TH, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(const HaloParams p) {
  using Cfg = HaloCfg<T, TH, BN, WGM, WGN>;
  constexpr int EB = Cfg::EB, VE = Cfg::VE, CK = Cfg::CK, HPIX = Cfg::HPIX, NPV = Cfg::NPV, NBV = Cfg::NBV;
  constexpr int TP = Cfg::TP, TC = Cfg::TC, APS = Cfg::APS;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Abuf = smem;
  char* const Bbuf = smem + Cfg::A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // block -> (image, tile y, tile x), channel tile
  int bt = blockIdx.x;
  // workgroups go round-robin over the 8 XCDs: give every XCD (its own L2) a contiguous run of tiles so that horizontally
  // adjacent tiles share their halo columns in one L2 (kernel level +1-2 % on the HBM-bound layers, null on the step)
  if ((gridDim.x & 7) == 0) bt = (bt & 7) * (int)(gridDim.x >> 3) + (bt >> 3);
  const int tx = bt % p.tiles_x;
  bt /= p.tiles_x;
  const int ty = bt % p.tiles_y;
  const int n = bt / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * 16;
  const int n0 = blockIdx.y * BN;

  const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(p.s0.ptr, p.s0.bytes);
  const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(p.s1.ptr ? p.s1.ptr : p.s0.ptr, p.s1.ptr ? p.s1.bytes : 0u);
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.w, p.w_bytes);

  // ---- per-thread staging geometry, computed once (the main loop only adds block-uniform scalars)
  const int hv = tid & 3;                       // 16-byte vector inside the 64-byte pixel chunk (fixed per thread)
  const int hgrp = halo_group(tid);
  int h_full[NPV], h_half[NPV];                 // pixel index in a full-res / half-res (upsampled) source, -1 = outside
  const int Hh = p.H >> 1, Wh = p.W >> 1;
#pragma unroll
  for (int i = 0; i < NPV; ++i) {
    const int hp = hgrp + i * 64;
    const int hy = hp / 18, hx = hp - hy * 18;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = hp < HPIX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    h_full[i] = ok ? (n * p.H + y) * p.W + x : -1;
    h_half[i] = ok ? (n * Hh + (y >> 1)) * Wh + (x >> 1) : -1;
  }
  // weights come in the "halo pack": [chunk][tap][K rows][64 B], the 16-byte pieces of a row already XOR-swizzled the way
  // the LDS image wants them.  A stage (chunk, filter row) = three contiguous BN x 64-byte slabs -> every wave-instruction
  // copies 1 KiB of consecutive bytes (the stamp profile showed the former [K][tap][C] gather costing as much as the MFMAs).
  // Data-gradient mode walks the taps backwards (tap 8 - t) instead of flipping the halo accesses.
  int b_toff[NBV];                               // byte offset of this thread's vector inside its slab, -1 = unused slot
  int b_s[NBV];                                  // which of the three taps of the stage
#pragma unroll
  for (int i = 0; i < NBV; ++i) {
    const int idx = tid + i * 256;
    b_s[i] = idx / (BN * 4);
    b_toff[i] = idx < 3 * BN * 4 ? (idx - b_s[i] * (BN * 4)) * 16 : -1;
  }
  const uint32_t slab_bytes = (uint32_t)p.K * 64u;         // one (chunk, tap) slab

  u32x4_t areg[NPV], breg0[NBV], breg1[NBV];   // weights are prefetched TWO filter rows ahead (two register sets)
  float sc[VE], sh[VE];
  bool aff = false, relu = false;

  auto load_halo = [&](int cc) {
    const int c = cc * CK;
    const bool first = c < p.s0.C;
    const HaloSrc& sd = first ? p.s0 : p.s1;
    const int cl = (first ? c : c - p.s0.C) + hv * VE;
    aff = sd.scale != nullptr;
    relu = sd.relu != 0;
    if (aff) {
#pragma unroll
      for (int j = 0; j < VE; j += 4) {
        const f32x4_t s4 = *reinterpret_cast<const f32x4_t*>(sd.scale + cl + j);
        const f32x4_t h4 = *reinterpret_cast<const f32x4_t*>(sd.shift + cl + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[j + e] = s4[e]; sh[j + e] = h4[e]; }
      }
    }
    const bool up = sd.up != 0;
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int pix = up ? h_half[i] : h_full[i];
      const uint32_t off = (uint32_t)(pix * sd.C + cl) * (uint32_t)EB;
      if (first) areg[i] = buf_load16(rs0, pix >= 0 ? off : kOOB);
      else areg[i] = buf_load16(rs1, pix >= 0 ? off : kOOB);
    }
  };

  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int hp = hgrp + i * 64;
      if (hp >= HPIX) continue;
      u32x4_t v = areg[i];
      if (aff) {
        v = AffineRelu<T>::run(v, sc, sh, relu);
        if (h_full[i] < 0) v = u32x4_t{0, 0, 0, 0};     // zero padding is applied after BN+ReLU
      }
      *reinterpret_cast<u32x4_t*>(Abuf + hp * APS + hv * 16) = v;
    }
  };

  // weights of filter row r, chunk cc: rows n0..n0+BN, three taps
  auto load_b = [&](u32x4_t (&breg)[NBV], int stage) {
    const int cc = stage / 3, r = stage - cc * 3;
#pragma unroll
    for (int i = 0; i < NBV; ++i) {
      const int tap = p.flip ? 8 - (3 * r + b_s[i]) : 3 * r + b_s[i];
      const uint32_t base = (uint32_t)(cc * 9 + tap) * slab_bytes + (uint32_t)n0 * 64u;     // block-uniform per tap
      breg[i] = buf_load16(rsw, b_toff[i] >= 0 ? base + (uint32_t)b_toff[i] : kOOB);
    }
  };
  auto store_b = [&](const u32x4_t (&breg)[NBV], int buf) {
    char* B = Bbuf + buf * Cfg::B_BYTES;
#pragma unroll
    for (int i = 0; i < NBV; ++i)
      if (b_toff[i] >= 0) *reinterpret_cast<u32x4_t*>(B + b_s[i] * (BN * 64) + b_toff[i]) = breg[i];     // linear copy
  };

  // ---- accumulators and per-lane fragment bases (fragment reads are base + immediate)
  const int wrow0 = (wave / WGN) * Cfg::WR;
  const int wch0 = (wave % WGN) * Cfg::WCH;
  f32x4_t acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int li = lane & 15, kg = lane >> 4;
  const char* const a_lane = Abuf + (wrow0 * 18 + li) * APS + kg * 16;
  const int b_lane = (wch0 + li) * 64 + ((kg ^ swz(li)) << 4);       // swz(row) only depends on li: rows step by 16

  auto compute = [&](int bbuf, int r) {
    const char* A = a_lane + r * (18 * APS);
    const char* B = Bbuf + bbuf * Cfg::B_BYTES + b_lane;
    // fragments of tap s+1 are read before the MFMAs of tap s are issued (explicit one-tap software pipeline)
    u32x4_t wf[2][TC], xf[2][TP];
#pragma unroll
    for (int a = 0; a < TC; ++a) wf[0][a] = *reinterpret_cast<const u32x4_t*>(B + (a * 16) * 64);
#pragma unroll
    for (int b = 0; b < TP; ++b) xf[0][b] = *reinterpret_cast<const u32x4_t*>(A + (b * 18) * APS);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (s < 2) {
#pragma unroll
        for (int a = 0; a < TC; ++a) wf[(s + 1) & 1][a] = *reinterpret_cast<const u32x4_t*>(B + ((s + 1) * BN + a * 16) * 64);
#pragma unroll
        for (int b = 0; b < TP; ++b) xf[(s + 1) & 1][b] = *reinterpret_cast<const u32x4_t*>(A + (b * 18 + s + 1) * APS);
      }
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[a][b] = Mma<T>::run(wf[s & 1][a], xf[s & 1][b], acc[a][b]);
    }
    // pin the issue order: [reads tap0][reads tap1][MFMAs tap0][reads tap2][MFMAs tap1][MFMAs tap2]
    constexpr int NR = TC + TP, NM = TC * TP * (sizeof(T) == 4 ? 4 : 1);
    __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
  };

#ifdef VK_STAMP
  unsigned long long t_begin;
  VK_T(t_begin)
#endif
  // ---- pipeline over (chunk, filter row) stages.  Halo: single LDS buffer, next chunk waits in registers for three
  // stages.  Weights: two LDS buffers + two register sets, i.e. the global loads of stage s+2 are issued before the
  // MFMAs of stage s and written to LDS after the MFMAs of stage s+1 (a full stage of slack for the L2/HBM latency).
  const int nst = p.nchunks * 3;
  load_halo(0);
  load_b(breg0, 0);
  store_halo();
  store_b(breg0, 0);
  if (nst > 1) load_b(breg1, 1);
  __syncthreads();

#ifdef VK_STAMP
  unsigned long long tk_load = 0, tk_comp = 0, tk_store = 0, tk_bar = 0;
#endif
  auto stage = [&](int st, u32x4_t (&ld_regs)[NBV], const u32x4_t (&st_regs)[NBV]) {
    const int cc = st / 3, r = st - cc * 3;
    const bool next_chunk = cc + 1 < p.nchunks;
#ifdef VK_STAMP
    unsigned long long t0, t1, t2, t3, t4;
#endif
    VK_T(t0)
    if (st + 2 < nst) load_b(ld_regs, st + 2);
    if (r == 0 && next_chunk) load_halo(cc + 1);
    VK_T(t1)
    compute(st & 1, r);
    VK_T(t2)
    if (r == 2 && next_chunk) {
      __syncthreads();                                  // every wave is done reading this chunk's halo
      store_halo();
    }
    if (st + 1 < nst) store_b(st_regs, (st + 1) & 1);
    VK_T(t3)
    __syncthreads();
    VK_T(t4)
#ifdef VK_STAMP
    tk_load += t1 - t0; tk_comp += t2 - t1; tk_store += t3 - t2; tk_bar += t4 - t3;
#endif
  };
  for (int st = 0; st < nst; st += 2) {
    stage(st, breg0, breg1);                            // even stage: reads B[0], stores set 1 -> B[1], loads set 0
    if (st + 1 < nst) stage(st + 1, breg1, breg0);      // odd stage : reads B[1], stores set 0 -> B[0], loads set 1
  }

#ifdef VK_STAMP
  unsigned long long te0, te1;
  VK_T(te0)
#endif
  halo_epilogue<T, TH, BN, TP, TC>(smem, acc, p, n, y0, x0, n0, wrow0, wch0);
#ifdef VK_STAMP
  VK_T(te1)
  if (p.stamps && lane == 0) {
    unsigned long long* o = p.stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
    o[0] = tk_load; o[1] = tk_comp; o[2] = tk_store; o[3] = tk_bar; o[4] = te1 - te0; o[5] = te0 - t_begin; o[6] = te1 - t_begin; o[7] = nst;
  }
#endif
}

// ---- column-staged variant (v2 main loop).  Same tile / halo / epilogue as above, different dataflow:
//   * a stage is (channel chunk, filter COLUMN s): the wave reads its TP+2 halo rows once at column shift s and uses each of them
//     for up to three filter rows (row h feeds output rows h, h-1, h-2) -> (TP + 2 + 3 TC) fragment reads per 3 TP TC MFMAs
//     instead of 3 (TP + TC): 0.375 reads per MFMA at TP = TC = 4, 0.23 at TP = 8 (the row-staged loop sits at 0.5 and is
//     LDS-bound: LDS reads + ds_write_b128 of the weights come to ~90 % of the LDS cycles at full MFMA rate).
//   * weights never touch a VGPR: the halo pack is the LDS image, so a stage's three taps are 3 BN / 16 linear 1-KiB pieces
//     copied by LDS-DMA (buffer_load_dwordx4 ... lds), issued one stage ahead into the other weight buffer and retired by
//     the vmcnt(0) of the stage's closing barrier.
//   * WGM x WGN waves (4 or 8) share one halo and one weight stream: an 8-wave 16x16x128 tile halves the per-MFMA weight
//     traffic (L2 -> LDS) of the 4-wave 8x16x128 tile.
//   * ADB: the halo image is double buffered (next chunk written during stage 1, no extra barrier); otherwise it is written
//     after an extra barrier in stage 2 (less LDS -> two workgroups per CU).
//   * STR = 2 (the three stride-2 3x3 convolutions that open layers 2-4, forward): the (2 TH + 1) x 33 input pixels under a TH x 16
//     output tile are staged with the odd and the even input columns in separate planes of every LDS row (slot = row * 34 +
//     (x & 1) * 17 + (x >> 1)), so that output pixel li of tap column s still reads slot li + const: the fragment reads keep the
//     conflict-free 96-byte stride of the stride-1 image; tile row b of filter row r reads halo row 2 b + r.
template <typename T, int TH, int BN, int WGM, int WGN, bool ADB, int STR = 1>
struct ColCfg {
  using Tr = ElemTraits<T>;
  static constexpr int EB = Tr::kBytes;
  static constexpr int VE = Tr::kVec;
  static constexpr int CK = 64 / EB;
  static constexpr int NW = WGM * WGN, NT = 64 * NW;
  static constexpr int HH = STR == 1 ? TH + 2 : 2 * TH + 1;         // staged input rows
  static constexpr int HWI = STR == 1 ? 18 : 33;                    // staged input columns
  static constexpr int RS = STR == 1 ? 18 : 34;                     // LDS slots per row
  static constexpr int HPIX = HH * HWI;
  static constexpr int APS = 96;
  static constexpr int A_BYTES = HH * RS * APS;
  static constexpr int NA = ADB ? 2 : 1;
  static constexpr int B_BYTES = 3 * BN * 64;
  static constexpr int NPV = (HPIX * 4 + NT - 1) / NT;
  static constexpr int NPIECE = 3 * BN / 16;           // 1-KiB LDS-DMA pieces per stage
  static constexpr int BM = TH * 16;
  static constexpr int TP = TH / WGM, TC = BN / WGN / 16;
  static constexpr int NX = STR * (TP - 1) + 3;                     // halo rows a wave reads per stage
  static constexpr int ESB = BN * EB + 16;
  // weight stage buffers: two (double buffering); three for the small channel tiles, so that a ONE-chunk reduction (decoder blocks 3/4:
  // C = 32) gets the weights of all three filter columns in one LDS-DMA round and runs its stages without a barrier between them
  static constexpr int NB = (BN <= 64 && STR == 1) ? 3 : 2;
  static constexpr int MAIN = NA * A_BYTES + NB * B_BYTES;
  static constexpr int ESLOTS = (BN / VE > 16) ? NW * 4 / (BN / VE / 16) : NW * 4;
  static constexpr int EPI = BM * ESB + ESLOTS * BN * 2 * 4;     // + [wave x 16-lane row][channel][2] partial sums
  static constexpr int SMEM = MAIN > EPI ? MAIN : EPI;
  static_assert(TH % WGM == 0 && BN % (16 * WGN) == 0 && (NW == 4 || NW == 8), "wave layout");
  static_assert(STR == 1 || STR == 2, "stride");
  static_assert(SMEM <= 160 * 1024, "main-loop / epilogue LDS image exceeds the 160 KiB of a CU");
  static_assert(EPI >= BM * ESB + ESLOTS * BN * 8, "epilogue partial-sum slots must fit behind the output tile");
};

template <typename T, int TH, int BN, int WGM, int WGN, bool ADB, int MINW, int STR = 1>
__global__ __launch_bounds__(64 * WGM * WGN, MINW) void conv3x3_col_kernel(const HaloParams p) {
  using Cfg = ColCfg<T, TH, BN, WGM, WGN, ADB, STR>;
  constexpr int EB = Cfg::EB, VE = Cfg::VE, CK = Cfg::CK, HPIX = Cfg::HPIX, NPV = Cfg::NPV, NT = Cfg::NT, NW = Cfg::NW;
  constexpr int TP = Cfg::TP, TC = Cfg::TC, APS = Cfg::APS, NPIECE = Cfg::NPIECE, RS = Cfg::RS, HWI = Cfg::HWI, NX = Cfg::NX;
  typedef __attribute__((address_space(3))) void lds_void;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Abuf = smem;
  char* const Bbuf = smem + Cfg::NA * Cfg::A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int bt = blockIdx.x;
  int ky = blockIdx.y;
  // workgroups go round-robin over the 8 XCDs: give every XCD (its own L2) a contiguous run of tiles so that horizontally
  // adjacent tiles share their halo columns in one L2 (kernel level +1-2 % on the HBM-bound layers, null on the step)
  if (p.kyn > 0) {
    // K > BN: the channel tiles of ONE pixel tile read the same halo.  1-D grid, id = xcd + 8 * (ky + kyn * t): they are
    // consecutive workgroups of one XCD, so the second one finds the halo in that XCD's L2 instead of re-reading HBM
    // (measured before: 1.34x the algorithmic traffic on the K = 256 layers).  Speed only: any placement gives the same result.
    const int j = bt >> 3;
    ky = j % p.kyn;
    bt = (bt & 7) * (int)((gridDim.x / p.kyn) >> 3) + j / p.kyn;
  } else if (p.kyn < 0) {
    // weights-stationary placement: XCD x runs channel tile x % kn only, so its L2 holds 1 / kn of the layer's weights (the
    // stream every workgroup re-reads, 3-5 MB per layer on the K >= 256 layers against 4 MB of L2) and the pixel tiles' halos
    // are fetched by kn XCDs instead of one
    const int kn = -p.kyn, xcd = bt & 7;
    ky = xcd % kn;
    bt = (xcd / kn) * (int)(gridDim.x >> 3) + (bt >> 3);
  } else if ((gridDim.x & 7) == 0) {
    bt = (bt & 7) * (int)(gridDim.x >> 3) + (bt >> 3);
  }
  const int tx = bt % p.tiles_x;
  bt /= p.tiles_x;
  const int ty = bt % p.tiles_y;
  const int n = bt / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * 16;
  const int n0 = ky * BN;

  const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(p.s0.ptr, p.s0.bytes);
  const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(p.s1.ptr ? p.s1.ptr : p.s0.ptr, p.s1.ptr ? p.s1.bytes : 0u);
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.w, p.w_bytes);

  // ---- halo staging geometry (as in the row-staged kernel, NT threads)
  const int hv = tid & 3;
  const int hgrp = halo_group(tid);
  int h_full[NPV], h_half[NPV];
  const int Hh = p.Hi >> 1, Wh = p.Wi >> 1;
#pragma unroll
  for (int i = 0; i < NPV; ++i) {
    const int hp = hgrp + i * (NT / 4);
    const int hy = hp / HWI, hx = hp - hy * HWI;
    const int y = STR * y0 - 1 + hy, x = STR * x0 - 1 + hx;
    const bool ok = hp < HPIX && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
    h_full[i] = ok ? (n * p.Hi + y) * p.Wi + x : -1;
    h_half[i] = ok ? (n * Hh + (y >> 1)) * Wh + (x >> 1) : -1;
  }
  u32x4_t areg[NPV];
  float sc[VE], sh[VE];
  bool aff = false, relu = false;

  auto load_halo = [&](int cc) {
    const int c = cc * CK;
    const bool first = c < p.s0.C;
    const HaloSrc& sd = first ? p.s0 : p.s1;
    const int cl = (first ? c : c - p.s0.C) + hv * VE;
    aff = sd.scale != nullptr;
    relu = sd.relu != 0;
    if (aff) {
#pragma unroll
      for (int j = 0; j < VE; j += 4) {
        const f32x4_t s4 = *reinterpret_cast<const f32x4_t*>(sd.scale + cl + j);
        const f32x4_t h4 = *reinterpret_cast<const f32x4_t*>(sd.shift + cl + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[j + e] = s4[e]; sh[j + e] = h4[e]; }
      }
    }
    const bool up = STR == 1 && sd.up != 0;
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int pix = up ? h_half[i] : h_full[i];
      const uint32_t off = (uint32_t)(pix * sd.C + cl) * (uint32_t)EB;
      if (first) areg[i] = buf_load16(rs0, pix >= 0 ? off : kOOB);
      else areg[i] = buf_load16(rs1, pix >= 0 ? off : kOOB);
    }
  };
  auto store_halo = [&](char* A) {
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int hp = hgrp + i * (NT / 4);
      if (hp >= HPIX) continue;
      u32x4_t v = areg[i];
      if (aff) {
        v = AffineRelu<T>::run(v, sc, sh, relu);
        if (h_full[i] < 0) v = u32x4_t{0, 0, 0, 0};     // zero padding is applied after BN+ReLU
      }
      int slot = hp;
      if (STR == 2) {                                       // odd / even input columns in separate planes of the row
        const int hy = hp / HWI, hx = hp - hy * HWI;
        slot = hy * RS + (hx & 1) * 17 + (hx >> 1);
      }
      *reinterpret_cast<u32x4_t*>(A + slot * APS + hv * 16) = v;
    }
  };

  // weights of stage (chunk cc, filter column s) -> weight buffer `buf`: slot r = tap (r, s); data-gradient mode takes tap 8 - t
  const uint32_t slab_bytes = (uint32_t)p.K * 64u;         // one (chunk, tap) slab of the halo pack
  const uint32_t lane16 = (uint32_t)lane * 16u;
  auto dma_b = [&](int cc, int s, int buf) {
#pragma unroll
    for (int i = 0; i < (NPIECE + NW - 1) / NW; ++i) {
      const int pc = wave + i * NW;                        // wave-uniform
      if (NPIECE % NW == 0 || pc < NPIECE) {
        const int r = pc / (BN / 16), sub = pc - r * (BN / 16);
        const int tap = p.flip ? 8 - (3 * r + s) : 3 * r + s;
        const uint32_t goff = (uint32_t)(cc * 9 + tap) * slab_bytes + (uint32_t)(n0 + sub * 16) * 64u;
        char* dst = Bbuf + buf * Cfg::B_BYTES + (r * BN + sub * 16) * 64;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void*)dst, 16, lane16, goff, 0, 0);
      }
    }
  };

  // ---- accumulators and per-lane fragment bases
  const int wrow0 = (wave / WGN) * TP;
  const int wch0 = (wave % WGN) * (TC * 16);
  f32x4_t acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int li = lane & 15, kg = lane >> 4;
  const int a_lane = (STR * wrow0 * RS + li) * APS + kg * 16;
  const int b_lane = (wch0 + li) * 64 + ((kg ^ swz(li)) << 4);

  auto compute = [&](const char* A, const char* B) {
    u32x4_t X[NX], W[3][TC];
#pragma unroll
    for (int h = 0; h < NX; ++h) X[h] = *reinterpret_cast<const u32x4_t*>(A + h * (RS * APS));
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int a = 0; a < TC; ++a) W[r][a] = *reinterpret_cast<const u32x4_t*>(B + (r * BN + a * 16) * 64);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[a][b] = Mma<T>::run(W[r][a], X[STR * b + r], acc[a][b]);
    // issue order: [halo rows + filter row 0][filter row 1][MFMAs row 0][filter row 2][MFMAs row 1][MFMAs row 2]
    constexpr int NM = TC * TP * (sizeof(T) == 4 ? 4 : 1);
    __builtin_amdgcn_sched_group_barrier(0x100, NX + TC, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, TC, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, TC, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
  };

#ifdef VK_STAMP
  unsigned long long t_begin, t_pro, tk_load = 0, tk_comp = 0, tk_store = 0, tk_bar = 0;
  VK_T(t_begin)
#endif
  // ---- prologue: first chunk's halo + stage 0 weights (split-K: this workgroup's slice of the chunks)
  const int ksp = p.ksplit > 1 ? p.ksplit : 1;
  const int c_begin = (int)((long)p.nchunks * blockIdx.z / ksp), c_end = (int)((long)p.nchunks * (blockIdx.z + 1) / ksp);
  load_halo(c_begin);
  dma_b(c_begin, 0, 0);
  const bool one_chunk = Cfg::NB == 3 && c_end - c_begin == 1 && !p.dbg;
  if (one_chunk) {                                        // all three filter columns now: the stages need no barrier between them
    dma_b(c_begin, 1, 1);
    dma_b(c_begin, 2, 2);
  }
  store_halo(Abuf);
  __syncthreads();                                        // also drains the LDS-DMA (vmcnt(0))
  VK_T(t_pro)

  int st = 0;
  if (one_chunk) {
    const char* A = Abuf + a_lane;
#pragma unroll
    for (int s = 0; s < 3; ++s) compute(A + (STR == 1 ? s : (s & 1) * 17 + (s >> 1)) * APS, Bbuf + s * Cfg::B_BYTES + b_lane);
    __syncthreads();                                      // the epilogue reuses the operand buffers
    st = 3;
  }
  for (int cc = c_begin; cc < c_end && !one_chunk; ++cc) {
    const bool next_chunk = cc + 1 < c_end;
    const char* A = Abuf + (ADB ? ((cc - c_begin) & 1) * Cfg::A_BYTES : 0) + a_lane;
    char* const Anext = Abuf + (ADB ? ((cc - c_begin + 1) & 1) * Cfg::A_BYTES : 0);
#pragma unroll
    for (int s = 0; s < 3; ++s, ++st) {
#ifdef VK_STAMP
      unsigned long long t0, t1, t2, t3, t4;
#endif
      VK_T(t0)
      // next stage's weights into the other buffer (all waves passed the barrier that closed its last readers)
      if (!(p.dbg & 2)) {
        if (s < 2) dma_b(cc, s + 1, (st + 1) & 1);
        else if (next_chunk) dma_b(cc + 1, 0, (st + 1) & 1);
      }
      if (s == 0 && next_chunk && !(p.dbg & 4)) load_halo(cc + 1);
      VK_T(t1)
      compute(A + (STR == 1 ? s : (s & 1) * 17 + (s >> 1)) * APS, Bbuf + (st & 1) * Cfg::B_BYTES + b_lane);
      VK_T(t2)
      if (ADB) {
        if (s == 1 && next_chunk && !(p.dbg & 4)) store_halo(Anext);
      } else {
        if (s == 2 && next_chunk && !(p.dbg & 4)) {
          __syncthreads();                                // every wave is done reading this chunk's halo
          store_halo(Anext);
        }
      }
      VK_T(t3)
      if (!(p.dbg & 8)) __syncthreads();
      VK_T(t4)
#ifdef VK_STAMP
      tk_load += t1 - t0; tk_comp += t2 - t1; tk_store += t3 - t2; tk_bar += t4 - t3;
#endif
    }
  }
#ifdef VK_STAMP
  unsigned long long te0, te1;
  VK_T(te0)
#endif
  if (p.ksplit > 1) {
    // partial tile straight from the accumulators: lane = pixel li, four consecutive channels per 16-byte store
    float* sl = p.slab + (size_t)blockIdx.z * ((size_t)p.N * p.H * p.W * p.K);
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) {
        const int y = y0 + wrow0 + b, x = x0 + li, ch = n0 + wch0 + a * 16 + kg * 4;
        if (y < p.H && x < p.W && ch < p.K) *reinterpret_cast<f32x4_t*>(sl + (((size_t)n * p.H + y) * p.W + x) * p.K + ch) = acc[a][b];
      }
    return;
  }
  halo_epilogue<T, TH, BN, TP, TC, NT, (BN <= 64 && TH == 16)>(smem, acc, p, n, y0, x0, n0, wrow0, wch0);
#ifdef VK_STAMP
  VK_T(te1)
  if (p.stamps && lane == 0) {
    unsigned long long* o = p.stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8;
    o[0] = tk_load; o[1] = tk_comp; o[2] = tk_store; o[3] = tk_bar; o[4] = te1 - te0; o[5] = t_pro - t_begin; o[6] = te1 - t_begin; o[7] = st;
  }
#endif
}

// ---- data gradient of the stride-2 3x3 convolutions (the openers of layers 2-4): dx[y][x] = sum over the taps (r, s) with
// y + 1 - r and x + 1 - s even of  W[.][r][s]^T dz[(y + 1 - r) / 2][(x + 1 - s) / 2].  The parity class (y & 1, x & 1) of an output
// pixel selects its taps — (0,0): the centre tap; (0,1), (1,0): two; (1,1): the four corners — so a workgroup stages ONE
// (TH + 1) x 17 tile of dz per channel chunk and feeds all nine taps from it like the stride-1 kernel, into FOUR accumulator sets
// (one per class: the 2 TH x 32 output pixels above the tile); tap (r, s) goes to class (r != 1, s != 1) and reads dz at offset
// (r == 0, s == 0).  No zero-filled operand rows and one staging pass instead of the tap-by-tap gather of the v0 kernel.
// Weights: the halo pack of the transposed filter ([chunk of K][forward tap][C][32]), taps not flipped.
template <typename T, int TH, int BN, int WGM, int WGN>
struct S2dCfg {
  using Tr = ElemTraits<T>;
  static constexpr int EB = Tr::kBytes, VE = Tr::kVec, CK = 64 / EB;
  static constexpr int NW = WGM * WGN, NT = 64 * NW;
  static constexpr int HH = TH + 1, RS = 17, HPIX = HH * RS, APS = 96;
  static constexpr int A_BYTES = HPIX * APS;
  static constexpr int B_BYTES = 3 * BN * 64;
  static constexpr int NPV = (HPIX * 4 + NT - 1) / NT;
  static constexpr int NPIECE = 3 * BN / 16;
  static constexpr int BM = TH * 16;
  static constexpr int TP = TH / WGM, TC = BN / WGN / 16;
  static constexpr int ESB = BN * EB + 16;
  static constexpr int MAIN = 2 * A_BYTES + 2 * B_BYTES;
  static constexpr int ESLOTS = (BN / VE > 16) ? NW * 4 / (BN / VE / 16) : NW * 4;
  static constexpr int EPI = BM * ESB + ESLOTS * BN * 2 * 4;
  static constexpr int SMEM = MAIN > EPI ? MAIN : EPI;
  static_assert(TH % WGM == 0 && BN % (16 * WGN) == 0 && NW == 8, "wave layout");
  static_assert(4 * TP * TC <= 32, "accumulator budget (four parity classes)");
  static_assert(SMEM <= 160 * 1024, "LDS image exceeds the 160 KiB of a CU");
};

template <typename T, int TH, int BN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN, 1) void conv3x3_s2dg_kernel(const HaloParams p) {
  using Cfg = S2dCfg<T, TH, BN, WGM, WGN>;
  constexpr int EB = Cfg::EB, VE = Cfg::VE, CK = Cfg::CK, HPIX = Cfg::HPIX, NPV = Cfg::NPV, NT = Cfg::NT, NW = Cfg::NW;
  constexpr int TP = Cfg::TP, TC = Cfg::TC, APS = Cfg::APS, NPIECE = Cfg::NPIECE, RS = Cfg::RS;
  typedef __attribute__((address_space(3))) void lds_void;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Abuf = smem;
  char* const Bbuf = smem + 2 * Cfg::A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int bt = blockIdx.x;
  const int ky = blockIdx.y;
  if ((gridDim.x & 7) == 0) bt = (bt & 7) * (int)(gridDim.x >> 3) + (bt >> 3);        // contiguous runs of tiles per XCD
  const int tx = bt % p.tiles_x;
  bt /= p.tiles_x;
  const int ty = bt % p.tiles_y;
  const int n = bt / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * 16;                    // tile origin in dz
  const int n0 = ky * BN;

  const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(p.s0.ptr, p.s0.bytes);
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.w, p.w_bytes);

  const int hv = tid & 3;
  const int hgrp = halo_group(tid);
  int h_full[NPV];
#pragma unroll
  for (int i = 0; i < NPV; ++i) {
    const int hp = hgrp + i * (NT / 4);
    const int hy = hp / RS, hx = hp - hy * RS;
    const int y = y0 + hy, x = x0 + hx;
    const bool ok = hp < HPIX && y < p.Hi && x < p.Wi;
    h_full[i] = ok ? (n * p.Hi + y) * p.Wi + x : -1;
  }
  u32x4_t areg[NPV];
  auto load_halo = [&](int cc) {
    const int cl = cc * CK + hv * VE;
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const uint32_t off = (uint32_t)(h_full[i] * p.s0.C + cl) * (uint32_t)EB;
      areg[i] = buf_load16(rs0, h_full[i] >= 0 ? off : kOOB);
    }
  };
  auto store_halo = [&](char* A) {
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int hp = hgrp + i * (NT / 4);
      if (hp < HPIX) *reinterpret_cast<u32x4_t*>(A + hp * APS + hv * 16) = areg[i];
    }
  };
  const uint32_t slab_bytes = (uint32_t)p.K * 64u;
  const uint32_t lane16 = (uint32_t)lane * 16u;
  auto dma_b = [&](int cc, int s, int buf) {
#pragma unroll
    for (int i = 0; i < (NPIECE + NW - 1) / NW; ++i) {
      const int pc = wave + i * NW;
      if (NPIECE % NW == 0 || pc < NPIECE) {
        const int r = pc / (BN / 16), sub = pc - r * (BN / 16);
        const uint32_t goff = (uint32_t)(cc * 9 + 3 * r + s) * slab_bytes + (uint32_t)(n0 + sub * 16) * 64u;
        char* dst = Bbuf + buf * Cfg::B_BYTES + (r * BN + sub * 16) * 64;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void*)dst, 16, lane16, goff, 0, 0);
      }
    }
  };

  const int wrow0 = (wave / WGN) * TP;
  const int wch0 = (wave % WGN) * (TC * 16);
  f32x4_t acc[4][TC][TP];                                  // [class (y & 1) * 2 + (x & 1)]
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) acc[q][a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int li = lane & 15, kg = lane >> 4;
  const int a_lane = (wrow0 * RS + li) * APS + kg * 16;
  const int b_lane = (wch0 + li) * 64 + ((kg ^ swz(li)) << 4);

  auto compute = [&](const char* A, const char* B, auto s_c) {
    constexpr int s = decltype(s_c)::value;
    u32x4_t X[TP + 1], W[3][TC];
#pragma unroll
    for (int h = 0; h < TP + 1; ++h) X[h] = *reinterpret_cast<const u32x4_t*>(A + ((s == 0 ? 1 : 0) + h * RS) * APS);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int a = 0; a < TC; ++a) W[r][a] = *reinterpret_cast<const u32x4_t*>(B + (r * BN + a * 16) * 64);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      constexpr int px = s != 1 ? 1 : 0;
      const int q = (r != 1 ? 2 : 0) + px, di = r == 0 ? 1 : 0;
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[q][a][b] = Mma<T>::run(W[r][a], X[b + di], acc[q][a][b]);
    }
  };

  load_halo(0);
  dma_b(0, 0, 0);
  store_halo(Abuf);
  __syncthreads();
  int st = 0;
  for (int cc = 0; cc < p.nchunks; ++cc) {
    const bool next_chunk = cc + 1 < p.nchunks;
    const char* A = Abuf + (cc & 1) * Cfg::A_BYTES + a_lane;
    char* const Anext = Abuf + ((cc + 1) & 1) * Cfg::A_BYTES;
    // stage 0
    dma_b(cc, 1, (st + 1) & 1);
    if (next_chunk) load_halo(cc + 1);
    compute(A, Bbuf + (st & 1) * Cfg::B_BYTES + b_lane, std::integral_constant<int, 0>{});
    __syncthreads();
    ++st;
    // stage 1
    dma_b(cc, 2, (st + 1) & 1);
    compute(A, Bbuf + (st & 1) * Cfg::B_BYTES + b_lane, std::integral_constant<int, 1>{});
    if (next_chunk) store_halo(Anext);
    __syncthreads();
    ++st;
    // stage 2
    if (next_chunk) dma_b(cc + 1, 0, (st + 1) & 1);
    compute(A, Bbuf + (st & 1) * Cfg::B_BYTES + b_lane, std::integral_constant<int, 2>{});
    __syncthreads();
    ++st;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q) __syncthreads();                                // the previous class's tile has left LDS
    halo_epilogue<T, TH, BN, TP, TC, NT, false, 2>(smem, acc[q], p, n, y0, x0, n0, wrow0, wch0, q >> 1, q & 1);
  }
}

// ---- software-pipelined form of the column-staged kernel for the K >= 128 classes (layers 2-4, decoder blocks 0-1: the dominant
// kernel of the training step).  With 111 KiB of LDS only ONE workgroup of conv3x3_col_kernel fits a CU, its eight waves run in
// step, and after every stage barrier all of them need their first operand fragments at the same moment: the 14 fragment reads of
// every wave queue up behind each other on the 128 B/clk LDS port while no MFMA is in flight (in-kernel stamps, r02: 2,600 cycles
// per stage against 1,536 cycles of MFMA work per SIMD).  Here the stage barrier sits INSIDE the MFMA sequence:
//     top of stage st : fragments of filter rows 0 and 1 + the halo rows are already in registers (read during stage st - 1)
//                       read filter row 2  |  MFMAs of filter rows 0, 1
//     s_waitcnt vmcnt(DMAW) lgkmcnt(0); s_barrier      -> weights of stage st + 1 have landed, nobody reads buffer st any more
//                       LDS-DMA of stage st + 3 into the buffer of stage st  |  fragment reads of stage st + 1  |  MFMAs of filter row 2
// so the reads of the next stage travel under 16 MFMAs per wave and the MFMA pipe only sees the barrier's own latency.  That
// needs three weight buffers (a stage's weights are requested three stages ahead), a partial vmcnt wait (only the DMAs issued
// during the previous stage may still be in flight — they are the youngest vector-memory operations of every wave by
// construction, the halo loads are issued in front of them) and two fragment sets in registers: 256 registers per wave instead
// of 128 (one workgroup per CU either way).  Same operand order, same accumulation order per accumulator (filter row 0, 1, 2 of
// every stage), same epilogue: bit-identical to conv3x3_col_kernel.
template <typename T, int TH, int BN, int WGM, int WGN>
struct ColqCfg : ColCfg<T, TH, BN, WGM, WGN, true> {
  using Base = ColCfg<T, TH, BN, WGM, WGN, true>;
  static constexpr int MAIN = 2 * Base::A_BYTES + 3 * Base::B_BYTES;
  static constexpr int SMEM = MAIN > Base::EPI ? MAIN : Base::EPI;
  static constexpr int DMAW = Base::NPIECE / Base::NW;           // LDS-DMA instructions per wave and stage
  static_assert(Base::NPIECE % Base::NW == 0, "every wave must issue the same number of weight DMAs per stage (partial vmcnt wait)");
  static_assert(DMAW >= 1 && DMAW <= 15, "vmcnt immediate");
  static_assert(SMEM <= 160 * 1024, "LDS image exceeds the 160 KiB of a CU");
};

template <typename T, int TH, int BN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN, 1) void conv3x3_colq_kernel(const HaloParams p) {
  using Cfg = ColqCfg<T, TH, BN, WGM, WGN>;
  constexpr int EB = Cfg::EB, VE = Cfg::VE, CK = Cfg::CK, HPIX = Cfg::HPIX, NPV = Cfg::NPV, NT = Cfg::NT, NW = Cfg::NW;
  constexpr int TP = Cfg::TP, TC = Cfg::TC, APS = Cfg::APS, NPIECE = Cfg::NPIECE;
  typedef __attribute__((address_space(3))) void lds_void;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Abuf = smem;
  char* const Bbuf = smem + 2 * Cfg::A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int bt = blockIdx.x;
  int ky = blockIdx.y;
  if (p.kyn > 0) {                                           // see conv3x3_col_kernel
    const int j = bt >> 3;
    ky = j % p.kyn;
    bt = (bt & 7) * (int)((gridDim.x / p.kyn) >> 3) + j / p.kyn;
  } else if (p.kyn < 0) {
    const int kn = -p.kyn, xcd = bt & 7;
    ky = xcd % kn;
    bt = (xcd / kn) * (int)(gridDim.x >> 3) + (bt >> 3);
  } else if ((gridDim.x & 7) == 0) {
    bt = (bt & 7) * (int)(gridDim.x >> 3) + (bt >> 3);
  }
  const int tx = bt % p.tiles_x;
  bt /= p.tiles_x;
  const int ty = bt % p.tiles_y;
  const int n = bt / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * 16;
  const int n0 = ky * BN;

  const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(p.s0.ptr, p.s0.bytes);
  const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(p.s1.ptr ? p.s1.ptr : p.s0.ptr, p.s1.ptr ? p.s1.bytes : 0u);
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.w, p.w_bytes);

  const int hv = tid & 3;
  const int hgrp = halo_group(tid);
  int h_full[NPV], h_half[NPV];
  const int Hh = p.H >> 1, Wh = p.W >> 1;
#pragma unroll
  for (int i = 0; i < NPV; ++i) {
    const int hp = hgrp + i * (NT / 4);
    const int hy = hp / 18, hx = hp - hy * 18;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = hp < HPIX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    h_full[i] = ok ? (n * p.H + y) * p.W + x : -1;
    h_half[i] = ok ? (n * Hh + (y >> 1)) * Wh + (x >> 1) : -1;
  }
  u32x4_t areg[NPV];
  float sc[VE], sh[VE];
  bool aff = false, relu = false;

  auto load_halo = [&](int cc) {
    const int c = cc * CK;
    const bool first = c < p.s0.C;
    const HaloSrc& sd = first ? p.s0 : p.s1;
    const int cl = (first ? c : c - p.s0.C) + hv * VE;
    aff = sd.scale != nullptr;
    relu = sd.relu != 0;
    if (aff) {
#pragma unroll
      for (int j = 0; j < VE; j += 4) {
        const f32x4_t s4 = *reinterpret_cast<const f32x4_t*>(sd.scale + cl + j);
        const f32x4_t h4 = *reinterpret_cast<const f32x4_t*>(sd.shift + cl + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[j + e] = s4[e]; sh[j + e] = h4[e]; }
      }
    }
    const bool up = sd.up != 0;
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int pix = up ? h_half[i] : h_full[i];
      const uint32_t off = (uint32_t)(pix * sd.C + cl) * (uint32_t)EB;
      if (first) areg[i] = buf_load16(rs0, pix >= 0 ? off : kOOB);
      else areg[i] = buf_load16(rs1, pix >= 0 ? off : kOOB);
    }
  };
  auto store_halo = [&](char* A) {
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int hp = hgrp + i * (NT / 4);
      if (hp >= HPIX) continue;
      u32x4_t v = areg[i];
      if (aff) {
        v = (p.dbg & 32) ? AffineReluScalar<T>::run(v, sc, sh, relu) : AffineRelu<T>::run(v, sc, sh, relu);
        if (h_full[i] < 0) v = u32x4_t{0, 0, 0, 0};
      }
      *reinterpret_cast<u32x4_t*>(A + hp * APS + hv * 16) = v;
    }
  };

  // weights of stage (chunk cc, filter column s) -> weight buffer s
  const uint32_t slab_bytes = (uint32_t)p.K * 64u;
  const uint32_t lane16 = (uint32_t)lane * 16u;
  auto dma_b = [&](int cc, int s, uint32_t oob = 0u) {       // oob = kOOB: issued, but out of the descriptor's range (fetches nothing; keeps the vmcnt pattern uniform)
#pragma unroll
    for (int i = 0; i < ((VK_COLQ_DIAG & 1) ? 1 : Cfg::DMAW); ++i) {      // diagnostic bit 1: one piece per wave and stage (timing only)
      const int pc = wave + i * NW;
      const int r = pc / (BN / 16), sub = pc - r * (BN / 16);
      const int tap = p.flip ? 8 - (3 * r + s) : 3 * r + s;
      const uint32_t goff = (uint32_t)(cc * 9 + tap) * slab_bytes + (uint32_t)(n0 + sub * 16) * 64u;
      char* dst = Bbuf + s * Cfg::B_BYTES + (r * BN + sub * 16) * 64;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void*)dst, 16, lane16 | oob, goff, 0, 0);     // (the VECTOR offset is the range-checked one)
    }
  };


  auto dma_one = [&](int cc, int s, int i, uint32_t oob) {
    const int pc = wave + i * NW;
    const int r = pc / (BN / 16), sub = pc - r * (BN / 16);
    const int tap = p.flip ? 8 - (3 * r + s) : 3 * r + s;
    const uint32_t goff = (uint32_t)(cc * 9 + tap) * slab_bytes + (uint32_t)(n0 + sub * 16) * 64u;
    char* dst = Bbuf + s * Cfg::B_BYTES + (r * BN + sub * 16) * 64;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void*)dst, 16, lane16 | oob, goff, 0, 0);
  };

  const int wrow0 = (wave / WGN) * TP;
  const int wch0 = (wave % WGN) * (TC * 16);
  f32x4_t acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int li = lane & 15, kg = lane >> 4;
  const int a_lane = (wrow0 * 18 + li) * APS + kg * 16;
  const int b_lane = (wch0 + li) * 64 + ((kg ^ swz(li)) << 4);

  // fragment sets: X = halo rows (rows 0..TP-1 arrive during the previous stage, rows TP and TP+1 during this one), Wr = filter rows
  u32x4_t X[TP + 2], W0[TC], W1[TC], W2[TC];
  auto rd = [](const char* q) { return *reinterpret_cast<const u32x4_t*>(q); };
  auto mfma_row = [&](const u32x4_t (&Wr)[TC], int r) {
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) acc[a][b] = Mma<T>::run(Wr[a], X[b + r], acc[a][b]);
  };
  constexpr int NM = TC * TP * (sizeof(T) == 4 ? 4 : 1);       // MFMA instructions per filter row
  constexpr int kWaitPart = (Cfg::DMAW & 0xF) | 0x70;          // s_waitcnt vmcnt(DMAW) lgkmcnt(0)  (expcnt untouched)
  constexpr int kWaitAll = 0x0070;                              // s_waitcnt vmcnt(0) lgkmcnt(0)
  static_assert(TP % 2 == 0, "the next stage's halo rows are fetched in two halves");

  // ---- prologue: first chunk's halo + the weights of its three stages
  const int ksp = p.ksplit > 1 ? p.ksplit : 1;
  const int c_begin = (int)((long)p.nchunks * blockIdx.z / ksp), c_end = (int)((long)p.nchunks * (blockIdx.z + 1) / ksp);
  load_halo(c_begin);
  dma_b(c_begin, 0);
  dma_b(c_begin, 1);
  dma_b(c_begin, 2);
  store_halo(Abuf);
  __syncthreads();                                        // also drains the LDS-DMA (vmcnt(0))
#pragma unroll
  for (int h = 0; h < TP; ++h) X[h] = rd(Abuf + a_lane + h * (18 * APS));
#pragma unroll
  for (int a = 0; a < TC; ++a) W0[a] = rd(Bbuf + b_lane + (a * 16) * 64);

  for (int cc = c_begin; cc < c_end; ++cc) {
    const bool next_chunk = cc + 1 < c_end;
    const char* const Acur = Abuf + ((cc - c_begin) & 1) * Cfg::A_BYTES + a_lane;
    char* const Anext = Abuf + ((cc - c_begin + 1) & 1) * Cfg::A_BYTES;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const char* const As = Acur + s * APS;
      const char* const Bs = Bbuf + s * Cfg::B_BYTES + b_lane;
      // the next stage's halo rows: same image one column on, or the next chunk's image (published by the barrier of stage 1)
      const bool more = s < 2 || next_chunk;
      const char* const An = s < 2 ? Acur + (s + 1) * APS : Anext + a_lane;
      const char* const Bn = Bbuf + ((s + 1) % 3) * Cfg::B_BYTES + b_lane;
      u32x4_t Xn[TP], W0n[TC];
      // -- filter row 0 (operands in registers); under it: halo row TP + filter row 1, first half of the next stage's rows
      X[TP] = rd(As + TP * (18 * APS));
#pragma unroll
      for (int a = 0; a < TC; ++a) W1[a] = rd(Bs + (BN + a * 16) * 64);
      if (more) {
#pragma unroll
        for (int h = 0; h < TP / 2; ++h) Xn[h] = rd(An + h * (18 * APS));
      }
      mfma_row(W0, 0);
      // -- filter row 1; under it: halo row TP + 1 + filter row 2, second half of the next stage's rows
      X[TP + 1] = rd(As + (TP + 1) * (18 * APS));
#pragma unroll
      for (int a = 0; a < TC; ++a) W2[a] = rd(Bs + (2 * BN + a * 16) * 64);
      if (more) {
#pragma unroll
        for (int h = TP / 2; h < TP; ++h) Xn[h] = rd(An + h * (18 * APS));
      }
      mfma_row(W1, 1);
      __builtin_amdgcn_sched_group_barrier(0x100, 1 + TC + TP / 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1 + TC + TP / 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
      if (s == 1 && next_chunk && !(VK_COLQ_DIAG & 2)) store_halo(Anext);          // diagnostic bit 2: no operand transform / halo store in the loop
      // every wave: its reads of this stage's weights / halo rows have returned, its halo stores are in LDS, and of its vector
      // memory operations only the DMAW youngest (the weights of stage st + 2) may still be in flight
      asm volatile("" ::: "memory");
      if (VK_COLQ_ILV) __builtin_amdgcn_s_waitcnt(kWaitPart);     // the last chunk issues its (out-of-range) DMAs too
      else if (next_chunk) __builtin_amdgcn_s_waitcnt((VK_COLQ_DIAG & 1) ? ((1 & 0xF) | 0x70) : kWaitPart);
      else __builtin_amdgcn_s_waitcnt(kWaitAll);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (VK_COLQ_ILV) {
        // r04: the weight DMAs of the stage are issued BETWEEN the MFMAs of filter row 2, not in front of them: all eight
        // waves leave the barrier together and queue 24 vector-memory instructions at the CU's address unit while no MFMA is in
        // flight (knock-out builds: 16 of them cost 12-15 % of the kernel).  Needs one basic block: the DMAs are issued in the last
        // chunk as well, out of the descriptor's range
        if (s == 0 && next_chunk && !(VK_COLQ_DIAG & 4)) load_halo(cc + 1);
#pragma unroll
        for (int a = 0; a < TC; ++a) W0n[a] = rd(Bn + (a * 16) * 64);
        const uint32_t oob = next_chunk ? 0u : kOOB;
        constexpr int NT2 = TC * TP, GRP = NT2 / (Cfg::DMAW + 1);
#pragma unroll
        for (int i = 0; i <= Cfg::DMAW; ++i) {
          const int lo = i * GRP, hi = i == Cfg::DMAW ? NT2 : (i + 1) * GRP;
#pragma unroll
          for (int q = lo; q < hi; ++q) acc[q / TP][q % TP] = Mma<T>::run(W2[q / TP], X[q % TP + 2], acc[q / TP][q % TP]);
          if (i < Cfg::DMAW) {
            __builtin_amdgcn_sched_barrier(0);
            dma_one(cc + 1, s, i, oob);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
      if (next_chunk) {
        if (s == 0 && !(VK_COLQ_DIAG & 4)) load_halo(cc + 1);   // in front of the DMAs: those stay the youngest operations (diagnostic bit 4: no halo loads)
        dma_b(cc + 1, s);                                   // stage st + 3 -> the buffer every wave has just finished reading
      }
      // -- filter row 2; under it: the next stage's filter row 0 (its weights became visible with the barrier)
      if (more) {
#pragma unroll
        for (int a = 0; a < TC; ++a) W0n[a] = rd(Bn + (a * 16) * 64);
      }
      mfma_row(W2, 2);
      }
      if (more) {
#pragma unroll
        for (int h = 0; h < TP; ++h) X[h] = Xn[h];
#pragma unroll
        for (int a = 0; a < TC; ++a) W0[a] = W0n[a];
      }
    }
  }
  if (VK_COLQ_ILV) __builtin_amdgcn_s_waitcnt(kWaitAll);    // the last chunk's out-of-range DMAs have retired
  __syncthreads();                                          // every wave is done with the operand buffers: the epilogue reuses them

  if (p.ksplit > 1) {
    float* sl = p.slab + (size_t)blockIdx.z * ((size_t)p.N * p.H * p.W * p.K);
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) {
        const int y = y0 + wrow0 + b, x = x0 + li, ch = n0 + wch0 + a * 16 + kg * 4;
        if (y < p.H && x < p.W && ch < p.K) *reinterpret_cast<f32x4_t*>(sl + (((size_t)n * p.H + y) * p.W + x) * p.K + ch) = acc[a][b];
      }
    return;
  }
  halo_epilogue<T, TH, BN, TP, TC, NT, ((VK_COLQ_PRE && sizeof(T) == 2) || (BN <= 64 && TH == 16))>(smem, acc, p, n, y0, x0, n0, wrow0, wch0);
}

// ---- staggered form of the pipelined kernel (r03): the same tile, operand order and epilogue (bit-identical results), but the two
// waves that share a SIMD run HALF A STAGE apart — see the comment in front of the main loop.  Four weight stage buffers.
template <typename T, int TH, int BN, int WGM, int WGN>
struct ColsCfg : ColCfg<T, TH, BN, WGM, WGN, true> {
  using Base = ColCfg<T, TH, BN, WGM, WGN, true>;
  static constexpr int MAIN = 2 * Base::A_BYTES + 4 * Base::B_BYTES;      // four weight stage buffers (see the kernel)
  static constexpr int SMEM = MAIN > Base::EPI ? MAIN : Base::EPI;
  static constexpr int DMAW = Base::NPIECE / Base::NW;           // LDS-DMA instructions per wave and stage
  static_assert(Base::NPIECE % Base::NW == 0, "every wave must issue the same number of weight DMAs per stage (partial vmcnt wait)");
  static_assert(DMAW >= 1 && DMAW <= 15, "vmcnt immediate");
  static_assert(SMEM <= 160 * 1024, "LDS image exceeds the 160 KiB of a CU");
};

template <typename T, int TH, int BN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN, 1) void conv3x3_cols_kernel(const HaloParams p) {
  using Cfg = ColsCfg<T, TH, BN, WGM, WGN>;
  constexpr int EB = Cfg::EB, VE = Cfg::VE, CK = Cfg::CK, HPIX = Cfg::HPIX, NPV = Cfg::NPV, NT = Cfg::NT, NW = Cfg::NW;
  constexpr int TP = Cfg::TP, TC = Cfg::TC, APS = Cfg::APS, NPIECE = Cfg::NPIECE;
  typedef __attribute__((address_space(3))) void lds_void;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Abuf = smem;
  char* const Bbuf = smem + 2 * Cfg::A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int bt = blockIdx.x;
  int ky = blockIdx.y;
  if (p.kyn > 0) {                                           // see conv3x3_col_kernel
    const int j = bt >> 3;
    ky = j % p.kyn;
    bt = (bt & 7) * (int)((gridDim.x / p.kyn) >> 3) + j / p.kyn;
  } else if (p.kyn < 0) {
    const int kn = -p.kyn, xcd = bt & 7;
    ky = xcd % kn;
    bt = (xcd / kn) * (int)(gridDim.x >> 3) + (bt >> 3);
  } else if ((gridDim.x & 7) == 0) {
    bt = (bt & 7) * (int)(gridDim.x >> 3) + (bt >> 3);
  }
  const int tx = bt % p.tiles_x;
  bt /= p.tiles_x;
  const int ty = bt % p.tiles_y;
  const int n = bt / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * 16;
  const int n0 = ky * BN;

  const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(p.s0.ptr, p.s0.bytes);
  const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(p.s1.ptr ? p.s1.ptr : p.s0.ptr, p.s1.ptr ? p.s1.bytes : 0u);
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.w, p.w_bytes);

  const int hv = tid & 3;
  const int hgrp = halo_group(tid);
  int h_full[NPV], h_half[NPV];
  const int Hh = p.H >> 1, Wh = p.W >> 1;
#pragma unroll
  for (int i = 0; i < NPV; ++i) {
    const int hp = hgrp + i * (NT / 4);
    const int hy = hp / 18, hx = hp - hy * 18;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = hp < HPIX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    h_full[i] = ok ? (n * p.H + y) * p.W + x : -1;
    h_half[i] = ok ? (n * Hh + (y >> 1)) * Wh + (x >> 1) : -1;
  }
  u32x4_t areg[NPV];
  float sc[VE], sh[VE];
  bool aff = false, relu = false;

  auto load_halo = [&](int cc) {
    const int c = cc * CK;
    const bool first = c < p.s0.C;
    const HaloSrc& sd = first ? p.s0 : p.s1;
    const int cl = (first ? c : c - p.s0.C) + hv * VE;
    aff = sd.scale != nullptr;
    relu = sd.relu != 0;
    if (aff) {
#pragma unroll
      for (int j = 0; j < VE; j += 4) {
        const f32x4_t s4 = *reinterpret_cast<const f32x4_t*>(sd.scale + cl + j);
        const f32x4_t h4 = *reinterpret_cast<const f32x4_t*>(sd.shift + cl + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[j + e] = s4[e]; sh[j + e] = h4[e]; }
      }
    }
    const bool up = sd.up != 0;
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int pix = up ? h_half[i] : h_full[i];
      const uint32_t off = (uint32_t)(pix * sd.C + cl) * (uint32_t)EB;
      if (first) areg[i] = buf_load16(rs0, pix >= 0 ? off : kOOB);
      else areg[i] = buf_load16(rs1, pix >= 0 ? off : kOOB);
    }
  };
  auto store_halo = [&](char* A) {
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int hp = hgrp + i * (NT / 4);
      if (hp >= HPIX) continue;
      u32x4_t v = areg[i];
      if (aff) {
        v = AffineRelu<T>::run(v, sc, sh, relu);
        if (h_full[i] < 0) v = u32x4_t{0, 0, 0, 0};
      }
      *reinterpret_cast<u32x4_t*>(A + hp * APS + hv * 16) = v;
    }
  };

  // weights of stage (chunk cc, filter column s) -> weight buffer s
  const uint32_t slab_bytes = (uint32_t)p.K * 64u;
  const uint32_t lane16 = (uint32_t)lane * 16u;
  auto dma_b = [&](int cc, int s, int buf) {
#pragma unroll
    for (int i = 0; i < Cfg::DMAW; ++i) {
      const int pc = wave + i * NW;
      const int r = pc / (BN / 16), sub = pc - r * (BN / 16);
      const int tap = p.flip ? 8 - (3 * r + s) : 3 * r + s;
      const uint32_t goff = (uint32_t)(cc * 9 + tap) * slab_bytes + (uint32_t)(n0 + sub * 16) * 64u;
      char* dst = Bbuf + buf * Cfg::B_BYTES + (r * BN + sub * 16) * 64;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void*)dst, 16, lane16, goff, 0, 0);
    }
  };

  const int wrow0 = (wave / WGN) * TP;
  const int wch0 = (wave % WGN) * (TC * 16);
  f32x4_t acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int li = lane & 15, kg = lane >> 4;
  const int a_lane = (wrow0 * 18 + li) * APS + kg * 16;
  const int b_lane = (wch0 + li) * 64 + ((kg ^ swz(li)) << 4);

  // fragment sets: X = the TP + 2 halo rows of a stage, Wr = its three filter rows.  Rows 0..TP-1 and W0 of stage st + 1 are read in
  // the second half (slot Y) of stage st, everything else under the MFMAs of the stage itself.
  u32x4_t X[TP + 2], W0[TC], W1[TC], W2[TC];
  auto rd = [](const char* q) { return *reinterpret_cast<const u32x4_t*>(q); };
  auto mfma_row = [&](const u32x4_t (&Wr)[TC], int r) {
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) acc[a][b] = Mma<T>::run(Wr[a], X[b + r], acc[a][b]);
  };
  constexpr int NM = TC * TP * (sizeof(T) == 4 ? 4 : 1);       // MFMA instructions per filter row
  constexpr int kWaitAll = 0x0070;                              // s_waitcnt vmcnt(0) lgkmcnt(0)  (expcnt untouched)
  constexpr int kWaitLds = 0xC07F;                              // s_waitcnt lgkmcnt(0)

#ifdef VK_STAMP
  unsigned long long t_begin, t_pro, tkX = 0, tkW = 0, tkB1 = 0, tkY = 0, tkB2 = 0;
  VK_T(t_begin)
#endif
  // ---- prologue (all waves together): first chunk's halo + the weights of its three stages into buffers 0..2
  const int ksp = p.ksplit > 1 ? p.ksplit : 1;
  const int c_begin = (int)((long)p.nchunks * blockIdx.z / ksp), c_end = (int)((long)p.nchunks * (blockIdx.z + 1) / ksp);
  load_halo(c_begin);
  dma_b(c_begin, 0, 0);
  dma_b(c_begin, 1, 1);
  dma_b(c_begin, 2, 2);
  store_halo(Abuf);
  __syncthreads();                                        // also drains the LDS-DMA (vmcnt(0))
#pragma unroll
  for (int h = 0; h < TP; ++h) X[h] = rd(Abuf + a_lane + h * (18 * APS));
#pragma unroll
  for (int a = 0; a < TC; ++a) W0[a] = rd(Bbuf + b_lane + (a * 16) * 64);

  // ---- the stagger: every stage is two slots separated by workgroup barriers,
  //        X: the stage's 3 * TC * TP MFMAs (with the fragment reads of filter rows 1 and 2 under them), then s_waitcnt vmcnt(0)
  //        Y: everything that is not matrix work — next chunk's halo requests (s = 0) / BN+ReLU transform + LDS stores (s = 1), the
  //           LDS-DMA of the weights three stages ahead, the first fragments of the next stage
  //      and the second-dispatched half of the waves (4-7: each shares a SIMD with one of waves 0-3) enters the loop ONE BARRIER
  //      later, so that on every SIMD one wave is in X while its partner is in Y: the matrix pipe always has exactly one customer and
  //      the VALU / LDS-store / DMA-issue work of the partner sits beside those MFMAs instead of beside its own copy.
  //      Slot t: early waves run X(st) at t = 2 st, Y(st) at 2 st + 1; late waves X(st) at 2 st + 1, Y(st) at 2 st + 2.
  //   weights: stage st lives in buffer st % 4.  Its last reads are in X(st) (slots 2 st / 2 st + 1); the DMA of stage st + 4 into the
  //      same buffer is issued in Y(st + 1) (slots 2 st + 3 / 2 st + 4) — behind a barrier that every reader has passed.  The DMA of
  //      stage st + 3 issued in Y(st) is waited for (vmcnt(0)) by its issuing wave at the end of X(st + 1) (slots 2 st + 2 /
  //      2 st + 3) and first read in Y(st + 2) (slots 2 st + 5 / 2 st + 6): a barrier lies between every wait and every read.
  //   halo image: chunk cc + 1 is stored in Y(cc, 1) (slots 6 cc + 3 / 6 cc + 4) into the buffer chunk cc - 1 was last read from in
  //      X(cc - 1, 2) (slots 6 cc - 2 / 6 cc - 1), and first read in Y(cc, 2) (slots 6 cc + 5 / 6 cc + 6).
  VK_T(t_pro)
  const bool late = wave >= NW / 2;
  if (late) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(kWaitLds);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  for (int cc = c_begin; cc < c_end; ++cc) {
    const bool next_chunk = cc + 1 < c_end;
    const char* const Acur = Abuf + ((cc - c_begin) & 1) * Cfg::A_BYTES + a_lane;
    char* const Anext = Abuf + ((cc - c_begin + 1) & 1) * Cfg::A_BYTES;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int st = 3 * (cc - c_begin) + s;
      const char* const As = Acur + s * APS;
      const char* const Bs = Bbuf + (st & 3) * Cfg::B_BYTES + b_lane;
      const bool more = s < 2 || next_chunk;
      const char* const An = s < 2 ? Acur + (s + 1) * APS : Anext + a_lane;
      const char* const Bn = Bbuf + ((st + 1) & 3) * Cfg::B_BYTES + b_lane;
#ifdef VK_STAMP
      unsigned long long t0, t1, t2, t3, t4, t5;
#endif
      VK_T(t0)
      // ---- slot X: the ten fragment reads of filter rows 1 and 2 first (40 LDS cycles), then the MFMAs in filter-row order — row 0
      // runs on registers filled in the previous slot Y and covers the latency of those reads
      X[TP] = rd(As + TP * (18 * APS));
#pragma unroll
      for (int a = 0; a < TC; ++a) W1[a] = rd(Bs + (BN + a * 16) * 64);
      X[TP + 1] = rd(As + (TP + 1) * (18 * APS));
#pragma unroll
      for (int a = 0; a < TC; ++a) W2[a] = rd(Bs + (2 * BN + a * 16) * 64);
      mfma_row(W0, 0);
      mfma_row(W1, 1);
      mfma_row(W2, 2);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 + 2 * TC, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3 * NM, 0);
      VK_T(t1)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(kWaitAll);                 // this wave's halo requests / weight DMAs of its previous slot Y have landed
      VK_T(t2)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      VK_T(t3)
      // ---- slot Y: the first fragments of the next stage are requested FIRST (their latency runs under the halo / DMA work below);
      // p.dbg & 16 (timing experiment, results unchanged): this slot at raised priority, so that its few vector instructions are
      // not queued behind the partner wave's MFMA stream
      if (p.dbg & 16) __builtin_amdgcn_s_setprio(1);
      if (more) {
#pragma unroll
        for (int h = 0; h < TP; ++h) X[h] = rd(An + h * (18 * APS));
#pragma unroll
        for (int a = 0; a < TC; ++a) W0[a] = rd(Bn + (a * 16) * 64);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (next_chunk) {
        if (s == 0) load_halo(cc + 1);                      // in front of the DMAs (vmcnt retires in order)
        if (s == 1) store_halo(Anext);
        dma_b(cc + 1, s, (st + 3) & 3);
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(kWaitLds);                 // fragment reads returned, halo stores in LDS
      if (p.dbg & 16) __builtin_amdgcn_s_setprio(0);
      VK_T(t4)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      VK_T(t5)
#ifdef VK_STAMP
      tkX += t1 - t0; tkW += t2 - t1; tkB1 += t3 - t2; tkY += t4 - t3; tkB2 += t5 - t4;
#endif
    }
  }
  if (!late) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                           // pairs with the late half's last slot
    asm volatile("" ::: "memory");
  }
  __builtin_amdgcn_s_waitcnt(kWaitAll);
#ifdef VK_STAMP
  {
    unsigned long long t_end;
    VK_T(t_end)
    if (p.stamps && lane == 0) {
      unsigned long long* o = p.stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8;
      o[0] = tkX; o[1] = tkW; o[2] = tkB1; o[3] = tkY; o[4] = tkB2; o[5] = t_pro - t_begin; o[6] = t_end - t_begin; o[7] = 3 * (c_end - c_begin);
    }
  }
#endif
  __syncthreads();                                          // every wave is done with the operand buffers: the epilogue reuses them

  if (p.ksplit > 1) {
    float* sl = p.slab + (size_t)blockIdx.z * ((size_t)p.N * p.H * p.W * p.K);
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) {
        const int y = y0 + wrow0 + b, x = x0 + li, ch = n0 + wch0 + a * 16 + kg * 4;
        if (y < p.H && x < p.W && ch < p.K) *reinterpret_cast<f32x4_t*>(sl + (((size_t)n * p.H + y) * p.W + x) * p.K + ch) = acc[a][b];
      }
    return;
  }
  halo_epilogue<T, TH, BN, TP, TC, NT, (BN <= 64 && TH == 16)>(smem, acc, p, n, y0, x0, n0, wrow0, wch0);
}

// ---- persistent form of the column-staged kernel for the K < 128 classes (layer1, decoder blocks 2-4: thousands of tiles, each a
// prologue, 3-18 pipeline stages and an epilogue — HBM / latency-bound).  A workgroup walks tiles t = blockIdx.x, + gridDim.x, ...
// and requests the NEXT tile's first halo chunk (global -> registers) during the last chunk of the current tile, so that the HBM
// round trip of a tile's prologue runs under the previous tile's MFMAs and epilogue instead of in front of its own; where the
// epilogue's LDS image ends below the weight buffers (BN <= 32) the next tile's first weight stage is in flight under the epilogue
// too.  Same arithmetic, same operand order, same per-tile statistics atomics as conv3x3_col_kernel: results are bit-identical.
template <typename T, int TH, int BN, int WGM, int WGN, int MINW>
__global__ __launch_bounds__(64 * WGM * WGN, MINW) void conv3x3_colp_kernel(const HaloParams p) {
  using Cfg = ColCfg<T, TH, BN, WGM, WGN, false>;
  constexpr int EB = Cfg::EB, VE = Cfg::VE, CK = Cfg::CK, HPIX = Cfg::HPIX, NPV = Cfg::NPV, NT = Cfg::NT, NW = Cfg::NW;
  constexpr int TP = Cfg::TP, TC = Cfg::TC, APS = Cfg::APS, NPIECE = Cfg::NPIECE;
  typedef __attribute__((address_space(3))) void lds_void;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Abuf = smem;
  char* const Bbuf = smem + Cfg::A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsp = p.N * p.tiles_y * p.tiles_x;               // spatial tiles
  const int total = nsp * p.kyn;                             // x channel tiles (kyn >= 1 here)
  const bool xcd_runs = (nsp & 7) == 0 && (gridDim.x & 7) == 0 && p.kyn == 1;

  const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(p.s0.ptr, p.s0.bytes);
  const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(p.s1.ptr ? p.s1.ptr : p.s0.ptr, p.s1.ptr ? p.s1.bytes : 0u);
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.w, p.w_bytes);

  const int hv = tid & 3;
  const int hgrp = halo_group(tid);
  const int Hh = p.H >> 1, Wh = p.W >> 1;
  // tile geometry: current tile and the one being prefetched
  struct TileGeo { int n, y0, x0, n0; };
  auto decode = [&](int t) {
    TileGeo g;
    const int ky = t % p.kyn;
    int bt = t / p.kyn;
    if (xcd_runs) bt = (bt & 7) * (nsp >> 3) + (bt >> 3);    // every XCD (its own L2) walks a contiguous run of tiles
    const int tx = bt % p.tiles_x;
    bt /= p.tiles_x;
    const int ty = bt % p.tiles_y;
    g.n = bt / p.tiles_y;
    g.y0 = ty * TH; g.x0 = tx * 16; g.n0 = ky * BN;
    return g;
  };
  auto halo_index = [&](const TileGeo& g, int (&hf)[NPV], int (&hh)[NPV]) {
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int hp = hgrp + i * (NT / 4);
      const int hy = hp / 18, hx = hp - hy * 18;
      const int y = g.y0 - 1 + hy, x = g.x0 - 1 + hx;
      const bool ok = hp < HPIX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      hf[i] = ok ? (g.n * p.H + y) * p.W + x : -1;
      hh[i] = ok ? (g.n * Hh + (y >> 1)) * Wh + (x >> 1) : -1;
    }
  };
  int h_full[NPV], h_half[NPV], h_full_n[NPV], h_half_n[NPV];
  u32x4_t areg[NPV];
  float sc[VE], sh[VE];
  bool aff = false, relu = false;

  auto load_halo = [&](int cc, const int (&hf)[NPV], const int (&hh)[NPV]) {
    const int c = cc * CK;
    const bool first = c < p.s0.C;
    const HaloSrc& sd = first ? p.s0 : p.s1;
    const int cl = (first ? c : c - p.s0.C) + hv * VE;
    aff = sd.scale != nullptr;
    relu = sd.relu != 0;
    if (aff) {
#pragma unroll
      for (int j = 0; j < VE; j += 4) {
        const f32x4_t s4 = *reinterpret_cast<const f32x4_t*>(sd.scale + cl + j);
        const f32x4_t h4 = *reinterpret_cast<const f32x4_t*>(sd.shift + cl + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[j + e] = s4[e]; sh[j + e] = h4[e]; }
      }
    }
    const bool up = sd.up != 0;
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int pix = up ? hh[i] : hf[i];
      const uint32_t off = (uint32_t)(pix * sd.C + cl) * (uint32_t)EB;
      if (first) areg[i] = buf_load16(rs0, pix >= 0 ? off : kOOB);
      else areg[i] = buf_load16(rs1, pix >= 0 ? off : kOOB);
    }
  };
  auto store_halo = [&](char* A, const int (&hf)[NPV]) {
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int hp = hgrp + i * (NT / 4);
      if (hp >= HPIX) continue;
      u32x4_t v = areg[i];
      if (aff) {
        v = AffineRelu<T>::run(v, sc, sh, relu);
        if (hf[i] < 0) v = u32x4_t{0, 0, 0, 0};         // zero padding is applied after BN+ReLU
      }
      *reinterpret_cast<u32x4_t*>(A + hp * APS + hv * 16) = v;
    }
  };
  const uint32_t slab_bytes = (uint32_t)p.K * 64u;
  const uint32_t lane16 = (uint32_t)lane * 16u;
  auto dma_b = [&](int n0, int cc, int s, int buf) {
#pragma unroll
    for (int i = 0; i < (NPIECE + NW - 1) / NW; ++i) {
      const int pc = wave + i * NW;
      if (NPIECE % NW == 0 || pc < NPIECE) {
        const int r = pc / (BN / 16), sub = pc - r * (BN / 16);
        const int tap = p.flip ? 8 - (3 * r + s) : 3 * r + s;
        const uint32_t goff = (uint32_t)(cc * 9 + tap) * slab_bytes + (uint32_t)(n0 + sub * 16) * 64u;
        char* dst = Bbuf + buf * Cfg::B_BYTES + (r * BN + sub * 16) * 64;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void*)dst, 16, lane16, goff, 0, 0);
      }
    }
  };

  const int wrow0 = (wave / WGN) * TP;
  const int wch0 = (wave % WGN) * (TC * 16);
  const int li = lane & 15, kg = lane >> 4;
  const int a_lane = (wrow0 * 18 + li) * APS + kg * 16;
  const int b_lane = (wch0 + li) * 64 + ((kg ^ swz(li)) << 4);
  f32x4_t acc[TC][TP];

  auto compute = [&](const char* A, const char* B) {
    u32x4_t X[TP + 2], W[3][TC];
#pragma unroll
    for (int h = 0; h < TP + 2; ++h) X[h] = *reinterpret_cast<const u32x4_t*>(A + h * (18 * APS));
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int a = 0; a < TC; ++a) W[r][a] = *reinterpret_cast<const u32x4_t*>(B + (r * BN + a * 16) * 64);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[a][b] = Mma<T>::run(W[r][a], X[b + r], acc[a][b]);
    constexpr int NM = TC * TP * (sizeof(T) == 4 ? 4 : 1);
    __builtin_amdgcn_sched_group_barrier(0x100, TP + 2 + TC, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, TC, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, TC, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
  };

  int t = blockIdx.x;
  if (t >= total) return;                                   // (grid is never larger than the tile count)
  TileGeo g = decode(t);
  halo_index(g, h_full, h_half);
  load_halo(0, h_full, h_half);
  double dsum[2] = {0.0, 0.0};                              // running BN sums of channel g.n0 + tid (threads < BN), see stats_flush
  int st = 0;                                               // running stage counter: weight buffer = st & 1, across tiles
  for (;;) {
    // ---- entry: the tile's first halo chunk is in areg (requested during the previous tile's last stage), no LDS-DMA is pending
    dma_b(g.n0, 0, 0, st & 1);                              // first weight stage: in flight while the halo is transformed and written
    store_halo(Abuf, h_full);
    __syncthreads();                                        // also drains the LDS-DMA (vmcnt(0))
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int tn = t + (int)gridDim.x;
    const bool more_tiles = tn < total;
    TileGeo gn = g;
    const char* const A = Abuf + a_lane;
    // one pipeline stage.  LAST = the tile's last channel chunk, written out separately so that its final stage contains NO LDS-DMA:
    // the barrier closing it and the barriers of the epilogue then need no vmcnt(0), and the next tile's halo loads (plain global
    // loads into registers) stay in flight across all of them, until store_halo at the top of the next iteration consumes them.
    auto stage = [&](int cc, auto s_c, auto last_c) {
      constexpr int s = decltype(s_c)::value;
      constexpr bool LAST = decltype(last_c)::value;
      if (s < 2) dma_b(g.n0, cc, s + 1, (st + 1) & 1);
      else if (!LAST) dma_b(g.n0, cc + 1, 0, (st + 1) & 1);
      if (s == 0 && !LAST) load_halo(cc + 1, h_full, h_half);
      if (s == 2 && LAST && more_tiles) {
        gn = decode(tn);
        halo_index(gn, h_full_n, h_half_n);
        load_halo(0, h_full_n, h_half_n);
      }
      compute(A + s * APS, Bbuf + (st & 1) * Cfg::B_BYTES + b_lane);
      if (s == 2 && !LAST) {
        __syncthreads();                                    // every wave is done reading this chunk's halo
        store_halo(Abuf, h_full);
      }
      __syncthreads();
      ++st;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    for (int cc = 0; cc + 1 < p.nchunks; ++cc) {
      stage(cc, I0{}, std::false_type{});
      stage(cc, I1{}, std::false_type{});
      stage(cc, I2{}, std::false_type{});
    }
    stage(p.nchunks - 1, I0{}, std::true_type{});
    stage(p.nchunks - 1, I1{}, std::true_type{});
    stage(p.nchunks - 1, I2{}, std::true_type{});
    // aff / relu / sc / sh now describe the prefetched chunk (chunk 0 of the next tile): exactly what its store_halo needs
    constexpr bool DEFER = VK_COLP_DEFER_STATS && (sizeof(T) == 2 || BN < 64);       // (the fp32 64-channel tile has no two registers left)
    halo_epilogue<T, TH, BN, TP, TC, NT, (BN <= 64 && TH == 16)>(smem, acc, p, g.n, g.y0, g.x0, g.n0, wrow0, wch0, 0, 0, DEFER ? dsum : nullptr);
    if (DEFER && (!more_tiles || gn.n0 != g.n0)) stats_flush<BN>(p, g.n0, dsum);
    if (!more_tiles) break;
    __syncthreads();                                        // epilogue reads of the LDS tile are over: the halo image may be rewritten
    t = tn;
    g = gn;
#pragma unroll
    for (int i = 0; i < NPV; ++i) { h_full[i] = h_full_n[i]; h_half[i] = h_half_n[i]; }
  }
}

// ---- stem: 7x7 stride-2 pad-3 convolution of the NHWC4 input (16-bit types), reference train.py:436 -> encoder.conv1.
// GEMM view: K = 7 filter rows x 32 (8 columns x 4 channels, the 8th column and the 4th channel are zero weights); the A operand of
// output pixel j and filter row r is the 64 contiguous bytes of input row 2i + r - 3 that start at pixel 2j - 3.  The tap-by-tap
// kernel gathered those 64 bytes per pixel and row from L1 / L2 (every input byte 28 times: 1.9 GB through the vector caches for a
// 67 MB input); here the 37 x 40-pixel input window of a 16 x 16 output tile is staged ONCE (11.8 KB) and the fragments are
// ds_read_b128 at byte 16 * (j + kg) of the row: consecutive lanes read consecutive 16-byte pieces (conflict-free, lanes with equal
// j + kg share one).  Global loads are 16-byte vectors of two pixels at even x (never straddle the image edge: the zero padding is
// the buffer bounds check); each is stored as two 8-byte halves one pixel to the left so that the window of pixel j starts 16-byte
// aligned.  The 28 weight fragments (7 rows x 4 channel tiles, 1 KB each) are copied to LDS by LDS-DMA in fragment order (lane =
// kg * 16 + channel: the reads are consecutive 16-byte pieces) while the window loads are in flight — fetched per filter row from
// L2 they put an L2 round trip in front of every 16 MFMAs (148 -> 126 us only).
// Wave layout, epilogue and BN statistics are those of the 16x16x64 tile: accumulation order r = 0..6.
struct StemTile {
  static constexpr int ROWS = 37, VPR = 20, PITCH = VPR * 16;        // staged rows, 16-byte vectors per row, row pitch
  static constexpr int A_BYTES = ROWS * PITCH;
  static constexpr int W_OFF = (A_BYTES + 1023) / 1024 * 1024, W_BYTES = 28 * 1024;
};

template <typename T>
__global__ __launch_bounds__(256, 3) void stem7x7_kernel(const HaloParams p) {
  using Cfg = ColCfg<T, 16, 64, 4, 1, false>;
  static_assert(sizeof(T) == 2, "16-bit types: a pixel of the NHWC4 input is 8 bytes");
  constexpr int NV = (StemTile::ROWS * StemTile::VPR + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bt = blockIdx.x;
  if ((gridDim.x & 7) == 0) bt = (bt & 7) * (int)(gridDim.x >> 3) + (bt >> 3);        // contiguous runs of tiles per XCD
  const int tx = bt % p.tiles_x;
  bt /= p.tiles_x;
  const int ty = bt % p.tiles_y;
  const int n = bt / p.tiles_y;
  const int y0 = ty * 16, x0 = tx * 16;
  const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(p.s0.ptr, p.s0.bytes);
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.w, p.w_bytes);

  // ---- input window -> registers
  u32x4_t areg[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = tid + i * 256;
    const int ry = v / StemTile::VPR, vc = v - ry * StemTile::VPR;
    const int y = 2 * y0 - 3 + ry, x = 2 * x0 - 4 + 2 * vc;
    const bool ok = v < StemTile::ROWS * StemTile::VPR && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
    areg[i] = buf_load16(rs0, ok ? (uint32_t)((n * p.Hi + y) * p.Wi + x) * 8u : kOOB);
  }
  const int li = lane & 15, kg = lane >> 4;
  // weight fragment (filter row r, channel tile a) = wp[a * 16 + li][r][kg * 8 ..] of lane kg * 16 + li -> LDS piece 4 r + a
  typedef __attribute__((address_space(3))) void lds_void;
  char* const Wl = smem + StemTile::W_OFF;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int pc = wave * 7 + i, r = pc >> 2, a = pc & 3;
    const uint32_t voff = (uint32_t)((((a * 16 + li) * 7 + r) * 32 + kg * 8) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void*)(Wl + pc * 1024), 16, voff, 0, 0, 0);
  }
  // ---- registers -> LDS, each pixel one slot to the left (the pixel at x = 2 x0 - 4 is not part of any window)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = tid + i * 256;
    if (v >= StemTile::ROWS * StemTile::VPR) continue;
    const int ry = v / StemTile::VPR, vc = v - ry * StemTile::VPR;
    char* row = smem + ry * StemTile::PITCH;
    if (vc > 0) *reinterpret_cast<u32x2_t*>(row + (2 * vc - 1) * 8) = u32x2_t{areg[i][0], areg[i][1]};
    *reinterpret_cast<u32x2_t*>(row + (2 * vc) * 8) = u32x2_t{areg[i][2], areg[i][3]};
  }
  __syncthreads();                                          // also drains the LDS-DMA (vmcnt(0))

  const int wrow0 = wave * 4;
  f32x4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const char* const xl = smem + (2 * wrow0) * StemTile::PITCH + 16 * (li + kg);
  const char* const wl = Wl + lane * 16;
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    u32x4_t X[4], Wc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) Wc[a] = *reinterpret_cast<const u32x4_t*>(wl + (4 * r + a) * 1024);
#pragma unroll
    for (int b = 0; b < 4; ++b) X[b] = *reinterpret_cast<const u32x4_t*>(xl + (2 * b + r) * StemTile::PITCH);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = Mma<T>::run(Wc[a], X[b], acc[a][b]);
  }
  __syncthreads();                                          // the epilogue reuses the window's LDS
  halo_epilogue<T, 16, 64, 4, 4, 256, false>(smem, acc, p, n, y0, x0, 0, wrow0, 0);
}

int stem_tile_launch(vk_dtype dt, int N, int H, int W, const void* x4, const void* wp, void* y, double* stats, hipStream_t st) {
  if (dt == VK_F32 || getenv("VK_NO_STEM_TILE")) return VK_ERR_UNSUPPORTED;
  if ((size_t)N * H * W * 8 >= (1ull << 31)) return VK_ERR_UNSUPPORTED;
  HaloParams p;
  memset((void*)&p, 0, sizeof(p));
  p.s0 = HaloSrc{x4, nullptr, nullptr, 4, 0, 0, (uint32_t)((size_t)N * H * W * 8)};
  p.w = wp;
  p.w_bytes = 64u * 7u * 32u * 2u;
  p.y0 = y;
  p.ld0 = 64;
  p.stats = stats;
  p.N = N; p.H = H / 2; p.W = W / 2; p.K = 64; p.C = 4;
  p.Hi = H; p.Wi = W;
  p.tiles_x = (p.W + 15) / 16;
  p.tiles_y = (p.H + 15) / 16;
  using Cfg = ColCfg<bf16_t, 16, 64, 4, 1, false>;
  constexpr int MAIN = StemTile::W_OFF + StemTile::W_BYTES;
  constexpr int SMEM = MAIN > Cfg::EPI ? MAIN : Cfg::EPI;
  const dim3 grid((unsigned)(N * p.tiles_y * p.tiles_x));
  vkh::ProfScope ps("stem_tile_16b", st, 2.0 * (double)N * p.H * p.W * 64.0 * 147.0, ((double)N * H * W * 4.0 + (double)N * p.H * p.W * 64.0) * 2.0);
  if (dt == VK_BF16) hipLaunchKernelGGL(stem7x7_kernel<bf16_t>, grid, dim3(256), SMEM, st, p);
  else hipLaunchKernelGGL(stem7x7_kernel<f16_t>, grid, dim3(256), SMEM, st, p);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

// ---- C == 16 (16-bit types): decoder block 4 conv2 and the data gradients whose reduction runs over 16 channels.
// A pixel is 32 bytes, so one K=32 MFMA step covers TWO taps x 16 channels; 9 taps = 5 steps (the 10th half reads zeros).
// Everything (18x18 halo, all 9 taps of weights) is staged once: these layers are HBM-bound, the kernel is a
// load -> 20..40 MFMAs -> store pipeline with 5 workgroups resident per CU.
template <typename T, int BN>
struct C16Cfg {
  static constexpr int TH = 16, HPIX = 18 * 18;
  static constexpr int APS = 48;                       // 32-byte pixel + 16 pad
  static constexpr int A_BYTES = HPIX * APS + 64;      // + a zero line for the non-existent 10th tap
  static constexpr int BS = 336;                       // weight row: 160 x 2 B + 16 pad
  static constexpr int B_BYTES = BN * BS;
  static constexpr int TP = 4, TC = BN / 16;
  static constexpr int ESB = BN * 2 + 16;
  static constexpr int EPI = 256 * ESB + 4 * 4 * BN * 2 * 4;     // + [wave][16-lane row][channel][2] partial sums
  static constexpr int MAIN = A_BYTES + B_BYTES;
  static constexpr int SMEM = MAIN > EPI ? MAIN : EPI;
};

template <typename T, int BN>
__global__ __launch_bounds__(256) void conv3x3_c16_kernel(const HaloParams p) {
  using Cfg = C16Cfg<T, BN>;
  constexpr int TH = 16, HPIX = Cfg::HPIX, APS = Cfg::APS, BS = Cfg::BS, TP = Cfg::TP, TC = Cfg::TC, VE = 8;
  static_assert(sizeof(T) == 2, "C16 kernel is for 16-bit element types (fp32 uses the generic kernel with 16-channel chunks)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Abuf = smem;
  char* const Bbuf = smem + Cfg::A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bt = blockIdx.x;
  // workgroups go round-robin over the 8 XCDs: give every XCD (its own L2) a contiguous run of tiles so that horizontally
  // adjacent tiles share their halo columns in one L2 (kernel level +1-2 % on the HBM-bound layers, null on the step)
  if ((gridDim.x & 7) == 0) bt = (bt & 7) * (int)(gridDim.x >> 3) + (bt >> 3);
  const int tx = bt % p.tiles_x;
  bt /= p.tiles_x;
  const int ty = bt % p.tiles_y;
  const int n = bt / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * 16;
  const int n0 = blockIdx.y * BN;
  const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(p.s0.ptr, p.s0.bytes);
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.w, p.w_bytes);

  // ---- stage halo (2 vectors per pixel) with the BN+ReLU prologue, and all weights of this channel tile
  const bool aff = p.s0.scale != nullptr, relu = p.s0.relu != 0;
  // lane map of the halo staging (r04): 8 consecutive lanes = 8 pixels, the same 16-byte piece — 48-byte pixels put 8 neighbouring
  // pixels on 8 different 16-byte slots of the 128-byte bank row; (pixel, piece) in thread order gave a 2-way conflict per store
  const int hv = (tid >> 3) & 1;
  const int hgrp = (tid & 7) | ((tid >> 4) << 3);
  float sc[VE], sh[VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) {
    sc[j] = aff ? p.s0.scale[hv * VE + j] : 1.f;
    sh[j] = aff ? p.s0.shift[hv * VE + j] : 0.f;
  }
  constexpr int NPV = (HPIX * 2 + 255) / 256;
  u32x4_t areg[NPV];
  bool aok[NPV];
#pragma unroll
  for (int i = 0; i < NPV; ++i) {
    const int hp = hgrp + i * 128;
    const int hy = hp / 18, hx = hp - hy * 18;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    aok[i] = hp < HPIX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    const uint32_t off = (uint32_t)(((n * p.H + y) * p.W + x) * 16 + hv * VE) * 2u;
    areg[i] = buf_load16(rs0, aok[i] ? off : kOOB);
  }
  constexpr int NBV = (BN * 18 + 255) / 256;          // 144 elements = 18 vectors per weight row
  u32x4_t breg[NBV];
#pragma unroll
  for (int i = 0; i < NBV; ++i) {
    const int idx = tid + i * 256;
    const int row = idx / 18, v = idx - row * 18;     // vector v = tap (v >> 1), channel half (v & 1)
    const int tap = p.flip ? 8 - (v >> 1) : (v >> 1);
    const bool ok = idx < BN * 18 && n0 + row < p.K;
    breg[i] = buf_load16(rsw, ok ? (uint32_t)(((n0 + row) * 9 + tap) * 16 + (v & 1) * VE) * 2u : kOOB);
  }
  if (tid < 4) *reinterpret_cast<u32x4_t*>(Abuf + HPIX * APS + tid * 16) = u32x4_t{0, 0, 0, 0};
  if (tid < BN) {     // zero the 16 padding elements (k = 144..159) of every weight row
    *reinterpret_cast<u32x4_t*>(Bbuf + tid * BS + 288) = u32x4_t{0, 0, 0, 0};
    *reinterpret_cast<u32x4_t*>(Bbuf + tid * BS + 304) = u32x4_t{0, 0, 0, 0};
  }
#pragma unroll
  for (int i = 0; i < NPV; ++i) {
    const int hp = hgrp + i * 128;
    if (hp >= HPIX) continue;
    u32x4_t v = areg[i];
    if (aff) {
      v = AffineRelu<T>::run(v, sc, sh, relu);
      if (!aok[i]) v = u32x4_t{0, 0, 0, 0};
    }
    *reinterpret_cast<u32x4_t*>(Abuf + hp * APS + hv * 16) = v;
  }
#pragma unroll
  for (int i = 0; i < NBV; ++i) {
    const int idx = tid + i * 256;
    if (idx < BN * 18) *reinterpret_cast<u32x4_t*>(Bbuf + (idx / 18) * BS + (idx % 18) * 16) = breg[i];
  }
  __syncthreads();

  // ---- 5 MFMA steps: lane (li, kg) covers tap 2*step + (kg >> 1), channels 8*(kg & 1)..+7
  const int wrow0 = wave * 4;
  const int li = lane & 15, kg = lane >> 4;
  f32x4_t acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int step = 0; step < 5; ++step) {
    const int tap = 2 * step + (kg >> 1);
    const int r = tap / 3, s = tap - r * 3;
    const bool zero = tap >= 9;
    // per-lane base of this step's A reads (the zero line when the tap does not exist)
    const int abase = zero ? HPIX * APS : ((wrow0 + r) * 18 + li + s) * APS + (kg & 1) * 16;
    const int astep = zero ? 0 : 18 * APS;
    u32x4_t wf[TC], xf[TP];
#pragma unroll
    for (int a = 0; a < TC; ++a) wf[a] = *reinterpret_cast<const u32x4_t*>(Bbuf + (a * 16 + li) * BS + step * 64 + kg * 16);
#pragma unroll
    for (int b = 0; b < TP; ++b) xf[b] = *reinterpret_cast<const u32x4_t*>(Abuf + abase + b * astep);
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) acc[a][b] = Mma<T>::run(wf[a], xf[b], acc[a][b]);
  }
  __syncthreads();
  halo_epilogue<T, TH, BN, TP, TC>(smem, acc, p, n, y0, x0, n0, wrow0, 0);
}

// ---- streaming form for the small-channel layers of decoder blocks 3 / 4 (r03): C_in = 16 or 32 (one source, optionally nearest-x2
// upsampled), K = 16 or 32, 16-bit types, stride 1 — forward with BatchNorm statistics, data gradient with the fused BN+ReLU-backward
// reduce and / or the 2x2 pooling (upsample backward).  These layers move 0.4-0.8 GB per launch for a few GFLOP: HBM-bound, and the
// tile kernels run them at 25-40 % of the HBM peak — every 16x16 tile is a load -> barrier -> MFMAs -> barrier -> LDS transposition
// -> store -> 32 fp64 atomics chain, and the only latency hiding is having 3-5 such workgroups per CU.
// Here every WAVE is its own pipeline and there is no workgroup barrier at all:
//   * a wave owns a strip of 16 output columns x RS rows and walks down it; per output row it needs ONE new input row
//     (18 pixels = 36 / 72 sixteen-byte vectors; the rows above are still in its private LDS ring): each input pixel is fetched
//     once per strip (+ 2 / 16 halo columns), BN+ReLU-transformed once, written to LDS once;
//   * the next three input rows are always in flight in registers (a shift-register queue), so the HBM round trip is covered by
//     the wave's own look-ahead instead of by neighbours;
//   * the filter (<= 9 x 32 x 32) lives in registers for the whole strip; fragment reads are ds_read_b128 from the ring with the
//     conflict-free 48 / 96-byte pixel stride of the tile kernels; 5 (C = 16: two taps per K = 32 step) or 9 MFMAs per row and
//     16-channel tile;
//   * output rows leave straight from the accumulators (lane = 4 consecutive channels of one pixel: 8-byte stores, 512 contiguous
//     bytes per row for K = 16); BatchNorm sums stay in registers for the whole strip and become 16 fp64 atomics per wave at its end.
// Arithmetic per output element: taps in ascending order, one chunk — at C = 16 that is the tile kernel's order (the same bits), at
// C = 32 the tile kernels add (filter column, filter row) instead: results differ in fp32 rounding order (a last-bit flip of the
// stored value on < 5 % of the elements, tests/test_ops_gpu.py::test_stream_conv_kernels_match_tile_kernels).  Statistics are over the
// stored (rounded) values, pooling adds the four rounded values in the tile kernels' order.
#ifndef VK_STREAM_DIAG
#define VK_STREAM_DIAG 0      // diagnostic builds only (tests/diag/stream_race_diag.py)
#endif
template <typename T, int CIN, bool UP>
struct StreamCfg {
  static constexpr int NSTEP = CIN == 16 ? 5 : 9;
  static constexpr int APS = CIN == 16 ? 48 : 96;          // LDS pixel stride (conflict-free fragment reads, as C16Cfg / ColCfg)
  static constexpr int NPX = UP ? 10 : 18;                 // staged source pixels per row
  static constexpr int VPP = CIN / 8;                      // 16-byte vectors per pixel
  static constexpr int NVEC = NPX * VPP;                   // vectors per staged row
  static constexpr int NLD = (NVEC + 63) / 64;             // loads per lane and row
  static constexpr int ROWB = CIN == 16 ? 1024 : 2048;     // ring slot
  static constexpr int RING = 4, DEPTH = 3;
  static constexpr int ZERO_OFF = RING * ROWB;
  static constexpr int WAVE_LDS = RING * ROWB + 64;        // + a zero line (the non-existent 10th tap at C = 16)
  static constexpr int SMEM = 4 * WAVE_LDS;
  static constexpr int RS = 32;                            // output rows per strip (even: 2x2 pooling pairs rows)
  static_assert(NPX * APS <= ROWB && NLD <= 2, "row image");
};

// MODE: 0 forward (operand transform, optional BatchNorm statistics), 2 data gradient + fused BN+ReLU-backward reduce of the layer
// below, 3 the same behind the 2x2 pooling (upsample backward).  Compile-time, so that the row loop is straight-line code per phase and
// the compiler can count which vector-memory operations are still in flight (runtime branches made it wait vmcnt(0) per row).
template <typename T, int CIN, int TC, bool UP, int MODE>
__global__ __launch_bounds__(256) void conv3x3_stream_kernel(const HaloParams p) {
  using Cfg = StreamCfg<T, CIN, UP>;
  constexpr int NSTEP = Cfg::NSTEP, APS = Cfg::APS, VPP = Cfg::VPP, NVEC = Cfg::NVEC, NLD = Cfg::NLD, ROWB = Cfg::ROWB;
  constexpr int VE = 8;
  const int RS = p.tiles_y;                                // strip height (launch_stream): a multiple of 8
  constexpr bool BNR = MODE >= 2, POOL = MODE == 3;
  static_assert(sizeof(T) == 2, "16-bit element types");
  static_assert(!(UP && MODE != 0), "the upsampled source occurs in forward launches only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* const ring = smem + wave * Cfg::WAVE_LDS;
  const int strips_x = (p.W + 15) / 16, strips_y = (p.H + RS - 1) / RS;
  int sid = (int)blockIdx.x * 4 + wave;
  if (sid >= p.N * strips_y * strips_x) return;               // no workgroup barrier anywhere below: waves are independent
  const int strip = sid;
  const int sx = sid % strips_x;
  sid /= strips_x;
  const int sy = sid % strips_y;
  const int n = sid / strips_y;
  const int x0 = sx * 16, ys = sy * RS, ye = min(p.H, ys + RS);
  const int li = lane & 15, kg = lane >> 4;
  const int Hs = p.H >> (UP ? 1 : 0), Ws = p.W >> (UP ? 1 : 0);
  const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(p.s0.ptr, p.s0.bytes);
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.w, p.w_bytes);

  // ---- the filter: registers, for the whole strip.  C = 16: plain [K][9][16] weights, step = taps (2 step, 2 step + 1);
  // C = 32: halo pack [9][K][4 swizzled 16-byte pieces], step = tap
  u32x4_t wf[NSTEP][TC];
#pragma unroll
  for (int st = 0; st < NSTEP; ++st)
#pragma unroll
    for (int a = 0; a < TC; ++a) {
      const int row = a * 16 + li;
      uint32_t off;
      bool ok = row < p.K;
      if (CIN == 16) {
        const int tap = 2 * st + (kg >> 1);
        ok = ok && tap < 9;
        const int tp = p.flip ? 8 - tap : tap;
        off = (uint32_t)(((row * 9 + tp) * 16 + (kg & 1) * VE) * 2);
      } else {
        const int tp = p.flip ? 8 - st : st;
        const int pos = kg ^ (((row >> 2) & 1) << 1);
        off = (uint32_t)(((tp * p.K + row) * 4 + pos) * 16);
      }
      wf[st][a] = buf_load16(rsw, ok ? off : kOOB);
    }
  if (lane < 4) *reinterpret_cast<u32x4_t*>(ring + Cfg::ZERO_OFF + lane * 16) = u32x4_t{0, 0, 0, 0};

  // ---- staging geometry of this lane (fixed for the strip): vector v -> source pixel hx, 16-byte piece hv
  const bool aff = MODE == 0 && p.s0.scale != nullptr, relu = p.s0.relu != 0;
  int st_off[NLD], ld_col[NLD];
  bool xok[NLD];
  float sc[VE], sh[VE];
#pragma unroll
  for (int q = 0; q < NLD; ++q) {
    const int v = lane + 64 * q;
    // lane -> (staged pixel, 16-byte piece): inside complete groups the order that keeps the 8 lanes of a ds_write_b128 lane group on 8
    // different 16-byte slots of the 128-byte bank row (see halo_group: 96-byte pixels pair p with p + 2, 48-byte pixels take 8
    // neighbouring pixels with the same piece); the incomplete last group keeps thread order
    int hx = v / VPP, hv = v % VPP;
    if (VPP == 4) {
      if ((hx | 3) < Cfg::NPX) hx = (hx & ~3) | ((hx & 1) << 1) | ((hx >> 1) & 1);
    } else {
      if ((v | 15) < NVEC) { hx = (v & 7) | ((v >> 4) << 3); hv = (v >> 3) & 1; }
    }
    const int xs = (UP ? (x0 >> 1) : x0) - 1 + hx;
    xok[q] = v < NVEC && (unsigned)xs < (unsigned)Ws;
    ld_col[q] = xs * CIN + hv * VE;
    st_off[q] = v < NVEC ? hx * APS + hv * 16 : -1;
  }
  if (MODE == 0) {
    // the channels of this lane's vector(s): VPP = 4: piece lane % 4 for both (64 % 4 == 0, the pixel permutation leaves the piece
    // alone); VPP = 2: one vector per lane (NLD == 1), its piece follows the lane map above
    static_assert(VPP == 4 || NLD == 1, "C = 16: one staged vector per lane");
    const int hv = VPP == 4 ? lane % VPP : (((lane | 15) < NVEC) ? (lane >> 3) & 1 : lane % VPP);
#pragma unroll
    for (int j = 0; j < VE; ++j) {
      sc[j] = aff ? p.s0.scale[hv * VE + j] : 1.f;
      sh[j] = aff ? p.s0.shift[hv * VE + j] : 0.f;
    }
  }
  auto issue = [&](int j, u32x4_t (&r)[NLD]) {      // request source row j (zeros outside the map)
    const bool rok = (unsigned)j < (unsigned)Hs;
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const uint32_t off = (uint32_t)(((n * Hs + j) * Ws) * CIN + ld_col[q]) * 2u;
      r[q] = buf_load16(rs0, (rok && xok[q]) ? off : kOOB);
    }
  };
  auto write_row = [&](int j, const u32x4_t (&r)[NLD]) {
    const bool rok = (unsigned)j < (unsigned)Hs;
    char* const slot = ring + (j & 3) * ROWB;
#if (VK_STREAM_DIAG & 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      if (st_off[q] < 0) continue;
      u32x4_t v = r[q];
      if (MODE == 0) {
        if (aff) v = AffineRelu<T>::run(v, sc, sh, relu);
        if (!(rok && xok[q])) v = u32x4_t{0, 0, 0, 0};     // zero padding applies AFTER the transform
      }
      *reinterpret_cast<u32x4_t*>(slot + st_off[q]) = v;
    }
#if (VK_STREAM_DIAG & 2)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  };

  // ---- fragment-read geometry: step -> (filter row r, column s); C = 16: per lane (two taps per step), C = 32: per step
  int fr_r16[CIN == 16 ? NSTEP : 1], fr_off16[CIN == 16 ? NSTEP : 1];
  if (CIN == 16) {
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int tap = 2 * st + (kg >> 1);
      const int r = tap / 3, sxx = tap - r * 3;
      fr_r16[st] = tap < 9 ? r : -100;
      fr_off16[st] = (li + sxx) * APS + (kg & 1) * 16;
    }
  }
  const int fr_base32 = kg * 16;                          // + pixel * APS, pixel = li + s or (li + s + 1) >> 1

  // ---- output side
  const int ld = p.ld0;
  const bool want_sums = p.stats != nullptr || p.bnr_sums != nullptr;
  const int xo = x0 + li;
  const bool x_ok = xo < p.W;
  float bsc[TC][4], bsh[TC][4], s1[TC][4], s2[TC][4], prev[TC][4];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ch = a * 16 + kg * 4 + e;
      bsc[a][e] = BNR && ch < ld ? p.bnr_scale[ch] : 0.f;
      bsh[a][e] = BNR && ch < ld ? p.bnr_shift[ch] : 0.f;
      s1[a][e] = 0.f; s2[a][e] = 0.f; prev[a][e] = 0.f;
    }
  const int Hh = p.H >> 1, Wh = p.W >> 1;
  auto round_t = [&](float (&f)[4]) {              // to T and back: everything downstream sees the stored values
    float g[8] = {f[0], f[1], f[2], f[3], 0.f, 0.f, 0.f, 0.f};
    const u32x4_t pk = Vec16<T>::pack(g);
    Vec16<T>::unpack(pk, g);
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = g[e];
    return u32x2_t{pk[0], pk[1]};
  };
  // byte offset of this lane's 4 channels (tile a) at output row y (pooled modes: the half-resolution pixel) in the output / in the
  // bnr_z tensor, as a 32-bit offset: lanes and rows that must not touch memory get an out-of-range offset (top bit set), which a
  // buffer load answers with zeros and the store skips — no 64-bit address arithmetic per row (r03: the per-row `if (ok)` blocks
  // with their own v_mad_u64 chains were a quarter of the instructions of the fused data-gradient row; K == 16 TC == ld here)
  const int Ho = POOL ? Hh : p.H, Wo = POOL ? Wh : p.W;
  const uint32_t out_bytes = (uint32_t)((size_t)p.N * Ho * Wo * ld * 2);
  const __amdgpu_buffer_rsrc_t rsbz = make_rsrc(BNR ? p.bnr_z : p.y0, out_bytes);
#ifdef VK_STREAM_BUFSTORE
  const __amdgpu_buffer_rsrc_t rsy = make_rsrc(p.y0, out_bytes);
#endif
  const bool lane_stores = x_ok && (!POOL || (li & 1) == 0);
  const uint32_t lane_ob = lane_stores ? (uint32_t)(((n * Ho) * Wo + (POOL ? (xo >> 1) : xo)) * ld + kg * 4) * 2u : kOOB;
  const uint32_t row_ob = (uint32_t)(Wo * ld * 2);
  auto out_off = [&](int a, int y) -> uint32_t {           // y < p.H is the caller's business
    return lane_ob + (uint32_t)(POOL ? (y >> 1) : y) * row_ob + (uint32_t)(a * 32);
  };
  // the z vectors of the fused BN+ReLU-backward reduce are requested one output row (pooled: one row pair) ahead
  u32x2_t zq[2][TC];
  auto z_issue = [&](int y, u32x2_t (&z)[TC]) {
    const bool ok = y < p.H;
#pragma unroll
    for (int a = 0; a < TC; ++a) z[a] = __builtin_amdgcn_raw_buffer_load_b64(rsbz, ok ? out_off(a, y) : kOOB, 0, 0);
  };
  // masks with the BN+ReLU of the layer below, adds the sums, stores 4 channels of one pixel at element offset eoff
  auto finish = [&](int a, float (&f)[4], u32x2_t pk, uint32_t boff, bool counted, u32x2_t zr) {
    if (BNR) {
      float zf[8];
      Vec16<T>::unpack(u32x4_t{zr[0], zr[1], 0u, 0u}, zf);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (!counted || !(fmaf(zf[e], bsc[a][e], bsh[a][e]) > 0.f)) f[e] = 0.f;      // pixels outside the map add nothing
        s1[a][e] += f[e];
        s2[a][e] += f[e] * zf[e];
      }
      pk = round_t(f);
    } else if (want_sums) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = counted ? f[e] : 0.f;
        s1[a][e] += v;
        s2[a][e] += v * v;
      }
    }
    // a plain (global) store under a lane predicate.  (r03 blamed MUBUF stores for the wrong 16-bit inference of 04e3d09..75808a4;
    // r04 found the cause elsewhere — an MFMA-result read 2 wait states behind its MFMA across a branch, see the sched_barrier in
    // row() — and tests/diag/vmcnt_order_probe.hip shows stores never leave the vmcnt queue ahead of older loads on this chip:
    // 0 of 2.1 M wave-probes.  VK_STREAM_BUFSTORE rebuilds the MUBUF form for tests/diag/stream_race_diag.py.)
#ifdef VK_STREAM_BUFSTORE      // diagnostic build only (r03 bug reconstruction, tests/diag/README.md): the MUBUF store form
#if (VK_STREAM_DIAG & 8)
    if ((int32_t)boff >= 0)
#endif
    __builtin_amdgcn_raw_buffer_store_b64(pk, rsy, boff, 0, 0);
#if (VK_STREAM_DIAG & 4)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#else
    if ((int32_t)boff >= 0) *reinterpret_cast<u32x2_t*>((char*)p.y0 + boff) = pk;
#endif
  };

  // ---- the row pipeline.  Source row j lives in queue register set j & 3 until it is written to ring slot j & 3; the loop is
  // unrolled over the phase of y so that every set index is a compile-time constant (a shift-register queue would MOVE registers
  // that are the destination of loads in flight, i.e. wait for them: measured, 1.4 us per row).  Look-ahead: three source rows.
  u32x4_t pre[4][NLD];
  const int jb = UP ? (ys >> 1) : ys;                      // ys is a multiple of RS = 32, so jb & 3 == 0
  issue(jb - 1, pre[3]);
  issue(jb, pre[0]);
  issue(jb + 1, pre[1]);
  issue(jb + 2, pre[2]);
  if (BNR) z_issue(POOL ? ys + 1 : ys, zq[0]);
  write_row(jb - 1, pre[3]);
  issue(jb + 3, pre[3]);
  write_row(jb, pre[0]);
  // one output row; PH = (y - ys) mod UF, compile-time.  Rows at or beyond H inside the last group are computed on zero rows and
  // not stored (lane predicate, no branch).
  auto row = [&](int y, auto ph_c) {
    constexpr int PH = decltype(ph_c)::value;
    // the source row this output row newly needs: non-UP: y + 1 every row; UP: (y + 1) >> 1 on odd rows
    if constexpr (!UP) {
      constexpr int S = (PH + 1) & 3;                      // slot of row y + 1
      write_row(y + 1, pre[S]);
      issue(y + 4, pre[PH & 3]);                           // row y + 4 shares the slot of row y, written one iteration ago
    } else if constexpr ((PH & 1) == 1) {
      constexpr int M1 = ((PH + 1) >> 1) & 3;              // slot of source row m + 1 = (y + 1) >> 1
      const int m1 = (y + 1) >> 1;
      write_row(m1, pre[M1]);
      issue(m1 + 3, pre[(M1 + 3) & 3]);
    }
    // next row's (row pair's) z vectors
    constexpr int ZC = POOL ? ((PH >> 1) & 1) : (PH & 1);  // set holding THIS row's z
    if constexpr (BNR && (!POOL || (PH & 1) == 1)) z_issue(POOL ? y + 2 : y + 1, zq[ZC ^ 1]);
    f32x4_t acc[TC];
#pragma unroll
    for (int a = 0; a < TC; ++a) acc[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      int addr;
      if (CIN == 16) {
        const int vrow = y - 1 + fr_r16[st];
        addr = fr_r16[st] < 0 ? Cfg::ZERO_OFF : (vrow & 3) * ROWB + fr_off16[st];
      } else {
        const int r = st / 3, sxx = st - r * 3;
        const int vrow = y - 1 + r;
        const int jr = UP ? (vrow >> 1) : vrow;
        const int pix = UP ? ((li + sxx + 1) >> 1) : (li + sxx);
        addr = (jr & 3) * ROWB + pix * APS + fr_base32;
      }
      const u32x4_t xf = *reinterpret_cast<const u32x4_t*>(ring + addr);
#pragma unroll
      for (int a = 0; a < TC; ++a) acc[a] = Mma<T>::run(wf[st][a], xf, acc[a]);
    }
    // Every MFMA of the row stays in front of every accumulator read of its epilogue.  r03's wrong 16-bit inference (root cause found in
    // r04): with MUBUF stores in the epilogue hipcc sank the row's LAST MFMA (tile 1) below tile 0's accumulator
    // reads, directly in front of the wave-uniform `want_sums` branch, and the block at the branch target opens with the
    // v_accvgpr_read of that MFMA's result — 2 wait states behind it on the taken edge (no statistics pointer) instead of the 8 the
    // hazard recogniser pads everywhere else: element 3 of every lane's channel group came out stale in every fourth output row.
    // With the barrier the last MFMA is followed by straight-line code (padded correctly); tools/mfma_hazard_audit.py checks the
    // shipped binary for this pattern along every control-flow path (tests/test_host_cpu.py).
#if !(VK_STREAM_DIAG & 16)      // 16: the r03 code without the barrier (diagnostic builds only)
    __builtin_amdgcn_sched_barrier(0);
#endif
    const bool y_ok = y < p.H;
#pragma unroll
    for (int a = 0; a < TC; ++a) {
      const int ch = a * 16 + kg * 4;
      float f[4] = {acc[a][0], acc[a][1], acc[a][2], acc[a][3]};
      u32x2_t pk = round_t(f);
      if constexpr (POOL) {
        // rows pair up inside the strip (ys and RS are even); the four ROUNDED values are added in the tile kernels' order:
        // (row 2q, col 2c), (2q, 2c + 1), (2q + 1, 2c), (2q + 1, 2c + 1)
        if constexpr ((PH & 1) == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) prev[a][e] = f[e];
        } else {
          float t[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float pn = dpp_f<0xB1>(prev[a][e]), cn = dpp_f<0xB1>(f[e]);      // the odd neighbour column
            t[e] = ((prev[a][e] + pn) + f[e]) + cn;
          }
          const u32x2_t pk2 = round_t(t);
          finish(a, t, pk2, y_ok ? out_off(a, y) : kOOB, lane_stores && y_ok, zq[ZC][a]);
        }
      } else {
        finish(a, f, pk, y_ok ? out_off(a, y) : kOOB, lane_stores && y_ok, zq[ZC][a]);
      }
    }
  };
  constexpr int UF = UP ? 8 : 4;
  for (int yb = ys; yb < ye; yb += UF) {
    row(yb, std::integral_constant<int, 0>{});
    row(yb + 1, std::integral_constant<int, 1>{});
    row(yb + 2, std::integral_constant<int, 2>{});
    row(yb + 3, std::integral_constant<int, 3>{});
    if constexpr (UP) {
      row(yb + 4, std::integral_constant<int, 4>{});
      row(yb + 5, std::integral_constant<int, 5>{});
      row(yb + 6, std::integral_constant<int, 6>{});
      row(yb + 7, std::integral_constant<int, 7>{});
    }
  }

  // ---- BatchNorm sums of the strip: 16 lanes of a row hold the same channels -> DPP row sum, one lane per row adds them
  if (want_sums) {
    double* const sp = (p.bnr_sums ? p.bnr_sums : p.stats) + (size_t)(strip % VK_STATS_REPLICAS) * 2 * ld;
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float u = row16_sum(s1[a][e]), v = row16_sum(s2[a][e]);
        const int ch = a * 16 + kg * 4 + e;
        if (li == 0 && ch < ld) {
          atomicAdd(sp + ch, (double)u);
          atomicAdd(sp + ld + ch, (double)v);
        }
      }
  }
}

// ---- eval only (r03): decoder block 4 conv1 -> BN+ReLU -> conv2 -> BN+ReLU -> head in ONE kernel over an overlapping tile, 16-bit types.
// In inference the BatchNorm affines are constants, so nothing forces the two 512^2 x 16 tensors between these three layers through
// HBM (training cannot do this: the batch statistics need the whole tensor between the convolutions).  Per 16 x 16 logits tile:
//   phase 0  the 12 x 12 SOURCE pixels (32 channels, half resolution) under the 22 x 22 upsampled window: BN+ReLU of the layer below,
//            staged once as T (zeros outside the map);
//   phase 1  conv1 on the 20 x 20 window: 25 groups of 16 linear pixels x 9 MFMAs (K = 32 = the 32 channels of one tap; the nearest-x2
//            upsample is the index (p + d + 1) >> 1 into the source tile), rounded to T like the stored z1, BN1+ReLU as the consumer
//            would apply it on load, zero outside the map (conv2's padding) -> LDS;
//   phase 2  conv2 on the 18 x 18 window: 21 groups x 5 MFMAs (two taps of 16 channels per K = 32 step), rounded to T like the stored
//            z2, BN2+ReLU in fp32 as the head applies it -> LDS (fp32);
//   phase 3  the head's 144 FMAs per pixel from that tile (the code of k_head_fwd) -> logits.
// Same operand order per accumulator as the kernels it replaces (conv3x3_stream_kernel, k_head_fwd): the same bits.  Redundant work:
// (20/16)^2 on conv1, (18/16)^2 on conv2; HBM traffic per image 4.2 MB in + 1 MB out instead of 38.8 MB.
struct TailCfg {
  static constexpr int S_STRIDE = 96, A1_STRIDE = 48, A2_PS = 20;
  static constexpr int S_BYTES = 144 * S_STRIDE;                    // 13,824
  static constexpr int A1_BYTES = 400 * A1_STRIDE + 64;             // + a zero line (the non-existent tenth tap)
  static constexpr int A2_BYTES = 324 * A2_PS * 4;
  static constexpr int SMEM = S_BYTES + A1_BYTES + A2_BYTES;        // 59,008
};

struct TailParams {
  const void* src;                 // [N][H/2][W/2][32] raw output of the layer below
  const float* s0;                 // its BatchNorm affine (scale, shift), ReLU
  const float* h0;
  const void* w1;                  // conv1: halo pack, 16 rows x 32 channels
  const float* s1;
  const float* h1;
  const void* w2;                  // conv2: plain [16][9][16]
  const float* s2;
  const float* h2;
  const float* hw;                 // head [9][16] fp32
  const float* hb;
  float* logits;                   // [N][H][W]
  int N, H, W;
  uint32_t src_bytes;
};

template <typename T>
__global__ __launch_bounds__(256) void dec4_tail_eval_kernel(const TailParams p) {
  using Cfg = TailCfg;
  static_assert(sizeof(T) == 2, "16-bit element types");
  constexpr int VE = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Sb = smem;
  char* const A1 = smem + Cfg::S_BYTES;
  float* const A2 = reinterpret_cast<float*>(smem + Cfg::S_BYTES + Cfg::A1_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kg = lane >> 4;
  const int tiles_x = p.W >> 4, tiles_y = p.H >> 4;
  int bt = blockIdx.x;
  const int tx0 = bt % tiles_x;
  bt /= tiles_x;
  const int ty0 = bt % tiles_y;
  const int n = bt / tiles_y;
  const int y0 = ty0 * 16, x0 = tx0 * 16;
  const int Hs = p.H >> 1, Ws = p.W >> 1;
  const int sy0 = (y0 >> 1) - 2, sx0 = (x0 >> 1) - 2;
  const __amdgpu_buffer_rsrc_t rss = make_rsrc(p.src, p.src_bytes);

  // ---- weights as constant fragments (layouts of conv3x3_stream_kernel: halo pack for C = 32, plain [K][9][16] for C = 16)
  u32x4_t wf1[9], wf2[5];
#pragma unroll
  for (int st = 0; st < 9; ++st) {
    const int pos = kg ^ (((li >> 2) & 1) << 1);
    wf1[st] = *reinterpret_cast<const u32x4_t*>((const char*)p.w1 + ((st * 16 + li) * 4 + pos) * 16);
  }
#pragma unroll
  for (int st = 0; st < 5; ++st) {
    const int tap = 2 * st + (kg >> 1);
    wf2[st] = u32x4_t{0, 0, 0, 0};
    if (tap < 9) wf2[st] = *reinterpret_cast<const u32x4_t*>((const char*)p.w2 + ((li * 9 + tap) * 16 + (kg & 1) * VE) * 2);
  }
  float sc1[8], sh1[8], sc2[4], sh2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sc1[e] = p.s1[kg * 4 + e]; sh1[e] = p.h1[kg * 4 + e];
    sc1[4 + e] = 0.f; sh1[4 + e] = 0.f;
    sc2[e] = p.s2[kg * 4 + e]; sh2[e] = p.h2[kg * 4 + e];
  }
  if (tid < 4) *reinterpret_cast<u32x4_t*>(A1 + 400 * Cfg::A1_STRIDE + tid * 16) = u32x4_t{0, 0, 0, 0};

  // ---- phase 0: source tile, 144 pixels x 4 pieces
  {
    float sc0[VE], sh0[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) { sc0[j] = p.s0[(tid & 3) * VE + j]; sh0[j] = p.h0[(tid & 3) * VE + j]; }
    u32x4_t r[3];
    bool ok[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int v = tid + 256 * q;
      const int px = v >> 2;
      const int sy = sy0 + px / 12, sx = sx0 + px % 12;
      ok[q] = v < 576 && (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws;
      r[q] = buf_load16(rss, ok[q] ? (uint32_t)((((n * Hs + sy) * Ws + sx) * 32 + (v & 3) * VE) * 2) : kOOB);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int v = tid + 256 * q;
      if (v >= 576) continue;
      u32x4_t x = AffineRelu<T>::run(r[q], sc0, sh0, true);
      if (!ok[q]) x = u32x4_t{0, 0, 0, 0};
      *reinterpret_cast<u32x4_t*>(Sb + (v >> 2) * Cfg::S_STRIDE + (v & 3) * 16) = x;
    }
  }
  __syncthreads();

  // ---- phase 1: conv1 on the 20 x 20 window (origin (y0 - 2, x0 - 2)); group = 16 consecutive linear pixels
  for (int g = wave; g < 25; g += 4) {
    const int pl = g * 16 + li;
    const int py = pl / 20, px = pl - py * 20;
    f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 9; ++st) {
      const int r = st / 3, sxx = st - r * 3;
      const int srow = (py + r + 1) >> 1, scol = (px + sxx + 1) >> 1;      // nearest-x2: window pixel + tap -> source tile pixel
      const u32x4_t xf = *reinterpret_cast<const u32x4_t*>(Sb + (srow * 12 + scol) * Cfg::S_STRIDE + kg * 16);
      acc = Mma<T>::run(wf1[st], xf, acc);
    }
    float f[8] = {acc[0], acc[1], acc[2], acc[3], 0.f, 0.f, 0.f, 0.f};
    const u32x4_t z1 = Vec16<T>::pack(f);                                   // the value conv1 would have stored
    u32x4_t a1 = AffineRelu<T>::run(z1, sc1, sh1, true);                    // ... and conv2 would have staged
    const int Y = y0 - 2 + py, X = x0 - 2 + px;
    if ((unsigned)Y >= (unsigned)p.H || (unsigned)X >= (unsigned)p.W) a1 = u32x4_t{0, 0, 0, 0};      // conv2's zero padding
    *reinterpret_cast<u32x2_t*>(A1 + pl * Cfg::A1_STRIDE + kg * 8) = u32x2_t{a1[0], a1[1]};
  }
  __syncthreads();

  // ---- phase 2: conv2 on the 18 x 18 window (origin (y0 - 1, x0 - 1)), two taps per K = 32 step
  for (int g = wave; g < 21; g += 4) {
    const int ql = g * 16 + li;
    const bool live = ql < 324;
    const int q = live ? ql : 0;
    const int qy = q / 18, qx = q - qy * 18;
    f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 5; ++st) {
      const int tap = 2 * st + (kg >> 1);
      const int r = tap / 3, sxx = tap - r * 3;
      const int addr = tap < 9 ? ((qy + r) * 20 + qx + sxx) * Cfg::A1_STRIDE + (kg & 1) * 16 : 400 * Cfg::A1_STRIDE;
      const u32x4_t xf = *reinterpret_cast<const u32x4_t*>(A1 + addr);
      acc = Mma<T>::run(wf2[st], xf, acc);
    }
    float f[8] = {acc[0], acc[1], acc[2], acc[3], 0.f, 0.f, 0.f, 0.f};
    const u32x4_t z2 = Vec16<T>::pack(f);                                   // the value conv2 would have stored
    Vec16<T>::unpack(z2, f);
    const int Y = y0 - 1 + qy, X = x0 - 1 + qx;
    const bool inside = (unsigned)Y < (unsigned)p.H && (unsigned)X < (unsigned)p.W;
    f32x4_t a2;
#pragma unroll
    for (int e = 0; e < 4; ++e) a2[e] = inside ? fmaxf(fmaf(f[e], sc2[e], sh2[e]), 0.f) : 0.f;      // as k_head_fwd stages it (fp32)
    if (live) *reinterpret_cast<f32x4_t*>(A2 + q * Cfg::A2_PS + kg * 4) = a2;
  }
  __syncthreads();

  // ---- phase 3: the head (k_head_fwd's arithmetic, same order)
  {
    const int ty = tid >> 4, tx = tid & 15;
    typedef float f32x2_t_ __attribute__((ext_vector_type(2)));
    f32x2_t_ acc_a = f32x2_t_{p.hb[0], 0.f}, acc_b = f32x2_t_{0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) {
        const float* a = A2 + ((ty + r) * 18 + tx + s_) * Cfg::A2_PS;
        const float* wt = p.hw + (r * 3 + s_) * 16;
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
          const f32x4_t av = *reinterpret_cast<const f32x4_t*>(a + j);
          acc_a = __builtin_elementwise_fma(f32x2_t_{av[0], av[1]}, f32x2_t_{wt[j], wt[j + 1]}, acc_a);
          acc_b = __builtin_elementwise_fma(f32x2_t_{av[2], av[3]}, f32x2_t_{wt[j + 2], wt[j + 3]}, acc_b);
        }
      }
    p.logits[((size_t)n * p.H + y0 + ty) * p.W + x0 + tx] = (acc_a[0] + acc_a[1]) + (acc_b[0] + acc_b[1]);
  }
}

int dec4_tail_eval_impl(vk_dtype dt, int N, int H, int W, const vk_src* src, const void* w1_pack, const float* scale1, const float* shift1,
                        const void* w2_plain, const float* scale2, const float* shift2, const float* head_w, const float* head_b, float* logits,
                        hipStream_t st) {
  VK_CHECK_ARG(dt != VK_F32, "vk_dec4_tail_eval: 16-bit element types only");
  VK_CHECK_ARG(src && src->ptr && src->C == 32 && src->up == 1 && src->scale && src->shift && src->relu, "vk_dec4_tail_eval: source must be 32 channels, upsampled, with its BN+ReLU");
  VK_CHECK_ARG(w1_pack && scale1 && shift1 && w2_plain && scale2 && shift2 && head_w && head_b && logits, "vk_dec4_tail_eval: null argument");
  VK_CHECK_ARG(N > 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0, "vk_dec4_tail_eval: %d x %d is not a multiple of 16", H, W);
  VK_CHECK_ARG((size_t)N * (H / 2) * (W / 2) * 32 * 2 < (1ull << 31), "vk_dec4_tail_eval: source of 2 GiB or more");
  TailParams p;
  p.src = src->ptr; p.s0 = src->scale; p.h0 = src->shift;
  p.w1 = w1_pack; p.s1 = scale1; p.h1 = shift1;
  p.w2 = w2_plain; p.s2 = scale2; p.h2 = shift2;
  p.hw = head_w; p.hb = head_b; p.logits = logits;
  p.N = N; p.H = H; p.W = W;
  p.src_bytes = (uint32_t)((size_t)N * (H / 2) * (W / 2) * 32 * 2);
  const unsigned nb = (unsigned)((size_t)N * (H / 16) * (W / 16));
  const double px = (double)N * H * W;
  vkh::ProfScope ps("dec4_tail_eval_16b", st, 2.0 * px * (9.0 * 32 * 16 + 9.0 * 16 * 16 + 144.0), px * (32.0 * 2 / 4 + 4.0));
  if (dt == VK_BF16) hipLaunchKernelGGL(dec4_tail_eval_kernel<bf16_t>, dim3(nb), dim3(256), TailCfg::SMEM, st, p);
  else hipLaunchKernelGGL(dec4_tail_eval_kernel<f16_t>, dim3(nb), dim3(256), TailCfg::SMEM, st, p);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

// ---- weight repack into the halo layout.  src: [rows][9][red] of T (rows = output channels of the GEMM: K for forward
// weights [K][3][3][C], C for the transposed data-gradient weights [C][3][3][K]; red = reduction channels).
// dst: [red / CK][9][rows][CK] with the four 16-byte pieces of every 64-byte row stored at piece ^ (((row >> 2) & 1) << 1).
template <typename T>
__global__ void k_pack_halo(int rows, int red, const T* __restrict__ src, T* __restrict__ dst) {
  constexpr int CK = 64 / ElemTraits<T>::kBytes, VE = ElemTraits<T>::kVec;
  const size_t total = (size_t)rows * 9 * red / VE;            // 16-byte vectors
  for (size_t v = blockIdx.x * (size_t)blockDim.x + threadIdx.x; v < total; v += (size_t)gridDim.x * blockDim.x) {
    // enumerate destination vectors: ((cc*9 + tap)*rows + k)*4 + pos
    const int pos = (int)(v & 3);
    size_t t = v >> 2;
    const int k = (int)(t % rows);
    t /= rows;
    const int tap = (int)(t % 9);
    const int cc = (int)(t / 9);
    const int j = pos ^ (((k >> 2) & 1) << 1);
    const u32x4_t x = *reinterpret_cast<const u32x4_t*>(src + ((size_t)k * 9 + tap) * red + cc * CK + j * VE);
    *reinterpret_cast<u32x4_t*>(dst + v * VE) = x;
  }
}

int halo_pack_impl(vk_dtype dt, int rows, int red, const void* src, void* dst, hipStream_t st) {
  const int eb = dt == VK_F32 ? 4 : 2;
  VK_CHECK_ARG(src && dst && rows > 0 && red % (64 / eb) == 0, "vk_halo_pack: reduction channels %d must be a multiple of %d", red, 64 / eb);
  const size_t nvec = (size_t)rows * 9 * red * eb / 16;
  unsigned nb = (unsigned)((nvec + 255) / 256);
  if (nb > 4096) nb = 4096;
  switch (dt) {
    case VK_F32: hipLaunchKernelGGL(k_pack_halo<float>, dim3(nb), dim3(256), 0, st, rows, red, (const float*)src, (float*)dst); break;
    case VK_BF16: hipLaunchKernelGGL(k_pack_halo<bf16_t>, dim3(nb), dim3(256), 0, st, rows, red, (const bf16_t*)src, (bf16_t*)dst); break;
    case VK_F16: hipLaunchKernelGGL(k_pack_halo<f16_t>, dim3(nb), dim3(256), 0, st, rows, red, (const f16_t*)src, (f16_t*)dst); break;
  }
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

// ------------------------------------------------------------------------------------------------ host
template <typename T, int TH, int BN, int WGM, int WGN>
static int launch_halo(HaloParams p, hipStream_t st) {
  using Cfg = HaloCfg<T, TH, BN, WGM, WGN>;
  p.tiles_x = (p.W + 15) / 16;
  p.tiles_y = (p.H + TH - 1) / TH;
  dim3 grid((unsigned)(p.N * p.tiles_y * p.tiles_x), (unsigned)((p.K + BN - 1) / BN), 1);
  static bool attr_done = false;
  if (!attr_done && Cfg::SMEM > 64 * 1024) {
    VK_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_halo_kernel<T, TH, BN, WGM, WGN>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
    attr_done = true;
  }
  {
    static const std::string tag_f = std::string("halo_") + (sizeof(T) == 4 ? "f32" : "16b") + "_t" + std::to_string(TH) + "_bn" + std::to_string(BN);
    static const std::string tag_d = tag_f + "_dgrad";
    const double macs = (double)p.N * p.H * p.W * p.K * 9.0 * p.C;
    const double bytes = ((double)p.N * p.H * p.W * (p.C + p.K) + 9.0 * p.K * p.C) * sizeof(T);
    const std::string& btag = p.flip ? tag_d : tag_f;
    const std::string dtag = getenv("VK_PROF_DETAIL") ? btag + ":H" + std::to_string(p.H) + "_K" + std::to_string(p.K) + "_C" + std::to_string(p.C) : btag;
    vkh::ProfScope ps(dtag.c_str(), st, 2.0 * macs, bytes);
    hipLaunchKernelGGL((conv3x3_halo_kernel<T, TH, BN, WGM, WGN>), grid, dim3(256), Cfg::SMEM, st, p);
  }
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

// y[e] = T(sum_s slab[s][e]) in slice order (reproducible), four elements per thread
template <typename T>
__global__ __launch_bounds__(256) void k_splitk_reduce(size_t n4, int splits, const float* __restrict__ slab, T* __restrict__ y) {
  const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (e >= n4) return;
  f32x4_t a = *reinterpret_cast<const f32x4_t*>(slab + e * 4);
  for (int s = 1; s < splits; ++s) a = a + *reinterpret_cast<const f32x4_t*>(slab + ((size_t)s * n4 + e) * 4);
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<f32x4_t*>(y + e * 4) = a;
  } else {
    float f[8] = {a[0], a[1], a[2], a[3], 0.f, 0.f, 0.f, 0.f};
    const u32x4_t pk = Vec16<T>::pack(f);
    *reinterpret_cast<u32x2_t*>(y + e * 4) = u32x2_t{pk[0], pk[1]};
  }
}

template <int PIPE, typename T, int TH, int BN, int WGM, int WGN, bool ADB, int MINW, int STR>
struct ColLaunch {
  using Cfg = ColCfg<T, TH, BN, WGM, WGN, ADB, STR>;
  static const void* kernel() { return (const void*)conv3x3_col_kernel<T, TH, BN, WGM, WGN, ADB, MINW, STR>; }
};
template <typename T, int TH, int BN, int WGM, int WGN, bool ADB, int MINW>
struct ColLaunch<1, T, TH, BN, WGM, WGN, ADB, MINW, 1> {
  using Cfg = ColqCfg<T, TH, BN, WGM, WGN>;
  static const void* kernel() { return (const void*)conv3x3_colq_kernel<T, TH, BN, WGM, WGN>; }
};
template <typename T, int TH, int BN, int WGM, int WGN, bool ADB, int MINW>
struct ColLaunch<2, T, TH, BN, WGM, WGN, ADB, MINW, 1> {
  using Cfg = ColsCfg<T, TH, BN, WGM, WGN>;
  static const void* kernel() { return (const void*)conv3x3_cols_kernel<T, TH, BN, WGM, WGN>; }
};

template <typename T, int TH, int BN, int WGM, int WGN, bool ADB, int MINW, int PIPE = 0, int STR = 1>
static int launch_col(HaloParams p, hipStream_t st) {
  using Sel = ColLaunch<PIPE, T, TH, BN, WGM, WGN, ADB, MINW, STR>;
  using Cfg = typename Sel::Cfg;
  const void* const kernel = Sel::kernel();
  p.tiles_x = (p.W + 15) / 16;
  p.tiles_y = (p.H + TH - 1) / TH;
  dim3 grid((unsigned)(p.N * p.tiles_y * p.tiles_x), (unsigned)((p.K + BN - 1) / BN), 1);
  // split-K for grids that leave most of the chip idle (batch-1 inference: 8-64 tiles per layer): slices of the channel
  // chunks in blockIdx.z, fp32 partial tiles into the caller's workspace, one reduce launch.  Only the plain path (no
  // statistics / accumulate / fused gradients), so training launches never take it.
  const size_t out_elems = (size_t)p.N * p.H * p.W * p.K;
  int ks = 1;
  if (p.slab && !p.stats && !p.bnr_z && !p.accumulate && !p.pool2 && p.split == 0 && !getenv("VK_NO_SPLITK")) {
    const long wgs = (long)grid.x * grid.y;
    const char* force = getenv("VK_SPLITK");
    ks = force ? atoi(force) : (wgs < 128 ? (int)(256 / wgs) : 1);
    if (ks > p.nchunks / 2) ks = p.nchunks / 2;             // at least two chunks (six pipeline stages) per slice
    if (ks > 32) ks = 32;
    while (ks > 1 && (size_t)ks * out_elems * sizeof(float) > p.slab_bytes) --ks;
    if (ks < 2) ks = 1;
  }
  p.ksplit = ks;
  grid.z = (unsigned)ks;
  p.kyn = 0;
  if (ks == 1 && grid.y > 1 && (grid.x & 7) == 0 && !getenv("VK_COL_NO_KYFAST")) {
    p.kyn = (int)grid.y;
    grid.x *= grid.y;
    grid.y = 1;
    const char* km = getenv("VK_COL_KYMODE");               // w: weights-stationary XCDs (needs 8 % kyn == 0 and whole runs of tiles)
    // default: on the layers whose weights do not fit an XCD's L2 beside the activations (>= 3 MB: layer 4, decoder block 0 conv1;
    // measured r02: +2-4 % there, -2 % on the 1.2 MB layer-3 weights)
    const bool wst = km ? km[0] == 'w' : (size_t)p.w_bytes >= (3u << 20);
    if (PIPE && wst && 8 % p.kyn == 0 && (grid.x / 8) * (8 / p.kyn) * p.kyn == grid.x) p.kyn = -p.kyn;
  }
  static bool attr_done = false;
  if (!attr_done && Cfg::SMEM > 64 * 1024) {
    VK_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
    attr_done = true;
  }
  {
    static const std::string tag_f = std::string(PIPE == 2 ? "cols_" : PIPE == 1 ? "colq_" : "col_") + (sizeof(T) == 4 ? "f32" : "16b") + "_t" + std::to_string(TH) + "_bn" +
                                     std::to_string(BN) + "_w" + std::to_string(WGM * WGN) + (STR == 2 ? "_s2" : "");
    static const std::string tag_d = tag_f + "_dgrad";
    const double macs = (double)p.N * p.H * p.W * p.K * 9.0 * p.C;
    const double bytes = ((double)p.N * ((double)p.Hi * p.Wi * p.C + (double)p.H * p.W * p.K) + 9.0 * p.K * p.C) * sizeof(T);
    const std::string& btag = p.flip ? tag_d : tag_f;
    const std::string dtag = getenv("VK_PROF_DETAIL") ? btag + ":H" + std::to_string(p.H) + "_K" + std::to_string(p.K) + "_C" + std::to_string(p.C) : btag;
    vkh::ProfScope ps(dtag.c_str(), st, 2.0 * macs, bytes);
    void* args[] = {(void*)&p};
    VK_CHECK_HIP(hipLaunchKernel(kernel, grid, dim3(Cfg::NT), args, Cfg::SMEM, st));
  }
  if (ks > 1) {
    vkh::ProfScope ps("splitk_reduce", st, 0.0, (double)out_elems * (4.0 * ks + sizeof(T)));
    hipLaunchKernelGGL(k_splitk_reduce<T>, dim3((unsigned)((out_elems / 4 + 255) / 256)), dim3(256), 0, st, out_elems / 4, ks, p.slab, (T*)p.y0);
  }
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

// persistent launch of conv3x3_colp_kernel: as many workgroups as stay resident (LDS-limited), at least two tiles each
template <typename T, int TH, int BN, int WGM, int WGN, int MINW>
static int launch_colp(HaloParams p, hipStream_t st) {
  using Cfg = ColCfg<T, TH, BN, WGM, WGN, false>;
  p.tiles_x = (p.W + 15) / 16;
  p.tiles_y = (p.H + TH - 1) / TH;
  p.kyn = (p.K + BN - 1) / BN;
  p.ksplit = 1;
  const long total = (long)p.N * p.tiles_y * p.tiles_x * p.kyn;
  int per_cu = (160 * 1024) / Cfg::SMEM;
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  long g = 256L * per_cu;
  if (const char* e = getenv("VK_COL_PERSIST_GRID")) g = atol(e) > 0 ? atol(e) : g;      // tests: few workgroups, many tiles each
  if (g > total) g = total;
  static bool attr_done = false;
  if (!attr_done && Cfg::SMEM > 64 * 1024) {
    VK_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_colp_kernel<T, TH, BN, WGM, WGN, MINW>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
    attr_done = true;
  }
  {
    static const std::string tag_f = std::string("col_") + (sizeof(T) == 4 ? "f32" : "16b") + "_t" + std::to_string(TH) + "_bn" + std::to_string(BN) +
                                     "_w" + std::to_string(WGM * WGN);
    static const std::string tag_d = tag_f + "_dgrad";
    const double macs = (double)p.N * p.H * p.W * p.K * 9.0 * p.C;
    const double bytes = ((double)p.N * p.H * p.W * (p.C + p.K) + 9.0 * p.K * p.C) * sizeof(T);
    const std::string& btag = p.flip ? tag_d : tag_f;
    const std::string dtag = getenv("VK_PROF_DETAIL") ? btag + ":H" + std::to_string(p.H) + "_K" + std::to_string(p.K) + "_C" + std::to_string(p.C) : btag;
    vkh::ProfScope ps(dtag.c_str(), st, 2.0 * macs, bytes);
    hipLaunchKernelGGL((conv3x3_colp_kernel<T, TH, BN, WGM, WGN, MINW>), dim3((unsigned)g), dim3(Cfg::NT), Cfg::SMEM, st, p);
  }
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

// the K < 128 classes: persistent kernel when a workgroup gets several tiles and nothing needs the split-K path
template <typename T, int BN>
static int launch_small(const HaloParams& p, hipStream_t st) {
  const long total = (long)p.N * ((p.H + 15) / 16) * ((p.W + 15) / 16) * ((p.K + BN - 1) / BN);
  // OFF by default (measured, r02): the prefetched halo + second index set cost 40-50 registers, i.e. 4 -> 2 (BN = 16) and 3 -> 2
  // (BN = 32) workgroups per CU: dec4.conv1 forward 242 -> 325 us, dec3.conv2 84 -> 108 us; on BN = 64 (equal occupancy) it gains
  // 2-3 % (layer1 forward 61.6 -> 59.6 us) — the co-resident workgroups already hide each other's prologue.  Kept as an opt-in and
  // tested (VK_COL_PERSIST=N: from N tiles on).
  // r02: on by default for BN = 64 when a workgroup gets at least two tiles (VK_COL_PERSIST=0 switches it off)
  const char* pe = getenv("VK_COL_PERSIST");
  const long min_tiles = pe ? (atol(pe) == 0 ? -1 : atol(pe)) : (BN == 64 ? 1024 : -1);
  const bool splitk_wanted = p.slab && !p.stats && !p.bnr_z && !p.accumulate && !p.pool2 && p.split == 0 && total < 128;
  if (min_tiles > 0 && !splitk_wanted && total >= min_tiles) return launch_colp<T, 16, BN, 4, 1, 2>(p, st);
  return launch_col<T, 16, BN, 4, 1, false, 2>(p, st);
}

// tile selection for the column-staged kernels.  VK_COL_ALT (diagnostic / tests) forces a shape of the K >= 128 class:
//   1: 4 waves, 16x16x128, 8 rows x 64 channels per wave (one wave per SIMD);  2: the 8-wave 16x16x128 tile;  3: the 4-wave 8x16x128 tile;
//   7: the 8-wave 8x16x128 tile;  8: the 4-wave 16x16x64 tile
template <typename T>
static int col_select(const HaloParams& p, hipStream_t st) {
  const char* alt_s = getenv("VK_COL_ALT");
  const int alt = alt_s ? atoi(alt_s) : 0;
  const long tiles16 = (long)p.N * ((p.H + 15) / 16) * ((p.W + 15) / 16);
  if (p.K >= 128) {
    const long kt = (p.K + 127) / 128;
    if (alt == 1) return launch_col<T, 16, 128, 2, 2, true, 1>(p, st);
    // a short reduction onto a channel count that is not a multiple of 128 (the concat gradient of decoder block 2: 64 -> 192):
    // 64-channel tiles waste no half-empty channel tile (201 -> 156 us stand-alone)
    // ... and a ONE-chunk reduction onto 128+ channels (the concat gradient of decoder block 3: 32 -> 128) as well: the 64-channel
    // tile gets all three filter columns in one LDS-DMA round (346 -> 300 us stand-alone)
    if (alt == 8 || (alt == 0 && p.nchunks <= 2 && p.K % 128 != 0) || (alt == 0 && p.nchunks == 1)) return launch_small<T, 64>(p, st);
    // a reduction of one or two channel chunks (the concat gradients of decoder blocks 2/3: HBM / epilogue-bound, thousands of
    // tiles): the 4-wave 8x16x128 tile, two workgroups per CU
    if (alt == 3 || (alt != 2 && alt != 7 && p.nchunks <= 2)) return launch_col<T, 8, 128, 2, 2, false, 2>(p, st);
    // too few 16x16 tiles to fill the chip (layer4: 256 workgroups of 8x16x128): the same tile on EIGHT waves (4 rows x 32
    // channels each) — one workgroup per CU either way, but two waves per SIMD instead of one (L4 dgrad 46.6 -> 41.3 us)
    // the 8-wave tiles run software-pipelined (conv3x3_colq_kernel); VK_COL_PIPE=0 (diagnostic / tests): the plain stage loop
    // VK_COL_PIPE=2: the staggered form (conv3x3_cols_kernel)
    const char* const pipe_s = getenv("VK_COL_PIPE");
    const int pipe = pipe_s ? atoi(pipe_s) : 1;
    if (alt == 7 || (alt != 2 && tiles16 * kt < 256))
      return pipe == 2 ? launch_col<T, 8, 128, 2, 4, true, 1, 2>(p, st)
             : pipe    ? launch_col<T, 8, 128, 2, 4, true, 1, 1>(p, st) : launch_col<T, 8, 128, 2, 4, true, 2>(p, st);
    // experiment (VK_COL_ALT=9): 8 rows x 32 channels per wave — 16 instead of 18 fragment reads per stage and wave (the K >= 128 loop asks
    // 75 % of the LDS read bandwidth at the full MFMA rate: DESIGN.md section 6)
    if (alt == 9) return launch_col<T, 16, 128, 2, 4, true, 1, 1>(p, st);
    return pipe == 2 ? launch_col<T, 16, 128, 4, 2, true, 1, 2>(p, st)       // 8 waves, 4 rows x 64 channels per wave
           : pipe    ? launch_col<T, 16, 128, 4, 2, true, 1, 1>(p, st) : launch_col<T, 16, 128, 4, 2, true, 2>(p, st);
  }
  if (p.K >= 64) return launch_small<T, 64>(p, st);
  // measured and rejected (r02): 32 x 16 pixel tiles for K <= 32 on the 256x256 / 512x512 maps (half the prologues / epilogues per
  // byte): dec3.conv1 forward 265 -> 255 us, dec4.conv1 244 -> 243 us, dec3.conv2 data gradient + reduce 109 -> 131 us
  if (p.K >= 32) return launch_small<T, 32>(p, st);
  return launch_small<T, 16>(p, st);
}

template <typename T>
static int halo_select(const HaloParams& p, hipStream_t st) {
  const bool old_loop = getenv("VK_HALO_ROWSTAGED") != nullptr;             // diagnostic: the v1 row-staged main loop
  if (!old_loop) return col_select<T>(p, st);
  const long tiles8 = (long)p.N * ((p.H + 7) / 8) * ((p.W + 15) / 16);
  const long tiles16 = (long)p.N * ((p.H + 15) / 16) * ((p.W + 15) / 16);
  if (p.K >= 128) {
    if (tiles8 * ((p.K + 127) / 128) >= 512) return launch_halo<T, 8, 128, 2, 2>(p, st);
    return launch_halo<T, 8, 64, 2, 2>(p, st);
  }
  if (p.K >= 64) {
    if (tiles16 >= 1024) return launch_halo<T, 16, 64, 4, 1>(p, st);
    return launch_halo<T, 8, 64, 2, 2>(p, st);
  }
  if (p.K >= 32) return launch_halo<T, 16, 32, 4, 1>(p, st);
  return launch_halo<T, 16, 16, 4, 1>(p, st);
}

template <typename T, int BN>
static int launch_c16(HaloParams p, hipStream_t st) {
  using Cfg = C16Cfg<T, BN>;
  p.tiles_x = (p.W + 15) / 16;
  p.tiles_y = (p.H + 15) / 16;
  dim3 grid((unsigned)(p.N * p.tiles_y * p.tiles_x), (unsigned)((p.K + BN - 1) / BN), 1);
  {
    static const std::string tag_f = std::string("c16_16b_bn") + std::to_string(BN);
    static const std::string tag_d = tag_f + "_dgrad";
    const std::string& btag = p.flip ? tag_d : tag_f;
    const double bytes = ((double)p.N * p.H * p.W * (16.0 + p.K * (p.pool2 ? 0.25 : 1.0)) + 9.0 * p.K * 16.0) * 2.0;
    vkh::ProfScope ps(btag.c_str(), st, 2.0 * (double)p.N * p.H * p.W * p.K * 144.0, bytes);
    hipLaunchKernelGGL((conv3x3_c16_kernel<T, BN>), grid, dim3(256), Cfg::SMEM, st, p);
  }
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

template <typename T, int BN>
static int launch_s2dg(HaloParams p, hipStream_t st) {
  using Cfg = S2dCfg<T, 8, BN, 4, 2>;
  p.tiles_x = (p.Wi + 15) / 16;
  p.tiles_y = (p.Hi + 7) / 8;
  dim3 grid((unsigned)(p.N * p.tiles_y * p.tiles_x), (unsigned)((p.K + BN - 1) / BN), 1);
  static bool attr_done = false;
  if (!attr_done && Cfg::SMEM > 64 * 1024) {
    VK_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_s2dg_kernel<T, 8, BN, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
    attr_done = true;
  }
  {
    static const std::string tag = std::string("s2dg_") + (sizeof(T) == 4 ? "f32" : "16b") + "_bn" + std::to_string(BN);
    const std::string dtag = getenv("VK_PROF_DETAIL") ? tag + ":H" + std::to_string(p.H) + "_K" + std::to_string(p.K) + "_C" + std::to_string(p.C) : tag;
    const double bytes = ((double)p.N * ((double)p.Hi * p.Wi * p.C + (double)p.H * p.W * p.K * (p.accumulate ? 2.0 : 1.0)) + 9.0 * p.K * p.C) * sizeof(T);
    vkh::ProfScope ps(dtag.c_str(), st, 2.0 * (double)p.N * p.Hi * p.Wi * p.K * 9.0 * p.C, bytes);
    hipLaunchKernelGGL((conv3x3_s2dg_kernel<T, 8, BN, 4, 2>), grid, dim3(Cfg::NT), Cfg::SMEM, st, p);
  }
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

template <typename T, int CIN, int TC, bool UP, int MODE>
static int launch_stream(const HaloParams& p, hipStream_t st) {
  using Cfg = StreamCfg<T, CIN, UP>;
  HaloParams q = p;
  // strip height: as tall as leaves >= 4096 strips (two rounds of the 2048 resident waves) — fewer prologues and statistics atomics
  // per output row; measured (profiles/r03/stream_strip_height.log): 512^2 maps best at 128 rows (dec4.conv2 data gradient + reduce
  // 185 -> 153 us), 256^2 maps at 32-64 (128 rows = 1024 strips: 75 -> 101 us).  VK_STREAM_RS overrides (tests / sweeps).
  const char* e_rs = getenv("VK_STREAM_RS");
  int RS = Cfg::RS;
  if (e_rs) {
    RS = atoi(e_rs);
  } else {
    const long per_row_block = (long)p.N * ((p.W + 15) / 16);
    while (RS * 2 <= 256 && per_row_block * ((p.H + 2 * RS - 1) / (2 * RS)) >= 4096) RS *= 2;
  }
  RS = (RS + 7) & ~7;
  if (RS < 8) RS = 8;
  q.tiles_y = RS;
  const long strips = (long)p.N * ((p.H + RS - 1) / RS) * ((p.W + 15) / 16);
  dim3 grid((unsigned)((strips + 3) / 4), 1, 1);
  static const std::string tag_f = std::string("stream_16b_c") + std::to_string(CIN) + (UP ? "up" : "") + "_k" + std::to_string(TC * 16);
  static const std::string tag_d = tag_f + "_dgrad";
  const std::string& tag = p.flip ? tag_d : tag_f;
  const double in_px = (double)p.N * (p.H >> (UP ? 1 : 0)) * (p.W >> (UP ? 1 : 0));
  const double out_px = (double)p.N * p.H * p.W * (p.pool2 ? 0.25 : 1.0);
  const double bytes = (in_px * CIN + out_px * p.K * (p.bnr_z ? 2.0 : 1.0) + 9.0 * p.K * CIN) * 2.0;
  vkh::ProfScope ps(tag.c_str(), st, 2.0 * (double)p.N * p.H * p.W * p.K * 9.0 * CIN, bytes);
  hipLaunchKernelGGL((conv3x3_stream_kernel<T, CIN, TC, UP, MODE>), grid, dim3(256), Cfg::SMEM, st, q);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

// the streaming kernel is instantiated for the launches of decoder blocks 3 / 4 that it is faster on (one source, K in one part, no
// accumulate); everything else stays on the tile kernels.  VK_NO_STREAM=1 (tests / A-B): tile kernels only;
// VK_STREAM_ALL=1: every instantiated combination, also where the tile kernels measured faster
template <typename T>
static int stream_try(const HaloParams& p, hipStream_t st) {
  if (getenv("VK_NO_STREAM")) return VK_ERR_UNSUPPORTED;
  if (p.s1.ptr || p.accumulate || p.split || p.ld0 != p.K) return VK_ERR_UNSUPPORTED;
  const bool up = p.s0.up != 0, bnr = p.bnr_z != nullptr, pool = p.pool2 != 0;
  if ((pool || up) && ((p.H | p.W) & 1)) return VK_ERR_UNSUPPORTED;
  const bool fwd = !p.flip && !bnr && !pool;                                     // forward launch (with or without statistics)
  if (fwd && p.C == 32 && up && p.K == 16) return launch_stream<T, 32, 1, true, 0>(p, st);        // dec4.conv1
  if (fwd && p.C == 16 && !up && p.K == 16) return launch_stream<T, 16, 1, false, 0>(p, st);      // dec4.conv2
  if (fwd && p.C == 32 && !up && p.K == 32) return launch_stream<T, 32, 2, false, 0>(p, st);      // dec3.conv2
  if (p.flip && !p.s0.scale && bnr && !pool && p.C == 16 && p.K == 16) return launch_stream<T, 16, 1, false, 2>(p, st);   // d(dec4.conv2)
  if (p.flip && !p.s0.scale && bnr && !pool && p.C == 32 && p.K == 32) return launch_stream<T, 32, 2, false, 2>(p, st);   // d(dec3.conv2)
  if (p.flip && !p.s0.scale && bnr && pool && p.C == 16 && p.K == 32) return launch_stream<T, 16, 2, false, 3>(p, st);    // d(dec4.conv1)
  return VK_ERR_UNSUPPORTED;
}

// returns VK_ERR_UNSUPPORTED when the shape is not covered (caller falls back to the tap-by-tap kernel)
// w: halo pack (vk_halo_pack) when `packed`, plain [K][3][3][C] otherwise — only the C == 16 kernel takes the plain layout
int conv3x3_halo_try(const vk_conv_desc* d, const void* w, int packed, void* y, void* y1, int split_k1, int accumulate, double* stats,
                     int pool2, const vk_bnr* bnr, hipStream_t st, void* workspace, size_t workspace_bytes) {
  // stride 2 (forward only): the K >= 128 openers of layers 2-4 — even input, no upsample / concat / fused gradient epilogues
  const bool s2 = d->stride == 2 && !d->transposed && d->H == 2 * d->Ho && d->W == 2 * d->Wo && d->K >= 128 && !d->src0.up && !d->src1.ptr &&
                  !pool2 && !bnr && !accumulate && !split_k1 && !getenv("VK_NO_S2_TILE");
  // stride-2 data gradient: dz [H][W] -> dx [2H][2W]; plain or accumulating store only
  const bool s2d = d->stride == 2 && d->transposed && d->Ho == 2 * d->H && d->Wo == 2 * d->W && d->K >= 64 && d->K % 64 == 0 && !d->src0.up &&
                   !d->src1.ptr && !d->src0.scale && !pool2 && !bnr && !split_k1 && !getenv("VK_NO_S2_TILE");
  if (d->R != 3 || d->S != 3 || d->pad != 1) return VK_ERR_UNSUPPORTED;
  if (!s2 && !s2d && (d->stride != 1 || d->H != d->Ho || d->W != d->Wo)) return VK_ERR_UNSUPPORTED;
  const int eb = d->dtype == VK_F32 ? 4 : 2;
  const int ck = 64 / eb;
  const int C = d->src0.C + (d->src1.ptr ? d->src1.C : 0);
  const bool c16 = (eb == 2) && C == 16 && !d->src1.ptr && !d->src0.up && !s2 && !s2d;
  if (c16 ? packed : !packed) return VK_ERR_UNSUPPORTED;
  if (!c16 && (d->src0.C % ck || (d->src1.ptr && d->src1.C % ck))) return VK_ERR_UNSUPPORTED;
  if (d->K % 16) return VK_ERR_UNSUPPORTED;
  if (pool2 && ((d->H | d->W) & 1)) return VK_ERR_UNSUPPORTED;
  if (d->src1.ptr && d->src1.up) return VK_ERR_UNSUPPORTED;
  if ((size_t)d->N * d->H * d->W * C * eb >= (1ull << 31) || (size_t)d->N * d->H * d->W >= (1ull << 31)) return VK_ERR_UNSUPPORTED;
  if (s2d && (size_t)d->N * d->Ho * d->Wo * d->K * eb >= (1ull << 32)) return VK_ERR_UNSUPPORTED;
  HaloParams p;
  auto mk = [&](const vk_src& s) {
    HaloSrc h;
    h.ptr = s.ptr; h.scale = s.scale; h.shift = s.shift; h.C = s.C; h.up = s.up; h.relu = s.relu;
    h.bytes = s.ptr ? (uint32_t)((size_t)d->N * (d->H >> s.up) * (d->W >> s.up) * s.C * eb) : 0u;
    return h;
  };
  p.s0 = mk(d->src0);
  if (d->src1.ptr) p.s1 = mk(d->src1);
  else p.s1 = HaloSrc{nullptr, nullptr, nullptr, 0, 0, 0, 0u};
  p.w = w;
  p.w_bytes = (uint32_t)((size_t)d->K * 9 * C * eb);
  p.y0 = y; p.y1 = y1; p.split = split_k1;
  p.ld0 = split_k1 ? split_k1 : d->K;
  p.ld1 = split_k1 ? d->K - split_k1 : 0;
  p.stats = stats;
  p.N = d->N; p.H = d->Ho; p.W = d->Wo; p.K = d->K; p.C = C;
  p.Hi = d->H; p.Wi = d->W;
  p.flip = d->transposed;
  p.accumulate = accumulate;
  p.pool2 = pool2;
  p.ksplit = 1;
  p.kyn = 0;
  p.slab = (reinterpret_cast<uintptr_t>(workspace) & 15) ? nullptr : (float*)workspace;      // 16-byte stores into the slab
  p.slab_bytes = p.slab ? workspace_bytes : 0;
  p.stamps = nullptr;
  p.dbg = getenv("VK_COL_DBG") ? atoi(getenv("VK_COL_DBG")) : 0;
  if (p.dbg & 1) p.w_bytes = 0;
#ifdef VK_STAMP
  { extern unsigned long long* g_vk_stamp_buf; p.stamps = g_vk_stamp_buf; }
#endif
  p.bnr_z = bnr ? bnr->z : nullptr;
  p.bnr_scale = bnr ? bnr->scale : nullptr;
  p.bnr_shift = bnr ? bnr->shift : nullptr;
  p.bnr_sums = bnr ? bnr->sums : nullptr;
  p.bnr_mask = bnr ? bnr->mask : nullptr;
  p.nchunks = c16 ? 1 : C / ck;
  if (p.bnr_mask && (c16 || s2 || s2d || pool2)) return VK_ERR_UNSUPPORTED;      // the block-tail form lives in the stride-1 tile epilogue only
  if (eb == 2 && !s2 && !s2d && !p.bnr_mask && (c16 || C == 32)) {      // small-channel decoder layers: the streaming kernel where it covers the launch
    const int rc = d->dtype == VK_BF16 ? stream_try<bf16_t>(p, st) : stream_try<f16_t>(p, st);
    if (rc != VK_ERR_UNSUPPORTED) return rc;
  }
  if (c16) {
    if (d->dtype == VK_BF16) return d->K >= 32 ? launch_c16<bf16_t, 32>(p, st) : launch_c16<bf16_t, 16>(p, st);
    return d->K >= 32 ? launch_c16<f16_t, 32>(p, st) : launch_c16<f16_t, 16>(p, st);
  }
  p.tiles_x = p.tiles_y = 0;
  if (s2d) {
    p.flip = 0;
    // 128 output channels per workgroup unless that leaves CUs idle (layer 4: 64 dz tiles x 2 channel tiles)
    const long tiles = (long)d->N * ((d->H + 7) / 8) * ((d->W + 15) / 16);
    const bool wide = d->K % 128 == 0 && tiles * (d->K / 128) >= 256;
    switch (d->dtype) {
      case VK_F32: return wide ? launch_s2dg<float, 128>(p, st) : launch_s2dg<float, 64>(p, st);
      case VK_BF16: return wide ? launch_s2dg<bf16_t, 128>(p, st) : launch_s2dg<bf16_t, 64>(p, st);
      case VK_F16: return wide ? launch_s2dg<f16_t, 128>(p, st) : launch_s2dg<f16_t, 64>(p, st);
    }
    return VK_ERR_ARG;
  }
  if (s2) {
    p.slab = nullptr;                                      // no split-K on this path
    p.slab_bytes = 0;
    switch (d->dtype) {
      case VK_F32: return launch_col<float, 8, 128, 2, 4, false, 1, 0, 2>(p, st);
      case VK_BF16: return launch_col<bf16_t, 8, 128, 2, 4, false, 1, 0, 2>(p, st);
      case VK_F16: return launch_col<f16_t, 8, 128, 2, 4, false, 1, 0, 2>(p, st);
    }
    return VK_ERR_ARG;
  }
  switch (d->dtype) {
    case VK_F32: return halo_select<float>(p, st);
    case VK_BF16: return halo_select<bf16_t>(p, st);
    case VK_F16: return halo_select<f16_t>(p, st);
  }
  return VK_ERR_ARG;
}

}  // namespace vk

extern "C" int vk_dec4_tail_eval(vk_dtype dtype, int N, int H, int W, const vk_src* src, const void* w1_pack, const float* scale1,
                                 const float* shift1, const void* w2_plain, const float* scale2, const float* shift2, const float* head_w9x16,
                                 const float* head_bias, float* logits, void* stream) {
  return vk::dec4_tail_eval_impl(dtype, N, H, W, src, w1_pack, scale1, shift1, w2_plain, scale2, shift2, head_w9x16, head_bias, logits,
                                 (hipStream_t)stream);
}

