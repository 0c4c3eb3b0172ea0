// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the U-Net path.
// Everything here is written for wave64 / MFMA / LDS directly; there is no other backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vk_unet.h"

namespace vk {

// ---------------------------------------------------------------- element types
struct bf16_t { uint16_t v; };   // storage-only wrappers: arithmetic is always done in fp32
struct f16_t  { uint16_t v; };

typedef __attribute__((ext_vector_type(8))) __bf16   bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float    f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((ext_vector_type(4))) short    s16x4_t;
typedef __attribute__((ext_vector_type(2))) float    f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16   bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) short    s16x2_t;

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int kBytes = 4;
  static constexpr int kVec = 4;     // elements per 16-byte vector
  static constexpr vk_dtype kDtype = VK_F32;
};
template <> struct ElemTraits<bf16_t> {
  static constexpr int kBytes = 2;
  static constexpr int kVec = 8;
  static constexpr vk_dtype kDtype = VK_BF16;
};
template <> struct ElemTraits<f16_t> {
  static constexpr int kBytes = 2;
  static constexpr int kVec = 8;
  static constexpr vk_dtype kDtype = VK_F16;
};

// NOTE: never __builtin_bit_cast an ext-vector ELEMENT (v[i]) directly: hipcc (ROCm 7.2) takes the address
// of the whole vector and silently returns element 0.  Go through these by-value helpers.
__device__ __forceinline__ float as_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u32(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ f16x2_t as_f16x2(uint32_t u) { return __builtin_bit_cast(f16x2_t, u); }
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return as_f32(b << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  // plain cast => v_cvt_pk_bf16_f32 on gfx950 (round-to-nearest-even, NaN stays NaN)
  __bf16 h = (__bf16)f;
  return (uint32_t)__builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float f16_bits_to_f32(uint32_t b) {
  return (float)__builtin_bit_cast(_Float16, (uint16_t)b);
}
__device__ __forceinline__ uint32_t f32_to_f16_bits(float f) {
  _Float16 h = (_Float16)f;
  return (uint32_t)__builtin_bit_cast(uint16_t, h);
}

// scalar load / store of one element as fp32
template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf16_bits_to_f32(p->v); }
template <> __device__ __forceinline__ float ld1<f16_t>(const f16_t* p) { return f16_bits_to_f32(p->v); }
template <typename T> __device__ __forceinline__ void st1(T* p, float f);
template <> __device__ __forceinline__ void st1<float>(float* p, float f) { *p = f; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float f) { p->v = (uint16_t)f32_to_bf16_bits(f); }
template <> __device__ __forceinline__ void st1<f16_t>(f16_t* p, float f) { p->v = (uint16_t)f32_to_f16_bits(f); }

// value after a round trip through T (what is actually stored)
template <typename T> __device__ __forceinline__ float round_to(float f);
template <> __device__ __forceinline__ float round_to<float>(float f) { return f; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float f) { return bf16_bits_to_f32(f32_to_bf16_bits(f)); }
template <> __device__ __forceinline__ float round_to<f16_t>(float f) { return f16_bits_to_f32(f32_to_f16_bits(f)); }

// 16-byte vector <-> kVec floats
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static __device__ __forceinline__ void unpack(u32x4_t v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = as_f32(v[i]);
  }
  static __device__ __forceinline__ u32x4_t pack(const float* f) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = as_u32(f[i]);
    return v;
  }
};
template <> struct Vec16<bf16_t> {
  static __device__ __forceinline__ void unpack(u32x4_t v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = as_f32(v[i] << 16);
      f[2 * i + 1] = as_f32(v[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ u32x4_t pack(const float* f) {
    // a vector conversion of the PAIR is one v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN); two scalar casts + shift + or
    // were four VALU instructions per pair — a quarter of the BN+ReLU operand transform of the small-channel convolutions
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{f[2 * i], f[2 * i + 1]}, bf16x2_t));
    return v;
  }
};
template <> struct Vec16<f16_t> {
  static __device__ __forceinline__ void unpack(u32x4_t v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = f16_bits_to_f32(v[i] & 0xffffu);
      f[2 * i + 1] = f16_bits_to_f32(v[i] >> 16);
    }
  }
  static __device__ __forceinline__ u32x4_t pack(const float* f) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{f[2 * i], f[2 * i + 1]}, f16x2_t));   // v_cvt_pk_f16_f32? (RNE)
    return v;
  }
};

// ---------------------------------------------------------------- BatchNorm apply (+ReLU) on one 16-byte operand vector
// v' = relu?(v * sc + sh), the transform every convolution applies to its operand while staging it (train-mode BatchNorm cannot
// be fused into the producing convolution).  On the small-channel decoder layers this VALU block is a real share of the kernel,
// so the 16-bit forms are written for the fewest instructions per PAIR of elements: 2 to unpack, ONE v_pk_fma_f32, ONE
// v_cvt_pk_*, ONE v_pk_max_i16 — ReLU on the packed result: a negative 16-bit float has its sign bit set, i.e. is a negative
// int16, so max(., 0) as int16 clears exactly the negatives (and -0); rounding is monotonic, so relu-then-round == round-then-relu.
// `relu` is a runtime flag: without it the floor is INT16_MIN and the max is the identity (no branch).
template <typename T> struct AffineRelu;
template <> struct AffineRelu<float> {
  static __device__ __forceinline__ u32x4_t run(u32x4_t v, const float* sc, const float* sh, bool relu) {
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float f = fmaf(as_f32(v[j]), sc[j], sh[j]);
      if (relu) f = fmaxf(f, 0.f);
      o[j] = as_u32(f);
    }
    return o;
  }
};
template <> struct AffineRelu<bf16_t> {
  static __device__ __forceinline__ u32x4_t run(u32x4_t v, const float* sc, const float* sh, bool relu) {
    const s16x2_t floor = relu ? s16x2_t{0, 0} : s16x2_t{(short)-32768, (short)-32768};
    u32x4_t o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x2_t x = {as_f32(v[i] << 16), as_f32(v[i] & 0xffff0000u)};
      x = __builtin_elementwise_fma(x, f32x2_t{sc[2 * i], sc[2 * i + 1]}, f32x2_t{sh[2 * i], sh[2 * i + 1]});
      const s16x2_t r = __builtin_bit_cast(s16x2_t, __builtin_convertvector(x, bf16x2_t));
      o[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(r, floor));
    }
    return o;
  }
};
template <> struct AffineRelu<f16_t> {
  static __device__ __forceinline__ u32x4_t run(u32x4_t v, const float* sc, const float* sh, bool relu) {
    const s16x2_t floor = relu ? s16x2_t{0, 0} : s16x2_t{(short)-32768, (short)-32768};
    u32x4_t o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x2_t x = __builtin_convertvector(as_f16x2(v[i]), f32x2_t);      // by-value helper: see the bit_cast note above
      x = __builtin_elementwise_fma(x, f32x2_t{sc[2 * i], sc[2 * i + 1]}, f32x2_t{sh[2 * i], sh[2 * i + 1]});
      const s16x2_t r = __builtin_bit_cast(s16x2_t, __builtin_convertvector(x, f16x2_t));
      o[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(r, floor));
    }
    return o;
  }
};

// The same transform with two scalar v_fma_f32 per pair instead of one v_pk_fma_f32 (experiment, VK_COL_DBG=32: the guide prices a
// packed f32 op beside MFMAs at +22 cycles against two scalar ones); the empty asm keeps the SLP vectoriser from re-packing them.
// Same arithmetic, same rounding: bit-identical results.
template <typename T> struct AffineReluScalar { static __device__ __forceinline__ u32x4_t run(u32x4_t v, const float* sc, const float* sh, bool relu) { return AffineRelu<T>::run(v, sc, sh, relu); } };
template <> struct AffineReluScalar<bf16_t> {
  static __device__ __forceinline__ u32x4_t run(u32x4_t v, const float* sc, const float* sh, bool relu) {
    const s16x2_t floor = relu ? s16x2_t{0, 0} : s16x2_t{(short)-32768, (short)-32768};
    u32x4_t o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float x0 = fmaf(as_f32(v[i] << 16), sc[2 * i], sh[2 * i]);
      asm volatile("" : "+v"(x0));
      float x1 = fmaf(as_f32(v[i] & 0xffff0000u), sc[2 * i + 1], sh[2 * i + 1]);
      asm volatile("" : "+v"(x1));
      const s16x2_t r = __builtin_bit_cast(s16x2_t, __builtin_convertvector(f32x2_t{x0, x1}, bf16x2_t));
      o[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(r, floor));
    }
    return o;
  }
};

// ---------------------------------------------------------------- MFMA wrappers (16x16 output tile)
// D[row = (lane>>4)*4 + reg][col = lane&15] += A[row][k] * B[k][col]
//   16-bit: lane holds A[lane&15][8*(lane>>4)+j], B[8*(lane>>4)+j][lane&15], j=0..7  (K = 32)
//   fp32  : lane holds A[lane&15][lane>>4],       B[lane>>4][lane&15]               (K = 4)
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ f32x4_t run(u32x4_t a, u32x4_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma<f16_t> {
  static __device__ __forceinline__ f32x4_t run(u32x4_t a, u32x4_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // a,b hold 4 consecutive k per lane; k is permuted consistently between A and B (the sum over k
  // does not care), so one ds_read_b128 feeds four 16x16x4 MFMAs.
  static __device__ __forceinline__ f32x4_t run(u32x4_t a, u32x4_t b, f32x4_t c) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f32(a[e]), as_f32(b[e]), c, 0, 0, 0);
    return c;
  }
};

// ---------------------------------------------------------------- buffer (SRD) loads with free bounds check
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4_t buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
}
static constexpr uint32_t kOOB = 0x80000000u;   // any tensor on this path is < 2 GiB (checked on the host)

// ---- LDS-DMA that hipcc does not see (r04).  `__builtin_amdgcn_raw_ptr_buffer_load_lds` is a pending LDS write in the compiler's
// waitcnt bookkeeping: every later LDS access with a known memory operand (the ds_read_b64_tr_b16 intrinsic among them) gets an
// `s_waitcnt vmcnt(0)` in front of it, which drains the whole prefetch before the first fragment read of the tile that runs beside it
// (wgrad_halo_batch_kernel with the builtin: +14 %).  Through inline assembly the instruction is invisible to that pass: the CALLER
// owns the completion — `lds_dma_wait_all()` before the barrier that publishes the bytes — and nothing else changes for the
// compiler's own counted waits (the hidden operations are the youngest of the wave when it waits for its register loads, so it can
// only over-wait; cdna_hip_programming.md section 5.7 item 1).  M0 is saved and restored inside the statement.
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
__device__ __forceinline__ i32x4_t make_rsrc_words(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  i32x4_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));       // stride 0, swizzle off
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
// 64 lanes x 16 bytes -> 1 KiB at LDS byte address lds_addr (wave-uniform); lanes with an out-of-range offset write zeros
__device__ __forceinline__ void lds_dma16_hidden(i32x4_t srd, uint32_t byte_off, uint32_t lds_addr) {
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(byte_off), "s"(srd), "s"(lds_addr)
               : "memory");
}
__device__ __forceinline__ void lds_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---------------------------------------------------------------- fast division by a runtime constant (n < 2^31)
struct FastDiv {
  uint32_t mul, shr, div;
};
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv d) { return d.div == 1 ? n : (__umulhi(n, d.mul) >> d.shr); }

// ---------------------------------------------------------------- wave / block reductions
// value of another lane through a DPP move (VALU rate; __shfl_xor is a ds_bpermute, i.e. an LDS-pipe instruction)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// sum over the 16 lanes of each row, left in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);     // row_half_mirror
  v += dpp_f<0x140>(v);     // row_mirror
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += dpp_f<0x142, 0xA>(v);      // row_bcast:15 -> rows 1 and 3 add the row before them
  v += dpp_f<0x143, 0xC>(v);      // row_bcast:31 -> rows 2 and 3 add rows 0+1: lane 63 holds the wave's sum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace vk

// host-side helpers -----------------------------------------------------------------------------
namespace vkh {
inline vk::FastDiv make_fastdiv(uint32_t d) {
  vk::FastDiv f;
  f.div = d;
  if (d <= 1) { f.mul = 0; f.shr = 0; f.div = 1; return f; }
  uint32_t lg = 0;
  while ((1u << lg) < d) ++lg;
  uint32_t p = 31 + lg;
  f.mul = (uint32_t)(((1ull << p) + d - 1) / d);
  f.shr = p - 32;
  return f;
}
void set_error(const char* fmt, ...);
int reserved_cus();          // vk_set_reserved_cus: compute units to leave to a concurrent communication kernel (0 = none)

// Optional per-launch timing with HIP events on the launch stream (vk_prof_enable / vk_prof_collect).
// tag: kernel family; flops / bytes: ALGORITHMIC work of this launch (DESIGN.md "roofline accounting").
struct ProfScope {
  int idx;
  hipStream_t st;
  hipEvent_t ev_end;
  ProfScope(const char* tag, hipStream_t stream, double flops, double bytes);
  ~ProfScope();
};
}  // namespace vkh

#define VK_CHECK_ARG(cond, ...)                \
  do {                                         \
    if (!(cond)) {                             \
      vkh::set_error(__VA_ARGS__);             \
      return VK_ERR_ARG;                       \
    }                                          \
  } while (0)

#define VK_CHECK_HIP(expr)                                                              \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      vkh::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return (int)e_;                                                                   \
    }                                                                                   \
  } while (0)

namespace vk {
// ---------------------------------------------------------------- batched weight gradients (wgrad_halo.hip; used by engine.hip)
// device table layout inside `tables` (VK_WGRAD_BATCH_TABLE_BYTES): [layers][segments][wg_first][tile reduce entries]
struct WgradBatchPlan {
  int nwg = 0, nsegs = 0, ntred = 0, nlayers = 0;
  size_t off_layers = 0, off_segs = 0, off_wgfirst = 0, off_tred = 0, slab_need = 0;
  double flops = 0.0, bytes = 0.0;
};
bool wgrad_batch_supports(const vk_conv_desc* d);
int wgrad_batch_build(const vk_conv_desc* descs, const void* const* dz, float* const* dw, int n, int target_blocks, void* tables,
                      size_t tables_bytes, WgradBatchPlan* plan);
int wgrad_batch_launch(vk_dtype dt, const WgradBatchPlan& plan, const void* tables, void* slab, size_t slab_bytes, hipStream_t st);
}  // namespace vk
