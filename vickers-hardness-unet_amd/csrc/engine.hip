// The U-Net plan: smp.Unet("resnet34", in_channels=3, classes=1) as a static schedule of the HIP
// kernels in this directory.  Mirrors the reference composition: train.py:372-378 (constructor
// arguments), train.py:436 (forward), :438 (BCE + Dice), :443/:448 (backward); topology restated
// from smp.  Host-only code except for the small weight-packing kernels.
//
// Memory model: the caller owns four flat buffers (fp32 params, fp32 grads, fp32 BN running stats,
// int64 num_batches_tracked) plus one workspace; this file only computes offsets into them.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "vk_common.h"

namespace vkh {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- event profiler -----------------------------------------------------------------------------
struct ProfRec {
  std::string tag;
  hipEvent_t a, b;
  double flops, bytes;
};
// guarded by g_prof_mu: the header promises re-entrant entry points (several host threads, one stream each)
static std::mutex g_prof_mu;
static std::atomic<bool> g_prof_on{false};
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t prof_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

ProfScope::ProfScope(const char* tag, hipStream_t stream, double flops, double bytes) : idx(-1), st(stream), ev_end(nullptr) {
  if (!g_prof_on.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_recs.size() >= 200000) return;
  ProfRec r{tag, prof_event(), prof_event(), flops, bytes};
  (void)hipEventRecord(r.a, st);
  idx = (int)g_recs.size();
  ev_end = r.b;                       // the record vector may be reallocated by another thread before the destructor runs
  g_recs.push_back(r);
}
ProfScope::~ProfScope() {
  if (idx >= 0) (void)hipEventRecord(ev_end, st);
}
}  // namespace vkh

extern "C" int vk_prof_enable(int on) {
  vkh::g_prof_on.store(on != 0);
  return VK_OK;
}

extern "C" int vk_prof_collect(char* buf, size_t buflen) {
  using namespace vkh;
  struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (ProfRec& r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      Agg& a = agg[r.tag];
      a.n += 1; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
    }
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  size_t off = 0;
  for (auto& kv : agg) {
    int w = snprintf(buf + off, off < buflen ? buflen - off : 0, "%s %ld %.6f %.6e %.6e\n", kv.first.c_str(), kv.second.n,
                     kv.second.ms, kv.second.flops, kv.second.bytes);
    if (w < 0 || off + (size_t)w >= buflen) break;
    off += (size_t)w;
  }
  return (int)off;
}

namespace vkh {
static std::atomic<int> g_reserved_cus{0};
int reserved_cus() { return g_reserved_cus.load(std::memory_order_relaxed); }
}  // namespace vkh
extern "C" int vk_set_reserved_cus(int n) {
  if (n < 0) n = 0;
  if (n > 192) n = 192;
  return vkh::g_reserved_cus.exchange(n);
}

extern "C" int vk_version(void) { return VK_ABI_VERSION; }
extern "C" const char* vk_last_error_string(void) { return vkh::g_err; }

namespace vk {
unsigned long long* g_vk_stamp_buf = nullptr;     // diagnostic (-DVK_STAMP) builds: per-wave cycle stamps of the tile kernels
}
extern "C" int vk_debug_set_stamp_buffer(void* p) {
  vk::g_vk_stamp_buf = (unsigned long long*)p;
  return VK_OK;
}

namespace vk {

__global__ void k_probe() {}

// bare MFMA loop: eight independent accumulator tiles per wave, operands in registers (vk_probe_mfma_rate)
__global__ __launch_bounds__(256) void k_mfma_rate(int iters, float* sink) {
  f32x4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // pseudo-random bf16 operands in +-[0.25, 1) per lane (random signs and mantissas: the data-dependent power of a real GEMM, sums stay
  // bounded) — on constant operands the same loop runs 94 % of the 2.5 PFLOP/s
  auto rnd = [](uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (x & 0x807F807Fu) | 0x3E803E80u | ((x >> 7) & 0x01000100u);
  };
  const uint32_t seed = (blockIdx.x * 256u + threadIdx.x) * 8u;
  const u32x4_t a = u32x4_t{rnd(seed), rnd(seed + 1), rnd(seed + 2), rnd(seed + 3)};
  const u32x4_t b = u32x4_t{rnd(seed + 4), rnd(seed + 5), rnd(seed + 6), rnd(seed + 7)};
  // inline assembly on fixed VGPR accumulators: the builtin form lets the register allocator rotate the loop-carried tiles through
  // AGPRs (a dozen v_accvgpr moves per iteration: the loop then measures those, 1.24 PFLOP/s)
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  // the MFMAs above are inline assembly: hipcc pads no hazard for them, and the first read of acc[7] below sits 1 wait state behind
  // its MFMA on the loop-exit edge (tools/mfma_hazard_audit.py, r04) — 8 states are required before a VALU read of a 16x16x32 result
  asm volatile("s_nop 7\n\ts_nop 1" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (t == 123.456f) sink[0] = t;                           // keeps the loop alive
}

// vk_debug_hold_cus: spin until a wall-clock deadline (100 MHz s_memrealtime), optionally streaming loads meanwhile
__global__ void k_hold_cus(unsigned long long ticks, const u32x4_t* __restrict__ traffic, size_t traffic_vecs, float* sink) {
  extern __shared__ char hold_lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0;
  size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) & (traffic_vecs ? traffic_vecs - 1 : 0);
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    if (traffic) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4_t v = __builtin_nontemporal_load(traffic + i);
        acc ^= v[0] ^ v[3];
        i = (i + (size_t)gridDim.x * blockDim.x) & (traffic_vecs - 1);
      }
    } else {
      __builtin_amdgcn_s_sleep(8);
    }
  }
  if (acc == 0x9E3779B9u && threadIdx.x == 0) {       // keeps the loads alive; LDS is declared, never needed
    hold_lds[0] = 1;
    sink[0] = (float)hold_lds[0];
  }
}

// ---------------------------------------------------------------- weight packing kernels
struct PackEntry {
  int64_t src, dst;     // element offsets: flat params / dgrad arena
  int K, RS, C, pad_;
};

// dgrad pack: arena[dst + (c*RS + t)*K + k] = T(params[src + (k*RS + t)*C + c])   (LDS-tiled transpose)
template <typename T>
__global__ __launch_bounds__(256) void k_pack_dgrad(const PackEntry* __restrict__ tab, const float* __restrict__ params,
                                                    T* __restrict__ arena) {
  const PackEntry e = tab[blockIdx.y];
  __shared__ float tl[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int kt_n = (e.K + 31) / 32, ct_n = (e.C + 31) / 32;
  const int ntile = kt_n * ct_n * e.RS;
  for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int t = tile % e.RS;
    const int rem = tile / e.RS;
    const int ct = rem % ct_n, kt = rem / ct_n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = kt * 32 + ty + 8 * i, c = ct * 32 + tx;
      tl[ty + 8 * i][tx] = (k < e.K && c < e.C) ? params[e.src + ((int64_t)k * e.RS + t) * e.C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = ct * 32 + ty + 8 * i, k = kt * 32 + tx;
      if (k < e.K && c < e.C) st1(arena + e.dst + ((int64_t)c * e.RS + t) * e.K + k, tl[tx][ty + 8 * i]);
    }
    __syncthreads();
  }
}

// halo pack of one 3x3 layer straight from the fp32 master weights [K][9][C] (layout: vk_halo_pack in vk_unet.h).
//   transposed == 0: forward weights, rows = K, reduction = C
//   transposed == 1: data-gradient weights (row = input channel c, reduction over k): element (c, tap, k) = params[(k*9 + tap)*C + c]
struct HaloPackEntry {
  int64_t src;          // element offset in the flat parameters
  int64_t dst;          // byte offset in the workspace
  int rows, red, transposed, pad_;
};

template <typename T>
__global__ __launch_bounds__(256) void k_pack_halo_tab(const HaloPackEntry* __restrict__ tab, const float* __restrict__ params, char* __restrict__ ws) {
  constexpr int VE = ElemTraits<T>::kVec, CK = 64 / ElemTraits<T>::kBytes;
  const HaloPackEntry e = tab[blockIdx.y];
  const float* src = params + e.src;
  u32x4_t* dst = reinterpret_cast<u32x4_t*>(ws + e.dst);
  const int rows = e.rows, red = e.red;
  const long total = (long)rows * 9 * red / VE;
  for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < total; v += (long)gridDim.x * blockDim.x) {
    int pos, k, tap, cc;
    if (!e.transposed) {            // consecutive threads -> consecutive reduction channels (contiguous in params)
      pos = (int)(v & 3);
      long t = v >> 2;
      k = (int)(t % rows); t /= rows;
      tap = (int)(t % 9); cc = (int)(t / 9);
    } else {                        // consecutive threads -> consecutive rows (contiguous in params)
      k = (int)(v % rows);
      long t = v / rows;
      pos = (int)(t & 3); t >>= 2;
      tap = (int)(t % 9); cc = (int)(t / 9);
    }
    const int j = pos ^ (((k >> 2) & 1) << 1);
    const int r0 = cc * CK + j * VE;
    float f[VE];
    if (!e.transposed) {
      // VE consecutive reduction channels: whole 16-byte vectors (parameter offsets are multiples of 4 floats, red and r0 of VE)
      const f32x4_t* s4 = reinterpret_cast<const f32x4_t*>(src + ((long)k * 9 + tap) * red + r0);
#pragma unroll
      for (int q = 0; q < VE / 4; ++q) {
        const f32x4_t t = s4[q];
#pragma unroll
        for (int i = 0; i < 4; ++i) f[4 * q + i] = t[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < VE; ++i) f[i] = src[((long)(r0 + i) * 9 + tap) * rows + k];
    }
    dst[(((long)cc * 9 + tap) * rows + k) * 4 + pos] = Vec16<T>::pack(f);
  }
}

template <typename T>
__global__ void k_cast_flat(size_t n, const float* __restrict__ src, T* __restrict__ dst) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st1(dst + i, src[i]);
}

// 16-bit working copy of the layers that read PLAIN weights (1x1 shortcuts, the C = 16 layer): the ranges of the flat buffer they
// occupy — 0.17 M of the 24.4 M parameters; every other layer reads a halo / stem / data-gradient pack built straight from fp32
struct CastRange {
  int64_t begin, count;
};
template <typename T>
__global__ void k_cast_ranges(const CastRange* __restrict__ tab, const float* __restrict__ src, T* __restrict__ dst) {
  const CastRange r = tab[blockIdx.y];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < r.count; i += (int64_t)gridDim.x * blockDim.x)
    st1(dst + r.begin + i, src[r.begin + i]);
}

// stem pack: wp[k][r][s*4 + c] = w[k][r][s][c] (s < 7, c < 3), zero elsewhere;  w is KRSC [64][7][7][3]
template <typename T>
__global__ void k_pack_stem(const float* __restrict__ w, T* __restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 64 * 7 * 32) return;
  const int j = i % 32, r = (i / 32) % 7, k = i / (32 * 7);
  const int s = j >> 2, c = j & 3;
  const float v = (s < 7 && c < 3) ? w[((k * 7 + r) * 7 + s) * 3 + c] : 0.f;
  st1(wp + i, v);
}

struct BnEvalEntry {
  int64_t g_off, b_off, rm_off, rv_off, out_off;   // out_off: scale at out_off, shift at out_off + C
  int C, pad_;
};
__global__ void k_bn_eval_all(const BnEvalEntry* __restrict__ tab, const float* __restrict__ params, const float* __restrict__ bufs,
                              float* __restrict__ arena, float eps) {
  const BnEvalEntry e = tab[blockIdx.x];
  for (int c = threadIdx.x; c < e.C; c += blockDim.x) {
    const float invstd = (float)(1.0 / sqrt((double)bufs[e.rv_off + c] + (double)eps));
    const float sc = params[e.g_off + c] * invstd;
    arena[e.out_off + c] = sc;
    arena[e.out_off + e.C + c] = params[e.b_off + c] - bufs[e.rm_off + c] * sc;
  }
}

__global__ void k_inc_i64(int n, int64_t* p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += 1;
}

// ---------------------------------------------------------------- plan structures
struct ConvL {
  std::string name;       // state-dict prefix; weight key = name + ".weight"
  int Cin, K, R, stride, pad;
  int64_t w_off;          // flat params (KRSC)
  int64_t wd_off;         // dgrad arena offset (-1: none)
  bool halo_fwd = false;  // forward runs on the 3x3 tile kernels and reads the halo pack of its weights
  bool halo_dg = false;   // same for the data gradient (the dgrad arena slot then holds the pack instead of plain [C][RS][K])
  int bn;                 // following BN (-1 for the head)
  int Hin, Hout;          // spatial sizes: rows ...
  int Win, Wout;          // ... and columns (r04: H != W allowed, both divisible by 32)
  // workspace
  char* z = nullptr;      // raw conv output [N][Hout][Hout][K]
  char* g = nullptr;      // gradient wrt activated output, overwritten in place by gradient wrt z
};

struct BnL {
  std::string name;
  int C;
  int64_t g_off, b_off, rm_off, rv_off;
  int idx;
  double count;           // N*H*W
  // workspace
  double* stats = nullptr;      // [2C] forward sums
  double* bsums = nullptr;      // [2C] backward sums
  float* scale = nullptr;       // [C]
  float* shift = nullptr;
  float* mean = nullptr;
  float* invstd = nullptr;
  float* coef = nullptr;        // [3C]
  int64_t arena_off = 0;        // offset of scale in the float arena
};

struct BlockL {
  int conv1, conv2, convd;      // convd = -1 for identity shortcut
  int Cin, C, stride, Hin, Hout, Win, Wout;
  bool in_has_grad_first;       // gradient buffer of the block input was already written (skip feature)
  char* out = nullptr;
  char* gout = nullptr;
};

struct DecL {
  int conv1, conv2;
  int Cup, Cskip, Cout, H, W;   // H x W = output resolution
};

struct Act {                     // a (possibly virtual) activated tensor
  const void* ptr;
  int C;
  const float* scale;
  const float* shift;
  int relu;
};

}  // namespace vk

using namespace vk;

constexpr int kMaxStages = 16;
constexpr int kMaxBatch = 40;                       // layers in one batched weight-gradient launch (the model has 35 of the class)
constexpr size_t kEngineSlabBytes = 176u << 20;     // slab workspace of a training plan: partial tiles of a whole-backward batch (~950 x 147 KB)

struct vk_unet {
  vk_unet_config cfg;
  int eb;                        // element bytes of the activation dtype
  std::vector<vk_tensor_info> infos;
  std::vector<ConvL> convs;
  std::vector<BnL> bns;
  std::vector<BlockL> blocks;
  std::vector<DecL> decs;
  int stem_conv = -1, head_conv = -1;
  int layer_last_block[4];
  int64_t n_params = 0, n_bufs = 0, n_dgrad = 0;
  int64_t head_w_off = 0, head_b_off = 0;
  std::vector<std::pair<int64_t, int64_t>> buckets;
  // workspace layout
  size_t ws_bytes = 0;
  size_t off_x4 = 0, off_pool = 0, off_argmax = 0, off_gpool = 0, off_dup = 0, off_dlogits = 0, off_loss_sums = 0;
  size_t off_stats = 0, off_bsums = 0, stats_bytes = 0, off_farena = 0, farena_floats = 0, off_wf = 0, off_wd = 0, off_wstem = 0;
  size_t off_tab_pack = 0, off_tab_bn = 0, off_wslab = 0, off_wh = 0, off_tab_halo = 0, off_splitk = 0, splitk_bytes = 0;
  std::vector<HaloPackEntry> halo_tab;
  std::vector<size_t> off_z, off_g, off_out, off_gout;
  std::vector<char> g_prereduced;      // per conv: its gradient buffer already holds masked g + sums (vk_bnr fusion)
  std::vector<char> tail_prereduced;   // per block: gout already holds g = gout * [out > 0] and bn2's sums are complete (vk_bnr.mask fusion)
  // bound pointers
  float* params = nullptr;
  float* grads = nullptr;
  float* bufs = nullptr;
  int64_t* nbt = nullptr;
  char* ws = nullptr;
  bool bound = false;
  // Weight gradients only feed the optimizer, so they run on a second (library-owned) stream beside the chain
  // dgrad -> BN backward -> dgrad ... of the caller's stream: fork event after the layer's dz is final, join at the end of every
  // backward stage (before the caller may all-reduce / read that stage's gradient bucket).  Opt-in (vk_unet_set_side_stream or
  // VK_SIDE_STREAM=1): the step gets 1.3 % faster, but kernels that share the chip each run longer, which makes every per-kernel
  // duration (rocprofv3, bench.py's roofline) a statement about the overlap instead of about the kernel.
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool side_dirty = false;
  // weight gradients of the 64 x 64 tile class are collected per backward stage and run as ONE launch (vk_conv_wgrad_batch)
  struct WItem {
    vk_conv_desc d;
    const void* dz;
    float* dw;
  };
  struct WPlan {
    bool built = false;
    int target = 0, n = 0;
    vk::WgradBatchPlan plan;
  };
  std::vector<WItem> pending;
  WPlan wplans[kMaxStages][2];      // [stage][0: every CU, 1: with CUs reserved for a collective]
  size_t off_tab_wbatch = 0;
  std::vector<PackEntry> pack_tab;
  std::vector<CastRange> cast_tab;
  size_t off_tab_cast = 0;
  std::vector<BnEvalEntry> bn_tab;
  std::map<std::string, std::pair<void*, std::vector<int>>> debug;

  int N() const { return cfg.N; }
};

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

void add_info(vk_unet* h, const std::string& name, int kind, std::vector<int> dims, int64_t offset, int64_t numel) {
  vk_tensor_info ti;
  memset(&ti, 0, sizeof(ti));
  snprintf(ti.name, sizeof(ti.name), "%s", name.c_str());
  ti.kind = kind;
  ti.ndim = (int)dims.size();
  for (size_t i = 0; i < dims.size() && i < 4; ++i) ti.dims[i] = dims[i];
  ti.offset = offset;
  ti.numel = numel;
  h->infos.push_back(ti);
}

int add_conv(vk_unet* h, const std::string& name, int Cin, int K, int R, int stride, int pad, int Hin, int Win, bool bias = false) {
  ConvL c;
  c.name = name;
  c.Cin = Cin; c.K = K; c.R = R; c.stride = stride; c.pad = pad;
  c.Hin = Hin; c.Win = Win;
  c.Hout = (Hin + 2 * pad - R) / stride + 1;
  c.Wout = (Win + 2 * pad - R) / stride + 1;
  c.w_off = h->n_params;
  c.wd_off = -1;
  c.bn = -1;
  const int64_t numel = (int64_t)K * Cin * R * R;
  add_info(h, name + ".weight", 0, {K, Cin, R, R}, c.w_off, numel);
  h->n_params += (int64_t)align_up((size_t)numel, 4);
  if (bias) {
    add_info(h, name + ".bias", 1, {K}, h->n_params, K);
    h->n_params += 4;
  }
  h->convs.push_back(c);
  return (int)h->convs.size() - 1;
}

int add_bn(vk_unet* h, const std::string& name, int C, double count) {
  BnL b;
  b.name = name;
  b.C = C;
  b.count = count;
  b.idx = (int)h->bns.size();
  b.g_off = h->n_params;
  add_info(h, name + ".weight", 1, {C}, b.g_off, C);
  h->n_params += C;
  b.b_off = h->n_params;
  add_info(h, name + ".bias", 1, {C}, b.b_off, C);
  h->n_params += C;
  b.rm_off = h->n_bufs;
  add_info(h, name + ".running_mean", 2, {C}, b.rm_off, C);
  h->n_bufs += C;
  b.rv_off = h->n_bufs;
  add_info(h, name + ".running_var", 2, {C}, b.rv_off, C);
  h->n_bufs += C;
  add_info(h, name + ".num_batches_tracked", 3, {}, b.idx, 1);
  h->bns.push_back(b);
  return b.idx;
}

void build_topology(vk_unet* h) {
  const int S = h->cfg.size, SW = h->cfg.width, N = h->cfg.N;
  auto cnt = [&](int Hres, int Wres) { return (double)N * Hres * Wres; };
  // --- encoder stem
  h->stem_conv = add_conv(h, "encoder.conv1", 3, 64, 7, 2, 3, S, SW);
  h->convs[h->stem_conv].bn = add_bn(h, "encoder.bn1", 64, cnt(S / 2, SW / 2));
  // --- encoder layers
  const int nblocks[4] = {3, 4, 6, 3};
  const int planes[4] = {64, 128, 256, 512};
  int inpl = 64, res = S / 4, resw = SW / 4;
  std::vector<int64_t> layer_start(4), layer3_mid(1);
  std::vector<int64_t> block_start;
  for (int L = 0; L < 4; ++L) {
    layer_start[L] = h->n_params;
    for (int b = 0; b < nblocks[L]; ++b) {
      block_start.push_back(h->n_params);
      const int stride = (b == 0 && L > 0) ? 2 : 1;
      const std::string pre = "encoder.layer" + std::to_string(L + 1) + "." + std::to_string(b);
      BlockL blk;
      blk.Cin = inpl; blk.C = planes[L]; blk.stride = stride; blk.Hin = res; blk.Hout = res / stride; blk.Win = resw; blk.Wout = resw / stride;
      blk.conv1 = add_conv(h, pre + ".conv1", inpl, planes[L], 3, stride, 1, res, resw);
      h->convs[blk.conv1].bn = add_bn(h, pre + ".bn1", planes[L], cnt(blk.Hout, blk.Wout));
      blk.conv2 = add_conv(h, pre + ".conv2", planes[L], planes[L], 3, 1, 1, blk.Hout, blk.Wout);
      h->convs[blk.conv2].bn = add_bn(h, pre + ".bn2", planes[L], cnt(blk.Hout, blk.Wout));
      blk.convd = -1;
      if (stride != 1 || inpl != planes[L]) {
        blk.convd = add_conv(h, pre + ".downsample.0", inpl, planes[L], 1, stride, 0, res, resw);
        h->convs[blk.convd].bn = add_bn(h, pre + ".downsample.1", planes[L], cnt(blk.Hout, blk.Wout));
      }
      // the input of block 0 of layers 2-4 is a skip feature whose gradient the decoder wrote first
      blk.in_has_grad_first = (b == 0 && L > 0);
      h->blocks.push_back(blk);
      inpl = planes[L];
      res = blk.Hout;
      resw = blk.Wout;
    }
    h->layer_last_block[L] = (int)h->blocks.size() - 1;
  }
  // --- decoder
  const int dec_in[5] = {512, 256, 128, 64, 32};
  const int dec_skip[5] = {256, 128, 64, 64, 0};
  const int dec_out[5] = {256, 128, 64, 32, 16};
  std::vector<int64_t> dec_start(5);
  int dres = S / 32, dresw = SW / 32;
  for (int i = 0; i < 5; ++i) {
    dec_start[i] = h->n_params;
    dres *= 2;
    dresw *= 2;
    const std::string pre = "decoder.blocks." + std::to_string(i);
    DecL d;
    d.Cup = dec_in[i]; d.Cskip = dec_skip[i]; d.Cout = dec_out[i]; d.H = dres; d.W = dresw;
    d.conv1 = add_conv(h, pre + ".conv1.0", dec_in[i] + dec_skip[i], dec_out[i], 3, 1, 1, dres, dresw);
    h->convs[d.conv1].bn = add_bn(h, pre + ".conv1.1", dec_out[i], cnt(dres, dresw));
    d.conv2 = add_conv(h, pre + ".conv2.0", dec_out[i], dec_out[i], 3, 1, 1, dres, dresw);
    h->convs[d.conv2].bn = add_bn(h, pre + ".conv2.1", dec_out[i], cnt(dres, dresw));
    h->decs.push_back(d);
  }
  // --- head
  const int64_t head_start = h->n_params;
  (void)head_start;
  h->head_conv = add_conv(h, "segmentation_head.0", 16, 1, 3, 1, 1, S, SW, /*bias=*/true);
  h->head_w_off = h->convs[h->head_conv].w_off;
  h->head_b_off = h->head_w_off + 144;
  h->n_params = (int64_t)align_up((size_t)h->n_params, 64);
  // --- dgrad arena offsets
  for (size_t i = 0; i < h->convs.size(); ++i) {
    if ((int)i == h->stem_conv || (int)i == h->head_conv) continue;
    ConvL& c = h->convs[i];
    c.wd_off = h->n_dgrad;
    h->n_dgrad += (int64_t)align_up((size_t)c.K * c.Cin * c.R * c.R, 8);
  }
  // --- gradient buckets in backward completion order (stage i completes bucket i)
  const int64_t P = h->n_params;
  // block_start indices: layer1: 0..2, layer2: 3..6, layer3: 7..12, layer4: 13..15
  h->buckets.push_back({dec_start[2], P});                       // stage 0: head + dec4 + dec3 + dec2
  h->buckets.push_back({dec_start[1], dec_start[2]});            // 1: dec1
  h->buckets.push_back({dec_start[0], dec_start[1]});            // 2: dec0
  h->buckets.push_back({block_start[15], dec_start[0]});         // 3: layer4.2
  h->buckets.push_back({block_start[14], block_start[15]});      // 4: layer4.1
  h->buckets.push_back({block_start[13], block_start[14]});      // 5: layer4.0
  h->buckets.push_back({block_start[10], block_start[13]});      // 6: layer3.3-5
  h->buckets.push_back({block_start[7], block_start[10]});       // 7: layer3.0-2
  h->buckets.push_back({block_start[3], block_start[7]});        // 8: layer2
  h->buckets.push_back({0, block_start[3]});                     // 9: layer1 + stem
}

void layout_workspace(vk_unet* h) {
  const int N = h->cfg.N, S = h->cfg.size, SW = h->cfg.width, eb = h->eb;
  const bool tr = h->cfg.training != 0;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  h->off_x4 = take((size_t)N * S * SW * 4 * eb);
  h->g_prereduced.assign(h->convs.size(), 0);
  h->tail_prereduced.assign(h->blocks.size(), 0);
  h->off_z.resize(h->convs.size());
  h->off_g.assign(h->convs.size(), 0);
  for (size_t i = 0; i < h->convs.size(); ++i) {
    const ConvL& c = h->convs[i];
    if ((int)i == h->head_conv) { h->off_z[i] = 0; continue; }
    const size_t bytes = (size_t)N * c.Hout * c.Wout * c.K * eb;
    h->off_z[i] = take(bytes);
    if (tr) h->off_g[i] = take(bytes);
  }
  h->off_pool = take((size_t)N * (S / 4) * (SW / 4) * 64 * eb);
  h->off_argmax = take((size_t)N * (S / 4) * (SW / 4) * 64);
  if (tr) h->off_gpool = take((size_t)N * (S / 4) * (SW / 4) * 64 * eb);
  h->off_out.resize(h->blocks.size());
  h->off_gout.assign(h->blocks.size(), 0);
  for (size_t b = 0; b < h->blocks.size(); ++b) {
    const BlockL& k = h->blocks[b];
    const size_t bytes = (size_t)N * k.Hout * k.Wout * k.C * eb;
    h->off_out[b] = take(bytes);
    if (tr) h->off_gout[b] = take(bytes);
  }
  if (tr) {
    size_t mx = 0;
    for (const DecL& d : h->decs) mx = std::max(mx, (size_t)N * d.H * d.W * d.Cup * eb);
    h->off_dup = take(mx);
    h->off_dlogits = take((size_t)N * S * SW * 4);
  }
  h->off_loss_sums = take(8 * sizeof(double));
  // per-BN statistics (one zero region) and float arena
  size_t st = 0, fl = 0;
  for (BnL& b : h->bns) {
    st += 2 * (size_t)b.C * sizeof(double) * VK_STATS_REPLICAS;
    b.arena_off = (int64_t)fl;
    fl += 7 * (size_t)b.C;
  }
  h->stats_bytes = st;
  h->off_stats = take(st);
  h->off_bsums = take(st);
  h->farena_floats = fl;
  h->off_farena = take(fl * sizeof(float));
  h->off_wf = (eb == 4) ? 0 : take((size_t)h->n_params * eb);
  h->off_wd = tr ? take((size_t)h->n_dgrad * eb) : 0;
  h->off_wh = take((size_t)h->n_params * eb);                  // halo packs of the forward weights (same offsets as the flat copy)
  h->off_tab_halo = take(2 * h->convs.size() * sizeof(HaloPackEntry));
  h->off_wstem = take(64 * 7 * 32 * eb);
  h->off_wslab = tr ? take(kEngineSlabBytes) : 0;
  h->off_tab_wbatch = tr ? take((size_t)kMaxStages * 2 * VK_WGRAD_BATCH_TABLE_BYTES) : 0;
  // eval plans with few tiles per layer (batch-1 inference) split the channel reduction: scratch for the partial tiles
  h->splitk_bytes = (!tr && (size_t)N * S * SW <= 4u * 512 * 512) ? VK_SPLITK_WORKSPACE_BYTES : 0;
  h->off_splitk = h->splitk_bytes ? take(h->splitk_bytes) : 0;
  h->off_tab_pack = take(h->convs.size() * sizeof(PackEntry));
  h->off_tab_cast = take(h->convs.size() * sizeof(CastRange));
  h->off_tab_bn = take(h->bns.size() * sizeof(BnEvalEntry));
  h->ws_bytes = off;
}

void assign_pointers(vk_unet* h) {
  char* ws = h->ws;
  h->debug.clear();
  const int N = h->cfg.N;
  for (size_t i = 0; i < h->convs.size(); ++i) {
    ConvL& c = h->convs[i];
    if ((int)i == h->head_conv) continue;
    c.z = ws + h->off_z[i];
    c.g = h->cfg.training ? ws + h->off_g[i] : nullptr;
    h->debug["z:" + c.name] = {c.z, {N, c.Hout, c.Wout, c.K}};
    if (c.g) h->debug["g:" + c.name] = {c.g, {N, c.Hout, c.Wout, c.K}};
  }
  for (size_t b = 0; b < h->blocks.size(); ++b) {
    BlockL& k = h->blocks[b];
    k.out = ws + h->off_out[b];
    k.gout = h->cfg.training ? ws + h->off_gout[b] : nullptr;
    const std::string nm = h->convs[k.conv1].name.substr(0, h->convs[k.conv1].name.size() - 6);  // strip ".conv1"
    h->debug["out:" + nm] = {k.out, {N, k.Hout, k.Wout, k.C}};
    if (k.gout) h->debug["gout:" + nm] = {k.gout, {N, k.Hout, k.Wout, k.C}};
  }
  const int S = h->cfg.size, SW = h->cfg.width;
  h->debug["x4"] = {ws + h->off_x4, {N, S, SW, 4}};
  h->debug["pool"] = {ws + h->off_pool, {N, S / 4, SW / 4, 64}};
  if (h->cfg.training) {
    h->debug["gpool"] = {ws + h->off_gpool, {N, S / 4, SW / 4, 64}};
    h->debug["dlogits"] = {ws + h->off_dlogits, {N, S, SW, 1}};
  }
  double* sp = (double*)(ws + h->off_stats);
  double* bp = (double*)(ws + h->off_bsums);
  float* fa = (float*)(ws + h->off_farena);
  for (BnL& b : h->bns) {
    b.stats = sp;
    b.bsums = bp;
    sp += 2 * b.C * VK_STATS_REPLICAS;
    bp += 2 * b.C * VK_STATS_REPLICAS;
    float* f = fa + b.arena_off;
    b.scale = f;
    b.shift = f + b.C;
    b.mean = f + 2 * b.C;
    b.invstd = f + 3 * b.C;
    b.coef = f + 4 * b.C;
    h->debug["scale:" + b.name] = {b.scale, {b.C}};
    h->debug["shift:" + b.name] = {b.shift, {b.C}};
  }
}

const void* fwd_weights(const vk_unet* h, const ConvL& c) {
  if (c.halo_fwd) return h->ws + h->off_wh + (size_t)c.w_off * h->eb;
  if (h->eb == 4) return h->params + c.w_off;
  return h->ws + h->off_wf + (size_t)c.w_off * h->eb;
}
const void* dgrad_weights(const vk_unet* h, const ConvL& c) { return h->ws + h->off_wd + (size_t)c.wd_off * h->eb; }

vk_src to_src(const Act& a, int up = 0) {
  vk_src s;
  s.ptr = a.ptr; s.C = a.C; s.up = up; s.scale = a.scale; s.shift = a.shift; s.relu = a.relu;
  return s;
}
vk_src null_src() {
  vk_src s;
  memset(&s, 0, sizeof(s));
  return s;
}

vk_conv_desc conv_desc(const vk_unet* h, const ConvL& c, const vk_src& s0, const vk_src& s1) {
  vk_conv_desc d;
  d.dtype = h->cfg.dtype;
  d.N = h->cfg.N; d.H = c.Hin; d.W = c.Win; d.Ho = c.Hout; d.Wo = c.Wout;
  d.K = c.K; d.R = c.R; d.S = c.R; d.stride = c.stride; d.pad = c.pad; d.transposed = 0;
  d.src0 = s0; d.src1 = s1;
  return d;
}

Act bn_act(const vk_unet* h, const ConvL& c) {   // relu(bn(z)) as a virtual tensor
  const BnL& b = h->bns[c.bn];
  return Act{c.z, c.K, b.scale, b.shift, 1};
}

#define RET_IF(expr)            \
  do {                          \
    int rc_ = (expr);           \
    if (rc_ != VK_OK) return rc_; \
  } while (0)

int finalize_bn(vk_unet* h, BnL& b, int training, hipStream_t st) {
  if (!training) return VK_OK;   // eval affine was produced for all layers at the start of forward
  return vk_bn_finalize(b.C, 1, b.stats, b.count, h->params + b.g_off, h->params + b.b_off, h->bufs + b.rm_off,
                        h->bufs + b.rv_off, 1e-5f, 0.1f, b.scale, b.shift, b.mean, b.invstd, st);
}

int run_conv(vk_unet* h, ConvL& c, const vk_src& s0, const vk_src& s1, int training, hipStream_t st) {
  vk_conv_desc d = conv_desc(h, c, s0, s1);
  BnL& b = h->bns[c.bn];
  if (c.halo_fwd && !training && h->splitk_bytes) RET_IF(vk_conv_fwd_splitk(&d, fwd_weights(h, c), c.z, h->ws + h->off_splitk, h->splitk_bytes, st));
  else if (c.halo_fwd) RET_IF(vk_conv_fwd_packed(&d, fwd_weights(h, c), c.z, nullptr, 0, 0, training ? b.stats : nullptr, st));
  else RET_IF(vk_conv_fwd(&d, fwd_weights(h, c), c.z, nullptr, 0, 0, training ? b.stats : nullptr, st));
  return finalize_bn(h, b, training, st);
}

}  // namespace

// =================================================================================================
extern "C" int vk_has_gfx950_code(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, (const void*)vk::k_probe) == hipSuccess ? 1 : 0;
}

extern "C" int vk_probe_mfma_rate(int iters, int waves_per_simd, float* sink, double* flops_out, void* stream) {
  VK_CHECK_ARG(iters >= 1 && waves_per_simd >= 1 && waves_per_simd <= 8 && sink, "vk_probe_mfma_rate: bad argument");
  int dev = 0, cus = 0;
  VK_CHECK_HIP(hipGetDevice(&dev));
  VK_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int blocks = cus * waves_per_simd;                  // 256 threads = one wave per SIMD of a CU
  if (flops_out) *flops_out = (double)blocks * 4.0 * (double)iters * 8.0 * 2.0 * 16.0 * 16.0 * 32.0;
  hipLaunchKernelGGL(vk::k_mfma_rate, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, iters, sink);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_debug_hold_cus(int workgroups, int threads, int lds_bytes, int microseconds, const void* traffic, size_t traffic_bytes,
                                 float* sink, void* stream) {
  VK_CHECK_ARG(workgroups >= 1 && workgroups <= 4096 && threads >= 64 && threads <= 1024 && threads % 64 == 0, "vk_debug_hold_cus: bad grid");
  VK_CHECK_ARG(lds_bytes >= 0 && lds_bytes <= 163840 && microseconds >= 1 && microseconds <= 2000000 && sink, "vk_debug_hold_cus: bad argument");
  VK_CHECK_ARG(!traffic || (traffic_bytes >= 65536 && (traffic_bytes & (traffic_bytes - 1)) == 0 && ((uintptr_t)traffic & 15) == 0),
               "vk_debug_hold_cus: traffic buffer must be a 16-byte aligned power of two >= 64 KiB");
  if (lds_bytes > 65536)
    VK_CHECK_HIP(hipFuncSetAttribute((const void*)vk::k_hold_cus, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  hipLaunchKernelGGL(vk::k_hold_cus, dim3((unsigned)workgroups), dim3((unsigned)threads), (size_t)lds_bytes, (hipStream_t)stream,
                     (unsigned long long)microseconds * 100ull, (const vk::u32x4_t*)traffic, traffic ? traffic_bytes / 16 : (size_t)0, sink);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_unet_create(const vk_unet_config* cfg, vk_unet** out) {
  VK_CHECK_ARG(cfg && out, "vk_unet_create: null argument");
  const int width = cfg->width > 0 ? cfg->width : cfg->size;
  VK_CHECK_ARG(cfg->N >= 1 && cfg->size >= 32 && cfg->size % 32 == 0 && width >= 32 && width % 32 == 0,
               "Wrong input shape height=%d, width=%d: must be divisible by 32", cfg->size, width);
  VK_CHECK_ARG(cfg->dtype == VK_F32 || cfg->dtype == VK_BF16 || cfg->dtype == VK_F16, "vk_unet_create: bad dtype");
  vk_unet* h = new vk_unet();
  h->cfg = *cfg;
  h->cfg.width = width;
  h->eb = cfg->dtype == VK_F32 ? 4 : 2;
  build_topology(h);
  layout_workspace(h);
  *out = h;
  return VK_OK;
}

extern "C" void vk_unet_destroy(vk_unet* h) {
  if (!h) return;
  if (h->side) {
    (void)hipStreamSynchronize(h->side);
    (void)hipEventDestroy(h->ev_fork);
    (void)hipEventDestroy(h->ev_join);
    (void)hipStreamDestroy(h->side);
  }
  delete h;
}
extern "C" int vk_unet_num_tensors(const vk_unet* h) { return h ? (int)h->infos.size() : 0; }
extern "C" int vk_unet_tensor_info(const vk_unet* h, int index, vk_tensor_info* out) {
  VK_CHECK_ARG(h && out && index >= 0 && index < (int)h->infos.size(), "vk_unet_tensor_info: bad index %d", index);
  *out = h->infos[index];
  return VK_OK;
}
extern "C" int64_t vk_unet_param_numel(const vk_unet* h) { return h ? h->n_params : 0; }
extern "C" int64_t vk_unet_buffer_numel(const vk_unet* h) { return h ? h->n_bufs : 0; }
extern "C" int64_t vk_unet_workspace_bytes(const vk_unet* h) { return h ? (int64_t)h->ws_bytes : 0; }
extern "C" int vk_unet_num_buckets(const vk_unet* h) { return h ? (int)h->buckets.size() : 0; }
extern "C" int vk_unet_bucket_range(const vk_unet* h, int bucket, int64_t* b, int64_t* e) {
  VK_CHECK_ARG(h && b && e && bucket >= 0 && bucket < (int)h->buckets.size(), "vk_unet_bucket_range: bad bucket %d", bucket);
  *b = h->buckets[bucket].first;
  *e = h->buckets[bucket].second;
  return VK_OK;
}

extern "C" int vk_unet_bind(vk_unet* h, float* params, float* grads, float* bn_buffers, int64_t* nbt, void* workspace,
                            size_t workspace_bytes) {
  VK_CHECK_ARG(h && params && bn_buffers && nbt && workspace, "vk_unet_bind: null argument");
  VK_CHECK_ARG(!h->cfg.training || grads, "vk_unet_bind: a training plan needs a gradient buffer");
  VK_CHECK_ARG(workspace_bytes >= h->ws_bytes, "vk_unet_bind: workspace too small (%zu < %zu)", workspace_bytes, h->ws_bytes);
  VK_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)params & 15) == 0, "vk_unet_bind: buffers must be 256-byte (workspace) / 16-byte (params) aligned");
  h->params = params; h->grads = grads; h->bufs = bn_buffers; h->nbt = nbt; h->ws = (char*)workspace;
  for (auto& st_ : h->wplans) { st_[0].built = false; st_[1].built = false; }      // their tables hold the old pointers
  h->pending.clear();
  assign_pointers(h);
  // device tables
  // which layers run on the 3x3 tile kernels (halo pack weights): ask the kernels' own predicate with the real descriptors
  h->halo_tab.clear();
  {
    auto plain = [&](int C) { vk_src s = null_src(); s.ptr = h->ws; s.C = C; return s; };
    auto fwd_uses = [&](ConvL& c, int C0, int C1, int up) {
      vk_src s0 = plain(C0), s1 = C1 ? plain(C1) : null_src();
      s0.up = up;
      vk_conv_desc d = conv_desc(h, c, s0, s1);
      return vk_conv_uses_halo_pack(&d) != 0;
    };
    for (size_t i = 0; i < h->convs.size(); ++i) {
      ConvL& c = h->convs[i];
      c.halo_fwd = c.halo_dg = false;
      if ((int)i == h->stem_conv || (int)i == h->head_conv) continue;
      c.halo_fwd = fwd_uses(c, c.Cin, 0, 0);
    }
    for (DecL& d : h->decs) {          // decoder conv1: upsampled + skip operand
      ConvL& c = h->convs[d.conv1];
      c.halo_fwd = fwd_uses(c, d.Cup, d.Cskip, 1);
    }
    for (size_t i = 0; i < h->convs.size(); ++i) {
      ConvL& c = h->convs[i];
      if (c.wd_off >= 0 && h->cfg.training) {
        vk_conv_desc d;
        d.dtype = h->cfg.dtype;
        d.N = h->cfg.N; d.H = c.Hout; d.W = c.Wout; d.Ho = c.Hin; d.Wo = c.Win;
        d.K = c.Cin; d.R = c.R; d.S = c.R; d.stride = c.stride; d.pad = c.pad; d.transposed = 1;
        d.src0 = plain(c.K);
        d.src1 = null_src();
        c.halo_dg = vk_conv_uses_halo_pack(&d) != 0;
      }
      HaloPackEntry e;
      e.src = c.w_off; e.pad_ = 0;
      if (c.halo_fwd) {
        e.dst = (int64_t)(h->off_wh + (size_t)c.w_off * h->eb); e.rows = c.K; e.red = c.Cin; e.transposed = 0;
        h->halo_tab.push_back(e);
      }
      if (c.halo_dg) {
        e.dst = (int64_t)(h->off_wd + (size_t)c.wd_off * h->eb); e.rows = c.Cin; e.red = c.K; e.transposed = 1;
        h->halo_tab.push_back(e);
      }
    }
  }
  h->pack_tab.clear();
  for (size_t i = 0; i < h->convs.size(); ++i) {
    const ConvL& c = h->convs[i];
    if (c.wd_off < 0 || c.halo_dg) continue;
    PackEntry e;
    e.src = c.w_off; e.dst = c.wd_off; e.K = c.K; e.RS = c.R * c.R; e.C = c.Cin; e.pad_ = 0;
    h->pack_tab.push_back(e);
  }
  h->cast_tab.clear();
  for (size_t i = 0; i < h->convs.size(); ++i) {
    const ConvL& c = h->convs[i];
    if ((int)i == h->stem_conv || (int)i == h->head_conv || c.halo_fwd) continue;     // stem: own pack; head: fp32; tile layers: halo pack
    h->cast_tab.push_back(CastRange{c.w_off, (int64_t)c.K * c.Cin * c.R * c.R});
  }
  h->bn_tab.clear();
  for (const BnL& b : h->bns) {
    BnEvalEntry e;
    e.g_off = b.g_off; e.b_off = b.b_off; e.rm_off = b.rm_off; e.rv_off = b.rv_off; e.out_off = b.arena_off; e.C = b.C; e.pad_ = 0;
    h->bn_tab.push_back(e);
  }
  if (!h->pack_tab.empty())
    VK_CHECK_HIP(hipMemcpy(h->ws + h->off_tab_pack, h->pack_tab.data(), h->pack_tab.size() * sizeof(PackEntry), hipMemcpyHostToDevice));
  if (!h->halo_tab.empty())
    VK_CHECK_HIP(hipMemcpy(h->ws + h->off_tab_halo, h->halo_tab.data(), h->halo_tab.size() * sizeof(HaloPackEntry), hipMemcpyHostToDevice));
  if (!h->cast_tab.empty())
    VK_CHECK_HIP(hipMemcpy(h->ws + h->off_tab_cast, h->cast_tab.data(), h->cast_tab.size() * sizeof(CastRange), hipMemcpyHostToDevice));
  VK_CHECK_HIP(hipMemcpy(h->ws + h->off_tab_bn, h->bn_tab.data(), h->bn_tab.size() * sizeof(BnEvalEntry), hipMemcpyHostToDevice));
  if (h->cfg.training && !h->side && getenv("VK_SIDE_STREAM")) RET_IF(vk_unet_set_side_stream(h, 1));
  h->bound = true;
  return VK_OK;
}

template <typename T>
static int refresh_t(vk_unet* h, hipStream_t st) {
  vkh::ProfScope ps_("weights_repack", st, 0.0, (double)h->n_params * (4.0 + 2.0 * sizeof(T)));
  if (sizeof(T) != 4 && !h->cast_tab.empty())
    hipLaunchKernelGGL(k_cast_ranges<T>, dim3(16, (unsigned)h->cast_tab.size()), dim3(256), 0, st, (const CastRange*)(h->ws + h->off_tab_cast),
                       h->params, (T*)(h->ws + h->off_wf));
  if (h->cfg.training && !h->pack_tab.empty())
    hipLaunchKernelGGL(k_pack_dgrad<T>, dim3(64, (unsigned)h->pack_tab.size()), dim3(256), 0, st,
                       (const PackEntry*)(h->ws + h->off_tab_pack), h->params, (T*)(h->ws + h->off_wd));
  if (!h->halo_tab.empty())
    hipLaunchKernelGGL(k_pack_halo_tab<T>, dim3(32, (unsigned)h->halo_tab.size()), dim3(256), 0, st,
                       (const HaloPackEntry*)(h->ws + h->off_tab_halo), h->params, h->ws);
  hipLaunchKernelGGL(k_pack_stem<T>, dim3((64 * 7 * 32 + 255) / 256), dim3(256), 0, st,
                     h->params + h->convs[h->stem_conv].w_off, (T*)(h->ws + h->off_wstem));
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_unet_refresh_weights(vk_unet* h, void* stream) {
  VK_CHECK_ARG(h && h->bound, "vk_unet_refresh_weights: plan not bound");
  hipStream_t st = (hipStream_t)stream;
  switch (h->cfg.dtype) {
    case VK_F32: return refresh_t<float>(h, st);
    case VK_BF16: return refresh_t<bf16_t>(h, st);
    case VK_F16: return refresh_t<f16_t>(h, st);
  }
  return VK_ERR_ARG;
}

extern "C" int vk_unet_set_side_stream(vk_unet* h, int enable) {
  VK_CHECK_ARG(h, "vk_unet_set_side_stream: null plan");
  if (enable && !h->side && h->cfg.training) {
    VK_CHECK_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
    VK_CHECK_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    VK_CHECK_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  } else if (!enable && h->side) {
    VK_CHECK_HIP(hipStreamSynchronize(h->side));
    (void)hipEventDestroy(h->ev_fork);
    (void)hipEventDestroy(h->ev_join);
    (void)hipStreamDestroy(h->side);
    h->side = nullptr;
    h->ev_fork = h->ev_join = nullptr;
    h->side_dirty = false;
  }
  return VK_OK;
}

extern "C" int vk_unet_zero_grad(vk_unet* h, void* stream) {
  VK_CHECK_ARG(h && h->bound && h->grads, "vk_unet_zero_grad: no gradient buffer bound");
  VK_CHECK_HIP(hipMemsetAsync(h->grads, 0, (size_t)h->n_params * sizeof(float), (hipStream_t)stream));
  return VK_OK;
}

extern "C" int vk_unet_debug_tensor(const vk_unet* h, const char* name, void** ptr, int dims[4]) {
  VK_CHECK_ARG(h && h->bound && name && ptr && dims, "vk_unet_debug_tensor: bad argument");
  auto it = h->debug.find(name);
  VK_CHECK_ARG(it != h->debug.end(), "vk_unet_debug_tensor: unknown tensor '%s'", name);
  *ptr = it->second.first;
  for (int i = 0; i < 4; ++i) dims[i] = i < (int)it->second.second.size() ? it->second.second[i] : 1;
  return VK_OK;
}

// ------------------------------------------------------------------------------------------------ forward
extern "C" int vk_unet_forward(vk_unet* h, const float* x, float* logits, int training, void* stream) {
  VK_CHECK_ARG(h && h->bound && x && logits, "vk_unet_forward: plan not bound or null tensor");
  VK_CHECK_ARG(!training || h->cfg.training, "vk_unet_forward: training forward needs a training plan");
  hipStream_t st = (hipStream_t)stream;
  const int N = h->cfg.N, S = h->cfg.size, SW = h->cfg.width;
  const vk_dtype dt = h->cfg.dtype;
  if (training) {
    VK_CHECK_HIP(hipMemsetAsync(h->ws + h->off_stats, 0, h->stats_bytes, st));
  } else {
    hipLaunchKernelGGL(k_bn_eval_all, dim3((unsigned)h->bns.size()), dim3(128), 0, st, (const BnEvalEntry*)(h->ws + h->off_tab_bn),
                       h->params, h->bufs, (float*)(h->ws + h->off_farena), 1e-5f);
    VK_CHECK_HIP(hipGetLastError());
  }
  // stem
  void* x4 = h->ws + h->off_x4;
  RET_IF(vk_input_transform(dt, N, S, SW, x, x4, st));
  ConvL& stem = h->convs[h->stem_conv];
  BnL& sbn = h->bns[stem.bn];
  RET_IF(vk_stem_fwd(dt, N, S, SW, x4, h->ws + h->off_wstem, stem.z, training ? sbn.stats : nullptr, st));
  RET_IF(finalize_bn(h, sbn, training, st));
  void* pool = h->ws + h->off_pool;
  RET_IF(vk_bn_relu_maxpool(dt, N, S / 2, SW / 2, 64, stem.z, sbn.scale, sbn.shift, pool, (uint8_t*)(h->ws + h->off_argmax), st));
  // encoder blocks
  Act cur{pool, 64, nullptr, nullptr, 0};
  for (BlockL& k : h->blocks) {
    ConvL& c1 = h->convs[k.conv1];
    ConvL& c2 = h->convs[k.conv2];
    RET_IF(run_conv(h, c1, to_src(cur), null_src(), training, st));
    RET_IF(run_conv(h, c2, to_src(bn_act(h, c1)), null_src(), training, st));
    const BnL& b2 = h->bns[c2.bn];
    const size_t pixels = (size_t)N * k.Hout * k.Wout;
    if (k.convd >= 0) {
      ConvL& cd = h->convs[k.convd];
      RET_IF(run_conv(h, cd, to_src(cur), null_src(), training, st));
      const BnL& bd = h->bns[cd.bn];
      RET_IF(vk_bn_add_relu(dt, pixels, k.C, c2.z, b2.scale, b2.shift, cd.z, bd.scale, bd.shift, k.out, st));
    } else {
      RET_IF(vk_bn_add_relu(dt, pixels, k.C, c2.z, b2.scale, b2.shift, cur.ptr, nullptr, nullptr, k.out, st));
    }
    cur = Act{k.out, k.C, nullptr, nullptr, 0};
  }
  // decoder
  Act skips[5];
  skips[0] = Act{h->blocks[h->layer_last_block[2]].out, 256, nullptr, nullptr, 0};   // f4
  skips[1] = Act{h->blocks[h->layer_last_block[1]].out, 128, nullptr, nullptr, 0};   // f3
  skips[2] = Act{h->blocks[h->layer_last_block[0]].out, 64, nullptr, nullptr, 0};    // f2
  skips[3] = bn_act(h, stem);                                                         // f1 = relu(bn1(conv1 x))
  skips[4] = Act{nullptr, 0, nullptr, nullptr, 0};
  Act xd = cur;   // f5
  for (size_t i = 0; i < h->decs.size(); ++i) {
    DecL& d = h->decs[i];
    ConvL& c1 = h->convs[d.conv1];
    ConvL& c2 = h->convs[d.conv2];
    // inference, 16-bit, SMALL plans (single-image inference: launch-bound): the last block and the head as ONE kernel over an
    // overlapping tile (vk_dec4_tail_eval; the folded BatchNorm affines are constants, nothing forces the two 512^2 x 16 tensors through
    // HBM).  Measured (profiles/r03/eval_tail_fusion.log): batch 1 0.691 -> 0.671 ms, but batch 16 1.690 -> 1.734 ms — the fused
    // tile's redundant halo work and three barriers cost more than the 33 MB per image they save against the streaming kernels — so
    // large plans keep the three launches.  VK_NO_TAIL_FUSION=1 / VK_TAIL_FUSION=1: never / always
    const bool small_plan = (size_t)N * S * SW <= 4u * 512 * 512;
    if (!training && i + 1 == h->decs.size() && dt != VK_F32 && !d.Cskip && d.Cup == 32 && c1.K == 16 && c2.K == 16 && c1.halo_fwd &&
        !c2.halo_fwd && S % 16 == 0 && SW % 16 == 0 && !getenv("VK_NO_TAIL_FUSION") && (small_plan || getenv("VK_TAIL_FUSION"))) {
      const vk_src s0 = to_src(xd, 1);
      const BnL& b1 = h->bns[c1.bn];
      const BnL& b2 = h->bns[c2.bn];
      return vk_dec4_tail_eval(dt, N, S, SW, &s0, fwd_weights(h, c1), b1.scale, b1.shift, fwd_weights(h, c2), b2.scale, b2.shift,
                               h->params + h->head_w_off, h->params + h->head_b_off, logits, st);
    }
    RET_IF(run_conv(h, c1, to_src(xd, 1), d.Cskip ? to_src(skips[i]) : null_src(), training, st));
    RET_IF(run_conv(h, c2, to_src(bn_act(h, c1)), null_src(), training, st));
    xd = bn_act(h, c2);
  }
  // head
  vk_src hs = to_src(xd);
  RET_IF(vk_head_fwd(dt, N, S, SW, &hs, h->params + h->head_w_off, h->params + h->head_b_off, logits, st));
  if (training) {
    hipLaunchKernelGGL(k_inc_i64, dim3(1), dim3(64), 0, st, (int)h->bns.size(), h->nbt);
    VK_CHECK_HIP(hipGetLastError());
  }
  return VK_OK;
}

extern "C" int vk_unet_loss(vk_unet* h, const float* logits, const float* target, float* loss_out, float grad_scale, float w_bce,
                            float w_dice, void* stream) {
  VK_CHECK_ARG(h && h->bound && logits && target && loss_out, "vk_unet_loss: plan not bound or null tensor");
  const size_t count = (size_t)h->cfg.N * h->cfg.size * h->cfg.width;
  float* dl = h->cfg.training ? (float*)(h->ws + h->off_dlogits) : nullptr;
  return vk_bce_dice_loss(count, logits, target, (double*)(h->ws + h->off_loss_sums), loss_out, dl, grad_scale, w_bce, w_dice, stream);
}

// ------------------------------------------------------------------------------------------------ backward
#ifndef VK_BN_APPLY_FUSED_MAXC_DEFAULT
#define VK_BN_APPLY_FUSED_MAXC_DEFAULT 0
#endif
namespace {

// BatchNorm backward, phase 2: coefficients (a, b, c) + dgamma / dbeta from the reduction sums, then dz = a*g + b*z + c.  Two launches
// (k_bn_bwd_coeffs: 8 channels per workgroup, fp64; then the sweep reads 3 C floats) or ONE (vk_bn_bwd_apply_fused: every workgroup of
// the sweep adds the 32 replicas of all C channels itself — 512 C bytes from L2 per workgroup): the second form takes a launch-bound
// ~6 us kernel off the critical path where C is small and the tensor large (VK_BN_APPLY_FUSED_MAXC, measured in profiles/r04/).
int bn_bwd_phase2(vk_unet* h, BnL& b, int C, size_t pixels, const void* dy, const void* z, int mask_mode, const void* mask_src, void* dz,
                  void* g_out, int g_acc, hipStream_t st) {
  static const int maxc = getenv("VK_BN_APPLY_FUSED_MAXC") ? atoi(getenv("VK_BN_APPLY_FUSED_MAXC")) : VK_BN_APPLY_FUSED_MAXC_DEFAULT;
  if (C <= maxc)
    return vk_bn_bwd_apply_fused(h->cfg.dtype, pixels, C, dy, z, mask_mode, b.scale, b.shift, mask_src, b.bsums, b.count, h->params + b.g_off,
                                 b.mean, b.invstd, h->grads + b.g_off, h->grads + b.b_off, dz, g_out, g_acc, st);
  RET_IF(vk_bn_bwd_coeffs(C, b.bsums, b.count, h->params + b.g_off, b.mean, b.invstd, h->grads + b.g_off, h->grads + b.b_off, b.coef, st));
  return vk_bn_bwd_apply(h->cfg.dtype, pixels, C, dy, z, mask_mode, b.scale, b.shift, mask_src, b.coef, dz, g_out, g_acc, st);
}

// gradient wrt activated output (in c.g) -> gradient wrt z (in place); dgamma/dbeta accumulated
// prereduced: the producer of c.g already stored g = dy * mask and accumulated the sums (vk_bnr fusion)
int bn_relu_bwd_inplace(vk_unet* h, ConvL& c, bool prereduced, hipStream_t st) {
  BnL& b = h->bns[c.bn];
  const size_t pixels = (size_t)h->cfg.N * c.Hout * c.Wout;
  if (!prereduced) RET_IF(vk_bn_bwd_reduce(h->cfg.dtype, pixels, c.K, c.g, c.z, 1, b.scale, b.shift, nullptr, b.bsums, st));
  return bn_bwd_phase2(h, b, c.K, pixels, c.g, c.z, prereduced ? 0 : 1, nullptr, c.g, nullptr, 0, st);
}

// stream the weight-gradient kernels run on: the side stream, forked here behind everything enqueued on `st` so far
// (a failed fork is an error of the call, never a silent fall-back onto the caller's stream)
int wgrad_stream(vk_unet* h, hipStream_t st, hipStream_t* out) {
  *out = st;
  if (!h->side) return VK_OK;
  VK_CHECK_HIP(hipEventRecord(h->ev_fork, st));
  VK_CHECK_HIP(hipStreamWaitEvent(h->side, h->ev_fork, 0));
  h->side_dirty = true;
  *out = h->side;
  return VK_OK;
}

int join_side(vk_unet* h, hipStream_t st) {
  if (!h->side || !h->side_dirty) return VK_OK;
  VK_CHECK_HIP(hipEventRecord(h->ev_join, h->side));
  VK_CHECK_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
  h->side_dirty = false;
  return VK_OK;
}

int conv_wgrad(vk_unet* h, ConvL& c, const vk_src& s0, const vk_src& s1, hipStream_t st) {
  vk_conv_desc d = conv_desc(h, c, s0, s1);
  // the 64 x 64 tile class waits for the end of its backward stage (flush_wgrads): dz (c.g) and the operands are this layer's own
  // buffers and stay untouched until the next step
  if (!h->side && vk::wgrad_batch_supports(&d)) {
    h->pending.push_back(vk_unet::WItem{d, c.g, h->grads + c.w_off});
    return VK_OK;
  }
  hipStream_t ws;
  RET_IF(wgrad_stream(h, st, &ws));
  return vk_conv_wgrad(&d, c.g, h->grads + c.w_off, h->ws + h->off_wslab, VK_WGRAD_WORKSPACE_BYTES, ws);
}

vk_bnr bnr_of(vk_unet* h, ConvL& target) {     // fused BN+ReLU backward reduce descriptor for the layer `target`
  BnL& b = h->bns[target.bn];
  vk_bnr r;
  r.z = target.z; r.scale = b.scale; r.shift = b.shift; r.sums = b.bsums; r.mask = nullptr; r.accumulate = 0;
  return r;
}

vk_conv_desc dgrad_desc(vk_unet* h, ConvL& c) {
  vk_conv_desc d;
  d.dtype = h->cfg.dtype;
  d.N = h->cfg.N; d.H = c.Hout; d.W = c.Wout; d.Ho = c.Hin; d.Wo = c.Win;
  d.K = c.Cin; d.R = c.R; d.S = c.R; d.stride = c.stride; d.pad = c.pad; d.transposed = 1;
  vk_src s;
  s.ptr = c.g; s.C = c.K; s.up = 0; s.scale = nullptr; s.shift = nullptr; s.relu = 0;
  d.src0 = s;
  d.src1 = null_src();
  return d;
}

// dgrad of conv c written to `into.g`, with the BN+ReLU backward reduce of layer `into` fused when the tile kernels
// cover the shape; *fused tells the caller whether that happened
int conv_dgrad_into(vk_unet* h, ConvL& c, ConvL& into, bool* fused, hipStream_t st) {
  vk_conv_desc d = dgrad_desc(h, c);
  vk_bnr r = bnr_of(h, into);
  int rc = getenv("VK_NO_BNR_FUSION") ? VK_ERR_UNSUPPORTED : vk_conv_dgrad_fused(&d, dgrad_weights(h, c), into.g, nullptr, 0, 0, &r, st);
  *fused = rc == VK_OK;
  if (rc != VK_ERR_UNSUPPORTED) return rc;
  if (c.halo_dg) return vk_conv_fwd_packed(&d, dgrad_weights(h, c), into.g, nullptr, 0, 0, nullptr, st);
  return vk_conv_fwd(&d, dgrad_weights(h, c), into.g, nullptr, 0, 0, nullptr, st);
}

// dx = dgrad(dz of conv c) into y (and y1 for the channel split)
int conv_dgrad(vk_unet* h, ConvL& c, void* y, void* y1, int split, int accumulate, hipStream_t st) {
  vk_conv_desc d;
  d.dtype = h->cfg.dtype;
  d.N = h->cfg.N; d.H = c.Hout; d.W = c.Wout; d.Ho = c.Hin; d.Wo = c.Win;
  d.K = c.Cin; d.R = c.R; d.S = c.R; d.stride = c.stride; d.pad = c.pad; d.transposed = 1;
  vk_src s;
  s.ptr = c.g; s.C = c.K; s.up = 0; s.scale = nullptr; s.shift = nullptr; s.relu = 0;
  d.src0 = s;
  d.src1 = null_src();
  if (c.halo_dg) return vk_conv_fwd_packed(&d, dgrad_weights(h, c), y, y1, split, accumulate, nullptr, st);
  return vk_conv_fwd(&d, dgrad_weights(h, c), y, y1, split, accumulate, nullptr, st);
}

int backward_decoder(vk_unet* h, int i, hipStream_t st) {
  DecL& d = h->decs[i];
  ConvL& c1 = h->convs[d.conv1];
  ConvL& c2 = h->convs[d.conv2];
  const int N = h->cfg.N;
  ConvL& stem = h->convs[h->stem_conv];
  // inputs of conv1 (same virtual tensors as in forward)
  Act xprev, skip{nullptr, 0, nullptr, nullptr, 0};
  void* g_prev;       // gradient buffer of the upsampled input (low resolution)
  if (i == 0) {
    BlockL& f5 = h->blocks[h->layer_last_block[3]];
    xprev = Act{f5.out, 512, nullptr, nullptr, 0};
    g_prev = f5.gout;
  } else {
    ConvL& pc2 = h->convs[h->decs[i - 1].conv2];
    xprev = bn_act(h, pc2);
    g_prev = pc2.g;
  }
  void* g_skip = nullptr;
  if (i <= 2) {
    BlockL& fb = h->blocks[h->layer_last_block[2 - i]];
    skip = Act{fb.out, fb.C, nullptr, nullptr, 0};
    g_skip = fb.gout;
  } else if (i == 3) {
    skip = bn_act(h, stem);
    g_skip = stem.g;
  }
  // conv2 unit (its gradient was pre-masked/pre-reduced by the producer when that kernel supports the fusion)
  // Order inside a unit: BN backward -> DATA gradient -> weight gradient.  The weight gradient only feeds the optimizer; issued
  // after the data gradient it can run (side stream, vk_unet_set_side_stream) beside the HBM-bound BatchNorm backward of the
  // NEXT layer instead of beside an MFMA-bound data gradient.
  RET_IF(bn_relu_bwd_inplace(h, c2, h->g_prereduced[d.conv2] != 0, st));
  bool fused1 = false;
  RET_IF(conv_dgrad_into(h, c2, c1, &fused1, st));
  RET_IF(conv_wgrad(h, c2, to_src(bn_act(h, c1)), null_src(), st));
  // conv1 unit
  RET_IF(bn_relu_bwd_inplace(h, c1, fused1, st));
  // data gradient with the nearest-x2 upsample backward fused into its epilogue (the full-resolution d_up never exists)
  // and, for i > 0, the BN+ReLU backward reduce of the previous decoder block's conv2;
  // shapes the tile kernels do not cover fall back to dgrad + a separate 2x2-sum pass
  if (i > 0) h->g_prereduced[h->decs[i - 1].conv2] = 0;
  {
    vk_conv_desc dd = dgrad_desc(h, c1);
    vk_bnr r;
    const bool want_bnr = i > 0 && !getenv("VK_NO_BNR_FUSION");
    if (want_bnr) r = bnr_of(h, h->convs[h->decs[i - 1].conv2]);
    const int rc = vk_conv_dgrad_fused(&dd, dgrad_weights(h, c1), g_prev, d.Cskip ? g_skip : nullptr, d.Cskip ? d.Cup : 0, 1,
                                       want_bnr ? &r : nullptr, st);
    if (rc == VK_OK && want_bnr) h->g_prereduced[h->decs[i - 1].conv2] = 1;
    if (rc != VK_OK && rc != VK_ERR_UNSUPPORTED) return rc;
    if (rc == VK_ERR_UNSUPPORTED) {
      void* dup = h->ws + h->off_dup;
      if (d.Cskip) RET_IF(conv_dgrad(h, c1, dup, g_skip, d.Cup, 0, st));
      else RET_IF(conv_dgrad(h, c1, dup, nullptr, 0, 0, st));
      RET_IF(vk_upsample2x_bwd(h->cfg.dtype, N, d.H, d.W, d.Cup, dup, g_prev, 0, st));
    }
  }
  return conv_wgrad(h, c1, to_src(xprev, 1), d.Cskip ? to_src(skip) : null_src(), st);
}

int backward_block(vk_unet* h, int bi, hipStream_t st) {
  BlockL& k = h->blocks[bi];
  ConvL& c1 = h->convs[k.conv1];
  ConvL& c2 = h->convs[k.conv2];
  BnL& b2 = h->bns[c2.bn];
  const vk_dtype dt = h->cfg.dtype;
  const size_t pixels = (size_t)h->cfg.N * k.Hout * k.Wout;
  // block input (materialised) and its gradient buffer
  Act xin;
  void* gin;
  if (bi == 0) {
    xin = Act{h->ws + h->off_pool, 64, nullptr, nullptr, 0};
    gin = h->ws + h->off_gpool;
  } else {
    BlockL& pb = h->blocks[bi - 1];
    xin = Act{pb.out, pb.C, nullptr, nullptr, 0};
    gin = pb.gout;
  }
  // tail: out = relu(bn2(z2) + shortcut);  g = gout * (out > 0).  r04: when the NEXT block is an identity block, the data gradient of
  // its conv1 — the kernel that completes this block's gout — has already masked it with [out > 0] and added bn2's sums (vk_bnr.mask,
  // see below): no reduce pass here and the apply passes read g as it stands (mask mode 0: one tensor read less each)
  const bool pre = h->tail_prereduced[(size_t)bi] != 0;
  h->tail_prereduced[(size_t)bi] = 0;
  const int mm = pre ? 0 : 2;
  if (!pre) RET_IF(vk_bn_bwd_reduce(dt, pixels, k.C, k.gout, c2.z, 2, nullptr, nullptr, k.out, b2.bsums, st));
  if (k.convd < 0) {
    // identity shortcut: gin (+)= g
    RET_IF(bn_bwd_phase2(h, b2, k.C, pixels, k.gout, c2.z, mm, k.out, c2.g, gin, k.in_has_grad_first ? 1 : 0, st));
  } else {
    ConvL& cd = h->convs[k.convd];
    BnL& bd = h->bns[cd.bn];
    RET_IF(bn_bwd_phase2(h, b2, k.C, pixels, k.gout, c2.z, mm, k.out, c2.g, nullptr, 0, st));
    RET_IF(vk_bn_bwd_reduce(dt, pixels, k.C, k.gout, cd.z, mm, nullptr, nullptr, k.out, bd.bsums, st));
    RET_IF(bn_bwd_phase2(h, bd, k.C, pixels, k.gout, cd.z, mm, k.out, cd.g, nullptr, 0, st));
  }
  // conv2 (data gradient first, weight gradient after it: see backward_decoder)
  bool fused1 = false;
  RET_IF(conv_dgrad_into(h, c2, c1, &fused1, st));
  RET_IF(conv_wgrad(h, c2, to_src(bn_act(h, c1)), null_src(), st));
  // conv1
  RET_IF(bn_relu_bwd_inplace(h, c1, fused1, st));
  // gin was written by the identity shortcut above or (downsample blocks) by the decoder skip gradient
  // Identity block behind another block: this data gradient COMPLETES the output gradient of block bi - 1 (gin = its gout), so its
  // epilogue applies that block's tail — mask [out > 0], sums of g and g * z2 for bn2 — while the vectors are in registers: the
  // separate reduce pass (3 tensor reads) and the mask reads of the apply passes go (train.py:443/:448: autograd's relu / add /
  // batch_norm backward nodes of torchvision's BasicBlock).  Downsample blocks finish gin with their 1x1 data gradient: not fused.
  bool tail_fused = false;
  const int tail_maxc = getenv("VK_TAIL_BNR_MAXC") ? atoi(getenv("VK_TAIL_BNR_MAXC")) : 1 << 20;      // experiment knob: fuse only up to this width
  if (k.convd < 0 && bi > 0 && c1.halo_dg && k.C <= tail_maxc && !getenv("VK_NO_TAIL_BNR_FUSION")) {
    BlockL& pb = h->blocks[bi - 1];
    ConvL& pc2 = h->convs[pb.conv2];
    vk_conv_desc dd = dgrad_desc(h, c1);
    vk_bnr r;
    r.z = pc2.z; r.scale = nullptr; r.shift = nullptr; r.sums = h->bns[pc2.bn].bsums; r.mask = pb.out; r.accumulate = 1;
    const int rc = vk_conv_dgrad_fused(&dd, dgrad_weights(h, c1), gin, nullptr, 0, 0, &r, st);
    if (rc == VK_OK) { tail_fused = true; h->tail_prereduced[(size_t)bi - 1] = 1; }
    else if (rc != VK_ERR_UNSUPPORTED) return rc;
  }
  if (!tail_fused) RET_IF(conv_dgrad(h, c1, gin, nullptr, 0, 1, st));
  if (k.convd >= 0) {
    ConvL& cd = h->convs[k.convd];
    RET_IF(conv_dgrad(h, cd, gin, nullptr, 0, 1, st));
    RET_IF(conv_wgrad(h, c1, to_src(xin), null_src(), st));
    return conv_wgrad(h, cd, to_src(xin), null_src(), st);
  }
  return conv_wgrad(h, c1, to_src(xin), null_src(), st);
}

int backward_stem(vk_unet* h, hipStream_t st) {
  const int N = h->cfg.N, S = h->cfg.size, SW = h->cfg.width;
  ConvL& stem = h->convs[h->stem_conv];
  // stem.g holds the skip gradient of f1 (from decoder block 3); add the maxpool path
  if (getenv("VK_NO_POOL_BNR_FUSION")) {
    RET_IF(vk_maxpool_bwd(h->cfg.dtype, N, S / 2, SW / 2, 64, h->ws + h->off_gpool, (const uint8_t*)(h->ws + h->off_argmax), stem.g, st));
    RET_IF(bn_relu_bwd_inplace(h, stem, false, st));
  } else {
    // one pass: maxpool backward + mask + BN-backward sums (saves a read-modify-write and a read of the 256x256x64 gradient)
    BnL& b = h->bns[stem.bn];
    RET_IF(vk_maxpool_bwd_bn_reduce(h->cfg.dtype, N, S / 2, SW / 2, 64, h->ws + h->off_gpool, (const uint8_t*)(h->ws + h->off_argmax), stem.z,
                                    b.scale, b.shift, stem.g, b.bsums, st));
    // r04: the stem's dz has ONE reader, the weight gradient below (no data gradient: the input needs none) — the 16-bit kernel forms
    // dz = a*g + b*z + c itself while staging, so the apply pass (g, z -> dz: 805 MB at bs 32) is not run.  VK_NO_STEM_BNA=1: the pass
    if (h->cfg.dtype != VK_F32 && !getenv("VK_NO_STEM_BNA")) {
      RET_IF(vk_bn_bwd_coeffs(64, b.bsums, b.count, h->params + b.g_off, b.mean, b.invstd, h->grads + b.g_off, h->grads + b.b_off, b.coef, st));
      hipStream_t ws;
      RET_IF(wgrad_stream(h, st, &ws));
      const int rc = vk_stem_wgrad_bn(h->cfg.dtype, N, S, SW, h->ws + h->off_x4, stem.g, stem.z, b.coef, h->grads + stem.w_off,
                                      h->ws + h->off_wslab, VK_WGRAD_WORKSPACE_BYTES, ws);
      if (rc != VK_ERR_UNSUPPORTED) return rc;
      // (shape outside the tile kernel: coefficients are in place, finish with the separate apply pass)
      RET_IF(vk_bn_bwd_apply(h->cfg.dtype, (size_t)N * (S / 2) * (SW / 2), 64, stem.g, stem.z, 0, b.scale, b.shift, nullptr, b.coef, stem.g, nullptr, 0, st));
    } else {
      RET_IF(bn_relu_bwd_inplace(h, stem, true, st));
    }
  }
  hipStream_t ws;
  RET_IF(wgrad_stream(h, st, &ws));
  return vk_stem_wgrad(h->cfg.dtype, N, S, SW, h->ws + h->off_x4, stem.g, h->grads + stem.w_off, h->ws + h->off_wslab, VK_WGRAD_WORKSPACE_BYTES, ws);
}

// the collected weight gradients of a stage: one batched launch (tables built once per stage and per workgroup budget), a single one
// through the ordinary entry point
int flush_wgrads(vk_unet* h, int stage, hipStream_t st) {
  const int n = (int)h->pending.size();
  if (n == 0) return VK_OK;
  char* const slab = h->ws + h->off_wslab;
  auto each = [&]() {
    int rc = VK_OK;
    for (const vk_unet::WItem& it : h->pending) {
      rc = vk_conv_wgrad(&it.d, it.dz, it.dw, slab, kEngineSlabBytes, st);
      if (rc != VK_OK) break;
    }
    h->pending.clear();
    return rc;
  };
  if (n == 1 || n > kMaxBatch || stage < 0 || stage >= kMaxStages) return each();
  const int target = 256 - vkh::reserved_cus();
  const int slot = target == 256 ? 0 : 1;
  vk_unet::WPlan& wp = h->wplans[stage][slot];
  char* const tables = h->ws + h->off_tab_wbatch + ((size_t)stage * 2 + slot) * VK_WGRAD_BATCH_TABLE_BYTES;
  if (!wp.built || wp.target != target || wp.n != n) {
    if (wp.built) VK_CHECK_HIP(hipStreamSynchronize(st));          // a launch that reads the old tables may still be in flight
    vk_conv_desc descs[kMaxBatch];
    const void* dz[kMaxBatch];
    float* dw[kMaxBatch];
    for (int i = 0; i < n; ++i) { descs[i] = h->pending[(size_t)i].d; dz[i] = h->pending[(size_t)i].dz; dw[i] = h->pending[(size_t)i].dw; }
    wp.plan = vk::WgradBatchPlan();
    const int rc = vk::wgrad_batch_build(descs, dz, dw, n, target, tables, VK_WGRAD_BATCH_TABLE_BYTES, &wp.plan);
    if (rc != VK_OK) { h->pending.clear(); return rc; }
    wp.built = true; wp.target = target; wp.n = n;
  }
  if (wp.plan.slab_need > kEngineSlabBytes) return each();
  const int rc = vk::wgrad_batch_launch(h->cfg.dtype, wp.plan, tables, slab, kEngineSlabBytes, st);
  h->pending.clear();
  return rc;
}

int backward_stage(vk_unet* h, const float* dlogits, int stage, hipStream_t st) {
  const int N = h->cfg.N, S = h->cfg.size, SW = h->cfg.width;
  switch (stage) {
    case 0: {
      VK_CHECK_HIP(hipMemsetAsync(h->ws + h->off_bsums, 0, h->stats_bytes, st));
      ConvL& last = h->convs[h->decs[4].conv2];
      vk_src hs = to_src(bn_act(h, last));
      {
        vk_bnr r = bnr_of(h, last);
        const bool fuse = !getenv("VK_NO_BNR_FUSION");
        const float* dl = dlogits ? dlogits : (const float*)(h->ws + h->off_dlogits);
        if (fuse) RET_IF(vk_head_bwd_fused(h->cfg.dtype, N, S, SW, &hs, h->params + h->head_w_off, dl, last.g, h->grads + h->head_w_off,
                                           h->grads + h->head_b_off, &r, h->ws + h->off_wslab, VK_HEAD_WORKSPACE_BYTES, st));
        else RET_IF(vk_head_bwd(h->cfg.dtype, N, S, SW, &hs, h->params + h->head_w_off, dl, last.g, h->grads + h->head_w_off,
                                h->grads + h->head_b_off, h->ws + h->off_wslab, VK_HEAD_WORKSPACE_BYTES, st));
        h->g_prereduced[h->decs[4].conv2] = fuse ? 1 : 0;
      }
      RET_IF(backward_decoder(h, 4, st));
      RET_IF(backward_decoder(h, 3, st));
      return backward_decoder(h, 2, st);
    }
    case 1: return backward_decoder(h, 1, st);
    case 2: return backward_decoder(h, 0, st);
    case 3: return backward_block(h, 15, st);
    case 4: return backward_block(h, 14, st);
    case 5: return backward_block(h, 13, st);
    case 6:
      for (int b = 12; b >= 10; --b) RET_IF(backward_block(h, b, st));
      return VK_OK;
    case 7:
      for (int b = 9; b >= 7; --b) RET_IF(backward_block(h, b, st));
      return VK_OK;
    case 8:
      for (int b = 6; b >= 3; --b) RET_IF(backward_block(h, b, st));
      return VK_OK;
    case 9:
      for (int b = 2; b >= 0; --b) RET_IF(backward_block(h, b, st));
      return backward_stem(h, st);
  }
  vkh::set_error("vk_unet_backward: bad stage %d", stage);
  return VK_ERR_ARG;
}

}  // namespace

extern "C" int vk_unet_backward(vk_unet* h, const float* dlogits, int stage_begin, int stage_end, void* stream) {
  VK_CHECK_ARG(h && h->bound && h->cfg.training && h->grads, "vk_unet_backward: needs a bound training plan");
  VK_CHECK_ARG(stage_begin >= 0 && stage_end <= (int)h->buckets.size() && stage_begin <= stage_end, "vk_unet_backward: bad stage range");
  for (int s = stage_begin; s < stage_end; ++s) {
    int rc = backward_stage(h, dlogits, s, (hipStream_t)stream);
    // the collected weight gradients run at the end of the CALL (every stage of the range in one batch) — a caller that needs a
    // stage's gradients early (the data-parallel reducer: one call per stage) gets them per stage
    if (rc == VK_OK && (s + 1 == stage_end || (int)h->pending.size() > kMaxBatch - 8)) rc = flush_wgrads(h, s, (hipStream_t)stream);
    else if (rc != VK_OK) h->pending.clear();
    const int rj = join_side(h, (hipStream_t)stream);        // also after a failed stage: never leave the side stream un-joined
    if (rc != VK_OK) return rc;
    if (rj != VK_OK) return rj;
  }
  return VK_OK;
}
