// Diagnostic (r04): do vector-memory STORES retire (decrement vmcnt) ahead of OLDER vector-memory LOADS on gfx950?
//
// Why: the r03 streaming convolution kernel produced wrong, run-to-run different results when its epilogue used MUBUF stores
// (`raw_buffer_store`) beside three rows of MUBUF loads in flight behind the compiler's partial `s_waitcnt vmcnt(n)` waits; with global
// stores under a lane predicate the same code was correct.  The compiler's waits count the stores and are right IF vmcnt retires in
// issue order (MI355X_MICROARCH.md: "Loads, stores, atomics and LDS-DMA count together, in issue order").  This probe decides that
// premise with one launch per variant:
//
//   per wave and probe:   destination VGPRs := sentinel
//                         buffer_load_dwordx4   (one cold 1 KiB line group per wave: 16 KiB stride through a 2 GiB buffer)
//                         K x  store            (8 bytes per lane, lines made L2-resident by a warm-up store + vmcnt(0))
//                         s_waitcnt vmcnt(K)    == "the load has returned" under in-order retirement
//                         ds_write_b128 of the destination VGPRs (an immediate copy: the data goes out with the instruction)
//                         s_waitcnt vmcnt(0) lgkmcnt(0); compare copy and final registers with the expected pattern
//
//   a lane whose early copy still holds the sentinel while its final registers hold the pattern proves that the wait was satisfied
//   while the load was outstanding, i.e. that one of the K younger stores left the counter first.
//
// Store variants: 0 buffer_store in range, 1 buffer_store with EVERY lane out of range (offset 0x80000000 against the descriptor's
// num_records: the store is dropped by the range check), 2 buffer_store with lanes 32-63 out of range, 3 global_store.
// Load variants (the younger operations are buffer_load_dwordx2 instead of stores — do cheap loads overtake an older HBM miss?):
// 4 every lane out of range (answered with zeros, no memory access), 5 L2-resident lines, 6 lanes 32-63 out of range.
// Controls: WAIT = K + 1 (no wait for the load: every lane must show the sentinel => the detection works),
//           WAIT = 0     (everything waited for: no lane may show it).
//
// Build + run (GPU box):  hipcc -O2 --offload-arch=gfx950 tests/diag/vmcnt_order_probe.hip -o /tmp/vmcnt_probe && /tmp/vmcnt_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;

static constexpr uint32_t kSentinel = 0xDEADBEEFu;
static constexpr uint32_t kOOB = 0x80000000u;
static constexpr int kColdStride = 16384;       // bytes between the 1 KiB pieces two probes read
static constexpr int kHotPerWave = 4096;        // bytes of store target per wave (8 stores x 512 B)

__host__ __device__ inline uint32_t pattern(uint64_t byte_off, int dword) {   // what the cold buffer holds
  uint32_t v = (uint32_t)(byte_off >> 2) + (uint32_t)dword;
  v = v * 2654435761u + 12345u;
  return v == kSentinel ? v + 1 : v;
}

__global__ void fill_cold(uint32_t* cold, size_t n_probes) {
  // only the 1 KiB piece of each 16 KiB stride is ever read
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = n_probes * 256;                 // dwords
  for (; t < total; t += (size_t)gridDim.x * blockDim.x) {
    size_t probe = t >> 8, dw = t & 255;
    uint64_t byte_off = probe * (uint64_t)kColdStride + dw * 4;
    cold[byte_off >> 2] = pattern(byte_off & ~15ull, (int)(dw & 3));
  }
}

__global__ void flush_caches(uint32_t* junk, size_t n) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; t < n; t += (size_t)gridDim.x * blockDim.x) junk[t] = (uint32_t)t;
}

struct Counts {
  unsigned long long probes_lanes;     // lanes x probes examined
  unsigned long long early_sentinel;   // early copy == sentinel in all four dwords (load not back at the wait)
  unsigned long long early_torn;       // early copy mixes sentinel and data
  unsigned long long early_ok;         // early copy == pattern
  unsigned long long final_bad;        // final registers != pattern (would be a broken probe)
  unsigned long long waves_hit;        // wave-probes with at least one sentinel / torn lane
};

#define ST_BUF(n)  "buffer_store_dwordx2 %[sd], %[so], %[srdh], 0 offen offset:" #n "\n\t"
#define ST_GLB(n)  "global_store_dwordx2 %[ga], %[sd], off offset:" #n "\n\t"
#define LD_BUF(n)  "buffer_load_dwordx2 %[t], %[so], %[srdh], 0 offen offset:" #n "\n\t"      // a YOUNGER LOAD instead of a store
#define REP1(S) S(0)
#define REP2(S) S(0) S(512)
#define REP4(S) S(0) S(512) S(1024) S(1536)
#define REP8(S) S(0) S(512) S(1024) S(1536) S(2048) S(2560) S(3072) S(3584)

// one probe; K and the wait immediate are compile-time through the strings
#define PROBE_BODY(STORES, WAITN)                                                                    \
  asm volatile(                                                                                        \
      "s_nop 4\n\t"                                                                                    \
      "buffer_load_dwordx4 %[d], %[lo], %[srdc], 0 offen\n\t"                                          \
      STORES                                                                                           \
      "s_waitcnt vmcnt(" #WAITN ")\n\t"                                                                \
      "ds_write_b128 %[la], %[d]\n\t"                                                                  \
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                              \
      : [d] "+&v"(d), [t] "=&v"(tscratch)                                                              \
      : [lo] "v"(load_off), [srdc] "s"(srd_cold), [sd] "v"(sdata), [so] "v"(store_off), [srdh] "s"(srd_hot), \
        [ga] "v"(gaddr), [la] "v"(lds_addr)                                                            \
      : "memory")

template <int K, int VAR, int WAITSEL>   // WAITSEL: 0 -> vmcnt(K) (the question), 1 -> vmcnt(K + 1) (control: no wait), 2 -> vmcnt(0)
__global__ __launch_bounds__(256) void probe_kernel(const uint32_t* cold, uint64_t cold_bytes, uint32_t* hot, uint32_t hot_bytes,
                                                    int iters, Counts* counts) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[256 * 4];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const size_t wave = (size_t)blockIdx.x * 4 + wave_in_block;
  // descriptors: wave-uniform by construction (kernel arguments only)
  const uint64_t cb = (uint64_t)cold, hb = (uint64_t)hot;
  i32x4 srd_cold, srd_hot;
  srd_cold[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)cb);
  srd_cold[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(cb >> 32));
  srd_cold[2] = __builtin_amdgcn_readfirstlane((int)0xFFFFFFFFu);              // num_records: offsets are < 4 GiB by construction
  srd_cold[3] = 0x00020000;
  srd_hot[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)hb);
  srd_hot[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(hb >> 32));
  srd_hot[2] = __builtin_amdgcn_readfirstlane((int)hot_bytes);
  srd_hot[3] = 0x00020000;
  const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)(&lds[threadIdx.x * 4]);         // LDS byte address of this lane's 16-byte slot
  const uint32_t hot_lane = (uint32_t)(wave * kHotPerWave + lane * 8);
  uint32_t store_off = hot_lane;
  if (VAR == 1 || VAR == 4) store_off = kOOB;
  if ((VAR == 2 || VAR == 6) && lane >= 32) store_off = kOOB;
  const uint64_t gaddr = hb + hot_lane;
  const u32x2 sdata = {(uint32_t)wave, (uint32_t)lane};
  unsigned long long n_sent = 0, n_torn = 0, n_ok = 0, n_bad = 0, n_hit = 0;
  for (int it = 0; it < iters; ++it) {
    // warm the store lines into L2 (plain stores, all waited for)
#pragma unroll
    for (int k = 0; k < 8; ++k) *reinterpret_cast<volatile u32x2*>((char*)hot + hot_lane + k * 512) = sdata;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const uint64_t probe = wave * (uint64_t)iters + it;
    const uint64_t boff = probe * (uint64_t)kColdStride + (uint64_t)lane * 16;
    if (boff + 16 > cold_bytes) break;
    const uint32_t load_off = (uint32_t)boff;
    u32x4 d = {kSentinel, kSentinel, kSentinel, kSentinel};
    u32x2 tscratch;
    if constexpr (VAR == 3) {
      if constexpr (K == 1) { if constexpr (WAITSEL == 0) PROBE_BODY(REP1(ST_GLB), 1); else if constexpr (WAITSEL == 1) PROBE_BODY(REP1(ST_GLB), 2); else PROBE_BODY(REP1(ST_GLB), 0); }
      if constexpr (K == 2) { if constexpr (WAITSEL == 0) PROBE_BODY(REP2(ST_GLB), 2); else if constexpr (WAITSEL == 1) PROBE_BODY(REP2(ST_GLB), 3); else PROBE_BODY(REP2(ST_GLB), 0); }
      if constexpr (K == 4) { if constexpr (WAITSEL == 0) PROBE_BODY(REP4(ST_GLB), 4); else if constexpr (WAITSEL == 1) PROBE_BODY(REP4(ST_GLB), 5); else PROBE_BODY(REP4(ST_GLB), 0); }
      if constexpr (K == 8) { if constexpr (WAITSEL == 0) PROBE_BODY(REP8(ST_GLB), 8); else if constexpr (WAITSEL == 1) PROBE_BODY(REP8(ST_GLB), 9); else PROBE_BODY(REP8(ST_GLB), 0); }
    } else if constexpr (VAR >= 4) {
      if constexpr (K == 1) { if constexpr (WAITSEL == 0) PROBE_BODY(REP1(LD_BUF), 1); else if constexpr (WAITSEL == 1) PROBE_BODY(REP1(LD_BUF), 2); else PROBE_BODY(REP1(LD_BUF), 0); }
      if constexpr (K == 2) { if constexpr (WAITSEL == 0) PROBE_BODY(REP2(LD_BUF), 2); else if constexpr (WAITSEL == 1) PROBE_BODY(REP2(LD_BUF), 3); else PROBE_BODY(REP2(LD_BUF), 0); }
      if constexpr (K == 4) { if constexpr (WAITSEL == 0) PROBE_BODY(REP4(LD_BUF), 4); else if constexpr (WAITSEL == 1) PROBE_BODY(REP4(LD_BUF), 5); else PROBE_BODY(REP4(LD_BUF), 0); }
      if constexpr (K == 8) { if constexpr (WAITSEL == 0) PROBE_BODY(REP8(LD_BUF), 8); else if constexpr (WAITSEL == 1) PROBE_BODY(REP8(LD_BUF), 9); else PROBE_BODY(REP8(LD_BUF), 0); }
      asm volatile("" :: "v"(tscratch));
    } else {
      if constexpr (K == 1) { if constexpr (WAITSEL == 0) PROBE_BODY(REP1(ST_BUF), 1); else if constexpr (WAITSEL == 1) PROBE_BODY(REP1(ST_BUF), 2); else PROBE_BODY(REP1(ST_BUF), 0); }
      if constexpr (K == 2) { if constexpr (WAITSEL == 0) PROBE_BODY(REP2(ST_BUF), 2); else if constexpr (WAITSEL == 1) PROBE_BODY(REP2(ST_BUF), 3); else PROBE_BODY(REP2(ST_BUF), 0); }
      if constexpr (K == 4) { if constexpr (WAITSEL == 0) PROBE_BODY(REP4(ST_BUF), 4); else if constexpr (WAITSEL == 1) PROBE_BODY(REP4(ST_BUF), 5); else PROBE_BODY(REP4(ST_BUF), 0); }
      if constexpr (K == 8) { if constexpr (WAITSEL == 0) PROBE_BODY(REP8(ST_BUF), 8); else if constexpr (WAITSEL == 1) PROBE_BODY(REP8(ST_BUF), 9); else PROBE_BODY(REP8(ST_BUF), 0); }
    }
    // the early copy (LDS) and the final registers against the expected pattern
    const volatile uint32_t* e = &lds[threadIdx.x * 4];
    int sent = 0, good = 0, fin = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t want = pattern(boff, j);
      const uint32_t got = e[j];
      sent += got == kSentinel;
      good += got == want;
      fin += d[j] == want;
    }
    n_sent += sent == 4;
    n_ok += good == 4;
    n_torn += (sent != 4 && good != 4);
    n_bad += fin != 4;
    const bool hit = good != 4;
    n_hit += (lane == 0) && (__ballot(hit) != 0ull);
  }
  atomicAdd(&counts->probes_lanes, (unsigned long long)iters);
  if (n_sent) atomicAdd(&counts->early_sentinel, n_sent);
  if (n_torn) atomicAdd(&counts->early_torn, n_torn);
  if (n_ok) atomicAdd(&counts->early_ok, n_ok);
  if (n_bad) atomicAdd(&counts->final_bad, n_bad);
  if (n_hit) atomicAdd(&counts->waves_hit, n_hit);
}

template <int K, int VAR, int WAITSEL>
static void run(const char* name, const uint32_t* cold, uint64_t cold_bytes, uint32_t* hot, uint32_t hot_bytes, uint32_t* junk, size_t junk_n,
                int blocks, int iters, Counts* dcounts) {
  CK(hipMemset(dcounts, 0, sizeof(Counts)));
  flush_caches<<<2048, 256>>>(junk, junk_n);                       // 768 MiB of writes: the cold lines leave L2 and the Infinity Cache
  CK(hipDeviceSynchronize());
  probe_kernel<K, VAR, WAITSEL><<<blocks, 256>>>(cold, cold_bytes, hot, hot_bytes, iters, dcounts);
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  Counts c;
  CK(hipMemcpy(&c, dcounts, sizeof(c), hipMemcpyDeviceToHost));
  const unsigned long long wave_probes = (unsigned long long)blocks * 4 * iters;
  printf("%-34s K=%d wait=%-9s wave-probes %8llu lanes %10llu | early==sentinel %10llu torn %8llu ok %10llu | final bad %llu | wave-probes hit %llu\n",
         name, K, WAITSEL == 0 ? "vmcnt(K)" : WAITSEL == 1 ? "vmcnt(K+1)" : "vmcnt(0)", wave_probes, c.probes_lanes, c.early_sentinel, c.early_torn,
         c.early_ok, c.final_bad, c.waves_hit);
  fflush(stdout);
}

int main() {
  const int blocks = 4096, iters = 8;                              // 16,384 waves x 8 = 131,072 wave-probes per variant
  const size_t n_probes = (size_t)blocks * 4 * iters;
  const uint64_t cold_bytes = (uint64_t)n_probes * kColdStride;    // 2 GiB
  const uint32_t hot_bytes = (uint32_t)((size_t)blocks * 4 * kHotPerWave);   // 64 MiB
  const size_t junk_n = (size_t)768 << 18;                         // 768 MiB of dwords
  uint32_t *cold, *hot, *junk;
  Counts* dcounts;
  CK(hipMalloc(&cold, cold_bytes));
  CK(hipMalloc(&hot, hot_bytes));
  CK(hipMalloc(&junk, junk_n * 4));
  CK(hipMalloc(&dcounts, sizeof(Counts)));
  fill_cold<<<4096, 256>>>(cold, n_probes);
  CK(hipDeviceSynchronize());
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs; %zu wave-probes per variant, cold buffer %.1f GiB (stride %d B), hot %u MiB\n", prop.gcnArchName, prop.multiProcessorCount,
         n_probes, cold_bytes / 1073741824.0, kColdStride, hot_bytes >> 20);
#define VARIANTS(K)                                                                                                              \
  run<K, 0, 0>("buffer_store in range", cold, cold_bytes, hot, hot_bytes, junk, junk_n, blocks, iters, dcounts);                 \
  run<K, 1, 0>("buffer_store all lanes OOB", cold, cold_bytes, hot, hot_bytes, junk, junk_n, blocks, iters, dcounts);            \
  run<K, 2, 0>("buffer_store lanes 32-63 OOB", cold, cold_bytes, hot, hot_bytes, junk, junk_n, blocks, iters, dcounts);          \
  run<K, 3, 0>("global_store", cold, cold_bytes, hot, hot_bytes, junk, junk_n, blocks, iters, dcounts);                          \
  run<K, 4, 0>("younger LOADS, all lanes OOB", cold, cold_bytes, hot, hot_bytes, junk, junk_n, blocks, iters, dcounts);          \
  run<K, 5, 0>("younger LOADS, L2-hot in range", cold, cold_bytes, hot, hot_bytes, junk, junk_n, blocks, iters, dcounts);        \
  run<K, 6, 0>("younger LOADS, lanes 32-63 OOB", cold, cold_bytes, hot, hot_bytes, junk, junk_n, blocks, iters, dcounts);
  // controls first: the detection must see 100 % with no wait and 0 with a full wait
  run<4, 0, 1>("CONTROL no wait (buffer_store)", cold, cold_bytes, hot, hot_bytes, junk, junk_n, blocks, iters, dcounts);
  run<4, 0, 2>("CONTROL full wait (buffer_store)", cold, cold_bytes, hot, hot_bytes, junk, junk_n, blocks, iters, dcounts);
  run<4, 1, 2>("CONTROL full wait (all OOB)", cold, cold_bytes, hot, hot_bytes, junk, junk_n, blocks, iters, dcounts);
  VARIANTS(1)
  VARIANTS(2)
  VARIANTS(4)
  VARIANTS(8)
  return 0;
}
