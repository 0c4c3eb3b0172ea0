// Geometry post-processing on the device for batched probability maps: the step after
// Segmenter.infer in the reference's GUIs that turns a mask into the product's output, the indentation diagonals.
//
//   reference: ui_infer_rectangle.py:291-381 postprocess_minarearect_multi (ui_infer_quadrilateral.py:446-490 shares steps 1-3)
//     1  mask = (prob01 >= bin_thresh) * 255                                   numpy
//     2  morphologyEx OPEN then CLOSE, MORPH_ELLIPSE k x k, iterations         cv2
//     3  connectedComponentsWithStats(connectivity=8); keep area >= min_area   cv2
//     4  per kept component: findContours -> minAreaRect -> boxPoints -> int32 cv2
//     5  diagonals = longest corner pair + the remaining pair                  numpy
//
//   here, all on the GPU, for B maps of one size at once (byte / index work: no MFMA, lanes along x, bit-exact integers):
//     k_geom_morph      : steps 1-2, one launch per erosion / dilation (the first one reads the float map and thresholds on the
//                         fly); the structuring element is a table of per-row half widths (3 -> cross, 5 -> square minus corners)
//     k_geom_init/merge/flatten : 8-connected labelling by union-find with atomicMin on a label image; a component's
//                         representative is its first pixel in raster order, so ids come out in cv2 / scipy raster order
//     k_geom_runs_area  : areas by horizontal runs (one atomic per <= 64-pixel run, not per pixel)
//     k_geom_count / k_geom_scan / k_geom_assign : ordered compaction of the representatives -> label ids 1..N, kept slots
//     k_geom_clean_rows : clean mask + per-row left / right extremes of every kept component (every hull vertex is one)
//     k_geom_rect       : one workgroup per kept component: hull of the two extreme chains (monotone chain), rotating
//                         calipers over all hull edges in parallel, corners, int32 truncation, diagonals
//
// The rectangle arithmetic is float32 in the operation order of oracle/geometry_oracle.py (contraction off) so that both
// produce the same bits; distances of the int32 corners are float64 like numpy's.
#include <limits.h>
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "vk_common.h"

#pragma clang fp contract(off)

namespace vk {

struct GeomSE {            // structuring element: rows dy = -r..r, columns -hw[dy + r] .. +hw[dy + r]
  int r;
  int hw[7];
};

// ---------------------------------------------------------------------------------------------- steps 1-2
// out = erode / dilate (src) with the structuring element; src is the uint8 mask or (FROM_PROB) the float map thresholded on
// the fly.  Outside pixels never win (cv2's default border for morphology).
template <bool ERODE, bool FROM_PROB>
__global__ __launch_bounds__(256) void k_geom_morph(int h, int w, const GeomSE se, const void* __restrict__ src, float thresh,
                                                    uint8_t* __restrict__ dst) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const size_t img = (size_t)blockIdx.z * h * w;
  const float* pf = (const float*)src + img;
  const uint8_t* pm = (const uint8_t*)src + img;
  bool acc = ERODE;
  for (int dy = -se.r; dy <= se.r; ++dy) {
    const int yy = y + dy;
    if ((unsigned)yy >= (unsigned)h) continue;
    const int hwid = se.hw[dy + se.r];
    for (int dx = -hwid; dx <= hwid; ++dx) {
      const int xx = x + dx;
      if ((unsigned)xx >= (unsigned)w) continue;
      const bool v = FROM_PROB ? (pf[(size_t)yy * w + xx] >= thresh) : (pm[(size_t)yy * w + xx] != 0);
      acc = ERODE ? (acc && v) : (acc || v);
    }
  }
  dst[img + (size_t)y * w + x] = acc ? 255 : 0;
}

__global__ __launch_bounds__(256) void k_geom_binarize(size_t n, const float* __restrict__ prob, float thresh, uint8_t* __restrict__ dst) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = prob[i] >= thresh ? 255 : 0;
}

// ---------------------------------------------------------------------------------------------- step 3: labelling
// Labels are pixel indices inside one image (int32: h * w < 2^31); -1 = background.  Reads of the label image that race with
// other workgroups' atomicMin may return an OLDER parent; every older parent is still an ancestor, and every decision is
// re-validated by the value the atomic returns, so stale reads cost iterations, never correctness.
__device__ __forceinline__ int geom_find(const int* L, int i) {
  while (true) {
    const int p = __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == i) return i;
    i = p;
  }
}

__device__ __forceinline__ void geom_union(int* L, int a, int b) {
  while (true) {
    a = geom_find(L, a);
    b = geom_find(L, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(L + b, a);          // hang the larger representative under the smaller one
    if (old == b) return;                          // b was still a representative: done
    b = old;                                       // somebody re-parented b meanwhile: continue from its new parent
  }
}

// every foreground pixel starts with the start of its horizontal run INSIDE its 64-lane wave as parent (runs crossing a wave
// boundary are joined by k_geom_merge)
__global__ __launch_bounds__(256) void k_geom_init(int h, int w, const uint8_t* __restrict__ mask, int* __restrict__ L) {
  const int lane = threadIdx.x & 63;
  const int x = blockIdx.x * 64 + lane;
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const size_t img = (size_t)blockIdx.z * h * w;
  const bool in = x < w && y < h;
  const bool fg = in && mask[img + (size_t)y * w + x] != 0;
  const unsigned long long m = __ballot(fg);
  if (!in) return;
  int lab = -1;
  if (fg) {
    // highest zero bit below this lane -> the run starts right after it
    const unsigned long long below = lane ? (~m & ((1ull << lane) - 1ull)) : 0ull;
    const int start = below ? 64 - __builtin_clzll(below) : 0;
    lab = y * w + (x - lane + start);
  }
  L[img + (size_t)y * w + x] = lab;
}

__global__ __launch_bounds__(256) void k_geom_merge(int h, int w, int* __restrict__ Lb) {
  const int lane = threadIdx.x & 63;
  const int x = blockIdx.x * 64 + lane;
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  int* L = Lb + (size_t)blockIdx.z * h * w;
  const int p = y * w + x;
  if (L[p] < 0) return;       // background (a label never becomes negative)
  auto fg = [&](int yy, int xx) { return (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w && L[yy * w + xx] >= 0; };
  if (lane == 0 && fg(y, x - 1)) geom_union(L, p, p - 1);                 // run continuing from the wave on the left
  if (fg(y - 1, x)) {
    // the pixel above joins everything the two diagonal neighbours could: they touch it horizontally.  One union per
    // vertical contact of two runs is enough: only where the run above starts or this run starts
    if (!fg(y - 1, x - 1) || !fg(y, x - 1) || lane == 0) geom_union(L, p, p - w);
  } else {
    if (fg(y - 1, x - 1)) geom_union(L, p, p - w - 1);
    if (fg(y - 1, x + 1)) geom_union(L, p, p - w + 1);
  }
}

__global__ __launch_bounds__(256) void k_geom_flatten(size_t n_img, int batch, int* __restrict__ Lb) {
  const size_t total = n_img * batch;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int* L = Lb + (i / n_img) * n_img;
    const int p = (int)(i % n_img);
    const int v = L[p];
    if (v >= 0) {
      int r = v;
      while (true) { const int q = L[r]; if (q == r) break; r = q; }       // parents are final here (previous launch ended)
      L[p] = r;    // a concurrent reader sees either the old parent or the representative: both lead to the representative
    }
  }
}

// areas by horizontal runs: the first lane of every run inside a wave adds the run's length to its representative
__global__ __launch_bounds__(256) void k_geom_runs_area(int h, int w, const int* __restrict__ Lb, int* __restrict__ area_b) {
  const int lane = threadIdx.x & 63;
  const int x = blockIdx.x * 64 + lane;
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const size_t img = (size_t)blockIdx.z * h * w;
  const bool in = x < w && y < h;
  const int lab = in ? Lb[img + (size_t)y * w + x] : -1;
  const unsigned long long m = __ballot(lab >= 0);
  if (lab < 0) return;
  const bool start = lane == 0 || !((m >> (lane - 1)) & 1ull);
  if (!start) return;
  const unsigned long long rest = ~(m >> lane);                 // first zero above this lane ends the run
  const int len = rest ? __builtin_ctzll(rest) : 64 - lane;
  atomicAdd(area_b + img + lab, len);
}

// ordered compaction, pass 1: per 1024-pixel chunk, how many representatives and how many KEPT representatives
__global__ __launch_bounds__(256) void k_geom_count(size_t n_img, int nchunk, const int* __restrict__ Lb, const int* __restrict__ area_b,
                                                    int min_area, int* __restrict__ cnt /*[B][2][nchunk]*/) {
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int* L = Lb + (size_t)b * n_img;
  const int* A = area_b + (size_t)b * n_img;
  int roots = 0, kept = 0;
  for (int k = 0; k < 4; ++k) {
    const size_t p = (size_t)chunk * 1024 + k * 256 + threadIdx.x;
    const bool root = p < n_img && L[p] == (int)p;
    const bool keep = root && A[p] >= min_area;
    roots += __popcll(__ballot(root));
    kept += __popcll(__ballot(keep));
  }
  __shared__ int s[2][4];
  if ((threadIdx.x & 63) == 0) { s[0][threadIdx.x >> 6] = roots; s[1][threadIdx.x >> 6] = kept; }
  __syncthreads();
  if (threadIdx.x == 0) {
    cnt[((size_t)b * 2 + 0) * nchunk + chunk] = s[0][0] + s[0][1] + s[0][2] + s[0][3];
    cnt[((size_t)b * 2 + 1) * nchunk + chunk] = s[1][0] + s[1][1] + s[1][2] + s[1][3];
  }
}

// pass 2: exclusive scan of the chunk counts of one image (one workgroup per image and kind), totals to `tot`
__global__ __launch_bounds__(256) void k_geom_scan(int nchunk, int* __restrict__ cnt, int* __restrict__ tot /*[B][2]*/) {
  int* c = cnt + (size_t)blockIdx.x * nchunk;
  __shared__ int part[256];
  const int per = (nchunk + 255) / 256;
  const int lo = threadIdx.x * per, hi = min(nchunk, lo + per);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += c[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) { const int v = part[i]; part[i] = run; run += v; }
    tot[blockIdx.x] = run;
  }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int i = lo; i < hi; ++i) { const int v = c[i]; c[i] = run; run += v; }
}

struct GeomComp {       // one kept component
  int root, label, area, pad_;
};

// pass 3: every representative learns its label id (1 + raster rank among all components) and, when kept, its slot (raster
// rank among the kept ones).  area[root] is replaced by the slot (or -1) for the passes that follow.
__global__ __launch_bounds__(256) void k_geom_assign(size_t n_img, int nchunk, const int* __restrict__ Lb, int* __restrict__ area_b,
                                                     int min_area, const int* __restrict__ cnt, int max_comp, GeomComp* __restrict__ comps) {
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int* L = Lb + (size_t)b * n_img;
  int* A = area_b + (size_t)b * n_img;
  int base_r = cnt[((size_t)b * 2 + 0) * nchunk + chunk];
  int base_k = cnt[((size_t)b * 2 + 1) * nchunk + chunk];
  __shared__ int s[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < 4; ++k) {
    const size_t p = (size_t)chunk * 1024 + k * 256 + threadIdx.x;
    const bool root = p < n_img && L[p] == (int)p;
    const int area = root ? A[p] : 0;
    const bool keep = root && area >= min_area;
    const unsigned long long mr = __ballot(root), mk = __ballot(keep);
    if (lane == 0) { s[0][wave] = __popcll(mr); s[1][wave] = __popcll(mk); }
    __syncthreads();
    int off_r = base_r, off_k = base_k;
    for (int q = 0; q < wave; ++q) { off_r += s[0][q]; off_k += s[1][q]; }
    const unsigned long long below = lane ? ((1ull << lane) - 1ull) : 0ull;
    if (root) {
      const int label = off_r + __popcll(mr & below) + 1;
      int slot = -1;
      if (keep) {
        slot = off_k + __popcll(mk & below);
        if (slot < max_comp) comps[(size_t)b * max_comp + slot] = GeomComp{(int)p, label, area, 0};
        else slot = -2;                       // kept (stays in the clean mask) but beyond the detection list's capacity
      }
      A[p] = slot;
    }
    base_r += s[0][0] + s[0][1] + s[0][2] + s[0][3];
    base_k += s[1][0] + s[1][1] + s[1][2] + s[1][3];
    __syncthreads();
  }
}

// clean mask (every component with area >= min_area) + per-row extremes of the components that have a slot
__global__ __launch_bounds__(256) void k_geom_clean_rows(int h, int w, const int* __restrict__ Lb, const int* __restrict__ slot_b, int max_comp,
                                                         uint8_t* __restrict__ clean, int* __restrict__ rows /*[B][max_comp][h][2]*/) {
  const int lane = threadIdx.x & 63;
  const int x = blockIdx.x * 64 + lane;
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const size_t img = (size_t)blockIdx.z * h * w;
  const bool in = x < w && y < h;
  const int lab = in ? Lb[img + (size_t)y * w + x] : -1;
  const int slot = lab >= 0 ? slot_b[img + lab] : -1;
  if (in) clean[img + (size_t)y * w + x] = (slot >= 0 || slot == -2) ? 255 : 0;
  // runs of EQUAL slot inside the wave (two kept components can be horizontal neighbours only through background, but a kept
  // and a dropped one may alternate): compare with the left neighbour's slot
  const int left = __shfl_up(slot, 1, 64);
  const bool start = slot >= 0 && (lane == 0 || left != slot);
  const int right = __shfl_down(slot, 1, 64);
  const bool end = slot >= 0 && (lane == 63 || right != slot);
  if (slot < 0) return;
  int* r = rows + (((size_t)blockIdx.z * max_comp + slot) * h + y) * 2;
  if (start) atomicMin(r, x);
  if (end) atomicMax(r + 1, x);
}

__global__ __launch_bounds__(256) void k_geom_rows_init(size_t pairs, int* __restrict__ rows) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) {
    rows[2 * i] = INT_MAX;       // left extreme
    rows[2 * i + 1] = -1;        // right extreme
  }
}

// ---------------------------------------------------------------------------------------------- steps 4-5
struct GeomRectOut {      // mirrors vk_geom_det (include/vk_unet.h)
  int label, area;
  int box[8];
  float cx, cy, rw, rh, ux, uy;
  int hull_n, pad_;
  double d1, d2, d_mean;
};
static_assert(sizeof(GeomRectOut) == sizeof(vk_geom_det), "vk_geom_det layout");

constexpr int GEOM_MAX_H = 4096;

__global__ __launch_bounds__(256) void k_geom_rect(int h, int max_comp, const int* __restrict__ tot, const GeomComp* __restrict__ comps,
                                                   const int* __restrict__ rows, GeomRectOut* __restrict__ out, int* __restrict__ counts) {
#pragma clang fp contract(off)
  const int b = blockIdx.y, slot = blockIdx.x;
  const int nkept = tot[b * 2 + 1];
  if (slot == 0 && threadIdx.x == 0) counts[b] = nkept;
  if (slot >= nkept || slot >= max_comp) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* xl = (int*)smem;                         // [GEOM_MAX_H] left extreme per row, later the hull x
  int* xr = xl + GEOM_MAX_H;                    // [GEOM_MAX_H]
  int* hx = xr + GEOM_MAX_H;                    // [2 * GEOM_MAX_H] hull vertices
  int* hy = hx + 2 * GEOM_MAX_H;
  __shared__ int s_y0, s_y1, s_nl, s_nr, s_m;
  __shared__ float s_area[4];
  __shared__ int s_idx[4];
  const int* r = rows + ((size_t)b * max_comp + slot) * h * 2;
  if (threadIdx.x == 0) { s_y0 = INT_MAX; s_y1 = -1; }
  __syncthreads();
  int y0 = INT_MAX, y1 = -1;
  for (int y = threadIdx.x; y < h; y += 256) {
    const int a = r[2 * y], c = r[2 * y + 1];
    xl[y] = a; xr[y] = c;
    if (a <= c) { y0 = min(y0, y); y1 = max(y1, y); }
  }
  atomicMin(&s_y0, y0);
  atomicMax(&s_y1, y1);
  __syncthreads();
  y0 = s_y0; y1 = s_y1;
  // a connected component occupies every row between its first and last one.  Two lanes build the two chains:
  //   left chain, top -> bottom, must turn LEFT-bulging: keep p when cross(a, b, p) > 0 with image coordinates (y down)
  //   right chain, bottom -> top, likewise
  // chains are written into hx/hy: left from index 0 upwards, right into the second half
  if (threadIdx.x == 0) {
    int n = 0;
    for (int y = y0; y <= y1; ++y) {
      const int px = xl[y];
      while (n >= 2) {
        const long c = (long)(hx[n - 1] - hx[n - 2]) * (y - hy[n - 2]) - (long)(hy[n - 1] - hy[n - 2]) * (px - hx[n - 2]);
        if (c >= 0) --n; else break;            // going down the left side the hull turns clockwise on screen: cross < 0 keeps
      }
      hx[n] = px; hy[n] = y; ++n;
    }
    s_nl = n;
  } else if (threadIdx.x == 64) {
    int* rx = hx + GEOM_MAX_H;
    int* ry = hy + GEOM_MAX_H;
    int n = 0;
    for (int y = y1; y >= y0; --y) {
      const int px = xr[y];
      while (n >= 2) {
        const long c = (long)(rx[n - 1] - rx[n - 2]) * (y - ry[n - 2]) - (long)(ry[n - 1] - ry[n - 2]) * (px - rx[n - 2]);
        if (c >= 0) --n; else break;
      }
      rx[n] = px; ry[n] = y; ++n;
    }
    s_nr = n;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // join: left chain (top -> bottom) then right chain (bottom -> top); drop the right chain's first point when it equals the left
    // chain's last (single-pixel bottom row) and its last when it equals the left chain's first (single-pixel top row)
    int n = s_nl;
    const int nr = s_nr;
    const int* rx = hx + GEOM_MAX_H;
    const int* ry = hy + GEOM_MAX_H;
    int first = 0, last = nr;
    if (nr > 0 && rx[0] == hx[n - 1] && ry[0] == hy[n - 1]) first = 1;
    if (last > first && rx[last - 1] == hx[0] && ry[last - 1] == hy[0]) last -= 1;
    for (int i = first; i < last; ++i) { hx[n] = rx[i]; hy[n] = ry[i]; ++n; }
    // the junctions can leave a collinear middle point (e.g. top edge continuing a straight side): one cleaning pass over the
    // closed polygon keeps strictly convex vertices only (the canonical first vertex, top-most then left-most, always stays)
    if (n >= 3) {
      int m = 0;
      int* tx = xl;          // row tables are no longer needed
      int* ty = xr;
      for (int i = 0; i < n; ++i) {
        const int ax = hx[(i + n - 1) % n], ay = hy[(i + n - 1) % n], bx = hx[i], by = hy[i], cx = hx[(i + 1) % n], cy = hy[(i + 1) % n];
        const long c = (long)(bx - ax) * (cy - ay) - (long)(by - ay) * (cx - ax);
        if (c != 0 || i == 0) { tx[m] = bx; ty[m] = by; ++m; }
      }
      for (int i = 0; i < m; ++i) { hx[i] = tx[i]; hy[i] = ty[i]; }
      n = m;
    }
    s_m = n;
  }
  __syncthreads();
  const int m = s_m;
  // rotating calipers, every hull edge in parallel (float32, operation order of oracle/geometry_oracle.py min_area_rect)
  const int nedge = m == 1 ? 0 : (m == 2 ? 1 : m);
  float best = INFINITY;
  int besti = INT_MAX;
  for (int i = threadIdx.x; i < nedge; i += 256) {
    const int j = (i + 1) % m;
    const float dx = (float)(hx[j] - hx[i]), dy = (float)(hy[j] - hy[i]);
    const float ln = sqrtf(dx * dx + dy * dy);
    const float ux = dx / ln, uy = dy / ln;
    const float vx = -uy, vy = ux;
    float smin = INFINITY, smax = -INFINITY, tmin = INFINITY, tmax = -INFINITY;
    for (int k = 0; k < m; ++k) {
      const float fx = (float)hx[k], fy = (float)hy[k];
      const float s = fx * ux + fy * uy;
      const float t = fx * vx + fy * vy;
      smin = fminf(smin, s); smax = fmaxf(smax, s);
      tmin = fminf(tmin, t); tmax = fmaxf(tmax, t);
    }
    const float area = (smax - smin) * (tmax - tmin);
    if (area < best) { best = area; besti = i; }       // ascending i per thread: the first minimal edge of this thread
  }
  // argmin with the smallest index on ties
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if ((threadIdx.x & 63) == 0) { s_area[threadIdx.x >> 6] = best; s_idx[threadIdx.x >> 6] = besti; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (int q = 1; q < 4; ++q)
    if (s_area[q] < best || (s_area[q] == best && s_idx[q] < besti)) { best = s_area[q]; besti = s_idx[q]; }
  const GeomComp cmp = comps[(size_t)b * max_comp + slot];
  GeomRectOut o;
  o.label = cmp.label; o.area = cmp.area; o.hull_n = m; o.pad_ = 0;
  float cx, cy, ux = 1.f, uy = 0.f, rw = 0.f, rh = 0.f, cor[4][2];
  if (m == 1) {
    cx = (float)hx[0]; cy = (float)hy[0];
    for (int q = 0; q < 4; ++q) { cor[q][0] = cx; cor[q][1] = cy; }
  } else {
    const int i = besti, j = (besti + 1) % m;
    const float dx = (float)(hx[j] - hx[i]), dy = (float)(hy[j] - hy[i]);
    const float ln = sqrtf(dx * dx + dy * dy);
    ux = dx / ln; uy = dy / ln;
    const float vx = -uy, vy = ux;
    float smin = INFINITY, smax = -INFINITY, tmin = INFINITY, tmax = -INFINITY;
    for (int k = 0; k < m; ++k) {
      const float fx = (float)hx[k], fy = (float)hy[k];
      const float s = fx * ux + fy * uy;
      const float t = fx * vx + fy * vy;
      smin = fminf(smin, s); smax = fmaxf(smax, s);
      tmin = fminf(tmin, t); tmax = fmaxf(tmax, t);
    }
    const float sc = (smin + smax) * 0.5f, tc = (tmin + tmax) * 0.5f;
    cx = sc * ux + tc * vx;
    cy = sc * uy + tc * vy;
    rw = smax - smin; rh = tmax - tmin;
    const float ab[4][2] = {{smin, tmin}, {smax, tmin}, {smax, tmax}, {smin, tmax}};
    for (int q = 0; q < 4; ++q) {
      cor[q][0] = ab[q][0] * ux + ab[q][1] * vx;
      cor[q][1] = ab[q][0] * uy + ab[q][1] * vy;
    }
  }
  o.cx = cx; o.cy = cy; o.rw = rw; o.rh = rh; o.ux = ux; o.uy = uy;
  for (int q = 0; q < 4; ++q) { o.box[2 * q] = (int)cor[q][0]; o.box[2 * q + 1] = (int)cor[q][1]; }     // astype(int32): truncation
  // diagonals (ui_infer_rectangle.py:352-364): longest of the six pair distances (first in (0,1),(0,2),(0,3),(1,2),(1,3),(2,3) on
  // ties: Python's stable sort), the other two corners give the second
  double dmax = -1.0;
  int i1 = 0, j1 = 1;
  for (int a = 0; a < 4; ++a)
    for (int c = a + 1; c < 4; ++c) {
      const double ddx = (double)(o.box[2 * a] - o.box[2 * c]), ddy = (double)(o.box[2 * a + 1] - o.box[2 * c + 1]);
      const double d = sqrt(ddx * ddx + ddy * ddy);
      if (d > dmax) { dmax = d; i1 = a; j1 = c; }
    }
  int rest[2], nr = 0;
  for (int q = 0; q < 4; ++q)
    if (q != i1 && q != j1) rest[nr++] = q;
  const double ex = (double)(o.box[2 * rest[0]] - o.box[2 * rest[1]]), ey = (double)(o.box[2 * rest[0] + 1] - o.box[2 * rest[1] + 1]);
  o.d1 = dmax;
  o.d2 = sqrt(ex * ex + ey * ey);
  o.d_mean = 0.5 * (o.d1 + o.d2);
  out[(size_t)b * max_comp + slot] = o;
}


// ---------------------------------------------------------------------------------------------- the 4-vertex fit
// ui_infer_quadrilateral.py:262-530 (the newer GUI): per kept component, dilate by the fit_outset_px ellipse, trace the external
// border (cv2.findContours RETR_EXTERNAL / CHAIN_APPROX_SIMPLE: Suzuki-Abe border following), take the convex hull, run
// cv2.approxPolyDP with an epsilon bisection on both polygons until exactly four vertices come out, fall back to four consecutive
// vertices of a 1 % approximation and then to the hull's extreme points, pick the candidate with the best (quality, area) key,
// order it clockwise and measure its diagonals.  oracle/quad_oracle.py restates the same steps in numpy; both sides use the same
// types in the same operation order (integer-valued coordinates: every Douglas-Peucker distance is exact in float64).
//
// One WAVE per component (the control flow of border following and of the Douglas-Peucker stack is sequential; the 64 lanes share
// the loads of a tracing step and the arg-max scans over a polygon).  Everything that decides control flow is wave-uniform.
struct GeomQuadOut {      // mirrors vk_geom_quad (include/vk_unet.h)
  int label, area;
  int box[8];
  float cx, cy;
  int valid, branch, n_candidates, contour_n, hull_n, flags;
  double quality, d1, d2, d_mean;
};
static_assert(sizeof(GeomQuadOut) == sizeof(vk_geom_quad), "vk_geom_quad layout");

constexpr int GQ_NC = 16384;         // capacity of the contour and of the approximation buffers (points)
constexpr int GQ_NH = 4096;          // capacity of the hull
constexpr int GQ_LDS = (2 * GQ_NC + GQ_NH) * 4;
constexpr int GQ_MAX_STEPS = 1 << 21;

// chain-code steps: 0..7 = E, NE, N, NW, W, SW, S, SE with y growing downwards (OpenCV's CV_INIT_3X3_DELTAS order)
__device__ __forceinline__ int gq_dx(int s) { return (int)((0x901Au >> (2 * s)) & 3u) - 1; }
__device__ __forceinline__ int gq_dy(int s) { return (int)((0xA901u >> (2 * s)) & 3u) - 1; }
__device__ __forceinline__ uint32_t gq_pack(int x, int y) { return (uint32_t)x | ((uint32_t)y << 16); }
__device__ __forceinline__ int gq_x(uint32_t p) { return (int)(p & 0xFFFFu); }
__device__ __forceinline__ int gq_y(uint32_t p) { return (int)(p >> 16); }

// wave arg-max of (value, smallest key among equal values); every lane returns the winner
__device__ __forceinline__ void gq_wave_argmax(double& v, int& key) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(v, o, 64);
    const int ok = __shfl_xor(key, o, 64);
    if (ov > v || (ov == v && ok < key)) { v = ov; key = ok; }
  }
}

// cv2.arcLength(poly, closed): float32 segment lengths summed in float64.  Lengths are >= 1 with 24-bit mantissas and the sum stays
// below 2^20, so every partial sum is exact in float64 and the order of the additions cannot matter.
__device__ double gq_arc_length(const uint32_t* src, int count) {
#pragma clang fp contract(off)
  if (count <= 1) return 0.0;
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += 64) {
    const uint32_t a = src[i == 0 ? count - 1 : i - 1], b = src[i];
    const float dx = (float)(gq_x(b) - gq_x(a)), dy = (float)(gq_y(b) - gq_y(a));
    s += (double)sqrtf(dx * dx + dy * dy);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  return s;
}

// cv2.approxPolyDP(src, epsilon, closed = true) -> dst[0 .. return value).  buf has GQ_NC words: output points grow from the bottom,
// the slice stack (start | end << 16) from the top; they cannot meet (every stacked slice still owes at least one output point).
__device__ int gq_approx_closed(const uint32_t* src, int count, double epsilon, uint32_t* buf) {
#pragma clang fp contract(off)
  if (count == 0) return 0;
  const double eps = epsilon * epsilon;
  int n = 0, top = GQ_NC;
  // 1. three farthest-point sweeps
  int pos = 0, right_start = 0;
  bool le_eps = false;
  for (int it = 0; it < 3; ++it) {
    pos = (pos + right_start) % count;
    const uint32_t sp = src[pos];
    const double sx = (double)gq_x(sp), sy = (double)gq_y(sp);
    double best = 0.0;
    int bj = 0x7FFFFFFF;
    for (int j = 1 + (int)threadIdx.x; j < count; j += 64) {
      int q = pos + j; if (q >= count) q -= count;
      const uint32_t pt = src[q];
      const double dx = (double)gq_x(pt) - sx, dy = (double)gq_y(pt) - sy;
      const double d = dx * dx + dy * dy;
      if (d > best) { best = d; bj = j; }
    }
    gq_wave_argmax(best, bj);
    if (best > 0.0) right_start = bj;       // "dist > max_dist" never fires on all-zero distances: right_start keeps its value
    le_eps = best <= eps;
  }
  if (!le_eps) {
    const int a = pos % count, b = (right_start + a) % count;
    buf[--top] = (uint32_t)b | ((uint32_t)a << 16);      // right slice
    buf[--top] = (uint32_t)a | ((uint32_t)b << 16);
  } else {
    buf[n++] = src[pos];
  }
  // 2. split slices at their farthest point
  while (top < GQ_NC) {
    const uint32_t sl = buf[top++];
    const int s0 = (int)(sl & 0xFFFFu), s1 = (int)(sl >> 16);
    const uint32_t sp = src[s0], ep = src[s1];
    int first = s0 + 1; if (first >= count) first = 0;
    bool le = true;
    int split = 0;
    if (first != s1) {
      const double sx = (double)gq_x(sp), sy = (double)gq_y(sp);
      const double dx = (double)gq_x(ep) - sx, dy = (double)gq_y(ep) - sy;
      int len = s1 - first; if (len < 0) len += count;          // points strictly between the slice ends
      double best = 0.0;
      int bk = 0x7FFFFFFF;
      for (int k = threadIdx.x; k < len; k += 64) {
        int q = first + k; if (q >= count) q -= count;
        const uint32_t pt = src[q];
        const double d = fabs(((double)gq_y(pt) - sy) * dx - ((double)gq_x(pt) - sx) * dy);
        if (d > best) { best = d; bk = k; }
      }
      gq_wave_argmax(best, bk);
      if (best > 0.0) { split = first + bk; if (split >= count) split -= count; }
      le = best * best <= eps * (dx * dx + dy * dy);
    }
    if (le) {
      buf[n++] = sp;
    } else {
      buf[--top] = (uint32_t)split | ((uint32_t)s1 << 16);
      buf[--top] = (uint32_t)s0 | ((uint32_t)split << 16);
    }
  }
  // 3. in-place clean-up of nearly straight runs (sequential; reads near the wrap-around see rewritten entries, as in OpenCV)
  const int cnt = n;
  int new_count = n;
  if (cnt == 0) return 0;
  int rp = cnt - 1;
  uint32_t start_pt = buf[rp]; rp = rp + 1 >= cnt ? 0 : rp + 1;
  int wpos = rp;
  uint32_t pt = buf[rp]; rp = rp + 1 >= cnt ? 0 : rp + 1;
  for (int i = 0; i < cnt && new_count > 2; ++i) {
    const uint32_t end_pt = buf[rp]; rp = rp + 1 >= cnt ? 0 : rp + 1;
    const double dx = (double)(gq_x(end_pt) - gq_x(start_pt)), dy = (double)(gq_y(end_pt) - gq_y(start_pt));
    const double px = (double)(gq_x(pt) - gq_x(start_pt)), py = (double)(gq_y(pt) - gq_y(start_pt));
    const double dist = fabs(px * dy - py * dx);
    const double inner = px * (double)(gq_x(end_pt) - gq_x(pt)) + py * (double)(gq_y(end_pt) - gq_y(pt));
    if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0.0 && dy != 0.0 && inner >= 0.0) {
      --new_count;
      buf[wpos] = start_pt = end_pt;
      wpos = wpos + 1 >= cnt ? 0 : wpos + 1;
      pt = buf[rp]; rp = rp + 1 >= cnt ? 0 : rp + 1;
      ++i;
      continue;
    }
    buf[wpos] = start_pt = pt;
    wpos = wpos + 1 >= cnt ? 0 : wpos + 1;
    pt = end_pt;
  }
  return new_count;
}

struct GqQuad { float x[4], y[4]; };

// _order_quad_cw (:266-277)
__device__ GqQuad gq_order_cw(const GqQuad& q) {
#pragma clang fp contract(off)
  const float cx = (((q.x[0] + q.x[1]) + q.x[2]) + q.x[3]) / 4.f, cy = (((q.y[0] + q.y[1]) + q.y[2]) + q.y[3]) / 4.f;
  float ang[4];
  int idx[4] = {0, 1, 2, 3};
  for (int i = 0; i < 4; ++i) ang[i] = atan2f(q.y[i] - cy, q.x[i] - cx);
  for (int i = 1; i < 4; ++i)                    // stable ascending insertion sort
    for (int j = i; j > 0 && ang[idx[j]] < ang[idx[j - 1]]; --j) { const int t = idx[j]; idx[j] = idx[j - 1]; idx[j - 1] = t; }
  GqQuad r;
  for (int i = 0; i < 4; ++i) { r.x[i] = q.x[idx[3 - i]]; r.y[i] = q.y[idx[3 - i]]; }
  int k = 0;
  for (int i = 1; i < 4; ++i)
    if (r.y[i] < r.y[k] || (r.y[i] == r.y[k] && r.x[i] < r.x[k])) k = i;
  GqQuad o;
  for (int i = 0; i < 4; ++i) { o.x[i] = r.x[(i + k) & 3]; o.y[i] = r.y[(i + k) & 3]; }
  return o;
}

// _poly_area (:298-301): float32 shoelace, the two dot products summed left to right
__device__ double gq_area(const GqQuad& q) {
#pragma clang fp contract(off)
  float s1 = 0.f, s2 = 0.f;
  for (int i = 0; i < 4; ++i) {
    s1 = s1 + q.x[i] * q.y[(i + 1) & 3];
    s2 = s2 + q.y[i] * q.x[(i + 1) & 3];
  }
  return fabs((double)(s1 - s2)) * 0.5;
}

// _is_convex_quad (:280-295)
__device__ bool gq_convex(const GqQuad& q) {
#pragma clang fp contract(off)
  bool all_ge = true, all_le = true;
  for (int i = 0; i < 4; ++i) {
    const int b = (i + 1) & 3, c = (i + 2) & 3;
    const float v1x = q.x[b] - q.x[i], v1y = q.y[b] - q.y[i], v2x = q.x[c] - q.x[b], v2y = q.y[c] - q.y[b];
    const float cr = v1x * v2y - v1y * v2x;
    all_ge = all_ge && cr >= 0.f;
    all_le = all_le && cr <= 0.f;
  }
  return all_ge || all_le;
}

// _quad_quality (:304-330)
__device__ double gq_quality(const GqQuad& q) {
#pragma clang fp contract(off)
  float d[4];
  for (int i = 0; i < 4; ++i) {
    const float vx = q.x[i] - q.x[(i + 1) & 3], vy = q.y[i] - q.y[(i + 1) & 3];
    d[i] = sqrtf(vx * vx + vy * vy);
  }
  const float peri = (((d[0] + d[1]) + d[2]) + d[3]) + 1e-6f;
  double pen = 0.0;
  for (int i = 0; i < 4; ++i) {
    const int a = (i + 3) & 3, c = (i + 1) & 3;
    const float v1x = q.x[a] - q.x[i], v1y = q.y[a] - q.y[i], v2x = q.x[c] - q.x[i], v2y = q.y[c] - q.y[i];
    const float dot = v1x * v2x + v1y * v2y;
    const float n1 = sqrtf(v1x * v1x + v1y * v1y), n2 = sqrtf(v2x * v2x + v2y * v2y);
    float cs = dot / (n1 * n2 + 1e-6f);
    cs = fminf(fmaxf(cs, -1.f), 1.f);
    const double ang = acos((double)cs) * (180.0 / 3.14159265358979323846);
    pen += (ang >= 15.0 && ang <= 165.0) ? 0.0 : 1.0;
  }
  const double ang_pen = pen / 4.0;
  const float dmax = fmaxf(fmaxf(d[0], d[1]), fmaxf(d[2], d[3])), dmin = fminf(fminf(d[0], d[1]), fminf(d[2], d[3]));
  const float ratio = (dmax + 1e-6f) / (dmin + 1e-6f);
  const float dev = fabsf(ratio - 1.f);
  const float ed_pen = dev < 1.f ? dev : 1.f;             // min(1.0, |ratio - 1|)
  const float shape = 1.f - 0.5f * ed_pen;
  const float size = peri / (peri + 1000.f);
  return (1.0 - 0.5 * ang_pen) * (double)shape * (double)size;
}

struct GqBest {
  bool have;
  GqQuad q;
  double quality, area;
  int ncand;
};
__device__ __forceinline__ void gq_offer(GqBest& b, const GqQuad& cand) {     // stable sort, reverse=True, [0]: the first maximal key
  const double ql = gq_quality(cand), ar = gq_area(cand);
  ++b.ncand;
  if (!b.have || ql > b.quality || (ql == b.quality && ar > b.area)) { b.have = true; b.q = cand; b.quality = ql; b.area = ar; }
}
__device__ __forceinline__ GqQuad gq_from(const uint32_t* p, int i0, int i1, int i2, int i3) {
  GqQuad q;
  const int ix[4] = {i0, i1, i2, i3};
  for (int i = 0; i < 4; ++i) { q.x[i] = (float)gq_x(p[ix[i]]); q.y[i] = (float)gq_y(p[ix[i]]); }
  return q;
}

// _try_poly_dp (:352-378): epsilon bisection on one polygon; true + the ordered quadrilateral when one was found
__device__ bool gq_try_poly(const uint32_t* src, int count, uint32_t* buf, GqQuad* out) {
#pragma clang fp contract(off)
  const double peri = gq_arc_length(src, count);
  double lo = 0.001 * peri, hi = 0.08 * peri;
  for (int it = 0; it < 25; ++it) {
    const double mid = 0.5 * (lo + hi);
    const int n = gq_approx_closed(src, count, mid, buf);
    if (n == 4) {
      const GqQuad cand = gq_order_cw(gq_from(buf, 0, 1, 2, 3));
      if (gq_area(cand) > 10.0 && gq_convex(cand)) { *out = cand; return true; }
      lo = mid;
    } else if (n > 4) {
      lo = mid;
    } else {
      hi = mid;
    }
    if (fabs(hi - lo) < 1e-6) break;
  }
  return false;
}

__global__ __launch_bounds__(64) void k_geom_quad(int h, int w, int max_comp, const int* __restrict__ tot, const GeomComp* __restrict__ comps,
                                                  const int* __restrict__ rows, const int* __restrict__ Lb, const GeomSE se,
                                                  GeomQuadOut* __restrict__ out, int* __restrict__ counts) {
#pragma clang fp contract(off)
  const int b = blockIdx.y, slot = blockIdx.x;
  const int nkept = tot[b * 2 + 1];
  if (slot == 0 && threadIdx.x == 0) counts[b] = nkept;
  if (slot >= nkept || slot >= max_comp) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* cont = (uint32_t*)smem;              // [GQ_NC] contour points (x | y << 16)
  uint32_t* buf = cont + GQ_NC;                  // [GQ_NC] approximation output + slice stack; scratch before that
  uint32_t* hull = buf + GQ_NC;                  // [GQ_NH]
  const int lane = threadIdx.x;
  const GeomComp cmp = comps[(size_t)b * max_comp + slot];
  const int* L = Lb + (size_t)b * h * w;
  const int* r = rows + ((size_t)b * max_comp + slot) * h * 2;
  const int R = se.r;
  GeomQuadOut o;
  o.label = cmp.label; o.area = cmp.area;
  for (int i = 0; i < 8; ++i) o.box[i] = 0;
  o.cx = o.cy = 0.f; o.valid = 0; o.branch = 0; o.n_candidates = 0; o.contour_n = 0; o.hull_n = 0; o.flags = 0;
  o.quality = 0.0; o.d1 = o.d2 = o.d_mean = 0.0;

  // ---- row extremes of the dilated component: the dilation of the component's row extremes.  buf[y] = xl | xr << 16 (xl > xr: empty)
  int y0 = 0x7FFFFFFF, y1 = -1;
  for (int y = lane; y < h; y += 64) {
    int xl = 0x7FFF, xr = -1;
    for (int dy = -R; dy <= R; ++dy) {
      const int yy = y + dy;
      if ((unsigned)yy >= (unsigned)h) continue;
      const int a = r[2 * yy], c = r[2 * yy + 1];
      if (a > c) continue;
      const int hwid = se.hw[dy + R];
      xl = min(xl, max(a - hwid, 0));
      xr = max(xr, min(c + hwid, w - 1));
    }
    if (xl <= xr) { y0 = min(y0, y); y1 = max(y1, y); buf[y] = (uint32_t)xl | ((uint32_t)xr << 16); }
    else buf[y] = 0x0000FFFFu;               // xl = 65535 > xr = 0
  }
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) { y0 = min(y0, __shfl_xor(y0, ofs, 64)); y1 = max(y1, __shfl_xor(y1, ofs, 64)); }
  // ---- hull (counter-clockwise on screen from the top-left: left chain down, right chain up), as k_geom_rect builds it
  uint32_t* lch = cont;                          // chains in the contour buffer (free until the trace)
  uint32_t* rch = cont + GQ_NC / 2;
  int nl = 0, nr = 0;
  for (int y = y0; y <= y1; ++y) {
    const int px = (int)(buf[y] & 0xFFFFu);
    while (nl >= 2) {
      const uint32_t p1 = lch[nl - 1], p2 = lch[nl - 2];
      const long c = (long)(gq_x(p1) - gq_x(p2)) * (y - gq_y(p2)) - (long)(gq_y(p1) - gq_y(p2)) * (px - gq_x(p2));
      if (c >= 0) --nl; else break;
    }
    lch[nl++] = gq_pack(px, y);
  }
  for (int y = y1; y >= y0; --y) {
    const int px = (int)(buf[y] >> 16);
    while (nr >= 2) {
      const uint32_t p1 = rch[nr - 1], p2 = rch[nr - 2];
      const long c = (long)(gq_x(p1) - gq_x(p2)) * (y - gq_y(p2)) - (long)(gq_y(p1) - gq_y(p2)) * (px - gq_x(p2));
      if (c >= 0) --nr; else break;
    }
    rch[nr++] = gq_pack(px, y);
  }
  const int x_start = (int)(buf[y0] & 0xFFFFu);          // raster-first pixel of the dilated component: (x_start, y0)
  int first = 0, last = nr;
  if (nr > 0 && rch[0] == lch[nl - 1]) first = 1;
  if (last > first && rch[last - 1] == lch[0]) last -= 1;
  int n = nl;
  for (int i = first; i < last; ++i) lch[n++] = rch[i];    // n <= 2 h <= GQ_NC / 2: stays inside the left half ... or runs into rch[i' < i]: already consumed
  int m = n;
  if (n >= 3) {          // drop collinear junction points (the first vertex always stays); result into buf, then hull
    m = 0;
    for (int i = 0; i < n; ++i) {
      const uint32_t A = lch[(i + n - 1) % n], Bp = lch[i], Cp = lch[(i + 1) % n];
      const long c = (long)(gq_x(Bp) - gq_x(A)) * (gq_y(Cp) - gq_y(A)) - (long)(gq_y(Bp) - gq_y(A)) * (gq_x(Cp) - gq_x(A));
      if (c != 0 || i == 0) buf[m++] = Bp;
    }
  } else {
    for (int i = 0; i < n; ++i) buf[i] = lch[i];
  }
  // cv2.convexHull order: from the right-most (then bottom-most) vertex, clockwise on screen = the list above reversed and rotated
  bool hull_ok = m <= GQ_NH;
  if (!hull_ok) o.flags |= 2;
  if (hull_ok) {
    int st = 0;
    for (int i = 1; i < m; ++i) {
      const uint32_t a = buf[i], c = buf[st];
      if (gq_x(a) > gq_x(c) || (gq_x(a) == gq_x(c) && gq_y(a) > gq_y(c))) st = i;
    }
    if (m >= 3) {
      for (int i = lane; i < m; i += 64) { int q = st - i; if (q < 0) q += m; hull[i] = buf[q]; }
    } else if (m == 2) {        // two points: descending (x, y)
      hull[0] = buf[st]; hull[1] = buf[1 - st];
    } else if (m == 1) {
      hull[0] = buf[0];
    }
  }
  o.hull_n = m;

  // ---- external border of the dilated component (Suzuki-Abe as cv2 runs it); membership of the 8 neighbours of the current pixel
  // from ONE cooperative load of the (2R + 3)^2 window of the label image
  const int W3 = 2 * R + 3, ncell = W3 * W3;
  unsigned long long sem0[8], sem1[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    unsigned long long a0 = 0ull, a1 = 0ull;
    for (int dy = -R; dy <= R; ++dy)
      for (int dx = -se.hw[dy + R]; dx <= se.hw[dy + R]; ++dx) {
        const int cell = (gq_dy(s) + dy + R + 1) * W3 + (gq_dx(s) + dx + R + 1);
        if (cell < 64) a0 |= 1ull << cell; else a1 |= 1ull << (cell - 64);
      }
    sem0[s] = a0; sem1[s] = a1;
  }
  const int c0y = lane / W3 - (R + 1), c0x = lane % W3 - (R + 1);
  const int c1y = (lane + 64) / W3 - (R + 1), c1x = (lane + 64) % W3 - (R + 1);
  auto neighbours = [&](int x, int y) -> int {       // bit s set: the neighbour in direction s belongs to the dilated component
    bool in0 = false, in1 = false;
    if (lane < ncell) {
      const int yy = y + c0y, xx = x + c0x;
      in0 = (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w && L[(size_t)yy * w + xx] == cmp.root;
    }
    if (lane + 64 < ncell) {
      const int yy = y + c1y, xx = x + c1x;
      in1 = (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w && L[(size_t)yy * w + xx] == cmp.root;
    }
    const unsigned long long m0 = __ballot(in0), m1 = __ballot(in1);
    int nb = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const bool inside = (unsigned)(x + gq_dx(s)) < (unsigned)w && (unsigned)(y + gq_dy(s)) < (unsigned)h;   // the dilation exists inside the map only
      nb |= ((inside && ((m0 & sem0[s]) | (m1 & sem1[s])) != 0ull) ? 1 : 0) << s;
    }
    return nb;
  };
  __builtin_amdgcn_s_waitcnt(0);                 // the hull's LDS traffic is done before the contour buffer is reused
  __builtin_amdgcn_wave_barrier();
  int cn = 0;
  bool cont_ok = true;
  {
    const int x0 = x_start, yy0 = y0;
    int nb = neighbours(x0, yy0);
    int s = 4;
    // first neighbour, clockwise from W: NW, N, NE, E, SE, S, SW
    int found = -1;
    for (int t = 0; t < 7; ++t) { s = (s - 1) & 7; if ((nb >> s) & 1) { found = s; break; } }
    if (found < 0) {
      cont[cn++] = gq_pack(x0, yy0);             // single pixel
    } else {
      const int x1 = x0 + gq_dx(s), yy1 = yy0 + gq_dy(s);
      int x3 = x0, y3 = yy0, px = x0, py = yy0;
      int prev_s = s ^ 4;
      for (int step = 0;; ++step) {
        if (step > 0) nb = neighbours(x3, y3);
        // next border pixel: counter-clockwise from the direction after s
        const int k = (s + 1) & 7;
        const int rot = ((nb >> k) | (nb << (8 - k))) & 0xFF;
        s = (k + __builtin_ctz(rot | 0x100)) & 7;
        const int x4 = x3 + gq_dx(s), y4 = y3 + gq_dy(s);
        if (s != prev_s) {
          if (cn < GQ_NC) cont[cn] = gq_pack(px, py);
          ++cn;
          prev_s = s;
        }
        px += gq_dx(s); py += gq_dy(s);
        if (x4 == x0 && y4 == yy0 && x3 == x1 && y3 == yy1) break;
        if (step >= GQ_MAX_STEPS || rot == 0) { cont_ok = false; o.flags |= 4; break; }
        x3 = x4; y3 = y4;
        s = (s + 4) & 7;
      }
    }
    if (cn > GQ_NC) { cont_ok = false; o.flags |= 1; }
  }
  o.contour_n = cn;
  __builtin_amdgcn_wave_barrier();

  // ---- robust_quadrilateral_from_contour (:336-420)
  GqBest best;
  best.have = false; best.quality = 0.0; best.area = 0.0; best.ncand = 0;
  const int npts = cont_ok ? cn : 0;
  const bool enough = cn >= 4;                    // `if pts.shape[0] < 4: return None`
  if (enough) {
    GqQuad cand;
    if (cont_ok && gq_try_poly(cont, npts, buf, &cand)) gq_offer(best, cand);
    if (hull_ok && gq_try_poly(hull, m, buf, &cand)) gq_offer(best, cand);
    o.branch = 1;
    if (!best.have) {
      o.branch = 2;
      for (int which = 0; which < 2; ++which) {
        const uint32_t* src = which == 0 ? cont : hull;
        const int cnt = which == 0 ? npts : (hull_ok ? m : 0);
        if (cnt == 0) continue;
        const int k = gq_approx_closed(src, cnt, 0.01 * gq_arc_length(src, cnt), buf);
        if (k > 4) {
          for (int s = 0; s < (k < 12 ? k : 12); ++s) {
            cand = gq_order_cw(gq_from(buf, s % k, (s + 1) % k, (s + 2) % k, (s + 3) % k));
            if (gq_area(cand) > 10.0 && gq_convex(cand)) gq_offer(best, cand);
          }
        }
      }
    }
    if (!best.have && hull_ok && m >= 1) {
      o.branch = 3;
      int iminx = 0, imaxx = 0, iminy = 0, imaxy = 0;
      for (int i = 1; i < m; ++i) {              // first occurrence of every extreme, like np.argmin / np.argmax
        if (gq_x(hull[i]) < gq_x(hull[iminx])) iminx = i;
        if (gq_x(hull[i]) > gq_x(hull[imaxx])) imaxx = i;
        if (gq_y(hull[i]) < gq_y(hull[iminy])) iminy = i;
        if (gq_y(hull[i]) > gq_y(hull[imaxy])) imaxy = i;
      }
      cand = gq_order_cw(gq_from(hull, iminy, imaxx, imaxy, iminx));
      if (gq_area(cand) > 10.0) gq_offer(best, cand);
    }
  }
  o.n_candidates = best.ncand;
  if (best.have) {
    const GqQuad q = gq_order_cw(best.q);
    o.valid = 1;
    o.quality = best.quality;
    int sx = 0, sy = 0;
    for (int i = 0; i < 4; ++i) { o.box[2 * i] = (int)q.x[i]; o.box[2 * i + 1] = (int)q.y[i]; sx += o.box[2 * i]; sy += o.box[2 * i + 1]; }
    o.cx = (float)((double)sx / 4.0);
    o.cy = (float)((double)sy / 4.0);
    double dmax = -1.0;
    int i1 = 0, j1 = 1;
    for (int a = 0; a < 4; ++a)
      for (int c = a + 1; c < 4; ++c) {
        const double ddx = (double)(o.box[2 * a] - o.box[2 * c]), ddy = (double)(o.box[2 * a + 1] - o.box[2 * c + 1]);
        const double d = sqrt(ddx * ddx + ddy * ddy);
        if (d > dmax) { dmax = d; i1 = a; j1 = c; }
      }
    int rest[2], nrr = 0;
    for (int qi = 0; qi < 4; ++qi)
      if (qi != i1 && qi != j1) rest[nrr++] = qi;
    const double ex = (double)(o.box[2 * rest[0]] - o.box[2 * rest[1]]), ey = (double)(o.box[2 * rest[0] + 1] - o.box[2 * rest[1] + 1]);
    o.d1 = dmax;
    o.d2 = sqrt(ex * ex + ey * ey);
    o.d_mean = 0.5 * (o.d1 + o.d2);
  } else {
    o.branch = 0;
  }
  if (lane == 0) out[(size_t)b * max_comp + slot] = o;
}

// ---------------------------------------------------------------------------------------------- host
static size_t al256(size_t v) { return (v + 255) / 256 * 256; }

struct GeomLayout {
  size_t m0, m1, L, area, cnt, tot, comps, rows, total;
  int nchunk;
};

static GeomLayout geom_layout(const vk_geom_desc* d, int batch) {
  GeomLayout g;
  const size_t n = (size_t)d->h * d->w;
  g.nchunk = (int)((n + 1023) / 1024);
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = al256(off + bytes); return o; };
  g.m0 = take(n * batch);
  g.m1 = take(n * batch);
  g.L = take(n * batch * sizeof(int));
  g.area = take(n * batch * sizeof(int));
  g.cnt = take((size_t)batch * 2 * g.nchunk * sizeof(int));
  g.tot = take((size_t)batch * 2 * sizeof(int));
  g.comps = take((size_t)batch * d->max_components * sizeof(GeomComp));
  g.rows = take((size_t)batch * d->max_components * d->h * 2 * sizeof(int));
  g.total = off;
  return g;
}

static int geom_check(const vk_geom_desc* d, int batch, const char* who) {
  VK_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  VK_CHECK_ARG(batch >= 1 && batch <= 65535, "%s: batch %d outside 1..65535", who, batch);
  VK_CHECK_ARG(d->h >= 1 && d->w >= 1 && d->h <= GEOM_MAX_H && d->w <= 16384, "%s: map size %dx%d outside 1..%d rows x 1..16384 columns", who,
               d->h, d->w, GEOM_MAX_H);
  VK_CHECK_ARG((size_t)d->h * d->w < (1u << 30), "%s: more than 2^30 pixels per map", who);
  VK_CHECK_ARG(d->morph_kernel >= 1 && d->morph_kernel <= 7 && (d->morph_kernel & 1), "%s: morph_kernel %d must be odd, 1..7", who, d->morph_kernel);
  VK_CHECK_ARG(d->open_iter >= 0 && d->close_iter >= 0 && d->open_iter <= 16 && d->close_iter <= 16, "%s: iterations outside 0..16", who);
  VK_CHECK_ARG(d->min_area >= 1, "%s: min_area %d < 1", who, d->min_area);
  VK_CHECK_ARG(d->max_components >= 1 && d->max_components <= 4096, "%s: max_components %d outside 1..4096", who, d->max_components);
  return VK_OK;
}

static GeomSE make_se(int k) {      // cv::getStructuringElement(MORPH_ELLIPSE, (k, k)) restated (see the oracle's header)
  GeomSE se;
  se.r = k / 2;
  for (int i = 0; i < 7; ++i) se.hw[i] = 0;
  const int r = k / 2, c = k / 2;
  const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
  for (int i = 0; i < k; ++i) {
    const int dy = i - r;
    int dx = (int)lrint(c * sqrt(((double)r * r - (double)dy * dy) * inv_r2));
    const int j1 = c - dx < 0 ? 0 : c - dx, j2 = c + dx + 1 > k ? k : c + dx + 1;
    se.hw[i] = (j2 - j1 - 1) / 2;           // symmetric about the centre column
  }
  return se;
}

}  // namespace vk

using namespace vk;

extern "C" int64_t vk_geom_workspace_bytes(const vk_geom_desc* d, int batch) {
  if (geom_check(d, batch, "vk_geom_workspace_bytes") != VK_OK) return -1;
  return (int64_t)geom_layout(d, batch).total;
}

// steps 1-3 + clean mask + per-row extremes of the kept components, shared by both GUIs' post-processing
static int geom_front(const vk_geom_desc* d, int batch, const float* prob, uint8_t* clean, char* ws, const GeomLayout& g, hipStream_t st) {
  const int h = d->h, w = d->w;
  const size_t n = (size_t)h * w;
  uint8_t* m0 = (uint8_t*)(ws + g.m0);
  uint8_t* m1 = (uint8_t*)(ws + g.m1);
  int* L = (int*)(ws + g.L);
  int* area = (int*)(ws + g.area);
  int* cnt = (int*)(ws + g.cnt);
  int* tot = (int*)(ws + g.tot);
  GeomComp* comps = (GeomComp*)(ws + g.comps);
  int* rows = (int*)(ws + g.rows);
  const dim3 grid2((w + 63) / 64, (h + 3) / 4, batch), blk(256);
  // steps 1-2: threshold fused into the first morphological pass
  const GeomSE se = make_se(d->morph_kernel);
  const bool morph = d->morph_kernel > 1;
  const int no = morph ? d->open_iter : 0, nc = morph ? d->close_iter : 0;
  const int passes = 2 * no + 2 * nc;
  uint8_t* cur = nullptr;
  if (passes == 0) {
    hipLaunchKernelGGL(k_geom_binarize, dim3((unsigned)std::min<size_t>((n * batch + 255) / 256, 4096)), blk, 0, st, n * batch, prob, d->bin_thresh, m0);
    cur = m0;
  } else {
    int pass = 0;
    auto run = [&](bool erode_op) {
      uint8_t* dst = (cur == m0) ? m1 : m0;
      if (pass == 0) {
        if (erode_op) hipLaunchKernelGGL((k_geom_morph<true, true>), grid2, blk, 0, st, h, w, se, (const void*)prob, d->bin_thresh, dst);
        else hipLaunchKernelGGL((k_geom_morph<false, true>), grid2, blk, 0, st, h, w, se, (const void*)prob, d->bin_thresh, dst);
      } else {
        if (erode_op) hipLaunchKernelGGL((k_geom_morph<true, false>), grid2, blk, 0, st, h, w, se, (const void*)cur, 0.f, dst);
        else hipLaunchKernelGGL((k_geom_morph<false, false>), grid2, blk, 0, st, h, w, se, (const void*)cur, 0.f, dst);
      }
      cur = dst;
      ++pass;
    };
    for (int i = 0; i < no; ++i) run(true);        // OPEN  = erode^n, dilate^n   (ui_infer_rectangle.py:329-330)
    for (int i = 0; i < no; ++i) run(false);
    for (int i = 0; i < nc; ++i) run(false);       // CLOSE = dilate^n, erode^n   (ui_infer_rectangle.py:331-332)
    for (int i = 0; i < nc; ++i) run(true);
  }
  // step 3
  VK_CHECK_HIP(hipMemsetAsync(area, 0, n * batch * sizeof(int), st));
  hipLaunchKernelGGL(k_geom_init, grid2, blk, 0, st, h, w, (const uint8_t*)cur, L);
  hipLaunchKernelGGL(k_geom_merge, grid2, blk, 0, st, h, w, L);
  hipLaunchKernelGGL(k_geom_flatten, dim3((unsigned)std::min<size_t>((n * batch + 255) / 256, 8192)), blk, 0, st, n, batch, L);
  hipLaunchKernelGGL(k_geom_runs_area, grid2, blk, 0, st, h, w, (const int*)L, area);
  hipLaunchKernelGGL(k_geom_count, dim3(g.nchunk, batch), blk, 0, st, n, g.nchunk, (const int*)L, (const int*)area, d->min_area, cnt);
  hipLaunchKernelGGL(k_geom_scan, dim3(2 * batch), blk, 0, st, g.nchunk, cnt, tot);
  hipLaunchKernelGGL(k_geom_assign, dim3(g.nchunk, batch), blk, 0, st, n, g.nchunk, (const int*)L, area, d->min_area, (const int*)cnt,
                     d->max_components, comps);
  {
    const size_t pairs = (size_t)batch * d->max_components * h;
    hipLaunchKernelGGL(k_geom_rows_init, dim3((unsigned)std::min<size_t>((pairs + 255) / 256, 4096)), blk, 0, st, pairs, rows);
  }
  hipLaunchKernelGGL(k_geom_clean_rows, grid2, blk, 0, st, h, w, (const int*)L, (const int*)area, d->max_components, clean, rows);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_geom_minarearect(const vk_geom_desc* d, int batch, const float* prob, uint8_t* clean, vk_geom_det* dets, int* counts,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  int rc = geom_check(d, batch, "vk_geom_minarearect");
  if (rc != VK_OK) return rc;
  VK_CHECK_ARG(prob && clean && dets && counts && workspace, "vk_geom_minarearect: null buffer");
  const GeomLayout g = geom_layout(d, batch);
  VK_CHECK_ARG(workspace_bytes >= g.total, "vk_geom_minarearect: workspace too small (%zu < %zu)", workspace_bytes, g.total);
  VK_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "vk_geom_minarearect: workspace must be 256-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  vkh::ProfScope ps("geom_minarearect", st, 0.0, (double)batch * d->h * d->w * (4.0 + 1.0));
  rc = geom_front(d, batch, prob, clean, ws, g, st);
  if (rc != VK_OK) return rc;
  // steps 4-5
  const size_t lds = (size_t)GEOM_MAX_H * 6 * sizeof(int);
  static bool attr_done = false;
  if (!attr_done) {
    VK_CHECK_HIP(hipFuncSetAttribute((const void*)k_geom_rect, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(k_geom_rect, dim3(d->max_components, batch), dim3(256), lds, st, d->h, d->max_components, (const int*)(ws + g.tot),
                     (const GeomComp*)(ws + g.comps), (const int*)(ws + g.rows), (GeomRectOut*)dets, counts);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}

extern "C" int vk_geom_quadrilateral(const vk_geom_desc* d, int fit_outset_px, int batch, const float* prob, uint8_t* clean, vk_geom_quad* dets,
                                     int* counts, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = geom_check(d, batch, "vk_geom_quadrilateral");
  if (rc != VK_OK) return rc;
  VK_CHECK_ARG(prob && clean && dets && counts && workspace, "vk_geom_quadrilateral: null buffer");
  VK_CHECK_ARG(fit_outset_px >= 0 && fit_outset_px <= 3, "vk_geom_quadrilateral: fit_outset_px %d outside 0..3", fit_outset_px);
  VK_CHECK_ARG(d->w <= 16384 && d->h <= GEOM_MAX_H, "vk_geom_quadrilateral: map too large for 16-bit packed coordinates");
  const GeomLayout g = geom_layout(d, batch);
  VK_CHECK_ARG(workspace_bytes >= g.total, "vk_geom_quadrilateral: workspace too small (%zu < %zu)", workspace_bytes, g.total);
  VK_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "vk_geom_quadrilateral: workspace must be 256-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  vkh::ProfScope ps("geom_quadrilateral", st, 0.0, (double)batch * d->h * d->w * (4.0 + 1.0));
  rc = geom_front(d, batch, prob, clean, ws, g, st);
  if (rc != VK_OK) return rc;
  // the fit dilates with getStructuringElement(MORPH_ELLIPSE, max(3, 2 * outset + 1)) (ui_infer_quadrilateral.py:476-479); 0 = no dilation
  GeomSE se_fit = make_se(fit_outset_px > 0 ? std::max(3, 2 * fit_outset_px + 1) : 1);
  static bool attr_done = false;
  if (!attr_done) {
    VK_CHECK_HIP(hipFuncSetAttribute((const void*)k_geom_quad, hipFuncAttributeMaxDynamicSharedMemorySize, GQ_LDS));
    attr_done = true;
  }
  hipLaunchKernelGGL(k_geom_quad, dim3(d->max_components, batch), dim3(64), (size_t)GQ_LDS, st, d->h, d->w, d->max_components,
                     (const int*)(ws + g.tot), (const GeomComp*)(ws + g.comps), (const int*)(ws + g.rows), (const int*)(ws + g.L), se_fit,
                     (GeomQuadOut*)dets, counts);
  VK_CHECK_HIP(hipGetLastError());
  return VK_OK;
}
