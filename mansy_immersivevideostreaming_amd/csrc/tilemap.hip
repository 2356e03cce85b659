// Per-tile viewport hit map + IoU, integer and bit-exact:
//   find_block_covered_by_point / find_tiles_covered_by_viewport / _find_regions_covered_by_fov
//   (viewport_prediction/utils/common.py:37-58, 83-127), pixel centre int(x*W), int(y*H)
//   (utils/results.py:15-18, predict.py:40-43), IoU (results.py:21, predict.py:46).
// One thread per (sample, t) point; output one uint64 per point (bit row*nw+col), coalesced 8-byte
// stores.  The wrap-around cases collapse to: columns/rows of the FoV rectangle taken modulo the
// frame, evaluated with the reference's "an exact multiple belongs to the lower tile" rule on each
// of the (<= 2 x 2) wrapped regions.
#include "mansy_kernels.h"

namespace {

__device__ __forceinline__ int pymod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }
// find_block_covered_by_point: Python's x // bw (FLOOR division, also for the negative pixels of a centre left of / above the
// frame), minus one when x is a positive exact multiple -- as one truncating division: (x - 1) / bw for x > 0,
// (x - (bw - 1)) / bw for x <= 0.
__device__ __forceinline__ int block_of(int x, int bw) { return (x > 0 ? x - 1 : x - (bw - 1)) / bw; }
// Tiles of one axis covered by the FoV interval [lo, hi] on a frame of `size` pixels, tiles `bw` wide, `n` of them.
// _find_regions_covered_by_fov splits the interval per axis exactly like this:
//   lo >= 0 && hi <= size : [lo, hi]
//   lo <  0 && hi <= size : [0, hi] and [lo % size, size]
//   lo >= 0 && hi >  size : [0, hi % size] and [lo, size]
// (neither -- lo < 0 && hi > size -- cannot happen for fov < size and yields nothing), and each interval [p, q] marks the numpy
// slice [block_of(p) : block_of(q) + 1] of the axis: a negative bound counts from the END of the axis and only then clips to 0, a
// bound past the end clips to n (tests/golden/tilemap_px_outside.npz: the imported function on centres outside the frame -- raw
// predictions reach it, predict.py:40-45).  With tA = block_of(lo or lo % size), tB = block_of(hi or hi % size),
// tS = block_of(size) and block_of(0) = 0:
//   one interval : bits [bound(tA), bound(tB + 1))
//   two intervals: bits [0, bound(tB + 1)) | [bound(tA), bound(tS + 1))
// -- two divisions per axis and selects instead of branches: neighbouring points fall into different cases, and a wave that takes
// every branch pays for all of them (round 3: the kernel was ALU-bound at ~290 integer instructions per point, a third of them
// executed for the other lanes' cases).
__device__ __forceinline__ int slice_bound(int i, int n) { return i < 0 ? max(i + n, 0) : min(i, n); }
__device__ __forceinline__ unsigned below(int k) { return k >= 32 ? 0xFFFFFFFFu : ((1u << (k & 31)) - 1u); }      // bits [0, k), 0 <= k <= 32
__device__ __forceinline__ unsigned span(int t1, int t2, int n) {          // numpy slice [t1 : t2 + 1] of an axis of n tiles (empty when reversed)
  return below(slice_bound(t2 + 1, n)) & ~below(slice_bound(t1, n));
}
__device__ __forceinline__ unsigned axis_mask(int lo, int hi, int size, int bw, int n) {
  const bool neg = lo < 0, over = hi > size;
  const int tA = block_of(neg ? pymod(lo, size) : lo, bw), tB = block_of(over ? pymod(hi, size) : hi, bw), tS = block_of(size, bw);
  const unsigned m = (neg != over) ? (span(0, tB, n) | span(tA, tS, n)) : span(tA, tB, n);
  return (neg && over) ? 0u : m;
}

__device__ __forceinline__ unsigned long long tilemap_px(int x, int y, int W, int H, int tw, int th, int nw, int nh, int fov_w, int fov_h) {
  const int hw = fov_w / 2, hh = fov_h / 2;
  const unsigned rows = axis_mask(y - hh, y + hh, H, th, nh), cols = axis_mask(x - hw, x + hw, W, tw, nw);
  // the covered regions are the cross product of the x intervals and the y intervals, so the union of their tile rectangles is
  // (union of the row masks) x (union of the column masks): one outer product
  if (nw == 8 && nh == 8) {
    // byte r of the map = cols if bit r of rows is set: replicate rows into every byte, keep bit r in byte r, turn "byte != 0" into 0xFF
    const unsigned long long t = ((unsigned long long)(rows & 0xFFu) * 0x0101010101010101ull) & 0x8040201008040201ull;
    const unsigned long long sel = (((t + 0x7F7F7F7F7F7F7F7Full) & 0x8080808080808080ull) >> 7) * 0xFFull;
    return sel & ((unsigned long long)(cols & 0xFFu) * 0x0101010101010101ull);
  }
  unsigned long long m = 0ull;
  for (int r = 0; r < nh; ++r)
    if ((rows >> r) & 1u) m |= (unsigned long long)cols << (r * nw);
  return m;
}

// STD: the geometry of config.yml (2560x1440 frame, 8x8 tiles, 600x300 FoV) as compile-time constants -- the ~10 integer
// divisions / modulos per point by run-time divisors (tile size, frame size) are what bounds the generic kernel (0.95 TB/s);
// with constant divisors and the row loop unrolled the kernel is a stream of 8-byte loads and stores.
template <bool STD>
__global__ __launch_bounds__(256) void tilemap_kernel(const float* __restrict__ xy, long long n, int W, int H, int nw, int nh, int fov_w,
                                                      int fov_h, unsigned long long* __restrict__ maps) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (STD) { W = 2560; H = 1440; nw = 8; nh = 8; fov_w = 600; fov_h = 300; }
  const float2 p = *reinterpret_cast<const float2*>(xy + 2 * i);
  // np.float32 * python int stays float32 under numpy>=2 (and under value-based casting): round to f32, then truncate
  const int px = (int)(p.x * (float)W), py = (int)(p.y * (float)H);
  maps[i] = tilemap_px(px, py, W, H, W / nw, H / nh, nw, nh, fov_w, fov_h);
}

__global__ __launch_bounds__(256) void tilemap_iou_kernel(const unsigned long long* __restrict__ a, const unsigned long long* __restrict__ b,
                                                          long long n, double* __restrict__ iou) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  iou[i] = (double)__popcll(a[i] & b[i]) / (double)__popcll(a[i] | b[i]);
}

__global__ __launch_bounds__(256) void tilemap_or_kernel(const unsigned long long* __restrict__ maps, long long ngroups, int group,
                                                         unsigned long long* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= ngroups) return;
  unsigned long long m = 0ull;
  for (int k = 0; k < group; ++k) m |= maps[i * group + k];
  out[i] = m;
}

// accuracy (IoU), recall, precision, f1 of results.py:21-31 from two maps; doubles [n,4]
__global__ __launch_bounds__(256) void tilemap_metrics_kernel(const unsigned long long* __restrict__ gt, const unsigned long long* __restrict__ pred,
                                                              long long n, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double tp = (double)__popcll(gt[i] & pred[i]);
  const double uni = (double)__popcll(gt[i] | pred[i]);
  const double fp = (double)__popcll(pred[i]) - tp, fn = (double)__popcll(gt[i]) - tp;
  const double recall = tp / (tp + fn), precision = tp / (tp + fp);
  out[4 * i + 0] = tp / uni;
  out[4 * i + 1] = recall;
  out[4 * i + 2] = precision;
  out[4 * i + 3] = (recall + precision == 0.0) ? 0.0 : recall * precision * 2.0 / (recall + precision);
}

}  // namespace

int mansy_launch_tilemap_metrics(const unsigned long long* gt, const unsigned long long* pred, long long n, double* out, hipStream_t st) {
  if (n <= 0) return MANSY_OK;                      // empty input: nothing to do (pointers of empty buffers may be null)
  MANSY_REQUIRE(gt && pred && out, "tilemap_metrics: null pointer");
  MANSY_LAUNCH(tilemap_metrics_kernel, dim3(mansy_ceil_div(n, 256)), dim3(256), 0, st, gt, pred, n, out);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_tilemap(const float* xy, long long n, int W, int H, int nw, int nh, int fov_w, int fov_h, unsigned long long* maps,
                         hipStream_t st) {
  MANSY_REQUIRE(nw >= 1 && nh >= 1 && nw * nh <= 64 && nw <= 32 && nh <= 32, "tilemap: grid %dx%d does not fit a uint64 map", nw, nh);
  MANSY_REQUIRE(fov_w < W && fov_h < H, "tilemap: FoV must be smaller than the frame");
  if (n <= 0) return MANSY_OK;
  MANSY_REQUIRE(xy && maps, "tilemap: null pointer");
  if (W == 2560 && H == 1440 && nw == 8 && nh == 8 && fov_w == 600 && fov_h == 300)
    MANSY_LAUNCH(tilemap_kernel<true>, dim3(mansy_ceil_div(n, 256)), dim3(256), 0, st, xy, n, W, H, nw, nh, fov_w, fov_h, maps);
  else
    MANSY_LAUNCH(tilemap_kernel<false>, dim3(mansy_ceil_div(n, 256)), dim3(256), 0, st, xy, n, W, H, nw, nh, fov_w, fov_h, maps);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_tilemap_iou(const unsigned long long* a, const unsigned long long* b, long long n, double* iou, hipStream_t st) {
  if (n <= 0) return MANSY_OK;
  MANSY_REQUIRE(a && b && iou, "tilemap_iou: null pointer");
  MANSY_LAUNCH(tilemap_iou_kernel, dim3(mansy_ceil_div(n, 256)), dim3(256), 0, st, a, b, n, iou);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_tilemap_or_groups(const unsigned long long* maps, long long ngroups, int group, unsigned long long* out, hipStream_t st) {
  MANSY_REQUIRE(group >= 1, "tilemap_or: bad group size");
  if (ngroups <= 0) return MANSY_OK;
  MANSY_REQUIRE(maps && out, "tilemap_or: null pointer");
  MANSY_LAUNCH(tilemap_or_kernel, dim3(mansy_ceil_div(ngroups, 256)), dim3(256), 0, st, maps, ngroups, group, out);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
