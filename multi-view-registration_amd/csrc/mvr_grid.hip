// csrc/mvr_grid.hip -- exact 1-NN for queries that come with a BOUND, one thread per query (gfx950).
//
// The culled kernel of mvr_cull.hip answers 64 queries at a time against 64-point cells: ~450 distance evaluations
// per query however tight the query's bound is -- its granularity is the cell.  In an ICP run almost every search
// after the first pass comes with a bound a fraction of a millimetre wide: the forward search of a pass starts from
// the distance of its previous match (the clouds move by little between passes), the reverse search of a matched
// target from its forward distance.  For those the right tool is what the reference gets from its kd-tree
// (tree_->nearestKSearch inside icp.align, registrator.cpp:569, and determineReciprocalCorrespondences, :502/:649):
// look at the handful of points inside the ball and nothing else.
//
// Structure: a UNIFORM GRID over every point set in its canonical (upload) coordinates -- pose-invariant like the
// Hilbert ordering, built once per scan (cell ids -> radix sort -> dense cell-start array: O(1) lookup, no hashing) and
// shared by all posed copies; plus a one-byte-per-cell distance map ("how many cells to the nearest occupied one").
// Per pass a posed copy only refreshes its coordinates in grid order (one streaming launch for all views).
// Search: the query is mapped into the target's canonical frame (inverse pose), the cells overlapped by the ball of its
// bound (plus a margin far above every rounding involved) are walked row by row -- a row of cells along x is ONE
// contiguous range of the grid-ordered array -- and every candidate is evaluated in the POSED frame with the float
// formula of all the other search kernels: same d2 bits, lowest ORIGINAL index on ties, so the result is bit-identical
// to the brute-force and the culled kernels.  A query without any bound (no previous match) first reads the distance
// map: two thirds of a turntable scan has no counterpart in its neighbour and leaves after one byte.  A query whose ball
// is WIDE (no bound but close to the target, or a seed that moved far: the first passes) is not walked by one thread --
// a wave is as slow as its slowest lane and those queries come in clusters -- but flagged for the culled kernel, which
// then runs over the flagged positions in place (its blocks without a flagged query leave at once).
// Compiled with -ffp-contract=off.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <iterator>
#include <type_traits>
#include <vector>

#include "mvr_internal.h"

namespace mvr {

CellGrid::~CellGrid()
{
  if (block) (void)hipFree(block);            // (start, gperm, graw, g2h, h2g and dt are carved out of it)
  if (ready) (void)hipEventDestroy(ready);
}

namespace {

struct GridGeom { float lo[3]; float inv_h; int dim[3]; };

__device__ __forceinline__ int cell_of(float x, float lo, float inv_h, int dim)
{
  // monotone in x (float subtract, multiply and floor are monotone): a point inside [a, b] gets a cell inside
  // [cell_of(a), cell_of(b)] -- what the search's cell range relies on.  Clamped: points ON the upper face.
  const float f = floorf((x - lo) * inv_h);
  return (int)fminf(fmaxf(f, 0.f), (float)(dim - 1));
}

__global__ void cell_id_kernel(const float4 *__restrict__ p, size_t n, GridGeom g, uint32_t *__restrict__ cid, uint32_t *__restrict__ idx)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = p[i];
  const int x = cell_of(v.x, g.lo[0], g.inv_h, g.dim[0]), y = cell_of(v.y, g.lo[1], g.inv_h, g.dim[1]), z = cell_of(v.z, g.lo[2], g.inv_h, g.dim[2]);
  cid[i] = (uint32_t)((z * g.dim[1] + y) * g.dim[0] + x);
  idx[i] = (uint32_t)i;
}

// distance map: dt[c] = Chebyshev distance, in cells, from cell c to the nearest occupied cell, capped (255 = farther than
// `steps`).  The Chebyshev ball is a cube, so the minimum over the occupied cells of max(|dx|, |dy|, |dz|) separates:
//   a(c) = min over dx of |dx| with (x + dx, y, z) occupied;  b(c) = min over dy of max(|dy|, a(x, y + dy, z));
//   dt(c) = min over dz of max(|dz|, b(x, y, z + dz))
// -- three launches that each look at a window of 2 steps + 1 cells along one axis, instead of one dilation launch per
// step over all 27 neighbours (round 2: up to 12 launches of 47 us per scan, half of a grid's build time).
__global__ void dt_occupancy_kernel(const uint32_t *__restrict__ start, size_t cells, uint8_t *__restrict__ out)
{
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < cells) out[c] = start[c + 1] > start[c] ? 0 : 255;
}
__global__ void dt_axis_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int nx, int ny, int nz, int axis, int steps)
{
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= (size_t)nx * ny * nz) return;
  const int x = (int)(c % nx), y = (int)((c / nx) % ny), z = (int)(c / ((size_t)nx * ny));
  const int pos = axis == 0 ? x : axis == 1 ? y : z, len = axis == 0 ? nx : axis == 1 ? ny : nz;
  const size_t stride = axis == 0 ? 1 : axis == 1 ? (size_t)nx : (size_t)nx * ny;
  auto val = [&](size_t cc) -> int { return (int)in[cc]; };
  int best = val(c);
  for (int d = 1; d <= steps && best > d; ++d) {           // a value found at distance d cannot be beaten by anything farther than it
    if (pos - d >= 0) best = min(best, max(d, val(c - (size_t)d * stride)));
    if (pos + d < len) best = min(best, max(d, val(c + (size_t)d * stride)));
  }
  out[c] = (uint8_t)(best > steps ? 255 : best);
}

// start[c] = first grid position whose cell id is >= c; start[cells] = n.  From the SORTED cell ids: the first point of a
// run of equal ids knows its cell's start; an empty cell takes the start of the next occupied one -- a suffix minimum,
// done as an inclusive min-scan over the reversed array.
__global__ void cell_first_kernel(const uint32_t *__restrict__ sorted_cid, size_t n, size_t cells, uint32_t *__restrict__ start)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0) start[cells] = (uint32_t)n;
  if (k >= n) return;
  const uint32_t c = sorted_cid[k];
  if (k == 0 || sorted_cid[k - 1] != c) start[c] = (uint32_t)k;
}
struct MinU32 { __host__ __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a < b ? a : b; } };
// a random-access iterator over p[len - 1], p[len - 2], ..., p[0] (hipCUB's scans take any such iterator)
struct ReverseU32 {
  typedef std::random_access_iterator_tag iterator_category;
  typedef uint32_t value_type; typedef ptrdiff_t difference_type; typedef uint32_t *pointer; typedef uint32_t &reference;
  uint32_t *last;       // &p[len - 1]
  __host__ __device__ uint32_t &operator*() const { return *last; }
  __host__ __device__ uint32_t &operator[](ptrdiff_t i) const { return *(last - i); }
  __host__ __device__ ReverseU32 operator+(ptrdiff_t i) const { return ReverseU32{last - i}; }
  __host__ __device__ ReverseU32 operator-(ptrdiff_t i) const { return ReverseU32{last + i}; }
  __host__ __device__ ptrdiff_t operator-(const ReverseU32 &o) const { return o.last - last; }
  __host__ __device__ ReverseU32 &operator+=(ptrdiff_t i) { last -= i; return *this; }
  __host__ __device__ ReverseU32 &operator-=(ptrdiff_t i) { last += i; return *this; }
  __host__ __device__ ReverseU32 &operator++() { --last; return *this; }
  __host__ __device__ ReverseU32 operator++(int) { ReverseU32 t = *this; --last; return t; }
  __host__ __device__ ReverseU32 &operator--() { ++last; return *this; }
  __host__ __device__ bool operator==(const ReverseU32 &o) const { return last == o.last; }
  __host__ __device__ bool operator!=(const ReverseU32 &o) const { return last != o.last; }
  __host__ __device__ bool operator<(const ReverseU32 &o) const { return last > o.last; }
};
inline ReverseU32 thrust_like_reverse(uint32_t *p, size_t len) { return ReverseU32{p + (len ? len - 1 : 0)}; }
// the eight entries behind start[cells] (the walk reads four consecutive starts with one load, the staged walk eight with two): all n
constexpr size_t kStartPad = 8;
__global__ void pad_start_kernel(uint32_t *__restrict__ start, size_t cells, uint32_t n)
{
  if (threadIdx.x < kStartPad) start[cells + 1 + threadIdx.x] = n;
}

__global__ void grid_gather_kernel(const float4 *__restrict__ p, const uint32_t *__restrict__ gperm, size_t n, float4 *__restrict__ graw)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t o = gperm[k];
  float4 v = p[o];
  v.w = __uint_as_float(o);
  graw[k] = v;
}

// (for several grids at once, blockIdx.y = grid: a registration's first pass asked for twelve, one launch each with the gap between them)
struct G2HBatch { const uint32_t *gperm[kBatchClouds], *inv[kBatchClouds]; uint32_t *g2h[kBatchClouds], *h2g[kBatchClouds]; unsigned long long n[kBatchClouds]; };
__global__ void g2h_kernel(G2HBatch b)
{
  const int g = blockIdx.y;
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < (size_t)b.n[g]) { const uint32_t h = b.inv[g][b.gperm[g][k]]; b.g2h[g][k] = h; b.h2g[g][h] = (uint32_t)k; }
}

// posed coordinates in grid order: the SAME function of the same inputs as the posed points themselves
// (pose_point_f64 of the canonical point), so gsorted[k].xyz == pts[gperm[k]].xyz bit for bit
struct GridPoseBatch { const float4 *graw[kBatchClouds]; float4 *out[kBatchClouds]; unsigned long long n[kBatchClouds]; Mat44d T[kBatchClouds]; const Mat44d *Tp[kBatchClouds]; };
__global__ void grid_pose_kernel(GridPoseBatch b)
{
  const int cl = blockIdx.y;
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= b.n[cl]) return;
  const float4 p = b.graw[cl][k];
  float4 v = pose_point_f64(b.Tp[cl] ? *b.Tp[cl] : b.T[cl], p);
  v.w = p.w;
  b.out[cl][k] = v;
}

template <bool FMA>
__device__ __forceinline__ float gdist2(const float4 t, float qx, float qy, float qz)
{
  const float dx = t.x - qx, dy = t.y - qy, dz = t.z - qz;
  if (FMA) return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}
// The ball of a bounded query in the searched cloud's CANONICAL frame: centre r = minv (q, 1), radius rad.
// What the margin has to cover (everything else is exact): the query q and the posed target points are FLOAT roundings
// of the exact motion -- half an ulp per coordinate at the POSED magnitude (3e-5 mm at |p| ~ 1e3 mm), which the inverse
// pose lengthens by at most `stretch`; the float distance formula (relative, the 1.00001); and the float arithmetic of
// the cell range -- the rounding of r itself and of r -+ rad, half an ulp each at the CANONICAL magnitude.  The two
// frames can differ by orders of magnitude (raw scans kept in a world frame 1e5 mm from the origin, posed next to it:
// ulp(1e5) = 8e-3 mm), so each part scales with its own frame's coordinates (the second term was missing).
struct Ball { float rx, ry, rz, rad; };
__device__ __forceinline__ Ball grid_ball(const GridPair &a, const float4 q, const float bound)
{
  const double *mi = a.pose_dev ? a.pose_dev->minv : a.minv;
  const float stretch = a.pose_dev ? a.pose_dev->stretch : a.stretch;
  const double qx = q.x, qy = q.y, qz = q.z;
  Ball b;
  // (fused multiply-adds and the bare v_sqrt_f32 -- one ulp -- on purpose: nothing here is a result, the margins are tens of ulps
  // wide, and the walk is bound by the vector instructions a wave issues: twenty fewer per 64 queries)
  b.rx = (float)(__builtin_fma(mi[2], qz, __builtin_fma(mi[1], qy, mi[0] * qx)) + mi[3]);
  b.ry = (float)(__builtin_fma(mi[6], qz, __builtin_fma(mi[5], qy, mi[4] * qx)) + mi[7]);
  b.rz = (float)(__builtin_fma(mi[10], qz, __builtin_fma(mi[9], qy, mi[8] * qx)) + mi[11]);
  b.rad = (__builtin_amdgcn_sqrtf(bound) * 1.00001f + (1.0e-3f + 4.0e-6f * (fabsf(q.x) + fabsf(q.y) + fabsf(q.z)))) * stretch +
          2.0e-6f * (fabsf(b.rx) + fabsf(b.ry) + fabsf(b.rz));
  return b;
}

#ifndef MVR_GRID_THREADS
#define MVR_GRID_THREADS 128
#endif
constexpr int kGridThreads = MVR_GRID_THREADS;      // (threads per block of the walk: 128 and 256 measured equal -- 281 against 282.5 us of kernels per pass --, 512 slower -- 297; 128 it is since the two waves of a block stage ONE region together, MVR_STAGE_BLOCK; tools/ab_variant_kernels.sh)
#ifndef MVR_GRID_ROW4
#define MVR_GRID_ROW4 1
#endif
struct __attribute__((packed, aligned(4))) Start4 { uint32_t a, b, c, d; };      // four consecutive cell starts, 4-byte aligned (the array is padded by four entries)

// ---- where a cell's points begin: the dense table or the compact one (CellGrid).  `A` = GridPair or PartDesc.
constexpr uint32_t kSegOcc = 0x80000000u;
template <class A> __device__ __forceinline__ uint32_t cell_start_of(const A &a, uint32_t c)
{
  if (!a.dir) return a.start[c];
  const uint32_t d = a.dir[c >> 5];
  return (d & kSegOcc) ? a.recs[((d & ~kSegOcc) << 5) + (c & 31u)] : d;
}
// the range of the cells [c0, c1e) of one row (c1e - c0 - 1 = nxm cells more than the first): dense -- one 16-byte load from the
// first entry brings both ends while the range is up to three cells long; compact -- the directory entry of c0's segment, which is
// c1e's too in seven rows of eight, then both ends from the segment's record (one 16-byte load when they lie within four entries
// of each other and of the record's end), or the entry itself for a segment without points
template <class A> __device__ __forceinline__ void cell_range_of(const A &a, uint32_t c0, uint32_t nxm, uint32_t &s, uint32_t &e)
{
  if (!a.dir) {
    const Start4 v = *reinterpret_cast<const Start4 *>(a.start + c0);
    s = v.a;
    e = nxm == 0u ? v.b : nxm == 1u ? v.c : v.d;
    if (nxm > 2u) e = a.start[c0 + nxm + 1u];
    return;
  }
  const uint32_t c1e = c0 + nxm + 1u;
  const uint32_t d0 = a.dir[c0 >> 5];
  if ((c1e >> 5) == (c0 >> 5)) {
    if (!(d0 & kSegOcc)) { s = d0; e = d0; return; }
    const uint32_t *r = a.recs + ((d0 & ~kSegOcc) << 5);
    if (nxm <= 2u && (c0 & 31u) <= 28u) {
      const Start4 v = *reinterpret_cast<const Start4 *>(r + (c0 & 31u));
      s = v.a; e = nxm == 0u ? v.b : nxm == 1u ? v.c : v.d;
    } else { s = r[c0 & 31u]; e = r[c1e & 31u]; }
    return;
  }
  s = (d0 & kSegOcc) ? a.recs[((d0 & ~kSegOcc) << 5) + (c0 & 31u)] : d0;
  e = cell_start_of(a, c1e);
}
// the distance map: per cell, or per cube of 2^dt_shift cells per axis; what its byte says about the nearest point, in mm
template <class A> __device__ __forceinline__ uint32_t dt_of(const A &a, int cx, int cy, int cz)
{
  if (a.dt_shift == 0) return a.dt[((size_t)cz * a.dim[1] + cy) * a.dim[0] + cx];
  return a.dt[((size_t)(cz >> a.dt_shift) * a.dtdim[1] + (cy >> a.dt_shift)) * a.dtdim[0] + (cx >> a.dt_shift)];
}
template <class A> __device__ __forceinline__ float dt_least_of(const A &a, uint32_t d)
{
  // the query's (clamped) cell -- or cube -- is d cells / cubes from the nearest occupied one: every point is at least (d - 1)
  // cell / cube edges away along some axis; 255 = farther than the map was built for
  return (d == 255u ? (float)a.dt_max : (float)d - 1.f) * a.h * (float)(1 << a.dt_shift);
}

// ---- wave-wide reductions and scans without the LDS crossbar: DPP row shifts inside the rows of 16 lanes, the four row results
// read as scalars (a __shfl_* is a ds_bpermute: it takes the LDS's issue slots, which the staged walk below needs for its points)
template <int CTRL> __device__ __forceinline__ int dpp_keep(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }      // lanes without a source keep v
template <int CTRL> __device__ __forceinline__ int dpp_zero(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }       // ... get 0
__device__ __forceinline__ int wave_min_i32(int v)
{
  v = min(v, dpp_keep<0x111>(v)); v = min(v, dpp_keep<0x112>(v)); v = min(v, dpp_keep<0x114>(v)); v = min(v, dpp_keep<0x118>(v));      // lane 15 of a row: the row's minimum
  return min(min(__builtin_amdgcn_readlane(v, 15), __builtin_amdgcn_readlane(v, 31)), min(__builtin_amdgcn_readlane(v, 47), __builtin_amdgcn_readlane(v, 63)));
}
__device__ __forceinline__ int wave_max_i32(int v)
{
  v = max(v, dpp_keep<0x111>(v)); v = max(v, dpp_keep<0x112>(v)); v = max(v, dpp_keep<0x114>(v)); v = max(v, dpp_keep<0x118>(v));
  return max(max(__builtin_amdgcn_readlane(v, 15), __builtin_amdgcn_readlane(v, 31)), max(__builtin_amdgcn_readlane(v, 47), __builtin_amdgcn_readlane(v, 63)));
}
// inclusive prefix sum over the 64 lanes (Hillis-Steele inside the rows, then the row totals carried across: three scalar adds)
__device__ __forceinline__ uint32_t wave_scan_incl_u32(uint32_t x, uint32_t *total)
{
  int v = (int)x;
  v += dpp_zero<0x111>(v); v += dpp_zero<0x112>(v); v += dpp_zero<0x114>(v); v += dpp_zero<0x118>(v);
  const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47), r3 = __builtin_amdgcn_readlane(v, 63);
  const int row = (int)(threadIdx.x & 63) >> 4;
  v += row == 0 ? 0 : row == 1 ? r0 : row == 2 ? r0 + r1 : r0 + r1 + r2;
  *total = (uint32_t)(r0 + r1 + r2 + r3);
  return (uint32_t)v;
}

// ---- the STAGED walk (round 4).  What bounds the plain walk below is not bytes and not cache lines but the NUMBER of vector
// memory instructions a wave issues: every 16-byte-per-lane load costs the CU's texture-address path ~19 cycles whatever its
// lanes touch (tools/exp_ta.hip; TA_TA_BUSY / SQ_INSTS_VMEM_RD = 19.6 cycles in the walk), and a wave of the settled ring step
// issues ~90 of them: a range load per row of cells and four point loads per round, for as many rounds as its LONGEST lane needs.
// The 64 queries of a wave are neighbours on a surface (Hilbert order), so the cells their balls overlap lie in one small box.
// With STAGE the wave
//   0. lets every lane ask for the ranges of its first rows of cells at once (they depend on its own ball only);
//   1. reduces the walking lanes' cell boxes to their hull [X0, X1] x [Y0, Y1] x [Z0, Z1]  (six DPP reductions);
//   2. marks, per row of that box, the cells some lane wants (an LDS OR per lane and row: the hull is mostly air -- a surface runs
//      through it at an angle) and reads where each row's wanted cells begin and end: two 4-byte loads per lane for the whole wave;
//   3. prefix-sums the rows' lengths and copies their points -- a row's wanted cells are one contiguous range of the grid-ordered
//      array -- into LDS by LDS-DMA loads (global_load_lds_dwordx4: per-lane source, 64 consecutive staged positions per
//      instruction, no registers, no ds_write), with a table row -> {grid position, staged position} of its first point;
//   4. every lane then walks ONLY ITS OWN cells, row by row as before, but from LDS: a ds_read_b64 turns a row's range into staged
//      positions, four ds_read_b128 per round of four points -- the LDS serves a wave's 16-byte read in 4 cycles, not 19.
// Rows are staged in order while they fit (kStagePts points, kStageRows rows): a row behind that is walked from global memory by
// the lanes that want it -- the staging degrades row by row, there is no cliff.  Same candidates (a lane's own cells, or real
// target points behind a short row as before), same `take`, same 64-bit candidate, same tie rule: bit-identical to the plain
// walk (tests/test_gpu_ring.py, test_gpu_exact.py with grid_stage 0 / 1 / 2).  The probe and the wide / listed routes are
// untouched.  registrator.cpp:644-649 is what this answers.
// What it buys, what it does not (profiles/r04_*): the forward launch of the 12 x 200k ring 113 -> 96-100 us (vector memory
// instructions 3.3e6 -> 1.4e6 per launch, L1 accesses 4.8e7 -> 2.4e7, TA busy 93 % -> 63 % of the CU-busy cycles), at 45 % MORE
// vector ALU instructions (the staging: 4.9e7 -> 7.1e7, the VALUs now 57 % busy) and 16.6 KB of LDS per block (eight waves per SIMD
// still: 62 registers).  No unit is saturated any more; what is left is a wave's chain of ~40 dependent steps times what the CU
// can keep resident.  Built on top, measured, NOT kept: the rounds of all lanes dealt evenly through a list in LDS
// (the dealt rounds took 4 k cycles per wave instead of 12 k -- making the list cost 13 k, slot allocation by LDS atomics or by a
// prefix sum alike); more points per wave at fewer waves per SIMD (256 / 320 / 448 points: 98 / 122 / 141 us); a per-cell table
// in LDS instead of the lanes' own range loads (a box of up to 7 cells along x: 20 % of the waves are wider); the probe as part
// of the staged step (97 -> 113 us); start bounds without the seed's two gathers (seed_delta: exact, equal).  The REVERSE
// launch's queries are 2.4 x sparser (the matched targets): its waves want ~350 points, 14 % of them fit, and it stays with the
// plain walk (70 us against 72 staged).  Memory-side traffic is unchanged (286 MB per forward launch): a wave's lines were
// shared through the L1 / L2 before, they are fetched once per wave now -- the misses are the same lines.
#ifndef MVR_STAGE_PTS
#define MVR_STAGE_PTS 224
#endif
#ifndef MVR_STAGE_ROWS
#define MVR_STAGE_ROWS 64
#endif
#ifndef MVR_STAGE_BURST
#define MVR_STAGE_BURST 4
#endif
#ifndef MVR_STAGE_PROBE
#define MVR_STAGE_PROBE 0
#endif
#ifndef MVR_STAGE_BLOCK
#define MVR_STAGE_BLOCK (MVR_GRID_THREADS == 128)
#endif
constexpr int kStageRows = MVR_STAGE_ROWS;     // rows of cells of a wave's box: one or two per lane
constexpr int kStagePts = MVR_STAGE_PTS;       // points a wave stages at most (16 bytes each; + the table: ~7.5 KB of LDS per wave at 384)

// G lanes share a query: lane `sub` of the group walks rows sub, sub + G, ... of the ball's rows of cells, the group's
// answers meet in log2(G) shuffles.  (One lane per query leaves a wave waiting for its widest ball -- rows times points
// of dependent round trips; dealing the rows to G lanes cuts that chain G-fold and the spread between lanes with it.
// The query, its seed and its cell range are loaded by all G lanes from the same addresses: one request.)
// (at least seven waves per SIMD: what the staged walk's LDS admits -- it sits at 74 registers without the hint, one too many; the plain
// instantiations need fewer than 64 and keep their eighth wave)
#ifndef MVR_STAGE_WAVES
#define MVR_STAGE_WAVES 7
#endif
#if MVR_STAGE_WAVES
#define MVR_GRID_OCC __attribute__((amdgpu_waves_per_eu(MVR_STAGE_WAVES, 8)))
#else
#define MVR_GRID_OCC
#endif
template <bool FMA, int G, bool STAGE>
__global__ void __launch_bounds__(kGridThreads) MVR_GRID_OCC nn_grid_kernel(GridBatch batch, XcdMap map, unsigned long long *__restrict__ evals, unsigned long long *__restrict__ stage_stat)
{
  static_assert(!STAGE || G == 1, "the staged walk is the one-lane-per-query walk");
  constexpr uint32_t kPerBlock = kGridThreads / G;
  uint32_t pair = 0, set = 0;
  if (!xcd_map_block(map, blockIdx.x, &pair, &set)) return;
  const GridPair &a = batch.p[pair];
  const float cap2 = batch.cap2;
  const uint32_t nq = a.qlist ? min(*a.qcount, a.q_count) : a.q_count;
  const uint32_t sub = threadIdx.x % G;
  const uint32_t pos = set * kPerBlock + threadIdx.x / G;
  unsigned long long n_eval = 0;
  bool went_wide = false, went_cull = false;
#ifdef MVR_STAGE_CLOCK
  // diagnostics build: shader-clock stamps between the phases of a wave, summed per outcome into stage_stat[16 + 8 * outcome + phase]
  // (phases: 0 prologue, 1 box + rows + table, 2 point staging, 3 walk, 4 epilogue; [.. + 7] = waves)
  long long ck[6]; int ckn = 0;
#define MVR_CK() do { ck[ckn++] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define MVR_CK() do { } while (0)
#endif
  MVR_CK();
  // (per-lane state that outlives the prologue: the staged walk is a WAVE-wide step, so it cannot sit inside `if (pos < nq)`)
  const bool live = pos < nq;
  bool walks = false, answers = false;
  uint32_t qpos = 0;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  int x0 = 0, x1 = -1, y0 = 0, y1 = -1, z0 = 0, z1 = -1;
  unsigned long long best = ~0ull;
  const uint32_t nt4 = a.nt - 4u;     // (a grid is only built for a set of four points or more)
  // The candidate so far as ONE 64-bit number, (bits of d2, original index): d2 >= +0, so the order of the bits is the order of
  // the distances, and "nearer, or as near with the smaller index" is one unsigned comparison.  It starts at (bound, none):
  // candidates beyond the bound cannot be the answer; at the bound they can (inclusive).
  // (WHERE in the grid order the winner lies is not kept: its Hilbert position, which the keys and the marks want, comes
  // from its original index through the ordering's inverse -- one gather per query either way, three instructions less
  // per candidate round and four registers: 62 instead of 66, the eighth wave per SIMD)
  auto take = [&](const float4 t) {
    const unsigned long long cand = ((unsigned long long)__float_as_uint(gdist2<FMA>(t, q.x, q.y, q.z)) << 32) | __float_as_uint(t.w);
    best = cand < best ? cand : best;
  };
  // the walk of a box of cells [x0, x1] x [y0, y1] x [z0, z1]: rows of cells (x-runs: ONE contiguous
  // range of the grid-ordered array each); the next row's range is requested while this row's points are evaluated.
  // (a row's range = two entries of the cell-start array at most three cells apart in all but the widest balls:
  // ONE 16-byte load from the first of them brings both)
  auto walk_box = [&](const int wx0, const int wx1, const int wy0, const int wy1, const int wz0, const int wz1) {
    const uint32_t nxm = (uint32_t)(wx1 - wx0);
    auto row_range = [&](uint32_t row, uint32_t &s, uint32_t &e) { cell_range_of(a, row + (uint32_t)wx0, nxm, s, e); };
    const int ny = wy1 - wy0 + 1, nrows = ny * (wz1 - wz0 + 1);
    uint32_t s, e;
    uint32_t row = (uint32_t)((wz0 * a.dim[1] + wy0) * a.dim[0]);          // rows in y-major order: the next one is dim[0] further, or at the next z
    const uint32_t row_step = (uint32_t)a.dim[0], z_step = (uint32_t)((a.dim[1] - ny) * a.dim[0]);
    int yy = 0;
    row_range(row, s, e);
    for (int it = 0; it < nrows; ++it) {
      uint32_t s2 = 0, e2 = 0;
      if (it + 1 < nrows) {
        row += row_step;
        if (++yy == ny) { yy = 0; row += z_step; }
        row_range(row, s2, e2);
      }
      if (sub == 0) n_eval += e - s;
      if (G == 1) {
        // one lane per query: four CONSECUTIVE points per round from one address (the loads differ by their immediate offsets).
        // A short row runs on into the points behind it -- points of the target all the same: looking at one more changes
        // nothing (whatever lies within the bound lies in the ball's cells and is looked at anyway); only the array's end is
        // kept clear of (the last round starts four points before it at the latest).
        for (uint32_t k = s; k < e; k += 4) {
          const uint32_t kb = min(k, nt4);
          const float4 *__restrict__ p4 = a.gts + kb;
          const float4 t0 = p4[0], t1 = p4[1], t2 = p4[2], t3 = p4[3];
          take(t0); take(t1); take(t2); take(t3);
        }
      } else {
        // G lanes per query: they take CONSECUTIVE points of the row -- one instruction, one cache line per query
        // (what a gather costs in the L1 is the number of distinct lines its lanes touch, not the bytes per lane:
        // tools/exp_ta.hip), two rounds in flight
        for (uint32_t k = s + sub; k < e; k += 2 * G) {
          const uint32_t k1 = k + G < e ? k + G : k;
          const float4 t0 = a.gts[k], t1 = a.gts[k1];
          take(t0); take(t1);
        }
      }
      s = s2; e = e2;
    }
  };
  // (prologue state that outlives it: with STAGE the probe of a wide ball is itself a staged walk, so the prologue is cut in two)
  float rx = 0.f, ry = 0.f, rz = 0.f;
  bool worth = false, seeded = false, wants_probe = false, wide = false;
  int px0 = 0, px1 = -1, py0 = 0, py1 = -1, pz0 = 0, pz1 = -1;      // the probe's 2 x 2 x 2 cells
  if (live) {
    qpos = a.qlist ? a.qlist[pos] : a.q_begin + pos;            // position in the query cloud's Hilbert order
    // (the three loads that depend on nothing but the position go out together: the prologue is a chain of dependent gathers,
    // query -> previous match -> its grid position -> its coordinates, and every link not waited for separately is a round trip less)
    q = a.qs[qpos];
    const uint32_t start_bits = a.qbound ? a.qbound[qpos] : 0xFFFFFFFFu;
    const unsigned long long pkey = a.seed_from_keys ? a.keys[qpos] : kKeyInit;
    const uint32_t prev = (uint32_t)pkey;
    // the bound: the cap, or a distance within which a point is KNOWN to exist (inclusive)
    float bound = cap2;
    if (start_bits <= __float_as_uint(bound)) { bound = __uint_as_float(start_bits); seeded = true; }
    const float pdelta = (a.seed_from_keys || a.cert) ? (a.pose_dev ? a.pose_dev->delta + (a.qpose_dev ? a.qpose_dev->delta : 1.0e30f) : a.delta) : -1.f;
    if (prev < a.nt) {
      // the query's match of the pass before is still a point of the target: its distance NOW bounds the nearest point's.
      // seed_delta: while the two clouds have moved by little since (delta: an upper bound from the poses, through the device
      // records or by value) that distance is at most the old one + delta -- no need to gather the point's coordinates (two
      // dependent gathers at the head of every wave).  What the slack covers: the float roundings of the four posed points
      // involved (half an ulp per coordinate at their magnitude: the 4e-6 |q|_1 term, as in grid_ball), of the two distance
      // formulas and of the square root (the 1.000001 factors); the old and the new distance are compared as real numbers.
      const float delta = pdelta;
      if (delta >= 0.f && delta <= batch.delta_max) {
        const float b = __builtin_amdgcn_sqrtf(__uint_as_float((uint32_t)(pkey >> 32))) * 1.000001f + delta + (1.0e-3f + 4.0e-6f * (fabsf(q.x) + fabsf(q.y) + fabsf(q.z)));
        const float b2 = b * b * 1.000001f;
        if (b2 <= bound) { bound = b2; seeded = true; }
      } else {
        const float d = gdist2<FMA>(a.gts[a.h2g[prev]], q.x, q.y, q.z);      // (the same point, read from the array the walk is about to read)
        if (d <= bound) { bound = d; seeded = true; }
      }
    }
    // the ball in the target's canonical frame (the mapping itself is done in double)
    const Ball ball = grid_ball(a, q, bound);
    rx = ball.rx; ry = ball.ry; rz = ball.rz;
    const float rad = ball.rad;
    x0 = cell_of(rx - rad, a.lo[0], a.inv_h, a.dim[0]); x1 = cell_of(rx + rad, a.lo[0], a.inv_h, a.dim[0]);
    y0 = cell_of(ry - rad, a.lo[1], a.inv_h, a.dim[1]); y1 = cell_of(ry + rad, a.lo[1], a.inv_h, a.dim[1]);
    z0 = cell_of(rz - rad, a.lo[2], a.inv_h, a.dim[2]); z1 = cell_of(rz + rad, a.lo[2], a.inv_h, a.dim[2]);
    worth = true;
    if (!seeded) {
      // nothing known: is any target point near at all?  The query's own cell (clamped into the grid) is dt cells from the
      // nearest occupied one, so every target point is at least (dt - 1) cell edges away along some axis -- plus
      // however far the query itself lies outside the grid
      const int cx = cell_of(rx, a.lo[0], a.inv_h, a.dim[0]), cy = cell_of(ry, a.lo[1], a.inv_h, a.dim[1]), cz = cell_of(rz, a.lo[2], a.inv_h, a.dim[2]);
      worth = !(dt_least_of(a, dt_of(a, cx, cy, cz)) > rad);
    }
    if (a.cert) {
      // a rim certificate: an earlier pass proved that NO target point lies within the cap + a margin of this query, and the two
      // clouds have moved by less than what is left of that margin since: there is still none within the cap -- no distance map,
      // no probe, no listed set.  (What is left must stay above what the float roundings of the posed points can amount to.)
      const float cm = a.cert[qpos];
      if (cm > 0.f) {
        float rem = (!seeded && pdelta >= 0.f) ? cm - pdelta : 0.f;
        if (rem > 2.0e-3f + 4.0e-6f * (fabsf(q.x) + fabsf(q.y) + fabsf(q.z))) worth = false; else rem = 0.f;
        a.cert[qpos] = rem;
      }
    }
    best = ((unsigned long long)__float_as_uint(bound) << 32) | kNone;
    const int nrows = (y1 - y0 + 1) * (z1 - z0 + 1);
    wide = worth && nrows > batch.light_rows;
    wants_probe = G == 1 && worth && batch.probe && nrows > batch.probe_rows;
    if (wants_probe) {
      // A wide ball is a LOOSE bound more often than a far neighbour: the seed of a pass after a large motion (the first passes
      // of a registration, a restart) is the old match, a millimetre or two off, while the nearest point is where it always
      // is -- in the cell next to the query.  So before the query leaves for the slow lanes: the 2 x 2 x 2 cells nearest to it
      // (its own and, per axis, the neighbour on the side it leans to -- four short rows).  Whatever that finds is a point of
      // the target: its distance is a valid (inclusive) bound, and the ball of THAT bound is what is walked or handed on.
      const float fx = (rx - a.lo[0]) * a.inv_h, fy = (ry - a.lo[1]) * a.inv_h, fz = (rz - a.lo[2]) * a.inv_h;
      const int cx = cell_of(rx, a.lo[0], a.inv_h, a.dim[0]), cy = cell_of(ry, a.lo[1], a.inv_h, a.dim[1]), cz = cell_of(rz, a.lo[2], a.inv_h, a.dim[2]);
      px0 = max(cx - (fx - (float)cx < 0.5f ? 1 : 0), 0); px1 = min(px0 + 1, a.dim[0] - 1);
      py0 = max(cy - (fy - (float)cy < 0.5f ? 1 : 0), 0); py1 = min(py0 + 1, a.dim[1] - 1);
      pz0 = max(cz - (fz - (float)cz < 0.5f ? 1 : 0), 0); pz1 = min(pz0 + 1, a.dim[2] - 1);
    }
  }
  // the second half of the prologue, once the probe (if any) has been walked: its find as the new bound, and where the query goes
  auto route = [&]() {
    if (wants_probe && (uint32_t)best != kNone) {              // (its distance <= the old bound: the walk only takes candidates within it)
      const Ball ball = grid_ball(a, q, __uint_as_float((uint32_t)(best >> 32)));
      const float rad = ball.rad;
      x0 = cell_of(rx - rad, a.lo[0], a.inv_h, a.dim[0]); x1 = cell_of(rx + rad, a.lo[0], a.inv_h, a.dim[0]);
      y0 = cell_of(ry - rad, a.lo[1], a.inv_h, a.dim[1]); y1 = cell_of(ry + rad, a.lo[1], a.inv_h, a.dim[1]);
      z0 = cell_of(rz - rad, a.lo[2], a.inv_h, a.dim[2]); z1 = cell_of(rz + rad, a.lo[2], a.inv_h, a.dim[2]);
      wide = (y1 - y0 + 1) * (z1 - z0 + 1) > batch.light_rows;       // (still wide: it goes where it would have gone without the probe)
      // The ball of the new bound inside the cells the probe has just looked at: every point within the bound HAS been
      // looked at, the candidate is the answer (ties and all: the same comparison chose it) -- no second walk.  A ball of
      // half a cell edge or less always is, which is what the nearest point of a query in the overlap leaves.
      if (x0 >= px0 && x1 <= px1 && y0 >= py0 && y1 <= py1 && z0 >= pz0 && z1 <= pz1) worth = false;
    }
    // wide balls are not walked here: without a bound they come in clusters (the rim of the overlap) and go to the culled
    // kernel, which answers 64 neighbouring queries at once; with a bound they are scattered and get a wave each
    bool to_cull = wide && a.heavy != nullptr && (!seeded || a.wide_list == nullptr);
    if (G == 1 && a.heavy != nullptr && a.wide_list != nullptr) {
      // ... unless the wave is full of them (a stretch of rim whose matches lie far: the same 64 queries are one set of the culled kernel)
      if (__popcll(__ballot(wide)) >= batch.cluster) to_cull = wide;
    }
    const bool to_wave = wide && !to_cull && a.wide_list != nullptr;
    if (a.heavy && sub == 0) a.heavy[a.qlist ? pos : qpos] = to_cull ? 1 : 0;
    went_wide = to_wave && sub == 0;
    went_cull = to_cull;
    // (a wide query that goes on with a bound from the probe: the launch that answers it starts from the start bound or the
    // seed again -- it finds the probe's point itself; what the probe saves there is nothing, what it costs is four short rows)
    answers = !to_cull && !to_wave;
    walks = answers && worth;
  };
  // (the probe as part of the STAGED step -- its four rows from LDS too -- was built and measured: the forward launch of a settled
  // pass 97 -> 113 us, the pass that restarts from the prior 0.58 -> 0.63 ms: a wave with one probing lane stages the hull of the
  // probe boxes AND walks the refined balls the plain way afterwards, at 72 registers instead of 65.  -DMVR_STAGE_PROBE=1 builds it.)
  constexpr bool kStageProbe = STAGE && MVR_STAGE_PROBE;
  if constexpr (!kStageProbe) {
    if (live) {
      if (wants_probe) walk_box(px0, px1, py0, py1, pz0, pz1);
      route();
    }
  }
  bool staged = false;
  MVR_CK();
  if constexpr (STAGE) {
    static_assert(kStageRows == 64, "one row of the box per lane");
    // (MVR_STAGE_BLOCK, the default with blocks of 128 threads: the two waves of a block stage ONE region for their 128 queries -- the
    // hull of both, a row of it per THREAD, twice the points -- instead of one each.  Settled passes equal (the hull of 128 Hilbert
    // neighbours holds 1.4 x the points of 64, the waves wait for each other at five barriers); the passes after a large motion, whose
    // balls are wide, fit the shared 384 points where they overflowed 192: pass 1 of a window from the prior 0.58 -> 0.54 ms, its first
    // ten passes 0.372 -> 0.364 ms.  -DMVR_GRID_THREADS=256 -DMVR_STAGE_BLOCK=0 builds the per-wave form.)
    constexpr bool kBlk = MVR_STAGE_BLOCK != 0;
    static_assert(!kBlk || kGridThreads == 128 || kGridThreads == 256, "block-wide staging: two or four waves per block");
    constexpr int kWaves = kGridThreads / 64, kSets = kBlk ? 1 : kWaves;
    constexpr int kRowsT = kBlk ? kWaves * kStageRows : kStageRows, kPtsT = kBlk ? kWaves * kStagePts : kStagePts, kClear = kBlk ? 4 * kGridThreads : 256;
    typedef typename std::conditional<(kRowsT > 255), uint16_t, uint8_t>::type mark_t;      // (row + 1 per staged position: 256 rows need nine bits)
    constexpr int kMarkBytesRaw = (kPtsT + 3 + 63) / 64 * 64 * (int)sizeof(mark_t);
    constexpr int kMarkBytes = kMarkBytesRaw < kClear ? kClear : (kMarkBytesRaw + kClear - 1) / kClear * kClear;      // (a multiple of what the threads clear with one store each)
    __shared__ float4 s_pts[kSets][kPtsT + 4];
    __shared__ uint2 s_rows[kSets][kRowsT];          // per row of the box: {grid position of its first staged point, staged position of that point} ({~0, ~0}: not staged)
    __shared__ uint32_t s_mask[kSets][kRowsT];       // per row of the box: the cells (bit c = cell X0 + c) some lane's ball overlaps
    __shared__ uint32_t s_mark[kSets][kMarkBytes / 4];   // per staged position (a byte each): row + 1 where a row's points begin, else 0
    __shared__ int s_red[kBlk ? kWaves : 1][8];       // (block-wide staging: the waves' partial results meet here)
    __shared__ uint32_t s_ends[kBlk ? kRowsT : 1], s_cmax[kBlk ? (kPtsT + 3 + 63) / 64 + 1 : 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sw = kBlk ? 0 : wv, tid = kBlk ? (int)threadIdx.x : lane;
    float4 *const pts = s_pts[sw];
    uint2 *const rowtab = s_rows[sw];
    uint32_t *const rmask = s_mask[sw];
    mark_t *const mark = reinterpret_cast<mark_t *>(s_mark[sw]);
    auto sync = [&]() {
      if constexpr (kBlk) __syncthreads();
      else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    };
    // what a lane walks in the staged step: the box of its ball -- or, for a ball that is to be probed first, the probe's
    // 2 x 2 x 2 cells (the probe IS a walk: in the passes after a large motion nearly every lane probes, and four rows from
    // global memory per lane was what those passes' launches were made of)
    const bool pb = kStageProbe && wants_probe;
    const bool swalk = kStageProbe ? (live && worth && (wants_probe || !wide)) : walks;
    const int bx0 = pb ? px0 : x0, bx1 = pb ? px1 : x1, by0 = pb ? py0 : y0, by1 = pb ? py1 : y1, bz0 = pb ? pz0 : z0, bz1 = pb ? pz1 : z1;
    // ---- 0. every walking lane asks for the ranges of its first rows of cells NOW (they depend on its own ball only): the
    // answers travel while the wave stages.  Rows in y-major order, as the plain walk takes them.
    const int ny = by1 - by0 + 1, nrows = ny * (bz1 - bz0 + 1);
    const uint32_t nxm = (uint32_t)(bx1 - bx0);
    auto row_range = [&](uint32_t row, uint32_t &rs_, uint32_t &re_) { cell_range_of(a, row + (uint32_t)bx0, nxm, rs_, re_); };
    uint32_t row_g = (uint32_t)((bz0 * a.dim[1] + by0) * a.dim[0]);
    const uint32_t row_step = (uint32_t)a.dim[0], z_step_g = (uint32_t)((a.dim[1] - ny) * a.dim[0]);
    int yy = 0;
    constexpr int kBurst = MVR_STAGE_BURST;
    uint32_t bs[kBurst], be[kBurst];
#pragma unroll
    for (int u = 0; u < kBurst; ++u) { bs[u] = 0; be[u] = 0; }
    if (swalk) {
#pragma unroll
      for (int u = 0; u < kBurst; ++u)
        if (u < nrows) {
          row_range(row_g, bs[u], be[u]);
          row_g += row_step;
          if (++yy == ny) { yy = 0; row_g += z_step_g; }
        }
    }
    // ---- 1. the box of the walking lanes' cells
    int X0 = wave_min_i32(swalk ? bx0 : 0x7FFFFFFF), X1 = wave_max_i32(swalk ? bx1 : -1);
    int Y0 = 0, Y1 = -1, Z0 = 0, Z1 = -1;
    if constexpr (kBlk) {
      Y0 = wave_min_i32(swalk ? by0 : 0x7FFFFFFF); Y1 = wave_max_i32(swalk ? by1 : -1);
      Z0 = wave_min_i32(swalk ? bz0 : 0x7FFFFFFF); Z1 = wave_max_i32(swalk ? bz1 : -1);
      if (lane == 0) { s_red[wv][0] = X0; s_red[wv][1] = X1; s_red[wv][2] = Y0; s_red[wv][3] = Y1; s_red[wv][4] = Z0; s_red[wv][5] = Z1; }
      __syncthreads();
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        if (w == 0) { X0 = s_red[0][0]; X1 = s_red[0][1]; Y0 = s_red[0][2]; Y1 = s_red[0][3]; Z0 = s_red[0][4]; Z1 = s_red[0][5]; }
        else { X0 = min(X0, s_red[w][0]); X1 = max(X1, s_red[w][1]); Y0 = min(Y0, s_red[w][2]); Y1 = max(Y1, s_red[w][3]); Z0 = min(Z0, s_red[w][4]); Z1 = max(Z1, s_red[w][5]); }
      }
    }
    uint32_t why = 0;      // diagnostics: 1 staged in full, 2 more rows than the table holds (the rows behind it from global memory), 3 too wide, 4 more points than fit (the rows behind from global memory)
    if (X1 >= X0) {
      if constexpr (!kBlk) {
        Y0 = wave_min_i32(swalk ? by0 : 0x7FFFFFFF); Y1 = wave_max_i32(swalk ? by1 : -1);
        Z0 = wave_min_i32(swalk ? bz0 : 0x7FFFFFFF); Z1 = wave_max_i32(swalk ? bz1 : -1);
      }
      const int NY = Y1 - Y0 + 1, R = NY * (Z1 - Z0 + 1), W = X1 - X0 + 1;
      if (W > 32) why = 3u;
      else {
        const int Rs = min(R, kRowsT);          // rows of the box that get a table entry: the first Rs in y-major order
        // ---- 2a. which cells of which rows are wanted at all: every lane ORs the x-range of its ball into the masks of its rows
        // (the box is the hull of the balls -- a surface runs through it at an angle, most of its cells are nobody's)
        rmask[tid] = 0u;
        reinterpret_cast<uint32_t *>(mark)[tid] = 0u;
        if (kMarkBytes > kClear) { for (int i = kClear / 4 + tid; i < kMarkBytes / 4; i += kClear / 4) reinterpret_cast<uint32_t *>(mark)[i] = 0u; }      // (kClear / 4 = the threads that clear: 64 per wave form, all of the block)
        sync();
        const uint32_t z_step_r = (uint32_t)(NY - ny);
        if (swalk) {
          const uint32_t bits = ((nxm >= 31u ? 0xFFFFFFFFu : ((2u << nxm) - 1u))) << (uint32_t)(bx0 - X0);
          uint32_t ridx = (uint32_t)((bz0 - Z0) * NY + (by0 - Y0));
          int ryi = 0;
          for (int k = 0; k < nrows; ++k) {
            if (ridx < (uint32_t)kRowsT) __hip_atomic_fetch_or(&rmask[ridx], bits, __ATOMIC_RELAXED, kBlk ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_WAVEFRONT);
            ridx += 1u;
            if (++ryi == ny) { ryi = 0; ridx += z_step_r; }
          }
        }
        sync();
        // ---- 2b. the rows of the box, a lane each (row `lane` = (Y0 + lane % NY, Z0 + lane / NY)): where its wanted cells
        // [first, last] begin and end in the grid-ordered array
        uint32_t sA = 0, len = 0, P = 0;
        const uint32_t m = tid < Rs ? rmask[tid] : 0u;
        if (m) {
          const int zq = (int)(((float)tid + 0.5f) * (1.0f / (float)NY)), yq = tid - zq * NY;      // (tid / NY: the float quotient is off by 1e-5 at most, the true one sits 1 / (2 NY) from an integer)
          const uint32_t base = (uint32_t)(((Z0 + zq) * a.dim[1] + (Y0 + yq)) * a.dim[0] + X0);
          const uint32_t first = (uint32_t)__builtin_ctz(m), last = 31u - (uint32_t)__builtin_clz(m);
          sA = cell_start_of(a, base + first);
          len = cell_start_of(a, base + last + 1u) - sA;
        }
        uint32_t off = wave_scan_incl_u32(len, &P) - len;
        if constexpr (kBlk) {      // (the second wave's rows begin behind the first wave's)
          if (lane == 0) s_red[wv][6] = (int)P;
          __syncthreads();
          P = 0u;
#pragma unroll
          for (int w = 0; w < kWaves; ++w) { if (w < wv) off += (uint32_t)s_red[w][6]; P += (uint32_t)s_red[w][6]; }
        }
        // rows are staged in order while they fit: a row that does not fit any more (and every row behind it) is walked from
        // global memory by the lanes that want it -- the staging degrades row by row instead of failing for the wave
        staged = true;
        const bool fits = off + len <= (uint32_t)kPtsT;
        why = P > (uint32_t)kPtsT ? 4u : R > kRowsT ? 2u : 1u;
        rowtab[tid] = fits ? make_uint2(sA, off) : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        if (fits && len) mark[off] = (mark_t)(tid + 1);
        uint32_t Ps = P;
        if (P > (uint32_t)kPtsT) {      // (what is staged: the rows before the first that does not fit -- `fits` is monotone)
          if constexpr (kBlk) {
            s_ends[tid] = off + len;
            const int nf = __popcll(__ballot(fits));
            if (lane == 0) s_red[wv][7] = nf;
            __syncthreads();
            int nfit = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) nfit += s_red[w][7];
            Ps = nfit ? s_ends[nfit - 1] : 0u;
          } else {
            const int nfit = __popcll(__ballot(fits));
            Ps = nfit ? (uint32_t)__builtin_amdgcn_readlane((int)(off + len), nfit - 1) : 0u;
          }
        }
        sync();
        MVR_CK();
        // ---- 3. the points: staged position j <- the row whose range holds it: the last row that begins at or before j -- a
        // running maximum over the positions' marks (row + 1 where a row begins), 64 positions at a time; three more positions
        // repeat the last point, so that a round of four never reads what was not staged.  64 consecutive positions per
        // instruction, and the instruction is an LDS-DMA load (global_load_lds_dwordx4: per-lane source address, destination
        // = a wave-uniform LDS address + 16 x lane): no registers for the data, no ds_write, ALL of a wave's point loads in
        // flight at once.
        if constexpr (kBlk) {
          // (chunks of 64 staged positions alternate between the two waves; the running maximum of the marks is taken inside a
          // chunk first, the chunks' maxima meet in LDS, then every chunk adds what lies before it)
          constexpr int kChunks = (kPtsT + 3 + 63) / 64, kOwn = (kChunks + kWaves - 1) / kWaves;
          const uint32_t Pm1 = Ps ? Ps - 1u : 0u, n_stage = Ps ? Ps + 3u : 0u;
          int vv[kOwn];
#pragma unroll
          for (int k = 0; k < kOwn; ++k) {
            const int c = kWaves * k + wv;
            vv[k] = 0;
            if (c < kChunks && (uint32_t)(c * 64) < n_stage) {
              const uint32_t j = (uint32_t)(c * 64 + lane);
              int v = j <= Pm1 ? (int)mark[j] : 0;
              v = max(v, dpp_keep<0x111>(v)); v = max(v, dpp_keep<0x112>(v)); v = max(v, dpp_keep<0x114>(v)); v = max(v, dpp_keep<0x118>(v));
              const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47), r3 = __builtin_amdgcn_readlane(v, 63);
              const int q4 = lane >> 4;
              const int before = q4 == 0 ? 0 : q4 == 1 ? r0 : q4 == 2 ? max(r0, r1) : max(max(r0, r1), r2);
              vv[k] = max(v, before);
              if (lane == 0) s_cmax[c] = (uint32_t)max(max(r0, r1), max(r2, r3));
            }
          }
          __syncthreads();
#pragma unroll
          for (int k = 0; k < kOwn; ++k) {
            const int c = kWaves * k + wv;
            if (c < kChunks && (uint32_t)(c * 64) < n_stage) {
              int carry = 0;
              for (int cc = 0; cc < c; ++cc) carry = max(carry, (int)s_cmax[cc]);
              const uint32_t j = (uint32_t)(c * 64 + lane), jj = min(j, Pm1);
              const int v = max(vv[k], carry);
              const uint2 t = rowtab[v - 1];          // (v >= 1: position 0 is the beginning of the first row with points)
              if (j < n_stage)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.gts + (t.x + (jj - t.y))),
                                                 (__attribute__((address_space(3))) void *)(pts + c * 64), 16, 0, 0);
            }
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else
        if (Ps) {
          const uint32_t Pm1 = Ps - 1u, n_stage = Ps + 3u;
          int carry = 0;
#pragma unroll
          for (int c = 0; c < (kStagePts + 3 + 63) / 64; ++c) {
            if ((uint32_t)(c * 64) < n_stage) {          // (uniform)
              const uint32_t j = (uint32_t)(c * 64 + lane), jj = min(j, Pm1);
              int v = j <= Pm1 ? (int)mark[j] : 0;
              v = max(v, dpp_keep<0x111>(v)); v = max(v, dpp_keep<0x112>(v)); v = max(v, dpp_keep<0x114>(v)); v = max(v, dpp_keep<0x118>(v));      // running maximum inside the rows of 16 lanes
              const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47), r3 = __builtin_amdgcn_readlane(v, 63);
              const int q4 = lane >> 4;
              const int before = q4 == 0 ? carry : q4 == 1 ? max(carry, r0) : q4 == 2 ? max(carry, max(r0, r1)) : max(carry, max(max(r0, r1), r2));
              v = max(v, before);
              carry = max(max(carry, max(r0, r1)), max(r2, r3));
              const uint2 t = rowtab[v - 1];          // (v >= 1: position 0 is the beginning of the first row with points)
              if (j < n_stage)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.gts + (t.x + (jj - t.y))),
                                                 (__attribute__((address_space(3))) void *)(pts + c * 64), 16, 0, 0);
            }
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the DMA writes count as vector memory operations of this wave
        }
        sync();
        MVR_CK();
        // ---- 4. every lane walks its own cells: a row's range in the grid-ordered array (asked for above, kBurst rows at a
        // time) becomes a range of staged positions through the row's table entry -- or stays what it is for a row that was
        // not staged
        if (swalk) {
          uint32_t ridx = (uint32_t)((bz0 - Z0) * NY + (by0 - Y0));
          int ryi = 0;
          for (int k0 = 0; k0 < nrows; k0 += kBurst) {
            if (k0) {
#pragma unroll
              for (int u = 0; u < kBurst; ++u)
                if (k0 + u < nrows) {
                  row_range(row_g, bs[u], be[u]);
                  row_g += row_step;
                  if (++yy == ny) { yy = 0; row_g += z_step_g; }
                }
            }
#pragma unroll
            for (int u = 0; u < kBurst; ++u)
              if (k0 + u < nrows) {
                uint2 t = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
                if (ridx < (uint32_t)kRowsT) t = rowtab[ridx];
                ridx += 1u;
                if (++ryi == ny) { ryi = 0; ridx += z_step_r; }
                n_eval += be[u] - bs[u];
                if (t.y != 0xFFFFFFFFu) {
                  const uint32_t ka = t.y + (bs[u] - t.x), kb = t.y + (be[u] - t.x);
                  for (uint32_t k = ka; k < kb; k += 4) {
                    const float4 *p4 = pts + k;
                    const float4 t0 = p4[0], t1 = p4[1], t2 = p4[2], t3 = p4[3];
                    take(t0); take(t1); take(t2); take(t3);
                  }
                } else {
                  for (uint32_t k = bs[u]; k < be[u]; k += 4) {
                    const float4 *__restrict__ p4 = a.gts + min(k, nt4);
                    const float4 t0 = p4[0], t1 = p4[1], t2 = p4[2], t3 = p4[3];
                    take(t0); take(t1); take(t2); take(t3);
                  }
                }
              }
          }
        }
      }
    }
    if (stage_stat && lane == 0 && why) atomicAdd(stage_stat + (size_t)((blockIdx.x * 4u + (uint32_t)wv) & 63u) * 64u + why + (a.qlist ? 8u : 0u), 1ull);      // (64 shards of 64 counters; + 8: a launch over a compacted query list, i.e. the reverse searches)
    // ---- 5. the rest of the prologue: what the probe found, where the query goes; a lane whose probe did not settle it (its new
    // ball reaches beyond the probed cells, or the probe found nothing) walks that ball the plain way
    if constexpr (kStageProbe) {
      if (staged) {
        if (live) route();
        if (walks && !wants_probe) walks = false;          // (walked above, from LDS)
      } else if (live) {
        if (wants_probe) walk_box(px0, px1, py0, py1, pz0, pz1);
        route();
      }
    } else if (staged) walks = false;                      // (walked above, from LDS)
  }
#ifdef MVR_STAGE_CLOCK
  if (!staged) { while (ckn < 4) { ck[ckn] = (long long)__builtin_readcyclecounter(); ++ckn; } }      // (a wave that was not staged: phases 1 / 2 hold whatever the attempt cost)
#endif
  if (walks) walk_box(x0, x1, y0, y1, z0, z1);      // (the plain walk: every walking lane without STAGE; with it, the lanes the staged step left over)
  MVR_CK();
  if (answers) {
    // the group's answer: the smallest (d2, index) of its lanes
#pragma unroll
    for (int o = 1; o < G; o <<= 1) {
      const unsigned long long ob = __shfl_xor(best, o, 64);
      if (ob < best) best = ob;
    }
    const float bd = __uint_as_float((uint32_t)(best >> 32));
    const uint32_t bi = (uint32_t)best;
    if (sub == 0) {
      const bool found = bi != kNone && bd <= cap2;
      const uint32_t ord = a.qlist ? pos : (a.key_by_pos ? qpos : __float_as_uint(q.w));
      uint32_t low = bi;
      if (found && (a.key_by_pos || a.mark)) {
        const uint32_t hp = a.tinv[bi];              // the match's position in its set's Hilbert order
        if (a.key_by_pos) low = hp;
        if (a.mark) __hip_atomic_store(&a.mark[hp], __float_as_uint(bd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      a.keys[ord] = found ? (((nnkey_t)__float_as_uint(bd) << 32) | low) : kKeyInit;
    }
  }
  if (G == 1 && a.cull_sets) {   // a wave = one 64-query set of the culled kernel: listed if any of its queries was flagged
    if (__ballot(went_cull) != 0ull && (threadIdx.x & 63) == 0) a.cull_sets[atomicAdd(a.cull_count, 1u)] = (set * kPerBlock + threadIdx.x) >> 6;
  }
  {   // the wave's wide queries: one counter bump per wave, ordinals in lane order
    const unsigned long long m = __ballot(went_wide);
    if (m) {
      const int lane = threadIdx.x & 63;
      uint32_t base = 0;
      if (lane == __ffsll((long long)m) - 1) base = atomicAdd(a.wide_count, (uint32_t)__popcll(m));
      base = (uint32_t)__shfl((int)base, __ffsll((long long)m) - 1, 64);
      if (went_wide) a.wide_list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = pos;
    }
  }
  if (evals) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_eval += __shfl_xor(n_eval, o, 64);
    if ((threadIdx.x & 63) == 0 && n_eval) {
      unsigned long long *s = evals + (size_t)((set * 4u + (threadIdx.x >> 6)) & (kEvalShards - 1)) * kEvalStride;
      atomicAdd(s, n_eval);
      atomicAdd(s + kEvalRegion, n_eval);
    }
  }
#ifdef MVR_STAGE_CLOCK
  MVR_CK();
  if (STAGE && stage_stat && (threadIdx.x & 63) == 0) {
    unsigned long long *o = stage_stat + (size_t)((blockIdx.x * 4u + (threadIdx.x >> 6)) & 63u) * 64u + 16 + (staged ? 8 : 0);
    for (int k = 0; k < 5; ++k) atomicAdd(o + k, (unsigned long long)(ck[k + 1] - ck[k]));
    atomicAdd(o + 7, 1ull);
  }
#endif
#undef MVR_CK
}

// ---- the wide bounded queries: ONE WAVE per query.  The rows of cells its ball overlaps are dealt to the lanes (one
// contiguous range of the grid-ordered array each), their lengths prefix-summed, and the candidates -- a few hundred
// for a 4 mm ball -- shared out evenly: lane l takes candidates l, l + 64, ...  (a thread walking them alone would hold
// its whole wave up: a wave is as slow as its slowest lane).
#ifndef MVR_TAIL_WAVES
#define MVR_TAIL_WAVES 4
#endif
constexpr int kWideWaves = MVR_TAIL_WAVES;      // waves per block of the stragglers' launches (wide queries: a wave each; listed sets: a block each)
template <bool FMA>
__device__ __forceinline__ void nn_grid_wide_body(const GridBatch &batch, unsigned long long *__restrict__ evals, const uint32_t bx, const uint32_t nbx)
{
  const GridPair &a = batch.p[blockIdx.y];
  if (!a.wide_count) return;
  __shared__ uint32_t row_s[kWideWaves][64], row_off[kWideWaves][65];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t n_wide = *a.wide_count, stride = nbx * kWideWaves;
  const float cap2 = batch.cap2;
  unsigned long long n_eval = 0;
  for (uint32_t i = bx * kWideWaves + wv; i < n_wide; i += stride) {
    const uint32_t pos = a.wide_list[i];
    const uint32_t qpos = a.qlist ? a.qlist[pos] : a.q_begin + pos;
    const float4 q = a.qs[qpos];
    float bound = cap2;
    if (a.qbound) { const uint32_t v = a.qbound[qpos]; if (v <= __float_as_uint(bound)) bound = __uint_as_float(v); }
    if (a.seed_from_keys) {
      const uint32_t prev = (uint32_t)a.keys[qpos];
      if (prev < a.nt) { const float d = gdist2<FMA>(a.gts[a.h2g[prev]], q.x, q.y, q.z); if (d <= bound) bound = d; }
    }
    const Ball ball = grid_ball(a, q, bound);
    const float rad = ball.rad, rx = ball.rx, ry = ball.ry, rz = ball.rz;
    const int x0 = cell_of(rx - rad, a.lo[0], a.inv_h, a.dim[0]), x1 = cell_of(rx + rad, a.lo[0], a.inv_h, a.dim[0]);
    const int y0 = cell_of(ry - rad, a.lo[1], a.inv_h, a.dim[1]), y1 = cell_of(ry + rad, a.lo[1], a.inv_h, a.dim[1]);
    const int z0 = cell_of(rz - rad, a.lo[2], a.inv_h, a.dim[2]), z1 = cell_of(rz + rad, a.lo[2], a.inv_h, a.dim[2]);
    const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
    nnkey_t best = ((nnkey_t)__float_as_uint(bound) << 32) | kNone;        // (d2 bits, original index): the minimum is the answer, ties to the lowest index
    uint32_t bk = 0;
    for (int r0 = 0; r0 < nrows; r0 += 64) {
      const int r = r0 + lane;
      uint32_t s = 0, len = 0;
      if (r < nrows) {
        const uint32_t row = (uint32_t)(((z0 + r / ny) * a.dim[1] + (y0 + r % ny)) * a.dim[0]);
        s = cell_start_of(a, row + (uint32_t)x0);
        len = cell_start_of(a, row + (uint32_t)x1 + 1u) - s;
      }
      uint32_t inc = len;                       // inclusive prefix sum over the lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)inc, o, 64); if (lane >= o) inc += v; }
      const uint32_t total = (uint32_t)__shfl((int)inc, 63, 64);
      row_s[wv][lane] = s; row_off[wv][lane] = inc - len;
      if (lane == 63) row_off[wv][64] = total;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (uint32_t j = lane; j < total; j += 64) {
        int lo = 0, hi = 63;                    // the row whose range holds candidate j: last row with off <= j
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (row_off[wv][mid] <= j) lo = mid; else hi = mid - 1; }
        const uint32_t k = row_s[wv][lo] + (j - row_off[wv][lo]);
        const float4 t = a.gts[k];
        const nnkey_t key = ((nnkey_t)__float_as_uint(gdist2<FMA>(t, q.x, q.y, q.z)) << 32) | __float_as_uint(t.w);
        if (key < best) { best = key; bk = k; }
      }
      n_eval += total;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // the wave's answer: the smallest (d2, index); the lane that holds it knows where it sits
    nnkey_t m = best;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const nnkey_t v = (nnkey_t)__shfl_xor((unsigned long long)m, o, 64); m = v < m ? v : m; }
    const unsigned long long who = __ballot(best == m && (uint32_t)best != kNone);
    const bool found = (uint32_t)m != kNone && __uint_as_float((uint32_t)(m >> 32)) <= cap2;
    if (found && who) {
      const int src = __ffsll((long long)who) - 1;
      const uint32_t wk = (uint32_t)__shfl((int)bk, src, 64);
      if (lane == 0) {
        const uint32_t ord = a.qlist ? pos : (a.key_by_pos ? qpos : __float_as_uint(q.w));
        uint32_t low = (uint32_t)m;
        if (a.key_by_pos || a.mark) {
          const uint32_t hp = a.g2h[wk];
          if (a.key_by_pos) low = hp;
          if (a.mark) __hip_atomic_store(&a.mark[hp], (uint32_t)(m >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        a.keys[ord] = (m & 0xFFFFFFFF00000000ull) | low;
      }
    } else if (lane == 0) {
      const uint32_t ord = a.qlist ? pos : (a.key_by_pos ? qpos : __float_as_uint(q.w));
      a.keys[ord] = kKeyInit;
    }
  }
  if (evals && lane == 0 && n_eval) {
    unsigned long long *s = evals + (size_t)((bx * kWideWaves + wv) & (kEvalShards - 1)) * kEvalStride;
    atomicAdd(s, n_eval);
    atomicAdd(s + kEvalRegion, n_eval);
  }
}

}  // namespace

// ---- build: once per point set, from a cloud that holds the set's canonical coordinates.  Batched: the bounding boxes of
// all the sets that need a grid come back in ONE round trip, every grid is one allocation (its arrays carved out of it), the
// temporaries live in a scratch buffer of the context that only ever grows, and nothing waits for the stream afterwards --
// a grid carries an event its first user waits for.  `on` is the stream the build is enqueued on: the context's own, or its
// side stream so that the grids of a registration are built WHILE its first pass searches (ring_passes).
namespace {
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
__global__ void bbox_many_partial_kernel(const float4 *const *__restrict__ pts, const unsigned long long *__restrict__ n, float *__restrict__ part)
{
  // blockIdx.y = cloud; 64 blocks per cloud; part[cloud][block][6]
  const int cl = blockIdx.y;
  const float4 *p = pts[cl];
  const size_t cnt = (size_t)n[cl];
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = p[i];
    lo[0] = fminf(lo[0], v.x); lo[1] = fminf(lo[1], v.y); lo[2] = fminf(lo[2], v.z);
    hi[0] = fmaxf(hi[0], v.x); hi[1] = fmaxf(hi[1], v.y); hi[2] = fmaxf(hi[2], v.z);
  }
  __shared__ float sh[4][6];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], o, 64)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o, 64)); }
  if ((threadIdx.x & 63) == 0) for (int k = 0; k < 3; ++k) { sh[threadIdx.x >> 6][k] = lo[k]; sh[threadIdx.x >> 6][3 + k] = hi[k]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) for (int k = 0; k < 3; ++k) { sh[0][k] = fminf(sh[0][k], sh[w][k]); sh[0][3 + k] = fmaxf(sh[0][3 + k], sh[w][3 + k]); }
    for (int k = 0; k < 6; ++k) part[((size_t)cl * gridDim.x + blockIdx.x) * 6 + k] = sh[0][k];
  }
}
}  // namespace

// ---- the COMPACT tables of several grids in one go (blockIdx.y = grid): cell ids of all scans sorted TOGETHER under the key
// (grid << 25) | cell id, then per grid the occupied segments of 32 cell ids -> a record of 32 starts each and the directory
// (CellGrid), and the distance map per 2 x 2 x 2 cells.  ~20 launches for all the grids of a registration where the dense
// tables took 16 stream operations per grid.
namespace {
struct CGArgs {
  const float4 *pts[kBatchClouds]; unsigned long long n[kBatchClouds], off[kBatchClouds], dtoff[kBatchClouds]; GridGeom geom[kBatchClouds];
  uint32_t *gperm[kBatchClouds]; float4 *graw[kBatchClouds]; uint32_t *dir[kBatchClouds], *recs[kBatchClouds]; uint32_t nseg[kBatchClouds];
  uint8_t *dt[kBatchClouds]; int dtdim[kBatchClouds][3]; int dt_steps[kBatchClouds];
  uint32_t *dense[kBatchClouds]; unsigned long long cells[kBatchClouds];      // optional: the dense table to expand the compact one into ([cells + 1 + kStartPad])
};
static_assert(sizeof(CGArgs) <= 4096, "CGArgs travels as a kernel argument");
__global__ void cg_cell_id_kernel(CGArgs a, uint32_t *__restrict__ key, uint32_t *__restrict__ idx)
{
  const int g = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)a.n[g]) return;
  const float4 v = a.pts[g][i];
  const GridGeom &gg = a.geom[g];
  const int x = cell_of(v.x, gg.lo[0], gg.inv_h, gg.dim[0]), y = cell_of(v.y, gg.lo[1], gg.inv_h, gg.dim[1]), z = cell_of(v.z, gg.lo[2], gg.inv_h, gg.dim[2]);
  key[(size_t)a.off[g] + i] = ((uint32_t)g << 25) | (uint32_t)((z * gg.dim[1] + y) * gg.dim[0] + x);
  idx[(size_t)a.off[g] + i] = (uint32_t)i;
}
// sorted: grid order of every grid; the first point of every occupied segment is flagged (the grid number is part of the key: a
// new grid is a new segment by itself); the distance map's cubes that hold a point are marked in its first buffer
__global__ void cg_heads_kernel(CGArgs a, const uint32_t *__restrict__ key, const uint32_t *__restrict__ idx, uint32_t *__restrict__ flag, uint8_t *__restrict__ dt_a)
{
  const int g = blockIdx.y;
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= (size_t)a.n[g]) return;
  const size_t at = (size_t)a.off[g] + k;
  const uint32_t ky = key[at], o = idx[at];
  flag[at] = (k == 0 || (key[at - 1] >> 5) != (ky >> 5)) ? 1u : 0u;
  a.gperm[g][k] = o;
  float4 v = a.pts[g][o];
  v.w = __uint_as_float(o);
  a.graw[g][k] = v;
  const GridGeom &gg = a.geom[g];
  const uint32_t cid = ky & 0x1FFFFFFu;
  const int x = (int)(cid % (uint32_t)gg.dim[0]), y = (int)((cid / (uint32_t)gg.dim[0]) % (uint32_t)gg.dim[1]), z = (int)(cid / ((uint32_t)gg.dim[0] * (uint32_t)gg.dim[1]));
  dt_a[(size_t)a.dtoff[g] + ((size_t)(z >> 1) * a.dtdim[g][1] + (y >> 1)) * a.dtdim[g][0] + (x >> 1)] = 0;
}
// per occupied segment (its first point): the record of its 32 starts (a merge of the segment's sorted cell ids with 0 .. 31),
// and the segment's id and first position for the directory kernel
__global__ void cg_records_kernel(CGArgs a, const uint32_t *__restrict__ key, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ slotg,
                                  uint32_t *__restrict__ occseg, uint32_t *__restrict__ occpos)
{
  const int g = blockIdx.y;
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)a.n[g], off = (size_t)a.off[g];
  if (k >= n || !flag[off + k]) return;
  const uint32_t slot = slotg[off + k] - slotg[off];
  const uint32_t seg_key = key[off + k] >> 5;
  occseg[off + slot] = seg_key & ((1u << 20) - 1u);       // (the grid's bits shifted out: 25 - 5 = 20 bits of segment id)
  occpos[off + slot] = (uint32_t)k;
  uint32_t *rec = a.recs[g] + ((size_t)slot << 5);
  size_t p = k;
  for (uint32_t j = 0; j < 32u; ++j) {
    while (p < n && (key[off + p] >> 5) == seg_key && (key[off + p] & 31u) < j) ++p;
    rec[j] = (uint32_t)p;
  }
}
// the directory: an occupied segment points to its record, an empty one holds the position its cells begin at -- the first
// point of the next occupied segment (n behind the last)
__global__ void cg_dir_kernel(CGArgs a, const uint32_t *__restrict__ slotg, const uint32_t *__restrict__ occseg, const uint32_t *__restrict__ occpos)
{
  const int g = blockIdx.y;
  const size_t sg = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (sg >= (size_t)a.nseg[g] + 2) return;
  const size_t n = (size_t)a.n[g], off = (size_t)a.off[g];
  const uint32_t nocc = slotg[off + n] - slotg[off];
  uint32_t lo = 0, hi = nocc;                        // first occupied segment with id >= sg
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (occseg[off + mid] < (uint32_t)sg) lo = mid + 1; else hi = mid; }
  a.dir[g][sg] = (lo < nocc && occseg[off + lo] == (uint32_t)sg) ? (lo | kSegOcc) : (lo < nocc ? occpos[off + lo] : (uint32_t)n);
}
// the DENSE table from the compact one (a grid whose walks are to read one entry per lookup: the dense form is 8-11 % faster to
// walk, the compact one five times faster to build -- so it is built compact and expanded: one streaming launch)
__global__ void cg_expand_kernel(CGArgs a)
{
  const int g = blockIdx.y;
  if (!a.dense[g]) return;
  const size_t cells = (size_t)a.cells[g], tot = cells + 1 + kStartPad;
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < tot; c += (size_t)gridDim.x * blockDim.x) {
    uint32_t v = (uint32_t)a.n[g];
    if (c <= cells) {
      const uint32_t d = a.dir[g][c >> 5];
      v = (d & kSegOcc) ? a.recs[g][((size_t)(d & ~kSegOcc) << 5) + (c & 31u)] : d;
    }
    a.dense[g][c] = v;
  }
}
__global__ void cg_fill_kernel(CGArgs a, uint8_t *__restrict__ dt_a)
{
  const int g = blockIdx.y;
  const size_t cells2 = (size_t)a.dtdim[g][0] * a.dtdim[g][1] * a.dtdim[g][2];
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells2; c += (size_t)gridDim.x * blockDim.x) dt_a[(size_t)a.dtoff[g] + c] = 255;
}
// dt_axis_kernel for all the grids of the batch: from -> to along `axis`; pass 1 of 3 writes the grids' own arrays (to_own), the
// others the scratch buffer -- (a, own, a, own): scratch -> own -> scratch -> own
__global__ void cg_dt_axis_kernel(CGArgs a, const uint8_t *__restrict__ from_s, uint8_t *__restrict__ to_s, int axis, int from_own, int to_own)
{
  const int g = blockIdx.y;
  const int nx = a.dtdim[g][0], ny = a.dtdim[g][1], nz = a.dtdim[g][2], steps = a.dt_steps[g];
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= (size_t)nx * ny * nz) return;
  const uint8_t *in = from_own ? a.dt[g] : from_s + (size_t)a.dtoff[g];
  uint8_t *out = to_own ? a.dt[g] : to_s + (size_t)a.dtoff[g];
  const int x = (int)(c % nx), y = (int)((c / nx) % ny), z = (int)(c / ((size_t)nx * ny));
  const int pos = axis == 0 ? x : axis == 1 ? y : z, len = axis == 0 ? nx : axis == 1 ? ny : nz;
  const size_t stride = axis == 0 ? 1 : axis == 1 ? (size_t)nx : (size_t)nx * ny;
  int best = (int)in[c];
  for (int d = 1; d <= steps && best > d; ++d) {
    if (pos - d >= 0) best = min(best, max(d, (int)in[c - (size_t)d * stride]));
    if (pos + d < len) best = min(best, max(d, (int)in[c + (size_t)d * stride]));
  }
  out[c] = (uint8_t)(best > steps ? 255 : best);
}
}  // namespace

static int build_compact_grids(Ctx *c, const std::vector<Cloud *> &clouds, const std::vector<std::shared_ptr<CellGrid> > &grids, hipStream_t on)
{
  for (size_t base = 0; base < clouds.size(); base += kBatchClouds) {
    const int m = (int)std::min<size_t>(kBatchClouds, clouds.size() - base);
    CGArgs a;
    size_t total = 0, nmax = 0, dt_total = 0, dt_max = 0, seg_max = 0, tmp_tables = 0, cells_max = 0;
    std::vector<size_t> o_dir((size_t)m, 0), o_rec((size_t)m, 0);
    for (int k = 0; k < kBatchClouds; ++k) {
      a.dense[k] = nullptr; a.cells[k] = 0;
      a.pts[k] = nullptr; a.n[k] = a.off[k] = a.dtoff[k] = 0; a.gperm[k] = nullptr; a.graw[k] = nullptr; a.dir[k] = a.recs[k] = nullptr; a.nseg[k] = 0; a.dt[k] = nullptr;
      a.dtdim[k][0] = a.dtdim[k][1] = a.dtdim[k][2] = 1; a.dt_steps[k] = 0; a.geom[k] = GridGeom{{0, 0, 0}, 1.f, {1, 1, 1}};
    }
    for (int k = 0; k < m; ++k) {
      const Cloud &cl = *clouds[base + (size_t)k];
      const CellGrid &g = *grids[base + (size_t)k];
      a.pts[k] = cl.pts; a.n[k] = g.n; a.off[k] = total; a.dtoff[k] = dt_total;
      for (int j = 0; j < 3; ++j) { a.geom[k].lo[j] = g.lo[j]; a.geom[k].dim[j] = g.dim[j]; a.dtdim[k][j] = g.dtdim[j]; }
      a.geom[k].inv_h = g.inv_h;
      a.gperm[k] = g.gperm; a.graw[k] = g.graw; a.dir[k] = g.dir; a.recs[k] = g.recs; a.nseg[k] = g.nseg; a.dt[k] = g.dt; a.dt_steps[k] = g.dt_steps;
      const size_t c2 = (size_t)g.dtdim[0] * g.dtdim[1] * g.dtdim[2];
      total += g.n; nmax = std::max(nmax, g.n); dt_total += align256(c2); dt_max = std::max(dt_max, c2); seg_max = std::max(seg_max, (size_t)g.nseg + 2);
      if (g.start) {          // a dense grid built through the compact form: directory and records are temporaries
        const size_t cells = (size_t)g.dim[0] * g.dim[1] * g.dim[2], max_slots = std::min(g.n, (size_t)g.nseg + 1) + 1;
        a.dense[k] = g.start; a.cells[k] = cells; cells_max = std::max(cells_max, cells + 1 + kStartPad);
        o_dir[(size_t)k] = tmp_tables; tmp_tables += align256(((size_t)g.nseg + 2) * 4);
        o_rec[(size_t)k] = tmp_tables; tmp_tables += align256(max_slots * 32 * 4);
      }
    }
    if (total + 1 > 0x7FFFFFFFull) return set_error(c, MVR_E_ARG, "grid build: too many points in one batch");
    int hi_bit = 25;
    while ((1 << (hi_bit - 25)) < m) ++hi_bit;
    size_t sort_bytes = 0, scan_bytes = 0;
    { uint32_t *z = nullptr;
      MVR_HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, z, z, z, z, (int)total, 0, hi_bit, on));
      MVR_HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, z, z, (int)(total + 1), on)); }
    const size_t cub_bytes = std::max(sort_bytes, scan_bytes) + 256;
    const size_t w = align256((total + 1) * 4);
    const size_t o_ka = 0, o_kb = o_ka + w, o_ia = o_kb + w, o_ib = o_ia + w, o_fl = o_ib + w, o_sl = o_fl + w, o_os = o_sl + w, o_op = o_os + w, o_dt = o_op + w,
                 o_cub = o_dt + align256(dt_total), o_tab = o_cub + align256(cub_bytes), need = o_tab + tmp_tables;
    if (c->scratch_cap < need && c->side_stream) (void)hipStreamSynchronize(c->side_stream);      // (the buffer is about to be replaced: nobody may still be using it)
    if (int rc = ensure(c, c->scratch, c->scratch_cap, need)) return rc;
    uint32_t *ka = reinterpret_cast<uint32_t *>(c->scratch + o_ka), *kb = reinterpret_cast<uint32_t *>(c->scratch + o_kb), *ia = reinterpret_cast<uint32_t *>(c->scratch + o_ia),
             *ib = reinterpret_cast<uint32_t *>(c->scratch + o_ib), *fl = reinterpret_cast<uint32_t *>(c->scratch + o_fl), *sl = reinterpret_cast<uint32_t *>(c->scratch + o_sl),
             *os = reinterpret_cast<uint32_t *>(c->scratch + o_os), *op = reinterpret_cast<uint32_t *>(c->scratch + o_op);
    uint8_t *dt_s = reinterpret_cast<uint8_t *>(c->scratch + o_dt);
    void *cub = c->scratch + o_cub;
    for (int k = 0; k < m; ++k)
      if (a.dense[k]) { a.dir[k] = reinterpret_cast<uint32_t *>(c->scratch + o_tab + o_dir[(size_t)k]); a.recs[k] = reinterpret_cast<uint32_t *>(c->scratch + o_tab + o_rec[(size_t)k]); }
    const unsigned nb = (unsigned)((nmax + 255) / 256), db = (unsigned)((dt_max + 255) / 256), sb = (unsigned)((seg_max + 255) / 256);
    hipLaunchKernelGGL(cg_cell_id_kernel, dim3(nb, (unsigned)m), dim3(256), 0, on, a, ka, ia);
    size_t b1 = cub_bytes;
    MVR_HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(cub, b1, ka, kb, ia, ib, (int)total, 0, hi_bit, on));
    hipLaunchKernelGGL(cg_fill_kernel, dim3(std::min(db, 1024u), (unsigned)m), dim3(256), 0, on, a, dt_s);
    MVR_HIP_TRY(c, hipMemsetAsync(fl + total, 0, 4, on));
    hipLaunchKernelGGL(cg_heads_kernel, dim3(nb, (unsigned)m), dim3(256), 0, on, a, kb, ib, fl, dt_s);
    size_t b2 = cub_bytes;
    MVR_HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(cub, b2, fl, sl, (int)(total + 1), on));
    hipLaunchKernelGGL(cg_records_kernel, dim3(nb, (unsigned)m), dim3(256), 0, on, a, kb, fl, sl, os, op);
    hipLaunchKernelGGL(cg_dir_kernel, dim3(sb, (unsigned)m), dim3(256), 0, on, a, sl, os, op);
    if (cells_max) hipLaunchKernelGGL(cg_expand_kernel, dim3((unsigned)std::min<size_t>((cells_max + 255) / 256, 4096), (unsigned)m), dim3(256), 0, on, a);
    hipLaunchKernelGGL(cg_dt_axis_kernel, dim3(db, (unsigned)m), dim3(256), 0, on, a, dt_s, dt_s, 0, 0, 1);
    hipLaunchKernelGGL(cg_dt_axis_kernel, dim3(db, (unsigned)m), dim3(256), 0, on, a, dt_s, dt_s, 1, 1, 0);
    hipLaunchKernelGGL(cg_dt_axis_kernel, dim3(db, (unsigned)m), dim3(256), 0, on, a, dt_s, dt_s, 2, 0, 1);
    MVR_HIP_TRY(c, hipGetLastError());
  }
  return MVR_OK;
}

int ensure_grids(Ctx *c, Cloud *const *canon, int count, double reach, hipStream_t on, hipEvent_t after)
{
  std::vector<Cloud *> todo;
  for (int k = 0; k < count; ++k) {
    Cloud *cl = canon[k];
    if (!cl || !cl->canonical || cl->n < 4 || cl->n > 0x7FFFFFFFull) continue;      // (hipCUB's sort takes an int count, the walk reads four points at a time; such a cloud keeps the culled kernel)
    if (cl->grid && cl->grid->n == cl->n) continue;
    cl->grid.reset();
    auto it = c->grids.find(cl->set_id);
    if (it != c->grids.end()) { cl->grid = it->second.lock(); if (cl->grid && cl->grid->n == cl->n) continue; cl->grid.reset(); }
    if (std::find(todo.begin(), todo.end(), cl) == todo.end()) todo.push_back(cl);
  }
  if (todo.empty()) return MVR_OK;
  MVR_MAY_BLOCK(c, "a point set has no grid yet");
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  const bool side = on != c->stream;
  // a side stream starts behind `after` (an event of the context's stream from BEFORE the pass it is to overlap), and any
  // stream behind the previous user of the scratch buffer
  if (side && after) MVR_HIP_TRY(c, hipStreamWaitEvent(on, after, 0));
  if (c->scratch_event && c->scratch_stream != on) MVR_HIP_TRY(c, hipStreamWaitEvent(on, c->scratch_event, 0));
  // ---- bounding boxes of all of them: one launch, one round trip
  const int m = (int)todo.size();
  constexpr int kBoxBlocks = 64;
  constexpr size_t kCellsCap = 32000000;
  std::vector<const float4 *> hp((size_t)m); std::vector<unsigned long long> hn((size_t)m);
  size_t nmax = 0, cells_max = 0;
  for (int k = 0; k < m; ++k) { hp[(size_t)k] = todo[(size_t)k]->pts; hn[(size_t)k] = todo[(size_t)k]->n; nmax = std::max(nmax, todo[(size_t)k]->n); }
  const size_t head = align256((size_t)m * 8) * 2 + align256((size_t)m * kBoxBlocks * 6 * sizeof(float));
  bool have_boxes = true;          // every cloud brought its bounding box along from its upload: no launch, no round trip
  for (int k = 0; k < m; ++k) have_boxes = have_boxes && todo[(size_t)k]->bbox_set == todo[(size_t)k]->set_id && todo[(size_t)k]->bbox_set != 0;
  const bool legacy = c->grid_index == 0;          // (the round-3 build: its temporaries; the other forms size their own in build_compact_grids)
  // (sized ONCE for the largest grid there can be: growing it later would mean freeing it under the pass this build overlaps)
  size_t sort_bytes = 0, scan_bytes = 0;
  if (legacy) {
    uint32_t *z = nullptr;
    MVR_HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, z, z, z, z, (int)nmax, 0, 32, on));
    auto rit = thrust_like_reverse(z, kCellsCap + 1);
    MVR_HIP_TRY(c, hipcub::DeviceScan::InclusiveScan(nullptr, scan_bytes, rit, rit, MinU32(), (int)(kCellsCap + 1), on));
  }
  const size_t cub_bytes = std::max(sort_bytes, scan_bytes) + 256;
  const size_t o_cid_a = 0, o_cid_b = o_cid_a + align256(nmax * 4), o_idx = o_cid_b + align256(nmax * 4), o_dtt = o_idx + align256(nmax * 4),
               o_cub = o_dtt + align256(kCellsCap), tmp_total = o_cub + align256(cub_bytes);
  if (legacy || !have_boxes) { if (int rc = ensure(c, c->scratch, c->scratch_cap, legacy ? std::max(head, tmp_total) : head)) return rc; }
  const float4 **d_pts = reinterpret_cast<const float4 **>(c->scratch);
  unsigned long long *d_n = reinterpret_cast<unsigned long long *>(c->scratch + align256((size_t)m * 8));
  float *d_part = reinterpret_cast<float *>(c->scratch + 2 * align256((size_t)m * 8));
  std::vector<float> h_part((size_t)m * kBoxBlocks * 6);
  if (have_boxes) {
    for (int k = 0; k < m; ++k)
      for (int b = 0; b < kBoxBlocks; ++b)
        for (int j = 0; j < 6; ++j) h_part[((size_t)k * kBoxBlocks + b) * 6 + j] = todo[(size_t)k]->bbox[j];
  } else {
  MVR_HIP_TRY(c, hipMemcpyAsync(d_pts, hp.data(), (size_t)m * 8, hipMemcpyHostToDevice, on));
  MVR_HIP_TRY(c, hipMemcpyAsync(d_n, hn.data(), (size_t)m * 8, hipMemcpyHostToDevice, on));
  hipLaunchKernelGGL(bbox_many_partial_kernel, dim3(kBoxBlocks, (unsigned)m), dim3(256), 0, on, d_pts, d_n, d_part);
  MVR_HIP_TRY(c, hipMemcpyAsync(h_part.data(), d_part, h_part.size() * sizeof(float), hipMemcpyDeviceToHost, on));
  MVR_HIP_TRY(c, hipStreamSynchronize(on));
  }
  host_mark("    grids: bounding boxes back");
  // ---- geometry and one allocation per grid
  std::vector<std::shared_ptr<CellGrid> > gs((size_t)m);
  for (int k = 0; k < m; ++k) {
    const size_t n = todo[(size_t)k]->n;
    float hb[6] = {3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    for (int b = 0; b < kBoxBlocks; ++b)
      for (int j = 0; j < 3; ++j) {
        hb[j] = std::min(hb[j], h_part[((size_t)k * kBoxBlocks + b) * 6 + j]);
        hb[3 + j] = std::max(hb[3 + j], h_part[((size_t)k * kBoxBlocks + b) * 6 + 3 + j]);
      }
    auto g = std::make_shared<CellGrid>();
    g->n = n;
    double ext[3];
    for (int j = 0; j < 3; ++j) { g->lo[j] = hb[j]; ext[j] = std::max(1e-6, (double)hb[3 + j] - (double)hb[j]); }
    // ~grid_cell_points points per cell on a SURFACE of about the bounding box's face area (a scan is a sheet, not a
    // volume); never more than 512 cells per axis / 32 M cells (128 MB of cell starts, of which only the cells near the
    // surface are ever read)
    const double area = ext[0] * ext[1] + ext[1] * ext[2] + ext[0] * ext[2];
    double h = std::sqrt((double)std::max(1, c->grid_cell_points) * area / (double)n);
    h = std::max(h, std::max(ext[0], std::max(ext[1], ext[2])) / 512.0);
    for (;;) {
      double cells = 1.0;
      for (int j = 0; j < 3; ++j) { g->dim[j] = (int)std::floor(ext[j] / h) + 1; cells *= g->dim[j]; }
      if (cells <= 32.0e6) break;
      h *= 1.25;
    }
    g->h = (float)h; g->inv_h = (float)(1.0 / h);
    const size_t cells = (size_t)g->dim[0] * g->dim[1] * g->dim[2];
    cells_max = std::max(cells_max, cells);
    const bool compact = c->grid_index == 1;
    if (c->grid_index == 2) {
      // the default: the DENSE table (one entry per lookup: 8-11 % faster walks), built through the compact form (its sort of all
      // scans at once, its records, its distance map per 2 x 2 x 2 cells) and expanded by one streaming launch
      g->dt_shift = 1;
      for (int j = 0; j < 3; ++j) g->dtdim[j] = (g->dim[j] + 1) >> 1;
      g->nseg = (uint32_t)((cells + 1 + 31) / 32);
      const size_t cells2 = (size_t)g->dtdim[0] * g->dtdim[1] * g->dtdim[2];
      const size_t o_start = 0, o_gperm = o_start + align256((cells + 1 + kStartPad) * 4), o_graw = o_gperm + align256(n * 4), o_g2h = o_graw + align256(n * sizeof(float4)),
                   o_h2g = o_g2h + align256(n * 4), o_dt = o_h2g + align256(n * 4), total = o_dt + align256(cells2);
      if (hipMalloc(&g->block, total) != hipSuccess) { (void)hipGetLastError(); return set_error(c, MVR_E_NOMEM, "grid build: out of device memory"); }
      g->start = reinterpret_cast<uint32_t *>(g->block + o_start); g->gperm = reinterpret_cast<uint32_t *>(g->block + o_gperm);
      g->graw = reinterpret_cast<float4 *>(g->block + o_graw); g->g2h = reinterpret_cast<uint32_t *>(g->block + o_g2h);
      g->h2g = reinterpret_cast<uint32_t *>(g->block + o_h2g); g->dt = reinterpret_cast<uint8_t *>(g->block + o_dt);
      const int steps2 = (int)std::ceil(std::max(0.0, reach) / (2.0 * h)) + 2;
      g->dt_steps = std::min(kGridDtMax, std::max(2, steps2));
      g->via_compact = true;
      gs[(size_t)k] = g;
      continue;
    }
    if (compact) {
      // segment directory + a record per occupied segment (at most one per point, at most one per segment) + the distance map per
      // 2 x 2 x 2 cells; gperm / graw / g2h / h2g as in the dense form
      g->nseg = (uint32_t)((cells + 1 + 31) / 32);
      g->dt_shift = 1;
      for (int j = 0; j < 3; ++j) g->dtdim[j] = (g->dim[j] + 1) >> 1;
      const size_t cells2 = (size_t)g->dtdim[0] * g->dtdim[1] * g->dtdim[2], max_slots = std::min(n, (size_t)g->nseg + 1) + 1;
      const size_t o_dir = 0, o_recs = o_dir + align256(((size_t)g->nseg + 2) * 4), o_gperm = o_recs + align256(max_slots * 32 * 4), o_graw = o_gperm + align256(n * 4),
                   o_g2h = o_graw + align256(n * sizeof(float4)), o_h2g = o_g2h + align256(n * 4), o_dt = o_h2g + align256(n * 4), total = o_dt + align256(cells2);
      if (hipMalloc(&g->block, total) != hipSuccess) { (void)hipGetLastError(); return set_error(c, MVR_E_NOMEM, "grid build: out of device memory"); }
      g->dir = reinterpret_cast<uint32_t *>(g->block + o_dir); g->recs = reinterpret_cast<uint32_t *>(g->block + o_recs);
      g->gperm = reinterpret_cast<uint32_t *>(g->block + o_gperm); g->graw = reinterpret_cast<float4 *>(g->block + o_graw);
      g->g2h = reinterpret_cast<uint32_t *>(g->block + o_g2h); g->h2g = reinterpret_cast<uint32_t *>(g->block + o_h2g); g->dt = reinterpret_cast<uint8_t *>(g->block + o_dt);
      const int steps2 = (int)std::ceil(std::max(0.0, reach) / (2.0 * h)) + 2;      // in cubes of two cells
      g->dt_steps = std::min(kGridDtMax, std::max(2, steps2));
      gs[(size_t)k] = g;
      continue;
    }
    const size_t o_start = 0, o_gperm = o_start + align256((cells + 1 + kStartPad) * 4), o_graw = o_gperm + align256(n * 4), o_g2h = o_graw + align256(n * sizeof(float4)),
                 o_h2g = o_g2h + align256(n * 4), o_dt = o_h2g + align256(n * 4), total = o_dt + align256(cells);
    if (hipMalloc(&g->block, total) != hipSuccess) { (void)hipGetLastError(); return set_error(c, MVR_E_NOMEM, "grid build: out of device memory"); }
    g->start = reinterpret_cast<uint32_t *>(g->block + o_start); g->gperm = reinterpret_cast<uint32_t *>(g->block + o_gperm);
    g->graw = reinterpret_cast<float4 *>(g->block + o_graw); g->g2h = reinterpret_cast<uint32_t *>(g->block + o_g2h);
    g->h2g = reinterpret_cast<uint32_t *>(g->block + o_h2g); g->dt = reinterpret_cast<uint8_t *>(g->block + o_dt);
    for (int j = 0; j < 3; ++j) g->dtdim[j] = g->dim[j];
    int steps = (int)std::ceil(std::max(0.0, reach) / h) + 2;       // enough to rule out `reach` (dt - 1 cell edges > reach + a cell), at most kGridDtMax
    g->dt_steps = std::min(kGridDtMax, std::max(2, steps));
    gs[(size_t)k] = g;
  }
  host_mark("    grids: allocated");
  {   // the compact grids: all of them together
    std::vector<Cloud *> cc; std::vector<std::shared_ptr<CellGrid> > cg;
    for (int k = 0; k < m; ++k) if (gs[(size_t)k]->dir || gs[(size_t)k]->via_compact) { cc.push_back(todo[(size_t)k]); cg.push_back(gs[(size_t)k]); }
    if (!cc.empty()) {
      if (int rc = build_compact_grids(c, cc, cg, on)) return rc;
      hipEvent_t ev = nullptr;          // (one event for the batch: every grid of it is ready behind it)
      for (size_t k = 0; k < cc.size(); ++k) {
        if (side) {
          MVR_HIP_TRY(c, hipEventCreateWithFlags(&cg[k]->ready, hipEventDisableTiming));
          MVR_HIP_TRY(c, hipEventRecord(cg[k]->ready, on));
        }
        cc[k]->grid = cg[k];
        c->grids[cc[k]->set_id] = cg[k];
      }
      (void)ev;
    }
  }
  // ---- temporaries: cell ids (two buffers for the sort), indices, the distance map's second buffer, hipCUB's scratch
  (void)cells_max;
  uint32_t *cid_a = reinterpret_cast<uint32_t *>(c->scratch + o_cid_a), *cid_b = reinterpret_cast<uint32_t *>(c->scratch + o_cid_b),
           *idx_a = reinterpret_cast<uint32_t *>(c->scratch + o_idx);
  uint8_t *dt_tmp = reinterpret_cast<uint8_t *>(c->scratch + o_dtt);
  void *cub = c->scratch + o_cub;
  for (int k = 0; k < m; ++k) {
    if (gs[(size_t)k]->dir || gs[(size_t)k]->via_compact) continue;
    Cloud &cl = *todo[(size_t)k];
    CellGrid &g = *gs[(size_t)k];
    const size_t n = g.n, cells = (size_t)g.dim[0] * g.dim[1] * g.dim[2];
    GridGeom gg;
    for (int j = 0; j < 3; ++j) { gg.lo[j] = g.lo[j]; gg.dim[j] = g.dim[j]; }
    gg.inv_h = g.inv_h;
    const unsigned nb = (unsigned)((n + 255) / 256), cb = (unsigned)((cells + 255) / 256);
    hipLaunchKernelGGL(cell_id_kernel, dim3(nb), dim3(256), 0, on, cl.pts, n, gg, cid_a, idx_a);
    int bits = 1;
    while (((size_t)1 << bits) < cells) ++bits;
    size_t b1 = cub_bytes;
    MVR_HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(cub, b1, cid_a, cid_b, idx_a, g.gperm, (int)n, 0, bits, on));
    MVR_HIP_TRY(c, hipMemsetAsync(g.start, 0xFF, (cells + 1 + kStartPad) * 4, on));
    hipLaunchKernelGGL(cell_first_kernel, dim3(nb), dim3(256), 0, on, cid_b, n, cells, g.start);
    auto rit = thrust_like_reverse(g.start, cells + 1);
    size_t b2 = cub_bytes;
    MVR_HIP_TRY(c, hipcub::DeviceScan::InclusiveScan(cub, b2, rit, rit, MinU32(), (int)(cells + 1), on));
    hipLaunchKernelGGL(pad_start_kernel, dim3(1), dim3(64), 0, on, g.start, cells, (uint32_t)n);
    hipLaunchKernelGGL(grid_gather_kernel, dim3(nb), dim3(256), 0, on, cl.pts, g.gperm, n, g.graw);
    hipLaunchKernelGGL(dt_occupancy_kernel, dim3(cb), dim3(256), 0, on, g.start, cells, dt_tmp);
    hipLaunchKernelGGL(dt_axis_kernel, dim3(cb), dim3(256), 0, on, dt_tmp, g.dt, g.dim[0], g.dim[1], g.dim[2], 0, g.dt_steps);
    hipLaunchKernelGGL(dt_axis_kernel, dim3(cb), dim3(256), 0, on, g.dt, dt_tmp, g.dim[0], g.dim[1], g.dim[2], 1, g.dt_steps);
    hipLaunchKernelGGL(dt_axis_kernel, dim3(cb), dim3(256), 0, on, dt_tmp, g.dt, g.dim[0], g.dim[1], g.dim[2], 2, g.dt_steps);
    MVR_HIP_TRY(c, hipGetLastError());
    if (side) {               // whoever uses the grid first on another stream waits for this
      MVR_HIP_TRY(c, hipEventCreateWithFlags(&g.ready, hipEventDisableTiming));
      MVR_HIP_TRY(c, hipEventRecord(g.ready, on));
    }
    cl.grid = gs[(size_t)k];
    c->grids[cl.set_id] = gs[(size_t)k];
  }
  for (auto i2 = c->grids.begin(); i2 != c->grids.end();) i2 = i2->second.expired() ? c->grids.erase(i2) : std::next(i2);
  host_mark("    grids: builds enqueued");
  if (!c->scratch_event) MVR_HIP_TRY(c, hipEventCreateWithFlags(&c->scratch_event, hipEventDisableTiming));
  MVR_HIP_TRY(c, hipEventRecord(c->scratch_event, on));
  c->scratch_stream = on;
  return MVR_OK;
}

bool ensure_grid(Ctx *c, Cloud &canon, double reach)
{
  Cloud *one = &canon;
  if (ensure_grids(c, &one, 1, reach, c->stream, nullptr) != MVR_OK) return false;
  return canon.grid != nullptr && canon.grid->n == canon.n;
}

// ---- the grid of a cloud over its OWN coordinates: the growing model of the sequential mode (`*target += aligned source`,
// registrator.cpp:576) searched as ONE point set -- one walk per query, where the part-by-part search (nn_parts_kernel) walks a
// ball in each of the ~5 merged scans that cover a query's surface.  Compact form (segment directory + records), points in grid
// order with w = the point's index in the cloud: the keys of a walk over it are the keys of a search of the merged cloud.
int ensure_model_grid(Ctx *c, Cloud &t, double reach, bool *ok)
{
  *ok = false;
  if (t.n < 4 || t.n > 0x7FFFFFF0ull || !t.segs.empty() || t.pts_stale) return MVR_OK;
  if (t.mgrid && t.mgrid_ok && t.mgrid->n == t.n) { *ok = true; return MVR_OK; }
  t.mgrid.reset(); t.mgrid_ok = false;
  MVR_MAY_BLOCK(c, "the model has no grid yet");
  float bb[6];
  if (int rc = cloud_bbox(c, t.pts, t.n, bb)) return rc;
  auto g = std::make_shared<CellGrid>();
  g->n = t.n;
  double ext[3];
  for (int j = 0; j < 3; ++j) { g->lo[j] = bb[j]; ext[j] = std::max(1e-6, (double)bb[3 + j] - (double)bb[j]); }
  const double area = ext[0] * ext[1] + ext[1] * ext[2] + ext[0] * ext[2];
  double h = std::sqrt((double)std::max(1, c->seq_cell_points) * area / (double)t.n);
  h = std::max(h, std::max(ext[0], std::max(ext[1], ext[2])) / 512.0);
  for (;;) {
    double cells = 1.0;
    for (int j = 0; j < 3; ++j) { g->dim[j] = (int)std::floor(ext[j] / h) + 1; cells *= g->dim[j]; }
    if (cells <= 32.0e6) break;
    h *= 1.25;
  }
  g->h = (float)h; g->inv_h = (float)(1.0 / h);
  const size_t cells = (size_t)g->dim[0] * g->dim[1] * g->dim[2], n = t.n;
  g->nseg = (uint32_t)((cells + 1 + 31) / 32);
  g->dt_shift = 1;
  for (int j = 0; j < 3; ++j) g->dtdim[j] = (g->dim[j] + 1) >> 1;
  const size_t cells2 = (size_t)g->dtdim[0] * g->dtdim[1] * g->dtdim[2], max_slots = std::min(n, (size_t)g->nseg + 1) + 1;
  const size_t o_dir = 0, o_recs = o_dir + align256(((size_t)g->nseg + 2) * 4), o_gperm = o_recs + align256(max_slots * 32 * 4), o_graw = o_gperm + align256(n * 4),
               o_dt = o_graw + align256((n + 4) * sizeof(float4)), total = o_dt + align256(cells2);
  if (hipMalloc(&g->block, total) != hipSuccess) { (void)hipGetLastError(); return set_error(c, MVR_E_NOMEM, "model grid: out of device memory"); }
  g->dir = reinterpret_cast<uint32_t *>(g->block + o_dir); g->recs = reinterpret_cast<uint32_t *>(g->block + o_recs);
  g->gperm = reinterpret_cast<uint32_t *>(g->block + o_gperm); g->graw = reinterpret_cast<float4 *>(g->block + o_graw);
  g->dt = reinterpret_cast<uint8_t *>(g->block + o_dt);
  const int steps2 = (int)std::ceil(std::max(0.0, reach) / (2.0 * h)) + 2;
  g->dt_steps = std::min(kGridDtMax, std::max(2, steps2));
  std::vector<Cloud *> cc{&t}; std::vector<std::shared_ptr<CellGrid> > cg{g};
  if (int rc = build_compact_grids(c, cc, cg, c->stream)) return rc;
  if (!c->scratch_event) MVR_HIP_TRY(c, hipEventCreateWithFlags(&c->scratch_event, hipEventDisableTiming));
  MVR_HIP_TRY(c, hipEventRecord(c->scratch_event, c->stream));
  c->scratch_stream = c->stream;
  t.mgrid = g; t.mgrid_ok = true;
  *ok = true;
  return MVR_OK;
}

// the forward searches of q's Hilbert-ordered points in t's model grid: keys by the query's ORIGINAL index, low word = the match's
// index in t (what the culled kernel over t's composite ordering leaves)
GridPair make_model_pair(const Cloud &q, const Cloud &t, nnkey_t *keys)
{
  GridPair p;
  const CellGrid &g = *t.mgrid;
  p.qs = q.sorted; p.q_begin = 0; p.q_count = (uint32_t)q.n;
  p.gts = g.graw; p.ts = t.sorted; p.start = g.start; p.dir = g.dir; p.recs = g.recs; p.dt_shift = g.dt_shift;
  for (int k = 0; k < 3; ++k) { p.dtdim[k] = g.dtdim[k]; p.lo[k] = g.lo[k]; p.dim[k] = g.dim[k]; }
  p.dt = g.dt; p.dt_max = g.dt_steps; p.inv_h = g.inv_h; p.h = g.h;
  p.tinv = t.order ? t.order->inv : nullptr;
  p.stretch = std::nextafterf(1.000001f, INFINITY);      // (the grid lies in the frame the queries are in: the identity map)
  p.nt = (uint32_t)t.n;
  p.keys = keys;
  return p;
}

// posed copies: coordinates in grid order (and the grid-position -> Hilbert-position map, once per ordering)
int grid_coords_prepare(Ctx *c, Cloud *cl, bool *ok)
{
  *ok = false;
  if (!cl || !cl->grid || !cl->pose_known || cl->n == 0 || cl->grid->n != cl->n) return MVR_OK;
  if (int rc = ensure(c, cl->gsorted, cl->gsorted_cap, cl->n)) return rc;
  if (cl->grid->ready && !cl->grid->ready_waited) {      // built on the side stream: this is its first use on the main one
    MVR_HIP_TRY(c, hipStreamWaitEvent(c->stream, cl->grid->ready, 0));
    cl->grid->ready_waited = true;
  }
  if (cl->order && cl->grid->built_for != cl->order.get()) {
    // (queued: flush_g2h() launches the maps of all the grids a batch asked for at once, ahead of the batch's own launch)
    // (the job keeps grid and ordering alive: a call that fails between here and the flush leaves it for the next flush)
    bool queued = false;
    for (const Ctx::G2HJob &j : c->g2h_todo) queued = queued || (j.grid == cl->grid && j.order == cl->order);
    if (!queued) c->g2h_todo.push_back(Ctx::G2HJob{cl->grid, cl->order, cl->n});
  }
  *ok = true;
  return MVR_OK;
}

int flush_g2h(Ctx *c)
{
  for (size_t base = 0; base < c->g2h_todo.size(); base += kBatchClouds) {
    G2HBatch b;
    const int m = (int)std::min<size_t>(kBatchClouds, c->g2h_todo.size() - base);
    size_t nmax = 0;
    for (int k = 0; k < kBatchClouds; ++k) {
      if (k < m) {
        const Ctx::G2HJob &j = c->g2h_todo[base + (size_t)k];
        b.gperm[k] = j.grid->gperm; b.inv[k] = j.order->inv; b.g2h[k] = j.grid->g2h; b.h2g[k] = j.grid->h2g; b.n[k] = j.n; nmax = std::max(nmax, j.n);
        j.grid->built_for = j.order.get();
      }
      else { b.gperm[k] = b.inv[k] = nullptr; b.g2h[k] = b.h2g[k] = nullptr; b.n[k] = 0; }
    }
    if (nmax) hipLaunchKernelGGL(g2h_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)m), dim3(256), 0, c->stream, b);
  }
  c->g2h_todo.clear();
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int refresh_grid_coords_batch(Ctx *c, Cloud *const *posed, int count)
{
  for (int base = 0; base < count; base += kBatchClouds) {
    GridPoseBatch b;
    const int m = std::min(kBatchClouds, count - base);
    size_t nmax = 0;
    int used = 0;
    for (int k = 0; k < m; ++k) {
      Cloud *cl = posed[base + k];
      if (!cl || cl->gcoords_valid) continue;
      bool ok = false;
      if (int rc = grid_coords_prepare(c, cl, &ok)) return rc;
      if (!ok) continue;
      b.graw[used] = cl->grid->graw; b.out[used] = cl->gsorted; b.n[used] = cl->n;
      std::memcpy(b.T[used].m, cl->pose, sizeof b.T[used].m);
      b.Tp[used] = (c->pose_from_table && cl->pose_dev) ? &cl->pose_dev->T : nullptr;
      if (b.Tp[used]) { if (int rc = ensure_pose_table(c)) return rc; }
      nmax = std::max(nmax, cl->n);
      cl->gcoords_valid = true;
      ++used;
    }
    if (int rc = flush_g2h(c)) return rc;
    if (!used) continue;
    for (int k = used; k < kBatchClouds; ++k) { b.graw[k] = nullptr; b.out[k] = nullptr; b.n[k] = 0; b.Tp[k] = nullptr; }
    ProfScope ps(c, MVR_K_XFORM, 32.0 * (double)nmax * used);
    hipLaunchKernelGGL(grid_pose_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)used), dim3(256), 0, c->stream, b);
    MVR_HIP_TRY(c, hipGetLastError());
  }
  return MVR_OK;
}

GridPair make_grid_pair(const Cloud &q, size_t q_begin, size_t q_count, const Cloud &t, nnkey_t *keys)
{
  GridPair p;
  const CellGrid &g = *t.grid;
  p.qs = q.sorted; p.q_begin = (uint32_t)q_begin; p.q_count = (uint32_t)q_count;
  p.gts = t.gsorted; p.ts = t.sorted; p.start = g.start; p.dir = g.dir; p.recs = g.recs; p.dt_shift = g.dt_shift; for (int k = 0; k < 3; ++k) p.dtdim[k] = g.dtdim[k]; p.dt = g.dt; p.g2h = g.g2h; p.h2g = g.h2g; p.tinv = t.order ? t.order->inv : nullptr; p.dt_max = g.dt_steps;
  p.stretch = std::nextafterf((float)(t.pose_stretch * (1.0 + 1e-6)), INFINITY);      // (the ball is in posed space, the cells in the canonical frame)
  for (int k = 0; k < 3; ++k) { p.lo[k] = g.lo[k]; p.dim[k] = g.dim[k]; }
  p.inv_h = g.inv_h; p.h = g.h;
  // inverse of the pose's affine map x -> A x + t (column-major 4 x 4, column-vector; A is a rotation up to float rounding:
  // the true inverse of the 3 x 3 by cofactors, in double): r = A^-1 p - A^-1 t
  const double *T = t.pose;
  const double A[3][3] = {{T[0], T[4], T[8]}, {T[1], T[5], T[9]}, {T[2], T[6], T[10]}};
  const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                     A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
  const double id = 1.0 / det;
  const double I[3][3] = {{(A[1][1] * A[2][2] - A[1][2] * A[2][1]) * id, (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * id, (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * id},
                          {(A[1][2] * A[2][0] - A[1][0] * A[2][2]) * id, (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * id, (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * id},
                          {(A[1][0] * A[2][1] - A[1][1] * A[2][0]) * id, (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * id, (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * id}};
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k) p.minv[4 * r + k] = I[r][k];
    p.minv[4 * r + 3] = -(I[r][0] * T[12] + I[r][1] * T[13] + I[r][2] * T[14]);
  }
  p.nt = (uint32_t)t.n;
  p.keys = keys;
  p.pose_dev = t.pose_dev;      // (a pass enqueued ahead of its poses: minv / stretch above are placeholders, the kernels read the device record)
  return p;
}

int launch_nn_grid_batch(Ctx *c, const GridPair *pairs, int n_pairs, float cap2, bool fma)
{
  for (int base = 0; base < n_pairs; base += kGridBatchPairs) {
    GridBatch batch;
    XcdMap map;
    const int m = std::min(kGridBatchPairs, n_pairs - base);
    const int lanes = (c->grid_lanes == 1 || c->grid_lanes == 2 || c->grid_lanes == 8) ? c->grid_lanes : 4;
    const size_t per_block = (size_t)(kGridThreads / lanes);
    size_t total = 0;
    for (int k = 0; k < kBatchPairs; ++k) map.sets[k] = 0;
    for (int k = 0; k < kGridBatchPairs; ++k) {
      batch.p[k] = k < m ? pairs[base + k] : GridPair{};
      if (k < m && batch.p[k].nt == 0) batch.p[k].q_count = 0;
      // what the walk takes for granted: four points or more (it reads four at a time), and the ordering's inverse where a match's position is asked for
      if (k < m && batch.p[k].q_count != 0 && (batch.p[k].nt < 4 || ((batch.p[k].key_by_pos || batch.p[k].mark) && !batch.p[k].tinv)))
        return set_error(c, MVR_E_ARG, "nn_grid: a target of fewer than four points, or without its ordering");
      map.sets[k] = k < m ? (uint32_t)(((size_t)batch.p[k].q_count + per_block - 1) / per_block) : 0u;
      total += map.sets[k];
    }
    batch.cap2 = cap2;
    // (a lone pair's launch -- the reverse search of a sequential align, a few thousand waves -- leaves the chip room for the
    // wave-per-query launch behind it: its wide balls go there earlier than those of a fused pass's launch, which fills the chip)
    batch.light_rows = n_pairs == 1 ? std::min(c->grid_light_rows, c->grid_light_rows_lone) : c->grid_light_rows;
    batch.cluster = c->grid_cluster;
    batch.probe = c->grid_probe;
    batch.probe_rows = std::min(c->grid_probe_rows, batch.light_rows);
    batch.delta_max = 1.0e-3f * (float)c->seed_delta_um;
    if (total == 0) continue;
    map.n_pairs = (uint32_t)m;
    unsigned grid_blocks = 0;
    xcd_map_plan(map, c->cull_slices, &grid_blocks);
    const bool per_launch = c->prof && !c->prof_totals;
    if (per_launch) MVR_HIP_TRY(c, hipMemsetAsync(c->evals, 0, kEvalRegion * sizeof(unsigned long long), c->stream));
    ProfScope ps(c, MVR_K_NN_GRID, per_launch ? c->evals : nullptr, (int)(kEvalRegion * sizeof(unsigned long long)), 1.0, 0.0, per_launch ? kEvalShards : 0);
#define MVR_GRID_LAUNCH(F, GG) hipLaunchKernelGGL((nn_grid_kernel<F, GG, false>), dim3(grid_blocks), dim3(kGridThreads), 0, c->stream, batch, map, c->evals, (unsigned long long *)nullptr)
#define MVR_GRID_LAUNCH_STAGED(F) hipLaunchKernelGGL((nn_grid_kernel<F, 1, true>), dim3(grid_blocks), dim3(kGridThreads), 0, c->stream, batch, map, c->evals, c->stage_stat)
    // (grid_stage 1: the staged walk for launches over a scan's OWN query order only -- the forward searches; a launch over a compacted
    // list, the reverse searches, has queries 2.4 x sparser: with a region per WAVE its boxes did not fit, measured equal or slower.
    // grid_stage 2, the default since a block's two waves stage one region of 448 points together: both -- the step -1.4 %)
    bool listed = false;
    for (int k = 0; k < m; ++k) listed = listed || (batch.p[k].q_count != 0 && batch.p[k].qlist != nullptr);
    // (a LONE pair's launch over a list -- the reverse search of a sequential align, 3 waves per SIMD -- stays plain: staged it measured
    // 1.5 % of an align slower; grid_stage_lone = 2 stages it all the same)
    const bool stage = lanes == 1 && c->grid_stage && (c->grid_stage == 2 || !listed) && (n_pairs > 1 || (c->grid_stage_lone && (!listed || c->grid_stage_lone == 2)));
    if (stage) { if (fma) MVR_GRID_LAUNCH_STAGED(true); else MVR_GRID_LAUNCH_STAGED(false); }
    else
    switch (lanes) {
      case 1: if (fma) MVR_GRID_LAUNCH(true, 1); else MVR_GRID_LAUNCH(false, 1); break;
      case 2: if (fma) MVR_GRID_LAUNCH(true, 2); else MVR_GRID_LAUNCH(false, 2); break;
      case 8: if (fma) MVR_GRID_LAUNCH(true, 8); else MVR_GRID_LAUNCH(false, 8); break;
      default: if (fma) MVR_GRID_LAUNCH(true, 4); else MVR_GRID_LAUNCH(false, 4); break;
    }
#undef MVR_GRID_LAUNCH
#undef MVR_GRID_LAUNCH_STAGED
    MVR_HIP_TRY(c, hipGetLastError());
  }
  return MVR_OK;
}

template <bool FMA>
__global__ void __launch_bounds__(64 * kWideWaves) nn_grid_wide_kernel(GridBatch batch, unsigned long long *__restrict__ evals)
{
  nn_grid_wide_body<FMA>(batch, evals, blockIdx.x, gridDim.x);
}

// ---- the listed query sets (flagged queries: the rim of the overlap, matches that lie far): ONE BLOCK per set.  The
// culled kernel answered these by walking its box hierarchy -- some fifty dependent loads per wave, 40 us for a set
// whatever the number of waves, a latency chain.  With a grid the candidates are an address computation here too: the
// union of the flagged queries' balls in cells -> its rows of cells (one range each) -> every point of those ranges
// into LDS once, by all 256 threads -> every query against every staged point, the four waves sharing out the points
// (lane = query; a broadcast LDS read feeds 64 evaluations).  Same bounds (inclusive), same distances, same
// (d2, original index) minimum: the keys the culled kernel would have written.
constexpr int kSetRows = 512;        // rows of cells staged per round
#ifndef MVR_SET_POINTS
#define MVR_SET_POINTS (MVR_TAIL_WAVES == 8 ? 1024 : 768)
#endif
constexpr int kSetPoints = MVR_SET_POINTS;      // points staged per round (768 with four waves: the shared arrays stay under 20 KB, eight blocks per CU)
template <bool FMA>
__device__ __forceinline__ void nn_grid_set_body(const GridBatch &batch, unsigned long long *__restrict__ evals, const uint32_t bx, const uint32_t nbx)
{
  const GridPair &a = batch.p[blockIdx.y];
  if (!a.cull_count) return;
  __shared__ float4 lq[64];                       // query (x, y, z, bound as bits)
  __shared__ int ubox[8];
  constexpr int W = kWideWaves, kThreads = 64 * W, kRowsPer = (kSetRows + kThreads - 1) / kThreads;
  __shared__ uint32_t rs[kSetRows], roff[kSetRows + 1], wtot[W];
  __shared__ float4 lpts[kSetPoints];
  __shared__ unsigned long long lbest[W][64];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const float cap2 = batch.cap2;
  const uint32_t n_sets = *a.cull_count;
  unsigned long long n_eval = 0;
  for (uint32_t si = bx; si < n_sets; si += nbx) {
    const uint32_t set = a.cull_sets[si];
    // ---- the set's queries (wave 0): bound, ball in cells, union of the flagged ones
    bool valid = false, grant = false;
    uint32_t qpos = 0;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wv == 0) {
      const uint32_t pos = set * 64u + (uint32_t)lane;
      int c0[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, c1[3] = {-1, -1, -1};
      float bound = cap2;
      if (pos < a.q_count) {
        qpos = a.q_begin + pos;
        valid = a.heavy[qpos] != 0;
      }
      if (valid) {
        q = a.qs[qpos];
        bool bounded = false;
        if (a.qbound) { const uint32_t v = a.qbound[qpos]; if (v <= __float_as_uint(bound)) { bound = __uint_as_float(v); bounded = true; } }
        if (a.seed_from_keys) {
          const uint32_t prev = (uint32_t)a.keys[qpos];
          if (prev < a.nt) { const float d = gdist2<FMA>(a.gts[a.h2g[prev]], q.x, q.y, q.z); if (d <= bound) { bound = d; bounded = true; } }
        }
        // rim certificates: a query without any bound is searched a margin BEYOND the cap -- nothing within that earns it a
        // certificate (the walk then leaves it alone while the clouds move by less); what it finds between the cap and the margin
        // is no match (the key below takes matches within the cap only) and no certificate
        grant = a.cert != nullptr && !bounded && batch.cert_margin > 0.f;
        if (grant) { const float cm = __builtin_amdgcn_sqrtf(cap2) * 1.000001f + batch.cert_margin; bound = cm * cm * 1.000001f; }
        const Ball ball = grid_ball(a, q, bound);
        const float rad = ball.rad, rx = ball.rx, ry = ball.ry, rz = ball.rz;
        c0[0] = cell_of(rx - rad, a.lo[0], a.inv_h, a.dim[0]); c1[0] = cell_of(rx + rad, a.lo[0], a.inv_h, a.dim[0]);
        c0[1] = cell_of(ry - rad, a.lo[1], a.inv_h, a.dim[1]); c1[1] = cell_of(ry + rad, a.lo[1], a.inv_h, a.dim[1]);
        c0[2] = cell_of(rz - rad, a.lo[2], a.inv_h, a.dim[2]); c1[2] = cell_of(rz + rad, a.lo[2], a.inv_h, a.dim[2]);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) { c0[k] = min(c0[k], __shfl_xor(c0[k], o, 64)); c1[k] = max(c1[k], __shfl_xor(c1[k], o, 64)); }
      lq[lane] = make_float4(q.x, q.y, q.z, bound);
      if (lane == 0) { for (int k = 0; k < 3; ++k) { ubox[k] = c0[k]; ubox[3 + k] = c1[k]; } }
    }
    __syncthreads();
    const int X0 = ubox[0], Y0 = ubox[1], Z0 = ubox[2], X1 = ubox[3], Y1 = ubox[4], Z1 = ubox[5];
    const float4 mq = lq[lane];                     // this lane's query in every wave
    unsigned long long best = ((unsigned long long)__float_as_uint(mq.w) << 32) | 0xFFFFFFFFull;      // a candidate AT the bound still beats it
    if (X1 >= X0) {                                 // (no flagged query: nothing to do -- cannot happen for a listed set)
      const int ny = Y1 - Y0 + 1, rows = ny * (Z1 - Z0 + 1);
      for (int r0 = 0; r0 < rows; r0 += kSetRows) {
        const int nr = min(kSetRows, rows - r0);
        // ---- the rows' ranges, and where each begins in the staged list (block-wide exclusive scan of the lengths)
        uint32_t mine = 0;
        for (int r = t * kRowsPer; r < min(nr, t * kRowsPer + kRowsPer); ++r) {
          const int rr = r0 + r;
          const uint32_t row = (uint32_t)(((Z0 + rr / ny) * a.dim[1] + (Y0 + rr % ny)) * a.dim[0]);
          const uint32_t s = cell_start_of(a, row + (uint32_t)X0), e = cell_start_of(a, row + (uint32_t)X1 + 1u);
          rs[r] = s; roff[r] = e - s;
          mine += e - s;
        }
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        uint32_t before = incl - mine;
        for (int w = 0; w < wv; ++w) before += wtot[w];
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) total += wtot[w];
        for (int r = t * kRowsPer; r < min(nr, t * kRowsPer + kRowsPer); ++r) { const uint32_t len = roff[r]; roff[r] = before; before += len; }
        if (t == 0) roff[nr] = total;
        __syncthreads();
        // ---- stage the points of those ranges, kSetPoints at a time, and evaluate: wave w takes every fourth point
        for (uint32_t base = 0; base < total; base += kSetPoints) {
          const uint32_t cnt = min((uint32_t)kSetPoints, total - base);
          // (one staged point per thread and round, its row found by bisection of the offsets: independent loads, all in
          // flight at once -- a thread copying its rows point after point waited for every one of them)
#pragma unroll
          for (int u = 0; u < kSetPoints / kThreads; ++u) {
            const uint32_t j = base + (uint32_t)(u * kThreads + t);
            if (j - base < cnt) {
              int lo = 0, hi = nr - 1;
              while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (roff[mid] <= j) lo = mid; else hi = mid - 1; }
              lpts[j - base] = a.gts[rs[lo] + (j - roff[lo])];
            }
          }
          __syncthreads();
          uint32_t j = (uint32_t)wv;
          for (; j + (uint32_t)W < cnt; j += 2u * W) {            // two points in flight
            const float4 p0 = lpts[j], p1 = lpts[j + (uint32_t)W];
            const unsigned long long k0 = ((unsigned long long)__float_as_uint(gdist2<FMA>(p0, mq.x, mq.y, mq.z)) << 32) | __float_as_uint(p0.w);
            const unsigned long long k1 = ((unsigned long long)__float_as_uint(gdist2<FMA>(p1, mq.x, mq.y, mq.z)) << 32) | __float_as_uint(p1.w);
            const unsigned long long k = k0 < k1 ? k0 : k1;
            best = k < best ? k : best;
          }
          if (j < cnt) {
            const float4 p = lpts[j];
            const unsigned long long key = ((unsigned long long)__float_as_uint(gdist2<FMA>(p, mq.x, mq.y, mq.z)) << 32) | __float_as_uint(p.w);
            best = key < best ? key : best;
          }
          n_eval += cnt;
          __syncthreads();
        }
      }
    }
    // ---- the four waves' answers meet; wave 0 writes the flagged queries' keys (and the start bounds of the reverse searches)
    lbest[wv][lane] = best;
    __syncthreads();
    if (wv == 0 && valid) {
      unsigned long long m = lbest[0][lane];
#pragma unroll
      for (int w = 1; w < W; ++w) { const unsigned long long v = lbest[w][lane]; m = v < m ? v : m; }
      const uint32_t bi = (uint32_t)m;
      const bool found = bi != kNone && __uint_as_float((uint32_t)(m >> 32)) <= cap2;
      const uint32_t ord = a.key_by_pos ? qpos : __float_as_uint(q.w);
      uint32_t low = bi;
      if (found && (a.key_by_pos || a.mark)) {
        const uint32_t hp = a.tinv[bi];
        if (a.key_by_pos) low = hp;
        if (a.mark) __hip_atomic_store(&a.mark[hp], (uint32_t)(m >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      a.keys[ord] = found ? ((m & 0xFFFFFFFF00000000ull) | low) : kKeyInit;
      if (a.cert) a.cert[qpos] = (grant && bi == kNone) ? batch.cert_margin : 0.f;
    }
    __syncthreads();                                // lq / ubox / lbest are reused by the next set
  }
  if (evals && lane == 0 && n_eval) {
    unsigned long long *sh = evals + (size_t)((bx * (uint32_t)W + (uint32_t)wv) & (kEvalShards - 1)) * kEvalStride;
    atomicAdd(sh, n_eval * (64ull / W));      // every staged point is evaluated by the 64 lanes of ONE of the W waves
    atomicAdd(sh + kEvalRegion, n_eval * (64ull / W));
  }
}

template <bool FMA>
__global__ void __launch_bounds__(64 * kWideWaves, 2048 / (64 * kWideWaves)) nn_grid_set_kernel(GridBatch batch, unsigned long long *__restrict__ evals)
{
  nn_grid_set_body<FMA>(batch, evals, blockIdx.x, gridDim.x);
}

// both stragglers' launches in one: the first `wide_blocks` blocks of a pair give the wide bounded queries a wave each,
// the others take the listed query sets -- neither waits for the other (two launches of 8 and 24 us ran one after the other)
static_assert(kSetPoints % (64 * kWideWaves) == 0 && (kWideWaves == 4 || kWideWaves == 8), "the two bodies share a block shape");
template <bool FMA>
__global__ void __launch_bounds__(64 * kWideWaves, 2048 / (64 * kWideWaves)) nn_grid_tail_kernel(GridBatch batch, unsigned long long *__restrict__ evals, uint32_t wide_blocks)
{
  if (blockIdx.x < wide_blocks) nn_grid_wide_body<FMA>(batch, evals, blockIdx.x, wide_blocks);
  else nn_grid_set_body<FMA>(batch, evals, blockIdx.x - wide_blocks, gridDim.x - wide_blocks);
}

// wide bounded queries and listed sets of the same pairs in one launch
int launch_nn_grid_tail_batch(Ctx *c, const GridPair *pairs, int n_pairs, float cap2, bool fma)
{
  for (int base = 0; base < n_pairs; base += kGridBatchPairs) {
    GridBatch batch;
    const int m = std::min(kGridBatchPairs, n_pairs - base);
    bool any = false;
    for (int k = 0; k < kGridBatchPairs; ++k) {
      batch.p[k] = k < m ? pairs[base + k] : GridPair{};
      GridPair &p = batch.p[k];
      const bool live = k < m && p.nt != 0 && p.q_count != 0;
      if (!live || !p.wide_list) p.wide_count = nullptr;
      if (!live || !p.cull_sets || !p.heavy || !p.tinv || p.qlist) p.cull_count = nullptr;
      any = any || p.wide_count != nullptr || p.cull_count != nullptr;
    }
    if (!any) continue;
    batch.cap2 = cap2;
    batch.light_rows = c->grid_light_rows;
    batch.cert_margin = 1.0e-3f * (float)c->rim_cert_um;
    const unsigned wide_blocks = (unsigned)std::max(1, c->n_cu * c->grid_wide_waves / (std::max(1, m) * kWideWaves * 4));
    const unsigned set_blocks = (unsigned)std::max(1, c->n_cu * (2048 / (64 * kWideWaves)) / std::max(1, m));
    const bool per_launch = c->prof && !c->prof_totals;
    if (per_launch) MVR_HIP_TRY(c, hipMemsetAsync(c->evals, 0, kEvalRegion * sizeof(unsigned long long), c->stream));
    ProfScope ps(c, MVR_K_NN_WIDE, per_launch ? c->evals : nullptr, (int)(kEvalRegion * sizeof(unsigned long long)), 1.0, 0.0, per_launch ? kEvalShards : 0);
    if (fma) hipLaunchKernelGGL((nn_grid_tail_kernel<true>), dim3(wide_blocks + set_blocks, (unsigned)m), dim3(64 * kWideWaves), 0, c->stream, batch, c->evals, wide_blocks);
    else hipLaunchKernelGGL((nn_grid_tail_kernel<false>), dim3(wide_blocks + set_blocks, (unsigned)m), dim3(64 * kWideWaves), 0, c->stream, batch, c->evals, wide_blocks);
    MVR_HIP_TRY(c, hipGetLastError());
  }
  return MVR_OK;
}

// ---- a COMPOSITE target, searched part by part: the growing model of the sequential mode (registrator.cpp:563-577) is a
// concatenation of posed scans that never move once appended; each keeps its raw scan's pose-invariant grid (GridPart).
// One thread per source query walks the parts one after the other with a RUNNING bound: the first part that has points
// next to the query gives a bound a few tenths of a millimetre wide (a probe of the 27 cells around the query), and every
// later part is either ruled out by one byte of its distance map or walked over a handful of cells.  The reference's
// kd-tree over the whole merged cloud (rebuilt at every align, registrator.cpp:569) becomes K small address computations.
// Candidates are evaluated on the posed coordinates with the float formula of every other search; the minimum is over
// (d2, COMPOSITE original index), so the result equals a search of the merged cloud bit for bit.  A query whose ball
// cannot be bounded here (no point next to it in any part, yet possibly one within the cap) is flagged for the culled
// kernel, which searches the composite index for it.
namespace {

// G lanes share a query: lane `sub` takes parts sub, sub + G, ...  (One lane per query is a chain of some forty dependent
// loads at 200k queries -- 3 waves per SIMD, nothing to hide the latency behind: 442 us at ten parts.)  Two phases:
//   A  every lane looks at its parts' distance-map byte; the first of its parts that has points NEXT to the query (d <= 1)
//      is probed -- the 27 cells around the query -- and the group's best find becomes the bound of ALL its lanes;
//   B  every lane walks the ball of that bound in each of its parts that the distance map does not rule out.
// A query without any part next to it has no bound: if some part may still hold a point within the cap it is flagged for
// the culled kernel (the rim of the coverage: few, and clustered), else it has no neighbour.
template <bool FMA, int G>
__global__ void __launch_bounds__(256) nn_parts_kernel(const float4 *__restrict__ qs, uint32_t nq, const PartDesc *__restrict__ parts, int n_parts,
                                                       float cap2, int max_rows, nnkey_t *__restrict__ keys, uint8_t *__restrict__ heavy,
                                                       const uint32_t *__restrict__ qbound, unsigned long long *__restrict__ evals)
{
  const uint32_t pos = (blockIdx.x * 256u + threadIdx.x) / G;
  const int sub = (int)(threadIdx.x % G);
  unsigned long long n_eval = 0;
  const bool live = pos < nq;
  const float4 q = live ? qs[pos] : make_float4(0.f, 0.f, 0.f, 0.f);
  const double qx = q.x, qy = q.y, qz = q.z;
  const float qn1 = fabsf(q.x) + fabsf(q.y) + fabsf(q.z);
  float bd = cap2;                   // nothing beyond it is wanted; AT it, it is (inclusive)
  uint32_t bi = kNone;
  bool in_reach = false, too_wide = false;
  // a query that comes WITH a bound (the distance, now, of the point it matched when this scan was last aligned: some target
  // point lies within it) needs no probe: phase B at once
  bool seeded = false;
  if (live && qbound) { const uint32_t v = qbound[pos]; if (v <= __float_as_uint(cap2)) { bd = __uint_as_float(v); seeded = true; } }
  auto map_into = [&](const PartDesc &a, float &rx, float &ry, float &rz, float &slack) {
    rx = (float)(((a.minv[0] * qx + a.minv[1] * qy) + a.minv[2] * qz) + a.minv[3]);
    ry = (float)(((a.minv[4] * qx + a.minv[5] * qy) + a.minv[6] * qz) + a.minv[7]);
    rz = (float)(((a.minv[8] * qx + a.minv[9] * qy) + a.minv[10] * qz) + a.minv[11]);
    slack = 2.0e-6f * (fabsf(rx) + fabsf(ry) + fabsf(rz));            // (see grid_ball)
  };
  auto walk = [&](const PartDesc &a, int x0, int x1, int y0, int y1, int z0, int z1) {
    // rows of cells in y-major order; the next row's range is requested while this row's points are evaluated
    const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
    uint32_t row = (uint32_t)((z0 * a.dim[1] + y0) * a.dim[0]);
    const uint32_t row_step = (uint32_t)a.dim[0], z_step = (uint32_t)((a.dim[1] - ny) * a.dim[0]);
    uint32_t s = cell_start_of(a, row + (uint32_t)x0), e = cell_start_of(a, row + (uint32_t)x1 + 1u);
    int yy = 0;
    for (int it = 0; it < nrows; ++it) {
        uint32_t s2 = 0, e2 = 0;
        if (it + 1 < nrows) {
          row += row_step;
          if (++yy == ny) { yy = 0; row += z_step; }
          s2 = cell_start_of(a, row + (uint32_t)x0); e2 = cell_start_of(a, row + (uint32_t)x1 + 1u);
        }
        n_eval += e - s;
        for (uint32_t i = s; i < e; i += 4) {
          const uint32_t last = e - 1u;
          const uint32_t i1 = min(i + 1u, last), i2 = min(i + 2u, last), i3 = min(i + 3u, last);
          const float4 t0 = a.gts[i], t1 = a.gts[i1], t2 = a.gts[i2], t3 = a.gts[i3];
          const float d0 = gdist2<FMA>(t0, q.x, q.y, q.z), d1 = gdist2<FMA>(t1, q.x, q.y, q.z);
          const float d2 = gdist2<FMA>(t2, q.x, q.y, q.z), d3 = gdist2<FMA>(t3, q.x, q.y, q.z);
          const uint32_t o0 = __float_as_uint(t0.w), o1 = __float_as_uint(t1.w), o2 = __float_as_uint(t2.w), o3 = __float_as_uint(t3.w);
          if (d0 < bd || (d0 == bd && o0 < bi)) { bd = d0; bi = o0; }
          if (d1 < bd || (d1 == bd && o1 < bi)) { bd = d1; bi = o1; }
          if (d2 < bd || (d2 == bd && o2 < bi)) { bd = d2; bi = o2; }
          if (d3 < bd || (d3 == bd && o3 < bi)) { bd = d3; bi = o3; }
        }
        s = s2; e = e2;
    }
  };
  auto share = [&]() {             // the group's smallest (d2, index)
#pragma unroll
    for (int o = 1; o < G; o <<= 1) {
      const float od = __shfl_xor(bd, o, 64);
      const uint32_t oi = (uint32_t)__shfl_xor((int)bi, o, 64);
      if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
  };
  // ---- A: a bound.  Every lane reads its parts' distance-map bytes; ONE part of the group that has points next to the
  // query (d = 0 before d = 1) is probed -- the 27 cells around the query -- by the lane that owns it; should that probe
  // find nothing within the cap, the next candidate is tried.
  uint32_t cand_mask = 0;            // bit j: this lane's j-th part (k = sub + j G) has d <= 1; bit 16 + j: d == 0
  if (live && !seeded) {
    int j = 0;
    for (int k = sub; k < n_parts; k += G, ++j) {
      const PartDesc &a = parts[k];
      float rx, ry, rz, slack;
      map_into(a, rx, ry, rz, slack);
      const float rad = (sqrtf(cap2) * 1.00001f + (1.0e-3f + 4.0e-6f * qn1)) * a.stretch + slack;
      const int cx = cell_of(rx, a.lo[0], a.inv_h, a.dim[0]), cy = cell_of(ry, a.lo[1], a.inv_h, a.dim[1]), cz = cell_of(rz, a.lo[2], a.inv_h, a.dim[2]);
      const uint32_t d = dt_of(a, cx, cy, cz);
      // the query's (clamped) cell is d cells from the nearest occupied one: every point of this part is at least
      // (d - 1) cell edges away along some axis
      if (dt_least_of(a, d) > rad) continue;
      in_reach = true;                                   // this part may hold a point within the cap
      if (d <= 1u && j < 16) cand_mask |= (1u << j) | (d == 0u ? (1u << (16 + j)) : 0u);
    }
  }
  for (int round = 0; round < 2 * 16 * G; ++round) {       // (ends as soon as a probe succeeds or the candidates run out)
    // the group's next candidate: the occupied-cell ones first, lowest lane, lowest part
    const uint32_t mine = (cand_mask >> 16) ? 2u : (cand_mask & 0xFFFFu) ? 1u : 0u;
    uint32_t best_rank = mine, owner = (uint32_t)sub;
#pragma unroll
    for (int o = 1; o < G; o <<= 1) {
      const uint32_t orank = (uint32_t)__shfl_xor((int)best_rank, o, 64), oown = (uint32_t)__shfl_xor((int)owner, o, 64);
      if (orank > best_rank || (orank == best_rank && oown < owner)) { best_rank = orank; owner = oown; }
    }
    if (best_rank == 0u) break;                            // (uniform over the group)
    if ((uint32_t)sub == owner) {
      const uint32_t pick = (cand_mask >> 16) ? (cand_mask >> 16) : (cand_mask & 0xFFFFu);
      const int j = __ffs((int)pick) - 1;
      cand_mask &= ~((1u << j) | (1u << (16 + j)));
      const PartDesc &a = parts[sub + j * G];
      float rx, ry, rz, slack;
      map_into(a, rx, ry, rz, slack);
      const int cx = cell_of(rx, a.lo[0], a.inv_h, a.dim[0]), cy = cell_of(ry, a.lo[1], a.inv_h, a.dim[1]), cz = cell_of(rz, a.lo[2], a.inv_h, a.dim[2]);
      walk(a, max(cx - 1, 0), min(cx + 1, a.dim[0] - 1), max(cy - 1, 0), min(cy + 1, a.dim[1] - 1), max(cz - 1, 0), min(cz + 1, a.dim[2] - 1));
    }
    share();
    if (bi != kNone) break;                                // (uniform: shared)
  }
  const bool have = seeded || bi != kNone;     // (uniform over the group)
  // ---- B: the exact ball of the bound, in every part that can reach into it
  if (live && have) {
    for (int k = sub; k < n_parts; k += G) {
      const PartDesc &a = parts[k];
      float rx, ry, rz, slack;
      map_into(a, rx, ry, rz, slack);
      const float rad = (sqrtf(bd) * 1.00001f + (1.0e-3f + 4.0e-6f * qn1)) * a.stretch + slack;
      const int cx = cell_of(rx, a.lo[0], a.inv_h, a.dim[0]), cy = cell_of(ry, a.lo[1], a.inv_h, a.dim[1]), cz = cell_of(rz, a.lo[2], a.inv_h, a.dim[2]);
      const uint32_t d = dt_of(a, cx, cy, cz);
      if (dt_least_of(a, d) > rad) continue;
      const int x0 = cell_of(rx - rad, a.lo[0], a.inv_h, a.dim[0]), x1 = cell_of(rx + rad, a.lo[0], a.inv_h, a.dim[0]);
      const int y0 = cell_of(ry - rad, a.lo[1], a.inv_h, a.dim[1]), y1 = cell_of(ry + rad, a.lo[1], a.inv_h, a.dim[1]);
      const int z0 = cell_of(rz - rad, a.lo[2], a.inv_h, a.dim[2]), z1 = cell_of(rz + rad, a.lo[2], a.inv_h, a.dim[2]);
      if ((y1 - y0 + 1) * (z1 - z0 + 1) > max_rows) { too_wide = true; continue; }
      walk(a, x0, x1, y0, y1, z0, z1);
    }
    share();
  }
  // a lane's verdict on its own parts -> the group's
  unsigned flag = (too_wide || (!have && in_reach)) ? 1u : 0u;
#pragma unroll
  for (int o = 1; o < G; o <<= 1) flag |= (unsigned)__shfl_xor((int)flag, o, 64);
  if (live && sub == 0) {
    const bool found = bi != kNone && bd <= cap2;
    keys[__float_as_uint(q.w)] = found ? (((nnkey_t)__float_as_uint(bd) << 32) | bi) : kKeyInit;
    heavy[pos] = flag ? 1 : 0;
  }
  if (evals) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_eval += __shfl_xor(n_eval, o, 64);
    if ((threadIdx.x & 63) == 0 && n_eval) {
      unsigned long long *sh = evals + (size_t)((blockIdx.x * 4u + (threadIdx.x >> 6)) & (kEvalShards - 1)) * kEvalStride;
      atomicAdd(sh, n_eval);
      atomicAdd(sh + kEvalRegion, n_eval);
    }
  }
}

// a part's coordinates in its grid's order, by the arithmetic that made its points: f32(pose * canonical) as
// transform_f64_kernel, then (kind 2) the f32 matrix as transform_f32_kernel -- the same operations on the same inputs, so
// out[k].xyz == pts[base + gperm[k]].xyz bit for bit; w = bits(base + original index)
struct Mat34fp { float m[12]; };
__global__ void parts_coords_kernel(const float4 *__restrict__ graw, float4 *__restrict__ out, size_t n, Mat44d T, Mat34fp F, int kind, uint32_t base)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const float4 p = graw[k];
  float4 v = pose_point_f64(T, p);
  if (kind == 2) {
    float4 o;
    o.x = ((F.m[0] * v.x + F.m[1] * v.y) + F.m[2] * v.z) + F.m[3];
    o.y = ((F.m[4] * v.x + F.m[5] * v.y) + F.m[6] * v.z) + F.m[7];
    o.z = ((F.m[8] * v.x + F.m[9] * v.y) + F.m[10] * v.z) + F.m[11];
    v = o;
  }
  v.w = __uint_as_float(base + __float_as_uint(p.w));
  out[k] = v;
}

}  // namespace

namespace {
__global__ void merge_flagged_keys_kernel(const float4 *__restrict__ sorted, const uint8_t *__restrict__ flags, const nnkey_t *__restrict__ by_pos, size_t n,
                                          nnkey_t *__restrict__ keys)
{
  const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos < n && flags[pos]) keys[__float_as_uint(sorted[pos].w)] = by_pos[pos];
}
}  // namespace

int launch_merge_flagged_keys(Ctx *c, const float4 *sorted, const uint8_t *flags, const nnkey_t *by_pos, size_t n, nnkey_t *keys)
{
  if (n == 0) return MVR_OK;
  hipLaunchKernelGGL(merge_flagged_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, sorted, flags, by_pos, n, keys);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int fill_part_coords(Ctx *c, Cloud &target, GridPart &part)
{
  if (!part.grid || part.grid->n != part.n || !target.gsorted || part.base + part.n > target.gsorted_cap) return set_error(c, MVR_E_ARG, "fill_part_coords");
  Mat44d T; Mat34fp F;
  std::memcpy(T.m, part.pose, sizeof T.m);
  for (int r = 0; r < 3; ++r) for (int k = 0; k < 4; ++k) F.m[4 * r + k] = part.fin[r + 4 * k];
  ProfScope ps(c, MVR_K_XFORM, 32.0 * (double)part.n);
  hipLaunchKernelGGL(parts_coords_kernel, dim3((unsigned)((part.n + 255) / 256)), dim3(256), 0, c->stream, part.grid->graw, target.gsorted + part.base, part.n, T, F,
                     part.kind, (uint32_t)part.base);
  MVR_HIP_TRY(c, hipGetLastError());
  part.gs_filled = true;
  return MVR_OK;
}

int launch_nn_parts(Ctx *c, const Cloud &q, int count, float cap2, bool fma, nnkey_t *keys, uint8_t *heavy, const uint32_t *qbound)
{
  if (q.n == 0 || count <= 0) return MVR_OK;
  const bool per_launch = c->prof && !c->prof_totals;
  if (per_launch) MVR_HIP_TRY(c, hipMemsetAsync(c->evals, 0, kEvalRegion * sizeof(unsigned long long), c->stream));
  ProfScope ps(c, MVR_K_NN_GRID, per_launch ? c->evals : nullptr, (int)(kEvalRegion * sizeof(unsigned long long)), 1.0, 0.0, per_launch ? kEvalShards : 0);
  // lanes per query (they share out the parts): as many as there are parts, up to eight
  int lanes = count <= 1 ? 1 : count <= 2 ? 2 : count <= 4 ? 4 : 8;
  if (c->parts_lanes == 1 || c->parts_lanes == 2 || c->parts_lanes == 4 || c->parts_lanes == 8) lanes = c->parts_lanes;
  const unsigned blocks = (unsigned)((q.n * (size_t)lanes + 255) / 256);
  const int max_rows = c->parts_max_rows;      // a ball that still spans more rows of cells than this goes to the culled kernel
#define MVR_PARTS_LAUNCH(F, GG) hipLaunchKernelGGL((nn_parts_kernel<F, GG>), dim3(blocks), dim3(256), 0, c->stream, q.sorted, (uint32_t)q.n, c->d_parts, count, cap2, max_rows, keys, heavy, qbound, c->evals)
  switch (lanes) {
    case 1: if (fma) MVR_PARTS_LAUNCH(true, 1); else MVR_PARTS_LAUNCH(false, 1); break;
    case 2: if (fma) MVR_PARTS_LAUNCH(true, 2); else MVR_PARTS_LAUNCH(false, 2); break;
    case 4: if (fma) MVR_PARTS_LAUNCH(true, 4); else MVR_PARTS_LAUNCH(false, 4); break;
    default: if (fma) MVR_PARTS_LAUNCH(true, 8); else MVR_PARTS_LAUNCH(false, 8); break;
  }
#undef MVR_PARTS_LAUNCH
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_nn_grid_sets_batch(Ctx *c, const GridPair *pairs, int n_pairs, float cap2, bool fma)
{
  for (int base = 0; base < n_pairs; base += kGridBatchPairs) {
    GridBatch batch;
    const int m = std::min(kGridBatchPairs, n_pairs - base);
    bool any = false;
    for (int k = 0; k < kGridBatchPairs; ++k) {
      batch.p[k] = k < m ? pairs[base + k] : GridPair{};
      GridPair &p = batch.p[k];
      if (k >= m || p.nt == 0 || p.q_count == 0 || !p.cull_sets || !p.heavy || !p.tinv || p.qlist) p.cull_count = nullptr;
      any = any || p.cull_count != nullptr;
    }
    if (!any) continue;
    batch.cap2 = cap2;
    batch.light_rows = c->grid_light_rows;
    batch.cert_margin = 1.0e-3f * (float)c->rim_cert_um;
    const unsigned blocks_x = (unsigned)std::max(1, c->n_cu * (2048 / (64 * kWideWaves)) / std::max(1, m));      // listed sets are strided over (every block's loop ends at the count)
    const bool per_launch = c->prof && !c->prof_totals;
    if (per_launch) MVR_HIP_TRY(c, hipMemsetAsync(c->evals, 0, kEvalRegion * sizeof(unsigned long long), c->stream));
    ProfScope ps(c, MVR_K_NN_WIDE, per_launch ? c->evals : nullptr, (int)(kEvalRegion * sizeof(unsigned long long)), 1.0, 0.0, per_launch ? kEvalShards : 0);
    if (fma) hipLaunchKernelGGL((nn_grid_set_kernel<true>), dim3(blocks_x, (unsigned)m), dim3(64 * kWideWaves), 0, c->stream, batch, c->evals);
    else hipLaunchKernelGGL((nn_grid_set_kernel<false>), dim3(blocks_x, (unsigned)m), dim3(64 * kWideWaves), 0, c->stream, batch, c->evals);
    MVR_HIP_TRY(c, hipGetLastError());
  }
  return MVR_OK;
}

int launch_nn_grid_wide_batch(Ctx *c, const GridPair *pairs, int n_pairs, float cap2, bool fma)
{
  for (int base = 0; base < n_pairs; base += kGridBatchPairs) {
    GridBatch batch;
    const int m = std::min(kGridBatchPairs, n_pairs - base);
    bool any = false;
    for (int k = 0; k < kGridBatchPairs; ++k) {
      batch.p[k] = k < m ? pairs[base + k] : GridPair{};
      if (k >= m || batch.p[k].nt == 0 || batch.p[k].q_count == 0 || !batch.p[k].wide_list) batch.p[k].wide_count = nullptr;
      any = any || batch.p[k].wide_count != nullptr;
    }
    if (!any) continue;
    batch.cap2 = cap2;
    batch.light_rows = c->grid_light_rows;
    // how many are on the lists is only known on the device: a fixed number of waves per pair strides over its list
    // (every wave's loop ends at the count: the grid always drains)
    const unsigned blocks_x = (unsigned)std::max(1, c->n_cu * c->grid_wide_waves / (std::max(1, m) * kWideWaves));
    const bool per_launch = c->prof && !c->prof_totals;
    if (per_launch) MVR_HIP_TRY(c, hipMemsetAsync(c->evals, 0, kEvalRegion * sizeof(unsigned long long), c->stream));
    ProfScope ps(c, MVR_K_NN_WIDE, per_launch ? c->evals : nullptr, (int)(kEvalRegion * sizeof(unsigned long long)), 1.0, 0.0, per_launch ? kEvalShards : 0);
    if (fma) hipLaunchKernelGGL((nn_grid_wide_kernel<true>), dim3(blocks_x, (unsigned)m), dim3(64 * kWideWaves), 0, c->stream, batch, c->evals);
    else hipLaunchKernelGGL((nn_grid_wide_kernel<false>), dim3(blocks_x, (unsigned)m), dim3(64 * kWideWaves), 0, c->stream, batch, c->evals);
    MVR_HIP_TRY(c, hipGetLastError());
  }
  return MVR_OK;
}

}  // namespace mvr
