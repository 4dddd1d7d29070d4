// csrc/mvr_index.hip -- spatial index for the exact culled nearest-neighbour search (gfx950);
// the search kernel itself lives in mvr_cull.hip.
//
// The GPU-native counterpart of the kd-tree the reference gets from PCL/FLANN (tree_->nearestKSearch
// inside icp.align, registrator.cpp:569, and inside determineReciprocalCorrespondences, :502/:649):
//  * every cloud keeps a Hilbert-ordered copy of its points (w = bits of the original index).  The
//    ORDER belongs to the point set and is shared by all posed copies of a scan, so it is built once
//    (bbox -> 30-bit Hilbert codes -> hipCUB radix sort -> perm / inv) -- from the first POSED copy that is
//    searched, not from the raw scan: cells are compact in the frame the codes were computed in, and a
//    rotation between that frame and the frame of the search inflates every box (+15 % evaluations measured);
//  * three levels of AABBs over that order, always computed from the CURRENT coordinates (the order
//    only affects speed): one box per 64-point cell (cbox, a 128-byte record per tile), one per
//    256-point tile (tlo / thi), one per 64 consecutive tiles (sbox);
//  * after a rigid motion only the sorted copy and the boxes are refreshed: two launches for all the
//    views of a global iteration (refresh_sorted_kernel / super_box_kernel, blockIdx.y = cloud).  A posed
//    copy made by mvr_cloud_transform_batch is refreshed right there, by posing the SOURCE's sorted copy in
//    order (refresh_posed_batch) instead of gathering the posed points through the permutation;
//  * the reciprocal glue of the culled mode: flag matched targets in sorted space -- or record a
//    matching d2 per target, the bound its reverse search starts from (seed_bounds / flag_matched_batch) --
//    and their ordered compaction (hipCUB DeviceSelect, or count / scan / scatter per 256-position chunk for
//    all pairs of a fused pass) into the query list of the reverse search.
// Compiled with -ffp-contract=off.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstring>
#include <vector>

#include "mvr_internal.h"

namespace mvr {

uint32_t *OrderPool::take(size_t bytes, size_t *got)
{
  int best = -1;
  for (int k = 0; k < (int)free.size(); ++k)        // smallest buffer that fits and is not absurdly larger
    if (free[k].bytes >= bytes && free[k].bytes <= 4 * bytes + (1u << 20) && (best < 0 || free[k].bytes < free[best].bytes)) best = k;
  if (best < 0) return nullptr;
  uint32_t *p = free[best].p;
  *got = free[best].bytes;
  free.erase(free.begin() + best);
  return p;
}

void OrderPool::give(uint32_t *p, size_t bytes)
{
  if (!p) return;
  if (closed || free.size() >= 8) { (void)hipFree(p); return; }
  free.push_back(Buf{p, bytes});
}

void OrderPool::close()
{
  closed = true;
  for (Buf &b : free) (void)hipFree(b.p);
  free.clear();
}

Order::~Order()
{
  if (arena) return;                    // (perm / inv live in a block shared by a batch of orderings: the last owner frees it)
  if (pool) { pool->give(perm, perm_bytes); pool->give(inv, inv_bytes); }
  else { if (perm) (void)hipFree(perm); if (inv) (void)hipFree(inv); }
}

namespace {

constexpr int kSub = 32;   // min-tracking sub-tile
#ifndef MVR_SORT_LO_BIT
#define MVR_SORT_LO_BIT 6
#endif
// The sort ignores the lowest 6 bits of the 30-bit Hilbert code (stable: ties keep input order): 8 bits per axis
// order 2M points as well as 10 do (+0.1 % evaluations) and the radix sort of a growing target needs a pass less
// (sequential mode 0.73 -> 0.71 ms/align; 7 bits per axis: the same, 6: 2 % slower).
constexpr int kSortLoBit = MVR_SORT_LO_BIT;

// ------------------------------------------------------------------ index build

__global__ void bbox_partial_kernel(const float4 *__restrict__ p, size_t n, float *__restrict__ part /* [blocks][6] */)
{
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = p[i];
    lo[0] = fminf(lo[0], v.x); lo[1] = fminf(lo[1], v.y); lo[2] = fminf(lo[2], v.z);
    hi[0] = fmaxf(hi[0], v.x); hi[1] = fmaxf(hi[1], v.y); hi[2] = fmaxf(hi[2], v.z);
  }
  __shared__ float s[6][256];
  for (int k = 0; k < 3; ++k) { s[k][threadIdx.x] = lo[k]; s[3 + k][threadIdx.x] = hi[k]; }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int k = 0; k < 3; ++k) {
        s[k][threadIdx.x] = fminf(s[k][threadIdx.x], s[k][threadIdx.x + o]);
        s[3 + k][threadIdx.x] = fmaxf(s[3 + k][threadIdx.x], s[3 + k][threadIdx.x + o]);
      }
    __syncthreads();
  }
  if (threadIdx.x < 6) part[blockIdx.x * 6 + threadIdx.x] = s[threadIdx.x][0];
}

// one wave: lane l folds partial rows l, l+64, ...; then a 64-lane shuffle reduction
__global__ void bbox_final_kernel(const float *__restrict__ part, int blocks, float *__restrict__ bbox)
{
  const int lane = threadIdx.x;
  float v[6] = {3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
  for (int b = lane; b < blocks; b += 64)
    for (int k = 0; k < 6; ++k) v[k] = (k < 3) ? fminf(v[k], part[b * 6 + k]) : fmaxf(v[k], part[b * 6 + k]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    for (int k = 0; k < 6; ++k) {
      const float w = __shfl_xor(v[k], o, 64);
      v[k] = (k < 3) ? fminf(v[k], w) : fmaxf(v[k], w);
    }
#pragma unroll
  for (int k = 0; k < 6; ++k) if (lane == k) bbox[k] = v[k];
}

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
  v &= 0x3FFu;
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

__global__ void morton_kernel(const float4 *__restrict__ p, size_t n, const float *__restrict__ bbox,
                              uint32_t *__restrict__ code, uint32_t *__restrict__ idx)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = p[i];
  const float c[3] = {v.x, v.y, v.z};
  uint32_t q[3];
  for (int k = 0; k < 3; ++k) {
    const float ext = bbox[3 + k] - bbox[k];
    float f = ext > 0.f ? (c[k] - bbox[k]) / ext : 0.f;
    f = fminf(fmaxf(f, 0.f), 1.f);                   // also maps NaN to 0
    q[k] = (uint32_t)(f * 1023.0f);
  }
  // 30-bit Hilbert index (Skilling's transpose algorithm): unlike the Morton
  // curve it has no jumps, so 256 consecutive points form a compact patch
  // (measured on the 200k turntable pair: worst wave 24 overlapping tiles
  // instead of 142, mean 5.4 instead of 8.8).
  uint32_t X[3] = {q[0], q[1], q[2]};
  for (uint32_t Qb = 1u << 9; Qb > 1; Qb >>= 1) {
    const uint32_t P = Qb - 1;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (X[k] & Qb) X[0] ^= P;
      else { const uint32_t t = (X[0] ^ X[k]) & P; X[0] ^= t; X[k] ^= t; }
    }
  }
  X[1] ^= X[0]; X[2] ^= X[1];
  uint32_t t = 0;
  for (uint32_t Qb = 1u << 9; Qb > 1; Qb >>= 1) if (X[2] & Qb) t ^= Qb - 1;
  X[0] ^= t; X[1] ^= t; X[2] ^= t;
  code[i] = (spread10(X[0]) << 2) | (spread10(X[1]) << 1) | spread10(X[2]);
  idx[i] = (uint32_t)i;
}

__global__ void finish_order_kernel(const uint32_t *__restrict__ perm, size_t n, uint32_t *__restrict__ inv)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) inv[perm[k]] = (uint32_t)k;
}

// minimum / maximum of a float over the 64 lanes, the same value in every lane: a running min / max inside each row of
// 16 lanes by DPP row shifts (a lane with no source keeps its own value: min and max are idempotent), then the row results
// (lanes 15, 31, 47, 63) as scalars
template <bool MAX>
__device__ __forceinline__ float wave_minmax_f32(float v)
{
  auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : fminf(a, b); };
  int b = __float_as_int(v);
  v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(b, b, 0x111, 0xF, 0xF, false))); b = __float_as_int(v);
  v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(b, b, 0x112, 0xF, 0xF, false))); b = __float_as_int(v);
  v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(b, b, 0x114, 0xF, 0xF, false))); b = __float_as_int(v);
  v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(b, b, 0x118, 0xF, 0xF, false))); b = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 15)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 31));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 47)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 63));
  return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ float wave_min_f32(float v) { return wave_minmax_f32<false>(v); }
__device__ __forceinline__ float wave_max_f32(float v) { return wave_minmax_f32<true>(v); }

// sorted[k] = {pts[perm[k]].xyz, bits(perm[k])}; one wave per 256-point tile also reduces the AABBs
// of its four 64-point cells (cbox[tile][cell] = {lo, hi}) and of the tile (their union)
// blockIdx.y = cloud of the batch (posed scans of one global iteration are refreshed in one launch)
__global__ void __launch_bounds__(256) refresh_sorted_kernel(RefreshBatch rb)
{
  const int cloud = blockIdx.y;
  // the pose straight from where the host put it (pinned memory, one 128-byte read per block), block 0 of the cloud leaving the
  // device record behind for the launches that follow (what a launch of its own did at the head of every pass: 4 us + a gap)
  // (whichever way the pose comes -- pinned memory, device record, kernel argument -- the block's 16 doubles go through LDS:
  // one address space for the loop below)
  __shared__ Mat44d s_T;
  const double *tin = rb.Tin[cloud];
  if (rb.from[cloud]) {
    const double *tsrc = tin ? tin : (rb.Tp[cloud] ? rb.Tp[cloud]->m : nullptr);
    // (a pose that comes from the host's pinned table carries, in the slot of its always-zero element [3], how far the cloud has
    // moved since the pass before -- ring_passes puts it there; it goes into the device record and the element is zero again)
    if (threadIdx.x < 16) s_T.m[threadIdx.x] = (tin && threadIdx.x == 3) ? 0.0 : (tsrc ? tsrc[threadIdx.x] : rb.T[cloud].m[threadIdx.x]);
    __syncthreads();
    if (tin && blockIdx.x == 0 && threadIdx.x == 0) make_pose_rec(s_T.m, (float)tin[3], reinterpret_cast<PoseRec *>(const_cast<Mat44d *>(rb.Tp[cloud])));
  }
  const float4 *__restrict__ pts = rb.pts[cloud];
  const uint32_t *__restrict__ perm = rb.perm[cloud];
  const float4 *__restrict__ from = rb.from[cloud];
  const size_t n = rb.n[cloud];
  float4 *__restrict__ sorted = rb.sorted[cloud], *__restrict__ tlo = rb.tlo[cloud], *__restrict__ thi = rb.thi[cloud],
         *__restrict__ cbox = rb.cbox[cloud];
  const int lane = threadIdx.x & 63;
  const size_t tile = (size_t)rb.tile_begin[cloud] + (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t base = tile * kCullTile;
  if (base >= n) return;
  float tl[3] = {3.0e38f, 3.0e38f, 3.0e38f}, th[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
  for (int r = 0; r < kCullTile / 64; ++r) {
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};     // an empty cell is never needed
    const size_t k = base + r * 64 + lane;
    if (k < n) {
      float4 v;
      if (from) {                     // posed copy of an indexed cloud: pose its sorted copy, in order
        const float4 p = from[k];
        const Mat44d &T = s_T;
        v = pose_point_f64(T, p);
        v.w = p.w;
        if (rb.xsrc[cloud]) rb.xdst[cloud][k] = pose_point_f64(T, rb.xsrc[cloud][k]);     // and the points in original order
        if (rb.graw[cloud]) {                                                             // and in the order of the set's cell grid
          const float4 g = rb.graw[cloud][k];
          float4 w = pose_point_f64(T, g);
          w.w = g.w;
          rb.gout[cloud][k] = w;
        }
      } else {
        const uint32_t o = perm[k];
        v = pts[o];
        v.w = __uint_as_float(o);
      }
      sorted[k] = v;
      lo[0] = hi[0] = v.x; lo[1] = hi[1] = v.y; lo[2] = hi[2] = v.z;
    }
    // box of the 64 points: minimum / maximum over the wave without the LDS crossbar (DPP row shifts, then the four row
    // results read as scalars; six 6-step shuffle reductions per cell kept this kernel on the ds_bpermute pipe)
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2) { lo[k2] = wave_min_f32(lo[k2]); hi[k2] = wave_max_f32(hi[k2]); }
    if (lane == 0) {
      cbox[(tile * 4 + r) * 2 + 0] = make_float4(lo[0], lo[1], lo[2], 0.f);
      cbox[(tile * 4 + r) * 2 + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
    }
    for (int k2 = 0; k2 < 3; ++k2) { tl[k2] = fminf(tl[k2], lo[k2]); th[k2] = fmaxf(th[k2], hi[k2]); }
  }
  if (lane == 0) {
    tlo[tile] = make_float4(tl[0], tl[1], tl[2], 0.f);
    thi[tile] = make_float4(th[0], th[1], th[2], 0.f);
  }
}

// one wave per super box: the union of 64 consecutive tile boxes
__global__ void __launch_bounds__(64) super_box_kernel(RefreshBatch rb)
{
  const int cloud = blockIdx.y;
  const float4 *__restrict__ tlo = rb.tlo[cloud], *__restrict__ thi = rb.thi[cloud];
  float4 *__restrict__ sbox = rb.sbox[cloud];
  const size_t n_tiles = ((size_t)rb.n[cloud] + kCullTile - 1) / kCullTile;
  const int lane = threadIdx.x;
  const size_t sb = (size_t)(rb.tile_begin[cloud] / 64u) + blockIdx.x;       // (a partial refresh starts at the super box of its first tile)
  if (sb * 64 >= n_tiles) return;
  const size_t t = sb * 64 + lane;
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  if (t < n_tiles) {
    const float4 a = tlo[t], b = thi[t];
    lo[0] = a.x; lo[1] = a.y; lo[2] = a.z; hi[0] = b.x; hi[1] = b.y; hi[2] = b.z;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    for (int k = 0; k < 3; ++k) {
      lo[k] = fminf(lo[k], __shfl_xor(lo[k], o, 64));
      hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o, 64));
    }
  if (lane == 0) {
    sbox[2 * sb] = make_float4(lo[0], lo[1], lo[2], 0.f);
    sbox[2 * sb + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
  }
}

// dst's ordering grows by an appended cloud's own: perm / inv entries old_n .. old_n + n_s - 1
__global__ void append_order_kernel(uint32_t *__restrict__ perm, uint32_t *__restrict__ inv, const uint32_t *__restrict__ sperm,
                                    const uint32_t *__restrict__ sinv, size_t n_s, uint32_t old_n)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_s) return;
  perm[old_n + k] = old_n + sperm[k];
  inv[old_n + k] = old_n + sinv[k];
}

// ------------------------------------------------------ reciprocal glue (sorted space)

__global__ void flag_matched_kernel(const nnkey_t *__restrict__ keys, const uint32_t *__restrict__ qperm,
                                    size_t q_begin, size_t q_count, double max2, const uint32_t *__restrict__ tinv,
                                    uint8_t *__restrict__ flags)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= q_count) return;
  const size_t i = qperm ? qperm[q_begin + k] : (q_begin + k);
  const nnkey_t key = keys[i];
  const uint32_t j = (uint32_t)key;
  if (j == kNone) return;
  const float d2 = __uint_as_float((uint32_t)(key >> 32));
  if ((double)d2 > max2) return;
  flags[tinv[j]] = 1;          // same value from every writer
}

// bound[sorted target position] = bits of the forward d2 of A source that matched the target (array preset to ~0)
__global__ void seed_bounds_kernel(const nnkey_t *__restrict__ keys, const uint32_t *__restrict__ qperm, size_t q_begin,
                                   size_t q_count, double max2, const uint32_t *__restrict__ tinv, uint32_t *__restrict__ bound,
                                   uint32_t *__restrict__ seed, uint32_t *__restrict__ zero_word)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0 && zero_word) *zero_word = 0u;      // (optional: a counter of the launches that follow, put back to zero on the way)
  if (k >= q_count) return;
  const size_t i = qperm ? qperm[q_begin + k] : (q_begin + k);
  const nnkey_t key = keys[i];
  const uint32_t j = (uint32_t)key;
  const uint32_t tpos = j != kNone ? tinv[j] : kNone;
  if (seed) seed[k] = tpos;          // (optional: where this query's match sits -- what the scan's NEXT align starts from, Ctx::seq_seed)
  if (j == kNone) return;
  if ((double)__uint_as_float((uint32_t)(key >> 32)) > max2) return;
  __hip_atomic_store(&bound[tpos], (uint32_t)(key >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // any match's distance will do (see flag_matched_batch_kernel)
}

__global__ void flag_matched_batch_kernel(GlueBatch b)
{
  const GluePair &a = b.p[blockIdx.y];
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.q_count) return;
  const size_t i = a.by_pos ? (size_t)(a.q_begin + k) : (size_t)a.qperm[a.q_begin + k];      // key slot
  const nnkey_t key = a.keys[i];
  const uint32_t j = (uint32_t)key;
  if (j == kNone) return;
  const float d2 = __uint_as_float((uint32_t)(key >> 32));
  if ((double)d2 > b.max2) return;
  // (by_pos: the forward keys carry the match's sorted position -- no original-index -> position gather here)
  const uint32_t tpos = a.by_pos ? j : a.tinv[j];
  // the reverse search of that target may start from this distance: the point that matched it is that close
  if (a.bound) {
    // ANY matching source's distance is a valid start bound, so the writers are not ordered: a relaxed store, last
    // one wins (an atomic min per match cost 40 us per ring step and pruned 0.3 % more).  Which one wins only moves
    // the amount of pruning from run to run, never a result.
    __hip_atomic_store(&a.bound[tpos], (uint32_t)(key >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else a.flags[tpos] = 1;
}

// ---- ordered compaction of the flags of ALL pairs of a batch (blockIdx.y = pair): count per chunk of 1024 positions,
// exclusive scan of the chunk counts (one block per pair), then every chunk writes its flagged positions in order.
// A thread takes FOUR consecutive positions (one 16-byte load of the bounds): with one position per thread the two launches
// were 9.4k blocks of bookkeeping each on the 12 x 200k ring (6 + 11 us); a quarter of the blocks, a quarter of the counts
// every block adds up.
constexpr int kChunk = 256;                   // threads of a chunk's block
constexpr int kChunkPer = 4;                  // positions per thread
constexpr int kChunkPos = kChunk * kChunkPer; // positions per chunk
struct __attribute__((packed, aligned(4))) Bound4 { uint32_t a, b, c, d; };      // (a pair's stretch of the bounds starts anywhere: 4-byte aligned)

__device__ __forceinline__ unsigned flagged4(const GluePair &a, const size_t base)      // bit j: position base + j is flagged
{
  unsigned m = 0;
  if (a.bound && base + kChunkPer <= a.nt) {
    const Bound4 v = *reinterpret_cast<const Bound4 *>(a.bound + base);
    m = (v.a != 0xFFFFFFFFu ? 1u : 0u) | (v.b != 0xFFFFFFFFu ? 2u : 0u) | (v.c != 0xFFFFFFFFu ? 4u : 0u) | (v.d != 0xFFFFFFFFu ? 8u : 0u);
  } else {
#pragma unroll
    for (int j = 0; j < kChunkPer; ++j)
      if (base + j < a.nt && (a.bound ? a.bound[base + j] != 0xFFFFFFFFu : a.flags[base + j] != 0)) m |= 1u << j;
  }
  return m;
}

__device__ __forceinline__ unsigned chunk_rank(const unsigned m, unsigned *total)      // flagged positions of the block before this thread's four
{
  __shared__ unsigned wsum[kChunk / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long lower = (1ull << lane) - 1ull;
  unsigned mine = 0, wave_all = 0;
#pragma unroll
  for (int j = 0; j < kChunkPer; ++j) {
    const unsigned long long bj = __ballot((m >> j) & 1u);
    mine += (unsigned)__popcll(bj & lower);
    wave_all += (unsigned)__popcll(bj);
  }
  if (lane == 0) wsum[wave] = wave_all;
  __syncthreads();
  unsigned before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kChunk / 64; ++w) { if (w < wave) before += wsum[w]; all += wsum[w]; }
  *total = all;
  return before + mine;
}

__device__ __forceinline__ void chunk_write(const GluePair &a, const unsigned m, const size_t base, const unsigned first)
{
#pragma unroll
  for (int j = 0; j < kChunkPer; ++j)
    if ((m >> j) & 1u) {
      const unsigned at = first + (unsigned)__popc(m & ((1u << j) - 1u));
      a.list[at] = (uint32_t)(base + j); a.slot[base + j] = at;
    }
}

__global__ void __launch_bounds__(kChunk) count_flags_batch_kernel(GlueBatch b)
{
  const GluePair &a = b.p[blockIdx.y];
  if ((size_t)blockIdx.x * kChunkPos >= a.nt) return;            // block-uniform
  unsigned total;
  (void)chunk_rank(flagged4(a, (size_t)blockIdx.x * kChunkPos + (size_t)threadIdx.x * kChunkPer), &total);
  if (threadIdx.x == 0) a.chunks[blockIdx.x] = total;
}

// exclusive scan of a pair's chunk counts by ONE block of 1024 threads: 1024 counts per round, a wave-level shuffle scan
// and one hop across the sixteen waves (the 256-thread Hillis-Steele version took 31 us for the 9.4k chunks of a
// 2.4M-point target -- the sequential mode's merged model)
constexpr int kScanThreads = 1024;
__global__ void __launch_bounds__(kScanThreads) scan_chunks_batch_kernel(GlueBatch b)
{
  const GluePair &a = b.p[blockIdx.x];
  const size_t n = ((size_t)a.nt + kChunkPos - 1) / kChunkPos;
  __shared__ unsigned wave_tot[kScanThreads / 64];
  __shared__ unsigned carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (size_t base = 0; base < n; base += kScanThreads) {
    const size_t i = base + threadIdx.x;
    const unsigned v = i < n ? a.chunks[i] : 0u;
    unsigned inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned u = (unsigned)__shfl_up((int)inc, o, 64); if (lane >= o) inc += u; }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    unsigned before = carry;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    if (i < n) a.chunks[i] = before + inc - v;               // exclusive
    __syncthreads();
    if (threadIdx.x == kScanThreads - 1) carry = before + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *a.qcount = carry;
}

__global__ void __launch_bounds__(kChunk) compact_flags_batch_kernel(GlueBatch b)
{
  const GluePair &a = b.p[blockIdx.y];
  if ((size_t)blockIdx.x * kChunkPos >= a.nt) return;            // block-uniform
  const size_t base = (size_t)blockIdx.x * kChunkPos + (size_t)threadIdx.x * kChunkPer;
  const unsigned m = flagged4(a, base);
  unsigned total;
  const unsigned r = chunk_rank(m, &total);
  chunk_write(a, m, base, a.chunks[blockIdx.x] + r);
}

// the same without the scan launch, for pairs of up to kOwnPrefixChunks chunks: chunks[] holds the RAW counts and every
// block sums the counts before its own (a few hundred L2-resident words; the scan launch cost 6 us per pass for that)
constexpr unsigned kOwnPrefixChunks = 1024;
__global__ void __launch_bounds__(kChunk) compact_flags_own_prefix_batch_kernel(GlueBatch b)
{
  const GluePair &a = b.p[blockIdx.y];
  if ((size_t)blockIdx.x * kChunkPos >= a.nt) return;            // block-uniform
  const size_t base = (size_t)blockIdx.x * kChunkPos + (size_t)threadIdx.x * kChunkPer;
  const unsigned m = flagged4(a, base);              // (requested before the counts are summed: the two waits overlap)
  __shared__ unsigned psum[kChunk / 64];
  unsigned mine = 0;
  for (unsigned c = threadIdx.x; c < blockIdx.x; c += kChunk) mine += a.chunks[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if ((threadIdx.x & 63) == 0) psum[threadIdx.x >> 6] = mine;
  __syncthreads();
  unsigned prefix = 0;
#pragma unroll
  for (int w = 0; w < kChunk / 64; ++w) prefix += psum[w];
  unsigned total;
  const unsigned r = chunk_rank(m, &total);
  chunk_write(a, m, base, prefix + r);
  if (threadIdx.x == 0 && (size_t)(blockIdx.x + 1) * kChunkPos >= a.nt) *a.qcount = prefix + total;      // the pair's last chunk knows the count
}

__global__ void scatter_slot_kernel(const uint32_t *__restrict__ list, const uint32_t *__restrict__ count, size_t cap,
                                    uint32_t *__restrict__ slot)
{
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < cap && p < *count) slot[list[p]] = (uint32_t)p;
}

int ensure_cub(Ctx *c, size_t bytes)
{
  if (c->cub_cap >= bytes) return MVR_OK;
  MVR_MAY_BLOCK(c, "hipCUB scratch has to grow");
  if (c->cub_tmp) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->cub_tmp); c->cub_tmp = nullptr; c->cub_cap = 0; }
  MVR_HIP_TRY(c, hipMalloc(&c->cub_tmp, bytes + 4096));
  c->cub_cap = bytes + 4096;
  return MVR_OK;
}

}  // namespace

void new_point_set(Ctx *c, Cloud &cl)
{
  cl.set_id = c->next_set_id++;
  cl.order.reset();
  cl.forget_pose(); cl.canonical = true;        // its points, as they are now, DEFINE the new set: canonical coordinates
  cl.stale_coords();
}

void inherit_point_set(Cloud &dst, const Cloud &src)
{
  // same point set, same indexing: the ordering stays valid.  Keep an order
  // the destination already holds for this very set (a scan re-posed every
  // step from its never-searched raw copy), otherwise take the source's.
  const bool same = (dst.set_id == src.set_id) && dst.order && dst.order->n == src.n;
  dst.set_id = src.set_id;
  if (src.order) dst.order = src.order;
  else if (!same) dst.order.reset();
  dst.forget_pose();      // (the callers that know the pose say so afterwards)
  dst.stale_coords();
}

int cloud_bbox(Ctx *c, const float4 *pts, size_t n, float out[6])
{
  if (n == 0) return MVR_E_ARG;
  MVR_MAY_BLOCK(c, "a bounding box is read back");
  if (int rc = ensure(c, c->partials, c->partials_cap, (size_t)1024 * 32)) return rc;
  const int bb = (int)std::min<size_t>(256, (n + 255) / 256);
  float *part = reinterpret_cast<float *>(c->partials);
  hipLaunchKernelGGL(bbox_partial_kernel, dim3(bb), dim3(256), 0, c->stream, pts, n, part);
  hipLaunchKernelGGL(bbox_final_kernel, dim3(1), dim3(64), 0, c->stream, part, bb, c->bbox);
  MVR_HIP_TRY(c, hipMemcpyAsync(out, c->bbox, 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MVR_OK;
}

bool extend_point_set(Ctx *c, Cloud &dst, size_t old_n, const Cloud &src)
{
  const size_t add = src.n, n = old_n + add;
  if (c->nn_mode == 0 || old_n == 0 || add == 0 || n > 0xFFFFFFF0ull) return false;
  if (!dst.order || dst.order->n != old_n || !src.order || src.order->n != add || dst.order == src.order) return false;
  if (dst.order.use_count() != 1) return false;                          // another cloud shares dst's ordering: it must not change under it
  if (dst.order->perm_bytes < n * sizeof(uint32_t) || dst.order->inv_bytes < n * sizeof(uint32_t)) return false;
  hipLaunchKernelGGL(append_order_kernel, dim3((unsigned)((add + 255) / 256)), dim3(256), 0, c->stream, dst.order->perm, dst.order->inv,
                     src.order->perm, src.order->inv, add, (uint32_t)old_n);
  if (hipGetLastError() != hipSuccess) return false;
  dst.order->n = n;
  c->orders.erase(dst.set_id);
  dst.set_id = c->next_set_id++;                                         // a different point set, with an ordering already
  c->orders[dst.set_id] = dst.order;
  // the sorted copy and the boxes of the points that were there stay valid (they did not move): only the tail is refreshed
  const bool head_ok = dst.coords_valid && dst.sorted != nullptr;
  dst.coords_valid = false;
  dst.fresh_tiles = head_ok ? old_n / kCullTile : 0;
  return true;
}

// ---- the orderings of SEVERAL point sets in one go (a registration's first pass needs one per scan: twelve times bounding box ->
// codes -> radix sort -> inverse was ~180 launches and 24 allocations, 1.2 ms of a host-bound 3.7 ms pass).  The clouds' codes
// are sorted TOGETHER under a composite key, (cloud << 30) | 30-bit Hilbert code, on the bits [kSortLoBit, 30 + bits of the cloud
// number): the radix sort is stable, so inside a cloud the order is exactly the one the per-cloud sort gives (same bits compared,
// ties in input order) -- the permutations are bit-identical to the one-at-a-time build.
namespace {
struct OrderBatchArgs { const float4 *pts[kBatchClouds]; unsigned long long n[kBatchClouds], off[kBatchClouds]; uint32_t *perm[kBatchClouds], *inv[kBatchClouds]; };
__global__ void bbox_many_kernel(OrderBatchArgs a, float *__restrict__ part /* [cloud][64 blocks][6] */)
{
  const int cl = blockIdx.y;
  const float4 *p = a.pts[cl];
  const size_t cnt = (size_t)a.n[cl];
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
// codes of all clouds: the bounding box of a cloud from its 64 partial rows (every block folds them itself: 384 floats from the L2),
// then exactly morton_kernel's arithmetic
__global__ void hilbert_many_kernel(OrderBatchArgs a, const float *__restrict__ part, int part_blocks, unsigned long long *__restrict__ key, uint32_t *__restrict__ idx)
{
  const int cl = blockIdx.y;
  __shared__ float bb[6];
  if (threadIdx.x < 6) {
    const bool is_hi = threadIdx.x >= 3;
    float v = is_hi ? -3.0e38f : 3.0e38f;
    for (int b = 0; b < part_blocks; ++b) { const float w = part[((size_t)cl * part_blocks + b) * 6 + threadIdx.x]; v = is_hi ? fmaxf(v, w) : fminf(v, w); }
    bb[threadIdx.x] = v;
  }
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)a.n[cl]) return;
  const float4 v = a.pts[cl][i];
  const float c[3] = {v.x, v.y, v.z};
  uint32_t q[3];
  for (int k = 0; k < 3; ++k) {
    const float ext = bb[3 + k] - bb[k];
    float f = ext > 0.f ? (c[k] - bb[k]) / ext : 0.f;
    f = fminf(fmaxf(f, 0.f), 1.f);
    q[k] = (uint32_t)(f * 1023.0f);
  }
  uint32_t X[3] = {q[0], q[1], q[2]};
  for (uint32_t Qb = 1u << 9; Qb > 1; Qb >>= 1) {
    const uint32_t P = Qb - 1;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (X[k] & Qb) X[0] ^= P;
      else { const uint32_t t = (X[0] ^ X[k]) & P; X[0] ^= t; X[k] ^= t; }
    }
  }
  X[1] ^= X[0]; X[2] ^= X[1];
  uint32_t t = 0;
  for (uint32_t Qb = 1u << 9; Qb > 1; Qb >>= 1) if (X[2] & Qb) t ^= Qb - 1;
  X[0] ^= t; X[1] ^= t; X[2] ^= t;
  const uint32_t code = (spread10(X[0]) << 2) | (spread10(X[1]) << 1) | spread10(X[2]);
  key[(size_t)a.off[cl] + i] = ((unsigned long long)cl << 30) | code;
  idx[(size_t)a.off[cl] + i] = (uint32_t)i;
}
__global__ void finish_many_kernel(OrderBatchArgs a, const uint32_t *__restrict__ sorted_idx)
{
  const int cl = blockIdx.y;
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= (size_t)a.n[cl]) return;
  const uint32_t o = sorted_idx[(size_t)a.off[cl] + k];
  a.perm[cl][k] = o;
  a.inv[cl][o] = (uint32_t)k;
}
size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
}  // namespace

// clouds that have no ordering yet (and none in the context's cache) get one, all together; the others are left alone.
// Failure is not fatal: prepare_index builds what is still missing, one by one.
static int build_orders_batch(Ctx *c, Cloud *const *clouds, int count)
{
  std::vector<Cloud *> todo;
  for (int k = 0; k < count; ++k) {
    Cloud *cl = clouds[k];
    if (!cl || cl->n == 0 || cl->n > 0x3FFFFFFFull) continue;
    if (cl->order && cl->order->n == cl->n) continue;
    auto it = c->orders.find(cl->set_id);
    if (it != c->orders.end()) { auto o = it->second.lock(); if (o && o->n == cl->n) continue; }
    bool dup = false;
    for (Cloud *t : todo) dup = dup || t->set_id == cl->set_id;
    if (!dup) todo.push_back(cl);
  }
  if (todo.size() < 2 || !c->order_batch) return MVR_OK;
  MVR_MAY_BLOCK(c, "point sets have no ordering yet");
  constexpr int kBoxBlocks = 64;
  for (size_t base = 0; base < todo.size(); base += kBatchClouds) {
    const int m = (int)std::min<size_t>(kBatchClouds, todo.size() - base);
    OrderBatchArgs a;
    size_t total = 0, nmax = 0, arena_bytes = 0;
    std::vector<size_t> want((size_t)m);
    for (int k = 0; k < kBatchClouds; ++k) { a.pts[k] = nullptr; a.n[k] = 0; a.off[k] = 0; a.perm[k] = a.inv[k] = nullptr; }
    for (int k = 0; k < m; ++k) {
      Cloud *cl = todo[base + (size_t)k];
      a.pts[k] = cl->pts; a.n[k] = cl->n; a.off[k] = total;
      total += cl->n; nmax = std::max(nmax, cl->n);
      want[(size_t)k] = up256(std::max(cl->n, cl->cap) * sizeof(uint32_t));      // sized for the cloud's CAPACITY, as the one-by-one build does
      arena_bytes += 2 * want[(size_t)k];
    }
    if (total > 0x7FFFFFFFull) return MVR_OK;      // (hipCUB takes an int count: such a batch is built one by one)
    int hi_bit = 30;
    while ((1 << (hi_bit - 30)) < m) ++hi_bit;
    size_t cub_bytes = 0;
    { unsigned long long *zk = nullptr; uint32_t *zv = nullptr;
      MVR_HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, cub_bytes, zk, zk, zv, zv, (int)total, kSortLoBit, hi_bit, c->stream)); }
    const size_t o_part = 0, o_ka = o_part + up256((size_t)m * kBoxBlocks * 6 * sizeof(float)), o_kb = o_ka + up256(total * 8), o_ia = o_kb + up256(total * 8),
                 o_ib = o_ia + up256(total * 4), o_cub = o_ib + up256(total * 4), need = o_cub + up256(cub_bytes + 256);
    if (int rc = ensure(c, c->oscratch, c->oscratch_cap, need)) return rc;
    char *block = nullptr;
    if (hipMalloc(&block, arena_bytes) != hipSuccess) { (void)hipGetLastError(); return MVR_OK; }
    std::shared_ptr<char> arena(block, [](char *p) { if (p) (void)hipFree(p); });
    std::vector<std::shared_ptr<Order> > ords((size_t)m);
    size_t at = 0;
    for (int k = 0; k < m; ++k) {
      auto ord = std::make_shared<Order>();
      ord->arena = arena; ord->n = todo[base + (size_t)k]->n;
      ord->perm = reinterpret_cast<uint32_t *>(block + at); at += want[(size_t)k];
      ord->inv = reinterpret_cast<uint32_t *>(block + at); at += want[(size_t)k];
      ord->perm_bytes = ord->inv_bytes = want[(size_t)k];
      a.perm[k] = ord->perm; a.inv[k] = ord->inv;
      ords[(size_t)k] = ord;
    }
    float *part = reinterpret_cast<float *>(c->oscratch + o_part);
    unsigned long long *ka = reinterpret_cast<unsigned long long *>(c->oscratch + o_ka), *kb = reinterpret_cast<unsigned long long *>(c->oscratch + o_kb);
    uint32_t *ia = reinterpret_cast<uint32_t *>(c->oscratch + o_ia), *ib = reinterpret_cast<uint32_t *>(c->oscratch + o_ib);
    ProfScope ps(c, MVR_K_GLUE, 40.0 * (double)total);
    const unsigned nb = (unsigned)((nmax + 255) / 256);
    hipLaunchKernelGGL(bbox_many_kernel, dim3(kBoxBlocks, (unsigned)m), dim3(256), 0, c->stream, a, part);
    hipLaunchKernelGGL(hilbert_many_kernel, dim3(nb, (unsigned)m), dim3(256), 0, c->stream, a, part, kBoxBlocks, ka, ia);
    size_t cb = cub_bytes;
    MVR_HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(c->oscratch + o_cub, cb, ka, kb, ia, ib, (int)total, kSortLoBit, hi_bit, c->stream));
    hipLaunchKernelGGL(finish_many_kernel, dim3(nb, (unsigned)m), dim3(256), 0, c->stream, a, ib);
    MVR_HIP_TRY(c, hipGetLastError());
    for (int k = 0; k < m; ++k) {
      Cloud *cl = todo[base + (size_t)k];
      cl->order = ords[(size_t)k];
      c->orders[cl->set_id] = ords[(size_t)k];
      for (Cloud &o : c->slots) if (&o != cl && o.set_id == cl->set_id && o.n == cl->n && !o.order) o.order = ords[(size_t)k];      // every resident copy of the set shares it
      cl->stale_coords();
    }
  }
  for (auto it = c->orders.begin(); it != c->orders.end();) it = it->second.expired() ? c->orders.erase(it) : std::next(it);
  return MVR_OK;
}

// ordering (once per point set) and buffers; *stale = the sorted copy / boxes must be refreshed
static int prepare_index(Ctx *c, Cloud &cl, bool *stale)
{
  *stale = false;
  const size_t n = cl.n;
  if (n == 0) return MVR_OK;
  const size_t tiles = (n + kCullTile - 1) / kCullTile;
  if (!cl.order || cl.order->n != n) {
    cl.order.reset();
    auto it = c->orders.find(cl.set_id);
    if (it != c->orders.end()) {
      cl.order = it->second.lock();
      if (cl.order && cl.order->n != n) cl.order.reset();
    }
    cl.stale_coords();
  }
  if (!cl.order) {
    MVR_MAY_BLOCK(c, "a point set has no ordering yet");
    // sort once per point set: bbox -> 30-bit Morton codes -> radix sort -> perm / inv
    if (c->sort_cap < n) {
      (void)hipStreamSynchronize(c->stream);
      if (c->codes_a) (void)hipFree(c->codes_a);
      if (c->codes_b) (void)hipFree(c->codes_b);
      if (c->idx_a) (void)hipFree(c->idx_a);
      c->codes_a = c->codes_b = c->idx_a = nullptr; c->sort_cap = 0;
      const size_t cap = n + n / 4 + 1024;
      MVR_HIP_TRY(c, hipMalloc(&c->codes_a, cap * 4));
      MVR_HIP_TRY(c, hipMalloc(&c->codes_b, cap * 4));
      MVR_HIP_TRY(c, hipMalloc(&c->idx_a, cap * 4));
      c->sort_cap = cap;
    }
    if (int rc = ensure(c, c->partials, c->partials_cap, (size_t)1024 * 32)) return rc;
    auto ord = std::make_shared<Order>();
    ord->pool = c->order_pool;
    // sized for the cloud's CAPACITY: a target reserved for all its scans gets buffers it can keep re-using as it grows
    const size_t want = std::max(n, cl.cap) * sizeof(uint32_t);
    for (uint32_t **slot : {&ord->perm, &ord->inv}) {
      size_t got = 0;
      uint32_t *p = c->order_pool ? c->order_pool->take(want, &got) : nullptr;
      if (!p) { MVR_HIP_TRY(c, hipMalloc(&p, want)); got = want; }
      *slot = p;
      (slot == &ord->perm ? ord->perm_bytes : ord->inv_bytes) = got;
    }
    ord->n = n;
    ProfScope ps(c, MVR_K_GLUE, 40.0 * (double)n);
    const int bb = (int)std::min<size_t>(256, (n + 255) / 256);
    float *part = reinterpret_cast<float *>(c->partials);
    hipLaunchKernelGGL(bbox_partial_kernel, dim3(bb), dim3(256), 0, c->stream, cl.pts, n, part);
    hipLaunchKernelGGL(bbox_final_kernel, dim3(1), dim3(64), 0, c->stream, part, bb, c->bbox);
    hipLaunchKernelGGL(morton_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, cl.pts, n, c->bbox,
                       c->codes_a, c->idx_a);
    size_t bytes = 0;
    MVR_HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, c->codes_a, c->codes_b, c->idx_a, ord->perm, (int)n, kSortLoBit, 30,
                                                      c->stream));
    if (int rc = ensure_cub(c, bytes)) return rc;
    MVR_HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(c->cub_tmp, bytes, c->codes_a, c->codes_b, c->idx_a, ord->perm, (int)n, kSortLoBit,
                                                      30, c->stream));
    hipLaunchKernelGGL(finish_order_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, ord->perm, n,
                       ord->inv);
    MVR_HIP_TRY(c, hipGetLastError());
    cl.order = ord;
    // drop dead cache entries, remember this one
    for (auto it = c->orders.begin(); it != c->orders.end();) it = it->second.expired() ? c->orders.erase(it) : std::next(it);
    c->orders[cl.set_id] = ord;
    // every resident copy of this point set shares the new ordering
    for (Cloud &o : c->slots) if (&o != &cl && o.set_id == cl.set_id && o.n == n && !o.order) o.order = ord;
    cl.stale_coords();
  }
  if (!cl.coords_valid) {
    // buffers follow the cloud's CAPACITY (a target reserved for all its scans is never re-allocated as it grows)
    const size_t room = std::max(n, cl.cap), room_tiles = (room + kCullTile - 1) / kCullTile;
    const float4 *sorted_before = cl.sorted;
    if (int rc = ensure(c, cl.sorted, cl.sorted_cap, cl.sorted_cap >= n ? n : room)) return rc;
    if (cl.sorted != sorted_before) cl.fresh_tiles = 0;          // a new buffer holds nothing
    if (cl.tiles_cap < tiles) {
      MVR_MAY_BLOCK(c, "a cloud's box arrays have to grow");
      cl.fresh_tiles = 0;
      // the four box arrays are ONE allocation (tlo is its start: a registration's first pass made four hipMalloc per view here,
      // and synchronised before each batch of them -- needed only when there IS an old block the GPU may still be reading)
      if (cl.tlo) { (void)hipStreamSynchronize(c->stream); (void)hipFree(cl.tlo); }
      cl.tlo = cl.thi = cl.cbox = cl.sbox = nullptr; cl.tiles_cap = 0;
      const size_t cap = std::max(tiles + tiles / 4, room_tiles) + 16;
      const size_t n_sb = (cap / 64 + 2) * 2;
      float4 *block = nullptr;
      MVR_HIP_TRY(c, hipMalloc(&block, (cap * 10 + n_sb) * sizeof(float4)));
      cl.tlo = block; cl.thi = block + cap; cl.cbox = block + 2 * cap; cl.sbox = block + 10 * cap;
      cl.tiles_cap = cap;
    }
    *stale = true;
  }
  return MVR_OK;
}

// from / T: optional, per cloud (see RefreshBatch)
static int refresh_batch(Ctx *c, Cloud *const *clouds, int count, const float4 *const *from = nullptr, const double *T = nullptr,
                         const float4 *const *xsrc = nullptr, const Mat44d *const *Tp = nullptr, bool with_grid = false)
{
  // a pass enqueued ahead of its poses whose device records nobody has filled yet: this launch does it when it poses EVERY view
  // of the table (the ordinary case: a ring pass poses all its views in one call), else the launch made for that goes first
  bool fill_table = false;
  if (Tp && c->pose_tab_pending) {
    int with = 0;
    for (int k = 0; k < count; ++k)
      if (from && from[k] && Tp[k]) {
        const PoseRec *r = reinterpret_cast<const PoseRec *>(Tp[k]);
        if (r >= c->pose_tab_cur && r < c->pose_tab_cur + c->pose_tab_n && clouds[k]->n) ++with;
      }
    fill_table = with == c->pose_tab_n && !c->pose_prep_launch;
    if (!fill_table && with) { if (int rc = ensure_pose_table(c)) return rc; }
    if (fill_table) c->pose_tab_pending = false;
  }
  for (int base = 0; base < count; base += kBatchClouds) {
    RefreshBatch rb;
    const int m = std::min(kBatchClouds, count - base);
    size_t tmax = 0; double work = 0.0;
    for (int k = 0; k < kBatchClouds; ++k) {
      Cloud *cl = k < m ? clouds[base + k] : nullptr;
      rb.from[k] = (cl && from) ? from[base + k] : nullptr;
      if (rb.from[k]) std::memcpy(rb.T[k].m, T + (size_t)(base + k) * 16, sizeof rb.T[k].m);
      rb.Tp[k] = (rb.from[k] && Tp) ? Tp[base + k] : nullptr;
      rb.Tin[k] = (rb.Tp[k] && fill_table) ? c->pose_in_cur + 16 * (reinterpret_cast<const PoseRec *>(rb.Tp[k]) - c->pose_tab_cur) : nullptr;
      rb.xsrc[k] = (rb.from[k] && xsrc) ? xsrc[base + k] : nullptr;
      rb.xdst[k] = rb.xsrc[k] ? cl->pts : nullptr;
      rb.graw[k] = nullptr; rb.gout[k] = nullptr;
      if (rb.from[k] && with_grid && !cl->gcoords_valid) {      // a posed copy whose set has a cell grid: its grid-ordered coordinates too
        bool ok = false;
        if (int rc = grid_coords_prepare(c, cl, &ok)) return rc;
        if (ok) { rb.graw[k] = cl->grid->graw; rb.gout[k] = cl->gsorted; cl->gcoords_valid = true; work += 32.0 * (double)cl->n; }
      }
      rb.pts[k] = cl ? cl->pts : nullptr; rb.perm[k] = cl ? cl->order->perm : nullptr; rb.n[k] = cl ? cl->n : 0;
      rb.sorted[k] = cl ? cl->sorted : nullptr; rb.tlo[k] = cl ? cl->tlo : nullptr; rb.thi[k] = cl ? cl->thi : nullptr;
      rb.cbox[k] = cl ? cl->cbox : nullptr; rb.sbox[k] = cl ? cl->sbox : nullptr;
      // a cloud that only grew at its end (fresh_tiles) refreshes from the tile of its first new point on; posed
      // copies (from) are always refreshed whole
      const size_t t0 = (cl && !rb.from[k]) ? std::min(cl->fresh_tiles, (cl->n + kCullTile - 1) / kCullTile) : 0;
      rb.tile_begin[k] = (unsigned)t0;
      if (cl) { tmax = std::max(tmax, (cl->n + kCullTile - 1) / kCullTile - t0); work += 36.0 * (double)(cl->n - t0 * kCullTile); }
    }
    if (int rc = flush_g2h(c)) return rc;      // (the position maps of the grids this batch met for the first time: one launch)
    if (tmax == 0) continue;
    ProfScope ps(c, MVR_K_GLUE, work);
    hipLaunchKernelGGL(refresh_sorted_kernel, dim3((unsigned)((tmax + 3) / 4), (unsigned)m), dim3(256), 0, c->stream, rb);
    // The super boxes are read by the culled kernel alone.  Posed copies that have just got their grid-ordered coordinates are
    // about to be searched over their grids (every pass of a registration but the first): their super boxes wait until a
    // culled launch asks for them (flush_super_boxes) -- a launch of 4 us per pass that nothing read.
    // (Every refresh leaves them, not only those: a culled launch flushes ALL stale clouds of the context in one launch -- the
    // source and the grown model of a sequential align were two super-box launches, one behind each refresh.)
    const bool lazy = c->lazy_super != 0;
    if (!lazy) hipLaunchKernelGGL(super_box_kernel, dim3((unsigned)((tmax + 63) / 64 + 1), (unsigned)m), dim3(64), 0, c->stream, rb);     // (+1: a partial range may straddle one more super box)
    MVR_HIP_TRY(c, hipGetLastError());
    for (int k = 0; k < m; ++k) { clouds[base + k]->coords_valid = true; clouds[base + k]->fresh_tiles = 0; clouds[base + k]->super_stale = lazy && rb.n[k] != 0; }
  }
  return MVR_OK;
}

int flush_super_boxes(Ctx *c)
{
  // (the clouds belong to the context the caller made; a worker -- a group of pairs on its own stream -- finds them there.  The
  // passes that fork flush on the caller's stream BEFORE the fork, so a worker normally finds nothing stale; if it does, it
  // refreshes on its own stream and leaves the flag: the others then do the same -- identical bytes, no order needed)
  Ctx *owner = c->parent ? c->parent : c;
  const bool keep_flag = owner != c;
  Cloud *stale[kBatchClouds];
  int m = 0;
  auto launch = [&]() -> int {
    if (m == 0) return MVR_OK;
    RefreshBatch rb;
    size_t tmax = 0;
    for (int k = 0; k < kBatchClouds; ++k) {
      Cloud *cl = k < m ? stale[k] : nullptr;
      rb.tlo[k] = cl ? cl->tlo : nullptr; rb.thi[k] = cl ? cl->thi : nullptr; rb.sbox[k] = cl ? cl->sbox : nullptr;
      rb.n[k] = cl ? cl->n : 0; rb.tile_begin[k] = 0;
      if (cl) { tmax = std::max(tmax, (cl->n + kCullTile - 1) / kCullTile); if (!keep_flag) cl->super_stale = false; }
    }
    hipLaunchKernelGGL(super_box_kernel, dim3((unsigned)((tmax + 63) / 64 + 1), (unsigned)m), dim3(64), 0, c->stream, rb);
    MVR_HIP_TRY(c, hipGetLastError());
    m = 0;
    return MVR_OK;
  };
  for (Cloud &cl : owner->slots) {
    if (!cl.super_stale) continue;
    if (!cl.tlo || !cl.sbox || cl.n == 0 || !cl.coords_valid) { if (!keep_flag) cl.super_stale = false; continue; }      // (nothing to bring up to date: the next refresh writes them)
    stale[m++] = &cl;
    if (m == kBatchClouds) { if (int rc = launch()) return rc; }
  }
  return launch();
}

int refresh_posed_batch(Ctx *c, int count, Cloud *const *dst, Cloud *const *src, const double *T, bool with_pts, char *handled)
{
  std::vector<Cloud *> todo; std::vector<const float4 *> from, xsrc; std::vector<double> Ts; std::vector<const Mat44d *> Tps;
  {   // what the loop below would build one by one, together first: the orderings of the sets that have none (from the POSED copies,
      // see below), then the sources' sorted copies in one refresh launch
    std::vector<Cloud *> ds, ss;
    for (int k = 0; k < count; ++k) {
      Cloud *d = dst[k], *s = src[k];
      if (!d || !s || d == s || d->n == 0 || d->n != s->n || d->set_id != s->set_id) continue;
      ds.push_back(d); ss.push_back(s);
    }
    if (ds.size() >= 2) {
      if (int rc = build_orders_batch(c, ds.data(), (int)ds.size())) return rc;
      // (only the sources whose set HAS its ordering by now: one that has none gets it in the loop below, from its posed copy)
      std::vector<Cloud *> ready;
      for (size_t k = 0; k < ss.size(); ++k) if (ds[k]->order && ds[k]->order->n == ds[k]->n && ss[k]->order == ds[k]->order) ready.push_back(ss[k]);
      if (ready.size() >= 2) { if (int rc = ensure_index_batch(c, ready.data(), (int)ready.size())) return rc; }
    }
  }
  for (int k = 0; k < count; ++k) {
    Cloud *d = dst[k], *s = src[k];
    if (handled) handled[k] = 0;
    if (!d || !s || d == s || d->n == 0 || d->n != s->n || d->set_id != s->set_id) continue;
    if (std::find(todo.begin(), todo.end(), d) != todo.end()) continue;      // posed twice in one call: the lazy refresh sorts it out
    // A point set without an ordering gets it from the POSED copy: cells are compact boxes in the frame the order
    // was built in, and the search runs in the posed frame (an order built in the raw sensor frame, rotated by the
    // pose, inflates every box: measured 15 % more distance evaluations on the turntable ring).
    bool stale = false;
    if (int rc = prepare_index(c, *d, &stale)) return rc;   // ordering (shared by all copies of the set) + buffers
    if (int rc = ensure_index(c, *s)) return rc;            // the source's sorted copy: gathered once, then reused every pose
    if (!stale || d->order != s->order) continue;
    todo.push_back(d); from.push_back(s->sorted); xsrc.push_back(s->pts); Ts.insert(Ts.end(), T + (size_t)k * 16, T + (size_t)k * 16 + 16);
    Tps.push_back((c->pose_from_table && d->pose_dev) ? &d->pose_dev->T : nullptr);      // (an address in device memory: not read here)
    if (handled) handled[k] = 1;
  }
  // a rider: a cloud whose index the caller knows to be needed right after these posed copies (the grown model of the sequential
  // mode: its tail is refreshed by the launch that poses the next source, not by one of its own behind it)
  if (!todo.empty() && c->refresh_rider && std::find(todo.begin(), todo.end(), c->refresh_rider) == todo.end()) {
    Cloud *r = c->refresh_rider;
    bool stale = false;
    if (r->n && r->order && prepare_index(c, *r, &stale) == MVR_OK && stale) {
      todo.push_back(r); from.push_back(nullptr); xsrc.push_back(nullptr); Ts.insert(Ts.end(), 16, 0.0); Tps.push_back(nullptr);
    }
  }
  return todo.empty() ? MVR_OK : refresh_batch(c, todo.data(), (int)todo.size(), from.data(), Ts.data(), with_pts ? xsrc.data() : nullptr, Tps.data(),
                                             c->ring_search != 0 && c->pair_fused);
}

int ensure_index(Ctx *c, Cloud &cl)
{
  bool stale = false;
  if (int rc = prepare_index(c, cl, &stale)) return rc;
  if (!stale) return MVR_OK;
  Cloud *one = &cl;
  return refresh_batch(c, &one, 1);
}

int ensure_index_batch(Ctx *c, Cloud *const *clouds, int count)
{
  if (count >= 2) { if (int rc = build_orders_batch(c, clouds, count)) return rc; }
  host_mark("      index batch: orderings");
  std::vector<Cloud *> todo;
  for (int k = 0; k < count; ++k) {
    Cloud *cl = clouds[k];
    if (std::find(todo.begin(), todo.end(), cl) != todo.end()) continue;
    bool stale = false;
    if (int rc = prepare_index(c, *cl, &stale)) return rc;
    if (stale) todo.push_back(cl);
  }
  host_mark("      index batch: buffers");
  const int rc = todo.empty() ? MVR_OK : refresh_batch(c, todo.data(), (int)todo.size());
  host_mark("      index batch: refreshed");
  return rc;
}

int launch_compact_flags_batch(Ctx *c, const GlueBatch &b, int n_pairs)
{
  size_t tmax = 0; double work = 0.0;
  for (int k = 0; k < n_pairs; ++k) { tmax = std::max(tmax, (size_t)b.p[k].nt); work += 10.0 * (double)b.p[k].nt; }
  if (tmax == 0) return MVR_OK;
  ProfScope ps(c, MVR_K_GLUE, work);
  const dim3 grid((unsigned)((tmax + kChunkPos - 1) / kChunkPos), (unsigned)n_pairs);
  hipLaunchKernelGGL(count_flags_batch_kernel, grid, dim3(kChunk), 0, c->stream, b);
  if (grid.x <= kOwnPrefixChunks) {
    hipLaunchKernelGGL(compact_flags_own_prefix_batch_kernel, grid, dim3(kChunk), 0, c->stream, b);
  } else {
    hipLaunchKernelGGL(scan_chunks_batch_kernel, dim3((unsigned)n_pairs), dim3(kScanThreads), 0, c->stream, b);
    hipLaunchKernelGGL(compact_flags_batch_kernel, grid, dim3(kChunk), 0, c->stream, b);
  }
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_mark_sorted(Ctx *c, const nnkey_t *keys, const uint32_t *qperm, size_t q_begin, size_t q_count, double max2,
                       const uint32_t *tinv, size_t nt, uint8_t *flags, uint32_t *list, uint32_t *count, uint32_t *slot)
{
  if (q_count == 0 || nt == 0) return MVR_OK;
  if (int rc = launch_flag_matched(c, keys, qperm, q_begin, q_count, max2, tinv, nt, flags)) return rc;
  ProfScope ps(c, MVR_K_GLUE, 2.0 * (double)nt);
  // ordered compaction: list = sorted positions of the matched targets, in Hilbert order
  hipcub::CountingInputIterator<uint32_t> it(0);
  size_t bytes = 0;
  MVR_HIP_TRY(c, hipcub::DeviceSelect::Flagged(nullptr, bytes, it, flags, list, count, (int)nt, c->stream));
  if (int rc = ensure_cub(c, bytes)) return rc;
  MVR_HIP_TRY(c, hipcub::DeviceSelect::Flagged(c->cub_tmp, bytes, it, flags, list, count, (int)nt, c->stream));
  const size_t cap = std::min(q_count, nt);
  hipLaunchKernelGGL(scatter_slot_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, c->stream, list, count, cap, slot);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_flag_matched_batch(Ctx *c, const GlueBatch &b, int n_pairs)
{
  size_t qmax = 0; double work = 0.0;
  for (int k = 0; k < n_pairs; ++k) { qmax = std::max(qmax, (size_t)b.p[k].q_count); work += 16.0 * (double)b.p[k].q_count; }
  if (qmax == 0) return MVR_OK;
  ProfScope ps(c, MVR_K_GLUE, work);
  hipLaunchKernelGGL(flag_matched_batch_kernel, dim3((unsigned)((qmax + 255) / 256), (unsigned)n_pairs), dim3(256), 0, c->stream, b);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_seed_bounds(Ctx *c, const nnkey_t *keys, const uint32_t *qperm, size_t q_begin, size_t q_count, double max2,
                       const uint32_t *tinv, size_t nt, uint32_t *bound, uint32_t *seed_out, uint32_t *zero_word)
{
  if (q_count == 0 || nt == 0) return MVR_OK;
  ProfScope ps(c, MVR_K_GLUE, 16.0 * (double)q_count + 4.0 * (double)nt);
  MVR_HIP_TRY(c, hipMemsetAsync(bound, 0xFF, nt * sizeof(uint32_t), c->stream));
  hipLaunchKernelGGL(seed_bounds_kernel, dim3((unsigned)((q_count + 255) / 256)), dim3(256), 0, c->stream, keys, qperm, q_begin,
                     q_count, max2, tinv, bound, seed_out, zero_word);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_flag_matched(Ctx *c, const nnkey_t *keys, const uint32_t *qperm, size_t q_begin, size_t q_count, double max2,
                        const uint32_t *tinv, size_t nt, uint8_t *flags)
{
  if (q_count == 0 || nt == 0) return MVR_OK;
  ProfScope ps(c, MVR_K_GLUE, 16.0 * (double)q_count + 1.0 * (double)nt);
  MVR_HIP_TRY(c, hipMemsetAsync(flags, 0, nt, c->stream));
  hipLaunchKernelGGL(flag_matched_kernel, dim3((unsigned)((q_count + 255) / 256)), dim3(256), 0, c->stream, keys, qperm,
                     q_begin, q_count, max2, tinv, flags);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

}  // namespace mvr
