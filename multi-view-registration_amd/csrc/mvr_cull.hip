// csrc/mvr_cull.hip -- the exact culled 1-NN kernel (gfx950) over the index of mvr_index.hip.
//
// Same results as the brute-force kernel of mvr_nn.hip (bit-identical d2,
// lowest ORIGINAL index on ties) with O(N * k) instead of O(N^2) distance
// evaluations -- the GPU-native counterpart of the kd-tree the reference gets
// from PCL/FLANN (tree_->nearestKSearch inside icp.align, registrator.cpp:569,
// and inside determineReciprocalCorrespondences, :502/:649).
//
// One BLOCK = one set of 64*Q Hilbert-consecutive queries, held by W waves (template parameter: 1, 2 or 4);
// wave w owns the target tiles with tile mod W == w, so the serial chain of tiles a query set has to visit
// is cut W ways.  The waves share nothing but a per-query "best so far" in LDS (ds_min_u32 on the float
// bits, read without a barrier: it only ever decreases, a stale value merely prunes less) and meet once, at
// the end, to combine their partial results.  A lone launch is bound by its longest chain and likes W = 2;
// a launch that keeps the chip full (the fused pass: all scan pairs of a step, blockIdx.y = pair) does the
// same work with fewer, better filled passes at W = 1.
//
// Inside a wave the 64 lanes are 4 GROUPS of 16: every group holds the SAME 64*Q queries (4*Q per lane)
// and evaluates them against its own 64-point CELL staged in LDS.  One ds_read_b128 therefore feeds 4*Q
// distance evaluations per lane instead of Q: with one query per lane (the first version of this kernel)
// the LDS pipe was the busiest unit of the CU (in-kernel cycle stamps, tools/build_stamp.sh: 25k cycles
// per tile against 4.4k of VALU work).  The four groups read four different addresses; the group regions
// are 65 float4 apart so that they fall into disjoint LDS banks.
//
// Culling is three-level: super boxes (64 tiles) decide which blocks of tile boxes are loaded at all; a
// 256-point tile is looked at only if SOME query of the set can still find an equal-or-closer point inside
// the tile's box (exact per-query point/box test with a 1e-5 relative safety margin for the rounding of the
// box distance); the same test then runs on the boxes of the tile's four 64-point cells, one cell per lane
// group, and only the cells that pass are queued.  The wave evaluates four queued cells at a time, one per
// lane group, wherever they come from (tools/cull_model.py: 64-point boxes need 2.3x fewer evaluations than
// 256-point ones on the turntable pair); the next four are prefetched into registers meanwhile and
// re-validated against the new bounds before they are staged.  Every point that could win or tie is still
// evaluated: results stay exact.  The inner loop is the brute-force one (min tracking per 32-target
// sub-tile); the index is recovered once, at the very end, by a re-scan of the winning sub-tile that the 4
// lane groups share (8 points each).
//
// A query may come with a KNOWN bound (qbound): the reverse search of a matched target starts from the
// distance of the source point that matched it, so its first cells are not chosen blind.  Keys go to the
// query's original index, to its list / sorted position (compacted or flagged queries), or -- key_by_pos,
// the fused pass -- to its absolute sorted position, so that every consumer reads them coalesced.
// Compiled with -ffp-contract=off.
#include "mvr_internal.h"

namespace mvr {
namespace {

constexpr int kSub = 32;        // min-tracking sub-tile
constexpr int kGrpPitch = 65;   // float4 between the lane groups' LDS regions (64 + 1: disjoint banks)

template <bool FMA>
__device__ __forceinline__ float dist2(const float4 t, float qx, float qy, float qz)
{
  const float dx = t.x - qx, dy = t.y - qy, dz = t.z - qz;
  if (FMA) return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

__device__ __forceinline__ float box_dist2(const float qlo[3], const float qhi[3], const float4 lo, const float4 hi)
{
  const float dx = fmaxf(0.f, fmaxf(lo.x - qhi[0], qlo[0] - hi.x));
  const float dy = fmaxf(0.f, fmaxf(lo.y - qhi[1], qlo[1] - hi.y));
  const float dz = fmaxf(0.f, fmaxf(lo.z - qhi[2], qlo[2] - hi.z));
  return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ void wave_lds_sync()
{
  // LDS hand-off between lanes of ONE wave: DS ops of a wave complete in order;
  // the fences keep the compiler from moving accesses across this point.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// lowest original index among points [first, first + count) of the sorted target array at distance exactly `d`,
// packed with where it sits: (original index << 32) | sorted position (~0: none); the loads are independent (one L2
// round trip)
using winner_t = unsigned long long;
constexpr winner_t kNoWinner = ~0ull;
template <bool FMA, int COUNT>
__device__ __forceinline__ winner_t span_argmin(const float4 *__restrict__ ts, uint32_t nt, uint32_t first, float d,
                                                float qx, float qy, float qz)
{
  float4 p[COUNT];
#pragma unroll
  for (int k = 0; k < COUNT; ++k) p[k] = ts[min(first + (uint32_t)k, nt - 1u)];
  winner_t best = kNoWinner;
#pragma unroll
  for (int k = 0; k < COUNT; ++k)
    if (first + (uint32_t)k < nt && dist2<FMA>(p[k], qx, qy, qz) == d)
      best = min(best, ((winner_t)__float_as_uint(p[k].w) << 32) | (winner_t)(first + (uint32_t)k));
  return best;
}

// the rare tie path: a whole sub-tile, 2 points at a time (keeps the register footprint of the hot path)
template <bool FMA>
__device__ __forceinline__ winner_t sub_argmin(const float4 *__restrict__ ts, uint32_t nt, uint32_t sub, float d, float qx,
                                            float qy, float qz)
{
  winner_t best = kNoWinner;
#pragma unroll 1
  for (int k = 0; k < kSub; k += 2) best = min(best, span_argmin<FMA, 2>(ts, nt, sub * kSub + (uint32_t)k, d, qx, qy, qz));
  return best;
}

__device__ __forceinline__ float lane_value(float v, int lane_uniform)     // v of one lane, as a wave-uniform (SGPR) value
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_uniform));
}

#ifdef MVR_STAMP    // diagnostic build only (tools/build_stamp.sh): where does a wave's lifetime go?
#define MVR_CLK() __builtin_readcyclecounter()
#else
#define MVR_CLK() 0ull
#endif

template <bool FMA, int Q, int W>
__device__ __forceinline__ void
nn_cull_body(const float4 *__restrict__ qs, uint32_t q_begin, uint32_t q_count, const uint8_t *__restrict__ qflags,
               const uint32_t *__restrict__ qlist, const uint32_t *__restrict__ qcount, const float4 *__restrict__ ts, uint32_t nt,
               const float4 *__restrict__ tlo, const float4 *__restrict__ thi, const float4 *__restrict__ cbox,
               const float4 *__restrict__ sbox, uint32_t n_tiles, float cap2, nnkey_t *__restrict__ keys, uint32_t key_by_pos,
               const uint32_t *__restrict__ qbound, uint32_t seed_from_keys, uint32_t *__restrict__ mark, uint32_t set,
               unsigned long long *__restrict__ evals)
{
  constexpr int NQ = 4 * Q;      // queries per lane
  constexpr int NB = 64 * Q;     // queries per block
  static_assert(W == 1 || W == 2 || W == 4, "waves per query set");
  static_assert(4 * NB * sizeof(nnkey_t) <= 4 * kGrpPitch * sizeof(float4), "partials must fit the tile buffers");
  __shared__ float4 lds[W][4 * kGrpPitch];
  __shared__ unsigned sbest[NB];
  // readfirstlane tells the compiler the wave index is wave-uniform: everything
  // derived from it (tile ids, loop conditions) then lives in SGPRs / scalar branches
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = lane >> 4, l16 = lane & 15;
  const unsigned long long st_entry = MVR_CLK();
#ifdef MVR_TRACE      // diagnostic build: phase marks of wave 0 (100 MHz clock), first occurrence of each
  const unsigned long long tr_rt0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long trm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned trq = 0, tr_open = 0, tr_exp = 0;
  unsigned long long tr_t_open = 0, tr_t_exp = 0, tr_t_next = 0;
#define MVR_MARK(i) do { if (!trm[i]) trm[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define MVR_TIC() const unsigned long long tic_ = __builtin_amdgcn_s_memrealtime()
#define MVR_TOC(acc, cnt) do { acc += __builtin_amdgcn_s_memrealtime() - tic_; ++cnt; } while (0)
#else
#define MVR_MARK(i) do { } while (0)
#define MVR_TIC() do { } while (0)
#define MVR_TOC(acc, cnt) do { } while (0)
#endif
#ifdef MVR_STAMP
  const unsigned long long st_rt0 = __builtin_amdgcn_s_memrealtime();
  if (evals && set == 0 && threadIdx.x == 0) atomicExch(evals + kEvalRegion + 15, st_rt0);
#endif
  float4 *T = lds[wv];
  // Queries = sorted positions [q_begin, q_begin + q_count) of the query cloud.  With qflags only the positions
  // whose flag is set are real queries -- the reverse search runs over the matched targets IN PLACE, no compaction:
  // matched targets cluster, so most blocks are either well filled or leave at once.  Key slot: the query's original
  // index (w) without flags, its sorted position with flags.  With qlist the queries are the sorted positions
  // qlist[p], p < *qcount (compacted matched targets: the form used when the target is much larger than the set of
  // queries, e.g. the merged target of the sequential mode); key slot = p.
  const uint32_t nq = qlist ? min(*qcount, q_count) : q_count;
  const uint32_t b_begin = set * NB;
  if (b_begin >= nq) return;                      // block-uniform

  // This wave's own tiles (own tile i <-> tile W i + wv) are looked at in ballot BLOCKS of 64; block k lies
  // inside the super boxes W k .. W k + W - 1 (one box per 64 consecutive tiles of the Hilbert order).  Lane l
  // keeps the box of block sbase + l, so one load instruction decides which blocks are worth opening at all:
  // a query set reads a handful of the tile boxes instead of all of them.  First chunk requested here, early.
  const uint32_t n_own = (n_tiles > (uint32_t)wv) ? (n_tiles - (uint32_t)wv + (uint32_t)W - 1u) / (uint32_t)W : 0u;
  const uint32_t n_blk = (n_own + 63u) / 64u, n_super = (n_tiles + 63u) / 64u;
  float sblo[3], sbhi[3];
  auto load_block_boxes = [&](uint32_t sbase) {
    sblo[0] = sblo[1] = sblo[2] = 3.0e38f; sbhi[0] = sbhi[1] = sbhi[2] = -3.0e38f;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const uint32_t sb = (uint32_t)W * (sbase + lane) + (uint32_t)j;
      if (sbase + lane < n_blk && sb < n_super) {
        const float4 a = sbox[2 * (size_t)sb], b = sbox[2 * (size_t)sb + 1];
        sblo[0] = fminf(sblo[0], a.x); sblo[1] = fminf(sblo[1], a.y); sblo[2] = fminf(sblo[2], a.z);
        sbhi[0] = fmaxf(sbhi[0], b.x); sbhi[1] = fmaxf(sbhi[1], b.y); sbhi[2] = fmaxf(sbhi[2], b.z);
      }
    }
  };
  load_block_boxes(0);

  // query q of this lane = position b_begin + 16 q + l16 (the same in all 4 groups); slots that hold no real
  // query (past the end, or not flagged) repeat a real one of the block (a duplicate changes no bound)
  float qx[NQ], qy[NQ], qz[NQ], best[NQ];
  uint32_t bsub[NQ];
  bool real[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const uint32_t pos = b_begin + q * 16 + l16;
    real[q] = pos < nq && (!qflags || qflags[q_begin + pos] != 0);
    const uint32_t at = pos < nq ? pos : b_begin;
    const float4 p = qs[qlist ? qlist[at] : (q_begin + at)];
    qx[q] = p.x; qy[q] = p.y; qz[q] = p.z;
    best[q] = __builtin_inff(); bsub[q] = kNone;
  }
  uint32_t nvalid = NB;
  if (qflags || b_begin + NB > nq) {             // block-uniform
    float rx = 0.f, ry = 0.f, rz = 0.f;
    bool have = false;
    nvalid = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const unsigned long long m = __ballot(real[q]) & 0xFFFFull;       // every lane group holds the same slots
      nvalid += (uint32_t)__popcll(m);
      if (!have && m) {
        const int l = __ffsll((long long)m) - 1;
        rx = lane_value(qx[q], l); ry = lane_value(qy[q], l); rz = lane_value(qz[q], l);
        have = true;
      }
    }
    if (!have) return;                            // nothing to search here: all waves of the block agree, before the barrier
#pragma unroll
    for (int q = 0; q < NQ; ++q) if (!real[q]) { qx[q] = rx; qy[q] = ry; qz[q] = rz; }
  }
  float qlo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, qhi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    qlo[0] = fminf(qlo[0], qx[q]); qlo[1] = fminf(qlo[1], qy[q]); qlo[2] = fminf(qlo[2], qz[q]);
    qhi[0] = fmaxf(qhi[0], qx[q]); qhi[1] = fmaxf(qhi[1], qy[q]); qhi[2] = fmaxf(qhi[2], qz[q]);
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1)
    for (int k = 0; k < 3; ++k) {
      qlo[k] = fminf(qlo[k], __shfl_xor(qlo[k], o, 64));
      qhi[k] = fmaxf(qhi[k], __shfl_xor(qhi[k], o, 64));
    }
#pragma unroll
  for (int k = 0; k < 3; ++k) {       // the set's box is wave-uniform: keep it in SGPRs
    qlo[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(qlo[k])));
    qhi[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(qhi[k])));
  }
  // the per-query bounds start at +inf, or at a distance the caller KNOWS a point within (qbound: the reverse
  // search of a matched target starts from the distance of the source point that matched it -- no blind first cells)
  // ... or at the distance of the point this very query matched LAST time (seed_from_keys: `keys` still holds the
  // previous result of the same query set against the same target set, low word = the match's sorted position): both
  // clouds have moved since, but that point is still a point of the target, so its distance NOW is an exact, inclusive
  // bound -- the search of an ICP step starts where the previous step ended instead of blind.
  for (int i = threadIdx.x; i < NB; i += 64 * W) {
    uint32_t seed = 0x7F800000u;
    const uint32_t pos = b_begin + (uint32_t)i, at = pos < nq ? pos : b_begin;
    if (qbound) {
      const uint32_t v = qbound[qlist ? qlist[at] : (q_begin + at)];
      if (v < seed) seed = v;       // (~0 = no bound)
    }
    if (seed_from_keys) {
      const uint32_t prev = (uint32_t)keys[q_begin + at];
      if (prev < nt) {
        const float4 tp = ts[prev], qp = qs[q_begin + at];
        const uint32_t v = __float_as_uint(dist2<FMA>(tp, qp.x, qp.y, qp.z));
        if (v < seed) seed = v;
      }
    }
    sbest[i] = seed;
  }
  __syncthreads();
  MVR_MARK(0);

  float U = cap2;                 // wave-uniform: no query of this set needs a point farther than U
  bool U_stale = qbound != nullptr || seed_from_keys != 0;       // seeded bounds: U is their maximum, derived when first needed
  uint32_t cells_done = 0, tiles_tested = 0;

  auto shared_bound = [&](int j) {
    return fminf(__uint_as_float(__atomic_load_n(&sbest[j * 64 + lane], __ATOMIC_RELAXED)), cap2);
  };

  // register prefetch buffer: the NEXT four cells' points travel from L2 while the
  // current four are being evaluated out of LDS (coordinates only: the index, w, is
  // not needed before the final re-scan).  A missing cell / point is +inf: never a minimum.
  float pre[4][3];
  auto fetch = [&](const uint32_t (&cells)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t j = cells[r] * 64u + lane;
      pre[r][0] = pre[r][1] = pre[r][2] = __builtin_inff();
      if (cells[r] != kNone && j < nt) { const float4 p = ts[j]; pre[r][0] = p.x; pre[r][1] = p.y; pre[r][2] = p.z; }
    }
  };
  auto stage = [&](const uint32_t (&cells)[4]) {      // cell r -> LDS region of lane group r (a dropped cell becomes +inf)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool live = cells[r] != kNone;
      T[r * kGrpPitch + lane] = live ? make_float4(pre[r][0], pre[r][1], pre[r][2], 0.f)
                                     : make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.f);
    }
    wave_lds_sync();
  };

  // evaluates the four cells currently staged in this wave's LDS buffer: lane group g takes cell g
  auto process = [&](const uint32_t (&cells)[4]) {
    const float4 *Tg = T + g * kGrpPitch;
    // this lane group's cell (bit masks: any select chain over the group number gets turned into a scratch-memory table)
    const uint32_t mycell = (cells[0] & (0u - (uint32_t)(g == 0))) | (cells[1] & (0u - (uint32_t)(g == 1))) |
                            (cells[2] & (0u - (uint32_t)(g == 2))) | (cells[3] & (0u - (uint32_t)(g == 3)));
#pragma unroll 1
    for (int s = 0; s < 64; s += kSub) {
      float m[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) m[q] = __builtin_inff();
#ifdef MVR_SKIP_LOOP          // counter experiments only: everything but the distance loop (results are wrong)
      for (int k = 0; k < 2; k += 2) {
#else
#pragma unroll 2
      for (int k = 0; k < kSub; k += 2) {
#endif
        const float4 a = Tg[s + k], b = Tg[s + k + 1];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const float da = dist2<FMA>(a, qx[q], qy[q], qz[q]);
          const float db = dist2<FMA>(b, qx[q], qy[q], qz[q]);
          m[q] = __builtin_fminf(__builtin_fminf(m[q], da), db);
        }
      }
      const uint32_t sub = mycell * (64 / kSub) + (uint32_t)s / kSub;      // = first sorted index / kSub
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (m[q] < best[q]) { best[q] = m[q]; bsub[q] = sub; }
        else if (m[q] == best[q] && bsub[q] != kNone && m[q] < 3.0e38f) {
          // exact tie between two sub-tiles (visited in any order): keep the one
          // holding the lowest original index.  Rare; only duplicates / symmetric data.
          const winner_t o1 = sub_argmin<FMA>(ts, nt, bsub[q], best[q], qx[q], qy[q], qz[q]);
          const winner_t o2 = sub_argmin<FMA>(ts, nt, sub, best[q], qx[q], qy[q], qz[q]);
          if (o2 < o1) bsub[q] = sub;
        }
      }
    }
    wave_lds_sync();     // all lanes done reading before the buffer is overwritten
#pragma unroll
    for (int r = 0; r < 4; ++r) cells_done += (cells[r] != kNone) ? 1u : 0u;
    // publish this wave's bests; U (their set-wide maximum) is re-derived only when a ballot needs it
#pragma unroll
    for (int q = 0; q < NQ; ++q) atomicMin(&sbest[q * 16 + l16], __float_as_uint(best[q]));
    U_stale = true;
  };
  auto refresh_U = [&]() {       // wave max-reduction of the set-wide bounds
    if (!U_stale) return;
    float w = 0.f;
#pragma unroll
    for (int j = 0; j < Q; ++j) w = fmaxf(w, shared_bound(j));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w = fmaxf(w, __shfl_xor(w, o, 64));
    U = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fminf(cap2, w))));
    U_stale = false;
  };

  // Exact per-query test of FOUR boxes at once, one per lane group: every lane measures its 4 Q queries
  // against its group's box (the 16 lanes of a group hold all 64 Q queries between them), one ballot
  // says which of the four boxes SOME query of the set can still find an equal-or-closer point in.
  // (Testing one box with one query per lane was a ~60-instruction dependent chain per box; thirty of
  // those per wave were a third of a wave's life -- tools/block_trace.py.)
  auto needed4 = [&](const float lo_x, const float lo_y, const float lo_z, const float hi_x, const float hi_y,
                     const float hi_z, unsigned *hot = nullptr) -> unsigned {
    bool need = false, inside = false;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float dx = fmaxf(0.f, fmaxf(lo_x - qx[q], qx[q] - hi_x));
      const float dy = fmaxf(0.f, fmaxf(lo_y - qy[q], qy[q] - hi_y));
      const float dz = fmaxf(0.f, fmaxf(lo_z - qz[q], qz[q] - hi_z));
      const float pb = dx * dx + dy * dy + dz * dz;
      const float bound = fminf(__uint_as_float(__atomic_load_n(&sbest[q * 16 + l16], __ATOMIC_RELAXED)), cap2);
      need |= (pb * 0.99999f <= bound);
      inside |= (pb == 0.f);
    }
    ++tiles_tested;
    if (hot) {           // boxes that CONTAIN a query of the set: the cells most likely to settle its bound
      const unsigned long long h = __ballot(inside);
      *hot = ((h & 0xFFFFull) ? 1u : 0u) | ((h & 0xFFFF0000ull) ? 2u : 0u) | ((h & 0xFFFF00000000ull) ? 4u : 0u) |
             ((h & 0xFFFF000000000000ull) ? 8u : 0u);
    }
    const unsigned long long b = __ballot(need);
    return ((b & 0xFFFFull) ? 1u : 0u) | ((b & 0xFFFF0000ull) ? 2u : 0u) | ((b & 0xFFFF00000000ull) ? 4u : 0u) |
           ((b & 0xFFFF000000000000ull) ? 8u : 0u);
  };

  // candidate stream over this wave's own tiles: round 0 = tiles whose box overlaps the
  // query box, round 1 = the rest within U.  Distances are NaN where there is no block / tile.
  auto block_dist = [&](uint32_t sbase) {
    return (sbase + lane < n_blk) ? box_dist2(qlo, qhi, make_float4(sblo[0], sblo[1], sblo[2], 0.f), make_float4(sbhi[0], sbhi[1], sbhi[2], 0.f))
                                  : __builtin_nanf("");
  };
  int round = 0;
  uint32_t sbase = 0;                 // first block of the chunk of 64 blocks held in sb_lb
  bool chunk_open = false;
  float sb_lb = block_dist(0);        // set box <-> block box
  unsigned long long sb_mask = 0ull;  // blocks of the chunk still to open in this round
  uint32_t cbase = 0;                 // first own tile of the open block
  unsigned long long mask = 0ull, mask0 = 0ull;      // its candidate tiles: remaining / as balloted
  float lb = 0.f;
  float4 blo = make_float4(0.f, 0.f, 0.f, 0.f), bhi = blo;     // this lane's tile box of the open block
  float4 rec = blo;                   // cell boxes of the block's first 8 candidates: lane 8 c + k = float4 k of candidate c
  int last_b = 0;                     // bit of the tile last returned
  auto advance = [&]() -> uint32_t {
    if (n_blk == 0) return kNone;
    if (round == 1) refresh_U();
    for (;;) {
      while (mask == 0ull) {
        while (sb_mask == 0ull) {
          if (chunk_open) {           // this chunk is done for this round
            sbase += 64;
            if (sbase >= n_blk) {
              if (round == 1) return kNone;
              round = 1; sbase = 0;
              refresh_U();
            }
            if (n_blk > 64) { load_block_boxes(sbase); sb_lb = block_dist(sbase); }     // a single chunk stays in registers
          }
          chunk_open = true;
          // round 1 re-opens overlapping blocks too: they can hold tiles that do not overlap
          sb_mask = __ballot((round == 0) ? (sb_lb == 0.f) : (sb_lb * 0.99999f <= U));
        }
        const int k = __ffsll((long long)sb_mask) - 1;
        sb_mask &= sb_mask - 1;
        if (round == 1 && lane_value(sb_lb, k) * 0.99999f > U) continue;   // U shrank since the ballot
        MVR_TIC();
        cbase = (sbase + (uint32_t)k) * 64u;
        const uint32_t i = cbase + lane;
        const uint32_t t = (uint32_t)W * i + (uint32_t)wv;
        lb = __builtin_nanf("");
        if (i < n_own) { blo = tlo[t]; bhi = thi[t]; lb = box_dist2(qlo, qhi, blo, bhi); }
        mask = mask0 = __ballot((round == 0) ? (lb == 0.f) : (lb > 0.f && lb * 0.99999f <= U));
        // one load fetches the cell boxes of up to 8 candidate tiles: lane l takes float4 (l & 7) of candidate l >> 3
        uint32_t mine = kNone;
        unsigned long long mm = mask0;
#pragma unroll
        for (int cnd = 0; cnd < 8; ++cnd) {
          if (mm == 0ull) break;
          const int bit = __ffsll((long long)mm) - 1;
          mm &= mm - 1;
          if ((lane >> 3) == cnd) mine = (uint32_t)W * (cbase + (uint32_t)bit) + (uint32_t)wv;
        }
        if (mine != kNone) rec = cbox[(size_t)mine * 8 + (lane & 7)];
        MVR_TOC(tr_t_open, tr_open);
      }
      const int b = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      if (round == 1 && lane_value(lb, b) * 0.99999f > U) continue;     // U shrank since the ballot
      last_b = b;
      return (uint32_t)W * (cbase + (uint32_t)b) + (uint32_t)wv;
    }
  };

  // queue of needed cells, a deque: entry k lives in lane k & 63 of these registers (written under a lane
  // compare, read back with v_readlane); cells whose box CONTAINS a query of the set enter at the front, the
  // others at the back (measured on the ring step: 5.5 % fewer evaluations, launches 2.5 % shorter; looking a
  // few tiles ahead for such a cell before the first pop was tried too: 0.6 % fewer evaluations, 1 % slower).
  // It never holds more than 3 + 4 cells.
  uint32_t pq_cell = 0;
  float pq_box[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  uint32_t pq_head = 0, pq_tail = 0;
  // the four cells of a candidate tile (one 128-byte record {lo, hi} x 4): lane group g tests cell g
  auto expand_tile = [&](uint32_t tile) {
    MVR_TIC();
    const int cnd = __popcll(mask0 & ((1ull << last_b) - 1ull));         // rank among the block's candidates
    float4 clo, chi;
    if (cnd < 8) {                       // prefetched: float4 k of candidate c sits in lane 8 c + k
      const int src = 8 * cnd + 2 * g;
      clo = make_float4(__shfl(rec.x, src, 64), __shfl(rec.y, src, 64), __shfl(rec.z, src, 64), 0.f);
      chi = make_float4(__shfl(rec.x, src + 1, 64), __shfl(rec.y, src + 1, 64), __shfl(rec.z, src + 1, 64), 0.f);
    } else {                             // beyond the prefetched eight: read it now
      clo = cbox[(size_t)tile * 8 + 2 * g]; chi = cbox[(size_t)tile * 8 + 2 * g + 1];
    }
#ifndef MVR_NO_HOT_FIRST
    unsigned hot = 0;
    const unsigned nd = needed4(clo.x, clo.y, clo.z, chi.x, chi.y, chi.z, &hot);
#else
    const unsigned hot = 0;
    const unsigned nd = needed4(clo.x, clo.y, clo.z, chi.x, chi.y, chi.z);
#endif
#pragma unroll
    for (int cidx = 0; cidx < 4; ++cidx) {
      if ((nd >> cidx) & 1u) {
        // a cell that contains a query goes to the FRONT of the queue: evaluated before the merely near ones queued
        // earlier, it tightens the bounds they will be re-validated against
        const bool front = ((hot >> cidx) & 1u) != 0;
        if (front) --pq_head;
        const int at = (int)((front ? pq_head : pq_tail) & 63u), from = 16 * cidx;          // any lane of group cidx holds that cell's box
        pq_cell = (lane == at) ? tile * 4u + (uint32_t)cidx : pq_cell;
        const float bx[6] = {lane_value(clo.x, from), lane_value(clo.y, from), lane_value(clo.z, from),
                             lane_value(chi.x, from), lane_value(chi.y, from), lane_value(chi.z, from)};
#pragma unroll
        for (int k = 0; k < 6; ++k) pq_box[k] = (lane == at) ? bx[k] : pq_box[k];
        if (!front) ++pq_tail;
      }
    }
    MVR_TOC(tr_t_exp, tr_exp);
  };

  // up to four queued cells (fewer only when the candidate stream is exhausted); 0 = done
  auto next_quad = [&](uint32_t (&cells)[4], float (&boxes)[4][6]) -> uint32_t {
#ifdef MVR_TRACE
    const unsigned long long tnq = __builtin_amdgcn_s_memrealtime();
#endif
    while (pq_tail - pq_head < 4u) {
      const uint32_t tile = advance();
      if (tile == kNone) break;
      expand_tile(tile);
    }
    const uint32_t n = min(pq_tail - pq_head, 4u);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      cells[r] = kNone;
      if ((uint32_t)r < n) {
        const int at = (int)((pq_head + (uint32_t)r) & 63u);
        cells[r] = (uint32_t)__builtin_amdgcn_readlane((int)pq_cell, at);
#pragma unroll
        for (int k = 0; k < 6; ++k) boxes[r][k] = lane_value(pq_box[k], at);
      }
    }
    pq_head += n;
#ifdef MVR_TRACE
    tr_t_next += __builtin_amdgcn_s_memrealtime() - tnq;
#endif
    return n;
  };

  unsigned long long st_stage = 0, st_adv = 0, st_proc = 0, st_reval = 0;
  const unsigned long long st_begin = MVR_CLK();
  uint32_t cur[4], nxt[4];
  float cbx[4][6], nbx[4][6];
  // Wave priority: everything but the distance loop -- candidate search, queue bookkeeping, re-validation, the merge --
  // is a chain of dependent ballots, read-lanes and box loads that is a third of the instructions and half of the
  // time.  Those phases run at raised priority, so a wave in them issues ahead of the waves of its SIMD
  // that are in the distance loop (which have instruction-level parallelism to spare).  Measured on the ring step:
  // fused launches 0.396 -> 0.374 ms; priority 1 and 3 do the same.
#ifndef MVR_SETPRIO
#define MVR_SETPRIO 3
#endif
  __builtin_amdgcn_s_setprio(MVR_SETPRIO);
  uint32_t n_cur = next_quad(cur, cbx);
  MVR_MARK(1);
  if (n_cur) fetch(cur);
  while (n_cur) {
    const unsigned long long t0 = MVR_CLK();
    uint32_t n_nxt = next_quad(nxt, nbx);     // chosen with the bounds as they are NOW, while the current four land
    const unsigned long long t1 = MVR_CLK();
    stage(cur);                               // registers -> LDS (waits for the prefetch)
    MVR_MARK(2);
#ifdef MVR_TRACE
    ++trq;
#endif
    if (n_nxt) fetch(nxt);                    // in flight during the evaluation below
    const unsigned long long t2 = MVR_CLK();
    MVR_MARK(3);
    __builtin_amdgcn_s_setprio(0);
    process(cur);
    __builtin_amdgcn_s_setprio(MVR_SETPRIO);
    MVR_MARK(4);
    const unsigned long long t3 = MVR_CLK();
    // the bounds have shrunk: re-validate the prefetched cells (a cell that is no longer needed is dropped,
    // its lane group idles; if none is left the next four are requested)
    while (n_nxt) {
      float bxg[6];                     // lane group g re-tests cell g of the prefetched four
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        bxg[k] = nbx[0][k];
#pragma unroll
        for (int r = 1; r < 4; ++r) bxg[k] = (g == r) ? nbx[r][k] : bxg[k];
      }
      const unsigned nd = needed4(bxg[0], bxg[1], bxg[2], bxg[3], bxg[4], bxg[5]);
      uint32_t live = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (!((nd >> r) & 1u)) nxt[r] = kNone;        // (an empty slot has a stale box: it stays empty either way)
        live += (nxt[r] != kNone) ? 1u : 0u;
      }
      if (live) break;
      n_nxt = next_quad(nxt, nbx);
      if (n_nxt) fetch(nxt);
    }
    st_adv += t1 - t0; st_stage += t2 - t1; st_proc += t3 - t2; st_reval += MVR_CLK() - t3;
#pragma unroll
    for (int r = 0; r < 4; ++r) cur[r] = nxt[r];
    n_cur = n_nxt;
  }
  (void)cbx;
  const unsigned long long st_loopend = MVR_CLK();
  MVR_MARK(5);

#ifdef MVR_SKIP_MERGE         // counter experiments only: no merge, no key output (results are wrong)
  if (cells_done != 0xFFFFFFFFu) return;
#endif
  // ---- combine: 4 W partial results per query (W waves x 4 lane groups) meet in LDS (over the tile buffers).
  // The waves share out the 4 Q lane columns (16 queries each): wave wv finishes columns wv, wv + W, ...
  // The queries are re-read before the barrier (their latency hides behind the wait for the slowest wave).
  constexpr int NC = 4 * Q / W;                                 // columns per wave
  // The finishing lane needs its column's query again.  One wave per set (W == 1): column j of a lane IS its own query
  // j, still in registers (qx, qy, qz) -- only its original index is fetched, and only by the lanes that store a key
  // under it.  (Re-reading the queries ahead of the barrier, as the W > 1 path does to hide their latency behind the
  // wait for the slowest wave, made the compiler park 64 bytes per lane in scratch: 150 MB of spill traffic per fused
  // launch, measured with the WRITE_SIZE counter.)
  float fqx[NC], fqy[NC], fqz[NC];
  uint32_t fqw[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    if (W == 1) { fqx[j] = qx[j]; fqy[j] = qy[j]; fqz[j] = qz[j]; fqw[j] = kNone; }
    else {
      const uint32_t pos = b_begin + (uint32_t)((wv + j * W) * 16 + l16), rpos = pos < nq ? pos : b_begin;
      const float4 f = qs[qlist ? qlist[rpos] : (q_begin + rpos)];         // this query's coordinates (and its original index)
      fqx[j] = f.x; fqy[j] = f.y; fqz[j] = f.z; fqw[j] = __float_as_uint(f.w);
    }
  }
  // W == 1: the four partial results of a query sit in the four lane groups of this one wave, same register, lanes 16
  // apart: two cross-lane steps combine them -- no LDS round trip, no barrier, nothing parked in scratch meanwhile.
  // W > 1: the waves meet in LDS (over the tile buffers).
  nnkey_t *part = reinterpret_cast<nnkey_t *>(&lds[0][0]);      // W > 1 only: [4 W][NB]: (d2 bits, sub-tile)
  nnkey_t pm[NC];
  bool tie[NC];      // another partial reached the same distance in a different sub-tile (rare: duplicates / symmetric data)
  if constexpr (W == 1) {
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const nnkey_t mine = ((nnkey_t)__float_as_uint(best[j]) << 32) | bsub[j];
      nnkey_t m = min(mine, (nnkey_t)__shfl_xor(mine, 16, 64));
      m = min(m, (nnkey_t)__shfl_xor(m, 32, 64));
      pm[j] = m;
      int other = (uint32_t)(mine >> 32) == (uint32_t)(m >> 32) && (uint32_t)mine != (uint32_t)m && (uint32_t)mine != kNone;
      other |= __shfl_xor(other, 16, 64);
      other |= __shfl_xor(other, 32, 64);
      tie[j] = other != 0;
    }
  } else {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      part[(wv * 4 + g) * NB + q * 16 + l16] = ((nnkey_t)__float_as_uint(best[q]) << 32) | bsub[q];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int i = (wv + j * W) * 16 + l16;
      nnkey_t pk[4 * W];
      pm[j] = kKeyInit;
#pragma unroll
      for (int p = 0; p < 4 * W; ++p) { pk[p] = part[p * NB + i]; pm[j] = min(pm[j], pk[p]); }
      tie[j] = false;
#pragma unroll
      for (int p = 0; p < 4 * W; ++p)
        tie[j] |= (uint32_t)(pk[p] >> 32) == (uint32_t)(pm[j] >> 32) && (uint32_t)pk[p] != (uint32_t)pm[j] && (uint32_t)pk[p] != kNone;
    }
  }
  MVR_MARK(6);
  // the index (lowest original index at distance == best): one re-scan of the winning sub-tile, 8 points
  // per lane group (MVR_CULL_CB columns' loads in flight at a time: 2 needs 64 registers and spills; measured equal)
#ifndef MVR_CULL_CB
#define MVR_CULL_CB 1
#endif
  constexpr int CB = NC < MVR_CULL_CB ? NC : MVR_CULL_CB;
#pragma unroll
  for (int j0 = 0; j0 < NC; j0 += CB) {
    float4 cand[CB][kSub / 4];
#pragma unroll
    for (int jj = 0; jj < CB; ++jj) {
      const uint32_t first = (uint32_t)pm[j0 + jj] * kSub + g * (kSub / 4);     // (clamped below: harmless when there is no winner)
#pragma unroll
      for (int k = 0; k < kSub / 4; ++k) cand[jj][k] = ts[min(first + (uint32_t)k, nt - 1u)];
    }
#pragma unroll
    for (int jj = 0; jj < CB; ++jj) {
      const int j = j0 + jj;
      const int i = (wv + j * W) * 16 + l16;
      const uint32_t pos = b_begin + (uint32_t)i;
      const uint32_t dbits = (uint32_t)(pm[j] >> 32), sub = (uint32_t)pm[j];
      const float d = __uint_as_float(dbits);
      const bool found = sub != kNone && d <= cap2;
      winner_t ow = kNoWinner;           // (original index << 32) | sorted position: the minimum is the lowest ORIGINAL index
      if (found) {
        const uint32_t first = sub * kSub + g * (kSub / 4);
#pragma unroll
        for (int k = 0; k < kSub / 4; ++k)
          if (first + (uint32_t)k < nt && dist2<FMA>(cand[jj][k], fqx[j], fqy[j], fqz[j]) == d)
            ow = min(ow, ((winner_t)__float_as_uint(cand[jj][k].w) << 32) | (winner_t)(first + (uint32_t)k));
        if (tie[j]) {          // (wave-uniform for W == 1: every lane group of the column sees the same flag)
#pragma unroll 1
          for (int p = 0; p < 4 * W; ++p) {
            nnkey_t k;
            if constexpr (W == 1) k = (nnkey_t)__shfl(((nnkey_t)__float_as_uint(best[j]) << 32) | bsub[j], p * 16 + l16, 64);
            else k = part[p * NB + i];
            if ((uint32_t)(k >> 32) == dbits && (uint32_t)k != sub && (uint32_t)k != kNone)
              ow = min(ow, span_argmin<FMA, kSub / 4>(ts, nt, (uint32_t)k * kSub + g * (kSub / 4), d, fqx[j], fqy[j], fqz[j]));
          }
        }
      }
      ow = min(ow, (winner_t)__shfl_xor(ow, 16, 64));
      ow = min(ow, (winner_t)__shfl_xor(ow, 32, 64));
      if (g == 0 && pos < nq && (!qflags || qflags[q_begin + pos] != 0)) {
        // key slot: sorted / list position, or original index (or, on request, the absolute sorted position).
        // Low word: the winner's original index -- or, key_by_pos, its SORTED POSITION: the consumers of the fused
        // pass (start bounds, reciprocal filter, moments) work in sorted space, so no original-index -> position
        // gather is left between the stages.
        uint32_t ord = (qlist || (qflags && !key_by_pos)) ? pos : q_begin + pos;      // (flagged queries of a fused pass keep their absolute slots)
        if (!(qflags || qlist) && !key_by_pos) ord = (W == 1) ? __float_as_uint(qs[q_begin + pos].w) : fqw[j];
        const uint32_t low = key_by_pos ? (uint32_t)ow : (uint32_t)(ow >> 32);
        keys[ord] = (found && ow != kNoWinner) ? (((nnkey_t)dbits << 32) | low) : kKeyInit;
        // the matched target's reverse search may start from this distance (the fused pass: `low` is its position)
        if (mark && found && ow != kNoWinner) __hip_atomic_store(&mark[low], dbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // one column's candidates at a time: with the loads of all columns hoisted to the top the kernel no longer fits its
    // 96 registers and parks 32 bytes per lane in scratch (86 MB written per fused launch for 19 MB of keys)
    __builtin_amdgcn_sched_barrier(0);
  }
  MVR_MARK(7);
  if (evals && lane == 0) {
    const unsigned long long e = (unsigned long long)cells_done * 64ull * nvalid;
    // sharded counters: one 128-byte line per shard, or thousands of waves serialise on one address
    unsigned long long *a = evals + (size_t)(set & (kEvalShards - 1)) * kEvalStride;
    unsigned long long *b = a + kEvalRegion;
    atomicAdd(a, e);              // this launch (profiling)
    atomicAdd(b, e);              // running total (mvr_icp_stats.evals)
    atomicMax(b + 1, (unsigned long long)cells_done);     // diagnostics: heaviest wave (cells evaluated)
    atomicMax(b + 2, (unsigned long long)tiles_tested);
#ifdef MVR_TRACE
    if (set < kTraceBlocks) {     // block record: start (first wave), end (last wave), cells (max over waves), XCC id
      unsigned long long *tr = evals + 2 * kEvalRegion + kTraceRec * (size_t)set;
      if (wv == 0) {
        tr[0] = tr_rt0;
        for (int k = 0; k < 8; ++k) tr[4 + k] = trm[k];
        tr[12] = trq; tr[13] = tiles_tested; tr[14] = ((unsigned long long)tr_open << 32) | tr_exp;
        tr[15] = (tr_t_open << 40) | (tr_t_exp << 20) | tr_t_next;
      }
      atomicMax(tr + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
      atomicMax(tr + 2, (unsigned long long)cells_done);
      unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      tr[3] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
#ifdef MVR_STAMP
    atomicAdd(b + 4, st_stage); atomicAdd(b + 5, st_adv); atomicAdd(b + 6, st_proc); atomicAdd(b + 7, st_reval);
    atomicAdd(b + 8, st_loopend - st_begin); atomicAdd(b + 9, 1ull); atomicAdd(b + 10, st_begin - st_entry);
    atomicAdd(b + 11, MVR_CLK() - st_loopend); atomicAdd(b + 12, MVR_CLK() - st_entry);
    atomicAdd(b + 13, __builtin_amdgcn_s_memrealtime() - st_rt0);
    {   // lifetime by number of tiles evaluated: shard k slot 3 = sum of cycles, shard 32 + k slot 3 = waves
      const unsigned long long k = min((unsigned long long)cells_done, 31ull);
      atomicAdd(evals + kEvalRegion + k * kEvalStride + 3, MVR_CLK() - st_entry);
      atomicAdd(evals + kEvalRegion + (32 + k) * kEvalStride + 3, 1ull);
    }
    // histogram of wave end times (5 us bins since block 0 started): shard k, slot 14 = bin k; shard 0 slot 15 = base
    unsigned long long *h = evals + kEvalRegion;
    const unsigned long long base = atomicAdd(h + 15, 0ull);
    if (base != 0 && set != 0) {
      const unsigned long long bin = min((__builtin_amdgcn_s_memrealtime() - base) / 500ull, (unsigned long long)(kEvalShards - 1));
      atomicAdd(h + bin * kEvalStride + 14, 1ull);
    }
#endif
  }
  (void)st_entry; (void)st_stage; (void)st_adv; (void)st_proc; (void)st_reval; (void)st_begin; (void)st_loopend;
}

// One kernel per (queries per set, waves per set), compiled for 4 waves per SIMD (<= 128 VGPRs; it uses 106).
// Round 1 ran it at 5 (<= 96 VGPRs) with a few registers parked in scratch -- which looked free in the timings and was
// not in the counters: every wave writes its 24-72 spilled bytes per lane out and reads them back, 60-190 MB of
// WRITE_SIZE per fused launch for 29 MB of keys and bounds.  At 4 waves nothing spills, the launch moves 151 MB
// instead of 276-448 (tools/measure_traffic.sh) and the ring step takes the same time (0.883 vs 0.884 ms).
#ifndef MVR_CULL_WAVES
#define MVR_CULL_WAVES 4
#endif
#ifdef MVR_CULL_WAVES_MAX          // experiments: cap the residency as well
#define MVR_CULL_WAVES_ATTR MVR_CULL_WAVES, MVR_CULL_WAVES_MAX
#else
#define MVR_CULL_WAVES_ATTR MVR_CULL_WAVES
#endif
// ONE launch runs the searches of all scan pairs of a global iteration (the arguments of up to kBatchPairs pairs travel
// by value): the chip is filled by the union of their query sets and the tail of one pair's search is covered by the
// others.  Block -> (pair, query set), XCD-aware: blocks b and b + 8 share an XCD (MI355X deals workgroups round-robin
// over its 8 XCDs; which XCD block 0 gets is not fixed and nothing here depends on it -- speed only), and every XCD has
// its own 4 MB L2.  With a plain (set, pair) grid every XCD walks through EVERY pair's target: 8 copies of each target
// (3.6 MB with its boxes) cross the fabric per launch -- 448 MB measured for 12 pairs against ~100 MB of distinct
// bytes.  Here a pair is cut into `slices` interleaved slices of its query sets (set mod slices; slices = 8 / gcd(pairs,
// 8), so that pairs * slices fills the 8 XCDs evenly) and slice u of the launch goes to the blocks with b mod 8 ==
// u mod 8, one slice after the other: an XCD works on one target at a time and a target is read by `slices` XCDs
// instead of 8.  Interleaved slices keep the XCDs balanced: heavy query sets are neighbours in the Hilbert order.
template <bool FMA, int Q, int W>
// (residency: the four-waves-per-set instantiation is the lone launch of an align -- a dozen waves per SIMD in all, bound by its
// chains: one more resident wave per SIMD measured 2.5 % faster there; the fused launches keep MVR_CULL_WAVES, beyond which they spill)
__global__ void __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(Q == 1 ? (W == 4 ? MVR_CULL_WAVES + 1 : MVR_CULL_WAVES) : 3)))
nn_cull_kernel(CullBatch batch, XcdMap map, unsigned long long *__restrict__ evals)
{
  uint32_t pair = 0, set = 0;
  if (!xcd_map_block(map, blockIdx.x, &pair, &set)) return;          // block-uniform
  pair = (uint32_t)__builtin_amdgcn_readfirstlane((int)pair); set = (uint32_t)__builtin_amdgcn_readfirstlane((int)set);
  const CullPair &a = batch.p[pair];
  nn_cull_body<FMA, Q, W>(a.qs, a.q_begin, a.q_count, a.qflags, a.qlist, a.qcount, a.ts, a.nt, a.tlo, a.thi, a.cbox, a.sbox, a.n_tiles, batch.cap2,
                          a.keys, a.key_by_pos, a.qbound, a.seed_from_keys, a.key_by_pos ? a.mark : nullptr, set, evals);
}

// The same search for the query sets on a LIST (flagged queries are few and clustered: a block per set of the whole
// range would start tens of thousands of blocks that leave at once).  How many sets are listed is only known on the
// device: a fixed number of blocks per pair strides over its list, W waves per set (few sets: latency counts).
template <bool FMA, int W>
__global__ void __launch_bounds__(64 * W)
nn_cull_list_kernel(CullBatch batch, unsigned long long *__restrict__ evals)
{
  const CullPair &a = batch.p[blockIdx.y];
  if (!a.setcount) return;
  const uint32_t n = *a.setcount;
  for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
    const uint32_t set = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.setlist[i]);
    nn_cull_body<FMA, 1, W>(a.qs, a.q_begin, a.q_count, a.qflags, a.qlist, a.qcount, a.ts, a.nt, a.tlo, a.thi, a.cbox, a.sbox, a.n_tiles, batch.cap2,
                            a.keys, a.key_by_pos, a.qbound, a.seed_from_keys, a.key_by_pos ? a.mark : nullptr, set, evals);
    __syncthreads();           // the next set reuses the block's LDS
  }
}

}  // namespace

int launch_nn_cull_list_batch(Ctx *c, const CullPair *pairs, int n_pairs, float cap2, bool fma)
{
  if (int rc = flush_super_boxes(c)) return rc;      // (the super boxes a posing launch left to whoever reads them)
  for (int base = 0; base < n_pairs; base += kBatchPairs) {
    CullBatch batch;
    const int m = std::min(kBatchPairs, n_pairs - base);
    bool any = false;
    for (int k = 0; k < kBatchPairs; ++k) {
      batch.p[k] = k < m ? pairs[base + k] : CullPair{};
      if (k >= m || batch.p[k].nt == 0 || batch.p[k].q_count == 0 || !batch.p[k].setlist || !batch.p[k].qflags) batch.p[k].setcount = nullptr;
      any = any || batch.p[k].setcount != nullptr;
    }
    batch.cap2 = cap2;
    if (!any) continue;
    const int W = (c->cull_list_w == 1 || c->cull_list_w == 4) ? c->cull_list_w : 2;
    const dim3 grid((unsigned)std::max(1, c->n_cu * 8 / std::max(1, m)), (unsigned)m);
    const bool per_launch = c->prof && !c->prof_totals;
    if (per_launch) MVR_HIP_TRY(c, hipMemsetAsync(c->evals, 0, kEvalRegion * sizeof(unsigned long long), c->stream));
    ProfScope ps(c, MVR_K_NN_WIDE, per_launch ? c->evals : nullptr, (int)(kEvalRegion * sizeof(unsigned long long)), 1.0, 0.0, per_launch ? kEvalShards : 0);
#define MVR_LAUNCH_LIST(F, WW) hipLaunchKernelGGL((nn_cull_list_kernel<F, WW>), grid, dim3(64 * WW), 0, c->stream, batch, c->evals)
    if (fma) { if (W == 1) MVR_LAUNCH_LIST(true, 1); else if (W == 2) MVR_LAUNCH_LIST(true, 2); else MVR_LAUNCH_LIST(true, 4); }
    else     { if (W == 1) MVR_LAUNCH_LIST(false, 1); else if (W == 2) MVR_LAUNCH_LIST(false, 2); else MVR_LAUNCH_LIST(false, 4); }
#undef MVR_LAUNCH_LIST
    MVR_HIP_TRY(c, hipGetLastError());
  }
  return MVR_OK;
}

int launch_nn_cull_batch(Ctx *c, const CullPair *pairs, int n_pairs, float cap2, bool fma)
{
  if (int rc = flush_super_boxes(c)) return rc;      // (the super boxes a posing launch left to whoever reads them)
  for (int base = 0; base < n_pairs; base += kBatchPairs) {
    CullBatch batch;
    const int m = std::min(kBatchPairs, n_pairs - base);
    size_t qmax = 0; double upper = 0.0;
    for (int k = 0; k < kBatchPairs; ++k) {
      batch.p[k] = k < m ? pairs[base + k] : CullPair{};
      if (k < m && batch.p[k].nt == 0) batch.p[k].q_count = 0;
      qmax = std::max(qmax, (size_t)batch.p[k].q_count);
      upper += (double)batch.p[k].q_count * (double)batch.p[k].nt;
    }
    batch.cap2 = cap2;
    if (qmax == 0) continue;
    // measured (tools/cull_sweep.sh): the kernel is bound by its longest serial tile chain, so small query
    // sets win until there are far more sets than the chip holds at once
    int Q = (qmax <= (size_t)c->n_cu * 8 * 4 * 64) ? 1 : 2;
    if (c->cull_q == 1 || c->cull_q == 2) Q = c->cull_q;      // tuning override
    // waves sharing one query set: a lone search is bound by its longest chain of cells, two waves per set halve
    // it; a fused batch keeps the chip full whatever the chains are, and one wave per set then does the same work
    // with fewer, better filled passes (measured on the 12-pair ring: 1.52 ms/step with W = 1, 1.73 with 2, 2.27 with 4)
    XcdMap map;
    size_t sets = 0;
    for (int k = 0; k < kBatchPairs; ++k) {
      map.sets[k] = k < m ? (uint32_t)(((size_t)batch.p[k].q_count + 64 * Q - 1) / (64 * Q)) : 0u;
      sets += map.sets[k];
    }
    // ... and a launch of few sets (a lone align of 200k points: 3.1k sets, twelve waves per SIMD in all) is bound by that chain
    // outright: four waves per set (the sequential mode's align against a 2 M-point model: 0.45 -> 0.42 ms, 12 x 50k: 0.26 ->
    // 0.22; 4 x 200k equal)
    int W = sets >= (size_t)c->n_cu * 40 ? 1 : sets >= (size_t)c->n_cu * 16 ? 2 : 4;
    if (c->cull_w == 1 || c->cull_w == 2 || c->cull_w == 4) W = c->cull_w;      // tuning override
    map.n_pairs = (uint32_t)m;
    unsigned grid_blocks = 0;
    xcd_map_plan(map, c->cull_slices, &grid_blocks);          // slices = 8 / gcd(pairs, 8) unless forced (8: no locality, every XCD visits every pair)
    const dim3 grid(grid_blocks);            // one block (W cooperating waves) per query set, dealt as xcd_map_block says
    // region A (evaluations) is only read back by the profiler: per launch at level 1 (cleared here, copied out
    // after the launch), as a running total at level 2 (cleared when profiling starts, read once at the end)
    const bool per_launch = c->prof && !c->prof_totals;
    if (per_launch) MVR_HIP_TRY(c, hipMemsetAsync(c->evals, 0, kEvalRegion * sizeof(unsigned long long), c->stream));
#ifdef MVR_TRACE
    MVR_HIP_TRY(c, hipMemsetAsync(c->evals + 2 * kEvalRegion, 0, kTraceRec * kTraceBlocks * sizeof(unsigned long long), c->stream));
#endif
    ProfScope ps(c, MVR_K_NN, per_launch ? c->evals : nullptr, (int)(kEvalRegion * sizeof(unsigned long long)), 1.0,
                 per_launch ? upper : 0.0, per_launch ? kEvalShards : 0);
#define MVR_LAUNCH_CULL(F, QQ, WW) hipLaunchKernelGGL((nn_cull_kernel<F, QQ, WW>), grid, dim3(64 * WW), 0, c->stream, batch, map, c->evals)
#define MVR_LAUNCH_CULL_W(F, QQ)                                                                  \
  do { if (W == 1) MVR_LAUNCH_CULL(F, QQ, 1); else if (W == 2) MVR_LAUNCH_CULL(F, QQ, 2); else MVR_LAUNCH_CULL(F, QQ, 4); } while (0)
    if (fma) { if (Q == 2) MVR_LAUNCH_CULL_W(true, 2); else MVR_LAUNCH_CULL_W(true, 1); }
    else     { if (Q == 2) MVR_LAUNCH_CULL_W(false, 2); else MVR_LAUNCH_CULL_W(false, 1); }
#undef MVR_LAUNCH_CULL_W
#undef MVR_LAUNCH_CULL
    MVR_HIP_TRY(c, hipGetLastError());
  }
  return MVR_OK;
}

int launch_nn_cull(Ctx *c, const Cloud &q, size_t q_begin, size_t q_count, const uint8_t *qflags, const Cloud &t, float cap2,
                   bool fma, nnkey_t *keys)
{
  if (q_count == 0 || t.n == 0) return MVR_OK;
  if (q.n > 0xFFFFFFF0ull || t.n > 0xFFFFFFF0ull) return set_error(c, MVR_E_ARG, "cloud too large for 32-bit indices");
  const CullPair one = make_cull_pair(q, q_begin, q_count, qflags, t, keys);
  return launch_nn_cull_batch(c, &one, 1, cap2, fma);
}

namespace {
template <bool FMA>
__global__ void seed_to_bound_kernel(const float4 *__restrict__ qs, uint32_t nq, const float4 *__restrict__ ts, uint32_t nt,
                                     const uint32_t *__restrict__ seed, uint32_t *__restrict__ bound)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const uint32_t prev = seed[i];
  uint32_t v = 0xFFFFFFFFu;
  if (prev < nt) { const float4 q = qs[i]; v = __float_as_uint(dist2<FMA>(ts[prev], q.x, q.y, q.z)); }
  bound[i] = v;
}
__global__ void keys_to_seed_kernel(const float4 *__restrict__ qs, uint32_t nq, const nnkey_t *__restrict__ keys, const uint32_t *__restrict__ tinv,
                                    uint32_t *__restrict__ seed)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const uint32_t low = (uint32_t)keys[__float_as_uint(qs[i].w)];
  seed[i] = low != kNone ? tinv[low] : 0xFFFFFFFFu;
}
}  // namespace

int launch_seed_to_bound(Ctx *c, const float4 *qs, size_t nq, const float4 *ts, size_t nt, const uint32_t *seed, bool fma, uint32_t *bound)
{
  if (nq == 0) return MVR_OK;
  const unsigned blocks = (unsigned)((nq + 255) / 256);
  if (fma) hipLaunchKernelGGL(seed_to_bound_kernel<true>, dim3(blocks), dim3(256), 0, c->stream, qs, (uint32_t)nq, ts, (uint32_t)nt, seed, bound);
  else hipLaunchKernelGGL(seed_to_bound_kernel<false>, dim3(blocks), dim3(256), 0, c->stream, qs, (uint32_t)nq, ts, (uint32_t)nt, seed, bound);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_keys_to_seed(Ctx *c, const float4 *qs, size_t nq, const nnkey_t *keys, const uint32_t *tinv, uint32_t *seed)
{
  if (nq == 0) return MVR_OK;
  hipLaunchKernelGGL(keys_to_seed_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, c->stream, qs, (uint32_t)nq, keys, tinv, seed);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

CullPair make_cull_pair(const Cloud &q, size_t q_begin, size_t q_count, const uint8_t *qflags, const Cloud &t, nnkey_t *keys)
{
  CullPair p;
  p.qs = q.sorted; p.ts = t.sorted; p.tlo = t.tlo; p.thi = t.thi; p.cbox = t.cbox; p.sbox = t.sbox; p.qflags = qflags; p.keys = keys;
  p.q_begin = (uint32_t)q_begin; p.q_count = (uint32_t)q_count; p.nt = (uint32_t)t.n;
  p.n_tiles = (uint32_t)((t.n + kCullTile - 1) / kCullTile);
  return p;
}

}  // namespace mvr
