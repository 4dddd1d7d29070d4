// csrc/mvr_nn.hip -- K2/K3: exact brute-force 1-NN for gfx950 (MI355X).
//
// Replaces tree_->nearestKSearch(p, 1, ...) inside pcl::IterativeClosestPoint::
// align (mvr/src/registrator.cpp:569,920,1012,1024) and inside
// CorrespondenceEstimation::determineReciprocalCorrespondences (:502, :649).
//
// Bound: FP32 VALU issue -- 3 sub + 3 mul + 2 add per point pair in
// the spec's non-contracted form, nothing to contract onto MFMA.  Design:
//  * lane = query: every lane keeps Q query points and their running minima
//    in VGPRs; no cross-lane traffic in the inner loop.
//  * target points are staged HBM/L2 -> LDS in coalesced 16-byte-per-lane
//    tiles (double buffered, one barrier per tile) and read back as wave-wide
//    broadcasts (one LDS address per wave): one LDS read feeds 64*Q distance
//    evaluations.
//  * inside a sub-tile of SUB targets only min(d2) is tracked (v_min3_f32
//    covers two targets per instruction: 8.5 VALU per pair instead of 11 with
//    a compare + two selects); the winning sub-tile is re-scanned ONCE per
//    query at flush time to recover the index, with the spec's tie rule
//    (lowest index).
//  * the (query-block x target-tile) unit space is dealt evenly to a grid that
//    is sized to the chip's residency (persistent blocks, no tail wave);
//    partial results of different blocks meet in a packed 64-bit atomic min on
//    (d2 bits << 32 | index), which also implements the tie rule.
// Compiled with -ffp-contract=off: every float op below is rounded as written.
#include "mvr_internal.h"

namespace mvr {

namespace {

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

__device__ __forceinline__ float4 load_target(const float4 *__restrict__ t, uint32_t j, uint32_t nt)
{
  // tail padding: far away but finite, so d2 stays finite and never wins
  return (j < nt) ? t[j] : make_float4(1.0e18f, 1.0e18f, 1.0e18f, 1.0f);
}

template <bool FMA, int Q, int SUB>
__global__ void __launch_bounds__(kNNThreads)
nn_kernel(const float4 *__restrict__ qpts, uint32_t q_begin, uint32_t q_count,
          const uint32_t *__restrict__ qlist, const uint32_t *__restrict__ qcount,
          const float4 *__restrict__ tpts, uint32_t nt, nnkey_t *__restrict__ keys)
{
  constexpr int QB = kNNThreads * Q;               // queries per block
  constexpr int STG = kNNTile / kNNThreads;        // staged points per thread per tile
  __shared__ float4 tile[2][kNNTile];

  const uint32_t tid = threadIdx.x;
  const uint32_t nq = qlist ? min(*qcount, q_count) : q_count;
  const uint32_t n_qb = (nq + QB - 1) / QB;
  const uint32_t n_tiles = (nt + kNNTile - 1) / kNNTile;
  const uint64_t units = (uint64_t)n_qb * n_tiles;
  const uint64_t u_begin = units * blockIdx.x / gridDim.x;
  const uint64_t u_end = units * (blockIdx.x + 1) / gridDim.x;
  if (u_begin >= u_end) return;

  float qx[Q], qy[Q], qz[Q], best[Q];
  uint32_t bsub[Q];
  uint32_t cur_qb = kNone;

  auto stage_load = [&](uint64_t u, float4 (&r)[STG]) {
    const uint32_t tl = (uint32_t)(u % n_tiles);
#pragma unroll
    for (int k = 0; k < STG; ++k) r[k] = load_target(tpts, tl * kNNTile + k * kNNThreads + tid, nt);
  };
  auto stage_store = [&](int buf, const float4 (&r)[STG]) {
#pragma unroll
    for (int k = 0; k < STG; ++k) tile[buf][k * kNNThreads + tid] = r[k];
  };
  auto flush = [&]() {
    if (cur_qb == kNone) return;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const uint32_t ord = cur_qb * QB + q * kNNThreads + tid;
      if (ord >= nq) continue;
      // recover the index inside the winning sub-tile (ascending, strict <:
      // lowest index among equal distances)
      const uint32_t base = bsub[q] * SUB;
      float bd = __builtin_inff();
      uint32_t bi = kNone;
      for (int k = 0; k < SUB; ++k) {
        const uint32_t j = base + k;
        if (j < nt) {
          const float d = dist2<FMA>(tpts[j], qx[q], qy[q], qz[q]);
          if (d < bd) { bd = d; bi = j; }
        }
      }
      const nnkey_t key = ((nnkey_t)__float_as_uint(bd) << 32) | bi;
      atomicMin(&keys[qlist ? ord : (q_begin + ord)], key);
    }
  };

  float4 stg[STG];
  stage_load(u_begin, stg);
  stage_store(0, stg);
  __syncthreads();
  int buf = 0;

  for (uint64_t u = u_begin; u < u_end; ++u) {
    const uint32_t qb = (uint32_t)(u / n_tiles);
    const uint32_t tl = (uint32_t)(u - (uint64_t)qb * n_tiles);
    if (qb != cur_qb) {
      flush();
      cur_qb = qb;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const uint32_t ord = qb * QB + q * kNNThreads + tid;
        float4 p = make_float4(0.f, 0.f, 0.f, 1.f);
        if (ord < nq) p = qpts[qlist ? qlist[ord] : (q_begin + ord)];
        qx[q] = p.x; qy[q] = p.y; qz[q] = p.z;
        best[q] = __builtin_inff();
        bsub[q] = 0;
      }
    }
    const bool more = (u + 1 < u_end);
    if (more) stage_load(u + 1, stg);      // in flight during the compute below

    const float4 *__restrict__ T = tile[buf];
#pragma unroll 1
    for (int s = 0; s < kNNTile; s += SUB) {
      float m[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) m[q] = __builtin_inff();
#pragma unroll
      for (int k = 0; k < SUB; k += 2) {
        const float4 a = T[s + k], b = T[s + k + 1];   // wave-uniform address: LDS broadcast
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const float da = dist2<FMA>(a, qx[q], qy[q], qz[q]);
          const float db = dist2<FMA>(b, qx[q], qy[q], qz[q]);
          m[q] = __builtin_fminf(__builtin_fminf(m[q], da), db);
        }
      }
      const uint32_t sub = tl * (kNNTile / SUB) + (uint32_t)s / SUB;
#pragma unroll
      for (int q = 0; q < Q; ++q)
        if (m[q] < best[q]) { best[q] = m[q]; bsub[q] = sub; }   // strict: earliest sub-tile wins ties
    }

    if (more) stage_store(buf ^ 1, stg);
    __syncthreads();
    buf ^= 1;
  }
  flush();
}

__global__ void fill_u64_kernel(nnkey_t *p, size_t n, nnkey_t v)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

using nn_fn = void (*)(const float4 *, uint32_t, uint32_t, const uint32_t *, const uint32_t *, const float4 *,
                       uint32_t, nnkey_t *);

template <bool FMA>
nn_fn pick(int q, int sub)
{
  // (Q, SUB) variants kept for tuning (mvr_ctx_tune); default = (4, 32)
  if (q == 2 && sub == 32) return nn_kernel<FMA, 2, 32>;
  if (q == 4 && sub == 16) return nn_kernel<FMA, 4, 16>;
  if (q == 4 && sub == 64) return nn_kernel<FMA, 4, 64>;
  if (q == 6 && sub == 32) return nn_kernel<FMA, 6, 32>;
  if (q == 8 && sub == 32) return nn_kernel<FMA, 8, 32>;
  if (q == 8 && sub == 16) return nn_kernel<FMA, 8, 16>;
  return nn_kernel<FMA, 4, 32>;
}

}  // namespace

int launch_fill_u64(Ctx *c, nnkey_t *p, size_t n, nnkey_t v)
{
  if (n == 0) return MVR_OK;
  if (v == kKeyInit) {
    MVR_HIP_TRY(c, hipMemsetAsync(p, 0xFF, n * sizeof(nnkey_t), c->stream));
    return MVR_OK;
  }
  const int blocks = (int)std::min<size_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(fill_u64_kernel, dim3(blocks), dim3(256), 0, c->stream, p, n, v);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_nn(Ctx *c, const float4 *q, size_t q_begin, size_t q_count, const uint32_t *qlist,
              const uint32_t *qcount, const float4 *t, size_t nt, bool fma, nnkey_t *keys)
{
  if (q_count == 0 || nt == 0) return MVR_OK;   // keys stay at kKeyInit = "no neighbour"
  if (q_begin + q_count > 0xFFFFFFF0ull || nt > 0xFFFFFFF0ull)
    return set_error(c, MVR_E_ARG, "cloud too large for 32-bit indices");
  int Q = c->nn_q, SUB = c->nn_sub;
  const bool known = (Q == 2 && SUB == 32) || (Q == 4 && (SUB == 16 || SUB == 32 || SUB == 64)) ||
                     (Q == 6 && SUB == 32) || (Q == 8 && (SUB == 16 || SUB == 32));
  if (!known) { Q = 8; SUB = 32; }
  // persistent grid sized to residency (32 KB LDS per block => at most 5 per CU).
  // Measured on MI355X (tools/nn_sweep*.sh): 2 blocks/CU (2 waves/SIMD) with 8
  // queries per lane is fastest; more resident waves only add contention.
  const uint64_t resident = (uint64_t)c->n_cu * (uint64_t)std::max(1, std::min(c->nn_blocks_per_cu, 5));
  const uint64_t n_tiles = (nt + kNNTile - 1) / kNNTile;
  auto n_units = [&](int qq) { return ((q_count + (uint64_t)kNNThreads * qq - 1) / ((uint64_t)kNNThreads * qq)) * n_tiles; };
  // small problems: fewer queries per lane so that every resident block gets work
  if (SUB == 32) while (Q > 2 && n_units(Q) < 2 * resident) Q = (Q == 6) ? 4 : Q / 2;
  const uint64_t units = n_units(Q);
  const uint32_t blocks = (uint32_t)std::min<uint64_t>(units, resident);
  // evaluations: queries x targets; for the reverse pass the query count lives on the device
  ProfScope ps(c, MVR_K_NN, qlist ? qcount : nullptr, 4, (double)nt, (double)q_count * (double)nt);
  const nn_fn fn = fma ? pick<true>(Q, SUB) : pick<false>(Q, SUB);
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(kNNThreads), 0, c->stream, q, (uint32_t)q_begin, (uint32_t)q_count,
                     qlist, qcount, t, (uint32_t)nt, keys);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

}  // namespace mvr
