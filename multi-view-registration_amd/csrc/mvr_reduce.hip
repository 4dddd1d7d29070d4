// csrc/mvr_reduce.hip -- HBM-bound kernels of the ICP hot path for gfx950:
//   K1  transform_f32 / transform_f64  (ICP::transformCloud, getTransformedPoints)
//   K3  mark_kernel                    (dedupe of matched targets for the reciprocal pass)
//   K5  pass1_kernel                   (reciprocal filter + centroid sums)
//   K6  pass2_kernel                   (3x3 covariance about the centroids)
//   K7  fitness_kernel                 (getFitnessScore reduction)
//   K8  moments2_kernel                (raw second moments for LUM::computeEdge)
// All streaming/gather work: one 16-byte point per lane per access, f64
// accumulators per lane, wave reduction with __shfl_down (64 lanes), one LDS
// hop across the 4 waves, per-block partials to HBM and a fixed-order final
// sum -- bitwise reproducible run to run (no float atomics).
// Compiled with -ffp-contract=off.
#include "mvr_internal.h"

namespace mvr {

namespace {

struct Vec3d { double x, y, z; };

constexpr int kRT = 256;          // threads per reduction block
constexpr int kMaxBlocks = 1024;  // partial rows

// Sum of a double over the 64 lanes of a wave, the same value in every lane, without the LDS crossbar: an inclusive
// scan inside each row of 16 lanes by DPP row shifts (lanes shifted in from outside the row read as 0), then the four row
// totals -- lanes 15, 31, 47, 63 -- are read as scalars and added in order.  (`__shfl_down` of a double is two
// `ds_bpermute` per step: the 29 sums of the moments kernel cost 348 of them per wave and made that kernel LDS-bound,
// 59 us for the 12 pairs of a ring step.)  Fixed order, so the sums are reproducible run to run.
__device__ __forceinline__ double dpp_row_shr(double v, int ctrl)
{
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  int lo = (int)(unsigned)b, hi = (int)(unsigned)(b >> 32);
  switch (ctrl) {      // the control word is an immediate
    case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xF, 0xF, true); break;
    case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xF, 0xF, true); break;
    case 4: lo = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xF, 0xF, true); break;
    default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xF, 0xF, true); break;
  }
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
__device__ __forceinline__ double lane_value(double v, int lane)
{
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)lo));
}
__device__ __forceinline__ double wave_sum(double v)
{
  v += dpp_row_shr(v, 1);
  v += dpp_row_shr(v, 2);
  v += dpp_row_shr(v, 4);
  v += dpp_row_shr(v, 8);
  return ((lane_value(v, 15) + lane_value(v, 31)) + lane_value(v, 47)) + lane_value(v, 63);
}

// block-wide sums of K per-thread doubles -> row `blockIdx.x` of partials
template <int K>
__device__ __forceinline__ void block_partials_at(const double (&acc)[K], double *__restrict__ partials, unsigned row)
{
  __shared__ double lds[kRT / 64][K];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double s = wave_sum(acc[k]);
    if (lane == 0) lds[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double s = lds[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < kRT / 64; ++w) s += lds[w][threadIdx.x];
    partials[(size_t)row * K + threadIdx.x] = s;
  }
}

template <int K>
__device__ __forceinline__ void block_partials(const double (&acc)[K], double *__restrict__ partials)
{
  __shared__ double lds[kRT / 64][K];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double s = wave_sum(acc[k]);
    if (lane == 0) lds[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double s = lds[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < kRT / 64; ++w) s += lds[w][threadIdx.x];
    partials[(size_t)blockIdx.x * K + threadIdx.x] = s;
  }
}

// fixed-order sum of `rows` partial rows of K doubles (one block, K <= 32).
// Thread (g, k) = (tid / 32, tid % 32) adds rows g, g+8, ... of column k, so a
// row is read as one contiguous run of K doubles; the 8 row groups are then
// added in a fixed order -> bitwise reproducible.
template <int K>
__device__ __forceinline__ void sum_rows(const double *__restrict__ partials, int rows, double (&out)[K],
                                         double * /*unused*/)
{
  static_assert(K <= 32, "sum_rows: K <= 32");
  constexpr int KP = (K <= 8) ? 8 : ((K <= 16) ? 16 : 32);   // columns padded to a power of two
  constexpr int G = kRT / KP;                                 // row groups
  __shared__ double grp[G][KP];
  __shared__ double tot[32];
  const int g = threadIdx.x / KP, k = threadIdx.x % KP;
  double s = 0.0;
  if (k < K) {
#pragma unroll 8
    for (int r = g; r < rows; r += G) s += partials[(size_t)r * K + k];
  }
  grp[g][k] = s;
  __syncthreads();
  if (threadIdx.x < K) {
    double t = grp[0][threadIdx.x];
#pragma unroll
    for (int gg = 1; gg < G; ++gg) t += grp[gg][threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < K; ++kk) out[kk] = tot[kk];
}

// ---------------------------------------------------------------- K1 transforms

struct Mat34f { float m[12]; };   // rows of the 3x4, row-major

__global__ void transform_f32_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n, Mat34f T)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = in[i];
  float4 o;
  o.x = ((T.m[0] * p.x + T.m[1] * p.y) + T.m[2] * p.z) + T.m[3];
  o.y = ((T.m[4] * p.x + T.m[5] * p.y) + T.m[6] * p.z) + T.m[7];
  o.z = ((T.m[8] * p.x + T.m[9] * p.y) + T.m[10] * p.z) + T.m[11];
  o.w = 1.0f;
  out[i] = o;
}

__global__ void transform_f64_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n, Mat44d T)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = in[i];
  const double x = p.x, y = p.y, z = p.z;
  const double d = 1.0 / (((T.m[3] * x + T.m[7] * y) + T.m[11] * z) + T.m[15]);
  float4 o;
  o.x = (float)((((T.m[0] * x + T.m[4] * y) + T.m[8] * z) + T.m[12]) * d);
  o.y = (float)((((T.m[1] * x + T.m[5] * y) + T.m[9] * z) + T.m[13]) * d);
  o.z = (float)((((T.m[2] * x + T.m[6] * y) + T.m[10] * z) + T.m[14]) * d);
  o.w = 1.0f;
  out[i] = o;
}

// ---- target sharding over ranks: local <-> global target indices inside the u64 NN keys ----
// out = signed keys a MIN all-reduce can combine: (d2 bits << 32) | GLOBAL index, INT64_MAX for "none"
// (d2 >= 0, so the sign bit is clear and signed order = (d2, index) order)
__global__ void export_keys_kernel(const nnkey_t *__restrict__ keys, size_t n, SegTable st, long long *__restrict__ out)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const nnkey_t k = keys[i];
  uint32_t j = (uint32_t)k;
  if (j == kNone || (k >> 63)) { out[i] = 0x7FFFFFFFFFFFFFFFll; return; }
  for (uint32_t s = 0; s < st.n; ++s)
    if (j >= st.lb[s] && j - st.lb[s] < st.cnt[s]) { j = st.gb[s] + (j - st.lb[s]); break; }
  out[i] = (long long)((k & 0xFFFFFFFF00000000ull) | j);
}

// the reduced keys back into this shard's terms: a match this shard owns becomes a local index,
// everything else "none" -- so every correspondence is processed by exactly one rank
__global__ void import_keys_kernel(const long long *__restrict__ in, size_t n, SegTable st, nnkey_t *__restrict__ keys)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = (unsigned long long)in[i];
  const uint32_t j = (uint32_t)k;
  nnkey_t o = kKeyInit;
  if (k != 0x7FFFFFFFFFFFFFFFull && j != kNone) {
    if (st.n == 0) o = k;
    for (uint32_t s = 0; s < st.n; ++s)
      if (j >= st.gb[s] && j - st.gb[s] < st.cnt[s]) { o = (k & 0xFFFFFFFF00000000ull) | (st.lb[s] + (j - st.gb[s])); break; }
  }
  keys[i] = o;
}

// up to kBatchClouds clouds in one launch (blockIdx.y = cloud): the per-view kernels of a global iteration
// are a few microseconds each, so a dozen separate launches cost more than the work
__global__ void transform_f64_batch_kernel(XformBatch b)
{
  const int k = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.n[k]) return;
  const float4 o = pose_point_f64(b.Tp[k] ? *b.Tp[k] : b.T[k], b.src[k][i]);
  b.dst[k][i] = o;
}

__global__ void unpack_xyz_kernel(const float *__restrict__ packed, float4 *__restrict__ out, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = make_float4(packed[3 * i], packed[3 * i + 1], packed[3 * i + 2], 1.0f);
}

__global__ void pack_xyz_kernel(const float4 *__restrict__ in, float *__restrict__ packed, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = in[i];
  packed[3 * i] = p.x; packed[3 * i + 1] = p.y; packed[3 * i + 2] = p.z;
}

__global__ void decode_keys_kernel(const nnkey_t *__restrict__ keys, size_t n, uint32_t *__restrict__ idx,
                                   float *__restrict__ d2)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const nnkey_t k = keys[i];
  const uint32_t j = (uint32_t)k;
  if (idx) idx[i] = j;
  if (d2) d2[i] = (j == kNone) ? __builtin_inff() : __uint_as_float((uint32_t)(k >> 32));
}

// ------------------------------------------------------------------- K3 mark
// Every source point whose NN is within max_dist marks its target; the first
// marker (atomicCAS) appends the target to the list of distinct reverse
// queries.  O(Ns) however large the target has grown (registrator.cpp:576).
__global__ void mark_kernel(const nnkey_t *__restrict__ keys, size_t q_begin, size_t q_count, double max2,
                            uint32_t *__restrict__ slot, uint32_t *__restrict__ list, uint32_t *__restrict__ count)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t j = kNone;
  bool win = false;
  if (k < q_count) {
    const nnkey_t key = keys[q_begin + k];
    j = (uint32_t)key;
    const float d2 = __uint_as_float((uint32_t)(key >> 32));
    if (j != kNone && !((double)d2 > max2)) win = (atomicCAS(&slot[j], kNone, kMarked) == kNone);
  }
  // one counter add per wave: winners take consecutive list positions
  const unsigned long long mask = __ballot(win);
  if (mask == 0ull) return;
  const int lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(count, (uint32_t)__popcll(mask));
  base = __shfl(base, __ffsll((long long)mask) - 1, 64);
  if (win) {
    const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    list[pos] = j;
    slot[j] = pos;      // read only by later kernels on the same stream
  }
}

// -------------------------------------------------------------- K5 pass 1
// acc: 0 n, 1..3 sum p, 4..6 sum q, 7 sum d2
__global__ void __launch_bounds__(kRT)
pass1_kernel(const float4 *__restrict__ src, const float4 *__restrict__ tgt, const nnkey_t *__restrict__ keys,
             const nnkey_t *__restrict__ rkeys, const uint32_t *__restrict__ slot,
             const uint32_t *__restrict__ qperm, const uint32_t *__restrict__ tinv, size_t q_begin,
             size_t q_count, double max2, int reciprocal, int32_t *__restrict__ match,
             double *__restrict__ partials)
{
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < q_count; k += stride) {
    const size_t i = qperm ? (size_t)qperm[q_begin + k] : (q_begin + k);
    const nnkey_t key = keys[i];
    const uint32_t j = (uint32_t)key;
    const float d2 = __uint_as_float((uint32_t)(key >> 32));
    bool ok = (j != kNone) && !((double)d2 > max2);
    if (ok && reciprocal) {
      // App. A.2: NN of t_j in the source must be i itself, within max_dist
      const uint32_t tj = tinv ? tinv[j] : j;
      const nnkey_t rk = rkeys[slot ? slot[tj] : tj];        // compacted list (brute force) or in place (culled)
      const float dr = __uint_as_float((uint32_t)(rk >> 32));
      ok = ((uint32_t)rk == (uint32_t)i) && !((double)dr > max2);
    }
    match[i] = ok ? (int32_t)j : -1;
    if (ok) {
      const float4 p = src[i], q = tgt[j];
      acc[0] += 1.0;
      acc[1] += (double)p.x; acc[2] += (double)p.y; acc[3] += (double)p.z;
      acc[4] += (double)q.x; acc[5] += (double)q.y; acc[6] += (double)q.z;
      acc[7] += (double)d2;
    }
  }
  block_partials<8>(acc, partials);
}

// moments: [0] n, [1..3] mean p, [4..6] mean q, [7] mse
// (+ [17] = number of distinct reverse queries, for the evals accounting)
__global__ void __launch_bounds__(kRT) pass1_final_kernel(const double *__restrict__ partials, int rows,
                                                           const uint32_t *__restrict__ count,
                                                           double *__restrict__ moments)
{
  __shared__ double lds[8];
  double s[8];
  sum_rows<8>(partials, rows, s, lds);
  if (threadIdx.x == 0) {
    const double n = s[0];
    moments[0] = n;
    for (int k = 1; k < 8; ++k) moments[k] = (n > 0) ? s[k] / n : 0.0;
    moments[17] = count ? (double)*count : 0.0;
  }
}

// -------------------------------------------------------------- K6 pass 2
__global__ void __launch_bounds__(kRT)
pass2_kernel(const float4 *__restrict__ src, const float4 *__restrict__ tgt, const int32_t *__restrict__ match,
             const uint32_t *__restrict__ qperm, size_t q_begin, size_t q_count, const double *__restrict__ moments,
             double *__restrict__ partials)
{
  const double mpx = moments[1], mpy = moments[2], mpz = moments[3];
  const double mqx = moments[4], mqy = moments[5], mqz = moments[6];
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < q_count; k += stride) {
    const size_t i = qperm ? (size_t)qperm[q_begin + k] : (q_begin + k);
    const int32_t j = match[i];
    if (j < 0) continue;
    const float4 p = src[i], q = tgt[j];
    const double px = (double)p.x - mpx, py = (double)p.y - mpy, pz = (double)p.z - mpz;
    const double qx = (double)q.x - mqx, qy = (double)q.y - mqy, qz = (double)q.z - mqz;
    acc[0] += qx * px; acc[1] += qx * py; acc[2] += qx * pz;
    acc[3] += qy * px; acc[4] += qy * py; acc[5] += qy * pz;
    acc[6] += qz * px; acc[7] += qz * py; acc[8] += qz * pz;
  }
  block_partials<9>(acc, partials);
}

// (eval_totals, optional: the search kernels' running evaluation counters -- kEvalShards of them, kEvalStride apart; their sum
// goes to moments[18] so that the statistic travels with the moments' copy instead of one of its own)
__global__ void __launch_bounds__(kRT) pass2_final_kernel(const double *__restrict__ partials, int rows,
                                                           double *__restrict__ moments, const unsigned long long *__restrict__ eval_totals,
                                                           double *__restrict__ host_out, uint32_t host_seq)
{
  __shared__ double lds[9];
  double s[9];
  sum_rows<9>(partials, rows, s, lds);
  double ev = 0.0;
  if (eval_totals && threadIdx.x < 64) {
    unsigned long long v = threadIdx.x < kEvalShards ? eval_totals[(size_t)threadIdx.x * kEvalStride] : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    ev = (double)v;
    if (threadIdx.x == 0) moments[18] = ev;
  }
  if (threadIdx.x == 0) {
    const double n = moments[0];
    double sig[9];
    for (int k = 0; k < 9; ++k) { sig[k] = (n > 0) ? s[k] / n : 0.0; moments[8 + k] = sig[k]; }
    if (host_out) {
      // the iteration's whole row goes to the host from here: no copy packet behind this launch, no wake-up from a
      // synchronise -- the host spins on the word (one block, one fence)
      for (int k = 0; k < 8; ++k) host_out[k] = moments[k];
      for (int k = 0; k < 9; ++k) host_out[8 + k] = sig[k];
      host_out[17] = moments[17];
      host_out[18] = eval_totals ? ev : moments[18];
      __threadfence_system();
      __hip_atomic_store(reinterpret_cast<uint32_t *>(host_out + 64), host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// -------------------------------------------------------------- K8 moments2
// 28 sums about `origin`: n, p(3), q(3), pp(6), qq(6), pq(9)

__global__ void __launch_bounds__(kRT)
moments2_kernel(const float4 *__restrict__ src, const float4 *__restrict__ tgt, const int32_t *__restrict__ match,
                const nnkey_t *__restrict__ keys, const uint32_t *__restrict__ qperm, size_t q_begin, size_t q_count, Vec3d o,
                double *__restrict__ partials)
{
  double acc[29];       // [28] = sum of the correspondences' d2 (the f32 values the search found), when keys are given
#pragma unroll
  for (int k = 0; k < 29; ++k) acc[k] = 0.0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < q_count; k += stride) {
    const size_t i = qperm ? (size_t)qperm[q_begin + k] : (q_begin + k);
    const int32_t j = match[i];
    if (j < 0) continue;
    const float4 p4 = src[i], q4 = tgt[j];
    const double px = (double)p4.x - o.x, py = (double)p4.y - o.y, pz = (double)p4.z - o.z;
    const double qx = (double)q4.x - o.x, qy = (double)q4.y - o.y, qz = (double)q4.z - o.z;
    acc[0] += 1.0;
    acc[1] += px; acc[2] += py; acc[3] += pz;
    acc[4] += qx; acc[5] += qy; acc[6] += qz;
    acc[7] += px * px; acc[8] += px * py; acc[9] += px * pz; acc[10] += py * py; acc[11] += py * pz; acc[12] += pz * pz;
    acc[13] += qx * qx; acc[14] += qx * qy; acc[15] += qx * qz; acc[16] += qy * qy; acc[17] += qy * qz; acc[18] += qz * qz;
    acc[19] += px * qx; acc[20] += px * qy; acc[21] += px * qz;
    acc[22] += py * qx; acc[23] += py * qy; acc[24] += py * qz;
    acc[25] += pz * qx; acc[26] += pz * qy; acc[27] += pz * qz;
    if (keys) acc[28] += (double)__uint_as_float((uint32_t)(keys[i] >> 32));
  }
  block_partials<29>(acc, partials);
}

// K5 + K8 in one pass for the global mode: the reciprocal filter of pass1_kernel and the sums of
// moments2_kernel (same acceptance, same accumulation order, so the same bits), without the centroid
// reduction, its final kernel and the match[] round trip that only the two-pass covariance needs.
// BYPOS (the fused global pass): the forward keys are stored by SORTED position and the points are read from
// the Hilbert-ordered copies (src = sorted source, tgt = sorted target, w = original index); the key's low word is
// the match's sorted position (the search kernel reports it), so no index lookup is left: neighbouring queries read
// neighbouring matches.  Same points, same acceptance, same order of additions as the walk over the original arrays.
template <bool BYPOS>
__device__ __forceinline__ void accept_moments2_body(const float4 *__restrict__ src, const float4 *__restrict__ tgt,
                                                     const nnkey_t *__restrict__ keys, const nnkey_t *__restrict__ rkeys,
                                                     const uint32_t *__restrict__ slot, const uint32_t *__restrict__ qperm,
                                                     const uint32_t *__restrict__ tinv, size_t q_begin, size_t q_count, double max2,
                                                     int reciprocal, Vec3d o, unsigned block, unsigned n_blocks,
                                                     double *__restrict__ partials)
{
  double acc[29];
#pragma unroll
  for (int k = 0; k < 29; ++k) acc[k] = 0.0;
  const size_t stride = (size_t)n_blocks * blockDim.x;
  // Every accepted match sits at the end of a chain of dependent gathers (qperm -> key -> tinv -> reverse key ->
  // the two points), so a thread walks kUn of its queries at once: each stage issues its loads for all of them
  // before the next stage waits.  The sums are still added in the order k, k + stride, ... (same bits as a
  // one-at-a-time walk).
#ifndef MVR_ACCEPT_UN
#define MVR_ACCEPT_UN 3
#endif
  constexpr int kUn = MVR_ACCEPT_UN;        // measured: 2 and 3 equal (68 us for the 12 pairs of the ring step), 4: 78 us, 6: 93 us (registers cost residency)
  for (size_t k0 = (size_t)block * blockDim.x + threadIdx.x; k0 < q_count; k0 += kUn * stride) {
    size_t i[kUn];                 // key slot (BYPOS: sorted position), then the query's original index
    nnkey_t key[kUn];
    uint32_t j[kUn], tpos[kUn];    // match: original index, position in tgt[]
    float d2[kUn];
    bool ok[kUn];
    float4 p4[kUn], q4[kUn];
#pragma unroll
    for (int u = 0; u < kUn; ++u) {
      const size_t k = k0 + (size_t)u * stride;
      ok[u] = k < q_count;
      i[u] = 0;
      if (ok[u]) i[u] = (!BYPOS && qperm) ? (size_t)qperm[q_begin + k] : (q_begin + k);
    }
#pragma unroll
    for (int u = 0; u < kUn; ++u) {
      key[u] = ~(nnkey_t)0; p4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok[u]) { key[u] = keys[i[u]]; if (BYPOS) p4[u] = src[i[u]]; }
    }
#pragma unroll
    for (int u = 0; u < kUn; ++u) {
      j[u] = (uint32_t)key[u];
      d2[u] = __uint_as_float((uint32_t)(key[u] >> 32));
      ok[u] = ok[u] && (j[u] != kNone) && !((double)d2[u] > max2);
      if (BYPOS) i[u] = (size_t)__float_as_uint(p4[u].w);
    }
    if (BYPOS) {            // the fused pass's forward keys carry the match's sorted position itself
#pragma unroll
      for (int u = 0; u < kUn; ++u) tpos[u] = ok[u] ? j[u] : 0u;
    } else {
#pragma unroll
      for (int u = 0; u < kUn; ++u) tpos[u] = j[u];
    }
    if (reciprocal) {
      uint32_t tj[kUn];
      nnkey_t rk[kUn];
      if (BYPOS) {
#pragma unroll
        for (int u = 0; u < kUn; ++u) tj[u] = tpos[u];
      } else {
#pragma unroll
        for (int u = 0; u < kUn; ++u) { tj[u] = j[u]; if (ok[u] && tinv) tj[u] = tinv[j[u]]; }
      }
      if (slot) {                                          // reverse keys by list position (compacted queries); by sorted position otherwise
#pragma unroll
        for (int u = 0; u < kUn; ++u) if (ok[u]) tj[u] = slot[tj[u]];
      }
#pragma unroll
      for (int u = 0; u < kUn; ++u) { rk[u] = 0; if (ok[u]) rk[u] = rkeys[tj[u]]; }
#pragma unroll
      for (int u = 0; u < kUn; ++u) {
        const float dr = __uint_as_float((uint32_t)(rk[u] >> 32));
        ok[u] = ok[u] && ((uint32_t)rk[u] == (uint32_t)i[u]) && !((double)dr > max2);
      }
    }
#pragma unroll
    for (int u = 0; u < kUn; ++u) {
      q4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok[u]) { if (!BYPOS) p4[u] = src[i[u]]; q4[u] = tgt[tpos[u]]; }
    }
#pragma unroll
    for (int u = 0; u < kUn; ++u) {
      if (!ok[u]) continue;
      const double px = (double)p4[u].x - o.x, py = (double)p4[u].y - o.y, pz = (double)p4[u].z - o.z;
      const double qx = (double)q4[u].x - o.x, qy = (double)q4[u].y - o.y, qz = (double)q4[u].z - o.z;
      acc[0] += 1.0;
      acc[1] += px; acc[2] += py; acc[3] += pz;
      acc[4] += qx; acc[5] += qy; acc[6] += qz;
      acc[7] += px * px; acc[8] += px * py; acc[9] += px * pz; acc[10] += py * py; acc[11] += py * pz; acc[12] += pz * pz;
      acc[13] += qx * qx; acc[14] += qx * qy; acc[15] += qx * qz; acc[16] += qy * qy; acc[17] += qy * qz; acc[18] += qz * qz;
      acc[19] += px * qx; acc[20] += px * qy; acc[21] += px * qz;
      acc[22] += py * qx; acc[23] += py * qy; acc[24] += py * qz;
      acc[25] += pz * qx; acc[26] += pz * qy; acc[27] += pz * qz;
      acc[28] += (double)d2[u];
    }
  }
  block_partials_at<29>(acc, partials, block);
}

__global__ void __launch_bounds__(kRT)
accept_moments2_kernel(const float4 *__restrict__ src, const float4 *__restrict__ tgt, const nnkey_t *__restrict__ keys,
                       const nnkey_t *__restrict__ rkeys, const uint32_t *__restrict__ slot, const uint32_t *__restrict__ qperm,
                       const uint32_t *__restrict__ tinv, size_t q_begin, size_t q_count, double max2, int reciprocal, Vec3d o,
                       double *__restrict__ partials)
{
  accept_moments2_body<false>(src, tgt, keys, rkeys, slot, qperm, tinv, q_begin, q_count, max2, reciprocal, o, blockIdx.x, gridDim.x, partials);
}

// the same for all scan pairs of a global pass in one launch (blockIdx.y = pair); a pair uses its OWN number of
// blocks as the grid stride, so its sums are accumulated in exactly the order of the one-pair launch
__global__ void __launch_bounds__(kRT) accept_moments2_batch_kernel(GlueBatch b)
{
  const GluePair &a = b.p[blockIdx.y];
  if ((int)blockIdx.x >= a.blocks) return;
  // the start bounds the forward launch marked have been consumed by the reverse launch: back to "no target matched",
  // ready for the next pass's forward launch (nothing in this kernel reads them)
  if (blockIdx.x == 0 && threadIdx.x == 0) { if (a.zero_a) *a.zero_a = 0u; if (a.zero_b) *a.zero_b = 0u; if (a.zero_c) *a.zero_c = 0u; }
  if (a.bound && b.reciprocal)
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.nt; i += (unsigned long long)a.blocks * blockDim.x)
      a.bound[i] = 0xFFFFFFFFu;
  const Vec3d o{b.origin[0], b.origin[1], b.origin[2]};
  if (a.by_pos)
    accept_moments2_body<true>(a.qs, a.ts, a.keys, a.rkeys, a.slot, nullptr, a.tinv, (size_t)a.q_begin, (size_t)a.q_count, b.max2, b.reciprocal, o,
                               blockIdx.x, (unsigned)a.blocks, a.partials);
  else
    accept_moments2_body<false>(a.src, a.tgt, a.keys, a.rkeys, a.slot, a.qperm, a.tinv, (size_t)a.q_begin, (size_t)a.q_count, b.max2, b.reciprocal, o,
                                blockIdx.x, (unsigned)a.blocks, a.partials);
}

// out (32 doubles): [0] n, [1..3] origin, [4..6] sp, [7..9] sq, [10..15] spp, [16..21] sqq, [22..30] spq, [31] sum d2 (or 0)
__global__ void __launch_bounds__(kRT) moments2_final_kernel(const double *__restrict__ partials, int rows, Vec3d o,
                                                              double *__restrict__ out)
{
  __shared__ double lds[29];
  double s[29];
  sum_rows<29>(partials, rows, s, lds);
  if (threadIdx.x == 0) {
    out[0] = s[0]; out[1] = o.x; out[2] = o.y; out[3] = o.z;
    for (int k = 1; k < 28; ++k) out[3 + k] = s[k];
    out[31] = s[28];
  }
}

__global__ void __launch_bounds__(kRT) moments2_final_batch_kernel(GlueBatch b)
{
  const GluePair &a = b.p[blockIdx.x];
  __shared__ double lds[29];
  double s[29];
  sum_rows<29>(a.partials, a.blocks, s, lds);
  if (threadIdx.x == 0) {
    a.out[0] = s[0]; a.out[1] = b.origin[0]; a.out[2] = b.origin[1]; a.out[3] = b.origin[2];
    for (int k = 1; k < 28; ++k) a.out[3 + k] = s[k];
    a.out[31] = s[28];
    if (b.done_word) {
      // this block's row is on its way to the host; the last block to get here tells the host that all of them are
      // (a dozen blocks: a dozen fences, not the thousands that made a last-block reduction of the sums themselves a loss)
      __threadfence_system();
      const uint32_t before = __hip_atomic_fetch_add(b.done_counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (before + 1u == gridDim.x) {
        __hip_atomic_store(b.done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the next launch that counts here starts after this one has ended)
        __threadfence_system();
        __hip_atomic_store(b.done_word, b.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// -------------------------------------------------------------- K10 point-to-plane
// Extension (BASELINE config 2; no counterpart in the reference).
// PCL TransformationEstimationPointToPlaneLLS: row a = [p x n, n], d = n.(q - p),
// all in double.  acc: 0..20 upper triangle of a a^T (row-major), 21..26 a*d, 27 count, 28 d^2
__global__ void __launch_bounds__(kRT)
p2plane_kernel(const float4 *__restrict__ src, const float4 *__restrict__ tgt, const float4 *__restrict__ tnrm,
               const int32_t *__restrict__ match, const uint32_t *__restrict__ qperm, size_t q_begin, size_t q_count,
               double *__restrict__ partials)
{
  double acc[29];
#pragma unroll
  for (int k = 0; k < 29; ++k) acc[k] = 0.0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < q_count; k += stride) {
    const size_t i = qperm ? (size_t)qperm[q_begin + k] : (q_begin + k);
    const int32_t j = match[i];
    if (j < 0) continue;
    const float4 p = src[i], q = tgt[j], n4 = tnrm[j];
    const double sx = p.x, sy = p.y, sz = p.z, nx = n4.x, ny = n4.y, nz = n4.z;
    double a[6];
    a[0] = nz * sy - ny * sz; a[1] = nx * sz - nz * sx; a[2] = ny * sx - nx * sy;
    a[3] = nx; a[4] = ny; a[5] = nz;
    const double d = nx * (double)q.x + ny * (double)q.y + nz * (double)q.z - nx * sx - ny * sy - nz * sz;
    int t = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int cc = r; cc < 6; ++cc) acc[t++] += a[r] * a[cc];
#pragma unroll
    for (int r = 0; r < 6; ++r) acc[21 + r] += a[r] * d;
    acc[27] += 1.0;
    acc[28] += d * d;
  }
  block_partials<29>(acc, partials);
}

__global__ void __launch_bounds__(kRT) p2plane_final_kernel(const double *__restrict__ partials, int rows,
                                                             double *__restrict__ out)
{
  double s[29];
  sum_rows<29>(partials, rows, s, nullptr);
  if (threadIdx.x == 0)
    for (int k = 0; k < 29; ++k) out[k] = s[k];
}

struct Mat33f { float m[9]; };
struct Mat33d { double m[9]; };

// normals rotate with the cloud: n' = R n, same operation order as the points, no translation
__global__ void rotate_normals_f32_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n, Mat33f R)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = in[i];
  float4 o;
  o.x = (R.m[0] * v.x + R.m[1] * v.y) + R.m[2] * v.z;
  o.y = (R.m[3] * v.x + R.m[4] * v.y) + R.m[5] * v.z;
  o.z = (R.m[6] * v.x + R.m[7] * v.y) + R.m[8] * v.z;
  o.w = 0.0f;
  out[i] = o;
}

__global__ void rotate_normals_f64_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n, Mat33d R)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = in[i];
  const double x = v.x, y = v.y, z = v.z;
  float4 o;
  o.x = (float)((R.m[0] * x + R.m[1] * y) + R.m[2] * z);
  o.y = (float)((R.m[3] * x + R.m[4] * y) + R.m[5] * z);
  o.z = (float)((R.m[6] * x + R.m[7] * y) + R.m[8] * z);
  o.w = 0.0f;
  out[i] = o;
}

// -------------------------------------------------------------- K7 fitness
__global__ void __launch_bounds__(kRT)
fitness_kernel(const nnkey_t *__restrict__ keys, size_t n, double max_range, double *__restrict__ partials)
{
  double acc[2] = {0, 0};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const nnkey_t key = keys[i];
    if ((uint32_t)key == kNone) continue;
    const float d2 = __uint_as_float((uint32_t)(key >> 32));
    if ((double)d2 <= max_range) { acc[0] += (double)d2; acc[1] += 1.0; }
  }
  block_partials<2>(acc, partials);
}

__global__ void __launch_bounds__(kRT) fitness_final_kernel(const double *__restrict__ partials, int rows,
                                                             double *__restrict__ moments)
{
  __shared__ double lds[2];
  double s[2];
  sum_rows<2>(partials, rows, s, lds);
  if (threadIdx.x == 0) { moments[0] = s[0]; moments[1] = s[1]; }
}

inline int reduce_blocks(const Ctx *c, size_t n)
{
  const size_t want = (n + kRT - 1) / kRT;
  // few partial rows keep the single-block final sum short; 2 blocks per CU
  // already put > 8 MB of 16-byte loads in flight
  const size_t cap = std::min<size_t>(kMaxBlocks, (size_t)c->n_cu);
  return (int)std::max<size_t>(1, std::min(want, cap));
}

// the launches that carry 29 f64 sums per lane (moments2 / accept_moments2, one pair or a batch): a block's epilogue -- 29
// wave reductions, the hop through LDS, a partial row -- costs five times what a lane's three queries cost when a 200k-query
// pair gets a block per CU; a quarter of the rows (12 queries per lane, one round of waves for a 12-pair batch instead of
// three) measured 5-7 us less per ring pass.  One rule for the whole family: its one-pair and batched launches add up in
// the same order.
inline int moments_blocks(const Ctx *c, size_t n)
{
  const size_t want = (n + kRT - 1) / kRT;
  size_t cap = std::max<size_t>(32, (size_t)c->n_cu / 4);
  if (c->reduce_rows > 0) cap = (size_t)c->reduce_rows;      // (tuning: "reduce_rows")
  return (int)std::max<size_t>(1, std::min(want, std::min<size_t>(kMaxBlocks, cap)));
}

inline int ensure_partials(Ctx *c, size_t doubles) { return ensure(c, c->partials, c->partials_cap, doubles); }

}  // namespace

int launch_transform_f32(Ctx *c, const float4 *in, float4 *out, size_t n, const float T[16])
{
  if (n == 0) return MVR_OK;
  Mat34f M;
  for (int r = 0; r < 3; ++r) for (int k = 0; k < 4; ++k) M.m[4 * r + k] = T[r + 4 * k];
  ProfScope ps(c, MVR_K_XFORM, 32.0 * (double)n);
  hipLaunchKernelGGL(transform_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, in, out, n, M);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_transform_f64(Ctx *c, const float4 *in, float4 *out, size_t n, const double T[16])
{
  if (n == 0) return MVR_OK;
  Mat44d M;
  for (int k = 0; k < 16; ++k) M.m[k] = T[k];
  ProfScope ps(c, MVR_K_XFORM, 32.0 * (double)n);
  hipLaunchKernelGGL(transform_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, in, out, n, M);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_transform_f64_batch(Ctx *c, int count, const float4 *const *in, float4 *const *out, const size_t *n, const double *T,
                               const Mat44d *const *Tp)
{
  for (int base = 0; base < count; base += kBatchClouds) {
    XformBatch b;
    const int m = std::min(kBatchClouds, count - base);
    size_t nmax = 0; double work = 0.0;
    for (int k = 0; k < kBatchClouds; ++k) {
      const bool live = k < m;
      b.src[k] = live ? in[base + k] : nullptr; b.dst[k] = live ? out[base + k] : nullptr; b.n[k] = live ? n[base + k] : 0;
      for (int j = 0; j < 16; ++j) b.T[k].m[j] = live ? T[(size_t)(base + k) * 16 + j] : 0.0;
      b.Tp[k] = (live && Tp) ? Tp[base + k] : nullptr;
      nmax = std::max(nmax, (size_t)b.n[k]); work += 32.0 * (double)b.n[k];
    }
    if (nmax == 0) continue;
    ProfScope ps(c, MVR_K_XFORM, work);
    hipLaunchKernelGGL(transform_f64_batch_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)m), dim3(256), 0, c->stream, b);
    MVR_HIP_TRY(c, hipGetLastError());
  }
  return MVR_OK;
}

int launch_export_keys(Ctx *c, const nnkey_t *keys, size_t n, const SegTable &st, long long *out)
{
  if (n == 0) return MVR_OK;
  hipLaunchKernelGGL(export_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, keys, n, st, out);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_import_keys(Ctx *c, const long long *in, size_t n, const SegTable &st, nnkey_t *keys)
{
  if (n == 0) return MVR_OK;
  hipLaunchKernelGGL(import_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, in, n, st, keys);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_unpack_xyz(Ctx *c, const float *packed, float4 *out, size_t n)
{
  if (n == 0) return MVR_OK;
  hipLaunchKernelGGL(unpack_xyz_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, packed, out, n);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_pack_xyz(Ctx *c, const float4 *in, float *packed, size_t n)
{
  if (n == 0) return MVR_OK;
  hipLaunchKernelGGL(pack_xyz_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, in, packed, n);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_decode_keys(Ctx *c, const nnkey_t *keys, size_t n, uint32_t *idx, float *d2)
{
  if (n == 0) return MVR_OK;
  hipLaunchKernelGGL(decode_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, keys, n, idx, d2);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_mark(Ctx *c, const nnkey_t *keys, size_t q_begin, size_t q_count, double max2, uint32_t *slot,
                uint32_t *list, uint32_t *count)
{
  if (q_count == 0) return MVR_OK;
  ProfScope ps(c, MVR_K_GLUE, 16.0 * (double)q_count);
  hipLaunchKernelGGL(mark_kernel, dim3((unsigned)((q_count + 255) / 256)), dim3(256), 0, c->stream, keys, q_begin,
                     q_count, max2, slot, list, count);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_pass1(Ctx *c, const float4 *src, const float4 *tgt, const nnkey_t *keys, const nnkey_t *rkeys,
                 const uint32_t *slot, const uint32_t *count, const uint32_t *qperm, const uint32_t *tinv,
                 size_t q_begin, size_t q_count, double max2, bool reciprocal, int32_t *match, double *moments)
{
  const int blocks = reduce_blocks(c, q_count);
  if (int rc = ensure_partials(c, (size_t)kMaxBlocks * 32)) return rc;
  // K5 touches Ns*(12+4+4) + 12*M algorithmic bytes (M <= Ns)
  ProfScope ps(c, MVR_K_REDUCE, 20.0 * (double)q_count);
  hipLaunchKernelGGL(pass1_kernel, dim3(blocks), dim3(kRT), 0, c->stream, src, tgt, keys, rkeys, slot, qperm,
                     tinv, q_begin, q_count, max2, reciprocal ? 1 : 0, match, c->partials);
  hipLaunchKernelGGL(pass1_final_kernel, dim3(1), dim3(kRT), 0, c->stream, c->partials, blocks,
                     reciprocal ? count : nullptr, moments);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_pass2(Ctx *c, const float4 *src, const float4 *tgt, const int32_t *match, const uint32_t *qperm,
                 size_t q_begin, size_t q_count, double *moments, const unsigned long long *eval_totals, double *host_out, uint32_t host_seq)
{
  const int blocks = reduce_blocks(c, q_count);
  if (int rc = ensure_partials(c, (size_t)kMaxBlocks * 32)) return rc;
  ProfScope ps(c, MVR_K_REDUCE, 20.0 * (double)q_count);
  hipLaunchKernelGGL(pass2_kernel, dim3(blocks), dim3(kRT), 0, c->stream, src, tgt, match, qperm, q_begin,
                     q_count, moments, c->partials);
  hipLaunchKernelGGL(pass2_final_kernel, dim3(1), dim3(kRT), 0, c->stream, c->partials, blocks, moments, eval_totals, host_out, host_seq);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_moments2(Ctx *c, const float4 *src, const float4 *tgt, const int32_t *match, const nnkey_t *keys, const uint32_t *qperm,
                    size_t q_begin, size_t q_count, const double origin[3], double *out)
{
  const int blocks = moments_blocks(c, q_count);
  if (int rc = ensure_partials(c, (size_t)kMaxBlocks * 32)) return rc;
  Vec3d o{origin[0], origin[1], origin[2]};
  ProfScope ps(c, MVR_K_REDUCE, 20.0 * (double)q_count);
  hipLaunchKernelGGL(moments2_kernel, dim3(blocks), dim3(kRT), 0, c->stream, src, tgt, match, keys, qperm, q_begin, q_count,
                     o, c->partials);
  hipLaunchKernelGGL(moments2_final_kernel, dim3(1), dim3(kRT), 0, c->stream, c->partials, blocks, o, out);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_accept_moments2(Ctx *c, const float4 *src, const float4 *tgt, const nnkey_t *keys, const nnkey_t *rkeys,
                           const uint32_t *slot, const uint32_t *qperm, const uint32_t *tinv, size_t q_begin, size_t q_count,
                           double max2, bool reciprocal, const double origin[3], double *out)
{
  const int blocks = moments_blocks(c, q_count);
  if (int rc = ensure_partials(c, (size_t)kMaxBlocks * 32)) return rc;
  Vec3d o{origin[0], origin[1], origin[2]};
  ProfScope ps(c, MVR_K_REDUCE, 40.0 * (double)q_count);
  hipLaunchKernelGGL(accept_moments2_kernel, dim3(blocks), dim3(kRT), 0, c->stream, src, tgt, keys, rkeys, slot, qperm, tinv,
                     q_begin, q_count, max2, reciprocal ? 1 : 0, o, c->partials);
  hipLaunchKernelGGL(moments2_final_kernel, dim3(1), dim3(kRT), 0, c->stream, c->partials, blocks, o, out);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int reduce_blocks_for(const Ctx *c, size_t n) { return moments_blocks(c, n); }

int launch_accept_moments2_batch(Ctx *c, const GlueBatch &b, int n_pairs)
{
  int bmax = 0; double work = 0.0;
  for (int k = 0; k < n_pairs; ++k) { bmax = std::max(bmax, b.p[k].blocks); work += 40.0 * (double)b.p[k].q_count; }
  if (bmax == 0) return MVR_OK;
  ProfScope ps(c, MVR_K_REDUCE, work);
  hipLaunchKernelGGL(accept_moments2_batch_kernel, dim3((unsigned)bmax, (unsigned)n_pairs), dim3(kRT), 0, c->stream, b);
  hipLaunchKernelGGL(moments2_final_batch_kernel, dim3((unsigned)n_pairs), dim3(kRT), 0, c->stream, b);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_p2plane(Ctx *c, const float4 *src, const float4 *tgt, const float4 *tnrm, const int32_t *match,
                   const uint32_t *qperm, size_t q_begin, size_t q_count, double *out)
{
  const int blocks = reduce_blocks(c, q_count);
  if (int rc = ensure_partials(c, (size_t)kMaxBlocks * 32)) return rc;
  ProfScope ps(c, MVR_K_REDUCE, 32.0 * (double)q_count);
  hipLaunchKernelGGL(p2plane_kernel, dim3(blocks), dim3(kRT), 0, c->stream, src, tgt, tnrm, match, qperm, q_begin,
                     q_count, c->partials);
  hipLaunchKernelGGL(p2plane_final_kernel, dim3(1), dim3(kRT), 0, c->stream, c->partials, blocks, out);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_rotate_normals_f32(Ctx *c, const float4 *in, float4 *out, size_t n, const float T[16])
{
  if (n == 0) return MVR_OK;
  Mat33f R;
  for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) R.m[3 * r + k] = T[r + 4 * k];
  hipLaunchKernelGGL(rotate_normals_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, in, out, n, R);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_rotate_normals_f64(Ctx *c, const float4 *in, float4 *out, size_t n, const double T[16])
{
  if (n == 0) return MVR_OK;
  Mat33d R;
  for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) R.m[3 * r + k] = T[r + 4 * k];
  hipLaunchKernelGGL(rotate_normals_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, in, out, n, R);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

int launch_fitness(Ctx *c, const nnkey_t *keys, size_t n, double max_range, double *moments)
{
  const int blocks = reduce_blocks(c, n);
  if (int rc = ensure_partials(c, (size_t)kMaxBlocks * 32)) return rc;
  ProfScope ps(c, MVR_K_REDUCE, 8.0 * (double)n);
  hipLaunchKernelGGL(fitness_kernel, dim3(blocks), dim3(kRT), 0, c->stream, keys, n, max_range, c->partials);
  hipLaunchKernelGGL(fitness_final_kernel, dim3(1), dim3(kRT), 0, c->stream, c->partials, blocks, moments);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

namespace {
__global__ void add_f64_kernel(double *__restrict__ dst, const double *__restrict__ src, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}
}  // namespace
int launch_add_f64(Ctx *c, double *dst, const double *src, size_t n)
{
  if (n == 0) return MVR_OK;
  hipLaunchKernelGGL(add_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, dst, src, n);
  MVR_HIP_TRY(c, hipGetLastError());
  return MVR_OK;
}

}  // namespace mvr
