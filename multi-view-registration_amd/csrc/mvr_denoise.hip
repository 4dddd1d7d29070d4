// csrc/mvr_denoise.hip -- PointCloud::denoise on the GPU (mvr/src/point_cloud.cpp:423-465,
// graph construction :467-500).
//
// The reference links the points by the Delaunay edges (CGAL) no longer than `triangle_length`, takes the
// connected components (boost) and drops the components with fewer than `segment_threshold` points.  No
// triangulation is needed for that: the Euclidean minimum spanning tree is a subgraph of the Delaunay
// triangulation, so two points are joined by Delaunay edges <= r exactly when they are joined in the graph of
// ALL point pairs <= r.  The components -- and therefore the output -- are identical (oracle/mvr_oracle.c states
// the same and is checked against scipy's Delaunay and radius graphs; exact duplicate points, which CGAL
// collapses into one vertex, are ordinary members of their component here).
//
//   1. uniform grid with cells >= r (10 bits per axis), points radix-sorted by cell key z|y|x: the 27
//      neighbouring cells of a point are 9 contiguous key ranges of the sorted array;
//   2. lock-free union-find (hook the larger root under the smaller with atomicCAS, path halving): one thread
//      per point walks its 9 ranges and unites with every earlier point within r (distance in double, the
//      reference's `sqrt(squared_distance) > threshold -> skip`);
//   3. labels = roots = smallest index of each component (boost::connected_components numbers components by
//      their first vertex), component sizes by atomicAdd, kept points ordered by (label, index) -- the order in
//      which the reference pushes them into the denoised cloud -- with one 64-bit radix sort.
// HBM/latency-bound integer work; nothing here is reshaped into a GEMM.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "mvr_internal.h"

namespace mvr {
namespace {

__global__ void bbox_kernel(const float4 *__restrict__ p, size_t n, float *__restrict__ out /* 6, pre-set to +-max */)
{
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = p[i];
    lo[0] = fminf(lo[0], v.x); lo[1] = fminf(lo[1], v.y); lo[2] = fminf(lo[2], v.z);
    hi[0] = fmaxf(hi[0], v.x); hi[1] = fmaxf(hi[1], v.y); hi[2] = fmaxf(hi[2], v.z);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    for (int k = 0; k < 3; ++k) {
      lo[k] = fminf(lo[k], __shfl_xor(lo[k], o, 64));
      hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o, 64));
    }
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 3; ++k) {      // float min/max through the ordered-int trick is not needed: few waves, CAS loop
      unsigned *alo = reinterpret_cast<unsigned *>(out + k), *ahi = reinterpret_cast<unsigned *>(out + 3 + k);
      unsigned old = *alo;
      while (__uint_as_float(old) > lo[k]) { const unsigned prev = atomicCAS(alo, old, __float_as_uint(lo[k])); if (prev == old) break; old = prev; }
      old = *ahi;
      while (__uint_as_float(old) < hi[k]) { const unsigned prev = atomicCAS(ahi, old, __float_as_uint(hi[k])); if (prev == old) break; old = prev; }
    }
}

struct Grid { double lo[3]; double inv_h; };

__device__ __forceinline__ uint32_t cell_key(const float4 p, const Grid g, uint32_t c[3])
{
  const double v[3] = {p.x, p.y, p.z};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double t = (v[k] - g.lo[k]) * g.inv_h;
    c[k] = t > 0.0 ? (t < 1023.0 ? (uint32_t)t : 1023u) : 0u;
  }
  return (c[2] << 20) | (c[1] << 10) | c[0];
}

__global__ void key_kernel(const float4 *__restrict__ p, size_t n, Grid g, uint32_t *__restrict__ key, uint32_t *__restrict__ idx)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t c[3];
  key[i] = cell_key(p[i], g, c);
  idx[i] = (uint32_t)i;
}

__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t x)
{
  for (;;) {
    const uint32_t p = __atomic_load_n(&parent[x], __ATOMIC_RELAXED);
    if (p == x) return x;
    const uint32_t gp = __atomic_load_n(&parent[p], __ATOMIC_RELAXED);
    if (gp != p) __atomic_store_n(&parent[x], gp, __ATOMIC_RELAXED);       // path halving: any ancestor is a valid parent
    x = p;
  }
}

__device__ __forceinline__ void uf_unite(uint32_t *parent, uint32_t a, uint32_t b)
{
  for (;;) {
    a = uf_find(parent, a); b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) { const uint32_t t = a; a = b; b = t; }
    if (atomicCAS(&parent[a], a, b) == a) return;             // the larger root goes under the smaller one
  }
}

__global__ void unite_kernel(const float4 *__restrict__ p, const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sidx,
                             size_t n, double r, uint32_t *__restrict__ parent)
{
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  const uint32_t ka = skey[a], i = sidx[a];
  const float4 pi = p[i];
  const int cx = (int)(ka & 1023u), cy = (int)((ka >> 10) & 1023u), cz = (int)(ka >> 20);
  for (int dz = -1; dz <= 1; ++dz)
    for (int dy = -1; dy <= 1; ++dy) {
      const int z = cz + dz, y = cy + dy;
      if (z < 0 || z > 1023 || y < 0 || y > 1023) continue;
      const uint32_t k0 = ((uint32_t)z << 20) | ((uint32_t)y << 10) | (uint32_t)(cx > 0 ? cx - 1 : 0);
      const uint32_t k1 = ((uint32_t)z << 20) | ((uint32_t)y << 10) | (uint32_t)(cx < 1023 ? cx + 1 : 1023);
      size_t lo = 0, hi = n;
      while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (skey[mid] < k0) lo = mid + 1; else hi = mid; }
      for (size_t b = lo; b < n && skey[b] <= k1; ++b) {
        const uint32_t j = sidx[b];
        if (j >= i) continue;                                  // every pair once
        const float4 pj = p[j];
        const double dx = (double)pi.x - (double)pj.x, dyy = (double)pi.y - (double)pj.y, dzz = (double)pi.z - (double)pj.z;
        if (sqrt(dx * dx + dyy * dyy + dzz * dzz) > r) continue;      // point_cloud.cpp:490-491
        uf_unite(parent, i, j);
      }
    }
}

__global__ void iota_kernel(uint32_t *__restrict__ v, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (uint32_t)i;
}

// One round of pointer jumping: every node adopts its grandparent.  All depths halve at once, with independent
// gathers instead of the dependent walk of a find -- the object is one giant component whose tree comes out of the
// union pass thousands of links deep, and 200k concurrent finds through it took 2.3 ms.
__global__ void jump_kernel(uint32_t *__restrict__ parent, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t p = __atomic_load_n(&parent[i], __ATOMIC_RELAXED);
  const uint32_t gp = __atomic_load_n(&parent[p], __ATOMIC_RELAXED);
  if (gp != p) __atomic_store_n(&parent[i], gp, __ATOMIC_RELAXED);       // any ancestor is a valid parent
}

__global__ void label_kernel(uint32_t *__restrict__ parent, size_t n, uint32_t *__restrict__ label, uint32_t *__restrict__ size,
                             uint32_t *__restrict__ counters)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  uint32_t r = 0;
  if (live) {
    r = uf_find(parent, (uint32_t)i);      // (whatever depth the jumping rounds left: correctness does not depend on their number)
    // The label goes to its OWN array: other threads' finds are still halving paths through node i, and one of them
    // could store a non-root ancestor into parent[i] after a root written there (then read back as a label of size 0).
    // Roots are final here (no unions run concurrently): label = smallest index of the component.
    label[i] = r;
    if (r == (uint32_t)i) atomicAdd(&counters[0], 1u);          // number of components
  }
  // component sizes: the lanes of a wave that share a root add once (most of a wave belongs to the one big component)
  unsigned long long todo = __ballot(live);
  const int lane = threadIdx.x & 63;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t rl = (uint32_t)__builtin_amdgcn_readlane((int)r, leader);
    const unsigned long long same = __ballot(live && r == rl) & todo;
    if (lane == leader) atomicAdd(&size[rl], (uint32_t)__popcll(same));
    todo &= ~same;
  }
}

__global__ void order_kernel(const uint32_t *__restrict__ label, const uint32_t *__restrict__ size, size_t n, uint32_t thr,
                             unsigned long long *__restrict__ okey, uint32_t *__restrict__ counters)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t l = label[i];
  const bool keep = size[l] >= thr;
  okey[i] = keep ? (((unsigned long long)l << 32) | (unsigned long long)i) : ~0ull;      // (component, index); dropped points last
  if (keep) atomicAdd(&counters[1], 1u);
}

__global__ void gather_kernel(const float4 *__restrict__ in, const unsigned long long *__restrict__ okey, size_t kept,
                              float4 *__restrict__ out, uint32_t *__restrict__ index_out)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= kept) return;
  const uint32_t i = (uint32_t)okey[k];
  out[k] = in[i];
  if (index_out) index_out[k] = i;
}

// All scratch arrays are carved out of ONE grow-only buffer kept by the context (twelve hipMalloc + hipFree per call
// cost more than the kernels: 5.3 ms per 204k-point scan, 1.5 ms of it on the GPU).
struct Carver {
  char *base = nullptr;
  size_t off = 0;
  template <class T> void plan(size_t count) { off = ((off + 255) & ~(size_t)255) + std::max<size_t>(count, 1) * sizeof(T); }
  template <class T> T *take(size_t count)
  {
    off = (off + 255) & ~(size_t)255;
    T *p = reinterpret_cast<T *>(base + off);
    off += std::max<size_t>(count, 1) * sizeof(T);
    return p;
  }
};

}  // namespace

int denoise_cloud(Ctx *c, Cloud &cl, int segment_threshold, double triangle_length, size_t *n_kept, size_t *n_components,
                  uint32_t *host_index)
{
  const size_t n = cl.n;
  if (n_kept) *n_kept = 0;
  if (n_components) *n_components = 0;
  if (n == 0) return MVR_OK;
  if (n > 0xFFFFFFF0ull) return set_error(c, MVR_E_ARG, "cloud too large for 32-bit indices");
  if (!(triangle_length >= 0.0)) return set_error(c, MVR_E_ARG, "triangle_length must be >= 0");
  size_t bytes = 0, bytes2 = 0;            // hipCUB temporaries: sizes only depend on n
  MVR_HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                                    (int)n, 0, 30, c->stream));
  MVR_HIP_TRY(c, hipcub::DeviceRadixSort::SortKeys(nullptr, bytes2, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (int)n, 0, 64, c->stream));
  Carver plan;
  plan.plan<float>(8); for (int k = 0; k < 6; ++k) plan.plan<uint32_t>(n); plan.plan<uint32_t>(4);
  plan.plan<unsigned long long>(n); plan.plan<unsigned long long>(n); plan.plan<float4>(n); plan.plan<uint32_t>(n); plan.plan<char>(std::max(bytes, bytes2));
  if (int rc = ensure(c, c->dn_arena, c->dn_arena_cap, plan.off + 256)) return rc;
  Carver s; s.base = c->dn_arena;
  float *bbox = s.take<float>(8);
  uint32_t *key_a = s.take<uint32_t>(n), *key_b = s.take<uint32_t>(n), *idx_a = s.take<uint32_t>(n), *idx_b = s.take<uint32_t>(n),
           *parent = s.take<uint32_t>(n), *size = s.take<uint32_t>(n), *counters = s.take<uint32_t>(4);
  unsigned long long *okey_a = s.take<unsigned long long>(n), *okey_b = s.take<unsigned long long>(n);
  float4 *out = s.take<float4>(n);
  uint32_t *d_index = s.take<uint32_t>(n);
  char *tmp = s.take<char>(std::max(bytes, bytes2));
  const unsigned nb = (unsigned)((n + 255) / 256);
  ProfScope ps(c, MVR_K_GLUE, 200.0 * (double)n);
  // 1. grid
  const float init[6] = {3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
  float hb[6];
  MVR_HIP_TRY(c, hipMemcpyAsync(bbox, init, sizeof init, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(bbox_kernel, dim3(std::min(nb, 64u)), dim3(256), 0, c->stream, cl.pts, n, bbox);      // few waves: they all CAS the same six words
  MVR_HIP_TRY(c, hipMemcpyAsync(hb, bbox, sizeof hb, hipMemcpyDeviceToHost, c->stream));
  MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
  Grid g;
  double ext = 0.0;
  for (int k = 0; k < 3; ++k) { g.lo[k] = hb[k]; ext = std::max(ext, (double)hb[3 + k] - (double)hb[k]); }
  const double h = std::max(std::max(triangle_length, ext / 1023.0) * 1.000001, 1e-30);     // cells >= r: neighbours within r are <= 1 cell away
  g.inv_h = 1.0 / h;
  hipLaunchKernelGGL(key_kernel, dim3(nb), dim3(256), 0, c->stream, cl.pts, n, g, key_a, idx_a);
  MVR_HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(tmp, bytes, key_a, key_b, idx_a, idx_b, (int)n, 0, 30, c->stream));
  // 2. components
  hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(256), 0, c->stream, parent, n);
  hipLaunchKernelGGL(unite_kernel, dim3(nb), dim3(256), 0, c->stream, cl.pts, key_b, idx_b, n, triangle_length, parent);
  MVR_HIP_TRY(c, hipMemsetAsync(size, 0, n * sizeof(uint32_t), c->stream));
  MVR_HIP_TRY(c, hipMemsetAsync(counters, 0, 4 * sizeof(uint32_t), c->stream));
  for (int round = 0; round < 16; ++round) hipLaunchKernelGGL(jump_kernel, dim3(nb), dim3(256), 0, c->stream, parent, n);
  uint32_t *label = key_a;          // the unsorted keys are dead since the sort
  hipLaunchKernelGGL(label_kernel, dim3(nb), dim3(256), 0, c->stream, parent, n, label, size, counters);
  // 3. keep and order
  const uint32_t thr = segment_threshold > 0 ? (uint32_t)segment_threshold : 0u;
  hipLaunchKernelGGL(order_kernel, dim3(nb), dim3(256), 0, c->stream, label, size, n, thr, okey_a, counters);
  MVR_HIP_TRY(c, hipcub::DeviceRadixSort::SortKeys(tmp, bytes2, okey_a, okey_b, (int)n, 0, 64, c->stream));
  uint32_t hc[4];
  MVR_HIP_TRY(c, hipMemcpyAsync(hc, counters, sizeof hc, hipMemcpyDeviceToHost, c->stream));
  MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
  const size_t kept = hc[1];
  if (kept) {
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((kept + 255) / 256)), dim3(256), 0, c->stream, cl.pts, okey_b, kept, out, d_index);
    MVR_HIP_TRY(c, hipMemcpyAsync(cl.pts, out, kept * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    if (cl.has_normals) {
      hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((kept + 255) / 256)), dim3(256), 0, c->stream, cl.nrm, okey_b, kept, out, (uint32_t *)nullptr);
      MVR_HIP_TRY(c, hipMemcpyAsync(cl.nrm, out, kept * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    }
    if (host_index) MVR_HIP_TRY(c, hipMemcpyAsync(host_index, d_index, kept * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  }
  MVR_HIP_TRY(c, hipGetLastError());
  MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));          // host_index is complete on return
  cl.n = kept;
  cl.segs.clear();
  new_point_set(c, cl);
  if (n_kept) *n_kept = kept;
  if (n_components) *n_components = hc[0];
  return MVR_OK;
}

}  // namespace mvr
