// csrc/mvr_ctx.hip -- C-ABI entry points of libmvr_hip.so (include/mvr_hip.h):
// context, device clouds, and the host side of the ICP iteration
// (pcl::IterativeClosestPoint::align) driving the HIP
// kernels of mvr_nn.hip / mvr_index.hip / mvr_reduce.hip.  One host<->device
// round trip per ICP iteration: the 18 f64 moments come back, the 3x3 SVD runs
// on the host, the new 4x4 goes out as a kernel argument.
#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "mvr_internal.h"

namespace mvr {

int set_error(Ctx *c, int status, const char *what, hipError_t e)
{
  if (c) {
    c->last_error = what ? what : "";
    if (e != hipSuccess) { c->last_error += ": "; c->last_error += hipGetErrorString(e); }
  }
  return status;
}

void host_mark(const char *what)
{
  static const bool on = std::getenv("MVR_TRACE_HOST") != nullptr;
  if (!on) return;
  using namespace std::chrono;
  static double last = 0.0;
  const double t = duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
  std::fprintf(stderr, "[mvr host] %8.3f ms  %s\n", last == 0.0 ? 0.0 : t - last, what);
  last = t;
}

static hipEvent_t take_event(Ctx *c)
{
  if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}

ProfScope::ProfScope(Ctx *ctx, int family, double w) : c(ctx), fam(family), work(w)
{
  if (!c->prof || !((c->prof_mask >> family) & 1u)) return;
  a = take_event(c); b = take_event(c);
  if (!a || !b) { a = b = nullptr; return; }
  (void)hipEventRecord(a, c->stream);
}
ProfScope::ProfScope(Ctx *ctx, int family, const void *dev_count, int bytes, double per_cnt, double upper_bound, int shards)
    : ProfScope(ctx, family, upper_bound)
{
  d_count = dev_count; d_bytes = bytes; per_count = per_cnt; n_shards = shards;
}
ProfScope::~ProfScope()
{
  if (!c->prof || !a || !b) return;
  (void)hipEventRecord(b, c->stream);
  const uint64_t *hc = nullptr;
  const size_t need = n_shards > 0 ? kEvalRegion : 1;
  if (d_count && c->h_counts && c->h_counts_used + need <= kProfCounts) {
    uint64_t *slot = c->h_counts + c->h_counts_used;
    *slot = 0;
    if (hipMemcpyAsync(slot, d_count, (size_t)d_bytes, hipMemcpyDeviceToHost, c->stream) == hipSuccess) {
      hc = slot; c->h_counts_used += need;
    }
  }
  c->recs.push_back(ProfRec{fam, a, b, work, hc, n_shards, per_count});
}

namespace {

int prof_drain(Ctx *c)
{
  for (Ctx *w : c->workers) {          // worker launches are accounted in the parent
    if (int rc = prof_drain(w)) return rc;
    for (int f = 0; f < MVR_K_COUNT; ++f) {
      c->prof_launches[f] += w->prof_launches[f]; c->prof_ms[f] += w->prof_ms[f]; c->prof_work[f] += w->prof_work[f];
      w->prof_launches[f] = 0; w->prof_ms[f] = 0.0; w->prof_work[f] = 0.0;
    }
  }
  if (c->prof_totals && c->evals) {      // level 2: the culled kernel's evaluations of this context since the last drain
    MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
    uint64_t h[kEvalRegion];
    MVR_HIP_TRY(c, hipMemcpy(h, c->evals, sizeof h, hipMemcpyDeviceToHost));
    double s = 0.0;
    for (int k = 0; k < kEvalShards; ++k) s += (double)h[(size_t)k * kEvalStride];
    c->prof_work[MVR_K_NN] += s;
    MVR_HIP_TRY(c, hipMemset(c->evals, 0, sizeof h));
  }
  if (c->recs.empty()) return MVR_OK;
  MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (auto &r : c->recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      c->prof_launches[r.family] += 1; c->prof_ms[r.family] += ms;
      if (r.h_count && r.n_shards > 0) {
        double s = 0.0;
        for (int k = 0; k < r.n_shards; ++k) s += (double)r.h_count[(size_t)k * kEvalStride];
        c->prof_work[r.family] += s * r.per_count;
      } else {
        c->prof_work[r.family] += r.h_count ? (double)*r.h_count * r.per_count : r.work;
      }
    }
    c->event_pool.push_back(r.a); c->event_pool.push_back(r.b);
  }
  c->recs.clear();
  c->h_counts_used = 0;
  return MVR_OK;
}

bool slot_ok(int s) { return s >= 0 && s < MVR_MAX_SLOTS; }

int cloud_reserve(Ctx *c, Cloud &cl, size_t cap, bool keep)
{
  if (cl.cap >= cap) return MVR_OK;
  MVR_MAY_BLOCK(c, "a cloud has to grow");
  size_t ncap = std::max(cap, cl.cap + cl.cap / 2);
  float4 *np = nullptr;
  MVR_HIP_TRY(c, hipMalloc(&np, ncap * sizeof(float4)));
  if (keep && cl.n) {
    hipError_t e = hipMemcpyAsync(np, cl.pts, cl.n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipFree(np); return set_error(c, MVR_E_HIP, "cloud grow copy", e); }
  }
  if (cl.pts) { (void)hipStreamSynchronize(c->stream); (void)hipFree(cl.pts); }
  cl.pts = np; cl.cap = ncap;
  return MVR_OK;
}

void cloud_free(Cloud &cl)
{
  if (cl.pts) (void)hipFree(cl.pts);
  if (cl.sorted) (void)hipFree(cl.sorted);
  if (cl.tlo) (void)hipFree(cl.tlo);        // (thi / cbox / sbox live in the same block: prepare_index)
  if (cl.nrm) (void)hipFree(cl.nrm);
  if (cl.gsorted) (void)hipFree(cl.gsorted);
  cl.order.reset(); cl.grid.reset();
  cl = Cloud();
}

// dst = T * src was just computed by the f64 transform kernel: remember the pose when src holds its set's canonical
// coordinates and T is a rigid motion (what the grid search needs to walk the set's pose-invariant grid)
// an upper bound (mm) of | Tnew p - Told p | over the points p of the box [lo, hi] (bbox = lo xyz, hi xyz): with c the box's centre,
// r its half diagonal and D = Anew - Aold: | D c + (tnew - told) | + |D|_F r.  Affine poses only (else < 0).
double pose_motion_bound(const double *Told, const double *Tnew, const float bbox[6])
{
  for (const double *T : {Told, Tnew}) if (T[3] != 0.0 || T[7] != 0.0 || T[11] != 0.0 || T[15] != 1.0) return -1.0;
  double cb[3], r2 = 0.0, f2 = 0.0, v[3];
  for (int k = 0; k < 3; ++k) { cb[k] = 0.5 * ((double)bbox[k] + (double)bbox[3 + k]); const double hd = 0.5 * ((double)bbox[3 + k] - (double)bbox[k]); r2 += hd * hd; }
  for (int i = 0; i < 3; ++i) {
    v[i] = Tnew[12 + i] - Told[12 + i];
    for (int k = 0; k < 3; ++k) { const double d = Tnew[4 * k + i] - Told[4 * k + i]; v[i] += d * cb[k]; f2 += d * d; }
  }
  const double b = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + std::sqrt(f2) * std::sqrt(r2);
  return std::isfinite(b) ? b * (1.0 + 1e-9) + 1e-12 : -1.0;
}

void note_pose(Cloud &dst, const Cloud &src, bool src_canonical, const double T[16])
{
  dst.canonical = false; dst.pose_known = false; dst.fin_known = false; dst.parts.clear();
  // how far the copy moves with this pose (seed_delta): from the pose it had last, if that was a pose of the same point set
  const bool chain = src_canonical && &dst != &src && dst.last_pose_set == src.set_id && dst.last_pose_set != 0 && dst.moved >= 0.0 && src.bbox_set == src.set_id;
  const double step = chain ? pose_motion_bound(dst.last_pose, T, src.bbox) : -1.0;
  dst.moved = (chain && step >= 0.0) ? dst.moved + step : -1.0;
  dst.last_pose_set = 0;
  if (!src_canonical || &dst == &src) return;
  if (T[3] != 0.0 || T[7] != 0.0 || T[11] != 0.0 || T[15] != 1.0) return;
  // NEARLY rigid: the poses of a registration are products with PCL-style float 4x4s (lum.getTransformation is an
  // Eigen::Affine3f), each orthonormal to ~1e-7 only, and the product of a few hundred of them is off by 1e-5 and more
  // (a 2e-6 bar here silently sent every pass after the ~200th back to the culled kernel).  The search maps its queries
  // with the TRUE inverse of the 3 x 3 (make_grid_pair), which is right for any invertible matrix; what rigidity buys is
  // that a ball stays a ball.  With G = A^T A and e = |G - I|_F the smallest singular value of A is at least
  // sqrt(1 - e) (Weyl), so the inverse lengthens no distance by more than 1 / sqrt(1 - e): the search widens its ball by
  // that factor.  Accepted up to e = 1e-3 (a ball 0.05 % wider); beyond that the scan is simply not grid-searched.
  double e2 = 0.0;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      const double d = T[4 * a] * T[4 * b] + T[4 * a + 1] * T[4 * b + 1] + T[4 * a + 2] * T[4 * b + 2];
      const double x = d - (a == b ? 1.0 : 0.0);
      e2 += x * x;
    }
  const double e = std::sqrt(e2);
  if (!(e <= 1e-3)) return;
  dst.pose_stretch = 1.0 / std::sqrt(1.0 - e);
  dst.pose_known = true;
  std::memcpy(dst.pose, T, 16 * sizeof(double));
  std::memcpy(dst.last_pose, T, 16 * sizeof(double)); dst.last_pose_set = src.set_id;
  dst.grid = src.grid;
}

float cap_from_max2(double max2)
{
  // smallest float bound that still admits every float d2 with (double)d2 <= max2
  if (!(max2 < (double)FLT_MAX)) return INFINITY;
  float f = (float)max2;
  if ((double)f < max2) f = std::nextafterf(f, INFINITY);
  return f;
}

// forward NN + (optional) reciprocal pass for the source queries at positions
// [qb, qb+qn) (positions in the source's Morton order in culled mode, original
// indices in brute-force mode: either way the ranges [0,n) partition the
// queries, and every per-pair sum is additive over such a partition).
// Leaves keys[], slot[], rkeys[] ready for pass1.
// qperm / tinv: the culled mode's orderings; slot / count: the brute-force mode's compacted reverse queries
// (the culled mode searches the flagged targets in place: reverse keys are indexed by sorted target position)
struct SearchPlan { const uint32_t *qperm = nullptr; const uint32_t *tinv = nullptr; const uint32_t *slot = nullptr; const uint32_t *count = nullptr; };

// forward pass: keys[i] = (d2, LOCAL target index) of source query i (kKeyInit beyond the cap / no target)
int search_forward(Ctx *c, Cloud &src, Cloud &tgt, size_t qb, size_t qn, double max_dist, bool fma, double *evals, SearchPlan *plan)
{
  const size_t ns = src.n, nt = tgt.n;
  const double max2 = max_dist * max_dist;
  *plan = SearchPlan();
  if (int rc = ensure(c, c->keys, c->keys_cap, ns)) return rc;
  if (int rc = ensure(c, c->match, c->match_cap, ns)) return rc;
  if (c->nn_mode != 0 && ns > 0 && nt > 0) {
    if (int rc = ensure_index(c, src)) return rc;
    if (int rc = ensure_index(c, tgt)) return rc;
    plan->qperm = src.order->perm;
    plan->tinv = tgt.order->inv;
    // keys are written (not min-combined) by exactly one wave per query
    return launch_nn_cull(c, src, qb, qn, nullptr, tgt, cap_from_max2(max2), fma, c->keys);
  }
  if (int rc = launch_fill_u64(c, c->keys + qb, qn, kKeyInit)) return rc;
  if (int rc = launch_nn(c, src.pts, qb, qn, nullptr, nullptr, tgt.pts, nt, fma, c->keys)) return rc;
  if (evals) *evals += (double)qn * (double)nt;
  return MVR_OK;
}

// reciprocal pass on the keys in c->keys (from search_forward, or imported after a reduction over ranks):
// the distinct matched targets are searched in the WHOLE source.  Leaves slot[], rkeys[] ready for pass1.
int search_reciprocal(Ctx *c, Cloud &src, Cloud &tgt, size_t qb, size_t qn, double max_dist, bool fma, SearchPlan *plan)
{
  const size_t ns = src.n, nt = tgt.n;
  const double max2 = max_dist * max_dist;
  plan->slot = plan->count = nullptr;
  if (nt == 0 || qn == 0) return MVR_OK;
  if (c->nn_mode != 0 && ns > 0) {
    if (int rc = ensure(c, c->flags, c->flags_cap, nt)) return rc;
    // a matched target has a source point at its forward distance: its reverse search starts from that bound
    if (int rc = ensure(c, c->bound, c->bound_cap, nt)) return rc;
    if (int rc = launch_seed_bounds(c, c->keys, plan->qperm, qb, qn, max2, plan->tinv, nt, c->bound)) return rc;
    if (nt >= (size_t)std::max(0, c->inplace_ratio) * qn) {
      // the target is much larger than the set of queries (merged target of the sequential mode): compact the
      // matched targets into an ordered list first, or most blocks of the reverse launch would find nothing to do
      const size_t nl = std::min(qn, nt);
      if (int rc = ensure(c, c->slot, c->slot_cap, nt)) return rc;
      if (int rc = ensure(c, c->list, c->list_cap, nl)) return rc;
      if (int rc = ensure(c, c->rkeys, c->rkeys_cap, nl)) return rc;
      plan->slot = c->slot; plan->count = c->count;
      if (int rc = launch_mark_sorted(c, c->keys, plan->qperm, qb, qn, max2, plan->tinv, nt, c->flags, c->list, c->count, c->slot)) return rc;
      CullPair p = make_cull_pair(tgt, 0, nl, nullptr, src, c->rkeys);
      p.qlist = c->list; p.qcount = c->count; p.qbound = c->bound;
      return launch_nn_cull_batch(c, &p, 1, cap_from_max2(max2), fma);
    }
    // matched targets are flagged in sorted space and searched IN PLACE (no compaction): rkeys[sorted position]
    if (int rc = ensure(c, c->rkeys, c->rkeys_cap, nt)) return rc;
    if (int rc = launch_flag_matched(c, c->keys, plan->qperm, qb, qn, max2, plan->tinv, nt, c->flags)) return rc;
    if (ns > 0xFFFFFFF0ull || nt > 0xFFFFFFF0ull) return set_error(c, MVR_E_ARG, "cloud too large for 32-bit indices");
    CullPair p = make_cull_pair(tgt, 0, nt, c->flags, src, c->rkeys);
    p.qbound = c->bound;
    return launch_nn_cull_batch(c, &p, 1, cap_from_max2(max2), fma);
  }
  const size_t nl = std::min(qn, nt);
  if (int rc = ensure(c, c->slot, c->slot_cap, nt)) return rc;
  if (int rc = ensure(c, c->list, c->list_cap, nl)) return rc;
  if (int rc = ensure(c, c->rkeys, c->rkeys_cap, nl)) return rc;
  plan->slot = c->slot; plan->count = c->count;
  MVR_HIP_TRY(c, hipMemsetAsync(c->slot, 0xFF, nt * sizeof(uint32_t), c->stream));
  MVR_HIP_TRY(c, hipMemsetAsync(c->count, 0, sizeof(uint32_t), c->stream));
  if (int rc = launch_mark(c, c->keys, qb, qn, max2, c->slot, c->list, c->count)) return rc;
  if (int rc = launch_fill_u64(c, c->rkeys, nl, kKeyInit)) return rc;
  return launch_nn(c, tgt.pts, 0, nl, c->list, c->count, src.pts, ns, fma, c->rkeys);
}

int run_search(Ctx *c, Cloud &src, Cloud &tgt, size_t qb, size_t qn, double max_dist, bool reciprocal, bool fma,
               double *evals, SearchPlan *plan)
{
  if (int rc = search_forward(c, src, tgt, qb, qn, max_dist, fma, evals, plan)) return rc;
  if (reciprocal && src.n > 0) { if (int rc = search_reciprocal(c, src, tgt, qb, qn, max_dist, fma, plan)) return rc; }
  return MVR_OK;
}

SegTable seg_table(const Cloud &cl)
{
  SegTable st;
  st.n = (uint32_t)cl.segs.size();
  for (uint32_t k = 0; k < st.n; ++k) { st.lb[k] = cl.segs[k].local_begin; st.cnt[k] = cl.segs[k].count; st.gb[k] = cl.segs[k].global_begin; }
  return st;
}

int read_moments(Ctx *c, size_t n_doubles)
{
  MVR_HIP_TRY(c, hipMemcpyAsync(c->h_moments, c->moments, n_doubles * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MVR_OK;
}

double now_ms()
{
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// the align's variant: the last sums launch has stored the row and then the word `seq` into mapped pinned memory itself
// (launch_pass2's host_out); the host spins on the word, looks at the stream every few thousand rounds so that a failed
// launch ends the wait, and gives up after wait_timeout_ms.  The row lands in h_moments, where read_moments leaves it.
int ensure_align_row(Ctx *c)
{
  if (c->h_align) return MVR_OK;
  if (hipHostMalloc(reinterpret_cast<void **>(&c->h_align), 66 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    (void)hipGetLastError(); c->h_align = nullptr; return MVR_E_HIP;
  }
  std::memset(c->h_align, 0, 66 * sizeof(double));
  if (hipHostGetDevicePointer(reinterpret_cast<void **>(&c->d_align), c->h_align, 0) != hipSuccess) {
    (void)hipGetLastError(); (void)hipHostFree(c->h_align); c->h_align = nullptr; c->d_align = nullptr; return MVR_E_HIP;
  }
  return MVR_OK;
}

int spin_moments(Ctx *c, uint32_t seq, size_t n_doubles)
{
  const uint32_t *word = reinterpret_cast<const uint32_t *>(c->h_align + 64);
  const double t0 = now_ms();
  for (unsigned n = 1;; ++n) {
    if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == seq) break;
    __builtin_ia32_pause();
    if ((n & 0xFFFu) == 0u) {
      const hipError_t e = hipStreamQuery(c->stream);
      if (e != hipSuccess && e != hipErrorNotReady) return set_error(c, MVR_E_HIP, "an iteration of mvr_icp_align failed on the device", e);
      if (e == hipSuccess && __atomic_load_n(word, __ATOMIC_ACQUIRE) != seq) {
        // the stream has drained and the word has not come: whatever kept the launch from running is the stream's to report
        MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == seq) break;
        return set_error(c, MVR_E_HIP, "mvr_icp_align: the sums launch left no row");
      }
      if (now_ms() - t0 > (double)c->wait_timeout_ms) return set_error(c, MVR_E_HIP, "an iteration of mvr_icp_align did not finish in time");
    }
  }
  std::memcpy(c->h_moments, c->h_align, n_doubles * sizeof(double));
  return MVR_OK;
}

}  // namespace
}  // namespace mvr

using namespace mvr;

#define API __attribute__((visibility("default")))
#define CTX(p) reinterpret_cast<Ctx *>(p)

extern "C" {

API const char *mvr_strerror(int s)
{
  switch (s) {
    case MVR_OK: return "ok";
    case MVR_E_ARG: return "invalid argument";
    case MVR_E_HIP: return "HIP runtime error or no usable GPU";
    case MVR_E_NOCORR: return "not enough correspondences (< 3)";
    case MVR_E_NOMEM: return "out of memory";
    case MVR_E_SINGULAR: return "singular linear system";
    case MVR_E_RCCL: return "RCCL error or RCCL library not available";
    default: return "unknown status";
  }
}

API const char *mvr_last_error(const mvr_ctx *ctx) { return ctx ? reinterpret_cast<const Ctx *>(ctx)->last_error.c_str() : ""; }

// low_priority: the stream this context creates for itself gets the lowest priority (worker contexts)
static int ctx_create_impl(mvr_ctx **out, int device_id, void *hip_stream, bool low_priority);

API int mvr_ctx_create_on_stream(mvr_ctx **out, int device_id, void *hip_stream)
{
  return ctx_create_impl(out, device_id, hip_stream, false);
}

static int ctx_create_impl(mvr_ctx **out, int device_id, void *hip_stream, bool low_priority)
{
  if (!out) return MVR_E_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MVR_E_HIP;   // no GPU: fail loudly, no fallback
  if (device_id < 0 || device_id >= ndev) return MVR_E_ARG;
  if (hipSetDevice(device_id) != hipSuccess) return MVR_E_HIP;
  Ctx *c = new (std::nothrow) Ctx();
  if (!c) return MVR_E_NOMEM;
  c->device = device_id;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { delete c; return MVR_E_HIP; }
  c->n_cu = prop.multiProcessorCount; c->clock_mhz = prop.clockRate / 1000; c->name = prop.gcnArchName;
  if (const char *m = std::getenv("MVR_NN_MODE")) c->nn_mode = std::atoi(m);   // 0 brute force, 1 culled (default)
  if (const char *m = std::getenv("MVR_CULL_Q")) c->cull_q = std::atoi(m);     // 64-query groups per set: 1, 2 (0 = auto)
  if (const char *m = std::getenv("MVR_CULL_W")) c->cull_w = std::atoi(m);     // waves per query set: 1, 2, 4 (default)
  if (const char *m = std::getenv("MVR_CULL_SLICES")) c->cull_slices = std::atoi(m);   // XCD dealing of a pair's query sets: 1, 2, 4, 8 (0 = auto)
  if (const char *m = std::getenv("MVR_SEED_FORWARD")) c->seed_forward = std::atoi(m) != 0;
  if (const char *m = std::getenv("MVR_PARTS_LANES")) c->parts_lanes = std::atoi(m);
  if (const char *m = std::getenv("MVR_PARTS_MAX_ROWS")) c->parts_max_rows = std::max(1, std::atoi(m));
  if (const char *m = std::getenv("MVR_GRID_DEBUG")) c->grid_debug = std::atoi(m) != 0;
  if (const char *m = std::getenv("MVR_SEQ_SEED")) c->seq_seed = std::atoi(m) != 0;
  if (const char *m = std::getenv("MVR_ALIGN_SPIN")) c->align_spin = std::atoi(m) != 0;      // 0: an align's iteration row by copy + synchronise
  if (const char *m = std::getenv("MVR_SEQ_SEARCH")) c->seq_search = std::max(0, std::min(3, std::atoi(m)));    // align against a target made of posed scans: 1 = through the scans' grids, 0 = culled kernel
  if (const char *m = std::getenv("MVR_RING_SEARCH")) c->ring_search = std::atoi(m);   // fused pass: 1 = grid search for bounded queries, 0 = culled kernel only
  if (const char *m = std::getenv("MVR_INPLACE_RATIO")) c->inplace_ratio = std::atoi(m);
  if (const char *m = std::getenv("MVR_PAIR_FUSED")) c->pair_fused = std::atoi(m) != 0;
  if (const char *m = std::getenv("MVR_PIPELINE")) c->pipeline = std::atoi(m) != 0;      // 0: no pass is enqueued ahead of its poses
  if (const char *m = std::getenv("MVR_GRID_INDEX")) c->grid_index = std::max(0, std::min(2, std::atoi(m)));      // 0 dense cell starts, 1 compact, 2 by size (default)
  if (const char *m = std::getenv("MVR_GRID_STAGE")) c->grid_stage = std::max(0, std::min(2, std::atoi(m)));      // the staged walk: 0 off, 1 forward launches, 2 all (default)
  if (const char *m = std::getenv("MVR_POSED_REFRESH")) c->posed_refresh = std::atoi(m) != 0;
  if (const char *m = std::getenv("MVR_PAIR_GROUPS")) c->pair_groups = std::max(1, std::min(8, std::atoi(m)));
  if (const char *m = std::getenv("MVR_PAIR_STREAMS")) c->pair_streams = std::max(1, std::min(16, std::atoi(m)));
  if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
  else {
    // A worker's stream runs one group of pairs beside the caller's: at the LOWEST priority the caller's group gets the
    // compute units first and finishes its chain of small kernels without queueing behind the worker's searches, the
    // worker fills what is left (measured on the ring step: 0.978 -> 0.968 ms; the HIGHEST priority: 0.98).
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    const hipError_t e = low_priority ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, lo)
                                      : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return MVR_E_HIP; }
    c->own_stream = true;
  }
  if (hipMalloc(&c->count, 64) != hipSuccess || hipMalloc(&c->evals, (2 * kEvalRegion + kTraceRec * kTraceBlocks) * sizeof(uint64_t)) != hipSuccess ||
      hipMalloc(&c->bbox, 64) != hipSuccess || hipMalloc(&c->moments, 64 * sizeof(double)) != hipSuccess ||
      hipHostMalloc(&c->h_moments, 64 * sizeof(double)) != hipSuccess ||
      hipHostMalloc(&c->h_counts, kProfCounts * sizeof(uint64_t)) != hipSuccess ||
      hipHostMalloc(&c->h_evals, kEvalRegion * sizeof(uint64_t)) != hipSuccess) {
    mvr_ctx_destroy(reinterpret_cast<mvr_ctx *>(c));
    return MVR_E_HIP;
  }
  (void)hipMemset(c->evals, 0, (2 * kEvalRegion + kTraceRec * kTraceBlocks) * sizeof(uint64_t));
  *out = reinterpret_cast<mvr_ctx *>(c);
  return MVR_OK;
}

API int mvr_ctx_create(mvr_ctx **out, int device_id) { return mvr_ctx_create_on_stream(out, device_id, nullptr); }

API int mvr_ctx_destroy(mvr_ctx *ctx)
{
  if (!ctx) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)mvr_ctx_comm_destroy(ctx);         // (a world's communicators are lent, not owned: just dropped)
  for (Ctx *w : c->workers) {
    w->slots[0] = Cloud(); w->slots[1] = Cloud();      // borrowed views: nothing to free
    (void)mvr_ctx_destroy(reinterpret_cast<mvr_ctx *>(w));
  }
  c->workers.clear();
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  for (auto &r : c->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  for (auto &s : c->slots) cloud_free(s);
  for (auto &kv : c->seq_seeds) if (kv.second.d) (void)hipFree(kv.second.d);
  c->seq_seeds.clear();
  if (c->seed_bound) (void)hipFree(c->seed_bound);
  c->orders.clear(); c->grids.clear();
  if (c->order_pool) c->order_pool->close();          // orderings that outlive the context free their buffers themselves
  void *bufs[] = {c->keys, c->rkeys, c->slot, c->list, c->match, c->flags, c->count, c->evals, c->partials, c->moments,
                  c->codes_a, c->codes_b, c->idx_a, c->cub_tmp, c->bbox, c->batch_table, c->bcert, c->bkeys, c->brkeys, c->bbound, c->bound, c->dn_arena, c->bpartials, c->blist, c->bslot, c->bchunks, c->dist_table, c->bheavy, c->bwide, c->bwide_count, c->bcull_sets};
  for (void *b : bufs) if (b) (void)hipFree(b);
  if (c->stage_stat) (void)hipFree(c->stage_stat);
  if (c->proj_rows) (void)hipFree(c->proj_rows);
  if (c->oscratch) (void)hipFree(c->oscratch);
  if (c->d_parts) (void)hipFree(c->d_parts);
  if (c->h_parts) (void)hipHostFree(c->h_parts);
  if (c->side_stream) { (void)hipStreamSynchronize(c->side_stream); (void)hipStreamDestroy(c->side_stream); }
  if (c->side_after) (void)hipEventDestroy(c->side_after);
  if (c->scratch_event) (void)hipEventDestroy(c->scratch_event);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->h_stall) (void)hipHostFree(c->h_stall);
  if (c->seq_keys) (void)hipFree(c->seq_keys);
  if (c->seq_row) (void)hipFree(c->seq_row);
  if (c->h_pose_in) (void)hipHostFree(c->h_pose_in);
  if (c->pose_tab) (void)hipFree(c->pose_tab);
  if (c->done_counter) (void)hipFree(c->done_counter);
  if (c->gate) { if (c->gate_is_signal) (void)hipFree(c->gate); else (void)hipHostFree(c->gate); }
  if (c->h_done) (void)hipHostFree(c->h_done);
  if (c->h_moments) (void)hipHostFree(c->h_moments);
  if (c->h_align) (void)hipHostFree(c->h_align);
  if (c->h_table) (void)hipHostFree(c->h_table);
  if (c->h_counts) (void)hipHostFree(c->h_counts);
  if (c->h_evals) (void)hipHostFree(c->h_evals);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return MVR_OK;
}

API int mvr_ctx_sync(mvr_ctx *ctx)
{
  if (!ctx) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MVR_OK;
}

API int mvr_device_info(mvr_ctx *ctx, char *name, size_t cap, int *n_cu, int *clock_mhz)
{
  if (!ctx) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  if (name && cap) { std::snprintf(name, cap, "%s", c->name.c_str()); }
  if (n_cu) *n_cu = c->n_cu;
  if (clock_mhz) *clock_mhz = c->clock_mhz;
  return MVR_OK;
}

// ------------------------------------------------------------------- clouds

API int mvr_cloud_reserve(mvr_ctx *ctx, int slot, size_t cap)
{
  if (!ctx || !slot_ok(slot)) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  return cloud_reserve(c, c->slots[slot], cap, true);
}

API int mvr_cloud_upload(mvr_ctx *ctx, int slot, const float *xyz, size_t n, size_t stride)
{
  if (!ctx || !slot_ok(slot) || (n && !xyz) || (stride != 16 && stride != 12)) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  Cloud &cl = c->slots[slot];
  cl.n = 0;
  cl.has_normals = false;
  cl.segs.clear();
  new_point_set(c, cl);
  if (int rc = cloud_reserve(c, cl, n, false)) return rc;
  if (n) {
    if (stride == 16) {
      MVR_HIP_TRY(c, hipMemcpyAsync(cl.pts, xyz, n * 16, hipMemcpyHostToDevice, c->stream));
    } else {
      float *tmp = nullptr;
      MVR_HIP_TRY(c, hipMalloc(&tmp, n * 12));
      hipError_t e = hipMemcpyAsync(tmp, xyz, n * 12, hipMemcpyHostToDevice, c->stream);
      int rc = (e == hipSuccess) ? launch_unpack_xyz(c, tmp, cl.pts, n) : set_error(c, MVR_E_HIP, "upload", e);
      (void)hipStreamSynchronize(c->stream);
      (void)hipFree(tmp);
      if (rc) return rc;
    }
    // the caller keeps ownership of xyz and may change it right after return: the upload waits for its copy anyway -- and
    // brings the cloud's bounding box back with it (two small launches behind the copy): the grid build of a registration's
    // first pass then starts without a round trip of its own (0.4 ms of a host-bound pass)
    if (c->nn_mode != 0 && n >= 4 && cloud_bbox(c, cl.pts, n, cl.bbox) == MVR_OK) cl.bbox_set = cl.set_id;
    else MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  cl.n = n;
  cl.canonical = true;            // these are the point set's canonical coordinates (the grid search's frame of reference)
  return MVR_OK;
}

API int mvr_cloud_download(mvr_ctx *ctx, int slot, float *xyz, size_t cap, size_t stride, size_t *n)
{
  if (!ctx || !slot_ok(slot) || (stride != 16 && stride != 12)) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  const Cloud &cl = c->slots[slot];
  if (n) *n = cl.n;
  const size_t m = std::min(cap, cl.n);
  if (!m || !xyz) return MVR_OK;
  if (stride == 16) {
    MVR_HIP_TRY(c, hipMemcpyAsync(xyz, cl.pts, m * 16, hipMemcpyDeviceToHost, c->stream));
    MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
  } else {
    float *tmp = nullptr;
    MVR_HIP_TRY(c, hipMalloc(&tmp, m * 12));
    int rc = launch_pack_xyz(c, cl.pts, tmp, m);
    hipError_t e = hipMemcpyAsync(xyz, tmp, m * 12, hipMemcpyDeviceToHost, c->stream);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (rc) return rc;
    if (e != hipSuccess) return set_error(c, MVR_E_HIP, "download", e);
  }
  return MVR_OK;
}

// EXTENSION (point-to-plane): normals ride along with the points
API int mvr_cloud_upload_normals(mvr_ctx *ctx, int slot, const float *nxyz, size_t n, size_t stride)
{
  if (!ctx || !slot_ok(slot) || (n && !nxyz) || (stride != 16 && stride != 12)) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  Cloud &cl = c->slots[slot];
  if (n != cl.n) return set_error(c, MVR_E_ARG, "normals: one per point expected");
  cl.has_normals = false;
  if (n == 0) return MVR_OK;
  if (int rc = ensure(c, cl.nrm, cl.nrm_cap, n)) return rc;
  if (stride == 16) {
    MVR_HIP_TRY(c, hipMemcpyAsync(cl.nrm, nxyz, n * 16, hipMemcpyHostToDevice, c->stream));
    MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
  } else {
    float *tmp = nullptr;
    MVR_HIP_TRY(c, hipMalloc(&tmp, n * 12));
    hipError_t e = hipMemcpyAsync(tmp, nxyz, n * 12, hipMemcpyHostToDevice, c->stream);
    int rc = (e == hipSuccess) ? launch_unpack_xyz(c, tmp, cl.nrm, n) : set_error(c, MVR_E_HIP, "upload normals", e);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (rc) return rc;
  }
  cl.has_normals = true;
  return MVR_OK;
}

API int mvr_cloud_download_normals(mvr_ctx *ctx, int slot, float *nxyz, size_t cap, size_t stride, size_t *n)
{
  if (!ctx || !slot_ok(slot) || (stride != 16 && stride != 12)) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  const Cloud &cl = c->slots[slot];
  const size_t have = cl.has_normals ? cl.n : 0;
  if (n) *n = have;
  const size_t m = std::min(cap, have);
  if (!m || !nxyz) return MVR_OK;
  if (stride == 16) {
    MVR_HIP_TRY(c, hipMemcpyAsync(nxyz, cl.nrm, m * 16, hipMemcpyDeviceToHost, c->stream));
    MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
  } else {
    float *tmp = nullptr;
    MVR_HIP_TRY(c, hipMalloc(&tmp, m * 12));
    int rc = launch_pack_xyz(c, cl.nrm, tmp, m);
    hipError_t e = hipMemcpyAsync(nxyz, tmp, m * 12, hipMemcpyDeviceToHost, c->stream);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (rc) return rc;
    if (e != hipSuccess) return set_error(c, MVR_E_HIP, "download normals", e);
  }
  return MVR_OK;
}

API int mvr_cloud_size(mvr_ctx *ctx, int slot, size_t *n)
{
  if (!ctx || !slot_ok(slot) || !n) return MVR_E_ARG;
  *n = CTX(ctx)->slots[slot].n;
  return MVR_OK;
}

API int mvr_cloud_clear(mvr_ctx *ctx, int slot)
{
  if (!ctx || !slot_ok(slot)) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  c->slots[slot].n = 0;
  c->slots[slot].has_normals = false;
  c->slots[slot].segs.clear();
  new_point_set(c, c->slots[slot]);
  return MVR_OK;
}

API int mvr_cloud_denoise(mvr_ctx *ctx, int slot, int segment_threshold, double triangle_length, size_t *n_kept,
                          size_t *n_components, uint32_t *kept_index)
{
  if (!ctx || !slot_ok(slot)) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  return denoise_cloud(c, c->slots[slot], segment_threshold, triangle_length, n_kept, n_components, kept_index);
}

API int mvr_cloud_copy(mvr_ctx *ctx, int dst, int src)
{
  if (!ctx || !slot_ok(dst) || !slot_ok(src)) return MVR_E_ARG;
  if (dst == src) return MVR_OK;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  Cloud &d = c->slots[dst]; const Cloud &s = c->slots[src];
  d.n = 0;
  if (int rc = cloud_reserve(c, d, s.n, false)) return rc;
  if (s.n) {
    ProfScope ps(c, MVR_K_XFORM, 32.0 * (double)s.n);
    MVR_HIP_TRY(c, hipMemcpyAsync(d.pts, s.pts, s.n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
  }
  d.n = s.n;
  inherit_point_set(d, s);
  d.canonical = s.canonical; d.pose_known = s.pose_known; d.pose_stretch = s.pose_stretch; std::memcpy(d.pose, s.pose, sizeof d.pose); d.grid = s.grid;
  d.fin_known = s.fin_known; std::memcpy(d.fin, s.fin, sizeof d.fin);
  d.parts = s.parts;
  for (GridPart &gp : d.parts) gp.gs_filled = false;          // (the grid-ordered coordinates are not copied: written again at the first search)
  d.segs = s.segs;
  d.has_normals = false;
  if (s.has_normals && s.n) {
    if (int rc = ensure(c, d.nrm, d.nrm_cap, s.n)) return rc;
    MVR_HIP_TRY(c, hipMemcpyAsync(d.nrm, s.nrm, s.n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    d.has_normals = true;
  }
  return MVR_OK;
}

API int mvr_cloud_append(mvr_ctx *ctx, int dst, int src)
{
  if (!ctx || !slot_ok(dst) || !slot_ok(src)) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  Cloud &d = c->slots[dst];
  const size_t add = c->slots[src].n;         // read before a self-append grows it
  // What the grown cloud is made of, when both sides are posed copies of scans (the sequential mode's model: view 0 posed,
  // then one aligned view after the other): recorded so that mvr_icp_align can search it through the scans' grids.
  auto provenance = [](const Cloud &x, size_t base, GridPart *gp) {
    if (!(x.pose_known || x.fin_known) || x.n == 0 || !x.segs.empty()) return false;
    gp->set_id = x.set_id; gp->n = x.n; gp->base = base; gp->kind = x.fin_known ? 2 : 1;
    std::memcpy(gp->pose, x.pose, sizeof gp->pose); std::memcpy(gp->fin, x.fin, sizeof gp->fin);
    gp->grid = x.grid; gp->gs_filled = false;
    return true;
  };
  std::vector<GridPart> parts_after;
  {
    bool ok = dst != src && d.segs.empty() && add > 0;
    if (ok) {
      if (!d.parts.empty()) parts_after = d.parts;
      else if (d.n > 0) { GridPart p0; ok = provenance(d, 0, &p0); if (ok) parts_after.push_back(p0); }
    }
    GridPart pn;
    if (ok && provenance(c->slots[src], d.n, &pn)) parts_after.push_back(pn); else parts_after.clear();
  }
  if (!d.segs.empty()) {                       // a shard: the appended points continue its global numbering
    uint32_t gend = 0;
    for (const Seg &sg : d.segs) gend = std::max(gend, sg.global_begin + sg.count);
    if ((int)d.segs.size() >= kMaxSegs) return set_error(c, MVR_E_ARG, "too many segments in a sharded cloud");
    d.segs.push_back(Seg{(uint32_t)d.n, (uint32_t)add, gend});
  }
  if (int rc = cloud_reserve(c, d, d.n + add, true)) return rc;
  const Cloud &s = c->slots[src];
  if (add) {
    d.mgrid_ok = false;
    ProfScope ps(c, MVR_K_XFORM, 32.0 * (double)add);
    MVR_HIP_TRY(c, hipMemcpyAsync(d.pts + d.n, s.pts, add * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    // a different point set.  Its ordering: the old one extended by the appended scan's own (the points that were
    // there did not move: registrator.cpp:576 only ever appends), or -- when that is not possible -- rebuilt on next use
    if (dst != src && d.segs.empty() && extend_point_set(c, d, d.n, s)) {
      // a different point set all the same: what it is now DEFINES it (as new_point_set says for the other branch); a pose
      // or a grid of the set it grew from is not its own
      // (for a cloud made of parts gcoords_valid says "the parts' gs_filled flags can be trusted": the stretches of
      // gsorted[] already written stay valid here -- those points did not move)
      const bool keep_gs = d.gcoords_valid && !parts_after.empty() && !d.parts.empty();
      d.forget_pose(); d.canonical = true; d.gcoords_valid = keep_gs;
    } else new_point_set(c, d);
    if (!parts_after.empty()) d.parts.swap(parts_after);
    // normals survive only if both parts carry them
    const bool keep = s.has_normals && (d.has_normals || d.n == 0);
    if (keep) {
      if (d.nrm_cap < d.n + add) {
        float4 *nn = nullptr;
        const size_t ncap = std::max(d.n + add, d.cap);
        MVR_HIP_TRY(c, hipMalloc(&nn, ncap * sizeof(float4)));
        if (d.n) MVR_HIP_TRY(c, hipMemcpyAsync(nn, d.nrm, d.n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
        MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (d.nrm) (void)hipFree(d.nrm);
        d.nrm = nn; d.nrm_cap = ncap;
      }
      MVR_HIP_TRY(c, hipMemcpyAsync(d.nrm + d.n, s.nrm, add * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    }
    d.has_normals = keep;
  }
  d.n += add;
  return MVR_OK;
}

API int mvr_cloud_transform(mvr_ctx *ctx, int dst, int src, const double T[16])
{
  if (!ctx || !slot_ok(dst) || !slot_ok(src) || !T) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  // a posed COPY of a plain cloud that is going to be searched: the batch path with one entry -- where the set's ordering exists,
  // ONE launch poses the source's sorted copy in order and writes the points in original order (and the grid-ordered copy) with
  // it, instead of a transform launch now and a gather through the permutation at the first search (the sequential mode poses
  // its source this way before every align: two launches and a dispatch gap less)
  if (dst != src && c->nn_mode != 0 && c->posed_refresh && !c->slots[src].has_normals && c->slots[src].segs.empty() && c->slots[src].n)
    return mvr_cloud_transform_batch(ctx, 1, &dst, &src, T);
  const size_t n = c->slots[src].n;
  if (dst != src) { c->slots[dst].n = 0; if (int rc = cloud_reserve(c, c->slots[dst], n, false)) return rc; }
  if (int rc = launch_transform_f64(c, c->slots[src].pts, c->slots[dst].pts, n, T)) return rc;
  c->slots[dst].n = n;
  const bool src_canon = c->slots[src].canonical;
  if (dst != src) { inherit_point_set(c->slots[dst], c->slots[src]); c->slots[dst].segs = c->slots[src].segs; }
  else c->slots[dst].forget_pose();      // moved in place: no longer the upload coordinates
  c->slots[dst].stale_coords();
  note_pose(c->slots[dst], c->slots[src], src_canon && dst != src, T);
  if (c->slots[src].has_normals && n) {
    if (int rc = ensure(c, c->slots[dst].nrm, c->slots[dst].nrm_cap, n)) return rc;
    if (int rc = launch_rotate_normals_f64(c, c->slots[src].nrm, c->slots[dst].nrm, n, T)) return rc;
  }
  c->slots[dst].has_normals = c->slots[src].has_normals;
  return MVR_OK;
}

API int mvr_cloud_transform_batch(mvr_ctx *ctx, int count, const int *dst, const int *src, const double *T)
{
  if (!ctx || count < 0 || (count && (!dst || !src || !T))) return MVR_E_ARG;
  for (int k = 0; k < count; ++k) if (!slot_ok(dst[k]) || !slot_ok(src[k])) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  // All entries run in ONE launch (blockIdx.y = cloud): a slot written twice, or written by one entry and read by
  // another, would be a data race (and a destination that has to grow would free a buffer another entry still reads).
  // Refused up front; dst[k] == src[k] (in place) is fine.
  for (int j = 0; j < count; ++j)
    for (int k = 0; k < count; ++k)
      if (j != k && (dst[j] == dst[k] || dst[j] == src[k]))
        return set_error(c, MVR_E_ARG, "mvr_cloud_transform_batch: a destination slot appears twice or is another entry's source");
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  std::vector<const float4 *> in((size_t)count);
  std::vector<float4 *> out((size_t)count);
  std::vector<size_t> n((size_t)count);
  std::vector<char> single((size_t)count, 0);
  for (int k = 0; k < count; ++k) {
    if (c->slots[src[k]].has_normals) {          // normals ride along: the one-cloud path rotates them too
      if (int rc = mvr_cloud_transform(ctx, dst[k], src[k], T + (size_t)k * 16)) return rc;
      single[k] = 1;
      continue;
    }
    n[k] = c->slots[src[k]].n;
    if (dst[k] != src[k]) { c->slots[dst[k]].n = 0; if (int rc = cloud_reserve(c, c->slots[dst[k]], n[k], false)) return rc; }
  }
  for (int k = 0; k < count; ++k) {              // pointers are taken only after every reserve is done
    if (single[k]) { n[k] = 0; in[k] = nullptr; out[k] = nullptr; continue; }
    in[k] = c->slots[src[k]].pts; out[k] = c->slots[dst[k]].pts;
  }
  // bookkeeping first (host side only): the destinations are posed copies of their sources' point sets
  for (int k = 0; k < count; ++k) {
    if (!in[k]) continue;
    c->slots[dst[k]].n = n[k];
    const bool src_canon = c->slots[src[k]].canonical;
    // (the destination takes the source's segment table with the point set, as the one-cloud path does: a slot that WAS a shard
    // must not keep its stale table -- mvr_cloud_append, provenance() and seg_table() all read it)
    if (dst[k] != src[k]) { inherit_point_set(c->slots[dst[k]], c->slots[src[k]]); c->slots[dst[k]].segs = c->slots[src[k]].segs; }
    else c->slots[dst[k]].forget_pose();
    c->slots[dst[k]].stale_coords();
    if (c->pose_from_table && c->slots[dst[k]].pose_dev && dst[k] != src[k]) {
      // a pass enqueued ahead of its poses: the kernels read pose, inverse and stretch from the destination's device
      // record, which pose_prep_kernel fills for ANY invertible matrix (a pose that is not nearly rigid gets a wider
      // ball, not another kernel); the host's copy of the pose follows when the solve has produced it (ring_passes)
      Cloud &d = c->slots[dst[k]];
      d.canonical = false; d.pose_known = src_canon; d.grid = src_canon ? c->slots[src[k]].grid : nullptr; d.posed_by_table = true;
    } else
    note_pose(c->slots[dst[k]], c->slots[src[k]], src_canon && dst[k] != src[k], T + (size_t)k * 16);
    c->slots[dst[k]].has_normals = false;
  }
  // Culled search: the posed copies are about to be searched -- bring their index up to date here, from the
  // sources' sorted copies (read in order) instead of a gather through the permutation at the first search.
  // Where the ordering of the point set exists already (every pose but a scan's first) the SAME launch writes
  // the posed points in original order too, and the plain transform launch below has nothing left to do.
  std::vector<int> cand;
  if (c->nn_mode != 0 && c->posed_refresh)
    for (int k = 0; k < count; ++k) {
      if (!in[k] || dst[k] == src[k] || n[k] == 0) continue;
      cand.push_back(k);
    }
  std::vector<char> done((size_t)count, 0);
  for (int pass = 0; pass < 2; ++pass) {                    // pass 0: with the points, before the transform launch; pass 1: the rest, after it
    std::vector<Cloud *> d, s; std::vector<double> Ts; std::vector<int> which;
    for (int k : cand) {
      Cloud &dc = c->slots[dst[k]];
      if (done[k]) continue;
      if (pass == 0 && !(dc.order && dc.order->n == n[k])) continue;      // no ordering yet: it is built from the posed points (pass 1)
      d.push_back(&dc); s.push_back(&c->slots[src[k]]); Ts.insert(Ts.end(), T + (size_t)k * 16, T + (size_t)k * 16 + 16);
      which.push_back(k);
    }
    std::vector<char> handled(d.size(), 0);
    // (a pass queued ahead of its poses leaves the points in original order out: 77 MB of the step's 230 MB of posing
    // traffic that no kernel of a fused pass reads -- ring_passes writes them once at the end of the stretch)
    const bool skip_pts = pass == 0 && c->pose_from_table && c->skip_posed_pts;
    if (!d.empty()) { if (int rc = refresh_posed_batch(c, (int)d.size(), d.data(), s.data(), Ts.data(), pass == 0 && !skip_pts, handled.data())) return rc; }
    if (skip_pts) for (size_t i = 0; i < which.size(); ++i) if (handled[i]) c->slots[dst[which[i]]].pts_stale = true;
    if (pass == 0) {
      for (size_t i = 0; i < which.size(); ++i) if (handled[i]) { done[which[i]] = 1; n[which[i]] = 0; }    // posed by the refresh launch
      std::vector<const Mat44d *> Tp((size_t)count, nullptr);
      if (c->pose_from_table) {
        bool any = false;
        for (int k = 0; k < count; ++k) if (in[k] && n[k] && c->slots[dst[k]].pose_dev && dst[k] != src[k]) { Tp[k] = &c->slots[dst[k]].pose_dev->T; any = true; }
        if (any) { if (int rc = ensure_pose_table(c)) return rc; }
      }
      if (int rc = launch_transform_f64_batch(c, count, in.data(), out.data(), n.data(), T, Tp.data())) return rc;
      for (size_t i = 0; i < which.size(); ++i) if (handled[i]) n[which[i]] = c->slots[dst[which[i]]].n;
    }
  }
  return MVR_OK;
}

API int mvr_cloud_transform_f32(mvr_ctx *ctx, int dst, int src, const float T[16])
{
  if (!ctx || !slot_ok(dst) || !slot_ok(src) || !T) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  const size_t n = c->slots[src].n;
  if (dst != src) { c->slots[dst].n = 0; if (int rc = cloud_reserve(c, c->slots[dst], n, false)) return rc; }
  if (int rc = launch_transform_f32(c, c->slots[src].pts, c->slots[dst].pts, n, T)) return rc;
  c->slots[dst].n = n;
  if (dst != src) { inherit_point_set(c->slots[dst], c->slots[src]); c->slots[dst].segs = c->slots[src].segs; }
  else c->slots[dst].forget_pose();
  c->slots[dst].stale_coords();
  if (c->slots[src].has_normals && n) {
    if (int rc = ensure(c, c->slots[dst].nrm, c->slots[dst].nrm_cap, n)) return rc;
    if (int rc = launch_rotate_normals_f32(c, c->slots[src].nrm, c->slots[dst].nrm, n, T)) return rc;
  }
  c->slots[dst].has_normals = c->slots[src].has_normals;
  return MVR_OK;
}

// ----------------------------------------------------------------- hot path

API int mvr_nn(mvr_ctx *ctx, int qs, int ts, int fma, uint32_t *idx, float *d2)
{
  if (!ctx || !slot_ok(qs) || !slot_ok(ts)) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  Cloud &q = c->slots[qs], &t = c->slots[ts];
  if (q.n == 0) return MVR_OK;
  if (int rc = ensure(c, c->keys, c->keys_cap, q.n)) return rc;
  if (int rc = launch_fill_u64(c, c->keys, q.n, kKeyInit)) return rc;
  if (c->nn_mode != 0 && t.n > 0) {
    if (int rc = ensure_index(c, q)) return rc;
    if (int rc = ensure_index(c, t)) return rc;
    if (int rc = launch_nn_cull(c, q, 0, q.n, nullptr, t, INFINITY, fma != 0, c->keys)) return rc;
  } else {
    if (int rc = launch_nn(c, q.pts, 0, q.n, nullptr, nullptr, t.pts, t.n, fma != 0, c->keys)) return rc;
  }
  uint32_t *didx = nullptr; float *dd2 = nullptr;
  MVR_HIP_TRY(c, hipMalloc(&didx, q.n * 4));
  if (hipMalloc(&dd2, q.n * 4) != hipSuccess) { (void)hipFree(didx); return set_error(c, MVR_E_HIP, "hipMalloc"); }
  int rc = launch_decode_keys(c, c->keys, q.n, didx, dd2);
  hipError_t e = hipSuccess;
  if (!rc && idx) e = hipMemcpyAsync(idx, didx, q.n * 4, hipMemcpyDeviceToHost, c->stream);
  if (!rc && e == hipSuccess && d2) e = hipMemcpyAsync(d2, dd2, q.n * 4, hipMemcpyDeviceToHost, c->stream);
  hipError_t e2 = hipStreamSynchronize(c->stream);
  (void)hipFree(didx); (void)hipFree(dd2);
  if (rc) return rc;
  if (e != hipSuccess) return set_error(c, MVR_E_HIP, "mvr_nn copy", e);
  if (e2 != hipSuccess) return set_error(c, MVR_E_HIP, "mvr_nn sync", e2);
  return MVR_OK;
}

static int pair_common(Ctx *c, int ss, int ts, double max_dist, int reciprocal, int fma, size_t qb, size_t qn,
                       double *evals, SearchPlan *plan)
{
  Cloud &s = c->slots[ss], &t = c->slots[ts];
  if (int rc = run_search(c, s, t, qb, qn, max_dist, reciprocal != 0, fma != 0, evals, plan)) return rc;
  return launch_pass1(c, s.pts, t.pts, c->keys, c->rkeys, plan->slot, plan->count, plan->qperm, plan->tinv, qb, qn,
                      max_dist * max_dist, reciprocal != 0 && t.n > 0, c->match, c->moments);
}

API int mvr_correspondences(mvr_ctx *ctx, int ss, int ts, double max_dist, int reciprocal, int fma,
                            int32_t *query, int32_t *match, float *dist2, size_t cap, size_t *m)
{
  if (!ctx || !slot_ok(ss) || !slot_ok(ts) || !m) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  *m = 0;
  const size_t ns = c->slots[ss].n;
  if (ns == 0) return MVR_OK;
  SearchPlan plan;
  if (int rc = pair_common(c, ss, ts, max_dist, reciprocal, fma, 0, ns, nullptr, &plan)) return rc;
  std::vector<int32_t> hm(ns);
  std::vector<nnkey_t> hk(ns);
  MVR_HIP_TRY(c, hipMemcpyAsync(hm.data(), c->match, ns * 4, hipMemcpyDeviceToHost, c->stream));
  MVR_HIP_TRY(c, hipMemcpyAsync(hk.data(), c->keys, ns * 8, hipMemcpyDeviceToHost, c->stream));
  MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
  size_t k = 0;
  for (size_t i = 0; i < ns; ++i) {
    if (hm[i] < 0) continue;
    if (k < cap) {
      if (query) query[k] = (int32_t)i;
      if (match) match[k] = hm[i];
      if (dist2) { uint32_t b = (uint32_t)(hk[i] >> 32); std::memcpy(&dist2[k], &b, 4); }
    }
    ++k;
  }
  *m = k;
  return MVR_OK;
}

API int mvr_pair_moments(mvr_ctx *ctx, int ss, int ts, double max_dist, int reciprocal, int fma,
                         mvr_pair_moments_t *out)
{
  if (!ctx || !slot_ok(ss) || !slot_ok(ts) || !out) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  std::memset(out, 0, sizeof *out);
  const Cloud &s = c->slots[ss], &t = c->slots[ts];
  if (s.n == 0) return MVR_OK;
  SearchPlan plan;
  if (int rc = pair_common(c, ss, ts, max_dist, reciprocal, fma, 0, s.n, nullptr, &plan)) return rc;
  if (int rc = launch_pass2(c, s.pts, t.pts, c->match, plan.qperm, 0, s.n, c->moments)) return rc;
  if (int rc = read_moments(c, 17)) return rc;
  const double *h = c->h_moments;
  out->n = h[0];
  for (int k = 0; k < 3; ++k) { out->mean_src[k] = h[1 + k]; out->mean_tgt[k] = h[4 + k]; }
  out->mse = h[7];
  for (int k = 0; k < 9; ++k) out->sigma[k] = h[8 + k];
  return MVR_OK;
}

static int moments2_impl(Ctx *c, int ss, int ts, double max_dist, int reciprocal, int fma, size_t qb, size_t qn,
                         const double origin[3], double *dev_out)
{
  const Cloud &s = c->slots[ss], &t = c->slots[ts];
  if (qb > s.n) qb = s.n;
  if (qn > s.n - qb) qn = s.n - qb;
  SearchPlan plan;
  Cloud &sc = c->slots[ss], &tc = c->slots[ts];
  if (int rc = run_search(c, sc, tc, qb, qn, max_dist, reciprocal != 0, fma != 0, nullptr, &plan)) return rc;
  return launch_accept_moments2(c, s.pts, t.pts, c->keys, c->rkeys, plan.slot, plan.qperm, plan.tinv, qb, qn, max_dist * max_dist,
                                reciprocal != 0 && t.n > 0, origin, dev_out);
}

API int mvr_pair_moments2_dev(mvr_ctx *ctx, int ss, int ts, double max_dist, int reciprocal, int fma, size_t qb,
                              size_t qn, const double origin[3], double *dev_out)
{
  if (!ctx || !slot_ok(ss) || !slot_ok(ts) || !origin || !dev_out) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  return moments2_impl(c, ss, ts, max_dist, reciprocal, fma, qb, qn, origin, dev_out);
}

API int mvr_pair_moments2(mvr_ctx *ctx, int ss, int ts, double max_dist, int reciprocal, int fma, size_t qb,
                          size_t qn, const double origin[3], mvr_pair_moments2_t *out)
{
  if (!ctx || !slot_ok(ss) || !slot_ok(ts) || !origin || !out) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = moments2_impl(c, ss, ts, max_dist, reciprocal, fma, qb, qn, origin, c->moments)) return rc;
  if (int rc = read_moments(c, 32)) return rc;
  static_assert(sizeof(mvr_pair_moments2_t) == 32 * sizeof(double), "moments2 layout");
  std::memcpy(out, c->h_moments, sizeof *out);
  return MVR_OK;
}

// Worker k of a context: created on first use, with a private non-blocking stream.
static int get_worker(Ctx *c, size_t k, Ctx **out)
{
  while (c->workers.size() <= k) {
    mvr_ctx *w = nullptr;
    if (int rc = ctx_create_impl(&w, c->device, nullptr, true)) return set_error(c, rc, "worker context");
    CTX(w)->parent = c;
    c->workers.push_back(CTX(w));
  }
  Ctx *w = c->workers[k];
  w->nn_mode = c->nn_mode; w->nn_q = c->nn_q; w->nn_sub = c->nn_sub; w->nn_blocks_per_cu = c->nn_blocks_per_cu;
  w->cull_q = c->cull_q; w->cull_w = c->cull_w; w->cull_slices = c->cull_slices; w->ring_search = c->ring_search; w->grid_light_rows = c->grid_light_rows; w->grid_light_rows_lone = c->grid_light_rows_lone; w->grid_probe = c->grid_probe; w->grid_probe_rows = c->grid_probe_rows; w->reduce_rows = c->reduce_rows; w->grid_wide = c->grid_wide; w->grid_lanes = c->grid_lanes; w->grid_stage = c->grid_stage; w->grid_stage_lone = c->grid_stage_lone; w->seed_delta_um = c->seed_delta_um; w->rim_cert_um = c->rim_cert_um; w->grid_cluster = c->grid_cluster; w->grid_sets = c->grid_sets; w->grid_tail = c->grid_tail; w->cull_list_w = c->cull_list_w; w->grid_wide_waves = c->grid_wide_waves; w->grid_cell_points = c->grid_cell_points; w->prof = c->prof; w->prof_mask = c->prof_mask; w->prof_totals = c->prof_totals;
  if (!w->ev_join && hipEventCreateWithFlags(&w->ev_join, hipEventDisableTiming) != hipSuccess)
    return set_error(c, MVR_E_HIP, "worker event");
  *out = w;
  return MVR_OK;
}

// All scan pairs of a global pass with ONE launch per stage (culled mode): forward searches, flagging of the
// matched targets, reverse searches, filter + raw moments, final sums -- 6 launches for V pairs instead of ~10 V
// on worker streams.  The pairs of one launch fill the chip together, so the tail of one pair's search is covered
// by the others.  Same results as the one-pair calls, bit for bit.
// c owns the clouds; w (c itself, or one of its workers) provides the stream and the work buffers.
// phases: bit 0 = the forward searches, bit 1 = everything after them (so that the forward launches of several
// groups can all be enqueued before any group's long tail of small launches).
static int pair_batch_fused(Ctx *c, Ctx *w, int n_pairs, const int *src, const int *dst, double max_dist, int reciprocal, int fma,
                            const size_t *q_begin, const size_t *q_count, const double origin[3], double *table, int phases = 3,
                            int pair_base = 0)
{
  std::vector<size_t> off_s((size_t)n_pairs + 1, 0), off_t((size_t)n_pairs + 1, 0), off_p((size_t)n_pairs + 1, 0), qb((size_t)n_pairs), qn((size_t)n_pairs);
  for (int k = 0; k < n_pairs; ++k) {
    const Cloud &s = c->slots[src[k]], &t = c->slots[dst[k]];
    if (s.n > 0xFFFFFFF0ull || t.n > 0xFFFFFFF0ull) return set_error(c, MVR_E_ARG, "cloud too large for 32-bit indices");
    size_t b = q_begin ? q_begin[k] : 0, n = q_count ? q_count[k] : s.n;
    if (b > s.n) b = s.n;
    if (n > s.n - b) n = s.n - b;
    if (t.n == 0) n = 0;
    qb[k] = b; qn[k] = n;
    off_s[k + 1] = off_s[k] + s.n; off_t[k + 1] = off_t[k] + t.n;
    off_p[k + 1] = off_p[k] + (size_t)reduce_blocks_for(c, n) * 29;
  }
  const nnkey_t *keys_before = w->bkeys;
  if (int rc = ensure(w, w->bkeys, w->bkeys_cap, off_s[n_pairs])) return rc;
  if (int rc = ensure(w, w->brkeys, w->brkeys_cap, off_t[n_pairs])) return rc;
  const uint32_t *bound_before = w->bbound;
  if (int rc = ensure(w, w->bbound, w->bbound_cap, off_t[n_pairs])) return rc;
  if (w->bbound != bound_before) w->bbound_clean = false;
  if (int rc = ensure(w, w->bpartials, w->bpartials_cap, off_p[n_pairs])) return rc;
  if (int rc = ensure(w, w->blist, w->blist_cap, off_t[n_pairs])) return rc;
  if (int rc = ensure(w, w->bslot, w->bslot_cap, off_t[n_pairs])) return rc;
  std::vector<size_t> off_c((size_t)n_pairs + 1, 0);
  for (int k = 0; k < n_pairs; ++k) off_c[k + 1] = off_c[k] + (c->slots[dst[k]].n + 255) / 256;
  if (int rc = ensure(w, w->bchunks, w->bchunks_cap, off_c[n_pairs] + (size_t)n_pairs)) return rc;
  uint32_t *counts = w->bchunks + off_c[n_pairs];
  const double max2 = max_dist * max_dist;
  const float cap2 = cap_from_max2(max2);
  if ((phases & 2) && c->last_batch.size() >= (size_t)(pair_base + n_pairs))
    for (int k = 0; k < n_pairs; ++k) {
      BatchPairRec &r = c->last_batch[(size_t)(pair_base + k)];
      r.w = w; r.off_s = off_s[k]; r.off_t = off_t[k]; r.qb = qb[k]; r.qn = qn[k]; r.src = src[k]; r.dst = dst[k]; r.max2 = max2; r.reciprocal = reciprocal;
      r.src_set = c->slots[src[k]].set_id; r.dst_set = c->slots[dst[k]].set_id;
    }
  // Do the forward keys left in bkeys[] by the previous fused pass on this context belong to the very same searches
  // (same point sets in the same order -- a posed copy keeps its scan's set id and ordering --, same query ranges,
  // same layout)?  Then every forward search starts from the distance of its previous match (seed_from_keys): the
  // passes of a registration differ by a small motion.  The bound is computed from the CURRENT coordinates, so it is
  // exact whatever happened to the poses in between; results do not depend on it.
  std::vector<unsigned long long> sig;
  sig.push_back((unsigned long long)n_pairs); sig.push_back((unsigned long long)(fma != 0));
  for (int k = 0; k < n_pairs; ++k) {
    const Cloud &s = c->slots[src[k]], &t = c->slots[dst[k]];
    const unsigned long long v[6] = {s.set_id, t.set_id, (unsigned long long)s.n, (unsigned long long)t.n, (unsigned long long)qb[k], (unsigned long long)qn[k]};
    sig.insert(sig.end(), v, v + 6);
  }
  const bool seed = c->seed_forward && (phases & 1) && keys_before == w->bkeys && w->fused_sig == sig;
  if (phases & 1) w->fused_sig = sig;
  std::vector<CullPair> fwd((size_t)n_pairs), rev((size_t)n_pairs);
  for (int k = 0; k < n_pairs; ++k) {
    const Cloud &s = c->slots[src[k]], &t = c->slots[dst[k]];
    fwd[k] = make_cull_pair(s, qb[k], qn[k], nullptr, t, w->bkeys + off_s[k]);
    fwd[k].key_by_pos = 1;       // forward keys live in sorted space from here on: slot = the query's position, low word = the match's position
    fwd[k].seed_from_keys = seed ? 1u : 0u;
    // reverse queries = the distinct matched targets, compacted in Hilbert order (list position = key slot)
    rev[k] = make_cull_pair(t, 0, std::min(qn[k], t.n), nullptr, s, w->brkeys + off_t[k]);
    rev[k].qlist = w->blist + off_t[k]; rev[k].qcount = counts + k;
    rev[k].qbound = w->bbound + off_t[k];      // a matched target has a source point at the forward distance: its search starts there
  }
  // Which kernel: bounded queries (a forward search seeded by its previous match, every reverse search) walk the uniform
  // grid, one thread per query, a few dozen evaluations each (mvr_grid.hip); queries without a bound -- the first pass --
  // take the culled kernel.  Both are exact; the choice never shows in a result.
  bool grid_ok = c->ring_search != 0 && cap2 < FLT_MAX;
  for (int k = 0; k < n_pairs && grid_ok; ++k) {
    const Cloud &s = c->slots[src[k]], &t = c->slots[dst[k]];
    if (qn[k] == 0) continue;
    grid_ok = s.grid && t.grid && s.gcoords_valid && t.gcoords_valid && s.grid->n == s.n && t.grid->n == t.n &&
              s.order && t.order && s.grid->built_for == s.order.get() && t.grid->built_for == t.order.get();
  }
  std::vector<GridPair> gfwd, grev;
  if (grid_ok) {
    if (int rc = ensure(w, w->bheavy, w->bheavy_cap, off_s[n_pairs] + off_t[n_pairs])) return rc;      // forward flags by source position, then reverse flags by list position
    if (int rc = ensure(w, w->bwide, w->bwide_cap, off_s[n_pairs] + off_t[n_pairs])) return rc;        // the lists of wide bounded queries, same layout
    if (!w->bwide_count) {                                                                              // their counters: zero now, put back to zero by every pass's moments launch
      MVR_MAY_BLOCK(w, "the wide-query counters are not allocated yet");
      MVR_HIP_TRY(w, hipMalloc(&w->bwide_count, 3 * kWideCounters * sizeof(uint32_t)));
      MVR_HIP_TRY(w, hipMemsetAsync(w->bwide_count, 0, 3 * kWideCounters * sizeof(uint32_t), w->stream));
    }
    if (int rc = ensure(w, w->bcull_sets, w->bcull_sets_cap, off_s[n_pairs] / 64 + (size_t)n_pairs)) return rc;      // the forward sets that hold a flagged query
    if (c->rim_cert_um > 0) {          // rim certificates: one float per forward query, kept from pass to pass like the keys
      const float *before = w->bcert;
      if (int rc = ensure(w, w->bcert, w->bcert_cap, off_s[n_pairs])) return rc;
      if (w->bcert != before) w->bcert_zero = false;
    }
    if (n_pairs > kWideCounters) grid_ok = false;
  }
  if (grid_ok) {
    gfwd.resize((size_t)n_pairs); grev.resize((size_t)n_pairs);
    for (int k = 0; k < n_pairs; ++k) {
      const Cloud &s = c->slots[src[k]], &t = c->slots[dst[k]];
      if (qn[k] == 0) { gfwd[k] = GridPair(); grev[k] = GridPair(); continue; }
      gfwd[k] = make_grid_pair(s, qb[k], qn[k], t, w->bkeys + off_s[k]);
      gfwd[k].key_by_pos = 1; gfwd[k].seed_from_keys = seed ? 1u : 0u; gfwd[k].mark = fwd[k].mark;
      // seed_delta: how far the pair has moved since the searches that left the keys -- from the device records of a pass that
      // is enqueued ahead of its poses, else from what note_pose() has added up since both clouds were last searched
      gfwd[k].qpose_dev = s.pose_dev;
      if (c->rim_cert_um > 0 && w->bcert) gfwd[k].cert = w->bcert + off_s[k];
      gfwd[k].delta = (s.moved >= 0.0 && t.moved >= 0.0 && s.last_pose_set == s.set_id && t.last_pose_set == t.set_id)
                          ? std::nextafterf((float)((s.moved + t.moved) * (1.0 + 1e-6)), INFINITY) : -1.f;
      gfwd[k].heavy = w->bheavy + off_s[k];      // wide balls are left to the culled kernel: same keys, same marks
      grev[k] = make_grid_pair(t, 0, std::min(qn[k], t.n), s, w->brkeys + off_t[k]);
      grev[k].qlist = rev[k].qlist; grev[k].qcount = rev[k].qcount; grev[k].qbound = rev[k].qbound;
      grev[k].heavy = w->bheavy + off_s[n_pairs] + off_t[k];
      if (c->cull_list && c->grid_lanes == 1) { gfwd[k].cull_sets = w->bcull_sets + off_s[k] / 64 + (size_t)k; gfwd[k].cull_count = w->bwide_count + 2 * kWideCounters + k; }
      if (c->grid_wide) {
        grev[k].heavy = nullptr;                 // every reverse query has a bound: the wide ones all take the wave-per-query launch
        gfwd[k].wide_list = w->bwide + off_s[k]; gfwd[k].wide_count = w->bwide_count + k;
        grev[k].wide_list = w->bwide + off_s[n_pairs] + off_t[k]; grev[k].wide_count = w->bwide_count + kWideCounters + k;
      }
    }
  }
  // Who records where the reverse searches start (a matching d2 per matched target)?  The forward searches themselves when
  // they are the grid walk (+5 us there, a launch of 12 us less), a separate launch over the keys when the culled kernel
  // answers every query (marking inside it measured 3 % slower); fused_mark = 2 / 0 force one or the other.  Decided with
  // the forward launches (phase 1), remembered for the rest of the pass.
  // (unseeded_grid: the forward searches of a pass WITHOUT seeds -- a registration's first -- walk the grid too: every query probes
  // the cells next to it, what finds nothing there goes to the listed sets; the culled kernel took 0.59 ms for such a pass)
  const bool grid_fwd = grid_ok && (seed || c->unseeded_grid);
  if (phases & 1) w->marked_in_search = reciprocal && (c->fused_mark == 2 || (c->fused_mark == 1 && grid_fwd));
  if (w->marked_in_search)
    for (int k = 0; k < n_pairs; ++k) if (qn[k]) { fwd[k].mark = w->bbound + off_t[k]; if (grid_ok) gfwd[k].mark = fwd[k].mark; }
  if (phases & 1) {
    // (the certificates belong to the searches the keys belong to: another set of searches, or none before, starts without any)
    if (w->bcert && (!seed || !w->bcert_zero || w->bcert_cap2 != cap2)) {      // (... or searches with another cap)
      MVR_HIP_TRY(w, hipMemsetAsync(w->bcert, 0, w->bcert_cap * sizeof(float), w->stream));
      w->bcert_zero = true; w->bcert_cap2 = cap2;
    }
    if (reciprocal && !w->bbound_clean) MVR_HIP_TRY(w, hipMemsetAsync(w->bbound, 0xFF, w->bbound_cap * sizeof(uint32_t), w->stream));
    if (reciprocal) w->bbound_clean = false;        // dirty until this pass's moments launch has put it back
    if (grid_fwd) {
      if (int rc = launch_nn_grid_batch(w, gfwd.data(), n_pairs, cap2, fma != 0)) return rc;
      if (c->grid_debug) {          // diagnostics (tune key grid_debug): how many queries left the thread-per-query walk, per pass
        MVR_MAY_BLOCK(c, "grid_debug reads counters back");
        std::vector<uint32_t> cnt((size_t)kWideCounters); std::vector<uint8_t> hv(off_s[n_pairs]);
        MVR_HIP_TRY(w, hipStreamSynchronize(w->stream));
        MVR_HIP_TRY(w, hipMemcpy(cnt.data(), w->bwide_count, kWideCounters * sizeof(uint32_t), hipMemcpyDeviceToHost));
        MVR_HIP_TRY(w, hipMemcpy(hv.data(), w->bheavy, hv.size(), hipMemcpyDeviceToHost));
        size_t wide = 0, cull = 0, sets = 0;
        for (int k = 0; k < n_pairs; ++k) {
          wide += cnt[k];
          for (size_t i = 0; i < qn[k]; i += 64) { size_t f = 0; for (size_t j = i; j < std::min(qn[k], i + 64); ++j) f += hv[off_s[k] + j]; cull += f; sets += f != 0; }
        }
        std::fprintf(stderr, "[grid] forward: %zu queries, %zu to a wave each, %zu to the culled kernel in %zu sets\n", off_s[n_pairs], wide, cull, sets);
      }
      // the listed sets: over the grid (a block per set stages the union of the flagged balls) while that union is a few
      // hundred rows of cells -- with dense scans (small cells) it is thousands of points per set and the culled kernel's
      // box hierarchy is the better tool again (36 x 1M: 5.9 ms per pass against 6.4).  grid_sets = 2 forces the grid.
      bool sets_on_grid = c->cull_list && c->grid_lanes == 1 && c->grid_sets;
      if (sets_on_grid && c->grid_sets == 1)
        for (int k = 0; k < n_pairs && sets_on_grid; ++k) {
          if (!qn[k]) continue;
          const double span = (2.0 * max_dist + 4.0) / (double)c->slots[dst[k]].grid->h;      // cells across a rim set's union, roughly
          sets_on_grid = span * span <= 300.0;
        }
      if (c->grid_wide && sets_on_grid && c->grid_tail) {           // the wide queries and the listed sets in ONE launch
        if (int rc = launch_nn_grid_tail_batch(w, gfwd.data(), n_pairs, cap2, fma != 0)) return rc;
      } else {
      if (c->grid_wide) { if (int rc = launch_nn_grid_wide_batch(w, gfwd.data(), n_pairs, cap2, fma != 0)) return rc; }
      // ... and the queries it flagged, in place (blocks without a flagged query leave at once)
      for (int k = 0; k < n_pairs; ++k) {
        fwd[k].qflags = qn[k] ? w->bheavy + off_s[k] : nullptr;
        fwd[k].setlist = gfwd[k].cull_sets; fwd[k].setcount = gfwd[k].cull_count;
      }
      if (sets_on_grid) { if (int rc = launch_nn_grid_sets_batch(w, gfwd.data(), n_pairs, cap2, fma != 0)) return rc; }
      else if (c->cull_list && c->grid_lanes == 1) { if (int rc = launch_nn_cull_list_batch(w, fwd.data(), n_pairs, cap2, fma != 0)) return rc; }
      else if (int rc = launch_nn_cull_batch(w, fwd.data(), n_pairs, cap2, fma != 0)) return rc;
      }
    } else if (int rc = launch_nn_cull_batch(w, fwd.data(), n_pairs, cap2, fma != 0)) return rc;
  }
  if (!(phases & 2)) return MVR_OK;
  const bool recip = reciprocal != 0;
  for (int base = 0; base < n_pairs; base += kBatchPairs) {
    const int m = std::min(kBatchPairs, n_pairs - base);
    GlueBatch gb;
    gb.max2 = max2; gb.reciprocal = recip ? 1 : 0;
    for (int k = 0; k < 3; ++k) gb.origin[k] = origin[k];
    for (int j = 0; j < m; ++j) {
      const int k = base + j;
      const Cloud &s = c->slots[src[k]], &t = c->slots[dst[k]];
      GluePair &g = gb.p[j];
      g.src = s.pts; g.tgt = t.pts; g.qs = s.sorted; g.ts = t.sorted; g.by_pos = 1;
      g.keys = w->bkeys + off_s[k]; g.rkeys = w->brkeys + off_t[k];
      g.qperm = (s.order && qn[k]) ? s.order->perm : nullptr; g.tinv = (t.order && qn[k]) ? t.order->inv : nullptr;
      g.bound = w->bbound + off_t[k];
      g.list = w->blist + off_t[k]; g.slot = w->bslot + off_t[k]; g.chunks = w->bchunks + off_c[k]; g.qcount = counts + k;
      g.nt = qn[k] ? t.n : 0;
      g.partials = w->bpartials + off_p[k]; g.out = table + (size_t)k * 32;
      g.q_begin = qb[k]; g.q_count = qn[k]; g.blocks = reduce_blocks_for(c, qn[k]);
      if (grid_ok) { g.zero_a = w->bwide_count + k; g.zero_b = w->bwide_count + kWideCounters + k; g.zero_c = w->bwide_count + 2 * kWideCounters + k; }
    }
    if (recip) {
      if (!w->marked_in_search) { if (int rc = launch_flag_matched_batch(w, gb, m)) return rc; }      // (else: marked by the forward launches themselves)
      if (int rc = launch_compact_flags_batch(w, gb, m)) return rc;
      if (grid_ok) {
        if (int rc = launch_nn_grid_batch(w, grev.data() + base, m, cap2, fma != 0)) return rc;
        if (c->grid_debug && c->grid_wide) {
          MVR_MAY_BLOCK(c, "grid_debug reads counters back");
          std::vector<uint32_t> cnt((size_t)2 * kWideCounters), qc((size_t)n_pairs);
          MVR_HIP_TRY(w, hipStreamSynchronize(w->stream));
          MVR_HIP_TRY(w, hipMemcpy(cnt.data(), w->bwide_count, cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
          MVR_HIP_TRY(w, hipMemcpy(qc.data(), counts, qc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
          size_t wide = 0, all = 0;
          for (int j = 0; j < m; ++j) { wide += cnt[kWideCounters + base + j]; all += qc[base + j]; }
          std::fprintf(stderr, "[grid] reverse: %zu queries, %zu to a wave each\n", all, wide);
        }
        if (c->grid_wide) { if (int rc = launch_nn_grid_wide_batch(w, grev.data() + base, m, cap2, fma != 0)) return rc; }
        else
        for (int j = 0; j < m; ++j) rev[base + j].qflags = qn[base + j] ? w->bheavy + off_s[n_pairs] + off_t[base + j] : nullptr;     // the wide ones, by list position
        if (!c->grid_wide) { if (int rc = launch_nn_cull_batch(w, rev.data() + base, m, cap2, fma != 0)) return rc; }
      } else if (int rc = launch_nn_cull_batch(w, rev.data() + base, m, cap2, fma != 0)) return rc;
    }
    bool will_launch = false;
    for (int j = 0; j < m; ++j) will_launch = will_launch || gb.p[j].blocks > 0;
    if (c->signal_armed && will_launch && w == c && base + kBatchPairs >= n_pairs && c->done_counter && c->d_done) {      // the chain's last launch carries its completion word
      gb.done_word = c->d_done; gb.done_counter = c->done_counter; gb.done_seq = c->signal_seq;
      c->signal_armed = false; c->signal_sent = true;
    }
    if (int rc = launch_accept_moments2_batch(w, gb, m)) return rc;
  }
  if (recip) w->bbound_clean = true;
  return MVR_OK;
}

API int mvr_pair_moments2_batch(mvr_ctx *ctx, int n_pairs, const int *src, const int *dst, double max_dist,
                                int reciprocal, int fma, const size_t *q_begin, const size_t *q_count,
                                const double origin[3], mvr_pair_moments2_t *out, double *dev_out)
{
  if (!ctx || n_pairs < 0 || (n_pairs && (!src || !dst)) || !origin || (!out && !dev_out)) return MVR_E_ARG;
  for (int k = 0; k < n_pairs; ++k) if (!slot_ok(src[k]) || !slot_ok(dst[k])) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  if (n_pairs == 0) return MVR_OK;
  const int n_workers = std::max(1, std::min(c->pair_streams, n_pairs));
  double *table = dev_out;
  if (!table) {                      // results wanted on the host: stage them in a device table first
    if (int rc = ensure(c, c->batch_table, c->batch_cap, (size_t)n_pairs * 32)) return rc;
    table = c->batch_table;
  }
  // everything the workers only READ is prepared here, on the caller's stream
  if (c->nn_mode != 0) {
    std::vector<Cloud *> used;
    for (int k = 0; k < n_pairs; ++k) { used.push_back(&c->slots[src[k]]); used.push_back(&c->slots[dst[k]]); }
    if (int rc = ensure_index_batch(c, used.data(), (int)used.size())) return rc;      // one refresh launch for all views
    host_mark("  batch: indices");
    if (c->pair_fused && c->ring_search && c->fused_passes > 0 && max_dist * max_dist < (double)FLT_MAX) {
      // the grid search's side: every posed copy finds its set's grid (built once, from the cloud that holds the set's
      // canonical coordinates) and refreshes its coordinates in grid order -- one launch for all views.  Not in the
      // first fused pass of a context: it has no seeds and would not use a grid, and a caller that runs a single pass
      // should not pay for twelve of them (0.85 ms each at 200k points; tools/first_passes.py)
      std::sort(used.begin(), used.end());
      used.erase(std::unique(used.begin(), used.end()), used.end());
      std::vector<Cloud *> need, canon;
      for (Cloud *cl : used) {
        if (cl->grid || !cl->pose_known || cl->n == 0) continue;
        for (Cloud &o : c->slots)
          if (o.set_id == cl->set_id && o.canonical && o.n == cl->n) { need.push_back(cl); canon.push_back(&o); break; }
      }
      if (!canon.empty()) {
        (void)ensure_grids(c, canon.data(), (int)canon.size(), max_dist + 0.5, c->stream, nullptr);      // (a set left without a grid keeps the culled kernel)
        for (size_t k = 0; k < need.size(); ++k) if (canon[k]->grid && canon[k]->grid->n == need[k]->n) need[k]->grid = canon[k]->grid;
      }
      if (int rc = refresh_grid_coords_batch(c, used.data(), (int)used.size())) return rc;
    }
  }
  if (c->nn_mode != 0 && c->pair_fused) {
    host_mark("  batch: grid coordinates");
    ++c->fused_passes;
    // Optionally in G groups of pairs, group 0 on the caller's stream and the others on worker streams: a group's
    // glue kernels and the tail of its searches then overlap the other groups' searches.
    // (only when there are pairs to spare: with fewer than four per group the second stream just doubles the launches --
    // measured on 2 / 3 / 4 / 6 pairs of 200k: equal or up to 4 % slower)
    // (a pass enqueued ahead of its poses runs its pairs as ONE group: the groups measure the same since the grid search --
    // 0.508 / 0.506 / 0.504 ms with 1 / 2 / 3 -- and one stream needs no fork / join events between queued passes)
    const int G = (n_pairs >= 4 * c->pair_groups && !c->pose_from_table && !c->single_group) ? c->pair_groups : 1;
    c->last_batch.assign((size_t)n_pairs, BatchPairRec());
    if (G != 1 || out) c->signal_armed = false;      // (the sums launch is then not the last thing on the stream)
    if (G == 1) {
      if (int rc = pair_batch_fused(c, c, n_pairs, src, dst, max_dist, reciprocal, fma, q_begin, q_count, origin, table)) return rc;
    } else {
      // waves per query set: decided for the whole pass (the groups share the chip), not per group
      size_t sets = 0;
      for (int k = 0; k < n_pairs; ++k) sets += ((q_count ? std::min(q_count[k], c->slots[src[k]].n) : c->slots[src[k]].n) + 63) / 64;
      const int saved_w = c->cull_w;
      if (c->cull_w == 0 && sets >= (size_t)c->n_cu * 40) c->cull_w = 1;
      int status = MVR_OK;
      if (!c->ev_fork && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) status = set_error(c, MVR_E_HIP, "fork event");
      if (status == MVR_OK) status = flush_super_boxes(c);      // (what every group may read, before the fork: a group on the culled kernel then finds its super boxes)
      if (status == MVR_OK && hipEventRecord(c->ev_fork, c->stream) != hipSuccess) status = set_error(c, MVR_E_HIP, "fork");
      int forked = 0;
      for (int phase = 1; phase <= 2; ++phase)                      // all forward searches first, then each group's remaining stages
        for (int g = G - 1; g >= 0 && status == MVR_OK; --g) {      // workers first, the caller's own group last
          const int lo = (int)((long long)n_pairs * g / G), hi = (int)((long long)n_pairs * (g + 1) / G);
          Ctx *w = c;
          if (g > 0) {
            if ((status = get_worker(c, (size_t)(g - 1), &w)) != MVR_OK) break;
            if (phase == 1) {
              if (hipStreamWaitEvent(w->stream, c->ev_fork, 0) != hipSuccess) { status = set_error(c, MVR_E_HIP, "fork"); break; }
              forked = std::max(forked, g);
            }
          }
          status = pair_batch_fused(c, w, hi - lo, src + lo, dst + lo, max_dist, reciprocal, fma, q_begin ? q_begin + lo : nullptr,
                                    q_count ? q_count + lo : nullptr, origin, table + (size_t)lo * 32, phase, lo);
          if (status != MVR_OK && w != c) c->last_error = w->last_error;
        }
      for (int g = 1; g <= forked; ++g) {                            // join, always
        Ctx *w = c->workers[(size_t)(g - 1)];
        if (w->ev_join && hipEventRecord(w->ev_join, w->stream) == hipSuccess) (void)hipStreamWaitEvent(c->stream, w->ev_join, 0);
        else (void)hipStreamSynchronize(w->stream);
      }
      c->cull_w = saved_w;
      if (status != MVR_OK) return status;
    }
    host_mark("  batch: searches and sums enqueued");
    // (seed_delta: the keys these searches leave belong to the clouds' poses of NOW -- what they move from here on counts from zero)
    for (int k = 0; k < n_pairs; ++k)
      for (int sl : {src[k], dst[k]}) { Cloud &cl = c->slots[sl]; if (cl.pose_known && cl.last_pose_set == cl.set_id && !cl.posed_by_table) cl.moved = 0.0; }
    if (out) {
      std::vector<double> h((size_t)n_pairs * 32);
      MVR_HIP_TRY(c, hipMemcpyAsync(h.data(), table, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
      for (int k = 0; k < n_pairs; ++k) std::memcpy(&out[k], &h[(size_t)k * 32], sizeof(mvr_pair_moments2_t));
    }
    return MVR_OK;
  }
  c->last_batch.clear();
  if (int rc = flush_super_boxes(c)) return rc;      // (before the fork: the workers search these views with the culled kernel)
  if (!c->ev_fork) MVR_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  MVR_HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
  int status = MVR_OK;
  for (int k = 0; k < n_pairs && status == MVR_OK; ++k) {
    Ctx *w = nullptr;
    if ((status = get_worker(c, (size_t)(k % n_workers), &w)) != MVR_OK) break;
    if (k < n_workers && hipStreamWaitEvent(w->stream, c->ev_fork, 0) != hipSuccess) { status = set_error(c, MVR_E_HIP, "fork"); break; }
    w->slots[0] = c->slots[src[k]];        // borrowed views (pointers + shared ordering); the parent keeps ownership
    w->slots[1] = c->slots[dst[k]];
    const size_t ns = w->slots[0].n;
    const size_t qb = q_begin ? q_begin[k] : 0, qn = q_count ? q_count[k] : ns;
    status = moments2_impl(w, 0, 1, max_dist, reciprocal, fma, qb, qn, origin, table + (size_t)k * 32);
    if (status != MVR_OK) c->last_error = w->last_error;
  }
  // join: the caller's stream continues after every worker; always executed so that no borrowed view outlives the call
  for (int k = 0; k < n_workers && (size_t)k < c->workers.size(); ++k) {
    Ctx *w = c->workers[k];
    if (w->ev_join && hipEventRecord(w->ev_join, w->stream) == hipSuccess) (void)hipStreamWaitEvent(c->stream, w->ev_join, 0);
    else (void)hipStreamSynchronize(w->stream);
    w->slots[0] = Cloud(); w->slots[1] = Cloud();
  }
  if (status != MVR_OK) return status;
  if (out) {
    static_assert(sizeof(mvr_pair_moments2_t) == 32 * sizeof(double), "moments2 layout");
    std::vector<double> h((size_t)n_pairs * 32);
    MVR_HIP_TRY(c, hipMemcpyAsync(h.data(), table, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < n_pairs; ++k) std::memcpy(&out[k], &h[(size_t)k * 32], sizeof(mvr_pair_moments2_t));
  }
  return MVR_OK;
}

// The accepted correspondences of pair k of the last fused batch, read back from where its searches left them: the forward
// keys live in sorted space (slot = the query's Hilbert position, low word = the match's Hilbert position), the reverse
// keys by list position with the ORIGINAL index of the source point they found.  Acceptance as the filter + sums launch
// decides it (accept_moments2_body): a match within max2 whose target's reverse search, also within max2, names the query.
API int mvr_pair_batch_correspondences(mvr_ctx *ctx, int k, int32_t *query, int32_t *match, float *dist2, size_t cap, size_t *m)
{
  if (!ctx || !m) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  *m = 0;
  if (k < 0 || (size_t)k >= c->last_batch.size() || !c->last_batch[(size_t)k].w)
    return set_error(c, MVR_E_ARG, "no fused batch on this context holds that pair (culled mode with pair_fused only)");
  const BatchPairRec r = c->last_batch[(size_t)k];
  const Cloud &s = c->slots[r.src], &t = c->slots[r.dst];
  if (s.set_id != r.src_set || t.set_id != r.dst_set || !s.order || !t.order)
    return set_error(c, MVR_E_ARG, "the clouds of that pair have changed since the batch");
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  if (r.qn == 0) return MVR_OK;
  Ctx *w = r.w;
  const size_t ns = s.n, nt = t.n;
  std::vector<nnkey_t> keys(r.qn), rkeys(nt);
  std::vector<uint32_t> slot(nt), sperm(ns), tperm(nt);
  MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (w != c) MVR_HIP_TRY(c, hipStreamSynchronize(w->stream));
  MVR_HIP_TRY(c, hipMemcpy(keys.data(), w->bkeys + r.off_s + r.qb, r.qn * sizeof(nnkey_t), hipMemcpyDeviceToHost));
  MVR_HIP_TRY(c, hipMemcpy(sperm.data(), s.order->perm, ns * 4, hipMemcpyDeviceToHost));
  MVR_HIP_TRY(c, hipMemcpy(tperm.data(), t.order->perm, nt * 4, hipMemcpyDeviceToHost));
  if (r.reciprocal) {
    MVR_HIP_TRY(c, hipMemcpy(rkeys.data(), w->brkeys + r.off_t, nt * sizeof(nnkey_t), hipMemcpyDeviceToHost));
    MVR_HIP_TRY(c, hipMemcpy(slot.data(), w->bslot + r.off_t, nt * 4, hipMemcpyDeviceToHost));
  }
  struct Rec { int32_t q, mt; uint32_t d; };
  std::vector<Rec> acc;
  for (size_t p = 0; p < r.qn; ++p) {
    const nnkey_t key = keys[p];
    const uint32_t tpos = (uint32_t)key, db = (uint32_t)(key >> 32);
    float d; std::memcpy(&d, &db, 4);
    if (tpos == kNone || tpos >= nt || (double)d > r.max2) continue;
    const uint32_t qi = sperm[r.qb + p];
    if (r.reciprocal) {
      const uint32_t li = slot[tpos];
      if (li >= nt) continue;
      const nnkey_t rk = rkeys[li];
      const uint32_t rb = (uint32_t)(rk >> 32);
      float dr; std::memcpy(&dr, &rb, 4);
      if ((uint32_t)rk != qi || (double)dr > r.max2) continue;
    }
    acc.push_back(Rec{(int32_t)qi, (int32_t)tperm[tpos], db});
  }
  std::sort(acc.begin(), acc.end(), [](const Rec &a, const Rec &b) { return a.q < b.q; });
  for (size_t i = 0; i < acc.size() && i < cap; ++i) {
    if (query) query[i] = acc[i].q;
    if (match) match[i] = acc[i].mt;
    if (dist2) std::memcpy(&dist2[i], &acc[i].d, 4);
  }
  *m = acc.size();
  return MVR_OK;
}

}  // extern "C" (reopened below)

// ---- the pass loop of the global registration (registrator.cpp:625-664), pipelined ------------------------------------
// One pass = a chain of ~15 launches (pose the scans, refresh their indices, forward searches, glue, reverse searches,
// filter + sums) whose only input from the previous pass is V poses, produced by a 75 us host solve.  Enqueued the
// ordinary way -- after that solve, launch by launch -- the GPU idles through the solve, through the wake-up after
// hipStreamSynchronize and, worst, whenever a kernel ends before the host has pushed the next one (a launch costs the
// host ~6 us, the small kernels run 4-16 us): 136 of 475 us per pass (profiles/r02_h_step_timeline_groups1.txt).
// Here pass k+1's whole chain is enqueued WHILE pass k runs, behind a gate (hipStreamWaitValue32) that the host opens
// with one store after its solve; the kernels read the poses from a device table (PoseRec, filled by a one-wave kernel
// at the head of the chain from pinned host memory) instead of kernel arguments, and the host learns that a pass has
// drained from a word the stream writes into pinned memory (hipStreamWriteValue32), spinning instead of sleeping in
// hipStreamSynchronize.  tools/exp_gate.hip measures the hand-off alone: 24 us with synchronize, ~1 us with the gate.
// Results are the same bits: the chain is the same chain, only the route of the poses differs.
namespace mvr {
namespace {

__global__ void pose_prep_kernel(const double *__restrict__ in, PoseRec *__restrict__ out, int n_views)
{
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_views) return;
  double T[16];
  for (int j = 0; j < 16; ++j) T[j] = in[16 * v + j];
  const float delta = (float)T[3];      // (see refresh_sorted_kernel: the table's element [3] carries the cloud's motion since the pass before)
  T[3] = 0.0;
  make_pose_rec(T, delta, out + v);
}

}  // namespace

int ensure_pose_table(Ctx *c)
{
  if (!c->pose_tab_pending) return MVR_OK;
  c->pose_tab_pending = false;
  hipLaunchKernelGGL(pose_prep_kernel, dim3((unsigned)((c->pose_tab_n + 63) / 64)), dim3(64), 0, c->stream, c->pose_in_cur, c->pose_tab_cur, c->pose_tab_n);
  return hipGetLastError() == hipSuccess ? MVR_OK : set_error(c, MVR_E_HIP, "pose_prep_kernel");
}

namespace {

// note_pose's test: an affine matrix whose 3 x 3 is a rotation up to e = |A^T A - I|_F <= 1e-3
bool pose_nearly_rigid(const double *T)
{
  if (T[3] != 0.0 || T[7] != 0.0 || T[11] != 0.0 || T[15] != 1.0) return false;
  double e2 = 0.0;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      const double d = T[4 * a] * T[4 * b] + T[4 * a + 1] * T[4 * b + 1] + T[4 * a + 2] * T[4 * b + 2];
      const double x = d - (a == b ? 1.0 : 0.0);
      e2 += x * x;
    }
  return std::sqrt(e2) <= 1e-3;
}

int pipe_setup(Ctx *c, int n_views)
{
  if (!c->gate) {
    int can = 0;
    if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->device) != hipSuccess || !can) return MVR_E_HIP;
    // the gate: signal memory where the runtime offers it (the wait is then a barrier packet of the queue, no wave spins),
    // else a word of pinned host memory
    if (hipExtMallocWithFlags(reinterpret_cast<void **>(&c->gate), 8, hipMallocSignalMemory) == hipSuccess && c->gate) c->gate_is_signal = true;
    else {
      (void)hipGetLastError();
      c->gate = nullptr; c->gate_is_signal = false;
      if (hipHostMalloc(reinterpret_cast<void **>(&c->gate), 64, hipHostMallocMapped) != hipSuccess) { c->gate = nullptr; return MVR_E_HIP; }
    }
    *c->gate = 0u;
  }
  if (!c->h_done) {
    if (hipHostMalloc(reinterpret_cast<void **>(&c->h_done), 64, hipHostMallocMapped) != hipSuccess) { c->h_done = nullptr; return MVR_E_HIP; }
    *c->h_done = 0u;
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&c->d_done), c->h_done, 0) != hipSuccess) return MVR_E_HIP;
  }
  if (!c->done_counter) {
    if (hipMalloc(&c->done_counter, 64) != hipSuccess) { c->done_counter = nullptr; return MVR_E_HIP; }
    if (hipMemset(c->done_counter, 0, 64) != hipSuccess) return MVR_E_HIP;
  }
  if (c->pose_tab_cap < (size_t)n_views) {
    if (c->h_pose_in) (void)hipHostFree(c->h_pose_in);
    if (c->pose_tab) (void)hipFree(c->pose_tab);
    c->h_pose_in = nullptr; c->pose_tab = nullptr; c->pose_tab_cap = 0;
    const size_t cap = (size_t)std::max(n_views, 16);
    if (hipHostMalloc(reinterpret_cast<void **>(&c->h_pose_in), 2 * cap * 16 * sizeof(double), hipHostMallocMapped) != hipSuccess) { c->h_pose_in = nullptr; return MVR_E_HIP; }
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&c->d_pose_in), c->h_pose_in, 0) != hipSuccess) return MVR_E_HIP;
    if (hipMalloc(&c->pose_tab, 2 * cap * sizeof(PoseRec)) != hipSuccess) { c->pose_tab = nullptr; return MVR_E_HIP; }
    c->pose_tab_cap = cap;
  }
  if (!c->pipe_ops_warm) {
    // the runtime sets the two stream operations up lazily, per stream, at their first use (milliseconds): have that happen
    // now, behind whatever the stream is busy with, not in the first pass that is queued ahead
    (void)hipStreamWriteValue32(c->stream, c->d_done, *c->h_done, 0);
    (void)hipStreamWaitValue32(c->stream, c->gate, 0u, hipStreamWaitValueGte, 0xFFFFFFFFu);        // (any value is >= 0: never blocks)
    c->pipe_ops_warm = true;
  }
  return MVR_OK;
}

struct Spin { double t0 = now_ms(); unsigned n = 0; };
// the host's wait for "pass seq has drained": a spin on pinned memory (a wake-up from hipStreamSynchronize costs tens of
// microseconds), with a look at the stream every few thousand rounds so that a failed launch ends the wait, and a bound
int wait_done(Ctx *c, uint32_t seq, bool *timed_out = nullptr, bool *async_error = nullptr)
{
  Spin sp;
  if (timed_out) *timed_out = false;
  if (async_error) *async_error = false;
  for (;;) {
    if ((int32_t)(__atomic_load_n(c->h_done, __ATOMIC_ACQUIRE) - seq) >= 0) return MVR_OK;
    __builtin_ia32_pause();
    if ((++sp.n & 0xFFFu) == 0u) {
      const hipError_t e = hipStreamQuery(c->stream);
      if (e != hipSuccess && e != hipErrorNotReady) return set_error(c, MVR_E_HIP, "a pass of the pipelined ring run failed on the device", e);
      if (e == hipSuccess && (int32_t)(__atomic_load_n(c->h_done, __ATOMIC_ACQUIRE) - seq) >= 0) return MVR_OK;
      if (int rc = comm_poll(c, async_error == nullptr)) { if (async_error) *async_error = true; return rc; }      // (a peer's failure surfaces here; a caller that asks for it aborts the communicator ITSELF, after it has opened the gate of a chain queued behind this pass)
      if (now_ms() - sp.t0 > (double)c->wait_timeout_ms) {
        // (with a communicator the caller aborts it -- after it has opened the gate of the chain queued behind this pass:
        // ncclCommAbort waits for the stream, and the stream would wait for the gate)
        if (timed_out) *timed_out = true;
        return set_error(c, c->comm ? MVR_E_RCCL : MVR_E_HIP, "a pass of the pipelined ring run did not finish in time");
      }
    }
  }
}

}  // namespace

int ring_passes(Ctx *c, int n_steps, const PassLoop &L, double timing_ms[3])
{
  double sum[3] = {0.0, 0.0, 0.0};
  const int V = L.n_views;
  c->pass_ms.clear();
  double t_prev = now_ms();
  auto pass_done = [&]() { const double t = now_ms(); c->pass_ms.push_back(t - t_prev); t_prev = t; };
  auto pipe_possible = [&]() {
    if (!c->pipeline || c->nn_mode == 0 || !c->pair_fused || c->grid_debug || V < 2) return false;
    // A pass that contains a collective over MORE THAN ONE rank is never queued behind the gate: RCCL's enqueue may synchronise
    // with the stream, and a stream parked behind a gate only this host thread can open would then hold the rank inside
    // ncclAllReduce, outside every bounded wait.  A communicator of one rank has no peer to wait for and stays
    // pipelined; `pipeline_multi_rank` = 1 lifts the rule (to be set only once a recorded two-GPU run has shown it safe).
    if (c->comm && c->comm_world > 1 && !c->pipeline_multi_rank) return false;
    for (int v = 0; v < V; ++v) {
      if (L.posed_slots[v] == L.raw_slots[v] || c->slots[L.raw_slots[v]].has_normals) return false;
      for (int u = 0; u < v; ++u) if (L.posed_slots[u] == L.posed_slots[v]) return false;
    }
    return true;
  };
  // The scans' cell grids are needed from the second pass on.  When more passes follow in this call they are built on the
  // context's side stream WHILE this pass's searches run (12 x 200k: 8.8 ms of the second pass in round 2), behind an
  // event taken before the pass was enqueued; failure is not fatal -- the next pass builds what is missing.
  auto prebuild_grids = [&]() {
    if (!c->ring_search || c->nn_mode == 0 || !c->pair_fused || !(L.reach > 0.0) || !(L.reach * L.reach < (double)FLT_MAX)) return;
    std::vector<Cloud *> canon;
    for (int v = 0; v < V; ++v) { Cloud &r = c->slots[L.raw_slots[v]]; if (r.canonical && r.n && !(r.grid && r.grid->n == r.n)) canon.push_back(&r); }
    if (canon.empty() || !c->side_after) return;
    if (!c->side_stream && hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess) { c->side_stream = nullptr; return; }
    (void)ensure_grids(c, canon.data(), (int)canon.size(), L.reach + 0.5, c->side_stream, c->side_after);
  };
  auto plain_pass = [&](bool more_follow) -> int {
    const double t0 = now_ms();
    host_mark("pass: begin");
    if (more_follow) {
      if (!c->side_after && hipEventCreateWithFlags(&c->side_after, hipEventDisableTiming) != hipSuccess) c->side_after = nullptr;
      if (c->side_after) (void)hipEventRecord(c->side_after, c->stream);
    }
    // set-up that overlaps the pass: the grids on the side stream, the pipe's few allocations.  Enqueued BEFORE the pass's own
    // chain by default (setup_first): the host is the bottleneck of a registration's first pass whichever comes first, the
    // bounding boxes come back at once from an idle GPU (behind the pass they waited 0.4 ms), and the grids are ready when the
    // second pass wants them.  In any case NOT behind a pass that contains a collective: streams share hardware queues, and a collective that
    // a lost peer keeps from finishing would hold the side stream's packets behind it -- the host would wait for the side
    // stream instead of reaching the bounded wait that ends such a pass (found with the injected stall of the tests).
    const bool setup_first = c->comm != nullptr || c->inject_stall_at >= 0 || c->setup_first;
    if (more_follow && setup_first) { prebuild_grids(); if (pipe_possible()) (void)pipe_setup(c, V); }
    if (int rc = L.enqueue(L.self)) return rc;
    host_mark("pass: chain enqueued");
    if (more_follow && !setup_first) { prebuild_grids(); host_mark("pass: grids enqueued on the side stream"); if (pipe_possible()) (void)pipe_setup(c, V); host_mark("pass: pipe set up"); }
    const double t1 = now_ms();
    if (int rc = stream_wait(c)) return rc;
    host_mark("pass: drained");
    const double t2 = now_ms();
    const int rc = L.solve(L.self);
    sum[0] += t1 - t0; sum[1] += t2 - t1; sum[2] += now_ms() - t2;
    return rc;
  };
  auto all_rigid = [&]() { for (int v = 0; v < V; ++v) if (!pose_nearly_rigid(L.poses + 16 * (size_t)v)) return false; return true; };
  bool steady = L.sig != 0 && c->pipe_steady_sig == L.sig && c->pipe_steady_events == c->blocking_events;
  // one group of pairs from the first pass on when passes may be queued ahead later: the queued chain runs its pairs as one
  // group, and a change of the grouping in mid-run would re-size the work buffers and orphan the seeds in them (measured: the
  // first queued pass of a registration then cost 2.5 ms instead of 0.43)
  struct GroupGuard { Ctx *c; bool saved; ~GroupGuard() { c->single_group = saved; } } group_guard{c, c->single_group};
  if (pipe_possible() && n_steps >= 2) c->single_group = true;
  int k = 0;
  while (k < n_steps) {
    if (!(steady && n_steps - k >= 2 && pipe_possible() && all_rigid() && pipe_setup(c, V) == MVR_OK)) {
      const unsigned long long ev0 = c->blocking_events;
      if (int rc = plain_pass(n_steps - k >= 2)) return rc;
      steady = c->blocking_events == ev0 && c->fused_passes >= 2;      // (the grids exist from the second fused pass on)
      ++k;
      pass_done();
      continue;
    }
    // ---- a pipelined stretch: passes k, k + 1, ... while nothing has to wait or grow
    const size_t cap = c->pose_tab_cap;
    int parity = 0;
    std::vector<double> last_in((size_t)V * 16);
    // (seed_delta: how far every view moves from the poses written last to these, an upper bound over the scan's bounding box,
    // travels in the pinned table's always-zero element [3] of the view's pose; the posing launch moves it into the device record.
    // The first poses of a stretch have no predecessor in the table: 1e30 = unknown, the searches gather their seeds as before.)
    bool have_last = false;
    auto write_poses = [&](int par, const double *P) {
      double *dstp = c->h_pose_in + (size_t)par * cap * 16;
      std::memcpy(dstp, P, (size_t)V * 16 * sizeof(double));
      for (int v = 0; v < V; ++v) {
        const Cloud &raw = c->slots[L.raw_slots[v]];
        double d = -1.0;
        if (have_last && raw.bbox_set == raw.set_id && raw.bbox_set != 0) d = pose_motion_bound(last_in.data() + 16 * (size_t)v, P + 16 * (size_t)v, raw.bbox);
        dstp[16 * (size_t)v + 3] = d >= 0.0 ? (double)std::nextafterf((float)(d * (1.0 + 1e-6)), INFINITY) : 1.0e30;
      }
      std::memcpy(last_in.data(), P, (size_t)V * 16 * sizeof(double));
      have_last = true;
    };
    auto open_gate = [&](uint32_t seq) { __atomic_store_n(c->gate, seq, __ATOMIC_RELEASE); };
    auto enqueue_chain = [&](int par, bool gated, uint32_t *seq_out) -> int {
      for (int v = 0; v < V; ++v) c->slots[L.posed_slots[v]].pose_dev = c->pose_tab + (size_t)par * cap + (size_t)v;
      const uint32_t seq = ++c->pipe_seq;
      *seq_out = seq;
      if (gated) MVR_HIP_TRY(c, hipStreamWaitValue32(c->stream, c->gate, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
      // the device records of these poses are filled by the chain's first launch that reads them (the posing launch does it
      // on the way, any other reader through ensure_pose_table): no launch of its own at the head of the chain
      c->signal_seq = seq; c->signal_sent = false; c->signal_armed = L.ends_with_sums;
      c->pose_in_cur = c->d_pose_in + (size_t)par * cap * 16; c->pose_tab_cur = c->pose_tab + (size_t)par * cap; c->pose_tab_n = V; c->pose_tab_pending = true;
      int rc = L.enqueue(L.self);
      if (rc == MVR_OK) rc = ensure_pose_table(c);       // (a chain that never posed anything)
      // whatever was queued drains into the completion word, also after a failed enqueue (the caller waits for it)
      // (the final sums launch has taken the word along when it is the chain's last operation: no packet of its own then)
      const bool sent = c->signal_sent && rc == MVR_OK;
      c->signal_armed = false; c->signal_sent = false;
      if (!sent && hipStreamWriteValue32(c->stream, c->d_done, seq, 0) != hipSuccess && rc == MVR_OK) rc = set_error(c, MVR_E_HIP, "hipStreamWriteValue32");
      return rc;
    };
    auto leave = [&]() {      // back to by-value poses; the host's pose bookkeeping catches up with what the last chain computed
      c->pose_from_table = false; c->no_sync = false; c->skip_posed_pts = false; c->pose_tab_pending = false;
      std::vector<const float4 *> in; std::vector<float4 *> out; std::vector<size_t> nn; std::vector<double> Ts;
      for (int v = 0; v < V; ++v) {
        Cloud &d = c->slots[L.posed_slots[v]];
        d.pose_dev = nullptr;
        if (!d.posed_by_table) continue;                 // (a rank of a sharded run poses only the views its share touches)
        const bool was_known = d.pose_known;
        d.posed_by_table = false;
        note_pose(d, c->slots[L.raw_slots[v]], was_known, last_in.data() + 16 * (size_t)v);
        if (d.pose_known && d.last_pose_set == d.set_id) d.moved = 0.0;      // (the last queued pass searched the view at exactly this pose)
        if (d.pts_stale) {                               // the points in original order, left out while the passes were queued ahead
          in.push_back(c->slots[L.raw_slots[v]].pts); out.push_back(d.pts); nn.push_back(d.n);
          Ts.insert(Ts.end(), last_in.begin() + 16 * (size_t)v, last_in.begin() + 16 * (size_t)v + 16);
          d.pts_stale = false;
        }
      }
      if (!in.empty()) (void)launch_transform_f64_batch(c, (int)in.size(), in.data(), out.data(), nn.data(), Ts.data());
    };
    c->pose_from_table = true; c->skip_posed_pts = true;
    write_poses(parity, L.poses);
    uint32_t seq_cur = 0, seq_next = 0;
    double t0 = now_ms();
    const unsigned long long ev_first = c->blocking_events;
    int rc = enqueue_chain(parity, false, &seq_cur);
    sum[0] += now_ms() - t0;
    if (rc != MVR_OK) { (void)hipStreamSynchronize(c->stream); leave(); return rc; }
    bool fall_back = c->blocking_events != ev_first;      // the first chain itself was not in steady state after all
    for (;;) {
      const bool more = k + 1 < n_steps && !fall_back;
      bool next_failed = false;
      if (more) {          // pass k+1's chain goes into the queue now, behind the gate, while pass k runs
        t0 = now_ms();
        c->no_sync = true;
        next_failed = enqueue_chain(parity ^ 1, true, &seq_next) != MVR_OK;
        c->no_sync = false;
        sum[0] += now_ms() - t0;
      }
      t0 = now_ms();
      bool timed_out = false, async_error = false;
      rc = wait_done(c, seq_cur, &timed_out, &async_error);
      const double t1 = now_ms();
      if (rc == MVR_OK) rc = L.solve(L.self);
      sum[1] += t1 - t0; sum[2] += now_ms() - t1;
      if (rc != MVR_OK) {                              // nothing may stay behind a closed gate
        if (more) { const std::vector<double> safe = last_in; write_poses(parity ^ 1, safe.data()); open_gate(seq_next); }
        if (timed_out) {                               // (releases an injected stall too)
          const bool had_comm = c->comm != nullptr;
          const int r2 = comm_abort(c, "a pass did not finish in time: a peer failed or never arrived");
          if (had_comm) rc = r2; else (void)set_error(c, rc, "a pass of the pipelined ring run did not finish in time");
        } else if (async_error && c->comm) {           // RCCL reported an asynchronous error (wait_done only reports): abort now, the gate is open
          const std::string why = c->last_error;
          rc = comm_abort(c, why.c_str());
        }
        (void)drain_bounded(c, 5000);
        leave();
        return rc;
      }
      ++k;
      ++c->piped_passes;
      pass_done();
      if (!more) break;
      if (next_failed || !all_rigid()) {
        // the queued chain cannot be used (it is incomplete, or a pose left the nearly-rigid range the grid search accepts):
        // let it run on the poses of the pass before -- valid input, results unused -- and go on the ordinary way
        { const std::vector<double> safe = last_in; write_poses(parity ^ 1, safe.data()); }
        open_gate(seq_next);
        (void)wait_done(c, seq_next);
        MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
        fall_back = true;
        break;
      }
      write_poses(parity ^ 1, L.poses);
      open_gate(seq_next);
      parity ^= 1;
      seq_cur = seq_next;
    }
    leave();
    steady = !fall_back;
  }
  c->pipe_steady_sig = steady ? L.sig : 0; c->pipe_steady_events = c->blocking_events;
  if (timing_ms) for (int j = 0; j < 3; ++j) timing_ms[j] = sum[j];
  return MVR_OK;
}

unsigned long long pass_loop_sig(Ctx *c, int n_views, const int *posed_slots, const int *raw_slots, int ne, const int *edge_src, const int *edge_tgt,
                                 double max_dist, int reciprocal, int fma, int extra)
{
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&h](unsigned long long v) { for (int b = 0; b < 8; ++b) { h ^= (v >> (8 * b)) & 0xFFull; h *= 1099511628211ull; } };
  mix((unsigned long long)n_views); mix((unsigned long long)ne); mix((unsigned long long)reciprocal); mix((unsigned long long)fma); mix((unsigned long long)extra);
  unsigned long long md; std::memcpy(&md, &max_dist, 8); mix(md);
  for (int v = 0; v < n_views; ++v) {
    mix((unsigned long long)posed_slots[v]); mix((unsigned long long)raw_slots[v]);
    mix(c->slots[raw_slots[v]].set_id); mix((unsigned long long)c->slots[raw_slots[v]].n);
  }
  for (int e = 0; e < ne; ++e) { mix((unsigned long long)edge_src[e]); mix((unsigned long long)edge_tgt[e]); }
  return h | 1ull;
}

}  // namespace mvr

namespace {
struct RingRun {
  mvr_ctx *ctx; int n_views; const int *posed_slots, *raw_slots; int ne; const int *edge_src, *edge_tgt; double max_dist; int reciprocal, fma;
  const double *origin; int lum_iterations; double *poses, *lum_pose; float *pair_T; double *pair_n, *pair_mse; int *lum_iters; double *rows;
  std::vector<int> ss, ts; std::vector<double> pn, pm, h;
  int passes_left = 1;      // the per-pair transformations (a 3 x 3 SVD per edge: 6 us between two passes, the GPU waiting) are only computed for the LAST pass of the run -- the one the caller's pair_T describes
  static int enqueue(void *p)
  {
    RingRun &r = *static_cast<RingRun *>(p);
    Ctx *c = CTX(r.ctx);
    if (int rc = mvr_cloud_transform_batch(r.ctx, r.n_views, r.posed_slots, r.raw_slots, r.poses)) return rc;
    host_mark("  posed (orderings, sorted copies)");
    // the 32 doubles per edge go straight into pinned host memory (the final kernels write them over the bus: no copy
    // to enqueue, nothing to wait for but the stream itself)
    if (c->h_table_cap < (size_t)std::max(r.ne, 1) * 32) {
      MVR_MAY_BLOCK(c, "the host edge table has to grow");
      // (the stream is only waited for when there IS an old table it may still write to: a fresh context has none, and the wait
      // held the host behind the orderings and the posing of a registration's first pass -- 0.5 ms once its allocations were fast)
      if (c->h_table) { MVR_HIP_TRY(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->h_table); }
      c->h_table = nullptr; c->h_table_cap = 0;
      const size_t cap = (size_t)std::max(r.ne, 16) * 32;
      MVR_HIP_TRY(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_table), cap * sizeof(double), hipHostMallocMapped));
      c->h_table_cap = cap;
    }
    double *d_table = nullptr;
    MVR_HIP_TRY(c, hipHostGetDevicePointer(reinterpret_cast<void **>(&d_table), c->h_table, 0));
    if (r.ne) { if (int rc = mvr_pair_moments2_batch(r.ctx, r.ne, r.ss.data(), r.ts.data(), r.max_dist, r.reciprocal, r.fma, nullptr, nullptr, r.origin, nullptr, d_table)) return rc; }
    return MVR_OK;
  }
  static int solve(void *p)
  {
    RingRun &r = *static_cast<RingRun *>(p);
    Ctx *c = CTX(r.ctx);
    r.h.assign(c->h_table, c->h_table + (size_t)r.ne * 32);
    if (r.rows && r.ne) std::memcpy(r.rows, r.h.data(), r.h.size() * sizeof(double));
    const int rc = mvr_ring_host_step(r.n_views, r.ne, r.edge_src, r.edge_tgt, r.h.data(), r.origin, r.lum_iterations, r.poses, r.lum_pose, --r.passes_left <= 0 ? r.pair_T : nullptr,
                                      r.pair_n ? r.pair_n : r.pn.data(), r.pair_mse ? r.pair_mse : r.pm.data(), r.lum_iters);
    return rc != MVR_OK ? set_error(c, rc, "LUM solve") : MVR_OK;
  }
};
}  // namespace

extern "C" {

// n_steps outer passes of registrationLUM without a host language in between (the Python / C++ drivers spent a tenth of a
// 1.2 ms step on their own bookkeeping between the calls this chains); from the third pass on, pipelined (ring_passes).
API int mvr_ring_run(mvr_ctx *ctx, int n_steps, int n_views, const int *posed_slots, const int *raw_slots, int ne,
                     const int *edge_src, const int *edge_tgt, double max_dist, int reciprocal, int fma, const double origin[3],
                     int lum_iterations, double *poses, double *lum_pose, float *pair_T, double *pair_n, double *pair_mse,
                     int *lum_iters, double *rows, double *timing_ms)
{
  if (!ctx || n_steps < 0 || n_views < 2 || ne < 0 || !posed_slots || !raw_slots || (ne && (!edge_src || !edge_tgt)) || !origin || !poses || !lum_pose)
    return MVR_E_ARG;
  for (int e = 0; e < ne; ++e) if (edge_src[e] < 0 || edge_src[e] >= n_views || edge_tgt[e] < 0 || edge_tgt[e] >= n_views) return MVR_E_ARG;
  for (int v = 0; v < n_views; ++v) if (!slot_ok(posed_slots[v]) || !slot_ok(raw_slots[v])) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  RingRun r{ctx, n_views, posed_slots, raw_slots, ne, edge_src, edge_tgt, max_dist, reciprocal, fma, origin, lum_iterations, poses, lum_pose, pair_T,
            pair_n, pair_mse, lum_iters, rows, {}, {}, {}, {}, {}};
  r.ss.resize((size_t)ne); r.ts.resize((size_t)ne); r.pn.resize((size_t)ne); r.pm.resize((size_t)ne);
  r.passes_left = n_steps;
  for (int e = 0; e < ne; ++e) { r.ss[(size_t)e] = posed_slots[edge_src[e]]; r.ts[(size_t)e] = posed_slots[edge_tgt[e]]; }
  PassLoop L;
  L.n_views = n_views; L.posed_slots = posed_slots; L.raw_slots = raw_slots; L.poses = poses;
  L.enqueue = &RingRun::enqueue; L.solve = &RingRun::solve; L.self = &r; L.ends_with_sums = true;
  L.sig = pass_loop_sig(c, n_views, posed_slots, raw_slots, ne, edge_src, edge_tgt, max_dist, reciprocal, fma, 0);
  L.reach = max_dist;
  return ring_passes(c, n_steps, L, timing_ms);
}

// one pass: mvr_ring_run with n_steps = 1 (posing, all pairs' fused searches and sums, the host solve)
API int mvr_ring_step(mvr_ctx *ctx, int n_views, const int *posed_slots, const int *raw_slots, int ne, const int *edge_src,
                      const int *edge_tgt, double max_dist, int reciprocal, int fma, const double origin[3], int lum_iterations,
                      double *poses, double *lum_pose, float *pair_T, double *pair_n, double *pair_mse, int *lum_iters,
                      double *rows, double *timing_ms)
{
  return mvr_ring_run(ctx, 1, n_views, posed_slots, raw_slots, ne, edge_src, edge_tgt, max_dist, reciprocal, fma, origin, lum_iterations, poses,
                      lum_pose, pair_T, pair_n, pair_mse, lum_iters, rows, timing_ms);
}

// ---- target sharding over ranks (sequential mode): forward keys out, reduced keys in ----
static int forward_keys_impl(Ctx *c, Cloud &s, Cloud &t, double max_dist, int fma, long long *dev_keys)
{
  if (s.n == 0) return MVR_OK;
  SearchPlan plan;
  if (int rc = search_forward(c, s, t, 0, s.n, max_dist, fma != 0, nullptr, &plan)) return rc;
  return launch_export_keys(c, c->keys, s.n, seg_table(t), dev_keys);
}

API int mvr_nn_forward_keys(mvr_ctx *ctx, int ss, int ts, double max_dist, int fma, long long *dev_keys)
{
  if (!ctx || !slot_ok(ss) || !slot_ok(ts) || !dev_keys) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  return forward_keys_impl(c, c->slots[ss], c->slots[ts], max_dist, fma, dev_keys);
}

static int moments2_from_keys_impl(Ctx *c, Cloud &s, Cloud &t, const long long *dev_keys, double max_dist, int reciprocal,
                                   int fma, const double origin[3], double *dev_out);

API int mvr_pair_moments2_from_keys(mvr_ctx *ctx, int ss, int ts, const long long *dev_keys, double max_dist, int reciprocal,
                                    int fma, const double origin[3], double *dev_out)
{
  if (!ctx || !slot_ok(ss) || !slot_ok(ts) || !dev_keys || !origin || !dev_out) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  return moments2_from_keys_impl(c, c->slots[ss], c->slots[ts], dev_keys, max_dist, reciprocal, fma, origin, dev_out);
}

static int moments2_from_keys_impl(Ctx *c, Cloud &s, Cloud &t, const long long *dev_keys, double max_dist, int reciprocal,
                                   int fma, const double origin[3], double *dev_out)
{
  SearchPlan plan;
  if (int rc = ensure(c, c->keys, c->keys_cap, s.n)) return rc;
  if (int rc = ensure(c, c->match, c->match_cap, s.n)) return rc;
  if (c->nn_mode != 0 && s.n > 0 && t.n > 0) {
    if (int rc = ensure_index(c, s)) return rc;
    if (int rc = ensure_index(c, t)) return rc;
    plan.qperm = s.order->perm; plan.tinv = t.order->inv;
  }
  if (int rc = launch_import_keys(c, dev_keys, s.n, seg_table(t), c->keys)) return rc;
  if (reciprocal && s.n > 0) { if (int rc = search_reciprocal(c, s, t, 0, s.n, max_dist, fma != 0, &plan)) return rc; }
  return launch_accept_moments2(c, s.pts, t.pts, c->keys, c->rkeys, plan.slot, plan.qperm, plan.tinv, 0, s.n, max_dist * max_dist,
                                reciprocal != 0 && t.n > 0, origin, dev_out);
}

API int mvr_cloud_set_global_base(mvr_ctx *ctx, int slot, size_t global_begin)
{
  if (!ctx || !slot_ok(slot)) return MVR_E_ARG;
  Cloud &cl = CTX(ctx)->slots[slot];
  if (cl.n > 0xFFFFFFF0ull || global_begin + cl.n > 0xFFFFFFF0ull) return MVR_E_ARG;
  cl.segs.assign(1, Seg{0u, (uint32_t)cl.n, (uint32_t)global_begin});
  return MVR_OK;
}

API int mvr_cloud_append_range(mvr_ctx *ctx, int dst, int src, size_t src_begin, size_t count, size_t global_begin)
{
  if (!ctx || !slot_ok(dst) || !slot_ok(src) || dst == src) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  Cloud &d = c->slots[dst];
  const Cloud &s = c->slots[src];
  if (src_begin > s.n || count > s.n - src_begin) return set_error(c, MVR_E_ARG, "append range outside the source cloud");
  if (d.n + count > 0xFFFFFFF0ull || global_begin + count > 0xFFFFFFF0ull) return MVR_E_ARG;
  if (d.segs.empty() && d.n) d.segs.assign(1, Seg{0u, (uint32_t)d.n, 0u});      // what was there keeps its own numbering
  if ((int)d.segs.size() >= kMaxSegs) return set_error(c, MVR_E_ARG, "too many segments in a sharded cloud");
  if (int rc = cloud_reserve(c, d, d.n + count, true)) return rc;
  if (count) MVR_HIP_TRY(c, hipMemcpyAsync(d.pts + d.n, s.pts + src_begin, count * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
  d.segs.push_back(Seg{(uint32_t)d.n, (uint32_t)count, (uint32_t)global_begin});
  d.n += count;
  d.has_normals = false;
  new_point_set(c, d);
  return MVR_OK;
}

API int mvr_pair_moments2_from_corr(mvr_ctx *ctx, int ss, int ts, const int32_t *query, const int32_t *match,
                                    size_t m, const double origin[3], mvr_pair_moments2_t *out)
{
  if (!ctx || !slot_ok(ss) || !slot_ok(ts) || !origin || !out || (m && (!query || !match))) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  const Cloud &s = c->slots[ss], &t = c->slots[ts];
  std::memset(out, 0, sizeof *out);
  for (int k = 0; k < 3; ++k) out->origin[k] = origin[k];
  if (s.n == 0) return MVR_OK;
  std::vector<int32_t> hm(s.n, -1);
  for (size_t k = 0; k < m; ++k) {
    if (query[k] < 0 || (size_t)query[k] >= s.n || match[k] < 0 || (size_t)match[k] >= t.n)
      return set_error(c, MVR_E_ARG, "correspondence index out of range");
    hm[query[k]] = match[k];
  }
  if (int rc = ensure(c, c->match, c->match_cap, s.n)) return rc;
  MVR_HIP_TRY(c, hipMemcpyAsync(c->match, hm.data(), s.n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));     // hm goes out of scope
  if (int rc = launch_moments2(c, s.pts, t.pts, c->match, nullptr, nullptr, 0, s.n, origin, c->moments)) return rc;
  if (int rc = read_moments(c, 32)) return rc;
  std::memcpy(out, c->h_moments, sizeof *out);
  return MVR_OK;
}

// ---- mvr_icp_align against a target MADE OF POSED SCANS (the growing model of the sequential mode) --------------------
// Everything the part-by-part search needs, or false when the align has to take the culled kernel: every part's scan and
// the source's scan must be resident in canonical form (their grids are found or built here, batched), the target's
// grid-ordered coordinates are written for the parts that do not have them yet, the current source gets the pose and the
// grid of the posed source it is a copy of, and the part table goes to the device.
// the current source of an align is an exact copy of a posed scan: give it that scan's pose, grid and grid-ordered
// coordinates, so that the REVERSE searches (matched targets looking for their nearest source point, each with the bound
// its forward match gives) can walk the source's grid instead of the culled kernel's box hierarchy
static bool prepare_source_grid(Ctx *c, const Cloud &posed_src, Cloud &cur, double max_dist)
{
  if (!c->seq_search || c->nn_mode == 0 || !posed_src.pose_known || posed_src.fin_known || posed_src.n == 0 || posed_src.n != cur.n) return false;
  if (!(max_dist * max_dist < (double)FLT_MAX)) return false;
  Cloud *src_canon = nullptr;
  for (Cloud &o : c->slots) if (o.set_id == posed_src.set_id && o.canonical && o.n == posed_src.n) { src_canon = &o; break; }
  if (!src_canon) return false;
  if (ensure_grids(c, &src_canon, 1, max_dist + 0.5, c->stream, nullptr) != MVR_OK || !src_canon->grid || src_canon->grid->n != src_canon->n) return false;
  // (the align searches the posed source where it lies: its grid-ordered coordinates may be current already -- the posing launch
  // writes them along when the set has its grid)
  const bool current = &cur == &posed_src && cur.gcoords_valid && cur.grid == src_canon->grid && cur.gsorted;
  cur.pose_known = true; cur.pose_stretch = posed_src.pose_stretch; std::memcpy(cur.pose, posed_src.pose, sizeof cur.pose);
  cur.grid = src_canon->grid;
  if (current) return true;
  cur.gcoords_valid = false;
  Cloud *one = &cur;
  return refresh_grid_coords_batch(c, &one, 1) == MVR_OK && cur.gcoords_valid;
}

static bool prepare_parts_search(Ctx *c, const Cloud &posed_src, Cloud &cur, Cloud &tgt, double max_dist)
{
  if (c->seq_search < 2 || c->nn_mode == 0 || !posed_src.pose_known || posed_src.fin_known || posed_src.n == 0) return false;
  if (!(max_dist * max_dist < (double)FLT_MAX)) return false;
  if (tgt.parts.empty() && (tgt.pose_known || tgt.fin_known) && tgt.segs.empty() && tgt.n > 0) {
    // a single posed scan (the model before its first append: view 0 under its pose) is a composite of one part
    GridPart gp;
    gp.set_id = tgt.set_id; gp.n = tgt.n; gp.base = 0; gp.kind = tgt.fin_known ? 2 : 1;
    std::memcpy(gp.pose, tgt.pose, sizeof gp.pose); std::memcpy(gp.fin, tgt.fin, sizeof gp.fin);
    gp.grid = nullptr; gp.gs_filled = false;
    tgt.parts.push_back(gp);
    tgt.gcoords_valid = false;              // (a posed scan's gsorted[] carries original indices without a base: same thing for base 0, but written by another kernel)
  }
  if (tgt.parts.empty()) return false;
  size_t total = 0;
  for (const GridPart &gp : tgt.parts) { if (gp.base != total || gp.n == 0 || (gp.kind != 1 && gp.kind != 2)) return false; total += gp.n; }
  if (total != tgt.n || tgt.n > 0xFFFFFFF0ull) return false;
  auto canon_of = [&](unsigned long long set_id, size_t n) -> Cloud * {
    for (Cloud &o : c->slots) if (o.set_id == set_id && o.canonical && o.n == n) return &o;
    return nullptr;
  };
  std::vector<Cloud *> canon;
  for (const GridPart &gp : tgt.parts) { Cloud *o = canon_of(gp.set_id, gp.n); if (!o) return false; canon.push_back(o); }
  Cloud *src_canon = canon_of(posed_src.set_id, posed_src.n);
  if (!src_canon) return false;
  canon.push_back(src_canon);
  if (ensure_grids(c, canon.data(), (int)canon.size(), max_dist + 0.5, c->stream, nullptr) != MVR_OK) return false;
  for (Cloud *o : canon) if (!o->grid || o->grid->n != o->n) return false;
  // the target's coordinates in grid order, part after part
  const float4 *before = tgt.gsorted;
  if (ensure(c, tgt.gsorted, tgt.gsorted_cap, std::max(tgt.cap, tgt.n)) != MVR_OK) return false;
  if (tgt.gsorted != before || !tgt.gcoords_valid) for (GridPart &gp : tgt.parts) gp.gs_filled = false;
  for (size_t k = 0; k < tgt.parts.size(); ++k) {
    GridPart &gp = tgt.parts[k];
    gp.grid = canon[k]->grid;
    if (gp.grid->ready && !gp.grid->ready_waited) { if (hipStreamWaitEvent(c->stream, gp.grid->ready, 0) != hipSuccess) return false; gp.grid->ready_waited = true; }
    if (!gp.gs_filled && fill_part_coords(c, tgt, gp) != MVR_OK) return false;
  }
  tgt.gcoords_valid = true;
  // the current source: an exact copy of the posed source, so its pose and its scan's grid
  cur.pose_known = true; cur.pose_stretch = posed_src.pose_stretch; std::memcpy(cur.pose, posed_src.pose, sizeof cur.pose);
  cur.grid = src_canon->grid; cur.gcoords_valid = false;
  Cloud *one = &cur;
  if (refresh_grid_coords_batch(c, &one, 1) != MVR_OK || !cur.gcoords_valid) return false;
  // the part table
  const size_t K = tgt.parts.size();
  if (c->parts_cap < K) {
    if (may_block(c, "not in steady state: the part table has to grow") != MVR_OK) return false;
    (void)hipStreamSynchronize(c->stream);
    if (c->d_parts) (void)hipFree(c->d_parts);
    if (c->h_parts) (void)hipHostFree(c->h_parts);
    c->d_parts = nullptr; c->h_parts = nullptr; c->parts_cap = 0;
    const size_t cap = std::max<size_t>(64, 2 * K);
    if (hipMalloc(&c->d_parts, cap * sizeof(PartDesc)) != hipSuccess || hipHostMalloc(reinterpret_cast<void **>(&c->h_parts), cap * sizeof(PartDesc)) != hipSuccess) return false;
    c->parts_cap = cap;
  }
  for (size_t k = 0; k < K; ++k) {
    const GridPart &gp = tgt.parts[k];
    const CellGrid &g = *gp.grid;
    PartDesc &d = c->h_parts[k];
    d.gts = tgt.gsorted + gp.base; d.start = g.start; d.dir = g.dir; d.recs = g.recs; d.dt_shift = g.dt_shift; for (int k3 = 0; k3 < 3; ++k3) d.dtdim[k3] = g.dtdim[k3]; d.dt = g.dt;
    for (int j = 0; j < 3; ++j) { d.lo[j] = g.lo[j]; d.dim[j] = g.dim[j]; }
    d.inv_h = g.inv_h; d.h = g.h; d.dt_max = g.dt_steps; d.base = (uint32_t)gp.base; d.n = (uint32_t)gp.n;
    // canonical -> posed: the f64 pose, then (kind 2) the align's f32 matrix
    double M[16];
    std::memcpy(M, gp.pose, sizeof M);
    if (gp.kind == 2) { double F[16]; for (int j = 0; j < 16; ++j) F[j] = (double)gp.fin[j]; mvr_mat4d_mul(F, gp.pose, M); }
    if (M[3] != 0.0 || M[7] != 0.0 || M[11] != 0.0 || M[15] != 1.0) return false;
    double e2 = 0.0;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        const double dd = M[4 * a] * M[4 * b] + M[4 * a + 1] * M[4 * b + 1] + M[4 * a + 2] * M[4 * b + 2], x = dd - (a == b ? 1.0 : 0.0);
        e2 += x * x;
      }
    const double e = std::sqrt(e2);
    if (!(e <= 1e-3)) return false;                        // (note_pose's bar: beyond it a ball is no ball any more)
    d.stretch = std::nextafterf((float)((1.0 / std::sqrt(1.0 - e)) * (1.0 + 1e-6)), INFINITY);
    const double A[3][3] = {{M[0], M[4], M[8]}, {M[1], M[5], M[9]}, {M[2], M[6], M[10]}};
    const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) + A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    const double id = 1.0 / det;
    const double I[3][3] = {{(A[1][1] * A[2][2] - A[1][2] * A[2][1]) * id, (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * id, (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * id},
                            {(A[1][2] * A[2][0] - A[1][0] * A[2][2]) * id, (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * id, (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * id},
                            {(A[1][0] * A[2][1] - A[1][1] * A[2][0]) * id, (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * id, (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * id}};
    for (int r = 0; r < 3; ++r) {
      for (int j = 0; j < 3; ++j) d.minv[4 * r + j] = I[r][j];
      d.minv[4 * r + 3] = -(I[r][0] * M[12] + I[r][1] * M[13] + I[r][2] * M[14]);
    }
  }
  if (hipMemcpyAsync(c->d_parts, c->h_parts, K * sizeof(PartDesc), hipMemcpyHostToDevice, c->stream) != hipSuccess) return false;
  return true;
}

// the search of one iteration through the parts: forward = nn_parts_kernel (+ the culled kernel for the queries it
// flags), reverse = the grid walk over the CURRENT SOURCE's grid for the matched targets, compacted by the kernels of
// the fused pass (no hipCUB).  Leaves keys / slot / rkeys as run_search does.
static int run_search_parts(Ctx *c, Cloud &cur, Cloud &tgt, double max_dist, bool reciprocal, bool fma, SearchPlan *plan, bool parts_forward)
{
  const size_t ns = cur.n, nt = tgt.n;
  const double max2 = max_dist * max_dist;
  const float cap2 = cap_from_max2(max2);
  *plan = SearchPlan();
  if (int rc = ensure(c, c->keys, c->keys_cap, ns)) return rc;
  if (int rc = ensure(c, c->match, c->match_cap, ns)) return rc;
  if (int rc = ensure(c, c->rkeys, c->rkeys_cap, std::max(ns, std::min(ns, nt)))) return rc;
  if (int rc = ensure(c, c->bheavy, c->bheavy_cap, ns)) return rc;
  if (int rc = ensure_index(c, cur)) return rc;
  if (int rc = ensure_index(c, tgt)) return rc;
  plan->qperm = cur.order->perm; plan->tinv = tgt.order->inv;
  if (ns > 0xFFFFFFF0ull || nt > 0xFFFFFFF0ull) return set_error(c, MVR_E_ARG, "cloud too large for 32-bit indices");
  // every query starts from the distance of the point it matched when this scan was last aligned, if it was (Ctx::seq_seed)
  const uint32_t *qb = nullptr;
  if (c->seq_seed) {
    auto it = c->seq_seeds.find(cur.set_id);
    if (it != c->seq_seeds.end() && it->second.d && it->second.n == ns) {
      if (int rc = ensure(c, c->seed_bound, c->seed_bound_cap, ns)) return rc;
      if (int rc = launch_seed_to_bound(c, cur.sorted, ns, tgt.sorted, nt, it->second.d, fma, c->seed_bound)) return rc;
      qb = c->seed_bound;
    }
  }
  // seq_search 2: forward through the parts' grids (measured slower)
  const bool through_parts = parts_forward && c->seq_search == 2;
  // seq_search 3: forward through ONE grid over the model's own coordinates (Cloud::mgrid): the thread-per-query walk, then the
  // wide bounded queries a wave each and the flagged 64-query sets a block each, as a fused pass's forward searches
  bool model_ok = false;
  if (c->seq_search == 3 && !through_parts) { if (int rc = ensure_model_grid(c, tgt, max_dist + 0.5, &model_ok)) return rc; }
  if (model_ok) {
    if (!c->bwide_count) {
      MVR_MAY_BLOCK(c, "the wide-query counters are not allocated yet");
      MVR_HIP_TRY(c, hipMalloc(&c->bwide_count, 3 * kWideCounters * sizeof(uint32_t)));
      MVR_HIP_TRY(c, hipMemsetAsync(c->bwide_count, 0, 3 * kWideCounters * sizeof(uint32_t), c->stream));
    }
    if (int rc = ensure(c, c->bwide, c->bwide_cap, std::max(ns, std::min(ns, nt)))) return rc;
    if (int rc = ensure(c, c->bcull_sets, c->bcull_sets_cap, ns / 64 + 2)) return rc;
    GridPair f = make_model_pair(cur, tgt, c->keys);
    f.qbound = qb;
    f.heavy = c->bheavy;
    f.wide_list = c->bwide; f.wide_count = c->bwide_count + kWideCounters - 2;
    f.cull_sets = c->bcull_sets; f.cull_count = c->bwide_count + 2 * kWideCounters + kWideCounters - 1;
    MVR_HIP_TRY(c, hipMemsetAsync(f.wide_count, 0, sizeof(uint32_t), c->stream));
    MVR_HIP_TRY(c, hipMemsetAsync(f.cull_count, 0, sizeof(uint32_t), c->stream));
    if (int rc = launch_nn_grid_batch(c, &f, 1, cap2, fma)) return rc;
    if (c->grid_debug) {           // diagnostics: how many queries left the thread-per-query walk
      std::vector<uint8_t> hv(ns); uint32_t wc = 0, sc = 0;
      MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
      MVR_HIP_TRY(c, hipMemcpy(hv.data(), c->bheavy, ns, hipMemcpyDeviceToHost));
      MVR_HIP_TRY(c, hipMemcpy(&wc, f.wide_count, 4, hipMemcpyDeviceToHost));
      MVR_HIP_TRY(c, hipMemcpy(&sc, f.cull_count, 4, hipMemcpyDeviceToHost));
      size_t fl = 0;
      for (size_t i = 0; i < ns; ++i) fl += hv[i];
      if (qb) {
        std::vector<uint32_t> sbv(ns);
        MVR_HIP_TRY(c, hipMemcpy(sbv.data(), qb, ns * 4, hipMemcpyDeviceToHost));
        size_t z = 0, in = 0, out = 0, none = 0, fz = 0, fin = 0, fout = 0, fnone = 0;
        uint32_t capb; std::memcpy(&capb, &cap2, 4);
        for (size_t i = 0; i < ns; ++i) {
          const uint32_t v = sbv[i];
          if (v == 0) { ++z; fz += hv[i]; } else if (v == 0xFFFFFFFFu) { ++none; fnone += hv[i]; } else if (v <= capb) { ++in; fin += hv[i]; } else { ++out; fout += hv[i]; }
        }
        std::fprintf(stderr, "[model]   start bounds: %zu zero (%zu flagged), %zu within the cap (%zu), %zu beyond (%zu), %zu none (%zu)\n", z, fz, in, fin, out, fout, none, fnone);
      }
      std::fprintf(stderr, "[model] nt %zu, h %.3f, dim %d x %d x %d, %zu queries (%s): %u to a wave each, %zu flagged in %u sets\n", nt, tgt.mgrid->h, tgt.mgrid->dim[0], tgt.mgrid->dim[1],
                   tgt.mgrid->dim[2], ns, qb ? "seeded" : "no seeds", wc, fl, sc);
    }
    if (c->seq_model_tail == 0) { if (int rc = launch_nn_grid_tail_batch(c, &f, 1, cap2, fma)) return rc; }
    else {
      // the wide bounded queries a wave each; the flagged sets through the culled kernel over the model's composite ordering
      // (its box hierarchy throws out the sets that have nothing within the cap at once), keys by sorted position, then merged
      if (int rc = launch_nn_grid_wide_batch(c, &f, 1, cap2, fma)) return rc;
      CullPair p = make_cull_pair(cur, 0, ns, c->bheavy, tgt, c->rkeys);
      p.qbound = qb; p.setlist = f.cull_sets; p.setcount = f.cull_count;
      // (several lanes per query -- grid_lanes 2 / 4 / 8 -- write no set list: the culled kernel then starts a block per set and the
      // blocks without a flagged query leave at once)
      if (c->grid_lanes == 2 || c->grid_lanes == 4 || c->grid_lanes == 8) { p.setlist = nullptr; p.setcount = nullptr; if (int rc = launch_nn_cull_batch(c, &p, 1, cap2, fma)) return rc; }
      else
      if (int rc = launch_nn_cull_list_batch(c, &p, 1, cap2, fma)) return rc;
      if (int rc = launch_merge_flagged_keys(c, cur.sorted, c->bheavy, c->rkeys, ns, c->keys)) return rc;
    }
  } else
  if (!through_parts) {
    // forward: the culled kernel over the target's composite index (keys are written by exactly one wave per query)
    CullPair fp = make_cull_pair(cur, 0, ns, nullptr, tgt, c->keys);
    fp.qbound = qb;
    if (int rc = launch_nn_cull_batch(c, &fp, 1, cap2, fma)) return rc;
  } else {
  if (int rc = launch_nn_parts(c, cur, (int)tgt.parts.size(), cap2, fma, c->keys, c->bheavy, qb)) return rc;
  if (c->grid_debug) {           // diagnostics: how many queries the parts could not answer
    std::vector<uint8_t> hv(ns);
    MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
    MVR_HIP_TRY(c, hipMemcpy(hv.data(), c->bheavy, ns, hipMemcpyDeviceToHost));
    size_t f = 0, sets = 0;
    for (size_t i = 0; i < ns; i += 64) { size_t g = 0; for (size_t j = i; j < std::min(ns, i + 64); ++j) g += hv[j]; f += g; sets += g != 0; }
    std::fprintf(stderr, "[parts] %zu parts, %zu queries: %zu left to the culled kernel in %zu of %zu sets\n", tgt.parts.size(), ns, f, sets, (ns + 63) / 64);
  }
  {   // the flagged queries: the culled kernel over the composite index, keys by SORTED position into rkeys[], then merged
    CullPair p = make_cull_pair(cur, 0, ns, c->bheavy, tgt, c->rkeys);
    p.qbound = qb;
    if (int rc = launch_nn_cull_batch(c, &p, 1, cap2, fma)) return rc;
    if (int rc = launch_merge_flagged_keys(c, cur.sorted, c->bheavy, c->rkeys, ns, c->keys)) return rc;
  }
  }
  uint32_t *seed_out = nullptr;
  if (c->seq_seed) {
    if (c->seq_seeds.size() >= 256 && !c->seq_seeds.count(cur.set_id)) {      // (scans that are gone: start over rather than grow without bound)
      MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
      for (auto &kv : c->seq_seeds) if (kv.second.d) (void)hipFree(kv.second.d);
      c->seq_seeds.clear();
    }
    Ctx::SeedBuf &sb = c->seq_seeds[cur.set_id];
    if (int rc = ensure(c, sb.d, sb.cap, ns)) return rc;
    sb.n = ns;
    seed_out = sb.d;
    // (with reciprocal searches the launch that sets the reverse searches' start bounds walks the same keys: it leaves the seeds)
    if (!reciprocal || nt == 0) { if (int rc = launch_keys_to_seed(c, cur.sorted, ns, c->keys, tgt.order->inv, sb.d)) return rc; }
  }
  if (!reciprocal || nt == 0) return MVR_OK;
  // reverse: the distinct matched targets, each starting from the distance of the source that matched it
  if (int rc = ensure(c, c->bound, c->bound_cap, nt)) return rc;
  if (!c->bwide_count) {
    MVR_MAY_BLOCK(c, "the wide-query counters are not allocated yet");
    MVR_HIP_TRY(c, hipMalloc(&c->bwide_count, 3 * kWideCounters * sizeof(uint32_t)));
    MVR_HIP_TRY(c, hipMemsetAsync(c->bwide_count, 0, 3 * kWideCounters * sizeof(uint32_t), c->stream));
  }
  uint32_t *rev_wide_count = c->bwide_count + kWideCounters - 1;      // (a counter the fused pass does not use; put to zero by the bounds launch)
  if (int rc = launch_seed_bounds(c, c->keys, plan->qperm, 0, ns, max2, plan->tinv, nt, c->bound, seed_out, rev_wide_count)) return rc;
  const size_t nl = std::min(ns, nt), chunks = (nt + 255) / 256;
  if (int rc = ensure(c, c->slot, c->slot_cap, nt)) return rc;
  if (int rc = ensure(c, c->list, c->list_cap, nl)) return rc;
  if (int rc = ensure(c, c->bchunks, c->bchunks_cap, chunks + 8)) return rc;
  if (int rc = ensure(c, c->bwide, c->bwide_cap, nl)) return rc;
  GlueBatch gb;
  gb.max2 = max2; gb.reciprocal = 1; gb.origin[0] = gb.origin[1] = gb.origin[2] = 0.0;
  GluePair &g = gb.p[0];
  g.bound = c->bound; g.list = c->list; g.slot = c->slot; g.chunks = c->bchunks; g.qcount = c->count; g.nt = nt;
  if (int rc = launch_compact_flags_batch(c, gb, 1)) return rc;
  plan->slot = c->slot; plan->count = c->count;
  GridPair rev = make_grid_pair(tgt, 0, nl, cur, c->rkeys);
  rev.qlist = c->list; rev.qcount = c->count; rev.qbound = c->bound;
  rev.wide_list = c->bwide; rev.wide_count = rev_wide_count;
  if (int rc = launch_nn_grid_batch(c, &rev, 1, cap2, fma)) return rc;
  return launch_nn_grid_wide_batch(c, &rev, 1, cap2, fma);
}

// pcl::registration::DefaultConvergenceCriteria::hasConverged, evaluated after every iteration on the
// INCREMENTAL transformation and the mean squared correspondence distance of that iteration
namespace {
struct Criteria {
  double rot_thr, trans_thr, rel_mse, abs_mse = 1e-12, prev_mse = DBL_MAX;
  int max_iterations;
  explicit Criteria(const mvr_icp_params *p)
      : rot_thr(1.0 - p->transformation_epsilon), trans_thr(p->transformation_epsilon), rel_mse(p->euclidean_fitness_eps), max_iterations(p->max_iterations) {}
  bool converged(const float tr[16], double cur_mse, int iters, int *state)
  {
    *state = MVR_CONV_NOT;
    if (iters >= max_iterations) { *state = MVR_CONV_ITERATIONS; return true; }
    const double cos_angle = 0.5 * ((double)tr[0] + (double)tr[5] + (double)tr[10] - 1.0);
    const double t2 = (double)tr[12] * (double)tr[12] + (double)tr[13] * (double)tr[13] + (double)tr[14] * (double)tr[14];
    if (cos_angle >= rot_thr && t2 <= trans_thr) { *state = MVR_CONV_TRANSFORM; return true; }
    if (std::fabs(cur_mse - prev_mse) < abs_mse) { *state = MVR_CONV_ABS_MSE; return true; }
    if (std::fabs(cur_mse - prev_mse) / prev_mse < rel_mse) { *state = MVR_CONV_REL_MSE; return true; }
    prev_mse = cur_mse;
    return false;
  }
};
}  // namespace

API int mvr_icp_align(mvr_ctx *ctx, int ss, int ts, int os, const mvr_icp_params *p, float T_out[16],
                      mvr_icp_stats *st)
{
  if (!ctx || !slot_ok(ss) || !slot_ok(ts) || (os >= 0 && !slot_ok(os)) || !p || !T_out) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  const double t0 = now_ms();
  const size_t ns = c->slots[ss].n;
  if (p->point_to_plane && c->slots[ts].n && !c->slots[ts].has_normals)
    return set_error(c, MVR_E_ARG, "point-to-plane needs target normals (mvr_cloud_upload_normals)");
  // App. A.1: input_transformed = *input (guess == identity).  The first iteration searches the source WHERE IT LIES: the
  // copy PCL makes is only needed once the source moves (a second iteration, which the reference's settings never reach) --
  // a device copy, an index refresh and a grid posing per align that nothing read differently (-25 us of 0.45 ms)
  Cloud &scratch = c->slots[kScratchCur];
  Cloud *curp = &c->slots[ss];
  bool on_scratch = false;
  const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (c->nn_mode != 0 && ns) { if (int rc = ensure_index(c, c->slots[ss])) return rc; }   // sort the source once, share it
  MVR_HIP_TRY(c, hipMemsetAsync(c->evals + kEvalRegion, 0, kEvalRegion * sizeof(uint64_t), c->stream));   // running totals of the culled kernel
  float fin[16], tr[16];
  std::memcpy(fin, I, sizeof I); std::memcpy(tr, I, sizeof I);
  Criteria crit(p);          // DefaultConvergenceCriteria (App. A.4)
  double cur_mse = 0.0, evals = 0.0, fwdq = 0.0, evals_total = 0.0;
  int iters = 0, converged = 0, state = MVR_CONV_NOT, ncorr = 0, status = MVR_OK;
  Cloud &tgt = c->slots[ts];
  // a target made of posed scans (the sequential mode's model) is searched through the scans' grids -- in the first
  // iteration, while the current source still IS the posed source (later iterations, which the reference's settings never
  // reach, take the culled kernel: the moved source's coordinates are no longer a known pose of its scan)
  // (seq_search 2: forward through the parts as well; measured on the 12 x 200k sweep it does NOT pay -- DESIGN.md 4.5 -- so
  // the default, 1, keeps the culled kernel for the forward search and takes the grid for the reverse one)
  const bool parts_ok = ns > 0 && tgt.n > 0 && ts != ss && prepare_parts_search(c, c->slots[ss], *curp, tgt, p->max_corr_dist);
  const bool rev_grid_ok = parts_ok || (ns > 0 && tgt.n > 0 && ts != ss && p->use_reciprocal && tgt.n <= 0xFFFFFFF0ull &&
                                        prepare_source_grid(c, c->slots[ss], *curp, p->max_corr_dist));
  // (point-to-point: the iteration's row comes to the host from the last sums launch itself, no copy, no synchronise)
  const bool spin = c->align_spin && !p->point_to_plane && ensure_align_row(c) == MVR_OK;
  do {
    double ev = 0.0;
    SearchPlan plan;
    if (rev_grid_ok && iters == 0) { if (int rc = run_search_parts(c, *curp, tgt, p->max_corr_dist, p->use_reciprocal != 0, p->fma_dist != 0, &plan, parts_ok)) return rc; }
    else
    if (int rc = run_search(c, *curp, tgt, 0, ns, p->max_corr_dist, p->use_reciprocal != 0, p->fma_dist != 0, &ev, &plan)) return rc;
    if (int rc = launch_pass1(c, curp->pts, tgt.pts, c->keys, c->rkeys, plan.slot, plan.count, plan.qperm, plan.tinv, 0, ns,
                              p->max_corr_dist * p->max_corr_dist, p->use_reciprocal != 0 && tgt.n > 0, c->match,
                              c->moments)) return rc;
    if (p->point_to_plane) { if (int rc = launch_p2plane(c, curp->pts, tgt.pts, tgt.nrm, c->match, plan.qperm, 0, ns, c->moments + 32)) return rc; }
    else if (int rc = launch_pass2(c, curp->pts, tgt.pts, c->match, plan.qperm, 0, ns, c->moments, (st && c->nn_mode != 0) ? c->evals + kEvalRegion : nullptr,
                               spin ? c->d_align : nullptr, spin ? ++c->align_seq : 0u)) return rc;
    // the search kernels' running evaluation totals ride along with the moments (the statistic needs neither a wait nor a copy
    // of its own: the last sums launch adds the counters up into moments[18]; point-to-plane keeps the copy)
    if (st && c->nn_mode != 0 && p->point_to_plane)
      MVR_HIP_TRY(c, hipMemcpyAsync(c->h_evals, c->evals + kEvalRegion, kEvalRegion * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    if (spin) { if (int rc = spin_moments(c, c->align_seq, 19)) return rc; }
    else if (int rc = read_moments(c, p->point_to_plane ? 64 : 19)) return rc;
    const double *h = c->h_moments;
    if (st && c->nn_mode != 0 && !p->point_to_plane) evals_total = h[18];
    fwdq += (double)ns;
    evals += (c->nn_mode != 0) ? 0.0 : ev + h[17] * (double)ns;   // forward Ns*Nt + reverse Nt'*Ns (brute force)
    ncorr = (int)h[0];
    if (h[0] < 3.0) { state = MVR_CONV_NO_CORRESPONDENCES; converged = 0; status = MVR_E_NOCORR; break; }
    if (p->point_to_plane) {
      if (p2plane_solve(h + 32, h + 32 + 21, tr) != MVR_OK) { state = MVR_CONV_NO_CORRESPONDENCES; converged = 0; status = MVR_E_SINGULAR; break; }
    } else {
      umeyama_from_moments(h + 1, h + 4, h + 8, tr, nullptr);
    }
    cur_mse = h[7];
    mvr_mat4f_mul(tr, fin, fin);
    ++iters;
    converged = crit.converged(tr, cur_mse, iters, &state) ? 1 : 0;
    if (!converged) {          // (the moved source is only needed by another iteration: the output is final * the ORIGINAL input)
      if (!on_scratch) {         // the source moves for the first time: into the scratch cloud, which the later iterations search and move
        scratch.n = 0;
        if (int rc = cloud_reserve(c, scratch, ns, false)) return rc;
        if (int rc = launch_transform_f32(c, curp->pts, scratch.pts, ns, tr)) return rc;
        scratch.n = ns;
        inherit_point_set(scratch, c->slots[ss]);
        on_scratch = true;
        curp = &scratch;
      } else if (int rc = launch_transform_f32(c, curp->pts, curp->pts, ns, tr)) return rc;
      curp->stale_coords();
      curp->pose_known = false; curp->fin_known = false; curp->grid.reset(); curp->parts.clear();
    }
  } while (!converged);
  // output = final * (*input), from the ORIGINAL input: alias-safe (registrator.cpp:920)
  if (os >= 0) {
    if (os != ss) { c->slots[os].n = 0; if (int rc = cloud_reserve(c, c->slots[os], ns, false)) return rc; }
    if (int rc = launch_transform_f32(c, c->slots[ss].pts, c->slots[os].pts, ns, fin)) return rc;
    c->slots[os].n = ns;
    if (os != ss) {
      inherit_point_set(c->slots[os], c->slots[ss]);
      // the output is the f32 matrix `fin` applied to a cloud with a known f64 pose: a posed copy still, by known arithmetic
      // (what the sequential mode appends to its model, registrator.cpp:576 -- see GridPart)
      const Cloud &in = c->slots[ss];
      if (in.pose_known && !in.fin_known) {
        Cloud &o = c->slots[os];
        o.fin_known = true; std::memcpy(o.fin, fin, sizeof fin); std::memcpy(o.pose, in.pose, sizeof o.pose); o.pose_stretch = in.pose_stretch; o.grid = in.grid;
      }
    } else {
      // the aliased align(*source_) of registrator.cpp:920: the slot's points moved in place by a FLOAT matrix -- they are
      // neither the set's upload coordinates nor a known f64 pose of them any more, and a grid built in the old frame
      // must not be walked with them (the other in-place paths: mvr_cloud_transform, _f32, _batch)
      c->slots[os].forget_pose();
    }
    c->slots[os].stale_coords();
    if (c->slots[ss].has_normals && ns) {      // ICP::transformCloud rotates the source normals too
      if (int rc = ensure(c, c->slots[os].nrm, c->slots[os].nrm_cap, ns)) return rc;
      if (int rc = launch_rotate_normals_f32(c, c->slots[ss].nrm, c->slots[os].nrm, ns, fin)) return rc;
    }
    c->slots[os].has_normals = c->slots[ss].has_normals;
  }
  std::memcpy(T_out, fin, sizeof fin);
  if (st && c->nn_mode != 0) {
    evals = evals_total;
    if (p->point_to_plane) { evals = 0.0; for (int k = 0; k < kEvalShards; ++k) evals += (double)c->h_evals[(size_t)k * kEvalStride]; }
  }
  if (st) {
    st->iterations = iters; st->converged = converged; st->state = state; st->n_corr = ncorr; st->mse = cur_mse;
    st->evals = evals; st->fwd_queries = fwdq; st->ms = now_ms() - t0;
  }
  return status;
}

// ---- sequential mode with the growing target SHARDED by points over the ranks (registrator.cpp:563-577 is
// loop-carried and does not shard by pair).  One IterativeClosestPoint::align of the full source (every rank holds it)
// against the distributed target, as one native loop per rank:
//   forward NN in the rank's shard -> Ns signed keys (d2 bits << 32 | GLOBAL target index)
//   ncclAllReduce(ncclInt64, ncclMin) of the keys on the context's stream     (ties -> lowest global index: the single-GPU rule)
//   reciprocal check + raw second moments of the matches whose target the rank OWNS
//   ncclAllReduce(ncclDouble, ncclSum) of the 32 sums (+ the iteration's failure count)
//   one D2H of 40 doubles, Umeyama on the host, convergence criteria, transform of the current source
// A rank whose local work fails still joins both collectives of the iteration, with neutral contributions and a raised
// failure count: every rank then leaves the loop in the same iteration with an error instead of one rank leaving its
// peers inside a collective.  Without a communicator the context is a world of one and the collectives are no-ops.
API int mvr_seq_align_sharded(mvr_ctx *ctx, int ss, int ts, int os, const mvr_icp_params *p, const double origin[3], float T_out[16],
                              mvr_icp_stats *st)
{
  if (!ctx || !slot_ok(ss) || !slot_ok(ts) || (os >= 0 && !slot_ok(os)) || !p || !origin || !T_out || ss == ts || os == ts) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  if (p->point_to_plane) return set_error(c, MVR_E_ARG, "the sharded align is point-to-point (the reference's estimator)");
  if (c->comm_broken) return set_error(c, MVR_E_RCCL, "the communicator of this context was aborted");
  const double t0 = now_ms();
  const size_t ns = c->slots[ss].n;
  Cloud &cur = c->slots[kScratchCur];
  cur.n = 0;
  // Everything that can fail for LOCAL reasons is reported through the iteration's collectives (neutral keys, a zero row, a failure
  // count of 1), so that every rank leaves the iteration together (the set-up below and the transform inside the loop
  // used to return at once, leaving the peers to wait_timeout_ms).  Only the two buffers the collectives themselves run on have
  // to exist for that: a rank that cannot even allocate those cannot report, and its peers leave through the timeout.
  if (int rc = ensure(c, c->seq_keys, c->seq_keys_cap, std::max<size_t>(ns, 1))) return rc;
  if (!c->seq_row) MVR_HIP_TRY(c, hipMalloc(&c->seq_row, 40 * sizeof(double)));
  int carry = cloud_reserve(c, cur, ns, false);
  if (carry == MVR_OK && ns) { const hipError_t e = hipMemcpyAsync(cur.pts, c->slots[ss].pts, ns * sizeof(float4), hipMemcpyDeviceToDevice, c->stream); if (e != hipSuccess) carry = set_error(c, MVR_E_HIP, "copy of the source", e); }
  if (carry == MVR_OK) cur.n = ns;
  if (carry == MVR_OK && c->nn_mode != 0 && ns) carry = ensure_index(c, c->slots[ss]);
  if (carry == MVR_OK) { inherit_point_set(cur, c->slots[ss]); cur.segs.clear(); }
  const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float fin[16], tr[16];
  std::memcpy(fin, I, sizeof I); std::memcpy(tr, I, sizeof I);
  Criteria crit(p);
  double cur_mse = 0.0, fwdq = 0.0;
  int iters = 0, converged = 0, state = MVR_CONV_NOT, ncorr = 0, status = MVR_OK;
  Cloud &tgt = c->slots[ts];
  do {
    int local = carry;          // (a failure of the set-up, or of the transform that ended the iteration before)
    if (local == MVR_OK && c->inject_fail_at >= 0 && c->dist_pass == c->inject_fail_at) local = set_error(c, MVR_E_HIP, "injected failure of this rank's local work");
    ++c->dist_pass;
    if (local == MVR_OK) local = forward_keys_impl(c, cur, tgt, p->max_corr_dist, p->fma_dist, c->seq_keys);
    if (local != MVR_OK && ns) (void)hipMemsetAsync(c->seq_keys, 0x7F, ns * sizeof(long long), c->stream);      // "no neighbour here" (any key above every real one)
    if (ns) { if (int rc = comm_allreduce(c, c->seq_keys, ns, kReduceMinI64)) return rc; }
    if (local == MVR_OK) local = moments2_from_keys_impl(c, cur, tgt, c->seq_keys, p->max_corr_dist, p->use_reciprocal, p->fma_dist, origin, c->seq_row);
    // [32] of the row counts the ranks whose local work failed in this iteration
    c->h_moments[48] = local == MVR_OK ? 0.0 : 1.0;
    if (local != MVR_OK) (void)hipMemsetAsync(c->seq_row, 0, 32 * sizeof(double), c->stream);
    MVR_HIP_TRY(c, hipMemcpyAsync(c->seq_row + 32, c->h_moments + 48, sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (int rc = comm_allreduce(c, c->seq_row, 33, kReduceSumF64)) return rc;
    MVR_HIP_TRY(c, hipMemcpyAsync(c->h_moments, c->seq_row, 33 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (int rc = stream_wait(c)) return rc;
    const double *h = c->h_moments;
    if (h[32] > 0.0) return local != MVR_OK ? local : set_error(c, MVR_E_RCCL, "a peer's local work failed in this iteration");
    fwdq += (double)ns;
    ncorr = (int)std::llround(h[0]);
    if (h[0] < 3.0) { state = MVR_CONV_NO_CORRESPONDENCES; converged = 0; status = MVR_E_NOCORR; break; }
    mvr_pair_moments2_t m2;
    std::memset(&m2, 0, sizeof m2);
    m2.n = h[0];
    for (int k = 0; k < 3; ++k) { m2.origin[k] = origin[k]; m2.sp[k] = h[4 + k]; m2.sq[k] = h[7 + k]; }      // (the origin is a constant, not a sum over ranks)
    for (int k = 0; k < 6; ++k) { m2.spp[k] = h[10 + k]; m2.sqq[k] = h[16 + k]; }
    for (int k = 0; k < 9; ++k) m2.spq[k] = h[22 + k];
    m2.sum_d2 = h[31];
    mvr_pair_moments_t mom;
    if (int rc = mvr_moments_from_moments2(&m2, &mom)) return set_error(c, rc, "moments of the sharded align");
    if (int rc = mvr_umeyama_from_moments(&mom, tr, nullptr)) return set_error(c, rc, "Umeyama of the sharded align");
    cur_mse = h[31] / h[0];                          // mean of the correspondences' (f32) squared distances, as PCL has it
    carry = launch_transform_f32(c, cur.pts, cur.pts, ns, tr);          // (reported through the NEXT iteration's collectives, if there is one)
    cur.stale_coords();
    mvr_mat4f_mul(tr, fin, fin);
    ++iters;
    converged = crit.converged(tr, cur_mse, iters, &state) ? 1 : 0;
  } while (!converged);
  if (carry != MVR_OK) return carry;
  if (os >= 0) {            // output = final * (*input), from the ORIGINAL input (App. A.1)
    if (os != ss) { c->slots[os].n = 0; if (int rc = cloud_reserve(c, c->slots[os], ns, false)) return rc; }
    if (int rc = launch_transform_f32(c, c->slots[ss].pts, c->slots[os].pts, ns, fin)) return rc;
    c->slots[os].n = ns;
    if (os != ss) inherit_point_set(c->slots[os], c->slots[ss]);
    else c->slots[os].forget_pose();
    c->slots[os].stale_coords();
    c->slots[os].has_normals = false;
  }
  std::memcpy(T_out, fin, sizeof fin);
  if (st) {
    st->iterations = iters; st->converged = converged; st->state = state; st->n_corr = ncorr; st->mse = cur_mse;
    st->evals = 0.0; st->fwd_queries = fwdq; st->ms = now_ms() - t0;
  }
  return status;
}

// Registrator::registrationICP (registrator.cpp:526-588) with the growing target sharded over the ranks: view order
// 1, V-1, 2, ..., each view aligned against everything merged so far, pose_v <- T_icp * pose_v (:574), and
// `*target += transformed_source` (:576) as "every rank appends ITS slice of the aligned scan to its shard".
// Rank g of G owns points [n g / G, n (g + 1) / G) of every merged scan; global point numbers run scan after scan in
// merge order.  All ranks call this with the same arguments and get the same poses and logs.
API int mvr_seq_run_sharded(mvr_ctx *ctx, int n_views, const int *raw_slots, int target_slot, int source_slot, int out_slot,
                            const mvr_icp_params *p, const double origin[3], int repeat, double *poses, int *align_view, float *align_T,
                            mvr_icp_stats *align_stats, int *n_aligns)
{
  if (!ctx || n_views < 2 || !raw_slots || !p || !origin || !poses || repeat < 0) return MVR_E_ARG;
  if (!slot_ok(target_slot) || !slot_ok(source_slot) || !slot_ok(out_slot) || target_slot == source_slot || target_slot == out_slot || source_slot == out_slot)
    return MVR_E_ARG;
  for (int v = 0; v < n_views; ++v)
    if (!slot_ok(raw_slots[v]) || raw_slots[v] == target_slot || raw_slots[v] == source_slot || raw_slots[v] == out_slot) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  const size_t G = (size_t)(c->comm ? c->comm_world : 1), g = (size_t)(c->comm ? c->comm_rank : 0);
  std::vector<int> order;                                // registrator.cpp:530-541
  for (int i = 1; i < n_views / 2; ++i) { order.push_back(i); order.push_back(n_views - i); }
  if (n_views / 2 >= 1 && (order.empty() || order.back() != n_views / 2)) order.push_back(n_views / 2);
  int done = 0;
  if (n_aligns) *n_aligns = 0;
  for (int r = 0; r < repeat; ++r) {
    // target <- this rank's slice of the posed view 0
    size_t n0 = c->slots[raw_slots[0]].n, base = 0;
    if (int rc = mvr_cloud_transform(ctx, out_slot, raw_slots[0], poses)) return rc;
    if (int rc = mvr_cloud_clear(ctx, target_slot)) return rc;
    {
      const size_t lo = n0 * g / G, hi = n0 * (g + 1) / G;
      if (int rc = mvr_cloud_append_range(ctx, target_slot, out_slot, lo, hi - lo, base + lo)) return rc;
    }
    base += n0;
    for (int v : order) {
      if (int rc = mvr_cloud_transform(ctx, source_slot, raw_slots[v], poses + 16 * (size_t)v)) return rc;
      float T[16];
      mvr_icp_stats st;
      std::memset(&st, 0, sizeof st);
      const int rc = mvr_seq_align_sharded(ctx, source_slot, target_slot, out_slot, p, origin, T, &st);
      if (rc != MVR_OK && rc != MVR_E_NOCORR) return rc;      // (PCL's "not enough correspondences": the driver goes on, registrator.cpp:569-574)
      if (align_view) align_view[done] = v;
      if (align_T) std::memcpy(align_T + 16 * (size_t)done, T, sizeof T);
      if (align_stats) align_stats[done] = st;
      ++done;
      if (n_aligns) *n_aligns = done;
      double Td[16], P[16];
      for (int k = 0; k < 16; ++k) Td[k] = (double)T[k];
      mvr_mat4d_mul(Td, poses + 16 * (size_t)v, P);            // pose_v <- T_icp * pose_v
      std::memcpy(poses + 16 * (size_t)v, P, sizeof P);
      const size_t nv = c->slots[out_slot].n, lo = nv * g / G, hi = nv * (g + 1) / G;
      if (int rc2 = mvr_cloud_append_range(ctx, target_slot, out_slot, lo, hi - lo, base + lo)) return rc2;
      base += nv;
    }
  }
  return MVR_OK;
}

// Registrator::registrationICP (registrator.cpp:526-588) on ONE GPU, as one native call: `repeat` sweeps, each posing view 0 into
// the model, then views 1, V-1, 2, ... aligned against everything merged so far (mvr_icp_align), pose_v <- T_icp * pose_v (:574)
// and `*target += aligned source` (:576).  What a host language's loop over the same calls does (include/mvr/registrator.hpp:
// registrationICPDevice; tests / bench: Python), without that language between the calls.  reserve_points: the model's final
// size if known (0: the sum of the scans), reserved once so that the model grows in place.
API int mvr_seq_run(mvr_ctx *ctx, int n_views, const int *raw_slots, int target_slot, int source_slot, int out_slot,
                    const mvr_icp_params *p, int repeat, double *poses, int *align_view, float *align_T, mvr_icp_stats *align_stats,
                    int *n_aligns)
{
  if (!ctx || n_views < 2 || !raw_slots || !p || !poses || repeat < 0) return MVR_E_ARG;
  if (!slot_ok(target_slot) || !slot_ok(source_slot) || !slot_ok(out_slot) || target_slot == source_slot || target_slot == out_slot || source_slot == out_slot)
    return MVR_E_ARG;
  for (int v = 0; v < n_views; ++v)
    if (!slot_ok(raw_slots[v]) || raw_slots[v] == target_slot || raw_slots[v] == source_slot || raw_slots[v] == out_slot) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  std::vector<int> order;                                // registrator.cpp:530-541
  for (int i = 1; i < n_views / 2; ++i) { order.push_back(i); order.push_back(n_views - i); }
  if (n_views / 2 >= 1 && (order.empty() || order.back() != n_views / 2)) order.push_back(n_views / 2);
  size_t total = 0;
  for (int v = 0; v < n_views; ++v) total += c->slots[raw_slots[v]].n;
  int done = 0;
  if (n_aligns) *n_aligns = 0;
  for (int r = 0; r < repeat; ++r) {
    if (int rc = mvr_cloud_transform(ctx, target_slot, raw_slots[0], poses)) return rc;       // :562
    if (int rc = mvr_cloud_reserve(ctx, target_slot, total)) return rc;
    for (int v : order) {
      // (the model grew by the previous align's output: the launch that poses this source refreshes the model's tail on the way)
      c->refresh_rider = c->seq_rider ? &c->slots[target_slot] : nullptr;
      const int rct = mvr_cloud_transform(ctx, source_slot, raw_slots[v], poses + 16 * (size_t)v);              // :565
      c->refresh_rider = nullptr;
      if (rct != MVR_OK) return rct;
      float T[16];
      mvr_icp_stats st;
      std::memset(&st, 0, sizeof st);
      const int rc = mvr_icp_align(ctx, source_slot, target_slot, out_slot, p, T, &st);                          // :566-569
      if (rc != MVR_OK && rc != MVR_E_NOCORR) return rc;      // (PCL's "not enough correspondences": the driver goes on, registrator.cpp:569-574)
      if (align_view) align_view[done] = v;
      if (align_T) std::memcpy(align_T + 16 * (size_t)done, T, sizeof T);
      if (align_stats) align_stats[done] = st;
      ++done;
      if (n_aligns) *n_aligns = done;
      double Td[16], P[16];
      for (int k = 0; k < 16; ++k) Td[k] = (double)T[k];
      mvr_mat4d_mul(Td, poses + 16 * (size_t)v, P);            // :573-574
      std::memcpy(poses + 16 * (size_t)v, P, sizeof P);
      if (int rc2 = mvr_cloud_append(ctx, target_slot, out_slot)) return rc2;                                    // :576
    }
  }
  return MVR_OK;
}

API int mvr_fitness(mvr_ctx *ctx, int is, int ts, const float T[16], double max_range, int fma, double *score)
{
  if (!ctx || !slot_ok(is) || !slot_ok(ts) || !T || !score) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  *score = DBL_MAX;
  const size_t ns = c->slots[is].n;
  if (ns == 0) return MVR_OK;
  Cloud &tmp = c->slots[kScratchTmp];
  tmp.n = 0;
  if (int rc = cloud_reserve(c, tmp, ns, false)) return rc;
  if (int rc = launch_transform_f32(c, c->slots[is].pts, tmp.pts, ns, T)) return rc;
  tmp.n = ns;
  inherit_point_set(tmp, c->slots[is]);
  Cloud &t = c->slots[ts];
  if (int rc = ensure(c, c->keys, c->keys_cap, ns)) return rc;
  if (int rc = launch_fill_u64(c, c->keys, ns, kKeyInit)) return rc;
  if (c->nn_mode != 0 && t.n > 0) {
    if (int rc = ensure_index(c, tmp)) return rc;
    if (int rc = ensure_index(c, t)) return rc;
    if (!c->slots[is].order) c->slots[is].order = tmp.order;     // keep the sort for the next call
    if (int rc = launch_nn_cull(c, tmp, 0, ns, nullptr, t, cap_from_max2(max_range), fma != 0, c->keys)) return rc;
  } else {
    if (int rc = launch_nn(c, tmp.pts, 0, ns, nullptr, nullptr, t.pts, t.n, fma != 0, c->keys)) return rc;
  }
  if (int rc = launch_fitness(c, c->keys, ns, max_range, c->moments)) return rc;
  if (int rc = read_moments(c, 2)) return rc;
  if (c->h_moments[1] > 0) *score = c->h_moments[0] / c->h_moments[1];
  return MVR_OK;
}

// ----------------------------------------------------------- instrumentation

API int mvr_ctx_tune(mvr_ctx *ctx, const char *key, int value)
{
  if (!ctx || !key) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  if (!std::strcmp(key, "nn_q")) c->nn_q = value;
  else if (!std::strcmp(key, "nn_sub")) c->nn_sub = value;
  else if (!std::strcmp(key, "nn_blocks_per_cu")) c->nn_blocks_per_cu = value;
  else if (!std::strcmp(key, "nn_mode")) c->nn_mode = value;
  else if (!std::strcmp(key, "cull_q")) c->cull_q = value;
  else if (!std::strcmp(key, "cull_w")) c->cull_w = value;
  else if (!std::strcmp(key, "cull_slices")) c->cull_slices = value;
  else if (!std::strcmp(key, "seed_forward")) c->seed_forward = value != 0;
  else if (!std::strcmp(key, "lazy_super")) { if (value < 0 || value > 1) return MVR_E_ARG; c->lazy_super = value; }
  else if (!std::strcmp(key, "fused_mark")) { if (value < 0 || value > 2) return MVR_E_ARG; c->fused_mark = value; }
  else if (!std::strcmp(key, "ring_search")) c->ring_search = value;
  else if (!std::strcmp(key, "seq_search")) { if (value < 0 || value > 3) return MVR_E_ARG; c->seq_search = value; }
  else if (!std::strcmp(key, "unseeded_grid")) { if (value < 0 || value > 1) return MVR_E_ARG; c->unseeded_grid = value; }
  else if (!std::strcmp(key, "seq_model_tail")) { if (value < 0 || value > 1) return MVR_E_ARG; c->seq_model_tail = value; }
  else if (!std::strcmp(key, "seq_cell_points")) { if (value < 1) return MVR_E_ARG; c->seq_cell_points = value; }
  else if (!std::strcmp(key, "seq_seed")) {          // 0 also forgets what the aligns so far have left behind
    c->seq_seed = value != 0;
    if (!c->seq_seed) {
      MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
      for (auto &kv : c->seq_seeds) if (kv.second.d) (void)hipFree(kv.second.d);
      c->seq_seeds.clear();
    }
  }
  else if (!std::strcmp(key, "parts_lanes")) { if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return MVR_E_ARG; c->parts_lanes = value; }
  else if (!std::strcmp(key, "parts_max_rows")) { if (value < 1) return MVR_E_ARG; c->parts_max_rows = value; }
  else if (!std::strcmp(key, "grid_light_rows")) { if (value < 1) return MVR_E_ARG; c->grid_light_rows = value; }
  else if (!std::strcmp(key, "grid_light_rows_lone")) { if (value < 1) return MVR_E_ARG; c->grid_light_rows_lone = value; }
  else if (!std::strcmp(key, "align_spin")) c->align_spin = value != 0;
  else if (!std::strcmp(key, "seq_rider")) c->seq_rider = value != 0;
  else if (!std::strcmp(key, "reduce_rows")) { if (value < 0 || value > 1024) return MVR_E_ARG; c->reduce_rows = value; }
  else if (!std::strcmp(key, "grid_probe")) c->grid_probe = value != 0;
  else if (!std::strcmp(key, "grid_probe_rows")) { if (value < 1) return MVR_E_ARG; c->grid_probe_rows = value; }
  else if (!std::strcmp(key, "setup_first")) c->setup_first = value != 0;
  else if (!std::strcmp(key, "pose_prep_launch")) c->pose_prep_launch = value != 0;
  else if (!std::strcmp(key, "grid_wide")) c->grid_wide = value != 0;
  else if (!std::strcmp(key, "grid_debug")) c->grid_debug = value != 0;
  else if (!std::strcmp(key, "cull_list")) c->cull_list = value != 0;
  else if (!std::strcmp(key, "grid_sets")) { if (value < 0 || value > 2) return MVR_E_ARG; c->grid_sets = value; }
  else if (!std::strcmp(key, "grid_tail")) c->grid_tail = value != 0;
  else if (!std::strcmp(key, "cull_list_w")) { if (value != 1 && value != 2 && value != 4) return MVR_E_ARG; c->cull_list_w = value; }
  else if (!std::strcmp(key, "grid_cluster")) { if (value < 1) return MVR_E_ARG; c->grid_cluster = value; }
  else if (!std::strcmp(key, "grid_wide_waves")) { if (value < 1 || value > 64) return MVR_E_ARG; c->grid_wide_waves = value; }
  else if (!std::strcmp(key, "grid_stage")) { if (value < 0 || value > 2) return MVR_E_ARG; c->grid_stage = value; }
  else if (!std::strcmp(key, "grid_stage_lone")) { if (value < 0 || value > 2) return MVR_E_ARG; c->grid_stage_lone = value; }      // a lone pair's launches: 0 never staged, 1 (default) the ones over a scan's own query order, 2 all
  else if (!std::strcmp(key, "grid_stage_stat")) {
    if (value && !c->stage_stat) { MVR_HIP_TRY(c, hipMalloc(&c->stage_stat, 64 * 64 * sizeof(unsigned long long))); }
    if (c->stage_stat) { MVR_HIP_TRY(c, hipStreamSynchronize(c->stream)); MVR_HIP_TRY(c, hipMemset(c->stage_stat, 0, 64 * 64 * sizeof(unsigned long long))); }
    if (!value && c->stage_stat) { (void)hipFree(c->stage_stat); c->stage_stat = nullptr; }
  }
  else if (!std::strcmp(key, "order_batch")) c->order_batch = value != 0;
  else if (!std::strcmp(key, "rim_cert_um")) { if (value < 0) return MVR_E_ARG; c->rim_cert_um = value; }
  else if (!std::strcmp(key, "seed_delta_um")) { if (value < 0) return MVR_E_ARG; c->seed_delta_um = value; }
  else if (!std::strcmp(key, "grid_lanes")) { if (value != 1 && value != 2 && value != 4 && value != 8) return MVR_E_ARG; c->grid_lanes = value; }
  else if (!std::strcmp(key, "grid_index")) { if (value < 0 || value > 2) return MVR_E_ARG; c->grid_index = value; }
  else if (!std::strcmp(key, "grid_cell_points")) { if (value < 1) return MVR_E_ARG; c->grid_cell_points = value; }
  else if (!std::strcmp(key, "pipeline")) c->pipeline = value != 0;
  else if (!std::strcmp(key, "pipeline_multi_rank")) c->pipeline_multi_rank = value != 0;
  else if (!std::strcmp(key, "wait_timeout_ms")) { if (value < 1) return MVR_E_ARG; c->wait_timeout_ms = value; }
  else if (!std::strcmp(key, "inject_fail_pass")) { c->inject_fail_at = value; c->dist_pass = 0; }
  else if (!std::strcmp(key, "inject_stall_pass")) { c->inject_stall_at = value; c->dist_pass = 0; }
  else if (!std::strcmp(key, "pair_fused")) c->pair_fused = value != 0;
  else if (!std::strcmp(key, "posed_refresh")) c->posed_refresh = value != 0;
  else if (!std::strcmp(key, "pair_groups")) { if (value < 1 || value > 8) return MVR_E_ARG; c->pair_groups = value; }
  else if (!std::strcmp(key, "inplace_ratio")) c->inplace_ratio = value;
  else if (!std::strcmp(key, "pair_streams")) c->pair_streams = value < 1 ? 1 : (value > 16 ? 16 : value);
  else return MVR_E_ARG;
  return MVR_OK;
}

API int mvr_debug_order(mvr_ctx *ctx, int slot, uint32_t *perm, size_t cap, size_t *n)
{
  if (!ctx || !slot_ok(slot) || !n) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  const Cloud &cl = c->slots[slot];
  *n = cl.order ? cl.order->n : 0;
  if (cl.order && perm && cap >= cl.order->n) {
    MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
    MVR_HIP_TRY(c, hipMemcpy(perm, cl.order->perm, cl.order->n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  }
  return MVR_OK;
}

API int mvr_ctx_pass_log(mvr_ctx *ctx, double *ms, int cap, int *n)
{
  if (!ctx || !n || cap < 0 || (cap && !ms)) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  *n = (int)c->pass_ms.size();
  for (int k = 0; k < cap && k < *n; ++k) ms[k] = c->pass_ms[(size_t)k];
  return MVR_OK;
}

API int mvr_ctx_stat(mvr_ctx *ctx, const char *key, double *value)
{
  if (!ctx || !key || !value) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  if (!std::strcmp(key, "piped_passes")) *value = (double)c->piped_passes;
  else if (!std::strcmp(key, "fused_passes")) *value = (double)c->fused_passes;
  else if (!std::strcmp(key, "blocking_events")) *value = (double)c->blocking_events;
  else if (!std::strcmp(key, "pipeline")) *value = (double)c->pipeline;
  else if (!std::strncmp(key, "stage_", 6) && c->stage_stat) {      // staged walk diagnostics: waves by outcome since grid_stage_stat was set
    unsigned long long h[64] = {0};
    std::vector<unsigned long long> raw(64 * 64);
    MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
    MVR_HIP_TRY(c, hipMemcpy(raw.data(), c->stage_stat, raw.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int sh = 0; sh < 64; ++sh) for (int k = 0; k < 64; ++k) h[k] += raw[(size_t)sh * 64 + k];
    if (!std::strncmp(key, "stage_raw_", 10)) { const int k = std::atoi(key + 10); if (k < 0 || k > 63) return MVR_E_ARG; *value = (double)h[k]; }
    else if (!std::strcmp(key, "stage_staged")) *value = (double)(h[1] + h[9]);
    else if (!std::strcmp(key, "stage_rows")) *value = (double)(h[2] + h[10]);
    else if (!std::strcmp(key, "stage_width")) *value = (double)(h[3] + h[11]);
    else if (!std::strcmp(key, "stage_points")) *value = (double)(h[4] + h[12]);
    else return MVR_E_ARG;
  }
  else return MVR_E_ARG;
  return MVR_OK;
}

API int mvr_debug_counters(mvr_ctx *ctx, uint64_t out[4], int reset)
{
  if (!ctx || !out) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  std::vector<uint64_t> h(2 * kEvalRegion);
  MVR_HIP_TRY(c, hipMemcpyAsync(h.data(), c->evals, 2 * kEvalRegion * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
  out[0] = out[1] = out[2] = out[3] = 0;
  for (int k = 0; k < kEvalShards; ++k) {
    out[0] += h[(size_t)k * kEvalStride];
    out[1] += h[kEvalRegion + (size_t)k * kEvalStride];
    out[2] = std::max<uint64_t>(out[2], h[kEvalRegion + (size_t)k * kEvalStride + 1]);
    out[3] = std::max<uint64_t>(out[3], h[kEvalRegion + (size_t)k * kEvalStride + 2]);
  }
  if (std::getenv("MVR_STAMP_DUMP")) {     // diagnostic builds (-DMVR_STAMP): per-phase cycle sums of the culled kernel
    uint64_t ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < kEvalShards; ++k) for (int j = 0; j < 10; ++j) ph[j] += h[kEvalRegion + (size_t)k * kEvalStride + 4 + j];
    if (ph[5]) std::fprintf(stderr, "[mvr stamp] prologue %.0f epilogue %.0f lifetime %.0f cycles, %.2f us (100 MHz clock)\n", (double)ph[6] / ph[5],
                            (double)ph[7] / ph[5], (double)ph[8] / ph[5], (double)ph[9] / ph[5] / 100.0);
    std::fprintf(stderr, "[mvr stamp] waves %llu  cycles/wave: stage %.0f advance %.0f process %.0f revalidate %.0f  loop-total %.0f\n",
                 (unsigned long long)ph[5], ph[5] ? (double)ph[0] / ph[5] : 0.0, ph[5] ? (double)ph[1] / ph[5] : 0.0,
                 ph[5] ? (double)ph[2] / ph[5] : 0.0, ph[5] ? (double)ph[3] / ph[5] : 0.0, ph[5] ? (double)ph[4] / ph[5] : 0.0);
  }
  if (const char *path = std::getenv("MVR_STAMP_TRACE")) {     // per-block records of the LAST culled launch -> text file
    std::vector<uint64_t> tr(kTraceRec * kTraceBlocks);
    MVR_HIP_TRY(c, hipMemcpy(tr.data(), c->evals + 2 * kEvalRegion, tr.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (FILE *f = std::fopen(path, "w")) {
      for (size_t k = 0; k < kTraceBlocks; ++k)
        if (tr[kTraceRec * k + 1]) {
          std::fprintf(f, "%zu", k);
          for (size_t j = 0; j < kTraceRec; ++j) std::fprintf(f, " %llu", (unsigned long long)tr[kTraceRec * k + j]);
          std::fprintf(f, "\n");
        }
      std::fclose(f);
    }
  }
  if (std::getenv("MVR_STAMP_DUMP")) {
    std::fprintf(stderr, "[mvr stamp] waves by tiles evaluated (count:mean kcycles):");
    for (int k = 0; k < 32; ++k) {
      const uint64_t n = h[kEvalRegion + (size_t)(32 + k) * kEvalStride + 3], cyc = h[kEvalRegion + (size_t)k * kEvalStride + 3];
      if (n) std::fprintf(stderr, " %d=%llu:%.0f", k, (unsigned long long)n, (double)cyc / n / 1000.0);
    }
    std::fprintf(stderr, "\n");
    std::fprintf(stderr, "[mvr stamp] wave end-time histogram (5 us bins, all launches since reset):");
    for (int k = 0; k < kEvalShards; ++k) std::fprintf(stderr, " %llu", (unsigned long long)h[kEvalRegion + (size_t)k * kEvalStride + 14]);
    std::fprintf(stderr, "\n");
  }
  if (reset) MVR_HIP_TRY(c, hipMemsetAsync(c->evals, 0, 2 * kEvalRegion * sizeof(uint64_t), c->stream));
  return MVR_OK;
}

API int mvr_prof_enable(mvr_ctx *ctx, int on)
{
  if (!ctx) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  if (int rc = prof_drain(c)) return rc;                    // settle what the previous mode collected
  c->prof = on != 0;
  c->prof_mask = (on == 2) ? ((1u << MVR_K_NN) | (1u << MVR_K_NN_GRID) | (1u << MVR_K_NN_WIDE)) : ~0u;       // 2: time the search kernels only (cheapest)
  c->prof_totals = on == 2;
  for (Ctx *w : c->workers) { w->prof = c->prof; w->prof_mask = c->prof_mask; w->prof_totals = c->prof_totals; w->prof_totals = c->prof_totals; }
  return MVR_OK;
}

API int mvr_prof_reset(mvr_ctx *ctx)
{
  if (!ctx) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  if (int rc = prof_drain(c)) return rc;
  for (int k = 0; k < MVR_K_COUNT; ++k) { c->prof_launches[k] = 0; c->prof_ms[k] = 0; c->prof_work[k] = 0; }
  return MVR_OK;
}

API int mvr_prof_get(mvr_ctx *ctx, int family, uint64_t *launches, double *ms, double *work)
{
  if (!ctx || family < 0 || family >= MVR_K_COUNT) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  if (int rc = prof_drain(c)) return rc;
  if (launches) *launches = c->prof_launches[family];
  if (ms) *ms = c->prof_ms[family];
  if (work) *work = c->prof_work[family];
  return MVR_OK;
}

}  // extern "C"
